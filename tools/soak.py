"""Repeat the overlap pass of the bench workload and compare every pass's records with the first one's (the stage
runs on three streams with event joins: a missing dependency would show as a pass that differs)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flye_amd import config, gpu, workloads
n_pass = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs, mo, preset = workloads.ecoli_pb50()
cfg = config.preset(preset)
ctx = gpu.Context(17, 0); ctx.set_reads(rs)
vi = gpu.VertexIndex(ctx, 1.0); vi.build(cfg)
det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg, min_overlap=config.DETECTOR_MIN_OVERLAP)
q = np.arange(0, 2 * rs.n, 2, dtype=np.uint32)
first = None
for i in range(n_pass):
    qq = q if i % 3 else q[: len(q) // (1 + i % 4)]          # vary the batch size too (buffers are reused across calls)
    r = det.getSeqOverlapsBatch(qq)
    h = hashlib.sha256(r.recs.tobytes() + r.query_off.tobytes() + r.stats.tobytes()).hexdigest()
    key = len(qq)
    if first is None:
        first = {}
    if key in first:
        assert first[key] == h, f"pass {i} ({key} queries) differs"
    else:
        first[key] = h
print(f"{n_pass} passes, {len(first)} batch sizes: all repeats identical")
