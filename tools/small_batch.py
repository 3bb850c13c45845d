"""Device time of small fg_overlaps calls on the bench workload (what one getSeqOverlaps-sized request costs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flye_amd import config, gpu, workloads
rs, mo, preset = workloads.ecoli_pb50()
cfg = config.preset(preset)
ctx = gpu.Context(17, 0); ctx.set_reads(rs)
vi = gpu.VertexIndex(ctx, 1.0); vi.build(cfg)
det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg, min_overlap=config.DETECTOR_MIN_OVERLAP)
for n in (1, 16, 64, 256, 1024, 4096):
    q = np.arange(2000, 2000 + 2 * n, 2, dtype=np.uint32)
    det.getSeqOverlapsBatch(q)
    r = det.getSeqOverlapsBatch(q)
    kt = ctx.kernel_times()
    top = sorted(((v[0] * 1e3, k, v[1]) for k, v in kt.items()), reverse=True)[:9]
    print(f"{n:5d} reads: device {r.device_seconds * 1e3:.2f} ms;", ", ".join(f"{k} {t:.2f} ({c})" for t, k, c in top), flush=True)
