import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flye_amd import config, gpu, workloads
rs, min_ovlp, preset = workloads.ecoli_pb50() if len(sys.argv) < 2 else workloads.dmel_ont30(scale=float(sys.argv[1]))
cfg = config.preset(preset)
ctx = gpu.Context(17, 0); ctx.set_reads(rs)
vi = gpu.VertexIndex(ctx, 1.0); vi.build(cfg)
det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
q = np.arange(0, 2 * rs.n, 2, dtype=np.uint32)
ref = None
for sm in [0, 2048, 8192, 32768, 1 << 30]:
    os.environ["FG_SORT_STREAM_MAX"] = str(sm)
    r = det.getSeqOverlapsBatch(q)
    kt = ctx.kernel_times()
    h = hash(r.recs.tobytes())
    if ref is None: ref = h
    print("streamMax", sm, "same" if h == ref else "DIFFERENT", {k: round(v[0]*1e3, 2) for k, v in kt.items() if "sort" in k}, flush=True)
