import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["FG_SORT_TRACE"] = "1"
import numpy as np
from flye_amd import config, gpu, workloads
rs, mo, preset = workloads.ecoli_pb50()
cfg = config.preset(preset)
ctx = gpu.Context(17, 0); ctx.set_reads(rs)
vi = gpu.VertexIndex(ctx, 1.0); vi.build(cfg)
det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
q = np.arange(0, 2 * rs.n, 2, dtype=np.uint32)
det.getSeqOverlapsBatch(q)
sys.stderr.write("---- second pass ----\n")
det.getSeqOverlapsBatch(q)
kt = ctx.kernel_times()
for k in sorted(kt):
    if "sort" in k: print(k, round(kt[k][0]*1e3, 3), kt[k][1])
