import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flye_amd import config, gpu, workloads
rs, min_ovlp, preset = workloads.ecoli_pb50()
cfg = config.preset(preset)
ctx = gpu.Context(17, 0); ctx.set_reads(rs)
vi = gpu.VertexIndex(ctx, 1.0); vi.build(cfg)
det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
q = np.arange(0, 2 * rs.n, 2, dtype=np.uint32)
for ab in [int(x) for x in (sys.argv[1:] or ['0','1','2','4','8','15','0'])]:
    os.environ["FG_ABLATE"] = str(ab)
    r = det.getSeqOverlapsBatch(q)
    kt = ctx.kernel_times()
    print("ablate", ab, "dp_groups", r.dp_groups, "dp_el", r.dp_elements, {k: round(v[0]*1e3, 2) for k, v in kt.items() if "chain" in k or "sort" in k or "group" in k or "dp_list" in k}, flush=True)
