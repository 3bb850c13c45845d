"""Where does the probe step go on a table far beyond the caches?  D. melanogaster-like workload, the probe kernels'
times with the probes in query order, partitioned by table region, and with parts of the sorted probe switched off
(FG_ABLATE_PROBE: 1 = no scatter of the results, 2 = no list-bound reads; results are then wrong).
    python tools/probe_ablate.py [scale]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flye_amd import config, gpu, workloads
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
rs, mo, preset = workloads.dmel_ont30(scale=scale)
cfg = config.preset(preset)
ctx = gpu.Context(17, 0); ctx.set_reads(rs)
vi = gpu.VertexIndex(ctx, 1.0); st = vi.build(cfg)
print(f"dmel x{scale}: {rs.n} reads {rs.total_bases / 1e9:.2f} Gbp, {st['selected_kmers'] / 1e6:.0f} M keys, build {st['build_seconds']:.2f} s", flush=True)
det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
q = np.arange(0, 2 * rs.n, 2, dtype=np.uint32)[: rs.n // 4]      # a quarter of the reads: the probe step is what is looked at
det.getSeqOverlapsBatch(q)
for part, abl in ((0, 0), (1, 0), (1, 1), (1, 2), (1, 3)):
    os.environ["FG_PROBE_PARTITION"] = str(part)
    os.environ["FG_ABLATE_PROBE"] = str(abl)
    t = time.time(); res = det.getSeqOverlapsBatch(q); dt = time.time() - t
    kt = ctx.kernel_times()
    pr = {k: round(v[0] * 1e3, 1) for k, v in kt.items() if k.startswith("k_probe") or k == "k_fill"}
    print(f"partition {part} ablate {abl}: pass {dt * 1e3:.0f} ms, {pr}, sum probe {sum(v for k, v in pr.items() if k != 'k_fill'):.0f} ms", flush=True)
