"""What ONE rank of an N-rank weak-scaling run does, measured on one GPU: genome x N, the full
index, queries = shard 0 of N (flye_amd/dist.py).  python tools/rank_share.py N"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flye_amd import config, dist, gpu, workloads
N = int(sys.argv[1])
rs, min_ovlp, preset = workloads.ecoli_pb50(seed=12345, scale=float(N))
cfg = config.preset(preset)
ctx = gpu.Context(int(cfg["kmer_size"]), 0); ctx.set_reads(rs)
vi = gpu.VertexIndex(ctx, 1.0); st = vi.build(cfg)
# the assemble stage's detector always runs with minimumOverlap = 1000 (main_assemble.cpp:174, :231)
det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg, min_overlap=config.DETECTOR_MIN_OVERLAP)
q = dist.shard_queries(rs.n, 0, N)
bp = int(rs.length[(q // 2).astype(np.int64)].sum())
det.getSeqOverlapsBatch(q)
ts = []
for _ in range(3):
    t = time.perf_counter(); res = det.getSeqOverlapsBatch(q); ts.append(time.perf_counter() - t)
t = sorted(ts)[1]
kt = ctx.kernel_times()
print(f"N={N}: {rs.n} reads, rank share {len(q)} queries / {bp/1e6:.1f} Mbp: {t*1e3:.1f} ms -> {bp/t/1e9:.3f} Gbp/s per rank; "
      f"hits/bp {res.seed_hits/res.query_bp:.2f}; " + str({k: round(v[0]*1e3, 1) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])[:8]}))
