"""Run a named workload end to end on the GPU, report throughput and per-kernel times,
and compare a sample of queries with the CPU oracle (same index, imported)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flye_amd import config, gpu, workloads
from oracle import oracle as O

name = sys.argv[1]
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
t = time.time()
if name == "ecoli_pb50":
    rs, min_ovlp, preset = workloads.ecoli_pb50(scale=scale)
elif name == "dmel_ont30":
    rs, min_ovlp, preset = workloads.dmel_ont30(scale=scale)
elif name == "hifi30":
    rs, min_ovlp, preset = workloads.hifi30(genome_len=int(4_640_000 * scale))
else:
    raise SystemExit("unknown workload")
cfg = config.preset(preset)
if name == "hifi30":
    # the pipeline passes --hifi-error 0.003 (flye/assembly/assemble.py:58-60) for real HiFi reads; these
    # synthetic reads carry 0.3 % error EACH (0.6 % pairwise), so the gate is set where overlaps survive
    cfg["assemble_ovlp_divergence"] = 0.01
print(f"{name} scale {scale}: {rs.n} reads, {rs.total_bases/1e6:.1f} Mbp, min_ovlp {min_ovlp}, gen {time.time()-t:.1f}s", flush=True)
k = int(cfg["kmer_size"])
ctx = gpu.Context(k, 0)
ctx.set_reads(rs)
vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
st = vi.build(cfg)
print("index:", {a: st[a] for a in ("selected_kmers", "index_entries", "repetitive_kmers", "repetitive_frequency", "sample_rate")},
      "build %.3f s" % st["build_seconds"], flush=True)
# the assemble stage's detector always runs with minimumOverlap = 1000 (main_assemble.cpp:174, :231)
det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg, min_overlap=config.DETECTOR_MIN_OVERLAP)
if name == "hifi30":
    det.p.max_divergence = cfg["assemble_ovlp_divergence"]
q = np.arange(0, 2 * rs.n, 2, dtype=np.uint32)
for rep in range(2):
    t = time.time(); res = det.getSeqOverlapsBatch(q); dt = time.time() - t
    print(f"pass {rep}: {dt*1e3:.1f} ms wall, {res.device_seconds*1e3:.1f} ms device -> {rs.total_bases/dt/1e9:.3f} Gbp/s; "
          f"{len(res.recs)} overlaps, hits/bp {res.seed_hits/res.query_bp:.2f}, dp/bp {res.dp_elements/res.query_bp:.2f}", flush=True)
print({k_: round(v[0]*1e3, 2) for k_, v in sorted(ctx.kernel_times().items(), key=lambda kv: -kv[1][0])}, flush=True)
# sampled parity
ex = vi.export()
o = O.Oracle(k)
o.set_reads(rs)
o.import_index(O.IndexExport(ex.keys, ex.key_off, ex.entries, ex.repetitive), vi.getSampleRate())
rng = np.random.default_rng(1)
sample = np.sort(rng.choice(rs.n, size=min(rs.n, int(sys.argv[3]) if len(sys.argv) > 3 else 200), replace=False))
op = O.detector_params(cfg, min_overlap=config.DETECTOR_MIN_OVERLAP, max_divergence=det.p.max_divergence)
t = time.time(); ores = o.overlaps(op, (2 * sample).astype(np.uint32)); to = time.time() - t
got = np.concatenate([res.of(int(i)) for i in sample]) if len(sample) else res.recs[:0]
same = (len(got) == len(ores.recs) and all(np.array_equal(got[f], ores.recs[f]) for f in
        ("cur_id", "ext_id", "cur_begin", "cur_end", "ext_begin", "ext_end", "score", "edit_distance"))
        and np.array_equal(got["seq_divergence"].view(np.uint32), ores.recs["seq_divergence"].view(np.uint32)))
print(f"sample of {len(sample)} queries: {len(got)} records, identical to oracle: {same} (oracle {to:.1f} s, {ores.query_bp/to/1e6:.1f} Mbp/s on {o.threads} threads)")
sys.exit(0 if same else 1)
