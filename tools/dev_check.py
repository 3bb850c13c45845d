"""Developer check: HIP path vs oracle on one synthetic set (prints diagnostics)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flye_amd import synth, config, gpu
from oracle import oracle as O

kind = sys.argv[1] if len(sys.argv) > 1 else "pb_raw"
preset = sys.argv[2] if len(sys.argv) > 2 else "raw"
glen = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
cov = int(sys.argv[4]) if len(sys.argv) > 4 else 30
extra = dict(a.split("=") for a in sys.argv[5:])
rs = synth.simulate(seed=int(extra.get("seed", 777)), genome_len=glen, coverage=cov, kind=kind,
                    n_homopolymers=int(extra.get("hp", 8)), n_tandems=int(extra.get("tr", 8)),
                    n_repeat_families=int(extra.get("rep", 4)))
rs = rs.filter_min_len(int(extra.get("minlen", 1000)))
cfg = config.preset(preset)
k = int(cfg["kmer_size"])
print("reads", rs.n, "bases", rs.total_bases, flush=True)

o = O.Oracle(k, threads=8)
o.set_reads(rs)
t = time.time(); ost = o.build_index(cfg); print("oracle build %.2fs" % (time.time() - t), ost, flush=True)
oex = o.export_index()

ctx = gpu.Context(k, 0)
ctx.set_reads(rs)
vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
t = time.time(); gst = vi.build(cfg); print("gpu build %.2fs" % (time.time() - t), gst, flush=True)
print("build kernels:", {k_: round(v[0] * 1e3, 3) for k_, v in ctx.kernel_times().items()})
gex = vi.export()
same = (np.array_equal(gex.keys, oex.keys) and np.array_equal(gex.key_off, oex.key_off)
        and np.array_equal(gex.entries, oex.entries) and np.array_equal(gex.repetitive, oex.repetitive))
print("INDEX SAME:", same, len(gex.keys), len(oex.keys), len(gex.entries), len(oex.entries),
      len(gex.repetitive), len(oex.repetitive))
for f in ("total_kmers", "selected_kmers", "index_entries", "repetitive_kmers", "repetitive_frequency"):
    if gst[f] != ost[f]:
        print("  stat differs", f, gst[f], ost[f])
if np.float32(gst["sample_rate"]).tobytes() != np.float32(ost["sample_rate"]).tobytes():
    print("  sample_rate differs", gst["sample_rate"], ost["sample_rate"])

rcq = int(extra.get("rc", 0)); mo = int(extra.get("maxovlp", 0)); fl = int(extra.get("forcelocal", 0))
q = np.arange(1 if rcq else 0, 2 * rs.n, 2, dtype=np.uint32)
if extra.get("mixq"):
    q = np.arange(0, 2 * rs.n, dtype=np.uint32)
det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
if float(extra.get("maxdiv", 1.0)) != 1.0:
    det.p.max_divergence = float(extra["maxdiv"])
op = O.detector_params(cfg, max_divergence=float(extra.get("maxdiv", 1.0)))
t = time.time(); ores = o.overlaps(op, q, max_overlaps=mo, force_local=fl); to = time.time() - t
t = time.time(); gres = det.getSeqOverlapsBatch(q, forceLocal=fl, maxOverlaps=mo); tg = time.time() - t
print("oracle ovlp %.3fs (%d recs)  gpu ovlp %.3fs wall, %.3fs device (%d recs)" %
      (to, len(ores.recs), tg, gres.device_seconds, len(gres.recs)))
print("ovlp kernels (ms):", {k_: round(v[0] * 1e3, 3) for k_, v in ctx.kernel_times().items()})
print("counters gpu", gres.query_kmers, gres.seed_hits, gres.dp_groups, gres.dp_elements)
print("counters ora", ores.query_kmers, ores.seed_hits, ores.dp_groups, ores.dp_elements)
ol, gl = ores.lines(), gres.lines()
print("OVERLAPS SAME:", ol == gl, " stats same:", np.array_equal(ores.stats.view(np.uint32), gres.stats.view(np.uint32)))
if ol != gl:
    so, sg = set(ol), set(gl)
    print("  only oracle:", len(so - sg), " only gpu:", len(sg - so))
    for x in sorted(so - sg)[:5]: print("   O", x)
    for x in sorted(sg - so)[:5]: print("   G", x)
    sys.exit(1)
if not same:
    sys.exit(1)
