"""Developer check: minimizer index build + overlaps without base-level alignment."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flye_amd import synth, config, gpu
from oracle import oracle as O
ok = True
for preset, kind, seed in [("hifi", "hifi", 11), ("corrected", "hifi03", 12), ("hifi", "pb_raw", 13), ("corrected", "hifi", 14)]:
    rs = synth.simulate(seed=seed, genome_len=80_000, coverage=25, kind=kind, n_homopolymers=30, n_tandems=40).filter_min_len(1000)
    cfg = config.preset(preset)
    if os.environ.get("NO_NUCL"): cfg["reads_base_alignment"] = 0.0
    o = O.Oracle(17); o.set_reads(rs); ost = o.build_index(cfg); oex = o.export_index()
    ctx = gpu.Context(17, 0); ctx.set_reads(rs)
    vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"]))); gst = vi.build(cfg); gex = vi.export()
    same = (np.array_equal(gex.keys, oex.keys) and np.array_equal(gex.key_off, oex.key_off)
            and np.array_equal(gex.entries, oex.entries) and np.array_equal(gex.repetitive, oex.repetitive))
    sr = np.float32(gst["sample_rate"]).tobytes() == np.float32(ost["sample_rate"]).tobytes()
    print(preset, kind, "INDEX SAME:", same, "sample rate same:", sr, gst["index_entries"], ost["index_entries"], "build ms", round(gst["build_seconds"]*1e3, 2))
    for w in (() if os.environ.get("FAST") else (1, 2, 5, 19)):
        st2 = vi.buildIndexMinimizers(1, w, 100.0); g2 = vi.export()
        o2 = o.build_index_minimizers(1, w, 100.0); e2 = o.export_index()
        s2 = np.array_equal(g2.keys, e2.keys) and np.array_equal(g2.entries, e2.entries) and np.array_equal(g2.key_off, e2.key_off)
        print("   window", w, "same:", s2, st2["index_entries"])
        ok &= s2
    vi.build(cfg); o.build_index(cfg)
    det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
    q = np.arange(0, 2 * rs.n, dtype=np.uint32)
    gres = det.getSeqOverlapsBatch(q)
    ores = o.overlaps(O.detector_params(cfg), q)
    print("   OVERLAPS SAME:", gres.lines() == ores.lines(), len(gres.recs), "device ms", round(gres.device_seconds*1e3, 2),
          {k: round(v[0]*1e3, 2) for k, v in ctx.kernel_times().items()})
    ok &= same and sr and gres.lines() == ores.lines()
sys.exit(0 if ok else 1)
