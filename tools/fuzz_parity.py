"""Randomised parity sweep: HIP path vs CPU oracle over many small configurations
(read model, preset, k, window, detector flags, query mix).  Exit status 1 on any mismatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flye_amd import config, gpu, synth
from oracle import oracle as O



def run(seed=0, n_cases=30, verbose=True):
  rng = np.random.default_rng(seed)
  bad = 0
  t0 = time.time()
  for case in range(n_cases):
      kind = rng.choice(["pb_raw", "ont_raw", "hifi", "hifi03"])
      preset = rng.choice(["raw", "corrected", "hifi", "subasm"]) if kind.startswith("hifi") else rng.choice(["raw", "raw", "corrected"])
      cfg = config.preset(preset)
      k = int(rng.choice([15, 17, 17, 17, 21, 31])) if cfg["use_minimizers"] else int(rng.choice([13, 15, 17, 17]))
      glen = int(rng.integers(15_000, 60_000))
      if kind.startswith("hifi"):
          glen = max(glen, 45_000)   # 15 kb reads on a genome of their own size: every read overlaps every read over its
                                     # whole length, densely -- minutes of DP for the oracle and the GPU alike, no new coverage
      cov = int(rng.integers(8, 30))
      rs = synth.simulate(seed=int(rng.integers(1, 1 << 30)), genome_len=glen, coverage=cov, kind=kind,
                          n_homopolymers=int(rng.integers(0, 60)), n_tandems=int(rng.integers(0, 60)),
                          n_repeat_families=int(rng.integers(0, 12)), circular=int(rng.integers(0, 2))).filter_min_len(int(rng.choice([0, 500, 1000, 3000])))
      if rs.n < 3:
          continue
      first = int(rng.choice([0, 0, 2 * int(rng.integers(1, 1000))]))
      ctx = gpu.Context(k, 0); ctx.set_reads(rs, first)
      o = O.Oracle(k); o.set_reads(rs, first)
      if cfg["use_minimizers"]:
          w = int(rng.choice([1, 3, 5, 10, 16]))
          vi = gpu.VertexIndex(ctx, 2.0)
          gst = vi.buildIndexMinimizers(1, w, cfg["repeat_kmer_rate"]); ost = o.build_index_minimizers(1, w, cfg["repeat_kmer_rate"])
      else:
          sel = float(np.float32(rng.choice([0.25, 0.40, 0.75]))); tf = int(rng.choice([0, 20, 100])); mf = int(rng.choice([2, 2, 3]))
          rr = float(rng.choice([100.0, 20.0]))
          vi = gpu.VertexIndex(ctx, 1.0); vi.countKmers()
          gst = vi.buildIndexUnevenCoverage(mf, sel, tf, rr); ost = o.build_index_solid(mf, sel, tf, rr, 1.0)
      ge, oe = vi.export(), o.export_index()
      same_idx = (np.array_equal(ge.keys, oe.keys) and np.array_equal(ge.key_off, oe.key_off) and
                  np.array_equal(ge.entries, oe.entries) and np.array_equal(ge.repetitive, oe.repetitive) and
                  np.float32(gst["sample_rate"]).tobytes() == np.float32(ost["sample_rate"]).tobytes())
      dk = dict(min_overlap=int(rng.choice([100, 500, 1000, 2000])), only_max_ext=bool(rng.integers(0, 2)),
                max_overhang=int(rng.choice([0, 100, 500, 1500])), nucl_alignment=bool(cfg["reads_base_alignment"]) and bool(rng.integers(0, 2)))
      cfg2 = dict(cfg); cfg2["maximum_jump"] = float(rng.choice([300, 1500, 1500, 5000])); cfg2["hpc_scoring_on"] = float(rng.integers(0, 2))
      maxdiv = float(np.float32(rng.choice([1.0, 0.3, 0.05])))
      mo = int(rng.choice([0, 0, 3, 25])); fl = bool(rng.integers(0, 2))
      keep = bool(rng.integers(0, 2)); part = bool(rng.integers(0, 3) == 0)
      if part: mo = 0      # the marks exist only with maxOverlaps = 0
      qsel = rng.choice(["fwd", "rc", "all", "some"])
      allq = first + np.arange(0, 2 * rs.n)
      q = {"fwd": allq[::2], "rc": allq[1::2], "all": allq, "some": rng.choice(allq, size=max(1, rs.n // 2), replace=True)}[qsel].astype(np.uint32)
      det = gpu.OverlapDetector(ctx, vi, int(cfg2["maximum_jump"]), dk["min_overlap"], dk["max_overhang"], keep, dk["only_max_ext"],
                                maxdiv, dk["nucl_alignment"], part, bool(cfg2["hpc_scoring_on"]))
      op = O.detector_params(cfg2, max_divergence=maxdiv, keep_alignment=keep, partition_bad_mappings=part, **dk)
      # cost bound, by measurement and whatever the flags are: a degenerate index (a few thousand k-mers of
      # frequency > 100 on HiFi-like reads: hundreds of seed hits per base, 10^4..10^5-hit groups whose DP
      # scans hundreds of candidates per element) costs the CPU oracle -- and the reference -- minutes per
      # read.  The oracle is timed on two queries; the query list is cut to what fits the budget.
      tp = time.time(); o.overlaps(op, q[:2], max_overlaps=mo, force_local=fl); tp = (time.time() - tp) / min(2, len(q))
      budget = float(os.environ.get("FUZZ_CASE_SECONDS", "40"))
      cut = ""
      if tp * len(q) > budget:
          keepq = max(2, int(budget / tp))
          cut = f" (queries cut {len(q)} -> {keepq}: {tp:.1f} s per query on the CPU)"
          q = q[:keepq]
      # sort-record form: 32-bit keys where they fit (always, at these sizes), or forced to the packed
      # 64-bit records / the plain 64-bit keys + values that large inputs use
      kmode = rng.choice(["auto", "packed", "key64"])
      os.environ.pop("FG_FORCE_KEY64", None); os.environ.pop("FG_PACKED_KEYS", None)
      if kmode != "auto":
          os.environ["FG_FORCE_KEY64"] = "1"
          os.environ["FG_PACKED_KEYS"] = "1" if kmode == "packed" else "0"
      tg = time.time()
      gres = det.getSeqOverlapsBatch(q, forceLocal=fl, maxOverlaps=mo)
      tg = time.time() - tg
      os.environ.pop("FG_FORCE_KEY64", None); os.environ.pop("FG_PACKED_KEYS", None)
      to = time.time()
      ores = o.overlaps(op, q, max_overlaps=mo, force_local=fl)
      to = time.time() - to
      same = (gres.lines() == ores.lines() and np.array_equal(gres.query_off, ores.query_off) and
              np.array_equal(gres.stats.view(np.uint32), ores.stats.view(np.uint32)) and
              np.array_equal(gres.recs["edit_distance"], ores.recs["edit_distance"]) and
              (not keep or (np.array_equal(gres.match_off, ores.match_off) and np.array_equal(gres.matches, ores.matches))) and
              (not part or np.array_equal(gres.needs_trim, ores.needs_trim)))
      tag = "ok " if (same and same_idx) else "BAD"
      if not (same and same_idx): bad += 1
      if verbose: print(f"{tag} case {case}: {kind}/{preset} k={k} reads={rs.n} {dk} jump={int(cfg2['maximum_jump'])} maxdiv={maxdiv} mo={mo} fl={fl} keep={keep} part={part} keys={kmode} q={qsel} "
            f"recs={len(gres.recs)} index_same={same_idx} gpu={tg:.2f}s oracle={to:.2f}s{cut}", flush=True)
  print(f"{n_cases} cases, {bad} mismatching, {time.time()-t0:.0f} s")
  return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30) else 0)
