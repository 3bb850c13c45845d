"""BASELINE.json configs 3-5 on one MI355X (configs 1-2 = tests/test_gpu_parity.py::test_full_size_properties
and bench.py): full-size index build + overlap pass with the assemble stage's detector (minimumOverlap 1000,
main_assemble.cpp:174), size-independent properties over everything, and a record-for-record comparison
with the CPU oracle on a sample of queries.  The oracle gets the part of the device-built index the sampled
reads can touch (helpers.sub_index): a query looks up its own k-mers only."""
import time

import numpy as np
import pytest

from helpers import sub_index

pytestmark = pytest.mark.gpu


def _properties(res, det_min, n_reads):
    rr = res.recs
    assert len(rr) > 0
    assert np.all(rr["cur_end"] - rr["cur_begin"] >= det_min) and np.all(rr["ext_end"] - rr["ext_begin"] >= det_min)
    assert np.all(rr["cur_begin"] >= 0) and np.all(rr["ext_begin"] >= 0)
    assert np.all(rr["cur_end"] < rr["cur_len"]) and np.all(rr["ext_end"] < rr["ext_len"])
    assert np.all(rr["ext_id"] < 2 * n_reads)
    assert np.all(np.isfinite(rr["seq_divergence"])) and np.all(rr["seq_divergence"] >= 0)
    # ascending extId inside each query list (reference emission order); the offsets partition the records
    same_q = rr["cur_id"][1:] == rr["cur_id"][:-1]
    assert np.all(rr["ext_id"][1:][same_q] > rr["ext_id"][:-1][same_q])       # onlyMaxExt: one record per target
    assert int(res.query_off[-1]) == len(rr) and np.all(np.diff(res.query_off.astype(np.int64)) >= 0)


def _sampled_oracle_check(rs, cfg, vi, det, res, queries, sample_pos, k):
    """records of queries[sample_pos] in `res` against the oracle on the touched part of the index"""
    from oracle import oracle as O
    ex = vi.export()
    reads = (queries[sample_pos] // 2).astype(np.int64)
    sub = sub_index(ex, rs, reads, k)
    del ex
    o = O.Oracle(k)
    o.set_reads(rs)
    o.import_index(sub, vi.getSampleRate())
    p = O.detector_params(cfg, min_overlap=det.p.min_overlap, max_divergence=det.p.max_divergence)
    ores = o.overlaps(p, queries[sample_pos])
    got = np.concatenate([res.of(int(i)) for i in sample_pos])
    assert len(got) == len(ores.recs) > 0
    for f in ("cur_id", "ext_id", "cur_begin", "cur_end", "cur_len", "ext_begin", "ext_end", "ext_len", "score",
              "edit_distance"):
        assert np.array_equal(got[f], ores.recs[f]), f
    assert np.array_equal(got["seq_divergence"].view(np.uint32), ores.recs["seq_divergence"].view(np.uint32))
    return len(got)


def test_config3_dmel_ont30_full_size(built):
    """configs[2]: D. melanogaster ONT 30x proxy, 136 Mb genome, 3.9 Gbp of reads, one GPU."""
    from flye_amd import config, gpu, workloads
    t0 = time.time()
    rs, min_ovlp, preset = workloads.dmel_ont30()
    cfg = config.preset(preset)
    k = int(cfg["kmer_size"])
    ctx = gpu.Context(k, 0)
    ctx.set_reads(rs)
    vi = gpu.VertexIndex(ctx, 1.0)
    st = vi.build(cfg)
    det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
    assert det.p.min_overlap == config.DETECTOR_MIN_OVERLAP
    q = np.arange(0, 2 * rs.n, 2, dtype=np.uint32)
    t1 = time.time()
    res = det.getSeqOverlapsBatch(q)
    t2 = time.time()
    assert rs.total_bases > 3.5e9 and st["index_entries"] > 1.5e9
    _properties(res, config.DETECTOR_MIN_OVERLAP, rs.n)
    rng = np.random.default_rng(2)
    sample = np.sort(rng.choice(len(q), size=120, replace=False))
    n = _sampled_oracle_check(rs, cfg, vi, det, res, q, sample, k)
    ctx.close()
    print(f"dmel_ont30: {rs.n} reads {rs.total_bases / 1e9:.2f} Gbp, setup {t1 - t0:.0f} s, pass {t2 - t1:.1f} s "
          f"({rs.total_bases / (t2 - t1) / 1e9:.2f} Gbp/s incl. first-call allocations), {len(res.recs)} overlaps, "
          f"{n} sampled records identical to the oracle")


def test_config4_synthetic_10gb_rank0_of_8(built):
    """configs[3]: 10 Gbp of ONT-profile reads sharded over 8 GPUs = on each rank the whole index (replicated,
    DESIGN.md §6) and the queries i % 8 == rank.  This is rank 0's share, on one GPU."""
    from flye_amd import config, dist, gpu, workloads
    t0 = time.time()
    rs, min_ovlp, preset = workloads.synth10g_ont()
    cfg = config.preset(preset)
    k = int(cfg["kmer_size"])
    ctx = gpu.Context(k, 0)
    ctx.set_reads(rs)
    vi = gpu.VertexIndex(ctx, 1.0)
    st = vi.build(cfg)
    det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
    q = dist.shard_queries(rs.n, 0, 8)
    t1 = time.time()
    res = det.getSeqOverlapsBatch(q)
    t2 = time.time()
    assert rs.total_bases > 9e9 and st["index_entries"] > 3e9
    _properties(res, config.DETECTOR_MIN_OVERLAP, rs.n)
    bp = int(rs.length[(q // 2).astype(np.int64)].sum())
    rng = np.random.default_rng(4)
    sample = np.sort(rng.choice(len(q), size=100, replace=False))
    n = _sampled_oracle_check(rs, cfg, vi, det, res, q, sample, k)
    ctx.close()
    print(f"synth10g rank 0/8: {rs.n} reads {rs.total_bases / 1e9:.2f} Gbp indexed ({st['index_entries'] / 1e9:.2f} G entries), "
          f"setup {t1 - t0:.0f} s, {len(q)} queries {bp / 1e9:.2f} Gbp in {t2 - t1:.1f} s, {len(res.recs)} overlaps, "
          f"{n} sampled records identical to the oracle")


def test_config5_hifi_parameters_at_scale(built):
    """configs[4] parameters (asm_hifi.cfg: minimizer index w = 10, base-level divergence on homopolymer-
    compressed sequence, --hifi-error gate) on the largest HiFi read set that keeps the suite short:
    100 Mb genome, 30x, 3 Gbp.  (CHM13 itself is 93 Gbp: the residency table is in DESIGN.md §6.)"""
    from flye_amd import config, gpu, workloads
    t0 = time.time()
    rs, min_ovlp, preset = workloads.hifi30(genome_len=100_000_000)
    cfg = config.preset(preset)
    k = int(cfg["kmer_size"])
    ctx = gpu.Context(k, 0)
    ctx.set_reads(rs)
    vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
    st = vi.build(cfg)
    det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
    assert det.p.nucl_alignment and det.p.use_hpc
    # the synthetic reads carry 0.3 % error each (0.6 % pairwise): the gate sits where true overlaps pass
    det.p.max_divergence = 0.01
    q = np.arange(0, 2 * rs.n, 2, dtype=np.uint32)
    t1 = time.time()
    res = det.getSeqOverlapsBatch(q)
    t2 = time.time()
    assert rs.total_bases > 2.5e9
    _properties(res, config.DETECTOR_MIN_OVERLAP, rs.n)
    assert np.all(res.recs["seq_divergence"] < 0.01) and np.all(res.recs["edit_distance"] >= 0)
    rng = np.random.default_rng(6)
    sample = np.sort(rng.choice(len(q), size=100, replace=False))
    n = _sampled_oracle_check(rs, cfg, vi, det, res, q, sample, k)
    ctx.close()
    print(f"hifi 100 Mb: {rs.n} reads {rs.total_bases / 1e9:.2f} Gbp, setup {t1 - t0:.0f} s, pass {t2 - t1:.1f} s, "
          f"{len(res.recs)} overlaps, {n} sampled records identical to the oracle")
