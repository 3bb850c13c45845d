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


def _sampled_oracle_check(rs, cfg, vi, det, res, queries, sample_pos, k, ex=None):
    """records of queries[sample_pos] in `res` against the oracle on the touched part of the index"""
    from oracle import oracle as O
    if ex is None:
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


def config5_rank0_of_8(genome_len: int, n_sample: int = 120):
    """What test_config5_hifi_proxy_rank0_of_8_bounded_memory does, at any genome size (tools/config5_scale.py runs it
    at 1 Gb = a third of CHM13)."""

    import torch
    from flye_amd import config, dist, gpu, workloads
    W = 8
    t0 = time.time()
    rs, min_ovlp, preset = workloads.hifi30(genome_len=genome_len)
    cfg = config.preset(preset)
    k = int(cfg["kmer_size"])
    assert rs.total_bases > 25 * genome_len
    t_gen = time.time() - t0

    # ---- the index the one-GPU way (in 8 key-range steps), kept on the host as the yardstick
    ctx = gpu.Context(k, 0)
    ctx.set_reads(rs)
    vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
    hist = vi.begin(cfg)
    ranges = dist.balanced_bin_ranges(hist, W)
    sums = np.zeros(2, np.uint64)
    for r in range(W):
        sums = vi.build_range(*ranges[r])       # running totals of this context
    st_full = vi.finish(None)
    full = vi.export()
    ctx.close()
    del vi, ctx
    t_full = time.time() - t0 - t_gen
    assert st_full["index_entries"] > 4 * genome_len
    key_lo = [np.uint64(lo) << np.uint64(2 * k - 12) for lo, hi in ranges]
    cut = np.searchsorted(full.keys, np.array(key_lo, np.uint64)).tolist() + [len(full.keys)]
    rcut = np.searchsorted(full.repetitive, np.array(key_lo, np.uint64)).tolist() + [len(full.repetitive)]

    # ---- rank 0 of 8
    ctx = gpu.Context(k, 0)
    ctx.set_reads(rs)
    reads_bytes, _ = gpu.memory_stats(reset_peak=True)
    vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
    hist0 = vi.begin(cfg)
    assert np.array_equal(hist0, hist)
    vi.build_range(*ranges[0])
    st = vi.finish(sums)
    _, peak = gpu.memory_stats()
    e0 = int(hist[ranges[0][0]:ranges[0][1]].sum())
    total_kmers = int(np.maximum(rs.length.astype(np.int64) - k, 0).sum())
    batch = 256 << 20
    # reads + one bit per k-mer position + the larger of {batch scratch: 8 B hash + 1 B flag per position,
    # this rank's sort: 4 arrays of 8 B + 8 B of run-length scratch per entry} + the piece + 1 GB of slack
    bound = reads_bytes + total_kmers // 8 + max(9 * batch, 40 * e0) + 20 * e0 + (1 << 30)
    assert peak <= bound, (peak, bound)
    piece = vi.export()
    assert np.array_equal(piece.keys, full.keys[cut[0]:cut[1]])
    a, b = int(full.key_off[cut[0]]), int(full.key_off[cut[1]])
    assert np.array_equal(piece.key_off, full.key_off[cut[0]:cut[1] + 1] - np.uint64(a))
    assert np.array_equal(piece.entries, full.entries[a:b])
    assert np.array_equal(piece.repetitive, full.repetitive[rcut[0]:rcut[1]])
    assert st["repetitive_frequency"] == st_full["repetitive_frequency"]

    # ---- the gather, straight into the context's arrays: the own piece device to device, the other seven ranks'
    # pieces (here cut from the yardstick) where their broadcasts would land
    K, E, R = len(full.keys), len(full.entries), len(full.repetitive)
    fp, pp, psz = vi.gather_begin(K, E, R)
    dev = torch.device("cuda", 0)
    keys, off, ent, rep = (dist._view(fp[0], K, dev), dist._view(fp[1], K + 1, dev), dist._view(fp[2], E, dev),
                           dist._view(fp[3], R, dev))
    keys[:cut[1]] = dist._view(pp[0], psz[0], dev)
    off[:cut[1]] = dist._view(pp[1], psz[0] + 1, dev)[:psz[0]]
    ent[:b] = dist._view(pp[2], psz[1], dev)
    rep[:rcut[1]] = dist._view(pp[3], psz[2], dev)
    keys[cut[1]:] = torch.from_numpy(full.keys[cut[1]:].view(np.int64))
    off[cut[1]:] = torch.from_numpy(full.key_off[cut[1]:].view(np.int64))
    ent[b:] = torch.from_numpy(full.entries[b:].view(np.int64))
    rep[rcut[1]:] = torch.from_numpy(full.repetitive[rcut[1]:].view(np.int64))
    torch.cuda.synchronize()
    del keys, off, ent, rep
    sample_rate = float(np.float32(rs.total_bases) / np.float32(E))
    vi.gather_end(sample_rate)
    vi.stats = dict(st, selected_kmers=K, index_entries=E, repetitive_kmers=R, sample_rate=float(np.float32(sample_rate)))
    assert np.float32(sample_rate).tobytes() == np.float32(st_full["sample_rate"]).tobytes()
    got = vi.export()
    assert all(np.array_equal(x, y) for x, y in ((got.keys, full.keys), (got.key_off, full.key_off),
                                                 (got.entries, full.entries), (got.repetitive, full.repetitive)))
    del got, piece
    index_bytes, _ = gpu.memory_stats()
    t_rank = time.time() - t0 - t_gen - t_full

    # ---- rank 0's share of the overlap stage
    det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
    assert det.p.nucl_alignment and det.p.use_hpc
    det.p.max_divergence = 0.01
    q = dist.shard_queries(rs.n, 0, W)
    t1 = time.time()
    res = det.getSeqOverlapsBatch(q)
    t2 = time.time()
    _properties(res, config.DETECTOR_MIN_OVERLAP, rs.n)
    assert np.all(res.recs["seq_divergence"] < 0.01) and np.all(res.recs["edit_distance"] >= 0)
    bp = int(rs.length[(q // 2).astype(np.int64)].sum())
    rng = np.random.default_rng(6)
    sample = np.sort(rng.choice(len(q), size=n_sample, replace=False))
    n = _sampled_oracle_check(rs, cfg, vi, det, res, q, sample, k, ex=full)
    _, peak_all = gpu.memory_stats()
    ctx.close()
    print(f"hifi proxy {genome_len / 1e6:.0f} Mb, rank 0/8: {rs.n} reads {rs.total_bases / 1e9:.2f} Gbp (gen {t_gen:.0f} s), one-GPU build + export "
          f"{t_full:.0f} s, rank build + gather {t_rank:.0f} s; reads {reads_bytes / 1e9:.2f} GB, rank-0 entries {e0 / 1e6:.0f} M of "
          f"{E / 1e6:.0f} M, build peak {peak / 1e9:.2f} GB (bound {bound / 1e9:.2f}), reads + full index {index_bytes / 1e9:.2f} GB, "
          f"peak incl. overlap scratch {peak_all / 1e9:.2f} GB; {len(q)} queries {bp / 1e9:.2f} Gbp in {t2 - t1:.1f} s "
          f"({bp / (t2 - t1) / 1e9:.2f} Gbp/s), {len(res.recs)} overlaps, {n} sampled records identical to the oracle")


def test_config5_hifi_proxy_rank0_of_8_bounded_memory(built):
    """configs[4] "Human CHM13 HiFi 30x, 8 x MI355X" on the survey's proxy (310 Mb genome, 30x,
    9.3 Gbp of HiFi reads, asm_hifi.cfg: minimizer index w = 10, base-level divergence on homopolymer-compressed
    sequence).  The gate: the synthetic reads carry 0.3 % error EACH, so true overlaps diverge by ~0.6 %; the
    --hifi-error 0.003 gate of the real data set would reject them all, the test's gate is 0.01.

    What one rank of eight does, on one GPU: the batched selection over all reads, ITS key range sorted and
    run-length encoded, finish with the sums over all ranks -> its piece; the gather of the eight pieces straight
    into the context's own arrays; then its share of the queries (i % 8 == 0) against the full index.
    Checked: (1) the device memory the library holds never exceeds a stated bound during the rank's build;
    (2) the piece equals the same key range of the index built the one-GPU way; (3) the gathered index equals
    that index; (4) >= 100 sampled query reads against the CPU oracle, record for record."""
    config5_rank0_of_8(310_000_000)
