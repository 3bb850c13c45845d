"""Parity tests proper: the HIP path, called through the C ABI, against the golden
vectors of the compiled reference and against the CPU oracle on the same seeded
inputs.  Bit-exact (integers and the float's bit pattern)."""
import ctypes as C

import numpy as np
import pytest

from helpers import (bits_to_float, case_queries, check_index_stats, golden_lines, golden_reads,
                     index_digest)

pytestmark = pytest.mark.gpu

GOLDEN_CASES = ["raw_pb", "raw_ont_rc", "raw_div", "raw_local", "hifi", "corrected_local", "hifi_rc_max", "raw_pb_aln"]


def _gpu_setup(rs, cfg):
    from flye_amd import gpu
    ctx = gpu.Context(int(cfg["kmer_size"]), 0)
    ctx.set_reads(rs)
    vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
    st = vi.build(cfg)
    det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
    return ctx, vi, st, det


def _oracle_setup(rs, cfg):
    from oracle import oracle as O
    o = O.Oracle(int(cfg["kmer_size"]))
    o.set_reads(rs)
    st = o.build_index(cfg)
    return o, st


def _same_index(a, b):
    return (np.array_equal(a.keys, b.keys) and np.array_equal(a.key_off, b.key_off)
            and np.array_equal(a.entries, b.entries) and np.array_equal(a.repetitive, b.repetitive))


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_golden_reference_vectors(built, golden_cases, name):
    from flye_amd import config
    case = golden_cases[name]
    rs = golden_reads(case)
    cfg = config.preset(case["preset"])
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    check_index_stats(st, case["index"])
    assert index_digest(vi.export()) == case["index"]["sha256"]
    det.p.max_divergence = bits_to_float(case["max_div_bits"])
    det.p.keep_alignment = int(case.get("keep_aln", False))
    res = det.getSeqOverlapsBatch(case_queries(case, rs.n), forceLocal=case.get("force_local", False),
                                  maxOverlaps=case.get("max_overlaps", 0))
    assert res.lines() == golden_lines(name)
    assert len(res.recs) == case["n_overlaps"]


@pytest.mark.parametrize("seed,kind,opts", [
    (1, "pb_raw", dict(mixed=True)),
    (2, "ont_raw", dict(max_overlaps=7, hp=50, tr=80)),
    (3, "pb_raw", dict(force_local=True, max_div=0.2)),
    (4, "pb_raw", dict(first_id=1000, rep=12)),
    (5, "hifi", dict(preset="hifi", mixed=True, max_div=0.01, cov=20)),
    (6, "hifi03", dict(preset="corrected", max_overlaps=9, hp=60, tr=60, cov=20)),
    (7, "pb_raw", dict(preset="hifi", cov=20)),     # divergent pairs through the edit-distance kernel
    (8, "pb_raw", dict(all_primaries=True, max_overlaps=5, mixed=True)),   # limit is tested per target group
    (9, "hifi", dict(preset="corrected", all_primaries=True, max_overlaps=3, cov=15, rep=10)),
    # keepAlignment: the kmerMatches lists themselves, element by element
    (10, "pb_raw", dict(keep_aln=True, mixed=True, tr=40)),
    (11, "hifi", dict(preset="hifi", keep_aln=True, all_primaries=True, max_overlaps=6, cov=15, rep=10, tr=40)),
    (12, "ont_raw", dict(keep_aln=True, all_primaries=True, max_div=0.25, hp=40)),
])
def test_against_oracle_variants(built, seed, kind, opts):
    from flye_amd import config, gpu, synth
    from oracle import oracle as O
    rs = synth.simulate(seed=seed, genome_len=50_000, coverage=opts.get("cov", 30), kind=kind,
                        n_homopolymers=opts.get("hp", 8), n_tandems=opts.get("tr", 8),
                        n_repeat_families=opts.get("rep", 4)).filter_min_len(1000)
    cfg = config.preset(opts.get("preset", "raw"))
    first = opts.get("first_id", 0)
    ctx = gpu.Context(17, 0)
    ctx.set_reads(rs, first)
    vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
    gst = vi.build(cfg)
    det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
    det.p.max_divergence = opts.get("max_div", 1.0)
    det.p.only_max_ext = 0 if opts.get("all_primaries") else 1
    det.p.keep_alignment = int(opts.get("keep_aln", False))
    o = O.Oracle(17)
    o.set_reads(rs, first)
    ost = o.build_index(cfg)
    assert _same_index(vi.export(), o.export_index())
    for f in ("total_kmers", "selected_kmers", "index_entries", "repetitive_kmers", "repetitive_frequency"):
        assert gst[f] == ost[f]
    q = first + (np.arange(0, 2 * rs.n) if opts.get("mixed") else np.arange(0, 2 * rs.n, 2))
    q = q.astype(np.uint32)
    gres = det.getSeqOverlapsBatch(q, forceLocal=opts.get("force_local", False),
                                   maxOverlaps=opts.get("max_overlaps", 0))
    ores = o.overlaps(O.detector_params(cfg, max_divergence=opts.get("max_div", 1.0),
                                        only_max_ext=not opts.get("all_primaries"),
                                        keep_alignment=opts.get("keep_aln", False)), q,
                      max_overlaps=opts.get("max_overlaps", 0), force_local=opts.get("force_local", False))
    assert gres.lines() == ores.lines()
    if opts.get("keep_aln"):
        assert np.array_equal(gres.match_off, ores.match_off)
        assert np.array_equal(gres.matches, ores.matches)
        assert len(gres.matches) > 2 * len(gres.recs) > 0
        r0 = gres.recs[0]
        m0 = gres.kmerMatches(0)
        assert tuple(m0[0]) == (r0["cur_begin"], r0["ext_begin"]) and tuple(m0[-1]) == (r0["cur_end"], r0["ext_end"])
    else:
        assert gres.match_off is None
    assert np.array_equal(gres.query_off, ores.query_off)
    assert np.array_equal(gres.stats.view(np.uint32), ores.stats.view(np.uint32))
    for f in ("chain_length", "filtered_positions", "edit_distance", "hpc_len_cur", "hpc_len_ext"):
        assert np.array_equal(gres.recs[f], ores.recs[f])
    assert (gres.query_kmers, gres.seed_hits) == (ores.query_kmers, ores.seed_hits)
    if not opts.get("max_overlaps"):   # with a limit the reference stops visiting groups early
        assert (gres.dp_groups, gres.dp_elements) == (ores.dp_groups, ores.dp_elements)
        assert 0 <= gres.dp_elements_small <= gres.dp_elements      # the one-kernel chaining class's share (bench's roofline)
    # batch invariance: any sub-batch gives the same per-read lists
    sub = q[5:40:3]
    part = det.getSeqOverlapsBatch(sub, forceLocal=opts.get("force_local", False),
                                   maxOverlaps=opts.get("max_overlaps", 0))
    pos = {int(x): i for i, x in enumerate(q)}
    for j, rid in enumerate(sub):
        a, b = part.of(j), gres.of(pos[int(rid)])
        assert a.tobytes() == b.tobytes()
        if opts.get("keep_aln"):
            ia, ib = int(part.query_off[j]), int(gres.query_off[pos[int(rid)]])
            for t in range(len(a)):
                assert np.array_equal(part.kmerMatches(ia + t), gres.kmerMatches(ib + t))


@pytest.mark.parametrize("name", ["edges_raw", "edges_hifi", "edges_raw_max", "edges_raw_aln", "edges_hifi_aln"])
def test_read_aligner_style_golden(built, golden_cases, name):
    """fg_set_queries + only_max_ext = 0: reads from a second container against an index of
    "edge" sequences, every primary overlap (ReadAligner::alignReads flags,
    read_aligner.cpp:178-217), vs the reference's golden output and the oracle."""
    from flye_amd import config, gpu
    from oracle import oracle as O
    from helpers import edges_setup, golden_queries
    case = golden_cases[name]
    edges = golden_reads(case)
    reads = golden_queries(case)
    cfg = config.preset(case["preset"])
    wnd, dk = edges_setup(case, cfg)
    ctx = gpu.Context(int(cfg["kmer_size"]), 0)
    ctx.set_reads(edges, 0)
    vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
    st = vi.buildIndexMinimizers(1, wnd, cfg["repeat_kmer_rate"])
    check_index_stats(st, case["index"])
    assert index_digest(vi.export()) == case["index"]["sha256"]
    ctx.set_queries(reads, 2 * edges.n)
    det = gpu.OverlapDetector(ctx, vi, int(cfg["maximum_jump"]), dk["min_overlap"], dk["max_overhang"],
                              dk.get("keep_alignment", False), dk["only_max_ext"], 1.0, dk["nucl_alignment"], False, bool(cfg["hpc_scoring_on"]))
    q = (2 * edges.n + np.arange(0, 2 * reads.n)).astype(np.uint32)      # both strands
    mo = case.get("max_overlaps", 0)
    res = det.getSeqOverlapsBatch(q, maxOverlaps=mo)
    fwd = res.query_ids % 2 == 0
    lines = res.lines()
    got = [l for i in np.nonzero(fwd)[0] for l in lines[int(res.query_off[i]):int(res.query_off[i + 1])]]
    assert got == golden_lines(name)
    o = O.Oracle(int(cfg["kmer_size"]))
    o.set_reads(edges, 0)
    o.build_index_minimizers(1, wnd, cfg["repeat_kmer_rate"])
    o.set_queries(reads, 2 * edges.n)
    ores = o.overlaps(O.detector_params(cfg, **dk), q, max_overlaps=mo)
    assert lines == ores.lines()
    assert np.array_equal(res.stats.view(np.uint32), ores.stats.view(np.uint32))
    # ids of the query container are rejected when they collide with the indexed ones
    with pytest.raises(gpu.FlyeGpuError):
        ctx.set_queries(reads, 2 * edges.n - 2)
    # back to "queries are the indexed reads"
    ctx.set_reads(edges, 0)
    vi.buildIndexMinimizers(1, wnd, cfg["repeat_kmer_rate"])
    self_res = det.getSeqOverlapsBatch(np.arange(0, 2 * edges.n, 2, dtype=np.uint32))
    o2 = O.Oracle(int(cfg["kmer_size"]))
    o2.set_reads(edges, 0)
    o2.build_index_minimizers(1, wnd, cfg["repeat_kmer_rate"])
    assert self_res.lines() == o2.overlaps(O.detector_params(cfg, **dk), np.arange(0, 2 * edges.n, 2)).lines()


@pytest.mark.parametrize("name", ["repeat_raw", "repeat_hifi"])
def test_repeat_stage_golden(built, golden_cases, name):
    """The RepeatGraph::build flag set (repeat_graph.cpp:72-97): keep_alignment, every primary,
    base-level divergence, partition_bad_mappings (gated-out primaries come back marked for the
    caller's checkIdyAndTrim) -- vs the reference's vectors and the oracle."""
    from flye_amd import config, gpu
    from oracle import oracle as O
    from helpers import check_repeat_stage_result, repeat_stage_setup
    case = golden_cases[name]
    seqs = golden_reads(case)
    cfg = config.preset(case["preset"])
    wnd, dk = repeat_stage_setup(case, cfg)
    ctx = gpu.Context(int(cfg["kmer_size"]), 0)
    ctx.set_reads(seqs, 0)
    vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
    st = vi.buildIndexMinimizers(1, wnd, cfg["repeat_kmer_rate"])
    check_index_stats(st, case["index"])
    det = gpu.OverlapDetector(ctx, vi, int(cfg["maximum_jump"]), dk["min_overlap"], dk["max_overhang"], True,
                              dk["only_max_ext"], dk["max_divergence"], dk["nucl_alignment"], True,
                              bool(cfg["hpc_scoring_on"]))
    q = np.arange(0, 2 * seqs.n, dtype=np.uint32)      # both strands
    res = det.getSeqOverlapsBatch(q)
    o = O.Oracle(int(cfg["kmer_size"]))
    o.set_reads(seqs, 0)
    o.build_index_minimizers(1, wnd, cfg["repeat_kmer_rate"])
    ores = o.overlaps(O.detector_params(cfg, **dk), q)
    assert res.lines() == ores.lines()
    assert np.array_equal(res.needs_trim, ores.needs_trim)
    assert np.array_equal(res.match_off, ores.match_off) and np.array_equal(res.matches, ores.matches)
    assert np.array_equal(res.stats.view(np.uint32), ores.stats.view(np.uint32))
    fwd = det.getSeqOverlapsBatch(q[::2])
    check_repeat_stage_result(fwd, case, golden_cases)
    # the marks exist only with max_overlaps = 0 (the only way the reference uses the flag)
    with pytest.raises(gpu.FlyeGpuError) as e:
        det.getSeqOverlapsBatch(q[:2], maxOverlaps=3)
    assert e.value.code == -7


def _median3_killer(n):
    k = n // 2
    a = np.zeros(n, np.uint64)
    for i in range(1, k + 1):
        if i % 2 == 1:
            a[i - 1] = i
            a[i] = k + i
        a[k + i - 1] = 2 * i
    return a


@pytest.mark.parametrize("stream_max", [None, 600, 10 ** 9, "many"])
def test_device_sort_equals_std_sort(built, monkeypatch, stream_max):
    """k_sort_hits on its own: tie-heavy, patterned and adversarial inputs must come
    out in exactly the permutation std::sort produces (heapsort fallback included).  Every form of the level
    loop: the streamed and the closed-form partition at any piece size, and the many-pieces rule of large chunks
    forced on at this size."""
    from flye_amd import gpu
    from oracle import oracle as O
    if stream_max == "many":
        monkeypatch.setenv("FG_SORT_MANY_MIN", "2")
        monkeypatch.setenv("FG_SORT_STREAM_MANY", "50000")
    elif stream_max is not None:      # which partition form the level kernel uses above the LDS piece size
        monkeypatch.setenv("FG_SORT_STREAM_MAX", str(stream_max))
    rng = np.random.default_rng(7)
    segs = []
    for n in list(range(0, 70)) + [100, 128, 129, 191, 192, 193, 255, 257, 1000, 4097, 20000, 70000]:
        for hi in (2, 5, 50, 1 << 40):
            segs.append(rng.integers(0, hi, size=n, dtype=np.uint64))
    for n in (17, 64, 65, 1000, 5000):
        segs += [np.arange(n, dtype=np.uint64), np.arange(n, dtype=np.uint64)[::-1].copy(),
                 np.zeros(n, np.uint64), (np.arange(n) % 3).astype(np.uint64),
                 (np.arange(n) // 7).astype(np.uint64)]
    for n in (64, 130, 1000, 20000, 100000):
        segs += [_median3_killer(n), _median3_killer(n) // 3]
    off = np.zeros(len(segs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in segs])
    keys = np.concatenate(segs)
    ctx = gpu.Context(17, 0)
    sk, perm = ctx.debug_sort_pairs(keys, off)
    for i, s in enumerate(segs):
        a, b = int(off[i]), int(off[i + 1])
        want = O.std_sort_perm(s)
        assert np.array_equal(perm[a:b], want), f"segment {i} (n={len(s)}) permutation differs from std::sort"
        assert np.array_equal(sk[a:b], s[want])


def test_container_mirror_and_divergence_threshold(built, golden_cases):
    """OverlapContainer mirror: estimateOverlaperParameters draws the reference's rand()
    sequence and lands on the reference's threshold bit for bit; lazySeqOverlaps of a
    reverse-complement id is complement() of the forward list (overlap.cpp:528-574)."""
    from flye_amd import config, gpu
    case = golden_cases["raw_div"]
    rs = golden_reads(case)
    cfg = config.preset("raw")
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    oc = gpu.OverlapContainer(det)
    libc = C.CDLL(None)
    libc.srand(1)
    libc.rand.restype = C.c_int
    mean = oc.estimateOverlaperParameters(libc.rand)
    assert np.float32(mean).view(np.uint32) == int(case["mean_div_bits"], 16)
    thr = oc.setDivergenceThreshold(cfg["assemble_ovlp_divergence"], bool(cfg["assemble_divergence_relative"]))
    assert np.float32(thr).view(np.uint32) == int(case["max_div_bits"], 16)
    oc.prefetch([0, 2, 4, 7])
    f = oc.lazySeqOverlaps(6)
    r = oc.lazySeqOverlaps(7)
    assert len(f) == len(r) and r.tobytes() == gpu.complement(f).tobytes()
    one = oc.quickSeqOverlaps(6)
    assert one.tobytes() == f.tobytes()


def test_batching_container_under_threads(built):
    """include/flye_gpu_bridge.h: 16 threads ask for one read at a time (lazy, quick, both
    strands, repeats), the dispatcher turns that into a few device batches; every list equals
    the direct fg_overlaps result and lazy reverse ids get the complemented forward list."""
    import threading
    from flye_amd import config, gpu, synth
    rs = synth.simulate(seed=31, genome_len=60_000, coverage=25, kind="pb_raw", n_tandems=30).filter_min_len(1000)
    cfg = config.preset("raw")
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    det.p.max_divergence = 0.3
    allq = np.arange(0, 2 * rs.n, dtype=np.uint32)
    direct = det.getSeqOverlapsBatch(allq)                      # lazy class: maxOverlaps 0, not local
    direct_q = det.getSeqOverlapsBatch(allq, forceLocal=True, maxOverlaps=5)
    direct_stats = np.sort(det.getSeqOverlapsBatch(allq[::2]).stats)
    oc = gpu.BatchingOverlapContainer(det, max_batch=64, linger_us=300)
    rng = np.random.default_rng(3)
    work = [rng.permutation(np.concatenate([allq, allq[: rs.n]])) for _ in range(16)]
    errors = []

    bad_id = 2 * rs.n + 6

    def worker(t):
        try:
            for j, rid in enumerate(work[t]):
                rid = int(rid)
                if t == 3 and j % 50 == 0:
                    # an id outside the container is this caller's error alone: the other threads' requests that
                    # share its batch are answered normally and nothing is cached for it
                    for call in (lambda: oc.lazySeqOverlaps(bad_id), lambda: oc.quickSeqOverlaps(bad_id, 5, True)):
                        try:
                            call()
                            raise AssertionError("bad id accepted")
                        except gpu.FlyeGpuError as e:
                            assert e.code == -3
                got = oc.lazySeqOverlaps(rid)
                want = direct.of(rid & ~1)
                if rid & 1:
                    want = gpu.complement(want)
                assert got.tobytes() == want.tobytes(), ("lazy", rid)
                if (j + t) % 7 == 0:
                    q = oc.quickSeqOverlaps(rid, 5, True)
                    assert q.tobytes() == direct_q.of(rid).tobytes(), ("quick", rid)
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    oc.prefetch(allq[:40])
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    stc = oc.stats()
    n_lazy = sum(len(w) for w in work)
    assert stc["requests"] >= n_lazy
    assert stc["cached_overlaps"] == sum(len(direct.of(i)) for i in range(0, 2 * rs.n, 2))
    # every forward read went to the device once for the lazy class, in far fewer calls than reads
    quick_reads = stc["reads_computed"] - rs.n
    assert quick_reads >= 0 and stc["device_calls"] < (rs.n + quick_reads) // 2
    assert stc["cache_hits"] > n_lazy // 2
    # OvlpDivStats: the lazy class contributes exactly the values of one pass over the forward reads
    lazy_stats = oc.divergenceStats()
    assert len(lazy_stats) >= len(direct_stats)
    oc.setDivergenceThreshold(0.05)
    tight = oc.quickSeqOverlaps(0)
    det.p.max_divergence = 0.05
    oc.close()
    assert tight.tobytes() == det.getSeqOverlapsBatch(np.array([0], np.uint32)).recs.tobytes()


def test_bridge_read_ahead(built):
    """The quick path computes the records behind the highest id asked for in the same device call: worker threads that
    walk the container in id order, one getSeqOverlaps call each in flight (processInParallel's pattern), are
    answered from those results in a few device calls; lists stay the direct ones, for a walk in id order, for
    random jumps, and across a change of the divergence threshold (results computed ahead under the old one are
    dropped)."""
    import itertools
    import threading
    from flye_amd import config, gpu, synth
    rs = synth.simulate(seed=37, genome_len=80_000, coverage=30, kind="pb_raw", n_tandems=20).filter_min_len(1000)
    cfg = config.preset("raw")
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    det.p.max_divergence = 0.3
    allq = np.arange(0, 2 * rs.n, dtype=np.uint32)
    direct = det.getSeqOverlapsBatch(allq)
    oc = gpu.BatchingOverlapContainer(det, max_batch=64, linger_us=100)
    errors = []

    def walk(ids, want):
        counter = itertools.count()
        lock = threading.Lock()

        def worker():
            try:
                while True:
                    with lock:
                        j = next(counter)
                    if j >= len(ids):
                        return
                    rid = int(ids[j])
                    assert oc.quickSeqOverlaps(rid, 0, False).tobytes() == want.of(rid).tobytes(), rid
            except Exception as e:      # noqa: BLE001
                errors.append(repr(e))
        th = [threading.Thread(target=worker) for _ in range(8)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    fwd = allq[::2]
    walk(fwd, direct)
    assert not errors, errors[:3]
    s1 = oc.stats()
    assert s1["ahead_hits"] > 0.7 * len(fwd), s1          # most calls never waited for the device
    assert s1["device_calls"] < len(fwd) // 16, s1        # 8 callers in flight, yet far fewer calls than reads / 8
    # the reverse strand, then random jumps over both strands
    walk(allq[1::2], direct)
    walk(np.random.default_rng(1).permutation(allq), direct)
    assert not errors, errors[:3]
    # a new threshold: nothing computed ahead under the old one may be handed out
    walk(fwd[: len(fwd) // 2], direct)
    oc.setDivergenceThreshold(0.05)
    det.p.max_divergence = 0.05
    tight = det.getSeqOverlapsBatch(allq)
    walk(fwd[len(fwd) // 2:], tight)
    assert not errors, errors[:3]
    assert sum(len(tight.of(int(i))) for i in fwd) < sum(len(direct.of(int(i))) for i in fwd)   # the gate did change lists
    oc.close()


def test_bridge_neighbour_read_ahead(built):
    """Extender's way of asking (extender.cpp:44-98): the overlaps of the current read, then the reads on the other
    side of them, longest overlap first, a few of them, then on to one of those.  The scheduler computes the records
    named by the overlaps it hands out ahead of their requests; lists stay the direct ones and most requests of the
    walk are answered without a device call of their own."""
    import threading
    from flye_amd import config, gpu, synth
    rs = synth.simulate(seed=41, genome_len=600_000, coverage=30, kind="pb_raw").filter_min_len(1000)
    cfg = config.preset("raw")
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    det.p.max_divergence = 0.3
    allq = np.arange(0, 2 * rs.n, dtype=np.uint32)
    direct = det.getSeqOverlapsBatch(allq)
    oc = gpu.BatchingOverlapContainer(det, max_batch=64, linger_us=100)
    errors = []
    asked = [0]
    cache, cache_lock = {}, threading.Lock()    # OverlapContainer's own cache sits above the seam, shared by its threads:
                                                # a read reaches getSeqOverlaps once

    def extend(start, steps):
        try:
            def lazy(fid):
                with cache_lock:
                    have = cache.get(fid)
                    if have is None:
                        asked[0] += 1
                if have is None:
                    have = oc.quickSeqOverlaps(fid, 0, False)
                    assert have.tobytes() == direct.of(fid).tobytes(), fid
                    with cache_lock:
                        cache[fid] = have
                return have
            cur = start
            visited = {cur & ~1}
            for _ in range(steps):
                ov = lazy(cur & ~1)
                cand = ov[np.argsort(-(ov["cur_end"] - ov["cur_begin"]), kind="stable")]
                fresh = [int(o["ext_id"]) for o in cand if (int(o["ext_id"]) & ~1) not in visited]
                nxt = None
                for e in fresh[:4]:             # the first few candidates are looked at, the first usable one is taken
                    if len(lazy(e & ~1)) and nxt is None:
                        nxt = e
                if nxt is None:
                    return
                visited.add(nxt & ~1)
                cur = nxt
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=extend, args=(int(s), 60)) for s in np.random.default_rng(2).choice(allq[::2], 6, replace=False)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    s1 = oc.stats()
    oc.close()
    assert asked[0] > 200
    print("neighbour read-ahead:", asked[0], "reads asked for,", s1)
    assert s1["ahead_hits"] > 0.5 * asked[0], (s1, asked[0])        # most reads of the walk were there already
    assert s1["device_calls"] < 0.6 * asked[0], (s1, asked[0])


def test_internal_chunking_is_invisible(built, monkeypatch):
    """fg_overlaps cuts big batches into chunks bounded by k-mers / seed hits (and halves a
    chunk whose hits exceed the budget); results must not depend on the cut."""
    from flye_amd import config, gpu, synth
    rs = synth.simulate(seed=21, genome_len=60_000, coverage=25, kind="pb_raw").filter_min_len(1000)
    cfg = config.preset("raw")
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    det.p.keep_alignment = 1
    q = np.arange(0, 2 * rs.n, dtype=np.uint32)
    whole = det.getSeqOverlapsBatch(q, maxOverlaps=11)
    base = (whole.recs.tobytes(), whole.query_off.tobytes(), whole.stats.tobytes(), whole.seed_hits,
            whole.dp_groups, whole.dp_elements, whole.match_off.tobytes(), whole.matches.tobytes())
    for kb, hb in ((200_000, 1 << 40), (1 << 40, 50_000), (90_000, 30_000)):
        monkeypatch.setenv("FG_KMER_BUDGET", str(kb))
        monkeypatch.setenv("FG_HIT_BUDGET", str(hb))
        part = det.getSeqOverlapsBatch(q, maxOverlaps=11)
        got = (part.recs.tobytes(), part.query_off.tobytes(), part.stats.tobytes(), part.seed_hits,
               part.dp_groups, part.dp_elements, part.match_off.tobytes(), part.matches.tobytes())
        assert got == base, (kb, hb)
    monkeypatch.delenv("FG_KMER_BUDGET")
    monkeypatch.delenv("FG_HIT_BUDGET")
    # 64-bit sort keys (used when record index + position do not fit 32 bits) give the same result
    monkeypatch.setenv("FG_FORCE_KEY64", "1")
    monkeypatch.setenv("FG_PACKED_KEYS", "0")      # the plain 64-bit key + value form first
    k64 = det.getSeqOverlapsBatch(q, maxOverlaps=11)
    assert (k64.recs.tobytes(), k64.query_off.tobytes(), k64.stats.tobytes()) == base[:3]
    # ... whose LDS pieces are narrowed to 32 bits relative to the piece's minimum when the piece spans
    # little enough; force the wide fallback for most / all pieces too
    for lim in (3_000_000, 0):
        monkeypatch.setenv("FG_NARROW_MAX", str(lim))
        wide = det.getSeqOverlapsBatch(q, maxOverlaps=11)
        assert (wide.recs.tobytes(), wide.query_off.tobytes(), wide.stats.tobytes()) == base[:3], lim
    monkeypatch.delenv("FG_NARROW_MAX")
    # ... or packed into ONE 64-bit record per hit (record, curPos, extPos; ordered on the upper bits)
    monkeypatch.setenv("FG_PACKED_KEYS", "1")
    for lim in (None, 3_000_000, 0):
        if lim is not None:
            monkeypatch.setenv("FG_NARROW_MAX", str(lim))
        pk = det.getSeqOverlapsBatch(q, maxOverlaps=11)
        assert (pk.recs.tobytes(), pk.query_off.tobytes(), pk.stats.tobytes(), pk.match_off.tobytes(),
                pk.matches.tobytes()) == base[:3] + base[6:], lim
    monkeypatch.setenv("FG_SORT_STREAM_MAX", "600")      # closed-form partitions on the packed records too
    pk = det.getSeqOverlapsBatch(q, maxOverlaps=11)
    assert (pk.recs.tobytes(), pk.query_off.tobytes(), pk.stats.tobytes()) == base[:3]
    monkeypatch.delenv("FG_SORT_STREAM_MAX")
    monkeypatch.delenv("FG_NARROW_MAX")
    monkeypatch.delenv("FG_PACKED_KEYS")
    monkeypatch.delenv("FG_FORCE_KEY64")
    # the stream layout is invisible too: chaining classes all on the main stream, LDS sort batches on the side
    # stream, group classes cut elsewhere
    for env in ({"FG_CHAIN_STREAMS": "1"}, {"FG_SORT_STREAMS": "2"}, {"FG_CHAIN_HUGE_MIN": "300"},
                {"FG_CHAIN_STREAMS": "1", "FG_KMER_BUDGET": "200000"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        alt = det.getSeqOverlapsBatch(q, maxOverlaps=11)
        got = (alt.recs.tobytes(), alt.query_off.tobytes(), alt.stats.tobytes(), alt.seed_hits,
               alt.dp_groups, alt.dp_elements, alt.match_off.tobytes(), alt.matches.tobytes())
        assert got == base, env
        for k_ in env:
            monkeypatch.delenv(k_)


def test_edge_cases(built):
    from flye_amd import config, gpu, synth
    cfg = config.preset("raw")
    rs = synth.simulate(seed=5, genome_len=20_000, coverage=12, kind="pb_raw").filter_min_len(1000)
    # add degenerate reads: shorter than k, exactly k, and an unrelated read with no hits
    from flye_amd.synth import ReadSet
    extra_len = np.array([5, 17, 18, 3000], np.int32)
    rng = np.random.default_rng(3)
    words = [rs.words]
    offs = list(rs.word_off)
    for L in extra_len:
        nw = (int(L) + 31) // 32
        words.append(rng.integers(0, 1 << 63, size=nw, dtype=np.uint64))
        offs.append(offs[-1] + nw)
    rs2 = ReadSet(np.concatenate(words), np.array(offs, np.uint64), np.concatenate([rs.length, extra_len]),
                  np.zeros(rs.n + 4, np.int64), np.zeros(rs.n + 4, np.uint8), int(rs.total_bases + extra_len.sum()))
    ctx, vi, st, det = _gpu_setup(rs2, cfg)
    from oracle import oracle as O
    o, ost = _oracle_setup(rs2, cfg)
    assert _same_index(vi.export(), o.export_index())
    q = np.arange(0, 2 * rs2.n, dtype=np.uint32)
    gres = det.getSeqOverlapsBatch(q)
    ores = o.overlaps(O.detector_params(cfg), q)
    assert gres.lines() == ores.lines()
    # empty batch
    empty = det.getSeqOverlapsBatch(np.empty(0, np.uint32))
    assert len(empty.recs) == 0 and len(empty.query_off) == 1
    det.p.keep_alignment = 1
    empty = det.getSeqOverlapsBatch(np.empty(0, np.uint32))
    assert len(empty.recs) == 0 and len(empty.match_off) == 1 and len(empty.matches) == 0
    aln = det.getSeqOverlapsBatch(q)
    assert aln.recs.tobytes() == gres.recs.tobytes()
    det.p.keep_alignment = 0
    # unsupported flag combinations fail loudly
    det.p.partition_bad_mappings = 1
    with pytest.raises(gpu.FlyeGpuError) as e:
        det.getSeqOverlapsBatch(q[:2], maxOverlaps=2)
    assert e.value.code == -7
    det.p.partition_bad_mappings = 0
    with pytest.raises(gpu.FlyeGpuError):
        det.getSeqOverlapsBatch(np.array([2 * rs2.n + 10], np.uint32))   # id outside the container
    # call order
    ctx2 = gpu.Context(17, 0)
    ctx2.set_reads(rs)
    det2 = gpu.OverlapDetector.for_assemble(ctx2, gpu.VertexIndex(ctx2, 1.0), cfg)
    with pytest.raises(gpu.FlyeGpuError) as e:
        det2.getSeqOverlapsBatch(q[:2])
    assert e.value.code == -4


def test_full_size_properties(built):
    """BASELINE.json configs[1] at full size (E. coli PB 50x, ~230 Mbp): size-independent
    properties + a sampled record-for-record comparison with the oracle."""
    from flye_amd import config, gpu, workloads
    from oracle import oracle as O
    rs, min_ovlp, preset = workloads.ecoli_pb50()
    cfg = config.preset(preset)
    k = int(cfg["kmer_size"])
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    ex = vi.export()
    # index invariants: keys ascending & unique, lists ascending, every entry decodes to its key
    assert np.all(np.diff(ex.keys.astype(np.int64)) > 0)
    cnt = np.diff(ex.key_off.astype(np.int64))
    assert cnt.sum() == len(ex.entries) == st["index_entries"]
    d = np.diff(ex.entries.astype(np.int64))
    inner = np.ones(len(ex.entries) - 1, bool)
    inner[(ex.key_off[1:-1].astype(np.int64) - 1)[cnt[:-1] > 0]] = False
    assert np.all(d[inner] > 0)
    rng = np.random.default_rng(0)
    pick = rng.integers(0, len(ex.entries), size=1_000_000)
    rec = (ex.entries[pick] >> np.uint64(32)).astype(np.int64)
    pos = (ex.entries[pick] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    read = rec >> 1
    L = rs.length[read].astype(np.int64)
    q = np.where(rec & 1, L - pos - k, pos)           # forward position of the stored k-mer
    w0 = rs.word_off[read].astype(np.int64) + (q >> 5)
    sh = ((q & 31) * 2).astype(np.uint64)
    lo = rs.words[w0] >> sh
    hi = np.where(sh > 0, rs.words[np.minimum(w0 + 1, len(rs.words) - 1)] << ((np.uint64(64) - sh) % np.uint64(64)), 0)
    mask = np.uint64((1 << (2 * k)) - 1)
    x = (lo | hi.astype(np.uint64)) & mask
    fw = np.zeros_like(x)
    t = x.copy()
    for _ in range(k):
        fw = (fw << np.uint64(2)) | (t & np.uint64(3))
        t >>= np.uint64(2)
    rv = (~x) & mask
    canon = np.minimum(fw, rv)
    key_of_entry = ex.keys[np.searchsorted(ex.key_off, pick.astype(np.uint64), side="right") - 1]
    assert np.array_equal(canon, key_of_entry)
    assert np.array_equal((rv < fw), (rec & 1).astype(bool))    # stored in the canonical orientation
    # the full all-vs-all pass, twice: deterministic.  The detector runs with minimumOverlap = 1000 whatever
    # --min-ovlp says (main_assemble.cpp:174, :231; --min-ovlp only filters read length, :183) -- the bench
    # configuration
    det_min = config.DETECTOR_MIN_OVERLAP
    assert det.p.min_overlap == det_min
    allq = np.arange(0, 2 * rs.n, 2, dtype=np.uint32)
    r1 = det.getSeqOverlapsBatch(allq)
    r2 = det.getSeqOverlapsBatch(allq)
    assert r1.recs.tobytes() == r2.recs.tobytes() and len(r1.recs) > 100_000
    rr = r1.recs
    assert np.all(rr["cur_end"] - rr["cur_begin"] >= det_min) and np.all(rr["ext_end"] - rr["ext_begin"] >= det_min)
    assert np.all(rr["cur_end"] < rr["cur_len"]) and np.all(rr["ext_end"] < rr["ext_len"])
    assert np.all(rr["seq_divergence"] < 1.0)
    # ascending extId inside each query list (reference emission order)
    same_q = rr["cur_id"][1:] == rr["cur_id"][:-1]
    assert np.all(rr["ext_id"][1:][same_q] > rr["ext_id"][:-1][same_q])
    # sampled oracle comparison on the same (device-built, invariant-checked) index
    o = O.Oracle(k)
    o.set_reads(rs)
    o.import_index(O.IndexExport(ex.keys, ex.key_off, ex.entries, ex.repetitive), vi.getSampleRate())
    sample = np.sort(rng.choice(rs.n, size=300, replace=False))
    sq = (2 * sample).astype(np.uint32)
    ores = o.overlaps(O.detector_params(cfg, min_overlap=det_min), sq)
    want = ores.lines()
    got = []
    for i in sample:
        part = r1.of(int(i))
        bits = part["seq_divergence"].view(np.uint32)
        got += [f"{p['cur_id']} {p['cur_begin']} {p['cur_end']} {p['cur_len']} {p['ext_id']} {p['ext_begin']} "
                f"{p['ext_end']} {p['ext_len']} {p['score']} {bits[j]:08x}" for j, p in enumerate(part)]
    assert got == want
    # and with --min-ovlp (the N90 value) as the detector's minOverlap -- what Extender's safeOverlap sees
    det.p.min_overlap = min_ovlp
    strict = det.getSeqOverlapsBatch(sq[:120])
    assert len(strict.recs) > 0 and strict.lines() == o.overlaps(O.detector_params(cfg, min_overlap=min_ovlp), sq[:120]).lines()


def test_randomised_parity_sweep(built):
    """A fixed-seed slice of tools/fuzz_parity.py: random read models, presets, k, windows,
    detector flags and query mixes; index and every overlap record must equal the oracle's."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(seed=5, n_cases=10, verbose=False) == 0


def test_repeat_graph_flag_set_within_time_budget(built):
    """RepeatGraph::build's detector (repeat_graph.cpp:84-93): no overhang test, kmerMatches kept, all
    primaries, base-level divergence, bad mappings partitioned -- on noisy-overlap input (HiFi-like reads
    of a small repeat-rich genome, every read against every read) where most chains join unrelated
    substrings, so that the edit distances run into the thousands.  This is the configuration of the
    randomised sweep's seed-41 case 9, which took the round-1 oracle 5 minutes; the device side must stay
    bounded: O(ND) gives up after ED_EMAX rounds, the banded bit-vector kernel finishes the rest."""
    import time
    from flye_amd import config, gpu, synth
    from oracle import oracle as O
    rs = synth.simulate(seed=4109, genome_len=46_000, coverage=28, kind="hifi", n_homopolymers=30, n_tandems=30,
                        n_repeat_families=9).filter_min_len(1000)
    cfg = config.preset("hifi")
    ctx = gpu.Context(17, 0)
    ctx.set_reads(rs)
    vi = gpu.VertexIndex(ctx, 2.0)
    vi.buildIndexMinimizers(1, 10, cfg["repeat_kmer_rate"])
    det = gpu.OverlapDetector(ctx, vi, int(cfg["maximum_jump"]), 2000, 0, True, False, 0.3, True, True, True)
    q = np.arange(0, 2 * rs.n, dtype=np.uint32)
    det.getSeqOverlapsBatch(q[:4])                      # first-call allocations stay out of the budget
    t = time.perf_counter()
    gres = det.getSeqOverlapsBatch(q)
    dt = time.perf_counter() - t
    kt = ctx.kernel_times()
    o = O.Oracle(17)
    o.set_reads(rs)
    o.build_index_minimizers(1, 10, cfg["repeat_kmer_rate"])
    p = O.detector_params(cfg, min_overlap=2000, max_divergence=0.3, only_max_ext=False, max_overhang=0,
                          nucl_alignment=True, keep_alignment=True, partition_bad_mappings=True)
    ores = o.overlaps(p, q)
    assert gres.lines() == ores.lines() and len(gres.recs) > 1000
    assert np.array_equal(gres.recs["edit_distance"], ores.recs["edit_distance"])
    assert np.array_equal(gres.needs_trim, ores.needs_trim) and np.array_equal(gres.matches, ores.matches)
    assert int(gres.recs["edit_distance"].max()) > 512      # beyond ED_EMAX: the bit-vector kernel was needed ...
    assert "k_edit_myers" in kt
    assert dt < 5.0, f"{dt:.2f} s for {len(gres.recs)} records: {kt}"   # ... and the pass stays bounded


@pytest.mark.parametrize("name", ["raw_pb", "hifi_rc_max"])
def test_lookup_table_split_into_parts(built, golden_cases, monkeypatch, name):
    """The narrow lookup table keeps a 30-bit key index per slot and is split by key range when a part would
    hold more than 2^30 - 2 keys (D. melanogaster: 565 M keys, one part; 10 Gbp of reads: 1.4 G keys, two).
    FG_TABLE_PART_KEYS forces the split at golden-case size: results must not change."""
    case = golden_cases[name]
    rs = golden_reads(case)
    from flye_amd import config
    cfg = config.preset(case["preset"])
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    monkeypatch.setenv("FG_TABLE_PART_KEYS", str(st["selected_kmers"] // 5 + 1))
    ctx2, vi2, st2, det2 = _gpu_setup(rs, cfg)
    assert index_digest(vi2.export()) == case["index"]["sha256"]
    det2.p.max_divergence = bits_to_float(case["max_div_bits"])
    res = det2.getSeqOverlapsBatch(case_queries(case, rs.n), forceLocal=case.get("force_local", False),
                                   maxOverlaps=case.get("max_overlaps", 0))
    assert res.lines() == golden_lines(name)


@pytest.mark.parametrize("name", ["raw_pb", "hifi", "corrected_local"])
def test_index_build_batches_and_slices_are_invisible(built, golden_cases, monkeypatch, name):
    """The selection runs in batches of reads (FG_INDEX_BATCH_KMERS k-mer positions of scratch) and the
    sort in slices of key bins (FG_INDEX_SLICE_ENTRIES): forced to dozens of batches and slices at golden-case
    size, the index must still be the reference's (vertex_index.cpp:25-125, :389-483), bit for bit."""
    case = golden_cases[name]
    rs = golden_reads(case)
    from flye_amd import config
    cfg = config.preset(case["preset"])
    monkeypatch.setenv("FG_INDEX_BATCH_KMERS", str(int(rs.total_bases) // 37))
    monkeypatch.setenv("FG_INDEX_SLICE_ENTRIES", str(max(1000, int(case["index"]["index_entries"]) // 23)))
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    check_index_stats(st, case["index"])
    assert index_digest(vi.export()) == case["index"]["sha256"]
    det.p.max_divergence = bits_to_float(case["max_div_bits"])
    res = det.getSeqOverlapsBatch(case_queries(case, rs.n), forceLocal=case.get("force_local", False),
                                  maxOverlaps=case.get("max_overlaps", 0))
    assert res.lines() == golden_lines(name)     # the "owns an entry" bits come from the batched selection too


def test_solid_selection_in_steps_with_sliced_counters(built, golden_cases):
    """fg_index_kmer_hist / count_slice / batch_freq / batch_select / selection_done: the counters of a key RANGE
    only (what one rank of several holds).  Two 'ranks' played one after the other on one context: each counts
    its own range and writes the frequencies it knows; their sum is the complete array (checked against a
    whole-range count), the selection from it equals the one-call build's, and a build_range outside the counted
    range is refused."""
    import torch
    from flye_amd import config, dist, gpu
    case = golden_cases["raw_pb"]
    rs = golden_reads(case)
    cfg = config.preset("raw")
    ctx = gpu.Context(17, 0)
    ctx.set_reads(rs)
    vi = gpu.VertexIndex(ctx, 1.0)
    khist = vi.kmer_hist()
    assert int(khist.sum()) == int(np.maximum(rs.length.astype(np.int64) - 17, 0).sum())
    ranges = dist.balanced_bin_ranges(khist, 2)
    dev = torch.device("cuda", 0)
    parts, distinct = [], 0
    for r in range(2):
        d, nb = vi.count_slice(cfg, *ranges[r])
        distinct += d
        assert nb == 1
        ptr, n = vi.batch_freq(0)
        parts.append(dist._view(ptr, n, dev, "<i4").clone())
        if r == 1:
            with pytest.raises(gpu.FlyeGpuError):       # selection not finished
                vi.build_range(*ranges[1])
    assert distinct == int(case["index"]["total_kmers"])
    assert int(((parts[0] != 0) & (parts[1] != 0)).sum()) == 0      # every k-mer is counted by exactly one range
    # rank 1's context state is live: complete its array with rank 0's share, select, build ITS range
    ptr, n = vi.batch_freq(0)
    dist._view(ptr, n, dev, "<i4").add_(parts[0])
    vi.batch_select(0)
    hist = vi.selection_done()
    with pytest.raises(gpu.FlyeGpuError):               # rank 0's range was not counted here
        vi.build_range(*ranges[0])
    vi.build_range(*ranges[1])
    # the whole build in one call selects the same positions per bin
    ctx1 = gpu.Context(17, 0)
    ctx1.set_reads(rs)
    vi1 = gpu.VertexIndex(ctx1, 1.0)
    assert np.array_equal(vi1.begin(cfg), hist)
    assert int(hist.sum()) > 0
    ctx.close(); ctx1.close()


def test_import_rejects_malformed_arrays(built, golden_cases):
    """fg_import_index / fg_index_gather_end check the CSR on the device before anything reads lists through it."""
    from flye_amd import config, gpu
    case = golden_cases["raw_pb"]
    rs = golden_reads(case)
    cfg = config.preset("raw")
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    ex = vi.export()
    good = index_digest(ex)
    for breakit in ("first", "last", "order", "keys"):
        keys, off = ex.keys.copy(), ex.key_off.copy()
        if breakit == "first":
            off[0] = 1
        elif breakit == "last":
            off[-1] += 1
        elif breakit == "order":
            off[5], off[6] = off[6] + 3, off[5]
        else:
            keys[10], keys[11] = keys[11], keys[10]
        with pytest.raises(gpu.FlyeGpuError) as e:
            vi.import_index(gpu.IndexExport(keys, off, ex.entries, ex.repetitive), vi.getSampleRate())
        assert e.value.code == -3
    vi.import_index(ex, vi.getSampleRate())
    assert index_digest(vi.export()) == good
    ctx.close()


def test_memory_stats_follow_the_build(built, golden_cases):
    from flye_amd import config, gpu
    case = golden_cases["hifi"]
    rs = golden_reads(case)
    cfg = config.preset("hifi")
    now0, _ = gpu.memory_stats(reset_peak=True)
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    now1, peak1 = gpu.memory_stats()
    assert peak1 >= now1 > now0
    ctx.close()
    now2, _ = gpu.memory_stats()
    assert now2 <= now0       # (contexts of earlier tests may have been collected meanwhile)


@pytest.mark.parametrize("mode", ["direct", "hash"])
def test_kmer_counter_forms_agree(built, golden_cases, monkeypatch, mode):
    """KmerCounter (vertex_index.cpp:499-616) as a direct-addressed array (large read sets) and as a hashed table
    sized by the input (small ones): the same index either way."""
    case = golden_cases["raw_ont_rc"]
    rs = golden_reads(case)
    from flye_amd import config
    monkeypatch.setenv("FG_COUNT_MODE", mode)
    ctx, vi, st, det = _gpu_setup(rs, config.preset(case["preset"]))
    check_index_stats(st, case["index"])
    assert index_digest(vi.export()) == case["index"]["sha256"]
    kt = ctx.kernel_times()
    assert "k_count" in kt
    ctx.close()


@pytest.mark.parametrize("name", ["raw_ont_rc", "hifi", "raw_local"])
def test_partitioned_probes_are_invisible(built, golden_cases, monkeypatch, name):
    """A lookup table beyond the caches is probed region by region (the probes of a batch are radix-partitioned by
    table region first, fg_overlap.hip k_probe_emit / k_probe_sorted).  Forced on at golden-case size, in several
    sub-batches: same records, reverse-complement queries and self hits included."""
    case = golden_cases[name]
    rs = golden_reads(case)
    from flye_amd import config
    cfg = config.preset(case["preset"])
    monkeypatch.setenv("FG_PROBE_PARTITION", "1")
    monkeypatch.setenv("FG_PROBE_SUB_KMERS", str(int(rs.total_bases) // 7))
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    det.p.max_divergence = bits_to_float(case["max_div_bits"])
    res = det.getSeqOverlapsBatch(case_queries(case, rs.n), forceLocal=case.get("force_local", False),
                                  maxOverlaps=case.get("max_overlaps", 0))
    assert res.lines() == golden_lines(name)
    assert "k_probe_emit" in ctx.kernel_times()
    ctx.close()


@pytest.mark.parametrize("name,env", [("raw_ont_rc", {"FG_SORT_MANY_MIN": "2", "FG_SORT_STREAM_MANY": "100000"}),
                                      ("hifi", {"FG_SORT_MANY_MIN": "2", "FG_SORT_STREAM_MANY": "100000"}),
                                      # the record forms of large inputs: packed 64-bit records, 64-bit keys + values
                                      ("raw_pb", {"FG_SORT_MANY_MIN": "2", "FG_SORT_STREAM_MANY": "100000", "FG_FORCE_KEY64": "1"}),
                                      ("hifi", {"FG_SORT_MANY_MIN": "2", "FG_SORT_STREAM_MANY": "100000", "FG_FORCE_KEY64": "1",
                                                "FG_PACKED_KEYS": "0"})])
def test_sort_tiers_are_invisible(built, golden_cases, monkeypatch, name, env):
    """The many-pieces streaming rule of the hit sort's levels (large chunks only by default), forced on at golden-case
    size inside the overlap stage itself: same records."""
    case = golden_cases[name]
    rs = golden_reads(case)
    from flye_amd import config
    cfg = config.preset(case["preset"])
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    det.p.max_divergence = bits_to_float(case["max_div_bits"])
    res = det.getSeqOverlapsBatch(case_queries(case, rs.n), forceLocal=case.get("force_local", False),
                                  maxOverlaps=case.get("max_overlaps", 0))
    assert res.lines() == golden_lines(name)
    ctx.close()


def test_bridge_bulk_mode_for_unpredictable_walks(built):
    """A caller that asks for the container's reads in an order nobody can guess (Extender starts from the reads in
    hash order, extender.cpp:376-381; estimateGlobalCoverage picks them with rand(), chimera.cpp:76): after
    FGB_BULK_TRIGGER requests of a class the dispatcher computes ALL forward records of that class in a few large
    device calls and answers from them.  Lists stay the direct ones, in both classes the assemble stage uses."""
    import itertools
    import threading
    from flye_amd import config, gpu, synth
    rs = synth.simulate(seed=41, genome_len=150_000, coverage=30, kind="pb_raw", n_tandems=20).filter_min_len(1000)
    cfg = config.preset("raw")
    ctx, vi, st, det = _gpu_setup(rs, cfg)
    det.p.max_divergence = 0.3
    fwd = np.arange(0, 2 * rs.n, 2, dtype=np.uint32)
    direct = {0: det.getSeqOverlapsBatch(fwd), 100: det.getSeqOverlapsBatch(fwd, maxOverlaps=100)}
    oc = gpu.BatchingOverlapContainer(det, max_batch=64, linger_us=100)
    order = np.random.default_rng(3).permutation(len(fwd))
    errors = []
    counter = itertools.count()
    lock = threading.Lock()

    def worker():
        try:
            while True:
                with lock:
                    j = next(counter)
                if j >= 2 * len(order):
                    return
                mo = 0 if j % 2 == 0 else 100
                i = int(order[j // 2])
                got = oc.quickSeqOverlaps(int(fwd[i]), mo, False)
                assert got.tobytes() == direct[mo].of(i).tobytes(), (i, mo)
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=worker) for _ in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    s = oc.stats()
    assert s["requests"] == 2 * len(fwd)
    assert s["ahead_hits"] > 0.8 * s["requests"], s
    assert s["device_calls"] < 0.1 * s["requests"], s
    oc.close()
    ctx.close()


@pytest.mark.parametrize("name", ["raw_pb", "hifi", "repeat_raw"])
def test_two_lanes_are_invisible(built, golden_cases, monkeypatch, name):
    """Sub-ranges of a call's queries run on two lanes side by side (a second set of scratch, streams and a helper
    thread, fg_overlap.hip) when a chunk's hits exceed the budget; forced here at golden-case size -- tiny budgets, so
    that both the budget-driven cut and the forced one are taken: same records, same order, kmerMatches included."""
    case = golden_cases[name]
    rs = golden_reads(case)
    from flye_amd import config
    cfg = config.preset(case["preset"])
    from flye_amd import gpu
    if name.startswith("repeat"):
        from helpers import repeat_stage_setup
        wnd, dk = repeat_stage_setup(case, cfg)
        ctx = gpu.Context(int(cfg["kmer_size"]), 0)
        ctx.set_reads(rs, 0)
        vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
        vi.buildIndexMinimizers(1, wnd, cfg["repeat_kmer_rate"])
        det = gpu.OverlapDetector(ctx, vi, int(cfg["maximum_jump"]), dk["min_overlap"], dk["max_overhang"], True,
                                  dk["only_max_ext"], dk["max_divergence"], dk["nucl_alignment"], True,
                                  bool(cfg["hpc_scoring_on"]))
    else:
        ctx, vi, st, det = _gpu_setup(rs, cfg)
        det.p.max_divergence = bits_to_float(case["max_div_bits"])
    q = np.arange(0, 2 * rs.n, dtype=np.uint32) if name.startswith("repeat") else case_queries(case, rs.n)
    one = det.getSeqOverlapsBatch(q, forceLocal=case.get("force_local", False), maxOverlaps=case.get("max_overlaps", 0))
    for env in ({"FG_HIT_BUDGET": "20000", "FG_KMER_BUDGET": str(1 << 30)}, {"FG_LANE_MIN_HITS": "1000", "FG_LANE_SPLIT": "3"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        two = det.getSeqOverlapsBatch(q, forceLocal=case.get("force_local", False), maxOverlaps=case.get("max_overlaps", 0))
        assert two.recs.tobytes() == one.recs.tobytes() and np.array_equal(two.query_off, one.query_off)
        if det.p.keep_alignment:
            assert np.array_equal(two.match_off, one.match_off) and np.array_equal(two.matches, one.matches)
            assert np.array_equal(two.needs_trim, one.needs_trim)
        for k_ in env:
            monkeypatch.delenv(k_)
    if not name.startswith("repeat"):
        assert one.lines() == golden_lines(name)
    ctx.close()
