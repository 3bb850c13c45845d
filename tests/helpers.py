import gzip
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_reads(case):
    from flye_amd import synth
    rs = synth.simulate(**case["sim"]).filter_min_len(case["min_read_len"])
    sha = hashlib.sha256(rs.words.tobytes() + rs.length.tobytes()).hexdigest()
    assert sha == case["reads_sha256"], "simulator no longer reproduces the golden inputs"
    return rs


def golden_lines(name):
    with gzip.open(os.path.join(GOLDEN, name + ".ovlp.gz"), "rt") as f:
        return [l.strip() for l in f if l.strip() and not l.startswith("#")]


def index_digest(ix):
    h = hashlib.sha256()
    for a in (ix.keys, ix.key_off, ix.entries, ix.repetitive):
        h.update(np.ascontiguousarray(a, np.uint64).tobytes())
    return h.hexdigest()


def bits_to_float(hexbits: str) -> float:
    return float(np.array([int(hexbits, 16)], np.uint32).view(np.float32)[0])


def case_queries(case, n_reads):
    start = 1 if case.get("rc_queries") else 0
    return np.arange(start, 2 * n_reads, 2, dtype=np.uint32)


def check_index_stats(st, gold):
    for f in ("total_kmers", "selected_kmers", "index_entries", "repetitive_kmers", "repetitive_frequency"):
        assert int(st[f]) == int(gold[f]), (f, st[f], gold[f])
    assert np.float32(st["sample_rate"]).view(np.uint32) == int(gold["sample_rate_bits"], 16)


def golden_queries(case):
    """Second container of a ReadAligner-style case (ids continue after the indexed ones)."""
    from flye_amd import synth
    return synth.simulate(**case["queries_sim"])


def edges_setup(case, cfg):
    """(window, detector kwargs) of a ReadAligner-style golden case."""
    wnd = int(cfg["minimizer_window"]) if cfg["use_minimizers"] else 1
    return wnd, dict(min_overlap=case["min_overlap"], only_max_ext=bool(case["only_max"]),
                     max_overhang=case["max_overhang"], nucl_alignment=case["nucl_aln"],
                     **({"keep_alignment": True} if case.get("keep_aln") else {}))


def repeat_stage_setup(case, cfg):
    """(window, detector kwargs) of a RepeatGraph::build-style golden case (repeat_graph.cpp:72-97)
    with partitionBadMappings on: the primaries that fail the gate come back marked."""
    wnd, dk = edges_setup(case, cfg)
    dk.update(max_divergence=float(np.float32(case["max_div"])), partition_bad_mappings=True)
    return wnd, dk


def check_repeat_stage_result(res, case, golden_cases):
    """`res` (oracle or GPU, partition_bad_mappings = 1, gate = case max_div) against the reference's
    vectors: the unmarked records are what the reference returns with that gate, marked + unmarked
    are what it returns with the gate open, and a record is marked iff it fails the gate."""
    lines = res.lines()
    trim = res.needs_trim.astype(bool)
    assert [l for l, t in zip(lines, trim) if not t] == golden_lines(case["name"])
    all_name = case["name"] + "_all"
    if all_name in golden_cases:
        assert lines == golden_lines(all_name)
    maxd = np.float32(case["max_div"])
    assert np.array_equal(trim, ~(res.recs["seq_divergence"] < maxd))
    assert trim.any() and not trim.all()


# ---- edit-distance pairs (tests/golden/edlib_pairs.json) -----------------------------------------
def edit_pair(spec):
    """One (a, b) pair of 0..3 arrays from a seeded description: b is a copy of a with
    sub/ins/del errors at rate ``err`` (err >= 1: unrelated random strings); ``hp`` > 0 plants
    homopolymer runs (for the compressed form)."""
    rng = np.random.default_rng(spec["seed"])
    n = spec["n"]
    a = rng.integers(0, 4, size=n, dtype=np.uint8)
    if spec.get("hp", 0):
        for _ in range(spec["hp"]):
            if n < 4:
                break
            p = int(rng.integers(0, max(1, n - 2)))
            l = int(rng.integers(2, 12))
            a[p:p + l] = a[p]
    err = spec["err"]
    if err >= 1:
        b = rng.integers(0, 4, size=spec.get("m", n), dtype=np.uint8)
        return a, b
    out = []
    i = 0
    r = rng.random(size=2 * n + 16)
    c = rng.integers(0, 4, size=2 * n + 16, dtype=np.uint8)
    t = 0
    while i < n:
        x = r[t]
        if x < err / 3:                       # substitution
            out.append((a[i] + 1 + c[t] % 3) & 3)
            i += 1
        elif x < 2 * err / 3:                 # insertion
            out.append(c[t])
        elif x < err:                         # deletion
            i += 1
        else:
            out.append(a[i])
            i += 1
        t += 1
    b = np.array(out, dtype=np.uint8)
    if "shift" in spec:                       # unequal lengths: drop a prefix of b
        b = b[spec["shift"]:]
    return a, b


def hpc(x):
    """homopolymerCompression (alignment.cpp:52-70) of a 0..3 array."""
    x = np.asarray(x, np.uint8)
    if len(x) == 0:
        return x
    keep = np.ones(len(x), bool)
    keep[1:] = x[1:] != x[:-1]
    return x[keep]


def pairs_readset(pairs):
    """ReadSet whose reads 2i, 2i+1 are the pair's strings (for fg_debug_edit_distances)."""
    from flye_amd import synth
    seqs = []
    for a, b in pairs:
        seqs += [a, b]
    return synth.ReadSet.from_arrays(seqs)


# ---- sampled oracle parity at sizes where the oracle cannot hold (or build) the whole index ---------------
def canonical_kmers(rs, r, k):
    """Canonical k-mers (Kmer repr, kmer.h:32-52) of every forward position of read r, incl. the one the
    reference's iteration drops (a superset is harmless here)."""
    L = int(rs.length[r])
    w = rs.words[int(rs.word_off[r]):int(rs.word_off[r + 1])]
    sh = np.arange(32, dtype=np.uint64) * np.uint64(2)
    b = ((w[:, None] >> sh[None, :]) & np.uint64(3)).reshape(-1)[:L].astype(np.uint64)
    n = L - k + 1
    if n <= 0:
        return np.empty(0, np.uint64)
    fw = np.zeros(n, np.uint64)
    rv = np.zeros(n, np.uint64)
    for t in range(k):
        fw = (fw << np.uint64(2)) | b[t:t + n]
        rv = rv | ((np.uint64(3) - b[t:t + n]) << np.uint64(2 * t))
    return np.minimum(fw, rv)


def sub_index(ex, rs, reads, k):
    """The part of an exported index the given query reads can touch: getSeqOverlaps looks up the k-mers
    of the query only (overlap.cpp:176-196), so the oracle gives the same records on this sub-index as on
    the whole one -- at a size it can import in seconds."""
    from oracle import oracle as O
    km = np.unique(np.concatenate([canonical_kmers(rs, int(r), k) for r in reads]))
    idx = np.searchsorted(ex.keys, km)
    ok = idx < len(ex.keys)
    ok[ok] &= ex.keys[idx[ok]] == km[ok]
    sel = idx[ok]
    off = ex.key_off.astype(np.int64)
    cnt = off[sel + 1] - off[sel]
    new_off = np.zeros(len(sel) + 1, np.int64)
    new_off[1:] = np.cumsum(cnt)
    pos = np.repeat(off[sel] - new_off[:-1], cnt) + np.arange(int(new_off[-1]), dtype=np.int64)
    return O.IndexExport(np.ascontiguousarray(ex.keys[sel]), new_off.astype(np.uint64),
                         np.ascontiguousarray(ex.entries[pos]), ex.repetitive)
