"""Exact edit distance (the base-level divergence step, reference src/sequence/alignment.cpp:218-247
-> edlibAlign(NW, TASK_DISTANCE, k = -1), src/sequence/edlib.cpp:141-296).

tests/golden/edlib_pairs.json holds what the REFERENCE's edlib returned for seeded string pairs
(tests/golden/make_edlib_golden.py).  CPU: the oracle's bit-vector restatement and its plain DP
against those values; GPU: both device kernels (O(ND) and the banded bit-vector one, one wave and
a workgroup per pair) through the C ABI against the same values."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, edit_pair, hpc, pairs_readset


def _golden():
    with open(os.path.join(GOLDEN, "edlib_pairs.json")) as f:
        return json.load(f)


def test_oracle_edit_distance_equals_reference_edlib(built):
    from oracle import oracle as O
    for g in _golden():
        a, b = edit_pair(g["spec"])
        assert (len(a), len(b)) == (g["n"], g["m"])
        assert O.edit_distance(a, b) == g["dist"], g["spec"]
        ha, hb = hpc(a), hpc(b)
        assert (len(ha), len(hb)) == (g["hpc_n"], g["hpc_m"])
        assert O.edit_distance(ha, hb) == g["hpc_dist"], g["spec"]
        if g["n"] * max(1, g["dist"]) < 3e7:      # the scalar DP on what it finishes quickly
            assert O.edit_distance_dp(a, b) == g["dist"]


def test_oracle_band_doubling_from_any_start(built):
    """The band logic (column ranges per 64-row block, +1 boundaries, score bookkeeping) must give
    the exact distance from whatever k the doubling starts at."""
    from oracle import oracle as O
    rng = np.random.default_rng(7)
    for _ in range(300):
        n = int(rng.integers(0, 700))
        spec = dict(seed=int(rng.integers(1, 1 << 30)), n=n, err=float(rng.choice([0.0, 0.02, 0.2, 1.0])),
                    m=int(rng.integers(0, 700)))
        if rng.integers(0, 3) == 0 and spec["err"] < 1:
            spec["shift"] = int(rng.integers(0, 100))
        a, b = edit_pair(spec)
        want = O.edit_distance_dp(a, b)
        for k0 in (1, 2, 5, 64, 100000):
            assert O.edit_distance_k0(a, b, k0) == want, (spec, k0)


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "ref_dumper")),
                    reason="oracle/_ref/ref_dumper not built")
def test_oracle_edit_distance_live_against_reference(built):
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    pairs = []
    for _ in range(60):
        spec = dict(seed=int(rng.integers(1, 1 << 30)), n=int(rng.integers(1, 3000)),
                    err=float(rng.choice([0.0, 0.01, 0.1, 0.3, 1.0])), m=int(rng.integers(1, 3000)), hp=int(rng.integers(0, 30)))
        pairs.append(edit_pair(spec))
    assert [O.edit_distance(a, b) for a, b in pairs] == O.ref_edlib_distances(pairs)


# ---- device kernels ------------------------------------------------------------------------------
def _device_distances(pairs, use_hpc, env=None):
    from flye_amd import gpu
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = str(v)
    try:
        ctx = gpu.Context(17, 0)
        ctx.set_reads(pairs_readset(pairs))
        d, la, lb = ctx.debug_edit_distances(len(pairs), use_hpc)
        kt = ctx.kernel_times()
        ctx.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return d, la, lb, kt


@pytest.mark.gpu
@pytest.mark.parametrize("use_hpc", [False, True])
def test_device_edit_distance_equals_reference_edlib(built, use_hpc):
    gold = _golden()
    pairs = [edit_pair(g["spec"]) for g in gold]
    d, la, lb, kt = _device_distances(pairs, use_hpc)
    key = ("hpc_dist", "hpc_n", "hpc_m") if use_hpc else ("dist", "n", "m")
    assert la.tolist() == [g[key[1]] for g in gold] and lb.tolist() == [g[key[2]] for g in gold]
    assert d.tolist() == [g[key[0]] for g in gold]
    # all three kernels took part: O(ND), the one-wave and the workgroup bit-vector sweeps
    assert {"k_edit_distance", "k_edit_myers", "k_edit_myers_wide"} <= set(kt)


@pytest.mark.gpu
def test_device_bitvector_kernel_on_small_distances(built):
    """FG_ED_EMAX = 3 sends nearly every pair through the banded bit-vector kernel, FG_ED_LDS_BASES = 2048
    sends the longer ones there without an O(ND) attempt (band doubling starts at 64)."""
    gold = [g for g in _golden() if g["n"] <= 33000]
    pairs = [edit_pair(g["spec"]) for g in gold]
    d, la, lb, kt = _device_distances(pairs, False, env=dict(FG_ED_EMAX=3, FG_ED_LDS_BASES=2048))
    assert d.tolist() == [g["dist"] for g in gold]
    assert "k_edit_myers" in kt


@pytest.mark.gpu
def test_device_edit_distance_random_against_oracle(built):
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    pairs = []
    for _ in range(400):
        big = rng.integers(0, 8) == 0
        n = int(rng.integers(1, 20000 if big else 1500))
        spec = dict(seed=int(rng.integers(1, 1 << 30)), n=n, err=float(rng.choice([0.0, 0.003, 0.05, 0.15, 0.4, 1.0])),
                    m=int(rng.integers(1, 20000 if big else 1500)), hp=int(rng.integers(0, 40)))
        if rng.integers(0, 4) == 0 and spec["err"] < 1:
            spec["shift"] = int(rng.integers(0, min(n, 600)))
        pairs.append(edit_pair(spec))
    want = [O.edit_distance(a, b) for a, b in pairs]
    for env in (None, dict(FG_ED_EMAX=8)):
        d, la, lb, kt = _device_distances(pairs, False, env=env)
        assert d.tolist() == want
    hw = [O.edit_distance(hpc(a), hpc(b)) for a, b in pairs]
    d, la, lb, kt = _device_distances(pairs, True)
    assert d.tolist() == hw and la.tolist() == [len(hpc(a)) for a, _ in pairs]


@pytest.mark.gpu
def test_device_edit_distance_long_pairs_many_strips(built):
    """Repeat-stage sized substrings (disjointig against disjointig): dozens of 4096-row strips per pair, the
    8-wave workgroup's strip pipeline (each wave a few hundred columns behind the strip above), band doubling
    from 64 (no O(ND) attempt above 32 kb), unequal lengths; against the oracle's bit-vector form."""
    from oracle import oracle as O
    specs = [dict(seed=501, n=150_000, err=0.03), dict(seed=502, n=260_000, err=0.004, hp=500),
             dict(seed=503, n=70_000, err=1.0, m=61_000), dict(seed=504, n=120_000, err=0.08, shift=9_000),
             dict(seed=505, n=49_153, err=0.02), dict(seed=506, n=200_000, err=0.0)]
    pairs = [edit_pair(s) for s in specs]
    want = [O.edit_distance(a, b) for a, b in pairs]
    d, la, lb, kt = _device_distances(pairs, False)
    assert d.tolist() == want and la.tolist() == [len(a) for a, _ in pairs]
    assert "k_edit_myers_wide" in kt
    hw = [O.edit_distance(hpc(a), hpc(b)) for a, b in pairs]
    d, la, lb, kt = _device_distances(pairs, True)
    assert d.tolist() == hw


# ---- banded affine-gap alignment with CIGAR (getAlignmentCigarKsw, alignment.cpp:102-216) ----------
def _ksw_golden():
    with open(os.path.join(GOLDEN, "ksw_pairs.json")) as f:
        return json.load(f)


def _ksw_check(got, g):
    import hashlib
    bits, cig = got
    assert bits == g["err_bits"], g["spec"]
    assert hashlib.sha256(cig.encode()).hexdigest() == g["cigar_sha256"], g["spec"]
    if "cigar" in g:
        assert cig == g["cigar"]


def test_oracle_ksw_cigar_equals_reference(built):
    """The oracle's restatement of ksw_extz2 (same byte state as the vector code, incl. what it leaves around
    the band and past array ends) against the reference's getAlignmentCigarKsw: error-rate bits and CIGAR."""
    from oracle import oracle as O
    for g in _ksw_golden():
        a, b = edit_pair(g["spec"])
        assert (len(a), len(b)) == (g["tlen"], g["qlen"])
        _ksw_check(O.ksw_cigar(a, b), g)


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "ref_dumper")),
                    reason="oracle/_ref/ref_dumper not built")
def test_oracle_ksw_cigar_live_against_reference(built):
    from oracle import oracle as O
    rng = np.random.default_rng(21)
    pairs = []
    for _ in range(150):
        small = rng.integers(0, 2) == 0
        n = int(rng.integers(1, 90 if small else 2500))
        spec = dict(seed=int(rng.integers(1, 1 << 30)), n=n, err=float(rng.choice([0.0, 0.02, 0.15, 1.0])),
                    m=int(rng.integers(1, 90 if small else 2500)), hp=int(rng.integers(0, 20)))
        a, b = edit_pair(spec)
        if len(a) and len(b):
            pairs.append((a, b))
    assert [O.ksw_cigar(a, b) for a, b in pairs] == O.ref_ksw_cigars(pairs)


@pytest.mark.gpu
def test_device_ksw_cigar_equals_reference(built):
    """fg_align_cigar_ksw (the DP + backtrack on the device, decoding on the host) against what the reference's
    getAlignmentCigarKsw returned for the fixture pairs."""
    from flye_amd import gpu
    gold = _ksw_golden()
    pairs = [edit_pair(g["spec"]) for g in gold]
    ctx = gpu.Context(17, 0)
    got = ctx.align_cigar_ksw(pairs)
    for g, x in zip(gold, got):
        _ksw_check(x, g)
    assert "k_ksw_extz2" in ctx.kernel_times()


@pytest.mark.gpu
def test_device_ksw_cigar_random_against_oracle(built, monkeypatch):
    from flye_amd import gpu
    from oracle import oracle as O
    rng = np.random.default_rng(33)
    pairs = []
    for _ in range(500):
        small = rng.integers(0, 3) == 0
        n = int(rng.integers(1, 100 if small else 4000))
        spec = dict(seed=int(rng.integers(1, 1 << 30)), n=n, err=float(rng.choice([0.0, 0.01, 0.08, 0.25, 1.0])),
                    m=int(rng.integers(1, 100 if small else 4000)), hp=int(rng.integers(0, 30)))
        if rng.integers(0, 4) == 0 and spec["err"] < 1:
            spec["shift"] = int(rng.integers(0, min(n, 400)))
        a, b = edit_pair(spec)
        if len(a) and len(b):
            pairs.append((a, b))
    want = [O.ksw_cigar(a, b) for a, b in pairs]
    ctx = gpu.Context(17, 0)
    assert ctx.align_cigar_ksw(pairs) == want
    # the same in several scratch-bounded sub-batches
    monkeypatch.setenv("FG_KSW_SCRATCH_BYTES", str(40 << 20))
    assert ctx.align_cigar_ksw(pairs) == want


@pytest.mark.gpu
def test_device_ksw_lds_rings_equal_literal_state(built, monkeypatch):
    """Read-sized pairs, bands 64 .. 1024 and beyond: the kernel with the byte state in LDS rings returns what the
    literal one (whole arrays in memory, pinned above) returns, and both kernels ran."""
    from flye_amd import gpu
    rng = np.random.default_rng(44)
    pairs = []
    for i in range(120):
        n = int(rng.integers(3000, 30000))
        spec = dict(seed=int(rng.integers(1, 1 << 30)), n=n, err=float(rng.choice([0.0, 0.02, 0.12])), hp=int(rng.integers(0, 25)))
        if i % 3 == 1:
            spec["shift"] = int(rng.integers(100, 2500))      # lengths apart: the band doubles until it connects the corners
        if i % 3 == 2:
            spec["shift"] = int(rng.integers(0, 100))
        if i % 10 == 9:
            spec.update(err=1.0, m=int(n * rng.uniform(0.9, 1.1)))
        pairs.append(edit_pair(spec))
    pairs += [edit_pair(dict(seed=7, n=40, err=0.1, m=2000)), edit_pair(dict(seed=8, n=2000, err=0.1, m=40))]   # tiny target / query
    ctx = gpu.Context(17, 0)
    got = ctx.align_cigar_ksw(pairs)
    kt = ctx.kernel_times()
    assert "k_ksw_extz2_lds" in kt and "k_ksw_extz2" in kt
    monkeypatch.setenv("FG_KSW_LITERAL", "1")
    lit = ctx.align_cigar_ksw(pairs)
    assert "k_ksw_extz2_lds" not in ctx.kernel_times()
    assert got == lit
    assert len({len(c) for _, c in got}) > 50
