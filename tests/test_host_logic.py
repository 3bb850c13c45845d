import numpy as np


def test_complement_is_involution_and_matches_definition():
    from flye_amd import gpu
    r = np.zeros(3, gpu.REC_DTYPE)
    r["cur_id"] = [0, 4, 7]; r["ext_id"] = [3, 8, 10]
    r["cur_begin"] = [10, 0, 5]; r["cur_end"] = [2000, 1500, 999]; r["cur_len"] = [3000, 1600, 1000]
    r["ext_begin"] = [7, 100, 0]; r["ext_end"] = [1900, 1700, 900]; r["ext_len"] = [2500, 1800, 901]
    c = gpu.complement(r)
    # overlap.h:118-147: begin/end swap and map x -> len - x - 1, ids to their rc
    assert list(c["cur_begin"]) == [3000 - 2000 - 1, 1600 - 1500 - 1, 0]
    assert list(c["cur_end"]) == [3000 - 10 - 1, 1599, 1000 - 5 - 1]
    assert list(c["cur_id"]) == [1, 5, 6] and list(c["ext_id"]) == [2, 9, 11]
    back = gpu.complement(c)
    for f in r.dtype.names:
        assert np.array_equal(back[f], r[f])


def test_shard_queries_partition():
    from flye_amd import dist
    n, first = 1001, 40
    ids = [dist.shard_queries(n, r, 8, first) for r in range(8)]
    allids = np.sort(np.concatenate(ids))
    assert np.array_equal(allids, first + 2 * np.arange(n))
    assert max(len(x) for x in ids) - min(len(x) for x in ids) <= 1
    assert all(np.all(dist.owner_of((x - first) // 2, 8) == r) for r, x in enumerate(ids))


def test_synth_is_deterministic():
    from flye_amd import synth
    a = synth.simulate(seed=9, genome_len=20000, coverage=5, kind="ont_raw")
    b = synth.simulate(seed=9, genome_len=20000, coverage=5, kind="ont_raw")
    c = synth.simulate(seed=10, genome_len=20000, coverage=5, kind="ont_raw")
    assert np.array_equal(a.words, b.words) and np.array_equal(a.length, b.length)
    assert not np.array_equal(a.length, c.length)
    sub = a.subset([0, 2])
    assert sub.n == 2 and sub.length[1] == a.length[2]


def test_bench_byte_model_and_stage_attribution():
    """bench.py's algorithmic bytes (16.25 + 44 m + 20 d B/bp): the kernels of a stage share the
    stage's bytes, k_chain_small is credited with the DP elements of the groups it processed only, and the stage
    figures add up to the whole formula."""
    import bench
    bp, m, d, ovl, d_small = 1e6, 1.5, 1.1, 0.008, 0.8
    per = {k: bench.algorithmic_bytes(k, bp, m, d, ovl, d_small) for k in
           ("k_probe", "k_fill", "k_sort_level", "k_sort_lds", "k_chain_dp", "k_chain_small", "k_group_prep")}
    assert per["k_probe"] == (0.25 + 16.0) * bp
    assert per["k_fill"] == m * 20.0 * bp
    assert per["k_sort_level"] == per["k_sort_lds"] == m * 24.0 * bp
    assert per["k_chain_dp"] == (d * 20.0 + 44.0 * ovl) * bp
    assert per["k_chain_small"] == d_small * 20.0 * bp < per["k_chain_dp"]
    assert per["k_group_prep"] == 0.0
    assert bench.algorithmic_bytes("k_chain_small", bp, m, d, ovl) == per["k_chain_dp"]     # no counter: the stage's bytes
    times = {"k_probe": (1e-3, 1), "k_fill": (1e-3, 1), "k_exscan": (1e-4, 3), "k_sort_level": (2e-3, 20), "k_sort_lds": (1e-3, 1),
             "k_chain_small": (2e-3, 1), "k_group_prep": (1e-3, 2), "k_prim_gather": (1e-4, 1), "host:shim": (5e-3, 1)}
    st = bench.stage_rooflines(times, bp, m, d, ovl, None)
    assert set(st) == {"seed_collect", "hit_sort", "chain"}
    total = sum(v["algorithmic_bytes"] for v in st.values())
    assert abs(total - (16.25 + 44.0 * m + 20.0 * d + 44.0 * ovl) * bp) <= 3
    assert st["hit_sort"]["kernels"] == ["k_sort_lds", "k_sort_level"] and abs(st["hit_sort"]["exclusive_ms"] - 3.0) < 1e-9
    assert "traffic" not in st["chain"]
