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
