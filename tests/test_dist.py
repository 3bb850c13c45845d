"""N>1 layout on CPU: two gloo ranks shard the queries by sequence id, each computes
its shard (with the oracle standing in for the device), results are exchanged with
torch.distributed and must equal the single-process result.  Also rehearses the
bench's barrier + max-over-ranks timing reduction."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as td
    from flye_amd import config, dist, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    rs = synth.simulate(seed=31, genome_len=30_000, coverage=20, kind="hifi").filter_min_len(1000)
    cfg = config.preset("corrected")
    cfg["reads_base_alignment"] = 0.0   # keep the CPU test quick
    o = O.Oracle(17, threads=2)
    o.set_reads(rs)
    o.build_index(cfg)                  # index replicated on every rank
    mine = dist.shard_queries(rs.n, rank, world)
    res = o.overlaps(O.detector_params(cfg), mine)
    td.barrier()
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    assert t.item() == 0.5 + world - 1
    bp = torch.tensor([res.query_bp], dtype=torch.int64)
    td.all_reduce(bp, op=td.ReduceOp.SUM)
    gathered = [None] * world
    td.all_gather_object(gathered, (mine.tolist(), [res.lines()[int(res.query_off[i]):int(res.query_off[i + 1])]
                                                    for i in range(len(mine))]))
    if rank == 0:
        merged = dist.merge_sharded([g[0] for g in gathered], [g[1] for g in gathered])
        flat = [l for lst in merged for l in lst]
        allq = np.arange(0, 2 * rs.n, 2, dtype=np.uint32)
        ref = o.overlaps(O.detector_params(cfg), allq)
        assert flat == ref.lines()
        assert bp.item() == ref.query_bp == rs.total_bases
        open(os.path.join(out_dir, "ok"), "w").write(str(len(flat)))
    td.destroy_process_group()


def test_two_rank_sharding_gloo(built, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert int(open(tmp_path / "ok").read()) > 100


# ---- sharded index build: the exchange on CPU tensors ----------------------------------------------------
def _piece_worker(rank, world, port, out_dir):
    """Each rank holds the piece of the oracle's index that a key-range-sharded build would leave it with
    (bins cut by dist.balanced_bin_ranges); sums all-reduce + all-gather of the pieces over gloo must give
    every rank the oracle's whole index."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as td
    from flye_amd import config, dist, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    rs = synth.simulate(seed=77, genome_len=50_000, coverage=20, kind="pb_raw", n_tandems=20).filter_min_len(1000)
    cfg = config.preset("raw")
    k = 17
    o = O.Oracle(k, threads=2)
    o.set_reads(rs)
    o.build_index(cfg)
    full = o.export_index()
    # entries per key bin stand in for the device's histogram of accepted positions
    shift = max(0, 2 * k - 12)
    cnt = np.diff(full.key_off.astype(np.int64))
    hist = np.bincount((full.keys >> np.uint64(shift)).astype(np.int64), weights=cnt, minlength=4096)
    ranges = dist.balanced_bin_ranges(hist, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == 4096 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    lo, hi = ranges[rank]
    sel = ((full.keys >> np.uint64(shift)) >= lo) & ((full.keys >> np.uint64(shift)) < hi)
    idx = np.nonzero(sel)[0]
    rsel = ((full.repetitive >> np.uint64(shift)) >= lo) & ((full.repetitive >> np.uint64(shift)) < hi)
    if len(idx):
        a, b = int(full.key_off[idx[0]]), int(full.key_off[idx[-1] + 1])
        pk, po, pe = full.keys[idx], full.key_off[idx[0]:idx[-1] + 2] - np.uint64(a), full.entries[a:b]
    else:
        pk, po, pe = full.keys[:0], np.zeros(1, np.uint64), full.entries[:0]
    piece = tuple(torch.from_numpy(np.ascontiguousarray(x).view(np.int64)) for x in (pk, po, pe, full.repetitive[rsel]))
    # the two sums of filterFrequentKmers, piecewise, then over all ranks
    part = torch.tensor([int(len(pe)), int(len(pk))], dtype=torch.int64)
    td.all_reduce(part)
    assert part.tolist() == [len(full.entries), len(full.keys)]
    keys, off, ent, rep, (K, E, R), moved = dist.allgather_pieces(piece, rank, world, torch.device("cpu"))
    assert (K, E, R) == (len(full.keys), len(full.entries), len(full.repetitive))
    assert np.array_equal(keys[:K].numpy().view(np.uint64), full.keys)
    assert np.array_equal(off.numpy().view(np.uint64), full.key_off)
    assert np.array_equal(ent[:E].numpy().view(np.uint64), full.entries)
    assert np.array_equal(rep[:R].numpy().view(np.uint64), full.repetitive)
    assert moved >= 8 * (2 * K + E + R)
    # the host form of the same assembly
    pieces = [None] * world
    td.all_gather_object(pieces, (pk, po, pe, full.repetitive[rsel]))
    from flye_amd.gpu import IndexExport
    cat = dist.concat_pieces([IndexExport(*p) for p in pieces])
    assert np.array_equal(cat.keys, full.keys) and np.array_equal(cat.key_off, full.key_off)
    assert np.array_equal(cat.entries, full.entries) and np.array_equal(cat.repetitive, full.repetitive)
    open(os.path.join(out_dir, f"ok{rank}"), "w").write(str(K))
    td.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_index_pieces_allgather_gloo(built, tmp_path, world):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_piece_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(int(open(tmp_path / f"ok{r}").read()) > 1000 for r in range(world))


# ---- option B (index sharded by target read + seed-hit exchange): the receiver's order ---------------------------
def _kmers_with_flip(rs, r, k):
    """forward k-mers of read r at positions 0 .. len-k-1 (kmer.h:193-198): canonical value and whether it was flipped"""
    L = int(rs.length[r])
    w = rs.words[int(rs.word_off[r]):int(rs.word_off[r + 1])]
    sh = np.arange(32, dtype=np.uint64) * np.uint64(2)
    b = ((w[:, None] >> sh[None, :]) & np.uint64(3)).reshape(-1)[:L].astype(np.uint64)
    n = L - k
    if n <= 0:
        return np.empty(0, np.uint64), np.empty(0, bool)
    fw = np.zeros(n, np.uint64)
    rv = np.zeros(n, np.uint64)
    for t in range(k):
        fw = (fw << np.uint64(2)) | b[t:t + n]
        rv = rv | ((np.uint64(3) - b[t:t + n]) << np.uint64(2 * t))
    return np.minimum(fw, rv), rv < fw


def _emit_hits(ex, rs, r, k, keep=None):
    """Seed hits of forward read r against the exported index `ex` in the reference's emission order
    (overlap.cpp:176-196 + vertex_index.h:158-174); `keep(record)` restricts the index to a shard's entries."""
    canon, flip = _kmers_with_flip(rs, r, k)
    rep = set(ex.repetitive.tolist())
    cur, ext, eid = [], [], []
    idx = np.searchsorted(ex.keys, canon)
    for p in range(len(canon)):
        i = int(idx[p])
        if int(canon[p]) in rep or i >= len(ex.keys) or ex.keys[i] != canon[p]:
            continue
        for e in ex.entries[int(ex.key_off[i]):int(ex.key_off[i + 1])].tolist():
            rec, pos = e >> 32, e & 0xFFFFFFFF
            if keep is not None and not keep(rec):
                continue
            if flip[p]:
                pos = int(rs.length[rec >> 1]) - pos - k
                rec ^= 1
            if rec == 2 * r and pos == p:
                continue                    # the trivial match (overlap.cpp:188-190)
            cur.append(p); ext.append(pos); eid.append(rec)
    return np.array(cur, np.int64), np.array(ext, np.int64), np.array(eid, np.int64), flip


def _option_b_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    from flye_amd import config, dist, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    k = 17
    rs = synth.simulate(seed=5, genome_len=12_000, coverage=14, kind="hifi", n_repeat_families=3, n_tandems=8,
                        n_homopolymers=8).filter_min_len(1000)
    cfg = config.preset("corrected")
    o = O.Oracle(k, threads=1)
    o.set_reads(rs)
    o.build_index(cfg)
    full = o.export_index()
    # this rank's index shard = the entries whose TARGET read it owns; it probes ALL queries against it and sends
    # each query's hits to the query's owner (read i -> rank i % world), in an arbitrary order
    rng = np.random.default_rng(100 + rank)
    outbox = [dict() for _ in range(world)]
    for r in range(rs.n):
        cur, ext, eid, _ = _emit_hits(full, rs, r, k, keep=lambda rec: dist.owner_of_target(rec, world) == rank)
        perm = rng.permutation(len(cur))
        outbox[int(dist.owner_of(r, world))][r] = (cur[perm], ext[perm], eid[perm])
    inbox = [None] * world
    td.all_gather_object(inbox, outbox)                      # the all-to-all, as objects
    checked = ties = moved = 0
    for r in range(rank, rs.n, world):
        parts = [inbox[g][rank][r] for g in range(world)]
        cur = np.concatenate([p[0] for p in parts]); ext = np.concatenate([p[1] for p in parts])
        eid = np.concatenate([p[2] for p in parts])
        want_cur, want_ext, want_eid, flip = _emit_hits(full, rs, r, k)
        order = dist.option_b_receive_order(cur, ext, eid, flip, rs.length[(eid >> 1)], k)
        cur, ext, eid = cur[order], ext[order], eid[order]
        assert np.array_equal(cur, want_cur) and np.array_equal(ext, want_ext) and np.array_equal(eid, want_eid)
        # the unstable hit sort (overlap.cpp:201-204) then runs on identical input: identical permutation
        keys = (eid.astype(np.uint64) << np.uint64(32)) | cur.astype(np.uint64)
        sp = O.std_sort_perm(keys)
        assert np.array_equal(ext[sp], want_ext[O.std_sort_perm((want_eid.astype(np.uint64) << np.uint64(32)) | want_cur.astype(np.uint64))])
        # ... which an arrival-order input would not give wherever keys tie
        arrival = np.concatenate([p[1] for p in parts])
        akeys = np.concatenate([(p[2].astype(np.uint64) << np.uint64(32)) | p[0].astype(np.uint64) for p in parts])
        moved += int(not np.array_equal(arrival[O.std_sort_perm(akeys)], ext[sp]))
        ties += int(len(keys) - len(np.unique(keys)))
        checked += len(keys)
    open(os.path.join(out_dir, f"b_ok{rank}"), "w").write(repr((checked, ties, moved)))
    td.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_option_b_receiver_restores_emission_order_gloo(built, tmp_path, world):
    """SURVEY.md §8(e) option B: index sharded by target read, every rank probes all queries against its shard,
    hits travel to the query's owner, which re-orders them by (curPos, stored record, stored position) before the
    std::sort emulation.  Here with the oracle's index and numpy seed collection standing in for the device."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_option_b_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [eval(open(tmp_path / f"b_ok{r}").read()) for r in range(world)]
    assert sum(x[0] for x in res) > 5000        # hits checked
    assert sum(x[1] for x in res) > 0           # the case has tied (extId, curPos) keys ...
    assert sum(x[2] for x in res) > 0           # ... and arrival order would have changed the sorted result


def test_balanced_bin_ranges():
    from flye_amd import dist
    rng = np.random.default_rng(0)
    hist = rng.integers(0, 1000, size=4096)
    hist[:100] *= 50                                    # canonical k-mers crowd the low bins
    for world in (1, 2, 3, 8):
        r = dist.balanced_bin_ranges(hist, world)
        assert r[0][0] == 0 and r[-1][1] == 4096 and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        share = np.array([hist[a:b].sum() for a, b in r]) / hist.sum()
        assert share.max() < 1.0 / world + 0.05
    assert dist.balanced_bin_ranges(np.zeros(4096), 4)[-1][1] == 4096


# ---- the device path on two ranks (one GPU box: both ranks on device 0, collectives through gloo) ---------
def _gpu_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hashlib
    import torch.distributed as td
    from flye_amd import config, dist, gpu, synth
    from helpers import index_digest
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    for preset, kind in (("raw", "pb_raw"), ("hifi", "hifi")):
        rs = synth.simulate(seed=91, genome_len=60_000, coverage=25, kind=kind, n_tandems=30).filter_min_len(1000)
        cfg = config.preset(preset)
        ctx = gpu.Context(17, 0)
        ctx.set_reads(rs)
        vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
        st = dist.build_index_sharded(vi, cfg, rank, world, on_device=False)
        det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
        det.p.max_divergence = 0.3
        mine = dist.shard_queries(rs.n, rank, world)
        res = det.getSeqOverlapsBatch(mine)
        digest = index_digest(vi.export())
        # single-process reference on the same GPU: ordinary build, all queries
        ctx1 = gpu.Context(17, 0)
        ctx1.set_reads(rs)
        vi1 = gpu.VertexIndex(ctx1, float(int(cfg["assemble_kmer_sample"])))
        st1 = vi1.build(cfg)
        det1 = gpu.OverlapDetector.for_assemble(ctx1, vi1, cfg)
        det1.p.max_divergence = 0.3
        one = det1.getSeqOverlapsBatch(np.arange(0, 2 * rs.n, 2, dtype=np.uint32))
        assert digest == index_digest(vi1.export()), preset
        for f in ("selected_kmers", "index_entries", "repetitive_kmers", "repetitive_frequency"):
            assert st[f] == st1[f], (preset, f)
        assert np.float32(st["sample_rate"]).tobytes() == np.float32(st1["sample_rate"]).tobytes()
        lines1 = one.lines()
        want = [l for i in range(rank, rs.n, world) for l in lines1[int(one.query_off[i]):int(one.query_off[i + 1])]]
        assert res.lines() == want and len(want) > 0, preset
        out[preset] = (st["piece"], st["collective_bytes"], len(want))
        ctx.close(); ctx1.close()
    open(os.path.join(out_dir, f"gpu_ok{rank}"), "w").write(repr(out))
    td.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_sharded_build_and_overlaps_on_device(built, tmp_path):
    """Two processes, each with its own context on the GPU: key-range-sharded index build (begin / build_range
    / sums all-reduce / finish / pieces all-gather / import) and read-sharded overlap stage; index and
    overlaps must equal the single-process ones."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [eval(open(tmp_path / f"gpu_ok{k}").read()) for k in range(2)]
    for preset in ("raw", "hifi"):
        assert r[0][preset][0] != r[1][preset][0] and min(r[0][preset][0][1], r[1][preset][0][1]) > 0   # both built a real piece


@pytest.mark.gpu
def test_device_to_device_collectives_path_one_rank(built):
    """flye_amd/dist.py with on_device=True (what bench.py --gpus N selects with the nccl backend): RCCL all-reduce
    of the batch frequencies and the sums, broadcasts into the context's own full-size arrays, on device memory
    wrapped through __cuda_array_interface__.  RCCL refuses two ranks on one device, so a fresh child process runs
    ONE rank with every collective forced on (tools/sharded_build_check.py); index and overlaps must equal the
    ordinary build's."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"],
                          os.path.join(ROOT, "tools", "sharded_build_check.py")], capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("index identical True") == 2 and "False" not in out.stdout
