"""Multi-GPU layout of the path (option A): one process per GPU.

* INDEX BUILD, sharded by key range (``build_index_sharded``).  ``balanced_bin_ranges`` cuts the 4096 key bins
  into ``world`` contiguous ranges of about equal weight; rank r holds the exact k-mer counters of ITS range only,
  sorts and run-length encodes only its range.  Collectives: an all-reduce of the k-mer frequencies per batch of
  reads (solid mode: each rank knows the counts of its own key range), an all-reduce of filterFrequentKmers' two
  integer sums (vertex_index.cpp:175-184 takes them over ALL keys), then an all-gather of the CSR pieces -- keys,
  list offsets, entries, repetitive keys, in rank order = key order -- straight into every rank's own full-size
  index arrays (``fg_index_gather_begin / _end``).  With the nccl backend (= RCCL over xGMI) everything travels
  device to device.
* OVERLAP STAGE: reads shard by sequence id, rank r owns the forward reads i with i % world == r and
  computes their lists against its full index copy: no data-path collective; only the barrier / max-time
  reduction of the bench.
"""
from __future__ import annotations

import os
import time

import numpy as np


def shard_queries(n_reads: int, rank: int, world: int, first_id: int = 0) -> np.ndarray:
    """FastaRecord ids (forward strand) owned by ``rank``."""
    idx = np.arange(rank, n_reads, world, dtype=np.int64)
    return (first_id + 2 * idx).astype(np.uint32)


def owner_of(read_index, world: int):
    return np.asarray(read_index) % world


def merge_sharded(per_rank_ids, per_rank_lists):
    """Reassemble per-read overlap lists in read order from per-rank results."""
    merged = {}
    for ids, lists in zip(per_rank_ids, per_rank_lists):
        for rid, lst in zip(ids, lists):
            merged[int(rid)] = lst
    return [merged[k] for k in sorted(merged)]


# ---- option B of SURVEY.md §8(e): index sharded by TARGET read, seed hits exchanged -----------------------------
# Not the layout bench.py runs (the replicated index fits 288 GB for every BASELINE config); what is
# kept here, tested on CPU (tests/test_dist.py), is the part of it that bit parity hangs on: the order in which the
# query's owner must line the received hits up before the std::sort emulation.
def owner_of_target(record, world: int):
    """the rank whose index shard holds the entries of this stored record (forward or reverse strand of read i)"""
    return (np.asarray(record) >> 1) % world


def option_b_receive_order(cur_pos, ext_pos, ext_id, flip_at_cur, ext_len, k: int):
    """Hits (curPos, extPos, extId) of ONE query gathered from the index shards in arbitrary order -> the
    permutation that restores the reference's emission order, i.e. the input order of its unstable hit sort
    (overlap.cpp:176-204): ascending curPos, and per query k-mer ascending STORED global position
    (vertex_index.cpp:108-114).  The stored (record, position) of a hit is recovered from what was reported:
    a flipped query k-mer reports (record ^ 1, len - pos - k) (vertex_index.h:158-174), and whether the k-mer at
    curPos was flipped is known to the receiver (``flip_at_cur[curPos]``); ``ext_len[i]`` = length of hit i's
    target.  (curPos, stored record, stored position) is a total order: no two hits of a query share it."""
    cur_pos = np.asarray(cur_pos, np.int64)
    ext_pos = np.asarray(ext_pos, np.int64)
    ext_id = np.asarray(ext_id, np.int64)
    fl = np.asarray(flip_at_cur, bool)[cur_pos]
    rec = np.where(fl, ext_id ^ 1, ext_id)
    pos = np.where(fl, np.asarray(ext_len, np.int64) - ext_pos - k, ext_pos)
    return np.lexsort((pos, rec, cur_pos))


# ---- sharded index build ------------------------------------------------------------------------------
def balanced_bin_ranges(hist, world: int):
    """``world`` contiguous bin ranges [lo, hi) covering all bins, each holding about 1/world of the accepted
    k-mer positions (cut where the running sum passes r/world of the total).  Identical on every rank: the
    histogram is."""
    h = np.asarray(hist, dtype=np.float64)
    n = len(h)
    cs = np.concatenate([[0.0], np.cumsum(h)])
    total = cs[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        c = int(np.searchsorted(cs, target, side="left"))
        cuts.append(min(n, max(cuts[-1], c)))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def concat_pieces(pieces):
    """CSR pieces (keys, key_off, entries, repetitive) of ascending key ranges -> one index: the offsets of
    piece r are shifted by the entries of the pieces before it.  Host form of what the all-gather assembles."""
    from .gpu import IndexExport
    keys = np.concatenate([p.keys for p in pieces])
    ent = np.concatenate([p.entries for p in pieces])
    rep = np.concatenate([p.repetitive for p in pieces])
    off = [np.zeros(1, np.uint64)]
    base = 0
    for p in pieces:
        off.append(p.key_off[1:].astype(np.uint64) + np.uint64(base))
        base += len(p.entries)
    return IndexExport(keys, np.concatenate(off), ent, rep)


def allgather_pieces(piece, rank: int, world: int, dev):
    """All-gather of the ranks' CSR pieces (torch int64 tensors keys[nk], key_off[nk + 1] relative to the
    piece, entries[ne], repetitive[nr]; ascending key ranges in rank order) into the full arrays on every
    rank.  Pieces differ in size: their sizes are exchanged first, then every rank broadcasts its slice of
    the assembled arrays (a ring all-gather's volume, no padding).  Returns (keys, key_off, entries,
    repetitive, (K, E, R), bytes moved)."""
    import torch
    import torch.distributed as td
    pk, po, pe, pr = piece
    nk, ne, nr = len(pk), len(pe), len(pr)
    sizes = torch.zeros((world, 3), dtype=torch.int64, device=dev)
    sizes[rank] = torch.tensor([nk, ne, nr], dtype=torch.int64, device=dev)
    td.all_reduce(sizes)
    sz = sizes.cpu().numpy()
    K, E, R = (int(x) for x in sz.sum(axis=0))
    kb = np.concatenate([[0], np.cumsum(sz[:, 0])]).astype(np.int64)
    eb = np.concatenate([[0], np.cumsum(sz[:, 1])]).astype(np.int64)
    rb = np.concatenate([[0], np.cumsum(sz[:, 2])]).astype(np.int64)
    keys = torch.empty(max(K, 1), dtype=torch.int64, device=dev)
    off = torch.empty(K + 1, dtype=torch.int64, device=dev)
    ent = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
    rep = torch.empty(max(R, 1), dtype=torch.int64, device=dev)
    keys[kb[rank]:kb[rank + 1]] = pk
    off[kb[rank]:kb[rank + 1]] = po[:nk] + int(eb[rank])     # list offsets shift by the entries of the pieces before
    ent[eb[rank]:eb[rank + 1]] = pe
    rep[rb[rank]:rb[rank + 1]] = pr
    off[K] = E
    moved = world * 24
    for r in range(world):
        for arr, b in ((keys, kb), (off, kb), (ent, eb), (rep, rb)):
            if b[r + 1] > b[r]:
                td.broadcast(arr[b[r]:b[r + 1]], src=r)
                moved += int(b[r + 1] - b[r]) * 8
    return keys, off, ent, rep, (K, E, R), moved


class _DevArr:
    """device memory as a ``__cuda_array_interface__`` object (torch wraps it without copying)"""

    def __init__(self, ptr: int, n: int, typestr: str = "<i8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _view(ptr: int, n: int, cuda_dev, typestr="<i8"):
    """torch tensor over n elements of device memory at ptr (no copy)"""
    import torch
    if n == 0 or not ptr:
        return torch.empty(0, dtype=torch.int64 if typestr == "<i8" else torch.int32, device=cuda_dev)
    return torch.as_tensor(_DevArr(ptr, n, typestr), device=cuda_dev)


def _all_reduce_dev(t, on_device: bool):
    """sum over the ranks, in place, of a device tensor: RCCL on the tensor itself, or (gloo rehearsal) through a
    host copy"""
    import torch
    import torch.distributed as td
    if on_device:
        td.all_reduce(t)
        # the library reads the array on ITS stream next (fg_index_batch_select): the collective has to be complete on
        # the host's clock, not just ordered on torch's stream
        torch.cuda.synchronize()
    else:
        h = t.cpu()
        td.all_reduce(h)
        t.copy_(h)


def _broadcast_dev(t, src: int, on_device: bool):
    import torch.distributed as td
    if t.numel() == 0:
        return
    if on_device:
        td.broadcast(t, src=src)
    else:
        h = t.cpu()
        td.broadcast(h, src=src)
        t.copy_(h)


def gather_pieces_inplace(vi, rank: int, world: int, on_device: bool, sample_rate_of, force=False):
    """All-gather of the ranks' CSR pieces (ascending key ranges in rank order) straight into the context's own
    full-size arrays (fg_index_gather_begin / _end): no second copy of the index.  Pieces differ in size: their sizes
    are exchanged first, then every rank broadcasts its slice (a ring all-gather's volume, no padding).
    ``sample_rate_of(E)`` gives VertexIndex::getSampleRate() for the whole index.  Returns ((K, E, R), bytes moved)."""
    import torch
    import torch.distributed as td
    cuda = torch.device("cuda", torch.cuda.current_device())
    red = cuda if on_device else torch.device("cpu")
    (nk, ne, nr), _ = vi.device_arrays()
    sizes = torch.zeros((world, 3), dtype=torch.int64, device=red)
    sizes[rank] = torch.tensor([nk, ne, nr], dtype=torch.int64, device=red)
    if world > 1 or force:
        td.all_reduce(sizes)
    sz = sizes.cpu().numpy()
    K, E, R = (int(x) for x in sz.sum(axis=0))
    kb = np.concatenate([[0], np.cumsum(sz[:, 0])]).astype(np.int64)
    eb = np.concatenate([[0], np.cumsum(sz[:, 1])]).astype(np.int64)
    rb = np.concatenate([[0], np.cumsum(sz[:, 2])]).astype(np.int64)
    full, piece, psz = vi.gather_begin(K, E, R)
    assert psz == [nk, ne, nr]
    keys, off, ent, rep = _view(full[0], K, cuda), _view(full[1], K + 1, cuda), _view(full[2], E, cuda), _view(full[3], R, cuda)
    pk, po, pe, pr = _view(piece[0], nk, cuda), _view(piece[1], nk + 1, cuda), _view(piece[2], ne, cuda), _view(piece[3], nr, cuda)
    if nk:
        keys[kb[rank]:kb[rank + 1]] = pk
        off[kb[rank]:kb[rank + 1]] = po[:nk] + int(eb[rank])     # list offsets shift by the entries of the pieces before
    if ne:
        ent[eb[rank]:eb[rank + 1]] = pe
    if nr:
        rep[rb[rank]:rb[rank + 1]] = pr
    off[K:K + 1] = E
    moved = world * 24
    if world > 1 or force:
        for r in range(world):
            for arr, b in ((keys, kb), (off, kb), (ent, eb), (rep, rb)):
                if b[r + 1] > b[r]:
                    _broadcast_dev(arr[b[r]:b[r + 1]], r, on_device)
                    moved += int(b[r + 1] - b[r]) * 8
    torch.cuda.synchronize()
    vi.gather_end(sample_rate_of(E))
    vi.stats = dict(vi.stats or {}, selected_kmers=K, index_entries=E, repetitive_kmers=R,
                    sample_rate=float(np.float32(sample_rate_of(E))))
    return (K, E, R), moved


def build_index_sharded(vi, cfg: dict, rank: int, world: int, on_device: bool):
    """The build main_assemble.cpp:195-223 selects, sharded over the ranks of the default process group
    (SURVEY.md §8e).  ``on_device``: collectives on device memory (nccl = RCCL over xGMI); otherwise through host
    copies (gloo rehearsal).

    * solid k-mers: the key bins are cut into ``world`` ranges holding equal numbers of k-mer positions
      (``fg_index_kmer_hist``, identical on every rank); rank r keeps the exact counters of ITS range only (an
      eighth of the 4^k array at 8 ranks) and counts those k-mers over all reads; then, batch of reads by batch
      (bounded scratch), every rank writes the frequencies it knows, an all-reduce makes the array complete, and
      every rank runs the per-read selection on it (replicated: it is cheap and leaves every rank with the same
      selection bits, so nothing else of the selection is exchanged);
    * minimizers: the selection needs no counters and no exchange; ranges are balanced on the accepted positions;
    * rank r sorts and run-length encodes its range (the same range its counters cover); all-reduce of
      filterFrequentKmers' two sums; finish; all-gather of the CSR pieces in place.
    Returns the index statistics plus what the collectives moved."""
    import torch
    import torch.distributed as td
    cuda = torch.device("cuda", torch.cuda.current_device())
    red = cuda if on_device else torch.device("cpu")
    # FLYE_FORCE_COLLECTIVES: run every collective also in a one-rank group (tools/sharded_build_check.py: one rank
    # on a one-GPU box then exercises every RCCL call of this path)
    coll = world > 1 or bool(os.environ.get("FLYE_FORCE_COLLECTIVES"))
    t0 = time.perf_counter()
    freq_bytes = 0
    distinct = 0
    if cfg["use_minimizers"]:
        hist = vi.begin(cfg)
        ranges = balanced_bin_ranges(hist, world)
    else:
        ranges = balanced_bin_ranges(vi.kmer_hist(), world)
        distinct, n_batches = vi.count_slice(cfg, *ranges[rank])
        for b in range(n_batches):
            ptr, n = vi.batch_freq(b)
            if coll and n:
                _all_reduce_dev(_view(ptr, n, cuda, "<i4"), on_device)
                freq_bytes += 4 * n
            vi.batch_select(b)
        vi.selection_done()
    sums = vi.build_range(*ranges[rank])
    t1 = time.perf_counter()
    tot = torch.tensor(np.concatenate([sums.astype(np.int64), [distinct]]), device=red)
    if coll:
        td.all_reduce(tot)
    tot = tot.cpu().numpy()
    st = dict(vi.finish(tot[:2].astype(np.uint64)))
    st["total_kmers"] = int(tot[2])
    (nk, ne, nr), _ = vi.device_arrays()
    t2 = time.perf_counter()
    total_bases = np.float32(vi.ctx.rs.total_bases)

    def sample_rate_of(E):
        # VertexIndex::getSampleRate(): the ctor value, or totalLen / totalEntries over the WHOLE index (vertex_index.cpp:480-482)
        if cfg["use_minimizers"]:
            return float(total_bases / np.float32(E)) if E else float("inf")
        return vi._sample_rate_init

    (K, E, R), moved = gather_pieces_inplace(vi, rank, world, on_device, sample_rate_of, force=coll)
    t3 = time.perf_counter()
    st.update(vi.stats)
    st.update(bin_range=ranges[rank], piece=(int(nk), int(ne), int(nr)),
              collective_bytes=moved + 24 + freq_bytes, freq_allreduce_bytes=freq_bytes,
              select_and_sort_s=t1 - t0, finish_s=t2 - t1, allgather_s=t3 - t2, import_s=0.0, build_seconds=t3 - t0)
    vi.stats = st
    return st
