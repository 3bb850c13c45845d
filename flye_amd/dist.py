"""Multi-GPU layout of the path (option A): one process per GPU.

* INDEX BUILD, sharded by key range.  Every rank runs the k-mer selection over all reads (it needs every
  read: exact counts, per-read thresholds / minimizers) and gets the same histogram of accepted positions
  per key bin; ``balanced_bin_ranges`` cuts the bins into ``world`` contiguous ranges of about equal size;
  rank r sorts and run-length encodes only its range (``fg_index_build_range``).  Two collectives:
  an all-reduce of filterFrequentKmers' two integer sums (vertex_index.cpp:175-184 takes them over ALL
  keys), then an all-gather of the CSR pieces -- keys, list offsets, entries, repetitive keys, in rank
  order = key order -- after which every rank imports the concatenation (``fg_import_index``) and holds
  the full index.  With the nccl backend (= RCCL over xGMI) the pieces travel device to device.
* OVERLAP STAGE: reads shard by sequence id, rank r owns the forward reads i with i % world == r and
  computes their lists against its full index copy: no data-path collective; only the barrier / max-time
  reduction of the bench.
"""
from __future__ import annotations

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

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}


def build_index_sharded(vi, cfg: dict, rank: int, world: int, on_device: bool):
    """The build main_assemble.cpp:195-223 selects, sharded over the ranks of the default process group.
    ``on_device``: collectives on device tensors (nccl = RCCL); otherwise through host tensors (gloo).
    Returns the index statistics plus what the collectives moved."""
    import torch
    import torch.distributed as td
    dev = torch.device("cuda", torch.cuda.current_device()) if on_device else torch.device("cpu")
    t0 = time.perf_counter()
    hist = vi.begin(cfg)
    ranges = balanced_bin_ranges(hist, world)
    sums = vi.build_range(*ranges[rank])
    t1 = time.perf_counter()
    tot = torch.tensor(sums.astype(np.int64), device=dev)
    td.all_reduce(tot)
    st = dict(vi.finish(tot.cpu().numpy().astype(np.uint64)))
    t2 = time.perf_counter()

    (nk, ne, nr), ptrs = vi.device_arrays()
    if on_device:
        def view(ptr, n):
            return torch.as_tensor(_DevArr(ptr, n), device=dev) if n else torch.empty(0, dtype=torch.int64, device=dev)
        piece = (view(ptrs[0], nk), view(ptrs[1], nk + 1), view(ptrs[2], ne), view(ptrs[3], nr))
    else:
        ex = vi.export()
        piece = tuple(torch.from_numpy(np.ascontiguousarray(a).view(np.int64))
                      for a in (ex.keys, ex.key_off, ex.entries, ex.repetitive))
    keys, off, ent, rep, (K, E, R), moved = allgather_pieces(piece, rank, world, dev)
    if on_device:
        torch.cuda.synchronize()
    t3 = time.perf_counter()
    # VertexIndex::getSampleRate(): the ctor value, or totalLen / totalEntries over the WHOLE index (vertex_index.cpp:480-482)
    if cfg["use_minimizers"]:
        sample_rate = float(np.float32(vi.ctx.rs.total_bases) / np.float32(E)) if E else float("inf")
    else:
        sample_rate = vi._sample_rate_init
    if on_device:
        vi.import_index((K, E, R), sample_rate, on_device=True,
                        ptrs=(keys.data_ptr(), off.data_ptr(), ent.data_ptr(), rep.data_ptr()))
    else:
        from .gpu import IndexExport
        vi.import_index(IndexExport(keys[:K].numpy().view(np.uint64), off.numpy().view(np.uint64),
                                    ent[:E].numpy().view(np.uint64), rep[:R].numpy().view(np.uint64)), sample_rate)
    t4 = time.perf_counter()
    st.update(selected_kmers=K, index_entries=E, repetitive_kmers=R, sample_rate=float(np.float32(sample_rate)),
              bin_range=ranges[rank], piece=(int(nk), int(ne), int(nr)),
              collective_bytes=moved + 16, select_and_sort_s=t1 - t0, finish_s=t2 - t1,
              allgather_s=t3 - t2, import_s=t4 - t3, build_seconds=t4 - t0)
    vi.stats = st
    return st
