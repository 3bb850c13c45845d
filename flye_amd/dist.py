"""Multi-GPU layout of the overlap stage (option A).

Reads shard by sequence id: rank r owns the forward reads i with i % world == r and
computes their overlap lists against a full copy of the index resident in its own
HBM, so the data path needs no collective; only the barrier / max-time reduction of
the bench and an optional gather of result counts go through torch.distributed
(RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

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
