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
