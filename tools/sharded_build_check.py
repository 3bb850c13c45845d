"""The device-to-device path of the sharded index build (flye_amd/dist.py, on_device=True: the batch frequencies, the
sums and the CSR pieces as torch tensors over __cuda_array_interface__, RCCL all-reduce / broadcast on them, the
gather straight into the context's arrays), on however many ranks the launcher gives -- one rank on a one-GPU box
exercises every call of that path (FLYE_FORCE_COLLECTIVES: the collectives run in a one-rank group too).
    python -m torch.distributed.run --nproc-per-node N tools/sharded_build_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import torch.distributed as td
from flye_amd import config, dist, gpu, synth
from helpers import index_digest

rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
os.environ["FLYE_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(local)
td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
for preset, kind in (("raw", "pb_raw"), ("hifi", "hifi")):
    rs = synth.simulate(seed=91, genome_len=200_000, coverage=25, kind=kind, n_tandems=30).filter_min_len(1000)
    cfg = config.preset(preset)
    ctx = gpu.Context(17, local); ctx.set_reads(rs)
    vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
    st = dist.build_index_sharded(vi, cfg, rank, world, on_device=True)
    ctx1 = gpu.Context(17, local); ctx1.set_reads(rs)
    vi1 = gpu.VertexIndex(ctx1, float(int(cfg["assemble_kmer_sample"])))
    st1 = vi1.build(cfg)
    same = index_digest(vi.export()) == index_digest(vi1.export())
    det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg); det1 = gpu.OverlapDetector.for_assemble(ctx1, vi1, cfg)
    q = dist.shard_queries(rs.n, rank, world)
    a, b = det.getSeqOverlapsBatch(q), det1.getSeqOverlapsBatch(q)
    print(f"rank {rank}/{world} {preset}: index identical {same}, sample rate bits equal "
          f"{np.float32(st['sample_rate']).tobytes() == np.float32(st1['sample_rate']).tobytes()}, overlaps identical "
          f"{a.recs.tobytes() == b.recs.tobytes()} ({len(a.recs)}), piece {st['piece']}, moved {st['collective_bytes']} B in "
          f"{st['allgather_s'] * 1e3:.1f} ms", flush=True)
    assert same and a.recs.tobytes() == b.recs.tobytes()
td.destroy_process_group()
