#!/usr/bin/env python3
"""Headline benchmark: Gbp of reads overlapped per second (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the overlap stage (seed collection -> sort -> chaining ->
OverlapRange records on the host) over this rank's shard of the query reads, with
the 2-bit reads and the k-mer index already resident in HBM.  Workload at N = 1:
BASELINE.json configs[1], "E.coli PB 50x" (synthetic, SURVEY.md §8d generator:
4.64 Mb genome with planted repeats, 50x PacBio-raw-like reads, asm_raw_reads.cfg
parameters).  For N > 1 the genome grows with N (weak scaling): reads shard by
sequence id (flye_amd/dist.py), every rank holds the whole index, no data-path
collective (SURVEY.md §8e option A).

The JSON line also carries
  roofline     for the kernel that holds the device longest by itself (ranked in an untimed pass with the
               chaining stage serialised): algorithmic bytes (DESIGN.md "Algorithmic bytes") / its HIP-event
               time per launch in the timed region
  cpu_baseline the reference itself (Flye 2.8.1's own sources compiled into oracle/_ref/ref_dumper,
               kind "reference") on this host's cores: its index build over all reads, then its
               overlap stage on a bounded prefix of the same queries, records compared with the GPU's
  cpu_port     the CPU oracle (oracle/, a port of the reference algorithm) on the same queries
               against the same (device-built) index; it becomes cpu_baseline when the reference
               binary is absent.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes(kernel: str, bp: float, m: float, d: float, ovl_per_bp: float, d_small: float = None) -> float:
    """Per-launch algorithmic bytes of each overlap-stage kernel (SURVEY.md §8d:
    B = 16.25 + 44 m + 20 d B/bp, split by the kernel that owns each term).  k_chain_small is credited with the DP
    elements of the groups IT processed (d_small, counted on the device: fg_overlap_batch.dp_elements_small) and with
    none of the record bytes -- the other chaining kernels own the rest of the stage's bytes."""
    per_bp = {
        "k_probe": 0.25 + 16.0,                 # 2-bit read + one 16 B slot probe per k-mer
        "k_fill": m * (8.0 + 12.0),             # index entry read + hit write
        "k_sort_hits": m * 24.0,                # one read + one write of each 12 B hit
        "k_chain": d * (12.0 + 8.0) + 44.0 * ovl_per_bp,  # hit read + score/backptr write + record
    }
    base = kernel.split("<")[0]
    if base == "k_chain_small" and d_small is not None:
        return d_small * (12.0 + 8.0) * bp
    if base.startswith("k_chain"):
        base = "k_chain"
    if base.startswith("k_sort"):       # k_sort_level / k_sort_wide / k_sort_lds share the sort's bytes
        base = "k_sort_hits"
    return per_bp.get(base, 0.0) * bp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scale", type=float, default=None, help="genome scale (default: --gpus for weak scaling, 1 for strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: the genome (reads, index) grows with --gpus, per-rank queries stay ~fixed; strong: the "
                         "E. coli workload itself is split over the ranks")
    ap.add_argument("--cpu-sample-bp", type=float, default=250e6)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cpu-reference", action="store_true",
                    help="skip the compiled reference itself (oracle/_ref/ref_dumper: its own index build over all "
                         "reads, ~1 min on 16 cores, + getSeqOverlaps over a bounded prefix of the queries)")
    ap.add_argument("--cpu-reference-bp", type=float, default=250e6,
                    help="query prefix (bp) the reference's overlap stage is timed on")
    ap.add_argument("--cpu-reference-threads", type=int, default=0, help="0 = the CPUs this process may use")
    ap.add_argument("--no-assemble-stage", action="store_true",
                    help="skip the reference's whole `flye-modules assemble` stage (pure, and with the device seams)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    # rehearsal on a one-GPU box: FLYE_BENCH_BACKEND=gloo puts every rank on the visible GPUs
    # round-robin and reduces through CPU tensors (RCCL refuses two ranks on one device)
    backend = os.environ.get("FLYE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    import torch.distributed as td
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            td.init_process_group(backend, rank=rank, world_size=world)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    from flye_amd import config, dist, gpu, workloads

    scale = args.scale if args.scale is not None else (float(world) if args.scaling == "weak" else 1.0)
    t0 = time.time()
    rs, min_ovlp, preset = workloads.ecoli_pb50(seed=12345, scale=scale)
    cfg = config.preset(preset)
    t_gen = time.time() - t0

    ctx = gpu.Context(int(cfg["kmer_size"]), local)
    t0 = time.time()
    ctx.set_reads(rs)
    t_upload = time.time() - t0
    vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
    t0 = time.perf_counter()
    if world > 1:
        # index build sharded by key range over the ranks: k-mer counters of the rank's own key range, frequency
        # all-reduce per batch of reads, sums all-reduce, all-gather of the CSR pieces in place (RCCL over xGMI with
        # the nccl backend), every rank then holds the whole index (flye_amd/dist.py)
        st = dist.build_index_sharded(vi, cfg, rank, world, on_device=(backend == "nccl"))
    else:
        st = vi.build(cfg)
    t_index_wall = time.perf_counter() - t0
    # --min-ovlp of the pipeline driver (N90 rule) only filters reads by length (main_assemble.cpp:183, done in
    # workloads.ecoli_pb50) and feeds Extender; the DETECTOR always runs with minimumOverlap = 1000
    # (main_assemble.cpp:174 overrides the parameter before the detector is built at :229-238)
    det_min_ovlp = config.DETECTOR_MIN_OVERLAP
    det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg, min_overlap=det_min_ovlp)
    queries = dist.shard_queries(rs.n, rank, world)
    my_bp = int(rs.length[(queries // 2).astype(np.int64)].sum())

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    def step():
        return det.getSeqOverlapsBatch(queries)

    for _ in range(args.warmup):
        res = step()
    # One more untimed pass with the chaining stage's size classes one after the other on the main stream: in the
    # timed passes they run side by side on three streams, where an event-bracketed duration includes the time a
    # kernel shares the chip with the others.  The serial pass says which kernel holds the device longest by itself.
    serial_ktimes = {}
    if rank == 0 and not os.environ.get("FLYE_BENCH_NO_SERIAL_PASS"):     # (the counter passes of tools/profile_round.sh want ONE pass)
        os.environ["FG_CHAIN_STREAMS"] = "1"
        step()
        serial_ktimes = {k_: v_ for k_, v_ in ctx.kernel_times().items() if not k_.startswith("host:")}
        del os.environ["FG_CHAIN_STREAMS"]
    barrier()
    t0 = time.perf_counter()
    ktimes = {}
    dev_s = 0.0
    for _ in range(args.steps):
        res = step()
        dev_s += res.device_seconds
        for name, (sec, n) in ctx.kernel_times().items():
            a = ktimes.setdefault(name, [0.0, 0])
            a[0] += sec
            a[1] += n
    barrier()
    elapsed = time.perf_counter() - t0

    tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    bp = torch.tensor([my_bp], dtype=torch.float64, device=red_dev)
    if world > 1:
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        td.all_reduce(bp, op=td.ReduceOp.SUM)
    elapsed = float(tt.item())
    total_bp = float(bp.item())

    if rank == 0:
        value = total_bp * args.steps / elapsed / 1e9
        m = res.seed_hits / max(1, res.query_bp)
        d = res.dp_elements / max(1, res.query_bp)
        ovl = len(res.recs) / max(1, res.query_bp)
        # dominant kernel of the timed region (HIP events on the library stream)
        # (k_chain_dp and the sum of the ~28 k_sort_level launches are within a few per cent of each other:
        # within 5 % of the maximum the kernel with the fewest launches is taken, so that the line does not
        # flip between two kernels from run to run)
        # dominance by the serial pass's exclusive times; duration and launch count from the timed region
        if not serial_ktimes:
            serial_ktimes = {k_: v_ for k_, v_ in ktimes.items() if not k_.startswith("host:")}
        top = max(v_[0] for v_ in serial_ktimes.values())
        dom_name = min((kv for kv in serial_ktimes.items() if kv[1][0] >= 0.95 * top), key=lambda kv: (kv[1][1], -kv[1][0]))[0]
        dom_sec, dom_n = ktimes[dom_name]
        launches_per_step = max(1, dom_n // args.steps)
        avg_launch_s = dom_sec / max(1, dom_n)
        d_small = res.dp_elements_small / max(1, res.query_bp)
        alg = algorithmic_bytes(dom_name, res.query_bp, m, d, ovl, d_small) / launches_per_step
        achieved = alg / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # PMC traffic (FETCH_SIZE + WRITE_SIZE passes of tools/pmc_traffic.py) cannot be collected inside this
        # process; the committed figure is used only while it was measured on THESE kernel sources and THIS
        # detector configuration (stamp written by tools/pmc_traffic.py), otherwise traffic is null
        traffic, traffic_note = None, "profiles/hbm_traffic.json absent"
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof):
            try:
                tj = json.load(open(prof))
                stamp = tj.get("_stamp", {})
                if stamp.get("source_sha256") != kernel_source_digest():
                    traffic_note = "stale: kernel sources changed since the PMC passes"
                elif stamp.get("min_overlap") != det_min_ovlp:
                    traffic_note = "stale: PMC passes ran another detector configuration"
                else:
                    traffic = tj.get(dom_name, {}).get("bytes_per_launch")
                    traffic_note = "rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE passes of this source state (profiles/)"
            except Exception as e:  # noqa: BLE001
                traffic_note = f"unreadable: {e}"
        stages = stage_rooflines(serial_ktimes, res.query_bp, m, d, ovl, prof if os.path.exists(prof) and traffic_note.startswith("rocprofv3") else None)
        dom_stage = max(stages.items(), key=lambda kv: kv[1]["exclusive_ms"])[0] if stages else None
        line = {
            "metric": "Gbp reads overlapped/sec", "value": round(value, 6), "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "E.coli PB 50x (synthetic, raw-read cfg): overlap stage, index resident",
                       "genome_bp": int(4_640_000 * scale), "reads": rs.n, "read_bp": rs.total_bases,
                       "queries_per_rank": int(len(queries)), "min_overlap": det_min_ovlp, "min_read_len": min_ovlp,
                       "kmer": int(cfg["kmer_size"]),
                       "sharding": (f"queries: reads by id over {world} ranks, no data-path collective; index: built sharded by "
                                    f"key range (counters, sort) + all-gather, then resident on every rank") if world > 1
                                   else "one rank: all queries, index built and resident on this GPU"},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": int(alg), "avg_launch_ms": round(avg_launch_s * 1e3, 4),
                         "launch": ("one level of the hit sort = the k_sort_level launch together with the k_sort_wide launch "
                                    "for pieces above 16 k hits (levels 0-7 only) and the k_sort_route launch behind it, "
                                    "bracketed by one pair of HIP events") if dom_name == "k_sort_level" else "one kernel launch",
                         "dominant_by": "exclusive device time in an untimed pass with the chaining classes serialised "
                                        "(FG_CHAIN_STREAMS=1); ms per pass there: " +
                                        ", ".join(f"{k_} {v_[0] * 1e3:.1f}" for k_, v_ in
                                                  sorted(serial_ktimes.items(), key=lambda kv: -kv[1][0])[:6])},
            # whole stages (all launches of all their kernels against the stage's algorithmic bytes), so that a fraction
            # cannot be moved by where a stage is cut into kernels; times = exclusive device time of the serialised pass
            "roofline_stage": (dict(stages[dom_stage], stage=dom_stage, bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s") if dom_stage else None),
            "stages": stages,
            "work": {"seed_hits_per_bp": round(m, 4), "dp_elements_per_bp": round(d, 4),
                     "dp_elements_small_per_bp": round(d_small, 4),
                     "overlaps": int(len(res.recs)), "device_ms_per_step": round(dev_s / args.steps * 1e3, 3),
                     "kernel_ms_per_step": {k: round(v[0] / args.steps * 1e3, 3) for k, v in
                                            sorted(ktimes.items(), key=lambda kv: -kv[1][0])},
                     "kernel_ms_note": "event-bracketed; k_group_prep, k_chain_dp and k_chain_finish run on three streams "
                                       "side by side, so their sums exceed the wall time they occupy",
                     "index_build_s": round(st["build_seconds"], 4), "index_build_wall_s": round(t_index_wall, 4),
                     # build + ONE pass over all queries, reads already uploaded (SURVEY §8d "end-to-end figure")
                     "end_to_end_gbps": round(total_bp / (t_index_wall + elapsed / args.steps) / 1e9, 4),
                     "index_build_collectives": ({"bytes": int(st["collective_bytes"]), "allgather_s": round(st["allgather_s"], 4),
                                                  "piece_of_rank0": list(st["piece"])} if world > 1 else None),
                     "read_gen_s": round(t_gen, 2),
                     "upload_s": round(t_upload, 3), "index_entries": int(st["index_entries"])},
        }
        if world == 1 and not args.no_cpu:
            port = cpu_port(rs, cfg, vi, queries, args.cpu_sample_bp, det_min_ovlp)
            # parity on the timed workload itself: the sample's records must match
            port["sample_records_identical"] = bool(port.pop("_same")(res))
            ref = None
            if not args.no_cpu_reference:
                ref = cpu_reference(rs, preset, det_min_ovlp, queries, res,
                                    args.cpu_reference_threads or effective_cpus(), args.cpu_reference_bp)
                if ref is not None and "error" not in ref and not args.no_assemble_stage:
                    # (the program builds its own index on this GPU beside the bench's: both fit)
                    ref["reference_program_with_device_seams_assemble"] = assemble_stage(
                        rs, preset, min_ovlp, args.cpu_reference_threads or effective_cpus())
            if ref is not None and "error" not in ref:
                # the reference itself (Flye's own code on this host's cores) is THE cpu baseline;
                # the port (oracle/) is reported beside it
                line["cpu_baseline"] = ref
                line["cpu_reference"] = ref
                line["cpu_port"] = port
            else:
                line["cpu_baseline"] = port
                if ref is not None:
                    line["cpu_reference"] = ref
        print(json.dumps(line), flush=True)
    if world > 1:
        td.destroy_process_group()


STAGES = {
    # stage -> (kernel name prefixes, algorithmic bytes per bp as a function of m, d, overlaps/bp: SURVEY.md §8d)
    "seed_collect": (("k_probe", "k_fill", "k_exscan"), lambda m, d, o: 16.25 + 20.0 * m),
    "hit_sort": (("k_sort",), lambda m, d, o: 24.0 * m),
    "chain": (("k_group", "k_chain", "k_prim"), lambda m, d, o: 20.0 * d + 44.0 * o),
}


def stage_rooflines(serial_ktimes, bp, m, d, ovl, traffic_file):
    """Per stage of the pass: algorithmic GB, exclusive device ms (serialised pass), achieved GB/s and fraction of
    the 8 TB/s HBM roofline; counted traffic (FETCH_SIZE + WRITE_SIZE of every kernel of the stage, profiles/) and
    its ratio to the algorithmic bytes when the committed counters belong to this source state."""
    tj = json.load(open(traffic_file)) if traffic_file else {}
    out = {}
    for name, (prefixes, per_bp) in STAGES.items():
        ks = [k for k in serial_ktimes if k.split("<")[0].startswith(prefixes)]
        sec = sum(serial_ktimes[k][0] for k in ks)
        if sec <= 0:
            continue
        alg = per_bp(m, d, ovl) * bp
        ent = {"kernels": sorted(ks), "algorithmic_bytes": int(alg), "exclusive_ms": round(sec * 1e3, 3),
               "achieved": round(alg / sec / 1e9, 2), "frac": round(alg / sec / 1e9 / HBM_PEAK_GBS, 5)}
        counted = [tj[k]["bytes_per_pass"] for k in tj if k.split("<")[0].startswith(prefixes) and isinstance(tj[k], dict)
                   and "bytes_per_pass" in tj[k]]
        if counted:
            ent["traffic"] = int(sum(counted))
            ent["traffic_ratio"] = round(sum(counted) / alg, 2)
        out[name] = ent
    return out


def kernel_source_digest() -> str:
    """sha256 over the device sources: what profiles/hbm_traffic.json is stamped with."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "flye_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def effective_cpus() -> int:
    """CPUs this process may really use: affinity mask, capped by the cgroup CPU quota
    (the GPU box shows 256 hardware threads but grants 16 CPUs' worth of time)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_port(rs, cfg, vi, queries, sample_bp, min_ovlp):
    """The CPU oracle on this host's cores over a bounded sample of the same queries
    against the same index (imported from the device, so the CPU does not spend
    minutes rebuilding it).  Checker/baseline only -- never the measured product."""
    from oracle import oracle as O
    cores = effective_cpus()
    o = O.Oracle(int(cfg["kmer_size"]), threads=cores)
    o.set_reads(rs)
    ex = vi.export()
    o.import_index(O.IndexExport(ex.keys, ex.key_off, ex.entries, ex.repetitive), vi.getSampleRate())
    lens = rs.length[(queries // 2).astype(np.int64)]
    n = int(np.searchsorted(np.cumsum(lens), sample_bp)) + 1
    n = max(1, min(n, len(queries)))
    sample = queries[:n]
    p = O.detector_params(cfg, min_overlap=min_ovlp)
    t0 = time.perf_counter()
    ores = o.overlaps(p, sample, threads=cores)
    dt = time.perf_counter() - t0

    def same(gres):
        end = int(gres.query_off[n])
        g = gres.recs[:end]
        return (len(g) == len(ores.recs) and
                all(np.array_equal(g[f], ores.recs[f]) for f in
                    ("cur_id", "ext_id", "cur_begin", "cur_end", "ext_begin", "ext_end", "score")) and
                np.array_equal(g["seq_divergence"].view(np.uint32), ores.recs["seq_divergence"].view(np.uint32)))

    return {"value": round(ores.query_bp / dt / 1e9, 6), "unit": "Gbp/s", "cores": cores, "kind": "port",
            "sample": f"first {n} query reads ({ores.query_bp} bp) of the same workload, same index, "
                      f"{dt:.2f} s wall", "_same": same}


def cpu_reference(rs, preset, min_ovlp, queries, gres, threads, sample_bp=250e6):
    """The reference itself (Flye 2.8.1 sources compiled by oracle/Makefile, unmodified) on this
    host: reads written as FASTA, its own countKmers + buildIndexUnevenCoverage over ALL reads,
    then getSeqOverlaps for a prefix of the forward reads through its processInParallel; every
    record of the prefix is compared with the GPU's (floats by bit pattern).
    Checker/baseline only."""
    import tempfile
    from flye_amd import config
    from oracle import oracle as O
    if not O.have_ref():
        return {"error": "oracle/_ref/ref_dumper not built"}
    lens = rs.length[(queries // 2).astype(np.int64)]
    n = max(1, min(int(np.searchsorted(np.cumsum(lens), sample_bp)) + 1, len(queries)))
    if not np.array_equal(queries[:n], np.arange(0, 2 * n, 2)):
        return {"error": "reference run needs the rank-0, N=1 query order"}
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "reads.fasta")
        ov = os.path.join(tmp, "ovlp.txt")
        rs.write_fasta(fa)
        t0 = time.perf_counter()
        info = O.run_ref(fa, params_string=config.params_string(preset), threads=threads, min_read_len=0,
                         min_overlap=min_ovlp, query_limit=n, ovlp_out=ov)
        wall = time.perf_counter() - t0
        same = records_equal_ref_file(gres.recs[:int(gres.query_off[n])], ov)
        # the same program linked with our seam definitions in front (integration/flye_seam.cpp: index build and
        # getSeqOverlaps on the device, batched over its worker threads): what the kernels buy the reference's own
        # stage, host side included.  Same FASTA, flags and threads; the overlap file must be byte-identical.
        seam = None
        if O.have_ref_gpu():
            import filecmp
            ov2 = os.path.join(tmp, "ovlp_seam.txt")
            try:
                t1 = time.perf_counter()
                i2 = O.run_ref(fa, params_string=config.params_string(preset), threads=threads, min_read_len=0,
                               min_overlap=min_ovlp, query_limit=n, ovlp_out=ov2, binary=O.REF_DUMPER_GPU)
                seam = {"overlap_s": i2["overlap_s"], "index_s": i2["index_s"], "process_s": round(time.perf_counter() - t1, 2),
                        "value": round(i2["queried_bp"] / i2["overlap_s"] / 1e9, 6), "unit": "Gbp/s",
                        "output_byte_identical_to_reference": filecmp.cmp(ov, ov2, shallow=False),
                        "what": "the reference's own program (oracle/_ref/ref_dumper_gpu = the same driver and Flye sources "
                                "with VertexIndex builds and OverlapDetector::getSeqOverlaps replaced at link time), "
                                f"{threads} worker threads, one getSeqOverlaps call each in flight"}
            except Exception as e:  # noqa: BLE001
                seam = {"error": str(e)[:200]}
    return {"value": round(info["queried_bp"] / info["overlap_s"] / 1e9, 6), "unit": "Gbp/s", "cores": threads,
            # its own index build + its overlap stage over the same reads: what work.end_to_end_gbps stands beside
            "end_to_end_gbps": (round(info["queried_bp"] / (info["index_s"] + info["overlap_s"]) / 1e9, 6) if n == len(queries) else None),
            "reference_program_with_device_seams": seam,
            "kind": "reference", "overlap_s": info["overlap_s"], "index_s": info["index_s"], "load_s": info["load_s"],
            "overlaps": info["overlaps"],
            "sample": f"first {n} forward reads ({info['queried_bp']} bp) of the same workload through the reference's "
                      f"processInParallel on {threads} threads; index built by the reference over all reads; "
                      f"{wall:.1f} s wall in total", "gpu_records_identical_to_reference": same}


def assemble_stage(rs, preset, min_ovlp, threads):
    """The reference's whole assemble stage (`flye-modules assemble`, main_assemble.cpp:123-257: reads -> index ->
    estimateOverlaperParameters -> Extender::assembleDisjointigs with ChimeraDetector -> ConsensusGenerator ->
    draft_assembly.fasta) on the bench reads, twice: compiled from the reference alone (oracle/_ref/flye_assemble)
    and with the device seams linked in front (oracle/_ref/flye_assemble_gpu).  Same FASTA, cfg, --min-ovlp and
    thread count.  With more than one thread the disjointigs depend on thread timing in the reference itself
    (extender.cpp:269), so the two drafts are compared by size only; byte identity at one thread is
    tests/test_seam.py's."""
    import tempfile
    from flye_amd import config
    from oracle import oracle as O
    if not (O.have_assemble() and O.have_assemble_gpu()):
        return {"error": "oracle/_ref/flye_assemble(_gpu) not built"}
    out = {"threads": threads, "min_ovlp": int(min_ovlp)}
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "reads.fasta")
        rs.write_fasta(fa)
        cfgp = config.write_cfg(os.path.join(tmp, "asm.cfg"), preset)
        for name, binary in (("device_seams", O.FLYE_ASSEMBLE_GPU), ("reference", O.FLYE_ASSEMBLE)):
            draft = os.path.join(tmp, name + ".fasta")
            try:
                info = O.run_assemble(fa, cfgp, draft, threads=threads, min_ovlp=min_ovlp, binary=binary)
                data = open(draft, "rb").read()
                out[name] = {"wall_s": round(info["wall_s"], 2), "extend_s_log": info.get("extend_s_log"),
                             "draft_bytes": len(data), "disjointigs": data.count(b">")}
                if "seams" in info and name == "device_seams":
                    out[name]["seams"] = info["seams"]
            except Exception as e:  # noqa: BLE001
                out[name] = {"error": str(e)[:300]}
    return out


def records_equal_ref_file(recs, path) -> bool:
    """GPU records against ref_dumper's --ovlp-out text (one OverlapRange per line, the float as
    its bit pattern in hex), column by column."""
    import pandas as pd
    names = ["cur_id", "cur_begin", "cur_end", "cur_len", "ext_id", "ext_begin", "ext_end", "ext_len", "score", "div"]
    df = pd.read_csv(path, sep=" ", comment="#", header=None, names=names, dtype={"div": str})
    if len(df) != len(recs):
        return False
    for f in names[:-1]:
        if not np.array_equal(df[f].to_numpy(np.int64), recs[f].astype(np.int64)):
            return False
    bits = np.array([int(x, 16) for x in df["div"]], dtype=np.uint32) if len(df) else np.empty(0, np.uint32)
    return bool(np.array_equal(bits, recs["seq_divergence"].view(np.uint32)))


if __name__ == "__main__":
    main()
