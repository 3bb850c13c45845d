"""A named workload (tools/run_workload.py's: dmel_ont30, hifi30, ecoli_pb50) through the COMPILED REFERENCE on the
host beside the device: the reference builds ITS index over all reads (countKmers + buildIndex*, the real code) and
runs getSeqOverlaps for the first N forward reads; every record of that prefix is compared with the device's (floats by
bit pattern).  Pins index and overlaps of the larger configurations against the reference itself, where the test
suite compares with the oracle on a device-built index.  Checker only (oracle/_ref).
    python tools/reference_check.py dmel_ont30 0.25 2000 [threads]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from flye_amd import config, gpu, workloads
from oracle import oracle as O

name = sys.argv[1]
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n_q = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
threads = int(sys.argv[4]) if len(sys.argv) > 4 else bench.effective_cpus()
if name == "ecoli_pb50":
    rs, min_ovlp, preset = workloads.ecoli_pb50(scale=scale)
elif name == "dmel_ont30":
    rs, min_ovlp, preset = workloads.dmel_ont30(scale=scale)
elif name == "hifi30":
    rs, min_ovlp, preset = workloads.hifi30(genome_len=int(4_640_000 * scale))
else:
    raise SystemExit("unknown workload")
if not O.have_ref():
    raise SystemExit("oracle/_ref/ref_dumper not built")
cfg = config.preset(preset)
print(f"{name} x {scale}: {rs.n} reads, {rs.total_bases / 1e6:.1f} Mbp", flush=True)
ctx = gpu.Context(int(cfg["kmer_size"]), 0)
ctx.set_reads(rs)
vi = gpu.VertexIndex(ctx, float(int(cfg["assemble_kmer_sample"])))
t0 = time.perf_counter(); st = vi.build(cfg); t_build = time.perf_counter() - t0
det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg, min_overlap=config.DETECTOR_MIN_OVERLAP)
n_q = min(n_q, rs.n)
q = np.arange(0, 2 * n_q, 2, dtype=np.uint32)
t0 = time.perf_counter(); res = det.getSeqOverlapsBatch(q); t_dev = time.perf_counter() - t0
print(f"device: index {t_build:.2f} s ({int(st['index_entries'])} entries), {n_q} queries in {t_dev * 1e3:.0f} ms, {len(res.recs)} records", flush=True)
with tempfile.TemporaryDirectory() as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    ov = os.path.join(tmp, "ovlp.txt")
    rs.write_fasta(fa)
    t0 = time.perf_counter()
    # the reference's index build says nothing for minutes: a line a minute, so that the box's watchdog sees a live run
    import threading
    done = threading.Event()
    def heartbeat():
        while not done.wait(60.0):
            print(f"  reference running, {time.perf_counter() - t0:.0f} s", flush=True)
    threading.Thread(target=heartbeat, daemon=True).start()
    info = O.run_ref(fa, params_string=config.params_string(preset), threads=threads, min_read_len=0,
                     min_overlap=config.DETECTOR_MIN_OVERLAP, query_limit=n_q, ovlp_out=ov)
    wall = time.perf_counter() - t0
    done.set()
    same = bench.records_equal_ref_file(res.recs, ov)
out = {"workload": name, "scale": scale, "reads": int(rs.n), "read_bp": int(rs.total_bases), "queries": int(n_q),
       "device_index_entries": int(st["index_entries"]), "device_records": int(len(res.recs)),
       "reference_overlaps": int(info["overlaps"]), "reference_index_s": info["index_s"], "reference_overlap_s": info["overlap_s"],
       "reference_threads": threads, "reference_wall_s": round(wall, 1), "device_index_s": round(t_build, 3),
       "records_identical_to_reference": bool(same)}
print(json.dumps(out), flush=True)
sys.exit(0 if same else 1)
