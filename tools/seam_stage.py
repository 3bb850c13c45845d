"""The reference's own program (oracle/ref_dumper.cpp driving Flye's VertexIndex / OverlapDetector through
processInParallel) on the bench workload, twice: as compiled from the reference alone (oracle/_ref/ref_dumper) and
linked with our seam definitions in front (oracle/_ref/ref_dumper_gpu, integration/flye_seam.cpp: index build and
getSeqOverlaps on the device, batched over its worker threads by fgb_*).  Same FASTA, same flags, same thread
count; the two overlap files must be byte-identical.  Prints the stage times of both.
    python tools/seam_stage.py [scale] [threads]"""
import filecmp, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flye_amd import config, workloads
from oracle import oracle as O

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
if not (O.have_ref() and O.have_ref_gpu()):
    raise SystemExit("oracle/_ref/ref_dumper and ref_dumper_gpu are needed (python -c 'import __graft_entry__ as g; g.build()' "
                     "where /root/reference is present)")
try:
    cpus = max(1, int(open("/sys/fs/cgroup/cpu.max").read().split()[0]) // 100000)
except Exception:
    cpus = os.cpu_count() or 8
threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(cpus, os.cpu_count() or cpus)
rs, min_ovlp, preset = workloads.ecoli_pb50(scale=scale)
print(f"E. coli PB 50x x{scale}: {rs.n} reads, {rs.total_bases / 1e6:.1f} Mbp; {threads} threads", flush=True)
with tempfile.TemporaryDirectory() as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    rs.write_fasta(fa)
    out = {}
    for name, binary in (("device seams", O.REF_DUMPER_GPU), ("reference alone", None)):
        ov = os.path.join(tmp, name.replace(" ", "_") + ".txt")
        t = time.perf_counter()
        info = O.run_ref(fa, params_string=config.params_string(preset), threads=threads, min_read_len=0,
                         min_overlap=config.DETECTOR_MIN_OVERLAP, ovlp_out=ov, binary=binary)
        out[name] = (info, ov, time.perf_counter() - t)
        print(f"{name:16s}: index {info['index_s']:.2f} s, overlaps of all {rs.n} reads {info['overlap_s']:.2f} s "
              f"({info['queried_bp'] / info['overlap_s'] / 1e9:.3f} Gbp/s), {info['overlaps']} records, process {out[name][2]:.1f} s", flush=True)
    same = filecmp.cmp(out["device seams"][1], out["reference alone"][1], shallow=False)
    a, b = out["device seams"][0], out["reference alone"][0]
    print(f"overlap files byte-identical: {same}; overlap stage {b['overlap_s'] / a['overlap_s']:.1f}x, index build {b['index_s'] / a['index_s']:.1f}x")
    sys.exit(0 if same else 1)
