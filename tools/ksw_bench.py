"""Throughput of fg_align_cigar_ksw (getAlignmentCigarKsw on the device) on a batch of read-sized pairs, next to the
reference's own ksw2 on this host's cores (oracle/_ref/ref_dumper --ksw-pairs, one thread)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from flye_amd import gpu
from helpers import edit_pair
from oracle import oracle as O
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
rng = np.random.default_rng(5)
pairs = [edit_pair(dict(seed=int(rng.integers(1, 1 << 30)), n=int(length * rng.uniform(0.6, 1.4)), err=float(rng.choice([0.01, 0.03, 0.1])),
                        hp=20)) for _ in range(n_pairs)]
bp = sum(len(a) for a, _ in pairs)
ctx = gpu.Context(17, 0)
ctx.align_cigar_ksw(pairs[:8])
got = ctx.align_cigar_ksw(pairs); dt = ctx.last_align_seconds
kt = ctx.kernel_times()
dev = sum(v[0] for k, v in kt.items() if k.startswith("k_ksw_extz2"))
print(f"{n_pairs} pairs, {bp / 1e6:.1f} Mbp of target: {dt * 1e3:.0f} ms in fg_align_cigar_ksw, kernels {dev * 1e3:.0f} ms -> {n_pairs / dt:.0f} alignments/s, "
      f"{bp / dev / 1e9:.3f} Gbp/s in the kernel")
if os.environ.get("KSW_BENCH_PHASES"):
    # where the kernel's time goes: runs with one phase switched off (their records are wrong and thrown away)
    for dbg, what in ((1, "without the DP"), (2, "without the backtrack"), (4, "backtrack one cell at a time")):
        os.environ["FG_KSW_DEBUG"] = str(dbg)
        ctx.align_cigar_ksw(pairs)
        k = sum(v[0] for kk, v in ctx.kernel_times().items() if kk.startswith("k_ksw_extz2"))
        print(f"  {what}: kernels {k * 1e3:.1f} ms")
    del os.environ["FG_KSW_DEBUG"]
if os.environ.get("KSW_BENCH_ONLY"):
    sys.exit(0)
# the same batch with the state in memory (the literal emulation of the vector code): records must be identical
os.environ["FG_KSW_LITERAL"] = "1"
lit = ctx.align_cigar_ksw(pairs); dl = ctx.last_align_seconds
del os.environ["FG_KSW_LITERAL"]
kl = sum(v[0] for k, v in ctx.kernel_times().items() if k.startswith("k_ksw_extz2"))
print(f"state in memory instead of LDS rings: kernel {kl * 1e3:.0f} ms, call {dl * 1e3:.0f} ms; records identical: {lit == got}")
if O.have_ref():
    sample = pairs[:min(200, n_pairs)]
    t = time.perf_counter(); ref = O.ref_ksw_cigars(sample); dr = time.perf_counter() - t
    print(f"reference ksw2 on one core: {len(sample) / dr:.0f} alignments/s; device records identical: {got[:len(sample)] == ref}")
