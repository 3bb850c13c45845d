"""One segment at a time through fg_debug_sort_pairs against the oracle's std::sort permutation: which input shapes
the device sort (level kernels, k_sort_mid, k_sort_lds) gets wrong, stalls on, or is slow on."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flye_amd import gpu
from oracle import oracle as O


def killer(n):
    k = n // 2
    a = np.zeros(n, np.uint64)
    for i in range(1, k + 1):
        if i % 2 == 1:
            a[i - 1] = i
            a[i] = k + i
        a[k + i - 1] = 2 * i
    return a


rng = np.random.default_rng(7)
cases = []
for n in (449, 600, 1000, 3000, 4096, 4097, 9000, 20000, 70000):
    for hi in (2, 5, 50, 1 << 20, 1 << 40):
        cases.append((f"random n={n} hi={hi}", rng.integers(0, hi, size=n, dtype=np.uint64)))
for n in (1000, 5000):
    cases += [(f"asc {n}", np.arange(n, dtype=np.uint64)), (f"desc {n}", np.arange(n, dtype=np.uint64)[::-1].copy()),
              (f"zeros {n}", np.zeros(n, np.uint64)), (f"mod3 {n}", (np.arange(n) % 3).astype(np.uint64)),
              (f"div7 {n}", (np.arange(n) // 7).astype(np.uint64))]
for n in (1000, 20000, 100000):
    cases += [(f"killer {n}", killer(n)), (f"killer/3 {n}", killer(n) // 3)]
ctx = gpu.Context(17, 0)
bad = 0
for name, s in cases:
    off = np.array([0, len(s)], np.uint64)
    t0 = time.perf_counter()
    try:
        sk, perm = ctx.debug_sort_pairs(s.copy(), off)
        want = O.std_sort_perm(s)
        ok = np.array_equal(perm, want)
        msg = "ok" if ok else "WRONG permutation"
    except Exception as e:  # noqa: BLE001
        ok, msg = False, f"ERROR {e}"
    bad += not ok
    print(f"{name:28s} {msg:20s} {(time.perf_counter() - t0) * 1e3:8.1f} ms", flush=True)
print("bad:", bad)
sys.exit(1 if bad else 0)
