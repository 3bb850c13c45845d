"""Generates tests/golden/edlib_pairs.json: seeded descriptions of string pairs
(tests/helpers.edit_pair regenerates the strings) + the distance the REFERENCE's edlib
returns for each (oracle/_ref/ref_dumper --edlib-pairs = edlibAlign(NW, TASK_DISTANCE, k = -1),
the call of src/sequence/alignment.cpp:233-238), plain and homopolymer-compressed.

    python tests/golden/make_edlib_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import edit_pair, hpc  # noqa: E402
from oracle import oracle as O  # noqa: E402

SPECS = (
    # tiny and degenerate
    [dict(seed=1, n=0, err=1.0, m=0), dict(seed=2, n=0, err=1.0, m=7), dict(seed=3, n=5, err=1.0, m=0),
     dict(seed=4, n=1, err=0.0), dict(seed=5, n=1, err=1.0, m=1), dict(seed=6, n=63, err=0.1),
     dict(seed=7, n=64, err=0.1), dict(seed=8, n=65, err=0.1), dict(seed=9, n=200, err=0.0)]
    # true overlaps: D small (the O(ND) kernel's range)
    + [dict(seed=20 + i, n=n, err=e, hp=h) for i, (n, e, h) in enumerate(
        [(1000, 0.01, 0), (3000, 0.05, 20), (8000, 0.12, 60), (15000, 0.005, 100), (15000, 0.02, 0),
         (30000, 0.01, 200), (4095, 0.1, 0), (4096, 0.1, 0), (4097, 0.1, 0)])]
    # beyond ED_EMAX = 512: the band-doubling bit-vector kernel, one and several 4096-row strips
    + [dict(seed=40 + i, n=n, err=e, hp=h) for i, (n, e, h) in enumerate(
        [(6000, 0.15, 0), (9000, 0.12, 50), (12000, 0.25, 0), (20000, 0.08, 100), (33000, 0.05, 0)])]
    # unrelated substrings (what RepeatGraph::build's detector -- no overhang test, base-level
    # divergence, repeat_graph.cpp:84-93 -- asks for between spurious chains)
    + [dict(seed=60 + i, n=n, err=1.0, m=m) for i, (n, m) in enumerate(
        [(700, 700), (2000, 2600), (5000, 4100), (4096, 4096), (9000, 9500), (13000, 8200)])]
    # unequal lengths of related strings
    + [dict(seed=80, n=5000, err=0.03, shift=900), dict(seed=81, n=10000, err=0.1, shift=3000)]
    # longer than the O(ND) kernel's LDS piece (32768) and than ED_BIG_MIN (49152: multi-wave workgroup)
    + [dict(seed=90, n=40000, err=0.02), dict(seed=91, n=60000, err=0.03, hp=300),
       dict(seed=92, n=52000, err=0.15), dict(seed=93, n=50000, err=1.0, m=51000)]
)


def main():
    pairs = [edit_pair(s) for s in SPECS]
    plain = O.ref_edlib_distances(pairs)
    comp = O.ref_edlib_distances([(hpc(a), hpc(b)) for a, b in pairs])
    out = [dict(spec=s, n=int(len(a)), m=int(len(b)), dist=int(d), hpc_n=int(len(hpc(a))), hpc_m=int(len(hpc(b))),
                hpc_dist=int(h)) for s, (a, b), d, h in zip(SPECS, pairs, plain, comp)]
    with open(os.path.join(HERE, "edlib_pairs.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(len(out), "pairs; max distance", max(plain))


if __name__ == "__main__":
    main()
