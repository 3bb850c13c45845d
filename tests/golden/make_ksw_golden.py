"""Generates tests/golden/ksw_pairs.json: seeded string pairs (tests/helpers.edit_pair) + what the REFERENCE's
getAlignmentCigarKsw returns for each (oracle/_ref/ref_dumper --ksw-pairs: ksw_extz2_sse of the reference's
lib/minimap2 with Flye's scores, band 64 doubling, global backtrack, CIGAR decoded into = X I D runs,
src/sequence/alignment.cpp:102-216): the error rate's bit pattern, the number of CIGAR runs and a sha256 of
the CIGAR text (short ones verbatim).

    python tests/golden/make_ksw_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import edit_pair  # noqa: E402
from oracle import oracle as O  # noqa: E402

SPECS = (
    # shorter than the band / than one 16-byte vector: the vector code's out-of-array reads and writes matter here
    [dict(seed=1000 + i, n=n, err=1.0, m=m) for i, (n, m) in enumerate(
        [(1, 1), (1, 9), (7, 2), (15, 16), (16, 15), (17, 33), (31, 64), (64, 64), (65, 63), (90, 95), (96, 97), (100, 40)])]
    + [dict(seed=1100 + i, n=n, err=e) for i, (n, e) in enumerate([(20, 0.0), (50, 0.1), (63, 0.2), (80, 0.05), (127, 0.1)])]
    # true overlaps at several error rates, with homopolymers
    + [dict(seed=1200 + i, n=n, err=e, hp=h) for i, (n, e, h) in enumerate(
        [(500, 0.0, 0), (1000, 0.01, 0), (3000, 0.05, 20), (8000, 0.12, 60), (15000, 0.005, 100), (12000, 0.02, 0), (20000, 0.15, 0)])]
    # length differences beyond the band: 64 -> 128 -> 256 -> ...
    + [dict(seed=1300 + i, n=n, err=e, shift=sh) for i, (n, e, sh) in enumerate(
        [(2000, 0.02, 70), (3000, 0.02, 130), (4000, 0.05, 300), (6000, 0.01, 1200)])]
    + [dict(seed=1400 + i, n=n, err=1.0, m=m) for i, (n, m) in enumerate([(300, 900), (1500, 200), (2500, 2400), (60, 1000)])]
    # low complexity
    + [dict(seed=1500, n=600, err=0.03, hp=200), dict(seed=1501, n=2000, err=0.1, hp=600)]
)


def main():
    pairs = [edit_pair(s) for s in SPECS]
    pairs = [(a, b) for a, b in pairs]
    ref = O.ref_ksw_cigars(pairs)
    out = []
    for s, (a, b), (bits, cig) in zip(SPECS, pairs, ref):
        e = dict(spec=s, tlen=int(len(a)), qlen=int(len(b)), err_bits=bits, runs=len(cig.split()) if cig else 0,
                 cigar_sha256=hashlib.sha256(cig.encode()).hexdigest())
        if len(cig) < 200:
            e["cigar"] = cig
        out.append(e)
    with open(os.path.join(HERE, "ksw_pairs.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(len(out), "pairs")


if __name__ == "__main__":
    main()
