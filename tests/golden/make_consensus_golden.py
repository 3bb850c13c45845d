"""Generates tests/golden/consensus_pairs.json: seeded overlapping read pairs (tests/helpers.edit_pair) and the sha256 of
what the REFERENCE's ConsensusGenerator::generateConsensuses (src/sequence/consensus_generator.cpp:18-126: per pair
one getAlignmentCigarKsw, decodeCigar, switch-position search, stitched sequence) prints for them through
oracle/_ref/ref_dumper --consensus-pairs, one thread.

    python tests/golden/make_consensus_golden.py
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

# overlaps between consecutive reads of a disjointig: 3-12 kb, raw-read error rates (two reads at 12 % each differ
# by ~20 %), some with homopolymers, a few with unequal lengths (band doubling)
SPECS = ([dict(seed=5000 + i, n=3000 + (i * 37) % 9000, err=(0.02, 0.1, 0.2, 0.25)[i % 4], hp=(0, 30)[i % 2]) for i in range(360)]
         + [dict(seed=6000 + i, n=4000 + 500 * i, err=0.15, shift=40 + 10 * i) for i in range(24)])


def main():
    pairs = [edit_pair(s) for s in SPECS]
    text, info = O.ref_consensus(pairs, threads=1)
    assert text.count("\n") == len(pairs)
    json.dump({"specs": SPECS, "pairs": len(pairs), "sha256": hashlib.sha256(text.encode()).hexdigest(),
               "reference_one_thread_s": round(info["consensus_s"], 3)},
              open(os.path.join(HERE, "consensus_pairs.json"), "w"), indent=0)
    print(len(pairs), "pairs,", info)


if __name__ == "__main__":
    main()
