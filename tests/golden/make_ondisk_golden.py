"""Generates tests/golden/ondisk_*.gz / ondisk.json: the reference's own on-disk text forms for one
golden case -- every OverlapRange through OverlapRange::dump (src/sequence/overlap.h:227-236) and the
loaded reads through SequenceContainer::writeFasta (sequence_container.cpp:330-357) -- written by the
UNMODIFIED reference (oracle/_ref/ref_dumper --dump-out / --fasta-out).

    python tests/golden/make_ondisk_golden.py
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from flye_amd import config, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASE = "raw_ont_rc"     # reverse-complement queries: '-' names on the query side, both strands on the target side


def main():
    case = json.load(open(os.path.join(HERE, "cases.json")))[CASE]
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "reads.fasta")
        synth.simulate(fasta_path=fa, **case["sim"])
        dump, fasta = os.path.join(tmp, "dump.txt"), os.path.join(tmp, "out.fasta")
        cmd = [O.REF_DUMPER, "--reads", fa, "--params", config.params_string(case["preset"]), "--threads", "8",
               "--min-read-len", str(case["min_read_len"]), "--max-overlaps", str(case.get("max_overlaps", 0)),
               "--rc-queries", "--dump-out", dump, "--fasta-out", fasta]
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(dump).read()
        ftext = open(fasta).read()
    with gzip.GzipFile(os.path.join(HERE, "ondisk_dump.txt.gz"), "wb", mtime=0) as f:
        f.write(text.encode())
    meta = dict(case=CASE, dump_lines=text.count("\n"), fasta_sha256=hashlib.sha256(ftext.encode()).hexdigest(),
                fasta_bytes=len(ftext), fasta_head=ftext[:200])
    json.dump(meta, open(os.path.join(HERE, "ondisk.json"), "w"), indent=1, sort_keys=True)
    print(meta["dump_lines"], "dump lines;", meta["fasta_bytes"], "FASTA bytes")


if __name__ == "__main__":
    main()
