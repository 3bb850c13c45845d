"""Generates tests/golden/assemble_cases.json: the reference's own assemble stage
(`flye-modules assemble` = /root/reference/src/assemble/main_assemble.cpp:123-257: index build,
estimateOverlaperParameters, Extender::assembleDisjointigs with ChimeraDetector, ConsensusGenerator;
compiled unmodified into oracle/_ref/flye_assemble by oracle/Makefile) on seeded read sets, one thread.
Stored per case: sha256 / size / record count of draft_assembly.fasta and the number of disjointigs -- what
tests/test_seam.py expects of the SAME program with the device seams linked in (oracle/_ref/flye_assemble_gpu).

    python tests/golden/make_assemble_golden.py
"""
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from flye_amd import config, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {
    # solid k-mer index (countKmers + buildIndexUnevenCoverage), k-mer divergence, relative gate from
    # estimateOverlaperParameters' rand()-picked reads
    "asm_raw": dict(preset="raw", min_ovlp=3000,
                    sim=dict(seed=77, genome_len=300_000, coverage=30, kind="pb_raw")),
    # minimizer index, base-level divergence on homopolymer-compressed sequence (edlib), absolute gate
    "asm_hifi": dict(preset="hifi", min_ovlp=5000,
                     sim=dict(seed=78, genome_len=250_000, coverage=25, kind="hifi03", n_repeat_families=3)),
}


def run_case(case, binary=None, threads=1, env=None):
    """-> (fasta bytes, info)"""
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "reads.fasta")
        synth.simulate(fasta_path=fa, **case["sim"])
        cfg = config.write_cfg(os.path.join(tmp, "asm.cfg"), case["preset"])
        out = os.path.join(tmp, "draft_assembly.fasta")
        info = O.run_assemble(fa, cfg, out, threads=threads, min_ovlp=case["min_ovlp"], binary=binary, env=env)
        return open(out, "rb").read(), info


def digest(fasta: bytes) -> dict:
    return {"sha256": hashlib.sha256(fasta).hexdigest(), "bytes": len(fasta), "records": fasta.count(b">")}


def main():
    meta = {}
    for name, case in CASES.items():
        fasta, info = run_case(case)
        meta[name] = dict(case, **digest(fasta))
        print(name, meta[name]["records"], "disjointigs,", meta[name]["bytes"], "bytes,", f"{info['wall_s']:.1f} s", flush=True)
    json.dump(meta, open(os.path.join(HERE, "assemble_cases.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
