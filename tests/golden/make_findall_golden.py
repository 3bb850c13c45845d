"""Generates tests/golden/findall_*.txt.gz: OverlapContainer::findAllOverlaps (reference
src/sequence/overlap.cpp:625-665 = lazySeqOverlaps for every forward sequence +
ensureTransitivity(false) + filterOverlaps) with RepeatGraph::build's detector
(repeat_graph.cpp:84-93: no overhang test, kmerMatches kept, every primary, base-level
divergence, bad mappings partitioned through checkIdyAndTrim / ksw2), run by the UNMODIFIED
reference (oracle/_ref/ref_dumper --find-all, one thread).  Lines are sorted inside each
sequence's list (their stored order depends on hash-table iteration order).

    python tests/golden/make_findall_golden.py
"""
import gzip
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
    "findall_repeat": dict(preset="raw", sim=dict(seed=214, genome_len=70_000, coverage=4, kind="hifi03", median_len=15000,
                                                   min_len=6000, max_len=30000, n_repeat_families=10, repeat_div_permille=40),
                           min_overlap=1000, max_div=0.01),
    "findall_repeat_hpc": dict(preset="hifi", sim=dict(seed=215, genome_len=60_000, coverage=4, kind="hifi", median_len=15000,
                                                       min_len=6000, max_len=30000, n_repeat_families=8, n_homopolymers=40,
                                                       repeat_div_permille=30),
                               min_overlap=1000, max_div=0.005),
}


def run_case(case, binary=None, threads=1, env=None):
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "seqs.fasta")
        synth.simulate(fasta_path=fa, **case["sim"])
        cfgd = config.preset(case["preset"])
        wnd = int(cfgd["minimizer_window"]) if cfgd["use_minimizers"] else 1
        params = config.params_string(case["preset"]) + f",use_minimizers=1,minimizer_window={wnd}"   # repeat_graph.cpp:75-77
        out = os.path.join(tmp, "o.txt")
        info = O.run_ref(fa, params_string=params, threads=threads, min_read_len=0, min_overlap=case["min_overlap"],
                         only_max=False, max_overhang=0, nucl_aln=True, keep_aln=True, max_div=case["max_div"],
                         find_all=True, partition_bad=True, ovlp_out=out, binary=binary, env=env)
        return open(out).read(), info


def main():
    meta = {}
    for name, case in CASES.items():
        text, info = run_case(case)
        with gzip.GzipFile(os.path.join(HERE, name + ".txt.gz"), "wb", mtime=0) as f:
            f.write(text.encode())
        meta[name] = dict(case, overlaps=info["find_all_overlaps"])
        print(name, info)
    json.dump(meta, open(os.path.join(HERE, "findall_cases.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
