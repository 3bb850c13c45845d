"""The compiled Flye-side binding (integration/flye_seam.cpp): the reference's own dumper program with
OverlapDetector::getSeqOverlaps and the VertexIndex build replaced AT LINK TIME by the device path
(oracle/_ref/ref_dumper_gpu, integration/Makefile).  Everything above the two seams is the
reference's compiled code -- OverlapContainer (quick / lazy, estimateOverlaperParameters with libc
rand(), findAllOverlaps, ensureTransitivity, filterOverlaps), processInParallel worker threads,
checkIdyAndTrim (ksw2) -- so these tests compare "Flye with the GPU hot path" with the golden files
the pure reference wrote: byte-identical text, 8 caller threads through the batch scheduler
(fgb_quick_ex)."""
import gzip
import json
import os
import sys

import pytest

from helpers import GOLDEN

sys.path.insert(0, GOLDEN)

pytestmark = pytest.mark.gpu


def _have():
    from oracle import oracle as O
    return O.have_ref_gpu()


def _run_case(case, tmp_path, threads=8):
    from flye_amd import config, synth
    from oracle import oracle as O
    fa = str(tmp_path / "reads.fasta")
    synth.simulate(fasta_path=fa, **case["sim"])
    params = config.params_string(case["preset"])
    extra = {}
    if case.get("minimizer_index"):
        cfgd = config.preset(case["preset"])
        wnd = int(cfgd["minimizer_window"]) if cfgd["use_minimizers"] else 1
        params += f",use_minimizers=1,minimizer_window={wnd}"
        extra = dict(only_max=case["only_max"], max_overhang=case["max_overhang"], nucl_aln=case["nucl_aln"],
                     min_overlap=case["min_overlap"])
        if "max_div" in case:
            extra["max_div"] = case["max_div"]
        if "queries_sim" in case:
            qfa = str(tmp_path / "queries.fasta")
            synth.simulate(fasta_path=qfa, **case["queries_sim"])
            extra["queries_fasta"] = qfa
    out = str(tmp_path / "ovlp.txt")
    info = O.run_ref(fa, params_string=params, threads=threads, min_read_len=case["min_read_len"],
                     max_overlaps=case.get("max_overlaps", 0), force_local=case.get("force_local", False),
                     div_mode=case.get("div_mode", "none"), ovlp_out=out, rc_queries=case.get("rc_queries", False),
                     keep_aln=case.get("keep_aln", False), binary=O.REF_DUMPER_GPU, **extra)
    return open(out).read(), info


@pytest.mark.parametrize("name", [
    "raw_pb",            # solid k-mer index, assemble-stage detector
    "raw_div",           # + estimateOverlaperParameters / setDivergenceThreshold (rand()-picked reads, relative gate)
    "raw_ont_rc",        # reverse-complement queries, maxOverlaps
    "raw_local",         # forceLocal
    "hifi",              # minimizer index, base-level divergence, homopolymer compression
    "hifi_rc_max",
    "edges_raw_aln",     # queries from a SECOND container (reads vs edges): sequences travel with the requests; kmerMatches
    "edges_hifi",
    "repeat_raw",        # RepeatGraph::build's flag set
])
def test_reference_program_with_device_seams_reproduces_golden(built, golden_cases, tmp_path, name):
    if not _have():
        pytest.skip("oracle/_ref/ref_dumper_gpu not built (needs /root/reference at build time)")
    case = golden_cases[name]
    text, info = _run_case(case, tmp_path)
    with gzip.open(os.path.join(GOLDEN, name + ".ovlp.gz"), "rt") as f:
        want = f.read()
    assert info["overlaps"] == case["n_overlaps"]
    assert text == want      # header line (gate, estimated mean divergence, sample rate bits) + every OverlapRange


@pytest.mark.parametrize("name", ["findall_repeat", "findall_repeat_hpc"])
def test_find_all_overlaps_above_the_device_seam(built, tmp_path, name):
    """OverlapContainer::findAllOverlaps + ensureTransitivity + filterOverlaps (overlap.cpp:576-741) and
    checkIdyAndTrim on the records the device marks needs_trim: reference code over device results."""
    if not _have():
        pytest.skip("oracle/_ref/ref_dumper_gpu not built")
    import make_findall_golden as M
    from oracle import oracle as O
    with open(os.path.join(GOLDEN, "findall_cases.json")) as f:
        meta = json.load(f)
    text, info = M.run_case(M.CASES[name], binary=O.REF_DUMPER_GPU, threads=8)
    with gzip.open(os.path.join(GOLDEN, name + ".txt.gz"), "rt") as f:
        want = f.read()
    assert info["find_all_overlaps"] == meta[name]["overlaps"] > 0
    assert text == want
