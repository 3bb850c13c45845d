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


# ---- the reference's whole assemble stage over the seams ---------------------------------
def _assemble_cases():
    with open(os.path.join(GOLDEN, "assemble_cases.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["asm_raw", "asm_hifi"])
def test_assemble_stage_with_device_seams_writes_the_reference_draft(built, name):
    """`flye-modules assemble` (main_assemble.cpp:123-257: index build, estimateOverlaperParameters with libc
    rand(), Extender::assembleDisjointigs, ChimeraDetector::estimateGlobalCoverage, ConsensusGenerator) compiled
    from the reference's files with integration/flye_seam.o in front: one thread -> draft_assembly.fasta
    byte-identical to what the pure reference program wrote (tests/golden/make_assemble_golden.py)."""
    from oracle import oracle as O
    if not O.have_assemble_gpu():
        pytest.skip("oracle/_ref/flye_assemble_gpu not built (needs /root/reference at build time)")
    import hashlib
    import make_assemble_golden as M
    case = _assemble_cases()[name]
    fasta, info = M.run_case(case, binary=O.FLYE_ASSEMBLE_GPU, threads=1)
    assert fasta.count(b">") == case["records"] > 0
    assert len(fasta) == case["bytes"]
    assert hashlib.sha256(fasta).hexdigest() == case["sha256"]
    s = info["seams"]
    assert s["get_seq_overlaps_calls"] > 0 and s["bridge"]["device_calls"] > 0 and s["consensus_pairs"] > 0
    print(f"{name}: wall {info['wall_s']:.1f} s, seams {s}")


def test_assemble_stage_many_threads_completes(built):
    """16 caller threads through the batch scheduler: the disjointigs depend on thread timing in the reference
    itself (extender.cpp:269 takes reads under a mutex in arrival order), so only the shape is checked."""
    from oracle import oracle as O
    if not O.have_assemble_gpu():
        pytest.skip("oracle/_ref/flye_assemble_gpu not built")
    import make_assemble_golden as M
    case = _assemble_cases()["asm_raw"]
    fasta, info = M.run_case(case, binary=O.FLYE_ASSEMBLE_GPU, threads=16)
    assert fasta.count(b">") >= 1 and 0.8 * case["bytes"] < len(fasta) < 1.3 * case["bytes"]
    s = info["seams"]
    assert s["bridge"]["requests"] >= s["bridge"]["device_calls"] > 0


def test_consensus_alignments_as_one_device_batch(built):
    """ConsensusGenerator::generateConsensuses through the seam's generateAlignments (every pair of every
    disjointig in ONE fg_align_cigar_ksw batch) and, for single callers, the alignment dispatcher: the
    consensus sequences equal the reference's (tests/golden/consensus_pairs.json) and the stage runs at >= 10x
    the reference's one-thread rate."""
    from oracle import oracle as O
    if not _have():
        pytest.skip("oracle/_ref/ref_dumper_gpu not built")
    import hashlib
    from helpers import edit_pair
    with open(os.path.join(GOLDEN, "consensus_pairs.json")) as f:
        gold = json.load(f)
    pairs = [edit_pair(s) for s in gold["specs"]]
    text, info = O.ref_consensus(pairs, threads=16, binary=O.REF_DUMPER_GPU)
    assert hashlib.sha256(text.encode()).hexdigest() == gold["sha256"]
    rate = gold["pairs"] / info["consensus_s"]
    ref_s = gold["reference_one_thread_s"]
    if O.have_ref():        # the compiled reference on THIS host, one thread, same pairs
        ref_text, ref_info = O.ref_consensus(pairs, threads=1)
        assert ref_text == text
        ref_s = ref_info["consensus_s"]
    ref_rate = gold["pairs"] / ref_s
    print(f"consensus: {gold['pairs']} pairs in {info['consensus_s'] * 1e3:.0f} ms = {rate:.0f}/s; reference, one thread: {ref_rate:.0f}/s")
    # a batch this small is bound by the LATENCY of its longest alignment on one wave, not by
    # throughput: the bar here is "no slower than the reference's one thread"; tools/ksw_bench.py has the batch rates
    assert rate >= 1.0 * ref_rate
