"""The CPU oracle (oracle/flye_oracle.cpp) against the golden vectors produced by
the compiled, unmodified reference (tests/golden/make_golden.py)."""
import pytest

from helpers import (bits_to_float, case_queries, check_index_stats, golden_lines, golden_reads,
                     index_digest)

CASES = ["raw_pb", "raw_ont_rc", "raw_div", "raw_local", "hifi", "corrected_local", "hifi_rc_max", "raw_pb_aln"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(built, golden_cases, name):
    from flye_amd import config
    from oracle import oracle as O
    case = golden_cases[name]
    rs = golden_reads(case)
    cfg = config.preset(case["preset"])
    o = O.Oracle(int(cfg["kmer_size"]))
    o.set_reads(rs)
    st = o.build_index(cfg)
    check_index_stats(st, case["index"])
    assert index_digest(o.export_index()) == case["index"]["sha256"]
    p = O.detector_params(cfg, max_divergence=bits_to_float(case["max_div_bits"]),
                          keep_alignment=case.get("keep_aln", False))
    res = o.overlaps(p, case_queries(case, rs.n), max_overlaps=case.get("max_overlaps", 0),
                     force_local=case.get("force_local", False))
    assert res.lines() == golden_lines(name)
    assert len(res.recs) == case["n_overlaps"]


@pytest.mark.parametrize("name", ["edges_raw", "edges_hifi", "edges_raw_max", "edges_raw_aln", "edges_hifi_aln"])
def test_oracle_read_aligner_style_golden(built, golden_cases, name):
    """Queries from a second container against an index of other sequences, all primaries
    (ReadAligner::alignReads flags), vs the reference's output."""
    import numpy as np
    from flye_amd import config
    from oracle import oracle as O
    from helpers import edges_setup, golden_queries
    case = golden_cases[name]
    edges = golden_reads(case)
    reads = golden_queries(case)
    cfg = config.preset(case["preset"])
    wnd, dk = edges_setup(case, cfg)
    o = O.Oracle(int(cfg["kmer_size"]))
    o.set_reads(edges, 0)
    st = o.build_index_minimizers(1, wnd, cfg["repeat_kmer_rate"])
    check_index_stats(st, case["index"])
    assert index_digest(o.export_index()) == case["index"]["sha256"]
    o.set_queries(reads, 2 * edges.n)
    q = 2 * edges.n + np.arange(0, 2 * reads.n, 2)
    res = o.overlaps(O.detector_params(cfg, **dk), q, max_overlaps=case.get("max_overlaps", 0))
    assert res.lines() == golden_lines(name)
    assert len(res.recs) == case["n_overlaps"] > 0


@pytest.mark.parametrize("name", ["repeat_raw", "repeat_hifi"])
def test_oracle_repeat_stage_golden(built, golden_cases, name):
    """RepeatGraph::build flags: sequences against themselves, every primary, kmerMatches kept,
    base-level divergence; gated-out primaries returned marked (partition_bad_mappings)."""
    import numpy as np
    from flye_amd import config
    from oracle import oracle as O
    from helpers import check_repeat_stage_result, repeat_stage_setup
    case = golden_cases[name]
    seqs = golden_reads(case)
    cfg = config.preset(case["preset"])
    wnd, dk = repeat_stage_setup(case, cfg)
    o = O.Oracle(int(cfg["kmer_size"]))
    o.set_reads(seqs, 0)
    st = o.build_index_minimizers(1, wnd, cfg["repeat_kmer_rate"])
    check_index_stats(st, case["index"])
    assert index_digest(o.export_index()) == case["index"]["sha256"]
    res = o.overlaps(O.detector_params(cfg, **dk), np.arange(0, 2 * seqs.n, 2))
    check_repeat_stage_result(res, case, golden_cases)
