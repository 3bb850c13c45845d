"""On-disk text forms against what the reference itself wrote
(tests/golden/make_ondisk_golden.py): OverlapRange::dump / load (overlap.h:227-251), the
"\\tAln\\t<edgeId>\\t" record of ReadAligner::storeAlignments (read_aligner.cpp:321-339) and
SequenceContainer::writeFasta (sequence_container.cpp:330-357).  Host-only functions of the library:
no GPU needed; the overlap records come from the golden file of the same case."""
import gzip
import hashlib
import json
import os

import numpy as np

from helpers import GOLDEN, golden_lines, golden_reads


def _records_of_case(name):
    from flye_amd import gpu
    lines = golden_lines(name)
    recs = np.zeros(len(lines), gpu.REC_DTYPE)
    for i, l in enumerate(lines):
        t = l.split()
        for f, v in zip(("cur_id", "cur_begin", "cur_end", "cur_len", "ext_id", "ext_begin", "ext_end", "ext_len", "score"), t):
            recs[f][i] = int(v)
        recs["seq_divergence"].view(np.uint32)[i] = int(t[9], 16)
    return recs


def test_overlap_dump_and_load_equal_the_reference_text(built, golden_cases):
    from flye_amd import gpu
    meta = json.load(open(os.path.join(GOLDEN, "ondisk.json")))
    case = golden_cases[meta["case"]]
    recs = _records_of_case(meta["case"])
    names = [f"r{i}" for i in range(10**5)]           # the simulator's FASTA headers
    # ids of the kept reads: the dumper loads with --min-read-len, ids are dense over the KEPT reads
    from flye_amd import synth
    full = synth.simulate(**case["sim"])
    kept = np.nonzero(full.length > case["min_read_len"])[0]
    kept_names = [names[i] for i in kept]
    name_of = lambda rid: gpu.seq_name(kept_names, 0, rid)      # noqa: E731
    with gzip.open(os.path.join(GOLDEN, "ondisk_dump.txt.gz"), "rt") as f:
        want = f.read().splitlines()
    assert len(want) == meta["dump_lines"] == len(recs)
    got = gpu.dump_overlaps(recs, name_of, name_of)
    assert got == want
    # load: the reference's own lines come back as the same records
    ids = {gpu.seq_name(kept_names, 0, r): r for r in range(2 * len(kept_names))}
    back = gpu.load_overlaps(want, ids.__getitem__, ids.__getitem__)
    for f in ("cur_id", "cur_begin", "cur_end", "cur_len", "ext_id", "ext_begin", "ext_end", "ext_len", "score"):
        assert np.array_equal(back[f], recs[f]), f
    # the text carries 6 significant digits: re-dumping the loaded records reproduces it
    assert gpu.dump_overlaps(back, name_of, name_of) == want
    # ReadAligner::storeAlignments' record
    aln = gpu.dump_overlaps(recs[:5], name_of, name_of, edge_ids=[7, -7, 12, 1, 0])
    assert aln == [f"\tAln\t{e}\t{l}" for e, l in zip([7, -7, 12, 1, 0], want[:5])]


def test_fasta_writer_equals_the_reference_text(built, golden_cases):
    from flye_amd import gpu
    meta = json.load(open(os.path.join(GOLDEN, "ondisk.json")))
    case = golden_cases[meta["case"]]
    from flye_amd import synth
    full = synth.simulate(**case["sim"])
    kept = np.nonzero(full.length > case["min_read_len"])[0]
    rs = golden_reads(case)
    text = gpu.fasta_text(rs, [f"r{i}" for i in kept])
    assert text[:200] == meta["fasta_head"] and len(text) == meta["fasta_bytes"]
    assert hashlib.sha256(text.encode()).hexdigest() == meta["fasta_sha256"]
