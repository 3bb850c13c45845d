"""Live check of the oracle against the compiled reference (only where
/root/reference and oracle/_ref exist, i.e. the build container).  Minimizer
configs only: the reference's raw-read build spends ~90 s in its 8 GiB flat
counter, the raw path is covered by the golden vectors."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.skipif(not (O.have_ref() and os.path.isdir("/root/reference/flye/config/bin_cfg")),
                                reason="reference build not available")


@pytest.mark.parametrize("preset,kind,opts", [
    ("hifi", "hifi", dict()),
    ("corrected", "hifi03", dict(rc_queries=True, max_overlaps=15)),
])
def test_oracle_equals_reference(built, tmp_path, preset, kind, opts):
    from flye_amd import config, synth
    fa = str(tmp_path / "r.fasta")
    rs = synth.simulate(seed=4242, genome_len=30_000, coverage=20, kind=kind, fasta_path=fa).filter_min_len(1000)
    info = O.run_ref(fa, config="/root/reference/flye/config/bin_cfg/" + config.CFG_FILES[preset], threads=4,
                     min_read_len=1000, index_out=str(tmp_path / "i.txt"), ovlp_out=str(tmp_path / "o.txt"),
                     rc_queries=opts.get("rc_queries", False), max_overlaps=opts.get("max_overlaps", 0))
    cfg = config.preset(preset)
    o = O.Oracle(17, threads=4)
    o.set_reads(rs)
    st = o.build_index(cfg)
    hdr, refidx = O.parse_ref_index(str(tmp_path / "i.txt"), rs)
    assert o.export_index().same_as(refidx)
    assert np.float32(st["sample_rate"]).view(np.uint32) == int(hdr["sampleRateBits"], 16)
    q = np.arange(1 if opts.get("rc_queries") else 0, 2 * rs.n, 2)
    res = o.overlaps(O.detector_params(cfg), q, max_overlaps=opts.get("max_overlaps", 0))
    ref = [l.strip() for l in open(tmp_path / "o.txt") if not l.startswith("#")]
    assert res.lines() == ref and len(ref) == info["overlaps"] > 0
