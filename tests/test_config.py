import os

import numpy as np
import pytest

REF_CFG = "/root/reference/flye/config/bin_cfg"


def _parse(path, out):
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        if line.startswith("%include"):
            _parse(os.path.join(os.path.dirname(path), line.split()[1]), out)
            continue
        k, v = line.split("=")
        out[k.strip()] = float(np.float32(float(v)))
    return out


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name", ["raw", "corrected", "hifi", "subasm"])
def test_presets_equal_reference_cfg_files(name):
    from flye_amd import config
    ref = _parse(os.path.join(REF_CFG, config.CFG_FILES[name]), {})
    mine = config.preset(name)
    for k, v in mine.items():
        if k == "minimizer_window" and not mine["use_minimizers"]:
            continue
        assert ref[k] == v, (name, k, ref[k], v)


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name", ["raw", "corrected", "hifi", "subasm"])
def test_assemble_stage_cfg_equals_reference_cfg_files(name, tmp_path):
    """the cfg file written for the reference's assemble program holds the reference's values for every
    key that program reads"""
    from flye_amd import config
    ref = _parse(os.path.join(REF_CFG, config.CFG_FILES[name]), {})
    mine = _parse(config.write_cfg(str(tmp_path / "a.cfg"), name), {})
    for k, v in mine.items():
        if k == "minimizer_window" and not mine["use_minimizers"]:
            continue
        assert ref[k] == v, (name, k, ref[k], v)
    for k in ("max_coverage_drop_rate", "chimera_window", "min_reads_in_disjointig", "max_inner_fraction",
              "low_cutoff_warning", "add_unassembled_reads", "max_extensions_drop_rate", "max_inner_reads"):
        assert k in mine


def test_min_overlap_rule():
    from flye_amd import config
    assert config.min_overlap_from_reads([8000] * 10) == 5000
    assert config.min_overlap_from_reads([1200] * 10) == 1000
    assert config.min_overlap_from_reads([2600] * 10 + [9000]) == 3000
    assert config.min_overlap_from_reads([9000] * 10, meta=True) == 3000
    assert config.min_overlap_from_reads([9000] * 10, "subasm") == 1000
