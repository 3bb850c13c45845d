"""Per-read-type parameter presets reaching the overlap hot path.

Values restate the reference's shipped config files (flat ``key = float`` text,
reference ``src/common/config.h:36-72``); every value is stored as a float32
exactly like ``Config::_parameters`` (``config.h:69,100``):

* ``flye/config/bin_cfg/asm_raw_reads.cfg:8-32``   -> ``raw``
* ``flye/config/bin_cfg/asm_corrected_reads.cfg``  -> ``corrected``
* ``flye/config/bin_cfg/asm_hifi.cfg:8-32``        -> ``hifi``
* ``flye/config/bin_cfg/asm_subasm.cfg``           -> ``subasm``
* ``flye/config/bin_cfg/asm_defaults.cfg:5``       -> ``meta_read_filter_kmer_freq``

``tests/test_config.py`` re-parses the real files when /root/reference is
present and checks these tables against them.
"""
from __future__ import annotations

import numpy as np

_COMMON = {
    "meta_read_filter_kmer_freq": 100,
    "repeat_kmer_rate": 100,
    "maximum_jump": 1500,
}

# what the callers of the path in the assemble stage read besides the presets: Extender / ChimeraDetector
# (reference src/assemble/extender.cpp, chimera.cpp; values flye/config/bin_cfg/asm_defaults.cfg:8-13 and the
# per-read-type files' low_cutoff_warning / add_unassembled_reads).  Only used to write a cfg file for the
# reference's own `flye-modules assemble` program when it runs over the device seams (tests, bench).
ASSEMBLE_STAGE = {
    "max_coverage_drop_rate": 5, "max_extensions_drop_rate": 5, "chimera_window": 100,
    "min_reads_in_disjointig": 4, "max_inner_reads": 10, "max_inner_fraction": 0.25,
}
ASSEMBLE_STAGE_BY_PRESET = {
    "raw": {"low_cutoff_warning": 1, "add_unassembled_reads": 0},
    "corrected": {"low_cutoff_warning": 0, "add_unassembled_reads": 0},
    "hifi": {"low_cutoff_warning": 0, "add_unassembled_reads": 0},
    "subasm": {"low_cutoff_warning": 0, "add_unassembled_reads": 1},
}

PRESETS = {
    "raw": dict(_COMMON, kmer_size=17, use_minimizers=0, minimizer_window=0,
                reads_base_alignment=0, assemble_kmer_sample=1,
                meta_read_top_kmer_rate=0.40, maximum_overhang=1500,
                assemble_ovlp_divergence=0.10, assemble_divergence_relative=1,
                hpc_scoring_on=0),
    "corrected": dict(_COMMON, kmer_size=17, use_minimizers=1, minimizer_window=5,
                      reads_base_alignment=1, assemble_kmer_sample=2,
                      meta_read_top_kmer_rate=0.75, maximum_overhang=500,
                      assemble_ovlp_divergence=0.03, assemble_divergence_relative=0,
                      hpc_scoring_on=0),
    "hifi": dict(_COMMON, kmer_size=17, use_minimizers=1, minimizer_window=10,
                 reads_base_alignment=1, assemble_kmer_sample=2,
                 meta_read_top_kmer_rate=0.75, maximum_overhang=500,
                 assemble_ovlp_divergence=0.01, assemble_divergence_relative=0,
                 hpc_scoring_on=1),
    "subasm": dict(_COMMON, kmer_size=31, use_minimizers=1, minimizer_window=10,
                   reads_base_alignment=1, assemble_kmer_sample=2,
                   meta_read_top_kmer_rate=0.75, maximum_jump=500, maximum_overhang=100,
                   assemble_ovlp_divergence=0.02, assemble_divergence_relative=0,
                   hpc_scoring_on=0),
}

# constants of the assemble stage (reference src/assemble/main_assemble.cpp)
MIN_FREQ = 2            # :207
DETECTOR_MIN_OVERLAP = 1000  # :174, :231 -- always 1000 in `assemble`

CFG_FILES = {"raw": "asm_raw_reads.cfg", "corrected": "asm_corrected_reads.cfg",
             "hifi": "asm_hifi.cfg", "subasm": "asm_subasm.cfg"}


def preset(name: str) -> dict:
    """Preset with every value rounded through float32 like Config::get()."""
    return {k: float(np.float32(v)) for k, v in PRESETS[name].items()}


def params_string(name: str) -> str:
    """``key=value,...`` form accepted by the reference's ``Config::addParameters``
    (``config.h:84-96``) -- lets the test-side reference dumper run without cfg files."""
    return ",".join(f"{k}={v!r}" if isinstance(v, float) else f"{k}={v}"
                    for k, v in PRESETS[name].items())


def assemble_stage(name: str) -> dict:
    """Every key `flye-modules assemble` reads (main_assemble.cpp, extender.cpp, chimera.cpp, overlap.cpp,
    vertex_index.cpp) for read type ``name``."""
    d = dict(PRESETS[name])
    d.update(ASSEMBLE_STAGE)
    d.update(ASSEMBLE_STAGE_BY_PRESET[name])
    return d


def write_cfg(path: str, name: str, extra: dict | None = None) -> str:
    """A flat ``key = value`` file in the format Config::load parses (config.h:36-72) holding
    ``assemble_stage(name)``: the --config argument of the reference's assemble program."""
    d = assemble_stage(name)
    if extra:
        d.update(extra)
    with open(path, "w") as f:
        for k, v in d.items():
            f.write(f"{k} = {v!r}\n" if isinstance(v, float) else f"{k} = {v}\n")
    return path


def min_overlap_from_reads(lengths, read_type="raw", meta=False) -> int:
    """Minimum-overlap selection of the pipeline driver (reference
    flye/config/configurator.py:51-59, :86-96; ranges flye/config/py_cfg.py:31-37):
    N90 of the read lengths rounded to 1 kb, clamped to [1000, 5000].  The assemble
    stage then drops reads with length <= max(min_read, min_overlap)
    (src/assemble/main_assemble.cpp:183)."""
    ls = sorted((int(x) for x in lengths), reverse=True)
    total = sum(ls)
    n90, acc = 0, 0
    for l in ls:
        acc += l
        if acc > 0.90 * total:
            n90 = l
            break
    lo, hi = (1000, 1000) if read_type == "subasm" else (1000, 5000)
    if meta:
        hi = min(hi, 3000)
    return max(lo, min(hi, int(round(n90 / 1000)) * 1000))
