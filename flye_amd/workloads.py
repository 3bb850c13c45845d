"""Named synthetic workloads = BASELINE.json configs (SURVEY.md §8d generators)."""
from __future__ import annotations

from . import config, synth


def ecoli_pb50(seed=12345, scale=1.0, coverage=50):
    """configs[0]/[1]: "E.coli PB 50x": 4.64 Mb genome (x scale) with 7 x 5 kb and
    50 x 1.3 kb planted repeat copies, PacBio-raw error model, ~232 Mbp of reads;
    min overlap from the N90 rule, reads <= min overlap dropped."""
    glen = int(4_640_000 * scale)
    fam = max(1, int(round(8 * scale)))
    rs = synth.simulate(seed=seed, genome_len=glen, coverage=coverage, kind="pb_raw",
                        n_repeat_families=fam, repeat_len=(1300, 5000), repeat_copies=(5, 9),
                        repeat_div_permille=10, n_homopolymers=int(400 * scale), n_tandems=int(300 * scale))
    min_ovlp = config.min_overlap_from_reads(rs.length, "raw")
    return rs.filter_min_len(min_ovlp), min_ovlp, "raw"


def dmel_ont30(seed=3, scale=1.0):
    """configs[2]: D. melanogaster ONT 30x proxy: 136 Mb genome, ~20 % repeats."""
    glen = int(136_000_000 * scale)
    rs = synth.simulate(seed=seed, genome_len=glen, coverage=30, kind="ont_raw",
                        n_repeat_families=max(1, int(300 * scale)), repeat_len=(500, 10000),
                        repeat_copies=(10, 60), repeat_div_permille=30,
                        n_homopolymers=int(20000 * scale), n_tandems=int(20000 * scale))
    min_ovlp = config.min_overlap_from_reads(rs.length, "raw")
    return rs.filter_min_len(min_ovlp), min_ovlp, "raw"


def synth10g_ont(seed=11, scale=1.0):
    """configs[3]: "Synthetic 10 Gb ONT-error-profile reads": 334 Mb genome, 30x ONT-raw model
    (10 % error, log-normal lengths), ~15 % repeats -- 10 Gbp of reads at scale 1."""
    glen = int(334_000_000 * scale)
    rs = synth.simulate(seed=seed, genome_len=glen, coverage=30, kind="ont_raw",
                        n_repeat_families=max(1, int(500 * scale)), repeat_len=(500, 10000),
                        repeat_copies=(10, 40), repeat_div_permille=30,
                        n_homopolymers=int(40000 * scale), n_tandems=int(40000 * scale))
    min_ovlp = config.min_overlap_from_reads(rs.length, "raw")
    return rs.filter_min_len(min_ovlp), min_ovlp, "raw"


def hifi30(seed=5, genome_len=4_640_000):
    """HiFi-parameter workload (asm_hifi.cfg) on an E. coli sized genome."""
    scale = genome_len / 4_640_000
    rs = synth.simulate(seed=seed, genome_len=genome_len, coverage=30, kind="hifi03",
                        n_repeat_families=max(1, int(round(8 * scale))), repeat_len=(1300, 5000),
                        repeat_copies=(5, 9), n_homopolymers=int(400 * scale), n_tandems=int(300 * scale))
    min_ovlp = config.min_overlap_from_reads(rs.length, "hifi")
    return rs.filter_min_len(min_ovlp), min_ovlp, "hifi"
