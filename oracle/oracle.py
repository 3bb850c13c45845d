"""TEST INFRASTRUCTURE -- ctypes binding of oracle/liboracle.so (the CPU
restatement) and helpers to drive oracle/_ref/ref_dumper (the compiled,
unmodified reference).  Imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DUMPER = os.path.join(_HERE, "_ref", "ref_dumper")

REC_DTYPE = np.dtype([("cur_id", "<u4"), ("ext_id", "<u4"), ("cur_begin", "<i4"),
                      ("cur_end", "<i4"), ("cur_len", "<i4"), ("ext_begin", "<i4"),
                      ("ext_end", "<i4"), ("ext_len", "<i4"), ("score", "<i4"),
                      ("seq_divergence", "<f4"), ("chain_length", "<i4"),
                      ("filtered_positions", "<i4"), ("edit_distance", "<i4"),
                      ("hpc_len_cur", "<i4"), ("hpc_len_ext", "<i4")])
assert REC_DTYPE.itemsize == 60


class IndexStats(C.Structure):
    _fields_ = [("total_kmers", C.c_uint64), ("selected_kmers", C.c_uint64),
                ("index_entries", C.c_uint64), ("repetitive_kmers", C.c_uint64),
                ("repetitive_frequency", C.c_uint64), ("mean_frequency", C.c_float),
                ("sample_rate", C.c_float), ("build_seconds", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class DetectorParams(C.Structure):
    _fields_ = [("max_jump", C.c_int32), ("min_overlap", C.c_int32),
                ("max_overhang", C.c_int32), ("keep_alignment", C.c_uint8),
                ("only_max_ext", C.c_uint8), ("nucl_alignment", C.c_uint8),
                ("partition_bad_mappings", C.c_uint8), ("use_hpc", C.c_uint8),
                ("pad_", C.c_uint8 * 3), ("max_divergence", C.c_float)]


def detector_params(cfg: dict, min_overlap=1000, max_divergence=1.0, only_max_ext=True, max_overhang=None,
                    nucl_alignment=None, keep_alignment=False, partition_bad_mappings=False):
    """OverlapDetector ctor arguments as main_assemble.cpp:229-238 passes them (or, with the
    overrides, as read_aligner.cpp:186-192 does)."""
    return DetectorParams(max_jump=int(cfg["maximum_jump"]), min_overlap=int(min_overlap),
                          max_overhang=int(cfg["maximum_overhang"] if max_overhang is None else max_overhang),
                          keep_alignment=int(keep_alignment), only_max_ext=int(only_max_ext),
                          nucl_alignment=int(bool(cfg["reads_base_alignment"]) if nucl_alignment is None
                                             else bool(nucl_alignment)),
                          partition_bad_mappings=int(partition_bad_mappings), use_hpc=int(bool(cfg["hpc_scoring_on"])),
                          max_divergence=float(max_divergence))


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run __graft_entry__.build()")
        L = C.CDLL(path)
        L.fo_create.restype = C.c_void_p
        L.fo_create.argtypes = [C.c_int]
        L.fo_destroy.argtypes = [C.c_void_p]
        L.fo_set_reads.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.fo_set_queries.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.fo_build_index_solid.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_float,
                                           C.c_float, C.c_int, C.POINTER(IndexStats)]
        L.fo_build_index_minimizers.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int,
                                                C.POINTER(IndexStats)]
        L.fo_minimizers.restype = C.c_int64
        L.fo_minimizers.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_int64]
        L.fo_import_index.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint64, C.c_void_p, C.c_float]
        L.fo_export_index.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.fo_overlaps.argtypes = [C.c_void_p, C.POINTER(DetectorParams), C.c_void_p, C.c_uint32,
                                  C.c_int32, C.c_uint8, C.c_int, C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_uint64)]
        L.fo_fetch.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.fo_fetch_trim.restype = C.c_uint64
        L.fo_fetch_trim.argtypes = [C.c_void_p, C.c_void_p]
        L.fo_fetch_matches.restype = C.c_uint64
        L.fo_fetch_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.fo_edit_distance.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.fo_edit_distance_dp.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.fo_edit_distance_k0.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.fo_ksw_cigar.restype = C.c_int64
        L.fo_ksw_cigar.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.POINTER(C.c_float)]
        L.fo_introsort_mismatches.restype = C.c_int64
        L.fo_introsort_mismatches.argtypes = [C.c_void_p, C.c_int64]
        L.fo_std_sort_perm.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        _LIB = L
    return _LIB


class IndexExport:
    def __init__(self, keys, key_off, entries, repetitive):
        self.keys, self.key_off, self.entries, self.repetitive = keys, key_off, entries, repetitive

    def same_as(self, other) -> bool:
        return (np.array_equal(self.keys, other.keys) and np.array_equal(self.key_off, other.key_off)
                and np.array_equal(self.entries, other.entries)
                and np.array_equal(self.repetitive, other.repetitive))

    def nonempty(self) -> "IndexExport":
        """Drop keys whose list is empty: they are indistinguishable from absent
        keys at lookup time (overlap.cpp:183 guards with kmerFreq())."""
        cnt = np.diff(self.key_off.astype(np.int64))
        keep = cnt > 0
        off = np.zeros(int(keep.sum()) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(cnt[keep])
        return IndexExport(self.keys[keep], off, self.entries, self.repetitive)


def _export(fn, handle):
    nk, ne, nr = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = fn(handle, C.byref(nk), C.byref(ne), C.byref(nr), None, None, None, None)
    if rc != 0:
        raise RuntimeError(f"export failed: {rc}")
    keys = np.empty(nk.value, np.uint64)
    off = np.empty(nk.value + 1, np.uint64)
    ent = np.empty(ne.value, np.uint64)
    rep = np.empty(nr.value, np.uint64)
    rc = fn(handle, C.byref(nk), C.byref(ne), C.byref(nr), keys.ctypes.data, off.ctypes.data,
            ent.ctypes.data, rep.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"export failed: {rc}")
    return IndexExport(keys, off, ent, rep)


class OverlapResult:
    def __init__(self, query_off, recs, stat_off, stats, counters):
        self.query_off, self.recs, self.stat_off, self.stats = query_off, recs, stat_off, stats
        self.needs_trim = self.match_off = self.matches = None
        (self.query_bp, self.query_kmers, self.seed_hits, self.dp_groups,
         self.dp_elements) = [int(x) for x in counters]

    def lines(self):
        """Canonical text form shared with ref_dumper's --ovlp-out."""
        r = self.recs
        bits = r["seq_divergence"].view(np.uint32)
        out = [f"{r['cur_id'][i]} {r['cur_begin'][i]} {r['cur_end'][i]} {r['cur_len'][i]} "
               f"{r['ext_id'][i]} {r['ext_begin'][i]} {r['ext_end'][i]} {r['ext_len'][i]} "
               f"{r['score'][i]} {bits[i]:08x}" for i in range(len(r))]
        if getattr(self, "match_off", None) is not None:   # keep_alignment: + count and digest of kmerMatches
            h = match_hashes(self.match_off, self.matches)
            cnt = np.diff(self.match_off.astype(np.int64))
            out = [f"{l} {cnt[i]} {int(h[i]):016x}" for i, l in enumerate(out)]
        return out


class Oracle:
    def __init__(self, k=17, threads=None):
        self.L = lib()
        self.h = self.L.fo_create(k)
        self.k = k
        self.threads = threads or min(16, os.cpu_count() or 1)

    def close(self):
        if self.h:
            self.L.fo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_reads(self, rs, first_seq_id=0):
        self.rs = rs
        self.first_id = first_seq_id
        rc = self.L.fo_set_reads(self.h, rs.n, rs.words.ctypes.data, rs.word_off.ctypes.data,
                                 rs.length.ctypes.data, first_seq_id)
        assert rc == 0

    def set_queries(self, rs, first_seq_id):
        """Second container holding the queries (ReadAligner-style)."""
        self.qrs = rs
        rc = self.L.fo_set_queries(self.h, rs.n, rs.words.ctypes.data, rs.word_off.ctypes.data,
                                   rs.length.ctypes.data, first_seq_id)
        assert rc == 0

    def build_index_solid(self, min_freq, select_rate, tandem_freq, repeat_rate, sample_rate_init=1.0):
        st = IndexStats()
        rc = self.L.fo_build_index_solid(self.h, min_freq, select_rate, tandem_freq, repeat_rate,
                                         sample_rate_init, self.threads, C.byref(st))
        if rc != 0:
            raise RuntimeError(f"fo_build_index_solid: {rc}")
        return st.as_dict()

    def build_index_minimizers(self, min_coverage, window, repeat_rate):
        st = IndexStats()
        rc = self.L.fo_build_index_minimizers(self.h, min_coverage, window, repeat_rate,
                                              self.threads, C.byref(st))
        if rc != 0:
            raise RuntimeError(f"fo_build_index_minimizers: {rc}")
        return st.as_dict()

    def build_index(self, cfg: dict):
        """Index build exactly as main_assemble.cpp:195-223 selects it."""
        if cfg["use_minimizers"]:
            return self.build_index_minimizers(1, int(cfg["minimizer_window"]), cfg["repeat_kmer_rate"])
        return self.build_index_solid(2, cfg["meta_read_top_kmer_rate"],
                                      int(cfg["meta_read_filter_kmer_freq"]), cfg["repeat_kmer_rate"],
                                      float(int(cfg["assemble_kmer_sample"])))

    def minimizers(self, read, window):
        cap = int(self.rs.length[read]) + 1
        out = np.empty(cap, np.int32)
        n = self.L.fo_minimizers(self.h, read, window, out.ctypes.data, cap)
        return out[:n].copy()

    def export_index(self):
        return _export(self.L.fo_export_index, self.h)

    def import_index(self, ex: IndexExport, sample_rate: float):
        rc = self.L.fo_import_index(self.h, len(ex.keys), ex.keys.ctypes.data, ex.key_off.ctypes.data,
                                    ex.entries.ctypes.data, len(ex.repetitive),
                                    ex.repetitive.ctypes.data, sample_rate)
        assert rc == 0

    def overlaps(self, params: DetectorParams, query_ids, max_overlaps=0, force_local=False,
                 threads=None):
        q = np.ascontiguousarray(query_ids, dtype=np.uint32)
        nr, ns = C.c_uint64(), C.c_uint64()
        rc = self.L.fo_overlaps(self.h, C.byref(params), q.ctypes.data, len(q), max_overlaps,
                                int(force_local), threads or self.threads, C.byref(nr), C.byref(ns))
        if rc != 0:
            raise RuntimeError(f"fo_overlaps: {rc}")
        qo = np.empty(len(q) + 1, np.uint64)
        so = np.empty(len(q) + 1, np.uint64)
        recs = np.empty(nr.value, REC_DTYPE)
        stats = np.empty(ns.value, np.float32)
        cnt = np.zeros(5, np.uint64)
        self.L.fo_fetch(self.h, qo.ctypes.data, recs.ctypes.data, so.ctypes.data, stats.ctypes.data,
                        cnt.ctypes.data)
        res = OverlapResult(qo, recs, so, stats, cnt)
        if params.partition_bad_mappings:
            res.needs_trim = np.zeros(nr.value, np.uint8)
            self.L.fo_fetch_trim(self.h, res.needs_trim.ctypes.data)
        if params.keep_alignment:
            nm = self.L.fo_fetch_matches(self.h, None, None)
            res.match_off = np.empty(nr.value + 1, np.uint64)
            res.matches = np.empty((nm, 2), np.int32)
            self.L.fo_fetch_matches(self.h, res.match_off.ctypes.data, res.matches.ctypes.data)
        return res


def match_hashes(match_off: np.ndarray, matches: np.ndarray) -> np.ndarray:
    """Order-sensitive 64-bit digest of each record's kmerMatches list:
    sum_i (i+1) * (cur_i * 0x9E3779B97F4A7C15 + ext_i + 1) mod 2^64.  The golden files
    store (count, digest) per overlap instead of the lists themselves."""
    off = match_off.astype(np.int64)
    n = len(off) - 1
    m = matches.astype(np.int64).astype(np.uint64)
    with np.errstate(over="ignore"):
        v = m[:, 0] * np.uint64(0x9E3779B97F4A7C15) + m[:, 1] + np.uint64(1)
        idx = np.arange(len(m), dtype=np.int64) - np.repeat(off[:-1], np.diff(off)) + 1
        v = v * idx.astype(np.uint64)
        cs = np.concatenate([[np.uint64(0)], np.cumsum(v, dtype=np.uint64)])
        return (cs[off[1:]] - cs[off[:-1]]) if n else np.zeros(0, np.uint64)


def edit_distance(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().fo_edit_distance(a.ctypes.data, len(a), b.ctypes.data, len(b))


def edit_distance_dp(a: np.ndarray, b: np.ndarray) -> int:
    """The plain scalar DP (slow; what the bit-vector form is checked against)."""
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().fo_edit_distance_dp(a.ctypes.data, len(a), b.ctypes.data, len(b))


def edit_distance_k0(a: np.ndarray, b: np.ndarray, k0: int) -> int:
    """The bit-vector form with the band doubling started at k0."""
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().fo_edit_distance_k0(a.ctypes.data, len(a), b.ctypes.data, len(b), int(k0))


def ref_edlib_distances(pairs):
    """Edit distances of (a, b) pairs of 0..3 arrays through the REFERENCE's edlib
    (oracle/_ref/ref_dumper --edlib-pairs)."""
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        for a, b in pairs:
            # a leading '^' keeps empty strings parseable as tokens
            f.write("^" + "".join("ACGT"[x] for x in a) + " ^" + "".join("ACGT"[x] for x in b) + "\n")
        path = f.name
    try:
        out = subprocess.run([REF_DUMPER, "--edlib-pairs", path], check=True, capture_output=True, text=True)
    finally:
        os.unlink(path)
    return [int(x) for x in out.stdout.split()]


def ksw_cigar(trg: np.ndarray, qry: np.ndarray):
    """getAlignmentCigarKsw(trg, qry) (alignment.cpp:102-216) by the oracle's restatement of ksw_extz2:
    (error-rate bit pattern as hex, CIGAR text "<len><op> ...")."""
    t = np.ascontiguousarray(trg, np.uint8)
    q = np.ascontiguousarray(qry, np.uint8)
    cap = len(t) + len(q) + 8
    ops = np.empty(cap, np.uint8)
    lens = np.empty(cap, np.int32)
    err = C.c_float()
    n = lib().fo_ksw_cigar(t.ctypes.data, len(t), q.ctypes.data, len(q), ops.ctypes.data, lens.ctypes.data, cap, C.byref(err))
    bits = int(np.array([err.value], np.float32).view(np.uint32)[0])
    return f"{bits:08x}", " ".join(f"{int(lens[i])}{chr(ops[i])}" for i in range(n))


def ref_ksw_cigars(pairs):
    """The same through the REFERENCE's getAlignmentCigarKsw (oracle/_ref/ref_dumper --ksw-pairs)."""
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        for a, b in pairs:
            f.write("^" + "".join("ACGT"[x] for x in a) + " ^" + "".join("ACGT"[x] for x in b) + "\n")
        path = f.name
    try:
        out = subprocess.run([REF_DUMPER, "--ksw-pairs", path], check=True, capture_output=True, text=True)
    finally:
        os.unlink(path)
    res = []
    for line in out.stdout.splitlines():
        t = line.split(" ", 1)
        res.append((t[0], t[1] if len(t) > 1 else ""))
    return res


def introsort_mismatches(keys: np.ndarray) -> int:
    keys = np.ascontiguousarray(keys, np.uint64)
    return lib().fo_introsort_mismatches(keys.ctypes.data, len(keys))


def std_sort_perm(keys: np.ndarray) -> np.ndarray:
    keys = np.ascontiguousarray(keys, np.uint64)
    out = np.empty(len(keys), np.uint32)
    lib().fo_std_sort_perm(keys.ctypes.data, len(keys), out.ctypes.data)
    return out


# --- the compiled reference ---------------------------------------------------
REF_DUMPER_GPU = os.path.join(_HERE, "_ref", "ref_dumper_gpu")


def have_ref_gpu() -> bool:
    """oracle/_ref/ref_dumper_gpu: the same dumper with the reference's getSeqOverlaps / index build
    replaced at link time by integration/flye_seam.cpp (the compiled Flye-side binding)."""
    return os.path.exists(REF_DUMPER_GPU)


def have_ref() -> bool:
    return os.path.exists(REF_DUMPER)


def run_ref(fasta, params_string=None, config=None, threads=8, min_read_len=0, max_overlaps=0,
            force_local=False, min_overlap=1000, div_mode="none", index_out=None, ovlp_out=None,
            query_limit=None, rc_queries=False, queries_fasta=None, only_max=None, max_overhang=None,
            nucl_aln=None, keep_aln=False, max_div=None, find_all=False, partition_bad=False, binary=None, env=None):
    cmd = [binary or REF_DUMPER, "--reads", fasta, "--threads", str(threads), "--min-read-len", str(min_read_len),
           "--max-overlaps", str(max_overlaps), "--force-local", str(int(force_local)),
           "--min-overlap", str(min_overlap), "--div-mode", div_mode]
    if params_string:
        cmd += ["--params", params_string]
    if config:
        cmd += ["--config", config]
    if index_out:
        cmd += ["--index-out", index_out]
    if ovlp_out:
        cmd += ["--ovlp-out", ovlp_out]
    if query_limit is not None:
        cmd += ["--query-limit", str(query_limit)]
    if rc_queries:
        cmd += ["--rc-queries"]
    if queries_fasta:
        cmd += ["--queries", queries_fasta]
    if only_max is not None:
        cmd += ["--only-max", str(int(only_max))]
    if max_overhang is not None:
        cmd += ["--max-overhang", str(int(max_overhang))]
    if nucl_aln is not None:
        cmd += ["--nucl-aln", str(int(nucl_aln))]
    if keep_aln:
        cmd += ["--keep-aln", "1"]
    if max_div is not None:
        cmd += ["--max-div", repr(float(max_div))]
    if find_all:
        cmd += ["--find-all"]
    if partition_bad:
        cmd += ["--partition-bad", "1"]
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, env=env)
    if os.environ.get("FLYE_REF_STDERR"):       # e.g. with FGB_TRACE=1: what the program wrote to stderr
        sys.stderr.write(out.stderr)
    return json.loads(out.stdout.strip().splitlines()[-1])


def parse_ref_index(path, rs):
    """ref_dumper --index-out -> IndexExport in (record<<32|pos) form."""
    rec_off = np.zeros(2 * rs.n + 1, np.uint64)
    ln = rs.length.astype(np.uint64)
    rec_off[1:] = np.cumsum(np.repeat(ln, 2))
    keys, off, ent, rep = [], [0], [], []
    header = None
    with open(path) as f:
        for line in f:
            t = line.split()
            if t[0] == "S":
                header = {t[i]: t[i + 1] for i in range(1, len(t), 2)}
            elif t[0] == "K":
                keys.append(int(t[1], 16))
                ent.extend(int(x) for x in t[3:])
                off.append(len(ent))
            elif t[0] == "R":
                rep.append(int(t[1], 16))
    g = np.array(ent, np.uint64)
    rec = np.searchsorted(rec_off, g, side="right").astype(np.uint64) - np.uint64(1)
    pos = g - rec_off[rec.astype(np.int64)]
    return header, IndexExport(np.array(keys, np.uint64), np.array(off, np.uint64),
                               (rec << np.uint64(32)) | pos, np.array(rep, np.uint64))


# --- the reference's own assemble stage -------------------------------------------------------
FLYE_ASSEMBLE = os.path.join(_HERE, "_ref", "flye_assemble")            # main_assemble.cpp:123-257, compiled unmodified
FLYE_ASSEMBLE_GPU = os.path.join(_HERE, "_ref", "flye_assemble_gpu")    # the same objects behind integration/flye_seam.o


def have_assemble() -> bool:
    return os.path.exists(FLYE_ASSEMBLE)


def have_assemble_gpu() -> bool:
    return os.path.exists(FLYE_ASSEMBLE_GPU)


def _log_stage_seconds(log_text):
    """wall-clock marks of the stage's log lines ([YYYY-MM-DD hh:mm:ss], one-second resolution)"""
    import datetime
    import re
    marks = {}
    for line in log_text.splitlines():
        m = re.match(r"\[(\d{4}-\d\d-\d\d \d\d:\d\d:\d\d)\] (\w+): (.*)", line)
        if m:
            t = datetime.datetime.strptime(m.group(1), "%Y-%m-%d %H:%M:%S").timestamp()
            for key, pat in (("read", "Reading sequences"), ("extend", "Extending reads"),
                             ("assembled", "Assembled "), ("consensus", "Generating sequence"), ("write", "Writing FASTA")):
                if m.group(3).startswith(pat) and key not in marks:
                    marks[key] = t
    return marks


def run_assemble(fasta, cfg_path, out_fasta, threads=1, min_ovlp=5000, binary=None, env=None, log_path=None):
    """`flye-modules assemble --reads .. --out-asm .. --config .. --threads .. --min-ovlp ..` (the argv
    flye/assembly/assemble.py:43-69 builds), by the pure reference program or the one with the device seams.
    Returns wall seconds, the log's stage marks and, for the seam program, what the seams did
    (FLYE_GPU_STATS, integration/flye_seam.cpp)."""
    import time
    log_path = log_path or out_fasta + ".log"
    stats_path = out_fasta + ".seam.json"
    e = dict(os.environ if env is None else env)
    e["FLYE_GPU_STATS"] = stats_path
    cmd = [binary or FLYE_ASSEMBLE, "--reads", fasta, "--out-asm", out_fasta, "--config", cfg_path, "--log", log_path,
           "--threads", str(threads), "--min-ovlp", str(min_ovlp), "--debug"]
    t = time.perf_counter()
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, env=e)
    wall = time.perf_counter() - t
    if os.environ.get("FLYE_REF_STDERR"):
        sys.stderr.write(out.stderr)
    info = {"wall_s": wall, "marks": _log_stage_seconds(open(log_path).read())}
    m = info["marks"]
    if "extend" in m and "consensus" in m:
        info["extend_s_log"] = m["consensus"] - m["extend"]     # Extender::assembleDisjointigs, to the second
    if os.path.exists(stats_path):
        info["seams"] = json.load(open(stats_path))
    return info


def ref_consensus(pairs, threads=1, binary=None):
    """ConsensusGenerator::generateConsensuses (consensus_generator.cpp:18-126) over two-read disjointigs, one
    per (left, right) pair of 0..3 arrays, by the compiled reference (or the program with the device seams):
    -> (stdout text: one "name sequence" line per pair, {"pairs", "threads", "consensus_s"})."""
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        for a, b in pairs:
            f.write("^" + "".join("ACGT"[x] for x in a) + " ^" + "".join("ACGT"[x] for x in b) + "\n")
        path = f.name
    try:
        from flye_amd import config
        out = subprocess.run([binary or REF_DUMPER, "--threads", str(threads), "--params", config.params_string("raw"),
                              "--consensus-pairs", path], check=True, capture_output=True, text=True)
    finally:
        os.unlink(path)
    if os.environ.get("FLYE_REF_STDERR"):
        sys.stderr.write(out.stderr)
    info = json.loads([l for l in out.stderr.strip().splitlines() if l.startswith("{")][-1])
    return out.stdout, info
