"""ctypes binding of libflyegpu.so + a host-side mirror of the reference's
``VertexIndex`` / ``OverlapDetector`` / ``OverlapContainer`` interface.

The reference is C++ and has no FFI; the binding a Flye
maintainer would add is the C++ stub in INTEGRATION.md.  This Python mirror keeps
the same names, argument meaning and error behaviour so that the parity tests read
like calls into the reference:

* ``VertexIndex.countKmers / buildIndexUnevenCoverage / buildIndexMinimizers /
  clear / getSampleRate``  (reference src/sequence/vertex_index.h:213-218, :260)
* ``OverlapDetector(...)`` ctor arguments (src/sequence/overlap.h:313-336)
* ``OverlapContainer.quickSeqOverlaps / lazySeqOverlaps /
  estimateOverlaperParameters / setDivergenceThreshold``
  (src/sequence/overlap.cpp:518-574, :744-827)

There is no CPU fallback: if the HIP library or a device is missing every entry
point raises ``FlyeGpuError``.
"""
from __future__ import annotations

import ctypes as C
import os

import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLYE_GPU_LIB: another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("FLYE_GPU_LIB") or os.path.join(_HERE, "lib", "libflyegpu.so")

REC_DTYPE = np.dtype([("cur_id", "<u4"), ("ext_id", "<u4"), ("cur_begin", "<i4"),
                      ("cur_end", "<i4"), ("cur_len", "<i4"), ("ext_begin", "<i4"),
                      ("ext_end", "<i4"), ("ext_len", "<i4"), ("score", "<i4"),
                      ("seq_divergence", "<f4"), ("chain_length", "<i4"),
                      ("filtered_positions", "<i4"), ("edit_distance", "<i4"),
                      ("hpc_len_cur", "<i4"), ("hpc_len_ext", "<i4")])

ABI_SYMBOLS = ["fg_abi_version", "fg_create", "fg_destroy", "fg_strerror", "fg_last_error",
               "fg_container_info", "fg_set_reads", "fg_set_queries", "fg_build_index_solid", "fg_build_index_minimizers",
               "fg_index_begin_solid", "fg_index_begin_minimizers", "fg_index_build_range", "fg_index_finish",
               "fg_index_kmer_hist", "fg_index_count_slice", "fg_index_batch_freq", "fg_index_batch_select",
               "fg_index_selection_done", "fg_index_gather_begin", "fg_index_gather_end", "fg_memory_stats",
               "fg_import_index", "fg_index_device_arrays", "fg_clear_index", "fg_export_index", "fg_overlaps", "fg_release_batch",
               "fg_kernel_times", "fg_debug_sort_pairs", "fg_debug_edit_distances", "fg_align_cigar_ksw", "fg_release_cigars"]


class FlyeGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libflyegpu error {code}: {msg}")
        self.code = code


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


class OverlapBatch(C.Structure):
    _fields_ = [("n_queries", C.c_uint32), ("n_recs", C.c_uint64), ("query_off", C.c_void_p),
                ("recs", C.c_void_p), ("n_div_stats", C.c_uint64), ("div_stats_off", C.c_void_p),
                ("div_stats", C.c_void_p), ("n_matches", C.c_uint64), ("match_off", C.c_void_p),
                ("matches", C.c_void_p), ("needs_trim", C.c_void_p), ("query_bp", C.c_uint64), ("query_kmers", C.c_uint64),
                ("seed_hits", C.c_uint64), ("dp_groups", C.c_uint64), ("dp_elements", C.c_uint64),
                ("dp_elements_small", C.c_uint64),
                ("device_seconds", C.c_double), ("owner_", C.c_void_p)]


class CigarBatch(C.Structure):
    _fields_ = [("n_pairs", C.c_uint32), ("run_off", C.POINTER(C.c_uint64)), ("ops", C.POINTER(C.c_uint8)),
                ("lens", C.POINTER(C.c_int32)), ("err_rate", C.POINTER(C.c_float)), ("owner_", C.c_void_p)]


class BridgeStats(C.Structure):
    _fields_ = [("device_calls", C.c_uint64), ("reads_computed", C.c_uint64), ("requests", C.c_uint64),
                ("cache_hits", C.c_uint64), ("cached_overlaps", C.c_uint64), ("reads_ahead", C.c_uint64),
                ("ahead_hits", C.c_uint64)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("seconds", C.c_double), ("launches", C.c_uint64)]


_LIB = None


def load_library():
    """Load libflyegpu.so.  Raises if it is missing -- the product never falls
    back to a CPU path."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FlyeGpuError(-100, f"{LIB_PATH} not built: run __graft_entry__.build()")
        # One HIP runtime per process.  torch ships its own copy of libamdhip64 / libhsa-runtime64; a Python
        # process that uses both this library and torch.cuda (flye_amd/dist.py wraps context memory as torch
        # tensors for the collectives) must load torch's FIRST -- the other order leaves torch with "No HIP GPUs
        # are available" (measured: pytest process, library loaded before the first torch.cuda call).  A C / C++
        # consumer of the C ABI never sees torch.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.fg_abi_version.restype = C.c_int
        L.fg_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
        L.fg_destroy.argtypes = [C.c_void_p]
        L.fg_strerror.restype = C.c_char_p
        L.fg_strerror.argtypes = [C.c_int]
        L.fg_last_error.restype = C.c_char_p
        L.fg_last_error.argtypes = [C.c_void_p]
        L.fg_set_reads.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.fg_set_queries.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.fg_build_index_solid.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_float,
                                           C.c_float, C.POINTER(IndexStats)]
        L.fg_build_index_minimizers.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                                C.POINTER(IndexStats)]
        L.fg_index_begin_solid.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_float, C.c_void_p]
        L.fg_index_begin_minimizers.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p]
        L.fg_index_build_range.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.fg_index_finish.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(IndexStats)]
        L.fg_index_kmer_hist.argtypes = [C.c_void_p, C.c_void_p]
        L.fg_index_count_slice.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_float, C.c_uint32,
                                           C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.fg_index_batch_freq.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.fg_index_batch_select.argtypes = [C.c_void_p, C.c_uint32]
        L.fg_index_selection_done.argtypes = [C.c_void_p, C.c_void_p]
        L.fg_index_gather_begin.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fg_index_gather_end.argtypes = [C.c_void_p, C.c_float]
        L.fg_memory_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
        L.fg_import_index.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                      C.c_void_p, C.c_float, C.c_int]
        L.fg_index_device_arrays.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.fg_clear_index.argtypes = [C.c_void_p]
        L.fg_export_index.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.fg_overlaps.argtypes = [C.c_void_p, C.POINTER(DetectorParams), C.c_void_p, C.c_uint32,
                                  C.c_int32, C.c_uint8, C.POINTER(OverlapBatch)]
        L.fg_release_batch.argtypes = [C.POINTER(OverlapBatch)]
        L.fg_kernel_times.argtypes = [C.c_void_p, C.POINTER(KernelTime), C.c_int]
        L.fg_debug_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.fg_debug_edit_distances.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fg_align_cigar_ksw.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(CigarBatch)]
        L.fg_release_cigars.argtypes = [C.POINTER(CigarBatch)]
        # include/flye_gpu_bridge.h
        L.fgb_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(DetectorParams), C.c_uint32, C.c_uint32]
        L.fgb_destroy.argtypes = [C.c_void_p]
        L.fgb_lazy.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.fgb_quick.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_uint8, C.c_void_p, C.c_uint64,
                                C.POINTER(C.c_uint64)]
        L.fgb_prefetch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.fgb_set_divergence_threshold.argtypes = [C.c_void_p, C.c_float]
        L.fgb_divergence_stats.restype = C.c_uint64
        L.fgb_divergence_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.fgb_get_stats.argtypes = [C.c_void_p, C.POINTER(BridgeStats)]
        # on-disk text forms (host only)
        L.fg_overlap_dump.restype = C.c_int64
        L.fg_overlap_dump.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64]
        L.fg_overlap_load.argtypes = [C.c_char_p, C.c_void_p] + [C.POINTER(C.c_uint32)] * 4
        L.fg_alignment_dump.restype = C.c_int64
        L.fg_alignment_dump.argtypes = [C.c_int64, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64]
        L.fg_fasta_record.restype = C.c_int64
        L.fg_fasta_record.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_char_p, C.c_uint64]
        _LIB = L
    return _LIB


class IndexExport:
    def __init__(self, keys, key_off, entries, repetitive):
        self.keys, self.key_off, self.entries, self.repetitive = keys, key_off, entries, repetitive


class _Arena:
    """Owns one fg_overlap_batch; released when the last array view is gone."""

    def __init__(self, lib, batch):
        self.lib, self.batch = lib, batch

    def __del__(self):
        try:
            self.lib.fg_release_batch(C.byref(self.batch))
        except Exception:
            pass

    def view(self, ptr, ctype, count, dtype):
        if not count:
            return np.empty(0, dtype)
        buf = (ctype * count).from_address(ptr)
        buf._arena = self            # the numpy view keeps buf (its base), buf keeps the arena
        return np.frombuffer(buf, dtype=dtype)


class OverlapResult:
    """Flat result of one batched ``getSeqOverlaps`` call.  The arrays are zero-copy
    views of the library-owned arena; the arena is released when the last view dies."""

    def __init__(self, lib, query_ids, batch):
        arena = _Arena(lib, batch)
        b = batch
        nq = len(query_ids)
        self.query_ids = query_ids
        self.query_off = arena.view(b.query_off, C.c_uint64, nq + 1, np.uint64)
        self.stat_off = arena.view(b.div_stats_off, C.c_uint64, nq + 1, np.uint64)
        self.recs = arena.view(b.recs, C.c_uint8, b.n_recs * REC_DTYPE.itemsize, REC_DTYPE)
        self.stats = arena.view(b.div_stats, C.c_float, b.n_div_stats, np.float32)
        # keep_alignment: kmerMatches of recs[i] = matches[match_off[i]:match_off[i+1]] ((cur, ext) rows)
        self.match_off = self.matches = None
        if b.match_off:
            self.match_off = arena.view(b.match_off, C.c_uint64, b.n_recs + 1, np.uint64)
            self.matches = arena.view(b.matches, C.c_int32, 2 * b.n_matches, np.int32).reshape(-1, 2)
        # partition_bad_mappings: 1 = failed the divergence gate, there for the caller's checkIdyAndTrim
        self.needs_trim = arena.view(b.needs_trim, C.c_uint8, b.n_recs, np.uint8) if b.needs_trim else None
        self.query_bp, self.query_kmers = b.query_bp, b.query_kmers
        self.seed_hits, self.dp_groups, self.dp_elements = b.seed_hits, b.dp_groups, b.dp_elements
        self.dp_elements_small = b.dp_elements_small
        self.device_seconds = b.device_seconds

    def of(self, i):
        return self.recs[int(self.query_off[i]):int(self.query_off[i + 1])]

    def kmerMatches(self, i):
        """OverlapRange::kmerMatches of record i (keep_alignment detectors only)."""
        return self.matches[int(self.match_off[i]):int(self.match_off[i + 1])]

    def match_digests(self):
        """(count, order-sensitive 64-bit digest) per record:
        sum_j (j+1) * (cur_j * 0x9E3779B97F4A7C15 + ext_j + 1) mod 2^64."""
        off = self.match_off.astype(np.int64)
        m = self.matches.astype(np.int64).astype(np.uint64)
        with np.errstate(over="ignore"):
            v = m[:, 0] * np.uint64(0x9E3779B97F4A7C15) + m[:, 1] + np.uint64(1)
            j = np.arange(len(m), dtype=np.int64) - np.repeat(off[:-1], np.diff(off)) + 1
            cs = np.concatenate([[np.uint64(0)], np.cumsum(v * j.astype(np.uint64), dtype=np.uint64)])
            return np.diff(off), cs[off[1:]] - cs[off[:-1]]

    def lines(self):
        r = self.recs
        bits = r["seq_divergence"].view(np.uint32)
        out = [f"{r['cur_id'][i]} {r['cur_begin'][i]} {r['cur_end'][i]} {r['cur_len'][i]} "
               f"{r['ext_id'][i]} {r['ext_begin'][i]} {r['ext_end'][i]} {r['ext_len'][i]} "
               f"{r['score'][i]} {bits[i]:08x}" for i in range(len(r))]
        if self.match_off is not None:
            cnt, dg = self.match_digests()
            out = [f"{l} {cnt[i]} {int(dg[i]):016x}" for i, l in enumerate(out)]
        return out


def memory_stats(reset_peak=False):
    """(device bytes the library holds now, peak since the last reset) over all contexts of the process"""
    now, peak = C.c_uint64(), C.c_uint64()
    load_library().fg_memory_stats(C.byref(now), C.byref(peak), 1 if reset_peak else 0)
    return now.value, peak.value


class Context:
    """One fg_ctx: a SequenceContainer's reads resident in HBM on one GPU."""

    def __init__(self, kmer_size=17, device=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.fg_create(C.byref(h), device, kmer_size)
        if rc != 0:
            raise FlyeGpuError(rc, self.L.fg_strerror(rc).decode())
        self.h = h
        self.k = kmer_size
        self.first_id = 0
        self.n_reads = 0

    def _check(self, rc):
        if rc != 0:
            raise FlyeGpuError(rc, f"{self.L.fg_strerror(rc).decode()}: {self.L.fg_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.fg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_reads(self, rs, first_seq_id=0):
        self.rs = rs
        self.first_id = first_seq_id
        self.n_reads = rs.n
        self._check(self.L.fg_set_reads(self.h, rs.n, rs.words.ctypes.data, rs.word_off.ctypes.data,
                                        rs.length.ctypes.data, first_seq_id))

    def set_queries(self, rs, first_seq_id):
        """Second container holding the queries (reads vs graph edges, read_aligner.cpp:178-217)."""
        self.qrs = rs
        self._check(self.L.fg_set_queries(self.h, rs.n, rs.words.ctypes.data, rs.word_off.ctypes.data,
                                          rs.length.ctypes.data, first_seq_id))

    def debug_sort_pairs(self, keys, seg_off):
        """Device hit-sort kernel on independent segments; returns (sorted keys, permutation)."""
        k = np.ascontiguousarray(keys, np.uint64).copy()
        off = np.ascontiguousarray(seg_off, np.uint64)
        v = np.empty(len(k), np.uint32)
        for i in range(len(off) - 1):
            a, b = int(off[i]), int(off[i + 1])
            v[a:b] = np.arange(b - a, dtype=np.uint32)
        self._check(self.L.fg_debug_sort_pairs(self.h, k.ctypes.data, v.ctypes.data, off.ctypes.data,
                                               len(off) - 1))
        return k, v

    def debug_edit_distances(self, n_pairs, use_hpc=False):
        """Device edit-distance kernels on the pairs (read 2i, read 2i+1) of the container;
        returns (distances, lengths of A, lengths of B)."""
        d = np.empty(n_pairs, np.int32)
        la = np.empty(n_pairs, np.int32)
        lb = np.empty(n_pairs, np.int32)
        self._check(self.L.fg_debug_edit_distances(self.h, n_pairs, int(bool(use_hpc)), d.ctypes.data,
                                                   la.ctypes.data, lb.ctypes.data))
        return d, la, lb

    def align_cigar_ksw(self, pairs):
        """getAlignmentCigarKsw (alignment.cpp:102-216) of (target, query) pairs of 0..3 arrays on the device:
        list of (error-rate bit pattern as hex, CIGAR text "<len><op> ...")."""
        n = len(pairs)
        trg = np.concatenate([np.asarray(a, np.uint8) for a, _ in pairs]) if n else np.empty(0, np.uint8)
        qry = np.concatenate([np.asarray(b, np.uint8) for _, b in pairs]) if n else np.empty(0, np.uint8)
        toff = np.zeros(n + 1, np.uint64)
        qoff = np.zeros(n + 1, np.uint64)
        toff[1:] = np.cumsum([len(a) for a, _ in pairs])
        qoff[1:] = np.cumsum([len(b) for _, b in pairs])
        trg = np.ascontiguousarray(trg if len(trg) else np.zeros(1, np.uint8))
        qry = np.ascontiguousarray(qry if len(qry) else np.zeros(1, np.uint8))
        b = CigarBatch()
        t0 = time.perf_counter()
        self._check(self.L.fg_align_cigar_ksw(self.h, n, trg.ctypes.data, toff.ctypes.data, qry.ctypes.data,
                                              qoff.ctypes.data, C.byref(b)))
        self.last_align_seconds = time.perf_counter() - t0      # the C call alone (the text below is test harness)
        out = []
        for i in range(n):
            a0, a1 = int(b.run_off[i]), int(b.run_off[i + 1])
            bits = int(np.array([b.err_rate[i]], np.float32).view(np.uint32)[0])
            out.append((f"{bits:08x}", " ".join(f"{b.lens[k]}{chr(b.ops[k])}" for k in range(a0, a1))))
        self.L.fg_release_cigars(C.byref(b))
        return out

    def kernel_times(self):
        arr = (KernelTime * 64)()
        n = self.L.fg_kernel_times(self.h, arr, 64)
        return {arr[i].name.decode(): (arr[i].seconds, arr[i].launches) for i in range(min(n, 64))}


class VertexIndex:
    """Mirror of reference VertexIndex (src/sequence/vertex_index.h:66-300)."""

    def __init__(self, ctx: Context, sample_rate: float):
        self.ctx = ctx
        self._sample_rate_init = float(sample_rate)
        self._counted = False
        self.stats = None

    def countKmers(self):
        """vertex_index.cpp:19-22.  Counting runs fused with the build on the device;
        this only records that the caller asked for it (k > 17 raises like :504-507)."""
        if self.ctx.k > 17:
            raise FlyeGpuError(-6, "Can't use flat counter for k-mer size > 17")
        self._counted = True

    def buildIndexUnevenCoverage(self, globalMinFreq: int, selectRate: float, tandemFreq: int,
                                 repeat_kmer_rate: float = 100.0):
        if not self._counted:
            raise FlyeGpuError(-4, "countKmers() must be called first")
        st = IndexStats()
        L = self.ctx.L
        self.ctx._check(L.fg_build_index_solid(self.ctx.h, globalMinFreq, selectRate, tandemFreq,
                                               repeat_kmer_rate, self._sample_rate_init, C.byref(st)))
        self.stats = st.as_dict()
        return self.stats

    def buildIndexMinimizers(self, minCoverage: int, wndLen: int, repeat_kmer_rate: float = 100.0):
        st = IndexStats()
        L = self.ctx.L
        self.ctx._check(L.fg_build_index_minimizers(self.ctx.h, minCoverage, wndLen, repeat_kmer_rate,
                                                    C.byref(st)))
        self.stats = st.as_dict()
        return self.stats

    def build(self, cfg: dict):
        """Index build exactly as main_assemble.cpp:195-223 selects it."""
        if cfg["use_minimizers"]:
            return self.buildIndexMinimizers(1, int(cfg["minimizer_window"]), cfg["repeat_kmer_rate"])
        self.countKmers()
        return self.buildIndexUnevenCoverage(2, cfg["meta_read_top_kmer_rate"],
                                             int(cfg["meta_read_filter_kmer_freq"]),
                                             cfg["repeat_kmer_rate"])

    # ---- the build in steps (sharded over GPUs: flye_amd/dist.py) -------------------------------------
    INDEX_BINS = 4096

    def begin(self, cfg: dict) -> np.ndarray:
        """Selection step of the build main_assemble.cpp:195-223 selects; returns the number of accepted
        k-mer positions per key bin."""
        hist = np.zeros(self.INDEX_BINS, np.uint64)
        L, h = self.ctx.L, self.ctx.h
        if cfg["use_minimizers"]:
            self.ctx._check(L.fg_index_begin_minimizers(h, 1, int(cfg["minimizer_window"]), cfg["repeat_kmer_rate"],
                                                        hist.ctypes.data))
        else:
            self.countKmers()
            self.ctx._check(L.fg_index_begin_solid(h, 2, cfg["meta_read_top_kmer_rate"],
                                                   int(cfg["meta_read_filter_kmer_freq"]), cfg["repeat_kmer_rate"],
                                                   self._sample_rate_init, hist.ctypes.data))
        return hist

    # solid-mode selection in steps of its own (counters of a key range only; batches of reads)
    def kmer_hist(self) -> np.ndarray:
        """ALL k-mer positions per key bin (what the ranks' key ranges are balanced on)."""
        hist = np.zeros(self.INDEX_BINS, np.uint64)
        self.ctx._check(self.ctx.L.fg_index_kmer_hist(self.ctx.h, hist.ctypes.data))
        return hist

    def count_slice(self, cfg: dict, bin_lo: int, bin_hi: int):
        """-> (distinct canonical k-mers of the range, number of read batches)"""
        self.countKmers()
        d, nb = C.c_uint64(), C.c_uint32()
        self.ctx._check(self.ctx.L.fg_index_count_slice(self.ctx.h, 2, cfg["meta_read_top_kmer_rate"],
                                                        int(cfg["meta_read_filter_kmer_freq"]), cfg["repeat_kmer_rate"],
                                                        self._sample_rate_init, int(bin_lo), int(bin_hi), C.byref(d), C.byref(nb)))
        return d.value, nb.value

    def batch_freq(self, batch: int):
        """-> (device pointer of the batch's uint32 frequencies, their number)"""
        p, n = C.c_void_p(), C.c_uint64()
        self.ctx._check(self.ctx.L.fg_index_batch_freq(self.ctx.h, int(batch), C.byref(p), C.byref(n)))
        return p.value or 0, n.value

    def batch_select(self, batch: int):
        self.ctx._check(self.ctx.L.fg_index_batch_select(self.ctx.h, int(batch)))

    def selection_done(self) -> np.ndarray:
        hist = np.zeros(self.INDEX_BINS, np.uint64)
        self.ctx._check(self.ctx.L.fg_index_selection_done(self.ctx.h, hist.ctypes.data))
        return hist

    def gather_begin(self, n_keys: int, n_entries: int, n_rep: int):
        """-> (full array device pointers [keys, key_off, entries, repetitive], the own piece's, piece sizes
        (keys, entries, repetitive))"""
        full = (C.c_void_p * 4)()
        piece = (C.c_void_p * 4)()
        sizes = (C.c_uint64 * 3)()
        self.ctx._check(self.ctx.L.fg_index_gather_begin(self.ctx.h, int(n_keys), int(n_entries), int(n_rep), full, piece, sizes))
        return [x or 0 for x in full], [x or 0 for x in piece], [int(x) for x in sizes]

    def gather_end(self, sample_rate: float):
        self.ctx._check(self.ctx.L.fg_index_gather_end(self.ctx.h, float(sample_rate)))

    def build_range(self, bin_lo: int, bin_hi: int) -> np.ndarray:
        sums = np.zeros(2, np.uint64)
        self.ctx._check(self.ctx.L.fg_index_build_range(self.ctx.h, int(bin_lo), int(bin_hi), sums.ctypes.data))
        return sums

    def finish(self, total_sums=None):
        st = IndexStats()
        ts = None if total_sums is None else np.ascontiguousarray(total_sums, np.uint64)
        self.ctx._check(self.ctx.L.fg_index_finish(self.ctx.h, None if ts is None else ts.ctypes.data, C.byref(st)))
        self.stats = st.as_dict()
        return self.stats

    def import_index(self, ex: "IndexExport", sample_rate: float, on_device=False, ptrs=None):
        """fg_import_index from host arrays (``ex``) or device pointers (``ptrs`` = keys, key_off, entries,
        repetitive as integers, with the counts in ``ex`` = (n_keys, n_entries, n_rep))."""
        L, h = self.ctx.L, self.ctx.h
        if on_device:
            nk, ne, nr = ex
            self.ctx._check(L.fg_import_index(h, nk, ptrs[0], ptrs[1], ne, ptrs[2], nr, ptrs[3], float(sample_rate), 1))
        else:
            k = np.ascontiguousarray(ex.keys, np.uint64)
            o = np.ascontiguousarray(ex.key_off, np.uint64)
            e = np.ascontiguousarray(ex.entries, np.uint64)
            r = np.ascontiguousarray(ex.repetitive, np.uint64)
            self.ctx._check(L.fg_import_index(h, len(k), k.ctypes.data, o.ctypes.data, len(e), e.ctypes.data, len(r),
                                              r.ctypes.data, float(sample_rate), 0))
            nk, ne, nr = len(k), len(e), len(r)
        self.stats = dict(self.stats or {}, selected_kmers=nk, index_entries=ne, repetitive_kmers=nr,
                          sample_rate=float(np.float32(sample_rate)))

    def device_arrays(self):
        """(n_keys, n_entries, n_rep), (keys, key_off, entries, repetitive) device pointers of the built index."""
        n = [C.c_uint64() for _ in range(3)]
        p = [C.c_void_p() for _ in range(4)]
        self.ctx._check(self.ctx.L.fg_index_device_arrays(self.ctx.h, *[C.byref(x) for x in n], *[C.byref(x) for x in p]))
        return tuple(x.value for x in n), tuple(x.value or 0 for x in p)

    def clear(self):
        self.ctx._check(self.ctx.L.fg_clear_index(self.ctx.h))

    def getSampleRate(self) -> float:
        return self.stats["sample_rate"] if self.stats else self._sample_rate_init

    def export(self) -> IndexExport:
        L, h = self.ctx.L, self.ctx.h
        nk, ne, nr = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.ctx._check(L.fg_export_index(h, C.byref(nk), C.byref(ne), C.byref(nr), None, None, None, None))
        keys = np.empty(nk.value, np.uint64)
        off = np.empty(nk.value + 1, np.uint64)
        ent = np.empty(ne.value, np.uint64)
        rep = np.empty(nr.value, np.uint64)
        self.ctx._check(L.fg_export_index(h, C.byref(nk), C.byref(ne), C.byref(nr), keys.ctypes.data,
                                          off.ctypes.data, ent.ctypes.data, rep.ctypes.data))
        return IndexExport(keys, off, ent, rep)


class OverlapDetector:
    """Ctor arguments of reference OverlapDetector (overlap.h:313-336)."""

    def __init__(self, ctx: Context, vertexIndex: VertexIndex, maxJump: int, minOverlap: int,
                 maxOverhang: int, keepAlignment: bool, onlyMaxExt: bool, maxDivergence: float,
                 nuclAlignment: bool, partitionBadMappings: bool, useHpc: bool):
        self.ctx, self.index = ctx, vertexIndex
        self.p = DetectorParams(max_jump=maxJump, min_overlap=minOverlap, max_overhang=maxOverhang,
                                keep_alignment=int(keepAlignment), only_max_ext=int(onlyMaxExt),
                                nucl_alignment=int(nuclAlignment),
                                partition_bad_mappings=int(partitionBadMappings), use_hpc=int(useHpc),
                                max_divergence=float(maxDivergence))

    @classmethod
    def for_assemble(cls, ctx, index, cfg, min_overlap=1000):
        """The detector main_assemble.cpp:229-238 builds."""
        return cls(ctx, index, int(cfg["maximum_jump"]), min_overlap, int(cfg["maximum_overhang"]),
                   False, True, 1.0, bool(cfg["reads_base_alignment"]), False,
                   bool(cfg["hpc_scoring_on"]))

    def getSeqOverlapsBatch(self, query_ids, forceLocal=False, maxOverlaps=0) -> OverlapResult:
        q = np.ascontiguousarray(query_ids, dtype=np.uint32)
        b = OverlapBatch()
        L = self.ctx.L
        self.ctx._check(L.fg_overlaps(self.ctx.h, C.byref(self.p), q.ctypes.data, len(q), maxOverlaps,
                                      int(bool(forceLocal)), C.byref(b)))
        return OverlapResult(L, q, b)


def seq_name(read_names, first_id, rec_id) -> str:
    """SequenceContainer::seqName: '+' / '-' + the FASTA header (sequence_container.cpp:62, :75)."""
    i = int(rec_id) - int(first_id)
    return ("-" if i & 1 else "+") + read_names[i >> 1]


def dump_overlaps(recs: np.ndarray, cur_name, ext_name, edge_ids=None):
    """OverlapRange::dump (overlap.h:227-236) of every record; ``cur_name`` / ``ext_name`` map a record id
    to its container name.  With ``edge_ids`` the lines take ReadAligner::storeAlignments' form
    (read_aligner.cpp:333-335)."""
    L = load_library()
    buf = C.create_string_buffer(1024)
    out = []
    recs = np.ascontiguousarray(recs)
    for i in range(len(recs)):
        r = recs[i:i + 1]
        cn, en = cur_name(r["cur_id"][0]).encode(), ext_name(r["ext_id"][0]).encode()
        if edge_ids is None:
            n = L.fg_overlap_dump(r.ctypes.data, cn, en, buf, len(buf))
        else:
            n = L.fg_alignment_dump(int(edge_ids[i]), r.ctypes.data, cn, en, buf, len(buf))
        if n < 0:
            raise FlyeGpuError(int(n), "fg_overlap_dump")
        if n >= len(buf):
            buf = C.create_string_buffer(int(n) + 1)
            continue_n = (L.fg_overlap_dump(r.ctypes.data, cn, en, buf, len(buf)) if edge_ids is None else
                          L.fg_alignment_dump(int(edge_ids[i]), r.ctypes.data, cn, en, buf, len(buf)))
            assert continue_n == n
        out.append(buf.value.decode())
    return out


def load_overlaps(lines, cur_id_of, ext_id_of) -> np.ndarray:
    """OverlapRange::load (overlap.h:238-251): ids through the caller's recordByName lookups."""
    L = load_library()
    recs = np.zeros(len(lines), REC_DTYPE)
    offs = [C.c_uint32() for _ in range(4)]
    for i, line in enumerate(lines):
        raw = line.encode()
        rc = L.fg_overlap_load(raw, recs[i:i + 1].ctypes.data, *[C.byref(o) for o in offs])
        if rc != 0:
            raise FlyeGpuError(rc, f"malformed overlap line {i}")
        recs["cur_id"][i] = cur_id_of(raw[offs[0].value:offs[0].value + offs[1].value].decode())
        recs["ext_id"][i] = ext_id_of(raw[offs[2].value:offs[2].value + offs[3].value].decode())
    return recs


def fasta_text(rs, names) -> str:
    """SequenceContainer::writeFasta(records, file, onlyPositiveStrand = true) (sequence_container.cpp:330-357)."""
    L = load_library()
    parts = []
    for i in range(rs.n):
        w = np.ascontiguousarray(rs.words[int(rs.word_off[i]):int(rs.word_off[i + 1])])
        n = int(rs.length[i])
        cap = n + n // 80 + len(names[i]) + 8
        buf = C.create_string_buffer(cap)
        got = L.fg_fasta_record(names[i].encode(), w.ctypes.data, n, buf, cap)
        assert 0 <= got <= cap
        parts.append(buf.raw[:got].decode())
    return "".join(parts)


def complement(recs: np.ndarray) -> np.ndarray:
    """OverlapRange::complement (overlap.h:118-147) on a record array."""
    out = recs.copy()
    out["cur_begin"] = recs["cur_len"] - recs["cur_end"] - 1
    out["cur_end"] = recs["cur_len"] - recs["cur_begin"] - 1
    out["ext_begin"] = recs["ext_len"] - recs["ext_end"] - 1
    out["ext_end"] = recs["ext_len"] - recs["ext_begin"] - 1
    out["cur_id"] = recs["cur_id"] ^ 1
    out["ext_id"] = recs["ext_id"] ^ 1
    return out


class OverlapContainer:
    """Mirror of reference OverlapContainer's query side (overlap.cpp:510-827).

    Flye's worker threads ask for one read at a time; the device wants batches.
    ``prefetch`` computes and caches a batch, ``lazySeqOverlaps`` serves single reads
    from the cache (computing a batch of one on a miss)."""

    def __init__(self, ovlpDetect: OverlapDetector, first_id=None, n_reads=None):
        self.det = ovlpDetect
        self.ctx = ovlpDetect.ctx
        self._cache = {}
        self._mean_true_ovlp_div = 0.0
        self.divergence_stats = []

    def quickSeqOverlaps(self, readId: int, maxOverlaps: int = 0, forceLocal: bool = False):
        res = self.det.getSeqOverlapsBatch([readId], forceLocal, maxOverlaps)
        self.divergence_stats.extend(res.stats.tolist())
        return res.recs.copy()

    def prefetch(self, readIds):
        fwd = sorted({int(r) & ~1 for r in readIds if (int(r) & ~1) not in self._cache})
        if not fwd:
            return
        res = self.det.getSeqOverlapsBatch(fwd, False, 0)
        self.divergence_stats.extend(res.stats.tolist())
        for i, rid in enumerate(fwd):
            f = res.of(i).copy()
            self._cache[rid] = (f, complement(f))

    def lazySeqOverlaps(self, readId: int):
        fid = int(readId) & ~1
        if fid not in self._cache:
            self.prefetch([fid])
        f, r = self._cache[fid]
        return f if (int(readId) & 1) == 0 else r

    def indexSize(self):
        return sum(len(f) for f, _ in self._cache.values())

    def estimateOverlaperParameters(self, libc_rand=None):
        """overlap.cpp:744-817: median over 1000 rand()-picked records of the divergence of
        each record's longest overlap.  ``libc_rand`` defaults to glibc's rand() so the
        draw sequence is the reference's."""
        if libc_rand is None:
            libc = C.CDLL(None)
            libc.rand.restype = C.c_int
            libc_rand = libc.rand
        n_records = 2 * self.ctx.n_reads
        picks = [self.ctx.first_id + (libc_rand() % n_records) for _ in range(1000)]
        res = self.det.getSeqOverlapsBatch(picks, False, 0)
        divs = []
        for i in range(len(picks)):
            o = res.of(i)
            if len(o):
                rng = o["cur_end"] - o["cur_begin"]
                divs.append(o["seq_divergence"][int(np.argmax(rng))])  # first maximum, strict >
        if divs:
            v = sorted(divs)
            self._mean_true_ovlp_div = float(np.float32(v[len(v) // 2]))  # utils.h median()
            self.divergence_stats = []
        else:
            self._mean_true_ovlp_div = 0.5
        return self._mean_true_ovlp_div

    def setDivergenceThreshold(self, threshold: float, isRelative: bool):
        """overlap.cpp:820-827"""
        base = np.float32(self._mean_true_ovlp_div) if isRelative else np.float32(0.0)
        self.det.p.max_divergence = float(base + np.float32(threshold))
        return self.det.p.max_divergence


class BatchingOverlapContainer:
    """Binding of include/flye_gpu_bridge.h: the native batch scheduler that turns the
    one-read-at-a-time calls of Flye's worker threads (overlap.cpp:518-574) into device
    batches.  Every method may be called from any number of Python threads (ctypes releases
    the GIL for the duration of a call); the context must not be used directly meanwhile."""

    def __init__(self, det: OverlapDetector, max_batch=4096, linger_us=200):
        self.ctx = det.ctx
        self.L = det.ctx.L
        h = C.c_void_p()
        self.ctx._check(self.L.fgb_create(C.byref(h), self.ctx.h, C.byref(det.p), max_batch, linger_us))
        self.h = h

    def close(self):
        if self.h:
            self.L.fgb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lazySeqOverlaps(self, readId: int) -> np.ndarray:
        p, n = C.c_void_p(), C.c_uint64()
        self.ctx._check(self.L.fgb_lazy(self.h, int(readId), C.byref(p), C.byref(n)))
        if not n.value:
            return np.empty(0, REC_DTYPE)
        buf = (C.c_uint8 * (n.value * REC_DTYPE.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=REC_DTYPE).copy()

    def quickSeqOverlaps(self, readId: int, maxOverlaps: int = 0, forceLocal: bool = False) -> np.ndarray:
        n = C.c_uint64()
        cap = 4096
        while True:
            out = np.empty(cap, REC_DTYPE)
            self.ctx._check(self.L.fgb_quick(self.h, int(readId), maxOverlaps, int(bool(forceLocal)),
                                             out.ctypes.data, cap, C.byref(n)))
            if n.value <= cap:
                return out[:n.value]
            cap = int(n.value)

    def prefetch(self, readIds):
        q = np.ascontiguousarray(readIds, dtype=np.uint32)
        self.ctx._check(self.L.fgb_prefetch(self.h, q.ctypes.data, len(q)))

    def setDivergenceThreshold(self, maxDivergence: float):
        self.ctx._check(self.L.fgb_set_divergence_threshold(self.h, float(maxDivergence)))

    def divergenceStats(self) -> np.ndarray:
        n = self.L.fgb_divergence_stats(self.h, None, 0)
        out = np.empty(n, np.float32)
        self.L.fgb_divergence_stats(self.h, out.ctypes.data, n)
        return out

    def stats(self) -> dict:
        st = BridgeStats()
        self.L.fgb_get_stats(self.h, C.byref(st))
        return {k: int(getattr(st, k)) for k, _ in st._fields_}
