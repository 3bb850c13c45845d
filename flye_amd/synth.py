"""ctypes wrapper of the seeded read simulator (flye_amd/csrc/synth.cpp)."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Params(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("genomeLen", C.c_int64),
                ("nRepeatFamilies", C.c_int32), ("repeatMinLen", C.c_int32),
                ("repeatMaxLen", C.c_int32), ("repeatMinCopies", C.c_int32),
                ("repeatMaxCopies", C.c_int32), ("repeatDivPermille", C.c_int32),
                ("nHomopolymers", C.c_int32), ("nTandems", C.c_int32),
                ("targetBases", C.c_int64), ("lenModel", C.c_int32),
                ("medianLen", C.c_int32), ("sdPermille", C.c_int32),
                ("minLen", C.c_int32), ("maxLen", C.c_int32),
                ("errPermille10", C.c_int32), ("subPct", C.c_int32),
                ("insPct", C.c_int32), ("circular", C.c_int32), ("readSeed", C.c_uint64)]


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "lib", "libflyesynth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run __graft_entry__.build()")
        lib = C.CDLL(path)
        lib.fs_create.restype = C.c_void_p
        lib.fs_create.argtypes = [C.POINTER(_Params)]
        lib.fs_destroy.argtypes = [C.c_void_p]
        for f in ("fs_num_reads", "fs_num_words", "fs_total_bases"):
            getattr(lib, f).restype = C.c_int64
            getattr(lib, f).argtypes = [C.c_void_p]
        lib.fs_copy.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        lib.fs_write_fasta.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64]
        lib.fs_write_fasta.restype = C.c_int
        _LIB = lib
    return _LIB


@dataclass
class ReadSet:
    """Forward-strand reads, 2-bit packed, each read word-aligned."""
    words: np.ndarray      # uint64
    word_off: np.ndarray   # uint64, n+1
    length: np.ndarray     # int32, n
    origin: np.ndarray     # int64 template start in the genome
    strand: np.ndarray     # uint8
    total_bases: int

    @property
    def n(self) -> int:
        return int(self.length.shape[0])

    def subset(self, idx) -> "ReadSet":
        idx = np.asarray(idx, dtype=np.int64)
        nw = (self.word_off[idx + 1] - self.word_off[idx]).astype(np.int64)
        off = np.zeros(len(idx) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(nw)
        words = np.empty(int(off[-1]), dtype=np.uint64)
        for j, i in enumerate(idx):
            words[int(off[j]):int(off[j + 1])] = self.words[int(self.word_off[i]):int(self.word_off[i + 1])]
        return ReadSet(words, off, self.length[idx].copy(), self.origin[idx].copy(),
                       self.strand[idx].copy(), int(self.length[idx].sum()))

    def filter_min_len(self, min_len: int) -> "ReadSet":
        """Drop reads with length <= min_len (reference
        src/sequence/sequence_container.cpp:102: kept iff length > minReadLength)."""
        keep = np.nonzero(self.length > min_len)[0]
        return self if len(keep) == self.n else self.subset(keep)

    @staticmethod
    def from_arrays(seqs) -> "ReadSet":
        """Reads given as arrays of 0..3 (A, C, G, T), packed like DnaSequence (sequence.h:54-69)."""
        lens = np.array([len(x) for x in seqs], dtype=np.int32)
        nw = (lens.astype(np.int64) + 31) // 32
        off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(nw)
        words = np.zeros(int(off[-1]), dtype=np.uint64)
        sh = np.arange(32, dtype=np.uint64) * np.uint64(2)
        for i, x in enumerate(seqs):
            if not len(x):
                continue
            pad = np.zeros(int(nw[i]) * 32, dtype=np.uint64)
            pad[:len(x)] = np.asarray(x, dtype=np.uint64) & np.uint64(3)
            words[int(off[i]):int(off[i + 1])] = (pad.reshape(-1, 32) << sh[None, :]).sum(axis=1, dtype=np.uint64)
        n = len(seqs)
        return ReadSet(words, off, lens, np.zeros(n, np.int64), np.zeros(n, np.uint8), int(lens.sum()))

    def write_fasta(self, path: str) -> None:
        with open(path, "w") as f:
            for i in range(self.n):
                w = self.words[int(self.word_off[i]):int(self.word_off[i + 1])]
                n = int(self.length[i])
                sh = (np.arange(32, dtype=np.uint64) * np.uint64(2))
                b = ((w[:, None] >> sh[None, :]) & np.uint64(3)).reshape(-1)[:n]
                f.write(f">r{i}\n")
                f.write(np.array(list("ACGT"))[b.astype(np.int64)].astype("S1").tobytes().decode())
                f.write("\n")


# named workloads ------------------------------------------------------------
def simulate(seed=12345, genome_len=60_000, coverage=30, kind="pb_raw",
             n_repeat_families=4, repeat_len=(800, 3000), repeat_copies=(2, 5),
             repeat_div_permille=20, n_homopolymers=8, n_tandems=8,
             median_len=None, min_len=None, max_len=None, circular=1,
             fasta_path=None, read_seed=0) -> ReadSet:
    """kind: pb_raw (12 % error 15/40/45 sub/ins/del, log-normal sigma .5, median
    ~8.1 kb = e^9), ont_raw (10 % 25/25/50, sigma .8, median ~9.9 kb), hifi
    (0.5 % error, N(15 kb, 2 kb)) -- the read models of SURVEY.md §8(d).  ``read_seed`` != 0
    draws the reads from their own stream: same genome (``seed``), different reads."""
    if kind == "pb_raw":
        d = dict(lenModel=0, medianLen=8103, sdPermille=0, minLen=2000, maxLen=40000,
                 errPermille10=1200, subPct=15, insPct=40)
    elif kind == "ont_raw":
        d = dict(lenModel=1, medianLen=9897, sdPermille=0, minLen=1000, maxLen=150000,
                 errPermille10=1000, subPct=25, insPct=25)
    elif kind == "hifi":
        d = dict(lenModel=2, medianLen=15000, sdPermille=133, minLen=5000, maxLen=30000,
                 errPermille10=50, subPct=34, insPct=33)
    elif kind == "hifi03":
        d = dict(lenModel=2, medianLen=15000, sdPermille=133, minLen=5000, maxLen=30000,
                 errPermille10=30, subPct=34, insPct=33)
    else:
        raise ValueError(kind)
    if median_len is not None:
        d["medianLen"] = int(median_len)
    if min_len is not None:
        d["minLen"] = int(min_len)
    if max_len is not None:
        d["maxLen"] = int(max_len)
    p = _Params(seed=seed, genomeLen=genome_len, nRepeatFamilies=n_repeat_families,
                repeatMinLen=repeat_len[0], repeatMaxLen=repeat_len[1],
                repeatMinCopies=repeat_copies[0], repeatMaxCopies=repeat_copies[1],
                repeatDivPermille=repeat_div_permille, nHomopolymers=n_homopolymers,
                nTandems=n_tandems, targetBases=int(genome_len * coverage),
                circular=circular, readSeed=read_seed, **d)
    lib = _lib()
    h = lib.fs_create(C.byref(p))
    try:
        n = lib.fs_num_reads(h)
        nw = lib.fs_num_words(h)
        words = np.empty(nw, dtype=np.uint64)
        off = np.empty(n + 1, dtype=np.uint64)
        ln = np.empty(n, dtype=np.int32)
        org = np.empty(n, dtype=np.int64)
        st = np.empty(n, dtype=np.uint8)
        lib.fs_copy(h, words.ctypes.data, off.ctypes.data, ln.ctypes.data,
                    org.ctypes.data, st.ctypes.data)
        tb = lib.fs_total_bases(h)
        if fasta_path is not None:
            if lib.fs_write_fasta(h, fasta_path.encode(), 0, -1) != 0:
                raise OSError(f"cannot write {fasta_path}")
    finally:
        lib.fs_destroy(h)
    return ReadSet(words, off, ln, org, st, int(tb))
