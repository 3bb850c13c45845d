/* flye_gpu.h -- C ABI of the MI355X-native overlap hot path (libflyegpu.so).
 *
 * Drop-in boundary for the two seams of Flye 2.8.1's sequence library
 * (citations are into the reference tree):
 *
 *   index build        VertexIndex::countKmers / buildIndexUnevenCoverage /
 *                      buildIndexMinimizers / clear / getSampleRate
 *                      (src/sequence/vertex_index.h:213-218, :260;
 *                       src/sequence/vertex_index.cpp:19-125, :389-483)
 *   per-read overlaps  OverlapDetector::getSeqOverlaps, reached only through
 *                      OverlapContainer::quickSeqOverlaps / lazySeqOverlaps
 *                      (src/sequence/overlap.h:338-345, overlap.cpp:99-508,
 *                       :518-574)
 *
 * The reference has no FFI: these entry points are what a C++ binding inside
 * VertexIndex / OverlapContainer would call (see INTEGRATION.md for the stub).
 * Conventions: every call returns an int status (FG_OK = 0, < 0 = error, text
 * via fg_strerror); no exceptions cross the boundary; plain pointers + sizes;
 * a context is bound to one HIP device and may be used by one host thread at
 * a time.  There is NO CPU fallback: without a usable HIP device fg_create
 * fails with FG_ERR_NO_DEVICE.
 */
#ifndef FLYE_GPU_H
#define FLYE_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FG_ABI_VERSION 4

enum {
	FG_OK = 0,
	FG_ERR_NO_DEVICE = -1,   /* no HIP device / HIP runtime failure at create */
	FG_ERR_HIP = -2,         /* a HIP call failed; fg_last_error() has the text */
	FG_ERR_ARG = -3,         /* bad argument */
	FG_ERR_STATE = -4,       /* call order: reads not set / index not built */
	FG_ERR_KMER_TOO_FREQUENT = -5, /* vertex_index.cpp:372 "k-mer is too frequent" */
	FG_ERR_KMER_SIZE = -6,   /* k > 17 with the flat counter, vertex_index.cpp:504-507;
	                            k > 32 never fits Kmer::KmerRepr (kmer.h:19) */
	FG_ERR_UNSUPPORTED = -7, /* flag combination not built yet */
	FG_ERR_NOMEM = -8
};

typedef struct fg_ctx fg_ctx;

/* Parameters::get().kmerSize (src/common/config.h:103-115) is process-global
 * in the reference; here it is a property of the context.
 * The FIRST fg_create of a process initialises the HIP runtime, which draws from libc's rand() stream; the
 * call parks the caller's stream in a private state array for that time (initstate / setstate) so that the
 * reference's own rand() consumers (overlap.cpp:752-756, chimera.cpp:76, sequence_container.cpp:318-328) see
 * the numbers they would have seen.  That swap is process-wide: make the first fg_create from a thread
 * beside which no other thread calls rand() (Flye: the main thread, at index build).  Later calls swap nothing. */
int  fg_create(fg_ctx** out, int device, int kmer_size);
void fg_destroy(fg_ctx* ctx);
const char* fg_strerror(int code);
const char* fg_last_error(const fg_ctx* ctx);
int  fg_abi_version(void);

/* Id range of the indexed container and of the optional query container (n_fwd forward records
 * each, ids first_id + 2i and their reverse complements + 1); any pointer may be NULL. */
int fg_container_info(const fg_ctx* ctx, uint32_t* first_id, uint32_t* n_fwd,
                      uint32_t* query_first_id, uint32_t* query_n_fwd);

/* SequenceContainer contents (src/sequence/sequence_container.cpp:48-79,
 * :359-392).  Forward strands only; the reverse complement of read i is
 * implied (FastaRecord::Id first_seq_id + 2i is the forward record, +1 its
 * reverse complement, sequence_container.h:27-33).  Packing is DnaSequence's
 * (src/sequence/sequence.h:54-69): 32 nt per uint64, nt j at bits (j%32)*2,
 * A,C,G,T = 0..3; each read starts on a word boundary; word_off has n+1
 * entries.  N-replacement (sequence_container.cpp:318-328) stays with the
 * caller because it draws from libc rand().  The buffers are copied to HBM;
 * the caller may free them on return. */
int fg_set_reads(fg_ctx* ctx, uint32_t n_fwd, const uint64_t* words,
                 const uint64_t* word_off, const int32_t* len,
                 uint32_t first_seq_id);

/* Optional second SequenceContainer holding the QUERIES, for callers whose queries are not
 * the indexed sequences: ReadAligner::alignReads indexes the graph edge sequences and
 * queries every read against them (src/repeat_graph/read_aligner.cpp:178-217).  Same
 * layout as fg_set_reads; the ids must not overlap the indexed container's (the reference
 * draws both from one process-wide counter, sequence_container.cpp:16, :55-60).  n_fwd = 0
 * returns to "queries are the indexed reads".  fg_set_reads() also resets it. */
int fg_set_queries(fg_ctx* ctx, uint32_t n_fwd, const uint64_t* words,
                   const uint64_t* word_off, const int32_t* len,
                   uint32_t first_seq_id);

struct fg_index_stats {
	uint64_t total_kmers;      /* KmerCounter::_numKmers: distinct canonical k-mers
	                              ("Total k-mers", vertex_index.cpp:589); 0 in
	                              minimizer mode */
	uint64_t selected_kmers;   /* _kmerIndex.size() ("Selected k-mers", :121, :473) */
	uint64_t index_entries;    /* "Index size" (:122) / "K-mer index size" (:474) */
	uint64_t repetitive_kmers; /* _repetitiveKmers.size() */
	uint64_t repetitive_frequency; /* _repetitiveFrequency (:186) */
	float    mean_frequency;   /* meanFrequency (:185) */
	float    sample_rate;      /* VertexIndex::getSampleRate(): ctor value, or
	                              totalLen/entries after buildIndexMinimizers (:480-482) */
	double   build_seconds;    /* device time of the build, HIP events */
};

/* countKmers() + buildIndexUnevenCoverage(min_freq, select_rate, tandem_freq)
 * with Config "repeat_kmer_rate" = repeat_rate (vertex_index.cpp:19-125,
 * :173-212, :316-358, :499-590).  sample_rate_init is the VertexIndex ctor
 * argument (main_assemble.cpp:195-196). */
int fg_build_index_solid(fg_ctx* ctx, int32_t min_freq, float select_rate,
                         int32_t tandem_freq, float repeat_rate,
                         float sample_rate_init, struct fg_index_stats* out);

/* buildIndexMinimizers(min_coverage, window) (vertex_index.cpp:389-483,
 * kmer.h:206-262). */
int fg_build_index_minimizers(fg_ctx* ctx, int32_t min_coverage, int32_t window,
                              float repeat_rate, struct fg_index_stats* out);

/* The same builds in steps, for sharding over GPUs and for bounded memory:
 *   begin        the k-mer selection over ALL reads of the container (it needs every read: exact counts,
 *                per-read frequency thresholds / minimizers); hist[FG_INDEX_BINS] receives the number of
 *                accepted k-mer positions per key bin (bin = canonical k-mer >> max(0, 2k - 12));
 *   build_range  sorts and run-length encodes the keys of bins [bin_lo, bin_hi) -- each rank its own
 *                range; sums[2] = this context's running share of filterFrequentKmers' two integer sums
 *                (vertex_index.cpp:175-184), to be added up over the ranks;
 *   finish       total_sums (NULL = this context's own) fix the repetitive frequency; the context then
 *                holds a usable index over the ranges it built (fg_export_index / fg_index_device_arrays
 *                give the pieces; the all-gathered concatenation goes into fg_import_index).
 * fg_build_index_solid / _minimizers = begin + build_range(0, FG_INDEX_BINS) + finish(NULL). */
#define FG_INDEX_BINS 4096
int fg_index_begin_solid(fg_ctx* ctx, int32_t min_freq, float select_rate, int32_t tandem_freq,
                         float repeat_rate, float sample_rate_init, uint64_t* hist);
int fg_index_begin_minimizers(fg_ctx* ctx, int32_t min_coverage, int32_t window, float repeat_rate,
                              uint64_t* hist);
int fg_index_build_range(fg_ctx* ctx, uint32_t bin_lo, uint32_t bin_hi, uint64_t* sums);
int fg_index_finish(fg_ctx* ctx, const uint64_t* total_sums, struct fg_index_stats* out);

/* The solid-mode selection in steps of its own: bounded memory on one GPU, and on several GPUs the exact counters
 * -- the 4^k * 4 B array that KmerCounter's flat array + overflow map become (vertex_index.cpp:499-616) -- held
 * only for a rank's own key range ("GPU g counts only canonical k-mers [of its slice] over all
 * reads"):
 *   kmer_hist       hist[FG_INDEX_BINS] = ALL k-mer positions per key bin: what the ranks' key ranges are balanced
 *                   on before anything is selected (the same ranges then serve fg_index_build_range);
 *   count_slice     allocates the counters of bins [bin_lo, bin_hi) and counts those k-mers over all reads;
 *                   *distinct = this range's share of "Total k-mers" (vertex_index.cpp:589; adds up over ranks);
 *                   *n_batches = the batches of reads the selection then runs in (<= FG_INDEX_BATCH_KMERS k-mer
 *                   positions each, default 2^28: the selection's scratch is bounded by that, not by the read set);
 *   batch_freq      KmerCounter::getFreq of every k-mer position of batch b as far as THIS context counted it (0
 *                   outside its range) into a device array (*d_freq, *n uint32): summed over the ranks in place
 *                   (all-reduce) it is the complete array; on one GPU it already is.  The array is complete when
 *                   the call returns; the caller's reduction must itself be COMPLETE (host-synchronised: the library
 *                   works on its own streams) before batch_select is called;
 *   batch_select    yieldFrequentKmers (vertex_index.cpp:316-358) over the batch's reads from that array;
 *   selection_done  releases the batch scratch; hist[] as fg_index_begin_solid gives it.
 * fg_index_begin_solid = count_slice(0, FG_INDEX_BINS) + {batch_freq, batch_select} per batch + selection_done.
 * fg_index_build_range then accepts only bins inside the counted range (the finish step asks the counters
 * about the keys it built, vertex_index.cpp:70-71). */
int fg_index_kmer_hist(fg_ctx* ctx, uint64_t* hist);
int fg_index_count_slice(fg_ctx* ctx, int32_t min_freq, float select_rate, int32_t tandem_freq, float repeat_rate,
                         float sample_rate_init, uint32_t bin_lo, uint32_t bin_hi, uint64_t* distinct,
                         uint32_t* n_batches);
int fg_index_batch_freq(fg_ctx* ctx, uint32_t batch, uint32_t** d_freq, uint64_t* n);
int fg_index_batch_select(fg_ctx* ctx, uint32_t batch);
int fg_index_selection_done(fg_ctx* ctx, uint64_t* hist);

/* The all-gather of a sharded build without a second copy of the index.  After fg_index_finish the context holds
 * its own piece.  gather_begin sets that piece aside and makes the context's arrays the full-size ones
 * (n_keys, n_entries, n_repetitive = the totals over the ranks); full[4] / piece[4] receive the DEVICE pointers
 * of {keys, key_off (n_keys + 1), entries, repetitive keys} of the full arrays and of the own piece,
 * piece_sizes[3] = {keys, entries, repetitive} of the piece.  The caller's collective writes every rank's piece
 * -- the own one included, list offsets shifted by the entries of the pieces before -- straight into the full
 * arrays; gather_end frees the piece, checks the arrays (FG_ERR_ARG when offsets or keys are out of order) and
 * builds the lookup structures. */
int fg_index_gather_begin(fg_ctx* ctx, uint64_t n_keys, uint64_t n_entries, uint64_t n_repetitive,
                          uint64_t** full, uint64_t** piece, uint64_t* piece_sizes);
int fg_index_gather_end(fg_ctx* ctx, float sample_rate);

/* Device bytes this library holds right now and at most since the last reset (all contexts of the process;
 * every device allocation of the library is counted); reset_peak != 0 restarts the peak at the current value. */
int fg_memory_stats(uint64_t* bytes_now, uint64_t* bytes_peak, int reset_peak);

/* An index given as CSR arrays in the layout fg_export_index writes (keys ascending, key_off[n_keys + 1],
 * entries ascending per key), in host memory or -- on_device != 0 -- in this context's device memory
 * (e.g. torch tensors filled by an RCCL all-gather).  sample_rate = VertexIndex::getSampleRate().
 * The arrays are checked on the device (offsets start at 0, never decrease, end at n_entries; keys strictly
 * ascending): FG_ERR_ARG otherwise. */
int fg_import_index(fg_ctx* ctx, uint64_t n_keys, const uint64_t* keys, const uint64_t* key_off,
                    uint64_t n_entries, const uint64_t* entries, uint64_t n_repetitive,
                    const uint64_t* repetitive_keys, float sample_rate, int on_device);

/* Device pointers of the built index (valid until the next build / import / clear), for collectives
 * that run on device memory. */
int fg_index_device_arrays(fg_ctx* ctx, uint64_t* n_keys, uint64_t* n_entries, uint64_t* n_repetitive,
                           const uint64_t** keys, const uint64_t** key_off, const uint64_t** entries,
                           const uint64_t** repetitive_keys);

/* VertexIndex::clear() (vertex_index.cpp:486-496) */
int fg_clear_index(fg_ctx* ctx);

/* Read-only export of the built index for parity tests: keys ascending,
 * key_off[n_keys+1] into entries; an entry is (record_index << 32) | position
 * with record_index = FastaRecord id - first_seq_id, i.e. the same order as the
 * reference's global position (vertex_index.cpp:108-114).  Pass NULL pointers
 * to query the sizes. */
int fg_export_index(fg_ctx* ctx, uint64_t* n_keys, uint64_t* n_entries,
                    uint64_t* n_repetitive, uint64_t* keys, uint64_t* key_off,
                    uint64_t* entries, uint64_t* repetitive_keys);

/* OverlapDetector constructor arguments (overlap.h:313-336) */
struct fg_detector_params {
	int32_t max_jump;
	int32_t min_overlap;
	int32_t max_overhang;          /* 0 => _checkOverhang = false */
	uint8_t keep_alignment;        /* 1: also return every overlap's kmerMatches (the chain thinned
	                                  to one match per > k query bases, overlap.cpp:368-377,
	                                  398-405) in fg_overlap_batch.matches */
	uint8_t only_max_ext;          /* 1: best overlap per target (assemble); 0: all primaries
	                                  not contained in a better one (overlap.cpp:441-458) */
	uint8_t nucl_alignment;        /* base-level divergence (alignment.cpp:218-247) */
	uint8_t partition_bad_mappings;/* 1: primaries that FAIL the divergence gate are returned too,
	                                  marked in fg_overlap_batch.needs_trim, in the position
	                                  where the caller splices in the result of its own
	                                  checkIdyAndTrim (overlap.cpp:474-485; the ksw2 alignment
	                                  stays on the host).  Only with max_overlaps = 0, as every
	                                  caller in the reference uses it (overlap.h:323);
	                                  FG_ERR_UNSUPPORTED otherwise */
	uint8_t use_hpc;
	uint8_t pad_[3];
	float   max_divergence;        /* OverlapDetector::_maxDivergence (mutable,
	                                  set by setDivergenceThreshold, overlap.cpp:820-827) */
};

/* One OverlapRange (overlap.h:20-279) plus the integers the float was made of.
 * seq_divergence is computed on the HOST with glibc logf/float division from the
 * device integers, exactly as overlap.cpp:417-423 / alignment.cpp:244-245 do. */
struct fg_overlap_rec {
	uint32_t cur_id, ext_id;
	int32_t  cur_begin, cur_end, cur_len;
	int32_t  ext_begin, ext_end, ext_len;
	int32_t  score;
	float    seq_divergence;
	int32_t  chain_length;        /* chainLength (overlap.cpp:366) */
	int32_t  filtered_positions;  /* repetitive query positions in range (:407-413) */
	int32_t  edit_distance;       /* -1 unless nucl_alignment */
	int32_t  hpc_len_cur, hpc_len_ext; /* compared string lengths (after HPC if on) */
};

struct fg_overlap_batch {
	uint32_t n_queries;
	uint64_t n_recs;
	uint64_t* query_off;           /* n_queries + 1, into recs */
	struct fg_overlap_rec* recs;   /* per query: reference emission order
	                                  (ascending ext_id, overlap.cpp:216-234) */
	uint64_t n_div_stats;
	uint64_t* div_stats_off;       /* n_queries + 1 */
	float*   div_stats;            /* OvlpDivStats::add() values (:488-506) */
	/* keep_alignment only (else 0 / NULL): OverlapRange::kmerMatches of recs[i] is the
	 * (cur, ext) int32 pairs matches[2*match_off[i]] .. matches[2*match_off[i+1]-1] */
	uint64_t n_matches;            /* pairs in total */
	uint64_t* match_off;           /* n_recs + 1 */
	int32_t* matches;
	/* partition_bad_mappings only (else NULL): needs_trim[i] = 1 when recs[i] did not pass
	 * seq_divergence < max_divergence and is there for the caller's checkIdyAndTrim */
	uint8_t* needs_trim;           /* n_recs */
	/* work counters of this call (for the roofline's m and d) */
	uint64_t query_bp, query_kmers, seed_hits, dp_groups, dp_elements;
	uint64_t dp_elements_small;    /* of dp_elements: in groups of <= 256 hits (the one-kernel chaining path) */
	double   device_seconds;       /* HIP-event time of the whole call */
	void*    owner_;               /* library arena; release with fg_release_batch */
};

/* getSeqOverlaps for a batch of FastaRecord ids (forward or reverse-complement
 * ids), as OverlapContainer::quickSeqOverlaps(id, max_overlaps, force_local)
 * would return them one by one (overlap.cpp:518-526). */
int fg_overlaps(fg_ctx* ctx, const struct fg_detector_params* p,
                const uint32_t* query_ids, uint32_t n_queries,
                int32_t max_overlaps, uint8_t force_local,
                struct fg_overlap_batch* out);
void fg_release_batch(struct fg_overlap_batch* b);

/* Per-kernel device time of the most recent fg_overlaps / build call, measured
 * with hipEvents on the library's own stream.  names[i] are static strings. */
struct fg_kernel_time { const char* name; double seconds; uint64_t launches; };
int fg_kernel_times(fg_ctx* ctx, struct fg_kernel_time* out, int max_entries);

/* Test hook: run the device hit-sort kernel (std::sort order by key, ties as GCC
 * libstdc++ introsort leaves them) on n_seg independent segments of (key, val)
 * pairs, in place in the caller's host arrays; seg_off has n_seg + 1 entries. */
int fg_debug_sort_pairs(fg_ctx* ctx, uint64_t* keys, uint32_t* vals,
                        const uint64_t* seg_off, uint32_t n_seg);

/* Test hook: the exact global edit distance (what edlibAlign(NW, TASK_DISTANCE, k = -1) returns,
 * reference src/sequence/alignment.cpp:233-238, src/sequence/edlib.cpp:141-296) of n_pairs string
 * pairs through the device kernels of the base-level divergence step.  Pair i = the forward
 * strands of reads 2i (rows) and 2i+1 (columns) of the container given to fg_set_reads;
 * use_hpc != 0 compresses homopolymers first (alignment.cpp:52-70).  out_len_a / out_len_b
 * receive the (compressed) lengths. */
int fg_debug_edit_distances(fg_ctx* ctx, uint32_t n_pairs, int use_hpc, int32_t* out_dist,
                            int32_t* out_len_a, int32_t* out_len_b);

/* getAlignmentCigarKsw (src/sequence/alignment.cpp:102-216) for a batch of (target, query)
 * string pairs: banded affine-gap global alignment (ksw_extz2 of the reference's lib/minimap2 with match 2,
 * mismatch -4, gap open 4, gap extend 2; band 64 doubling while too narrow; global backtrack) with the CIGAR
 * decoded into runs of '=', 'X', 'I', 'D' and the error rate (mismatches + indel bases) / max(length).
 * trg / qry: one byte per base (0..3), pair i at [off[i], off[i + 1]).  The DP and the backtrack run on the
 * device; the decoding of the M runs and the float on the host.  Runs of pair i: ops / lens[run_off[i] ..
 * run_off[i + 1]). */
struct fg_cigar_batch {
	uint32_t  n_pairs;
	uint64_t* run_off;
	uint8_t*  ops;        /* '=', 'X', 'I', 'D' */
	int32_t*  lens;
	float*    err_rate;   /* n_pairs */
	void*     owner_;
};
int fg_align_cigar_ksw(fg_ctx* ctx, uint32_t n_pairs, const uint8_t* trg, const uint64_t* trg_off,
                       const uint8_t* qry, const uint64_t* qry_off, struct fg_cigar_batch* out);
void fg_release_cigars(struct fg_cigar_batch* b);

#ifdef __cplusplus
}
#endif
#endif
