/* flye_gpu_bridge.h -- host-side batch scheduler above the C ABI of flye_gpu.h
 * (SURVEY.md §8f, N1).
 *
 * Flye asks for overlaps ONE read at a time from up to --threads worker threads
 * (OverlapContainer::lazySeqOverlaps / quickSeqOverlaps, src/sequence/overlap.cpp:
 * 518-574, called from extender.cpp, chimera.cpp, repeat_graph.cpp,
 * read_aligner.cpp); the device wants batches, and an fg_ctx may only be used by
 * one host thread at a time.  This container is what OverlapContainer's query
 * side becomes: any number of threads call fgb_lazy / fgb_quick concurrently and
 * block; ONE dispatcher thread owns the fg_ctx, drains the pending requests every
 * `linger_us` microseconds (or as soon as `max_batch` are waiting) into one
 * fg_overlaps call per (max_overlaps, force_local) class, stores the lazily
 * computed lists exactly as overlap.cpp:555-571 does (forward list + its
 * OverlapRange::complement()ed twin, overlap.h:118-147) and wakes the callers.
 * A per-read result is a pure function of (read, index, parameters), so the
 * batching is invisible in the results.
 *
 * All functions are thread safe unless noted.  Status codes are flye_gpu.h's.
 */
#ifndef FLYE_GPU_BRIDGE_H
#define FLYE_GPU_BRIDGE_H

#include "flye_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fgb_container fgb_container;

/* ctx must have its index built and must not be used directly by the caller while
 * the container lives.  params is copied (max_divergence can be changed later with
 * fgb_set_divergence_threshold).  keep_alignment / partition_bad_mappings are passed
 * through to fg_overlaps; their extra arrays (kmerMatches, needs_trim) come back through
 * fgb_quick_ex only -- fgb_lazy / fgb_quick / fgb_prefetch answer FG_ERR_UNSUPPORTED on a
 * container created with either flag. */
int  fgb_create(fgb_container** out, fg_ctx* ctx, const struct fg_detector_params* params,
                uint32_t max_batch, uint32_t linger_us);
void fgb_destroy(fgb_container* c);

/* OverlapContainer::lazySeqOverlaps(readId) (overlap.cpp:528-574): maxOverlaps = 0,
 * forceLocal = false, computed for the forward id, cached, the reverse-complement
 * id served by the complemented list.  *recs stays valid until fgb_destroy. */
int fgb_lazy(fgb_container* c, uint32_t read_id, const struct fg_overlap_rec** recs, uint64_t* n);

/* OverlapContainer::quickSeqOverlaps(readId, maxOverlaps, forceLocal) (overlap.cpp:
 * 518-526): not cached.  Writes at most cap records, *n is the full count. */
int fgb_quick(fgb_container* c, uint32_t read_id, int32_t max_overlaps, uint8_t force_local,
              struct fg_overlap_rec* out, uint64_t cap, uint64_t* n);

/* OverlapDetector::getSeqOverlaps(fastaRec, forceLocal, divStats, maxOverlaps)
 * (overlap.cpp:99-508) for ONE record, as the reference's OverlapContainer calls it from many
 * threads (overlap.cpp:518-526, :547-551): everything of the result that belongs to this read --
 * records, kmerMatches (keep_alignment), needs_trim marks (partition_bad_mappings: the caller runs
 * its own checkIdyAndTrim on those, overlap.cpp:474-485) and the values getSeqOverlaps appends to
 * OvlpDivStats (:500-506).  Not cached.
 *
 * words == NULL: read_id is a record of the indexed container (or of the fg_set_queries one).
 * words != NULL: the record lives in a container the device does not hold (reads against graph
 * edges, read_aligner.cpp:178-217): its sequence -- len bases as given, DnaSequence packing
 * (sequence.h:54-69: 32 nt per uint64, nt j at bits (j%32)*2) -- travels with the request; the
 * dispatcher uploads the waiting foreign records of a batch as a temporary query container.
 * read_id is only used to label the records (cur_id) and must not be an id of the indexed container.
 * The arrays stay valid until fgb_release_result(). */
struct fgb_result {
	uint64_t n;                         /* overlaps */
	const struct fg_overlap_rec* recs;
	const uint64_t* match_off;          /* keep_alignment: n + 1 offsets, in pairs */
	const int32_t* matches;             /* (cur, ext) pairs */
	const uint8_t* needs_trim;          /* partition_bad_mappings: n marks */
	uint64_t n_div_stats;
	const float* div_stats;
	void* owner_;
};
int fgb_quick_ex(fgb_container* c, uint32_t read_id, const uint64_t* words, int32_t len,
                 int32_t max_overlaps, uint8_t force_local, struct fgb_result* out);
void fgb_release_result(struct fgb_result* r);

/* Hint: these reads will be asked for (e.g. Extender::assembleDisjointigs warming the
 * cache over all forward reads, extender.cpp:363-382).  Returns at once; the lists are
 * computed in batches of max_batch by the dispatcher. */
int fgb_prefetch(fgb_container* c, const uint32_t* read_ids, uint32_t n);

/* OverlapContainer::setDivergenceThreshold's effect on the detector (overlap.cpp:
 * 820-827): later device calls use the new gate; cached lists are kept, as in the
 * reference. */
int fgb_set_divergence_threshold(fgb_container* c, float max_divergence);

/* OvlpDivStats values appended so far (overlap.h:283-309): copies min(cap, count)
 * floats, returns the count. */
uint64_t fgb_divergence_stats(fgb_container* c, float* out, uint64_t cap);

struct fgb_stats {
	uint64_t device_calls;     /* fg_overlaps calls made */
	uint64_t reads_computed;   /* query ids sent to the device */
	uint64_t requests;         /* fgb_lazy + fgb_quick calls */
	uint64_t cache_hits;       /* fgb_lazy calls answered without waiting for the device */
	uint64_t cached_overlaps;  /* OverlapContainer::indexSize() */
	uint64_t reads_ahead;      /* of reads_computed: records computed ahead of their request (quick path) */
	uint64_t ahead_hits;       /* fgb_quick / fgb_quick_ex calls answered from those */
};
void fgb_get_stats(fgb_container* c, struct fgb_stats* out);

/* ---- on-disk text forms: what a stage boundary writes and reads --------------
 * OverlapRange::dump / load (src/sequence/overlap.h:227-251): one overlap as
 *   "<curName> <curBegin> <curEnd> <curLen> <extName> <extBegin> <extEnd> <extLen> -1 -1 <score> <div>"
 * with the names SequenceContainer::seqName gives ('+' / '-' + FASTA header,
 * sequence_container.cpp:62, :75) and the divergence in operator<<(float)'s default form (%g).
 * fg_overlap_dump writes the line (no newline) and returns its length, or the length needed when it
 * does not fit cap.  fg_overlap_load parses one: ids are the caller's business (recordByName), the
 * names come back as offsets into the line; returns FG_OK or FG_ERR_ARG on a malformed line. */
int64_t fg_overlap_dump(const struct fg_overlap_rec* rec, const char* cur_name, const char* ext_name,
                        char* buf, uint64_t cap);
int fg_overlap_load(const char* line, struct fg_overlap_rec* rec, uint32_t* cur_name_off, uint32_t* cur_name_len,
                    uint32_t* ext_name_off, uint32_t* ext_name_len);
/* One record of ReadAligner::storeAlignments (src/repeat_graph/read_aligner.cpp:321-339):
 * "\tAln\t<edgeId>\t" + the dump line; chains are introduced by a "Chain" line the caller writes. */
int64_t fg_alignment_dump(int64_t edge_id, const struct fg_overlap_rec* rec, const char* read_name,
                          const char* edge_name, char* buf, uint64_t cap);
/* SequenceContainer::writeFasta's form of one record (sequence_container.cpp:330-357): ">" + name and
 * the sequence in slices of 80, from DnaSequence packing (32 nt per uint64).  Returns the length
 * written / needed as above. */
int64_t fg_fasta_record(const char* name, const uint64_t* words, int32_t len, char* buf, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
