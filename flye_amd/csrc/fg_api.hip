// C ABI of libflyegpu.so (include/flye_gpu.h): context, read upload, index
// build/export entry points and the error boundary.  No exception leaves this
// file; there is no CPU fallback -- without a HIP device fg_create fails.
#include "fg_ctx.h"
#include <chrono>

#include <algorithm>
#include <new>

#include <mutex>
#include <atomic>

static std::mutex g_poolMutex;
static std::vector<BatchOwner*> g_pool;

BatchOwner* BatchOwner::acquire()
{
	std::lock_guard<std::mutex> lock(g_poolMutex);
	if (g_pool.empty()) return new BatchOwner;
	BatchOwner* b = g_pool.back();
	g_pool.pop_back();
	return b;
}

void BatchOwner::release(BatchOwner* b)
{
	if (!b) return;
	std::lock_guard<std::mutex> lock(g_poolMutex);
	if (g_pool.size() < 2) { b->nRecs = 0; b->stats.clear(); g_pool.push_back(b); }
	else delete b;
}

namespace {

// Initialising the HIP runtime draws from libc's rand() stream (measured: tools/rand_stream_probe.py; only
// the first runtime initialisation of a process does).  The reference picks the reads of
// estimateOverlaperParameters with rand() (overlap.cpp:752-756, also sequence_container.cpp:318-328,
// chimera.cpp:76) and never seeds it, so a host program that creates a context first would see other
// picks.  While this guard lives the process draws from a private state array; the caller's stream
// continues exactly where it was (glibc keeps the position inside the state array it hands back).
// Only the FIRST fg_create of a process takes the guard (later ones initialise nothing that draws), and swapping
// glibc's process-wide state is not thread safe: that first call must come from a thread beside which no other
// thread uses rand() -- in Flye the main thread building the index (flye_gpu.h, fg_create).
struct RandStreamGuard {
	char buf[128];
	char* old = nullptr;
	RandStreamGuard()
	{
		static std::atomic<bool> firstDone{false};
		if (!firstDone.exchange(true)) old = initstate(1u, buf, sizeof(buf));
	}
	~RandStreamGuard() { if (old) setstate(old); }
};

template <class F>
int guarded(fg_ctx* c, F f)
{
	int rc = FG_OK;
	try { f(); return FG_OK; }
	catch (const FgError& e) { if (c) c->lastError = e.msg; rc = e.code; }
	catch (const std::bad_alloc&) { if (c) c->lastError = "host allocation failed"; rc = FG_ERR_NOMEM; }
	catch (const std::exception& e) { if (c) c->lastError = e.what(); rc = FG_ERR_HIP; }
	// a call that failed half way may have left launches behind on either stream: nothing of the context's
	// scratch is reused before they have drained
	for (fg_ctx* x : {c, c ? c->lane2.get() : (fg_ctx*)nullptr})		// the second lane of fg_overlaps has streams of its own
	{
		if (x && x->stream2) (void)hipStreamSynchronize(x->stream2);
		if (x && x->stream3) (void)hipStreamSynchronize(x->stream3);
		if (x && x->stream) (void)hipStreamSynchronize(x->stream);
	}
	return rc;
}

} // namespace

extern "C" {

int fg_abi_version(void) { return FG_ABI_VERSION; }

const char* fg_strerror(int code)
{
	switch (code)
	{
	case FG_OK: return "ok";
	case FG_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
	case FG_ERR_HIP: return "HIP runtime error";
	case FG_ERR_ARG: return "invalid argument";
	case FG_ERR_STATE: return "invalid call order";
	case FG_ERR_KMER_TOO_FREQUENT: return "k-mer is too frequent";
	case FG_ERR_KMER_SIZE: return "unsupported k-mer size";
	case FG_ERR_UNSUPPORTED: return "flag combination not supported yet";
	case FG_ERR_NOMEM: return "out of memory";
	default: return "unknown error";
	}
}

const char* fg_last_error(const fg_ctx* ctx) { return ctx ? ctx->lastError.c_str() : ""; }

int fg_create(fg_ctx** out, int device, int kmer_size)
{
	if (!out) return FG_ERR_ARG;
	*out = nullptr;
	if (kmer_size < 1 || kmer_size > 32) return FG_ERR_KMER_SIZE;
	RandStreamGuard keepCallersRandStream;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return FG_ERR_NO_DEVICE;
	if (device < 0 || device >= count) return FG_ERR_NO_DEVICE;
	if (hipSetDevice(device) != hipSuccess) return FG_ERR_NO_DEVICE;
	fg_ctx* c = new (std::nothrow) fg_ctx;
	if (!c) return FG_ERR_NOMEM;
	c->device = device;
	c->k = kmer_size;
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
		hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
		hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) != hipSuccess ||
		hipEventCreateWithFlags(&c->evJoin3, hipEventDisableTiming) != hipSuccess ||
		hipEventCreateWithFlags(&c->evOff, hipEventDisableTiming) != hipSuccess ||
		hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming) != hipSuccess ||
		hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming) != hipSuccess)
	{
		delete c;
		return FG_ERR_NO_DEVICE;
	}
	for (auto& e : c->evPiece)
		if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete c; return FG_ERR_NO_DEVICE; }
	c->timer.stream = c->stream;
	*out = c;
	return FG_OK;
}

void fg_destroy(fg_ctx* ctx)
{
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	delete ctx;
}

int fg_container_info(const fg_ctx* c, uint32_t* first_id, uint32_t* n_fwd, uint32_t* query_first_id, uint32_t* query_n_fwd)
{
	if (!c) return FG_ERR_ARG;
	if (first_id) *first_id = c->firstId;
	if (n_fwd) *n_fwd = c->nReads;
	if (query_first_id) *query_first_id = c->hasQ ? c->qFirstId : 0;
	if (query_n_fwd) *query_n_fwd = c->hasQ ? c->nQReads : 0;
	return FG_OK;
}

int fg_set_reads(fg_ctx* c, uint32_t n, const uint64_t* words, const uint64_t* word_off,
				 const int32_t* len, uint32_t first_seq_id)
{
	if (!c || (n && (!words || !word_off || !len))) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		if ((u64)first_seq_id + 2ULL * n > 0xFFFFFFFFULL) throw FgError{FG_ERR_ARG, "sequence ids overflow uint32"};
		c->indexBuilt = false;
		c->hasQ = false; c->nQReads = 0; c->hQLen.clear();
		c->nReads = n;
		c->firstId = first_seq_id;
		c->totalWords = n ? word_off[n] : 0;
		c->hLen.assign(len, len + n);
		c->hKmerOff.assign(n + 1, 0);
		c->totalBases = 0;
		c->maxLen = 0;
		for (u32 i = 0; i < n; ++i)
		{
			if (len[i] < 0) throw FgError{FG_ERR_ARG, "negative read length"};
			if ((u64)(len[i] + 31) / 32 > word_off[i + 1] - word_off[i])
				throw FgError{FG_ERR_ARG, "word_off does not cover read " + std::to_string(i)};
			c->hKmerOff[i + 1] = c->hKmerOff[i] + (u64)std::max(0, len[i] - c->k);
			c->totalBases += len[i];
			c->maxLen = std::max(c->maxLen, len[i]);
		}
		c->totalKmers = c->hKmerOff[n];
		c->dWords.alloc(c->totalWords + 2);
		c->dWordOff.alloc(n + 1);
		c->dLen.alloc(n);
		c->dKmerOff.alloc(n + 1);
		hipStream_t s = c->stream;
		HIP_CHECK(hipMemsetAsync(c->dWords.p + c->totalWords, 0, 16, s));
		if (n)
		{
			HIP_CHECK(hipMemcpyAsync(c->dWords.p, words, c->totalWords * 8, hipMemcpyHostToDevice, s));
			HIP_CHECK(hipMemcpyAsync(c->dWordOff.p, word_off, (n + 1) * 8ULL, hipMemcpyHostToDevice, s));
			HIP_CHECK(hipMemcpyAsync(c->dLen.p, len, n * 4ULL, hipMemcpyHostToDevice, s));
		}
		HIP_CHECK(hipMemcpyAsync(c->dKmerOff.p, c->hKmerOff.data(), (n + 1) * 8ULL, hipMemcpyHostToDevice, s));
		HIP_CHECK(hipStreamSynchronize(s));
	});
}

int fg_set_queries(fg_ctx* c, uint32_t n, const uint64_t* words, const uint64_t* word_off,
				   const int32_t* len, uint32_t first_seq_id)
{
	if (!c || (n && (!words || !word_off || !len))) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		c->hasQ = n > 0;
		c->nQReads = n; c->qFirstId = first_seq_id; c->qMaxLen = 0;
		c->hQLen.clear();
		if (!n) { c->dQWords.release(); c->dQWordOff.release(); c->dQLen.release(); return; }
		if ((u64)first_seq_id + 2ULL * n > 0xFFFFFFFFULL) throw FgError{FG_ERR_ARG, "sequence ids overflow uint32"};
		// the reference hands out ids from one process-wide counter: the two containers never share ids
		const u64 a0 = c->firstId, a1 = (u64)c->firstId + 2ULL * c->nReads, b0 = first_seq_id, b1 = (u64)first_seq_id + 2ULL * n;
		if (a0 < b1 && b0 < a1) throw FgError{FG_ERR_ARG, "query ids overlap the ids of the indexed container"};
		c->hQLen.assign(len, len + n);
		for (u32 i = 0; i < n; ++i)
		{
			if (len[i] < 0) throw FgError{FG_ERR_ARG, "negative read length"};
			if ((u64)(len[i] + 31) / 32 > word_off[i + 1] - word_off[i])
				throw FgError{FG_ERR_ARG, "word_off does not cover query " + std::to_string(i)};
			c->qMaxLen = std::max(c->qMaxLen, len[i]);
		}
		const u64 nw = word_off[n];
		c->dQWords.alloc(nw + 2); c->dQWordOff.alloc(n + 1); c->dQLen.alloc(n);
		hipStream_t s = c->stream;
		HIP_CHECK(hipMemsetAsync(c->dQWords.p + nw, 0, 16, s));
		HIP_CHECK(hipMemcpyAsync(c->dQWords.p, words, nw * 8, hipMemcpyHostToDevice, s));
		HIP_CHECK(hipMemcpyAsync(c->dQWordOff.p, word_off, (n + 1) * 8ULL, hipMemcpyHostToDevice, s));
		HIP_CHECK(hipMemcpyAsync(c->dQLen.p, len, n * 4ULL, hipMemcpyHostToDevice, s));
		HIP_CHECK(hipStreamSynchronize(s));
	});
}

int fg_build_index_solid(fg_ctx* c, int32_t min_freq, float select_rate, int32_t tandem_freq,
						 float repeat_rate, float sample_rate_init, struct fg_index_stats* out)
{
	if (!c || !out) return FG_ERR_ARG;
	if (!(select_rate >= 0.0f && select_rate < 1.0f)) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		fgBuildIndexSolid(c, min_freq, select_rate, tandem_freq, repeat_rate, sample_rate_init, out);
	});
}

int fg_build_index_minimizers(fg_ctx* c, int32_t min_coverage, int32_t window, float repeat_rate,
							  struct fg_index_stats* out)
{
	if (!c || !out) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		fgBuildIndexMinimizers(c, min_coverage, window, repeat_rate, out);
	});
}

int fg_index_begin_solid(fg_ctx* c, int32_t min_freq, float select_rate, int32_t tandem_freq, float repeat_rate,
						 float sample_rate_init, uint64_t* hist)
{
	if (!c) return FG_ERR_ARG;
	if (!(select_rate >= 0.0f && select_rate < 1.0f)) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		fgIndexBeginSolid(c, min_freq, select_rate, tandem_freq, repeat_rate, sample_rate_init, hist);
	});
}

int fg_index_begin_minimizers(fg_ctx* c, int32_t min_coverage, int32_t window, float repeat_rate, uint64_t* hist)
{
	if (!c) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		fgIndexBeginMinimizers(c, min_coverage, window, repeat_rate, hist);
	});
}

int fg_index_build_range(fg_ctx* c, uint32_t bin_lo, uint32_t bin_hi, uint64_t* sums)
{
	if (!c) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		unsigned long long s2[2];
		fgIndexBuildRange(c, bin_lo, bin_hi, s2);
		if (sums) { sums[0] = s2[0]; sums[1] = s2[1]; }
	});
}

int fg_index_finish(fg_ctx* c, const uint64_t* total_sums, struct fg_index_stats* out)
{
	if (!c || !out) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		unsigned long long s2[2] = {0, 0};
		if (total_sums) { s2[0] = total_sums[0]; s2[1] = total_sums[1]; }
		fgIndexFinish(c, total_sums ? s2 : nullptr, out);
	});
}

int fg_index_kmer_hist(fg_ctx* c, uint64_t* hist)
{
	if (!c || !hist) return FG_ERR_ARG;
	return guarded(c, [&]() { HIP_CHECK(hipSetDevice(c->device)); fgIndexKmerHist(c, hist); });
}

int fg_index_count_slice(fg_ctx* c, int32_t min_freq, float select_rate, int32_t tandem_freq, float repeat_rate,
						 float sample_rate_init, uint32_t bin_lo, uint32_t bin_hi, uint64_t* distinct, uint32_t* n_batches)
{
	if (!c) return FG_ERR_ARG;
	if (!(select_rate >= 0.0f && select_rate < 1.0f)) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		fgIndexCountSlice(c, min_freq, select_rate, tandem_freq, repeat_rate, sample_rate_init, bin_lo, bin_hi, distinct, n_batches);
	});
}

int fg_index_batch_freq(fg_ctx* c, uint32_t batch, uint32_t** d_freq, uint64_t* n)
{
	if (!c) return FG_ERR_ARG;
	return guarded(c, [&]() { HIP_CHECK(hipSetDevice(c->device)); fgIndexBatchFreq(c, batch, d_freq, n); });
}

int fg_index_batch_select(fg_ctx* c, uint32_t batch)
{
	if (!c) return FG_ERR_ARG;
	return guarded(c, [&]() { HIP_CHECK(hipSetDevice(c->device)); fgIndexBatchSelect(c, batch); });
}

int fg_index_selection_done(fg_ctx* c, uint64_t* hist)
{
	if (!c) return FG_ERR_ARG;
	return guarded(c, [&]() { HIP_CHECK(hipSetDevice(c->device)); fgIndexSelectionDone(c, hist); });
}

int fg_index_gather_begin(fg_ctx* c, uint64_t n_keys, uint64_t n_entries, uint64_t n_repetitive, uint64_t** full,
						  uint64_t** piece, uint64_t* piece_sizes)
{
	if (!c || !full || !piece || !piece_sizes) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		fgIndexGatherBegin(c, n_keys, n_entries, n_repetitive, (u64**)full, (u64**)piece, (u64*)piece_sizes);
	});
}

int fg_index_gather_end(fg_ctx* c, float sample_rate)
{
	if (!c) return FG_ERR_ARG;
	return guarded(c, [&]() { HIP_CHECK(hipSetDevice(c->device)); fgIndexGatherEnd(c, sample_rate); });
}

int fg_memory_stats(uint64_t* bytes_now, uint64_t* bytes_peak, int reset_peak)
{
	if (bytes_now) *bytes_now = g_fgDevBytes.load();
	if (bytes_peak) *bytes_peak = g_fgDevPeak.load();
	if (reset_peak) g_fgDevPeak.store(g_fgDevBytes.load());
	return FG_OK;
}

int fg_import_index(fg_ctx* c, uint64_t n_keys, const uint64_t* keys, const uint64_t* key_off, uint64_t n_entries,
					const uint64_t* entries, uint64_t n_repetitive, const uint64_t* repetitive_keys, float sample_rate,
					int on_device)
{
	if (!c || !key_off || (n_keys && !keys) || (n_entries && !entries) || (n_repetitive && !repetitive_keys)) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		fgImportIndex(c, n_keys, keys, key_off, n_entries, entries, n_repetitive, repetitive_keys, sample_rate, on_device);
	});
}

int fg_index_device_arrays(fg_ctx* c, uint64_t* n_keys, uint64_t* n_entries, uint64_t* n_repetitive,
						   const uint64_t** keys, const uint64_t** key_off, const uint64_t** entries,
						   const uint64_t** repetitive_keys)
{
	if (!c) return FG_ERR_ARG;
	if (!c->indexBuilt) return FG_ERR_STATE;
	if (n_keys) *n_keys = c->nKeys;
	if (n_entries) *n_entries = c->nEntries;
	if (n_repetitive) *n_repetitive = c->nRep;
	if (keys) *keys = c->dKeys.p;
	if (key_off) *key_off = c->dKeyOff.p;
	if (entries) *entries = c->dEntries.p;
	if (repetitive_keys) *repetitive_keys = c->dRepKeys.p;
	return FG_OK;
}

int fg_clear_index(fg_ctx* c)
{
	if (!c) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		c->indexBuilt = false;
		c->indexBuild.reset();
		c->gathering = false;
		c->gKeys.release(); c->gKeyOff.release(); c->gEntries.release(); c->gRepKeys.release();
		c->dKeys.release(); c->dKeyOff.release(); c->dEntries.release(); c->dRepKeys.release();
		c->dTable.release(); c->dIndexedBits.release();
		c->nKeys = c->nEntries = c->nRep = c->tableSlots = 0;
	});
}

int fg_export_index(fg_ctx* c, uint64_t* n_keys, uint64_t* n_entries, uint64_t* n_repetitive,
					uint64_t* keys, uint64_t* key_off, uint64_t* entries, uint64_t* repetitive_keys)
{
	if (!c || !n_keys || !n_entries || !n_repetitive) return FG_ERR_ARG;
	if (!c->indexBuilt) return FG_ERR_STATE;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		*n_keys = c->nKeys; *n_entries = c->nEntries; *n_repetitive = c->nRep;
		hipStream_t s = c->stream;
		if (keys && c->nKeys) HIP_CHECK(hipMemcpyAsync(keys, c->dKeys.p, c->nKeys * 8, hipMemcpyDeviceToHost, s));
		if (key_off) HIP_CHECK(hipMemcpyAsync(key_off, c->dKeyOff.p, (c->nKeys + 1) * 8, hipMemcpyDeviceToHost, s));
		if (entries && c->nEntries) HIP_CHECK(hipMemcpyAsync(entries, c->dEntries.p, c->nEntries * 8, hipMemcpyDeviceToHost, s));
		if (repetitive_keys && c->nRep) HIP_CHECK(hipMemcpyAsync(repetitive_keys, c->dRepKeys.p, c->nRep * 8, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
	});
}

int fg_overlaps(fg_ctx* c, const struct fg_detector_params* p, const uint32_t* query_ids,
				uint32_t n_queries, int32_t max_overlaps, uint8_t force_local,
				struct fg_overlap_batch* out)
{
	if (!c || !p || !out || (n_queries && !query_ids)) return FG_ERR_ARG;
	memset(out, 0, sizeof(*out));
	if (!c->indexBuilt) return FG_ERR_STATE;
	if (p->partition_bad_mappings && max_overlaps != 0) return FG_ERR_UNSUPPORTED;
	if (p->max_jump <= 0 || p->min_overlap <= 0 || max_overlaps < 0) return FG_ERR_ARG;
	{
		const u32 base = c->hasQ ? c->qFirstId : c->firstId;
		const u32 cnt = c->hasQ ? c->nQReads : c->nReads;
		for (u32 i = 0; i < n_queries; ++i)
			if (query_ids[i] < base || query_ids[i] - base >= 2 * cnt) return FG_ERR_ARG;
	}
	const int rc = guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		fgOverlaps(c, p, query_ids, n_queries, max_overlaps, force_local, out);
	});
	if (rc != FG_OK)
	{
		BatchOwner::release((BatchOwner*)out->owner_);
		memset(out, 0, sizeof(*out));
	}
	return rc;
}

int fg_debug_sort_pairs(fg_ctx* c, uint64_t* keys, uint32_t* vals, const uint64_t* seg_off, uint32_t n_seg)
{
	if (!c || !seg_off || (seg_off[n_seg] && (!keys || !vals))) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		fgDebugSortPairs(c, keys, vals, seg_off, n_seg);
	});
}

int fg_debug_edit_distances(fg_ctx* c, uint32_t n_pairs, int use_hpc, int32_t* out_dist, int32_t* out_len_a,
							int32_t* out_len_b)
{
	if (!c || (n_pairs && (!out_dist || !out_len_a || !out_len_b))) return FG_ERR_ARG;
	return guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		fgDebugEditDistances(c, n_pairs, use_hpc, out_dist, out_len_a, out_len_b);
	});
}

namespace {
struct CigarOwner {
	std::vector<uint64_t> runOff;
	std::vector<uint8_t> ops;
	std::vector<int32_t> lens;
	std::vector<float> err;
};
}

int fg_align_cigar_ksw(fg_ctx* c, uint32_t n_pairs, const uint8_t* trg, const uint64_t* trg_off, const uint8_t* qry,
					   const uint64_t* qry_off, struct fg_cigar_batch* out)
{
	if (!c || !out || !trg_off || !qry_off || (trg_off[n_pairs] && !trg) || (qry_off[n_pairs] && !qry)) return FG_ERR_ARG;
	memset(out, 0, sizeof(*out));
	CigarOwner* own = nullptr;
	const int rc = guarded(c, [&]()
	{
		HIP_CHECK(hipSetDevice(c->device));
		std::vector<u64> runOff;
		std::vector<u32> runs;
		fgKswAlign(c, n_pairs, trg, trg_off, qry, qry_off, runOff, runs);
		const auto tDec = std::chrono::steady_clock::now();
		own = new CigarOwner;
		own->runOff.assign(n_pairs + 1, 0);
		own->err.assign(n_pairs, 0.0f);
		// the decoding loop of alignment.cpp:172-211, on the host: M runs split into '=' / 'X'.  Pairs are
		// independent: slices of the batch (cut by bases) go to the context's worker threads.
		u64 totalBp = 0;
		for (u32 i = 0; i < n_pairs; ++i) totalBp += (trg_off[i + 1] - trg_off[i]) + (qry_off[i + 1] - qry_off[i]);
		unsigned nThreads = totalBp < (1u << 20) ? 1u : std::max(1u, std::min(fg_usable_cpus(), 32u));
		if (getenv("FG_SHIM_THREADS")) nThreads = std::max(1, atoi(getenv("FG_SHIM_THREADS")));
		std::vector<u32> cut(nThreads + 1, n_pairs);
		cut[0] = 0;
		{
			u64 acc = 0; unsigned t = 1;
			for (u32 i = 0; i < n_pairs && t < nThreads; ++i)
			{
				acc += (trg_off[i + 1] - trg_off[i]) + (qry_off[i + 1] - qry_off[i]);
				while (t < nThreads && acc >= totalBp * t / nThreads) cut[t++] = i + 1;
			}
		}
		std::vector<std::vector<uint8_t>> tOps(nThreads);
		std::vector<std::vector<int32_t>> tLens(nThreads);
		std::atomic<int> workerFailed{0};		// no exception may leave a pool thread
		c->shimPool.run(nThreads, [&](unsigned th)
		{
		  try
		  {
			// thread-local vectors (the headers of tOps[] sit next to each other: appending through them would
			// bounce one cache line between all threads), handed over at the end
			std::vector<uint8_t> ops;
			std::vector<int32_t> lens;
			for (u32 i = cut[th]; i < cut[th + 1]; ++i)
			{
				const uint8_t* t = trg + trg_off[i];
				const uint8_t* q = qry + qry_off[i];
				const size_t trgLen = trg_off[i + 1] - trg_off[i], qryLen = qry_off[i + 1] - qry_off[i];
				size_t posQry = 0, posTrg = 0;
				int numMiss = 0, numIndels = 0;
				const size_t first = ops.size();
				for (u64 k = runOff[i]; k < runOff[i + 1]; ++k)
				{
					const int size = (int)(runs[k] >> 4);
					const u32 op = runs[k] & 0xf;
					if (op == 0)
					{
						for (int x = 0; x < size; ++x)
						{
							const char match = t[posTrg + x] == q[posQry + x] ? '=' : 'X';
							if (x == 0 || match != (char)ops.back()) { ops.push_back((uint8_t)match); lens.push_back(1); }
							else ++lens.back();
							numMiss += match == 'X';
						}
						posQry += size; posTrg += size;
					}
					else if (op == 1) { ops.push_back('I'); lens.push_back(size); posQry += size; numIndels += size; }
					else { ops.push_back('D'); lens.push_back(size); posTrg += size; numIndels += size; }
				}
				own->runOff[i + 1] = ops.size() - first;		// count; summed below
				own->err[i] = float(numMiss + numIndels) / std::max(trgLen, qryLen);
			}
			tOps[th].swap(ops); tLens[th].swap(lens);
		  }
		  catch (...) { workerFailed.store(1); }
		});
		if (workerFailed.load()) throw std::bad_alloc();
		for (u32 i = 0; i < n_pairs; ++i) own->runOff[i + 1] += own->runOff[i];
		own->ops.reserve(own->runOff[n_pairs]); own->lens.reserve(own->runOff[n_pairs]);
		for (unsigned th = 0; th < nThreads; ++th)
		{
			own->ops.insert(own->ops.end(), tOps[th].begin(), tOps[th].end());
			own->lens.insert(own->lens.end(), tLens[th].begin(), tLens[th].end());
		}
		if (getenv("FG_KSW_TRACE"))
			fprintf(stderr, "[ksw] decode on %u threads %.1f ms\n", nThreads, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tDec).count());
	});
	if (rc != FG_OK) { delete own; return rc; }
	out->n_pairs = n_pairs;
	out->run_off = own->runOff.data();
	out->ops = own->ops.data();
	out->lens = own->lens.data();
	out->err_rate = own->err.data();
	out->owner_ = own;
	return FG_OK;
}

void fg_release_cigars(struct fg_cigar_batch* b)
{
	if (!b) return;
	delete (CigarOwner*)b->owner_;
	memset(b, 0, sizeof(*b));
}

void fg_release_batch(struct fg_overlap_batch* b)
{
	if (!b) return;
	BatchOwner::release((BatchOwner*)b->owner_);
	memset(b, 0, sizeof(*b));
}

int fg_kernel_times(fg_ctx* c, struct fg_kernel_time* out, int max_entries)
{
	if (!c || (max_entries > 0 && !out)) return FG_ERR_ARG;
	int n = (int)c->timer.last.size();
	for (int i = 0; i < n && i < max_entries; ++i) out[i] = c->timer.last[i];
	return n;
}

} // extern "C"
