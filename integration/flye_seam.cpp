// The Flye-side binding, compiled: OUR definitions of the reference's own member functions at the
// two seams of the hot path, written against the UNMODIFIED headers under
// /root/reference/src.  Linked in front of the reference's objects (integration/Makefile,
// -Wl,--allow-multiple-definition: the first definition of a symbol wins), they turn any
// program built on the reference's sequence library into "Flye with the MI355X hot path":
//
//   VertexIndex::countKmers / buildIndexUnevenCoverage / buildIndexMinimizers / clear
//       (src/sequence/vertex_index.cpp:19-125, :389-496)   -> fg_set_reads + fg_build_index_*
//   OverlapDetector::getSeqOverlaps
//       (src/sequence/overlap.cpp:99-508)                   -> fgb_quick_ex (the batch scheduler:
//       Flye's worker threads call this one read at a time, include/flye_gpu_bridge.h)
//   getAlignmentCigarKsw
//       (src/sequence/alignment.cpp:102-216)                -> fg_align_cigar_ksw (ksw2's banded
//       affine-gap alignment with CIGAR on the device), through one dispatcher thread that merges
//       what the caller threads ask for at the same time into one device batch
//   ConsensusGenerator::generateAlignments
//       (src/sequence/consensus_generator.cpp:82-126)       -> all alignments of all disjointigs as ONE
//       device batch (the reference hands them out one per thread)
//
// Everything above the seams stays reference code, compiled from the reference's files:
// OverlapContainer::quickSeqOverlaps / lazySeqOverlaps (cache + complemented twin),
// estimateOverlaperParameters (libc rand()), setDivergenceThreshold, findAllOverlaps,
// ensureTransitivity, filterOverlaps, buildIntervalTree, checkIdyAndTrim (ksw2) ...
// No reference file is edited and no reference text is copied: the packed read words are read
// through DnaSequence's public atRaw(), private members are reached as members.
//
// oracle/ref_dumper.cpp linked this way is oracle/_ref/ref_dumper_gpu; tests/test_seam.py compares
// its output with the golden files the pure reference produced.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <cuckoohash_map.hh>
#include "IntervalTree.h"

#include "sequence/sequence_container.h"
#include "sequence/vertex_index.h"
#include "sequence/overlap.h"
#include "sequence/alignment.h"
#include "sequence/consensus_generator.h"
#include "common/config.h"
#include "common/logger.h"
#include "common/parallel.h"
#include <chrono>
#include <condition_variable>

#include "flye_gpu.h"
#include "flye_gpu_bridge.h"

namespace {

// ---- what the seams did, for whoever runs the program (FLYE_GPU_STATS=<path>: written at exit as JSON) ----
double nowS() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct SeamStats
{
	std::mutex mu;
	double uploadS = 0, indexBuildS = 0;
	std::atomic<uint64_t> ovlpCalls{0}, ovlpNanos{0};				// getSeqOverlaps calls, time inside (summed over threads)
	std::atomic<uint64_t> kswCalls{0}, kswAheadHits{0}, kswNanos{0};	// getAlignmentCigarKsw calls / served by a batch computed ahead
	uint64_t kswBatches = 0, kswPairs = 0;							// device batches of alignments, pairs in them
	double kswDeviceS = 0;
	uint64_t consensusPairs = 0; double consensusS = 0;				// ConsensusGenerator::generateAlignments
	fgb_stats bridge{0, 0, 0, 0, 0, 0, 0};							// summed over the batch schedulers that lived
	void addBridge(fgb_container* c)
	{
		fgb_stats st; fgb_get_stats(c, &st);
		std::lock_guard<std::mutex> g(mu);
		bridge.device_calls += st.device_calls; bridge.reads_computed += st.reads_computed; bridge.requests += st.requests;
		bridge.cache_hits += st.cache_hits; bridge.reads_ahead += st.reads_ahead; bridge.ahead_hits += st.ahead_hits;
	}
	~SeamStats()
	{
		const char* path = getenv("FLYE_GPU_STATS");
		if (!path) return;
		FILE* f = fopen(path, "w");
		if (!f) return;
		fprintf(f, "{\"upload_s\": %.6f, \"index_build_s\": %.6f, \"get_seq_overlaps_calls\": %llu, \"get_seq_overlaps_s_sum\": %.6f, "
				"\"ksw_calls\": %llu, \"ksw_ahead_hits\": %llu, \"ksw_s_sum\": %.6f, \"ksw_device_batches\": %llu, \"ksw_pairs\": %llu, "
				"\"ksw_device_s\": %.6f, \"consensus_pairs\": %llu, \"consensus_s\": %.6f, "
				"\"bridge\": {\"device_calls\": %llu, \"reads_computed\": %llu, \"requests\": %llu, \"reads_ahead\": %llu, \"ahead_hits\": %llu}}\n",
				uploadS, indexBuildS, (unsigned long long)ovlpCalls.load(), ovlpNanos.load() * 1e-9,
				(unsigned long long)kswCalls.load(), (unsigned long long)kswAheadHits.load(), kswNanos.load() * 1e-9,
				(unsigned long long)kswBatches, (unsigned long long)kswPairs, kswDeviceS,
				(unsigned long long)consensusPairs, consensusS,
				(unsigned long long)bridge.device_calls, (unsigned long long)bridge.reads_computed, (unsigned long long)bridge.requests,
				(unsigned long long)bridge.reads_ahead, (unsigned long long)bridge.ahead_hits);
		fclose(f);
	}
};
SeamStats g_stats;		// defined first: destroyed after everything below has reported into it

void check(int rc, fg_ctx* ctx, const char* what)
{
	if (rc == FG_OK) return;
	std::string msg = std::string("flye_gpu: ") + what + ": " + fg_strerror(rc);
	if (ctx) msg += std::string(" (") + fg_last_error(ctx) + ")";
	throw std::runtime_error(msg);	// -> Flye's terminate handler (src/common/utils.h:82-107)
}

// FastaRecord::Id::_id through the public signedId() (sequence_container.h:44-45, invertible)
uint32_t rawId(FastaRecord::Id id)
{
	const int s = id.signedId();
	return s > 0 ? 2u * (uint32_t)(s - 1) : 2u * (uint32_t)(-s) - 1u;
}

// DnaSequence's packing (sequence.h:54-69: 32 nt per word, nt j at bits (j % 32) * 2), rebuilt through
// the public accessor, word by word; a one-line rawChunk() accessor in DnaSequence would make this a memcpy
void packWords(const DnaSequence& seq, uint64_t* out)
{
	const size_t n = seq.length();
	for (size_t w = 0; w * 32 < n; ++w)
	{
		const size_t e = std::min<size_t>(32, n - w * 32);
		uint64_t x = 0;
		for (size_t j = 0; j < e; ++j) x |= (uint64_t)seq.atRaw(w * 32 + j) << (j * 2);
		out[w] = x;
	}
}
void packSequence(const DnaSequence& seq, std::vector<uint64_t>& words)
{
	const size_t w0 = words.size();
	words.resize(w0 + (seq.length() + 31) / 32, 0);
	packWords(seq, words.data() + w0);
}

// device side of one VertexIndex: the context (reads + index in HBM) and one batch scheduler per detector
struct GpuIndex
{
	fg_ctx* ctx = nullptr;
	uint32_t firstId = 0, nFwd = 0;
	std::mutex mu;
	// one batch scheduler per detector, remembered with the constructor arguments it was made from (a detector
	// re-created at the same address with other arguments gets a new one) and the gate it was last given
	struct Det { fgb_container* cont; fg_detector_params made; float gate; };
	std::map<const OverlapDetector*, Det> detectors;

	~GpuIndex()
	{
		for (auto& kv : detectors) { g_stats.addBridge(kv.second.cont); fgb_destroy(kv.second.cont); }
		if (ctx) fg_destroy(ctx);
	}
};

std::mutex g_mu;
std::map<const VertexIndex*, std::unique_ptr<GpuIndex>> g_index;

GpuIndex* findIndex(const VertexIndex* vi)
{
	std::lock_guard<std::mutex> g(g_mu);
	auto it = g_index.find(vi);
	return it == g_index.end() ? nullptr : it->second.get();
}

void ensureAlignmentContext();

// upload the container's forward strands (the reverse complements are implied by the id layout,
// sequence_container.h:27-33, sequence_container.cpp:55-60)
GpuIndex* createIndex(const VertexIndex* vi, const SequenceContainer& seqs)
{
	const double t0 = nowS();
	std::unique_ptr<GpuIndex> gi(new GpuIndex);
	const int device = getenv("FLYE_GPU_DEVICE") ? atoi(getenv("FLYE_GPU_DEVICE")) : 0;
	check(fg_create(&gi->ctx, device, (int)Parameters::get().kmerSize), nullptr, "fg_create");
	// the alignment context too, here on the thread that builds the index and not lazily from a worker thread
	// (the first HIP initialisation of a process swaps libc's rand() state for its duration, fg_api.hip)
	ensureAlignmentContext();
	std::vector<const DnaSequence*> fwd;
	std::vector<uint64_t> off(1, 0);
	std::vector<int32_t> len;
	bool first = true;
	uint32_t expect = 0;
	for (const auto& rec : seqs.iterSeqs())
	{
		if (!rec.id.strand()) continue;
		const uint32_t id = rawId(rec.id);
		if (first) { gi->firstId = id; expect = id; first = false; }
		if (id != expect) throw std::runtime_error("flye_gpu: forward record ids are expected to step by 2");
		expect += 2;
		fwd.push_back(&rec.sequence);
		off.push_back(off.back() + (rec.sequence.length() + 31) / 32);
		len.push_back((int32_t)rec.sequence.length());
	}
	gi->nFwd = (uint32_t)len.size();
	// reads are independent: the words are rebuilt on the program's worker threads (215 M atRaw calls for E. coli
	// PB 50x took 0.5 s on one thread)
	std::vector<uint64_t> words(std::max<uint64_t>(1, off.back()), 0);
	{
		const size_t nThreads = std::max<size_t>(1, std::min<size_t>(Parameters::get().numThreads, 64));
		std::atomic<size_t> next(0);
		auto work = [&]()
		{
			while (true)
			{
				const size_t a = next.fetch_add(64);
				if (a >= fwd.size()) return;
				for (size_t i = a; i < std::min(fwd.size(), a + 64); ++i) packWords(*fwd[i], words.data() + off[i]);
			}
		};
		std::vector<std::thread> th;
		for (size_t t = 1; t < nThreads; ++t) th.emplace_back(work);
		work();
		for (auto& t : th) t.join();
	}
	check(fg_set_reads(gi->ctx, gi->nFwd, words.data(), off.data(), len.data(), gi->firstId), gi->ctx, "fg_set_reads");
	{ std::lock_guard<std::mutex> g(g_stats.mu); g_stats.uploadS += nowS() - t0; }
	std::lock_guard<std::mutex> g(g_mu);
	GpuIndex* raw = gi.get();
	g_index[vi] = std::move(gi);
	return raw;
}

} // namespace

// ---- seam 1: VertexIndex ----------------------------------------------------------------------
void VertexIndex::countKmers()
{
	// vertex_index.cpp:19-22; the counting runs fused with the build on the device.  The flat counter's
	// k <= 17 limit (vertex_index.cpp:504-507) is the library's FG_ERR_KMER_SIZE at build time.
	if (!findIndex(this)) createIndex(this, _seqContainer);
}

void VertexIndex::buildIndexUnevenCoverage(int globalMinFreq, float selectRate, int tandemFreq)
{
	GpuIndex* gi = findIndex(this);
	if (!gi) throw std::runtime_error("flye_gpu: countKmers() must be called first");
	fg_index_stats st;
	const double t0 = nowS();
	check(fg_build_index_solid(gi->ctx, globalMinFreq, selectRate, tandemFreq, (float)Config::get("repeat_kmer_rate"),
							   _sampleRate, &st), gi->ctx, "fg_build_index_solid");
	{ std::lock_guard<std::mutex> g(g_stats.mu); g_stats.indexBuildS += nowS() - t0; }
	_repetitiveFrequency = st.repetitive_frequency;
	Logger::get().debug() << "Total k-mers " << st.total_kmers;					// vertex_index.cpp:589
	Logger::get().debug() << "Repetitive k-mer frequency: " << st.repetitive_frequency;	// :188-189
	Logger::get().debug() << "Filtered " << st.repetitive_kmers << " repetitive k-mers";
	Logger::get().debug() << "Selected k-mers: " << st.selected_kmers;			// :121-124
	Logger::get().debug() << "Index size: " << st.index_entries;
}

void VertexIndex::buildIndexMinimizers(int minCoverage, int wndLen)
{
	GpuIndex* gi = findIndex(this);
	if (!gi) gi = createIndex(this, _seqContainer);
	fg_index_stats st;
	const double t0 = nowS();
	check(fg_build_index_minimizers(gi->ctx, minCoverage, wndLen, (float)Config::get("repeat_kmer_rate"), &st), gi->ctx,
		  "fg_build_index_minimizers");
	{ std::lock_guard<std::mutex> g(g_stats.mu); g_stats.indexBuildS += nowS() - t0; }
	_repetitiveFrequency = st.repetitive_frequency;
	_sampleRate = st.sample_rate;												// vertex_index.cpp:480-482
	Logger::get().debug() << "Selected k-mers: " << st.selected_kmers;			// :473-476
	Logger::get().debug() << "K-mer index size: " << st.index_entries;
	Logger::get().debug() << "Mean k-mer frequency: " << st.mean_frequency;
	Logger::get().debug() << "Minimizer rate: " << _sampleRate;
}

void VertexIndex::clear()
{
	// vertex_index.cpp:486-496 (also the destructor's path): the device side of this index goes
	std::unique_ptr<GpuIndex> dead;
	{
		std::lock_guard<std::mutex> g(g_mu);
		auto it = g_index.find(this);
		if (it != g_index.end()) { dead = std::move(it->second); g_index.erase(it); }
	}
}

// ---- seam 3: getAlignmentCigarKsw ------------------------------------------------------
// alignment.cpp:102-216: banded affine-gap global alignment with CIGAR, called by checkIdyAndTrim (:306-495, on
// the records getSeqOverlaps marks for trimming) and by the consensus stage.  checkIdyAndTrim itself -- the
// interval search over the CIGAR, its std::sort, the coordinate mapping through the compression tables --
// stays the reference's compiled code and reaches this definition through the symbol.
//
// The callers are N worker threads with ONE pair each (processInParallel); a device call of one pair costs what
// a call of hundreds does.  As on the overlap side (fgb_*), callers block and one dispatcher thread owns the
// context: whatever is waiting after a short linger goes to the device as one batch.
namespace {
struct AlnResult { std::vector<uint8_t> trg, qry; std::vector<CigOp> cigar; float errRate = 0; };

uint64_t bytesHash(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b)
{
	uint64_t h = 1469598103934665603ULL ^ (a.size() * 0x9E3779B97F4A7C15ULL) ^ (b.size() << 32);
	auto eat = [&h](const std::vector<uint8_t>& v)
	{
		size_t i = 0;
		for (; i + 8 <= v.size(); i += 8) { uint64_t w; memcpy(&w, v.data() + i, 8); h = (h ^ w) * 1099511628211ULL; h ^= h >> 29; }
		for (; i < v.size(); ++i) h = (h ^ v[i]) * 1099511628211ULL;
	};
	eat(a); eat(b);
	return h;
}

// alignments computed ahead for the calling thread (getSeqOverlaps knows all records of a read that will be
// trimmed and sends them to the device as ONE batch; checkIdyAndTrim then asks for them one by one), found by a
// hash of the two byte strings
struct Ahead
{
	std::vector<AlnResult> jobs;
	std::unordered_multimap<uint64_t, size_t> byHash;
	void clear() { jobs.clear(); byHash.clear(); }
	void index() { byHash.clear(); for (size_t i = 0; i < jobs.size(); ++i) byHash.emplace(bytesHash(jobs[i].trg, jobs[i].qry), i); }
	const AlnResult* find(const std::vector<uint8_t>& trg, const std::vector<uint8_t>& qry) const
	{
		if (jobs.empty()) return nullptr;
		auto range = byHash.equal_range(bytesHash(trg, qry));
		for (auto it = range.first; it != range.second; ++it)
			if (jobs[it->second].trg == trg && jobs[it->second].qry == qry) return &jobs[it->second];
		return nullptr;
	}
};
thread_local Ahead t_ahead;

class AlnDispatcher
{
public:
	// blocks until every job has its cigar and error rate
	void align(AlnResult* jobs, size_t n)
	{
		if (!n) return;
		Ticket tk{jobs, n, false, std::string()};
		std::unique_lock<std::mutex> lk(mu);
		startLocked();
		queue.push_back(&tk);
		pending += n;
		cvWork.notify_one();
		cvDone.wait(lk, [&] { return tk.done; });
		if (!tk.error.empty()) throw std::runtime_error(tk.error);
	}
	void ensureContext()
	{
		std::lock_guard<std::mutex> g(ctxMu);
		if (ctx) return;
		const int device = getenv("FLYE_GPU_DEVICE") ? atoi(getenv("FLYE_GPU_DEVICE")) : 0;
		check(fg_create(&ctx, device, (int)Parameters::get().kmerSize), nullptr, "fg_create");
	}
	~AlnDispatcher()
	{
		{ std::lock_guard<std::mutex> g(mu); stop = true; }
		cvWork.notify_all();
		if (worker.joinable()) worker.join();
		if (ctx) fg_destroy(ctx);
	}
private:
	struct Ticket { AlnResult* jobs; size_t n; bool done; std::string error; };
	std::mutex mu, ctxMu;
	std::condition_variable cvWork, cvDone;
	std::deque<Ticket*> queue;
	size_t pending = 0;
	bool stop = false, started = false;
	std::thread worker;
	fg_ctx* ctx = nullptr;		// alignments need no reads and no index: one context for the process

	void startLocked() { if (!started) { started = true; worker = std::thread([this] { run(); }); } }
	void run()
	{
		const unsigned lingerUs = getenv("FLYE_GPU_ALN_LINGER_US") ? (unsigned)atoi(getenv("FLYE_GPU_ALN_LINGER_US")) : 150u;
		// a batch is worth waiting for while more callers can still arrive: as many pairs as the program has threads
		const size_t full = std::max<size_t>(1, Parameters::get().numThreads);
		std::unique_lock<std::mutex> lk(mu);
		while (true)
		{
			cvWork.wait(lk, [&] { return stop || !queue.empty(); });
			if (stop) return;
			if (pending < full && lingerUs)
				cvWork.wait_for(lk, std::chrono::microseconds(lingerUs), [&] { return stop || pending >= full; });
			std::vector<Ticket*> batch(queue.begin(), queue.end());
			queue.clear(); pending = 0;
			lk.unlock();
			std::string err;
			try { deviceBatch(batch); } catch (const std::exception& e) { err = e.what(); }
			lk.lock();
			for (Ticket* t : batch) { t->error = err; t->done = true; }
			cvDone.notify_all();
		}
	}
	void deviceBatch(const std::vector<Ticket*>& batch)
	{
		ensureContext();
		std::vector<uint8_t> trg, qry;
		std::vector<uint64_t> toff(1, 0), qoff(1, 0);
		size_t n = 0;
		for (Ticket* t : batch)
			for (size_t i = 0; i < t->n; ++i, ++n)
			{
				trg.insert(trg.end(), t->jobs[i].trg.begin(), t->jobs[i].trg.end()); toff.push_back(trg.size());
				qry.insert(qry.end(), t->jobs[i].qry.begin(), t->jobs[i].qry.end()); qoff.push_back(qry.size());
			}
		if (trg.empty()) trg.push_back(0);
		if (qry.empty()) qry.push_back(0);
		fg_cigar_batch b;
		const double t0 = nowS();
		check(fg_align_cigar_ksw(ctx, (uint32_t)n, trg.data(), toff.data(), qry.data(), qoff.data(), &b), ctx, "fg_align_cigar_ksw");
		{ std::lock_guard<std::mutex> g(g_stats.mu); g_stats.kswBatches += 1; g_stats.kswPairs += n; g_stats.kswDeviceS += nowS() - t0; }
		size_t j = 0;
		for (Ticket* t : batch)
			for (size_t i = 0; i < t->n; ++i, ++j)
			{
				AlnResult& r = t->jobs[i];
				r.cigar.clear();
				r.cigar.reserve(b.run_off[j + 1] - b.run_off[j]);
				for (uint64_t k = b.run_off[j]; k < b.run_off[j + 1]; ++k) r.cigar.push_back({(char)b.ops[k], (int)b.lens[k]});
				r.errRate = b.err_rate[j];
			}
		fg_release_cigars(&b);
	}
};
AlnDispatcher g_aln;

void ensureAlignmentContext() { g_aln.ensureContext(); }

void sequenceBytes(const DnaSequence& seq, size_t begin, size_t len, std::vector<uint8_t>& out)
{
	out.resize(len);
	for (size_t i = 0; i < len; ++i) out[i] = (uint8_t)seq.atRaw(i + begin);
}

// homopolymerCompression's sequence (alignment.cpp:52-70) as bytes
void compressedBytes(const DnaSequence& seq, int32_t start, int32_t length, bool doCompression, std::vector<uint8_t>& out)
{
	out.clear();
	for (int32_t i = 0; i < length; ++i)
	{
		const uint8_t b = (uint8_t)seq.atRaw((size_t)i + start);
		if (!doCompression || i == 0 || out.back() != b) out.push_back(b);
	}
}
} // namespace

float getAlignmentCigarKsw(const DnaSequence& trgSeq, size_t trgBegin, size_t trgLen,
						   const DnaSequence& qrySeq, size_t qryBegin, size_t qryLen,
						   float maxAlnErr, std::vector<CigOp>& cigarOut)
{
	(void)maxAlnErr;
	const double t0 = nowS();
	AlnResult one;
	sequenceBytes(trgSeq, trgBegin, trgLen, one.trg);
	sequenceBytes(qrySeq, qryBegin, qryLen, one.qry);
	g_stats.kswCalls.fetch_add(1);
	if (const AlnResult* r = t_ahead.find(one.trg, one.qry))
	{
		g_stats.kswAheadHits.fetch_add(1);
		cigarOut = r->cigar;
		return r->errRate;
	}
	g_aln.align(&one, 1);
	cigarOut.swap(one.cigar);
	g_stats.kswNanos.fetch_add((uint64_t)((nowS() - t0) * 1e9));
	return one.errRate;
}

// ---- seam 4: ConsensusGenerator::generateAlignments --------------------------------------
// consensus_generator.cpp:82-126: one getAlignmentCigarKsw + decodeCigar per pair of consecutive reads of every
// disjointig, handed out one per thread.  Here every pair of every disjointig goes to the device in ONE batch
// (a device call of one pair costs what a call of hundreds does); the decoding into gapped strings stays the
// reference's decodeCigar, on the program's worker threads.  generateConsensuses reaches this definition through
// the symbol (checked with objdump -dr: R_X86_64_PLT32 against the global name).
ConsensusGenerator::AlignmentsMap
	ConsensusGenerator::generateAlignments(const std::vector<ContigPath>& contigs, bool verbose)
{
	const double t0 = nowS();
	typedef std::pair<const ContigPath*, size_t> AlnTask;
	std::vector<AlnTask> tasks;
	for (auto& path : contigs)
		for (size_t i = 0; i + 1 < path.sequences.size(); ++i) tasks.emplace_back(&path, i);
	std::vector<AlnResult> jobs(tasks.size());
	std::function<void(const size_t&)> extract = [&](const size_t& t)
	{
		const ContigPath* path = tasks[t].first;
		const size_t i = tasks[t].second;
		sequenceBytes(path->sequences[i], path->overlaps[i].curBegin, path->overlaps[i].curRange(), jobs[t].trg);
		sequenceBytes(path->sequences[i + 1], path->overlaps[i].extBegin, path->overlaps[i].extRange(), jobs[t].qry);
	};
	std::vector<size_t> order(tasks.size());
	for (size_t t = 0; t < order.size(); ++t) order[t] = t;
	processInParallel(order, extract, Parameters::get().numThreads, false);
	g_aln.align(jobs.data(), jobs.size());

	AlignmentsMap alnMap;
	std::mutex mapMutex;
	std::function<void(const size_t&)> decode = [&](const size_t& t)
	{
		const ContigPath* path = tasks[t].first;
		const size_t i = tasks[t].second;
		std::string alignedLeft, alignedRight;
		decodeCigar(jobs[t].cigar, path->sequences[i], path->overlaps[i].curBegin,
					path->sequences[i + 1], path->overlaps[i].extBegin, alignedLeft, alignedRight);
		std::lock_guard<std::mutex> lock(mapMutex);
		alnMap[&path->overlaps[i]] = {alignedLeft, alignedRight, path->overlaps[i].curBegin, path->overlaps[i].extBegin};
	};
	processInParallel(order, decode, Parameters::get().numThreads, verbose);
	{ std::lock_guard<std::mutex> g(g_stats.mu); g_stats.consensusPairs += tasks.size(); g_stats.consensusS += nowS() - t0; }
	return alnMap;
}

// ---- seam 2: OverlapDetector::getSeqOverlaps ---------------------------------------------------------
std::vector<OverlapRange>
OverlapDetector::getSeqOverlaps(const FastaRecord& fastaRec, bool forceLocal, OvlpDivStats& divStats, int maxOverlaps) const
{
	GpuIndex* gi = findIndex(&_vertexIndex);
	if (!gi) throw std::runtime_error("flye_gpu: getSeqOverlaps on a VertexIndex that was never built");
	const double tCall = nowS();
	fgb_container* cont = nullptr;
	{
		// one batch scheduler per detector (its constructor arguments, overlap.h:313-336); the gate is
		// mutable (setDivergenceThreshold, overlap.cpp:820-827) and followed here
		std::lock_guard<std::mutex> g(gi->mu);
		fg_detector_params p;
		memset(&p, 0, sizeof(p));
		p.max_jump = _maxJump; p.min_overlap = _minOverlap; p.max_overhang = _maxOverhang;
		p.keep_alignment = _keepAlignment; p.only_max_ext = _onlyMaxExt; p.nucl_alignment = _nuclAlignment;
		p.partition_bad_mappings = _partitionBadMappings; p.use_hpc = _useHpc;
		auto it = gi->detectors.find(this);
		if (it != gi->detectors.end() && memcmp(&it->second.made, &p, sizeof(p)) != 0)
		{
			// another detector lives at this address now: its scheduler is made from ITS arguments
			g_stats.addBridge(it->second.cont);
			fgb_destroy(it->second.cont);
			gi->detectors.erase(it);
			it = gi->detectors.end();
		}
		if (it == gi->detectors.end())
		{
			fg_detector_params withGate = p;
			withGate.max_divergence = _maxDivergence;
			const uint32_t maxBatch = getenv("FLYE_GPU_MAX_BATCH") ? (uint32_t)atoi(getenv("FLYE_GPU_MAX_BATCH")) : 4096u;
			const uint32_t linger = getenv("FLYE_GPU_LINGER_US") ? (uint32_t)atoi(getenv("FLYE_GPU_LINGER_US")) : 200u;
			check(fgb_create(&cont, gi->ctx, &withGate, maxBatch, linger), gi->ctx, "fgb_create");
			it = gi->detectors.emplace(this, GpuIndex::Det{cont, p, _maxDivergence}).first;
		}
		cont = it->second.cont;
		if (it->second.gate != _maxDivergence)
		{
			check(fgb_set_divergence_threshold(cont, _maxDivergence), gi->ctx, "fgb_set_divergence_threshold");
			it->second.gate = _maxDivergence;
		}
	}

	// a record of the indexed container goes by id; any other record (ReadAligner's reads against graph
	// edges, read_aligner.cpp:178-217) takes its sequence along
	const uint32_t id = rawId(fastaRec.id);
	const bool indexed = id >= gi->firstId && id - gi->firstId < 2 * gi->nFwd &&
						 fastaRec.sequence.length() == (size_t)_seqContainer.seqLen(fastaRec.id);
	std::vector<uint64_t> words;
	if (!indexed) { packSequence(fastaRec.sequence, words); if (words.empty()) words.push_back(0); }
	fgb_result res;
	check(fgb_quick_ex(cont, id, indexed ? nullptr : words.data(), (int32_t)fastaRec.sequence.length(), maxOverlaps,
					   forceLocal, &res), gi->ctx, "fgb_quick_ex");

	std::vector<OverlapRange> detectedOverlaps;
	detectedOverlaps.reserve(res.n);
	// every record of this read that checkIdyAndTrim will realign: one device batch ahead of the calls
	t_ahead.clear();
	if (res.needs_trim)
	{
		for (uint64_t i = 0; i < res.n; ++i)
			if (res.needs_trim[i])
			{
				const fg_overlap_rec& r = res.recs[i];
				t_ahead.jobs.emplace_back();
				compressedBytes(fastaRec.sequence, r.cur_begin, r.cur_end - r.cur_begin, _useHpc, t_ahead.jobs.back().trg);
				compressedBytes(_seqContainer.getSeq(FastaRecord::Id(r.ext_id)), r.ext_begin, r.ext_end - r.ext_begin, _useHpc,
								t_ahead.jobs.back().qry);
			}
		g_aln.align(t_ahead.jobs.data(), t_ahead.jobs.size());
		t_ahead.index();
	}
	for (uint64_t i = 0; i < res.n; ++i)
	{
		const fg_overlap_rec& r = res.recs[i];
		OverlapRange ovlp(FastaRecord::Id(r.cur_id), FastaRecord::Id(r.ext_id), r.cur_begin, r.ext_begin, r.cur_len, r.ext_len);
		ovlp.curEnd = r.cur_end; ovlp.extEnd = r.ext_end;
		ovlp.score = r.score; ovlp.seqDivergence = r.seq_divergence;
		if (res.match_off)		// overlap.cpp:368-377, :398-405
		{
			ovlp.kmerMatches = new std::vector<std::pair<int32_t, int32_t>>();
			ovlp.kmerMatches->reserve(res.match_off[i + 1] - res.match_off[i]);
			for (uint64_t j = res.match_off[i]; j < res.match_off[i + 1]; ++j)
				ovlp.kmerMatches->emplace_back(res.matches[2 * j], res.matches[2 * j + 1]);
		}
		if (res.needs_trim && res.needs_trim[i])
		{
			// failed the gate: the reference's own ksw2 trimming decides which parts stay (overlap.cpp:474-485)
			auto trimmedOverlaps = checkIdyAndTrim(ovlp, fastaRec.sequence, _seqContainer.getSeq(ovlp.extId),
												   _maxDivergence, _minOverlap, _useHpc);
			for (auto& trimOvlp : trimmedOverlaps) detectedOverlaps.push_back(trimOvlp);
		}
		else detectedOverlaps.push_back(ovlp);
	}
	for (uint64_t i = 0; i < res.n_div_stats; ++i) divStats.add(res.div_stats[i]);	// overlap.cpp:500-506
	t_ahead.clear();
	fgb_release_result(&res);
	g_stats.ovlpCalls.fetch_add(1);
	g_stats.ovlpNanos.fetch_add((uint64_t)((nowS() - tCall) * 1e9));
	return detectedOverlaps;
}
