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
//       affine-gap alignment with CIGAR on the device)
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
#include "common/config.h"
#include "common/logger.h"

#include "flye_gpu.h"
#include "flye_gpu_bridge.h"

namespace {

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
// the public accessor; a one-line rawChunk() accessor in DnaSequence would make this a memcpy
void packSequence(const DnaSequence& seq, std::vector<uint64_t>& words)
{
	const size_t n = seq.length();
	const size_t w0 = words.size();
	words.resize(w0 + (n + 31) / 32, 0);
	for (size_t i = 0; i < n; ++i)
		words[w0 + i / 32] |= (uint64_t)seq.atRaw(i) << ((i % 32) * 2);
}

// device side of one VertexIndex: the context (reads + index in HBM) and one batch scheduler per detector
struct GpuIndex
{
	fg_ctx* ctx = nullptr;
	uint32_t firstId = 0, nFwd = 0;
	std::mutex mu;
	std::map<const OverlapDetector*, std::pair<fgb_container*, float>> detectors;	// + the gate it was last given

	~GpuIndex()
	{
		for (auto& kv : detectors) fgb_destroy(kv.second.first);
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

// upload the container's forward strands (the reverse complements are implied by the id layout,
// sequence_container.h:27-33, sequence_container.cpp:55-60)
GpuIndex* createIndex(const VertexIndex* vi, const SequenceContainer& seqs)
{
	std::unique_ptr<GpuIndex> gi(new GpuIndex);
	const int device = getenv("FLYE_GPU_DEVICE") ? atoi(getenv("FLYE_GPU_DEVICE")) : 0;
	check(fg_create(&gi->ctx, device, (int)Parameters::get().kmerSize), nullptr, "fg_create");
	std::vector<uint64_t> words, off(1, 0);
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
		packSequence(rec.sequence, words);
		off.push_back(words.size());
		len.push_back((int32_t)rec.sequence.length());
	}
	gi->nFwd = (uint32_t)len.size();
	if (words.empty()) words.push_back(0);
	check(fg_set_reads(gi->ctx, gi->nFwd, words.data(), off.data(), len.data(), gi->firstId), gi->ctx, "fg_set_reads");
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
	check(fg_build_index_solid(gi->ctx, globalMinFreq, selectRate, tandemFreq, (float)Config::get("repeat_kmer_rate"),
							   _sampleRate, &st), gi->ctx, "fg_build_index_solid");
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
	check(fg_build_index_minimizers(gi->ctx, minCoverage, wndLen, (float)Config::get("repeat_kmer_rate"), &st), gi->ctx,
		  "fg_build_index_minimizers");
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
namespace {
std::mutex g_alnMu;
fg_ctx* g_alnCtx = nullptr;		// alignments need no reads and no index: one context for the process

struct AlnResult { std::vector<uint8_t> trg, qry; std::vector<CigOp> cigar; float errRate; };
// alignments computed ahead for the calling thread (getSeqOverlaps knows all records of a read that will be
// trimmed and sends them to the device as ONE batch; checkIdyAndTrim then asks for them one by one)
thread_local std::vector<AlnResult> t_ahead;

void alignBatch(std::vector<AlnResult>& jobs)
{
	if (jobs.empty()) return;
	std::vector<uint8_t> trg, qry;
	std::vector<uint64_t> toff(1, 0), qoff(1, 0);
	for (auto& j : jobs)
	{
		trg.insert(trg.end(), j.trg.begin(), j.trg.end()); toff.push_back(trg.size());
		qry.insert(qry.end(), j.qry.begin(), j.qry.end()); qoff.push_back(qry.size());
	}
	if (trg.empty()) trg.push_back(0);
	if (qry.empty()) qry.push_back(0);
	fg_cigar_batch b;
	{
		std::lock_guard<std::mutex> g(g_alnMu);
		if (!g_alnCtx)
		{
			const int device = getenv("FLYE_GPU_DEVICE") ? atoi(getenv("FLYE_GPU_DEVICE")) : 0;
			check(fg_create(&g_alnCtx, device, (int)Parameters::get().kmerSize), nullptr, "fg_create");
		}
		check(fg_align_cigar_ksw(g_alnCtx, (uint32_t)jobs.size(), trg.data(), toff.data(), qry.data(), qoff.data(), &b),
			  g_alnCtx, "fg_align_cigar_ksw");
	}
	for (size_t i = 0; i < jobs.size(); ++i)
	{
		jobs[i].cigar.clear();
		for (uint64_t k = b.run_off[i]; k < b.run_off[i + 1]; ++k) jobs[i].cigar.push_back({(char)b.ops[k], (int)b.lens[k]});
		jobs[i].errRate = b.err_rate[i];
	}
	fg_release_cigars(&b);
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
	AlnResult one;
	one.trg.resize(trgLen); one.qry.resize(qryLen);
	for (size_t i = 0; i < trgLen; ++i) one.trg[i] = (uint8_t)trgSeq.atRaw(i + trgBegin);
	for (size_t i = 0; i < qryLen; ++i) one.qry[i] = (uint8_t)qrySeq.atRaw(i + qryBegin);
	for (auto& r : t_ahead)
		if (r.trg == one.trg && r.qry == one.qry) { cigarOut = r.cigar; return r.errRate; }
	std::vector<AlnResult> jobs(1);
	jobs[0].trg.swap(one.trg); jobs[0].qry.swap(one.qry);
	alignBatch(jobs);
	cigarOut = jobs[0].cigar;
	return jobs[0].errRate;
}

// ---- seam 2: OverlapDetector::getSeqOverlaps ---------------------------------------------------------
std::vector<OverlapRange>
OverlapDetector::getSeqOverlaps(const FastaRecord& fastaRec, bool forceLocal, OvlpDivStats& divStats, int maxOverlaps) const
{
	GpuIndex* gi = findIndex(&_vertexIndex);
	if (!gi) throw std::runtime_error("flye_gpu: getSeqOverlaps on a VertexIndex that was never built");
	fgb_container* cont = nullptr;
	{
		// one batch scheduler per detector (its constructor arguments, overlap.h:313-336); the gate is
		// mutable (setDivergenceThreshold, overlap.cpp:820-827) and followed here
		std::lock_guard<std::mutex> g(gi->mu);
		auto it = gi->detectors.find(this);
		if (it == gi->detectors.end())
		{
			fg_detector_params p;
			memset(&p, 0, sizeof(p));
			p.max_jump = _maxJump; p.min_overlap = _minOverlap; p.max_overhang = _maxOverhang;
			p.keep_alignment = _keepAlignment; p.only_max_ext = _onlyMaxExt; p.nucl_alignment = _nuclAlignment;
			p.partition_bad_mappings = _partitionBadMappings; p.use_hpc = _useHpc;
			p.max_divergence = _maxDivergence;
			const uint32_t maxBatch = getenv("FLYE_GPU_MAX_BATCH") ? (uint32_t)atoi(getenv("FLYE_GPU_MAX_BATCH")) : 4096u;
			const uint32_t linger = getenv("FLYE_GPU_LINGER_US") ? (uint32_t)atoi(getenv("FLYE_GPU_LINGER_US")) : 200u;
			check(fgb_create(&cont, gi->ctx, &p, maxBatch, linger), gi->ctx, "fgb_create");
			it = gi->detectors.emplace(this, std::make_pair(cont, _maxDivergence)).first;
		}
		cont = it->second.first;
		if (it->second.second != _maxDivergence)
		{
			check(fgb_set_divergence_threshold(cont, _maxDivergence), gi->ctx, "fgb_set_divergence_threshold");
			it->second.second = _maxDivergence;
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
				t_ahead.emplace_back();
				compressedBytes(fastaRec.sequence, r.cur_begin, r.cur_end - r.cur_begin, _useHpc, t_ahead.back().trg);
				compressedBytes(_seqContainer.getSeq(FastaRecord::Id(r.ext_id)), r.ext_begin, r.ext_end - r.ext_begin, _useHpc,
								t_ahead.back().qry);
			}
		alignBatch(t_ahead);
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
	return detectedOverlaps;
}
