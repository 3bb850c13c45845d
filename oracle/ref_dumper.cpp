// TEST INFRASTRUCTURE -- not shipped, never on the product path.
//
// Driver around the UNMODIFIED reference (Flye 2.8.1) sources where they lie
// under /root/reference: loads a FASTA/FASTQ through the reference's own
// SequenceContainer, builds the reference VertexIndex exactly the way
// src/assemble/main_assemble.cpp:158-242 does, runs
// OverlapContainer::quickSeqOverlaps() for every forward read through the
// reference's processInParallel, and prints index contents and OverlapRange
// records in a stable text form (floats as raw bit patterns).
//
// Built only by oracle/Makefile into oracle/_ref/ (git-ignored).  It is the
// generator of tests/golden/* and the "reference" CPU baseline of bench.py.
// No reference source text is copied here; the reference is reached through
// its headers and object files.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
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
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <cuckoohash_map.hh>
#include "IntervalTree.h"

// the dumper needs to read private index state; class layout is unaffected
#define private public
#define protected public
#include "sequence/sequence_container.h"
#include "sequence/vertex_index.h"
#include "sequence/overlap.h"
#include "common/config.h"
#include "common/parallel.h"
#include "sequence/edlib.h"
#include "sequence/alignment.h"
#include "sequence/consensus_generator.h"
#undef private
#undef protected

static uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(int argc, char** argv)
{
	std::string reads, queriesFile, params, config, indexOut, ovlpOut, dumpOut, fastaOut, divMode = "none";
	int onlyMax = 1, maxOverhang = -1, nuclAln = -1, keepAln = 0, partitionBad = 0;
	float maxDiv = 1.0f;
	int threads = 1, minReadLen = 0, maxOverlaps = 0, forceLocal = 0;
	int minOverlap = 1000;	// main_assemble.cpp:174
	long queryLimit = -1;
	bool rcQueries = false, findAll = false;
	for (int i = 1; i < argc; ++i)
	{
		std::string a = argv[i];
		auto next = [&]() { return std::string(argv[++i]); };
		if (a == "--reads") reads = next();
		else if (a == "--params") params = next();
		else if (a == "--config") config = next();
		else if (a == "--threads") threads = atoi(next().c_str());
		else if (a == "--min-read-len") minReadLen = atoi(next().c_str());
		else if (a == "--max-overlaps") maxOverlaps = atoi(next().c_str());
		else if (a == "--force-local") forceLocal = atoi(next().c_str());
		else if (a == "--min-overlap") minOverlap = atoi(next().c_str());
		else if (a == "--div-mode") divMode = next();
		else if (a == "--index-out") indexOut = next();
		else if (a == "--ovlp-out") ovlpOut = next();
		else if (a == "--dump-out") dumpOut = next();	// every overlap through the reference's OverlapRange::dump
		else if (a == "--fasta-out") fastaOut = next();	// the loaded reads through SequenceContainer::writeFasta
		else if (a == "--query-limit") queryLimit = atol(next().c_str());
		else if (a == "--rc-queries") rcQueries = true;
		else if (a == "--find-all") findAll = true;	// OverlapContainer::findAllOverlaps (overlap.cpp:625-665)
		else if (a == "--queries") queriesFile = next();	// second container (ReadAligner-style)
		else if (a == "--only-max") onlyMax = atoi(next().c_str());
		else if (a == "--max-overhang") maxOverhang = atoi(next().c_str());
		else if (a == "--nucl-aln") nuclAln = atoi(next().c_str());
		else if (a == "--keep-aln") keepAln = atoi(next().c_str());
		else if (a == "--partition-bad") partitionBad = atoi(next().c_str());
		else if (a == "--max-div") maxDiv = strtof(next().c_str(), nullptr);
		else if (a == "--ksw-pairs")
		{
			// kernel-level pin of the banded affine-gap alignment: every line holds a target
			// and a query string ('^' in front); prints what the reference's getAlignmentCigarKsw
			// (alignment.cpp:102-216: ksw_extz2_sse, band 64 doubling, global backtrack, decoded CIGAR)
			// returns: the error rate's bit pattern and the CIGAR runs
			std::ifstream in(next());
			std::string qa, qb;
			while (in >> qa >> qb)
			{
				qa.erase(0, 1); qb.erase(0, 1);
				DnaSequence trg(qa), qry(qb);
				std::vector<CigOp> cigar;
				float err = getAlignmentCigarKsw(trg, 0, trg.length(), qry, 0, qry.length(), 1.0f, cigar);
				printf("%08x", fbits(err));
				for (auto& c : cigar) printf(" %d%c", c.len, c.op);
				printf("\n");
			}
			return 0;
		}
		else if (a == "--consensus-pairs")
		{
			// ConsensusGenerator::generateConsensuses (consensus_generator.cpp:18-126) over two-read "disjointigs":
			// every line of the file is one (left read, right read) pair overlapping over their whole length, i.e. one
			// getAlignmentCigarKsw + decodeCigar + switch-position search per pair, handed out over --threads worker
			// threads (give --threads BEFORE this flag).  Prints every consensus sequence; the time goes to stderr.
			std::ifstream in(next());
			std::string qa, qb;
			std::vector<ContigPath> contigs;
			while (in >> qa >> qb)
			{
				qa.erase(0, 1); qb.erase(0, 1);
				ContigPath path;
				path.name = "pair_" + std::to_string(contigs.size());
				path.sequences.push_back(DnaSequence(qa));
				path.sequences.push_back(DnaSequence(qb));
				OverlapRange ovlp(FastaRecord::ID_NONE, FastaRecord::ID_NONE, 0, 0, (int32_t)qa.size(), (int32_t)qb.size());
				ovlp.curEnd = (int32_t)qa.size(); ovlp.extEnd = (int32_t)qb.size();
				path.overlaps.push_back(ovlp);
				contigs.push_back(path);
			}
			if (!params.empty()) Config::addParameters(params);	// give --params BEFORE this flag (maximum_jump is read)
			Parameters::get().numThreads = threads;
			Parameters::get().kmerSize = 17;
			ConsensusGenerator gen;
			{
				// untimed warm-up on two pairs (a program with device seams starts its runtime here)
				std::vector<ContigPath> warm(contigs.begin(), contigs.begin() + std::min<size_t>(2, contigs.size()));
				gen.generateConsensuses(warm, false);
			}
			auto t0 = std::chrono::steady_clock::now();
			auto recs = gen.generateConsensuses(contigs, false);
			const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			for (auto& r : recs) printf("%s %s\n", r.description.c_str(), r.sequence.str().c_str());
			fprintf(stderr, "{\"pairs\": %zu, \"threads\": %d, \"consensus_s\": %.6f}\n", contigs.size(), threads, sec);
			return 0;
		}
		else if (a == "--edlib-pairs")
		{
			// kernel-level pin of the edit-distance restatements: every line of the file holds two
			// ACGT strings, each behind a '^' (so that an empty string is still a token); prints what the reference's own call (alignment.cpp:233-238: NW mode,
			// distance task, k = -1) returns for each pair
			std::ifstream in(next());
			std::string qa, qb;
			while (in >> qa >> qb)
			{
				qa.erase(0, 1); qb.erase(0, 1);
				auto cfg = edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_DISTANCE, nullptr, 0);
				auto res = edlibAlign(qa.c_str(), (int)qa.size(), qb.c_str(), (int)qb.size(), cfg);
				printf("%d\n", res.editDistance);
				edlibFreeAlignResult(res);
			}
			return 0;
		}
		else { fprintf(stderr, "unknown arg %s\n", a.c_str()); return 2; }
	}
	if (!config.empty()) Config::load(config);
	if (!params.empty()) Config::addParameters(params);
	Parameters::get().numThreads = threads;
	Parameters::get().kmerSize = (int)Config::get("kmer_size");
	Parameters::get().minimumOverlap = minOverlap;
	Parameters::get().unevenCoverage = false;

	using clk = std::chrono::steady_clock;
	auto t0 = clk::now();
	SequenceContainer readsContainer;
	readsContainer.loadFromFile(reads, minReadLen);
	readsContainer.buildPositionIndex();
	auto t1 = clk::now();

	VertexIndex vertexIndex(readsContainer, (int)Config::get("assemble_kmer_sample"));
	const int MIN_FREQ = 2;
	bool useMinimizers = Config::get("use_minimizers");
	if (useMinimizers)
	{
		vertexIndex.buildIndexMinimizers(1, (int)Config::get("minimizer_window"));
	}
	else
	{
		vertexIndex.countKmers();
		// freq of every k-mer position is lost after build; dump nothing here
		vertexIndex.buildIndexUnevenCoverage(MIN_FREQ,
				Config::get("meta_read_top_kmer_rate"),
				(int)Config::get("meta_read_filter_kmer_freq"));
	}
	auto t2 = clk::now();

	if (!indexOut.empty())
	{
		FILE* f = fopen(indexOut.c_str(), "w");
		fprintf(f, "S sampleRateBits %08x repFreq %zu numKmers %zu keys %zu rep %zu\n",
				fbits(vertexIndex._sampleRate), vertexIndex._repetitiveFrequency,
				(size_t)vertexIndex._kmerCounter._numKmers,
				vertexIndex._kmerIndex.size(), vertexIndex._repetitiveKmers.size());
		std::vector<std::pair<size_t, std::vector<size_t>>> keys;
		for (const auto& kv : vertexIndex._kmerIndex.lock_table())
		{
			Kmer k = kv.first;
			std::vector<size_t> pos;
			for (uint32_t j = 0; j < kv.second.size; ++j) pos.push_back(kv.second.data[j].get());
			keys.push_back({k.numRepr(), pos});
		}
		std::sort(keys.begin(), keys.end());
		for (auto& kv : keys)
		{
			fprintf(f, "K %zx %zu", kv.first, kv.second.size());
			for (auto p : kv.second) fprintf(f, " %zu", p);
			fprintf(f, "\n");
		}
		std::vector<size_t> rep;
		for (const auto& kv : vertexIndex._repetitiveKmers.lock_table())
		{
			Kmer k = kv.first;
			rep.push_back(k.numRepr());
		}
		std::sort(rep.begin(), rep.end());
		for (auto r : rep) fprintf(f, "R %zx\n", r);
		fclose(f);
	}

	// optional second container: queries differ from the indexed sequences, as in
	// ReadAligner::alignReads (src/repeat_graph/read_aligner.cpp:178-217); its ids
	// continue after the first container's (process-global g_nextSeqId)
	SequenceContainer queryContainer;
	if (!queriesFile.empty())
	{
		queryContainer.loadFromFile(queriesFile, minReadLen);
		queryContainer.buildPositionIndex();
	}
	const SequenceContainer& qc = queriesFile.empty() ? readsContainer : queryContainer;
	OverlapDetector ovlp(readsContainer, vertexIndex,
						 (int)Config::get("maximum_jump"),
						 Parameters::get().minimumOverlap,
						 maxOverhang >= 0 ? maxOverhang : (int)Config::get("maximum_overhang"),
						 /*store alignment*/ (bool)keepAln, /*only max*/ (bool)onlyMax,
						 /*div threshold*/ maxDiv,
						 nuclAln >= 0 ? (bool)nuclAln : (bool)Config::get("reads_base_alignment"),
						 /*partition bad*/ (bool)partitionBad,
						 (bool)Config::get("hpc_scoring_on"));
	OverlapContainer readOverlaps(ovlp, qc);
	float meanDiv = 0.0f;
	if (divMode == "assemble")
	{
		readOverlaps.estimateOverlaperParameters();
		readOverlaps.setDivergenceThreshold(
				(float)Config::get("assemble_ovlp_divergence"),
				(bool)Config::get("assemble_divergence_relative"));
		meanDiv = readOverlaps._meanTrueOvlpDiv;
	}
	auto t3 = clk::now();

	if (findAll)
	{
		// the repeat stage's use (repeat_graph.cpp:96-99): every forward sequence through lazySeqOverlaps,
		// ensureTransitivity(false), filterOverlaps -- all reference code above getSeqOverlaps.  The stored
		// lists are printed per sequence with their lines sorted: the order inside a list depends on the
		// iteration order of a concurrently filled hash table (overlap.cpp:580-585), i.e. on thread timing.
		readOverlaps.findAllOverlaps();
		FILE* f = ovlpOut.empty() ? stdout : fopen(ovlpOut.c_str(), "w");
		size_t total = 0;
		for (const auto& seq : qc.iterSeqs())
		{
			std::vector<std::string> lines;
			for (auto& o : readOverlaps.lazySeqOverlaps(seq.id))
			{
				char buf[256];
				snprintf(buf, sizeof(buf), "%u %d %d %d %u %d %d %d %d %08x %zu", o.curId._id, o.curBegin, o.curEnd, o.curLen,
						 o.extId._id, o.extBegin, o.extEnd, o.extLen, o.score, fbits(o.seqDivergence),
						 o.kmerMatches ? o.kmerMatches->size() : (size_t)0);
				lines.push_back(buf);
			}
			std::sort(lines.begin(), lines.end());
			fprintf(f, "# seq %u: %zu overlaps\n", seq.id._id, lines.size());
			for (auto& l : lines) fprintf(f, "%s\n", l.c_str());
			total += lines.size();
		}
		if (f != stdout) fclose(f);
		printf("{\"find_all_overlaps\": %zu, \"threads\": %d}\n", total, threads);
		return 0;
	}

	std::vector<FastaRecord::Id> queries;
	for (const auto& seq : qc.iterSeqs())
	{
		if (seq.id.strand() != rcQueries) queries.push_back(seq.id);
		if (queryLimit >= 0 && (long)queries.size() >= queryLimit) break;
	}
	std::vector<std::vector<OverlapRange>> results(queries.size());
	std::unordered_map<uint32_t, size_t> slot;
	for (size_t i = 0; i < queries.size(); ++i) slot[queries[i]._id] = i;
	std::atomic<size_t> queriedBp(0);
	std::function<void(const FastaRecord::Id&)> work =
	[&](const FastaRecord::Id& id)
	{
		results[slot[id._id]] = readOverlaps.quickSeqOverlaps(id, maxOverlaps, forceLocal);
		queriedBp += qc.seqLen(id);
	};
	processInParallel(queries, work, threads, false);
	auto t4 = clk::now();

	size_t total = 0;
	if (!ovlpOut.empty())
	{
		FILE* f = fopen(ovlpOut.c_str(), "w");
		fprintf(f, "# maxDivBits %08x meanDivBits %08x sampleRateBits %08x\n",
				fbits(ovlp._maxDivergence), fbits(meanDiv), fbits(vertexIndex._sampleRate));
		for (auto& vec : results)
			for (auto& o : vec)
			{
				fprintf(f, "%u %d %d %d %u %d %d %d %d %08x", o.curId._id, o.curBegin,
						o.curEnd, o.curLen, o.extId._id, o.extBegin, o.extEnd, o.extLen,
						o.score, fbits(o.seqDivergence));
				if (keepAln)
				{
					// (count, order-sensitive digest) of kmerMatches, see oracle.py match_hashes
					unsigned long long h = 0, i = 0;
					size_t cnt = o.kmerMatches ? o.kmerMatches->size() : 0;
					if (o.kmerMatches)
						for (auto& m : *o.kmerMatches)
							h += (++i) * ((unsigned long long)(long long)m.first * 0x9E3779B97F4A7C15ULL +
										  (unsigned long long)(long long)m.second + 1ULL);
					fprintf(f, " %zu %016llx", cnt, h);
				}
				fprintf(f, "\n");
				++total;
			}
		fclose(f);
	}
	else for (auto& vec : results) total += vec.size();
	if (!dumpOut.empty())
	{
		// the on-disk forms (overlap.h:227-236; read_aligner.cpp:321-339 puts "\tAln\t<edgeId>\t" in front)
		std::ofstream os(dumpOut);
		for (auto& vec : results)
			for (auto& o : vec) { o.dump(os, qc, readsContainer); os << "\n"; }
	}
	if (!fastaOut.empty())
		SequenceContainer::writeFasta(readsContainer.iterSeqs(), fastaOut, /*only positive strand*/ true);

	auto sec = [](clk::time_point a, clk::time_point b)
		{ return std::chrono::duration<double>(b - a).count(); };
	printf("{\"load_s\": %.4f, \"index_s\": %.4f, \"estimate_s\": %.4f, \"overlap_s\": %.4f, "
		   "\"queried_bp\": %zu, \"queries\": %zu, \"overlaps\": %zu, \"threads\": %d, "
		   "\"max_div_bits\": \"%08x\"}\n",
		   sec(t0, t1), sec(t1, t2), sec(t2, t3), sec(t3, t4), (size_t)queriedBp,
		   queries.size(), total, threads, fbits(ovlp._maxDivergence));
	return 0;
}
