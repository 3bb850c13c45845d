// TEST INFRASTRUCTURE -- CPU restatement ("oracle") of Flye 2.8.1's overlap hot
// path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library, and only as the checker.  The product (libflyegpu.so) never
// links, loads or calls anything in here.
//
// Parity status: PINNED -- checked record-for-record against the unmodified
// reference compiled from /root/reference (oracle/_ref/ref_dumper; see
// tests/test_oracle_vs_ref.py) and against the committed golden vectors that
// the same dumper produced (tests/golden/, tests/test_oracle_golden.py).  The
// reference's own tests hold no vectors for this path.
//
// This is a restatement of BEHAVIOUR on flat arrays, written from scratch; each
// function cites the reference lines it follows.  The unstable sorts call the
// real std::sort of this toolchain's libstdc++ (the same one the reference is
// built with), which is what makes it the ground truth for the emulated
// introsort of the HIP path (include/introsort_emul.h).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <array>
#include <deque>
#include <functional>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../include/flye_gpu.h"
#include "../include/introsort_emul.h"

namespace {

typedef uint64_t u64;
typedef uint32_t u32;

struct Hit { int32_t cur, ext; u32 extId; };

// open-addressing map canonical k-mer -> (offset,count | repetitive)
struct Slot { u64 key; u64 off; u32 cnt; u32 flag; };	// flag: 0 empty, 1 list, 2 repetitive

struct Index {
	std::vector<u64> keys;		// ascending canonical k-mers kept in _kmerIndex (may have empty lists)
	std::vector<u64> keyOff;	// keys.size()+1
	std::vector<u64> entries;	// global positions, ascending per key
	std::vector<u64> repetitive;// ascending
	std::vector<Slot> table;
	u64 mask = 0;
	float sampleRate = 1.0f;
	bool built = false;
};

struct Ctx {
	int k = 17;
	u32 n = 0;
	u32 firstId = 0;
	std::vector<u64> words, wordOff;
	std::vector<int32_t> len;
	// optional second container holding the queries (ReadAligner-style use,
	// src/repeat_graph/read_aligner.cpp:178-217); ids continue after the indexed ones
	bool hasQ = false;
	u32 qn = 0, qFirstId = 0;
	std::vector<u64> qWords, qWordOff;
	std::vector<int32_t> qLen;
	std::vector<u64> recOff;	// 2n+1 global offsets of records (fwd, rc interleaved)
	Index idx;
	// last overlap call
	std::vector<u64> outOff, statOff;
	std::vector<fg_overlap_rec> outRecs;
	std::vector<float> outStats;
	std::vector<uint8_t> outTrim;		// partition_bad_mappings: per output record
	std::vector<u64> outMatchOff;		// keep_alignment: per output record, in pairs
	std::vector<int32_t> outMatches;	// (cur, ext) pairs
	u64 cntKmers = 0, cntHits = 0, cntGroups = 0, cntDp = 0, cntBp = 0;
};

inline u64 mixHash(u64 x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
	x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
	return x;
}

// sequence.h:120-129 atRaw(), with the lazy reverse-complement flag
inline u32 baseAt(const Ctx& c, u32 read, int32_t pos, bool rc, bool fromQ = false)
{
	const std::vector<u64>& W = fromQ ? c.qWords : c.words;
	const u64 w0 = fromQ ? c.qWordOff[read] : c.wordOff[read];
	int32_t L = fromQ ? c.qLen[read] : c.len[read];
	int32_t p = rc ? L - 1 - pos : pos;
	u32 b = (u32)((W[w0 + (p >> 5)] >> ((p & 31) * 2)) & 3);
	return rc ? (~b & 3) : b;
}

// kmer.h:16-109, :131-204: all k-mers of a strand at positions 0..len-k-1 (the
// last one, at len-k, is never yielded), forward repr (first base most
// significant), canonical = min(fwd, revcomp), flipped = revcomp < fwd.
template <class F>
inline void forEachKmer(const Ctx& c, u32 read, bool rc, F f, bool fromQ = false)
{
	const int k = c.k;
	const int32_t L = fromQ ? c.qLen[read] : c.len[read];
	if (L < k) return;
	const int32_t nk = L - k;
	const u64 mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
	u64 fw = 0, rv = 0;
	for (int i = 0; i < k - 1; ++i)
	{
		u64 b = baseAt(c, read, i, rc, fromQ);
		fw = (fw << 2) | b;
		rv = (rv >> 2) | ((3 - b) << (2 * (k - 1)));
	}
	for (int32_t p = 0; p < nk; ++p)
	{
		u64 b = baseAt(c, read, p + k - 1, rc, fromQ);
		fw = ((fw << 2) | b) & mask;
		rv = (rv >> 2) | ((3 - b) << (2 * (k - 1)));
		f(p, fw, rv);
	}
}

// kmer.h:91-98 splitmix64 finaliser
inline u64 kmerHash(u64 x)
{
	u64 z = (x += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

struct KmerPos { u64 fw; int32_t pos; };

// kmer.h:206-262 yieldMinimizers: literal monotone-deque semantics
void minimizers(const Ctx& c, u32 read, int window, std::vector<KmerPos>& out)
{
	out.clear();
	if (window == 1)
	{
		forEachKmer(c, read, false, [&](int32_t p, u64 fw, u64) { out.push_back({fw, p}); });
		return;
	}
	struct QE { u64 fw; int32_t pos; u64 hash; };
	std::deque<QE> q;
	forEachKmer(c, read, false, [&](int32_t p, u64 fw, u64 rv)
	{
		u64 h = kmerHash(std::min(fw, rv));
		while (!q.empty() && q.back().hash > h) q.pop_back();
		q.push_back({fw, p, h});
		if (q.front().pos <= p - window)
		{
			while (q.front().pos <= p - window) q.pop_front();
			while (q.size() >= 2 && q[0].hash == q[1].hash) q.pop_front();
		}
		if (out.empty() || out.back().pos != q.front().pos) out.push_back({q.front().fw, q.front().pos});
	});
}

void parallelFor(u32 n, int threads, const std::function<void(u32, int)>& fn)
{
	if (threads <= 1 || n < 2)
	{
		for (u32 i = 0; i < n; ++i) fn(i, 0);
		return;
	}
	std::atomic<u32> next(0);
	std::vector<std::thread> pool;
	for (int t = 0; t < threads; ++t)
		pool.emplace_back([&, t]()
		{
			while (true)
			{
				u32 i = next.fetch_add(1);
				if (i >= n) return;
				fn(i, t);
			}
		});
	for (auto& th : pool) th.join();
}

// sequence_container.cpp:359-392 + .h:200-213: global position of (record, pos)
inline u64 globalPos(const Ctx& c, u32 rec, int32_t pos) { return c.recOff[rec] + (u64)pos; }

// sequence_container.h:220-235 seqPosition (binary search instead of hint table)
inline void seqPosition(const Ctx& c, u64 g, u32& rec, int32_t& pos, int32_t& len)
{
	size_t hi = std::upper_bound(c.recOff.begin(), c.recOff.end(), g) - c.recOff.begin();
	rec = (u32)(hi - 1);
	pos = (int32_t)(g - c.recOff[rec]);
	len = c.len[rec >> 1];
}

void buildTable(Index& ix)
{
	size_t need = (ix.keys.size() + ix.repetitive.size()) * 2 + 16;
	size_t cap = 16;
	while (cap < need) cap <<= 1;
	ix.table.assign(cap, Slot{0, 0, 0, 0});
	ix.mask = cap - 1;
	auto put = [&](u64 key, u64 off, u32 cnt, u32 flag)
	{
		u64 h = mixHash(key) & ix.mask;
		while (ix.table[h].flag) h = (h + 1) & ix.mask;
		ix.table[h] = Slot{key, off, cnt, flag};
	};
	for (size_t i = 0; i < ix.keys.size(); ++i)
		put(ix.keys[i], ix.keyOff[i], (u32)(ix.keyOff[i + 1] - ix.keyOff[i]), 1);
	for (u64 r : ix.repetitive) put(r, 0, 0, 2);
	ix.built = true;
}

inline const Slot* lookup(const Index& ix, u64 key)
{
	u64 h = mixHash(key) & ix.mask;
	while (ix.table[h].flag)
	{
		if (ix.table[h].key == key) return &ix.table[h];
		h = (h + 1) & ix.mask;
	}
	return nullptr;
}

// vertex_index.cpp:173-212 filterFrequentKmers on (key, capacity) pairs
void filterFrequent(std::vector<u64>& keys, std::vector<u32>& cap, int minCoverage, float rate,
					std::vector<u64>& repetitive, fg_index_stats& st)
{
	size_t total = 0, unique = 0;
	for (size_t i = 0; i < keys.size(); ++i)
		if (cap[i] >= (size_t)minCoverage) { total += cap[i]; unique += 1; }
	float mean = (float)total / (unique + 1);
	size_t repFreq = rate * mean;
	std::vector<u64> k2; std::vector<u32> c2;
	for (size_t i = 0; i < keys.size(); ++i)
	{
		if (cap[i] > repFreq) repetitive.push_back(keys[i]);
		else { k2.push_back(keys[i]); c2.push_back(cap[i]); }
	}
	keys.swap(k2); cap.swap(c2);
	st.mean_frequency = mean;
	st.repetitive_frequency = repFreq;
}

// run-length encode a sorted vector
void rle(const std::vector<u64>& v, std::vector<u64>& keys, std::vector<u32>& cnt)
{
	keys.clear(); cnt.clear();
	for (size_t i = 0; i < v.size();)
	{
		size_t j = i;
		while (j < v.size() && v[j] == v[i]) ++j;
		keys.push_back(v[i]); cnt.push_back((u32)(j - i));
		i = j;
	}
}

struct Sel { u64 canon; u64 gpos; u32 freq; };

// vertex_index.cpp:316-358 yieldFrequentKmers for one forward read; freq lookups
// through a sorted (key,count) table = KmerCounter::getFreq (:593-616, exact count)
void selectFrequent(const Ctx& c, u32 read, const std::vector<u64>& cKeys, const std::vector<u32>& cCnt,
					float selectRate, int tandemFreq, std::vector<Sel>& out)
{
	out.clear();
	struct P { u64 canon; int32_t pos; bool flip; u32 freq; };
	std::vector<P> all;
	forEachKmer(c, read, false, [&](int32_t p, u64 fw, u64 rv)
	{
		bool flip = rv < fw;
		u64 cn = flip ? rv : fw;
		size_t j = std::lower_bound(cKeys.begin(), cKeys.end(), cn) - cKeys.begin();
		all.push_back({cn, p, flip, cCnt[j]});
	});
	if (all.empty()) return;
	std::vector<u32> fr(all.size());
	for (size_t i = 0; i < all.size(); ++i) fr[i] = all[i].freq;
	std::sort(fr.begin(), fr.end(), [](u32 a, u32 b) { return a > b; });
	const size_t maxKmers = selectRate * all.size();
	const u32 minFreq = fr[maxKmers];
	std::unordered_map<u64, u32> local;
	for (auto& p : all) ++local[p.canon];
	const int32_t L = c.len[read];
	for (auto& p : all)
	{
		if (p.freq < minFreq) continue;
		if (tandemFreq > 0 && local[p.canon] > (u32)tandemFreq) continue;
		// canonical orientation: vertex_index.cpp:76-85
		u32 rec = 2 * read + (p.flip ? 1 : 0);
		int32_t pos = p.flip ? L - p.pos - c.k : p.pos;
		out.push_back({p.canon, globalPos(c, rec, pos), p.freq});
	}
}

void finishIndex(Ctx& c, std::vector<u64>& keys, std::vector<std::pair<u64, u64>>& ent, fg_index_stats& st)
{
	// ent = (canon, gpos) for every accepted position; keys = _kmerIndex keys (ascending)
	std::sort(ent.begin(), ent.end());
	Index& ix = c.idx;
	ix.keys = keys;
	ix.keyOff.assign(keys.size() + 1, 0);
	ix.entries.resize(ent.size());
	size_t e = 0;
	for (size_t i = 0; i < keys.size(); ++i)
	{
		ix.keyOff[i] = e;
		while (e < ent.size() && ent[e].first == keys[i]) { ix.entries[e] = ent[e].second; ++e; }
	}
	ix.keyOff[keys.size()] = e;
	st.selected_kmers = keys.size();
	st.index_entries = ent.size();
	st.repetitive_kmers = ix.repetitive.size();
	buildTable(ix);
}

} // namespace

extern "C" {

struct fo_ctx { Ctx c; };

fo_ctx* fo_create(int k) { fo_ctx* h = new fo_ctx; h->c.k = k; return h; }
void fo_destroy(fo_ctx* h) { delete h; }

int fo_set_reads(fo_ctx* h, u32 n, const u64* words, const u64* wordOff, const int32_t* len, u32 firstId)
{
	Ctx& c = h->c;
	c.n = n; c.firstId = firstId;
	c.wordOff.assign(wordOff, wordOff + n + 1);
	c.words.assign(words, words + wordOff[n]);
	c.len.assign(len, len + n);
	c.recOff.assign(2 * (size_t)n + 1, 0);
	u64 off = 0;
	for (u32 i = 0; i < n; ++i)
	{
		c.recOff[2 * i] = off; off += len[i];
		c.recOff[2 * i + 1] = off; off += len[i];
	}
	c.recOff[2 * (size_t)n] = off;
	c.idx = Index();
	return 0;
}

// queries from a second container (empty n resets to "queries = indexed reads")
int fo_set_queries(fo_ctx* h, u32 n, const u64* words, const u64* wordOff, const int32_t* len, u32 firstId)
{
	Ctx& c = h->c;
	c.hasQ = n > 0;
	c.qn = n; c.qFirstId = firstId;
	if (n)
	{
		c.qWordOff.assign(wordOff, wordOff + n + 1);
		c.qWords.assign(words, words + wordOff[n]);
		c.qLen.assign(len, len + n);
	}
	return 0;
}

// countKmers + buildIndexUnevenCoverage (vertex_index.cpp:19-125, :499-590)
int fo_build_index_solid(fo_ctx* h, int32_t minFreq, float selectRate, int32_t tandemFreq,
						 float repeatRate, float sampleRateInit, int threads, fg_index_stats* st)
{
	Ctx& c = h->c;
	if (c.k > 17) return FG_ERR_KMER_SIZE;
	memset(st, 0, sizeof(*st));
	c.idx = Index();
	c.idx.sampleRate = sampleRateInit;
	// exact counts of canonical k-mers over forward reads
	std::vector<u64> all;
	for (u32 r = 0; r < c.n; ++r)
		forEachKmer(c, r, false, [&](int32_t, u64 fw, u64 rv) { all.push_back(std::min(fw, rv)); });
	std::sort(all.begin(), all.end());
	std::vector<u64> cKeys; std::vector<u32> cCnt;
	rle(all, cKeys, cCnt);
	std::vector<u64>().swap(all);
	st->total_kmers = cKeys.size();

	// pass 1 (:41-59): capacities of selected k-mers with freq >= minFreq
	std::vector<std::vector<Sel>> sel(c.n);
	parallelFor(c.n, threads, [&](u32 r, int) { selectFrequent(c, r, cKeys, cCnt, selectRate, tandemFreq, sel[r]); });
	std::vector<u64> capList;
	for (auto& v : sel) for (auto& s : v) if (s.freq >= (u32)minFreq) capList.push_back(s.canon);
	std::sort(capList.begin(), capList.end());
	std::vector<u64> keys; std::vector<u32> cap;
	rle(capList, keys, cap);
	filterFrequent(keys, cap, minFreq, repeatRate, c.idx.repetitive, *st);
	for (u32 cp : cap) if ((size_t)cp + 1 > 32 * 1024 * 1024 / 5) return FG_ERR_KMER_TOO_FREQUENT; // :370-373

	// pass 2 (:65-104)
	std::vector<std::pair<u64, u64>> ent;
	for (auto& v : sel)
		for (auto& s : v)
		{
			if (s.freq < (u32)minFreq || s.freq > st->repetitive_frequency) continue;
			if (!std::binary_search(keys.begin(), keys.end(), s.canon)) continue;
			ent.push_back({s.canon, s.gpos});
		}
	finishIndex(c, keys, ent, *st);
	st->sample_rate = c.idx.sampleRate;
	return 0;
}

// buildIndexMinimizers (vertex_index.cpp:389-483)
int fo_build_index_minimizers(fo_ctx* h, int32_t minCoverage, int32_t window, float repeatRate,
							  int threads, fg_index_stats* st)
{
	Ctx& c = h->c;
	if (window < 1) return FG_ERR_ARG;
	memset(st, 0, sizeof(*st));
	c.idx = Index();
	size_t totalLen = 0;
	for (u32 r = 0; r < c.n; ++r) totalLen += c.len[r];
	std::vector<std::vector<KmerPos>> mins(c.n);
	parallelFor(c.n, threads, [&](u32 r, int) { minimizers(c, r, window, mins[r]); });
	const int k = c.k;
	std::vector<u64> capList;
	auto canonOf = [k](u64 fw, bool& flip)
	{
		u64 rv = 0, t = fw;
		for (int i = 0; i < k; ++i) { rv = (rv << 2) | (~t & 3); t >>= 2; }
		flip = rv < fw;
		return flip ? rv : fw;
	};
	for (auto& v : mins) for (auto& m : v) { bool f; capList.push_back(canonOf(m.fw, f)); }
	std::sort(capList.begin(), capList.end());
	std::vector<u64> keys; std::vector<u32> cap;
	rle(capList, keys, cap);
	filterFrequent(keys, cap, minCoverage, repeatRate, c.idx.repetitive, *st);
	for (u32 cp : cap) if ((size_t)cp + 1 > 32 * 1024 * 1024 / 5) return FG_ERR_KMER_TOO_FREQUENT;
	std::vector<std::pair<u64, u64>> ent;
	for (u32 r = 0; r < c.n; ++r)
		for (auto& m : mins[r])
		{
			bool flip; u64 cn = canonOf(m.fw, flip);
			if (!std::binary_search(keys.begin(), keys.end(), cn)) continue;	// repetitive (:442)
			u32 rec = 2 * r + (flip ? 1 : 0);
			int32_t pos = flip ? c.len[r] - m.pos - k : m.pos;
			ent.push_back({cn, globalPos(c, rec, pos)});
		}
	finishIndex(c, keys, ent, *st);
	float rate = (float)totalLen / ent.size();	// :480-482
	c.idx.sampleRate = rate;
	st->sample_rate = rate;
	return 0;
}

// minimizer sketch of one forward read, for kernel-level parity tests
int64_t fo_minimizers(fo_ctx* h, u32 read, int window, int32_t* posOut, int64_t cap)
{
	std::vector<KmerPos> v;
	minimizers(h->c, read, window, v);
	for (size_t i = 0; i < v.size() && (int64_t)i < cap; ++i) posOut[i] = v[i].pos;
	return (int64_t)v.size();
}

// import an index built elsewhere (entries as (record<<32|pos)); used by bench's
// cpu_baseline leg so the CPU times the overlap stage on the same index
int fo_import_index(fo_ctx* h, u64 nKeys, const u64* keys, const u64* keyOff, const u64* entries,
					u64 nRep, const u64* rep, float sampleRate)
{
	Ctx& c = h->c;
	Index& ix = c.idx;
	ix = Index();
	ix.keys.assign(keys, keys + nKeys);
	ix.keyOff.assign(keyOff, keyOff + nKeys + 1);
	ix.entries.resize(keyOff[nKeys]);
	for (u64 i = 0; i < keyOff[nKeys]; ++i)
		ix.entries[i] = c.recOff[entries[i] >> 32] + (entries[i] & 0xffffffffu);
	ix.repetitive.assign(rep, rep + nRep);
	ix.sampleRate = sampleRate;
	buildTable(ix);
	return 0;
}

// same export form as fg_export_index
int fo_export_index(fo_ctx* h, u64* nKeys, u64* nEntries, u64* nRep, u64* keys, u64* keyOff,
					u64* entries, u64* rep)
{
	Ctx& c = h->c;
	const Index& ix = c.idx;
	if (!ix.built) return FG_ERR_STATE;
	*nKeys = ix.keys.size(); *nEntries = ix.entries.size(); *nRep = ix.repetitive.size();
	if (keys) memcpy(keys, ix.keys.data(), ix.keys.size() * 8);
	if (keyOff) memcpy(keyOff, ix.keyOff.data(), ix.keyOff.size() * 8);
	if (entries)
		for (size_t i = 0; i < ix.entries.size(); ++i)
		{
			u32 rec; int32_t pos, len;
			seqPosition(c, ix.entries[i], rec, pos, len);
			entries[i] = ((u64)rec << 32) | (u32)pos;
		}
	if (rep) memcpy(rep, ix.repetitive.data(), ix.repetitive.size() * 8);
	return 0;
}

} // extern "C"

namespace {

// alignment.cpp:52-70 homopolymerCompression of seq[start, start+length)
void extractSeq(const Ctx& c, u32 recIdx, int32_t start, int32_t length, bool hpc, std::vector<uint8_t>& out,
				bool fromQ = false)
{
	out.clear();
	const u32 read = recIdx >> 1;
	const bool rc = recIdx & 1;
	for (int32_t i = 0; i < length; ++i)
	{
		uint8_t b = (uint8_t)baseAt(c, read, start + i, rc, fromQ);
		if (!hpc || i == 0 || out.back() != b) out.push_back(b);
	}
}

// exact unit-cost global edit distance (what edlibAlign(NW, DISTANCE, k=-1)
// returns, edlib.cpp:141-296): Ukkonen band doubling over a plain scalar DP.  Kept as the
// slow, obviously-right form the bit-vector version below is tested against.
int editDistanceDP(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b)
{
	const int n = (int)a.size(), m = (int)b.size();
	if (n == 0) return m;
	if (m == 0) return n;
	const int INF = 1 << 29;
	int kband = std::max(64, std::abs(n - m) + 1);
	std::vector<int> prev, cur;
	while (true)
	{
		// cells (i,j) with |i-j| <= kband
		prev.assign(m + 2, INF); cur.assign(m + 2, INF);
		for (int j = 0; j <= std::min(m, kband); ++j) prev[j] = j;
		for (int i = 1; i <= n; ++i)
		{
			int lo = std::max(0, i - kband), hi = std::min(m, i + kband);
			if (lo > 0) cur[lo - 1] = INF;
			for (int j = lo; j <= hi; ++j)
			{
				int v;
				if (j == 0) v = i;
				else
				{
					v = prev[j - 1] + (a[i - 1] != b[j - 1]);
					if (prev[j] + 1 < v) v = prev[j] + 1;
					if (j > lo && cur[j - 1] + 1 < v) v = cur[j - 1] + 1;
				}
				cur[j] = v;
			}
			if (hi < m) cur[hi + 1] = INF;
			prev.swap(cur);
		}
		int d = prev[m];
		if (d <= kband) return d;
		kband *= 2;
	}
}

// Upper bound D' >= D of the global edit distance of a (rows) and b (columns), with D' == D
// whenever D <= k: Myers' bit-vector recurrence (Myers 1999, in Hyyro's block form with a
// horizontal carry in {-1, 0, +1}) over 64-row blocks, each block swept only over the columns
// its rows can reach on a path of cost <= k (diagonals J - i in [dlo, dhi], Ukkonen).  Cells
// outside a block's column range count as "reached by +1 steps" from the computed region, so
// every value is the cost of a real path.  This is the published algorithm edlib implements
// (edlib.cpp:141-296 drives it with k = 64, 128, ...); the row-block-major sweep with the
// bottom row's deltas kept in place is our own arrangement (the HIP kernel uses the same one
// with 64 blocks per wave).
int editDistanceBand(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b, int k)
{
	const int n = (int)a.size(), m = (int)b.size();
	const int delta = m - n;
	const int dhi = (k + delta) / 2, dlo = -((k - delta) / 2);	// k >= |delta|
	std::vector<int8_t> hrow(m, 1);		// D'[r][j+1] - D'[r][j] along the last finished block's bottom row
	int prevCh = 0;						// columns [0, prevCh) of hrow are computed values, the rest is "+1"
	long long T = 0;					// D'[r0][cl]
	for (int r0 = 0; r0 < n; r0 += 64)
	{
		const int r1 = std::min(n, r0 + 64), rows = r1 - r0;
		const int cl = std::max(0, r0 + dlo), ch = std::min(m, r1 + dhi);
		const int nextCl = r1 < n ? std::max(0, r1 + dlo) : m;
		u64 peq[4] = {0, 0, 0, 0};
		for (int i = 0; i < rows; ++i) peq[a[r0 + i] & 3] |= 1ULL << i;
		u64 Pv = ~0ULL, Mv = 0;
		long long acc = T + rows;		// D'[r1][cl]: the left boundary is a column of +1 steps
		const u64 top = 1ULL << (rows - 1);
		for (int j = cl; j < ch; ++j)
		{
			const int hin = j < prevCh ? hrow[j] : 1;
			u64 Eq = peq[b[j] & 3];
			const u64 Xv = Eq | Mv;
			if (hin < 0) Eq |= 1ULL;
			const u64 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
			u64 Ph = Mv | ~(Xh | Pv);
			u64 Mh = Pv & Xh;
			const int hout = ((Ph & top) ? 1 : 0) - ((Mh & top) ? 1 : 0);
			Ph <<= 1; Mh <<= 1;
			if (hin < 0) Mh |= 1ULL; else if (hin > 0) Ph |= 1ULL;
			Pv = Mh | ~(Xv | Ph);
			Mv = Ph & Xv;
			hrow[j] = (int8_t)hout;
			if (j < nextCl) acc += hout;
		}
		T = acc;	// D'[r1][nextCl] (r1 == n: D'[n][m])
		prevCh = ch;
	}
	return (int)T;
}

// the exact distance: band doubling from k0 until the bound is met (edlib.cpp:194-212 starts
// at 64; any start gives the same value)
int editDistance(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b, int k0 = 64)
{
	const int n = (int)a.size(), m = (int)b.size();
	if (n == 0) return m;
	if (m == 0) return n;
	long long k = std::max(std::max(k0, 1), std::abs(n - m));
	while (true)
	{
		const int d = editDistanceBand(a, b, (int)std::min<long long>(k, n + m));
		if (d <= k) return d;
		k *= 2;
	}
}

struct Cand {
	int32_t curBegin, curEnd, extBegin, extEnd, score, chainLength, filtered;
	float div;
	std::vector<std::pair<int32_t, int32_t>> matches;	// kmerMatches (keep_alignment)
};

// overlap.cpp:29-69 overlapTest
bool overlapTest(const fg_detector_params& P, u32 curId, u32 extId, int32_t curLen, int32_t extLen,
				 const Cand& o, bool forceLocal)
{
	int32_t curRange = o.curEnd - o.curBegin, extRange = o.extEnd - o.extBegin;
	if (curRange < P.min_overlap || extRange < P.min_overlap) return false;
	const float OVLP_DIV = 0.5;
	float lengthDiff = abs(curRange - extRange);
	if (lengthDiff > OVLP_DIV * std::min(curRange, extRange)) return false;
	if (curId == extId)
	{
		int32_t inter = std::min(o.curEnd, o.extEnd) - std::max(o.curBegin, o.extBegin);
		if (inter > curRange / 2) return false;
	}
	if (curId == (extId ^ 1))
	{
		int32_t inter = std::min(o.curEnd, extLen - o.extBegin) - std::max(o.curBegin, extLen - o.extEnd);
		if (inter > curRange / 2) return false;
	}
	if (!forceLocal && P.max_overhang > 0)
	{
		int32_t ovh = std::max(std::min(o.curBegin, o.extBegin),
							   std::min(curLen - o.curEnd, extLen - o.extEnd));
		if (ovh > P.max_overhang) return false;
	}
	return true;
}

struct Scratch {
	std::vector<Hit> hits, group;
	std::vector<int32_t> score, back, filtered;
	std::vector<size_t> order;
	std::vector<uint8_t> sa, sb;
};

// overlap.cpp:99-508 getSeqOverlaps for one FastaRecord (recIdx = id - firstId)
void seqOverlaps(const Ctx& c, const fg_detector_params& P, u32 recIdx, bool forceLocal, int maxOverlaps,
				 Scratch& S, std::vector<fg_overlap_rec>& out, std::vector<float>& stats, u64* counters,
				 std::vector<std::vector<std::pair<int32_t, int32_t>>>* outMatches = nullptr,
				 std::vector<uint8_t>* outTrim = nullptr)
{
	const Index& ix = c.idx;
	const int k = c.k;
	const u32 read = recIdx >> 1;
	const bool qrc = recIdx & 1;
	const bool fromQ = c.hasQ;
	const int32_t curLen = fromQ ? c.qLen[read] : c.len[read];
	const u32 curId = (fromQ ? c.qFirstId : c.firstId) + recIdx;
	const float minKmerSurvivalRate = 0.01;
	const float LG_GAP = 2, SM_GAP = 0.5;
	S.hits.clear(); S.filtered.clear();

	// seed collection (:176-196); lookups as vertex_index.h:139-246
	forEachKmer(c, read, qrc, [&](int32_t p, u64 fw, u64 rv)
	{
		++counters[0];
		bool flip = rv < fw;
		const Slot* s = lookup(ix, flip ? rv : fw);
		if (!s) return;
		if (s->flag == 2) { S.filtered.push_back(p); return; }
		for (u32 j = 0; j < s->cnt; ++j)
		{
			u32 rec; int32_t pos, len;
			seqPosition(c, ix.entries[s->off + j], rec, pos, len);
			if (flip) { rec ^= 1; pos = len - pos - k; }
			if (c.firstId + rec == curId && pos == p) continue;	// no trivial matches
			S.hits.push_back({p, pos, c.firstId + rec});
		}
	}, fromQ);
	counters[1] += S.hits.size();

	std::sort(S.hits.begin(), S.hits.end(), [](const Hit& a, const Hit& b)
			  { return a.extId != b.extId ? a.extId < b.extId : a.cur < b.cur; });

	const int STAT_WND = 10000;
	struct Wnd { int32_t range; float div; };
	std::vector<Wnd> wnd(curLen / STAT_WND + 1, Wnd{0, 0.0f});
	size_t detected = 0;

	size_t gEnd = 0;
	bool tieInQuery = false;	// FO_STATS: equal (extId, curPos) keys inside a group that reaches the DP
	const size_t nh = S.hits.size();
	while (gEnd < nh)
	{
		if (maxOverlaps != 0 && detected >= (size_t)maxOverlaps) break;
		size_t gBeg = gEnd;
		size_t unique = 0;
		int32_t prev = 0;
		while (gEnd < nh && S.hits[gBeg].extId == S.hits[gEnd].extId)
		{
			if (S.hits[gEnd].cur != prev) { ++unique; prev = S.hits[gEnd].cur; }
			++gEnd;
		}
		if (unique < minKmerSurvivalRate * P.min_overlap) continue;

		S.group.assign(S.hits.begin() + gBeg, S.hits.begin() + gEnd);
		std::vector<Hit>& M = S.group;
		const u32 extId = M.front().extId;
		const u32 extRec = extId - c.firstId;
		const int32_t extLen = c.len[extRec >> 1];

		int32_t minCur = M.front().cur, maxCur = M.back().cur;
		int32_t minExt = INT32_MAX, maxExt = INT32_MIN;
		for (auto& m : M) { minExt = std::min(minExt, m.ext); maxExt = std::max(maxExt, m.ext); }
		if (maxCur - minCur < P.min_overlap || maxExt - minExt < P.min_overlap) continue;
		if (P.max_overhang > 0 && !forceLocal)
		{
			if (std::min(minCur, minExt) > P.max_overhang) continue;
			if (std::min(curLen - maxCur, extLen - maxExt) > P.max_overhang) continue;
		}
		++counters[2];
		counters[3] += M.size();
		for (size_t t = 1; t < M.size(); ++t) if (M[t].cur == M[t - 1].cur) { tieInQuery = true; break; }

		const int32_t n = (int32_t)M.size();
		S.score.assign(n, 0);
		S.back.assign(n, -1);
		const bool extSorted = extLen > curLen;
		if (extSorted && getenv("FO_STATS4"))
		{
			static std::atomic<unsigned long long> A[4];	// ext-sorted groups, elems, already strictly ascending groups, elems
			bool asc = true;
			for (size_t t = 1; t < M.size(); ++t) if (M[t].ext <= M[t - 1].ext) { asc = false; break; }
			A[0] += 1; A[1] += M.size(); if (asc) { A[2] += 1; A[3] += M.size(); }
			if ((A[0] & 0x1FFF) == 0) fprintf(stderr, "ext-sorted groups %llu (%llu el): already strictly ascending %llu (%llu el)\n",
				(unsigned long long)A[0], (unsigned long long)A[1], (unsigned long long)A[2], (unsigned long long)A[3]);
		}
		if (extSorted)
			std::sort(M.begin(), M.end(), [](const Hit& a, const Hit& b) { return a.ext < b.ext; });

		// chaining DP (:277-323)
		for (int32_t i = 1; i < n; ++i)
		{
			int32_t maxScore = 0, maxId = 0;
			const int32_t curNext = M[i].cur, extNext = M[i].ext;
			int32_t scanned = 0;
			int32_t validBefore = 0; bool brokeClose = false;	// FO_STATS6: valid candidates met before the closing one
			for (int32_t j = i - 1; j >= 0; --j)
			{
				++scanned;
				const int32_t curPrev = M[j].cur, extPrev = M[j].ext;
				if (0 < curNext - curPrev && curNext - curPrev < P.max_jump &&
					0 < extNext - extPrev && extNext - extPrev < P.max_jump)
				{
					int32_t matchScore = std::min(std::min(curNext - curPrev, extNext - extPrev), k);
					int32_t jumpDiv = abs((curNext - curPrev) - (extNext - extPrev));
					int32_t gapCost = (jumpDiv > 100 ? LG_GAP : SM_GAP) * jumpDiv;
					int32_t nextScore = S.score[j] + matchScore - gapCost;
					if (nextScore > maxScore)
					{
						maxScore = nextScore;
						maxId = j;
						if (jumpDiv == 0 && curNext - curPrev < k) { brokeClose = true; break; }
					}
					++validBefore;
				}
				if (extSorted && extNext - extPrev > P.max_jump) break;
				if (!extSorted && curNext - curPrev > P.max_jump) break;
			}
			counters[4] += scanned; counters[5] += scanned > 16; counters[6] += scanned > 64;
			if (getenv("FO_STATS6"))
			{
				// how many elements end their scan at a same-diagonal predecessor closer than k with NO valid candidate in
				// between (then the outcome is known without the scan), by how far back that predecessor is
				static std::atomic<unsigned long long> F[8];	// [0] elements, [1] at i-1, [2] at i-2, [3] i-3, [4] i-4, [5] 5..8, [6] further
				F[0] += 1;
				if (brokeClose && validBefore == 0)
					F[scanned == 1 ? 1 : scanned == 2 ? 2 : scanned == 3 ? 3 : scanned == 4 ? 4 : scanned <= 8 ? 5 : 6] += 1;
				if ((F[0] & 0x3FFFFF) == 0)
				{
					fprintf(stderr, "closing predecessor with nothing valid in between:");
					for (int b = 1; b < 7; ++b) fprintf(stderr, " %.3f", (double)F[b] / F[0]);
					fprintf(stderr, " of %llu elements (at i-1, i-2, i-3, i-4, 5..8 back, further)\n", (unsigned long long)F[0]);
				}
			}
			if (getenv("FO_STATS3"))
			{
				static std::atomic<unsigned long long> L[12];	// elements by scan length 1..10, >10; [11] = count
				L[scanned <= 10 ? scanned - 1 : 10] += 1;
				if (((L[11] += 1) & 0x3FFFFF) == 0)
				{
					fprintf(stderr, "scan length share:");
					for (int b = 0; b < 11; ++b) fprintf(stderr, " %.3f", (double)L[b] / L[11]);
					fprintf(stderr, "\n");
				}
			}
			if (getenv("FO_STATS2"))
			{
				static std::atomic<unsigned long long> H[8];	// candidate steps at depth <=8,<=16,<=32,<=64,<=128,<=256,more
				int lim[7] = {8, 16, 32, 64, 128, 256, 1 << 30}, prev = 0;
				for (int b = 0; b < 7; ++b) { int hi = std::min(scanned, lim[b]); if (hi > prev) H[b] += hi - prev; prev = std::max(prev, hi); }
				H[7] += 1;
				if ((H[7] & 0xFFFFF) == 0)
				{
					fprintf(stderr, "steps by depth:");
					for (int b = 0; b < 7; ++b) fprintf(stderr, " %.2f", (double)H[b] / H[7]);
					fprintf(stderr, " per element\n");
				}
			}
			S.score[i] = std::max(maxScore, k);
			if (maxScore > k) S.back[i] = maxId;
		}

		if (getenv("FO_STATS4"))
		{
			// how often are the two per-group sorts free of ties that could matter?  (a) ext re-sort: duplicate
			// extPos; (b) score-order sort: two elements with a back pointer sharing a score
			static std::atomic<unsigned long long> G[8];	// groups, elems, extSorted groups, their elems, ext-tie groups, their elems, score-tie groups, their elems
			G[0] += 1; G[1] += n;
			if (extSorted)
			{
				G[2] += 1; G[3] += n;
				bool tie = false;
				for (int32_t t = 1; t < n; ++t) if (M[t].ext == M[t - 1].ext) { tie = true; break; }
				if (tie) { G[4] += 1; G[5] += n; }
			}
			std::vector<int32_t> live;
			for (int32_t t = 0; t < n; ++t) if (S.back[t] != -1) live.push_back(S.score[t]);
			std::sort(live.begin(), live.end());
			bool stie = false;
			for (size_t t = 1; t < live.size(); ++t) if (live[t] == live[t - 1]) { stie = true; break; }
			if (stie) { G[6] += 1; G[7] += n; }
			if ((G[0] & 0x3FFF) == 0)
				fprintf(stderr, "groups %llu elems %llu | extSorted %llu (%llu el), with ext ties %llu (%llu el) | score ties among back!=-1: %llu (%llu el)\n",
						(unsigned long long)G[0], (unsigned long long)G[1], (unsigned long long)G[2], (unsigned long long)G[3],
						(unsigned long long)G[4], (unsigned long long)G[5], (unsigned long long)G[6], (unsigned long long)G[7]);
		}

		// backtracking in descending score order (:326-427)
		S.order.resize(n);
		std::iota(S.order.begin(), S.order.end(), 0);
		const std::vector<int32_t>& sc = S.score;
		std::sort(S.order.begin(), S.order.end(), [&sc](size_t a, size_t b) { return sc[a] > sc[b]; });

		if (getenv("FO_STATS5"))
		{
			// would the visiting order ever have to choose between two elements of equal score that are BOTH still
			// unconsumed when their score comes up?  (dry run on a copy of the back pointers)
			static std::atomic<unsigned long long> H[4];	// groups, elems, ambiguous groups, their elems
			std::vector<int32_t> bk(S.back.begin(), S.back.begin() + n);
			bool amb = false;
			for (int32_t a = 0; a < n && !amb; )
			{
				int32_t b = a;
				while (b + 1 < n && sc[S.order[b + 1]] == sc[S.order[a]]) ++b;
				int alive = 0;
				for (int32_t t = a; t <= b; ++t) alive += bk[S.order[t]] != -1;
				if (alive >= 2) amb = true;
				for (int32_t t = a; t <= b; ++t)
				{
					int32_t pos = (int32_t)S.order[t];
					if (bk[pos] == -1) continue;
					while (pos != -1) { const int32_t np = bk[pos]; bk[pos] = -1; pos = np; }
				}
				a = b + 1;
			}
			H[0] += 1; H[1] += n; if (amb) { H[2] += 1; H[3] += n; }
			if ((H[0] & 0x3FFF) == 0)
				fprintf(stderr, "groups %llu (%llu el): two live elements of one score at visiting time in %llu (%llu el)\n",
						(unsigned long long)H[0], (unsigned long long)H[1], (unsigned long long)H[2], (unsigned long long)H[3]);
		}
		std::vector<Cand> cands;
		for (size_t oi = 0; oi < S.order.size(); ++oi)
		{
			int32_t start = (int32_t)S.order[oi];
			if (S.back[start] == -1) continue;
			int32_t last = start, first = 0, chainLength = 0;
			int32_t pos = start;
			std::vector<std::pair<int32_t, int32_t>> kmerMatches;
			while (pos != -1)
			{
				first = pos;
				++chainLength;
				// chain thinned to one match per > k query bases (:368-377)
				if (P.keep_alignment &&
					(kmerMatches.empty() || kmerMatches.back().first - M[pos].cur > k))
					kmerMatches.emplace_back(M[pos].cur, M[pos].ext);
				int32_t np = S.back[pos];
				S.back[pos] = -1;
				pos = np;
			}
			Cand o;
			o.curBegin = M[first].cur; o.extBegin = M[first].ext;
			o.curEnd = M[last].cur + k - 1; o.extEnd = M[last].ext + k - 1;
			o.score = S.score[last] - S.score[first] + k - 1;
			o.chainLength = chainLength;
			if (!overlapTest(P, curId, extId, curLen, extLen, o, forceLocal)) continue;
			if (P.keep_alignment)	// :398-405
			{
				kmerMatches.emplace_back(o.curBegin, o.extBegin);
				std::reverse(kmerMatches.begin(), kmerMatches.end());
				kmerMatches.emplace_back(o.curEnd, o.extEnd);
				o.matches.swap(kmerMatches);
			}
			int32_t fpos = 0;
			for (int32_t p : S.filtered)
			{
				if (p < o.curBegin) continue;
				if (p > o.curEnd) break;
				++fpos;
			}
			o.filtered = fpos;
			float normLen = std::max(o.curEnd - o.curBegin, o.extEnd - o.extBegin) - fpos;
			float matchRate = (float)chainLength * ix.sampleRate / normLen;
			matchRate = std::min(matchRate, 1.0f);
			o.div = std::log(1 / matchRate) / k;
			cands.push_back(o);
		}

		// primary selection (:431-458)
		std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.score > b.score; });
		std::vector<Cand> prim;
		if (P.only_max_ext) { if (!cands.empty()) prim.push_back(cands.front()); }
		else
		{
			for (auto& o : cands)
			{
				bool contained = false;
				for (auto& p : prim)
					if (p.curBegin <= o.curBegin && o.curEnd <= p.curEnd &&
						p.extBegin <= o.extBegin && o.extEnd <= p.extEnd && p.score > o.score)
					{ contained = true; break; }
				if (!contained) prim.push_back(o);
			}
		}

		// divergence gate (:461-493)
		for (auto& o : prim)
		{
			fg_overlap_rec r;
			r.cur_id = curId; r.ext_id = extId;
			r.cur_begin = o.curBegin; r.cur_end = o.curEnd; r.cur_len = curLen;
			r.ext_begin = o.extBegin; r.ext_end = o.extEnd; r.ext_len = extLen;
			r.score = o.score; r.chain_length = o.chainLength; r.filtered_positions = o.filtered;
			r.edit_distance = -1; r.hpc_len_cur = 0; r.hpc_len_ext = 0;
			float div = o.div;
			if (P.nucl_alignment)
			{
				// alignment.cpp:218-247 getAlignmentErrEdlib
				extractSeq(c, recIdx, o.curBegin, o.curEnd - o.curBegin, P.use_hpc, S.sa, fromQ);
				extractSeq(c, extRec, o.extBegin, o.extEnd - o.extBegin, P.use_hpc, S.sb);
				int d = editDistance(S.sb, S.sa);
				r.edit_distance = d;
				r.hpc_len_cur = (int32_t)S.sa.size(); r.hpc_len_ext = (int32_t)S.sb.size();
				div = (float)d / std::max(S.sb.size(), S.sa.size());
			}
			r.seq_divergence = div;
			// :470-485; a primary that fails the gate goes to checkIdyAndTrim when
			// _partitionBadMappings is set: that ksw2 step is not restated here, the record is
			// emitted in its place and marked (fo_fetch_trim)
			const bool pass = div < P.max_divergence;
			if (pass || P.partition_bad_mappings)
			{
				out.push_back(r); ++detected;
				if (outTrim) outTrim->push_back(!pass);
				if (outMatches) outMatches->push_back(o.matches);
			}
			size_t w = o.curBegin / STAT_WND;
			if (o.curEnd - o.curBegin > wnd[w].range) { wnd[w].range = o.curEnd - o.curBegin; wnd[w].div = div; }
		}
	}
	for (auto& w : wnd) if (w.range > 0) stats.push_back(w.div);
	counters[7] += tieInQuery;
	if (getenv("FO_STATS7"))
	{
		// queries whose surviving groups hold no tied key: their result would be the same under ANY sort, even one
		// restricted to the hits of the surviving groups
		static std::atomic<unsigned long long> Q[6];	// queries, hits, tie-free queries, their hits; no tie at all: queries, hits
		bool anyTie = false;
		for (size_t t = 1; t < nh && !anyTie; ++t) anyTie = S.hits[t].extId == S.hits[t - 1].extId && S.hits[t].cur == S.hits[t - 1].cur;
		Q[0] += 1; Q[1] += nh; if (!tieInQuery) { Q[2] += 1; Q[3] += nh; } if (!anyTie) { Q[4] += 1; Q[5] += nh; }
		if ((Q[0] & 0x3FF) == 0)
			fprintf(stderr, "queries %llu (%llu hits): no tie inside a group that reaches the DP in %llu (%llu hits); no tie at all in %llu (%llu hits)\n",
					(unsigned long long)Q[0], (unsigned long long)Q[1], (unsigned long long)Q[2], (unsigned long long)Q[3],
					(unsigned long long)Q[4], (unsigned long long)Q[5]);
	}
}

} // namespace

extern "C" {

int fo_overlaps(fo_ctx* h, const fg_detector_params* P, const u32* queryIds, u32 nq, int32_t maxOverlaps,
				uint8_t forceLocal, int threads, u64* nRecs, u64* nStats)
{
	Ctx& c = h->c;
	if (!c.idx.built) return FG_ERR_STATE;
	if (P->partition_bad_mappings && maxOverlaps != 0) return FG_ERR_UNSUPPORTED;
	std::vector<std::vector<fg_overlap_rec>> res(nq);
	std::vector<std::vector<uint8_t>> trim(nq);
	std::vector<std::vector<std::vector<std::pair<int32_t, int32_t>>>> mt(nq);
	std::vector<std::vector<float>> st(nq);
	int T = std::max(1, threads);
	std::vector<Scratch> scratch(T);
	std::vector<std::array<u64, 8>> cnt(T, std::array<u64, 8>{0, 0, 0, 0, 0, 0, 0, 0});
	const u32 qBase = c.hasQ ? c.qFirstId : c.firstId;
	const u32 qCount = c.hasQ ? c.qn : c.n;
	for (u32 i = 0; i < nq; ++i)
		if (queryIds[i] < qBase || queryIds[i] - qBase >= 2 * qCount) return FG_ERR_ARG;
	parallelFor(nq, T, [&](u32 i, int t)
	{
		seqOverlaps(c, *P, queryIds[i] - qBase, forceLocal, maxOverlaps, scratch[t], res[i], st[i], cnt[t].data(),
					P->keep_alignment ? &mt[i] : nullptr, P->partition_bad_mappings ? &trim[i] : nullptr);
	});
	c.outTrim.clear();
	if (P->partition_bad_mappings)
		for (u32 i = 0; i < nq; ++i) c.outTrim.insert(c.outTrim.end(), trim[i].begin(), trim[i].end());
	c.outMatchOff.assign(1, 0); c.outMatches.clear();
	if (P->keep_alignment)
		for (u32 i = 0; i < nq; ++i)
			for (auto& v : mt[i])
			{
				for (auto& pr : v) { c.outMatches.push_back(pr.first); c.outMatches.push_back(pr.second); }
				c.outMatchOff.push_back(c.outMatches.size() / 2);
			}
	c.outOff.assign(nq + 1, 0); c.statOff.assign(nq + 1, 0);
	c.outRecs.clear(); c.outStats.clear();
	c.cntBp = 0;
	for (u32 i = 0; i < nq; ++i)
	{
		c.outOff[i] = c.outRecs.size(); c.statOff[i] = c.outStats.size();
		c.outRecs.insert(c.outRecs.end(), res[i].begin(), res[i].end());
		c.outStats.insert(c.outStats.end(), st[i].begin(), st[i].end());
		c.cntBp += c.hasQ ? c.qLen[(queryIds[i] - qBase) >> 1] : c.len[(queryIds[i] - qBase) >> 1];
	}
	c.outOff[nq] = c.outRecs.size(); c.statOff[nq] = c.outStats.size();
	c.cntKmers = c.cntHits = c.cntGroups = c.cntDp = 0;
	for (auto& a : cnt) { c.cntKmers += a[0]; c.cntHits += a[1]; c.cntGroups += a[2]; c.cntDp += a[3]; }
	if (getenv("FO_STATS"))
	{
		u64 st[4] = {0, 0, 0, 0};
		for (auto& a : cnt) { st[0] += a[4]; st[1] += a[5]; st[2] += a[6]; st[3] += a[7]; }
		fprintf(stderr, "queries with tied (extId, curPos) keys inside a chained group: %llu of %u\n", (unsigned long long)st[3], nq);
		fprintf(stderr, "look-back: %.2f candidates scanned per DP element; %.2f%% of elements scan > 16, %.2f%% > 64\n",
				(double)st[0] / std::max<u64>(1, c.cntDp), 100.0 * st[1] / std::max<u64>(1, c.cntDp), 100.0 * st[2] / std::max<u64>(1, c.cntDp));
	}
	*nRecs = c.outRecs.size(); *nStats = c.outStats.size();
	return 0;
}

int fo_fetch(fo_ctx* h, u64* queryOff, fg_overlap_rec* recs, u64* statOff, float* stats, u64* counters)
{
	Ctx& c = h->c;
	if (queryOff) memcpy(queryOff, c.outOff.data(), c.outOff.size() * 8);
	if (recs) memcpy(recs, c.outRecs.data(), c.outRecs.size() * sizeof(fg_overlap_rec));
	if (statOff) memcpy(statOff, c.statOff.data(), c.statOff.size() * 8);
	if (stats) memcpy(stats, c.outStats.data(), c.outStats.size() * 4);
	if (counters)
	{
		counters[0] = c.cntBp; counters[1] = c.cntKmers; counters[2] = c.cntHits;
		counters[3] = c.cntGroups; counters[4] = c.cntDp;
	}
	return 0;
}

// kmerMatches of the records of the last fo_overlaps call (keep_alignment): matchOff has
// nRecs + 1 entries (in pairs), matches 2 ints (cur, ext) per pair
u64 fo_fetch_matches(fo_ctx* h, u64* matchOff, int32_t* matches)
{
	Ctx& c = h->c;
	if (matchOff) memcpy(matchOff, c.outMatchOff.data(), c.outMatchOff.size() * 8);
	if (matches) memcpy(matches, c.outMatches.data(), c.outMatches.size() * 4);
	return c.outMatches.size() / 2;
}

// needs-trim marks of the records of the last fo_overlaps call (partition_bad_mappings)
u64 fo_fetch_trim(fo_ctx* h, uint8_t* flags)
{
	Ctx& c = h->c;
	if (flags) memcpy(flags, c.outTrim.data(), c.outTrim.size());
	return c.outTrim.size();
}

// exact NW edit distance of two 0..3 strings (kernel-level parity tests)
int fo_edit_distance(const uint8_t* a, int n, const uint8_t* b, int m)
{
	std::vector<uint8_t> va(a, a + n), vb(b, b + m);
	return editDistance(va, vb);
}

// the same through the plain scalar DP / through the bit-vector form started at band k0
int fo_edit_distance_dp(const uint8_t* a, int n, const uint8_t* b, int m)
{
	std::vector<uint8_t> va(a, a + n), vb(b, b + m);
	return editDistanceDP(va, vb);
}
int fo_edit_distance_k0(const uint8_t* a, int n, const uint8_t* b, int m, int k0)
{
	std::vector<uint8_t> va(a, a + n), vb(b, b + m);
	return editDistance(va, vb, k0);
}

// ---- banded affine-gap global alignment with CIGAR -----------------------------------
// getAlignmentCigarKsw (reference src/sequence/alignment.cpp:102-216) = ksw_extz2_sse of minimap2 2.17
// (reference lib/minimap2/ksw2_extz2_sse.c, built with sse2only=1, reference Makefile:21) with match 2,
// mismatch -4, gap open 4, gap extend 2, band 64 doubling while the band cannot hold the length difference,
// no z-drop, global backtrack (lib/minimap2/ksw2.h:116-152), then the CIGAR decoded into runs of '=', 'X',
// 'I', 'D'.  The alignment PATH depends on the formulation's tie rules, on its 8-bit difference arithmetic
// and on what the 16-byte-wide vector loop leaves in cells around the band, so the restatement below keeps
// the same state -- five byte arrays u, v, x, y, s indexed by target position, the target and the reversed
// query behind them in one zeroed buffer (the vector code's loads and stores run past array ends into the
// neighbouring array, which this reproduces) -- and applies the per-cell byte operations of the "gap
// left-alignment" loop, one anti-diagonal at a time.
struct KswOut { bool zdropped = false; std::vector<uint32_t> cigar; };

static inline uint8_t u8(int v) { return (uint8_t)v; }
static inline int8_t s8(uint8_t v) { return (int8_t)v; }

static KswOut kswExtz2Global(const uint8_t* query, int qlen, const uint8_t* target, int tlen, int8_t m, const int8_t* mat,
							 int8_t q, int8_t e, int w)
{
	KswOut out;
	if (m <= 0 || qlen <= 0 || tlen <= 0) return out;
	const int qe = q + e;
	const uint8_t qe2 = u8(qe * 2), maxSc = u8(mat[0] + qe * 2);
	const int8_t scMch = mat[0], scMis = mat[1], scN = mat[m * m - 1] == 0 ? (int8_t)-e : mat[m * m - 1];
	if (w < 0) w = std::max(tlen, qlen);
	const int T16 = (tlen + 15) / 16 * 16, Q16 = (qlen + 15) / 16 * 16;
	int nCol = std::min(qlen, tlen);
	nCol = ((nCol < w + 1 ? nCol : w + 1) + 15) / 16 + 1;
	int minSc = mat[1];
	for (int t = 1; t < m * m; ++t) minSc = std::min<int>(minSc, mat[t]);
	if (-minSc > 2 * qe) return out;
	// one buffer, the vector code's layout: u | v | x | y | s | target | reversed query (+ 16 spare bytes)
	std::vector<uint8_t> mem((size_t)T16 * 6 + Q16 + 32, 0);
	uint8_t* U = mem.data(); uint8_t* V = U + T16; uint8_t* X = V + T16; uint8_t* Y = X + T16; uint8_t* S = Y + T16;
	uint8_t* sf = S + T16; uint8_t* qr = sf + T16;
	const size_t rowBytes = (size_t)nCol * 16;
	std::vector<uint8_t> P((size_t)(qlen + tlen - 1) * rowBytes + 16, 0);
	std::vector<int> off(qlen + tlen - 1), offEnd(qlen + tlen - 1);
	for (int t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	memcpy(sf, target, tlen);
	int lastSt = -1, lastEn = -1;
	std::vector<uint8_t> nu, nv, nx, ny, nd;
	for (int r = 0; r < qlen + tlen - 1; ++r)
	{
		int st = 0, en = tlen - 1;
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < ((r - w + 1) >> 1)) st = (r - w + 1) >> 1;
		if (en > ((r + w) >> 1)) en = (r + w) >> 1;
		if (st > en) { out.zdropped = true; return out; }
		const int st0 = st, en0 = en;
		st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
		uint8_t x1, v1;
		if (st > 0)
		{
			if (st - 1 >= lastSt && st - 1 <= lastEn) { x1 = X[st - 1]; v1 = V[st - 1]; }
			else x1 = v1 = 0;
		}
		else { x1 = 0; v1 = r ? u8(q) : 0; }
		if (en >= r) { Y[r] = 0; U[r] = r ? u8(q) : 0; }
		// scores of the diagonal, in 16-byte pieces from st0 (pieces may run past en0 and past the array)
		const uint8_t* qrr = qr + (qlen - 1 - r);
		for (int t = st0; t <= en0; t += 16)
		{
			uint8_t tmp[16];
			for (int b = 0; b < 16; ++b)
			{
				const uint8_t sq = sf[t + b], sb = qrr[t + b];
				const bool wild = sq == u8(m - 1) || sb == u8(m - 1);
				tmp[b] = u8(wild ? scN : (sq == sb ? scMch : scMis));
			}
			memcpy(S + t, tmp, 16);
		}
		// every cell of [st, en] from the previous diagonal's state
		const int nCell = en - st + 1;
		nu.assign(nCell, 0); nv.assign(nCell, 0); nx.assign(nCell, 0); ny.assign(nCell, 0); nd.assign(nCell, 0);
		for (int t = st; t <= en; ++t)
		{
			const uint8_t xt1 = t == st ? x1 : X[t - 1], vt1 = t == st ? v1 : V[t - 1];
			const uint8_t ut = U[t];
			uint8_t z = u8(S[t] + qe2);
			const uint8_t a = u8(xt1 + vt1), b = u8(Y[t] + ut);
			uint8_t d = s8(a) > s8(z) ? 1 : 0;
			z = s8(z) > 0 ? z : 0;
			z = std::max(z, a);
			if (s8(b) > s8(z)) d = 2;
			z = std::max(z, b);
			z = std::min(z, maxSc);
			nu[t - st] = u8(z - vt1);
			nv[t - st] = u8(z - ut);
			const uint8_t zq = u8(z - q);
			const uint8_t a2 = u8(a - zq), b2 = u8(b - zq);
			if (s8(a2) > 0) { nx[t - st] = a2; d |= 0x08; }
			if (s8(b2) > 0) { ny[t - st] = b2; d |= 0x10; }
			nd[t - st] = d;
		}
		memcpy(U + st, nu.data(), nCell); memcpy(V + st, nv.data(), nCell);
		memcpy(X + st, nx.data(), nCell); memcpy(Y + st, ny.data(), nCell);
		memcpy(P.data() + (size_t)r * rowBytes, nd.data(), std::min<size_t>(nCell, rowBytes));
		off[r] = st; offEnd[r] = en;
		lastSt = st; lastEn = en;
	}
	// backtrack from (tlen - 1, qlen - 1) (ksw2.h:116-152, rotated matrix, no introns)
	std::vector<uint32_t> cig;
	auto push = [&](uint32_t op, int len)
	{
		if (cig.empty() || op != (cig.back() & 0xf)) cig.push_back((uint32_t)len << 4 | op);
		else cig.back() += (uint32_t)len << 4;
	};
	int i = tlen - 1, j = qlen - 1, state = 0;
	while (i >= 0 && j >= 0)
	{
		const int r = i + j;
		int force = -1;
		if (i < off[r]) force = 2;
		if (i > offEnd[r]) force = 1;
		const uint32_t tmp = force < 0 ? P[(size_t)r * rowBytes + i - off[r]] : 0;
		if (state == 0) state = tmp & 7;
		else if (!((tmp >> (state + 2)) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (force >= 0) state = force;
		if (state == 0) { push(0, 1); --i; --j; }
		else if (state == 1 || state == 3) { push(2, 1); --i; }
		else { push(1, 1); --j; }
	}
	if (i >= 0) push(2, i + 1);
	if (j >= 0) push(1, j + 1);
	std::reverse(cig.begin(), cig.end());
	out.cigar.swap(cig);
	return out;
}

struct CigRun { char op; int32_t len; };

// getAlignmentCigarKsw: band doubling, decode; returns the error rate (alignment.cpp:102-216)
static float alignmentCigarKsw(const std::vector<uint8_t>& trg, const std::vector<uint8_t>& qry, std::vector<CigRun>& cigarOut)
{
	const int8_t a = 2, b = -4;
	const int8_t subsMat[] = {a, b, b, b, 0,  b, a, b, b, 0,  b, b, a, b, 0,  b, b, b, a, 0,  0, 0, 0, 0, 0};
	KswOut ez;
	int bandWidth = 64;
	for (;;)
	{
		ez = kswExtz2Global(qry.data(), (int)qry.size(), trg.data(), (int)trg.size(), 5, subsMat, 4, 2, bandWidth);
		if (!ez.zdropped) break;
		if (bandWidth > (int)std::max(qry.size(), trg.size())) break;
		bandWidth *= 2;
	}
	int numMiss = 0, numIndels = 0;
	cigarOut.clear();
	size_t posQry = 0, posTrg = 0;
	for (uint32_t c : ez.cigar)
	{
		const int size = (int)(c >> 4);
		const char op = "MID"[c & 0xf];
		if (op == 'M')
		{
			for (int t = 0; t < size; ++t)
			{
				const char match = trg[posTrg + t] == qry[posQry + t] ? '=' : 'X';
				if (t == 0 || match != cigarOut.back().op) cigarOut.push_back({match, 1});
				else ++cigarOut.back().len;
				numMiss += match == 'X';
			}
			posQry += size; posTrg += size;
		}
		else if (op == 'I') { cigarOut.push_back({'I', size}); posQry += size; numIndels += size; }
		else { cigarOut.push_back({'D', size}); posTrg += size; numIndels += size; }
	}
	return float(numMiss + numIndels) / std::max(trg.size(), qry.size());
}

extern "C" {
// decoded CIGAR of getAlignmentCigarKsw(trg, qry) as (op, len) pairs; returns the number of runs (writes at most cap)
int64_t fo_ksw_cigar(const uint8_t* trg, int tlen, const uint8_t* qry, int qlen, uint8_t* ops, int32_t* lens, int64_t cap, float* errRate)
{
	std::vector<uint8_t> t(trg, trg + tlen), q(qry, qry + qlen);
	std::vector<CigRun> c;
	const float er = alignmentCigarKsw(t, q, c);
	if (errRate) *errRate = er;
	for (size_t i = 0; i < c.size() && (int64_t)i < cap; ++i) { ops[i] = (uint8_t)c[i].op; lens[i] = c[i].len; }
	return (int64_t)c.size();
}
} // extern "C"

// ---- introsort emulation self-test against the real std::sort ---------------
struct KV { u64 key; u32 val; };
struct KVAcc {
	typedef KV T;
	KV* p;
	KV load(int i) const { return p[i]; }
	void store(int i, const KV& v) { p[i] = v; }
	bool less(const KV& a, const KV& b) const { return a.key < b.key; }
};

// sorts (keys, vals) with the emulation and with std::sort; returns the number of
// positions where the two permutations differ (0 = identical)
int64_t fo_introsort_mismatches(const u64* keys, int64_t n)
{
	std::vector<KV> a(n), b(n);
	for (int64_t i = 0; i < n; ++i) { a[i] = {keys[i], (u32)i}; b[i] = a[i]; }
	std::sort(a.begin(), a.end(), [](const KV& x, const KV& y) { return x.key < y.key; });
	KVAcc acc{b.data()};
	int stack[fgsort::STACK_INTS];
	fgsort::sort(acc, 0, (int)n, stack);
	int64_t bad = 0;
	for (int64_t i = 0; i < n; ++i) bad += (a[i].val != b[i].val) || (a[i].key != b[i].key);
	return bad;
}

// std::sort permutation itself (vals out) so GPU kernels can be checked directly
void fo_std_sort_perm(const u64* keys, int64_t n, u32* permOut)
{
	std::vector<KV> a(n);
	for (int64_t i = 0; i < n; ++i) a[i] = {keys[i], (u32)i};
	std::sort(a.begin(), a.end(), [](const KV& x, const KV& y) { return x.key < y.key; });
	for (int64_t i = 0; i < n; ++i) permOut[i] = a[i].val;
}

} // extern "C"
