// Overlap stage on the device: seed collection, exact std::sort-order hit sort,
// per-target chaining DP + backtrack + primary selection.  All kernels here are
// hand-written for gfx950 (64-lane waves); no library primitive is used.
//
// Restates OverlapDetector::getSeqOverlaps (reference src/sequence/overlap.cpp:99-508)
// phase by phase for a whole batch of query reads:
//   k_probe / k_fill    seed collection                 overlap.cpp:176-196,
//                       (+ index lookups)               vertex_index.h:139-246
//   k_sort_level / _wide / _lds   std::sort by (extId, curPos), exact permutation
//                                                       overlap.cpp:201-204
//   k_group_*           equal-extId runs                overlap.cpp:216-234
//   k_chain             prefilter, optional re-sort by extPos, chaining DP,
//                       backtrack, overlapTest, primary selection
//                                                       overlap.cpp:235-458, :29-69
//   host shim           seqDivergence (libm), _maxDivergence gate, maxOverlaps
//                       prefix rule, window statistics  overlap.cpp:417-423, :461-506
//
// Data layout in HBM: hits are a structure of arrays -- key = extId<<32 | curPos
// (so the reference's (extId, curPos) comparator is one integer compare; packed into
// 32 bits when record index and position fit, or with extPos into one 64-bit record
// PK otherwise) and val = extPos -- dense per query in the reference's emission order
// (ascending curPos, per k-mer ascending stored position).
#include "fg_ctx.h"
#include "fg_wavesort.h"
#include "fg_devprim.h"
#include <atomic>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <functional>
#include <thread>
#include <type_traits>

#define WG 256
#define FLAG_SELF (1ULL << 63)
#define FLAG_FLIP (1ULL << 62)
#define OFF_MASK ((1ULL << 38) - 1)

namespace {

// ---- block helpers ------------------------------------------------------------
// exclusive scan of one u32 per thread over the 256-thread block; *total = block sum
__device__ __forceinline__ u32 block_exscan(u32 v, u32* sh /* >= WG/64 + 1 */, u32* total)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	u32 inc = v;
	for (int o = 1; o < 64; o <<= 1)
	{
		u32 t = __shfl_up(inc, o);
		if (lane >= o) inc += t;
	}
	__syncthreads();
	if (lane == 63) sh[w] = inc;
	__syncthreads();
	u32 base = 0, tot = 0;
	for (int i = 0; i < WG / 64; ++i) { u32 s = sh[i]; if (i < w) base += s; tot += s; }
	*total = tot;
	return base + inc - v;
}

// single-block exclusive scan of n u64 (n+1 outputs), n = number of queries
__global__ void k_exscan(const u64* __restrict__ in, u64* __restrict__ out, u32 n)
{
	__shared__ u64 sh[1024 / 64];
	__shared__ u64 carry;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (u32 base = 0; base < n; base += 1024)
	{
		const u32 i = base + threadIdx.x;
		const u64 v = i < n ? in[i] : 0;
		u64 inc = v;
		for (int o = 1; o < 64; o <<= 1)
		{
			u64 t = __shfl_up(inc, o);
			if (lane >= o) inc += t;
		}
		if (lane == 63) sh[w] = inc;
		__syncthreads();
		u64 b = carry;
		for (int j = 0; j < w; ++j) b += sh[j];
		if (i < n) out[i] = b + inc - v;
		__syncthreads();
		if (threadIdx.x == 1023) carry = b + inc;
		__syncthreads();
	}
	if (threadIdx.x == 0) out[n] = carry;
}

// ---- seed collection ---------------------------------------------------------------
// per query k-mer: canonicalise, one probe; remember the slot value (+ flip and
// self-hit flags) so that the fill pass needs no second probe.
// words/wordOff/len: the QUERY container; kmerOff/indexedBits: the indexed container's k-mer
// numbering (indexedBits == nullptr when the queries live in their own container: no
// query can then hit itself)
template <bool WIDE>
__global__ void k_probe(const u32* __restrict__ query, const u64* __restrict__ words,
						const u64* __restrict__ wordOff, const i32* __restrict__ len,
						const u64* __restrict__ kmerOff, const u64* __restrict__ qKmerOff, int k,
						FgTable table,
						const u32* __restrict__ indexedBits, u64* __restrict__ probe,
						u64* __restrict__ hitCnt, u64* __restrict__ filtCnt)
{
	__shared__ u32 shA[WG / 64], shB[WG / 64];
	const u32 q = blockIdx.x;
	const u32 rec = query[q];
	const u32 r = rec >> 1;
	const bool rc = rec & 1;
	const i32 L = len[r];
	const i32 nk = L - k;
	const u64* w = words + wordOff[r];
	u64* pr = probe + qKmerOff[q];
	const u64 kbase = indexedBits ? kmerOff[r] : 0;
	u32 hits = 0, filt = 0;
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		const i32 qf = rc ? nk - p : p;	// forward-strand position of the same k-mer (L-k-p)
		u64 a, b;
		fg_kmer_pair(w, qf, k, a, b);
		const u64 fw = rc ? b : a, rv = rc ? a : b;
		const bool flip = rv < fw;
		// the "this position owns an entry" bit is fetched alongside the probe (its address only
		// depends on the position), not behind it
		// forward position nk (= L-k) is never a forward k-mer position (kmer.h:193-198)
		const u64 bit = kbase + (u64)qf;
		const u32 selfWord = (indexedBits && qf < nk) ? indexedBits[bit >> 5] : 0u;
		u64 v = fg_probe<WIDE>(table, flip ? rv : fw);
		if (v != 0)
		{
			const u32 cnt = (u32)(v & FG_CNT_MASK);
			if (cnt == FG_CNT_REPETITIVE) ++filt;
			else
			{
				const u32 self = (selfWord >> (bit & 31)) & 1u;
				hits += cnt - self;
				if (self) v |= FLAG_SELF;
				if (flip) v |= FLAG_FLIP;
			}
		}
		pr[p] = v;
	}
	// block sums
	for (int o = 32; o > 0; o >>= 1) { hits += __shfl_down(hits, o); filt += __shfl_down(filt, o); }
	if ((threadIdx.x & 63) == 0) { shA[threadIdx.x >> 6] = hits; shB[threadIdx.x >> 6] = filt; }
	__syncthreads();
	if (threadIdx.x == 0)
	{
		u64 h = 0, f = 0;
		for (int i = 0; i < WG / 64; ++i) { h += shA[i]; f += shB[i]; }
		hitCnt[q] = h; filtCnt[q] = f;
	}
}

// ---- seed collection against a lookup table beyond the caches: probes partitioned by table region ----------------
// A table of tens of GB (D. melanogaster: 18 GB, 10 Gbp of reads: 46 GB) serves random 64-byte probes at a tenth of
// the rate of one that sits in the Infinity Cache (4 against 40 G probes/s: every probe is a TLB miss and a DRAM
// page of its own).  So the probes of a batch of queries are first ordered by the table region they fall into:
//   k_probe_emit    per query k-mer: (region << 34 | canonical k-mer, position | flags)
//   one pass of the onesweep radix sort (fg_devprim.h) on the region bits -- a stable partition into 256 regions
//   k_probe_sorted  probes region by region (a region = 1/256 of the table: inside the Infinity Cache and the TLB
//                   reach while it is worked on); a hit scatters its value to the position's slot of `probe`
//                   (misses -- four in five -- write nothing: the array starts zeroed)
//   k_probe_count   the per-query totals k_probe gives, from the array
// Narrow tables only (k <= 17: the k-mer leaves room for the region in one 64-bit key).
#define PROBE_REGION_BITS 8
#define PART_KEY_BITS 34
__global__ void k_probe_emit(u32 q0, const u32* __restrict__ query, const u64* __restrict__ words,
							 const u64* __restrict__ wordOff, const i32* __restrict__ len,
							 const u64* __restrict__ kmerOff, const u64* __restrict__ qKmerOff, int k, FgTable table,
							 int regionShift, const u32* __restrict__ indexedBits, u64* __restrict__ keys, u64* __restrict__ vals)
{
	const u32 q = q0 + blockIdx.x;
	const u32 rec = query[q];
	const u32 r = rec >> 1;
	const bool rc = rec & 1;
	const i32 nk = len[r] - k;
	const u64* w = words + wordOff[r];
	const u64 out0 = qKmerOff[q] - qKmerOff[q0];
	const u64 kbase = indexedBits ? kmerOff[r] : 0;
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		const i32 qf = rc ? nk - p : p;
		u64 a, b;
		fg_kmer_pair(w, qf, k, a, b);
		const u64 fw = rc ? b : a, rv = rc ? a : b;
		const bool flip = rv < fw;
		const u64 key = flip ? rv : fw;
		const u64 bit = kbase + (u64)qf;
		const u32 self = (indexedBits && qf < nk) ? ((indexedBits[bit >> 5] >> (bit & 31)) & 1u) : 0u;
		u32 part = 0;
		if (table.nParts > 1)
			while (part + 1 < table.nParts && key >= table.bound[part + 1]) ++part;
		const u32 g = __umulhi((u32)(fg_mix(key) >> 32), table.groups[part]);
		const u64 region = (table.slotBase[part] + (u64)g * 8u) >> regionShift;
		keys[out0 + p] = (region << PART_KEY_BITS) | key;
		vals[out0 + p] = (qKmerOff[q] + (u64)p) | (self ? FLAG_SELF : 0ULL) | (flip ? FLAG_FLIP : 0ULL);
	}
}

__global__ void k_probe_sorted(const u64* __restrict__ keys, const u64* __restrict__ vals, u64 n, FgTable table,
							   u64* __restrict__ probe, int ablate)
{
	const u64 i = (u64)blockIdx.x * WG + threadIdx.x;
	if (i >= n) return;
	u64 v;
	if (ablate & 2)
	{
		// timing experiment (results become wrong): the table lines only, no list bounds
		const u64 key = keys[i] & ((1ULL << PART_KEY_BITS) - 1);
		u32 part = 0;
		if (table.nParts > 1)
			while (part + 1 < table.nParts && key >= table.bound[part + 1]) ++part;
		const u64 hit = fg_probe_slot(table, part, __umulhi((u32)(fg_mix(key) >> 32), table.groups[part]), key);
		v = hit == FG_EMPTY_KEY ? 0 : ((hit & FG_IDX_MASK) << FG_CNT_BITS | 1);
	}
	else v = fg_probe<false>(table, keys[i] & ((1ULL << PART_KEY_BITS) - 1));
	if (v == 0) return;
	if (ablate & 1) { if (v == 12345) probe[0] = v; return; }		// timing experiment: no scatter
	const u64 val = vals[i];
	if ((u32)(v & FG_CNT_MASK) != FG_CNT_REPETITIVE) v |= val & (FLAG_SELF | FLAG_FLIP);
	probe[val & ~(FLAG_SELF | FLAG_FLIP)] = v;
}

__global__ void k_probe_count(const u64* __restrict__ qKmerOff, const u64* __restrict__ probe,
							  u64* __restrict__ hitCnt, u64* __restrict__ filtCnt)
{
	__shared__ u32 shA[WG / 64], shB[WG / 64];
	const u32 q = blockIdx.x;
	const u64 a = qKmerOff[q], b = qKmerOff[q + 1];
	u32 hits = 0, filt = 0;
	for (u64 i = a + threadIdx.x; i < b; i += WG)
	{
		const u64 v = probe[i];
		if (v == 0) continue;
		const u32 cnt = (u32)(v & FG_CNT_MASK);
		if (cnt == FG_CNT_REPETITIVE) ++filt;
		else hits += cnt - ((v & FLAG_SELF) ? 1u : 0u);
	}
	for (int o = 32; o > 0; o >>= 1) { hits += __shfl_down(hits, o); filt += __shfl_down(filt, o); }
	if ((threadIdx.x & 63) == 0) { shA[threadIdx.x >> 6] = hits; shB[threadIdx.x >> 6] = filt; }
	__syncthreads();
	if (threadIdx.x == 0)
	{
		u64 h = 0, f = 0;
		for (int i = 0; i < WG / 64; ++i) { h += shA[i]; f += shB[i]; }
		hitCnt[q] = h; filtCnt[q] = f;
	}
}

// expand every hit list in query order (overlap.cpp:176-196); a stored entry
// (record, pos) is reported in the query k-mer's orientation (vertex_index.h:158-174).
// Per 256 query positions: owners publish (output start, list offset, count, flags) in
// LDS, then the block walks the OUTPUT slots -- thread o finds its owner by bisection of the
// starts -- so the 12-byte hit records leave as coalesced stores.
// KT = u64: key = extId << 32 | curPos.  KT = u32 (when record index and position fit 32 bits
// together): key = record << curBits | curPos -- same order, a third less sort traffic.
// KT = PK: one 64-bit record per hit (fg_ctx.h), no value array.  Consumers read any of the
// three through HitKeyView<KT>.
#ifndef FILL_ITEMS
#define FILL_ITEMS 4
#endif
template <class KT>
__global__ void k_fill(const u32* __restrict__ query, const i32* __restrict__ len, const i32* __restrict__ qLen,
					   const u64* __restrict__ qKmerOff, int k, u32 firstId, int curBits,
					   const u64* __restrict__ probe, const u64* __restrict__ entries,
					   const u64* __restrict__ hitOff, const u64* __restrict__ filtOff,
					   KT* __restrict__ hitKey, u32* __restrict__ hitVal, i32* __restrict__ filtPos)
{
	// FILL_ITEMS consecutive positions per thread and step: a step is a chain of barriers, LDS searches and
	// dependent loads (~4 us whatever it carries), and a read of 10^4 positions took 40 of them at one position per
	// thread
	constexpr int FI = FILL_ITEMS;
	__shared__ u32 sh[WG / 64 + 1];
	__shared__ u32 sStart[WG * FI + 1];
	__shared__ u64 sOff[WG * FI];		// list offset | FLAG_FLIP | FLAG_SELF (the list holds this position's own, trivial entry)
	const u32 q = blockIdx.x;
	const u32 rec = query[q];
	const i32 L = qLen[rec >> 1];
	const i32 nk = L - k;
	const u64* pr = probe + qKmerOff[q];
	u64 hbase = hitOff[q];
	u64 fbase = filtOff[q];
	for (i32 p0 = 0; p0 < nk; p0 += WG * FI)
	{
		u64 vv[FI];
		u32 eff[FI];
		u32 effSum = 0, repSum = 0;
#pragma unroll
		for (int i = 0; i < FI; ++i)
		{
			const i32 p = p0 + (i32)threadIdx.x * FI + i;
			const u64 v = p < nk ? pr[p] : 0;
			u32 cnt = (u32)(v & FG_CNT_MASK);
			const bool rep = (cnt == FG_CNT_REPETITIVE);
			if (rep) cnt = 0;
			// the trivial self hit is dropped in the output walk below (no search here: a per-position
			// binary search of the list put log2(cnt) dependent loads on every step of the block)
			const bool hasSelf = cnt && (v & FLAG_SELF);
			eff[i] = cnt - (hasSelf ? 1u : 0u);
			vv[i] = (((v >> FG_CNT_BITS) & OFF_MASK) | (v & FLAG_FLIP) | (hasSelf ? FLAG_SELF : 0ULL));
			if (rep) { vv[i] = ~0ULL; ++repSum; }		// marks a repetitive position (no list)
			effSum += eff[i];
		}
		u32 tot, ftot;
		u32 start = block_exscan(effSum, sh, &tot);
		u32 fstart = block_exscan(repSum, sh, &ftot);
#pragma unroll
		for (int i = 0; i < FI; ++i)
		{
			const i32 p = p0 + (i32)threadIdx.x * FI + i;
			const bool rep = vv[i] == ~0ULL;
			if (rep) filtPos[fbase + fstart++] = p;
			sStart[threadIdx.x * FI + i] = start;
			sOff[threadIdx.x * FI + i] = rep ? 0ULL : vv[i];
			start += eff[i];
		}
		if (threadIdx.x == 0) sStart[WG * FI] = tot;
		__syncthreads();
		for (u32 o = threadIdx.x; o < tot; o += WG)
		{
			// owner = last t with sStart[t] <= o (it has a non-empty list)
			u32 lo = 0, hi = WG * FI;
			while (hi - lo > 1) { const u32 m = (lo + hi) >> 1; if (sStart[m] <= o) lo = m; else hi = m; }
			const u32 t = lo;
			const u32 j = o - sStart[t];
			const u64 so = sOff[t];
			u64 e = entries[(so & OFF_MASK) + j];
			if (so & FLAG_SELF)
			{
				// no trivial matches (overlap.cpp:188-190): the list is ascending and holds this
				// position's own entry exactly once; output j is entry j before it, entry j + 1 after
				const i32 pq = p0 + (i32)t;
				const bool fl = so & FLAG_FLIP;
				const u64 own = ((u64)(rec ^ (fl ? 1u : 0u)) << 32) | (u32)(fl ? L - pq - k : pq);
				if (e >= own) e = entries[(so & OFF_MASK) + j + 1];
			}
			u32 srec = (u32)(e >> 32);
			i32 spos = (i32)(u32)e;
			if (so & FLAG_FLIP) { spos = len[srec >> 1] - spos - k; srec ^= 1u; }
			if constexpr (std::is_same<KT, PK>::value)
				hitKey[hbase + o] = PK(((u64)srec << (curBits + FG_PK_VALBITS)) | ((u64)(u32)(p0 + (i32)t) << FG_PK_VALBITS) |
									   (u64)(u32)spos);
			else
			{
				if (sizeof(KT) == 8) hitKey[hbase + o] = (KT)(((u64)(firstId + srec) << 32) | (u32)(p0 + (i32)t));
				else hitKey[hbase + o] = (KT)((srec << curBits) | (u32)(p0 + (i32)t));
				hitVal[hbase + o] = (u32)spos;
			}
		}
		__syncthreads();
		hbase += tot;
		fbase += ftot;
	}
}

// ---- exact std::sort order of every query's hits (fg_wavesort.h) ---------------------
// The introsort recursion is run level by level over ALL queries of the chunk: every piece
// larger than SORT_CAP is one task of k_sort_level (one wave = one Hoare partition in global
// memory, L2-resident), its two halves become tasks of the next level, pieces that fit
// SORT_CAP elements are queued for k_sort_lds (one wave per piece, entirely in LDS).  A read
// with 10^5..10^6 hits thus spreads over more waves at every level instead of serialising
// ~10 levels on one wave.
#ifndef SORT_CAP
#define SORT_CAP 448	// levels + LDS kernel at 256 / 320 / 384 / 448 / 512 / 1024: 20.5 / 19.1 / 19.5 / 18.8 / 20.1 / 21.4 ms
#endif
#ifndef SORT_LDS_WAVES
#define SORT_LDS_WAVES 1	// pieces per block: one (4 -> 1: 8.86 -> 7.89 ms; uneven pieces hold a shared block)
#endif
#ifndef SORT_LEVEL_WAVES
#define SORT_LEVEL_WAVES 1	// tasks per block of k_sort_level
#endif
struct SortTask { u64 start; u32 n; u32 depth; };

// the value array that goes with a key type: a real u32 array, or nothing for packed records
template <class KT> struct ValPtr {
	typedef u32* type;
	static __device__ __forceinline__ type at(u32* v, u64 off) { return v + off; }
};
template <> struct ValPtr<PK> {
	typedef NoVal type;
	static __device__ __forceinline__ type at(u32*, u64) { return NoVal{}; }
};

// block-aggregated append (one atomic per list and block).  Lists: big (one wave per task in
// k_sort_level), wide (> wideMin elements: a whole workgroup per task, k_sort_wide), small
// (<= SORT_CAP: k_sort_lds).  counts: [0] big, [2] wide of the next level; smallCnt: small pieces of all levels.
struct SortLists {
	SortTask* small;
	u32 wideMin, smallCap;
	// A level with >= manyMin pieces has enough one-wave tasks in flight to hide the streamed partition's serial chain,
	// and that form moves 12 B per hit against the closed form's ~30: there pieces of up to streamMany hits stay
	// one-wave tasks and are streamed; a level with few pieces uses streamMax / wideMin (closed form, whole workgroups)
	u32 manyMin, streamMany, streamMax;
};
// the level's thresholds: what k_sort_level streams (row word [1]) and what goes to whole workgroups
__device__ __forceinline__ u32 sort_level_mode(const SortLists& L, u32 nPieces, u32* __restrict__ row, bool writer)
{
	const bool many = L.manyMin && nPieces >= L.manyMin;
	if (writer) row[1] = many ? L.streamMany : L.streamMax;
	return many ? (L.streamMany > L.wideMin ? L.streamMany : L.wideMin) : L.wideMin;
}
__device__ __forceinline__ void sort_route(const SortTask& t, bool valid, const SortLists& L, u32 wideMin, SortTask* __restrict__ big,
										   SortTask* __restrict__ wide, u32* __restrict__ counts /* [0] big, [2] wide */,
										   u32* __restrict__ smallCnt)
{
	__shared__ u32 wcnt[3][WG / 64];
	__shared__ u32 base[3];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const bool isWide = valid && t.n > wideMin && t.n > SORT_CAP;
	const bool isBig = valid && t.n > SORT_CAP && !isWide;
	const bool isSmall = valid && t.n >= 2 && t.n <= SORT_CAP;
	const u64 mB = __ballot(isBig), mS = __ballot(isSmall), mW = __ballot(isWide);
	if (lane == 0) { wcnt[0][wv] = (u32)__popcll(mB); wcnt[1][wv] = (u32)__popcll(mS); wcnt[2][wv] = (u32)__popcll(mW); }
	__syncthreads();
	if (threadIdx.x < 3)
	{
		u32 tot = 0;
		for (int i = 0; i < WG / 64; ++i) tot += wcnt[threadIdx.x][i];
		base[threadIdx.x] = tot ? atomicAdd(threadIdx.x == 1 ? smallCnt : &counts[threadIdx.x], tot) : 0u;
	}
	__syncthreads();
	const u64 below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
	u32 oB = base[0], oS = base[1], oW = base[2];
	for (int i = 0; i < wv; ++i) { oB += wcnt[0][i]; oS += wcnt[1][i]; oW += wcnt[2][i]; }
	if (isBig) big[oB + __popcll(mB & below)] = t;
	if (isWide) wide[oW + __popcll(mW & below)] = t;
	if (isSmall)
	{
		const u32 slot = oS + __popcll(mS & below);
		if (slot < L.smallCap) L.small[slot] = t;	// the count keeps the true total; the host checks it
	}
	__syncthreads();
}

// Task counts of the level loop live on the device, one row of four u32 per level: [0] one-wave tasks, [1] unused,
// [2] workgroup (wide) tasks of that level; the number of small pieces queued for k_sort_lds (all levels together) sits
// in its own word.  The kernels of level l read row l and route the children into row l + 1, looping over the tasks
// with the grid they were given -- so the host can launch several levels in a row with grids sized by an upper bound
// and read the real counts back only now and then (a round trip per level was 80 us of an otherwise idle chip for
// each of the ~12 tail levels that hold a handful of pieces).
__global__ void k_sort_init(const u64* __restrict__ hitOff, u32 nq, SortLists L, SortTask* __restrict__ big,
							SortTask* __restrict__ wide, u32* __restrict__ levelCnt, u32* __restrict__ smallCnt)
{
	const u32 q = blockIdx.x * WG + threadIdx.x;
	SortTask t{0, 0, 0};
	if (q < nq)
	{
		const u64 n = hitOff[q + 1] - hitOff[q];
		t.start = hitOff[q]; t.n = (u32)n; t.depth = n >= 2 ? 2 * fgsort::floor_log2_((int)n) : 0;
	}
	const u32 wideMin = sort_level_mode(L, nq, levelCnt, blockIdx.x == 0 && threadIdx.x == 0);
	sort_route(t, q < nq, L, wideMin, big, wide, levelCnt, smallCnt);
}

// a workgroup of SORT_WIDE_WAVES waves per task: one partition of a huge piece
#ifndef SORT_WIDE_WAVES
#define SORT_WIDE_WAVES 8
#endif
template <class KT>
__global__ void __launch_bounds__(SORT_WIDE_WAVES * 64)
k_sort_wide(const SortTask* __restrict__ tasks, const u32* __restrict__ levelCnt, KT* __restrict__ hitKey,
			u32* __restrict__ hitVal, u32* __restrict__ posScratch, u64 nHits,
			SortTask* __restrict__ kids)
{
	__shared__ int shm[2 * SORT_WIDE_WAVES];
	const u32 nTasks = fg_uni(levelCnt[2]);
	SortTask* children = kids + 2 * (size_t)fg_uni(levelCnt[0]);	// behind the one-wave tasks' children
	for (u32 ti = blockIdx.x; ti < nTasks; ti += gridDim.x)
	{
		SortTask t = tasks[ti];
		t.start = fg_uni(t.start); t.n = fg_uni(t.n); t.depth = fg_uni(t.depth);
		KT* K = hitKey + t.start;
		typename ValPtr<KT>::type V = ValPtr<KT>::at(hitVal, t.start);
		SortTask c0{0, 0, 0}, c1{0, 0, 0};
		if (t.depth == 0)
		{
			if (threadIdx.x == 0) { wsort::PtrAcc<KT, typename ValPtr<KT>::type> acc{K, V}; fgsort::heap_sort_(acc, 0, (int)t.n); }
		}
		else
		{
			const int cut = wsort::partition_cf_block<KT, SORT_WIDE_WAVES>(K, V, (int)t.n, posScratch + t.start,
																		   posScratch + nHits + t.start, shm);
			c0 = SortTask{t.start, (u32)cut, t.depth - 1};
			c1 = SortTask{t.start + (u64)cut, t.n - (u32)cut, t.depth - 1};
		}
		if (threadIdx.x == 0) { children[2 * (u64)ti] = c0; children[2 * (u64)ti + 1] = c1; }
		__syncthreads();		// shm is reused by the block's next task
	}
}

// one wave per task: one partition (or the depth-limit heapsort); children to slots 2i, 2i+1
template <class KT>
__global__ void k_sort_level(const SortTask* __restrict__ tasks, const u32* __restrict__ levelCnt, KT* __restrict__ hitKey,
							 u32* __restrict__ hitVal, u32* __restrict__ posScratch, u64 nHits,
							 SortTask* __restrict__ children)
{
	const int lane = threadIdx.x & 63;
	const u32 nTasks = fg_uni(levelCnt[0]);
	const u32 streamMax = fg_uni(levelCnt[1]);		// set by the kernel that routed this level's pieces (sort_level_mode)
	for (u32 ti = blockIdx.x * SORT_LEVEL_WAVES + (threadIdx.x >> 6); ti < nTasks; ti += gridDim.x * SORT_LEVEL_WAVES)
	{
		SortTask t = tasks[ti];
		t.start = fg_uni(t.start); t.n = fg_uni(t.n); t.depth = fg_uni(t.depth);
		KT* K = hitKey + t.start;
		typename ValPtr<KT>::type V = ValPtr<KT>::at(hitVal, t.start);
		SortTask c0{0, 0, 0}, c1{0, 0, 0};
		if (t.depth == 0)
		{
			if (lane == 0) { wsort::PtrAcc<KT, typename ValPtr<KT>::type> acc{K, V}; fgsort::heap_sort_(acc, 0, (int)t.n); }
		}
		else
		{
			// many medium pieces in flight: the streamed form moves fewer bytes; few huge pieces:
			// the closed form has no serial chain
			const int cut = t.n <= streamMax
				? wsort::partition_stream(K, V, 0, (int)t.n)
				: wsort::partition_cf(K, V, 0, (int)t.n, posScratch + t.start, posScratch + nHits + t.start);
			c0 = SortTask{t.start, (u32)cut, t.depth - 1};
			c1 = SortTask{t.start + (u64)cut, t.n - (u32)cut, t.depth - 1};
		}
		if (lane == 0) { children[2 * (u64)ti] = c0; children[2 * (u64)ti + 1] = c1; }
	}
}

__global__ void k_sort_route(const SortTask* __restrict__ children, const u32* __restrict__ levelCnt, SortLists L,
							 SortTask* __restrict__ big, SortTask* __restrict__ wide, u32* __restrict__ nextCnt,
							 u32* __restrict__ smallCnt)
{
	const u32 nChildren = 2 * (levelCnt[0] + levelCnt[2]);
	const u32 wideMin = sort_level_mode(L, nChildren, nextCnt, blockIdx.x == 0 && threadIdx.x == 0);
	for (u32 base = blockIdx.x * WG; base < nChildren; base += gridDim.x * WG)		// uniform per block: sort_route has barriers
	{
		const u32 i = base + threadIdx.x;
		SortTask t{0, 0, 0};
		if (i < nChildren) t = children[i];
		sort_route(t, i < nChildren, L, wideMin, big, wide, nextCnt, smallCnt);
	}
}

template <class KT>
__global__ void __launch_bounds__(SORT_LDS_WAVES * 64)
k_sort_lds(const SortTask* __restrict__ tasks, u32 nTasks,
		   KT* __restrict__ hitKey, u32* __restrict__ hitVal, int curBits, u64 narrowMax,
		   u32* __restrict__ posScratch, u64 nHits)
{
	// the LDS piece always holds 32-bit keys (12 B per hit with the position scratch)
	__shared__ u32 sK[SORT_LDS_WAVES][SORT_CAP];
	__shared__ u32 sV[SORT_LDS_WAVES][SORT_CAP];
	__shared__ unsigned short sPL[SORT_LDS_WAVES][SORT_CAP], sPR[SORT_LDS_WAVES][SORT_CAP];
	__shared__ int stack[SORT_LDS_WAVES][3 * 40];
	__shared__ int small[SORT_LDS_WAVES][3 * 8];
	const int wv = threadIdx.x >> 6;
	const int lane = threadIdx.x & 63;
	for (u32 ti = blockIdx.x * SORT_LDS_WAVES + wv; ti < nTasks; ti += gridDim.x * SORT_LDS_WAVES)
	{
	SortTask t = tasks[ti];
	t.start = fg_uni(t.start); t.n = fg_uni(t.n); t.depth = fg_uni(t.depth);
	KT* K = hitKey + t.start;
	u32* V = hitVal + t.start;
	const int n = (int)t.n;
	if constexpr (std::is_same<KT, PK>::value)
	{
		// packed records: key = v >> 24 ((record, curPos), <= 40 bits), value = the low 24 bits.
		// Narrowed to 32 bits relative to the piece's minimum key when the piece spans little enough.
		const u64 vmask = (1ULL << FG_PK_VALBITS) - 1;
		u64 pk[(SORT_CAP + 63) / 64];
		u64 mn = ~0ULL, mx = 0;
#pragma unroll
		for (int j = 0; j < (SORT_CAP + 63) / 64; ++j)
		{
			const int i = j * 64 + lane;
			const u64 rec = i < n ? K[i].v : 0;
			pk[j] = rec;
			const u64 key = rec >> FG_PK_VALBITS;
			if (i < n) { mn = key < mn ? key : mn; mx = key > mx ? key : mx; }
		}
		for (int o = 32; o > 0; o >>= 1)
		{
			const u64 a = wsort::shflk(mn, lane ^ o), b = wsort::shflk(mx, lane ^ o);
			mn = a < mn ? a : mn; mx = b > mx ? b : mx;
		}
		mn = fg_uni(mn); mx = fg_uni(mx);
		if (mx - mn <= narrowMax)
		{
#pragma unroll
			for (int j = 0; j < (SORT_CAP + 63) / 64; ++j)
			{
				const int i = j * 64 + lane;
				if (i < n) { sK[wv][i] = (u32)((pk[j] >> FG_PK_VALBITS) - mn); sV[wv][i] = (u32)(pk[j] & vmask); }
			}
			wsort::wave_mem_fence();
			wsort::wave_sort<u32, unsigned short>(sK[wv], sV[wv], n, sPL[wv], sPR[wv], stack[wv], small[wv], 0, (int)t.depth);
			for (int i = lane; i < n; i += 64) K[i] = PK((((u64)sK[wv][i] + mn) << FG_PK_VALBITS) | (u64)sV[wv][i]);
		}
		else
		{
			// a piece spanning more than 2^32 keys (rare): keys widened in place in global memory, the
			// values parked in the position scratch, position lists in LDS
			u64* K64 = (u64*)K;
			u32* vals = posScratch + t.start;
#pragma unroll
			for (int j = 0; j < (SORT_CAP + 63) / 64; ++j)
			{
				const int i = j * 64 + lane;
				if (i < n) { K64[i] = pk[j] >> FG_PK_VALBITS; vals[i] = (u32)(pk[j] & vmask); }
			}
			wsort::wave_mem_fence();
			wsort::wave_sort<u64, unsigned short>(K64, vals, n, sPL[wv], sPR[wv], stack[wv], small[wv], 0, (int)t.depth);
			for (int i = lane; i < n; i += 64)
			{
				const u64 key = K64[i];
				const u32 val = vals[i];
				K[i] = PK((key << FG_PK_VALBITS) | (u64)val);
			}
		}
		continue;
	}
	else if constexpr (sizeof(KT) == 4)
	{
		for (int i = lane; i < n; i += 64) { sK[wv][i] = (u32)K[i]; sV[wv][i] = V[i]; }
		wsort::wave_mem_fence();
		wsort::wave_sort<u32, unsigned short>(sK[wv], sV[wv], n, sPL[wv], sPR[wv], stack[wv], small[wv], 0, (int)t.depth);
		for (int i = lane; i < n; i += 64) { K[i] = (KT)sK[wv][i]; V[i] = sV[wv][i]; }
		continue;
	}
	else
	{
	// 64-bit keys (extId << 32 | curPos): a piece of <= SORT_CAP hits of one query usually spans few
	// target records, so (extId << curBits | curPos) minus the piece's minimum fits 32 bits -- an
	// order-preserving map, hence the same permutation at the 32-bit kernel's cost.  curBits = 0
	// (arbitrary keys, fg_debug_sort_pairs): the keys themselves are tried.
	const u64 lowMask = curBits ? (1ULL << curBits) - 1 : 0;
	u64 pk[SORT_CAP / 64];
	u64 mn = ~0ULL, mx = 0;
#pragma unroll
	for (int j = 0; j < SORT_CAP / 64; ++j)
	{
		const int i = j * 64 + lane;
		u64 k64 = i < n ? (u64)K[i] : 0;
		if (curBits) k64 = ((k64 >> 32) << curBits) | (k64 & 0xFFFFFFFFULL);
		pk[j] = k64;
		if (i < n) { mn = k64 < mn ? k64 : mn; mx = k64 > mx ? k64 : mx; }
	}
	for (int o = 32; o > 0; o >>= 1)
	{
		const u64 a = wsort::shflk(mn, lane ^ o), b = wsort::shflk(mx, lane ^ o);
		mn = a < mn ? a : mn; mx = b > mx ? b : mx;
	}
	mn = fg_uni(mn); mx = fg_uni(mx);
	if (mx - mn <= narrowMax)
	{
#pragma unroll
		for (int j = 0; j < SORT_CAP / 64; ++j)
		{
			const int i = j * 64 + lane;
			if (i < n) { sK[wv][i] = (u32)(pk[j] - mn); sV[wv][i] = V[i]; }
		}
		wsort::wave_mem_fence();
		wsort::wave_sort<u32, unsigned short>(sK[wv], sV[wv], n, sPL[wv], sPR[wv], stack[wv], small[wv], 0, (int)t.depth);
		for (int i = lane; i < n; i += 64)
		{
			const u64 p = (u64)sK[wv][i] + mn;
			K[i] = (KT)(curBits ? (((p >> curBits) << 32) | (p & lowMask)) : p);
			V[i] = sV[wv][i];
		}
		continue;
	}
	// a piece that spans more than 2^32 packed keys: sorted where it lies, in global memory
	wsort::wave_sort<KT, u32>(K, V, n, posScratch + t.start, posScratch + nHits + t.start, stack[wv], small[wv], 0,
							  (int)t.depth);
	}
	}
}

// ---- target groups ---------------------------------------------------------------------
template <class KT>
__global__ void k_group_count(const u64* __restrict__ hitOff, HitKeyView<KT> hitKey, u64* __restrict__ groupCnt)
{
	__shared__ u32 sh[WG / 64];
	const u32 q = blockIdx.x;
	const u64 b = hitOff[q], e = hitOff[q + 1];
	u32 c = 0;
	for (u64 i = b + threadIdx.x; i < e; i += WG)
		c += (i == b) || (hitKey.ext_raw(i) != hitKey.ext_raw(i - 1));
	for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) { u64 t = 0; for (int i = 0; i < WG / 64; ++i) t += sh[i]; groupCnt[q] = t; }
}

// group boundaries + what later kernels need of the keys: target id, first / last query position
template <class KT>
__global__ void k_group_fill(const u64* __restrict__ hitOff, HitKeyView<KT> hitKey,
							 const u64* __restrict__ groupOff, u64* __restrict__ groupStart,
							 u32* __restrict__ groupQuery, u32* __restrict__ groupExt,
							 u32* __restrict__ groupFirstCur, u32* __restrict__ groupLastCur)
{
	__shared__ u32 sh[WG / 64 + 1];
	const u32 q = blockIdx.x;
	const u64 b = hitOff[q], e = hitOff[q + 1];
	u64 gbase = groupOff[q];
	// GF consecutive hits per thread and step: one block scan (two barriers) per GF * 256 hits
	constexpr int GF = 4;
	for (u64 i0 = b; i0 < e; i0 += (u64)WG * GF)
	{
		const u64 first = i0 + (u64)threadIdx.x * GF;
		u32 raw[GF + 1];
		raw[0] = (first > b && first - 1 < e) ? hitKey.ext_raw(first - 1) : 0u;
#pragma unroll
		for (int t = 0; t < GF; ++t) raw[t + 1] = first + t < e ? hitKey.ext_raw(first + t) : 0u;
		u32 nHead = 0;
		bool head[GF];
#pragma unroll
		for (int t = 0; t < GF; ++t)
		{
			const u64 i = first + t;
			head[t] = i < e && ((i == b) || (raw[t + 1] != raw[t]));
			nHead += head[t];
		}
		u32 tot;
		u32 pos = block_exscan(nHead, sh, &tot);
#pragma unroll
		for (int t = 0; t < GF; ++t)
			if (head[t])
			{
				const u64 i = first + t;
				const u64 g = gbase + pos;
				groupStart[g] = i; groupQuery[g] = q;
				groupExt[g] = hitKey.ext(i); groupFirstCur[g] = hitKey.cur(i);
				if (i > b) groupLastCur[g - 1] = hitKey.cur(i - 1);	// closes the previous group of this query
				++pos;
			}
		gbase += tot;
	}
	if (threadIdx.x == 0 && e > b) groupLastCur[gbase - 1] = hitKey.cur(e - 1);
}

__global__ void k_prim_count(const u64* __restrict__ groupOff, const u32* __restrict__ primFlag,
							 const u32* __restrict__ dpSize, u64* __restrict__ primCnt,
							 u64* __restrict__ dpGroups, u64* __restrict__ dpElems, const u64* __restrict__ groupStart,
							 u64 nGroups, u64 nHits, u32 smallMax, unsigned long long* __restrict__ smallElems)
{
	__shared__ u32 sh[4][WG / 64];
	const u32 q = blockIdx.x;
	const u64 b = groupOff[q], e = groupOff[q + 1];
	u32 c = 0, dg = 0, de = 0, ds = 0;
	for (u64 i = b + threadIdx.x; i < e; i += WG)
	{
		c += primFlag[i];
		const u32 d = dpSize[i];
		dg += d != 0; de += d;
		if (d && smallMax)		// the class k_group_list routes to k_chain_small: by the group's hits before the prefilter
		{
			const u64 n = (i + 1 < nGroups ? groupStart[i + 1] : nHits) - groupStart[i];
			ds += n <= smallMax ? d : 0u;
		}
	}
	for (int o = 32; o > 0; o >>= 1) { c += __shfl_down(c, o); dg += __shfl_down(dg, o); de += __shfl_down(de, o); ds += __shfl_down(ds, o); }
	if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = c; sh[1][threadIdx.x >> 6] = dg; sh[2][threadIdx.x >> 6] = de; sh[3][threadIdx.x >> 6] = ds; }
	__syncthreads();
	if (threadIdx.x == 0)
	{
		u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
		for (int i = 0; i < WG / 64; ++i) { t0 += sh[0][i]; t1 += sh[1][i]; t2 += sh[2][i]; t3 += sh[3][i]; }
		primCnt[q] = t0; dpGroups[q] = t1; dpElems[q] = t2;
		if (t3) atomicAdd(smallElems, (unsigned long long)t3);
	}
}

// primaries of every group -> dense PrimRec array in (query, group, selection) order
__global__ void k_prim_gather(const u64* __restrict__ groupOff, const u32* __restrict__ primCount,
							  const u64* __restrict__ groupStart, const u32* __restrict__ groupExt,
							  const u32* __restrict__ gCur, const u32* __restrict__ gExt, const int4* __restrict__ cand,
							  const i32* __restrict__ len, u32 firstId, int k,
							  const u64* __restrict__ filtOff, const i32* __restrict__ filtPos,
							  const u64* __restrict__ primOff, PrimRec* __restrict__ out,
							  u64* __restrict__ primNode, u64* __restrict__ primBase, u64* __restrict__ matchSize)
{
	__shared__ u32 sh[WG / 64 + 1];
	const u32 q = blockIdx.x;
	const u64 b = groupOff[q], e = groupOff[q + 1];
	u64 obase = primOff[q];
	const i32* fp = filtPos + filtOff[q];
	const i32 nf = (i32)(filtOff[q + 1] - filtOff[q]);
	for (u64 i0 = b; i0 < e; i0 += WG)
	{
		const u64 g = i0 + threadIdx.x;
		const u32 cnt = g < e ? primCount[g] : 0u;
		u32 tot;
		const u32 pos = block_exscan(cnt, sh, &tot);
		if (cnt)
		{
			const u64 g0 = groupStart[g];
			const u32 extId = groupExt[g];
			const i32 extLen = len[(extId - firstId) >> 1];
			for (u32 a = 0; a < cnt; ++a)
			{
				const int4 c4 = cand[g0 + a];
				PrimRec r;
				r.query = q; r.extId = extId;
				r.curBegin = (i32)gCur[g0 + c4.x]; r.extBegin = (i32)gExt[g0 + c4.x];
				r.curEnd = (i32)gCur[g0 + c4.y] + k - 1; r.extEnd = (i32)gExt[g0 + c4.y] + k - 1;
				r.extLen = extLen; r.score = c4.w; r.chainLength = c4.z;
				// repetitive query positions inside [curBegin, curEnd] (overlap.cpp:407-413)
				i32 lo = 0, hi = nf;
				while (lo < hi) { const i32 m = (lo + hi) >> 1; if (fp[m] < r.curBegin) lo = m + 1; else hi = m; }
				const i32 first = lo;
				hi = nf;
				while (lo < hi) { const i32 m = (lo + hi) >> 1; if (fp[m] <= r.curEnd) lo = m + 1; else hi = m; }
				r.filtered = lo - first;
				r.editDistance = -1; r.hpcLenCur = 0; r.hpcLenExt = 0;
				out[obase + pos + a] = r;
				if (primNode)	// keep_alignment
				{
					primNode[obase + pos + a] = g0 + (u64)c4.y;
					primBase[obase + pos + a] = g0;
					matchSize[obase + pos + a] = (u64)c4.z + 2;
				}
			}
		}
		obase += tot;
	}
}

// kmerMatches of every primary (overlap.cpp:368-377, 398-405): the chain is walked again from
// its last hit along the DP's back pointers (chainLength nodes -- where the consuming walk of
// k_chain_finish stopped), keeping a match when it lies > k query bases before the last kept
// one; then (curBegin, extBegin) in front and (curEnd, extEnd) behind.  One thread per
// primary; the list is written right-aligned into the primary's slot of chainLength + 2 pairs.
__global__ void k_chain_matches(const PrimRec* __restrict__ prims, u64 nPrim, const u64* __restrict__ primNode,
								const u64* __restrict__ primBase, const u64* __restrict__ matchOff,
								const u32* __restrict__ gCur, const u32* __restrict__ gExt,
								const i32* __restrict__ gBack, int k, u64* __restrict__ matches,
								u32* __restrict__ matchCnt)
{
	const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= nPrim) return;
	const PrimRec r = prims[p];
	const u64 base = primBase[p];
	u64* slot = matches + matchOff[p];
	const i32 ub = r.chainLength + 2;
	i32 pos = (i32)(primNode[p] - base);
	i32 kept = 0, lastCur = 0;
	for (i32 step = 0; step < r.chainLength; ++step)
	{
		const i32 cu = (i32)gCur[base + pos];
		if (kept == 0 || lastCur - cu > k)
		{
			slot[ub - 2 - kept] = (u64)(u32)cu | ((u64)gExt[base + pos] << 32);
			lastCur = cu;
			++kept;
		}
		pos = gBack[base + pos];
	}
	slot[ub - 2 - kept] = (u64)(u32)r.curBegin | ((u64)(u32)r.extBegin << 32);
	slot[ub - 1] = (u64)(u32)r.curEnd | ((u64)(u32)r.extEnd << 32);
	matchCnt[p] = (u32)kept + 2;
}

template <class T>
T fetchScalar(fg_ctx* c, const T* dptr)
{
	static_assert(sizeof(T) <= 8, "scalar");
	c->hScalar.reserve(8);
	HIP_CHECK(hipMemcpyAsync(c->hScalar.p, dptr, sizeof(T), hipMemcpyDeviceToHost, c->stream));
	HIP_CHECK(hipStreamSynchronize(c->stream));
	T v;
	memcpy(&v, c->hScalar.p, sizeof(T));
	return v;
}

} // namespace

// std::sort order of each segment [segOff[i], segOff[i+1]) of device arrays K, V
#define SORT_MAX_LEVELS 128		// rows of level counts (the depth budget of 2 log2 n bounds the levels: n < 2^31)
template <class KT>
static void sortSegments(fg_ctx* c, const u64* dSegOff, u32 nSeg, KT* dK, u32* dV, u64 nHits, int curBits = 0)
{
	hipStream_t s = c->stream;
	// pieces are disjoint; even the median-of-3 killer stays far below one task per 8 hits
	const u32 smallCap = (u32)std::min<u64>(nHits / 8 + 4ULL * nSeg + 1024, 0x7fffffffULL);
	const u64 bigCap = nHits / SORT_CAP + nSeg + 16;
	c->dTmp32.reserve(std::max<u64>(2 * nHits + 2, 4 * c->hitCapHint + 16));	// the chaining stage wants 4 per hit of it: sized once
	c->dSortTasks.reserve((size_t)smallCap * sizeof(SortTask));
	c->dSortBig.reserve((size_t)(6 * bigCap) * sizeof(SortTask));
	c->dListCnt.reserve(4);
	c->dSortCnt.reserve(4 * (SORT_MAX_LEVELS + 2) + 4);
	SortTask* smallT = (SortTask*)c->dSortTasks.p;
	SortTask* bigA = (SortTask*)c->dSortBig.p;
	SortTask* bigB = bigA + bigCap;
	SortTask* kids = bigB + bigCap;	// 2 * bigCap
	SortTask* wideA = kids + 2 * bigCap;
	SortTask* wideB = wideA + bigCap;
	u32* levelCnt = c->dSortCnt.p;							// row l at levelCnt + 4 l
	u32* smallCnt = c->dSortCnt.p + 4 * (SORT_MAX_LEVELS + 2);
	const u32 streamMax = getenv("FG_SORT_STREAM_MAX") ? (u32)atoi(getenv("FG_SORT_STREAM_MAX")) : 8192u;
	const u32 wideMin = getenv("FG_SORT_WIDE_MIN") ? (u32)atoi(getenv("FG_SORT_WIDE_MIN")) : 16384u;
	// Levels with many pieces of a LARGE chunk (>= 2^29 hits: a level's bytes take milliseconds, longer than the serial
	// chain of its longest streamed piece) stream pieces of up to nHits / 8192 hits.  Measured (tools/many_ab.sh):
	// dmel x 0.25, 1.5 G hits per chunk, sort levels 537 -> 431 ms per pass; on the bench workload (0.33 G hits, a
	// level = 1 ms) the same rule costs 0.5-1.6 ms, hence the gate.  FG_SORT_MANY_MIN=0: fixed thresholds everywhere.
	const u32 manyMin = getenv("FG_SORT_MANY_MIN") ? (u32)atoi(getenv("FG_SORT_MANY_MIN")) : 4096u;
	const u32 streamMany = getenv("FG_SORT_STREAM_MANY") ? (u32)atoi(getenv("FG_SORT_STREAM_MANY"))
		: (nHits >= (1ULL << 29) ? (u32)std::min<u64>(std::max<u64>(nHits / 8192, streamMax), 262144) : streamMax);
	const SortLists lists{smallT, wideMin, smallCap, manyMin, std::max(streamMany, streamMax), streamMax};
	HIP_CHECK(hipMemsetAsync(c->dSortCnt.p, 0, c->dSortCnt.bytes(), s));
	{ ScopedK t(c->timer, "k_sort_level");
	  hipLaunchKernelGGL(k_sort_init, (nSeg + WG - 1) / WG, WG, 0, s, dSegOff, nSeg, lists, bigA, wideA, levelCnt, smallCnt); }
	// counts of level `lvl` (and the small pieces queued so far) to the host
	u32 cnt[3] = {0, 0, 0};		// big, small (total), wide
	auto fetchCounts = [&](int lvl)
	{
		c->hScalar.reserve(8);
		HIP_CHECK(hipMemcpyAsync(c->hScalar.p, levelCnt + 4 * lvl, 16, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipMemcpyAsync((char*)c->hScalar.p + 16, smallCnt, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		u32 row[5];
		memcpy(row, c->hScalar.p, 20);
		cnt[0] = row[0]; cnt[2] = row[2]; cnt[1] = row[4];
	};
	fetchCounts(0);
	static const bool trace = getenv("FG_SORT_TRACE") != nullptr;
	static const char* levelNames[32] = {
		"k_sort_level#00", "k_sort_level#01", "k_sort_level#02", "k_sort_level#03", "k_sort_level#04", "k_sort_level#05",
		"k_sort_level#06", "k_sort_level#07", "k_sort_level#08", "k_sort_level#09", "k_sort_level#10", "k_sort_level#11",
		"k_sort_level#12", "k_sort_level#13", "k_sort_level#14", "k_sort_level#15", "k_sort_level#16", "k_sort_level#17",
		"k_sort_level#18", "k_sort_level#19", "k_sort_level#20", "k_sort_level#21", "k_sort_level#22", "k_sort_level#23",
		"k_sort_level#24", "k_sort_level#25", "k_sort_level#26", "k_sort_level#27", "k_sort_level#28", "k_sort_level#29",
		"k_sort_level#30", "k_sort_level#31+"};
	const u64 narrowMax = getenv("FG_NARROW_MAX") ? strtoull(getenv("FG_NARROW_MAX"), nullptr, 10) : 0xFFFFFFFFULL;
	// levels between two read-backs of the counts: grids are sized by an upper bound (a level has at most twice the
	// tasks of the one before) and the kernels loop over what is really there.  FG_SORT_LEVEL_BATCH=1: a read-back per
	// level, as until round 3 (also what the per-level trace uses).
	const int levelBatch = trace ? 1 : (getenv("FG_SORT_LEVEL_BATCH") ? std::max(1, atoi(getenv("FG_SORT_LEVEL_BATCH"))) : 4);
	int level = 0;
	while (cnt[0] || cnt[2])
	{
		u64 ubBig = cnt[0], ubWide = cnt[2];
		for (int b = 0; b < levelBatch && level < SORT_MAX_LEVELS; ++b)
		{
			if (trace) fprintf(stderr, "sort level %d: %u one-wave tasks, %u wide tasks, %u small pieces so far\n", level, cnt[0], cnt[2], cnt[1]);
			ScopedK t(c->timer, trace ? levelNames[level < 31 ? level : 31] : "k_sort_level");
			const u32* rowL = levelCnt + 4 * level;
			u32* rowN = levelCnt + 4 * (level + 1);
			if (ubWide)	// the long poles first
				hipLaunchKernelGGL(k_sort_wide<KT>, (unsigned)std::min<u64>(ubWide, 2048), SORT_WIDE_WAVES * 64, 0, s, wideA, rowL, dK, dV,
								   c->dTmp32.p, nHits, kids);
			if (ubBig)
				hipLaunchKernelGGL(k_sort_level<KT>, (unsigned)std::min<u64>((ubBig + SORT_LEVEL_WAVES - 1) / SORT_LEVEL_WAVES, 1u << 20),
								   SORT_LEVEL_WAVES * 64, 0, s, bigA, rowL, dK, dV, c->dTmp32.p, nHits, kids);
			const u64 ubKids = 2 * (ubBig + ubWide);
			hipLaunchKernelGGL(k_sort_route, (unsigned)std::min<u64>((ubKids + WG - 1) / WG, 4096), WG, 0, s, kids, rowL, lists, bigB, wideB,
							   rowN, smallCnt);
			++level;
			std::swap(bigA, bigB);
			std::swap(wideA, wideB);
			// what the next level can hold at most
			const u64 nb = std::min<u64>(ubKids, bigCap), nw = std::min<u64>(ubKids, nHits / wideMin + 1);
			ubBig = nb; ubWide = nw;
		}
		if (level >= SORT_MAX_LEVELS) throw FgError{FG_ERR_HIP, "internal: sort level budget exceeded"};
		fetchCounts(level);
	}
	if (cnt[1] > smallCap) throw FgError{FG_ERR_HIP, "internal: sort task queue overflow"};
	if (cnt[1])
	{
		ScopedK t(c->timer, "k_sort_lds");
		hipLaunchKernelGGL(k_sort_lds<KT>, (cnt[1] + SORT_LDS_WAVES - 1) / SORT_LDS_WAVES, SORT_LDS_WAVES * 64, 0, s,
						   smallT, cnt[1], dK, dV, curBits, narrowMax, c->dTmp32.p, nHits);
	}
}

void fgDebugSortPairs(fg_ctx* c, u64* keys, u32* vals, const u64* segOff, u32 nSeg)
{
	hipStream_t s = c->stream;
	const u64 n = segOff[nSeg];
	DevBuf<u64> dK, dOff; DevBuf<u32> dV;
	dK.alloc(n + 1); dV.alloc(n + 1); dOff.alloc(nSeg + 1);
	HIP_CHECK(hipMemcpyAsync(dK.p, keys, n * 8, hipMemcpyHostToDevice, s));
	HIP_CHECK(hipMemcpyAsync(dV.p, vals, n * 4, hipMemcpyHostToDevice, s));
	HIP_CHECK(hipMemcpyAsync(dOff.p, segOff, (nSeg + 1) * 8ULL, hipMemcpyHostToDevice, s));
	if (nSeg) sortSegments<u64>(c, dOff.p, nSeg, dK.p, dV.p, n);
	HIP_CHECK(hipMemcpyAsync(keys, dK.p, n * 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipMemcpyAsync(vals, dV.p, n * 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
}

// seed collection's probe step with the probes partitioned by table region (see k_probe_emit): fills c->dProbe,
// c->dCntA (hits per query), c->dCntB (repetitive positions per query) like k_probe
static void probePartitioned(fg_ctx* c, u32 nq, const u64* localOff, const u64* qWords, const u64* qWordOff, const i32* qLen)
{
	hipStream_t s = c->stream;
	const u64 totalQK = localOff[nq];
	HIP_CHECK(hipMemsetAsync(c->dProbe.p, 0, totalQK * 8, s));
	int regionShift = 0;
	while ((c->tableSlots >> regionShift) > (1ULL << PROBE_REGION_BITS)) ++regionShift;
	const u64 subBudget = getenv("FG_PROBE_SUB_KMERS") ? std::max<u64>(1, strtoull(getenv("FG_PROBE_SUB_KMERS"), nullptr, 10)) : (256ULL << 20);
	u32 q0 = 0;
	while (q0 < nq)
	{
		u32 q1 = q0 + 1;
		while (q1 < nq && localOff[q1 + 1] - localOff[q0] <= subBudget) ++q1;
		const u64 n = localOff[q1] - localOff[q0];
		if (n)
		{
			c->dPartK0.reserve(n); c->dPartV0.reserve(n); c->dPartK1.reserve(n); c->dPartV1.reserve(n);
			c->dPartScratch.reserve(fgprim::radixSortScratchBytes(n));
			{ ScopedK t(c->timer, "k_probe_emit");
			  hipLaunchKernelGGL(k_probe_emit, q1 - q0, WG, 0, s, q0, c->dQuery.p, qWords, qWordOff, qLen, c->dKmerOff.p, c->dQKmerOff.p,
								 c->k, c->table, regionShift, c->hasQ ? (const u32*)nullptr : c->dIndexedBits.p, c->dPartK0.p, c->dPartV0.p); }
			int which;
			{ ScopedK t(c->timer, "k_probe_partition");
			  which = fgprim::radixSortPairs(s, c->dPartK0.p, c->dPartV0.p, c->dPartK1.p, c->dPartV1.p, n, PART_KEY_BITS,
											 PART_KEY_BITS + PROBE_REGION_BITS, c->dPartScratch.p); }
			{ ScopedK t(c->timer, "k_probe");
			  hipLaunchKernelGGL(k_probe_sorted, (unsigned)((n + WG - 1) / WG), WG, 0, s, which ? c->dPartK1.p : c->dPartK0.p,
								 which ? c->dPartV1.p : c->dPartV0.p, n, c->table, c->dProbe.p,
								 getenv("FG_ABLATE_PROBE") ? atoi(getenv("FG_ABLATE_PROBE")) : 0); }
		}
		q0 = q1;
	}
	{ ScopedK t(c->timer, "k_probe_count");
	  hipLaunchKernelGGL(k_probe_count, nq, WG, 0, s, c->dQKmerOff.p, c->dProbe.p, c->dCntA.p, c->dCntB.p); }
}

// Device part of one chunk of queries [qa, qb): seed collection -> sort -> groups ->
// chaining -> (edit distance) -> compacted primaries in c->hPrim / offsets in c->hOff.
// Returns false (nothing done) when the chunk's hits exceed the budget and it can be split.
struct ChunkResult { u64 nPrim, nHits, dpGroups, dpElems, dpElemsSmall, nMatchSlots; };

// Probe step of one chunk of queries [qa, qb) (bounded by the k-mer budget): every query k-mer's table value in
// c->dProbe, hits / repetitive positions per query in c->dCntA / c->dCntB; hitsPerQuery = the former on the host.
// The chunk's hits may exceed the hit budget: the caller then runs the rest of the stage over sub-ranges of the
// chunk's queries -- the probes are NOT repeated (they were, once per halving, until round 3: a D. melanogaster-
// sized pass probed every read four times).
static void probeChunk(fg_ctx* c, const u32* hq, const u64* hQKmerOff, u32 qa, u32 qb, std::vector<u64>& hitsPerQuery)
{
	hipStream_t s = c->stream;
	const int k = c->k;
	const u32 nq = qb - qa;
	const u64 totalQK = hQKmerOff[qb] - hQKmerOff[qa];
	std::vector<u64> localOff(nq + 1);
	for (u32 i = 0; i <= nq; ++i) localOff[i] = hQKmerOff[qa + i] - hQKmerOff[qa];
	c->dQuery.reserve(nq); c->dQKmerOff.reserve(nq + 1);
	c->dProbe.reserve(totalQK);
	c->dHitOff.reserve(nq + 1); c->dFiltOff.reserve(nq + 1);
	c->dCntA.reserve(nq + 1); c->dCntB.reserve(nq + 1); c->dGroupCnt.reserve(nq + 1); c->dGroupOff.reserve(nq + 1);
	c->dPrimCnt.reserve(nq + 1); c->dPrimOff.reserve(nq + 1); c->dDpGroups.reserve(nq + 1); c->dDpElems.reserve(nq + 1);
	c->dListCnt.reserve(4);
	HIP_CHECK(hipMemcpyAsync(c->dQuery.p, hq + qa, nq * 4ULL, hipMemcpyHostToDevice, s));
	HIP_CHECK(hipMemcpyAsync(c->dQKmerOff.p, localOff.data(), (nq + 1) * 8ULL, hipMemcpyHostToDevice, s));

	const u64* qWords = c->hasQ ? c->dQWords.p : c->dWords.p;
	const u64* qWordOff = c->hasQ ? c->dQWordOff.p : c->dWordOff.p;
	const i32* qLen = c->hasQ ? c->dQLen.p : c->dLen.p;
	// FG_PROBE_PARTITION=1: probes ordered by table region first (measured, D. melanogaster-sized table of 9 GB:
	// 26 ms against 20 ms in query order per 0.48 G probes -- the table lines are not what the probe waits for)
	const bool partition = !c->tableWide && c->k <= 17 &&
		(getenv("FG_PROBE_PARTITION") ? atoi(getenv("FG_PROBE_PARTITION")) != 0 : false);
	if (partition) probePartitioned(c, nq, localOff.data(), qWords, qWordOff, qLen);
	else
	{ ScopedK t(c->timer, "k_probe");
	  if (c->tableWide)
		hipLaunchKernelGGL(k_probe<true>, nq, WG, 0, s, c->dQuery.p, qWords, qWordOff, qLen, c->dKmerOff.p,
						   c->dQKmerOff.p, k, c->table, c->hasQ ? (const u32*)nullptr : c->dIndexedBits.p,
						   c->dProbe.p, c->dCntA.p, c->dCntB.p);
	  else
		hipLaunchKernelGGL(k_probe<false>, nq, WG, 0, s, c->dQuery.p, qWords, qWordOff, qLen, c->dKmerOff.p,
						   c->dQKmerOff.p, k, c->table, c->hasQ ? (const u32*)nullptr : c->dIndexedBits.p,
						   c->dProbe.p, c->dCntA.p, c->dCntB.p); }
	hitsPerQuery.resize(nq);
	HIP_CHECK(hipMemcpyAsync(hitsPerQuery.data(), c->dCntA.p, nq * 8ULL, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
}

// The stage behind the probes for the queries [sub0, sub0 + nq) of the probed chunk: seed expansion -> sort ->
// groups -> chaining -> (edit distance) -> compacted primaries in c->hPrim (behind primBase records) / offsets in
// c->hOff.
static void deviceSub(fg_ctx* c, const fg_detector_params* p, uint8_t forceLocal, u32 sub0, u32 nq, u64 primBase,
					  u64 hitCapHint /* the largest sub-range of the chunk: scratch is sized once, not regrown */, ChunkResult* res)
{
	hipStream_t s = c->stream;
	const int k = c->k;
	const u32* dQuery = c->dQuery.p + sub0;
	const u64* dQKmerOff = c->dQKmerOff.p + sub0;		// values index the chunk's probe array
	c->curQuery = dQuery;
	const i32* qLen = c->hasQ ? c->dQLen.p : c->dLen.p;
	// per-query scratch of THIS lane (the probe step reserved the first lane's for the whole chunk)
	c->dHitOff.reserve(nq + 1); c->dFiltOff.reserve(nq + 1);
	c->dGroupCnt.reserve(nq + 1); c->dGroupOff.reserve(nq + 1);
	c->dPrimCnt.reserve(nq + 1); c->dPrimOff.reserve(nq + 1); c->dDpGroups.reserve(nq + 1); c->dDpElems.reserve(nq + 1);
	c->dListCnt.reserve(4);
	{ ScopedK t(c->timer, "k_exscan");
	  hipLaunchKernelGGL(k_exscan, 1, 1024, 0, s, c->dCntA.p + sub0, c->dHitOff.p, nq);
	  hipLaunchKernelGGL(k_exscan, 1, 1024, 0, s, c->dCntB.p + sub0, c->dFiltOff.p, nq); }
	const u64 nHits = fetchScalar(c, c->dHitOff.p + nq);
	const u64 nFilt = fetchScalar(c, c->dFiltOff.p + nq);
	res->nHits = nHits;
	const u64 hitCap = std::max(nHits, hitCapHint);		// what the per-hit buffers below are reserved for
	c->hitCapHint = hitCap;
	c->dFiltPos.reserve(nFilt + 1);
	// 32-bit sort keys when (record index, query position) fit together
	int curBits = 1, recBits = 1;
	while ((1LL << curBits) < (long long)(c->hasQ ? c->qMaxLen : c->maxLen)) ++curBits;
	while ((1ULL << recBits) < 2ULL * c->nReads) ++recBits;
	const bool key32 = curBits + recBits <= 32 && !getenv("FG_FORCE_KEY64");
	// key mode: 0 = 32-bit keys + values; when (record, curPos) need more than 32 bits: 1 = packed 64-bit
	// records (PK: record, curPos, extPos in one word, 8 bytes per hit in the sort levels instead of 12;
	// FG_PACKED_KEYS=0 turns it off), else 2 = 64-bit keys + values
	const bool canPack = recBits + curBits + FG_PK_VALBITS <= 64 && c->maxLen < (1 << FG_PK_VALBITS);
	const bool wantPack = !(getenv("FG_PACKED_KEYS") && atoi(getenv("FG_PACKED_KEYS")) == 0);
	const int keyMode = key32 ? 0 : ((canPack && wantPack) ? 1 : 2);
	if (keyMode == 0)
	{
		c->dHitKey32.reserve(hitCap + 1); c->dHitVal.reserve(hitCap + 1);
		{ ScopedK t(c->timer, "k_fill");
		  hipLaunchKernelGGL(k_fill<u32>, nq, WG, 0, s, dQuery, c->dLen.p, qLen, dQKmerOff, k, c->firstId, curBits,
							 c->dProbe.p, c->dEntries.p, c->dHitOff.p, c->dFiltOff.p, c->dHitKey32.p, c->dHitVal.p, c->dFiltPos.p); }
		sortSegments<u32>(c, c->dHitOff.p, nq, c->dHitKey32.p, c->dHitVal.p, nHits);
	}
	else if (keyMode == 1)
	{
		c->dHitKey.reserve(hitCap + 1);
		{ ScopedK t(c->timer, "k_fill");
		  hipLaunchKernelGGL(k_fill<PK>, nq, WG, 0, s, dQuery, c->dLen.p, qLen, dQKmerOff, k, c->firstId, curBits,
							 c->dProbe.p, c->dEntries.p, c->dHitOff.p, c->dFiltOff.p, (PK*)c->dHitKey.p, (u32*)nullptr, c->dFiltPos.p); }
		sortSegments<PK>(c, c->dHitOff.p, nq, (PK*)c->dHitKey.p, (u32*)nullptr, nHits, curBits);
	}
	else
	{
		c->dHitKey.reserve(hitCap + 1); c->dHitVal.reserve(hitCap + 1);
		{ ScopedK t(c->timer, "k_fill");
		  hipLaunchKernelGGL(k_fill<u64>, nq, WG, 0, s, dQuery, c->dLen.p, qLen, dQKmerOff, k, c->firstId, 0,
							 c->dProbe.p, c->dEntries.p, c->dHitOff.p, c->dFiltOff.p, c->dHitKey.p, c->dHitVal.p, c->dFiltPos.p); }
		sortSegments<u64>(c, c->dHitOff.p, nq, c->dHitKey.p, c->dHitVal.p, nHits, curBits);
	}
	const HitKeyView<u32> hk32{c->dHitKey32.p, c->dHitVal.p, curBits, c->firstId};
	const HitKeyView<u64> hk64{c->dHitKey.p, c->dHitVal.p, curBits, c->firstId};
	const HitKeyView<PK> hkp{(const PK*)c->dHitKey.p, nullptr, curBits, c->firstId};
	{ ScopedK t(c->timer, "k_group_count");
	  if (keyMode == 0) hipLaunchKernelGGL(k_group_count<u32>, nq, WG, 0, s, c->dHitOff.p, hk32, c->dGroupCnt.p);
	  else if (keyMode == 1) hipLaunchKernelGGL(k_group_count<PK>, nq, WG, 0, s, c->dHitOff.p, hkp, c->dGroupCnt.p);
	  else hipLaunchKernelGGL(k_group_count<u64>, nq, WG, 0, s, c->dHitOff.p, hk64, c->dGroupCnt.p); }
	{ ScopedK t(c->timer, "k_exscan");
	  hipLaunchKernelGGL(k_exscan, 1, 1024, 0, s, c->dGroupCnt.p, c->dGroupOff.p, nq); }
	const u64 nGroups = fetchScalar(c, c->dGroupOff.p + nq);
	if (nGroups >= 0xFFFFFFFFULL) throw FgError{FG_ERR_ARG, "too many target groups in one chunk"};
	c->dGroupStart.reserve(nGroups + 1); c->dGroupQuery.reserve(nGroups + 1);
	c->dGroupExt.reserve(nGroups + 1); c->dGroupFirstCur.reserve(nGroups + 1); c->dGroupLastCur.reserve(nGroups + 1);
	c->dPrimFlag.reserve(nGroups + 1); c->dDpSize.reserve(nGroups + 1);
	{ ScopedK t(c->timer, "k_group_fill");
	  if (keyMode == 0) hipLaunchKernelGGL(k_group_fill<u32>, nq, WG, 0, s, c->dHitOff.p, hk32, c->dGroupOff.p, c->dGroupStart.p,
										   c->dGroupQuery.p, c->dGroupExt.p, c->dGroupFirstCur.p, c->dGroupLastCur.p);
	  else if (keyMode == 1) hipLaunchKernelGGL(k_group_fill<PK>, nq, WG, 0, s, c->dHitOff.p, hkp, c->dGroupOff.p, c->dGroupStart.p,
												c->dGroupQuery.p, c->dGroupExt.p, c->dGroupFirstCur.p, c->dGroupLastCur.p);
	  else hipLaunchKernelGGL(k_group_fill<u64>, nq, WG, 0, s, c->dHitOff.p, hk64, c->dGroupOff.p, c->dGroupStart.p,
							  c->dGroupQuery.p, c->dGroupExt.p, c->dGroupFirstCur.p, c->dGroupLastCur.p); }
	fgChainStage(c, p, forceLocal, nGroups, nHits, keyMode, curBits);
	c->dSmallElems.reserve(1);
	HIP_CHECK(hipMemsetAsync(c->dSmallElems.p, 0, 8, s));
	{ ScopedK t(c->timer, "k_prim_count");
	  hipLaunchKernelGGL(k_prim_count, nq, WG, 0, s, c->dGroupOff.p, c->dPrimFlag.p, c->dDpSize.p, c->dPrimCnt.p,
						 c->dDpGroups.p, c->dDpElems.p, c->dGroupStart.p, nGroups, nHits, fgChainSmallMax(), c->dSmallElems.p); }
	{ ScopedK t(c->timer, "k_exscan");
	  hipLaunchKernelGGL(k_exscan, 1, 1024, 0, s, c->dPrimCnt.p, c->dPrimOff.p, nq); }
	const u64 nPrim = fetchScalar(c, c->dPrimOff.p + nq);
	c->dPrimOut.reserve((nPrim + 1) * sizeof(PrimRec));
	const bool keepAln = p->keep_alignment;
	if (keepAln) { c->dPrimNode.reserve(nPrim + 1); c->dPrimBase.reserve(nPrim + 1); c->dMatchSize.reserve(nPrim + 1);
				   c->dMatchOff.reserve(nPrim + 2); c->dMatchCnt.reserve(nPrim + 1); }
	{ ScopedK t(c->timer, "k_prim_gather");
	  hipLaunchKernelGGL(k_prim_gather, nq, WG, 0, s, c->dGroupOff.p, c->dPrimFlag.p, c->dGroupStart.p, c->dGroupExt.p,
						 c->dCur.p, c->dExt.p, c->dCand.p, c->dLen.p, c->firstId, k, c->dFiltOff.p, c->dFiltPos.p,
						 c->dPrimOff.p, (PrimRec*)c->dPrimOut.p, keepAln ? c->dPrimNode.p : (u64*)nullptr,
						 keepAln ? c->dPrimBase.p : (u64*)nullptr, keepAln ? c->dMatchSize.p : (u64*)nullptr); }
	if (p->nucl_alignment) fgEditDistances(c, (PrimRec*)c->dPrimOut.p, nPrim, p->use_hpc);
	res->nMatchSlots = 0;
	if (keepAln && nPrim)
	{
		if (nPrim >= 0xFFFFFFFFULL) throw FgError{FG_ERR_ARG, "too many overlaps in one chunk"};
		{ ScopedK t(c->timer, "k_exscan");
		  hipLaunchKernelGGL(k_exscan, 1, 1024, 0, s, c->dMatchSize.p, c->dMatchOff.p, (u32)nPrim); }
		const u64 slots = fetchScalar(c, c->dMatchOff.p + nPrim);
		res->nMatchSlots = slots;
		c->dMatches.reserve(slots + 1);
		{ ScopedK t(c->timer, "k_chain_matches");
		  hipLaunchKernelGGL(k_chain_matches, (unsigned)((nPrim + WG - 1) / WG), WG, 0, s, (const PrimRec*)c->dPrimOut.p, nPrim,
							 c->dPrimNode.p, c->dPrimBase.p, c->dMatchOff.p, c->dCur.p, c->dExt.p, c->dBack.p, k,
							 c->dMatches.p, c->dMatchCnt.p); }
		c->hMatches.reserve(slots + 1); c->hMatchOff.reserve(nPrim + 1); c->hMatchCnt.reserve(nPrim + 1);
		ScopedK t(c->timer, "copy_results_d2h");
		HIP_CHECK(hipMemcpyAsync(c->hMatches.p, c->dMatches.p, slots * 8, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipMemcpyAsync(c->hMatchOff.p, c->dMatchOff.p, (nPrim + 1) * 8, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipMemcpyAsync(c->hMatchCnt.p, c->dMatchCnt.p, nPrim * 4, hipMemcpyDeviceToHost, s));
	}
	// the primaries of all chunks of the call end up one behind the other in the pinned buffer: this chunk's land
	// behind the primBase records of the chunks before it
	if ((primBase + nPrim + 1) * sizeof(PrimRec) > c->hPrim.n)
	{
		HIP_CHECK(hipStreamSynchronize(s));		// copies of earlier chunks may still be on their way into the old buffer
		c->hPrim.reserveKeep((primBase + nPrim + 1) * sizeof(PrimRec), primBase * sizeof(PrimRec));
	}
	c->hOff.reserve(3 * (size_t)(nq + 1) + 1);
	{ ScopedK t(c->timer, "copy_results_d2h");
	  HIP_CHECK(hipMemcpyAsync(c->hOff.p, c->dPrimOff.p, (nq + 1) * 8ULL, hipMemcpyDeviceToHost, s));
	  HIP_CHECK(hipMemcpyAsync(c->hOff.p + (nq + 1), c->dDpGroups.p, nq * 8ULL, hipMemcpyDeviceToHost, s));
	  HIP_CHECK(hipMemcpyAsync(c->hOff.p + 2 * (size_t)(nq + 1), c->dDpElems.p, nq * 8ULL, hipMemcpyDeviceToHost, s));
	  HIP_CHECK(hipMemcpyAsync(c->hOff.p + 3 * (size_t)(nq + 1), c->dSmallElems.p, 8, hipMemcpyDeviceToHost, s));
	  HIP_CHECK(hipEventRecord(c->evOff, s));
	  // the records in pieces, an event behind each: the caller does not wait for them here (the host shim's
	  // threads wait for the piece they read)
	  const int nPieces = nPrim >= 100000 ? FG_D2H_PIECES : 1;	// a small result is one copy (the unused events are recorded all the same)
	  for (int i = 0; i < FG_D2H_PIECES; ++i)
	  {
		  const u64 a = i < nPieces ? nPrim * i / nPieces : nPrim, b = i < nPieces ? nPrim * (i + 1) / nPieces : nPrim;
		  if (b > a)
			  HIP_CHECK(hipMemcpyAsync(c->hPrim.p + (primBase + a) * sizeof(PrimRec), c->dPrimOut.p + a * sizeof(PrimRec),
									   (b - a) * sizeof(PrimRec), hipMemcpyDeviceToHost, s));
		  HIP_CHECK(hipEventRecord(c->evPiece[i], s));
		  c->pieceEnd[i] = primBase + b;
	  } }
	HIP_CHECK(hipEventSynchronize(c->evOff));
	if (keepAln) HIP_CHECK(hipStreamSynchronize(s));		// the match lists are read by the caller right away
	res->nPrim = nPrim;
	res->dpGroups = 0; res->dpElems = 0;
	res->dpElemsSmall = c->hOff.p[3 * (size_t)(nq + 1)];
	for (u32 i = 0; i < nq; ++i) { res->dpGroups += c->hOff.p[(nq + 1) + i]; res->dpElems += c->hOff.p[2 * (size_t)(nq + 1) + i]; }
}

// ---- two lanes ------------------------------------------------------------------------------------------------
// The stage behind the probes alternates between bandwidth-bound kernels (seed expansion, the sort levels) and
// issue- / latency-bound ones that end on a handful of long waves (the per-group kernels of the chaining stage).  A call
// with enough hits is therefore cut into a few sub-ranges that two LANES work through side by side: the second lane is
// a context of its own scratch, streams and timers (c->lane2) whose read / index / probe buffers are views of this
// one's, driven by a helper thread; while one lane sits in its chaining kernels the other runs its sort levels.
// Results do not depend on the cut (test_internal_chunking_is_invisible); FG_LANES=1 keeps everything on one lane.
// Measured (MI355X, gpurun_out/r03_lane_sweep*.log, r03_lanes_wl.log): D. melanogaster-like x0.5 (its chunks need several
// sub-ranges anyway) 2185 -> 2009 ms per pass.
static fg_ctx* laneTwo(fg_ctx* c)
{
	if (!c->lane2)
	{
		std::unique_ptr<fg_ctx> l(new fg_ctx);
		l->device = c->device;
		if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess ||
			hipStreamCreateWithFlags(&l->stream2, hipStreamNonBlocking) != hipSuccess ||
			hipStreamCreateWithFlags(&l->stream3, hipStreamNonBlocking) != hipSuccess ||
			hipEventCreateWithFlags(&l->evJoin3, hipEventDisableTiming) != hipSuccess ||
			hipEventCreateWithFlags(&l->evOff, hipEventDisableTiming) != hipSuccess ||
			hipEventCreateWithFlags(&l->evFork, hipEventDisableTiming) != hipSuccess ||
			hipEventCreateWithFlags(&l->evJoin, hipEventDisableTiming) != hipSuccess)
			throw FgError{FG_ERR_HIP, "streams of the second lane"};
		for (auto& e : l->evPiece)
			if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) throw FgError{FG_ERR_HIP, "events of the second lane"};
		l->timer.stream = l->stream;
		c->lane2 = std::move(l);
	}
	fg_ctx* l = c->lane2.get();
	// views of everything the stage reads (refreshed per chunk: the probe buffers may have been regrown)
	l->k = c->k; l->nReads = c->nReads; l->firstId = c->firstId; l->maxLen = c->maxLen;
	l->totalKmers = c->totalKmers; l->totalBases = c->totalBases;
	l->hasQ = c->hasQ; l->nQReads = c->nQReads; l->qFirstId = c->qFirstId; l->qMaxLen = c->qMaxLen;
	l->indexBuilt = c->indexBuilt; l->sampleRate = c->sampleRate; l->nKeys = c->nKeys; l->nEntries = c->nEntries; l->nRep = c->nRep;
	l->tableSlots = c->tableSlots; l->tableWide = c->tableWide; l->table = c->table;
	l->timer.enabled = c->timer.enabled;
	l->dWords.alias(c->dWords); l->dWordOff.alias(c->dWordOff); l->dLen.alias(c->dLen); l->dKmerOff.alias(c->dKmerOff);
	l->dQWords.alias(c->dQWords); l->dQWordOff.alias(c->dQWordOff); l->dQLen.alias(c->dQLen);
	l->dKeyOff.alias(c->dKeyOff); l->dEntries.alias(c->dEntries); l->dTable.alias(c->dTable); l->dIndexedBits.alias(c->dIndexedBits);
	l->dQuery.alias(c->dQuery); l->dQKmerOff.alias(c->dQKmerOff); l->dProbe.alias(c->dProbe);
	l->dCntA.alias(c->dCntA); l->dCntB.alias(c->dCntB);
	return l;
}

// one sub-range's results, kept until the host shim runs
struct SubResult {
	u32 qa = 0, qb = 0;				// queries of the call
	int lane = 0;
	u64 lanePrimBase = 0;			// its primaries lie behind this many records of its lane's pinned buffer
	ChunkResult cr{};
	std::vector<u64> primOff;		// qb - qa + 1, local
	std::vector<u64> mData, mOff;	// keep_alignment
};

void fgOverlaps(fg_ctx* c, const fg_detector_params* p, const u32* queryIds, u32 nq, i32 maxOverlaps,
				uint8_t forceLocal, fg_overlap_batch* out)
{
	hipStream_t s = c->stream;
	const int k = c->k;
	c->timer.reset();	// a call that failed half way leaves recorded events behind: back to the pool
	c->hitCapHint = 0;
	const auto tHost0 = std::chrono::steady_clock::now();
	// the two bracket events come from the timer's pool and go back to it on every exit path
	struct EvPair {
		KernelTimer& t; hipEvent_t a, b;
		explicit EvPair(KernelTimer& t_) : t(t_), a(t_.get()), b(nullptr) { try { b = t_.get(); } catch (...) { t_.pool.push_back(a); throw; } }
		~EvPair() { t.pool.push_back(a); t.pool.push_back(b); }
	} evp(c->timer);
	const hipEvent_t evA = evp.a, evB = evp.b;
	HIP_CHECK(hipEventRecord(evA, s));

	BatchOwner* own = BatchOwner::acquire();
	out->owner_ = own;
	out->n_queries = nq;
	own->queryOff.assign(nq + 1, 0);
	own->statOff.assign(nq + 1, 0);

	// query table
	std::vector<u32> hq(nq);
	std::vector<u64> hQKmerOff(nq + 1, 0);
	u64 queryBp = 0;
	for (u32 i = 0; i < nq; ++i)
	{
		hq[i] = queryIds[i] - (c->hasQ ? c->qFirstId : c->firstId);
		const i32 L = (c->hasQ ? c->hQLen : c->hLen)[hq[i] >> 1];
		hQKmerOff[i + 1] = hQKmerOff[i] + (u64)std::max(0, L - k);
		queryBp += L;
	}
	out->query_bp = queryBp;
	out->query_kmers = hQKmerOff[nq];
	if (nq == 0)
	{
		out->query_off = own->queryOff.data(); out->div_stats_off = own->statOff.data();
		if (p->keep_alignment) { own->matchOff.assign(1, 0); out->match_off = own->matchOff.data(); }
		if (p->partition_bad_mappings) { own->needsTrim.assign(1, 0); out->needs_trim = own->needsTrim.data(); }
		return;
	}
#ifdef FG_SORT_ABLATE
	if (getenv("FG_ABLATE"))
	{
		const int ab = atoi(getenv("FG_ABLATE"));
		HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(wsort::g_ablate), &ab, sizeof(int)));
	}
#endif

	// The batch is cut into chunks so that the per-chunk scratch (8 B per query k-mer,
	// ~70 B per seed hit) stays bounded whatever the caller passes; a chunk whose hits
	// exceed the budget is halved.  E. coli 50x is one chunk.
	u64 kmerBudget = getenv("FG_KMER_BUDGET") ? strtoull(getenv("FG_KMER_BUDGET"), nullptr, 10) : (1ULL << 30);
	u64 hitBudget = getenv("FG_HIT_BUDGET") ? strtoull(getenv("FG_HIT_BUDGET"), nullptr, 10) : (3ULL << 29);
	if (!getenv("FG_KMER_BUDGET") && !getenv("FG_HIT_BUDGET"))
	{
		// ... and by what the device has left beside the resident index (a CHM13-sized index leaves ~80 of 288 GB):
		// the chunk scratch may grow into the free memory plus what it already holds, less a reserve
		size_t freeB = 0, totalB = 0;
		HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
		const u64 held = c->dHitKey.bytes() + c->dHitKey32.bytes() + c->dHitVal.bytes() + c->dCur.bytes() + c->dExt.bytes() +
						 c->dScore.bytes() + c->dBack.bytes() + c->dTmp32.bytes() + c->dCand.bytes() + c->dProbe.bytes();
		const u64 avail = (u64)freeB + held;
		const u64 usable = avail > (8ULL << 30) ? (avail - (4ULL << 30)) / 10 * 9 : avail / 2;
		hitBudget = std::min(hitBudget, std::max<u64>(16ULL << 20, usable / 100 * 85 / 80));	// ~70 B per hit, grow-only slack
		kmerBudget = std::min(kmerBudget, std::max<u64>(16ULL << 20, usable / 100 * 15 / 9));	// 8 B per query k-mer
	}
	std::vector<std::pair<u32, u32>> chunks;	// [qa, qb) bounded by the k-mer budget
	{
		u32 qa = 0;
		while (qa < nq)
		{
			u32 qb = qa + 1;
			while (qb < nq && hQKmerOff[qb + 1] - hQKmerOff[qa] <= kmerBudget) ++qb;
			chunks.push_back({qa, qb});
			qa = qb;
		}
	}
	const bool keepAln = p->keep_alignment;
	std::vector<u64> mData, mOff(1, 0);	// keep_alignment: compacted kmerMatches per primary
	std::vector<u64> primOffAll(nq + 1, 0);
	std::vector<const PrimRec*> primPtr(nq, nullptr);	// primary j (counted over the call) of query qi = primPtr[qi][j]
	std::vector<u64> hitsPerQuery;
	u64 nPrim = 0;
	out->seed_hits = 0; out->dp_groups = 0; out->dp_elements = 0; out->dp_elements_small = 0;
	const int nLanes = getenv("FG_LANES") ? std::max(1, std::min(2, atoi(getenv("FG_LANES")))) : 2;
	const u64 laneMinHits = getenv("FG_LANE_MIN_HITS") ? strtoull(getenv("FG_LANE_MIN_HITS"), nullptr, 10) : (48ULL << 20);
	const u32 laneSplit = getenv("FG_LANE_SPLIT") ? (u32)std::max(2, atoi(getenv("FG_LANE_SPLIT"))) : 4u;
	bool laned = false;
	u64 lanePrim[2] = {0, 0};		// primaries in each lane's pinned buffer so far
	std::vector<std::unique_ptr<SubResult>> subs;
	for (const auto& ch : chunks)
	{
		probeChunk(c, hq.data(), hQKmerOff.data(), ch.first, ch.second, hitsPerQuery);
		const u32 cn = ch.second - ch.first;
		u64 chunkHits = 0;
		for (u32 i = 0; i < cn; ++i) chunkHits += hitsPerQuery[i];
		// sub-ranges of the chunk's queries whose hits fit the budget (a single query above it is one of its own);
		// with two lanes a chunk of enough hits is cut into at least laneSplit of them
		// Two lanes where the chunk has to be cut anyway (its hits exceed the budget: then each lane takes pieces of half
		// the budget).  Cutting a chunk that fits only to have two lanes pays on some workloads and costs on others
		// (every piece ends on the long waves of its chaining kernels): E. coli PB 50x 55.9 -> 53.9 ms in two halves,
		// 64 ms in four; HiFi parameters 265 -> 289 ms -- so that is left to FG_LANE_MIN_HITS (experiments).
		const bool forcedCut = getenv("FG_LANE_MIN_HITS") && chunkHits >= laneMinHits && cn >= 2 * laneSplit;
		const bool twoLanes = nLanes == 2 && (chunkHits > hitBudget || forcedCut);
		const u64 budget = !twoLanes ? hitBudget
			: (chunkHits > hitBudget ? std::max<u64>(1, hitBudget / 2) : std::max<u64>(1, (chunkHits + laneSplit - 1) / laneSplit));
		std::vector<std::pair<u32, u32>> ranges;
		u64 maxSubHits = 0;
		// experiment: FG_LANE_FRACS="25,50,25" cuts the chunk at those shares of its hits (uneven pieces put the two
		// lanes out of phase: one in its sort levels while the other is in its chaining kernels)
		std::vector<u64> cuts;
		if (twoLanes && getenv("FG_LANE_FRACS") && chunkHits / 2 <= hitBudget)
		{
			u64 accPct = 0;
			for (const char* q = getenv("FG_LANE_FRACS"); *q;)
			{
				accPct += strtoull(q, (char**)&q, 10);
				if (*q == ',') ++q;
				cuts.push_back(chunkHits / 100 * std::min<u64>(accPct, 100));
			}
		}
		size_t cutI = 0;
		u64 before = 0;
		for (u32 sa = 0; sa < cn;)
		{
			u32 sb = sa + 1;
			u64 acc = hitsPerQuery[sa];
			if (!cuts.empty())
			{
				const u64 upTo = cutI + 1 < cuts.size() ? cuts[cutI] : chunkHits;
				while (sb < cn && before + acc + hitsPerQuery[sb] <= upTo) acc += hitsPerQuery[sb++];
				if (cutI + 1 >= cuts.size()) { while (sb < cn) acc += hitsPerQuery[sb++]; }
				++cutI;
			}
			else
				while (sb < cn && acc + hitsPerQuery[sb] <= budget) acc += hitsPerQuery[sb++];
			ranges.push_back({sa, sb});
			maxSubHits = std::max(maxSubHits, acc);
			before += acc;
			sa = sb;
		}
		const size_t sub0Index = subs.size();
		for (auto& r : ranges)
		{
			subs.emplace_back(new SubResult);
			subs.back()->qa = ch.first + r.first; subs.back()->qb = ch.first + r.second;
		}
		fg_ctx* lanes[2] = {c, nullptr};
		const bool useTwo = twoLanes && ranges.size() >= 2;
		if (useTwo) { lanes[1] = laneTwo(c); if (!laned) lanes[1]->timer.reset(); laned = true; }
		std::atomic<size_t> next{0};
		std::exception_ptr laneErr[2];
		auto work = [&](int lane)
		{
			try
			{
				fg_ctx* lc = lanes[lane];
				(void)hipSetDevice(lc->device);
				while (true)
				{
					const size_t i = next.fetch_add(1);
					if (i >= ranges.size()) break;
					SubResult& sr = *subs[sub0Index + i];
					sr.lane = lane;
					sr.lanePrimBase = lanePrim[lane];
					deviceSub(lc, p, forceLocal, ranges[i].first, ranges[i].second - ranges[i].first, lanePrim[lane], maxSubHits, &sr.cr);
					const u32 n = ranges[i].second - ranges[i].first;
					sr.primOff.assign(lc->hOff.p, lc->hOff.p + n + 1);
					if (keepAln)
					{
						sr.mOff.assign(1, 0);
						sr.mData.reserve(sr.cr.nMatchSlots);
						for (u64 j = 0; j < sr.cr.nPrim; ++j)
						{
							const u64 slotEnd = lc->hMatchOff.p[j + 1];
							const u32 cnt = lc->hMatchCnt.p[j];
							sr.mData.insert(sr.mData.end(), lc->hMatches.p + slotEnd - cnt, lc->hMatches.p + slotEnd);
							sr.mOff.push_back(sr.mData.size());
						}
					}
					lanePrim[lane] += sr.cr.nPrim;
				}
				if (useTwo) HIP_CHECK(hipStreamSynchronize(lc->stream));	// the next chunk's probes overwrite what the sub-ranges read
			}
			catch (...) { laneErr[lane] = std::current_exception(); }
		};
		if (useTwo)
		{
			std::thread helper(work, 1);
			work(0);
			helper.join();
		}
		else work(0);
		for (int l = 0; l < 2; ++l) if (laneErr[l]) std::rethrow_exception(laneErr[l]);
	}
	// the sub-ranges in query order: offsets over the whole call, where each query's primaries lie
	for (auto& srp : subs)
	{
		SubResult& sr = *srp;
		fg_ctx* lc = sr.lane ? c->lane2.get() : c;
		const PrimRec* base = (const PrimRec*)lc->hPrim.p + sr.lanePrimBase;
		for (u32 i = 0; i < sr.qb - sr.qa; ++i)
		{
			primOffAll[sr.qa + i + 1] = nPrim + sr.primOff[i + 1];
			primPtr[sr.qa + i] = base - nPrim;
		}
		if (keepAln)
			for (u64 j = 0; j < sr.cr.nPrim; ++j)
			{
				mData.insert(mData.end(), sr.mData.begin() + sr.mOff[j], sr.mData.begin() + sr.mOff[j + 1]);
				mOff.push_back(mData.size());
			}
		out->seed_hits += sr.cr.nHits; out->dp_groups += sr.cr.dpGroups; out->dp_elements += sr.cr.dpElems;
		out->dp_elements_small += sr.cr.dpElemsSmall;
		nPrim += sr.cr.nPrim;
	}
	HIP_CHECK(hipEventRecord(evB, s));
	// no wait for the last chunk's records here: piece by piece in the shim's first pass (one lane; with two, both
	// lanes' copies have been waited for above)
	const u64* hPrimOff = primOffAll.data();
	const auto tHost1 = std::chrono::steady_clock::now();
	const int dev = c->device;
	std::atomic<int> waitErr{0};		// worker threads must not throw
	const bool pieceWait = !laned && subs.size() <= 1;
	auto waitPrim = [c, dev, &waitErr, pieceWait](u64 endIdx)
	{
		if (!pieceWait) return;
		// primaries below endIdx are on the host when the first piece reaching that far has landed (copies of one
		// stream complete in order; earlier chunks' pieces lie below this chunk's first)
		(void)hipSetDevice(dev);
		for (int i = 0; i < FG_D2H_PIECES; ++i)
			if (c->pieceEnd[i] >= endIdx || i == FG_D2H_PIECES - 1)
			{
				const hipError_t e = hipEventSynchronize(c->evPiece[i]);
				if (e != hipSuccess) waitErr.store((int)e);
				return;
			}
	};

	// ---- host shim: floats with the host libm, the gate, prefix rule, window stats ----
	// two passes over the queries, both fanned out over host threads: (1) divergence,
	// gate and per-query counts, (2) after a prefix sum, the records themselves
	const float sampleRate = c->sampleRate;
	const float maxDiv = p->max_divergence;
	const bool nucl = p->nucl_alignment;
	const bool partition = p->partition_bad_mappings;	// only with maxOverlaps == 0 (fg_api.hip)
	const int STAT_WND = 10000;
	// result-sized scratch lives in the context (grow-only, never zero-filled: pass 1 writes every element)
	if (c->shimDiv.size() < nPrim) { c->shimDiv.resize(nPrim + nPrim / 8); c->shimKeep.resize(c->shimDiv.size()); }
	if (c->shimNStat.size() < nq) { c->shimNStat.resize(nq + nq / 8); c->shimNMatch.resize(c->shimNStat.size()); }
	float* div = c->shimDiv.data();
	uint8_t* keep = c->shimKeep.data();
	u32* nStat = c->shimNStat.data();
	u64* nMatch = c->shimNMatch.data();
	std::vector<std::vector<float>> statVals;
	// host threads of the shim: the hardware threads, capped by the cgroup CPU quota (a container that
	// shows 256 threads but is granted 16 CPUs of time runs the shim slower on 32 threads than on 16)
	const unsigned usableCpus = fg_usable_cpus();
	unsigned nThreads = std::max(1u, std::min(usableCpus, 32u));
	if (getenv("FG_SHIM_THREADS")) nThreads = std::max(1, atoi(getenv("FG_SHIM_THREADS")));
	if (nPrim < 20000) nThreads = 1;
	statVals.resize(nThreads);
	std::vector<std::vector<u32>> statQ(nThreads);
	auto pass1 = [&](unsigned t)
	{
		struct Wnd { i32 range; float div; };
		std::vector<Wnd> wnd;
		const u32 q0 = (u32)((u64)nq * t / nThreads), q1 = (u32)((u64)nq * (t + 1) / nThreads);
		if (q1 > q0) waitPrim(hPrimOff[q1]);
		if (waitErr.load()) return;
		for (u32 qi = q0; qi < q1; ++qi)
		{
			const i32 curLen = (c->hasQ ? c->hQLen : c->hLen)[hq[qi] >> 1];
			wnd.assign(curLen / STAT_WND + 1, Wnd{0, 0.0f});
			size_t detected = 0;
			u32 prevExt = 0xFFFFFFFFu;
			if (keepAln) nMatch[qi] = 0;
			const PrimRec* hPrim = primPtr[qi];
			for (u64 j = hPrimOff[qi]; j < hPrimOff[qi + 1]; ++j)
			{
				const PrimRec& r = hPrim[j];
				// groups are visited in ascending extId; the limit is tested only at a group
				// start, against the overlaps accepted so far (overlap.cpp:218-219) -- a group
				// with several primaries (onlyMaxExt = false) is never cut in the middle
				if (r.extId != prevExt)
				{
					if (maxOverlaps != 0 && detected >= (size_t)maxOverlaps)
					{
						for (u64 jj = j; jj < hPrimOff[qi + 1]; ++jj) keep[jj] = 0;	// the scratch is not zero-filled
						break;
					}
					prevExt = r.extId;
				}
				// overlap.cpp:414-423
				float normLen = std::max(r.curEnd - r.curBegin, r.extEnd - r.extBegin) - r.filtered;
				float matchRate = (float)r.chainLength * sampleRate / normLen;
				matchRate = std::min(matchRate, 1.0f);
				float d = std::log(1 / matchRate) / k;
				if (nucl)	// alignment.cpp:244-245
					d = (float)r.editDistance / std::max((size_t)r.hpcLenExt, (size_t)r.hpcLenCur);
				div[j] = d;
				if (d < maxDiv) { keep[j] = 1; ++detected; }
				else if (partition) { keep[j] = 2; ++detected; }	// handed back for the caller's checkIdyAndTrim
				else keep[j] = 0;
				if (keep[j] && keepAln) nMatch[qi] += mOff[j + 1] - mOff[j];
				const size_t w = r.curBegin / STAT_WND;
				if (r.curEnd - r.curBegin > wnd[w].range) { wnd[w].range = r.curEnd - r.curBegin; wnd[w].div = d; }
			}
			own->queryOff[qi + 1] = detected;
			u32 ns = 0;
			for (auto& w : wnd) if (w.range > 0) { statVals[t].push_back(w.div); ++ns; }
			nStat[qi] = ns;
		}
	};
	auto runThreads = [&](const std::function<void(unsigned)>& fn) { c->shimPool.run(nThreads, fn); };
	if (!pieceWait) HIP_CHECK(hipStreamSynchronize(s));		// several sub-ranges: their copies are simply waited for
	runThreads(pass1);
	HIP_CHECK(hipStreamSynchronize(s));		// everything has landed by now; also surfaces a failed copy
	if (waitErr.load()) throw FgError{FG_ERR_HIP, std::string("waiting for the result copy: ") + hipGetErrorString((hipError_t)waitErr.load())};
	own->queryOff[0] = 0;
	for (u32 qi = 0; qi < nq; ++qi)
	{
		own->queryOff[qi + 1] += own->queryOff[qi];
		own->statOff[qi + 1] = own->statOff[qi] + nStat[qi];
	}
	own->reserveRecs(own->queryOff[nq]);
	own->nRecs = own->queryOff[nq];
	if (partition) own->needsTrim.assign(own->nRecs, 0);
	std::vector<u64> qMatchOff;		// first pair of each query's records
	if (keepAln)
	{
		qMatchOff.assign(nq + 1, 0);
		for (u32 qi = 0; qi < nq; ++qi) qMatchOff[qi + 1] = qMatchOff[qi] + nMatch[qi];
		own->reserveMatches(qMatchOff[nq]);
		own->matchOff.assign(own->nRecs + 1, 0);
		own->matchOff[own->nRecs] = qMatchOff[nq];
	}
	own->stats.clear();
	own->stats.reserve(own->statOff[nq]);
	for (unsigned t = 0; t < nThreads; ++t) own->stats.insert(own->stats.end(), statVals[t].begin(), statVals[t].end());
	auto pass2 = [&](unsigned t)
	{
		const u32 q0 = (u32)((u64)nq * t / nThreads), q1 = (u32)((u64)nq * (t + 1) / nThreads);
		for (u32 qi = q0; qi < q1; ++qi)
		{
			const i32 curLen = (c->hasQ ? c->hQLen : c->hLen)[hq[qi] >> 1];
			fg_overlap_rec* dst = own->recs + own->queryOff[qi];
			u64 mo = keepAln ? qMatchOff[qi] : 0;
			const PrimRec* hPrim = primPtr[qi];
			for (u64 j = hPrimOff[qi]; j < hPrimOff[qi + 1]; ++j)
			{
				if (!keep[j]) continue;
				const PrimRec& r = hPrim[j];
				if (partition) own->needsTrim[dst - own->recs] = keep[j] == 2;
				if (keepAln)
				{
					const u64 cnt = mOff[j + 1] - mOff[j];
					own->matchOff[dst - own->recs] = mo;
					memcpy(own->matches + 2 * mo, mData.data() + mOff[j], cnt * 8);	// (cur, ext) int32 pairs
					mo += cnt;
				}
				fg_overlap_rec o;
				o.cur_id = queryIds[qi]; o.ext_id = r.extId;
				o.cur_begin = r.curBegin; o.cur_end = r.curEnd; o.cur_len = curLen;
				o.ext_begin = r.extBegin; o.ext_end = r.extEnd; o.ext_len = r.extLen;
				o.score = r.score; o.seq_divergence = div[j];
				o.chain_length = r.chainLength; o.filtered_positions = r.filtered;
				o.edit_distance = r.editDistance; o.hpc_len_cur = r.hpcLenCur; o.hpc_len_ext = r.hpcLenExt;
				// the records are written once and not read again by this thread: streaming stores keep the
				// read-for-ownership of every destination line off the memory bus
				static_assert(sizeof(fg_overlap_rec) % 4 == 0, "record of 32-bit fields");
				const int* src32 = (const int*)&o;
				int* dst32 = (int*)dst;
#pragma unroll
				for (unsigned w4 = 0; w4 < sizeof(fg_overlap_rec) / 4; ++w4) __builtin_nontemporal_store(src32[w4], dst32 + w4);
				++dst;
			}
		}
		__builtin_ia32_sfence();
	};
	runThreads(pass2);
	out->n_recs = own->nRecs;
	out->query_off = own->queryOff.data();
	out->recs = own->recs;
	out->n_div_stats = own->stats.size();
	out->div_stats_off = own->statOff.data();
	out->div_stats = own->stats.data();
	if (partition) out->needs_trim = own->needsTrim.data();
	if (keepAln)
	{
		out->n_matches = own->matchOff[own->nRecs];
		out->match_off = own->matchOff.data();
		out->matches = own->matches;
	}
	float ms = 0;
	HIP_CHECK(hipEventElapsedTime(&ms, evA, evB));
	out->device_seconds = laned ? std::chrono::duration<double>(tHost1 - tHost0).count() : ms * 1e-3;
	c->timer.collect();
	if (laned && c->lane2)
	{
		// the second lane's kernel times, name by name
		c->lane2->timer.collect();
		for (const auto& kt : c->lane2->timer.last)
		{
			bool found = false;
			for (auto& mine : c->timer.last)
				if (!strcmp(mine.name, kt.name)) { mine.seconds += kt.seconds; mine.launches += kt.launches; found = true; break; }
			if (!found) c->timer.last.push_back(kt);
		}
	}
	const auto tHost2 = std::chrono::steady_clock::now();
	c->timer.last.push_back(fg_kernel_time{"host:launch+sync (wall, includes the device time)",
							std::chrono::duration<double>(tHost1 - tHost0).count(), 1});
	c->timer.last.push_back(fg_kernel_time{"host:shim (divergence, gate, records)",
							std::chrono::duration<double>(tHost2 - tHost1).count(), 1});
}
