// Chaining stage of getSeqOverlaps for every target group of a batch
// (reference src/sequence/overlap.cpp:216-458, overlapTest :29-69), in three kernels:
//
//   k_group_prep    one wave per group: distinct-position count, span and overhang
//                   prefilter (:220-262); survivors get their (cur, ext) columns written in
//                   DP order -- as sorted, or re-sorted by extPos in std::sort order when
//                   extLen > curLen (:268-275)
//   k_chain_dp      the chaining DP (:277-323), one wave per group, no LDS: lane L keeps
//                   element i-1-L in registers (window shifted by one lane per step), the
//                   reference's look-back scan ("first strictly better j going down, two
//                   early exits") becomes a wave-wide exclusive prefix max (DPP) + a ballot
//                   of the exits; look-backs deeper than 64 continue from memory
//   k_chain_finish  one wave per group: chain starts in descending-score std::sort order
//                   (:331-334), backtracking with consumption (:338-383), overlapTest,
//                   primary selection (:431-439, onlyMaxExt)
#include "fg_wavesort.h"

#include <algorithm>

#define WG 256
#define I32_MIN ((i32)0x80000000)

struct ChainParams {
	int k, maxJump, minOverlap, maxOverhang;
	int checkOverhang, forceLocal;
	float minUnique;	// minKmerSruvivalRate * _minOverlap as a float (overlap.cpp:110, :235)
	u32 firstId;		// FastaRecord id of the first indexed record
	u32 qFirstId;		// ... of the first query record (= firstId unless a query container is set)
	int onlyMaxExt;
	int keepAln;		// the DP's back pointers must survive the backtracking (k_chain_matches re-walks them)
	int ablate;			// timing experiments only (FG_ABLATE env; results become wrong)
};

namespace {

struct CandAcc {	// candidates by descending score (overlap.cpp:432-434); w = score
	typedef int4 T;
	int4* c;
	__device__ int4 load(int i) const { return c[i]; }
	__device__ void store(int i, const int4& x) { c[i] = x; }
	__device__ bool less(const int4& a, const int4& b) const { return a.w > b.w; }
};

// overlap.cpp:29-69; the float comparison is evaluated in float exactly as written there
__device__ __forceinline__ bool overlap_test(const ChainParams& P, u32 curId, u32 extId, i32 curLen,
											 i32 extLen, i32 cb, i32 ce, i32 eb, i32 ee)
{
	const i32 curRange = ce - cb, extRange = ee - eb;
	if (curRange < P.minOverlap || extRange < P.minOverlap) return false;
	const float lengthDiff = (float)abs(curRange - extRange);
	if (lengthDiff > 0.5f * (float)min(curRange, extRange)) return false;
	if (curId == extId)
	{
		const i32 inter = min(ce, ee) - max(cb, eb);
		if (inter > curRange / 2) return false;
	}
	if (curId == (extId ^ 1u))
	{
		const i32 inter = min(ce, extLen - eb) - max(cb, extLen - ee);
		if (inter > curRange / 2) return false;
	}
	if (!P.forceLocal && P.checkOverhang)
	{
		const i32 ovh = max(min(cb, eb), min(curLen - ce, extLen - ee));
		if (ovh > P.maxOverhang) return false;
	}
	return true;
}

// ---- lists -------------------------------------------------------------------------------
// block-aggregated append of the calling thread's LIST_ITEMS consecutive groups to up to two
// lists: one atomic per list and block of LIST_ITEMS * 256 groups (same-address atomics
// serialise; the list kernels were bound by exactly these atomics at one per 256 groups)
#define LIST_ITEMS 8
__device__ __forceinline__ void append2_multi(const bool (&a)[LIST_ITEMS], const bool (&b)[LIST_ITEMS], u32 g0,
											  u32* listA, u32* listB, u32* counts)
{
	__shared__ u32 wtot[2][WG / 64];
	__shared__ u32 base[2];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	u32 cA = 0, cB = 0;
#pragma unroll
	for (int t = 0; t < LIST_ITEMS; ++t) { cA += a[t]; cB += b[t]; }
	u32 iA = cA, iB = cB;	// inclusive scan over the wave
	for (int o = 1; o < 64; o <<= 1)
	{
		const u32 x = __shfl_up(iA, o), y = __shfl_up(iB, o);
		if (lane >= o) { iA += x; iB += y; }
	}
	if (lane == 63) { wtot[0][wv] = iA; wtot[1][wv] = iB; }
	__syncthreads();
	if (threadIdx.x < 2)
	{
		u32 t = 0;
		for (int i = 0; i < WG / 64; ++i) t += wtot[threadIdx.x][i];
		base[threadIdx.x] = t ? atomicAdd(&counts[threadIdx.x], t) : 0u;
	}
	__syncthreads();
	u32 oA = base[0] + iA - cA, oB = base[1] + iB - cB;
	for (int i = 0; i < wv; ++i) { oA += wtot[0][i]; oB += wtot[1][i]; }
#pragma unroll
	for (int t = 0; t < LIST_ITEMS; ++t)
	{
		if (a[t]) listA[oA++] = g0 + (u32)t;
		if (b[t]) listB[oB++] = g0 + (u32)t;
	}
}

// groups that can still have >= minUnique distinct query positions (unique <= size) and whose
// query span reaches minOverlap (the first and last hit of the sorted group give it: overlap.cpp:
// 244-249 would drop the group anyway, here it never costs k_group_prep a wave)
__global__ void k_group_list(u64 nGroups, u64 nHits, const u64* __restrict__ groupStart, u32 minSize,
							 const u32* __restrict__ groupFirstCur, const u32* __restrict__ groupLastCur, i32 minOverlap,
							 u32* __restrict__ list, u32* __restrict__ listBig, u32 bigMin,
							 u32* __restrict__ listFused, u32 fusedMax /* 0: no fused class */, u32* __restrict__ counts,
							 u32* __restrict__ primCount, u32* __restrict__ dpSize)
{
	const u64 g0 = ((u64)blockIdx.x * WG + threadIdx.x) * LIST_ITEMS;
	bool a[LIST_ITEMS], b[LIST_ITEMS], f[LIST_ITEMS], none[LIST_ITEMS];
	u64 start = g0 < nGroups ? groupStart[g0] : nHits;
#pragma unroll
	for (int t = 0; t < LIST_ITEMS; ++t)
	{
		const u64 g = g0 + t;
		u64 n = 0;
		bool ok = false;
		if (g < nGroups)
		{
			primCount[g] = 0;
			dpSize[g] = 0;
			const u64 gend = (g + 1 < nGroups) ? groupStart[g + 1] : nHits;
			n = gend - start;
			if (n >= minSize && n > 0)
				ok = (i32)groupLastCur[g] - (i32)groupFirstCur[g] >= minOverlap;
			start = gend;
		}
		f[t] = ok && n <= fusedMax;				// the whole chaining stage in one kernel (k_chain_small)
		a[t] = ok && n > fusedMax && n <= bigMin;
		b[t] = ok && n > bigMin;	// groups that k_group_prep handles in global memory
		none[t] = false;
	}
	append2_multi(a, b, (u32)g0, list, listBig, counts);
	if (fusedMax)
	{
		__syncthreads();	// the helper's shared counters are reused
		append2_multi(f, none, (u32)g0, listFused, listFused, counts + 2);
	}
}

#ifndef FIN_CAP_S
#define FIN_CAP_S 256
#endif		// <= : LDS, 4 groups per block; larger groups run on global scratch.
#ifndef FIN_CAP_M
#define FIN_CAP_M 1024	// LDS class of small calls (see fgChainStage)
#endif
#ifndef FIN_WAVES_S
#define FIN_WAVES_S 1	// waves (= groups) per block of the <= 256-hit class / of the global-scratch class: one,
					// so that a block's slot frees when ITS group is done (4 -> 1: 4.97 -> 4.46 and 5.83 -> 5.37 ms)
#endif
#ifndef FIN_WAVES_G
#define FIN_WAVES_G 1
#endif
#ifndef FIN_BT_CAP
#define FIN_BT_CAP 1024	// global-scratch groups up to this size walk their back pointers in LDS
#endif
// (Measured at the bench workload: BT cap 512 / 768 / 1024 / 1536 / 2048 -> k_chain_finish<global>
// 7.0 / 6.2 / 6.1 / 6.4 / 8.0 ms against 7.8 ms with the walk in global memory; staging the SORT
// of 257..1024-hit groups in LDS as well costs more occupancy than it saves latency.)
// groups that passed the prefilter, by size class
__global__ void k_dp_list(u64 nGroups, const u32* __restrict__ dpSize, u32 hugeMin, u32 doneMax /* groups up to this size are finished already */,
						  u32* __restrict__ listSmall, u32* __restrict__ listMid, u32* __restrict__ listBig, u32* __restrict__ counts)
{
	const u64 g0 = ((u64)blockIdx.x * WG + threadIdx.x) * LIST_ITEMS;
	bool a[LIST_ITEMS], b[LIST_ITEMS], h[LIST_ITEMS], none[LIST_ITEMS];
#pragma unroll
	for (int t = 0; t < LIST_ITEMS; ++t)
	{
		const u32 n = g0 + t < nGroups ? dpSize[g0 + t] : 0u;
		a[t] = n > doneMax && n <= FIN_CAP_S;
		b[t] = n > FIN_CAP_S && n > doneMax && n <= hugeMin;
		h[t] = n > hugeMin;
		none[t] = false;
	}
	append2_multi(a, b, (u32)g0, listSmall, listMid, counts);
	__syncthreads();	// the helper's shared counters are reused
	append2_multi(h, none, (u32)g0, listBig, listBig, counts + 2);
}

// ---- prep --------------------------------------------------------------------------------
#ifndef PREP_CAP
#define PREP_CAP 320	// 128 / 192 / 256 / 320 / 512 measured: 5.2 / 4.95 / 4.8 / 4.7 / 5.15 ms
#endif
#ifndef PREP_WAVES
#define PREP_WAVES 1
#endif
template <class KT>
__global__ void __launch_bounds__(PREP_WAVES * 64)
k_group_prep(ChainParams P, const u32* __restrict__ list, u32 nList, u64 nGroups, u64 nHits,
			 const u64* __restrict__ groupStart, const u32* __restrict__ groupQuery,
			 const u32* __restrict__ query, const i32* __restrict__ len, const i32* __restrict__ qLen,
			 HitKeyView<KT> hitKey, const u32* __restrict__ groupExt,
			 u32* __restrict__ gCur, u32* __restrict__ gExt, u32* __restrict__ gAux /* 4 u32 per hit */,
			 u32* __restrict__ dpSize, uint8_t* __restrict__ groupExtSorted)
{
	__shared__ u32 sExt[PREP_WAVES][PREP_CAP];
	__shared__ u32 sCur[PREP_WAVES][PREP_CAP];
	__shared__ unsigned short sPL[PREP_WAVES][PREP_CAP], sPR[PREP_WAVES][PREP_CAP];
	__shared__ int stack[PREP_WAVES][3 * 40];
	__shared__ int small[PREP_WAVES][3 * 8];
	const int wv = threadIdx.x >> 6;
	const int lane = threadIdx.x & 63;
	const u32 li = blockIdx.x * PREP_WAVES + wv;
	if (li >= nList) return;
	const u64 g = fg_uni(list[li]);
	const u64 g0 = fg_uni(groupStart[g]);
	const u64 gend = (g + 1 < nGroups) ? fg_uni(groupStart[g + 1]) : nHits;
	const i32 n = (i32)(gend - g0);

	// the lookups that only the survivors need are issued first, so that their round trips overlap
	// the pass over the hits instead of following it
	const u32 qrec = query[groupQuery[g]];
	const u32 extRec = groupExt[g] - P.firstId;
	const i32 curLen = qLen[qrec >> 1];
	const i32 extLen = len[extRec >> 1];
	const i32 minCur = (i32)hitKey.cur(g0), maxCur = (i32)hitKey.cur(g0 + n - 1);

	// distinct query positions (overlap.cpp:220-235; prevPos starts at 0) and ext span; groups that
	// fit the LDS piece are staged on the way
	const bool inLds = n <= PREP_CAP;
	u32 uniq = 0;
	i32 minExt = 0x7fffffff, maxExt = I32_MIN;
	for (i32 i = lane; i < n; i += 64)
	{
		const u32 c = hitKey.cur(g0 + i);
		const u32 pc = i ? hitKey.cur(g0 + i - 1) : 0u;
		uniq += (c != pc);
		const i32 e = (i32)hitKey.val(g0 + i);
		minExt = min(minExt, e); maxExt = max(maxExt, e);
		if (inLds) { sCur[wv][i] = c; sExt[wv][i] = (u32)e; }
	}
	for (int o = 32; o > 0; o >>= 1)
	{
		uniq += __shfl_xor(uniq, o);
		minExt = min(minExt, __shfl_xor(minExt, o));
		maxExt = max(maxExt, __shfl_xor(maxExt, o));
	}
	if ((float)uniq < P.minUnique) return;
	if (maxCur - minCur < P.minOverlap || maxExt - minExt < P.minOverlap) return;
	if (P.checkOverhang && !P.forceLocal)
	{
		if (min(minCur, minExt) > P.maxOverhang) return;
		if (min(curLen - maxCur, extLen - maxExt) > P.maxOverhang) return;
	}
	if (lane == 0) { dpSize[g] = (u32)n; groupExtSorted[g] = extLen > curLen ? 1 : 0; }

	u32* oc = gCur + g0;
	u32* oe = gExt + g0;
	const bool extSorted = extLen > curLen;
	// A group whose target positions already ascend strictly needs no re-sort: std::sort leaves a strictly ascending
	// sequence as it is, whatever its trajectory (half of the re-sorted groups of the bench workload, a third of
	// their hits: collinear seeds of a true overlap with no stray hit in between).
	if (inLds)
	{
		wsort::wave_mem_fence();
		if (extSorted)
		{
			bool asc = true;
			for (i32 i = lane + 1; i < n; i += 64) asc = asc && sExt[wv][i] > sExt[wv][i - 1];
			if (__builtin_amdgcn_ballot_w64(!asc))
				wsort::wave_sort<u32, unsigned short>(sExt[wv], sCur[wv], n, sPL[wv], sPR[wv], stack[wv], small[wv]);
		}
		for (i32 i = lane; i < n; i += 64) { oc[i] = sCur[wv][i]; oe[i] = sExt[wv][i]; }
	}
	else if (!extSorted)
	{
		for (i32 i = lane; i < n; i += 64) { oc[i] = hitKey.cur(g0 + i); oe[i] = hitKey.val(g0 + i); }
	}
	else
	{
		bool asc = true;
		for (i32 i = lane; i < n; i += 64)
		{
			const u32 e = hitKey.val(g0 + i);
			oc[i] = hitKey.cur(g0 + i); oe[i] = e;
			if (i > 0) asc = asc && e > hitKey.val(g0 + i - 1);
		}
		wsort::wave_mem_fence();
		if (__builtin_amdgcn_ballot_w64(!asc))
		{
			u32* aux = gAux + 4 * g0;
			wsort::wave_sort<u32, u32>(oe, oc, n, aux, aux + n, stack[wv], small[wv]);
		}
	}
}

// ---- DP: one wave per group, no LDS ---------------------------------------------------------
// lane L <- lane L-1, lane 0 <- fill (DPP wave_shr:1)
__device__ __forceinline__ i32 wave_shr1(i32 v, i32 fill)
{
	return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
}
// inclusive prefix max over the 64 lanes with DPP row shifts + row broadcasts
__device__ __forceinline__ i32 wave_incl_max(i32 v)
{
	v = max(v, __builtin_amdgcn_update_dpp(I32_MIN, v, 0x111, 0xf, 0xf, false));	// row_shr:1
	v = max(v, __builtin_amdgcn_update_dpp(I32_MIN, v, 0x112, 0xf, 0xf, false));	// row_shr:2
	v = max(v, __builtin_amdgcn_update_dpp(I32_MIN, v, 0x114, 0xf, 0xf, false));	// row_shr:4
	v = max(v, __builtin_amdgcn_update_dpp(I32_MIN, v, 0x118, 0xf, 0xf, false));	// row_shr:8
	v = max(v, __builtin_amdgcn_update_dpp(I32_MIN, v, 0x142, 0xa, 0xf, false));	// row_bcast:15
	v = max(v, __builtin_amdgcn_update_dpp(I32_MIN, v, 0x143, 0xc, 0xf, false));	// row_bcast:31
	return v;
}

// The reference's look-back loop (overlap.cpp:285-316: scan j downwards, keep the first
// strictly better score, two early exits) runs 64 candidates per step, and only for the
// elements that need it:
//  * tile = 64 consecutive elements, element 64 T + L on lane L, in registers (cur, ext, score,
//    back); the previous tile stays in registers too, older ones in a 256-element LDS ring,
//    anything older in memory;
//  * an element whose predecessor i-1 lies on the same diagonal less than k ahead takes that
//    predecessor at once (first candidate of the scan, improves on 0, triggers the early exit,
//    overlap.cpp:301-307): score[i] = score[i-1] + (cur[i] - cur[i-1]), back[i] = i-1.  Such
//    elements form RUNS behind a head element; they are found for the whole tile in parallel
//    and get their scores by one vector add when their head is known -- 64 % of the elements
//    at the bench workload never enter the serial loop;
//  * a head at lane il scans elements i-1 .. i-64 = lanes il-1 .. 0 of its own tile, then lanes
//    63 .. il of the previous one: candidate scores are computed in place, rotated into scan
//    order through the LDS crossbar (one ds_bpermute), an exclusive prefix max (DPP row shifts
//    + row broadcasts) tells every candidate whether it would have improved the running best,
//    ballots of the exit conditions cut the scan.  The "beyond maxJump" exit is monotone along
//    the scan (the group is sorted by the coordinate it tests), so its position is a popcount.
//    Look-backs deeper than 64 take 64 more candidates per step from the ring / memory.
#ifndef DP_WAVES
#define DP_WAVES 1	// one group per block: a block's slot frees as soon as ITS group is done
#endif
#ifndef DP_RING
#define DP_RING 256
#endif

// one 64-candidate step of the scan.  ns: candidate scores in SCAN order (lane 0 = first scanned),
// I32_MIN where the candidate is out of range; mA: scan-order mask of "same diagonal, closer than
// k" candidates; rhoB: scan position of the first candidate beyond the window (64 = none).
// Updates (maxScore, maxId); returns true when the scan ends inside this step.
__device__ __forceinline__ bool dp_scan_step(i32 ns, u64 mA, int rhoB, i32 firstJ, i32& maxScore, i32& maxId)
{
	// exclusive prefix: the shift fills lane 0 with 0, which the max with maxScore (>= 0) absorbs
	const i32 exc = max(wave_incl_max(__builtin_amdgcn_update_dpp(0, ns, 0x138, 0xf, 0xf, true)), maxScore);
	const u64 updM = __builtin_amdgcn_ballot_w64(ns > exc);
	const u64 stopM = (updM & mA) | (rhoB < 64 ? (1ULL << rhoB) : 0ULL);
	const u64 lim = stopM ? (((stopM & (0 - stopM)) << 1) - 1) : ~0ULL;
	const u64 um = updM & lim;
	if (um)
	{
		const int lu = 63 - __clzll(um);
		maxScore = __builtin_amdgcn_readlane(ns, __builtin_amdgcn_readfirstlane(lu));
		maxId = firstJ - lu;
	}
	return stopM != 0;
}

// |a - b| of two values read as unsigned (one instruction; the operands are in-range distances
// whenever the result is used)
__device__ __forceinline__ u32 abs_diff(i32 a, i32 b)
{
	u32 r;
	asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

// WHOLE: the whole group (<= DP_RING hits) sits in LDS arrays cur / ext / score: deep look-backs read those, no ring
template <bool EXTS, bool WHOLE = false>
__device__ __forceinline__ void dp_group(const int k, const i32 maxJump, const i32 n, const u32* __restrict__ cur,
										 const u32* __restrict__ ext, i32* score, i32* __restrict__ back,
										 i32* ringC, i32* ringE, i32* ringS)
{
	const int lane = threadIdx.x & 63;
	const i32 negLane4 = (-1 - lane) * 4;
	// "no such element" (the tile before the first, positions before the group's start): coordinates so far
	// away that the candidate is out of range and beyond the window by the ordinary tests
	const i32 FAR = -(1 << 30);
	i32 pc = FAR, pe = FAR, ps = 0;		// previous tile
	// Upper bound for the deep look-backs: no candidate j scores more than score[j] + k (matchScore <= k, gapCost >= 0),
	// so once (the largest score among ALL elements up to j) + k <= the running best, nothing further back can replace
	// it -- the reference's scan would walk on through the window (HiFi data: ~270 candidates per head, 4-5 steps of
	// 64) without ever updating.  Lane b holds the largest score of tiles 0 .. b (first 64 tiles; beyond: no bound).
	i32 pmv = 0x7fffffff, pmRun = I32_MIN;
	i32 ntc = lane < n ? (i32)cur[lane] : 0, nte = lane < n ? (i32)ext[lane] : 0;
	asm volatile("" : "+v"(ntc), "+v"(nte));	// waited for here, so that no wait for them is left inside the loop (see its end)
	for (i32 tb0 = 0; tb0 < n; tb0 += 64)
	{
		const i32 tc = ntc, te = nte;
		ntc = tb0 + 64 + lane < n ? (i32)cur[tb0 + 64 + lane] : 0;
		nte = tb0 + 64 + lane < n ? (i32)ext[tb0 + 64 + lane] : 0;
		const bool valid = tb0 + lane < n;
		// last element of the previous tile (meaningless for the first tile: elements 0 and 1 are never fast)
		const i32 pcl = __builtin_amdgcn_readlane(pc, 63), pel = __builtin_amdgcn_readlane(pe, 63),
				  psl = __builtin_amdgcn_readlane(ps, 63);
		const i32 dc0 = tc - wave_shr1(tc, pcl), de0 = te - wave_shr1(te, pel);
		const bool fast = valid && dc0 == de0 && (u32)(dc0 - 1) < (u32)(min(k, maxJump) - 1) && tb0 + lane >= 2;
		u64 headM = __builtin_amdgcn_ballot_w64(valid && !fast);
		// nearest head at or below every lane (-1: the run started in an earlier tile)
		const i32 hl = wave_incl_max(fast ? -1 : lane);
		const i32 chead = __builtin_amdgcn_ds_bpermute(max(hl, 0) << 2, tc);
		const i32 off = tc - (hl >= 0 ? chead : pcl);
		i32 ts = psl + off;				// right for the leading run; the others are set when their head is done
		i32 tbk = tb0 + lane - 1;		// right for every run element
		if (tb0 == 0)
		{
			// element 0: score 0, no predecessor, never scanned for (overlap.cpp:266-267, :277)
			ts = lane == 0 ? 0 : ts;
			tbk = lane == 0 ? -1 : tbk;
			headM &= ~1ULL;
		}
		while (headM)
		{
			const int il = __builtin_amdgcn_readfirstlane(__ffsll((long long)headM) - 1);
			headM &= headM - 1;
			const i32 i = tb0 + il;
			const i32 cn = __builtin_amdgcn_readlane(tc, il), en = __builtin_amdgcn_readlane(te, il);
			i32 maxScore = 0, maxId = 0;
			bool done;
			{
				// candidates i-1 .. i-64 in place: lanes below il hold this tile's elements, the others the previous tile's
				const bool own = lane < il;
				const i32 cp = own ? tc : pc, ep = own ? te : pe, sj = own ? ts : ps;
				const i32 dc = cn - cp, de = en - ep;
				const bool inr = max((u32)(dc - 1), (u32)(de - 1)) < (u32)(maxJump - 1);
				const i32 jd = (i32)abs_diff(dc, de);
				i32 nsRaw = sj + min(min(dc, de), k) - (jd > 100 ? 2 * jd : (jd >> 1));
				asm volatile("" : "+v"(nsRaw));	// keep the arithmetic out of a conditional block
				const i32 ns = inr ? nsRaw : I32_MIN;
				// every mask is one bare compare; they are combined as scalars
				const u64 inrM = __builtin_amdgcn_ballot_w64(inr);
				const u64 mB = __builtin_amdgcn_ballot_w64((EXTS ? de : dc) > maxJump);
				const u64 mAn = __builtin_amdgcn_ballot_w64(dc == de) & __builtin_amdgcn_ballot_w64(dc < k) & inrM;
				// scan order: position r <- lane (il - 1 - r) mod 64
				const i32 nsR = __builtin_amdgcn_ds_bpermute((il * 4 + negLane4) & 0xfc, ns);
				u64 mA = 0;
				if (mAn)
				{
					const u64 R = __brevll(mAn);
					const int sh = (64 - il) & 63;
					mA = sh ? (R >> sh) | (R << (64 - sh)) : R;
				}
				done = dp_scan_step(nsR, mA, __popcll(~mB), i - 1, maxScore, maxId);
			}
			for (i32 jb = i - 65; jb >= 0 && !done; jb -= 64)
			{
				{
					const int blk = jb >> 6;		// every candidate of this and the following steps lies in tiles 0 .. blk
					if (blk < 64 && __builtin_amdgcn_readlane(pmv, blk) <= maxScore - k) break;
				}
				// deeper: lane = scan position; the previous DP_RING elements of finished tiles from LDS,
				// anything older from memory (stored by this wave)
				const i32 j = jb - lane;
				i32 cp = FAR, ep = FAR, sj = 0;
				if (j >= 0)
				{
					if (WHOLE) { cp = (i32)cur[j]; ep = (i32)ext[j]; sj = score[j]; }
					else if (j >= tb0 - DP_RING)
					{
						cp = ringC[j & (DP_RING - 1)]; ep = ringE[j & (DP_RING - 1)]; sj = ringS[j & (DP_RING - 1)];
					}
					else
					{
						__builtin_amdgcn_s_waitcnt(0);
						cp = (i32)cur[j]; ep = (i32)ext[j];
						sj = __hip_atomic_load(&score[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
				}
				const i32 dc = cn - cp, de = en - ep;
				const bool inr = max((u32)(dc - 1), (u32)(de - 1)) < (u32)(maxJump - 1);
				const i32 jd = (i32)abs_diff(dc, de);
				i32 nsRaw = sj + min(min(dc, de), k) - (jd > 100 ? 2 * jd : (jd >> 1));
				asm volatile("" : "+v"(nsRaw));
				const i32 ns = inr ? nsRaw : I32_MIN;
				const u64 inrM = __builtin_amdgcn_ballot_w64(inr);
				const u64 mB = __builtin_amdgcn_ballot_w64((EXTS ? de : dc) > maxJump);
				const u64 mA = __builtin_amdgcn_ballot_w64(dc == de) & __builtin_amdgcn_ballot_w64(dc < k) & inrM;
				done = dp_scan_step(ns, mA, __popcll(~mB), jb, maxScore, maxId);
			}
			const i32 sNew = max(maxScore, k);
			ts = hl == il ? sNew + off : ts;	// the head (off = 0) and its run inside this tile
			tbk = lane == il ? (maxScore > k ? maxId : -1) : tbk;
		}
		// The next tile's coordinates (loaded at the top of this iteration) are taken in HERE, before this tile's
		// stores go out: the counter of outstanding memory operations retires in order, so a wait placed behind the
		// stores (or behind the next prefetch, where the compiler puts it by itself and then waits for everything
		// because the loads sit in branches) would cost a full round trip per tile.
		asm volatile("" : "+v"(ntc), "+v"(nte));
		if (valid)
		{
			score[tb0 + lane] = ts; back[tb0 + lane] = tbk;
			if (!WHOLE) { ringC[(tb0 + lane) & (DP_RING - 1)] = tc; ringE[(tb0 + lane) & (DP_RING - 1)] = te; ringS[(tb0 + lane) & (DP_RING - 1)] = ts; }
		}
		if (WHOLE) wsort::wave_mem_fence();		// the next tile's deep look-backs read these scores from LDS
		{
			const i32 tileMax = __builtin_amdgcn_readlane(wave_incl_max(valid ? ts : I32_MIN), 63);
			pmRun = max(pmRun, tileMax);
			const int t = tb0 >> 6;
			if (t < 64) pmv = lane == t ? pmRun : pmv;
		}
		pc = tc; pe = te; ps = ts;
	}
}

__global__ void __launch_bounds__(DP_WAVES * 64)
k_chain_dp(ChainParams P, const u32* __restrict__ list, u32 nList, u64 nGroups, u64 nHits,
		   const u64* __restrict__ groupStart, const u32* __restrict__ groupQuery,
		   const u32* __restrict__ query, const i32* __restrict__ len, const i32* __restrict__ qLen,
		   const uint8_t* __restrict__ groupExtSorted,
		   const u32* __restrict__ gCur, const u32* __restrict__ gExt,
		   i32* __restrict__ gScore, i32* __restrict__ gBack)
{
	const u32 li = blockIdx.x * DP_WAVES + (threadIdx.x >> 6);
	if (li >= nList) return;
	const u64 g = fg_uni(list[li]);
	const u64 g0 = fg_uni(groupStart[g]);
	const u64 gend = (g + 1 < nGroups) ? fg_uni(groupStart[g + 1]) : nHits;
	const i32 n = (i32)(gend - g0);
	const bool extSorted = fg_uni((u32)groupExtSorted[g]) != 0;	// decided by k_group_prep
	__shared__ i32 ring[DP_WAVES][3][DP_RING];
	i32* ringC = ring[threadIdx.x >> 6][0];
	i32* ringE = ring[threadIdx.x >> 6][1];
	i32* ringS = ring[threadIdx.x >> 6][2];
	if (extSorted) dp_group<true>(P.k, P.maxJump, n, gCur + g0, gExt + g0, gScore + g0, gBack + g0, ringC, ringE, ringS);
	else dp_group<false>(P.k, P.maxJump, n, gCur + g0, gExt + g0, gScore + g0, gBack + g0, ringC, ringE, ringS);
}

// ---- score order without the introsort emulation, where it cannot matter ------------------------------------------
// The backtracking visits the elements in descending score order and skips those without a back pointer wherever they
// stand (overlap.cpp:331-340).  So only the RELATIVE order of the elements that have one matters, and that order is
// the same under any sorting algorithm unless two of them share a score: 3 groups in 4 of the bench workload have no
// such tie (oracle, FO_STATS4).  Those take a bitonic network over (score, index) words held in registers -- ~100
// instructions for <= 64 hits against the ~10x of the partition-by-partition emulation; a group with a tie among its
// live elements (found by one neighbour compare on the sorted words) takes the emulation as before.
// R words per lane, element e = r * 64 + lane.  Returns the number of live elements (their indices, in visiting order,
// in oval[0 .. nLive)), or -1 when the order depends on std::sort's tie handling (nothing written).
template <int R>
__device__ __forceinline__ i32 fast_score_order(const i32 n, const i32* score, const i32* back, u32* oval)
{
	const int lane = threadIdx.x & 63;
	u32 v[R];
	bool wide = false;
#pragma unroll
	for (int r = 0; r < R; ++r)
	{
		const i32 e = r * 64 + lane;
		const bool live = e < n && back[e] != -1;
		const i32 sc = live ? score[e] : 0;
		wide |= live && (u32)sc >= (1u << 22);	// (index: 10 bits; scores this large are not seen: positions have 24 bits)
		v[r] = live ? ((u32)((1 << 22) - 1 - sc) << 10) | (u32)e : 0xFFFFFFFFu;
	}
	if (__builtin_amdgcn_ballot_w64(wide)) return -1;
#pragma unroll
	for (int k = 2; k <= 64 * R; k <<= 1)
	{
#pragma unroll
		for (int j = k >> 1; j > 0; j >>= 1)
		{
			if (j >= 64)
			{
#pragma unroll
				for (int r = 0; r < R; ++r)
				{
					const int rp = r ^ (j >> 6);
					if (rp > r)
					{
						const bool asc = ((r * 64) & k) == 0;	// k > 64 here: the bit sits in r; k == 64 R: always ascending
						const u32 lo = min(v[r], v[rp]), hi = max(v[r], v[rp]);
						v[r] = asc ? lo : hi; v[rp] = asc ? hi : lo;
					}
				}
			}
			else
			{
#pragma unroll
				for (int r = 0; r < R; ++r)
				{
					const u32 o = (u32)__shfl_xor((int)v[r], j);
					const bool asc = (((r * 64 + lane) & k) == 0);
					const bool lower = (lane & j) == 0;
					v[r] = (asc == lower) ? min(v[r], o) : max(v[r], o);
				}
			}
		}
	}
	// sorted ascending: live elements first (descending score), the others (all ones) behind them
	bool tie = false;
	i32 nLive = 0;
#pragma unroll
	for (int r = 0; r < R; ++r)
	{
		u32 nx = (u32)__shfl_down((int)v[r], 1);
		const u32 first = r + 1 < R ? (u32)__builtin_amdgcn_readlane((int)v[r + 1 < R ? r + 1 : r], 0) : 0xFFFFFFFFu;
		if (lane == 63) nx = first;
		tie |= nx != 0xFFFFFFFFu && (nx >> 10) == (v[r] >> 10);
		nLive += __popcll(__builtin_amdgcn_ballot_w64(v[r] != 0xFFFFFFFFu));
	}
	if (__builtin_amdgcn_ballot_w64(tie)) return -1;
#pragma unroll
	for (int r = 0; r < R; ++r)
		if (v[r] != 0xFFFFFFFFu) oval[r * 64 + lane] = v[r] & 1023u;
	return nLive;
}

// (The groups that live in global memory -- more than 256 hits -- keep the emulation: a stable LSD radix sort of their
// live (score, index) pairs with LDS counters, also with four tiles' loads in flight and the next pass's histogram
// taken while scattering, was measured no faster than the emulation it would replace -- HiFi 30x, 420 k such groups
// of ~1500 hits: 26 ms for the 77 % of groups without a tie against ~33 ms before, and the bench workload's
// k_chain_finish<global> went from 5.9 to 6.6 ms -- its scattered 4-byte stores are what it waits for.  The bitonic
// network on words in LDS for groups of 257..1024 hits: HiFi 87 -> 82 ms, bench workload 5.7 -> 6.3 ms: dropped too.)

// ---- finish ------------------------------------------------------------------------------
// One group's backtracking stage on one wave (the body of k_chain_finish and of the fused small-group kernel):
// score / back: the DP's arrays (LDS when USE_LDS, else the group's global ones; back is consumed), okey / oval:
// n u32 each for the score order, pl / pr: the sort's position lists (u16 in LDS, else u32 behind oval), btLds: LDS
// copy of the back pointers for the walk (BT_CAP > 0 only).
template <bool USE_LDS, int BT_CAP>
__device__ __forceinline__ void finish_group(const ChainParams& P, const i32 n, const u32 curId, const u32 extId, const i32 curLen,
											 const i32 extLen, const u32* cur, const u32* ext, i32* score, i32* back, u32* okey, u32* oval,
											 unsigned short* plLds, int* stack, int* small, i32* btLds, int4* cd, u32* primCountG)
{
	const int lane = threadIdx.x & 63;
	const int k = P.k;
	// chain starts in descending score order, ties as std::sort leaves them (overlap.cpp:331-334)
	i32 nOrder = -1;		// entries of oval to visit
	if (USE_LDS && !(P.ablate & 64))
	{
		if (n <= 64) nOrder = fast_score_order<1>(n, score, back, oval);
		else if (n <= 128) nOrder = fast_score_order<2>(n, score, back, oval);
		else if (n <= 256) nOrder = fast_score_order<4>(n, score, back, oval);
	}
	const bool exact = nOrder < 0;
	if (exact)
	{
		for (i32 i = lane; i < n; i += 64) { okey[i] = (u32)(0x7fffffff - score[i]); oval[i] = (u32)i; }
		nOrder = n;
	}
	wsort::wave_mem_fence();
	if (exact && !(P.ablate & 2))
	{
		if (USE_LDS)
			wsort::wave_sort<u32, unsigned short>(okey, oval, n, plLds, plLds + n, stack, small);
		else if (BT_CAP > 0 && n <= BT_CAP)
		{
			// the walk's LDS (4 B per hit) first serves as the sort's two position lists
			unsigned short* pl = (unsigned short*)btLds;
			wsort::wave_sort<u32, unsigned short>(okey, oval, n, pl, pl + BT_CAP, stack, small);
		}
		else wsort::wave_sort<u32, u32>(okey, oval, n, oval + n, oval + 2 * n, stack, small);
	}

	if (P.ablate & 8) return;
	if (!USE_LDS && P.keepAln)
	{
		// consume a copy (the sort's position scratch is free now), keep gBack intact
		i32* bc = (i32*)(oval + n);
		for (i32 i = lane; i < n; i += 64) bc[i] = back[i];
		back = bc;
	}
	if (BT_CAP > 0 && n <= BT_CAP)
	{
		// the serial pointer chase below is latency bound: keep the back pointers in LDS
		for (i32 i = lane; i < n; i += 64) btLds[i] = back[i];
		back = btLds;
	}
	// backtracking with consumption, overlapTest, primary selection.  Consumption only ever
	// turns back[] entries into -1, so a start whose entry already is -1 can be skipped for good:
	// the wave screens 64 order entries at once and lane 0 walks only the survivors (re-checking
	// each, since a chain walked in between may have consumed it).
	i32 ncand = 0;
	int4 best = make_int4(0, 0, 0, 0);
	wsort::wave_mem_fence();
	// The walk itself runs on a WINDOW of 64 consecutive back pointers held one per lane: a chain moves to
	// smaller indices, mostly by a few elements at a time (runs of same-diagonal hits move by one), so one
	// coalesced load serves dozens of hops that are then register reads (v_readlane) instead of dependent
	// round trips to LDS (~100 cycles) or to memory (~1 us: the HiFi workload's 1500-hit groups spent 82 of
	// 292 ms there).  Consumed entries are marked in the register and written back when the window moves.
	i32 wbase = -0x40000000, win = -1;
	bool dirty = false;
	auto flushWin = [&]()
	{
		if (dirty)
		{
			if (wbase + lane < n) back[wbase + lane] = win;
			wsort::wave_mem_fence();
			dirty = false;
		}
	};
	auto loadWin = [&](i32 pos)
	{
		flushWin();
		wbase = max(0, pos - 63);
		if (USE_LDS || (BT_CAP > 0 && n <= BT_CAP)) win = wbase + lane < n ? back[wbase + lane] : -1;
		else win = wbase + lane < n ? __hip_atomic_load(&back[wbase + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
	};
	const bool inMemory = !USE_LDS && !(BT_CAP > 0 && n <= BT_CAP);
	// (LDS-resident groups keep the plain lane-0 walk below: their chains are short and jump further, the
	// window reloads cost more than the LDS hops -- 9.9 against 7.4 ms on the bench workload)
	for (i32 oi0 = 0; inMemory && oi0 < nOrder; oi0 += 64)
	{
		flushWin();		// the screening below reads the array itself
		const bool in = oi0 + lane < nOrder;
		const i32 st = in ? (i32)oval[oi0 + lane] : 0;
		i32 bk = -1;
		if (in)
		{
			if (USE_LDS || (BT_CAP > 0 && n <= BT_CAP)) bk = back[st];
			else bk = __hip_atomic_load(&back[st], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		u64 m = __builtin_amdgcn_ballot_w64(bk != -1);
		while (m)
		{
			const int l = __ffsll((long long)m) - 1;
			m &= m - 1;
			const i32 start = __builtin_amdgcn_readlane(st, __builtin_amdgcn_readfirstlane(l));
			if (start < wbase || start >= wbase + 64) loadWin(start);
			if (__builtin_amdgcn_readlane(win, __builtin_amdgcn_readfirstlane(start - wbase)) == -1) continue;	// consumed meanwhile
			i32 firstM = 0, chainLength = 0, pos = start;
			while (true)
			{
				if (pos < wbase || pos >= wbase + 64) loadWin(pos);
				const int idx = __builtin_amdgcn_readfirstlane(pos - wbase);
				const i32 np = __builtin_amdgcn_readlane(win, idx);
				win = lane == idx ? -1 : win;
				dirty = true;
				firstM = pos;
				++chainLength;
				if (np == -1) break;
				pos = np;
			}
			const i32 cb = (i32)cur[firstM], eb = (i32)ext[firstM];
			const i32 ce = (i32)cur[start] + k - 1, ee = (i32)ext[start] + k - 1;
			if (overlap_test(P, curId, extId, curLen, extLen, cb, ce, eb, ee))
			{
				const int4 c4 = make_int4(firstM, start, chainLength, score[start] - score[firstM] + k - 1);
				if (lane == 0) cd[ncand] = c4;
				if (ncand == 0 || c4.w > best.w) best = c4;	// <= 16 candidates: insertion sort = stable
				++ncand;
			}
		}
	}
	for (i32 oi0 = 0; !inMemory && oi0 < nOrder; oi0 += 64)
	{
		const bool in = oi0 + lane < nOrder;
		const i32 st = in ? (i32)oval[oi0 + lane] : 0;
		const i32 bk = in ? back[st] : -1;
		u64 m = __builtin_amdgcn_ballot_w64(bk != -1);
		while (m)
		{
			const int l = __ffsll((long long)m) - 1;
			m &= m - 1;
			const i32 start = __builtin_amdgcn_readlane(st, __builtin_amdgcn_readfirstlane(l));
			// every start still in m is alive: the mask is screened again after each walk (the 64 starts of a
			// batch are mostly the upper elements of ONE chain -- its scores rise along it -- and the walk from
			// its tip consumes them all: one parallel look instead of one lane-0 look per dead start)
			if (lane == 0)
			{
				i32 firstM = 0, chainLength = 0, pos = start;
				while (pos != -1)
				{
					firstM = pos;
					++chainLength;
					const i32 np = back[pos];
					back[pos] = -1;
					pos = np;
				}
				const i32 cb = (i32)cur[firstM], eb = (i32)ext[firstM];
				const i32 ce = (i32)cur[start] + k - 1, ee = (i32)ext[start] + k - 1;
				if (overlap_test(P, curId, extId, curLen, extLen, cb, ce, eb, ee))
				{
					const int4 c4 = make_int4(firstM, start, chainLength, score[start] - score[firstM] + k - 1);
					cd[ncand] = c4;
					if (ncand == 0 || c4.w > best.w) best = c4;	// <= 16 candidates: insertion sort = stable
					++ncand;
				}
			}
			if (m)
			{
				wsort::wave_mem_fence();
				const i32 bk2 = in ? back[st] : -1;
				m &= __builtin_amdgcn_ballot_w64(bk2 != -1);
			}
		}
	}
	wsort::wave_mem_fence();
	if (lane != 0) return;
	if (ncand == 0) return;
	// candidates in descending score order as std::sort leaves them (overlap.cpp:432-434):
	// <= 16 elements is a plain (stable) insertion sort, more goes through the emulation
	if (ncand > 16)
	{
		CandAcc acc{cd};
		fgsort::sort(acc, 0, ncand, stack);	// 3*40 ints >= fgsort::STACK_INTS
		best = cd[0];
	}
	if (P.onlyMaxExt)
	{
		cd[0] = best;
		*primCountG = 1;
		return;
	}
	if (ncand <= 16)
		for (i32 a = 1; a < ncand; ++a)
		{
			const int4 v = cd[a];
			i32 b = a - 1;
			while (b >= 0 && cd[b].w < v.w) { cd[b + 1] = cd[b]; --b; }
			cd[b + 1] = v;
		}
	// keep a candidate unless a kept, strictly higher scoring one contains it (overlap.cpp:441-458)
	i32 nprim = 0;
	for (i32 a = 0; a < ncand; ++a)
	{
		const int4 o = cd[a];
		const i32 ocb = (i32)cur[o.x], oce = (i32)cur[o.y], oeb = (i32)ext[o.x], oee = (i32)ext[o.y];
		bool contained = false;
		for (i32 b = 0; b < nprim && !contained; ++b)
		{
			const int4 pr = cd[b];
			contained = pr.w > o.w && (i32)cur[pr.x] <= ocb && oce <= (i32)cur[pr.y] &&
						(i32)ext[pr.x] <= oeb && oee <= (i32)ext[pr.y];
		}
		if (!contained) cd[nprim++] = o;	// nprim <= a: never overwrites an unread candidate
	}
	*primCountG = (u32)nprim;
}


// CAP = 0: everything in global scratch; otherwise the group (<= CAP hits) is staged in LDS
// BT_CAP > 0 (CAP = 0 only): groups of <= BT_CAP hits walk their back pointers in LDS
template <int CAP, int FIN_WAVES, int BT_CAP = 0>
__global__ void __launch_bounds__(FIN_WAVES * 64)
k_chain_finish(ChainParams P, const u32* __restrict__ list, u32 nList, u64 nGroups, u64 nHits,
			   const u64* __restrict__ groupStart, const u32* __restrict__ groupQuery,
			   const u32* __restrict__ query, const i32* __restrict__ len, const i32* __restrict__ qLen,
			   const u32* __restrict__ groupExt,
			   const u32* __restrict__ gCur, const u32* __restrict__ gExt, i32* __restrict__ gScore,
			   i32* __restrict__ gBack, u32* __restrict__ gAux /* 4 u32 per hit */, int4* __restrict__ cand,
			   u32* __restrict__ primCount)
{
	constexpr bool USE_LDS = CAP > 0;
	// dynamic LDS: per wave CAP * 20 bytes (score, back, order key, order value, 2 x u16 scratch)
	extern __shared__ __attribute__((aligned(16))) char finLds[];
	__shared__ int stack[FIN_WAVES][3 * 40];
	__shared__ int small[FIN_WAVES][3 * 8];
	__shared__ i32 btLds[FIN_WAVES][BT_CAP > 0 ? BT_CAP : 1];
	const int wv = threadIdx.x >> 6;
	const int lane = threadIdx.x & 63;
	const u32 li = blockIdx.x * FIN_WAVES + wv;
	if (li >= nList) return;
	const u64 g = fg_uni(list[li]);
	const u64 g0 = fg_uni(groupStart[g]);
	const u64 gend = (g + 1 < nGroups) ? fg_uni(groupStart[g + 1]) : nHits;
	const i32 n = (i32)(gend - g0);
	const u32 q = groupQuery[g];
	const u32* cur = gCur + g0;
	const u32* ext = gExt + g0;
	i32 *score, *back; u32 *okey, *oval;
	unsigned short* plLds = nullptr;
	if (USE_LDS)
	{
		char* base = finLds + (size_t)wv * CAP * 20;
		score = (i32*)base; back = score + CAP; okey = (u32*)(back + CAP); oval = okey + CAP;
		plLds = (unsigned short*)(oval + CAP);
		for (i32 i = lane; i < n; i += 64) { score[i] = gScore[g0 + i]; back[i] = gBack[g0 + i]; }
		wsort::wave_mem_fence();
	}
	else
	{
		score = gScore + g0; back = gBack + g0; okey = gAux + 4 * g0; oval = okey + n;
	}
	const u32 qrec = query[q];
	const u32 extId = groupExt[g];
	finish_group<USE_LDS, BT_CAP>(P, n, P.qFirstId + qrec, extId, qLen[qrec >> 1], len[(extId - P.firstId) >> 1], cur, ext, score, back,
								  okey, oval, plLds, stack[wv], small[wv], btLds[wv], cand + g0, primCount + g);
}

// ---- the whole chaining stage of a small group in ONE kernel -----------------------------------------------------
// Groups of <= FIN_CAP_S hits (nine in ten of the groups that reach the DP on raw reads) go through prefilter, optional
// re-sort, DP, score-order sort, backtracking and primary selection on one wave without leaving LDS: what
// k_group_prep, k_chain_dp and k_chain_finish<lds256> did in three launches, handing (cur, ext) and (score, back)
// to one another through global memory.  Only what later kernels read is written out: the (cur, ext) columns in DP
// order (k_prim_gather, k_chain_matches), the back pointers when kmerMatches are wanted, the candidates.
template <class KT>
__global__ void __launch_bounds__(64)
k_chain_small(ChainParams P, const u32* __restrict__ list, u32 nList, u64 nGroups, u64 nHits,
			  const u64* __restrict__ groupStart, const u32* __restrict__ groupQuery,
			  const u32* __restrict__ query, const i32* __restrict__ len, const i32* __restrict__ qLen,
			  HitKeyView<KT> hitKey, const u32* __restrict__ groupExt,
			  u32* __restrict__ gCur, u32* __restrict__ gExt, i32* __restrict__ gBack, int4* __restrict__ cand,
			  u32* __restrict__ dpSize, u32* __restrict__ primCount, int cap /* largest group: 28 B of LDS per hit */)
{
	extern __shared__ __attribute__((aligned(16))) char smallLds[];
	u32* const sCur = (u32*)smallLds; u32* const sExt = sCur + cap;
	i32* const sScore = (i32*)(sExt + cap); i32* const sBack = sScore + cap;
	u32* const sOkey = (u32*)(sBack + cap); u32* const sOval = sOkey + cap;
	unsigned short* const sP = (unsigned short*)(sOval + cap);
	__shared__ int stack[3 * 40];
	__shared__ int small[3 * 8];
	const int lane = threadIdx.x;
	const u32 li = blockIdx.x;
	if (li >= nList) return;
	const u64 g = fg_uni(list[li]);
	const u64 g0 = fg_uni(groupStart[g]);
	const u64 gend = (g + 1 < nGroups) ? fg_uni(groupStart[g + 1]) : nHits;
	const i32 n = (i32)(gend - g0);
	const u32 qrec = query[groupQuery[g]];
	const u32 extId = groupExt[g];
	const i32 curLen = qLen[qrec >> 1];
	const i32 extLen = len[(extId - P.firstId) >> 1];
	const i32 minCur = (i32)hitKey.cur(g0), maxCur = (i32)hitKey.cur(g0 + n - 1);
	// distinct query positions (overlap.cpp:220-235; prevPos starts at 0), ext span; the hits are staged on the way
	u32 uniq = 0;
	i32 minExt = 0x7fffffff, maxExt = I32_MIN;
	for (i32 i = lane; i < n; i += 64)
	{
		const u32 c = hitKey.cur(g0 + i);
		const u32 pc = i ? hitKey.cur(g0 + i - 1) : 0u;
		uniq += (c != pc);
		const i32 e = (i32)hitKey.val(g0 + i);
		minExt = min(minExt, e); maxExt = max(maxExt, e);
		sCur[i] = c; sExt[i] = (u32)e;
	}
	for (int o = 32; o > 0; o >>= 1)
	{
		uniq += __shfl_xor(uniq, o);
		minExt = min(minExt, __shfl_xor(minExt, o));
		maxExt = max(maxExt, __shfl_xor(maxExt, o));
	}
	if ((float)uniq < P.minUnique) return;
	if (maxCur - minCur < P.minOverlap || maxExt - minExt < P.minOverlap) return;
	if (P.checkOverhang && !P.forceLocal)
	{
		if (min(minCur, minExt) > P.maxOverhang) return;
		if (min(curLen - maxCur, extLen - maxExt) > P.maxOverhang) return;
	}
	if (lane == 0) dpSize[g] = (u32)n;
	const bool extSorted = extLen > curLen;
	wsort::wave_mem_fence();
	if (extSorted)
	{
		// (strictly ascending target positions need no re-sort: see k_group_prep)
		bool asc = true;
		for (i32 i = lane + 1; i < n; i += 64) asc = asc && sExt[i] > sExt[i - 1];
		if (__builtin_amdgcn_ballot_w64(!asc))
			wsort::wave_sort<u32, unsigned short>(sExt, sCur, n, sP, sP + n, stack, small);
	}
	for (i32 i = lane; i < n; i += 64) { gCur[g0 + i] = sCur[i]; gExt[g0 + i] = sExt[i]; }
	wsort::wave_mem_fence();
	if (extSorted) dp_group<true, true>(P.k, P.maxJump, n, sCur, sExt, sScore, sBack, nullptr, nullptr, nullptr);
	else dp_group<false, true>(P.k, P.maxJump, n, sCur, sExt, sScore, sBack, nullptr, nullptr, nullptr);
	wsort::wave_mem_fence();
	if (P.keepAln)
		for (i32 i = lane; i < n; i += 64) gBack[g0 + i] = sBack[i];	// k_chain_matches walks them again; the copy in LDS is consumed
	finish_group<true, 0>(P, n, P.qFirstId + qrec, extId, curLen, extLen, sCur, sExt, sScore, sBack, sOkey, sOval, sP, stack, small,
						  nullptr, cand + g0, primCount + g);
}

u32 fetchU32(fg_ctx* c, const u32* dptr)
{
	u32 v;
	HIP_CHECK(hipMemcpyAsync(&v, dptr, 4, hipMemcpyDeviceToHost, c->stream));
	HIP_CHECK(hipStreamSynchronize(c->stream));
	return v;
}

} // namespace

// All target groups of the batch -> dPrimFlag[g] = number of primaries (their (first, last,
// chainLength, score) tuples at the head of the group's dCand region), dDpSize[g]
// Groups of <= this many hits take the one-kernel path (k_chain_small); 0: none do (FG_CHAIN_FUSED=0, the three-kernel
// path for all).  Its LDS: 28 B per hit of the largest group it takes; FG_FUSED_CAP for experiments.
u32 fgChainSmallMax()
{
	if (getenv("FG_CHAIN_FUSED") && atoi(getenv("FG_CHAIN_FUSED")) == 0) return 0u;
	return getenv("FG_FUSED_CAP") ? (u32)std::max(64, std::min(1024, atoi(getenv("FG_FUSED_CAP")))) : (u32)FIN_CAP_S;
}

void fgChainStage(fg_ctx* c, const fg_detector_params* p, uint8_t forceLocal, u64 nGroups, u64 nHits, int keyMode,
				  int curBits)
{
	hipStream_t s = c->stream;
	ChainParams cp;
	cp.k = c->k; cp.maxJump = p->max_jump; cp.minOverlap = p->min_overlap; cp.maxOverhang = p->max_overhang;
	cp.checkOverhang = p->max_overhang > 0; cp.forceLocal = forceLocal ? 1 : 0;
	{
		const float minKmerSruvivalRate = 0.01;	// overlap.cpp:110
		cp.minUnique = minKmerSruvivalRate * p->min_overlap;
	}
	cp.firstId = c->firstId;
	cp.qFirstId = c->hasQ ? c->qFirstId : c->firstId;
	cp.onlyMaxExt = p->only_max_ext ? 1 : 0;
	cp.keepAln = p->keep_alignment ? 1 : 0;
	cp.ablate = getenv("FG_ABLATE") ? atoi(getenv("FG_ABLATE")) : 0;
	if (!nGroups) return;
	const i32* qLen = c->hasQ ? c->dQLen.p : c->dLen.p;
	// smallest group size that can still have >= minUnique distinct query positions
	u32 minSize = 0;
	while ((float)minSize < cp.minUnique) ++minSize;
	if (minSize == 0) minSize = 1;
	c->dListSmall.reserve(nGroups + 1); c->dListBig.reserve(nGroups + 1); c->dListDp.reserve(nGroups + 1);
	c->dGroupExtSorted.reserve(nGroups + 1);
	c->dListCnt.reserve(4);
	const u64 hitCap = std::max<u64>(nHits, c->hitCapHint);	// sized once for the largest sub-range of the chunk
	c->dCur.reserve(hitCap + 16); c->dExt.reserve(hitCap + 16);
	c->dScore.reserve(hitCap + 16); c->dBack.reserve(hitCap + 16);
	c->dTmp32.reserve(4 * hitCap + 16);
	c->dCand.reserve(hitCap + 1);
	const unsigned gridG = (unsigned)((nGroups + (u64)WG * LIST_ITEMS - 1) / ((u64)WG * LIST_ITEMS));
	const u32 fusedMax = fgChainSmallMax();
	const bool fused = fusedMax != 0;
	if (fused) c->dListFused.reserve(nGroups + 1);
	HIP_CHECK(hipMemsetAsync(c->dListCnt.p, 0, 16, s));
	{ ScopedK t(c->timer, "k_group_list");
	  hipLaunchKernelGGL(k_group_list, gridG, WG, 0, s, nGroups, nHits, c->dGroupStart.p, minSize, c->dGroupFirstCur.p,
						 c->dGroupLastCur.p, (i32)p->min_overlap, c->dListSmall.p, c->dListBig.p, (u32)PREP_CAP,
						 c->dListFused.p, fusedMax, c->dListCnt.p, c->dPrimFlag.p, c->dDpSize.p); }
	u32 nPrep[4];
	c->hScalar.reserve(8);
	HIP_CHECK(hipMemcpyAsync(c->hScalar.p, c->dListCnt.p, 16, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	memcpy(nPrep, c->hScalar.p, 16);
	const u32 nFused = fused ? nPrep[2] : 0u;
	if (!nPrep[0] && !nPrep[1] && !nFused) return;
	if (fused)
	{
		// Main stream: the small groups, start to end in one kernel.  Beside it on the side streams: the larger groups'
		// prefilter (stream 2), their size classes listed, then the two classes' DP -> finish chains (streams 2 and 3).
		const bool others = nPrep[0] || nPrep[1];
		const bool sideStreams = !(getenv("FG_CHAIN_STREAMS") && atoi(getenv("FG_CHAIN_STREAMS")) == 1);	// 1: everything on the main stream
		if (others && sideStreams)
		{
			HIP_CHECK(hipEventRecord(c->evFork, s));
			HIP_CHECK(hipStreamWaitEvent(c->stream2, c->evFork, 0));
		}
		if (nFused)
		{
			ScopedK t(c->timer, "k_chain_small");
#define SMALL_ARGS(view) cp, c->dListFused.p, nFused, nGroups, nHits, c->dGroupStart.p, c->dGroupQuery.p, c->curQuery, c->dLen.p, qLen, \
			view, c->dGroupExt.p, c->dCur.p, c->dExt.p, c->dBack.p, c->dCand.p, c->dDpSize.p, c->dPrimFlag.p, (int)fusedMax
			const size_t ldsSmall = (size_t)fusedMax * 28;
			if (keyMode == 0)
				hipLaunchKernelGGL(k_chain_small<u32>, nFused, 64, ldsSmall, s, SMALL_ARGS((HitKeyView<u32>{c->dHitKey32.p, c->dHitVal.p, curBits, c->firstId})));
			else if (keyMode == 1)
				hipLaunchKernelGGL(k_chain_small<PK>, nFused, 64, ldsSmall, s, SMALL_ARGS((HitKeyView<PK>{(const PK*)c->dHitKey.p, nullptr, curBits, c->firstId})));
			else
				hipLaunchKernelGGL(k_chain_small<u64>, nFused, 64, ldsSmall, s, SMALL_ARGS((HitKeyView<u64>{c->dHitKey.p, c->dHitVal.p, curBits, c->firstId})));
#undef SMALL_ARGS
		}
		if (!others) return;
		hipStream_t s2 = sideStreams ? c->stream2 : s, s3 = sideStreams ? c->stream3 : s;
		const u32* prepList[2] = {c->dListSmall.p, c->dListBig.p};
#define PREP_ARGS(view, cls) cp, prepList[cls], nPrep[cls], nGroups, nHits, c->dGroupStart.p, c->dGroupQuery.p, c->curQuery, c->dLen.p, qLen, \
		view, c->dGroupExt.p, c->dCur.p, c->dExt.p, c->dTmp32.p, c->dDpSize.p, c->dGroupExtSorted.p
		for (int cls = 1; cls >= 0; --cls)
		{
			if (!nPrep[cls]) continue;
			ScopedK t(c->timer, "k_group_prep", s2);
			const unsigned gridP = (nPrep[cls] + PREP_WAVES - 1) / PREP_WAVES;
			if (keyMode == 0)
				hipLaunchKernelGGL(k_group_prep<u32>, gridP, PREP_WAVES * 64, 0, s2,
								   PREP_ARGS((HitKeyView<u32>{c->dHitKey32.p, c->dHitVal.p, curBits, c->firstId}), cls));
			else if (keyMode == 1)
				hipLaunchKernelGGL(k_group_prep<PK>, gridP, PREP_WAVES * 64, 0, s2,
								   PREP_ARGS((HitKeyView<PK>{(const PK*)c->dHitKey.p, nullptr, curBits, c->firstId}), cls));
			else
				hipLaunchKernelGGL(k_group_prep<u64>, gridP, PREP_WAVES * 64, 0, s2,
								   PREP_ARGS((HitKeyView<u64>{c->dHitKey.p, c->dHitVal.p, curBits, c->firstId}), cls));
		}
#undef PREP_ARGS
		// (the group lists of k_group_list are dead once the preps have run: dListSmall is reused for the empty small class)
		HIP_CHECK(hipMemsetAsync(c->dListCnt.p, 0, 16, s2));
		const bool smallCall2 = (nPrep[0] + nPrep[1] + nFused) < 65536u && !getenv("FG_CHAIN_NO_SMALL_CALL");
		const u32 hugeMin2 = getenv("FG_CHAIN_HUGE_MIN") ? (u32)atoi(getenv("FG_CHAIN_HUGE_MIN")) : (smallCall2 ? (u32)FIN_CAP_M : 4096u);
		{ ScopedK t(c->timer, "k_dp_list", s2);
		  hipLaunchKernelGGL(k_dp_list, gridG, WG, 0, s2, nGroups, c->dDpSize.p, hugeMin2, fusedMax, c->dListSmall.p, c->dListDp.p,
							 c->dListBig.p, c->dListCnt.p); }
		u32 hc2[4];
		HIP_CHECK(hipMemcpyAsync(c->hScalar.p, c->dListCnt.p, 16, hipMemcpyDeviceToHost, s2));
		HIP_CHECK(hipStreamSynchronize(s2));		// the side stream only: the small groups' kernel keeps running
		memcpy(hc2, c->hScalar.p, 16);
		const u32* lists2[3] = {c->dListSmall.p, c->dListDp.p, c->dListBig.p};
		if (hc2[2] && sideStreams)
		{
			HIP_CHECK(hipEventRecord(c->evJoin3, s2));		// (used as a fork here: stream 3 starts behind the lists)
			HIP_CHECK(hipStreamWaitEvent(s3, c->evJoin3, 0));
		}
#define FIN_ARGS2(cls) cp, lists2[cls], hc2[cls], nGroups, nHits, c->dGroupStart.p, c->dGroupQuery.p, c->curQuery, c->dLen.p, \
		qLen, c->dGroupExt.p, c->dCur.p, c->dExt.p, c->dScore.p, c->dBack.p, c->dTmp32.p, c->dCand.p, c->dPrimFlag.p
#define DP_ARGS2(cls) cp, lists2[cls], hc2[cls], nGroups, nHits, c->dGroupStart.p, c->dGroupQuery.p, c->curQuery, c->dLen.p, qLen, \
		c->dGroupExtSorted.p, c->dCur.p, c->dExt.p, c->dScore.p, c->dBack.p
		for (int cls = 2; cls >= 1; --cls)
		{
			if (!hc2[cls]) continue;
			hipStream_t on = cls == 2 ? s3 : s2;
			{ ScopedK t(c->timer, "k_chain_dp", on);
			  hipLaunchKernelGGL(k_chain_dp, (hc2[cls] + DP_WAVES - 1) / DP_WAVES, DP_WAVES * 64, 0, on, DP_ARGS2(cls)); }
			if (cls == 1 && smallCall2 && hugeMin2 <= (u32)FIN_CAP_M)
			{
				ScopedK t(c->timer, "k_chain_finish<lds1024>", on);
				hipLaunchKernelGGL((k_chain_finish<FIN_CAP_M, 1>), hc2[cls], 64, FIN_CAP_M * 20, on, FIN_ARGS2(cls));
			}
			else
			{
				ScopedK t(c->timer, "k_chain_finish<global>", on);
				hipLaunchKernelGGL((k_chain_finish<0, FIN_WAVES_G, FIN_BT_CAP>), (hc2[cls] + FIN_WAVES_G - 1) / FIN_WAVES_G, FIN_WAVES_G * 64, 0, on,
								   FIN_ARGS2(cls));
			}
		}
		if (hc2[0])		// groups above the fused kernel's cap that still fit the 256-hit LDS class (FG_FUSED_CAP < 256)
		{
			{ ScopedK t(c->timer, "k_chain_dp", s2);
			  hipLaunchKernelGGL(k_chain_dp, (hc2[0] + DP_WAVES - 1) / DP_WAVES, DP_WAVES * 64, 0, s2, DP_ARGS2(0)); }
			{ ScopedK t(c->timer, "k_chain_finish<lds256>", s2);
			  hipLaunchKernelGGL((k_chain_finish<FIN_CAP_S, FIN_WAVES_S>), (hc2[0] + FIN_WAVES_S - 1) / FIN_WAVES_S, FIN_WAVES_S * 64,
								 FIN_CAP_S * 20 * FIN_WAVES_S, s2, FIN_ARGS2(0)); }
		}
#undef DP_ARGS2
#undef FIN_ARGS2
		if (sideStreams)
		{
			HIP_CHECK(hipEventRecord(c->evJoin, s2));
			HIP_CHECK(hipStreamWaitEvent(s, c->evJoin, 0));
			if (hc2[2])
			{
				HIP_CHECK(hipEventRecord(c->evJoin3, s3));
				HIP_CHECK(hipStreamWaitEvent(s, c->evJoin3, 0));
			}
		}
		return;
	}
	// Both stages below run their big-group class on the side stream beside the small-group class on the main
	// one: the classes are disjoint sets of groups, and the big-group kernels end with a handful of waves on an
	// otherwise idle chip (long serial work per wave).  FG_CHAIN_STREAMS=1 puts everything on the main stream.
	const bool twoStreams = !(getenv("FG_CHAIN_STREAMS") && atoi(getenv("FG_CHAIN_STREAMS")) == 1);
	auto fork = [&](bool both) -> hipStream_t
	{
		if (!twoStreams || !both) return s;
		HIP_CHECK(hipEventRecord(c->evFork, s));
		HIP_CHECK(hipStreamWaitEvent(c->stream2, c->evFork, 0));
		return c->stream2;
	};
	auto join = [&](hipStream_t side)
	{
		if (side == s) return;
		HIP_CHECK(hipEventRecord(c->evJoin, side));
		HIP_CHECK(hipStreamWaitEvent(s, c->evJoin, 0));
	};
	{
		hipStream_t sBig = fork(nPrep[0] && nPrep[1]);
		const u32* prepList[2] = {c->dListSmall.p, c->dListBig.p};
#define PREP_ARGS(view, cls) cp, prepList[cls], nPrep[cls], nGroups, nHits, c->dGroupStart.p, c->dGroupQuery.p, c->curQuery, c->dLen.p, qLen, \
		view, c->dGroupExt.p, c->dCur.p, c->dExt.p, c->dTmp32.p, c->dDpSize.p, c->dGroupExtSorted.p
		for (int cls = 1; cls >= 0; --cls)
		{
			if (!nPrep[cls]) continue;
			hipStream_t on = cls ? sBig : s;
			ScopedK t(c->timer, "k_group_prep", on);
			const unsigned gridP = (nPrep[cls] + PREP_WAVES - 1) / PREP_WAVES;
			if (keyMode == 0)
				hipLaunchKernelGGL(k_group_prep<u32>, gridP, PREP_WAVES * 64, 0, on,
								   PREP_ARGS((HitKeyView<u32>{c->dHitKey32.p, c->dHitVal.p, curBits, c->firstId}), cls));
			else if (keyMode == 1)
				hipLaunchKernelGGL(k_group_prep<PK>, gridP, PREP_WAVES * 64, 0, on,
								   PREP_ARGS((HitKeyView<PK>{(const PK*)c->dHitKey.p, nullptr, curBits, c->firstId}), cls));
			else
				hipLaunchKernelGGL(k_group_prep<u64>, gridP, PREP_WAVES * 64, 0, on,
								   PREP_ARGS((HitKeyView<u64>{c->dHitKey.p, c->dHitVal.p, curBits, c->firstId}), cls));
		}
#undef PREP_ARGS
		join(sBig);
	}
	HIP_CHECK(hipMemsetAsync(c->dListCnt.p, 0, 16, s));
	// A small call (a few hundred reads at most: what a caller thread's single request or the start of a read-ahead
	// ramp looks like) is all latency: a handful of 257..1024-hit groups then sort and walk in LDS (20 KB per wave,
	// which the bulk case cannot afford: occupancy) instead of on global scratch, ~1 ms off such a call.
	const bool smallCall = (nPrep[0] + nPrep[1]) < 65536u && !getenv("FG_CHAIN_NO_SMALL_CALL");
	const u32 hugeMin = getenv("FG_CHAIN_HUGE_MIN") ? (u32)atoi(getenv("FG_CHAIN_HUGE_MIN")) : (smallCall ? (u32)FIN_CAP_M : 4096u);
	{ ScopedK t(c->timer, "k_dp_list");
	  hipLaunchKernelGGL(k_dp_list, gridG, WG, 0, s, nGroups, c->dDpSize.p, hugeMin, 0u, c->dListSmall.p, c->dListDp.p,
						 c->dListBig.p, c->dListCnt.p); }
	u32 hc[4];
	HIP_CHECK(hipMemcpyAsync(c->hScalar.p, c->dListCnt.p, 16, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	memcpy(hc, c->hScalar.p, 16);
	const u32* lists[3] = {c->dListSmall.p, c->dListDp.p, c->dListBig.p};	// <= 256 hits, <= hugeMin, more
	// three independent chains of two kernels: the <= 256-hit groups on the main stream, the two classes of
	// larger groups (few groups, long serial work per wave) beside them on side streams
	const bool side = twoStreams && hc[0] && (hc[1] || hc[2]);
	hipStream_t sMid = s, sHuge = s;
	if (side)
	{
		HIP_CHECK(hipEventRecord(c->evFork, s));
		HIP_CHECK(hipStreamWaitEvent(c->stream2, c->evFork, 0));
		HIP_CHECK(hipStreamWaitEvent(c->stream3, c->evFork, 0));
		sMid = c->stream2; sHuge = c->stream3;
	}
#define FIN_ARGS(cls) cp, lists[cls], hc[cls], nGroups, nHits, c->dGroupStart.p, c->dGroupQuery.p, c->curQuery, c->dLen.p, \
		qLen, c->dGroupExt.p, c->dCur.p, c->dExt.p, c->dScore.p, c->dBack.p, c->dTmp32.p, c->dCand.p, c->dPrimFlag.p
#define DP_ARGS(cls) cp, lists[cls], hc[cls], nGroups, nHits, c->dGroupStart.p, c->dGroupQuery.p, c->curQuery, c->dLen.p, qLen, \
		c->dGroupExtSorted.p, c->dCur.p, c->dExt.p, c->dScore.p, c->dBack.p
	for (int cls = 2; cls >= 1; --cls)	// the longest chains first
	{
		if (!hc[cls]) continue;
		hipStream_t on = cls == 2 ? sHuge : sMid;
		{ ScopedK t(c->timer, "k_chain_dp", on);
		  hipLaunchKernelGGL(k_chain_dp, (hc[cls] + DP_WAVES - 1) / DP_WAVES, DP_WAVES * 64, 0, on, DP_ARGS(cls)); }
		if (cls == 1 && smallCall && hugeMin <= (u32)FIN_CAP_M)
		{
			ScopedK t(c->timer, "k_chain_finish<lds1024>", on);
			hipLaunchKernelGGL((k_chain_finish<FIN_CAP_M, 1>), hc[cls], 64, FIN_CAP_M * 20, on, FIN_ARGS(cls));
		}
		else
		{
			ScopedK t(c->timer, "k_chain_finish<global>", on);
			hipLaunchKernelGGL((k_chain_finish<0, FIN_WAVES_G, FIN_BT_CAP>), (hc[cls] + FIN_WAVES_G - 1) / FIN_WAVES_G, FIN_WAVES_G * 64, 0, on,
							   FIN_ARGS(cls));
		}
	}
	if (hc[0])
	{
		{ ScopedK t(c->timer, "k_chain_dp");
		  hipLaunchKernelGGL(k_chain_dp, (hc[0] + DP_WAVES - 1) / DP_WAVES, DP_WAVES * 64, 0, s, DP_ARGS(0)); }
		{ ScopedK t(c->timer, "k_chain_finish<lds256>");
		  hipLaunchKernelGGL((k_chain_finish<FIN_CAP_S, FIN_WAVES_S>), (hc[0] + FIN_WAVES_S - 1) / FIN_WAVES_S, FIN_WAVES_S * 64,
							 FIN_CAP_S * 20 * FIN_WAVES_S, s, FIN_ARGS(0)); }
	}
	if (side)
	{
		HIP_CHECK(hipEventRecord(c->evJoin, c->stream2));
		HIP_CHECK(hipStreamWaitEvent(s, c->evJoin, 0));
		HIP_CHECK(hipEventRecord(c->evJoin3, c->stream3));
		HIP_CHECK(hipStreamWaitEvent(s, c->evJoin3, 0));
	}
#undef DP_ARGS
#undef FIN_ARGS
}
