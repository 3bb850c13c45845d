// Wave-level (64 lanes) exact emulation of GCC libstdc++ std::sort on (key, val)
// pairs, as a permutation: the quicksort phase of introsort is reproduced partition
// by partition (median of (first+1, mid, last-1) to first; unguarded Hoare
// partition), the final insertion sort as a stable placement, the depth-limit
// heapsort by the sequential emulation of include/introsort_emul.h.
//
// Used for the three tie-sensitive sorts of getSeqOverlaps (reference
// src/sequence/overlap.cpp:201-204, :269-275, :331-334).  K / V may live in global
// memory or in LDS (the functions are inlined; hipcc resolves the address space).
//
// Closed form of the two-pointer Hoare loop used everywhere below: with
//   l_1 < l_2 < ...  the positions (left to right)  whose key is >= pivot,
//   r_1 > r_2 > ...  the positions (right to left)  whose key is <= pivot,
// the loop swaps (l_k, r_k) for k = 1..K where K = max{k : l_k < r_k}, and returns
// cut = min(l_{K+1}, r_K)  (r_0 := last).
#pragma once

#include "fg_ctx.h"
#include "../../include/introsort_emul.h"

namespace wsort {

// timing experiments only (build with -DFG_SORT_ABLATE, then FG_ABLATE env): 16 = skip the register quicksort
// phase of sort_small, 32 = skip its final placement, 64 = skip partition pass B.  Compiled out otherwise: a
// check is a scalar load of a global in the middle of a dependent chain, once per partition (measured: a
// similar check in k_chain_dp's head loop cost 20 % of that kernel).
#ifdef FG_SORT_ABLATE
__device__ int g_ablate = 0;
#define FG_SORT_ABLATED(bit) (g_ablate & (bit))
#else
#define FG_SORT_ABLATED(bit) false
#endif

template <class KT>
struct KV { KT k; u32 v; };

template <class KT, class VP = u32*>
struct PtrAcc {
	typedef KV<KT> T;
	KT* K; VP V;
	__device__ __forceinline__ T load(int i) const { return T{K[i], V[i]}; }
	__device__ __forceinline__ void store(int i, const T& x) { K[i] = x.k; V[i] = x.v; }
	__device__ __forceinline__ bool less(const T& a, const T& b) const { return a.k < b.k; }
};

__device__ __forceinline__ int nth_set_bit(u64 m, int n)
{
	int pos = 0;
	u32 c = __popc((u32)m);
	if (n >= (int)c) { n -= c; pos += 32; m >>= 32; }
	c = __popc((u32)m & 0xFFFFu); if (n >= (int)c) { n -= c; pos += 16; m >>= 16; }
	c = __popc((u32)m & 0xFFu); if (n >= (int)c) { n -= c; pos += 8; m >>= 8; }
	c = __popc((u32)m & 0xFu); if (n >= (int)c) { n -= c; pos += 4; m >>= 4; }
	c = __popc((u32)m & 0x3u); if (n >= (int)c) { n -= c; pos += 2; m >>= 2; }
	c = (u32)m & 1u; if (n >= (int)c) { pos += 1; }
	return pos;
}
__device__ __forceinline__ int nth_set_bit_desc(u64 m, int n) { return 63 - nth_set_bit(__brevll(m), n); }

__device__ __forceinline__ u64 shflk(u64 v, int src)
{
	u32 lo = __shfl((u32)v, src), hi = __shfl((u32)(v >> 32), src);
	return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u32 shflk(u32 v, int src) { return __shfl(v, src); }
__device__ __forceinline__ PK shflk(PK v, int src) { return PK(shflk(v.v, src)); }

// whole-wave rotation by one lane (DPP wave_ror:1 = 0x13C, wave_rol:1 = 0x134)
template <int CTRL>
__device__ __forceinline__ u32 rot_key(u32 v)
{
	return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ u64 rot_key(u64 v)
{
	const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, CTRL, 0xf, 0xf, false);
	const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), CTRL, 0xf, 0xf, false);
	return ((u64)hi << 32) | lo;
}

// wave-uniform values the compiler cannot prove uniform (loaded from memory): pinning them to
// SGPRs turns the sort's control flow into scalar branches and its index arithmetic into SALU
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// element of a uniform lane as a scalar
__device__ __forceinline__ u32 lane_get(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ u64 lane_get(u64 v, int l)
{
	return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)v, l);
}
// data written by some lanes of this wave is re-read by other lanes
__device__ __forceinline__ void wave_mem_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

template <class KT>
__device__ __forceinline__ int median3(KT ka, KT kb, KT kc)
{
	if (ka < kb) { if (kb < kc) return 1; else if (ka < kc) return 2; else return 0; }
	else if (ka < kc) return 0;
	else if (kb < kc) return 2;
	return 1;
}

// Hoare partition of one element per lane, lanes [lo,hi) take part (at most 63 lanes,
// so stop ranks stay below 63 and lane 63 can serve as the dump target of the pushes);
// returns the cut lane; key/val are replaced by the lane's new element.
// Stops are matched through the LDS crossbar instead of bit-select arithmetic: every
// left stop pushes its lane id to lane rankL, every right stop to lane rankR (counted
// from the top); lane r then holds the pair (l_{r+1}, r_{r+1}) and tests l < r.
template <class KT>
__device__ __forceinline__ int lane_partition(KT& key, u32& val, int lo, int hi, KT pk)
{
	const int lane = threadIdx.x & 63;
	const bool in = lane >= lo && lane < hi;
	const bool ge = in && key >= pk;
	const bool le = in && key <= pk;
	const u64 mL = __builtin_amdgcn_ballot_w64(ge), mR = __builtin_amdgcn_ballot_w64(le);
	const int cL = __popcll(mL), cR = __popcll(mR);
	const int rankL = __builtin_amdgcn_mbcnt_hi((u32)(mL >> 32), __builtin_amdgcn_mbcnt_lo((u32)mL, 0));
	const int belowR = __builtin_amdgcn_mbcnt_hi((u32)(mR >> 32), __builtin_amdgcn_mbcnt_lo((u32)mR, 0));
	const int rankR = cR - 1 - belowR;	// meaningful on right stops only
	const int idxL = __builtin_amdgcn_ds_permute((ge ? rankL : 63) << 2, lane);
	const int idxR = __builtin_amdgcn_ds_permute((le ? rankR : 63) << 2, lane);
	const int nPairs = cL < cR ? cL : cR;
	const int K = __popcll(__builtin_amdgcn_ballot_w64(lane < nPairs && idxL < idxR));	// l_k < r_k is monotone in k
	const int partL = __builtin_amdgcn_ds_bpermute((ge ? rankL : lane) << 2, idxR);
	const int partR = __builtin_amdgcn_ds_bpermute((le ? rankR : lane) << 2, idxL);
	int src = lane;
	if (ge && rankL < K) src = partL;
	else if (le && rankR < K) src = partR;
	key = shflk(key, src);
	val = __shfl(val, src);
	const int lK1 = (cL > K) ? __builtin_amdgcn_readlane(idxL, __builtin_amdgcn_readfirstlane(K)) : 0x7fffffff;
	const int rK = (K >= 1) ? __builtin_amdgcn_readlane(idxR, __builtin_amdgcn_readfirstlane(K - 1)) : hi;
	return lK1 < rK ? lK1 : rK;
}

// segment of <= 64 elements entirely in registers: quicksort phase, then the final
// insertion sort as a stable rank.  stk: >= 24 ints private to the wave.
template <class KT>
__device__ __forceinline__ void sort_small(KT* K, u32* V, int first, int n, int depth, int* stk)
{
	const int lane = threadIdx.x & 63;
	first = uni(first); n = uni(n); depth = uni(depth);
	KT key = lane < n ? K[first + lane] : (KT)0;
	u32 val = lane < n ? V[first + lane] : 0u;
	// pending segments: the smaller part is continued, the larger one waits (only if it is
	// still > 16), so at most two wait at any time (<= 64 -> <= 32 -> done); a, b, d and the
	// cut are wave-uniform, the "stack" is two scalar slots
	int sp = 0, p0a = 0, p0b = 0, p0d = 0, p1a = 0, p1b = 0, p1d = 0;
	int a = 0, b = n, d = depth;
	(void)stk;
	while (true)
	{
		while (b - a > 16 && !FG_SORT_ABLATED(16))
		{
			if (d == 0)
			{
				if (lane < n) { K[first + lane] = key; V[first + lane] = val; }
				wave_mem_fence();
				if (lane == 0) { PtrAcc<KT> acc{K, V}; fgsort::heap_sort_(acc, first + a, first + b); }
				wave_mem_fence();
				if (lane < n) { key = K[first + lane]; val = V[first + lane]; }
				break;
			}
			--d;
			// a, b are scalars: the three pivot candidates and the swap of the median into
			// position a go through v_readlane and selects, no LDS round trip
			const int mid = a + (b - a) / 2;
			const KT ka = lane_get(key, a + 1), kb = lane_get(key, mid), kc = lane_get(key, b - 1);
			const int m3 = median3(ka, kb, kc);
			const int pick = m3 == 0 ? a + 1 : (m3 == 1 ? mid : b - 1);
			const KT pk = m3 == 0 ? ka : (m3 == 1 ? kb : kc);
			const KT k0 = lane_get(key, a);
			const u32 v0 = lane_get(val, a), vp = lane_get(val, pick);
			key = lane == a ? pk : (lane == pick ? k0 : key);
			val = lane == a ? vp : (lane == pick ? v0 : val);
			const int cut = lane_partition(key, val, a + 1, b, pk);
			int oa, ob;
			if (cut - a < b - cut) { oa = cut; ob = b; b = cut; }
			else { oa = a; ob = cut; a = cut; }
			if (ob - oa > 16)
			{
				if (sp == 0) { p0a = oa; p0b = ob; p0d = d; } else { p1a = oa; p1b = ob; p1d = d; }
				++sp;
			}
		}
		if (sp == 0) break;
		--sp;
		if (sp == 0) { a = p0a; b = p0b; d = p0d; } else { a = p1a; b = p1b; d = p1d; }
	}
	// final insertion sort = stable placement.  After the quicksort phase every element
	// sits inside its own <= 16-element leaf and the leaves are mutually ordered, so the
	// target is lane - #(greater keys among the 15 lanes below) + #(smaller keys among the
	// 15 lanes above).  Neighbours arrive by DPP wave rotations (no LDS traffic); the lane
	// id travels with the key, so the count does not depend on the rotation direction.
	int pos = lane;
	if (!FG_SORT_ABLATED(32))
	{
		// Each pair of lanes at distance <= 15 is compared once: a copy of every element
		// travels 15 lanes upwards; the lane it visits and the visitor both book the outcome
		// (the visitor in `acc`, which rotates along and is brought home by one bpermute).
		// The key copies are SHIFTED (zeros enter at lane 0, and lanes >= n hold the maximum
		// key), so neither a wrapped-around nor an out-of-range visitor ever counts as greater.
		const int probe = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_update_dpp(0, lane, 0x13C, 0xf, 0xf, false));
		if (probe == 63)	// wave_ror:1 / wave_shr:1 hand lane i the value of lane i-1
		{
			const KT hk = lane < n ? key : ~(KT)0;	// as a host, a padding lane is never overtaken
			KT vk = hk; int acc = 0;
	#pragma unroll
			for (int d = 0; d < 15; ++d)
			{
				vk = rot_key<0x138>(vk);
				acc = __builtin_amdgcn_update_dpp(0, acc, 0x13C, 0xf, 0xf, false);
				// a greater visitor from below must end up above me
				const int c = vk > hk ? 1 : 0;
				pos -= c;
				acc += c;
			}
			pos += __shfl(acc, (lane + 15) & 63);
		}
		else
		{
			// direction-agnostic form: the visitor's lane id travels with the key
			KT vk = key; int vid = lane; int acc = 0;
			for (int d = 0; d < 15; ++d)
			{
				vk = rot_key<0x13C>(vk);
				vid = __builtin_amdgcn_update_dpp(0, vid, 0x13C, 0xf, 0xf, false);
				acc = __builtin_amdgcn_update_dpp(0, acc, 0x13C, 0xf, 0xf, false);
				const int dist = vid - lane;
				const bool ok = vid < n && lane < n && dist >= -15 && dist <= 15;
				const bool fromBelow = dist < 0;
				const int c = (ok && (fromBelow ? (vk > key) : (vk < key))) ? 1 : 0;
				pos += fromBelow ? -c : c;
				acc += fromBelow ? c : -c;
			}
			const int delta = __builtin_amdgcn_readfirstlane((lane - vid) & 63);
			pos += __shfl(acc, (lane + delta) & 63);
		}
	}
	wave_mem_fence();
	if (lane < n) { K[first + pos] = key; V[first + pos] = val; }
}

// segment of > 64 elements in memory: pivot to first, then the closed form of the
// Hoare loop.  Pass A lists the stop positions of both pointers (posL: keys >= pivot,
// left to right; posR: keys <= pivot, left to right, read back to front); pass B swaps
// pair k while l_k < r_k.  No step depends on the data of the previous tile, so the
// loads of several tiles are in flight together.  posL/posR: >= (last-first) entries each.
template <class KT, class PT>
__device__ __forceinline__ int partition_big(KT* K, u32* V, int first, int last, PT* posL, PT* posR)
{
	const int lane = threadIdx.x & 63;
	first = uni(first); last = uni(last);
	const int mid = first + (last - first) / 2;
	const KT ka = K[first + 1], kb = K[mid], kc = K[last - 1];
	const int m3 = uni(median3(ka, kb, kc));
	const int pick = m3 == 0 ? first + 1 : (m3 == 1 ? mid : last - 1);
	const KT pk = m3 == 0 ? ka : (m3 == 1 ? kb : kc);
	if (lane == 0)
	{
		const KT k0 = K[first]; const u32 v0 = V[first]; const u32 vp = V[pick];
		K[first] = pk; V[first] = vp;
		K[pick] = k0; V[pick] = v0;
	}
	wave_mem_fence();
	const int base = first + 1;
	const int m = last - base;
	const u64 below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
	int cL = 0, cR = 0;
	int t = 0;
	for (; t + 256 <= m; t += 256)
	{
		KT k0 = K[base + t + lane], k1 = K[base + t + 64 + lane], k2 = K[base + t + 128 + lane], k3 = K[base + t + 192 + lane];
#define FG_TILE(kk, off) { \
		const bool ge = kk >= pk, le = kk <= pk; \
		const u64 mL = __builtin_amdgcn_ballot_w64(ge), mR = __builtin_amdgcn_ballot_w64(le); \
		if (ge) posL[cL + __popcll(mL & below)] = (PT)(t + off + lane); \
		if (le) posR[cR + __popcll(mR & below)] = (PT)(t + off + lane); \
		cL += __popcll(mL); cR += __popcll(mR); }
		FG_TILE(k0, 0) FG_TILE(k1, 64) FG_TILE(k2, 128) FG_TILE(k3, 192)
	}
	for (; t < m; t += 64)
	{
		const int i = t + lane;
		const bool valid = i < m;
		const KT kk = valid ? K[base + i] : (KT)0;
		const bool ge = valid && kk >= pk, le = valid && kk <= pk;
		const u64 mL = __builtin_amdgcn_ballot_w64(ge), mR = __builtin_amdgcn_ballot_w64(le);
		if (ge) posL[cL + __popcll(mL & below)] = (PT)i;
		if (le) posR[cR + __popcll(mR & below)] = (PT)i;
		cL += __popcll(mL); cR += __popcll(mR);
	}
#undef FG_TILE
	wave_mem_fence();
	const int nPairs = cL < cR ? cL : cR;
	int nSwap = 0;
	for (int t2 = 0; t2 < (FG_SORT_ABLATED(64) ? 0 : nPairs); t2 += 64)
	{
		const int kq = t2 + lane;
		const bool valid = kq < nPairs;
		const int a = valid ? (int)posL[kq] : 0;
		const int b = valid ? (int)posR[cR - 1 - kq] : 0;
		const bool sw = valid && a < b;
		const u64 ms = __builtin_amdgcn_ballot_w64(sw);
		if (sw)
		{
			const KT xa = K[base + a], xb = K[base + b];
			const u32 ya = V[base + a], yb = V[base + b];
			K[base + a] = xb; V[base + a] = yb;
			K[base + b] = xa; V[base + b] = ya;
		}
		nSwap += __popcll(ms);
		if (ms != __builtin_amdgcn_ballot_w64(valid)) break;	// l_k < r_k is monotone in k
	}
	wave_mem_fence();
	const int lK1 = (nSwap < cL) ? (int)posL[nSwap] : 0x7fffffff;
	const int rK = (nSwap >= 1) ? (int)posR[cR - nSwap] : m;
	return uni(base + (lK1 < rK ? lK1 : rK));
}

// The same closed form without any serial chain, for one big segment in global memory.
// Phase 1 counts the stops per 64-element tile (running prefixes F_t, P_t of left / right
// stops, nothing else is stored, so the loads of many tiles are in flight together).  With
// f(p) = #left stops below p and g(p) = #right stops at or above p, l_k < r_k holds iff some
// p has f(p) >= k and g(p) >= k, hence K = max_p min(f(p), g(p)); f rises and g falls, so the
// maximum sits in the tile where they cross.  Phase 3 lists only the 2K stops that swap,
// phase 4 swaps pair k = (l_k, r_k); no iteration of either depends on another.
// sL / sR: >= (last - first) entries each (lists in front, tile prefixes behind the middle).
template <class KT, class VP = u32*>
__device__ __forceinline__ int partition_cf(KT* K, VP V, int first, int last, u32* sL, u32* sR)
{
	const int lane = threadIdx.x & 63;
	first = uni(first); last = uni(last);
	const int mid = first + (last - first) / 2;
	const KT ka = K[first + 1], kb = K[mid], kc = K[last - 1];
	const int m3 = uni(median3(ka, kb, kc));
	const int pick = m3 == 0 ? first + 1 : (m3 == 1 ? mid : last - 1);
	const KT pk = m3 == 0 ? ka : (m3 == 1 ? kb : kc);
	if (lane == 0)
	{
		const KT k0 = K[first]; const u32 v0 = V[first]; const u32 vp = V[pick];
		K[first] = pk; V[first] = vp;
		K[pick] = k0; V[pick] = v0;
	}
	wave_mem_fence();
	const int base = first + 1;
	const int m = last - base;
	const int T = (m + 63) >> 6;
	const KT* Kb = K + base;
	u32* pfL = sL + (m >> 1) + 2;	// the lists hold K <= m/2 entries (2K distinct positions)
	u32* pfR = sR + (m >> 1) + 2;

	// ---- phase 1: tile prefixes ----
	int totL = 0, totR = 0;
	for (int t0 = 0; t0 < T; t0 += 64)
	{
		int myL = 0, myR = 0;
		for (int tt = 0; tt < 64 && t0 + tt < T; tt += 8)
		{
			KT kk[8];
	#pragma unroll
			for (int u = 0; u < 8; ++u)
			{
				const int i = (t0 + tt + u) * 64 + lane;
				kk[u] = i < m ? Kb[i] : (KT)0;
			}
	#pragma unroll
			for (int u = 0; u < 8; ++u)
			{
				const int i = (t0 + tt + u) * 64 + lane;
				const bool valid = i < m;
				const u64 mL = __builtin_amdgcn_ballot_w64(valid && kk[u] >= pk);
				const u64 mR = __builtin_amdgcn_ballot_w64(valid && kk[u] <= pk);
				if (lane == tt + u) { myL = totL; myR = totR; }
				totL += __popcll(mL); totR += __popcll(mR);
			}
		}
		if (t0 + lane < T) { pfL[t0 + lane] = (u32)myL; pfR[t0 + lane] = (u32)myR; }
	}
	if (lane == 0) { pfL[T] = (u32)totL; pfR[T] = (u32)totR; }
	wave_mem_fence();

	// largest tile index t in [0, T] for which pred(t) holds (pred true on a prefix, true at 0)
	auto last_true = [&](auto pred) -> int
	{
		int res = 0;
		for (int t0 = 0; t0 <= T; t0 += 64)
		{
			const int t = t0 + lane;
			const int c = __popcll(__builtin_amdgcn_ballot_w64(t <= T && pred(t)));
			if (c > 0) res = t0 + c - 1;
			if (c < 64) break;
		}
		return res;
	};
	const u64 below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));

	// ---- phase 2: K and the cut ----
	int Kp;
	{
		const int ts = last_true([&](int t) { return (int)pfL[t] <= totR - (int)pfR[t]; });
		const int i = ts * 64 + lane;
		const bool valid = ts < T && i < m;
		const KT kk = valid ? Kb[i] : (KT)0;
		const u64 mL = __builtin_amdgcn_ballot_w64(valid && kk >= pk);
		const u64 mR = __builtin_amdgcn_ballot_w64(valid && kk <= pk);
		const int Ft = uni((int)pfL[ts]), Gt = totR - uni((int)pfR[ts]);
		const int f = Ft + __popcll(mL & below), g = Gt - __popcll(mR & below);
		int v = valid ? (f < g ? f : g) : 0;
		if (lane == 0 && ts < T)	// position just behind the tile
		{
			const int fn = (int)pfL[ts + 1], gn = totR - (int)pfR[ts + 1];
			const int vn = fn < gn ? fn : gn;
			// lane 0's own candidate is p = tile start: f = Ft <= Gt = g, value Ft, never above vn's
			// competitors' maximum unless it is the maximum itself; keep both
			v = v > vn ? v : vn;
		}
		for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o); v = v > w ? v : w; }
		Kp = uni(v);
	}
	int lK1 = 0x7fffffff, rK = m;
	if (Kp < totL)
	{
		const int tl = last_true([&](int t) { return t < T && (int)pfL[t] <= Kp; });
		const int i = tl * 64 + lane;
		const bool valid = i < m;
		const KT kk = valid ? Kb[i] : (KT)0;
		const u64 mL = __builtin_amdgcn_ballot_w64(valid && kk >= pk);
		lK1 = tl * 64 + nth_set_bit(mL, Kp - uni((int)pfL[tl]));
	}
	if (Kp >= 1)
	{
		const int idx = totR - Kp;	// 0-based index from the left of r_K among the right stops
		const int tr = last_true([&](int t) { return t < T && (int)pfR[t] <= idx; });
		const int i = tr * 64 + lane;
		const bool valid = i < m;
		const KT kk = valid ? Kb[i] : (KT)0;
		const u64 mR = __builtin_amdgcn_ballot_w64(valid && kk <= pk);
		rK = tr * 64 + nth_set_bit(mR, idx - uni((int)pfR[tr]));
	}
	const int cut = base + (lK1 < rK ? lK1 : rK);

	// ---- phase 3: the 2K stops that swap ----
	if (Kp > 0)
	{
		const int firstR = totR - Kp;
		for (int t0 = 0; t0 < T; t0 += 8)
		{
			KT kk[8]; int fl[8], fr[8];
	#pragma unroll
			for (int u = 0; u < 8; ++u)
			{
				const int t = t0 + u;
				const int i = t * 64 + lane;
				kk[u] = (t < T && i < m) ? Kb[i] : (KT)0;
				fl[u] = t < T ? (int)pfL[t] : 0x7fffffff;
				fr[u] = t < T ? (int)pfR[t + 1] : 0;
			}
	#pragma unroll
			for (int u = 0; u < 8; ++u)
			{
				const int t = t0 + u;
				const int i = t * 64 + lane;
				const bool valid = t < T && i < m;
				const int Fl = uni(fl[u]), Fr1 = uni(fr[u]);
				if (Fl < Kp)	// some left stop of this tile has rank < K
				{
					const bool ge = valid && kk[u] >= pk;
					const u64 mL = __builtin_amdgcn_ballot_w64(ge);
					const int r = Fl + __popcll(mL & below);
					if (ge && r < Kp) sL[r] = (u32)i;
				}
				if (Fr1 > firstR)	// some right stop of this tile is among the last K
				{
					const bool le = valid && kk[u] <= pk;
					const u64 mR = __builtin_amdgcn_ballot_w64(le);
					const int r = Fr1 - __popcll(mR) + __popcll(mR & below);
					if (le && r >= firstR) sR[r - firstR] = (u32)i;
				}
			}
		}
		wave_mem_fence();
		// ---- phase 4: swap pair k = (l_k, r_k), k = 1..K ----
		for (int k0 = 0; k0 < Kp; k0 += 128)
		{
			const int q0 = k0 + lane, q1 = k0 + 64 + lane;
			const bool s0 = q0 < Kp, s1 = q1 < Kp;
			const int a0 = s0 ? (int)sL[q0] : 0, b0 = s0 ? (int)sR[Kp - 1 - q0] : 0;
			const int a1 = s1 ? (int)sL[q1] : 0, b1 = s1 ? (int)sR[Kp - 1 - q1] : 0;
			KT xa0 = 0, xb0 = 0, xa1 = 0, xb1 = 0; u32 ya0 = 0, yb0 = 0, ya1 = 0, yb1 = 0;
			if (s0) { xa0 = Kb[a0]; xb0 = Kb[b0]; ya0 = V[base + a0]; yb0 = V[base + b0]; }
			if (s1) { xa1 = Kb[a1]; xb1 = Kb[b1]; ya1 = V[base + a1]; yb1 = V[base + b1]; }
			if (s0) { K[base + a0] = xb0; V[base + a0] = yb0; K[base + b0] = xa0; V[base + b0] = ya0; }
			if (s1) { K[base + a1] = xb1; V[base + a1] = yb1; K[base + b1] = xa1; V[base + b1] = ya1; }
		}
	}
	wave_mem_fence();
	return uni(cut);
}

// partition_cf by a whole workgroup of WW waves on one huge segment [0, n): the tiles of phases
// 1 and 3 and the pairs of phase 4 are dealt to the waves, phase 2 is computed by every wave.
// shm: >= 2 * WW ints of LDS.  All threads of the block must call.
template <class KT, int WW, class VP = u32*>
__device__ __forceinline__ int partition_cf_block(KT* K, VP V, int n, u32* sL, u32* sR, int* shm)
{
	const int lane = threadIdx.x & 63;
	const int wv = uni((int)(threadIdx.x >> 6));
	n = uni(n);
	const int mid = n / 2;
	const KT ka = K[1], kb = K[mid], kc = K[n - 1];
	const int m3 = uni(median3(ka, kb, kc));
	const int pick = m3 == 0 ? 1 : (m3 == 1 ? mid : n - 1);
	const KT pk = m3 == 0 ? ka : (m3 == 1 ? kb : kc);
	__syncthreads();	// every wave has read the candidates
	if (threadIdx.x == 0)
	{
		const KT k0 = K[0]; const u32 v0 = V[0]; const u32 vp = V[pick];
		K[0] = pk; V[0] = vp;
		K[pick] = k0; V[pick] = v0;
	}
	__syncthreads();
	const int base = 1;
	const int m = n - base;
	const int T = (m + 63) >> 6;
	const KT* Kb = K + base;
	u32* pfL = sL + (m >> 1) + 2;
	u32* pfR = sR + (m >> 1) + 2;

	// ---- phase 1: tile prefixes, a contiguous range of tiles per wave ----
	const int per = (((T + WW - 1) / WW) + 63) & ~63;
	const int tBeg = wv * per < T ? wv * per : T;
	const int tEnd = tBeg + per < T ? tBeg + per : T;
	int locL = 0, locR = 0;
	for (int t0 = tBeg; t0 < tEnd; t0 += 64)
	{
		int myL = 0, myR = 0;
		for (int tt = 0; tt < 64 && t0 + tt < tEnd; tt += 8)
		{
			KT kk[8];
	#pragma unroll
			for (int u = 0; u < 8; ++u)
			{
				const int i = (t0 + tt + u) * 64 + lane;
				kk[u] = (t0 + tt + u < tEnd && i < m) ? Kb[i] : (KT)0;
			}
	#pragma unroll
			for (int u = 0; u < 8; ++u)
			{
				const int i = (t0 + tt + u) * 64 + lane;
				const bool valid = t0 + tt + u < tEnd && i < m;
				const u64 mL = __builtin_amdgcn_ballot_w64(valid && kk[u] >= pk);
				const u64 mR = __builtin_amdgcn_ballot_w64(valid && kk[u] <= pk);
				if (lane == tt + u) { myL = locL; myR = locR; }
				locL += __popcll(mL); locR += __popcll(mR);
			}
		}
		if (t0 + lane < tEnd) { pfL[t0 + lane] = (u32)myL; pfR[t0 + lane] = (u32)myR; }
	}
	if (lane == 0) { shm[wv] = locL; shm[WW + wv] = locR; }
	__syncthreads();
	int offL = 0, offR = 0, totL = 0, totR = 0;
	for (int w = 0; w < WW; ++w)
	{
		const int a = shm[w], b = shm[WW + w];
		if (w < wv) { offL += a; offR += b; }
		totL += a; totR += b;
	}
	offL = uni(offL); offR = uni(offR); totL = uni(totL); totR = uni(totR);
	for (int t = tBeg + lane; t < tEnd; t += 64) { pfL[t] += (u32)offL; pfR[t] += (u32)offR; }
	if (threadIdx.x == 0) { pfL[T] = (u32)totL; pfR[T] = (u32)totR; }
	__syncthreads();

	auto last_true = [&](auto pred) -> int
	{
		int res = 0;
		for (int t0 = 0; t0 <= T; t0 += 64)
		{
			const int t = t0 + lane;
			const int c = __popcll(__builtin_amdgcn_ballot_w64(t <= T && pred(t)));
			if (c > 0) res = t0 + c - 1;
			if (c < 64) break;
		}
		return res;
	};
	const u64 below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));

	// ---- phase 2 (every wave): K and the cut ----
	int Kp;
	{
		const int ts = last_true([&](int t) { return (int)pfL[t] <= totR - (int)pfR[t]; });
		const int i = ts * 64 + lane;
		const bool valid = ts < T && i < m;
		const KT kk = valid ? Kb[i] : (KT)0;
		const u64 mL = __builtin_amdgcn_ballot_w64(valid && kk >= pk);
		const u64 mR = __builtin_amdgcn_ballot_w64(valid && kk <= pk);
		const int Ft = uni((int)pfL[ts]), Gt = totR - uni((int)pfR[ts]);
		const int f = Ft + __popcll(mL & below), g = Gt - __popcll(mR & below);
		int v = valid ? (f < g ? f : g) : 0;
		if (lane == 0 && ts < T)
		{
			const int fn = (int)pfL[ts + 1], gn = totR - (int)pfR[ts + 1];
			const int vn = fn < gn ? fn : gn;
			v = v > vn ? v : vn;
		}
		for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o); v = v > w ? v : w; }
		Kp = uni(v);
	}
	int lK1 = 0x7fffffff, rK = m;
	if (Kp < totL)
	{
		const int tl = last_true([&](int t) { return t < T && (int)pfL[t] <= Kp; });
		const int i = tl * 64 + lane;
		const bool valid = i < m;
		const KT kk = valid ? Kb[i] : (KT)0;
		const u64 mL = __builtin_amdgcn_ballot_w64(valid && kk >= pk);
		lK1 = tl * 64 + nth_set_bit(mL, Kp - uni((int)pfL[tl]));
	}
	if (Kp >= 1)
	{
		const int idx = totR - Kp;
		const int tr = last_true([&](int t) { return t < T && (int)pfR[t] <= idx; });
		const int i = tr * 64 + lane;
		const bool valid = i < m;
		const KT kk = valid ? Kb[i] : (KT)0;
		const u64 mR = __builtin_amdgcn_ballot_w64(valid && kk <= pk);
		rK = tr * 64 + nth_set_bit(mR, idx - uni((int)pfR[tr]));
	}
	const int cut = base + (lK1 < rK ? lK1 : rK);

	// ---- phase 3: the 2K stops that swap, groups of 8 tiles dealt round-robin ----
	if (Kp > 0)
	{
		const int firstR = totR - Kp;
		for (int t0 = wv * 8; t0 < T; t0 += 8 * WW)
		{
			KT kk[8]; int fl[8], fr[8];
	#pragma unroll
			for (int u = 0; u < 8; ++u)
			{
				const int t = t0 + u;
				const int i = t * 64 + lane;
				kk[u] = (t < T && i < m) ? Kb[i] : (KT)0;
				fl[u] = t < T ? (int)pfL[t] : 0x7fffffff;
				fr[u] = t < T ? (int)pfR[t + 1] : 0;
			}
	#pragma unroll
			for (int u = 0; u < 8; ++u)
			{
				const int t = t0 + u;
				const int i = t * 64 + lane;
				const bool valid = t < T && i < m;
				const int Fl = uni(fl[u]), Fr1 = uni(fr[u]);
				if (Fl < Kp)
				{
					const bool ge = valid && kk[u] >= pk;
					const u64 mL = __builtin_amdgcn_ballot_w64(ge);
					const int r = Fl + __popcll(mL & below);
					if (ge && r < Kp) sL[r] = (u32)i;
				}
				if (Fr1 > firstR)
				{
					const bool le = valid && kk[u] <= pk;
					const u64 mR = __builtin_amdgcn_ballot_w64(le);
					const int r = Fr1 - __popcll(mR) + __popcll(mR & below);
					if (le && r >= firstR) sR[r - firstR] = (u32)i;
				}
			}
		}
	}
	__syncthreads();
	// ---- phase 4: swap pair k = (l_k, r_k) ----
	for (int k0 = wv * 128; k0 < Kp; k0 += 128 * WW)
	{
		const int q0 = k0 + lane, q1 = k0 + 64 + lane;
		const bool s0 = q0 < Kp, s1 = q1 < Kp;
		const int a0 = s0 ? (int)sL[q0] : 0, b0 = s0 ? (int)sR[Kp - 1 - q0] : 0;
		const int a1 = s1 ? (int)sL[q1] : 0, b1 = s1 ? (int)sR[Kp - 1 - q1] : 0;
		KT xa0 = 0, xb0 = 0, xa1 = 0, xb1 = 0; u32 ya0 = 0, yb0 = 0, ya1 = 0, yb1 = 0;
		if (s0) { xa0 = Kb[a0]; xb0 = Kb[b0]; ya0 = V[base + a0]; yb0 = V[base + b0]; }
		if (s1) { xa1 = Kb[a1]; xb1 = Kb[b1]; ya1 = V[base + a1]; yb1 = V[base + b1]; }
		if (s0) { K[base + a0] = xb0; V[base + a0] = yb0; K[base + b0] = xa0; V[base + b0] = ya0; }
		if (s1) { K[base + a1] = xb1; V[base + a1] = yb1; K[base + b1] = xa1; V[base + b1] = ya1; }
	}
	__syncthreads();
	return uni(cut);
}

// Same partition, streamed: the two pointers advance in chunks of <= 64 from both ends
// (every element is read once, only swapped elements are written: ~18 B per element against
// ~30 B for the closed form), at the price of a serial dependence between the chunks.
// Used where enough independent pieces are in flight to hide that latency.
// The window [f, l) always holds untouched elements and the literal loop state after
// the swaps done so far; its last <= 63 elements are finished in registers.
template <class KT, class VP = u32*>
__device__ __forceinline__ int partition_stream(KT* K, VP V, int first, int last)
{
	const int lane = threadIdx.x & 63;
	first = uni(first); last = uni(last);
	const int mid = first + (last - first) / 2;
	const KT ka = K[first + 1], kb = K[mid], kc = K[last - 1];
	const int m3 = uni(median3(ka, kb, kc));
	const int pick = m3 == 0 ? first + 1 : (m3 == 1 ? mid : last - 1);
	const KT pk = m3 == 0 ? ka : (m3 == 1 ? kb : kc);
	wave_mem_fence();
	if (lane == 0)
	{
		const KT k0 = K[first]; const u32 v0 = V[first]; const u32 vp = V[pick];
		K[first] = pk; V[first] = vp;
		K[pick] = k0; V[pick] = v0;
	}
	wave_mem_fence();
	int f = first + 1, l = last;	// untouched window [f, l)
	// The loads of the next window are issued BEFORE the stores of the current one (they touch
	// disjoint elements: stores land in the consumed zones, loads come from the untouched
	// window): memory operations retire in order on this counter, so a load issued behind a
	// store would put the store's round trip on the serial chain as well.
	int wl = (l - f) / 2 < 64 ? (l - f) / 2 : 64;
	KT kL = 0, kR = 0; u32 vL = 0, vR = 0;
	if (l - f > 63 && lane < wl) { kL = K[f + lane]; vL = V[f + lane]; kR = K[l - 1 - lane]; vR = V[l - 1 - lane]; }
	while (l - f > 63)
	{
		const bool valid = lane < wl;
		const bool geL = valid && kL >= pk;
		const bool leR = valid && kR <= pk;
		const u64 mL = __builtin_amdgcn_ballot_w64(geL), mR = __builtin_amdgcn_ballot_w64(leR);
		const int cL = __popcll(mL), cR = __popcll(mR);
		const int m = cL < cR ? cL : cR;
		int nf = f, nl = l;
		int dstL = 0, dstR = 0;
		bool stL = false, stR = false;
		if (m > 0)
		{
			// stop lists through the LDS crossbar (as lane_partition): lane r learns the lane of
			// the r-th left stop and of the r-th right stop (counted from its window's outer end)
			const int rankL = __builtin_amdgcn_mbcnt_hi((u32)(mL >> 32), __builtin_amdgcn_mbcnt_lo((u32)mL, 0));
			const int rankR = __builtin_amdgcn_mbcnt_hi((u32)(mR >> 32), __builtin_amdgcn_mbcnt_lo((u32)mR, 0));
			const int idxL = __builtin_amdgcn_ds_permute((geL ? rankL : 63) << 2, lane);
			const int idxR = __builtin_amdgcn_ds_permute((leR ? rankR : 63) << 2, lane);
			stL = geL && rankL < m;
			stR = leR && rankR < m;
			dstL = l - 1 - __builtin_amdgcn_ds_bpermute((stL ? rankL : lane) << 2, idxR);
			dstR = f + __builtin_amdgcn_ds_bpermute((stR ? rankR : lane) << 2, idxL);
			const int lastL = __builtin_amdgcn_readlane(idxL, uni(m - 1)), lastR = __builtin_amdgcn_readlane(idxR, uni(m - 1));
			nf = f + lastL + 1;
			nl = l - 1 - lastR;
		}
		else
		{
			if (cL == 0) nf += wl;
			if (cR == 0) nl -= wl;
		}
		const int nwl = (nl - nf) / 2 < 64 ? (nl - nf) / 2 : 64;
		KT nkL = 0, nkR = 0; u32 nvL = 0, nvR = 0;
		if (nl - nf > 63 && lane < nwl) { nkL = K[nf + lane]; nvL = V[nf + lane]; nkR = K[nl - 1 - lane]; nvR = V[nl - 1 - lane]; }
		if (stL) { K[dstL] = kL; V[dstL] = vL; }
		if (stR) { K[dstR] = kR; V[dstR] = vR; }
		f = nf; l = nl; wl = nwl;
		kL = nkL; vL = nvL; kR = nkR; vR = nvR;
	}
	const int W = l - f;
	KT key = lane < W ? K[f + lane] : (KT)0;
	u32 val = lane < W ? V[f + lane] : 0u;
	const KT key0 = key; const u32 val0 = val;
	const int cutLane = lane_partition(key, val, 0, W, pk);
	if (lane < W && (key != key0 || val != val0)) { K[f + lane] = key; V[f + lane] = val; }
	wave_mem_fence();
	return uni(f + cutLane);
}


// std::sort(K[first0 .. first0+n), by key) with V carried along.  stk: >= 3*40 ints,
// sstk: >= 24 ints, posL/posR: >= n entries each, all private to the calling wave.
// depth0 >= 0 continues an introsort whose depth budget is already partly spent.
// All 64 lanes must call.
template <class KT, class PT>
__device__ __forceinline__ void wave_sort(KT* K, u32* V, int n, PT* posL, PT* posR, int* stk, int* sstk,
										  int first0 = 0, int depth0 = -1)
{
	n = uni(n); first0 = uni(first0); depth0 = uni(depth0);
	if (n < 2) return;
	const int lane = threadIdx.x & 63;
	int sp = 0;
	int first = first0, last = first0 + n, depth = depth0 >= 0 ? depth0 : 2 * fgsort::floor_log2_(n);
	while (true)
	{
		if (last - first <= 64)
		{
			if (last - first >= 2) sort_small(K, V, first, last - first, depth, sstk);
		}
		else if (depth == 0)
		{
			wave_mem_fence();
			if (lane == 0) { PtrAcc<KT> acc{K, V}; fgsort::heap_sort_(acc, first, last); }
			wave_mem_fence();
		}
		else
		{
			--depth;
			const int cut = partition_big(K, V, first, last, posL, posR);
			if (cut - first < last - cut)
			{
				stk[sp++] = cut; stk[sp++] = last; stk[sp++] = depth;
				last = cut;
			}
			else
			{
				stk[sp++] = first; stk[sp++] = cut; stk[sp++] = depth;
				first = cut;
			}
			continue;
		}
		if (sp == 0) break;
		depth = uni(stk[--sp]); last = uni(stk[--sp]); first = uni(stk[--sp]);
	}
	wave_mem_fence();
}

} // namespace wsort
