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

template <class KT>
struct KV { KT k; u32 v; };

template <class KT>
struct PtrAcc {
	typedef KV<KT> T;
	KT* K; u32* V;
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

// Hoare partition of one element per lane, lanes [lo,hi) take part; returns the cut
// lane; key/val are replaced by the lane's new element.
template <class KT>
__device__ __forceinline__ int lane_partition(KT& key, u32& val, int lo, int hi, KT pk)
{
	const int lane = threadIdx.x & 63;
	const bool in = lane >= lo && lane < hi;
	const bool ge = in && key >= pk;
	const bool le = in && key <= pk;
	const u64 mL = __ballot(ge), mR = __ballot(le);
	const int cL = __popcll(mL), cR = __popcll(mR);
	const u64 below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
	const u64 above = (lane == 63) ? 0ULL : (~0ULL << (lane + 1));
	const int rankL = __popcll(mL & below);
	const int rankR = __popcll(mR & above);
	const int partnerOfL = (ge && rankL < cR) ? nth_set_bit_desc(mR, rankL) : -1;
	const bool swapL = ge && partnerOfL > lane;
	const int K = __popcll(__ballot(swapL));
	int src = lane;
	if (swapL) src = partnerOfL;
	else if (le && rankR < K) src = nth_set_bit(mL, rankR);
	key = shflk(key, src);
	val = __shfl(val, src);
	const int lK1 = (cL > K) ? nth_set_bit(mL, K) : 0x7fffffff;
	const int rK = (K >= 1) ? nth_set_bit_desc(mR, K - 1) : hi;
	return lK1 < rK ? lK1 : rK;
}

// segment of <= 64 elements entirely in registers: quicksort phase, then the final
// insertion sort as a stable rank.  stk: >= 24 ints private to the wave.
template <class KT>
__device__ __forceinline__ void sort_small(KT* K, u32* V, int first, int n, int depth, int* stk)
{
	const int lane = threadIdx.x & 63;
	KT key = lane < n ? K[first + lane] : (KT)0;
	u32 val = lane < n ? V[first + lane] : 0u;
	int sp = 0;
	int a = 0, b = n, d = depth;
	while (true)
	{
		while (b - a > 16)
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
			const int mid = a + (b - a) / 2;
			const KT ka = shflk(key, a + 1), kb = shflk(key, mid), kc = shflk(key, b - 1);
			const int m3 = median3(ka, kb, kc);
			const int pick = m3 == 0 ? a + 1 : (m3 == 1 ? mid : b - 1);
			const KT pk = m3 == 0 ? ka : (m3 == 1 ? kb : kc);
			const int src = lane == a ? pick : (lane == pick ? a : lane);
			key = shflk(key, src);
			val = __shfl(val, src);
			const int cut = lane_partition(key, val, a + 1, b, pk);
			if (cut - a < b - cut) { stk[sp++] = cut; stk[sp++] = b; stk[sp++] = d; b = cut; }
			else { stk[sp++] = a; stk[sp++] = cut; stk[sp++] = d; a = cut; }
		}
		if (sp == 0) break;
		d = stk[--sp]; b = stk[--sp]; a = stk[--sp];
	}
	int rank = 0;
	for (int j = 0; j < n; ++j)
	{
		const KT kj = shflk(key, j);
		rank += (kj < key) || (kj == key && j < lane);
	}
	wave_mem_fence();
	if (lane < n) { K[first + rank] = key; V[first + rank] = val; }
}

// segment of > 64 elements in memory: pivot to first, the partition streamed in
// chunks of <= 64 from both ends, the last <= 64 untouched elements in registers
template <class KT>
__device__ __forceinline__ int partition_big(KT* K, u32* V, int first, int last)
{
	const int lane = threadIdx.x & 63;
	const int mid = first + (last - first) / 2;
	const KT ka = K[first + 1], kb = K[mid], kc = K[last - 1];
	const int m3 = median3(ka, kb, kc);
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
	while (l - f > 64)
	{
		const int W = l - f;
		const int wl = W / 2 < 64 ? W / 2 : 64;
		const bool valid = lane < wl;
		const int iL = f + lane, iR = l - 1 - lane;
		KT kL = 0, kR = 0; u32 vL = 0, vR = 0;
		if (valid) { kL = K[iL]; vL = V[iL]; kR = K[iR]; vR = V[iR]; }
		const bool geL = valid && kL >= pk;
		const bool leR = valid && kR <= pk;
		const u64 mL = __ballot(geL), mR = __ballot(leR);
		const int cL = __popcll(mL), cR = __popcll(mR);
		const int m = cL < cR ? cL : cR;
		if (m > 0)
		{
			const u64 below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
			const int rankL = __popcll(mL & below), rankR = __popcll(mR & below);
			if (geL && rankL < m)
			{
				const int dst = l - 1 - nth_set_bit(mR, rankL);
				K[dst] = kL; V[dst] = vL;
			}
			if (leR && rankR < m)
			{
				const int dst = f + nth_set_bit(mL, rankR);
				K[dst] = kR; V[dst] = vR;
			}
			const int lastL = nth_set_bit(mL, m - 1), lastR = nth_set_bit(mR, m - 1);
			f = f + lastL + 1;
			l = l - 1 - lastR;
		}
		else
		{
			if (cL == 0) f += wl;
			if (cR == 0) l -= wl;
		}
	}
	const int W = l - f;
	KT key = lane < W ? K[f + lane] : (KT)0;
	u32 val = lane < W ? V[f + lane] : 0u;
	const KT key0 = key; const u32 val0 = val;
	const int cutLane = lane_partition(key, val, 0, W, pk);
	if (lane < W && (key != key0 || val != val0)) { K[f + lane] = key; V[f + lane] = val; }
	wave_mem_fence();
	return f + cutLane;
}

// std::sort(K[0..n), by key) with V carried along.  stk: >= 3*40 ints, sstk: >= 24
// ints, both private to the calling wave.  All 64 lanes must call.
template <class KT>
__device__ __forceinline__ void wave_sort(KT* K, u32* V, int n, int* stk, int* sstk)
{
	if (n < 2) return;
	const int lane = threadIdx.x & 63;
	int sp = 0;
	int first = 0, last = n, depth = 2 * fgsort::floor_log2_(n);
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
			const int cut = partition_big(K, V, first, last);
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
		depth = stk[--sp]; last = stk[--sp]; first = stk[--sp];
	}
	wave_mem_fence();
}

} // namespace wsort
