// Base-level divergence of the primary overlaps: exact unit-cost global edit distance of
// the two overlapping substrings, optionally homopolymer-compressed.
//
// Reference: getAlignmentErrEdlib (src/sequence/alignment.cpp:218-247) =
// homopolymerCompression (:52-70) of cur[curBegin, curBegin+curRange) and
// ext[extBegin, extBegin+extRange), then edlibAlign(NW, TASK_DISTANCE, k = -1)
// (src/sequence/edlib.cpp:141-296), whose result -- the exact global edit distance --
// does not depend on the algorithm.  Edlib is a banded Myers bit-vector DP; here the
// value is computed by the O(ND) furthest-reaching-point recurrence (Ukkonen /
// Landau-Vishkin), which fits a wave better for the common case (true overlaps, D small):
// one wave per overlap, one diagonal per lane, 32 bases compared per 64-bit XOR, both
// 2-bit packed (and compressed) strings staged in LDS.
#include "fg_ctx.h"

#include <algorithm>

namespace {

#define NEG_INF (-(1 << 29))

// base t of record `rec` (strand-aware, reference sequence.h:120-129)
__device__ __forceinline__ u32 base_at(const u64* __restrict__ w, i32 L, bool rc, i32 pos)
{
	const i32 p = rc ? L - 1 - pos : pos;
	const u32 b = (u32)((w[p >> 5] >> ((p & 31) * 2)) & 3);
	return rc ? (~b & 3) : b;
}

// extract [start, start+len) of a record into a packed 2-bit LDS string (zeroed before),
// dropping repeated bases when hpc; returns the packed length.  All 64 lanes call.
__device__ int extract_seq(const u64* __restrict__ w, i32 L, bool rc, i32 start, i32 len, bool hpc,
						   u32* __restrict__ dst)
{
	const int lane = threadIdx.x & 63;
	int outLen = 0;
	u32 carry = 4;	// base before the current chunk (4 = none)
	for (i32 t0 = 0; t0 < len; t0 += 64)
	{
		const i32 t = t0 + lane;
		const bool valid = t < len;
		const u32 b = valid ? base_at(w, L, rc, start + t) : 0u;
		u32 prev = __shfl_up(b, 1);
		if (lane == 0) prev = carry;
		const bool keep = valid && (!hpc || prev != b);
		const u64 m = __ballot(keep);
		const u64 below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
		if (keep)
		{
			const int o = outLen + __popcll(m & below);
			atomicOr(&dst[o >> 4], b << ((o & 15) * 2));
		}
		outLen += __popcll(m);
		carry = __shfl(b, 63);
	}
	return outLen;
}

// 32 bases (64 bits) of a packed string starting at base i; the string is followed by
// >= 2 zero words
__device__ __forceinline__ u64 get64(const u32* __restrict__ s, int i)
{
	const int w = i >> 4, sh = (i & 15) * 2;
	const u64 lo = (u64)s[w] | ((u64)s[w + 1] << 32);
	const u64 hi = (u64)s[w + 2];
	return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

// furthest row reachable from (i, i+d) along matches
__device__ __forceinline__ int snake(const u32* __restrict__ A, int n, const u32* __restrict__ B, int m, int i, int d)
{
	int j = i + d;
	while (i < n && j < m)
	{
		const u64 x = get64(A, i) ^ get64(B, j);
		int eq = x ? (__ffsll((long long)x) - 1) >> 1 : 32;
		const int lim = min(n - i, m - j);
		if (eq > lim) eq = lim;
		i += eq; j += eq;
		if (eq < 32) break;
	}
	return i;
}

// edit distance of A (n bases, rows) and B (m bases, columns) by furthest-reaching points.
// L0/L1: per-wave scratch of >= n + m + 5 ints each.  All 64 lanes call; result uniform.
__device__ int edit_distance_wave(const u32* __restrict__ A, int n, const u32* __restrict__ B, int m,
								  int* __restrict__ L0, int* __restrict__ L1)
{
	const int lane = threadIdx.x & 63;
	if (n == 0) return m;
	if (m == 0) return n;
	const int OFF = n + 2;		// index of diagonal d is d + OFF, d in [-n-1, m+1]
	const int target = m - n;
	int* prev = L0; int* cur = L1;
	// e = 0
	const int t0 = snake(A, n, B, m, 0, 0);
	if (target == 0 && t0 == n) return 0;
	if (lane == 0) { prev[OFF] = t0; prev[OFF - 1] = NEG_INF; prev[OFF + 1] = NEG_INF; }
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	for (int e = 1; ; ++e)
	{
		const int lo = max(-e, -n), hi = min(e, m);
		const int plo = max(-(e - 1), -n), phi = min(e - 1, m);	// diagonals written in round e-1
		bool found = false;
		for (int d0 = lo; d0 <= hi; d0 += 64)
		{
			const int d = d0 + lane;
			if (d <= hi)
			{
				// values of round e-1 exist on diagonals [-(e-1), e-1]; others count as -inf
				const int a = (d - 1 >= plo && d - 1 <= phi) ? prev[d - 1 + OFF] : NEG_INF;
				const int b = (d >= plo && d <= phi) ? prev[d + OFF] + 1 : NEG_INF;
				const int c = (d + 1 >= plo && d + 1 <= phi) ? prev[d + 1 + OFF] + 1 : NEG_INF;
				int t = max(a, max(b, c));
				t = min(t, min(n, m - d));
				if (t >= 0 && t + d >= 0) t = snake(A, n, B, m, t, d);
				cur[d + OFF] = t;
				if (d == target && t == n) found = true;
			}
		}
		if (__ballot(found)) return e;
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		int* tmp = prev; prev = cur; cur = tmp;
	}
}

__global__ void __launch_bounds__(64)
k_edit_distance(PrimRec* __restrict__ prims, u64 nPrim, const u32* __restrict__ query,
				const u64* __restrict__ qWords, const u64* __restrict__ qWordOff, const i32* __restrict__ qLen,
				const u64* __restrict__ words, const u64* __restrict__ wordOff, const i32* __restrict__ len,
				u32 firstId, int useHpc, int seqWords /* LDS u32 words per string */, int* __restrict__ scratch,
				u64 scratchPerWave)
{
	extern __shared__ u32 lds[];
	u32* A = lds;
	u32* B = lds + seqWords;
	const int lane = threadIdx.x;
	int* L0 = scratch + (u64)blockIdx.x * scratchPerWave;
	int* L1 = L0 + scratchPerWave / 2;
	for (u64 p = blockIdx.x; p < nPrim; p += gridDim.x)
	{
		const PrimRec r = prims[p];
		const u32 qrec = query[r.query];
		const u32 erec = r.extId - firstId;
		for (int i = lane; i < 2 * seqWords; i += 64) lds[i] = 0;
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		const i32 curRange = r.curEnd - r.curBegin, extRange = r.extEnd - r.extBegin;
		const int n = extract_seq(qWords + qWordOff[qrec >> 1], qLen[qrec >> 1], qrec & 1, r.curBegin, curRange, useHpc, A);
		const int m = extract_seq(words + wordOff[erec >> 1], len[erec >> 1], erec & 1, r.extBegin, extRange, useHpc, B);
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		const int dist = edit_distance_wave(A, n, B, m, L0, L1);
		if (lane == 0)
		{
			prims[p].editDistance = dist;
			prims[p].hpcLenCur = n;
			prims[p].hpcLenExt = m;
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	}
}

} // namespace

// fills editDistance / hpcLen* of nPrim device-resident primaries
void fgEditDistances(fg_ctx* c, PrimRec* dPrims, u64 nPrim, int useHpc)
{
	if (!nPrim) return;
	hipStream_t s = c->stream;
	const int maxLen = std::max(c->maxLen, c->hasQ ? c->qMaxLen : 0);
	const int seqWords = ((maxLen + 15) / 16 + 4 + 1) & ~1;	// + zero padding for get64
	const size_t ldsBytes = (size_t)2 * seqWords * 4;
	if (ldsBytes > 160 * 1024)
		throw FgError{FG_ERR_UNSUPPORTED, "reads longer than 320 kb are not supported by the edit-distance kernel yet"};
	if (ldsBytes > 64 * 1024)
		HIP_CHECK(hipFuncSetAttribute((const void*)k_edit_distance, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
	const unsigned grid = (unsigned)std::min<u64>(nPrim, 2048);
	const u64 perWave = 2 * ((u64)2 * maxLen + 8);
	c->dEditScratch.reserve((size_t)grid * perWave);
	ScopedK t(c->timer, "k_edit_distance");
	hipLaunchKernelGGL(k_edit_distance, grid, 64, ldsBytes, s, dPrims, nPrim, c->dQuery.p,
					   c->hasQ ? c->dQWords.p : c->dWords.p, c->hasQ ? c->dQWordOff.p : c->dWordOff.p,
					   c->hasQ ? c->dQLen.p : c->dLen.p, c->dWords.p, c->dWordOff.p,
					   c->dLen.p, c->firstId, useHpc, seqWords, c->dEditScratch.p, perWave);
}
