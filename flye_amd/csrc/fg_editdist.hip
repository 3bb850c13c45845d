// Base-level divergence of the primary overlaps: exact unit-cost global edit distance of
// the two overlapping substrings, optionally homopolymer-compressed.
//
// Reference: getAlignmentErrEdlib (src/sequence/alignment.cpp:218-247) =
// homopolymerCompression (:52-70) of cur[curBegin, curBegin+curRange) and
// ext[extBegin, extBegin+extRange), then edlibAlign(NW, TASK_DISTANCE, k = -1)
// (src/sequence/edlib.cpp:141-296), whose result -- the exact global edit distance --
// does not depend on the algorithm.  Two kernels, both with a bounded cost:
//
//   k_edit_ond    one wave per overlap, both strings staged in LDS as bit planes: the O(ND)
//                 furthest-reaching-point recurrence (Ukkonen / Landau-Vishkin), one diagonal per
//                 lane, 64 bases compared per XOR.  Right for true overlaps (D small).  It gives
//                 up after ED_EMAX rounds (or at once for substrings longer than the LDS piece)
//                 and queues the pair for
//   k_edit_myers  Myers' bit-vector recurrence (Hyyro's block form, the algorithm edlib itself
//                 runs, edlib.cpp:194-212 with k = 64, 128, ...) over an Ukkonen band that doubles
//                 until the distance fits: lane L owns the 64-row block L of a 4096-row strip, the
//                 lanes sweep the strip's columns skewed by one (lane L is at column t - L in step
//                 t), the horizontal delta and the text character travel from lane to lane by DPP,
//                 the strip's bottom-row deltas stay in a global bit-plane buffer for the strip
//                 below.  Large pairs get a workgroup of ED_BIG_WAVES waves that sweep consecutive
//                 strips concurrently, each one a few hundred columns behind the strip above.
//
// Strings are held as BIT PLANES: per 64 bases one u64 of low bits and one u64 of high bits
// -- exactly the form the bit-vector recurrence needs for its match masks, and 64 bases per
// compare for the O(ND) snakes.
#include "fg_ctx.h"

#include <algorithm>

#ifndef ED_EMAX
#define ED_EMAX 512			// O(ND) rounds before the bit-vector path takes over
#endif
#define ED_LDS_BASES 32768	// longest substring k_edit_ond stages in LDS (2 x 8 KB per wave)
#define ED_STRIP 4096		// rows per strip = 64 lanes x 64-row blocks
#define ED_BIG_MIN 49152	// pairs with a longer substring get a multi-wave workgroup
#define ED_BIG_WAVES 8
#define ED_CHUNK_BLOCKS 8	// multi-wave: 64-column blocks per round between two barriers

namespace {

#define NEG_INF (-(1 << 29))

struct EdCounters { u32 nSmall, nBig; unsigned long long maxWordsSmall, maxWordsBig; };

// base t of record `rec` (strand-aware, reference sequence.h:120-129)
__device__ __forceinline__ u32 base_at(const u64* __restrict__ w, i32 L, bool rc, i32 pos)
{
	const i32 p = rc ? L - 1 - pos : pos;
	const u32 b = (u32)((w[p >> 5] >> ((p & 31) * 2)) & 3);
	return rc ? (~b & 3) : b;
}

// u64 words a pair needs in the bit-vector kernel's slab: both strings as planes (+ 2 zero
// blocks each) and the two delta planes of the text
__host__ __device__ __forceinline__ u64 ed_words_a(i32 curRange) { return 2ULL * ((u64)(curRange + 63) / 64 + 2); }
__host__ __device__ __forceinline__ u64 ed_words_b(i32 extRange) { return 2ULL * ((u64)(extRange + 63) / 64 + 2); }
__host__ __device__ __forceinline__ u64 ed_words_h(i32 extRange) { return 2ULL * ((u64)(extRange + 63) / 64 + 2); }

// [start, start+len) of a record -> bit planes at dst (LDS or global): dst[2b] = low bits, dst[2b+1]
// = high bits of bases 64b .. 64b+63; repeated bases dropped when hpc (alignment.cpp:52-70).  The
// kept bases of each 64-base chunk are pushed to the low lanes through the LDS crossbar, two
// ballots turn them into plane words, and the words are appended in scalar registers -- no atomics.
// Writes (returned length + 63) / 64 blocks and two zero blocks behind them.  All 64 lanes call.
__device__ int extract_planes(const u64* __restrict__ w, i32 L, bool rc, i32 start, i32 len, bool hpc,
							  u64* __restrict__ dst)
{
	const int lane = threadIdx.x & 63;
	int outLen = 0, ob = 0, fill = 0;
	u64 acc0 = 0, acc1 = 0;
	u32 carry = 4;	// base before the current chunk (4 = none)
	for (i32 t0 = 0; t0 < len; t0 += 64)
	{
		const i32 t = t0 + lane;
		const bool valid = t < len;
		const u32 b = valid ? base_at(w, L, rc, start + t) : 0u;
		u32 prev = __shfl_up(b, 1);
		if (lane == 0) prev = carry;
		const bool keep = valid && (!hpc || prev != b);
		const u64 m = __builtin_amdgcn_ballot_w64(keep);
		const int cnt = __popcll(m);
		const int rank = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0));
		// kept lanes push their base to lane `rank`; the others to lane 63 + (never read: cnt <= 64
		// and a lane only counts below cnt).  With cnt == 64 every lane keeps and nothing is dumped.
		const u32 cb = (u32)__builtin_amdgcn_ds_permute((keep ? rank : 63) << 2, (int)b);
		const bool in = lane < cnt;
		// with cnt < 64 lane 63 may hold a dumped value, but 63 >= cnt: masked by `in`
		const u64 p0 = __builtin_amdgcn_ballot_w64(in && (cb & 1u)), p1 = __builtin_amdgcn_ballot_w64(in && (cb & 2u));
		if (cnt)
		{
			acc0 |= p0 << fill; acc1 |= p1 << fill;
			if (fill + cnt >= 64)
			{
				if (lane == 0) { dst[2 * ob] = acc0; dst[2 * ob + 1] = acc1; }
				++ob;
				acc0 = fill ? p0 >> (64 - fill) : 0ULL;
				acc1 = fill ? p1 >> (64 - fill) : 0ULL;
				fill = fill + cnt - 64;
			}
			else fill += cnt;
		}
		outLen += cnt;
		carry = __shfl(b, 63);
	}
	if (lane == 0)
	{
		if (fill) { dst[2 * ob] = acc0; dst[2 * ob + 1] = acc1; ++ob; }
		dst[2 * ob] = 0; dst[2 * ob + 1] = 0; dst[2 * ob + 2] = 0; dst[2 * ob + 3] = 0;
	}
	return outLen;
}

// 64 bases of both planes starting at base i (the string is followed by >= 1 zero block)
__device__ __forceinline__ void fetch_planes(const u64* __restrict__ P, int i, u64& a0, u64& a1)
{
	const int wq = i >> 6, sh = i & 63;
	const u64 l0 = P[2 * wq], l1 = P[2 * wq + 1], h0 = P[2 * wq + 2], h1 = P[2 * wq + 3];
	a0 = sh ? (l0 >> sh) | (h0 << (64 - sh)) : l0;
	a1 = sh ? (l1 >> sh) | (h1 << (64 - sh)) : l1;
}

// furthest row reachable from (i, i+d) along matches
__device__ __forceinline__ int snake(const u64* __restrict__ A, int n, const u64* __restrict__ B, int m, int i, int d)
{
	int j = i + d;
	while (i < n && j < m)
	{
		u64 a0, a1, b0, b1;
		fetch_planes(A, i, a0, a1);
		fetch_planes(B, j, b0, b1);
		const u64 x = (a0 ^ b0) | (a1 ^ b1);
		int eq = x ? (__ffsll((long long)x) - 1) : 64;
		const int lim = min(n - i, m - j);
		if (eq > lim) eq = lim;
		i += eq; j += eq;
		if (eq < 64) break;
	}
	return i;
}

// Edit distance of A (n bases, rows) and B (m bases, columns) by furthest-reaching points, or -1
// when it exceeds eMax.  Diagonal d lives at index d + eMax + 2 of L0 / L1 (2 * eMax + 5 ints
// each, private to the wave).  All 64 lanes call; result uniform.
__device__ int edit_distance_ond(const u64* __restrict__ A, int n, const u64* __restrict__ B, int m,
								 int* __restrict__ L0, int* __restrict__ L1, int eMax)
{
	const int lane = threadIdx.x & 63;
	const int OFF = eMax + 2;
	const int target = m - n;
	if (target > eMax || -target > eMax) return -1;
	int* prev = L0; int* cur = L1;
	// e = 0
	const int t0 = snake(A, n, B, m, 0, 0);
	if (target == 0 && t0 == n) return 0;
	if (lane == 0) { prev[OFF] = t0; prev[OFF - 1] = NEG_INF; prev[OFF + 1] = NEG_INF; }
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	for (int e = 1; e <= eMax; ++e)
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
		if (__builtin_amdgcn_ballot_w64(found)) return e;
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		int* tmp = prev; prev = cur; cur = tmp;
	}
	return -1;
}

struct EdSeqs {
	const u32* query;
	const u64* qWords; const u64* qWordOff; const i32* qLen;
	const u64* words; const u64* wordOff; const i32* len;
	u32 firstId;
};

// ---- phase 1: O(ND), bounded --------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_edit_ond(PrimRec* __restrict__ prims, u64 nPrim, EdSeqs S, int useHpc, int ldsBlocks /* 16 B blocks per string */,
		   int eMax, int* __restrict__ scratch, u32* __restrict__ listSmall, u32* __restrict__ listBig,
		   EdCounters* __restrict__ cnt)
{
	extern __shared__ __attribute__((aligned(16))) u64 edLds[];
	u64* A = edLds;
	u64* B = edLds + 2 * (size_t)ldsBlocks;
	const int lane = threadIdx.x;
	const int perArr = 2 * eMax + 5;
	int* L0 = scratch + (u64)blockIdx.x * 2 * perArr;
	int* L1 = L0 + perArr;
	const i32 capBases = (ldsBlocks - 2) * 64;
	for (u64 p = blockIdx.x; p < nPrim; p += gridDim.x)
	{
		const PrimRec r = prims[p];
		const u32 qrec = S.query[r.query];
		const u32 erec = r.extId - S.firstId;
		const i32 curRange = r.curEnd - r.curBegin, extRange = r.extEnd - r.extBegin;
		int dist = -1, n = 0, m = 0, k0 = 64;
		if (max(curRange, extRange) <= capBases)
		{
			n = extract_planes(S.qWords + S.qWordOff[qrec >> 1], S.qLen[qrec >> 1], qrec & 1, r.curBegin, curRange, useHpc, A);
			m = extract_planes(S.words + S.wordOff[erec >> 1], S.len[erec >> 1], erec & 1, r.extBegin, extRange, useHpc, B);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			if (n == 0 || m == 0) dist = max(n, m);
			else dist = edit_distance_ond(A, n, B, m, L0, L1, eMax);
			k0 = 2 * eMax;	// the distance is known to exceed eMax
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		}
		if (lane == 0)
		{
			if (dist >= 0)
			{
				prims[p].editDistance = dist;
				prims[p].hpcLenCur = n;
				prims[p].hpcLenExt = m;
			}
			else
			{
				const unsigned long long wordsNeeded = ed_words_a(curRange) + ed_words_b(extRange) + ed_words_h(extRange);
				const bool big = max(curRange, extRange) > ED_BIG_MIN;
				atomicMax(big ? &cnt->maxWordsBig : &cnt->maxWordsSmall, wordsNeeded);
				const u32 slot = atomicAdd(big ? &cnt->nBig : &cnt->nSmall, 1u);
				(big ? listBig : listSmall)[slot] = (u32)p;
				prims[p].editDistance = -k0;	// where the band doubling starts
			}
		}
	}
}

// ---- phase 2: banded bit-vector recurrence -----------------------------------------------------
// lane L <- lane L-1, lane 0 <- fill (DPP wave_shr:1)
__device__ __forceinline__ u32 ed_shr1(u32 v, u32 fill)
{
	return (u32)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ u32 ed_lane32(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }

// state of one wave's sweep over one strip
struct EdStrip {
	int s;				// strip index
	int L;				// 64-row blocks (lanes) in the strip
	int cl, W;			// first column (multiple of 64), number of columns
	int prevCh;			// columns below this carry the deltas of the strip above; others count as +1
	int nextCl;			// hout of columns below this are part of the score's path along the bottom row
	int t;				// next step
	int sum;			// rows + sum of hout over [cl, nextCl)
	u32 PvL, PvH, MvL, MvH, B0L, B0H, B1L, B1H, pipe;
	u64 outP, outM;		// bottom-row deltas of the current 64-column block
};

__device__ __forceinline__ void ed_strip_bounds(int s, int n, int m, int dlo, int dhi, int& cl, int& ch)
{
	const int r0 = s * ED_STRIP;
	const int r1 = min(n, r0 + ED_STRIP);
	cl = max(0, r0 + dlo) & ~63;
	ch = min(m, r1 + dhi);
	ch = min(m, (ch + 63) & ~63);
}

__device__ __forceinline__ void ed_strip_begin(EdStrip& st, int s, const u64* __restrict__ gA, int n, int m, int dlo, int dhi)
{
	const int lane = threadIdx.x & 63;
	const int r0 = s * ED_STRIP;
	const int rows = min(n - r0, ED_STRIP);
	st.s = s;
	st.L = (rows + 63) >> 6;
	int ch;
	ed_strip_bounds(s, n, m, dlo, dhi, st.cl, ch);
	st.W = ch - st.cl;
	st.prevCh = 0;
	if (s > 0) { int pcl; ed_strip_bounds(s - 1, n, m, dlo, dhi, pcl, st.prevCh); }
	st.nextCl = m;
	if (r0 + ED_STRIP < n) { int nch; ed_strip_bounds(s + 1, n, m, dlo, dhi, st.nextCl, nch); }
	st.t = 0;
	st.sum = st.L * 64;		// the left boundary: a column of +1 steps over the (padded) rows
	const u64 b0 = lane < st.L ? gA[2 * ((size_t)(r0 >> 6) + lane)] : 0ULL;
	const u64 b1 = lane < st.L ? gA[2 * ((size_t)(r0 >> 6) + lane) + 1] : 0ULL;
	st.B0L = (u32)b0; st.B0H = (u32)(b0 >> 32); st.B1L = (u32)b1; st.B1H = (u32)(b1 >> 32);
	st.PvL = ~0u; st.PvH = ~0u; st.MvL = 0; st.MvH = 0; st.pipe = 0;
	st.outP = 0; st.outM = 0;
}

// Steps [st.t, st.t + 64) of the strip (one 64-column block for lane 0); returns true when the
// strip is finished.  gB: text planes; gH: delta planes (plus, minus) per 64-column block, read
// for the block lane 0 enters and written for the block the bottom lane leaves.
template <bool SHARED_H>
__device__ __forceinline__ bool ed_strip_block(EdStrip& st, const u64* __restrict__ gB, u64* __restrict__ gH)
{
	const int lane = threadIdx.x & 63;
	const int T = st.W + st.L - 1;
	const int tEnd = min(T, st.t + 64);
	// inputs of lane 0 for this block of columns
	u64 bw0 = 0, bw1 = 0, hp = ~0ULL, hm = 0;
	if (st.t < st.W)
	{
		const size_t blk = (size_t)(st.cl + st.t) >> 6;
		bw0 = fg_uni(gB[2 * blk]); bw1 = fg_uni(gB[2 * blk + 1]);
		if (st.cl + st.t < st.prevCh)
		{
			if (SHARED_H)
			{
				hp = fg_uni((u64)__hip_atomic_load(&gH[2 * blk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
				hm = fg_uni((u64)__hip_atomic_load(&gH[2 * blk + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
			}
			else { hp = fg_uni(gH[2 * blk]); hm = fg_uni(gH[2 * blk + 1]); }
		}
	}
	const int bl = st.L - 1;
	for (int t = st.t; t < tEnd; ++t)
	{
		const int bit = t & 63;
		const u32 fill = (u32)((bw0 >> bit) & 1) | ((u32)((bw1 >> bit) & 1) << 1) | ((u32)((hp >> bit) & 1) << 2) |
						 ((u32)((hm >> bit) & 1) << 3);
		const u32 in = ed_shr1(st.pipe, fill);
		const int j = t - lane;
		if ((u32)j < (u32)st.W && lane < st.L)
		{
			const u32 c0 = (u32)((i32)(in << 31) >> 31), c1 = (u32)((i32)(in << 30) >> 31);
			const u32 hpIn = (in >> 2) & 1u, hmIn = (in >> 3) & 1u;
			u32 eqL = ~(st.B0L ^ c0) & ~(st.B1L ^ c1);
			const u32 eqH = ~(st.B0H ^ c0) & ~(st.B1H ^ c1);
			const u32 xvL = eqL | st.MvL, xvH = eqH | st.MvH;
			eqL |= hmIn;
			const u64 pv = ((u64)st.PvH << 32) | st.PvL;
			const u64 sumv = ((((u64)(eqH & st.PvH)) << 32) | (eqL & st.PvL)) + pv;
			const u32 xhL = ((u32)sumv ^ st.PvL) | eqL, xhH = ((u32)(sumv >> 32) ^ st.PvH) | eqH;
			u32 phL = st.MvL | ~(xhL | st.PvL), phH = st.MvH | ~(xhH | st.PvH);
			u32 mhL = st.PvL & xhL, mhH = st.PvH & xhH;
			const u32 hpo = phH >> 31, hmo = mhH >> 31;
			phH = (phH << 1) | (phL >> 31); phL = (phL << 1) | hpIn;
			mhH = (mhH << 1) | (mhL >> 31); mhL = (mhL << 1) | hmIn;
			st.PvL = mhL | ~(xvL | phL); st.PvH = mhH | ~(xvH | phH);
			st.MvL = phL & xvL; st.MvH = phH & xvH;
			st.pipe = (in & 3u) | (hpo << 2) | (hmo << 3);
		}
		// what the bottom lane produced in this step: column jb of the strip's bottom row
		const int jb = t - bl;
		if (jb >= 0)	// jb < W holds for every t < T
		{
			const u32 x = ed_lane32(st.pipe, bl);
			const u32 op = (x >> 2) & 1u, om = (x >> 3) & 1u;
			const int col = st.cl + jb;
			if (col < st.nextCl) st.sum += (int)op - (int)om;
			st.outP |= (u64)op << (col & 63); st.outM |= (u64)om << (col & 63);
			if ((col & 63) == 63 || jb == st.W - 1)
			{
				if (lane == 0)
				{
					if (SHARED_H)
					{
						__hip_atomic_store(&gH[2 * ((size_t)col >> 6)], st.outP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						__hip_atomic_store(&gH[2 * ((size_t)col >> 6) + 1], st.outM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
					else { gH[2 * ((size_t)col >> 6)] = st.outP; gH[2 * ((size_t)col >> 6) + 1] = st.outM; }
				}
				st.outP = 0; st.outM = 0;
			}
		}
	}
	st.t = tEnd;
	return tEnd >= T;
}

// columns of the strip's bottom row that are stored (exclusive bound, absolute)
__device__ __forceinline__ int ed_strip_stored(const EdStrip& st)
{
	const int jb = st.t - (st.L - 1);	// columns [0, jb) of the strip have left the bottom lane
	if (jb >= st.W) return st.cl + st.W;
	return jb <= 0 ? st.cl : ((st.cl + jb) & ~63);
}

// vertical deltas of the padded rows below row n in the strip's last column (bottom lane's state)
__device__ __forceinline__ int ed_pad_correction(const EdStrip& st, int n)
{
	const int pad = st.s * ED_STRIP + st.L * 64 - n;	// 0..63
	if (pad == 0) return 0;
	const u64 pv = ((u64)ed_lane32(st.PvH, st.L - 1) << 32) | ed_lane32(st.PvL, st.L - 1);
	const u64 mv = ((u64)ed_lane32(st.MvH, st.L - 1) << 32) | ed_lane32(st.MvL, st.L - 1);
	const u64 mask = ~0ULL << (64 - pad);
	return __popcll(pv & mask) - __popcll(mv & mask);
}

// Upper bound D' >= D of the edit distance of A (n rows) and B (m columns), equal to D whenever
// D <= k (k >= |m - n|): only cells on diagonals a path of cost <= k can visit are computed
// (dlo <= J - i <= dhi), whatever lies outside a strip's column range counts as reached by +1
// steps from the computed region.  NW waves (the whole block) call; NW > 1 sweeps strips
// w, w + NW, ... on wave w, each following the strip above at a distance the shared progress
// words enforce.  shm: 2 * NW + 1 ints.
template <int NW>
__device__ int ed_myers_band(const u64* __restrict__ gA, int n, const u64* __restrict__ gB, int m,
							 u64* __restrict__ gH, int k, int* shm)
{
	const int lane = threadIdx.x & 63;
	const int wv = fg_uni((i32)(threadIdx.x >> 6));
	const int delta = m - n;
	const int dhi = (k + delta) / 2, dlo = -((k - delta) / 2);
	const int S = (n + ED_STRIP - 1) / ED_STRIP;
	int total = 0;
	EdStrip st;
	if (NW == 1)
	{
		for (int s = 0; s < S; ++s)
		{
			ed_strip_begin(st, s, gA, n, m, dlo, dhi);
			while (!ed_strip_block<false>(st, gB, gH)) {}
			total += st.sum;
			if (s == S - 1) total -= ed_pad_correction(st, n);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");	// the next strip reads what lane 0 stored
		}
		return total;
	}
	int* sStrip = shm; int* sProg = shm + NW; int* sTotal = shm + 2 * NW;
	int s = wv;
	bool have = s < S;
	if (have) ed_strip_begin(st, s, gA, n, m, dlo, dhi);
	if (threadIdx.x == 0) *sTotal = 0;
	// every round the wave with the lowest strip in flight advances by a block at least: the bound is
	// never reached by a correct schedule, it only turns a scheduling bug into a wrong answer (-1, which
	// the tests catch) instead of a hung GPU
	long long roundsLeft = 4LL * ((long long)S + 4) * ((long long)(m >> 6) + ED_STRIP / 64 + 4);
	while (true)
	{
		if (--roundsLeft < 0) { __syncthreads(); return -1; }
		if (lane == 0) { sStrip[wv] = have ? s : 0x7fffffff; sProg[wv] = have ? ed_strip_stored(st) : 0; }
		__syncthreads();
		bool any = false;
		for (int w = 0; w < NW; ++w) any |= sStrip[w] != 0x7fffffff;
		if (!any) break;
		int limit = 0x7fffffff;		// lane 0 may enter blocks that end at or below this column
		if (have && s > 0)
		{
			const int pw = (wv + NW - 1) % NW;
			const int ps = fg_uni(sStrip[pw]);
			if (ps == s - 1) limit = fg_uni(sProg[pw]);
			else if (ps < s - 1) limit = 0;
		}
		__syncthreads();
		if (have)
		{
			for (int b = 0; b < ED_CHUNK_BLOCKS; ++b)
			{
				// the block lane 0 enters next needs the strip above up to its end -- unless it lies
				// beyond that strip's range (then it counts as +1) or lane 0 has no columns left
				if (st.t < st.W && st.cl + st.t < st.prevCh)
				{
					const int need = min(st.cl + st.t + 64, st.prevCh);
					if (need > limit) break;
				}
				if (ed_strip_block<true>(st, gB, gH))
				{
					int add = st.sum;
					if (s == S - 1) add -= ed_pad_correction(st, n);
					if (lane == 0) atomicAdd(sTotal, add);
					s += NW;
					have = s < S;
					if (have) ed_strip_begin(st, s, gA, n, m, dlo, dhi);
					break;	// the new strip's dependency is looked at in the next round
				}
			}
		}
	}
	total = *sTotal;
	__syncthreads();
	return total;
}

template <int NW>
__global__ void __launch_bounds__(NW * 64)
k_edit_myers(PrimRec* __restrict__ prims, const u32* __restrict__ list, u32 nList, EdSeqs S, int useHpc,
			 u64* __restrict__ slabs, u64 slabWords)
{
	__shared__ int shm[2 * NW + 1];
	__shared__ int sLen[2];
	const int wv = threadIdx.x >> 6;
	u64* slab = slabs + (size_t)blockIdx.x * slabWords;
	for (u32 li = blockIdx.x; li < nList; li += gridDim.x)
	{
		const u32 p = list[li];
		const PrimRec r = prims[p];
		const u32 qrec = S.query[r.query];
		const u32 erec = r.extId - S.firstId;
		const i32 curRange = r.curEnd - r.curBegin, extRange = r.extEnd - r.extBegin;
		u64* gA = slab;
		u64* gB = gA + ed_words_a(curRange);
		u64* gH = gB + ed_words_b(extRange);
		if (wv == 0)
		{
			const int n = extract_planes(S.qWords + S.qWordOff[qrec >> 1], S.qLen[qrec >> 1], qrec & 1, r.curBegin, curRange, useHpc, gA);
			if (threadIdx.x == 0) sLen[0] = n;
		}
		if (wv == (NW > 1 ? 1 : 0))
		{
			const int m = extract_planes(S.words + S.wordOff[erec >> 1], S.len[erec >> 1], erec & 1, r.extBegin, extRange, useHpc, gB);
			if ((threadIdx.x & 63) == 0) sLen[1] = m;
		}
		__syncthreads();
		const int n = sLen[0], m = sLen[1];
		int dist;
		if (n == 0 || m == 0) dist = max(n, m);
		else
		{
			long long k = max(-r.editDistance, 1);	// k_edit_ond left -k0 there
			k = max(k, (long long)abs(m - n));
			while (true)
			{
				const int kk = (int)min(k, (long long)n + m);
				dist = ed_myers_band<NW>(gA, n, gB, m, gH, kk, shm);
				if (dist <= kk || kk >= n + m) break;	// k = n + m covers every cell: nothing left to widen
				k *= 2;
			}
		}
		if (threadIdx.x == 0)
		{
			prims[p].editDistance = dist;
			prims[p].hpcLenCur = n;
			prims[p].hpcLenExt = m;
		}
		__syncthreads();
	}
}

} // namespace

// fills editDistance / hpcLen* of nPrim device-resident primaries
void fgEditDistances(fg_ctx* c, PrimRec* dPrims, u64 nPrim, int useHpc)
{
	if (!nPrim) return;
	if (nPrim >= 0xFFFFFFFFULL) throw FgError{FG_ERR_ARG, "too many overlaps in one chunk"};
	hipStream_t s = c->stream;
	const int maxLen = std::max(c->maxLen, c->hasQ ? c->qMaxLen : 0);
	const int eMax = getenv("FG_ED_EMAX") ? std::max(1, atoi(getenv("FG_ED_EMAX"))) : ED_EMAX;
	// LDS piece of the O(ND) kernel: the longest sequence, capped (longer substrings go to the bit-vector kernel)
	const int capBases = std::min(maxLen, getenv("FG_ED_LDS_BASES") ? atoi(getenv("FG_ED_LDS_BASES")) : ED_LDS_BASES);
	const int ldsBlocks = (capBases + 63) / 64 + 2;
	const size_t ldsBytes = (size_t)2 * ldsBlocks * 16;
	if (ldsBytes > 64 * 1024)
		HIP_CHECK(hipFuncSetAttribute((const void*)k_edit_ond, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
	const unsigned grid = (unsigned)std::min<u64>(nPrim, 4096);
	c->dEditScratch.reserve((size_t)grid * 2 * (2 * eMax + 5));
	c->dEditList.reserve(2 * nPrim + 2);
	c->dEditCnt.reserve(sizeof(EdCounters));
	EdSeqs S{c->curQuery, c->hasQ ? c->dQWords.p : c->dWords.p, c->hasQ ? c->dQWordOff.p : c->dWordOff.p,
			 c->hasQ ? c->dQLen.p : c->dLen.p, c->dWords.p, c->dWordOff.p, c->dLen.p, c->firstId};
	u32* listSmall = c->dEditList.p;
	u32* listBig = c->dEditList.p + nPrim + 1;
	HIP_CHECK(hipMemsetAsync(c->dEditCnt.p, 0, sizeof(EdCounters), s));
	{ ScopedK t(c->timer, "k_edit_distance");
	  hipLaunchKernelGGL(k_edit_ond, grid, 64, ldsBytes, s, dPrims, nPrim, S, useHpc, ldsBlocks, eMax, c->dEditScratch.p,
						 listSmall, listBig, (EdCounters*)c->dEditCnt.p); }
	EdCounters hc;
	HIP_CHECK(hipMemcpyAsync(&hc, c->dEditCnt.p, sizeof(EdCounters), hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	if (!hc.nSmall && !hc.nBig) return;
	// one slab per resident block, sized for the class's largest pair; the blocks walk the list
	const u64 budgetWords = (getenv("FG_ED_SLAB_BYTES") ? strtoull(getenv("FG_ED_SLAB_BYTES"), nullptr, 10) : (4ULL << 30)) / 8;
	const u64 wS = hc.maxWordsSmall + 8, wB = hc.maxWordsBig + 8;
	const unsigned gridS = hc.nSmall ? (unsigned)std::max<u64>(1, std::min<u64>(std::min<u64>(hc.nSmall, 4096), budgetWords / 2 / wS)) : 0;
	const unsigned gridB = hc.nBig ? (unsigned)std::max<u64>(1, std::min<u64>(std::min<u64>(hc.nBig, 512), budgetWords / 2 / wB)) : 0;
	c->dEditSlab.reserve((size_t)gridS * wS + (size_t)gridB * wB + 8);
	if (hc.nSmall)
	{
		ScopedK t(c->timer, "k_edit_myers");
		hipLaunchKernelGGL(k_edit_myers<1>, gridS, 64, 0, s, dPrims, listSmall, hc.nSmall, S, useHpc, c->dEditSlab.p, wS);
	}
	if (hc.nBig)
	{
		ScopedK t(c->timer, "k_edit_myers_wide");
		hipLaunchKernelGGL(k_edit_myers<ED_BIG_WAVES>, gridB, ED_BIG_WAVES * 64, 0, s, dPrims, listBig, hc.nBig, S, useHpc,
						   c->dEditSlab.p + (size_t)gridS * wS, wB);
	}
}

// kernel-level entry: exact edit distances of nPairs (A_i, B_i) given as the reads 2i, 2i+1 of the
// context's container (forward strands, whole reads), through the same two kernels
void fgDebugEditDistances(fg_ctx* c, u32 nPairs, int useHpc, i32* outDist, i32* outLenA, i32* outLenB)
{
	if (2ULL * nPairs > c->nReads) throw FgError{FG_ERR_ARG, "the container holds fewer than 2 * n_pairs reads"};
	if (c->hasQ) throw FgError{FG_ERR_STATE, "debug edit distances run on the indexed container only"};
	if (!nPairs) return;
	hipStream_t s = c->stream;
	std::vector<PrimRec> h(nPairs);
	std::vector<u32> q(nPairs);
	for (u32 i = 0; i < nPairs; ++i)
	{
		PrimRec r{};
		r.query = i; q[i] = 4 * i;	// record index of read 2i, forward strand
		r.extId = c->firstId + 2 * (2 * i + 1);
		r.curBegin = 0; r.curEnd = c->hLen[2 * i];
		r.extBegin = 0; r.extEnd = c->hLen[2 * i + 1];
		r.extLen = c->hLen[2 * i + 1];
		r.editDistance = -1;
		h[i] = r;
	}
	DevBuf<char> dPrims;
	dPrims.alloc((size_t)nPairs * sizeof(PrimRec));
	c->dQuery.reserve(nPairs);
	HIP_CHECK(hipMemcpyAsync(dPrims.p, h.data(), (size_t)nPairs * sizeof(PrimRec), hipMemcpyHostToDevice, s));
	HIP_CHECK(hipMemcpyAsync(c->dQuery.p, q.data(), nPairs * 4ULL, hipMemcpyHostToDevice, s));
	c->curQuery = c->dQuery.p;
	c->timer.reset();
	fgEditDistances(c, (PrimRec*)dPrims.p, nPairs, useHpc);
	HIP_CHECK(hipMemcpyAsync(h.data(), dPrims.p, (size_t)nPairs * sizeof(PrimRec), hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	c->timer.collect();
	for (u32 i = 0; i < nPairs; ++i) { outDist[i] = h[i].editDistance; outLenA[i] = h[i].hpcLenCur; outLenB[i] = h[i].hpcLenExt; }
}
