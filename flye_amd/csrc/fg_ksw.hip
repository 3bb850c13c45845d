// Banded affine-gap global alignment with CIGAR: getAlignmentCigarKsw (reference
// src/sequence/alignment.cpp:102-216) = ksw_extz2_sse of minimap2 2.17 (reference lib/minimap2/
// ksw2_extz2_sse.c, built with sse2only=1) with Flye's scores (match 2, mismatch -4, gap open 4, gap extend 2),
// band 64 doubling while the band cannot hold the length difference, no z-drop, global backtrack
// (lib/minimap2/ksw2.h:116-152); the host decodes the M/I/D runs into = / X / I / D and computes the error
// rate (floats never come from the device).  SURVEY.md §8(f) N3: the step behind the divergence gate for
// RepeatGraph::build's detector (checkIdyAndTrim, alignment.cpp:306-495) and for the consensus stage.
//
// The alignment PATH is a property of that formulation: Suzuki-Kasahara difference recurrences in 8-bit
// arithmetic, strict-greater tie rules recorded per cell, and a 16-byte-wide loop that also computes cells
// outside the band from whatever the arrays hold there.  So the kernel keeps the same byte state as the
// vector code -- u, v, x, y, s indexed by target position, the target and the reversed query behind them in
// one zeroed buffer whose neighbouring arrays absorb the vector code's loads and stores past array ends --
// and applies the per-cell byte operations of its "gap left-alignment" loop, one anti-diagonal at a time.
// One wave per alignment: every cell of a diagonal depends only on the previous diagonal, so the cells of a
// diagonal (<= band + 31) are spread over the lanes in 64-cell pieces taken from the top (a piece reads its
// left neighbour's old x, v: descending order keeps that value unwritten).  The direction byte of every cell
// goes to a per-alignment backtrack matrix in global memory; lane 0 walks it from (tlen-1, qlen-1).
#include "fg_ctx.h"
#include <chrono>

#include <algorithm>

namespace {

struct KswJob {
	u64 trgOff, qryOff;		// into the batch's byte strings
	i32 tlen, qlen;
	i32 w;					// the band the reference's doubling loop ends with (known on the host: the
							// "band too narrow" test depends on the lengths only)
	i32 feasible;			// 0: even that band fails (the reference then returns an empty CIGAR)
	u64 memOff, pOff, cigOff;	// this alignment's state buffer, backtrack matrix, CIGAR slots
};

__device__ __forceinline__ void ksw_bounds(int r, int qlen, int tlen, int w, int& st, int& en)
{
	st = 0; en = tlen - 1;
	if (st < r - qlen + 1) st = r - qlen + 1;
	if (en > r) en = r;
	if (st < ((r - w + 1) >> 1)) st = (r - w + 1) >> 1;
	if (en > ((r + w) >> 1)) en = (r + w) >> 1;
}

__device__ __forceinline__ void ksw_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

#define KSW_TILE_BYTES 4096
// one wave per workgroup: LDS accesses of a wave complete in order, so all the ring code needs between a write
// and another lane's read is that the compiler keeps the order (and no wait for the direction bytes on their way
// to memory, which a memory fence would include: ~1.5 us per diagonal)
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// the runs of one alignment (written last to first into its slots) -> first to last into the batch's dense list
__device__ __forceinline__ void ksw_emit_runs(const u32* cig, int n, u32 jb, u32* __restrict__ dense, u32* __restrict__ total,
											   u32* __restrict__ runBase, u32* __restrict__ nCigar, int lane)
{
	u32 base = 0;
	if (lane == 0) base = atomicAdd(total, (u32)n);
	base = (u32)__builtin_amdgcn_readfirstlane((int)base);
	ksw_fence();
	for (int k = lane; k < n; k += 64) dense[base + k] = cig[n - 1 - k];
	if (lane == 0) { nCigar[jb] = (u32)n; runBase[jb] = base; }
}


__global__ void __launch_bounds__(64)
k_ksw_extz2(const KswJob* __restrict__ jobs, const u32* __restrict__ order, u32 nJobs, const uint8_t* __restrict__ trgAll,
			const uint8_t* __restrict__ qryAll, uint8_t* __restrict__ scratch, u32* __restrict__ cigars, u32* __restrict__ nCigar,
			u32* __restrict__ dense, u32* __restrict__ total, u32* __restrict__ runBase)
{
	const int lane = threadIdx.x;
	for (u32 jo = blockIdx.x; jo < nJobs; jo += gridDim.x)
	{
		const u32 jb = order[jo];		// this kernel's share of the batch
		const KswJob J = jobs[jb];
		const int tlen = J.tlen, qlen = J.qlen, w = J.w;
		if (!J.feasible || tlen <= 0 || qlen <= 0) { if (lane == 0) { nCigar[jb] = 0; runBase[jb] = 0; } continue; }
		const uint8_t* target = trgAll + J.trgOff;
		const uint8_t* query = qryAll + J.qryOff;
		const int T16 = (tlen + 15) / 16 * 16;
		// Flye's parameters (alignment.cpp:106-109, :124-130): 5 x 5 matrix, the wildcard row / column scores 0
		const int q = 4, e = 2, qe = q + e, m = 5;
		const int scMch = 2, scMis = -4, scN = -e;
		const uint8_t qe2 = (uint8_t)(qe * 2), maxSc = (uint8_t)(scMch + qe * 2);
		int nCol = min(qlen, tlen);
		nCol = ((nCol < w + 1 ? nCol : w + 1) + 15) / 16 + 1;
		const size_t rowBytes = (size_t)nCol * 16;
		uint8_t* U = scratch + J.memOff;		// zeroed by the host for every batch
		uint8_t* V = U + T16; uint8_t* X = V + T16; uint8_t* Y = X + T16; uint8_t* S = Y + T16;
		uint8_t* sf = S + T16; uint8_t* qr = sf + T16;
		uint8_t* P = scratch + J.pOff;
		for (int t = lane; t < qlen; t += 64) qr[t] = query[qlen - 1 - t];
		for (int t = lane; t < tlen; t += 64) sf[t] = target[t];
		ksw_fence();
		int lastSt = -1, lastEn = -1;
		const int R = qlen + tlen - 1;
		for (int r = 0; r < R; ++r)
		{
			int st, en;
			ksw_bounds(r, qlen, tlen, w, st, en);
			const int st0 = st, en0 = en;	// st <= en: the host checked this band
			st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
			uint8_t x1, v1;
			if (st > 0)
			{
				if (st - 1 >= lastSt && st - 1 <= lastEn) { x1 = X[st - 1]; v1 = V[st - 1]; }
				else x1 = v1 = 0;
			}
			else { x1 = 0; v1 = r ? (uint8_t)q : 0; }
			if (en >= r && lane == 0) { Y[r] = 0; U[r] = r ? (uint8_t)q : 0; }
			// scores of the diagonal in 16-byte pieces from st0 (a piece may run past en0, past the end of s into
			// the target bytes, and read the reversed query behind the target: all inside this buffer); one piece
			// after the other, as the vector loop stores them
			const uint8_t* qrr = qr + (qlen - 1 - r);
			for (int t0 = st0; t0 <= en0; t0 += 64)
			{
				// four pieces at a time are independent of each other unless the band spans the whole target (then
				// a piece's spill past s into sf[0..14] can be read by a later piece of the same diagonal): such
				// tiny targets take them one by one
				const bool tiny = T16 <= w + 32;
				if (!tiny)
				{
					const int t = t0 + lane;
					const int pieceEnd = t0 + ((min(en0, t0 + 63) - t0) / 16) * 16 + 15;
					uint8_t sc = 0;
					const bool act = t <= pieceEnd;
					if (act)
					{
						const uint8_t sq = sf[t], sb = qrr[t];
						const bool wild = sq == (uint8_t)(m - 1) || sb == (uint8_t)(m - 1);
						sc = (uint8_t)(wild ? scN : (sq == sb ? scMch : scMis));
					}
					if (act) S[t] = sc;
				}
				else
				{
					for (int p0 = t0; p0 <= min(en0, t0 + 63); p0 += 16)
					{
						uint8_t sc = 0;
						if (lane < 16)
						{
							const uint8_t sq = sf[p0 + lane], sb = qrr[p0 + lane];
							const bool wild = sq == (uint8_t)(m - 1) || sb == (uint8_t)(m - 1);
							sc = (uint8_t)(wild ? scN : (sq == sb ? scMch : scMis));
						}
						ksw_fence();
						if (lane < 16) S[p0 + lane] = sc;
						ksw_fence();
					}
				}
			}
			ksw_fence();
			// the cells of [st, en], 64 at a time from the top
			const int nCell = en - st + 1;
			uint8_t* Prow = P + (size_t)r * rowBytes;
			for (int c0 = ((nCell - 1) / 64) * 64; c0 >= 0; c0 -= 64)
			{
				const int t = st + c0 + lane;
				if (t <= en)
				{
					const uint8_t xt1 = t == st ? x1 : X[t - 1], vt1 = t == st ? v1 : V[t - 1];
					const uint8_t ut = U[t];
					uint8_t z = (uint8_t)(S[t] + qe2);
					const uint8_t a = (uint8_t)(xt1 + vt1), b = (uint8_t)(Y[t] + ut);
					uint8_t d = (int8_t)a > (int8_t)z ? 1 : 0;
					z = (int8_t)z > 0 ? z : 0;
					z = max(z, a);
					if ((int8_t)b > (int8_t)z) d = 2;
					z = max(z, b);
					z = min(z, maxSc);
					const uint8_t nu = (uint8_t)(z - vt1), nv = (uint8_t)(z - ut);
					const uint8_t zq = (uint8_t)(z - q);
					const uint8_t a2 = (uint8_t)(a - zq), b2 = (uint8_t)(b - zq);
					uint8_t nx = 0, ny = 0;
					if ((int8_t)a2 > 0) { nx = a2; d |= 0x08; }
					if ((int8_t)b2 > 0) { ny = b2; d |= 0x10; }
					// (every load of this piece is a source of these stores: they cannot pass the loads)
					U[t] = nu; V[t] = nv; X[t] = nx; Y[t] = ny;
					if ((size_t)(t - st) < rowBytes) Prow[t - st] = d;
				}
			}
			ksw_fence();
			lastSt = st; lastEn = en;
		}
		// backtrack (ksw2.h:116-152: rotated matrix, no introns), lane 0; runs come out last to first
		u32* cig = cigars + J.cigOff;
		int n = 0;
		if (lane == 0)
		{
			int i = tlen - 1, j = qlen - 1, state = 0;
			u32 curOp = 0xFFFFFFFFu, curLen = 0;
			auto push = [&](u32 op, u32 len)
			{
				if (op == curOp) curLen += len;
				else { if (curOp != 0xFFFFFFFFu) cig[n++] = curLen << 4 | curOp; curOp = op; curLen = len; }
			};
			while (i >= 0 && j >= 0)
			{
				const int r = i + j;
				int st, en;
				ksw_bounds(r, qlen, tlen, w, st, en);
				st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
				int force = -1;
				if (i < st) force = 2;
				if (i > en) force = 1;
				const u32 tmp = force < 0 ? (u32)P[(size_t)r * rowBytes + i - st] : 0u;
				if (state == 0) state = tmp & 7;
				else if (!((tmp >> (state + 2)) & 1)) state = 0;
				if (state == 0) state = tmp & 7;
				if (force >= 0) state = force;
				if (state == 0) { push(0, 1); --i; --j; }
				else if (state == 1 || state == 3) { push(2, 1); --i; }
				else { push(1, 1); --j; }
			}
			if (i >= 0) push(2, (u32)i + 1);
			if (j >= 0) push(1, (u32)j + 1);
			if (curOp != 0xFFFFFFFFu) cig[n++] = curLen << 4 | curOp;
		}
		n = __builtin_amdgcn_readfirstlane(n);
		ksw_emit_runs(cig, n, jb, dense, total, runBase, nCigar, lane);
	}
}

// ---- the same recurrences with the state in LDS ------------------------------------------------------------
// Only a window of the byte arrays is alive: cells left of st - 1 are never read again, cells right of the
// highest index touched so far still hold their initial zeros.  So u, v, x, y, s live in RINGS of RING bytes
// (index = target position mod RING; a slot is zeroed before a new position takes it over), the target and
// query bytes the scores need sit in two more rings refilled 64 bytes at a time.  A diagonal is then ONE LDS
// round trip: every lane reads what its cells need (NCH cells per lane, 64 apart), all lanes compute, all lanes
// write -- the vector code's order (scores first, then 16-byte pieces from the right end) only matters through
// which value a read sees, and reading everything before writing anything sees the same values.  Same bytes as
// the kernel above for every cell it computes; used when the band does not span the whole target (otherwise
// the vector code's spill past the end of s into the target bytes is read back) and the rings fit (band <= 512).
// The backtrack walks tiles of the direction matrix staged in the same LDS (the rings are dead by then).
template <int RING, int NCH, int G, int ROWMAX>
__global__ void __launch_bounds__(64)
k_ksw_extz2_lds(const KswJob* __restrict__ jobs, const u32* __restrict__ order, u32 nJobs, const uint8_t* __restrict__ trgAll,
				const uint8_t* __restrict__ qryAll, uint8_t* __restrict__ scratch, u32* __restrict__ cigars, u32* __restrict__ nCigar,
				u32* __restrict__ dense, u32* __restrict__ total, u32* __restrict__ runBase, u32 dbg)
{
	// direction bytes are staged G rows at a time and leave as 16-byte stores: a byte store per lane is a memory
	// transaction per lane, and two of those per diagonal were what the whole kernel waited for
	constexpr int SMEM = 7 * RING + G * ROWMAX > KSW_TILE_BYTES ? 7 * RING + G * ROWMAX : KSW_TILE_BYTES;
	__shared__ __attribute__((aligned(16))) uint8_t smem[SMEM];
	uint8_t* const sU = smem; uint8_t* const sV = smem + RING; uint8_t* const sX = smem + 2 * RING; uint8_t* const sY = smem + 3 * RING;
	uint8_t* const sS = smem + 4 * RING; uint8_t* const sT = smem + 5 * RING; uint8_t* const sQ = smem + 6 * RING;
	uint8_t* const sStage = smem + 7 * RING;
	uint8_t* const sTile = smem;
	const int lane = threadIdx.x;
	constexpr int M = RING - 1;
	for (u32 jo = blockIdx.x; jo < nJobs; jo += gridDim.x)
	{
		const u32 jb = order[jo];
		const KswJob J = jobs[jb];
		const int tlen = J.tlen, qlen = J.qlen, w = J.w;
		const uint8_t* target = trgAll + J.trgOff;
		const uint8_t* query = qryAll + J.qryOff;
		const int T16 = (tlen + 15) / 16 * 16;
		const int q = 4, e = 2, qe = q + e, m = 5;
		const int scMch = 2, scMis = -4, scN = -e;
		const uint8_t qe2 = (uint8_t)(qe * 2), maxSc = (uint8_t)(scMch + qe * 2);
		int nCol = min(qlen, tlen);
		nCol = ((nCol < w + 1 ? nCol : w + 1) + 15) / 16 + 1;
		const int rowBytes = nCol * 16;
		uint8_t* P = scratch + J.pOff;
		lds_fence();		// the previous job's walk is done with the tile
		// rings: positions [0, RING) start as zeros / as the first bytes of the strings
		for (int i = lane; i < RING; i += 64)
		{
			sU[i] = 0; sV[i] = 0; sX[i] = 0; sY[i] = 0; sS[i] = 0;
			sT[i] = i < tlen ? target[i] : 0;
			sQ[i] = i < qlen ? query[i] : 0;
		}
		int zeroed = RING;		// state slots hold positions [zeroed - RING, zeroed)
		int tLoaded = RING;		// target ring holds positions [tLoaded - RING, tLoaded)
		int qLoaded = RING;		// query ring holds positions [qLoaded - RING, qLoaded)
		lds_fence();
		int lastSt = -1, lastEn = -1;
		const int R = (dbg & 1) ? 0 : qlen + tlen - 1;		// dbg: timing experiments (FG_KSW_DEBUG), results are then wrong
		for (int r = 0; r < R; ++r)
		{
			int st, en;
			ksw_bounds(r, qlen, tlen, w, st, en);
			const int st0 = st, en0 = en;
			st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
			// highest position this diagonal touches: its cells, and the last score piece (clipped to the array)
			const int pieceEnd = min(T16 - 1, st0 + (en0 - st0) / 16 * 16 + 15);
			const int hi = max(en, pieceEnd);
			if (zeroed <= hi || tLoaded <= pieceEnd || qLoaded <= r - st0)
			{
				while (zeroed <= hi)
				{
					// the slots taken over held positions below st - 1 - 64: dead (RING >= band + 98)
					sU[(zeroed + lane) & M] = 0; sV[(zeroed + lane) & M] = 0; sX[(zeroed + lane) & M] = 0;
					sY[(zeroed + lane) & M] = 0; sS[(zeroed + lane) & M] = 0;
					zeroed += 64;
				}
				while (tLoaded <= pieceEnd)
				{
					const int t = tLoaded + lane;
					sT[t & M] = t < tlen ? target[t] : 0;
					tLoaded += 64;
				}
				while (qLoaded <= r - st0)
				{
					const int jq = qLoaded + lane;
					sQ[jq & M] = jq < qlen ? query[jq] : 0;
					qLoaded += 64;
				}
				lds_fence();
			}
			const bool prevHas = st - 1 >= lastSt && st - 1 <= lastEn;
			// read: every lane reads all seven bytes of each of its positions, needed or not (any ring index is a
			// valid address) -- no branches between the reads, one wait for all of them
			uint8_t ut[NCH], yt[NCH], xt1[NCH], vt1[NCH], ss[NCH], tq[NCH], tb[NCH];
#pragma unroll
			for (int c = 0; c < NCH; ++c)
			{
				const int t = st + 64 * c + lane;
				tq[c] = sT[t & M]; tb[c] = sQ[(r - t) & M]; ss[c] = sS[t & M];
				ut[c] = sU[t & M]; yt[c] = sY[t & M];
				xt1[c] = sX[(t - 1) & M]; vt1[c] = sV[(t - 1) & M];
			}
			lds_fence();
			// compute, write
			uint8_t* Prow = sStage + (r & (G - 1)) * rowBytes;
#pragma unroll
			for (int c = 0; c < NCH; ++c)
			{
				const int t = st + 64 * c + lane;
				// the score pieces cover st0 .. pieceEnd (query position r - t, zeros behind its start); outside
				// them s keeps what an earlier diagonal left
				const bool piece = t >= st0 && t <= pieceEnd;
				const uint8_t sb = r - t >= 0 ? tb[c] : (uint8_t)0;
				const bool wild = tq[c] == (uint8_t)(m - 1) || sb == (uint8_t)(m - 1);
				const uint8_t sc = piece ? (uint8_t)(wild ? scN : (tq[c] == sb ? scMch : scMis)) : ss[c];
				if (piece) sS[t & M] = sc;
				if (t <= en)
				{
					uint8_t x1 = xt1[c], v1 = vt1[c], u = ut[c], y = yt[c];
					if (t == st)
					{
						// left of the first cell: the previous diagonal's cell if it had one there
						if (st > 0) { if (!prevHas) x1 = v1 = 0; }
						else { x1 = 0; v1 = r ? (uint8_t)q : 0; }
					}
					if (t == r) { y = 0; u = r ? (uint8_t)q : 0; }		// the first row's boundary (en >= r here)
					uint8_t z = (uint8_t)(sc + qe2);
					const uint8_t a = (uint8_t)(x1 + v1), b = (uint8_t)(y + u);
					uint8_t d = (int8_t)a > (int8_t)z ? 1 : 0;
					z = (int8_t)z > 0 ? z : 0;
					z = max(z, a);
					if ((int8_t)b > (int8_t)z) d = 2;
					z = max(z, b);
					z = min(z, maxSc);
					const uint8_t nu = (uint8_t)(z - v1), nv = (uint8_t)(z - u);
					const uint8_t zq = (uint8_t)(z - q);
					const uint8_t a2 = (uint8_t)(a - zq), b2 = (uint8_t)(b - zq);
					uint8_t nx = 0, ny = 0;
					if ((int8_t)a2 > 0) { nx = a2; d |= 0x08; }
					if ((int8_t)b2 > 0) { ny = b2; d |= 0x10; }
					sU[t & M] = nu; sV[t & M] = nv; sX[t & M] = nx; sY[t & M] = ny;
					if (t - st < rowBytes) Prow[t - st] = d;
				}
			}
			lds_fence();
			lastSt = st; lastEn = en;
			if ((r & (G - 1)) == G - 1 || r == R - 1)
			{
				const int r0 = r & ~(G - 1);
				uint4* dst = (uint4*)(P + (size_t)r0 * rowBytes);		// rows are 16-byte multiples, P is 16-byte aligned
				for (int k = lane; k < (r - r0 + 1) * (rowBytes / 16); k += 64) dst[k] = ((const uint4*)sStage)[k];
			}
		}
		ksw_fence();		// the direction bytes are in memory
		// backtrack on tiles of the direction matrix staged in LDS; every lane follows the same walk (broadcast
		// reads), lane 0 records the runs.  While the walk is in the match state, the lanes look at the next 64
		// cells down the diagonal at once and take the whole stretch that stays in that state as one run.
		u32* cig = cigars + J.cigOff;
		int n = 0;
		{
			const int tileRows = max(1, min(128, KSW_TILE_BYTES / rowBytes));
			int tileLo = 0x7fffffff, tileHi = -1;	// rows [tileLo, tileHi] are staged
			int i = (dbg & 2) ? -1 : tlen - 1, j = qlen - 1, state = 0;
			u32 curOp = 0xFFFFFFFFu, curLen = 0;
			bool lookAhead = !(dbg & 4);
			auto push = [&](u32 op, u32 len)
			{
				if (op == curOp) curLen += len;
				else { if (curOp != 0xFFFFFFFFu) { if (lane == 0) cig[n] = curLen << 4 | curOp; ++n; } curOp = op; curLen = len; }
			};
			while (i >= 0 && j >= 0)
			{
				const int r = i + j;
				if (r < tileLo || r > tileHi)
				{
					lds_fence();
					tileHi = r; tileLo = max(0, r - tileRows + 1);
					const int bytes = (tileHi - tileLo + 1) * rowBytes;
					const uint4* src = (const uint4*)(P + (size_t)tileLo * rowBytes);	// rows are 16-byte multiples
					for (int k = lane; k < bytes / 16; k += 64) ((uint4*)sTile)[k] = src[k];
					lds_fence();
				}
				if (state == 0 && lookAhead)
				{
					const int ik = i - lane, jk = j - lane, rk = r - 2 * lane;
					bool isM = false;
					if (ik >= 0 && jk >= 0 && rk >= tileLo)
					{
						int stk, enk;
						ksw_bounds(rk, qlen, tlen, w, stk, enk);
						stk = stk / 16 * 16; enk = (enk + 16) / 16 * 16 - 1;
						if (ik >= stk && ik <= enk) isM = (sTile[(rk - tileLo) * rowBytes + ik - stk] & 7) == 0;
					}
					const u64 notM = ~__ballot(isM);
					const int L = notM ? __builtin_ctzll(notM) : 64;
					lookAhead = false;		// the cell behind the stretch takes the step below
					if (L > 0) { push(0, (u32)L); i -= L; j -= L; continue; }
				}
				lookAhead = !(dbg & 4);
				int st, en;
				ksw_bounds(r, qlen, tlen, w, st, en);
				st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
				int force = -1;
				if (i < st) force = 2;
				if (i > en) force = 1;
				const u32 tmp = force < 0 ? (u32)sTile[(r - tileLo) * rowBytes + i - st] : 0u;
				if (state == 0) state = tmp & 7;
				else if (!((tmp >> (state + 2)) & 1)) state = 0;
				if (state == 0) state = tmp & 7;
				if (force >= 0) state = force;
				if (state == 0) { push(0, 1); --i; --j; }
				else if (state == 1 || state == 3) { push(2, 1); --i; }
				else { push(1, 1); --j; }
			}
			if (i >= 0) push(2, (u32)i + 1);
			if (j >= 0) push(1, (u32)j + 1);
			if (curOp != 0xFFFFFFFFu) { if (lane == 0) cig[n] = curLen << 4 | curOp; ++n; }
		}
		ksw_emit_runs(cig, n, jb, dense, total, runBase, nCigar, lane);
	}
}

// ---- the same recurrences with the state in REGISTERS --------------------------------------------------------------
// The LDS form above pays two LDS round trips per anti-diagonal (read what the cells need, write what they produce), and
// a diagonal depends on the one before: ~1.2 us per diagonal at band 64 however empty the chip is -- a 10 kb pair takes
// ~20 ms, and a batch of a few hundred pairs (the consensus of one genome's disjointigs) is bound by that latency.
// Here u, v, x, y, s of the cells a diagonal can touch live in registers, as a WINDOW relative to the diagonal's first
// cell: cell 64 c + lane of the window = target position st + 64 c + lane (c < NCH; 64 NCH > band + 30).  The left
// neighbour's old x, v come by one DPP lane shift (lane 0 of a chunk from lane 63 of the chunk below); st only ever grows,
// by 16 at a time (it is rounded down to the vector width), and then the window slides down 16 lanes through the LDS
// crossbar (ds_bpermute: no memory access) -- positions that slide in from above were never touched and hold the zeros of
// the reference's calloc'ed arrays, positions that slide out below are never read again except st - 1, which is taken
// before the slide.  Target and query bytes still come from LDS rings (they do not depend on the recurrence).
// Same bytes as the two kernels above for every cell it computes, same direction matrix, same backtrack.
__device__ __forceinline__ u32 ksw_shr1(u32 v, u32 fill)
{
	return (u32)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false);	// lane L <- lane L - 1, lane 0 <- fill
}

template <int RING, int NCH, int G, int ROWMAX>
__global__ void __launch_bounds__(64)
k_ksw_extz2_reg(const KswJob* __restrict__ jobs, const u32* __restrict__ order, u32 nJobs, const uint8_t* __restrict__ trgAll,
				const uint8_t* __restrict__ qryAll, uint8_t* __restrict__ scratch, u32* __restrict__ cigars, u32* __restrict__ nCigar,
				u32* __restrict__ dense, u32* __restrict__ total, u32* __restrict__ runBase, u32 dbg)
{
	constexpr int SMEM = 2 * RING + G * ROWMAX > KSW_TILE_BYTES ? 2 * RING + G * ROWMAX : KSW_TILE_BYTES;
	__shared__ __attribute__((aligned(16))) uint8_t smem[SMEM];
	uint8_t* const sT = smem; uint8_t* const sQ = smem + RING;
	uint8_t* const sStage = smem + 2 * RING;
	uint8_t* const sTile = smem;
	const int lane = threadIdx.x;
	constexpr int M = RING - 1;
	const int slideAddr = ((lane + 16) & 63) << 2;
	for (u32 jo = blockIdx.x; jo < nJobs; jo += gridDim.x)
	{
		const u32 jb = order[jo];
		const KswJob J = jobs[jb];
		const int tlen = J.tlen, qlen = J.qlen, w = J.w;
		const uint8_t* target = trgAll + J.trgOff;
		const uint8_t* query = qryAll + J.qryOff;
		const int T16 = (tlen + 15) / 16 * 16;
		const int q = 4, e = 2, qe = q + e, m = 5;
		const int scMch = 2, scMis = -4, scN = -e;
		const uint8_t qe2 = (uint8_t)(qe * 2), maxSc = (uint8_t)(scMch + qe * 2);
		int nCol = min(qlen, tlen);
		nCol = ((nCol < w + 1 ? nCol : w + 1) + 15) / 16 + 1;
		const int rowBytes = nCol * 16;
		uint8_t* P = scratch + J.pOff;
		lds_fence();		// the previous job's walk is done with the tile
		for (int i = lane; i < RING; i += 64)
		{
			sT[i] = i < tlen ? target[i] : 0;
			sQ[i] = i < qlen ? query[i] : 0;
		}
		int tLoaded = RING;		// target ring holds positions [tLoaded - RING, tLoaded)
		int qLoaded = RING;		// query ring holds positions [qLoaded - RING, qLoaded)
		lds_fence();
		// the window: all zeros (calloc), first cell at position 0
		u32 U[NCH], V[NCH], X[NCH], Y[NCH], S[NCH];
#pragma unroll
		for (int c = 0; c < NCH; ++c) { U[c] = 0; V[c] = 0; X[c] = 0; Y[c] = 0; S[c] = 0; }
		int wst = 0;
		int lastSt = -1, lastEn = -1;
		const int R = (dbg & 1) ? 0 : qlen + tlen - 1;		// dbg: timing experiments (FG_KSW_DEBUG), results are then wrong
		for (int r = 0; r < R; ++r)
		{
			int st, en;
			ksw_bounds(r, qlen, tlen, w, st, en);
			const int st0 = st, en0 = en;
			st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
			const int pieceEnd = min(T16 - 1, st0 + (en0 - st0) / 16 * 16 + 15);
			if (tLoaded <= pieceEnd || qLoaded <= r - st0)
			{
				while (tLoaded <= pieceEnd)
				{
					const int t = tLoaded + lane;
					sT[t & M] = t < tlen ? target[t] : 0;
					tLoaded += 64;
				}
				while (qLoaded <= r - st0)
				{
					const int jq = qLoaded + lane;
					sQ[jq & M] = jq < qlen ? query[jq] : 0;
					qLoaded += 64;
				}
				lds_fence();
			}
			const bool prevHas = st - 1 >= lastSt && st - 1 <= lastEn;
			// the target / query bytes of this diagonal's cells: issued first, waited for behind the register work below
			uint8_t tq[NCH], tb[NCH];
#pragma unroll
			for (int c = 0; c < NCH; ++c)
			{
				const int t = st + 64 * c + lane;
				tq[c] = sT[t & M]; tb[c] = sQ[(r - t) & M];
			}
			// the window follows st: 16 lanes down (st - 1 = cell 15 of the old window is kept for the first cell)
			u32 xEdge = 0, vEdge = 0;
			if (st != wst)
			{
				xEdge = (u32)__builtin_amdgcn_readlane((int)X[0], 15);
				vEdge = (u32)__builtin_amdgcn_readlane((int)V[0], 15);
				u32 nU[NCH], nV[NCH], nX[NCH], nY[NCH], nS[NCH];
#pragma unroll
				for (int c = 0; c < NCH; ++c)
				{
					const bool low = lane < 48;
					const u32 aU = (u32)__builtin_amdgcn_ds_bpermute(slideAddr, (int)U[c]), bU = c + 1 < NCH ? (u32)__builtin_amdgcn_ds_bpermute(slideAddr, (int)U[c + 1 < NCH ? c + 1 : c]) : 0u;
					const u32 aV = (u32)__builtin_amdgcn_ds_bpermute(slideAddr, (int)V[c]), bV = c + 1 < NCH ? (u32)__builtin_amdgcn_ds_bpermute(slideAddr, (int)V[c + 1 < NCH ? c + 1 : c]) : 0u;
					const u32 aX = (u32)__builtin_amdgcn_ds_bpermute(slideAddr, (int)X[c]), bX = c + 1 < NCH ? (u32)__builtin_amdgcn_ds_bpermute(slideAddr, (int)X[c + 1 < NCH ? c + 1 : c]) : 0u;
					const u32 aY = (u32)__builtin_amdgcn_ds_bpermute(slideAddr, (int)Y[c]), bY = c + 1 < NCH ? (u32)__builtin_amdgcn_ds_bpermute(slideAddr, (int)Y[c + 1 < NCH ? c + 1 : c]) : 0u;
					const u32 aS = (u32)__builtin_amdgcn_ds_bpermute(slideAddr, (int)S[c]), bS = c + 1 < NCH ? (u32)__builtin_amdgcn_ds_bpermute(slideAddr, (int)S[c + 1 < NCH ? c + 1 : c]) : 0u;
					nU[c] = low ? aU : bU; nV[c] = low ? aV : bV; nX[c] = low ? aX : bX; nY[c] = low ? aY : bY; nS[c] = low ? aS : bS;
				}
#pragma unroll
				for (int c = 0; c < NCH; ++c) { U[c] = nU[c]; V[c] = nV[c]; X[c] = nX[c]; Y[c] = nY[c]; S[c] = nS[c]; }
				wst = st;
			}
			// the left neighbours' old x, v, before any cell of this diagonal is written
			u32 x1r[NCH], v1r[NCH];
#pragma unroll
			for (int c = 0; c < NCH; ++c)
			{
				const u32 fx = c ? (u32)__builtin_amdgcn_readlane((int)X[c ? c - 1 : 0], 63) : 0u;
				const u32 fv = c ? (u32)__builtin_amdgcn_readlane((int)V[c ? c - 1 : 0], 63) : 0u;
				x1r[c] = ksw_shr1(X[c], fx); v1r[c] = ksw_shr1(V[c], fv);
			}
			lds_fence();		// tq, tb
			uint8_t* Prow = sStage + (r & (G - 1)) * rowBytes;
#pragma unroll
			for (int c = 0; c < NCH; ++c)
			{
				const int t = st + 64 * c + lane;
				// the score pieces cover st0 .. pieceEnd (query position r - t, zeros behind its start); outside
				// them s keeps what an earlier diagonal left
				const bool piece = t >= st0 && t <= pieceEnd;
				const uint8_t sb = r - t >= 0 ? tb[c] : (uint8_t)0;
				const bool wild = tq[c] == (uint8_t)(m - 1) || sb == (uint8_t)(m - 1);
				const uint8_t sc = piece ? (uint8_t)(wild ? scN : (tq[c] == sb ? scMch : scMis)) : (uint8_t)S[c];
				S[c] = sc;
				if (t <= en)
				{
					uint8_t x1 = (uint8_t)x1r[c], v1 = (uint8_t)v1r[c], u = (uint8_t)U[c], y = (uint8_t)Y[c];
					if (t == st)
					{
						// left of the first cell: the previous diagonal's cell if it had one there
						if (st > 0) { if (prevHas) { x1 = (uint8_t)xEdge; v1 = (uint8_t)vEdge; } else x1 = v1 = 0; }
						else { x1 = 0; v1 = r ? (uint8_t)q : 0; }
					}
					if (t == r) { y = 0; u = r ? (uint8_t)q : 0; }		// the first row's boundary (en >= r here)
					uint8_t z = (uint8_t)(sc + qe2);
					const uint8_t a = (uint8_t)(x1 + v1), b = (uint8_t)(y + u);
					uint8_t d = (int8_t)a > (int8_t)z ? 1 : 0;
					z = (int8_t)z > 0 ? z : 0;
					z = max(z, a);
					if ((int8_t)b > (int8_t)z) d = 2;
					z = max(z, b);
					z = min(z, maxSc);
					const uint8_t nu = (uint8_t)(z - v1), nv = (uint8_t)(z - u);
					const uint8_t zq = (uint8_t)(z - q);
					const uint8_t a2 = (uint8_t)(a - zq), b2 = (uint8_t)(b - zq);
					uint8_t nx = 0, ny = 0;
					if ((int8_t)a2 > 0) { nx = a2; d |= 0x08; }
					if ((int8_t)b2 > 0) { ny = b2; d |= 0x10; }
					U[c] = nu; V[c] = nv; X[c] = nx; Y[c] = ny;
					if (t - st < rowBytes) Prow[t - st] = d;
				}
			}
			lds_fence();
			lastSt = st; lastEn = en;
			if ((r & (G - 1)) == G - 1 || r == R - 1)
			{
				const int r0 = r & ~(G - 1);
				uint4* dst = (uint4*)(P + (size_t)r0 * rowBytes);		// rows are 16-byte multiples, P is 16-byte aligned
				for (int k = lane; k < (r - r0 + 1) * (rowBytes / 16); k += 64) dst[k] = ((const uint4*)sStage)[k];
			}
		}
		ksw_fence();		// the direction bytes are in memory
		// backtrack on tiles of the direction matrix staged in LDS; every lane follows the same walk (broadcast
		// reads), lane 0 records the runs.  While the walk is in the match state, the lanes look at the next 64
		// cells down the diagonal at once and take the whole stretch that stays in that state as one run.
		u32* cig = cigars + J.cigOff;
		int n = 0;
		{
			const int tileRows = max(1, min(128, KSW_TILE_BYTES / rowBytes));
			int tileLo = 0x7fffffff, tileHi = -1;	// rows [tileLo, tileHi] are staged
			int i = (dbg & 2) ? -1 : tlen - 1, j = qlen - 1, state = 0;
			u32 curOp = 0xFFFFFFFFu, curLen = 0;
			bool lookAhead = !(dbg & 4);
			auto push = [&](u32 op, u32 len)
			{
				if (op == curOp) curLen += len;
				else { if (curOp != 0xFFFFFFFFu) { if (lane == 0) cig[n] = curLen << 4 | curOp; ++n; } curOp = op; curLen = len; }
			};
			while (i >= 0 && j >= 0)
			{
				const int r = i + j;
				if (r < tileLo || r > tileHi)
				{
					lds_fence();
					tileHi = r; tileLo = max(0, r - tileRows + 1);
					const int bytes = (tileHi - tileLo + 1) * rowBytes;
					const uint4* src = (const uint4*)(P + (size_t)tileLo * rowBytes);	// rows are 16-byte multiples
					for (int k = lane; k < bytes / 16; k += 64) ((uint4*)sTile)[k] = src[k];
					lds_fence();
				}
				if (state == 0 && lookAhead)
				{
					const int ik = i - lane, jk = j - lane, rk = r - 2 * lane;
					bool isM = false;
					if (ik >= 0 && jk >= 0 && rk >= tileLo)
					{
						int stk, enk;
						ksw_bounds(rk, qlen, tlen, w, stk, enk);
						stk = stk / 16 * 16; enk = (enk + 16) / 16 * 16 - 1;
						if (ik >= stk && ik <= enk) isM = (sTile[(rk - tileLo) * rowBytes + ik - stk] & 7) == 0;
					}
					const u64 notM = ~__ballot(isM);
					const int L = notM ? __builtin_ctzll(notM) : 64;
					lookAhead = false;		// the cell behind the stretch takes the step below
					if (L > 0) { push(0, (u32)L); i -= L; j -= L; continue; }
				}
				lookAhead = !(dbg & 4);
				int st, en;
				ksw_bounds(r, qlen, tlen, w, st, en);
				st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
				int force = -1;
				if (i < st) force = 2;
				if (i > en) force = 1;
				const u32 tmp = force < 0 ? (u32)sTile[(r - tileLo) * rowBytes + i - st] : 0u;
				if (state == 0) state = tmp & 7;
				else if (!((tmp >> (state + 2)) & 1)) state = 0;
				if (state == 0) state = tmp & 7;
				if (force >= 0) state = force;
				if (state == 0) { push(0, 1); --i; --j; }
				else if (state == 1 || state == 3) { push(2, 1); --i; }
				else { push(1, 1); --j; }
			}
			if (i >= 0) push(2, (u32)i + 1);
			if (j >= 0) push(1, (u32)j + 1);
			if (curOp != 0xFFFFFFFFu) { if (lane == 0) cig[n] = curLen << 4 | curOp; ++n; }
		}
		ksw_emit_runs(cig, n, jb, dense, total, runBase, nCigar, lane);
	}
}

// does the band w connect (0, 0) with (tlen-1, qlen-1)?  ksw_extz2 gives up ("band too narrow", st > en on some
// diagonal r) otherwise.  st = max(0, r-qlen+1, (r-w+1)>>1), en = min(tlen-1, r, (r+w)>>1): of the nine
// (lower, upper) pairs only r-qlen+1 > (r+w)>>1 and (r-w+1)>>1 > tlen-1 can happen, and both differences never
// decrease with r, so the last diagonal decides.
bool bandFeasible(int qlen, int tlen, int w)
{
	const int r = qlen + tlen - 2;
	int st = 0, en = tlen - 1;
	if (st < r - qlen + 1) st = r - qlen + 1;
	if (en > r) en = r;
	if (st < ((r - w + 1) >> 1)) st = (r - w + 1) >> 1;
	if (en > ((r + w) >> 1)) en = (r + w) >> 1;
	return st <= en;
}

} // namespace

// ksw-form CIGARs (len << 4 | op, op 0 = M, 1 = I, 2 = D, first run first) of nPairs (target, query) byte-string
// pairs; run counts in nRuns, runs of pair i at runs[runOff[i] ..)
void fgKswAlign(fg_ctx* c, u32 nPairs, const uint8_t* trg, const u64* trgOff, const uint8_t* qry, const u64* qryOff,
				std::vector<u64>& runOff, std::vector<u32>& runs)
{
	hipStream_t s = c->stream;
	runOff.assign(nPairs + 1, 0);
	runs.clear();
	if (!nPairs) return;
	c->timer.reset();
	const bool trace = getenv("FG_KSW_TRACE") != nullptr;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
	auto t0 = now();
	std::vector<KswJob> jobs(nPairs);
	std::vector<u64> cigCap(nPairs);
	for (u32 i = 0; i < nPairs; ++i)
	{
		KswJob& J = jobs[i];
		J.trgOff = trgOff[i]; J.qryOff = qryOff[i];
		const u64 tl = trgOff[i + 1] - trgOff[i], ql = qryOff[i + 1] - qryOff[i];
		if (tl > 0x3FFFFFFF || ql > 0x3FFFFFFF) throw FgError{FG_ERR_ARG, "sequence too long for the alignment kernel"};
		J.tlen = (i32)tl; J.qlen = (i32)ql;
		// the reference's loop (alignment.cpp:147-159): band 64, doubled while the band is too narrow, given up once
		// it exceeds both lengths
		int w = 64;
		bool ok = false;
		if (tl && ql)
			for (;;)
			{
				ok = bandFeasible(J.qlen, J.tlen, w);
				if (ok) break;
				if (w > std::max(J.qlen, J.tlen)) break;
				w *= 2;
			}
		J.w = w; J.feasible = ok ? 1 : 0;
		cigCap[i] = ok ? (u64)J.tlen + J.qlen + 2 : 0;
	}
	const u64 nTrg = trgOff[nPairs], nQry = qryOff[nPairs];
	DevBuf<uint8_t> dTrg, dQry;
	dTrg.alloc(nTrg + 64); dQry.alloc(nQry + 64);
	if (nTrg) HIP_CHECK(hipMemcpyAsync(dTrg.p, trg, nTrg, hipMemcpyHostToDevice, s));
	if (nQry) HIP_CHECK(hipMemcpyAsync(dQry.p, qry, nQry, hipMemcpyHostToDevice, s));
	// sub-batches bounded by scratch memory (state buffer + backtrack matrix per alignment)
	const u64 budget = getenv("FG_KSW_SCRATCH_BYTES") ? strtoull(getenv("FG_KSW_SCRATCH_BYTES"), nullptr, 10) : (8ULL << 30);
	std::vector<u32> nRunsAll(nPairs, 0);
	std::vector<std::vector<u32>> parts;
	u32 a = 0;
	DevBuf<uint8_t> dScratch;
	DevBuf<KswJob> dJobs;
	DevBuf<u32> dCig, dN, dOrder, dDense, dBase, dTotal;
	dTotal.alloc(1);
	std::vector<u64> partOff(nPairs, 0);
	while (a < nPairs)
	{
		u32 b = a;
		u64 memTotal = 0, pTotal = 0, cigTotal = 0;
		std::vector<KswJob> sub;
		while (b < nPairs)
		{
			KswJob J = jobs[b];
			const u64 T16 = ((u64)J.tlen + 15) / 16 * 16, Q16 = ((u64)J.qlen + 15) / 16 * 16;
			u64 nCol = std::min(J.qlen, J.tlen);
			nCol = ((nCol < (u64)J.w + 1 ? nCol : (u64)J.w + 1) + 15) / 16 + 1;
			const u64 memB = J.feasible ? T16 * 6 + Q16 + 64 : 0;
			const u64 pB = J.feasible ? ((u64)J.qlen + J.tlen) * nCol * 16 + 64 : 0;
			if (b > a && (memTotal + pTotal + memB + pB > budget)) break;
			J.memOff = memTotal; J.pOff = pB;	// pOff fixed up below (behind all state buffers)
			memTotal += memB; pTotal += pB; J.cigOff = cigTotal; cigTotal += cigCap[b];
			sub.push_back(J);
			++b;
		}
		u64 pRun = memTotal;
		for (auto& J : sub) { const u64 pB = J.pOff; J.pOff = pRun; pRun += pB; }
		auto t1 = now();
		dScratch.reserve(memTotal + pTotal + 64);
		dJobs.reserve(sub.size()); dCig.reserve(cigTotal + 1); dN.reserve(sub.size());
		dDense.reserve(cigTotal + 1); dBase.reserve(sub.size());
		HIP_CHECK(hipMemsetAsync(dTotal.p, 0, 4, s));
		HIP_CHECK(hipMemsetAsync(dScratch.p, 0, memTotal + 64, s));		// the state buffers start zeroed (kcalloc)
		HIP_CHECK(hipMemcpyAsync(dJobs.p, sub.data(), sub.size() * sizeof(KswJob), hipMemcpyHostToDevice, s));
		{
			// the batch by kernel: state in memory (tiny targets, where the vector code's spill matters; infeasible and
			// empty pairs; bands beyond the rings), or in LDS rings of the size the band needs (band + 98 <= RING)
			const bool litOnly = getenv("FG_KSW_LITERAL") != nullptr;
			std::vector<u32> order[5];
			for (u32 i = 0; i < sub.size(); ++i)
			{
				const KswJob& J = sub[i];
				const i32 T16 = (J.tlen + 15) / 16 * 16;
				int cls = 0;
				if (!litOnly && J.feasible && J.tlen > 0 && J.qlen > 0 && T16 > J.w + 32)
					cls = J.w <= 64 ? 1 : J.w <= 128 ? 2 : J.w <= 256 ? 3 : J.w <= 512 ? 4 : 0;	// cells per lane: (band + 31) / 64
				order[cls].push_back(i);
			}
			std::vector<u32> flat;
			for (auto& o : order) flat.insert(flat.end(), o.begin(), o.end());
			dOrder.reserve(flat.size());
			HIP_CHECK(hipMemcpyAsync(dOrder.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, s));
			HIP_CHECK(hipStreamSynchronize(s));		// flat goes out of scope below
			u32 at = 0;
			for (int cls = 0; cls < 5; ++cls)
			{
				const u32 cnt = (u32)order[cls].size();
				if (!cnt) continue;
				ScopedK t(c->timer, cls == 0 ? "k_ksw_extz2" : "k_ksw_extz2_lds");
				const unsigned g = std::min<unsigned>(cnt, 8192u);
				const u32* ord = dOrder.p + at;
#define KSW_ARGS dJobs.p, ord, cnt, dTrg.p, dQry.p, dScratch.p, dCig.p, dN.p, dDense.p, dTotal.p, dBase.p
				const u32 dbg = getenv("FG_KSW_DEBUG") ? (u32)atoi(getenv("FG_KSW_DEBUG")) : 0u;
				// state in registers (FG_KSW_REG=0: in LDS rings, the round-2 form)
				const bool regForm = !(getenv("FG_KSW_REG") && atoi(getenv("FG_KSW_REG")) == 0);
				if (cls == 0) hipLaunchKernelGGL(k_ksw_extz2, g, 64, 0, s, KSW_ARGS);
				else if (regForm && cls == 1) hipLaunchKernelGGL((k_ksw_extz2_reg<256, 2, 8, 96>), g, 64, 0, s, KSW_ARGS, dbg);
				else if (regForm && cls == 2) hipLaunchKernelGGL((k_ksw_extz2_reg<256, 3, 8, 160>), g, 64, 0, s, KSW_ARGS, dbg);
				else if (regForm && cls == 3) hipLaunchKernelGGL((k_ksw_extz2_reg<512, 5, 4, 288>), g, 64, 0, s, KSW_ARGS, dbg);
				else if (regForm) hipLaunchKernelGGL((k_ksw_extz2_reg<1024, 9, 2, 544>), g, 64, 0, s, KSW_ARGS, dbg);
				else if (cls == 1) hipLaunchKernelGGL((k_ksw_extz2_lds<256, 2, 8, 96>), g, 64, 0, s, KSW_ARGS, dbg);
				else if (cls == 2) hipLaunchKernelGGL((k_ksw_extz2_lds<256, 3, 8, 160>), g, 64, 0, s, KSW_ARGS, dbg);
				else if (cls == 3) hipLaunchKernelGGL((k_ksw_extz2_lds<512, 5, 4, 288>), g, 64, 0, s, KSW_ARGS, dbg);
				else hipLaunchKernelGGL((k_ksw_extz2_lds<1024, 9, 2, 544>), g, 64, 0, s, KSW_ARGS, dbg);
#undef KSW_ARGS
				at += cnt;
			}
		}
		std::vector<u32> hN(sub.size()), hBase(sub.size());
		u32 total = 0;
		HIP_CHECK(hipMemcpyAsync(hN.data(), dN.p, sub.size() * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipMemcpyAsync(hBase.data(), dBase.p, sub.size() * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipMemcpyAsync(&total, dTotal.p, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		std::vector<u32> hRuns(total + 1);
		if (total) HIP_CHECK(hipMemcpy(hRuns.data(), dDense.p, (size_t)total * 4, hipMemcpyDeviceToHost));
		auto t2 = now();
		if (trace) fprintf(stderr, "[ksw] sub-batch of %zu: setup %.1f ms, alloc+kernels+D2H %.1f ms (scratch %.2f GB, %u runs)\n",
						   sub.size(), ms(t0, t1), ms(t1, t2), (memTotal + pTotal) / 1e9, total);
		for (u32 i = 0; i < sub.size(); ++i)
		{
			runOff[a + i + 1] = runOff[a + i] + hN[i];
			runs.insert(runs.end(), hRuns.begin() + hBase[i], hRuns.begin() + hBase[i] + hN[i]);
		}
		a = b;
	}
	c->timer.collect();
	if (trace) fprintf(stderr, "[ksw] total %.1f ms\n", ms(t0, now()));
}
