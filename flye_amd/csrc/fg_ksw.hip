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

__global__ void __launch_bounds__(64)
k_ksw_extz2(const KswJob* __restrict__ jobs, u32 nJobs, const uint8_t* __restrict__ trgAll, const uint8_t* __restrict__ qryAll,
			uint8_t* __restrict__ scratch, u32* __restrict__ cigars, u32* __restrict__ nCigar)
{
	const int lane = threadIdx.x;
	for (u32 jb = blockIdx.x; jb < nJobs; jb += gridDim.x)
	{
		const KswJob J = jobs[jb];
		const int tlen = J.tlen, qlen = J.qlen, w = J.w;
		if (!J.feasible || tlen <= 0 || qlen <= 0) { if (lane == 0) nCigar[jb] = 0; continue; }
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
			nCigar[jb] = (u32)n;
		}
		ksw_fence();
	}
}

// does the band w connect (0, 0) with (tlen-1, qlen-1)?  (ksw2_extz2_sse.c: "if (st > en) zdropped")
bool bandFeasible(int qlen, int tlen, int w)
{
	for (int r = 0; r < qlen + tlen - 1; ++r)
	{
		int st = 0, en = tlen - 1;
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < ((r - w + 1) >> 1)) st = (r - w + 1) >> 1;
		if (en > ((r + w) >> 1)) en = (r + w) >> 1;
		if (st > en) return false;
	}
	return true;
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
	DevBuf<u32> dCig, dN;
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
		dScratch.reserve(memTotal + pTotal + 64);
		dJobs.reserve(sub.size()); dCig.reserve(cigTotal + 1); dN.reserve(sub.size());
		HIP_CHECK(hipMemsetAsync(dScratch.p, 0, memTotal + 64, s));		// the state buffers start zeroed (kcalloc)
		HIP_CHECK(hipMemcpyAsync(dJobs.p, sub.data(), sub.size() * sizeof(KswJob), hipMemcpyHostToDevice, s));
		{
			ScopedK t(c->timer, "k_ksw_extz2");
			const unsigned grid = (unsigned)std::min<size_t>(sub.size(), 8192);
			hipLaunchKernelGGL(k_ksw_extz2, grid, 64, 0, s, dJobs.p, (u32)sub.size(), dTrg.p, dQry.p, dScratch.p, dCig.p, dN.p);
		}
		std::vector<u32> hCig(cigTotal + 1), hN(sub.size());
		HIP_CHECK(hipMemcpyAsync(hN.data(), dN.p, sub.size() * 4, hipMemcpyDeviceToHost, s));
		if (cigTotal) HIP_CHECK(hipMemcpyAsync(hCig.data(), dCig.p, cigTotal * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		for (u32 i = 0; i < sub.size(); ++i)
		{
			runOff[a + i + 1] = runOff[a + i] + hN[i];
			// the device wrote the runs last to first
			for (u32 k = hN[i]; k-- > 0;) runs.push_back(hCig[sub[i].cigOff + k]);
		}
		a = b;
	}
	c->timer.collect();
}
