// Index build on the device: exact canonical k-mer counting, per-read solid
// k-mer selection, minimizer sketch, and the k-mer -> sorted position list index
// (CSR + open-addressing probe table).
//
// Behaviour restated from the reference (Flye 2.8.1), file:line in each kernel:
//   KmerCounter::count/getFreq          src/sequence/vertex_index.cpp:499-616
//   VertexIndex::yieldFrequentKmers     src/sequence/vertex_index.cpp:316-358
//   buildIndexUnevenCoverage            src/sequence/vertex_index.cpp:25-125
//   filterFrequentKmers                 src/sequence/vertex_index.cpp:173-212
//   buildIndexMinimizers                src/sequence/vertex_index.cpp:389-483
//   yieldMinimizers                     src/sequence/kmer.h:206-262
//
// Design (MI355X-first, not the reference's cuckoo-map-of-vectors):
//   * counting: one u32 counter per possible k-mer, direct addressed (4^k * 4 B =
//     68.7 GB at k = 17, sized for 288 GB HBM), one global atomic per k-mer, no CAS
//     loop, no overflow map, exact; freed after the build like _kmerCounter.clear();
//   * selection threshold: one workgroup per read, LDS histogram radix-select;
//   * index: accepted (k-mer, position) pairs are emitted unordered, ordered by two
//     stable LSD radix sorts (rocPRIM device primitive) and cut into a CSR by a
//     head-flag scan; lookups go through a linear-probing table of 16-byte slots
//     {key, offset<<24|count} at load <= 0.5, one dwordx4 load per probe.
#include "fg_ctx.h"

#include <rocprim/rocprim.hpp>

#define WG 256

namespace {

// ---------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T block_sum(T v, T* sh /* >= WG/64 */)
{
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
	const int w = threadIdx.x >> 6;
	__syncthreads();
	if ((threadIdx.x & 63) == 0) sh[w] = v;
	__syncthreads();
	T t = 0;
	if (threadIdx.x == 0) for (int i = 0; i < WG / 64; ++i) t += sh[i];
	return t;	// valid on thread 0
}

// vertex_index.cpp:520-558: one increment per canonical k-mer of every forward read
__global__ void k_count(const u64* __restrict__ words, const u64* __restrict__ wordOff,
						const i32* __restrict__ len, int k, u32* __restrict__ counts,
						unsigned long long* __restrict__ distinct)
{
	__shared__ u32 sh[WG / 64];
	const u32 r = blockIdx.x;
	const i32 nk = len[r] - k;
	const u64* w = words + wordOff[r];
	u32 local = 0;
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		u64 fw, rv;
		fg_kmer_pair(w, p, k, fw, rv);
		const u64 cn = fw < rv ? fw : rv;
		local += (atomicAdd(&counts[cn], 1u) == 0u);
	}
	u32 t = block_sum(local, sh);
	if (threadIdx.x == 0 && t) atomicAdd(distinct, (unsigned long long)t);
}

// KmerCounter::getFreq for every k-mer position (vertex_index.cpp:326-333)
__global__ void k_freq(const u64* __restrict__ words, const u64* __restrict__ wordOff,
					   const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
					   const u32* __restrict__ counts, u32* __restrict__ freq)
{
	const u32 r = blockIdx.x;
	const i32 nk = len[r] - k;
	const u64* w = words + wordOff[r];
	u32* f = freq + kmerOff[r];
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		u64 fw, rv;
		fg_kmer_pair(w, p, k, fw, rv);
		f[p] = counts[fw < rv ? fw : rv];
	}
}

// vertex_index.cpp:336-344: threshold = frequency at rank (size_t)(selectRate * n)
// of the descending order; all k-mers with freq >= threshold are kept.
#define HBINS 2048
__global__ void k_threshold(const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
							const u32* __restrict__ freq, float selectRate, u32* __restrict__ thr)
{
	__shared__ u32 hist[HBINS];
	__shared__ u32 shThr;
	__shared__ u32 shCnt[WG / 64];
	const u32 r = blockIdx.x;
	const i32 nk = len[r] - k;
	if (nk <= 0) { if (threadIdx.x == 0) thr[r] = 0; return; }
	const u32* f = freq + kmerOff[r];
	for (int i = threadIdx.x; i < HBINS; i += WG) hist[i] = 0;
	__syncthreads();
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		u32 v = f[p];
		atomicAdd(&hist[v < HBINS - 1 ? v : HBINS - 1], 1u);
	}
	__syncthreads();
	const u64 maxKmers = (u64)(selectRate * (float)(u64)nk);	// float multiply, truncation (:339)
	if (threadIdx.x == 0)
	{
		// largest v with #(freq >= v) > maxKmers
		u64 ge = (u64)nk;	// #(freq >= 0)
		u32 v = 0;
		while (v < HBINS - 1 && ge - hist[v] > maxKmers) { ge -= hist[v]; ++v; }
		shThr = v;
	}
	__syncthreads();
	if (shThr < HBINS - 1) { if (threadIdx.x == 0) thr[r] = shThr; return; }
	// rare: the threshold lies in the overflow bin -> bisect on the exact values
	u32 lo = HBINS - 1, hi = 0xFFFFFFFFu;	// invariant: #(f >= lo) > maxKmers
	while (lo < hi)
	{
		const u32 mid = lo + (u32)(((u64)hi - lo + 1) / 2);
		u32 c = 0;
		for (i32 p = threadIdx.x; p < nk; p += WG) c += (f[p] >= mid);
		u32 t = block_sum(c, shCnt);
		if (threadIdx.x == 0) shThr = t;
		__syncthreads();
		if ((u64)shThr > maxKmers) lo = mid; else hi = mid - 1;
		__syncthreads();
	}
	if (threadIdx.x == 0) thr[r] = lo;
}

// flags: bit0 selected (freq >= per-read threshold), bit1 tandem candidate
__global__ void k_mark(const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
					   const u32* __restrict__ freq, const u32* __restrict__ thr, i32 tandemFreq,
					   uint8_t* __restrict__ flags, unsigned long long* __restrict__ nCand)
{
	__shared__ u32 sh[WG / 64];
	const u32 r = blockIdx.x;
	const i32 nk = len[r] - k;
	if (nk <= 0) return;
	const u32* f = freq + kmerOff[r];
	uint8_t* fl = flags + kmerOff[r];
	const u32 t = thr[r];
	u32 c = 0;
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		const u32 v = f[p];
		uint8_t b = v >= t;
		if (b && tandemFreq > 0 && v > (u32)tandemFreq) { b |= 2; ++c; }
		fl[p] = b;
	}
	u32 tot = block_sum(c, sh);
	if (threadIdx.x == 0 && tot) atomicAdd(nCand, (unsigned long long)tot);
}

// vertex_index.cpp:346-355: k-mers occurring > tandemFreq times inside ONE read are
// dropped.  Only candidates (global freq > tandemFreq) can qualify; they are
// counted exactly in a (read, k-mer) keyed table.
__global__ void k_tandem_insert(const u64* __restrict__ words, const u64* __restrict__ wordOff,
								const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
								const uint8_t* __restrict__ flags, u64* __restrict__ tkeys,
								u32* __restrict__ tcnt, u64 tmask)
{
	const u32 r = blockIdx.x;
	const i32 nk = len[r] - k;
	const u64* w = words + wordOff[r];
	const uint8_t* fl = flags + kmerOff[r];
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		if (!(fl[p] & 2)) continue;
		u64 fw, rv;
		fg_kmer_pair(w, p, k, fw, rv);
		const u64 key = ((u64)r << 34) | (fw < rv ? fw : rv);
		u64 h = fg_mix(key) & tmask;
		while (true)
		{
			u64 old = atomicCAS((unsigned long long*)&tkeys[h], (unsigned long long)FG_EMPTY_KEY,
								(unsigned long long)key);
			if (old == FG_EMPTY_KEY || old == key) { atomicAdd(&tcnt[h], 1u); break; }
			h = (h + 1) & tmask;
		}
	}
}

__global__ void k_tandem_apply(const u64* __restrict__ words, const u64* __restrict__ wordOff,
							   const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
							   uint8_t* __restrict__ flags, const u64* __restrict__ tkeys,
							   const u32* __restrict__ tcnt, u64 tmask, i32 tandemFreq)
{
	const u32 r = blockIdx.x;
	const i32 nk = len[r] - k;
	const u64* w = words + wordOff[r];
	uint8_t* fl = flags + kmerOff[r];
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		if (!(fl[p] & 2)) continue;
		u64 fw, rv;
		fg_kmer_pair(w, p, k, fw, rv);
		const u64 key = ((u64)r << 34) | (fw < rv ? fw : rv);
		u64 h = fg_mix(key) & tmask;
		while (tkeys[h] != key) h = (h + 1) & tmask;
		if (tcnt[h] > (u32)tandemFreq) fl[p] = 0;
	}
}

// bit0 := accepted for the index (selected, not tandem, freq >= minFreq)
__global__ void k_accept(const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
						 const u32* __restrict__ freq, i32 minFreq, uint8_t* __restrict__ flags,
						 unsigned long long* __restrict__ nAcc)
{
	__shared__ u32 sh[WG / 64];
	const u32 r = blockIdx.x;
	const i32 nk = len[r] - k;
	if (nk <= 0) return;
	const u32* f = freq + kmerOff[r];
	uint8_t* fl = flags + kmerOff[r];
	u32 c = 0;
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		uint8_t b = (fl[p] & 1) && f[p] >= (u32)minFreq;
		fl[p] = b;
		c += b;
	}
	u32 tot = block_sum(c, sh);
	if (threadIdx.x == 0 && tot) atomicAdd(nAcc, (unsigned long long)tot);
}

// ---- minimizer sketch (kmer.h:206-262) ------------------------------------------
// The sketch equals "the sequence of distinct deque fronts".  Whenever hash[p] is
// strictly below the w previous hashes (or p == 0) the deque collapses to [p]
// whatever came before (a sync point), so the stretch up to the next sync point
// is an independent piece: sync points are found in parallel, then one lane runs
// the literal deque loop over one piece.
#define MAXW 64
__global__ void k_minimizers(const u64* __restrict__ words, const u64* __restrict__ wordOff,
							 const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k, int w,
							 u64* __restrict__ hashes /* per k-mer position scratch */,
							 uint8_t* __restrict__ flags, unsigned long long* __restrict__ nAcc)
{
	__shared__ u32 sh[WG / 64];
	const u32 r = blockIdx.x;
	const i32 nk = len[r] - k;
	if (nk <= 0) return;
	const u64* wd = words + wordOff[r];
	u64* hs = hashes + kmerOff[r];
	uint8_t* fl = flags + kmerOff[r];
	if (w == 1)
	{
		u32 c = 0;
		for (i32 p = threadIdx.x; p < nk; p += WG) { fl[p] = 1; ++c; }
		u32 tot = block_sum(c, sh);
		if (threadIdx.x == 0) atomicAdd(nAcc, (unsigned long long)tot);
		return;
	}
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		u64 fw, rv;
		fg_kmer_pair(wd, p, k, fw, rv);
		hs[p] = fg_kmer_hash(fw < rv ? fw : rv);
		fl[p] = 0;
	}
	__syncthreads();
	u32 c = 0;
	for (i32 p0 = threadIdx.x; p0 < nk; p0 += WG)
	{
		// is p0 a sync point?
		const u64 h0 = hs[p0];
		bool sync = true;
		for (i32 j = p0 - 1; j >= 0 && j >= p0 - w; --j)
			if (hs[j] <= h0) { sync = false; break; }
		if (!sync) continue;
		// literal deque from an empty state, until the next sync point
		i32 qpos[MAXW + 2]; u64 qh[MAXW + 2];
		int head = 0, tail = 0;	// ring of capacity MAXW+2
		const int CAP = MAXW + 2;
		i32 lastEmit = -1;
		for (i32 p = p0; p < nk; ++p)
		{
			const u64 h = hs[p];
			if (p > p0)
			{
				// stop at the next sync point (it starts its own piece)
				bool s2 = true;
				for (i32 j = p - 1; j >= 0 && j >= p - w; --j)
					if (hs[j] <= h) { s2 = false; break; }
				if (s2) break;
			}
			while (tail != head && qh[(tail + CAP - 1) % CAP] > h) tail = (tail + CAP - 1) % CAP;
			qpos[tail] = p; qh[tail] = h; tail = (tail + 1) % CAP;
			if (qpos[head] <= p - w)
			{
				while (qpos[head] <= p - w) head = (head + 1) % CAP;
				while ((tail + CAP - head) % CAP >= 2 && qh[head] == qh[(head + 1) % CAP]) head = (head + 1) % CAP;
			}
			if (lastEmit != qpos[head])
			{
				lastEmit = qpos[head];
				if (!fl[lastEmit]) { fl[lastEmit] = 1; ++c; }
			}
		}
	}
	u32 tot = block_sum(c, sh);
	if (threadIdx.x == 0 && tot) atomicAdd(nAcc, (unsigned long long)tot);
}

// canonical-orientation entry of every accepted position (vertex_index.cpp:76-85):
// value = (record << posBits) | position with record = 2*read (+1 if the k-mer was
// flipped, position mirrored)
__global__ void k_emit(const u64* __restrict__ words, const u64* __restrict__ wordOff,
					   const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
					   const uint8_t* __restrict__ flags, int posBits, u64 keyLo, u64 keyHi /* canonical k-mers in [lo, hi) */,
					   u64* __restrict__ ecanon, u64* __restrict__ evalue, unsigned long long* __restrict__ cursor)
{
	const u32 r = blockIdx.x;
	const i32 L = len[r];
	const i32 nk = L - k;
	const u64* w = words + wordOff[r];
	const uint8_t* fl = flags + kmerOff[r];
	const bool all = keyLo == 0 && keyHi == ~0ULL;
	auto inSlice = [&](i32 p) -> bool
	{
		if (all) return true;
		u64 fw, rv;
		fg_kmer_pair(w, p, k, fw, rv);
		const u64 c = fw < rv ? fw : rv;
		return c >= keyLo && c < keyHi;
	};
	// ONE cursor atomic per read: same-address atomics serialise at ~11 ns each, and one per
	// wave and step (what the compiler's aggregation of a per-element atomicAdd gives) was 3.4 M of
	// them = 39 of the build's 85 ms.  The emission order is irrelevant (radix sort follows).
	__shared__ u32 shw[WG / 64];
	__shared__ u64 shBase;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	u32 mine = 0;
	for (i32 p = threadIdx.x; p < nk; p += WG) mine += (fl[p] && inSlice(p)) ? 1u : 0u;
	for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
	if (lane == 0) shw[wv] = mine;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		u32 tot = 0;
		for (int i = 0; i < WG / 64; ++i) tot += shw[i];
		shBase = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ULL;
	}
	__syncthreads();
	u64 base = shBase;
	for (i32 p0 = 0; p0 < nk; p0 += WG)
	{
		const i32 p = p0 + (i32)threadIdx.x;
		const bool take = p < nk && fl[p] && inSlice(p);
		const u64 m = __ballot(take);
		__syncthreads();
		if (lane == 0) shw[wv] = (u32)__popcll(m);
		__syncthreads();
		u32 before = 0, tot = 0;
		for (int i = 0; i < WG / 64; ++i) { const u32 c = shw[i]; if (i < wv) before += c; tot += c; }
		if (take)
		{
			u64 fw, rv;
			fg_kmer_pair(w, p, k, fw, rv);
			const bool flip = rv < fw;
			const u64 slot = base + before + (u64)__popcll(m & ((lane == 0) ? 0ULL : (~0ULL >> (64 - lane))));
			ecanon[slot] = flip ? rv : fw;
			evalue[slot] = ((u64)(2 * r + (flip ? 1 : 0)) << posBits) | (u64)(flip ? L - p - k : p);
		}
		base += tot;
	}
}

// accepted positions per key bin (bin = canonical k-mer >> binShift): what the slicing of the build --
// by memory on one GPU, by rank on several -- balances on
#define FG_INDEX_BINS 4096
__global__ void k_bin_hist(const u64* __restrict__ words, const u64* __restrict__ wordOff,
						   const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
						   const uint8_t* __restrict__ flags, int binShift, unsigned long long* __restrict__ hist)
{
	__shared__ u32 sh[FG_INDEX_BINS];
	for (int i = threadIdx.x; i < FG_INDEX_BINS; i += WG) sh[i] = 0;
	__syncthreads();
	const u32 r = blockIdx.x;
	const i32 nk = len[r] - k;
	const u64* w = words + wordOff[r];
	const uint8_t* fl = flags + kmerOff[r];
	for (i32 p = threadIdx.x; p < nk; p += WG)
		if (fl[p])
		{
			u64 fw, rv;
			fg_kmer_pair(w, p, k, fw, rv);
			atomicAdd(&sh[(fw < rv ? fw : rv) >> binShift], 1u);
		}
	__syncthreads();
	for (int i = threadIdx.x; i < FG_INDEX_BINS; i += WG)
		if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

__global__ void k_heads(const u64* __restrict__ c, u64 n, u32* __restrict__ flag)
{
	const u64 i = (u64)blockIdx.x * WG + threadIdx.x;
	if (i < n) flag[i] = (i == 0 || c[i] != c[i - 1]) ? 1u : 0u;
}

__global__ void k_keys(const u64* __restrict__ c, u64 n, const u32* __restrict__ flag,
					   const u32* __restrict__ inc, u64* __restrict__ ukeys, u64* __restrict__ kstart)
{
	const u64 i = (u64)blockIdx.x * WG + threadIdx.x;
	if (i < n && flag[i]) { ukeys[inc[i] - 1] = c[i]; kstart[inc[i] - 1] = i; }
}

// filterFrequentKmers sums (vertex_index.cpp:175-184)
__global__ void k_capstats(const u64* __restrict__ kstart, u64 nKeys, i32 minCoverage,
						   unsigned long long* __restrict__ out /* total, unique */)
{
	__shared__ u64 sh[WG / 64];
	const u64 j = (u64)blockIdx.x * WG + threadIdx.x;
	u64 cap = 0, uq = 0;
	if (j < nKeys)
	{
		cap = kstart[j + 1] - kstart[j];
		if (cap >= (u64)minCoverage) uq = 1; else cap = 0;
	}
	u64 t = block_sum(cap, sh);
	if (threadIdx.x == 0 && t) atomicAdd(&out[0], (unsigned long long)t);
	u64 u = block_sum(uq, sh);
	if (threadIdx.x == 0 && u) atomicAdd(&out[1], (unsigned long long)u);
}

// vertex_index.cpp:189-202 (repetitive keys leave the index), :70-71 (entries are
// written only when minFreq <= freq <= repFreq), :370-373 (capacity limit)
__global__ void k_classify(const u64* __restrict__ ukeys, const u64* __restrict__ kstart, u64 nKeys,
						   u64 repFreq, const u32* __restrict__ counts /* null in minimizer mode */,
						   u32* __restrict__ isRep, u32* __restrict__ keep, u64* __restrict__ size,
						   u32* __restrict__ err)
{
	const u64 j = (u64)blockIdx.x * WG + threadIdx.x;
	if (j >= nKeys) return;
	const u64 cap = kstart[j + 1] - kstart[j];
	const bool rep = cap > repFreq;
	bool filled = !rep;
	if (filled && counts) filled = (u64)counts[ukeys[j]] <= repFreq;
	isRep[j] = rep;
	keep[j] = !rep;
	size[j] = filled ? cap : 0;
	if (!rep && cap + 1 > (u64)(32 * 1024 * 1024 / 5)) *err = 1;
}

// totals of one part under the final repFreq: repetitive keys, kept keys, entries (same rules as k_classify)
__global__ void k_classify_count(const u64* __restrict__ ukeys, const u64* __restrict__ kstart, u64 nKeys, u64 repFreq,
								 const u32* __restrict__ counts, unsigned long long* __restrict__ out /* rep, keep, entries */)
{
	__shared__ u64 sh[WG / 64];
	const u64 j = (u64)blockIdx.x * WG + threadIdx.x;
	u64 rep = 0, keep = 0, ent = 0;
	if (j < nKeys)
	{
		const u64 cap = kstart[j + 1] - kstart[j];
		const bool r = cap > repFreq;
		bool filled = !r;
		if (filled && counts) filled = (u64)counts[ukeys[j]] <= repFreq;
		rep = r; keep = !r; ent = filled ? cap : 0;
	}
	u64 t = block_sum(rep, sh);
	if (threadIdx.x == 0 && t) atomicAdd(&out[0], (unsigned long long)t);
	t = block_sum(keep, sh);
	if (threadIdx.x == 0 && t) atomicAdd(&out[1], (unsigned long long)t);
	t = block_sum(ent, sh);
	if (threadIdx.x == 0 && t) atomicAdd(&out[2], (unsigned long long)t);
}

// one part's keys / offsets / repetitive keys into the final arrays, behind what earlier parts wrote
__global__ void k_finalize(const u64* __restrict__ ukeys, u64 nKeys, const u32* __restrict__ isRep,
						   const u32* __restrict__ repIdx, const u32* __restrict__ keepIdx,
						   const u64* __restrict__ off, u64 keepBase, u64 repBase, u64 entBase,
						   u64* __restrict__ keys, u64* __restrict__ keyOff, u64* __restrict__ repKeys)
{
	const u64 j = (u64)blockIdx.x * WG + threadIdx.x;
	if (j >= nKeys) return;
	if (isRep[j]) repKeys[repBase + repIdx[j]] = ukeys[j];
	else { keys[keepBase + keepIdx[j]] = ukeys[j]; keyOff[keepBase + keepIdx[j]] = entBase + off[j]; }
}

__global__ void k_entries(const u64* __restrict__ evalue, u64 n, const u32* __restrict__ inc,
						  const u64* __restrict__ kstart, const u64* __restrict__ size,
						  const u64* __restrict__ off, int posBits, u64 entBase, u64* __restrict__ entries)
{
	const u64 i = (u64)blockIdx.x * WG + threadIdx.x;
	if (i >= n) return;
	const u32 j = inc[i] - 1;
	if (size[j] == 0) return;
	const u64 v = evalue[i];
	entries[entBase + off[j] + (i - kstart[j])] = ((v >> posBits) << 32) | (v & ((1ULL << posBits) - 1));
}

// keys with a non-empty list (by their index in the sorted key array) and the repetitive keys -> slots
template <bool WIDE>
__global__ void k_table_insert(const u64* __restrict__ keys, const u64* __restrict__ keyOff, u64 nKeys,
							   const u64* __restrict__ repKeys, u64 nRep, FgTable T, u64* __restrict__ slots,
							   ulonglong2* __restrict__ wide)
{
	const u64 j = (u64)blockIdx.x * WG + threadIdx.x;
	u64 key, idx;
	if (j < nKeys)
	{
		if (keyOff[j + 1] == keyOff[j]) return;	// an empty list behaves like an absent key (overlap.cpp:183)
		key = keys[j]; idx = j;
	}
	else if (j < nKeys + nRep) { key = repKeys[j - nKeys]; idx = FG_EMPTY_KEY; }
	else return;
	u32 p = 0;
	while (p + 1 < T.nParts && key >= T.bound[p + 1]) ++p;
	const u64 mix = fg_mix(key);
	const u32 slotsInPart = T.groups[p] * 8u;
	u32 h = __umulhi((u32)(mix >> 32), T.groups[p]) * 8u + ((u32)mix & 7u);
	if (WIDE)
	{
		while (true)
		{
			const u64 old = atomicCAS((unsigned long long*)&wide[T.slotBase[p] + h].x, (unsigned long long)FG_EMPTY_KEY,
									  (unsigned long long)key);
			if (old == FG_EMPTY_KEY) { wide[T.slotBase[p] + h].y = idx; break; }
			h = h + 1 == slotsInPart ? 0 : h + 1;
		}
	}
	else
	{
		// group by group from the key's own: the first empty slot in group order takes it (a group is full
		// before anything spills into the next one -- what lets a probe stop at a group with an empty slot)
		const u64 v = (key << FG_IDX_BITS) | (idx == FG_EMPTY_KEY ? FG_IDX_MASK : (idx - T.keyBase[p]));
		h &= ~7u;
		while (true)
		{
			const u64 old = atomicCAS((unsigned long long*)&slots[T.slotBase[p] + h], (unsigned long long)FG_EMPTY_KEY,
									  (unsigned long long)v);
			if (old == FG_EMPTY_KEY) break;
			h = h + 1 == slotsInPart ? 0 : h + 1;
		}
	}
}

// one bit per forward k-mer position: does this position own an index entry?
// (lets the seed collector skip the trivial self hit, overlap.cpp:188-190,
// without searching the list)
__global__ void k_indexed_bits(const u64* __restrict__ words, const u64* __restrict__ wordOff,
							   const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
							   const uint8_t* __restrict__ flags /* the build's selection, or null */,
							   FgTable T, int wide, const u64* __restrict__ entries, u32* __restrict__ bits)
{
	const u32 r = blockIdx.x;
	const i32 L = len[r];
	const i32 nk = L - k;
	const u64* w = words + wordOff[r];
	const uint8_t* fl = flags ? flags + kmerOff[r] : nullptr;
	const u64 base = kmerOff[r];
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		if (fl && !fl[p]) continue;
		u64 fw, rv;
		fg_kmer_pair(w, p, k, fw, rv);
		const bool flip = rv < fw;
		const u64 v = wide ? fg_probe<true>(T, flip ? rv : fw) : fg_probe<false>(T, flip ? rv : fw);
		if (v == 0 || (v & FG_CNT_MASK) == FG_CNT_REPETITIVE) continue;
		if (!fl)
		{
			// an imported index carries no selection flags: a position owns an entry iff its own
			// (record, position) is in the k-mer's list (ascending, vertex_index.cpp:108-114)
			const u64 own = ((u64)(2 * r + (flip ? 1u : 0u)) << 32) | (u32)(flip ? L - p - k : p);
			const u64* e = entries + ((v >> FG_CNT_BITS) & ((1ULL << 38) - 1));
			u32 lo = 0, hi = (u32)(v & FG_CNT_MASK);
			while (lo < hi) { const u32 m = (lo + hi) >> 1; if (e[m] < own) lo = m + 1; else hi = m; }
			if (lo >= (u32)(v & FG_CNT_MASK) || e[lo] != own) continue;
		}
		atomicOr(&bits[(base + p) >> 5], 1u << ((base + p) & 31));
	}
}

// ---- host helpers -----------------------------------------------------------------
struct Prim {
	fg_ctx* c;
	DevBuf<char> tmp;
	void sortPairs(u64* kin, u64* kout, u64* vin, u64* vout, u64 n, int bits)
	{
		if (n == 0) return;
		size_t bytes = 0;
		HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, bits, c->stream));
		tmp.reserve(bytes);
		HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, n, 0, bits, c->stream));
	}
	template <class T>
	void incScan(const T* in, T* out, u64 n)
	{
		if (n == 0) return;
		size_t bytes = 0;
		HIP_CHECK(rocprim::inclusive_scan(nullptr, bytes, in, out, n, rocprim::plus<T>(), c->stream));
		tmp.reserve(bytes);
		HIP_CHECK(rocprim::inclusive_scan(tmp.p, bytes, in, out, n, rocprim::plus<T>(), c->stream));
	}
	template <class T>
	void excScan(const T* in, T* out, u64 n)
	{
		if (n == 0) return;
		size_t bytes = 0;
		HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), c->stream));
		tmp.reserve(bytes);
		HIP_CHECK(rocprim::exclusive_scan(tmp.p, bytes, in, out, T(0), n, rocprim::plus<T>(), c->stream));
	}
};

template <class T>
T fetch(fg_ctx* c, const T* dptr)
{
	T v;
	HIP_CHECK(hipMemcpyAsync(&v, dptr, sizeof(T), hipMemcpyDeviceToHost, c->stream));
	HIP_CHECK(hipStreamSynchronize(c->stream));
	return v;
}

int bitsFor(u64 v) { int b = 1; while ((1ULL << b) <= v && b < 63) ++b; return b; }
unsigned gridFor(u64 n) { return (unsigned)((n + WG - 1) / WG); }


// ---- the build in three steps ------------------------------------------------------------------
//   begin        k-mer selection over ALL reads (solid: exact counts + per-read frequency threshold +
//                tandem filter, vertex_index.cpp:19-125; minimizers: kmer.h:206-262) -> one flag per
//                k-mer position, and the number of accepted positions per key bin
//   build range  for the keys of bins [lo, hi): emit (canonical k-mer, position) pairs, two stable
//                LSD radix sorts (position, then k-mer), run-length encode -> one PART (unique keys,
//                list starts, sorted positions) + its share of filterFrequentKmers' sums
//                (vertex_index.cpp:175-184).  Ranges wider than the memory budget are cut.
//   finish       with the sums over ALL keys: repetitive frequency, classification, CSR arrays in key
//                order, probe table, indexed bits.
// One GPU runs begin, the whole key space, finish.  Several GPUs each run begin (replicated: the
// selection needs every read), the range their rank owns (flye_amd/dist.py balances the ranges on
// the bin histogram), exchange the two sums, finish their piece, all-gather the pieces and import the
// concatenation (fgImportIndex) -- SURVEY.md §8(e).
struct IndexPart {
	DevBuf<u64> ukeys, kstart, evalue;
	DevBuf<u32> inc;
	u64 nKeys = 0, E = 0;
};

struct IndexBuild {
	bool solid = false;
	DevBuf<u32> counts;			// solid mode: exact count of every possible k-mer
	DevBuf<uint8_t> flags;		// 1 = this k-mer position contributes an entry
	int posBits = 0, binShift = 0;
	i32 minCoverage = 0;
	float repeatRate = 0, sampleRateInit = 1.0f;
	u64 totalDistinct = 0;
	std::vector<u64> hist;		// accepted positions per bin
	std::vector<std::unique_ptr<IndexPart>> parts;	// ascending key ranges
	unsigned long long sums[2] = {0, 0};
	double seconds = 0;
};

struct BuildClock {
	fg_ctx* c; hipEvent_t a, b;
	BuildClock(fg_ctx* c_) : c(c_), a(c_->timer.get()), b(nullptr)
	{
		try { b = c_->timer.get(); } catch (...) { c_->timer.pool.push_back(a); throw; }
		HIP_CHECK(hipEventRecord(a, c->stream));
	}
	double stop()
	{
		HIP_CHECK(hipEventRecord(b, c->stream));
		HIP_CHECK(hipEventSynchronize(b));
		float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
		return ms * 1e-3;
	}
	~BuildClock() { c->timer.pool.push_back(a); c->timer.pool.push_back(b); }
};

IndexBuild* buildState(fg_ctx* c)
{
	if (!c->indexBuild) throw FgError{FG_ERR_STATE, "no index build in progress (call the begin step first)"};
	return (IndexBuild*)c->indexBuild.get();
}

void clearIndex(fg_ctx* c)
{
	c->indexBuilt = false;
	c->dKeys.release(); c->dKeyOff.release(); c->dEntries.release(); c->dRepKeys.release();
	c->dTable.release(); c->dIndexedBits.release();
	c->nKeys = c->nEntries = c->nRep = c->tableSlots = 0;
}

void binHistogram(fg_ctx* c, IndexBuild* B, u64* histOut)
{
	hipStream_t s = c->stream;
	const int k = c->k;
	B->binShift = std::max(0, 2 * k - 12);
	DevBuf<unsigned long long> dh;
	dh.alloc(FG_INDEX_BINS);
	HIP_CHECK(hipMemsetAsync(dh.p, 0, FG_INDEX_BINS * 8, s));
	if (c->nReads)
	{
		ScopedK t(c->timer, "k_bin_hist");
		hipLaunchKernelGGL(k_bin_hist, c->nReads, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, k,
						   B->flags.p, B->binShift, dh.p);
	}
	B->hist.assign(FG_INDEX_BINS, 0);
	HIP_CHECK(hipMemcpyAsync(B->hist.data(), dh.p, FG_INDEX_BINS * 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	if (histOut) memcpy(histOut, B->hist.data(), FG_INDEX_BINS * 8);
}

// one slice of keys -> one part
void buildPart(fg_ctx* c, IndexBuild* B, Prim& prim, u32 binLo, u32 binHi, u64 E)
{
	if (E == 0) return;
	hipStream_t s = c->stream;
	const int k = c->k;
	const u32 n = c->nReads;
	const u64 keyLo = (u64)binLo << B->binShift;
	const u64 keyHi = binHi >= FG_INDEX_BINS ? ~0ULL : ((u64)binHi << B->binShift);
	std::unique_ptr<IndexPart> part(new IndexPart);
	DevBuf<u64> ecanon, ecanon2, evalue2;
	DevBuf<unsigned long long> cursor;
	ecanon.alloc(E); part->evalue.alloc(E); cursor.alloc(1);
	HIP_CHECK(hipMemsetAsync(cursor.p, 0, 8, s));
	{
		ScopedK t(c->timer, "k_emit");
		hipLaunchKernelGGL(k_emit, n, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, k, B->flags.p,
						   B->posBits, (binLo == 0 && binHi >= FG_INDEX_BINS) ? 0ULL : keyLo,
						   (binLo == 0 && binHi >= FG_INDEX_BINS) ? ~0ULL : keyHi, ecanon.p, part->evalue.p, cursor.p);
	}
	if (fetch(c, cursor.p) != E) throw FgError{FG_ERR_HIP, "internal: slice emission does not match the bin histogram"};
	ecanon2.alloc(E); evalue2.alloc(E);
	const int valBits = B->posBits + bitsFor(2ULL * n);
	{
		ScopedK t(c->timer, "radix_sort_pairs(rocprim)");
		prim.sortPairs(part->evalue.p, evalue2.p, ecanon.p, ecanon2.p, E, valBits);
		prim.sortPairs(ecanon2.p, ecanon.p, evalue2.p, part->evalue.p, E, 2 * k);
	}
	ecanon2.release(); evalue2.release();
	// run-length encode the sorted k-mers
	DevBuf<u32> flag;
	flag.alloc(E); part->inc.alloc(E);
	{ ScopedK t(c->timer, "k_heads"); hipLaunchKernelGGL(k_heads, gridFor(E), WG, 0, s, ecanon.p, E, flag.p); }
	{ ScopedK t(c->timer, "scan(rocprim)"); prim.incScan(flag.p, part->inc.p, E); }
	const u64 nKeys = fetch(c, part->inc.p + (E - 1));
	part->ukeys.alloc(nKeys); part->kstart.alloc(nKeys + 1);
	{
		ScopedK t(c->timer, "k_keys");
		hipLaunchKernelGGL(k_keys, gridFor(E), WG, 0, s, ecanon.p, E, flag.p, part->inc.p, part->ukeys.p, part->kstart.p);
	}
	HIP_CHECK(hipMemcpyAsync(part->kstart.p + nKeys, &E, 8, hipMemcpyHostToDevice, s));
	// filterFrequentKmers' two integer sums (its two float operations run on the host in the finish step)
	DevBuf<unsigned long long> sums;
	sums.alloc(2);
	HIP_CHECK(hipMemsetAsync(sums.p, 0, 16, s));
	{
		ScopedK t(c->timer, "k_capstats");
		hipLaunchKernelGGL(k_capstats, gridFor(nKeys), WG, 0, s, part->kstart.p, nKeys, B->minCoverage, sums.p);
	}
	unsigned long long hs[2];
	HIP_CHECK(hipMemcpyAsync(hs, sums.p, 16, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	B->sums[0] += hs[0]; B->sums[1] += hs[1];
	part->nKeys = nKeys; part->E = E;
	B->parts.push_back(std::move(part));
}

} // namespace

void fgIndexBuildRange(fg_ctx* c, u32 binLo, u32 binHi, unsigned long long* sumsOut)
{
	IndexBuild* B = buildState(c);
	if (binLo > binHi || binHi > FG_INDEX_BINS) throw FgError{FG_ERR_ARG, "bin range outside [0, 4096]"};
	BuildClock clock(c);
	Prim prim{c};
	// slices of at most `budget` entries (sort scratch = 4 x 8 bytes per entry of the slice)
	const u64 budget = getenv("FG_INDEX_SLICE_ENTRIES") ? strtoull(getenv("FG_INDEX_SLICE_ENTRIES"), nullptr, 10) : (768ULL << 20);
	u32 lo = binLo;
	while (lo < binHi)
	{
		u32 hi = lo;
		u64 e = 0;
		while (hi < binHi && (hi == lo || e + B->hist[hi] <= budget)) { e += B->hist[hi]; ++hi; }
		buildPart(c, B, prim, lo, hi, e);
		lo = hi;
	}
	if (sumsOut) { sumsOut[0] = B->sums[0]; sumsOut[1] = B->sums[1]; }
	B->seconds += clock.stop();
}

// the index over what the parts hold: keys ascending, lists ascending (record, position)
void fgIndexFinish(fg_ctx* c, const unsigned long long* totalSums, fg_index_stats* st)
{
	IndexBuild* B = buildState(c);
	hipStream_t s = c->stream;
	BuildClock clock(c);
	Prim prim{c};
	memset(st, 0, sizeof(*st));
	const unsigned long long t0 = totalSums ? totalSums[0] : B->sums[0], t1 = totalSums ? totalSums[1] : B->sums[1];
	// vertex_index.cpp:185-186, the two float operations exactly as written there
	size_t totalKmers = t0, uniqueKmers = t1;
	float meanFrequency = (float)totalKmers / (uniqueKmers + 1);
	size_t repFreq = B->repeatRate * meanFrequency;
	st->mean_frequency = meanFrequency;
	st->repetitive_frequency = repFreq;
	st->total_kmers = B->totalDistinct;

	// totals first (the final arrays are allocated once), then part by part behind one another
	DevBuf<unsigned long long> tot;
	tot.alloc(3);
	HIP_CHECK(hipMemsetAsync(tot.p, 0, 24, s));
	for (auto& p : B->parts)
	{
		ScopedK t(c->timer, "k_classify");
		hipLaunchKernelGGL(k_classify_count, gridFor(p->nKeys), WG, 0, s, p->ukeys.p, p->kstart.p, p->nKeys, (u64)repFreq,
						   B->solid ? B->counts.p : (const u32*)nullptr, tot.p);
	}
	unsigned long long ht[3];
	HIP_CHECK(hipMemcpyAsync(ht, tot.p, 24, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	const u64 nRep = ht[0], nKeep = ht[1], nEnt = ht[2];
	clearIndex(c);
	c->dKeys.alloc(nKeep); c->dKeyOff.alloc(nKeep + 1); c->dEntries.alloc(nEnt); c->dRepKeys.alloc(nRep);
	u64 keepBase = 0, repBase = 0, entBase = 0;
	DevBuf<u32> err;
	err.alloc(1);
	HIP_CHECK(hipMemsetAsync(err.p, 0, 4, s));
	for (auto& p : B->parts)
	{
		const u64 nKeys = p->nKeys;
		DevBuf<u32> isRep, keep, repIdx, keepIdx;
		DevBuf<u64> size, off;
		isRep.alloc(nKeys + 1); keep.alloc(nKeys + 1); repIdx.alloc(nKeys + 1); keepIdx.alloc(nKeys + 1);
		size.alloc(nKeys + 1); off.alloc(nKeys + 1);
		// one extra zero element so that the exclusive scans also yield the totals
		HIP_CHECK(hipMemsetAsync(isRep.p + nKeys, 0, 4, s));
		HIP_CHECK(hipMemsetAsync(keep.p + nKeys, 0, 4, s));
		HIP_CHECK(hipMemsetAsync(size.p + nKeys, 0, 8, s));
		{
			ScopedK t(c->timer, "k_classify");
			hipLaunchKernelGGL(k_classify, gridFor(nKeys), WG, 0, s, p->ukeys.p, p->kstart.p, nKeys, (u64)repFreq,
							   B->solid ? B->counts.p : (const u32*)nullptr, isRep.p, keep.p, size.p, err.p);
		}
		{
			ScopedK t(c->timer, "scan(rocprim)");
			prim.excScan(isRep.p, repIdx.p, nKeys + 1);
			prim.excScan(keep.p, keepIdx.p, nKeys + 1);
			prim.excScan(size.p, off.p, nKeys + 1);
		}
		const u64 pRep = fetch(c, repIdx.p + nKeys), pKeep = fetch(c, keepIdx.p + nKeys), pEnt = fetch(c, off.p + nKeys);
		{
			ScopedK t(c->timer, "k_finalize");
			hipLaunchKernelGGL(k_finalize, gridFor(nKeys), WG, 0, s, p->ukeys.p, nKeys, isRep.p, repIdx.p, keepIdx.p, off.p,
							   keepBase, repBase, entBase, c->dKeys.p, c->dKeyOff.p, c->dRepKeys.p);
		}
		{
			ScopedK t(c->timer, "k_entries");
			hipLaunchKernelGGL(k_entries, gridFor(p->E), WG, 0, s, p->evalue.p, p->E, p->inc.p, p->kstart.p, size.p, off.p,
							   B->posBits, entBase, c->dEntries.p);
		}
		HIP_CHECK(hipStreamSynchronize(s));
		keepBase += pKeep; repBase += pRep; entBase += pEnt;
		p.reset();	// this part's memory goes before the next one's scratch comes
	}
	B->parts.clear();
	if (keepBase != nKeep || repBase != nRep || entBase != nEnt) throw FgError{FG_ERR_HIP, "internal: part totals disagree"};
	if (fetch(c, err.p)) { clearIndex(c); throw FgError{FG_ERR_KMER_TOO_FREQUENT, "k-mer is too frequent"}; }
	HIP_CHECK(hipMemcpyAsync(c->dKeyOff.p + nKeep, &nEnt, 8, hipMemcpyHostToDevice, s));
	B->counts.release();
	c->nKeys = nKeep; c->nEntries = nEnt; c->nRep = nRep;
	fgIndexLookupStructures(c, B->flags.p);
	st->selected_kmers = nKeep;
	st->index_entries = nEnt;
	st->repetitive_kmers = nRep;
	if (B->solid) c->sampleRate = B->sampleRateInit;
	else
	{
		// vertex_index.cpp:480-482: _sampleRate = (float)totalLen / totalEntries
		size_t totalLen = c->totalBases, totalEntries = c->nEntries;
		c->sampleRate = (float)totalLen / totalEntries;
	}
	st->sample_rate = c->sampleRate;
	st->build_seconds = B->seconds + clock.stop();
	c->indexBuild.reset();
	c->timer.collect();
}

// probe table (load <= 0.5) + one "owns an entry" bit per forward k-mer position, from the CSR arrays in
// the context.  flags: the build's per-position selection, or null (imported index: the lists are searched)
void fgIndexLookupStructures(fg_ctx* c, const uint8_t* flags)
{
	hipStream_t s = c->stream;
	const u64 nKeep = c->nKeys, nRep = c->nRep;
	const bool wide = 2 * c->k > 64 - FG_IDX_BITS;
	FgTable T;
	memset(&T, 0, sizeof(T));
	T.keyOff = (const unsigned long long*)c->dKeyOff.p;
	// parts of equal key count along the sorted key array; the repetitive keys fall into the part their value lies in
	const u64 perPart = wide ? ~0ULL : (getenv("FG_TABLE_PART_KEYS") ? strtoull(getenv("FG_TABLE_PART_KEYS"), nullptr, 10)
																	 : ((1ULL << FG_IDX_BITS) - 2));
	const u32 nParts = wide || nKeep == 0 ? 1u : (u32)((nKeep + perPart - 1) / perPart);
	if (nParts > FG_TABLE_MAX_PARTS) throw FgError{FG_ERR_UNSUPPORTED, "more than 16 * 2^30 distinct k-mers in the index"};
	T.nParts = nParts;
	std::vector<u64> partKeys(nParts, 0);
	for (u32 p = 0; p < nParts; ++p)
	{
		T.keyBase[p] = (u64)p * (wide ? 0 : perPart);
		partKeys[p] = std::min<u64>(nKeep - T.keyBase[p], wide ? nKeep : perPart);
		T.bound[p] = p == 0 ? 0ULL : fetch(c, c->dKeys.p + T.keyBase[p]);
	}
	T.bound[nParts] = ~0ULL;
	std::vector<u64> hRep(nRep);
	if (nRep) HIP_CHECK(hipMemcpyAsync(hRep.data(), c->dRepKeys.p, nRep * 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	for (u64 r : hRep)
	{
		u32 p = 0;
		while (p + 1 < nParts && r >= T.bound[p + 1]) ++p;
		++partKeys[p];
	}
	u64 slots = 0;
	for (u32 p = 0; p < nParts; ++p)
	{
		// in 8-slot groups; load 0.25 while that keeps the table small (fewer spilled groups: k_probe 5.3 ms
		// against 5.8 at load 0.5 on the bench workload), load 0.5 for big indexes (10 Gbp of reads: 1.4 G keys =
		// 23 GB instead of 46).  FG_TABLE_LOAD_PCT for experiments.
		const u64 loadPct = getenv("FG_TABLE_LOAD_PCT") ? std::max(5, std::min(90, atoi(getenv("FG_TABLE_LOAD_PCT"))))
													 : ((nKeep + nRep) * 32 <= (2ULL << 30) ? 25 : 50);
		const u64 g = std::max<u64>(16, (partKeys[p] * 100 / loadPct + 7) / 8);
		if (g > 0xFFFFFFFFULL) throw FgError{FG_ERR_UNSUPPORTED, "lookup table part too large"};
		T.slotBase[p] = slots;
		T.groups[p] = (u32)g;
		slots += g * 8;
	}
	c->tableSlots = slots;
	c->tableWide = wide;
	c->dTable.alloc(slots * (wide ? 2 : 1));
	HIP_CHECK(hipMemsetAsync(c->dTable.p, 0xFF, slots * (wide ? 16 : 8), s));
	T.slots = wide ? nullptr : (const unsigned long long*)c->dTable.p;
	T.wide = wide ? (const ulonglong2*)c->dTable.p : nullptr;
	c->table = T;
	if (nKeep + nRep)
	{
		ScopedK t(c->timer, "k_table_insert");
		if (wide)
			hipLaunchKernelGGL(k_table_insert<true>, gridFor(nKeep + nRep), WG, 0, s, c->dKeys.p, c->dKeyOff.p, nKeep,
							   c->dRepKeys.p, nRep, T, (u64*)nullptr, (ulonglong2*)c->dTable.p);
		else
			hipLaunchKernelGGL(k_table_insert<false>, gridFor(nKeep + nRep), WG, 0, s, c->dKeys.p, c->dKeyOff.p, nKeep,
							   c->dRepKeys.p, nRep, T, c->dTable.p, (ulonglong2*)nullptr);
	}
	c->dIndexedBits.alloc((c->totalKmers + 31) / 32 + 1);
	HIP_CHECK(hipMemsetAsync(c->dIndexedBits.p, 0, c->dIndexedBits.bytes(), s));
	if (c->nReads && c->totalKmers)
	{
		ScopedK t(c->timer, "k_indexed_bits");
		hipLaunchKernelGGL(k_indexed_bits, c->nReads, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, c->k,
						   flags, c->table, c->tableWide ? 1 : 0, c->dEntries.p, c->dIndexedBits.p);
	}
	HIP_CHECK(hipStreamSynchronize(s));
	c->indexBuilt = true;
}

void fgIndexBeginSolid(fg_ctx* c, i32 minFreq, float selectRate, i32 tandemFreq, float repeatRate,
					   float sampleRateInit, u64* histOut)
{
	if (c->k > 17) throw FgError{FG_ERR_KMER_SIZE, "Can't use flat counter for k-mer size > 17"};
	hipStream_t s = c->stream;
	const int k = c->k;
	const u32 n = c->nReads;
	clearIndex(c);
	c->timer.reset();
	c->indexBuild.reset();
	std::shared_ptr<IndexBuild> B(new IndexBuild);
	BuildClock clock(c);
	B->solid = true; B->minCoverage = minFreq; B->repeatRate = repeatRate; B->sampleRateInit = sampleRateInit;
	B->posBits = bitsFor((u64)c->maxLen);

	const u64 space = 1ULL << (2 * k);
	B->counts.alloc(space);
	DevBuf<unsigned long long> scal;	// [0] distinct, [1] tandem candidates, [2] accepted
	scal.alloc(4);
	HIP_CHECK(hipMemsetAsync(scal.p, 0, 32, s));
	{ ScopedK t(c->timer, "memset_counts"); HIP_CHECK(hipMemsetAsync(B->counts.p, 0, space * 4, s)); }
	DevBuf<u32> freq, thr;
	freq.alloc(c->totalKmers); B->flags.alloc(c->totalKmers); thr.alloc(n);
	if (n)
	{
		{ ScopedK t(c->timer, "k_count");
		  hipLaunchKernelGGL(k_count, n, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p, k, B->counts.p, scal.p); }
		{ ScopedK t(c->timer, "k_freq");
		  hipLaunchKernelGGL(k_freq, n, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, k, B->counts.p, freq.p); }
		{ ScopedK t(c->timer, "k_threshold");
		  hipLaunchKernelGGL(k_threshold, n, WG, 0, s, c->dLen.p, c->dKmerOff.p, k, freq.p, selectRate, thr.p); }
		{ ScopedK t(c->timer, "k_mark");
		  hipLaunchKernelGGL(k_mark, n, WG, 0, s, c->dLen.p, c->dKmerOff.p, k, freq.p, thr.p, tandemFreq, B->flags.p, scal.p + 1); }
	}
	unsigned long long h[4];
	HIP_CHECK(hipMemcpyAsync(h, scal.p, 32, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	B->totalDistinct = h[0];
	if (h[1] > 0)
	{
		u64 slots = 1024;
		while (slots < 2 * h[1]) slots <<= 1;
		DevBuf<u64> tkeys; DevBuf<u32> tcnt;
		tkeys.alloc(slots); tcnt.alloc(slots);
		HIP_CHECK(hipMemsetAsync(tkeys.p, 0xFF, slots * 8, s));
		HIP_CHECK(hipMemsetAsync(tcnt.p, 0, slots * 4, s));
		{ ScopedK t(c->timer, "k_tandem_insert");
		  hipLaunchKernelGGL(k_tandem_insert, n, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, k,
							 B->flags.p, tkeys.p, tcnt.p, slots - 1); }
		{ ScopedK t(c->timer, "k_tandem_apply");
		  hipLaunchKernelGGL(k_tandem_apply, n, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, k,
							 B->flags.p, tkeys.p, tcnt.p, slots - 1, tandemFreq); }
		HIP_CHECK(hipStreamSynchronize(s));
	}
	if (n)
	{
		ScopedK t(c->timer, "k_accept");
		hipLaunchKernelGGL(k_accept, n, WG, 0, s, c->dLen.p, c->dKmerOff.p, k, freq.p, minFreq, B->flags.p, scal.p + 2);
	}
	freq.release();
	binHistogram(c, B.get(), histOut);
	B->seconds = clock.stop();
	c->indexBuild = B;
}

void fgIndexBeginMinimizers(fg_ctx* c, i32 minCoverage, i32 window, float repeatRate, u64* histOut)
{
	if (window < 1 || window > MAXW) throw FgError{FG_ERR_ARG, "wrong minimizer length"};
	hipStream_t s = c->stream;
	const int k = c->k;
	const u32 n = c->nReads;
	clearIndex(c);
	c->timer.reset();
	c->indexBuild.reset();
	std::shared_ptr<IndexBuild> B(new IndexBuild);
	BuildClock clock(c);
	B->solid = false; B->minCoverage = minCoverage; B->repeatRate = repeatRate;
	B->posBits = bitsFor((u64)c->maxLen);
	DevBuf<unsigned long long> scal;
	scal.alloc(4);
	HIP_CHECK(hipMemsetAsync(scal.p, 0, 32, s));
	B->flags.alloc(c->totalKmers);
	{
		DevBuf<u64> hashes;
		hashes.alloc(window == 1 ? 1 : c->totalKmers);
		if (n)
		{
			ScopedK t(c->timer, "k_minimizers");
			hipLaunchKernelGGL(k_minimizers, n, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, k,
							   window, hashes.p, B->flags.p, scal.p + 2);
		}
		HIP_CHECK(hipStreamSynchronize(s));
	}
	binHistogram(c, B.get(), histOut);
	B->seconds = clock.stop();
	c->indexBuild = B;
}

void fgBuildIndexSolid(fg_ctx* c, i32 minFreq, float selectRate, i32 tandemFreq, float repeatRate,
					   float sampleRateInit, fg_index_stats* st)
{
	fgIndexBeginSolid(c, minFreq, selectRate, tandemFreq, repeatRate, sampleRateInit, nullptr);
	fgIndexBuildRange(c, 0, FG_INDEX_BINS, nullptr);
	fgIndexFinish(c, nullptr, st);
}

void fgBuildIndexMinimizers(fg_ctx* c, i32 minCoverage, i32 window, float repeatRate, fg_index_stats* st)
{
	fgIndexBeginMinimizers(c, minCoverage, window, repeatRate, nullptr);
	fgIndexBuildRange(c, 0, FG_INDEX_BINS, nullptr);
	fgIndexFinish(c, nullptr, st);
}

// An index given as CSR arrays (host or device memory): what a rank assembles from the all-gathered pieces
// of a sharded build, or a saved index.  keys ascending, key_off[nKeys + 1], entries (record << 32 | pos)
// ascending per key, as fg_export_index writes them.
void fgImportIndex(fg_ctx* c, u64 nKeys, const u64* keys, const u64* keyOff, u64 nEnt, const u64* entries, u64 nRep,
				   const u64* repKeys, float sampleRate, int onDevice)
{
	hipStream_t s = c->stream;
	clearIndex(c);
	c->timer.reset();
	c->indexBuild.reset();
	const hipMemcpyKind kind = onDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
	c->dKeys.alloc(nKeys); c->dKeyOff.alloc(nKeys + 1); c->dEntries.alloc(nEnt); c->dRepKeys.alloc(nRep);
	if (nKeys) HIP_CHECK(hipMemcpyAsync(c->dKeys.p, keys, nKeys * 8, kind, s));
	HIP_CHECK(hipMemcpyAsync(c->dKeyOff.p, keyOff, (nKeys + 1) * 8, kind, s));
	if (nEnt) HIP_CHECK(hipMemcpyAsync(c->dEntries.p, entries, nEnt * 8, kind, s));
	if (nRep) HIP_CHECK(hipMemcpyAsync(c->dRepKeys.p, repKeys, nRep * 8, kind, s));
	c->nKeys = nKeys; c->nEntries = nEnt; c->nRep = nRep;
	c->sampleRate = sampleRate;
	fgIndexLookupStructures(c, nullptr);
	c->timer.collect();
}
