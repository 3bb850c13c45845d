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
//   * counting: one u32 counter per possible k-mer of a key RANGE, direct addressed (the whole key space:
//     4^k * 4 B = 68.7 GB at k = 17, sized for 288 GB HBM; one rank of eight: an eighth of it), in chunks of
//     8 GiB (one allocation of 64 GiB takes seconds, eight of 8 GiB do not), one global atomic per k-mer, no
//     CAS loop, no overflow map, exact; freed after the build like _kmerCounter.clear();
//   * selection in BATCHES of reads: frequencies, per-read thresholds (one workgroup per read, LDS histogram
//     radix-select), tandem filter, minimizer deque run on a bounded batch scratch; what stays is ONE BIT per
//     k-mer position of the whole read set ("this position contributes an entry");
//   * index: the accepted (k-mer, position) pairs of a key range are emitted in ascending position order
//     (deterministic offsets, no cursor atomics), ordered by ONE stable LSD radix sort on the k-mer
//     (fg_devprim.h, hand-written onesweep) and cut into a CSR by a head-flag scan; lookups go through a
//     probing table of 8-byte slots in 64-byte groups, one line per probe.
// Every kernel and device primitive here is hand-written; no library primitive is used.
#include "fg_ctx.h"
#include "fg_devprim.h"

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

// The exact counters of the canonical k-mers in [keyLo, keyHi), in one of two forms:
//   direct  one u32 per possible k-mer of the range; chunk c holds keys keyLo + [c << CB, (c + 1) << CB)
//           (a read set that fills a good part of the 4^k key space: 10 Gbp of reads at k = 17);
//   hashed  when the reads hold far fewer k-mers than the range has keys (E. coli 50x: 215 M positions against
//           17 G keys): an open-addressing table of 8-byte slots (key << 30 | count), sized by the k-mer positions
//           that fall into the range -- 4 GB instead of 69, no multi-second allocation, counters that stay in
//           reach of the TLB.  Only chosen below 2^30 k-mer positions, so a count cannot reach the key bits; the
//           all-ones slot (key T^17, never canonical) marks an empty one.
#define FG_COUNT_CHUNK_BITS 31		// 2^31 u32 counters = 8 GiB per allocation
#define FG_COUNT_MAX_CHUNKS 8		// k = 17: 4^17 / 2^31
#define FG_COUNT_HBITS 30
struct CountView {
	u32* chunk[FG_COUNT_MAX_CHUNKS];
	unsigned long long* table;		// hashed form (else null)
	u64 mask;						// slots - 1
	u64 keyLo, keyHi;
};
__device__ __forceinline__ bool cv_has(const CountView& cv, u64 key) { return key >= cv.keyLo && key < cv.keyHi; }
// one more occurrence of `key`; true when it is the first
__device__ __forceinline__ bool cv_add(const CountView& cv, u64 key)
{
	if (!cv.table)
	{
		const u64 o = key - cv.keyLo;
		return atomicAdd(cv.chunk[o >> FG_COUNT_CHUNK_BITS] + (o & ((1ULL << FG_COUNT_CHUNK_BITS) - 1)), 1u) == 0u;
	}
	u64 h = fg_mix(key) & cv.mask;
	while (true)
	{
		// a slot only ever goes from empty to ONE key: a stale read of it errs towards "empty" and the CAS decides
		unsigned long long cur = cv.table[h];
		if (cur == FG_EMPTY_KEY)
		{
			cur = atomicCAS(&cv.table[h], (unsigned long long)FG_EMPTY_KEY, (unsigned long long)((key << FG_COUNT_HBITS) | 1ULL));
			if (cur == FG_EMPTY_KEY) return true;
		}
		if ((cur >> FG_COUNT_HBITS) == key) { atomicAdd(&cv.table[h], 1ULL); return false; }
		h = (h + 1) & cv.mask;
	}
}
// KmerCounter::getFreq of a key of the range (after the counting kernel)
__device__ __forceinline__ u32 cv_get(const CountView& cv, u64 key)
{
	if (!cv.table)
	{
		const u64 o = key - cv.keyLo;
		return cv.chunk[o >> FG_COUNT_CHUNK_BITS][o & ((1ULL << FG_COUNT_CHUNK_BITS) - 1)];
	}
	u64 h = fg_mix(key) & cv.mask;
	while (true)
	{
		const unsigned long long cur = cv.table[h];
		if (cur == FG_EMPTY_KEY) return 0u;
		if ((cur >> FG_COUNT_HBITS) == key) return (u32)(cur & ((1ULL << FG_COUNT_HBITS) - 1));
		h = (h + 1) & cv.mask;
	}
}

// vertex_index.cpp:520-558: one increment per canonical k-mer of every forward read -- here: of those whose
// canonical form lies in the counted key range (everything on one GPU, a rank's share on several)
__global__ void k_count(const u64* __restrict__ words, const u64* __restrict__ wordOff,
						const i32* __restrict__ len, int k, CountView cv,
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
		if (cv_has(cv, cn)) local += cv_add(cv, cn) ? 1u : 0u;
	}
	u32 t = block_sum(local, sh);
	if (threadIdx.x == 0 && t) atomicAdd(distinct, (unsigned long long)t);
}

// KmerCounter::getFreq for every k-mer position of the reads r0 .. r0 + gridDim.x - 1 (vertex_index.cpp:326-333);
// positions whose k-mer lies outside the counted range get 0 (their count is another rank's: the ranks' arrays
// add up to the complete one).  freq is the batch's array: position p of read r at kmerOff[r] - kmerOff[r0] + p
__global__ void k_freq(u32 r0, const u64* __restrict__ words, const u64* __restrict__ wordOff,
					   const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
					   CountView cv, u32* __restrict__ freq)
{
	const u32 r = r0 + blockIdx.x;
	const i32 nk = len[r] - k;
	const u64* w = words + wordOff[r];
	u32* f = freq + (kmerOff[r] - kmerOff[r0]);
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		u64 fw, rv;
		fg_kmer_pair(w, p, k, fw, rv);
		const u64 cn = fw < rv ? fw : rv;
		f[p] = cv_has(cv, cn) ? cv_get(cv, cn) : 0u;
	}
}

// vertex_index.cpp:336-344: threshold = frequency at rank (size_t)(selectRate * n)
// of the descending order; all k-mers with freq >= threshold are kept.
#define HBINS 2048
__global__ void k_threshold(u32 r0, const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
							const u32* __restrict__ freq, float selectRate, u32* __restrict__ thr)
{
	__shared__ u32 hist[HBINS];
	__shared__ u32 shThr;
	__shared__ u32 shCnt[WG / 64];
	const u32 r = r0 + blockIdx.x;
	const i32 nk = len[r] - k;
	if (nk <= 0) { if (threadIdx.x == 0) thr[blockIdx.x] = 0; return; }
	const u32* f = freq + (kmerOff[r] - kmerOff[r0]);
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
	if (shThr < HBINS - 1) { if (threadIdx.x == 0) thr[blockIdx.x] = shThr; return; }
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
	if (threadIdx.x == 0) thr[blockIdx.x] = lo;
}

// flags (one byte per position of the batch): bit0 selected (freq >= per-read threshold), bit1 tandem candidate
__global__ void k_mark(u32 r0, const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
					   const u32* __restrict__ freq, const u32* __restrict__ thr, i32 tandemFreq,
					   uint8_t* __restrict__ flags, unsigned long long* __restrict__ nCand)
{
	__shared__ u32 sh[WG / 64];
	const u32 r = r0 + blockIdx.x;
	const i32 nk = len[r] - k;
	if (nk <= 0) return;
	const u64 off = kmerOff[r] - kmerOff[r0];
	const u32* f = freq + off;
	uint8_t* fl = flags + off;
	const u32 t = thr[blockIdx.x];
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
__global__ void k_tandem_insert(u32 r0, const u64* __restrict__ words, const u64* __restrict__ wordOff,
								const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
								const uint8_t* __restrict__ flags, u64* __restrict__ tkeys,
								u32* __restrict__ tcnt, u64 tmask)
{
	const u32 r = r0 + blockIdx.x;
	const i32 nk = len[r] - k;
	const u64* w = words + wordOff[r];
	const uint8_t* fl = flags + (kmerOff[r] - kmerOff[r0]);
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		if (!(fl[p] & 2)) continue;
		u64 fw, rv;
		fg_kmer_pair(w, p, k, fw, rv);
		const u64 key = ((u64)blockIdx.x << 34) | (fw < rv ? fw : rv);
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

__global__ void k_tandem_apply(u32 r0, const u64* __restrict__ words, const u64* __restrict__ wordOff,
							   const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
							   uint8_t* __restrict__ flags, const u64* __restrict__ tkeys,
							   const u32* __restrict__ tcnt, u64 tmask, i32 tandemFreq)
{
	const u32 r = r0 + blockIdx.x;
	const i32 nk = len[r] - k;
	const u64* w = words + wordOff[r];
	uint8_t* fl = flags + (kmerOff[r] - kmerOff[r0]);
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		if (!(fl[p] & 2)) continue;
		u64 fw, rv;
		fg_kmer_pair(w, p, k, fw, rv);
		const u64 key = ((u64)blockIdx.x << 34) | (fw < rv ? fw : rv);
		u64 h = fg_mix(key) & tmask;
		while (tkeys[h] != key) h = (h + 1) & tmask;
		if (tcnt[h] > (u32)tandemFreq) fl[p] = 0;
	}
}

// bit0 := accepted for the index (selected, not tandem, freq >= minFreq)
__global__ void k_accept(u32 r0, const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
						 const u32* __restrict__ freq, i32 minFreq, uint8_t* __restrict__ flags,
						 unsigned long long* __restrict__ nAcc)
{
	__shared__ u32 sh[WG / 64];
	const u32 r = r0 + blockIdx.x;
	const i32 nk = len[r] - k;
	if (nk <= 0) return;
	const u64 off = kmerOff[r] - kmerOff[r0];
	const u32* f = freq + off;
	uint8_t* fl = flags + off;
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

// the batch's byte flags -> the read set's bit array: thread per 32-bit word touched by positions
// [posBase, posBase + nPos) (a word at a batch boundary is shared with the neighbouring batch: atomicOr)
__global__ void k_pack_bits(const uint8_t* __restrict__ flags, u64 posBase, u64 nPos, u32* __restrict__ bits)
{
	const u64 w = (posBase >> 5) + (u64)blockIdx.x * WG + threadIdx.x;
	if (w > ((posBase + nPos - 1) >> 5)) return;
	u32 v = 0;
#pragma unroll 8
	for (int j = 0; j < 32; ++j)
	{
		const u64 pos = w * 32 + j;
		if (pos >= posBase && pos < posBase + nPos && flags[pos - posBase]) v |= 1u << j;
	}
	if (v) atomicOr(&bits[w], v);
}

__device__ __forceinline__ bool fg_bit(const u32* __restrict__ bits, u64 i) { return (bits[i >> 5] >> (i & 31)) & 1u; }

// ---- minimizer sketch (kmer.h:206-262) ------------------------------------------
// The sketch equals "the sequence of distinct deque fronts".  Whenever hash[p] is
// strictly below the w previous hashes (or p == 0) the deque collapses to [p]
// whatever came before (a sync point), so the stretch up to the next sync point
// is an independent piece: sync points are found in parallel, then one lane runs
// the literal deque loop over one piece.
#define MAXW 64
__global__ void k_minimizers(u32 r0, const u64* __restrict__ words, const u64* __restrict__ wordOff,
							 const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k, int w,
							 u64* __restrict__ hashes /* scratch, one per k-mer position of the batch */,
							 uint8_t* __restrict__ flags /* of the batch */, unsigned long long* __restrict__ nAcc)
{
	__shared__ u32 sh[WG / 64];
	const u32 r = r0 + blockIdx.x;
	const i32 nk = len[r] - k;
	if (nk <= 0) return;
	const u64* wd = words + wordOff[r];
	u64* hs = hashes + (kmerOff[r] - kmerOff[r0]);
	uint8_t* fl = flags + (kmerOff[r] - kmerOff[r0]);
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
// flipped, position mirrored).  The pairs of a key range leave in ASCENDING VALUE order: read by read, a read's
// unflipped positions in ascending p (record 2r), then its flipped ones in DESCENDING p (record 2r + 1, position
// L - p - k ascending) -- so that one stable sort by k-mer leaves every list in ascending (record, position) order
// (vertex_index.cpp:108-114) and no sort by value is needed.  Pass 1 (k_emit_count) counts a read's pairs in the
// range, a scan over the reads gives its first slot, pass 2 (k_emit_write) places them.
__device__ __forceinline__ bool emit_take(const u64* __restrict__ w, const u32* __restrict__ bits, u64 base, i32 p, int k,
										  bool all, u64 keyLo, u64 keyHi, u64& canon, bool& flip)
{
	if (!fg_bit(bits, base + (u64)p)) return false;
	u64 fw, rv;
	fg_kmer_pair(w, p, k, fw, rv);
	flip = rv < fw;
	canon = flip ? rv : fw;
	return all || (canon >= keyLo && canon < keyHi);
}

__global__ void k_emit_count(const u64* __restrict__ words, const u64* __restrict__ wordOff,
							 const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
							 const u32* __restrict__ bits, u64 keyLo, u64 keyHi, u64* __restrict__ perRead)
{
	__shared__ u32 sh[WG / 64];
	const u32 r = blockIdx.x;
	const i32 nk = len[r] - k;
	const u64* w = words + wordOff[r];
	const u64 base = kmerOff[r];
	const bool all = keyLo == 0 && keyHi == ~0ULL;
	u32 mine = 0;
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		u64 cn; bool fl;
		mine += emit_take(w, bits, base, p, k, all, keyLo, keyHi, cn, fl) ? 1u : 0u;
	}
	const u32 t = block_sum(mine, sh);
	if (threadIdx.x == 0) perRead[r] = t;
}

__global__ void k_emit_write(const u64* __restrict__ words, const u64* __restrict__ wordOff,
							 const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
							 const u32* __restrict__ bits, int posBits, u64 keyLo, u64 keyHi,
							 const u64* __restrict__ readStart /* exclusive scan of perRead */,
							 u64* __restrict__ ecanon, u64* __restrict__ evalue)
{
	__shared__ u32 shU[WG / 64], shF[WG / 64];
	const u32 r = blockIdx.x;
	const i32 L = len[r];
	const i32 nk = L - k;
	const u64 first = readStart[r], cnt = readStart[r + 1] - first;
	if (cnt == 0) return;
	const u64* w = words + wordOff[r];
	const u64 base = kmerOff[r];
	const bool all = keyLo == 0 && keyHi == ~0ULL;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const u64 below = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
	// unflipped pairs of the read in the range: they come first
	u32 mineU = 0;
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		u64 cn; bool fl;
		if (emit_take(w, bits, base, p, k, all, keyLo, keyHi, cn, fl) && !fl) ++mineU;
	}
	for (int o = 32; o > 0; o >>= 1) mineU += __shfl_down(mineU, o);
	if (lane == 0) shU[wv] = mineU;
	__syncthreads();
	u64 totU = 0;
	for (int i = 0; i < WG / 64; ++i) totU += shU[i];
	__syncthreads();
	u64 doneU = 0, doneF = 0;	// pairs of each kind at positions below the current step
	for (i32 p0 = 0; p0 < nk; p0 += WG)
	{
		const i32 p = p0 + (i32)threadIdx.x;
		u64 cn = 0; bool fl = false;
		const bool take = p < nk && emit_take(w, bits, base, p, k, all, keyLo, keyHi, cn, fl);
		const u64 mU = __ballot(take && !fl), mF = __ballot(take && fl);
		if (lane == 0) { shU[wv] = (u32)__popcll(mU); shF[wv] = (u32)__popcll(mF); }
		__syncthreads();
		u32 beforeU = 0, beforeF = 0, stepU = 0, stepF = 0;
		for (int i = 0; i < WG / 64; ++i)
		{
			const u32 cu = shU[i], cf = shF[i];
			if (i < wv) { beforeU += cu; beforeF += cf; }
			stepU += cu; stepF += cf;
		}
		if (take)
		{
			u64 slot;
			if (!fl) slot = first + doneU + beforeU + (u64)__popcll(mU & below);
			else
			{
				// flipped pairs: the one at the highest p has the lowest mirrored position and goes first
				const u64 rankF = doneF + beforeF + (u64)__popcll(mF & below);		// rank in ascending p
				slot = first + totU + (cnt - totU - 1 - rankF);
			}
			ecanon[slot] = cn;
			evalue[slot] = ((u64)(2 * r + (fl ? 1 : 0)) << posBits) | (u64)(fl ? L - p - k : p);
		}
		doneU += stepU; doneF += stepF;
		__syncthreads();
	}
}

// positions per key bin (bin = canonical k-mer >> binShift) -- of the accepted positions (bits != null: what the
// slicing of the build, by memory on one GPU and by rank on several, balances on), or of ALL k-mer positions
// (bits == null: what the ranks' counter ranges are balanced on before anything is selected)
#define FG_INDEX_BINS 4096
__global__ void k_bin_hist(const u64* __restrict__ words, const u64* __restrict__ wordOff,
						   const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
						   const u32* __restrict__ bits, int binShift, unsigned long long* __restrict__ hist)
{
	__shared__ u32 sh[FG_INDEX_BINS];
	for (int i = threadIdx.x; i < FG_INDEX_BINS; i += WG) sh[i] = 0;
	__syncthreads();
	const u32 r = blockIdx.x;
	const i32 nk = len[r] - k;
	const u64* w = words + wordOff[r];
	const u64 base = kmerOff[r];
	for (i32 p = threadIdx.x; p < nk; p += WG)
		if (!bits || fg_bit(bits, base + (u64)p))
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
						   u64 repFreq, int solid /* 0 in minimizer mode */, CountView cv,
						   u32* __restrict__ isRep, u32* __restrict__ keep, u64* __restrict__ size,
						   u32* __restrict__ err)
{
	const u64 j = (u64)blockIdx.x * WG + threadIdx.x;
	if (j >= nKeys) return;
	const u64 cap = kstart[j + 1] - kstart[j];
	const bool rep = cap > repFreq;
	bool filled = !rep;
	if (filled && solid) filled = (u64)cv_get(cv, ukeys[j]) <= repFreq;
	isRep[j] = rep;
	keep[j] = !rep;
	size[j] = filled ? cap : 0;
	if (!rep && cap + 1 > (u64)(32 * 1024 * 1024 / 5)) *err = 1;
}

// totals of one part under the final repFreq: repetitive keys, kept keys, entries (same rules as k_classify)
__global__ void k_classify_count(const u64* __restrict__ ukeys, const u64* __restrict__ kstart, u64 nKeys, u64 repFreq,
								 int solid, CountView cv, unsigned long long* __restrict__ out /* rep, keep, entries */)
{
	__shared__ u64 sh[WG / 64];
	const u64 j = (u64)blockIdx.x * WG + threadIdx.x;
	u64 rep = 0, keep = 0, ent = 0;
	if (j < nKeys)
	{
		const u64 cap = kstart[j + 1] - kstart[j];
		const bool r = cap > repFreq;
		bool filled = !r;
		if (filled && solid) filled = (u64)cv_get(cv, ukeys[j]) <= repFreq;
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
// Form 1, after a build: the array holds the build's selection bits; a selected position owns an entry unless its
// k-mer left the index (repetitive, or a list left empty) -- those bits are cleared in place.
template <bool WIDE>
__global__ void k_indexed_clear(const u64* __restrict__ words, const u64* __restrict__ wordOff,
								const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
								FgTable T, u32* __restrict__ bits)
{
	const u32 r = blockIdx.x;
	const i32 nk = len[r] - k;
	const u64* w = words + wordOff[r];
	const u64 base = kmerOff[r];
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		if (!fg_bit(bits, base + (u64)p)) continue;
		u64 fw, rv;
		fg_kmer_pair(w, p, k, fw, rv);
		const u64 v = fg_probe<WIDE>(T, rv < fw ? rv : fw);
		if (v == 0 || (v & FG_CNT_MASK) == FG_CNT_REPETITIVE)
			atomicAnd(&bits[(base + p) >> 5], ~(1u << ((base + p) & 31)));
	}
}
// Form 2, an imported index (no selection at hand; bits start zeroed): a position owns an entry iff its own
// (record, position) is in the k-mer's list (ascending, vertex_index.cpp:108-114)
template <bool WIDE>
__global__ void k_indexed_search(const u64* __restrict__ words, const u64* __restrict__ wordOff,
								 const i32* __restrict__ len, const u64* __restrict__ kmerOff, int k,
								 FgTable T, const u64* __restrict__ entries, u32* __restrict__ bits)
{
	const u32 r = blockIdx.x;
	const i32 L = len[r];
	const i32 nk = L - k;
	const u64* w = words + wordOff[r];
	const u64 base = kmerOff[r];
	for (i32 p = threadIdx.x; p < nk; p += WG)
	{
		u64 fw, rv;
		fg_kmer_pair(w, p, k, fw, rv);
		const bool flip = rv < fw;
		const u64 v = fg_probe<WIDE>(T, flip ? rv : fw);
		if (v == 0 || (v & FG_CNT_MASK) == FG_CNT_REPETITIVE) continue;
		const u64 own = ((u64)(2 * r + (flip ? 1u : 0u)) << 32) | (u32)(flip ? L - p - k : p);
		const u64* e = entries + ((v >> FG_CNT_BITS) & ((1ULL << 38) - 1));
		u32 lo = 0, hi = (u32)(v & FG_CNT_MASK);
		while (lo < hi) { const u32 m = (lo + hi) >> 1; if (e[m] < own) lo = m + 1; else hi = m; }
		if (lo >= (u32)(v & FG_CNT_MASK) || e[lo] != own) continue;
		atomicOr(&bits[(base + p) >> 5], 1u << ((base + p) & 31));
	}
}

// an imported CSR must be well formed before anything reads lists through it: offsets start at 0, never
// decrease, end at nEntries; keys strictly ascending
__global__ void k_check_csr(const u64* __restrict__ keys, const u64* __restrict__ keyOff, u64 nKeys, u64 nEntries,
							u32* __restrict__ bad)
{
	const u64 j = (u64)blockIdx.x * WG + threadIdx.x;
	if (j > nKeys) return;
	bool ok = true;
	if (j == 0) ok = keyOff[0] == 0;
	if (j == nKeys) ok = ok && keyOff[nKeys] == nEntries;
	if (j < nKeys) ok = ok && keyOff[j] <= keyOff[j + 1] && keyOff[j + 1] <= nEntries;
	if (j + 1 < nKeys) ok = ok && keys[j] < keys[j + 1];
	if (!ok) atomicExch(bad, 1u);
}

// ---- host helpers -----------------------------------------------------------------
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


// ---- the build in steps ------------------------------------------------------------------------
//   selection    over ALL reads (solid: exact counts + per-read frequency threshold + tandem filter,
//                vertex_index.cpp:19-125; minimizers: kmer.h:206-262), in batches of reads on a bounded scratch
//                -> one BIT per k-mer position, and the number of accepted positions per key bin.
//                Solid mode in steps of its own (several GPUs): count slice -> per batch
//                {frequencies of the slice's k-mers -> summed over the ranks -> select} -> done;
//   build range  for the keys of bins [lo, hi): emit (canonical k-mer, position) pairs in position order, one
//                stable LSD radix sort by k-mer, run-length encode -> one PART (unique keys, list starts, sorted
//                positions) + its share of filterFrequentKmers' sums (vertex_index.cpp:175-184).  Ranges wider
//                than the memory budget are cut;
//   finish       with the sums over ALL keys: repetitive frequency, classification, CSR arrays in key
//                order, probe table, indexed bits.
// One GPU runs the selection, the whole key space, finish.  Several GPUs each run the selection (counters of
// their key range only), the range their rank owns, exchange the two sums, finish their piece and gather the
// pieces into full-size arrays (fg_index_gather_begin / _end) -- SURVEY.md §8(e).
struct IndexPart {
	DevBuf<u64> ukeys, kstart, evalue;
	DevBuf<u32> inc;
	u64 nKeys = 0, E = 0;
};

struct IndexBuild {
	bool solid = false;
	// solid mode: exact counters of the canonical k-mers of the key bins [cntBinLo, cntBinHi)
	DevBuf<u32> countChunk[FG_COUNT_MAX_CHUNKS];
	DevBuf<unsigned long long> countTable;
	CountView cv{};
	i32 minFreq = 0, tandemFreq = 0;
	float selectRate = 0;
	// selection: one bit per k-mer position of the read set; scratch of the batch in work
	DevBuf<u32> bits;
	std::vector<u32> batchStart;	// first read of each batch; back() = number of reads
	u64 batchCap = 0;				// k-mer positions of the largest batch
	u32 batchReads = 0;				// reads of the largest batch
	DevBuf<u32> freq, thr;
	DevBuf<uint8_t> flags;
	DevBuf<u64> hashes;
	DevBuf<unsigned long long> scal;	// [0] distinct, [1] tandem candidates (per batch), [2] accepted
	bool selectionDone = false;
	int posBits = 0, binShift = 0;
	i32 minCoverage = 0;
	float repeatRate = 0, sampleRateInit = 1.0f;
	u64 totalDistinct = 0;
	std::vector<u64> hist;		// accepted positions per bin
	std::vector<std::unique_ptr<IndexPart>> parts;	// ascending key ranges
	unsigned long long sums[2] = {0, 0};
	DevBuf<char> scratch;		// radix sort status words / scan tile sums
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
	c->gathering = false;
	c->gKeys.release(); c->gKeyOff.release(); c->gEntries.release(); c->gRepKeys.release();
	c->dKeys.release(); c->dKeyOff.release(); c->dEntries.release(); c->dRepKeys.release();
	c->dTable.release(); c->dIndexedBits.release();
	c->nKeys = c->nEntries = c->nRep = c->tableSlots = 0;
}

int binShiftFor(int k) { return std::max(0, 2 * k - 12); }

// per key bin: accepted positions (bits) or all k-mer positions (bits == null)
void binHistogram(fg_ctx* c, const u32* bits, int binShift, std::vector<u64>& hist)
{
	hipStream_t s = c->stream;
	DevBuf<unsigned long long> dh;
	dh.alloc(FG_INDEX_BINS);
	HIP_CHECK(hipMemsetAsync(dh.p, 0, FG_INDEX_BINS * 8, s));
	if (c->nReads)
	{
		ScopedK t(c->timer, "k_bin_hist");
		hipLaunchKernelGGL(k_bin_hist, c->nReads, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, c->k,
						   bits, binShift, dh.p);
	}
	hist.assign(FG_INDEX_BINS, 0);
	HIP_CHECK(hipMemcpyAsync(hist.data(), dh.p, FG_INDEX_BINS * 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
}

// what every selection starts with: a cleared index, the bit array, the batches
std::shared_ptr<IndexBuild> beginCommon(fg_ctx* c)
{
	hipStream_t s = c->stream;
	clearIndex(c);
	c->timer.reset();
	c->indexBuild.reset();
	std::shared_ptr<IndexBuild> B(new IndexBuild);
	B->posBits = bitsFor((u64)c->maxLen);
	B->binShift = binShiftFor(c->k);
	B->bits.alloc((c->totalKmers + 31) / 32 + 1);
	HIP_CHECK(hipMemsetAsync(B->bits.p, 0, B->bits.bytes(), s));
	B->scal.alloc(4);
	HIP_CHECK(hipMemsetAsync(B->scal.p, 0, 32, s));
	// batches of whole reads, at most `budget` k-mer positions each (a longer read is a batch of its own)
	const u64 budget = getenv("FG_INDEX_BATCH_KMERS") ? std::max<u64>(1, strtoull(getenv("FG_INDEX_BATCH_KMERS"), nullptr, 10))
													   : (256ULL << 20);
	B->batchStart.assign(1, 0);
	u64 acc = 0;
	u32 first = 0;
	for (u32 r = 0; r < c->nReads; ++r)
	{
		const u64 nk = c->hKmerOff[r + 1] - c->hKmerOff[r];
		if (r > first && acc + nk > budget)
		{
			B->batchCap = std::max(B->batchCap, acc); B->batchReads = std::max(B->batchReads, r - first);
			B->batchStart.push_back(r); first = r; acc = 0;
		}
		acc += nk;
	}
	B->batchCap = std::max(B->batchCap, acc); B->batchReads = std::max(B->batchReads, c->nReads - first);
	if (c->nReads) B->batchStart.push_back(c->nReads);
	return B;
}

void packBatch(fg_ctx* c, IndexBuild* B, u32 r0, u32 r1)
{
	const u64 posBase = c->hKmerOff[r0], nPos = c->hKmerOff[r1] - posBase;
	if (!nPos) return;
	const u64 words = ((posBase + nPos - 1) >> 5) - (posBase >> 5) + 1;
	ScopedK t(c->timer, "k_pack_bits");
	hipLaunchKernelGGL(k_pack_bits, gridFor(words), WG, 0, c->stream, B->flags.p, posBase, nPos, B->bits.p);
}

// one slice of keys -> one part
void buildPart(fg_ctx* c, IndexBuild* B, u32 binLo, u32 binHi, u64 E)
{
	if (E == 0) return;
	hipStream_t s = c->stream;
	const int k = c->k;
	const u32 n = c->nReads;
	const bool all = binLo == 0 && binHi >= FG_INDEX_BINS;
	const u64 keyLo = all ? 0ULL : ((u64)binLo << B->binShift);
	const u64 keyHi = (all || binHi >= FG_INDEX_BINS) ? ~0ULL : ((u64)binHi << B->binShift);
	if (E > fgprim::RS_MAX_N) throw FgError{FG_ERR_ARG, "index slice above 2^30 entries (FG_INDEX_SLICE_ENTRIES)"};
	std::unique_ptr<IndexPart> part(new IndexPart);
	// where each read's pairs go: count, scan over the reads
	DevBuf<u64> readStart;
	readStart.alloc((u64)n + 1);
	HIP_CHECK(hipMemsetAsync(readStart.p + n, 0, 8, s));
	{
		ScopedK t(c->timer, "k_emit");
		hipLaunchKernelGGL(k_emit_count, n, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, k, B->bits.p,
						   keyLo, keyHi, readStart.p);
	}
	B->scratch.reserve(std::max<size_t>(fgprim::radixSortScratchBytes(E),
										fgprim::scanScratchElems(std::max<u64>(E, (u64)n + 1)) * 8));
	{
		ScopedK t(c->timer, "scan");
		fgprim::scan<u64>(s, readStart.p, readStart.p, (u64)n + 1, false, (u64*)B->scratch.p);
	}
	if (fetch(c, readStart.p + n) != E) throw FgError{FG_ERR_HIP, "internal: slice emission does not match the bin histogram"};
	DevBuf<u64> ecanon, ecanon2, evalue2;
	ecanon.alloc(E); part->evalue.alloc(E);
	{
		ScopedK t(c->timer, "k_emit");
		hipLaunchKernelGGL(k_emit_write, n, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, k, B->bits.p,
						   B->posBits, keyLo, keyHi, readStart.p, ecanon.p, part->evalue.p);
	}
	readStart.release();
	ecanon2.alloc(E); evalue2.alloc(E);
	{
		// the pairs lie in ascending (record, position) order: ONE stable sort by k-mer
		ScopedK t(c->timer, "radix_sort_pairs");
		if (fgprim::radixSortPairs(s, ecanon.p, part->evalue.p, ecanon2.p, evalue2.p, E, 0, 2 * k, B->scratch.p))
		{
			ecanon.swap(ecanon2);
			part->evalue.swap(evalue2);
		}
	}
	ecanon2.release(); evalue2.release();
	// run-length encode the sorted k-mers
	DevBuf<u32> flag;
	flag.alloc(E); part->inc.alloc(E);
	{ ScopedK t(c->timer, "k_heads"); hipLaunchKernelGGL(k_heads, gridFor(E), WG, 0, s, ecanon.p, E, flag.p); }
	{ ScopedK t(c->timer, "scan"); fgprim::scan<u32>(s, flag.p, part->inc.p, E, true, (u32*)B->scratch.p); }
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

// ---- selection, solid mode ----------------------------------------------------------------------------
void fgIndexKmerHist(fg_ctx* c, u64* histOut)
{
	c->timer.reset();
	std::vector<u64> h;
	binHistogram(c, nullptr, binShiftFor(c->k), h);
	memcpy(histOut, h.data(), FG_INDEX_BINS * 8);
	c->timer.collect();
}

void fgIndexCountSlice(fg_ctx* c, i32 minFreq, float selectRate, i32 tandemFreq, float repeatRate, float sampleRateInit,
					   u32 binLo, u32 binHi, u64* distinctOut, u32* nBatchesOut)
{
	if (c->k > 17) throw FgError{FG_ERR_KMER_SIZE, "Can't use flat counter for k-mer size > 17"};
	if (binLo > binHi || binHi > FG_INDEX_BINS) throw FgError{FG_ERR_ARG, "bin range outside [0, 4096]"};
	hipStream_t s = c->stream;
	const int k = c->k;
	const u32 n = c->nReads;
	std::shared_ptr<IndexBuild> B = beginCommon(c);
	BuildClock clock(c);
	B->solid = true; B->minCoverage = minFreq; B->minFreq = minFreq; B->repeatRate = repeatRate;
	B->sampleRateInit = sampleRateInit; B->selectRate = selectRate; B->tandemFreq = tandemFreq;
	const u64 space = 1ULL << (2 * k);
	B->cv.keyLo = std::min(space, (u64)binLo << B->binShift);
	B->cv.keyHi = binHi >= FG_INDEX_BINS ? space : std::min(space, (u64)binHi << B->binShift);
	const u64 nCnt = B->cv.keyHi - B->cv.keyLo;
	// k-mer positions whose canonical form falls into the range: what a hashed counter is sized by
	std::vector<u64> khist;
	binHistogram(c, nullptr, B->binShift, khist);
	u64 nInRange = 0;
	for (u32 b = binLo; b < binHi; ++b) nInRange += khist[b];
	u64 slots = 1024;
	while (slots < 2 * nInRange) slots <<= 1;
	const char* mode = getenv("FG_COUNT_MODE");		// "direct" / "hash": experiments and tests
	bool hashed = c->totalKmers < (1ULL << FG_COUNT_HBITS) && slots * 8 <= nCnt * 4 / 4;
	if (mode && !strcmp(mode, "direct")) hashed = false;
	if (mode && !strcmp(mode, "hash") && c->totalKmers < (1ULL << FG_COUNT_HBITS)) hashed = true;
	if (hashed)
	{
		ScopedK t(c->timer, "memset_counts");
		B->countTable.alloc(slots);
		B->cv.table = B->countTable.p;
		B->cv.mask = slots - 1;
		HIP_CHECK(hipMemsetAsync(B->countTable.p, 0xFF, slots * 8, s));
	}
	else
	{
		const u64 chunk = 1ULL << FG_COUNT_CHUNK_BITS;
		const u64 nChunks = (nCnt + chunk - 1) / chunk;
		if (nChunks > FG_COUNT_MAX_CHUNKS) throw FgError{FG_ERR_KMER_SIZE, "counter range too wide"};
		ScopedK t(c->timer, "memset_counts");
		for (u64 i = 0; i < nChunks; ++i)
		{
			const u64 cnt = std::min(chunk, nCnt - i * chunk);
			B->countChunk[i].alloc(cnt);
			B->cv.chunk[i] = B->countChunk[i].p;
			HIP_CHECK(hipMemsetAsync(B->countChunk[i].p, 0, cnt * 4, s));
		}
	}
	if (n)
	{
		ScopedK t(c->timer, "k_count");
		hipLaunchKernelGGL(k_count, n, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p, k, B->cv, B->scal.p);
	}
	B->totalDistinct = fetch(c, (const unsigned long long*)B->scal.p);
	B->freq.alloc(B->batchCap); B->flags.alloc(B->batchCap); B->thr.alloc(std::max<u32>(1, B->batchReads));
	if (distinctOut) *distinctOut = B->totalDistinct;
	if (nBatchesOut) *nBatchesOut = (u32)(B->batchStart.size() - 1);
	B->seconds = clock.stop();
	c->indexBuild = B;
}

static void batchRange(fg_ctx* c, IndexBuild* B, u32 batch, u32& r0, u32& r1)
{
	if (!B->solid || B->selectionDone) throw FgError{FG_ERR_STATE, "no solid-mode selection in progress"};
	if ((size_t)batch + 1 >= B->batchStart.size()) throw FgError{FG_ERR_ARG, "batch index out of range"};
	r0 = B->batchStart[batch]; r1 = B->batchStart[batch + 1];
	(void)c;
}

// frequencies of the batch's k-mer positions as far as THIS context counted them (0 for k-mers outside its key
// range): device pointer + element count, for the caller to sum over the ranks in place
void fgIndexBatchFreq(fg_ctx* c, u32 batch, u32** dFreq, u64* nPos)
{
	IndexBuild* B = buildState(c);
	u32 r0, r1;
	batchRange(c, B, batch, r0, r1);
	BuildClock clock(c);
	{
		ScopedK t(c->timer, "k_freq");
		hipLaunchKernelGGL(k_freq, r1 - r0, WG, 0, c->stream, r0, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, c->k,
						   B->cv, B->freq.p);
	}
	if (dFreq) *dFreq = B->freq.p;
	if (nPos) *nPos = c->hKmerOff[r1] - c->hKmerOff[r0];
	B->seconds += clock.stop();
}

// yieldFrequentKmers over the batch's reads from the (complete) frequencies in the batch scratch -> selection bits
void fgIndexBatchSelect(fg_ctx* c, u32 batch)
{
	IndexBuild* B = buildState(c);
	u32 r0, r1;
	batchRange(c, B, batch, r0, r1);
	hipStream_t s = c->stream;
	const int k = c->k;
	const u32 nb = r1 - r0;
	BuildClock clock(c);
	HIP_CHECK(hipMemsetAsync(B->scal.p + 1, 0, 8, s));
	{ ScopedK t(c->timer, "k_threshold");
	  hipLaunchKernelGGL(k_threshold, nb, WG, 0, s, r0, c->dLen.p, c->dKmerOff.p, k, B->freq.p, B->selectRate, B->thr.p); }
	{ ScopedK t(c->timer, "k_mark");
	  hipLaunchKernelGGL(k_mark, nb, WG, 0, s, r0, c->dLen.p, c->dKmerOff.p, k, B->freq.p, B->thr.p, B->tandemFreq, B->flags.p,
						 B->scal.p + 1); }
	const u64 nCand = fetch(c, (const unsigned long long*)B->scal.p + 1);
	if (nCand > 0)
	{
		u64 slots = 1024;
		while (slots < 2 * nCand) slots <<= 1;
		DevBuf<u64> tkeys; DevBuf<u32> tcnt;
		tkeys.alloc(slots); tcnt.alloc(slots);
		HIP_CHECK(hipMemsetAsync(tkeys.p, 0xFF, slots * 8, s));
		HIP_CHECK(hipMemsetAsync(tcnt.p, 0, slots * 4, s));
		{ ScopedK t(c->timer, "k_tandem_insert");
		  hipLaunchKernelGGL(k_tandem_insert, nb, WG, 0, s, r0, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, k,
							 B->flags.p, tkeys.p, tcnt.p, slots - 1); }
		{ ScopedK t(c->timer, "k_tandem_apply");
		  hipLaunchKernelGGL(k_tandem_apply, nb, WG, 0, s, r0, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, k,
							 B->flags.p, tkeys.p, tcnt.p, slots - 1, B->tandemFreq); }
		HIP_CHECK(hipStreamSynchronize(s));
	}
	{ ScopedK t(c->timer, "k_accept");
	  hipLaunchKernelGGL(k_accept, nb, WG, 0, s, r0, c->dLen.p, c->dKmerOff.p, k, B->freq.p, B->minFreq, B->flags.p, B->scal.p + 2); }
	packBatch(c, B, r0, r1);
	B->seconds += clock.stop();
}

void fgIndexSelectionDone(fg_ctx* c, u64* histOut)
{
	IndexBuild* B = buildState(c);
	BuildClock clock(c);
	B->freq.release(); B->flags.release(); B->thr.release(); B->hashes.release();
	binHistogram(c, B->bits.p, B->binShift, B->hist);
	B->selectionDone = true;
	if (histOut) memcpy(histOut, B->hist.data(), FG_INDEX_BINS * 8);
	B->seconds += clock.stop();
}

void fgIndexBeginSolid(fg_ctx* c, i32 minFreq, float selectRate, i32 tandemFreq, float repeatRate,
					   float sampleRateInit, u64* histOut)
{
	u32 nBatches = 0;
	fgIndexCountSlice(c, minFreq, selectRate, tandemFreq, repeatRate, sampleRateInit, 0, FG_INDEX_BINS, nullptr, &nBatches);
	for (u32 b = 0; b < nBatches; ++b)
	{
		fgIndexBatchFreq(c, b, nullptr, nullptr);
		fgIndexBatchSelect(c, b);
	}
	fgIndexSelectionDone(c, histOut);
}

// ---- selection, minimizer mode ------------------------------------------------------------------------
void fgIndexBeginMinimizers(fg_ctx* c, i32 minCoverage, i32 window, float repeatRate, u64* histOut)
{
	if (window < 1 || window > MAXW) throw FgError{FG_ERR_ARG, "wrong minimizer length"};
	hipStream_t s = c->stream;
	std::shared_ptr<IndexBuild> B = beginCommon(c);
	BuildClock clock(c);
	B->solid = false; B->minCoverage = minCoverage; B->repeatRate = repeatRate;
	B->flags.alloc(B->batchCap);
	B->hashes.alloc(window == 1 ? 1 : B->batchCap);
	for (size_t b = 0; b + 1 < B->batchStart.size(); ++b)
	{
		const u32 r0 = B->batchStart[b], r1 = B->batchStart[b + 1];
		{
			ScopedK t(c->timer, "k_minimizers");
			hipLaunchKernelGGL(k_minimizers, r1 - r0, WG, 0, s, r0, c->dWords.p, c->dWordOff.p, c->dLen.p, c->dKmerOff.p, c->k,
							   window, B->hashes.p, B->flags.p, B->scal.p + 2);
		}
		packBatch(c, B.get(), r0, r1);
	}
	HIP_CHECK(hipStreamSynchronize(s));
	B->flags.release(); B->hashes.release();
	binHistogram(c, B->bits.p, B->binShift, B->hist);
	B->selectionDone = true;
	if (histOut) memcpy(histOut, B->hist.data(), FG_INDEX_BINS * 8);
	B->seconds = clock.stop();
	c->indexBuild = B;
}

// ---- build range / finish --------------------------------------------------------------------------------
void fgIndexBuildRange(fg_ctx* c, u32 binLo, u32 binHi, unsigned long long* sumsOut)
{
	IndexBuild* B = buildState(c);
	if (!B->selectionDone) throw FgError{FG_ERR_STATE, "the selection is not finished (fg_index_selection_done)"};
	if (binLo > binHi || binHi > FG_INDEX_BINS) throw FgError{FG_ERR_ARG, "bin range outside [0, 4096]"};
	if (B->solid && binLo < binHi)
	{
		// the finish step asks the counters about the keys of this range (vertex_index.cpp:70-71)
		const u64 lo = (u64)binLo << B->binShift, hi = binHi >= FG_INDEX_BINS ? (1ULL << (2 * c->k)) : ((u64)binHi << B->binShift);
		if (lo < B->cv.keyLo || std::min<u64>(hi, 1ULL << (2 * c->k)) > B->cv.keyHi)
			throw FgError{FG_ERR_ARG, "bin range outside the range this context counted"};
	}
	BuildClock clock(c);
	// slices of at most `budget` entries (sort buffers = 4 x 8 bytes per entry of the slice)
	const u64 budget = std::min<u64>(fgprim::RS_MAX_N, getenv("FG_INDEX_SLICE_ENTRIES") ? strtoull(getenv("FG_INDEX_SLICE_ENTRIES"), nullptr, 10)
																						  : (768ULL << 20));
	u32 lo = binLo;
	while (lo < binHi)
	{
		u32 hi = lo;
		u64 e = 0;
		while (hi < binHi && (hi == lo || e + B->hist[hi] <= budget)) { e += B->hist[hi]; ++hi; }
		buildPart(c, B, lo, hi, e);
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
						   B->solid ? 1 : 0, B->cv, tot.p);
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
		B->scratch.reserve(fgprim::scanScratchElems(nKeys + 1) * 8);
		// one extra zero element so that the exclusive scans also yield the totals
		HIP_CHECK(hipMemsetAsync(isRep.p + nKeys, 0, 4, s));
		HIP_CHECK(hipMemsetAsync(keep.p + nKeys, 0, 4, s));
		HIP_CHECK(hipMemsetAsync(size.p + nKeys, 0, 8, s));
		{
			ScopedK t(c->timer, "k_classify");
			hipLaunchKernelGGL(k_classify, gridFor(nKeys), WG, 0, s, p->ukeys.p, p->kstart.p, nKeys, (u64)repFreq,
							   B->solid ? 1 : 0, B->cv, isRep.p, keep.p, size.p, err.p);
		}
		{
			ScopedK t(c->timer, "scan");
			fgprim::scan<u32>(s, isRep.p, repIdx.p, nKeys + 1, false, (u32*)B->scratch.p);
			fgprim::scan<u32>(s, keep.p, keepIdx.p, nKeys + 1, false, (u32*)B->scratch.p);
			fgprim::scan<u64>(s, size.p, off.p, nKeys + 1, false, (u64*)B->scratch.p);
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
	for (auto& ch : B->countChunk) ch.release();
	B->countTable.release();
	B->scratch.release();
	c->nKeys = nKeep; c->nEntries = nEnt; c->nRep = nRep;
	// the selection bits become the "owns an entry" bits, in place
	c->dIndexedBits.swap(B->bits);
	fgIndexLookupStructures(c, true);
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
// the context.  bitsHoldSelection: c->dIndexedBits holds the build's selection bits (cleared in place where a
// selected position's k-mer left the index); otherwise (imported index) the lists are searched
void fgIndexLookupStructures(fg_ctx* c, bool bitsHoldSelection)
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
	const u64 bitWords = (c->totalKmers + 31) / 32 + 1;
	if (!bitsHoldSelection || c->dIndexedBits.n < bitWords)
	{
		bitsHoldSelection = false;
		c->dIndexedBits.alloc(bitWords);
		HIP_CHECK(hipMemsetAsync(c->dIndexedBits.p, 0, c->dIndexedBits.bytes(), s));
	}
	if (c->nReads && c->totalKmers)
	{
		ScopedK t(c->timer, "k_indexed_bits");
		if (bitsHoldSelection)
		{
			if (wide) hipLaunchKernelGGL(k_indexed_clear<true>, c->nReads, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p,
										 c->dKmerOff.p, c->k, c->table, c->dIndexedBits.p);
			else hipLaunchKernelGGL(k_indexed_clear<false>, c->nReads, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p,
									c->dKmerOff.p, c->k, c->table, c->dIndexedBits.p);
		}
		else
		{
			if (wide) hipLaunchKernelGGL(k_indexed_search<true>, c->nReads, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p,
										 c->dKmerOff.p, c->k, c->table, c->dEntries.p, c->dIndexedBits.p);
			else hipLaunchKernelGGL(k_indexed_search<false>, c->nReads, WG, 0, s, c->dWords.p, c->dWordOff.p, c->dLen.p,
									c->dKmerOff.p, c->k, c->table, c->dEntries.p, c->dIndexedBits.p);
		}
	}
	HIP_CHECK(hipStreamSynchronize(s));
	c->indexBuilt = true;
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

// the CSR arrays in the context must be well formed before lists are read through them
static void checkCsr(fg_ctx* c)
{
	DevBuf<u32> bad;
	bad.alloc(1);
	HIP_CHECK(hipMemsetAsync(bad.p, 0, 4, c->stream));
	{
		ScopedK t(c->timer, "k_check_csr");
		hipLaunchKernelGGL(k_check_csr, gridFor(c->nKeys + 1), WG, 0, c->stream, c->dKeys.p, c->dKeyOff.p, c->nKeys, c->nEntries, bad.p);
	}
	if (fetch(c, bad.p))
	{
		clearIndex(c);
		throw FgError{FG_ERR_ARG, "malformed index arrays: key_off must start at 0, never decrease and end at n_entries; keys must ascend"};
	}
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
	checkCsr(c);
	fgIndexLookupStructures(c, false);
	c->timer.collect();
}

// The all-gather of a sharded build without a second copy of the index: this context's own piece (what finish left)
// is set aside, the context's arrays become the full-size ones, and the caller's collective writes every rank's
// piece -- its own included -- straight into them (full[0..3] = keys, key_off, entries, repetitive keys;
// piece[0..3] the same of the own piece, pieceSizes = {keys, entries, repetitive}).  fgIndexGatherEnd frees the
// piece, checks the arrays and builds the lookup structures.
void fgIndexGatherBegin(fg_ctx* c, u64 nKeys, u64 nEntries, u64 nRep, u64** full, u64** piece, u64* pieceSizes)
{
	if (!c->indexBuilt || c->gathering) throw FgError{FG_ERR_STATE, "gather needs a finished piece (fg_index_finish)"};
	c->timer.reset();
	c->indexBuilt = false;
	c->dTable.release(); c->dIndexedBits.release();
	c->gKeys.swap(c->dKeys); c->gKeyOff.swap(c->dKeyOff); c->gEntries.swap(c->dEntries); c->gRepKeys.swap(c->dRepKeys);
	c->gNKeys = c->nKeys; c->gNEntries = c->nEntries; c->gNRep = c->nRep;
	c->dKeys.alloc(nKeys); c->dKeyOff.alloc(nKeys + 1); c->dEntries.alloc(nEntries); c->dRepKeys.alloc(nRep);
	c->nKeys = nKeys; c->nEntries = nEntries; c->nRep = nRep;
	c->gathering = true;
	full[0] = c->dKeys.p; full[1] = c->dKeyOff.p; full[2] = c->dEntries.p; full[3] = c->dRepKeys.p;
	piece[0] = c->gKeys.p; piece[1] = c->gKeyOff.p; piece[2] = c->gEntries.p; piece[3] = c->gRepKeys.p;
	pieceSizes[0] = c->gNKeys; pieceSizes[1] = c->gNEntries; pieceSizes[2] = c->gNRep;
}

void fgIndexGatherEnd(fg_ctx* c, float sampleRate)
{
	if (!c->gathering) throw FgError{FG_ERR_STATE, "no gather in progress"};
	c->gathering = false;
	c->gKeys.release(); c->gKeyOff.release(); c->gEntries.release(); c->gRepKeys.release();
	c->sampleRate = sampleRate;
	checkCsr(c);
	fgIndexLookupStructures(c, false);
	c->timer.collect();
}
