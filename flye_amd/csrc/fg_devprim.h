// Device-wide primitives of the index build, hand-written for gfx950 (64-lane waves):
//   * exclusive / inclusive prefix sums (reduce per tile -> scan of the tile sums, recursively -> scan per tile);
//   * a STABLE least-significant-digit radix sort of (u64 key, u64 value) pairs on 8-bit digits, "onesweep" form:
//     one pass over the keys counts every digit of every pass; each pass then reads and writes every pair once --
//     a tile ranks its pairs by wave-level ballots (equal digits keep their input order), learns where its digits
//     start from the tiles before it by decoupled look-back over per-(tile, digit) status words, reorders the
//     tile through LDS and writes runs of equal digits.  Passes in which all keys share one digit are skipped.
// What the build sorts: the (canonical k-mer, position) pairs emitted for a range of key bins
// (reference src/sequence/vertex_index.cpp:41-114: cuckoo-map upserts + a std::sort of every position list; here
// the pairs are emitted in ascending position order and ONE stable sort by k-mer leaves every list sorted).
#pragma once
#include "fg_ctx.h"

namespace fgprim {

// ---- prefix sums ------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

#if defined(__HIPCC__)

// exclusive scan of one value per thread over a 256-thread block; *total = block sum (to every thread)
template <class T>
__device__ __forceinline__ T block_exscan_256(T v, T* sh /* >= 4 */, T* total)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	T inc = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1)
	{
		const T t = __shfl_up(inc, o);
		if (lane >= o) inc += t;
	}
	__syncthreads();
	if (lane == 63) sh[w] = inc;
	__syncthreads();
	T base = 0, tot = 0;
#pragma unroll
	for (int i = 0; i < SCAN_THREADS / 64; ++i) { const T s = sh[i]; if (i < w) base += s; tot += s; }
	*total = tot;
	return base + inc - v;
}

template <class T>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_reduce(const T* __restrict__ in, u64 n, T* __restrict__ tileSum)
{
	__shared__ T sh[SCAN_THREADS / 64];
	const u64 base = (u64)blockIdx.x * SCAN_TILE;
	T v = 0;
#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; ++i)
	{
		const u64 j = base + (u64)i * SCAN_THREADS + threadIdx.x;
		if (j < n) v += in[j];
	}
	T tot;
	(void)block_exscan_256(v, sh, &tot);
	if (threadIdx.x == 0) tileSum[blockIdx.x] = tot;
}

// one tile: thread t owns the SCAN_ITEMS consecutive elements behind base + t * SCAN_ITEMS (read before anything
// is written: in == out is allowed); tileOff = exclusive sums of the tiles (null: one tile)
template <class T, bool INCLUSIVE>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_tile(const T* in, T* out, u64 n, const T* __restrict__ tileOff)
{
	__shared__ T sh[SCAN_THREADS / 64];
	const u64 first = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
	T x[SCAN_ITEMS];
	T mine = 0;
#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; ++i) { x[i] = first + i < n ? in[first + i] : T(0); mine += x[i]; }
	T tot;
	T run = block_exscan_256(mine, sh, &tot) + (tileOff ? tileOff[blockIdx.x] : T(0));
#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; ++i)
	{
		if (INCLUSIVE) run += x[i];
		if (first + i < n) out[first + i] = run;
		if (!INCLUSIVE) run += x[i];
	}
}

#endif // __HIPCC__

// scratch of a scan of n elements, in elements of T
inline u64 scanScratchElems(u64 n)
{
	u64 tot = 0;
	while (n > (u64)SCAN_TILE) { n = (n + SCAN_TILE - 1) / SCAN_TILE; tot += n; }
	return tot + 1;
}

#if defined(__HIPCC__)
// out[i] = sum of in[0 .. i) (or in[0 .. i] when inclusive); scratch: scanScratchElems(n) elements
template <class T>
void scan(hipStream_t s, const T* in, T* out, u64 n, bool inclusive, T* scratch)
{
	if (n == 0) return;
	const u64 nTiles = (n + SCAN_TILE - 1) / SCAN_TILE;
	if (nTiles > 0x7fffffffULL) throw FgError{FG_ERR_ARG, "scan: too many elements"};
	const T* tileOff = nullptr;
	if (nTiles > 1)
	{
		hipLaunchKernelGGL(k_scan_reduce<T>, (unsigned)nTiles, SCAN_THREADS, 0, s, in, n, scratch);
		scan<T>(s, scratch, scratch, nTiles, false, scratch + nTiles);
		tileOff = scratch;
	}
	if (inclusive) hipLaunchKernelGGL((k_scan_tile<T, true>), (unsigned)nTiles, SCAN_THREADS, 0, s, in, out, n, tileOff);
	else hipLaunchKernelGGL((k_scan_tile<T, false>), (unsigned)nTiles, SCAN_THREADS, 0, s, in, out, n, tileOff);
}
#endif

// ---- stable LSD radix sort of (u64, u64) pairs ----------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;
constexpr int RS_RADIX = 256;
constexpr int RS_MAX_PASSES = 8;
constexpr u32 RS_FLAG_AGG = 1u << 30, RS_FLAG_PREFIX = 2u << 30, RS_VAL_MASK = (1u << 30) - 1u;
constexpr u64 RS_MAX_N = (1ULL << 30) - 1;		// a status word carries a 30-bit running count

#if defined(__HIPCC__)

namespace {		// kernels of a header shared by several translation units: one internal copy each

// digit counts of every pass in one sweep over the keys: hist[pass * 256 + digit]
__global__ void __launch_bounds__(RS_THREADS) k_rs_hist(const u64* __restrict__ keys, u64 n, int beginBit, int nPasses,
														 unsigned long long* __restrict__ hist)
{
	__shared__ u32 sh[RS_MAX_PASSES * RS_RADIX];
	for (int i = threadIdx.x; i < nPasses * RS_RADIX; i += RS_THREADS) sh[i] = 0;
	__syncthreads();
	const u64 per = ((n + gridDim.x - 1) / gridDim.x + RS_THREADS - 1) / RS_THREADS * RS_THREADS;
	const u64 a = (u64)blockIdx.x * per, b = a + per < n ? a + per : n;
	for (u64 i = a + threadIdx.x; i < b; i += RS_THREADS)
	{
		const u64 k = keys[i] >> beginBit;
		for (int p = 0; p < nPasses; ++p) atomicAdd(&sh[p * RS_RADIX + (int)((k >> (8 * p)) & 255)], 1u);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < nPasses * RS_RADIX; i += RS_THREADS)
		if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

// One pass.  Tiles take their number from a ticket counter, so every tile a tile waits for in the look-back has
// started before it and never waits for a later one: the waits end.  A spin bound turns a protocol bug into an
// error flag instead of a hung GPU.
__global__ void __launch_bounds__(RS_THREADS)
k_rs_onesweep(const u64* __restrict__ keysIn, const u64* __restrict__ valsIn, u64* __restrict__ keysOut,
			  u64* __restrict__ valsOut, u64 n, int shift, const u64* __restrict__ digitStart /* 256: exclusive counts */,
			  u32* status /* tiles * 256 */, u32* ticket, u32* err)
{
	__shared__ u64 sKey[RS_TILE];
	__shared__ u64 sVal[RS_TILE];
	__shared__ u32 waveHist[RS_WAVES][RS_RADIX];
	__shared__ u32 sTileStart[RS_RADIX];	// first slot of each digit inside the reordered tile
	__shared__ long long sOffs[RS_RADIX];	// destination of slot j of digit d = sOffs[d] + j
	__shared__ u32 shScan[RS_THREADS / 64];
	__shared__ u32 sTile;
	if (threadIdx.x == 0) sTile = atomicAdd(ticket, 1u);
#pragma unroll
	for (int w = 0; w < RS_WAVES; ++w) waveHist[w][threadIdx.x] = 0;
	__syncthreads();
	const u32 tile = sTile;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const u64 base = (u64)tile * RS_TILE + (u64)wv * (64 * RS_ITEMS);
	const u64 below = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));

	// the wave's 512 pairs, item i of lane l = element i * 64 + l of the wave's stretch (input order = (item, lane))
	u64 k[RS_ITEMS], v[RS_ITEMS];
	u32 rk[RS_ITEMS];
#pragma unroll
	for (int i = 0; i < RS_ITEMS; ++i)
	{
		const u64 j = base + (u64)i * 64 + lane;
		k[i] = j < n ? keysIn[j] : ~0ULL;
		v[i] = j < n ? valsIn[j] : 0ULL;
	}
#pragma unroll
	for (int i = 0; i < RS_ITEMS; ++i)
	{
		const bool valid = base + (u64)i * 64 + lane < n;
		const u32 d = (u32)(k[i] >> shift) & 255u;
		// lanes of this item holding the same digit
		u64 peers = __ballot(valid);
#pragma unroll
		for (int b = 0; b < 8; ++b)
		{
			const bool bit = (d >> b) & 1u;
			const u64 m = __ballot(bit);
			peers &= bit ? m : ~m;
		}
		const int leader = valid ? (int)__ffsll((unsigned long long)peers) - 1 : lane;
		u32 pre = 0;
		if (valid && lane == leader)
		{
			pre = waveHist[wv][d];
			waveHist[wv][d] = pre + (u32)__popcll(peers);
		}
		pre = __shfl(pre, leader);
		rk[i] = pre + (u32)__popcll(peers & below);
	}
	__syncthreads();

	// thread d: digit d of this tile
	const int d = threadIdx.x;
	u32 cnt = 0;
#pragma unroll
	for (int w = 0; w < RS_WAVES; ++w) { const u32 c = waveHist[w][d]; waveHist[w][d] = cnt; cnt += c; }
	// publish, then look back
	u32* const st = status + (u64)tile * RS_RADIX + d;
	__hip_atomic_store(st, (tile == 0 ? RS_FLAG_PREFIX : RS_FLAG_AGG) | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	u32 excl = 0;
	if (tile > 0)
	{
		long long t = (long long)tile - 1;
		u32 spins = 0;
		while (true)
		{
			const u32 s = __hip_atomic_load(status + (u64)t * RS_RADIX + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const u32 flag = s >> 30;
			if (flag == 0)
			{
				if (++spins > (1u << 26)) { atomicExch(err, 1u); break; }
				__builtin_amdgcn_s_sleep(1);
				continue;
			}
			excl += s & RS_VAL_MASK;
			if (flag == 2) break;
			--t;
		}
		__hip_atomic_store(st, RS_FLAG_PREFIX | ((excl + cnt) & RS_VAL_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	u32 tot;
	const u32 tileStart = block_exscan_256<u32>(cnt, shScan, &tot);
	sTileStart[d] = tileStart;
	sOffs[d] = (long long)(digitStart[d] + excl) - (long long)tileStart;
	__syncthreads();

	// reorder through LDS: slot = digit start in the tile + pairs of the waves below + rank inside the wave
#pragma unroll
	for (int i = 0; i < RS_ITEMS; ++i)
	{
		const bool valid = base + (u64)i * 64 + lane < n;
		const u32 dg = (u32)(k[i] >> shift) & 255u;
		if (valid)
		{
			const u32 slot = sTileStart[dg] + waveHist[wv][dg] + rk[i];
			sKey[slot] = k[i];
			sVal[slot] = v[i];
		}
	}
	__syncthreads();
#pragma unroll
	for (int i = 0; i < RS_ITEMS; ++i)
	{
		const u32 j = (u32)i * RS_THREADS + threadIdx.x;
		if (j < tot)
		{
			const u64 kk = sKey[j];
			const long long dst = sOffs[(u32)(kk >> shift) & 255u] + (long long)j;
			keysOut[dst] = kk;
			valsOut[dst] = sVal[j];
		}
	}
}

} // namespace

#endif // __HIPCC__

// scratch bytes of a sort of n pairs (status words + histograms + digit starts + ticket + error flag)
inline size_t radixSortScratchBytes(u64 n)
{
	const u64 nTiles = (n + RS_TILE - 1) / RS_TILE;
	return (size_t)(nTiles * RS_RADIX * 4 + RS_MAX_PASSES * RS_RADIX * 8 * 2 + 256);
}

#if defined(__HIPCC__)
// Sorts n (key, value) pairs by bits [beginBit, endBit) of the key, stable.  (k0, v0) hold the input, (k1, v1) are
// buffers of the same size; returns 0 when the result is in (k0, v0), 1 when it is in (k1, v1).
inline int radixSortPairs(hipStream_t s, u64* k0, u64* v0, u64* k1, u64* v1, u64 n, int beginBit, int endBit, char* scratch)
{
	if (n <= 1 || endBit <= beginBit) return 0;
	if (n > RS_MAX_N) throw FgError{FG_ERR_ARG, "radix sort: more than 2^30 - 1 pairs in one call"};
	const int nPasses = (endBit - beginBit + 7) / 8;
	if (nPasses > RS_MAX_PASSES) throw FgError{FG_ERR_ARG, "radix sort: more than 64 key bits"};
	const u64 nTiles = (n + RS_TILE - 1) / RS_TILE;
	u32* status = (u32*)scratch;
	unsigned long long* hist = (unsigned long long*)(scratch + nTiles * RS_RADIX * 4);
	u64* digitStart = (u64*)(hist + RS_MAX_PASSES * RS_RADIX);
	u32* ticket = (u32*)(digitStart + RS_MAX_PASSES * RS_RADIX);
	u32* err = ticket + 1;
	HIP_CHECK(hipMemsetAsync(hist, 0, RS_MAX_PASSES * RS_RADIX * 8, s));
	HIP_CHECK(hipMemsetAsync(ticket, 0, 8, s));
	const unsigned histBlocks = (unsigned)std::min<u64>(2048, (n + RS_THREADS * 16 - 1) / (RS_THREADS * 16));
	hipLaunchKernelGGL(k_rs_hist, histBlocks, RS_THREADS, 0, s, k0, n, beginBit, nPasses, hist);
	std::vector<unsigned long long> h((size_t)nPasses * RS_RADIX);
	HIP_CHECK(hipMemcpyAsync(h.data(), hist, h.size() * 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	std::vector<u64> starts((size_t)nPasses * RS_RADIX);
	std::vector<char> skip(nPasses, 0);
	for (int p = 0; p < nPasses; ++p)
	{
		u64 run = 0;
		for (int d = 0; d < RS_RADIX; ++d)
		{
			starts[(size_t)p * RS_RADIX + d] = run;
			if (h[(size_t)p * RS_RADIX + d] == n) skip[p] = 1;		// every key has this digit: the pass moves nothing
			run += h[(size_t)p * RS_RADIX + d];
		}
	}
	HIP_CHECK(hipMemcpyAsync(digitStart, starts.data(), starts.size() * 8, hipMemcpyHostToDevice, s));
	int cur = 0;
	for (int p = 0; p < nPasses; ++p)
	{
		if (skip[p]) continue;
		HIP_CHECK(hipMemsetAsync(status, 0, nTiles * RS_RADIX * 4, s));
		HIP_CHECK(hipMemsetAsync(ticket, 0, 4, s));
		hipLaunchKernelGGL(k_rs_onesweep, (unsigned)nTiles, RS_THREADS, 0, s, cur ? k1 : k0, cur ? v1 : v0, cur ? k0 : k1,
						   cur ? v0 : v1, n, beginBit + 8 * p, digitStart + (size_t)p * RS_RADIX, status, ticket, err);
		cur ^= 1;
	}
	u32 herr = 0;
	HIP_CHECK(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));		// also keeps `starts` alive until its copy has run
	if (herr) throw FgError{FG_ERR_HIP, "internal: radix sort look-back did not finish"};
	return cur;
}
#endif

} // namespace fgprim
