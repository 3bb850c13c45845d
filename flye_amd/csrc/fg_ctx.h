// Internal context of libflyegpu.so (host side) + device helpers shared by the
// kernels.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <memory>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <atomic>
#include <algorithm>

#include "../../include/flye_gpu.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef int32_t i32;

#define FG_EMPTY_KEY 0xFFFFFFFFFFFFFFFFULL
#define FG_CNT_BITS 24
#define FG_CNT_MASK 0xFFFFFFu
#define FG_CNT_REPETITIVE 0xFFFFFFu	// probe result for a k-mer of _repetitiveKmers
#define FG_IDX_BITS 30				// narrow slot: key << 30 | key index inside its part (all ones = repetitive)
#define FG_IDX_MASK 0x3FFFFFFFULL
#define FG_TABLE_MAX_PARTS 16

// The lookup table as the kernels see it (passed by value).  Narrow form (2k <= 34 bits, i.e. k <= 17): one
// u64 per slot; keys are split into parts by key RANGE so that an index inside a part fits 30 bits
// (one part up to 2^30 - 2 keys; the split follows the sorted key array, equal key counts per part).
// Wide form (k > 17): {key, key index} pairs, one part.  Linear probing inside a part, load <= 0.5.
struct FgTable {
	const unsigned long long* slots;
	const ulonglong2* wide;
	const unsigned long long* keyOff;
	unsigned nParts;
	unsigned long long bound[FG_TABLE_MAX_PARTS + 1];	// part p: keys in [bound[p], bound[p + 1])
	unsigned long long slotBase[FG_TABLE_MAX_PARTS];
	unsigned long long keyBase[FG_TABLE_MAX_PARTS];
	unsigned groups[FG_TABLE_MAX_PARTS];				// 8-slot groups of part p
};

struct FgError { int code; std::string msg; };

#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	throw FgError{FG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + \
	" (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"}; } } while (0)

// --- device buffer -----------------------------------------------------------
// device bytes this library holds right now / at most since the last reset, over all contexts of the process
// (fg_memory_stats): every device allocation of the library goes through DevBuf
inline std::atomic<unsigned long long> g_fgDevBytes{0}, g_fgDevPeak{0};
inline void fgDevAccount(long long delta)
{
	const unsigned long long now = g_fgDevBytes.fetch_add((unsigned long long)delta) + (unsigned long long)delta;
	unsigned long long pk = g_fgDevPeak.load();
	while (delta > 0 && now > pk && !g_fgDevPeak.compare_exchange_weak(pk, now)) {}
}

template <class T>
struct DevBuf {
	T* p = nullptr;
	size_t n = 0;
	bool owned = true;		// false: a view of another context's buffer (the second lane of fg_overlaps)
	DevBuf() {}
	DevBuf(const DevBuf&) = delete;
	DevBuf& operator=(const DevBuf&) = delete;
	~DevBuf() { release(); }
	void release()
	{
		if (p && owned) { (void)hipFree(p); fgDevAccount(-(long long)(n * sizeof(T))); }
		p = nullptr; n = 0; owned = true;
	}
	void alias(const DevBuf& o) { release(); p = o.p; n = o.n; owned = false; }
	void swap(DevBuf& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(owned, o.owned); }
	void alloc(size_t count)
	{
		release();
		if (count == 0) count = 1;
		hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
		if (e != hipSuccess)
			throw FgError{FG_ERR_NOMEM, "hipMalloc of " + std::to_string(count * sizeof(T)) + " bytes: " + hipGetErrorString(e)};
		n = count;
		fgDevAccount((long long)(n * sizeof(T)));
	}
	// grow-only (keeps capacity between batches)
	void reserve(size_t count) { if (count > n) alloc(count + count / 8); }
	size_t bytes() const { return n * sizeof(T); }
};

// page-locked host staging buffer (grow-only)
template <class T>
struct PinnedBuf {
	T* p = nullptr;
	size_t n = 0;
	PinnedBuf() {}
	PinnedBuf(const PinnedBuf&) = delete;
	PinnedBuf& operator=(const PinnedBuf&) = delete;
	~PinnedBuf() { if (p) (void)hipHostFree(p); }
	void reserve(size_t count)
	{
		if (count <= n) return;
		if (p) { (void)hipHostFree(p); p = nullptr; n = 0; }
		count += count / 8;
		hipError_t e = hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault);
		if (e != hipSuccess)
			throw FgError{FG_ERR_NOMEM, "hipHostMalloc of " + std::to_string(count * sizeof(T)) + " bytes: " + hipGetErrorString(e)};
		n = count;
	}
	// grow and keep the first `used` elements (no copy is pending on them: the caller has synchronised)
	void reserveKeep(size_t count, size_t used)
	{
		if (count <= n) return;
		count += count / 2;
		T* q = nullptr;
		hipError_t e = hipHostMalloc((void**)&q, count * sizeof(T), hipHostMallocDefault);
		if (e != hipSuccess)
			throw FgError{FG_ERR_NOMEM, "hipHostMalloc of " + std::to_string(count * sizeof(T)) + " bytes: " + hipGetErrorString(e)};
		if (p && used) memcpy(q, p, used * sizeof(T));
		if (p) (void)hipHostFree(p);
		p = q; n = count;
	}
};

// --- per-kernel timing with HIP events on the library stream -------------------
struct KernelTimer {
	struct Ev { const char* name; hipEvent_t a, b; };
	std::vector<Ev> evs;
	std::vector<hipEvent_t> pool;
	hipStream_t stream = nullptr;
	bool enabled = true;
	std::vector<fg_kernel_time> last;

	hipEvent_t get()
	{
		if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
		hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); return e;
	}
	// on: the stream the bracketed launches go to (default: the context's main stream)
	size_t begin(const char* name, hipStream_t on = nullptr)
	{
		if (!enabled) return 0;
		Ev ev{name, get(), get()};
		HIP_CHECK(hipEventRecord(ev.a, on ? on : stream));
		evs.push_back(ev);
		return evs.size() - 1;
	}
	void end(size_t id, hipStream_t on = nullptr)
	{
		if (!enabled) return;
		HIP_CHECK(hipEventRecord(evs[id].b, on ? on : stream));
	}
	// forget uncollected measurements (a previous call failed half way): the events go back to the pool
	void reset()
	{
		for (auto& ev : evs) { pool.push_back(ev.a); pool.push_back(ev.b); }
		evs.clear();
	}
	// after a stream sync: fold into per-name sums
	void collect()
	{
		last.clear();
		std::vector<std::string> order;
		std::map<std::string, size_t> idx;
		for (auto& ev : evs)
		{
			float ms = 0;
			HIP_CHECK(hipEventElapsedTime(&ms, ev.a, ev.b));
			auto it = idx.find(ev.name);
			if (it == idx.end())
			{
				idx[ev.name] = last.size();
				last.push_back(fg_kernel_time{ev.name, 0.0, 0});
				it = idx.find(ev.name);
			}
			last[it->second].seconds += ms * 1e-3;
			last[it->second].launches += 1;
			pool.push_back(ev.a); pool.push_back(ev.b);
		}
		evs.clear();
	}
	~KernelTimer() { for (auto e : pool) (void)hipEventDestroy(e); }
};

struct ScopedK {
	KernelTimer& t; size_t id; hipStream_t on;
	ScopedK(KernelTimer& t_, const char* name, hipStream_t on_ = nullptr) : t(t_), id(t_.begin(name, on_)), on(on_) {}
	~ScopedK() { try { t.end(id, on); } catch (...) {} }
};

// Worker threads of the host shim, kept between calls: spawning 2 x 16 std::threads per fg_overlaps call and
// faulting in fresh result-sized vectors cost more than the shim's own arithmetic.
// CPUs this process may actually use: the cgroup's quota where there is one (a container that shows 256 threads but
// is granted 16 CPUs of time runs host loops slower on 32 threads than on 16)
inline unsigned fg_usable_cpus()
{
	static const unsigned cpus = []
	{
		unsigned n = std::max(1u, std::thread::hardware_concurrency());
		if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r"))
		{
			char quota[32]; long period = 0;
			if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
				n = std::min<unsigned>(n, (unsigned)std::max(1L, atol(quota) / period));
			fclose(f);
		}
		// one process per GPU (torchrun): the ranks of this node share those CPUs
		if (const char* lw = getenv("LOCAL_WORLD_SIZE"))
			if (atoi(lw) > 1) n = std::max(1u, n / (unsigned)atoi(lw));
		return n;
	}();
	return cpus;
}

struct ShimPool {
	std::vector<std::thread> threads;
	std::mutex mu;
	std::condition_variable cvGo, cvDone;
	const std::function<void(unsigned)>* job = nullptr;
	unsigned long long epoch = 0;
	unsigned pending = 0, nActive = 0;
	bool stop = false;

	void ensure(unsigned n)
	{
		while (threads.size() < n)
		{
			const unsigned id = (unsigned)threads.size();
			threads.emplace_back([this, id]
			{
				unsigned long long seen = 0;
				std::unique_lock<std::mutex> lk(mu);
				while (true)
				{
					cvGo.wait(lk, [&] { return stop || (epoch != seen && id < nActive); });
					if (stop) return;
					seen = epoch;
					const std::function<void(unsigned)>* f = job;
					lk.unlock();
					(*f)(id);
					lk.lock();
					if (--pending == 0) cvDone.notify_all();
				}
			});
		}
	}
	// fn(0 .. n-1), one call per thread; returns when all are done.  n == 1 runs inline.
	void run(unsigned n, const std::function<void(unsigned)>& fn)
	{
		if (n <= 1) { fn(0); return; }
		ensure(n);
		std::unique_lock<std::mutex> lk(mu);
		job = &fn; nActive = n; pending = n; ++epoch;
		cvGo.notify_all();
		cvDone.wait(lk, [&] { return pending == 0; });
		// threads beyond n that woke up went back to sleep (id >= nActive); the epoch they skipped is harmless
	}
	~ShimPool()
	{
		{ std::lock_guard<std::mutex> g(mu); stop = true; }
		cvGo.notify_all();
		for (auto& t : threads) if (t.joinable()) t.join();
	}
};

struct fg_ctx {
	int device = 0;
	int k = 17;
	hipStream_t stream = nullptr;
	// side stream of the chaining stage: the big-group kernels run there beside the small-group ones of the main
	// stream (fork / join by events), so that neither class waits for the other's last waves
	hipStream_t stream2 = nullptr, stream3 = nullptr;
	hipEvent_t evFork = nullptr, evJoin = nullptr, evJoin3 = nullptr;
	// the primaries of a chunk leave for the host in FG_D2H_PIECES copies; the shim's threads start on a piece as
	// soon as ITS copy has landed (piece i covers the primaries below pieceEnd[i], counted over the whole call)
#define FG_D2H_PIECES 4
	hipEvent_t evOff = nullptr, evPiece[FG_D2H_PIECES] = {};
	unsigned long long pieceEnd[FG_D2H_PIECES] = {};
	std::string lastError;
	KernelTimer timer;

	// reads (the index container == the query container for now)
	u32 nReads = 0;
	u32 firstId = 0;
	u64 totalWords = 0, totalBases = 0, totalKmers = 0;
	i32 maxLen = 0;
	std::vector<i32> hLen;
	std::vector<u64> hKmerOff;
	DevBuf<u64> dWords;		// +2 padding words
	DevBuf<u64> dWordOff;	// n+1
	DevBuf<i32> dLen;		// n
	DevBuf<u64> dKmerOff;	// n+1 prefix of max(len-k,0)

	// optional second container holding the queries (ReadAligner-style use: reads vs graph
	// edges, reference src/repeat_graph/read_aligner.cpp:178-217); when absent queries are
	// the indexed reads themselves
	bool hasQ = false;
	u32 nQReads = 0, qFirstId = 0;
	i32 qMaxLen = 0;
	std::vector<i32> hQLen;
	DevBuf<u64> dQWords, dQWordOff;
	DevBuf<i32> dQLen;

	// index
	bool indexBuilt = false;
	float sampleRate = 1.0f;
	u64 nKeys = 0, nEntries = 0, nRep = 0, tableSlots = 0;
	DevBuf<u64> dKeys;		// ascending canonical k-mers of _kmerIndex (incl. empty lists)
	DevBuf<u64> dKeyOff;	// nKeys+1
	DevBuf<u64> dEntries;	// (record<<32 | pos), ascending per key
	DevBuf<u64> dRepKeys;	// ascending
	DevBuf<u64> dTable;			// narrow slots, or ulonglong2 {key, index} pairs when tableWide
	bool tableWide = false;
	FgTable table{};
	DevBuf<u32> dIndexedBits;	// one bit per forward k-mer position: contributes an entry

	// second lane of fg_overlaps: a context of its own scratch, streams and timers whose read / index / probe buffers
	// are views of this one's (fg_overlap.hip: sub-ranges of a call's queries run on two lanes side by side)
	std::unique_ptr<fg_ctx> lane2;
	std::shared_ptr<void> indexBuild;	// state between the steps of an index build (fg_index.hip)
	// between fg_index_gather_begin and _end: this context's own piece of a sharded build, set aside while the
	// arrays above are the full-size ones the ranks' pieces are gathered into
	bool gathering = false;
	DevBuf<u64> gKeys, gKeyOff, gEntries, gRepKeys;
	u64 gNKeys = 0, gNEntries = 0, gNRep = 0;

	// overlap-stage scratch (grow-only)
	DevBuf<u32> dQuery;			// query record indices
	u64 hitCapHint = 0;			// hits of the largest sub-range of the chunk in work (per-hit scratch is reserved for it)
	const u32* curQuery = nullptr;	// ... of the sub-range of the chunk the stage is working on (group / primary records index it)
	DevBuf<u64> dQKmerOff;		// per query prefix of k-mer counts
	DevBuf<u64> dProbe;			// per query k-mer: table value (0 = miss)
	// probes partitioned by table region (tables beyond the caches): (region | k-mer, position) pairs, twice, + sort scratch
	DevBuf<u64> dPartK0, dPartV0, dPartK1, dPartV1;
	DevBuf<char> dPartScratch;
	DevBuf<u64> dHitOff;		// per query hit offsets (nq+1)
	DevBuf<u64> dFiltOff;		// per query repetitive-position offsets (nq+1)
	DevBuf<i32> dFiltPos;
	DevBuf<u64> dHitKey;		// extId<<32 | curPos
	DevBuf<u32> dHitVal;		// extPos
	DevBuf<u32> dHitKey32;		// record << curBits | curPos, when that fits 32 bits (sort only)
	DevBuf<i32> dScore, dBack;
	DevBuf<int4> dCand;
	DevBuf<u64> dGroupStart;	// group boundaries (indices into hits)
	DevBuf<u32> dGroupQuery;
	// per group: its target id and the query position of its first and last hit (what the
	// chaining kernels and the span prefilter need of the sorted keys)
	DevBuf<u32> dGroupExt, dGroupFirstCur, dGroupLastCur;
	DevBuf<uint8_t> dGroupExtSorted;	// written by k_group_prep: the DP runs in extPos order (overlap.cpp:268-275)
	DevBuf<u32> dTmp32;
	DevBuf<u64> dCntA, dCntB, dGroupCnt, dGroupOff, dPrimCnt, dPrimOff, dDpGroups, dDpElems;
	DevBuf<u32> dPrimFlag, dDpSize, dListSmall, dListBig, dListDp, dListFused, dListCnt;
	DevBuf<u32> dCur, dExt;		// (cur, ext) columns of the groups in DP order
	DevBuf<char> dPrimOut;	// PrimRec array
	// keep_alignment: per primary the chain's last hit (global hit index), its group's first
	// hit, the slot offsets (chainLength + 2 pairs each) and the thinned chain itself
	DevBuf<u64> dPrimNode, dPrimBase, dMatchSize, dMatchOff, dMatches;
	DevBuf<u32> dMatchCnt;
	PinnedBuf<u64> hMatches, hMatchOff;
	PinnedBuf<u32> hMatchCnt;
	DevBuf<char> dSortTasks, dSortBig;
	DevBuf<unsigned long long> dSmallElems;	// DP elements in groups of <= 256 hits (work counter)
	DevBuf<u32> dSortCnt;		// task counts of the sort's level loop, a row per level (fg_overlap.hip)
	DevBuf<int> dEditScratch;
	DevBuf<u32> dEditList;		// pairs queued for the bit-vector kernel (two lists)
	DevBuf<char> dEditCnt;
	DevBuf<u64> dEditSlab;		// per-block string planes + delta planes of the bit-vector kernel
	PinnedBuf<char> hPrim;
	PinnedBuf<u64> hOff;
	PinnedBuf<u64> hScalar;		// staging of the counts the host reads between kernels (pinned: no bounce buffer)
	// host shim: worker threads and result-sized scratch kept between calls
	ShimPool shimPool;
	std::vector<float> shimDiv;
	std::vector<uint8_t> shimKeep;
	std::vector<u32> shimNStat;
	std::vector<u64> shimNMatch;

	~fg_ctx()
	{
		if (evFork) (void)hipEventDestroy(evFork);
		if (evJoin) (void)hipEventDestroy(evJoin);
		if (evJoin3) (void)hipEventDestroy(evJoin3);
		if (evOff) (void)hipEventDestroy(evOff);
		for (auto e : evPiece) if (e) (void)hipEventDestroy(e);
		if (stream2) (void)hipStreamDestroy(stream2);
		if (stream3) (void)hipStreamDestroy(stream3);
		if (stream) (void)hipStreamDestroy(stream);
	}
};

// --- device helpers ----------------------------------------------------------
#if defined(__HIPCC__)

// wave-uniform values loaded through the vector memory path: pinned to SGPRs so that loop
// control becomes scalar branches and base addresses scalar operands
__device__ __forceinline__ i32 fg_uni(i32 v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u32 fg_uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 fg_uni(u64 v)
{
	return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32)) << 32) |
		   (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
}

// One 64-bit sort RECORD for workloads whose (record, query position) pairs do not fit 32 bits:
// (record << (curBits + 24)) | (curPos << 24) | extPos.  Ordered (and tested for >=, <=) on the upper
// bits only -- the (extId, curPos) comparator of overlap.cpp:201-204; the target position rides in the
// low 24 bits, so a hit stays 8 bytes in the sort levels instead of 8 + 4.  == / != look at all bits.
#define FG_PK_VALBITS 24
struct PK {
	u64 v;
	PK() = default;
	__host__ __device__ PK(int x) : v((u64)(unsigned)x) {}
	__host__ __device__ explicit PK(u64 x) : v(x) {}
};
__device__ __forceinline__ bool operator<(const PK& a, const PK& b) { return (a.v >> FG_PK_VALBITS) < (b.v >> FG_PK_VALBITS); }
__device__ __forceinline__ bool operator>(const PK& a, const PK& b) { return (a.v >> FG_PK_VALBITS) > (b.v >> FG_PK_VALBITS); }
__device__ __forceinline__ bool operator<=(const PK& a, const PK& b) { return (a.v >> FG_PK_VALBITS) <= (b.v >> FG_PK_VALBITS); }
__device__ __forceinline__ bool operator>=(const PK& a, const PK& b) { return (a.v >> FG_PK_VALBITS) >= (b.v >> FG_PK_VALBITS); }
__device__ __forceinline__ bool operator==(const PK& a, const PK& b) { return a.v == b.v; }
__device__ __forceinline__ bool operator!=(const PK& a, const PK& b) { return a.v != b.v; }

// "no value array": reads give 0, writes vanish (the value travels inside a PK record)
struct NoVal {
	struct Ref {
		__device__ __forceinline__ operator u32() const { return 0u; }
		__device__ __forceinline__ const Ref& operator=(u32) const { return *this; }
	};
	__device__ __forceinline__ Ref operator[](long long) const { return Ref{}; }
	__device__ __forceinline__ NoVal operator+(long long) const { return NoVal{}; }
};

// the sorted hit keys, typed at compile time: u64 = (extId << 32 | curPos),
// u32 = ((extId - firstId) << curBits | curPos), PK = the packed record above
template <class KT> struct HitKeyView;
template <> struct HitKeyView<u64> {
	const u64* k; const u32* v; int curBits; u32 firstId;
	__device__ __forceinline__ u32 ext(u64 i) const { return (u32)(k[i] >> 32); }
	__device__ __forceinline__ u32 ext_raw(u64 i) const { return (u32)(k[i] >> 32); }
	__device__ __forceinline__ u32 cur(u64 i) const { return (u32)k[i]; }
	__device__ __forceinline__ u32 val(u64 i) const { return v[i]; }
};
template <> struct HitKeyView<u32> {
	const u32* k; const u32* v; int curBits; u32 firstId;
	__device__ __forceinline__ u32 ext(u64 i) const { return (k[i] >> curBits) + firstId; }
	__device__ __forceinline__ u32 ext_raw(u64 i) const { return k[i] >> curBits; }
	__device__ __forceinline__ u32 cur(u64 i) const { return k[i] & ((1u << curBits) - 1u); }
	__device__ __forceinline__ u32 val(u64 i) const { return v[i]; }
};
template <> struct HitKeyView<PK> {
	const PK* k; const u32* v; int curBits; u32 firstId;
	__device__ __forceinline__ u32 ext(u64 i) const { return (u32)(k[i].v >> (FG_PK_VALBITS + curBits)) + firstId; }
	__device__ __forceinline__ u32 ext_raw(u64 i) const { return (u32)(k[i].v >> (FG_PK_VALBITS + curBits)); }
	__device__ __forceinline__ u32 cur(u64 i) const { return (u32)(k[i].v >> FG_PK_VALBITS) & ((1u << curBits) - 1u); }
	__device__ __forceinline__ u32 val(u64 i) const { return (u32)k[i].v & ((1u << FG_PK_VALBITS) - 1u); }
};

__device__ __forceinline__ u64 fg_mix(u64 x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
	x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
	return x;
}

// splitmix64 finaliser = Kmer::hash() (reference src/sequence/kmer.h:91-98)
__device__ __forceinline__ u64 fg_kmer_hash(u64 x)
{
	u64 z = (x += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

// 2k bits of the packed forward read starting at base q (base q in bits 0-1).
// `w` points at the read's first word; the caller guarantees q + k <= len.
__device__ __forceinline__ u64 fg_chunk(const u64* __restrict__ w, i32 q, int k)
{
	const int sh = (q & 31) * 2;
	u64 lo = w[q >> 5] >> sh;
	if (sh + 2 * k > 64) lo |= w[(q >> 5) + 1] << (64 - sh);
	return (k == 32) ? lo : (lo & ((1ULL << (2 * k)) - 1));
}

// reverse the order of the k 2-bit groups: packed chunk -> Kmer repr (first
// base most significant, kmer.h:32-36)
__device__ __forceinline__ u64 fg_rev2(u64 x, int k)
{
	u64 r = __brevll(x);
	r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
	return r >> (64 - 2 * k);
}

// forward / reverse-complement representation of the k-mer at forward position
// q; rc repr is simply the complemented packed chunk (kmer.h:39-52)
__device__ __forceinline__ void fg_kmer_pair(const u64* __restrict__ w, i32 q, int k, u64& fw, u64& rv)
{
	const u64 x = fg_chunk(w, q, k);
	const u64 mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
	fw = fg_rev2(x, k);
	rv = ~x & mask;
}

// the slot of `key` in part p of a narrow table, starting at group g: the slot's value, or FG_EMPTY_KEY when absent
__device__ __forceinline__ u64 fg_probe_slot(const FgTable& T, u32 p, u32 g, u64 key)
{
	const u32 nGroups = T.groups[p];
	while (true)
	{
		const ulonglong2* grp = (const ulonglong2*)(T.slots + T.slotBase[p] + (u64)g * 8u);
		const ulonglong2 a = grp[0], b = grp[1], c = grp[2], d = grp[3];
		const u64 sl[8] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
		u64 hit = FG_EMPTY_KEY;
		bool empty = false;
#pragma unroll
		for (int i = 0; i < 8; ++i)
		{
			if ((sl[i] >> FG_IDX_BITS) == key) hit = sl[i];
			empty |= sl[i] == FG_EMPTY_KEY;
		}
		if (hit != FG_EMPTY_KEY || empty) return hit;
		g = g + 1 == nGroups ? 0 : g + 1;
	}
}

// probe: returns off << 24 | cnt of the k-mer's list, FG_CNT_REPETITIVE in the low bits for a k-mer of
// _repetitiveKmers, or 0 when absent.  One 8-byte slot read decides a miss (4 in 5 query k-mers: the
// table is half the size of a 16-byte layout -- E. coli 50x: 134 MB, inside the 256 MiB Infinity Cache);
// a hit reads the list bounds from the key-ordered offsets.
template <bool WIDE>
__device__ __forceinline__ u64 fg_probe(const FgTable& T, u64 key)
{
	u32 p = 0;
	if (T.nParts > 1)
		while (p + 1 < T.nParts && key >= T.bound[p + 1]) ++p;
	const u64 mix = fg_mix(key);
	const u32 nGroups = T.groups[p];
	u32 g = __umulhi((u32)(mix >> 32), nGroups);
	if (WIDE)
	{
		// {key, index} pairs, linear probing over the part's slots
		const u32 slotsInPart = nGroups * 8u;
		u32 h = g * 8u + ((u32)mix & 7u);
		while (true)
		{
			const ulonglong2 s = T.wide[T.slotBase[p] + h];
			if (s.x == FG_EMPTY_KEY) return 0;
			if (s.x != key) { h = h + 1 == slotsInPart ? 0 : h + 1; continue; }
			if (s.y == FG_EMPTY_KEY) return FG_CNT_REPETITIVE;
			const u64 o = T.keyOff[s.y];
			return (o << FG_CNT_BITS) | (T.keyOff[s.y + 1] - o);
		}
	}
	// narrow: the key's 8-slot group is one 64-byte line, fetched whole; insertion fills a group before it
	// spills into the next, so an empty slot anywhere in the group settles a miss without a second access
	const u64 hit = fg_probe_slot(T, p, g, key);
	if (hit == FG_EMPTY_KEY) return 0;
	if ((hit & FG_IDX_MASK) == FG_IDX_MASK) return FG_CNT_REPETITIVE;
	const u64 idx = T.keyBase[p] + (hit & FG_IDX_MASK);
	const u64 o = T.keyOff[idx];
	return (o << FG_CNT_BITS) | (T.keyOff[idx + 1] - o);
}

#endif // __HIPCC__

// result arena of one fg_overlaps call.  recs is a raw grow-only buffer (no value
// initialisation of ~100 MB per call); released arenas go back to a small per-process pool
// so that steady-state calls neither page-fault nor munmap.
struct BatchOwner {
	std::vector<u64> queryOff, statOff;
	fg_overlap_rec* recs = nullptr;
	size_t recCap = 0, nRecs = 0;
	std::vector<float> stats;
	std::vector<uint8_t> needsTrim;	// partition_bad_mappings: nRecs
	std::vector<u64> matchOff;	// keep_alignment: nRecs + 1, in pairs
	int32_t* matches = nullptr;	// (cur, ext) pairs
	size_t matchCap = 0;
	void reserveMatches(size_t nPairs)
	{
		if (nPairs <= matchCap) return;
		free(matches);
		matchCap = nPairs + nPairs / 8 + 16;
		matches = (int32_t*)malloc(matchCap * 8);
		if (!matches) { matchCap = 0; throw std::bad_alloc(); }
	}
	void reserveRecs(size_t n)
	{
		if (n <= recCap) return;
		free(recs);
		recCap = n + n / 8 + 16;
		recs = (fg_overlap_rec*)malloc(recCap * sizeof(fg_overlap_rec));
		if (!recs) { recCap = 0; throw std::bad_alloc(); }
	}
	~BatchOwner() { free(recs); free(matches); }
	static BatchOwner* acquire();
	static void release(BatchOwner* b);
};

// one primary overlap candidate as the device hands it to the host shim
struct PrimRec {
	u32 query;		// index into the batch
	u32 extId;
	i32 curBegin, curEnd, extBegin, extEnd, extLen, score, chainLength, filtered;
	i32 editDistance, hpcLenCur, hpcLenExt;
};

// implemented in fg_index.hip / fg_overlap.hip
void fgBuildIndexSolid(fg_ctx* c, i32 minFreq, float selectRate, i32 tandemFreq, float repeatRate,
					   float sampleRateInit, fg_index_stats* st);
void fgBuildIndexMinimizers(fg_ctx* c, i32 minCoverage, i32 window, float repeatRate, fg_index_stats* st);
// the same in steps (sharded builds): begin -> ranges of key bins -> finish; fg_index.hip
void fgIndexBeginSolid(fg_ctx* c, i32 minFreq, float selectRate, i32 tandemFreq, float repeatRate,
					   float sampleRateInit, u64* histOut);
// the solid selection in steps (bounded memory, counters of a key range only): count -> per batch of reads
// {frequencies -> [sum over ranks] -> select} -> done
void fgIndexKmerHist(fg_ctx* c, u64* histOut);
void fgIndexCountSlice(fg_ctx* c, i32 minFreq, float selectRate, i32 tandemFreq, float repeatRate, float sampleRateInit,
					   u32 binLo, u32 binHi, u64* distinctOut, u32* nBatchesOut);
void fgIndexBatchFreq(fg_ctx* c, u32 batch, u32** dFreq, u64* nPos);
void fgIndexBatchSelect(fg_ctx* c, u32 batch);
void fgIndexSelectionDone(fg_ctx* c, u64* histOut);
void fgIndexGatherBegin(fg_ctx* c, u64 nKeys, u64 nEntries, u64 nRep, u64** full, u64** piece, u64* pieceSizes);
void fgIndexGatherEnd(fg_ctx* c, float sampleRate);
void fgIndexBeginMinimizers(fg_ctx* c, i32 minCoverage, i32 window, float repeatRate, u64* histOut);
void fgIndexBuildRange(fg_ctx* c, u32 binLo, u32 binHi, unsigned long long* sumsOut);
void fgIndexFinish(fg_ctx* c, const unsigned long long* totalSums, fg_index_stats* st);
void fgIndexLookupStructures(fg_ctx* c, bool bitsHoldSelection);
void fgImportIndex(fg_ctx* c, u64 nKeys, const u64* keys, const u64* keyOff, u64 nEnt, const u64* entries, u64 nRep,
				   const u64* repKeys, float sampleRate, int onDevice);
// keyMode: 0 = 32-bit keys, 1 = packed 64-bit records (PK), 2 = 64-bit keys + values
u32 fgChainSmallMax();
void fgChainStage(fg_ctx* c, const fg_detector_params* p, uint8_t forceLocal, u64 nGroups, u64 nHits, int keyMode,
				  int curBits);
void fgEditDistances(fg_ctx* c, PrimRec* dPrims, u64 nPrim, int useHpc);
void fgKswAlign(fg_ctx* c, u32 nPairs, const uint8_t* trg, const u64* trgOff, const uint8_t* qry, const u64* qryOff,
				std::vector<u64>& runOff, std::vector<u32>& runs);
void fgDebugSortPairs(fg_ctx* c, u64* keys, u32* vals, const u64* segOff, u32 nSeg);
void fgDebugEditDistances(fg_ctx* c, u32 nPairs, int useHpc, i32* outDist, i32* outLenA, i32* outLenB);
void fgOverlaps(fg_ctx* c, const fg_detector_params* p, const u32* queryIds, u32 nq, i32 maxOverlaps,
				uint8_t forceLocal, fg_overlap_batch* out);
