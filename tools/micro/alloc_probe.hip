// What does device memory cost to get?  hipMalloc / first-touch memset / second memset / hipFree by size,
// and the same through a stream-ordered pool (hipMallocAsync).  Guides the sizing of the index build's scratch.
//   hipcc --offload-arch=gfx950 -O2 -o alloc_probe alloc_probe.hip && ./alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main()
{
	CK(hipSetDevice(0));
	CK(hipFree(nullptr));
	hipStream_t s; CK(hipStreamCreate(&s));
	const size_t sizes[] = {64ULL << 20, 512ULL << 20, 2ULL << 30, 8ULL << 30, 32ULL << 30, 64ULL << 30};
	for (int round = 0; round < 2; ++round)
		for (size_t sz : sizes)
		{
			void* p = nullptr;
			double t0 = now(); CK(hipMalloc(&p, sz)); double t1 = now();
			CK(hipMemsetAsync(p, 0, sz, s)); CK(hipStreamSynchronize(s)); double t2 = now();
			CK(hipMemsetAsync(p, 0, sz, s)); CK(hipStreamSynchronize(s)); double t3 = now();
			CK(hipFree(p)); double t4 = now();
			printf("round %d hipMalloc %6.2f GB: malloc %8.2f ms, first memset %8.2f ms, second memset %8.2f ms (%.0f GB/s), free %8.2f ms\n",
				   round, sz / 1e9, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, sz / 1e9 / (t3 - t2), (t4 - t3) * 1e3);
			fflush(stdout);
		}
	// is it the size of ONE allocation?  64 GiB as 2 x 32, 4 x 16, 8 x 8 GiB held at the same time
	for (int parts : {2, 4, 8, 1})
	{
		void* p[8] = {};
		const size_t each = (64ULL << 30) / parts;
		double t0 = now();
		for (int i = 0; i < parts; ++i) CK(hipMalloc(&p[i], each));
		double t1 = now();
		for (int i = 0; i < parts; ++i) CK(hipMemsetAsync(p[i], 0, each, s));
		CK(hipStreamSynchronize(s)); double t2 = now();
		for (int i = 0; i < parts; ++i) CK(hipFree(p[i]));
		double t3 = now();
		printf("64 GiB as %d x %.0f GiB: malloc %8.2f ms, memset %8.2f ms, free %8.2f ms\n", parts, each / double(1ULL << 30),
			   (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
		fflush(stdout);
	}
	// one allocation just below / above 64 GiB, and 128 GiB
	for (size_t sz : {60ULL << 30, (64ULL << 30) - (2ULL << 20), 64ULL << 30, 66ULL << 30, 128ULL << 30})
	{
		void* p = nullptr;
		double t0 = now(); CK(hipMalloc(&p, sz)); double t1 = now();
		CK(hipFree(p)); double t2 = now();
		printf("hipMalloc %.3f GiB: malloc %8.2f ms, free %8.2f ms\n", sz / double(1ULL << 30), (t1 - t0) * 1e3, (t2 - t1) * 1e3);
		fflush(stdout);
	}
	// stream-ordered pool with a release threshold: the second allocation of a size comes out of the pool
	hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0));
	unsigned long long thr = ~0ULL; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
	for (int round = 0; round < 2; ++round)
		for (size_t sz : sizes)
		{
			void* p = nullptr;
			double t0 = now(); CK(hipMallocAsync(&p, sz, s)); CK(hipStreamSynchronize(s)); double t1 = now();
			CK(hipMemsetAsync(p, 0, sz, s)); CK(hipStreamSynchronize(s)); double t2 = now();
			CK(hipFreeAsync(p, s)); CK(hipStreamSynchronize(s)); double t3 = now();
			printf("round %d hipMallocAsync %6.2f GB: malloc %8.2f ms, memset %8.2f ms, free %8.2f ms\n", round, sz / 1e9,
				   (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
			fflush(stdout);
		}
	return 0;
}
