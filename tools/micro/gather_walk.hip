// Microbenchmark for the lane-per-group DP idea: every lane walks backwards through its own
// region of 16-byte records (one gather per step, dependent on a trivial computation).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int VEC>
__global__ void walk(const int4* __restrict__ nodes, int regionLen, int steps, int* __restrict__ out, int nRegions)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nRegions) return;
	const int4* base = nodes + (size_t)r * regionLen;
	int j = regionLen - 1;
	int acc = 0, best = 0;
	for (int s = 0; s < steps; ++s)
	{
		int4 v;
		if (VEC == 4) v = base[j];
		else { const int* p = (const int*)(base + j); v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = 0; }
		const int dc = acc - v.x, de = acc - v.y;
		const int jd = abs(dc - de);
		const int ns = v.z + min(min(dc, de), 17) - (jd > 100 ? 2 * jd : (jd >> 1));
		if (ns > best) best = ns;
		acc += v.x & 3;
		j = j > 0 ? j - 1 : regionLen - 1;
	}
	out[r] = best + acc;
}

int main(int argc, char** argv)
{
	const int regionLen = argc > 1 ? atoi(argv[1]) : 256;	// records per lane region
	const int nRegions = argc > 2 ? atoi(argv[2]) : (1 << 20);
	const int steps = argc > 3 ? atoi(argv[3]) : 1024;
	int4* d; int* o;
	const size_t n = (size_t)regionLen * nRegions;
	CK(hipMalloc(&d, n * sizeof(int4))); CK(hipMalloc(&o, nRegions * 4));
	CK(hipMemset(d, 1, n * sizeof(int4)));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	for (int vec : {4, 3})
		for (int rep = 0; rep < 2; ++rep)
		{
			CK(hipEventRecord(a));
			if (vec == 4) hipLaunchKernelGGL(walk<4>, nRegions / 256, 256, 0, 0, d, regionLen, steps, o, nRegions);
			else hipLaunchKernelGGL(walk<3>, nRegions / 256, 256, 0, 0, d, regionLen, steps, o, nRegions);
			CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
			float ms; CK(hipEventElapsedTime(&ms, a, b));
			printf("regionLen %d regions %d steps %d vec %d: %.3f ms -> %.2f G lane-steps/s (%.1f MB table)\n", regionLen, nRegions, steps,
				   vec, ms, (double)nRegions * steps / ms / 1e6, n * 16.0 / 1e6);
		}
	return 0;
}
