// Does initialising the HIP runtime / creating a context consume values of libc's rand() stream?
// (glibc, default seed 1: the first value is 1804289383.)  Matters for bit parity of
// OverlapContainer::estimateOverlaperParameters (overlap.cpp:752-756), which draws its reads with rand().
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
int main()
{
	int n = 0;
	(void)hipGetDeviceCount(&n);
	void* p = nullptr;
	(void)hipMalloc(&p, 1 << 20);
	hipStream_t s; (void)hipStreamCreate(&s);
	(void)hipStreamSynchronize(s);
	printf("devices %d; first rand() after HIP init: %d (untouched stream: 1804289383)\n", n, rand());
	return 0;
}
