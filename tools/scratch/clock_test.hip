// Shader clock and dependent-VALU latency on the device: hipcc --offload-arch=gfx950 -O3 clock_test.hip -o clock_test && ./clock_test
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out, long long *t, int n, float a, float b)
{
	float x = out[threadIdx.x];
	const long long c0 = clock64(), w0 = wall_clock64();
	for (int i = 0; i < n; i++) { x = fmaf(x, a, b); x = fmaf(x, a, b); x = fmaf(x, a, b); x = fmaf(x, a, b); }
	const long long c1 = clock64(), w1 = wall_clock64();
	out[threadIdx.x] = x;
	if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main()
{
	float *d; long long *t, h[2];
	hipMalloc(&d, 4096); hipMemset(d, 0, 4096); hipMalloc(&t, 16);
	for (int blocks : { 1, 1024, 8192 })
		for (int threads : { 64, 256 })
		{
			const int n = 200000;
			hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, t, n, 1.0001f, 0.5f);
			hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
			printf("blocks %5d x %3d: %lld clk, %lld x10ns -> clk/fma %.2f, ns/fma %.2f, s_memtime MHz %.0f\n", blocks, threads, h[0], h[1],
				h[0] / (4.0 * n), h[1] * 10.0 / (4.0 * n), h[0] / (h[1] * 10.0) * 1000.0);
		}
	return 0;
}
