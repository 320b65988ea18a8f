// Do kernels on two streams overlap on this GPU? Each kernel: `blocks` workgroups of 256 threads spinning for ~`us` microseconds.
// usage: ./overlap_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void spin(long long ticks, int *sink)
{
	const long long t0 = wall_clock64();
	int x = 0;
	while (wall_clock64() - t0 < ticks) x++;
	if (x == -1) *sink = x;
}
// memory-streaming kernel: reads n floats
__global__ void stream_read(const float4 *p, size_t n, float *sink)
{
	float s = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
	if (s == 1.234f) *sink = s;
}
int main()
{
	hipStream_t a, b;
	hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
	int *sink; hipMalloc(&sink, 64);
	float4 *buf; const size_t n = (size_t)64 << 20; hipMalloc(&buf, n * 16); hipMemset(buf, 0, n * 16);
	const long long ticks = 100 * 100; // wall_clock64 ticks at 100 MHz: 100 us
	auto run = [&](const char *name, int blocksA, int blocksB, bool two_streams, int kind) {
		for (int rep = 0; rep < 3; rep++)
		{
			hipDeviceSynchronize();
			auto t0 = std::chrono::high_resolution_clock::now();
			for (int i = 0; i < 10; i++)
			{
				hipLaunchKernelGGL(spin, dim3(blocksA), dim3(256), 0, a, ticks, sink);
				if (kind == 0) hipLaunchKernelGGL(spin, dim3(blocksB), dim3(256), 0, two_streams ? b : a, ticks, sink);
				else hipLaunchKernelGGL(stream_read, dim3(blocksB), dim3(256), 0, two_streams ? b : a, buf, n, (float *)sink);
			}
			hipDeviceSynchronize();
			const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 10;
			if (rep == 2) printf("%-60s %8.1f us per pair\n", name, us);
		}
	};
	run("spin 256 blocks + spin 256 blocks, one stream", 256, 256, false, 0);
	run("spin 256 blocks + spin 256 blocks, two streams", 256, 256, true, 0);
	run("spin 2048 blocks + spin 2048 blocks, one stream", 2048, 2048, false, 0);
	run("spin 2048 blocks + spin 2048 blocks, two streams", 2048, 2048, true, 0);
	run("spin 4096 blocks (2 rounds?) + spin 256, two streams", 4096, 256, true, 0);
	run("spin 256 blocks + read 1 GiB (2048 blocks), one stream", 256, 2048, false, 1);
	run("spin 256 blocks + read 1 GiB (2048 blocks), two streams", 256, 2048, true, 1);
	run("spin 2048 blocks + read 1 GiB (2048 blocks), two streams", 2048, 2048, true, 1);
	run("spin 2048 blocks + read 1 GiB (512 blocks), two streams", 2048, 512, true, 1);
	return 0;
}
