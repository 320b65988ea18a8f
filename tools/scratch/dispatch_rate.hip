// How fast does the chip start single-wave workgroups? Each workgroup spins for `us` microseconds (wall clock) and
// exits; grid = n workgroups of 64 threads. With 8192 wave slots the ideal time is ceil(n / 8192) * us.
// Build: hipcc --offload-arch=gfx950 -O3 -o dispatch_rate dispatch_rate.hip ; run: ./dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int LDS_WORDS>
__global__ void __launch_bounds__(64) k_spin(int ticks, int jitter, unsigned *sink)
{
	__shared__ unsigned lds[LDS_WORDS > 0 ? LDS_WORDS : 1];
	if (LDS_WORDS > 0) lds[threadIdx.x] = threadIdx.x;
	// durations differ from workgroup to workgroup like the blend waves' do (hash of the block id)
	const unsigned h = (blockIdx.x * 2654435761u) >> 16;
	const long long want = ticks + (jitter ? (long long)(h % (unsigned)jitter) : 0);
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < want) __builtin_amdgcn_s_sleep(1);
	if (LDS_WORDS > 0 && lds[(threadIdx.x + 1) & 63] == 0xffffffffu) sink[0] = 1;
}
template <int L>
static void run(const char *name, int n, int us, int jitter_us)
{
	unsigned *sink; hipMalloc(&sink, 4);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	float best = 1e9f;
	for (int rep = 0; rep < 5; rep++)
	{
		hipEventRecord(a);
		hipLaunchKernelGGL(k_spin<L>, dim3(n), dim3(64), 0, 0, us * 100, jitter_us * 100, sink);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		if (ms < best) best = ms;
	}
	const float mean = us + jitter_us * 0.5f;
	printf("%-10s n=%6d spin=%3d+[0,%3d) us: %8.1f us  (ideal %7.1f us at 8192 slots; %6.0f workgroups/us)\n", name, n, us, jitter_us, best * 1e3f,
		(float)n / 8192.0f * mean, n / (best * 1e3f));
	hipFree(sink);
}
int main()
{
	for (int n : { 8192, 32768 })
		for (int us : { 0, 5, 20, 50 })
		{
			run<0>("no LDS", n, us, 0);
			run<896>("3.5 KB LDS", n, us, 0);
		}
	run<0>("no LDS", 32768, 10, 80);
	run<896>("3.5 KB LDS", 32768, 10, 80);
	run<896>("3.5 KB LDS", 16384, 10, 80);
	return 0;
}
