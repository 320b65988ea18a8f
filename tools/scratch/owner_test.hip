#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ int wave_scan_max_i32(int v)
{
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x111, 0xF, 0xF, false));
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x112, 0xF, 0xF, false));
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x114, 0xF, 0xF, false));
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x118, 0xF, 0xF, false));
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x142, 0xA, 0xF, false));
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x143, 0xC, 0xF, false));
	return v;
}
__device__ __forceinline__ int pair_owner_scan(int *row, int lane, int seg_a, int seg_b)
{
	// lanes talk to each other through LDS here: wave-scope fences + wave barrier make that defined (without
	// them the compiler forwards this lane's own -1 to the load)
	row[lane] = -1;
	if (seg_b > seg_a) row[seg_a] = lane;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	const int m = row[lane];
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	return wave_scan_max_i32(m);
}
__device__ __forceinline__ int pair_owner(uint32_t incl, uint32_t j)
{
	int lo = 0, hi = 63;
	for (int it = 0; it < 6; it++) { const int mid = (lo + hi) >> 1; const uint32_t v = (uint32_t)__shfl((int)incl, mid); if (v > j) hi = mid; else lo = mid + 1; }
	return lo;
}
__global__ void k(const uint32_t *n, int *bad)
{
	__shared__ int s_own[256];
	const int lane = threadIdx.x & 63;
	const uint32_t my_n = n[blockIdx.x * 256 + threadIdx.x];
	uint32_t incl = my_n;
	for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off); if (lane >= off) incl += t; }
	const uint32_t excl = incl - my_n;
	const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
	for (uint32_t kk = 0; kk < total; kk += 64)
	{
		const uint32_t j = kk + lane;
		const bool valid = j < total;
		const int seg_a = (int)max((long long)excl - (long long)kk, 0ll);
		const int seg_b = (int)min((long long)incl - (long long)kk, 64ll);
		const int o1 = max(pair_owner_scan(s_own + (threadIdx.x & ~63), lane, seg_a, seg_b), 0);
		const int o2 = pair_owner(incl, valid ? j : total - 1);
		if (valid && o1 != o2) atomicAdd(bad, 1);
	}
}
int main()
{
	const int N = 256 * 64;
	uint32_t *h = (uint32_t *)malloc(N * 4), *d; int *db, hb = 0;
	srand(1);
	for (int i = 0; i < N; i++) { int r = rand() % 100; h[i] = r < 60 ? 0 : (r < 90 ? rand() % 20 : (r < 98 ? rand() % 300 : rand() % 9000)); }
	hipMalloc(&d, N * 4); hipMalloc(&db, 4);
	hipMemcpy(d, h, N * 4, hipMemcpyHostToDevice); hipMemset(db, 0, 4);
	hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, 0, d, db);
	hipMemcpy(&hb, db, 4, hipMemcpyDeviceToHost);
	printf("mismatches=%d err=%s\n", hb, hipGetErrorString(hipGetLastError()));
	return 0;
}
