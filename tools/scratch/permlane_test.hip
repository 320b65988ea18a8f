// What do v_permlane32_swap / v_permlane16_swap do on gfx950, and does the folded 8-value wave reduction built on them
// put the totals where k_render_bwd expects them?  hipcc --offload-arch=gfx950 -O3 -o permlane_test permlane_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fold32(float a, float b)
{
	const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
	return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float fold16(float a, float b)
{
	const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
	return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float row_sum(float x)
{
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false));
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false));
	return x;
}
__global__ void k(float *out)
{
	const int lane = threadIdx.x;
	// raw semantics
	const u2 r = __builtin_amdgcn_permlane32_swap(1000u + lane, 2000u + lane, false, false);
	out[lane] = (float)r.x; out[64 + lane] = (float)r.y;
	const u2 q = __builtin_amdgcn_permlane16_swap(1000u + lane, 2000u + lane, false, false);
	out[128 + lane] = (float)q.x; out[192 + lane] = (float)q.y;
	// folded reduction of 8 values: value i of lane l = (i + 1) * 100 + l  -> total = 64 (i + 1) 100 + 2016
	float v[8];
	for (int i = 0; i < 8; i++) v[i] = (float)((i + 1) * 100 + lane);
	const float f0 = fold32(v[0], v[4]), f1 = fold32(v[1], v[5]), f2 = fold32(v[2], v[6]), f3 = fold32(v[3], v[7]);
	const float g0 = row_sum(fold16(f0, f2)), g1 = row_sum(fold16(f1, f3));
	out[256 + lane] = g0; out[320 + lane] = g1;
}
int main()
{
	float *d, h[384];
	hipMalloc(&d, sizeof(h));
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
	hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	const char *names[4] = { "permlane32_swap .x", "permlane32_swap .y", "permlane16_swap .x", "permlane16_swap .y" };
	for (int a = 0; a < 4; a++) { printf("%s:", names[a]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%g", l, h[64 * a + l]); printf("\n"); }
	for (int r = 0; r < 4; r++)
		printf("row %d: g0 = %g (value id %g)  g1 = %g (value id %g)\n", r, h[256 + 16 * r], (h[256 + 16 * r] - 2016) / 6400 - 1, h[320 + 16 * r], (h[320 + 16 * r] - 2016) / 6400 - 1);
	return 0;
}
