// Test of the paired cross-lane fold of k_render_bwd (two entries x nine per-lane partial sums -> 18 totals in 18 different lanes).
// build: hipcc --offload-arch=gfx950 -O3 -o fold18_test fold18_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef unsigned int bwd_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fold32(float a, float b)
{
	const bwd_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
	return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float fold16(float a, float b)
{
	const bwd_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
	return __uint_as_float(r.x) + __uint_as_float(r.y);
}
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ float dpp_add(float x)
{
	return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWMASK, 0xF, false));
}
// -> the lane's total (valid in the writer lanes), the entry (0 / 1) and the component (0..8) it belongs to
__device__ __forceinline__ float fold18(const float (&a)[9], const float (&b)[9], const int lane, int &entry, int &comp, bool &writer)
{
	float f[9];
#pragma unroll
	for (int i = 0; i < 9; i++) f[i] = fold32(a[i], b[i]);         // lanes 0-31: a's partial sums, 32-63: b's
	float g[4];
#pragma unroll
	for (int k = 0; k < 4; k++) g[k] = fold16(f[2 * k], f[2 * k + 1]); // rows: (a, 2k) (a, 2k+1) (b, 2k) (b, 2k+1)
	const float s0 = dpp_add<0x128>(g[0]), s1 = dpp_add<0x128>(g[1]), s2 = dpp_add<0x128>(g[2]), s3 = dpp_add<0x128>(g[3]); // row_ror:8
	const bool hi8 = (lane & 8) != 0;
	const float h0 = hi8 ? s1 : s0, h1 = hi8 ? s3 : s2;              // half-rows: k = 0 | 1 and k = 2 | 3
	const float u0 = dpp_add<0x141>(h0), u1 = dpp_add<0x141>(h1);    // row_half_mirror
	float m = (lane & 4) ? u1 : u0;
	m = dpp_add<0xB1>(m); m = dpp_add<0x4E>(m);                       // the quad's total in all its lanes
	float z = f[8];
	z = dpp_add<0xB1>(z); z = dpp_add<0x4E>(z); z = dpp_add<0x141>(z); z = dpp_add<0x140>(z); // row sums
	z = dpp_add<0x142, 0xA>(z);                                       // rows 1, 3 += lane 15 of rows 0, 2
	const int r = lane >> 4, p = (lane >> 3) & 1, which = (lane >> 2) & 1;
	const bool last = (lane & 31) == 31;
	entry = r >> 1;
	comp = last ? 8 : 2 * (which * 2 + p) + (r & 1);
	writer = last || (lane & 3) == 0;
	return last ? z : m;
}
__global__ void k(const float *in, float *out, int *meta)
{
	const int lane = threadIdx.x;
	float a[9], b[9];
	for (int i = 0; i < 9; i++) { a[i] = in[(0 * 9 + i) * 64 + lane]; b[i] = in[(1 * 9 + i) * 64 + lane]; }
	int e, c; bool w;
	const float v = fold18(a, b, lane, e, c, w);
	out[lane] = v; meta[lane] = w ? (e * 16 + c) : -1;
}
int main()
{
	float h[2 * 9 * 64], o[64]; int m[64];
	srand(1);
	for (auto &x : h) x = (float)(rand() % 2001 - 1000) / 64.0f;
	float *di, *dq; int *dm;
	hipMalloc(&di, sizeof(h)); hipMalloc(&dq, 256); hipMalloc(&dm, 256);
	hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dq, dm);
	hipMemcpy(o, dq, 256, hipMemcpyDeviceToHost); hipMemcpy(m, dm, 256, hipMemcpyDeviceToHost);
	int seen[2][9] = {}, bad = 0;
	for (int l = 0; l < 64; l++)
	{
		if (m[l] < 0) continue;
		const int e = m[l] / 16, c = m[l] % 16;
		double want = 0; for (int j = 0; j < 64; j++) want += h[(e * 9 + c) * 64 + j];
		seen[e][c]++;
		if (fabs(o[l] - want) > 1e-3) { bad++; printf("lane %d entry %d comp %d got %f want %f\n", l, e, c, o[l], want); }
	}
	for (int e = 0; e < 2; e++) for (int c = 0; c < 9; c++) if (seen[e][c] != 1) { bad++; printf("entry %d comp %d written by %d lanes\n", e, c, seen[e][c]); }
	printf("bad=%d\n", bad);
	return bad != 0;
}
