// How fast does MI355X deliver SCATTERED rows? 0.7 M of 6 M rows (a random subset in increasing order, like k_bin's candidates):
//  A  12 aligned 16-byte loads per lane from 192-byte rows
//  B  the foveated k_bin's colour reads: 11 x 16 B + 4 B from 180-byte rows (4-byte aligned), 3 x 16 B from 48-byte rows, 16 B
//  C  B + k_bin's scattered stores: 48-byte record, 32 bytes of a 64-byte level row, two 4-byte words, all at the Gaussian's index
//  D  B + the same bytes stored at the CANDIDATE's index (dense, coalesced); only the radius goes to the Gaussian's index
// build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

struct Bufs { const char *sh; const char *dc; const char *op; float4 *rec; float4 *lvl; unsigned *lr; int *radii; };

template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const Bufs b, const unsigned *idx, int n, float *sink)
{
	float acc = 0.f;
	for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
	{
		const unsigned g = idx[i];
		if (MODE == 0)
		{
			const float4 *row = (const float4 *)(b.sh + (size_t)g * 192);
			float4 v[12];
#pragma unroll
			for (int k = 0; k < 12; k++) v[k] = row[k];
#pragma unroll
			for (int k = 0; k < 12; k++) acc += v[k].x + v[k].w;
			continue;
		}
		const float *sh = (const float *)(b.sh + (size_t)g * 180);
		f4u v[11];
#pragma unroll
		for (int k = 0; k < 11; k++) v[k] = *(const f4u *)(sh + 4 * k);
		const float last = sh[44];
		const f4u *dcp = (const f4u *)(b.dc + (size_t)g * 48);
		const f4u d0 = dcp[0], d1 = dcp[1], d2 = dcp[2];
		const f4u op = *(const f4u *)(b.op + (size_t)g * 16);
		float s = last + d0.x + d1.y + d2.z + op.w;
#pragma unroll
		for (int k = 0; k < 11; k++) s += v[k].x * v[k].y + v[k].z * v[k].w;
		acc += s;
		if (MODE == 2 || MODE == 3)
		{
			const size_t o = MODE == 2 ? g : (size_t)i;
			b.rec[3 * o] = make_float4(s, s, s, s); b.rec[3 * o + 1] = make_float4(s, 1, 2, 3); b.rec[3 * o + 2] = make_float4(s, 4, 5, 6);
			b.lvl[4 * o] = make_float4(s, 0, 0, 0); b.lvl[4 * o + 1] = make_float4(s, 0, 0, 1);
			b.lr[o] = (unsigned)i;
			b.radii[g] = i; // (the radii are an output tensor: always at the Gaussian's index)
		}
		if (MODE == 4) b.radii[g] = i;                 // E: one scattered 4-byte store
		if (MODE == 5) { b.rec[3 * (size_t)g] = make_float4(s, s, s, s); b.rec[3 * (size_t)g + 1] = make_float4(s, 1, 2, 3); b.rec[3 * (size_t)g + 2] = make_float4(s, 4, 5, 6); } // F: scattered 48 B
		if (MODE == 6) { b.rec[3 * (size_t)i] = make_float4(s, s, s, s); b.rec[3 * (size_t)i + 1] = make_float4(s, 1, 2, 3); b.rec[3 * (size_t)i + 2] = make_float4(s, 4, 5, 6); } // G: dense 48 B
		if (MODE == 7) { b.lvl[4 * (size_t)g] = make_float4(s, 0, 0, 0); b.lvl[4 * (size_t)g + 1] = make_float4(s, 0, 0, 1); b.lvl[4 * (size_t)g + 2] = make_float4(s, 0, 0, 0); b.lvl[4 * (size_t)g + 3] = make_float4(s, 0, 0, 1); } // H: scattered aligned 64 B
	}
	if (acc == 12345.678f) sink[0] = acc;
}

int main()
{
	const size_t P = 6000000;
	char *sh, *dc, *op; float4 *rec, *lvl; unsigned *lr, *didx; int *radii; float *sink;
	CK(hipMalloc(&sh, P * 192)); CK(hipMemset(sh, 0, P * 192)); CK(hipMalloc(&dc, P * 48)); CK(hipMemset(dc, 0, P * 48));
	CK(hipMalloc(&op, P * 16)); CK(hipMemset(op, 0, P * 16)); CK(hipMalloc(&rec, P * 48)); CK(hipMalloc(&lvl, P * 64));
	CK(hipMalloc(&lr, P * 4)); CK(hipMalloc(&radii, P * 4)); CK(hipMalloc(&sink, 4));
	const Bufs b = { sh, dc, op, rec, lvl, lr, radii };
	std::mt19937 rng(1);
	const int nsel = 700000;
	std::vector<unsigned> all(P); for (size_t i = 0; i < P; i++) all[i] = (unsigned)i;
	std::shuffle(all.begin(), all.end(), rng);
	std::vector<unsigned> sel(all.begin(), all.begin() + nsel);
	std::sort(sel.begin(), sel.end());
	CK(hipMalloc(&didx, nsel * 4)); CK(hipMemcpy(didx, sel.data(), nsel * 4, hipMemcpyHostToDevice));
	for (int wg_per_cu : { 2, 8 })
	{
		auto run = [&](auto kern, const char *name) {
			hipEvent_t a, e; CK(hipEventCreate(&a)); CK(hipEventCreate(&e));
			const int grid = 256 * wg_per_cu;
			hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, b, didx, nsel, sink);
			CK(hipEventRecord(a));
			for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, b, didx, nsel, sink);
			CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
			float ms; CK(hipEventElapsedTime(&ms, a, e)); ms /= 5;
			printf("waves/SIMD=%d %s: %7.1f us  rows/us %7.0f\n", wg_per_cu, name, ms * 1e3, nsel / (ms * 1e3));
		};
		run(k_gather<0>, "A 12x16B aligned loads, 192-B rows      ");
		run(k_gather<1>, "B k_bin's colour reads (180-B rows ...) ");
		run(k_gather<2>, "C B + stores at the Gaussian's index     ");
		run(k_gather<3>, "D B + stores at the candidate's index    ");
		run(k_gather<4>, "E B + one scattered 4-byte store         ");
		run(k_gather<5>, "F B + scattered 48-byte record           ");
		run(k_gather<6>, "G B + dense 48-byte record               ");
		run(k_gather<7>, "H B + scattered aligned 64-byte row      ");
	}
	return 0;
}
