// Backward pass: per-tile back-to-front gradient traversal, then the per-Gaussian chain rule.
//
// Replaces (reference, paths under fov3dgs/submodules/diff-gaussian-rasterization/cuda_rasterizer/;
// the _pcheck_obb_sum copy differs only by the power<-4.5 skip at backward.cu:495):
//   renderCUDA (backward)     backward.cu:399-557
//   computeCov2DCUDA          backward.cu:144-274
//   preprocessCUDA (backward) backward.cu:346-396  (+ computeColorFromSH bwd :20-139, computeCov3D bwd :278-341)
//
// MI355X design: the reference issues 9 float atomics per contributing (pixel, Gaussian) pair.
// Here every lane first sums over its PPL pixels, the 64 lanes of the wave are then reduced with
// DPP adds (all lanes look at the same Gaussian in lock-step), and one lane issues the 9 atomics:
// 256x fewer L2 atomics per tile-instance at PPL=4. computeCov2D and the preprocess backward are
// fused into one per-Gaussian kernel.
#include "common.h"

namespace fr {

__device__ __forceinline__ float bwd_exp(float p) { return __builtin_amdgcn_exp2f(p * 1.4426950408889634f); }

__device__ __forceinline__ float wave_sum_b(float x)
{
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false));
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false));
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xA, 0xF, false));
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xC, 0xF, false));
	return x; // total lives in lane 63
}

struct BwdRenderArgs {
	int W, H, gx;
	const uint2 *ranges;
	const uint32_t *tile_order;
	const uint32_t *point_list;
	const float4 *rec;
	const float *bg;
	const float *final_T;
	const uint32_t *n_contrib;
	const float *dL_dpix;
	float *dL_dmean2D;  // [P,3]
	float *dL_dconic;   // [P,4]
	float *dL_dopacity; // [P]
	float *dL_dcolor;   // [P,3]
};

template <bool CUTOFF, int PPL>
__global__ void __launch_bounds__(256 / PPL) k_render_bwd(const BwdRenderArgs a)
{
	constexpr int NT = 256 / PPL;
	constexpr int NW = NT / 64; // waves per tile, each owning a band of 16 / NW rows (see tile_row)
	__shared__ float4 s0[NT];
	__shared__ float4 s1[NT];
	__shared__ float s2[NT];
	__shared__ int sid[NT];
	__shared__ unsigned long long s_reach[NW][NW]; // [band][staging wave]: staged entries that can touch the band

	const int tile = (int)a.tile_order[blockIdx.x]; // longest lists first
	const int tx = tile % a.gx, ty = tile / a.gx;
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int lx = tid & 15;
	const int px = tx * FR_TILE + lx;
	const float pxf = (float)px;
	const uint2 range = a.ranges[tile];
	const int n = (int)(range.y - range.x);
	if (n == 0) return;
	const size_t plane = (size_t)a.W * a.H;
	const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
	const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;

	float T[PPL], Tfin[PPL], pyf[PPL], acc0[PPL], acc1[PPL], acc2[PPL], lastA[PPL], lc0[PPL], lc1[PPL], lc2[PPL];
	float dp0[PPL], dp1[PPL], dp2[PPL], bgdot[PPL];
	int lastc[PPL];
	int max_last = 0;
#pragma unroll
	for (int k = 0; k < PPL; k++)
	{
		const int py = ty * FR_TILE + tile_row<PPL>(tid, k);
		pyf[k] = (float)py;
		const bool inside = px < a.W && py < a.H;
		const size_t pid = (size_t)a.W * py + px;
		Tfin[k] = inside ? a.final_T[pid] : 0.0f;
		T[k] = Tfin[k];
		lastc[k] = inside ? (int)a.n_contrib[pid] : 0;
		max_last = max(max_last, lastc[k]);
		dp0[k] = inside ? a.dL_dpix[pid] : 0.0f;
		dp1[k] = inside ? a.dL_dpix[plane + pid] : 0.0f;
		dp2[k] = inside ? a.dL_dpix[2 * plane + pid] : 0.0f;
		bgdot[k] = bg0 * dp0[k] + bg1 * dp1[k] + bg2 * dp2[k];
		acc0[k] = acc1[k] = acc2[k] = 0.0f; lastA[k] = 0.0f; lc0[k] = lc1[k] = lc2[k] = 0.0f;
	}
	// nothing behind the deepest contributor of this wave / tile needs to be visited
	int wave_last = max_last;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, off));
	__shared__ int tile_last_s;
	if (tid == 0) tile_last_s = 0;
	__syncthreads();
	if (lane == 0) atomicMax(&tile_last_s, wave_last);
	__syncthreads();
	const int tile_last = tile_last_s; // entries [0, tile_last) can contribute
	if (tile_last == 0) return;

	// walk the list back to front, starting at the tile's deepest contributor; NT (= 64) entries per batch,
	// the next batch's records are prefetched into registers while this one is processed
	uint32_t pid = 0;
	float4 p0 = make_float4(0, 0, 0, 0), p1 = p0;
	float p2 = 0.f;
	if (tid < tile_last)
	{
		pid = a.point_list[range.x + tile_last - 1 - tid];
		const float4 *r = a.rec + 3 * (size_t)pid;
		p0 = r[0]; p1 = r[1]; p2 = r[2].x;
	}
	for (int top = tile_last; top > 0; top -= NT)
	{
		__syncthreads();
		const int cnt = min(NT, top);
		const bool staged = tid < cnt;
		if (staged) { s0[tid] = p0; s1[tid] = p1; s2[tid] = p2; sid[tid] = (int)pid; }
		{
			// entries that cannot touch a wave's rows are skipped for that wave: same thresholds as the per-pixel
			// tests below (power < -4.5, alpha < 1/255 <=> power < -ln(255 opacity)), see splat_reaches()
			const float thr_a = -__logf(255.0f * p1.y) - 0.01f;
			const float thr = CUTOFF ? fmaxf(-4.5f, thr_a) : thr_a;
#pragma unroll
			for (int w = 0; w < NW; w++)
			{
				const bool reach = staged && band_reaches<PPL>(w, tx, ty, p0.x, p0.y, p0.z, p0.w, p1.x, thr);
				const unsigned long long m = __ballot(reach);
				if (lane == 0) s_reach[w][tid >> 6] = m;
			}
		}
		if (top - NT - tid > 0)
		{
			pid = a.point_list[range.x + top - NT - 1 - tid];
			const float4 *r = a.rec + 3 * (size_t)pid;
			p0 = r[0]; p1 = r[1]; p2 = r[2].x;
		}
		__syncthreads();
		for (int sw = 0; sw < NW; sw++)
		for (unsigned long long rm = uniform_u64(s_reach[tid >> 6][sw]); rm; rm &= rm - 1)
		{
			const int j = sw * 64 + __builtin_ctzll(rm);
			const int pos = top - 1 - j; // 0-based position in the tile list
			if (pos >= wave_last) continue; // wave-uniform
			const float4 g0 = s0[j];
			const float4 g1 = s1[j];
			const float cb = s2[j];
			const float dx = g0.x - pxf;
			const float adx2 = (g0.z * dx) * dx;
			const float bdx = g0.w * dx;
			float r_c0 = 0, r_c1 = 0, r_c2 = 0, r_mx = 0, r_my = 0, r_ka = 0, r_kb = 0, r_kc = 0, r_op = 0;
			bool any = false;
#pragma unroll
			for (int k = 0; k < PPL; k++)
			{
				if (pos >= lastc[k]) continue;
				const float dy = g0.y - pyf[k];
				const float s = fmaf(g1.x * dy, dy, adx2);
				const float power = fmaf(-0.5f, s, -(bdx * dy));
				if (power > 0.0f) continue;
				if (CUTOFF && power < -4.5f) continue;
				const float G = bwd_exp(power);
				const float alpha = fminf(0.99f, g1.y * G);
				if (alpha < 1.0f / 255.0f) continue;
				T[k] = T[k] / (1.f - alpha);
				const float dchannel_dcolor = alpha * T[k];
				float dL_dalpha;
				{
					acc0[k] = lastA[k] * lc0[k] + (1.f - lastA[k]) * acc0[k]; lc0[k] = g1.z;
					acc1[k] = lastA[k] * lc1[k] + (1.f - lastA[k]) * acc1[k]; lc1[k] = g1.w;
					acc2[k] = lastA[k] * lc2[k] + (1.f - lastA[k]) * acc2[k]; lc2[k] = cb;
					dL_dalpha = (g1.z - acc0[k]) * dp0[k] + (g1.w - acc1[k]) * dp1[k] + (cb - acc2[k]) * dp2[k];
				}
				r_c0 += dchannel_dcolor * dp0[k];
				r_c1 += dchannel_dcolor * dp1[k];
				r_c2 += dchannel_dcolor * dp2[k];
				dL_dalpha *= T[k];
				lastA[k] = alpha;
				dL_dalpha += (-Tfin[k] / (1.f - alpha)) * bgdot[k];
				const float dL_dG = g1.y * dL_dalpha;
				const float gdx = G * dx, gdy = G * dy;
				const float dG_ddelx = -gdx * g0.z - gdy * g0.w;
				const float dG_ddely = -gdy * g1.x - gdx * g0.w;
				r_mx += dL_dG * dG_ddelx * ddelx_dx;
				r_my += dL_dG * dG_ddely * ddely_dy;
				r_ka += -0.5f * gdx * dx * dL_dG;
				r_kb += -0.5f * gdx * dy * dL_dG;
				r_kc += -0.5f * gdy * dy * dL_dG;
				r_op += G * dL_dalpha;
				any = true;
			}
			if (__any(any))
			{
				r_c0 = wave_sum_b(r_c0); r_c1 = wave_sum_b(r_c1); r_c2 = wave_sum_b(r_c2);
				r_mx = wave_sum_b(r_mx); r_my = wave_sum_b(r_my);
				r_ka = wave_sum_b(r_ka); r_kb = wave_sum_b(r_kb); r_kc = wave_sum_b(r_kc);
				r_op = wave_sum_b(r_op);
				if (lane == 63)
				{
					const int id = sid[j];
					atomicAdd(&a.dL_dcolor[3 * (size_t)id], r_c0);
					atomicAdd(&a.dL_dcolor[3 * (size_t)id + 1], r_c1);
					atomicAdd(&a.dL_dcolor[3 * (size_t)id + 2], r_c2);
					atomicAdd(&a.dL_dmean2D[3 * (size_t)id], r_mx);
					atomicAdd(&a.dL_dmean2D[3 * (size_t)id + 1], r_my);
					atomicAdd(&a.dL_dconic[4 * (size_t)id], r_ka);
					atomicAdd(&a.dL_dconic[4 * (size_t)id + 1], r_kb);
					atomicAdd(&a.dL_dconic[4 * (size_t)id + 3], r_kc);
					atomicAdd(&a.dL_dopacity[id], r_op);
				}
			}
		}
	}
}

// ---- per-Gaussian chain rule ----------------------------------------------------------------
struct M3b { float c[3][3]; };
__device__ __forceinline__ M3b mb_cols(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2)
{
	M3b m; m.c[0][0] = a0; m.c[0][1] = a1; m.c[0][2] = a2; m.c[1][0] = b0; m.c[1][1] = b1; m.c[1][2] = b2;
	m.c[2][0] = c0; m.c[2][1] = c1; m.c[2][2] = c2; return m;
}
__device__ __forceinline__ M3b mb_mul(const M3b &a, const M3b &b)
{
	M3b r;
#pragma unroll
	for (int col = 0; col < 3; col++)
#pragma unroll
		for (int row = 0; row < 3; row++)
			r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2];
	return r;
}
__device__ __forceinline__ M3b mb_t(const M3b &a)
{
	M3b r;
#pragma unroll
	for (int col = 0; col < 3; col++)
#pragma unroll
		for (int row = 0; row < 3; row++) r.c[col][row] = a.c[row][col];
	return r;
}

struct BwdPreArgs {
	int P, D, M, W, H;
	float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
	const float *means3D, *scales, *rotations, *shs, *shs_rest, *cov3D_precomp, *colors_precomp;
	const float *viewmatrix, *projmatrix, *campos;
	const int *radii;
	const float4 *rec;
	const float *cov3D_ws;
	const float *dL_dmean2D, *dL_dconic, *dL_dcolor;
	float *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dsh_rest, *dL_dscale, *dL_drot;
	const uint32_t *vis_list;  // forward's compact list of projected Gaussians
	const uint32_t *vis_count; // its length (device)
};

// row: the 64-byte row the forward pass left for this vis_list entry (xyz | raw scale | rotation | 3D covariance), or
// null when the covariances were an input: one coalesced row instead of gathers from four tensors.
__device__ __forceinline__ void preprocess_bwd_one(const BwdPreArgs &a, const int idx, const float4 *row)
{
	const float *vm = a.viewmatrix, *proj = a.projmatrix;
	float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
	if (row != nullptr) { r0 = row[0]; r1 = row[1]; r2 = row[2]; r3 = row[3]; }
	const float m[3] = { row ? r0.x : a.means3D[3 * idx], row ? r0.y : a.means3D[3 * idx + 1], row ? r0.z : a.means3D[3 * idx + 2] };
	const float cov_row[6] = { r2.z, r2.w, r3.x, r3.y, r3.z, r3.w };
	const float *cov3D = row ? cov_row : a.cov3D_precomp + 6 * (size_t)idx;
	const float fx = a.focal_x, fy = a.focal_y;
	float dmean[3];
	float dcov[6];
	// ---- 2D covariance backward: backward.cu:144-274 ----
	{
		const float dconic[3] = { a.dL_dconic[4 * (size_t)idx], a.dL_dconic[4 * (size_t)idx + 1], a.dL_dconic[4 * (size_t)idx + 3] };
		float t[3];
		t[0] = vm[0] * m[0] + vm[4] * m[1] + vm[8] * m[2] + vm[12];
		t[1] = vm[1] * m[0] + vm[5] * m[1] + vm[9] * m[2] + vm[13];
		t[2] = vm[2] * m[0] + vm[6] * m[1] + vm[10] * m[2] + vm[14];
		const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
		const float txtz = t[0] / t[2], tytz = t[1] / t[2];
		t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
		t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
		const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
		const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
		const M3b J = mb_cols(fx / t[2], 0, -(fx * t[0]) / (t[2] * t[2]), 0, fy / t[2], -(fy * t[1]) / (t[2] * t[2]), 0, 0, 0);
		const M3b Wm = mb_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
		const M3b Vrk = mb_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
		const M3b Tm = mb_mul(Wm, J);
		const M3b c2 = mb_mul(mb_mul(mb_t(Tm), mb_t(Vrk)), Tm);
		const float ca = c2.c[0][0] + 0.3f, cbb = c2.c[0][1], cc = c2.c[1][1] + 0.3f;
		const float denom = ca * cc - cbb * cbb;
		float dL_da = 0, dL_db = 0, dL_dc = 0;
		const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define TT(i, j) Tm.c[i][j]
#define VV(i, j) Vrk.c[i][j]
		if (denom2inv != 0)
		{
			dL_da = denom2inv * (-cc * cc * dconic[0] + 2 * cbb * cc * dconic[1] + (denom - ca * cc) * dconic[2]);
			dL_dc = denom2inv * (-ca * ca * dconic[2] + 2 * ca * cbb * dconic[1] + (denom - ca * cc) * dconic[0]);
			dL_db = denom2inv * 2 * (cbb * cc * dconic[0] - (denom + 2 * cbb * cbb) * dconic[1] + ca * cbb * dconic[2]);
			dcov[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
			dcov[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
			dcov[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
			dcov[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
			dcov[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
			dcov[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
		}
		else
		{
#pragma unroll
			for (int i = 0; i < 6; i++) dcov[i] = 0;
		}
		const float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
			(TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
		const float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
			(TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
		const float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
			(TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
		const float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
			(TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
		const float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
			(TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
		const float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
			(TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
#undef TT
#undef VV
		const float dL_dJ00 = Wm.c[0][0] * dL_dT00 + Wm.c[0][1] * dL_dT01 + Wm.c[0][2] * dL_dT02;
		const float dL_dJ02 = Wm.c[2][0] * dL_dT00 + Wm.c[2][1] * dL_dT01 + Wm.c[2][2] * dL_dT02;
		const float dL_dJ11 = Wm.c[1][0] * dL_dT10 + Wm.c[1][1] * dL_dT11 + Wm.c[1][2] * dL_dT12;
		const float dL_dJ12 = Wm.c[2][0] * dL_dT10 + Wm.c[2][1] * dL_dT11 + Wm.c[2][2] * dL_dT12;
		const float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
		const float dL_dtx = x_grad_mul * -fx * tz2 * dL_dJ02;
		const float dL_dty = y_grad_mul * -fy * tz2 * dL_dJ12;
		const float dL_dtz = -fx * tz2 * dL_dJ00 - fy * tz2 * dL_dJ11 + (2 * fx * t[0]) * tz3 * dL_dJ02 + (2 * fy * t[1]) * tz3 * dL_dJ12;
		dmean[0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
		dmean[1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
		dmean[2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;
#pragma unroll
		for (int i = 0; i < 6; i++) if (a.dL_dcov3D != nullptr) a.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
	}
	// ---- projection backward: backward.cu:370-387 ----
	{
		const float hw = proj[3] * m[0] + proj[7] * m[1] + proj[11] * m[2] + proj[15];
		const float m_w = 1.0f / (hw + 0.0000001f);
		const float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
		const float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
		const float d2x = a.dL_dmean2D[3 * (size_t)idx], d2y = a.dL_dmean2D[3 * (size_t)idx + 1];
		dmean[0] += (proj[0] * m_w - proj[3] * mul1) * d2x + (proj[1] * m_w - proj[3] * mul2) * d2y;
		dmean[1] += (proj[4] * m_w - proj[7] * mul1) * d2x + (proj[5] * m_w - proj[7] * mul2) * d2y;
		dmean[2] += (proj[8] * m_w - proj[11] * mul1) * d2x + (proj[9] * m_w - proj[11] * mul2) * d2y;
	}
	// ---- SH backward: backward.cu:20-139 ----
	if (a.colors_precomp == nullptr && a.shs != nullptr)
	{
		// Coefficients and their gradients are kept in registers in the concatenated order (k, channel) -> 3k + ch
		// and moved 16 bytes at a time: every lane works on its own Gaussian, i.e. its own cache lines, and 48
		// dword loads + 48 dword stores per Gaussian kept the address unit busy for most of this kernel.
		// Split storage (shs = DC [P,1,3], shs_rest = [P,M-1,3]): slots 3.. come from / go to the rest tensors.
		const bool split = a.shs_rest != nullptr;
		const int nrest = split ? (a.M - 1) * 3 : a.M * 3 - 3;     // floats available after the DC triple
		const float *sh_r = split ? a.shs_rest + (size_t)idx * (a.M - 1) * 3 : a.shs + (size_t)idx * a.M * 3 + 3;
		float *dsh_r = split ? a.dL_dsh_rest + (size_t)idx * (a.M - 1) * 3 : a.dL_dsh + (size_t)idx * a.M * 3 + 3;
		float *dsh0 = split ? a.dL_dsh + 3 * (size_t)idx : a.dL_dsh + (size_t)idx * a.M * 3;
		typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
		float shv[48], g[48];
#pragma unroll
		for (int i = 0; i < 48; i++) { shv[i] = 0.0f; g[i] = 0.0f; }
		const int nuse = 3 * ((a.D + 1) * (a.D + 1)) - 3; // rest floats the active degree reads / writes
		if (nrest >= 45)
		{
#pragma unroll
			for (int q = 0; q < 11; q++) { const f4u v = *(const f4u *)(sh_r + 4 * q); shv[3 + 4 * q] = v.x; shv[4 + 4 * q] = v.y; shv[5 + 4 * q] = v.z; shv[6 + 4 * q] = v.w; }
			shv[47] = sh_r[44];
		}
		else
		{
#pragma unroll
			for (int i = 0; i < 45; i++) if (i < nuse) shv[3 + i] = sh_r[i];
		}
		const uint32_t clamp_bits = __float_as_uint(a.rec[3 * (size_t)idx + 2].z);
		const float dox = m[0] - a.campos[0], doy = m[1] - a.campos[1], doz = m[2] - a.campos[2];
		const float len = sqrtf(dox * dox + doy * doy + doz * doz);
		const float x = dox / len, y = doy / len, z = doz / len;
		float dRGB[3];
#pragma unroll
		for (int ch = 0; ch < 3; ch++) dRGB[ch] = a.dL_dcolor[3 * (size_t)idx + ch] * (((clamp_bits >> ch) & 1u) ? 0.f : 1.f);
		float ddx[3] = { 0, 0, 0 }, ddy[3] = { 0, 0, 0 }, ddz[3] = { 0, 0, 0 };
		const int deg = a.D;
#define SHV(k, ch) shv[3 * (k) + (ch)]
#define DSH(k, w) { const float w_ = (w); g[3 * (k)] = w_ * dRGB[0]; g[3 * (k) + 1] = w_ * dRGB[1]; g[3 * (k) + 2] = w_ * dRGB[2]; }
		dsh0[0] = FR_SH_C0 * dRGB[0]; dsh0[1] = FR_SH_C0 * dRGB[1]; dsh0[2] = FR_SH_C0 * dRGB[2];
		if (deg > 0)
		{
			DSH(1, -FR_SH_C1 * y); DSH(2, FR_SH_C1 * z); DSH(3, -FR_SH_C1 * x);
#pragma unroll
			for (int ch = 0; ch < 3; ch++)
			{
				ddx[ch] = -FR_SH_C1 * SHV(3, ch);
				ddy[ch] = -FR_SH_C1 * SHV(1, ch);
				ddz[ch] = FR_SH_C1 * SHV(2, ch);
			}
			if (deg > 1)
			{
				const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
				DSH(4, FR_SH_C2_0 * xy); DSH(5, FR_SH_C2_1 * yz); DSH(6, FR_SH_C2_2 * (2.f * zz - xx - yy));
				DSH(7, FR_SH_C2_3 * xz); DSH(8, FR_SH_C2_4 * (xx - yy));
#pragma unroll
				for (int ch = 0; ch < 3; ch++)
				{
					ddx[ch] += FR_SH_C2_0 * y * SHV(4, ch) + FR_SH_C2_2 * 2.f * -x * SHV(6, ch) + FR_SH_C2_3 * z * SHV(7, ch) + FR_SH_C2_4 * 2.f * x * SHV(8, ch);
					ddy[ch] += FR_SH_C2_0 * x * SHV(4, ch) + FR_SH_C2_1 * z * SHV(5, ch) + FR_SH_C2_2 * 2.f * -y * SHV(6, ch) + FR_SH_C2_4 * 2.f * -y * SHV(8, ch);
					ddz[ch] += FR_SH_C2_1 * y * SHV(5, ch) + FR_SH_C2_2 * 2.f * 2.f * z * SHV(6, ch) + FR_SH_C2_3 * x * SHV(7, ch);
				}
				if (deg > 2)
				{
					DSH(9, FR_SH_C3_0 * y * (3.f * xx - yy));
					DSH(10, FR_SH_C3_1 * xy * z);
					DSH(11, FR_SH_C3_2 * y * (4.f * zz - xx - yy));
					DSH(12, FR_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy));
					DSH(13, FR_SH_C3_4 * x * (4.f * zz - xx - yy));
					DSH(14, FR_SH_C3_5 * z * (xx - yy));
					DSH(15, FR_SH_C3_6 * x * (xx - 3.f * yy));
#pragma unroll
					for (int ch = 0; ch < 3; ch++)
					{
						ddx[ch] += (
							FR_SH_C3_0 * SHV(9, ch) * 3.f * 2.f * xy +
							FR_SH_C3_1 * SHV(10, ch) * yz +
							FR_SH_C3_2 * SHV(11, ch) * -2.f * xy +
							FR_SH_C3_3 * SHV(12, ch) * -3.f * 2.f * xz +
							FR_SH_C3_4 * SHV(13, ch) * (-3.f * xx + 4.f * zz - yy) +
							FR_SH_C3_5 * SHV(14, ch) * 2.f * xz +
							FR_SH_C3_6 * SHV(15, ch) * 3.f * (xx - yy));
						ddy[ch] += (
							FR_SH_C3_0 * SHV(9, ch) * 3.f * (xx - yy) +
							FR_SH_C3_1 * SHV(10, ch) * xz +
							FR_SH_C3_2 * SHV(11, ch) * (-3.f * yy + 4.f * zz - xx) +
							FR_SH_C3_3 * SHV(12, ch) * -3.f * 2.f * yz +
							FR_SH_C3_4 * SHV(13, ch) * -2.f * xy +
							FR_SH_C3_5 * SHV(14, ch) * -2.f * yz +
							FR_SH_C3_6 * SHV(15, ch) * -3.f * 2.f * xy);
						ddz[ch] += (
							FR_SH_C3_1 * SHV(10, ch) * xy +
							FR_SH_C3_2 * SHV(11, ch) * 4.f * 2.f * yz +
							FR_SH_C3_3 * SHV(12, ch) * 3.f * (2.f * zz - xx - yy) +
							FR_SH_C3_4 * SHV(13, ch) * 4.f * 2.f * xz +
							FR_SH_C3_5 * SHV(14, ch) * (xx - yy));
					}
				}
			}
		}
#undef SHV
#undef DSH
		// gradients of the rest coefficients of the active degree (the others stay at the caller's zero fill)
#pragma unroll
		for (int q = 0; q < 12; q++)
		{
			if (4 * q + 4 <= nuse) *(f4u *)(dsh_r + 4 * q) = (f4u){ g[3 + 4 * q], g[4 + 4 * q], g[5 + 4 * q], g[6 + 4 * q] };
			else
			{
#pragma unroll
				for (int j = 0; j < 4; j++) if (4 * q + j < nuse) dsh_r[4 * q + j] = g[3 + 4 * q + j];
			}
		}
		const float dvx = ddx[0] * dRGB[0] + ddx[1] * dRGB[1] + ddx[2] * dRGB[2];
		const float dvy = ddy[0] * dRGB[0] + ddy[1] * dRGB[1] + ddy[2] * dRGB[2];
		const float dvz = ddz[0] * dRGB[0] + ddz[1] * dRGB[1] + ddz[2] * dRGB[2];
		// dnormvdv: auxiliary.h:107-117
		const float sum2 = dox * dox + doy * doy + doz * doz;
		const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
		dmean[0] += ((+sum2 - dox * dox) * dvx - doy * dox * dvy - doz * dox * dvz) * invsum32;
		dmean[1] += (-dox * doy * dvx + (sum2 - doy * doy) * dvy - doz * doy * dvz) * invsum32;
		dmean[2] += (-dox * doz * dvx - doy * doz * dvy + (sum2 - doz * doz) * dvz) * invsum32;
	}
	a.dL_dmean3D[3 * (size_t)idx] = dmean[0];
	a.dL_dmean3D[3 * (size_t)idx + 1] = dmean[1];
	a.dL_dmean3D[3 * (size_t)idx + 2] = dmean[2];
	// ---- 3D covariance backward: backward.cu:278-341 ----
	if (a.cov3D_precomp == nullptr && a.scales != nullptr)
	{
		const float4 q = row ? make_float4(r1.z, r1.w, r2.x, r2.y) : ((const float4 *)a.rotations)[idx];
		const float r = q.x, x = q.y, y = q.z, z = q.w;
		const M3b R = mb_cols(
			1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
			2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
			2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
		const float s[3] = { a.scale_modifier * (row ? r0.w : a.scales[3 * idx]), a.scale_modifier * (row ? r1.x : a.scales[3 * idx + 1]),
			a.scale_modifier * (row ? r1.y : a.scales[3 * idx + 2]) };
		const M3b S = mb_cols(s[0], 0, 0, 0, s[1], 0, 0, 0, s[2]);
		M3b Mm = mb_mul(S, R);
		const M3b dSigma = mb_cols(
			dcov[0], 0.5f * dcov[1], 0.5f * dcov[2],
			0.5f * dcov[1], dcov[3], 0.5f * dcov[4],
			0.5f * dcov[2], 0.5f * dcov[4], dcov[5]);
#pragma unroll
		for (int i = 0; i < 3; i++)
#pragma unroll
			for (int j = 0; j < 3; j++) Mm.c[i][j] = 2.0f * Mm.c[i][j];
		const M3b dM = mb_mul(Mm, dSigma);
		const M3b Rt = mb_t(R);
		M3b dMt = mb_t(dM);
		float ds[3];
#pragma unroll
		for (int i = 0; i < 3; i++) ds[i] = Rt.c[i][0] * dMt.c[i][0] + Rt.c[i][1] * dMt.c[i][1] + Rt.c[i][2] * dMt.c[i][2];
#pragma unroll
		for (int i = 0; i < 3; i++)
#pragma unroll
			for (int j = 0; j < 3; j++) dMt.c[i][j] *= s[i];
#define Dm(i, j) dMt.c[i][j]
		float4 dq;
		dq.x = 2 * z * (Dm(0, 1) - Dm(1, 0)) + 2 * y * (Dm(2, 0) - Dm(0, 2)) + 2 * x * (Dm(1, 2) - Dm(2, 1));
		dq.y = 2 * y * (Dm(1, 0) + Dm(0, 1)) + 2 * z * (Dm(2, 0) + Dm(0, 2)) + 2 * r * (Dm(1, 2) - Dm(2, 1)) - 4 * x * (Dm(2, 2) + Dm(1, 1));
		dq.z = 2 * x * (Dm(1, 0) + Dm(0, 1)) + 2 * r * (Dm(2, 0) - Dm(0, 2)) + 2 * z * (Dm(1, 2) + Dm(2, 1)) - 4 * y * (Dm(2, 2) + Dm(0, 0));
		dq.w = 2 * r * (Dm(0, 1) - Dm(1, 0)) + 2 * x * (Dm(2, 0) + Dm(0, 2)) + 2 * y * (Dm(1, 2) + Dm(2, 1)) - 4 * z * (Dm(1, 1) + Dm(0, 0));
#undef Dm
		a.dL_dscale[3 * (size_t)idx] = ds[0]; a.dL_dscale[3 * (size_t)idx + 1] = ds[1]; a.dL_dscale[3 * (size_t)idx + 2] = ds[2];
		((float4 *)a.dL_drot)[idx] = dq;
	}
}

// Grid-stride over the forward pass's visible list: dense waves instead of one thread per Gaussian with
// ~90 % of the lanes returning immediately. Entries culled after projection (radii reset to 0) are skipped.
__global__ void __launch_bounds__(256) k_preprocess_bwd(const BwdPreArgs a)
{
	const int V = (int)*a.vis_count;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x)
	{
		const int idx = (int)a.vis_list[i];
		if (a.radii[idx] > 0) preprocess_bwd_one(a, idx, a.cov3D_precomp ? nullptr : (const float4 *)a.cov3D_ws + 4 * (size_t)i);
	}
}

#ifndef FR_BWD_PPL
#define FR_BWD_PPL 4
#endif

int launch_backward(const fr_backward_args *a)
{
	hipStream_t stream = (hipStream_t)a->stream;
	const int gx = (a->W + FR_TILE - 1) / FR_TILE, gy = (a->H + FR_TILE - 1) / FR_TILE, T = gx * gy;
	GeomWS geom = carve_geom(a->variant, (size_t)a->P, (char *)a->geometry);
	ImageWS img = carve_image(a->variant, a->W, a->H, (char *)a->image);
	BinWS bin = carve_bin(a->R, (char *)a->binning);
	auto mark = [&](int i) { if (a->stage_events && a->stage_events[i]) (void)hipEventRecord((hipEvent_t)a->stage_events[i], stream); };
	mark(0);
	if (a->R > 0)
	{
		BwdRenderArgs r;
		r.W = a->W; r.H = a->H; r.gx = gx; r.ranges = img.ranges; r.tile_order = img.tile_order; r.point_list = bin.point_list; r.rec = geom.rec;
		r.bg = a->background; r.final_T = img.final_T; r.n_contrib = img.n_contrib; r.dL_dpix = a->dL_dpix;
		r.dL_dmean2D = a->dL_dmean2D; r.dL_dconic = a->dL_dconic; r.dL_dopacity = a->dL_dopacity; r.dL_dcolor = a->dL_dcolor;
		constexpr int PPL = FR_BWD_PPL;
		if (a->variant == FR_VARIANT_ORIGINAL)
			hipLaunchKernelGGL((k_render_bwd<false, PPL>), dim3(T), dim3(256 / PPL), 0, stream, r);
		else
			hipLaunchKernelGGL((k_render_bwd<true, PPL>), dim3(T), dim3(256 / PPL), 0, stream, r);
		int rc = check_launch("render_bwd", stream, a->debug);
		if (rc) return rc;
	}
	mark(1);
	BwdPreArgs p;
	p.P = a->P; p.D = a->D; p.M = a->M; p.W = a->W; p.H = a->H;
	p.tanfovx = a->tanfovx; p.tanfovy = a->tanfovy;
	p.focal_y = a->H / (2.0f * a->tanfovy); p.focal_x = a->W / (2.0f * a->tanfovx);
	p.scale_modifier = a->scale_modifier;
	p.means3D = a->means3D; p.scales = a->scales; p.rotations = a->rotations; p.shs = a->shs; p.shs_rest = a->shs_rest;
	p.cov3D_precomp = a->cov3D_precomp; p.colors_precomp = a->colors_precomp;
	p.viewmatrix = a->viewmatrix; p.projmatrix = a->projmatrix; p.campos = a->campos;
	p.radii = a->radii; p.rec = geom.rec; p.cov3D_ws = geom.cov3D;
	p.dL_dmean2D = a->dL_dmean2D; p.dL_dconic = a->dL_dconic; p.dL_dcolor = a->dL_dcolor;
	p.dL_dmean3D = a->dL_dmean3D; p.dL_dcov3D = a->dL_dcov3D; p.dL_dsh = a->dL_dsh; p.dL_dsh_rest = a->dL_dsh_rest; p.dL_dscale = a->dL_dscale; p.dL_drot = a->dL_drot;
	p.vis_list = geom.vis_list; p.vis_count = geom.slab_ctr + 1;
	const int pblocks = (a->P + 255) / 256;
	hipLaunchKernelGGL(k_preprocess_bwd, dim3(pblocks < 2048 ? pblocks : 2048), dim3(256), 0, stream, p);
	int rc2 = check_launch("preprocess_bwd", stream, a->debug);
	mark(2);
	return rc2;
}

} // namespace fr
