// Per-Gaussian stages: projection / 2D covariance / SH colour, fused tile counting with the
// oriented-box and foveal-level culls, and instance emission into per-tile buckets.
//
// Replaces (reference, paths under fov3dgs/submodules/):
//   preprocessCUDA           diff-gaussian-rasterization/cuda_rasterizer/forward.cu:155-262
//                            …_pcheck_obb_sum/cuda_rasterizer/forward.cu:155-293, …_fov_pcheck_obb/…/forward.cu:105-238
//   OBB_test / filter        …_pcheck_obb_sum/cuda_rasterizer/rasterizer_impl.cu:70-146, …_fov_pcheck_obb/…:264-383
//   duplicateWithKeys        …/rasterizer_impl.cu:70-111 (R0), :150-214 (RS), :423-486 (RF)
//   compute_tile_levels_cuda / compute_tile_level_infos_cuda   …_fov_pcheck_obb/…/rasterizer_impl.cu:120-260
//   compute_fov_colors       …_fov_pcheck_obb/…/rasterizer_impl.cu:490-530
//   checkFrustum             …/rasterizer_impl.cu:54-66
//
// MI355X design: the reference materialises a per-(Gaussian,tile) bool bitmap between a cull
// kernel and the key-emission kernel and needs two P-long prefix scans plus two host syncs.
// Here the cull test is evaluated inside preprocess (count phase, bumping one counter per TILE)
// and re-evaluated in emit (no bitmap, no per-Gaussian scan); instances are bucketed by tile
// at emission time through per-tile cursors, so no global 64-bit radix sort is needed.
#include "common.h"
#include "tile_scan.h"
#include <cstdlib>

namespace fr {

// ---- glm-like column-major 3x3 helper; the summation order of mul() is the contract ----------
struct M3 { float c[3][3]; };
__device__ __forceinline__ M3 m3_cols(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2)
{
	M3 m; m.c[0][0] = a0; m.c[0][1] = a1; m.c[0][2] = a2; m.c[1][0] = b0; m.c[1][1] = b1; m.c[1][2] = b2;
	m.c[2][0] = c0; m.c[2][1] = c1; m.c[2][2] = c2; return m;
}
__device__ __forceinline__ M3 m3_mul(const M3 &a, const M3 &b)
{
	M3 r;
#pragma unroll
	for (int col = 0; col < 3; col++)
#pragma unroll
		for (int row = 0; row < 3; row++)
			r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2];
	return r;
}
__device__ __forceinline__ M3 m3_t(const M3 &a)
{
	M3 r;
#pragma unroll
	for (int col = 0; col < 3; col++)
#pragma unroll
		for (int row = 0; row < 3; row++) r.c[col][row] = a.c[row][col];
	return r;
}

// ---- RF eccentricity level of a tile: RF rasterizer_impl.cu:86-177, auxiliary.h:55-66 ----------
__device__ __forceinline__ void ncd2dir(float nx, float ny, float rw, float rh, float out[3])
{
	const float vx = (nx - 0.5f) * rw, vy = (ny - 0.5f) * rh, vz = 1.0f;
	const float d = sqrtf(vx * vx + vy * vy + vz * vz);
	out[0] = vx / d; out[1] = vy / d; out[2] = vz / d;
}
__device__ float tile_level(int tx, int ty, int W, int H, float gaze_x, float gaze_y, float alpha)
{
	const float riw = 2.0f, rvd = 1.0f, sqrt_max_ps = 3.4641016151377544f;
	const float step = (float)(((double)sqrt_max_ps - 1.) / (double)(float)(FR_FOV_LEVELS - 1));
	const float px = (float)(tx * FR_TILE + FR_TILE / 2), py = (float)(ty * FR_TILE + FR_TILE / 2);
	const float rih = (float)H / (float)W * riw;
	const float nx = px / W, ny = py / H;
	float tdir[3], gdir[3], cdir[3];
	ncd2dir(nx, ny, riw, rih, tdir);
	ncd2dir(gaze_x, gaze_y, riw, rih, gdir);
	ncd2dir(0.5f, 0.5f, riw, rih, cdir);
	const float ecc = acosf(gdir[0] * tdir[0] + gdir[1] * tdir[1] + gdir[2] * tdir[2]);
	const float ecc_c = acosf(tdir[0] * cdir[0] + tdir[1] * cdir[1] + tdir[2] * cdir[2]);
	const float pool = alpha * ecc * ecc;
	const float amin = (float)((double)ecc_c - (double)pool * 0.5);
	const float amax = (float)((double)ecc_c + (double)pool * 0.5);
	const float ax = (float)(((double)nx - 0.5) * (double)riw), ay = (float)(((double)ny - 0.5) * (double)rih);
	const float dist = sqrtf(ax * ax + ay * ay + rvd * rvd);
	const float major = (tanf(amax) - tanf(amin)) * rvd;
	const float minor = 2.0f * dist * tanf(pool * 0.5f);
	const float area = (float)(3.14159265358979323846 * (double)major * (double)minor * (double)0.25f);
	const float r2p = W / riw;
	const float ps = sqrtf(area) * r2p;
	float level;
	if (ps <= 1) level = 0; else level = (sqrtf(ps) - 1) / step;
	if ((double)level > ((double)(float)FR_FOV_LEVELS - 0.1)) level = (float)((double)(float)FR_FOV_LEVELS - 0.1);
	return level;
}

// One thread per tile: level of the tile and of its 4 neighbours, finite-difference gradients, conservative tile minimum and the
// two-level-blend flag. out = float[5][T]: level, tile_min, grad_x, grad_y, blending.
// lv_bbox[k] (zeroed by the caller): bounding box of the tiles with tile_min < k, see walk_rect().
// A workgroup owns a 14 x 14 patch of tiles: every level is evaluated once (plus the patch's one-tile halo: 16 x 16 threads,
// one level each) and shared through LDS -- the level function (acos, tan, three sqrt) is ~500 instructions.
// mmfr_level >= 0: the multi-model baseline's map for that level (mmfr rasterizer_impl.cu:246-262,277-304): tile_min is
// clamped at 0 and kept in row 0 (the blend kernel's value); row 1, which the level filter of the binning kernels and
// the level boxes read, becomes 0 for the tiles this level renders (tile_min in (level - 0.5, level + 1)) and 5 for the
// skipped ones -- with highest_levels == 0 the filter `row 1 < highest_level + 1` is then exactly "not skipped".
#define FR_LV_PATCH 14 // tiles per side a workgroup of k_tile_levels owns (+ a one-tile halo = its 16 x 16 threads)
__global__ void __launch_bounds__(256) k_tile_levels(int T, int gx, int gy, int W, int H, float gaze_x, float gaze_y, float alpha, float *out,
	uint32_t *lv_bbox, uint32_t *slab_ctr, float mmfr_level)
{
	// the counters of the two kernels that follow are cleared here (one fill command less at the head of the frame)
	if (blockIdx.x == 0)
		for (int i = threadIdx.x; i < FR_SLAB_CTR_WORDS; i += 256) slab_ctr[i] = 0;
	// A workgroup's 16 x 16 threads evaluate ONE level each: the 14 x 14 tiles it owns and their one-tile halo (the level
	// function is a ~500-instruction dependent chain: a 16 x 16 patch whose first 64 threads also did the halo had a critical
	// path of two of them).
	__shared__ float s_lv[16][16]; // [y][x] of the haloed patch; -1 = outside the grid
	const int pxn = (gx + FR_LV_PATCH - 1) / FR_LV_PATCH;
	const int bx = (int)blockIdx.x % pxn, by = (int)blockIdx.x / pxn;
	const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
	const int tx = bx * FR_LV_PATCH + lx - 1, ty = by * FR_LV_PATCH + ly - 1;
	const bool in_grid = tx >= 0 && tx < gx && ty >= 0 && ty < gy;
	const bool live = in_grid && lx >= 1 && lx <= FR_LV_PATCH && ly >= 1 && ly <= FR_LV_PATCH; // a tile this workgroup owns
	const int idx = ty * gx + tx;
	s_lv[ly][lx] = in_grid ? tile_level(tx, ty, W, H, gaze_x, gaze_y, alpha) : -1.0f;
	__syncthreads();
	const float lf = s_lv[ly][lx];
	const int xl = max(lx - 1, 0), xr = min(lx + 1, 15), yd = max(ly - 1, 0), yu = min(ly + 1, 15); // (only owned tiles use them)
	const float right = s_lv[ly][xr], left = s_lv[ly][xl], up = s_lv[yu][lx], down = s_lv[yd][lx];
	float gxv = 0, gyv = 0;
	if (right != -1 && left != -1) gxv = (right - left) / 2.0f;
	else if (right != -1) gxv = right - lf;
	else if (left != -1) gxv = lf - left;
	if (up != -1 && down != -1) gyv = (up - down) / 2.0f;
	else if (up != -1) gyv = up - lf;
	else if (down != -1) gyv = lf - down;
	const float max_delta = (float)(0.5 * (double)(fabsf(gxv) + fabsf(gyv)));
	const bool mmfr = mmfr_level >= 0.0f;
	float tmin = lf - max_delta;
	if (mmfr && tmin < 0.0f) tmin = 0.0f;
	const float tmin_i = (float)f2i(tmin);
	const bool blending = ((tmin - tmin_i) > 0.5f) && (tmin_i < (float)(FR_FOV_LEVELS - 1));
	const float real_tmin = tmin;
	if (mmfr) tmin = (real_tmin > mmfr_level - 0.5f && real_tmin < mmfr_level + 1.0f) ? 0.0f : 5.0f; // the filter key
	if (live)
	{
		out[idx] = mmfr ? real_tmin : lf;
		out[T + idx] = tmin;
		out[2 * T + idx] = gxv;
		out[3 * T + idx] = gyv;
		out[4 * T + idx] = blending ? 1.0f : 0.0f;
	}
	// bounding boxes: wave reduction -> LDS -> one global atomic per workgroup and component (a cache line
	// takes only ~90 atomics/us, so every box has its own line: FR_LV_BBOX_STRIDE)
	__shared__ uint32_t s_bb[(FR_FOV_LEVELS + 1) * 4];
	if (threadIdx.x < (FR_FOV_LEVELS + 1) * 4) s_bb[threadIdx.x] = 0;
	__syncthreads();
	for (int k = 0; k <= FR_FOV_LEVELS; k++)
	{
		const bool in = live && tmin < (float)k;
		uint32_t v[4] = { in ? (uint32_t)(gx - tx) : 0u, in ? (uint32_t)(gy - ty) : 0u, in ? (uint32_t)(tx + 1) : 0u, in ? (uint32_t)(ty + 1) : 0u };
#pragma unroll
		for (int c = 0; c < 4; c++)
		{
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) v[c] = max(v[c], (uint32_t)__shfl_xor((int)v[c], off));
			if ((threadIdx.x & 63) == 0 && v[c] != 0) atomicMax(&s_bb[4 * k + c], v[c]);
		}
	}
	__syncthreads();
	if (threadIdx.x < (FR_FOV_LEVELS + 1) * 4 && s_bb[threadIdx.x] != 0)
		atomicMax(&lv_bbox[(threadIdx.x >> 2) * FR_LV_BBOX_STRIDE + (threadIdx.x & 3)], s_bb[threadIdx.x]);
}

// ---- SH colour: forward.cu:20-71 (full) and RF rasterizer_impl.cu:37-84 (rest only) -----------
// sh points at the navail = 3*M floats of this Gaussian; REST: coefficient k is stored at slot k-1 and coefficient 0
// comes from dc[3] if that is given (split storage: features_dc / features_rest), else it is left out (RF).
// Every lane reads its own Gaussian, i.e. its own cache lines, and a CU's address unit handles such a
// scattered access lane by lane: 48 dword loads per Gaussian made this the slowest part of k_bin. The
// coefficients are therefore fetched 16 bytes at a time (dword-aligned: 45 floats per Gaussian in RF) and
// folded into the three channel sums in the reference's order (term k is basis_k * coefficient, summed
// k = 0..15 left to right), so the result is bit-identical to the scalar formulation.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
// the 45 / 48 coefficient floats of one Gaussian as twelve 16-byte loads (usual case: M = 16 coefficients allocated).
// Fetched unconditionally, so the loads are issued back to back and cost ONE memory round trip; k_bin issues them at
// the head of a slab and evaluates the colour after the tile walk (sh_eval), so that the round trip -- and the lines
// it pulls, the largest share of the kernel's traffic -- overlaps the projection and the walk instead of following them.
struct ShRows { f4u v[12]; };
template <bool REST>
__device__ __forceinline__ void sh_fetch(const float *__restrict__ sh, ShRows &r)
{
#pragma unroll
	for (int q = 0; q < 11; q++) r.v[q] = *(const f4u *)(sh + 4 * q);
	if (REST) r.v[11] = (f4u){ sh[44], 0.0f, 0.0f, 0.0f };
	else r.v[11] = *(const f4u *)(sh + 44);
}
__device__ __forceinline__ void sh_basis(float dx, float dy, float dz, float basis[16])
{
	const float len = sqrtf(dx * dx + dy * dy + dz * dz);
	const float x = dx / len, y = dy / len, z = dz / len;
	const float xx = x * x, yy = y * y, zz = z * z;
	const float xy = x * y, yz = y * z, xz = x * z;
	// basis_k with the association of the reference's expressions; terms 1 and 3 are subtracted there
	const float b[16] = {
		FR_SH_C0, -(FR_SH_C1 * y), FR_SH_C1 * z, -(FR_SH_C1 * x),
		FR_SH_C2_0 * xy, FR_SH_C2_1 * yz, FR_SH_C2_2 * (2.0f * zz - xx - yy), FR_SH_C2_3 * xz, FR_SH_C2_4 * (xx - yy),
		FR_SH_C3_0 * y * (3.0f * xx - yy), FR_SH_C3_1 * xy * z, FR_SH_C3_2 * y * (4.0f * zz - xx - yy),
		FR_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), FR_SH_C3_4 * x * (4.0f * zz - xx - yy),
		FR_SH_C3_5 * z * (xx - yy), FR_SH_C3_6 * x * (xx - 3.0f * yy) };
#pragma unroll
	for (int k = 0; k < 16; k++) basis[k] = b[k];
}
// colour from fetched rows; dc: coefficient 0 of a REST layout (split storage / packed rows), or null (RF: left out)
template <bool REST>
__device__ __forceinline__ void sh_eval(int deg, const ShRows &r, const float *dc, float dx, float dy, float dz, float out[3])
{
	float basis[16];
	sh_basis(dx, dy, dz, basis);
	constexpr int SKIP = REST ? 1 : 0;
	const int nfl = 3 * ((deg + 1) * (deg + 1) - SKIP); // floats in use
	float acc[3] = { 0.0f, 0.0f, 0.0f };
	if (REST && dc != nullptr)
	{
#pragma unroll
		for (int ch = 0; ch < 3; ch++) acc[ch] = basis[0] * dc[ch]; // == 0 + C0 * sh[0] of the unsplit sum
	}
#pragma unroll
	for (int i = 0; i < 48 - 3 * SKIP; i++)
		if (i < nfl) acc[i % 3] = acc[i % 3] + basis[i / 3 + SKIP] * r.v[i / 4][i % 4];
#pragma unroll
	for (int ch = 0; ch < 3; ch++) out[ch] = acc[ch] + 0.5f;
}
template <bool REST>
__device__ __forceinline__ void sh_colour(int deg, int navail, const float *__restrict__ sh, const float *__restrict__ dc,
	float dx, float dy, float dz, float out[3])
{
	constexpr int SKIP = REST ? 1 : 0;
	if (navail >= 48 - 3 * SKIP)
	{
		ShRows r;
		sh_fetch<REST>(sh, r);
		sh_eval<REST>(deg, r, dc, dx, dy, dz, out);
		return;
	}
	// fewer than 16 coefficients allocated: only what the active degree uses is read
	float basis[16];
	sh_basis(dx, dy, dz, basis);
	const int nfl = 3 * ((deg + 1) * (deg + 1) - SKIP); // floats in use
	float acc[3] = { 0.0f, 0.0f, 0.0f };
	if (REST && dc != nullptr)
	{
#pragma unroll
		for (int ch = 0; ch < 3; ch++) acc[ch] = basis[0] * dc[ch];
	}
#pragma unroll
	for (int q = 0; q < 12; q++)
	{
		if (4 * q + 4 <= nfl)
		{
			const f4u v = *(const f4u *)(sh + 4 * q);
#pragma unroll
			for (int j = 0; j < 4; j++) { const int i = 4 * q + j; acc[i % 3] = acc[i % 3] + basis[i / 3 + SKIP] * v[j]; }
		}
		else
		{
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				const int i = 4 * q + j;
				if (i < nfl) acc[i % 3] = acc[i % 3] + basis[i / 3 + SKIP] * sh[i];
			}
		}
	}
#pragma unroll
	for (int ch = 0; ch < 3; ch++) out[ch] = acc[ch] + 0.5f;
}

typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float *p)
{
	const nt_f4 v = __builtin_nontemporal_load((const nt_f4 *)p);
	return make_float4(v.x, v.y, v.z, v.w);
}
// raw per-Gaussian inputs of the projection
struct RawGaussian { float p[3], sc[3]; float4 q; float hl; };

#define FR_ER_CHUNK 512 // region-major emission: list entries of a region a workgroup of k_emit_regions takes at a time (two slabs of 64 per wave)
struct PreArgs {
	int P, D, M, W, H, gx, gy;
	float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
	const float *means3D, *scales, *rotations, *opacities, *shs, *cov3D_precomp, *colors_precomp;
	const float *shs_rest; // split SH storage: shs = DC [P,1,3], shs_rest = [P,M-1,3]; else null
	const float *viewmatrix, *projmatrix, *campos;
	const float *shs_dcs, *highest_levels;
	const float *packed_geom, *packed_colour, *packed_cull; // optional packed copies of the model (fovraster.h), else null
	const float *tile_lv; // RF float[5][T]
	const uint32_t *lv_bbox; // RF [5][FR_LV_BBOX_STRIDE], see walk_rect()
	int lds_tiles;           // RF: tile_min and the blend flags are staged in LDS (see k_bin)
	int T;
	int *radii;
	GeomWS geom;
	uint32_t *tile_count;
	uint32_t *hist; // [blocks][T] per-workgroup tile histograms (LDSH)
	int write_cov3D; // keep the 3D covariances for the backward pass (training variants only)
	int raw;         // scales / rotations / opacities are raw parameters: activate on the fly (fr_forward_args.raw_activations)
	int prefiltered; // fr_forward_args.prefiltered: a Gaussian behind the near plane is an error (slab_ctr[0] reports it)
	int proj_waves, proj_cpw; // the cull pass's grid in waves and the chunks each of them took (k_bin finds the regions from them)
	int wbase_lds;            // k_bin keeps its copy of the cull pass's running counts in LDS (launch_bin: when it fits)
	int fuse_scan;            // k_bin: the LAST workgroup to finish runs the tile scan (ts) as the kernel's tail -- no k_tile_scan launch
	TileScanArgs ts;
	int regions, region_rx;   // region-major emission: regions of the tile grid (0 = off) and regions per row (GeomWS::rtab / wlist)
	int wcap;                 // ... entries of a workgroup's segment of wlist
};

// Projection of one Gaussian: everything up to the tile rectangle.
struct Proj {
	bool alive;
	float pix_x, pix_y, depth, conic_a, conic_b, conic_c;
	float cov0, cov1, lambda1, lambda2;
	int radius, x0, y0, x1, y1;
	uint32_t tnum;
};
// Upper bound of the squared spectral norm of the view matrix' 3x3 block (1 for a rigid camera): the spectral
// radius of W^T W is at most its largest absolute row sum.
__device__ __forceinline__ float view_norm2_bound(const float *vm)
{
	const float c0[3] = { vm[0], vm[1], vm[2] }, c1[3] = { vm[4], vm[5], vm[6] }, c2[3] = { vm[8], vm[9], vm[10] };
	const float g00 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2], g11 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2],
		g22 = c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2];
	const float g01 = fabsf(c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2]), g02 = fabsf(c0[0] * c2[0] + c0[1] * c2[1] + c0[2] * c2[2]),
		g12 = fabsf(c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2]);
	return fmaxf(g00 + g01 + g02, fmaxf(g01 + g11 + g12, g02 + g12 + g22)) * 1.001f;
}

// Conservative visibility test of k_project (more than half of the Gaussians in front of the camera of a
// room-scale scene cannot reach the frame). Exact part: the near plane (auxiliary.h:139-164, same expression as
// the full projection). Bound: every entry of the 2D covariance T^T Sigma T (forward.cu:74-113) is at most
// B = rho(Sigma) |T|_F^2 in magnitude (+0.3 on the diagonal), so lambda1 <= 2.42 B + 1.05 and the radius is at
// most r_ub; |T|_F <= |W|_2 |J|_F with J's four non-zero entries; getRect is monotone in the radius, so an empty
// rectangle for r_ub (RF: clipped to the level box, see walk_rect) means the reference drops the splat as well
// (forward.cu:229-231). Everything here is evaluated approximately (no double, one division) and padded by
// 1 % + 2 px, far above the rounding of either formulation; NaN/inf anywhere makes the test pass.
// rho: upper bound of the spectral norm of the 3D covariance
template <bool FOV>
__device__ __forceinline__ bool frame_test_rho(const PreArgs &a, const float *vm, const float *pm, const float p[3], const float rho, float hl, float wn2,
	const uint4 *lvb /* the five level boxes, one uint4 apart (k_project keeps them in LDS) */)
{
	const float tz = vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14];
	if (tz <= 0.2f) return false;
	const float hx = pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12];
	const float hy = pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13];
	const float hw = pm[3] * p[0] + pm[7] * p[1] + pm[11] * p[2] + pm[15];
	const float t0 = vm[0] * p[0] + vm[4] * p[1] + vm[8] * p[2] + vm[12];
	const float t1 = vm[1] * p[0] + vm[5] * p[1] + vm[9] * p[2] + vm[13];
	const float p_w = 1.0f / (hw + 0.0000001f);
	const float pix_x = ((hx * p_w + 1.0f) * a.W - 1.0f) * 0.5f, pix_y = ((hy * p_w + 1.0f) * a.H - 1.0f) * 0.5f;
	const float iz = __builtin_amdgcn_rcpf(tz);
	const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
	const float cx = fminf(limx, fmaxf(-limx, t0 * iz)), cy = fminf(limy, fmaxf(-limy, t1 * iz));
	const float ja = a.focal_x * iz, jb = a.focal_y * iz;
	const float jf = (ja * ja) * (1.0f + cx * cx) + (jb * jb) * (1.0f + cy * cy); // |J|_F^2
	const float lam_ub = 2.42f * (rho * (wn2 * jf)) * 1.01f + 1.05f;
	const float r_ub = ceilf(3.0f * __builtin_sqrtf(lam_ub)) * 1.01f + 2.0f;
	if (!(r_ub < 1e9f)) return true;
	int x0, y0, x1, y1;
	get_rect_f(pix_x, pix_y, r_ub, a.gx, a.gy, x0, y0, x1, y1);
	if (FOV)
	{
		const int k = (int)fminf(fmaxf(ceilf(hl + 1.0f), 0.0f), 4.0f);
		const uint4 b = lvb[k];
		x0 = max(x0, a.gx - (int)b.x); y0 = max(y0, a.gy - (int)b.y);
		x1 = min(x1, (int)b.z); y1 = min(y1, (int)b.w);
	}
	return x1 > x0 && y1 > y0;
}
// Sigma = M^T M with M = S R, and R = (1 - 2|v|^2) I + 2 v v^T + 2 r [v]x has the singular values 1 and
// sqrt((1 - 2|v|^2)^2 + 4 r^2 |v|^2) (= 1 for a unit quaternion; the reference does not normalise here):
// rho(Sigma) <= (scale_modifier * smax)^2 * this factor. Without the modifier it is the fourth float of a
// packed_cull row (fovraster.h).
__device__ __forceinline__ float rho_unit(const float sc[3], const float4 q)
{
	const float vv = q.y * q.y + q.z * q.z + q.w * q.w;
	const float smax = fmaxf(fabsf(sc[0]), fmaxf(fabsf(sc[1]), fabsf(sc[2])));
	return smax * smax * fmaxf(1.0f, (1.0f - 2.0f * vv) * (1.0f - 2.0f * vv) + 4.0f * q.x * q.x * vv);
}
template <bool FOV>
__device__ __forceinline__ bool frame_test(const PreArgs &a, const float *vm, const float *pm, int idx, const float p[3], const float sc[3], float4 q,
	float hl, float wn2, const uint4 *lvb)
{
	const float tz = vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14];
	if (tz <= 0.2f) return false;
	float rho;
	if (a.cov3D_precomp != nullptr)
	{
		const float *c = a.cov3D_precomp + 6 * (size_t)idx;
		rho = fabsf(c[0]) + fabsf(c[3]) + fabsf(c[5]) + 2.0f * (fabsf(c[1]) + fabsf(c[2]) + fabsf(c[4]));
	}
	else rho = (a.scale_modifier * a.scale_modifier) * rho_unit(sc, q);
	return frame_test_rho<FOV>(a, vm, pm, p, rho, hl, wn2, lvb);
}

// The reference's per-Gaussian projection (forward.cu:155-262): near plane, 3D covariance, EWA 2D covariance,
// conic, radius, tile rectangle. sc / q: raw scales and rotation (unused with cov3D_precomp).
// stash[4]: receives (xyz | raw scale | rotation | 3D covariance) for the backward pass if want_stash (the caller's registers).
__device__ __forceinline__ Proj project_gaussian(const PreArgs &a, const float *vm, const float *pm, int idx, const float p[3], const float sc[3], float4 q,
	float4 *stash, const bool want_stash)
{
	Proj r; r.alive = false; r.tnum = 0; r.radius = 0; r.x0 = r.y0 = r.x1 = r.y1 = 0;
	r.pix_x = r.pix_y = r.depth = r.conic_a = r.conic_b = r.conic_c = r.cov0 = r.cov1 = r.lambda1 = r.lambda2 = 0.f;
	// near cull: auxiliary.h:139-164
	const float hx = pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12];
	const float hy = pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13];
	const float hw = pm[3] * p[0] + pm[7] * p[1] + pm[11] * p[2] + pm[15];
	const float p_w = 1.0f / (hw + 0.0000001f);
	const float projx = hx * p_w, projy = hy * p_w;
	float t[3];
	t[0] = vm[0] * p[0] + vm[4] * p[1] + vm[8] * p[2] + vm[12];
	t[1] = vm[1] * p[0] + vm[5] * p[1] + vm[9] * p[2] + vm[13];
	t[2] = vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14];
	r.depth = t[2];
	if (r.depth <= 0.2f) return r;

	// Jacobian of the projection and T = W * J (EWA, forward.cu:74-113)
	const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
	const float txtz = t[0] / t[2], tytz = t[1] / t[2];
	const float tx = fminf(limx, fmaxf(-limx, txtz)) * t[2];
	const float ty = fminf(limy, fmaxf(-limy, tytz)) * t[2];
	const M3 J = m3_cols(a.focal_x / t[2], 0, -(a.focal_x * tx) / (t[2] * t[2]),
		0, a.focal_y / t[2], -(a.focal_y * ty) / (t[2] * t[2]),
		0, 0, 0);
	const M3 Wm = m3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
	const M3 Tm = m3_mul(Wm, J);
	// ndc2Pix is evaluated in double in the reference (auxiliary.h:41-44)
	r.pix_x = (float)((((double)projx + 1.0) * a.W - 1.0) * 0.5);
	r.pix_y = (float)((((double)projy + 1.0) * a.H - 1.0) * 0.5);

	float cov3D[6];
	if (a.cov3D_precomp != nullptr)
	{
#pragma unroll
		for (int i = 0; i < 6; i++) cov3D[i] = a.cov3D_precomp[6 * (size_t)idx + i];
	}
	const float mod = a.scale_modifier;
	const float s0 = mod * sc[0], s1 = mod * sc[1], s2 = mod * sc[2];

	// 3D covariance: forward.cu:118-152
	if (a.cov3D_precomp == nullptr)
	{
		const float rr = q.x, x = q.y, y = q.z, z = q.w;
		const M3 S = m3_cols(s0, 0, 0, 0, s1, 0, 0, 0, s2);
		const M3 R = m3_cols(
			1.f - 2.f * (y * y + z * z), 2.f * (x * y - rr * z), 2.f * (x * z + rr * y),
			2.f * (x * y + rr * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - rr * x),
			2.f * (x * z - rr * y), 2.f * (y * z + rr * x), 1.f - 2.f * (x * x + y * y));
		const M3 Mm = m3_mul(S, R);
		const M3 Sg = m3_mul(m3_t(Mm), Mm);
		cov3D[0] = Sg.c[0][0]; cov3D[1] = Sg.c[0][1]; cov3D[2] = Sg.c[0][2];
		cov3D[3] = Sg.c[1][1]; cov3D[4] = Sg.c[1][2]; cov3D[5] = Sg.c[2][2];
		if (want_stash)
		{
			stash[0] = make_float4(p[0], p[1], p[2], sc[0]);
			stash[1] = make_float4(sc[1], sc[2], q.x, q.y);
			stash[2] = make_float4(q.z, q.w, cov3D[0], cov3D[1]);
			stash[3] = make_float4(cov3D[2], cov3D[3], cov3D[4], cov3D[5]);
		}
	}

	// 2D covariance (EWA): forward.cu:74-113
	float cov[3];
	{
		const M3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
		const M3 c = m3_mul(m3_mul(m3_t(Tm), m3_t(Vrk)), Tm);
		cov[0] = c.c[0][0] + 0.3f; cov[1] = c.c[0][1]; cov[2] = c.c[1][1] + 0.3f;
	}
	const float det = cov[0] * cov[2] - cov[1] * cov[1];
	if (det == 0.0f) return r;
	const float det_inv = 1.f / det;
	r.conic_a = cov[2] * det_inv; r.conic_b = -cov[1] * det_inv; r.conic_c = cov[0] * det_inv;
	const float mid = 0.5f * (cov[0] + cov[2]);
	r.lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
	r.lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
	const float my_radius = ceilf(3.f * sqrtf(fmaxf(r.lambda1, r.lambda2)));
	r.radius = f2i(my_radius);
	get_rect(r.pix_x, r.pix_y, r.radius, a.gx, a.gy, r.x0, r.y0, r.x1, r.y1);
	r.tnum = (uint32_t)(r.y1 - r.y0) * (uint32_t)(r.x1 - r.x0);
	r.cov0 = cov[0]; r.cov1 = cov[1];
	r.alive = r.tnum != 0;
	return r;
}

// wave-wide reductions / broadcasts (64 lanes)
__device__ __forceinline__ uint32_t wave_add_u32(uint32_t v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off);
	return v;
}
__device__ __forceinline__ float wave_min_f32(float v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
	return v;
}
__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
	return v;
}
__device__ __forceinline__ float bcast_f(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
__device__ __forceinline__ int bcast_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// ---- wave-balanced (Gaussian, tile) pair processing ------------------------------------------
// The reference walks each splat's tile rectangle serially in the splat's own thread; rectangles
// range from 1 tile to the whole frame (a near-camera splat covers all 8160 tiles at 1080p), so on
// a wave64 most lanes idle behind the largest one. Here the rectangles of the 64 Gaussians of a
// wave are concatenated into one list of pairs (prefix sum of the tile counts) and processed 64
// pairs per step: lane l of step k handles pair k+l, finds the owning lane by binary search on
// the prefix sums (ds_bpermute) and pulls the owner's parameters with shuffles. Results go back
// to the owner through ballots: a lane's pairs are contiguous, so it masks out its own segment.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
#pragma unroll
	for (int off = 1; off < 64; off <<= 1)
	{
		const uint32_t n = (uint32_t)__shfl_up((int)v, off);
		if (lane >= off) v += n;
	}
	return v;
}
// inclusive prefix maximum over the 64 lanes with DPP row shifts / row broadcasts (no LDS round trips)
__device__ __forceinline__ int wave_scan_max_i32(int v)
{
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x111, 0xF, 0xF, false)); // row_shr:1
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x112, 0xF, 0xF, false)); // row_shr:2
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x114, 0xF, 0xF, false)); // row_shr:4
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x118, 0xF, 0xF, false)); // row_shr:8
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x142, 0xA, 0xF, false)); // row_bcast15 -> rows 1,3
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x143, 0xC, 0xF, false)); // row_bcast31 -> rows 2,3
	return v;
}
// Owner lane of every pair of the current step: each lane whose segment [seg_a, seg_b) is non-empty drops
// its lane id at position seg_a of a 64-entry LDS row; a prefix maximum spreads it over the segment
// (segments are disjoint, ordered by lane). Two LDS writes + one read + six DPP ops, instead of a six-deep
// chain of dependent ds_bpermute in a binary search.
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
// smallest lane L with incl[L] > j  (j < total)
__device__ __forceinline__ int pair_owner(uint32_t incl, uint32_t j)
{
	int lo = 0, hi = 63;
#pragma unroll
	for (int it = 0; it < 6; it++)
	{
		const int mid = (lo + hi) >> 1;
		const uint32_t v = (uint32_t)__shfl((int)incl, mid);
		if (v > j) hi = mid; else lo = mid + 1;
	}
	return lo;
}
// bits [a, b) of a 64-bit mask, for 0 <= a <= b <= 64
__device__ __forceinline__ unsigned long long seg_mask(int a, int b)
{
	if (b <= a) return 0ull;
	const unsigned long long hi = b >= 64 ? ~0ull : ((1ull << b) - 1ull);
	return hi & ~((1ull << a) - 1ull);
}

// Stage 1: the part of the reference's preprocess every Gaussian must go through -- near plane, projected
// centre -- plus a conservative frame test (frame_test), streaming over the whole cloud with persistent waves (the next
// chunks' inputs prefetched into registers). Typically 10-30 % of a scene survive; everything expensive (covariance
// chain, eigen axes, tile walk, SH) then runs on dense waves in k_bin.
// The survivors are listed IN INDEX ORDER (GeomWS): every wave takes a run of CONSECUTIVE 64-Gaussian chunks and
// leaves its survivors' indices (and, foveated variants, their input rows) in a region of its own, in the order it meets
// them -- no counter, no atomics, no LDS staging (round 2 appended to one list through a device-wide counter: one atomic per
// ~450 survivors, a 45 us queue of flushes at the end of the kernel before those were batched per workgroup) -- and
// leaves its count; k_bin strings the regions together.
#ifndef FR_PROJ_THREADS
#define FR_PROJ_THREADS 1024
#endif
#ifndef FR_PROJ_DEPTH
#define FR_PROJ_DEPTH 3 // (2: 0.120 ms, 3: 0.110 ms, 4: 0.131 ms on the bench scene)
#endif
// PACKED: position and covariance bound come as one float4 per Gaussian (packed_cull) instead of 3 + 3 + 4 floats
// from three tensors; with 5 instead of 11 registers per chunk the ring can be deeper.
#ifndef FR_PROJ_DEPTH_PACKED
#define FR_PROJ_DEPTH_PACKED 4
#endif
// chunks per wave and the first slot of a wave's region, for a grid of nwaves waves over P Gaussians
__host__ __device__ inline int proj_chunks_per_wave(int P, int nwaves) { const int nchunks = (P + 63) / 64; return (nchunks + nwaves - 1) / nwaves; }
template <int VARIANT, bool PACKED = false>
__global__ void __launch_bounds__(FR_PROJ_THREADS) k_project(const PreArgs a)
{
	constexpr int DEPTH = PACKED ? FR_PROJ_DEPTH_PACKED : FR_PROJ_DEPTH;
	constexpr bool FOV = VARIANT == FR_VARIANT_FOV_PCHECK_OBB;
	const int lane = threadIdx.x & 63;
	// RF: the five level boxes of the frame test, in LDS (one dependent global load per Gaussian otherwise)
	__shared__ uint4 s_lvb[5];
	if (FOV && threadIdx.x < 5) s_lvb[threadIdx.x] = *(const uint4 *)(a.lv_bbox + threadIdx.x * FR_LV_BBOX_STRIDE);
	if (FOV) __syncthreads();
	// Unpacked model: a survivor's eleven input floats are in this wave's registers right now, and k_bin would have to
	// fetch them again through four gathers that touch every cache line of means3D / scales / rotations / highest_levels
	// (a line holds 8-10 Gaussians, one in eight survives: 264 MB of lines per frame for 34 MB of rows). So the survivor
	// stores its ROW (48 bytes, GeomWS::crow) beside its index. A k_bin wave then reads a contiguous 3 KB.
	constexpr bool ROWS = !PACKED && FOV; // the level box leaves one Gaussian in eight; at one in three (plain frames) the rows cost
	                                       // more than k_bin's gathers, whose lines are then mostly used (training step +1.5 %)
	const int nwaves = (int)gridDim.x * (FR_PROJ_THREADS / 64);
	const int nchunks = (a.P + 63) / 64;
	const int wave_gid = (int)blockIdx.x * (FR_PROJ_THREADS / 64) + (int)(threadIdx.x >> 6);
	const int cpw = proj_chunks_per_wave(a.P, nwaves);
	const int c0 = wave_gid * cpw, c1 = min(nchunks, c0 + cpw); // this wave's chunks
	const uint32_t row_base = (uint32_t)wave_gid * (uint32_t)cpw * 64u; // ... and its region: a slot per Gaussian it visits
	uint32_t nrow = 0; // survivors so far (wave-uniform)
	const bool have_sr = a.cov3D_precomp == nullptr;
	const float wn2 = view_norm2_bound(a.viewmatrix);
	// The camera matrices go to scalar registers once: read through the argument pointers inside the loop they are
	// re-fetched for every chunk (the stores below may alias them as far as the compiler knows), and waiting for
	// them also waits for the prefetches issued before.
	float vm[16], pm[16];
#pragma unroll
	for (int i = 0; i < 16; i++)
	{
		vm[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.viewmatrix[i])));
		pm[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.projmatrix[i])));
	}
	// No branch around the prefetch loads (a conditional load is waited for where its branch ends): out-of-range lanes
	// read the last Gaussian; without scales / rotations the unused values come from the cov3D_precomp array.
	const float *sc_src = have_sr ? a.scales : a.cov3D_precomp;
	const float *q_src = have_sr ? a.rotations : a.cov3D_precomp;
	bool odd_level = false; // RF: a highest level other than 0, 1, 2, 3 went by (see step)
	auto fetch = [&](const int chunk) __attribute__((always_inline))
	{
		const size_t i = (size_t)min(chunk * 64 + lane, a.P - 1);
		RawGaussian w;
		if (PACKED)
		{
			const float4 c = ((const float4 *)a.packed_cull)[i];
			w.p[0] = c.x; w.p[1] = c.y; w.p[2] = c.z; w.sc[0] = c.w; w.sc[1] = w.sc[2] = 0.0f; w.q = make_float4(0, 0, 0, 0);
		}
		else
		{
#pragma unroll
			// (non-temporal: the cloud is read once per frame, 264 MB that would only push the rows the binning kernels come back
			// to out of the caches: k_project 84 -> 79 us)
			for (int k = 0; k < 3; k++) { w.p[k] = __builtin_nontemporal_load(a.means3D + 3 * i + k); w.sc[k] = __builtin_nontemporal_load(sc_src + 3 * i + k); }
			w.q = nt_load4(q_src + 4 * i);
		}
		w.hl = FOV ? __builtin_nontemporal_load(a.highest_levels + i) : 0.0f;
		return w;
	};
	auto step = [&](const RawGaussian &cur, const int chunk) __attribute__((always_inline))
	{
		const int idx = chunk * 64 + lane;
		bool maybe = false;
		// RF: the binning kernels filter tiles by `tile level < highest level + 1` from a 4-bit-per-tile LDS table, which is exact
		// for highest levels 0, 1, 2, 3 (what a model holds); anything else sends them to the tiles' float levels in global memory
		if (FOV && chunk < c1 && idx < a.P && !(cur.hl == 0.0f || cur.hl == 1.0f || cur.hl == 2.0f || cur.hl == 3.0f)) odd_level = true;
		if (chunk < c1 && idx < a.P)
		{
			if (PACKED) maybe = frame_test_rho<FOV>(a, vm, pm, cur.p, (a.scale_modifier * a.scale_modifier) * cur.sc[0], cur.hl, wn2, s_lvb);
			else if (!FOV && a.raw)
			{
				// raw parameters: exp is monotone, so the largest scale is exp(largest raw scale) -- one exp instead of
				// three --, and the normalised quaternion is a unit one up to rounding (rho_unit's factor: 1 + 4 vv eps). (The
				// rotations are still fetched: not reading them here takes 20 us off this kernel and puts 45 us on k_bin, which
				// then finds their lines cold.)
				const float smax = act_scale(fmaxf(cur.sc[0], fmaxf(cur.sc[1], cur.sc[2])));
				maybe = frame_test_rho<FOV>(a, vm, pm, cur.p, (a.scale_modifier * a.scale_modifier) * (smax * smax) * 1.00001f, cur.hl, wn2, s_lvb);
			}
			else maybe = frame_test<FOV>(a, vm, pm, idx, cur.p, cur.sc, cur.q, cur.hl, wn2, s_lvb);
			a.radii[idx] = 0; // whole lines (a store with the survivors masked out is a partial-line write); k_bin, which runs
			                  // after this kernel, writes the radius of every survivor
			// auxiliary.h:156-160: the reference traps on a near-culled point of a cloud declared prefiltered
			if (a.prefiltered && !maybe && (vm[2] * cur.p[0] + vm[6] * cur.p[1] + vm[10] * cur.p[2] + vm[14]) <= 0.2f) atomicOr(a.geom.slab_ctr, 1u);
		}
		const unsigned long long m = __ballot(maybe);
		if (maybe)
		{
			const uint32_t slot = row_base + nrow + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
			a.geom.vis_seg[slot] = (uint32_t)idx;
			if (ROWS)
			{
				float4 *row = a.geom.crow + 3 * (size_t)slot;
				row[0] = make_float4(cur.p[0], cur.p[1], cur.p[2], cur.sc[0]);
				row[1] = make_float4(cur.sc[1], cur.sc[2], cur.q.x, cur.q.y);
				row[2] = make_float4(cur.q.z, cur.q.w, cur.hl, __uint_as_float((uint32_t)idx));
			}
		}
		nrow += (uint32_t)__popcll(m);
	};
#ifdef FR_PROJ_TIMERS
	const uint64_t tm0 = wall_clock64(); uint64_t tm_first = 0;
#endif
	// FR_PROJ_DEPTH chunks in flight per wave, each in its own register set that is refilled in place
	RawGaussian R[DEPTH];
#pragma unroll
	for (int d = 0; d < DEPTH; d++) R[d] = fetch(min(c0 + d, nchunks - 1));
	for (int base = c0; base < c1; base += DEPTH)
	{
#pragma unroll
		for (int d = 0; d < DEPTH; d++)
		{
			step(R[d], base + d);
#ifdef FR_PROJ_TIMERS
			if (tm_first == 0) tm_first = wall_clock64();
#endif
			R[d] = fetch(min(base + d + DEPTH, nchunks - 1));
		}
	}
	// the wave's number of survivors: k_bin strings the regions together (it scans these counts at its start and finds the
	// region of every item from the running sums -- no pass of its own over the survivors, no counter, no waiting. Tried: a
	// kernel that copies the regions into one list (11 us of launch and dependent round trips); the copy at the end of this
	// kernel with every workgroup adding up the counts of those in front of it (the polling of the workgroups that finish
	// first halves the memory bandwidth of those still streaming: 74 -> 157 us).
	// The wave's count leaves as a RETURNING device-scope atomic: it is performed at the memory side once its value is back, so
	// the workgroup can be counted as done without a release fence (an agent-scope fence writes the L2's dirty lines back --
	// this kernel's radii and candidate rows -- once per wave: 80 -> 250 us).
	if (lane == 0)
	{
		const uint32_t was = atomicExch(a.geom.proj_counts + wave_gid, nrow);
		asm volatile("" :: "v"(was));
	}
	if (FOV && __any(odd_level) && lane == 0) atomicOr(a.geom.slab_ctr + 3, 1u);
	// The LAST workgroup to get here turns the counts into running sums (wbase[w] = first item of wave w's region, wbase[waves] =
	// slab_ctr[1] = the number of items): 8192 counts, eight per thread, ~2 us at the tail of a kernel whose workgroups finish
	// within a few us of each other -- where every workgroup of round 3's binning kernel scanned them for itself in its prologue.
	__shared__ uint32_t s_last, s_part[FR_PROJ_THREADS / 64];
	__syncthreads();
	if (threadIdx.x == 0) s_last = atomicAdd(a.geom.slab_ctr + 2, 1u) == gridDim.x - 1 ? 1u : 0u;
	__syncthreads();
	if (s_last)
	{
		// (the counts are read with device-scope atomic loads: they come from the memory side, not from this XCD's L2)
		const int per = (nwaves + FR_PROJ_THREADS - 1) / FR_PROJ_THREADS; // consecutive waves per thread
		const int w0 = (int)threadIdx.x * per, w1 = min(nwaves, w0 + per);
		uint32_t cv[FR_PROJ_MAX_WAVES / FR_PROJ_THREADS], mine = 0;
#pragma unroll
		for (int k = 0; k < FR_PROJ_MAX_WAVES / FR_PROJ_THREADS; k++)
		{
			cv[k] = w0 + k < w1 ? __hip_atomic_load(a.geom.proj_counts + w0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
			mine += cv[k];
		}
		const uint32_t sc = wave_incl_scan_u32(mine, lane);
		if (lane == 63) s_part[threadIdx.x >> 6] = sc;
		__syncthreads();
		uint32_t off = 0;
		for (int w = 0; w < (int)(threadIdx.x >> 6); w++) off += s_part[w];
		uint32_t run = off + sc - mine;
#pragma unroll
		for (int k = 0; k < FR_PROJ_MAX_WAVES / FR_PROJ_THREADS; k++)
			if (w0 + k < w1) { a.geom.wbase[w0 + k] = run; run += cv[k]; }
		if (w1 == nwaves && w0 < w1) { a.geom.wbase[nwaves] = run; a.geom.slab_ctr[1] = run; }
		if (threadIdx.x < FR_MAX_REGIONS) a.geom.rtotal[threadIdx.x] = 0; // (the regions' list lengths: added up by k_bin's workgroups, the kernel behind this one)
	}
#ifdef FR_PROJ_TIMERS
	if (lane == 0)
	{
		// developer build (tools/proj_stats.py): per-wave (start, first chunk done, end) in ns / 10, in the unused covariance rows
		float *d = a.geom.cov3D + (size_t)wave_gid * 4;
		d[0] = (float)(tm0 & 0xffffff); d[1] = (float)(tm_first - tm0); d[2] = (float)(wall_clock64() - tm0); d[3] = (float)(c1 - c0);
	}
#endif
}

// ---- per-item rows through LDS ------------------------------------------------------------------
// Everything the library keeps per candidate is a row of N float4 indexed by ITEM, and a wave works on 64 consecutive items:
// its rows are one contiguous piece of memory, but a lane's own row lies 16 N bytes from its neighbour's, so a load or store
// instruction of the lanes' own rows touches 24-32 cache lines (~3 cycles per lane of the CU's one address unit: eleven such
// stores per slab were a quarter of round 2's binning kernel). Transposed through the wave's LDS rows an instruction moves one
// contiguous kilobyte. st: the wave's staging area of >= 64 N float4; nitems: valid items of the wave (0..64).
template <int N>
__device__ __forceinline__ void rows_store(const float4 (&rows)[N], float4 *st, float4 *dst, const int nitems, const int lane)
{
#pragma unroll
	for (int i = 0; i < N; i++) st[N * lane + i] = rows[i];
	FR_WAVE_LDS_SYNC();
#pragma unroll
	for (int i = 0; i < N; i++) { const int c = i * 64 + lane; if (c < N * nitems) dst[c] = st[c]; }
	FR_WAVE_LDS_SYNC();
}
// the two halves of the load apart, so that a kernel can have the next slab's pieces in flight while it works on this one
template <int N>
__device__ __forceinline__ void rows_fetch(float4 (&piece)[N], const float4 *src, const int nitems, const int lane)
{
#pragma unroll
	for (int i = 0; i < N; i++) { const int c = i * 64 + lane; piece[i] = c < N * nitems ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f); }
}
template <int N>
__device__ __forceinline__ void rows_unpack(float4 (&rows)[N], const float4 (&piece)[N], float4 *st, const int lane)
{
#pragma unroll
	for (int i = 0; i < N; i++) st[i * 64 + lane] = piece[i];
	FR_WAVE_LDS_SYNC();
#pragma unroll
	for (int i = 0; i < N; i++) rows[i] = st[N * lane + i];
	FR_WAVE_LDS_SYNC();
}

// a candidate's inputs by index: the packed row, or gathers from the model's tensors
template <bool PACKED, bool FOV>
__device__ __forceinline__ void load_candidate(const PreArgs &a, const int idx, RawGaussian &g)
{
	if (PACKED)
	{
		// one 64-byte row instead of four or five mostly-unused cache lines
		const float4 *pg = (const float4 *)a.packed_geom + 4 * (size_t)idx;
		const float4 g0 = pg[0], g1 = pg[1], g2 = pg[2];
		g.p[0] = g0.x; g.p[1] = g0.y; g.p[2] = g0.z;
		g.sc[0] = g0.w; g.sc[1] = g1.x; g.sc[2] = g1.y;
		g.q = make_float4(g1.z, g1.w, g2.x, g2.y);
		g.hl = g2.z;
	}
	else
	{
#pragma unroll
		for (int i = 0; i < 3; i++) g.p[i] = a.means3D[3 * (size_t)idx + i];
		if (a.cov3D_precomp == nullptr)
		{
#pragma unroll
			for (int i = 0; i < 3; i++) g.sc[i] = a.scales[3 * (size_t)idx + i];
			g.q = ((const float4 *)a.rotations)[idx];
		}
		if (FOV) g.hl = a.highest_levels[idx];
	}
}

// The full projection of ONE candidate (the rest of preprocessCUDA: 3D covariance, EWA 2D covariance, conic, radius, OBB axes, the
// clipped rectangle to walk) and the rows it leaves: walk record, blend record (position in the colour slots), training: the
// backward pass's input row (the front end of k_bin).
struct GeomOut { float4 wrow[4], rrow[3], stash[4]; float inv_qnorm; int radius; bool alive; };
template <int VARIANT>
__device__ __forceinline__ void geom_item(const PreArgs &a, const float *cam_vm, const float *cam_pm, const uint4 *lvbox, const bool valid,
	const int idx, RawGaussian g, const bool raw, GeomOut &o)
{
	constexpr bool CULL = VARIANT != FR_VARIANT_ORIGINAL;
	constexpr bool FOV = is_fov(VARIANT);
	constexpr bool LEVELCOL = VARIANT == FR_VARIANT_FOV_PCHECK_OBB;
	Proj pr; pr.alive = false; pr.tnum = 0; pr.x0 = pr.y0 = pr.x1 = pr.y1 = 0; pr.radius = 0;
	pr.pix_x = pr.pix_y = pr.depth = pr.conic_a = pr.conic_b = pr.conic_c = 0.f;
#pragma unroll
	for (int i = 0; i < 4; i++) o.stash[i] = make_float4(0, 0, 0, 0);
	float4 ev = make_float4(0, 0, 0, 0);
	float2 el = make_float2(0, 0);
	bool boxtest = false;
	o.inv_qnorm = 1.0f;
	if (valid)
	{
		if (raw)
		{
#pragma unroll
			for (int i = 0; i < 3; i++) g.sc[i] = act_scale(g.sc[i]);
			g.q = act_rotation(g.q, &o.inv_qnorm);
		}
		pr = project_gaussian(a, cam_vm, cam_pm, idx, g.p, g.sc, g.q, o.stash, a.write_cov3D != 0);
		if (pr.alive)
		{
			if (CULL && pr.tnum > 1)
			{
				// eigen axes of the 2D covariance: RS forward.cu:244-265 (normalize() restated as 1/sqrt)
				float e1x = -pr.cov1, e1y = pr.cov0 - pr.lambda1, e2x = -pr.cov1, e2y = pr.cov0 - pr.lambda2;
				const float n1 = 1.0f / sqrtf(e1x * e1x + e1y * e1y);
				e1x *= n1; e1y *= n1;
				const float n2 = 1.0f / sqrtf(e2x * e2x + e2y * e2y);
				e2x *= n2; e2y *= n2;
				ev = make_float4(e1x, e1y, e2x, e2y);
				el = make_float2(3.0f * sqrtf(pr.lambda1), 3.0f * sqrtf(pr.lambda2));
			}
			const WalkRect wr = walk_rect<CULL, FOV>(pr.pix_x, pr.pix_y, pr.radius, a.gx, a.gy, ev, el, g.hl, lvbox, 1);
			pr.x0 = wr.x0; pr.y0 = wr.y0; pr.x1 = wr.x1; pr.y1 = wr.y1; pr.tnum = wr.tnum; boxtest = wr.boxtest;
			pr.alive = wr.tnum != 0;
		}
	}
	o.alive = pr.alive; o.radius = pr.radius;
	const uint32_t flags = (pr.alive ? 1u : 0u) | (boxtest ? 2u : 0u);
	o.wrow[0] = make_float4(pr.pix_x, pr.pix_y, ev.x, ev.y);
	o.wrow[1] = make_float4(ev.z, ev.w, el.x, el.y);
	o.wrow[2] = make_float4(__uint_as_float((uint32_t)idx | (flags << 30)), pr.depth, __uint_as_float((uint32_t)pr.x0 | ((uint32_t)pr.y0 << 16)),
		__uint_as_float((uint32_t)(pr.x1 - pr.x0)));
	o.wrow[3] = make_float4(__uint_as_float(pr.tnum), g.hl, 0.0f, 0.0f);
	// the blend record; its colour slots carry the position until k_bin has evaluated the colour (RF: the colours live in
	// the level rows, the slots keep the position)
	o.rrow[0] = make_float4(pr.pix_x, pr.pix_y, pr.conic_a, pr.conic_b);
	o.rrow[1] = make_float4(pr.conic_c, LEVELCOL ? g.hl : 0.0f, g.p[0], g.p[1]);
	o.rrow[2] = make_float4(g.p[2], pr.depth, (FOV && !LEVELCOL) ? g.hl : 0.0f, __int_as_float(idx));
}

// The colour(s) of one item (forward.cu:20-71 computeColorFromSH; RF rasterizer_impl.cu:37-84, 490-530 compute_fov_colors), for
// the items that landed in a tile only: the SH rows are the largest read of a frame (180-192 bytes per Gaussian, unaligned), and
// a quarter of the candidates that were projected need none. r: the item's blend record as geom_item left it (position in the
// colour slots); on return the record with opacity / colour / clamp bits (variants with one colour per Gaussian) or, RF, the
// level rows lv[lo..hi] of the item's level range lr (the others stay zero: nobody reads them).
template <int VARIANT, bool PACKED>
__device__ __forceinline__ void colour_item(const PreArgs &a, const bool rows_ok, const uint32_t lr, float4 (&r)[3], float4 (&lv)[FR_FOV_LEVELS])
{
	constexpr bool FOV = is_fov(VARIANT);
	constexpr bool LEVELCOL = VARIANT == FR_VARIANT_FOV_PCHECK_OBB;
	const int idx = __float_as_int(r[2].w);
	const float dirx = r[1].z - a.campos[0], diry = r[1].w - a.campos[1], dirz = r[2].x - a.campos[2];
	const float *pcol = PACKED ? a.packed_colour + 64 * (size_t)idx : nullptr;
	if (!LEVELCOL)
	{
		float rgb[3] = { 0, 0, 0 };
		uint32_t clamp_bits = 0;
		const float op_in = PACKED ? a.packed_geom[16 * (size_t)idx + 12] : a.opacities[idx];
		if (a.colors_precomp == nullptr)
		{
			float c[3];
			if (rows_ok)
			{
				ShRows sh;
				if (PACKED) { sh_fetch<true>(pcol, sh); const float dc[3] = { pcol[45], pcol[46], pcol[47] }; sh_eval<true>(a.D, sh, dc, dirx, diry, dirz, c); }
				else if (a.shs_rest != nullptr)
				{
					sh_fetch<true>(a.shs_rest + (size_t)idx * (a.M - 1) * 3, sh);
					const float dc[3] = { a.shs[3 * (size_t)idx], a.shs[3 * (size_t)idx + 1], a.shs[3 * (size_t)idx + 2] };
					sh_eval<true>(a.D, sh, dc, dirx, diry, dirz, c);
				}
				else { sh_fetch<false>(a.shs + (size_t)idx * a.M * 3, sh); sh_eval<false>(a.D, sh, nullptr, dirx, diry, dirz, c); }
			}
			else if (a.shs_rest != nullptr)
				sh_colour<true>(a.D, (a.M - 1) * 3, a.shs_rest + (size_t)idx * (a.M - 1) * 3, a.shs + 3 * (size_t)idx, dirx, diry, dirz, c);
			else
				sh_colour<false>(a.D, a.M * 3, a.shs + (size_t)idx * a.M * 3, nullptr, dirx, diry, dirz, c);
#pragma unroll
			for (int ch = 0; ch < 3; ch++) { if (c[ch] < 0) clamp_bits |= 1u << ch; rgb[ch] = fmaxf(c[ch], 0.0f); }
		}
		else
		{
#pragma unroll
			for (int ch = 0; ch < 3; ch++) rgb[ch] = a.colors_precomp[3 * (size_t)idx + ch];
		}
		const float opacity = (!PACKED && !FOV && a.raw) ? act_opacity(op_in) : op_in;
		r[1] = make_float4(r[1].x, opacity, rgb[0], rgb[1]);
		// third part: the Gaussian's index (the statistics of the training variants and the gradients are per Gaussian); the
		// shared-model foveated variant (no backward, no clamp bits needed) carries the Gaussian's highest level in the clamp slot
		r[2] = make_float4(rgb[2], r[2].y, FOV ? r[2].z : __uint_as_float(clamp_bits), r[2].w);
	}
	else
	{
		// RF rasterizer_impl.cu:490-530: per-level colours; all four levels' DC colours (12 floats) and opacities are fetched
		// with the SH coefficients: one round trip
		const f4u *dcp = (const f4u *)(PACKED ? pcol + 48 : a.shs_dcs + (size_t)idx * 3 * FR_FOV_LEVELS);
		const f4u dc0 = dcp[0], dc1 = dcp[1], dc2 = dcp[2];
		const f4u opl = *(const f4u *)(PACKED ? a.packed_geom + 16 * (size_t)idx + 12 : a.opacities + (size_t)idx * FR_FOV_LEVELS);
		float rest[3];
		if (rows_ok)
		{
			ShRows sh;
			sh_fetch<true>(PACKED ? pcol : a.shs + (size_t)idx * a.M * 3, sh);
			sh_eval<true>(a.D, sh, nullptr, dirx, diry, dirz, rest);
		}
		else sh_colour<true>(a.D, a.M * 3, a.shs + (size_t)idx * a.M * 3, nullptr, dirx, diry, dirz, rest);
		const float dcs[12] = { dc0.x, dc0.y, dc0.z, dc0.w, dc1.x, dc1.y, dc1.z, dc1.w, dc2.x, dc2.y, dc2.z, dc2.w };
		const float ops[4] = { opl.x, opl.y, opl.z, opl.w };
		static_assert(FR_FOV_LEVELS == 4, "level data is fetched as float4s");
		const int lo = (int)(lr & 0xffu), hi = (int)((lr >> 8) & 0xffu);
#pragma unroll
		for (int l = 0; l < FR_FOV_LEVELS; l++)
		{
			if (l >= lo && l <= hi)
			{
				float4 v;
				v.x = fmaxf(FR_SH_C0 * dcs[3 * l] + rest[0], 0.0f);
				v.y = fmaxf(FR_SH_C0 * dcs[3 * l + 1] + rest[1], 0.0f);
				v.z = fmaxf(FR_SH_C0 * dcs[3 * l + 2] + rest[2], 0.0f);
				v.w = ops[l];
				lv[l] = v;
			}
		}
	}
}

// Stage 2 (the rest of preprocessCUDA, RS forward.cu:155-293 / RF :105-238; OBB_test RS rasterizer_impl.cu:70-146, filter RF :264-383;
// the colours, forward.cu:20-71 / RF rasterizer_impl.cu:490-530): for every survivor of the cull pass
//   (1) the full projection (geom_item: 3D covariance, EWA 2D covariance, conic, radius, OBB axes, the clipped rectangle to walk)
//       -- inputs: the candidate row k_project stored (CROW: foveated variants), or the packed row / gathers from the model's tensors;
//   (2) the tiles it really lands in (OBB / foveal tests, three walks by splat size), counted in a per-workgroup LDS histogram;
//   (3) for the items that landed somewhere, the colour(s);
// and its rows out: walk record (k_emit), blend record, RF level rows, training: the backward pass's input row and the cleared
// gradient row -- all through LDS as contiguous kilobytes.
// One persistent workgroup of FR_BIN_THREADS threads per CU; a wave takes the 64-item slabs wave, wave + waves, ... (STATIC: rounds
// 2-3 handed the slabs out through eight atomic counters for balance -- 15 000 returning atomics on eight addresses were 45 us of a
// 115-us walk kernel, more than its tile walks; k_emit takes the same slabs, because its bucket offsets are per workgroup).
// The three parts are bound by different things -- gathers and a 600-instruction dependent chain; VALU / LDS work that leaves the
// memory system idle; the frame's largest scattered read with no arithmetic to speak of -- and at sixteen waves per CU one wave's
// part runs under the others'. Round 4 took the kernel apart to see: projection alone (k_geom, one wave per cull-pass region, 94
// registers) 37 us, walks alone 73 us (46 + a 27-us skeleton), colours alone 94 us -- 204 us as three kernels, 182 as two (walks +
// colours in one), 160 as this one (round 3's form of it, with dynamic slabs at eight waves per CU: 175-178).
// LDSH: the counters are an LDS-private histogram (T <= 16 Ki tiles) added once per workgroup to the tiles' global counters
// (the returned values are the workgroup's starts inside the buckets: hist[block][tile]); otherwise (huge tile grids) global
// atomics on tile_count.
// LDSH == 2: 16-bit counts, two tiles per word (tile grids of more than 16 Ki tiles: 4K frames); a workgroup then bins fewer
// than 65 536 items (launch_bin admits the mode only when every wave's share of the slabs is below FR_HIST16_MAX_SLABS).
// Per item it also leaves GeomWS::lrange: FR_ITEM_NONE when no tile is left (k_emit skips the item; its radius is cleared), else
// the level range of RF rasterizer_impl.cu:374-381.
#define BUMP_TILE(ti) do { if (LDSH == 2) atomicAdd(&lds_hist[(ti) >> 1], 1u << (16 * ((ti) & 1))); \
	else if (LDSH) atomicAdd(&lds_hist[(ti)], 1u); else atomicAdd(&a.tile_count[(ti)], 1u); } while (0)
static_assert((FR_BIN_THREADS / 64) * FR_HIST16_MAX_SLABS * 64 <= 65535, "a workgroup's items must fit a 16-bit tile count (LDSH == 2)");
template <int VARIANT, int LDSH, bool PACKED = false, bool CROW = false>
__global__ void __launch_bounds__(FR_BIN_THREADS) k_bin(const PreArgs a)
{
	static_assert(!(PACKED && CROW), "the packed model layout has its own rows");
	constexpr bool CULL = VARIANT != FR_VARIANT_ORIGINAL;
	constexpr bool FOV = is_fov(VARIANT);                            // level map + level filter
	constexpr bool LEVELCOL = VARIANT == FR_VARIANT_FOV_PCHECK_OBB;  // per-level colours / opacities (RF)
	static_assert(!(PACKED && FOV && !LEVELCOL), "the shared-model foveated variant has no packed layout");
		extern __shared__ __attribute__((aligned(16))) uint32_t lds_hist[];
	const int lane = threadIdx.x & 63;
	// RF: every pair step looks its tile's level (and, if kept, its blend flag) up; from global memory those
	// were two dependent ~1 us round trips in a loop that a near-camera splat runs a hundred times. When they
	// fit beside the histogram, the workgroup keeps them in LDS.
	const int hist_words = LDSH == 2 ? (a.T + 1) / 2 : (LDSH ? a.T : 0);
	uint32_t *lds_tab = lds_hist + hist_words;
	const int tab_words = (a.T + 7) / 8;
	// (the table answers `tile level < highest level + 1` only for highest levels 0..3: k_project says if it saw another)
	const bool ldst = FOV && a.lds_tiles && a.geom.slab_ctr[3] == 0u;
	if (FOV && a.lds_tiles)
	{
		// Four bits per tile: min(max(int(tile_min), 0), 7) and the blend flag in bit 3 -- all the walks ask of a tile (the filter
		// compares tile_min with an INTEGER, the level ranges take int(tile_min)): 4 KiB instead of 33 KiB of floats and bits for a
		// 1080p frame, and a 1440p frame's table fits beside its histogram. Sixteen loads at a time (written as a plain copy loop
		// every iteration waited for its own load: a 10 us prologue on every CU before any work); eight lanes pack a word.
		const float *gmin = a.tile_lv + a.T, *gbl = a.tile_lv + 4 * (size_t)a.T;
		for (int t0 = threadIdx.x; t0 < a.T; t0 += 16 * FR_BIN_THREADS)
		{
			float v[16], f[16];
#pragma unroll
			for (int k = 0; k < 16; k++) { const int t = min(t0 + k * FR_BIN_THREADS, a.T - 1); v[k] = gmin[t]; f[k] = gbl[t]; }
#pragma unroll
			for (int k = 0; k < 16; k++)
			{
				const int t = t0 + k * FR_BIN_THREADS;
				// (a NaN level -- a gaze far outside the frame with a steep alpha -- fails `tile_min < highest level + 1` for every Gaussian in the
				// reference, RF rasterizer_impl.cu:802: coded as 7, which fails it for the highest levels 0..3 the table serves)
				uint32_t nib = t < a.T ? ((v[k] == v[k] ? (uint32_t)min(max(f2i(v[k]), 0), 7) : 7u) | (f[k] != 0.0f ? 8u : 0u)) : 0u;
				nib <<= 4 * (lane & 7);
				nib |= (uint32_t)__shfl_xor((int)nib, 1); nib |= (uint32_t)__shfl_xor((int)nib, 2); nib |= (uint32_t)__shfl_xor((int)nib, 4);
				if ((lane & 7) == 0 && (t >> 3) < tab_words) lds_tab[t >> 3] = nib;
			}
		}
	}
	if (LDSH)
		for (int t = threadIdx.x; t < hist_words; t += FR_BIN_THREADS) lds_hist[t] = 0;
	// "giant" splats (FR_GIANT_TNUM+ tiles, up to the whole frame = 128 wave steps) are set aside here and walked
	// by ALL waves of the workgroup after the slab loop: left to the wave that met them they were the kernel's
	// critical path
	__shared__ int s_gitem[FR_GIANT_MAX];
	__shared__ uint32_t s_gcount[FR_GIANT_MAX], s_gmask[FR_GIANT_MAX];
	__shared__ uint32_t s_ng;
	__shared__ uint4 s_lvbox[5]; // RF: the five level boxes every candidate's walk rectangle is clipped to (walk_rect)
	if (FOV && threadIdx.x < 5) s_lvbox[threadIdx.x] = *(const uint4 *)(a.lv_bbox + threadIdx.x * FR_LV_BBOX_STRIDE);
	if (threadIdx.x < FR_GIANT_MAX) { s_gcount[threadIdx.x] = 0; s_gmask[threadIdx.x] = 0; }
	// Which slabs a WORKGROUP takes is static (k_emit's workgroup of the same number replays them: the bucket offsets are per
	// workgroup); which of its waves takes which is not: the waves pull the workgroup's q-th slab from a counter in LDS (round 6: a
	// wave's five slabs differ by the splats they hold, the slowest wave of a workgroup ran 20 % behind the mean and everybody waited
	// for it at the barrier in front of the giant splats and the flush).
	__shared__ uint32_t s_next_slab;
	if (threadIdx.x == 0) { s_ng = 0; s_next_slab = 0; }
	__syncthreads();
	// tile ti against a splat whose filter bound is olim = highest level + 1: does the tile pass, and its level bits
	// (1 << min(max(int(tile_min), 0), 3), | 16 if it blends two levels) -- from the 4-bit table, or from the floats
#define TILE_FILTER(ti, olim, pass_out, bits_out) do { \
		if (ldst) { const uint32_t nb_ = (lds_tab[(ti) >> 3] >> (4 * ((ti) & 7))) & 15u; \
			pass_out = (float)(nb_ & 7u) < (olim); bits_out = (1u << min(nb_ & 7u, 3u)) | ((nb_ & 8u) << 1); } \
		else { const float lv_ = tile_min[(ti)]; pass_out = lv_ < (olim); \
			bits_out = pass_out ? ((1u << min(max(f2i(lv_), 0), 3)) | (tile_bl[(ti)] != 0.0f ? 16u : 0u)) : 0u; } } while (0)
	// Work unit = a "slab" of 64 consecutive items, handled by ONE wave; there is no workgroup
	// barrier inside the loop (a few near-camera splats make some slabs 100x more expensive than others,
	// and waiting for the slowest wave of a workgroup at every slab cost a quarter of the kernel).
	// (dynamic LDS behind the histogram and the table, 16-byte aligned: the waves' staging rows / the pair loop's 64-byte row per
	// lane (see there), and the owner rows of pair_owner_scan)
	float4 *const s_orec = (float4 *)(lds_hist + ((hist_words + ((FOV && a.lds_tiles) ? tab_words : 0) + 3) & ~3));
	int *const s_own = (int *)(s_orec + 4 * FR_BIN_THREADS);
	// [proj_waves + 1] first item of every cull-pass region: a copy in LDS, or -- when the histogram of a 4K tile grid leaves no room
	// for its 32 KiB -- the table in global memory (the slab's binary search then costs thirteen L2 round trips)
	// (two typed pointers, never one selected pointer: that one would be a generic address and every probe a flat load)
	const uint32_t *const s_wbase = (const uint32_t *)(s_own + FR_BIN_THREADS);
	const uint32_t *const g_wbase = a.geom.wbase;
	if (a.wbase_lds)
	{
		uint32_t *const dst = (uint32_t *)(s_own + FR_BIN_THREADS);
		// (sixteen loads at a time, see the table fill above; visible after the workgroup barrier below)
		for (int w0 = threadIdx.x; w0 <= a.proj_waves; w0 += 16 * FR_BIN_THREADS)
		{
			uint32_t v[16];
#pragma unroll
			for (int k = 0; k < 16; k++) v[k] = a.geom.wbase[min(w0 + k * FR_BIN_THREADS, a.proj_waves)];
#pragma unroll
			for (int k = 0; k < 16; k++) if (w0 + k * FR_BIN_THREADS <= a.proj_waves) dst[w0 + k * FR_BIN_THREADS] = v[k];
		}
		__syncthreads();
	}
	float cam_vm[16], cam_pm[16]; // camera matrices in scalar registers
	{
#pragma unroll
		for (int i = 0; i < 16; i++)
		{
			cam_vm[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.viewmatrix[i])));
			cam_pm[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.projmatrix[i])));
		}
	}
	const int V = (int)a.geom.slab_ctr[1]; // entries of vis_list (the cull pass's last workgroup)
	const int nslabs = (V + 63) / 64;
	const int nwaves = (int)gridDim.x * (FR_BIN_THREADS / 64);
	const float *tile_min = FOV ? a.tile_lv + a.T : nullptr;
	const float *tile_bl = FOV ? a.tile_lv + 4 * (size_t)a.T : nullptr;
	// Wave-uniform walk of ONE splat's rectangle: lanes take tiles k0 + lane, k0 + lane + kstep, ... of the on
	// tiles; returns the number of tiles kept and (RF) their level bits. No owner search, no shuffles.
	auto walk_uniform = [&](const int ox0, const int oy0, const int ow, const uint32_t on, const Obb &ob, const float olim,
		const uint32_t k0, const uint32_t kstep, uint32_t &cnt, uint32_t &bits) __attribute__((always_inline))
	{
		const float rw = 1.0f / (float)ow;
		for (uint32_t k = k0; k < on; k += kstep)
		{
			const uint32_t j = k + lane;
			const bool valid = j < on;
			int ry = (int)(((float)j + 0.5f) * rw); // j / ow for j < 2^23, fixed up below
			int rx = (int)j - ry * ow;
			if (rx < 0) { ry--; rx += ow; } else if (rx >= ow) { ry++; rx -= ow; }
			const int x = ox0 + rx, y = oy0 + ry;
			const int ti = valid ? y * a.gx + x : 0;
			bool pass = valid;
			uint32_t m = 0;
			if (CULL)
			{
				uint32_t lb = 0;
				if (FOV) { bool lp; TILE_FILTER(ti, olim, lp, lb); pass = pass && lp; }
				pass = pass && obb_hits_tile(ob, x, y);
				if (FOV && pass) m = lb;
			}
			if (pass) BUMP_TILE(ti);
			cnt += (uint32_t)__popcll(__ballot(pass));
			if (FOV)
			{
#pragma unroll
				for (int bit = 0; bit < 5; bit++)
					if (__ballot((m >> bit) & 1u)) bits |= 1u << bit;
			}
		}
	};
	// int(lowest) / int(highest) of RF rasterizer_impl.cu:374-381 from the per-level bits: truncation is
	// monotone, so int(min(levels)) == min(int(level)); lowest starts at the Gaussian's own level
	auto range_from_mask = [&](const uint32_t lvmask, float &lowest, float &highest, bool &be_blend) __attribute__((always_inline))
	{
		const int lo_bit = __ffs((int)(lvmask & 15u)) - 1, hi_bit = 31 - __clz((int)(lvmask & 15u));
		lowest = fminf(lowest, (float)lo_bit);
		highest = fmaxf(highest, (float)hi_bit);
		be_blend = (lvmask & 16u) != 0;
	};
	// what the item leaves behind: FR_ITEM_NONE or its level range (RF rasterizer_impl.cu:374-381)
	auto range_word = [&](const uint32_t count, const float lowest, const float highest, const bool be_blend) __attribute__((always_inline))
	{
		if (count == 0) return FR_ITEM_NONE;
		if (!FOV) return 0u;
		const int lo = f2i(lowest);
		int hi = f2i(highest);
		if (be_blend) hi = min(hi + 1, FR_FOV_LEVELS - 1);
		return (uint32_t)(lo & 0xff) | ((uint32_t)(hi & 0xff) << 8);
	};
	float4 *const orec = s_orec + 4 * (threadIdx.x & ~63);
	// (uniform) the usual SH storage -- coefficients given, all 16 allocated -- is fetched as whole 16-byte pieces
	const bool rows_ok = a.colors_precomp == nullptr &&
		(PACKED || (LEVELCOL ? a.M * 3 >= 45 : (a.shs_rest != nullptr ? (a.M - 1) * 3 >= 45 : a.M * 3 >= 48)));
	for (;;)
	{
	uint32_t q_ = 0;
	if (lane == 0) q_ = atomicAdd(&s_next_slab, 1u);
	q_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)q_);
	constexpr int BW_ = FR_BIN_THREADS / 64;
	const int slab = (int)blockIdx.x * BW_ + (int)(q_ % BW_) + (int)(q_ / BW_) * nwaves; // (increasing in q)
	if (slab >= nslabs) break;
	const int nv = min(64, V - slab * 64);
	const int item = slab * 64 + lane;
	float4 wr[4];
	float3 fpos = make_float3(0.f, 0.f, 0.f), fconic = make_float3(0.f, 0.f, 0.f); // what the blend record needs beside the walk record
	{
		// the slab's candidates: item i lives in the region of the cull-pass wave whose running count covers it (uniform binary
		// search for the slab's first item, the lanes step on from there: a region holds ~300 survivors)
		RawGaussian g;
		g.p[0] = g.p[1] = g.p[2] = 0.f; g.sc[0] = g.sc[1] = g.sc[2] = 0.f; g.q = make_float4(0, 0, 0, 0); g.hl = 0.f;
		int gidx = 0;
		const bool valid = item < V;
		if (valid)
		{
			// region of the item: wb[lo] <= first < wb[hi], then forward over the (rare, short) regions the slab spans
			auto slot_in = [&](const uint32_t *wb) -> uint32_t {
				int lo = 0, hi = a.proj_waves;
				const uint32_t first = (uint32_t)slab * 64u;
				while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wb[mid] <= first) lo = mid; else hi = mid; }
				int w = lo;
				while ((uint32_t)item >= wb[w + 1]) w++;
				return (uint32_t)w * (uint32_t)a.proj_cpw * 64u + ((uint32_t)item - wb[w]);
			};
			const uint32_t slot = a.wbase_lds ? slot_in(s_wbase) : slot_in(g_wbase);
			if (CROW)
			{
				// the row k_project stored beside the index (48 bytes: the lanes' rows lie in one or two contiguous regions)
				const float4 *cr = a.geom.crow + 3 * (size_t)slot;
				const float4 r0 = cr[0], r1 = cr[1], r2 = cr[2];
				g.p[0] = r0.x; g.p[1] = r0.y; g.p[2] = r0.z;
				g.sc[0] = r0.w; g.sc[1] = r1.x; g.sc[2] = r1.y;
				g.q = make_float4(r1.z, r1.w, r2.x, r2.y);
				g.hl = r2.z;
				gidx = (int)__float_as_uint(r2.w);
			}
			else
			{
				gidx = (int)a.geom.vis_seg[slot];
				load_candidate<PACKED, FOV>(a, gidx, g);
			}
		}
		GeomOut go;
		geom_item<VARIANT>(a, cam_vm, cam_pm, s_lvbox, valid, gidx, g, !PACKED && !FOV && a.raw && a.cov3D_precomp == nullptr, go);
		if (valid)
		{
			a.radii[gidx] = go.alive ? go.radius : 0;
			a.geom.vis_list[item] = (uint32_t)gidx;
		}
		rows_store<4>(go.wrow, orec, a.geom.wrec + 4 * (size_t)slab * 64, nv, lane);
		rows_store<3>(go.rrow, orec, a.geom.rec + 3 * (size_t)slab * 64, nv, lane);
		if (a.write_cov3D)
		{
			rows_store<4>(go.stash, orec, (float4 *)a.geom.cov3D + 4 * (size_t)slab * 64, nv, lane);
			const float4 ac[4] = { make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(go.inv_qnorm, 0.f, 0.f, 0.f) };
			rows_store<4>(ac, orec, a.geom.acc + 4 * (size_t)slab * 64, nv, lane);
		}
#pragma unroll
		for (int i = 0; i < 4; i++) wr[i] = go.wrow[i];
		fpos = make_float3(go.rrow[1].z, go.rrow[1].w, go.rrow[2].x);
		fconic = make_float3(go.rrow[0].z, go.rrow[0].w, go.rrow[1].x);
	}
	const uint32_t idf = __float_as_uint(wr[2].x), xy = __float_as_uint(wr[2].z);
	const int idx = (int)(idf & 0x3fffffffu);
	const bool alive = item < V && ((idf >> 30) & 1u) != 0, boxtest = ((idf >> 31) & 1u) != 0;
	const float4 ev = make_float4(wr[0].z, wr[0].w, wr[1].x, wr[1].y);
	const float2 el = make_float2(wr[1].z, wr[1].w);
	const int x0 = (int)(xy & 0xffffu), y0 = (int)(xy >> 16), ow0 = (int)__float_as_uint(wr[2].w);
	const uint32_t tnum = alive ? __float_as_uint(wr[3].x) : 0u;
	const float hl = wr[3].y;
	uint32_t count = 0;
	float lowest = hl, highest = 0;
	bool be_blend = false;
	// ---- count the tiles this splat really lands in (and bump the per-tile counters) ----
	// single-tile splats need no box test (RS rasterizer_impl.cu:99-102); handle them in place
	const bool in_place = alive && tnum == 1 && !boxtest;
	if (in_place)
	{
		bool keep = true;
		const int ti = y0 * a.gx + x0;
		if (FOV)
		{
			uint32_t lb;
			TILE_FILTER(ti, hl + 1, keep, lb);
			// (int(lowest) / int(highest) are all that is used of them: the level's integer part stands for the level)
			if (keep) { const float level = (float)(31 - __clz((int)(lb & 15u))); lowest = level; highest = level; be_blend = (lb & 16u) != 0u; }
		}
		if (keep) { BUMP_TILE(ti); count = 1; }
	}
	// everything else: wave-balanced pair loop
	bool deferred = false;
	{
		uint32_t lvmask = 0; // FOV: bit l = some kept tile has int(level) == l; bit 4 = some kept tile blends
		// Splats with at least a wave's worth of tiles are walked by the whole wave ONE AT A TIME: the owner is
		// wave-uniform (scalar registers), so a step needs no owner search and no shuffles and is ~3x shorter
		// than a step of the mixed loop below. A frame-filling splat is 128 such steps and sits on the
		// kernel's critical path.
		if (alive && !in_place && tnum >= FR_GIANT_TNUM)
		{
			const uint32_t slot = atomicAdd(&s_ng, 1u);
			if (slot < FR_GIANT_MAX) { s_gitem[slot] = item; deferred = true; }
		}
		const bool big = alive && !in_place && !deferred && tnum >= FR_BIG_TNUM;
		if (__ballot(big)) __builtin_amdgcn_s_setprio(3);
		for (unsigned long long bigm = __ballot(big); bigm; bigm &= bigm - 1)
		{
			const int L = __ffsll((long long)bigm) - 1;
			const int ox0 = bcast_i(x0, L), oy0 = bcast_i(y0, L), ow = bcast_i(ow0, L);
			const uint32_t on = (uint32_t)bcast_i((int)tnum, L);
			const float4 oev = make_float4(bcast_f(ev.x, L), bcast_f(ev.y, L), bcast_f(ev.z, L), bcast_f(ev.w, L));
			const float2 oel = make_float2(bcast_f(el.x, L), bcast_f(el.y, L));
			const Obb ob = make_obb(bcast_f(wr[0].x, L), bcast_f(wr[0].y, L), oev, oel);
			uint32_t cnt = 0, bits = 0;
			walk_uniform(ox0, oy0, ow, on, ob, bcast_f(hl, L) + 1, 0u, 64u, cnt, bits);
			if (lane == L) { count = cnt; lvmask = bits; }
		}
		__builtin_amdgcn_s_setprio(0);
		const uint32_t my_n = (alive && !in_place && !big && !deferred) ? tnum : 0u;
		const uint32_t incl = wave_incl_scan_u32(my_n, lane);
		const uint32_t excl = incl - my_n;
		const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		// Every lane leaves what a pair of its splat needs in a 64-byte LDS row (the corner extremes of its box included:
		// make_obb once per splat, not once per pair); a pair then reads its OWNER's row -- four LDS reads, most of them
		// broadcasts -- instead of pulling thirteen registers through ds_bpermute, divides by the rectangle's width with a
		// reciprocal, and evaluates the box test without branches.
		if (total != 0)
		{
			const Obb ob = make_obb(wr[0].x, wr[0].y, ev, el);
			float4 *mine = orec + 4 * lane;
			mine[0] = wr[0];
			mine[1] = wr[1];
			mine[2] = make_float4(ob.vxmin, ob.vxmax, ob.vymin, ob.vymax);
			mine[3] = make_float4(__uint_as_float(xy), __int_as_float(max(ow0, 1)), __uint_as_float(excl), hl + 1);
			FR_WAVE_LDS_SYNC();
		}
		for (uint32_t k = 0; k < total; k += 64)
		{
			const uint32_t j = k + lane;
			const bool valid = j < total;
			const int seg_a = (int)max((long long)excl - (long long)k, 0ll);
			const int seg_b = (int)min((long long)incl - (long long)k, 64ll);
			const int owner = max(pair_owner_scan(s_own + (threadIdx.x & ~63), lane, seg_a, seg_b), 0);
			const float4 *orow = orec + 4 * owner;
			const float4 o3 = orow[3];
			const uint32_t oxy = __float_as_uint(o3.x);
			const int ow = __float_as_int(o3.y);
			const uint32_t local = (valid ? j : total - 1) - __float_as_uint(o3.z);
			// local / ow and local % ow for local < 2^23 (a splat of this loop has fewer than 64 tiles)
			int qy = (int)(((float)local + 0.5f) * __builtin_amdgcn_rcpf((float)ow));
			int rx = (int)local - qy * ow;
			if (rx < 0) { qy--; rx += ow; } else if (rx >= ow) { qy++; rx -= ow; }
			const int x = (int)(oxy & 0xffffu) + rx, y = (int)(oxy >> 16) + qy;
			const int ti = y * a.gx + x;
			bool pass = valid;
			uint32_t m = 0;
			if (CULL)
			{
				const float4 o0 = orow[0], o1 = orow[1], o2 = orow[2];
				uint32_t lb = 0;
				if (FOV)
				{
					bool lp;
					TILE_FILTER(valid ? ti : 0, o3.w, lp, lb);
					pass = pass && lp;
				}
				{
					// obb_hits_tile() without its early returns (same expressions, same comparisons: a NaN fails no test)
					const float tpx = (float)x * (float)FR_TILE + (float)FR_TILE / 2.0f, tpy = (float)y * (float)FR_TILE + (float)FR_TILE / 2.0f;
					const bool cx_ok = !((o2.y - tpx) < -8.0f || (o2.x - tpx) > 8.0f);
					const bool cy_ok = !((o2.w - tpy) < -8.0f || (o2.z - tpy) > 8.0f);
					const float xp = tpx + 8.0f - o0.x, xm = tpx - 8.0f - o0.x, yp = tpy + 8.0f - o0.y, ym = tpy - 8.0f - o0.y;
					const float a1p = xp * o0.z, a1m = xm * o0.z, b1p = yp * o0.w, b1m = ym * o0.w;
					const float mn1 = fminf(a1p, a1m) + fminf(b1p, b1m), mx1 = fmaxf(a1p, a1m) + fmaxf(b1p, b1m);
					const bool e1_ok = !(o1.z < mn1 || -o1.z > mx1);
					const float a2p = xp * o1.x, a2m = xm * o1.x, b2p = yp * o1.y, b2m = ym * o1.y;
					const float mn2 = fminf(a2p, a2m) + fminf(b2p, b2m), mx2 = fmaxf(a2p, a2m) + fmaxf(b2p, b2m);
					const bool e2_ok = !(o1.w < mn2 || -o1.w > mx2);
					pass = pass && cx_ok && cy_ok && e1_ok && e2_ok;
				}
				if (FOV) m = pass ? lb : 0u;
			}
			if (pass) BUMP_TILE(ti);
			// hand the results back to the owners: my pairs of this step are lanes [seg_a, seg_b)
			const unsigned long long mine = seg_mask(seg_a, seg_b);
			count += (uint32_t)__popcll(__ballot(pass) & mine);
			if (FOV)
			{
#pragma unroll
				for (int bit = 0; bit < 5; bit++)
					if (__ballot((m >> bit) & 1u) & mine) lvmask |= 1u << bit;
			}
		}
		if (total != 0)
		{
			// the rows are rewritten by the next slab: everyone is done reading
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
		}
		if (FOV && (my_n != 0 || big) && count != 0) range_from_mask(lvmask, lowest, highest, be_blend);
	}
	// what the item leaves: one dense word (a giant splat's is written after the workgroup has walked it); candidates whose
	// tiles were all rejected lose their radius (RS rasterizer_impl.cu:141-145)
	const uint32_t lr = (alive && !deferred) ? range_word(count, lowest, highest, be_blend) : FR_ITEM_NONE;
	if (item < V && !deferred) a.geom.lrange[item] = lr;
	if (alive && !deferred && count == 0) a.radii[idx] = 0;
	// ---- the colours of the slab's items that landed in a tile; the blend record leaves with them (RF: the level rows) ----
	if (__any(lr != FR_ITEM_NONE))
	{
		// (the record this wave made above, put together again from what it kept: not read back through the cache)
		float4 r[3], lv[FR_FOV_LEVELS];
		r[0] = make_float4(wr[0].x, wr[0].y, fconic.x, fconic.y);
		r[1] = make_float4(fconic.z, LEVELCOL ? hl : 0.0f, fpos.x, fpos.y);
		r[2] = make_float4(fpos.z, wr[2].y, (FOV && !LEVELCOL) ? hl : 0.0f, __int_as_float(idx));
#pragma unroll
		for (int l = 0; l < FR_FOV_LEVELS; l++) lv[l] = make_float4(0, 0, 0, 0);
		if (lr != FR_ITEM_NONE) colour_item<VARIANT, PACKED>(a, rows_ok, lr, r, lv);
		if (LEVELCOL) rows_store<4>(lv, orec, a.geom.lvl + 4 * (size_t)slab * 64, nv, lane);
		else rows_store<3>(r, orec, a.geom.rec + 3 * (size_t)slab * 64, nv, lane);
	}
	} // slab loop
	// ---- giant splats: every wave of the workgroup takes every (FR_BIN_THREADS / 64)-th step ----
	__syncthreads();
	const int ng = min((int)s_ng, FR_GIANT_MAX);
	for (int g = 0; g < ng; g++)
	{
		const float4 *wr = a.geom.wrec + 4 * (size_t)s_gitem[g];
		const float4 w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
		const Obb ob = make_obb(w0.x, w0.y, make_float4(w0.z, w0.w, w1.x, w1.y), make_float2(w1.z, w1.w));
		const uint32_t xy = __float_as_uint(w2.z);
		uint32_t cnt = 0, bits = 0;
		walk_uniform((int)(xy & 0xffffu), (int)(xy >> 16), (int)__float_as_uint(w2.w), __float_as_uint(w3.x), ob, w3.y + 1,
			(uint32_t)(threadIdx.x & ~63u), (uint32_t)FR_BIN_THREADS, cnt, bits);
		if (lane == 0) { atomicAdd(&s_gcount[g], cnt); if (FOV) atomicOr(&s_gmask[g], bits); }
	}
	__syncthreads();
	if ((int)threadIdx.x < ng)
	{
		const int gitem = s_gitem[threadIdx.x];
		const uint32_t gcount = s_gcount[threadIdx.x];
		const float4 w2 = a.geom.wrec[4 * (size_t)gitem + 2], w3 = a.geom.wrec[4 * (size_t)gitem + 3];
		float lowest = w3.y, highest = 0.0f;
		bool be_blend = false;
		if (FOV && gcount != 0) range_from_mask(s_gmask[threadIdx.x], lowest, highest, be_blend);
		const uint32_t lr = range_word(gcount, lowest, highest, be_blend);
		a.geom.lrange[gitem] = lr;
		if (gcount == 0) a.radii[__float_as_uint(w2.x) & 0x3fffffffu] = 0;
		else
		{
			float4 r[3], lv[FR_FOV_LEVELS];
#pragma unroll
			for (int i = 0; i < 3; i++) r[i] = a.geom.rec[3 * (size_t)gitem + i];
#pragma unroll
			for (int l = 0; l < FR_FOV_LEVELS; l++) lv[l] = make_float4(0, 0, 0, 0);
			colour_item<VARIANT, PACKED>(a, rows_ok, lr, r, lv);
			if (LEVELCOL)
			{
#pragma unroll
				for (int l = 0; l < FR_FOV_LEVELS; l++) a.geom.lvl[(size_t)gitem * FR_FOV_LEVELS + l] = lv[l];
			}
			else { a.geom.rec[3 * (size_t)gitem + 1] = r[1]; a.geom.rec[3 * (size_t)gitem + 2] = r[2]; }
		}
	}
	if (LDSH == 1 && a.regions)
	{
		// Region-major emission (k_emit_regions): the workgroup lists its items BY SCREEN REGION (8 x 8 tiles) -- an item once per region
		// its walk rectangle reaches -- in a segment of its own, sorted by region: a counting pass over its items' dense words and walk
		// records (written a moment ago: they come from L2), a scan over the regions, an appending pass. k_emit_regions then has
		// workgroups that own a region's tile buckets, where k_emit's own a share of every tile's.
		__shared__ uint32_t s_rneed[FR_MAX_REGIONS], s_roff[FR_MAX_REGIONS], s_rsum[4], s_rok;
		constexpr int BW_ = FR_BIN_THREADS / 64;
		const int R = a.regions, RX = a.region_rx;
		__syncthreads(); // (the giant splats' dense words are in)
		for (int t = threadIdx.x; t < R; t += FR_BIN_THREADS) s_rneed[t] = 0;
		__syncthreads();
		auto each_region = [&](auto &&fn) __attribute__((always_inline))
		{
			for (int q = (int)(threadIdx.x >> 6); ; q += BW_)
			{
				const int slab = (int)blockIdx.x * BW_ + q % BW_ + (q / BW_) * nwaves;
				if (slab >= nslabs) break;
				const int item = slab * 64 + lane;
				if (item >= V || a.geom.lrange[item] == FR_ITEM_NONE) continue;
				const float4 w2 = a.geom.wrec[4 * (size_t)item + 2];
				const uint32_t xy = __float_as_uint(w2.z);
				const int x0 = (int)(xy & 0xffffu), y0 = (int)(xy >> 16), w = max((int)__float_as_uint(w2.w), 1);
				const int h = max((int)(__float_as_uint(a.geom.wrec[4 * (size_t)item + 3].x) / (uint32_t)w), 1);
				const int rx0 = x0 / FR_REGION_TILES, rx1 = (x0 + w - 1) / FR_REGION_TILES, ry0 = y0 / FR_REGION_TILES, ry1 = (y0 + h - 1) / FR_REGION_TILES;
				for (int ry = ry0; ry <= ry1; ry++)
					for (int rx = rx0; rx <= rx1; rx++) fn(item, ry * RX + rx);
			}
		};
		each_region([&](const int, const int r) { atomicAdd(&s_rneed[r], 1u); });
		__syncthreads();
		{
			const uint32_t v = (int)threadIdx.x < R ? s_rneed[threadIdx.x] : 0u; // (R <= 256 <= FR_BIN_THREADS)
			const uint32_t incl = wave_incl_scan_u32(v, lane);
			if (threadIdx.x < 256 && lane == 63) s_rsum[threadIdx.x >> 6] = incl;
			__syncthreads();
			uint32_t off = 0, total = 0;
#pragma unroll
			for (int w = 0; w < 4; w++) { const uint32_t ws = s_rsum[w]; if (w < (int)(threadIdx.x >> 6)) off += ws; total += ws; }
			if ((int)threadIdx.x < R)
			{
				s_roff[threadIdx.x] = off + incl - v;
				s_rneed[threadIdx.x] = 0; // (the appending pass's cursors)
				a.geom.rtab[(size_t)blockIdx.x * FR_MAX_REGIONS + threadIdx.x] = make_uint2(off + incl - v, total <= (uint32_t)a.wcap ? v : 0u);
				if (v != 0 && total <= (uint32_t)a.wcap) atomicAdd(a.geom.rtotal + threadIdx.x, v); // (the region's list length: k_emit_regions' chunks)
			}
			if (threadIdx.x == 0)
			{
				s_rok = total <= (uint32_t)a.wcap ? 1u : 0u;
				if (total > (uint32_t)a.wcap) atomicOr(a.geom.slab_ctr + 5, 1u); // the segment is too small: the frame is emitted the other way
			}
		}
		__syncthreads();
		if (s_rok)
		{
			uint32_t *const seg = a.geom.wlist + (size_t)blockIdx.x * a.wcap;
			each_region([&](const int item, const int r) { seg[s_roff[r] + atomicAdd(&s_rneed[r], 1u)] = (uint32_t)item; });
		}
	}
	if (LDSH)
	{
		// The workgroup's share of every tile's bucket: its histogram is ADDED to the tile's global counter, and what the
		// counter held before is where the share starts inside the bucket (k_emit's cursors). One returning atomic per
		// (workgroup, tile it touched) -- ~0.8 M per 1080p frame, issued in bulk at the end of workgroups that finish at
		// different times -- instead of storing all 256 x 8160 histogram rows and scanning them column by column in a kernel
		// of its own (k_hist_colscan, 12.5 us of a frame, most of it launch and latency). The order of the shares inside a
		// bucket is whatever order the atomics arrive in; the per-tile sort does not care.
		__syncthreads();
		uint32_t *out = a.hist + (size_t)blockIdx.x * a.T;
		for (int t0 = threadIdx.x; t0 < a.T; t0 += 4 * FR_BIN_THREADS)
		{
			uint32_t h[4], o[4];
#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				const int t = t0 + k * FR_BIN_THREADS;
				h[k] = t < a.T ? (LDSH == 2 ? (lds_hist[t >> 1] >> (16 * (t & 1))) & 0xffffu : lds_hist[t]) : 0u;
			}
#pragma unroll
			for (int k = 0; k < 4; k++) o[k] = h[k] ? atomicAdd(&a.tile_count[t0 + k * FR_BIN_THREADS], h[k]) : 0u;
#pragma unroll
			for (int k = 0; k < 4; k++) { const int t = t0 + k * FR_BIN_THREADS; if (t < a.T && h[k]) out[t] = o[k]; }
		}
		// The tile scan as the tail of this kernel (round 6; a kernel of its own it was 15 us + a 9-us launch gap of pure latency on
		// the frame's critical path): the LAST workgroup to get here -- found with a counter, as in k_project -- has every
		// workgroup's share in the tile counters (the shares left as RETURNING device-scope atomics: performed at the memory side
		// before their values came back and were stored above, so no fence is needed; the counts are read back with device-scope
		// loads) and scans them with its first eight waves in the LDS the histogram and the staging rows lived in; the other
		// four waves leave (a finished wave no longer counts at the workgroup barrier).
		if (LDSH == 1 && a.fuse_scan)
		{
			__shared__ uint32_t s_last_wg;
			__syncthreads();
			if (threadIdx.x == 0) s_last_wg = atomicAdd(a.geom.slab_ctr + 4, 1u) == gridDim.x - 1 ? 1u : 0u;
			__syncthreads();
			if (!s_last_wg || threadIdx.x >= FR_TILE_SCAN_THREADS) return;
			tile_scan_body<FR_TILE_SCAN_THREADS, true>(a.ts, lds_hist);
			if (a.regions)
			{
				// chunks of every region's list (k_emit_regions pulls them from a counter): rchunk[r] = first chunk of region r, [R] = all
				__syncthreads();
				const int t = (int)threadIdx.x;
				const uint32_t nr = t < a.regions ? __hip_atomic_load(a.geom.rtotal + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
				const uint32_t ch = (nr + FR_ER_CHUNK - 1) / FR_ER_CHUNK;
				const uint32_t incl = wave_incl_scan_u32(ch, lane);
				if (t < 256 && lane == 63) lds_hist[t >> 6] = incl;
				__syncthreads();
				uint32_t off = 0, tot = 0;
#pragma unroll
				for (int w = 0; w < 4; w++) { const uint32_t ws = lds_hist[w]; if (w < (t >> 6)) off += ws; tot += ws; }
				if (t < a.regions) a.geom.rchunk[t] = off + incl - ch;
				if (t == 0) a.geom.rchunk[a.regions] = tot;
			}
		}
	}
}
#undef BUMP_TILE
#undef TILE_FILTER

// One thread per Gaussian: re-walk the rect, repeat the cull test and append (depth,id) to the
// tile's bucket through the tile cursor. Order inside a bucket is arbitrary; the per-tile sort
// on (depth bits, id) restores the reference's stable order. Big rects: whole wave, as above.
struct EmitArgs {
	int P, gx, gy, T;
	const int *radii;
	GeomWS geom;
	const float *highest_levels;
	const float *tile_lv;
	const uint32_t *lv_bbox;
	int lds_tiles;
	const uint2 *ranges;
	uint32_t *cursor;
	uint64_t *entries;
	const uint32_t *hist; // [blocks][T] exclusive prefix over workgroups (LDSH)
	const uint32_t *totals; // ImageWS::totals: [0] = the frame's number of instances (k_tile_scan)
	uint32_t capacity;      // instances `entries` has room for ...
	uint32_t items_cap;     // ... and blend work items the frame's blend grid covers (see fr_forward)
};
// LDSH: the workgroup's write cursor of every tile lives in LDS, initialised to
// tile start + (instances of the same tile owned by lower-numbered workgroups).
#ifndef FR_EMIT_THREADS
#define FR_EMIT_THREADS FR_BIN_THREADS // (a workgroup takes the slabs k_bin's workgroup of the same number took; its waves share them out)
#endif
// LDSH == 2 (see k_bin): the LDS cursors are 16-bit offsets inside the workgroup's share, two tiles per word; where the share
// starts comes from global memory with every entry
#ifdef FR_EMIT_NOSTORE
#define FR_EMIT_ST(slot, v) do { if ((slot) == 0xffffffffu) a.entries[0] = (v); } while (0) // (timing experiment: the cursors are bumped, nothing is stored)
#elif defined(FR_EMIT_COALESCED)
#define FR_EMIT_ST(slot, v) do { (void)(slot); a.entries[(size_t)blockIdx.x * 20000 + (threadIdx.x & ~63u) * 26u + ((uint32_t)tm_seq++ * 64u + lane) % 1600u] = (v); } while (0) // (timing experiment: a wave's entries to consecutive slots of its own)
#else
#define FR_EMIT_ST(slot, v) a.entries[(slot)] = (v)
#endif
#define NEXT_SLOT(ti) (LDSH == 2 ? a.ranges[(ti)].x + pre_row[(ti)] + ((atomicAdd(&lds_cur[(ti) >> 1], 1u << (16 * ((ti) & 1))) >> (16 * ((ti) & 1))) & 0xffffu) \
	: LDSH ? atomicAdd(&lds_cur[(ti)], 1u) : a.ranges[(ti)].x + atomicAdd(&a.cursor[(ti)], 1u))
template <int VARIANT, int LDSH>
__global__ void __launch_bounds__(FR_EMIT_THREADS) k_emit(const EmitArgs a)
{
	constexpr bool CULL = VARIANT != FR_VARIANT_ORIGINAL;
	constexpr bool FOV = VARIANT == FR_VARIANT_FOV_PCHECK_OBB;
	extern __shared__ __attribute__((aligned(16))) uint32_t lds_cur[];
#ifdef FR_EMIT_TIMERS
	const uint64_t tm_entry = wall_clock64(); uint64_t tm_pro = 0, tm_loop = 0, tm_walk = 0, tm_big = 0, tm_drain = 0, tm_pre = 0, tm_mid = 0, tm_ta = 0; int tm_slabs = 0, tm_steps = 0;
#endif
	// launched before the host knows the frame's instance count (fr_forward): when the binning workspace turns out too
	// small nothing is emitted, the host replays the stage with a larger one
	if (a.totals[0] > a.capacity || a.totals[5] > a.items_cap) return;
	const int lane = threadIdx.x & 63;
#ifdef FR_EMIT_COALESCED
	uint32_t tm_seq = 0;
#endif
	// RF: the tiles' levels as k_bin keeps them: four bits per tile, min(max(int(tile_min), 0), 7) (the filter compares
	// with an integer; exact for highest levels 0..3, k_project says if it saw another: then the floats in global memory)
	const int cur_words = LDSH == 2 ? (a.T + 1) / 2 : (LDSH ? a.T : 0);
	const uint32_t *const pre_row = a.hist + (LDSH ? (size_t)blockIdx.x * a.T : 0); // where this workgroup's share of every tile's bucket starts
	uint32_t *lds_tab = lds_cur + cur_words;
	const int tab_words = (a.T + 7) / 8;
	const bool ldst = FOV && a.lds_tiles && a.geom.slab_ctr[3] == 0u;
	if (ldst)
	{
		const float *gmin = a.tile_lv + a.T;
		for (int t0 = threadIdx.x; t0 < a.T; t0 += 16 * FR_EMIT_THREADS) // (sixteen loads at a time, see k_bin)
		{
			float v[16];
#pragma unroll
			for (int k = 0; k < 16; k++) v[k] = gmin[min(t0 + k * FR_EMIT_THREADS, a.T - 1)];
#pragma unroll
			for (int k = 0; k < 16; k++)
			{
				const int t = t0 + k * FR_EMIT_THREADS;
				uint32_t nib = t < a.T ? (v[k] == v[k] ? (uint32_t)min(max(f2i(v[k]), 0), 7) : 7u) : 0u; // (NaN level: no Gaussian passes, see k_bin)
				nib <<= 4 * (lane & 7);
				nib |= (uint32_t)__shfl_xor((int)nib, 1); nib |= (uint32_t)__shfl_xor((int)nib, 2); nib |= (uint32_t)__shfl_xor((int)nib, 4);
				if ((lane & 7) == 0 && (t >> 3) < tab_words) lds_tab[t >> 3] = nib;
			}
		}
	}
#define TILE_PASSES(ti, olim) (ldst ? (float)((lds_tab[(ti) >> 3] >> (4 * ((ti) & 7))) & 7u) < (olim) : tile_min[(ti)] < (olim))
	if (LDSH == 2)
		for (int t = threadIdx.x; t < cur_words; t += FR_EMIT_THREADS) lds_cur[t] = 0;
	else if (LDSH)
	{
		// (pre[t] is only defined for the tiles this workgroup counted instances in -- the only cursors it will use)
		const uint32_t *pre = pre_row;
		for (int t0 = threadIdx.x; t0 < a.T; t0 += 16 * FR_EMIT_THREADS)
		{
			uint32_t st[16], pr[16];
#pragma unroll
			for (int k = 0; k < 16; k++) { const int t = min(t0 + k * FR_EMIT_THREADS, a.T - 1); st[k] = a.ranges[t].x; pr[k] = pre[t]; }
#pragma unroll
			for (int k = 0; k < 16; k++) if (t0 + k * FR_EMIT_THREADS < a.T) lds_cur[t0 + k * FR_EMIT_THREADS] = st[k] + pr[k];
		}
	}
	__syncthreads();
	__shared__ int s_own[FR_EMIT_THREADS];
	// (dynamic LDS behind the cursors and the table, 16-byte aligned: the pair loop's 64-byte owner row per lane, see k_bin)
	float4 *const s_orec = (float4 *)(lds_cur + ((cur_words + (ldst ? tab_words : 0) + 3) & ~3));
	__shared__ int s_gidx[FR_GIANT_MAX]; // vis_list positions of the giant splats, walked by the whole workgroup at the end (see k_bin)
	__shared__ uint32_t s_ng, s_next_slab;
	if (threadIdx.x == 0) { s_ng = 0; s_next_slab = 0; }
	__syncthreads();
#ifdef FR_EMIT_TIMERS
	tm_pro = wall_clock64() - tm_entry;
#endif
	const float *tile_min = FOV ? a.tile_lv + a.T : nullptr;
	auto walk_uniform = [&](const int ox0, const int oy0, const int ow, const uint32_t on, const Obb &ob, const float olim,
		const uint64_t opay, const uint32_t k0, const uint32_t kstep) __attribute__((always_inline))
	{
		const float rw = 1.0f / (float)ow;
		for (uint32_t k = k0; k < on; k += kstep)
		{
			const uint32_t j = k + lane;
			const bool valid = j < on;
			int ry = (int)(((float)j + 0.5f) * rw);
			int rx = (int)j - ry * ow;
			if (rx < 0) { ry--; rx += ow; } else if (rx >= ow) { ry++; rx -= ow; }
			const int x = ox0 + rx, y = oy0 + ry;
			const int ti = valid ? y * a.gx + x : 0;
			bool pass = valid;
			if (CULL)
			{
				if (FOV) pass = pass && TILE_PASSES(ti, olim);
				pass = pass && obb_hits_tile(ob, x, y);
			}
			if (pass) FR_EMIT_ST(NEXT_SLOT(ti), opay);
		}
	};
	const int V = (int)a.geom.slab_ctr[1];
	const int nslabs = (V + 63) / 64; // wave-sized slabs, as in k_bin
	// a slab's inputs: every lane's walk record and what k_bin found out about the item (fetched one slab ahead, see the loop below)
	struct SlabIn { float4 w0, w1, w2, w3; uint32_t lr; };
	auto fetch_slab = [&](const int slab, SlabIn &in) __attribute__((always_inline))
	{
		const int item = min(slab * 64 + lane, max(V - 1, 0)); // (clamped, not predicated: the loads go out together)
		const float4 *wr = a.geom.wrec + 4 * (size_t)item;
		in.lr = a.geom.lrange[item];
		in.w0 = wr[0]; in.w1 = wr[1]; in.w2 = wr[2]; in.w3 = wr[3];
	};
	auto process = [&](const int slab, const SlabIn &in) __attribute__((always_inline))
	{
	const int item = slab * 64 + lane;
	// everything k_emit needs about the entry sits in its walk record (one coalesced 64-byte read per lane instead of
	// a chain of dependent gathers through the Gaussian index)
	int idx = 0;
	bool alive = false, boxtest = false;
	int x0 = 0, y0 = 0, x1 = 0;
	uint32_t tnum = 0;
	float cx = 0.f, cy = 0.f, hl = 0.f;
	uint32_t depth_bits = 0;
	float4 ev = make_float4(0, 0, 0, 0);
	float2 el = make_float2(0, 0);
	if (item < V)
	{
		// (what k_bin found out about the item: one that landed in no tile is not walked again)
		const uint32_t lr = in.lr;
		const float4 w0 = in.w0, w1 = in.w1, w2 = in.w2, w3 = in.w3;
		const uint32_t idf = __float_as_uint(w2.x), xy = __float_as_uint(w2.z);
		idx = (int)(idf & 0x3fffffffu);
		alive = lr != FR_ITEM_NONE && ((idf >> 30) & 1u) != 0; boxtest = (idf >> 31) & 1u;
		cx = w0.x; cy = w0.y; ev = make_float4(w0.z, w0.w, w1.x, w1.y); el = make_float2(w1.z, w1.w);
		depth_bits = __float_as_uint(w2.y);
		x0 = (int)(xy & 0xffffu); y0 = (int)(xy >> 16); x1 = x0 + (int)__float_as_uint(w2.w);
		tnum = alive ? __float_as_uint(w3.x) : 0u;
		hl = w3.y;
	}
	// the instance's payload is the ITEM (position in vis_list): the per-item records are dense, and items are in index order,
	// so the per-tile sort by (depth bits, item) gives the reference's stable order
	const uint64_t payload = ((uint64_t)depth_bits << 32) | (uint32_t)item;
	const bool in_place = alive && tnum == 1 && !boxtest; // survived k_bin's level test, no box test (single-tile splat)
	if (in_place) FR_EMIT_ST(NEXT_SLOT(y0 * a.gx + x0), payload);
	{
		bool deferred = false;
		if (alive && !in_place && tnum >= FR_GIANT_TNUM)
		{
			const uint32_t slot = atomicAdd(&s_ng, 1u);
			if (slot < FR_GIANT_MAX) { s_gidx[slot] = item; deferred = true; }
		}
		// big splats: whole wave, wave-uniform owner (see k_bin)
		const bool big = alive && !in_place && !deferred && tnum >= FR_BIG_TNUM;
#ifdef FR_EMIT_TIMERS
		const uint64_t tg0 = wall_clock64(); tm_pre += tg0 - tm_ta;
#endif
		for (unsigned long long bigm = __ballot(big); bigm; bigm &= bigm - 1)
		{
			const int L = __ffsll((long long)bigm) - 1;
			const int ox0 = bcast_i(x0, L), oy0 = bcast_i(y0, L), ow = bcast_i(x1, L) - ox0;
			const float4 oev = make_float4(bcast_f(ev.x, L), bcast_f(ev.y, L), bcast_f(ev.z, L), bcast_f(ev.w, L));
			const float2 oel = make_float2(bcast_f(el.x, L), bcast_f(el.y, L));
			const Obb ob = make_obb(bcast_f(cx, L), bcast_f(cy, L), oev, oel);
			const uint64_t opay = ((uint64_t)(uint32_t)bcast_i((int)depth_bits, L) << 32) | (uint32_t)(slab * 64 + L);
			walk_uniform(ox0, oy0, ow, (uint32_t)bcast_i((int)tnum, L), ob, bcast_f(hl, L) + 1, opay, 0u, 64u);
		}
#ifdef FR_EMIT_TIMERS
		const uint64_t tg1 = wall_clock64(); tm_big += tg1 - tg0;
#endif
		const uint32_t my_n = (alive && !in_place && !big && !deferred) ? tnum : 0u;
		const uint32_t incl = wave_incl_scan_u32(my_n, lane);
		const uint32_t excl = incl - my_n;
		const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		// every lane leaves what a pair of its splat needs in a 64-byte LDS row (k_bin's pair loop: the corner extremes of its box
		// included -- make_obb once per splat, not once per pair); a pair reads its OWNER's row -- four LDS reads, most of them
		// broadcasts -- divides by the rectangle's width with a reciprocal and evaluates the box test without branches (this loop
		// pulled thirteen registers through ds_bpermute, divided twice and rebuilt the box per pair: 60 of the kernel's 104 us)
		float4 *const orec = s_orec + 4 * (threadIdx.x & ~63);
#ifdef FR_EMIT_TIMERS
		const uint64_t tw0 = wall_clock64(); tm_steps += (int)((total + 63) / 64); tm_mid += tw0 - tg1;
#endif
		if (total != 0)
		{
			const Obb ob = make_obb(cx, cy, ev, el);
			float4 *mine = orec + 4 * lane;
			mine[0] = make_float4(cx, cy, ev.x, ev.y);
			mine[1] = make_float4(ev.z, ev.w, el.x, el.y);
			mine[2] = make_float4(ob.vxmin, ob.vxmax, ob.vymin, ob.vymax);
			mine[3] = make_float4(__uint_as_float((uint32_t)x0 | ((uint32_t)y0 << 16)), __int_as_float(max(x1 - x0, 1)), __uint_as_float(excl), hl + 1);
			FR_WAVE_LDS_SYNC();
		}
		for (uint32_t k = 0; k < total; k += 64)
		{
			const uint32_t j = k + lane;
			const bool valid = j < total;
			const int seg_a = (int)max((long long)excl - (long long)k, 0ll);
			const int seg_b = (int)min((long long)incl - (long long)k, 64ll);
			const int owner = max(pair_owner_scan(s_own + (threadIdx.x & ~63), lane, seg_a, seg_b), 0);
			const float4 *orow = orec + 4 * owner;
			const float4 o3 = orow[3];
			const uint32_t oxy = __float_as_uint(o3.x);
			const int ow = __float_as_int(o3.y);
			const uint32_t local = (valid ? j : total - 1) - __float_as_uint(o3.z);
			// local / ow and local % ow for local < 2^23 (a splat of this loop has fewer than 64 tiles)
			int qy = (int)(((float)local + 0.5f) * __builtin_amdgcn_rcpf((float)ow));
			int rx = (int)local - qy * ow;
			if (rx < 0) { qy--; rx += ow; } else if (rx >= ow) { qy++; rx -= ow; }
			const int x = (int)(oxy & 0xffffu) + rx, y = (int)(oxy >> 16) + qy;
			const int ti = y * a.gx + x;
			const uint64_t opay = ((uint64_t)(uint32_t)__shfl((int)depth_bits, owner) << 32) | (uint32_t)(slab * 64 + owner);
			bool pass = valid;
			if (CULL)
			{
				const float4 o0 = orow[0], o1 = orow[1], o2 = orow[2];
				if (FOV) pass = pass && TILE_PASSES(valid ? ti : 0, o3.w);
				{
					// obb_hits_tile() without its early returns (same expressions, same comparisons: a NaN fails no test)
					const float tpx = (float)x * (float)FR_TILE + (float)FR_TILE / 2.0f, tpy = (float)y * (float)FR_TILE + (float)FR_TILE / 2.0f;
					const bool cx_ok = !((o2.y - tpx) < -8.0f || (o2.x - tpx) > 8.0f);
					const bool cy_ok = !((o2.w - tpy) < -8.0f || (o2.z - tpy) > 8.0f);
					const float xp = tpx + 8.0f - o0.x, xm = tpx - 8.0f - o0.x, yp = tpy + 8.0f - o0.y, ym = tpy - 8.0f - o0.y;
					const float a1p = xp * o0.z, a1m = xm * o0.z, b1p = yp * o0.w, b1m = ym * o0.w;
					const float mn1 = fminf(a1p, a1m) + fminf(b1p, b1m), mx1 = fmaxf(a1p, a1m) + fmaxf(b1p, b1m);
					const bool e1_ok = !(o1.z < mn1 || -o1.z > mx1);
					const float a2p = xp * o1.x, a2m = xm * o1.x, b2p = yp * o1.y, b2m = ym * o1.y;
					const float mn2 = fminf(a2p, a2m) + fminf(b2p, b2m), mx2 = fmaxf(a2p, a2m) + fmaxf(b2p, b2m);
					const bool e2_ok = !(o1.w < mn2 || -o1.w > mx2);
					pass = pass && cx_ok && cy_ok && e1_ok && e2_ok;
				}
			}
			if (pass) FR_EMIT_ST(NEXT_SLOT(ti), opay);
		}
		if (total != 0)
		{
			// the rows are rewritten by the next slab: everyone is done reading
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
		}
#ifdef FR_EMIT_TIMERS
		tm_walk += wall_clock64() - tw0;
#endif
	}
	}; // process(slab)
	{
		// the slabs k_bin's wave of the same number took (the bucket offsets are per workgroup): wave, wave + waves, ...
		constexpr int BW = FR_BIN_THREADS / 64;
		const int bin_waves = (int)gridDim.x * BW;
		// the workgroup's s-th slab: k_bin's wave (s mod BW) of this workgroup took it in its round s / BW
		auto slab_of = [&](const int s) { return (int)blockIdx.x * BW + s % BW + (s / BW) * bin_waves; };
		// one slab AHEAD: the next slab's records are in flight while this one's pairs are walked (the kernel's waves spent 73 %
		// of their time parked on s_waitcnt: a slab's record loads in front of its walk, its stores behind)
		// (which wave takes the workgroup's q-th slab: whoever asks first -- a counter in LDS, as in k_bin)
		auto grab = [&]() -> int {
			uint32_t q_ = 0;
			if (lane == 0) q_ = atomicAdd(&s_next_slab, 1u);
			return slab_of(__builtin_amdgcn_readfirstlane((int)q_)); // (grows with q)
		};
		SlabIn cur, nxt;
		int slab = grab();
		if (slab < nslabs) fetch_slab(slab, cur);
#ifdef FR_EMIT_TIMERS
		const uint64_t tl0 = wall_clock64();
#endif
		for (; slab < nslabs; )
		{
#ifdef FR_EMIT_TIMERS
			tm_ta = wall_clock64();
#endif
			const int next = grab();
			const bool more = next < nslabs;
			if (more) fetch_slab(next, nxt);
			process(slab, cur);
#ifdef FR_EMIT_TIMERS
			{ const uint64_t td0 = wall_clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tm_drain += wall_clock64() - td0; }
#endif
			if (more) cur = nxt;
			slab = next;
#ifdef FR_EMIT_TIMERS
			tm_slabs++;
#endif
		}
#ifdef FR_EMIT_TIMERS
		tm_loop = wall_clock64() - tl0;
#endif
	}
#ifdef FR_EMIT_TIMERS
	const uint64_t tb0 = wall_clock64();
#endif
	__syncthreads();
#ifdef FR_EMIT_TIMERS
	if (lane == 0)
	{
		// developer build: per-wave (total so far, prologue, slab loop, wait at the barrier, slabs, pair steps) in 10-ns ticks, in the
		// covariance rows the inference variants do not use
		float *d = a.geom.cov3D + (size_t)((int)blockIdx.x * (FR_EMIT_THREADS / 64) + (int)(threadIdx.x >> 6)) * 8;
		d[0] = (float)(wall_clock64() - tm_entry); d[1] = (float)tm_pro; d[2] = (float)tm_loop; d[3] = (float)(wall_clock64() - tb0);
		d[4] = (float)tm_slabs; d[5] = (float)tm_steps; d[6] = (float)tm_walk; d[7] = (float)tm_big; d[3] = (float)tm_drain; d[1] = (float)tm_pre; d[4] = (float)tm_mid; // (developer layout: d[1] grab + fetch issue + unpack + single-tile stores, d[3] the wait for the slab's stores / the next records, d[4] scan + owner rows)
	}
#endif
	const int ng = min((int)s_ng, FR_GIANT_MAX);
	for (int g = 0; g < ng; g++)
	{
		const float4 *wr = a.geom.wrec + 4 * (size_t)s_gidx[g];
		const float4 w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
		const Obb ob = make_obb(w0.x, w0.y, make_float4(w0.z, w0.w, w1.x, w1.y), make_float2(w1.z, w1.w));
		const uint32_t xy = __float_as_uint(w2.z);
		const uint64_t opay = ((uint64_t)__float_as_uint(w2.y) << 32) | (uint32_t)s_gidx[g];
		walk_uniform((int)(xy & 0xffffu), (int)(xy >> 16), (int)__float_as_uint(w2.w), __float_as_uint(w3.x), ob, w3.y + 1, opay,
			(uint32_t)(threadIdx.x & ~63u), (uint32_t)FR_EMIT_THREADS);
	}
}
#undef NEXT_SLOT
#undef TILE_PASSES

// ---- region-major emission (round 6) --------------------------------------------------------------------------------------------
// k_emit's workgroups own a SHARE of every tile's bucket: a workgroup contributes 2.4-2.9 entries per tile, 6 M eight-byte stores each
// dirty a partial line (WRITE_SIZE 3.7 x the payload), and with 84 KiB of LDS per workgroup (cursors of all 8160 tiles + the owner rows)
// one workgroup of twelve waves is all a CU holds: the kernel is latency-bound at 2.2 waves per SIMD. Here a workgroup owns a REGION
// (8 x 8 tiles) -- one of G workgroups per region, each taking the lists a group of binning workgroups left for it (GeomWS::rtab /
// wlist, k_bin's tail) -- so its stores fall into 64 buckets whose lines it fills within microseconds; a wave stages the pairs that
// pass in LDS, counts them per tile, reserves its share of the 64 buckets with ONE returning atomic per tile and flush, and writes.
// No table of 8160 cursors, no prologue, no big / giant splats (an item is cut to the region: at most 64 tiles, every item goes through
// the pair loop), ~1000 small workgroups that the hardware places as CUs fall free. The order inside a bucket is arbitrary, as before.
struct EmitRegArgs {
	int gx, gy, T, RX, R, bin_wgs, wcap;
	GeomWS geom;
	const float *tile_lv;
	const uint2 *ranges;
	uint32_t *cursor;       // ImageWS::tile_count: zero behind the tile scan
	uint64_t *entries;
	const uint32_t *totals;
	uint32_t capacity, items_cap;
};
#define FR_ER_THREADS 256
#define FR_ER_WAVES (FR_ER_THREADS / 64)
#define FR_ER_STAGE 384 // pairs a wave stages between two flushes
// The work is handed out in CHUNKS of a region's list (all binning workgroups' lists for the region strung together): GeomWS::rchunk
// (k_bin's last workgroup) says how many chunks every region has, the workgroups pull chunk numbers from one counter -- the regions
// around the gaze hold ten times the entries of those at the rim, and a grid of one workgroup per (region, fixed share) ended with
// its densest regions (243 us; 94 with 8192 small shares).
template <int VARIANT>
__global__ void __launch_bounds__(FR_ER_THREADS) k_emit_regions(const EmitRegArgs a)
{
	constexpr bool CULL = VARIANT != FR_VARIANT_ORIGINAL;
	constexpr bool FOV = VARIANT == FR_VARIANT_FOV_PCHECK_OBB;
	constexpr int RT = FR_REGION_TILES;
	if (a.totals[0] > a.capacity || a.totals[5] > a.items_cap) return;
	__shared__ uint32_t s_rchunk[FR_MAX_REGIONS + 1];
	__shared__ uint32_t s_pre[FR_BIN_BLOCKS + 1], s_src[FR_BIN_BLOCKS], s_part[FR_ER_WAVES], s_chunk;
	__shared__ float s_tmin[RT * RT];
	__shared__ __attribute__((aligned(16))) float4 s_orec[FR_ER_WAVES][64 * 4];
	__shared__ int s_own[FR_ER_WAVES][64];
	__shared__ __attribute__((aligned(8))) uint64_t s_pay[FR_ER_WAVES][FR_ER_STAGE];
	__shared__ uint8_t s_tl[FR_ER_WAVES][FR_ER_STAGE];
	__shared__ uint32_t s_cnt[FR_ER_WAVES][RT * RT], s_base[FR_ER_WAVES][RT * RT];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (int t = threadIdx.x; t <= a.R; t += FR_ER_THREADS) s_rchunk[t] = a.geom.rchunk[t];
	__syncthreads();
	const uint32_t nchunks = s_rchunk[a.R];
	float4 *const orec = s_orec[wv];
	uint64_t *const pay = s_pay[wv];
	uint8_t *const stl = s_tl[wv];
	int cur_r = -1, tx0 = 0, ty0 = 0, n = 0, npre = 1;
	uint32_t n_staged = 0; // (wave-uniform)
#ifdef FR_ER_TIMERS
	const uint64_t tm_start = wall_clock64(); uint64_t tm_pro = 0, tm_fetch = 0, tm_proc = 0, tm_flush = 0; int tm_chunks = 0, tm_flushes = 0, tm_regions = 0;
#define ERT(x) x
#else
#define ERT(x)
#endif
	// a wave's staged pairs leave: counted per tile, the wave's share of every bucket reserved with one returning atomic, written
	auto flush = [&]() __attribute__((always_inline))
	{
		ERT(const uint64_t tf0 = wall_clock64(); tm_flushes++;)
		s_cnt[wv][lane] = 0;
		FR_WAVE_LDS_SYNC();
		uint32_t rank[FR_ER_STAGE / 64];
#pragma unroll
		for (int k = 0; k < FR_ER_STAGE / 64; k++)
		{
			const uint32_t i = (uint32_t)k * 64u + (uint32_t)lane;
			rank[k] = i < n_staged ? atomicAdd(&s_cnt[wv][stl[i]], 1u) : 0u;
		}
		FR_WAVE_LDS_SYNC();
		{
			const uint32_t c = s_cnt[wv][lane];
			const int x = tx0 + (lane & (RT - 1)), y = ty0 + (lane >> 3);
			uint32_t base = 0;
			if (c != 0) { const int ti = y * a.gx + x; base = a.ranges[ti].x + atomicAdd(&a.cursor[ti], c); }
			s_base[wv][lane] = base;
		}
		FR_WAVE_LDS_SYNC();
#pragma unroll
		for (int k = 0; k < FR_ER_STAGE / 64; k++)
		{
			const uint32_t i = (uint32_t)k * 64u + (uint32_t)lane;
			if (i < n_staged) a.entries[s_base[wv][stl[i]] + rank[k]] = pay[i];
		}
		FR_WAVE_LDS_SYNC(); // (the staging rows are written again)
		n_staged = 0;
		ERT(tm_flush += wall_clock64() - tf0;)
	};
	struct SlabIn { int item; float4 w0, w1, w2, w3; bool valid; };
	auto fetch = [&](const int jbase, const int jend, SlabIn &in) __attribute__((always_inline))
	{
		const int j = jbase + lane;
		in.valid = j < jend;
		// which binning workgroup's list entry j comes from: the last b with s_pre[b] <= j
		const uint32_t jj = (uint32_t)min(j, n - 1);
		int lo = 0, hi = npre;
		while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pre[mid] <= jj) lo = mid; else hi = mid; }
		in.item = (int)a.geom.wlist[s_src[lo] + (jj - s_pre[lo])];
		const float4 *wr = a.geom.wrec + 4 * (size_t)in.item;
		in.w0 = wr[0]; in.w1 = wr[1]; in.w2 = wr[2]; in.w3 = wr[3];
	};
	auto process = [&](const SlabIn &in) __attribute__((always_inline))
	{
		const int item = in.item;
		const float4 w0 = in.w0, w1 = in.w1, w2 = in.w2, w3 = in.w3;
		const uint32_t idf = __float_as_uint(w2.x), xy = __float_as_uint(w2.z);
		const bool boxtest = ((idf >> 31) & 1u) != 0;
		const float cx = w0.x, cy = w0.y;
		const float4 ev = make_float4(w0.z, w0.w, w1.x, w1.y);
		const float2 el = make_float2(w1.z, w1.w);
		const uint32_t depth_bits = __float_as_uint(w2.y);
		// the walk rectangle, cut to the region
		const int x0 = (int)(xy & 0xffffu), y0 = (int)(xy >> 16), rw = max((int)__float_as_uint(w2.w), 1);
		const int rh = max((int)(__float_as_uint(w3.x) / (uint32_t)rw), 1);
		const int qx0 = max(x0, tx0), qx1 = min(x0 + rw, min(tx0 + RT, a.gx)), qy0 = max(y0, ty0), qy1 = min(y0 + rh, min(ty0 + RT, a.gy));
		const int qw = max(qx1 - qx0, 0), qh = max(qy1 - qy0, 0);
		const uint32_t my_n = in.valid ? (uint32_t)(qw * qh) : 0u;
		const uint32_t incl = wave_incl_scan_u32(my_n, lane);
		const uint32_t excl = incl - my_n;
		const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		if (total == 0) return;
		{
			const Obb ob = make_obb(cx, cy, ev, el);
			float4 *mine = orec + 4 * lane;
			mine[0] = make_float4(cx, cy, ev.x, ev.y);
			mine[1] = make_float4(ev.z, ev.w, el.x, el.y);
			mine[2] = make_float4(ob.vxmin, ob.vxmax, ob.vymin, ob.vymax);
			// (bit 16 of the width word: the box test applies -- a one-tile rectangle, or a variant without culling, skips it)
			mine[3] = make_float4(__uint_as_float((uint32_t)qx0 | ((uint32_t)qy0 << 16)), __int_as_float(max(qw, 1) | (boxtest ? 0x10000 : 0)), __uint_as_float(excl), w3.y + 1);
			FR_WAVE_LDS_SYNC();
		}
		for (uint32_t k = 0; k < total; k += 64)
		{
			if (n_staged + 64u > (uint32_t)FR_ER_STAGE) flush();
			const uint32_t jp = k + lane;
			const bool pv = jp < total;
			const int seg_a = (int)max((long long)excl - (long long)k, 0ll);
			const int seg_b = (int)min((long long)incl - (long long)k, 64ll);
			const int owner = max(pair_owner_scan(s_own[wv], lane, seg_a, seg_b), 0);
			const float4 *orow = orec + 4 * owner;
			const float4 o3 = orow[3];
			const uint32_t oxy = __float_as_uint(o3.x);
			const int owf = __float_as_int(o3.y), ow = owf & 0xffff;
			const bool obox = (owf & 0x10000) != 0;
			const uint32_t local = (pv ? jp : total - 1) - __float_as_uint(o3.z);
			int qy = (int)(((float)local + 0.5f) * __builtin_amdgcn_rcpf((float)ow));
			int rx = (int)local - qy * ow;
			if (rx < 0) { qy--; rx += ow; } else if (rx >= ow) { qy++; rx -= ow; }
			const int x = (int)(oxy & 0xffffu) + rx, y = (int)(oxy >> 16) + qy;
			const int tl = ((y - ty0) & (RT - 1)) * RT + ((x - tx0) & (RT - 1));
			bool pass = pv;
			if (CULL)
			{
				const float4 o0 = orow[0], o1 = orow[1], o2 = orow[2];
				if (FOV) pass = pass && s_tmin[tl] < o3.w;
				{
					// obb_hits_tile() without its early returns (same expressions, same comparisons: a NaN fails no test)
					const float tpx = (float)x * (float)FR_TILE + (float)FR_TILE / 2.0f, tpy = (float)y * (float)FR_TILE + (float)FR_TILE / 2.0f;
					const bool cx_ok = !((o2.y - tpx) < -8.0f || (o2.x - tpx) > 8.0f);
					const bool cy_ok = !((o2.w - tpy) < -8.0f || (o2.z - tpy) > 8.0f);
					const float xp = tpx + 8.0f - o0.x, xm = tpx - 8.0f - o0.x, yp = tpy + 8.0f - o0.y, ym = tpy - 8.0f - o0.y;
					const float a1p = xp * o0.z, a1m = xm * o0.z, b1p = yp * o0.w, b1m = ym * o0.w;
					const float mn1 = fminf(a1p, a1m) + fminf(b1p, b1m), mx1 = fmaxf(a1p, a1m) + fmaxf(b1p, b1m);
					const bool e1_ok = !(o1.z < mn1 || -o1.z > mx1);
					const float a2p = xp * o1.x, a2m = xm * o1.x, b2p = yp * o1.y, b2m = ym * o1.y;
					const float mn2 = fminf(a2p, a2m) + fminf(b2p, b2m), mx2 = fmaxf(a2p, a2m) + fmaxf(b2p, b2m);
					const bool e2_ok = !(o1.w < mn2 || -o1.w > mx2);
					pass = pass && (!obox || (cx_ok && cy_ok && e1_ok && e2_ok));
				}
			}
			const uint64_t opay = ((uint64_t)(uint32_t)__shfl((int)depth_bits, owner) << 32) | (uint32_t)__shfl(item, owner);
			const unsigned long long m = __ballot(pass);
			if (pass)
			{
				const uint32_t pos = n_staged + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
				pay[pos] = opay;
				stl[pos] = (uint8_t)tl;
			}
			n_staged += (uint32_t)__popcll(m);
		}
		// the rows are rewritten by the next slab: everyone is done reading
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
	};
	for (;;)
	{
		ERT(const uint64_t tp0 = wall_clock64();)
		__syncthreads(); // (everybody is done with the previous chunk's tables)
		if (threadIdx.x == 0) s_chunk = atomicAdd(a.geom.slab_ctr + 6, 1u);
		__syncthreads();
		const uint32_t chunk = s_chunk;
		if (chunk >= nchunks) break;
		int r;
		{
			int lo = 0, hi = a.R; // the last region with s_rchunk[r] <= chunk
			while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_rchunk[mid] <= chunk) lo = mid; else hi = mid; }
			r = lo;
		}
		if (r != cur_r)
		{
			// the region's list = the binning workgroups' lists for it, strung together: two of them per thread, a scan over the workgroup
			cur_r = r; tx0 = (r % a.RX) * RT; ty0 = (r / a.RX) * RT; ERT(tm_regions++;)
			const int b = 2 * (int)threadIdx.x;
			const uint2 e0 = b < a.bin_wgs ? a.geom.rtab[(size_t)b * FR_MAX_REGIONS + r] : make_uint2(0u, 0u);
			const uint2 e1 = b + 1 < a.bin_wgs ? a.geom.rtab[(size_t)(b + 1) * FR_MAX_REGIONS + r] : make_uint2(0u, 0u);
			const uint32_t both = e0.y + e1.y;
			const uint32_t incl = wave_incl_scan_u32(both, lane);
			if (lane == 63) s_part[wv] = incl;
			if (FOV && threadIdx.x >= FR_ER_THREADS - 64)
			{
				const int x = tx0 + (lane & (RT - 1)), y = ty0 + (lane >> 3);
				s_tmin[lane] = (x < a.gx && y < a.gy) ? a.tile_lv[a.T + y * a.gx + x] : __int_as_float(0x7fc00000); // (NaN: no Gaussian passes)
			}
			__syncthreads();
			uint32_t off = 0, tot = 0;
#pragma unroll
			for (int w = 0; w < FR_ER_WAVES; w++) { const uint32_t ws = s_part[w]; if (w < wv) off += ws; tot += ws; }
			const uint32_t ex = off + incl - both;
			if (b < FR_BIN_BLOCKS) { s_pre[b] = ex; s_src[b] = (uint32_t)b * (uint32_t)a.wcap + e0.x; }
			if (b + 1 < FR_BIN_BLOCKS) { s_pre[b + 1] = ex + e0.y; s_src[b + 1] = (uint32_t)(b + 1) * (uint32_t)a.wcap + e1.x; }
			if (threadIdx.x == 0) s_pre[FR_BIN_BLOCKS] = tot;
			__syncthreads();
			n = (int)s_pre[FR_BIN_BLOCKS];
			npre = min(a.bin_wgs, FR_BIN_BLOCKS);
		}
		const int j0 = (int)(chunk - s_rchunk[r]) * FR_ER_CHUNK, jend = min(n, j0 + FR_ER_CHUNK);
		// two slabs per wave, both fetched before the first is walked
		ERT(const uint64_t tq0 = wall_clock64(); tm_pro += tq0 - tp0; tm_chunks++;)
		SlabIn sa, sb;
		const int ja = j0 + wv * 64, jb = ja + FR_ER_WAVES * 64;
		const bool has_a = ja < jend, has_b = jb < jend; // (wave-uniform)
		if (has_a) fetch(ja, jend, sa);
		if (has_b) fetch(jb, jend, sb);
		ERT(if (has_a) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const uint64_t tq1 = wall_clock64(); tm_fetch += tq1 - tq0; const uint64_t fl0 = tm_flush;)
		if (has_a) process(sa);
		if (has_b) process(sb);
		if (n_staged != 0) flush(); // (the next chunk may belong to another region)
		ERT(tm_proc += (wall_clock64() - tq1) - (tm_flush - fl0);)
	}
#ifdef FR_ER_TIMERS
	if (lane == 0)
	{
		float *d = a.geom.cov3D + (size_t)((int)blockIdx.x * FR_ER_WAVES + wv) * 8; // (developer build: the covariance rows the inference variants do not use)
		d[0] = (float)(wall_clock64() - tm_start); d[1] = (float)tm_pro; d[2] = (float)tm_fetch; d[3] = (float)tm_proc; d[4] = (float)tm_flush;
		d[5] = (float)tm_chunks; d[6] = (float)tm_flushes; d[7] = (float)tm_regions;
	}
#endif
#undef ERT
}

__global__ void k_mark_visible(int P, const float *means3D, const float *vm, uint8_t *present)
{
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= P) return;
	const float z = vm[2] * means3D[3 * idx] + vm[6] * means3D[3 * idx + 1] + vm[10] * means3D[3 * idx + 2] + vm[14];
	present[idx] = !(z <= 0.2f);
}

// Packed copies of a static model (fovraster.h: fr_forward_args.packed_geom / packed_colour). One thread per output
// float4: the writes are perfectly coalesced, the reads are small strided gathers that run once per model.
__global__ void k_pack_geom(int P, const float *means3D, const float *scales, const float *rotations, const float *opacities,
	int levels, const float *highest_levels, float4 *out)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (size_t)P * 4) return;
	const size_t g = i >> 2;
	const int part = (int)(i & 3);
	float4 v;
	if (part == 0) v = make_float4(means3D[3 * g], means3D[3 * g + 1], means3D[3 * g + 2], scales[3 * g]);
	else if (part == 1) v = make_float4(scales[3 * g + 1], scales[3 * g + 2], rotations[4 * g], rotations[4 * g + 1]);
	else if (part == 2) v = make_float4(rotations[4 * g + 2], rotations[4 * g + 3], highest_levels ? highest_levels[g] : 0.0f, 0.0f);
	else
	{
		const float *o = opacities + g * (size_t)levels;
		v = make_float4(o[0], levels > 1 ? o[1] : 0.0f, levels > 2 ? o[2] : 0.0f, levels > 3 ? o[3] : 0.0f);
	}
	out[i] = v;
}
__global__ void k_pack_cull(int P, const float *means3D, const float *scales, const float *rotations, float4 *out)
{
	const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= (size_t)P) return;
	const float sc[3] = { scales[3 * g], scales[3 * g + 1], scales[3 * g + 2] };
	out[g] = make_float4(means3D[3 * g], means3D[3 * g + 1], means3D[3 * g + 2], rho_unit(sc, ((const float4 *)rotations)[g]));
}
__global__ void k_pack_colour(int P, const float *shs, const float *shs_rest, const float *shs_dcs, float *out)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (size_t)P * 64) return;
	const size_t g = i >> 6;
	const int k = (int)(i & 63);
	float v = 0.0f;
	if (shs_dcs != nullptr)
	{
		// RF: shs = coefficients 1..15
		if (k < 45) v = shs[g * 45 + k];
		else if (k >= 48 && k < 60) v = shs_dcs[g * 12 + (k - 48)];
	}
	else if (shs_rest != nullptr) { if (k < 45) v = shs_rest[g * 45 + k]; else if (k < 48) v = shs[g * 3 + (k - 45)]; }
	else { if (k < 45) v = shs[g * 48 + 3 + k]; else if (k < 48) v = shs[g * 48 + (k - 45)]; }
	out[i] = v;
}
int launch_pack_geom(int P, const float *means3D, const float *scales, const float *rotations, const float *opacities, int levels,
	const float *highest_levels, float *out, hipStream_t stream)
{
	const size_t n = (size_t)P * 4;
	hipLaunchKernelGGL(k_pack_geom, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P, means3D, scales, rotations, opacities, levels,
		highest_levels, (float4 *)out);
	return check_launch("pack_geom", stream, 0);
}
int launch_pack_cull(int P, const float *means3D, const float *scales, const float *rotations, float *out, hipStream_t stream)
{
	hipLaunchKernelGGL(k_pack_cull, dim3((unsigned)(((size_t)P + 255) / 256)), dim3(256), 0, stream, P, means3D, scales, rotations, (float4 *)out);
	return check_launch("pack_cull", stream, 0);
}
int launch_pack_colour(int P, const float *shs, const float *shs_rest, const float *shs_dcs, float *out, hipStream_t stream)
{
	const size_t n = (size_t)P * 64;
	hipLaunchKernelGGL(k_pack_colour, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P, shs, shs_rest, shs_dcs, out);
	return check_launch("pack_colour", stream, 0);
}

// ---- launchers -------------------------------------------------------------------------------
// bytes of the RF tile table in LDS: tile_min floats + one blend bit per tile
static inline size_t lds_tile_table_bytes(int T) { return (size_t)((T + 7) / 8) * sizeof(uint32_t); } // four bits per tile
int launch_tile_levels(FwdCtx &c)
{
	const fr_forward_args *a = c.a;
	hipLaunchKernelGGL(k_tile_levels, dim3(((c.gx + FR_LV_PATCH - 1) / FR_LV_PATCH) * ((c.gy + FR_LV_PATCH - 1) / FR_LV_PATCH)), dim3(256), 0, c.stream,
		c.T, c.gx, c.gy, a->W, a->H, a->gaze_x, a->gaze_y, a->alpha, c.img.tile_lv, c.img.lv_bbox, c.geom.slab_ctr,
		a->variant == FR_VARIANT_MMFR_PCHECK_OBB ? a->cur_level : -1.0f);
	return check_launch("tile_levels", c.stream, a->debug);
}

static PreArgs make_pre_args(FwdCtx &c)
{
	const fr_forward_args *a = c.a;
	PreArgs p;
	p.lds_tiles = 0; p.wbase_lds = 0;
	p.P = a->P; p.D = a->D; p.M = a->M; p.W = a->W; p.H = a->H; p.gx = c.gx; p.gy = c.gy;
	p.tanfovx = a->tanfovx; p.tanfovy = a->tanfovy; p.focal_x = c.focal_x; p.focal_y = c.focal_y;
	p.scale_modifier = a->scale_modifier;
	p.means3D = a->means3D; p.scales = a->scales; p.rotations = a->rotations; p.opacities = a->opacities;
	p.shs = a->shs; p.shs_rest = a->shs_rest; p.cov3D_precomp = a->cov3D_precomp; p.colors_precomp = a->colors_precomp;
	p.viewmatrix = a->viewmatrix; p.projmatrix = a->projmatrix; p.campos = a->campos;
	p.packed_geom = a->packed_geom; p.packed_colour = a->packed_colour; p.packed_cull = a->packed_cull;
	p.shs_dcs = a->shs_dcs; p.highest_levels = a->highest_levels; p.tile_lv = c.img.tile_lv; p.lv_bbox = c.img.lv_bbox; p.T = c.T;
	p.radii = a->radii; p.geom = c.geom; p.tile_count = c.img.tile_count; p.hist = c.img.hist; p.raw = a->raw_activations;
	p.write_cov3D = has_backward(a->variant) ? 1 : 0;
	p.prefiltered = a->prefiltered;
	p.proj_waves = c.proj_waves; p.proj_cpw = c.proj_cpw;
	p.fuse_scan = 0;
	p.regions = 0; p.region_rx = 0; p.wcap = 0;
	return p;
}

// stage "project": per-Gaussian projection + conservative culling -> vis_list
int launch_project(FwdCtx &c)
{
	const fr_forward_args *a = c.a;
	PreArgs p = make_pre_args(c);
	{
		// persistent waves: exactly as many workgroups as the device keeps resident (a second, partial round of
		// workgroups would run on a half-empty chip)
		static thread_local int resident_of[8][6]; // per host thread and device (zero = not asked yet)
		int cur_dev = 0;
		(void)hipGetDevice(&cur_dev);
		int *const resident = resident_of[cur_dev & 7];
		const bool packed = a->packed_cull != nullptr;
		const int vslot = a->variant == FR_VARIANT_ORIGINAL ? 0 : (is_fov(a->variant) ? 1 : 2); // the cull pass only knows "level box or not"
		const int slot = vslot + (packed ? 3 : 0);
		if (resident[slot] == 0)
		{
			int per_cu = 0, dev = 0;
			hipDeviceProp_t prop;
			const void *fns[6] = { (const void *)k_project<FR_VARIANT_ORIGINAL>, (const void *)k_project<FR_VARIANT_FOV_PCHECK_OBB>,
				(const void *)k_project<FR_VARIANT_PCHECK_OBB>, (const void *)k_project<FR_VARIANT_ORIGINAL, true>,
				(const void *)k_project<FR_VARIANT_FOV_PCHECK_OBB, true>, (const void *)k_project<FR_VARIANT_PCHECK_OBB, true> };
			const void *fn = fns[slot];
			if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
				hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, FR_PROJ_THREADS, 0) != hipSuccess || per_cu < 1)
			{ per_cu = 4; prop.multiProcessorCount = 256; }
			resident[slot] = per_cu * prop.multiProcessorCount;
		}
		const int pchunks = (a->P + FR_PROJ_THREADS - 1) / FR_PROJ_THREADS;
		const int pmax = resident[slot] < FR_PROJ_MAX_WAVES / (FR_PROJ_THREADS / 64) ? resident[slot] : FR_PROJ_MAX_WAVES / (FR_PROJ_THREADS / 64);
		const dim3 pgrid(pchunks < pmax ? pchunks : pmax), pblock(FR_PROJ_THREADS);
		c.proj_waves = (int)pgrid.x * (FR_PROJ_THREADS / 64);
		c.proj_cpw = proj_chunks_per_wave(a->P, c.proj_waves);
		p.proj_waves = c.proj_waves; p.proj_cpw = c.proj_cpw;
#define LAUNCH_PROJ(V) do { if (packed) hipLaunchKernelGGL((k_project<V, true>), pgrid, pblock, 0, c.stream, p); \
	else hipLaunchKernelGGL((k_project<V>), pgrid, pblock, 0, c.stream, p); } while (0)
		switch (vslot)
		{
		case 0: LAUNCH_PROJ(FR_VARIANT_ORIGINAL); break;
		case 1: LAUNCH_PROJ(FR_VARIANT_FOV_PCHECK_OBB); break;
		default: LAUNCH_PROJ(FR_VARIANT_PCHECK_OBB); break;
		}
#undef LAUNCH_PROJ
		return check_launch("project", c.stream, a->debug);
	}
}

// stage "bin": projection of the cull pass's survivors, tile counts (LDS histograms), colours, item rows (k_bin)
int launch_bin(FwdCtx &c)
{
	const fr_forward_args *a = c.a;
	PreArgs p = make_pre_args(c);
	int nblk = bin_blocks(a->P);
	const dim3 block(FR_BIN_THREADS);
	// Where the tiles are counted: an LDS histogram per workgroup of 32-bit counts (grids up to 16 Ki tiles), of 16-bit counts
	// (up to 34 816 tiles: a workgroup then takes fewer than 65 536 items, so the resident workgroups must cover all
	// P Gaussians with a margin), or global counters.
	static thread_local int cus = 0;
	if (cus == 0)
	{
		int dev = 0; hipDeviceProp_t prop;
		cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 64;
		(void)hipGetLastError();
	}
	const int64_t wgs_lo = nblk < cus ? nblk : cus;
	c.hist_mode = c.img.hist == nullptr ? 0 : (c.T <= FR_LDS_HIST_MAX_TILES ? 1 :
		(wgs_lo * (FR_BIN_THREADS / 64) * (int64_t)(FR_HIST16_MAX_SLABS - 1) * 64 >= (int64_t)a->P ? 2 : 0)); // (wave w takes slabs w, w + waves, ...)
	const bool ldsh = c.hist_mode != 0;
	// LDS per workgroup: tile histogram (+ RF: the 4-bit tile table), the waves' staging / owner rows, the cull pass's running counts
	p.lds_tiles = (is_fov(a->variant) && ldsh) ? 1 : 0;
	const size_t hist_bytes = c.hist_mode == 2 ? (size_t)((c.T + 1) / 2) * sizeof(uint32_t) : (ldsh ? (size_t)c.T * sizeof(uint32_t) : 0);
	const size_t lds_fixed = ((hist_bytes + (p.lds_tiles ? lds_tile_table_bytes(c.T) : 0) + 15) & ~(size_t)15) +
		(size_t)FR_BIN_THREADS * (4 * sizeof(float4) + sizeof(int));
	const size_t wbase_bytes = (size_t)(c.proj_waves + 1) * sizeof(uint32_t);
	p.wbase_lds = lds_fixed + wbase_bytes <= 156u * 1024u ? 1 : 0; // (a 4K grid's 16-bit histogram + a large cloud's 32 KiB of counts do not both fit)
	size_t lds = lds_fixed + (p.wbase_lds ? wbase_bytes : 0);
	// the tile scan as this kernel's tail (k_bin): 32-bit LDS histograms, a frame that publishes its totals through the pinned block
	// (c.totals_host_dev / c.totals_seq are set), at most FR_SCAN_MAX_TILES tiles; FOVRASTER_FUSE_SCAN=0 keeps the kernel of its own
	static const bool fuse_ok = []() { const char *e = getenv("FOVRASTER_FUSE_SCAN"); return !(e && e[0] == '0'); }();
	c.scan_fused = fuse_ok && c.hist_mode == 1 && c.T <= FR_SCAN_MAX_TILES && !a->debug;
	if (c.scan_fused)
	{
		p.fuse_scan = 1; p.ts = make_tile_scan_args(c);
		const size_t scan_bytes = (size_t)tile_scan_lds_words<FR_TILE_SCAN_THREADS>() * sizeof(uint32_t);
		if (lds < scan_bytes) lds = scan_bytes;
	}
	// Never more workgroups than the device keeps resident (one per CU: a second one doubles the histogram flushes -- one returning
	// atomic per workgroup and touched tile -- and the table prologues for slabs that one workgroup's sixteen waves already cover).
	auto launch = [&](const void *fn, void (*kern)(const PreArgs), size_t dyn) {
		static thread_local struct { const void *fn; size_t dyn; int wgs; } cache[48];
		static thread_local int ncache = 0;
		int wgs = 0;
		for (int i = 0; i < ncache; i++) if (cache[i].fn == fn && cache[i].dyn == dyn) wgs = cache[i].wgs;
		if (wgs == 0)
		{
			if (dyn > 64u * 1024u && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess)
			{ set_error("hipFuncSetAttribute(k_bin): %s", hipGetErrorString(hipGetLastError())); return FR_ERR_HIP; }
			int per_cu = 0, dev = 0;
			hipDeviceProp_t prop;
			if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
				hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, FR_BIN_THREADS, dyn) != hipSuccess || per_cu < 1)
			{ per_cu = 1; prop.multiProcessorCount = 256; (void)hipGetLastError(); }
			wgs = per_cu * prop.multiProcessorCount;
			if (ncache < 48) { cache[ncache].fn = fn; cache[ncache].dyn = dyn; cache[ncache].wgs = wgs; ncache++; }
		}
		nblk = nblk < wgs ? nblk : wgs;
		p.regions = c.regions; p.region_rx = c.region_rx; p.wcap = (int)(wlist_cap((size_t)a->P) / (size_t)nblk);
		c.wcap = p.wcap;
		hipLaunchKernelGGL(kern, dim3(nblk), block, dyn, c.stream, p);
		return FR_OK;
	};
	int lrc = FR_OK;
	// the packed model layout and the candidate rows are compile-time variants of the kernel (run-time tests on the pointers cost
	// the ordinary path 5 %): packed needs both packed tensors; the rows exist when k_project stored them (foveated variants,
	// unpacked cull pass, scales + rotations given)
	// region-major emission: the workgroups also list their items by screen region (k_emit_regions)
	{
		static const int env_mode = []() { const char *e = getenv("FOVRASTER_EMIT_REGIONS"); return e ? atoi(e) : 0; }(); // (developer A / B runs)
		const int mode = a->emit_regions | env_mode;
		const int rx = (c.gx + FR_REGION_TILES - 1) / FR_REGION_TILES, ry = (c.gy + FR_REGION_TILES - 1) / FR_REGION_TILES;
		c.regions = (mode != 0 && c.hist_mode == 1 && c.scan_fused && rx * ry <= FR_MAX_REGIONS && !a->debug) ? rx * ry : 0; // (the chunk table is made behind the fused scan)
		c.region_rx = rx;
	}
	const bool packed = a->packed_geom && a->packed_colour;
	const bool crow = !packed && is_fov(a->variant) && a->packed_cull == nullptr && a->cov3D_precomp == nullptr;
#define LAUNCH_BIN_PC(V, PK, CR) do { if (c.hist_mode == 2) lrc = launch((const void *)k_bin<V, 2, PK, CR>, k_bin<V, 2, PK, CR>, lds); \
	else if (ldsh) lrc = launch((const void *)k_bin<V, 1, PK, CR>, k_bin<V, 1, PK, CR>, lds); \
	else lrc = launch((const void *)k_bin<V, 0, PK, CR>, k_bin<V, 0, PK, CR>, lds); } while (0)
#define LAUNCH_BIN(V) do { if (packed) LAUNCH_BIN_PC(V, true, false); else if (crow) LAUNCH_BIN_PC(V, false, true); else LAUNCH_BIN_PC(V, false, false); } while (0)
	switch (a->variant)
	{
	case FR_VARIANT_ORIGINAL: LAUNCH_BIN(FR_VARIANT_ORIGINAL); break;
	case FR_VARIANT_FOV_PCHECK_OBB: LAUNCH_BIN(FR_VARIANT_FOV_PCHECK_OBB); break;
	case FR_VARIANT_MMFR_PCHECK_OBB:      // plain colours + the level filter (on the skip key, see k_tile_levels): the same kernel
	case FR_VARIANT_NAIVE_FOV_PCHECK_OBB: // (no packed instantiation: validate_forward refuses the packed tensors)
		if (crow) LAUNCH_BIN_PC(FR_VARIANT_NAIVE_FOV_PCHECK_OBB, false, true); else LAUNCH_BIN_PC(FR_VARIANT_NAIVE_FOV_PCHECK_OBB, false, false);
		break;
	default: LAUNCH_BIN(FR_VARIANT_PCHECK_OBB); break; // every other cull variant bins alike
	}
#undef LAUNCH_BIN
#undef LAUNCH_BIN_PC
	if (lrc) return lrc;
	c.bin_wgs = nblk;
	return check_launch("bin", c.stream, a->debug);
}

static int launch_emit_regions(FwdCtx &c)
{
	const fr_forward_args *a = c.a;
	EmitRegArgs e;
	e.gx = c.gx; e.gy = c.gy; e.T = c.T; e.RX = c.region_rx; e.R = c.regions; e.bin_wgs = c.bin_wgs; e.wcap = c.wcap;
	e.geom = c.geom; e.tile_lv = c.img.tile_lv; e.ranges = c.img.ranges; e.cursor = c.img.tile_count; e.entries = c.bin.entries;
	e.totals = c.img.totals; e.capacity = (uint32_t)c.capacity; e.items_cap = (uint32_t)c.items_cap;
	// as many workgroups as the device keeps resident (four of 37 KiB of LDS per CU): they pull the chunks from a counter
	static thread_local int wgs = 0;
	if (wgs == 0)
	{
		int per_cu = 0, dev = 0; hipDeviceProp_t prop;
		if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
			hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_emit_regions<FR_VARIANT_FOV_PCHECK_OBB>, FR_ER_THREADS, 0) != hipSuccess || per_cu < 1)
		{ per_cu = 4; prop.multiProcessorCount = 256; (void)hipGetLastError(); }
		wgs = per_cu * prop.multiProcessorCount;
	}
	const dim3 grid((unsigned)wgs), block(FR_ER_THREADS);
	switch (a->variant)
	{
	case FR_VARIANT_ORIGINAL: hipLaunchKernelGGL((k_emit_regions<FR_VARIANT_ORIGINAL>), grid, block, 0, c.stream, e); break;
	case FR_VARIANT_FOV_PCHECK_OBB:
	case FR_VARIANT_MMFR_PCHECK_OBB:
	case FR_VARIANT_NAIVE_FOV_PCHECK_OBB: hipLaunchKernelGGL((k_emit_regions<FR_VARIANT_FOV_PCHECK_OBB>), grid, block, 0, c.stream, e); break;
	default: hipLaunchKernelGGL((k_emit_regions<FR_VARIANT_PCHECK_OBB>), grid, block, 0, c.stream, e); break;
	}
	return check_launch("emit_regions", c.stream, a->debug);
}

int launch_emit(FwdCtx &c)
{
	const fr_forward_args *a = c.a;
	// region-major emission when k_bin listed the items by region and every workgroup's segment held them (c.regions_ok: the host's copy
	// of slab_ctr[5]); the share-per-workgroup kernel below otherwise
	if (c.regions && c.regions_ok) return launch_emit_regions(c);
	EmitArgs e;
	e.P = a->P; e.gx = c.gx; e.gy = c.gy; e.T = c.T; e.radii = a->radii; e.geom = c.geom;
	e.highest_levels = a->highest_levels; e.tile_lv = c.img.tile_lv; e.lv_bbox = c.img.lv_bbox; e.ranges = c.img.ranges;
	e.cursor = c.img.tile_count; e.entries = c.bin.entries; e.hist = c.img.hist;
	e.totals = c.img.totals; e.capacity = (uint32_t)c.capacity; e.items_cap = (uint32_t)c.items_cap;
	const bool ldsh = c.hist_mode != 0; // (as k_bin counted: launch_bin)
	const dim3 grid(c.bin_wgs), block(FR_EMIT_THREADS);
	e.lds_tiles = (is_fov(a->variant) && ldsh) ? 1 : 0;
	const size_t lds = (((c.hist_mode == 2 ? (size_t)((c.T + 1) / 2) * sizeof(uint32_t) : (ldsh ? (size_t)c.T * sizeof(uint32_t) : 0)) +
		(e.lds_tiles ? lds_tile_table_bytes(c.T) : 0) + 15) & ~(size_t)15) + (size_t)FR_EMIT_THREADS * 4 * sizeof(float4); // + the pair loop's owner rows
	if (lds > 64u * 1024u)
	{
		// (cursors + owner rows exceed the default 64 KiB for every tile grid of a 1080p frame: every instantiation gets the raised limit)
		static const hipError_t once = []() {
			const void *fns[] = { (const void *)k_emit<FR_VARIANT_ORIGINAL, 0>, (const void *)k_emit<FR_VARIANT_ORIGINAL, 1>, (const void *)k_emit<FR_VARIANT_ORIGINAL, 2>,
				(const void *)k_emit<FR_VARIANT_PCHECK_OBB, 0>, (const void *)k_emit<FR_VARIANT_PCHECK_OBB, 1>, (const void *)k_emit<FR_VARIANT_PCHECK_OBB, 2>,
				(const void *)k_emit<FR_VARIANT_FOV_PCHECK_OBB, 0>, (const void *)k_emit<FR_VARIANT_FOV_PCHECK_OBB, 1>, (const void *)k_emit<FR_VARIANT_FOV_PCHECK_OBB, 2> };
			for (const void *fn : fns)
			{
				const hipError_t e1 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (FR_EMIT_THREADS > 768 ? 154 : 156) * 1024);
				if (e1 != hipSuccess) return e1;
			}
			return hipSuccess; }();
		if (once != hipSuccess) { set_error("hipFuncSetAttribute(k_emit): %s", hipGetErrorString(once)); return FR_ERR_HIP; }
	}
#define LAUNCH_EMIT(V) do { if (c.hist_mode == 2) hipLaunchKernelGGL((k_emit<V, 2>), grid, block, lds, c.stream, e); \
	else if (ldsh) hipLaunchKernelGGL((k_emit<V, 1>), grid, block, lds, c.stream, e); \
	else hipLaunchKernelGGL((k_emit<V, 0>), grid, block, lds, c.stream, e); } while (0)
	switch (a->variant)
	{
	case FR_VARIANT_ORIGINAL: LAUNCH_EMIT(FR_VARIANT_ORIGINAL); break;
	case FR_VARIANT_FOV_PCHECK_OBB:
	case FR_VARIANT_MMFR_PCHECK_OBB:
	case FR_VARIANT_NAIVE_FOV_PCHECK_OBB: LAUNCH_EMIT(FR_VARIANT_FOV_PCHECK_OBB); break; // the level filter is the same
	default: LAUNCH_EMIT(FR_VARIANT_PCHECK_OBB); break;
	}
#undef LAUNCH_EMIT
	return check_launch("emit", c.stream, a->debug);
}

int launch_mark_visible(int P, const float *means3D, const float *vm, uint8_t *present, hipStream_t s)
{
	hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, vm, present);
	return check_launch("mark_visible", s, false);
}

} // namespace fr
