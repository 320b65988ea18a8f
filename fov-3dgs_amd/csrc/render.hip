// Per-tile front-to-back alpha blending (forward).
//
// Replaces (reference, paths under fov3dgs/submodules/):
//   renderCUDA            diff-gaussian-rasterization/cuda_rasterizer/forward.cu:267-384           (ORIGINAL)
//                         …_pcheck_obb_sum/cuda_rasterizer/forward.cu:298-430  (+ power<-4.5, counts, contributions)
//                         …_pcheck_obb/cuda_rasterizer/forward.cu:243-384      (inference; no final_T / n_contrib)
//   renderCUDA (RF)       …_fov_pcheck_obb/cuda_rasterizer/forward.cu:490-609  single-level tiles
//   renderCUDA_blending   …_fov_pcheck_obb/cuda_rasterizer/forward.cu:262-476  two-level tiles
//
// MI355X design: a tile is blended by 256/PPL threads, each owning PPL pixels of one column
// (rows ry, ry+16/PPL, ...). With PPL=4 a tile is ONE wave64: the per-instance record broadcast
// out of LDS is amortised over four pixels per lane, `done` votes are wave ballots instead of
// workgroup barriers, and each 16x4 pixel strip is skipped wave-uniformly once it saturates.
// Instance records are gathered as 48-byte AoS (3 x b128) per Gaussian by the whole group.
// The foveated renderer handles single-level and two-level tiles in one launch (the reference
// launches two full grids that early-return on each other's tiles).
#include "common.h"

namespace fr {

__device__ __forceinline__ float fast_exp(float p)
{
	// exp(p) = 2^(p*log2 e) on the transcendental unit (v_exp_f32); |rel err| ~ 2 ulp for p in [-6,0]
	return __builtin_amdgcn_exp2f(p * 1.4426950408889634f);
}

// sum over the 64 lanes of a wave, result valid in every lane (returned through readlane 63)
__device__ __forceinline__ float wave_sum(float x)
{
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false)); // row_half_mirror
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false)); // row_mirror
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xA, 0xF, false)); // row_bcast15 -> rows 1,3
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xC, 0xF, false)); // row_bcast31 -> rows 2,3
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

typedef float v2f __attribute__((ext_vector_type(2)));

// power = -0.5 (A dx^2 + C dy^2) - B dx dy for two rows at once (packed fp32: v_pk_mul/v_pk_fma)
__device__ __forceinline__ v2f power2(v2f dy, float C, float adx2, float bdx)
{
	const v2f cdy = C * dy;
	const v2f s = __builtin_elementwise_fma(cdy, dy, (v2f){ adx2, adx2 });
	const v2f nb = -(bdx * dy);
	return __builtin_elementwise_fma((v2f){ -0.5f, -0.5f }, s, nb);
}

// One pixel's front-to-back update for one splat, fully predicated (no lane-divergent branch: hipcc turns
// nested divergent ifs on bool state into long chains of scalar mask merges).
__device__ __forceinline__ void blend_px(bool hit, float alpha, float cr, float cg, float cb,
	float &T, float &C0, float &C1, float &C2, bool &done, float &w_out, bool &acc_out)
{
	const float test_T = T * (1.0f - alpha);
	const bool live = hit && !(alpha < 1.0f / 255.0f);
	const bool sat = live && (test_T < 0.0001f);
	const bool acc = live && !sat;
	const float w = acc ? alpha * T : 0.0f;
	C0 = fmaf(cr, w, C0); C1 = fmaf(cg, w, C1); C2 = fmaf(cb, w, C2);
	T = acc ? test_T : T;
	done = done || sat;
	w_out = w; acc_out = acc;
}

struct RenderArgs {
	int W, H, gx;
	const uint2 *ranges;
	const uint32_t *point_list;
	const float4 *rec;
	const float4 *lvl;      // RF
	const float *tile_lv;   // RF float[5][T]
	const uint32_t *tile_order; // tiles sorted by descending list length (longest first), or null
	int T;
	const float *bg;
	float *out_color;
	float *final_T;
	uint32_t *n_contrib;
	int *gaussians_count;   // RS / MAX / LWMC
	float *contributions;   // RS / MAX / LWMC
	const float *loss_map;  // LWMC
};

// ---------------- ORIGINAL / PCHECK_OBB_SUM / PCHECK_OBB ----------------
// Instances are staged NT (= threads of the group) at a time: with PPL = 4 that is 64 records = 2.3 KiB of
// LDS per wave, so occupancy is bounded by registers, not LDS. The records of the NEXT batch are
// prefetched into registers before the current batch is blended (the point_list -> record gather is two
// dependent global loads). RS semantics that depend on the reference's 256-entry batches
// (gaussians_count: +1 per entry of every batch a still-live tile fetches, RS forward.cu:349-361) are kept
// by taking the "tile finished" decision only at multiples of 256 entries.
template <int VARIANT, int PPL>
__global__ void __launch_bounds__(256 / PPL) k_render(const RenderArgs a)
{
	constexpr int NT = 256 / PPL;        // threads per tile == staging batch
	constexpr int RSTEP = 16 / PPL;      // row distance between a lane's pixels
	constexpr bool CUTOFF = VARIANT != FR_VARIANT_ORIGINAL;
	constexpr bool SUM = VARIANT == FR_VARIANT_PCHECK_OBB_SUM;   // contributions += alpha*T, count per fetched entry
	constexpr bool PMAX = VARIANT == FR_VARIANT_PCHECK_OBB_MAX;  // contributions = max alpha*T, count per in-support pixel
	constexpr bool LWMC = VARIANT == FR_VARIANT_PCHECK_OBB_LWMC; // per-pixel loss to its max-contribution Gaussian
	constexpr bool FETCHCNT = SUM || LWMC;                       // gaussians_count per fetched entry (256-batches)
	constexpr bool NEEDID = SUM || PMAX || LWMC;
	constexpr bool AUX = VARIANT != FR_VARIANT_PCHECK_OBB; // final_T / n_contrib kept for backward

	__shared__ float4 s0[NT];
	__shared__ float4 s1[NT];
	__shared__ float s2[NT];
	__shared__ int sid[NEEDID ? NT : 1];

	const int tile = a.tile_order ? (int)a.tile_order[blockIdx.x] : (int)blockIdx.x;
	const int tx = tile % a.gx, ty = tile / a.gx;
	const int tid = threadIdx.x;
	const int lx = tid & 15, ry = tid >> 4;
	const int px = tx * FR_TILE + lx;
	const float pxf = (float)px;
	const uint2 range = a.ranges[tile];
	const int n = (int)(range.y - range.x);

	float T[PPL], C0[PPL], C1[PPL], C2[PPL], pyf[PPL];
	uint32_t last[PPL];
	bool done[PPL], inside[PPL];
#pragma unroll
	for (int k = 0; k < PPL; k++)
	{
		const int py = ty * FR_TILE + ry + k * RSTEP;
		pyf[k] = (float)py;
		inside[k] = px < a.W && py < a.H;
		done[k] = !inside[k];
		T[k] = 1.0f; C0[k] = C1[k] = C2[k] = 0.0f; last[k] = 0;
	}
	float best_w[LWMC ? PPL : 1];   // LWMC forward.cu:347-348: running max contribution per pixel ...
	int best_id[LWMC ? PPL : 1];    // ... and whose it is (defaults to Gaussian 0, as in the reference)
#pragma unroll
	for (int k = 0; k < (LWMC ? PPL : 1); k++) { best_w[k] = 0.0f; best_id[k] = 0; }

	// prefetch registers
	uint32_t pid = 0;
	float4 p0 = make_float4(0, 0, 0, 0), p1 = p0;
	float p2 = 0.f;
	if (tid < n)
	{
		pid = a.point_list[range.x + tid];
		const float4 *r = a.rec + 3 * (size_t)pid;
		p0 = r[0]; p1 = r[1]; p2 = r[2].x;
	}
	bool finished = false; // SUM: every pixel saturated, only counting until the next 256 boundary
	for (int base = 0; base < n; base += NT)
	{
		bool all_done = true;
#pragma unroll
		for (int k = 0; k < PPL; k++) all_done = all_done && done[k];
		const bool wg_done = __syncthreads_and(all_done) != 0; // also fences the LDS reuse
		if (FETCHCNT)
		{
			if ((base & 255) == 0) { if (wg_done) break; }
			finished = wg_done;
		}
		else if (wg_done) break;
		if (base + tid < n)
		{
			s0[tid] = p0; s1[tid] = p1; s2[tid] = p2;
			if (NEEDID) sid[tid] = (int)pid;
			if (FETCHCNT) atomicAdd(&a.gaussians_count[pid], 1);
		}
		if (base + NT + tid < n)
		{
			pid = a.point_list[range.x + base + NT + tid];
			const float4 *r = a.rec + 3 * (size_t)pid;
			p0 = r[0]; p1 = r[1]; p2 = r[2].x;
		}
		__syncthreads();
		// SUM && finished: nothing left to blend, the loop only keeps counting (no `continue` here: this
		// loop carries barriers, see the note in k_bin)
		const int cnt = (FETCHCNT && finished) ? 0 : min(NT, n - base);
		static_assert(PPL == 4, "the packed inner loop handles four rows per lane");
		const v2f py01 = { pyf[0], pyf[1] }, py23 = { pyf[2], pyf[3] };
		for (int j = 0; j < cnt; j++)
		{
			if (__all(done[0] && done[1] && done[2] && done[3])) break; // wave saturated
			const float4 g0 = s0[j];
			const float4 g1 = s1[j];
			const float dx = g0.x - pxf;
			const float adx2 = (g0.z * dx) * dx;     // A*dx*dx
			const float bdx = g0.w * dx;             // B*dx
			const v2f pw01 = power2(g0.y - py01, g1.x, adx2, bdx);
			const v2f pw23 = power2(g0.y - py23, g1.x, adx2, bdx);
			const float pw[4] = { pw01.x, pw01.y, pw23.x, pw23.y };
			bool hit[4];
#pragma unroll
			for (int k = 0; k < 4; k++) hit[k] = !done[k] && !(pw[k] > 0.0f) && !(CUTOFF && pw[k] < -4.5f);
			if (!__any(hit[0] || hit[1] || hit[2] || hit[3])) continue; // splat misses every live pixel of the tile
			const float cb = s2[j];
			float contrib_sum = 0.0f, contrib_max = 0.0f;
			bool any_contrib = false;
			if (PMAX)
			{
				// …_max forward.cu:381: +1 for every live pixel inside the splat's support (before the alpha test)
				const int c = __popcll(__ballot(hit[0])) + __popcll(__ballot(hit[1])) + __popcll(__ballot(hit[2])) + __popcll(__ballot(hit[3]));
				if ((tid & 63) == 0) atomicAdd(&a.gaussians_count[sid[j]], c);
			}
#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				if (__any(hit[k]))   // wave-uniform: skip the exp for strips the splat does not reach
				{
					const float alpha = fminf(0.99f, g1.y * fast_exp(pw[k]));
					float w; bool acc;
					blend_px(hit[k], alpha, g1.z, g1.w, cb, T[k], C0[k], C1[k], C2[k], done[k], w, acc);
					last[k] = acc ? (uint32_t)(base + j + 1) : last[k];
					if (SUM) { contrib_sum += w; any_contrib = any_contrib || acc; }
					if (PMAX) { contrib_max = fmaxf(contrib_max, w); any_contrib = any_contrib || acc; }
					if (LWMC) { const bool better = acc && (w > best_w[k]); best_w[k] = better ? w : best_w[k]; best_id[k] = better ? sid[j] : best_id[k]; }
				}
			}
			if (SUM)
			{
				if (__any(any_contrib))
				{
					const float tot = wave_sum(contrib_sum);
					if ((tid & 63) == 0) atomicAdd(&a.contributions[sid[j]], tot);
				}
			}
			if (PMAX)
			{
				if (__any(any_contrib))
				{
					// atomicMaxFloat of the reference; values are >= 0, so the integer order of the bit patterns is the float order
					float m = contrib_max;
#pragma unroll
					for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
					if ((tid & 63) == 0) atomicMax((unsigned int *)&a.contributions[sid[j]], __float_as_uint(m));
				}
			}
		}
	}

	const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
	const size_t plane = (size_t)a.W * a.H;
#pragma unroll
	for (int k = 0; k < PPL; k++)
	{
		if (!inside[k]) continue;
		const size_t pid2 = (size_t)a.W * (size_t)(ty * FR_TILE + ry + k * RSTEP) + px;
		if (AUX) { a.final_T[pid2] = T[k]; a.n_contrib[pid2] = last[k]; }
		if (LWMC) atomicAdd(&a.contributions[best_id[k]], a.loss_map[pid2]); // …_count forward.cu:435
		a.out_color[pid2] = fmaf(T[k], bg0, C0[k]);
		a.out_color[plane + pid2] = fmaf(T[k], bg1, C1[k]);
		a.out_color[2 * plane + pid2] = fmaf(T[k], bg2, C2[k]);
	}
}

// ---------------- FOV_PCHECK_OBB: single-level and two-level tiles ----------------
template <int PPL>
__global__ void __launch_bounds__(256 / PPL) k_render_fov(const RenderArgs a)
{
	constexpr int NT = 256 / PPL;
	constexpr int RSTEP = 16 / PPL;
	__shared__ float4 s0[NT];   // x, y, A, B
	__shared__ float2 s1[NT];   // C, highest_level
	__shared__ float4 sl1[NT];  // level L1: r, g, b, opacity
	__shared__ float4 sl2[NT];  // level L2 (two-level tiles only)

	const int tile = a.tile_order ? (int)a.tile_order[blockIdx.x] : (int)blockIdx.x;
	const int tx = tile % a.gx, ty = tile / a.gx;
	const int tid = threadIdx.x;
	const int lx = tid & 15, ry = tid >> 4;
	const int px = tx * FR_TILE + lx;
	const float pxf = (float)px;
	const uint2 range = a.ranges[tile];
	const int n = (int)(range.y - range.x);
	const float tlf = a.tile_lv[a.T + tile];                    // tile_min
	const bool blending = a.tile_lv[4 * (size_t)a.T + tile] != 0.0f;
	const int L1 = f2i(tlf);
	const int L2 = L1 + 1;
	const float L2f = tlf + 1.0f;
	const float tgx = a.tile_lv[2 * (size_t)a.T + tile], tgy = a.tile_lv[3 * (size_t)a.T + tile];

	float T1[PPL], T2[PPL], A0[PPL], A1[PPL], A2[PPL], B0[PPL], B1[PPL], B2[PPL], pyf[PPL], est[PPL];
	bool d1[PPL], d2[PPL], inside[PPL];
#pragma unroll
	for (int k = 0; k < PPL; k++)
	{
		const int ly = ry + k * RSTEP;
		const int py = ty * FR_TILE + ly;
		pyf[k] = (float)py;
		inside[k] = px < a.W && py < a.H;
		T1[k] = T2[k] = 1.0f; A0[k] = A1[k] = A2[k] = B0[k] = B1[k] = B2[k] = 0.0f;
		est[k] = tlf + ((float)lx * tgx + (float)ly * tgy) / (float)FR_TILE;
		if (blending) { d1[k] = !inside[k] || (est[k] > (float)L2); d2[k] = !inside[k]; }
		else { d1[k] = !inside[k]; d2[k] = true; }
	}

	// prefetch registers
	float4 p0 = make_float4(0, 0, 0, 0), pl1 = p0, pl2 = p0;
	float2 p1 = make_float2(0, 0);
	auto fetch = [&](int e)
	{
		const uint32_t id = a.point_list[range.x + e];
		const float4 *r = a.rec + 3 * (size_t)id;
		p0 = r[0];
		const float4 r1 = r[1];
		p1 = make_float2(r1.x, r1.y);
		pl1 = a.lvl[(size_t)id * FR_FOV_LEVELS + L1];
		if (blending) pl2 = a.lvl[(size_t)id * FR_FOV_LEVELS + L2];
	};
	if (tid < n) fetch(tid);
	for (int base = 0; base < n; base += NT)
	{
		bool all_done = true;
#pragma unroll
		for (int k = 0; k < PPL; k++) all_done = all_done && d1[k] && d2[k];
		if (__syncthreads_and(all_done)) break;
		if (base + tid < n) { s0[tid] = p0; s1[tid] = p1; sl1[tid] = pl1; if (blending) sl2[tid] = pl2; }
		if (base + NT + tid < n) fetch(base + NT + tid);
		__syncthreads();
		const int cnt = min(NT, n - base);
		static_assert(PPL == 4, "the packed inner loop handles four rows per lane");
		const v2f py01 = { pyf[0], pyf[1] }, py23 = { pyf[2], pyf[3] };
		for (int j = 0; j < cnt; j++)
		{
			bool lane_done = true;
#pragma unroll
			for (int k = 0; k < 4; k++) lane_done = lane_done && d1[k] && d2[k];
			if (__all(lane_done)) break;
			const float4 g0 = s0[j];
			const float2 g1 = s1[j];
			const float dx = g0.x - pxf;
			const float adx2 = (g0.z * dx) * dx;
			const float bdx = g0.w * dx;
			const v2f pw01 = power2(g0.y - py01, g1.x, adx2, bdx);
			const v2f pw23 = power2(g0.y - py23, g1.x, adx2, bdx);
			const float pw[4] = { pw01.x, pw01.y, pw23.x, pw23.y };
			bool hit[4];
#pragma unroll
			for (int k = 0; k < 4; k++) hit[k] = !(d1[k] && d2[k]) && !(pw[k] > 0.0f || pw[k] < -4.5f);
			if (!__any(hit[0] || hit[1] || hit[2] || hit[3])) continue;
			const float4 c1 = sl1[j];
			if (!blending)
			{
#pragma unroll
				for (int k = 0; k < 4; k++)
				{
					if (__any(hit[k]))
					{
						const float alpha = fminf(0.99f, c1.w * fast_exp(pw[k]));
						float w; bool acc;
						blend_px(hit[k], alpha, c1.x, c1.y, c1.z, T1[k], A0[k], A1[k], A2[k], d1[k], w, acc);
					}
				}
			}
			else
			{
				const float4 c2 = sl2[j];
				const bool l2_ok = !((g1.y + 1.0f) < L2f); // the Gaussian exists at level L2
#pragma unroll
				for (int k = 0; k < 4; k++)
				{
					if (__any(hit[k]))
					{
						const float ev = fast_exp(pw[k]);
						float w; bool acc;
						blend_px(hit[k] && !d1[k], fminf(0.99f, c1.w * ev), c1.x, c1.y, c1.z, T1[k], A0[k], A1[k], A2[k], d1[k], w, acc);
						blend_px(hit[k] && !d2[k] && l2_ok, fminf(0.99f, c2.w * ev), c2.x, c2.y, c2.z, T2[k], B0[k], B1[k], B2[k], d2[k], w, acc);
					}
				}
			}
		}
	}

	const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
	const size_t plane = (size_t)a.W * a.H;
#pragma unroll
	for (int k = 0; k < PPL; k++)
	{
		if (!inside[k]) continue;
		const size_t pid = (size_t)a.W * (size_t)(ty * FR_TILE + ry + k * RSTEP) + px;
		float o0 = fmaf(bg0, T1[k], A0[k]), o1 = fmaf(bg1, T1[k], A1[k]), o2 = fmaf(bg2, T1[k], A2[k]);
		if (blending)
		{
			const float q0 = fmaf(bg0, T2[k], B0[k]), q1 = fmaf(bg1, T2[k], B1[k]), q2 = fmaf(bg2, T2[k], B2[k]);
			float x = fabsf(est[k] - ((float)L1 + 0.5f)) / 0.5f;
			x = fmaxf(0.0f, fminf(1.0f, x));
			const float bT = 3 * x * x - 2 * x * x * x;
			const float w1 = 1 - bT;
			o0 = o0 * w1 + q0 * (1.f - w1);
			o1 = o1 * w1 + q1 * (1.f - w1);
			o2 = o2 * w1 + q2 * (1.f - w1);
		}
		a.out_color[pid] = o0;
		a.out_color[plane + pid] = o1;
		a.out_color[2 * plane + pid] = o2;
	}
}

#ifndef FR_RENDER_PPL
#define FR_RENDER_PPL 4
#endif

int launch_render(FwdCtx &c)
{
	const fr_forward_args *a = c.a;
	RenderArgs r;
	r.W = a->W; r.H = a->H; r.gx = c.gx;
	r.ranges = c.img.ranges; r.point_list = c.bin.point_list; r.rec = c.geom.rec; r.lvl = c.geom.lvl;
	r.tile_lv = c.img.tile_lv; r.tile_order = c.img.tile_order; r.T = c.T; r.bg = a->background; r.out_color = a->out_color;
	r.final_T = c.img.final_T; r.n_contrib = c.img.n_contrib;
	r.gaussians_count = a->gaussians_count; r.contributions = a->contributions; r.loss_map = a->loss_map;
	constexpr int PPL = FR_RENDER_PPL;
	const dim3 grid(c.T), block(256 / PPL);
	switch (a->variant)
	{
	case FR_VARIANT_ORIGINAL: hipLaunchKernelGGL((k_render<FR_VARIANT_ORIGINAL, PPL>), grid, block, 0, c.stream, r); break;
	case FR_VARIANT_PCHECK_OBB_SUM: hipLaunchKernelGGL((k_render<FR_VARIANT_PCHECK_OBB_SUM, PPL>), grid, block, 0, c.stream, r); break;
	case FR_VARIANT_PCHECK_OBB: hipLaunchKernelGGL((k_render<FR_VARIANT_PCHECK_OBB, PPL>), grid, block, 0, c.stream, r); break;
	case FR_VARIANT_PCHECK_OBB_MAX: hipLaunchKernelGGL((k_render<FR_VARIANT_PCHECK_OBB_MAX, PPL>), grid, block, 0, c.stream, r); break;
	case FR_VARIANT_PCHECK_OBB_LWMC: hipLaunchKernelGGL((k_render<FR_VARIANT_PCHECK_OBB_LWMC, PPL>), grid, block, 0, c.stream, r); break;
	default: hipLaunchKernelGGL((k_render_fov<PPL>), grid, block, 0, c.stream, r); break;
	}
	return check_launch("render", c.stream, a->debug);
}

} // namespace fr
