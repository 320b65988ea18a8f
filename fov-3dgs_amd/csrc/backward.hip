// Backward pass: per-tile back-to-front gradient traversal, then the per-Gaussian chain rule.
//
// Replaces (reference, paths under fov3dgs/submodules/diff-gaussian-rasterization/cuda_rasterizer/;
// the _pcheck_obb_sum copy differs only by the power<-4.5 skip at backward.cu:495):
//   renderCUDA (backward)     backward.cu:399-557
//   computeCov2DCUDA          backward.cu:144-274
//   preprocessCUDA (backward) backward.cu:346-396  (+ computeColorFromSH bwd :20-139, computeCov3D bwd :278-341)
//
// MI355X design. The reference issues 9 float atomics per contributing (pixel, Gaussian) pair into four [P, .] arrays.
// Here (k_render_bwd) a work item is one band of a tile (one wave, two pixels per lane, as in the forward blend); for
// every list entry that reaches the band the 64 lanes' nine partial sums (three colour sums and the six moments of
// G dL/dalpha about the splat centre: what backward.cu:524-541's five gradients are linear in) are reduced with gfx950's cross-row swaps
// (v_permlane32_swap / v_permlane16_swap fold eight values into two registers in 12 instructions, four DPP steps finish
// each row; 26 instructions instead of 54 for nine separate butterflies) so that the totals end up in nine DIFFERENT
// lanes, which add them with ONE atomic instruction into ONE 64-byte row per Gaussian (`acc`, indexed by the
// Gaussian's position in the forward pass's visible list: dense, zeroed by the forward pass, read back coalesced).
// k_preprocess_bwd then walks the visible list and writes every visible Gaussian's gradient rows: the mean2D / conic / opacity
// gradients from the moments, then the fused cov2D / projection / SH / cov3D chain rule (see there). All gradient stores and the
// zero fill beside k_render_bwd are non-temporal (FR_ST); the narrow tensors are written in whole lines (SmallSet).
#include "common.h"

namespace fr {

typedef float bv2 __attribute__((ext_vector_type(2)));
// x + y of a packed pair as ONE v_add_f32 (left to itself the compiler pairs such sums up: three moves and a packed add for two)
__device__ __forceinline__ float hsum(bv2 a)
{
	float r;
	asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a.x), "v"(a.y));
	return r;
}
__device__ __forceinline__ float bwd_exp(float p) { return __builtin_amdgcn_exp2f(p * 1.4426950408889634f); }

// sum over the wave, valid in lane 63
__device__ __forceinline__ float wave_sum_b(float x)
{
	x = row_sum16(x);
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xA, 0xF, false)); // row_bcast15 -> rows 1, 3
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xC, 0xF, false)); // row_bcast31 -> rows 2, 3
	return x;
}

template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ float dpp_add(float x)
{
	return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWMASK, 0xF, false));
}
// The nine per-lane partial sums of TWO list entries folded together (tools/scratch/fold18_test.hip): 18 totals end up in 18 different
// lanes. a / b: the first / second entry's sums (colour r g b, M10 M01 M20 M11 M02, M00). Every step halves the lanes a value is
// spread over while another value moves into the freed half: cross-half swaps (v_permlane32_swap: lanes 0-31 a's, 32-63 b's), cross-row
// swaps (v_permlane16_swap: rows of 16 lanes hold (a, 2k) (a, 2k+1) (b, 2k) (b, 2k+1)), then within the rows DPP adds with a select
// between two registers each (half-rows of 8: k = 0 | 1 and 2 | 3; quads: which register), two quad steps; the two M00 ride a fold of
// their own. 47 instructions for 18 totals where the one-entry fold takes 38 for 9.
// -> the lane's total (valid where `writer`), whose it is (entry 0 / 1) and which component (0..8) of the gradient-sum row
__device__ __forceinline__ float fold18(const float (&a)[9], const float (&b)[9], const int lane, int &entry, int &comp, bool &writer)
{
	float f[9];
#pragma unroll
	for (int i = 0; i < 9; i++) f[i] = fold32(a[i], b[i]);
	float g[4];
#pragma unroll
	for (int k = 0; k < 4; k++) g[k] = fold16(f[2 * k], f[2 * k + 1]);
	const float s0 = dpp_add<0x128>(g[0]), s1 = dpp_add<0x128>(g[1]), s2 = dpp_add<0x128>(g[2]), s3 = dpp_add<0x128>(g[3]); // row_ror:8
	const bool hi8 = (lane & 8) != 0;
	const float h0 = hi8 ? s1 : s0, h1 = hi8 ? s3 : s2;
	const float u0 = dpp_add<0x141>(h0), u1 = dpp_add<0x141>(h1);    // row_half_mirror: lane l + lane 7 - l of its group of eight
	float m = (lane & 4) ? u1 : u0;
	m = dpp_add<0xB1>(m); m = dpp_add<0x4E>(m);                       // the quad's total in all its lanes
	float z = row_sum16(f[8]);
	z = dpp_add<0x142, 0xA>(z);                                       // rows 1, 3 += lane 15 of rows 0, 2: lane 31 a's M00, lane 63 b's
	const int r = lane >> 4, p = (lane >> 3) & 1, which = (lane >> 2) & 1;
	const bool last = (lane & 31) == 31;
	entry = r >> 1;
	comp = last ? 8 : 2 * (which * 2 + p) + (r & 1);
	writer = last || (lane & 3) == 0;
	return last ? z : m;
}

struct BwdRenderArgs {
	int W, H, gx;
	const uint2 *ranges;
	const uint32_t *render_items; // forward's work items: tile << 3 | band
	uint32_t n_items;
	const uint32_t *point_list;
	const float4 *rec;
	const float *bg;
	const float *final_T;
	const uint32_t *n_contrib;
	const float *dL_dpix;
	float *acc; // [V][16] gradient sums per visible-list entry, see GeomWS::acc
	uint32_t *pairs; // optional diagnostic (fr_backward_args.blend_pairs): [T], (band, entry) pairs evaluated, or null
};

template <bool CUTOFF>
__global__ void __launch_bounds__(64, 6) k_render_bwd(const BwdRenderArgs a)
{
	constexpr int PPL = 2;
	__shared__ float4 s0[64];  // x, y, conic a, conic b
	__shared__ float4 s1[64];  // conic c, opacity, r, g
	__shared__ float4 s2[64];  // b, row of the gradient sums (int bits), tq (the forward blend's threshold on q = -power), -

	if (blockIdx.x >= a.n_items) return;
	const uint32_t item = a.render_items[blockIdx.x]; // longest lists first
	const int tile = (int)(item >> 3), wv = (int)(item & 1u);
	const int tx = tile % a.gx, ty = tile / a.gx;
	const int lane = threadIdx.x;
	const int tid = wv * 64 + lane;
	const int lx = tid & 15;
	const int px = tx * FR_TILE + lx;
	const float pxf = (float)px;
	const uint2 range = a.ranges[tile];
	const int n = (int)(range.y - range.x);
	if (n == 0) return;
	const size_t plane = (size_t)a.W * a.H;
	const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];

	// per pixel (the lane's two pixels as packed pairs, v_pk_* arithmetic): transmittance behind the current entry and the
	// colour accumulated behind it, as its product A with dL/dpixel
	static_assert(PPL == 2, "the lane's pixels are handled as one packed pair");
	bv2 T, Tfin, pyp, A, dp0, dp1, dp2, bgdot;
	int lastc[PPL];
	int wave_last = 0;
#pragma unroll
	for (int k = 0; k < PPL; k++)
	{
		const int py = ty * FR_TILE + tile_row<PPL>(tid, k);
		pyp[k] = (float)py;
		const bool inside = px < a.W && py < a.H;
		const size_t pid = (size_t)a.W * py + px;
		Tfin[k] = inside ? a.final_T[pid] : 0.0f;
		lastc[k] = inside ? (int)a.n_contrib[pid] : 0;
		wave_last = max(wave_last, lastc[k]);
		dp0[k] = inside ? a.dL_dpix[pid] : 0.0f;
		dp1[k] = inside ? a.dL_dpix[plane + pid] : 0.0f;
		dp2[k] = inside ? a.dL_dpix[2 * plane + pid] : 0.0f;
	}
	T = Tfin;
	bgdot = bg0 * dp0 + bg1 * dp1 + bg2 * dp2;
	A = (bv2){ 0.f, 0.f };
	// nothing behind the deepest contributor of this wave's pixels needs to be visited
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, off));
	if (wave_last == 0) return;

	// walk the list back to front from there, 64 entries per batch; the next batch's records are prefetched into
	// registers while this one is processed
	float4 p0 = make_float4(0, 0, 0, 0), p1 = p0;
	float2 p2 = make_float2(0, 0);
	auto fetch = [&](int e)
	{
		const uint32_t id = a.point_list[range.x + e]; // the item: its record and its row of gradient sums share the index
		const float4 *r = a.rec + 3 * (size_t)id;
		p0 = r[0]; p1 = r[1];
		p2 = make_float2(r[2].x, __uint_as_float(id));
	};
	// the forward blend's skip tests (k_render: outside the support, alpha < 1/255) as its ONE threshold on q = -power: the very same
	// expression, so that the backward pass takes the gradient of exactly the pairs the forward pass blended
	auto q_threshold = [&](float opacity, float &lq) { lq = logf(255.0f * opacity); return lq >= 0.0f ? (CUTOFF ? fminf(4.5f, lq) : lq) : 0.0f; };
	if (lane < wave_last) fetch(wave_last - 1 - lane);
	uint32_t npairs = 0;
	// an entry's nine sums wait for the next entry's: two entries are folded together (fold18)
	float pend[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	int pend_row = 0;
	bool have_pend = false;
	auto fold_one = [&](const float (&v)[9], const int row) __attribute__((always_inline))
	{
		// eight sums folded into two registers: afterwards row r of 16 lanes holds value 2 r in t0 and 2 r + 1 in t1
		const float f0 = fold32(v[0], v[4]), f1 = fold32(v[1], v[5]), f2 = fold32(v[2], v[6]), f3 = fold32(v[3], v[7]);
		const float t0 = row_sum16(fold16(f0, f2)), t1 = row_sum16(fold16(f1, f3));
		const float t8 = wave_sum_b(v[8]); // lane 63
		const int sel = lane & 15;
		const bool writer = sel < 2 || lane == 63;
		if (writer)
		{
			const int comp = lane == 63 ? 8 : 2 * (lane >> 4) + sel;
			const float val = lane == 63 ? t8 : (sel ? t1 : t0);
			atomicAdd(a.acc + 16 * (size_t)row + comp, val); // nine lanes, one cache line
		}
	};
	for (int top = wave_last; top > 0; top -= 64)
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the previous batch has been read by all lanes
		__builtin_amdgcn_wave_barrier();
		const bool staged = lane < min(64, top);
		float lq;
		const float tq = q_threshold(p1.y, lq);
		if (staged) { s0[lane] = p0; s1[lane] = p1; s2[lane] = make_float4(p2.x, p2.y, tq, 0.0f); }
		unsigned long long reach;
		{
			// entries that cannot touch this wave's rows are skipped: same thresholds as the per-pixel tests below
			// (power < -4.5, alpha < 1/255 <=> power < -ln(255 opacity)), see splat_reaches()
			const float thr_a = -lq - 0.01f;
			const float thr = CUTOFF ? fmaxf(-4.5f, thr_a) : thr_a;
			reach = __ballot(staged && lq >= 0.0f && band_reaches<PPL>(wv, tx, ty, p0.x, p0.y, p0.z, p0.w, p1.x, thr));
			npairs += (uint32_t)__popcll(reach);
			}
		if (top - 64 - lane > 0) fetch(top - 64 - 1 - lane);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // lanes read entries other lanes staged
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		for (unsigned long long rm = reach; rm; rm &= rm - 1)
		{
			const int j = __builtin_ctzll(rm);
			const int pos = top - 1 - j; // 0-based position in the tile list
			const float4 g0 = s0[j];
			const float4 g1 = s1[j];
			const float4 g2 = s2[j];
			const uint32_t tqb = __float_as_uint(g2.z);
			const float dx = g0.x - pxf;
			const float adx2 = (g0.z * dx) * dx;
			const float bdx = g0.w * dx;
			const bv2 dy = g0.y - pyp;
			const bv2 s = __builtin_elementwise_fma(g1.x * dy, dy, (bv2){ adx2, adx2 });
			const bv2 q = __builtin_elementwise_fma((bv2){ 0.5f, 0.5f }, s, bdx * dy); // -power, bit for bit (render.hip qform2)
			const bv2 pe = q * -1.4426950408889634f;
			const bv2 G = (bv2){ __builtin_amdgcn_exp2f(pe.x), __builtin_amdgcn_exp2f(pe.y) };
			bv2 alpha = g1.y * G;
			alpha.x = fminf(0.99f, alpha.x); alpha.y = fminf(0.99f, alpha.y);
			// backward.cu:468-497: behind the pixel's last contributor / outside the support / negligible. Everything
			// below is predicated: a pixel that is not `on` keeps its state and adds exact zeros (its G is cleared first,
			// so no infinity of a far-away splat can meet a zero factor).
			const bool on_x = pos < lastc[0] && __float_as_uint(q.x) <= tqb; // 0 <= q <= tq: in the support and alpha >= 1/255
			const bool on_y = pos < lastc[1] && __float_as_uint(q.y) <= tqb;
			const bool any = on_x || on_y;
			float v[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }; // colour r g b, moments M10 M01 M20 M11 M02 M00 of G dL/dalpha
			if (__any(any))
			{
				// A pixel that is not `on` takes part with alpha = 0 and G = 0: 1 / (1 - 0) = 1 leaves its transmittance, 0 c + 1 A its
				// colour behind, and every sum gets an exact zero (G is cleared, so no infinity of a far-away splat meets a zero factor).
				const bv2 am = (bv2){ on_x ? alpha.x : 0.0f, on_y ? alpha.y : 0.0f };
				const bv2 Gm = (bv2){ on_x ? G.x : 0.0f, on_y ? G.y : 0.0f };
				const bv2 om = 1.0f - am;
				// T / (1 - alpha) (backward.cu:503: the transmittance in front of the entry, recovered by division -- thousands of times
				// in a row on a deep list, so every rounding counts): q0 = T * rcp(1 - alpha) is good to ~1.5 ulp; one residual step,
				// q = q0 + (T - (1 - alpha) q0) * rcp, leaves the quotient within half an ulp and a bit, like the reference's division
				// (the reference divides twice per pixel, ~24 instructions; a Newton step on the reciprocal instead -- round 4 -- costs
				// the same two instructions as this one and left T 4 x noisier than the reference's on lists 5 000 entries deep:
				// tests/test_full_size_parity.py, S-6M-T, dL_dcolor against the double-precision oracle)
				const bv2 r = (bv2){ __builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y) };
				const bv2 q0 = T * r;
				const bv2 Tn = __builtin_elementwise_fma(__builtin_elementwise_fma(-om, q0, T), r, q0);
				T = Tn;
				const bv2 wgt = am * Tn;                                       // d channel / d colour
				// (the lane's two pixels' terms as a product and a fused multiply-add: two plain instructions where a packed product and
				// the sum of its halves are three instruction slots)
				v[0] = fmaf(wgt.y, dp0.y, wgt.x * dp0.x); v[1] = fmaf(wgt.y, dp1.y, wgt.x * dp1.x); v[2] = fmaf(wgt.y, dp2.y, wgt.x * dp2.x);
				// backward.cu:507-514 keeps the colour accumulated behind the entry per channel and forms sum_ch (c_ch - behind_ch) dpix_ch;
				// here the products with dpix are taken first: cdot = c . dpix of this entry, A = behind . dpix -- one state per pixel
				// instead of three, updated with the entry itself once its gradient has been taken (the reference's last_alpha /
				// last_color are the same update, applied one entry later)
				const bv2 cdot = __builtin_elementwise_fma((bv2){ g2.x, g2.x }, dp2, __builtin_elementwise_fma((bv2){ g1.w, g1.w }, dp1, g1.z * dp0));
				bv2 dA = (cdot - A) * Tn;
				dA = __builtin_elementwise_fma(-(Tfin * r), bgdot, dA);         // the background's share
				A = __builtin_elementwise_fma(am, cdot, om * A);
				// the six moments of G dL/dalpha about the splat's centre: what the gradients of the mean, the conic and the opacity are
				// linear in (backward.cu:524-541 forms the products per pixel; k_preprocess_bwd forms them once per Gaussian)
				// A lane's two pixels share their column, i.e. dx: the x factors are taken out of the lane's sums (M10 = dx (w0 + w1),
				// M20 = dx M10, M11 = dx M01): three packed products and three plain ones where six packed products were
				const bv2 w = Gm * dA;
				const float m00 = hsum(w);
				const float m10 = m00 * dx, m20 = m10 * dx;
				const bv2 wy = w * dy;
				const float m01 = hsum(wy);
				const float m11 = m01 * dx;
				const float m02 = fmaf(wy.y, dy.y, wy.x * dy.x);
				v[3] = m10; v[4] = m01; v[5] = m20; v[6] = m11; v[7] = m02; v[8] = m00;
			}
			if (__any(any))
			{
				const int row = __float_as_int(g2.y); // the entry's row of gradient sums (wave-uniform)
				if (!have_pend)
				{
			#pragma unroll
					for (int i = 0; i < 9; i++) pend[i] = v[i];
					pend_row = row; have_pend = true;
				}
				else
				{
					int e, comp; bool writer;
					const float val = fold18(pend, v, lane, e, comp, writer);
					if (writer) atomicAdd(a.acc + 16 * (size_t)(e ? row : pend_row) + comp, val); // eighteen lanes, two cache lines
					have_pend = false;
				}
			}
		}
	}
	if (have_pend) fold_one(pend, pend_row); // an odd entry is left
	if (a.pairs != nullptr && lane == 0 && npairs != 0) atomicAdd(a.pairs + tile, npairs);
}

// ---- per-Gaussian chain rule ------------------------------------------------------------------
// Written from the matrix form of the forward pass rather than entry by entry.
//   forward.cu:74-113:  M = U Sigma U^T (2x2), U = Jac R (2x3): Jac = d(pixel) / d(camera point) with the clamped
//                        tangents, R = world-to-camera rotation, Sigma the 3D covariance;  a = M00 + 0.3, b = M01,
//                        c = M11 + 0.3;  conic Q = [[a b][b c]]^-1.
//   Given Ghat = [[gA gB][gB gC]], the gradient of the loss w.r.t. the conic as k_render_bwd sums it (gB is half the
//   derivative w.r.t. the off-diagonal parameter, backward.cu:536-538):
//       H  := dL/dM = -Q Ghat Q   (from d(X^-1) = -X^-1 dX X^-1; symmetric)
//       dL/dSigma = U^T H U,      dL/dU = 2 H U Sigma,      dL/dJac = dL/dU R^T,
//   and dL/dt through the four non-constant entries of Jac. The reference writes the same chain out per matrix entry
//   (backward.cu:144-274) with the quotient rule on the recomputed covariance; its 1e-7 guard on det^2 is kept as the
//   factor `guard` below.
// The NARROW dense gradient tensors (4-16 bytes a row: positions, opacity, scales, rotations, the DC coefficients of split SH storage).
// A visible Gaussian's row in such a tensor is a fraction of a 64-byte memory transaction, and with one Gaussian in three visible the
// rows of a wave are 64 partial transactions per tensor over zeros the fill had just written: 124 us of k_preprocess_bwd's 516 for 6 %
// of its payload (timed by leaving the stores out). So the rows of these tensors are written TOGETHER WITH THE ZEROS AROUND THEM, as
// whole lines, by the kernel that has the rows: the tensors are cut into groups of 32 rows; a group without a visible Gaussian
// (radii > 0) is cleared by k_fill_groups beside k_render_bwd like everything else; in any other group every visible row's owner writes
// the zeros in front of it (back to the previous visible row of the group) and, when it is the group's last, those behind it -- a
// wave's 64 consecutive visible rows and their zeros are one contiguous range, put together in LDS and stored 16 bytes a lane.
// off: where the tensor's row sits in a lane's staging row of FR_SMALL_ROW floats.
#define FR_SMALL_MAX 6
#define FR_SMALL_ROW 17      // 3 mean2D + 1 opacity + 3 mean3D + 3 scale + 4 rotation + 3 DC (odd: lanes' rows start in different LDS banks)
#define FR_SMALL_GROUPS 16   // 32-row groups (512 rows) a wave puts together at a time; a wave whose rows span more takes several turns
struct SmallSet { int n; float *p[FR_SMALL_MAX]; int w[FR_SMALL_MAX]; int off[FR_SMALL_MAX]; };

struct BwdPreArgs {
	int P, D, M, W, H;
	float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
	const float *means3D, *scales, *rotations, *shs, *shs_rest, *cov3D_precomp, *colors_precomp;
	const float *viewmatrix, *projmatrix, *campos;
	const int *radii;
	const float4 *rec;
	const float *cov3D_ws;
	float4 *acc;               // gradient sums per visible-list entry (k_render_bwd); every row is cleared again once it has been read
	float *dL_dmean2D, *dL_dconic, *dL_dcolor, *dL_dopacity; // dense [P, .] outputs filled from them (dL_dconic / dL_dcolor optional)
	float *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dsh_rest, *dL_dscale, *dL_drot;
	const uint32_t *vis_list;  // forward's compact list of projected Gaussians
	const uint32_t *vis_count; // its length (device)
	const uint32_t *lrange;    // per item: FR_ITEM_NONE when it landed in no tile (k_bin)
	int raw;                   // dL_dscale / dL_drot / dL_dopacity w.r.t. the model's raw parameters (fr_backward_args.raw_activations)
	int row_sparse;            // the outputs are COMPACT: row i belongs to the Gaussian vis_list[i] (fr_backward_args.row_sparse)
	int M0;                    // coefficients in dL_dsh's rows ([., M0, 3]): M, or 1 with split SH storage
	SmallSet small;            // dense tensors: the narrow ones this kernel writes in whole lines (n == 0: every row on its own)
	const uint32_t *range_bounds; // fr_backward_args.num_ranges > 1: [K + 1] first visible-list entry of every range of rows (k_range_bounds) ...
	int range_k;                  // ... and the range this launch covers; -1: the whole list
};

// The per-Gaussian pass in pieces (fr_backward_args.num_ranges): range k covers the rows (Gaussian indices) [row_lo(k), row_lo(k + 1)),
// row_lo(k) = (P k / K) rounded down to a multiple of 32 -- a group of SmallSet's whole-line scheme never straddles two ranges. The
// visible list is in index order, so a range of rows is a range of list entries: out[k] = first entry whose Gaussian index is
// >= row_lo(k) (one binary search per range: twenty dependent loads, on the helper stream beside k_render_bwd), out[K] = V.
#define FR_MAX_RANGES 16
__host__ __device__ __forceinline__ int range_row_lo(int P, int k, int K) { return k >= K ? P : (int)(((long long)P * k / K) & ~31ll); }
__global__ void k_range_bounds(const uint32_t *vis_list, const uint32_t *vis_count, int P, int K, uint32_t *out)
{
	const int k = threadIdx.x;
	if (k > K) return;
	const int V = (int)*vis_count;
	if (k == K) { out[k] = (uint32_t)V; return; }
	const uint32_t want = (uint32_t)range_row_lo(P, k, K);
	int lo = 0, hi = V; // first entry in [0, V] with vis_list[entry] >= want
	while (lo < hi) { const int mid = (lo + hi) >> 1; if (vis_list[mid] < want) lo = mid + 1; else hi = mid; }
	out[k] = (uint32_t)lo;
}

// The gradient rows are written once and read by nobody in this library, and so are the zeros of k_fill_zero: both leave with
// NON-TEMPORAL stores (`global_store ... nt`: the lines are first to go from L2). Plain stores left 1.5 GB of dirty zero lines and
// 0.5 GB of rows pushing k_render_bwd's records and this kernel's own reads out of the cache: k_render_bwd beside the fill 0.48 -> 0.43
// ms, k_preprocess_bwd 0.51 -> 0.41, the fill itself 0.365 -> 0.30 (S-6M, A / B on one box). Non-temporal LOADS of the rows this
// kernel reads once (stash, sums, records, coefficients) cost 30 us instead: left as they are.
typedef float nt_f4v __attribute__((ext_vector_type(4)));
#define FR_ST(p, v) __builtin_nontemporal_store((v), (p))
__device__ __forceinline__ void st_f4(float4 *p, const float4 v) { __builtin_nontemporal_store((nt_f4v){ v.x, v.y, v.z, v.w }, (nt_f4v *)p); }
struct V3 { float x, y, z; };
__device__ __forceinline__ float dot3(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 axpy3(float s, const V3 &a, const V3 &b) { return { s * a.x + b.x, s * a.y + b.y, s * a.z + b.z }; }
// y = S x for the symmetric matrix stored as (xx, xy, xz, yy, yz, zz)
__device__ __forceinline__ V3 symv(const float *S, const V3 &v)
{
	return { S[0] * v.x + S[1] * v.y + S[2] * v.z, S[1] * v.x + S[3] * v.y + S[4] * v.z, S[2] * v.x + S[4] * v.y + S[5] * v.z };
}

// Real spherical-harmonics basis up to degree 3 in the reference's sign convention (forward.cu:30-59) and its
// gradient w.r.t. the unit direction: colour = sum_k basis_k sh_k, so dL/dsh_k = basis_k g and
// dL/ddir = sum_k (sh_k . g) grad basis_k -- one dot product and three multiply-adds per coefficient.
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float *bas /*[16]*/, V3 *grd /*[16]*/)
{
	bas[0] = FR_SH_C0; grd[0] = { 0, 0, 0 };
	if (deg < 1) return;
	bas[1] = -FR_SH_C1 * y; grd[1] = { 0, -FR_SH_C1, 0 };
	bas[2] = FR_SH_C1 * z;  grd[2] = { 0, 0, FR_SH_C1 };
	bas[3] = -FR_SH_C1 * x; grd[3] = { -FR_SH_C1, 0, 0 };
	if (deg < 2) return;
	const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
	bas[4] = FR_SH_C2_0 * xy;                  grd[4] = { FR_SH_C2_0 * y, FR_SH_C2_0 * x, 0 };
	bas[5] = FR_SH_C2_1 * yz;                  grd[5] = { 0, FR_SH_C2_1 * z, FR_SH_C2_1 * y };
	bas[6] = FR_SH_C2_2 * (2.f * zz - xx - yy); grd[6] = { -2.f * FR_SH_C2_2 * x, -2.f * FR_SH_C2_2 * y, 4.f * FR_SH_C2_2 * z };
	bas[7] = FR_SH_C2_3 * xz;                  grd[7] = { FR_SH_C2_3 * z, 0, FR_SH_C2_3 * x };
	bas[8] = FR_SH_C2_4 * (xx - yy);           grd[8] = { 2.f * FR_SH_C2_4 * x, -2.f * FR_SH_C2_4 * y, 0 };
	if (deg < 3) return;
	bas[9] = FR_SH_C3_0 * y * (3.f * xx - yy);            grd[9] = { 6.f * FR_SH_C3_0 * xy, 3.f * FR_SH_C3_0 * (xx - yy), 0 };
	bas[10] = FR_SH_C3_1 * xy * z;                        grd[10] = { FR_SH_C3_1 * yz, FR_SH_C3_1 * xz, FR_SH_C3_1 * xy };
	bas[11] = FR_SH_C3_2 * y * (4.f * zz - xx - yy);      grd[11] = { -2.f * FR_SH_C3_2 * xy, FR_SH_C3_2 * (4.f * zz - xx - 3.f * yy), 8.f * FR_SH_C3_2 * yz };
	bas[12] = FR_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy); grd[12] = { -6.f * FR_SH_C3_3 * xz, -6.f * FR_SH_C3_3 * yz, 3.f * FR_SH_C3_3 * (2.f * zz - xx - yy) };
	bas[13] = FR_SH_C3_4 * x * (4.f * zz - xx - yy);      grd[13] = { FR_SH_C3_4 * (4.f * zz - 3.f * xx - yy), -2.f * FR_SH_C3_4 * xy, 8.f * FR_SH_C3_4 * xz };
	bas[14] = FR_SH_C3_5 * z * (xx - yy);                 grd[14] = { 2.f * FR_SH_C3_5 * xz, -2.f * FR_SH_C3_5 * yz, FR_SH_C3_5 * (xx - yy) };
	bas[15] = FR_SH_C3_6 * x * (xx - 3.f * yy);           grd[15] = { 3.f * FR_SH_C3_6 * (xx - yy), -6.f * FR_SH_C3_6 * xy, 0 };
}

// slot: the Gaussian's position in the visible list (its row of gradient sums; with `stash` != null also its 64-byte row
// (xyz | raw scale | rotation | 3D covariance) the forward pass left there: one coalesced row instead of four gathers)
// sh_row: the Gaussian's 45 rest coefficients in LDS (k_preprocess_bwd fetched the wave's rows together), replaced in place by
// their gradients (stored together as well); null: read / written here, per lane
// sv: the lane's staging row of the narrow tensors (BwdPreArgs.small; null: every row stored on its own)
__device__ __forceinline__ void preprocess_bwd_one(const BwdPreArgs &a, const int idx, const int slot, const float4 *stash, float *sh_row, float *sv)
{
	const size_t orow = a.row_sparse ? (size_t)slot : (size_t)idx; // the row of the gradient tensors this Gaussian's gradients go to
	const float *vm = a.viewmatrix, *pm = a.projmatrix;
	float4 st0 = make_float4(0, 0, 0, 0), st1 = st0, st2 = st0, st3 = st0;
	if (stash != nullptr) { st0 = stash[0]; st1 = stash[1]; st2 = stash[2]; st3 = stash[3]; }
	const V3 mean = stash ? V3{ st0.x, st0.y, st0.z } : V3{ a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2] };
	float Sigma[6];
	if (stash) { Sigma[0] = st2.z; Sigma[1] = st2.w; Sigma[2] = st3.x; Sigma[3] = st3.y; Sigma[4] = st3.z; Sigma[5] = st3.w; }
	else
	{
#pragma unroll
		for (int i = 0; i < 6; i++) Sigma[i] = a.cov3D_precomp[6 * (size_t)idx + i];
	}
	// what k_render_bwd summed for this Gaussian
	const float4 ac0 = a.acc[4 * (size_t)slot], ac1 = a.acc[4 * (size_t)slot + 1], ac2 = a.acc[4 * (size_t)slot + 2];
	// fr_backward is idempotent (the reference allocates fresh zeros per call, rasterize_points.cu:171-179): the row is read
	// exactly once per call, so the reader leaves it cleared for a second backward pass over the same forward state
	// (retain_graph, torch.autograd.grad twice). The last quarter (1 / |raw quaternion|) belongs to the forward pass and stays.
	a.acc[4 * (size_t)slot] = a.acc[4 * (size_t)slot + 1] = a.acc[4 * (size_t)slot + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
	const float g_col[3] = { ac0.x, ac0.y, ac0.z };
	// k_render_bwd sums the moments M_ij = sum over pixels of G dL/dalpha dx^i dy^j (dx, dy: centre - pixel); with the conic Q and the
	// opacity o of the forward pass (backward.cu:524-541): dL/dmean2D = -o Q (M10, M01) * (W / 2, H / 2), dL/dconic = -o/2 (M20, M11, M02)
	// (the off-diagonal one counted once), dL/dopacity = M00
	const float4 rq0 = a.rec[3 * (size_t)slot], rq1 = a.rec[3 * (size_t)slot + 1];
	const float m10 = ac0.w, m01 = ac1.x, opac = rq1.y;
	const float g_px = -opac * (rq0.z * m10 + rq0.w * m01) * (0.5f * a.W), g_py = -opac * (rq1.x * m01 + rq0.w * m10) * (0.5f * a.H);
	const float gA = -0.5f * opac * ac1.y, gB = -0.5f * opac * ac1.z, gC = -0.5f * opac * ac1.w;
	if (sv) { sv[0] = g_px; sv[1] = g_py; sv[2] = 0.0f; }
	else { FR_ST(a.dL_dmean2D + 3 * orow, g_px); FR_ST(a.dL_dmean2D + 3 * orow + 1, g_py); }
	if (a.row_sparse) a.dL_dmean2D[3 * orow + 2] = 0.0f; // (the dense tensors get their zeros from the fill)
	// (raw parameters: through the sigmoid, o (1 - o), and below through exp and the normalisation -- what k_activate_bwd does
	// for all P Gaussians, here only for the rows that are not zero anyway)
	{
		const float g_op = a.raw ? ac2.x * opac * (1.0f - opac) : ac2.x;
		if (sv) sv[3] = g_op; else FR_ST(a.dL_dopacity + orow, g_op);
	}
	if (a.dL_dcolor != nullptr) { FR_ST(a.dL_dcolor + 3 * orow, g_col[0]); FR_ST(a.dL_dcolor + 3 * orow + 1, g_col[1]); FR_ST(a.dL_dcolor + 3 * orow + 2, g_col[2]); }
	if (a.dL_dconic != nullptr) { FR_ST(a.dL_dconic + 4 * orow, gA); FR_ST(a.dL_dconic + 4 * orow + 1, gB); FR_ST(a.dL_dconic + 4 * orow + 3, gC); if (a.row_sparse) a.dL_dconic[4 * orow + 2] = 0.0f; }

	// rows of the camera rotation: t = R mean + translation, R[i][r] = vm[4 r + i]
	const V3 Rx = { vm[0], vm[4], vm[8] }, Ry = { vm[1], vm[5], vm[9] }, Rz = { vm[2], vm[6], vm[10] };
	const float tz = dot3(Rz, mean) + vm[14];
	const float tx_raw = dot3(Rx, mean) + vm[12], ty_raw = dot3(Ry, mean) + vm[13];
	const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
	const float ux = tx_raw / tz, uy = ty_raw / tz;
	const bool clamp_x = ux < -limx || ux > limx, clamp_y = uy < -limy || uy > limy; // forward.cu:82-87: no gradient through a clamped tangent
	const float tx = fminf(limx, fmaxf(-limx, ux)) * tz, ty = fminf(limy, fmaxf(-limy, uy)) * tz;
	const float fx = a.focal_x, fy = a.focal_y, iz = 1.f / tz, iz2 = iz * iz;
	// U = Jac R, rows u0, u1
	const float j00 = fx * iz, j02 = -(fx * tx) * iz2, j11 = fy * iz, j12 = -(fy * ty) * iz2;
	const V3 u0 = axpy3(j00, Rx, { j02 * Rz.x, j02 * Rz.y, j02 * Rz.z });
	const V3 u1 = axpy3(j11, Ry, { j12 * Rz.x, j12 * Rz.y, j12 * Rz.z });
	// H = -Q Ghat Q with the conic Q the forward pass stored
	const float4 rc0 = rq0, rc1 = rq1; // (..., conic a, conic b), (conic c, opacity, ...)
	const float qa = rc0.z, qb = rc0.w, qc = rc1.x;
	const float k00 = qa * gA + qb * gB, k01 = qa * gB + qb * gC, k10 = qb * gA + qc * gB, k11 = qb * gB + qc * gC; // Q Ghat
	const float detq = qa * qc - qb * qb;                     // = 1 / det M
	const float guard = -1.0f / (1.0f + 0.0000001f * detq * detq); // -(det^2 / (det^2 + 1e-7)), backward.cu:190
	const float h00 = guard * (k00 * qa + k01 * qb), h01 = guard * (k00 * qb + k01 * qc), h11 = guard * (k10 * qb + k11 * qc);
	// Y = H U (rows y0, y1); dL/dSigma = U^T Y, packed with the off-diagonals counted twice
	const V3 y0 = axpy3(h00, u0, { h01 * u1.x, h01 * u1.y, h01 * u1.z });
	const V3 y1 = axpy3(h01, u0, { h11 * u1.x, h11 * u1.y, h11 * u1.z });
	float gSigma[6];
	gSigma[0] = u0.x * y0.x + u1.x * y1.x;
	gSigma[3] = u0.y * y0.y + u1.y * y1.y;
	gSigma[5] = u0.z * y0.z + u1.z * y1.z;
	gSigma[1] = (u0.x * y0.y + u1.x * y1.y) + (u0.y * y0.x + u1.y * y1.x);
	gSigma[2] = (u0.x * y0.z + u1.x * y1.z) + (u0.z * y0.x + u1.z * y1.x);
	gSigma[4] = (u0.y * y0.z + u1.y * y1.z) + (u0.z * y0.y + u1.z * y1.y);
	if (a.dL_dcov3D != nullptr)
	{
#pragma unroll
		for (int i = 0; i < 6; i++) FR_ST(a.dL_dcov3D + 6 * orow + i, gSigma[i]);
	}
	// dL/dU = 2 Y Sigma (rows w0, w1); dL/dJac[i][k] = dL/dU row i . R row k
	const V3 w0 = symv(Sigma, y0), w1 = symv(Sigma, y1);
	const float gj00 = 2.f * dot3(w0, Rx), gj02 = 2.f * dot3(w0, Rz), gj11 = 2.f * dot3(w1, Ry), gj12 = 2.f * dot3(w1, Rz);
	// Jac = [[fx/tz, 0, -fx tx/tz^2], [0, fy/tz, -fy ty/tz^2]]
	const float iz3 = iz2 * iz;
	const float g_tx = clamp_x ? 0.f : -fx * iz2 * gj02;
	const float g_ty = clamp_y ? 0.f : -fy * iz2 * gj12;
	const float g_tz = -iz2 * (fx * gj00 + fy * gj11) + 2.f * iz3 * (fx * tx * gj02 + fy * ty * gj12);
	// back through t = R mean + translation
	V3 g_mean = { Rx.x * g_tx + Ry.x * g_ty + Rz.x * g_tz, Rx.y * g_tx + Ry.y * g_ty + Rz.y * g_tz, Rx.z * g_tx + Ry.z * g_ty + Rz.z * g_tz };

	// ---- screen position: pixel = ((p_hom.xy / p_hom.w) + 1) S / 2 (backward.cu:370-387; the S / 2 is already in g_px, g_py)
	{
		const float hx = pm[0] * mean.x + pm[4] * mean.y + pm[8] * mean.z + pm[12];
		const float hy = pm[1] * mean.x + pm[5] * mean.y + pm[9] * mean.z + pm[13];
		const float hw = pm[3] * mean.x + pm[7] * mean.y + pm[11] * mean.z + pm[15];
		const float iw = 1.0f / (hw + 0.0000001f);
		const float nx = hx * iw * iw, ny = hy * iw * iw; // d(h.x / w) / d w = -h.x / w^2
		g_mean.x += (pm[0] * iw - pm[3] * nx) * g_px + (pm[1] * iw - pm[3] * ny) * g_py;
		g_mean.y += (pm[4] * iw - pm[7] * nx) * g_px + (pm[5] * iw - pm[7] * ny) * g_py;
		g_mean.z += (pm[8] * iw - pm[11] * nx) * g_px + (pm[9] * iw - pm[11] * ny) * g_py;
	}
	// ---- colour from spherical harmonics (backward.cu:20-139)
	if (a.colors_precomp == nullptr && a.shs != nullptr)
	{
		// Coefficients are read and their gradients written 16 bytes at a time in the concatenated order
		// (k, channel) -> 3 k + channel: every lane works on its own Gaussian, i.e. its own cache lines, and 48 dword
		// loads + 48 dword stores per Gaussian kept the address unit busy for most of this kernel.
		// Split storage (shs = DC [P,1,3], shs_rest = [P,M-1,3]): slots 3.. come from / go to the rest tensors.
		const bool split = a.shs_rest != nullptr;
		const int nrest = split ? (a.M - 1) * 3 : a.M * 3 - 3;     // floats available after the DC triple
		const float *sh_r = split ? a.shs_rest + (size_t)idx * (a.M - 1) * 3 : a.shs + (size_t)idx * a.M * 3 + 3;
		float *dsh_r = split ? a.dL_dsh_rest + orow * (a.M - 1) * 3 : a.dL_dsh + orow * a.M * 3 + 3;
		float *dsh0 = split ? a.dL_dsh + 3 * orow : a.dL_dsh + orow * a.M * 3;
		const int nuse = 3 * ((a.D + 1) * (a.D + 1)) - 3; // rest floats the active degree reads / writes
		// a channel clamped at zero in the forward pass passes no gradient (forward.cu:63-70)
		const uint32_t clamp_bits = __float_as_uint(a.rec[3 * (size_t)slot + 2].z);
		const float g[3] = { (clamp_bits & 1u) ? 0.f : g_col[0], (clamp_bits & 2u) ? 0.f : g_col[1], (clamp_bits & 4u) ? 0.f : g_col[2] };
		const V3 off = { mean.x - a.campos[0], mean.y - a.campos[1], mean.z - a.campos[2] };
		const float len2 = dot3(off, off), ilen = 1.0f / sqrtf(len2);
		const V3 dir = { off.x * ilen, off.y * ilen, off.z * ilen };
		float bas[16];
		V3 grd[16];
#pragma unroll
		for (int k = 0; k < 16; k++) { bas[k] = 0.f; grd[k] = { 0, 0, 0 }; }
		sh_basis_grad(a.D, dir.x, dir.y, dir.z, bas, grd);
		if (sv && split) { sv[14] = bas[0] * g[0]; sv[15] = bas[0] * g[1]; sv[16] = bas[0] * g[2]; }
		else { FR_ST(dsh0, bas[0] * g[0]); FR_ST(dsh0 + 1, bas[0] * g[1]); FR_ST(dsh0 + 2, bas[0] * g[2]); }
		V3 g_dir = { 0, 0, 0 };
		if (sh_row != nullptr)
		{
			// coefficient k's triple out of the LDS row, its gradient triple bas_k g back into the same three words (basis functions
			// beyond the active degree are zero: so are their gradients)
#pragma unroll
			for (int k = 1; k < 16; k++)
			{
				float *c = sh_row + 3 * (k - 1);
				if (3 * k <= nuse) // (a coefficient of the active degree: the others were not fetched)
				{
					const float wk = c[0] * g[0] + c[1] * g[1] + c[2] * g[2];
					g_dir = axpy3(wk, grd[k], g_dir);
				}
				c[0] = bas[k] * g[0]; c[1] = bas[k] * g[1]; c[2] = bas[k] * g[2];
			}
		}
		else
		{
		typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
		float coef[48];
#pragma unroll
		for (int i = 0; i < 48; i++) coef[i] = 0.0f;
		if (nrest >= 45)
		{
#pragma unroll
			for (int q = 0; q < 11; q++) { const f4u t = *(const f4u *)(sh_r + 4 * q); coef[3 + 4 * q] = t.x; coef[4 + 4 * q] = t.y; coef[5 + 4 * q] = t.z; coef[6 + 4 * q] = t.w; }
			coef[47] = sh_r[44];
		}
		else
		{
#pragma unroll
			for (int i = 0; i < 45; i++) if (i < nuse) coef[3 + i] = sh_r[i];
		}
#pragma unroll
		for (int k = 1; k < 16; k++)
		{
			const float wk = coef[3 * k] * g[0] + coef[3 * k + 1] * g[1] + coef[3 * k + 2] * g[2];
			g_dir = axpy3(wk, grd[k], g_dir);
		}
		// gradients of the rest coefficients of the active degree (the others stay zero), formed as they are stored:
		// flat index f = 3 k + channel -> basis_k * g_channel (no second 48-register array)
#define FR_GSH(f) (bas[(f) / 3] * g[(f) % 3])
#pragma unroll
		for (int q = 0; q < 12; q++)
		{
			if (4 * q + 4 <= nuse) *(f4u *)(dsh_r + 4 * q) = (f4u){ FR_GSH(3 + 4 * q), FR_GSH(4 + 4 * q), FR_GSH(5 + 4 * q), FR_GSH(6 + 4 * q) };
			else
			{
#pragma unroll
				for (int j = 0; j < 4; j++) if (4 * q + j < nuse && 3 + 4 * q + j < 48) dsh_r[4 * q + j] = FR_GSH(3 + 4 * q + j);
			}
		}
#undef FR_GSH
		// (compact rows: the coefficients beyond the active degree get their zeros here -- the dense tensors' come from the fill)
		if (a.row_sparse)
			for (int f = nuse; f < nrest; f++) dsh_r[f] = 0.0f;
		}
		// dir = off / |off|: d dir / d off = (I - dir dir^T) / |off|
		const float along = dot3(dir, g_dir);
		g_mean.x += (g_dir.x - along * dir.x) * ilen;
		g_mean.y += (g_dir.y - along * dir.y) * ilen;
		g_mean.z += (g_dir.z - along * dir.z) * ilen;
	}
	if (sv) { sv[4] = g_mean.x; sv[5] = g_mean.y; sv[6] = g_mean.z; }
	else
	{
		FR_ST(a.dL_dmean3D + 3 * orow, g_mean.x);
		FR_ST(a.dL_dmean3D + 3 * orow + 1, g_mean.y);
		FR_ST(a.dL_dmean3D + 3 * orow + 2, g_mean.z);
	}
	// ---- 3D covariance: Sigma = A^T A, A = diag(s) B(q), B the matrix forward.cu:127-137 builds from the quaternion
	// (backward.cu:278-341). With Gs the symmetric gradient matrix (off-diagonals halved): dL/dA = 2 A Gs,
	// dL/ds_i = B_i . (dL/dA)_i, dL/dB_i = s_i (dL/dA)_i (rows), then through the quadratic entries of B.
	if (a.cov3D_precomp == nullptr && a.scales != nullptr)
	{
		const float4 q = stash ? make_float4(st1.z, st1.w, st2.x, st2.y) : ((const float4 *)a.rotations)[idx];
		const float r = q.x, x = q.y, y = q.z, z = q.w;
		const float s[3] = { a.scale_modifier * (stash ? st0.w : a.scales[3 * idx]), a.scale_modifier * (stash ? st1.x : a.scales[3 * idx + 1]),
			a.scale_modifier * (stash ? st1.y : a.scales[3 * idx + 2]) };
		const V3 B[3] = { { 1.f - 2.f * (y * y + z * z), 2.f * (x * y + r * z), 2.f * (x * z - r * y) },
			{ 2.f * (x * y - r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + r * x) },
			{ 2.f * (x * z + r * y), 2.f * (y * z - r * x), 1.f - 2.f * (x * x + y * y) } };
		const float Gs[6] = { gSigma[0], 0.5f * gSigma[1], 0.5f * gSigma[2], gSigma[3], 0.5f * gSigma[4], gSigma[5] };
		float gs[3];
		V3 D[3]; // dL/dB rows
#pragma unroll
		for (int i = 0; i < 3; i++)
		{
			const V3 e = symv(Gs, B[i]);            // (B Gs) row i; dL/dA row i = 2 s_i e
			gs[i] = 2.f * s[i] * dot3(B[i], e);
			const float f = 2.f * s[i] * s[i];
			D[i] = { f * e.x, f * e.y, f * e.z };
		}
		// B's entries are quadratic in (r, x, y, z): collect the antisymmetric and symmetric pairs of dL/dB
		const float a01 = D[0].y - D[1].x, a20 = D[2].x - D[0].z, a12 = D[1].z - D[2].y;
		const float p01 = D[0].y + D[1].x, p02 = D[0].z + D[2].x, p12 = D[1].z + D[2].y;
		float4 gq;
		gq.x = 2.f * (z * a01 + y * a20 + x * a12);
		gq.y = 2.f * (y * p01 + z * p02 + r * a12) - 4.f * x * (D[1].y + D[2].z);
		gq.z = 2.f * (x * p01 + r * a20 + z * p12) - 4.f * y * (D[0].x + D[2].z);
		gq.w = 2.f * (r * a01 + x * p02 + y * p12) - 4.f * z * (D[0].x + D[1].y);
		if (a.raw && stash)
		{
			// d exp(raw) = scale; d (v / |v|) = (g - u (u . g)) / |v| with u the unit quaternion of the forward pass
			gs[0] *= st0.w; gs[1] *= st1.x; gs[2] *= st1.y;
			const float inv = a.acc[4 * (size_t)slot + 3].x;
			if (inv > 0.f)
			{
				const float along = q.x * gq.x + q.y * gq.y + q.z * gq.z + q.w * gq.w;
				gq = make_float4((gq.x - q.x * along) * inv, (gq.y - q.y * along) * inv, (gq.z - q.z * along) * inv, (gq.w - q.w * along) * inv);
			}
			else gq = make_float4(gq.x * -inv, gq.y * -inv, gq.z * -inv, gq.w * -inv); // clamped denominator: x / 1e-12
		}
		if (sv) { sv[7] = gs[0]; sv[8] = gs[1]; sv[9] = gs[2]; sv[10] = gq.x; sv[11] = gq.y; sv[12] = gq.z; sv[13] = gq.w; }
		else
		{
			FR_ST(a.dL_dscale + 3 * orow, gs[0]); FR_ST(a.dL_dscale + 3 * orow + 1, gs[1]); FR_ST(a.dL_dscale + 3 * orow + 2, gs[2]);
			st_f4((float4 *)a.dL_drot + orow, gq);
		}
	}
}

// Zero-fill of the gradient tensors: every workgroup takes a contiguous share of every tensor (16-byte non-temporal stores, see FR_ST).
#define FR_FILL_MAX 12
#ifndef FR_FILL_BLOCKS
#define FR_FILL_BLOCKS 4096
#endif
struct FillArgs { int n; float *p[FR_FILL_MAX]; size_t words[FR_FILL_MAX]; };
__global__ void __launch_bounds__(256) k_fill_zero(const FillArgs a)
{
	for (int t = 0; t < a.n; t++)
	{
		const size_t quads = a.words[t] / 4, tail = a.words[t] - 4 * quads;
		nt_f4v *q = (nt_f4v *)a.p[t];
		for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store((nt_f4v){ 0.f, 0.f, 0.f, 0.f }, q + i);
		if (blockIdx.x == 0 && threadIdx.x < tail) a.p[t][4 * quads + threadIdx.x] = 0.0f;
	}
}

// The narrow tensors' share of the fill (SmallSet): the 32-row groups WITHOUT a visible Gaussian. One wave per 64 rows (two groups):
// their radii as one coalesced load, a ballot, and the zeros of an empty group as 16-byte (both groups) or 4-byte stores.
__global__ void __launch_bounds__(256) k_fill_groups(const int P, const int *radii, const SmallSet ss)
{
	const int lane = threadIdx.x & 63;
	const int nblk = (P + 63) / 64;
	for (int b = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)); b < nblk; b += (int)gridDim.x * 4)
	{
		const int row = 64 * b + lane;
		const unsigned long long m = __ballot(row < P && radii[row] > 0);
		const bool e0 = (uint32_t)m == 0u, e1 = (uint32_t)(m >> 32) == 0u;
		if (!e0 && !e1) continue;
		for (int t = 0; t < ss.n; t++)
		{
			const int w = ss.w[t];
			float *base = ss.p[t] + (size_t)64 * b * w; // (64 rows of w floats: a multiple of 16 bytes from the tensor's start)
			if (e0 && e1 && 64 * b + 64 <= P)
			{
				for (int q = lane; q < 16 * w; q += 64) __builtin_nontemporal_store((nt_f4v){ 0.f, 0.f, 0.f, 0.f }, (nt_f4v *)base + q);
				continue;
			}
			for (int o = lane; o < 64 * w; o += 64)
			{
				const int r = w == 1 ? o : (w == 3 ? o / 3 : o >> 2);
				if ((r < 32 ? e0 : e1) && 64 * b + r < P) __builtin_nontemporal_store(0.0f, base + o);
			}
		}
	}
}

// One narrow tensor's rows lo .. lo + n - 1 (the wave's visible rows and the zeros between / around them), put together as an image
// of the range in LDS and stored 16 bytes a lane: the image starts `pad` words into img so that LDS word and global word are aligned
// alike (the tensors are 16-byte aligned); the first and the last quad of the range may be partial and go word by word.
// vals: the lanes' staging rows; mine: this lane holds a visible row of the range (row `idx`).
template <int W>
__device__ __forceinline__ void expand_rows(float *dst, const int off, const int lo, const int n, float *img, const float *vals,
	const bool mine, const int idx, const int lane)
{
	const int first_word = lo * W, pad = first_word & 3, total = n * W, nq = (pad + total + 3) >> 2;
	float4 *img4 = (float4 *)img;
	for (int q = lane; q < nq; q += 64) img4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
	FR_WAVE_LDS_SYNC();
	if (mine)
	{
#pragma unroll
		for (int c = 0; c < W; c++) img[pad + (idx - lo) * W + c] = vals[lane * FR_SMALL_ROW + off + c];
	}
	FR_WAVE_LDS_SYNC();
	float *base = dst + ((size_t)first_word - pad); // 16-byte aligned
	for (int q = lane; q < nq; q += 64)
	{
		const float4 v = img4[q];
		const int o = 4 * q - pad; // the range's word index of v.x
		if (o >= 0 && o + 3 < total) __builtin_nontemporal_store((nt_f4v){ v.x, v.y, v.z, v.w }, (nt_f4v *)(base + 4 * q));
		else
		{
			if (o >= 0 && o < total) __builtin_nontemporal_store(v.x, base + 4 * q);
			if (o + 1 >= 0 && o + 1 < total) __builtin_nontemporal_store(v.y, base + 4 * q + 1);
			if (o + 2 >= 0 && o + 2 < total) __builtin_nontemporal_store(v.z, base + 4 * q + 2);
			if (o + 3 >= 0 && o + 3 < total) __builtin_nontemporal_store(v.w, base + 4 * q + 3);
		}
	}
	FR_WAVE_LDS_SYNC(); // the image is the next tensor's
}

// Grid-stride over the forward pass's visible list: dense waves instead of one thread per Gaussian with ~70 % of the
// lanes returning immediately. Entries culled after projection (radii reset to 0) are skipped. The rows of all other
// Gaussians are zero: launch_backward clears the output tensors on a helper stream WHILE k_render_bwd runs (that kernel
// is bound by arithmetic and atomics, the fill by HBM writes), so the caller need not zero-fill them -- the wide tensor (rest
// coefficients) in full, of the narrow ones (SmallSet) the 32-row groups without a visible Gaussian; the zeros of the other groups
// leave this kernel together with the rows. (Tried: one
// kernel that walks all Gaussians in index order, clears every chunk's rows with coalesced stores and works the
// chunk's visible ones off from an LDS list -- every row written once, no fill at all: 722 us against 539 + fill; at
// 158 registers the kernel does not have the occupancy to stream 1.5 GB of zeros.)
__global__ void __launch_bounds__(256) k_preprocess_bwd(const BwdPreArgs a)
{
	// The SH rows of a wave's 64 Gaussians move TOGETHER: fifteen lanes read / write one row's 180 bytes of rest coefficients as
	// 12-byte pieces (four rows per instruction) and the rows wait in LDS, where every lane works on its own. One lane per row
	// with 16-byte accesses -- 64 different rows per instruction -- spent 100 us on these reads and 175 us on the writes of a
	// 0.5 ms kernel (both measured by leaving them out).
	constexpr int ROWF = 45;              // rest floats of a degree-3 row
	__shared__ __attribute__((aligned(16))) float s_rows[4][64 * ROWF];
	__shared__ int s_idx[4][64];          // per lane of the wave: its Gaussian, or -1
	__shared__ long long s_orow[4][64];   // ... and the row of the gradient tensors it writes, or -1
	// the narrow tensors written in whole lines (SmallSet): the lanes' staging rows (the image of a range is put together in the wave's
	// s_rows, free again by then: 32 * FR_SMALL_GROUPS rows of at most 4 floats + 3 of padding fit its 64 * 45)
	__shared__ float s_small[4][64 * FR_SMALL_ROW];
	static_assert(32 * FR_SMALL_GROUPS * 4 + 8 <= 64 * ROWF, "the image of a range lives in the wave's SH rows");
	const bool expand = a.small.n != 0;
	const int V_all = (int)*a.vis_count;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const bool have_sh = a.colors_precomp == nullptr && a.shs != nullptr;
	const bool split = a.shs_rest != nullptr;
	const int nrest = split ? (a.M - 1) * 3 : a.M * 3 - 3;
	const bool coop = have_sh && nrest >= ROWF;   // (fewer coefficients per row: the per-lane path of preprocess_bwd_one)
	const int nuse = 3 * ((a.D + 1) * (a.D + 1)) - 3;
	const size_t src_stride = split ? (size_t)(a.M - 1) * 3 : (size_t)a.M * 3, src_off = split ? 0 : 3;
	const float *src = split ? a.shs_rest : a.shs;
	float *dst = split ? a.dL_dsh_rest : a.dL_dsh;
	const int r_in = lane / 15, part = lane - 15 * r_in; // this lane's row of a group of four and its piece of that row
	// (a range of the list: entries [i_lo, V) with V = its end; the whole list otherwise)
	const int i_lo = a.range_k >= 0 ? (int)a.range_bounds[a.range_k] : 0;
	const int V = a.range_k >= 0 ? (int)a.range_bounds[a.range_k + 1] : V_all;
	for (int base = i_lo + (blockIdx.x * blockDim.x + wv * 64); base < V; base += gridDim.x * blockDim.x)
	{
		const int i = base + lane;
		const int idx = i < V ? (int)a.vis_list[i] : 0;
		// (radii[idx] > 0, read from the item's dense word instead of a 64-byte line per Gaussian: k_bin clears the radius of exactly
		// the items it marks FR_ITEM_NONE)
		const bool alive = i < V && a.lrange[i] != FR_ITEM_NONE;
		if (coop)
		{
			s_idx[wv][lane] = alive ? idx : -1;
			s_orow[wv][lane] = alive ? (a.row_sparse ? (long long)i : (long long)idx) : -1;
			FR_WAVE_LDS_SYNC();
			float t[16][3];
#pragma unroll
			for (int it = 0; it < 16; it++)
			{
				const int r = 4 * it + r_in;
				const int g = lane < 60 ? s_idx[wv][r] : -1;
				t[it][0] = t[it][1] = t[it][2] = 0.0f;
				if (g >= 0 && 3 * part < nuse)
				{
					const float *q = src + (size_t)g * src_stride + src_off + 3 * part;
					t[it][0] = q[0]; t[it][1] = q[1]; t[it][2] = q[2];
				}
			}
#pragma unroll
			for (int it = 0; it < 16; it++)
				if (lane < 60)
				{
					float *w = &s_rows[wv][(4 * it + r_in) * ROWF + 3 * part];
					w[0] = t[it][0]; w[1] = t[it][1]; w[2] = t[it][2];
				}
			FR_WAVE_LDS_SYNC();
		}
		if (alive) preprocess_bwd_one(a, idx, i, a.cov3D_precomp ? nullptr : (const float4 *)a.cov3D_ws + 4 * (size_t)i, coop ? &s_rows[wv][lane * ROWF] : nullptr,
			expand ? &s_small[wv][lane * FR_SMALL_ROW] : nullptr);
		else if (a.row_sparse && i < V)
		{
			// a candidate that landed in no tile: its compact row is all zeros (every row of the compact tensors is written)
			const size_t r = (size_t)i;
			for (int k = 0; k < 3; k++) { a.dL_dmean2D[3 * r + k] = 0.f; a.dL_dmean3D[3 * r + k] = 0.f; if (a.dL_dscale) a.dL_dscale[3 * r + k] = 0.f; if (a.dL_dcolor) a.dL_dcolor[3 * r + k] = 0.f; }
			a.dL_dopacity[r] = 0.f;
			if (a.dL_drot) ((float4 *)a.dL_drot)[r] = make_float4(0.f, 0.f, 0.f, 0.f);
			if (a.dL_dconic) ((float4 *)a.dL_dconic)[r] = make_float4(0.f, 0.f, 0.f, 0.f);
			if (a.dL_dcov3D) for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * r + k] = 0.f;
			if (a.dL_dsh) for (int k = 0; k < 3 * a.M0; k++) a.dL_dsh[3 * (size_t)a.M0 * r + k] = 0.f;
			if (a.dL_dsh_rest) for (int k = 0; k < 3 * (a.M - 1); k++) a.dL_dsh_rest[3 * (size_t)(a.M - 1) * r + k] = 0.f;
		}
		if (coop)
		{
			FR_WAVE_LDS_SYNC();
			// the gradient rows leave the way the coefficients came (dense tensors: the active degree's part, the rest is the fill's;
			// compact rows: all of it)
			const int nstore = a.row_sparse ? ROWF : nuse;
#pragma unroll
			for (int it = 0; it < 16; it++)
			{
				const int r = 4 * it + r_in;
				const long long o = lane < 60 ? s_orow[wv][r] : -1;
				if (o >= 0 && 3 * part < nstore)
				{
					const float *w = &s_rows[wv][r * ROWF + 3 * part];
					float *q = dst + (size_t)o * src_stride + src_off + 3 * part;
					FR_ST(q, w[0]); FR_ST(q + 1, w[1]); FR_ST(q + 2, w[2]);
				}
			}
			FR_WAVE_LDS_SYNC(); // the rows and tables are rewritten by the next round
		}
		if (expand)
		{
			// the wave's visible rows (ascending) in turns of at most FR_SMALL_GROUPS groups -- one turn unless the rows are far apart
			FR_WAVE_LDS_SYNC();
			for (unsigned long long am = __ballot(alive); am != 0ull;)
			{
				const int l0 = __ffsll((long long)am) - 1;
				const int first = __builtin_amdgcn_readlane(idx, l0), g0 = first >> 5;
				const unsigned long long sm = __ballot(alive && (idx >> 5) - g0 < FR_SMALL_GROUPS) & am;
				const int l1 = 63 - __clzll((long long)sm);
				const int last = __builtin_amdgcn_readlane(idx, l1), g1 = last >> 5;
				// the visibility bits (radii > 0) of the first and of the last group: one load, one ballot
				const int brow = lane < 32 ? 32 * g0 + lane : 32 * g1 + (lane - 32);
				const unsigned long long bm = __ballot(brow < a.P && a.radii[brow] > 0);
				// the range: back to the row behind the previous visible one of the first group, on to the end of the last group
				// unless another visible row follows in it (that row's owner writes the zeros in front of it)
				const uint32_t below = (uint32_t)bm & ((1u << (first & 31)) - 1u);
				const int lo = below != 0u ? 32 * g0 + (31 - __clz((int)below)) + 1 : 32 * g0;
				const uint32_t above = (last & 31) == 31 ? 0u : (uint32_t)(bm >> 32) >> ((last & 31) + 1);
				const int hi = above != 0u ? last : min(32 * g1 + 31, a.P - 1);
				const bool mine = ((sm >> lane) & 1ull) != 0ull;
				for (int t = 0; t < a.small.n; t++)
				{
					const int w = a.small.w[t];
					if (w == 1) expand_rows<1>(a.small.p[t], a.small.off[t], lo, hi - lo + 1, s_rows[wv], s_small[wv], mine, idx, lane);
					else if (w == 3) expand_rows<3>(a.small.p[t], a.small.off[t], lo, hi - lo + 1, s_rows[wv], s_small[wv], mine, idx, lane);
					else expand_rows<4>(a.small.p[t], a.small.off[t], lo, hi - lo + 1, s_rows[wv], s_small[wv], mine, idx, lane);
				}
				am &= ~sm;
			}
		}
	}
}

static int launch_render_bwd(const fr_backward_args *a, const GeomWS &geom, const ImageWS &img, const BinWS &bin, int gx, int T, hipStream_t stream)
{
	BwdRenderArgs r;
	r.W = a->W; r.H = a->H; r.gx = gx; r.ranges = img.ranges; r.render_items = img.render_items; r.n_items = 2u * (uint32_t)T;
	r.point_list = bin.point_list; r.rec = geom.rec;
	r.bg = a->background; r.final_T = img.final_T; r.n_contrib = img.n_contrib; r.dL_dpix = a->dL_dpix;
	r.acc = (float *)geom.acc;
	r.pairs = a->blend_pairs;
	if (r.pairs != nullptr)
	{
		const hipError_t e = hipMemsetAsync(r.pairs, 0, sizeof(uint32_t) * (size_t)T, stream);
		if (e != hipSuccess) { set_error("hipMemsetAsync(blend_pairs): %s", hipGetErrorString(e)); return FR_ERR_HIP; }
	}
	if (a->variant == FR_VARIANT_ORIGINAL)
		hipLaunchKernelGGL((k_render_bwd<false>), dim3(r.n_items), dim3(64), 0, stream, r);
	else
		hipLaunchKernelGGL((k_render_bwd<true>), dim3(r.n_items), dim3(64), 0, stream, r);
	return check_launch("render_bwd", stream, a->debug);
}

// the dense gradient tensors of a backward call cleared by ONE kernel on stream `fs` (seven fill commands in a row ran at 2.7 TB/s
// beside k_render_bwd, the small ones at 1 TB/s, and outlasted it by 50 us); tensors are 16-byte aligned and their sizes multiples of 4
// the narrow tensors of a dense backward call (SmallSet): written in whole lines by k_preprocess_bwd + k_fill_groups instead of fill + rows
static SmallSet small_set(const fr_backward_args *a)
{
	SmallSet ss; ss.n = 0;
#ifdef FR_NO_EXPAND // (developer switch for A / B timing: every row stored on its own, everything cleared by k_fill_zero)
	return ss;
#endif
	if (a->row_sparse || a->outputs_zeroed || a->radii == nullptr) return ss;
	// (all of them or none: k_preprocess_bwd stages every narrow row once the set is not empty. The 16-byte stores of the ranges want
	// 16-byte aligned tensors -- what any allocator hands out; a host with odd pointers gets the rows stored one by one)
	bool usable = true;
	auto add = [&](float *p, int w, int off) { if (!p || ((uintptr_t)p & 15) != 0) usable = false; else { ss.p[ss.n] = p; ss.w[ss.n] = w; ss.off[ss.n] = off; ss.n++; } };
	add(a->dL_dmean2D, 3, 0); add(a->dL_dopacity, 1, 3); add(a->dL_dmean3D, 3, 4);
	if (a->cov3D_precomp == nullptr && a->scales != nullptr) { add(a->dL_dscale, 3, 7); add(a->dL_drot, 4, 10); }
	if (a->colors_precomp == nullptr && a->shs != nullptr && a->shs_rest != nullptr) add(a->dL_dsh, 3, 14);
	if (!usable) ss.n = 0;
	return ss;
}

int launch_gradient_fill(const fr_backward_args *a, hipStream_t fs, bool events, bool whole)
{
	if (a->row_sparse) return FR_OK; // (compact rows: k_preprocess_bwd writes every row it is handed, nothing to clear)
	SmallSet ss = small_set(a);
	if (whole) ss.n = 0;
	auto narrow = [&](const void *p) { for (int t = 0; t < ss.n; t++) if (ss.p[t] == p) return true; return false; };
	const bool have_sh = a->colors_precomp == nullptr && a->shs != nullptr;
	const size_t P = (size_t)a->P;
	const size_t m0 = have_sh ? (a->shs_rest ? 1 : (size_t)a->M) : 0;
	struct { void *p; size_t bytes; } fills[] = {
		{ a->dL_dmean3D, 12 * P }, { a->dL_dmean2D, 12 * P }, { a->dL_dopacity, 4 * P }, { a->dL_dscale, 12 * P }, { a->dL_drot, 16 * P },
		{ a->dL_dsh, 12 * m0 * P }, { a->dL_dsh_rest, have_sh && a->shs_rest ? 12 * ((size_t)a->M - 1) * P : 0 },
		{ a->dL_dcolor, 12 * P }, { a->dL_dconic, 16 * P }, { a->dL_dcov3D, 24 * P } };
	FillArgs fa; fa.n = 0;
	for (auto &f : fills)
		if (f.p && f.bytes && !narrow(f.p))
		{
			if (((uintptr_t)f.p & 15) != 0 || fa.n == FR_FILL_MAX)
			{
				const hipError_t e = hipMemsetAsync(f.p, 0, f.bytes, fs);
				if (e != hipSuccess) { set_error("hipMemsetAsync(gradient): %s", hipGetErrorString(e)); return FR_ERR_HIP; }
				continue;
			}
			fa.p[fa.n] = (float *)f.p; fa.words[fa.n] = f.bytes / 4; fa.n++;
		}
	if (fa.n || ss.n)
	{
		// (optional events 3 / 4: around the fill kernels, on the stream they run on)
		if (events && a->stage_events && a->stage_events[3]) (void)hipEventRecord((hipEvent_t)a->stage_events[3], fs);
		int rcf = FR_OK;
		if (fa.n)
		{
			hipLaunchKernelGGL(k_fill_zero, dim3(FR_FILL_BLOCKS), dim3(256), 0, fs, fa);
			rcf = check_launch("fill_zero", fs, a->debug);
		}
		if (!rcf && ss.n)
		{
			const int nblk = (a->P + 63) / 64;
			hipLaunchKernelGGL(k_fill_groups, dim3((unsigned)min((nblk + 3) / 4, 8192)), dim3(256), 0, fs, a->P, a->radii, ss);
			rcf = check_launch("fill_groups", fs, a->debug);
		}
		if (events && a->stage_events && a->stage_events[4]) (void)hipEventRecord((hipEvent_t)a->stage_events[4], fs);
		if (rcf) return rcf;
	}
	return FR_OK;
}

int launch_backward(const fr_backward_args *a)
{
	hipStream_t stream = (hipStream_t)a->stream;
	const int gx = (a->W + FR_TILE - 1) / FR_TILE, gy = (a->H + FR_TILE - 1) / FR_TILE, T = gx * gy;
	GeomWS geom = carve_geom(a->variant, (size_t)a->P, (char *)a->geometry);
	ImageWS img = carve_image(a->variant, a->W, a->H, (char *)a->image);
	BinWS bin = carve_bin(a->R, (char *)a->binning);
	auto mark = [&](int i) { if (a->stage_events && a->stage_events[i]) (void)hipEventRecord((hipEvent_t)a->stage_events[i], stream); };
	mark(0);
	const int K = (a->num_ranges > 1 && a->range_done != nullptr && !a->row_sparse && a->P >= 64 * a->num_ranges)
		? (a->num_ranges < FR_MAX_RANGES ? a->num_ranges : FR_MAX_RANGES) : 1;
	// The gradient tensors are written in full by this call: their rows are cleared here, on the helper stream, while
	// k_render_bwd runs on the caller's (1.5 GB of fills at 6 M Gaussians, ~0.3 ms that the reference -- and round 1 --
	// spend before the backward pass starts).
	{
		AuxStream *ax = (a->R > 0 && !a->debug && !a->outputs_zeroed && !a->row_sparse) ? aux_stream(stream) : nullptr;
		hipStream_t fs = stream;
		if (ax)
		{
			if (hipEventRecord(ax->fork, stream) != hipSuccess || hipStreamWaitEvent(ax->s, ax->fork, 0) != hipSuccess) { (void)hipGetLastError(); ax = nullptr; }
			else fs = ax->s;
		}
		if (!a->outputs_zeroed)
		{
			const int rcf = launch_gradient_fill(a, fs, true, false);
			if (rcf)
			{
				// (whatever was enqueued on the helper stream is joined before the caller gets its tensors back)
				if (ax) { (void)hipEventRecord(ax->join, ax->s); (void)hipStreamWaitEvent(stream, ax->join, 0); }
				return rcf;
			}
		}
		if (K > 1)
		{
			// (where every range of rows starts in the visible list: beside the fills, off the critical path)
			hipLaunchKernelGGL(k_range_bounds, dim3(1), dim3(64), 0, fs, geom.vis_list, geom.slab_ctr + 1, a->P, K, geom.slab_ctr + 8);
			const int rcb = check_launch("range_bounds", fs, a->debug);
			if (rcb) { if (ax) { (void)hipEventRecord(ax->join, ax->s); (void)hipStreamWaitEvent(stream, ax->join, 0); } return rcb; }
		}
		if (ax) (void)hipEventRecord(ax->join, ax->s); // waited for below, after k_render_bwd has been launched
		if (a->R > 0)
		{
			const int rc0 = launch_render_bwd(a, geom, img, bin, gx, T, stream);
			if (rc0)
			{
				// the fill is still in flight on the helper stream: the caller may free the gradient tensors once we return
				if (ax) (void)hipStreamWaitEvent(stream, ax->join, 0);
				return rc0;
			}
		}
		if (ax) (void)hipStreamWaitEvent(stream, ax->join, 0);
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
	p.radii = a->radii; p.rec = geom.rec; p.cov3D_ws = geom.cov3D; p.acc = geom.acc;
	p.dL_dmean2D = a->dL_dmean2D; p.dL_dconic = a->dL_dconic; p.dL_dcolor = a->dL_dcolor; p.dL_dopacity = a->dL_dopacity;
	p.dL_dmean3D = a->dL_dmean3D; p.dL_dcov3D = a->dL_dcov3D; p.dL_dsh = a->dL_dsh; p.dL_dsh_rest = a->dL_dsh_rest; p.dL_dscale = a->dL_dscale; p.dL_drot = a->dL_drot;
	p.vis_list = geom.vis_list; p.vis_count = geom.slab_ctr + 1; p.lrange = geom.lrange; p.raw = a->raw_activations;
	p.small = small_set(a);
	p.row_sparse = a->row_sparse; p.M0 = (a->colors_precomp == nullptr && a->shs != nullptr) ? (a->shs_rest ? 1 : a->M) : 0;
	p.range_bounds = geom.slab_ctr + 8; p.range_k = -1;
	const int pblocks = (a->P + 255) / 256;
	int rc2 = FR_OK;
	if (K > 1)
	{
		// one launch per range of rows; behind each the host is told that those rows of every gradient tensor are complete once the
		// stream gets here (a multi-GPU host starts summing them over its ranks while the later ranges are computed: fovraster.h)
		for (int k = 0; k < K && !rc2; k++)
		{
			p.range_k = k;
			const int rows = range_row_lo(a->P, k + 1, K) - range_row_lo(a->P, k, K);
			const int blocks = (rows + 255) / 256;
			hipLaunchKernelGGL(k_preprocess_bwd, dim3(blocks < 2048 ? (blocks > 0 ? blocks : 1) : 2048), dim3(256), 0, stream, p);
			rc2 = check_launch("preprocess_bwd", stream, a->debug);
			if (!rc2) a->range_done(a->range_user, k, range_row_lo(a->P, k, K), range_row_lo(a->P, k + 1, K));
		}
	}
	else
	{
		hipLaunchKernelGGL(k_preprocess_bwd, dim3(pblocks < 2048 ? pblocks : 2048), dim3(256), 0, stream, p);
		rc2 = check_launch("preprocess_bwd", stream, a->debug);
	}
	mark(2);
	return rc2;
}

} // namespace fr
