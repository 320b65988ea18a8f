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
// (rows ry, ry+16/PPL, ...); PPL = 2, i.e. two waves per tile, by default. A wave works through its tile's
// list serially and the frame ends with the slowest tile, so the per-entry instruction chain is what
// counts: the pixels of a lane are blended as packed pairs (v_pk_mul/v_pk_fma_f32), fully predicated, with
// the "finished" flag folded into the sign of the transmittance; the only branches are wave-uniform
// (all pixels finished / splat misses every live pixel). Instance records are gathered as 48-byte AoS
// (3 x b128) per Gaussian by the whole group and the next batch is prefetched into registers.
// The foveated renderer handles single-level and two-level tiles in one launch (the reference
// launches two full grids that early-return on each other's tiles).
#include "common.h"
#include <cstdlib>

#ifndef FR_RENDER_GROUP
#define FR_RENDER_GROUP 2       // RF: entries whose transmittance-independent part is evaluated together
#endif
#ifndef FR_RENDER_GROUP_PLAIN
#define FR_RENDER_GROUP_PLAIN 2 // plain variants (throughput-bound frames: no gain, no loss)
#endif

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

// q = -power = 0.5 (A dx^2 + C dy^2) + B dx dy for two rows at once. Negation commutes with rounding, so -q is bit for bit
// the power of power2(); what the blend needs of it is exp(-q) and the support test 0 <= q <= 4.5 (power > 0 and
// power < -4.5 are skipped, RS forward.cu:376-380), and on the BIT PATTERN that test is ONE unsigned comparison: a
// negative q (power > 0) has its sign bit set, i.e. is a huge unsigned number, and non-negative floats order like their
// bits. q is +0 where the reference's power is -0 or +0 (0.5 s is +0, and +0 + -0 = +0), which it does not skip either.
// (A NaN has all exponent bits set and fails the test; the reference lets a NaN power through to a NaN alpha: garbage in
// both cases, for a conic that is no conic.)
__device__ __forceinline__ v2f qform2(v2f dy, float C, float adx2, float bdx)
{
	const v2f cdy = C * dy;
	const v2f s = __builtin_elementwise_fma(cdy, dy, (v2f){ adx2, adx2 });
	return __builtin_elementwise_fma((v2f){ 0.5f, 0.5f }, s, bdx * dy);
}
__device__ __forceinline__ bool in_support(float q) { return __float_as_uint(q) <= 0x40900000u; } // 0 <= q <= 4.5
__device__ __forceinline__ v2f exp_neg_pair(v2f q)
{
	const v2f t = q * -1.4426950408889634f;
	return (v2f){ __builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y) };
}

// Two pixels (rows) of a lane: transmittance (negated once the pixel is finished) and colour sums.
struct Px2 { v2f T, C0, C1, C2; };
__device__ __forceinline__ v2f exp2_pair(v2f p)
{
	const v2f q = p * 1.4426950408889634f;
	return (v2f){ __builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y) };
}
// Front-to-back update of two pixels for one splat, fully predicated (forward.cu:349-372): skipped if the pixel
// is outside the splat's support / finished / alpha < 1/255; a pixel whose transmittance would drop below 1e-4
// is finished instead (sign flip) and keeps its T.
__device__ __forceinline__ void blend2(Px2 &s, bool in_x, bool in_y, v2f e, float4 c, v2f &w_out, bool &acc_x, bool &acc_y)
{
	v2f alpha = c.w * e;
	alpha.x = fminf(0.99f, alpha.x); alpha.y = fminf(0.99f, alpha.y);
	const v2f tt = s.T * (1.0f - alpha);
	v2f w = alpha * s.T;
	const bool live_x = in_x && s.T.x > 0.0f && !(alpha.x < 1.0f / 255.0f);
	const bool live_y = in_y && s.T.y > 0.0f && !(alpha.y < 1.0f / 255.0f);
	const bool sat_x = tt.x < 0.0001f, sat_y = tt.y < 0.0001f;
	w.x = (live_x && !sat_x) ? w.x : 0.0f;
	w.y = (live_y && !sat_y) ? w.y : 0.0f;
	s.C0 = __builtin_elementwise_fma((v2f){ c.x, c.x }, w, s.C0);
	s.C1 = __builtin_elementwise_fma((v2f){ c.y, c.y }, w, s.C1);
	s.C2 = __builtin_elementwise_fma((v2f){ c.z, c.z }, w, s.C2);
	s.T.x = live_x ? (sat_x ? -s.T.x : tt.x) : s.T.x;
	s.T.y = live_y ? (sat_y ? -s.T.y : tt.y) : s.T.y;
	w_out = w; acc_x = live_x && !sat_x; acc_y = live_y && !sat_y;
}
__device__ __forceinline__ void blend2(Px2 &s, bool in_x, bool in_y, v2f e, float4 c)
{
	v2f w; bool ax, ay;
	blend2(s, in_x, in_y, e, c, w, ax, ay);
}

// The same update with the bookkeeping folded differently (k_render_fov): a finished pixel carries T = 0 -- every product with it is
// an exact zero and T (1 - alpha) = 0 < 1e-4 keeps it finished, so no "still blending" test is needed -- and the transmittance the
// pixel ends with (the reference's T: the saturating Gaussian is not accumulated, forward.cu:367-372) is kept beside it in Tk;
// `ok` = the two skip tests of the reference that do not depend on T (outside the support, alpha < 1/255) as ONE comparison made by
// the caller. Two compares + four selects per pixel instead of four + three: a v_cmp costs two plain VALU slots on gfx950.
struct Px2k { v2f T, Tk, C0, C1, C2; };
template <bool CLAMP>
__device__ __forceinline__ void blend2k(Px2k &s, bool ok_x, bool ok_y, v2f e, float4 c, v2f &w_out, bool &acc_xo, bool &acc_yo)
{
	v2f alpha = c.w * e;
	if (CLAMP) { alpha.x = fminf(0.99f, alpha.x); alpha.y = fminf(0.99f, alpha.y); }
	const v2f tt = s.T * (1.0f - alpha);
	v2f w = alpha * s.T;
	const bool sat_x = tt.x < 0.0001f, sat_y = tt.y < 0.0001f;
	const bool acc_x = ok_x && !sat_x, acc_y = ok_y && !sat_y;
	w.x = acc_x ? w.x : 0.0f;
	w.y = acc_y ? w.y : 0.0f;
	s.C0 = __builtin_elementwise_fma((v2f){ c.x, c.x }, w, s.C0);
	s.C1 = __builtin_elementwise_fma((v2f){ c.y, c.y }, w, s.C1);
	s.C2 = __builtin_elementwise_fma((v2f){ c.z, c.z }, w, s.C2);
	s.Tk.x = acc_x ? tt.x : s.Tk.x;
	s.Tk.y = acc_y ? tt.y : s.Tk.y;
	const float nx = sat_x ? 0.0f : tt.x, ny = sat_y ? 0.0f : tt.y;
	s.T.x = ok_x ? nx : s.T.x;
	s.T.y = ok_y ? ny : s.T.y;
	w_out = w; acc_xo = acc_x; acc_yo = acc_y;
}
template <bool CLAMP>
__device__ __forceinline__ void blend2k(Px2k &s, bool ok_x, bool ok_y, v2f e, float4 c)
{
	v2f w; bool ax, ay;
	blend2k<CLAMP>(s, ok_x, ok_y, e, c, w, ax, ay);
}

struct RenderArgs {
	int W, H, gx;
	const uint2 *ranges;
	const uint32_t *point_list; // per-tile sorted lists of ITEMS (positions in vis_list): rec / lvl are per item
	const uint32_t *vis_list;   // item -> Gaussian index (the training variants' statistics are per Gaussian)
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
	const uint32_t *render_items; // work items, longest list first (k_tile_scan): tile << 3 | band | level state << 1 | two-level << 2 ...
	const uint32_t *totals;       // ... ImageWS::totals: [5] = their number (the grid may be an upper bound), [0] = the frame's instances
	uint32_t capacity;            // instances the binning workspace holds: a frame with more is not blended (the host replays it)
	float cur_level;              // MMFR
	uint32_t *round_flags;        // RS / LWMC: one bit per (tile, 256-entry round): the round's counts have an owner
	uint32_t *consumed;           // optional diagnostic (fr_forward_args.list_consumed): [T], entries fetched per tile, or null
	uint32_t *pairs;              // optional diagnostic (fr_forward_args.blend_pairs): [T], (band, entry) pairs evaluated, or null
};

// list_consumed: a wave reports how far into its tile's list it staged entries for blending (one atomic per wave, only when asked for)
__device__ __forceinline__ void report_consumed(const RenderArgs &a, int tile, int used, int lane, uint32_t npairs = 0)
{
	if (a.consumed != nullptr && lane == 0 && used > 0) atomicMax(a.consumed + tile, (uint32_t)used);
	if (a.pairs != nullptr && lane == 0 && npairs > 0) atomicAdd(a.pairs + tile, npairs);
}

// ---------------- ORIGINAL / PCHECK_OBB_SUM / PCHECK_OBB / _MAX / _LWMC ----------------
// One single-wave workgroup per work item (a band of eight rows of a tile; render_items, longest list first), every
// wave on its own: own batches of 64 entries, own reach mask, no workgroup barrier. The records of the NEXT batch are
// prefetched into registers before the current batch is blended (the point_list -> record gather is two dependent
// global loads).
// RS / LWMC count, per Gaussian, the list entries a tile FETCHES: gaussians_count += 1 for every entry of every
// 256-entry round the tile starts, and a round starts unless all 256 threads were done when it would (RS
// forward.cu:349-361) -- i.e. iff at least one of the two bands is not done at the round's first entry. Round 1 kept
// both bands in one workgroup with a barrier-coupled vote for that (400 us against 180 us for the same lists without
// the statistics, 37 % of the wave-cycles parked). Here the bands stay independent and CLAIM rounds: a wave that is not
// done at a round boundary sets the round's bit in round_flags (one atomicOr per wave and round); whoever finds the bit
// clear owns the round and adds the counts of its (up to) 256 entries as it walks them -- to the end of the round even
// if its own band finishes in between. A round nobody claims was not started.
template <int VARIANT, int PPL>
__global__ void __launch_bounds__(64) k_render(const RenderArgs a)
{
	// (FR_VARIANT_SUM_NOSTATS, internal: pcheck_obb_sum for a caller that drops the statistics, fr_forward_args.no_stats)
	constexpr bool CUTOFF = VARIANT != FR_VARIANT_ORIGINAL;
	constexpr bool SUM = VARIANT == FR_VARIANT_PCHECK_OBB_SUM;   // contributions += alpha*T, count per fetched entry
	constexpr bool PMAX = VARIANT == FR_VARIANT_PCHECK_OBB_MAX;  // contributions = max alpha*T, count per in-support pixel
	constexpr bool LWMC = VARIANT == FR_VARIANT_PCHECK_OBB_LWMC; // per-pixel loss to its max-contribution Gaussian
	constexpr bool FETCHCNT = SUM || LWMC;                       // gaussians_count per fetched entry (256-rounds)
	constexpr bool NEEDID = SUM || PMAX || LWMC;
	constexpr bool AUX = VARIANT != FR_VARIANT_PCHECK_OBB; // final_T / n_contrib kept for backward
	static_assert(PPL == 2, "work items encode two bands per tile");
	constexpr int HP = PPL / 2;

	__shared__ float4 s0[64];
	__shared__ float4 s1[64];
	__shared__ float2 s2[64];   // b, tq: the threshold on q = -power below which a pixel takes part (see the staging)
	__shared__ int sid[NEEDID ? 64 : 1];
	__shared__ int stqa[PMAX ? 64 : 1]; // _max: the alpha test's threshold on q beside the support's (see the staging)

	const int st = threadIdx.x; // lane = staging slot
	if (blockIdx.x >= a.totals[5] || a.totals[0] > a.capacity || a.totals[5] > gridDim.x) return;
	const uint32_t item = a.render_items[blockIdx.x];
	const int tile = (int)(item >> 3), wv = (int)(item & 1u); // band of this wave
	const int tx = tile % a.gx, ty = tile / a.gx;
	const int tid = wv * 64 + st;                              // position among the tile's 128 threads (row mapping)
	const int lx = tid & 15;
	const int px = tx * FR_TILE + lx;
	const float pxf = (float)px;
	const uint2 range = a.ranges[tile];
	const int n = (int)(range.y - range.x);

	// pixel state as packed row pairs; a finished pixel carries T = 0 and the transmittance it ended with in Tk (see Px2k)
	Px2k S[HP];
	float pyf[PPL];
	uint32_t last[PPL];
	bool inside[PPL];
#pragma unroll
	for (int k = 0; k < PPL; k++)
	{
		const int py = ty * FR_TILE + tile_row<PPL>(tid, k);
		pyf[k] = (float)py;
		inside[k] = px < a.W && py < a.H;
		S[k >> 1].T[k & 1] = inside[k] ? 1.0f : 0.0f;
		S[k >> 1].Tk[k & 1] = 1.0f;
		last[k] = 0;
	}
#pragma unroll
	for (int h = 0; h < HP; h++) S[h].C0 = S[h].C1 = S[h].C2 = (v2f){ 0.f, 0.f };
	float best_w[LWMC ? PPL : 1];   // LWMC forward.cu:347-348: running max contribution per pixel ...
	int best_id[LWMC ? PPL : 1];    // ... and whose it is (defaults to Gaussian 0, as in the reference)
#pragma unroll
	for (int k = 0; k < (LWMC ? PPL : 1); k++) { best_w[k] = 0.0f; best_id[k] = 0; }

	// prefetch registers
	uint32_t pid = 0, pgid = 0; // item, and (statistics) its Gaussian index
	float4 p0 = make_float4(0, 0, 0, 0), p1 = p0;
	float p2 = 0.f;
	if (st < n)
	{
		pid = a.point_list[range.x + st];
		const float4 *r = a.rec + 3 * (size_t)pid;
		p0 = r[0]; p1 = r[1];
		// (statistics: the Gaussian's index is the last word of the item's own record -- the same cache line as the colour's third
		// channel -- where round 4 gathered it from vis_list, a line of its own per entry)
		if (NEEDID) { const float4 r2 = r[2]; p2 = r2.x; pgid = __float_as_uint(r2.w); } else p2 = r[2].x;
	}
	bool counting = false; // FETCHCNT: this wave owns the counts of the current 256-entry round
	int used = 0;          // list entries this wave staged for blending (list_consumed)
	uint32_t npairs = 0;   // ... and how many of them can reach its band (blend_pairs)
	for (int base = 0; base < n; base += 64)
	{
		float tmax0 = -1.0f;
#pragma unroll
		for (int h = 0; h < HP; h++) tmax0 = fmaxf(tmax0, fmaxf(S[h].T.x, S[h].T.y));
		const bool wave_done = !__any(tmax0 > 0.0f);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the previous batch has been read by all lanes
		__builtin_amdgcn_wave_barrier();
		if (FETCHCNT)
		{
			if ((base & 255) == 0)
			{
				counting = false;
				if (!wave_done)
				{
					// the flag of round r of this tile: distinct for all (tile, round) pairs because a tile's range starts at least
					// ceil(n / 256) - 1 flag positions after its predecessor's and the tile index adds one more
					const uint32_t f = (range.x >> 8) + (uint32_t)tile + ((uint32_t)base >> 8);
					uint32_t old = 0;
					if (st == 0) old = atomicOr(a.round_flags + (f >> 5), 1u << (f & 31u));
					counting = (((uint32_t)__builtin_amdgcn_readfirstlane((int)old) >> (f & 31u)) & 1u) == 0;
				}
			}
			if (wave_done && !counting) break;
		}
		else if (wave_done) break;
		const bool staged = base + st < n;
		if (FETCHCNT && counting && staged) atomicAdd(&a.gaussians_count[pgid], 1);
		if (FETCHCNT && wave_done)
		{
			// this band is finished but owns the round: only its remaining counts are due
			if (base + 64 + st < n) pgid = __float_as_uint(a.rec[3 * (size_t)a.point_list[range.x + base + 64 + st] + 2].w);
			continue;
		}
		used = min(n, base + 64);
		// The skip tests that do not depend on the pixel's transmittance as ONE threshold on q = -power (forward.cu:349-365, RS :376-380):
		// power > 0, (CUTOFF) power < -4.5, and alpha = o e^-q < 1/255 <=> q > ln(255 o): a pixel takes part iff 0 <= q <= tq with
		// tq = min(4.5, ln(255 o)), one unsigned comparison on q's bits. ln(255 o) to an ulp (logf): the decision then differs from the
		// reference's own fp32 evaluation of o * exp(power) < 1/255 only inside that expression's rounding, like the exp2-based test it
		// replaces. The _max flavour counts pixels between the support test and the alpha test: tq is the support's alone there, and the
		// alpha test is a second threshold of the same form (stqa; -1 = no pixel passes, compared as signed integers: q >= 0 inside the
		// support) -- the one k_render_bwd applies, so both passes blend the same pairs in this variant too.
		const float lq = logf(255.0f * p1.y);
		// (lq >= 0 is false for the NaN of a negative opacity: such an entry takes part nowhere, as alpha < 1/255 says in the reference)
		const float tq = PMAX ? 4.5f : (lq >= 0.0f ? (CUTOFF ? fminf(4.5f, lq) : lq) : 0.0f);
		if (staged)
		{
			s0[st] = p0; s1[st] = p1; s2[st] = make_float2(p2, tq);
			if (NEEDID) sid[st] = (int)pgid;
			if (PMAX) stqa[st] = __float_as_int(lq >= 0.0f ? fminf(4.5f, lq) : -1.0f);
		}
		unsigned long long reach_own;
		{
			// which entries can touch this band at all (see splat_reaches)? alpha < 1/255 (forward.cu:336) <=> power <
			// -ln(255 opacity); the _max flavour counts pixels BEFORE the alpha test, so only the support cutoff applies
			const float thr_a = -lq - 0.01f;
			const float thr = PMAX ? -4.5f : (CUTOFF ? fmaxf(-4.5f, thr_a) : thr_a);
			// (an opacity below 1/255 -- or a NaN -- passes the alpha test nowhere: forward.cu:363)
			reach_own = __ballot(staged && (PMAX || lq >= 0.0f) && band_reaches<PPL>(wv, tx, ty, p0.x, p0.y, p0.z, p0.w, p1.x, thr));
			npairs += (uint32_t)__popcll(reach_own);
			}
		if (base + 64 + st < n)
		{
			pid = a.point_list[range.x + base + 64 + st];
			const float4 *r = a.rec + 3 * (size_t)pid;
			p0 = r[0]; p1 = r[1];
			if (NEEDID) { const float4 r2 = r[2]; p2 = r2.x; pgid = __float_as_uint(r2.w); } else p2 = r[2].x;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // lanes read entries other lanes staged
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		v2f pyp[HP];
#pragma unroll
		for (int h = 0; h < HP; h++) pyp[h] = (v2f){ pyf[2 * h], pyf[2 * h + 1] };
		// Entries are taken FR_RENDER_GROUP_PLAIN at a time: the part that does not depend on the running transmittance
		// (record fetch, power, support test, exp) is evaluated for all of them before the first one is blended,
		// so their dependency chains overlap (see k_render_fov).
		struct Ent { v2f e[HP]; bool inx[HP], iny[HP], okx[HP], oky[HP]; float4 col; int j; };
		auto prepare = [&](const int j, const bool valid)
		{
			Ent t;
			t.j = j;
			const float4 g0 = s0[j];
			const float4 g1 = s1[j];
			const float2 g2 = s2[j];
			const uint32_t tqb = __float_as_uint(g2.y);
			const float dx = g0.x - pxf;
			const float adx2 = (g0.z * dx) * dx;     // A*dx*dx
			const float bdx = g0.w * dx;             // B*dx
#pragma unroll
			for (int h = 0; h < HP; h++)
			{
				// (forward.cu:349-351: power > 0 is skipped; RS :376-380 also power < -4.5 -- one comparison on q's bits, see qform2)
				const v2f q = qform2(g0.y - pyp[h], g1.x, adx2, bdx);
				t.inx[h] = valid && __float_as_uint(q.x) <= tqb; // 0 <= q <= tq
				t.iny[h] = valid && __float_as_uint(q.y) <= tqb;
				// (_max: inx / iny = inside the support, what the count takes; okx / oky = ... and alpha >= 1/255, what is blended)
				t.okx[h] = PMAX ? t.inx[h] && __float_as_int(q.x) <= stqa[j] : t.inx[h];
				t.oky[h] = PMAX ? t.iny[h] && __float_as_int(q.y) <= stqa[j] : t.iny[h];
				t.e[h] = exp_neg_pair(q);
			}
			t.col = make_float4(g1.z, g1.w, g2.x, g1.y); // r, g, b, opacity
			return t;
		};
		auto blend = [&](const Ent &t, float &contrib_sum, bool &any_contrib)
		{
			const int j = t.j;
			float contrib_max = 0.0f;
			contrib_sum = 0.0f; any_contrib = false;
			if (PMAX)
			{
				// ..._max forward.cu:381: +1 for every live pixel inside the splat's support (before the alpha test)
				int c = 0;
#pragma unroll
				for (int h = 0; h < HP; h++)
					c += __popcll(__ballot(t.inx[h] && S[h].T.x > 0.0f)) + __popcll(__ballot(t.iny[h] && S[h].T.y > 0.0f));
				if (st == 0 && c != 0) atomicAdd(&a.gaussians_count[sid[j]], c);
			}
#pragma unroll
			for (int h = 0; h < HP; h++)
			{
				v2f w; bool ax, ay;
				blend2k<true>(S[h], t.okx[h], t.oky[h], t.e[h], t.col, w, ax, ay);
				if (AUX)
				{
					last[2 * h] = ax ? (uint32_t)(base + j + 1) : last[2 * h];
					last[2 * h + 1] = ay ? (uint32_t)(base + j + 1) : last[2 * h + 1];
				}
				if (SUM) { contrib_sum += w.x; contrib_sum += w.y; any_contrib = any_contrib || ax || ay; }
				if (PMAX) { contrib_max = fmaxf(contrib_max, fmaxf(w.x, w.y)); any_contrib = any_contrib || ax || ay; }
				if (LWMC)
				{
					const bool bx = ax && (w.x > best_w[2 * h]), by = ay && (w.y > best_w[2 * h + 1]);
					best_w[2 * h] = bx ? w.x : best_w[2 * h]; best_id[2 * h] = bx ? sid[j] : best_id[2 * h];
					best_w[2 * h + 1] = by ? w.y : best_w[2 * h + 1]; best_id[2 * h + 1] = by ? sid[j] : best_id[2 * h + 1];
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
					if (st == 0) atomicMax((unsigned int *)&a.contributions[sid[j]], __float_as_uint(m));
				}
			}
		};
		for (unsigned long long rm = reach_own; rm; )
		{
			int jj[FR_RENDER_GROUP_PLAIN];
			bool vv[FR_RENDER_GROUP_PLAIN];
#pragma unroll
			for (int g = 0; g < FR_RENDER_GROUP_PLAIN; g++)
			{
				vv[g] = rm != 0;
				jj[g] = vv[g] ? __builtin_ctzll(rm) : jj[0];
				rm &= rm - 1; // stays 0 once empty
			}
			float tmax = -1.0f;
#pragma unroll
			for (int h = 0; h < HP; h++) tmax = fmaxf(tmax, fmaxf(S[h].T.x, S[h].T.y));
			if (!__any(tmax > 0.0f)) break; // wave saturated
			Ent t[FR_RENDER_GROUP_PLAIN];
			bool anyhit = false;
#pragma unroll
			for (int g = 0; g < FR_RENDER_GROUP_PLAIN; g++)
			{
				t[g] = prepare(jj[g], vv[g]);
#pragma unroll
				for (int h = 0; h < HP; h++) anyhit = anyhit || (t[g].inx[h] && S[h].T.x > 0.0f) || (t[g].iny[h] && S[h].T.y > 0.0f);
			}
			if (!__any(anyhit)) continue; // the splats miss every live pixel of this wave's rows
			float csum[FR_RENDER_GROUP_PLAIN];
			bool cany[FR_RENDER_GROUP_PLAIN];
#pragma unroll
			for (int g = 0; g < FR_RENDER_GROUP_PLAIN; g++) blend(t[g], csum[g], cany[g]);
			if (SUM)
			{
				// RS forward.cu:400: contributions[id] += alpha T per pixel. The group's sums are reduced TOGETHER (cross-row folds, as in
				// k_render_bwd): a wave-wide sum per entry was a dependent chain of seven instructions and a one-lane atomic each;
				// here the totals end up in different lanes, which add them with one atomic instruction (377 -> 358 us).
				static_assert(FR_RENDER_GROUP_PLAIN == 2 || FR_RENDER_GROUP_PLAIN == 4, "contribution fold: two or four entries per group");
				bool anyg = false;
#pragma unroll
				for (int g = 0; g < FR_RENDER_GROUP_PLAIN; g++) anyg = anyg || cany[g];
				if (__any(anyg))
				{
					if (FR_RENDER_GROUP_PLAIN == 2)
					{
						// lanes 0-31: entry 0, lanes 32-63: entry 1; row sums, then rows 1 / 3 add their lower neighbours
						float f = row_sum16(fold32(csum[0], csum[1]));
						f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x142, 0xA, 0xF, false)); // row_bcast15
						const bool has0 = __any(cany[0]), has1 = __any(cany[1]);
						if ((st == 31 && has0) || (st == 63 && has1)) atomicAdd(&a.contributions[sid[st == 31 ? jj[0] : jj[1]]], f);
					}
					else
					{
						// row r of 16 lanes: entry (r & 1) * 2 + (r >> 1) ... fold32 pairs (0,1) and (2,3), fold16 interleaves them
						const int gi = FR_RENDER_GROUP_PLAIN - 1; // (indices kept in range for the two-entry build)
						const float f01 = fold32(csum[0], csum[1 & gi]), f23 = fold32(csum[2 & gi], csum[3 & gi]);
						const float f = row_sum16(fold16(f01, f23)); // rows: entry 0, entry 2, entry 1, entry 3
						const int row = st >> 4;
						const int g = ((row & 1) << 1) | (row >> 1);
						const unsigned has = (__any(cany[0]) ? 1u : 0u) | (__any(cany[1 & gi]) ? 2u : 0u) | (__any(cany[2 & gi]) ? 4u : 0u) | (__any(cany[3 & gi]) ? 8u : 0u);
						const int jg = g == 0 ? jj[0] : (g == 1 ? jj[1 & gi] : (g == 2 ? jj[2 & gi] : jj[3 & gi]));
						if ((st & 15) == 15 && ((has >> g) & 1u)) atomicAdd(&a.contributions[sid[jg]], f);
					}
				}
			}
		}
	}

	report_consumed(a, tile, used, st, npairs);
	const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
	const size_t plane = (size_t)a.W * a.H;
#pragma unroll
	for (int k = 0; k < PPL; k++)
	{
		if (!inside[k]) continue;
		const size_t pid2 = (size_t)a.W * (size_t)(ty * FR_TILE + tile_row<PPL>(tid, k)) + px;
		const float Tk = S[k >> 1].Tk[k & 1];
		if (AUX) { a.final_T[pid2] = Tk; a.n_contrib[pid2] = last[k]; }
		if (LWMC) atomicAdd(&a.contributions[best_id[k]], a.loss_map[pid2]); // …_count forward.cu:435
		a.out_color[pid2] = fmaf(Tk, bg0, S[k >> 1].C0[k & 1]);
		a.out_color[plane + pid2] = fmaf(Tk, bg1, S[k >> 1].C1[k & 1]);
		a.out_color[2 * plane + pid2] = fmaf(Tk, bg2, S[k >> 1].C2[k & 1]);
	}
}

// ---------------- FOV_PCHECK_OBB: single-level and two-level tiles ----------------
// Work item = one wave's share of a tile: a band of eight rows, and for a two-level tile one of its two level states
// (every wave then carries one state: 63 registers, 8 waves per SIMD, half the instructions per entry -- the longest
// waves of a frame are those of two-level tiles). The two partial results o * w1 and q * (1 - w1) are the two rounded
// products the reference adds (forward.cu:466-470); they meet by float atomicAdd on a zero-filled pixel, and
// x + y == y + x bit for bit, so the image does not depend on who comes first.
//
// One single-wave workgroup per item, items laid out longest list first by k_tile_scan. What was measured on the way
// here (S-6M bench frames, kernel time): one workgroup per (tile, band, level state) with the upper-level workgroups of
// single-level tiles returning at once -- 43 % of the grid, interleaved with the real work -- 226 us, the wave slots
// never more than ~40 % full (per-wave timers); the compact item list below, same dispatch, 148 us; persistent waves
// pulling the items from 64 queues 182 us (a wave cannot migrate: at the tail some SIMDs still hold eight busy waves
// while others are empty; handing the slot back after every item: 152 us, i.e. the hardware's placement of fresh
// workgroups IS the load balancer); s_setprio by round or by remaining list length: nothing.
template <int PPL>
__global__ void __launch_bounds__(64, 8) k_render_fov(const RenderArgs a)
{
	static_assert(PPL == 2, "work items encode two bands per tile");
	constexpr int HP = PPL / 2;
	// three rows of 16 bytes per staged entry, all with the same stride: one address register serves the three reads
	__shared__ float4 s0[64];   // x, y, A, B
	__shared__ float4 s1[64];   // C, highest_level, -, -
	__shared__ float4 sl1[64];  // this wave's level: r, g, b, opacity

	const int lane = threadIdx.x;
	const uint32_t idx = blockIdx.x;
	if (idx >= a.totals[5] || a.totals[0] > a.capacity || a.totals[5] > gridDim.x) return;
	const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
	const size_t plane = (size_t)a.W * a.H;
	{
		const uint32_t item = a.render_items[idx];
		const int tile = (int)(item >> 3), wv = (int)(item & 1u);
		const bool two_level = (item & 4u) != 0;
		const bool upper = (item & 2u) != 0; // this wave carries the state of level L2
		const int tx = tile % a.gx, ty = tile / a.gx;
		const int tid = wv * 64 + lane; // position inside the tile's 256 / PPL threads (row mapping)
		const int lx = tid & 15;
		const int px = tx * FR_TILE + lx;
		const float pxf = (float)px;
		const uint2 range = a.ranges[tile];
		const int n = (int)(range.y - range.x);
		const float tlf = a.tile_lv[a.T + tile];                    // tile_min
		const int L1 = f2i(tlf);
		const int L2 = L1 + 1;
		const float L2f = tlf + 1.0f;
		const float tgx = a.tile_lv[2 * (size_t)a.T + tile], tgy = a.tile_lv[3 * (size_t)a.T + tile];

		// Per-lane state of the lane's pixels (rows ry, ry + 4), kept as a packed pair so that the blend runs on
		// v_pk_mul / v_pk_fma. A finished pixel carries its transmittance NEGATED: "still blending" is T > 0, no
		// separate flag registers, and |T| is the value the reference keeps.
		Px2k S1[HP];
		float pyf[PPL];
		bool inside[PPL];
		auto est_of = [&](int k) { return tlf + ((float)lx * tgx + (float)tile_row<PPL>(tid, k) * tgy) / (float)FR_TILE; };
#pragma unroll
		for (int k = 0; k < PPL; k++)
		{
			const int ly = tile_row<PPL>(tid, k);
			const int py = ty * FR_TILE + ly;
			pyf[k] = (float)py;
			inside[k] = px < a.W && py < a.H;
			// RF forward.cu:262-476: level L1 stops contributing beyond est > L2; single-level tiles have no second state
			const bool done1 = (two_level && !upper) ? (!inside[k] || (est_of(k) > (float)L2)) : !inside[k];
			S1[k >> 1].T[k & 1] = done1 ? 0.0f : 1.0f; // a finished pixel carries T = 0 (blend2k) ...
			S1[k >> 1].Tk[k & 1] = 1.0f;                // ... and the transmittance it ends with here
		}
#pragma unroll
		for (int h = 0; h < HP; h++) S1[h].C0 = S1[h].C1 = S1[h].C2 = (v2f){ 0.f, 0.f };

#ifdef FR_TILE_TIMERS
		const uint64_t tm0 = wall_clock64(); uint32_t tm_proc = 0, tm_batches = 0; uint64_t tm_loop = 0, tm_sync = 0;
#endif
		// prefetch registers
		float4 p0 = make_float4(0, 0, 0, 0), pl1 = p0;
		float2 p1 = make_float2(0, 0);
		auto fetch = [&](int e)
		{
			const uint32_t id = a.point_list[range.x + e];
			const float4 *r = a.rec + 3 * (size_t)id;
			p0 = r[0];
			const float4 r1 = r[1];
			p1 = make_float2(r1.x, r1.y);
			pl1 = a.lvl[(size_t)id * FR_FOV_LEVELS + (upper ? L2 : L1)];
		};
		if (lane < n) fetch(lane);
		int used = 0; // list entries this wave staged for blending (list_consumed)
		uint32_t npairs = 0;
		for (int base = 0; base < n; base += 64)
		{
			float tmax0 = -1.0f;
#pragma unroll
			for (int h = 0; h < HP; h++) tmax0 = fmaxf(tmax0, fmaxf(S1[h].T.x, S1[h].T.y));
#ifdef FR_TILE_TIMERS
			const uint64_t tq0 = wall_clock64();
#endif
			if (!__any(tmax0 > 0.0f)) break;
			// the previous batch has been read by all lanes (wave-synchronous, fenced)
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
#ifdef FR_TILE_TIMERS
			tm_sync += wall_clock64() - tq0;
#endif
			const bool staged = base + lane < n;
			used = min(n, base + 64);
			// The two skip tests that do not depend on the pixel's transmittance as ONE threshold on q = -power (RF forward.cu:556-566):
			// outside the support (power > 0 or power < -4.5) and alpha = o e^-q < 1/255 <=> q > ln(255 o). With tq = min(4.5, ln(255 o))
			// a pixel takes part iff 0 <= q <= tq: one unsigned comparison on q's bits (in_support()). ln(255 o) to an ulp (logf, not the
			// hardware's approximate log): the decision then differs from the reference's own fp32 evaluation of o * exp(power) < 1/255
			// only inside that expression's rounding, like the exp2-based test it replaces.
			const float lq = logf(255.0f * pl1.w);
			const float tq = lq >= 0.0f ? fminf(4.5f, lq) : 0.0f; // (false for the NaN of a negative opacity: takes part nowhere)
			if (staged) { s0[lane] = p0; s1[lane] = make_float4(p1.x, tq, 0.0f, 0.0f); sl1[lane] = pl1; }
			unsigned long long reach_mask;
			{
				// alpha < 1/255 everywhere (forward.cu:563) <=> power < -ln(255 opacity): tighter than -4.5 for faint splats
				const float thr = fmaxf(-4.5f, -lq - 0.01f);
				// a wave that carries the level-L2 state skips the Gaussians that do not exist at L2 (RF forward.cu:399: about
				// half the list in a 0/1 tile) here, at one lane's cost, instead of walking them as no-ops
				const bool exists = !upper || !((p1.y + 1.0f) < L2f);
				reach_mask = __ballot(staged && exists && lq >= 0.0f && band_reaches<PPL>(wv, tx, ty, p0.x, p0.y, p0.z, p0.w, p1.x, thr));
				npairs += (uint32_t)__popcll(reach_mask);
				}
			if (base + 64 + lane < n) fetch(base + 64 + lane);
			// lanes read entries other lanes staged
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			v2f pyp[HP];
#pragma unroll
			for (int h = 0; h < HP; h++) pyp[h] = (v2f){ pyf[2 * h], pyf[2 * h + 1] };
#ifdef FR_TILE_TIMERS
			const uint64_t tq1 = wall_clock64();
			tm_proc += (uint32_t)__popcll(reach_mask); tm_batches++;
#endif
			// Entries are taken two at a time: everything that does not depend on the running transmittance (record
			// fetch, power, exp, alpha inputs) is evaluated for both before either is blended, so the two dependency
			// chains overlap.
			struct Ent { v2f e[HP]; bool inx[HP], iny[HP]; float4 c1; };
			auto prepare = [&](const int j, const bool valid)
			{
				Ent t;
				const float4 g0 = s0[j];
				const float2 g1 = *(const float2 *)&s1[j];
				const float g1x = g1.x;
				const uint32_t tqb = __float_as_uint(g1.y);
				const float dx = g0.x - pxf;
				const float adx2 = (g0.z * dx) * dx;
				const float bdx = g0.w * dx;
#pragma unroll
				for (int h = 0; h < HP; h++)
				{
					const v2f q = qform2(g0.y - pyp[h], g1x, adx2, bdx);
					// in the splat's support: RF forward.cu:556-560 (power > 0 and power < -4.5 are skipped)
					t.inx[h] = valid && __float_as_uint(q.x) <= tqb; // ... and alpha >= 1/255
					t.iny[h] = valid && __float_as_uint(q.y) <= tqb;
					t.e[h] = exp_neg_pair(q);
				}
				t.c1 = sl1[j];
				return t;
			};
			for (unsigned long long rm = reach_mask; rm; )
			{
				int jj[FR_RENDER_GROUP];
				bool vv[FR_RENDER_GROUP];
#pragma unroll
				for (int g = 0; g < FR_RENDER_GROUP; g++)
				{
					vv[g] = rm != 0;
					jj[g] = vv[g] ? __builtin_ctzll(rm) : jj[0];
					rm &= rm - 1; // stays 0 once empty
				}
				float tmax = -1.0f;
#pragma unroll
				for (int h = 0; h < HP; h++) tmax = fmaxf(tmax, fmaxf(S1[h].T.x, S1[h].T.y));
				if (!__any(tmax > 0.0f)) break;
				Ent t[FR_RENDER_GROUP];
#pragma unroll
				for (int g = 0; g < FR_RENDER_GROUP; g++) t[g] = prepare(jj[g], vv[g]);
#pragma unroll
				for (int g = 0; g < FR_RENDER_GROUP; g++)
#pragma unroll
					for (int h = 0; h < HP; h++) blend2k<true>(S1[h], t[g].inx[h], t[g].iny[h], t[g].e[h], t[g].c1);
			}
#ifdef FR_TILE_TIMERS
			tm_loop += wall_clock64() - tq1;
#endif
		}

#ifdef FR_TILE_TIMERS
		if (lane == 0)
		{
			// developer build only (tools/tile_cycles.py): per-ITEM records in the otherwise unused final_T / n_contrib arrays
			const uint32_t G = 4u * (uint32_t)a.T, b = idx;
			a.final_T[b] = (float)(wall_clock64() - tm0); a.final_T[G + b] = (float)(tm0 & 0xffffff);
			a.n_contrib[b] = tm_proc; a.n_contrib[G + b] = tm_batches; a.n_contrib[2 * G + b] = (uint32_t)n;
			a.n_contrib[3 * G + b] = (uint32_t)tile | ((uint32_t)wv << 16) | ((uint32_t)upper << 20) | ((uint32_t)two_level << 21);
			a.n_contrib[4 * G + b] = (uint32_t)tm_loop; a.n_contrib[5 * G + b] = (uint32_t)tm_sync;
		}
#endif
		report_consumed(a, tile, used, lane, npairs);
#pragma unroll
		for (int k = 0; k < PPL; k++)
		{
			if (!inside[k]) continue;
			const size_t pid = (size_t)a.W * (size_t)(ty * FR_TILE + tile_row<PPL>(tid, k)) + px;
			const float t1 = S1[k >> 1].Tk[k & 1];
			const float o0 = fmaf(bg0, t1, S1[k >> 1].C0[k & 1]), o1 = fmaf(bg1, t1, S1[k >> 1].C1[k & 1]), o2 = fmaf(bg2, t1, S1[k >> 1].C2[k & 1]);
			if (two_level)
			{
				// RF forward.cu:455-470: C1 * w1 + C2 * (1 - w1), w1 = 1 - smoothstep
				float x = fabsf(est_of(k) - ((float)L1 + 0.5f)) / 0.5f;
				x = fmaxf(0.0f, fminf(1.0f, x));
				const float bT = 3 * x * x - 2 * x * x * x;
				const float w1 = 1 - bT;
				const float w = upper ? (1.f - w1) : w1;
				atomicAdd(&a.out_color[pid], o0 * w);
				atomicAdd(&a.out_color[plane + pid], o1 * w);
				atomicAdd(&a.out_color[2 * plane + pid], o2 * w);
				continue;
			}
			a.out_color[pid] = o0;
			a.out_color[plane + pid] = o1;
			a.out_color[2 * plane + pid] = o2;
		}
	}
}

// ---------------- NAIVE_FOV_PCHECK_OBB: the shared-model foveated baseline (SMFR) ----------------
// …_naive_pcheck_obb/cuda_rasterizer/forward.cu:482-580 (single-level tiles) and :258-480 (two-level tiles): one colour
// and one opacity per Gaussian, so both level states of a two-level tile see the same alpha and stay in ONE wave here
// (the exponential, alpha and colour fetch are shared; only the two transmittance chains differ). Quirks kept as
// written: the alpha < 1/255 skip applies to a Gaussian only while the pixel's L1 state is still open (once L1 is
// finished, L2 also accumulates negligible alphas, forward.cu:388-425); L2 skips Gaussians whose highest level lies
// below the tile's upper level; L1 starts finished where the pixel's estimated level is beyond L2.
__device__ __forceinline__ void blend2_given(Px2 &s, bool ok_x, bool ok_y, v2f alpha, float4 c)
{
	// ok: every skip test but saturation has passed (incl. T > 0)
	const v2f tt = s.T * (1.0f - alpha);
	v2f w = alpha * s.T;
	const bool sat_x = tt.x < 0.0001f, sat_y = tt.y < 0.0001f;
	w.x = (ok_x && !sat_x) ? w.x : 0.0f;
	w.y = (ok_y && !sat_y) ? w.y : 0.0f;
	s.C0 = __builtin_elementwise_fma((v2f){ c.x, c.x }, w, s.C0);
	s.C1 = __builtin_elementwise_fma((v2f){ c.y, c.y }, w, s.C1);
	s.C2 = __builtin_elementwise_fma((v2f){ c.z, c.z }, w, s.C2);
	s.T.x = ok_x ? (sat_x ? -s.T.x : tt.x) : s.T.x;
	s.T.y = ok_y ? (sat_y ? -s.T.y : tt.y) : s.T.y;
}

template <int PPL>
__global__ void __launch_bounds__(64) k_render_smfr(const RenderArgs a)
{
	static_assert(PPL == 2, "work items encode two bands per tile");
	__shared__ float4 s0[64];   // x, y, A, B
	__shared__ float4 s1[64];   // C, opacity, highest level, -
	__shared__ float4 scol[64]; // r, g, b, -
	const int lane = threadIdx.x;
	if (blockIdx.x >= a.totals[5] || a.totals[0] > a.capacity || a.totals[5] > gridDim.x) return;
	const uint32_t item = a.render_items[blockIdx.x];
	const int tile = (int)(item >> 3), wv = (int)(item & 1u);
	const int tx = tile % a.gx, ty = tile / a.gx;
	const int tid = wv * 64 + lane;
	const int lx = tid & 15;
	const int px = tx * FR_TILE + lx;
	const float pxf = (float)px;
	const uint2 range = a.ranges[tile];
	const int n = (int)(range.y - range.x);
	const float tlf = a.tile_lv[a.T + tile];                    // tile_min
	const bool two_level = a.tile_lv[4 * (size_t)a.T + tile] != 0.0f;
	const int L1 = f2i(tlf);
	const int L2 = L1 + 1;
	const float L2f = tlf + 1.0f;
	const float tgx = a.tile_lv[2 * (size_t)a.T + tile], tgy = a.tile_lv[3 * (size_t)a.T + tile];
	Px2 S1, S2;
	float pyf[PPL], est[PPL];
	bool inside[PPL];
#pragma unroll
	for (int k = 0; k < PPL; k++)
	{
		const int ly = tile_row<PPL>(tid, k);
		const int py = ty * FR_TILE + ly;
		pyf[k] = (float)py;
		inside[k] = px < a.W && py < a.H;
		est[k] = tlf + ((float)lx * tgx + (float)ly * tgy) / (float)FR_TILE;
		const bool done1 = !inside[k] || (two_level && est[k] > (float)L2);
		const bool done2 = !inside[k] || !two_level;
		S1.T[k] = done1 ? -1.0f : 1.0f;
		S2.T[k] = done2 ? -1.0f : 1.0f;
	}
	S1.C0 = S1.C1 = S1.C2 = S2.C0 = S2.C1 = S2.C2 = (v2f){ 0.f, 0.f };
	float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, pc = p0;
	auto fetch = [&](int e)
	{
		const uint32_t id = a.point_list[range.x + e];
		const float4 *r = a.rec + 3 * (size_t)id;
		p0 = r[0];
		const float4 r1 = r[1], r2 = r[2];
		p1 = make_float4(r1.x, r1.y, r2.z, 0.0f);   // conic c, opacity, highest level (k_bin keeps it in the clamp slot)
		pc = make_float4(r1.z, r1.w, r2.x, 0.0f);
	};
	if (lane < n) fetch(lane);
	int used = 0;
	for (int base = 0; base < n; base += 64)
	{
		const float tmax0 = fmaxf(fmaxf(S1.T.x, S1.T.y), fmaxf(S2.T.x, S2.T.y));
		if (!__any(tmax0 > 0.0f)) break;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		used = min(n, base + 64);
		const bool staged = base + lane < n;
		if (staged) { s0[lane] = p0; s1[lane] = p1; scol[lane] = pc; }
		unsigned long long reach_mask;
		{
			// single-level tiles: alpha < 1/255 everywhere <=> power < -ln(255 opacity); in two-level tiles the L2 state may
			// take negligible alphas (see above), so only the support cutoff bounds the reach there
			const float thr = two_level ? -4.5f : fmaxf(-4.5f, -__logf(255.0f * p1.y) - 0.01f);
			reach_mask = __ballot(staged && band_reaches<PPL>(wv, tx, ty, p0.x, p0.y, p0.z, p0.w, p1.x, thr));
		}
		if (base + 64 + lane < n) fetch(base + 64 + lane);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		const v2f pyp = (v2f){ pyf[0], pyf[1] };
		for (unsigned long long rm = reach_mask; rm; rm &= rm - 1)
		{
			const int j = __builtin_ctzll(rm);
			const float tmax = fmaxf(fmaxf(S1.T.x, S1.T.y), fmaxf(S2.T.x, S2.T.y));
			if (!__any(tmax > 0.0f)) break;
			const float4 g0 = s0[j], g1 = s1[j], col = scol[j];
			const float dx = g0.x - pxf;
			const v2f pw = power2(g0.y - pyp, g1.x, (g0.z * dx) * dx, g0.w * dx);
			const bool inx = !(pw.x > 0.0f || pw.x < -4.5f), iny = !(pw.y > 0.0f || pw.y < -4.5f);
			v2f alpha = g1.y * exp2_pair(pw);
			alpha.x = fminf(0.99f, alpha.x); alpha.y = fminf(0.99f, alpha.y);
			const bool vis_x = !(alpha.x < 1.0f / 255.0f), vis_y = !(alpha.y < 1.0f / 255.0f);
			const bool open1_x = S1.T.x > 0.0f, open1_y = S1.T.y > 0.0f; // L1 still open BEFORE this Gaussian
			blend2_given(S1, inx && open1_x && vis_x, iny && open1_y && vis_y, alpha, col);
			if (two_level)
			{
				const bool l2_ok = !((g1.z + 1.0f) < L2f);
				// naive forward.cu:388-406: an open L1 that finds alpha negligible skips the Gaussian for L2 as well
				blend2_given(S2, inx && l2_ok && S2.T.x > 0.0f && (!open1_x || vis_x), iny && l2_ok && S2.T.y > 0.0f && (!open1_y || vis_y), alpha, col);
			}
		}
	}
	report_consumed(a, tile, used, lane);
	const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
	const size_t plane = (size_t)a.W * a.H;
#pragma unroll
	for (int k = 0; k < PPL; k++)
	{
		if (!inside[k]) continue;
		const size_t pid = (size_t)a.W * (size_t)(ty * FR_TILE + tile_row<PPL>(tid, k)) + px;
		const float t1 = fabsf(S1.T[k]), t2 = fabsf(S2.T[k]);
		float o0 = fmaf(bg0, t1, S1.C0[k]), o1 = fmaf(bg1, t1, S1.C1[k]), o2 = fmaf(bg2, t1, S1.C2[k]);
		if (two_level)
		{
			const float q0 = fmaf(bg0, t2, S2.C0[k]), q1 = fmaf(bg1, t2, S2.C1[k]), q2 = fmaf(bg2, t2, S2.C2[k]);
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

// ---------------- MMFR_PCHECK_OBB: one level's share of the multi-model foveated baseline ----------------
// …_mmfr_pcheck_obb/cuda_rasterizer/forward.cu:255-420 (two-level tiles) and :422-540 (single-level tiles): plain
// front-to-back blend of this level's model; skipped tiles stay zero; in a two-level tile a pixel whose estimated level
// est = tile_min + gradient . offset has int(est) != cur_level and lies below the blend zone contributes nothing, and
// every pixel is weighted by w1 = 1 - smoothstep((est - (int(est) + 0.5)) / 0.5) if int(est) == cur_level, else 1 - w1.
template <int PPL>
__global__ void __launch_bounds__(64) k_render_mmfr(const RenderArgs a)
{
	static_assert(PPL == 2, "work items encode two bands per tile");
	__shared__ float4 s0[64];   // x, y, A, B
	__shared__ float2 s1[64];   // C, opacity
	__shared__ float4 scol[64]; // r, g, b, -
	const int lane = threadIdx.x;
	if (blockIdx.x >= a.totals[5] || a.totals[0] > a.capacity || a.totals[5] > gridDim.x) return;
	const uint32_t item = a.render_items[blockIdx.x];
	const int tile = (int)(item >> 3), wv = (int)(item & 1u);
	const int tx = tile % a.gx, ty = tile / a.gx;
	const int tid = wv * 64 + lane;
	const int lx = tid & 15;
	const int px = tx * FR_TILE + lx;
	const float pxf = (float)px;
	const bool skipped = a.tile_lv[a.T + tile] != 0.0f;          // the filter key k_tile_levels leaves in row 1
	const uint2 range = a.ranges[tile];
	const int n = skipped ? 0 : (int)(range.y - range.x);
	const float tlf = a.tile_lv[tile];                             // row 0: tile_min clamped at 0
	const bool two_level = a.tile_lv[4 * (size_t)a.T + tile] != 0.0f;
	const float tgx = a.tile_lv[2 * (size_t)a.T + tile], tgy = a.tile_lv[3 * (size_t)a.T + tile];
	Px2 S;
	float pyf[PPL], wgt[PPL];
	bool inside[PPL];
#pragma unroll
	for (int k = 0; k < PPL; k++)
	{
		const int ly = tile_row<PPL>(tid, k);
		const int py = ty * FR_TILE + ly;
		pyf[k] = (float)py;
		inside[k] = px < a.W && py < a.H;
		bool done = !inside[k];
		wgt[k] = 1.0f;
		if (two_level)
		{
			const float est = tlf + ((float)lx * tgx + (float)ly * tgy) / (float)FR_TILE;
			const int L1 = f2i(est);
			float x = (est - ((float)L1 + 0.5f)) / 0.5f;
			const bool mine = (float)L1 == a.cur_level;
			if (x < 0.0f && !mine) done = true;
			x = fmaxf(0.0f, fminf(1.0f, x));
			const float bT = 3 * x * x - 2 * x * x * x;
			const float w1 = 1 - bT;
			wgt[k] = mine ? w1 : (float)(1.0 - (double)w1); // `1.0 - L1_w` is evaluated in double in the reference
		}
		S.T[k] = done ? -1.0f : 1.0f;
	}
	S.C0 = S.C1 = S.C2 = (v2f){ 0.f, 0.f };
	float4 p0 = make_float4(0, 0, 0, 0), pc = p0;
	float2 p1 = make_float2(0, 0);
	auto fetch = [&](int e)
	{
		const uint32_t id = a.point_list[range.x + e];
		const float4 *r = a.rec + 3 * (size_t)id;
		p0 = r[0];
		const float4 r1 = r[1];
		p1 = make_float2(r1.x, r1.y);
		pc = make_float4(r1.z, r1.w, r[2].x, 0.0f);
	};
	if (lane < n) fetch(lane);
	int used = 0;
	for (int base = 0; base < n; base += 64)
	{
		if (!__any(fmaxf(S.T.x, S.T.y) > 0.0f)) break;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		used = min(n, base + 64);
		const bool staged = base + lane < n;
		if (staged) { s0[lane] = p0; s1[lane] = p1; scol[lane] = pc; }
		const float thr = fmaxf(-4.5f, -__logf(255.0f * p1.y) - 0.01f);
		const unsigned long long reach_mask = __ballot(staged && band_reaches<PPL>(wv, tx, ty, p0.x, p0.y, p0.z, p0.w, p1.x, thr));
		if (base + 64 + lane < n) fetch(base + 64 + lane);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		const v2f pyp = (v2f){ pyf[0], pyf[1] };
		for (unsigned long long rm = reach_mask; rm; rm &= rm - 1)
		{
			const int j = __builtin_ctzll(rm);
			if (!__any(fmaxf(S.T.x, S.T.y) > 0.0f)) break;
			const float4 g0 = s0[j];
			const float2 g1 = s1[j];
			const float4 col = scol[j];
			const float dx = g0.x - pxf;
			const v2f pw = power2(g0.y - pyp, g1.x, (g0.z * dx) * dx, g0.w * dx);
			blend2(S, !(pw.x > 0.0f || pw.x < -4.5f), !(pw.y > 0.0f || pw.y < -4.5f), exp2_pair(pw), make_float4(col.x, col.y, col.z, g1.y));
		}
	}
	report_consumed(a, tile, used, lane);
	const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
	const size_t plane = (size_t)a.W * a.H;
#pragma unroll
	for (int k = 0; k < PPL; k++)
	{
		if (!inside[k]) continue;
		const size_t pid = (size_t)a.W * (size_t)(ty * FR_TILE + tile_row<PPL>(tid, k)) + px;
		const float t1 = fabsf(S.T[k]);
		const float w = skipped ? 0.0f : wgt[k];
		// skipped tiles are never written by the reference (the image starts as zeros); w == 1 in single-level tiles
		a.out_color[pid] = skipped ? 0.0f : (two_level ? fmaf(bg0, t1, S.C0[k]) * w : fmaf(bg0, t1, S.C0[k]));
		a.out_color[plane + pid] = skipped ? 0.0f : (two_level ? fmaf(bg1, t1, S.C1[k]) * w : fmaf(bg1, t1, S.C1[k]));
		a.out_color[2 * plane + pid] = skipped ? 0.0f : (two_level ? fmaf(bg2, t1, S.C2[k]) * w : fmaf(bg2, t1, S.C2[k]));
	}
}

int launch_render(FwdCtx &c)
{
	const fr_forward_args *a = c.a;
	RenderArgs r;
	r.W = a->W; r.H = a->H; r.gx = c.gx;
	r.ranges = c.img.ranges; r.point_list = c.bin.point_list; r.vis_list = c.geom.vis_list; r.rec = c.geom.rec; r.lvl = c.geom.lvl;
	r.tile_lv = c.img.tile_lv; r.tile_order = c.img.tile_order; r.T = c.T; r.bg = a->background; r.out_color = a->out_color;
	r.final_T = c.img.final_T; r.n_contrib = c.img.n_contrib;
	r.gaussians_count = a->gaussians_count; r.contributions = a->contributions; r.loss_map = a->loss_map;
	r.render_items = c.img.render_items; r.totals = c.img.totals; r.capacity = (uint32_t)c.capacity; r.cur_level = a->cur_level;
	constexpr int PPL = 2;
	r.round_flags = c.bin.round_flags;
	r.consumed = a->list_consumed; r.pairs = a->blend_pairs;
	if (has_stats(a->variant) && !a->no_stats && a->variant != FR_VARIANT_PCHECK_OBB_MAX && c.bin.round_flags)
	{
		const hipError_t e = hipMemsetAsync(c.bin.round_flags, 0, round_flag_words(c.capacity, c.T) * sizeof(uint32_t), c.stream);
		if (e != hipSuccess) { set_error("hipMemsetAsync(round_flags): %s", hipGetErrorString(e)); return FR_ERR_HIP; }
	}
	// work items: two bands per tile, and for RF two more per two-level tile -- how many of those the frame has is only known
	// on the device until the host has read the counts: the grid is then FwdCtx::items_cap (a little more than the previous
	// frame of the kind had; surplus workgroups leave at once -- they sit at the END of the grid, behind all the work -- and
	// a frame with more items than that is not blended but replayed, see frame_fits)
	const unsigned n_items = (unsigned)c.n_items;
#define FR_LAUNCH_RENDER(V, UNUSED_) hipLaunchKernelGGL((k_render<V, PPL>), dim3(n_items), dim3(64), 0, c.stream, r)
	switch (a->variant)
	{
	case FR_VARIANT_ORIGINAL: FR_LAUNCH_RENDER(FR_VARIANT_ORIGINAL, true); break;
	case FR_VARIANT_PCHECK_OBB_SUM:
		if (a->no_stats) FR_LAUNCH_RENDER(FR_VARIANT_SUM_NOSTATS, false); else FR_LAUNCH_RENDER(FR_VARIANT_PCHECK_OBB_SUM, false);
		break;
	case FR_VARIANT_PCHECK_OBB: FR_LAUNCH_RENDER(FR_VARIANT_PCHECK_OBB, true); break;
	case FR_VARIANT_PCHECK_OBB_MAX: FR_LAUNCH_RENDER(FR_VARIANT_PCHECK_OBB_MAX, true); break;
	case FR_VARIANT_PCHECK_OBB_LWMC: FR_LAUNCH_RENDER(FR_VARIANT_PCHECK_OBB_LWMC, false); break;
	case FR_VARIANT_NAIVE_FOV_PCHECK_OBB: hipLaunchKernelGGL((k_render_smfr<2>), dim3(n_items), dim3(64), 0, c.stream, r); break;
	case FR_VARIANT_MMFR_PCHECK_OBB: hipLaunchKernelGGL((k_render_mmfr<2>), dim3(n_items), dim3(64), 0, c.stream, r); break;
	default:
		hipLaunchKernelGGL((k_render_fov<2>), dim3(n_items), dim3(64), 0, c.stream, r);
		break;
	}
#undef FR_LAUNCH_RENDER
	return check_launch("render", c.stream, a->debug);
}

} // namespace fr
