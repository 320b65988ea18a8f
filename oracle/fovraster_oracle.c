/*
 * fovraster_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the rasterizer hot path of
 * horizon-research/Fov-3DGS (the tile-based 3D-Gaussian-splatting rasterizer
 * behind GaussianRasterizer / gaussian_renderer_fov.render()).  It exists so
 * that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
 * check / time-beside the HIP product path.  Nothing in the product package
 * may import, link or call it.
 *
 * PARITY STATUS: "parity unpinned by the reference's own binaries" -- the
 * reference arithmetic lives in CUDA (.cu) files that need nvcc + CUB and an
 * NVIDIA GPU; neither exists in the build container nor on the MI355X box, and
 * the reference ships no tests / golden vectors for this path.
 * The oracle is instead pinned (tests/test_oracle_pins.py) against the pieces of
 * the reference that DO import on CPU (utils/sh_utils.eval_sh,
 * utils/graphics_utils.getProjectionMatrix/getWorld2View2,
 * odak_perception.foveation.make_pooling_size_map_pixels) through golden vectors
 * committed under tests/golden/, and its backward is pinned against central
 * finite differences of its own forward in the double-precision build.
 *
 * Reference files followed (paths relative to
 * /root/reference/fov3dgs/submodules/):
 *   R0 = diff-gaussian-rasterization/cuda_rasterizer/
 *   RS = diff-gaussian-rasterization_pcheck_obb_sum/cuda_rasterizer/
 *   RP = diff-gaussian-rasterization_pcheck_obb/cuda_rasterizer/
 *   RF = diff-gaussian-rasterization_fov_pcheck_obb/cuda_rasterizer/
 * Every function below cites the file:line it restates.
 *
 * Build flavours: default `real` = float (bit-level restatement of the fp32
 * reference, including the places where C promotes to double); -DORC_DOUBLE
 * makes `real` = double (used only for finite-difference pins).
 * Compile with -ffp-contract=off so gcc does not fuse multiply-adds.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Threads: every loop over Gaussians / tiles below is independent per iteration, so the build with -fopenmp may
 * spread them over host cores (orc_set_threads; default 1 = the plain serial program). Integer / index outputs and
 * images are identical for any thread count (each iteration owns its outputs; the instance order is a total order);
 * only the per-Gaussian sums that cross tiles (RS contributions, backward gradients) are accumulated by atomic adds
 * in DOUBLE, whose order may move the rounded float by an ulp. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int orc_get_threads(void) { return g_threads; }
#ifdef _OPENMP
#define ORC_PRAGMA(x) _Pragma(#x)
#define ORC_PARALLEL_FOR(...) ORC_PRAGMA(omp parallel for schedule(__VA_ARGS__) num_threads(g_threads))
#define ORC_ATOMIC ORC_PRAGMA(omp atomic)
#else
#define ORC_PARALLEL_FOR(...)
#define ORC_ATOMIC
#endif

#ifdef ORC_DOUBLE
typedef double real;
#define r_sqrt sqrt
#define r_exp exp
#define r_ceil ceil
#define r_acos acos
#define r_tan tan
#define r_fabs fabs
#define r_fmin fmin
#define r_fmax fmax
#else
typedef float real;
#define r_sqrt sqrtf
#define r_exp expf
#define r_ceil ceilf
#define r_acos acosf
#define r_tan tanf
#define r_fabs fabsf
#define r_fmin fminf
#define r_fmax fmaxf
#endif

#define BLOCK_X 16
#define BLOCK_Y 16
#define BLOCK_SIZE 256
#define FOV_NUM 4 /* RF auxiliary.h:26 */

enum { ORC_R0 = 0, ORC_RS = 1, ORC_RP = 2, ORC_RF = 3, ORC_RMAX = 4, ORC_LWMC = 5, ORC_SMFR = 6, ORC_MMFR = 7 };
/* ORC_MMFR = …_mmfr_pcheck_obb, one level's share of the multi-model foveated baseline
 * (MM = diff-gaussian-rasterization_mmfr_pcheck_obb/cuda_rasterizer/). */
/* ORC_SMFR = …_naive_pcheck_obb, the paper's shared-model foveated baseline: RF's tile levels, level filter and lists,
 * one colour / opacity per Gaussian (NV = diff-gaussian-rasterization_naive_pcheck_obb/cuda_rasterizer/). */
#define ORC_IS_FOV(v) ((v) == ORC_RF || (v) == ORC_SMFR || (v) == ORC_MMFR)
/* ORC_RMAX = …_pcheck_obb_max, ORC_LWMC = …_pcheck_obb_loss_weighted_max_count: RS with different
 * per-Gaussian statistics (pruning metrics of prune.py / metric_mask_learn.py); same backward as RS. */

typedef struct {
	int32_t variant;
	int32_t P, D, M, W, H;
	int32_t prefiltered;
	int32_t pad_;
	real tanfovx, tanfovy, scale_modifier;
	real gaze_x, gaze_y, alpha;
	const real *bg, *viewmatrix, *projmatrix, *campos;
	const real *means3D, *scales, *rotations, *opacities, *shs;
	const real *cov3D_precomp, *colors_precomp;
	const real *shs_dcs, *highest_levels;
	const real *loss_map; /* LWMC: per-pixel weights, first H*W values are used */
	/* optional tile window [x0,x1) x [y0,y1) restricting binning + blending (bench cpu_baseline
	 * sampling only; all zeros = whole frame) */
	int32_t win[4];
	real cur_level; /* MMFR: the level this call renders */
} orc_in;

static int in_window(const orc_in *in, int tx, int ty)
{
	if (in->win[2] <= in->win[0] || in->win[3] <= in->win[1]) return 1;
	return tx >= in->win[0] && tx < in->win[2] && ty >= in->win[1] && ty < in->win[3];
}

typedef struct {
	/* per Gaussian */
	real *depths;          /* [P] */
	int32_t *radii;        /* [P] (after OBB/foveal cull reset) */
	real *means2D;         /* [P,2] */
	real *cov3D;           /* [P,6] */
	real *conic;           /* [P,3] */
	real *rgb;             /* [P,3] (R0/RS/RP) */
	uint8_t *clamped;      /* [P,3] */
	uint32_t *tiles_rect;  /* [P] rect tile count (pre-cull) */
	uint32_t *tiles_touched; /* [P] post cull */
	real *eigen_len;       /* [P,2] */
	real *eigen_vec;       /* [P,4] */
	int32_t *level_ranges; /* [P,2] RF */
	real *fov_colors;      /* [P,4,3] RF (unwritten slots left untouched) */
	/* per tile */
	real *tile_levels, *tile_gx, *tile_gy, *tile_min; /* [T] RF */
	uint8_t *tile_blend;   /* [T] RF */
	uint32_t *ranges;      /* [T,2] */
	/* binning */
	int64_t capacity;      /* entries available in point_list / keys */
	uint32_t *point_list;  /* [capacity] sorted gaussian ids */
	uint64_t *keys;        /* [capacity] sorted (tile<<32 | depth bits); float build only */
	int64_t num_rect;      /* out: instances before cull (D0) */
	int64_t num_rendered;  /* out: instances after cull (D) */
	/* image */
	real *color;           /* [3,H,W] */
	real *final_T;         /* [H,W] (R0/RS; RP/RF do not write it in the reference) */
	uint32_t *n_contrib;   /* [H,W] */
	/* RS */
	int32_t *gaussians_count; /* [P] */
	real *contributions;   /* [P] */
} orc_out;

typedef struct {
	const real *dL_dpix;   /* [3,H,W] */
	real *dL_dmean2D;      /* [P,3] */
	real *dL_dconic;       /* [P,4] (x,y,_,w) */
	real *dL_dopacity;     /* [P] */
	real *dL_dcolor;       /* [P,3] */
	real *dL_dmean3D;      /* [P,3] */
	real *dL_dcov3D;       /* [P,6] */
	real *dL_dsh;          /* [P,M,3] */
	real *dL_dscale;       /* [P,3] */
	real *dL_drot;         /* [P,4] */
} orc_grads;

/* ---- constants: R0 auxiliary.h:22-39 ---- */
static const real SH_C0 = (real)0.28209479177387814f;
static const real SH_C1 = (real)0.4886025119029199f;
static const real SH_C2[5] = { (real)1.0925484305920792f, (real)-1.0925484305920792f,
	(real)0.31539156525252005f, (real)-1.0925484305920792f, (real)0.5462742152960396f };
static const real SH_C3[7] = { (real)-0.5900435899266435f, (real)2.890611442640554f,
	(real)-0.4570457994644658f, (real)0.3731763325901154f, (real)-0.4570457994644658f,
	(real)1.445305721320277f, (real)-0.5900435899266435f };

/* saturating float->int like cvt.rzi.s32.f32 (NaN -> 0) */
static int f2i(real v)
{
	if (v != v) return 0;
	if (v >= (real)2147483648.0) return 2147483647;
	if (v <= (real)-2147483648.0) return (-2147483647 - 1);
	return (int)v;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* ---- tiny glm-like column-major mat3: m.c[col][row]; product order as glm ---- */
typedef struct { real c[3][3]; } m3;
static m3 m3_cols(real a0, real a1, real a2, real b0, real b1, real b2, real c0, real c1, real c2)
{
	m3 m; m.c[0][0] = a0; m.c[0][1] = a1; m.c[0][2] = a2;
	m.c[1][0] = b0; m.c[1][1] = b1; m.c[1][2] = b2;
	m.c[2][0] = c0; m.c[2][1] = c1; m.c[2][2] = c2; return m;
}
static m3 m3_mul(m3 a, m3 b)
{
	m3 r;
	for (int col = 0; col < 3; col++)
		for (int row = 0; row < 3; row++)
			r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2];
	return r;
}
static m3 m3_t(m3 a)
{
	m3 r;
	for (int col = 0; col < 3; col++)
		for (int row = 0; row < 3; row++) r.c[col][row] = a.c[row][col];
	return r;
}
static m3 m3_scale(real s, m3 a)
{
	for (int col = 0; col < 3; col++)
		for (int row = 0; row < 3; row++) a.c[col][row] = s * a.c[col][row];
	return a;
}

/* R0 auxiliary.h:41-44 (the arithmetic is double in the reference: literals 1.0/0.5) */
static real ndc2Pix(real v, int S)
{
	return (real)((((double)v + 1.0) * S - 1.0) * 0.5);
}

/* R0 auxiliary.h:46-56 */
static void getRect(real px, real py, int max_radius, int rmin[2], int rmax[2], int gx, int gy)
{
	real r = (real)max_radius;
	rmin[0] = imin(gx, imax(0, f2i((px - r) / BLOCK_X)));
	rmin[1] = imin(gy, imax(0, f2i((py - r) / BLOCK_Y)));
	rmax[0] = imin(gx, imax(0, f2i((px + r + BLOCK_X - 1) / BLOCK_X)));
	rmax[1] = imin(gy, imax(0, f2i((py + r + BLOCK_Y - 1) / BLOCK_Y)));
}

/* R0 auxiliary.h:58-77 */
static void transformPoint4x3(const real p[3], const real *m, real o[3])
{
	o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
	o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
	o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void transformPoint4x4(const real p[3], const real *m, real o[4])
{
	o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
	o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
	o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
	o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* SH basis-weighted sum.  first=0: R0 forward.cu:20-71 (16 coeffs incl. DC);
 * first=1: RF rasterizer_impl.cu:37-84 (rest only, sh[k-1], no DC term). */
static void sh_eval(int deg, const real *sh /* points at coefficient 0 (or 1 if rest) */, int rest,
	const real pos[3], const real campos[3], real out[3])
{
	real dir[3] = { pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2] };
	real len = r_sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
	dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
	real x = dir[0], y = dir[1], z = dir[2];
	for (int ch = 0; ch < 3; ch++)
	{
#define SHK(k) sh[3 * ((k) - rest) + ch] /* rest: coefficient k lives at slot k-1 */
		real result = rest ? (real)0 : SH_C0 * SHK(0);
		if (deg > 0)
		{
			result = result - SH_C1 * y * SHK(1) + SH_C1 * z * SHK(2) - SH_C1 * x * SHK(3);
			if (deg > 1)
			{
				real xx = x * x, yy = y * y, zz = z * z;
				real xy = x * y, yz = y * z, xz = x * z;
				result = result +
					SH_C2[0] * xy * SHK(4) +
					SH_C2[1] * yz * SHK(5) +
					SH_C2[2] * ((real)2.0 * zz - xx - yy) * SHK(6) +
					SH_C2[3] * xz * SHK(7) +
					SH_C2[4] * (xx - yy) * SHK(8);
				if (deg > 2)
				{
					result = result +
						SH_C3[0] * y * ((real)3.0 * xx - yy) * SHK(9) +
						SH_C3[1] * xy * z * SHK(10) +
						SH_C3[2] * y * ((real)4.0 * zz - xx - yy) * SHK(11) +
						SH_C3[3] * z * ((real)2.0 * zz - (real)3.0 * xx - (real)3.0 * yy) * SHK(12) +
						SH_C3[4] * x * ((real)4.0 * zz - xx - yy) * SHK(13) +
						SH_C3[5] * z * (xx - yy) * SHK(14) +
						SH_C3[6] * x * (xx - (real)3.0 * yy) * SHK(15);
				}
			}
		}
#undef SHK
		out[ch] = result + (real)0.5;
	}
}

/* R0 forward.cu:118-152 */
static void computeCov3D(const real scale[3], real mod, const real rot[4], real cov3D[6])
{
	m3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
	S.c[0][0] = mod * scale[0]; S.c[1][1] = mod * scale[1]; S.c[2][2] = mod * scale[2];
	real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
	m3 R = m3_cols(
		(real)1 - (real)2 * (y * y + z * z), (real)2 * (x * y - r * z), (real)2 * (x * z + r * y),
		(real)2 * (x * y + r * z), (real)1 - (real)2 * (x * x + z * z), (real)2 * (y * z - r * x),
		(real)2 * (x * z - r * y), (real)2 * (y * z + r * x), (real)1 - (real)2 * (x * x + y * y));
	m3 M = m3_mul(S, R);
	m3 Sigma = m3_mul(m3_t(M), M);
	cov3D[0] = Sigma.c[0][0]; cov3D[1] = Sigma.c[0][1]; cov3D[2] = Sigma.c[0][2];
	cov3D[3] = Sigma.c[1][1]; cov3D[4] = Sigma.c[1][2]; cov3D[5] = Sigma.c[2][2];
}

/* R0 forward.cu:74-113 */
static void computeCov2D(const real mean[3], real fx, real fy, real tfx, real tfy, const real *cov3D,
	const real *vm, real cov[3])
{
	real t[3];
	transformPoint4x3(mean, vm, t);
	const real limx = (real)1.3f * tfx, limy = (real)1.3f * tfy;
	const real txtz = t[0] / t[2], tytz = t[1] / t[2];
	t[0] = r_fmin(limx, r_fmax(-limx, txtz)) * t[2];
	t[1] = r_fmin(limy, r_fmax(-limy, tytz)) * t[2];
	m3 J = m3_cols(fx / t[2], 0, -(fx * t[0]) / (t[2] * t[2]),
		0, fy / t[2], -(fy * t[1]) / (t[2] * t[2]),
		0, 0, 0);
	m3 Wm = m3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
	m3 T = m3_mul(Wm, J);
	m3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
	m3 c = m3_mul(m3_mul(m3_t(T), m3_t(Vrk)), T);
	c.c[0][0] += (real)0.3f;
	c.c[1][1] += (real)0.3f;
	cov[0] = c.c[0][0]; cov[1] = c.c[0][1]; cov[2] = c.c[1][1];
}

/* RS auxiliary.h:66-154 (= RF auxiliary.h:80-168): separating-axis test of a 16x16 tile
 * (centre tile_p, half extent 8) against the oriented 3-sigma box. fminf/fmaxf NaN
 * semantics (return the non-NaN operand) are those of C99 fmin/fmax. */
static int OBB_check(real tpx, real tpy, const real vtx[4][2], const real center[2],
	const real v1[2], const real v2[2], real len1, real len2)
{
	real rel[4][2];
	for (int i = 0; i < 4; i++) { rel[i][0] = vtx[i][0] - tpx; rel[i][1] = vtx[i][1] - tpy; }
	real mn = rel[0][0], mx = mn;
	for (int i = 1; i < 4; i++) { mn = r_fmin(mn, rel[i][0]); mx = r_fmax(mx, rel[i][0]); }
	if (mx < (real)-8.0 || mn > (real)8.0) return 0;
	mn = rel[0][1]; mx = mn;
	for (int i = 1; i < 4; i++) { mn = r_fmin(mn, rel[i][1]); mx = r_fmax(mx, rel[i][1]); }
	if (mx < (real)-8.0 || mn > (real)8.0) return 0;
	real tv[4][2] = {
		{ tpx + (real)8.0 - center[0], tpy + (real)8.0 - center[1] },
		{ tpx - (real)8.0 - center[0], tpy + (real)8.0 - center[1] },
		{ tpx - (real)8.0 - center[0], tpy - (real)8.0 - center[1] },
		{ tpx + (real)8.0 - center[0], tpy - (real)8.0 - center[1] } };
	real d0 = tv[0][0] * v1[0] + tv[0][1] * v1[1];
	mn = d0; mx = d0;
	for (int i = 1; i < 4; i++) { real d = tv[i][0] * v1[0] + tv[i][1] * v1[1]; mn = r_fmin(mn, d); mx = r_fmax(mx, d); }
	if (len1 < mn || -len1 > mx) return 0;
	d0 = tv[0][0] * v2[0] + tv[0][1] * v2[1];
	mn = d0; mx = d0;
	for (int i = 1; i < 4; i++) { real d = tv[i][0] * v2[0] + tv[i][1] * v2[1]; mn = r_fmin(mn, d); mx = r_fmax(mx, d); }
	if (len2 < mn || -len2 > mx) return 0;
	return 1;
}

/* ---------------- preprocess ----------------
 * R0 forward.cu:155-262; RS forward.cu:155-293 (adds eigen data); RF forward.cu:105-238
 * (no SH colour, opacity kept per level); RP identical to RS without the SH colour.
 * Returns 0 ok, -1 if `prefiltered` is set but a point is near-culled (reference __trap()s). */
static int preprocess(const orc_in *in, orc_out *o)
{
	const int P = in->P, W = in->W, H = in->H;
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	const real focal_y = H / ((real)2.0 * in->tanfovy); /* R0 rasterizer_impl.cu:222-223 */
	const real focal_x = W / ((real)2.0 * in->tanfovx);
	const int with_eigen = in->variant != ORC_R0;
	const int with_sh = in->variant != ORC_RF;
	int trapped = 0;
	ORC_PARALLEL_FOR(static)
	for (int idx = 0; idx < P; idx++)
	{
		o->radii[idx] = 0;
		o->tiles_rect[idx] = 0;
		o->tiles_touched[idx] = 0;
		const real *p_orig = in->means3D + 3 * idx;
		/* in_frustum: R0 auxiliary.h:139-164 */
		real p_hom[4], p_view[3];
		transformPoint4x4(p_orig, in->projmatrix, p_hom);
		real p_w = (real)1.0 / (p_hom[3] + (real)0.0000001f);
		real p_proj[3] = { p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w };
		transformPoint4x3(p_orig, in->viewmatrix, p_view);
		if (p_view[2] <= (real)0.2f)
		{
			if (in->prefiltered) trapped = 1; /* benign race: every writer stores 1 */
			continue;
		}
		const real *cov3D;
		if (in->cov3D_precomp) cov3D = in->cov3D_precomp + 6 * idx;
		else
		{
			computeCov3D(in->scales + 3 * idx, in->scale_modifier, in->rotations + 4 * idx, o->cov3D + 6 * idx);
			cov3D = o->cov3D + 6 * idx;
		}
		real cov[3];
		computeCov2D(p_orig, focal_x, focal_y, in->tanfovx, in->tanfovy, cov3D, in->viewmatrix, cov);
		real det = cov[0] * cov[2] - cov[1] * cov[1];
		if (det == (real)0) continue;
		real det_inv = (real)1 / det;
		real conic[3] = { cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv };
		real mid = (real)0.5 * (cov[0] + cov[2]);
		real lambda1 = mid + r_sqrt(r_fmax((real)0.1f, mid * mid - det));
		real lambda2 = mid - r_sqrt(r_fmax((real)0.1f, mid * mid - det));
		real my_radius = r_ceil((real)3 * r_sqrt(r_fmax(lambda1, lambda2)));
		real pix[2] = { ndc2Pix(p_proj[0], W), ndc2Pix(p_proj[1], H) };
		int rmin[2], rmax[2];
		getRect(pix[0], pix[1], f2i(my_radius), rmin, rmax, gx, gy);
		uint32_t tnum = (uint32_t)(rmax[1] - rmin[1]) * (uint32_t)(rmax[0] - rmin[0]);
		if (tnum == 0) continue;
		if (with_eigen)
		{
			/* RS forward.cu:244-265 */
			real len1 = 0, len2 = 0, e1[2] = { 0, 0 }, e2[2] = { 0, 0 };
			if (tnum > 1)
			{
				real a1 = cov[0] - lambda1, b1 = cov[1], a2 = cov[0] - lambda2, b2 = cov[1];
				e1[0] = -b1; e1[1] = a1; e2[0] = -b2; e2[1] = a2;
				/* normalize(): rsqrtf in the reference; restated as 1/sqrt */
				real n1 = (real)1 / r_sqrt(e1[0] * e1[0] + e1[1] * e1[1]);
				e1[0] *= n1; e1[1] *= n1;
				real n2 = (real)1 / r_sqrt(e2[0] * e2[0] + e2[1] * e2[1]);
				e2[0] *= n2; e2[1] *= n2;
				len1 = (real)3 * r_sqrt(lambda1);
				len2 = (real)3 * r_sqrt(lambda2);
			}
			o->eigen_len[2 * idx] = len1; o->eigen_len[2 * idx + 1] = len2;
			o->eigen_vec[4 * idx] = e1[0]; o->eigen_vec[4 * idx + 1] = e1[1];
			o->eigen_vec[4 * idx + 2] = e2[0]; o->eigen_vec[4 * idx + 3] = e2[1];
		}
		if (with_sh)
		{
			if (in->colors_precomp == NULL)
			{
				real c[3];
				sh_eval(in->D, in->shs + (size_t)idx * in->M * 3, 0, p_orig, in->campos, c);
				for (int ch = 0; ch < 3; ch++)
				{
					o->clamped[3 * idx + ch] = (c[ch] < 0);
					o->rgb[3 * idx + ch] = r_fmax(c[ch], (real)0);
				}
			}
			else
				for (int ch = 0; ch < 3; ch++) o->rgb[3 * idx + ch] = in->colors_precomp[3 * idx + ch];
		}
		o->depths[idx] = p_view[2];
		o->radii[idx] = f2i(my_radius);
		o->means2D[2 * idx] = pix[0]; o->means2D[2 * idx + 1] = pix[1];
		o->conic[3 * idx] = conic[0]; o->conic[3 * idx + 1] = conic[1]; o->conic[3 * idx + 2] = conic[2];
		o->tiles_rect[idx] = tnum;
		o->tiles_touched[idx] = tnum;
	}
	return trapped ? -1 : 0;
}

/* ---------------- RF tile level map ----------------
 * RF rasterizer_impl.cu:86-177 (levels) and :182-260 (gradients, tile_min, blending flag);
 * ps2level RF auxiliary.h:55-66. Double promotions of the reference are kept. */
static void ncd2dir(real nx, real ny, real rw, real rh, real out[3])
{
	real v[3] = { (nx - (real)0.5f) * rw, (ny - (real)0.5f) * rh, (real)1.0f };
	real d = r_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
	out[0] = v[0] / d; out[1] = v[1] / d; out[2] = v[2] / d;
}
static void tile_levels(const orc_in *in, orc_out *o)
{
	const int W = in->W, H = in->H;
	const int twn = (W + 15) / BLOCK_X, thn = (H + 15) / 16, T = twn * thn;
	const real riw = (real)2.0f, rvd = (real)1.0f, sqrt_max_ps = (real)3.4641016151377544f;
	const real step = (real)(((double)sqrt_max_ps - 1.) / (double)(real)(FOV_NUM - 1));
	for (int idx = 0; idx < T; idx++)
	{
		int ty = idx / twn, tx = idx % twn;
		real px = (real)(tx * BLOCK_X + BLOCK_X / 2), py = (real)(ty * BLOCK_Y + BLOCK_Y / 2);
		real rih = (real)H / (real)W * riw;
		real ncd[2] = { px / W, py / H };
		real tdir[3], gdir[3], cdir[3];
		ncd2dir(ncd[0], ncd[1], riw, rih, tdir);
		ncd2dir(in->gaze_x, in->gaze_y, riw, rih, gdir);
		ncd2dir((real)0.5, (real)0.5, riw, rih, cdir);
		real ecc = r_acos(gdir[0] * tdir[0] + gdir[1] * tdir[1] + gdir[2] * tdir[2]);
		real ecc_c = r_acos(tdir[0] * cdir[0] + tdir[1] * cdir[1] + tdir[2] * cdir[2]);
		real pool = in->alpha * ecc * ecc;
		real amin = (real)((double)ecc_c - (double)pool * 0.5);
		real amax = (real)((double)ecc_c + (double)pool * 0.5);
		real ax = (real)(((double)ncd[0] - 0.5) * (double)riw), ay = (real)(((double)ncd[1] - 0.5) * (double)rih);
		real dist = r_sqrt(ax * ax + ay * ay + rvd * rvd);
		real major = (r_tan(amax) - r_tan(amin)) * rvd;
		real minor = (real)2.0f * dist * r_tan(pool * (real)0.5f);
		real area = (real)(3.14159265358979323846 * (double)major * (double)minor * (double)0.25f);
		real r2p = W / riw;
		real ps = r_sqrt(area) * r2p;
		real level;
		if (ps <= 1) level = 0; else level = (r_sqrt(ps) - 1) / step;
		if ((double)level > ((double)(real)FOV_NUM - 0.1)) level = (real)((double)(real)FOV_NUM - 0.1);
		o->tile_levels[idx] = level;
	}
	for (int idx = 0; idx < T; idx++)
	{
		int ty = idx / twn, tx = idx % twn;
		real lf = o->tile_levels[idx];
		real right = -1, left = -1, up = -1, down = -1;
		if (tx + 1 < twn) right = o->tile_levels[(tx + 1) + twn * ty];
		if (tx - 1 >= 0) left = o->tile_levels[(tx - 1) + twn * ty];
		if (ty + 1 < thn) up = o->tile_levels[tx + twn * (ty + 1)];
		if (ty - 1 >= 0) down = o->tile_levels[tx + twn * (ty - 1)];
		real gxv = 0, gyv = 0;
		if (right != -1 && left != -1) gxv = (right - left) / (real)2.0f;
		else if (right != -1) gxv = right - lf;
		else if (left != -1) gxv = lf - left;
		if (up != -1 && down != -1) gyv = (up - down) / (real)2.0f;
		else if (up != -1) gyv = up - lf;
		else if (down != -1) gyv = lf - down;
		real max_delta = (real)(0.5 * (double)(r_fabs(gxv) + r_fabs(gyv)));
		real tmin = lf - max_delta;
		if (in->variant == ORC_MMFR && tmin < 0) tmin = 0; /* MM rasterizer_impl.cu:249-251 */
		o->tile_min[idx] = tmin;
		real tmin_i = (real)f2i(tmin);
		o->tile_blend[idx] = ((tmin - tmin_i) > (real)0.5f && (tmin_i < (real)(FOV_NUM - 1))) ? 1 : 0;
		o->tile_gy[idx] = gyv;
		o->tile_gx[idx] = gxv;
	}
}

/* ---------------- cull + key emission + sort + ranges ----------------
 * R0 rasterizer_impl.cu:70-138,277-317; RS :70-214,397-469 (OBB_test + duplicateWithKeys
 * with cull bitmap); RF :264-383 (filter), :423-486. The CUB radix sort over
 * (tile<<32 | depth bits) is stable, i.e. ties keep emission order. */
typedef struct { uint32_t tile; uint32_t id; real depth; uint64_t seq; } inst_t;
static int inst_cmp(const void *a, const void *b)
{
	const inst_t *x = (const inst_t *)a, *y = (const inst_t *)b;
	if (x->tile != y->tile) return x->tile < y->tile ? -1 : 1;
	if (x->depth != y->depth) return x->depth < y->depth ? -1 : 1;
	if (x->seq != y->seq) return x->seq < y->seq ? -1 : 1;
	return 0;
}
/* One Gaussian's walk over its tile rectangle with the cull tests of its variant. emit == NULL: the counting pass
 * (tiles_touched, the radii reset, RF level_ranges -- the per-Gaussian outputs of OBB_test / filter); emit != NULL:
 * the duplicateWithKeys pass, instances inside the tile window are written to emit[0..] with sequence numbers seq0..
 * Returns the number of in-window instances. */
/* MM rasterizer_impl.cu:277-304 compute_tile_skips_cuda */
static int mmfr_skips(const orc_in *in, real tile_min)
{
	const real lb = in->cur_level - (real)0.5f, hb = in->cur_level + 1;
	return !(tile_min > lb && tile_min < hb);
}
static int64_t walk_one(const orc_in *in, orc_out *o, int idx, int gx, int gy, int twn, int cull, int fov, inst_t *emit, int64_t seq0)
{
	if (!(o->radii[idx] > 0)) return 0;
	const int mmfr = in->variant == ORC_MMFR; /* MM rasterizer_impl.cu:308-390: keep iff the tile is not skipped (and the OBB hits) */
	int64_t n = 0;
#define ORC_EMIT(X, Y) do { if (in_window(in, (X), (Y))) { if (emit) { emit[n].tile = (uint32_t)((Y) * gx + (X)); emit[n].depth = o->depths[idx]; \
	emit[n].id = (uint32_t)idx; emit[n].seq = (uint64_t)(seq0 + n); } n++; } } while (0)
	int rmin[2], rmax[2];
	real px = o->means2D[2 * idx], py = o->means2D[2 * idx + 1];
	getRect(px, py, o->radii[idx], rmin, rmax, gx, gy);
	uint32_t tnum = (uint32_t)(rmax[1] - rmin[1]) * (uint32_t)(rmax[0] - rmin[0]);
	uint32_t count = 0;
	real hl = (fov && !mmfr) ? in->highest_levels[idx] : 0;
	real lowest = hl, highest = 0;
	int be_blend = 0;
	if (!cull)
	{
		for (int y = rmin[1]; y < rmax[1]; y++)
			for (int x = rmin[0]; x < rmax[0]; x++)
			{
				ORC_EMIT(x, y);
				count++;
			}
	}
	else if (tnum == 1)
	{
		int keep = 1;
		if (fov)
		{
			uint32_t ti = (uint32_t)(rmin[1] * twn + rmin[0]);
			real level = o->tile_min[ti];
			keep = mmfr ? !mmfr_skips(in, level) : level < (hl + 1);
			if (keep) { lowest = level; highest = level; be_blend = o->tile_blend[ti] || be_blend; }
		}
		if (keep)
		{
			ORC_EMIT(rmin[0], rmin[1]);
			count = 1;
		}
	}
	else
	{
		const real *ev = o->eigen_vec + 4 * idx;
		real e1[2] = { ev[0], ev[1] }, e2[2] = { ev[2], ev[3] };
		real len1 = o->eigen_len[2 * idx], len2 = o->eigen_len[2 * idx + 1];
		real c[2] = { px, py };
		real d1x = len1 * e1[0], d1y = len1 * e1[1], d2x = len2 * e2[0], d2y = len2 * e2[1];
		real vtx[4][2] = {
			{ c[0] + d1x + d2x, c[1] + d1y + d2y },
			{ c[0] - d1x + d2x, c[1] - d1y + d2y },
			{ c[0] - d1x - d2x, c[1] - d1y - d2y },
			{ c[0] + d1x - d2x, c[1] + d1y - d2y } };
		for (int y = rmin[1]; y < rmax[1]; y++)
			for (int x = rmin[0]; x < rmax[0]; x++)
			{
				int inside = 1;
				real level = 0; int blending = 0;
				if (fov)
				{
					uint32_t ti = (uint32_t)(y * twn + x);
					blending = o->tile_blend[ti];
					level = o->tile_min[ti];
					inside = mmfr ? !mmfr_skips(in, level) : level < (hl + 1);
				}
				if (inside)
				{
					real tpx = (real)x * (real)BLOCK_X + (real)BLOCK_X / (real)2.0f;
					real tpy = (real)y * (real)BLOCK_Y + (real)BLOCK_Y / (real)2.0f;
					inside = OBB_check(tpx, tpy, vtx, c, e1, e2, len1, len2);
					if (inside)
					{
						count++;
						if (fov)
						{
							lowest = r_fmin(lowest, level); highest = r_fmax(highest, level);
							be_blend = blending || be_blend;
						}
						ORC_EMIT(x, y);
					}
				}
			}
	}
#undef ORC_EMIT
	if (emit) return n;
	o->tiles_touched[idx] = count;
	if (cull && count == 0) o->radii[idx] = 0;
	else if (fov)
	{
		/* RF rasterizer_impl.cu:374-381 */
		o->level_ranges[2 * idx] = f2i(lowest);
		int hi = f2i(highest);
		if (be_blend) hi = imin(hi + 1, FOV_NUM - 1);
		o->level_ranges[2 * idx + 1] = hi;
	}
	return n;
}

static int64_t bin_and_sort(const orc_in *in, orc_out *o)
{
	const int P = in->P, W = in->W, H = in->H;
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y, T = gx * gy;
	const int twn = (W + 15) / BLOCK_X;
	const int cull = in->variant != ORC_R0, fov = ORC_IS_FOV(in->variant);
	int64_t d0 = 0;
	for (int i = 0; i < P; i++) d0 += o->tiles_rect[i];
	o->num_rect = d0;
	/* pass 1: per-Gaussian counts (the reference's OBB_test / filter kernel + InclusiveSum) */
	int64_t *offs = (int64_t *)malloc(sizeof(int64_t) * ((size_t)P + 1));
	ORC_PARALLEL_FOR(dynamic, 1024)
	for (int idx = 0; idx < P; idx++) offs[idx + 1] = walk_one(in, o, idx, gx, gy, twn, cull, fov, NULL, 0);
	offs[0] = 0;
	for (int i = 0; i < P; i++) offs[i + 1] += offs[i];
	const int64_t n = offs[P];
	o->num_rendered = n;
	/* pass 2: duplicateWithKeys; the sequence number is the position the serial reference loop would emit at */
	inst_t *inst = (inst_t *)malloc(sizeof(inst_t) * (size_t)(n > 0 ? n : 1));
	ORC_PARALLEL_FOR(dynamic, 1024)
	for (int idx = 0; idx < P; idx++)
		if (offs[idx + 1] > offs[idx]) walk_one(in, o, idx, gx, gy, twn, cull, fov, inst + offs[idx], offs[idx]);
	free(offs);
	/* stable sort by (tile, depth): buckets by tile (counting sort, keeps emission order), then every bucket by
	 * (depth, sequence) -- the same total order as one sort over (tile, depth, sequence) */
	int64_t *tstart = (int64_t *)calloc((size_t)T + 1, sizeof(int64_t));
	for (int64_t i = 0; i < n; i++) tstart[inst[i].tile + 1]++;
	for (int t = 0; t < T; t++) tstart[t + 1] += tstart[t];
	inst_t *sorted = (inst_t *)malloc(sizeof(inst_t) * (size_t)(n > 0 ? n : 1));
	{
		int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * ((size_t)T + 1));
		memcpy(cur, tstart, sizeof(int64_t) * ((size_t)T + 1));
		for (int64_t i = 0; i < n; i++) sorted[cur[inst[i].tile]++] = inst[i];
		free(cur);
	}
	free(inst);
	inst = sorted;
	ORC_PARALLEL_FOR(dynamic, 8)
	for (int t = 0; t < T; t++)
		if (tstart[t + 1] - tstart[t] > 1) qsort(inst + tstart[t], (size_t)(tstart[t + 1] - tstart[t]), sizeof(inst_t), inst_cmp);
	free(tstart);
	/* identifyTileRanges: R0 rasterizer_impl.cu:116-138 (+ memset :310) */
	memset(o->ranges, 0, sizeof(uint32_t) * 2 * (size_t)T);
	for (int64_t i = 0; i < n; i++)
	{
		uint32_t cur = inst[i].tile;
		if (i == 0) o->ranges[2 * cur] = 0;
		else
		{
			uint32_t prev = inst[i - 1].tile;
			if (cur != prev) { o->ranges[2 * prev + 1] = (uint32_t)i; o->ranges[2 * cur] = (uint32_t)i; }
		}
		if (i == n - 1) o->ranges[2 * cur + 1] = (uint32_t)n;
	}
	if (n <= o->capacity)
	{
		for (int64_t i = 0; i < n; i++)
		{
			o->point_list[i] = inst[i].id;
#ifndef ORC_DOUBLE
			if (o->keys)
			{
				uint32_t bits; float d = inst[i].depth; memcpy(&bits, &d, 4);
				o->keys[i] = ((uint64_t)inst[i].tile << 32) | bits;
			}
#endif
		}
	}
	free(inst);
	return n;
}

/* RF rasterizer_impl.cu:490-530 */
static void compute_fov_colors(const orc_in *in, orc_out *o)
{
	ORC_PARALLEL_FOR(static)
	for (int idx = 0; idx < in->P; idx++)
	{
		if (!(o->radii[idx] > 0)) continue;
		real rest[3];
		sh_eval(in->D, in->shs + (size_t)idx * in->M * 3, 1, in->means3D + 3 * idx, in->campos, rest);
		for (int l = o->level_ranges[2 * idx]; l <= o->level_ranges[2 * idx + 1]; l++)
			for (int ch = 0; ch < 3; ch++)
			{
				real dc = in->shs_dcs[(size_t)idx * 3 * FOV_NUM + l * 3 + ch];
				real c = SH_C0 * dc + rest[ch];
				o->fov_colors[((size_t)idx * FOV_NUM + l) * 3 + ch] = r_fmax(c, (real)0);
			}
	}
}

/* RP rasterizer_impl.cu:118-137: colour evaluated only for Gaussians surviving the cull;
 * numerically the same SH polynomial as R0 forward.cu:20-71. Our preprocess() already
 * evaluated it for every rect-passing Gaussian, which is a superset; nothing to do. */

/* ---------------- blend: R0 / RS / RP ----------------
 * R0 forward.cu:267-384; RS forward.cu:298-430 (power<-4.5 skip, gaussians_count,
 * contributions); RP forward.cu:243-384 (w = alpha*T product order, no final_T/n_contrib). */
static void render_plain(const orc_in *in, orc_out *o)
{
	const int W = in->W, H = in->H, variant = in->variant;
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	const int cutoff = variant != ORC_R0;
	double *contrib = NULL;
	const int stats = (variant == ORC_RS || variant == ORC_RMAX || variant == ORC_LWMC);
	if (stats)
	{
		contrib = (double *)calloc((size_t)in->P, sizeof(double));
		memset(o->gaussians_count, 0, sizeof(int32_t) * (size_t)in->P);
	}
	/* the running-maximum statistics of RMAX / LWMC are not associative under atomics: those two stay serial */
	const int nthreads = (variant == ORC_RMAX || variant == ORC_LWMC) ? 1 : g_threads;
	(void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
	for (int tile = 0; tile < gx * gy; tile++)
		{
			const int ty = tile / gx, tx = tile % gx;
			if (!in_window(in, tx, ty)) continue;
			const uint32_t r0 = o->ranges[2 * (ty * gx + tx)], r1 = o->ranges[2 * (ty * gx + tx) + 1];
			const int n = (int)(r1 - r0);
			int all_done_pos = 0, never_done = 0;
			for (int ly = 0; ly < BLOCK_Y; ly++)
				for (int lx = 0; lx < BLOCK_X; lx++)
				{
					const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
					if (!(pxi < W && pyi < H)) continue; /* done from the start: position 0 */
					const real pixf[2] = { (real)pxi, (real)pyi };
					real T = 1, C[3] = { 0, 0, 0 };
					uint32_t contributor = 0, last_contributor = 0;
					int done_pos = -1;
					int max_point_idx = 0;          /* LWMC forward.cu:347-348 (quirk: defaults to Gaussian 0) */
					real max_point_contrib = 0;
					for (int j = 0; j < n; j++)
					{
						contributor++;
						const uint32_t g = o->point_list[r0 + j];
						const real dx = o->means2D[2 * g] - pixf[0], dy = o->means2D[2 * g + 1] - pixf[1];
						const real ca = o->conic[3 * g], cb = o->conic[3 * g + 1], cc = o->conic[3 * g + 2];
						const real power = (real)-0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
						if (power > (real)0) continue;
						if (cutoff && power < (real)-4.5f) continue;
						if (variant == ORC_RMAX) o->gaussians_count[g] += 1; /* …_max forward.cu:381: per pixel in support */
						const real alpha = r_fmin((real)0.99f, in->opacities[g] * r_exp(power));
						if (alpha < (real)1.0f / (real)255.0f) continue;
						const real test_T = T * (1 - alpha);
						if (test_T < (real)0.0001f) { done_pos = j + 1; break; }
						if (variant == ORC_RP)
						{
							const real w = alpha * T;
							for (int ch = 0; ch < 3; ch++) C[ch] += o->rgb[3 * g + ch] * w;
						}
						else
							for (int ch = 0; ch < 3; ch++) C[ch] += o->rgb[3 * g + ch] * alpha * T;
						if (variant == ORC_RS) { ORC_ATOMIC contrib[g] += (double)(alpha * T); }
						else if (variant == ORC_RMAX) { if ((double)(alpha * T) > contrib[g]) contrib[g] = (double)(alpha * T); } /* atomicMaxFloat, …_max forward.cu:400 */
						else if (variant == ORC_LWMC) { const real cv = alpha * T; if (cv > max_point_contrib) { max_point_contrib = cv; max_point_idx = (int)g; } }
						T = test_T;
						last_contributor = contributor;
					}
					if (done_pos < 0) never_done = 1; else if (done_pos > all_done_pos) all_done_pos = done_pos;
					const size_t pid = (size_t)W * pyi + pxi;
					if (variant != ORC_RP)
					{
						o->final_T[pid] = T;
						o->n_contrib[pid] = last_contributor;
					}
					if (variant == ORC_LWMC) contrib[max_point_idx] += (double)in->loss_map[pid]; /* …_count forward.cu:435 */
					for (int ch = 0; ch < 3; ch++) o->color[(size_t)ch * H * W + pid] = C[ch] + T * in->bg[ch];
				}
			if (variant == ORC_RS || variant == ORC_LWMC)
			{
				/* RS forward.cu:349-361: +1 per list entry fetched by a batch that a still-live
				 * tile loads; the vote happens once per 256-entry batch. */
				int rounds = (n + BLOCK_SIZE - 1) / BLOCK_SIZE;
				int executed = never_done ? rounds : imin(rounds, (all_done_pos + BLOCK_SIZE - 1) / BLOCK_SIZE);
				int fetched = imin(n, executed * BLOCK_SIZE);
				for (int j = 0; j < fetched; j++) { ORC_ATOMIC o->gaussians_count[o->point_list[r0 + j]] += 1; }
			}
		}
	if (contrib)
	{
		for (int i = 0; i < in->P; i++) o->contributions[i] = (real)contrib[i];
		free(contrib);
	}
}

/* ---------------- blend: RF ----------------
 * single-level tiles RF forward.cu:490-609; two-level tiles RF forward.cu:262-476. */
static void render_fov(const orc_in *in, orc_out *o)
{
	const int W = in->W, H = in->H;
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	const int twn = (W + 15) / BLOCK_X;
	const real start_blend = (real)0.5f, blend_width = (real)0.5f;
	ORC_PARALLEL_FOR(dynamic, 4)
	for (int tile = 0; tile < gx * gy; tile++)
		{
			const int ty = tile / gx, tx = tile % gx;
			if (!in_window(in, tx, ty)) continue;
			const uint32_t r0 = o->ranges[2 * (ty * gx + tx)], r1 = o->ranges[2 * (ty * gx + tx) + 1];
			const int n = (int)(r1 - r0);
			const uint32_t cur = (uint32_t)(tx + twn * ty);
			const int blending = o->tile_blend[cur];
			const real tlf = o->tile_min[cur];
			const int tli = f2i(tlf);
			for (int ly = 0; ly < BLOCK_Y; ly++)
				for (int lx = 0; lx < BLOCK_X; lx++)
				{
					const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
					if (!(pxi < W && pyi < H)) continue;
					const real pixf[2] = { (real)pxi, (real)pyi };
					const size_t pid = (size_t)W * pyi + pxi;
					if (!blending)
					{
						real T1 = 1, C1[3] = { 0, 0, 0 };
						for (int j = 0; j < n; j++)
						{
							const uint32_t g = o->point_list[r0 + j];
							const real dx = o->means2D[2 * g] - pixf[0], dy = o->means2D[2 * g + 1] - pixf[1];
							const real ca = o->conic[3 * g], cb = o->conic[3 * g + 1], cc = o->conic[3 * g + 2];
							const real power = (real)-0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
							if (power > (real)0 || power < (real)-4.5f) continue;
							const real alpha = r_fmin((real)0.99f, in->opacities[(size_t)g * FOV_NUM + tli] * r_exp(power));
							if (alpha < (real)1.0f / (real)255.0f) continue;
							const real test_T = T1 * (1 - alpha);
							if (test_T < (real)0.0001f) break;
							const real w = alpha * T1;
							const real *f = o->fov_colors + ((size_t)g * FOV_NUM + tli) * 3;
							C1[0] += f[0] * w; C1[1] += f[1] * w; C1[2] += f[2] * w;
							T1 = test_T;
						}
						for (int ch = 0; ch < 3; ch++) o->color[(size_t)ch * H * W + pid] = C1[ch] + in->bg[ch] * T1;
					}
					else
					{
						real T1 = 1, T2 = 1, C1[3] = { 0, 0, 0 }, C2[3] = { 0, 0, 0 };
						const real dxl = (real)lx, dyl = (real)ly;
						const real est = tlf + (dxl * o->tile_gx[cur] + dyl * o->tile_gy[cur]) / (real)BLOCK_X;
						const int L1 = tli, L2 = tli + 1;
						const real L2f = tlf + (real)1.0f;
						int L1_done = est > (real)L2, L2_done = 0;
						for (int j = 0; j < n; j++)
						{
							const uint32_t g = o->point_list[r0 + j];
							const real dx = o->means2D[2 * g] - pixf[0], dy = o->means2D[2 * g + 1] - pixf[1];
							const real ca = o->conic[3 * g], cb = o->conic[3 * g + 1], cc = o->conic[3 * g + 2];
							const real power = (real)-0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
							if (power > (real)0 || power < (real)-4.5f) continue;
							const real ev = r_exp(power);
							if (!L1_done)
							{
								const real a1 = r_fmin((real)0.99f, in->opacities[(size_t)g * FOV_NUM + L1] * ev);
								if (!(a1 < (real)1.0f / (real)255.0f))
								{
									const real tT = T1 * (1 - a1);
									L1_done = tT < (real)0.0001f;
									if (!L1_done)
									{
										const real w = a1 * T1;
										const real *f = o->fov_colors + ((size_t)g * FOV_NUM + L1) * 3;
										C1[0] += f[0] * w; C1[1] += f[1] * w; C1[2] += f[2] * w;
										T1 = tT;
									}
								}
							}
							if (!L2_done)
							{
								const real a2 = r_fmin((real)0.99f, in->opacities[(size_t)g * FOV_NUM + L2] * ev);
								const int skip2 = (a2 < (real)1.0f / (real)255.0f) || ((in->highest_levels[g] + 1) < L2f);
								if (!skip2)
								{
									const real tT = T2 * (1 - a2);
									L2_done = tT < (real)0.0001f;
									if (!L2_done)
									{
										const real w = a2 * T2;
										const real *f = o->fov_colors + ((size_t)g * FOV_NUM + L2) * 3;
										C2[0] += f[0] * w; C2[1] += f[1] * w; C2[2] += f[2] * w;
										T2 = tT;
									}
								}
							}
							if (L1_done && L2_done) break;
						}
						for (int ch = 0; ch < 3; ch++) { C1[ch] = C1[ch] + in->bg[ch] * T1; C2[ch] = C2[ch] + in->bg[ch] * T2; }
						real x = r_fabs(est - ((real)L1 + start_blend)) / blend_width;
						x = r_fmax((real)0, r_fmin((real)1, x));
						const real bT = 3 * x * x - 2 * x * x * x;
						const real w1 = 1 - bT;
						for (int ch = 0; ch < 3; ch++)
							o->color[(size_t)ch * H * W + pid] = C1[ch] * w1 + C2[ch] * ((real)1 - w1);
					}
				}
		}
}

/* ---------------- blend: SMFR (shared-model foveated baseline) ----------------
 * single-level tiles NV forward.cu:482-580; two-level tiles NV forward.cu:258-480. One colour and opacity per
 * Gaussian; in two-level tiles an OPEN L1 state skips a Gaussian with alpha < 1/255 for both states (the `continue`
 * at :403-405), a finished L1 lets L2 take any alpha (no test at :409-425). */
static void render_smfr(const orc_in *in, orc_out *o)
{
	const int W = in->W, H = in->H;
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	const int twn = (W + 15) / BLOCK_X;
	const real start_blend = (real)0.5f, blend_width = (real)0.5f;
	ORC_PARALLEL_FOR(dynamic, 4)
	for (int tile = 0; tile < gx * gy; tile++)
		{
			const int ty = tile / gx, tx = tile % gx;
			if (!in_window(in, tx, ty)) continue;
			const uint32_t r0 = o->ranges[2 * (ty * gx + tx)], r1 = o->ranges[2 * (ty * gx + tx) + 1];
			const int n = (int)(r1 - r0);
			const uint32_t cur = (uint32_t)(tx + twn * ty);
			const int blending = o->tile_blend[cur];
			const real tlf = o->tile_min[cur];
			const int tli = f2i(tlf);
			for (int ly = 0; ly < BLOCK_Y; ly++)
				for (int lx = 0; lx < BLOCK_X; lx++)
				{
					const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
					if (!(pxi < W && pyi < H)) continue;
					const real pixf[2] = { (real)pxi, (real)pyi };
					const size_t pid = (size_t)W * pyi + pxi;
					real T1 = 1, T2 = 1, C1[3] = { 0, 0, 0 }, C2[3] = { 0, 0, 0 };
					const real est = tlf + ((real)lx * o->tile_gx[cur] + (real)ly * o->tile_gy[cur]) / (real)BLOCK_X;
					const int L1 = tli, L2 = tli + 1;
					const real L2f = tlf + (real)1.0f;
					int L1_done = blending ? (est > (real)L2) : 0, L2_done = blending ? 0 : 1;
					for (int j = 0; j < n; j++)
					{
						const uint32_t g = o->point_list[r0 + j];
						const real dx = o->means2D[2 * g] - pixf[0], dy = o->means2D[2 * g + 1] - pixf[1];
						const real ca = o->conic[3 * g], cb = o->conic[3 * g + 1], cc = o->conic[3 * g + 2];
						const real power = (real)-0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
						if (power > (real)0 || power < (real)-4.5f) continue;
						const real alpha = r_fmin((real)0.99f, in->opacities[g] * r_exp(power));
						const int askip = alpha < (real)1.0f / (real)255.0f;
						if (!blending)
						{
							if (askip) continue;
							const real tT = T1 * (1 - alpha);
							if (tT < (real)0.0001f) break;
							const real w = alpha * T1;
							for (int ch = 0; ch < 3; ch++) C1[ch] += o->rgb[3 * g + ch] * w;
							T1 = tT;
							continue;
						}
						if (!L1_done)
						{
							if (askip) continue;
							const real tT = T1 * (1 - alpha);
							L1_done = tT < (real)0.0001f;
							if (!L1_done)
							{
								const real w = alpha * T1;
								for (int ch = 0; ch < 3; ch++) C1[ch] += o->rgb[3 * g + ch] * w;
								T1 = tT;
							}
						}
						if (!L2_done && !((in->highest_levels[g] + 1) < L2f))
						{
							const real tT = T2 * (1 - alpha);
							L2_done = tT < (real)0.0001f;
							if (!L2_done)
							{
								const real w = alpha * T2;
								for (int ch = 0; ch < 3; ch++) C2[ch] += o->rgb[3 * g + ch] * w;
								T2 = tT;
							}
						}
						if (L1_done && L2_done) break;
					}
					if (!blending)
					{
						for (int ch = 0; ch < 3; ch++) o->color[(size_t)ch * H * W + pid] = C1[ch] + in->bg[ch] * T1;
						continue;
					}
					for (int ch = 0; ch < 3; ch++) { C1[ch] = C1[ch] + in->bg[ch] * T1; C2[ch] = C2[ch] + in->bg[ch] * T2; }
					real x = r_fabs(est - ((real)L1 + start_blend)) / blend_width;
					x = r_fmax((real)0, r_fmin((real)1, x));
					const real bT = 3 * x * x - 2 * x * x * x;
					const real w1 = 1 - bT;
					for (int ch = 0; ch < 3; ch++) o->color[(size_t)ch * H * W + pid] = C1[ch] * w1 + C2[ch] * ((real)1 - w1);
				}
		}
}

/* ---------------- blend: MMFR, one level ----------------
 * MM forward.cu:255-420 (two-level tiles) and :422-540 (single-level tiles); skipped tiles keep the zero the image
 * starts with (MM rasterize_points.cu:79). */
static void render_mmfr(const orc_in *in, orc_out *o)
{
	const int W = in->W, H = in->H;
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	const int twn = (W + 15) / BLOCK_X;
	const real start_blend = (real)0.5f, blend_width = (real)0.5f;
	ORC_PARALLEL_FOR(dynamic, 4)
	for (int tile = 0; tile < gx * gy; tile++)
		{
			const int ty = tile / gx, tx = tile % gx;
			if (!in_window(in, tx, ty)) continue;
			const uint32_t r0 = o->ranges[2 * (ty * gx + tx)], r1 = o->ranges[2 * (ty * gx + tx) + 1];
			const int n = (int)(r1 - r0);
			const uint32_t cur = (uint32_t)(tx + twn * ty);
			const int blending = o->tile_blend[cur];
			const real tlf = o->tile_min[cur];
			const int skipped = mmfr_skips(in, tlf);
			for (int ly = 0; ly < BLOCK_Y; ly++)
				for (int lx = 0; lx < BLOCK_X; lx++)
				{
					const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
					if (!(pxi < W && pyi < H)) continue;
					const size_t pid = (size_t)W * pyi + pxi;
					if (skipped) { for (int ch = 0; ch < 3; ch++) o->color[(size_t)ch * H * W + pid] = 0; continue; }
					const real pixf[2] = { (real)pxi, (real)pyi };
					real T1 = 1, C1[3] = { 0, 0, 0 };
					int done = 0, L1 = 0;
					real x = 0;
					if (blending)
					{
						const real est = tlf + ((real)lx * o->tile_gx[cur] + (real)ly * o->tile_gy[cur]) / (real)BLOCK_X;
						L1 = f2i(est);
						x = (est - ((real)L1 + start_blend)) / blend_width;
						if (x < 0 && (real)L1 != in->cur_level) done = 1;
					}
					for (int j = 0; j < n && !done; j++)
					{
						const uint32_t g = o->point_list[r0 + j];
						const real dx = o->means2D[2 * g] - pixf[0], dy = o->means2D[2 * g + 1] - pixf[1];
						const real ca = o->conic[3 * g], cb = o->conic[3 * g + 1], cc = o->conic[3 * g + 2];
						const real power = (real)-0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
						if (power > (real)0 || power < (real)-4.5f) continue;
						const real alpha = r_fmin((real)0.99f, in->opacities[g] * r_exp(power));
						if (alpha < (real)1.0f / (real)255.0f) continue;
						const real tT = T1 * (1 - alpha);
						if (tT < (real)0.0001f) break;
						const real w = alpha * T1;
						for (int ch = 0; ch < 3; ch++) C1[ch] += o->rgb[3 * g + ch] * w;
						T1 = tT;
					}
					real used = 1;
					if (blending)
					{
						x = r_fmax((real)0, r_fmin((real)1, x));
						const real bT = 3 * x * x - 2 * x * x * x;
						const real w1 = 1 - bT;
						used = ((real)L1 == in->cur_level) ? w1 : (real)(1.0 - (double)w1);
					}
					for (int ch = 0; ch < 3; ch++)
					{
						const real c = C1[ch] + in->bg[ch] * T1;
						o->color[(size_t)ch * H * W + pid] = blending ? c * used : c;
					}
				}
		}
}

/* ---------------- public: forward ---------------- */
int64_t orc_forward(const orc_in *in, orc_out *o)
{
	if (in->P == 0) { o->num_rect = 0; o->num_rendered = 0; return 0; }
	if (preprocess(in, o) != 0) return -1;
	if (ORC_IS_FOV(in->variant)) tile_levels(in, o);
	int64_t n = bin_and_sort(in, o);
	if (n > o->capacity) return n; /* caller must retry with capacity >= n */
	if (in->variant == ORC_RF) { compute_fov_colors(in, o); render_fov(in, o); }
	else if (in->variant == ORC_SMFR) render_smfr(in, o); /* NV rasterizer_impl.cu:463-483: the colour is the plain SH colour */
	else if (in->variant == ORC_MMFR) render_mmfr(in, o);
	else render_plain(in, o);
	return n;
}

/* RF tile level map alone (for pinning against odak's pooling-size map) */
void orc_tile_levels(const orc_in *in, orc_out *o) { tile_levels(in, o); }

/* SH colour alone (for pinning against utils/sh_utils.eval_sh): out[P,3] unclamped, +0.5 */
void orc_sh_colors(const orc_in *in, int rest, real *out)
{
	for (int i = 0; i < in->P; i++)
		sh_eval(in->D, in->shs + (size_t)i * in->M * 3, rest, in->means3D + 3 * i, in->campos, out + 3 * i);
}

/* R0 rasterizer_impl.cu:54-66 (checkFrustum / mark_visible) */
void orc_mark_visible(const orc_in *in, uint8_t *present)
{
	ORC_PARALLEL_FOR(static)
	for (int i = 0; i < in->P; i++)
	{
		real pv[3];
		transformPoint4x3(in->means3D + 3 * i, in->viewmatrix, pv);
		present[i] = !(pv[2] <= (real)0.2f);
	}
}

/* ---------------- backward ----------------
 * render: R0 backward.cu:399-557 (RS differs only at :495, the power<-4.5 skip).
 * Float atomics of the reference have no defined order; the oracle accumulates every
 * per-Gaussian sum in double and rounds once. */
static void backward_render(const orc_in *in, const orc_out *o, const real *dL_dpix,
	double *d_mean2D /*[P,2]*/, double *d_conic /*[P,3] x,y,w*/, double *d_opacity, double *d_color /*[P,3]*/)
{
	const int W = in->W, H = in->H;
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	const int cutoff = in->variant != ORC_R0;
	const real ddelx_dx = (real)(0.5 * W), ddely_dy = (real)(0.5 * H);
	ORC_PARALLEL_FOR(dynamic, 4)
	for (int tile = 0; tile < gx * gy; tile++)
		{
			const int ty = tile / gx, tx = tile % gx;
			const uint32_t r0 = o->ranges[2 * (ty * gx + tx)], r1 = o->ranges[2 * (ty * gx + tx) + 1];
			const int n = (int)(r1 - r0);
			for (int ly = 0; ly < BLOCK_Y; ly++)
				for (int lx = 0; lx < BLOCK_X; lx++)
				{
					const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
					if (!(pxi < W && pyi < H)) continue;
					const size_t pid = (size_t)W * pyi + pxi;
					const real pixf[2] = { (real)pxi, (real)pyi };
					const real T_final = o->final_T[pid];
					real T = T_final;
					uint32_t contributor = (uint32_t)n;
					const int last_contributor = (int)o->n_contrib[pid];
					real accum_rec[3] = { 0, 0, 0 }, dL_dpixel[3], last_alpha = 0, last_color[3] = { 0, 0, 0 };
					for (int ch = 0; ch < 3; ch++) dL_dpixel[ch] = dL_dpix[(size_t)ch * H * W + pid];
					for (int j = 0; j < n; j++)
					{
						contributor--;
						if ((int)contributor >= last_contributor) continue;
						const uint32_t g = o->point_list[r1 - 1 - j];
						const real dx = o->means2D[2 * g] - pixf[0], dy = o->means2D[2 * g + 1] - pixf[1];
						const real ca = o->conic[3 * g], cb = o->conic[3 * g + 1], cc = o->conic[3 * g + 2];
						const real op = in->opacities[g];
						const real power = (real)-0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
						if (power > (real)0) continue;
						if (cutoff && power < (real)-4.5f) continue;
						const real G = r_exp(power);
						const real alpha = r_fmin((real)0.99f, op * G);
						if (alpha < (real)1.0f / (real)255.0f) continue;
						T = T / ((real)1 - alpha);
						const real dchannel_dcolor = alpha * T;
						real dL_dalpha = 0;
						for (int ch = 0; ch < 3; ch++)
						{
							const real c = o->rgb[3 * g + ch];
							accum_rec[ch] = last_alpha * last_color[ch] + ((real)1 - last_alpha) * accum_rec[ch];
							last_color[ch] = c;
							const real dL_dchannel = dL_dpixel[ch];
							dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
							{ ORC_ATOMIC d_color[3 * g + ch] += (double)(dchannel_dcolor * dL_dchannel); }
						}
						dL_dalpha *= T;
						last_alpha = alpha;
						real bg_dot = 0;
						for (int ch = 0; ch < 3; ch++) bg_dot += in->bg[ch] * dL_dpixel[ch];
						dL_dalpha += (-T_final / ((real)1 - alpha)) * bg_dot;
						const real dL_dG = op * dL_dalpha;
						const real gdx = G * dx, gdy = G * dy;
						const real dG_ddelx = -gdx * ca - gdy * cb;
						const real dG_ddely = -gdy * cc - gdx * cb;
						{ ORC_ATOMIC d_mean2D[2 * g] += (double)(dL_dG * dG_ddelx * ddelx_dx); }
						{ ORC_ATOMIC d_mean2D[2 * g + 1] += (double)(dL_dG * dG_ddely * ddely_dy); }
						{ ORC_ATOMIC d_conic[3 * g] += (double)((real)-0.5f * gdx * dx * dL_dG); }
						{ ORC_ATOMIC d_conic[3 * g + 1] += (double)((real)-0.5f * gdx * dy * dL_dG); }
						{ ORC_ATOMIC d_conic[3 * g + 2] += (double)((real)-0.5f * gdy * dy * dL_dG); }
						{ ORC_ATOMIC d_opacity[g] += (double)(G * dL_dalpha); }
					}
				}
		}
}

/* R0 backward.cu:144-274 (computeCov2DCUDA) */
static void backward_cov2D(const orc_in *in, int idx, const real *cov3D, real fx, real fy,
	const real dconic[3] /* x,y,w */, real dmean[3], real dcov[6])
{
	const real *vm = in->viewmatrix;
	real t[3];
	transformPoint4x3(in->means3D + 3 * idx, vm, t);
	const real limx = (real)1.3f * in->tanfovx, limy = (real)1.3f * in->tanfovy;
	const real txtz = t[0] / t[2], tytz = t[1] / t[2];
	t[0] = r_fmin(limx, r_fmax(-limx, txtz)) * t[2];
	t[1] = r_fmin(limy, r_fmax(-limy, tytz)) * t[2];
	const real x_grad_mul = (txtz < -limx || txtz > limx) ? 0 : 1;
	const real y_grad_mul = (tytz < -limy || tytz > limy) ? 0 : 1;
	m3 J = m3_cols(fx / t[2], 0, -(fx * t[0]) / (t[2] * t[2]),
		0, fy / t[2], -(fy * t[1]) / (t[2] * t[2]), 0, 0, 0);
	m3 Wm = m3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
	m3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
	m3 T = m3_mul(Wm, J);
	m3 c2 = m3_mul(m3_mul(m3_t(T), m3_t(Vrk)), T);
	real a = (c2.c[0][0] += (real)0.3f);
	real b = c2.c[0][1];
	real c = (c2.c[1][1] += (real)0.3f);
	real denom = a * c - b * b;
	real dL_da = 0, dL_db = 0, dL_dc = 0;
	real denom2inv = (real)1.0 / ((denom * denom) + (real)0.0000001f);
#define TT(i, j) T.c[i][j]
#define VV(i, j) Vrk.c[i][j]
	if (denom2inv != 0)
	{
		dL_da = denom2inv * (-c * c * dconic[0] + 2 * b * c * dconic[1] + (denom - a * c) * dconic[2]);
		dL_dc = denom2inv * (-a * a * dconic[2] + 2 * a * b * dconic[1] + (denom - a * c) * dconic[0]);
		dL_db = denom2inv * 2 * (b * c * dconic[0] - (denom + 2 * b * b) * dconic[1] + a * b * dconic[2]);
		dcov[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
		dcov[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
		dcov[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
		dcov[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
		dcov[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
		dcov[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
	}
	else
		for (int i = 0; i < 6; i++) dcov[i] = 0;
	real dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
		(TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
	real dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
		(TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
	real dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
		(TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
	real dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
		(TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
	real dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
		(TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
	real dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
		(TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
#undef TT
#undef VV
	real dL_dJ00 = Wm.c[0][0] * dL_dT00 + Wm.c[0][1] * dL_dT01 + Wm.c[0][2] * dL_dT02;
	real dL_dJ02 = Wm.c[2][0] * dL_dT00 + Wm.c[2][1] * dL_dT01 + Wm.c[2][2] * dL_dT02;
	real dL_dJ11 = Wm.c[1][0] * dL_dT10 + Wm.c[1][1] * dL_dT11 + Wm.c[1][2] * dL_dT12;
	real dL_dJ12 = Wm.c[2][0] * dL_dT10 + Wm.c[2][1] * dL_dT11 + Wm.c[2][2] * dL_dT12;
	real tz = (real)1 / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
	real dL_dtx = x_grad_mul * -fx * tz2 * dL_dJ02;
	real dL_dty = y_grad_mul * -fy * tz2 * dL_dJ12;
	real dL_dtz = -fx * tz2 * dL_dJ00 - fy * tz2 * dL_dJ11 + (2 * fx * t[0]) * tz3 * dL_dJ02 + (2 * fy * t[1]) * tz3 * dL_dJ12;
	/* transformVec4x3Transpose: R0 auxiliary.h:89-97 */
	dmean[0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
	dmean[1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
	dmean[2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;
}

/* R0 backward.cu:20-139 (SH backward) */
static void backward_sh(const orc_in *in, const orc_out *o, int idx, const real dL_dcolor[3], real dmean_add[3], real *dL_dsh)
{
	const real *pos = in->means3D + 3 * idx;
	real dir_orig[3] = { pos[0] - in->campos[0], pos[1] - in->campos[1], pos[2] - in->campos[2] };
	real len = r_sqrt(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
	real x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
	const real *sh = in->shs + (size_t)idx * in->M * 3;
	const int deg = in->D;
	real dRGB[3];
	for (int ch = 0; ch < 3; ch++) dRGB[ch] = dL_dcolor[ch] * (o->clamped[3 * idx + ch] ? 0 : 1);
	real dRGBdx[3] = { 0, 0, 0 }, dRGBdy[3] = { 0, 0, 0 }, dRGBdz[3] = { 0, 0, 0 };
#define SHV(k, ch) sh[3 * (k) + (ch)]
#define DSH(k, w) for (int ch = 0; ch < 3; ch++) dL_dsh[3 * (k) + ch] = (w) * dRGB[ch]
	DSH(0, SH_C0);
	if (deg > 0)
	{
		DSH(1, -SH_C1 * y); DSH(2, SH_C1 * z); DSH(3, -SH_C1 * x);
		for (int ch = 0; ch < 3; ch++)
		{
			dRGBdx[ch] = -SH_C1 * SHV(3, ch);
			dRGBdy[ch] = -SH_C1 * SHV(1, ch);
			dRGBdz[ch] = SH_C1 * SHV(2, ch);
		}
		if (deg > 1)
		{
			real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
			DSH(4, SH_C2[0] * xy); DSH(5, SH_C2[1] * yz); DSH(6, SH_C2[2] * ((real)2 * zz - xx - yy));
			DSH(7, SH_C2[3] * xz); DSH(8, SH_C2[4] * (xx - yy));
			for (int ch = 0; ch < 3; ch++)
			{
				dRGBdx[ch] += SH_C2[0] * y * SHV(4, ch) + SH_C2[2] * (real)2 * -x * SHV(6, ch) + SH_C2[3] * z * SHV(7, ch) + SH_C2[4] * (real)2 * x * SHV(8, ch);
				dRGBdy[ch] += SH_C2[0] * x * SHV(4, ch) + SH_C2[1] * z * SHV(5, ch) + SH_C2[2] * (real)2 * -y * SHV(6, ch) + SH_C2[4] * (real)2 * -y * SHV(8, ch);
				dRGBdz[ch] += SH_C2[1] * y * SHV(5, ch) + SH_C2[2] * (real)2 * (real)2 * z * SHV(6, ch) + SH_C2[3] * x * SHV(7, ch);
			}
			if (deg > 2)
			{
				DSH(9, SH_C3[0] * y * ((real)3 * xx - yy));
				DSH(10, SH_C3[1] * xy * z);
				DSH(11, SH_C3[2] * y * ((real)4 * zz - xx - yy));
				DSH(12, SH_C3[3] * z * ((real)2 * zz - (real)3 * xx - (real)3 * yy));
				DSH(13, SH_C3[4] * x * ((real)4 * zz - xx - yy));
				DSH(14, SH_C3[5] * z * (xx - yy));
				DSH(15, SH_C3[6] * x * (xx - (real)3 * yy));
				for (int ch = 0; ch < 3; ch++)
				{
					dRGBdx[ch] += (
						SH_C3[0] * SHV(9, ch) * (real)3 * (real)2 * xy +
						SH_C3[1] * SHV(10, ch) * yz +
						SH_C3[2] * SHV(11, ch) * (real)-2 * xy +
						SH_C3[3] * SHV(12, ch) * (real)-3 * (real)2 * xz +
						SH_C3[4] * SHV(13, ch) * ((real)-3 * xx + (real)4 * zz - yy) +
						SH_C3[5] * SHV(14, ch) * (real)2 * xz +
						SH_C3[6] * SHV(15, ch) * (real)3 * (xx - yy));
					dRGBdy[ch] += (
						SH_C3[0] * SHV(9, ch) * (real)3 * (xx - yy) +
						SH_C3[1] * SHV(10, ch) * xz +
						SH_C3[2] * SHV(11, ch) * ((real)-3 * yy + (real)4 * zz - xx) +
						SH_C3[3] * SHV(12, ch) * (real)-3 * (real)2 * yz +
						SH_C3[4] * SHV(13, ch) * (real)-2 * xy +
						SH_C3[5] * SHV(14, ch) * (real)-2 * yz +
						SH_C3[6] * SHV(15, ch) * (real)-3 * (real)2 * xy);
					dRGBdz[ch] += (
						SH_C3[1] * SHV(10, ch) * xy +
						SH_C3[2] * SHV(11, ch) * (real)4 * (real)2 * yz +
						SH_C3[3] * SHV(12, ch) * (real)3 * ((real)2 * zz - xx - yy) +
						SH_C3[4] * SHV(13, ch) * (real)4 * (real)2 * xz +
						SH_C3[5] * SHV(14, ch) * (xx - yy));
				}
			}
		}
	}
#undef SHV
#undef DSH
	real dL_ddir[3] = {
		dRGBdx[0] * dRGB[0] + dRGBdx[1] * dRGB[1] + dRGBdx[2] * dRGB[2],
		dRGBdy[0] * dRGB[0] + dRGBdy[1] * dRGB[1] + dRGBdy[2] * dRGB[2],
		dRGBdz[0] * dRGB[0] + dRGBdz[1] * dRGB[1] + dRGBdz[2] * dRGB[2] };
	/* dnormvdv: R0 auxiliary.h:107-117 */
	const real *v = dir_orig, *dv = dL_ddir;
	real sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
	real invsum32 = (real)1.0 / r_sqrt(sum2 * sum2 * sum2);
	dmean_add[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
	dmean_add[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
	dmean_add[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

/* R0 backward.cu:278-341 (computeCov3D backward) */
static void backward_cov3D(const real scale[3], real mod, const real rot[4], const real dcov[6], real dscale[3], real drot[4])
{
	real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
	m3 R = m3_cols(
		(real)1 - (real)2 * (y * y + z * z), (real)2 * (x * y - r * z), (real)2 * (x * z + r * y),
		(real)2 * (x * y + r * z), (real)1 - (real)2 * (x * x + z * z), (real)2 * (y * z - r * x),
		(real)2 * (x * z - r * y), (real)2 * (y * z + r * x), (real)1 - (real)2 * (x * x + y * y));
	m3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
	real s[3] = { mod * scale[0], mod * scale[1], mod * scale[2] };
	S.c[0][0] = s[0]; S.c[1][1] = s[1]; S.c[2][2] = s[2];
	m3 M = m3_mul(S, R);
	m3 dSigma = m3_cols(
		dcov[0], (real)0.5f * dcov[1], (real)0.5f * dcov[2],
		(real)0.5f * dcov[1], dcov[3], (real)0.5f * dcov[4],
		(real)0.5f * dcov[2], (real)0.5f * dcov[4], dcov[5]);
	m3 dM = m3_mul(m3_scale((real)2.0f, M), dSigma);
	m3 Rt = m3_t(R), dMt = m3_t(dM);
	for (int i = 0; i < 3; i++)
		dscale[i] = Rt.c[i][0] * dMt.c[i][0] + Rt.c[i][1] * dMt.c[i][1] + Rt.c[i][2] * dMt.c[i][2];
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) dMt.c[i][j] *= s[i];
#define D(i, j) dMt.c[i][j]
	drot[0] = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
	drot[1] = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) - 4 * x * (D(2, 2) + D(1, 1));
	drot[2] = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) - 4 * y * (D(2, 2) + D(0, 0));
	drot[3] = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) - 4 * z * (D(1, 1) + D(0, 0));
#undef D
}

/* Rasterizer::backward: R0 rasterizer_impl.cu:340-433; BACKWARD::preprocess R0 backward.cu:559-622;
 * preprocessCUDA bwd R0 backward.cu:346-396. All grads zero-initialised (rasterize_points.cu:171-179). */
int orc_backward(const orc_in *in, const orc_out *o, orc_grads *g)
{
	const int P = in->P, M = in->M;
	if (in->variant == ORC_RP || in->variant == ORC_RF) return -2;
	memset(g->dL_dmean2D, 0, sizeof(real) * 3 * (size_t)P);
	memset(g->dL_dconic, 0, sizeof(real) * 4 * (size_t)P);
	memset(g->dL_dopacity, 0, sizeof(real) * (size_t)P);
	memset(g->dL_dcolor, 0, sizeof(real) * 3 * (size_t)P);
	memset(g->dL_dmean3D, 0, sizeof(real) * 3 * (size_t)P);
	memset(g->dL_dcov3D, 0, sizeof(real) * 6 * (size_t)P);
	if (M > 0) memset(g->dL_dsh, 0, sizeof(real) * 3 * (size_t)M * (size_t)P);
	memset(g->dL_dscale, 0, sizeof(real) * 3 * (size_t)P);
	memset(g->dL_drot, 0, sizeof(real) * 4 * (size_t)P);
	if (P == 0) return 0;
	double *dm2 = (double *)calloc((size_t)P * 2, sizeof(double));
	double *dcn = (double *)calloc((size_t)P * 3, sizeof(double));
	double *dop = (double *)calloc((size_t)P, sizeof(double));
	double *dcl = (double *)calloc((size_t)P * 3, sizeof(double));
	backward_render(in, o, g->dL_dpix, dm2, dcn, dop, dcl);
	ORC_PARALLEL_FOR(static)
	for (int i = 0; i < P; i++)
	{
		g->dL_dmean2D[3 * i] = (real)dm2[2 * i]; g->dL_dmean2D[3 * i + 1] = (real)dm2[2 * i + 1];
		g->dL_dconic[4 * i] = (real)dcn[3 * i]; g->dL_dconic[4 * i + 1] = (real)dcn[3 * i + 1]; g->dL_dconic[4 * i + 3] = (real)dcn[3 * i + 2];
		g->dL_dopacity[i] = (real)dop[i];
		for (int ch = 0; ch < 3; ch++) g->dL_dcolor[3 * i + ch] = (real)dcl[3 * i + ch];
	}
	free(dm2); free(dcn); free(dop); free(dcl);
	const real focal_y = in->H / ((real)2.0 * in->tanfovy), focal_x = in->W / ((real)2.0 * in->tanfovx);
	const real *proj = in->projmatrix;
	ORC_PARALLEL_FOR(dynamic, 1024)
	for (int idx = 0; idx < P; idx++)
	{
		if (!(o->radii[idx] > 0)) continue;
		const real *cov3D = in->cov3D_precomp ? in->cov3D_precomp + 6 * idx : o->cov3D + 6 * idx;
		real dconic[3] = { g->dL_dconic[4 * idx], g->dL_dconic[4 * idx + 1], g->dL_dconic[4 * idx + 3] };
		real dmean[3];
		backward_cov2D(in, idx, cov3D, focal_x, focal_y, dconic, dmean, g->dL_dcov3D + 6 * idx);
		const real *m = in->means3D + 3 * idx;
		real m_hom[4];
		transformPoint4x4(m, proj, m_hom);
		real m_w = (real)1.0 / (m_hom[3] + (real)0.0000001f);
		real mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
		real mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
		const real d2x = g->dL_dmean2D[3 * idx], d2y = g->dL_dmean2D[3 * idx + 1];
		real dm[3];
		dm[0] = (proj[0] * m_w - proj[3] * mul1) * d2x + (proj[1] * m_w - proj[3] * mul2) * d2y;
		dm[1] = (proj[4] * m_w - proj[7] * mul1) * d2x + (proj[5] * m_w - proj[7] * mul2) * d2y;
		dm[2] = (proj[8] * m_w - proj[11] * mul1) * d2x + (proj[9] * m_w - proj[11] * mul2) * d2y;
		for (int k = 0; k < 3; k++) dmean[k] += dm[k];
		if (in->colors_precomp == NULL && in->shs)
		{
			real add[3];
			backward_sh(in, o, idx, g->dL_dcolor + 3 * idx, add, g->dL_dsh + (size_t)idx * M * 3);
			for (int k = 0; k < 3; k++) dmean[k] += add[k];
		}
		for (int k = 0; k < 3; k++) g->dL_dmean3D[3 * idx + k] = dmean[k];
		if (in->cov3D_precomp == NULL && in->scales)
			backward_cov3D(in->scales + 3 * idx, in->scale_modifier, in->rotations + 4 * idx,
				g->dL_dcov3D + 6 * idx, g->dL_dscale + 3 * idx, g->dL_drot + 4 * idx);
	}
	return 0;
}

int orc_sizeof_real(void) { return (int)sizeof(real); }
