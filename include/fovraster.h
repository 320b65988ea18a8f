/*
 * fovraster.h -- C ABI of libfovraster_hip.so, the MI355X (gfx950) rasterizer library.
 *
 * This is the drop-in boundary for the rasterizer hot path of horizon-research/Fov-3DGS.
 * Each entry point replaces one function of the reference's native layer (paths relative to
 * /root/reference/fov3dgs/submodules/):
 *
 *   fr_forward        <-  CudaRasterizer::Rasterizer::forward
 *                           diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:36-61      ("original")
 *                           diff-gaussian-rasterization_pcheck_obb_sum/cuda_rasterizer/rasterizer.h (RS, + counts)
 *                           diff-gaussian-rasterization_pcheck_obb/cuda_rasterizer/rasterizer.h     (RP)
 *                           diff-gaussian-rasterization_fov_pcheck_obb/cuda_rasterizer/rasterizer.h:31-64 (RF)
 *                         as called by RasterizeGaussiansCUDA (…/rasterize_points.cu:35-115; RS :35-135; RF :35-152)
 *   fr_backward       <-  CudaRasterizer::Rasterizer::backward (…/cuda_rasterizer/rasterizer.h:63-85)
 *                         as called by RasterizeGaussiansBackwardCUDA (…/rasterize_points.cu:117-196)
 *   fr_mark_visible   <-  CudaRasterizer::Rasterizer::markVisible (…/rasterizer.h:24-29; rasterize_points.cu:198-217)
 *   fr_resize_fn      <-  the std::function<char*(size_t)> buffer callbacks (…/rasterize_points.cu:27-33)
 *
 * Conventions: every pointer in the argument structs is a DEVICE pointer to fp32/int32 data laid
 * out exactly as the reference's tensors (row-major, contiguous) unless stated otherwise; NULL
 * stands for the reference's "empty tensor". The library never allocates or frees device memory:
 * the three workspaces are obtained through the caller's resize callbacks, exactly like the
 * reference's geometry/binning/image buffers, and must be kept alive by the caller for
 * fr_backward. All work is enqueued on `stream` (a hipStream_t) and on helper streams of the library that fork from / join
 * into it. fr_forward returns num_rendered like the reference, so it waits ONCE for that count (a 32-byte block the tile
 * scan writes into pinned host memory, polled) -- but not for the frame. fr_forward_begin / fr_forward_finish are the two
 * halves of that call around the wait: a host that keeps two frames in flight (two streams, two workspace sets) runs the
 * head of frame n + 1 beside the sort and the blend of frame n. fr_backward never synchronises.
 * All functions return 0 on success or a negative FR_ERR_* code; fr_last_error() gives the message
 * (thread-local).
 */
#ifndef FOVRASTER_H
#define FOVRASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FR_ABI_VERSION 9

/* rasterizer variants (the reference ships them as separate extensions; `cuda_type` strings of
 * fov3dgs/gaussian_wrapper.py:11-23) */
enum {
	FR_VARIANT_ORIGINAL = 0,       /* diff-gaussian-rasterization                    */
	FR_VARIANT_PCHECK_OBB_SUM = 1, /* …_pcheck_obb_sum (training; counts/contributions) */
	FR_VARIANT_PCHECK_OBB = 2,     /* …_pcheck_obb (inference)                        */
	FR_VARIANT_FOV_PCHECK_OBB = 3, /* …_fov_pcheck_obb (foveated inference)           */
	/* pruning-metric variants of the training rasterizer (prune.py / metric_mask_learn.py); forward
	 * statistics differ from …_sum, backward is identical */
	FR_VARIANT_PCHECK_OBB_MAX = 4, /* …_pcheck_obb_max: count per in-support pixel, contributions = max alpha*T */
	FR_VARIANT_PCHECK_OBB_LWMC = 5, /* …_pcheck_obb_loss_weighted_max_count: per-pixel loss credited to its
	                                 * max-contribution Gaussian */
	/* the paper's shared-model foveated baseline ("SMFR", fps/naiveFR): …_naive_pcheck_obb. Tile levels, the level
	 * filter and the instance lists of FOV_PCHECK_OBB, but ONE colour / opacity per Gaussian (shs [P,16,3],
	 * opacities [P,1], highest_levels [P,1]; no shs_dcs), blended per naive forward.cu:258-480 (two-level tiles)
	 * and :482-580 (single-level tiles). Inference only, no packed layout. */
	FR_VARIANT_NAIVE_FOV_PCHECK_OBB = 6,
	/* the paper's multi-model foveated baseline ("MMFR", fps/MMFR-Q): …_mmfr_pcheck_obb. ONE call renders the share of
	 * eccentricity level `cur_level` from that level's own model (plain inputs: shs [P,16,3], opacities [P,1]); the caller
	 * adds the calls of all levels up (gaussian_renderer_fov_mmfr/__init__.py:76-162). Tiles whose tile_min (clamped at 0)
	 * lies outside (cur_level - 0.5, cur_level + 1) are skipped (left zero); two-level tiles weight every pixel by the
	 * smoothstep of its estimated level (mmfr forward.cu:255-420). highest_levels must be given and hold ZEROS [P,1]
	 * (the skip test reuses the level filter). Inference only, no packed layout. */
	FR_VARIANT_MMFR_PCHECK_OBB = 7
};

enum {
	FR_OK = 0,
	FR_ERR_INVALID = -1,   /* bad argument combination (message says which) */
	FR_ERR_HIP = -2,       /* a HIP call or kernel failed */
	FR_ERR_ALLOC = -3,     /* a resize callback returned NULL */
	FR_ERR_PREFILTERED = -4 /* `prefiltered` was set and a Gaussian lies behind the near plane (the reference traps:
	                         * cuda_rasterizer/auxiliary.h:156-160); nothing was rendered */
};

/* Workspace callback: make the buffer at least `bytes` long and return its device address.
 * Same contract as the reference's resizeFunctional lambdas. */
typedef char *(*fr_resize_fn)(void *user, size_t bytes);

typedef struct fr_forward_args {
	int32_t variant;
	int32_t P;            /* number of Gaussians */
	int32_t D;            /* active SH degree (0..3) */
	int32_t M;            /* SH coefficients per Gaussian in `shs` (16; RF: 15 = rest only); 0 if shs == NULL */
	int32_t W, H;         /* image size */
	int32_t prefiltered;  /* the caller promises that no Gaussian is behind the near plane; a violation is FR_ERR_PREFILTERED */
	int32_t debug;        /* != 0: synchronise + check after every launch */
	float tanfovx, tanfovy;
	float scale_modifier;
	/* foveation (RF only) */
	float gaze_x, gaze_y; /* normalised gaze in [0,1]^2 (reference passes gazeArray and .item()s it) */
	float alpha;          /* pooling-size slope (0.05) */
	void *stream;         /* hipStream_t */
	/* inputs */
	const float *background;     /* [3] */
	const float *means3D;        /* [P,3] */
	const float *shs;            /* [P,M,3] or NULL */
	const float *colors_precomp; /* [P,3] or NULL */
	const float *opacities;      /* [P,1]; RF: [P,4] per level */
	const float *scales;         /* [P,3] or NULL */
	const float *rotations;      /* [P,4] or NULL */
	const float *cov3D_precomp;  /* [P,6] or NULL */
	const float *viewmatrix;     /* [4,4] as the reference passes it (world-to-camera, transposed) */
	const float *projmatrix;     /* [4,4] */
	const float *campos;         /* [3] */
	const float *shs_dcs;        /* RF: [P,4,3] per-level DC coefficient */
	const float *highest_levels; /* RF: [P,1] float */
	/* outputs (caller-allocated; out_color and radii are written in full, they need no initialisation) */
	float *out_color;            /* [3,H,W] */
	int32_t *radii;              /* [P] */
	int32_t *gaussians_count;    /* RS: [P], else NULL (zeroed by the call, then accumulated) */
	float *contributions;        /* RS: [P], else NULL (zeroed by the call, then accumulated) */
	/* workspaces */
	fr_resize_fn geometry_resize;
	fr_resize_fn binning_resize;
	fr_resize_fn image_resize;
	void *resize_user[3];        /* passed to the three callbacks in that order */
	/* result */
	int32_t num_rendered;        /* out: number of (Gaussian,tile) instances after culling */
	int32_t max_tile_instances;  /* out: longest per-tile list */
	/* optional profiling: HOST pointer to FR_NUM_STAGES + 1 event handles made by fr_event_create() (any of
	 * them may be NULL), or NULL. Event i is recorded on `stream` just before stage i (FR_STAGE_* order), event
	 * FR_NUM_STAGES after the last one; nothing is synchronised. Read the stage durations later with
	 * fr_event_elapsed_ms(ev[i], ev[i+1]). */
	void **stage_events;
	const float *loss_map;       /* LWMC: [>= H*W] per-pixel weights (the reference passes a [3,H,W] map and reads
	                              * its first plane, …_loss_weighted_max_count/cuda_rasterizer/forward.cu:435) */
	/* optional (not RF): the SH coefficients as the two tensors a 3DGS model stores them in, so that the caller
	 * need not concatenate them every step (GaussianModel.get_features, scene/gaussian_model.py:83-86): when
	 * set, `shs` is the DC part [P,1,3] and `shs_rest` the others [P,M-1,3]; M stays their total (16). */
	const float *shs_rest;
	/* optional: the per-Gaussian inputs of a STATIC model in the layout the binning kernel reads fastest (made once,
	 * e.g. when the model is loaded; the reference has no counterpart -- its inputs are the separate tensors above,
	 * which stay mandatory and must hold the same values). The binning kernel reads a candidate's parameters from
	 * five different tensors, i.e. five mostly-unused cache lines per Gaussian, and 180-byte SH rows that straddle
	 * lines; with these it reads one 64-byte row and one aligned 256-byte row instead. Results are bit-identical.
	 *   packed_geom   [P][16]: x y z | sx sy sz | q0 q1 q2 q3 | highest_level (RF, else 0) | 0 | opacity[4]
	 *                          (RF: the four levels; else opacity, 0, 0, 0)         needs scales + rotations
	 *   packed_colour [P][64]: SH coefficients 1..15 [45] | SH coefficient 0 [3] (not RF) | shs_dcs [4][3] (RF) | 0[4]
	 *                                                                             needs shs with M = 16 (RF: 15)
	 *   packed_cull   [P][4] : x y z | bound of the covariance's spectral norm at scale_modifier 1 (max scale^2, times
	 *                          the factor a non-unit quaternion adds): all the conservative cull pass reads
	 * All three or none. fr_pack_geom / fr_pack_colour / fr_pack_cull fill them on the device. */
	const float *packed_geom;
	const float *packed_colour;
	const float *packed_cull;
	float cur_level;             /* MMFR: the level (0..3) this call renders */
	/* optional (variants with one opacity per Gaussian and no levels: ORIGINAL, PCHECK_OBB and the training variants; not
	 * with cov3D_precomp or the packed layout): `scales`, `rotations` and `opacities` hold the model's RAW parameters
	 * (log scale, unnormalised quaternion, opacity logit) and the kernels apply GaussianModel's activations themselves
	 * (exp, x / max(|x|, 1e-12), sigmoid: scene/gaussian_model.py:200-240) -- the same device expressions as
	 * fr_activate_forward, so the image is bit-identical to activating first. Saves the two streaming passes over all P
	 * Gaussians around every training step. */
	int32_t raw_activations;
	int32_t num_candidates;      /* out: entries of the library's list of the Gaussians that survived its cull pass
	                              * (fr_geometry_vis_list): the rows of a row-sparse backward call */
	/* optional diagnostic output (NULL = off, no cost): uint32 [T], per tile the number of entries of its depth-sorted list the
	 * blend FETCHED before every pixel of the tile was finished -- the reference's loop runs in batches of 256 (forward.cu:333-347:
	 * `done` is voted on per batch), this build's in batches of 64, so the figure is the list position rounded up to 64 (capped at
	 * the list's length); two-level RF tiles: the larger of their two level states. Cleared by the call. Sum / num_rendered =
	 * the fraction of the frame's instances the blend consumes (bench.py: config.list_consumed_frac). */
	uint32_t *list_consumed;
	/* optional (PCHECK_OBB_SUM only): != 0 = the caller does not read gaussians_count / contributions (both may be NULL):
	 * eff_finetune.py:107-108 drops gs_count / contribs of every training step. The blend then keeps only what the backward pass
	 * needs (final_T, n_contrib); image, radii, lists and gradients are those of the variant. */
	int32_t no_stats;
	/* optional diagnostic output (NULL = off): uint32 [T], per tile the number of (band of eight rows, list entry) pairs the blend
	 * evaluated -- the entries of the fetched batches that can reach the band at all (the kernels' reach mask), two-level RF tiles
	 * counted per level state. Cleared by the call. The unit the blend kernels' VALU time is proportional to (bench.py /
	 * profiles: vector instructions per blended pair). */
	uint32_t *blend_pairs;
	/* != 0: every kernel and fill of the frame goes to `stream` itself -- no helper streams (by default the library forks the image /
	 * statistics fills and the sort of the short lists onto two helper streams of its own per launch stream, which pays off for ONE
	 * frame at a time). A host that keeps several frames in flight on several streams sets it: a process's streams share a handful of
	 * hardware queues (four by default), and a frame's helper stream that lands in another frame's queue serialises the two. */
	int32_t no_helper_streams;
	/* != 0 (experimental, off by default): region-major emission. The binning kernel also lists its items by screen region (8 x 8 tiles),
	 * and the instances are placed by workgroups that own a region's tile buckets (k_emit_regions) instead of workgroups that own a
	 * share of every tile's. Same lists (the order inside a bucket is arbitrary either way; the per-tile sort fixes it). Writes 1.5 x
	 * the instance payload to HBM where the default kernel writes 3.7 x, and is slower at present; needs 8 P + 1 MB more of
	 * the geometry workspace (always reserved), tile grids of at most 256 regions, the 32-bit LDS histograms. */
	int32_t emit_regions;
} fr_forward_args;

enum { FR_STAGE_TILE_LEVELS = 0, FR_STAGE_PROJECT = 1, FR_STAGE_BIN = 2, FR_STAGE_TILE_SCAN = 3, FR_STAGE_EMIT = 4,
	FR_STAGE_TILE_SORT = 5, FR_STAGE_RENDER = 6, FR_NUM_STAGES = 7 };

typedef struct fr_backward_args {
	int32_t variant;             /* ORIGINAL, PCHECK_OBB_SUM, PCHECK_OBB_MAX or PCHECK_OBB_LWMC */
	int32_t P, D, M, R;          /* R = num_rendered of the forward call */
	int32_t W, H;
	int32_t debug;
	float tanfovx, tanfovy;
	float scale_modifier;
	void *stream;
	const float *background, *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations,
		*cov3D_precomp, *viewmatrix, *projmatrix, *campos;
	const int32_t *radii;        /* [P] from forward: the forward call's output UNMODIFIED (the whole-line stores of the narrow gradient
	                              * tensors take `radii > 0` for "this row is written by the per-Gaussian pass": a clamped or edited copy leaves rows unwritten) */
	/* the three workspaces filled by fr_forward. The call accumulates into the geometry workspace's per-Gaussian gradient
	 * sums and clears them again before it ends: calling fr_backward twice over one forward state gives the same
	 * gradients twice (the reference: fresh torch::zeros per call), but two calls must not run concurrently on it. */
	char *geometry;
	const char *binning, *image;
	const float *dL_dpix;        /* [3,H,W] */
	/* outputs: caller-allocated, WRITTEN IN FULL by the call (rows of Gaussians the view does not touch become zero; the
	 * reference allocates them with torch::zeros, rasterize_points.cu:171-179, and only writes the visible rows): no
	 * initialisation needed */
	float *dL_dmean2D;           /* [P,3] */
	float *dL_dconic;            /* [P,2,2]; may be NULL: an intermediate of the reference (its sums are kept per visible-list entry here) */
	float *dL_dopacity;          /* [P,1] */
	float *dL_dcolor;            /* [P,3]; may be NULL unless colors_precomp is given (otherwise an intermediate) */
	float *dL_dmean3D;           /* [P,3] */
	float *dL_dcov3D;            /* [P,6]; may be NULL when cov3D_precomp is NULL (then it is an intermediate nobody reads) */
	float *dL_dsh;               /* [P,M,3] */
	float *dL_dscale;            /* [P,3] */
	float *dL_drot;              /* [P,4] */
	void **stage_events;         /* optional: HOST pointer to 5 event handles (any may be NULL): [0] [1] [2] on `stream` before the
	                              * tile pass (k_render_bwd), between it and the per-Gaussian pass (k_preprocess_bwd), and after that;
	                              * [3] [4] around the zero fill of the gradient tensors, on the helper stream it runs on */
	const float *shs_rest;       /* split SH input, see fr_forward_args.shs_rest ... */
	float *dL_dsh_rest;          /* ... then dL_dsh is [P,1,3] and dL_dsh_rest [P,M-1,3] */
	int32_t raw_activations;     /* as in the forward call: dL_dscale / dL_drot / dL_dopacity are then gradients w.r.t. the RAW
	                              * parameters (what fr_activate_backward would make of them) */
	/* optional (extension; the reference returns dense tensors, rasterize_points.cu:171-179): ROW-SPARSE gradients. The
	 * gradient tensors are then COMPACT -- [C, .] instead of [P, .], C = fr_forward_args.num_candidates of the forward call --
	 * and row i belongs to the Gaussian vis_list[i] (fr_geometry_vis_list: increasing indices; every Gaussian the view touches is
	 * in the list, a candidate that landed in no tile has a zero row). Every row is written, nothing is zero-filled: at 6 M
	 * Gaussians a training step's backward pass otherwise clears 1.5 GB to write 0.5 GB. */
	int32_t row_sparse;
	uint32_t *blend_pairs;       /* optional diagnostic, as fr_forward_args.blend_pairs: [T], pairs k_render_bwd evaluated; cleared by the call */
	/* != 0: the caller has zero-filled the dense gradient tensors itself (fr_backward_prefill, or any other way) and the fill is
	 * complete on `stream`: the call then only writes the rows of the Gaussians the view touches. Lets a host clear the 1.5 GB of a
	 * 6 M-Gaussian model's gradients beside the work BETWEEN its forward and backward calls (the image loss) instead of beside
	 * k_render_bwd, which pays 0.11 ms for the company. */
	int32_t outputs_zeroed;
	/* optional (extension; multi-GPU training): the per-Gaussian half of the call in `num_ranges` > 1 pieces over
	 * increasing, disjoint ranges of ROWS (Gaussian indices) that cover [0, P). As soon as the kernels that complete range k are
	 * enqueued on `stream` -- every gradient tensor's rows [row_lo, row_hi) are then final once the stream gets there, zeros included --
	 * the HOST function range_done(range_user, k, row_lo, row_hi) is called (from inside fr_backward, on the calling thread): a host
	 * records an event there and starts summing those rows over its ranks on a communication stream while the later ranges are still
	 * being computed. Range k starts at row (P k / num_ranges) rounded down to a multiple of 32 (the visible list is in index order,
	 * so a range of rows is a range of list entries): equal shares of the INDEX range, not of the visible Gaussians. Dense
	 * gradients only (ignored with row_sparse); at most 16 ranges. 0 / 1 or range_done == NULL: one piece, no call. */
	int32_t num_ranges;
	void (*range_done)(void *user, int32_t k, int32_t row_lo, int32_t row_hi);
	void *range_user;
} fr_backward_args;

int fr_abi_version(void);
const char *fr_last_error(void);

/* hipEvent wrappers so a host that only speaks the C ABI can time stages on the launch stream */
void *fr_event_create(void);
void fr_event_destroy(void *event);
int fr_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on `stop` */

int fr_forward(fr_forward_args *args);
/* The same call in two halves (fr_forward == begin, then finish):
 *   fr_forward_begin   validates, asks for the geometry and image workspaces, enqueues the head of the frame (fills, tile levels,
 *                      cull pass, projection, tile counts, tile scan) and returns without waiting for anything; *frame is
 *                      the handle of the frame in flight (NULL on error). `args` must stay alive and unchanged until
 *                      fr_forward_finish(*frame) has returned, which must be called exactly once, by the same host thread.
 *   fr_forward_finish  waits for the instance count (the frame's one host synchronisation), asks for the binning workspace
 *                      (possibly twice: once sized like the largest frame of the kind so far, again if that was too small
 *                      -- the last answer is the one in use), enqueues emission, sort, colours and blend, fills
 *                      args->num_rendered / max_tile_instances and releases the handle (also when it fails).
 * Frames in flight at the same time need their own stream and their own three workspaces each. */
typedef struct fr_frame fr_frame;
int fr_forward_begin(fr_forward_args *args, fr_frame **frame);
int fr_forward_finish(fr_frame *frame);
/* Drop a frame between its halves WITHOUT running its tail (a host that gives up on a frame, e.g. a garbage-collected handle):
 * joins the frame's helper stream into its launch stream, releases the pinned totals block and the handle; no wait, no callback,
 * no launch. out_color is left undefined. */
int fr_forward_abandon(fr_frame *frame);
/* Fill fr_forward_args.packed_geom / packed_colour / packed_cull (device buffers of P*16 / P*64 / P*4 floats) from the tensors of the
 * same names; opacities is [P,levels] with levels = 1 or 4, highest_levels / shs_dcs may be NULL (not RF);
 * shs_rest NULL: shs is [P,16,3] (RF: [P,15,3] = coefficients 1..15 and shs_dcs given), else shs = [P,1,3]. */
int fr_pack_geom(int32_t P, const float *means3D, const float *scales, const float *rotations, const float *opacities,
	int32_t levels, const float *highest_levels, float *packed_geom, void *stream);
int fr_pack_cull(int32_t P, const float *means3D, const float *scales, const float *rotations, float *packed_cull, void *stream);
int fr_pack_colour(int32_t P, const float *shs, const float *shs_rest, const float *shs_dcs, float *packed_colour, void *stream);
int fr_backward(const fr_backward_args *args);
/* Zero-fill the dense gradient tensors named in `args` (the dL_d* pointers with P, M, shs / shs_rest / colors_precomp as fr_backward
 * would be called; nothing else is read) with one kernel on `fill_stream`; pair with fr_backward_args.outputs_zeroed. No-op for
 * row_sparse. */
int fr_backward_prefill(const fr_backward_args *args, void *fill_stream);
int fr_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix,
	uint8_t *present /* [P] bool */, void *stream);

/* The parameter activations in front of the rasterizer in a training iteration, one pass each way (SURVEY.md 8f rank 3;
 * replaces GaussianModel.get_scaling / get_rotation / get_opacity, fov3dgs/scene/gaussian_model.py:200-240, i.e.
 * exp [P,3], x / max(|x|, 1e-12) [P,4], sigmoid [P,1], and their autograd). Backward: dL_d* are the gradients w.r.t. the
 * activated values (NULL = zero), dL_draw_* the results, written in full. */
int fr_activate_forward(int32_t P, const float *raw_scaling, const float *raw_rotation, const float *raw_opacity, float *scaling, float *rotation,
	float *opacity, void *stream);
int fr_activate_backward(int32_t P, const float *raw_scaling, const float *raw_rotation, const float *raw_opacity, const float *dL_dscaling,
	const float *dL_drotation, const float *dL_dopacity, float *dL_draw_scaling, float *dL_draw_rotation, float *dL_draw_opacity, void *stream);

/* Image loss of a training iteration, fused (replaces fov3dgs/utils/loss_utils.py:17-18 l1_loss and
 * :37-76 ssim as eff_finetune.py:124-125 combines them). img / target: [C,H,W] fp32 device tensors.
 *   forward:  partials[b] = (sum |img - target|, sum ssim_map) over tile b (16 x 32 pixels) of one channel,
 *             b < fr_l1_ssim_blocks(C,H,W); the caller adds them up (l1 = sum0 / (C H W), ssim = sum1 / (C H W)).
 *             dmaps [3,C,H,W] (optional, NULL = value only) keeps what the backward needs.
 *   finish:   out[0..2] = (loss, l1, ssim) from the partials on the device (sums in double, in block order: the same
 *             numbers on every run): l1 = sum0 / (C H W), ssim = sum1 / (C H W), loss = (1-lambda) l1 + lambda (1 - ssim).
 *   backward: dL_dimg = s (w_l1 * sign(img - target) + w_ssim * d(sum ssim_map)/d img), written in full; s = *grad_scale
 *             (a DEVICE scalar: the upstream gradient of the loss, read by the kernel -- no host round trip, no extra
 *             pass over the image) or 1 when grad_scale is NULL.
 *             For loss = (1-l) L1 + l (1 - SSIM): w_l1 = (1-l) / (C H W), w_ssim = -l / (C H W). */
int64_t fr_l1_ssim_blocks(int32_t C, int32_t H, int32_t W);
int fr_l1_ssim_forward(int32_t C, int32_t H, int32_t W, const float *img, const float *target, float *dmaps, float *partials, void *stream);
int fr_l1_ssim_finish(int32_t C, int32_t H, int32_t W, const float *partials, float lambda_dssim, float *out3, void *stream);
int fr_l1_ssim_backward(int32_t C, int32_t H, int32_t W, const float *img, const float *target, const float *dmaps, float w_l1, float w_ssim,
	const float *grad_scale, float *dL_dimg, void *stream);

/* Bytes fr_forward will request for the geometry / image workspaces (P, W, H dependent) and for the
 * binning workspace given a number of instances; lets a caller pre-size persistent buffers. */
size_t fr_geometry_bytes(int32_t variant, int32_t P);
size_t fr_image_bytes(int32_t variant, int32_t W, int32_t H);
size_t fr_binning_bytes(int32_t variant, int64_t num_instances);

/* Introspection for tests: device pointers to internal per-tile / per-instance state inside the workspaces.
 * ranges: uint32 [T,2]. point_list: uint32 [num_rendered], the per-tile depth-sorted lists -- of ITEMS: an item is a
 * position in the library's list of the Gaussians that survive its cull pass, which is in increasing Gaussian index
 * (fr_geometry_vis_list), so vis_list[point_list[i]] is the reference's point_list entry and the order is the
 * reference's. Everything the library keeps per candidate Gaussian (records, level colours, level ranges, walk records)
 * is indexed by item: dense rows instead of rows scattered over all P Gaussians. */
const uint32_t *fr_image_ranges(int32_t variant, int32_t W, int32_t H, const char *image);
const uint32_t *fr_binning_point_list(int32_t variant, int64_t num_instances, const char *binning);
const float *fr_image_final_T(int32_t variant, int32_t W, int32_t H, const char *image);
const uint32_t *fr_image_n_contrib(int32_t variant, int32_t W, int32_t H, const char *image);
/* per-item records float[.][12] = (x, y, conic a, conic b | conic c, opacity, r, g | b, depth, clamp bits, Gaussian index);
 * valid for items whose Gaussian has radii > 0 (RF: the second third holds (conic c, highest level, -, -)) */
const float *fr_geometry_records(int32_t variant, int32_t P, const char *geometry);
/* RF: device pointer to float[5][T] = levels, tile_min, grad_x, grad_y, blending(0/1 as float) */
const float *fr_image_tile_levels(int32_t W, int32_t H, const char *image);
/* the Gaussians that survived the cull pass, in increasing index: uint32 [count], with count = *fr_geometry_vis_count
 * (a device word); "item" i everywhere else means position i of this list */
const uint32_t *fr_geometry_vis_list(int32_t variant, int32_t P, const char *geometry);
const uint32_t *fr_geometry_vis_count(int32_t variant, int32_t P, const char *geometry);
/* walk record of item i: float[16] = (centre x, y, OBB axis 1 x, y | axis 2 x, y, half length 1, 2 |
 * id + flags << 30, depth bits, clipped rectangle x0 + y0 << 16, its width | tiles, highest level, -, -)
 * -- what the reference keeps as eigen_vecs / eigen_lengths (RS forward.cu:244-265) */
const float *fr_geometry_walk_records(int32_t variant, int32_t P, const char *geometry);
/* RF: per-item per-level (r, g, b, opacity) float[.][4][4] (compute_fov_colors, RF rasterizer_impl.cu:490-530; only
 * the levels of the Gaussian's level range are written) and the packed level ranges uint32 [.] = lo | hi << 8 */
const float *fr_geometry_level_colours(int32_t P, const char *geometry);
const uint32_t *fr_geometry_level_ranges(int32_t P, const char *geometry);

#ifdef __cplusplus
}
#endif
#endif /* FOVRASTER_H */
