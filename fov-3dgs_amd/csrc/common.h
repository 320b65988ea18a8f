// Internal shared definitions of libfovraster_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/fovraster.h"

#define FR_TILE 16            // tile edge in pixels (reference config.h:15-17 BLOCK_X/BLOCK_Y)
#define FR_TILE_PIX 256
#define FR_FOV_LEVELS 4       // RF auxiliary.h:26 fov_num
#define FR_SORT_LDS_MAX 8192  // longest per-tile list sorted inside LDS (64 KiB of u64 keys)
#ifndef FR_BIN_THREADS
#define FR_BIN_THREADS 768    // workgroup size of k_bin and k_emit (one persistent workgroup per CU): twelve waves hide each other's round trips
                              // (eight: foveated bin stage 176 us, twelve: 162, sixteen: 165 -- at sixteen the 128-register cap spills)
#endif
// persistent workgroups of the binning kernels: 2 per CU by LDS (up to 76 KiB each); k_bin's ~145 VGPRs let only one of
// them run at a time (2 waves/SIMD), the other finds the slab counters empty -- 384-thread workgroups (3 waves/SIMD)
// measured no faster (0.253 vs 0.247 ms): the kernel follows its heaviest slabs, not its occupancy
#define FR_BIN_BLOCKS 512
#define FR_PROJ_MAX_WAVES 8192                 // k_project's grid is capped to this many waves ...
#define FR_CROW_PAD (64 * FR_PROJ_MAX_WAVES)     // ... each of which may leave its last chunk's worth of row slots unused
#define FR_LDS_HIST_MAX_TILES 16384 // per-workgroup LDS tile histogram of 32-bit counts up to 64 KiB
#define FR_ITEM_NONE 0xffffffffu    // GeomWS::lrange of an item that lands in no tile
#define FR_LDS_HIST16_MAX_TILES 34816 // ... of 16-bit counts (two tiles per word) beyond that: a 4K frame has 32 400 tiles
#define FR_HIST16_MAX_SLABS (65535 / FR_BIN_THREADS) // slabs a wave of k_bin takes at most then (85 of 64 items for twelve waves): a
                                     // workgroup's waves x slabs x 64 items stay below 65 536, so no tile's 16-bit count can wrap
#define FR_BIG_TNUM 64        // splats with at least this many tiles are binned by a whole wave at a time
#define FR_GIANT_TNUM 1024     // ... and splats with this many by the whole workgroup, after its slab loop
#define FR_GIANT_MAX 64         // giant splats a workgroup can set aside (more: handled like big ones)
#define FR_SORT_SPLIT_MIN 16384      // tile lists with at least this many entries do not fit one workgroup's LDS: split by depth before sorting
#define FR_SORT_CHUNK_TARGET 960    // ... into chunks of about this many entries (just under the 1024-key sort size)
#define FR_SORT_FINE_BUCKETS 2048   // depth buckets of the split
// chunks are cut at multiples of the target in the running count, so a list of n entries yields at most
// n / target + 1 of them, and there are at most D / FR_SORT_SPLIT_MIN long lists
#define FR_SORT_MAX_CHUNKS(D) ((size_t)(D) / FR_SORT_CHUNK_TARGET + (size_t)(D) / FR_SORT_SPLIT_MIN + 16)
#define FR_LV_BBOX_STRIDE 32  // words between the level boxes of ImageWS::lv_bbox (one 128-byte line each)
#define FR_SLAB_CTR_WORDS 32  // one 128-byte line of frame counters (GeomWS::slab_ctr)

namespace fr {

// variant traits
#define FR_VARIANT_SUM_NOSTATS 100 // internal (k_render only): pcheck_obb_sum's blend without gaussians_count / contributions
__host__ __device__ inline bool has_stats(int v) { return v == FR_VARIANT_PCHECK_OBB_SUM || v == FR_VARIANT_PCHECK_OBB_MAX || v == FR_VARIANT_PCHECK_OBB_LWMC; }
__host__ __device__ inline bool has_backward(int v) { return v == FR_VARIANT_ORIGINAL || has_stats(v); }
// variants that bin by eccentricity level (tile level map, level filter): RF and the shared-model baseline
__host__ __device__ constexpr inline bool is_fov(int v) { return v == FR_VARIANT_FOV_PCHECK_OBB || v == FR_VARIANT_NAIVE_FOV_PCHECK_OBB || v == FR_VARIANT_MMFR_PCHECK_OBB; }

// ---- workspace layouts -------------------------------------------------------------------
// All sub-arrays are 256-byte aligned inside the caller's buffers.
__host__ __device__ inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Geometry workspace. Everything per CANDIDATE is indexed by the candidate's position in vis_list ("item"), the list of
// Gaussians that survive the cull pass IN INCREASING GAUSSIAN INDEX (k_project): the binning kernel's outputs
// are then dense rows written by consecutive lanes, and the blend kernels gather from a few tens of MB instead of from
// 48-byte rows spread 12 % dense over P of them. (Measured on MI355X, tools/scratch/gather_rate.hip: a store instruction
// whose 64 lanes hit 64 different lines costs a CU ~11 cycles per lane, a load ~3; round 2's per-Gaussian-index records
// were 7-9 such stores per candidate -- more than a third of k_bin.) Because items are in index order, sorting a tile's
// instances by (depth bits, item) gives the order the reference's stable sort by (tile, depth) of index-ordered keys does.
// rec = 3 float4 per item:
//   [0] = (mean2D.x, mean2D.y, conic.a, conic.b)
//   [1] = (conic.c, opacity, r, g)          RF: (conic.c, highest_level, 0, 0)
//   [2] = (b, depth, clamp bits, Gaussian index as int bits)      SMFR: (b, depth, highest level, Gaussian index)
struct GeomWS {
	float4 *rec;        // [3P]
	float *cov3D;       // training variants: [16P] one 64-byte row per item, written by k_bin for the
	                    // backward pass: (xyz | raw scale | rotation | 3D covariance) -- one coalesced row instead of four gathers
	float4 *acc;        // training variants: [4P] one 64-byte row of sums per item, zeroed by k_bin,
	                    //      accumulated by k_render_bwd (one atomic instruction per list entry and wave), read by k_preprocess_bwd:
	                    //      (dL/d colour r, g, b, M10 | M01, M20, M11, M02 | M00, -, -, - | 1 / |raw quaternion|, -, -, -) with M_ij the
	                    //      moments of G dL/dalpha about the splat centre (the mean2D / conic / opacity gradients are linear in them)
	float4 *wrec;       // [4P] walk record of item i at [4i..4i+3], written by k_bin for k_emit:
	                    //      (cx, cy, e1x, e1y | e2x, e2y, len1, len2 | Gaussian index + flags << 30, depth bits, x0 + y0 << 16, width |
	                    //      tiles, highest level, -, -); flags: 1 = lands in a tile, 2 = the OBB test applies
	float4 *lvl;        // [4P] RF per-level (r,g,b,opacity) of item i at [4i..4i+3] (k_bin)
	uint32_t *lrange;   // [P]  per item, written by k_bin: 0xffffffff = the item lands in no tile (culled everywhere), else the packed
	                    //      level range lo | hi<<8 (RF; 0 for the variants without levels)
	uint32_t *slab_ctr; // [FR_SLAB_CTR_WORDS] {prefiltered violation flag, number of entries in vis_list, workgroups of the cull pass that
	                    // are done, odd highest level seen, workgroups of k_bin that are done (its last one runs the tile scan)};
	                    // words 8 .. 8 + 16: the backward pass's range bounds (k_range_bounds, fr_backward_args.num_ranges)
	uint32_t *vis_list; // [P]  indices of the Gaussians that survive the cull pass, increasing
	uint32_t *vis_seg;  // [P + FR_CROW_PAD] the same indices as k_project's waves leave them: wave w of the cull pass owns the slots
	                    //      from w * (its chunks) * 64 on and fills them in the order it meets its survivors (its chunks are consecutive)
	uint32_t *proj_counts; // [FR_PROJ_MAX_WAVES] survivors of every wave of the cull pass ...
	uint32_t *wbase;       // [FR_PROJ_MAX_WAVES + 1] ... and their exclusive running sums (the cull pass's last workgroup): the first ITEM of
	                       // every wave's region; [waves] = the number of items
	float4 *crow;       // [3 (P + FR_CROW_PAD)] foveated variants' candidate rows (xyz, scale | scale.yz, rotation.xy | rotation.zw, highest
	                    //      level, index), same slots as vis_seg ...
	// region-major emission (round 6, k_emit_regions): the screen is cut into regions of 8 x 8 tiles; every binning workgroup leaves,
	// sorted by region, the items whose walk rectangle reaches the region (an item is listed once per region it reaches)
	uint2 *rtab;        // [FR_BIN_BLOCKS][FR_MAX_REGIONS] (offset inside the workgroup's segment of wlist, count) of (workgroup, region)
	uint32_t *wlist;    // [wlist_cap(P)] the workgroups' segments, one after the other (each: capacity / workgroups entries)
	uint32_t *rtotal;   // [FR_MAX_REGIONS] list entries of every region (zeroed by k_project's last workgroup, added up by k_bin's)
	uint32_t *rchunk;   // [FR_MAX_REGIONS + 1] first chunk (of FR_ER_CHUNK entries) of every region's list, [regions] = all chunks (k_bin's last workgroup)
	size_t bytes;
};
#define FR_REGION_TILES 8    // tiles per side of a region
#define FR_MAX_REGIONS 256   // region-major emission for tile grids of at most this many regions (1080p: 15 x 9 = 135; 1440p: 20 x 12)
__host__ __device__ inline size_t wlist_cap(size_t P) { return 2 * P + 65536; }
__host__ __device__ inline GeomWS carve_geom(int variant, size_t P, char *base)
{
	GeomWS g; size_t off = 0;
	g.rec = (float4 *)(base + off); off = align_up(off + P * 3 * sizeof(float4));
	// variants with a backward pass keep a 64-byte row per vis_list entry for it (see GeomWS::cov3D); the others only
	// need room for the developer timers
	const bool keeps = has_backward(variant);
	g.cov3D = (float *)(base + off); off = align_up(off + P * (keeps ? 16 : 6) * sizeof(float));
	g.acc = nullptr;
	if (keeps) { g.acc = (float4 *)(base + off); off = align_up(off + P * 4 * sizeof(float4)); }
	g.lvl = nullptr;
	g.wrec = (float4 *)(base + off); off = align_up(off + P * 4 * sizeof(float4));
	if (variant == FR_VARIANT_FOV_PCHECK_OBB) { g.lvl = (float4 *)(base + off); off = align_up(off + P * FR_FOV_LEVELS * sizeof(float4)); }
	g.lrange = (uint32_t *)(base + off); off = align_up(off + P * sizeof(uint32_t));
	g.slab_ctr = (uint32_t *)(base + off); off = align_up(off + FR_SLAB_CTR_WORDS * sizeof(uint32_t));
	g.vis_list = (uint32_t *)(base + off); off = align_up(off + P * sizeof(uint32_t));
	g.vis_seg = (uint32_t *)(base + off); off = align_up(off + (P + FR_CROW_PAD) * sizeof(uint32_t));
	g.proj_counts = (uint32_t *)(base + off); off = align_up(off + FR_PROJ_MAX_WAVES * sizeof(uint32_t));
	g.wbase = (uint32_t *)(base + off); off = align_up(off + (FR_PROJ_MAX_WAVES + 1) * sizeof(uint32_t));
	// candidate rows: only the foveated variants' cull pass stores them (k_project's ROWS); 52 B per Gaussian the plain and
	// training frames need not carry
	g.crow = nullptr;
	if (is_fov(variant)) { g.crow = (float4 *)(base + off); off = align_up(off + (P + FR_CROW_PAD) * 3 * sizeof(float4)); }
	g.rtab = (uint2 *)(base + off); off = align_up(off + (size_t)FR_BIN_BLOCKS * FR_MAX_REGIONS * sizeof(uint2));
	g.wlist = (uint32_t *)(base + off); off = align_up(off + wlist_cap(P) * sizeof(uint32_t));
	g.rtotal = (uint32_t *)(base + off); off = align_up(off + FR_MAX_REGIONS * sizeof(uint32_t));
	g.rchunk = (uint32_t *)(base + off); off = align_up(off + (FR_MAX_REGIONS + 1) * sizeof(uint32_t));
	g.bytes = off + 256;
	return g;
}

// Image workspace (per pixel / per tile).
struct ImageWS {
	float *final_T;       // [W*H]
	uint32_t *n_contrib;  // [W*H]
	uint2 *ranges;        // [T]
	uint32_t *tile_count; // [T]  instance counter, then emission cursor
	uint32_t *lv_bbox;    // [5][FR_LV_BBOX_STRIDE] RF: box of the tiles with tile_min < k as {gx - x0, gy - y0, x1, y1} (0 = empty), k = 0..4
	uint32_t *totals;     // [16] {num_instances, max per tile, #tiles with >= 2048, #tiles with 512..2047, #sort chunks, #blend items, #tiles with >= 4096, prefiltered violation, #tiles with >= 8192, #tiles with >= 16384}
	uint32_t *render_items; // [4T] blend work items, longest lists first: tile << 3 | band | level state << 1 | two-level << 2 (k_tile_scan)
	uint32_t *tile_order; // [T]  tile ids by descending list length (power-of-two buckets): longest first
	float *tile_lv;       // RF [5][T]: level, tile_min, grad_x, grad_y, blending
	uint32_t *hist;       // [FR_BIN_BLOCKS][T] per-workgroup tile histograms (null if T too large for LDS)
	size_t bytes;
};
__host__ __device__ inline int bin_blocks(int P)
{
	const int slabs = (P + FR_BIN_THREADS - 1) / FR_BIN_THREADS;
	return slabs < FR_BIN_BLOCKS ? (slabs > 0 ? slabs : 1) : FR_BIN_BLOCKS;
}
__host__ __device__ inline ImageWS carve_image(int variant, int W, int H, char *base)
{
	ImageWS s; size_t off = 0;
	const size_t N = (size_t)W * H;
	const size_t T = (size_t)((W + FR_TILE - 1) / FR_TILE) * ((H + FR_TILE - 1) / FR_TILE);
	s.final_T = (float *)(base + off); off = align_up(off + N * sizeof(float));
	s.n_contrib = (uint32_t *)(base + off); off = align_up(off + N * sizeof(uint32_t));
	s.ranges = (uint2 *)(base + off); off = align_up(off + T * sizeof(uint2));
	s.tile_count = (uint32_t *)(base + off); off = align_up(off + T * sizeof(uint32_t));
	s.lv_bbox = (uint32_t *)(base + off); off = align_up(off + 5 * FR_LV_BBOX_STRIDE * sizeof(uint32_t));
	s.totals = (uint32_t *)(base + off); off = align_up(off + 16 * sizeof(uint32_t));
	s.render_items = (uint32_t *)(base + off); off = align_up(off + 4 * T * sizeof(uint32_t));
	s.tile_order = (uint32_t *)(base + off); off = align_up(off + T * sizeof(uint32_t));
	s.tile_lv = nullptr;
	if (is_fov(variant)) { s.tile_lv = (float *)(base + off); off = align_up(off + 5 * T * sizeof(float)); }
	s.hist = nullptr;
	if (T <= FR_LDS_HIST16_MAX_TILES) { s.hist = (uint32_t *)(base + off); off = align_up(off + (size_t)FR_BIN_BLOCKS * T * sizeof(uint32_t)); }
	s.bytes = off + 256;
	return s;
}

// Binning workspace (per (Gaussian,tile) instance), laid out for a CAPACITY of D instances (>= the frame's number: the
// forward call may size it before the count is known, see fr_forward). point_list comes first, so that its address -- all
// the backward pass and the introspection entry points need -- does not depend on the capacity.
struct BinWS {
	uint32_t *point_list; // [D] gaussian ids, sorted per tile
	uint64_t *entries;    // [D] (depth bits << 32 | gaussian id), bucketed by tile
	uint64_t *entries2;   // [D] long lists regrouped into depth-ordered chunks (k_split_long)
	uint2 *chunks;        // [FR_SORT_MAX_CHUNKS(D)] ranges of those chunks inside entries2 / point_list
	uint32_t *round_flags; // RS / LWMC blend: one bit per (tile, 256-entry round), see k_render (last: earlier offsets do not depend on T)
	size_t bytes;
};
// flag f(tile, round) = floor(range_start / 256) + tile + round < D / 256 + T + 1
__host__ __device__ inline size_t round_flag_words(int64_t D, int64_t T) { return (size_t)((D / 256 + T + 64) / 32 + 1); }
__host__ __device__ inline BinWS carve_bin(int64_t D, char *base, int64_t T = 1 << 20)
{
	BinWS b; size_t off = 0;
	b.point_list = (uint32_t *)(base + off); off = align_up(off + (size_t)D * sizeof(uint32_t));
	b.entries = (uint64_t *)(base + off); off = align_up(off + (size_t)D * sizeof(uint64_t));
	b.entries2 = (uint64_t *)(base + off); off = align_up(off + (size_t)D * sizeof(uint64_t));
	b.chunks = (uint2 *)(base + off); off = align_up(off + FR_SORT_MAX_CHUNKS(D) * sizeof(uint2));
	b.round_flags = (uint32_t *)(base + off); off = align_up(off + round_flag_words(D, T) * sizeof(uint32_t));
	b.bytes = off + 256;
	return b;
}

// ---- small device helpers ----------------------------------------------------------------
// float -> int with the saturating / NaN->0 behaviour of v_cvt_i32_f32 (and of the CUDA cvt.rzi
// the reference relies on for off-screen splats)
__device__ __forceinline__ int f2i(float v)
{
	if (v != v) return 0;
	if (v >= 2147483648.0f) return 2147483647;
	if (v <= -2147483648.0f) return (-2147483647 - 1);
	return (int)v;
}

// tile rectangle of a splat: reference auxiliary.h:46-56
__device__ __forceinline__ void get_rect(float px, float py, int max_radius, int gx, int gy,
	int &x0, int &y0, int &x1, int &y1)
{
	const float r = (float)max_radius;
	x0 = min(gx, max(0, f2i((px - r) / FR_TILE)));
	y0 = min(gy, max(0, f2i((py - r) / FR_TILE)));
	x1 = min(gx, max(0, f2i((px + r + (FR_TILE - 1)) / FR_TILE)));
	y1 = min(gy, max(0, f2i((py + r + (FR_TILE - 1)) / FR_TILE)));
}

// same with a fractional radius (conservative early-out of k_project)
__device__ __forceinline__ void get_rect_f(float px, float py, float r, int gx, int gy, int &x0, int &y0, int &x1, int &y1)
{
	x0 = min(gx, max(0, f2i((px - r) / FR_TILE)));
	y0 = min(gy, max(0, f2i((py - r) / FR_TILE)));
	x1 = min(gx, max(0, f2i((px + r + (FR_TILE - 1)) / FR_TILE)));
	y1 = min(gy, max(0, f2i((py + r + (FR_TILE - 1)) / FR_TILE)));
}

// Rectangle of tiles the binning kernels WALK for one splat. The reference walks getRect()'s rectangle
// (the 3-sigma circle) and then rejects tiles with the OBB test (RS rasterizer_impl.cu:99-123) and, in RF,
// the level test tile_min < highest_level + 1 (RF rasterizer_impl.cu:349-372). A tile outside the
// axis-aligned box of the OBB's corners fails the first two axes of that OBB test, and a tile outside the
// bounding box of {tiles with tile_min < ceil(highest_level + 1)} fails the level test, so clipping the walk
// to both boxes (with a safety margin far above float rounding) drops only pairs that would be rejected
// anyway: results are unchanged, a third to a half of the pairs are never visited, and splats whose
// clipped rectangle is empty are culled before they reach the binning kernels at all.
// boxtest: the splat's FULL rectangle has more than one tile, i.e. the reference applies the OBB test.
struct WalkRect { int x0, y0, x1, y1; uint32_t tnum; bool boxtest; };
template <bool CULL, bool FOV>
// lv_box_stride: uint4 rows between two level boxes (the global table has one 128-byte line per box; k_bin keeps a dense LDS copy)
__device__ __forceinline__ WalkRect walk_rect(float px, float py, int radius, int gx, int gy, float4 ev, float2 el,
	float hl, const uint4 *__restrict__ lv_boxes, const int lv_box_stride = FR_LV_BBOX_STRIDE / 4)
{
	WalkRect w;
	get_rect(px, py, radius, gx, gy, w.x0, w.y0, w.x1, w.y1);
	w.boxtest = CULL && ((uint32_t)(w.y1 - w.y0) * (uint32_t)(w.x1 - w.x0) > 1u);
	if (w.boxtest)
	{
		// obb_hits_tile keeps tile tx only if min(vx) - 16 <= 16 tx <= max(vx) (up to rounding)
		const float hx = fabsf(el.x * ev.x) + fabsf(el.y * ev.z), hy = fabsf(el.x * ev.y) + fabsf(el.y * ev.w);
		if (hx < 1e30f) // false for NaN axes (degenerate covariance): the reference's test then rejects nothing here
		{
			const float d = 0.01f + 1e-4f * (fabsf(px) + hx);
			const float lo = floorf((px - hx - d) * (1.0f / FR_TILE)) - 1.0f, hi = floorf((px + hx + d) * (1.0f / FR_TILE)) + 1.0f;
			w.x0 = max(w.x0, (int)fmaxf(lo, -1e6f)); w.x1 = min(w.x1, (int)fminf(hi, 1e6f));
		}
		if (hy < 1e30f)
		{
			const float d = 0.01f + 1e-4f * (fabsf(py) + hy);
			const float lo = floorf((py - hy - d) * (1.0f / FR_TILE)) - 1.0f, hi = floorf((py + hy + d) * (1.0f / FR_TILE)) + 1.0f;
			w.y0 = max(w.y0, (int)fmaxf(lo, -1e6f)); w.y1 = min(w.y1, (int)fminf(hi, 1e6f));
		}
	}
	if (FOV)
	{
		const int k = (int)fminf(fmaxf(ceilf(hl + 1.0f), 0.0f), 4.0f); // NaN -> 0: nothing passes `level < NaN`
		const uint4 b = lv_boxes[k * lv_box_stride];
		w.x0 = max(w.x0, gx - (int)b.x); w.y0 = max(w.y0, gy - (int)b.y);
		w.x1 = min(w.x1, (int)b.z); w.y1 = min(w.y1, (int)b.w);
	}
	if (w.x1 <= w.x0 || w.y1 <= w.y0) { w.x1 = w.x0; w.y1 = w.y0; w.tnum = 0; }
	else w.tnum = (uint32_t)(w.y1 - w.y0) * (uint32_t)(w.x1 - w.x0);
	return w;
}

// Oriented-bounding-box vs tile separating-axis test: RS auxiliary.h:66-154.
// The reference takes min/max over the four box corners (x, then y) and over the four tile corners projected on
// the two box axes. Rounding is monotone, so the min/max of fl(u_i - t) is fl(min/max u_i - t) and the min/max of
// fl(a_i + b_j) over all sign combinations is fl(min/max a + min/max b): the corner extremes are formed once per
// splat and each axis needs two products per coordinate instead of four dot products -- same bits, half the work.
struct Obb {
	float cx, cy;        // splat centre (pixels)
	float e1x, e1y, e2x, e2y, len1, len2;
	float vxmin, vxmax, vymin, vymax; // extremes of the box corners
};
__device__ __forceinline__ Obb make_obb(float cx, float cy, float4 ev, float2 el)
{
	Obb o; o.cx = cx; o.cy = cy; o.e1x = ev.x; o.e1y = ev.y; o.e2x = ev.z; o.e2y = ev.w; o.len1 = el.x; o.len2 = el.y;
	const float d1x = fabsf(o.len1 * o.e1x), d1y = fabsf(o.len1 * o.e1y), d2x = fabsf(o.len2 * o.e2x), d2y = fabsf(o.len2 * o.e2y);
	// corners are cx +- d1x +- d2x (RS auxiliary.h:75-86), summed in that order
	o.vxmax = cx + d1x + d2x; o.vxmin = cx - d1x - d2x;
	o.vymax = cy + d1y + d2y; o.vymin = cy - d1y - d2y;
	return o;
}
__device__ __forceinline__ bool obb_hits_tile(const Obb &o, int tx, int ty)
{
	const float tpx = (float)tx * (float)FR_TILE + (float)FR_TILE / 2.0f;
	const float tpy = (float)ty * (float)FR_TILE + (float)FR_TILE / 2.0f;
	if ((o.vxmax - tpx) < -8.0f || (o.vxmin - tpx) > 8.0f) return false;
	if ((o.vymax - tpy) < -8.0f || (o.vymin - tpy) > 8.0f) return false;
	// tile corners relative to the splat centre: x in {xp, xm}, y in {yp, ym}
	const float xp = tpx + 8.0f - o.cx, xm = tpx - 8.0f - o.cx, yp = tpy + 8.0f - o.cy, ym = tpy - 8.0f - o.cy;
	{
		const float ap = xp * o.e1x, am = xm * o.e1x, bp = yp * o.e1y, bm = ym * o.e1y;
		const float mn = fminf(ap, am) + fminf(bp, bm), mx = fmaxf(ap, am) + fmaxf(bp, bm);
		if (o.len1 < mn || -o.len1 > mx) return false;
	}
	{
		const float ap = xp * o.e2x, am = xm * o.e2x, bp = yp * o.e2y, bm = ym * o.e2y;
		const float mn = fminf(ap, am) + fminf(bp, bm), mx = fmaxf(ap, am) + fmaxf(bp, bm);
		if (o.len2 < mn || -o.len2 > mx) return false;
	}
	return true;
}

// ---- helpers shared by the forward and backward tile kernels ----
// Can the splat reach the pixel rectangle [X0, X1] x [Y0, Y1] at all? power = -q/2 with the convex quadratic
// q = A dx^2 + 2 B dx dy + C dy^2, whose minimum over a box not containing the centre lies on one of the four
// edges (1-D minimiser clamped to the edge). Returns false only if every pixel of the rectangle is certain to be
// skipped by the blend (power below `thr` by a margin far above float rounding); degenerate conics are kept.
// This is evaluated ONCE per staged entry and wave band by the lane that stages the entry, so that the waves
// iterate only over entries that can touch their band: the 3-sigma box of the binning stage also admits tiles
// that only the box corners reach, and a two-wave tile halves the footprint once more.
__device__ __forceinline__ bool splat_reaches(float gx, float gy, float A, float B, float C, float thr,
	float X0, float X1, float Y0, float Y1)
{
	const float ax = X0 - gx, bx = X1 - gx, ay = Y0 - gy, by = Y1 - gy;
	if (!(A > 0.0f && C > 0.0f && A * C - B * B > 0.0f) || !(ax == ax) || !(ay == ay)) return true;
	if (ax <= 0.0f && bx >= 0.0f && ay <= 0.0f && by >= 0.0f) return true;
	const float iA = 1.0f / A, iC = 1.0f / C;
	float qmin = 3.0e38f;
#pragma unroll
	for (int e = 0; e < 2; e++)
	{
		const float dx = e ? bx : ax;
		const float dy = fminf(fmaxf(-(B * dx) * iC, ay), by);
		qmin = fminf(qmin, (A * dx + 2.0f * B * dy) * dx + C * dy * dy);
		const float ey = e ? by : ay;
		const float ex = fminf(fmaxf(-(B * ey) * iA, ax), bx);
		qmin = fminf(qmin, (A * ex + 2.0f * B * ey) * ex + C * ey * ey);
	}
	const float mx = fmaxf(fabsf(ax), fabsf(bx)), my = fmaxf(fabsf(ay), fabsf(by));
	const float mag = A * mx * mx + C * my * my + 2.0f * fabsf(B) * mx * my;
	return !(-0.5f * qmin < thr - (2e-5f * mag + 1e-3f));
}

// a wave-uniform 64-bit value moved to scalar registers (loop control on it then runs on the scalar unit)
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
	return ((unsigned long long)hi << 32) | lo;
}

// Rows of a tile are dealt to the waves of its workgroup in contiguous bands (16 / waves rows each); inside a
// band a lane owns rows r, r + 4, ... of its column: a small splat then misses the other band's wave entirely.
// (Alternating 4-row strips balance the two waves better but let most splats touch both: measured slower.)
template <int PPL>
__device__ __forceinline__ int tile_row(int tid, int k)
{
	constexpr int NW = 256 / PPL / 64; // waves per tile
	return (tid >> 6) * (16 / NW) + ((tid >> 4) & 3) + 4 * k;
}
// can the splat touch any pixel that wave w of tile (tx, ty) owns?
template <int PPL>
__device__ __forceinline__ bool band_reaches(int w, int tx, int ty, float gx, float gy, float A, float B, float C, float thr)
{
	constexpr int NW = 256 / PPL / 64;
	const float Y0 = (float)(ty * FR_TILE + w * (16 / NW));
	return splat_reaches(gx, gy, A, B, C, thr, (float)(tx * FR_TILE), (float)(tx * FR_TILE + 15), Y0, Y0 + (float)(16 / NW - 1));
}

// SH basis constants (reference auxiliary.h:22-39)
#define FR_SH_C0 0.28209479177387814f
#define FR_SH_C1 0.4886025119029199f
#define FR_SH_C2_0 1.0925484305920792f
#define FR_SH_C2_1 -1.0925484305920792f
#define FR_SH_C2_2 0.31539156525252005f
#define FR_SH_C2_3 -1.0925484305920792f
#define FR_SH_C2_4 0.5462742152960396f
#define FR_SH_C3_0 -0.5900435899266435f
#define FR_SH_C3_1 2.890611442640554f
#define FR_SH_C3_2 -0.4570457994644658f
#define FR_SH_C3_3 0.3731763325901154f
#define FR_SH_C3_4 -0.4570457994644658f
#define FR_SH_C3_5 1.445305721320277f
#define FR_SH_C3_6 -0.5900435899266435f

// ---- host-side launch plumbing ------------------------------------------------------------
void set_error(const char *fmt, ...);
int check_launch(const char *what, hipStream_t stream, bool debug);

// the lanes of ONE wave exchange data through LDS: writes before, reads after
#define FR_WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// ---- cross-row folds of gfx950 (k_render_bwd's nine gradient sums, k_render's contribution sums)
typedef unsigned int bwd_u2 __attribute__((ext_vector_type(2)));
// lanes 0-31 get a[l] + a[l + 32], lanes 32-63 get b[l - 32] + b[l]  (tools/scratch/permlane_test.hip)
__device__ __forceinline__ float fold32(float a, float b)
{
	const bwd_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
	return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// rows of 16 lanes: (a0 + a1, b0 + b1, a2 + a3, b2 + b3)
__device__ __forceinline__ float fold16(float a, float b)
{
	const bwd_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
	return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// sum over each row of 16 lanes, in all its lanes
__device__ __forceinline__ float row_sum16(float x)
{
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false)); // row_half_mirror
	x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false)); // row_mirror
	return x;
}

// GaussianModel's activations (scene/gaussian_model.py:200-240) as the kernels apply them: k_activate_fwd as a pass of its
// own, k_project / k_bin on the fly when the caller hands over raw parameters (fr_forward_args.raw_activations).
__device__ __forceinline__ float act_scale(float raw) { return expf(raw); }
__device__ __forceinline__ float act_opacity(float raw) { return 1.0f / (1.0f + expf(-raw)); }
__device__ __forceinline__ float4 act_rotation(float4 v, float *inv_norm = nullptr)
{
	const float n = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
	const float d = fmaxf(n, 1e-12f);
	// for the backward pass: 1 / |v|, negative (-1e12) where the denominator was clamped (no projection term there)
	if (inv_norm) *inv_norm = n > 1e-12f ? 1.0f / n : -1e12f;
	return make_float4(v.x / d, v.y / d, v.z / d, v.w / d);
}

// stage launchers (implemented in the .hip files)
struct FwdCtx {
	fr_forward_args *a;
	hipStream_t stream;
	int gx, gy, T;
	int fov_split;      // RF: the two level states of a two-level tile go to different waves (out_color was zero-filled)
	int bin_wgs;        // workgroups k_bin ran with (k_emit replays the same number)
	int hist_mode;      // k_bin / k_emit: 0 = global tile counters, 1 = LDS histogram of 32-bit counts, 2 = of 16-bit counts (launch_bin decides)
	int scan_fused;     // the tile scan ran as the tail of k_bin (launch_bin decides): no k_tile_scan launch
	int regions_ok;               // ... and no binning workgroup's segment overflowed (the host's copy of slab_ctr[5], read with the totals)
	int regions, region_rx, wcap; // region-major emission (launch_bin decides): regions of the tile grid (0 = off), regions per row, entries of a workgroup's segment
	int heavy4, heavy2; // tiles with >= 2048 / 512..2047 instances (leading entries of tile_order)
	int heavy8;         // tiles with >= 4096 instances
	int n_items;        // entries of ImageWS::render_items
	int proj_waves, proj_cpw; // the cull pass's grid in waves and the consecutive chunks each wave took
	int64_t capacity;   // instances the binning workspace was carved for
	int64_t items_cap;  // upper bound of the blend work items (the kernels' defensive bound checks)
	uint32_t *totals_host_dev; // device address of the host's pinned copy of totals[4] (+ sequence word), or null
	uint32_t totals_seq;       // this frame's sequence number for that word
	float focal_x, focal_y;
	GeomWS geom;
	ImageWS img;
	BinWS bin;
};
struct AuxStream { int device = -1; hipStream_t main = nullptr, s, s2; hipEvent_t fork, fork2, join, join2; bool ok = false; };
AuxStream *aux_stream(hipStream_t main); // helper streams of the calling host thread for work launched on `main` (binning.hip)
int launch_tile_levels(FwdCtx &c);
int launch_pack_geom(int P, const float *means3D, const float *scales, const float *rotations, const float *opacities, int levels,
	const float *highest_levels, float *out, hipStream_t stream);
int launch_pack_cull(int P, const float *means3D, const float *scales, const float *rotations, float *out, hipStream_t stream);
int launch_pack_colour(int P, const float *shs, const float *shs_rest, const float *shs_dcs, float *out, hipStream_t stream);
int launch_l1_ssim_forward(int C, int H, int W, const float *x, const float *y, float *dmaps, float *partials, hipStream_t stream);
int launch_l1_ssim_finish(int nblocks, double n, const float *partials, float lam, float *out3, hipStream_t stream);
int launch_l1_ssim_backward(int C, int H, int W, const float *x, const float *y, const float *dmaps, float w_l1, float w_ssim,
	const float *grad_scale, float *dL_dx, hipStream_t stream);
int launch_activate_forward(int P, const float *rs, const float *rq, const float *ro, float *s, float *q, float *o, hipStream_t stream);
int launch_activate_backward(int P, const float *rs, const float *rq, const float *ro, const float *gs, const float *gq, const float *go,
	float *ds, float *dq, float *dop, hipStream_t stream);
int launch_project(FwdCtx &c); // cull pass + the ordered compaction of its survivors
int launch_bin(FwdCtx &c);    // projection of the cull pass's survivors, tile counts, colours, item rows
int launch_tile_scan(FwdCtx &c);
int launch_emit(FwdCtx &c);
int launch_tile_sort(FwdCtx &c);
int launch_render(FwdCtx &c);
int launch_backward(const fr_backward_args *a);
// whole: clear every tensor in full (fr_backward_prefill: the caller's zeros come before a backward call that is told about them);
// else the narrow tensors only where no 32-row group holds a visible Gaussian (the rest of them is k_preprocess_bwd's, backward.hip)
int launch_gradient_fill(const fr_backward_args *a, hipStream_t fs, bool events, bool whole);
int launch_mark_visible(int P, const float *means3D, const float *vm, uint8_t *present, hipStream_t s);

} // namespace fr
