// extern "C" entry points of libfovraster_hip.so (see include/fovraster.h for the contract and
// for the reference interfaces each one replaces).
#include "common.h"
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>

namespace fr {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

int check_launch(const char *what, hipStream_t stream, bool debug)
{
	hipError_t e = hipGetLastError();
	static const bool trace = getenv("FR_TRACE") != nullptr;
	if (trace) { fprintf(stderr, "[fovraster] launched %s\n", what); fflush(stderr); }
	if (e == hipSuccess && debug) e = hipStreamSynchronize(stream);
	if (trace && debug) { fprintf(stderr, "[fovraster] %s done (%d)\n", what, (int)e); fflush(stderr); }
	if (e != hipSuccess)
	{
		set_error("%s: %s", what, hipGetErrorString(e));
		return FR_ERR_HIP;
	}
	return FR_OK;
}

#define FR_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s: %s", #call, hipGetErrorString(e_)); return FR_ERR_HIP; } } while (0)

static int validate_forward(const fr_forward_args *a)
{
	if (!a) { set_error("null args"); return FR_ERR_INVALID; }
	if (a->variant < FR_VARIANT_ORIGINAL || a->variant > FR_VARIANT_MMFR_PCHECK_OBB) { set_error("unknown variant %d", a->variant); return FR_ERR_INVALID; }
	if (a->P < 0 || a->W <= 0 || a->H <= 0) { set_error("bad sizes P=%d W=%d H=%d", a->P, a->W, a->H); return FR_ERR_INVALID; }
	if (a->P > (1 << 30) || a->W > 16 * 65535 || a->H > 16 * 65535) { set_error("too large: P=%d (max 2^30) W=%d H=%d (max 65535 tiles per axis)", a->P, a->W, a->H); return FR_ERR_INVALID; }
	if ((int64_t)((a->W + FR_TILE - 1) / FR_TILE) * ((a->H + FR_TILE - 1) / FR_TILE) >= (1 << 29)) { set_error("too many tiles (W=%d H=%d)", a->W, a->H); return FR_ERR_INVALID; }
	if (!a->out_color) { set_error("out_color is null"); return FR_ERR_INVALID; }
	if (a->P == 0) return FR_OK;
	if (!a->means3D || !a->opacities || !a->viewmatrix || !a->projmatrix || !a->campos || !a->background || !a->radii)
	{ set_error("a required pointer is null"); return FR_ERR_INVALID; }
	if (!a->geometry_resize || !a->binning_resize || !a->image_resize) { set_error("resize callbacks are required"); return FR_ERR_INVALID; }
	const bool fov = a->variant == FR_VARIANT_FOV_PCHECK_OBB;
	if (fov)
	{
		if (a->shs_rest) { set_error("shs_rest is not used by the foveated variant (its shs already is the rest part)"); return FR_ERR_INVALID; }
		if (!a->shs || !a->shs_dcs || !a->highest_levels) { set_error("foveated variant needs shs (rest), shs_dcs and highest_levels"); return FR_ERR_INVALID; }
		if (a->M != 15) { set_error("foveated variant expects M=15 rest coefficients, got %d", a->M); return FR_ERR_INVALID; }
	}
	else if (a->variant == FR_VARIANT_NAIVE_FOV_PCHECK_OBB || a->variant == FR_VARIANT_MMFR_PCHECK_OBB)
	{
		if (!a->shs || a->colors_precomp || a->shs_rest || !a->highest_levels) { set_error("the shared-model foveated variant needs shs [P,M,3] and highest_levels (no colors_precomp / shs_rest)"); return FR_ERR_INVALID; }
		if (a->M < (a->D + 1) * (a->D + 1)) { set_error("M=%d too small for SH degree %d", a->M, a->D); return FR_ERR_INVALID; }
		if (a->packed_geom || a->packed_colour || a->packed_cull) { set_error("the shared-model foveated variant has no packed layout"); return FR_ERR_INVALID; }
	}
	else
	{
		if ((a->shs == nullptr) == (a->colors_precomp == nullptr)) { set_error("provide exactly one of shs / colors_precomp"); return FR_ERR_INVALID; }
		if (a->shs && a->M < (a->D + 1) * (a->D + 1)) { set_error("M=%d too small for SH degree %d", a->M, a->D); return FR_ERR_INVALID; }
		if (a->shs_rest && (!a->shs || a->M < 2)) { set_error("shs_rest needs shs (the DC part) and M >= 2"); return FR_ERR_INVALID; }
	}
	if (a->D < 0 || a->D > 3) { set_error("SH degree %d not in 0..3", a->D); return FR_ERR_INVALID; }
	const bool has_sr = a->scales && a->rotations;
	if (has_sr == (a->cov3D_precomp != nullptr)) { set_error("provide exactly one of scales+rotations / cov3D_precomp"); return FR_ERR_INVALID; }
	if ((a->packed_geom != nullptr) != (a->packed_colour != nullptr) || (a->packed_geom != nullptr) != (a->packed_cull != nullptr))
	{ set_error("packed_geom, packed_colour and packed_cull go together"); return FR_ERR_INVALID; }
	if (a->packed_geom && (!has_sr || !a->shs || a->colors_precomp || a->M != (fov ? 15 : 16)))
	{ set_error("the packed model needs scales + rotations and shs with all 16 coefficients (M=%d)", a->M); return FR_ERR_INVALID; }
	if (a->raw_activations && (is_fov(a->variant) || !has_sr || a->packed_geom))
	{ set_error("raw_activations: plain variants with scales + rotations, without the packed layout"); return FR_ERR_INVALID; }
	if (a->no_stats && a->variant != FR_VARIANT_PCHECK_OBB_SUM) { set_error("no_stats: pcheck_obb_sum only"); return FR_ERR_INVALID; }
	if (has_stats(a->variant) && !a->no_stats && (!a->gaussians_count || !a->contributions)) { set_error("this variant needs gaussians_count and contributions"); return FR_ERR_INVALID; }
	if (a->variant == FR_VARIANT_PCHECK_OBB_LWMC && !a->loss_map) { set_error("pcheck_obb_loss_weighted_max_count needs loss_map"); return FR_ERR_INVALID; }
	return FR_OK;
}

} // namespace fr

using namespace fr;

extern "C" {

int fr_abi_version(void) { return FR_ABI_VERSION; }

void *fr_event_create(void)
{
	hipEvent_t e;
	if (hipEventCreate(&e) != hipSuccess) { set_error("hipEventCreate failed"); return nullptr; }
	return (void *)e;
}
void fr_event_destroy(void *event) { if (event) (void)hipEventDestroy((hipEvent_t)event); }
int fr_event_elapsed_ms(void *start, void *stop, float *ms)
{
	if (!start || !stop || !ms) { set_error("null event"); return FR_ERR_INVALID; }
	// (an event that was never recorded is the caller's mistake, not a device fault: the error is reported here and not left
	// behind as the runtime's sticky "last error" for whoever calls into HIP next)
	hipError_t e = hipEventSynchronize((hipEvent_t)stop);
	if (e == hipSuccess) e = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
	if (e != hipSuccess) { (void)hipGetLastError(); set_error("event timing: %s (was the event recorded?)", hipGetErrorString(e)); return FR_ERR_HIP; }
	return FR_OK;
}
const char *fr_last_error(void) { return g_err; }

size_t fr_geometry_bytes(int32_t variant, int32_t P) { return carve_geom(variant, (size_t)P, nullptr).bytes; }
size_t fr_image_bytes(int32_t variant, int32_t W, int32_t H) { return carve_image(variant, W, H, nullptr).bytes; }
size_t fr_binning_bytes(int32_t variant, int64_t n) { (void)variant; return carve_bin(n, nullptr).bytes; }

const uint32_t *fr_image_ranges(int32_t variant, int32_t W, int32_t H, const char *image) { return (const uint32_t *)carve_image(variant, W, H, (char *)image).ranges; }
const uint32_t *fr_binning_point_list(int32_t variant, int64_t n, const char *binning) { (void)variant; return carve_bin(n, (char *)binning).point_list; }
const float *fr_image_final_T(int32_t variant, int32_t W, int32_t H, const char *image) { return carve_image(variant, W, H, (char *)image).final_T; }
const uint32_t *fr_image_n_contrib(int32_t variant, int32_t W, int32_t H, const char *image) { return carve_image(variant, W, H, (char *)image).n_contrib; }
const float *fr_geometry_records(int32_t variant, int32_t P, const char *geometry) { return (const float *)carve_geom(variant, (size_t)P, (char *)geometry).rec; }
const uint32_t *fr_geometry_vis_list(int32_t variant, int32_t P, const char *geometry) { return carve_geom(variant, (size_t)P, (char *)geometry).vis_list; }
const uint32_t *fr_geometry_vis_count(int32_t variant, int32_t P, const char *geometry) { return carve_geom(variant, (size_t)P, (char *)geometry).slab_ctr + 1; }
const float *fr_geometry_walk_records(int32_t variant, int32_t P, const char *geometry) { return (const float *)carve_geom(variant, (size_t)P, (char *)geometry).wrec; }
const float *fr_geometry_level_colours(int32_t P, const char *geometry) { return (const float *)carve_geom(FR_VARIANT_FOV_PCHECK_OBB, (size_t)P, (char *)geometry).lvl; }
const uint32_t *fr_geometry_level_ranges(int32_t P, const char *geometry) { return carve_geom(FR_VARIANT_FOV_PCHECK_OBB, (size_t)P, (char *)geometry).lrange; }
const float *fr_image_tile_levels(int32_t W, int32_t H, const char *image) { return carve_image(FR_VARIANT_FOV_PCHECK_OBB, W, H, (char *)image).tile_lv; }

// ---- the forward call ---------------------------------------------------------------------------
// A frame is enqueued in two halves around its ONE host synchronisation (the instance count, which sizes the binning
// workspace -- the reference synchronises twice for it, rasterizer_impl.cu:281 and RS :401,422):
//   fr_forward_begin    workspaces (geometry, image), fills, tile levels, cull pass, projection, tile counts, tile scan
//   fr_forward_finish   waits for the count, binning workspace, emission, per-tile sort, colours, blend
// fr_forward is begin + finish. A host that keeps TWO frames in flight (two streams, two workspace sets) calls
// begin(n + 1) before finish(n): the latency-bound head of one frame then runs beside the sort and the blend of the other.
} // extern "C"

namespace fr {

// 64 bytes of pinned, device-mapped host memory per frame in flight: the tile scan writes the frame's totals and its sequence
// number there, the host polls (a copy command after the kernel costs ~10 us more of an idle GPU). One pool per PROCESS behind a
// mutex (a frame handle may be dropped by another thread than the one that made it -- a garbage collector's finalizer -- or after
// its thread has gone); the device address of mapped host memory belongs to the device that was current when it was asked for.
struct PinnedBlock
{
	uint32_t *host = nullptr, *dev = nullptr; int device = -1; bool busy = false;
	hipEvent_t quarantine = nullptr; bool quarantined = false; // a DROPPED frame's tile scan may still be on its way to the block
};
static std::mutex g_pinned_mu;
static PinnedBlock g_pinned_pool[64];
static PinnedBlock *take_pinned(int device)
{
	std::lock_guard<std::mutex> lock(g_pinned_mu);
	PinnedBlock *spare = nullptr;
	for (PinnedBlock &b : g_pinned_pool)
	{
		if (b.busy) continue;
		if (b.quarantined)
		{
			if (hipEventQuery(b.quarantine) != hipSuccess) { (void)hipGetLastError(); continue; }
			b.quarantined = false;
		}
		if (b.host && b.device == device) { b.busy = true; return &b; }
		if (!spare) spare = &b;
	}
	if (!spare) return nullptr; // every block of the pool in flight: the frame falls back to a copy + stream wait
	if (spare->host) (void)hipHostFree(spare->host);
	spare->host = spare->dev = nullptr; spare->device = device;
	void *h = nullptr, *d = nullptr;
	if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess)
	{ spare->host = (uint32_t *)h; spare->dev = (uint32_t *)d; for (int i = 0; i < 16; i++) spare->host[i] = 0; }
	else if (h) (void)hipHostFree(h);
	(void)hipGetLastError();
	if (!spare->host) return nullptr;
	spare->busy = true;
	return spare;
}
// A frame gives its block back. unread: its tile scan was enqueued and the host never read the totals (an error return between the
// scan and the count, fr_forward_abandon, a handle dropped by a finalizer): the write may still be on its way, so the block is not
// handed to another frame before an event on the frame's stream has passed.
static void release_pinned(PinnedBlock *b, bool unread, hipStream_t stream)
{
	if (!b) return;
	bool drained = !unread;
	if (unread)
	{
		std::lock_guard<std::mutex> lock(g_pinned_mu);
		if (!b->quarantine && hipEventCreateWithFlags(&b->quarantine, hipEventDisableTiming) != hipSuccess) b->quarantine = nullptr;
		if (b->quarantine && hipEventRecord(b->quarantine, stream) == hipSuccess) { b->quarantined = true; b->busy = false; return; }
		(void)hipGetLastError();
	}
	if (!drained) (void)hipStreamSynchronize(stream); // (no event to be had: wait the write out)
	std::lock_guard<std::mutex> lock(g_pinned_mu);
	b->busy = false;
}

// what the next frame of a kind (variant, P, W, H) asks its binning workspace for before its count is in
struct Guess { int32_t variant = -1, P = 0, W = 0, H = 0; int64_t capacity = 0; };
static Guess &guess_for(int variant) { static thread_local Guess guesses[8]; return guesses[variant & 7]; }

} // namespace fr

struct fr_frame
{
	fr_forward_args *a = nullptr;
	FwdCtx c;
	AuxStream *ax = nullptr;       // the frame's helper streams (null: everything on the launch stream)
	bool aux_pending = false;      // work on ax->s2 that the launch stream has not waited for yet
	PinnedBlock *pin = nullptr;
	bool scan_enqueued = false;    // the kernel that writes the pinned totals block is on the stream ...
	bool totals_read = false;      // ... and the host has read what it wrote
	bool empty = false;            // P == 0: nothing left to do
	// every exit that abandons the frame joins the helper stream first: the caller may free or reuse out_color / the
	// statistics arrays / the workspaces as soon as the call has returned
	void join_aux() { if (aux_pending && ax) (void)hipStreamWaitEvent(c.stream, ax->join2, 0); aux_pending = false; }
	~fr_frame() { join_aux(); release_pinned(pin, scan_enqueued && !totals_read, c.stream); }
};

extern "C" {

int fr_forward_begin(fr_forward_args *a, fr_frame **out)
{
	if (!out) { set_error("null frame handle"); return FR_ERR_INVALID; }
	*out = nullptr;
	int rc = validate_forward(a);
	if (rc) return rc;
	hipStream_t stream = (hipStream_t)a->stream;
	a->num_rendered = 0;
	a->max_tile_instances = 0;
	a->num_candidates = 0;
	fr_frame *f = new (std::nothrow) fr_frame;
	if (!f) { set_error("out of host memory"); return FR_ERR_ALLOC; }
	struct Guard { fr_frame *f; ~Guard() { delete f; } } guard = { f }; // (released at the end: an early return drops the frame)
	f->a = a;
	FwdCtx &c = f->c;
	c.a = a; c.stream = stream;
	if (a->P == 0)
	{
		// reference: RasterizeGaussiansCUDA returns the zero-initialised image when P == 0
		FR_HIP(hipMemsetAsync(a->out_color, 0, sizeof(float) * 3 * (size_t)a->W * a->H, stream));
		f->empty = true;
		*out = f; guard.f = nullptr;
		return FR_OK;
	}
	// optional per-stage timing: record the caller's events on the launch stream (no sync here)
	auto mark = [&](int i) { if (a->stage_events && a->stage_events[i]) (void)hipEventRecord((hipEvent_t)a->stage_events[i], stream); };
	c.gx = (a->W + FR_TILE - 1) / FR_TILE; c.gy = (a->H + FR_TILE - 1) / FR_TILE; c.T = c.gx * c.gy;
	c.focal_y = a->H / (2.0f * a->tanfovy);
	c.focal_x = a->W / (2.0f * a->tanfovx);

	char *gptr = a->geometry_resize(a->resize_user[0], carve_geom(a->variant, (size_t)a->P, nullptr).bytes);
	char *iptr = a->image_resize(a->resize_user[2], carve_image(a->variant, a->W, a->H, nullptr).bytes);
	if (!gptr || !iptr) { set_error("geometry/image resize callback returned null"); return FR_ERR_ALLOC; }
	c.geom = carve_geom(a->variant, (size_t)a->P, gptr);
	c.img = carve_image(a->variant, a->W, a->H, iptr);

	// RF: the two level states of a two-level tile are blended by different waves, which ADD their halves to the image
	// (clearing only those tiles inside k_tile_levels tripled that kernel: 11 -> 32 us; the fill command is 6 us)
	c.fov_split = a->variant == FR_VARIANT_FOV_PCHECK_OBB ? 1 : 0;
	// The frame's helper stream (s2 of the pair that belongs to this thread and launch stream) carries the large fills -- the
	// image (RF) and the training variants' two statistics arrays (7 + 6 us at the head of a 1080p foveated frame when they ran in
	// front of it) -- beside the cull / binning kernels; only the blend kernel at the END of the frame needs them and waits
	// (fr_forward_finish).
	f->ax = (!a->debug && !a->no_helper_streams) ? aux_stream(stream) : nullptr;
	hipStream_t fill_stream = stream;
	const bool stats = has_stats(a->variant) && !a->no_stats;
	if (f->ax && (c.fov_split || stats))
	{
		if (hipEventRecord(f->ax->fork, stream) != hipSuccess || hipStreamWaitEvent(f->ax->s2, f->ax->fork, 0) != hipSuccess) { (void)hipGetLastError(); f->ax = nullptr; }
		else fill_stream = f->ax->s2;
	}
	auto fills_done = [&]() { if (fill_stream != stream && hipEventRecord(f->ax->join2, f->ax->s2) == hipSuccess) f->aux_pending = true; };
	if (c.fov_split)
	{
		const hipError_t e = hipMemsetAsync(a->out_color, 0, sizeof(float) * 3 * (size_t)a->W * a->H, fill_stream);
		fills_done();
		if (e != hipSuccess) { set_error("hipMemsetAsync(out_color): %s", hipGetErrorString(e)); return FR_ERR_HIP; }
	}
	if (stats)
	{
		const hipError_t e1 = hipMemsetAsync(a->gaussians_count, 0, sizeof(int32_t) * (size_t)a->P, fill_stream);
		const hipError_t e2 = hipMemsetAsync(a->contributions, 0, sizeof(float) * (size_t)a->P, fill_stream);
		fills_done();
		if (e1 != hipSuccess || e2 != hipSuccess) { set_error("hipMemsetAsync(statistics): %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); return FR_ERR_HIP; }
	}
	// small clears in front of the first kernel: the per-tile instance counters (k_bin's workgroups add their shares to them;
	// the global-atomics path of huge tile grids counts in them directly) and, next to them, the level boxes (RF:
	// k_tile_levels raises them with atomicMax; it clears the slab counters itself)
	FR_HIP(hipMemsetAsync(c.img.tile_count, 0, (size_t)((char *)(c.img.lv_bbox + 5 * FR_LV_BBOX_STRIDE) - (char *)c.img.tile_count), stream)); // + lv_bbox
	if (!is_fov(a->variant)) // RF: k_tile_levels clears them
		FR_HIP(hipMemsetAsync(c.geom.slab_ctr, 0, FR_SLAB_CTR_WORDS * sizeof(uint32_t), stream));
	if (a->list_consumed) FR_HIP(hipMemsetAsync(a->list_consumed, 0, sizeof(uint32_t) * (size_t)c.T, stream));
	if (a->blend_pairs) FR_HIP(hipMemsetAsync(a->blend_pairs, 0, sizeof(uint32_t) * (size_t)c.T, stream));
	static thread_local uint32_t frame_seq = 0; // this frame's tag: the totals block's sequence word
	if (++frame_seq == 0) frame_seq = 1;
	c.totals_seq = frame_seq;
	mark(FR_STAGE_TILE_LEVELS);
	if (is_fov(a->variant)) { rc = launch_tile_levels(c); if (rc) return rc; }
	mark(FR_STAGE_PROJECT);
	rc = launch_project(c); if (rc) return rc;
	int cur_dev = 0;
	(void)hipGetDevice(&cur_dev);
	f->pin = a->debug ? nullptr : take_pinned(cur_dev);
	c.totals_host_dev = f->pin ? f->pin->dev : nullptr;
	c.scan_fused = 0;
	mark(FR_STAGE_BIN);
	f->scan_enqueued = true; // (from here on a dropped frame quarantines its block: ~fr_frame)
	rc = launch_bin(c); if (rc) return rc;
	mark(FR_STAGE_TILE_SCAN);
	if (!c.scan_fused) { rc = launch_tile_scan(c); if (rc) return rc; } // (else: k_bin's last workgroup did it)
	*out = f; guard.f = nullptr;
	return FR_OK;
}

int fr_forward_finish(fr_frame *f)
{
	if (!f) { set_error("null frame"); return FR_ERR_INVALID; }
	struct Guard { fr_frame *f; ~Guard() { delete f; } } guard = { f }; // the handle is released whatever happens
	if (f->empty) return FR_OK;
	fr_forward_args *a = f->a;
	FwdCtx &c = f->c;
	hipStream_t stream = c.stream;
	auto mark = [&](int i) { if (a->stage_events && a->stage_events[i]) (void)hipEventRecord((hipEvent_t)a->stage_events[i], stream); };
	// Everything behind the tile scan needs the frame's counts: the number of instances D (size of the binning workspace)
	// and the sort / blend class counts. They are on their way to the frame's pinned block, which the host polls: the
	// numbers land a few microseconds before k_tile_scan retires, so the launches that follow reach the queue while it still
	// runs and the GPU does not wait for the host.
	// The binning workspace is asked for BEFORE the count is in -- sized like the largest frame of this kind so far plus a
	// quarter -- while the GPU is still binning: the callback (the caller's allocator, a Python function behind ctypes in the
	// reference-shaped host code) is then off the critical path between the tile scan and k_emit. A frame that does not fit
	// asks again with its real size (the callback may therefore be called twice per frame; the last answer is the one in use).
	Guess &gs = guess_for(a->variant);
	const bool same_kind = gs.variant == a->variant && gs.P == a->P && gs.W == a->W && gs.H == a->H;
	char *early_bin = nullptr;
	const int64_t early_cap = (same_kind && !a->debug) ? gs.capacity : 0;
	if (early_cap > 0) early_bin = a->binning_resize(a->resize_user[1], carve_bin(early_cap, nullptr, c.T).bytes);

	uint32_t totals[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, candidates = 0, region_overflow = 0;
	uint32_t *const pinned = f->pin ? f->pin->host : nullptr;
	if (!pinned)
	{
		FR_HIP(hipMemcpyAsync(totals, c.img.totals, sizeof(totals), hipMemcpyDeviceToHost, stream));
		FR_HIP(hipMemcpyAsync(&candidates, c.geom.slab_ctr + 1, sizeof(candidates), hipMemcpyDeviceToHost, stream));
		FR_HIP(hipMemcpyAsync(&region_overflow, c.geom.slab_ctr + 5, sizeof(region_overflow), hipMemcpyDeviceToHost, stream));
		FR_HIP(hipStreamSynchronize(stream));
	}
	else
	{
		// poll the sequence word instead of sleeping on the stream: the numbers arrive a few microseconds before
		// the kernel retires, and the wake-up latency of a stream wait is saved. The stream is queried now and then
		// so that a failed launch cannot hang the caller.
		const volatile uint32_t *v = pinned;
		for (uint32_t spins = 1; __atomic_load_n(&pinned[4], __ATOMIC_ACQUIRE) != c.totals_seq; spins++)
			if ((spins & 0xffff) == 0)
			{
				const hipError_t q = hipStreamQuery(stream);
				if (q == hipErrorNotReady) continue;
				if (q != hipSuccess) { set_error("hip: %s", hipGetErrorString(q)); return FR_ERR_HIP; }
				if (__atomic_load_n(&pinned[4], __ATOMIC_ACQUIRE) != c.totals_seq) { set_error("tile scan did not publish its totals"); return FR_ERR_HIP; }
			}
		for (int i = 0; i < 4; i++) totals[i] = v[i];
		totals[5] = v[5]; totals[6] = v[6]; totals[7] = v[7]; candidates = v[8]; region_overflow = v[9];
	}
	c.regions_ok = region_overflow == 0 ? 1 : 0;
	f->totals_read = true;
	// reference auxiliary.h:156-160: a point behind the near plane although the caller said the cloud was prefiltered
	// (there: printf + __trap, which kills the context; here an error code)
	if (a->prefiltered && totals[7] != 0)
	{ set_error("Point is filtered although prefiltered is set. This shouldn't happen!"); return FR_ERR_PREFILTERED; }
	if (totals[0] > 0x7fffffffu) { set_error("too many instances (%u)", totals[0]); return FR_ERR_INVALID; }
	a->num_rendered = (int32_t)totals[0];
	a->max_tile_instances = (int32_t)totals[1];
	a->num_candidates = (int32_t)candidates;
	c.heavy4 = (int)totals[2]; c.heavy2 = (int)totals[3];
	c.n_items = (int)totals[5]; c.heavy8 = (int)totals[6];
	// what the next frame of this kind will ask for: a quarter of headroom over the largest frame seen
	const int64_t want = (int64_t)totals[0] + (int64_t)totals[0] / 4 + 65536;
	if (!same_kind) { gs.variant = a->variant; gs.P = a->P; gs.W = a->W; gs.H = a->H; gs.capacity = 0; }
	if (want > gs.capacity) gs.capacity = want;

	const bool fits_early = early_bin && (int64_t)totals[0] <= early_cap;
	const int64_t capacity = fits_early ? early_cap : (int64_t)totals[0];
	char *bptr = fits_early ? early_bin : a->binning_resize(a->resize_user[1], carve_bin(capacity, nullptr, c.T).bytes);
	if (!bptr) { set_error("binning resize callback returned null"); return FR_ERR_ALLOC; }
	c.bin = carve_bin(capacity, bptr, c.T);
	c.capacity = capacity;
	c.items_cap = (int64_t)(c.fov_split ? 4 : 2) * c.T; // two bands per tile, twice that for an RF two-level tile
	int rc = FR_OK;
	mark(FR_STAGE_EMIT);
	if (a->num_rendered > 0) { rc = launch_emit(c); if (rc) return rc; }
	mark(FR_STAGE_TILE_SORT);
	if (a->num_rendered > 0) { rc = launch_tile_sort(c); if (rc) return rc; }
	mark(FR_STAGE_RENDER);
	f->join_aux(); // the fills of the frame's head
	rc = launch_render(c);
	mark(FR_NUM_STAGES);
	return rc;
}

int fr_forward_abandon(fr_frame *f)
{
	if (!f) { set_error("null frame"); return FR_ERR_INVALID; }
	delete f; // ~fr_frame joins the helper stream and quarantines the pinned block (the frame's tile scan may not have written its totals yet)
	return FR_OK;
}

int fr_forward(fr_forward_args *a)
{
	fr_frame *f = nullptr;
	const int rc = fr_forward_begin(a, &f);
	if (rc) return rc;
	return fr_forward_finish(f);
}

int fr_pack_geom(int32_t P, const float *means3D, const float *scales, const float *rotations, const float *opacities,
	int32_t levels, const float *highest_levels, float *packed_geom, void *stream)
{
	if (P < 0 || levels < 1 || levels > 4 || (P > 0 && (!means3D || !scales || !rotations || !opacities || !packed_geom)))
	{ set_error("bad pack_geom arguments"); return FR_ERR_INVALID; }
	if (P == 0) return FR_OK;
	return launch_pack_geom(P, means3D, scales, rotations, opacities, levels, highest_levels, packed_geom, (hipStream_t)stream);
}

int fr_pack_cull(int32_t P, const float *means3D, const float *scales, const float *rotations, float *packed_cull, void *stream)
{
	if (P < 0 || (P > 0 && (!means3D || !scales || !rotations || !packed_cull))) { set_error("bad pack_cull arguments"); return FR_ERR_INVALID; }
	if (P == 0) return FR_OK;
	return launch_pack_cull(P, means3D, scales, rotations, packed_cull, (hipStream_t)stream);
}

int fr_pack_colour(int32_t P, const float *shs, const float *shs_rest, const float *shs_dcs, float *packed_colour, void *stream)
{
	if (P < 0 || (P > 0 && (!shs || !packed_colour)) || (shs_rest && shs_dcs)) { set_error("bad pack_colour arguments"); return FR_ERR_INVALID; }
	if (P == 0) return FR_OK;
	return launch_pack_colour(P, shs, shs_rest, shs_dcs, packed_colour, (hipStream_t)stream);
}

int fr_activate_forward(int32_t P, const float *raw_scaling, const float *raw_rotation, const float *raw_opacity, float *scaling, float *rotation,
	float *opacity, void *stream)
{
	if (P < 0 || (P > 0 && (!raw_scaling || !raw_rotation || !raw_opacity || !scaling || !rotation || !opacity))) { set_error("bad activate_forward arguments"); return FR_ERR_INVALID; }
	if (P == 0) return FR_OK;
	return launch_activate_forward(P, raw_scaling, raw_rotation, raw_opacity, scaling, rotation, opacity, (hipStream_t)stream);
}

int fr_activate_backward(int32_t P, const float *raw_scaling, const float *raw_rotation, const float *raw_opacity, const float *dL_dscaling,
	const float *dL_drotation, const float *dL_dopacity, float *dL_draw_scaling, float *dL_draw_rotation, float *dL_draw_opacity, void *stream)
{
	if (P < 0 || (P > 0 && (!raw_scaling || !raw_rotation || !raw_opacity || !dL_draw_scaling || !dL_draw_rotation || !dL_draw_opacity)))
	{ set_error("bad activate_backward arguments"); return FR_ERR_INVALID; }
	if (P == 0) return FR_OK;
	return launch_activate_backward(P, raw_scaling, raw_rotation, raw_opacity, dL_dscaling, dL_drotation, dL_dopacity, dL_draw_scaling, dL_draw_rotation,
		dL_draw_opacity, (hipStream_t)stream);
}

int64_t fr_l1_ssim_blocks(int32_t C, int32_t H, int32_t W)
{
	if (C <= 0 || H <= 0 || W <= 0) return 0;
	return (int64_t)C * ((H + 31) / 32) * ((W + 15) / 16); // one (sum |x - y|, sum ssim) pair per 16 x 32 tile, see loss.hip
}

int fr_l1_ssim_forward(int32_t C, int32_t H, int32_t W, const float *img, const float *target, float *dmaps, float *partials, void *stream)
{
	if (C <= 0 || H <= 0 || W <= 0 || C > 65535 || !img || !target || !partials) { set_error("bad l1_ssim_forward arguments"); return FR_ERR_INVALID; }
	return launch_l1_ssim_forward(C, H, W, img, target, dmaps, partials, (hipStream_t)stream);
}

int fr_l1_ssim_finish(int32_t C, int32_t H, int32_t W, const float *partials, float lambda_dssim, float *out3, void *stream)
{
	if (C <= 0 || H <= 0 || W <= 0 || C > 65535 || !partials || !out3) { set_error("bad l1_ssim_finish arguments"); return FR_ERR_INVALID; }
	return launch_l1_ssim_finish((int)fr_l1_ssim_blocks(C, H, W), (double)C * H * W, partials, lambda_dssim, out3, (hipStream_t)stream);
}

int fr_l1_ssim_backward(int32_t C, int32_t H, int32_t W, const float *img, const float *target, const float *dmaps, float w_l1, float w_ssim,
	const float *grad_scale, float *dL_dimg, void *stream)
{
	if (C <= 0 || H <= 0 || W <= 0 || C > 65535 || !img || !target || !dmaps || !dL_dimg) { set_error("bad l1_ssim_backward arguments"); return FR_ERR_INVALID; }
	return launch_l1_ssim_backward(C, H, W, img, target, dmaps, w_l1, w_ssim, grad_scale, dL_dimg, (hipStream_t)stream);
}

int fr_backward(const fr_backward_args *a)
{
	if (!a) { set_error("null args"); return FR_ERR_INVALID; }
	if (!has_backward(a->variant))
	{ set_error("backward exists only for the original and pcheck_obb_sum/_max/_loss_weighted_max_count variants (the reference's inference variants have none)"); return FR_ERR_INVALID; }
	if (a->P == 0) return FR_OK;
	if (!a->geometry || !a->image || (a->R > 0 && !a->binning) || !a->dL_dpix || !a->radii) { set_error("missing workspace / gradient pointer"); return FR_ERR_INVALID; }
	if (!a->dL_dmean2D || !a->dL_dopacity || !a->dL_dmean3D || (!a->dL_dcov3D && a->cov3D_precomp) || (!a->dL_dcolor && a->colors_precomp) || !a->dL_dscale || !a->dL_drot)
	{ set_error("missing gradient output pointer"); return FR_ERR_INVALID; }
	if (a->shs && !a->dL_dsh) { set_error("dL_dsh is null"); return FR_ERR_INVALID; }
	if (a->shs_rest && (!a->shs || !a->dL_dsh_rest)) { set_error("shs_rest needs shs and dL_dsh_rest"); return FR_ERR_INVALID; }
	if (a->raw_activations && a->cov3D_precomp) { set_error("raw_activations needs scales + rotations"); return FR_ERR_INVALID; }
	// (row_sparse promises that every compact row is written: with cov3D_precomp nobody writes dL_dscale / dL_drot)
	if (a->row_sparse && a->cov3D_precomp) { set_error("row_sparse needs scales + rotations (with cov3D_precomp the dL_dscale / dL_drot rows would stay unwritten)"); return FR_ERR_INVALID; }
	return launch_backward(a);
}

int fr_backward_prefill(const fr_backward_args *a, void *fill_stream)
{
	if (!a) { set_error("null args"); return FR_ERR_INVALID; }
	if (a->P <= 0) return FR_OK;
	return launch_gradient_fill(a, (hipStream_t)fill_stream, false, true);
}

int fr_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix, uint8_t *present, void *stream)
{
	(void)projmatrix;
	if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) { set_error("bad mark_visible arguments"); return FR_ERR_INVALID; }
	if (P == 0) return FR_OK;
	return launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
}

} // extern "C"
