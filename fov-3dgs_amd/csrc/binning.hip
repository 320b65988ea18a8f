// Tile binning: per-tile offsets from the per-tile counters, and the per-tile depth sort.
//
// Replaces (reference, paths under fov3dgs/submodules/diff-gaussian-rasterization*/cuda_rasterizer/):
//   cub::DeviceScan::InclusiveSum over tiles_touched[P] + cudaMemcpy D2H   rasterizer_impl.cu:277-281
//   cub::DeviceRadixSort::SortPairs over (tile<<32 | depth) keys           rasterizer_impl.cu:300-308
//   cudaMemset(ranges) + identifyTileRanges                                rasterizer_impl.cu:116-138,310-317
//
// MI355X design: the reference sorts D 12-byte pairs by a 45-bit key with a 6-pass global LSD
// radix sort (~152 B of HBM traffic per instance). Here instances were already bucketed by tile
// at emission, so (a) tile ranges fall out of an exclusive scan over T (<= 8160) counters and
// (b) each bucket is sorted independently by one workgroup inside the CU's 160 KiB LDS:
// one 8-byte read and one 4-byte write per instance. The sort key is (depth bits << 32 | id),
// which reproduces the order a stable sort of the reference's keys emitted in index order gives.
#include "common.h"
#include "tile_scan.h"
#include "tile_sort.h"
#include <cstdlib>

namespace fr {

__global__ void __launch_bounds__(FR_TILE_SCAN_THREADS) k_tile_scan(const TileScanArgs ts)
{
	__shared__ uint32_t lds[tile_scan_lds_words<FR_TILE_SCAN_THREADS>()];
	if (ts.T <= FR_SCAN_MAX_TILES) tile_scan_body<FR_TILE_SCAN_THREADS>(ts, lds);
	else tile_scan_atomics<FR_TILE_SCAN_THREADS>(ts);
}

// The two classes of short lists, one workgroup per list: the lists of 512..2047 entries (the tile scan's bins 10 and 11:
// tile_order[h4 .. h4 + mid), h4 = totals[2], mid = totals[3]) when !SHORTEST, the lists of <= 511 entries (the rest of
// tile_order, empty tiles included) when SHORTEST.
template <int THREADS, int ITEMS, bool SHORTEST>
__global__ void __launch_bounds__(THREADS) k_tile_msort(const uint2 *ranges, const uint32_t *tile_order, uint64_t *entries,
	uint32_t *point_list, int T, const uint32_t *totals, SpecLimits lim)
{
	extern __shared__ __attribute__((aligned(16))) uint64_t sk[];
	if (!frame_fits(totals, lim)) return;
	const uint32_t lo = totals[2] + (SHORTEST ? totals[3] : 0u), hi = SHORTEST ? (uint32_t)T : totals[2] + totals[3];
	for (uint32_t b = lo + blockIdx.x; b < hi; b += gridDim.x)
	{
		const uint2 rg = ranges[tile_order[b]];
		if (rg.y != rg.x) msort_list<THREADS, ITEMS, false>(rg, entries, point_list, sk);
		__syncthreads(); // the next list reuses the LDS keys
	}
}

// The classes of longer lists, each sorted WHOLE in LDS by one workgroup: tile_order[totals[lo_word] .. totals[hi_word]) -- the
// tile scan lays the tiles out longest first and counts the lists with >= 2048 / 4096 / 8192 / 16384 entries (totals[2], [6],
// [8], [9]), so a class is a slice of tile_order. The grid is normally one workgroup per list (the hardware's placement
// of fresh workgroups is the load balancer); the loop covers a grid sized from a bound that came out too small.
template <int THREADS, int ITEMS>
__global__ void __launch_bounds__(THREADS, 4) k_tile_msort_direct(const uint2 *ranges, const uint32_t *tile_order, uint64_t *entries,
	uint32_t *point_list, const uint32_t *totals, SpecLimits lim, int lo_word, int hi_word)
{
	extern __shared__ __attribute__((aligned(16))) uint64_t sk[];
	if (!frame_fits(totals, lim)) return;
	const uint32_t lo = totals[lo_word], hi = totals[hi_word];
	for (uint32_t b = lo + blockIdx.x; b < hi; b += gridDim.x)
	{
		msort_list<THREADS, ITEMS, false>(ranges[tile_order[b]], entries, point_list, sk);
		__syncthreads(); // the next list reuses the LDS keys
	}
}

// The chunks of the split long lists (k_split_long): chunks[0 .. totals[4]), any length; one workgroup per chunk (the grid is
// the bound FR_SORT_MAX_CHUNKS of the workspace's capacity; the loop is a safety net).
template <int THREADS, int ITEMS>
__global__ void __launch_bounds__(THREADS) k_tile_msort_chunks(const uint2 *chunks, uint64_t *entries2, uint32_t *point_list,
	const uint32_t *totals, SpecLimits lim)
{
	extern __shared__ __attribute__((aligned(16))) uint64_t sk[];
	if (!frame_fits(totals, lim)) return;
	const uint32_t count = totals[4];
	for (uint32_t c = blockIdx.x; c < count; c += gridDim.x)
	{
		msort_list<THREADS, ITEMS, true>(chunks[c], entries2, point_list, sk);
		__syncthreads();
	}
}

#define FR_SPLIT_LDS_KEYS 8192 // keys staged in LDS per round of k_split_long's scatter (two workgroups per CU)
// The lists that do not fit the LDS of one workgroup (>= FR_SORT_SPLIT_MIN = 16384 entries: tile_order[0 .. totals[9]), none in
// a 1080p S-6M frame) regrouped by depth into chunks (split_list, tile_sort.h); k_tile_msort_chunks sorts the chunks --
// independent 1024-key sorts that spread over the whole chip.
__global__ void __launch_bounds__(FR_SPLIT_THREADS) k_split_long(const uint2 *ranges, const uint32_t *tile_order, const uint64_t *entries,
	uint64_t *entries2, uint2 *chunks, uint32_t *totals, SpecLimits lim)
{
	extern __shared__ __attribute__((aligned(16))) uint64_t s_keys[];
	FR_SPLIT_LDS_DECL(L, s_keys, FR_SPLIT_LDS_KEYS);
	if (!frame_fits(totals, lim)) return;
	const uint32_t nlong = totals[9];
	for (uint32_t blk = blockIdx.x; blk < nlong; blk += gridDim.x)
	{
		split_list(ranges[tile_order[blk]], entries, entries2, chunks, totals + 4, L);
		__syncthreads(); // the next list reuses the histogram
	}
}

TileScanArgs make_tile_scan_args(FwdCtx &c)
{
	TileScanArgs ts;
	ts.T = c.T; ts.tile_count = c.img.tile_count; ts.ranges = c.img.ranges; ts.totals = c.img.totals; ts.tile_order = c.img.tile_order;
	ts.totals_host = c.totals_host_dev; ts.seq = c.totals_seq;
	ts.tile_blend = c.fov_split ? c.img.tile_lv + 4 * (size_t)c.T : (const float *)nullptr;
	ts.render_items = c.img.render_items; ts.prefilter_flag = c.geom.slab_ctr;
	return ts;
}

int launch_tile_scan(FwdCtx &c)
{
	hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(FR_TILE_SCAN_THREADS), 0, c.stream, make_tile_scan_args(c));
	return check_launch("tile_scan", c.stream, c.a->debug);
}

// Helper streams of the calling host thread, one pair per (device, launch stream), created on first use: two frames in flight on
// two launch streams must not share them (frame n + 1's fills would queue behind frame n's colours). The size classes of the
// per-tile sort are independent kernels; the classes with long lists hold a handful of tiles that each keep one CU busy
// for 50-80 us, so the (many) short lists are sorted meanwhile on the helper stream `s` (event fork / join); `s2` carries a
// frame's fills and its colour kernel (fr_forward_begin / _finish), `s` also the backward pass's gradient fills.
AuxStream *aux_stream(hipStream_t main)
{
	static thread_local AuxStream cache[8];
	static thread_local int next_victim = 0;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return nullptr;
	for (AuxStream &a : cache) if (a.device == dev && a.main == main) return a.ok ? &a : nullptr;
	AuxStream *slot = nullptr;
	for (AuxStream &a : cache) if (a.device < 0) { slot = &a; break; }
	if (!slot)
	{
		// (more than eight launch streams in one thread: the oldest pair is recycled; its streams drain first)
		slot = &cache[next_victim]; next_victim = (next_victim + 1) & 7;
		if (slot->ok)
		{
			(void)hipStreamSynchronize(slot->s); (void)hipStreamSynchronize(slot->s2);
			(void)hipStreamDestroy(slot->s); (void)hipStreamDestroy(slot->s2);
			(void)hipEventDestroy(slot->fork); (void)hipEventDestroy(slot->fork2); (void)hipEventDestroy(slot->join); (void)hipEventDestroy(slot->join2);
		}
	}
	AuxStream &a = *slot;
	a.device = dev; a.main = main;
	a.ok = hipStreamCreateWithFlags(&a.s, hipStreamNonBlocking) == hipSuccess &&
		hipStreamCreateWithFlags(&a.s2, hipStreamNonBlocking) == hipSuccess &&
		hipEventCreateWithFlags(&a.fork, hipEventDisableTiming) == hipSuccess &&
		hipEventCreateWithFlags(&a.fork2, hipEventDisableTiming) == hipSuccess &&
		hipEventCreateWithFlags(&a.join, hipEventDisableTiming) == hipSuccess &&
		hipEventCreateWithFlags(&a.join2, hipEventDisableTiming) == hipSuccess;
	(void)hipGetLastError();
	return a.ok ? &a : nullptr;
}

// Stage "tile_sort": one kernel per size class, every list sorted whole in LDS by one workgroup (sort_keys_lds, tile_sort.h):
//   <= 512 entries: one wave;  513..2047: 256 threads;  2048..4095: 512 threads;  4096..8191: 512 threads x 16 keys;
//   8192..16383: 1024 threads x 16 keys (139 KiB of LDS);  longer: regrouped by depth into chunks first (k_split_long).
// The classes are independent: the three of the long lists run on the launch stream, the two of the short lists meanwhile on
// the helper stream (event fork / join). Every kernel finds its lists from the tile scan's class counts in device memory (the
// grids come from the host's copy of them) and loops if its grid came out too small.
int launch_tile_sort(FwdCtx &c)
{
	const uint2 *rg = c.img.ranges;
	const uint32_t *ord = c.img.tile_order;
	uint32_t *totals = c.img.totals;
	const SpecLimits lim = { (uint32_t)c.capacity, (uint32_t)c.items_cap };
	static const bool serial = getenv("FR_SERIAL_SORT") != nullptr;
	const int h4 = c.heavy4, h8 = c.heavy8;
	const int longest = c.a->max_tile_instances;
	// long lists exist: the short ones are sorted meanwhile on the helper stream
	AuxStream *ax = (h4 > 0 && !serial && !c.a->debug && !c.a->no_helper_streams) ? aux_stream(c.stream) : nullptr;
	// (measured on the S-6M frames, stage time: this split 88 us; the 2048..4095 class on the helper stream too 94; on a third
	// stream 96; the 8192..16383 class -- a handful of workgroups that need a whole CU's LDS each -- on a third stream 95: the
	// stage is bound by the sum of the work, not by a chain)
	hipStream_t small = c.stream;
	if (ax)
	{
		(void)hipEventRecord(ax->fork, c.stream);
		(void)hipStreamWaitEvent(ax->s, ax->fork, 0);
		small = ax->s;
	}
	if (longest >= FR_SORT_SPLIT_MIN)
	{
		static const hipError_t lds_ok = hipFuncSetAttribute((const void *)k_split_long, hipFuncAttributeMaxDynamicSharedMemorySize, FR_SPLIT_LDS_KEYS * (int)sizeof(uint64_t));
		if (lds_ok != hipSuccess) { set_error("hipFuncSetAttribute(k_split_long): %s", hipGetErrorString(lds_ok)); return FR_ERR_HIP; }
		const int nsplit = h8 < 256 ? (h8 > 0 ? h8 : 1) : 256; // (how many lists are that long is only known on the device)
		hipLaunchKernelGGL(k_split_long, dim3(nsplit), dim3(FR_SPLIT_THREADS), FR_SPLIT_LDS_KEYS * sizeof(uint64_t), c.stream, rg, ord, c.bin.entries,
			c.bin.entries2, c.bin.chunks, totals, lim);
		const size_t max_chunks = FR_SORT_MAX_CHUNKS(c.capacity);
		hipLaunchKernelGGL((k_tile_msort_chunks<256, 8>), dim3((unsigned)max_chunks), dim3(256), 2304 * sizeof(uint64_t), c.stream,
			c.bin.chunks, c.bin.entries2, c.bin.point_list, totals, lim);
	}
	if (longest >= 8192 && h8 > 0)
	{
		constexpr int lds = (16384 + 1024) * (int)sizeof(uint64_t);
		static const hipError_t lds_ok = hipFuncSetAttribute((const void *)k_tile_msort_direct<1024, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		if (lds_ok != hipSuccess) { set_error("hipFuncSetAttribute(k_tile_msort_direct<1024, 16>): %s", hipGetErrorString(lds_ok)); return FR_ERR_HIP; }
		hipLaunchKernelGGL((k_tile_msort_direct<1024, 16>), dim3(h8 < 256 ? h8 : 256), dim3(1024), lds, c.stream, rg, ord, c.bin.entries, c.bin.point_list,
			totals, lim, 9, 8);
	}
	if (longest >= 4096 && h8 > 0)
	{
		constexpr int lds = (8192 + 512) * (int)sizeof(uint64_t);
		static const hipError_t lds_ok = hipFuncSetAttribute((const void *)k_tile_msort_direct<512, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		if (lds_ok != hipSuccess) { set_error("hipFuncSetAttribute(k_tile_msort_direct<512, 16>): %s", hipGetErrorString(lds_ok)); return FR_ERR_HIP; }
		hipLaunchKernelGGL((k_tile_msort_direct<512, 16>), dim3(h8), dim3(512), lds, c.stream, rg, ord, c.bin.entries, c.bin.point_list, totals, lim, 8, 6);
	}
	if (h4 - h8 > 0)
		hipLaunchKernelGGL((k_tile_msort_direct<512, 8>), dim3(h4 - h8), dim3(512), 4608 * sizeof(uint64_t), c.stream, rg, ord, c.bin.entries,
			c.bin.point_list, totals, lim, 6, 2);
	// (one workgroup per list of the class: a grid over all T tiles started 16 000 workgroups per frame only to find out that
	// the list belongs to another kernel)
	const int nmid = c.heavy2, nshort = c.T - c.heavy4 - c.heavy2;
	if (nmid > 0)
		hipLaunchKernelGGL((k_tile_msort<256, 8, false>), dim3(nmid), dim3(256), 2304 * sizeof(uint64_t), small, rg, ord, c.bin.entries, c.bin.point_list,
			c.T, totals, lim);
	if (nshort > 0)
		hipLaunchKernelGGL((k_tile_msort<64, 8, true>), dim3(nshort), dim3(64), 576 * sizeof(uint64_t), small, rg, ord, c.bin.entries, c.bin.point_list,
			c.T, totals, lim);
	if (ax)
	{
		(void)hipEventRecord(ax->join, ax->s);
		(void)hipStreamWaitEvent(c.stream, ax->join, 0);
	}
	return check_launch("tile_sort", c.stream, c.a->debug);
}

} // namespace fr
