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
#include <cstdlib>

namespace fr {

constexpr int FR_TILE_SCAN_THREADS = 512;
__global__ void __launch_bounds__(FR_TILE_SCAN_THREADS) k_tile_scan(const TileScanArgs ts)
{
	if (ts.T <= FR_SCAN_MAX_TILES) tile_scan_body<FR_TILE_SCAN_THREADS>(ts);
	else tile_scan_atomics<FR_TILE_SCAN_THREADS>(ts);
}

// All-ascending bitonic network (first step of each merge mirrors the partner index), so that
// virtual +inf padding above n never moves: comparators whose upper index is >= n are no-ops.
// GLOBAL: keys live in global memory and are exchanged between waves of this workgroup, so loads
// and stores go around the per-CU L1 (agent-scope relaxed atomics = sc1 accesses).
template <bool GLOBAL>
__device__ __forceinline__ uint64_t key_ld(const uint64_t *p)
{
	if (GLOBAL) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return *p;
}
template <bool GLOBAL>
__device__ __forceinline__ void key_st(uint64_t *p, uint64_t v)
{
	if (GLOBAL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else *p = v;
}
template <bool GLOBAL>
__device__ __forceinline__ void bitonic_sort(uint64_t *keys, int n, int npow2, int tid, int nthreads)
{
	for (int k = 2; k <= npow2; k <<= 1)
	{
		// flip step: i <-> i ^ (k-1)
		{
			const int half = k >> 1;
			for (int p = tid; p < (npow2 >> 1); p += nthreads)
			{
				const int blk = p / half, off = p - blk * half;
				const int i = blk * k + off, l = blk * k + (k - 1 - off);
				if (l < n)
				{
					const uint64_t a = key_ld<GLOBAL>(keys + i), b = key_ld<GLOBAL>(keys + l);
					if (a > b) { key_st<GLOBAL>(keys + i, b); key_st<GLOBAL>(keys + l, a); }
				}
			}
			__syncthreads();
		}
		for (int j = k >> 2; j > 0; j >>= 1)
		{
			for (int p = tid; p < (npow2 >> 1); p += nthreads)
			{
				const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
				const int l = i | j;
				if (l < n)
				{
					const uint64_t a = key_ld<GLOBAL>(keys + i), b = key_ld<GLOBAL>(keys + l);
					if (a > b) { key_st<GLOBAL>(keys + i, b); key_st<GLOBAL>(keys + l, a); }
				}
			}
			__syncthreads();
		}
	}
}

// ---- per-tile merge sort in LDS -------------------------------------------------------------
// One workgroup per tile, THREADS x ITEMS keys of capacity. Every thread sorts ITEMS consecutive keys
// in registers (odd-even transposition network), then log2(n / ITEMS) merge passes follow: a thread
// finds its ITEMS-long slice of the merged output by a merge-path binary search and merges it
// sequentially out of LDS into registers; results are written back in place after a barrier.
// LDS traffic is O(n log n) (vs O(n log^2 n) for the bitonic network), which is what bounds a CU that
// hosts several tiles at once. Keys are unique (the id is part of the key), so no stability issue.
template <int ITEMS>
__device__ __forceinline__ void reg_sort(uint64_t (&k)[ITEMS])
{
#pragma unroll
	for (int r = 0; r < ITEMS; r++)
#pragma unroll
		for (int i = (r & 1); i + 1 < ITEMS; i += 2)
		{
			const uint64_t lo = k[i] < k[i + 1] ? k[i] : k[i + 1];
			const uint64_t hi = k[i] < k[i + 1] ? k[i + 1] : k[i];
			k[i] = lo; k[i + 1] = hi;
		}
}

// Which lists are regrouped by depth before sorting (k_split_long): lists with >= 2048 entries, or -- `direct` -- only those
// with >= 4096 while the 2048..4095 class is sorted directly, one 512-thread workgroup per list. Either plan sorts every
// list; which one is faster depends on the frame: a few hundred long lists are latency-bound (split them all), THOUSANDS
// of lists of 2048..4095 entries (non-foveated / training frames) are throughput-bound (sort that class directly). The host
// chooses -- from the frame's class counts when it has them, from the previous frame of the kind when the stage is launched
// ahead of them -- and the kernels find their lists from the counts in device memory (totals[2] = lists with >= 2048
// entries, totals[6] = with >= 4096; they are the first entries of tile_order, longest first).
struct SortPlan { bool direct; uint32_t nlong, split_min, h4, h8; };
__device__ __forceinline__ SortPlan sort_plan(const uint32_t *totals, bool direct)
{
	SortPlan p;
	p.h4 = totals[2]; p.h8 = totals[6];
	p.direct = direct;
	p.nlong = p.direct ? p.h8 : p.h4;
	p.split_min = p.direct ? 2u * FR_SORT_SPLIT_MIN : (uint32_t)FR_SORT_SPLIT_MIN;
	return p;
}
// Kernels launched ahead of the frame's counts (fr_forward) leave without touching anything when the frame does not fit
// what they were sized for -- more instances than the binning workspace holds, or more blend work items than the blend
// grid has workgroups; the host then replays the whole stage.
struct SpecLimits { uint32_t capacity, items_cap; };
__device__ __forceinline__ bool frame_fits(const uint32_t *totals, const SpecLimits lim) { return totals[0] <= lim.capacity && totals[5] <= lim.items_cap; }

// One list of n keys (entries + rg.x ..) sorted into point_list by the whole workgroup; LDS holds THREADS x ITEMS keys.
// fallback: a list that does not fit is sorted in place in global memory by the bitonic network (chunks of a split list
// with thousands of equal depths); otherwise such a list is left to another kernel.
template <int THREADS, int ITEMS, bool FALLBACK>
__device__ __forceinline__ void msort_list(const uint2 rg, uint64_t *entries, uint32_t *point_list, uint64_t *sk)
{
	const int n = (int)(rg.y - rg.x);
	const int tid = threadIdx.x;
	uint64_t *src = entries + rg.x;
	uint32_t *dst = point_list + rg.x;
	if (n > THREADS * ITEMS)
	{
		if (!FALLBACK) return;
		int npow2 = 1;
		while (npow2 < n) npow2 <<= 1;
		bitonic_sort<true>(src, n, npow2, tid, THREADS);
		for (int i = tid; i < n; i += THREADS) dst[i] = (uint32_t)key_ld<true>(src + i);
		return;
	}
	// active capacity: ITEMS * 2^k >= n
	int runs = 1;
	while (runs * ITEMS < n) runs <<= 1;
	const int nact = runs * ITEMS;
	// LDS layout: one spare slot after every 8 keys (SK). A thread owns ITEMS = 8 consecutive keys, i.e. lanes are 64
	// bytes apart: unpadded, the 64 lanes of an access fall on 4 of the 32 eight-byte bank pairs (16-way conflict);
	// with the spare slot the lane stride is 72 bytes and all bank pairs are used.
#define SK(i) sk[(i) + ((i) >> 3)]
	for (int i = tid; i < nact; i += THREADS) SK(i) = i < n ? src[i] : ~0ull;
	__syncthreads();
	const bool act = tid < runs;
	uint64_t k[ITEMS];
	const int o = tid * ITEMS;
	if (act)
	{
#pragma unroll
		for (int i = 0; i < ITEMS; i++) k[i] = SK(o + i);
		reg_sort<ITEMS>(k);
#pragma unroll
		for (int i = 0; i < ITEMS; i++) SK(o + i) = k[i];
	}
	__syncthreads();
	for (int L = ITEMS; L < nact; L <<= 1)
	{
		if (act)
		{
			const int base = o & ~(2 * L - 1);
			const int d = o - base;                      // outputs before mine inside this pair of runs
			const int a0 = base, b0 = base + L;
#define A(x) SK(a0 + (x))
#define B(x) SK(b0 + (x))
			int lo = max(0, d - L), hi = min(d, L);
			while (lo < hi)
			{
				const int mid = (lo + hi) >> 1;
				if (A(mid) < B(d - 1 - mid)) lo = mid + 1; else hi = mid;
			}
			int i = lo, j = d - lo;
			uint64_t av = i < L ? A(i) : ~0ull, bv = j < L ? B(j) : ~0ull;
#pragma unroll
			for (int t = 0; t < ITEMS; t++)
			{
				const bool ta = av <= bv;
				k[t] = ta ? av : bv;
				if (ta) { i++; av = i < L ? A(i) : ~0ull; }
				else { j++; bv = j < L ? B(j) : ~0ull; }
			}
		}
		__syncthreads();
		if (act)
		{
#pragma unroll
			for (int t = 0; t < ITEMS; t++) SK(o + t) = k[t];
		}
		__syncthreads();
	}
	for (int i = tid; i < n; i += THREADS) dst[i] = (uint32_t)SK(i);
#undef A
#undef B
#undef SK
}

// One tile list per workgroup (ranges[tile_order[block]]): the lists with n_lo < n < n_hi that fit the kernel's LDS.
template <int THREADS, int ITEMS>
__global__ void __launch_bounds__(THREADS) k_tile_msort(const uint2 *ranges, const uint32_t *tile_order, uint64_t *entries,
	uint32_t *point_list, int n_lo, int n_hi, const uint32_t *totals, SpecLimits lim)
{
	extern __shared__ __attribute__((aligned(16))) uint64_t sk[];
	if (!frame_fits(totals, lim)) return;
	const uint2 rg = ranges[tile_order[blockIdx.x]];
	const int n = (int)(rg.y - rg.x);
	if (n <= n_lo || n >= n_hi) return; // another kernel sorts this list
	msort_list<THREADS, ITEMS, false>(rg, entries, point_list, sk);
}

// The lists of 2048..4095 entries when the frame sorts that class directly (sort_plan): tile_order[h8 .. h4). The grid is
// normally one workgroup per list (the hardware's placement of fresh workgroups is the load balancer); the loop only covers
// a grid that was sized ahead of the counts and came out too small.
template <int THREADS, int ITEMS>
__global__ void __launch_bounds__(THREADS) k_tile_msort_direct(const uint2 *ranges, const uint32_t *tile_order, uint64_t *entries,
	uint32_t *point_list, const uint32_t *totals, SpecLimits lim)
{
	extern __shared__ __attribute__((aligned(16))) uint64_t sk[];
	if (!frame_fits(totals, lim)) return;
	const SortPlan pl = sort_plan(totals, true);
	for (uint32_t b = pl.h8 + blockIdx.x; b < pl.h4; b += gridDim.x)
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

// Long tile lists (>= FR_SORT_SPLIT_MIN entries) are not sorted as one piece: a handful of them used to occupy one
// CU each for 50-80 us with sixteen-way merge passes while the rest of the chip had nothing left to do. A counting
// pass on the depth bits (a fixed monotone quantisation into FR_SORT_FINE_BUCKETS buckets) regroups the list into chunks of ~FR_SORT_CHUNK_TARGET entries with disjoint, increasing depth ranges;
// equal depths share a bucket, so sorting every chunk by (depth, id) sorts the list. The chunks are independent
// 1024-key sorts that spread over the whole chip. Keys are streamed from global memory twice (histogram,
// scatter); LDS holds only the histogram.
#define FR_SPLIT_REGS 16 // keys per thread held in registers by k_split_long (lists up to 16384 entries)
#define FR_SPLIT_LDS_KEYS 8192 // keys staged in LDS per round of its scatter
// The regrouped keys of a list of up to FR_SPLIT_REGS x 1024 entries go through LDS (s_keys: dynamic shared memory,
// FR_SPLIT_LDS_KEYS slots): the scatter by depth bucket happens there and the list leaves the workgroup as one coalesced copy. Scattered
// straight to global memory, every 8-byte store was a partial-sector write of its own -- ~11 cycles of the CU's memory
// pipeline each (tools/scratch/gather_rate.hip), 50 us for the longest list of a 1080p frame, which one workgroup = one CU
// handles alone: the kernel's whole duration.
__global__ void __launch_bounds__(1024) k_split_long(const uint2 *ranges, const uint32_t *tile_order, const uint64_t *entries,
	uint64_t *entries2, uint2 *chunks, uint32_t *totals, SpecLimits lim, int direct)
{
	extern __shared__ __attribute__((aligned(16))) uint64_t s_keys[];
	__shared__ uint32_t s_hist[FR_SORT_FINE_BUCKETS];      // counts -> exclusive offsets -> scatter cursors
	__shared__ uint32_t s_start[FR_SORT_FINE_BUCKETS + 1]; // compacted chunk starts
	__shared__ uint32_t s_wave[16], s_wave2[16];
	__shared__ uint32_t s_slot;
	if (!frame_fits(totals, lim)) return;
	const SortPlan pl = sort_plan(totals, direct != 0);
	uint32_t *chunk_ctr = totals + 4;
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	// the long lists are the first pl.nlong entries of tile_order (longest first); normally one workgroup per list, the loop
	// covers a grid sized ahead of the counts that came out too small
	for (uint32_t blk = blockIdx.x; blk < pl.nlong; blk += gridDim.x)
	{
#ifdef FR_SPLIT_TIMERS
	const uint64_t tm0 = wall_clock64(); uint64_t tm1 = 0, tm2 = 0, tm3 = 0, tm4 = 0, tm5 = 0;
#define TMS(x) x = wall_clock64()
#else
#define TMS(x)
#endif
	const uint2 rg = ranges[tile_order[blk]];
	const uint32_t n = rg.y - rg.x;
	if (n >= pl.split_min)
	{
	const uint64_t *src = entries + rg.x;
	uint64_t *dst = entries2 + rg.x;
	// 1. depth bucket: the bit pattern of a positive float orders like its value, so a fixed monotone map of the
	// bits needs no pass over the list: 128 buckets per octave from the near plane (0.2) up, 16 octaves, the rest
	// clamped into the last bucket (uneven buckets are fine, chunks are cut by count)
	for (int b = tid; b < FR_SORT_FINE_BUCKETS; b += 1024) s_hist[b] = 0;
	__syncthreads();
	constexpr uint32_t dmin = 0x3E4CCCCDu; // 0.2f
	constexpr int shift = 16;
#define FR_DEPTH_BUCKET(d) min((uint32_t)(FR_SORT_FINE_BUCKETS - 1), ((d) > dmin ? (d) - dmin : 0u) >> shift)
	// 2. histogram. Lists of up to FR_SPLIT_REGS x 1024 keys are read ONCE, all loads in flight together, and kept
	// in registers for the scatter below (a loop of dependent load -> LDS atomic iterations costs a memory round
	// trip per 1024 keys, and the longest list is this kernel's critical path)
	const bool in_regs = n <= FR_SPLIT_REGS * 1024u;
	uint64_t kreg[FR_SPLIT_REGS];
	if (in_regs)
	{
#pragma unroll
		for (int k = 0; k < FR_SPLIT_REGS; k++) { const uint32_t i = tid + 1024u * k; kreg[k] = i < n ? src[i] : 0ull; }
#pragma unroll
		for (int k = 0; k < FR_SPLIT_REGS; k++)
			if (tid + 1024u * k < n) atomicAdd(&s_hist[FR_DEPTH_BUCKET((uint32_t)(kreg[k] >> 32))], 1u);
	}
	else
		for (uint32_t i = tid; i < n; i += 1024) atomicAdd(&s_hist[FR_DEPTH_BUCKET((uint32_t)(src[i] >> 32))], 1u);
	TMS(tm1);
	__syncthreads();
	TMS(tm2);
	// 3. exclusive scan of the 2048 counts (two consecutive buckets per thread)
	const uint32_t c0 = s_hist[2 * tid], c1 = s_hist[2 * tid + 1];
	uint32_t sc = c0 + c1;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)sc, off); if (lane >= off) sc += v; }
	if (lane == 63) s_wave[wid] = sc;
	__syncthreads();
	uint32_t wave_off = 0;
#pragma unroll
	for (int w = 0; w < 16; w++) if (w < wid) wave_off += s_wave[w];
	const uint32_t e0 = wave_off + sc - (c0 + c1), e1 = e0 + c0; // exclusive offsets of my two buckets
	__syncthreads();
	s_hist[2 * tid] = e0; s_hist[2 * tid + 1] = e1;
	__syncthreads();
	// 4. a chunk starts where the running count crosses a multiple of the target (monotone in the bucket index)
	const uint32_t prev = tid == 0 ? 0u : s_hist[2 * tid - 1];
	const bool f0 = tid == 0 || (e0 / FR_SORT_CHUNK_TARGET) != (prev / FR_SORT_CHUNK_TARGET);
	const bool f1 = (e1 / FR_SORT_CHUNK_TARGET) != (e0 / FR_SORT_CHUNK_TARGET);
	uint32_t fs = (f0 ? 1u : 0u) + (f1 ? 1u : 0u);
	const uint32_t mine = fs;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)fs, off); if (lane >= off) fs += v; }
	if (lane == 63) s_wave2[wid] = fs;
	__syncthreads();
	uint32_t foff = 0, nchunks = 0;
#pragma unroll
	for (int w = 0; w < 16; w++) { if (w < wid) foff += s_wave2[w]; nchunks += s_wave2[w]; }
	uint32_t pos = foff + fs - mine;
	if (f0) s_start[pos++] = e0;
	if (f1) s_start[pos] = e1;
	TMS(tm3);
	if (tid == 0) { s_start[nchunks] = n; s_slot = atomicAdd(chunk_ctr, nchunks); }
	__syncthreads();
	TMS(tm4);
	for (uint32_t k = tid; k < nchunks; k += 1024) chunks[s_slot + k] = make_uint2(rg.x + s_start[k], rg.x + s_start[k + 1]);
	// 5. scatter (the offsets become cursors)
	if (in_regs)
	{
		// (FR_SPLIT_LDS_KEYS slots: two workgroups per CU; a longer list goes through them in rounds)
		uint32_t pos[FR_SPLIT_REGS];
#pragma unroll
		for (int k = 0; k < FR_SPLIT_REGS; k++)
			pos[k] = tid + 1024u * k < n ? atomicAdd(&s_hist[FR_DEPTH_BUCKET((uint32_t)(kreg[k] >> 32))], 1u) : 0xffffffffu;
		for (uint32_t base = 0; base < n; base += FR_SPLIT_LDS_KEYS)
		{
#pragma unroll
			for (int k = 0; k < FR_SPLIT_REGS; k++)
				if (pos[k] - base < (uint32_t)FR_SPLIT_LDS_KEYS) s_keys[pos[k] - base] = kreg[k];
			__syncthreads();
			const uint32_t m = min((uint32_t)FR_SPLIT_LDS_KEYS, n - base);
			for (uint32_t i = tid; i < m; i += 1024) dst[base + i] = s_keys[i];
			__syncthreads();
		}
		TMS(tm5);
	}
	else
		for (uint32_t i = tid; i < n; i += 1024)
		{
			const uint64_t key = src[i];
			dst[atomicAdd(&s_hist[FR_DEPTH_BUCKET((uint32_t)(key >> 32))], 1u)] = key;
		}
#undef FR_DEPTH_BUCKET
#ifdef FR_SPLIT_TIMERS
	if (tid == 0)
	{
		// developer build (tools/split_stats.py): per-list phase times in the tail of the chunk table
		uint32_t *d = (uint32_t *)(chunks + FR_SORT_MAX_CHUNKS(lim.capacity)) - 8 * (blk + 1);
		d[0] = n; d[1] = (uint32_t)(tm1 - tm0); d[2] = (uint32_t)(tm2 - tm0); d[3] = (uint32_t)(tm3 - tm0); d[4] = (uint32_t)(tm4 - tm0); d[5] = (uint32_t)(tm5 - tm0);
		d[6] = (uint32_t)(wall_clock64() - tm0); d[7] = (uint32_t)(tm0 & 0xffffff);
	}
#endif
	}
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

// Helper stream of the calling host thread (per device), created on first use. The size classes of the per-tile
// sort are independent kernels; the two classes with long lists hold a handful of tiles that each keep one CU busy
// for 50-80 us, so the (many) short lists are sorted meanwhile on the helper stream (event fork / join).
AuxStream *aux_stream()
{
	static thread_local AuxStream cache[8];
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return nullptr;
	AuxStream &a = cache[dev & 7];
	if (a.device != dev)
	{
		a.device = dev;
		a.ok = hipStreamCreateWithFlags(&a.s, hipStreamNonBlocking) == hipSuccess &&
			hipStreamCreateWithFlags(&a.s2, hipStreamNonBlocking) == hipSuccess &&
			hipEventCreateWithFlags(&a.fork, hipEventDisableTiming) == hipSuccess &&
			hipEventCreateWithFlags(&a.join, hipEventDisableTiming) == hipSuccess &&
			hipEventCreateWithFlags(&a.join2, hipEventDisableTiming) == hipSuccess;
		(void)hipGetLastError();
	}
	return a.ok ? &a : nullptr;
}

// Stage "tile_sort". counts_known: the host has the tile scan's class counts; otherwise the stage is launched ahead of
// them: the plan and the grids come from the previous frame of the kind (FwdCtx::hint_*), every kernel finds its lists from
// the counts in device memory and loops if its grid came out too small.
int launch_tile_sort(FwdCtx &c)
{
	const dim3 grid(c.T);
	const uint2 *rg = c.img.ranges;
	const uint32_t *ord = c.img.tile_order;
	uint32_t *totals = c.img.totals;
	const SpecLimits lim = { (uint32_t)c.capacity, (uint32_t)c.items_cap };
	static const bool serial = getenv("FR_SERIAL_SORT") != nullptr;
	const bool known = c.counts_known != 0;
	const int h4 = known ? c.heavy4 : c.hint_heavy4, h8 = known ? c.heavy8 : c.hint_heavy8;
	const bool direct = h4 - h8 >= FR_SORT_DIRECT_TILES;
	// grids ahead of the counts: a quarter more than last time (at least 64 workgroups)
	auto ahead = [&](int n) { const int g = n + n / 4 + 64; return g < c.T ? g : c.T; };
	const int nlong = known ? (direct ? h8 : h4) : ahead(direct ? h8 : h4);
	// long lists exist: the short ones are sorted meanwhile on the helper stream
	AuxStream *ax = (nlong > 0 && !serial && !c.a->debug) ? aux_stream() : nullptr;
	hipStream_t small = c.stream;
	if (ax)
	{
		(void)hipEventRecord(ax->fork, c.stream);
		(void)hipStreamWaitEvent(ax->s, ax->fork, 0);
		small = ax->s;
	}
	if (nlong > 0)
	{
		static const hipError_t lds_ok = hipFuncSetAttribute((const void *)k_split_long, hipFuncAttributeMaxDynamicSharedMemorySize, FR_SPLIT_LDS_KEYS * (int)sizeof(uint64_t));
		if (lds_ok != hipSuccess) { set_error("hipFuncSetAttribute(k_split_long): %s", hipGetErrorString(lds_ok)); return FR_ERR_HIP; }
		hipLaunchKernelGGL(k_split_long, dim3(nlong), dim3(1024), FR_SPLIT_LDS_KEYS * sizeof(uint64_t), c.stream, rg, ord, c.bin.entries, c.bin.entries2, c.bin.chunks,
			totals, lim, direct ? 1 : 0);
		const size_t max_chunks = FR_SORT_MAX_CHUNKS(c.capacity);
		hipLaunchKernelGGL((k_tile_msort_chunks<256, 8>), dim3((unsigned)max_chunks), dim3(256), 2304 * sizeof(uint64_t), c.stream,
			c.bin.chunks, c.bin.entries2, c.bin.point_list, totals, lim);
	}
	if (direct)
	{
		const int ndirect = known ? h4 - h8 : ahead(h4 - h8);
		hipLaunchKernelGGL((k_tile_msort_direct<512, 8>), dim3(ndirect), dim3(512), 4608 * sizeof(uint64_t), small, rg, ord, c.bin.entries,
			c.bin.point_list, totals, lim);
	}
	// (class boundaries: the 513..2047 class reads n_lo = 512, the 2048..4095 one belongs to the kernels above in either plan)
	if (!known || c.a->max_tile_instances > 512)
		hipLaunchKernelGGL((k_tile_msort<256, 8>), grid, dim3(256), 2304 * sizeof(uint64_t), small, rg, ord, c.bin.entries, c.bin.point_list,
			512, FR_SORT_SPLIT_MIN, totals, lim);
	hipLaunchKernelGGL((k_tile_msort<64, 8>), grid, dim3(64), 576 * sizeof(uint64_t), small, rg, ord, c.bin.entries, c.bin.point_list,
		0, 513, totals, lim);
	if (ax)
	{
		(void)hipEventRecord(ax->join, ax->s);
		(void)hipStreamWaitEvent(c.stream, ax->join, 0);
	}
	return check_launch("tile_sort", c.stream, c.a->debug);
}

} // namespace fr
