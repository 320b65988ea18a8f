// The tile scan: per-tile instance counts -> per-tile ranges, the frame's totals (published to the host), the tiles in
// longest-list-first order and the blend kernel's work items. One workgroup does it (k_tile_scan, binning.hip).
#pragma once
#include "common.h"

namespace fr {

struct TileScanArgs {
	int T;
	uint32_t *tile_count; uint2 *ranges; uint32_t *totals; uint32_t *tile_order;
	uint32_t *totals_host; uint32_t seq;
	const float *tile_blend; uint32_t *render_items; const uint32_t *prefilter_flag;
};

struct FwdCtx;
TileScanArgs make_tile_scan_args(FwdCtx &c); // binning.hip

// Single workgroup: exclusive scan of tile_count[T] -> ranges, reset the counters to serve as
// emission cursors, publish {total, max}.
// Also lays out the blend kernel's work items (render_items): one per wave that has something to do -- two bands per
// tile, and for the RF two-level tiles (blend flag in tile_blend, null otherwise) one such pair per level state --
// in the same longest-list-first order, so that the persistent blend waves pull the costliest items first and no
// workgroup is launched just to find out that its tile has a single level.
//
// Longest-list-first is a counting sort on bin(n) = bit length of the list (2^(b-1) <= n < 2^b; 65536 and up share bin 17).
// Round 2 counted and placed with LDS atomics on the 34 bin counters: 8160 tiles x 2 counters x 2 passes, 64 lanes of a
// wave on the same handful of addresses -- the LDS serialises those, and that, not the load chains, was the 28 us of
// k_tile_scan. Here every thread keeps a histogram of ITS run of tiles in LDS (hist[row][thread], row = 17 - bin, no
// two threads share a word, no atomics), one linear scan over the rows (longest first) turns the counts into each
// (bin, thread)'s first position, and the placement pass walks the same words. The order is now a function of the
// counts alone (stable: tile index ascending inside a bin), not of the order atomics retire in.
// A word packs {tiles: low 16 bits, two-level tiles: high 16 bits}: good for T <= 65535 (the 8K case, 129 600 tiles,
// takes tile_scan_atomics below).
constexpr int FR_TILE_SCAN_THREADS = 512; // threads of the scan (k_tile_scan's workgroup; the first eight waves of k_bin's last workgroup)
constexpr int FR_SCAN_BINS = 18;
constexpr int FR_SCAN_ROWS = 19; // one empty row: an odd stride for the linear scan (no LDS bank conflicts)
constexpr int FR_SCAN_MAX_TILES = 65535;

__device__ __forceinline__ int scan_bin(const uint32_t v) { return v ? min(32 - __clz((int)v), FR_SCAN_BINS - 1) : 0; }

// What tid 0 publishes once the counts are in: device totals + the host's pinned copy + the sequence number.
__device__ __forceinline__ void publish_totals(const TileScanArgs &ts, const uint32_t total, const uint32_t longest, const uint32_t h4,
                                               const uint32_t h8, const uint32_t mid, const uint32_t nitems, const uint32_t h16, const uint32_t h32)
{
	uint32_t *const totals = ts.totals; uint32_t *const totals_host = ts.totals_host;
	totals[0] = total; totals[1] = longest; totals[2] = h4; totals[3] = mid; totals[6] = h8;
	totals[4] = 0; // chunk counter of k_split_long
	totals[8] = h16; totals[9] = h32; // lists with >= 8192 / >= 16384 entries (the sort kernels find their classes from these)
	const uint32_t pf = *ts.prefilter_flag; // k_project: a Gaussian behind the near plane although `prefiltered` was set
	totals[7] = pf;
	totals[5] = nitems;
	// the host sizes the binning buffer from these: written straight into its pinned memory (no copy command)
	// and followed by this frame's sequence number, which the host polls for (it then prepares the next launches
	// while this kernel finishes)
	if (totals_host)
	{
		totals_host[0] = total; totals_host[1] = longest; totals_host[2] = h4; totals_host[3] = mid;
		totals_host[5] = nitems; totals_host[6] = h8; totals_host[7] = pf;
		totals_host[8] = ts.prefilter_flag[1]; // slab_ctr[1]: the number of cull-pass survivors (fr_forward_args.num_candidates)
		// slab_ctr[5]: a binning workgroup's region list did not fit its segment (raised with device-scope atomics by workgroups that may
		// sit on another XCD: read from the memory side)
		totals_host[9] = __hip_atomic_load(ts.prefilter_flag + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__threadfence_system();
		__hip_atomic_store(&totals_host[4], ts.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// Every thread owns runs of 16 consecutive tiles, one run per chunk of 16 x THREADS tiles (1080p: 8160 tiles, one chunk).
// One CU has ONE texture-address unit for all its waves, and a load or store whose lanes lie 64 bytes apart costs it
// ~3 cycles a lane (the lane-op price list): read and written run by run, the scan's 50 000 such lane
// operations were 24 of its 28 us. So global memory is touched with CONSECUTIVE lanes on consecutive words only, and
// the runs are cut out of / put together in LDS (run r at word 17 r: conflict-free both ways):
//   counts + blend flags -> LDS -> registers;  starts -> LDS -> ranges;  (tile, item offset) at its place -> LDS -> lists.
// Images above one chunk keep the coalesced loads and ranges and scatter the placement from the threads.
// LDS words the scan of THREADS threads needs (the caller provides them: k_tile_scan a static array, k_bin's last workgroup the
// dynamic LDS its histogram and staging rows lived in).
template <int THREADS>
constexpr int tile_scan_lds_words() { return FR_SCAN_ROWS * THREADS + 2 * (16 * THREADS + THREADS) + 3 * (THREADS / 64); }

// MEMSIDE: the counts are read with device-scope atomic loads -- the scan runs as the tail of the kernel whose other workgroups
// ADDED to them (returning atomics, performed at the memory side), not behind a kernel boundary.
template <int THREADS, bool MEMSIDE = false>
__device__ __forceinline__ void tile_scan_body(const TileScanArgs &ts, uint32_t *const lds)
{
	constexpr int RUN = 16;
	constexpr int CHUNK = RUN * THREADS;
	constexpr int PADDED = CHUNK + THREADS;
	const int T = ts.T;
	uint32_t *const tile_count = ts.tile_count; uint2 *const ranges = ts.ranges;
	uint32_t *const tile_order = ts.tile_order, *const render_items = ts.render_items;
	const float *const tile_blend = ts.tile_blend;
	constexpr int NW = THREADS / 64;
	uint32_t *const hist = lds;                          // [FR_SCAN_ROWS * THREADS]
	uint32_t *const la = hist + FR_SCAN_ROWS * THREADS;  // [PADDED] counts (bit 31: two-level tile), later: the tile at each place of the order
	uint32_t *const lb = la + PADDED;                    // [PADDED] list starts, later: the item offset of each place
	uint32_t *const wave_sum = lb + PADDED;              // [NW]
	uint32_t *const wave_hist = wave_sum + NW;           // [NW]
	uint32_t *const wave_max = wave_hist + NW;           // [NW]
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
#ifdef FR_SCAN_TIMERS
	const uint64_t tm0 = wall_clock64(); uint64_t tm1 = 0, tm2 = 0, tm3 = 0, tm4 = 0, tm5 = 0, tm6 = 0;
#define TMS(x) x = wall_clock64()
#else
#define TMS(x)
#endif
#pragma unroll
	for (int r = 0; r < FR_SCAN_ROWS; r++) hist[r * THREADS + tid] = 0;
	const int nchunks = (T + CHUNK - 1) / CHUNK;
	uint32_t cnt[RUN];
	uint32_t vmax = 0, carry = 0;
	// coalesced: counts and flags of chunk `base` -> la
	auto load_chunk = [&](const int base) {
		uint32_t v[RUN]; float bl[RUN];
#pragma unroll
		// (clamped addresses, not predicated loads: all 32 in flight together)
		for (int k = 0; k < RUN; k++)
		{
			const int i = base + k * THREADS + tid;
			v[k] = MEMSIDE ? __hip_atomic_load(tile_count + min(i, T - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tile_count[min(i, T - 1)];
		}
		const float *const blend = tile_blend ? tile_blend : (const float *)tile_count;
#pragma unroll
		for (int k = 0; k < RUN; k++) { const int i = base + k * THREADS + tid; bl[k] = blend[min(i, T - 1)]; }
#pragma unroll
		for (int k = 0; k < RUN; k++)
		{
			const int j = k * THREADS + tid, i = base + j;
			la[j + (j >> 4)] = i < T ? (v[k] | ((tile_blend && bl[k] != 0.0f) ? 0x80000000u : 0u)) : 0u;
		}
	};
#pragma unroll 1
	for (int c = 0; c < nchunks; c++)
	{
		const int base = c * CHUNK;
		load_chunk(base);
		__syncthreads();
		TMS(tm1);
		const int t0 = base + tid * RUN;
		uint32_t mine = 0;
#pragma unroll
		for (int kk = 0; kk < RUN; kk++)
		{
			cnt[kk] = la[tid * (RUN + 1) + kk];
			const uint32_t v = cnt[kk] & 0x7fffffffu;
			mine += v;
			vmax = max(vmax, v);
		}
		// (a chain of LDS read-modify-writes; ranking the run's tiles in registers instead -- 240 compares -- took 5 x as long)
#pragma unroll
		for (int kk = 0; kk < RUN; kk++)
			if (t0 + kk < T) hist[(FR_SCAN_BINS - 1 - scan_bin(cnt[kk] & 0x7fffffffu)) * THREADS + tid] += (cnt[kk] >> 31) ? 0x10001u : 1u;
		TMS(tm2);
		uint32_t s = mine; // inclusive scan inside the wave
#pragma unroll
		for (int off = 1; off < 64; off <<= 1)
		{
			const uint32_t n = __shfl_up(s, off);
			if (lane >= off) s += n;
		}
		if (lane == 63) wave_sum[wid] = s;
		__syncthreads();
		uint32_t wave_off = 0, chunk_total = 0;
#pragma unroll
		for (int w = 0; w < NW; w++) { const uint32_t ws = wave_sum[w]; if (w < wid) wave_off += ws; chunk_total += ws; }
		uint32_t run = carry + wave_off + s - mine;
#pragma unroll
		for (int kk = 0; kk < RUN; kk++) { lb[tid * (RUN + 1) + kk] = run; run += cnt[kk] & 0x7fffffffu; }
		carry += chunk_total;
		__syncthreads();
#pragma unroll
		for (int k = 0; k < RUN; k++)
		{
			const int j = k * THREADS + tid, i = base + j;
			if (i < T)
			{
				const uint32_t v = la[j + (j >> 4)] & 0x7fffffffu, st = lb[j + (j >> 4)];
				ranges[i] = v ? make_uint2(st, st + v) : make_uint2(0u, 0u); // empty tiles stay (0,0) like the reference's memset
			}
		}
		__syncthreads(); // (la, lb, wave_sum are reused)
	}
	TMS(tm3);
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) vmax = max(vmax, (uint32_t)__shfl_xor(vmax, off));
	if (lane == 0) wave_max[wid] = vmax;
	// the histogram rows, longest lists first, as one sequence: exclusive scan in place (thread j takes words 19 j ... 19 j + 18)
	uint32_t hv[FR_SCAN_ROWS], hsum = 0;
#pragma unroll
	for (int e = 0; e < FR_SCAN_ROWS; e++) { hv[e] = hist[tid * FR_SCAN_ROWS + e]; hsum += hv[e]; }
	uint32_t hs = hsum;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1)
	{
		const uint32_t n = __shfl_up(hs, off);
		if (lane >= off) hs += n;
	}
	if (lane == 63) wave_hist[wid] = hs;
	__syncthreads();
	uint32_t hoff = 0, htotal = 0, longest = 0;
#pragma unroll
	for (int w = 0; w < NW; w++) { const uint32_t ws = wave_hist[w]; if (w < wid) hoff += ws; htotal += ws; longest = max(longest, wave_max[w]); }
	uint32_t hrun = hoff + hs - hsum;
#pragma unroll
	for (int e = 0; e < FR_SCAN_ROWS; e++) { hist[tid * FR_SCAN_ROWS + e] = hrun; hrun += hv[e]; }
	__syncthreads();
	TMS(tm4);
	if (tid == 0)
	{
		// row r starts at word r x THREADS and holds the lists of bin 17 - r: what lies before the row of bin 11 has >= 2048 entries, ...
		const uint32_t before11 = hist[(FR_SCAN_BINS - 1 - 11) * THREADS] & 0xffffu; // lists with >= 2048 entries
		const uint32_t before12 = hist[(FR_SCAN_BINS - 1 - 12) * THREADS] & 0xffffu; // ... >= 4096
		const uint32_t before9 = hist[(FR_SCAN_BINS - 1 - 9) * THREADS] & 0xffffu;   // ... >= 512
		const uint32_t before13 = hist[(FR_SCAN_BINS - 1 - 13) * THREADS] & 0xffffu; // ... >= 8192
		const uint32_t before14 = hist[(FR_SCAN_BINS - 1 - 14) * THREADS] & 0xffffu; // ... >= 16384
		publish_totals(ts, carry, longest, before11, before12, before9 - before11, 2u * ((htotal & 0xffffu) + (htotal >> 16)), before13, before14);
	}
	TMS(tm5);
	// longest-processing-time-first order for the per-tile kernels (sort, blend): a frame's critical
	// path is its longest tile list, so those workgroups must start first. tile_count doubles as the
	// emission cursor in the global-atomics fallback and is reset here.
	const bool staged = nchunks == 1;
#pragma unroll 1
	for (int c = 0; c < nchunks; c++)
	{
		const int base = c * CHUNK, t0 = base + tid * RUN;
		if (!staged)
		{
			load_chunk(base);
			__syncthreads();
#pragma unroll
			for (int kk = 0; kk < RUN; kk++) cnt[kk] = la[tid * (RUN + 1) + kk];
			__syncthreads();
		}
#pragma unroll
		for (int k = 0; k < RUN; k++) { const int i = base + k * THREADS + tid; if (i < T) tile_count[i] = 0; }
#pragma unroll
		for (int kk = 0; kk < RUN; kk++)
		{
			const int i = t0 + kk;
			if (i < T)
			{
				const uint32_t v = cnt[kk] & 0x7fffffffu, two = cnt[kk] >> 31;
				uint32_t *const cur = &hist[(FR_SCAN_BINS - 1 - scan_bin(v)) * THREADS + tid];
				const uint32_t p = *cur;
				*cur = p + (two ? 0x10001u : 1u);
				const uint32_t place = p & 0xffffu, item = 2u * (place + (p >> 16));
				if (staged) { la[place] = (uint32_t)i | two << 31; lb[place] = item; }
				else
				{
					tile_order[place] = (uint32_t)i;
					uint32_t *it = render_items + item;
					it[0] = (uint32_t)i << 3 | two << 2; it[1] = (uint32_t)i << 3 | two << 2 | 1u;
					if (two) { it[2] = (uint32_t)i << 3 | 4u | 2u; it[3] = (uint32_t)i << 3 | 4u | 2u | 1u; }
				}
			}
		}
	}
	if (staged)
	{
		__syncthreads();
#pragma unroll
		for (int k = 0; k < RUN; k++)
		{
			const int q = k * THREADS + tid;
			if (q < T)
			{
				const uint32_t w = la[q], i = w & 0x7fffffffu, two = w >> 31;
				tile_order[q] = i;
				uint2 *it = (uint2 *)(render_items + lb[q]); // (item offsets are even)
				it[0] = make_uint2(i << 3 | two << 2, i << 3 | two << 2 | 1u);
				if (two) it[1] = make_uint2(i << 3 | 4u | 2u, i << 3 | 4u | 2u | 1u);
			}
		}
	}
#ifdef FR_SCAN_TIMERS
	TMS(tm6);
	if (tid == 0 && (ts.seq & 63u) == 0u)
		printf("tile scan (10 ns ticks): loads %d count %d scan+ranges %d hist %d publish %d place %d\n", (int)(tm1 - tm0), (int)(tm2 - tm1), (int)(tm3 - tm2), (int)(tm4 - tm3), (int)(tm5 - tm4), (int)(tm6 - tm5));
#endif
#undef TMS
}

// More than 65535 tiles (8K images): counters of their own per bin, LDS atomics (round 2's scan). Never fused into k_bin.
template <int THREADS>
__device__ __forceinline__ void tile_scan_atomics(const TileScanArgs &ts)
{
	const int T = ts.T;
	uint32_t *const tile_count = ts.tile_count; uint2 *const ranges = ts.ranges;
	uint32_t *const tile_order = ts.tile_order, *const render_items = ts.render_items;
	const float *const tile_blend = ts.tile_blend;
	constexpr int NW = THREADS / 64;
	__shared__ uint32_t bucket[FR_SCAN_BINS];
	__shared__ uint32_t ibucket[FR_SCAN_BINS];
	__shared__ uint32_t wave_sum[NW];
	__shared__ uint32_t wave_max[NW];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	if (tid < FR_SCAN_BINS) { bucket[tid] = 0; ibucket[tid] = 0; }
	__syncthreads();
	uint32_t vmax = 0, mine = 0;
	const int per = (T + THREADS - 1) / THREADS;
	const int t0 = tid * per, t1 = min(T, t0 + per);
	for (int i = t0; i < t1; i++)
	{
		const uint32_t v = tile_count[i];
		mine += v;
		vmax = max(vmax, v);
		atomicAdd(&bucket[scan_bin(v)], 1u);
		atomicAdd(&ibucket[scan_bin(v)], (tile_blend && tile_blend[i] != 0.0f) ? 4u : 2u);
	}
	uint32_t s = mine;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1)
	{
		const uint32_t n = __shfl_up(s, off);
		if (lane >= off) s += n;
	}
	if (lane == 63) wave_sum[wid] = s;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) vmax = max(vmax, (uint32_t)__shfl_xor(vmax, off));
	if (lane == 0) wave_max[wid] = vmax;
	__syncthreads();
	uint32_t wave_off = 0, block_total = 0, longest = 0;
	for (int w = 0; w < NW; w++) { const uint32_t ws = wave_sum[w]; if (w < wid) wave_off += ws; block_total += ws; longest = max(longest, wave_max[w]); }
	uint32_t run = wave_off + s - mine;
	for (int i = t0; i < t1; i++)
	{
		const uint32_t v = tile_count[i];
		ranges[i] = v ? make_uint2(run, run + v) : make_uint2(0u, 0u);
		run += v;
	}
	__syncthreads();
	if (tid == 0)
	{
		uint32_t h4 = 0;
		for (int b = 12; b < FR_SCAN_BINS; b++) h4 += bucket[b];
		const uint32_t h8 = h4 - bucket[12], mid = bucket[10] + bucket[11];
		uint32_t h16 = 0, h32 = 0;
		for (int b = 14; b < FR_SCAN_BINS; b++) h16 += bucket[b];
		for (int b = 15; b < FR_SCAN_BINS; b++) h32 += bucket[b];
		uint32_t nitems = 0, run2 = 0;
		for (int b = FR_SCAN_BINS - 1; b >= 0; b--) { const uint32_t c = ibucket[b]; ibucket[b] = nitems; nitems += c; }
		for (int b = FR_SCAN_BINS - 1; b >= 0; b--) { const uint32_t c = bucket[b]; bucket[b] = run2; run2 += c; }
		publish_totals(ts, block_total, longest, h4, h8, mid, nitems, h16, h32);
	}
	__syncthreads();
	for (int i = t0; i < t1; i++)
	{
		const uint32_t v = tile_count[i];
		const uint32_t two = (tile_blend && tile_blend[i] != 0.0f) ? 1u : 0u;
		tile_order[atomicAdd(&bucket[scan_bin(v)], 1u)] = (uint32_t)i;
		tile_count[i] = 0;
		uint32_t *it = render_items + atomicAdd(&ibucket[scan_bin(v)], two ? 4u : 2u);
		it[0] = (uint32_t)i << 3 | two << 2; it[1] = (uint32_t)i << 3 | two << 2 | 1u;
		if (two) { it[2] = (uint32_t)i << 3 | 4u | 2u; it[3] = (uint32_t)i << 3 | 4u | 2u | 1u; }
	}
}

} // namespace fr
