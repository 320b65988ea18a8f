// The per-tile depth sort's device code (kernels: binning.hip).
#pragma once
#include "common.h"

namespace fr {

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

// Kernels launched ahead of the frame's counts (fr_forward) leave without touching anything when the frame does not fit
// what they were sized for -- more instances than the binning workspace holds, or more blend work items than the blend
// grid has workgroups; the host then replays the whole stage.
struct SpecLimits { uint32_t capacity, items_cap; };
__device__ __forceinline__ bool frame_fits(const uint32_t *totals, const SpecLimits lim) { return totals[0] <= lim.capacity && totals[5] <= lim.items_cap; }


// LDS slot of key i: one spare slot after every ITEMS keys. A thread owns ITEMS consecutive keys, i.e. lanes are
// 8 ITEMS bytes apart: unpadded, the 64 lanes of an access fall on a few of the 32 eight-byte bank pairs (16-way conflict
// at ITEMS = 8); with the spare slot the lane stride is 8 (ITEMS + 1) bytes and all bank pairs are used.
template <int ITEMS> __device__ __forceinline__ int sk_slot(const int i) { return i + i / ITEMS; }

// Merge sort of the n keys in sk (slots sk_slot(0 .. nact - 1), nact = ITEMS x the power of two of threads that covers n,
// slots n .. nact - 1 hold ~0) by the whole workgroup. Every thread sorts ITEMS consecutive keys
// in registers (odd-even transposition network), then log2(nact / ITEMS) merge passes follow: a thread
// finds its ITEMS-long slice of the merged output by a merge-path binary search and merges it
// sequentially out of LDS into registers; results are written back in place after a barrier.
template <int THREADS, int ITEMS>
__device__ __forceinline__ int msort_active(const int n)
{
	int runs = 1;
	while (runs * ITEMS < n) runs <<= 1;
	return runs * ITEMS;
}
template <int THREADS, int ITEMS>
__device__ __forceinline__ void msort_lds(uint64_t *sk, const int nact)
{
	const int tid = threadIdx.x;
#define SK(i) sk[sk_slot<ITEMS>(i)]
	const bool act = tid * ITEMS < nact;
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
#undef A
#undef B
		}
		__syncthreads();
		if (act)
		{
#pragma unroll
			for (int t = 0; t < ITEMS; t++) SK(o + t) = k[t];
		}
		__syncthreads();
	}
#undef SK
}

// Sort n <= THREADS x ITEMS keys held in registers (key[t] = key number tid + t x THREADS of the list, ~0 beyond n) into
// sk (slots sk_slot(0 .. n - 1)) by the whole workgroup. The merge sort above pays ~3 us per pass whatever n is (a binary
// search and ITEMS dependent LDS reads per thread and pass): 15-25 us for a list, the latency floor of the sort stage. Depths of
// one tile's list are spread out, so an INTERPOLATION sort gets nearly there in one step: bucket = floor((depth bits -
// min) x NB / range) with NB = twice the capacity (monotone in the depth; equal depths share a bucket), an LDS
// histogram whose returning atomics also hand every key its rank inside its bucket, a scan, and every key goes straight to
// bucket start + rank. What is left is the order INSIDE the buckets: `occ` = the fullest bucket's count rounds of odd-even
// transposition over the whole array finish it (a run of k elements is sorted after k alternating rounds whatever the
// parity it starts on; pairs across a bucket boundary are already in order). occ is 2..8 for the lists of the S-6M
// frames; a list with a fuller bucket than FR_BUCKET_SORT_MAX_OCC (a wall of splats at one depth) takes the merge sort
// from where the scatter left it. The histogram lives in sk (the keys are in registers until it has been read).
#ifndef FR_BUCKET_SORT_MAX_OCC
#define FR_BUCKET_SORT_MAX_OCC 40
#endif
template <int THREADS, int ITEMS>
__device__ __forceinline__ void sort_keys_lds(const uint64_t (&key)[ITEMS], const int n, uint64_t *sk)
{
	constexpr int CAP = THREADS * ITEMS, NB = 2 * CAP, NW = THREADS / 64, PER = NB / THREADS;
	uint32_t *const hist = (uint32_t *)sk; // NB words = CAP key slots
	__shared__ uint32_t s_min[NW], s_max[NW], s_wave[NW], s_occ[NW];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	uint32_t dmin = 0xffffffffu, dmax = 0u;
#pragma unroll
	for (int t = 0; t < ITEMS; t++)
		if (tid + t * THREADS < n) { const uint32_t d = (uint32_t)(key[t] >> 32); dmin = min(dmin, d); dmax = max(dmax, d); }
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) { dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, off)); dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, off)); }
	if (lane == 0) { s_min[wid] = dmin; s_max[wid] = dmax; }
	for (int b = tid; b < NB; b += THREADS) hist[b] = 0;
	__syncthreads();
#pragma unroll
	for (int w = 0; w < NW; w++) { dmin = min(dmin, s_min[w]); dmax = max(dmax, s_max[w]); }
	const float inv = (float)NB / ((float)(dmax - dmin) + 1.0f);
	auto bucket_of = [&](const uint64_t k) { return min((uint32_t)(NB - 1), (uint32_t)((float)((uint32_t)(k >> 32) - dmin) * inv)); };
	uint32_t rk[ITEMS]; // rank inside the bucket, then the key's place
#pragma unroll
	for (int t = 0; t < ITEMS; t++)
		if (tid + t * THREADS < n) rk[t] = atomicAdd(&hist[bucket_of(key[t])], 1u);
	__syncthreads();
	// exclusive scan of the bucket counts (PER consecutive buckets per thread, read twice: no registers held), fullest bucket
	uint32_t sum = 0, occ = 0;
#pragma unroll
	for (int k = 0; k < PER; k++) { const uint32_t c = hist[tid * PER + k]; sum += c; occ = max(occ, c); }
	uint32_t sc = sum;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)sc, off); if (lane >= off) sc += v; }
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) occ = max(occ, (uint32_t)__shfl_xor((int)occ, off));
	if (lane == 63) s_wave[wid] = sc;
	if (lane == 0) s_occ[wid] = occ;
	__syncthreads();
	uint32_t run = sc - sum;
#pragma unroll
	for (int w = 0; w < NW; w++) { if (w < wid) run += s_wave[w]; occ = max(occ, s_occ[w]); }
#pragma unroll
	for (int k = 0; k < PER; k++) { const uint32_t c = hist[tid * PER + k]; hist[tid * PER + k] = run; run += c; }
	__syncthreads();
#pragma unroll
	for (int t = 0; t < ITEMS; t++)
		if (tid + t * THREADS < n) rk[t] += hist[bucket_of(key[t])];
	__syncthreads(); // the histogram has been read: the keys take its place
#pragma unroll
	for (int t = 0; t < ITEMS; t++)
		if (tid + t * THREADS < n) sk[sk_slot<ITEMS>((int)rk[t])] = key[t];
	if (occ > (uint32_t)FR_BUCKET_SORT_MAX_OCC)
	{
		const int nact = msort_active<THREADS, ITEMS>(n);
		for (int i = n + tid; i < nact; i += THREADS) sk[sk_slot<ITEMS>(i)] = ~0ull;
		__syncthreads();
		msort_lds<THREADS, ITEMS>(sk, nact);
		return;
	}
	__syncthreads();
	if (occ < 2u) return;
	for (uint32_t r = 0; r < occ; r++)
	{
		for (int p = (int)(r & 1u) + 2 * tid; p + 1 < n; p += 2 * THREADS)
		{
			const int ia = sk_slot<ITEMS>(p), ib = sk_slot<ITEMS>(p + 1);
			const uint64_t x = sk[ia], y = sk[ib];
			if (x > y) { sk[ia] = y; sk[ib] = x; }
		}
		__syncthreads();
	}
}

// One list of n keys (entries + rg.x ..) sorted into point_list by the whole workgroup; LDS (sk) holds THREADS x ITEMS keys
// (+ the spare slots). fallback: a list that does not fit is sorted in place in global memory by the bitonic network (chunks
// of a split list with thousands of equal depths); otherwise such a list is left to another kernel.
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
	uint64_t key[ITEMS];
#pragma unroll
	for (int t = 0; t < ITEMS; t++) { const int i = tid + t * THREADS; key[t] = i < n ? src[i] : ~0ull; }
	sort_keys_lds<THREADS, ITEMS>(key, n, sk);
	for (int i = tid; i < n; i += THREADS) dst[i] = (uint32_t)sk[sk_slot<ITEMS>(i)];
}

// ---- long lists: regrouped by depth ------------------------------------------------------------------------------
// Long tile lists (>= FR_SORT_SPLIT_MIN entries) are not sorted as one piece: a handful of them used to occupy one
// CU each for 50-80 us with sixteen-way merge passes while the rest of the chip had nothing left to do. A counting
// pass on the depth bits (a fixed monotone quantisation into FR_SORT_FINE_BUCKETS buckets) regroups the list into chunks of
// ~FR_SORT_CHUNK_TARGET entries with disjoint, increasing depth ranges; equal depths share a bucket, so sorting every
// chunk by (depth, id) sorts the list.
#define FR_SPLIT_THREADS 1024
#define FR_SPLIT_REGS 16 // keys per thread held in registers by split_list (lists up to 16384 entries)
// depth bucket: the bit pattern of a positive float orders like its value, so a fixed monotone map of the bits needs no
// pass over the list: 128 buckets per octave from the near plane (0.2) up, 16 octaves, the rest clamped into the last
// bucket (uneven buckets are fine, chunks are cut by count)
__device__ __forceinline__ uint32_t depth_bucket(const uint32_t d)
{
	constexpr uint32_t dmin = 0x3E4CCCCDu; // 0.2f
	return min((uint32_t)(FR_SORT_FINE_BUCKETS - 1), (d > dmin ? d - dmin : 0u) >> 16);
}
// LDS of the workgroups (1024 threads) that regroup a list: `keys` has `nkeys` slots (the regrouped keys of a list go
// through them in rounds and leave the workgroup as coalesced copies: scattered straight to global memory, every 8-byte
// store was a partial-sector write of its own -- ~11 cycles of the CU's memory pipeline each, tools/scratch/gather_rate.hip)
struct SplitLDS {
	uint64_t *keys; uint32_t nkeys;
	uint32_t *hist;  // [FR_SORT_FINE_BUCKETS] counts -> exclusive offsets -> scatter cursors
	uint32_t *start; // [FR_SORT_FINE_BUCKETS + 1] compacted chunk starts
	uint32_t *wave, *wave2; // [16] each
	uint32_t *word;  // [4] broadcast words
};
#define FR_SPLIT_LDS_DECL(L, dyn_keys, dyn_nkeys) \
	__shared__ uint32_t L##_hist[FR_SORT_FINE_BUCKETS]; __shared__ uint32_t L##_start[FR_SORT_FINE_BUCKETS + 1]; \
	__shared__ uint32_t L##_wave[16], L##_wave2[16], L##_word[4]; \
	const SplitLDS L = { dyn_keys, dyn_nkeys, L##_hist, L##_start, L##_wave, L##_wave2, L##_word }

// Load the list's keys (up to FR_SPLIT_REGS x 1024: ONCE, all loads in flight together, kept in registers for the
// scatter -- a loop of dependent load -> LDS atomic iterations costs a memory round trip per 1024 keys, and the longest
// list is the critical path of its kernel), count them per depth bucket and turn the counts into exclusive offsets
// (L.hist). All 1024 threads; ends with a barrier.
__device__ __forceinline__ void bucket_offsets(const uint64_t *src, const uint32_t n, const SplitLDS &L, uint64_t (&kreg)[FR_SPLIT_REGS], const bool in_regs)
{
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	for (int b = tid; b < FR_SORT_FINE_BUCKETS; b += FR_SPLIT_THREADS) L.hist[b] = 0;
	__syncthreads();
	if (in_regs)
	{
#pragma unroll
		for (int k = 0; k < FR_SPLIT_REGS; k++) { const uint32_t i = tid + 1024u * k; kreg[k] = i < n ? src[i] : 0ull; }
#pragma unroll
		for (int k = 0; k < FR_SPLIT_REGS; k++)
			if (tid + 1024u * k < n) atomicAdd(&L.hist[depth_bucket((uint32_t)(kreg[k] >> 32))], 1u);
	}
	else
		// (longer lists are streamed, FR_SPLIT_REGS keys per thread in flight at a time)
		for (uint32_t i0 = 0; i0 < n; i0 += FR_SPLIT_REGS * 1024u)
		{
#pragma unroll
			for (int k = 0; k < FR_SPLIT_REGS; k++) { const uint32_t i = i0 + tid + 1024u * k; kreg[k] = i < n ? src[i] : 0ull; }
#pragma unroll
			for (int k = 0; k < FR_SPLIT_REGS; k++)
				if (i0 + tid + 1024u * k < n) atomicAdd(&L.hist[depth_bucket((uint32_t)(kreg[k] >> 32))], 1u);
		}
	__syncthreads();
	// exclusive scan of the 2048 counts (two consecutive buckets per thread)
	static_assert(FR_SORT_FINE_BUCKETS == 2 * FR_SPLIT_THREADS, "two buckets per thread");
	const uint32_t c0 = L.hist[2 * tid], c1 = L.hist[2 * tid + 1];
	uint32_t sc = c0 + c1;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)sc, off); if (lane >= off) sc += v; }
	if (lane == 63) L.wave[wid] = sc;
	__syncthreads();
	uint32_t wave_off = 0;
#pragma unroll
	for (int w = 0; w < 16; w++) if (w < wid) wave_off += L.wave[w];
	const uint32_t e0 = wave_off + sc - (c0 + c1), e1 = e0 + c0; // exclusive offsets of my two buckets
	__syncthreads();
	L.hist[2 * tid] = e0; L.hist[2 * tid + 1] = e1;
	__syncthreads();
}

// Regroup the list rg of `entries` by depth into `entries2` (same range) and append its chunks to the chunk table
// (consecutive slots, in depth order). -> first slot and number of chunks. All 1024 threads.
__device__ __forceinline__ uint2 split_list(const uint2 rg, const uint64_t *entries, uint64_t *entries2, uint2 *chunks, uint32_t *chunk_ctr, const SplitLDS &L)
{
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const uint32_t n = rg.y - rg.x;
	const uint64_t *src = entries + rg.x;
	uint64_t *dst = entries2 + rg.x;
	const bool in_regs = n <= FR_SPLIT_REGS * 1024u;
	uint64_t kreg[FR_SPLIT_REGS];
	bucket_offsets(src, n, L, kreg, in_regs);
	// a chunk starts where the running count crosses a multiple of the target (monotone in the bucket index)
	const uint32_t e0 = L.hist[2 * tid], e1 = L.hist[2 * tid + 1];
	const uint32_t prev = tid == 0 ? 0u : L.hist[2 * tid - 1];
	const bool f0 = tid == 0 || (e0 / FR_SORT_CHUNK_TARGET) != (prev / FR_SORT_CHUNK_TARGET);
	const bool f1 = (e1 / FR_SORT_CHUNK_TARGET) != (e0 / FR_SORT_CHUNK_TARGET);
	uint32_t fs = (f0 ? 1u : 0u) + (f1 ? 1u : 0u);
	const uint32_t mine = fs;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)fs, off); if (lane >= off) fs += v; }
	if (lane == 63) L.wave2[wid] = fs;
	__syncthreads();
	uint32_t foff = 0, nchunks = 0;
#pragma unroll
	for (int w = 0; w < 16; w++) { if (w < wid) foff += L.wave2[w]; nchunks += L.wave2[w]; }
	uint32_t pos = foff + fs - mine;
	if (f0) L.start[pos++] = e0;
	if (f1) L.start[pos] = e1;
	if (tid == 0) { L.start[nchunks] = n; L.word[0] = atomicAdd(chunk_ctr, nchunks); }
	__syncthreads();
	const uint32_t slot = L.word[0];
	for (uint32_t k = tid; k < nchunks; k += 1024) chunks[slot + k] = make_uint2(rg.x + L.start[k], rg.x + L.start[k + 1]);
	// scatter (the offsets become cursors)
	if (in_regs)
	{
		uint32_t at[FR_SPLIT_REGS];
#pragma unroll
		for (int k = 0; k < FR_SPLIT_REGS; k++)
			at[k] = tid + 1024u * k < n ? atomicAdd(&L.hist[depth_bucket((uint32_t)(kreg[k] >> 32))], 1u) : 0xffffffffu;
		for (uint32_t base = 0; base < n; base += L.nkeys)
		{
#pragma unroll
			for (int k = 0; k < FR_SPLIT_REGS; k++)
				if (at[k] - base < L.nkeys) L.keys[at[k] - base] = kreg[k];
			__syncthreads();
			const uint32_t m = min(L.nkeys, n - base);
			for (uint32_t i = tid; i < m; i += 1024) dst[base + i] = L.keys[i];
			__syncthreads();
		}
	}
	else
	{
		for (uint32_t i0 = 0; i0 < n; i0 += FR_SPLIT_REGS * 1024u)
		{
#pragma unroll
			for (int k = 0; k < FR_SPLIT_REGS; k++) { const uint32_t i = i0 + tid + 1024u * k; kreg[k] = i < n ? src[i] : 0ull; }
#pragma unroll
			for (int k = 0; k < FR_SPLIT_REGS; k++)
				if (i0 + tid + 1024u * k < n) dst[atomicAdd(&L.hist[depth_bucket((uint32_t)(kreg[k] >> 32))], 1u)] = kreg[k];
		}
		__syncthreads();
	}
	return make_uint2(slot, nchunks);
}

} // namespace fr
