// pga_sort_wave.h -- the same permutation as pga_sort_exact.h (minimap2's unstable radix_sort_128x,
// ksort.h:101-151), executed by a whole wavefront (single-wave workgroups only).
//
// The cycle-leader walk of one level is sequential by definition; what the wave buys:
//   * levels whose digit is the same in every key of the array are skipped outright (one OR/AND pass finds them):
//     such a level leaves every run untouched (all records "home", ksort.h:132);
//   * the digit histogram (LDS atomics, 64 records per step);
//   * the long stretches of records that are already home: 64 records are tested per step, so an almost sorted
//     array -- anchors of a co-linear alignment, chain scores along a chain -- costs ~n/64 steps there;
//   * displacement cycles run against LDS: every bucket's head only moves forward and every slot at a head is
//     read once and written once, so each bucket gets a RS_WIN-record write-back window in LDS (filled and
//     flushed with coalesced accesses); one step of a cycle is an LDS exchange instead of a global round trip;
//   * insertion sorts of runs of <= 64 records (ksort.h:142) happen in LDS, one lane per run.
// Every lane of the wave must call these functions with identical arguments.
#pragma once
#include "pga_common.h"
#include "pga_sort_exact.h"

namespace pga {

#define RS_WIN 16
#define RS_NONE 0xffffffffu

struct __attribute__((aligned(16))) RsLds {
	u128 win[256 * RS_WIN];      // per-bucket window over [wbase, wbase+RS_WIN)
	u128 ins[64];                // insertion-sort staging
	uint32_t head[256], tail[256], wbase[256];
	unsigned long long prof[4];  // ticks: varying-bit pass, histograms, walks, run windows (diagnostics)
};

__device__ __forceinline__ u128 ld128(const u128 *p) { u128 v; v.x = p->x; v.y = p->y; return v; }
__device__ __forceinline__ int rl32(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
__device__ __forceinline__ void rs_fence_wave() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
__device__ __forceinline__ void rs_fence_wg() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// write the consumed part of bucket d's window back (positions [wbase, head))
__device__ __forceinline__ void rs_flush(u128 *beg, RsLds &L, int d, int lane)
{
	const uint32_t wb = L.wbase[d];
	if (wb == RS_NONE) return;
	const uint32_t cnt = L.head[d] - wb;
	if ((uint32_t)lane < cnt) beg[wb + lane] = L.win[d * RS_WIN + lane];
	rs_fence_wave();
	if (lane == 0) L.wbase[d] = RS_NONE;
	rs_fence_wave();
}

// one level (ksort.h:118-146) on [beg, beg+n)
__device__ inline bool rs_level_wave(u128 *beg, int64_t n, int shift, RsLds &L, int lane)
{
	const unsigned long long t0 = wall_clock64();
	for (int d = lane; d < 256; d += 64) L.head[d] = 0, L.wbase[d] = RS_NONE;
	rs_fence_wg();
	const uint32_t first = (uint32_t)((beg[0].x >> shift) & 255);
	bool diff = false;
	for (int64_t i = lane; i < n; i += 64) {
		const uint32_t d = (uint32_t)((beg[i].x >> shift) & 255);
		atomicAdd(&L.head[d], 1u);
		diff |= d != first;
	}
	rs_fence_wg();
	const unsigned long long t1 = wall_clock64();
	if (lane == 0) L.prof[1] += t1 - t0;
	if (!__ballot(diff)) return false;                         // one bucket: the walk is the identity
	if (lane == 0) { uint32_t pos = 0; for (int d = 0; d < 256; ++d) { const uint32_t c = L.head[d]; L.head[d] = pos; pos += c; L.tail[d] = pos; } }
	rs_fence_wg();
	for (int d = 0; d < 256; ++d) {
		uint32_t h = L.head[d]; const uint32_t tl = L.tail[d];
		rs_flush(beg, L, d, lane);                              // slots filled while d was a destination
		// bucket d is scanned through a 64-record register window (slots at or beyond a head still hold the original
		// records, and nothing but this loop writes them while d is the current bucket)
		uint32_t cw = h; unsigned long long fm = 0; bool have = false, dirty = false;
		u128 rec; rec.x = 0, rec.y = 0;
		while (h < tl) {
			if (!have || h - cw >= 64) {
				if (dirty) beg[cw + (uint32_t)lane] = rec;
				cw = h; dirty = false; have = true;
				const uint32_t pos = cw + (uint32_t)lane;
				if (pos < tl) rec = ld128(&beg[pos]);
				fm = __ballot(pos < tl && (uint32_t)((rec.x >> shift) & 255) != (uint32_t)d);
			}
			const unsigned long long m = fm >> (h - cw);             // foreign records at positions >= h
			if (m == 0) { h = cw + 64; continue; }
			h += (uint32_t)(__ffsll((long long)m) - 1);
			// displacement chain starting at the foreign record beg[h] (all lanes follow it; lane 0 owns the LDS writes)
			const int sl = (int)(h - cw);
			u128 carry;
			carry.x = (uint64_t)(uint32_t)rl32((int)(uint32_t)rec.x, sl) | (uint64_t)(uint32_t)rl32((int)(uint32_t)(rec.x >> 32), sl) << 32;
			carry.y = (uint64_t)(uint32_t)rl32((int)(uint32_t)rec.y, sl) | (uint64_t)(uint32_t)rl32((int)(uint32_t)(rec.y >> 32), sl) << 32;
			int dst = (int)((carry.x >> shift) & 255);
			do {
				const uint32_t hd = L.head[dst];
				uint32_t wb = L.wbase[dst];
				if (wb == RS_NONE || hd - wb >= RS_WIN) {
					rs_flush(beg, L, dst, lane);
					const uint32_t p2 = hd + (uint32_t)lane;
					if (lane < RS_WIN && p2 < L.tail[dst]) L.win[dst * RS_WIN + lane] = ld128(&beg[p2]);
					if (lane == 0) L.wbase[dst] = hd;
					wb = hd;
					rs_fence_wave();
				}
				const u128 nxt = L.win[dst * RS_WIN + (hd - wb)];
				rs_fence_wave();
				if (lane == 0) { L.win[dst * RS_WIN + (hd - wb)] = carry; L.head[dst] = hd + 1; }
				rs_fence_wave();
				carry = nxt;
				dst = (int)((carry.x >> shift) & 255);
			} while (dst != d);
			if (lane == sl) rec = carry, dirty = true;
			++h;
		}
		if (dirty) beg[cw + (uint32_t)lane] = rec;
		if (lane == 0) L.head[d] = h;
		rs_fence_wave();
	}
	rs_fence_wg();
	if (lane == 0) L.prof[2] += wall_clock64() - t1;
	return true;
}

// insertion-sort up to 64 records [b, e) through LDS: lane 0 sorts
__device__ __forceinline__ void rs_small_wave(u128 *beg, int64_t b, int64_t e, RsLds &L, int lane)
{
	const int m = (int)(e - b);
	if (m <= 1) return;
	if (lane < m) L.ins[lane] = ld128(&beg[b + lane]);
	rs_fence_wave();
	if (lane == 0) rs_insertion(L.ins, L.ins + m);
	rs_fence_wave();
	if (lane < m) beg[b + lane] = L.ins[lane];
	rs_fence_wg();
}

// Runs of records that agree on x >> hi_shift, found 64 records at a time: runs of <= 64 records are insertion-sorted
// here (in LDS, one lane per run; ksort.h:142), longer ones are handed to big(offset, length).
template <class Big>
__device__ inline void rs_runs_wave(u128 *beg, int64_t n, int hi_shift, RsLds &L, int lane, Big big)
{
	int64_t rb = 0;
	while (rb < n) {
		// a window of 64 records starting at rb: which of them start a new run?
		const int64_t pos = rb + lane;
		const bool in = pos < n;
		u128 rec; rec.x = ~0ULL, rec.y = 0;
		if (in) rec = ld128(&beg[pos]);
		const uint64_t hk = in ? (hi_shift >= 64 ? 0ULL : rec.x >> hi_shift) : ~0ULL;
		const uint64_t hi0 = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(hk >> 32), 0) << 32) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)hk, 0);
		const uint32_t hp_lo = (uint32_t)__shfl((int)(uint32_t)(hk & 0xffffffffULL), lane > 0 ? lane - 1 : 0), hp_hi = (uint32_t)__shfl((int)(uint32_t)(hk >> 32), lane > 0 ? lane - 1 : 0);
		const uint64_t hp = lane == 0 ? hi0 : ((uint64_t)hp_hi << 32 | (uint64_t)hp_lo);
		const bool start = in && lane > 0 && hk != hp;
		const unsigned long long sm = __ballot(start);
		const int n_in = (int)(n - rb < 64 ? n - rb : 64);
		const bool last_complete = (rb + n_in >= n);
		const unsigned long long starts = sm | 1ULL;                               // bit 0: the run at rb
		const int n_runs = __popcll(starts);
		const int n_sort = last_complete ? n_runs : n_runs - 1;
		if (n_sort == 0) {
			// a single run that continues beyond the window: find its end
			int64_t re = rb + n_in;
			for (;;) {
				const int64_t p2 = re + lane;
				const bool brk = p2 >= n || (hi_shift >= 64 ? 0ULL : beg[p2 < n ? p2 : n - 1].x >> hi_shift) != hi0;
				const unsigned long long bm = __ballot(brk);
				if (bm) { re += __ffsll((long long)bm) - 1; break; }
				re += 64;
			}
			if (re - rb > 64) big(rb, re - rb);
			else rs_small_wave(beg, rb, re, L, lane);
			rb = re;
			continue;
		}
		// complete runs inside the window: run r spans [s_r, s_{r+1}); the last run of the window is complete only
		// if the window reaches the end of the array, otherwise it restarts the next window.  Lane r sorts run r.
		int my_b = -1, my_e = -1; int64_t next_rb = rb + n_in;
		{
			unsigned long long mrest = starts; int r = 0, prev = -1;
			while (mrest) {
				const int bpos = __ffsll((long long)mrest) - 1;
				mrest &= mrest - 1;
				if (prev >= 0 && r - 1 == lane) my_b = prev, my_e = bpos;
				prev = bpos; ++r;
			}
			if (last_complete) { if (n_runs - 1 == lane) my_b = prev, my_e = n_in; }
			else next_rb = rb + prev;
		}
		// only windows holding an out-of-order run are rewritten
		const uint32_t xlo = (uint32_t)__shfl((int)(uint32_t)rec.x, lane > 0 ? lane - 1 : 0), xhi = (uint32_t)__shfl((int)(uint32_t)(rec.x >> 32), lane > 0 ? lane - 1 : 0);
		const uint64_t xprev = (uint64_t)xhi << 32 | xlo;
		const bool desc = in && lane > 0 && !start && rec.x < xprev && pos < next_rb;   // inside a complete run, smaller than its left neighbour
		if (__ballot(desc)) {
			L.ins[lane] = rec;
			rs_fence_wave();
			if (my_b >= 0 && my_e - my_b > 1) rs_insertion(L.ins + my_b, L.ins + my_e);
			rs_fence_wave();
			if (pos < next_rb) beg[pos] = L.ins[lane];
			rs_fence_wg();
		}
		rb = next_rb;
	}
}

// bits that differ somewhere in [beg, beg+n)
__device__ inline uint64_t rs_varying_bits(const u128 *beg, int64_t n, int lane)
{
	uint64_t o = 0, a = ~0ULL;
	for (int64_t i = lane; i < n; i += 64) { const uint64_t x = beg[i].x; o |= x; a &= x; }
	uint32_t olo = (uint32_t)o, ohi = (uint32_t)(o >> 32), alo = (uint32_t)a, ahi = (uint32_t)(a >> 32);
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		olo |= (uint32_t)__shfl_xor((int)olo, d); ohi |= (uint32_t)__shfl_xor((int)ohi, d);
		alo &= (uint32_t)__shfl_xor((int)alo, d); ahi &= (uint32_t)__shfl_xor((int)ahi, d);
	}
	return ((uint64_t)ohi << 32 | olo) ^ ((uint64_t)ahi << 32 | alo);
}

// the whole sort by one wave (small arrays, e.g. the chains of one query)
__device__ inline void radix_sort_128x_wave(u128 *beg, int64_t n, RsLds &L, int lane)
{
	if (n <= 64) { rs_small_wave(beg, 0, n, L, lane); return; }
	const uint64_t vary = rs_varying_bits(beg, n, lane);
	bool single_run = true;                                   // no level above has split the array yet
	for (int shift = 56; shift >= 0; shift -= 8) {
		if (((vary >> shift) & 255) == 0) continue;            // identity at this level for every run
		if (single_run) { rs_level_wave(beg, n, shift, L, lane); single_run = false; continue; }
		rs_runs_wave(beg, n, shift + 8, L, lane, [&](int64_t rb, int64_t len) { rs_level_wave(beg + rb, len, shift, L, lane); });
	}
}

} // namespace pga
