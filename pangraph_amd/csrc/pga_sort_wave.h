// pga_sort_wave.h -- the same permutation as pga_sort_exact.h (minimap2's unstable radix_sort_128x,
// ksort.h:101-151), executed by a whole wavefront.
//
// The cycle-leader walk of one level is sequential by definition, but three parts of it are not:
//   * the digit histogram (LDS atomics, 64 records per step);
//   * the long runs of records that are already in their bucket ("home": ksort.h:132 just steps over them):
//     the wave tests 64 records per step and jumps to the next displaced one, so an almost-sorted array --
//     anchors of a co-linear alignment, chain scores along a chain -- costs ~n/64 steps instead of n;
//   * the insertion sorts of buckets of <= 64 records (ksort.h:142): independent of each other, one lane each.
// Only the displacement chains themselves are walked one record at a time (uniformly by all lanes, lane 0
// stores).  Every lane of the wave must call these functions with identical arguments.
#pragma once
#include "pga_common.h"
#include "pga_sort_exact.h"

namespace pga {

__device__ __forceinline__ u128 ld128(const u128 *p) { u128 v; v.x = p->x; v.y = p->y; return v; }

// one level (ksort.h:118-146) on [beg, beg+n); head/tail: 256-entry LDS arrays
__device__ inline void rs_level_wave(u128 *beg, int64_t n, int shift, uint32_t *head, uint32_t *tail, int lane)
{
	for (int d = lane; d < 256; d += 64) head[d] = 0;
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	const uint32_t first = (uint32_t)((beg[0].x >> shift) & 255);
	bool diff = false;
	for (int64_t i = lane; i < n; i += 64) {
		const uint32_t d = (uint32_t)((beg[i].x >> shift) & 255);
		atomicAdd(&head[d], 1u);
		diff |= d != first;
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	if (!__ballot(diff)) return;                               // one bucket: the walk is the identity
	if (lane == 0) { uint32_t pos = 0; for (int d = 0; d < 256; ++d) { const uint32_t c = head[d]; head[d] = pos; pos += c; tail[d] = pos; } }
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	for (int d = 0; d < 256; ++d) {
		uint32_t h = head[d]; const uint32_t tl = tail[d];
		while (h < tl) {
			// jump over records that are already home
			const uint32_t pos = h + (uint32_t)lane;
			const bool foreign = pos >= tl || (uint32_t)((beg[pos < tl ? pos : tl - 1].x >> shift) & 255) != (uint32_t)d;
			const unsigned long long m = __ballot(foreign);
			if (m == 0) { h += 64; continue; }
			h += (uint32_t)(__ffsll((long long)m) - 1);
			if (h >= tl) break;
			// displacement chain starting at the foreign record beg[h] (all lanes follow it; lane 0 writes)
			u128 carry = ld128(&beg[h]);
			int dst = (int)((carry.x >> shift) & 255);
			do {
				const uint32_t hd = head[dst];
				const u128 nxt = ld128(&beg[hd]);
				if (lane == 0) { beg[hd] = carry; head[dst] = hd + 1; }
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");            // LDS is in order per wave; each record is read before it is overwritten
				carry = nxt;
				dst = (int)((carry.x >> shift) & 255);
			} while (dst != d);
			if (lane == 0) beg[h] = carry;
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			++h;
		}
		if (lane == 0) head[d] = h;
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

__device__ inline void radix_sort_128x_wave(u128 *beg, int64_t n, uint32_t *head, uint32_t *tail, int lane)
{
	if (n <= 64) { if (lane == 0) rs_insertion(beg, beg + n); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); return; }
	for (int shift = 56; shift >= 0; shift -= 8) {
		if (shift == 56) { rs_level_wave(beg, n, shift, head, tail, lane); continue; }
		// runs of equal higher-order bytes, found 64 records at a time from a known run start
		int64_t rb = 0;
		while (rb < n) {
			const uint64_t hi0 = beg[rb].x >> (shift + 8);
			// a window of 64 records starting at rb: which of them start a new run?
			const int64_t pos = rb + lane;
			const bool in = pos < n;
			const uint64_t hk = in ? beg[pos].x >> (shift + 8) : ~0ULL;
			const uint32_t hp_lo = (uint32_t)__shfl((int)(uint32_t)(hk & 0xffffffffULL), lane > 0 ? lane - 1 : 0), hp_hi = (uint32_t)__shfl((int)(uint32_t)(hk >> 32), lane > 0 ? lane - 1 : 0);
			const uint64_t hp = lane == 0 ? hi0 : ((uint64_t)hp_hi << 32 | (uint64_t)hp_lo);
			const bool start = in && lane > 0 && hk != hp;
			unsigned long long sm = __ballot(start);
			const int n_in = (int)(n - rb < 64 ? n - rb : 64);
			if (sm == 0 && n_in == 64 && rb + 64 < n) {
				// the run beginning at rb covers the whole window: it is longer than 64 unless it ends exactly here
				int64_t re = rb + 64;
				for (;;) {   // extend to the run's end
					const int64_t p2 = re + lane;
					const bool brk = p2 >= n || (beg[p2].x >> (shift + 8)) != hi0;
					const unsigned long long bm = __ballot(brk);
					if (bm) { re += __ffsll((long long)bm) - 1; break; }
					re += 64;
				}
				if (re - rb > 64) rs_level_wave(beg + rb, re - rb, shift, head, tail, lane);
				else { if (lane == 0) rs_insertion(beg + rb, beg + re); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
				rb = re;
				continue;
			}
			// complete runs inside the window: run r spans [s_r, s_{r+1}); the last run of the window is complete only
			// if the window reaches the end of the array; otherwise it restarts the next window
			unsigned long long starts = sm | 1ULL;                                    // bit 0: the run at rb
			const int n_runs = __popcll(starts);
			const bool last_complete = (rb + n_in >= n);
			const int n_sort = last_complete ? n_runs : n_runs - 1;
			if (n_sort == 0) {
				// a single run that continues beyond the window (<= 64 so far): find its end
				int64_t re = rb + n_in;
				for (;;) {
					const int64_t p2 = re + lane;
					const bool brk = p2 >= n || (beg[p2].x >> (shift + 8)) != hi0;
					const unsigned long long bm = __ballot(brk);
					if (bm) { re += __ffsll((long long)bm) - 1; break; }
					re += 64;
				}
				if (re - rb > 64) rs_level_wave(beg + rb, re - rb, shift, head, tail, lane);
				else { if (lane == 0) rs_insertion(beg + rb, beg + re); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
				rb = re;
				continue;
			}
			// lane r sorts run r (all of them have <= 64 records)
			int64_t my_b = -1, my_e = -1, next_rb = rb + n_in;
			{
				// position of the r-th set bit of `starts`
				unsigned long long mrest = starts; int r = 0; int prev = -1;
				// every lane walks the (short) bit list; at most 64 iterations
				while (mrest) {
					const int bpos = __ffsll((long long)mrest) - 1;
					mrest &= mrest - 1;
					if (prev >= 0) { if (r - 1 == lane) my_b = rb + prev, my_e = rb + bpos; }
					prev = bpos; ++r;
				}
				if (last_complete) { if (n_runs - 1 == lane) my_b = rb + prev, my_e = rb + n_in; }
				else next_rb = rb + prev;
			}
			if (my_b >= 0 && my_e - my_b > 1) rs_insertion(beg + my_b, beg + my_e);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			rb = next_rb;
		}
	}
}

} // namespace pga
