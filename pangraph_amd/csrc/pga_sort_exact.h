// pga_sort_exact.h -- device-side re-enactment of minimap2's UNSTABLE in-place MSD radix sort.
//
// Reference: packages/minimap2-sys/minimap2/ksort.h:101-151 instantiated as radix_sort_128x (misc.c:155-156):
// 8-bit digits from the top byte, a cycle-leader ("American flag") permutation per level, recursion into
// buckets of more than 64 records, stable insertion sort for smaller ones.  Where keys tie, the order it
// leaves is observable downstream, and that order is defined by the sequential walk of the
// permutation -- there is no closed form -- so a lane replays it.  Two exact shortcuts keep the replay short:
//   * levels are visited breadth-first (all runs of equal higher-order bytes at shift s, then s-8): runs are
//     independent, and re-sorting an already insertion-sorted run of <= 64 records is a no-op;
//   * a run whose records all share the digit at shift s is left untouched by that level (every record is
//     already "home", ksort.h:132), so the walk is skipped.
// Callers first try a parallel stable sort and use this only for arrays that actually contain equal keys.
#pragma once
#include "pga_common.h"

namespace pga {

__host__ __device__ inline void rs_insertion(u128 *beg, u128 *end) // ksort.h:107-117
{
	for (u128 *i = beg + 1; i < end; ++i) {
		if (i->x < (i - 1)->x) {
			u128 tmp = *i, *j = i;
			while (j > beg && tmp.x < (j - 1)->x) { *j = *(j - 1); --j; }
			*j = tmp;
		}
	}
}

// one level of ksort.h:118-146 on [beg,end) at `shift`; head/tail are 256-entry scratch arrays
__host__ __device__ inline void rs_level(u128 *beg, u128 *end, int shift, uint32_t *head, uint32_t *tail)
{
	for (int d = 0; d < 256; ++d) head[d] = 0;
	uint64_t first = (beg->x >> shift) & 255; bool same = true;
	for (u128 *i = beg; i != end; ++i) { uint64_t d = (i->x >> shift) & 255; ++head[d]; same &= (d == first); }
	if (same) return;                                   // identity permutation
	uint32_t pos = 0;
	for (int d = 0; d < 256; ++d) { uint32_t c = head[d]; head[d] = pos; pos += c; tail[d] = pos; }
	for (int d = 0; d < 256;) {
		if (head[d] == tail[d]) { ++d; continue; }
		int dst = (int)((beg[head[d]].x >> shift) & 255);
		if (dst == d) { ++head[d]; continue; }
		u128 carry = beg[head[d]];
		do {
			u128 put = carry;
			carry = beg[head[dst]];
			beg[head[dst]++] = put;
			dst = (int)((carry.x >> shift) & 255);
		} while (dst != d);
		beg[head[d]++] = carry;
	}
}

__host__ __device__ inline void radix_sort_128x_exact(u128 *beg, u128 *end, uint32_t *head, uint32_t *tail)
{
	const int64_t n = end - beg;
	if (n <= 64) { rs_insertion(beg, end); return; }    // ksort.h:149
	for (int shift = 56; shift >= 0; shift -= 8) {
		// runs = maximal stretches with equal bits above shift+8 (the whole array at the top level)
		u128 *rb = beg;
		while (rb < end) {
			u128 *re = rb + 1;
			if (shift == 56) re = end;
			else { const uint64_t hi = rb->x >> (shift + 8); while (re < end && (re->x >> (shift + 8)) == hi) ++re; }
			const int64_t m = re - rb;
			if (m > 64) rs_level(rb, re, shift, head, tail);
			else if (m > 1) rs_insertion(rb, re);
			rb = re;
		}
	}
}

} // namespace pga
