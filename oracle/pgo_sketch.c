/* pgo_sketch.c -- ORACLE (test infrastructure only).
 *
 * (w,k) symmetric minimizers of a DNA string, restating mm_sketch() (sketch.c:77-143, non-HPC mode:
 * pangraph never sets -H, packages/pangraph/src/align/minimap2_lib/align_with_minimap2_lib.rs:42-57).
 *
 * The reference streams over the bases with a ring buffer; this restatement is written as the
 * position-parallel DEFINITION the HIP kernel implements:
 *
 *   slot      every base position that is not skipped.  A position is skipped iff it is ACGT and its
 *             forward and reverse k-mer words are equal (sketch.c:110); skipped positions neither
 *             enter the window nor advance the run counter.
 *   l[s]      number of ACGT slots since the last non-ACGT slot (inclusive); an N slot has l=0
 *             (sketch.c:111,116).
 *   info[s]   (hash64(min(fwd,rev))<<8 | k,  rid<<32 | pos<<1 | strand) when l[s] >= k, else "none"
 *             = (UINT64_MAX, UINT64_MAX) (sketch.c:112-115).
 *   cur[s]    the RIGHTMOST slot of minimal info.x among slots s-w+1..s (none counts as +inf and slots
 *             before the sequence start count as none): this is what `min` holds after slot s, by the
 *             `<=` at sketch.c:123 and the `>=` rescan at sketch.c:129-132.
 *
 * Emissions at slot s, with prev = cur[s-1] (sketch.c:118-139):
 *   A  if l[s] == w+k-1 and prev is real: every slot t in s-w+1..s-1 (ascending) with
 *      info[t].x == info[prev].x and info[t].y != info[prev].y.
 *   B  if info[s].x <= info[prev].x:  emit prev when l[s] >= w+k and prev is real.
 *   C  else if prev == s-w (it leaves the window): emit prev when l[s] >= w+k-1 and prev is real; then,
 *      when l[s] >= w+k-1 and cur[s] is real, every slot t in s-w+1..s (ascending) with
 *      info[t].x == info[cur[s]].x and info[t].y != info[cur[s]].y.
 *   D  after the last slot: emit cur[last] when real (sketch.c:141-142).
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "pgo.h"

const uint8_t pgo_nt4[256] = { /* sketch.c:9-26: A/a=0 C/c=1 G/g=2 T/t/U/u=3, all else 4 */
	0, 1, 2, 3,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 0, 4, 1,  4, 4, 4, 2,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  3, 3, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 0, 4, 1,  4, 4, 4, 2,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  3, 3, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,
	4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4,  4, 4, 4, 4
};

uint64_t pgo_hash64(uint64_t key, uint64_t mask) /* sketch.c:28-38: invertible integer mix */
{
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

static size_t push(pg128 **out, size_t n, size_t *cap, pg128 v)
{
	if (n == *cap) {
		*cap = *cap ? *cap * 2 : 256;
		*out = (pg128*)realloc(*out, *cap * sizeof(pg128));
	}
	(*out)[n] = v;
	return n + 1;
}

size_t pgo_sketch(const char *seq, int len, int w, int k, uint32_t rid, pg128 **out, size_t n, size_t *cap)
{
	const pg128 none = { UINT64_MAX, UINT64_MAX };
	uint64_t mask = (1ULL << 2 * k) - 1, shift1 = 2 * (k - 1), fw = 0, rv = 0;
	pg128 *info;
	int *l, n_slot = 0, run = 0;
	assert(len > 0 && w > 0 && w < 256 && k > 0 && k <= 28); /* sketch.c:84 */

	/* pass 1: slots, l[] and info[] */
	info = (pg128*)malloc((size_t)len * sizeof(pg128));
	l = (int*)malloc((size_t)len * sizeof(int));
	for (int i = 0; i < len; ++i) {
		int c = pgo_nt4[(uint8_t)seq[i]];
		pg128 v = none;
		if (c < 4) {
			fw = (fw << 2 | (uint64_t)c) & mask;                  /* sketch.c:108 */
			rv = (rv >> 2) | (3ULL ^ (uint64_t)c) << shift1;      /* sketch.c:109 */
			if (fw == rv) continue;                               /* sketch.c:110: not a slot */
			int z = fw < rv ? 0 : 1;
			++run;
			if (run >= k) {                                       /* span == k < 256 in non-HPC mode */
				v.x = pgo_hash64(z ? rv : fw, mask) << 8 | (uint64_t)k;
				v.y = (uint64_t)rid << 32 | (uint64_t)(uint32_t)i << 1 | (uint64_t)z;
			}
		} else run = 0;
		info[n_slot] = v, l[n_slot] = run;
		++n_slot;
	}

	/* pass 2: per-slot emission rules */
	int prev = -1; /* cur[s-1]; -1 stands for the all-"none" initial window */
	for (int s = 0; s < n_slot; ++s) {
		int lo = s - w + 1 > 0 ? s - w + 1 : 0;
		pg128 pv = prev >= 0 ? info[prev] : none;
		int prev_real = pv.x != UINT64_MAX;
		/* cur[s]: rightmost minimum over lo..s; virtual "none" slots before 0 lose to any real slot and,
		   being older, also lose ties among nones */
		int cur = lo;
		for (int t = lo + 1; t <= s; ++t)
			if (info[t].x <= info[cur].x) cur = t;
		if (l[s] == w + k - 1 && prev_real) /* rule A */
			for (int t = lo; t < s; ++t)
				if (info[t].x == pv.x && info[t].y != pv.y) n = push(out, n, cap, info[t]);
		if (info[s].x <= pv.x) { /* rule B */
			if (l[s] >= w + k && prev_real) n = push(out, n, cap, pv);
		} else if (prev == s - w) { /* rule C (prev >= 0 here because pv is real: info[s].x > pv.x) */
			if (l[s] >= w + k - 1 && prev_real) n = push(out, n, cap, pv);
			if (l[s] >= w + k - 1 && info[cur].x != UINT64_MAX)
				for (int t = lo; t <= s; ++t)
					if (info[t].x == info[cur].x && info[t].y != info[cur].y) n = push(out, n, cap, info[t]);
		}
		prev = cur;
	}
	if (prev >= 0 && info[prev].x != UINT64_MAX) n = push(out, n, cap, info[prev]); /* rule D */
	free(info); free(l);
	return n;
}
