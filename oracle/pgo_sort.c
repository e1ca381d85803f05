/* pgo_sort.c -- ORACLE (test infrastructure only).
 *
 * minimap2 sorts with an UNSTABLE in-place MSD radix sort (ksort.h:101-151, instantiated at
 * misc.c:155-159).  Where keys tie, the order it leaves behind is observable downstream (anchor
 * indices in chaining, chain priority in backtracking), so the restatement has to
 * reproduce the same permutation, not just a sorted order.  The procedure is an "American flag" sort:
 * 8-bit digits from the most significant byte, a cycle-leader permutation per level, recursion into
 * buckets of more than 64 elements and a (stable) insertion sort for smaller ones.
 */
#include <string.h>
#include "pgo.h"

#define PGO_RS_SMALL 64

#define PGO_DEFINE_RADIX(NAME, T, KEY)                                                            \
static void NAME##_small(T *beg, T *end) /* ksort.h:107-117: stable insertion sort */             \
{                                                                                                 \
	for (T *i = beg + 1; i < end; ++i) {                                                          \
		if (KEY(*i) < KEY(*(i - 1))) {                                                            \
			T tmp = *i, *j = i;                                                                   \
			while (j > beg && KEY(tmp) < KEY(*(j - 1))) { *j = *(j - 1); --j; }                   \
			*j = tmp;                                                                             \
		}                                                                                         \
	}                                                                                             \
}                                                                                                 \
static void NAME##_level(T *beg, T *end, int shift) /* ksort.h:118-146 */                         \
{                                                                                                 \
	T *head[256], *tail[256];                                                                     \
	size_t cnt[256];                                                                              \
	memset(cnt, 0, sizeof(cnt));                                                                  \
	for (T *i = beg; i != end; ++i) ++cnt[(KEY(*i) >> shift) & 255];                              \
	T *pos = beg;                                                                                 \
	for (int d = 0; d < 256; ++d) { head[d] = pos; pos += cnt[d]; tail[d] = pos; }                \
	/* cycle-leader permutation: walk buckets in digit order; an element that is not home starts  \
	   a displacement chain that ends when an element belonging to the current bucket turns up */ \
	for (int d = 0; d < 256;) {                                                                   \
		if (head[d] == tail[d]) { ++d; continue; }                                                \
		int dst = (int)((KEY(*head[d]) >> shift) & 255);                                          \
		if (dst == d) { ++head[d]; continue; }                                                    \
		T carry = *head[d];                                                                       \
		do {                                                                                      \
			T put = carry;                                                                        \
			carry = *head[dst];                                                                   \
			*head[dst]++ = put;                                                                   \
			dst = (int)((KEY(carry) >> shift) & 255);                                             \
		} while (dst != d);                                                                       \
		*head[d]++ = carry;                                                                       \
	}                                                                                             \
	if (shift == 0) return;                                                                       \
	int next = shift > 8 ? shift - 8 : 0;                                                         \
	T *b = beg;                                                                                   \
	for (int d = 0; d < 256; ++d) {                                                               \
		T *e = tail[d];                                                                           \
		if (e - b > PGO_RS_SMALL) NAME##_level(b, e, next);                                       \
		else if (e - b > 1) NAME##_small(b, e);                                                   \
		b = e;                                                                                    \
	}                                                                                             \
}                                                                                                 \
void NAME(T *beg, T *end) /* ksort.h:147-151 */                                                   \
{                                                                                                 \
	if (end - beg <= PGO_RS_SMALL) NAME##_small(beg, end);                                        \
	else NAME##_level(beg, end, 56);                                                              \
}

#define PGO_KEY128(a) ((a).x)
#define PGO_KEY64(a) (a)
PGO_DEFINE_RADIX(pgo_radix_sort_128x, pg128, PGO_KEY128)
PGO_DEFINE_RADIX(pgo_radix_sort_64, uint64_t, PGO_KEY64)
