/* pgo_seed.c -- ORACLE (test infrastructure only).
 *
 * Query-side minimizer filtering, index probing, high-occurrence rescue and anchor expansion,
 * restating mm_seed_mz_flt / mm_seed_collect_all / mm_seed_select / mm_collect_matches (seed.c:5-131)
 * and skip_seed / collect_seed_hits (map.c:78-100,168-204).
 */
#include <stdlib.h>
#include <string.h>
#include "pgo.h"

#define SEED_TANDEM (1ULL<<42)  /* mmpriv.h:20 */
#define SEED_SELF   (1ULL<<43)  /* mmpriv.h:21 */

static int cmp_u64(const void *a, const void *b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }

/* seed.c:5-28.  Drops every query minimizer whose hash occurs more than q_occ_max times in the query
 * and more than n*q_occ_frac times.  Only the counts matter, so any sort will do. */
size_t pgo_seed_mz_flt(pg128 *mv, size_t n, int32_t q_occ_max, float q_occ_frac)
{
	if (n <= (size_t)q_occ_max || q_occ_frac <= 0.0f || q_occ_max <= 0) return n;
	uint64_t *h = (uint64_t*)malloc(n * 8);
	for (size_t i = 0; i < n; ++i) h[i] = mv[i].x;
	qsort(h, n, 8, cmp_u64);
	size_t j = 0;
	for (size_t i = 0; i < n; ++i) {
		/* count of mv[i].x in the sorted copy */
		size_t lo = 0, hi = n;
		while (lo < hi) { size_t m = (lo + hi) >> 1; if (h[m] < mv[i].x) lo = m + 1; else hi = m; }
		size_t first = lo; hi = n;
		while (lo < hi) { size_t m = (lo + hi) >> 1; if (h[m] <= mv[i].x) lo = m + 1; else hi = m; }
		int32_t cnt = (int32_t)(lo - first);
		if (cnt > q_occ_max && cnt > n * q_occ_frac) continue; /* size_t*float -> float compare, as seed.c:17 */
		mv[j++] = mv[i];
	}
	free(h);
	return j;
}

/* seed.c:56-96: inside every maximal streak of seeds with n > max_occ keep the max_high_occ
 * lowest-occurrence ones (ties: earliest first), and always drop n > max_max_occ. */
static void seed_select(int32_t n, pgo_seed_t *a, int len, int max_occ, int max_max_occ, int dist)
{
	if (n == 0 || n == 1) return;
	int32_t m = 0;
	for (int32_t i = 0; i < n; ++i) if (a[i].n > (uint32_t)max_occ) ++m;
	if (m == 0) return;
	int32_t last0 = -1;
	for (int32_t i = 0; i <= n; ++i) {
		if (i != n && a[i].n > (uint32_t)max_occ) continue;
		if (i - last0 > 1) {
			int32_t ps = last0 < 0 ? 0 : (int32_t)(a[last0].q_pos >> 1);
			int32_t pe = i == n ? len : (int32_t)(a[i].q_pos >> 1);
			int32_t st = last0 + 1, en = i;
			int32_t keep = (int32_t)((double)(pe - ps) / dist + .499);
			if (keep > 0) {
				if (keep > 128) keep = 128; /* MAX_MAX_HIGH_OCC */
				/* the reference runs a bounded max-heap on (n<<32|index) replacing the top only when a
				   strictly smaller n arrives (seed.c:78-87): the survivors are the `keep` smallest under
				   (n asc, index asc) */
				int32_t cnt = en - st;
				uint64_t *key = (uint64_t*)malloc((size_t)cnt * 8);
				for (int32_t j = 0; j < cnt; ++j) key[j] = (uint64_t)a[st + j].n << 32 | (uint32_t)(st + j);
				qsort(key, (size_t)cnt, 8, cmp_u64);
				for (int32_t j = 0; j < cnt && j < keep; ++j) a[(uint32_t)key[j]].flt = 1;
				free(key);
			}
			for (int32_t j = st; j < en; ++j) a[j].flt ^= 1;
			for (int32_t j = st; j < en; ++j)
				if (a[j].n > (uint32_t)max_max_occ) a[j].flt = 1;
		}
		last0 = i;
	}
}

/* map.c:78-100 with qname != NULL */
static int skip_seed(int64_t flag, uint64_t r, const pgo_seed_t *q, const char *qname, int qlen, const pgo_index_t *ix, int *is_self)
{
	*is_self = 0;
	if (qname && (flag & (MM_F_NO_DIAG | MM_F_NO_DUAL))) {
		const mm_idx_seq_t *s = &ix->hdr.seq[r >> 32];
		int cmp = strcmp(qname, s->name);
		if ((flag & MM_F_NO_DIAG) && cmp == 0 && (int)s->len == qlen) {
			if ((uint32_t)r >> 1 == (q->q_pos >> 1)) return 1;
			if ((r & 1) == (q->q_pos & 1)) *is_self = 1;
		}
		if ((flag & MM_F_NO_DUAL) && cmp > 0) return 1;
	}
	if (flag & (MM_F_FOR_ONLY | MM_F_REV_ONLY)) {
		if ((r & 1) == (q->q_pos & 1)) { if (flag & MM_F_REV_ONLY) return 1; }
		else if (flag & MM_F_FOR_ONLY) return 1;
	}
	return 0;
}

pg128 *pgo_collect_anchors(const pgo_index_t *ix, const mm_mapopt_t *opt, const char *qname, int qlen,
                           const pg128 *mv, size_t n_mv, int64_t *n_a_, int *rep_len_)
{
	/* seed.c:30-54: one seed per query minimizer that has hits */
	pgo_seed_t *m = (pgo_seed_t*)malloc((n_mv ? n_mv : 1) * sizeof(pgo_seed_t));
	int32_t n_m0 = 0;
	for (size_t i = 0; i < n_mv; ++i) {
		int t;
		const uint64_t *cr = pgo_index_get(ix, mv[i].x >> 8, &t);
		if (t == 0) continue;
		pgo_seed_t *q = &m[n_m0++];
		q->q_pos = (uint32_t)mv[i].y, q->q_span = mv[i].x & 0xff, q->cr = cr, q->n = (uint32_t)t;
		q->flt = 0, q->is_tandem = 0;
		if (i > 0 && mv[i].x >> 8 == mv[i - 1].x >> 8) q->is_tandem = 1;
		if (i < n_mv - 1 && mv[i].x >> 8 == mv[i + 1].x >> 8) q->is_tandem = 1;
	}
	/* seed.c:98-131 */
	int max_occ = opt->mid_occ;
	if (opt->occ_dist > 0 && opt->max_max_occ > max_occ) seed_select(n_m0, m, qlen, max_occ, opt->max_max_occ, opt->occ_dist);
	else for (int32_t i = 0; i < n_m0; ++i) if (m[i].n > (uint32_t)max_occ) m[i].flt = 1;
	int rep_st = 0, rep_en = 0, rep_len = 0;
	int64_t n_a = 0;
	int32_t n_m = 0;
	for (int32_t i = 0; i < n_m0; ++i) {
		pgo_seed_t *q = &m[i];
		if (q->flt) {
			int en = (int)(q->q_pos >> 1) + 1, st = en - (int)q->q_span;
			if (st > rep_en) { rep_len += rep_en - rep_st; rep_st = st, rep_en = en; }
			else rep_en = en;
		} else { n_a += q->n; m[n_m++] = *q; }
	}
	rep_len += rep_en - rep_st;
	/* map.c:168-204: expand */
	pg128 *a = (pg128*)malloc((size_t)(n_a ? n_a : 1) * sizeof(pg128));
	n_a = 0;
	for (int32_t i = 0; i < n_m; ++i) {
		const pgo_seed_t *q = &m[i];
		for (uint32_t k = 0; k < q->n; ++k) {
			uint64_t r = q->cr[k];
			int is_self;
			int32_t rpos = (int32_t)((uint32_t)r >> 1);
			if (skip_seed(opt->flag, r, q, qname, qlen, ix, &is_self)) continue;
			pg128 *p = &a[n_a++];
			if ((r & 1) == (q->q_pos & 1)) { /* same strand */
				p->x = (r & 0xffffffff00000000ULL) | (uint64_t)rpos;
				p->y = (uint64_t)q->q_span << 32 | q->q_pos >> 1;
			} else { /* opposite strand: query coordinate on the reverse complement (map.c:188-190) */
				p->x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint64_t)rpos;
				p->y = (uint64_t)q->q_span << 32 | (uint32_t)(qlen - ((int32_t)(q->q_pos >> 1) + 1 - (int32_t)q->q_span) - 1);
			}
			if (q->is_tandem) p->y |= SEED_TANDEM;
			if (is_self) p->y |= SEED_SELF;
		}
	}
	free(m);
	pgo_radix_sort_128x(a, a + n_a); /* map.c:202 -- tie order among equal x is part of the contract */
	*n_a_ = n_a, *rep_len_ = rep_len;
	return a;
}
