/* pgo_index.c -- ORACLE (test infrastructure only).
 *
 * Minimizer index over a set of sequences, restating mm_idx_str() (index.c:408-456) and what is
 * observable of it: mm_idx_get() returns, for a minimizer hash, the list of occurrences
 * y = rid<<32 | lastPos<<1 | strand in ASCENDING order (index.c:238-258 sorts multi-occurrence lists
 * with radix_sort_64; singletons are a list of one), and mm_idx_cal_max_occ() (index.c:186-207) is an
 * order statistic of the per-minimizer occurrence counts.  The 2^b khash buckets of the reference are
 * an implementation detail; here the index is one sorted key array + CSR occurrence lists, which is
 * also the layout the HIP backend uses.
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "pgo.h"

static int cmp128(const void *a, const void *b)
{
	const pg128 *p = (const pg128*)a, *q = (const pg128*)b;
	if (p->x != q->x) return p->x < q->x ? -1 : 1;
	return p->y < q->y ? -1 : p->y > q->y;
}

pgo_index_t *pgo_index_build(int w, int k, int bucket_bits, int n, const char **seq, const char **name)
{
	if (n <= 0) return 0; /* index.c:416 */
	pgo_index_t *ix = (pgo_index_t*)calloc(1, sizeof(pgo_index_t));
	uint64_t sum = 0;
	if (bucket_bits < 0) bucket_bits = 14;
	if (k * 2 < bucket_bits) bucket_bits = k * 2; /* index.c:47 */
	if (w < 1) w = 1;                              /* index.c:48 */
	ix->hdr.w = w, ix->hdr.k = k, ix->hdr.b = bucket_bits, ix->hdr.flag = name ? 0 : MM_I_NO_NAME;
	ix->hdr.n_seq = (uint32_t)n;
	ix->hdr.seq = (mm_idx_seq_t*)calloc((size_t)n, sizeof(mm_idx_seq_t));
	for (int i = 0; i < n; ++i) sum += strlen(seq[i]);
	ix->nt4 = (uint8_t*)malloc(sum ? sum : 1);
	pg128 *mz = 0; size_t n_mz = 0, cap = 0;
	sum = 0;
	for (int i = 0; i < n; ++i) {
		mm_idx_seq_t *s = &ix->hdr.seq[i];
		if (name && name[i]) {
			s->name = strdup(name[i]);
			for (int j = 0; j < i; ++j) /* index.c:436 asserts names are unique */
				assert(ix->hdr.seq[j].name == 0 || strcmp(ix->hdr.seq[j].name, s->name) != 0);
		}
		s->offset = sum, s->len = (uint32_t)strlen(seq[i]);
		for (uint32_t j = 0; j < s->len; ++j) ix->nt4[sum + j] = pgo_nt4[(uint8_t)seq[i][j]];
		sum += s->len;
		if (s->len > 0) n_mz = pgo_sketch(seq[i], (int)s->len, w, k, (uint32_t)i, &mz, n_mz, &cap);
	}
	/* group by hash (x>>8), occurrences ascending by y */
	qsort(mz, n_mz, sizeof(pg128), cmp128);
	ix->key = (uint64_t*)malloc((n_mz + 1) * 8);
	ix->occ_off = (uint64_t*)malloc((n_mz + 2) * 8);
	ix->occ = (uint64_t*)malloc((n_mz + 1) * 8);
	for (size_t i = 0; i < n_mz; ++i) {
		if (i == 0 || mz[i].x >> 8 != mz[i - 1].x >> 8) {
			ix->key[ix->n_keys] = mz[i].x >> 8;
			ix->occ_off[ix->n_keys++] = i;
		}
		ix->occ[i] = mz[i].y;
	}
	ix->occ_off[ix->n_keys] = n_mz;
	free(mz);
	return ix;
}

void pgo_index_free(pgo_index_t *ix)
{
	if (!ix) return;
	for (uint32_t i = 0; i < ix->hdr.n_seq; ++i) free(ix->hdr.seq[i].name);
	free(ix->hdr.seq); free(ix->nt4); free(ix->key); free(ix->occ_off); free(ix->occ); free(ix);
}

const uint64_t *pgo_index_get(const pgo_index_t *ix, uint64_t minier, int *n) /* index.c:84-98 */
{
	uint64_t lo = 0, hi = ix->n_keys;
	*n = 0;
	while (lo < hi) {
		uint64_t mid = (lo + hi) >> 1;
		if (ix->key[mid] < minier) lo = mid + 1; else hi = mid;
	}
	if (lo == ix->n_keys || ix->key[lo] != minier) return 0;
	*n = (int)(ix->occ_off[lo + 1] - ix->occ_off[lo]);
	return ix->occ + ix->occ_off[lo];
}

static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : x > y; }

int32_t pgo_index_cal_max_occ(const pgo_index_t *ix, float f) /* index.c:186-207 */
{
	if (f <= 0.) return INT32_MAX;
	size_t n = ix->n_keys;
	if (n == 0) return 1; /* the reference would read a[0] of an empty array; pangraph never indexes zero minimizers without also having zero queries */
	uint32_t *c = (uint32_t*)malloc(n * 4);
	for (size_t i = 0; i < n; ++i) c[i] = (uint32_t)(ix->occ_off[i + 1] - ix->occ_off[i]);
	qsort(c, n, 4, cmp_u32);
	size_t kk = (uint32_t)((1. - f) * n); /* double arithmetic on a float f, truncated (index.c:204) */
	uint32_t thres = c[kk] + 1;
	free(c);
	return (int32_t)thres;
}
