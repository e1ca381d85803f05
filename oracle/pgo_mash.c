/* pgo_mash.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of pangraph's guide-tree path.
 *
 *   hash              packages/pangraph/src/distance/mash/hash.rs:3-12
 *   minimizer sketch  packages/pangraph/src/distance/mash/minimizer.rs:49-160   (k = 15, w = 100 by default, :12-16)
 *   mash distance     packages/pangraph/src/distance/mash/mash_distance.rs:9-65
 *   neighbor joining  packages/pangraph/src/tree/neighbor_joining.rs:16-103
 *
 * The reference is Rust and cannot be built in this image.  What pins this file: the known-answer vectors of the reference's own
 * unit tests (hash.rs:19-27, minimizer.rs:188-210, mash_distance.rs:84-123 and :133-152, neighbor_joining.rs:113-151), checked in
 * tests/test_mash_cpu.py.  Those vectors fix the hash, the sketch and the distance bit for bit.  The f64 arithmetic of the
 * neighbor-joining step is restated from the published algorithm of the reference's dependency ndarray 0.16.1 (Cargo.lock:1391;
 * not vendored): `sum_axis` over the contiguous axis is an eight-accumulator unrolled fold per lane, over the other axis a row by
 * row accumulation; `argmin` (ndarray-stats 0.6.0) keeps the first minimum in row-major order.  The reference's tests of that step
 * use small integers, which no summation order can tell apart: the ORDER of the f64 sums is "parity unpinned".
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* hash.rs:3-12 (Jenkins' invertible 64-bit hash, the same function as minimap2's hash64) */
uint64_t pgo_mash_hash(uint64_t x, uint64_t mask)
{
	x = (~x + (x << 21)) & mask;
	x = x ^ (x >> 24);
	x = (x + (x << 3) + (x << 8)) & mask;
	x = x ^ (x >> 14);
	x = (x + (x << 2) + (x << 4)) & mask;
	x = x ^ (x >> 28);
	x = (x + (x << 31)) & mask;
	return x;
}

typedef struct { uint64_t value, position; } pgo_mz_t;

/* minimizer.rs:170-187: A/a 0, C/c 1, G/g 2, T/t/U/u 3, everything else 4 */
static int mash_code(unsigned char c)
{
	switch (c) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': case 'U': case 'u': return 3;
	default: return 4;
	}
}

#define MZ_MAX UINT64_MAX

static void mz_push(pgo_mz_t **out, size_t *n, size_t *cap, pgo_mz_t m)
{
	if (*n == *cap) { *cap = *cap ? *cap * 2 : 256; *out = (pgo_mz_t*)realloc(*out, *cap * sizeof(pgo_mz_t)); }
	(*out)[(*n)++] = m;
}

/* minimizer.rs:49-160.  Returns the number of minimizers appended to *out (0: the reference reports an error, :156-158). */
size_t pgo_mash_sketch(const char *seq, size_t len, uint64_t id, int k, int w, pgo_mz_t **out, size_t *n_out, size_t *cap)
{
	const size_t n0 = *n_out;
	uint64_t fwd = 0, rev = 0;
	const uint64_t mask = (1ULL << (2 * k)) - 1, shift = 2 * (uint64_t)(k - 1);
	pgo_mz_t min = { MZ_MAX, MZ_MAX };
	pgo_mz_t *window = (pgo_mz_t*)malloc((size_t)w * sizeof(pgo_mz_t));
	for (int i = 0; i < w; ++i) window[i].value = window[i].position = MZ_MAX;
	uint64_t l = 0;
	size_t bi = 0, mi = 0;
	for (size_t p = 0; p < len; ++p) {
		const uint64_t locus = (uint64_t)p + 1;                                   /* :70 */
		const int c = mash_code((unsigned char)seq[p]);
		pgo_mz_t nw = { MZ_MAX, MZ_MAX };
		if (c >= 4) l = 0;                                                        /* :73-75 */
		else {
			fwd = ((fwd << 2) | (uint64_t)c) & mask;                              /* :77 */
			rev = (rev >> 2) | ((uint64_t)(3 ^ c) << shift);                      /* :78 */
			++l;
			if (l >= (uint64_t)k) {                                               /* :80-87: the forward strand wins ties */
				const uint64_t pos = (id << 32) | (locus << 1);
				if (fwd <= rev) { nw.value = pgo_mash_hash(fwd, mask); nw.position = pos; }
				else { nw.value = pgo_mash_hash(rev, mask); nw.position = pos | 1; }
			}
		}
		window[bi] = nw;                                                          /* :93 */
		if (l == (uint64_t)(w + k - 1) && min.value != MZ_MAX) {                  /* :94-105 */
			for (size_t i = bi + 1; i < (size_t)w; ++i) if (min.value == window[i].value && min.position != window[i].position) mz_push(out, n_out, cap, window[i]);
			for (size_t i = 0; i <= bi; ++i) if (min.value == window[i].value && min.position != window[i].position) mz_push(out, n_out, cap, window[i]);
		}
		if (nw.value < min.value) {                                               /* :107-112 */
			if (l >= (uint64_t)(w + k) && min.value != MZ_MAX) mz_push(out, n_out, cap, min);
			min = nw; mi = bi;
		} else if (bi == mi) {                                                    /* :113-146 */
			if (l >= (uint64_t)(w + k - 1) && min.value != MZ_MAX) mz_push(out, n_out, cap, min);
			min.value = MZ_MAX;                                                   /* (keeps its position, :118) */
			for (size_t i = bi + 1; i < (size_t)w; ++i) if (window[i].value < min.value) { mi = i; min = window[i]; }
			for (size_t i = 0; i <= bi; ++i) if (window[i].value < min.value) { mi = i; min = window[i]; }
			if (l >= (uint64_t)(w + k - 1) && min.value != MZ_MAX) {
				for (size_t i = bi + 1; i < (size_t)w; ++i) if (min.value == window[i].value && min.position != window[i].position) mz_push(out, n_out, cap, window[i]);
				for (size_t i = 0; i <= bi; ++i) if (min.value == window[i].value && min.position != window[i].position) mz_push(out, n_out, cap, window[i]);
			}
		}
		if (++bi >= (size_t)w) bi = 0;                                            /* :148-151 */
	}
	if (min.value != MZ_MAX) mz_push(out, n_out, cap, min);                       /* :154-156 */
	free(window);
	return *n_out - n0;
}

static int cmp_mz_value(const void *a, const void *b)
{
	const pgo_mz_t *x = (const pgo_mz_t*)a, *y = (const pgo_mz_t*)b;
	return x->value < y->value ? -1 : x->value > y->value;
}

/* mash_distance.rs:9-65 over n sequences; dist is n x n, row major.  Returns 0, or -1 - i if sequence i has no minimizer
 * (the reference panics there, :19-20 / :51-54). */
int pgo_mash_distance(int n, const char *const *seqs, const size_t *lens, int k, int w, double *dist)
{
	if (n <= 0) return 0;
	pgo_mz_t *mz = NULL; size_t n_mz = 0, cap = 0;
	for (int i = 0; i < n; ++i)
		if (pgo_mash_sketch(seqs[i], lens[i], (uint64_t)i, k, w, &mz, &n_mz, &cap) == 0) { free(mz); return -1 - i; }
	qsort(mz, n_mz, sizeof(pgo_mz_t), cmp_mz_value);                              /* (:22; only the grouping by value matters below) */
	for (size_t i = 0; i < (size_t)n * (size_t)n; ++i) dist[i] = 0.0;
	uint8_t *seen = (uint8_t*)calloc((size_t)n, 1);
	int *hits = (int*)malloc((size_t)n * sizeof(int));
	for (size_t l = 0; l < n_mz;) {                                               /* :31-49 */
		size_t r = l;
		int nh = 0;
		while (r < n_mz && mz[r].value == mz[l].value) {
			const int s = (int)(mz[r].position >> 32);
			if (!seen[s]) { seen[s] = 1; hits[nh++] = s; }
			++r;
		}
		/* unique + sorted ids (:37-42); every pair i <= j of them counts one shared minimizer (:44-48) */
		for (int a = 0; a < nh; ++a) seen[hits[a]] = 0;
		for (int a = 1; a < nh; ++a) { const int v = hits[a]; int b = a - 1; while (b >= 0 && hits[b] > v) { hits[b + 1] = hits[b]; --b; } hits[b + 1] = v; }
		for (int a = 0; a < nh; ++a) for (int b = a; b < nh; ++b) dist[(size_t)hits[a] * n + hits[b]] += 1.0;
		l = r;
	}
	free(seen); free(hits); free(mz);
	for (int i = 0; i < n; ++i) {                                                 /* :51-62 */
		if (!(dist[(size_t)i * n + i] > 0.)) return -1 - i;
		for (int j = i + 1; j < n; ++j) {
			dist[(size_t)i * n + j] = 1.0 - dist[(size_t)i * n + j] / dist[(size_t)i * n + i];
			dist[(size_t)j * n + i] = dist[(size_t)i * n + j];
		}
		dist[(size_t)i * n + i] = 0.0;
	}
	return 0;
}

/* ndarray 0.16.1 numeric_util::unrolled_fold with f = +, init = 0: eight accumulators over blocks of eight, combined as
 * (p0+p4) + (p1+p5) + (p2+p6) + (p3+p7) into acc (in that order), then the remaining < 8 elements one by one. */
static double unrolled_sum(const double *xs, size_t n)
{
	double acc = 0.0, p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	while (n >= 8) { for (int i = 0; i < 8; ++i) p[i] = p[i] + xs[i]; xs += 8; n -= 8; }
	acc = acc + (p[0] + p[4]);
	acc = acc + (p[1] + p[5]);
	acc = acc + (p[2] + p[6]);
	acc = acc + (p[3] + p[7]);
	for (size_t i = 0; i < n; ++i) acc = acc + xs[i];
	return acc;
}

/* neighbor_joining.rs:47-63 on a dense m x m matrix (row major, leading dimension m): Q, diagonal +inf */
void pgo_nj_q_matrix(int m, const double *D, double *Q)
{
	double *sum0 = (double*)calloc((size_t)m, sizeof(double)), *sum1 = (double*)malloc((size_t)m * sizeof(double));
	for (int r = 0; r < m; ++r) for (int c = 0; c < m; ++c) sum0[c] = sum0[c] + D[(size_t)r * m + c];   /* sum_axis(Axis(0)): row by row */
	for (int r = 0; r < m; ++r) sum1[r] = unrolled_sum(D + (size_t)r * m, (size_t)m);                    /* sum_axis(Axis(1)): lane.sum() */
	for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j)
		Q[(size_t)i * m + j] = i == j ? INFINITY : (((double)m - 2.0) * D[(size_t)i * m + j] - sum0[j]) - sum1[i];   /* :60-61 */
	free(sum0); free(sum1);
}

/* neighbor_joining.rs:74-81 */
void pgo_nj_dist(int m, const double *D, int i, int j, double *dn)
{
	for (int c = 0; c < m; ++c) dn[c] = 0.5 * ((D[(size_t)i * m + c] + D[(size_t)j * m + c]) - D[(size_t)i * m + j]);
}

/* neighbor_joining.rs:16-35,83-103: leaves are nodes 0..n-1; join t (t = 0..n-2) creates node n + t with children
 * merges[2t], merges[2t+1] (the node lists' order: nodes[i] before nodes[j], i < j; the last join is the root (:27)). */
int pgo_nj_tree(int n, const double *dist, int32_t *merges)
{
	if (n < 2) return 0;
	int m = n, t = 0;
	double *D = (double*)malloc((size_t)n * n * sizeof(double)), *Q = (double*)malloc((size_t)n * n * sizeof(double)), *dn = (double*)malloc((size_t)n * sizeof(double));
	int32_t *node = (int32_t*)malloc((size_t)n * sizeof(int32_t));
	memcpy(D, dist, (size_t)n * n * sizeof(double));
	for (int i = 0; i < n; ++i) node[i] = i;
	while (m > 2) {
		pgo_nj_q_matrix(m, D, Q);
		int bi = 0, bj = 0; double best = Q[0];                                   /* argmin: first minimum in row-major order */
		for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) if (Q[(size_t)i * m + j] < best) { best = Q[(size_t)i * m + j]; bi = i; bj = j; }
		int i = bi < bj ? bi : bj, j = bi < bj ? bj : bi;                         /* :65-72 */
		if (i == j) { free(D); free(Q); free(dn); free(node); return -1; }       /* (all +inf / NaN: the reference errors) */
		merges[2 * t] = node[i]; merges[2 * t + 1] = node[j];
		node[i] = n + t; ++t;
		pgo_nj_dist(m, D, i, j, dn);                                              /* :93 */
		for (int c = 0; c < m; ++c) { D[(size_t)i * m + c] = dn[c]; D[(size_t)c * m + i] = dn[c]; }
		D[(size_t)i * m + i] = 0.0;
		/* remove row and column j (:98-99), compacting to (m-1) x (m-1) */
		for (int r = 0, rr = 0; r < m; ++r) {
			if (r == j) continue;
			for (int c = 0, cc = 0; c < m; ++c) { if (c == j) continue; Q[(size_t)rr * (m - 1) + cc] = D[(size_t)r * m + c]; ++cc; }
			++rr;
		}
		memcpy(D, Q, (size_t)(m - 1) * (m - 1) * sizeof(double));
		for (int r = j; r + 1 < m; ++r) node[r] = node[r + 1];
		--m;
	}
	merges[2 * t] = node[0]; merges[2 * t + 1] = node[1];                        /* :27 */
	free(D); free(Q); free(dn); free(node);
	return 0;
}
