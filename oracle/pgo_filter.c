/* pgo_filter.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the step right behind the alignment path.
 *
 *   split_matches / keep_groups / generate_subalignment / side_patches   packages/pangraph/src/pangraph/split_matches.rs:13-237
 *   add_flanking_indel, cigar_matches_len, cigar_total_len               packages/pangraph/src/align/bam/cigar.rs:14-24,60-96
 *   alignment_energy2                                                     packages/pangraph/src/align/energy.rs:37-54
 *   filter_matches, is_match_compatible, update_intervals                 packages/pangraph/src/pangraph/graph_merging.rs:187-240
 *   Interval::has_overlap_with                                            packages/pangraph/src/utils/interval.rs:42-46
 *   the caller: self_merge                                                packages/pangraph/src/pangraph/graph_merging.rs:95-128
 *
 * The reference is Rust and cannot be built in this image; this file is pinned by the known-answer vectors of the reference's unit
 * tests (split_matches.rs:243-593, energy.rs:86-110, graph_merging.rs:253-375), checked in tests/test_filter_cpu.py.
 * Records use the layout of pga_match_t (include/pga_align.h); CIGAR operations are len << 4 | op with op in "MIDNSHP=X".
 * Order: the reference sorts by energy with a STABLE sort over whatever order its parallel aligner returned (nondeterministic,
 * align_with_minimap2_lib.rs:65); here ties keep the order of the input records.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct {
	int32_t group, qry, ref, qry_len, qry_start, qry_end, ref_len, ref_start, ref_end, matches, length, quality, reverse, align, n_ambi, inv;
	double divergence;
	uint64_t cigar_off;
	uint32_t n_cigar, pad;
} pgo_match_t;

static int is_match_op(uint32_t op) { const uint32_t k = op & 15; return k == 0 || k == 7 || k == 8; }   /* M, =, X */

/* split_matches.rs:31-96; groups receives (start, end) index pairs; returns their number, -1 on an operation the reference rejects */
int pgo_keep_groups(const uint32_t *cig, uint32_t n, int thr, int32_t *groups)
{
	int ng = 0;
	long g_start = -1, last_match = -1;
	uint64_t M = 0, I = 0, D = 0;
	for (uint32_t i = 0; i < n; ++i) {
		const uint32_t k = cig[i] & 15, len = cig[i] >> 4;
		if (g_start < 0) { if (!is_match_op(cig[i])) continue; g_start = i; }        /* :41-46: leading indels are skipped */
		if (is_match_op(cig[i])) { M += len; I = 0; D = 0; last_match = i; }          /* :49-55 */
		else if (k == 1) I += len;
		else if (k == 2) D += len;
		else return -1;                                                              /* :62-65 */
		if ((I > D ? I : D) >= (uint64_t)thr) {                                      /* :69-81 */
			if (g_start >= 0 && last_match >= 0 && M >= (uint64_t)thr) { groups[2 * ng] = (int32_t)g_start; groups[2 * ng + 1] = (int32_t)last_match; ++ng; }
			g_start = -1; last_match = -1; M = I = D = 0;
		}
	}
	if (g_start >= 0 && last_match >= 0 && M >= (uint64_t)thr) { groups[2 * ng] = (int32_t)g_start; groups[2 * ng + 1] = (int32_t)last_match; ++ng; }   /* :84-89 */
	return ng;
}

typedef struct { pgo_match_t *m; uint32_t *cig; size_t n, cap, n_ops, cap_ops; } out_t;

static void out_push(out_t *o, pgo_match_t r, const uint32_t *ops, uint32_t n_ops)
{
	if (o->n == o->cap) { o->cap = o->cap ? o->cap * 2 : 64; o->m = (pgo_match_t*)realloc(o->m, o->cap * sizeof(pgo_match_t)); }
	while (o->n_ops + n_ops > o->cap_ops) { o->cap_ops = o->cap_ops ? o->cap_ops * 2 : 1024; o->cig = (uint32_t*)realloc(o->cig, o->cap_ops * sizeof(uint32_t)); }
	r.cigar_off = o->n_ops; r.n_cigar = n_ops;
	memcpy(o->cig + o->n_ops, ops, n_ops * sizeof(uint32_t));
	o->n_ops += n_ops;
	o->m[o->n++] = r;
}

/* cigar.rs:60-96 on a buffer with room for one more operation; kind 1 = insertion, 2 = deletion; leading != 0: from the front */
static uint32_t add_flanking_indel(uint32_t *ops, uint32_t n, uint32_t kind, uint32_t add_len, int leading)
{
	long replace = -1;
	if (leading) { for (uint32_t i = 0; i < n; ++i) { if (is_match_op(ops[i])) break; if ((ops[i] & 15) == kind) replace = i; } }
	else { for (long i = (long)n - 1; i >= 0; --i) { if (is_match_op(ops[i])) break; if ((ops[i] & 15) == kind) replace = i; } }
	if (replace >= 0) { ops[replace] = (((ops[replace] >> 4) + add_len) << 4) | kind; return n; }
	if (leading) { memmove(ops + 1, ops, n * sizeof(uint32_t)); ops[0] = add_len << 4 | kind; }
	else ops[n] = add_len << 4 | kind;
	return n + 1;
}

/* split_matches.rs:13-24 for one alignment, appended to o; returns 0, -1 on a rejected operation */
static int split_one(const pgo_match_t *a, const uint32_t *cig, int thr, out_t *o)
{
	const uint32_t n = a->n_cigar;
	int32_t *groups = (int32_t*)malloc((size_t)(n + 2) * sizeof(int32_t));
	const int ng = pgo_keep_groups(cig, n, thr, groups);
	if (ng < 0) { free(groups); return -1; }
	uint32_t *buf = (uint32_t*)malloc(((size_t)n + 8) * sizeof(uint32_t));
	for (int g = 0; g < ng; ++g) {
		const int s = groups[2 * g], e = groups[2 * g + 1];
		/* :98-146: positions of the group on the query (M, I, =, X) and on the reference (M, D, =, X) */
		int64_t qb = 0, rb = 0, qe, re;
		for (int i = 0; i < s; ++i) { const uint32_t k = cig[i] & 15, len = cig[i] >> 4; if (is_match_op(cig[i]) || k == 1) qb += len; if (is_match_op(cig[i]) || k == 2) rb += len; }
		qe = qb; re = rb;
		for (int i = s; i <= e; ++i) { const uint32_t k = cig[i] & 15, len = cig[i] >> 4; if (is_match_op(cig[i]) || k == 1) qe += len; if (is_match_op(cig[i]) || k == 2) re += len; }
		pgo_match_t r = *a;                                                          /* :150-189 */
		r.ref_start = a->ref_start + (int32_t)rb; r.ref_end = a->ref_start + (int32_t)re;
		if (!a->reverse) { r.qry_start = a->qry_start + (int32_t)qb; r.qry_end = a->qry_start + (int32_t)qe; }
		else { r.qry_start = a->qry_end - (int32_t)qe; r.qry_end = a->qry_end - (int32_t)qb; }
		uint32_t m = (uint32_t)(e - s + 1);
		memcpy(buf, cig + s, m * sizeof(uint32_t));
		int64_t ml = 0, tl = 0;
		for (uint32_t i = 0; i < m; ++i) { tl += buf[i] >> 4; if (is_match_op(buf[i])) ml += buf[i] >> 4; }
		r.matches = (int32_t)ml; r.length = (int32_t)tl;
		/* side_patches, :193-237 */
		if (r.ref_start > 0 && r.ref_start < thr) { const int d = r.ref_start; r.ref_start = 0; r.length += d; m = add_flanking_indel(buf, m, 2, (uint32_t)d, 1); }
		if (r.ref_end < r.ref_len && r.ref_len - r.ref_end < thr) { const int d = r.ref_len - r.ref_end; r.ref_end = r.ref_len; r.length += d; m = add_flanking_indel(buf, m, 2, (uint32_t)d, 0); }
		if (r.qry_start > 0 && r.qry_start < thr) { const int d = r.qry_start; r.qry_start = 0; r.length += d; m = add_flanking_indel(buf, m, 1, (uint32_t)d, !r.reverse); }
		if (r.qry_end < r.qry_len && r.qry_len - r.qry_end < thr) { const int d = r.qry_len - r.qry_end; r.qry_end = r.qry_len; r.length += d; m = add_flanking_indel(buf, m, 1, (uint32_t)d, r.reverse); }
		out_push(o, r, buf, m);
	}
	free(buf); free(groups);
	return 0;
}

/* energy.rs:37-54 */
double pgo_energy2(const pgo_match_t *a, double alpha, double beta)
{
	const int64_t L = a->matches;
	const double M = a->divergence * (double)L;
	int C = 4;
	if (a->qry_start == 0) --C;
	if (a->qry_end == a->qry_len) --C;
	if (a->ref_start == 0) --C;
	if (a->ref_end == a->ref_len) --C;
	return -(double)L + (double)C * alpha + M * beta;
}

typedef struct { double e; size_t idx; } ekey_t;
static int cmp_ekey(const void *x, const void *y)
{
	const ekey_t *a = (const ekey_t*)x, *b = (const ekey_t*)y;
	if (a->e < b->e) return -1;
	if (a->e > b->e) return 1;
	return a->idx < b->idx ? -1 : a->idx > b->idx;      /* stable */
}

/* flags & 1: self_merge's exclusion of self matches + split_matches (graph_merging.rs:105-113); flags & 2: filter_matches per group
 * (:187-216).  Output: malloc'ed records and CIGAR pool (cigar_off relative to it).  Returns the number of records, -1 on a CIGAR
 * operation the reference rejects. */
int64_t pgo_split_filter(int64_t n, const pgo_match_t *m, const uint32_t *cig, int thr, double alpha, double beta, int flags,
                         pgo_match_t **out_m, uint32_t **out_cig, uint64_t *out_n_ops)
{
	out_t o; memset(&o, 0, sizeof(o));
	for (int64_t i = 0; i < n; ++i) {
		if (flags & 1) {
			if (m[i].qry == m[i].ref) continue;                                      /* :107 (names inside a group are its sequence indices) */
			if (split_one(&m[i], cig + m[i].cigar_off, thr, &o) != 0) { free(o.m); free(o.cig); return -1; }
		} else out_push(&o, m[i], cig + m[i].cigar_off, m[i].n_cigar);
	}
	if (!(flags & 2)) { *out_m = o.m; *out_cig = o.cig; *out_n_ops = o.n_ops; return (int64_t)o.n; }
	/* filter_matches, group by group (one find_matches call each), groups in ascending order */
	out_t f; memset(&f, 0, sizeof(f));
	ekey_t *keys = (ekey_t*)malloc((o.n + 1) * sizeof(ekey_t));
	int32_t gmin = INT32_MAX, gmax = INT32_MIN;
	for (size_t i = 0; i < o.n; ++i) { if (o.m[i].group < gmin) gmin = o.m[i].group; if (o.m[i].group > gmax) gmax = o.m[i].group; }
	/* accepted intervals: simple arrays (block = (group, sequence index)) */
	typedef struct { int32_t blk, s, e; } iv_t;
	iv_t *iv = (iv_t*)malloc((2 * o.n + 1) * sizeof(iv_t));
	for (int64_t g = gmin; o.n && g <= gmax; ++g) {
		size_t nk = 0, niv = 0;
		for (size_t i = 0; i < o.n; ++i) if (o.m[i].group == g) {
			const double e = pgo_energy2(&o.m[i], alpha, beta);
			if (e < 0.0) { keys[nk].e = e; keys[nk].idx = i; ++nk; }                  /* :198 */
		}
		if (!nk) continue;
		qsort(keys, nk, sizeof(ekey_t), cmp_ekey);                                   /* :199 */
		for (size_t k = 0; k < nk; ++k) {
			const pgo_match_t *a = &o.m[keys[k].idx];
			int ok = 1;
			for (size_t j = 0; j < niv && ok; ++j) {                                  /* :218-230, interval.rs:42-46 */
				if (iv[j].blk == a->ref && iv[j].e > a->ref_start && iv[j].s < a->ref_end) ok = 0;
				if (iv[j].blk == a->qry && iv[j].e > a->qry_start && iv[j].s < a->qry_end) ok = 0;
			}
			if (!ok) continue;
			out_push(&f, *a, o.cig + a->cigar_off, a->n_cigar);
			iv[niv].blk = a->ref; iv[niv].s = a->ref_start; iv[niv].e = a->ref_end; ++niv;   /* :232-240 */
			iv[niv].blk = a->qry; iv[niv].s = a->qry_start; iv[niv].e = a->qry_end; ++niv;
		}
	}
	free(iv); free(keys); free(o.m); free(o.cig);
	*out_m = f.m; *out_cig = f.cig; *out_n_ops = f.n_ops;
	return (int64_t)f.n;
}
