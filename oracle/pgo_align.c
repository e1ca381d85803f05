/* pgo_align.c -- ORACLE (test infrastructure only).
 *
 * Base-level alignment of chained regions, restating mm_align_skeleton / mm_align1 / mm_align1_inv and
 * their helpers (align.c:9-45,47-167,240-314,316-344,355-498,575-1022) for the configuration pangraph
 * uses: no splicing, no short-read mode, no HPC, no query-strand mode, no ALT contigs, no junctions.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include "pgo.h"

#define SEED_LONG_JOIN (1ULL<<40)
#define SEED_IGNORE    (1ULL<<41)
#define SEED_TANDEM    (1ULL<<42)
#define SEED_SELF      (1ULL<<43)

static void gen_mat(int8_t *mat, int8_t a, int8_t b, int8_t sc_ambi) /* align.c:9-22, m=5 */
{
	a = a < 0 ? -a : a; b = b > 0 ? -b : b; sc_ambi = sc_ambi > 0 ? -sc_ambi : sc_ambi;
	for (int i = 0; i < 4; ++i) {
		for (int j = 0; j < 4; ++j) mat[i * 5 + j] = i == j ? a : b;
		mat[i * 5 + 4] = sc_ambi;
	}
	for (int j = 0; j < 5; ++j) mat[20 + j] = sc_ambi;
}

static void rev_bytes(uint32_t len, uint8_t *s) { for (uint32_t i = 0; i < len >> 1; ++i) { uint8_t t = s[i]; s[i] = s[len-1-i]; s[len-1-i] = t; } }

static void get_tseq(const pgo_index_t *ix, uint32_t rid, int32_t st, int32_t en, uint8_t *out) /* index.c:152-162 */
{
	const mm_idx_seq_t *s = &ix->hdr.seq[rid];
	if ((uint32_t)st >= s->len) return;
	if ((uint32_t)en > s->len) en = (int32_t)s->len;
	if (en > st) memcpy(out, ix->nt4 + s->offset + st, (size_t)(en - st));
}

static void track_zdrop(int32_t score, int i, int j, int32_t *max, int *max_i, int *max_j, int e, int *max_zdrop, int pos[2][2]) /* align.c:32-45 */
{
	if (score < *max) {
		int li = i - *max_i, lj = j - *max_j;
		int diff = li > lj ? li - lj : lj - li;
		int z = *max - score - diff * e;
		if (z > *max_zdrop) {
			*max_zdrop = z;
			pos[0][0] = *max_i, pos[0][1] = i;
			pos[1][0] = *max_j, pos[1][1] = j;
		}
	} else *max = score, *max_i = i, *max_j = j;
}

static int test_zdrop(const mm_mapopt_t *opt, const uint8_t *qseq, const uint8_t *tseq, uint32_t n_cigar, const uint32_t *cigar, const int8_t *mat) /* align.c:47-89 */
{
	int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
	int pos[2][2] = {{-1, -1}, {-1, -1}}, q_len, t_len;
	for (uint32_t k = 0; k < n_cigar; ++k) {
		uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == MM_CIGAR_MATCH) {
			for (uint32_t l = 0; l < len; ++l) {
				score += mat[tseq[i + l] * 5 + qseq[j + l]];
				track_zdrop(score, i + (int)l, j + (int)l, &max, &max_i, &max_j, opt->e, &max_zdrop, pos);
			}
			i += len, j += len;
		} else if (op == MM_CIGAR_INS || op == MM_CIGAR_DEL || op == 3) {
			score -= opt->q + opt->e * (int)len;
			if (op == MM_CIGAR_INS) j += len; else i += len;
			track_zdrop(score, i, j, &max, &max_i, &max_j, opt->e, &max_zdrop, pos);
		}
	}
	q_len = pos[1][1] - pos[1][0], t_len = pos[0][1] - pos[0][0];
	if (!(opt->flag & (MM_F_SPLICE|MM_F_SR|MM_F_FOR_ONLY|MM_F_REV_ONLY)) && max_zdrop > opt->zdrop_inv && q_len < opt->max_gap && t_len < opt->max_gap) {
		uint8_t *qseq2 = (uint8_t*)malloc((size_t)(q_len > 0 ? q_len : 1));
		int q_off, t_off;
		for (i = 0; i < q_len; ++i) {
			int c = qseq[pos[1][1] - i - 1];
			qseq2[i] = c >= 4 ? 4 : 3 - c;
		}
		score = pgo_ll_i16(q_len, qseq2, 5, mat, t_len, tseq + pos[0][0], opt->q, opt->e, &q_off, &t_off);
		free(qseq2);
		if (score >= opt->min_chain_score * opt->a && score >= opt->min_dp_max) return 2;
	}
	return max_zdrop > opt->zdrop ? 1 : 0;
}

static void cigar_append(mm_reg1_t *r, uint32_t n_cigar, const uint32_t *cigar) /* align.c:291-314; capacity rounding kept so mm_extra_t::capacity matches */
{
	mm_extra_t *p;
	if (n_cigar == 0) return;
	if (r->p == 0) {
		uint32_t cap = n_cigar + (uint32_t)sizeof(mm_extra_t) / 4;
		--cap, cap |= cap >> 1, cap |= cap >> 2, cap |= cap >> 4, cap |= cap >> 8, cap |= cap >> 16, ++cap;
		r->p = (mm_extra_t*)calloc(cap, 4);
		r->p->capacity = cap;
	} else if (r->p->n_cigar + n_cigar + sizeof(mm_extra_t) / 4 > r->p->capacity) {
		uint32_t cap = r->p->n_cigar + n_cigar + (uint32_t)sizeof(mm_extra_t) / 4;
		--cap, cap |= cap >> 1, cap |= cap >> 2, cap |= cap >> 4, cap |= cap >> 8, cap |= cap >> 16, ++cap;
		r->p = (mm_extra_t*)realloc(r->p, (size_t)cap * 4);
		r->p->capacity = cap;
	}
	p = r->p;
	if (p->n_cigar > 0 && (p->cigar[p->n_cigar - 1] & 0xf) == (cigar[0] & 0xf)) {
		p->cigar[p->n_cigar - 1] += cigar[0] >> 4 << 4;
		if (n_cigar > 1) memcpy(p->cigar + p->n_cigar, cigar + 1, (size_t)(n_cigar - 1) * 4);
		p->n_cigar += n_cigar - 1;
	} else {
		memcpy(p->cigar + p->n_cigar, cigar, (size_t)n_cigar * 4);
		p->n_cigar += n_cigar;
	}
}

static void fix_cigar(mm_reg1_t *r, const uint8_t *qseq, const uint8_t *tseq, int *qshift, int *tshift) /* align.c:91-167 */
{
	mm_extra_t *p = r->p;
	int32_t toff = 0, qoff = 0, to_shrink = 0;
	uint32_t k;
	*qshift = *tshift = 0;
	if (p->n_cigar <= 1) return;
	for (k = 0; k < p->n_cigar; ++k) { /* shift indels to the left while the flanking bases allow it */
		uint32_t op = p->cigar[k] & 0xf, len = p->cigar[k] >> 4;
		if (len == 0) to_shrink = 1;
		if (op == MM_CIGAR_MATCH) toff += len, qoff += len;
		else if (op == MM_CIGAR_INS || op == MM_CIGAR_DEL) {
			if (k > 0 && k < p->n_cigar - 1 && (p->cigar[k-1] & 0xf) == 0 && (p->cigar[k+1] & 0xf) == 0) {
				int l, prev_len = (int)(p->cigar[k-1] >> 4);
				const uint8_t *sq = op == MM_CIGAR_INS ? qseq : tseq;
				int32_t o = op == MM_CIGAR_INS ? qoff : toff;
				for (l = 0; l < prev_len; ++l)
					if (sq[o - 1 - l] != sq[o + (int)len - 1 - l]) break;
				if (l > 0) p->cigar[k-1] -= (uint32_t)l << 4, p->cigar[k+1] += (uint32_t)l << 4, qoff -= l, toff -= l;
				if (l == prev_len) to_shrink = 1;
			}
			if (op == MM_CIGAR_INS) qoff += len; else toff += len;
		} else if (op == 3) toff += len;
	}
	assert(qoff == r->qe - r->qs && toff == r->re - r->rs);
	for (k = 0; k + 2 < p->n_cigar; ++k) { /* runs like 5I6D7I become one I and one D */
		if ((p->cigar[k] & 0xf) > 0 && (p->cigar[k] & 0xf) + (p->cigar[k+1] & 0xf) == 3) {
			uint32_t l, s[3] = {0, 0, 0};
			for (l = k; l < p->n_cigar; ++l) {
				uint32_t op = p->cigar[l] & 0xf;
				if (op == MM_CIGAR_INS || op == MM_CIGAR_DEL || p->cigar[l] >> 4 == 0) s[op] += p->cigar[l] >> 4;
				else break;
			}
			if (s[1] > 0 && s[2] > 0 && l - k > 2) {
				p->cigar[k] = s[1] << 4 | MM_CIGAR_INS;
				p->cigar[k+1] = s[2] << 4 | MM_CIGAR_DEL;
				for (k += 2; k < l; ++k) p->cigar[k] &= 0xf;
				to_shrink = 1;
			}
			k = l;
		}
	}
	if (to_shrink) {
		int32_t l = 0;
		for (k = 0; k < p->n_cigar; ++k)
			if (p->cigar[k] >> 4 != 0) p->cigar[l++] = p->cigar[k];
		p->n_cigar = (uint32_t)l;
		for (k = 0, l = 0; k < p->n_cigar; ++k)
			if (k == p->n_cigar - 1 || (p->cigar[k] & 0xf) != (p->cigar[k+1] & 0xf)) p->cigar[l++] = p->cigar[k];
			else p->cigar[k+1] += p->cigar[k] >> 4 << 4;
		p->n_cigar = (uint32_t)l;
	}
	if ((p->cigar[0] & 0xf) == MM_CIGAR_INS || (p->cigar[0] & 0xf) == MM_CIGAR_DEL) { /* drop a leading I/D */
		int32_t l = (int32_t)(p->cigar[0] >> 4);
		if ((p->cigar[0] & 0xf) == MM_CIGAR_INS) {
			if (r->rev) r->qe -= l; else r->qs += l;
			*qshift = l;
		} else r->rs += l, *tshift = l;
		--p->n_cigar;
		memmove(p->cigar, p->cigar + 1, (size_t)p->n_cigar * 4);
	}
}

static void update_extra(mm_reg1_t *r, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int8_t q, int8_t e) /* align.c:240-289, log_gap=1, no eqx */
{
	uint32_t k, l;
	int32_t qshift, tshift, toff = 0, qoff = 0;
	double s = 0.0, max = 0.0;
	mm_extra_t *p = r->p;
	if (p == 0) return;
	fix_cigar(r, qseq, tseq, &qshift, &tshift);
	qseq += qshift, tseq += tshift;
	r->blen = r->mlen = 0;
	for (k = 0; k < p->n_cigar; ++k) {
		uint32_t op = p->cigar[k] & 0xf, len = p->cigar[k] >> 4;
		if (op == MM_CIGAR_MATCH) {
			int n_ambi = 0, n_diff = 0;
			for (l = 0; l < len; ++l) {
				int cq = qseq[qoff + l], ct = tseq[toff + l];
				if (ct > 3 || cq > 3) ++n_ambi;
				else if (ct != cq) ++n_diff;
				s += mat[ct * 5 + cq];
				if (s < 0) s = 0; else max = max > s ? max : s;
			}
			r->blen += len - n_ambi, r->mlen += len - (n_ambi + n_diff), p->n_ambi += n_ambi;
			toff += len, qoff += len;
		} else if (op == MM_CIGAR_INS || op == MM_CIGAR_DEL) {
			int n_ambi = 0;
			const uint8_t *sq = op == MM_CIGAR_INS ? qseq + qoff : tseq + toff;
			for (l = 0; l < len; ++l) if (sq[l] > 3) ++n_ambi;
			r->blen += len - n_ambi, p->n_ambi += n_ambi;
			s -= q + (double)e * pgo_log2f_approx((float)(1.0 + len));
			if (s < 0) s = 0;
			if (op == MM_CIGAR_INS) qoff += len; else toff += len;
		} else if (op == 3) toff += len;
	}
	p->dp_max = (int32_t)(max + .499);
	assert(qoff == r->qe - r->qs && toff == r->re - r->rs);
}

static void adjust_minier(const pgo_index_t *ix, const pg128 *a, int32_t *r, int32_t *q) /* align.c:355-371, non-HPC */
{
	*r = (int32_t)a->x - (ix->hdr.k >> 1);
	*q = (int32_t)a->y - (ix->hdr.k >> 1);
}

static int *long_gaps(int as1, int cnt1, const pg128 *a, int min_gap, int *n_) /* align.c:373-390 */
{
	int i, n, *K;
	*n_ = 0;
	for (i = 1, n = 0; i < cnt1; ++i) {
		int gap = ((int32_t)a[as1+i].y - (int32_t)a[as1+i-1].y) - ((int32_t)a[as1+i].x - (int32_t)a[as1+i-1].x);
		if (gap < -min_gap || gap > min_gap) ++n;
	}
	if (n <= 1) return 0;
	K = (int*)malloc((size_t)n * sizeof(int));
	for (i = 1, n = 0; i < cnt1; ++i) {
		int gap = ((int32_t)a[as1+i].y - (int32_t)a[as1+i-1].y) - ((int32_t)a[as1+i].x - (int32_t)a[as1+i-1].x);
		if (gap < -min_gap || gap > min_gap) K[n++] = i;
	}
	*n_ = n;
	return K;
}

static void filter_bad_seeds(int as1, int cnt1, pg128 *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt) /* align.c:392-431 */
{
	int max_st, max_en, n, i, k, max, *K;
	K = long_gaps(as1, cnt1, a, min_gap, &n);
	if (K == 0) return;
	max = 0, max_st = max_en = -1;
	for (k = 0;; ++k) {
		int gap, l, n_ins = 0, n_del = 0, qs, rs, max_diff = 0, max_diff_l = -1;
		if (k == n || k >= max_en) {
			if (max_en > 0)
				for (i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= SEED_IGNORE;
			max = 0, max_st = max_en = -1;
			if (k == n) break;
		}
		i = K[k];
		gap = ((int32_t)a[as1+i].y - (int32_t)a[as1+i-1].y) - (int32_t)(a[as1+i].x - a[as1+i-1].x);
		if (gap > 0) n_ins += gap; else n_del += -gap;
		qs = (int32_t)a[as1+i-1].y;
		rs = (int32_t)a[as1+i-1].x;
		for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
			int j = K[l], diff;
			if ((int32_t)a[as1+j].y - qs > max_ext_len || (int32_t)a[as1+j].x - rs > max_ext_len) break;
			gap = ((int32_t)a[as1+j].y - (int32_t)a[as1+j-1].y) - (int32_t)(a[as1+j].x - a[as1+j-1].x);
			if (gap > 0) n_ins += gap; else n_del += -gap;
			diff = n_ins + n_del - abs(n_ins - n_del);
			if (max_diff < diff) max_diff = diff, max_diff_l = l;
		}
		if (max_diff > diff_thres && max_diff > max) max = max_diff, max_st = k, max_en = max_diff_l;
	}
	free(K);
}

static void filter_bad_seeds_alt(int as1, int cnt1, pg128 *a, int min_gap, int max_ext) /* align.c:433-469 */
{
	int n, k, *K;
	K = long_gaps(as1, cnt1, a, min_gap, &n);
	if (K == 0) return;
	for (k = 0; k < n;) {
		int i = K[k], l;
		int gap1 = ((int32_t)a[as1+i].y - (int32_t)a[as1+i-1].y) - ((int32_t)a[as1+i].x - (int32_t)a[as1+i-1].x);
		int re1 = (int32_t)a[as1+i].x, qe1 = (int32_t)a[as1+i].y;
		gap1 = gap1 > 0 ? gap1 : -gap1;
		for (l = k + 1; l < n; ++l) {
			int j = K[l], gap2, q_span_pre, rs2, qs2, m;
			if ((int32_t)a[as1+j].y - qe1 > max_ext || (int32_t)a[as1+j].x - re1 > max_ext) break;
			gap2 = ((int32_t)a[as1+j].y - (int32_t)a[as1+j-1].y) - (int32_t)(a[as1+j].x - a[as1+j-1].x);
			q_span_pre = a[as1+j-1].y >> 32 & 0xff;
			rs2 = (int32_t)a[as1+j-1].x + q_span_pre;
			qs2 = (int32_t)a[as1+j-1].y + q_span_pre;
			m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
			gap2 = gap2 > 0 ? gap2 : -gap2;
			if (m > gap1 + gap2) break;
			re1 = (int32_t)a[as1+j].x, qe1 = (int32_t)a[as1+j].y;
			gap1 = gap2;
		}
		if (l > k + 1) {
			int j, end = K[l - 1];
			for (j = K[k]; j < end; ++j) a[as1 + j].y |= SEED_IGNORE;
			a[as1 + end].y |= SEED_LONG_JOIN;
		}
		k = l;
	}
	free(K);
}

static void fix_bad_ends(const mm_reg1_t *r, const pg128 *a, int bw, int min_match, int32_t *as, int32_t *cnt) /* align.c:471-509 */
{
	int32_t i, l, m;
	*as = r->as, *cnt = r->cnt;
	if (r->cnt < 3) return;
	m = l = a[r->as].y >> 32 & 0xff;
	for (i = r->as + 1; i < r->as + r->cnt - 1; ++i) {
		int32_t lq, lr, min, max, q_span = a[i].y >> 32 & 0xff;
		if (a[i].y & SEED_LONG_JOIN) break;
		lr = (int32_t)a[i].x - (int32_t)a[i-1].x;
		lq = (int32_t)a[i].y - (int32_t)a[i-1].y;
		min = lr < lq ? lr : lq, max = lr > lq ? lr : lq;
		if (max - min > l >> 1) *as = i;
		l += min;
		m += min < q_span ? min : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
	}
	*cnt = r->as + r->cnt - *as;
	m = l = a[r->as + r->cnt - 1].y >> 32 & 0xff;
	for (i = r->as + r->cnt - 2; i > *as; --i) {
		int32_t lq, lr, min, max, q_span = a[i+1].y >> 32 & 0xff;
		if (a[i+1].y & SEED_LONG_JOIN) break;
		lr = (int32_t)a[i+1].x - (int32_t)a[i].x;
		lq = (int32_t)a[i+1].y - (int32_t)a[i].y;
		min = lr < lq ? lr : lq, max = lr > lq ? lr : lq;
		if (max - min > l >> 1) *cnt = i + 1 - *as;
		l += min;
		m += min < q_span ? min : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
	}
}

/* align.c:316-344: dispatch of one DP problem (only the dual-gap kernel is reachable: asm presets have q != q2) */
static void align_pair(const mm_mapopt_t *opt, int qlen, const uint8_t *qseq, int tlen, const uint8_t *tseq, const int8_t *mat, int w, int end_bonus, int zdrop, int flag, pgo_extz_t *ez)
{
	if (opt->max_sw_mat > 0 && (int64_t)tlen * qlen > opt->max_sw_mat) {
		ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
		ez->max = 0, ez->score = ez->mqe = ez->mte = PGO_NEG_INF;
		ez->n_cigar = 0, ez->reach_end = 0;
		ez->zdropped = 1;
	} else {
		assert(!(opt->q == opt->q2 && opt->e == opt->e2)); /* single-affine ksw_extz2_sse is out of scope */
		pgo_extd2(qlen, qseq, tlen, tseq, 5, mat, (int8_t)opt->q, (int8_t)opt->e, (int8_t)opt->q2, (int8_t)opt->e2, w, zdrop, end_bonus, flag, ez);
	}
}

static void align1(const mm_mapopt_t *opt, const pgo_index_t *ix, int qlen, uint8_t *qseq0[2], mm_reg1_t *r, mm_reg1_t *r2, int n_a, pg128 *a, pgo_extz_t *ez) /* align.c:575-828 */
{
	int32_t rid = (int32_t)(a[r->as].x << 1 >> 33), rev = (int32_t)(a[r->as].x >> 63), as1, cnt1;
	uint8_t *tseq, *qseq;
	int32_t i, l, bw, bw_long, dropped = 0, rs0, re0, qs0, qe0;
	int32_t rs, re, qs, qe, rs1, qs1, re1, qe1;
	int32_t tlen_ref = (int32_t)ix->hdr.seq[rid].len;
	int8_t mat[25];

	r2->cnt = 0;
	if (r->cnt == 0) return;
	gen_mat(mat, (int8_t)opt->a, (int8_t)opt->b, (int8_t)opt->sc_ambi);
	bw = (int)(opt->bw * 1.5 + 1.);
	bw_long = (int)(opt->bw_long * 1.5 + 1.);
	if (bw_long < bw) bw_long = bw;

	if (!(opt->flag & MM_F_NO_END_FLT)) fix_bad_ends(r, a, opt->bw, opt->min_chain_score * 2, &as1, &cnt1);
	else as1 = r->as, cnt1 = r->cnt;
	filter_bad_seeds(as1, cnt1, a, 10, 40, opt->max_gap >> 1, 10);
	filter_bad_seeds_alt(as1, cnt1, a, 30, opt->max_gap >> 1);
	adjust_minier(ix, &a[as1], &rs, &qs);
	adjust_minier(ix, &a[as1 + cnt1 - 1], &re, &qe);
	assert(cnt1 > 0);

	/* DP windows beyond both chain ends (align.c:633-693) */
	rs0 = (int32_t)a[r->as].x + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
	qs0 = (int32_t)a[r->as].y + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
	if (rs0 < 0) rs0 = 0;
	assert(qs0 >= 0);
	rs1 = qs1 = 0;
	for (i = r->as - 1, l = 0; i >= 0 && a[i].x >> 32 == a[r->as].x >> 32; --i) {
		int32_t x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		int32_t y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		if (x < rs0 && y < qs0) {
			if (++l > opt->min_cnt) {
				l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
				rs1 = rs0 - l, qs1 = qs0 - l;
				if (rs1 < 0) rs1 = 0;
				break;
			}
		}
	}
	if (qs > 0 && rs > 0) {
		l = qs < opt->max_gap ? qs : opt->max_gap;
		qs1 = qs1 > qs - l ? qs1 : qs - l;
		qs0 = qs0 < qs1 ? qs0 : qs1;
		l += l * opt->a > opt->q ? (l * opt->a - opt->q) / opt->e : 0;
		l = l < opt->max_gap ? l : opt->max_gap;
		l = l < rs ? l : rs;
		rs1 = rs1 > rs - l ? rs1 : rs - l;
		rs0 = rs0 < rs1 ? rs0 : rs1;
		rs0 = rs0 < rs ? rs0 : rs;
	} else rs0 = rs, qs0 = qs;
	re0 = (int32_t)a[r->as + r->cnt - 1].x + 1;
	qe0 = (int32_t)a[r->as + r->cnt - 1].y + 1;
	re1 = tlen_ref, qe1 = qlen;
	for (i = r->as + r->cnt, l = 0; i < n_a && a[i].x >> 32 == a[r->as].x >> 32; ++i) {
		int32_t x = (int32_t)a[i].x + 1, y = (int32_t)a[i].y + 1;
		if (x > re0 && y > qe0) {
			if (++l > opt->min_cnt) {
				l = x - re0 > y - qe0 ? x - re0 : y - qe0;
				re1 = re0 + l, qe1 = qe0 + l;
				break;
			}
		}
	}
	if (qe < qlen && re < tlen_ref) {
		l = qlen - qe < opt->max_gap ? qlen - qe : opt->max_gap;
		qe1 = qe1 < qe + l ? qe1 : qe + l;
		qe0 = qe0 > qe1 ? qe0 : qe1;
		l += l * opt->a > opt->q ? (l * opt->a - opt->q) / opt->e : 0;
		l = l < opt->max_gap ? l : opt->max_gap;
		l = l < tlen_ref - re ? l : tlen_ref - re;
		re1 = re1 < re + l ? re1 : re + l;
		re0 = re0 > re1 ? re0 : re1;
	} else re0 = re, qe0 = qe;
	if (a[r->as].y & SEED_SELF) {
		int max_ext = r->qs > r->rs ? r->qs - r->rs : r->rs - r->qs;
		if (r->rs - rs0 > max_ext) rs0 = r->rs - max_ext;
		if (r->qs - qs0 > max_ext) qs0 = r->qs - max_ext;
		max_ext = r->qe > r->re ? r->qe - r->re : r->re - r->qe;
		if (re0 - r->re > max_ext) re0 = r->re + max_ext;
		if (qe0 - r->qe > max_ext) qe0 = r->qe + max_ext;
	}
	assert(re0 > rs0);
	tseq = (uint8_t*)malloc((size_t)(re0 - rs0));

	if (qs > 0 && rs > 0) { /* left extension on reversed sequences (align.c:702-722) */
		qseq = &qseq0[rev][qs0];
		get_tseq(ix, (uint32_t)rid, rs0, rs, tseq);
		rev_bytes((uint32_t)(qs - qs0), qseq);
		rev_bytes((uint32_t)(rs - rs0), tseq);
		align_pair(opt, qs - qs0, qseq, rs - rs0, tseq, mat, bw, opt->end_bonus, r->split_inv ? opt->zdrop_inv : opt->zdrop,
		           PGO_EZ_EXTZ_ONLY | PGO_EZ_RIGHT | PGO_EZ_REV_CIGAR, ez);
		if (ez->n_cigar > 0) {
			cigar_append(r, (uint32_t)ez->n_cigar, ez->cigar);
			r->p->dp_score += (int32_t)ez->max;
		}
		rs1 = rs - (ez->reach_end ? ez->mqe_t + 1 : ez->max_t + 1);
		qs1 = qs - (ez->reach_end ? qs - qs0 : ez->max_q + 1);
		rev_bytes((uint32_t)(qs - qs0), qseq);
	} else rs1 = rs, qs1 = qs;
	re1 = rs, qe1 = qs;
	assert(qs1 >= 0 && rs1 >= 0);

	for (i = 1; i < cnt1; ++i) { /* gap filling (align.c:726-787) */
		if ((a[as1+i].y & (SEED_IGNORE | SEED_TANDEM)) && i != cnt1 - 1) continue;
		adjust_minier(ix, &a[as1 + i], &re, &qe);
		re1 = re, qe1 = qe;
		if (i == cnt1 - 1 || (a[as1+i].y & SEED_LONG_JOIN) || (qe - qs >= opt->min_ksw_len && re - rs >= opt->min_ksw_len)) {
			int j, bw1 = bw_long, zdrop_code;
			if (a[as1+i].y & SEED_LONG_JOIN) bw1 = qe - qs > re - rs ? qe - qs : re - rs;
			qseq = &qseq0[rev][qs];
			get_tseq(ix, (uint32_t)rid, rs, re, tseq);
			align_pair(opt, qe - qs, qseq, re - rs, tseq, mat, bw1, -1, opt->zdrop, PGO_EZ_APPROX_MAX, ez);
			if ((zdrop_code = test_zdrop(opt, qseq, tseq, (uint32_t)ez->n_cigar, ez->cigar, mat)) != 0)
				align_pair(opt, qe - qs, qseq, re - rs, tseq, mat, bw1, -1, zdrop_code == 2 ? opt->zdrop_inv : opt->zdrop, 0, ez);
			if (ez->n_cigar > 0) cigar_append(r, (uint32_t)ez->n_cigar, ez->cigar);
			if (ez->zdropped) {
				if (!r->p) {
					uint32_t cap = (uint32_t)sizeof(mm_extra_t) / 4;
					--cap, cap |= cap >> 1, cap |= cap >> 2, cap |= cap >> 4, cap |= cap >> 8, cap |= cap >> 16, ++cap;
					r->p = (mm_extra_t*)calloc(cap, 4);
					r->p->capacity = cap;
				}
				for (j = i - 1; j >= 0; --j)
					if ((int32_t)a[as1 + j].x <= rs + ez->max_t) break;
				dropped = 1;
				if (j < 0) j = 0;
				r->p->dp_score += (int32_t)ez->max;
				re1 = rs + (ez->max_t + 1);
				qe1 = qs + (ez->max_q + 1);
				if (cnt1 - (j + 1) >= opt->min_cnt) {
					pgo_split_reg(r, r2, as1 + j + 1 - r->as, qlen, a);
					if (zdrop_code == 2) r2->split_inv = 1;
				}
				break;
			} else r->p->dp_score += ez->score;
			rs = re, qs = qe;
		}
	}

	if (!dropped && qe < qe0 && re < re0) { /* right extension (align.c:789-805) */
		qseq = &qseq0[rev][qe];
		get_tseq(ix, (uint32_t)rid, re, re0, tseq);
		align_pair(opt, qe0 - qe, qseq, re0 - re, tseq, mat, bw, opt->end_bonus, opt->zdrop, PGO_EZ_EXTZ_ONLY, ez);
		if (ez->n_cigar > 0) {
			cigar_append(r, (uint32_t)ez->n_cigar, ez->cigar);
			r->p->dp_score += (int32_t)ez->max;
		}
		re1 = re + (ez->reach_end ? ez->mqe_t + 1 : ez->max_t + 1);
		qe1 = qe + (ez->reach_end ? qe0 - qe : ez->max_q + 1);
	}
	assert(qe1 <= qlen);

	r->rs = rs1, r->re = re1;
	if (!rev) r->qs = qs1, r->qe = qe1;
	else r->qs = qlen - qe1, r->qe = qlen - qs1;
	assert(re1 - rs1 <= re0 - rs0);
	if (r->p) {
		get_tseq(ix, (uint32_t)rid, rs1, re1, tseq);
		qseq = &qseq0[r->rev][qs1];
		update_extra(r, qseq, tseq, mat, (int8_t)opt->q, (int8_t)opt->e);
	}
	free(tseq);
}

static int align1_inv(const mm_mapopt_t *opt, const pgo_index_t *ix, int qlen, uint8_t *qseq0[2], const mm_reg1_t *r1, const mm_reg1_t *r2, mm_reg1_t *r_inv, pgo_extz_t *ez) /* align.c:830-885 */
{
	int tl, ql, score, ret = 0, q_off, t_off;
	uint8_t *tseq, *qseq;
	int8_t mat[25];
	memset(r_inv, 0, sizeof(mm_reg1_t));
	if (!(r1->split & 1) || !(r2->split & 2)) return 0;
	if (r1->id != r1->parent && r1->parent != -2) return 0;
	if (r2->id != r2->parent && r2->parent != -2) return 0;
	if (r1->rid != r2->rid || r1->rev != r2->rev) return 0;
	ql = r1->rev ? r1->qs - r2->qe : r2->qs - r1->qe;
	tl = r2->rs - r1->re;
	if (ql < opt->min_chain_score || ql > opt->max_gap) return 0;
	if (tl < opt->min_chain_score || tl > opt->max_gap) return 0;
	gen_mat(mat, (int8_t)opt->a, (int8_t)opt->b, (int8_t)opt->sc_ambi);
	tseq = (uint8_t*)malloc((size_t)tl);
	get_tseq(ix, (uint32_t)r1->rid, r1->re, r2->rs, tseq);
	qseq = r1->rev ? &qseq0[0][r2->qe] : &qseq0[1][qlen - r2->qs];
	rev_bytes((uint32_t)ql, qseq);
	rev_bytes((uint32_t)tl, tseq);
	score = pgo_ll_i16(ql, qseq, 5, mat, tl, tseq, opt->q, opt->e, &q_off, &t_off);
	rev_bytes((uint32_t)ql, qseq);
	rev_bytes((uint32_t)tl, tseq);
	if (score < opt->min_dp_max) goto done;
	q_off = ql - (q_off + 1), t_off = tl - (t_off + 1);
	align_pair(opt, ql - q_off, qseq + q_off, tl - t_off, tseq + t_off, mat, (int)(opt->bw * 1.5), -1, opt->zdrop, PGO_EZ_EXTZ_ONLY, ez);
	if (ez->n_cigar == 0) goto done;
	cigar_append(r_inv, (uint32_t)ez->n_cigar, ez->cigar);
	r_inv->p->dp_score = (int32_t)ez->max;
	r_inv->id = -1;
	r_inv->parent = -1;
	r_inv->inv = 1;
	r_inv->rev = !r1->rev;
	r_inv->rid = r1->rid;
	r_inv->div = -1.0f;
	if (r_inv->rev == 0) {
		r_inv->qs = r2->qe + q_off;
		r_inv->qe = r_inv->qs + ez->max_q + 1;
	} else {
		r_inv->qe = r2->qs - q_off;
		r_inv->qs = r_inv->qe - (ez->max_q + 1);
	}
	r_inv->rs = r1->re + t_off;
	r_inv->re = r_inv->rs + ez->max_t + 1;
	update_extra(r_inv, &qseq[q_off], &tseq[t_off], mat, (int8_t)opt->q, (int8_t)opt->e);
	ret = 1;
done:
	free(tseq);
	return ret;
}

static mm_reg1_t *insert_reg(const mm_reg1_t *r, int i, int *n_regs, mm_reg1_t *regs) /* align.c:887-895 */
{
	regs = (mm_reg1_t*)realloc(regs, (size_t)(*n_regs + 1) * sizeof(mm_reg1_t));
	if (i + 1 != *n_regs) memmove(&regs[i + 2], &regs[i + 1], sizeof(mm_reg1_t) * (size_t)(*n_regs - i - 1));
	regs[i + 1] = *r;
	++*n_regs;
	return regs;
}

static int32_t recal_max_dp(const mm_reg1_t *r, double b2, int32_t match_sc) /* align.c:919-934 */
{
	int32_t n_gap = 0, n_mis;
	double gap_cost = 0.0;
	if (r->p == 0) return -1;
	for (uint32_t i = 0; i < r->p->n_cigar; ++i) {
		int32_t op = r->p->cigar[i] & 0xf, len = (int32_t)(r->p->cigar[i] >> 4);
		if (op == MM_CIGAR_INS || op == MM_CIGAR_DEL) {
			gap_cost += b2 + (double)pgo_log2f_approx((float)(1.0 + len));
			n_gap += len;
		}
	}
	n_mis = r->blen + (int32_t)r->p->n_ambi - r->mlen - n_gap;
	return (int32_t)(match_sc * (r->mlen - b2 * n_mis - gap_cost) + .499);
}

static double event_identity(const mm_reg1_t *r) /* align.c:897-917 */
{
	int32_t n_gapo = 0, n_gap = 0;
	if (r->p == 0) return -1.0f;
	for (uint32_t i = 0; i < r->p->n_cigar; ++i) {
		int32_t op = r->p->cigar[i] & 0xf, len = (int32_t)(r->p->cigar[i] >> 4);
		if (op == MM_CIGAR_INS || op == MM_CIGAR_DEL) ++n_gapo, n_gap += len;
	}
	return (double)r->mlen / (r->blen + (int32_t)r->p->n_ambi - n_gap + n_gapo);
}

void pgo_update_dp_max(int qlen, int n_regs, mm_reg1_t *regs, float frac, int a, int b) /* align.c:936-960 */
{
	int32_t max = -1, max2 = -1, i, max_i = -1;
	double div, b2;
	if (n_regs < 2) return;
	for (i = 0; i < n_regs; ++i) {
		mm_reg1_t *r = &regs[i];
		if (r->p == 0) continue;
		if (r->p->dp_max > max) max2 = max, max = r->p->dp_max, max_i = i;
		else if (r->p->dp_max > max2) max2 = r->p->dp_max;
	}
	if (max_i < 0 || max < 0 || max2 < 0) return;
	if (regs[max_i].qe - regs[max_i].qs < (double)qlen * frac) return;
	if (max2 < (double)max * frac) return;
	div = 1. - event_identity(&regs[max_i]);
	if (div < 0.02) div = 0.02;
	b2 = 0.5 / div;
	if (b2 * a < b) b2 = (double)a / b;
	for (i = 0; i < n_regs; ++i) {
		mm_reg1_t *r = &regs[i];
		if (r->p == 0) continue;
		r->p->dp_max = recal_max_dp(r, b2, a);
		if (r->p->dp_max < 0) r->p->dp_max = 0;
	}
}

mm_reg1_t *pgo_align_skeleton(const mm_mapopt_t *opt, const pgo_index_t *ix, int qlen, const char *qstr, int *n_regs_, mm_reg1_t *regs, pg128 *a) /* align.c:962-1022 */
{
	int32_t i, n_regs = *n_regs_, n_a;
	uint8_t *qseq0[2];
	pgo_extz_t ez;
	qseq0[0] = (uint8_t*)malloc((size_t)qlen * 2);
	qseq0[1] = qseq0[0] + qlen;
	for (i = 0; i < qlen; ++i) {
		qseq0[0][i] = pgo_nt4[(uint8_t)qstr[i]];
		qseq0[1][qlen - 1 - i] = qseq0[0][i] < 4 ? 3 - qseq0[0][i] : 4;
	}
	n_a = pgo_squeeze_a(n_regs, regs, a);
	memset(&ez, 0, sizeof(ez));
	for (i = 0; i < n_regs; ++i) {
		mm_reg1_t r2;
		align1(opt, ix, qlen, qseq0, &regs[i], &r2, n_a, a, &ez);
		if (r2.cnt > 0) regs = insert_reg(&r2, i, &n_regs, regs);
		if (i > 0 && regs[i].split_inv && !(opt->flag & MM_F_NO_INV)) {
			if (align1_inv(opt, ix, qlen, qseq0, &regs[i-1], &regs[i], &r2, &ez)) {
				regs = insert_reg(&r2, i, &n_regs, regs);
				++i;
			}
		}
	}
	*n_regs_ = n_regs;
	free(qseq0[0]);
	free(ez.cigar);
	pgo_filter_regs(opt, qlen, n_regs_, regs);
	if (qlen >= opt->rank_min_len) {
		pgo_update_dp_max(qlen, *n_regs_, regs, opt->rank_frac, opt->a, opt->b);
		pgo_filter_regs(opt, qlen, n_regs_, regs);
	}
	pgo_hit_sort(n_regs_, regs);
	return regs;
}
