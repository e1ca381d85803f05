/* pgo_ksw.c -- ORACLE (test infrastructure only).
 *
 * Dual-affine-gap extension / global alignment on anti-diagonals, restating ksw_extd2_sse()
 * (ksw2_extd2_sse.c:34-401), ksw_backtrack / ksw_apply_zdrop / ksw_reset_extz (ksw2.h:111-184), and the
 * striped local-alignment score ksw_ll_qinit / ksw_ll_i16 (ksw2_ll_sse.c:37-152).
 *
 * The SSE code keeps Suzuki-Kasahara DIFFERENCES in int8 lanes:
 *   u[t] = H(r,t)-H(r-1,t-1) ... see ksw2_extd2_sse.c:40-71.  A scalar restatement must keep three of
 * its artefacts to stay bit-exact in banded calls:
 *   (1) whole 16-lane vectors are evaluated: lanes [st0/16*16, (en0+16)/16*16-1] even though only
 *       [st0,en0] is inside the band, so cells just outside the band keep being recomputed from stale
 *       neighbours and feed back into the band through the t-1 dependency;
 *   (2) the score profile s[] is refreshed only for t in [st0, st0+16*ceil((en0-st0+1)/16)) and read past
 *       the end of the sequences (zero padding) (ksw2_extd2_sse.c:165-181);
 *   (3) every add/sub wraps modulo 256 and every compare is a signed int8 compare.
 * So this file evaluates the same per-lane recurrence, one lane at a time, in int8.
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "pgo.h"

static inline int8_t w8(int v) { return (int8_t)(uint8_t)v; } /* wrap like _mm_add_epi8/_mm_sub_epi8 */

static void ez_reset(pgo_extz_t *ez) /* ksw2.h:161-166 */
{
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0, ez->score = ez->mqe = ez->mte = PGO_NEG_INF;
	ez->n_cigar = 0, ez->zdropped = 0, ez->reach_end = 0;
}

static void push_cigar(pgo_extz_t *ez, uint32_t op, int len) /* ksw2.h:111-122 */
{
	if (ez->n_cigar == 0 || op != (ez->cigar[ez->n_cigar - 1] & 0xf)) {
		if (ez->n_cigar == ez->m_cigar) {
			ez->m_cigar = ez->m_cigar ? ez->m_cigar << 1 : 4;
			ez->cigar = (uint32_t*)realloc(ez->cigar, (size_t)ez->m_cigar << 2);
		}
		ez->cigar[ez->n_cigar++] = (uint32_t)len << 4 | op;
	} else ez->cigar[ez->n_cigar - 1] += (uint32_t)len << 4;
}

/* ksw2.h:128-159 with is_rot=1, min_intron_len=0 */
static void backtrack(pgo_extz_t *ez, int is_rev, const uint8_t *p, const int *off, const int *off_end, int n_col, int i0, int j0)
{
	int i = i0, j = j0, r, state = 0;
	ez->n_cigar = 0;
	while (i >= 0 && j >= 0) {
		int force_state = -1;
		uint32_t tmp;
		r = i + j;
		if (i < off[r]) force_state = 2;
		if (i > off_end[r]) force_state = 1;
		tmp = force_state < 0 ? p[(size_t)r * n_col + i - off[r]] : 0;
		if (state == 0) state = tmp & 7;
		else if (!(tmp >> (state + 2) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (force_state >= 0) state = force_state;
		if (state == 0) push_cigar(ez, MM_CIGAR_MATCH, 1), --i, --j;
		else if (state == 1 || state == 3) push_cigar(ez, MM_CIGAR_DEL, 1), --i;
		else push_cigar(ez, MM_CIGAR_INS, 1), --j;
	}
	if (i >= 0) push_cigar(ez, MM_CIGAR_DEL, i + 1);
	if (j >= 0) push_cigar(ez, MM_CIGAR_INS, j + 1);
	if (!is_rev)
		for (i = 0; i < ez->n_cigar >> 1; ++i) {
			uint32_t t = ez->cigar[i]; ez->cigar[i] = ez->cigar[ez->n_cigar - 1 - i]; ez->cigar[ez->n_cigar - 1 - i] = t;
		}
}

static int apply_zdrop(pgo_extz_t *ez, int32_t H, int r, int t, int zdrop, int8_t e) /* ksw2.h:168-184, is_rot=1 */
{
	if (H > (int32_t)ez->max) {
		ez->max = (uint32_t)H, ez->max_t = t, ez->max_q = r - t;
	} else if (t >= ez->max_t && r - t >= ez->max_q) {
		int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l;
		l = tl > ql ? tl - ql : ql - tl;
		if (zdrop >= 0 && (int32_t)ez->max - H > zdrop + l * e) { ez->zdropped = 1; return 1; }
	}
	return 0;
}

/* An OBSERVER, not part of the restatement: the product's kernels end an extension towards a block end early when alignment length alone proves
 * that the record is final (pangraph_amd/csrc/pga_dp.h, "length-bound stop": tlen <= 64, qlen >= w + 2 tlen, w >= 64).  The same rule is
 * evaluated here beside the reference's full sweep; when it says "final" the record is remembered and compared with the record the sweep ends
 * with.  pgo_lb_counters(): [0] problems in which the rule closed, [1] problems in which the full sweep still changed the record afterwards
 * (must stay 0: tests/test_oracle_cpu.py), [2] diagonals the sweeps ran after the rule had closed. */
static long long pgo_lb_cnt[3];
void pgo_lb_counters(long long out[3], int reset) { for (int i = 0; i < 3; ++i) { out[i] = __atomic_load_n(&pgo_lb_cnt[i], __ATOMIC_RELAXED); if (reset) __atomic_store_n(&pgo_lb_cnt[i], 0, __ATOMIC_RELAXED); } }
static int lb_gap(int q, int e, int q2, int e2, int L) { int a = q + e * L, b = q2 + e2 * L; return a < b ? a : b; }

void pgo_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
               int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, pgo_extz_t *ez)
{
	int lb_on = 0, lb_tail = 0, lb_hit = 0; pgo_extz_t lb_ez;
	int r, t, qe, qe_h = q + e /* ksw2_extd2_sse.c:73: taken BEFORE the (q,e)<->(q2,e2) swap and used for H at r==0 */, n_col, tlen16, qlen16, last_st, last_en, max_sc, min_sc, long_thres, long_diff;
	int with_cigar = !(flag & PGO_EZ_SCORE_ONLY), approx_max = !!(flag & PGO_EZ_APPROX_MAX);
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	int8_t *u, *v, *x, *y, *x2, *y2, *s, sc_mch, sc_mis, sc_N;
	uint8_t *sf, *qr, *p = 0;
	int *off = 0, *off_end = 0;

	ez_reset(ez);
	if (m <= 1 || qlen <= 0 || tlen <= 0) return;
	if (q2 + e2 < q + e) { t = q, q = q2, q2 = (int8_t)t, t = e, e = e2, e2 = (int8_t)t; }
	qe = q + e;
	sc_mch = mat[0], sc_mis = mat[1], sc_N = mat[m * m - 1] == 0 ? (int8_t)-e2 : mat[m * m - 1];
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	tlen16 = (tlen + 15) / 16 * 16;
	n_col = qlen < tlen ? qlen : tlen;
	n_col = (((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1) * 16;
	qlen16 = (qlen + 15) / 16 * 16;
	for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
		max_sc = max_sc > mat[t] ? max_sc : mat[t];
		min_sc = min_sc < mat[t] ? min_sc : mat[t];
	}
	if (-min_sc > 2 * (q + e)) return;
	long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	if (!approx_max && tlen <= 64 && w >= 64 && qlen >= w + 2 * tlen && sc_mch >= 0 && sc_mis <= sc_mch && sc_N <= sc_mch && q >= 0 && e >= 0 && q2 >= 0 && e2 >= 0) {
		lb_on = 1;
		lb_tail = tlen > 16 ? sc_mch * tlen - lb_gap(q, e, q2, e2, w + 32 - 2 * tlen) + (sc_mch + q + e) * (2 * tlen - 32) : INT32_MIN;
	}

	u = (int8_t*)malloc((size_t)tlen16 * 7);
	v = u + tlen16, x = v + tlen16, y = x + tlen16, x2 = y + tlen16, y2 = x2 + tlen16, s = y2 + tlen16;
	memset(u, -q - e, (size_t)tlen16 * 4);
	memset(x2, -q2 - e2, (size_t)tlen16 * 2);
	memset(s, 0, (size_t)tlen16);
	sf = (uint8_t*)calloc((size_t)tlen16 + 32, 1);
	qr = (uint8_t*)calloc((size_t)qlen16 + 48, 1);
	if (!approx_max) {
		H = (int32_t*)malloc((size_t)tlen16 * 4);
		for (t = 0; t < tlen16; ++t) H[t] = PGO_NEG_INF;
	}
	if (with_cigar) {
		p = (uint8_t*)malloc((size_t)(qlen + tlen - 1) * n_col + 16);
		off = (int*)malloc((size_t)(qlen + tlen - 1) * sizeof(int) * 2);
		off_end = off + qlen + tlen - 1;
	}
	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	memcpy(sf, target, (size_t)tlen);

	for (r = 0, last_st = last_en = -1; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0;
		int8_t x1, x21, v1;
		const uint8_t *qrr = qr + (qlen - 1 - r);
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
		if (en > (r + w) >> 1) en = (r + w) >> 1;
		if (st > en) { ez->zdropped = 1; break; }
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		if (st > 0) { /* ksw2_extd2_sse.c:146-158 */
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1];
			else x1 = w8(-q - e), x21 = w8(-q2 - e2), v1 = w8(-q - e);
		} else {
			x1 = w8(-q - e), x21 = w8(-q2 - e2);
			v1 = r == 0 ? w8(-q - e) : r < long_thres ? w8(-e) : r == long_thres ? w8(long_diff) : w8(-e2);
		}
		if (en >= r) {
			y[r] = w8(-q - e), y2[r] = w8(-q2 - e2);
			u[r] = r == 0 ? w8(-q - e) : r < long_thres ? w8(-e) : r == long_thres ? w8(long_diff) : w8(-e2);
		}
		/* score profile: written in 16-byte groups from st0 (ksw2_extd2_sse.c:165-185) */
		if (!(flag & PGO_EZ_GENERIC_SC)) {
			for (t = st0; t <= en0; t += 16)
				for (int l = 0; l < 16; ++l) {
					uint8_t a = sf[t + l], b = qrr[t + l];
					int8_t sc = a == b ? sc_mch : sc_mis;
					if (a == (uint8_t)(m - 1) || b == (uint8_t)(m - 1)) sc = sc_N;
					/* the vector store may run past s[tlen16): the reference owns that memory (sf follows s) and
					   never reads it back as a score, so it is simply dropped here */
					if (t + l < tlen16) s[t + l] = sc;
				}
		} else {
			for (t = st0; t <= en0; ++t) s[t] = mat[sf[t] * m + qrr[t]];
		}
		if (with_cigar) off[r] = st, off_end[r] = en;
		/* core loop over lanes st..en; x1/x21/v1 carry lane t-1 of the PREVIOUS diagonal */
		for (t = st; t <= en; ++t) {
			int8_t z = s[t], xt1 = x1, vt1 = v1, x2t1 = x21, ut = u[t], a, b, a2, b2, tmp;
			uint8_t d;
			x1 = x[t], v1 = v[t], x21 = x2[t];
			a = w8(xt1 + vt1), b = w8(y[t] + ut), a2 = w8(x2t1 + vt1), b2 = w8(y2[t] + ut);
			if (!(flag & PGO_EZ_RIGHT)) { /* ksw2_extd2_sse.c:238-258: ties keep the earlier state */
				d = 0;
				if (a > z) d = 1, z = a;
				if (b > z) d = 2, z = b;
				if (a2 > z) d = 3, z = a2;
				if (b2 > z) d = 4, z = b2;
			} else { /* ksw2_extd2_sse.c:285-305: ties move to the later state */
				d = z > a ? 0 : 1;  z = z > a ? z : a;
				d = z > b ? d : 2;  z = z > b ? z : b;
				d = z > a2 ? d : 3; z = z > a2 ? z : a2;
				d = z > b2 ? d : 4; z = z > b2 ? z : b2;
			}
			if (sc_mch < z) z = sc_mch;
			u[t] = w8(z - vt1), v[t] = w8(z - ut);
			tmp = w8(z - q);  a = w8(a - tmp),  b = w8(b - tmp);
			tmp = w8(z - q2); a2 = w8(a2 - tmp), b2 = w8(b2 - tmp);
			if (!(flag & PGO_EZ_RIGHT)) {
				x[t]  = w8((a  > 0 ? a  : 0) - qe);        if (a  > 0) d |= 0x08;
				y[t]  = w8((b  > 0 ? b  : 0) - qe);        if (b  > 0) d |= 0x10;
				x2[t] = w8((a2 > 0 ? a2 : 0) - (q2 + e2)); if (a2 > 0) d |= 0x20;
				y2[t] = w8((b2 > 0 ? b2 : 0) - (q2 + e2)); if (b2 > 0) d |= 0x40;
			} else {
				x[t]  = w8((0 > a  ? 0 : a)  - qe);        if (!(0 > a))  d |= 0x08;
				y[t]  = w8((0 > b  ? 0 : b)  - qe);        if (!(0 > b))  d |= 0x10;
				x2[t] = w8((0 > a2 ? 0 : a2) - (q2 + e2)); if (!(0 > a2)) d |= 0x20;
				y2[t] = w8((0 > b2 ? 0 : b2) - (q2 + e2)); if (!(0 > b2)) d |= 0x40;
			}
			if (with_cigar) p[(size_t)r * n_col + (t - st)] = d;
		}
		if (!approx_max) { /* ksw2_extd2_sse.c:322-366 */
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4, i;
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u[en0] : H[en0] + v[en0];
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4)
					for (i = 0; i < 4; ++i) {
						H[t + i] += (int32_t)v[t + i];
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
					}
				for (i = 0; i < 4; ++i)
					if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) {
					H[t] += (int32_t)v[t];
					if (H[t] > max_H) max_H = H[t], max_t = t;
				}
			} else H[0] = v[0] - qe_h, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte) ez->mte = H[en0], ez->mte_q = r - en0;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) ez->mqe = H[st0], ez->mqe_t = st0;
			if (apply_zdrop(ez, max_H, r, max_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
			if (lb_hit) __atomic_fetch_add(&pgo_lb_cnt[2], 1, __ATOMIC_RELAXED);
			if (lb_on && !lb_hit && (r & 7) == 7 && r >= 2 * tlen && r <= w + 30) {     /* the observer (see above) */
				int mm = (int32_t)ez->max < ez->mte ? (int32_t)ez->max : ez->mte;
				if (sc_mch * tlen - lb_gap(q, e, q2, e2, r + 3 - 2 * tlen) <= mm && lb_tail <= mm) lb_hit = 1, lb_ez = *ez;
			}
		} else { /* ksw2_extd2_sse.c:367-384: follow one cell per diagonal */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
					if (d0 > d1) H0 += d0; else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += v[last_H0_t];
				else ++last_H0_t, H0 += u[last_H0_t];
			} else H0 = v[0] - qe_h, last_H0_t = 0;
			if ((flag & PGO_EZ_APPROX_DROP) && apply_zdrop(ez, H0, r, last_H0_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	free(u); free(sf); free(qr); free(H);
	if (lb_hit) {
		__atomic_fetch_add(&pgo_lb_cnt[0], 1, __ATOMIC_RELAXED);
		if (!(ez->zdropped && ez->max == lb_ez.max && ez->max_t == lb_ez.max_t && ez->max_q == lb_ez.max_q && ez->mte == lb_ez.mte && ez->mte_q == lb_ez.mte_q &&
		      ez->mqe == lb_ez.mqe && ez->mqe_t == lb_ez.mqe_t && ez->score == lb_ez.score)) __atomic_fetch_add(&pgo_lb_cnt[1], 1, __ATOMIC_RELAXED);
	}
	if (with_cigar) { /* ksw2_extd2_sse.c:389-399 */
		int rev_cigar = !!(flag & PGO_EZ_REV_CIGAR);
		if (!ez->zdropped && !(flag & PGO_EZ_EXTZ_ONLY))
			backtrack(ez, rev_cigar, p, off, off_end, n_col, tlen - 1, qlen - 1);
		else if (!ez->zdropped && (flag & PGO_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > (int)ez->max) {
			ez->reach_end = 1;
			backtrack(ez, rev_cigar, p, off, off_end, n_col, ez->mqe_t, qlen - 1);
		} else if (ez->max_t >= 0 && ez->max_q >= 0)
			backtrack(ez, rev_cigar, p, off, off_end, n_col, ez->max_t, ez->max_q);
		free(p); free(off);
	}
}

/* ksw2_ll_sse.c:37-152.  Striped int16 local alignment: positions of the query padded to slen*8 are laid
 * out as vector j, lane l <-> position j + l*slen; padded positions score 0 against everything.  The
 * lazy-F loop, its early exit and the fact that E is derived from the pre-correction H are kept as is;
 * they decide which end coordinates tie-break to. */
static inline int16_t adds16(int a, int b) { int s = a + b; return (int16_t)(s > 32767 ? 32767 : s < -32768 ? -32768 : s); }
static inline int16_t subsu16(int16_t a, int16_t b) { uint16_t x = (uint16_t)a, y = (uint16_t)b; return (int16_t)(x > y ? x - y : 0); }
static inline int16_t max16(int16_t a, int16_t b) { return a > b ? a : b; }

int pgo_ll_i16(int qlen, const uint8_t *query, int m, const int8_t *mat, int tlen, const uint8_t *target,
               int gapo, int gape, int *qe, int *te)
{
	int slen = (qlen + 7) / 8, i, j, l, k, gmax = 0, qlen8 = slen * 8;
	int16_t *prof = (int16_t*)malloc((size_t)m * qlen8 * 2);
	int16_t *H0 = (int16_t*)calloc((size_t)qlen8 * 4, 2), *H1 = H0 + qlen8, *E = H1 + qlen8, *Hmax = E + qlen8;
	int16_t gapoe = (int16_t)(gapo + gape), ge = (int16_t)gape;
	for (int a = 0; a < m; ++a)
		for (j = 0; j < slen; ++j)
			for (l = 0; l < 8; ++l) {
				int pos = j + l * slen;
				prof[((size_t)a * slen + j) * 8 + l] = pos >= qlen ? 0 : mat[a * m + query[pos]];
			}
	*qe = *te = -1;
	for (i = 0; i < tlen; ++i) {
		int16_t f[8], h[8], mx[8], e[8];
		const int16_t *S = prof + (size_t)target[i] * slen * 8;
		int imax, done = 0;
		for (l = 0; l < 8; ++l) f[l] = 0, mx[l] = 0;
		h[0] = 0;
		for (l = 1; l < 8; ++l) h[l] = H0[(slen - 1) * 8 + l - 1]; /* shift the last vector up one lane */
		for (j = 0; j < slen; ++j) {
			for (l = 0; l < 8; ++l) {
				int16_t hh = adds16(h[l], S[j * 8 + l]);
				e[l] = E[j * 8 + l];
				hh = max16(hh, e[l]); hh = max16(hh, f[l]);
				mx[l] = max16(mx[l], hh);
				H1[j * 8 + l] = hh;
				hh = subsu16(hh, gapoe);
				e[l] = max16(subsu16(e[l], ge), hh);
				E[j * 8 + l] = e[l];
				f[l] = max16(subsu16(f[l], ge), hh);
				h[l] = H0[j * 8 + l];
			}
		}
		for (k = 0; k < 8 && !done; ++k) {
			for (l = 7; l > 0; --l) f[l] = f[l - 1];
			f[0] = 0;
			for (j = 0; j < slen; ++j) {
				int any = 0;
				for (l = 0; l < 8; ++l) {
					int16_t hh = max16(H1[j * 8 + l], f[l]);
					H1[j * 8 + l] = hh;
					hh = subsu16(hh, gapoe);
					f[l] = subsu16(f[l], ge);
					if (f[l] > hh) any = 1;
				}
				if (!any) { done = 1; break; }
			}
		}
		imax = 0;
		for (l = 0; l < 8; ++l) if (mx[l] > imax) imax = mx[l];
		/* __max_8 reduces with signed max over lanes; mx[] >= 0 so starting from 0 is the same */
		if (imax >= gmax) {
			gmax = imax, *te = i;
			memcpy(Hmax, H1, (size_t)qlen8 * 2);
		}
		{ int16_t *tmp = H1; H1 = H0; H0 = tmp; }
	}
	for (i = 0; i < qlen8; ++i)
		if ((int)(uint16_t)Hmax[i] == gmax) *qe = i / 8 + i % 8 * slen;
	{ /* H0/H1 may have been swapped: free the original block */
		int16_t *base = H0 < H1 ? H0 : H1;
		free(base);
	}
	free(prof);
	return gmax;
}
