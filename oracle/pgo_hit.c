/* pgo_hit.c -- ORACLE (test infrastructure only).
 *
 * Chain -> region bookkeeping, restating mm_gen_regs / mm_reg_set_coor / mm_cal_fuzzy_len / mm_split_reg /
 * mm_filter_regs / mm_hit_sort / mm_squeeze_a / mm_set_mapq / mm_set_inv_mapq (hit.c:8-123,188-218,
 * 290-329,396-466).  mm_set_parent / mm_select_sub are skipped by the reference under MM_F_ALL_CHAINS
 * (map.c:206-213,219-223), which pangraph always sets (-X).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include "pgo.h"

static void fuzzy_len(mm_reg1_t *r, const pg128 *a) /* hit.c:8-21 */
{
	r->mlen = r->blen = 0;
	if (r->cnt <= 0) return;
	r->mlen = r->blen = a[r->as].y >> 32 & 0xff;
	for (int i = r->as + 1; i < r->as + r->cnt; ++i) {
		int span = a[i].y >> 32 & 0xff;
		int tl = (int32_t)a[i].x - (int32_t)a[i-1].x;
		int ql = (int32_t)a[i].y - (int32_t)a[i-1].y;
		r->blen += tl > ql ? tl : ql;
		r->mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
	}
}

static void set_coor(mm_reg1_t *r, int32_t qlen, const pg128 *a) /* hit.c:23-38 (is_qstrand=0) */
{
	int32_t k = r->as, q_span = (int32_t)(a[k].y >> 32 & 0xff);
	r->rev = a[k].x >> 63;
	r->rid = (int32_t)(a[k].x << 1 >> 33);
	r->rs = (int32_t)a[k].x + 1 > q_span ? (int32_t)a[k].x + 1 - q_span : 0;
	r->re = (int32_t)a[k + r->cnt - 1].x + 1;
	if (!r->rev) {
		r->qs = (int32_t)a[k].y + 1 - q_span;
		r->qe = (int32_t)a[k + r->cnt - 1].y + 1;
	} else {
		r->qs = qlen - ((int32_t)a[k + r->cnt - 1].y + 1);
		r->qe = qlen - ((int32_t)a[k].y + 1 - q_span);
	}
	fuzzy_len(r, a);
}

static inline uint64_t mix64(uint64_t key) /* hit.c:40-50 (the unmasked variant of the sketch hash) */
{
	key = (~key + (key << 21));
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8));
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4));
	key = key ^ key >> 28;
	key = (key + (key << 31));
	return key;
}

mm_reg1_t *pgo_gen_regs(uint32_t hash, int qlen, int n_u, uint64_t *u, pg128 *a) /* hit.c:52-88 */
{
	if (n_u == 0) return 0;
	pg128 *z = (pg128*)malloc((size_t)n_u * sizeof(pg128));
	int i, k;
	for (i = k = 0; i < n_u; ++i) {
		uint32_t h = (uint32_t)mix64((mix64(a[k].x) + mix64(a[k].y)) ^ hash);
		z[i].x = u[i] ^ h;
		z[i].y = (uint64_t)k << 32 | (uint64_t)(int32_t)u[i];
		k += (int32_t)u[i];
	}
	pgo_radix_sort_128x(z, z + n_u);
	for (i = 0; i < n_u >> 1; ++i) { pg128 t = z[i]; z[i] = z[n_u-1-i]; z[n_u-1-i] = t; }
	mm_reg1_t *r = (mm_reg1_t*)calloc((size_t)n_u, sizeof(mm_reg1_t));
	for (i = 0; i < n_u; ++i) {
		mm_reg1_t *ri = &r[i];
		ri->id = i;
		ri->parent = -1; /* MM_PARENT_UNSET */
		ri->score = ri->score0 = (int32_t)(z[i].x >> 32);
		ri->hash = (uint32_t)z[i].x;
		ri->cnt = (int32_t)z[i].y;
		ri->as = (int32_t)(z[i].y >> 32);
		ri->div = -1.0f;
		set_coor(ri, qlen, a);
	}
	free(z);
	return r;
}

void pgo_split_reg(mm_reg1_t *r, mm_reg1_t *r2, int n, int qlen, pg128 *a) /* hit.c:106-123 */
{
	if (n <= 0 || n >= r->cnt) return;
	*r2 = *r;
	r2->id = -1;
	r2->sam_pri = 0;
	r2->p = 0;
	r2->split_inv = 0;
	r2->cnt = r->cnt - n;
	r2->score = (int32_t)(r->score * ((float)r2->cnt / r->cnt) + .499);
	r2->as = r->as + n;
	if (r->parent == r->id) r2->parent = -2; /* MM_PARENT_TMP_PRI */
	set_coor(r2, qlen, a);
	r->cnt -= r2->cnt;
	r->score -= r2->score;
	set_coor(r, qlen, a);
	r->split |= 1, r2->split |= 2;
}

void pgo_filter_regs(const mm_mapopt_t *opt, int qlen, int *n_regs, mm_reg1_t *regs) /* hit.c:290-309 */
{
	int i, k;
	for (i = k = 0; i < *n_regs; ++i) {
		mm_reg1_t *r = &regs[i];
		int flt = 0;
		if (!r->inv && !r->seg_split && r->cnt < opt->min_cnt) flt = 1;
		if (r->p) {
			if (r->mlen < opt->min_chain_score) flt = 1;
			else if (r->p->dp_max < opt->min_dp_max) flt = 1;
			else if (r->qs > qlen * opt->max_clip_ratio && qlen - r->qe > qlen * opt->max_clip_ratio) flt = 1;
			if (flt) free(r->p);
		}
		if (!flt) {
			if (k < i) regs[k++] = regs[i];
			else ++k;
		}
	}
	*n_regs = k;
}

void pgo_hit_sort(int *n_regs, mm_reg1_t *r) /* hit.c:188-218 (no ALT contigs) */
{
	int32_t i, n_aux, n = *n_regs;
	if (n <= 1) return;
	pg128 *aux = (pg128*)malloc((size_t)n * sizeof(pg128));
	mm_reg1_t *t = (mm_reg1_t*)malloc((size_t)n * sizeof(mm_reg1_t));
	for (i = n_aux = 0; i < n; ++i) {
		if (r[i].inv || r[i].cnt > 0) {
			int score = r[i].p ? r[i].p->dp_max : r[i].score;
			aux[n_aux].x = (uint64_t)score << 32 | r[i].hash;
			aux[n_aux++].y = (uint64_t)i;
		} else if (r[i].p) { free(r[i].p); r[i].p = 0; }
	}
	pgo_radix_sort_128x(aux, aux + n_aux);
	for (i = n_aux - 1; i >= 0; --i) t[n_aux - 1 - i] = r[aux[i].y];
	memcpy(r, t, sizeof(mm_reg1_t) * (size_t)n_aux);
	*n_regs = n_aux;
	free(aux); free(t);
}

int pgo_squeeze_a(int n_regs, mm_reg1_t *regs, pg128 *a) /* hit.c:311-329 */
{
	int i, as = 0;
	uint64_t *aux = (uint64_t*)malloc((size_t)(n_regs ? n_regs : 1) * 8);
	for (i = 0; i < n_regs; ++i) aux[i] = (uint64_t)regs[i].as << 32 | (uint64_t)i;
	pgo_radix_sort_64(aux, aux + n_regs);
	for (i = 0; i < n_regs; ++i) {
		mm_reg1_t *r = &regs[(int32_t)aux[i]];
		if (r->as != as) {
			memmove(&a[as], &a[r->as], (size_t)r->cnt * 16);
			r->as = as;
		}
		as += r->cnt;
	}
	free(aux);
	return as;
}

static void set_inv_mapq(int n_regs, mm_reg1_t *regs) /* hit.c:396-419 */
{
	int i, n_aux;
	if (n_regs < 3) return;
	for (i = 0; i < n_regs; ++i) if (regs[i].inv) break;
	if (i == n_regs) return;
	pg128 *aux = (pg128*)malloc((size_t)n_regs * 16);
	for (i = n_aux = 0; i < n_regs; ++i)
		if (regs[i].parent == i || regs[i].parent < 0)
			aux[n_aux].y = (uint64_t)i, aux[n_aux++].x = (uint64_t)regs[i].rid << 32 | (uint64_t)regs[i].rs;
	pgo_radix_sort_128x(aux, aux + n_aux);
	for (i = 1; i < n_aux - 1; ++i) {
		mm_reg1_t *inv = &regs[aux[i].y];
		if (inv->inv) {
			mm_reg1_t *l = &regs[aux[i-1].y], *r = &regs[aux[i+1].y];
			inv->mapq = l->mapq < r->mapq ? l->mapq : r->mapq;
		}
	}
	free(aux);
}

void pgo_set_mapq(int n_regs, mm_reg1_t *regs, int min_chain_sc, int match_sc, int rep_len) /* hit.c:421-466, is_sr=0 */
{
	static const float q_coef = 40.0f;
	int64_t sum_sc = 0;
	float uniq_ratio;
	int i;
	if (n_regs == 0) return;
	for (i = 0; i < n_regs; ++i)
		if (regs[i].parent == regs[i].id) sum_sc += regs[i].score;
	uniq_ratio = (float)sum_sc / (sum_sc + rep_len);
	for (i = 0; i < n_regs; ++i) {
		mm_reg1_t *r = &regs[i];
		if (r->inv) r->mapq = 0;
		else if (r->parent == r->id) {
			int mapq, subsc;
			float pen_s1 = (r->score > 100 ? 1.0f : 0.01f * r->score) * uniq_ratio;
			float pen_cm = r->cnt > 10 ? 1.0f : 0.1f * r->cnt;
			pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
			subsc = r->subsc > min_chain_sc ? r->subsc : min_chain_sc;
			if (r->p && r->p->dp_max2 > 0 && r->p->dp_max > 0) {
				float identity = (float)r->mlen / r->blen;
				float x = (float)r->p->dp_max2 * subsc / r->p->dp_max / r->score0;
				mapq = (int)(identity * pen_cm * q_coef * (1.0f - x * x) * logf((float)r->p->dp_max / match_sc));
				int mapq_alt = (int)(6.02f * identity * identity * (r->p->dp_max - r->p->dp_max2) / match_sc + .499f);
				mapq = mapq < mapq_alt ? mapq : mapq_alt;
			} else {
				float x = (float)subsc / r->score0;
				if (r->p) {
					float identity = (float)r->mlen / r->blen;
					mapq = (int)(identity * pen_cm * q_coef * (1.0f - x) * logf((float)r->p->dp_max / match_sc));
				} else mapq = (int)(pen_cm * q_coef * (1.0f - x) * logf(r->score));
			}
			mapq -= (int)(4.343f * logf(r->n_sub + 1) + .499f);
			mapq = mapq > 0 ? mapq : 0;
			r->mapq = mapq < 60 ? mapq : 60;
			if (r->p && r->p->dp_max > r->p->dp_max2 && r->mapq == 0) r->mapq = 1;
		} else r->mapq = 0;
	}
	set_inv_mapq(n_regs, regs);
}
