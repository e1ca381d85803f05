/* pgo_map.c -- ORACLE (test infrastructure only).
 *
 * One query through the whole path, restating mm_map_frag (map.c:227-374, single segment) and the option
 * presets (options.c:5-234); plus the minimap2-sys C-ABI (include/pga_mm2_abi.h) over the restatement so
 * that the same ctypes binding drives the reference build, this oracle and the HIP product.
 */
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include "pgo.h"

/* ---------------- options (options.c) ---------------- */
void mm_idxopt_init(mm_idxopt_t *opt) /* options.c:5-12 */
{
	memset(opt, 0, sizeof(*opt));
	opt->k = 15, opt->w = 10, opt->flag = 0;
	opt->bucket_bits = 14;
	opt->mini_batch_size = 50000000;
	opt->batch_size = 8000000000ULL;
}

void mm_mapopt_init(mm_mapopt_t *opt) /* options.c:14-64 */
{
	memset(opt, 0, sizeof(*opt));
	opt->seed = 11;
	opt->mid_occ_frac = 2e-4f;
	opt->min_mid_occ = 10, opt->max_mid_occ = 1000000;
	opt->sdust_thres = 0;
	opt->q_occ_frac = 0.01f;
	opt->min_cnt = 3, opt->min_chain_score = 40;
	opt->bw = 500, opt->bw_long = 20000;
	opt->max_gap = 5000, opt->max_gap_ref = -1;
	opt->max_chain_skip = 25, opt->max_chain_iter = 5000;
	opt->rmq_inner_dist = 1000, opt->rmq_size_cap = 100000;
	opt->rmq_rescue_size = 1000, opt->rmq_rescue_ratio = 0.1f;
	opt->chain_gap_scale = 0.8f, opt->chain_skip_scale = 0.0f;
	opt->max_max_occ = 4095, opt->occ_dist = 500;
	opt->mask_level = 0.5f, opt->mask_len = INT_MAX;
	opt->pri_ratio = 0.8f, opt->best_n = 5;
	opt->alt_drop = 0.15f;
	opt->a = 2, opt->b = 4, opt->q = 4, opt->e = 2, opt->q2 = 24, opt->e2 = 1;
	opt->sc_ambi = 1;
	opt->zdrop = 400, opt->zdrop_inv = 200;
	opt->end_bonus = -1;
	opt->min_dp_max = opt->min_chain_score * opt->a;
	opt->min_ksw_len = 200;
	opt->anchor_ext_len = 20, opt->anchor_ext_shift = 6;
	opt->max_clip_ratio = 1.0f;
	opt->mini_batch_size = 500000000;
	opt->max_sw_mat = 100000000;
	opt->cap_kalloc = 1000000000;
	opt->rank_min_len = 500, opt->rank_frac = 0.9f;
	opt->pe_ori = 0, opt->pe_bonus = 33;
}

int mm_set_opt(const char *preset, mm_idxopt_t *io, mm_mapopt_t *mo) /* options.c:88-162: only the presets pangraph can name */
{
	if (preset == 0) { mm_idxopt_init(io); mm_mapopt_init(mo); return 0; }
	if (strncmp(preset, "asm", 3) == 0) { /* options.c:115-130 */
		io->flag = 0, io->k = 19, io->w = 19;
		mo->bw = 1000, mo->bw_long = 100000;
		mo->max_gap = 10000;
		mo->flag |= MM_F_RMQ;
		mo->min_mid_occ = 50, mo->max_mid_occ = 500;
		mo->min_dp_max = 200;
		mo->best_n = 50;
		if (strcmp(preset, "asm5") == 0) mo->a = 1, mo->b = 19, mo->q = 39, mo->q2 = 81, mo->e = 3, mo->e2 = 1, mo->zdrop = mo->zdrop_inv = 200;
		else if (strcmp(preset, "asm10") == 0) mo->a = 1, mo->b = 9, mo->q = 16, mo->q2 = 41, mo->e = 2, mo->e2 = 1, mo->zdrop = mo->zdrop_inv = 200;
		else if (strcmp(preset, "asm20") == 0) mo->a = 1, mo->b = 4, mo->q = 6, mo->q2 = 26, mo->e = 2, mo->e2 = 1, mo->zdrop = mo->zdrop_inv = 200, io->w = 10;
		else return -1;
		return 0;
	}
	return -1; /* other minimap2 presets (map-ont, sr, splice, ...) are outside pangraph's path */
}

int mm_check_opt(const mm_idxopt_t *io, const mm_mapopt_t *mo) /* options.c:164-234 */
{
	if (mo->bw > mo->bw_long) return -8;
	if ((mo->flag & MM_F_RMQ) && (mo->flag & (MM_F_SR | MM_F_SPLICE))) return -7;
	if (io->k <= 0 || io->w <= 0) return -5;
	if (mo->best_n < 0) return -4;
	if (mo->pri_ratio < 0.0f || mo->pri_ratio > 1.0f) return -4;
	if ((mo->flag & MM_F_FOR_ONLY) && (mo->flag & MM_F_REV_ONLY)) return -3;
	if (mo->e <= 0 || mo->q <= 0) return -1;
	if ((mo->q != mo->q2 || mo->e != mo->e2) && !(mo->e > mo->e2 && mo->q + mo->e < mo->q2 + mo->e2)) return -2;
	if ((mo->q + mo->e) + (mo->q2 + mo->e2) > 127) return -1;
	if (mo->zdrop < mo->zdrop_inv) return -5;
	return 0;
}

void mm_mapopt_update(mm_mapopt_t *opt, const mm_idx_t *mi) /* options.c:66-80 */
{
	if (opt->mid_occ <= 0) {
		opt->mid_occ = pgo_index_cal_max_occ((const pgo_index_t*)mi, opt->mid_occ_frac);
		if (opt->mid_occ < opt->min_mid_occ) opt->mid_occ = opt->min_mid_occ;
		if (opt->max_mid_occ > opt->min_mid_occ && opt->mid_occ > opt->max_mid_occ) opt->mid_occ = opt->max_mid_occ;
	}
	if (opt->bw_long < opt->bw) opt->bw_long = opt->bw;
}

/* ---------------- one query (map.c:227-374) ---------------- */
static inline uint32_t x31_hash(const char *s) /* khash.h __ac_X31_hash_string */
{
	uint32_t h = (uint32_t)*s;
	if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)*s;
	return h;
}
static inline uint32_t wang_hash(uint32_t key) /* khash.h __ac_Wang_hash */
{
	key += ~(key << 15);
	key ^=  (key >> 10);
	key +=  (key << 3);
	key ^=  (key >> 6);
	key += ~(key << 11);
	key ^=  (key >> 16);
	return key;
}

mm_reg1_t *pgo_map(const pgo_index_t *ix, int qlen, const char *seq, int *n_regs, const mm_mapopt_t *opt, const char *qname)
{
	*n_regs = 0;
	if (qlen == 0) return 0;
	if (opt->max_qlen > 0 && qlen > opt->max_qlen) return 0;
	uint32_t hash = qname && !(opt->flag & MM_F_NO_HASH_NAME) ? x31_hash(qname) : 0;
	hash ^= wang_hash((uint32_t)qlen) + wang_hash((uint32_t)opt->seed);
	hash = wang_hash(hash);

	pg128 *mv = 0; size_t n_mv = 0, cap = 0;
	n_mv = pgo_sketch(seq, qlen, ix->hdr.w, ix->hdr.k, 0, &mv, 0, &cap);
	if (opt->q_occ_frac > 0.0f) n_mv = pgo_seed_mz_flt(mv, n_mv, opt->mid_occ, opt->q_occ_frac);
	int64_t n_a; int rep_len;
	pg128 *a = pgo_collect_anchors(ix, opt, qname, qlen, mv, n_mv, &n_a, &rep_len);
	free(mv);

	float chn_pen_gap = opt->chain_gap_scale * 0.01 * ix->hdr.k;   /* map.c:273: double product stored to float */
	float chn_pen_skip = opt->chain_skip_scale * 0.01 * ix->hdr.k;
	int n_regs0; uint64_t *u;
	a = pgo_lchain_rmq(opt->max_gap, opt->rmq_inner_dist, opt->bw, opt->max_chain_skip, opt->rmq_size_cap, opt->min_cnt,
	                   opt->min_chain_score, chn_pen_gap, chn_pen_skip, n_a, a, &n_regs0, &u);
	/* the long-join re-chaining (map.c:283-292) is disabled by MM_F_NO_LJOIN, which -X always sets */
	mm_reg1_t *regs0 = pgo_gen_regs(hash, qlen, n_regs0, u, a);
	/* chain_post is a no-op under MM_F_ALL_CHAINS; mm_est_err only sets `div`, which pangraph never reads */
	if (opt->flag & MM_F_CIGAR) {
		regs0 = pgo_align_skeleton(opt, ix, qlen, seq, &n_regs0, regs0, a);
	}
	regs0 = (mm_reg1_t*)realloc(regs0, sizeof(*regs0) * (size_t)n_regs0);
	pgo_set_mapq(n_regs0, regs0, opt->min_chain_score, opt->a, rep_len);
	free(a); free(u);
	*n_regs = n_regs0;
	if (n_regs0 == 0) { free(regs0); return 0; }
	return regs0;
}

/* ---------------- minimap2-sys C-ABI over the restatement ---------------- */
mm_idx_t *mm_idx_str(int w, int k, int is_hpc, int bucket_bits, int n, const char **seq, const char **name)
{
	if (is_hpc) return 0; /* homopolymer compression is outside pangraph's path */
	return (mm_idx_t*)pgo_index_build(w, k, bucket_bits, n, seq, name);
}
void mm_idx_destroy(mm_idx_t *mi) { pgo_index_free((pgo_index_t*)mi); }
struct mm_tbuf_s { int unused; };
mm_tbuf_t *mm_tbuf_init(void) { return (mm_tbuf_t*)calloc(1, sizeof(struct mm_tbuf_s)); }
void mm_tbuf_destroy(mm_tbuf_t *b) { free(b); }
mm_reg1_t *mm_map(const mm_idx_t *mi, int l_seq, const char *seq, int *n_regs, mm_tbuf_t *b, const mm_mapopt_t *opt, const char *name)
{
	(void)b;
	return pgo_map((const pgo_index_t*)mi, l_seq, seq, n_regs, opt, name);
}
double mm_event_identity(const mm_reg1_t *r) /* align.c:897-917 */
{
	int32_t n_gapo = 0, n_gap = 0;
	if (r->p == 0) return -1.0f;
	for (uint32_t i = 0; i < r->p->n_cigar; ++i) {
		int32_t op = r->p->cigar[i] & 0xf, len = (int32_t)(r->p->cigar[i] >> 4);
		if (op == MM_CIGAR_INS || op == MM_CIGAR_DEL) ++n_gapo, n_gap += len;
	}
	return (double)r->mlen / (r->blen + (int32_t)r->p->n_ambi - n_gap + n_gapo);
}
