/* pgo.h -- ORACLE (test infrastructure, never shipped, never linked into libpgalign.so).
 *
 * A plain-C, single-threaded, stage-separated CPU restatement of the block-alignment path that
 * `pangraph build` runs through minimap2.  Every function cites the reference file:line it
 * follows (paths relative to /root/reference/packages/minimap2-sys/minimap2/).  Parity of this
 * restatement is PINNED: tests/test_oracle_vs_ref.py checks it stage by stage and end to end against
 * oracle/_ref/libmm2ref.so (the reference's own C compiled by oracle/Makefile) and against the committed
 * golden vectors in tests/golden/ (incl. the reference's only known-answer test for this path,
 * packages/pangraph/src/align/minimap2_lib/align_with_minimap2_lib.rs:135-204).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load libpgoracle.so.
 */
#ifndef PGO_H
#define PGO_H

#include <stdint.h>
#include <stddef.h>
#include "../include/pga_mm2_abi.h"

typedef mm128_t pg128;

/* ---- sorting (pgo_sort.c) ---- */
void pgo_radix_sort_128x(pg128 *beg, pg128 *end);   /* ksort.h:101-151 + misc.c:155-156 */
void pgo_radix_sort_64(uint64_t *beg, uint64_t *end); /* misc.c:158-159 */

/* ---- sketch (pgo_sketch.c) ---- */
extern const uint8_t pgo_nt4[256];
uint64_t pgo_hash64(uint64_t key, uint64_t mask);
/* appends to *out (realloc'ed); returns new count */
size_t pgo_sketch(const char *seq, int len, int w, int k, uint32_t rid, pg128 **out, size_t n, size_t *cap);

/* ---- index (pgo_index.c) ---- */
typedef struct {
	mm_idx_t hdr;            /* ABI-visible header: must stay first */
	uint8_t *nt4;            /* 1 byte per base, all sequences concatenated at hdr.seq[i].offset */
	uint64_t n_keys;
	uint64_t *key;           /* distinct minimizer hashes, ascending */
	uint64_t *occ_off;       /* n_keys+1 */
	uint64_t *occ;           /* y values (rid<<32|pos<<1|strand), ascending within a key */
} pgo_index_t;

pgo_index_t *pgo_index_build(int w, int k, int bucket_bits, int n, const char **seq, const char **name);
void pgo_index_free(pgo_index_t *ix);
const uint64_t *pgo_index_get(const pgo_index_t *ix, uint64_t minier, int *n);
int32_t pgo_index_cal_max_occ(const pgo_index_t *ix, float f);

/* ---- seeding (pgo_seed.c) ---- */
typedef struct {
	uint32_t n, q_pos, q_span;
	uint8_t flt, is_tandem;
	const uint64_t *cr;
} pgo_seed_t;
size_t pgo_seed_mz_flt(pg128 *mv, size_t n, int32_t q_occ_max, float q_occ_frac);
pg128 *pgo_collect_anchors(const pgo_index_t *ix, const mm_mapopt_t *opt, const char *qname, int qlen,
                           const pg128 *mv, size_t n_mv, int64_t *n_a, int *rep_len);

/* ---- chaining (pgo_chain.c) ---- */
pg128 *pgo_lchain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc,
                      float chn_pen_gap, float chn_pen_skip, int64_t n, pg128 *a, int *n_u, uint64_t **u);

/* ---- DP (pgo_ksw.c) ---- */
typedef struct {
	uint32_t max; int zdropped;
	int max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end;
	int n_cigar, m_cigar;
	uint32_t *cigar;
} pgo_extz_t;
#define PGO_EZ_SCORE_ONLY 0x01
#define PGO_EZ_RIGHT      0x02
#define PGO_EZ_GENERIC_SC 0x04
#define PGO_EZ_APPROX_MAX 0x08
#define PGO_EZ_APPROX_DROP 0x10
#define PGO_EZ_EXTZ_ONLY  0x40
#define PGO_EZ_REV_CIGAR  0x80
#define PGO_NEG_INF (-0x40000000)
void pgo_lb_counters(long long out[3], int reset);   /* observer of the product's length-bound stop (pgo_ksw.c) */
void pgo_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
               int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, pgo_extz_t *ez);
int pgo_ll_i16(int qlen, const uint8_t *query, int m, const int8_t *mat, int tlen, const uint8_t *target,
               int gapo, int gape, int *qe, int *te);

/* ---- regions and base-level alignment (pgo_hit.c, pgo_align.c) ---- */
mm_reg1_t *pgo_gen_regs(uint32_t hash, int qlen, int n_u, uint64_t *u, pg128 *a);
void pgo_split_reg(mm_reg1_t *r, mm_reg1_t *r2, int n, int qlen, pg128 *a);
void pgo_filter_regs(const mm_mapopt_t *opt, int qlen, int *n_regs, mm_reg1_t *regs);
void pgo_hit_sort(int *n_regs, mm_reg1_t *r);
int pgo_squeeze_a(int n_regs, mm_reg1_t *regs, pg128 *a);
void pgo_set_mapq(int n_regs, mm_reg1_t *regs, int min_chain_sc, int match_sc, int rep_len);
void pgo_update_dp_max(int qlen, int n_regs, mm_reg1_t *regs, float frac, int a, int b);
mm_reg1_t *pgo_align_skeleton(const mm_mapopt_t *opt, const pgo_index_t *ix, int qlen, const char *qstr, int *n_regs, mm_reg1_t *regs, pg128 *a);

/* ---- whole query (pgo_map.c) ---- */
mm_reg1_t *pgo_map(const pgo_index_t *ix, int qlen, const char *seq, int *n_regs, const mm_mapopt_t *opt, const char *qname);

static inline float pgo_log2f_approx(float x) /* mmpriv.h:118-126 (bit trick; only valid for x>=2) */
{
	union { float f; uint32_t i; } z = { x };
	float r = (float)((z.i >> 23) & 255) - 128;
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	r += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return r;
}

#endif
