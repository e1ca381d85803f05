/* pga_mm2_abi.h -- the drop-in boundary, part 1: the minimap2-sys C-ABI.
 *
 * libpgalign.so exports exactly the C symbols the reference's Rust crate `minimap2` calls through
 * `minimap2-sys` (bindgen over packages/minimap2-sys/minimap2.h), with identical struct layouts, so
 * the unchanged crate links against it.  Each declaration cites the reference
 * interface it replaces (paths relative to /root/reference/packages/minimap2-sys/minimap2/).
 *
 * Layout contract (checked by tests/test_abi.py): sizeof(mm_idxopt_t)=24, sizeof(mm_mapopt_t)=248,
 * sizeof(mm_reg1_t)=80, sizeof(mm_extra_t)=24 (+4 B per CIGAR op), sizeof(mm_idx_t)=80,
 * sizeof(mm_idx_seq_t)=24.
 */
#ifndef PGA_MM2_ABI_H
#define PGA_MM2_ABI_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* minimap.h:10-46 -- the flag bits this path honours */
#define MM_F_NO_DIAG     (0x001LL)
#define MM_F_NO_DUAL     (0x002LL)
#define MM_F_CIGAR       (0x004LL)
#define MM_F_OUT_CG      (0x020LL)
#define MM_F_SPLICE      (0x080LL)
#define MM_F_SPLICE_FOR  (0x100LL)
#define MM_F_SPLICE_REV  (0x200LL)
#define MM_F_NO_LJOIN    (0x400LL)
#define MM_F_SR          (0x1000LL)
#define MM_F_FOR_ONLY    (0x100000LL)
#define MM_F_REV_ONLY    (0x200000LL)
#define MM_F_HEAP_SORT   (0x400000LL)
#define MM_F_ALL_CHAINS  (0x800000LL)
#define MM_F_EQX         (0x4000000LL)
#define MM_F_NO_END_FLT  (0x10000000LL)
#define MM_F_RMQ         (0x80000000LL)
#define MM_F_QSTRAND     (0x100000000LL)
#define MM_F_NO_INV      (0x200000000LL)
#define MM_F_NO_HASH_NAME (0x400000000LL)
#define MM_I_HPC         0x1
#define MM_I_NO_NAME     0x4

#define MM_CIGAR_MATCH 0
#define MM_CIGAR_INS   1
#define MM_CIGAR_DEL   2
#define MM_CIGAR_STR  "MIDNSHP=XB"   /* minimap.h:62 */

/* minimap.h:69-70 */
typedef struct { uint64_t x, y; } mm128_t;

/* minimap.h:73-78 */
typedef struct {
	char *name;
	uint64_t offset;
	uint32_t len;
	uint32_t is_alt;
} mm_idx_seq_t;

/* minimap.h:80-91.  The Rust side reads n_seq, seq[i].name, seq[i].len (packages/minimap2/src/map.rs:278-289);
 * B/I/km/h are opaque -- this library keeps its own device-resident index behind `h`. */
typedef struct {
	int32_t b, w, k, flag;
	uint32_t n_seq;
	int32_t index;
	int32_t n_alt;
	mm_idx_seq_t *seq;
	uint32_t *S;
	struct mm_idx_bucket_s *B;
	struct mm_idx_intv_s *I;
	void *km, *h;
} mm_idx_t;

/* minimap.h:94-100 */
typedef struct {
	uint32_t capacity;
	int32_t dp_score, dp_max, dp_max2;
	uint32_t n_ambi:30, trans_strand:2;
	uint32_t n_cigar;
	uint32_t cigar[];
} mm_extra_t;

/* minimap.h:102-119 */
typedef struct {
	int32_t id;
	int32_t cnt;
	int32_t rid;
	int32_t score;
	int32_t qs, qe, rs, re;
	int32_t parent, subsc;
	int32_t as;
	int32_t mlen, blen;
	int32_t n_sub;
	int32_t score0;
	uint32_t mapq:8, split:2, rev:1, inv:1, sam_pri:1, proper_frag:1, pe_thru:1, seg_split:1, seg_id:8, split_inv:1, is_alt:1, strand_retained:1, dummy:5;
	uint32_t hash;
	float div;
	mm_extra_t *p;
} mm_reg1_t;

/* minimap.h:122-126 */
typedef struct {
	short k, w, flag, bucket_bits;
	int64_t mini_batch_size;
	uint64_t batch_size;
} mm_idxopt_t;

/* minimap.h:128-181 */
typedef struct {
	int64_t flag;
	int seed;
	int sdust_thres;
	int max_qlen;
	int bw, bw_long;
	int max_gap, max_gap_ref;
	int max_frag_len;
	int max_chain_skip, max_chain_iter;
	int min_cnt;
	int min_chain_score;
	float chain_gap_scale;
	float chain_skip_scale;
	int rmq_size_cap, rmq_inner_dist;
	int rmq_rescue_size;
	float rmq_rescue_ratio;
	float mask_level;
	int mask_len;
	float pri_ratio;
	int best_n;
	float alt_drop;
	int a, b, q, e, q2, e2;
	int sc_ambi;
	int noncan;
	int junc_bonus;
	int zdrop, zdrop_inv;
	int end_bonus;
	int min_dp_max;
	int min_ksw_len;
	int anchor_ext_len, anchor_ext_shift;
	float max_clip_ratio;
	int rank_min_len;
	float rank_frac;
	int pe_ori, pe_bonus;
	float mid_occ_frac;
	float q_occ_frac;
	int32_t min_mid_occ, max_mid_occ;
	int32_t mid_occ;
	int32_t max_occ, max_max_occ, occ_dist;
	int64_t mini_batch_size;
	int64_t max_sw_mat;
	int64_t cap_kalloc;
	const char *split_prefix;
} mm_mapopt_t;

/* minimap.h:196-201: one per calling thread, never shared (packages/minimap2/src/buf.rs:6) */
typedef struct mm_tbuf_s mm_tbuf_t;

/* options.c:5-12 / :14-64 -- used by the crate's Default impls (packages/minimap2-sys/src/lib.rs:13-31) */
void mm_idxopt_init(mm_idxopt_t *opt);
void mm_mapopt_init(mm_mapopt_t *opt);
/* minimap.h:216, options.c:88-162: 0 on success, -1 on an unknown preset; preset==NULL sets defaults */
int mm_set_opt(const char *preset, mm_idxopt_t *io, mm_mapopt_t *mo);
/* minimap.h:217, options.c:164-234: 0 ok, negative error code */
int mm_check_opt(const mm_idxopt_t *io, const mm_mapopt_t *mo);
/* minimap.h:229, options.c:66-80: sets mo->mid_occ from the index */
void mm_mapopt_update(mm_mapopt_t *opt, const mm_idx_t *mi);
/* minimap.h:313, index.c:408-456: NUL-terminated ASCII sequences and names; returns NULL when n<=0;
 * inputs are copied; the result is released by mm_idx_destroy() */
mm_idx_t *mm_idx_str(int w, int k, int is_hpc, int bucket_bits, int n, const char **seq, const char **name);
/* minimap.h:327, index.c:57-82 */
void mm_idx_destroy(mm_idx_t *mi);
/* minimap.h:339,346, map.c:13-26 */
mm_tbuf_t *mm_tbuf_init(void);
void mm_tbuf_destroy(mm_tbuf_t *b);
/* minimap.h:368, map.c:376-381: returns a malloc()ed array of *n_regs records (NULL when none); every
 * reg.p is separately malloc()ed; the CALLER frees both with libc free() (packages/minimap2/src/map.rs:407-420) */
mm_reg1_t *mm_map(const mm_idx_t *mi, int l_seq, const char *seq, int *n_regs, mm_tbuf_t *b, const mm_mapopt_t *opt, const char *name);
/* mmpriv.h:67, align.c:911-917 */
double mm_event_identity(const mm_reg1_t *r);

#ifdef __cplusplus
}
#endif
#endif
