/* pgo_mapvar.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the re-alignment of a block's member sequences onto the merged
 * consensus.
 *
 *   map_variations                                   packages/pangraph/src/align/map_variations.rs:39-77
 *   align_with_nextclade                             packages/pangraph/src/align/nextclade/align_with_nextclade.rs:24-75
 *   align_nuc_simplestripe (band retry loop)         packages/pangraph/src/align/nextclade/align/align.rs:32-71
 *   simple_stripes, Band2d                           packages/pangraph/src/align/nextclade/align/band_2d.rs:36-57,72-131
 *   score_matrix                                     packages/pangraph/src/align/nextclade/align/score_matrix.rs:23-199
 *   backtrace                                        packages/pangraph/src/align/nextclade/align/backtrace.rs:17-85
 *   insertions_strip                                 packages/pangraph/src/align/nextclade/align/insertions_strip.rs:47-97
 *   find_nuc_changes                                 packages/pangraph/src/align/nextclade/analyze/nuc_changes.rs:18-71
 *   to_nuc, Nuc, lookup_nuc_scoring_matrix           packages/pangraph/src/align/nextclade/alphabet/nuc.rs:10-30,99-121,
 *                                                    packages/pangraph/src/align/nextclade/align/score_matrix_nuc.rs:6-30
 *   get_gap_open_close_scores_flat                   packages/pangraph/src/align/nextclade/align/gap_open.rs:6-10
 *   the caller: MergePromise::solve_promise          packages/pangraph/src/pangraph/reweave.rs:40-94
 *
 * The reference is Rust and cannot be built in this image; this file is pinned by the known-answer vectors of the reference's unit
 * tests (align_with_nextclade.rs:92-311, map_variations.rs:190-365, align.rs:191-250), checked in tests/test_mapvar_cpu.py.
 * Everything is integer arithmetic; there is no order-of-summation question on this path.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
	int32_t score_match, penalty_mismatch, penalty_gap_open, penalty_gap_extend;     /* params.rs:142-170: 3, 1, 6, 0 */
	int32_t left_terminal_gaps_free, right_terminal_gaps_free, gap_align_left;       /* true, true, Left */
	int32_t min_length, max_alignment_attempts, extra_band_width;                    /* map_variations.rs:45-52: 1, args (4), args (5) */
} pgo_mapvar_params_t;

typedef struct { uint32_t pos, alt; } pgo_sub_t;              /* alt: the query's letter (ASCII) */
typedef struct { uint32_t pos, len; } pgo_del_t;
typedef struct { uint32_t pos, len; uint64_t seq_off; } pgo_ins_t;   /* pos already in pangraph's convention (map_variations.rs:71-74: + 1) */

typedef struct {
	int32_t status;           /* 0 ok; 1 query shorter than min_length (align.rs:42-46); 2 letter to_nuc rejects (nuc.rs:99-121) */
	int32_t score, attempts, hit_boundary;
	uint32_t n_subs, n_dels, n_inss, n_ins_bases;
} pgo_mapvar_res_t;

enum { P_MATCH = 1, P_REF_GAP_MATRIX = 2, P_QRY_GAP_MATRIX = 4, P_REF_GAP_EXTEND = 8, P_QRY_GAP_EXTEND = 16, P_BOUNDARY = 32 };   /* score_matrix.rs:6-11 */
#define NO_ALIGN (-1000000000)                                                                                                     /* :13 */
#define NUC_N 14
#define NUC_GAP 15

static int to_nuc(char c)                                       /* nuc.rs:10-30 (enum order), :99-121 */
{
	static const char order[] = "TAWCYMHGKRDSBVN-";
	for (int i = 0; i < 16; ++i) if (order[i] == c) return i;
	return -1;
}
static const char nuc_chars[] = "TAWCYMHGKRDSBVN-";

/* score_matrix_nuc.rs:6-26: the letters T A W C Y M H G K R D S B V N are the 4-bit sets 1..15 over {T, A, C, G} and two of them match
 * when the sets intersect; the gap letter matches only N and itself (the 256 entries are pinned by tests/golden/nuc_matrix.json) */
static int nuc_match(int x, int y) { return (x == NUC_GAP || y == NUC_GAP) ? (x >= NUC_N && y >= NUC_N) : (((x + 1) & (y + 1)) != 0); }

int pgo_nuc_match(int x, int y) { return nuc_match(x, y); }
int pgo_to_nuc(int c) { return to_nuc((char)c); }

static int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : v > hi ? hi : v; }

/* band_2d.rs:36-57 */
static void simple_stripes(int mean_shift, int64_t band_width, int ref_len, int qry_len, int32_t *begin, int32_t *end)
{
	for (int i = 0; i <= ref_len; ++i) {
		begin[i] = (int32_t)clampi(-(int64_t)mean_shift - band_width + i, 0, qry_len);
		end[i] = (int32_t)clampi(-(int64_t)mean_shift + band_width + i + 1, 1, (int64_t)qry_len + 1);
	}
	begin[0] = 0;
	end[ref_len] = qry_len + 1;
}

/* one attempt: score_matrix + backtrace; aln_qry/aln_ref receive the alignment (letters 0..15), returns its length */
static int64_t align_pairwise(const uint8_t *qry, int qry_len, const uint8_t *ref, int ref_len, const pgo_mapvar_params_t *P,
                              const int32_t *sb, const int32_t *se, uint8_t *aln_qry, uint8_t *aln_ref, int *score_out, int *hit_out)
{
	const int n_rows = ref_len + 1, n_cols = qry_len + 1;
	int64_t *row0 = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_rows + 1));
	row0[0] = 0;
	for (int i = 0; i < n_rows; ++i) row0[i + 1] = row0[i] + (se[i] - sb[i]);                 /* band_2d.rs:160-171 */
	int32_t *scores = (int32_t*)calloc((size_t)row0[n_rows], sizeof(int32_t));
	int8_t *paths = (int8_t*)calloc((size_t)row0[n_rows], 1);
#define S(r, q) scores[row0[r] + ((q) - sb[r])]
#define Pth(r, q) paths[row0[r] + ((q) - sb[r])]
	const int left_align = P->gap_align_left ? 1 : 0;                                          /* score_matrix.rs:44-47 */
	const int gap_open = P->penalty_gap_open, ext = P->penalty_gap_extend;                     /* gap_open_close is flat: gap_open.rs:6-10 */
	Pth(0, 0) = 0; S(0, 0) = 0;
	for (int qpos = sb[0] + 1; qpos < se[0]; ++qpos) {                                         /* :63-75 */
		Pth(0, qpos) = P_REF_GAP_EXTEND + P_REF_GAP_MATRIX;
		if (P->left_terminal_gaps_free) S(0, qpos) = 0;
		else if (qpos == 1) S(0, 1) = -gap_open;
		else S(0, qpos) = S(0, qpos - 1) - ext;
	}
	int32_t *qry_gaps = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_cols);
	for (int i = 0; i < n_cols; ++i) qry_gaps[i] = NO_ALIGN;
	for (int ri = 1; ri <= ref_len; ++ri) {                                                    /* :83-198 */
		int32_t ref_gaps = NO_ALIGN;
		for (int qpos = sb[ri]; qpos < se[ri]; ++qpos) {
			int tmp_path = 0, origin = 0;
			int32_t score = NO_ALIGN, tmp_score;
			if (qpos == 0) {
				tmp_path = P_QRY_GAP_EXTEND;
				origin = P_QRY_GAP_MATRIX;
				if (P->left_terminal_gaps_free) score = 0;
				else if (ri == 1) score = -gap_open;
				else score = S(ri - 1, 0) - ext;
			} else {
				if (qpos > sb[ri - 1] && qpos - 1 < se[ri - 1]) {
					const int q = qry[qpos - 1], r = ref[ri - 1];
					if (q == NUC_N || r == NUC_N) score = S(ri - 1, qpos - 1) + P->score_match - 1;
					else if (nuc_match(q, r)) score = S(ri - 1, qpos - 1) + P->score_match;
					else score = S(ri - 1, qpos - 1) - P->penalty_mismatch;
					origin = P_MATCH;
				} else if (ri < ref_len && qpos < qry_len) tmp_path |= P_BOUNDARY;
				if (qpos > sb[ri]) {
					int32_t r_gap_extend, r_gap_open;
					if (ri != ref_len || !P->right_terminal_gaps_free) { r_gap_extend = ref_gaps - ext; r_gap_open = S(ri, qpos - 1) - gap_open; }
					else { r_gap_extend = ref_gaps; r_gap_open = S(ri, qpos - 1); }
					if (r_gap_extend >= r_gap_open && qpos > sb[ri] + 1) { tmp_score = r_gap_extend; tmp_path += P_REF_GAP_EXTEND; }
					else tmp_score = r_gap_open;
					ref_gaps = tmp_score;
					if (score - left_align < tmp_score) { score = tmp_score; origin = P_REF_GAP_MATRIX; }
				} else if (ri < n_rows - 1 && qpos < qry_len) tmp_path |= P_BOUNDARY;
				if (qpos < se[ri - 1]) {
					int32_t q_gap_extend, q_gap_open;
					if (qpos != qry_len || !P->right_terminal_gaps_free) { q_gap_extend = qry_gaps[qpos] - ext; q_gap_open = S(ri - 1, qpos) - gap_open; }
					else { q_gap_extend = qry_gaps[qpos]; q_gap_open = S(ri - 1, qpos); }
					/* (:175: stripes[ri - 2] is only evaluated when the comparison in front of it holds; in row 1 qry_gaps is NO_ALIGN) */
					if (q_gap_extend >= q_gap_open && ri >= 2 && qpos < se[ri - 2]) { tmp_score = q_gap_extend; tmp_path += P_QRY_GAP_EXTEND; }
					else tmp_score = q_gap_open;
					qry_gaps[qpos] = tmp_score;
					if (score - left_align < tmp_score) { score = tmp_score; origin = P_QRY_GAP_MATRIX; }
				} else if (qpos < n_cols - 1 && ri < ref_len) { qry_gaps[qpos] = NO_ALIGN; tmp_path |= P_BOUNDARY; }
			}
			tmp_path += origin;
			Pth(ri, qpos) = (int8_t)tmp_path;
			S(ri, qpos) = score;
		}
	}
	/* backtrace.rs:17-85; num_cols = the largest stripe end = qry_len + 1 (band_2d.rs:160-171) */
	int r_pos = n_rows - 1, q_pos = n_cols - 1, current_matrix = 0, hit = 0;
	int64_t n = 0;
	int bad = 0;
	while (r_pos > 0 || q_pos > 0) {
		if (q_pos < sb[r_pos] || q_pos >= se[r_pos]) { bad = 1; break; }                       /* the reference would panic (band_2d.rs:118-124) */
		const int origin = Pth(r_pos, q_pos);
		if (origin & P_BOUNDARY) hit = 1;
		if ((origin & P_MATCH) && current_matrix == 0) {
			--q_pos; --r_pos;
			aln_qry[n] = qry[q_pos]; aln_ref[n] = ref[r_pos]; ++n;
		} else if (((origin & P_REF_GAP_MATRIX) && current_matrix == 0) || current_matrix == P_REF_GAP_MATRIX) {
			--q_pos;
			aln_qry[n] = qry[q_pos]; aln_ref[n] = NUC_GAP; ++n;
			current_matrix = (origin & P_REF_GAP_EXTEND) ? P_REF_GAP_MATRIX : 0;
		} else if (((origin & P_QRY_GAP_MATRIX) && current_matrix == 0) || current_matrix == P_QRY_GAP_MATRIX) {
			aln_qry[n] = NUC_GAP;
			--r_pos;
			aln_ref[n] = ref[r_pos]; ++n;
			current_matrix = (origin & P_QRY_GAP_EXTEND) ? P_QRY_GAP_MATRIX : 0;
		} else { bad = 1; break; }                                                             /* unreachable!() in the reference */
	}
	for (int64_t i = 0; i < n / 2; ++i) {
		uint8_t t = aln_qry[i]; aln_qry[i] = aln_qry[n - 1 - i]; aln_qry[n - 1 - i] = t;
		t = aln_ref[i]; aln_ref[i] = aln_ref[n - 1 - i]; aln_ref[n - 1 - i] = t;
	}
	*score_out = S(n_rows - 1, n_cols - 1);
	*hit_out = hit;
#undef S
#undef Pth
	free(row0); free(scores); free(paths); free(qry_gaps);
	return bad ? -1 : n;
}

/* map_variations of one pair.  Output arrays must hold ref_len + qry_len + 2 entries each (ins_seq: qry_len bytes).
 * aln_out (may be NULL): two rows of ref_len + qry_len bytes, the gapped alignment as letters (stage tap), *aln_len its length. */
int pgo_map_variations(const char *ref_s, int ref_len, const char *qry_s, int qry_len, int mean_shift, int band_width,
                       const pgo_mapvar_params_t *P, pgo_mapvar_res_t *res, pgo_sub_t *subs, pgo_del_t *dels, pgo_ins_t *inss,
                       char *ins_seq, char *aln_out, int64_t *aln_len)
{
	memset(res, 0, sizeof(*res));
	uint8_t *ref = (uint8_t*)malloc((size_t)ref_len + 1), *qry = (uint8_t*)malloc((size_t)qry_len + 1);
	int status = 0;
	for (int i = 0; i < ref_len && !status; ++i) { const int c = to_nuc(ref_s[i]); if (c < 0) status = 2; else ref[i] = (uint8_t)c; }   /* align_with_nextclade.rs:30-31 */
	for (int i = 0; i < qry_len && !status; ++i) { const int c = to_nuc(qry_s[i]); if (c < 0) status = 2; else qry[i] = (uint8_t)c; }
	if (!status && qry_len < P->min_length) status = 1;                                        /* align.rs:42-46 */
	if (status) { free(ref); free(qry); res->status = status; return status; }

	int64_t bw = (int64_t)band_width + P->extra_band_width;                                     /* map_variations.rs:51 */
	int32_t *sb = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ref_len + 1)), *se = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ref_len + 1));
	uint8_t *aq = (uint8_t*)malloc((size_t)ref_len + qry_len + 2), *ar = (uint8_t*)malloc((size_t)ref_len + qry_len + 2);
	int attempt = 1, score = 0, hit = 0;
	simple_stripes(mean_shift, bw, ref_len, qry_len, sb, se);
	int64_t n = align_pairwise(qry, qry_len, ref, ref_len, P, sb, se, aq, ar, &score, &hit);
	while (n >= 0 && hit && attempt < P->max_alignment_attempts) {                             /* align.rs:55-62 */
		const int64_t a = mean_shift < 0 ? -(int64_t)mean_shift : mean_shift;
		const int64_t m1 = a > 1 ? a : 1;
		bw = 2 * bw > m1 ? 2 * bw : m1;
		simple_stripes(mean_shift, bw, ref_len, qry_len, sb, se);
		++attempt;
		n = align_pairwise(qry, qry_len, ref, ref_len, P, sb, se, aq, ar, &score, &hit);
	}
	free(sb); free(se);
	if (n < 0) { free(ref); free(qry); free(aq); free(ar); res->status = 3; return 3; }
	res->score = score; res->attempts = attempt; res->hit_boundary = hit;
	if (aln_out) { for (int64_t i = 0; i < n; ++i) { aln_out[i] = nuc_chars[aq[i]]; aln_out[(int64_t)ref_len + qry_len + i] = nuc_chars[ar[i]]; } *aln_len = n; }

	/* insertions_strip.rs:47-97: the query with the columns of reference gaps removed, those columns as insertions */
	uint8_t *stripped = (uint8_t*)malloc((size_t)ref_len + qry_len + 2);
	int64_t ns = 0; uint32_t n_inss = 0, n_ib = 0;
	{
		int32_t insertion_start = -1, ref_pos = -1; uint32_t cur = 0;
		for (int64_t i = 0; i < n; ++i) {
			if (ar[i] == NUC_GAP) {
				if (cur == 0) { insertion_start = ref_pos; inss[n_inss].seq_off = n_ib; }
				ins_seq[n_ib++] = nuc_chars[aq[i]]; ++cur;
			} else {
				stripped[ns++] = aq[i];
				++ref_pos;
				if (cur) { inss[n_inss].pos = (uint32_t)(insertion_start + 1); inss[n_inss].len = cur; ++n_inss; cur = 0; insertion_start = -1; }
			}
		}
		if (cur) { inss[n_inss].pos = (uint32_t)(insertion_start + 1); inss[n_inss].len = cur; ++n_inss; }
		/* (:91 sort by (pos, len): positions are strictly increasing, nothing to do) */
	}
	/* nuc_changes.rs:18-71 against the ungapped reference (align_with_nextclade.rs:40-44) */
	uint32_t n_subs = 0, n_dels = 0;
	int64_t n_del = 0, del_pos = -1, alignment_start = -1, alignment_end = -1; int before_alignment = 1;
	for (int64_t i = 0; i < ns; ++i) {
		const int d = stripped[i];
		if (d != NUC_GAP) {
			if (before_alignment) { alignment_start = i; before_alignment = 0; }
			else if (n_del > 0) { dels[n_dels].pos = (uint32_t)del_pos; dels[n_dels].len = (uint32_t)n_del; ++n_dels; n_del = 0; }
			alignment_end = i + 1;
		}
		const int r = ref[i];
		if (d != NUC_GAP && d != r) { subs[n_subs].pos = (uint32_t)i; subs[n_subs].alt = (uint32_t)nuc_chars[d]; ++n_subs; }
		else if (d == NUC_GAP && !before_alignment) { if (n_del == 0) del_pos = i; ++n_del; }
	}
	/* align_with_nextclade.rs:46-64: leading and trailing gaps become deletions, pushed BEHIND the sorted internal ones */
	if (alignment_start >= 0 && alignment_end >= 0) {
		if (alignment_start > 0) { dels[n_dels].pos = 0; dels[n_dels].len = (uint32_t)alignment_start; ++n_dels; }
		if (alignment_end < ref_len) { dels[n_dels].pos = (uint32_t)alignment_end; dels[n_dels].len = (uint32_t)(ref_len - alignment_end); ++n_dels; }
	} else { dels[n_dels].pos = 0; dels[n_dels].len = (uint32_t)ref_len; ++n_dels; }
	res->n_subs = n_subs; res->n_dels = n_dels; res->n_inss = n_inss; res->n_ins_bases = n_ib;
	free(ref); free(qry); free(aq); free(ar); free(stripped);
	return 0;
}
