/* Prints the layout of the minimap2-sys ABI structs as seen through whatever header ABI_HEADER names: sizes, offsets of the plain
 * fields, and -- for bit-fields -- the byte image of a zeroed struct with that one field set to all ones.  tests/test_abi.py builds it
 * twice (reference minimap.h, include/pga_mm2_abi.h) and compares the outputs. */
#include <stdio.h>
#include <stddef.h>
#include <string.h>
#include ABI_HEADER

#define OFF(T, f) printf(#T "." #f " %zu %zu\n", offsetof(T, f), sizeof(((T*)0)->f))
#define BITS(T, f) do { T v; memset(&v, 0, sizeof(v)); v.f = ~0u; printf(#T "." #f " bits"); { const unsigned char *p = (const unsigned char*)&v; size_t i; for (i = 0; i < sizeof(v); ++i) if (p[i]) printf(" %zu:%02x", i, p[i]); } printf("\n"); } while (0)

int main(void)
{
	printf("sizeof %zu %zu %zu %zu %zu %zu\n", sizeof(mm_idxopt_t), sizeof(mm_mapopt_t), sizeof(mm_reg1_t), sizeof(mm_extra_t), sizeof(mm_idx_t), sizeof(mm_idx_seq_t));
	OFF(mm_idxopt_t, k); OFF(mm_idxopt_t, w); OFF(mm_idxopt_t, flag); OFF(mm_idxopt_t, bucket_bits); OFF(mm_idxopt_t, mini_batch_size); OFF(mm_idxopt_t, batch_size);
	OFF(mm_mapopt_t, flag); OFF(mm_mapopt_t, seed); OFF(mm_mapopt_t, sdust_thres); OFF(mm_mapopt_t, max_qlen); OFF(mm_mapopt_t, bw); OFF(mm_mapopt_t, bw_long);
	OFF(mm_mapopt_t, max_gap); OFF(mm_mapopt_t, max_gap_ref); OFF(mm_mapopt_t, max_frag_len); OFF(mm_mapopt_t, max_chain_skip); OFF(mm_mapopt_t, max_chain_iter);
	OFF(mm_mapopt_t, min_cnt); OFF(mm_mapopt_t, min_chain_score); OFF(mm_mapopt_t, chain_gap_scale); OFF(mm_mapopt_t, chain_skip_scale); OFF(mm_mapopt_t, rmq_size_cap);
	OFF(mm_mapopt_t, rmq_inner_dist); OFF(mm_mapopt_t, rmq_rescue_size); OFF(mm_mapopt_t, rmq_rescue_ratio); OFF(mm_mapopt_t, mask_level); OFF(mm_mapopt_t, mask_len);
	OFF(mm_mapopt_t, pri_ratio); OFF(mm_mapopt_t, best_n); OFF(mm_mapopt_t, alt_drop); OFF(mm_mapopt_t, a); OFF(mm_mapopt_t, b); OFF(mm_mapopt_t, q); OFF(mm_mapopt_t, e);
	OFF(mm_mapopt_t, q2); OFF(mm_mapopt_t, e2); OFF(mm_mapopt_t, sc_ambi); OFF(mm_mapopt_t, noncan); OFF(mm_mapopt_t, junc_bonus); OFF(mm_mapopt_t, zdrop); OFF(mm_mapopt_t, zdrop_inv);
	OFF(mm_mapopt_t, end_bonus); OFF(mm_mapopt_t, min_dp_max); OFF(mm_mapopt_t, min_ksw_len); OFF(mm_mapopt_t, anchor_ext_len); OFF(mm_mapopt_t, anchor_ext_shift);
	OFF(mm_mapopt_t, max_clip_ratio); OFF(mm_mapopt_t, rank_min_len); OFF(mm_mapopt_t, rank_frac); OFF(mm_mapopt_t, pe_ori); OFF(mm_mapopt_t, pe_bonus); OFF(mm_mapopt_t, mid_occ_frac);
	OFF(mm_mapopt_t, q_occ_frac); OFF(mm_mapopt_t, min_mid_occ); OFF(mm_mapopt_t, max_mid_occ); OFF(mm_mapopt_t, mid_occ); OFF(mm_mapopt_t, max_occ); OFF(mm_mapopt_t, max_max_occ);
	OFF(mm_mapopt_t, occ_dist); OFF(mm_mapopt_t, mini_batch_size); OFF(mm_mapopt_t, max_sw_mat); OFF(mm_mapopt_t, cap_kalloc); OFF(mm_mapopt_t, split_prefix);
	OFF(mm_reg1_t, id); OFF(mm_reg1_t, cnt); OFF(mm_reg1_t, rid); OFF(mm_reg1_t, score); OFF(mm_reg1_t, qs); OFF(mm_reg1_t, qe); OFF(mm_reg1_t, rs); OFF(mm_reg1_t, re);
	OFF(mm_reg1_t, parent); OFF(mm_reg1_t, subsc); OFF(mm_reg1_t, as); OFF(mm_reg1_t, mlen); OFF(mm_reg1_t, blen); OFF(mm_reg1_t, n_sub); OFF(mm_reg1_t, score0);
	OFF(mm_reg1_t, hash); OFF(mm_reg1_t, div); OFF(mm_reg1_t, p);
	BITS(mm_reg1_t, mapq); BITS(mm_reg1_t, split); BITS(mm_reg1_t, rev); BITS(mm_reg1_t, inv); BITS(mm_reg1_t, sam_pri); BITS(mm_reg1_t, proper_frag); BITS(mm_reg1_t, pe_thru);
	BITS(mm_reg1_t, seg_split); BITS(mm_reg1_t, seg_id); BITS(mm_reg1_t, split_inv); BITS(mm_reg1_t, is_alt); BITS(mm_reg1_t, strand_retained);
	OFF(mm_extra_t, capacity); OFF(mm_extra_t, dp_score); OFF(mm_extra_t, dp_max); OFF(mm_extra_t, dp_max2); OFF(mm_extra_t, n_cigar); printf("mm_extra_t.cigar %zu\n", offsetof(mm_extra_t, cigar));
	BITS(mm_extra_t, n_ambi); BITS(mm_extra_t, trans_strand);
	OFF(mm_idx_t, b); OFF(mm_idx_t, w); OFF(mm_idx_t, k); OFF(mm_idx_t, flag); OFF(mm_idx_t, n_seq); OFF(mm_idx_t, index); OFF(mm_idx_t, n_alt); OFF(mm_idx_t, seq); OFF(mm_idx_t, S);
	OFF(mm_idx_seq_t, name); OFF(mm_idx_seq_t, offset); OFF(mm_idx_seq_t, len); OFF(mm_idx_seq_t, is_alt);
	printf("flags %lld %lld %lld %lld %lld %lld %lld %d\n", (long long)MM_F_NO_DIAG, (long long)MM_F_NO_DUAL, (long long)MM_F_CIGAR, (long long)MM_F_OUT_CG, (long long)MM_F_NO_LJOIN,
	       (long long)MM_F_ALL_CHAINS, (long long)MM_F_RMQ, MM_I_HPC);
	return 0;
}
