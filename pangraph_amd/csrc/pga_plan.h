// pga_plan.h -- records exchanged between the alignment driver (pga_align.cpp) and the device-side planner (pga_plan.hip)
#pragma once
#include "pga_common.h"

namespace pga {

struct PlanIn {                 // one region = one chain (or a piece split off one)
	uint64_t a_off;             // index of the QUERY's first compacted anchor in the device array
	uint64_t item_off;          // this region's slice of the item pool
	uint32_t item_cap;
	int32_t n_a;                // anchors of the query (neighbouring chains are scanned for the extension windows)
	int32_t as, cnt;            // the chain inside the query's anchors
	int32_t qlen, qid, base;    // query length, sequence index of the query, first sequence of its group
	int32_t pad;
};

struct PlanOut {
	int32_t status;             // 0 ok, 1 item slice too small (n_items says how many), 2 too many long gaps for the kernel, 3 empty chain
	int32_t rid, rev;
	int32_t r_rs, r_re, r_qs, r_qe, r_mlen, r_blen;          // chain_extent (hit.c:8-38)
	int32_t as1, cnt1, rs, qs, rs0, qs0, re0, qe0, T_re, T_qe; // what mm_align1 fixes before its first DP call (align.c:583-700)
	uint32_t n_items; int32_t n_long_gaps;
};

struct PlanItem {               // kind 0: a RUN of consecutive segments the identity probe answered "nM" (m mismatches in all, bw1 = how many segments);
	int32_t kind, i, rs, qs, re, qe, bw1, m, i_prev;          // kind 1 / 2: a segment that needs a DP problem (2: equally long windows, the probe said no); i = its
	int32_t pad[3];             // last anchor (chain-relative to as1), i_prev = the anchor it starts at
};

struct PlanParams {
	int32_t k, bw, bw_long, max_gap, min_cnt, min_chain_score, min_ksw_len, a, q, e, no_end_flt, probe_m_max;
	int32_t g_max, pad;         // long gaps of a region the kernel plans itself (<= PLAN_G_MAX of pga_plan.hip; PGA_PLAN_G_MAX lowers it: tests of the host route)
	int64_t max_sw_mat;
};

void plan_regions(const std::vector<PlanIn> &in, u128 *d_anchors, PkBases bases, const uint64_t *d_seq_off, const uint32_t *d_seq_len, const PlanParams &P,
                  std::vector<PlanOut> &out, std::vector<PlanItem> &items, hipStream_t st);
void gather_anchors(const std::vector<uint64_t> &idx, const u128 *d_anchors, std::vector<u128> &out, hipStream_t st);

} // namespace pga
