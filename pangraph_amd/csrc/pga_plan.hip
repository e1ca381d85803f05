// pga_plan.hip -- kernel group #7b: what mm_align1 decides BEFORE its first DP call (C/align.c:583-745), for every region of a batch, on the
// device, from the chain anchors where the chaining stage left them.  One WAVE per region:
//
//   chain_extent        coordinates and approximate match / block lengths of the chain (hit.c:8-38): two wave reductions
//   trim_chain_ends     mm_fix_bad_ends (align.c:471-509): a short walk from either end
//   long gaps           positions whose indel exceeds 10 / 30 bases, compacted into LDS; mm_filter_bad_seeds and mm_filter_bad_seeds_alt
//                       (align.c:392-469) walk that short list and flag anchors IN PLACE (MM_SEED_IGNORE, MM_SEED_LONG_JOIN): a region split off
//                       later is planned from the same, flagged anchors, as in the reference
//   windows             the extension windows (align.c:633-696), incl. the scans over neighbouring chains for the 4th anchor beyond either end
//   gap-fill segments   the greedy cut every >= min_ksw_len bases (align.c:726-745), EVERY SEGMENT PROBED ON THE SPOT: equally long windows that differ
//                       in at most m_max positions are "nM" (proof: pga_ksw_fast.hip; pga_align.cpp: Driver::lean_probes), and consecutive such
//                       segments collapse into one RUN.  What leaves the device is the handful of segments that need a DP problem and the runs between
//                       them -- a whole-genome chain of 250 000 anchors and 12 000 segments comes back as a few dozen records.
//
// The host (pga_align.cpp) used to download all anchors (0.5 GB per leaf batch), walk them four times per region and build a problem record per
// segment; it now keeps control flow over the records this kernel returns.  Integer logic only; every loop below names the host function it
// restates (pga_align.cpp) and the reference lines behind that.
#include "pga_common.h"
#include "pga_plan.h"
#include "pga_wave.h"

namespace pga {

static const uint64_t PA_LONG_JOIN = 1ULL << 40, PA_IGNORE = 1ULL << 41, PA_TANDEM = 1ULL << 42, PA_SELF = 1ULL << 43;
#define PLAN_G_MAX 4096          // long gaps of one region kept in LDS; a region with more goes back to the host path

struct DA {                       // device view of a query's compacted anchors (same accessors as pga_align.cpp: Anchors)
	u128 *a; int32_t n;
	__device__ __forceinline__ int32_t tpos(int i) const { return (int32_t)a[i].x; }
	__device__ __forceinline__ int32_t qpos(int i) const { return (int32_t)a[i].y; }
	__device__ __forceinline__ int32_t span(int i) const { return (int32_t)(a[i].y >> 32 & 0xff); }
	__device__ __forceinline__ uint64_t target_key(int i) const { return a[i].x >> 32; }
	__device__ __forceinline__ bool flagged(int i, uint64_t f) const { return (a[i].y & f) != 0; }
	__device__ __forceinline__ int32_t indel(int i) const { return (qpos(i) - qpos(i - 1)) - (tpos(i) - tpos(i - 1)); }
};

__device__ __forceinline__ int plan_min(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int plan_max(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int plan_abs(int a) { return a < 0 ? -a : a; }

// the n-th (0-based) set bit of m; m has more than n bits set
__device__ __forceinline__ int nth_bit(unsigned long long m, int n) { for (int k = 0; k < n; ++k) m &= m - 1; return __ffsll((long long)m) - 1; }

__global__ __launch_bounds__(64)
void k_plan_regions(const PlanIn *__restrict__ in, uint32_t n_regions, u128 *__restrict__ anchors, const uint32_t *__restrict__ seq_len,
                    PlanParams P, PlanOut *__restrict__ out)
{
	__shared__ int32_t s_G[PLAN_G_MAX];            // chain-relative indices of the long gaps (|indel| > 10), ascending
	__shared__ int32_t s_F[PLAN_G_MAX * 2];        // ranges to flag: (from, to) pairs, and single LONG_JOIN positions as (pos, -1)
	const int lane = threadIdx.x;
	const uint32_t rid_x = blockIdx.x;
	if (rid_x >= n_regions) return;
	const PlanIn R = in[rid_x];
	DA A{anchors + R.a_off, R.n_a};
	PlanOut O; memset(&O, 0, sizeof(O));
	const int32_t qlen = R.qlen;
	const int r_as = R.as, r_cnt = R.cnt;
	if (r_cnt <= 0) { O.status = 3; if (lane == 0) out[rid_x] = O; return; }
	const int first = r_as, last = r_as + r_cnt - 1;
	const uint64_t x0 = A.a[first].x;
	const int32_t rid = (int32_t)(x0 << 1 >> 33), rev = (int32_t)(x0 >> 63);
	const int32_t tlen_ref = (int32_t)seq_len[R.base + rid];
	// ---- chain_extent (pga_align.cpp; hit.c:8-38) ----
	int32_t r_rs, r_re, r_qs, r_qe, r_mlen, r_blen;
	{
		const int32_t sp0 = A.span(first);
		r_rs = plan_max(0, A.tpos(first) + 1 - sp0);
		r_re = A.tpos(last) + 1;
		const int32_t q_lo = A.qpos(first) + 1 - sp0, q_hi = A.qpos(last) + 1;
		if (rev) r_qs = qlen - q_hi, r_qe = qlen - q_lo; else r_qs = q_lo, r_qe = q_hi;
		uint32_t cov = 0, blk = 0;
		// (a whole-genome chain holds half a million anchors and ONE wave streams them: the loop is bound by the latency of a load, not by its bytes.  Round 6:
		// every anchor is loaded ONCE -- its predecessor is the neighbouring lane's anchor, one DPP move per field, the window's first lane takes the last anchor
		// of the window before -- and EIGHT windows of loads are in flight per trip: half the loads, twice the anchors per round trip)
		int32_t cx = (int32_t)A.a[first].x, cy = (int32_t)A.a[first].y;              // the anchor in front of the trip's first (uniform)
		for (int i0 = first + 1; i0 <= last; i0 += 512) {
			u128 cur[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) { const int i = i0 + 64 * u + lane; if (i <= last) cur[u] = A.a[i]; else { cur[u].x = cur[u].y = 0; } }
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const int i = i0 + 64 * u + lane;
				const int32_t x = (int32_t)cur[u].x, y = (int32_t)cur[u].y;
				const int32_t px = wave_shr1(x, cx), py = wave_shr1(y, cy);
				if (i <= last) {
					const int32_t dt = x - px, dq = y - py, sp = (int32_t)(cur[u].y >> 32 & 0xff);
					blk += (uint32_t)plan_max(dt, dq);
					cov += (uint32_t)((dt > sp && dq > sp) ? sp : plan_min(dt, dq));
				}
				cx = __builtin_amdgcn_readlane(x, 63); cy = __builtin_amdgcn_readlane(y, 63);      // (only read by a later window, which exists only if this one was full)
			}
		}
		r_mlen = (int32_t)((uint32_t)sp0 + (uint32_t)__builtin_amdgcn_readlane((int)wave_prefix_sum_incl(cov), 63));
		r_blen = (int32_t)((uint32_t)sp0 + (uint32_t)__builtin_amdgcn_readlane((int)wave_prefix_sum_incl(blk), 63));
	}
	O.rid = rid, O.rev = rev, O.r_rs = r_rs, O.r_re = r_re, O.r_qs = r_qs, O.r_qe = r_qe, O.r_mlen = r_mlen, O.r_blen = r_blen;
	// ---- trim_chain_ends (align.c:471-509) ----
	int32_t as1 = r_as, cnt1 = r_cnt;
	if (!P.no_end_flt && r_cnt >= 3) {
		const int bw = P.bw, min_match = P.min_chain_score * 2;
		// (uniform: every lane walks the same few anchors; the loads are broadcast)
		int32_t len, match;
		len = match = A.span(r_as);
		for (int i = r_as + 1; i < last; ++i) {
			if (A.flagged(i, PA_LONG_JOIN)) break;
			const int32_t dt = A.tpos(i) - A.tpos(i - 1), dq = A.qpos(i) - A.qpos(i - 1), lo = plan_min(dt, dq), hi = plan_max(dt, dq);
			if (hi - lo > len >> 1) as1 = i;
			len += lo, match += plan_min(lo, A.span(i));
			if (len >= bw << 1 || (match >= min_match && match >= bw) || match >= r_mlen >> 1) break;
		}
		cnt1 = last + 1 - as1;
		len = match = A.span(last);
		for (int i = last - 1; i > as1; --i) {
			if (A.flagged(i + 1, PA_LONG_JOIN)) break;
			const int32_t dt = A.tpos(i + 1) - A.tpos(i), dq = A.qpos(i + 1) - A.qpos(i), lo = plan_min(dt, dq), hi = plan_max(dt, dq);
			if (hi - lo > len >> 1) cnt1 = i + 1 - as1;
			len += lo, match += plan_min(lo, A.span(i + 1));
			if (len >= bw << 1 || (match >= min_match && match >= bw) || match >= r_mlen >> 1) break;
		}
	}
	// ---- long gaps of [as1, as1 + cnt1): chain-relative indices with |indel| > 10, in order ----
	int n_g = 0;
	{
		// (the same streaming form: an anchor is loaded once, its predecessor comes from the neighbouring lane, eight windows in flight)
		int32_t cx = (int32_t)A.a[as1].x, cy = (int32_t)A.a[as1].y;
		for (int i0 = 1; i0 < cnt1; i0 += 512) {
			u128 cur[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) { const int i = i0 + 64 * u + lane; if (i < cnt1) cur[u] = A.a[as1 + i]; else { cur[u].x = cur[u].y = 0; } }
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const int i = i0 + 64 * u + lane;
				const int32_t x = (int32_t)cur[u].x, y = (int32_t)cur[u].y;
				const int32_t px = wave_shr1(x, cx), py = wave_shr1(y, cy);
				const int32_t g = (y - py) - (x - px);                                         // A.indel(as1 + i)
				const bool lg = i < cnt1 && (g < -10 || g > 10);
				const unsigned long long m = __ballot(lg);
				if (lg) { const int o = n_g + __popcll(m & ((1ULL << lane) - 1)); if (o < PLAN_G_MAX) s_G[o] = i; }
				n_g += __popcll(m);
				cx = __builtin_amdgcn_readlane(x, 63); cy = __builtin_amdgcn_readlane(y, 63);
			}
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	if (n_g > P.g_max) { O.status = 2; if (lane == 0) out[rid_x] = O; return; }
	// ---- ignore_indel_bursts(A, as1, cnt1, 10, 40, max_gap >> 1, 10) (align.c:392-431) and join_crowded_gaps(A, as1, cnt1, 30, max_gap >> 1)
	// (align.c:433-469): both read indels only, so they are evaluated on the unflagged anchors and the flags are set afterwards ----
	int n_f = 0;
	if (lane == 0) {
		const int reach_bases = P.max_gap >> 1;
		if (n_g > 1) {
			const int n = n_g, weight_min = 40, reach_gaps = 10;
			int cur_from = -1, cur_to = -1, cur_weight = 0;
			for (int k = 0;; ++k) {
				if (k == n || k >= cur_to) {
					if (cur_to > 0) { s_F[2 * n_f] = s_G[cur_from]; s_F[2 * n_f + 1] = s_G[cur_to]; ++n_f; }
					cur_from = -1, cur_to = -1, cur_weight = 0;
					if (k == n) break;
				}
				const int i0 = as1 + s_G[k];
				int ins = 0, del = 0, best_w = 0, best_l = -1;
				{ const int g = A.indel(i0); if (g > 0) ins += g; else del -= g; }
				const int32_t q0 = A.qpos(i0 - 1), t0 = A.tpos(i0 - 1);
				for (int l = k + 1; l < n && l <= k + reach_gaps; ++l) {
					const int j = as1 + s_G[l];
					if (A.qpos(j) - q0 > reach_bases || A.tpos(j) - t0 > reach_bases) break;
					{ const int g = A.indel(j); if (g > 0) ins += g; else del -= g; }
					const int w = ins + del - plan_abs(ins - del);
					if (w > best_w) best_w = w, best_l = l;
				}
				if (best_w > weight_min && best_w > cur_weight) cur_from = k, cur_to = best_l, cur_weight = best_w;
			}
		}
		// the gaps above 30 bases: a sub-list of s_G
		int n30 = 0;
		for (int k = 0; k < n_g; ++k) { const int g = A.indel(as1 + s_G[k]); n30 += g < -30 || g > 30; }
		if (n30 > 1) {
			int k = 0;
			auto next30 = [&](int kk) { while (kk < n_g) { const int g = A.indel(as1 + s_G[kk]); if (g < -30 || g > 30) break; ++kk; } return kk; };
			k = next30(0);
			while (k < n_g) {
				const int i = as1 + s_G[k];
				int l = next30(k + 1), l_prev = k;
				int32_t t_end = A.tpos(i), q_end = A.qpos(i), g_prev = plan_abs(A.indel(i));
				bool joined = false;
				for (; l < n_g; l = next30(l + 1)) {
					const int j = as1 + s_G[l];
					if (A.qpos(j) - q_end > reach_bases || A.tpos(j) - t_end > reach_bases) break;
					const int32_t g = plan_abs(A.indel(j)), sp = A.span(j - 1);
					const int32_t room = plan_min(A.tpos(j - 1) + sp - t_end, A.qpos(j - 1) + sp - q_end);
					if (room > g_prev + g) break;
					t_end = A.tpos(j), q_end = A.qpos(j), g_prev = g;
					l_prev = l; joined = true;
				}
				if (joined) {
					const int lastg = s_G[l_prev];
					s_F[2 * n_f] = s_G[k]; s_F[2 * n_f + 1] = lastg; ++n_f;          // [G[k], last) ignored ...
					s_F[2 * n_f] = lastg; s_F[2 * n_f + 1] = -1; ++n_f;              // ... and `last` is the far side of one long gap
				}
				k = l;
			}
		}
	}
	n_f = __builtin_amdgcn_readfirstlane(n_f);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	for (int f = 0; f < n_f; ++f) {
		const int from = s_F[2 * f], to = s_F[2 * f + 1];
		if (to < 0) { if (lane == 0) A.a[as1 + from].y |= PA_LONG_JOIN; }
		else for (int i = from + lane; i < to; i += 64) A.a[as1 + i].y |= PA_IGNORE;
	}
	__threadfence();
	// ---- windows (align.c:633-696) ----
	const int half_k = P.k >> 1;
	int32_t rs = A.tpos(as1) - half_k, qs = A.qpos(as1) - half_k;
	int32_t re = A.tpos(as1 + cnt1 - 1) - half_k, qe = A.qpos(as1 + cnt1 - 1) - half_k;
	int32_t rs0 = plan_max(0, A.tpos(r_as) + 1 - A.span(r_as)), qs0 = A.qpos(r_as) + 1 - A.span(r_as);
	int32_t rs1 = 0, qs1 = 0, re0, qe0, re1, qe1;
	{
		// anchors of earlier chains on the same target, walked backwards: the (min_cnt + 1)-th one that lies before the chain on both sequences
		const uint64_t key = A.target_key(r_as);
		int l = 0; bool done = false;
		for (int i_hi = r_as - 1; i_hi >= 0 && !done; i_hi -= 64) {
			const int i = i_hi - lane;
			bool same = false, ok = false; int32_t x = 0, y = 0;
			if (i >= 0) { same = A.target_key(i) == key; if (same) { x = A.tpos(i) + 1 - A.span(i), y = A.qpos(i) + 1 - A.span(i); ok = x < rs0 && y < qs0; } }
			const unsigned long long ms = __ballot(same);
			const int n_same = ~ms ? __ffsll((long long)~ms) - 1 : 64;                 // the walk ends at the first anchor of another target
			const unsigned long long mo = __ballot(ok) & (n_same >= 64 ? ~0ULL : ((1ULL << n_same) - 1));
			const int c = __popcll(mo);
			if (l + c > P.min_cnt) {
				const int src = nth_bit(mo, P.min_cnt - l);
				const int32_t xs = __builtin_amdgcn_readlane(x, src), ys = __builtin_amdgcn_readlane(y, src);
				const int32_t ll = plan_max(rs0 - xs, qs0 - ys);
				rs1 = plan_max(0, rs0 - ll), qs1 = qs0 - ll;
				done = true;
			}
			l += c;
			if (n_same < 64) done = true;
		}
	}
	if (qs > 0 && rs > 0) {
		int32_t l = plan_min(qs, P.max_gap);
		qs1 = plan_max(qs1, qs - l);
		qs0 = plan_min(qs0, qs1);
		l += l * P.a > P.q ? (l * P.a - P.q) / P.e : 0;
		l = plan_min(plan_min(l, P.max_gap), rs);
		rs1 = plan_max(rs1, rs - l);
		rs0 = plan_min(plan_min(rs0, rs1), rs);
	} else rs0 = rs, qs0 = qs;
	re0 = A.tpos(r_as + r_cnt - 1) + 1, qe0 = A.qpos(r_as + r_cnt - 1) + 1;
	re1 = tlen_ref, qe1 = qlen;
	{
		const uint64_t key = A.target_key(r_as);
		int l = 0; bool done = false;
		for (int i_lo = r_as + r_cnt; i_lo < A.n && !done; i_lo += 64) {
			const int i = i_lo + lane;
			bool same = false, ok = false; int32_t x = 0, y = 0;
			if (i < A.n) { same = A.target_key(i) == key; if (same) { x = A.tpos(i) + 1, y = A.qpos(i) + 1; ok = x > re0 && y > qe0; } }
			const unsigned long long ms = __ballot(same);
			const int n_same = ~ms ? __ffsll((long long)~ms) - 1 : 64;
			const unsigned long long mo = __ballot(ok) & (n_same >= 64 ? ~0ULL : ((1ULL << n_same) - 1));
			const int c = __popcll(mo);
			if (l + c > P.min_cnt) {
				const int src = nth_bit(mo, P.min_cnt - l);
				const int32_t xs = __builtin_amdgcn_readlane(x, src), ys = __builtin_amdgcn_readlane(y, src);
				const int32_t ll = plan_max(xs - re0, ys - qe0);
				re1 = re0 + ll, qe1 = qe0 + ll;
				done = true;
			}
			l += c;
			if (n_same < 64) done = true;
		}
	}
	if (qe < qlen && re < tlen_ref) {
		int32_t l = plan_min(qlen - qe, P.max_gap);
		qe1 = plan_min(qe1, qe + l);
		qe0 = plan_max(qe0, qe1);
		l += l * P.a > P.q ? (l * P.a - P.q) / P.e : 0;
		l = plan_min(plan_min(l, P.max_gap), tlen_ref - re);
		re1 = plan_min(re1, re + l);
		re0 = plan_max(re0, re1);
	} else re0 = re, qe0 = qe;
	if (A.flagged(r_as, PA_SELF)) {
		int max_ext = plan_abs(r_qs - r_rs);
		if (r_rs - rs0 > max_ext) rs0 = r_rs - max_ext;
		if (r_qs - qs0 > max_ext) qs0 = r_qs - max_ext;
		max_ext = plan_abs(r_qe - r_re);
		if (re0 - r_re > max_ext) re0 = r_re + max_ext;
		if (qe0 - r_qe > max_ext) qe0 = r_qe + max_ext;
	}
	O.as1 = as1, O.cnt1 = cnt1, O.rs = rs, O.qs = qs, O.rs0 = rs0, O.qs0 = qs0, O.re0 = re0, O.qe0 = qe0;
	O.n_long_gaps = n_g; O.status = 0;
	if (lane == 0) out[rid_x] = O;
}

// Second half, step 1: the greedy cut of a region's chain into gap-fill segments (align.c:726-745).  Reads the (flagged) anchors, writes one record
// per segment into the region's slice of a device scratch array; equally long windows also get their identity probe laid out.
struct PlanSeg { int32_t i, i_prev, rs, qs, re, qe, bw1, m; uint64_t t_off, q_off; int32_t kind, qlen_full, q_rev, pad; };   // kind 0: not used, 1 / 2 needs a problem (2: probe-eligible), 3: probe pending; m: the probe's answer

__global__ __launch_bounds__(64)
void k_plan_cut(const PlanIn *__restrict__ in, uint32_t n_regions, const u128 *__restrict__ anchors, const uint64_t *__restrict__ seq_off, PlanParams P,
                PlanOut *__restrict__ out, PlanSeg *__restrict__ segs, uint32_t *__restrict__ n_segs)
{
	const int lane = threadIdx.x;
	const uint32_t rid_x = blockIdx.x;
	if (rid_x >= n_regions) return;
	const PlanIn R = in[rid_x];
	PlanOut O = out[rid_x];
	if (lane == 0) n_segs[rid_x] = 0;
	if (O.status == 2 || O.status == 3) return;
	const u128 *A = anchors + R.a_off;
	const int32_t qlen = R.qlen, rid = O.rid, rev = O.rev, as1 = O.as1, cnt1 = O.cnt1;
	const int half_k = P.k >> 1;
	const int32_t bw_long = P.bw_long;
	const uint64_t t_base = seq_off[R.base + rid], q_base = seq_off[R.qid];
	PlanSeg *sg = segs + R.item_off;
	uint32_t n = 0;
	int32_t seg_rs = O.rs, seg_qs = O.qs, seg_i_prev = 0;
	int32_t re = O.rs, qe = O.qs;
	u128 v_next; v_next.x = v_next.y = 0;
	if (1 + lane < cnt1) v_next = A[as1 + 1 + lane];
	for (int i0 = 1; i0 < cnt1; i0 += 64) {
		const int i = i0 + lane;
		bool use = false, lj = false; int32_t ce = 0, cq = 0;
		const u128 v = v_next;
		if (i + 64 < cnt1) v_next = A[as1 + i + 64];                                   // (the next window is on its way while this one is cut: one wave, half a million anchors)
		if (i < cnt1) {
			const bool skip = (v.y & (PA_IGNORE | PA_TANDEM)) != 0 && i != cnt1 - 1;
			use = !skip; lj = (v.y & PA_LONG_JOIN) != 0;
			ce = (int32_t)v.x - half_k, cq = (int32_t)v.y - half_k;
		}
		const unsigned long long m_lj = __ballot(lj);
		unsigned long long todo = __ballot(use);                                   // anchors of this window that have not been looked at
		while (todo) {
			// the first remaining anchor that closes a segment from (seg_rs, seg_qs)
			const bool cut = use && ((todo >> lane) & 1) && (i == cnt1 - 1 || lj || (cq - seg_qs >= P.min_ksw_len && ce - seg_rs >= P.min_ksw_len));
			const unsigned long long mc = __ballot(cut);
			if (!mc) break;
			const int src = __ffsll((long long)mc) - 1;
			const int32_t s_re = __builtin_amdgcn_readlane(ce, src), s_qe = __builtin_amdgcn_readlane(cq, src);
			const bool s_lj = (m_lj >> src) & 1;
			const int s_i = i0 + src;
			const int32_t tl = s_re - seg_rs, ql = s_qe - seg_qs;
			int32_t bw1 = bw_long;
			if (s_lj) bw1 = plan_max(ql, tl);
			const bool eligible = ql == tl && bw1 >= ql;
			const bool probe = eligible && P.probe_m_max >= 0 && ql > 0 && !(P.max_sw_mat > 0 && (int64_t)ql * tl > P.max_sw_mat);
			if (n < R.item_cap && lane == 0) {
				PlanSeg x; x.i = s_i, x.i_prev = seg_i_prev, x.rs = seg_rs, x.qs = seg_qs, x.re = s_re, x.qe = s_qe, x.bw1 = bw1, x.m = -1;
				x.t_off = t_base + (uint64_t)seg_rs, x.q_off = q_base, x.qlen_full = qlen, x.q_rev = rev, x.kind = probe ? 3 : eligible ? 2 : 1, x.pad = 0;
				sg[n] = x;
			}
			++n;
			seg_rs = s_re, seg_qs = s_qe, seg_i_prev = s_i;
			todo &= ~((2ULL << src) - 1);                                              // everything up to and including src has been consumed
		}
		// the last anchor looked at in this window that was not skipped sets what T.re / T.qe hold at the end
		const unsigned long long mu = __ballot(use);
		if (mu) { const int lastu = 63 - __clzll((long long)mu); re = __builtin_amdgcn_readlane(ce, lastu); qe = __builtin_amdgcn_readlane(cq, lastu); }
	}
	if (lane == 0) {
		O.T_re = re, O.T_qe = qe;
		if (cnt1 == 1) O.T_re = (int32_t)A[as1].x - half_k, O.T_qe = (int32_t)A[as1].y - half_k;
		O.status = n > R.item_cap ? 1 : 0;
		O.n_items = n;                                                            // (for now: the number of segments)
		out[rid_x] = O;
		n_segs[rid_x] = n <= R.item_cap ? n : 0;
	}
}

// step 2: the identity probes of all segments of all regions (pga_post.hip: k_seg_identity; proof: pga_ksw_fast.hip), one wave per segment slot
__global__ __launch_bounds__(256)
void k_plan_probe(PlanSeg *__restrict__ segs, uint64_t n_slots, PkBases bases, int m_max)
{
	const int lane = threadIdx.x & 63;
	const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
	for (uint64_t s = wave; s < n_slots; s += n_waves) {
		const int kind = segs[s].kind;
		if (kind != 3) continue;
		const PlanSeg X = segs[s];
		const int len = X.qe - X.qs;
		const uint64_t t = X.t_off;
		const uint64_t q = X.q_rev ? X.q_off + (uint64_t)(X.qlen_full - 1 - X.qs) : X.q_off + (uint64_t)X.qs;
		int m = 0;
		for (int b = 0; b < len; b += 64) {
			const int kq = b + lane;
			bool bad = false, diff = false;
			if (kq < len) {
				const int x = bases.at(t + (uint64_t)kq), y = bases.at(X.q_rev ? q - (uint64_t)kq : q + (uint64_t)kq);
				bad = (x | y) > 3;
				diff = X.q_rev ? x != 3 - y : x != y;
			}
			if (__ballot(bad)) { m = -1; break; }
			m += __popcll(__ballot(diff));
			if (m > m_max) { m = -1; break; }
		}
		if (lane == 0) { segs[s].m = m; segs[s].kind = m >= 0 ? 0 : 2; }           // kind 0 from here on: answered "nM"
	}
}

// step 3: consecutive answered segments collapse into one RUN; what is left are the segments that need a problem.  Items go to a compact pool.
template <bool WRITE>
__global__ __launch_bounds__(64)
void k_plan_collapse(const PlanIn *__restrict__ in, uint32_t n_regions, const PlanSeg *__restrict__ segs, const uint32_t *__restrict__ n_segs, PlanOut *__restrict__ out,
                     PlanItem *__restrict__ items, const uint64_t *__restrict__ item_off)
{
	const int lane = threadIdx.x;
	const uint32_t rid_x = blockIdx.x;
	if (rid_x >= n_regions) return;
	PlanOut O = out[rid_x];
	if (O.status != 0) return;
	const PlanSeg *sg = segs + in[rid_x].item_off;
	const int n = (int)n_segs[rid_x];
	// pass 1: how many items: a run before every problem segment that follows answered ones, the problem segments, a closing run
	uint32_t n_items = 0; bool open = false;
	for (int s0 = 0; s0 < n; s0 += 64) {
		const int s = s0 + lane;
		const bool ans = s < n && sg[s].kind == 0, hard = s < n && !ans;
		const unsigned long long ma = __ballot(ans), mh = __ballot(hard);
		// a run starts at an answered segment whose predecessor is not answered (or, for lane 0, when no run is open)
		const unsigned long long starts = ma & ~((ma << 1) | (open ? 1ULL : 0ULL));
		n_items += (uint32_t)(__popcll(starts) + __popcll(mh));
		const int lastv = s0 + 63 < n ? 63 : n - 1 - s0;
		open = (ma >> lastv) & 1;
	}
	if (!WRITE) { if (lane == 0) { O.n_items = n_items; out[rid_x] = O; } return; }
	// pass 2: write them, in order (one lane: a region has a few dozen items; the answered segments in between are summed by the wave)
	PlanItem *it = items + item_off[rid_x];
	uint32_t w = 0; PlanItem run; memset(&run, 0, sizeof(run)); bool ropen = false;
	for (int s0 = 0; s0 < n; s0 += 64) {
		const int s = s0 + lane;
		PlanSeg X; memset(&X, 0, sizeof(X)); X.kind = -1;
		if (s < n) X = sg[s];
		const bool ans = X.kind == 0, hard = s < n && !ans;
		unsigned long long mh = __ballot(hard);
		const uint32_t mm = ans ? (uint32_t)X.m : 0u;
		const uint32_t ps = wave_prefix_sum_incl(mm);                                   // mismatches of the answered segments up to this lane
		const int lim = s0 + 64 <= n ? 64 : n - s0;
		int pos = 0;
		while (pos < lim) {
			const unsigned long long rest = mh & ~((1ULL << pos) - 1);
			const int h = rest ? __ffsll((long long)rest) - 1 : lim;                      // next problem segment (or the end of the window)
			if (h > pos) {
				// answered segments [pos, h): they extend the open run or open one
				const uint32_t sum = (uint32_t)__builtin_amdgcn_readlane((int)ps, h - 1) - (pos > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)ps, pos - 1) : 0u);
				if (!ropen) { ropen = true; run.kind = 0; run.rs = __builtin_amdgcn_readlane(X.rs, pos); run.qs = __builtin_amdgcn_readlane(X.qs, pos); run.i_prev = __builtin_amdgcn_readlane(X.i_prev, pos); run.m = 0; run.bw1 = 0; }
				run.i = __builtin_amdgcn_readlane(X.i, h - 1); run.re = __builtin_amdgcn_readlane(X.re, h - 1); run.qe = __builtin_amdgcn_readlane(X.qe, h - 1);
				run.m += (int32_t)sum; run.bw1 += h - pos;
			}
			if (h < lim) {
				if (ropen) { if (lane == 0) it[w] = run; ++w; ropen = false; }
				PlanItem hi; hi.kind = __builtin_amdgcn_readlane(X.kind, h); hi.i = __builtin_amdgcn_readlane(X.i, h); hi.rs = __builtin_amdgcn_readlane(X.rs, h); hi.qs = __builtin_amdgcn_readlane(X.qs, h);
				hi.re = __builtin_amdgcn_readlane(X.re, h); hi.qe = __builtin_amdgcn_readlane(X.qe, h); hi.bw1 = __builtin_amdgcn_readlane(X.bw1, h); hi.m = 0; hi.i_prev = __builtin_amdgcn_readlane(X.i_prev, h);
				hi.pad[0] = hi.pad[1] = hi.pad[2] = 0;
				if (lane == 0) it[w] = hi;
				++w;
				pos = h + 1;
			} else pos = lim;
		}
	}
	if (ropen) { if (lane == 0) it[w] = run; ++w; }
	if (lane == 0 && w != n_items) { O.status = 1; out[rid_x] = O; }                      // (cannot happen: both passes count the same items)
}

__global__ void k_gather_anchors(const uint64_t *__restrict__ idx, uint32_t n, const u128 *__restrict__ a, u128 *__restrict__ out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = a[idx[i]];
}

// Small lists (the calls of the upper tree: a few hundred regions) are planned through PINNED host memory that the kernels read and write directly
// (the host's pointers are the device's): the three hand-overs of the sequence below then cost a stream synchronisation each instead of a copy dispatch
// and a synchronisation -- seven runtime dispatches less per call.  PGA_PLAN_ZEROCOPY_MAX=0: always through device copies.
void plan_regions(const std::vector<PlanIn> &in_, u128 *d_anchors, PkBases bases, const uint64_t *d_seq_off, const uint32_t *d_seq_len, const PlanParams &P,
                  std::vector<PlanOut> &out, std::vector<PlanItem> &items, hipStream_t st)
{
	std::vector<PlanIn> in = in_;
	const size_t n = in.size();
	out.resize(n);
	items.clear();
	if (!n) return;
	static const size_t zc_max = getenv("PGA_PLAN_ZEROCOPY_MAX") ? (size_t)atol(getenv("PGA_PLAN_ZEROCOPY_MAX")) : 1024;
	const bool zc = n <= zc_max;
	PinVec<PlanIn> h_in; PinVec<PlanOut> h_out;
	DBuf<PlanIn> d_in; DBuf<PlanOut> d_out;
	PlanIn *p_in; PlanOut *p_out;
	if (zc) { h_in.resize(n); memcpy(h_in.data(), in.data(), n * sizeof(PlanIn)); h_out.resize(n); p_in = h_in.data(); p_out = h_out.data(); }
	else { d_in.upload(in, st); d_out.alloc(n); p_in = d_in.p; p_out = d_out.p; }
	hipLaunchKernelGGL(k_plan_regions, dim3((unsigned)n), dim3(64), 0, st, p_in, (uint32_t)n, d_anchors, d_seq_len, P, p_out);
	PGA_HIP(hipGetLastError());
	if (!zc) PGA_HIP(hipMemcpyAsync(out.data(), d_out.p, n * sizeof(PlanOut), hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
	if (zc) memcpy(out.data(), h_out.data(), n * sizeof(PlanOut));
	// segment slices: a segment spans at least min_ksw_len bases on both sequences unless it ends at a LONG_JOIN anchor (a long gap) or at the chain's end
	uint64_t slots = 0;
	for (size_t i = 0; i < n; ++i) {
		const PlanOut &o = out[i];
		const int64_t span = std::max<int64_t>((int64_t)o.r_re - o.r_rs, (int64_t)o.r_qe - o.r_qs);
		in[i].item_cap = (o.status == 2 || o.status == 3) ? 0u : (uint32_t)(span / std::max(1, P.min_ksw_len) + o.n_long_gaps + 4);
		in[i].item_off = slots; slots += in[i].item_cap;
	}
	if (zc) memcpy(h_in.data(), in.data(), n * sizeof(PlanIn)); else d_in.upload(in, st);
	DBuf<PlanSeg> d_segs((size_t)slots + 1);
	PGA_HIP(hipMemsetAsync(d_segs.p, 0, ((size_t)slots + 1) * sizeof(PlanSeg), st));
	DBuf<uint32_t> d_nseg(n);
	hipLaunchKernelGGL(k_plan_cut, dim3((unsigned)n), dim3(64), 0, st, p_in, (uint32_t)n, d_anchors, d_seq_off, P, p_out, d_segs.p, d_nseg.p);
	if (slots) hipLaunchKernelGGL(k_plan_probe, dim3((unsigned)std::min<uint64_t>((slots + 3) / 4, 256 * 32)), dim3(256), 0, st, d_segs.p, slots, bases, P.probe_m_max);
	hipLaunchKernelGGL(k_plan_collapse<false>, dim3((unsigned)n), dim3(64), 0, st, p_in, (uint32_t)n, d_segs.p, d_nseg.p, p_out, (PlanItem*)nullptr, (const uint64_t*)nullptr);
	PGA_HIP(hipGetLastError());
	if (!zc) PGA_HIP(hipMemcpyAsync(out.data(), d_out.p, n * sizeof(PlanOut), hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
	if (zc) memcpy(out.data(), h_out.data(), n * sizeof(PlanOut));
	for (size_t i = 0; i < n; ++i) if (out[i].status == 1) throw std::runtime_error("pga: plan_regions: a region cut more segments than its span allows");
	// the items of all regions, one region after the other (the order the caller takes them in)
	std::vector<uint64_t> ioff(n + 1, 0);
	for (size_t i = 0; i < n; ++i) ioff[i + 1] = ioff[i] + (out[i].status == 0 ? out[i].n_items : 0);
	items.resize((size_t)ioff[n]);
	if (ioff[n]) {
		const bool zci = zc && ioff[n] <= 8 * zc_max;
		PinVec<uint64_t> h_ioff; PinVec<PlanItem> h_items; DBuf<uint64_t> d_ioff; DBuf<PlanItem> d_items;
		const uint64_t *p_ioff; PlanItem *p_items;
		if (zci) { h_ioff.resize(n + 1); memcpy(h_ioff.data(), ioff.data(), (n + 1) * sizeof(uint64_t)); h_items.resize((size_t)ioff[n]); p_ioff = h_ioff.data(); p_items = h_items.data(); }
		else { d_ioff.upload(ioff, st); d_items.alloc((size_t)ioff[n]); p_ioff = d_ioff.p; p_items = d_items.p; }
		hipLaunchKernelGGL(k_plan_collapse<true>, dim3((unsigned)n), dim3(64), 0, st, p_in, (uint32_t)n, d_segs.p, d_nseg.p, p_out, p_items, p_ioff);
		PGA_HIP(hipGetLastError());
		if (!zci) PGA_HIP(hipMemcpyAsync(items.data(), d_items.p, (size_t)ioff[n] * sizeof(PlanItem), hipMemcpyDeviceToHost, st));
		PGA_HIP(sync_stream(st));
		if (zci) memcpy(items.data(), h_items.data(), (size_t)ioff[n] * sizeof(PlanItem));
	}
}

void gather_anchors(const std::vector<uint64_t> &idx, const u128 *d_anchors, std::vector<u128> &out, hipStream_t st)
{
	out.resize(idx.size());
	if (idx.empty()) return;
	if (idx.size() <= 4096) {                                     // (small: through pinned memory the kernel reads and writes directly)
		PinVec<uint64_t> h_idx; h_idx.resize(idx.size()); memcpy(h_idx.data(), idx.data(), idx.size() * sizeof(uint64_t));
		PinVec<u128> h_o; h_o.resize(idx.size());
		hipLaunchKernelGGL(k_gather_anchors, dim3((unsigned)((idx.size() + 255) / 256)), dim3(256), 0, st, h_idx.data(), (uint32_t)idx.size(), d_anchors, h_o.data());
		PGA_HIP(hipGetLastError());
		PGA_HIP(sync_stream(st));
		memcpy(out.data(), h_o.data(), idx.size() * sizeof(u128));
		return;
	}
	DBuf<uint64_t> d_idx; d_idx.upload(idx, st);
	DBuf<u128> d_o(idx.size());
	hipLaunchKernelGGL(k_gather_anchors, dim3((unsigned)((idx.size() + 255) / 256)), dim3(256), 0, st, d_idx.p, (uint32_t)idx.size(), d_anchors, d_o.p);
	PGA_HIP(hipMemcpyAsync(out.data(), d_o.p, idx.size() * sizeof(u128), hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
}

} // namespace pga
