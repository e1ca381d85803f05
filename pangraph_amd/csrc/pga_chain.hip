// pga_chain.hip -- kernel group #4: co-linear chaining of every query's anchors.
//
// Replaces mg_lchain_rmq / comput_sc_simple / mg_chain_backtrack / mg_chain_bk_end / compact_a
// (reference: packages/minimap2-sys/minimap2/lchain.c:9-111,232-368) and the range-min AVL tree behind it
// (krmq.h).  minimap2's asm presets chain with MM_F_RMQ (options.c:119).
//
// Why a re-enactment and not a textbook parallel range-min: which of several equal-priority predecessors the
// reference picks depends on the shape of its AVL tree and on the order in which subtree minima were refreshed
// (krmq.h:110-150), i.e. on the whole insert/erase history.  Bit-exact chains therefore need that history.
// What the GPU adds is parallelism ACROSS histories: the tree is empty whenever the sweep crosses to another
// (strand, target) or jumps more than max_dist in target position (lchain.c:295), so the anchor array of a
// batch is cut into independent SEGMENTS at those points and every segment is swept by its own lane with
// index-addressed nodes (node id == anchor id, no allocator).  Thousands of segments are in flight per batch;
// within a segment the sweep is sequential, as in the reference.  Backtracking and compaction run one lane
// per query (they depend on the reference's unstable sort of the per-anchor scores, pga_sort_exact.h).
#include "pga_common.h"
#include "pga_sort_exact.h"
#include "pga_sort_wave.h"
#include "pga_wave.h"
#include "pga_pipeline.h"
#include "pga_wg_sort.h"
#include <rocprim/rocprim.hpp>
#include <cstdio>

namespace pga {

struct ChainParams {
	int32_t max_dist, max_dist_inner, bw, max_skip, cap, min_cnt, min_sc;
	float pen_gap, pen_skip;
};

struct __attribute__((aligned(16))) CNode {
	double pri;
	int32_t y;
	int32_t c[2];
	int32_t s;
	uint32_t size;
	int32_t bal;
};

#define CMAXD 64

__device__ __forceinline__ float mg_log2(float x) // mmpriv.h:118-126
{
	union { float f; uint32_t i; } z = { x };
	float log_2 = (float)(((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	log_2 = __fadd_rn(log_2, __fsub_rn(__fmul_rn(__fadd_rn(__fmul_rn(-0.34484843f, z.f), 2.02466578f), z.f), 0.67487759f));
	return log_2;
}

// lchain.c:232-248 (all float operations individually rounded: no FMA contraction)
__device__ __forceinline__ int32_t score_pair(const u128 ai, const u128 aj, float pen_gap, float pen_skip, int32_t *exact, int32_t *width)
{
	int32_t dq = (int32_t)ai.y - (int32_t)aj.y, dr, dd, dg, q_span, sc;
	dr = (int32_t)(ai.x - aj.x);
	*width = dd = dr > dq ? dr - dq : dq - dr;
	dg = dr < dq ? dr : dq;
	q_span = (int32_t)(aj.y >> 32 & 0xff);
	sc = q_span < dg ? q_span : dg;
	if (exact) *exact = (dd == 0 && dg <= q_span);
	if (dd || dq > q_span) {
		float lin_pen = __fadd_rn(__fmul_rn(pen_gap, (float)dd), __fmul_rn(pen_skip, (float)dg));
		float log_pen = dd >= 1 ? mg_log2((float)(dd + 1)) : 0.0f;
		sc -= (int)__fadd_rn(lin_pen, __fmul_rn(.5f, log_pen));
	}
	return sc;
}

// ---- AVL tree on nodes nd[0..), keys (y, index); RMQ=true maintains subtree minima of pri ----
template <bool RMQ> struct Tree {
	CNode *nd; int32_t root;
	__device__ __forceinline__ int cmp(int32_t ya, int32_t ia, int32_t b) const {
		int32_t yb = nd[b].y; return ya < yb ? -1 : ya > yb ? 1 : (ia > b) - (ia < b);
	}
	__device__ __forceinline__ uint32_t sz(int32_t x) const { return x < 0 ? 0u : nd[x].size; }
	__device__ __forceinline__ void refresh(int32_t p, int32_t q, int32_t r) { // krmq.h:110-113 with explicit children
		if (!RMQ) return;
		int32_t s = (q < 0 || nd[p].pri < nd[nd[q].s].pri) ? p : nd[q].s;
		s = (r < 0 || nd[s].pri < nd[nd[r].s].pri) ? s : nd[r].s;
		nd[p].s = s;
	}
	__device__ int32_t rotate1(int32_t p, int dir) { // krmq.h:115-126
		int opp = 1 - dir;
		int32_t q = nd[p].c[opp], s = nd[p].s;
		uint32_t size_p = nd[p].size;
		nd[p].size -= nd[q].size - sz(nd[q].c[dir]);
		nd[q].size = size_p;
		refresh(p, nd[p].c[dir], nd[q].c[dir]);
		nd[q].s = s;
		nd[p].c[opp] = nd[q].c[dir];
		nd[q].c[dir] = p;
		return q;
	}
	__device__ int32_t rotate2(int32_t p, int dir) { // krmq.h:128-149
		int opp = 1 - dir, b1;
		int32_t q = nd[p].c[opp], r = nd[q].c[dir], s = nd[p].s;
		uint32_t size_x_dir = sz(nd[r].c[dir]);
		nd[r].size = nd[p].size;
		nd[p].size -= nd[q].size - size_x_dir;
		nd[q].size -= size_x_dir + 1;
		refresh(p, nd[p].c[dir], nd[r].c[dir]);
		refresh(q, nd[q].c[opp], nd[r].c[opp]);
		nd[r].s = s;
		nd[p].c[opp] = nd[r].c[dir];
		nd[r].c[dir] = p;
		nd[q].c[dir] = nd[r].c[opp];
		nd[r].c[opp] = q;
		b1 = dir == 0 ? +1 : -1;
		if (nd[r].bal == b1) nd[q].bal = 0, nd[p].bal = -b1;
		else if (nd[r].bal == 0) nd[q].bal = nd[p].bal = 0;
		else nd[q].bal = b1, nd[p].bal = 0;
		nd[r].bal = 0;
		return r;
	}
	__device__ void insert(int32_t x) { // krmq.h:152-200; nd[x].y/pri are set by the caller
		uint8_t stack[CMAXD]; int32_t path[CMAXD];
		int32_t bp = root, bq = -1, p, q, r;
		int top = 0, path_len = 0, which = 0;
		const int32_t yx = nd[x].y;
		for (p = bp, q = bq; p >= 0; q = p, p = nd[p].c[which]) {
			int c = cmp(yx, x, p);
			if (nd[p].bal != 0) bq = q, bp = p, top = 0;
			stack[top++] = (uint8_t)(which = (c > 0));
			path[path_len++] = p;
		}
		nd[x].bal = 0, nd[x].size = 1, nd[x].c[0] = nd[x].c[1] = -1, nd[x].s = x;
		if (q < 0) root = x; else nd[q].c[which] = x;
		if (bp < 0) return;
		for (int i = 0; i < path_len; ++i) ++nd[path[i]].size;
		if (RMQ) for (int i = path_len - 1; i >= 0; --i) {
			refresh(path[i], nd[path[i]].c[0], nd[path[i]].c[1]);
			if (nd[path[i]].s != x) break;
		}
		for (p = bp, top = 0; p != x; p = nd[p].c[stack[top]], ++top) {
			if (stack[top] == 0) --nd[p].bal; else ++nd[p].bal;
		}
		if (nd[bp].bal > -2 && nd[bp].bal < 2) return;
		which = (nd[bp].bal < 0);
		int b1 = which == 0 ? +1 : -1;
		q = nd[bp].c[1 - which];
		if (nd[q].bal == b1) { r = rotate1(bp, which); nd[q].bal = nd[bp].bal = 0; }
		else r = rotate2(bp, which);
		if (bq < 0) root = r; else nd[bq].c[bp != nd[bq].c[0]] = r;
	}
	// krmq.h:203-285; path slot 0 is the reference's stack copy of the root ("fake"), encoded as -2
	__device__ __forceinline__ int32_t getc(int32_t p, int d) const { return p == -2 ? (d == 0 ? root : -1) : nd[p].c[d]; }
	__device__ __forceinline__ void setc(int32_t p, int d, int32_t v) { if (p == -2) { if (d == 0) root = v; } else nd[p].c[d] = v; }
	__device__ void erase(int32_t x) {
		int32_t path[CMAXD], p; uint8_t dir[CMAXD];
		int d = 0, i;
		{
			int c = -1; p = -2;
			const int32_t yx = nd[x].y;
			while (c) { int which = (c > 0); dir[d] = (uint8_t)which, path[d++] = p; p = getc(p, which); c = cmp(yx, x, p); }
		}
		for (i = 1; i < d; ++i) --nd[path[i]].size;
		if (nd[p].c[1] < 0) setc(path[d-1], dir[d-1], nd[p].c[0]);
		else {
			int32_t q = nd[p].c[1];
			if (nd[q].c[0] < 0) {
				nd[q].c[0] = nd[p].c[0];
				nd[q].bal = nd[p].bal;
				setc(path[d-1], dir[d-1], q);
				path[d] = q, dir[d++] = 1;
				nd[q].size = nd[p].size - 1;
			} else {
				int32_t r; int e = d++;
				for (;;) { dir[d] = 0, path[d++] = q; r = nd[q].c[0]; if (nd[r].c[0] < 0) break; q = r; }
				nd[r].c[0] = nd[p].c[0];
				nd[q].c[0] = nd[r].c[1];
				nd[r].c[1] = nd[p].c[1];
				nd[r].bal = nd[p].bal;
				setc(path[e-1], dir[e-1], r);
				path[e] = r, dir[e] = 1;
				for (i = e + 1; i < d; ++i) --nd[path[i]].size;
				nd[r].size = nd[p].size - 1;
			}
		}
		if (RMQ) for (i = d - 1; i >= 1; --i) refresh(path[i], nd[path[i]].c[0], nd[path[i]].c[1]);
		while (--d > 0) {
			int32_t q = path[d];
			int which = dir[d], other = 1 - which, b1 = 1, b2 = 2;
			if (which) b1 = -b1, b2 = -b2;
			nd[q].bal += b1;
			if (nd[q].bal == b1) break;
			else if (nd[q].bal == b2) {
				int32_t r = nd[q].c[other];
				if (nd[r].bal == -b1) setc(path[d-1], dir[d-1], rotate2(q, which));
				else {
					setc(path[d-1], dir[d-1], rotate1(q, which));
					if (nd[r].bal == 0) { nd[r].bal = -b1; nd[q].bal = b1; break; }
					else nd[r].bal = nd[q].bal = 0;
				}
			}
		}
	}
	// krmq.h:98-140, closed interval [(lo_y,lo_i),(hi_y,hi_i)]
	__device__ int32_t rmq(int32_t lo_y, int32_t lo_i, int32_t hi_y, int32_t hi_i) const {
		int32_t path[2][CMAXD], p, mn; int8_t pc[2][CMAXD];
		int plen[2] = {0, 0}, i, c, lca;
		if (root < 0) return -1;
		for (p = root; p >= 0;) { c = cmp(lo_y, lo_i, p); path[0][plen[0]] = p, pc[0][plen[0]++] = (int8_t)c; if (c < 0) p = nd[p].c[0]; else if (c > 0) p = nd[p].c[1]; else break; }
		for (p = root; p >= 0;) { c = cmp(hi_y, hi_i, p); path[1][plen[1]] = p, pc[1][plen[1]++] = (int8_t)c; if (c < 0) p = nd[p].c[0]; else if (c > 0) p = nd[p].c[1]; else break; }
		for (i = 0; i < plen[0] && i < plen[1]; ++i)
			if (path[0][i] == path[1][i] && pc[0][i] <= 0 && pc[1][i] >= 0) break;
		if (i == plen[0] || i == plen[1]) return -1;
		lca = i, mn = path[0][lca];
		for (i = lca + 1; i < plen[0]; ++i) if (pc[0][i] <= 0) {
			int32_t u = path[0][i], r = nd[u].c[1];
			if (nd[u].pri < nd[mn].pri) mn = u;
			if (r >= 0 && nd[nd[r].s].pri < nd[mn].pri) mn = nd[r].s;
		}
		for (i = lca + 1; i < plen[1]; ++i) if (pc[1][i] >= 0) {
			int32_t u = path[1][i], l = nd[u].c[0];
			if (nd[u].pri < nd[mn].pri) mn = u;
			if (l >= 0 && nd[nd[l].s].pri < nd[mn].pri) mn = nd[l].s;
		}
		return mn;
	}
};

struct Iter { int32_t stack[CMAXD]; int top; };


// ------------------------------------------------------------------------------------------------------------
// Fast path: one WAVE per segment, no tree.  The range-min query of lchain.c:311 is a masked minimum over the
// live window [st,i0) kept in LDS rings; it is exact whenever the minimum priority is unique.  If two live
// candidates inside the query range tie on the minimum, the reference's answer depends on its AVL shape, so the
// wave gives up and flags the segment; flagged segments are re-run by k_chain_segments (the tree re-enactment).
// The inner scan of lchain.c:322-349 is evaluated without the t[] array: t[j]==i holds exactly when some
// candidate visited earlier in the scan (one with a larger key, hence any candidate at all) has p[]==j and
// passes the bandwidth test, so "marked" is a set-membership stamp; the n_skip walk itself stays sequential.
// ring capacity (anchors: live window + the staged block) is a template parameter; 2048 covers the ~800-anchor windows of
// max_gap = 10 kb with room to spare (1024 was tried: too many segments outgrow it and have to be swept twice)
#define CF_WI 1024         // inner-window ring capacity: the most inner candidates the fast path takes (repeats crowd the window)
#define CF_MAXIN 256       // inner candidates handled from registers; more than that go through the chunked form of the same scan

// lchain.c:232-248 on unpacked fields (segment-local: x is the 32-bit target position)
__device__ __forceinline__ int32_t score_pair32(int32_t xi, int32_t yi, int32_t xj, int32_t yj, int32_t span_j, float pen_gap, float pen_skip, int32_t *exact, int32_t *width)
{
	const int32_t dq = yi - yj, dr = xi - xj;
	const int32_t dd = dr > dq ? dr - dq : dq - dr;
	const int32_t dg = dr < dq ? dr : dq;
	int32_t sc = span_j < dg ? span_j : dg;
	*width = dd;
	if (exact) *exact = (dd == 0 && dg <= span_j);
	if (dd || dq > span_j) {
		float lin_pen = __fadd_rn(__fmul_rn(pen_gap, (float)dd), __fmul_rn(pen_skip, (float)dg));
		float log_pen = dd >= 1 ? mg_log2((float)(dd + 1)) : 0.0f;
		sc -= (int)__fadd_rn(lin_pen, __fmul_rn(.5f, log_pen));
	}
	return sc;
}

// ring entry: the RMQ priority -(f + 0.5*pen_gap*(x+y)) (lchain.c:288) is recomputed from f, x, y where it is needed
// (keeping it would cost 8 of 21 B per ring slot, and the sweep is bounded by LDS occupancy at large batches)
struct __attribute__((aligned(8))) CfEnt { int32_t y, x; };
__device__ __forceinline__ double cf_pri(int32_t f, int32_t x, int32_t y, float pen_gap) { return -((double)f + 0.5 * (double)pen_gap * (double)(x + y)); }

// One WAVE per segment.  Anchors are staged into LDS rings 64 at a time (one coalesced load per block, the next block
// is in flight while the current one is swept) and f/p leave through the rings once per block, so the per-anchor
// critical path touches LDS only.  Inside the sweep the wave is its own synchronisation domain (wavefront-scope
// fences: LDS operations of one wave execute in order).
// (the phase clocks of the profile cost ~600 clocks per anchor: they only exist in the PROF instantiation)
#define CF_CLK() (PROF ? (long long)clock64() : 0LL)
template <int CF_W, bool PROF>
__global__ __launch_bounds__(64)
void k_chain_fast(const u128 *__restrict__ a, const uint64_t *__restrict__ seg_start, const uint32_t *__restrict__ seg_order, uint32_t n_seg,
                  uint64_t n_total, const uint64_t *__restrict__ q_aoff, int n_seq, ChainParams P,
                  int32_t *__restrict__ f, int32_t *__restrict__ pp, uint32_t *__restrict__ seg_flag, const uint32_t *__restrict__ only_flagged, unsigned long long *__restrict__ prof)
{
	constexpr int CF_M = CF_W - 1;
	__shared__ CfEnt r_e[CF_W];
	__shared__ int32_t r_f[CF_W];
	__shared__ uint8_t s_sp[CF_W];
	__shared__ int32_t r_p[CF_WI], r_t[CF_WI];
	__shared__ int32_t s_sc[CF_WI], s_j[CF_WI];
	__shared__ uint8_t s_fl[CF_WI];
	const int lane = threadIdx.x;
	const uint32_t sidx = blockIdx.x;
	if (sidx >= n_seg) return;
	const uint32_t sg = seg_order[sidx];
	if (only_flagged && !only_flagged[sg]) return;           // second attempt (larger rings) of the segments the first one gave up on
	const uint64_t b = seg_start[sg], e = sg + 1 < n_seg ? seg_start[sg + 1] : n_total;
	const int32_t n = (int32_t)(e - b);
	const u128 *A = a + b;
	int32_t *F = f + b, *PP = pp + b;
	// query-local index of the segment's first anchor: the RMQ upper key is (y_i, 0) in QUERY numbering (lchain.c:310)
	int qlo = 0, qhi = n_seq;
	while (qlo < qhi) { int m = (qlo + qhi) >> 1; if (q_aoff[m + 1] <= b) qlo = m + 1; else qhi = m; }
	const bool seg_is_query_start = (q_aoff[qlo] == b);
	int32_t max_dist = P.max_dist, max_dist_inner = P.max_dist_inner;
	if (max_dist < P.bw) max_dist = P.bw;
	if (max_dist_inner <= 0 || max_dist_inner >= max_dist) max_dist_inner = 0;
	for (int k = lane; k < CF_WI; k += 64) r_t[k] = -1;
	int32_t st = 0, st_in = 0, i0 = 0;
	bool bail = false; int why = 0;   // why: 1 ring overflow, 2 tree-size cap, 3 tied minimum, 4 too many inner candidates
	double sm_pri = 1e300; int32_t sm_arg = -1, sm_blk = -1, sm_ymin = 0, sm_ymax = 0;   // lane b: summary of ring block b
	// Shortcut for the co-linear stretch: when the anchor just before i is a candidate (its x differs, it is inside the x and y
	// ranges) and its priority is STRICTLY below that of every anchor inserted before it (p_floor: a running minimum, refreshed
	// from the block summaries once per block so that it forgets evicted anchors), it is the unique range minimum whatever the
	// window holds -- the eviction search, the scans and the wave reduction are skipped and st is brought up to date later.
	double p_floor = 1e300, p_last = 1e300;
	// The inner scan (lchain.c:322-349) only ever changes (max_f, max_j), and a candidate j scores at most f[j] + span[j] (the
	// penalties only subtract).  fs_floor = max of f + span over the anchors below i-1 (kept like p_floor: running, rebuilt from
	// the block summaries), fs_last = that of anchor i-1: when no candidate other than the one already taken can beat max_f, the
	// scan is skipped.
	const int32_t FS_NONE = -(1 << 30);
	int32_t fs_floor = FS_NONE, fs_last = FS_NONE, sm_fs = FS_NONE;
	const bool shortcut_ok = P.cap >= CF_W;                  // (the tree-size cap of lchain.c:304 cannot bind inside the ring)
	const unsigned long long c0 = wall_clock64();
	unsigned long long n_scan = 0, n_inner = 0, n_incand = 0, n_slow = 0, n_pev = 0, n_py = 0;
	long long tk_scan = 0, tk_red = 0, tk_best = 0, tk_inner = 0, tk_store = 0;
	u128 nxtv; nxtv.x = 0, nxtv.y = 0;
	if (lane < n) nxtv = A[lane];
	for (int32_t blk = 0; blk < n && !bail; blk += 64) {
		// stage this block, start the load of the next one
		// lane l also keeps anchor blk+l in registers: the sweep reads the anchor it is at, and the candidates of its own block
		// (the ones whose f is youngest), with v_readlane instead of an LDS round trip
		const int32_t bx = (int32_t)nxtv.x, by = (int32_t)nxtv.y, bsp = (int32_t)(nxtv.y >> 32 & 0xff);
		int32_t bf = 0;
		if (blk + lane < n) { const int32_t s = (blk + lane) & CF_M; r_e[s].x = bx; r_e[s].y = by; s_sp[s] = (uint8_t)bsp; }
		if (blk + 64 + lane < n) nxtv = A[blk + 64 + lane];
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		{
			// bring the window start up to date for the block's first anchor (the shortcut leaves it behind) and rebuild p_floor
			// from the summaries of the blocks that still hold live anchors: every anchor below blk is in one of them
			const int32_t xb = __builtin_amdgcn_readlane(bx, 0);
			for (;;) {
				const int32_t j = st + lane;
				const bool keep = j >= i0 || !(xb - r_e[j & CF_M].x > max_dist);
				const unsigned long long m = __ballot(keep);
				if (m) { st += __ffsll((long long)m) - 1; break; }
				st += 64;
			}
			p_floor = wave_min_f64_key((lane < CF_W / 64 && sm_blk >= 0 && sm_blk + 64 > st) ? sm_pri : 1e300);
			fs_floor = wave_max_i32((lane < CF_W / 64 && sm_blk >= 0 && sm_blk + 64 > st) ? sm_fs : FS_NONE);
		}
		// the ring must hold [st, blk+128)
		if (blk + 128 - st > CF_W) { bail = true; why = 1; break; }
		const int32_t blk_end = blk + 64 < n ? blk + 64 : n;
		for (int32_t i = blk; i < blk_end; ++i) {
			const int il = i - blk;
			const int32_t xi = __builtin_amdgcn_readlane(bx, il), yi = __builtin_amdgcn_readlane(by, il), q_span = __builtin_amdgcn_readlane(bsp, il);
			int32_t max_f = q_span, max_j = -1;
			// Co-linear stretch inside the block, all at once: anchors that each follow their predecessor on the same diagonal within its
			// span (dr == dq <= span: "exact" in lchain.c:322, so no inner scan) chain to it with f = f_pred + dr -- the shortcut's
			// conditions hold for every one of them (priorities fall strictly along the stretch), so f is f_first + (x - x_first) and
			// the stretch is written out by its lanes
			if (shortcut_ok && il > 0 && i0 == i - 1 && p_last < p_floor) {
				const int32_t xpl = wave_shr1(bx, 0), ypl = wave_shr1(by, 0), spl = wave_shr1(bsp, 0);
				const int32_t dr = bx - xpl, dq = by - ypl;
				const int32_t f_l = __builtin_amdgcn_readlane(bf, il - 1) + (bx - __builtin_amdgcn_readlane(bx, il - 1));
				const bool c_ok = lane >= il && blk + lane < blk_end && dr > 0 && dr <= max_dist && dq == dr && dr <= spl && dr < max_dist && f_l > bsp;
				const unsigned long long m2 = __ballot(c_ok) >> il;
				const int R = ~m2 ? __ffsll((long long)~m2) - 1 : 64;
				if (R >= 2) {
					if (lane >= il && lane < il + R) { bf = f_l; r_f[(blk + lane) & CF_M] = f_l; r_p[(blk + lane) & (CF_WI - 1)] = blk + lane - 1; }
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
					const int la = il + R - 2, lb = il + R - 1;
					p_floor = cf_pri(__builtin_amdgcn_readlane(bf, la), __builtin_amdgcn_readlane(bx, la), __builtin_amdgcn_readlane(by, la), P.pen_gap);
					p_last = cf_pri(__builtin_amdgcn_readlane(bf, lb), __builtin_amdgcn_readlane(bx, lb), __builtin_amdgcn_readlane(by, lb), P.pen_gap);
					{
						const int32_t fsm = wave_max_i32((lane >= il && lane < il + R - 1) ? bf + bsp : FS_NONE);
						fs_floor = max(max(fs_floor, fs_last), fsm);
						fs_last = __builtin_amdgcn_readlane(bf, lb) + __builtin_amdgcn_readlane(bsp, lb);
					}
					if (PROF) n_py += R;
					i0 = i + R - 1;
					i += R - 1;
					continue;
				}
			}
			long long k0 = CF_CLK(), k1 = k0, k2 = k0;
			int32_t best_j = -1;
			bool shortcut = false;
			if (shortcut_ok && il > 0 && i0 == i - 1) {
				const int32_t xp = __builtin_amdgcn_readlane(bx, il - 1), yp = __builtin_amdgcn_readlane(by, il - 1);
				if (xp != xi && xi - xp <= max_dist && yp < yi && yp > yi - max_dist && p_last < p_floor) { shortcut = true; best_j = i - 1; i0 = i; if (PROF) ++n_py; }
			}
			if (il > 0 && p_last < p_floor) p_floor = p_last;    // from here on p_floor covers every anchor below i
			if (!shortcut) {
			// 1. late insertion (lchain.c:281-293): anchors that share x with i are not yet candidates; priorities were stored when f was known
			if (i0 < i) { const int32_t x0 = i0 >= blk ? __builtin_amdgcn_readlane(bx, i0 - blk) : r_e[i0 & CF_M].x; if (x0 != xi) i0 = i; }
			// 2+3. eviction (lchain.c:295-309) folded into the range-min scan: x ascends inside a segment, so "evicted" is the
			// predicate x_i - x_j > max_dist; st trails the true window start and catches up while scanning
			k0 = CF_CLK();
			// 2. eviction (lchain.c:295-309): x ascends inside a segment, so the window start is a forward search
			for (;;) {
				const int32_t j = st + lane;
				const bool keep = j >= i0 || !(xi - r_e[j & CF_M].x > max_dist);
				const unsigned long long m = __ballot(keep);
				if (m) { st += __ffsll((long long)m) - 1; break; }
				st += 64;
			}
			// 3. range-min over [st, i0) with keys in ((y_i-max_dist, +inf), (y_i, query anchor 0)]
			double best = 1e300; bool tie = false;
			{
				const int32_t y_lo = yi - max_dist;
				auto scan64 = [&](int32_t j0, int32_t jend) {
					++n_scan;
					const int32_t j = j0 + lane;
					if (j >= st && j < jend) {
						const CfEnt ej = r_e[j & CF_M];
						const bool in = ej.y > y_lo && (ej.y < yi || (ej.y == yi && seg_is_query_start && j == 0));
						if (in) { const double pj = cf_pri(r_f[j & CF_M], ej.x, ej.y, P.pen_gap); if (pj < best) best = pj, best_j = j, tie = false; else if (pj == best) tie = true; }
					}
				};
				// (a) completed 64-anchor blocks below i0 are represented by their summaries (lane b holds ring block b): a block
				//     wholly inside the window and the y range contributes its minimum, one wholly outside the y range nothing,
				//     the others are scanned anchor by anchor
				const int32_t S = blk < (i0 & ~63) ? blk : (i0 & ~63);
				bool part = false;
				if (lane < CF_W / 64 && sm_blk >= 0 && sm_blk + 64 > st && sm_blk + 64 <= S) {
					const bool none = sm_ymax <= y_lo || sm_ymin > yi || (sm_ymin == yi && !(seg_is_query_start && sm_blk == 0));
					// wholly inside the window -- or partly evicted, but the block's unique minimum is still inside: then it is also
					// the (unique) minimum of the surviving part
					if ((sm_blk >= st || sm_arg >= st) && sm_ymin > y_lo && sm_ymax < yi) { best = sm_pri; best_j = sm_arg < 0 ? sm_blk : sm_arg; tie = sm_arg < 0; }
					else if (!none) part = true;
				}
				unsigned long long pm = __ballot(part);
				while (pm) {
					const int bsel = __ffsll((long long)pm) - 1;
					pm &= pm - 1;
					const int32_t bb = __builtin_amdgcn_readlane(sm_blk, bsel);
					scan64(bb, bb + 64);
				}
				// (b) anchors not covered by a summary: the block being swept (from the registers), or the one i0 still lingers in
				if (i0 > blk) {
					++n_scan;
					const int32_t j = blk + lane;
					if (j >= st && j < i0) {
						const bool in = by > y_lo && (by < yi || (by == yi && seg_is_query_start && j == 0));
						if (in) { const double pj = cf_pri(bf, bx, by, P.pen_gap); if (pj < best) best = pj, best_j = j, tie = false; else if (pj == best) tie = true; }
					}
				} else for (int32_t tj = S > st ? S : st; tj < i0; tj += 64) scan64(tj, i0);
			}
			if (i0 - st > P.cap) { bail = true; why = 2; break; }             // size cap of the tree (lchain.c:304): not handled here
			k1 = CF_CLK();
			{
				// the wave minimum of a double, as two 32-bit DPP reductions over its order-preserving integer image
				const unsigned long long bits = (unsigned long long)__double_as_longlong(best);
				const unsigned long long key = (bits >> 63) ? ~bits : (bits | 0x8000000000000000ULL);
				const uint32_t khi = (uint32_t)(key >> 32), klo = (uint32_t)key;
				const uint32_t mhi = wave_min_u32(khi);
				const uint32_t mlo = wave_min_u32(khi == mhi ? klo : 0xffffffffu);
				const bool is_min = khi == mhi && klo == mlo;
				const unsigned long long who = __ballot(best_j >= 0 && is_min);
				if (who == 0) best_j = -1;
				else {
					if (__popcll(who) > 1 || __ballot(tie && is_min)) { bail = true; why = 3; break; }
					best_j = __builtin_amdgcn_readlane(best_j, __ffsll((long long)who) - 1);
				}
			}
			}
			k2 = CF_CLK();
			long long k3 = k2;
			if (best_j >= 0) {
				int32_t exact, width; const int32_t j = best_j;
				CfEnt ej; int32_t fj, spj;
				if (j >= blk) { const int jl = j - blk; ej.x = __builtin_amdgcn_readlane(bx, jl); ej.y = __builtin_amdgcn_readlane(by, jl); fj = __builtin_amdgcn_readlane(bf, jl); spj = __builtin_amdgcn_readlane(bsp, jl); }
				else { ej = r_e[j & CF_M]; fj = r_f[j & CF_M]; spj = s_sp[j & CF_M]; }
				int32_t sc = fj + score_pair32(xi, yi, ej.x, ej.y, spj, P.pen_gap, P.pen_skip, &exact, &width);
				if (width <= P.bw && sc > max_f) max_f = sc, max_j = j;
				k3 = CF_CLK();
				if (!exact && (j == i - 1 ? fs_floor : max(fs_floor, fs_last)) <= max_f) { exact = 1; if (PROF) ++n_pev; }   // nobody left who could do better
				if (!exact && max_dist_inner > 0 && yi > 0) {
					// inner window start (lchain.c:300-303), exact; it is only needed here, so it is brought up to date here
					if (st_in < st) st_in = st;
					{
						// x ascends: if the anchor CF_MAXIN+1 slots back already lies outside the inner window, so does everything before it
						// (after a long shortcut stretch st_in is far behind and the forward search would crawl there 64 slots at a time)
						const int32_t lo = i0 - (CF_MAXIN + 1);
						if (lo > st_in && xi - r_e[lo & CF_M].x > max_dist_inner) st_in = lo + 1;
					}
					for (;;) {
						const int32_t jj = st_in + lane;
						const bool keep = jj >= i0 || !(xi - r_e[jj & CF_M].x > max_dist_inner);
						const unsigned long long m = __ballot(keep);
						if (m) { st_in += __ffsll((long long)m) - 1; break; }
						st_in += 64;
					}
					if (st_in > i0) st_in = i0;
				}
				if (!exact && max_dist_inner > 0 && st_in < i0 && yi > 0) {
					// lchain.c:322-349: candidates of the inner window with y in [y_i - inner, y_i - 1], visited by descending (y, j)
					const int32_t n_in = i0 - st_in;
					if (n_in > CF_WI) { bail = true; why = 4; break; }
					if (n_in > CF_MAXIN) {
						// ---- crowded inner window (repeats): the same scan, 64 candidates at a time from LDS ----
						++n_inner; n_incand += n_in;
						auto cand = [&](int32_t jc, int32_t &sc, bool &ok) -> bool {    // candidate jc: inside the y range? its score, inside the band?
							const CfEnt ec = r_e[jc & CF_M];
							if (!(ec.y <= yi - 1 && ec.y >= yi - max_dist_inner)) { sc = 0; ok = false; return false; }
							int32_t wdt;
							sc = r_f[jc & CF_M] + score_pair32(xi, yi, ec.x, ec.y, s_sp[jc & CF_M], P.pen_gap, P.pen_skip, nullptr, &wdt);
							ok = wdt <= P.bw;
							return true;
						};
						bool unsorted = false;
						for (int32_t c0 = 0; c0 < n_in; c0 += 64) {           // stamps: a candidate inside the band marks its predecessor
							const int32_t jc = i0 - 1 - (c0 + lane);
							if (jc >= st_in) {
								if (jc + 1 < i0 && r_e[(jc + 1) & CF_M].y < r_e[jc & CF_M].y) unsorted = true;
								int32_t sc; bool ok;
								if (cand(jc, sc, ok) && ok) { const int32_t pj = r_p[jc & (CF_WI - 1)]; if (pj >= st_in) r_t[pj & (CF_WI - 1)] = i; }
							}
						}
						__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
						const bool uns = __ballot(unsorted) != 0;
						int32_t n_slots = n_in;
						if (uns) {
							// rank the valid candidates by descending (y, j) (all-pairs count) and lay them out in scan order
							++n_slow;
							n_slots = 0;
							for (int32_t c0 = 0; c0 < n_in; c0 += 64) {
								const int32_t jc = i0 - 1 - (c0 + lane);
								int32_t sc = 0; bool ok = false;
								const bool val = jc >= st_in && cand(jc, sc, ok);
								n_slots += __popcll(__ballot(val));
								if (val) {
									const int32_t yc = r_e[jc & CF_M].y;
									int32_t rank = 0;
									for (int32_t jo = st_in; jo < i0; ++jo) {
										const int32_t yo = r_e[jo & CF_M].y;
										const bool vo = yo <= yi - 1 && yo >= yi - max_dist_inner;
										rank += (vo && (yo > yc || (yo == yc && jo > jc))) ? 1 : 0;
									}
									s_sc[rank] = sc; s_j[rank] = jc;
									s_fl[rank] = (uint8_t)((ok ? 1 : 0) | (r_t[jc & (CF_WI - 1)] == i ? 2 : 0));
								}
							}
							__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
						}
						int32_t n_skip = 0; bool stop = false;
						for (int32_t c0 = 0; c0 < n_slots && !stop; c0 += 64) {
							int32_t v_sc = 0, v_j = -1; bool v_ok = false, v_mk = false;
							if (uns) {
								const int32_t c = c0 + lane;
								if (c < n_slots) { const int32_t fl = s_fl[c]; v_ok = fl & 1; v_mk = (fl & 2) != 0; v_sc = s_sc[c]; v_j = s_j[c]; }
							} else {
								const int32_t jc = i0 - 1 - (c0 + lane);
								if (jc >= st_in) { v_j = jc; const bool val = cand(jc, v_sc, v_ok); v_mk = val && r_t[jc & (CF_WI - 1)] == i; }
							}
							const int32_t v = v_ok ? v_sc : INT32_MIN;
							const int32_t incl = wave_prefix_max_incl(v);
							int32_t excl = __builtin_amdgcn_update_dpp(max_f, incl, 0x138, 0xf, 0xf, false);   // wave_shr:1, lane 0 keeps the running maximum
							if (excl < max_f) excl = max_f;
							const bool is_max = v_ok && v_sc > excl;
							const unsigned long long mx = __ballot(is_max), inc = __ballot(v_ok && !is_max && v_mk);
							unsigned long long rest = mx; int last = -1, pos = 0;
							for (;;) {
								const int c = rest ? __ffsll((long long)rest) - 1 : 64;
								const unsigned long long range = (c >= 64 ? ~0ULL : ((1ULL << c) - 1)) & ~((1ULL << pos) - 1);
								n_skip += __popcll(inc & range);
								if (n_skip > P.max_skip) { stop = true; break; }
								if (c >= 64) break;
								rest &= rest - 1;
								last = c; if (n_skip > 0) --n_skip;
								pos = c + 1;
								if (pos >= 64) break;
							}
							if (last >= 0) { max_f = __builtin_amdgcn_readlane(v_sc, last); max_j = __builtin_amdgcn_readlane(v_j, last); }
						}
					} else {
					++n_inner; n_incand += n_in;
					// lane l of group k looks at candidate jc = i0-1-(l+64k): if y does not decrease with the index anywhere in the
					// window (the co-linear case), descending (y, j) IS descending index, i.e. ascending (k, l)
					int32_t c_sc[CF_MAXIN / 64], c_j[CF_MAXIN / 64]; bool c_ok[CF_MAXIN / 64], c_val[CF_MAXIN / 64], c_mk[CF_MAXIN / 64];
					bool unsorted = false;
#pragma unroll
					for (int k = 0; k < CF_MAXIN / 64; ++k) {
						const int32_t jc = i0 - 1 - (lane + 64 * k);
						c_val[k] = false; c_ok[k] = false; c_mk[k] = false; c_sc[k] = 0; c_j[k] = jc;
						if (64 * k >= n_in) continue;                        // (uniform: most inner windows fill one or two groups)
						if (jc >= st_in) {
							const CfEnt ec = r_e[jc & CF_M];
							if (jc + 1 < i0 && r_e[(jc + 1) & CF_M].y < ec.y) unsorted = true;
							if (ec.y <= yi - 1 && ec.y >= yi - max_dist_inner) {
								int32_t wdt;
								c_val[k] = true;
								c_sc[k] = r_f[jc & CF_M] + score_pair32(xi, yi, ec.x, ec.y, s_sp[jc & CF_M], P.pen_gap, P.pen_skip, nullptr, &wdt);
								c_ok[k] = wdt <= P.bw;
								if (c_ok[k]) { const int32_t pj = r_p[jc & (CF_WI - 1)]; if (pj >= st_in) r_t[pj & (CF_WI - 1)] = i; }
							}
						}
					}
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
					for (int k = 0; k < CF_MAXIN / 64; ++k) if (64 * k < n_in) c_mk[k] = c_val[k] && r_t[c_j[k] & (CF_WI - 1)] == i;
					if (__ballot(unsorted)) {
						// general case: rank by descending (y, j) among valid candidates (all-pairs count), scatter, reload in scan order
						++n_slow;
						int32_t n_valid = 0;
#pragma unroll
						for (int k = 0; k < CF_MAXIN / 64; ++k) n_valid += __popcll(__ballot(c_val[k]));
#pragma unroll
						for (int k = 0; k < CF_MAXIN / 64; ++k) {
							if (c_val[k]) {
								const int32_t jc = c_j[k], yc = r_e[jc & CF_M].y;
								int32_t rank = 0;
#pragma unroll 8
								for (int32_t jo = st_in; jo < i0; ++jo) {
									const int32_t yo = r_e[jo & CF_M].y;
									const bool vo = yo <= yi - 1 && yo >= yi - max_dist_inner;
									rank += (vo && (yo > yc || (yo == yc && jo > jc))) ? 1 : 0;
								}
								s_sc[rank] = c_sc[k]; s_j[rank] = jc;
								s_fl[rank] = (uint8_t)((c_ok[k] ? 1 : 0) | (c_mk[k] ? 2 : 0));
							}
						}
						__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
						for (int k = 0; k < CF_MAXIN / 64; ++k) {
							const int32_t c = lane + 64 * k;
							c_val[k] = c < n_valid; c_ok[k] = false; c_mk[k] = false;
							if (c_val[k]) { const int32_t fl = s_fl[c]; c_ok[k] = fl & 1; c_mk[k] = (fl & 2) != 0; c_sc[k] = s_sc[c]; c_j[k] = s_j[c]; }
						}
						__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
					}
					// the scan itself (lchain.c:330-346): "sc > max_f" marks strict prefix maxima, n_skip is replayed over the events
					int32_t n_skip = 0; bool stop = false;
#pragma unroll
					for (int k = 0; k < CF_MAXIN / 64; ++k) {
						if (stop || 64 * k >= n_in) break;
						const int32_t v = c_ok[k] ? c_sc[k] : INT32_MIN;
						const int32_t incl = wave_prefix_max_incl(v);
						int32_t excl = __builtin_amdgcn_update_dpp(max_f, incl, 0x138, 0xf, 0xf, false);   // wave_shr:1, lane 0 keeps the running maximum
						if (excl < max_f) excl = max_f;
						const bool is_max = c_ok[k] && c_sc[k] > excl;
						const unsigned long long mx = __ballot(is_max), inc = __ballot(c_ok[k] && !is_max && c_mk[k]);
						// between two new maxima n_skip only grows (by the marked candidates in between); a new maximum takes one off
						unsigned long long rest = mx; int last = -1, pos = 0;
						for (;;) {
							const int c = rest ? __ffsll((long long)rest) - 1 : 64;
							const unsigned long long range = (c >= 64 ? ~0ULL : ((1ULL << c) - 1)) & ~((1ULL << pos) - 1);
							n_skip += __popcll(inc & range);
							if (n_skip > P.max_skip) { stop = true; break; }
							if (c >= 64) break;
							rest &= rest - 1;
							last = c; if (n_skip > 0) --n_skip;
							pos = c + 1;
							if (pos >= 64) break;
						}
						if (last >= 0) { max_f = __builtin_amdgcn_readlane(c_sc[k], last); max_j = __builtin_amdgcn_readlane(c_j[k], last); }
					}
					}
				}
			}
			const long long k4 = CF_CLK();
			if (lane == il) bf = max_f;
			p_last = cf_pri(max_f, xi, yi, P.pen_gap);
			fs_floor = max(fs_floor, fs_last); fs_last = max_f + q_span;
			if (lane == 0) {
				r_f[i & CF_M] = max_f; r_p[i & (CF_WI - 1)] = max_j;
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			const long long k5 = CF_CLK();
			tk_scan += k1 - k0; tk_red += k2 - k1; tk_best += k3 - k2; tk_inner += k4 - k3; tk_store += k5 - k4;
		}
		if (bail) break;
		if (blk + lane < n) { F[blk + lane] = bf; PP[blk + lane] = r_p[(blk + lane) & (CF_WI - 1)]; }
		if (blk + 64 <= n) {
			// summary of the finished block: minimum priority (and whether it is unique), y range
			CfEnt eb; eb.x = bx, eb.y = by;
			const double pb = cf_pri(bf, eb.x, eb.y, P.pen_gap);
			const double mp = wave_min_f64(pb);
			const unsigned long long who = __ballot(pb == mp);
			const int32_t ymin = wave_min_i32(eb.y), ymax = wave_max_i32(eb.y), fsb = wave_max_i32(bf + bsp);
			if (lane == ((blk >> 6) & (CF_W / 64 - 1))) {
				sm_fs = fsb;
				sm_pri = mp; sm_arg = __popcll(who) == 1 ? blk + (int32_t)(__ffsll((long long)who) - 1) : -1;
				sm_blk = blk; sm_ymin = ymin; sm_ymax = ymax;
			}
		}
	}
	if (lane == 0) seg_flag[sg] = bail ? (uint32_t)(why ? why : 1) : 0u;
	if (PROF && prof && lane == 0) {
		const unsigned long long dt = wall_clock64() - c0;
		atomicAdd(&prof[0], dt); atomicMax(&prof[1], dt); atomicMax(&prof[2], (unsigned long long)n);
		atomicAdd(&prof[12], n_pev); atomicAdd(&prof[13], n_py);
		atomicAdd(&prof[3], n_scan); atomicAdd(&prof[4], n_inner); atomicAdd(&prof[5], n_incand); atomicAdd(&prof[6], n_slow);
		atomicAdd(&prof[7], (unsigned long long)tk_scan); atomicAdd(&prof[8], (unsigned long long)tk_red); atomicAdd(&prof[9], (unsigned long long)tk_best); atomicAdd(&prof[10], (unsigned long long)tk_inner); atomicAdd(&prof[11], (unsigned long long)tk_store);
	}
}
#undef CF_CLK

// one lane per segment: the sweep of lchain.c:276-357 restricted to anchors [b,e) (tree empty at b)
__global__ void k_chain_segments(const u128 *__restrict__ a, const uint64_t *__restrict__ seg_start, const uint32_t *__restrict__ seg_order, uint32_t n_seg,
                                 uint64_t n_total, const uint64_t *__restrict__ q_aoff, int n_seq, const uint32_t *__restrict__ seg_flag,
                                 ChainParams P, CNode *__restrict__ nd_main, CNode *__restrict__ nd_inner,
                                 int32_t *__restrict__ f, int32_t *__restrict__ pp, int32_t *__restrict__ t)
{
	uint32_t sidx = blockIdx.x * blockDim.x + threadIdx.x;
	if (sidx >= n_seg) return;
	const uint32_t sg = seg_order[sidx];
	if (seg_flag && !seg_flag[sg]) return;                     // the fast path already chained this segment
	const uint64_t b = seg_start[sg], e = sg + 1 < n_seg ? seg_start[sg + 1] : n_total;
	// the RMQ upper key (y_i, 0) is in QUERY numbering (lchain.c:310): only the query's very first anchor may equal y_i
	int qlo = 0, qhi = n_seq;
	while (qlo < qhi) { int m = (qlo + qhi) >> 1; if (q_aoff[m + 1] <= b) qlo = m + 1; else qhi = m; }
	const int32_t hi_i = -(int32_t)(b - q_aoff[qlo]);
	const int32_t n = (int32_t)(e - b);
	const u128 *A = a + b;
	int32_t *F = f + b, *PP = pp + b, *T = t + b;            // PP holds segment-local predecessor indices (-1 none)
	Tree<true> Tm; Tm.nd = nd_main + b; Tm.root = -1;
	Tree<false> Ti; Ti.nd = nd_inner + b; Ti.root = -1;
	int32_t max_dist = P.max_dist, max_dist_inner = P.max_dist_inner;
	if (max_dist < P.bw) max_dist = P.bw;
	if (max_dist_inner <= 0 || max_dist_inner >= max_dist) max_dist_inner = 0;
	int32_t st = 0, st_inner = 0, i0 = 0;
	for (int32_t i = 0; i < n; ++i) {
		const u128 ai = A[i];
		int32_t max_j = -1, q_span = (int32_t)(ai.y >> 32 & 0xff), max_f = q_span;
		if (i0 < i && A[i0].x != ai.x) {
			for (int32_t j = i0; j < i; ++j) {
				const u128 aj = A[j];
				const double pri = -((double)F[j] + 0.5 * (double)P.pen_gap * (double)((int32_t)aj.x + (int32_t)aj.y));
				Tm.nd[j].y = (int32_t)aj.y, Tm.nd[j].pri = pri;
				Tm.insert(j);
				if (max_dist_inner > 0) { Ti.nd[j].y = (int32_t)aj.y; Ti.insert(j); }
			}
			i0 = i;
		}
		// anchors [st,i0) are in the tree; [i0,i) share x with anchor i and are not inserted yet
		while (st < i && (ai.x >> 32 != A[st].x >> 32 || ai.x > A[st].x + (uint64_t)max_dist || (int32_t)Tm.sz(Tm.root) > P.cap)) {
			if (st < i0) Tm.erase(st);
			++st;
		}
		if (max_dist_inner > 0) {
			while (st_inner < i && (ai.x >> 32 != A[st_inner].x >> 32 || ai.x > A[st_inner].x + (uint64_t)max_dist_inner || (int32_t)Ti.sz(Ti.root) > P.cap)) {
				if (st_inner < i0) Ti.erase(st_inner);
				++st_inner;
			}
		}
		const int32_t yi = (int32_t)ai.y;
		int32_t q = Tm.rmq(yi - max_dist, INT32_MAX, yi, hi_i);
		if (q >= 0) {
			int32_t sc, exact, width, n_skip = 0, j = q;
			sc = F[j] + score_pair(ai, A[j], P.pen_gap, P.pen_skip, &exact, &width);
			if (width <= P.bw && sc > max_f) max_f = sc, max_j = j;
			if (!exact && Ti.root >= 0 && yi > 0) {
				// largest key <= (yi-1, +inf), then walk in descending key order (krmq.h:97-109,306-340)
				Iter it; int32_t p = Ti.root, lower = -1;
				while (p >= 0) { if (yi - 1 < Ti.nd[p].y) p = Ti.nd[p].c[0]; else lower = p, p = Ti.nd[p].c[1]; }
				if (lower >= 0) {
					it.top = -1;
					for (p = Ti.root; p >= 0;) { int c = Ti.cmp(Ti.nd[lower].y, lower, p); it.stack[++it.top] = p; if (c < 0) p = Ti.nd[p].c[0]; else if (c > 0) p = Ti.nd[p].c[1]; else break; }
					for (;;) {
						const int32_t ej = it.stack[it.top];
						if (Ti.nd[ej].y < yi - max_dist_inner) break;
						j = ej;
						sc = F[j] + score_pair(ai, A[j], P.pen_gap, P.pen_skip, nullptr, &width);
						if (width <= P.bw) {
							if (sc > max_f) { max_f = sc, max_j = j; if (n_skip > 0) --n_skip; }
							else if (T[j] == i) { if (++n_skip > P.max_skip) break; }
							if (PP[j] >= 0) T[PP[j]] = i;
						}
						// predecessor in key order
						p = Ti.nd[it.stack[it.top]].c[0];
						if (p >= 0) { for (; p >= 0; p = Ti.nd[p].c[1]) it.stack[++it.top] = p; }
						else {
							do { p = it.stack[it.top--]; } while (it.top >= 0 && p == Ti.nd[it.stack[it.top]].c[0]);
							if (it.top < 0) break;
						}
					}
				}
			}
		}
		F[i] = max_f, PP[i] = max_j;
	}
}

// T[] marks inside a segment use segment-local anchor numbers as the reference uses global ones: `t[j] == i`
// only ever compares marks written during the same query, and marks never cross a segment (the inner tree is
// empty at a segment start), so the numbering is immaterial as long as it is injective within a segment...
// EXCEPT for the initial zeros (calloc, lchain.c:268): t[j]==0 matches i==0 only for the first anchor of the
// whole query, which has no predecessors.  Segment-local numbering would make anchor 0 of EVERY segment match
// zero-initialised marks, but that anchor has an empty tree, so no mark is ever read for it.

__global__ void k_seg_flags(const u128 *__restrict__ a, uint64_t n, int32_t max_dist, uint32_t *__restrict__ flag)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	if (i == 0) { flag[i] = 1; return; }
	if (flag[i]) return;                                      // query start, set by k_mark_query_starts
	const uint64_t x = a[i].x, px = a[i - 1].x;
	flag[i] = (x >> 32 != px >> 32 || x > px + (uint64_t)max_dist) ? 1u : 0u;
}
__global__ void k_mark_query_starts(const uint64_t *__restrict__ q_aoff, int n_seq, uint64_t n, uint32_t *__restrict__ flag)
{
	int q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q < n_seq) { uint64_t o = q_aoff[q]; if (o < n && q_aoff[q + 1] > o) flag[o] = 1; }
}
__global__ void k_seg_starts(const uint32_t *__restrict__ flag, const uint64_t *__restrict__ pos, uint64_t n, uint64_t *__restrict__ seg_start)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && flag[i]) seg_start[pos[i]] = i;
}
__global__ void k_seg_len(const uint64_t *__restrict__ seg_start, uint32_t n_seg, uint64_t n, uint32_t *__restrict__ neg_len, uint32_t *__restrict__ id)
{
	uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_seg) return;
	uint64_t e = s + 1 < n_seg ? seg_start[s + 1] : n;
	neg_len[s] = 0xffffffffu - (uint32_t)(e - seg_start[s]); id[s] = s;
}
// make predecessor indices query-local (the segment kernel wrote segment-local ones)
__global__ void k_fix_pred(const uint64_t *__restrict__ seg_start, uint32_t n_seg, uint64_t n, const uint64_t *__restrict__ q_aoff, int n_seq,
                           const uint32_t *__restrict__ seg_id_incl, int32_t *__restrict__ pp)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	int32_t p = pp[i];
	if (p < 0) return;
	const uint64_t sb = seg_start[seg_id_incl[i] - 1];
	int lo = 0, hi = n_seq;
	while (lo < hi) { int m = (lo + hi) >> 1; if (q_aoff[m + 1] <= i) lo = m + 1; else hi = m; }
	pp[i] = (int32_t)(sb + (uint64_t)p - q_aoff[lo]);
}

// The backtrack of lchain.c:27-111 for one query, by one WAVE.
//   * candidate list, mark reset, chain copies: streaming, 64 lanes;
//   * the unstable sort of the candidates: radix_sort_128x_wave (exact replay);
//   * the walks along p[] (mg_chain_bk_end, lchain.c:9-25, and the collection loop lchain.c:72-76): sequential by
//     definition, but a walk almost always steps to an anchor a few slots below, so the wave keeps p/f/t of 64
//     consecutive anchors in registers (one coalesced load) and follows the links with readlane -- memory latency
//     is paid once per window instead of once per step.  t[]==2 of the reference is never observable (p[i] < i, a
//     walk cannot meet itself), so one walk records the path, finds the cut (max_i) and marks only the kept part.

__global__ __launch_bounds__(64)
void k_bt_list(int n_seq, const uint64_t *__restrict__ q_aoff, const int32_t *__restrict__ f_all, int32_t *__restrict__ t_all, u128 *__restrict__ z_all, ChainParams P,
               int64_t *__restrict__ n_z_out, int32_t *__restrict__ n_u_out, int32_t *__restrict__ n_v_out)
{
	const int q = blockIdx.x, lane = threadIdx.x;
	if (q >= n_seq) return;
	const uint64_t b = q_aoff[q];
	const int64_t n = (int64_t)(q_aoff[q + 1] - b);
	if (lane == 0) n_u_out[q] = 0, n_v_out[q] = 0, n_z_out[q] = 0;
	if (n == 0) return;
	const int32_t *f = f_all + b; int32_t *t = t_all + b; u128 *z = z_all + b;
	// candidate ends in index order (order-preserving compaction), marks cleared
	// (one wave streams a whole-genome query's half a million scores: EIGHT windows of loads in flight per trip -- with one the loop was a load latency per 64
	// anchors, 4.7 ms of a leaf batch's chain stage; the windows are compacted in order, so the list is the same)
	int64_t n_z = 0;
	for (int64_t i0 = 0; i0 < n; i0 += 512) {
		int32_t fv[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) { const int64_t i = i0 + 64 * u + lane; fv[u] = i < n ? f[i] : 0; }
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int64_t i = i0 + 64 * u + lane;
			const bool keep = i < n && fv[u] >= P.min_sc;
			if (i < n) t[i] = 0;
			const unsigned long long m = __ballot(keep);
			if (keep) { const int64_t o = n_z + __popcll(m & ((1ULL << lane) - 1)); z[o].x = (uint64_t)fv[u]; z[o].y = (uint64_t)i; }
			n_z += __popcll(m);
		}
	}
	if (lane == 0) n_z_out[q] = n_z;
}

__global__ __launch_bounds__(64)
void k_bt_walk(int n_seq, const uint64_t *__restrict__ q_aoff, const u128 *__restrict__ a, const int32_t *__restrict__ f_all,
               const int32_t *__restrict__ p_all, int32_t *__restrict__ t_all, int32_t *__restrict__ v_all, const u128 *__restrict__ z_all, const int64_t *__restrict__ n_z_in,
               uint64_t *__restrict__ u_all, u128 *__restrict__ w_all, uint64_t *__restrict__ u2_all, u128 *__restrict__ out_all,
               ChainParams P, int32_t *__restrict__ n_u_out, int32_t *__restrict__ n_v_out, unsigned long long *__restrict__ prof, uint32_t *__restrict__ ev_out)
{
	const int q = blockIdx.x, lane = threadIdx.x;
	if (q >= n_seq) return;
	const uint64_t b = q_aoff[q];
	const int64_t n = (int64_t)(q_aoff[q + 1] - b);
	const int64_t n_z = n_z_in[q];
	if (ev_out && lane == 0) ev_out[q] = 0;
	if (n == 0 || n_z == 0) return;
	// ORDER EVENTS: the places where the order of candidates with EQUAL scores can change the result (see chain_all).  A mark carries the score of
	// the candidate whose chain set it; bit 0: a walk stopped at a mark of its own score, bit 1: a candidate was found marked by a chain of its own
	// score, bit 2: two chains start at the same target position (equal keys in compact_a's sort), bit 3: two chains were emitted by candidates of equal
	// score (the order in which compact_a's sort receives the chains is then the candidate sort's tie order), bit 4: bits 2 and 3 together in a way that
	// shows in the sorted chain list.
	uint32_t ev = 0; int32_t last_emit = -1;
	const unsigned long long c0 = wall_clock64(), c1 = c0;
	const u128 *A = a + b; const int32_t *f = f_all + b, *p = p_all + b;
	int32_t *t = t_all + b, *v = v_all + b;
	const u128 *z = z_all + b; u128 *w = w_all + b, *out = out_all + b;
	uint64_t *u = u_all + b, *u2 = u2_all + b;
	// ---- walks (every lane executes the same control flow) ----
	const int32_t max_drop = P.bw;
	int64_t n_v = 0; int32_t n_u = 0;
	unsigned long long pc_walks = 0, pc_reload = 0, pc_iter = 0, pc_trips = 0, pc_marktk = 0, pc_walktk = 0;      // (verbose: where a query's walk time goes)
	// Almost every candidate is an inner anchor of a chain that a better candidate has already walked over: a batch whose 64 marks are all
	// set needs nothing.  The candidates of the batch after next and the marks of the next batch are requested while this one is looked at
	// (two dependent loads, ~2 us, per batch otherwise); marks are only ever set, so a mark read early can only err towards "look again".
	auto load_z = [&](int64_t kb_, int32_t &f_, int32_t &i_) { const int64_t km = kb_ - 1 - lane; f_ = 0; i_ = -1; if (kb_ > 0 && km >= 0) { const u128 e = z[km]; f_ = (int32_t)e.x; i_ = (int32_t)e.y; } };
	// EIGHT batches per trip (four until round 6): their candidates and then their marks are eight independent loads per lane, so a trip costs about two
	// memory round trips for 512 candidates instead of one and a half for 64 (a whole-genome query holds 300 k candidates, nearly all of them marked)
	constexpr int NB = 8;
	int32_t zfA[NB], ziA[NB], tmA[NB], zfB[NB], ziB[NB];
#pragma unroll
	for (int s4 = 0; s4 < NB; ++s4) load_z(n_z - 64 * s4, zfA[s4], ziA[s4]);
#pragma unroll
	for (int s4 = 0; s4 < NB; ++s4) load_z(n_z - 64 * NB - 64 * s4, zfB[s4], ziB[s4]);
#pragma unroll
	for (int s4 = 0; s4 < NB; ++s4) { tmA[s4] = 1; if (ziA[s4] >= 0) tmA[s4] = t[ziA[s4]]; }
	for (int64_t kb4 = n_z; kb4 > 0; kb4 -= 64 * NB) {
		int32_t zfC[NB], ziC[NB], tmC[NB]; ++pc_trips;
#pragma unroll
		for (int s4 = 0; s4 < NB; ++s4) { zfC[s4] = zfA[s4]; ziC[s4] = ziA[s4]; tmC[s4] = tmA[s4]; zfA[s4] = zfB[s4]; ziA[s4] = ziB[s4]; }
#pragma unroll
		for (int s4 = 0; s4 < NB; ++s4) { tmA[s4] = 1; if (ziA[s4] >= 0) tmA[s4] = t[ziA[s4]]; }      // (early: re-read below whenever it says "unmarked")
#pragma unroll
		for (int s4 = 0; s4 < NB; ++s4) load_z(kb4 - 2 * 64 * NB - 64 * s4, zfB[s4], ziB[s4]);
#pragma unroll
	for (int s4 = 0; s4 < NB; ++s4) {
		const int64_t kb = kb4 - 64 * s4;
		if (kb <= 0) break;
		// a batch of 64 candidates, highest rank in lane 0
		const int32_t zf = zfC[s4], zi = ziC[s4], tm = tmC[s4];
		unsigned long long todo = __ballot(zi >= 0 && tm == 0);
		if (__ballot(zi >= 0 && tm != 0 && tm == zf)) ev |= 2;
		while (todo) {
			// marks of the remaining candidates as of now (the previous chain may have covered some of them)
			const int32_t tnow = zi >= 0 && ((todo >> lane) & 1) ? t[zi] : 0;
			const bool open = zi >= 0 && ((todo >> lane) & 1) && tnow == 0;
			if (__ballot(tnow != 0 && tnow == zf)) ev |= 2;
			todo = __ballot(open);
			if (!todo) break;
			const int src = __ffsll((long long)todo) - 1;
			todo &= todo - 1;
			const int32_t e0 = rl(zi, src), zx = rl(zf, src); ++pc_walks; const unsigned long long wk0 = prof ? wall_clock64() : 0;
			// walk
			int32_t wb = e0 - 63; if (wb < 0) wb = 0;
			int32_t wp = -1, wf = 0, wt = 1;
			{ const int32_t idx = wb + lane; if (idx <= e0) wp = p[idx], wf = f[idx], wt = t[idx]; }
			int32_t cur = e0, m = 0, kept = 0, max_s = 0;
			const int64_t n_v0 = n_v;
			for (;;) {
				++pc_iter;
				// Co-linear stretch inside the window, all at once: while the path steps to the anchor just below (p[j] == j-1), that
				// anchor is unused (t == 0) and scores less (f[j-1] < f[j]), every step is a new maximum of z.x - f -- provided the
				// last step was one too (kept == m; at the start both sides are 0) -- so nothing can break the walk and `run` steps
				// collapse into an update of (cur, m, kept, max_s) and one coalesced store of the path
				if (kept == m) {
					const int32_t c = cur - wb, idx = wb + lane;
					const int32_t f_lo = wave_shr1(wf, 0), t_lo = wave_shr1(wt, 1);
					const bool ok = lane > 0 && lane <= c && wp == idx - 1 && t_lo == 0 && f_lo < wf;
					const unsigned long long mk = __ballot(ok);
					const unsigned long long sh = mk << (63 - c);                 // lane c -> bit 63
					const int run = ~sh ? __clzll((long long)~sh) : 64;
					if (run > 0) {
						if (lane < run) v[n_v0 + m + lane] = cur - lane;
						cur -= run; m += run; kept = m;
						max_s = zx - rl(wf, cur - wb);
						continue;
					}
				}
				const int32_t nxt = rl(wp, cur - wb);
				if (lane == 0) v[n_v0 + m] = cur;
				++m;
				int32_t sv, tn = 1;
				if (nxt < 0) sv = zx;
				else {
					if (nxt < wb) {
						++pc_reload;
						wb = nxt - 63; if (wb < 0) wb = 0;
						const int32_t idx = wb + lane;
						wp = -1, wf = 0, wt = 1;
						if (idx <= nxt) wp = p[idx], wf = f[idx], wt = t[idx];
					}
					sv = zx - rl(wf, nxt - wb);
					tn = rl(wt, nxt - wb);
				}
				if (sv > max_s) max_s = sv, kept = m;
				else if (max_s - sv > max_drop) break;
				if (nxt < 0 || tn != 0) { if (nxt >= 0 && tn == zx) ev |= 1; break; }
				cur = nxt;
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");              // the path was stored by whichever lanes held it
			const unsigned long long wk1 = prof ? wall_clock64() : 0;
			// kept part: the first `kept` path elements (the walk stops before max_i, lchain.c:72); marks stay even if the chain is dropped
			for (int32_t c = lane; c < kept; c += 64) t[v[n_v0 + c]] = zx;          // (nonzero: min_sc > 0) the claimant's score
			if (prof) { const unsigned long long wk2 = wall_clock64(); pc_walktk += wk1 - wk0; pc_marktk += wk2 - wk1; }
			// score of the chain: z.x - f[max_i]; max_i is the path element number `kept` (or -1 past the root)
			const int32_t sc = max_s;
			if (kept > 0 && sc >= P.min_sc && kept >= P.min_cnt) { if (lane == 0) u[n_u] = (uint64_t)sc << 32 | (uint64_t)kept, u2[n_u] = (uint64_t)zx; ++n_u; n_v += kept; if (zx == last_emit) ev |= 8; last_emit = zx; }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		}
	}
	}
	if (lane == 0) n_u_out[q] = n_u, n_v_out[q] = (int32_t)n_v;
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	const unsigned long long c2 = wall_clock64();
	// chains are ordered by the target position of their first anchor (compact_a, lchain.c:96-99)
	if (lane == 0) {
		int64_t kk = 0;
		for (int32_t i = 0; i < n_u; ++i) { const int32_t ni = (int32_t)u[i]; w[i].x = A[v[kk + ni - 1]].x; w[i].y = (uint64_t)kk << 32 | (uint64_t)i; kk += ni; }
		if (n_u > 0) { uint32_t h2[256], t2[256]; radix_sort_128x_exact(w, w + n_u, h2, t2); }
		// equal keys: up to 64 chains the reference's sort is an insertion sort (ksort.h:107-117, 140), i.e. stable -- two chains of equal key stand in
		// emission order, which is the descending order of their candidates' scores unless those are equal too (u2 still holds them)
		for (int32_t i = 1; i < n_u; ++i) if (w[i].x == w[i - 1].x) { ev |= 4; if (u2[(int32_t)w[i].y] == u2[(int32_t)w[i - 1].y]) ev |= 16; }
		if (n_u > 64 && (ev & 4) && (ev & 8)) ev |= 16;                        // a cycle-leader pass: any two chains emitted in tie order can move the equal keys
		// output offsets of the chains in their final order, stashed in w.x next to the chain word
		kk = 0;
		for (int32_t i = 0; i < n_u; ++i) { const int32_t j = (int32_t)w[i].y; u2[i] = u[j]; w[i].x = (uint64_t)kk; kk += (int32_t)u[j]; }
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	// (the chains themselves are copied to their slots by k_bt_copy, a thread per anchor: one wave going through a query's chains one after the
	// other paid two dependent loads per chain -- 3 ms for the 1 500 chains of a whole-genome query)
	for (int32_t c = lane; c < n_u; c += 64) u[c] = u2[c];
	if (ev_out && lane == 0) ev_out[q] = ev;
	if (prof && lane == 0) {
		const unsigned long long c3 = wall_clock64();
		atomicAdd(&prof[0], c1 - c0); atomicAdd(&prof[1], c2 - c1); atomicAdd(&prof[2], c3 - c2);
		atomicMax(&prof[3], c1 - c0); atomicMax(&prof[4], c2 - c1); atomicMax(&prof[5], c3 - c2);
		// of the query with the most anchors: walks, window reloads, walk iterations, candidate trips, ticks inside walks, ticks setting marks
		if ((unsigned long long)n >= atomicMax(&prof[6], (unsigned long long)n)) { prof[7] = pc_walks; prof[8] = pc_reload; prof[9] = pc_iter; prof[10] = pc_trips; prof[11] = pc_walktk; prof[12] = pc_marktk; }
	}
}

// Every kept anchor to its place in the compacted list: slot p of query q belongs to the chain c with w[c].x <= p < w[c].x + count (the chains'
// offsets in their final order, left there by k_bt_walk) and takes the chain's anchors in ascending order (the walk collected them backwards).
__global__ void k_bt_copy(int n_seq, const uint64_t *__restrict__ q_aoff, uint64_t n_a, const u128 *__restrict__ a, const int32_t *__restrict__ v_all, const uint64_t *__restrict__ u_all,
                          const u128 *__restrict__ w_all, const int32_t *__restrict__ n_u_all, const int32_t *__restrict__ n_v_all, u128 *__restrict__ out_all)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_a) return;
	int lo = 0, hi = n_seq;
	while (lo < hi) { const int m = (lo + hi) >> 1; if (q_aoff[m + 1] <= i) lo = m + 1; else hi = m; }
	const int q = lo;
	const uint64_t b = q_aoff[q];
	const int64_t p = (int64_t)(i - b);
	if (p >= (int64_t)n_v_all[q]) return;
	const u128 *w = w_all + b;
	int cl = 0, ch = n_u_all[q];                                // the last chain whose offset is <= p
	while (ch - cl > 1) { const int m = (cl + ch) >> 1; if ((int64_t)w[m].x <= p) cl = m; else ch = m; }
	const int64_t src0 = (int64_t)(w[cl].y >> 32), dst0 = (int64_t)w[cl].x;
	const int32_t ni = (int32_t)u_all[b + cl];
	out_all[b + p] = a[b + (uint64_t)v_all[b + src0 + (ni - (p - dst0) - 1)]];
}

// ---- a hint for the replay of the candidate sort (lchain.c:52: radix_sort_128x(z, z + n_z)) ------------------------------------------------
// A stable device sort of the candidates by (query, score) tells which buckets of the reference's unstable sort hold no two equal scores:
// their final order is the sorted order, whatever the sort did on the way (pga_sort_big.h).  Scores rise along a chain, so almost every bucket
// below the top level is one.  Slots behind a query's n_z candidates sort to the end of its segment (score field all ones).
__global__ void k_z_keys(const u128 *__restrict__ z, const uint64_t *__restrict__ q_aoff, const int64_t *__restrict__ n_z, int n_seq, uint64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ idx)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	int lo = 0, hi = n_seq;
	while (lo < hi) { int m = (lo + hi) >> 1; if (q_aoff[m + 1] <= i) lo = m + 1; else hi = m; }
	const bool valid = (int64_t)(i - q_aoff[lo]) < n_z[lo];
	key[i] = (uint64_t)lo << 32 | (valid ? (uint64_t)(uint32_t)z[i].x : 0xffffffffULL);
	idx[i] = (uint32_t)i;
}
__global__ void k_z_sorted(const u128 *__restrict__ z, const uint64_t *__restrict__ key, const uint32_t *__restrict__ idx, uint64_t n, const uint64_t *__restrict__ q_aoff, const int64_t *__restrict__ n_z,
                           uint64_t *__restrict__ sx, uint64_t *__restrict__ sy, uint32_t *__restrict__ q_tie)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t k = key[i];
	if ((uint32_t)k == 0xffffffffu) { sx[i] = sy[i] = 0; return; }
	sx[i] = (uint64_t)(uint32_t)k; sy[i] = z[idx[i]].y;
	if (i > 0 && key[i - 1] == k) q_tie[k >> 32] = 1;                    // same query, same score
}
// queries without equal scores: the sorted order is the answer
__global__ void k_z_take_sorted(u128 *__restrict__ z, const uint64_t *__restrict__ sx, const uint64_t *__restrict__ sy, const uint64_t *__restrict__ q_aoff, const int64_t *__restrict__ n_z, int n_seq,
                                const uint32_t *__restrict__ q_tie, uint64_t n)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	int lo = 0, hi = n_seq;
	while (lo < hi) { int m = (lo + hi) >> 1; if (q_aoff[m + 1] <= i) lo = m + 1; else hi = m; }
	if ((q_tie && q_tie[lo]) || (int64_t)(i - q_aoff[lo]) >= n_z[lo]) return;
	u128 v; v.x = sx[i]; v.y = sy[i]; z[i] = v;
}

__global__ void k_gather_chains(const uint64_t *__restrict__ u, const uint64_t *__restrict__ q_aoff, const uint64_t *__restrict__ coff, int n_seq, uint64_t *__restrict__ out)
{
	const int q = blockIdx.x;
	if (q >= n_seq) return;
	const uint64_t b = q_aoff[q], o = coff[q], n = coff[q + 1] - o;
	for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) out[o + i] = u[b + i];
}

// ---- which queries need the reference's tie order (speculative mode of chain_core) ----
__global__ void k_seg_query_flag(const uint32_t *__restrict__ seg_flag, const uint64_t *__restrict__ seg_start, uint32_t n_seg, const uint64_t *__restrict__ q_aoff, int n_seq, uint32_t *__restrict__ out)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_seg || !seg_flag[s]) return;
	const uint64_t i = seg_start[s];
	int lo = 0, hi = n_seq;
	while (lo < hi) { int m = (lo + hi) >> 1; if (q_aoff[m + 1] <= i) lo = m + 1; else hi = m; }
	out[lo] = 1;
}
__global__ void k_need_exact(int n_seq, const uint32_t *__restrict__ q_tie_x, const uint32_t *__restrict__ q_tie_f, const uint32_t *__restrict__ ev, const uint32_t *__restrict__ seg_q, uint32_t *__restrict__ need)
{
	const int q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n_seq) return;
	const uint32_t e = ev[q];
	const bool order_f = q_tie_f[q] && (e & (1u | 2u | 16u));                      // equal scores whose order shows
	const bool order_x = q_tie_x && q_tie_x[q] && seg_q && seg_q[q];               // equal anchor keys under the tree re-enactment (its shape is the insertion order)
	need[q] = order_f || order_x ? 1u : 0u;
}
__global__ void k_gather_queries(int n_sub, const uint64_t *__restrict__ src_off, const uint64_t *__restrict__ dst_off, const u128 *__restrict__ a, u128 *__restrict__ out)
{
	const int q = blockIdx.x;
	if (q >= n_sub) return;
	const uint64_t s = src_off[q], d = dst_off[q], n = dst_off[q + 1] - d;
	for (uint64_t i = (uint64_t)blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.y * blockDim.x) out[d + i] = a[s + i];
}

// One pass of the stage over the anchors a[0..n_a) of n_seq queries.
//   spec == false  the reference's procedure: the candidate ends of every query go through the replay of the unstable sort (lchain.c:52) where they hold
//                  equal scores; `a` must be in the reference's order.
//   spec == true   `a` may hold equal keys in stable order and the candidates are taken in STABLE order of their scores.  That is the reference's result
//                  whenever no ORDER EVENT shows (k_bt_walk) -- proof in chain_all -- and need_out[q] says for which queries one did.
static void chain_core(const DBuf<u128> &a, const DBuf<uint64_t> &q_aoff, const int n_seq, uint64_t n_a, const mm_mapopt_t &opt, int k, ChainResult &O, hipStream_t st, Timers *tm,
                       bool spec, const uint32_t *d_q_tie_x, std::vector<uint32_t> *need_out)
{
	if (need_out) need_out->assign((size_t)n_seq, 0u);
	O.n_u.assign((size_t)n_seq, 0); O.n_v.assign((size_t)n_seq, 0); O.u.clear(); O.a.clear(); O.d_a.release();
	if (n_a == 0) return;
	if (n_a >= (1ULL << 31)) throw std::runtime_error("pga: more than 2^31 anchors in one batch");
	ChainParams P;
	P.max_dist = opt.max_gap, P.max_dist_inner = opt.rmq_inner_dist, P.bw = opt.bw, P.max_skip = opt.max_chain_skip, P.cap = opt.rmq_size_cap;
	P.min_cnt = opt.min_cnt, P.min_sc = opt.min_chain_score;
	P.pen_gap = (float)(opt.chain_gap_scale * 0.01 * k);     // map.c:273: float*double*int evaluated in double, stored to float
	P.pen_skip = (float)(opt.chain_skip_scale * 0.01 * k);
	int32_t seg_dist = P.max_dist < P.bw ? P.bw : P.max_dist;
	const unsigned nba = (unsigned)((n_a + 255) / 256);
	// PGA_VERBOSE: host wall clock of the stage's steps (each mark drains the stream)
	const bool vmarks = getenv("PGA_VERBOSE") != nullptr;
	auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double t_mark = vmarks ? wall() : 0.0; std::string marks;
	auto mark = [&](const char *what) { if (!vmarks) return; (void)sync_stream(st); const double t = wall(); char b[96]; snprintf(b, sizeof b, " %s %.1f", what, (t - t_mark) * 1e3); marks += b; t_mark = t; };
	// segments
	DBuf<uint32_t> flag(n_a + 1); flag.zero(st);
	hipLaunchKernelGGL(k_mark_query_starts, dim3((unsigned)((n_seq + 255) / 256)), dim3(256), 0, st, q_aoff.p, n_seq, n_a, flag.p);
	hipLaunchKernelGGL(k_seg_flags, dim3(nba), dim3(256), 0, st, a.p, n_a, seg_dist, flag.p);
	DBuf<uint64_t> pos(n_a + 1);
	{
		size_t tb = 0;
		auto it = rocprim::make_transform_iterator(flag.p, [] __device__ (uint32_t v) { return (uint64_t)v; });
		PGA_HIP(rocprim::exclusive_scan(nullptr, tb, it, pos.p, (uint64_t)0, n_a + 1, rocprim::plus<uint64_t>(), st));
		DBuf<uint8_t> tmp(tb ? tb : 1);
		PGA_HIP(rocprim::exclusive_scan(tmp.p, tb, it, pos.p, (uint64_t)0, n_a + 1, rocprim::plus<uint64_t>(), st));
	}
	uint64_t n_seg64 = 0;
	PGA_HIP(hipMemcpyAsync(&n_seg64, pos.p + n_a, 8, hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
	const uint32_t n_seg = (uint32_t)n_seg64;
	DBuf<uint64_t> seg_start(n_seg);
	hipLaunchKernelGGL(k_seg_starts, dim3(nba), dim3(256), 0, st, flag.p, pos.p, n_a, seg_start.p);
	// inclusive segment id per anchor (for the predecessor fix-up): incl = excl + flag
	DBuf<uint32_t> seg_incl(n_a);
	{
		struct Op { const uint32_t *flag; const uint64_t *pos; uint32_t *out; };
		Op op{flag.p, pos.p, seg_incl.p};
		PGA_HIP(rocprim::transform(rocprim::make_counting_iterator<uint64_t>(0), rocprim::make_discard_iterator(), n_a,
		        [op] __device__ (uint64_t i) { op.out[i] = (uint32_t)(op.pos[i] + op.flag[i]); return 0; }, st));
	}
	// longest segments first, so that the lanes of a wave carry similar work
	DBuf<uint32_t> neg_len(n_seg), ord0(n_seg), neg_len2(n_seg), ord(n_seg);
	hipLaunchKernelGGL(k_seg_len, dim3((n_seg + 255) / 256), dim3(256), 0, st, seg_start.p, n_seg, n_a, neg_len.p, ord0.p);
	// (the default since round 6: sorts of at most WGS_CAP pairs in ONE launch of one workgroup, pga_wg_sort.h, where rocPRIM issues a block sort and up to ten
	// merge passes of two kernels each; all 1998 calls of the BASELINE build keep their digests; PGA_WG_SORT=0: rocPRIM)
	static const bool wg_sort = !(getenv("PGA_WG_SORT") && getenv("PGA_WG_SORT")[0] == '0');
	if (wg_sort && n_seg > 0 && n_seg <= WGS_CAP)
		hipLaunchKernelGGL((k_wg_sort_pairs<uint32_t>), dim3(1), dim3(WGS_NT), 0, st, neg_len.p, neg_len2.p, ord0.p, ord.p, n_seg, 32);
	else {
		size_t tb = 0;
		PGA_HIP(rocprim::radix_sort_pairs(nullptr, tb, neg_len.p, neg_len2.p, ord0.p, ord.p, n_seg, 0, 32, st));
		DBuf<uint8_t> tmp(tb ? tb : 1);
		PGA_HIP(rocprim::radix_sort_pairs(tmp.p, tb, neg_len.p, neg_len2.p, ord0.p, ord.p, n_seg, 0, 32, st));
	}
	mark("segments");
	DBuf<CNode> nd_main, nd_inner;                        // the tree re-enactment's nodes (64 B per anchor): only when a segment needs it
	DBuf<uint32_t> seg_q;                                 // queries with a segment the fast kernel handed to the tree re-enactment
	DBuf<int32_t> f(n_a), pp(n_a), t(n_a), v(n_a);
	t.zero(st);
	mark("buffers");
	{
		DBuf<uint32_t> seg_flag(n_seg);
		seg_q.alloc((size_t)n_seq); seg_q.zero(st);
		EventTimer et(st);
		const bool use_fast = !getenv("PGA_CHAIN_EXACT_ONLY");
		// (A/B mode without the fast kernel: EVERY segment runs the tree re-enactment, so every query counts as "has such a segment" for k_need_exact)
		if (!use_fast && n_seq > 0) PGA_HIP(hipMemsetD32Async((hipDeviceptr_t)seg_q.p, 1, (size_t)n_seq, st));
		const bool verbose = getenv("PGA_VERBOSE") != nullptr;
		const bool prof_on = verbose && getenv("PGA_CHAIN_PROF");
		DBuf<unsigned long long> cprof(16); if (prof_on) cprof.zero(st);
		if (use_fast && prof_on) hipLaunchKernelGGL((k_chain_fast<2048, true>), dim3(n_seg), dim3(64), 0, st, a.p, seg_start.p, ord.p, n_seg, n_a, q_aoff.p, n_seq, P, f.p, pp.p, seg_flag.p, (const uint32_t*)nullptr, cprof.p);
		else if (use_fast) {
			hipLaunchKernelGGL((k_chain_fast<2048, false>), dim3(n_seg), dim3(64), 0, st, a.p, seg_start.p, ord.p, n_seg, n_a, q_aoff.p, n_seq, P, f.p, pp.p, seg_flag.p, (const uint32_t*)nullptr, (unsigned long long*)nullptr);
		}
		const double ms_fast = verbose ? et.stop() : 0.0;
		// segments the fast kernel gave up on (rare: ring overflow, a tied minimum, a crowded inner window) are re-run by the tree kernel
		bool any_flagged = !use_fast;
		if (use_fast) {
			DBuf<uint32_t> n_fl(1);
			struct Fl { const uint32_t *f; };
			Fl fl{seg_flag.p};
			auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0), [fl] __device__ (uint32_t i) { return (uint32_t)(fl.f[i] != 0); });
			size_t tb = 0;
			PGA_HIP(rocprim::reduce(nullptr, tb, it, n_fl.p, 0u, (size_t)n_seg, rocprim::plus<uint32_t>(), st));
			DBuf<uint8_t> tmp(tb ? tb : 1);
			PGA_HIP(rocprim::reduce(tmp.p, tb, it, n_fl.p, 0u, (size_t)n_seg, rocprim::plus<uint32_t>(), st));
			any_flagged = n_fl.download(st)[0] != 0;
		}
		if (any_flagged) {
			if (use_fast) hipLaunchKernelGGL(k_seg_query_flag, dim3((n_seg + 255) / 256), dim3(256), 0, st, seg_flag.p, seg_start.p, n_seg, q_aoff.p, n_seq, seg_q.p);
			nd_main.alloc(n_a); nd_inner.alloc(n_a);
			hipLaunchKernelGGL(k_chain_segments, dim3((n_seg + 63) / 64), dim3(64), 0, st, a.p, seg_start.p, ord.p, n_seg, n_a, q_aoff.p, n_seq,
			                   use_fast ? seg_flag.p : (const uint32_t*)nullptr, P, nd_main.p, nd_inner.p, f.p, pp.p, t.p);
		}
		PGA_HIP(hipGetLastError());
		const double ms = et.stop(K_CHAIN);
		if (getenv("PGA_VERBOSE") && use_fast) {
			std::vector<uint32_t> fl = seg_flag.download(st); size_t nf = 0, why[5] = {0, 0, 0, 0, 0}; for (uint32_t v : fl) { nf += v != 0; ++why[v < 5 ? v : 1]; }
			fprintf(stderr, "[pga]   chain: %u segments, %zu re-run by the tree kernel (ring overflow %zu, size cap %zu, tied minimum %zu, inner candidates %zu), %.3f ms (fast kernel %.3f ms)\n", n_seg, nf, why[1], why[2], why[3], why[4], ms, ms_fast);
			if (nf) { std::vector<uint64_t> ss = seg_start.download(st); for (uint32_t s = 0; s < n_seg; ++s) if (fl[s]) fprintf(stderr, "[pga]     segment %u: %llu anchors, reason %u\n", s, (unsigned long long)((s + 1 < n_seg ? ss[s + 1] : n_a) - ss[s]), fl[s]); }
			std::vector<unsigned long long> pr = cprof.download(st);
			if (prof_on) fprintf(stderr, "[pga]   chain fast: longest segment %llu anchors, slowest %.2f ms, sum %.1f ms; scan iterations %.2f/anchor, inner scans %.3f/anchor with %.1f candidates, %llu unsorted\n",
			        pr[2], pr[1] * 1e-5, pr[0] * 1e-5, (double)pr[3] / (double)n_a, (double)pr[4] / (double)n_a, pr[4] ? (double)pr[5] / (double)pr[4] : 0.0, pr[6]);
			if (prof_on) fprintf(stderr, "[pga]   chain fast clocks/anchor: scan %.0f reduce %.0f best %.0f inner %.0f store %.0f; inner scans skipped by the f + span bound %.3f/anchor; shortcut taken %.3f\n", (double)pr[7] / n_a, (double)pr[8] / n_a, (double)pr[9] / n_a, (double)pr[10] / n_a, (double)pr[11] / n_a, (double)pr[12] / n_a, (double)pr[13] / n_a);
		}
		if (tm) { tm->kern[K_CHAIN].ms += ms; tm->kern[K_CHAIN].launches += 1; tm->kern[K_CHAIN].alg_bytes += 36.0 * (double)n_a; } // 16 B anchor read + f,p,v,t
	}
	hipLaunchKernelGGL(k_fix_pred, dim3(nba), dim3(256), 0, st, seg_start.p, n_seg, n_a, q_aoff.p, n_seq, seg_incl.p, pp.p);
	mark("chain kernels");
	// backtrack + compact, one lane per query
	DBuf<u128> z(n_a), w(n_a), out(n_a);
	DBuf<uint64_t> u(n_a), u2(n_a);
	DBuf<int32_t> n_u((size_t)n_seq), n_v((size_t)n_seq);
	{
		EventTimer et(st);
		const bool verbose = getenv("PGA_VERBOSE") != nullptr;
		DBuf<unsigned long long> prof(16); if (verbose) prof.zero(st);
		DBuf<int64_t> n_z((size_t)n_seq);
		DBuf<uint32_t> ev((size_t)n_seq), q_tie_f;
		hipLaunchKernelGGL(k_bt_list, dim3((unsigned)n_seq), dim3(64), 0, st, n_seq, q_aoff.p, f.p, t.p, z.p, P, n_z.p, n_u.p, n_v.p);
		double ms_list = 0, ms_sort = 0;
		et.mark();                                                            // (the three intervals of the backtrack are read behind the walk: nothing here waits for the device)
		EventTimer et2(st);
		static const bool z_hint = getenv("PGA_NO_Z_HINT") == nullptr;
		if (!z_hint && !spec) replay_sort_segments(z.p, n_a, q_aoff.p, n_z.p, n_seq, nullptr, st, tm);
		else {
			DBuf<uint64_t> key0(n_a), key1(n_a), sx(n_a), sy(n_a);
			DBuf<uint32_t> idx0(n_a), idx1(n_a), dupc(n_a);
			DBuf<uint32_t> &q_tie = q_tie_f; q_tie.alloc((size_t)n_seq);
			q_tie.zero(st);
			hipLaunchKernelGGL(k_z_keys, dim3(nba), dim3(256), 0, st, z.p, q_aoff.p, n_z.p, n_seq, n_a, key0.p, idx0.p);
			int bits = 1; while ((1LL << bits) < n_seq) ++bits;
			if (wg_sort && n_a <= WGS_CAP)
				hipLaunchKernelGGL((k_wg_sort_pairs<uint64_t>), dim3(1), dim3(WGS_NT), 0, st, key0.p, key1.p, idx0.p, idx1.p, (uint32_t)n_a, 32 + bits);
			else {
			size_t tb = 0;
			PGA_HIP(rocprim::radix_sort_pairs(nullptr, tb, key0.p, key1.p, idx0.p, idx1.p, n_a, 0, 32 + bits, st));
			DBuf<uint8_t> tmp(tb ? tb : 1);
			PGA_HIP(rocprim::radix_sort_pairs(tmp.p, tb, key0.p, key1.p, idx0.p, idx1.p, n_a, 0, 32 + bits, st));
			}
			hipLaunchKernelGGL(k_z_sorted, dim3(nba), dim3(256), 0, st, z.p, key1.p, idx1.p, n_a, q_aoff.p, n_z.p, sx.p, sy.p, q_tie.p);
			{
				struct Dp { const uint64_t *k; };
				Dp dp{key1.p};
				auto flag_it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), [dp] __device__ (uint64_t i) { return (uint32_t)(i > 0 && dp.k[i] == dp.k[i - 1]); });
				size_t tb2 = 0;
				PGA_HIP(rocprim::inclusive_scan(nullptr, tb2, flag_it, dupc.p, n_a, rocprim::plus<uint32_t>(), st));
				DBuf<uint8_t> tmp2(tb2 ? tb2 : 1);
				PGA_HIP(rocprim::inclusive_scan(tmp2.p, tb2, flag_it, dupc.p, n_a, rocprim::plus<uint32_t>(), st));
			}
			hipLaunchKernelGGL(k_z_take_sorted, dim3(nba), dim3(256), 0, st, z.p, sx.p, sy.p, q_aoff.p, n_z.p, n_seq, spec ? (const uint32_t*)nullptr : q_tie.p, n_a);
			const RsHint hint{sx.p, sy.p, dupc.p};
			if (!spec) { replay_sort_segments(z.p, n_a, q_aoff.p, n_z.p, n_seq, q_tie.p, st, tm, &hint); PGA_HIP(sync_stream(st)); }
			// (the buffers of this scope go back to the call's arena: whoever takes them next is queued behind these kernels on the same stream)
		}
		et2.mark();
		EventTimer et3(st);
		hipLaunchKernelGGL(k_bt_walk, dim3((unsigned)n_seq), dim3(64), 0, st, n_seq, q_aoff.p, a.p, f.p, pp.p, t.p, v.p, z.p, n_z.p, u.p, w.p, u2.p, out.p, P, n_u.p, n_v.p,
		                   verbose ? prof.p : (unsigned long long*)nullptr, ev.p);
		if (n_a) hipLaunchKernelGGL(k_bt_copy, dim3((unsigned)((n_a + 255) / 256)), dim3(256), 0, st, n_seq, q_aoff.p, (uint64_t)n_a, a.p, v.p, u.p, w.p, n_u.p, n_v.p, out.p);
		const double ms_walk = et3.stop(K_BACKTRACK);
		ms_list = et.finish(K_BACKTRACK); ms_sort = et2.finish();
		if (verbose) {
			std::vector<uint32_t> he = ev.download(st), hq = q_tie_f.n ? q_tie_f.download(st) : std::vector<uint32_t>();
			size_t c[4] = {0, 0, 0, 0}, ft = 0; for (size_t i = 0; i < he.size(); ++i) { c[0] += he[i] & 1; c[1] += (he[i] >> 1) & 1; c[2] += (he[i] >> 2) & 1; c[3] += he[i] != 0; ft += i < hq.size() && hq[i]; }
			fprintf(stderr, "[pga]   backtrack order events: %zu of %d queries (%zu with equal scores): walk stopped at an equal-score mark %zu, candidate marked by an equal score %zu, equal chain starts %zu\n", c[3], n_seq, ft, c[0], c[1], c[2]);
		}
		if (spec && need_out) {
			DBuf<uint32_t> need((size_t)n_seq);
			hipLaunchKernelGGL(k_need_exact, dim3((unsigned)((n_seq + 255) / 256)), dim3(256), 0, st, n_seq, d_q_tie_x, q_tie_f.p, ev.p, seg_q.p, need.p);
			*need_out = need.download(st);
		}
		const double ms = ms_list + ms_walk;                 // the sort replay is accounted under K_SORT
		if (verbose) {
			std::vector<unsigned long long> pr = prof.download(st);   // wall_clock64 ticks at 100 MHz
			fprintf(stderr, "[pga]   backtrack: %.3f ms = candidate lists %.3f + walks %.3f; sort replay %.3f (per-query max: walks %.2f, compact %.2f ms)\n", ms, ms_list, ms_walk, ms_sort,
			        pr[4] * 1e-5, pr[5] * 1e-5);
			fprintf(stderr, "[pga]   backtrack, the query with the most anchors (%llu): %llu walks, %llu window reloads, %llu walk iterations, %llu candidate trips; inside walks %.2f ms, setting marks %.2f ms\n",
			        pr[6], pr[7], pr[8], pr[9], pr[10], pr[11] * 1e-5, pr[12] * 1e-5);
		}
		if (tm) { tm->kern[K_BACKTRACK].ms += ms; tm->kern[K_BACKTRACK].launches += 1; tm->kern[K_BACKTRACK].alg_bytes += 40.0 * (double)n_a; } // f,p read + anchors read + compacted anchors written
	}
	PGA_HIP(hipGetLastError());
	mark("backtrack");
	std::vector<uint64_t> h_off;
	{ Downloads dl(st); dl.add(O.n_u, n_u.p, n_u.n); dl.add(O.n_v, n_v.p, n_v.n); dl.add(h_off, q_aoff.p, q_aoff.n); dl.wait(); }
	// the chains of a query are the first n_u entries of its slice of u: only those travel (a leaf part holds ~5 k chains in an array of
	// 57 M slots), and land at the same positions of the host array
	{
		std::vector<uint64_t> coff((size_t)n_seq + 1, 0);
		for (int q = 0; q < n_seq; ++q) coff[(size_t)q + 1] = coff[(size_t)q] + (uint64_t)O.n_u[(size_t)q];
		const uint64_t total = coff[(size_t)n_seq];
		O.u.resize(u.n);
		if (total) {
			DBuf<uint64_t> d_coff; d_coff.upload(coff, st);
			DBuf<uint64_t> uc((size_t)total);
			hipLaunchKernelGGL(k_gather_chains, dim3((unsigned)n_seq), dim3(64), 0, st, u.p, q_aoff.p, d_coff.p, n_seq, uc.p);
			PinVec<uint64_t> hc; download_to(hc, uc.p, (size_t)total, st);
			for (int q = 0; q < n_seq; ++q) if (O.n_u[(size_t)q] > 0) memcpy(O.u.data() + h_off[(size_t)q], hc.data() + coff[(size_t)q], (size_t)O.n_u[(size_t)q] * sizeof(uint64_t));
		}
	}
	mark("chains to host");
	if (O.want_host_anchors) download_to(O.a, out.p, out.n, st);
	O.d_a = std::move(out);
	mark("anchors to host");
	if (vmarks) fprintf(stderr, "[pga]   chain stage, host ms:%s\n", marks.c_str());
}

// The stage.  minimap2 sorts twice with an UNSTABLE in-place radix sort whose arrangement of equal keys is the outcome of its sequential walk
// (ksort.h:101-151): the anchors by target position before chaining (map.c:202) and the candidate chain ends by score before backtracking
// (lchain.c:52).  Replaying those walks (pga_sort_replay.hip) is exact and is what a batch used to wait for twice.  For almost every query the
// chains do NOT depend on either arrangement, and that is decidable while computing them in a stable order:
//   (1) Anchors with equal x (same strand, target, position) differ in their query position y.  mg_lchain_rmq inserts the anchors of one x
//       together, after all of them have been evaluated (lchain.c:285-292), so they are never one another's predecessors; trees are keyed by
//       (y, index) and two anchors with the same y differ in x, so no comparison between tree keys is decided by the index of a permuted anchor;
//       f[], and p[] as a relation between anchors, are the same for every arrangement -- as long as no range-min query meets a tied minimum
//       priority (then the answer is a function of the tree's shape, i.e. of the insertion order: those segments run the tree re-enactment, and a
//       query that holds equal keys AND such a segment needs the reference's order).
//   (2) mg_chain_backtrack visits the candidates in descending score.  Candidates of different scores are ordered whatever the sort does.  Take
//       the candidates of one score in any order: a walk only ever reads marks (lchain.c:14,17) and a finished walk only sets the marks of its
//       kept part, so two candidates c, c' of equal score can influence each other only if the later one's walk stops at a mark of the earlier
//       one's chain, or the later one is itself marked by it.  (If c's walk runs THROUGH anchors c' will keep, they lie behind c's cut: stopping
//       there instead changes neither its maximum nor its kept part.)  k_bt_walk stores the claimant's score in every mark and raises an order
//       event in exactly those two cases; none raised in the stable order = none can be raised in any order = the same set of chains.
//   (3) What still differs is the ORDER in which chains are emitted, i.e. the input order of compact_a's own unstable sort (lchain.c:96-99, key:
//       target position of the first anchor).  Distinct keys end up sorted whatever the input order; equal keys only matter if, in addition, two
//       chains were emitted by candidates of equal score (otherwise the emission order is the descending score order and is exact).
// Queries with an order event go through the reference's procedure: exact anchor order (seed_exact_order), chained again, exact candidate order.
// On the BASELINE build: no event of kind (2) at all, ~37 queries per step (of 230 000) with (3).
void chain_all(const SeqSet &S, SeedResult &SR, const mm_mapopt_t &opt, int k, ChainResult &O, hipStream_t st, Timers *tm, bool exact_order)
{
	// what only this stage reads of the seeding stage's result goes back to the arena when it returns (36 B per anchor that used to stay live through
	// the alignment stage of every batch in flight); the alignment stage reads h_q_aoff / h_rep_len only.  Freed blocks are reused in stream order.
	struct Release { SeedResult &R; ~Release() { R.raw_x.release(); R.raw_y.release(); R.srt_x.release(); R.srt_y.release(); R.dupc.release(); R.q_tie.release(); R.a.release(); } } release_on_return{SR};
	const int n_seq = S.n_seq;
	const bool verbose = getenv("PGA_VERBOSE") != nullptr;
	if (exact_order || exact_sorts_forced()) {
		if (!SR.exact) seed_exact_order(SR, nullptr, n_seq, st, tm);
		chain_core(SR.a, SR.q_aoff, n_seq, SR.n_a, opt, k, O, st, tm, false, nullptr, nullptr);
		return;
	}
	std::vector<uint32_t> need;
	chain_core(SR.a, SR.q_aoff, n_seq, SR.n_a, opt, k, O, st, tm, true, SR.exact ? nullptr : SR.q_tie.p, &need);
	std::vector<int> F;
	for (int q = 0; q < n_seq; ++q) if (need[(size_t)q]) F.push_back(q);
	if (verbose) fprintf(stderr, "[pga]   chain: %zu of %d queries need the reference's tie order\n", F.size(), n_seq);
	if (F.empty()) return;
	const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
	// the reference's procedure for the queries of F, as a batch of their own
	if (!SR.exact) { DBuf<uint32_t> d_need; d_need.upload(need, st); seed_exact_order(SR, d_need.p, n_seq, st, tm); }
	const int n_sub = (int)F.size();
	std::vector<uint64_t> src((size_t)n_sub), dst((size_t)n_sub + 1, 0);
	for (int i = 0; i < n_sub; ++i) { src[(size_t)i] = SR.h_q_aoff[(size_t)F[(size_t)i]]; dst[(size_t)i + 1] = dst[(size_t)i] + (SR.h_q_aoff[(size_t)F[(size_t)i] + 1] - src[(size_t)i]); }
	const uint64_t n_sub_a = dst[(size_t)n_sub];
	DBuf<uint64_t> d_src, d_dst; d_src.upload(src, st); d_dst.upload(dst, st);
	DBuf<u128> a_sub(n_sub_a ? n_sub_a : 1);
	hipLaunchKernelGGL(k_gather_queries, dim3((unsigned)n_sub, 64), dim3(256), 0, st, n_sub, d_src.p, d_dst.p, SR.a.p, a_sub.p);
	ChainResult R; R.want_host_anchors = O.want_host_anchors;
	chain_core(a_sub, d_dst, n_sub, n_sub_a, opt, k, R, st, tm, false, nullptr, nullptr);
	for (int i = 0; i < n_sub; ++i) {
		const int q = F[(size_t)i];
		const size_t n = (size_t)(dst[(size_t)i + 1] - dst[(size_t)i]);
		O.n_u[(size_t)q] = R.n_u[(size_t)i]; O.n_v[(size_t)q] = R.n_v[(size_t)i];
		if (n == 0) continue;
		if (R.n_u[(size_t)i] > 0) memcpy(O.u.data() + src[(size_t)i], R.u.data() + dst[(size_t)i], (size_t)R.n_u[(size_t)i] * sizeof(uint64_t));
		if (O.want_host_anchors) memcpy(O.a.data() + src[(size_t)i], R.a.data() + dst[(size_t)i], n * sizeof(u128));
		PGA_HIP(hipMemcpyAsync(O.d_a.p + src[(size_t)i], R.d_a.p + dst[(size_t)i], n * sizeof(u128), hipMemcpyDeviceToDevice, st));
	}
	PGA_HIP(sync_stream(st));
	if (verbose) fprintf(stderr, "[pga]   chain: the reference's procedure for those %d queries (%llu anchors): %.1f ms\n", n_sub, (unsigned long long)n_sub_a,
	                     (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0) * 1e3);
}

} // namespace pga
