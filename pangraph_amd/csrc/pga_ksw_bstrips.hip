// pga_ksw_bstrips.hip -- kernel #5i: ONE BANDED problem of ksw_extd2_sse in exact-maximum mode (C/ksw2_extd2_sse.c:34-401: the end extensions of
// mm_align1 -- band 1.5 * bw, z-drop, KSW_EZ_EXTZ_ONLY -- and banded fills) as a PIPELINE OF WAVES over several CUs, for the launches that hold only
// a few such problems.
//
// Why: the lane kernel (pga_ksw_lanes.hip) keeps a problem on ONE CU -- four waves, eight columns per lane, one barrier per diagonal -- and a
// diagonal costs ~1 300 instructions of each wave: 2.0 us, i.e. 3.2 ms for the ~1 540 diagonals an end extension into unrelated sequence runs
// before the reference's z-drop can fire, 48 ms for one that runs its 20 k diagonals out.  Every call of a build waits for a handful of those,
// round after round; only the leaf batches hold enough of them (thousands) to fill the device that way.  Here the band is cut into strips of 64
// columns, ONE COLUMN PER LANE, a wave per strip (pga_ksw_wstrips.hip does the same for unbanded problems): no barrier, no LDS in the diagonal
// loop, ~150 instructions of one wave per diagonal; strip k runs one block of 64 diagonals behind strip k - 1 and takes that strip's last column
// (x, v, x2, H per diagonal) from device memory, 64 diagonals per load.
//
// What the band adds to the unbanded formulation:
//   * the reference's ranges: a diagonal computes the columns [st, en] = [st0, en0] rounded outwards to sixteen (ksw2_extd2_sse.c:173-190), the
//     cells outside [st0, en0] from whatever the rows hold -- and what they leave in u, y, y2 is what a column starts from when the band reaches
//     it.  A lane keeps the rows of its column in registers whether or not the column is in range, computes exactly when st <= t <= en, refreshes
//     its score byte exactly when st0 <= t < st0 + span (the profile loop of :196-221 runs in blocks of sixteen from st0), takes the first-row
//     values on diagonal r = t (:184-190) and the fresh-edge values when it is column st (:176-183).  All six rows are int8 in the reference:
//     every stored value is wrapped to eight bits here (inside the band nothing wraps; outside it the rows hold garbage that may).
//   * strips come and go: strip k holds cells from the first diagonal whose en reaches its first column to the last whose st has not passed its
//     last column (both monotone in r; the host tabulates them).  A POOL of waves per problem takes strips in ascending order from an atomic
//     counter: a strip only ever waits for the strip before it, which an earlier taker holds or has finished, so the pipeline cannot deadlock
//     however few of the pool's waves are resident.
//   * z-drop while the strips run: the keys of a diagonal (one packed key per strip: clamped H, the reference's tie class, column) are combined
//     by atomicMax; every strip counts the blocks of 64 diagonals it has completed, and ONE more wave of the pool, the problem's EVALUATOR, waits
//     for a block's counter to reach the number of strips alive in it and takes the reference's per-diagonal decisions for that block, in order
//     (ksw2_extd2_sse.c:326-366, ksw2.h:167-184); a z-drop raises the stop flag, strips look at it once per block.  (The strip that completes a
//     block last is the rightmost, the one every later strip waits for: when it also evaluated the block, a diagonal cost 1.15 us instead of
//     0.55.)  Strips run at most the pipeline's depth beyond the stopping diagonal; what they write there nobody reads.
// The evaluator then walks the path back (ksw2.h:127-159) through a 64 x 64 LDS window.  A clamped maximum hands the problem
// back (n_cigar = -9: the workgroup kernel redoes it).  Parity: tests/test_gpu_parity.py (PGA_BSTRIPS=force sends every eligible problem here).
#include "pga_common.h"
#include "pga_dp.h"
#include "pga_wave.h"
#include <cstring>
#include <vector>

namespace pga {

#define KSW_NEG_INF (-0x40000000)
#define BS_W 64
#define BS_BT 64
#define EZ_RIGHT      0x02
#define EZ_APPROX_MAX 0x08
#define EZ_EXTZ_ONLY  0x40
#define EZ_REV_CIGAR  0x80

struct BsCtl { uint32_t next_strip, stop, roles, pad[5]; };      // one per problem, zeroed before the launch
static_assert(sizeof(BsCtl) == 32, "control block: four 64-bit words");

__host__ __device__ __forceinline__ void bs_range(int r, int qlen, int tlen, int w, int &st0, int &en0)
{
	int st = 0, en = tlen - 1;
	if (st < r - qlen + 1) st = r - qlen + 1;
	if (en > r) en = r;
	if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
	if (en > (r + w) >> 1) en = (r + w) >> 1;
	st0 = st, en0 = en;
}
__device__ __forceinline__ int bs_sx8(int v) { return __builtin_amdgcn_sbfe(v, 0, 8); }
__device__ __forceinline__ int bs_byte(uint32_t v, int sh) { return __builtin_amdgcn_sbfe((int)v, sh, 8); }

// the layout of a problem's words (64 bit each) behind bnd_off: control block | best key per diagonal | H[en0] per diagonal, H[st0] per diagonal
// (32 bit) | completed strips per block of 64 diagonals (32 bit) | boundary words: one row per strip over the diagonals of its LIFETIME only (the
// rows' offsets are in the problem's table: 4 MB instead of 25 for a 10 kb x 10 kb extension -- all of it is zeroed before every launch)
struct BsLayout { int T, n_strips, n_diag, nblk, n_col; size_t Ld, o_bnd, o_best, o_hen, o_hst, o_done, words; };   // words: without the boundary rows
__host__ __device__ inline BsLayout bs_layout(int qlen, int tlen, int w)
{
	BsLayout L;
	L.T = (tlen + 15) / 16 * 16;
	L.n_strips = (L.T + BS_W - 1) / BS_W;
	L.n_diag = qlen + tlen - 1;
	L.nblk = (L.n_diag + 63) / 64;
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	int n_col = qlen < tlen ? qlen : tlen;
	L.n_col = (((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1) * 16;
	L.Ld = (size_t)(qlen + tlen);
	L.o_best = sizeof(BsCtl) / 8;
	L.o_hen = L.o_best + L.Ld;
	L.o_hst = L.o_hen + (L.Ld + 1) / 2;
	L.o_done = L.o_hst + (L.Ld + 1) / 2;
	L.o_bnd = L.o_done + ((size_t)L.nblk + 1) / 2 + 2;
	L.words = L.o_bnd;
	return L;
}

// tab (32 bit words) of a problem: [0] waves in its pool, [1] first diagonal with an empty range (or n_diag), then first diagonal / last diagonal /
// offset of its boundary row for every strip (first > last: the band never reaches it), then the number of strips alive in every block of 64 diagonals
template <bool RIGHT>
__device__ __forceinline__ void bstrip_body(const DpJob &J, const uint32_t jl, uint8_t *s_win, PkBases bases, const DpParams &P,
               uint8_t *__restrict__ slab_all, const uint64_t *__restrict__ slab_off, unsigned long long *__restrict__ bnd_all, const uint64_t *__restrict__ bnd_off,
               const uint32_t *__restrict__ tab_all, const uint64_t *__restrict__ tab_off,
               DpRes *__restrict__ res, uint32_t *__restrict__ cigar_pool, unsigned long long *__restrict__ pool_cursor, unsigned long long pool_cap)
{
	const int lane = threadIdx.x;
	const uint64_t t_base = J.t_off, q_base = J.q_off;
	const int qlen = J.qlen, tlen = J.tlen, flag = J.flag, zdrop = J.zdrop, end_bonus = J.end_bonus;
	int w = J.w;
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
	const int qe_h = q + e;
	if (q2 + e2 < q + e) { int t = q; q = q2, q2 = t, t = e, e = e2, e2 = t; }
	const int qe = q + e, qe2 = q2 + e2;
	const int sc_mch = P.sc_mch, sc_mis = P.sc_mis, sc_N = P.sc_ambi == 0 ? -e2 : P.sc_ambi;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const BsLayout Lo = bs_layout(qlen, tlen, J.w);
	const int T = Lo.T, n_strips = Lo.n_strips, n_diag = Lo.n_diag, n_col = Lo.n_col;
	uint8_t *pmat = slab_all + slab_off[jl];
	uint32_t *cig_tmp = (uint32_t*)(pmat + (((size_t)n_diag * n_col + 15) & ~(size_t)15));
	unsigned long long *base_w = bnd_all + bnd_off[jl];
	BsCtl *ctl = (BsCtl*)base_w;
	unsigned long long *bnd = base_w + Lo.o_bnd, *best_arr = base_w + Lo.o_best;
	int32_t *hen_arr = (int32_t*)(base_w + Lo.o_hen), *hst_arr = (int32_t*)(base_w + Lo.o_hst);
	uint32_t *done = (uint32_t*)(base_w + Lo.o_done);
	const uint32_t *tab = tab_all + tab_off[jl];
	const int n_eff = (int)tab[1];                                  // diagonals [0, n_eff) have a range; diagonal n_eff, if it exists, ends the problem (z-dropped)
	const uint32_t *strip_r = tab + 2, *need = tab + 2 + 3 * (size_t)n_strips;
	const int nblk_eff = (n_eff + 63) / 64;
	uint32_t *s_key = (uint32_t*)s_win;
	const int INI1 = bs_sx8(-q - e), INI2 = bs_sx8(-q2 - e2);
	const uint32_t PK_INI = ((uint32_t)INI1 & 0xffu) | ((uint32_t)INI1 & 0xffu) << 8 | ((uint32_t)INI2 & 0xffu) << 16;
	auto target_at = [&](int i) -> int { return (i >= 0 && i < tlen) ? (int)bases.at(t_base + (uint64_t)(J.seq_rev ? tlen - 1 - i : i)) : 0; };
	auto query_at = [&](int j) -> int {
		if (j < 0 || j >= qlen) return 0;
		const int pj = J.qs + (J.seq_rev ? qlen - 1 - j : j);
		if (!J.q_rev) return bases.at(q_base + (uint64_t)(pj));
		const int c = bases.at(q_base + (uint64_t)(J.qlen_full - 1 - pj));
		return c < 4 ? 3 - c : 4;
	};
	auto first_row = [&](int r) -> int { return bs_sx8(r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2); };
	auto stopped = [&]() -> uint32_t { return __hip_atomic_load(&ctl->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

	// ---- roles: the first wave of the pool to arrive is the problem's EVALUATOR, the others take strips ----
	uint32_t role = 0;
	if (lane == 0) role = atomicAdd(&ctl->roles, 1u);
	role = (uint32_t)__builtin_amdgcn_readfirstlane((int)role);
	// the evaluator's state, and the reference's per-diagonal decisions for one completed block of 64 diagonals (true: the problem ends there)
	int ez_max = 0, ez_max_t = -1, ez_max_q = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1, ez_mte = KSW_NEG_INF, ez_mte_q = -1, ez_score = KSW_NEG_INF, ez_zdropped = 0, r_done = 0;
	int sat = 0;
	const LbStop LB = lb_stop_of(qlen, tlen, w, flag, q, e, q2, e2, sc_mch, sc_mis, sc_N, P.lb_mode == 2 ? 0 : P.lb_mode);   // the length-bound stop (pga_dp.h; the checked mode is the lane kernel's)
	auto eval_block = [&](int b) -> bool {
		bool halt = false;
		const int r0 = b * 64, r = r0 + lane;
		unsigned long long bestk = 0; int hen = KSW_NEG_INF, hst = KSW_NEG_INF;
		if (r < n_eff) {
			bestk = __hip_atomic_load(&best_arr[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			hen = __hip_atomic_load(&hen_arr[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			hst = __hip_atomic_load(&hst_arr[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		const uint32_t h16 = (uint32_t)(bestk >> 24) & 0xffffu;
		const int mH_l = (int)h16 - 32768, mt_l = (4095 - (int)((bestk >> 8) & 4095)) * BS_W + (63 - (int)(bestk & 255));
		const int sat_l = r < n_eff && (h16 == 0 || h16 == 65535u) ? 1 : 0;
		const int lim = n_eff - r0 < 64 ? n_eff - r0 : 64;
		for (int ii = 0; ii < lim; ++ii) {
			const int rr = r0 + ii;
			const int mH = __builtin_amdgcn_readlane(mH_l, ii), mt = __builtin_amdgcn_readlane(mt_l, ii);
			const int he = __builtin_amdgcn_readlane(hen, ii), hs = __builtin_amdgcn_readlane(hst, ii);
			sat |= __builtin_amdgcn_readlane(sat_l, ii);
			int st0, en0; bs_range(rr, qlen, tlen, w, st0, en0);
			r_done = rr + 1;
			if (en0 == tlen - 1) { if (he > ez_mte) ez_mte = he, ez_mte_q = rr - en0; if (rr == n_diag - 1) ez_score = he; }
			if (rr - st0 == qlen - 1 && hs > ez_mqe) ez_mqe = hs, ez_mqe_t = st0;
			const bool upd = mH > ez_max;
			const int tl = mt - ez_max_t, ql = (rr - mt) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
			const bool stop = !upd && tl >= 0 && ql >= 0 && zdrop >= 0 && ez_max - mH > zdrop + l * e2;
			if (upd) ez_max = mH, ez_max_t = mt, ez_max_q = rr - mt;
			if (stop) { ez_zdropped = 1, ez_score = KSW_NEG_INF; halt = true; break; }
			if (LB.on && (rr & 7) == 7 && lb_final(LB, rr, tlen, q, e, q2, e2, sc_mch, ez_max < ez_mte ? ez_max : ez_mte)) { ez_zdropped = 1; halt = true; break; }
		}
		if (!halt && r0 + lim >= n_eff && n_eff < n_diag) { ez_zdropped = 1; r_done = n_eff + 1; halt = true; }      // the range ran empty (ksw2_extd2_sse.c:172)
		if (sat) halt = true;
		return halt;
	};
	// a problem of ONE strip (a ring of at most 64 columns: the extensions towards a block end) is one wave: it evaluates its own blocks
	const bool solo = n_strips == 1;
	if (role != 0 || solo) {
	// ---- strips, in ascending order from the problem's counter ----
	for (;;) {
		uint32_t k = 0;
		if (lane == 0) k = atomicAdd(&ctl->next_strip, 1u);
		k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
		if ((int)k >= n_strips || stopped()) break;
		const int r_first = (int)strip_r[3 * k], r_last = (int)strip_r[3 * k + 1];
		if (r_first > r_last) continue;
		const int rf_left = k > 0 ? (int)strip_r[3 * k - 3] : 0, rl_left = k > 0 ? (int)strip_r[3 * k - 2] : -1;
		// rows are indexed by diagonal: the pointers are moved back by the row's first diagonal
		const unsigned long long *bnd_in = k > 0 && rf_left <= rl_left ? bnd + (size_t)strip_r[3 * k - 1] - rf_left : nullptr;
		unsigned long long *bnd_out = (int)k + 1 < n_strips ? bnd + (size_t)strip_r[3 * k + 2] - r_first : nullptr;
		const int c0 = (int)k * BS_W, t = c0 + lane;
		const int tb = target_at(t);
		// the lane's column: the rows at index t as the reference's freshly initialised arrays hold them (ksw2_extd2_sse.c:109-118)
		int U = INI1, Y = INI1, Y2 = INI2, S = 0, H = KSW_NEG_INF;
		uint32_t PK = PK_INI;
		int qb = query_at(r_first - 1 - t);                         // query[(r - 1) - t] for r = r_first: what the slide below starts from
		bool gone = false;
#ifdef PGA_BS_PROF
		long long pf_poll = 0, pf_comp = 0, pf_blk = 0, pf_n = 0; const long long pf_t0 = clock64();
#endif
		// ranges of the diagonal before the strip's first (what column st's edge rule looks at, ksw2_extd2_sse.c:176-183)
		int last_st = -1, last_en = -1;
		if (r_first > 0) { int a0, a1; bs_range(r_first - 1, qlen, tlen, w, a0, a1); last_st = a0 & ~15, last_en = ((a1 + 16) & ~15) - 1; }
		for (int b = r_first >> 6; b <= r_last >> 6 && !gone; ++b) {
			if (stopped()) { gone = true; break; }
			const int r_lo = b * 64 > r_first ? b * 64 : r_first, r_hi = b * 64 + 63 < r_last ? b * 64 + 63 : r_last;
			// what a diagonal needs that does not depend on the column, a diagonal per lane: its range, the first-row value, the query base that
			// enters lane 0 (the loop below takes them by v_readlane: a dozen scalar instructions per diagonal less)
			int v_st0, v_en0;
			bs_range(b * 64 + lane, qlen, tlen, w, v_st0, v_en0);
			const int v_fr = first_row(b * 64 + lane);
			const int qwin = query_at(b * 64 + lane - c0);
			unsigned long long inw = (unsigned long long)(uint32_t)KSW_NEG_INF << 32 | PK_INI;
			unsigned long long outw = 0;
			int hen_acc = 0, hst_acc = 0; bool hen_set = false, hst_set = false;
			uint8_t *prow = pmat + (size_t)r_lo * n_col;               // row of the direction matrix of the current diagonal
			// four sub-blocks of sixteen diagonals: the strip on the left publishes its last column after every sixteen, so this strip runs sixteen
			// (not sixty-four) diagonals behind it
			for (int sub = 0; sub < 4 && !gone; ++sub) {
				const int s_lo = b * 64 + 16 * sub > r_lo ? b * 64 + 16 * sub : r_lo, s_hi = b * 64 + 16 * sub + 15 < r_hi ? b * 64 + 16 * sub + 15 : r_hi;
				if (s_lo > s_hi) continue;
#ifdef PGA_BS_PROF
				const long long pf0 = clock64();
#endif
				{
					const int d = b * 64 + lane - 1;                      // the cell of diagonal d + 1 in column c0 reads the left strip's state after diagonal d
					if (bnd_in && d >= rf_left && d <= rl_left && d + 1 >= s_lo && d + 1 <= s_hi) {
						for (;;) {
							inw = __hip_atomic_load(&bnd_in[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							if (inw >> 31 & 1ULL) break;
							if (stopped()) { gone = true; break; }
							__builtin_amdgcn_s_sleep(4);
						}
					}
				}
				if (__ballot(gone) != 0ULL) { gone = true; break; }
#ifdef PGA_BS_PROF
				const long long pf1 = clock64(); pf_poll += pf1 - pf0;
#endif
				for (int r = s_lo; r <= s_hi; ++r) {
					const int i = r - b * 64;
					const int st0 = __builtin_amdgcn_readlane(v_st0, i), en0 = __builtin_amdgcn_readlane(v_en0, i), fr = __builtin_amdgcn_readlane(v_fr, i);
					const int st = st0 & ~15, en = ((en0 + 16) & ~15) - 1, span = ((en0 - st0) & ~15) + 16;
					qb = wave_shr1(qb, __builtin_amdgcn_readlane(qwin, i));
					// the column that joins on this diagonal starts from the first-row values (ksw2_extd2_sse.c:184-190)
					const bool join = t == r && en >= r;
					U = join ? fr : U; Y = join ? INI1 : Y; Y2 = join ? INI2 : Y2;
					// x, v, x2 (and H) of the column on the left as the previous diagonal left them; column st takes the edge values (:176-183)
					uint32_t lw = (uint32_t)wave_shr1((int)PK, __builtin_amdgcn_readlane((int)(uint32_t)inw, i));
					const int Hl = wave_shr1(H, __builtin_amdgcn_readlane((int)(uint32_t)(inw >> 32), i));
					{
						const bool keep = st > 0 && st - 1 >= last_st && st - 1 <= last_en;       // (uniform) the column on the left of st was computed on the last diagonal
						const uint32_t edge = st == 0 ? (((uint32_t)INI1 & 0xffu) | ((uint32_t)fr & 0xffu) << 8 | ((uint32_t)INI2 & 0xffu) << 16) : PK_INI;
						lw = (t == st && !keep) ? edge : lw;
					}
					// the score byte: refreshed over [st0, st0 + span) only (the profile loop runs in blocks of sixteen from st0, :196-221)
					{
						int sc = tb == qb ? sc_mch : sc_mis;
						sc = ((tb | qb) & 4) ? sc_N : sc;
						S = (t >= st0 && t < st0 + span && t < T) ? bs_sx8(sc) : S;
					}
					const bool in_rng = t >= st && t <= en;
					const int xt1 = bs_byte(lw, 0), vt1 = bs_byte(lw, 8), x2t1 = bs_byte(lw, 16);
					int z = S;
					int a = bs_sx8(xt1 + vt1), bb = bs_sx8(Y + U), a2 = bs_sx8(x2t1 + vt1), b2 = bs_sx8(Y2 + U), d;
					if (!RIGHT) {
						d = a > z ? 1 : 0; z = a > z ? a : z;
						d = bb > z ? 2 : d; z = bb > z ? bb : z;
						d = a2 > z ? 3 : d; z = a2 > z ? a2 : z;
						d = b2 > z ? 4 : d; z = b2 > z ? b2 : z;
					} else {
						d = z > a ? 0 : 1;  z = z > a ? z : a;
						d = z > bb ? d : 2; z = z > bb ? z : bb;
						d = z > a2 ? d : 3; z = z > a2 ? z : a2;
						d = z > b2 ? d : 4; z = z > b2 ? z : b2;
					}
					z = sc_mch < z ? sc_mch : z;
					const int un = bs_sx8(z - vt1), vn = bs_sx8(z - U);
					int tmp = bs_sx8(z - q); a = bs_sx8(a - tmp); bb = bs_sx8(bb - tmp);
					tmp = bs_sx8(z - q2); a2 = bs_sx8(a2 - tmp); b2 = bs_sx8(b2 - tmp);
					int xn, yn, x2n, y2n;
					if (!RIGHT) {
						xn = bs_sx8((a > 0 ? a : 0) - qe);     d |= a > 0 ? 0x08 : 0;
						yn = bs_sx8((bb > 0 ? bb : 0) - qe);   d |= bb > 0 ? 0x10 : 0;
						x2n = bs_sx8((a2 > 0 ? a2 : 0) - qe2); d |= a2 > 0 ? 0x20 : 0;
						y2n = bs_sx8((b2 > 0 ? b2 : 0) - qe2); d |= b2 > 0 ? 0x40 : 0;
					} else {
						xn = bs_sx8((0 > a ? 0 : a) - qe);     d |= !(0 > a) ? 0x08 : 0;
						yn = bs_sx8((0 > bb ? 0 : bb) - qe);   d |= !(0 > bb) ? 0x10 : 0;
						x2n = bs_sx8((0 > a2 ? 0 : a2) - qe2); d |= !(0 > a2) ? 0x20 : 0;
						y2n = bs_sx8((0 > b2 ? 0 : b2) - qe2); d |= !(0 > b2) ? 0x40 : 0;
					}
					// a column outside [st, en] keeps its rows; Un / Vn: what u[t], v[t] hold behind this diagonal
					const int Un = in_rng ? un : U, Vn = in_rng ? vn : bs_byte(PK, 8);
					const uint32_t pkn = ((uint32_t)xn & 0xffu) | ((uint32_t)vn & 0xffu) << 8 | ((uint32_t)x2n & 0xffu) << 16;
					U = Un; Y = in_rng ? yn : Y; Y2 = in_rng ? y2n : Y2; PK = in_rng ? pkn : PK;
					if (in_rng) prow[t - st] = (uint8_t)d;
					// H[t] += v[t] over [st0, en0); H[en0] = H[en0 - 1] (as the previous diagonal left it) + u[en0] (ksw2_extd2_sse.c:325-340); the maximum
					// with the reference's tie order (H[en0] first, then four lanes by (t - st0) & 3 over [st0, en1), then the tail) as one key per column
					{
						const int en1 = st0 + ((en0 - st0) & ~3);
						const bool act = t >= st0 && t <= en0, is_en = t == en0;
						const int h_en = Hl + Un, h_in = H + Vn;
						int h = (is_en && en0 > 0) ? h_en : h_in;
						h = (is_en && r == 0) ? Vn - qe_h : h;
						H = act ? h : H;
						const uint32_t field = is_en ? 8u : 7u - (t < en1 ? (uint32_t)((t - st0) & 3) : 4u);
						const int hc = h < -32768 ? -32768 : h > 32767 ? 32767 : h;
						const uint32_t key = ((uint32_t)(hc + 32768) << 16) | field << 12 | (uint32_t)(63 - lane);
						s_key[i * 65 + lane] = act ? key : 0u;
						// H[en0] and H[st0] of the diagonal for the end-of-sequence scores: collected a diagonal per lane, stored once per block
						const int le = en0 - c0, ls = st0 - c0;
						if ((unsigned)le < 64u) { const int hv = __builtin_amdgcn_readlane(H, le & 63); if (lane == i) hen_acc = hv, hen_set = true; }
						if ((unsigned)ls < 64u && r - st0 == qlen - 1) { const int hv = __builtin_amdgcn_readlane(H, ls & 63); if (lane == i) hst_acc = hv, hst_set = true; }
					}
					{
						const uint32_t w63 = (uint32_t)__builtin_amdgcn_readlane((int)PK, 63);
						const unsigned long long wv = 0x80000000ULL | (w63 & 0x00ffffffu) | (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(H, 63) << 32;
						outw = lane == i ? wv : outw;
					}
					prow += n_col;
					last_st = st, last_en = en;
				}
				if (bnd_out && outw) { __hip_atomic_store(&bnd_out[b * 64 + lane], outw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); outw = 0; }
#ifdef PGA_BS_PROF
				pf_comp += clock64() - pf1; pf_n += s_hi - s_lo + 1;
#endif
			}
			if (gone) break;
			if (hen_set) __hip_atomic_store(&hen_arr[b * 64 + lane], hen_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (hst_set) __hip_atomic_store(&hst_arr[b * 64 + lane], hst_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			{
				const int r = b * 64 + lane;
				if (r >= r_lo && r <= r_hi) {
					uint32_t kb = 0;
#pragma unroll 16
					for (int l = 0; l < 64; ++l) { const uint32_t o = s_key[lane * 65 + l]; kb = o > kb ? o : kb; }
					// equal H and class: the lower strip, then the lower column (the order the reference's scan meets them)
					const unsigned long long comb = (unsigned long long)(kb >> 12) << 20 | (unsigned long long)(4095 - (int)k) << 8 | (unsigned long long)(kb & 255u);
					if (kb) atomicMax(&best_arr[r], comb);
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			__threadfence();
			if (lane == 0) (void)__hip_atomic_fetch_add(&done[b], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
			if (solo) { __threadfence(); if (eval_block(b)) { gone = true; break; } }
		}
#ifdef PGA_BS_PROF
		if (lane == 0 && jl == 0 && (k % 6) == 2) printf("[bstrips prof] strip %u: %lld diagonals; cycles per diagonal: poll %lld compute %lld everything %lld\n", k, pf_n, pf_poll / (pf_n ? pf_n : 1), pf_comp / (pf_n ? pf_n : 1), (clock64() - pf_t0) / (pf_n ? pf_n : 1));
#endif
		if (gone) break;
	}

		if (!solo) return;
	}
	// ---- the evaluator: the reference's per-diagonal decisions (ksw2_extd2_sse.c:326-366, ksw2.h:167-184), block by block as the strips complete
	// them; a z-drop (or a clamped key, or a range that ran empty) raises the stop flag; then the walk back ----
	if (!solo) {
		bool halt = false;
		for (int b = 0; b < nblk_eff && !halt; ++b) {
			// (a guard, not a path: if a block's strips never report -- they cannot, by the ordering argument above -- the problem is handed back
			// to the workgroup kernel after ~20 M polls instead of hanging the launch)
			{ long long polls = 0; while (__hip_atomic_load(&done[b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need[b]) { if (++polls > 20000000LL) { sat = 1; break; } __builtin_amdgcn_s_sleep(8); } }
			if (sat) { halt = true; break; }
			__threadfence();
			halt = eval_block(b);
		}
		if (lane == 0) __hip_atomic_store(&ctl->stop, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // strips that are still out there leave
	}
	int ez_reach_end = 0;
	int n_cigar = 0, bi = -1, bj = -1;
	if (sat) {}
	else if (!ez_zdropped && !(flag & EZ_EXTZ_ONLY)) bi = tlen - 1, bj = qlen - 1;
	else if (!ez_zdropped && (flag & EZ_EXTZ_ONLY) && ez_mqe + end_bonus > ez_max) ez_reach_end = 1, bi = ez_mqe_t, bj = qlen - 1;
	else if (ez_max_t >= 0 && ez_max_q >= 0) bi = ez_max_t, bj = ez_max_q;
	{
		int i = bi, j = bj, state = 0; long long guard = 0;
		uint32_t last_op = 0xffffffffu, run_len = 0;
		auto cg_push = [&](uint32_t op, uint32_t len) {
			if (op == last_op) { run_len += len; return; }
			if (last_op != 0xffffffffu) { if (lane == 0) cig_tmp[n_cigar] = run_len << 4 | last_op; ++n_cigar; }
			last_op = op; run_len = len;
		};
		auto cg_flush = [&] { if (last_op != 0xffffffffu && n_cigar >= 0) { if (lane == 0) cig_tmp[n_cigar] = run_len << 4 | last_op; ++n_cigar; last_op = 0xffffffffu; } };
		while (i >= 0 && j >= 0) {
			if (++guard > 4000000) { n_cigar = -7; break; }
			const int r_hi = i + j, c_lo = i - (BS_BT - 1);
			{
				uint8_t wv[BS_BT];
#pragma unroll
				for (int row = 0; row < BS_BT; ++row) {
					const int r = r_hi - row, col = c_lo + lane;
					uint8_t val = 0;
					if (r >= 0 && col >= 0) {
						int st0, en0; bs_range(r, qlen, tlen, w, st0, en0);
						const int off = st0 / 16 * 16, off_end = (en0 + 16) / 16 * 16 - 1;
						if (st0 <= en0 && col >= off && col <= off_end) val = pmat[(size_t)r * n_col + (size_t)(col - off)];
					}
					wv[row] = val;
				}
#pragma unroll
				for (int row = 0; row < BS_BT; ++row) s_win[row * BS_BT + lane] = wv[row];
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			while (i >= 0 && j >= 0) {
				const int r = i + j, row = r_hi - r;
				if (row >= BS_BT || i < c_lo) break;
				int st0, en0; bs_range(r, qlen, tlen, w, st0, en0);
				const int off = st0 / 16 * 16, off_end = (en0 + 16) / 16 * 16 - 1;
				int force_state = -1;
				if (i < off) force_state = 2;
				if (i > off_end) force_state = 1;
				const uint32_t tmp = force_state < 0 ? s_win[row * BS_BT + (i - c_lo)] : 0;
				if (state == 0) state = tmp & 7;
				else if (!(tmp >> (state + 2) & 1)) state = 0;
				if (state == 0) state = tmp & 7;
				if (force_state >= 0) state = force_state;
				uint32_t op;
				if (state == 0) op = 0, --i, --j;
				else if (state == 1 || state == 3) op = 2, --i;
				else op = 1, --j;
				cg_push(op, 1u);
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		}
		if (bi >= 0 && bj >= 0 && n_cigar >= 0) {
			if (i >= 0) cg_push(2u, (uint32_t)(i + 1));
			if (j >= 0) cg_push(1u, (uint32_t)(j + 1));
		}
		cg_flush();
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	unsigned long long base = 0;
	if (lane == 0 && n_cigar > 0) base = atomicAdd(pool_cursor, (unsigned long long)n_cigar);
	base = ((unsigned long long)(unsigned)__shfl((int)(base >> 32), 0) << 32) | (unsigned)__shfl((int)(base & 0xffffffffULL), 0);
	const bool rev_cigar = flag & EZ_REV_CIGAR;
	if (n_cigar > 0 && base + (unsigned long long)n_cigar <= pool_cap)
		for (int c = lane; c < n_cigar; c += 64) cigar_pool[base + c] = rev_cigar ? cig_tmp[c] : cig_tmp[n_cigar - 1 - c];
	if (lane == 0) {
		DpRes R;
		R.max = ez_max, R.max_q = ez_max_q, R.max_t = ez_max_t, R.mqe = ez_mqe, R.mqe_t = ez_mqe_t, R.mte = ez_mte, R.mte_q = ez_mte_q;
		R.score = ez_score, R.zdropped = ez_zdropped, R.reach_end = ez_reach_end, R.n_cigar = sat ? -9 : n_cigar, R.pad = r_done, R.cigar_off = base;
		res[jl] = R;
	}
}

__global__ __launch_bounds__(64)
void k_bstrips(const DpJob *__restrict__ jobs, const uint32_t *__restrict__ blk_job, PkBases bases, DpParams P,
               uint8_t *__restrict__ slab_all, const uint64_t *__restrict__ slab_off, unsigned long long *__restrict__ bnd_all, const uint64_t *__restrict__ bnd_off,
               const uint32_t *__restrict__ tab_all, const uint64_t *__restrict__ tab_off,
               DpRes *__restrict__ res, uint32_t *__restrict__ cigar_pool, unsigned long long *__restrict__ pool_cursor, unsigned long long pool_cap)
{
	__shared__ __align__(16) uint8_t s_win[64 * 65 * 4];      // the keys of a block ([diagonal][lane], row stride 65 words); at the end the traceback window
	const uint32_t jl = blk_job[blockIdx.x];
	const DpJob J = jobs[jl];
	if (J.flag & EZ_RIGHT) bstrip_body<true>(J, jl, s_win, bases, P, slab_all, slab_off, bnd_all, bnd_off, tab_all, tab_off, res, cigar_pool, pool_cursor, pool_cap);
	else bstrip_body<false>(J, jl, s_win, bases, P, slab_all, slab_off, bnd_all, bnd_off, tab_all, tab_off, res, cigar_pool, pool_cursor, pool_cap);
}

// ---- host side ----
// 0: the lane kernel keeps every banded problem; 1 (default): launches that hold few of them, and the long ones of any launch; 2: every eligible one
int bstrips_mode()               // (read on every call: the parity tests switch it inside one process)
{
	const char *e = getenv("PGA_BSTRIPS");
	return !e ? 1 : !strcmp(e, "off") || !strcmp(e, "0") ? 0 : !strcmp(e, "force") ? 2 : 1;
}
int bstrips_max_problems() { static const int v = getenv("PGA_BSTRIPS_MAX") ? atoi(getenv("PGA_BSTRIPS_MAX")) : 32; return v; }
int bstrips_long_diagonals() { static const int v = getenv("PGA_BSTRIPS_LONG") ? atoi(getenv("PGA_BSTRIPS_LONG")) : 6000; return v; }
bool bstrips_eligible(const DpJob &j, const DpParams &P)
{
	if (j.flag & (PGA_JOB_LL | EZ_APPROX_MAX)) return false;
	if (j.qlen < 1 || j.tlen < 1 || j.qlen > 32000 || j.tlen > 32000) return false;
	if (!(P.sc_mch >= 0 && P.sc_mch < 127)) return false;
	// (a narrow ring -- an extension towards a block end a few bases away: 20 columns, 10 k diagonals -- is ONE strip: no hand-over at all, and a
	// diagonal of one column per lane costs a quarter of the one-wave lane kernel's)
	return j.qlen + j.tlen >= 128;
}
size_t bstrips_slab_bytes(const DpJob &j)
{
	const BsLayout L = bs_layout(j.qlen, j.tlen, j.w);
	return (((size_t)L.n_diag * L.n_col + 15) & ~(size_t)15) + 4 * ((size_t)j.qlen + j.tlen + 8) + 256;
}
uint32_t bstrips_table(const DpJob &j, std::vector<uint32_t> &tab, size_t *words);
size_t bstrips_words(const DpJob &j) { std::vector<uint32_t> scratch; size_t w = 0; (void)bstrips_table(j, scratch, &w); return w; }
// the problem's table (see k_bstrips) appended to `tab`; returns the number of waves in its pool, *words = 64-bit words of the problem's region
uint32_t bstrips_table(const DpJob &j, std::vector<uint32_t> &tab, size_t *words)
{
	const BsLayout L = bs_layout(j.qlen, j.tlen, j.w);
	const int w = j.w < 0 ? (j.tlen > j.qlen ? j.tlen : j.qlen) : j.w;
	int n_eff = L.n_diag;
	std::vector<int> st((size_t)L.n_diag), en((size_t)L.n_diag);
	for (int r = 0; r < L.n_diag; ++r) {
		int st0, en0; bs_range(r, j.qlen, j.tlen, w, st0, en0);
		if (st0 > en0) { n_eff = r; break; }
		st[(size_t)r] = st0 & ~15; en[(size_t)r] = ((en0 + 16) & ~15) - 1;
	}
	const size_t at = tab.size();
	tab.resize(at + 2 + 3 * (size_t)L.n_strips + (size_t)L.nblk, 0u);
	uint32_t *first = &tab[at + 2], *need = &tab[at + 2 + 3 * (size_t)L.n_strips];
	size_t row_words = 0;
	int alive_max = 0;
	// en and st do not decrease along the diagonals: a strip's diagonals are one interval
	int r_in = 0, r_out = 0;
	for (int k = 0; k < L.n_strips; ++k) {
		const int c0 = k * BS_W, c1 = c0 + BS_W - 1;
		while (r_in < n_eff && en[(size_t)r_in] < c0) ++r_in;
		if (r_out < r_in) r_out = r_in;
		while (r_out < n_eff && st[(size_t)r_out] <= c1) ++r_out;      // r_out: the first diagonal whose st has passed the strip
		if (r_in >= n_eff || r_out <= r_in) { first[3 * k] = 1, first[3 * k + 1] = 0, first[3 * k + 2] = (uint32_t)row_words; continue; }
		first[3 * k] = (uint32_t)r_in, first[3 * k + 1] = (uint32_t)(r_out - 1), first[3 * k + 2] = (uint32_t)row_words;
		row_words += (size_t)(r_out - r_in) + 1;
		for (int b = r_in >> 6; b <= (r_out - 1) >> 6; ++b) ++need[b];
	}
	for (int b = 0; b < L.nblk; ++b) alive_max = std::max(alive_max, (int)need[b]);
	// strips run sixteen diagonals apart: little more than the strips alive in one block of 64 diagonals are busy at a time (a wave that waits for
	// its turn still holds a slot and polls: PGA_BSTRIPS_POOL_EXTRA more than that, default 4)
	static const int extra = getenv("PGA_BSTRIPS_POOL_EXTRA") ? atoi(getenv("PGA_BSTRIPS_POOL_EXTRA")) : 4;
	const uint32_t pool = L.n_strips == 1 ? 1u : 1u + (uint32_t)std::max(1, std::min(L.n_strips, std::min(alive_max + extra, 56)));      // + the evaluator; a single strip evaluates itself
	tab[at] = pool, tab[at + 1] = (uint32_t)n_eff;
	if (words) *words = L.words + row_words + 2;
	return pool;
}

void launch_bstrips(unsigned n_blocks, const DpJob *jobs, const uint32_t *blk_job, PkBases bases, const DpParams &P, uint8_t *slab, const uint64_t *slab_off,
                    unsigned long long *bnd, const uint64_t *bnd_off, const uint32_t *tab, const uint64_t *tab_off, DpRes *res, uint32_t *pool, unsigned long long *cursor,
                    unsigned long long pool_cap, hipStream_t st)
{
	hipLaunchKernelGGL(k_bstrips, dim3(n_blocks), dim3(64), 0, st, jobs, blk_job, bases, P, slab, slab_off, bnd, bnd_off, tab, tab_off, res, pool, cursor, pool_cap);
}

} // namespace pga
