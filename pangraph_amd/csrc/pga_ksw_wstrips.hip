// pga_ksw_wstrips.hip -- kernel #5h: ONE large unbanded problem of ksw_extd2_sse (C/ksw2_extd2_sse.c:34-401 with w >= both lengths: the first
// passes of the 10 kb x 10 kb gap fills across rearrangements, KSW_EZ_APPROX_MAX, and their exact second passes under z-drop) as a PIPELINE OF
// WAVES: the target is cut into strips of 64 columns, one wave each, one column per lane, every row of the recurrence in registers.
//
// Why: the workgroup strips of pga_ksw_strips.hip (512 columns, rows in LDS, one barrier per diagonal) cost 1.2 us per diagonal whatever the
// strip holds, i.e. 25-36 ms for the 20 k diagonals of such a problem -- alone on a 256-CU device, on the critical path of its call.  A wave that
// owns ONE column per lane needs no barrier and no LDS: a cell takes u, y, y2 of its own column from the lane's registers, x, v, x2 of the column
// on its left from the neighbouring lane (one DPP move of a packed word), its query base from the neighbouring lane as well (the query slides
// along the strip one lane per diagonal).  A diagonal is ~90 instructions of one wave.  Strip k starts when strip k-1 is 64 diagonals ahead and
// reads the state of that strip's last column from device memory, 64 diagonals per load (each word carries its own valid bit: relaxed
// agent-scope accesses suffice); the pipeline is full after n_strips x ~130 diagonals and the whole matrix takes (qlen + tlen + that) steps.
//
// Semantics are those of pga_ksw_strips.hip (see there): direction bytes in ksw2's layout p[r][t - st(r)], approximate mode without a
// running corner score (the score of the backtracked path is evaluated at the end), exact mode with H per column, one packed key per strip and
// diagonal (clamped H, the reference's tie class, column) and ONE sweep over the recorded keys that takes the reference's z-drop and
// end-of-sequence decisions in order (ksw2_extd2_sse.c:326-366, ksw2.h:167-184).  No int8 wrap can occur inside an unbinding band, so the
// difference recurrence runs in plain 32-bit integers.
#include "pga_common.h"
#include "pga_dp.h"
#include "pga_wave.h"
#include <cstring>

namespace pga {

#define KSW_NEG_INF (-0x40000000)
#define WS_W 64             // columns per strip = lanes
#define WS_BT 64

__device__ __forceinline__ void ws_range(int r, int qlen, int tlen, int &st0, int &en0)
{
	st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0;
	en0 = r < tlen - 1 ? r : tlen - 1;
}
__device__ __forceinline__ int ws_sx8(uint32_t v, int sh) { return __builtin_amdgcn_sbfe((int)v, sh, 8); }
__device__ __forceinline__ unsigned long long ws_readlane64(unsigned long long v, int l)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
	return (unsigned long long)hi << 32 | lo;
}

template <bool EXACT>
__device__ __forceinline__ void wstrip_body(const DpJob &J, const uint32_t jl, const uint32_t k, uint8_t *s_win, PkBases bases, const DpParams &P,
               uint8_t *__restrict__ slab_all, const uint64_t *__restrict__ slab_off, unsigned long long *__restrict__ bnd_all, const uint64_t *__restrict__ bnd_off,
               uint32_t *__restrict__ done_ctr, DpRes *__restrict__ res, uint32_t *__restrict__ cigar_pool, unsigned long long *__restrict__ pool_cursor, unsigned long long pool_cap)
{
	const int lane = threadIdx.x;
	const uint64_t t_base = J.t_off, q_base = J.q_off;
	const int qlen = J.qlen, tlen = J.tlen;
	int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
	if (q2 + e2 < q + e) { int t = q; q = q2, q2 = t, t = e, e = e2, e2 = t; }
	const int qe = q + e, qe2 = q2 + e2;
	const int sc_mch = P.sc_mch, sc_mis = P.sc_mis, sc_N = P.sc_ambi == 0 ? -e2 : P.sc_ambi;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const int qe_h = P.q + P.e;
	const int n_strips = (tlen + WS_W - 1) / WS_W;
	const size_t Ld = (size_t)(qlen + tlen);
	const int c0 = (int)k * WS_W, c1 = c0 + WS_W < tlen ? c0 + WS_W : tlen;
	int n_col = qlen < tlen ? qlen : tlen;
	n_col = ((n_col + 15) / 16 + 1) * 16;                       // (w + 1 > both lengths)
	uint8_t *pmat = slab_all + slab_off[jl];
	uint32_t *cig_tmp = (uint32_t*)(pmat + (((size_t)(qlen + tlen - 1) * n_col + 15) & ~(size_t)15));
	unsigned long long *bnd = bnd_all + bnd_off[jl];
	const unsigned long long *bnd_in = k > 0 ? bnd + (size_t)(k - 1) * Ld : nullptr;             // written by strip k-1, indexed by diagonal
	unsigned long long *bnd_out = (int)k + 1 < n_strips ? bnd + (size_t)k * Ld : nullptr;
	// exact mode, behind the boundary words: H of the diagonal's last and first column per diagonal, then the best key of every diagonal
	int32_t *hen_arr = (int32_t*)(bnd + (size_t)(n_strips > 1 ? n_strips - 1 : 0) * Ld), *hst_arr = hen_arr + Ld;
	auto target_at = [&](int i) -> int { return (i >= 0 && i < tlen) ? (int)bases.at(t_base + (uint64_t)(J.seq_rev ? tlen - 1 - i : i)) : 0; };
	auto query_at = [&](int j) -> int {
		if (j < 0 || j >= qlen) return 0;
		const int pj = J.qs + (J.seq_rev ? qlen - 1 - j : j);
		if (!J.q_rev) return bases.at(q_base + (uint64_t)(pj));
		const int c = bases.at(q_base + (uint64_t)(J.qlen_full - 1 - pj));
		return c < 4 ? 3 - c : 4;
	};
	const int t = c0 + lane;
	const bool has_col = t < tlen;
	const int tb = target_at(t);
	// The lane's column: what its last cell left behind (ksw2's rows at index t): u, y, y2 as integers, x, v, x2 as the bytes of one word (the
	// word the lane on the right takes by DPP).  A column joins the matrix on diagonal r = t with the first-row values: y, y2 are the rows'
	// initial values anyway and u's first-row value only depends on t, so the registers simply start there and the loop knows no "join".
	const int u_first = t == 0 ? -q - e : t < long_thres ? -e : t == long_thres ? long_diff : -e2;
	int U = u_first, Y = -q - e, Y2 = -q2 - e2, H = KSW_NEG_INF;
	uint32_t PK = ((uint32_t)(-q - e) & 0xffu) | ((uint32_t)(-q - e) & 0xffu) << 8 | ((uint32_t)(-q2 - e2) & 0xffu) << 16;
	int qb = 0;
	const int r_first = c0, r_last = c1 - 1 + qlen - 1;          // the diagonals on which this strip holds cells
	const int d_last = c0 + qlen - 2;                           // last boundary diagonal the strip on the left publishes for this one
	const bool out_strip = bnd_out != nullptr;                  // (then the strip is full: its last column is lane 63)
	unsigned long long outw = 0;
	uint32_t *s_key = (uint32_t*)s_win;                         // exact mode: the keys of a block's cells, [diagonal][lane] with a row stride of 65 words
	unsigned long long *best_arr = (unsigned long long*)(hst_arr + Ld);   // exact mode: the best key of every diagonal over all strips
	int jq = -lane - 1;                                          // query row of the lane's cell on the diagonal BEFORE the block's first: r - t
	uint8_t *prow = pmat + (size_t)r_first * n_col;              // row of the direction matrix of the current diagonal, minus its first column st(r)
	{ int st0, en0; ws_range(r_first, qlen, tlen, st0, en0); prow -= st0 / 16 * 16; }
	for (int it0 = 0; r_first + it0 <= r_last; it0 += 64) {
		// the block's inputs: 64 query bases (diagonal r = r_first + it brings base it to lane 0) and 64 boundary words
		const int qwin = query_at(it0 + lane);
		unsigned long long inw = 0;
		{
			const int d = r_first - 1 + it0 + lane;               // the cell of diagonal d + 1 in column c0 reads the left strip's state after diagonal d
			if (bnd_in) {
				if (d >= c0 - 1 && d <= d_last)
					for (;;) { inw = __hip_atomic_load(&bnd_in[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (inw >> 31 & 1ULL) break; __builtin_amdgcn_s_sleep(2); }
			} else {
				// the matrix's first column: x, x2 of a fresh gap, v the first-column value of row d + 1 (ksw2_extd2_sse.c:191-194); no H to its left
				const int rr = d + 1, v1 = rr == 0 ? -q - e : rr < long_thres ? -e : rr == long_thres ? long_diff : -e2;
				inw = (unsigned long long)(uint32_t)KSW_NEG_INF << 32 | (((uint32_t)(-q - e) & 0xffu) | ((uint32_t)v1 & 0xffu) << 8 | ((uint32_t)(-q2 - e2) & 0xffu) << 16);
			}
		}
		const int i_end = __builtin_amdgcn_readfirstlane(r_last - (r_first + it0) + 1 < 64 ? r_last - (r_first + it0) + 1 : 64);
		for (int i = 0; i < i_end; ++i) {
			const int r = r_first + it0 + i;
			// the query slides along the strip: lane l held base j of diagonal r - 1, lane l + 1 needs it now
			qb = wave_shr1(qb, __builtin_amdgcn_readlane(qwin, i));
			++jq;
			const bool active = has_col && (unsigned)jq < (unsigned)qlen;
			// x, v, x2 (and H) of the column on the left as the previous diagonal left them
			const uint32_t lw = (uint32_t)wave_shr1((int)PK, __builtin_amdgcn_readlane((int)(uint32_t)inw, i));
			const int xl = ws_sx8(lw, 0), vl = ws_sx8(lw, 8), x2l = ws_sx8(lw, 16);
			int Hl = 0;
			if (EXACT) Hl = wave_shr1(H, __builtin_amdgcn_readlane((int)(uint32_t)(inw >> 32), i));
			int z = tb == qb ? sc_mch : sc_mis;
			z = ((tb | qb) & 4) ? sc_N : z;
			int a = xl + vl, b = Y + U, a2 = x2l + vl, b2 = Y2 + U;
			int d = a > z ? 1 : 0; z = a > z ? a : z;                 // the first of z, a, b, a2, b2 that attains the maximum (ksw2_extd2_sse.c:225-232)
			d = b > z ? 2 : d; z = b > z ? b : z;
			d = a2 > z ? 3 : d; z = a2 > z ? a2 : z;
			d = b2 > z ? 4 : d; z = b2 > z ? b2 : z;
			z = z < sc_mch ? z : sc_mch;
			const int un = z - vl, vn = z - U;
			int tmp = z - q; a -= tmp; b -= tmp;
			tmp = z - q2; a2 -= tmp; b2 -= tmp;
			d |= a > 0 ? 0x08 : 0; d |= b > 0 ? 0x10 : 0; d |= a2 > 0 ? 0x20 : 0; d |= b2 > 0 ? 0x40 : 0;
			const int xn = (a > 0 ? a : 0) - qe, yn = (b > 0 ? b : 0) - qe, x2n = (a2 > 0 ? a2 : 0) - qe2, y2n = (b2 > 0 ? b2 : 0) - qe2;
			if (active) {
				U = un; Y = yn; Y2 = y2n;
				PK = ((uint32_t)xn & 0xffu) | ((uint32_t)vn & 0xffu) << 8 | ((uint32_t)x2n & 0xffu) << 16;
				prow[(uint32_t)t] = (uint8_t)d;
			}
			if (EXACT) {
				// H[t] += v[t] over [st0, en0); H[en0] = H[en0-1] (as the previous diagonal left it) + u[en0]; the maximum with the reference's tie order
				// (H[en0] first, then four lanes by (t - st0) & 3 over [st0, en1), then the tail) is one key per column
				int st0, en0;
				ws_range(r, qlen, tlen, st0, en0);
				uint32_t key = 0;
				if (active) {
					const int en1 = st0 + (en0 - st0) / 4 * 4;
					int h; uint32_t field;
					if (t == en0) { h = r == 0 ? vn - qe_h : en0 > 0 ? Hl + un : H + vn; field = 8u; hen_arr[r] = h; }
					else { h = H + vn; field = 7u - (t < en1 ? (uint32_t)((t - st0) & 3) : 4u); }
					H = h;
					if (t == st0 && r - st0 == qlen - 1) hst_arr[r] = h;
					const int hc = h < -32768 ? -32768 : h > 32767 ? 32767 : h;
					key = ((uint32_t)(hc + 32768) << 16) | field << 12 | (uint32_t)(63 - lane);
				}
				s_key[i * 65 + lane] = key;                          // reduced once per block, a diagonal per lane
			}
			if (out_strip) {
				// the strip's last column feeds the strip on the right: collected lane by lane, flushed once per block
				const uint32_t w63 = (uint32_t)__builtin_amdgcn_readlane((int)PK, 63);
				const bool on63 = (unsigned)(r - (c0 + 63)) < (unsigned)qlen;
				unsigned long long w = on63 ? (0x80000000ULL | w63) : 0ULL;
				if (EXACT) w |= (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(H, 63) << 32;
				if (lane == i) outw = on63 ? w : 0ULL;
			}
			// the next diagonal's row: n_col further, minus the step of its first column st(r + 1) (sixteen at a time, once the diagonal has left row 0)
			prow += n_col;
			if (r + 2 - qlen > 0 && ((r + 2 - qlen) & 15) == 0) prow -= 16;
		}
		if (out_strip && outw) { __hip_atomic_store(&bnd_out[r_first + it0 + lane], outw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		outw = 0;
		if (EXACT) {
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			if (lane < i_end) {
				uint32_t kb = 0;
#pragma unroll 16
				for (int l = 0; l < 64; ++l) { const uint32_t o = s_key[lane * 65 + l]; kb = o > kb ? o : kb; }
				// equal H and class: the lower strip, then the lower column (the order the reference's scan meets them)
				const unsigned long long comb = (unsigned long long)(kb >> 12) << 20 | (unsigned long long)(4095 - (int)k) << 8 | (unsigned long long)(kb & 255u);
				if (kb) atomicMax(&best_arr[r_first + it0 + lane], comb);
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		}
	}
	// ---- the last strip of the problem to finish walks the path back and scores it ----
	__threadfence();
	uint32_t last = 0;
	if (lane == 0) last = atomicAdd(&done_ctr[jl], 1u);
	last = (uint32_t)__builtin_amdgcn_readfirstlane((int)last);
	if (last + 1 != (uint32_t)n_strips) return;
	__threadfence();
	int n_cigar = 0;
	int bi = tlen - 1, bj = qlen - 1;
	int ez_max = 0, ez_max_t = -1, ez_max_q = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1, ez_mte = KSW_NEG_INF, ez_mte_q = -1, zdropped = 0;
	if (EXACT) {
		// The maxima of all diagonals are on record: what the reference decides diagonal by diagonal (ksw_apply_zdrop, the end-of-sequence
		// scores) is decided here in one sweep, 64 diagonals per trip: a lane combines the strips' keys of its diagonal (equal H and field:
		// the lower strip, then the lower column), then the trip's diagonals are taken in order.
		const int n_diag = qlen + tlen - 1, zdrop = J.zdrop;
		int sat = 0;
		for (int r0 = 0; r0 < n_diag && !zdropped; r0 += 64) {
			const int r = r0 + lane;
			unsigned long long best = 0; int hen = KSW_NEG_INF, hst = KSW_NEG_INF;
			if (r < n_diag) {
				int st0, en0; ws_range(r, qlen, tlen, st0, en0);
				best = best_arr[r];
				hen = hen_arr[r];
				if (r - st0 == qlen - 1) hst = hst_arr[r];
			}
			const uint32_t h16 = (uint32_t)(best >> 24) & 0xffffu;
			const int mH_l = (int)h16 - 32768, mt_l = (4095 - (int)((best >> 8) & 4095)) * WS_W + (63 - (int)(best & 255));
			const int sat_l = r < n_diag && (h16 == 0 || h16 == 65535u) ? 1 : 0;
			const int lim = n_diag - r0 < 64 ? n_diag - r0 : 64;
			for (int ii = 0; ii < lim; ++ii) {
				const int rr = r0 + ii;
				const int mH = __builtin_amdgcn_readlane(mH_l, ii), mt = __builtin_amdgcn_readlane(mt_l, ii);
				const int he = __builtin_amdgcn_readlane(hen, ii), hs = __builtin_amdgcn_readlane(hst, ii);
				sat |= __builtin_amdgcn_readlane(sat_l, ii);
				int st0, en0; ws_range(rr, qlen, tlen, st0, en0);
				if (en0 == tlen - 1 && he > ez_mte) ez_mte = he, ez_mte_q = rr - en0;
				if (rr - st0 == qlen - 1 && hs > ez_mqe) ez_mqe = hs, ez_mqe_t = st0;
				if (mH > ez_max) ez_max = mH, ez_max_t = mt, ez_max_q = rr - mt;
				else if (mt >= ez_max_t && rr - mt >= ez_max_q) {
					const int tl = mt - ez_max_t, ql = (rr - mt) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
					if (zdrop >= 0 && ez_max - mH > zdrop + l * e2) { zdropped = 1; break; }
				}
			}
		}
		if (sat) {                                            // a maximum outside the keys' 16 bits: the workgroup kernel redoes the problem
			if (lane == 0) { DpRes R; memset(&R, 0, sizeof(R)); R.n_cigar = -9; res[jl] = R; }
			return;
		}
		if (zdropped) bi = ez_max_t, bj = ez_max_q;
	}
	int i = bi, j = bj, state = 0; long long guard = 0;
	uint32_t last_op = 0xffffffffu;
	uint32_t run_len = 0;
	auto cg_push = [&](uint32_t op, uint32_t len) {
		if (op == last_op) { run_len += len; return; }
		if (last_op != 0xffffffffu) { if (lane == 0) cig_tmp[n_cigar] = run_len << 4 | last_op; ++n_cigar; }
		last_op = op; run_len = len;
	};
	auto cg_flush = [&] { if (last_op != 0xffffffffu && n_cigar >= 0) { if (lane == 0) cig_tmp[n_cigar] = run_len << 4 | last_op; ++n_cigar; last_op = 0xffffffffu; } };
	while (i >= 0 && j >= 0) {
		if (++guard > 4000000) { n_cigar = -7; break; }
		const int r_hi = i + j, c_lo = i - (WS_BT - 1);
		{
			uint8_t wv[WS_BT];
#pragma unroll
			for (int row = 0; row < WS_BT; ++row) {
				const int r = r_hi - row, col = c_lo + lane;
				uint8_t val = 0;
				if (r >= 0 && col >= 0) {
					int st0, en0; ws_range(r, qlen, tlen, st0, en0);
					const int off = st0 / 16 * 16;
					if (st0 <= en0 && col >= st0 && col <= en0) val = pmat[(size_t)r * n_col + (col - off)];
				}
				wv[row] = val;
			}
#pragma unroll
			for (int row = 0; row < WS_BT; ++row) s_win[row * WS_BT + lane] = wv[row];
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		while (i >= 0 && j >= 0) {
			const int r = i + j, row = r_hi - r;
			if (row >= WS_BT || i < c_lo) break;
			int st0, en0; ws_range(r, qlen, tlen, st0, en0);
			const int off = st0 / 16 * 16, off_end = (en0 + 16) / 16 * 16 - 1;
			int force_state = -1;
			if (i < off) force_state = 2;
			if (i > off_end) force_state = 1;
			const uint32_t tmp = force_state < 0 ? s_win[row * WS_BT + (i - c_lo)] : 0;
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
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	// score of the path: cig_tmp runs from the end of the alignment to its start
	int score = KSW_NEG_INF;
	if (n_cigar > 0 && !zdropped) {
		int ti = tlen, qj = qlen; score = 0;
		for (int c = 0; c < n_cigar; ++c) {
			const uint32_t op = cig_tmp[c] & 0xf; const int len = (int)(cig_tmp[c] >> 4);
			if (op == 0) {
				ti -= len, qj -= len;
				int n_mis = 0, n_amb = 0;
				for (int b = 0; b < len; b += 64) {
					const int l = b + lane;
					bool mis = false, amb = false;
					if (l < len) { const int x = target_at(ti + l), y = query_at(qj + l); amb = ((x | y) & 4) != 0; mis = !amb && x != y; }
					n_mis += __popcll(__ballot(mis)); n_amb += __popcll(__ballot(amb));
				}
				score += sc_mch * (len - n_mis - n_amb) + sc_mis * n_mis + sc_N * n_amb;
			} else {
				const int g1 = q + len * e, g2 = q2 + len * e2;
				score -= g1 < g2 ? g1 : g2;
				if (op == 1) qj -= len; else ti -= len;
			}
		}
	}
	unsigned long long base = 0;
	if (lane == 0 && n_cigar > 0) base = atomicAdd(pool_cursor, (unsigned long long)n_cigar);
	base = ((unsigned long long)(unsigned)__shfl((int)(base >> 32), 0) << 32) | (unsigned)__shfl((int)(base & 0xffffffffULL), 0);
	if (n_cigar > 0 && base + (unsigned long long)n_cigar <= pool_cap)
		for (int c = lane; c < n_cigar; c += 64) cigar_pool[base + c] = cig_tmp[n_cigar - 1 - c];
	if (lane == 0) {
		DpRes R;
		R.max = ez_max, R.max_q = ez_max_q, R.max_t = ez_max_t, R.mqe = ez_mqe, R.mqe_t = ez_mqe_t, R.mte = ez_mte, R.mte_q = ez_mte_q;
		R.score = score, R.zdropped = zdropped, R.reach_end = 0, R.n_cigar = n_cigar, R.pad = qlen + tlen - 1, R.cigar_off = base;
		res[jl] = R;
	}
}

__global__ __launch_bounds__(64)
void k_wstrips(const DpJob *__restrict__ jobs, const uint32_t *__restrict__ blk_job, const uint32_t *__restrict__ blk_strip, PkBases bases, DpParams P,
               uint8_t *__restrict__ slab_all, const uint64_t *__restrict__ slab_off, unsigned long long *__restrict__ bnd_all, const uint64_t *__restrict__ bnd_off,
               uint32_t *__restrict__ done_ctr, DpRes *__restrict__ res, uint32_t *__restrict__ cigar_pool, unsigned long long *__restrict__ pool_cursor, unsigned long long pool_cap)
{
	__shared__ __align__(16) uint8_t s_win[64 * 65 * 4];      // traceback window (64 x 64 bytes); exact mode, before that: the keys of a block
	const uint32_t jl = blk_job[blockIdx.x], k = blk_strip[blockIdx.x];
	const DpJob J = jobs[jl];
	if (J.flag & 0x08) wstrip_body<false>(J, jl, k, s_win, bases, P, slab_all, slab_off, bnd_all, bnd_off, done_ctr, res, cigar_pool, pool_cursor, pool_cap);
	else wstrip_body<true>(J, jl, k, s_win, bases, P, slab_all, slab_off, bnd_all, bnd_off, done_ctr, res, cigar_pool, pool_cursor, pool_cap);
}

bool wstrips_on() { static const bool off = getenv("PGA_OLD_STRIPS") != nullptr; return !off; }
bool wstrips_eligible(const DpJob &j, const DpParams &P)
{
	static const int min_t = getenv("PGA_WSTRIPS_MIN") ? atoi(getenv("PGA_WSTRIPS_MIN")) : 1536;          // below ~24 strips the single-workgroup kernels win
	static const int min_x = getenv("PGA_WSTRIPS_EXACT_MIN") ? atoi(getenv("PGA_WSTRIPS_EXACT_MIN")) : 2048;
	if (min_t <= 0) return false;
	if (!(j.w >= j.qlen && j.w >= j.tlen && j.qlen >= 256 && j.qlen <= 16384 && j.tlen <= 16384 && P.sc_mch >= 0 && P.sc_mch < 127)) return false;
	if (j.flag == 0x08) return j.tlen >= min_t;
	return j.flag == 0 && min_x > 0 && j.tlen >= min_x;
}
int wstrips_count(const DpJob &j) { return (j.tlen + WS_W - 1) / WS_W; }
size_t wstrips_bnd_words(const DpJob &j)          // 64-bit words
{
	const size_t L = (size_t)(j.qlen + j.tlen), ns = (size_t)wstrips_count(j);
	return (ns > 1 ? ns - 1 : 0) * L + ((j.flag & 0x08) ? 0 : (2 * L + 1) / 2 + 1 + L) + 8;      // exact: hen, hst (32 bit), best key per diagonal (64 bit)
}

void launch_wstrips(unsigned n_blocks, const DpJob *jobs, const uint32_t *blk_job, const uint32_t *blk_strip, PkBases bases, const DpParams &P, uint8_t *slab,
                    const uint64_t *slab_off, unsigned long long *bnd, const uint64_t *bnd_off, uint32_t *done_ctr, DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st)
{
	hipLaunchKernelGGL(k_wstrips, dim3(n_blocks), dim3(64), 0, st, jobs, blk_job, blk_strip, bases, P, slab, slab_off, bnd, bnd_off, done_ctr, res, pool, cursor, pool_cap);
}

} // namespace pga
