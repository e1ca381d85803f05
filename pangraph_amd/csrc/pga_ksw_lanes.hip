// pga_ksw_lanes.hip -- kernel #5g: the dual-affine DP for BANDED problems (end extensions with band 1.5*bw under z-drop,
// banded gap fills) with every row of ksw2_extd2_sse.c:131-386 held in REGISTERS.
//
// The workgroup kernel of pga_ksw_wide.hip keeps the rows u, v, x, y, x2, y2, s (and H in exact mode) in LDS: one cell per thread
// and diagonal, ~17 LDS operations per cell and three workgroup barriers per diagonal (five with the exact maximum) -- 4-8 us per
// diagonal, whatever the band.  End extensions (band 1501, 1.5-40 k diagonals, exact maximum + z-drop) are what the rounds of a
// wave wait for.  Here a lane OWNS eight consecutive target columns:
//   * the seven int8 rows of its columns are sign-extended int16 pairs in VGPRs (v_pk_* arithmetic, two columns per
//     instruction; the reference's int8 wrap-around is applied where it stores), H is eight int32 registers;
//   * the value a cell takes from its left neighbour (x, v, x2 of column t-1 on the previous diagonal) is a register of the
//     same lane for seven columns out of eight, one DPP move (wave_shr:1) for the eighth, and a 4-byte LDS mailbox between
//     waves;
//   * the band slides over the lanes as a RING of NT*8 columns: a lane whose block left the band re-initialises its registers
//     with the values the reference's freshly allocated rows hold and takes the block NT*8 columns further right (same
//     stale-lane semantics as the LDS ring of the workgroup kernel);
//   * the query streams through an 8-byte register window (one LDS byte per lane and diagonal), the target bytes of a block
//     are loaded once; the score bytes come from two SWAR compares + v_perm;
//   * the exact maximum with the reference's tie order is a DPP wave reduction + one 8-byte partial per wave, and every
//     cross-wave value travels through mailboxes double-buffered by diagonal parity: ONE barrier per diagonal, which orders
//     LDS only (direction bytes are fire-and-forget 8-byte stores, fenced before the backtrack).
// Reference semantics, 16-lane rounding of the ranges, stale score bytes beyond the profile span and the traceback are the
// ones of pga_ksw_wide.hip (see there and pga_ksw.hip); the parity suites of tests/test_gpu_parity.py run both.
#include "pga_common.h"
#include "pga_dp.h"
#include "pga_wave.h"
#include "pga_pk16.h"
#include <cstdio>

namespace pga {

#define KSW_NEG_INF (-0x40000000)
#define EZ_RIGHT      0x02
#define EZ_APPROX_MAX 0x08
#define EZ_APPROX_DROP 0x10
#define EZ_EXTZ_ONLY  0x40
#define EZ_REV_CIGAR  0x80
#define LBT 64
#define LANES_C 8

__device__ __forceinline__ void diag_range_l(int r, int qlen, int tlen, int w, int &st0, int &en0)
{
	int st = 0, en = tlen - 1;
	if (st < r - qlen + 1) st = r - qlen + 1;
	if (en > r) en = r;
	if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
	if (en > (r + w) >> 1) en = (r + w) >> 1;
	st0 = st, en0 = en;
}

__device__ __forceinline__ s2_t sx8p(s2_t v) { return v << 8 >> 8; }                       // int8 wrap-around of both halves
__device__ __forceinline__ int sx8l(int v) { return __builtin_amdgcn_sbfe(v, 0, 8); }

struct LaneBox {              // cross-wave values of one diagonal (two copies, by diagonal parity)
	long long part[8];        // per wave: best (H, tie order) key
	uint32_t nb[8];           // per wave: x, v, x2 of its last column (what lane 0 of the next wave reads on the next diagonal)
	int hprev, u_en, v_en, h_en_old, h_st, h0v, h0u, pad;
};

template <int NT>
__global__ __launch_bounds__(NT)
void k_extd2_lanes(const DpJob *__restrict__ jobs, uint32_t n_jobs, const uint8_t *__restrict__ nt4, DpParams P,
                   uint32_t *__restrict__ job_counter, uint8_t *__restrict__ slab_all, size_t slab_bytes, int q_cap,
                   DpRes *__restrict__ res, uint32_t *__restrict__ cigar_pool, unsigned long long *__restrict__ pool_cursor, unsigned long long pool_cap)
{
	constexpr int NW = NT / 64, C = LANES_C, RC = NT * C;
	extern __shared__ __align__(16) uint8_t qq[];       // the query window, orientation and complement resolved
	__shared__ uint32_t s_job;
	__shared__ LaneBox s_box[2];
	__shared__ uint8_t s_win[LBT * LBT];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint8_t *slab = slab_all + (size_t)blockIdx.x * slab_bytes;
	int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
	const int qe_h = q + e;
	if (q2 + e2 < q + e) { int t = q; q = q2, q2 = t, t = e, e = e2, e2 = t; }
	const int qe = q + e, qe2 = q2 + e2;
	const int sc_mch = P.sc_mch, sc_mis = P.sc_mis, sc_N = P.sc_ambi == 0 ? -e2 : P.sc_ambi;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const uint32_t sc_tab = (uint32_t)(uint8_t)sc_mch | (uint32_t)(uint8_t)sc_mis << 8 | (uint32_t)(uint8_t)sc_N << 16 | (uint32_t)(uint8_t)sc_N << 24;
	const s2_t ZERO = splat2(0), ONE = splat2(1), MCH = splat2(sc_mch), Q1 = splat2(q), Q2 = splat2(q2), QE = splat2(qe), QE2 = splat2(qe2);
	const s2_t INI1 = splat2(sx8l(-q - e)), INI2 = splat2(sx8l(-q2 - e2));
	const uint32_t nb_init = (uint32_t)(uint8_t)(-q - e) | (uint32_t)(uint8_t)(-q - e) << 8 | (uint32_t)(uint8_t)(-q2 - e2) << 16;
	(void)q_cap;

	for (;;) {
		__syncthreads();
		if (tid == 0) s_job = atomicAdd(job_counter, 1u);
		__syncthreads();
		const uint32_t jid = s_job;
		if (jid >= n_jobs) break;
		const DpJob J = jobs[jid];
		const uint8_t *t_base = nt4 + J.t_off, *q_base = nt4 + J.q_off;
		const int qlen = J.qlen, tlen = J.tlen, flag = J.flag, zdrop = J.zdrop, end_bonus = J.end_bonus;
		const bool approx_max = flag & EZ_APPROX_MAX, right = flag & EZ_RIGHT;
		int w = J.w;
		if (w < 0) w = tlen > qlen ? tlen : qlen;
		const int T = (tlen + 15) / 16 * 16;
		int n_col = qlen < tlen ? qlen : tlen;
		n_col = (((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1) * 16;
		auto target_at = [&](int i) -> uint32_t { return (i >= 0 && i < tlen) ? (uint32_t)t_base[J.seq_rev ? tlen - 1 - i : i] : 0u; };
		auto query_at = [&](int j) -> int {
			int pj = J.qs + (J.seq_rev ? qlen - 1 - j : j);
			if (!J.q_rev) return q_base[pj];
			int c = q_base[J.qlen_full - 1 - pj];
			return c < 4 ? 3 - c : 4;
		};
		for (int j = tid; j < qlen; j += NT) qq[j] = (uint8_t)query_at(j);
		if (tid < NW) { s_box[1].nb[tid] = nb_init; s_box[0].nb[tid] = nb_init; }
		uint8_t *pmat = slab;
		uint32_t *cig_tmp = (uint32_t*)(pmat + (((size_t)(qlen + tlen - 1) * n_col + 15) & ~(size_t)15));

		// ---- the lane's block: eight columns in registers ----
		s2_t X[4], V[4], X2[4], U[4], Y[4], Y2[4];
		int H[8];
		uint32_t S0 = 0, S1 = 0, TB0 = 0, TB1 = 0, W0 = 0, W1 = 0;
		int blk = tid;
		auto fresh = [&](int r_prev) {
#pragma unroll
			for (int p = 0; p < 4; ++p) { X[p] = V[p] = U[p] = Y[p] = INI1; X2[p] = Y2[p] = INI2; }
#pragma unroll
			for (int i = 0; i < 8; ++i) H[i] = KSW_NEG_INF;
			S0 = S1 = 0;
			const int t0 = blk * C;
			TB0 = target_at(t0) | target_at(t0 + 1) << 8 | target_at(t0 + 2) << 16 | target_at(t0 + 3) << 24;
			TB1 = target_at(t0 + 4) | target_at(t0 + 5) << 8 | target_at(t0 + 6) << 16 | target_at(t0 + 7) << 24;
			// window bytes k = query[r_prev - k - t0]: what the shift at the top of diagonal r_prev + 1 expects
			W0 = W1 = 0;
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				const int j = r_prev - k - t0;
				const uint32_t b = (j >= 0 && j < qlen) ? (uint32_t)qq[j] : 0u;
				if (k < 4) W0 |= b << (8 * k); else W1 |= b << (8 * (k - 4));
			}
		};
		__syncthreads();
		fresh(-1);

		int ez_max = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1, ez_mte = KSW_NEG_INF, ez_mte_q = -1;
		int ez_score = KSW_NEG_INF, ez_zdropped = 0, ez_reach_end = 0;
		int H0 = 0, last_H0_t = 0, last_st = -1, last_en = -1;
		const int n_diag = qlen + tlen - 1;
		int r_done = 0;

		for (int r = 0; r < n_diag; ++r) {
			r_done = r + 1;
			int st0, en0;
			diag_range_l(r, qlen, tlen, w, st0, en0);
			if (st0 > en0) { ez_zdropped = 1; break; }
			const int st = st0 / 16 * 16, en = (en0 + 16) / 16 * 16 - 1;
			const int span = ((en0 - st0) / 16 + 1) * 16;
			int need_hi = en > st0 + span - 1 ? en : st0 + span - 1;
			if (need_hi > T - 1) need_hi = T - 1;
			// a lane whose block fell out of the band on the left takes the block one ring further right (fresh rows)
			if ((blk + NT) * C <= need_hi) { blk += NT; fresh(r - 1); }
			const int t0 = blk * C;
			LaneBox &bx = s_box[r & 1];
			const LaneBox &bp = s_box[(r & 1) ^ 1];
			// query byte of the block's first column
			{
				const int j = r - t0;
				const uint32_t b = (j >= 0 && j < qlen) ? (uint32_t)qq[j] : 0u;
				W1 = W1 << 8 | W0 >> 24; W0 = W0 << 8 | b;
			}
			// left neighbour: x, v, x2 of column t0-1 as the previous diagonal left them
			const uint32_t mine = ((uint32_t)as_i(X[3]) >> 16 & 0xffu) | ((uint32_t)as_i(V[3]) >> 16 & 0xffu) << 8 | ((uint32_t)as_i(X2[3]) >> 16 & 0xffu) << 16;
			uint32_t inc = (uint32_t)wave_shr1((int)mine, 0);
			if (lane == 0) inc = bp.nb[(wave + NW - 1) % NW];
			int xin = sx8l((int)inc), vin = sx8l((int)(inc >> 8)), x2in = sx8l((int)(inc >> 16));
			if (t0 == st) {
				if (st > 0) {
					if (!(st - 1 >= last_st && st - 1 <= last_en)) xin = sx8l(-q - e), x2in = sx8l(-q2 - e2), vin = sx8l(-q - e);
				} else {
					xin = sx8l(-q - e), x2in = sx8l(-q2 - e2);
					vin = r == 0 ? sx8l(-q - e) : r < long_thres ? sx8l(-e) : r == long_thres ? sx8l(long_diff) : sx8l(-e2);
				}
			}
			// score bytes of the columns in [st0, st0+span) (the others keep what an earlier diagonal left there)
			{
				int lo = st0 - t0, hi = (st0 + span < T ? st0 + span : T) - t0;
				lo = lo < 0 ? 0 : lo > 8 ? 8 : lo; hi = hi < 0 ? 0 : hi > 8 ? 8 : hi;
				if (hi > lo) {
					const unsigned long long m = (hi == 8 ? ~0ull : (1ull << (8 * hi)) - 1) & ~((1ull << (8 * lo)) - 1);
					const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
					const uint32_t nz0 = ((TB0 ^ W0) + 0x7f7f7f7fu) >> 7 & 0x01010101u, nn0 = (TB0 | W0) >> 2 & 0x01010101u;
					const uint32_t nz1 = ((TB1 ^ W1) + 0x7f7f7f7fu) >> 7 & 0x01010101u, nn1 = (TB1 | W1) >> 2 & 0x01010101u;
					S0 = (S0 & ~m0) | (__builtin_amdgcn_perm(0u, sc_tab, nz0 | nn0 << 1) & m0);
					S1 = (S1 & ~m1) | (__builtin_amdgcn_perm(0u, sc_tab, nz1 | nn1 << 1) & m1);
				}
			}
			const bool act = t0 >= st && t0 <= en;
			long long best = (long long)KSW_NEG_INF * 4294967296LL;
			if (act) {
				// the column that joins on this diagonal starts from the first-row values
				if (en >= r && r >= t0 && r < t0 + 8) {
					const int uj = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
#pragma unroll
					for (int p = 0; p < 4; ++p) {
						if (t0 + 2 * p == r) { U[p].x = (short)sx8l(uj); Y[p].x = (short)sx8l(-q - e); Y2[p].x = (short)sx8l(-q2 - e2); }
						if (t0 + 2 * p + 1 == r) { U[p].y = (short)sx8l(uj); Y[p].y = (short)sx8l(-q - e); Y2[p].y = (short)sx8l(-q2 - e2); }
					}
				}
				int cx = xin << 16, cv = vin << 16, cx2 = x2in << 16;        // the left neighbour's values ride in the high half
				uint32_t dpk[4];
#pragma unroll
				for (int p = 0; p < 4; ++p) {
					const int ox = as_i(X[p]), ov = as_i(V[p]), ox2 = as_i(X2[p]);
					const s2_t xt1 = as_s2((int)__builtin_amdgcn_alignbit((uint32_t)ox, (uint32_t)cx, 16));
					const s2_t vt1 = as_s2((int)__builtin_amdgcn_alignbit((uint32_t)ov, (uint32_t)cv, 16));
					const s2_t x2t1 = as_s2((int)__builtin_amdgcn_alignbit((uint32_t)ox2, (uint32_t)cx2, 16));
					cx = ox, cv = ov, cx2 = ox2;
					const s2_t ut = U[p], yt = Y[p], y2t = Y2[p];
					s2_t z = unpack_i8x2((p < 2 ? S0 >> (16 * p) : S1 >> (16 * (p - 2))) & 0xffffu);
					s2_t a = sx8p(xt1 + vt1), b = sx8p(yt + ut), a2 = sx8p(x2t1 + vt1), b2 = sx8p(y2t + ut);
					const s2_t zm = pmax(pmax(pmax(z, a), pmax(b, a2)), b2);
					s2_t d;
					{
						const s2_t n0 = pminu(zm - z, ONE), n1 = pminu(zm - a, ONE), n2 = pminu(zm - b, ONE), n3 = pminu(zm - a2, ONE);
						if (!right) d = n0 * (ONE + n1 * (ONE + n2 * (ONE + n3)));           // first of z, a, b, a2, b2 that attains the maximum
						else {
							const s2_t n4 = pminu(zm - b2, ONE);                                // last one that attains it
							d = ONE - n1;
							d = d * n2 + (ONE - n2) * splat2(2);
							d = d * n3 + (ONE - n3) * splat2(3);
							d = d * n4 + (ONE - n4) * splat2(4);
						}
					}
					z = pmin(zm, MCH);
					const s2_t un = sx8p(z - vt1), vn = sx8p(z - ut);
					s2_t tmp = sx8p(z - Q1); a = sx8p(a - tmp); b = sx8p(b - tmp);
					tmp = sx8p(z - Q2); a2 = sx8p(a2 - tmp); b2 = sx8p(b2 - tmp);
					if (!right) {
						d = d + pmin(pmax(a, ZERO), ONE) * splat2(8) + pmin(pmax(b, ZERO), ONE) * splat2(16) + pmin(pmax(a2, ZERO), ONE) * splat2(32) + pmin(pmax(b2, ZERO), ONE) * splat2(64);
					} else {
						d = d + (ONE - pmin(pmax(ZERO - a, ZERO), ONE)) * splat2(8) + (ONE - pmin(pmax(ZERO - b, ZERO), ONE)) * splat2(16)
						      + (ONE - pmin(pmax(ZERO - a2, ZERO), ONE)) * splat2(32) + (ONE - pmin(pmax(ZERO - b2, ZERO), ONE)) * splat2(64);
					}
					X[p] = sx8p(pmax(a, ZERO) - QE); Y[p] = sx8p(pmax(b, ZERO) - QE);
					X2[p] = sx8p(pmax(a2, ZERO) - QE2); Y2[p] = sx8p(pmax(b2, ZERO) - QE2);
					U[p] = un; V[p] = vn;
					dpk[p] = (uint32_t)as_i(d);
				}
				uint2 dd;
				dd.x = __builtin_amdgcn_perm(dpk[1], dpk[0], 0x06040200u);
				dd.y = __builtin_amdgcn_perm(dpk[3], dpk[2], 0x06040200u);
				*reinterpret_cast<uint2*>(pmat + (size_t)r * n_col + (t0 - st)) = dd;
			}
			// ---- what the workgroup needs from single columns: posted by their owners (updated this diagonal or not) ----
			if (!approx_max) {
				if (r > 0) {
					const int en1 = st0 + (en0 - st0) / 4 * 4;
					if (t0 + 7 >= st0 - 1 && t0 <= en0)
#pragma unroll
					for (int i = 0; i < 8; ++i) {
						const int t = t0 + i;
						const int vn = (i & 1) ? (int)V[i >> 1].y : (int)V[i >> 1].x;
						const int hold = H[i];
						if (t == en0 - 1) bx.hprev = hold;
						if (t == en0) { bx.u_en = (i & 1) ? (int)U[i >> 1].y : (int)U[i >> 1].x; bx.v_en = vn; bx.h_en_old = hold; }
						if (t >= st0 && t < en0) {
							const int h = hold + vn;
							H[i] = h;
							const unsigned ord = t < en1 ? 1u + ((unsigned)((t - st0) & 3) << 28) + (unsigned)t : 1u + (4u << 28) + (unsigned)t;
							const long long key = ((long long)h << 32) | (0xffffffffu - ord);
							best = key > best ? key : best;
						}
						if (t == st0) bx.h_st = H[i];
					}
					best = wave_max_i64(best);
					if (lane == 0) bx.part[wave] = best;
				} else if (t0 == 0) bx.v_en = (int)V[0].x;
			} else {
				const int h0t = r == 0 ? 0 : last_H0_t;
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					const int t = t0 + i;
					if (t == h0t) bx.h0v = (i & 1) ? (int)V[i >> 1].y : (int)V[i >> 1].x;
					if (t == h0t + 1) bx.h0u = (i & 1) ? (int)U[i >> 1].y : (int)U[i >> 1].x;
				}
			}
			if (lane == 63) bx.nb[wave] = ((uint32_t)as_i(X[3]) >> 16 & 0xffu) | ((uint32_t)as_i(V[3]) >> 16 & 0xffu) << 8 | ((uint32_t)as_i(X2[3]) >> 16 & 0xffu) << 16;
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
			// ---- uniform bookkeeping of the diagonal (every lane, from the mailboxes) ----
			bool stop = false;
			if (!approx_max) {
				int max_H, max_t, h_en_now, h_st_now;
				if (r > 0) {
					const int Hen = en0 > 0 ? bx.hprev + bx.u_en : bx.h_en_old + bx.v_en;
					long long bb = bx.part[0];
#pragma unroll
					for (int k = 1; k < NW; ++k) { const long long o = bx.part[k]; bb = o > bb ? o : bb; }
					{ const long long hk = ((long long)Hen << 32) | 0xffffffffu; if (hk > bb) bb = hk; }
					max_H = (int)(bb >> 32);
					const unsigned ord = 0xffffffffu - (unsigned)(bb & 0xffffffffLL);
					max_t = ord == 0 ? en0 : (int)((ord - 1) & 0x0fffffffu);
					h_en_now = Hen;
					h_st_now = st0 == en0 ? Hen : bx.h_st;
					if (en0 >= t0 && en0 < t0 + 8) {
#pragma unroll
						for (int i = 0; i < 8; ++i) if (t0 + i == en0) H[i] = Hen;
					}
				} else {
					const int h0 = bx.v_en - qe_h;
					if (t0 == 0) H[0] = h0;
					max_H = h0, max_t = 0, h_en_now = h0, h_st_now = h0;
				}
				if (en0 == tlen - 1) { const int h = h_en_now; if (h > ez_mte) ez_mte = h, ez_mte_q = r - en0; }
				if (r - st0 == qlen - 1) { const int h = h_st_now; if (h > ez_mqe) ez_mqe = h, ez_mqe_t = st0; }
				if (max_H > ez_max) ez_max = max_H, ez_max_t = max_t, ez_max_q = r - max_t;
				else if (max_t >= ez_max_t && r - max_t >= ez_max_q) {
					const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
					if (zdrop >= 0 && ez_max - max_H > zdrop + l * e2) { ez_zdropped = 1; stop = true; }
				}
				if (!stop && r == n_diag - 1 && en0 == tlen - 1) ez_score = h_en_now;
			} else {
				if (r > 0) {
					if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
						const int d0 = bx.h0v, d1 = bx.h0u;
						if (d0 > d1) H0 += d0; else H0 += d1, ++last_H0_t;
					} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += bx.h0v;
					else ++last_H0_t, H0 += bx.h0u;
				} else H0 = bx.h0v - qe_h, last_H0_t = 0;
				if (flag & EZ_APPROX_DROP) {
					if (H0 > ez_max) ez_max = H0, ez_max_t = last_H0_t, ez_max_q = r - last_H0_t;
					else if (last_H0_t >= ez_max_t && r - last_H0_t >= ez_max_q) {
						const int tl = last_H0_t - ez_max_t, ql = (r - last_H0_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
						if (zdrop >= 0 && ez_max - H0 > zdrop + l * e2) { ez_zdropped = 1; stop = true; }
					}
				}
				if (!stop && r == n_diag - 1 && en0 == tlen - 1) ez_score = H0;
			}
			if (stop) break;
			last_st = st, last_en = en;
		}

		// ---- backtrack by wave 0 (ksw2.h:127-159) through a 64x64 LDS window of the direction matrix ----
		int n_cigar = 0, bi = -1, bj = -1;
		if (!ez_zdropped && !(flag & EZ_EXTZ_ONLY)) bi = tlen - 1, bj = qlen - 1;
		else if (!ez_zdropped && (flag & EZ_EXTZ_ONLY) && ez_mqe + end_bonus > ez_max) ez_reach_end = 1, bi = ez_mqe_t, bj = qlen - 1;
		else if (ez_max_t >= 0 && ez_max_q >= 0) bi = ez_max_t, bj = ez_max_q;
		__threadfence_block();
		__syncthreads();
		if (wave == 0) {
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
				const int r_hi = i + j, c_lo = i - (LBT - 1);
				{
					uint8_t wv[LBT];
#pragma unroll
					for (int row = 0; row < LBT; ++row) {
						const int r = r_hi - row, col = c_lo + lane;
						uint8_t val = 0;
						if (r >= 0 && col >= 0) {
							int st0, en0; diag_range_l(r, qlen, tlen, w, st0, en0);
							const int off = st0 / 16 * 16, off_end = (en0 + 16) / 16 * 16 - 1;
							if (st0 <= en0 && col >= off && col <= off_end) val = pmat[(size_t)r * n_col + (col - off)];
						}
						wv[row] = val;
					}
#pragma unroll
					for (int row = 0; row < LBT; ++row) s_win[row * LBT + lane] = wv[row];
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				while (i >= 0 && j >= 0) {
					const int r = i + j, row = r_hi - r;
					if (row >= LBT || i < c_lo) break;
					int st0, en0; diag_range_l(r, qlen, tlen, w, st0, en0);
					const int off = st0 / 16 * 16, off_end = (en0 + 16) / 16 * 16 - 1;
					int force_state = -1;
					if (i < off) force_state = 2;
					if (i > off_end) force_state = 1;
					const uint32_t tmp = force_state < 0 ? s_win[row * LBT + (i - c_lo)] : 0;
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
			unsigned long long base = 0;
			if (lane == 0 && n_cigar > 0) base = atomicAdd(pool_cursor, (unsigned long long)n_cigar);
			base = ((unsigned long long)(unsigned)__shfl((int)(base >> 32), 0) << 32) | (unsigned)__shfl((int)(base & 0xffffffffULL), 0);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			const bool rev_cigar = flag & EZ_REV_CIGAR;
			if (n_cigar > 0 && base + (unsigned long long)n_cigar <= pool_cap)
				for (int c = lane; c < n_cigar; c += 64) cigar_pool[base + c] = rev_cigar ? cig_tmp[c] : cig_tmp[n_cigar - 1 - c];
			if (lane == 0) {
				DpRes R;
				R.max = ez_max, R.max_q = ez_max_q, R.max_t = ez_max_t, R.mqe = ez_mqe, R.mqe_t = ez_mqe_t, R.mte = ez_mte, R.mte_q = ez_mte_q;
				R.score = ez_score, R.zdropped = ez_zdropped, R.reach_end = ez_reach_end, R.n_cigar = n_cigar, R.pad = r_done, R.cigar_off = base;
				res[jid] = R;
			}
		}
	}
}

// a problem the lane kernel takes: its band ring fits the NT*8 columns of a 256-thread workgroup and its query fits LDS
bool lanes_eligible(const DpJob &j)
{
	if (j.flag & PGA_JOB_LL) return false;
	if (j.qlen < 1 || j.tlen < 1 || j.qlen > 48 * 1024) return false;
	const int T = (j.tlen + 15) / 16 * 16;
	const int w = j.w < 0 ? (j.tlen > j.qlen ? j.tlen : j.qlen) : j.w;
	int R = ((w < j.tlen ? w : j.tlen) + 15) / 16 * 16 + 96;
	if (R > T) R = T;
	return R <= 256 * LANES_C;
}

void launch_extd2_lanes(unsigned n_blocks, int q_cap, const DpJob *jobs, uint32_t n_jobs, const uint8_t *nt4, const DpParams &P, uint32_t *counter, uint8_t *slab, size_t slab_bytes,
                        DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st)
{
	static bool attr_set = false;
	if (!attr_set) { PGA_HIP(hipFuncSetAttribute((const void*)k_extd2_lanes<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); attr_set = true; }
	const size_t lds = ((size_t)q_cap + 15) & ~(size_t)15;
	hipLaunchKernelGGL(k_extd2_lanes<256>, dim3(n_blocks), dim3(256), lds, st, jobs, n_jobs, nt4, P, counter, slab, slab_bytes, q_cap, res, pool, cursor, pool_cap);
}

} // namespace pga
