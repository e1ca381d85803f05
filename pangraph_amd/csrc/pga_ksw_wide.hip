// pga_ksw_wide.hip -- kernel #5b: the dual-affine DP for WIDE or BANDED problems (end extensions up to
// max_gap x max_gap with band 1.5*bw, second exact passes, long-join segments): everything pga_ksw_fast.hip
// does not take.  Same per-lane int8 recurrence and the same 16-lane rounding / stale-profile behaviour as
// ksw2_extd2_sse.c:131-386 (see pga_ksw.hip), laid out for a 256-thread workgroup:
//   * all rows live in LDS as a RING over the band (dynamic, 10 B/column + 4 B/column of H in exact mode; a column's slot
//     is t mod R with R >= band + 96, so a 10 kb x 10 kb extension with band 2873 needs 42 KB, not 143 KB, and several
//     workgroups share a CU); a column that enters the band gets the initial values the reference's freshly
//     allocated rows hold, which keeps its stale-lane behaviour;
//   * both sequences are copied to LDS once per problem (orientation and complement resolved at copy time), so the
//     diagonal loop issues no global loads at all -- its direction-matrix stores are fire-and-forget;
//   * x, v and x2 -- the rows a cell reads at t-1 -- are double-buffered by diagonal parity, so the four waves
//     sweep a diagonal's band in parallel without a read/write hazard.  Band ranges only move right, so a
//     column that enters the band was never computed before and both buffers still hold its initial value:
//     the stale-lane semantics of the reference are preserved;
//   * 3 workgroup barriers per diagonal (5 in exact-max mode); the exact maximum with the reference's tie order
//     is a wave reduction + 4 LDS partials;
//   * wave 0 walks the direction matrix back through a 64x64 LDS window (fences only, as in the fast kernel).
#include <mutex>
#include "pga_common.h"
#include "pga_dp.h"
#include "pga_wave.h"
#include <cstdio>
#include "pga_pk16.h"

namespace pga {

#define KSW_NEG_INF (-0x40000000)
#define EZ_RIGHT      0x02
#define EZ_APPROX_MAX 0x08
#define EZ_APPROX_DROP 0x10
#define EZ_EXTZ_ONLY  0x40
#define EZ_REV_CIGAR  0x80
#define WBT 64
#define WIDE_LDS_MAX (152 * 1024)   // dynamic LDS the kernel may ask for (160 KB per CU minus its static arrays); pga_ksw.hip sizes classes with it

__device__ __forceinline__ int sx8w(int v) { return __builtin_amdgcn_sbfe(v, 0, 8); }

__device__ __forceinline__ void diag_range_w(int r, int qlen, int tlen, int w, int &st0, int &en0)
{
	int st = 0, en = tlen - 1;
	if (st < r - qlen + 1) st = r - qlen + 1;
	if (en > r) en = r;
	if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
	if (en > (r + w) >> 1) en = (r + w) >> 1;
	st0 = st, en0 = en;
}

__device__ int g_wide_no_fused = 0;   // PGA_NO_FUSED_APPROX=1 (A/B): the unbanded approximate passes take the general loop

template <int WIDE_NT>
__global__ __launch_bounds__(WIDE_NT)
void k_extd2_wide(const DpJob *__restrict__ jobs, uint32_t n_jobs, PkBases bases, DpParams P,
                  uint32_t *__restrict__ job_counter, uint8_t *__restrict__ slab_all, size_t slab_bytes, int r_cap, int seq_cap, int exact_rows,
                  DpRes *__restrict__ res, uint32_t *__restrict__ cigar_pool, unsigned long long *__restrict__ pool_cursor, unsigned long long pool_cap)
{
	extern __shared__ __align__(16) uint8_t dyn[];
	__shared__ uint32_t s_job;
	__shared__ long long s_part[WIDE_NT / 64];
	__shared__ int s_hprev;
	__shared__ int s_h0v[2], s_h0u[2];   // fused approximate path: v of the tracked column and u of its right neighbour, by diagonal parity
	__builtin_amdgcn_s_setprio(3);      // few, latency-bound workgroups: win issue arbitration against the tile kernels sharing the CU
	__shared__ uint8_t s_win[WBT * WBT];
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

	for (;;) {
		__syncthreads();
		if (tid == 0) s_job = atomicAdd(job_counter, 1u);
		__syncthreads();
		const uint32_t jid = s_job;
		if (jid >= n_jobs) break;
		const DpJob J = jobs[jid];
		const uint64_t t_base = J.t_off, q_base = J.q_off;          // base positions in the packed store
		const int qlen = J.qlen, tlen = J.tlen, flag = J.flag, zdrop = J.zdrop, end_bonus = J.end_bonus;
		const bool approx_max = flag & EZ_APPROX_MAX, right = flag & EZ_RIGHT;
		int w = J.w;
		if (w < 0) w = tlen > qlen ? tlen : qlen;
		const int T = (tlen + 15) / 16 * 16;
		int n_col = qlen < tlen ? qlen : tlen;
		n_col = (((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1) * 16;
		auto target_at = [&](int i) -> int { return i < tlen ? (int)bases.at(t_base + (uint64_t)(J.seq_rev ? tlen - 1 - i : i)) : 0; };
		auto query_at = [&](int j) -> int {
			if (j < 0 || j >= qlen) return 0;
			int pj = J.qs + (J.seq_rev ? qlen - 1 - j : j);
			if (!J.q_rev) return bases.at(q_base + (uint64_t)(pj));
			int c = bases.at(q_base + (uint64_t)(J.qlen_full - 1 - pj));
			return c < 4 ? 3 - c : 4;
		};
		// LDS rows: ring of R columns (R == T when the whole target fits the launch's ring capacity)
		int R = ((w < tlen ? w : tlen) + 15) / 16 * 16 + 96;
		if (R > T) R = T;
		if (R > r_cap) R = r_cap;                          // the launcher sized r_cap for every problem of the class
		int8_t *u = (int8_t*)dyn, *y = u + r_cap, *y2 = y + r_cap, *s = y2 + r_cap;
		int8_t *xb[2] = { s + r_cap, s + 2 * r_cap }, *vb[2] = { s + 3 * r_cap, s + 4 * r_cap }, *x2b[2] = { s + 5 * r_cap, s + 6 * r_cap };
		int32_t *H = (int32_t*)(s + 7 * r_cap);
		uint8_t *tq = (uint8_t*)(s + 7 * r_cap) + (exact_rows ? (size_t)4 * r_cap : 0), *qq = tq + seq_cap;   // sequences, when seq_cap > 0
		const bool seq_lds = seq_cap > 0;
		uint8_t *pmat = slab;
		uint32_t *cig_tmp = (uint32_t*)(pmat + (((size_t)(qlen + tlen - 1) * n_col + 15) & ~(size_t)15));
		const int init_n = R;
		for (int t = tid; t < init_n; t += WIDE_NT) {
			u[t] = y[t] = (int8_t)(-q - e); y2[t] = (int8_t)(-q2 - e2); s[t] = 0;
			xb[0][t] = xb[1][t] = vb[0][t] = vb[1][t] = (int8_t)(-q - e);
			x2b[0][t] = x2b[1][t] = (int8_t)(-q2 - e2);
			if (!approx_max) H[t] = KSW_NEG_INF;
		}
		const bool packed_ok = !right && w >= qlen && w >= tlen && R == T && sc_mch >= 0 && sc_mch < 127;
		const bool swar_profile = packed_ok && seq_lds;     // query stored REVERSED with 32 zero bytes on either side: column t of diagonal r reads qq[t + 32+qlen-1-r]
		if (seq_lds) {
			for (int i = tid; i < seq_cap; i += WIDE_NT) tq[i] = i < tlen ? (uint8_t)target_at(i) : (uint8_t)0;
			if (swar_profile) {
				for (int p = tid; p < seq_cap + 64; p += WIDE_NT) { const int j = qlen - 1 - (p - 32); qq[p] = (j >= 0 && j < qlen) ? (uint8_t)query_at(j) : (uint8_t)0; }
			} else for (int j = tid; j < qlen; j += WIDE_NT) qq[j] = (uint8_t)query_at(j);
		}
		int init_hi = R - 1;                               // highest column whose slot holds that column's state
		int ez_max = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1, ez_mte = KSW_NEG_INF, ez_mte_q = -1;
		int ez_score = KSW_NEG_INF, ez_zdropped = 0, ez_reach_end = 0;
		int H0 = 0, last_H0_t = 0, last_st = -1, last_en = -1;
		const int n_diag = qlen + tlen - 1;
		int r_done = 0;
		__syncthreads();

		int base = 0;                                       // multiple of R with base <= st-16: slot(t) = t-base (-R)
		int pend_slot = -1, pend_val = 0;
		// ---- unbanded approximate first passes (the 10 kb x 10 kb gap fills across rearrangements): ONE barrier per diagonal ----
		// Same packed two-column cell pass as below, but the score bytes are formed inside it (no separate profile pass), the first-row
		// values of the column that joins on this diagonal are taken by its owner directly (no store by thread 0 in front of a
		// barrier), and the two values the running corner score needs (v of the tracked column, u of its right neighbour) travel
		// through a parity-buffered mailbox.  What is left between two diagonals is the one barrier that orders the row arrays.
		const bool fused = packed_ok && swar_profile && approx_max && !(flag & EZ_APPROX_DROP) && !g_wide_no_fused;
		if (fused) {
			const s2_t ZERO = splat2(0), ONE = splat2(1), MCH = splat2(sc_mch), Q1 = splat2(q), Q2 = splat2(q2), QE = splat2(qe), QE2 = splat2(qe2);
			const s2_t EIGHT = splat2(8), C16 = splat2(16), C32 = splat2(32), C64 = splat2(64);
			for (int r = 0; r < n_diag; ++r) {
				r_done = r + 1;
				int st0, en0;
				diag_range_w(r, qlen, tlen, w, st0, en0);
				const int st = st0 / 16 * 16, en = (en0 + 16) / 16 * 16 - 1;
				const int8_t *xr = xb[r & 1], *vr = vb[r & 1], *x2r = x2b[r & 1];
				int8_t *xw = xb[(r + 1) & 1], *vw = vb[(r + 1) & 1], *x2w = x2b[(r + 1) & 1];
				int x1, x21, v1;
				if (st > 0) {
					if (st - 1 >= last_st && st - 1 <= last_en) { x1 = xr[st - 1], x21 = x2r[st - 1], v1 = vr[st - 1]; }
					else x1 = sx8w(-q - e), x21 = sx8w(-q2 - e2), v1 = sx8w(-q - e);
				} else {
					x1 = sx8w(-q - e), x21 = sx8w(-q2 - e2);
					v1 = r == 0 ? sx8w(-q - e) : r < long_thres ? sx8w(-e) : r == long_thres ? sx8w(long_diff) : sx8w(-e2);
				}
				const int u_join = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;   // first-row u of column r
				const int off_r = 32 + qlen - 1 - r;
				const int h0t = r == 0 ? 0 : last_H0_t;
				uint8_t *prow = pmat + (size_t)r * n_col - st;
				for (int t = st + 2 * tid; t <= en; t += 2 * WIDE_NT) {
					const int xl = t == st ? x1 : (int)xr[t - 1], vl = t == st ? v1 : (int)vr[t - 1], x2l = t == st ? x21 : (int)x2r[t - 1];
					const s2_t xt1 = pack2(xl, (int)xr[t]), vt1 = pack2(vl, (int)vr[t]), x2t1 = pack2(x2l, (int)x2r[t]);
					s2_t ut = unpack_i8x2(*reinterpret_cast<const uint16_t*>(u + t)), yt = unpack_i8x2(*reinterpret_cast<const uint16_t*>(y + t));
					s2_t y2t = unpack_i8x2(*reinterpret_cast<const uint16_t*>(y2 + t));
					if (t == r) { ut.x = (short)u_join; yt.x = (short)(-q - e); y2t.x = (short)(-q2 - e2); }
					else if (t + 1 == r) { ut.y = (short)u_join; yt.y = (short)(-q - e); y2t.y = (short)(-q2 - e2); }
					s2_t z;
					{
						const int a0 = tq[t], a1 = tq[t + 1], b0 = qq[t + off_r], b1 = qq[t + 1 + off_r];
						z.x = (short)(((a0 | b0) & 4) ? sc_N : a0 == b0 ? sc_mch : sc_mis);
						z.y = (short)(((a1 | b1) & 4) ? sc_N : a1 == b1 ? sc_mch : sc_mis);
					}
					s2_t a = xt1 + vt1, b = yt + ut, a2 = x2t1 + vt1, b2 = y2t + ut;
					const s2_t zm = pmax(pmax(pmax(z, a), pmax(b, a2)), b2);
					s2_t d;
					{
						const s2_t n0 = pminu(zm - z, ONE), n1 = pminu(zm - a, ONE), n2 = pminu(zm - b, ONE), n3 = pminu(zm - a2, ONE);
						d = n0 * (ONE + n1 * (ONE + n2 * (ONE + n3)));
					}
					z = pmin(zm, MCH);
					const s2_t un = z - vt1, vn = z - ut;
					s2_t tmp = z - Q1; a = a - tmp; b = b - tmp;
					tmp = z - Q2; a2 = a2 - tmp; b2 = b2 - tmp;
					s2_t xn, yn, x2n, y2n;
					{ const s2_t m = pmax(a, ZERO);  xn  = m - QE;  d = d + pmin(m, ONE) * EIGHT; }
					{ const s2_t m = pmax(b, ZERO);  yn  = m - QE;  d = d + pmin(m, ONE) * C16; }
					{ const s2_t m = pmax(a2, ZERO); x2n = m - QE2; d = d + pmin(m, ONE) * C32; }
					{ const s2_t m = pmax(b2, ZERO); y2n = m - QE2; d = d + pmin(m, ONE) * C64; }
					const uint16_t un8 = pack_i8x2(un), vn8 = pack_i8x2(vn);
					*reinterpret_cast<uint16_t*>(u + t) = un8; *reinterpret_cast<uint16_t*>(vw + t) = vn8;
					*reinterpret_cast<uint16_t*>(xw + t) = pack_i8x2(xn); *reinterpret_cast<uint16_t*>(y + t) = pack_i8x2(yn);
					*reinterpret_cast<uint16_t*>(x2w + t) = pack_i8x2(x2n); *reinterpret_cast<uint16_t*>(y2 + t) = pack_i8x2(y2n);
					*reinterpret_cast<uint16_t*>(prow + t) = pack_i8x2(d);
					// the stored int8 values (what the corner tracker reads back)
					if (t == h0t) s_h0v[r & 1] = (int)(int8_t)(vn8 & 0xff); else if (t + 1 == h0t) s_h0v[r & 1] = (int)(int8_t)(vn8 >> 8);
					if (t == h0t + 1) s_h0u[r & 1] = (int)(int8_t)(un8 & 0xff); else if (t + 1 == h0t + 1) s_h0u[r & 1] = (int)(int8_t)(un8 >> 8);
				}
				// the barrier orders the LDS rows only: the direction bytes are fire-and-forget (fenced before the backtrack)
				asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
				if (r > 0) {
					if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
						const int d0 = s_h0v[r & 1], d1 = s_h0u[r & 1];
						if (d0 > d1) H0 += d0; else H0 += d1, ++last_H0_t;
					} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += s_h0v[r & 1];
					else ++last_H0_t, H0 += s_h0u[r & 1];
				} else H0 = s_h0v[0] - qe_h, last_H0_t = 0;
				if (r == n_diag - 1 && en0 == tlen - 1) ez_score = H0;
				last_st = st, last_en = en;
			}
		} else
		for (int r = 0; r < n_diag; ++r) {
			r_done = r + 1;
			if (pend_slot >= 0) { if (tid == 0) H[pend_slot] = pend_val; pend_slot = -1; }   // last diagonal's H[en0] (nobody reads H before two more barriers)
			int st0, en0;
			diag_range_w(r, qlen, tlen, w, st0, en0);
			if (st0 > en0) { ez_zdropped = 1; break; }
			const int st = st0 / 16 * 16, en = (en0 + 16) / 16 * 16 - 1;
			while (st - 16 >= base + R) base += R;
			auto sl = [&](int t) { const int x = t - base; return x >= R ? x - R : x; };
			const int span = ((en0 - st0) / 16 + 1) * 16;
			// columns entering the band take over the slots of columns that left it long ago
			{
				int need_hi = en > st0 + span - 1 ? en : st0 + span - 1;
				if (need_hi > T - 1) need_hi = T - 1;
				if (need_hi > init_hi) {
					for (int t = init_hi + 1 + tid; t <= need_hi; t += WIDE_NT) {
						const int k = sl(t);
						u[k] = y[k] = (int8_t)(-q - e); y2[k] = (int8_t)(-q2 - e2); s[k] = 0;
						xb[0][k] = xb[1][k] = vb[0][k] = vb[1][k] = (int8_t)(-q - e);
						x2b[0][k] = x2b[1][k] = (int8_t)(-q2 - e2);
						if (!approx_max) H[k] = KSW_NEG_INF;
					}
					init_hi = need_hi;
					__syncthreads();
				}
			}
			const int8_t *xr = xb[r & 1], *vr = vb[r & 1], *x2r = x2b[r & 1];
			int8_t *xw = xb[(r + 1) & 1], *vw = vb[(r + 1) & 1], *x2w = x2b[(r + 1) & 1];
			int x1, x21, v1;
			if (st > 0) {
				if (st - 1 >= last_st && st - 1 <= last_en) { const int k = sl(st - 1); x1 = xr[k], x21 = x2r[k], v1 = vr[k]; }
				else x1 = sx8w(-q - e), x21 = sx8w(-q2 - e2), v1 = sx8w(-q - e);
			} else {
				x1 = sx8w(-q - e), x21 = sx8w(-q2 - e2);
				v1 = r == 0 ? sx8w(-q - e) : r < long_thres ? sx8w(-e) : r == long_thres ? sx8w(long_diff) : sx8w(-e2);
			}
			if (en >= r && tid == 0) {
				const int k = sl(r);
				y[k] = (int8_t)(-q - e), y2[k] = (int8_t)(-q2 - e2);
				u[k] = (int8_t)(r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2);
			}
			if (swar_profile) {
				// four columns per thread and iteration on 32-bit words: equal bytes <=> zero bytes of a ^ b (all values <= 4, so +0x7f sets
				// bit 7 exactly for non-zero bytes without carries); the score bytes are picked by v_perm from {match, mismatch, N, N}.
				// (writes may spill a few columns beyond the reference's 16-column blocks: unbanded problems never read those)
				const int off_r = 32 + qlen - 1 - r;
				const uint32_t sc_tab = (uint32_t)(uint8_t)sc_mch | (uint32_t)(uint8_t)sc_mis << 8 | (uint32_t)(uint8_t)sc_N << 16 | (uint32_t)(uint8_t)sc_N << 24;
				for (int t4 = (st0 & ~3) + 4 * tid; t4 < st0 + span && t4 < T; t4 += 4 * WIDE_NT) {
					const uint32_t a4 = *reinterpret_cast<const uint32_t*>(tq + t4);
					const int idx = t4 + off_r;
					const uint32_t lo = *reinterpret_cast<const uint32_t*>(qq + (idx & ~3)), hi = *reinterpret_cast<const uint32_t*>(qq + (idx & ~3) + 4);
					const uint32_t b4 = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(idx & 3));
					const uint32_t nz = ((a4 ^ b4) + 0x7f7f7f7fu) >> 7 & 0x01010101u;
					const uint32_t nn = (a4 | b4) >> 2 & 0x01010101u;
					*reinterpret_cast<uint32_t*>(s + t4) = __builtin_amdgcn_perm(0u, sc_tab, nz | nn << 1);
				}
			} else
			for (int o = tid; o < span; o += WIDE_NT) {
				const int t = st0 + o;
				if (t < T) {
					int a, b;
					if (seq_lds) { a = t < tlen ? (int)tq[t] : 0; const int j = r - t; b = (j >= 0 && j < qlen) ? (int)qq[j] : 0; }
					else a = target_at(t), b = query_at(r - t);
					int sc = a == b ? sc_mch : sc_mis;
					if (a == 4 || b == 4) sc = sc_N;
					s[sl(t)] = (int8_t)sc;
				}
			}
			__syncthreads();
			uint8_t *prow = pmat + (size_t)r * n_col - st;
			if (packed_ok) {
				// unbanded, left-aligned gaps, whole target in the ring (slot == column): two adjacent columns per thread in packed
				// 16-bit arithmetic.  No int8 wrap is emulated here: inside an unbinding band every cell in [st0,en0] depends only on
				// cells in range or on boundary values (see pga_ksw_fast.hip), and those stay within int8 by construction.
				const s2_t ZERO = splat2(0), ONE = splat2(1), MCH = splat2(sc_mch), Q1 = splat2(q), Q2 = splat2(q2), QE = splat2(qe), QE2 = splat2(qe2);
				const s2_t EIGHT = splat2(8), C16 = splat2(16), C32 = splat2(32), C64 = splat2(64);
				for (int t = st + 2 * tid; t <= en; t += 2 * WIDE_NT) {
					const int xl = t == st ? x1 : (int)xr[t - 1], vl = t == st ? v1 : (int)vr[t - 1], x2l = t == st ? x21 : (int)x2r[t - 1];
					const s2_t xt1 = pack2(xl, (int)xr[t]), vt1 = pack2(vl, (int)vr[t]), x2t1 = pack2(x2l, (int)x2r[t]);
					const s2_t ut = unpack_i8x2(*reinterpret_cast<const uint16_t*>(u + t)), yt = unpack_i8x2(*reinterpret_cast<const uint16_t*>(y + t));
					const s2_t y2t = unpack_i8x2(*reinterpret_cast<const uint16_t*>(y2 + t));
					s2_t z = unpack_i8x2(*reinterpret_cast<const uint16_t*>(s + t));
					s2_t a = xt1 + vt1, b = yt + ut, a2 = x2t1 + vt1, b2 = y2t + ut;
					const s2_t zm = pmax(pmax(pmax(z, a), pmax(b, a2)), b2);
					s2_t d;
					{
						const s2_t n0 = pminu(zm - z, ONE), n1 = pminu(zm - a, ONE), n2 = pminu(zm - b, ONE), n3 = pminu(zm - a2, ONE);
						d = n0 * (ONE + n1 * (ONE + n2 * (ONE + n3)));
					}
					z = pmin(zm, MCH);
					const s2_t un = z - vt1, vn = z - ut;
					s2_t tmp = z - Q1; a = a - tmp; b = b - tmp;
					tmp = z - Q2; a2 = a2 - tmp; b2 = b2 - tmp;
					s2_t xn, yn, x2n, y2n;
					{ const s2_t m = pmax(a, ZERO);  xn  = m - QE;  d = d + pmin(m, ONE) * EIGHT; }
					{ const s2_t m = pmax(b, ZERO);  yn  = m - QE;  d = d + pmin(m, ONE) * C16; }
					{ const s2_t m = pmax(a2, ZERO); x2n = m - QE2; d = d + pmin(m, ONE) * C32; }
					{ const s2_t m = pmax(b2, ZERO); y2n = m - QE2; d = d + pmin(m, ONE) * C64; }
					*reinterpret_cast<uint16_t*>(u + t) = pack_i8x2(un); *reinterpret_cast<uint16_t*>(vw + t) = pack_i8x2(vn);
					*reinterpret_cast<uint16_t*>(xw + t) = pack_i8x2(xn); *reinterpret_cast<uint16_t*>(y + t) = pack_i8x2(yn);
					*reinterpret_cast<uint16_t*>(x2w + t) = pack_i8x2(x2n); *reinterpret_cast<uint16_t*>(y2 + t) = pack_i8x2(y2n);
					*reinterpret_cast<uint16_t*>(prow + t) = pack_i8x2(d);
				}
			} else
			for (int t = st + tid; t <= en; t += WIDE_NT) {
				const int k = sl(t), k1 = t == st ? k : sl(t - 1);
				const int xt1 = t == st ? x1 : (int)xr[k1], vt1 = t == st ? v1 : (int)vr[k1], x2t1 = t == st ? x21 : (int)x2r[k1];
				const int ut = u[k], yo = y[k], y2o = y2[k];
				int z = s[k];
				int a = sx8w(xt1 + vt1), b = sx8w(yo + ut), a2 = sx8w(x2t1 + vt1), b2 = sx8w(y2o + ut), d;
				if (!right) {
					d = 0;
					if (a > z) d = 1, z = a;
					if (b > z) d = 2, z = b;
					if (a2 > z) d = 3, z = a2;
					if (b2 > z) d = 4, z = b2;
				} else {
					d = z > a ? 0 : 1;  z = z > a ? z : a;
					d = z > b ? d : 2;  z = z > b ? z : b;
					d = z > a2 ? d : 3; z = z > a2 ? z : a2;
					d = z > b2 ? d : 4; z = z > b2 ? z : b2;
				}
				if (sc_mch < z) z = sc_mch;
				u[k] = (int8_t)(z - vt1), vw[k] = (int8_t)(z - ut);
				int tmp = sx8w(z - q); a = sx8w(a - tmp), b = sx8w(b - tmp);
				tmp = sx8w(z - q2); a2 = sx8w(a2 - tmp), b2 = sx8w(b2 - tmp);
				if (!right) {
					xw[k]  = (int8_t)((a  > 0 ? a  : 0) - qe);  if (a  > 0) d |= 0x08;
					y[k]   = (int8_t)((b  > 0 ? b  : 0) - qe);  if (b  > 0) d |= 0x10;
					x2w[k] = (int8_t)((a2 > 0 ? a2 : 0) - qe2); if (a2 > 0) d |= 0x20;
					y2[k]  = (int8_t)((b2 > 0 ? b2 : 0) - qe2); if (b2 > 0) d |= 0x40;
				} else {
					xw[k]  = (int8_t)((0 > a  ? 0 : a)  - qe);  if (!(0 > a))  d |= 0x08;
					y[k]   = (int8_t)((0 > b  ? 0 : b)  - qe);  if (!(0 > b))  d |= 0x10;
					x2w[k] = (int8_t)((0 > a2 ? 0 : a2) - qe2); if (!(0 > a2)) d |= 0x20;
					y2[k]  = (int8_t)((0 > b2 ? 0 : b2) - qe2); if (!(0 > b2)) d |= 0x40;
				}
				prow[t] = (uint8_t)d;
			}
			__syncthreads();
			bool stop = false;
			if (!approx_max) {
				int max_H, max_t, h_en_now = 0, h_st_now = 0;
				if (r > 0) {
					// H[t] += v[t] for st0 <= t < en0, H[en0] = H[en0-1](before its update) + u[en0]  (ksw2_extd2_sse.c:325-340).
					// The old H[en0-1] is posted by the thread that updates it, so the update loop needs no barrier in front of it.
					const int en1 = st0 + (en0 - st0) / 4 * 4;
					long long best = (long long)KSW_NEG_INF * 4294967296LL;
					for (int t = st0 + tid; t < en0; t += WIDE_NT) {
						const int k = sl(t);
						const int hold = H[k], h = hold + vw[k];
						if (t == en0 - 1) s_hprev = hold;
						H[k] = h;
						const unsigned ord = t < en1 ? 1u + ((unsigned)((t - st0) & 3) << 28) + (unsigned)t : 1u + (4u << 28) + (unsigned)t;
						const long long key = ((long long)h << 32) | (0xffffffffu - ord);
						best = key > best ? key : best;
					}
					best = wave_max_i64(best);
					if (lane == 0) s_part[wave] = best;
					__syncthreads();
					const int hprev = en0 > 0 ? (en0 - 1 >= st0 ? s_hprev : H[sl(en0 - 1)]) : 0;
					const int Hen = en0 > 0 ? hprev + u[sl(en0)] : H[sl(en0)] + vw[sl(en0)];
					long long bb = lane < WIDE_NT / 64 ? s_part[lane] : (long long)KSW_NEG_INF * 4294967296LL;
					bb = wave_max_i64(bb);
					{ const long long hk = ((long long)Hen << 32) | 0xffffffffu; if (hk > bb) bb = hk; }
					max_H = (int)(bb >> 32);
					const unsigned ord = 0xffffffffu - (unsigned)(bb & 0xffffffffLL);
					max_t = ord == 0 ? en0 : (int)((ord - 1) & 0x0fffffffu);
					h_en_now = Hen;
					h_st_now = st0 == en0 ? Hen : H[sl(st0)];
					if (en0 == 0) __syncthreads();                    // (one-column target: H[0] itself was an input above)
					pend_slot = sl(en0); pend_val = Hen;             // H[en0] is replaced at the top of the next diagonal, behind its first barrier
				} else {
					const int h0 = vw[sl(0)] - qe_h;
					pend_slot = sl(0); pend_val = h0;
					max_H = h0, max_t = 0; h_en_now = h0; h_st_now = h0;
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
						const int d0 = vw[sl(last_H0_t)], d1 = u[sl(last_H0_t + 1)];
						if (d0 > d1) H0 += d0; else H0 += d1, ++last_H0_t;
					} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += vw[sl(last_H0_t)];
					else ++last_H0_t, H0 += u[sl(last_H0_t)];
				} else H0 = vw[sl(0)] - qe_h, last_H0_t = 0;
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
			// no barrier here: what the next diagonal writes before its own first barrier (score profile, entering columns, the
			// joining column's first-row values, the pending H[en0]) is read by nobody in the phase above
		}

		// ---- backtrack by wave 0 (ksw2.h:127-159), LDS window, fences only ----
		int n_cigar = 0, bi = -1, bj = -1;
		if (!ez_zdropped && !(flag & EZ_EXTZ_ONLY)) bi = tlen - 1, bj = qlen - 1;
		else if (!ez_zdropped && (flag & EZ_EXTZ_ONLY) && ez_mqe + end_bonus > ez_max) ez_reach_end = 1, bi = ez_mqe_t, bj = qlen - 1;
		else if (ez_max_t >= 0 && ez_max_q >= 0) bi = ez_max_t, bj = ez_max_q;
		__threadfence_block();
		__syncthreads();
		if (wave == 0) {
			int i = bi, j = bj, state = 0; long long guard = 0;
			uint32_t last_op = 0xffffffffu;
			uint32_t run_len = 0;                               // the operation being extended lives in registers: one store per operation, not a
			auto cg_push = [&](uint32_t op, uint32_t len) {     // read-modify-write of device memory per path step
				if (op == last_op) { run_len += len; return; }
				if (last_op != 0xffffffffu) { if (lane == 0) cig_tmp[n_cigar] = run_len << 4 | last_op; ++n_cigar; }
				last_op = op; run_len = len;
			};
			auto cg_flush = [&] { if (last_op != 0xffffffffu && n_cigar >= 0) { if (lane == 0) cig_tmp[n_cigar] = run_len << 4 | last_op; ++n_cigar; last_op = 0xffffffffu; } };
			while (i >= 0 && j >= 0) {
				if (++guard > 4000000) { n_cigar = -7; break; }
				const int r_hi = i + j, c_lo = i - (WBT - 1);
				{
					// all 64 rows of the window are requested before the first one is stored: 64 loads in flight instead of 64 round trips
					uint8_t wv[WBT];
#pragma unroll
					for (int row = 0; row < WBT; ++row) {
						const int r = r_hi - row, col = c_lo + lane;
						uint8_t val = 0;
						if (r >= 0 && col >= 0) {
							int st0, en0; diag_range_w(r, qlen, tlen, w, st0, en0);
							const int off = st0 / 16 * 16, off_end = (en0 + 16) / 16 * 16 - 1;
							if (st0 <= en0 && col >= off && col <= off_end) val = pmat[(size_t)r * n_col + (col - off)];
						}
						wv[row] = val;
					}
#pragma unroll
					for (int row = 0; row < WBT; ++row) s_win[row * WBT + lane] = wv[row];
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				while (i >= 0 && j >= 0) {
					const int r = i + j, row = r_hi - r;
					if (row >= WBT || i < c_lo) break;
					int st0, en0; diag_range_w(r, qlen, tlen, w, st0, en0);
					const int off = st0 / 16 * 16, off_end = (en0 + 16) / 16 * 16 - 1;
					int force_state = -1;
					if (i < off) force_state = 2;
					if (i > off_end) force_state = 1;
					const uint32_t tmp = force_state < 0 ? s_win[row * WBT + (i - c_lo)] : 0;
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

size_t wide_lds_bytes(int r_cap, int seq_cap, bool exact) { return (size_t)r_cap * 10 + (exact ? (size_t)r_cap * 4 : 0) + (seq_cap > 0 ? 2 * (size_t)seq_cap + 128 : 0); }

template <int NT> static void launch_wide_nt(unsigned n_blocks, size_t lds, hipStream_t st, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter, uint8_t *slab,
                                             size_t slab_bytes, int r_cap, int seq_cap, int exact, DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap)
{
	{	// a per-DEVICE function attribute, set once per device whatever thread comes first
		static std::mutex mu; static bool attr_set[64] = {};
		int dev = 0; PGA_HIP(hipGetDevice(&dev));
		std::lock_guard<std::mutex> lk(mu);
		if (!attr_set[dev & 63]) { PGA_HIP(hipFuncSetAttribute((const void*)k_extd2_wide<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS_MAX)); attr_set[dev & 63] = true; }
	}
	hipLaunchKernelGGL(k_extd2_wide<NT>, dim3(n_blocks), dim3(NT), lds, st, jobs, n_jobs, bases, P, counter, slab, slab_bytes, r_cap, seq_cap, exact, res, pool, cursor, pool_cap);
}

// n_threads: 256 for many problems (several workgroups per CU), 512 / 1024 when a class holds few, large problems:
// a workgroup that has a CU to itself needs the extra waves to hide its LDS latency
void launch_extd2_wide(unsigned n_blocks, int n_threads, int r_cap, int seq_cap, bool exact, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter, uint8_t *slab, size_t slab_bytes,
                       DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st)
{
	const size_t lds = wide_lds_bytes(r_cap, seq_cap, exact);
	static const bool no_fused = [] { const bool off = getenv("PGA_NO_FUSED_APPROX") != nullptr; if (off) { const int one = 1; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wide_no_fused), &one, sizeof(int)); } return off; }();
	(void)no_fused;
	if (n_threads >= 1024) launch_wide_nt<1024>(n_blocks, lds, st, jobs, n_jobs, bases, P, counter, slab, slab_bytes, r_cap, seq_cap, exact ? 1 : 0, res, pool, cursor, pool_cap);
	else if (n_threads >= 512) launch_wide_nt<512>(n_blocks, lds, st, jobs, n_jobs, bases, P, counter, slab, slab_bytes, r_cap, seq_cap, exact ? 1 : 0, res, pool, cursor, pool_cap);
	else launch_wide_nt<256>(n_blocks, lds, st, jobs, n_jobs, bases, P, counter, slab, slab_bytes, r_cap, seq_cap, exact ? 1 : 0, res, pool, cursor, pool_cap);
}

} // namespace pga
