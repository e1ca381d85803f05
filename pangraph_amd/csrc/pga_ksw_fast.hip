// pga_ksw_fast.hip -- kernel #5a: the register-resident variant of the dual-affine DP for problems whose band
// never binds (w >= qlen and w >= tlen) and whose target fits 64*C lanes: the ~200x200 gap-fill tiles between
// adjacent chain anchors, i.e. >95 % of all ksw_extd2_sse calls (median 207x207).
//
// Same recurrence and the same direction bytes as pga_ksw.hip / ksw2_extd2_sse.c:131-386, but
//   * one wavefront per problem, lane l owns target coordinates t = l + 64*c (c < C); the six difference values,
//     the exact-mode H and the lane's target base live in VGPRs for the whole problem: no LDS rows, no barriers;
//   * the t-1 neighbour is a wave shuffle (row rotate) with a cross-chunk carry read from lane 63 of the chunk
//     below BEFORE it is updated (chunks are swept high-to-low);
//   * the query base a lane needs on diagonal r is the one its left neighbour used on r-1, so query bases flow
//     through the lanes by the same shuffle and only lane 0 takes a new base per diagonal (from a 64-base
//     register block refilled by one coalesced load every 64 diagonals);
//   * with an unbinding band every cell inside [st0,en0] depends only on cells inside the previous diagonal's
//     range or on the explicit boundary values (ksw2_extd2_sse.c:146-163), so lanes outside the range -- which the
//     SSE code computes as a by-product of its 16-lane rounding -- are simply masked off;
//   * direction bytes stream to the wave's HBM slab (64 coalesced bytes per chunk and diagonal); the backtrack
//     pulls a 64-row x 128-column window around the path into LDS with coalesced loads and walks it there.
#include "pga_common.h"
#include "pga_dp.h"
#include "pga_wave.h"
#include "pga_pk16.h"

namespace pga {

#define KSW_NEG_INF (-0x40000000)
#define EZ_RIGHT      0x02
#define EZ_APPROX_MAX 0x08
#define EZ_EXTZ_ONLY  0x40
#define EZ_REV_CIGAR  0x80

__device__ __forceinline__ long long wave_max64f(long long v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		int lo = __shfl_xor((int)(v & 0xffffffffLL), d), hi = __shfl_xor((int)(v >> 32), d);
		long long o = ((long long)hi << 32) | (unsigned int)lo;
		v = o > v ? o : v;
	}
	return v;
}

#define BT_ROWS 64
#define BT_COLS 64

template <int C>
__global__ __launch_bounds__(64)
void k_extd2_fast(const DpJob *__restrict__ jobs, uint32_t n_jobs, PkBases bases, DpParams P,
                  uint32_t *__restrict__ job_counter, uint8_t *__restrict__ slab_all, size_t slab_bytes,
                  DpRes *__restrict__ res, uint32_t *__restrict__ cigar_pool, unsigned long long *__restrict__ pool_cursor, unsigned long long pool_cap)
{
	__shared__ uint8_t s_win[BT_ROWS * BT_COLS];
	const int lane = threadIdx.x;
	uint8_t *slab = slab_all + (size_t)blockIdx.x * slab_bytes;
	int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
	const int qe_h = q + e;
	if (q2 + e2 < q + e) { int t = q; q = q2, q2 = t, t = e, e = e2, e2 = t; }
	const int qe = q + e, qe2 = q2 + e2;
	const int sc_mch = P.sc_mch, sc_mis = P.sc_mis, sc_N = P.sc_ambi == 0 ? -e2 : P.sc_ambi;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;

	long long guard = 0;
	for (;;) {
		uint32_t jid = 0;
		if (lane == 0) jid = atomicAdd(job_counter, 1u);
		jid = (uint32_t)__shfl((int)jid, 0);
		if (jid >= n_jobs) break;
		const DpJob J = jobs[jid];
		const uint64_t t_base = J.t_off, q_base = J.q_off;          // base positions in the packed store
		const int qlen = J.qlen, tlen = J.tlen, flag = J.flag, zdrop = J.zdrop, end_bonus = J.end_bonus;
		const bool approx_max = flag & EZ_APPROX_MAX, right = flag & EZ_RIGHT;
		const int w = J.w;
		int n_col = qlen < tlen ? qlen : tlen;
		n_col = (((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1) * 16;
		uint8_t *pmat = slab;
		uint32_t *cig_tmp = (uint32_t*)(pmat + (((size_t)(qlen + tlen - 1) * n_col + 15) & ~(size_t)15));

		auto target_at = [&](int i) -> int { return i < tlen ? (int)bases.at(t_base + (uint64_t)(J.seq_rev ? tlen - 1 - i : i)) : 0; };
		auto query_at = [&](int j) -> int {
			if (j < 0 || j >= qlen) return 0;
			int pj = J.qs + (J.seq_rev ? qlen - 1 - j : j);
			if (!J.q_rev) return bases.at(q_base + (uint64_t)(pj));
			int c = bases.at(q_base + (uint64_t)(J.qlen_full - 1 - pj));
			return c < 4 ? 3 - c : 4;
		};

		int u[C], v[C], x[C], y[C], x2[C], y2[C], tb[C], qb[C], H[C];
#pragma unroll
		for (int c = 0; c < C; ++c) {
			u[c] = v[c] = x[c] = y[c] = -q - e; x2[c] = y2[c] = -q2 - e2;
			tb[c] = target_at(lane + 64 * c); qb[c] = 0; H[c] = KSW_NEG_INF;
		}
		// Gap fills between near-identical stretches, decided without the matrix: for two equally long sequences that differ
		// in m positions (no ambiguous bases), every alignment other than the main diagonal has at least one insertion and one
		// deletion, so it scores at most a*(n-1) - 2*min(q+e, q2+e2), while the diagonal scores a*n - (a+b)*m.  If the
		// diagonal wins STRICTLY it is the unique optimum; the recurrence is exact inside an unbinding band and its
		// left-aligned tie rule never fires, so the first approximate pass (KSW_EZ_APPROX_MAX only: no z-drop test, score
		// = H at the end cell) returns exactly "nM" with that score.
		if (flag == EZ_APPROX_MAX && qlen == tlen) {
			int n_mis = 0; bool ambi = false;
#pragma unroll
			for (int c = 0; c < C; ++c) {
				const int t = lane + 64 * c;
				if (t < tlen) { const int qc = query_at(t); ambi |= qc > 3 || tb[c] > 3; n_mis += qc != tb[c]; }
			}
			const unsigned long long any_ambi = __ballot(ambi);
			int m_tot = 0;                                                // sum over lanes (a lane holds at most C columns)
#pragma unroll
			for (int k = 1; k <= C; ++k) m_tot += __popcll(__ballot(n_mis >= k));
			const int g1 = qe < qe2 ? qe : qe2;
			if (!any_ambi && (sc_mch - sc_mis) * m_tot < sc_mch + 2 * g1) {
				unsigned long long base = 0;
				if (lane == 0) base = atomicAdd(pool_cursor, 1ULL);
				if (lane == 0) {
					if (base + 1 <= pool_cap) cigar_pool[base] = (uint32_t)tlen << 4;
					DpRes R;
					R.max = 0, R.max_q = -1, R.max_t = -1, R.mqe = KSW_NEG_INF, R.mqe_t = -1, R.mte = KSW_NEG_INF, R.mte_q = -1;
					R.score = sc_mch * (tlen - m_tot) + sc_mis * m_tot, R.zdropped = 0, R.reach_end = 0, R.n_cigar = 1, R.pad = 0x5A /* answered without the matrix */, R.cigar_off = base;
					res[jid] = R;
				}
				continue;
			}
		}
		int ez_max = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1, ez_mte = KSW_NEG_INF, ez_mte_q = -1;
		int ez_score = KSW_NEG_INF, ez_zdropped = 0, ez_reach_end = 0;
		if (flag == EZ_APPROX_MAX) {
			// ---- the bulk: first-pass gap fills (approximate mode, left-aligned gaps, no z-drop test), two chunks per VGPR ----
			// Lane l holds columns l+128p (low half) and l+128p+64 (high half) of pair p; all arithmetic is packed 16-bit
			// (values stay inside the reference's int8 range for every cell that matters).  Cells outside [st0,en0] are
			// computed too and hold garbage: with an unbinding band no cell inside the range ever reads them (see the file
			// header), and they never store a direction byte.
			constexpr int NP = C / 2;
			s2_t U[NP], V[NP], X[NP], Y[NP], X2[NP], Y2[NP], TB[NP], QB[NP];
#pragma unroll
			for (int p = 0; p < NP; ++p) {
				U[p] = V[p] = X[p] = Y[p] = splat2(-q - e); X2[p] = Y2[p] = splat2(-q2 - e2);
				TB[p].x = (short)tb[2 * p]; TB[p].y = (short)tb[2 * p + 1]; QB[p] = splat2(0);
			}
			int qblock = query_at(lane);
			{ const int q0 = __builtin_amdgcn_readlane(qblock, 0); if (lane == 0) QB[0].x = (short)q0; }
			const s2_t ZERO = splat2(0), ONE = splat2(1), MCH = splat2(sc_mch), DMIS = splat2(sc_mis - sc_mch), SCN = splat2(sc_N), Q1 = splat2(q), Q2 = splat2(q2), QE = splat2(qe), QE2 = splat2(qe2);
			const s2_t INIT1 = splat2(-q - e), INIT2 = splat2(-q2 - e2), EIGHT = splat2(8), C16 = splat2(16), C32 = splat2(32), C64 = splat2(64);
			int H0 = 0, last_H0_t = 0;
			const int n_diag = qlen + tlen - 1;
			for (int r = 0; r < n_diag; ++r) {
				const int st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0, en0 = r < tlen - 1 ? r : tlen - 1;
				const int st = st0 & ~15;
				const int bnd = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
				const s2_t BND = splat2(bnd);
				uint8_t *prow = pmat + (size_t)r * n_col - st;
#pragma unroll
				for (int p = NP - 1; p >= 0; --p) {
					if (128 * p > en0 || 128 * p + 127 < st0) continue;          // wave-uniform
					const int t_lo = lane + 128 * p, t_hi = t_lo + 64;
					// t-1 neighbours (pre-update): low half <- lane 63 of the pair below (its high half) or the boundary; high half <- own low half
					const unsigned ox = (unsigned)__builtin_amdgcn_readlane(as_i(X[p]), 63), ov = (unsigned)__builtin_amdgcn_readlane(as_i(V[p]), 63), ox2 = (unsigned)__builtin_amdgcn_readlane(as_i(X2[p]), 63);
					unsigned px = (unsigned)(-q - e) & 0xffffu, pv = (unsigned)bnd & 0xffffu, px2 = (unsigned)(-q2 - e2) & 0xffffu;
					if (p > 0) {
						px = (unsigned)__builtin_amdgcn_readlane(as_i(X[p > 0 ? p - 1 : 0]), 63) >> 16;
						pv = (unsigned)__builtin_amdgcn_readlane(as_i(V[p > 0 ? p - 1 : 0]), 63) >> 16;
						px2 = (unsigned)__builtin_amdgcn_readlane(as_i(X2[p > 0 ? p - 1 : 0]), 63) >> 16;
					}
					const s2_t xt1 = as_s2(wave_shr1(as_i(X[p]), (int)(px | ox << 16)));
					const s2_t vt1 = as_s2(wave_shr1(as_i(V[p]), (int)(pv | ov << 16)));
					const s2_t x2t1 = as_s2(wave_shr1(as_i(X2[p]), (int)(px2 | ox2 << 16)));
					// first row / first column values for the column that joins on this diagonal (ksw2_extd2_sse.c:160-163)
					const int jm = (t_lo == r ? 0xffff : 0) | (t_hi == r ? (int)0xffff0000 : 0);
					const s2_t ut = as_s2(bfi(jm, as_i(BND), as_i(U[p]))), yt = as_s2(bfi(jm, as_i(INIT1), as_i(Y[p]))), y2t = as_s2(bfi(jm, as_i(INIT2), as_i(Y2[p])));
					// score profile
					const s2_t dif = as_s2(as_i(TB[p]) ^ as_i(QB[p]));
					s2_t z = MCH + pminu(dif, ONE) * DMIS;
					{ const s2_t nf = pminu(as_s2((as_i(TB[p]) | as_i(QB[p])) >> 2 & 0x00010001), ONE); z = z + nf * (SCN - z); }
					s2_t a = xt1 + vt1, b = yt + ut, a2 = x2t1 + vt1, b2 = y2t + ut;
					const s2_t zm = pmax(pmax(pmax(z, a), pmax(b, a2)), b2);
					// direction: the first of (z, a, b, a2, b2) that reaches the maximum (strict > in the reference)
					s2_t d;
					{
						const s2_t n0 = pminu(zm - z, ONE), n1 = pminu(zm - a, ONE), n2 = pminu(zm - b, ONE), n3 = pminu(zm - a2, ONE);
						d = n0 * (ONE + n1 * (ONE + n2 * (ONE + n3)));
					}
					z = pmin(zm, MCH);
					U[p] = z - vt1; V[p] = z - ut;
					s2_t tmp = z - Q1; a = a - tmp; b = b - tmp;
					tmp = z - Q2; a2 = a2 - tmp; b2 = b2 - tmp;
					{ const s2_t m = pmax(a, ZERO);  X[p]  = m - QE;  d = d + pmin(m, ONE) * EIGHT; }
					{ const s2_t m = pmax(b, ZERO);  Y[p]  = m - QE;  d = d + pmin(m, ONE) * C16; }
					{ const s2_t m = pmax(a2, ZERO); X2[p] = m - QE2; d = d + pmin(m, ONE) * C32; }
					{ const s2_t m = pmax(b2, ZERO); Y2[p] = m - QE2; d = d + pmin(m, ONE) * C64; }
					if (t_lo >= st0 && t_lo <= en0) prow[t_lo] = (uint8_t)d.x;
					if (t_hi >= st0 && t_hi <= en0) prow[t_hi] = (uint8_t)d.y;
				}
				// H along the approximate path (ksw2_extd2_sse.c:367-384); no z-drop test without KSW_EZ_APPROX_DROP
				if (r > 0) {
					int d0 = 0, d1 = 0;
					{
						const int c0 = last_H0_t >> 6, l0 = last_H0_t & 63, c1 = (last_H0_t + 1) >> 6, l1 = (last_H0_t + 1) & 63;
#pragma unroll
						for (int p = 0; p < NP; ++p) {
							if ((c0 >> 1) == p) { const int w2 = rl(as_i(V[p]), l0); d0 = (c0 & 1) ? w2 >> 16 : (int)(short)(w2 & 0xffff); }
							if ((c1 >> 1) == p) { const int w2 = rl(as_i(U[p]), l1); d1 = (c1 & 1) ? w2 >> 16 : (int)(short)(w2 & 0xffff); }
						}
					}
					if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
						if (d0 > d1) H0 += d0; else H0 += d1, ++last_H0_t;
					} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += d0;
					else ++last_H0_t, H0 += d1;
				} else { const int w2 = __builtin_amdgcn_readlane(as_i(V[0]), 0); H0 = (int)(short)(w2 & 0xffff) - qe_h; last_H0_t = 0; }
				if (r == n_diag - 1 && en0 == tlen - 1) ez_score = H0;
				// query bases move one lane up for the next diagonal; lane 0 takes query[r+1]
				{
					const int nr = r + 1;
					if ((nr & 63) == 0) qblock = query_at(nr + lane);
					const int qnew = rl(qblock, nr & 63);
#pragma unroll
					for (int p = NP - 1; p >= 0; --p) {
						const unsigned own = (unsigned)__builtin_amdgcn_readlane(as_i(QB[p]), 63);
						const unsigned prev = p > 0 ? (unsigned)__builtin_amdgcn_readlane(as_i(QB[p > 0 ? p - 1 : 0]), 63) >> 16 : (unsigned)qnew & 0xffffu;
						QB[p] = as_s2(wave_shr1(as_i(QB[p]), (int)(prev | own << 16)));
					}
				}
			}
		} else {
		int qblock = query_at(lane);                 // query[0..63]
		qb[0] = lane == 0 ? __shfl(qblock, 0) : 0;   // diagonal 0: lane 0 needs query[0]

		int H0 = 0, last_H0_t = 0;
		const int n_diag = qlen + tlen - 1;

		for (int r = 0; r < n_diag; ++r) {
			const int st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0, en0 = r < tlen - 1 ? r : tlen - 1;
			const int st = st0 & ~15;
			const int bnd = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;   // first row / first column value
			uint8_t *prow = pmat + (size_t)r * n_col - st;
#pragma unroll
			for (int c = C - 1; c >= 0; --c) {
				if (64 * c > en0 || 64 * c + 63 < st0) continue;                 // wave-uniform
				const int t = lane + 64 * c;
				// t-1 neighbour: one DPP wave shift; lane 0 takes lane 63 of the chunk below (not yet updated on this diagonal,
				// chunks are swept high-to-low) or the boundary values of ksw2_extd2_sse.c:155-158
				const int cx = c > 0 ? __builtin_amdgcn_readlane(x[c > 0 ? c - 1 : 0], 63) : -q - e;
				const int cv = c > 0 ? __builtin_amdgcn_readlane(v[c > 0 ? c - 1 : 0], 63) : bnd;
				const int cx2 = c > 0 ? __builtin_amdgcn_readlane(x2[c > 0 ? c - 1 : 0], 63) : -q2 - e2;
				const int xt1 = wave_shr1(x[c], cx), vt1 = wave_shr1(v[c], cv), x2t1 = wave_shr1(x2[c], cx2);
				int ut = u[c], yt = y[c], y2t = y2[c];
				if (t == r) ut = bnd, yt = -q - e, y2t = -q2 - e2;               // ksw2_extd2_sse.c:160-163
				const bool act = t >= st0 && t <= en0;
				if (act) {
					const int a0 = tb[c], b0 = qb[c];
					int z = a0 == b0 ? sc_mch : sc_mis;
					if (a0 == 4 || b0 == 4) z = sc_N;
					int a = xt1 + vt1, b = yt + ut, a2 = x2t1 + vt1, b2 = y2t + ut, d;
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
					u[c] = z - vt1, v[c] = z - ut;
					int tmp = z - q; a -= tmp, b -= tmp;
					tmp = z - q2; a2 -= tmp, b2 -= tmp;
					if (!right) {
						x[c]  = (a  > 0 ? a  : 0) - qe;  if (a  > 0) d |= 0x08;
						y[c]  = (b  > 0 ? b  : 0) - qe;  if (b  > 0) d |= 0x10;
						x2[c] = (a2 > 0 ? a2 : 0) - qe2; if (a2 > 0) d |= 0x20;
						y2[c] = (b2 > 0 ? b2 : 0) - qe2; if (b2 > 0) d |= 0x40;
					} else {
						x[c]  = (0 > a  ? 0 : a)  - qe;  if (!(0 > a))  d |= 0x08;
						y[c]  = (0 > b  ? 0 : b)  - qe;  if (!(0 > b))  d |= 0x10;
						x2[c] = (0 > a2 ? 0 : a2) - qe2; if (!(0 > a2)) d |= 0x20;
						y2[c] = (0 > b2 ? 0 : b2) - qe2; if (!(0 > b2)) d |= 0x40;
					}
					prow[t] = (uint8_t)d;
				}
			}
			bool stop = false;
			if (!approx_max) {   // ksw2_extd2_sse.c:322-366
				int max_H, max_t, h_last = KSW_NEG_INF;
				if (r > 0) {
					// H[en0] first (from the pre-update neighbour), then H[t] += v[t] for st0 <= t < en0
					int Hen, uen = 0, ven = 0, hen_old = 0, Hen_src = 0;
					{
						const int cc = en0 >> 6, ll = en0 & 63, pc = (en0 - 1) >> 6, pl = (en0 - 1) & 63;
#pragma unroll
						for (int c = 0; c < C; ++c) {
							if (c == cc) { uen = __shfl(u[c], ll); ven = __shfl(v[c], ll); hen_old = __shfl(H[c], ll); }
							if (en0 > 0 && c == pc) Hen_src = __shfl(H[c], pl);
						}
					}
					Hen = en0 > 0 ? Hen_src + uen : hen_old + ven;
					const int en1 = st0 + (en0 - st0) / 4 * 4;
					long long best = ((long long)Hen << 32) | 0xffffffffu;
#pragma unroll
					for (int c = 0; c < C; ++c) {
						const int t = lane + 64 * c;
						if (t >= st0 && t < en0) {
							const int h = H[c] + v[c];
							H[c] = h;
							const unsigned ord = t < en1 ? 1u + ((unsigned)((t - st0) & 3) << 28) + (unsigned)t : 1u + (4u << 28) + (unsigned)t;
							const long long key = ((long long)h << 32) | (0xffffffffu - ord);
							best = key > best ? key : best;
						}
						if (t == en0) H[c] = Hen;
					}
					best = wave_max64f(best);
					max_H = (int)(best >> 32);
					const unsigned ord = 0xffffffffu - (unsigned)(best & 0xffffffffLL);
					max_t = ord == 0 ? en0 : (int)((ord - 1) & 0x0fffffffu);
				} else {
					const int v0 = __shfl(v[0], 0);
					if (lane == 0) H[0] = v0 - qe_h;
					max_H = v0 - qe_h, max_t = 0;
				}
				if (en0 == tlen - 1) {
					int h = 0; const int cc = en0 >> 6, ll = en0 & 63;
#pragma unroll
					for (int c = 0; c < C; ++c) if (c == cc) h = __shfl(H[c], ll);
					if (h > ez_mte) ez_mte = h, ez_mte_q = r - en0;
					h_last = h;
				}
				if (r - st0 == qlen - 1) {
					int h = 0; const int cc = st0 >> 6, ll = st0 & 63;
#pragma unroll
					for (int c = 0; c < C; ++c) if (c == cc) h = __shfl(H[c], ll);
					if (h > ez_mqe) ez_mqe = h, ez_mqe_t = st0;
				}
				if (max_H > ez_max) ez_max = max_H, ez_max_t = max_t, ez_max_q = r - max_t;
				else if (max_t >= ez_max_t && r - max_t >= ez_max_q) {
					const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
					if (zdrop >= 0 && ez_max - max_H > zdrop + l * e2) { ez_zdropped = 1; stop = true; }
				}
				if (!stop && r == n_diag - 1 && en0 == tlen - 1) ez_score = h_last;  // ksw2_extd2_sse.c:364-366: the z-drop break precedes the score assignment
			} else {                                                            // ksw2_extd2_sse.c:367-384
				if (r > 0) {
					int d0 = 0, d1 = 0;
					{
						const int c0 = last_H0_t >> 6, l0 = last_H0_t & 63, c1 = (last_H0_t + 1) >> 6, l1 = (last_H0_t + 1) & 63;
#pragma unroll
						for (int c = 0; c < C; ++c) { if (c == c0) d0 = rl(v[c], l0); if (c == c1) d1 = rl(u[c], l1); }
					}
					if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
						if (d0 > d1) H0 += d0; else H0 += d1, ++last_H0_t;
					} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += d0;
					else ++last_H0_t, H0 += d1;
				} else H0 = __builtin_amdgcn_readlane(v[0], 0) - qe_h, last_H0_t = 0;
				if (r == n_diag - 1 && en0 == tlen - 1) ez_score = H0;
			}
			if (stop) break;
			// query bases move one lane up for the next diagonal; lane 0 takes query[r+1]
			{
				const int nr = r + 1;
				if ((nr & 63) == 0) qblock = query_at(nr + lane);
				const int qnew = rl(qblock, nr & 63);
#pragma unroll
				for (int c = C - 1; c >= 0; --c) {
					const int carry = c > 0 ? __builtin_amdgcn_readlane(qb[c > 0 ? c - 1 : 0], 63) : qnew;
					qb[c] = wave_shr1(qb[c], carry);
				}
			}
		}

		}

		// ---- backtrack (ksw2.h:127-159): lane 0 walks an LDS window refilled by the whole wave ----
		int n_cigar = 0, bi = -1, bj = -1;
		if (!ez_zdropped && !(flag & EZ_EXTZ_ONLY)) bi = tlen - 1, bj = qlen - 1;
		else if (!ez_zdropped && (flag & EZ_EXTZ_ONLY) && ez_mqe + end_bonus > ez_max) ez_reach_end = 1, bi = ez_mqe_t, bj = qlen - 1;
		else if (ez_max_t >= 0 && ez_max_q >= 0) bi = ez_max_t, bj = ez_max_q;
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");                  // direction bytes written by other lanes
		{
			int i = bi, j = bj, state = 0;
			uint32_t last_op = 0xffffffffu;
			uint32_t run_len = 0;                               // the operation being extended lives in registers: one store per operation, not a
			auto cg_push = [&](uint32_t op, uint32_t len) {     // read-modify-write of device memory per path step
				if (op == last_op) { run_len += len; return; }
				if (last_op != 0xffffffffu) { if (lane == 0) cig_tmp[n_cigar] = run_len << 4 | last_op; ++n_cigar; }
				last_op = op; run_len = len;
			};
			auto cg_flush = [&] { if (last_op != 0xffffffffu && n_cigar >= 0) { if (lane == 0) cig_tmp[n_cigar] = run_len << 4 | last_op; ++n_cigar; last_op = 0xffffffffu; } };
			uint32_t *cig = cig_tmp;
			while (i >= 0 && j >= 0) {                                          // wave-uniform loop: every lane tracks (i,j,state)
				if (++guard > 1000000) { n_cigar = -7; break; }                  // safety net: a stuck wave would take the GPU down
				// window: rows r_hi-63 .. r_hi, target columns i-63 .. i (the path moves at most one column per step)
				const int r_hi = i + j, c_lo = i - (BT_COLS - 1);
				{
					// all rows of the window are requested before the first one is stored: 64 loads in flight instead of 64 round trips
					uint8_t wv[BT_ROWS];
#pragma unroll
					for (int row = 0; row < BT_ROWS; ++row) {
						const int r = r_hi - row, col = c_lo + lane;
						uint8_t val = 0;
						if (r >= 0 && col >= 0) {
							const int st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0, en0 = r < tlen - 1 ? r : tlen - 1;
							if (col >= st0 && col <= en0) val = pmat[(size_t)r * n_col + (col - (st0 & ~15))];
						}
						wv[row] = val;
					}
#pragma unroll
					for (int row = 0; row < BT_ROWS; ++row) s_win[row * BT_COLS + lane] = wv[row];
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // single-wave block: LDS is in order, a fence replaces the barrier
				// walk while the path stays inside the window
				while (i >= 0 && j >= 0) {
					const int r = i + j, row = r_hi - r;
					if (row >= BT_ROWS || i < c_lo) break;
					const int st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0, en0 = r < tlen - 1 ? r : tlen - 1;
					const int off = st0 & ~15, off_end = ((en0 + 16) & ~15) - 1;
					int force_state = -1;
					if (i < off) force_state = 2;
					if (i > off_end) force_state = 1;
					const uint32_t tmp = force_state < 0 ? s_win[row * BT_COLS + (i - c_lo)] : 0;
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
			if (bi >= 0 && bj >= 0) {
				if (i >= 0) cg_push(2u, (uint32_t)(i + 1));
				if (j >= 0) cg_push(1u, (uint32_t)(j + 1));
			}
			cg_flush();
		}
		unsigned long long base = 0;
		if (lane == 0 && n_cigar > 0) base = atomicAdd(pool_cursor, (unsigned long long)n_cigar);
		base = ((unsigned long long)(unsigned)__shfl((int)(base >> 32), 0) << 32) | (unsigned)__shfl((int)(base & 0xffffffffULL), 0);
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		const bool rev_cigar = flag & EZ_REV_CIGAR;
		if (base + (unsigned long long)n_cigar <= pool_cap)
			for (int c = lane; c < n_cigar; c += 64) cigar_pool[base + c] = rev_cigar ? cig_tmp[c] : cig_tmp[n_cigar - 1 - c];
		if (lane == 0) {
			DpRes R;
			R.max = ez_max, R.max_q = ez_max_q, R.max_t = ez_max_t, R.mqe = ez_mqe, R.mqe_t = ez_mqe_t, R.mte = ez_mte, R.mte_q = ez_mte_q;
			R.score = ez_score, R.zdropped = ez_zdropped, R.reach_end = ez_reach_end, R.n_cigar = n_cigar, R.pad = 0, R.cigar_off = base;
			res[jid] = R;
		}
	}
}

void launch_extd2_fast(int C, unsigned n_waves, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter, uint8_t *slab, size_t slab_bytes,
                       DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st)
{
	if (C <= 4) hipLaunchKernelGGL((k_extd2_fast<4>), dim3(n_waves), dim3(64), 0, st, jobs, n_jobs, bases, P, counter, slab, slab_bytes, res, pool, cursor, pool_cap);
	else hipLaunchKernelGGL((k_extd2_fast<8>), dim3(n_waves), dim3(64), 0, st, jobs, n_jobs, bases, P, counter, slab, slab_bytes, res, pool, cursor, pool_cap);
}

} // namespace pga
