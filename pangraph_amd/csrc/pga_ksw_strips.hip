// pga_ksw_strips.hip -- kernel #5f: ONE large unbanded first-pass gap fill (KSW_EZ_APPROX_MAX: global alignment of two windows of
// up to max_gap bases across a rearrangement, ksw2_extd2_sse.c:34-401 with w >= both lengths) spread over SEVERAL workgroups.
//
// A 10 kb x 10 kb matrix is 10^8 cells; one workgroup -- one CU -- sweeps it in 80-130 ms (pga_ksw_wide.hip), and in the dependent
// rounds at the end of a wave such a problem runs alone on a 256-CU device.  Here the target is cut into STRIPS of 512 columns, one
// workgroup each.  A cell reads its own column's state of the previous diagonal and the left neighbour's (x, v, x2) of the previous
// diagonal, so the only traffic between strips is three bytes per diagonal: the state of a strip's last column, handed to the strip
// on its right through a small array in device memory (one word per diagonal carrying the three values AND its own valid bit 31, so relaxed agent-scope accesses suffice).
// Strip k starts 512 diagonals after strip k-1 (its first column joins the matrix then), so the values it needs were written long
// before it asks: the pipeline runs without waiting once it is filled.  Workgroups are ordered (problem, strip) in the grid: a
// workgroup only ever waits for one with a lower block index, which the dispatcher started earlier.
//
// No running corner score is kept (it would have to travel across strips): the last workgroup to finish walks the direction matrix
// back (same walk as the other kernels) and evaluates the score OF THE PATH -- the optimum of a global alignment is the score of its
// backtracked path: match / mismatch / ambiguous columns from the bases, every gap at min(q + l*e, q2 + l*e2), which is what the two
// affine gap states of the recurrence charge an l-base gap (ksw2_extd2_sse.c: E = max(H - q, E) - e for both pairs, H = max of all).
// Arithmetic and direction bytes are the packed two-column cell pass of pga_ksw_wide.hip (no int8 wrap inside an unbinding band).
#include "pga_common.h"
#include "pga_dp.h"
#include "pga_wave.h"
#include "pga_pk16.h"
#include <cstring>

namespace pga {

#define KSW_NEG_INF (-0x40000000)
#define ST_S 512            // columns per strip (one pair of columns per compute thread and diagonal)
#define ST_NT 256           // compute threads per workgroup: one pair of columns each per diagonal (256-column strips measured the same: the cost of a diagonal is its barrier and LDS round trips)
#define ST_NTL 320          // launched threads: a fifth wave does nothing but fetch the left neighbour's boundary words
#define ST_QMAX 12288       // longest query window
#define ST_BT 64

__device__ __forceinline__ void st_range(int r, int qlen, int tlen, int &st0, int &en0)
{
	st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0;
	en0 = r < tlen - 1 ? r : tlen - 1;
}

__global__ __launch_bounds__(ST_NTL)
void k_approx_strips(const DpJob *__restrict__ jobs, const uint32_t *__restrict__ blk_job, const uint32_t *__restrict__ blk_strip, PkBases bases, DpParams P,
                     uint8_t *__restrict__ slab_all, const uint64_t *__restrict__ slab_off, uint32_t *__restrict__ bnd_all, const uint64_t *__restrict__ bnd_off,
                     uint32_t *__restrict__ done_ctr, DpRes *__restrict__ res, uint32_t *__restrict__ cigar_pool, unsigned long long *__restrict__ pool_cursor, unsigned long long pool_cap)
{
	__shared__ int8_t s_u[ST_S + 16], s_y[ST_S + 16], s_y2[ST_S + 16];
	__shared__ int8_t s_x[2][ST_S + 16], s_v[2][ST_S + 16], s_x2[2][ST_S + 16];
	__shared__ uint8_t s_t[ST_S + 16];
	__shared__ uint8_t s_q[ST_QMAX + 64];                 // query REVERSED with 32 zero bytes on either side: column t of diagonal r reads s_q[t + 32 + qlen-1-r]
	__shared__ uint8_t s_win[ST_BT * ST_BT];
	__shared__ uint32_t s_last;
	__shared__ uint32_t s_bnd[16];                        // boundary words of 2 x 8 diagonals: slot (d - first) & 15 holds bnd_in[d]
	// exact-maximum problems (second passes: flag == 0): H of the strip's columns; H of the diagonal's LAST column (it runs along the first
	// query row from strip to strip, then stays on the last target column); the waves' best keys of a diagonal, by diagonal parity
	__shared__ int32_t s_H[ST_S + 16];
	__shared__ int32_t s_hen;
	__shared__ uint32_t s_part[4][ST_NT / 64];            // (four diagonals deep: a diagonal's key is published two diagonals later, see s_late)
	__shared__ uint32_t s_late[4];                        // odd target length: the key of H[tlen-1], which is only known one diagonal late
	__shared__ int32_t s_hm2[2];                          // ... and H[tlen-2] as each diagonal left it, by parity
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t jl = blk_job[blockIdx.x], k = blk_strip[blockIdx.x];
	const DpJob J = jobs[jl];
	const uint64_t t_base = J.t_off, q_base = J.q_off;          // base positions in the packed store
	const int qlen = J.qlen, tlen = J.tlen;
	int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
	if (q2 + e2 < q + e) { int t = q; q = q2, q2 = t, t = e, e = e2, e2 = t; }
	const int qe = q + e, qe2 = q2 + e2;
	const int sc_mch = P.sc_mch, sc_mis = P.sc_mis, sc_N = P.sc_ambi == 0 ? -e2 : P.sc_ambi;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const int n_strips = (tlen + ST_S - 1) / ST_S;
	const bool exact = !(J.flag & 0x08);
	const int qe_h = P.q + P.e;
	const size_t Ld = (size_t)(qlen + tlen);
	const int c0 = (int)k * ST_S, c1 = c0 + ST_S < tlen ? c0 + ST_S : tlen;
	int n_col = qlen < tlen ? qlen : tlen;
	n_col = ((n_col + 15) / 16 + 1) * 16;                 // (w + 1 > both lengths)
	uint8_t *pmat = slab_all + slab_off[jl];
	uint32_t *cig_tmp = (uint32_t*)(pmat + (((size_t)(qlen + tlen - 1) * n_col + 15) & ~(size_t)15));
	uint32_t *bnd_in = k > 0 ? bnd_all + bnd_off[jl] + (size_t)(k - 1) * (size_t)(qlen + tlen) : nullptr;      // written by strip k-1, indexed by diagonal
	uint32_t *bnd_out = (int)k + 1 < n_strips ? bnd_all + bnd_off[jl] + (size_t)k * (size_t)(qlen + tlen) : nullptr;
	// exact mode, behind the boundary words: one key per strip and diagonal; H of the diagonal's last column and of its first column (for
	// the end-of-sequence scores) per diagonal; one word per strip: H of its last column at the moment the diagonal's end stood there
	uint32_t *keys_all = bnd_all + bnd_off[jl] + (size_t)(n_strips - 1) * Ld;
	uint32_t *keys = keys_all + (size_t)k * Ld;
	int32_t *hen_arr = (int32_t*)(keys_all + (size_t)n_strips * Ld), *hst_arr = hen_arr + Ld;
	uint32_t *hb_all = (uint32_t*)(hst_arr + Ld);
	auto target_at = [&](int i) -> int { return i < tlen ? (int)bases.at(t_base + (uint64_t)(J.seq_rev ? tlen - 1 - i : i)) : 0; };
	auto query_at = [&](int j) -> int {
		if (j < 0 || j >= qlen) return 0;
		const int pj = J.qs + (J.seq_rev ? qlen - 1 - j : j);
		if (!J.q_rev) return bases.at(q_base + (uint64_t)(pj));
		const int c = bases.at(q_base + (uint64_t)(J.qlen_full - 1 - pj));
		return c < 4 ? 3 - c : 4;
	};
	for (int t = tid; t < ST_S + 16; t += ST_NTL) {
		s_u[t] = s_y[t] = (int8_t)(-q - e); s_y2[t] = (int8_t)(-q2 - e2);
		s_x[0][t] = s_x[1][t] = s_v[0][t] = s_v[1][t] = (int8_t)(-q - e);
		s_x2[0][t] = s_x2[1][t] = (int8_t)(-q2 - e2);
		s_t[t] = (uint8_t)target_at(c0 + t);
		s_H[t] = KSW_NEG_INF;
	}
	if (tid == 0) s_hen = KSW_NEG_INF;
	if (tid < 4) s_late[tid] = 0;
	for (int p = tid; p < qlen + 64; p += ST_NTL) { const int j = qlen - 1 - (p - 32); s_q[p] = (j >= 0 && j < qlen) ? (uint8_t)query_at(j) : (uint8_t)0; }
	__syncthreads();

	const s2_t ZERO = splat2(0), ONE = splat2(1), MCH = splat2(sc_mch), Q1 = splat2(q), Q2 = splat2(q2), QE = splat2(qe), QE2 = splat2(qe2);
	const s2_t EIGHT = splat2(8), C16 = splat2(16), C32 = splat2(32), C64 = splat2(64);
	const int r_first = c0, r_last = c1 - 1 + qlen - 1;    // the diagonals on which this strip holds valid cells
	// The boundary words of the strip on the left are fetched by a wave of their own, eight diagonals per load, one block ahead, into
	// an LDS ring: the compute waves never wait for a load from device memory -- a wave that did would also wait for all of its own
	// direction-byte stores on every diagonal (loads and stores retire through one in-order counter).
	const bool comm = wave == ST_NT / 64;
	const int d_first = r_first - 1;                       // first boundary diagonal this strip reads
	const int d_last = c0 + qlen - 2;                      // last one the left strip publishes
	auto fetch_block = [&](int blk) {                      // diagonals d_first + 8*blk .. +7 -> ring slots (8*blk .. +7) & 15
		if (!bnd_in || lane >= 8) return;
		const int d = d_first + 8 * blk + lane;
		uint32_t w = 0;
		if (d >= 0 && d <= d_last) { do w = __hip_atomic_load(&bnd_in[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); while (!(w >> 31)); }
		s_bnd[(8 * blk + lane) & 15] = w;
	};
	if (comm) { fetch_block(0); fetch_block(1); }
	__syncthreads();
	for (int r = r_first; r <= r_last; ++r) {
		int st0, en0;
		st_range(r, qlen, tlen, st0, en0);
		const int lo = st0 > c0 ? st0 : c0, hi = en0 < c1 - 1 ? en0 : c1 - 1;
		const int8_t *xr = s_x[r & 1], *vr = s_v[r & 1], *x2r = s_x2[r & 1];
		int8_t *xw = s_x[(r + 1) & 1], *vw = s_v[(r + 1) & 1], *x2w = s_x2[(r + 1) & 1];
		const int it = r - r_first;
		if (comm) {
			// at the start of block b (it = 8b) the ring half of block b+1 is free again (block b-1 was read in the trips before)
			if ((it & 7) == 0 && it > 0) fetch_block(it / 8 + 1);
			asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
			continue;
		}
		if (exact && tid == 0 && r >= r_first + 2) {           // the key of the diagonal before the previous one: every contribution to it is in LDS by now
			uint32_t kk = s_late[(r - 2) & 3];
			s_late[(r - 2) & 3] = 0;
#pragma unroll
			for (int wv = 0; wv < ST_NT / 64; ++wv) { const uint32_t o = s_part[(r - 2) & 3][wv]; kk = o > kk ? o : kk; }
			keys[r - 2] = kk;
		}
		uint32_t kbest = 0;
		// left boundary of the strip's first column
		int x1, v1, x21;
		if (c0 == 0) {
			x1 = -q - e, x21 = -q2 - e2;
			v1 = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
		} else {
			const uint32_t nb = s_bnd[it & 15];                // = bnd_in[r - 1]
			x1 = (int)(int8_t)(nb & 0xff), v1 = (int)(int8_t)(nb >> 8 & 0xff), x21 = (int)(int8_t)(nb >> 16 & 0xff);
		}
		const int u_join = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;      // first-row u of column r
		const int off_r = 32 + qlen - 1 - r;
		const int st = st0 / 16 * 16;
		uint8_t *prow = pmat + (size_t)r * n_col - st;
		for (int t = (lo & ~1) + 2 * tid; t <= hi; t += 2 * ST_NT) {
			const int l = t - c0;                              // strip-local column (even)
			const int xl = l == 0 ? x1 : (int)xr[l - 1], vl = l == 0 ? v1 : (int)vr[l - 1], x2l = l == 0 ? x21 : (int)x2r[l - 1];
			const s2_t xt1 = pack2(xl, (int)xr[l]), vt1 = pack2(vl, (int)vr[l]), x2t1 = pack2(x2l, (int)x2r[l]);
			s2_t ut = unpack_i8x2(*reinterpret_cast<const uint16_t*>(s_u + l)), yt = unpack_i8x2(*reinterpret_cast<const uint16_t*>(s_y + l));
			s2_t y2t = unpack_i8x2(*reinterpret_cast<const uint16_t*>(s_y2 + l));
			if (t == r) { ut.x = (short)u_join; yt.x = (short)(-q - e); y2t.x = (short)(-q2 - e2); }
			else if (t + 1 == r) { ut.y = (short)u_join; yt.y = (short)(-q - e); y2t.y = (short)(-q2 - e2); }
			s2_t z;
			{
				const int a0 = s_t[l], a1 = s_t[l + 1], b0 = s_q[t + off_r], b1 = s_q[t + 1 + off_r];
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
			const uint16_t xn8 = pack_i8x2(xn), vn8 = pack_i8x2(vn), x2n8 = pack_i8x2(x2n);
			*reinterpret_cast<uint16_t*>(s_u + l) = pack_i8x2(un); *reinterpret_cast<uint16_t*>(vw + l) = vn8;
			*reinterpret_cast<uint16_t*>(xw + l) = xn8; *reinterpret_cast<uint16_t*>(s_y + l) = pack_i8x2(yn);
			*reinterpret_cast<uint16_t*>(x2w + l) = x2n8; *reinterpret_cast<uint16_t*>(s_y2 + l) = pack_i8x2(y2n);
			// direction bytes: only columns of this strip (a pair may reach one column beyond a ragged last strip: padding of the row)
			const uint16_t d8 = pack_i8x2(d);
			if (t + 1 < c0 + ST_S) *reinterpret_cast<uint16_t*>(prow + t) = d8; else prow[t] = (uint8_t)d8;
			// the strip's last column feeds the strip on the right
			if (bnd_out && t + 1 == c1 - 1 && c1 - 1 >= st0 && c1 - 1 <= en0)
				__hip_atomic_store(&bnd_out[r], 0x80000000u | (uint32_t)(xn8 >> 8) | (uint32_t)(vn8 >> 8) << 8 | (uint32_t)(x2n8 >> 8) << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (exact) {
				// H[t] += v[t] over [st0, en0); H[en0] = H[en0-1](as the previous diagonal left it) + u[en0] (ksw2_extd2_sse.c:325-340).  The
				// target length is even (eligibility), so tlen-2 and tlen-1 are one pair: the owner of en0 either holds en0-1 as the low column
				// of its pair, or en0 moved on by one column and H[en0-1] is what the previous diagonal's owner left in s_hen (in the strip on
				// the left when en0 is this strip's first column).  The maximum with the reference's tie order (H[en0] first, then four lanes
				// by (t - st0) & 3 over [st0, en1), then the tail) is one key per column: (clamp16(H) + 32768) << 16 | field << 12 | 511 - column.
				const int en1 = st0 + (en0 - st0) / 4 * 4;
				const uint16_t un8 = pack_i8x2(un);
				const int old0 = s_H[l], old1 = s_H[l + 1];
#pragma unroll
				for (int hh = 0; hh < 2; ++hh) {
					const int tc = t + hh, lc = l + hh;
					if (tc < lo || tc > hi) continue;
					const int vv = (int)(int8_t)(hh ? vn8 >> 8 : vn8 & 0xff), uu = (int)(int8_t)(hh ? un8 >> 8 : un8 & 0xff);
					int h; uint32_t field;
					if (tc == en0) {
						if (r == 0) h = vv - qe_h;
						else if (hh == 0 && en0 < r) {
							// odd target length, the diagonal's end rests on column tlen-1 = the LOW column of its pair: H[tlen-2] belongs to
							// another thread, which is changing it right now.  H[en0] of the PREVIOUS diagonal follows from what that diagonal
							// left behind: H[tlen-2](new) - v[tlen-2](new) is H[tlen-2] before its update, plus u[tlen-1](new).  Nobody reads
							// H[tlen-1] (there is no column to its right): its key and its value are simply recorded one diagonal late.
							if (r - 1 >= tlen) {
								const int hl = s_hm2[(r - 1) & 1] - vl + (int)ut.x;
								hen_arr[r - 1] = hl;
								const int hcl = hl < -32768 ? -32768 : hl > 32767 ? 32767 : hl;
								s_late[(r - 1) & 3] = ((uint32_t)(hcl + 32768) << 16) | 8u << 12 | (uint32_t)(511 - lc);
							}
							continue;
						} else {
							int hp;
							if (hh == 1) hp = old0;
							else if (lc > 0 || k == 0) hp = s_hen;
							else { uint32_t wd; do wd = __hip_atomic_load(&hb_all[k - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); while (!(wd >> 31)); hp = (int)(wd & 0x7fffffffu) - 0x20000000; }
							h = hp + uu;
						}
						field = 8u;
						s_hen = h;
						hen_arr[r] = h;
						if (bnd_out && tc == c1 - 1) __hip_atomic_store(&hb_all[k], 0x80000000u | ((uint32_t)(h + 0x20000000) & 0x7fffffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					} else {
						h = (hh ? old1 : old0) + vv;
						field = 7u - (tc < en1 ? (uint32_t)((tc - st0) & 3) : 4u);
						if ((tlen & 1) && tc == tlen - 2) s_hm2[r & 1] = h;
					}
					s_H[lc] = h;
					if (tc == st0 && r - st0 == qlen - 1) hst_arr[r] = h;
					const int hc = h < -32768 ? -32768 : h > 32767 ? 32767 : h;
					const uint32_t key = ((uint32_t)(hc + 32768) << 16) | field << 12 | (uint32_t)(511 - lc);
					kbest = key > kbest ? key : kbest;
				}
			}
		}
		if (exact) { kbest = wave_max_u32(kbest); if (lane == 0) s_part[r & 3][wave] = kbest; }
		// the barrier orders the LDS rows only: the direction bytes and the boundary word are fire-and-forget (a __syncthreads() would
		// wait for every outstanding store to device memory on every diagonal); they are fenced once, before the completion counter
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
	if (exact && tid == 0) {                                  // the keys of the strip's last two diagonals (and, odd target length, the late H[tlen-1])
		if ((tlen & 1) && c1 == tlen && r_last >= tlen) {
			const int lc = tlen - 1 - c0;
			const int hl = s_H[lc - 1] + (int)s_u[lc];          // (column tlen-2 is not part of the last diagonal: its H is the value before it)
			hen_arr[r_last] = hl; hst_arr[r_last] = hl;       // (the last diagonal is the single cell (tlen-1, qlen-1))
			const int hcl = hl < -32768 ? -32768 : hl > 32767 ? 32767 : hl;
			s_late[r_last & 3] = ((uint32_t)(hcl + 32768) << 16) | 8u << 12 | (uint32_t)(511 - lc);
		}
		for (int rr = r_last - 1 > r_first ? r_last - 1 : r_first; rr <= r_last; ++rr) {
			uint32_t kk = s_late[rr & 3];
#pragma unroll
			for (int wv = 0; wv < ST_NT / 64; ++wv) { const uint32_t o = s_part[rr & 3][wv]; kk = o > kk ? o : kk; }
			keys[rr] = kk;
		}
	}
	// ---- the last workgroup of the problem to finish walks the path back and scores it ----
	__threadfence();
	__syncthreads();
	if (tid == 0) s_last = atomicAdd(&done_ctr[jl], 1u);   // (every wave, the fetching one included, is past its last barrier)
	__syncthreads();
	if (s_last + 1 != (uint32_t)n_strips) return;
	__threadfence();
	if (wave != 0) return;
	int n_cigar = 0;
	int bi = tlen - 1, bj = qlen - 1;
	int ez_max = 0, ez_max_t = -1, ez_max_q = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1, ez_mte = KSW_NEG_INF, ez_mte_q = -1, zdropped = 0;
	if (exact) {
		// The maxima of all diagonals are on record: what the reference decides diagonal by diagonal (ksw_apply_zdrop, the end-of-sequence
		// scores) is decided here in one sweep, 64 diagonals per trip: a lane combines the strips' keys of its diagonal (equal H and field:
		// the lower strip, then the lower column), then the trip's diagonals are taken in order.
		const int n_diag = qlen + tlen - 1, zdrop = J.zdrop;
		int sat = 0;
		for (int r0 = 0; r0 < n_diag && !zdropped; r0 += 64) {
			const int r = r0 + lane;
			unsigned long long best = 0; int hen = KSW_NEG_INF, hst = KSW_NEG_INF;
			if (r < n_diag) {
				int st0, en0; st_range(r, qlen, tlen, st0, en0);
				for (int kk = 0; kk < n_strips; ++kk) {
					const int c0k = kk * ST_S, c1k = c0k + ST_S < tlen ? c0k + ST_S : tlen;
					if (c0k <= en0 && c1k - 1 >= st0) {
						const uint32_t key = keys_all[(size_t)kk * Ld + (size_t)r];
						const unsigned long long comb = (unsigned long long)(key >> 12) << 14 | (unsigned long long)(31 - kk) << 9 | (unsigned long long)(key & 511u);
						best = comb > best ? comb : best;
					}
				}
				hen = hen_arr[r];
				if (r - st0 == qlen - 1) hst = hst_arr[r];
			}
			const uint32_t h16 = (uint32_t)(best >> 18) & 0xffffu;
			const int mH_l = (int)h16 - 32768, mt_l = (31 - (int)((best >> 9) & 31)) * ST_S + (511 - (int)(best & 511));
			const int sat_l = r < n_diag && (h16 == 0 || h16 == 65535u) ? 1 : 0;
			const int lim = n_diag - r0 < 64 ? n_diag - r0 : 64;
			for (int ii = 0; ii < lim; ++ii) {
				const int rr = r0 + ii;
				const int mH = __builtin_amdgcn_readlane(mH_l, ii), mt = __builtin_amdgcn_readlane(mt_l, ii);
				const int he = __builtin_amdgcn_readlane(hen, ii), hs = __builtin_amdgcn_readlane(hst, ii);
				sat |= __builtin_amdgcn_readlane(sat_l, ii);
				int st0, en0; st_range(rr, qlen, tlen, st0, en0);
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
	uint32_t run_len = 0;                               // the operation being extended lives in registers: one store per operation, not a
	auto cg_push = [&](uint32_t op, uint32_t len) {     // read-modify-write of device memory per path step
		if (op == last_op) { run_len += len; return; }
		if (last_op != 0xffffffffu) { if (lane == 0) cig_tmp[n_cigar] = run_len << 4 | last_op; ++n_cigar; }
		last_op = op; run_len = len;
	};
	auto cg_flush = [&] { if (last_op != 0xffffffffu && n_cigar >= 0) { if (lane == 0) cig_tmp[n_cigar] = run_len << 4 | last_op; ++n_cigar; last_op = 0xffffffffu; } };
	while (i >= 0 && j >= 0) {
		if (++guard > 4000000) { n_cigar = -7; break; }
		const int r_hi = i + j, c_lo = i - (ST_BT - 1);
		{
			// all 64 rows of the window are requested before the first one is stored (a load-store pair per row would pay the memory
			// latency 64 times per window, and a 10 kb x 10 kb path crosses ~300 windows)
			uint8_t wv[ST_BT];
#pragma unroll
			for (int row = 0; row < ST_BT; ++row) {
				const int r = r_hi - row, col = c_lo + lane;
				uint8_t val = 0;
				if (r >= 0 && col >= 0) {
					int st0, en0; st_range(r, qlen, tlen, st0, en0);
					const int off = st0 / 16 * 16;
					if (st0 <= en0 && col >= st0 && col <= en0) val = pmat[(size_t)r * n_col + (col - off)];
				}
				wv[row] = val;
			}
#pragma unroll
			for (int row = 0; row < ST_BT; ++row) s_win[row * ST_BT + lane] = wv[row];
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		while (i >= 0 && j >= 0) {
			const int r = i + j, row = r_hi - r;
			if (row >= ST_BT || i < c_lo) break;
			int st0, en0; st_range(r, qlen, tlen, st0, en0);
			const int off = st0 / 16 * 16, off_end = (en0 + 16) / 16 * 16 - 1;
			int force_state = -1;
			if (i < off) force_state = 2;
			if (i > off_end) force_state = 1;
			const uint32_t tmp = force_state < 0 ? s_win[row * ST_BT + (i - c_lo)] : 0;
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

bool strips_eligible(const DpJob &j, const DpParams &P)
{
	static const int min_t = getenv("PGA_STRIPS_MIN") ? atoi(getenv("PGA_STRIPS_MIN")) : 2048;      // below four strips the single workgroup wins
	static const int min_x = getenv("PGA_STRIPS_EXACT_MIN") ? atoi(getenv("PGA_STRIPS_EXACT_MIN")) : 4096;   // exact second passes (every diagonal is computed, no early exit on z-drop): from 6 kb
	if (min_t <= 0) return false;
	if (!(j.w >= j.qlen && j.w >= j.tlen && j.qlen >= 256 && j.qlen + 64 <= ST_QMAX && P.sc_mch >= 0 && P.sc_mch < 127)) return false;
	if (j.flag == 0x08) return j.tlen >= min_t;
	return j.flag == 0 && min_x > 0 && j.tlen >= min_x && (j.tlen + ST_S - 1) / ST_S <= 32 && (j.tlen % ST_S) != 1;
}
size_t strips_slab_bytes(const DpJob &j)
{
	size_t n_col = (size_t)(j.qlen < j.tlen ? j.qlen : j.tlen);
	n_col = ((n_col + 15) / 16 + 1) * 16;
	return (((((size_t)(j.qlen + j.tlen - 1) * n_col + 15) & ~(size_t)15) + 4 * ((size_t)j.qlen + j.tlen + 8)) + 255) & ~(size_t)255;
}
int strips_count(const DpJob &j) { return (j.tlen + ST_S - 1) / ST_S; }
size_t strips_bnd_words(const DpJob &j)
{
	const size_t L = (size_t)(j.qlen + j.tlen), ns = (size_t)strips_count(j);
	return (ns > 1 ? ns - 1 : 0) * L + ((j.flag & 0x08) ? 0 : ns * L + 2 * L + ns + 8);      // exact mode: keys, H of the first / last column per diagonal, hand-off words
}

void launch_approx_strips(unsigned n_blocks, const DpJob *jobs, const uint32_t *blk_job, const uint32_t *blk_strip, PkBases bases, const DpParams &P, uint8_t *slab, const uint64_t *slab_off,
                          uint32_t *bnd, const uint64_t *bnd_off, uint32_t *done_ctr, DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st)
{
	hipLaunchKernelGGL(k_approx_strips, dim3(n_blocks), dim3(ST_NTL), 0, st, jobs, blk_job, blk_strip, bases, P, slab, slab_off, bnd, bnd_off, done_ctr, res, pool, cursor, pool_cap);
}

} // namespace pga
