// pga_ksw.hip -- kernel #5: batched dual-affine-gap extension / global alignment with backtrack.
//
// Replaces ksw_extd2_sse() (reference: packages/minimap2-sys/minimap2/ksw2_extd2_sse.c:34-401) together with
// ksw_backtrack / ksw_apply_zdrop (ksw2.h:127-184): the 57 % hotspot of the path.
//
// One wavefront per DP problem, anti-diagonal sweep, lane <-> target coordinate t:
//   * the six difference rows u,v,x,y,x2,y2 and the score profile s live in LDS (or in a per-wave HBM slab for
//     problems wider than LDS_T); the t-1 neighbour that SSE gets with _mm_slli_si128 comes from a wave shuffle,
//     with a scalar carry across 64-lane chunks;
//   * arithmetic is the reference's int8 difference recurrence, evaluated in 32-bit lanes and re-wrapped to
//     8 bits (v_bfe_i32) wherever the SSE code produces a byte, INCLUDING the lanes the SSE code computes
//     outside the band (16-lane rounding of [st0,en0], stale score profile, ksw2_extd2_sse.c:140-190): banded
//     extensions depend on them;
//   * the 1-byte/cell direction matrix streams to a per-wave HBM slab with coalesced 64-byte stores and is
//     walked back by lane 0; CIGARs are appended to a shared pool with one atomic per problem;
//   * exact-max diagonals reduce (H,t) with the reference's 4-lane-strided tie order encoded in the key.
// Persistent launch: a fixed grid of waves pulls problems from an atomic queue (problem sizes are ragged).
#include "pga_common.h"
#include "pga_dp.h"
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <array>
#include <cstdio>
#include <cstring>

namespace pga {

#define KSW_NEG_INF (-0x40000000)
#define EZ_RIGHT      0x02
#define EZ_APPROX_MAX 0x08
#define EZ_APPROX_DROP 0x10
#define EZ_EXTZ_ONLY  0x40
#define EZ_REV_CIGAR  0x80

__device__ __forceinline__ int sx8(int v) { return __builtin_amdgcn_sbfe(v, 0, 8); }

struct SeqView {
	PkBases nt; uint64_t t_base, q_base;   // the packed store; target window start / query sequence start in it
	int32_t qlen_full, qs, qlen, tlen;
	bool q_rev, seq_rev;
	__device__ __forceinline__ int target(int i) const { // 0 beyond the window (the reference's zero padding)
		if (i >= tlen) return 0;
		return nt.at(t_base + (uint64_t)(seq_rev ? tlen - 1 - i : i));
	}
	__device__ __forceinline__ int query(int j) const {
		if (j < 0 || j >= qlen) return 0;
		int pj = qs + (seq_rev ? qlen - 1 - j : j);
		if (!q_rev) return nt.at(q_base + (uint64_t)(pj));
		int c = nt.at(q_base + (uint64_t)(qlen_full - 1 - pj));
		return c < 4 ? 3 - c : 4;
	}
};

__device__ __forceinline__ void diag_range(int r, int qlen, int tlen, int w, int &st0, int &en0)
{
	int st = 0, en = tlen - 1;
	if (st < r - qlen + 1) st = r - qlen + 1;
	if (en > r) en = r;
	if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
	if (en > (r + w) >> 1) en = (r + w) >> 1;
	st0 = st, en0 = en;
}

__device__ __forceinline__ long long wave_max64(long long v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		int lo = __shfl_xor((int)(v & 0xffffffffLL), d), hi = __shfl_xor((int)(v >> 32), d);
		long long o = ((long long)hi << 32) | (unsigned int)lo;
		v = o > v ? o : v;
	}
	return v;
}

#define LDS_T 2048   // problems with tlen16 <= LDS_T keep their rows in LDS

__global__ __launch_bounds__(64)
void k_extd2(const DpJob *__restrict__ jobs, uint32_t n_jobs, PkBases bases, DpParams P,
             uint32_t *__restrict__ job_counter, uint8_t *__restrict__ slab_all, size_t slab_bytes,
             DpRes *__restrict__ res, uint32_t *__restrict__ cigar_pool, unsigned long long *__restrict__ pool_cursor, unsigned long long pool_cap)
{
	__shared__ int8_t s_rows[7 * LDS_T];
	__shared__ int32_t s_H[LDS_T];
	__shared__ uint32_t s_job;
	const int lane = threadIdx.x;
	uint8_t *slab = slab_all + (size_t)blockIdx.x * slab_bytes;

	for (;;) {
		if (lane == 0) s_job = atomicAdd(job_counter, 1u);
		__syncthreads();
		const uint32_t jid = s_job;
		__syncthreads();
		if (jid >= n_jobs) break;
		const DpJob J = jobs[jid];
		SeqView V;
		V.nt = bases, V.t_base = J.t_off, V.q_base = J.q_off, V.qlen_full = J.qlen_full, V.qs = J.qs, V.qlen = J.qlen, V.tlen = J.tlen;
		V.q_rev = J.q_rev, V.seq_rev = J.seq_rev;
		const int qlen = J.qlen, tlen = J.tlen, flag = J.flag, zdrop = J.zdrop, end_bonus = J.end_bonus;
		int w = J.w;
		int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
		const int qe_h = q + e;                                   // ksw2_extd2_sse.c:73 (taken before the swap)
		if (q2 + e2 < q + e) { int t = q; q = q2, q2 = t, t = e, e = e2, e2 = t; }
		const int qe = q + e, qe2 = q2 + e2;
		const int sc_mch = P.sc_mch, sc_mis = P.sc_mis, sc_N = P.sc_ambi == 0 ? -e2 : P.sc_ambi;
		const bool approx_max = flag & EZ_APPROX_MAX, right = flag & EZ_RIGHT;
		if (w < 0) w = tlen > qlen ? tlen : qlen;
		const int tlen16 = (tlen + 15) / 16 * 16;
		int n_col = qlen < tlen ? qlen : tlen;
		n_col = (((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1) * 16;
		int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
		if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
		const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;

		// row storage: LDS when it fits, else the head of this wave's HBM slab; the direction matrix follows
		int8_t *rows; int32_t *H; uint8_t *pmat;
		if (tlen16 <= LDS_T) { rows = s_rows; H = s_H; pmat = slab; }
		else { rows = (int8_t*)slab; H = (int32_t*)(slab + (((size_t)7 * tlen16 + 15) & ~(size_t)15)); pmat = (uint8_t*)(H + tlen16); }
		int8_t *u = rows, *v = u + tlen16, *x = v + tlen16, *y = x + tlen16, *x2 = y + tlen16, *y2 = x2 + tlen16, *s = y2 + tlen16;
		uint32_t *cig_tmp = (uint32_t*)(pmat + (((size_t)(qlen + tlen - 1) * n_col + 15) & ~(size_t)15));
		for (int t = lane; t < tlen16; t += 64) {
			u[t] = v[t] = x[t] = y[t] = (int8_t)(-q - e);
			x2[t] = y2[t] = (int8_t)(-q2 - e2);
			s[t] = 0;
			if (!approx_max) H[t] = KSW_NEG_INF;
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		__syncthreads();

		// ez
		int ez_max = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1, ez_mte = KSW_NEG_INF, ez_mte_q = -1;
		int ez_score = KSW_NEG_INF, ez_zdropped = 0, ez_reach_end = 0;
		int H0 = 0, last_H0_t = 0, last_st = -1, last_en = -1;

		for (int r = 0; r < qlen + tlen - 1; ++r) {
			int st0, en0;
			diag_range(r, qlen, tlen, w, st0, en0);
			if (st0 > en0) { ez_zdropped = 1; break; }
			const int st = st0 / 16 * 16, en = (en0 + 16) / 16 * 16 - 1;
			int x1, x21, v1;
			if (st > 0) {
				if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1];
				else x1 = sx8(-q - e), x21 = sx8(-q2 - e2), v1 = sx8(-q - e);
			} else {
				x1 = sx8(-q - e), x21 = sx8(-q2 - e2);
				v1 = r == 0 ? sx8(-q - e) : r < long_thres ? sx8(-e) : r == long_thres ? sx8(long_diff) : sx8(-e2);
			}
			if (en >= r && lane == 0) {
				y[r] = (int8_t)(-q - e), y2[r] = (int8_t)(-q2 - e2);
				u[r] = (int8_t)(r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2);
			}
			// score profile, refreshed in 16-lane groups starting at st0 (ksw2_extd2_sse.c:165-181)
			{
				const int span = ((en0 - st0) / 16 + 1) * 16;
				for (int o = lane; o < span; o += 64) {
					const int t = st0 + o;
					if (t < tlen16) {
						const int a = V.target(t), b = V.query(r - t);
						int sc = a == b ? sc_mch : sc_mis;
						if (a == 4 || b == 4) sc = sc_N;
						s[t] = (int8_t)sc;
					}
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			__syncthreads();
			uint8_t *prow = pmat + (size_t)r * n_col;
			for (int c0 = st; c0 <= en; c0 += 64) {
				const int t = c0 + lane;
				const bool act = t <= en;
				int xo = 0, vo = 0, x2o = 0, ut = 0, yo = 0, y2o = 0, z = 0;
				if (act) { xo = x[t], vo = v[t], x2o = x2[t], ut = u[t], yo = y[t], y2o = y2[t], z = s[t]; }
				int xt1 = __shfl_up(xo, 1), vt1 = __shfl_up(vo, 1), x2t1 = __shfl_up(x2o, 1);
				if (lane == 0) xt1 = x1, vt1 = v1, x2t1 = x21;
				x1 = __shfl(xo, 63), v1 = __shfl(vo, 63), x21 = __shfl(x2o, 63);
				if (act) {
					int a = sx8(xt1 + vt1), b = sx8(yo + ut), a2 = sx8(x2t1 + vt1), b2 = sx8(y2o + ut), d;
					if (!right) { // ties keep the earlier state (ksw2_extd2_sse.c:238-258)
						d = 0;
						if (a > z) d = 1, z = a;
						if (b > z) d = 2, z = b;
						if (a2 > z) d = 3, z = a2;
						if (b2 > z) d = 4, z = b2;
					} else {      // ties move to the later state (ksw2_extd2_sse.c:285-305)
						d = z > a ? 0 : 1;  z = z > a ? z : a;
						d = z > b ? d : 2;  z = z > b ? z : b;
						d = z > a2 ? d : 3; z = z > a2 ? z : a2;
						d = z > b2 ? d : 4; z = z > b2 ? z : b2;
					}
					if (sc_mch < z) z = sc_mch;
					u[t] = (int8_t)(z - vt1), v[t] = (int8_t)(z - ut);
					int tmp = sx8(z - q); a = sx8(a - tmp), b = sx8(b - tmp);
					tmp = sx8(z - q2); a2 = sx8(a2 - tmp), b2 = sx8(b2 - tmp);
					if (!right) {
						x[t]  = (int8_t)((a  > 0 ? a  : 0) - qe);  if (a  > 0) d |= 0x08;
						y[t]  = (int8_t)((b  > 0 ? b  : 0) - qe);  if (b  > 0) d |= 0x10;
						x2[t] = (int8_t)((a2 > 0 ? a2 : 0) - qe2); if (a2 > 0) d |= 0x20;
						y2[t] = (int8_t)((b2 > 0 ? b2 : 0) - qe2); if (b2 > 0) d |= 0x40;
					} else {
						x[t]  = (int8_t)((0 > a  ? 0 : a)  - qe);  if (!(0 > a))  d |= 0x08;
						y[t]  = (int8_t)((0 > b  ? 0 : b)  - qe);  if (!(0 > b))  d |= 0x10;
						x2[t] = (int8_t)((0 > a2 ? 0 : a2) - qe2); if (!(0 > a2)) d |= 0x20;
						y2[t] = (int8_t)((0 > b2 ? 0 : b2) - qe2); if (!(0 > b2)) d |= 0x40;
					}
					prow[t - st] = (uint8_t)d;
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			__syncthreads();
			bool stop = false;
			if (!approx_max) { // ksw2_extd2_sse.c:322-366
				int max_H, max_t;
				if (r > 0) {
					const int Hen = en0 > 0 ? H[en0 - 1] + u[en0] : H[en0] + v[en0];
					__syncthreads();
					const int en1 = st0 + (en0 - st0) / 4 * 4;
					long long best = ((long long)Hen << 32) | 0xffffffffu;                // the last cell wins every tie
					for (int t = st0 + lane; t < en0; t += 64) {
						const int h = H[t] + v[t];
						H[t] = h;
						const unsigned ord = t < en1 ? 1u + ((unsigned)((t - st0) & 3) << 28) + (unsigned)t : 1u + (4u << 28) + (unsigned)t;
						const long long key = ((long long)h << 32) | (0xffffffffu - ord);
						best = key > best ? key : best;
					}
					if (lane == 0) H[en0] = Hen;
					best = wave_max64(best);
					max_H = (int)(best >> 32);
					const unsigned ord = 0xffffffffu - (unsigned)(best & 0xffffffffLL);
					max_t = ord == 0 ? en0 : (int)((ord - 1) & 0x0fffffffu);
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
					__syncthreads();
				} else {
					if (lane == 0) H[0] = v[0] - qe_h;
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
					__syncthreads();
					max_H = H[0], max_t = 0;
				}
				if (en0 == tlen - 1) { const int h = H[en0]; if (h > ez_mte) ez_mte = h, ez_mte_q = r - en0; }
				if (r - st0 == qlen - 1) { const int h = H[st0]; if (h > ez_mqe) ez_mqe = h, ez_mqe_t = st0; }
				// ksw_apply_zdrop (ksw2.h:168-184)
				if (max_H > ez_max) ez_max = max_H, ez_max_t = max_t, ez_max_q = r - max_t;
				else if (max_t >= ez_max_t && r - max_t >= ez_max_q) {
					const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
					if (zdrop >= 0 && ez_max - max_H > zdrop + l * e2) { ez_zdropped = 1; stop = true; }
				}
				if (!stop && r == qlen + tlen - 2 && en0 == tlen - 1) ez_score = H[tlen - 1];
			} else {            // ksw2_extd2_sse.c:367-384: one tracked cell per diagonal
				if (r > 0) {
					if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
						const int d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
						if (d0 > d1) H0 += d0; else H0 += d1, ++last_H0_t;
					} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += v[last_H0_t];
					else ++last_H0_t, H0 += u[last_H0_t];
				} else H0 = v[0] - qe_h, last_H0_t = 0;
				if (flag & EZ_APPROX_DROP) {
					if (H0 > ez_max) ez_max = H0, ez_max_t = last_H0_t, ez_max_q = r - last_H0_t;
					else if (last_H0_t >= ez_max_t && r - last_H0_t >= ez_max_q) {
						const int tl = last_H0_t - ez_max_t, ql = (r - last_H0_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
						if (zdrop >= 0 && ez_max - H0 > zdrop + l * e2) { ez_zdropped = 1; stop = true; }
					}
				}
				if (!stop && r == qlen + tlen - 2 && en0 == tlen - 1) ez_score = H0;
			}
			if (stop) break;
			last_st = st, last_en = en;
		}

		// ---- backtrack (ksw2.h:127-159, is_rot=1) by lane 0; off/off_end are recomputed from r ----
		int n_cigar = 0;
		int bi = -1, bj = -1;
		if (!ez_zdropped && !(flag & EZ_EXTZ_ONLY)) bi = tlen - 1, bj = qlen - 1;
		else if (!ez_zdropped && (flag & EZ_EXTZ_ONLY) && ez_mqe + end_bonus > ez_max) ez_reach_end = 1, bi = ez_mqe_t, bj = qlen - 1;
		else if (ez_max_t >= 0 && ez_max_q >= 0) bi = ez_max_t, bj = ez_max_q;
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		__syncthreads();
		if (lane == 0 && bi >= 0 && bj >= 0) {
			int i = bi, j = bj, state = 0;
			uint32_t last_op = 0xffffffffu;
			auto push = [&](uint32_t op, uint32_t len) {
				if (n_cigar == 0 || op != last_op) { cig_tmp[n_cigar++] = len << 4 | op; last_op = op; }
				else cig_tmp[n_cigar - 1] += len << 4;
			};
			while (i >= 0 && j >= 0) {
				const int r = i + j;
				int st0, en0, force_state = -1;
				diag_range(r, qlen, tlen, w, st0, en0);
				const int off = st0 / 16 * 16, off_end = (en0 + 16) / 16 * 16 - 1;
				if (i < off) force_state = 2;
				if (i > off_end) force_state = 1;
				const uint32_t tmp = force_state < 0 ? pmat[(size_t)r * n_col + i - off] : 0;
				if (state == 0) state = tmp & 7;
				else if (!(tmp >> (state + 2) & 1)) state = 0;
				if (state == 0) state = tmp & 7;
				if (force_state >= 0) state = force_state;
				if (state == 0) push(0, 1), --i, --j;
				else if (state == 1 || state == 3) push(2, 1), --i;
				else push(1, 1), --j;
			}
			if (i >= 0) push(2, (uint32_t)(i + 1));
			if (j >= 0) push(1, (uint32_t)(j + 1));
		}
		n_cigar = __shfl(n_cigar, 0);
		unsigned long long base = 0;
		if (lane == 0 && n_cigar > 0) base = atomicAdd(pool_cursor, (unsigned long long)n_cigar);
		base = ((unsigned long long)(unsigned)__shfl((int)(base >> 32), 0) << 32) | (unsigned)__shfl((int)(base & 0xffffffffULL), 0);
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		__syncthreads();
		const bool rev_cigar = flag & EZ_REV_CIGAR;
		if (base + (unsigned long long)n_cigar <= pool_cap)
			for (int c = lane; c < n_cigar; c += 64) cigar_pool[base + c] = rev_cigar ? cig_tmp[c] : cig_tmp[n_cigar - 1 - c];
		if (lane == 0) {
			DpRes R;
			R.max = ez_max, R.max_q = ez_max_q, R.max_t = ez_max_t, R.mqe = ez_mqe, R.mqe_t = ez_mqe_t, R.mte = ez_mte, R.mte_q = ez_mte_q;
			R.score = ez_score, R.zdropped = ez_zdropped, R.reach_end = ez_reach_end, R.n_cigar = n_cigar, R.cigar_off = base;
			res[jid] = R;
		}
		__syncthreads();
	}
}

size_t dp_slab_bytes(int qlen, int tlen, int w)
{
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	size_t tlen16 = ((size_t)tlen + 15) / 16 * 16;
	size_t n_col = (size_t)(qlen < tlen ? qlen : tlen);
	n_col = (((n_col < (size_t)w + 1 ? n_col : (size_t)w + 1) + 15) / 16 + 1) * 16;
	size_t b = 0;
	if (tlen16 > LDS_T) b += ((7 * tlen16 + 15) & ~(size_t)15) + 4 * tlen16;   // only the single-wave fallback kernel keeps rows here
	b += (((size_t)(qlen + tlen - 1) * n_col + 15) & ~(size_t)15);
	b += 4 * ((size_t)qlen + tlen + 8);
	return (b + 255) & ~(size_t)255;
}

void launch_extd2_fast(int C, unsigned n_waves, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter, uint8_t *slab, size_t slab_bytes,
                       DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st);

size_t wide_lds_bytes(int r_cap, int seq_cap, bool exact);
void launch_extd2_wide(unsigned n_blocks, int n_threads, int r_cap, int seq_cap, bool exact, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter, uint8_t *slab, size_t slab_bytes,
                       DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st);

// Problem classes (each is one persistent launch):
//   0,1    register-resident kernel (pga_ksw_fast.hip), target <= 256 / <= 512 lanes, band never binding
//   2,3,4  workgroup kernel (pga_ksw_wide.hip) by LDS footprint of the band ring + sequences: <= 48 KB (three workgroups
//          per CU), <= 76 KB (two), <= 152 KB (one)
//   5      single-wave kernel with rows in the HBM slab (targets wider than LDS can hold; not reached by pangraph's windows)
//   6      local-alignment score queries of the inversion test (pga_ll.hip)
//   8      first-pass gap fills of nearly equal length: corridor kernel with an exactness proof per problem (pga_ksw_band.hip);
//          the few it cannot prove come back flagged and take their normal class in a second pass
//   9      large unbanded first-pass gap fills, one problem spread over several workgroups in column strips (pga_ksw_strips.hip)
//   10     banded problems whose band ring fits 2048 columns (end extensions, banded fills): rows in registers, one barrier per
//          diagonal (pga_ksw_lanes.hip); what classes 2 and 3 held before
//   11     like 10 with ONE wave per problem: rings of up to 512 columns (end extensions next to a block end, narrow banded fills)
//   7      like 4, but exact-maximum problems (14 instead of 10 B of LDS per column: launched apart so that the approximate
//          first passes of class 4 keep room for their sequences in LDS)
#define DP_NCLASS 14
#define WIDE_LDS_MAX (152 * 1024)
static inline int wide_ring(const DpJob &j)
{
	const int T = (j.tlen + 15) / 16 * 16;
	int w = j.w < 0 ? (j.tlen > j.qlen ? j.tlen : j.qlen) : j.w;
	int R = ((w < j.tlen ? w : j.tlen) + 15) / 16 * 16 + 96;
	return R > T ? T : R;
}
static inline int wide_seqcap(const DpJob &j) { return ((j.qlen > j.tlen ? j.qlen : j.tlen) + 15) / 16 * 16; }
size_t ll_lds_bytes(int t_cap);
size_t ll_multi_scratch_bytes();
int ll_groups(uint32_t n_jobs, int t_max);
void launch_ll_multi(int G, int t_cap, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, unsigned long long *scratch, DpRes *res, hipStream_t st);
void launch_ll_i16(unsigned n_blocks, int t_cap, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter,
                   unsigned long long *rowkey, size_t rowkey_stride, DpRes *res, hipStream_t st);

#define BAND_MAXLEN 1024
size_t band_slab_bytes(int max_diag);
void launch_gapfill_band(unsigned n_waves, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter, uint8_t *slab, size_t slab_bytes,
                         DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st);

bool strips_eligible(const DpJob &j, const DpParams &P);
size_t strips_slab_bytes(const DpJob &j);
int strips_count(const DpJob &j);
size_t strips_bnd_words(const DpJob &j);
void launch_approx_strips(unsigned n_blocks, const DpJob *jobs, const uint32_t *blk_job, const uint32_t *blk_strip, PkBases bases, const DpParams &P, uint8_t *slab, const uint64_t *slab_off,
                          uint32_t *bnd, const uint64_t *bnd_off, uint32_t *done_ctr, DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st);

bool wstrips_on();
bool wstrips_eligible(const DpJob &j, const DpParams &P);
int wstrips_count(const DpJob &j);
size_t wstrips_bnd_words(const DpJob &j);
void launch_wstrips(unsigned n_blocks, const DpJob *jobs, const uint32_t *blk_job, const uint32_t *blk_strip, PkBases bases, const DpParams &P, uint8_t *slab, const uint64_t *slab_off,
                    unsigned long long *bnd, const uint64_t *bnd_off, uint32_t *done_ctr, DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st);
int bstrips_mode();
int bstrips_max_problems();
int bstrips_long_diagonals();
bool bstrips_eligible(const DpJob &j, const DpParams &P);
size_t bstrips_slab_bytes(const DpJob &j);
size_t bstrips_words(const DpJob &j);
uint32_t bstrips_table(const DpJob &j, std::vector<uint32_t> &tab, size_t *words);
void launch_bstrips(unsigned n_blocks, const DpJob *jobs, const uint32_t *blk_job, PkBases bases, const DpParams &P, uint8_t *slab, const uint64_t *slab_off,
                    unsigned long long *bnd, const uint64_t *bnd_off, const uint32_t *tab, const uint64_t *tab_off, DpRes *res, uint32_t *pool, unsigned long long *cursor,
                    unsigned long long pool_cap, hipStream_t st);
bool pipe_eligible(const DpJob &j);
int pipe_mode();
size_t pipe_cig_bytes(int q_cap, int t_cap);
size_t pipe_chunk_bytes();
int pipe_max_chunks();
void launch_ext_pipe(unsigned n_blocks, int q_cap, int t_cap, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter, uint8_t *slab, uint32_t n_chunks,
                     DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st);
bool lanes_eligible(const DpJob &j, int nt);
size_t lanes_cig_bytes(int q_cap, int t_cap);
size_t lanes_chunk_bytes(int nt);
void launch_extd2_lanes(int nt, unsigned n_blocks, int q_cap, int t_cap, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter, uint8_t *slab, uint32_t n_chunks,
                        DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st);

static int dp_class(const DpJob &j, bool allow_band, const DpParams &P)
{
	static const bool no_lanes = getenv("PGA_NO_LANES") != nullptr;     // A/B: the workgroup kernel takes the banded problems again
	if (j.flag & PGA_JOB_LL) return 6;
	if ((wstrips_on() ? wstrips_eligible(j, P) : strips_eligible(j, P)) && (allow_band || (j.flag & EZ_APPROX_MAX))) return 9;      // (exact problems the strips handed back go to the workgroup kernel)
	if (allow_band && j.flag == EZ_APPROX_MAX && j.w >= j.qlen && j.w >= j.tlen && j.qlen >= 1 && j.tlen >= 1 && j.qlen <= BAND_MAXLEN && j.tlen <= BAND_MAXLEN &&
	    j.tlen - j.qlen <= 12 && j.qlen - j.tlen <= 12) return 8;
	const bool unbanded = j.w >= j.qlen && j.w >= j.tlen;
	if (unbanded && j.tlen <= 256) return 0;
	if (unbanded && j.tlen <= 512) return 1;
	static const bool no_narrow = getenv("PGA_NO_LANES_NARROW") != nullptr;   // A/B: narrow rings go to the four-wave workgroups again
	if (!no_lanes && !no_narrow && allow_band && lanes_eligible(j, 64)) return 11;
	if (!no_lanes && allow_band && lanes_eligible(j, 256)) return 10;      // (allow_band is off in the second pass over problems a kernel handed back)
	// (1024-thread workgroups were measured slower than the workgroup kernel: sixteen waves each pay the per-diagonal skeleton, four to a SIMD
	// -- 2 x 4 kb second passes 56 ms against 11-20 ms)
	const size_t rows = (size_t)14 * wide_ring(j), l = rows + 2 * (size_t)wide_seqcap(j);
	if (l <= 48 * 1024) return 2;
	if (l <= 76 * 1024) return 3;
	if (rows <= WIDE_LDS_MAX) return (j.flag & EZ_APPROX_MAX) ? 4 : 7;
	return 5;
}

// a few host threads for the per-problem loops (classification, gather, scatter): millions of problems per round
template <class F> static void host_parallel(size_t n, F f)
{
	const int nt = (int)std::min<size_t>((size_t)thread_budget(), n / 65536 + 1);
	if (nt <= 1) { f(0, n); return; }
	const size_t per = (n + nt - 1) / nt;
	pool_for((size_t)nt, nt, [&](size_t t) { const size_t lo = std::min(n, t * per), hi = std::min(n, lo + per); if (lo < hi) f(lo, hi); });
}

static void dp_run_impl(PkBases d_bases, const std::vector<DpJob> &jobs, const DpParams &P, std::vector<DpRes> &res, PinVec<uint32_t> &cigars, hipStream_t st, Timers *tm, int band_level);   // 0: no banded kernels, 1: lane kernels + strips, 2: and the workgroup pipeline

// Launch lanes: a grow-only scratch slab per lane and SET (and, with PGA_DP_SHARED=0, a priority stream per lane: lane 0, the million-tile class, at the lower priority).
// Sets are pooled per device and leased for one dp_run call: as many sets exist as calls ever ran concurrently on a device, whatever
// the number of host threads that came and went.  A slab lives in the set's own device-memory arena (blocks of a set are only ever
// used on the set's streams); a set returns to the pool with its streams drained.
#define DP_NLANE 9
struct LaneSet { int dev = 0, arena = 0; hipStream_t stream[DP_NLANE] = {}; DBuf<uint8_t> slab[DP_NLANE]; };

// SHARED launch streams (round 5).  With a set of four streams per concurrent call, six batches in flight (two query sets each) hold up to 48 lane
// streams on the 6 + 6 hardware queues of the two priority pools: four streams per queue, dealt by creation order -- and kernels of streams that share a
// queue run one after the other, so a 2 ms tile launch of one batch sat behind a 48 ms extension launch of another whenever their lanes happened to
// share a queue.  Now the device has ONE pool of launch streams, as many as there are hardware queues for them (PGA_DP_STREAMS, default 6 low + 6
// high priority, created together so that each lands on a queue of its own); a launch goes to the stream with the least estimated work outstanding
// (long classes and short classes therefore spread over the queues instead of colliding by accident), and the estimate is taken off when the host
// has seen the launch complete.  A call still leases a set of scratch slabs for itself (LaneSet: no stream of its own any more).
struct DpStreamPool {
	std::mutex mu; int dev = -1; std::vector<hipStream_t> st; std::vector<double> load; std::vector<int> pending;
	void init(int d)
	{
		dev = d;
		int n_lo = 6, n_hi = 6;
		if (const char *e = getenv("PGA_DP_STREAMS")) { if (sscanf(e, "%d,%d", &n_lo, &n_hi) != 2) n_lo = n_hi = 6; }
		int prio_lo = 0, prio_hi = 0;
		PGA_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
		for (int i = 0; i < n_lo + n_hi; ++i) { hipStream_t s; PGA_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, i < n_lo ? prio_lo : prio_hi)); st.push_back(s); }
		load.assign(st.size(), 0.0); pending.assign(st.size(), 0);
	}
};
static std::mutex g_dps_mu; static std::vector<DpStreamPool*> g_dps;
static DpStreamPool &dp_stream_pool(int dev)
{
	std::lock_guard<std::mutex> lk(g_dps_mu);
	for (DpStreamPool *p : g_dps) if (p->dev == dev) return *p;
	DpStreamPool *p = new DpStreamPool(); p->init(dev); g_dps.push_back(p); return *p;
}
static int dp_stream_pick(DpStreamPool &P, double est_ms)
{
	std::lock_guard<std::mutex> lk(P.mu);
	size_t best = 0;
	for (size_t i = 1; i < P.st.size(); ++i) if (P.load[i] < P.load[best] || (P.load[i] == P.load[best] && P.pending[i] < P.pending[best])) best = i;
	P.load[best] += est_ms; ++P.pending[best];
	return (int)best;
}
static void dp_stream_done(DpStreamPool &P, int i, double est_ms)
{
	std::lock_guard<std::mutex> lk(P.mu);
	P.load[(size_t)i] -= est_ms; if (--P.pending[(size_t)i] == 0 || P.load[(size_t)i] < 0) P.load[(size_t)i] = P.pending[(size_t)i] ? std::max(0.0, P.load[(size_t)i]) : 0.0;
}
static bool dp_shared_streams() { static const bool v = !(getenv("PGA_DP_SHARED") && getenv("PGA_DP_SHARED")[0] == '0'); return v; }   // slabs are grow-only and stay with the set: no allocation in the launch path
static std::mutex g_lane_mu;
static std::vector<LaneSet*> g_lane_idle;
static std::atomic<size_t> g_slab_total(0);                  // bytes held by the slabs of all sets of all devices
struct LaneLease {
	LaneSet *set = nullptr;
	LaneLease(int dev, const size_t (&need)[DP_NLANE])
	{
		{
			// best fit: the idle set that has to grow least; among those, the one that wastes least
			std::lock_guard<std::mutex> lk(g_lane_mu);
			long best = -1; size_t best_grow = 0, best_waste = 0;
			for (size_t i = 0; i < g_lane_idle.size(); ++i) if (g_lane_idle[i]->dev == dev) {
				size_t grow = 0, waste = 0;
				for (int l = 0; l < DP_NLANE; ++l) { const size_t c = g_lane_idle[i]->slab[l].cap; if (need[l] > c) grow += need[l] - c; else waste += c - need[l]; }
				if (best < 0 || grow < best_grow || (grow == best_grow && waste < best_waste)) best = (long)i, best_grow = grow, best_waste = waste;
			}
			if (best >= 0) { set = g_lane_idle[(size_t)best]; g_lane_idle.erase(g_lane_idle.begin() + best); }
		}
		if (set) return;
		set = new LaneSet(); set->dev = dev; set->arena = dev_lease_arena();
		int prio_lo = 0, prio_hi = 0;
		PGA_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));      // numerically lower = higher priority
		// The runtime keeps one pool of hardware queues PER PRIORITY (GPU_MAX_HW_QUEUES each): lanes at the low and at the high priority do
		// not share queues with one another nor with the batches' own (default-priority) streams.  Two lanes per pool, one of the two with the
		// long kernels (lane 2: end extensions, inversion tests; lane 3: strips) in each: with six batches in flight 12 + 12 streams on 6 + 6
		// queues (measured: lanes 1-3 all high 3.75-3.83 s per step, lanes {0,1} low / {2,3} high 3.60-3.64, {0,3} low / {1,2} high 3.49-3.61;
		// every lane at the default priority 4.6 s).  PGA_LANE_PRIO=lhhh etc. for experiments (l low, h high, n default).
		static const std::string pr = getenv("PGA_LANE_PRIO") && strlen(getenv("PGA_LANE_PRIO")) == DP_NLANE ? getenv("PGA_LANE_PRIO") : "lhhlhhlhl";
		if (!dp_shared_streams()) for (int l = 0; l < DP_NLANE; ++l) {
			if (pr[(size_t)l] == 'n') PGA_HIP(hipStreamCreateWithFlags(&set->stream[l], hipStreamNonBlocking));
			else PGA_HIP(hipStreamCreateWithPriority(&set->stream[l], hipStreamNonBlocking, pr[(size_t)l] == 'l' ? prio_lo : prio_hi));
		}
	}
	~LaneLease()
	{
		for (int l = 0; l < DP_NLANE; ++l) if (set->stream[l]) (void)sync_stream(set->stream[l]);
		// the slabs stay with the set as long as all sets together hold a reasonable share of the device; beyond that this set gives its
		// slabs back (to the block cache, which may drop them)
		static const size_t keep = (size_t)(getenv("PGA_SLAB_KEEP_GB") ? atof(getenv("PGA_SLAB_KEEP_GB")) : 96.0) << 30;
		if (g_slab_total.load() > keep) for (int l = 0; l < DP_NLANE; ++l) { g_slab_total -= set->slab[l].cap; set->slab[l].release(); }
		std::lock_guard<std::mutex> lk(g_lane_mu);
		g_lane_idle.push_back(set);
	}
	LaneLease(const LaneLease&) = delete; LaneLease &operator=(const LaneLease&) = delete;
};

void dp_lane_dump()
{
	std::lock_guard<std::mutex> lk(g_lane_mu);
	for (LaneSet *s : g_lane_idle) {
		fprintf(stderr, "[pga]   idle lane set (arena %d): slabs MB", s->arena);
		for (int l = 0; l < DP_NLANE; ++l) fprintf(stderr, " %zu", s->slab[l].cap >> 20);
		fprintf(stderr, "\n");
	}
}
// pga_trim(): the slabs of the sets no call holds at the moment go back to the block cache (an idle set's last user drained its launches)
size_t dp_trim_lane_sets()
{
	std::lock_guard<std::mutex> lk(g_lane_mu);
	size_t bytes = 0;
	for (LaneSet *s : g_lane_idle) for (int l = 0; l < DP_NLANE; ++l) { bytes += s->slab[l].cap; g_slab_total -= s->slab[l].cap; s->slab[l].release(); }
	return bytes;
}

int dp_lb_mode() { const char *e = getenv("PGA_LB"); return !e ? 0 : !strcmp(e, "off") || !strcmp(e, "0") ? 1 : !strcmp(e, "check") ? 2 : 0; }

void dp_run(PkBases d_bases, const std::vector<DpJob> &jobs, const DpParams &P, std::vector<DpRes> &res, PinVec<uint32_t> &cigars, hipStream_t st, Timers *tm)
{
	dp_run_impl(d_bases, jobs, P, res, cigars, st, tm, getenv("PGA_NO_BAND") == nullptr ? 2 : 0);
	// problems a kernel handed back, results spliced in (their CIGARs go behind the pool): -10 = the workgroup pipeline (pga_ksw_pipe.hip) met a sweep
	// that outgrows its ring or ran out of chunks -> the other banded kernels; -9 = the corridor kernel could not prove its answer exact, a banded
	// kernel met a clamped maximum or a dry pool -> the full-matrix / workgroup kernels
	for (int pass = 1; pass <= 2; ++pass) {
		std::vector<uint32_t> redo;
		for (size_t i = 0; i < res.size(); ++i) {
			if (res[i].n_cigar == -11) throw std::runtime_error("pga: the length-bound stop of an extension (pga_dp.h) closed on a record that the full sweep changed: qlen " + std::to_string(jobs[i].qlen) + " tlen " + std::to_string(jobs[i].tlen));
			if (res[i].n_cigar == (pass == 1 ? -10 : -9)) redo.push_back((uint32_t)i);
		}
		if (redo.empty()) continue;
		if (getenv("PGA_VERBOSE")) fprintf(stderr, "[pga]     %s: %zu of %zu problems handed back\n", pass == 1 ? "workgroup pipeline -> lane kernels / wave strips" : "corridor / banded kernels -> workgroup kernel", redo.size(), jobs.size());
		std::vector<DpJob> jb(redo.size());
		for (size_t k = 0; k < redo.size(); ++k) jb[k] = jobs[redo[k]];
		std::vector<DpRes> r2; PinVec<uint32_t> c2;
		dp_run_impl(d_bases, jb, P, r2, c2, st, tm, pass == 1 ? 1 : 0);
		const size_t base = cigars.size();
		cigars.resize(base + c2.size());
		if (c2.size()) memcpy(cigars.data() + base, c2.data(), c2.size() * sizeof(uint32_t));
		for (size_t k = 0; k < redo.size(); ++k) { res[redo[k]] = r2[k]; res[redo[k]].cigar_off += base; }
	}
}

static void dp_run_impl(PkBases d_bases, const std::vector<DpJob> &jobs, const DpParams &P, std::vector<DpRes> &res, PinVec<uint32_t> &cigars, hipStream_t st, Timers *tm, int band_level)
{
	const bool allow_band = band_level > 0;
	res.clear(); cigars.clear();
	const size_t n = jobs.size();
	if (n == 0) return;
	const double t_enter = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
	std::vector<uint32_t> cls[DP_NCLASS];
	size_t slab_max[DP_NCLASS] = {};
	std::vector<size_t> need(n);
	std::vector<uint8_t> cls_of(n);
	unsigned long long cig_total = 0;
	host_parallel(n, [&](size_t lo, size_t hi) {
		for (size_t i = lo; i < hi; ++i) {
			const bool is_ll = jobs[i].flag & PGA_JOB_LL;
			cls_of[i] = (uint8_t)dp_class(jobs[i], allow_band, P);
			need[i] = is_ll ? ll_multi_scratch_bytes() : cls_of[i] == 8 ? band_slab_bytes(jobs[i].qlen + jobs[i].tlen) : cls_of[i] == 9 ? strips_slab_bytes(jobs[i]) : dp_slab_bytes(jobs[i].qlen, jobs[i].tlen, jobs[i].w);
		}
	});
	{
		// bucket the problem ids by class, in index order, with per-thread partial counts
		const int nt = (int)std::min<size_t>((size_t)thread_budget(), n / 65536 + 1);
		const size_t per = (n + nt - 1) / nt;
		std::vector<std::array<size_t, DP_NCLASS>> cnt((size_t)nt);
		std::vector<std::array<size_t, DP_NCLASS>> mx((size_t)nt);
		std::vector<unsigned long long> cg((size_t)nt, 0);
		auto count = [&](int t) {
			cnt[t].fill(0); mx[t].fill(0);
			const size_t lo = std::min(n, (size_t)t * per), hi = std::min(n, lo + per);
			for (size_t i = lo; i < hi; ++i) {
				const int c = cls_of[i];
				++cnt[t][c];
				if (need[i] > mx[t][c]) mx[t][c] = need[i];
				if (!(jobs[i].flag & PGA_JOB_LL)) cg[t] += (unsigned long long)jobs[i].qlen + jobs[i].tlen + 2;   // worst case: one op per base
			}
		};
		pool_for((size_t)nt, nt, [&](size_t t) { count((int)t); });
		std::vector<std::array<size_t, DP_NCLASS>> base((size_t)nt);
		for (int c = 0; c < DP_NCLASS; ++c) {
			size_t tot = 0;
			for (int t = 0; t < nt; ++t) { base[t][c] = tot; tot += cnt[t][c]; if (mx[t][c] > slab_max[c]) slab_max[c] = mx[t][c]; }
			cls[c].resize(tot);
		}
		for (int t = 0; t < nt; ++t) cig_total += cg[t];
		auto fill = [&](int t) {
			std::array<size_t, DP_NCLASS> pos = base[t];
			const size_t lo = std::min(n, (size_t)t * per), hi = std::min(n, lo + per);
			for (size_t i = lo; i < hi; ++i) cls[cls_of[i]][pos[cls_of[i]]++] = (uint32_t)i;
		};
		pool_for((size_t)nt, nt, [&](size_t t) { fill((int)t); });
	}
	// class 12 (banded wave strips, pga_ksw_bstrips.hip): a launch that holds only a few banded exact problems is bound by the latency of ONE
	// of them on one CU -- those go over several CUs each; a launch that holds more keeps them on the lane kernels, but for its few longest (an
	// extension that runs its band out: 20 k diagonals).  "Few" is 32 (PGA_BSTRIPS_MAX).  Alone on the device a second self-merge round falls
	// from 6.3 to 3.5 ms and the 20 k-diagonal extension from 48 to 17.5 ms; with six batches in flight a step measures the same for 8 ... 64
	// problems per launch on strips (2.39-2.46 s against 2.42-2.49 s without; when every strip row spanned all diagonals -- 25 MB zeroed per
	// problem and launch -- 20 / 48 per launch cost 4 / 9 %)
	// class 13 (the workgroup pipeline, pga_ksw_pipe.hip): the lane kernel's four-wave class (rings of up to 2 048 columns: the end extensions);
	// PGA_PIPE=force: the one-wave class too (tests)
	// A single extension into unrelated sequence takes 4.4 ms there against 6.3 ms on the lane kernel (the waves the band has not reached cost nothing,
	// and no barrier paces the diagonal), 64 of them 4.8 against 6.7 ms; with more than a problem per CU in the launch the lane kernel's
	// throughput is the same (512: 9.3 against 8.9 ms, 4 096: 72 against 65 ms): the pipeline takes the launches of at most PGA_PIPE_MAX problems.
	// PGA_LB=check: only the lane kernel sweeps on behind the length-bound stop and compares (pga_ksw_lanes.hip); the pipeline and the wave strips would take the
	// stop unchecked -- so in the checked mode every problem the stop applies to stays with the lane kernel, and the check covers all of them
	const int lb_mode_now = dp_lb_mode();
	auto lb_kept_for_check = [&](const DpJob &j) {
		if (lb_mode_now != 2) return false;
		int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
		if (q2 + e2 < q + e) { std::swap(q, q2); std::swap(e, e2); }
		return lb_stop_of(j.qlen, j.tlen, j.w < 0 ? std::max(j.qlen, j.tlen) : j.w, j.flag, q, e, q2, e2, P.sc_mch, P.sc_mis, P.sc_ambi == 0 ? -e2 : P.sc_ambi, 0).on != 0;
	};
	static const size_t pipe_max = getenv("PGA_PIPE_MAX") ? (size_t)atoi(getenv("PGA_PIPE_MAX")) : 256;
	// (a launch that holds so few banded problems that the wave strips take them all keeps them there: a problem spread over two dozen CUs runs
	// its diagonal in 0.76 us, a pipeline on one CU in 1.35)
	const bool strips_take_all = bstrips_mode() == 1 && cls[10].size() + cls[11].size() <= (size_t)std::max(1, bstrips_max_problems());
	if (band_level >= 2 && pipe_mode() > 0 && (pipe_mode() == 2 || (cls[10].size() <= pipe_max && !strips_take_all))) {
		for (int c : {10, 11}) {
			if (c == 11 && pipe_mode() < 2) continue;
			std::vector<uint32_t> rest;
			for (uint32_t id : cls[c]) {
				if (pipe_eligible(jobs[id]) && !lb_kept_for_check(jobs[id])) { cls[13].push_back(id); cls_of[id] = 13; slab_max[13] = std::max(slab_max[13], need[id]); } else rest.push_back(id);
			}
			cls[c].swap(rest);
		}
		std::sort(cls[13].begin(), cls[13].end());
	}
	// stragglers of the one-wave class: an extension towards a block end that the length-bound stop (pga_dp.h) does NOT cover -- the query ends before the
	// band has slid off the target, or the target window is too long for the bounds to close -- sweeps its ~1 540 diagonals twenty columns wide, and a
	// launch of hundreds of covered ones (~150 diagonals each) waits for it: those go to the wave strips (one wave, a column per lane: 0.76 us per
	// diagonal against 2.1), whatever the launch holds
	std::vector<uint32_t> stragglers;
	if (allow_band && bstrips_mode() > 0 && dp_lb_mode() != 1) {
		int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
		if (q2 + e2 < q + e) { std::swap(q, q2); std::swap(e, e2); }
		const int sc_N = P.sc_ambi == 0 ? -e2 : P.sc_ambi;
		std::vector<uint32_t> rest;
		for (uint32_t id : cls[11]) {
			const DpJob &j = jobs[id];
			bool slow = false;
			if (!(j.flag & EZ_APPROX_MAX) && (int64_t)j.qlen + j.tlen > 600 && bstrips_eligible(j, P) && stragglers.size() < 64) {      // (a target window of more than 64 bases: never covered)
				const LbStop S = lb_stop_of(j.qlen, j.tlen, j.w < 0 ? std::max(j.qlen, j.tlen) : j.w, j.flag, q, e, q2, e2, P.sc_mch, P.sc_mis, sc_N, 0);
				slow = lb_mode_now == 2 ? !S.on : (!S.on || S.tail > -100);
			}
			if (slow) stragglers.push_back(id); else rest.push_back(id);
		}
		cls[11].swap(rest);
	}
	if (allow_band && bstrips_mode() > 0) {
		std::vector<uint32_t> elig;
		for (int c : {10, 11}) for (uint32_t id : cls[c]) if (bstrips_eligible(jobs[id], P) && !lb_kept_for_check(jobs[id])) elig.push_back(id);
		const size_t cap = (size_t)std::max(1, bstrips_max_problems());
		std::vector<uint32_t> take;
		if (bstrips_mode() == 2 || elig.size() <= cap) take = elig;
		else {
			// the longest ones only, as far as a z-drop can let them run: nominal diagonals above the threshold, largest first
			std::vector<uint32_t> lg;
			for (uint32_t id : elig) if (jobs[id].qlen + jobs[id].tlen - 1 >= bstrips_long_diagonals()) lg.push_back(id);
			std::stable_sort(lg.begin(), lg.end(), [&](uint32_t a, uint32_t b) { return jobs[a].qlen + jobs[a].tlen > jobs[b].qlen + jobs[b].tlen; });
			if (lg.size() > std::max<size_t>(2, cap / 4)) lg.resize(std::max<size_t>(2, cap / 4));   // (most of a bulk round's extensions are long by this measure and z-drop early: a handful, not all)
			take = lg;
		}
		{	// whatever the policy took: no more than 8 GB of matrices and boundary words per launch (the rest stays with the lane kernels)
			size_t bytes = 0, kept = 0;
			for (; kept < take.size(); ++kept) { bytes += bstrips_slab_bytes(jobs[take[kept]]) + 8 * bstrips_words(jobs[take[kept]]); if (bytes > ((size_t)8 << 30)) break; }
			take.resize(kept);
		}
		take.insert(take.end(), stragglers.begin(), stragglers.end());
		if (!take.empty()) {
			std::vector<uint8_t> mark(n, 0);
			for (uint32_t id : take) mark[id] = 1;
			for (int c : {10, 11}) {
				std::vector<uint32_t> rest;
				for (uint32_t id : cls[c]) if (!mark[id]) rest.push_back(id);
				cls[c].swap(rest);
			}
			std::sort(take.begin(), take.end());
			cls[12] = take;
			for (uint32_t id : take) { cls_of[id] = 12; need[id] = bstrips_slab_bytes(jobs[id]); slab_max[12] = std::max(slab_max[12], need[id]); }
		}
	}
	res.resize(n);
	DBuf<uint32_t> d_pool((size_t)cig_total + 1);
	// the CIGAR pool's cursor and every class's queue counters in one block, zeroed once (a memset dispatch per class before)
	DBuf<unsigned long long> d_cursor(1 + DP_NCLASS); d_cursor.zero(st);
	PGA_HIP(sync_stream(st));                      // the only use of the caller's stream: everything below is ordered inside the lane streams
	// The classes are independent persistent launches: each gets its own stream, so the handful of huge problems
	// (one workgroup each, latency-bound) run beside the millions of small tiles instead of after them.
	// (four streams, not one per class: HIP multiplexes streams onto a handful of hardware queues, and two classes that
	// land on the same queue run back to back)
	// (the classes of a lane share a scratch slab and a stream, i.e. run one behind the other: the classes of long single problems -- the lane kernel's bulk (10),
	// the wave strips (12), the workgroup pipeline (13), the inversion queries (6) -- each have a lane of their own since round 5: the bulk launch of a leaf
	// round used to start when the strips' 17 ms extension had finished, not beside it)
	// (and the tiles: a near-root call's 1 981 tile problems were launched behind the ONE problem of class 1 and the corridor fills on the same lane, 2.3 ms into
	// a round whose other classes take 2.2 ms: classes 0, 1 and 8 each on a lane of their own)
	static const int lane_of_class[DP_NCLASS] = {0, 7, 2, 3, 1, 1, 6, 1, 8, 3, 2, 1, 5, 4};   // tiles | the few largest problems | inversion queries + extensions | large problems
	int dev_id = 0; PGA_HIP(hipGetDevice(&dev_id));
	struct Launch { int c; int nt = 0; uint32_t *cnt_p = nullptr; hipStream_t cs = nullptr; int si = -1; double est = 0; bool zc = false; const DpJob *jobs_p = nullptr; DpRes *res_p = nullptr; PinVec<DpRes> hr; std::vector<uint32_t> *ids; PinVec<DpJob> jb; DBuf<DpJob> d_jobs; DBuf<DpRes> d_r; DBuf<uint32_t> d_cnt; size_t n_waves; hipEvent_t e0, e1;
	                DBuf<uint32_t> d_blk_job, d_blk_strip, d_bnd, d_tab; DBuf<uint64_t> d_slab_off, d_bnd_off, d_tab_off;
	                std::vector<uint32_t> h_bj, h_bs, h_tab; std::vector<uint64_t> h_so, h_bo, h_to; };   // (their host images: alive until the launch has been collected, so that nothing waits for the copies)   // (class 9: block tables, strip boundaries)
	std::vector<Launch> L;
	L.reserve(DP_NCLASS);
	// scratch budget per class: a slab is n_waves x the largest problem of the class, and n_waves is halved until it fits.  24 GB keeps
	// ~100 concurrent 10 kb x 10 kb direction matrices; concurrent parts and query sets each hold a lane set, so four of them stay inside HBM
	size_t budget = (size_t)(getenv("PGA_SLAB_GB") ? atof(getenv("PGA_SLAB_GB")) : 24.0) << 30;
	{ static const size_t dev_total = [] { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) != hipSuccess) tot = 0; return tot; }();   // (once: the query is a driver call)
	  if (dev_total && dev_total / 8 < budget) budget = dev_total / 8;
	  static const double share = [] { const char *e = getenv("PGA_MEM_SHARE"); const double v = e ? atof(e) : 1.0; return v > 0.0 && v <= 1.0 ? v : 1.0; }();
	  budget = (size_t)((double)budget * share); }
	const bool verbose = getenv("PGA_VERBOSE") != nullptr;
	auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t_begin = now();
	hipEvent_t ready;
	PGA_HIP(hipEventCreate(&ready));
	PGA_HIP(hipEventRecord(ready, st));                     // time base of the per-class start offsets printed under PGA_VERBOSE
	// scratch slabs: one grow-only buffer per launch lane (classes of a lane run one after the other and share it); sized
	// before anything is launched so that no buffer moves under a running kernel
	size_t waves_of[DP_NCLASS] = {0}, lane_need[DP_NLANE] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	uint32_t lanes_pool_chunks[2] = {0, 0}, pipe_pool_chunks = 0;
	for (int c = DP_NCLASS - 1; c >= 0; --c) {
		if (cls[c].empty()) continue;
		static const int c8w = getenv("PGA_C8_WAVES") ? atoi(getenv("PGA_C8_WAVES")) : 16;
		size_t n_waves = c == 11 ? 256 * (size_t)(getenv("PGA_C11_WAVES") ? atoi(getenv("PGA_C11_WAVES")) : 10) : c == 10 ? 256 * (size_t)(getenv("PGA_C10_WAVES") ? atoi(getenv("PGA_C10_WAVES")) : 2) : c == 8 ? 256 * (size_t)c8w : (c == 6 || c == 7) ? 256 : c == 5 ? 256 * 2 : c == 4 ? 256 : c == 3 ? 256 * 2 : c == 2 ? 256 * (getenv("PGA_C2_WAVES") ? atoi(getenv("PGA_C2_WAVES")) : 6) : 256 * 16;
		if (n_waves > cls[c].size()) n_waves = cls[c].size();
		if (c == 8) n_waves = std::min<size_t>(n_waves, (cls[c].size() + 1) / 2);      // a wave takes two problems at a time
		if (c == 9 || c == 12) {                                                        // every problem of the class is in flight at once, each with its whole matrix
			size_t tot = 0; for (uint32_t id : cls[c]) tot += need[id];
			waves_of[c] = cls[c].size();
			lane_need[lane_of_class[c]] = std::max(lane_need[lane_of_class[c]], tot);
			continue;
		}
		if (c == 13) {
			// one CIGAR buffer per workgroup + three chunks per workgroup (a workgroup keeps its chunks from problem to problem; a sweep that needs
			// more than the pool has left is handed back)
			static const int c13 = getenv("PGA_C13_WGS") ? atoi(getenv("PGA_C13_WGS")) : 2;
			n_waves = std::min<size_t>(cls[c].size(), 256 * (size_t)c13);
			int q_cap = 16, t_cap = 16;
			for (uint32_t id : cls[c]) { q_cap = std::max(q_cap, jobs[id].qlen); t_cap = std::max(t_cap, jobs[id].tlen); }
			const size_t chunk = pipe_chunk_bytes();
			const size_t pool = (n_waves * 3 + 64) * chunk;
			pipe_pool_chunks = (uint32_t)(pool / chunk);
			waves_of[c] = n_waves;
			lane_need[lane_of_class[c]] = std::max(lane_need[lane_of_class[c]], n_waves * pipe_cig_bytes(q_cap, t_cap) + pool + 256);
			continue;
		}
		if (c == 10 || c == 11) {
			// one CIGAR buffer per workgroup + a pool of direction-matrix chunks: what the class would need if every problem ran to its
			// last diagonal, but not more than half of the budget (most extensions z-drop early; a dry pool hands problems back)
			int q_cap = 16, t_cap = 16; size_t tot = 0;
			const size_t chunk = lanes_chunk_bytes(c == 11 ? 64 : 256);
			for (uint32_t id : cls[c]) { q_cap = std::max(q_cap, jobs[id].qlen); t_cap = std::max(t_cap, jobs[id].tlen); tot += need[id] + chunk; }
			const size_t cigb = n_waves * lanes_cig_bytes(q_cap, t_cap);
			// (round 6: a workgroup holds its first two chunks and asks one ahead; the bulk rounds of the BASELINE build -- 2 400 ... 3 150 extensions on 512
			// workgroups -- take 1 360 ... 1 614 chunks, and the pool used to be budget / 2 = 12 GB, which the block cache rounds to 16 GB per lane set: 96 GB
			// of the device held by six sets.  A quarter of the budget, CIGAR buffers included, so that the slab lands on a size class of the cache: 6 GB =
			// 3 000 chunks, twice what was seen; a dry pool still hands problems back)
			const size_t want = std::max(budget / 4, 5 * n_waves * chunk);
			size_t pool = std::min(tot, want > cigb + 256 + 2 * n_waves * chunk ? want - cigb - 256 : want);
			pool = std::max(pool, 2 * n_waves * chunk) / chunk * chunk;
			lanes_pool_chunks[c - 10] = (uint32_t)(pool / chunk) | (pool >= tot ? 0x80000000u : 0u);   // (top bit: the pool covers every problem in full: no reservation tiers)
			waves_of[c] = n_waves;
			lane_need[lane_of_class[c]] = std::max(lane_need[lane_of_class[c]], cigb + pool + 256);
			continue;
		}
		while (n_waves > 1 && n_waves * slab_max[c] > budget) n_waves /= 2;
		waves_of[c] = n_waves;
		lane_need[lane_of_class[c]] = std::max(lane_need[lane_of_class[c]], n_waves * slab_max[c]);
	}
	// four streams + four scratch slabs of this device, exclusive for the call (concurrent callers do not queue behind each other): the
	// idle set whose slabs fit the need best
	LaneLease lanes(dev_id, lane_need);
	// Leaves the scope BEFORE the lease does (declared after it): if anything throws between the first launch and the last collection, the launches that were
	// not collected are still running on the pool's shared streams -- they are drained here, and their estimates taken off the pool, before the slabs go back
	// to the idle list and the launches' device blocks to the arena.  (On the normal path every launch has been collected: si == -1, nothing to do.)
	struct DrainUncollected {
		std::vector<Launch> &L; int dev;
		~DrainUncollected() { for (Launch &X : L) if (X.si >= 0) { (void)sync_stream(X.cs); dp_stream_done(dp_stream_pool(dev), X.si, X.est); X.si = -1; } }
	} drain_uncollected{L, dev_id};
	hipStream_t *lane_stream = lanes.set->stream;
	DBuf<uint8_t> *lane_slab = lanes.set->slab;
	for (int l = 0; l < DP_NLANE; ++l) if (lane_need[l] > lane_slab[l].cap) {      // (the set is idle: its last user drained the streams)
		ArenaScope own(lanes.set->arena);
		g_slab_total -= lane_slab[l].cap; lane_slab[l].alloc(lane_need[l]); g_slab_total += lane_slab[l].cap;
	}
	// every class is prepared and launched in turn, the classes with few, long problems first: they are already running
	// while the host still lays out the million-tile classes
	// launch order: the classes of few, long problems first -- their workgroups need most of a CU's LDS and would otherwise wait until the
	// persistent waves of the million-problem classes (16 per CU, all of its LDS) have drained their queue
	int lane_si[DP_NLANE] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};
	static const int launch_order_bulk[DP_NCLASS] = {13, 12, 9, 11, 7, 6, 5, 4, 3, 10, 2, 8, 1, 0};
	// a round of the upper tree (a few hundred tiles, one or two banded fills): what it waits for are its longest single problems -- a thin tile of
	// 10 k diagonals, an approximate fill on the lane kernel -- and every launch costs the host ~60 us: those go out first (a first-round call: the
	// tile class ended 0.6-0.8 ms behind the others only because it was launched last)
	static const int launch_order_small[DP_NCLASS] = {13, 10, 0, 1, 12, 9, 11, 7, 6, 5, 4, 3, 2, 8};
	static const bool small_first = getenv("PGA_DP_SMALL_ORDER") != nullptr;     // (measured: the build is 5 % SLOWER with it -- off)
	const int *launch_order = small_first && cls[0].size() + cls[1].size() <= 4096 ? launch_order_small : launch_order_bulk;
	for (int oi = 0; oi < DP_NCLASS; ++oi) {
		const int c = launch_order[oi];
		if (cls[c].empty()) continue;
		std::vector<uint32_t> &ids = cls[c];
		// biggest problems first, so that the persistent waves finish together (the many small tiles of the
		// register-resident classes are uniform enough to skip the sort)
		if ((c >= 2 && c != 8) || ids.size() < 100000)
			std::stable_sort(ids.begin(), ids.end(), [&](uint32_t a, uint32_t b) { return (size_t)jobs[a].qlen * jobs[a].tlen > (size_t)jobs[b].qlen * jobs[b].tlen; });
		L.emplace_back();
		Launch &X = L.back();
		X.c = c; X.ids = &ids;
		PinVec<DpJob> &jb = X.jb; jb.resize(ids.size());       // stays alive until the class has been collected
		host_parallel(ids.size(), [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) jb[i] = jobs[ids[i]]; });
		// a launch of few problems takes its descriptors straight from the pinned list and writes its records straight into pinned memory (the
		// host's pointers are the device's): no copy in, no copy + synchronisation out -- three runtime dispatches and ~50 us less per class
		// and round, which is what the rounds of the upper tree are made of.  (A descriptor read costs a trip over the host link: not for the
		// launches that hold thousands of small tiles.)
		static const size_t zc_max = getenv("PGA_DP_ZEROCOPY_MAX") ? (size_t)atol(getenv("PGA_DP_ZEROCOPY_MAX")) : 2048;
		X.zc = ids.size() <= zc_max;
		if (X.zc) { X.hr.resize(ids.size()); X.jobs_p = jb.data(); X.res_p = X.hr.data(); }
		else { X.d_jobs.alloc(ids.size()); X.d_r.alloc(ids.size()); X.jobs_p = X.d_jobs.p; X.res_p = X.d_r.p; }
		if (c == 9) { X.d_cnt.alloc(ids.size()); X.cnt_p = X.d_cnt.p; } else X.cnt_p = reinterpret_cast<uint32_t*>(d_cursor.p + 1 + c);               // (class 12 keeps its counters in the problems' control blocks)               // (class 10: [1] is the cursor of its chunk pool)                // (class 9: one completion counter per problem)
		X.n_waves = waves_of[c];
		uint8_t *slab_p = lane_slab[lane_of_class[c]].p;
		static const bool serial = getenv("PGA_DP_SERIAL") != nullptr;       // diagnosis: every class alone on the GPU, one after the other
		static const bool on_main = getenv("PGA_DP_ON_MAIN") != nullptr;    // experiment: every class on the call's own stream (one hardware queue per batch)
		static const bool three = getenv("PGA_DP_THREE_LANES") != nullptr;   // experiment: lane 3's classes (strips, class 3) share lane 1's stream
		hipStream_t cs;
		if (on_main) cs = st;
		else if (dp_shared_streams()) {
			// what the launch is expected to hold its queue for (ms): the long banded classes by their longest problem's nominal diagonals, the others by count
			static const double base_ms[DP_NCLASS] = {1.5, 2.0, 4.0, 6.0, 10.0, 10.0, 10.0, 10.0, 0.5, 12.0, 5.0, 4.0, 2.0, 3.0};
			double est = base_ms[c] * std::max(1.0, (double)ids.size() / (c <= 1 || c == 8 ? 20000.0 : c == 10 || c == 11 || c == 13 ? 512.0 : 64.0));
			// (the classes of one lane share a scratch slab and therefore one stream: the lane's first launch picks it)
			DpStreamPool &SP = dp_stream_pool(dev_id);
			int &lsi = lane_si[lane_of_class[c]];
			if (lsi < 0) lsi = dp_stream_pick(SP, est);
			else { std::lock_guard<std::mutex> lk(SP.mu); SP.load[(size_t)lsi] += est; ++SP.pending[(size_t)lsi]; }
			X.si = lsi; X.est = est; cs = SP.st[(size_t)X.si];
		} else cs = lane_stream[serial ? 0 : (three && lane_of_class[c] == 3) ? 1 : lane_of_class[c]];
		X.cs = cs;
		// the problem list and the queue counter travel in the class's own lane stream: a copy queued in another stream can sit
		// behind a long kernel that happens to share its hardware queue (streams outnumber the queues), and the host would wait for it
		if (!X.zc) PGA_HIP(hipMemcpyAsync(X.d_jobs.p, jb.data(), jb.size() * sizeof(DpJob), hipMemcpyHostToDevice, cs));
		if (c == 9) PGA_HIP(hipMemsetAsync(X.d_cnt.p, 0, sizeof(uint32_t) * X.d_cnt.n, cs));
		PGA_HIP(hipEventCreate(&X.e0)); PGA_HIP(hipEventCreate(&X.e1));
		PGA_HIP(hipEventRecord(X.e0, cs));
		if (c == 9) {
			std::vector<uint32_t> &bj = X.h_bj, &bs = X.h_bs; std::vector<uint64_t> &so = X.h_so, &bo = X.h_bo; so.assign(ids.size(), 0); bo.assign(ids.size(), 0);
			uint64_t s_acc = 0, b_acc = 0;
			const bool ws = wstrips_on();                                   // wave strips (pga_ksw_wstrips.hip): 64-bit boundary words
			for (size_t i = 0; i < ids.size(); ++i) {
				const DpJob &j = jobs[ids[i]];
				so[i] = s_acc; s_acc += need[ids[i]];
				bo[i] = b_acc; b_acc += ws ? wstrips_bnd_words(j) : strips_bnd_words(j);
				for (int k2 = 0; k2 < (ws ? wstrips_count(j) : strips_count(j)); ++k2) { bj.push_back((uint32_t)i); bs.push_back((uint32_t)k2); }
			}
			X.d_blk_job.upload(bj, cs); X.d_blk_strip.upload(bs, cs); X.d_slab_off.upload(so, cs); X.d_bnd_off.upload(bo, cs);
			X.d_bnd.alloc(((size_t)b_acc + 1) * (ws ? 2 : 1)); X.d_bnd.zero(cs);
			if (ws) launch_wstrips((unsigned)bj.size(), X.jobs_p, X.d_blk_job.p, X.d_blk_strip.p, d_bases, P, slab_p, X.d_slab_off.p, (unsigned long long*)X.d_bnd.p, X.d_bnd_off.p, X.cnt_p, X.res_p, d_pool.p, d_cursor.p, cig_total, cs);
			else
			launch_approx_strips((unsigned)bj.size(), X.jobs_p, X.d_blk_job.p, X.d_blk_strip.p, d_bases, P, slab_p, X.d_slab_off.p, X.d_bnd.p, X.d_bnd_off.p, X.cnt_p, X.res_p, d_pool.p, d_cursor.p, cig_total, cs);
		} else if (c == 12) {
			std::vector<uint32_t> &bj = X.h_bj, &tab = X.h_tab; std::vector<uint64_t> &so = X.h_so, &bo = X.h_bo, &to = X.h_to; so.assign(ids.size(), 0); bo.assign(ids.size(), 0); to.assign(ids.size(), 0);
			uint64_t s_acc = 0, b_acc = 0;
			for (size_t i = 0; i < ids.size(); ++i) {
				const DpJob &j = jobs[ids[i]];
				so[i] = s_acc; s_acc += need[ids[i]];
				to[i] = tab.size();
				size_t words = 0;
				const uint32_t pool_waves = bstrips_table(j, tab, &words);
				bo[i] = b_acc; b_acc += words;
				for (uint32_t k2 = 0; k2 < pool_waves; ++k2) bj.push_back((uint32_t)i);
			}
			// (blocks of one problem are consecutive: the waves of a pool are dispatched together)
			X.d_blk_job.upload(bj, cs); X.d_slab_off.upload(so, cs); X.d_bnd_off.upload(bo, cs); X.d_tab.upload(tab, cs); X.d_tab_off.upload(to, cs);
			X.d_bnd.alloc(((size_t)b_acc + 1) * 2); X.d_bnd.zero(cs);
			X.n_waves = bj.size();
			launch_bstrips((unsigned)bj.size(), X.jobs_p, X.d_blk_job.p, d_bases, P, slab_p, X.d_slab_off.p, (unsigned long long*)X.d_bnd.p, X.d_bnd_off.p, X.d_tab.p, X.d_tab_off.p,
			               X.res_p, d_pool.p, d_cursor.p, cig_total, cs);
		} else if (c == 8) launch_gapfill_band((unsigned)X.n_waves, X.jobs_p, (uint32_t)ids.size(), d_bases, P, X.cnt_p, slab_p, slab_max[c], X.res_p, d_pool.p, d_cursor.p, cig_total, cs);
		else if (c == 6) {
			int t_cap = 16, t_max = 1;
			for (uint32_t id : ids) { t_cap = std::max(t_cap, std::max((jobs[id].tlen + 15) / 16 * 16, (jobs[id].qlen + 15) / 16 * 16)); t_max = std::max(t_max, jobs[id].tlen); }
			// a problem over several workgroups (pga_ll.hip: k_ll_multi) when the targets are long and the launch leaves the CUs for it; the lane's slab holds
			// a scratch record per problem (n_waves x slab_max[6] bytes >= problems x record: the groups of all problems are resident at once, so problems <= 224)
			const int G = ll_groups((uint32_t)ids.size(), t_max);
			if (G > 1 && ids.size() * ll_multi_scratch_bytes() <= X.n_waves * slab_max[c])
				launch_ll_multi(G, t_cap, X.jobs_p, (uint32_t)ids.size(), d_bases, P, (unsigned long long*)slab_p, X.res_p, cs);
			else
				launch_ll_i16((unsigned)X.n_waves, t_cap, X.jobs_p, (uint32_t)ids.size(), d_bases, P, X.cnt_p, (unsigned long long*)slab_p, slab_max[c] / 8, X.res_p, cs);
		} else if (c <= 1) launch_extd2_fast(c == 0 ? 4 : 8, (unsigned)X.n_waves, X.jobs_p, (uint32_t)ids.size(), d_bases, P, X.cnt_p, slab_p, slab_max[c], X.res_p, d_pool.p, d_cursor.p, cig_total, cs);
		else if (c == 13) {
			int q_cap = 16, t_cap = 16;
			for (uint32_t id : ids) q_cap = std::max(q_cap, jobs[id].qlen), t_cap = std::max(t_cap, jobs[id].tlen);
			launch_ext_pipe((unsigned)X.n_waves, q_cap, t_cap, X.jobs_p, (uint32_t)ids.size(), d_bases, P, X.cnt_p, slab_p, pipe_pool_chunks, X.res_p, d_pool.p, d_cursor.p, cig_total, cs);
		} else if (c == 10 || c == 11) {
			int q_cap = 16, t_cap = 16;
			for (uint32_t id : ids) q_cap = std::max(q_cap, jobs[id].qlen), t_cap = std::max(t_cap, jobs[id].tlen);
			launch_extd2_lanes(c == 11 ? 64 : 256, (unsigned)X.n_waves, q_cap, t_cap, X.jobs_p, (uint32_t)ids.size(), d_bases, P, X.cnt_p, slab_p, lanes_pool_chunks[c - 10], X.res_p, d_pool.p, d_cursor.p, cig_total, cs);
		} else if (c <= 4 || c == 7) {
			int r_cap = 0, seq_cap = 0; bool exact = false;
			for (uint32_t id : ids) { r_cap = std::max(r_cap, wide_ring(jobs[id])); seq_cap = std::max(seq_cap, wide_seqcap(jobs[id])); exact |= !(jobs[id].flag & EZ_APPROX_MAX); }
			if (wide_lds_bytes(r_cap, seq_cap, exact) > WIDE_LDS_MAX) seq_cap = 0;      // sequences stay in HBM for this launch
			// thousands of end extensions (class 2): they are bound by the per-diagonal latency of a workgroup, not by throughput, so
			// what counts is how many of them a CU holds -- without the 2 x 10 KB sequence copies six fit instead of three (-20 %)
			static const int c2_seq = getenv("PGA_C2_SEQ") ? atoi(getenv("PGA_C2_SEQ")) : 0;
			if (c == 2 && ids.size() > 1536 && !c2_seq) seq_cap = 0;
			// few problems: each workgroup effectively owns a CU, so give it the waves to hide its LDS latency -- as far as the
			// band has work for them (every wave of the group pays every phase of a diagonal): ~2 columns per thread
			int nt = 256;
			if (c == 4 || c == 7 || ids.size() <= 512) nt = r_cap >= 3000 ? 1024 : r_cap >= 1200 ? 512 : 256;
			if (getenv("PGA_WIDE_NT")) nt = atoi(getenv("PGA_WIDE_NT"));
			else if (c == 3) nt = 512;
			X.nt = nt;
			launch_extd2_wide((unsigned)X.n_waves, nt, r_cap, seq_cap, exact, X.jobs_p, (uint32_t)ids.size(), d_bases, P, X.cnt_p, slab_p, slab_max[c], X.res_p, d_pool.p, d_cursor.p, cig_total, cs);
		} else hipLaunchKernelGGL(k_extd2, dim3((unsigned)X.n_waves), dim3(64), 0, cs, X.jobs_p, (uint32_t)ids.size(), d_bases, P, X.cnt_p, slab_p, slab_max[c],
		                        X.res_p, d_pool.p, d_cursor.p, cig_total);
		PGA_HIP(hipGetLastError());
		PGA_HIP(hipEventRecord(X.e1, cs));
		(void)hipStreamQuery(cs);                             // push the packets out now: the class should start while the next one is prepared
	}
	const double t_launched = now();
	for (Launch &X : L) {
		const int c = X.c;
		std::vector<uint32_t> &ids = *X.ids;
		PGA_HIP(sync_event(X.e1));
		float msf = 0, ms_off = 0; PGA_HIP(hipEventElapsedTime(&msf, X.e0, X.e1)); (void)hipEventElapsedTime(&ms_off, ready, X.e0);
		const double ms = msf;
		busy_note(c == 6 ? K_LL : c == 8 ? K_BAND : (c == 9 || c == 12) ? K_STRIPS : (c == 10 || c == 11) ? K_LANES : c == 13 ? K_PIPE : c <= 1 ? K_EXTD2 : X.nt >= 1024 ? K_WIDE1024 : X.nt >= 512 ? K_WIDE512 : K_EXTD2_WIDE, X.e0, X.e1);
		(void)hipEventDestroy(X.e0); (void)hipEventDestroy(X.e1);
		if (verbose) fprintf(stderr, "[pga]     dp class %d: %zu problems, %.3f ms (queued at +%.1f ms), slab %.1f KB x %zu waves\n", c, ids.size(), ms, ms_off, slab_max[c] / 1024.0, X.n_waves);
		PinVec<DpRes> r;
		if (X.zc) r = std::move(X.hr); else download_to(r, X.d_r.p, ids.size(), X.cs);
		if (X.si >= 0) { dp_stream_done(dp_stream_pool(dev_id), X.si, X.est); X.si = -1; }
		if (tm) {
			// algorithmic bytes: 2-bit packed q+t reads; the tile kernel also gets the CIGAR bytes below.
			// cells: what the kernel's loops evaluated -- the corridor kernel 32 columns on every diagonal, the register tiles the whole
			// matrix (none when the identity proof answers), the workgroup kernel the band on the diagonals it ran (DpRes.pad = diagonals
			// done: z-drop ends an extension early), the local-alignment kernel the padded matrix
			std::vector<double> part_b((size_t)thread_budget() + 1, 0.0), part_c((size_t)thread_budget() + 1, 0.0);
			std::atomic<int> slot(0);
			host_parallel(ids.size(), [&](size_t lo, size_t hi) {
				double bases = 0, cells = 0;
				for (size_t i = lo; i < hi; ++i) {
					const DpJob &j = jobs[ids[i]];
					bases += (double)j.qlen + j.tlen;
					if (c == 8) cells += 32.0 * (double)(j.qlen + j.tlen - 1);
					else if (c == 9) cells += (double)j.qlen * j.tlen;
					else if (c == 6) cells += (double)((j.qlen + 7) / 8 * 8) * j.tlen;
					else if (c <= 1) cells += r[i].pad == 0x5A ? 0.0 : (double)j.qlen * j.tlen;
					else {
						const int nd = r[i].pad > 0 ? r[i].pad : j.qlen + j.tlen - 1, w = j.w < 0 ? (j.qlen > j.tlen ? j.qlen : j.tlen) : j.w;
						// the band's width per diagonal (ksw2_extd2_sse.c:173-181) is piecewise linear: every eighth diagonal stands for eight (a statistic:
						// exact but for the few diagonals where a bound changes; a loop over all diagonals of all problems cost a bulk round 8 ms of host time)
						auto width = [&](int d) { int st = 0, en = j.tlen - 1; if (st < d - j.qlen + 1) st = d - j.qlen + 1; if (en > d) en = d; if (st < (d - w + 1) >> 1) st = (d - w + 1) >> 1; if (en > (d + w) >> 1) en = (d + w) >> 1; return en >= st ? en - st + 1 : 0; };
						int d = 0;
						for (; d + 8 <= nd; d += 8) cells += 4.0 * (double)(width(d) + width(d + 7));
						for (; d < nd; ++d) cells += (double)width(d);
					}
				}
				const int s = slot.fetch_add(1) % (int)part_b.size();
				part_b[(size_t)s] += bases; part_c[(size_t)s] += cells;
			});
			double bases = 0, cells = 0; for (double x : part_b) bases += x; for (double x : part_c) cells += x;
			const int kk = c == 6 ? K_LL : c == 8 ? K_BAND : (c == 9 || c == 12) ? K_STRIPS : (c == 10 || c == 11) ? K_LANES : c == 13 ? K_PIPE : c <= 1 ? K_EXTD2 : X.nt >= 1024 ? K_WIDE1024 : X.nt >= 512 ? K_WIDE512 : K_EXTD2_WIDE;   // (wide: classes 2-5 and 7, by workgroup size)
			tm->kern[kk].ms += ms; tm->kern[kk].launches += 1; tm->kern[kk].alg_bytes += 0.5 * bases; tm->kern[kk].cells += cells; tm->dp_bases += bases;
		}
		host_parallel(ids.size(), [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) res[ids[i]] = r[i]; });
		if (const char *dump = getenv("PGA_DP_DUMP")) {           // development: geometry and outcome of every problem of the banded classes, appended as text
			if (c == 10 || c == 11 || c == 12 || c == 13 || c == 2 || c == 3) {
				static std::mutex dmu; std::lock_guard<std::mutex> lk(dmu);
				if (FILE *f = fopen(dump, "a")) {
					for (size_t i = 0; i < ids.size(); ++i) { const DpJob &j = jobs[ids[i]]; fprintf(f, "%d %d %d %d %d %d %d %d %d %d %d %d\n", c, j.qlen, j.tlen, j.w, j.flag, j.zdrop, j.end_bonus, r[i].pad, r[i].max, r[i].max_q, r[i].max_t, r[i].zdropped); }
					fclose(f);
				}
			}
		}
		if (((c >= 2 && c <= 4) || c == 7 || c == 10 || c == 11 || c == 12 || c == 13) && verbose) {
			double sq = 0, stl = 0, sw = 0, zd = 0, mt = 0, ext = 0, big = 0, dg = 0;
			for (size_t i = 0; i < ids.size(); ++i) {
				const DpJob &j = jobs[ids[i]];
				sq += j.qlen, stl += j.tlen, sw += j.w, zd += r[i].zdropped != 0, mt += r[i].max_t + r[i].max_q, ext += (j.flag & 0x40) != 0, big += need[ids[i]] > (8u << 20);
				dg += r[i].pad;
			}
			{
				// what the launch waits for is its longest problem: the tail of the diagonal counts
				int mx = 0; size_t n4k = 0, n10k = 0; for (size_t i = 0; i < ids.size(); ++i) { mx = std::max(mx, (int)r[i].pad); n4k += r[i].pad > 4000; n10k += r[i].pad > 10000; }
				fprintf(stderr, "[pga]       class %d: longest problem %d diagonals; %zu above 4 k, %zu above 10 k\n", c, mx, n4k, n10k);
			}
			const double m = (double)ids.size();
			fprintf(stderr, "[pga]       class %d: mean qlen %.0f tlen %.0f w %.0f; extension-only %.0f%%, z-dropped %.0f%%, mean max_q+max_t %.0f, ~diagonals %.0f, slab > 8 MB: %.0f\n", c, sq / m, stl / m, sw / m,
			        100 * ext / m, 100 * zd / m, mt / m, dg / m, big);
		}
	}
	(void)hipEventDestroy(ready);
	const double t_waited = now();
	const std::vector<unsigned long long> h_cursor = d_cursor.download(st);
	unsigned long long used = h_cursor[0];
	if (verbose) for (int c : {10, 11}) if (!cls[c].empty())
		fprintf(stderr, "[pga]       class %d: %llu of %u direction-matrix chunks taken (%zu problems on %zu workgroups)\n", c, h_cursor[(size_t)1 + c] >> 32, lanes_pool_chunks[c - 10] & 0x7fffffffu, cls[c].size(), waves_of[c]);
	if (used > cig_total) throw std::runtime_error("pga: CIGAR pool overflow");
	if (tm) { tm->kern[K_EXTD2].alg_bytes += 4.0 * (double)used; tm->dp_cigar_ops += (double)used; }
	download_to(cigars, d_pool.p, (size_t)used, st);
	if (verbose) fprintf(stderr, "[pga]     dp_run host: classify %.3f, prepare+launch %.3f, wait+collect %.3f, CIGAR download %.3f s\n", t_begin - t_enter, t_launched - t_begin, t_waited - t_launched, now() - t_waited);
}

} // namespace pga
