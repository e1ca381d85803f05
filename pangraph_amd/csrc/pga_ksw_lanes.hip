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
#include <mutex>
#include "pga_common.h"
#include "pga_dp.h"
#include "pga_wave.h"
#include "pga_pk16.h"
#include <cstdio>
#include <type_traits>

namespace pga {

#define KSW_NEG_INF (-0x40000000)
#define EZ_RIGHT      0x02
#define EZ_APPROX_MAX 0x08
#define EZ_APPROX_DROP 0x10
#define EZ_EXTZ_ONLY  0x40
#define EZ_REV_CIGAR  0x80
#define LBT 64
#define LANES_C 8
#define LANES_CHUNK (2u << 20)     // bytes of a direction-matrix chunk
#define LANES_CHUNK_NARROW (512u << 10)   // ... of the one-wave instantiation (rings of at most 512 columns)
#define LANES_MAXCHUNK 192
__host__ __device__ constexpr uint32_t lanes_chunk_of(int nt) { return nt <= 64 ? LANES_CHUNK_NARROW : LANES_CHUNK; }

__device__ __forceinline__ void diag_range_l(int r, int qlen, int tlen, int w, int &st0, int &en0)
{
	int st = 0, en = tlen - 1;
	if (st < r - qlen + 1) st = r - qlen + 1;
	if (en > r) en = r;
	if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
	if (en > (r + w) >> 1) en = (r + w) >> 1;
	st0 = st, en0 = en;
}

__device__ __forceinline__ int sx8l(int v) { return __builtin_amdgcn_sbfe(v, 0, 8); }
// Packed 16-bit operations as the instructions themselves: written through the vector extensions, min(x, 1) * c and friends are
// canonicalised into per-half compares and selects (three to five instructions where one v_pk_* does it).
__device__ __forceinline__ s2_t k_max(s2_t a, s2_t b) { int r; asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(as_i(a)), "v"(as_i(b))); return as_s2(r); }
__device__ __forceinline__ s2_t k_min(s2_t a, s2_t b) { int r; asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(as_i(a)), "v"(as_i(b))); return as_s2(r); }
__device__ __forceinline__ s2_t k_minu(s2_t a, s2_t b) { int r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(as_i(a)), "v"(as_i(b))); return as_s2(r); }
__device__ __forceinline__ s2_t k_mad(s2_t a, s2_t b, s2_t c) { int r; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(as_i(a)), "v"(as_i(b)), "v"(as_i(c))); return as_s2(r); }
__device__ __forceinline__ s2_t k_shr(s2_t sh, s2_t a) { int r; asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(r) : "v"(as_i(sh)), "v"(as_i(a))); return as_s2(r); }

struct __attribute__((aligned(16))) LaneRec {      // what a wave publishes per diagonal (two copies, by diagonal parity)
	uint32_t key;             // best packed (H, tie order, column) key among its columns
	uint32_t nb;              // x, v, x2 of its last column: what lane 0 of the next wave reads on the next diagonal
	int32_t h7;               // H of its last column
	uint32_t pad;
};

template <int NT>
__global__ __launch_bounds__(NT)
void k_extd2_lanes(const DpJob *__restrict__ jobs, uint32_t n_jobs, PkBases bases, DpParams P,
                   uint32_t *__restrict__ job_counter, uint8_t *__restrict__ slab_all, size_t cig_bytes, uint32_t n_chunks_arg, int q_cap,
                   DpRes *__restrict__ res, uint32_t *__restrict__ cigar_pool, unsigned long long *__restrict__ pool_cursor, unsigned long long pool_cap)
{
	constexpr int NW = NT / 64, C = LANES_C, RC = NT * C;
	constexpr uint32_t CHUNK = lanes_chunk_of(NT);
	extern __shared__ __align__(16) uint8_t qq[];       // the query window, orientation and complement resolved
	__shared__ uint32_t s_job;
	__shared__ LaneRec s_rec[2][NT / 64];
	__shared__ int s_hen[2], s_hst[2], s_h0v[2], s_h0u[2];
	__shared__ uint8_t s_win[LBT * LBT];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// scratch: one CIGAR buffer per workgroup, then the pool of direction-matrix CHUNKS.  A workgroup takes chunks as its diagonals
	// need them (most extensions z-drop after ~1.5 k diagonals and touch 2 MB of a matrix that would reserve 30 MB) and keeps them
	// for its next problems; job_counter[1] is the pool's cursor.  An exhausted pool hands the problem back (n_cigar = -9).
	const uint32_t n_chunks = n_chunks_arg & 0x7fffffffu;
	const bool tiers = !(n_chunks_arg >> 31);
	uint32_t *cig_tmp = (uint32_t*)(slab_all + (size_t)blockIdx.x * cig_bytes);
	uint8_t *pool_base = slab_all + (size_t)gridDim.x * cig_bytes;
	__shared__ uint32_t s_chunk[LANES_MAXCHUNK];
	__shared__ int s_have;
	if (threadIdx.x == 0) s_have = 0;
	int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
	const int qe_h = q + e;
	if (q2 + e2 < q + e) { int t = q; q = q2, q2 = t, t = e, e = e2, e2 = t; }
	const int qe = q + e, qe2 = q2 + e2;
	const int sc_mch = P.sc_mch, sc_mis = P.sc_mis, sc_N = P.sc_ambi == 0 ? -e2 : P.sc_ambi;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const uint32_t sc_tab = (uint32_t)(uint8_t)sc_mch | (uint32_t)(uint8_t)sc_mis << 8 | (uint32_t)(uint8_t)sc_N << 16 | (uint32_t)(uint8_t)sc_N << 24;
	const s2_t ZERO = splat2(0), ONE = splat2(1), FOUR = splat2(4), MCH = splat2(sc_mch << 8), Q1 = splat2(q << 8), Q2 = splat2(q2 << 8), QE = splat2(qe << 8), QE2 = splat2(qe2 << 8);
	const s2_t C8 = splat2(8), C16 = splat2(16), C32 = splat2(32), C64 = splat2(64), C120 = splat2(120), C15 = splat2(15), CM8 = splat2(-8), CM16 = splat2(-16), CM32 = splat2(-32), CM64 = splat2(-64);
	const s2_t INI1 = splat2((-q - e) << 8), INI2 = splat2((-q2 - e2) << 8);
	const uint32_t nb_init = (uint32_t)(uint8_t)(-q - e) | (uint32_t)(uint8_t)(-q - e) << 8 | (uint32_t)(uint8_t)(-q2 - e2) << 16;
	uint8_t *tq = qq + ((q_cap + 15) & ~15);            // the target window: no global load (and so no vmcnt wait behind the direction stores) in the diagonal loop

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
		auto target_at = [&](int i) -> uint32_t { return (i >= 0 && i < tlen) ? (uint32_t)bases.at(t_base + (uint64_t)(J.seq_rev ? tlen - 1 - i : i)) : 0u; };
		auto query_at = [&](int j) -> int {
			int pj = J.qs + (J.seq_rev ? qlen - 1 - j : j);
			if (!J.q_rev) return bases.at(q_base + (uint64_t)(pj));
			int c = bases.at(q_base + (uint64_t)(J.qlen_full - 1 - pj));
			return c < 4 ? 3 - c : 4;
		};
		// the windows, sixteen bases per thread and trip: consecutive j are consecutive store positions, ascending or descending (two word
		// loads + two mask loads per sixteen bases instead of thirty-two loads); bytes 0..3 = ACGT, 4 = anything else
		{
			const bool q_desc = (J.seq_rev != 0) != (J.q_rev != 0);
			const int64_t q_p0 = (int64_t)q_base + (J.q_rev ? (int64_t)J.qlen_full - 1 - J.qs - (J.seq_rev ? qlen - 1 : 0) : (int64_t)J.qs + (J.seq_rev ? qlen - 1 : 0));   // store position of j = 0
			const uint32_t cm = J.q_rev ? 0x03030303u : 0u;
			for (int j0 = 16 * tid; j0 < qlen; j0 += 16 * NT) {
				const int64_t lo = q_desc ? q_p0 - j0 - 15 : q_p0 + j0;
				if (lo < 0) { for (int j = j0; j < j0 + 16 && j < qlen; ++j) qq[j] = (uint8_t)query_at(j); continue; }
				uint32_t w, m; bases.window16((uint64_t)lo, w, m);
				if (q_desc) { w = __brev(w); w = ((w >> 1) & 0x55555555u) | ((w & 0x55555555u) << 1); m = __brev(m) >> 16; }
				uint4 o; uint32_t *op = &o.x;
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					uint32_t x = (w >> (8 * g)) & 0xffu; x = (x | x << 12) & 0x000f000fu; x = (x | x << 6) & 0x03030303u;
					const uint32_t y = (((m >> (4 * g)) & 0xfu) * 0x00204081u) & 0x01010101u;
					op[g] = ((x ^ cm) & ~(y * 3u)) | (y << 2);
				}
				*reinterpret_cast<uint4*>(qq + j0) = o;               // (the window is padded to sixteen: bytes behind qlen are never read as bases)
			}
			const int64_t t_p0 = (int64_t)t_base + (J.seq_rev ? tlen - 1 : 0);
			for (int i0 = 16 * tid; i0 < T + 16; i0 += 16 * NT) {
				uint4 o = make_uint4(0u, 0u, 0u, 0u);
				if (i0 < tlen) {
					const int64_t lo = J.seq_rev ? t_p0 - i0 - 15 : t_p0 + i0;
					if (lo < 0) { uint32_t *op = &o.x; for (int i = i0; i < i0 + 16; ++i) op[(i - i0) >> 2] |= target_at(i) << (8 * ((i - i0) & 3)); }
					else {
						uint32_t w, m; bases.window16((uint64_t)lo, w, m);
						if (J.seq_rev) { w = __brev(w); w = ((w >> 1) & 0x55555555u) | ((w & 0x55555555u) << 1); m = __brev(m) >> 16; }
						const int v = tlen - i0;                          // bases of this word inside the target: behind them the window holds zeros
						if (v < 16) { w &= (1u << (2 * v)) - 1u; m &= (1u << v) - 1u; }
						uint32_t *op = &o.x;
#pragma unroll
						for (int g = 0; g < 4; ++g) {
							uint32_t x = (w >> (8 * g)) & 0xffu; x = (x | x << 12) & 0x000f000fu; x = (x | x << 6) & 0x03030303u;
							const uint32_t y = (((m >> (4 * g)) & 0xfu) * 0x00204081u) & 0x01010101u;
							op[g] = (x & ~(y * 3u)) | (y << 2);
						}
					}
				}
				if (i0 + 16 <= T + 8) *reinterpret_cast<uint4*>(tq + i0) = o;
				else *reinterpret_cast<uint2*>(tq + i0) = make_uint2(o.x, o.y);     // T + 8 is the window's last byte
			}
		}
		const int rpc = CHUNK / n_col;                   // diagonals per chunk
		auto want_chunk = [&](int c) {                        // (thread 0) make sure chunk c of this problem exists
			if (c < LANES_MAXCHUNK && c >= s_have) {
				// the second half of the pool is reserved progressively for the workgroups with the lower indices (they hold the largest
				// problems: the queue is sorted): when the pool runs dry the others give up early instead of everybody late
				const uint32_t limit = tiers ? n_chunks - (uint32_t)((unsigned long long)(n_chunks / 2) * blockIdx.x / gridDim.x) : n_chunks;
				uint32_t id = 0xffffffffu;
				if (c < 2 || __hip_atomic_load(job_counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < limit) {
					id = atomicAdd(job_counter + 1, 1u);
					if (id >= n_chunks) id = 0xffffffffu;
				}
				s_chunk[c] = id;
				if (id != 0xffffffffu) s_have = c + 1;
			}
		};
		if (tid == 0) { want_chunk(0); want_chunk(1); }
		int row_c = 0, row_o = 0;                               // chunk and offset of the current diagonal
		uint8_t *prow = nullptr;                                // its row of direction bytes (advanced behind the store)

		// ---- the lane's block: eight columns in registers ----
		// Every int8 of the reference is held as value << 8 in a 16-bit half (low byte zero): additions and subtractions then wrap
		// exactly like the reference's epi8 arithmetic, signed maxima and minima order the same way, and no sign extension is ever
		// needed.
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
			TB0 = TB1 = 0;
			if (t0 < T) { const uint2 tb = *reinterpret_cast<const uint2*>(tq + t0); TB0 = tb.x, TB1 = tb.y; }
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
		// the length-bound stop (pga_dp.h): an extension towards a block end a few bases away is final after ~3 tlen diagonals
		const LbStop LB = lb_stop_of(qlen, tlen, w, flag, q, e, q2, e2, sc_mch, sc_mis, sc_N, P.lb_mode);
		int lb_hit = 0, lb_max = 0, lb_max_t = 0, lb_max_q = 0, lb_mte = 0, lb_mte_q = 0;      // (checked mode: the record as it was when the bounds closed)
		int r_done = 0;
		uint32_t q_next = tid == 0 && qlen > 0 ? (uint32_t)qq[0] : 0u;       // query[r - t0] of the first diagonal
		auto col8 = [&](const s2_t (&A)[4], int i) -> int { return __builtin_amdgcn_sbfe(as_i(A[i >> 1]), (i & 1) ? 24 : 8, 8); };   // the int8 of column i

		auto sel8 = [&](const int (&A)[8], int i) -> int {             // A[i], i lane-varying: three levels of selects
			const int a0 = (i & 1) ? A[1] : A[0], a1 = (i & 1) ? A[3] : A[2], a2 = (i & 1) ? A[5] : A[4], a3 = (i & 1) ? A[7] : A[6];
			const int b0 = (i & 2) ? a1 : a0, b1 = (i & 2) ? a3 : a2;
			return (i & 4) ? b1 : b0;
		};
		auto col8v = [&](const s2_t (&A)[4], int i) -> int {
			const int a0 = (i & 2) ? as_i(A[1]) : as_i(A[0]), a1 = (i & 2) ? as_i(A[3]) : as_i(A[2]);
			return __builtin_amdgcn_sbfe((i & 4) ? a1 : a0, (i & 1) ? 24 : 8, 8);
		};
		int sat = 0;

		uint4 nrec = make_uint4(0u, nb_init, (uint32_t)KSW_NEG_INF, 0u);   // the left neighbour wave's record of the previous diagonal
#ifdef PGA_LANES_PROF
		// (development: -DPGA_LANES_PROF splits a wave's cycles per diagonal into bookkeeping / cells / maximum / barrier / after-barrier and prints them)
		long long pf[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, pt = 0;
#define PF_MARK(i) { const long long c_ = clock64(); pf[i] += c_ - pt; pt = c_; }
#else
#define PF_MARK(i)
#endif
		auto diag_loop = [&](auto EXACT_T, auto RIGHT_T) {
		constexpr bool EXACT = decltype(EXACT_T)::value, RIGHT = decltype(RIGHT_T)::value;
#ifdef PGA_LANES_PROF
		pt = clock64();
#endif
		for (int r = 0; r < n_diag; ++r) {
			r_done = r + 1;
			int st0, en0;
			diag_range_l(r, qlen, tlen, w, st0, en0);
			if (st0 > en0) { ez_zdropped = 1; break; }
			const int st = st0 & ~15, en = ((en0 + 16) & ~15) - 1;         // (0 <= st0 <= en0: the reference's divisions by 16 are shifts)
			const int span = ((en0 - st0) & ~15) + 16;
			PF_MARK(8)
			if (row_o == 0) {
				if (tid == 0) want_chunk(row_c + 1);                 // one chunk ahead: visible behind this diagonal's barrier
				const uint32_t chunk_id = s_chunk[row_c];
				if (chunk_id == 0xffffffffu) { sat = 1; break; }
				prow = pool_base + (size_t)chunk_id * CHUNK;
			}
			if (++row_o == rpc) row_o = 0, ++row_c;
			PF_MARK(5)
			int need_hi = en > st0 + span - 1 ? en : st0 + span - 1;
			if (need_hi > T - 1) need_hi = T - 1;
			// a lane whose block fell out of the band on the left takes the block one ring further right (fresh rows)
			// (a lane moves on once in NT * 8 diagonals: the whole block hides behind a scalar branch, so that the ~80 register moves of
			// fresh() are not issued -- under an empty exec mask -- on every diagonal)
			const bool moves_on = (blk + NT) * C <= need_hi;
			if (__ballot(moves_on) != 0ULL) {
				if (moves_on) { blk += NT; fresh(r - 1); const int j = r - blk * C; q_next = (j >= 0 && j < qlen) ? (uint32_t)qq[j] : 0u; }
			}
			const int t0 = blk * C;
			LaneRec *recs = s_rec[r & 1];
			// query byte of the block's first column (requested one diagonal ahead: its LDS latency hides behind the barrier)
			W1 = W1 << 8 | W0 >> 24; W0 = W0 << 8 | q_next;
			{ const int j = r + 1 - t0; const uint32_t qb = qq[j < 0 ? 0 : j >= qlen ? qlen - 1 : j]; q_next = (unsigned)j < (unsigned)qlen ? qb : 0u; }
			PF_MARK(6)
			// left neighbour: x, v, x2 and H of column t0-1 as the previous diagonal left them (an int8 is the top byte of its half)
			const uint32_t mine = __builtin_amdgcn_perm((uint32_t)as_i(X2[3]), __builtin_amdgcn_perm((uint32_t)as_i(V[3]), (uint32_t)as_i(X[3]), 0x0c0c0703u), 0x0c070100u);
			uint32_t inc = (uint32_t)wave_shr1((int)mine, (int)nrec.y);
			const int hp_in = wave_shr1(H[7], (int)nrec.z);
			{
				const uint32_t c1 = (uint32_t)(uint8_t)(-q - e), c2 = (uint32_t)(uint8_t)(-q2 - e2);
				const uint32_t v1 = st > 0 ? c1 : (uint32_t)(uint8_t)(r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2);
				const bool fresh_edge = st == 0 || !(st - 1 >= last_st && st - 1 <= last_en);     // (uniform)
				inc = (t0 == st && fresh_edge) ? (c1 | v1 << 8 | c2 << 16) : inc;
			}
			PF_MARK(7)
			// score bytes of the columns in [st0, st0+span) (the others keep what an earlier diagonal left there)
			{
				int lo = st0 - t0, hi = (st0 + span < T ? st0 + span : T) - t0;
				lo = lo < 0 ? 0 : lo > 8 ? 8 : lo; hi = hi < lo ? lo : hi > 8 ? 8 : hi;
				// byte masks of [lo, hi) over the two words
				const int lo0 = lo > 4 ? 4 : lo, hi0 = hi > 4 ? 4 : hi, lo1 = lo - lo0, hi1 = hi - hi0;
				const uint32_t m0 = (hi0 == 4 ? 0xffffffffu : (1u << (8 * hi0)) - 1u) & ~(lo0 == 4 ? 0xffffffffu : (1u << (8 * lo0)) - 1u);
				const uint32_t m1 = (hi1 == 4 ? 0xffffffffu : (1u << (8 * hi1)) - 1u) & ~(lo1 == 4 ? 0xffffffffu : (1u << (8 * lo1)) - 1u);
				const uint32_t nz0 = ((TB0 ^ W0) + 0x7f7f7f7fu) >> 7 & 0x01010101u, nn0 = (TB0 | W0) >> 2 & 0x01010101u;
				const uint32_t nz1 = ((TB1 ^ W1) + 0x7f7f7f7fu) >> 7 & 0x01010101u, nn1 = (TB1 | W1) >> 2 & 0x01010101u;
				S0 = (S0 & ~m0) | (__builtin_amdgcn_perm(0u, sc_tab, nz0 | nn0 << 1) & m0);
				S1 = (S1 & ~m1) | (__builtin_amdgcn_perm(0u, sc_tab, nz1 | nn1 << 1) & m1);
			}
			PF_MARK(0)
			if (t0 >= st && t0 <= en) {
				// the column that joins on this diagonal starts from the first-row values
				if (en >= r && r >= t0 && r < t0 + 8) {
					const int uj = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
#pragma unroll
					for (int p = 0; p < 4; ++p) {
						if (t0 + 2 * p == r) { U[p].x = (short)(uj << 8); Y[p].x = (short)((-q - e) << 8); Y2[p].x = (short)((-q2 - e2) << 8); }
						if (t0 + 2 * p + 1 == r) { U[p].y = (short)(uj << 8); Y[p].y = (short)((-q - e) << 8); Y2[p].y = (short)((-q2 - e2) << 8); }
					}
				}
				// the left neighbour's values ride in the high half of the carry
				int cx = (int)(inc << 24), cv = (int)(inc << 16 & 0xff000000u), cx2 = (int)(inc << 8 & 0xff000000u);
				uint32_t dpk[4];
				// (written column-pair-parallel: the four pairs of a step are independent, and inline asm keeps its program order, so this
				// is the order that hides the 8-cycle latency of dependent packed operations)
				s2_t xt1[4], vt1[4], x2t1[4], z0[4], a[4], b[4], a2[4], b2[4], zm[4], d[4], z[4];
#pragma unroll
				for (int p = 0; p < 4; ++p) {
					const int ox = as_i(X[p]), ov = as_i(V[p]), ox2 = as_i(X2[p]);
					xt1[p] = as_s2((int)__builtin_amdgcn_alignbit((uint32_t)ox, (uint32_t)cx, 16));
					vt1[p] = as_s2((int)__builtin_amdgcn_alignbit((uint32_t)ov, (uint32_t)cv, 16));
					x2t1[p] = as_s2((int)__builtin_amdgcn_alignbit((uint32_t)ox2, (uint32_t)cx2, 16));
					cx = ox, cv = ov, cx2 = ox2;
					z0[p] = as_s2((int)__builtin_amdgcn_perm(0u, p < 2 ? S0 : S1, (p & 1) ? 0x030c020cu : 0x010c000cu));
				}
#pragma unroll
				for (int p = 0; p < 4; ++p) { a[p] = xt1[p] + vt1[p]; b[p] = Y[p] + U[p]; a2[p] = x2t1[p] + vt1[p]; b2[p] = Y2[p] + U[p]; }
				if constexpr (!RIGHT) {
					// the first of z, a, b, a2, b2 that attains the maximum = how many running maxima stay below it
					s2_t p1[4], p2[4], p3[4];
#pragma unroll
					for (int p = 0; p < 4; ++p) p1[p] = k_max(z0[p], a[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) p2[p] = k_max(p1[p], b[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) p3[p] = k_max(p2[p], a2[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) zm[p] = k_max(p3[p], b2[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = k_minu(zm[p] - z0[p], ONE);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = d[p] + k_minu(zm[p] - p1[p], ONE);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = d[p] + k_minu(zm[p] - p2[p], ONE);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = d[p] + k_minu(zm[p] - p3[p], ONE);
				} else {
					// the last one that attains it = how many maxima over a suffix reach it
					s2_t s3[4], s2[4], s1[4];
#pragma unroll
					for (int p = 0; p < 4; ++p) s3[p] = k_max(a2[p], b2[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) s2[p] = k_max(b[p], s3[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) s1[p] = k_max(a[p], s2[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) zm[p] = k_max(z0[p], s1[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = FOUR - k_minu(zm[p] - b2[p], ONE);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = d[p] - k_minu(zm[p] - s3[p], ONE);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = d[p] - k_minu(zm[p] - s2[p], ONE);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = d[p] - k_minu(zm[p] - s1[p], ONE);
				}
#pragma unroll
				for (int p = 0; p < 4; ++p) z[p] = k_min(zm[p], MCH);
#pragma unroll
				for (int p = 0; p < 4; ++p) { const s2_t un = z[p] - vt1[p], vn = z[p] - U[p]; U[p] = un; V[p] = vn; }
#pragma unroll
				for (int p = 0; p < 4; ++p) { const s2_t t1 = z[p] - Q1, t2 = z[p] - Q2; a[p] = a[p] - t1; b[p] = b[p] - t1; a2[p] = a2[p] - t2; b2[p] = b2[p] - t2; }
				if constexpr (!RIGHT) {                  // continuation bits: a > 0
#pragma unroll
					for (int p = 0; p < 4; ++p) { a[p] = k_max(a[p], ZERO); b[p] = k_max(b[p], ZERO); a2[p] = k_max(a2[p], ZERO); b2[p] = k_max(b2[p], ZERO); }
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = k_mad(k_min(a[p], ONE), C8, d[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = k_mad(k_min(b[p], ONE), C16, d[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = k_mad(k_min(a2[p], ONE), C32, d[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = k_mad(k_min(b2[p], ONE), C64, d[p]);
				} else {                                 // a >= 0: all of 0x78 minus the sign bits
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = k_mad(k_shr(C15, a[p]), CM8, d[p] + C120);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = k_mad(k_shr(C15, b[p]), CM16, d[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = k_mad(k_shr(C15, a2[p]), CM32, d[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) d[p] = k_mad(k_shr(C15, b2[p]), CM64, d[p]);
#pragma unroll
					for (int p = 0; p < 4; ++p) { a[p] = k_max(a[p], ZERO); b[p] = k_max(b[p], ZERO); a2[p] = k_max(a2[p], ZERO); b2[p] = k_max(b2[p], ZERO); }
				}
#pragma unroll
				for (int p = 0; p < 4; ++p) { X[p] = a[p] - QE; Y[p] = b[p] - QE; X2[p] = a2[p] - QE2; Y2[p] = b2[p] - QE2; dpk[p] = (uint32_t)as_i(d[p]); }
				uint2 dd;
				dd.x = __builtin_amdgcn_perm(dpk[1], dpk[0], 0x06040200u);
				dd.y = __builtin_amdgcn_perm(dpk[3], dpk[2], 0x06040200u);
				// (streamed past the caches: 70 GB of direction bytes per build are written and only the few on a path ever read back -- kept out
				// of L2 they do not push out what the latency-bound kernels of the other batches in flight live on)
				__builtin_nontemporal_store(((unsigned long long)dd.y << 32) | dd.x, reinterpret_cast<unsigned long long*>(prow + (t0 - st)));
			}
			prow += n_col;
			PF_MARK(1)
			// ---- H (exact mode): H[t] += v[t] over [st0, en0), H[en0] = H[en0-1](old) + u[en0] (ksw2_extd2_sse.c:325-340), and the
			// maximum with the reference's tie order (H[en0] first, then four int32 lanes by (t - st0) & 3 over [st0, en1), then the
			// tail; the first maximum wins) as ONE unsigned key per column:
			//   (clamp16(H) + 32768) << 16 | field << 12 | (4095 - (t - st)),  field = 8 for en0, 7 - class otherwise.
			// A maximum that hits the clamp is reported (the problem is redone by the workgroup kernel).
			uint32_t kbest = 0;
			if constexpr (EXACT) {
				const int lo = st0 - t0, hi = en0 - t0;                      // column i of the block is updated iff lo <= i < hi; i == hi is en0
				if (hi >= 0 && lo < 8) {
					const int e1 = st0 + (en0 - st0) / 4 * 4 - t0;           // class of column i: (i - lo) & 3 below e1, 4 from there on
					const uint32_t lowbase = 4095u - (uint32_t)(t0 - st) + (32768u << 16);
					const uint32_t span_u = (uint32_t)(hi - lo);
					// (the first diagonal and en0 == 0 -- the same diagonal unless the target is one base long -- take H[en0] from elsewhere: a
					// uniform case of its own, so that the eight columns of every other diagonal are straight-line selects: written as one loop
					// with the three-way choice inside, the compiler emitted six scalar branches per column)
					int prev_old = hp_in;
					if (r != 0 && en0 != 0) {
#pragma unroll
						for (int i = 0; i < 8; ++i) {
							const uint32_t rel = (uint32_t)(i - lo);
							const bool in = rel < span_u, is_en = i == hi;
							const int hold = H[i];
							const int hin = hold + col8(V, i), hen = prev_old + col8(U, i);
							const int h = is_en ? hen : in ? hin : hold;
							H[i] = h;
							prev_old = hold;
							const uint32_t field = is_en ? 8u : 7u - (i < e1 ? (rel & 3u) : 4u);
							const int hc = h < -32768 ? -32768 : h > 32767 ? 32767 : h;        // (v_med3_i32)
							const uint32_t key = ((uint32_t)hc << 16) + (lowbase - (uint32_t)i + (field << 12));
							kbest = (in | is_en) && key > kbest ? key : kbest;
						}
					} else {
						const bool r0 = r == 0;
#pragma unroll
						for (int i = 0; i < 8; ++i) {
							const uint32_t rel = (uint32_t)(i - lo);
							const bool in = rel < span_u, is_en = i == hi;
							const int hold = H[i], vn = col8(V, i);
							const int hen = r0 ? vn - qe_h : hold + vn;
							const int h = is_en ? hen : in ? hold + vn : hold;
							H[i] = h;
							prev_old = hold;
							const uint32_t field = is_en ? 8u : 7u - (i < e1 ? (rel & 3u) : 4u);
							const int hc = h < -32768 ? -32768 : h > 32767 ? 32767 : h;
							const uint32_t key = ((uint32_t)hc << 16) + (lowbase - (uint32_t)i + (field << 12));
							kbest = (in | is_en) && key > kbest ? key : kbest;
						}
					}
					// the two values the end-of-sequence scores need, while the diagonal runs along the last target column / query row
					if (en0 == tlen - 1 || r - st0 == qlen - 1) {          // (uniform: only while the diagonal runs along an edge of the matrix)
						if ((unsigned)hi < 8u) s_hen[r & 1] = sel8(H, hi);
						if ((unsigned)lo < 8u) s_hst[r & 1] = sel8(H, lo);
					}
				}
				kbest = wave_max_u32(kbest);
			} else {
				const int h0t = r == 0 ? 0 : last_H0_t;
				if (h0t + 1 >= t0 && h0t < t0 + 8) {
					int hv = 0, hu = 0;
#pragma unroll
					for (int i = 0; i < 8; ++i) {
						const int t = t0 + i;
						hv = t == h0t ? col8(V, i) : hv;
						hu = t == h0t + 1 ? col8(U, i) : hu;
					}
					if (h0t >= t0 && h0t < t0 + 8) s_h0v[r & 1] = hv;
					if (h0t + 1 >= t0 && h0t + 1 < t0 + 8) s_h0u[r & 1] = hu;
				}
			}
			PF_MARK(2)
			{
				// the wave's record: its key, and x, v, x2, H of its last column for lane 0 of the next wave
				const uint32_t mine_new = __builtin_amdgcn_perm((uint32_t)as_i(X2[3]), __builtin_amdgcn_perm((uint32_t)as_i(V[3]), (uint32_t)as_i(X[3]), 0x0c0c0703u), 0x0c070100u);
				const uint32_t p63 = (uint32_t)__builtin_amdgcn_readlane((int)mine_new, 63), h63 = (uint32_t)__builtin_amdgcn_readlane(H[7], 63);
				if (lane == 0) *reinterpret_cast<uint4*>(&recs[wave]) = make_uint4(kbest, p63, h63, 0u);
			}
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
			PF_MARK(3)
			// ---- uniform bookkeeping of the diagonal (every wave, from the records) ----
			bool stop = false;
			uint32_t k = 0;
#pragma unroll
			for (int wv = 0; wv < NW; ++wv) {
				const uint4 rc = *reinterpret_cast<const uint4*>(&recs[wv]);
				k = rc.x > k ? rc.x : k;
				if (wv == (wave + NW - 1) % NW) nrec = rc;
			}
			if constexpr (EXACT) {
				const uint32_t kh16 = k >> 16;
				const int max_H = (int)kh16 - 32768, max_t = st + 4095 - (int)(k & 4095u);
				sat |= (kh16 == 0 || kh16 == 65535u) ? 1 : 0;
				if (en0 == tlen - 1 || r - st0 == qlen - 1) {
					const int he = s_hen[r & 1], hs = s_hst[r & 1];
					if (en0 == tlen - 1) { if (he > ez_mte) ez_mte = he, ez_mte_q = r - en0; if (r == n_diag - 1) ez_score = he; }
					if (r - st0 == qlen - 1 && hs > ez_mqe) ez_mqe = hs, ez_mqe_t = st0;
				}
				// (straight-line: a branch costs more than all of this)
				const bool upd = max_H > ez_max;
				const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
				stop = !upd & (tl >= 0) & (ql >= 0) & (zdrop >= 0) & (ez_max - max_H > zdrop + l * e2);
				ez_max_t = upd ? max_t : ez_max_t; ez_max_q = upd ? r - max_t : ez_max_q; ez_max = upd ? max_H : ez_max;
				if (stop) ez_zdropped = 1, ez_score = KSW_NEG_INF;
				else if (LB.on && (r & 7) == 7 && !lb_hit && lb_final(LB, r, tlen, q, e, q2, e2, sc_mch, ez_max < ez_mte ? ez_max : ez_mte)) {
					if (P.lb_mode == 2) lb_hit = r + 1, lb_max = ez_max, lb_max_t = ez_max_t, lb_max_q = ez_max_q, lb_mte = ez_mte, lb_mte_q = ez_mte_q;
					else ez_zdropped = 1, stop = true;
				}
			} else {
				const int h0v = s_h0v[r & 1], h0u = s_h0u[r & 1];
				if (r > 0) {
					if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
						if (h0v > h0u) H0 += h0v; else H0 += h0u, ++last_H0_t;
					} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += h0v;
					else ++last_H0_t, H0 += h0u;
				} else H0 = h0v - qe_h, last_H0_t = 0;
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
			PF_MARK(4)
		}
		};
		if (!approx_max) { if (right) diag_loop(std::true_type{}, std::true_type{}); else diag_loop(std::true_type{}, std::false_type{}); }
		else { if (right) diag_loop(std::false_type{}, std::true_type{}); else diag_loop(std::false_type{}, std::false_type{}); }


#ifdef PGA_LANES_PROF
		if (lane == 0 && jid == 0) printf("[lanes prof] wave %d: %d diagonals; cycles per diagonal: ranges %lld chunk %lld ring+window %lld neighbour %lld score bytes %lld | cells %lld maximum %lld record+barrier %lld after %lld\n", wave, r_done,
		                                  pf[8] / r_done, pf[5] / r_done, pf[6] / r_done, pf[7] / r_done, pf[0] / r_done, pf[1] / r_done, pf[2] / r_done, pf[3] / r_done, pf[4] / r_done);
#endif
		// ---- backtrack by wave 0 (ksw2.h:127-159) through a 64x64 LDS window of the direction matrix ----
		int n_cigar = 0, bi = -1, bj = -1;
		const bool lb_broken = lb_hit && !sat && !(ez_zdropped && ez_max == lb_max && ez_max_t == lb_max_t && ez_max_q == lb_max_q && ez_mte == lb_mte && ez_mte_q == lb_mte_q &&
		                                            ez_mqe == KSW_NEG_INF && ez_score == KSW_NEG_INF);
		if (sat) {}                                                 // (no traceback: the problem is redone)
		else if (!ez_zdropped && !(flag & EZ_EXTZ_ONLY)) bi = tlen - 1, bj = qlen - 1;
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
							if (st0 <= en0 && col >= off && col <= off_end) val = pool_base[(size_t)s_chunk[r / rpc] * CHUNK + (size_t)(r % rpc) * n_col + (col - off)];
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
				R.score = ez_score, R.zdropped = ez_zdropped, R.reach_end = ez_reach_end, R.n_cigar = lb_broken ? -11 : sat ? -9 : n_cigar, R.pad = r_done, R.cigar_off = base;
				res[jid] = R;
			}
		}
	}
}

// a problem the lane kernel takes: its band ring fits the NT*8 columns of an NT-thread workgroup (256: end extensions and banded fills, many per
// CU; 64: the same for rings of at most 512 columns -- extensions towards a block end a few hundred bases away, thousands per round in the
// upper levels of a build: one wave, no workgroup barrier, four times as many in flight) and its windows fit LDS
bool lanes_eligible(const DpJob &j, int nt)
{
	const size_t chunk = lanes_chunk_of(nt);
	if (j.flag & PGA_JOB_LL) return false;
	if (j.qlen < 1 || j.tlen < 1 || j.qlen > 28 * 1024 || j.tlen > 28 * 1024) return false;
	const int T = (j.tlen + 15) / 16 * 16;
	const int w = j.w < 0 ? (j.tlen > j.qlen ? j.tlen : j.qlen) : j.w;
	int R = ((w < j.tlen ? w : j.tlen) + 15) / 16 * 16 + 96;
	if (R > T) R = T;
	if (R > nt * LANES_C) return false;
	int n_col = j.qlen < j.tlen ? j.qlen : j.tlen;
	n_col = (((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1) * 16;
	return ((size_t)j.qlen + j.tlen) / (chunk / (size_t)n_col) + 2 <= LANES_MAXCHUNK;
}

size_t lanes_cig_bytes(int q_cap, int t_cap) { return (4 * ((size_t)q_cap + t_cap + 8) + 255) & ~(size_t)255; }
size_t lanes_chunk_bytes(int nt) { return lanes_chunk_of(nt); }

template <int NT> static void launch_lanes_nt(unsigned n_blocks, size_t lds, hipStream_t st, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter, uint8_t *slab,
                                              size_t cig_bytes, uint32_t n_chunks, int q_cap, DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap)
{
	{	// a per-DEVICE function attribute, set once per device whatever thread comes first
		static std::mutex mu; static bool attr_set[64] = {};
		int dev = 0; PGA_HIP(hipGetDevice(&dev));
		std::lock_guard<std::mutex> lk(mu);
		if (!attr_set[dev & 63]) { PGA_HIP(hipFuncSetAttribute((const void*)k_extd2_lanes<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); attr_set[dev & 63] = true; }
	}
	hipLaunchKernelGGL(k_extd2_lanes<NT>, dim3(n_blocks), dim3(NT), lds, st, jobs, n_jobs, bases, P, counter, slab, cig_bytes, n_chunks, q_cap, res, pool, cursor, pool_cap);
}

void launch_extd2_lanes(int nt, unsigned n_blocks, int q_cap, int t_cap, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter, uint8_t *slab, uint32_t n_chunks,
                        DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st)
{
	const size_t lds = (((size_t)q_cap + 15) & ~(size_t)15) + (((size_t)t_cap + 15) & ~(size_t)15) + 16;
	if (nt <= 64) launch_lanes_nt<64>(n_blocks, lds, st, jobs, n_jobs, bases, P, counter, slab, lanes_cig_bytes(q_cap, t_cap), n_chunks, q_cap, res, pool, cursor, pool_cap);
	else launch_lanes_nt<256>(n_blocks, lds, st, jobs, n_jobs, bases, P, counter, slab, lanes_cig_bytes(q_cap, t_cap), n_chunks, q_cap, res, pool, cursor, pool_cap);
}

} // namespace pga
