// pga_ksw_pipe.hip -- kernel #5j: a BANDED exact problem of ksw_extd2_sse (C/ksw2_extd2_sse.c:34-401: the end extensions of mm_align1 -- band
// 1.5 * bw, z-drop, KSW_EZ_EXTZ_ONLY -- and banded fills) as a PIPELINE OF WAVES INSIDE ONE WORKGROUP.
//
// Why.  117 k such problems per step of the BASELINE build; nine out of ten are extensions into unrelated sequence that the reference sweeps
// for ~1 540 diagonals (the diagonal's maximum sits in a pure gap cell at the edge of the matrix, where ksw2.h:178 cannot fire, until the band cuts
// the edges off at r > w).  While r <= w the range is [0, r]: a TRIANGLE.  The lane kernel (pga_ksw_lanes.hip: four waves, eight columns per lane,
// one workgroup barrier per diagonal) runs every wave through every diagonal -- 1 250 instructions per wave and diagonal, of which the eight
// columns of a lane are half, whether or not the band has reached the wave -- 2.2 us per diagonal, 3.4 ms per such extension, 8 M wave
// instructions for 1.2 M cells.  The wave strips (pga_ksw_bstrips.hip) take the barrier away but hand columns over through device memory
// and hold a whole direction matrix per problem.  Here
//   * EIGHT waves of 64 lanes x FOUR columns (two packed pairs: the lane kernel's arithmetic, an int8 of the reference is value << 8 in a
//     16-bit half) hold a ring of 2 048 columns; a ninth wave is the problem's EVALUATOR;
//   * there is NO barrier in the sweep: wave k runs a block of sixteen diagonals when wave k - 1 has finished it, and takes the state of that
//     wave's last column (x, v, x2, H per diagonal) from an LDS ring; a wave the band has not reached yet costs nothing, a wave the band
//     has left is gone -- the instruction count of a triangle falls sevenfold and a diagonal costs what ONE wave's four columns cost;
//   * the per-diagonal maximum with the reference's tie order is one packed key per wave and diagonal in an LDS ring; the evaluator combines the
//     keys of the waves the range touches and takes the reference's decisions (ksw2_extd2_sse.c:326-366, ksw2.h:167-184) sixteen diagonals at
//     a time, behind the slowest wave; waves run at most 96 diagonals ahead of it (ring depth 128);
//   * direction bytes go to pooled 2 MB chunks as in the lane kernel; the evaluator walks the path back.
//   * the band slides over the waves as a RING: a wave whose 256 columns the band has left takes the block 2 048 columns further right (fresh
//     rows, as the reference's untouched arrays hold them); the band with its sixteen-rounding fits 1 792 columns, so a wave never holds two
//     blocks the band needs at once.
// A clamped maximum hands the problem back to the workgroup kernel (n_cigar = -9), a dry chunk pool to the lane kernel (-10).  Semantics (16-lane rounding of the ranges, stale rows outside them, the
// first-row and fresh-edge rules, int8 wrap-around) are the lane kernel's; parity: tests/test_gpu_parity.py (PGA_PIPE=force / off).
#include <mutex>
#include "pga_common.h"
#include "pga_dp.h"
#include "pga_wave.h"
#include "pga_pk16.h"
#include <cstdio>
#include <cstring>
#include <type_traits>

namespace pga {

#define KSW_NEG_INF (-0x40000000)
#define EZ_RIGHT      0x02
#define EZ_APPROX_MAX 0x08
#define EZ_EXTZ_ONLY  0x40
#define EZ_REV_CIGAR  0x80
#define PP_NW 8
#define PP_C 4
#define PP_WCOLS (64 * PP_C)
#define PP_RING (PP_NW * PP_WCOLS)
#define PP_D 128
#define PP_BLK 16
#define PP_NT ((PP_NW + 1) * 64)
#define PP_CHUNK (2u << 20)
#define PP_MAXCHUNK 192
#define PP_BT 64

__device__ __forceinline__ void pp_range(int r, int qlen, int tlen, int w, int &st0, int &en0)
{
	int st = 0, en = tlen - 1;
	if (st < r - qlen + 1) st = r - qlen + 1;
	if (en > r) en = r;
	if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
	if (en > (r + w) >> 1) en = (r + w) >> 1;
	st0 = st, en0 = en;
}
// the last column a diagonal touches: the sixteen-rounded range or the score bytes it refreshes, whichever reaches further (capped at the padded target)
__device__ __forceinline__ int pp_hi(int st0, int en0, int T)
{
	const int en = ((en0 + 16) & ~15) - 1, sp = st0 + ((en0 - st0) & ~15) + 15;
	int h = en > sp ? en : sp;
	return h > T - 1 ? T - 1 : h;
}
__device__ __forceinline__ s2_t pk_max(s2_t a, s2_t b) { int r; asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(as_i(a)), "v"(as_i(b))); return as_s2(r); }
__device__ __forceinline__ s2_t pk_min(s2_t a, s2_t b) { int r; asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(as_i(a)), "v"(as_i(b))); return as_s2(r); }
__device__ __forceinline__ s2_t pk_minu(s2_t a, s2_t b) { int r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(as_i(a)), "v"(as_i(b))); return as_s2(r); }
__device__ __forceinline__ s2_t pk_mad(s2_t a, s2_t b, s2_t c) { int r; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(as_i(a)), "v"(as_i(b)), "v"(as_i(c))); return as_s2(r); }
__device__ __forceinline__ s2_t pk_shr(s2_t sh, s2_t a) { int r; asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(r) : "v"(as_i(sh)), "v"(as_i(a))); return as_s2(r); }

__device__ __forceinline__ int pp_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void pp_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

__global__ __launch_bounds__(PP_NT, 5)
void k_ext_pipe(const DpJob *__restrict__ jobs, uint32_t n_jobs, PkBases bases, DpParams P,
                uint32_t *__restrict__ job_counter, uint8_t *__restrict__ slab_all, size_t cig_bytes, uint32_t n_chunks, int q_cap,
                DpRes *__restrict__ res, uint32_t *__restrict__ cigar_pool, unsigned long long *__restrict__ pool_cursor, unsigned long long pool_cap)
{
	extern __shared__ __align__(16) uint8_t qq[];       // the query window, orientation and complement resolved; behind it the target window
	__shared__ uint32_t s_job;
	__shared__ unsigned long long s_mail[PP_NW][PP_D];  // after diagonal r (slot r % D): x, v, x2 of the wave's last column (low word), its H (high word)
	__shared__ uint32_t s_key[PP_NW][PP_D];             // the wave's best packed key of diagonal r
	__shared__ int s_hen[PP_D], s_hst[PP_D];            // H[en0], H[st0] of diagonal r (while it runs along an edge of the matrix)
	__shared__ int s_prog[PP_NW];                       // wave k has finished the diagonals [0, s_prog[k])
	__shared__ int s_eval, s_stop, s_nlim, s_kind, s_dry;
	__shared__ uint32_t s_chunk[PP_MAXCHUNK];
	__shared__ int s_have;
	__shared__ uint8_t s_win[PP_BT * PP_BT];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint32_t *cig_tmp = (uint32_t*)(slab_all + (size_t)blockIdx.x * cig_bytes);
	uint8_t *pool_base = slab_all + (size_t)gridDim.x * cig_bytes;
	if (tid == 0) s_have = 0;
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
	uint8_t *tq = qq + ((q_cap + 15) & ~15);

	for (;;) {
		__syncthreads();
		if (tid == 0) s_job = atomicAdd(job_counter, 1u);
		__syncthreads();
		const uint32_t jid = s_job;
		if (jid >= n_jobs) break;
		const DpJob J = jobs[jid];
		const uint64_t t_base = J.t_off, q_base = J.q_off;
		const int qlen = J.qlen, tlen = J.tlen, flag = J.flag, zdrop = J.zdrop, end_bonus = J.end_bonus;
		const bool right = flag & EZ_RIGHT;
		int w = J.w;
		if (w < 0) w = tlen > qlen ? tlen : qlen;
		const int T = (tlen + 15) / 16 * 16;
		int n_col = qlen < tlen ? qlen : tlen;
		n_col = (((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1) * 16;
		const int n_diag = qlen + tlen - 1;
		const int rpc = PP_CHUNK / n_col;
		auto target_at = [&](int i) -> uint32_t { return (i >= 0 && i < tlen) ? (uint32_t)bases.at(t_base + (uint64_t)(J.seq_rev ? tlen - 1 - i : i)) : 0u; };
		auto query_at = [&](int j) -> int {
			int pj = J.qs + (J.seq_rev ? qlen - 1 - j : j);
			if (!J.q_rev) return bases.at(q_base + (uint64_t)(pj));
			int c = bases.at(q_base + (uint64_t)(J.qlen_full - 1 - pj));
			return c < 4 ? 3 - c : 4;
		};
		// the windows, sixteen bases per thread and trip (the lane kernel's staging): bytes 0..3 = ACGT, 4 = anything else
		{
			const bool q_desc = (J.seq_rev != 0) != (J.q_rev != 0);
			const int64_t q_p0 = (int64_t)q_base + (J.q_rev ? (int64_t)J.qlen_full - 1 - J.qs - (J.seq_rev ? qlen - 1 : 0) : (int64_t)J.qs + (J.seq_rev ? qlen - 1 : 0));
			const uint32_t cm = J.q_rev ? 0x03030303u : 0u;
			for (int j0 = 16 * tid; j0 < qlen; j0 += 16 * PP_NT) {
				const int64_t lo = q_desc ? q_p0 - j0 - 15 : q_p0 + j0;
				if (lo < 0) { for (int j = j0; j < j0 + 16 && j < qlen; ++j) qq[j] = (uint8_t)query_at(j); continue; }
				uint32_t ww, m; bases.window16((uint64_t)lo, ww, m);
				if (q_desc) { ww = __brev(ww); ww = ((ww >> 1) & 0x55555555u) | ((ww & 0x55555555u) << 1); m = __brev(m) >> 16; }
				uint4 o; uint32_t *op = &o.x;
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					uint32_t x = (ww >> (8 * g)) & 0xffu; x = (x | x << 12) & 0x000f000fu; x = (x | x << 6) & 0x03030303u;
					const uint32_t y = (((m >> (4 * g)) & 0xfu) * 0x00204081u) & 0x01010101u;
					op[g] = ((x ^ cm) & ~(y * 3u)) | (y << 2);
				}
				*reinterpret_cast<uint4*>(qq + j0) = o;
			}
			const int64_t t_p0 = (int64_t)t_base + (J.seq_rev ? tlen - 1 : 0);
			for (int i0 = 16 * tid; i0 < T + 16; i0 += 16 * PP_NT) {
				uint4 o = make_uint4(0u, 0u, 0u, 0u);
				if (i0 < tlen) {
					const int64_t lo = J.seq_rev ? t_p0 - i0 - 15 : t_p0 + i0;
					if (lo < 0) { uint32_t *op = &o.x; for (int i = i0; i < i0 + 16; ++i) op[(i - i0) >> 2] |= target_at(i) << (8 * ((i - i0) & 3)); }
					else {
						uint32_t ww, m; bases.window16((uint64_t)lo, ww, m);
						if (J.seq_rev) { ww = __brev(ww); ww = ((ww >> 1) & 0x55555555u) | ((ww & 0x55555555u) << 1); m = __brev(m) >> 16; }
						const int v = tlen - i0;
						if (v < 16) { ww &= (1u << (2 * v)) - 1u; m &= (1u << v) - 1u; }
						uint32_t *op = &o.x;
#pragma unroll
						for (int g = 0; g < 4; ++g) {
							uint32_t x = (ww >> (8 * g)) & 0xffu; x = (x | x << 12) & 0x000f000fu; x = (x | x << 6) & 0x03030303u;
							const uint32_t y = (((m >> (4 * g)) & 0xfu) * 0x00204081u) & 0x01010101u;
							op[g] = (x & ~(y * 3u)) | (y << 2);
						}
					}
				}
				if (i0 + 16 <= T + 8) *reinterpret_cast<uint4*>(tq + i0) = o;
				else *reinterpret_cast<uint2*>(tq + i0) = make_uint2(o.x, o.y);
			}
		}
		if (tid == 0) {
			// the first diagonal whose range is empty (the reference ends there, zdropped; monotone in r), else the end of the matrix
			int lo = 0, hi = n_diag;
			while (lo < hi) {
				const int mid = (lo + hi) >> 1; int a0, a1; pp_range(mid, qlen, tlen, w, a0, a1);
				if (a0 > a1) hi = mid; else lo = mid + 1;
			}
			s_nlim = lo; s_kind = lo < n_diag ? 1 : 0;
			// the first direction chunk (>= 1 024 diagonals; the evaluator takes more as the sweep goes on; a workgroup keeps its chunks for its next problems)
			int dry = 0;
			if (s_have < 1) { const uint32_t id = atomicAdd(job_counter + 1, 1u); if (id >= n_chunks) dry = 1; else { s_chunk[0] = id; s_have = 1; } }
			s_dry = dry; s_eval = 0; s_stop = 0;
		}
		if (tid < PP_NW) s_prog[tid] = 0;
		__syncthreads();
		const int n_lim = s_nlim;
		const bool dry = s_dry != 0;

		// the evaluator's record (wave PP_NW)
		int ez_max = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1, ez_mte = KSW_NEG_INF, ez_mte_q = -1;
		int ez_score = KSW_NEG_INF, ez_zdropped = 0, ez_reach_end = 0, r_done = 0, sat = dry ? 2 : 0;      // sat: 1 = a clamped maximum (-> the workgroup kernel), 2 = handed back to the lane kernel

		if (!dry && wave < PP_NW) {
			// ================================================= a compute wave: columns [t0, t0 + 4) per lane, fixed =================================================
			const int k = wave, left = (k + PP_NW - 1) % PP_NW;
			int col_lo = k * PP_WCOLS, t0 = col_lo + lane * PP_C;               // the wave's block of the ring (moves on by PP_RING columns when the band has left it)
			s2_t X[2], V[2], X2[2], U[2], Y[2], Y2[2];
			int H[4];
			uint32_t S = 0, TB = 0, W = 0, q_next = 0;
			bool started = false;
			auto take_block = [&]() {                                          // rows as the reference's freshly allocated arrays hold them
				t0 = col_lo + lane * PP_C;
#pragma unroll
				for (int p = 0; p < 2; ++p) { X[p] = V[p] = U[p] = Y[p] = INI1; X2[p] = Y2[p] = INI2; }
#pragma unroll
				for (int i = 0; i < 4; ++i) H[i] = KSW_NEG_INF;
				S = 0; TB = 0; started = false;
				if (t0 < T) TB = *reinterpret_cast<const uint32_t*>(tq + t0);
			};
			take_block();
			int last_st = -1, last_en = -1;
			auto col8 = [&](const s2_t (&A)[2], int i) -> int { return __builtin_amdgcn_sbfe(as_i(A[i >> 1]), (i & 1) ? 24 : 8, 8); };
			auto sel4 = [&](const int (&A)[4], int i) -> int { const int a0 = (i & 1) ? A[1] : A[0], a1 = (i & 1) ? A[3] : A[2]; return (i & 2) ? a1 : a0; };

#ifdef PP_PROF
			long long pp_c[4] = {0, 0, 0, 0}, pp_t = 0, pp_n = 0;
#define PP_T(i) { const long long c_ = clock64(); pp_c[i] += c_ - pp_t; pp_t = c_; }
#else
#define PP_T(i)
#endif
			auto sweep = [&](auto RIGHT_T) {
			constexpr bool RIGHT = decltype(RIGHT_T)::value;
			for (int r0 = 0; r0 < n_lim; r0 += PP_BLK) {
				const int r1 = r0 + PP_BLK - 1 < n_lim - 1 ? r0 + PP_BLK - 1 : n_lim - 1;
				if (pp_ld(&s_stop)) break;
				// has the band reached this wave?  has it left it?  (ranges only move to the right)
				int a0, a1, b0, b1;
				pp_range(r0, qlen, tlen, w, b0, b1);
				bool gone = false;
				while ((b0 & ~15) > col_lo + PP_WCOLS - 1) {                          // the band has left the block: the next turn of the ring, if the target goes that far
					if (col_lo + PP_RING > T - 1) { gone = true; break; }
					col_lo += PP_RING; take_block();
				}
				if (gone) break;                                                     // (progress is raised behind the loop)
				pp_range(r1, qlen, tlen, w, a0, a1);
				if (pp_hi(a0, a1, T) < col_lo) { if (lane == 0) pp_st(&s_prog[k], r1 + 1); continue; }
				// the wave on the left has finished these diagonals; the evaluator (and with it the wave on the right) has read the ring slots they reuse;
				// the rows' chunk exists
				PP_T(0)
				if (col_lo > 0) while (pp_ld(&s_prog[left]) < r1) { if (pp_ld(&s_stop)) break; __builtin_amdgcn_s_sleep(1); }
				PP_T(1)
				while (pp_ld(&s_eval) < r1 - (PP_D - 2 * PP_BLK) || pp_ld(&s_have) <= r1 / rpc) { if (pp_ld(&s_stop)) break; __builtin_amdgcn_s_sleep(2); }
				PP_T(2)
				if (pp_ld(&s_stop)) break;
				if (!started) {
					// window bytes i = query[(r0 - 1) - i - t0]: what the shift at the top of diagonal r0 expects; the base that enters there
					started = true; W = 0;
#pragma unroll
					for (int i = 0; i < 4; ++i) { const int j = r0 - 1 - i - t0; W |= ((j >= 0 && j < qlen) ? (uint32_t)qq[j] : 0u) << (8 * i); }
					{ const int j = r0 - t0; q_next = (j >= 0 && j < qlen) ? (uint32_t)qq[j] : 0u; }
					if (r0 > 0) { int c0, c1; pp_range(r0 - 1, qlen, tlen, w, c0, c1); last_st = c0 & ~15, last_en = ((c1 + 16) & ~15) - 1; }
				}
				// the left wave's last column after the diagonals r0 - 1 ... r1 - 1, a diagonal per lane
				unsigned long long inw = (unsigned long long)(uint32_t)KSW_NEG_INF << 32 | nb_init;
				if (col_lo > 0 && lane < PP_BLK && r0 - 1 + lane >= 0) inw = s_mail[left][(r0 - 1 + lane) & (PP_D - 1)];
				uint8_t *prow = pool_base + (size_t)s_chunk[r0 / rpc] * PP_CHUNK + (size_t)(r0 % rpc) * n_col;
				int row_o = r0 % rpc;
				for (int r = r0; r <= r1; ++r) {
					const int di = r - r0;
					int st0, en0;
					pp_range(r, qlen, tlen, w, st0, en0);
					const int st = st0 & ~15, en = ((en0 + 16) & ~15) - 1;
					const int span = ((en0 - st0) & ~15) + 16;
					W = W << 8 | q_next;
					{ const int j = r + 1 - t0; const uint32_t qb = qq[j < 0 ? 0 : j >= qlen ? qlen - 1 : j]; q_next = (unsigned)j < (unsigned)qlen ? qb : 0u; }
					// left neighbour: x, v, x2 and H of column t0 - 1 as the previous diagonal left them
					const uint32_t mine = __builtin_amdgcn_perm((uint32_t)as_i(X2[1]), __builtin_amdgcn_perm((uint32_t)as_i(V[1]), (uint32_t)as_i(X[1]), 0x0c0c0703u), 0x0c070100u);
					uint32_t inc = (uint32_t)wave_shr1((int)mine, __builtin_amdgcn_readlane((int)(uint32_t)inw, di));
					const int hp_in = wave_shr1(H[3], __builtin_amdgcn_readlane((int)(uint32_t)(inw >> 32), di));
					{
						const uint32_t c1 = (uint32_t)(uint8_t)(-q - e), c2 = (uint32_t)(uint8_t)(-q2 - e2);
						const uint32_t v1 = st > 0 ? c1 : (uint32_t)(uint8_t)(r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2);
						const bool fresh_edge = st == 0 || !(st - 1 >= last_st && st - 1 <= last_en);
						inc = (t0 == st && fresh_edge) ? (c1 | v1 << 8 | c2 << 16) : inc;
					}
					// score bytes of the columns in [st0, st0 + span)
					{
						int lo = st0 - t0, hi = (st0 + span < T ? st0 + span : T) - t0;
						lo = lo < 0 ? 0 : lo > 4 ? 4 : lo; hi = hi < lo ? lo : hi > 4 ? 4 : hi;
						const uint32_t m = (hi == 4 ? 0xffffffffu : (1u << (8 * hi)) - 1u) & ~(lo == 4 ? 0xffffffffu : (1u << (8 * lo)) - 1u);
						const uint32_t nz = ((TB ^ W) + 0x7f7f7f7fu) >> 7 & 0x01010101u, nn = (TB | W) >> 2 & 0x01010101u;
						S = (S & ~m) | (__builtin_amdgcn_perm(0u, sc_tab, nz | nn << 1) & m);
					}
					if (t0 >= st && t0 <= en) {
						if (en >= r && r >= t0 && r < t0 + 4) {
							const int uj = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
#pragma unroll
							for (int p = 0; p < 2; ++p) {
								if (t0 + 2 * p == r) { U[p].x = (short)(uj << 8); Y[p].x = (short)((-q - e) << 8); Y2[p].x = (short)((-q2 - e2) << 8); }
								if (t0 + 2 * p + 1 == r) { U[p].y = (short)(uj << 8); Y[p].y = (short)((-q - e) << 8); Y2[p].y = (short)((-q2 - e2) << 8); }
							}
						}
						int cx = (int)(inc << 24), cv = (int)(inc << 16 & 0xff000000u), cx2 = (int)(inc << 8 & 0xff000000u);
						uint32_t dpk[2];
						s2_t xt1[2], vt1[2], x2t1[2], z0[2], a[2], b[2], a2[2], b2[2], zm[2], d[2], z[2];
#pragma unroll
						for (int p = 0; p < 2; ++p) {
							const int ox = as_i(X[p]), ov = as_i(V[p]), ox2 = as_i(X2[p]);
							xt1[p] = as_s2((int)__builtin_amdgcn_alignbit((uint32_t)ox, (uint32_t)cx, 16));
							vt1[p] = as_s2((int)__builtin_amdgcn_alignbit((uint32_t)ov, (uint32_t)cv, 16));
							x2t1[p] = as_s2((int)__builtin_amdgcn_alignbit((uint32_t)ox2, (uint32_t)cx2, 16));
							cx = ox, cv = ov, cx2 = ox2;
							z0[p] = as_s2((int)__builtin_amdgcn_perm(0u, S, p ? 0x030c020cu : 0x010c000cu));
						}
#pragma unroll
						for (int p = 0; p < 2; ++p) { a[p] = xt1[p] + vt1[p]; b[p] = Y[p] + U[p]; a2[p] = x2t1[p] + vt1[p]; b2[p] = Y2[p] + U[p]; }
						if constexpr (!RIGHT) {
							s2_t p1[2], p2[2], p3[2];
#pragma unroll
							for (int p = 0; p < 2; ++p) p1[p] = pk_max(z0[p], a[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) p2[p] = pk_max(p1[p], b[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) p3[p] = pk_max(p2[p], a2[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) zm[p] = pk_max(p3[p], b2[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = pk_minu(zm[p] - z0[p], ONE);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = d[p] + pk_minu(zm[p] - p1[p], ONE);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = d[p] + pk_minu(zm[p] - p2[p], ONE);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = d[p] + pk_minu(zm[p] - p3[p], ONE);
						} else {
							s2_t s3[2], s2[2], s1[2];
#pragma unroll
							for (int p = 0; p < 2; ++p) s3[p] = pk_max(a2[p], b2[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) s2[p] = pk_max(b[p], s3[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) s1[p] = pk_max(a[p], s2[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) zm[p] = pk_max(z0[p], s1[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = FOUR - pk_minu(zm[p] - b2[p], ONE);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = d[p] - pk_minu(zm[p] - s3[p], ONE);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = d[p] - pk_minu(zm[p] - s2[p], ONE);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = d[p] - pk_minu(zm[p] - s1[p], ONE);
						}
#pragma unroll
						for (int p = 0; p < 2; ++p) z[p] = pk_min(zm[p], MCH);
#pragma unroll
						for (int p = 0; p < 2; ++p) { const s2_t un = z[p] - vt1[p], vn = z[p] - U[p]; U[p] = un; V[p] = vn; }
#pragma unroll
						for (int p = 0; p < 2; ++p) { const s2_t t1 = z[p] - Q1, t2 = z[p] - Q2; a[p] = a[p] - t1; b[p] = b[p] - t1; a2[p] = a2[p] - t2; b2[p] = b2[p] - t2; }
						if constexpr (!RIGHT) {
#pragma unroll
							for (int p = 0; p < 2; ++p) { a[p] = pk_max(a[p], ZERO); b[p] = pk_max(b[p], ZERO); a2[p] = pk_max(a2[p], ZERO); b2[p] = pk_max(b2[p], ZERO); }
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = pk_mad(pk_min(a[p], ONE), C8, d[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = pk_mad(pk_min(b[p], ONE), C16, d[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = pk_mad(pk_min(a2[p], ONE), C32, d[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = pk_mad(pk_min(b2[p], ONE), C64, d[p]);
						} else {
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = pk_mad(pk_shr(C15, a[p]), CM8, d[p] + C120);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = pk_mad(pk_shr(C15, b[p]), CM16, d[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = pk_mad(pk_shr(C15, a2[p]), CM32, d[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) d[p] = pk_mad(pk_shr(C15, b2[p]), CM64, d[p]);
#pragma unroll
							for (int p = 0; p < 2; ++p) { a[p] = pk_max(a[p], ZERO); b[p] = pk_max(b[p], ZERO); a2[p] = pk_max(a2[p], ZERO); b2[p] = pk_max(b2[p], ZERO); }
						}
#pragma unroll
						for (int p = 0; p < 2; ++p) { X[p] = a[p] - QE; Y[p] = b[p] - QE; X2[p] = a2[p] - QE2; Y2[p] = b2[p] - QE2; dpk[p] = (uint32_t)as_i(d[p]); }
						__builtin_nontemporal_store(__builtin_amdgcn_perm(dpk[1], dpk[0], 0x06040200u), reinterpret_cast<uint32_t*>(prow + (t0 - st)));
					}
					// ---- H and the per-column key (the lane kernel's formulation, four columns) ----
					uint32_t kbest = 0;
					{
						const int lo = st0 - t0, hi = en0 - t0;
						if (hi >= 0 && lo < 4) {
							const int e1 = st0 + (en0 - st0) / 4 * 4 - t0;
							const uint32_t lowbase = 4095u - (uint32_t)(t0 - st) + (32768u << 16);
							const uint32_t span_u = (uint32_t)(hi - lo);
							int prev_old = hp_in;
							if (r != 0 && en0 != 0) {
#pragma unroll
								for (int i = 0; i < 4; ++i) {
									const uint32_t rel = (uint32_t)(i - lo);
									const bool in = rel < span_u, is_en = i == hi;
									const int hold = H[i];
									const int hin = hold + col8(V, i), hen = prev_old + col8(U, i);
									const int h = is_en ? hen : in ? hin : hold;
									H[i] = h;
									prev_old = hold;
									const uint32_t field = is_en ? 8u : 7u - (i < e1 ? (rel & 3u) : 4u);
									const int hc = h < -32768 ? -32768 : h > 32767 ? 32767 : h;
									const uint32_t key = ((uint32_t)hc << 16) + (lowbase - (uint32_t)i + (field << 12));
									kbest = (in | is_en) && key > kbest ? key : kbest;
								}
							} else {
								const bool rz = r == 0;
#pragma unroll
								for (int i = 0; i < 4; ++i) {
									const uint32_t rel = (uint32_t)(i - lo);
									const bool in = rel < span_u, is_en = i == hi;
									const int hold = H[i], vn = col8(V, i);
									const int hen = rz ? vn - qe_h : hold + vn;
									const int h = is_en ? hen : in ? hold + vn : hold;
									H[i] = h;
									prev_old = hold;
									const uint32_t field = is_en ? 8u : 7u - (i < e1 ? (rel & 3u) : 4u);
									const int hc = h < -32768 ? -32768 : h > 32767 ? 32767 : h;
									const uint32_t key = ((uint32_t)hc << 16) + (lowbase - (uint32_t)i + (field << 12));
									kbest = (in | is_en) && key > kbest ? key : kbest;
								}
							}
							if (en0 == tlen - 1 || r - st0 == qlen - 1) {
								if ((unsigned)hi < 4u) s_hen[r & (PP_D - 1)] = sel4(H, hi);
								if ((unsigned)lo < 4u) s_hst[r & (PP_D - 1)] = sel4(H, lo);
							}
						}
					}
					kbest = wave_max_u32(kbest);
					{
						const uint32_t mine_new = __builtin_amdgcn_perm((uint32_t)as_i(X2[1]), __builtin_amdgcn_perm((uint32_t)as_i(V[1]), (uint32_t)as_i(X[1]), 0x0c0c0703u), 0x0c070100u);
						if (lane == 63) s_mail[k][r & (PP_D - 1)] = (unsigned long long)(uint32_t)H[3] << 32 | mine_new;
						if (lane == 0) s_key[k][r & (PP_D - 1)] = kbest;
					}
					prow += n_col;
					if (++row_o == rpc) { row_o = 0; if (r + 1 <= r1) prow = pool_base + (size_t)s_chunk[(r + 1) / rpc] * PP_CHUNK; }
					last_st = st, last_en = en;
				}
				if (lane == 0) pp_st(&s_prog[k], r1 + 1);
				PP_T(3)
#ifdef PP_PROF
				pp_n += r1 - r0 + 1;
#endif
			}
			};
#ifdef PP_PROF
			pp_t = clock64();
#endif
			if (right) sweep(std::true_type{}); else sweep(std::false_type{});
			if (lane == 0) pp_st(&s_prog[k], 0x7fffffff);
#ifdef PP_PROF
			if (lane == 0 && jid == 0 && pp_n) printf("[pipe prof] wave %d: %lld active diagonals; cycles per active diagonal: skip/range %lld, wait left %lld, wait eval+chunk %lld, sweep %lld\n", k, pp_n, pp_c[0] / pp_n, pp_c[1] / pp_n, pp_c[2] / pp_n, pp_c[3] / pp_n);
#endif
		} else if (!dry) {
			// ================================================= the evaluator =================================================
			const LbStop LB = lb_stop_of(qlen, tlen, w, flag, q, e, q2, e2, sc_mch, sc_mis, sc_N, P.lb_mode == 2 ? 0 : P.lb_mode);
			bool halt = false;
			for (int r0 = 0; r0 < n_lim && !halt; r0 += PP_BLK) {
				const int r1 = r0 + PP_BLK - 1 < n_lim - 1 ? r0 + PP_BLK - 1 : n_lim - 1;
				// direction chunks for the diagonals the waves may reach once this block is evaluated
				{
					int want = (r1 + 1 + PP_D + PP_BLK) / rpc; const int last = (n_lim - 1) / rpc; if (want > last) want = last;
					while (pp_ld(&s_have) <= want && !(sat & 2)) {
						uint32_t id = 0xffffffffu;
						if (pp_ld(&s_have) < PP_MAXCHUNK) { if (lane == 0) id = atomicAdd(job_counter + 1, 1u); id = (uint32_t)__builtin_amdgcn_readfirstlane((int)id); }
						if (id >= n_chunks) sat |= 2;
						else { const int h = pp_ld(&s_have); if (lane == 0) { s_chunk[h] = id; pp_st(&s_have, h + 1); } __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
					}
					if (sat & 2) { halt = true; break; }
				}
				// every wave is past the block
				for (;;) {
					const int p = lane < PP_NW ? pp_ld(&s_prog[lane]) : 0x7fffffff;
					if (wave_min_i32(p) > r1) break;
					__builtin_amdgcn_s_sleep(2);
				}
				// the keys of the block: lane = 16 * g + i holds diagonal r0 + i of the waves g and g + 4; a wave counts where the range touches its columns
				const int i = lane & 15, g = lane >> 4, r = r0 + i;
				int st0 = 0, en0 = -1;
				if (r <= r1) pp_range(r, qlen, tlen, w, st0, en0);
				uint32_t kk = 0;
				if (r <= r1) {
#pragma unroll
					for (int h = 0; h < 2; ++h) {
						// the block of wave wv that the range can touch: the first turn of the ring whose last column is not left of st0
						const int wv = g + 4 * h, behind = st0 - (wv * PP_WCOLS + PP_WCOLS - 1);
						const int lo = wv * PP_WCOLS + (behind > 0 ? (behind + PP_RING - 1) / PP_RING * PP_RING : 0);
						if (en0 >= lo) { const uint32_t o = s_key[wv][r & (PP_D - 1)]; kk = o > kk ? o : kk; }
					}
				}
				{ const uint32_t o = (uint32_t)__shfl_xor((int)kk, 16); kk = o > kk ? o : kk; }
				{ const uint32_t o = (uint32_t)__shfl_xor((int)kk, 32); kk = o > kk ? o : kk; }
				int he = KSW_NEG_INF, hs = KSW_NEG_INF;
				if (r <= r1 && (en0 == tlen - 1 || r - st0 == qlen - 1)) { he = s_hen[r & (PP_D - 1)]; hs = s_hst[r & (PP_D - 1)]; }
				const int st_l = st0 & ~15;
				for (int ii = 0; ii <= r1 - r0; ++ii) {
					const int rr = r0 + ii;
					const uint32_t kq = (uint32_t)__builtin_amdgcn_readlane((int)kk, ii);
					const int s0 = __builtin_amdgcn_readlane(st0, ii), e0 = __builtin_amdgcn_readlane(en0, ii), stq = __builtin_amdgcn_readlane(st_l, ii);
					const uint32_t kh16 = kq >> 16;
					const int max_H = (int)kh16 - 32768, max_t = stq + 4095 - (int)(kq & 4095u);
					sat |= (kh16 == 0 || kh16 == 65535u) ? 1 : 0;
					r_done = rr + 1;
					if (e0 == tlen - 1 || rr - s0 == qlen - 1) {
						const int hev = __builtin_amdgcn_readlane(he, ii), hsv = __builtin_amdgcn_readlane(hs, ii);
						if (e0 == tlen - 1) { if (hev > ez_mte) ez_mte = hev, ez_mte_q = rr - e0; if (rr == n_diag - 1) ez_score = hev; }
						if (rr - s0 == qlen - 1 && hsv > ez_mqe) ez_mqe = hsv, ez_mqe_t = s0;
					}
					const bool upd = max_H > ez_max;
					const int tl = max_t - ez_max_t, ql = (rr - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
					const bool stop = !upd & (tl >= 0) & (ql >= 0) & (zdrop >= 0) & (ez_max - max_H > zdrop + l * e2);
					ez_max_t = upd ? max_t : ez_max_t; ez_max_q = upd ? rr - max_t : ez_max_q; ez_max = upd ? max_H : ez_max;
					if (stop) { ez_zdropped = 1, ez_score = KSW_NEG_INF; halt = true; break; }
					if (sat) { halt = true; break; }
					if (LB.on && (rr & 7) == 7 && lb_final(LB, rr, tlen, q, e, q2, e2, sc_mch, ez_max < ez_mte ? ez_max : ez_mte)) { ez_zdropped = 1; halt = true; break; }
				}
				if (lane == 0) pp_st(&s_eval, r1 + 1);
			}
			if (!halt && n_lim < n_diag) { ez_zdropped = 1; r_done = n_lim + 1; }       // the range ran empty (ksw2_extd2_sse.c:172)
			if (lane == 0) pp_st(&s_stop, 1);
		}
		__threadfence_block();
		__syncthreads();
		// ---- the evaluator walks the path back (ksw2.h:127-159) through a 64 x 64 LDS window of the direction matrix ----
		if (wave == PP_NW) {
			int n_cigar = 0, bi = -1, bj = -1;
			if (sat) {}
			else if (!ez_zdropped && !(flag & EZ_EXTZ_ONLY)) bi = tlen - 1, bj = qlen - 1;
			else if (!ez_zdropped && (flag & EZ_EXTZ_ONLY) && ez_mqe + end_bonus > ez_max) ez_reach_end = 1, bi = ez_mqe_t, bj = qlen - 1;
			else if (ez_max_t >= 0 && ez_max_q >= 0) bi = ez_max_t, bj = ez_max_q;
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
				const int r_hi = i + j, c_lo = i - (PP_BT - 1);
				for (int part = 0; part < PP_BT; part += 8) {            // (eight loads in flight: the walk is a rare, short path and must not set the kernel's register count)
					uint8_t wv[8];
#pragma unroll
					for (int rw = 0; rw < 8; ++rw) {
						const int r = r_hi - (part + rw), col = c_lo + lane;
						uint8_t val = 0;
						if (r >= 0 && col >= 0) {
							int st0, en0; pp_range(r, qlen, tlen, w, st0, en0);
							const int off = st0 / 16 * 16, off_end = (en0 + 16) / 16 * 16 - 1;
							if (st0 <= en0 && col >= off && col <= off_end) val = pool_base[(size_t)s_chunk[r / rpc] * PP_CHUNK + (size_t)(r % rpc) * n_col + (col - off)];
						}
						wv[rw] = val;
					}
#pragma unroll
					for (int rw = 0; rw < 8; ++rw) s_win[(part + rw) * PP_BT + lane] = wv[rw];
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				while (i >= 0 && j >= 0) {
					const int r = i + j, row = r_hi - r;
					if (row >= PP_BT || i < c_lo) break;
					int st0, en0; pp_range(r, qlen, tlen, w, st0, en0);
					const int off = st0 / 16 * 16, off_end = (en0 + 16) / 16 * 16 - 1;
					int force_state = -1;
					if (i < off) force_state = 2;
					if (i > off_end) force_state = 1;
					const uint32_t tmp = force_state < 0 ? s_win[row * PP_BT + (i - c_lo)] : 0;
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
				R.score = ez_score, R.zdropped = ez_zdropped, R.reach_end = ez_reach_end, R.n_cigar = (sat & 1) ? -9 : sat ? -10 : n_cigar, R.pad = r_done, R.cigar_off = base;
				res[jid] = R;
			}
		}
	}
}

// a problem this kernel takes: exact maximum, its band ring fits the 2 048 columns (the lane kernel's test), its windows fit LDS
bool pipe_eligible(const DpJob &j)
{
	if (j.flag & (PGA_JOB_LL | EZ_APPROX_MAX)) return false;
	if (j.qlen < 1 || j.tlen < 1 || j.qlen > 28 * 1024 || j.tlen > 28 * 1024) return false;
	const int T = (j.tlen + 15) / 16 * 16;
	const int w = j.w < 0 ? (j.tlen > j.qlen ? j.tlen : j.qlen) : j.w;
	int R = ((w < j.tlen ? w : j.tlen) + 15) / 16 * 16 + 96;
	if (R > T) R = T;
	if (R > PP_RING - PP_WCOLS) return false;           // (a wave moves on as a whole: the band and its rounding leave room for one block)
	int n_col = j.qlen < j.tlen ? j.qlen : j.tlen;
	n_col = (((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1) * 16;
	return (size_t)n_col * 1024 <= PP_CHUNK && ((size_t)j.qlen + j.tlen) / (PP_CHUNK / (size_t)n_col) + 2 <= PP_MAXCHUNK;       // (a chunk holds at least 1 024 diagonals)
}
int pipe_mode()                // PGA_PIPE=off: the lane kernel keeps its problems; force: every eligible problem whatever the launch holds (tests); default: on
{
	const char *e = getenv("PGA_PIPE");
	return !e ? 1 : !strcmp(e, "off") || !strcmp(e, "0") ? 0 : !strcmp(e, "force") ? 2 : 1;
}
size_t pipe_cig_bytes(int q_cap, int t_cap) { return (4 * ((size_t)q_cap + t_cap + 8) + 255) & ~(size_t)255; }
size_t pipe_chunk_bytes() { return PP_CHUNK; }
int pipe_max_chunks() { return PP_MAXCHUNK; }

void launch_ext_pipe(unsigned n_blocks, int q_cap, int t_cap, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter, uint8_t *slab, uint32_t n_chunks,
                     DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st)
{
	const size_t lds = (((size_t)q_cap + 15) & ~(size_t)15) + (((size_t)t_cap + 15) & ~(size_t)15) + 16;
	{
		static std::mutex mu; static bool attr_set[64] = {};
		int dev = 0; PGA_HIP(hipGetDevice(&dev));
		std::lock_guard<std::mutex> lk(mu);
		if (!attr_set[dev & 63]) { PGA_HIP(hipFuncSetAttribute((const void*)k_ext_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); attr_set[dev & 63] = true; }
	}
	hipLaunchKernelGGL(k_ext_pipe, dim3(n_blocks), dim3(PP_NT), lds, st, jobs, n_jobs, bases, P, counter, slab, pipe_cig_bytes(q_cap, t_cap), n_chunks, q_cap, res, pool, cursor, pool_cap);
}

} // namespace pga
