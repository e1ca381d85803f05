// pga_ksw_band.hip -- kernel #5e: first-pass gap fills (ksw_extd2_sse with KSW_EZ_APPROX_MAX and an unbinding band,
// C/ksw2_extd2_sse.c:34-401, >95 % of all DP problems) computed inside a 32-column CORRIDOR around the main diagonal,
// with a proof per problem that the full matrix would have given the same answer.
//
// Why this is exact.  Let S be the best score of an alignment that stays inside the corridor (offsets t-j within
// [c-31, c+31], c ~ (tlen-qlen)/2).  An alignment that leaves the corridor on the low side contains at least I = 32-c
// inserted and D = I + (tlen-qlen) deleted bases, so it scores at most  a*(qlen-I) - g(I) - g(D)  (g = the cheaper of the
// two affine gap functions; splitting a gap only costs more); symmetrically on the high side.  If both bounds are
// STRICTLY below S, every alignment that leaves the corridor is strictly worse than the optimum: the optimal path lies
// inside, and so does every path that ties with any prefix of it (a tying excursion would give a tying global alignment).
// Hence along the optimal path H, the gap states it uses, the first-maximum direction choice (ksw2_extd2_sse.c:225-232) and
// the gap-extension bits (:240-247) are the same in the corridor as in the full matrix, and so are score and CIGAR.
// Problems that fail the test (long indels, low-complexity sequence) are flagged and go through the full kernel.
//
// Layout: absolute scores (int32), not ksw2's difference encoding -- cells outside the corridor are simply -inf.
// One wavefront handles TWO problems, 32 lanes each; lane l of a group owns column st(r)+l of anti-diagonal r, where the
// corridor start st(r) advances on every even r.  With that schedule a cell's diagonal predecessor (r-2, t-1) is always
// the lane's own older value, and exactly one DPP lane shift per diagonal brings either the left neighbour's (H,E,E2)
// or the right neighbour's (H,F,F2).  Both sequences of both problems sit in LDS (4 KB per wave); direction bytes go to
// the wave's slab slice (32 coalesced bytes per problem and diagonal) and come back through 64-row LDS windows for the
// backtrack, which the two groups walk side by side on lanes 0 and 32.
#include "pga_common.h"
#include "pga_dp.h"
#include "pga_wave.h"

namespace pga {

#define BAND_NEG (-(1 << 28))
#define BAND_MAXLEN 1024           // longest query / target taken (LDS sequence buffers)
#define BAND_ROWS 64               // backtrack window: diagonals per refill
#define BAND_MAXCIG 248            // CIGAR operations kept in LDS; a problem with more goes through the full kernel
#define EZ_APPROX_MAX 0x08

// lane l <- lane l+1 (wave_shl:1); the last lane keeps `last`
__device__ __forceinline__ int32_t wave_shl1(int32_t v, int32_t last) { return __builtin_amdgcn_update_dpp(last, v, 0x130, 0xf, 0xf, false); }

__device__ __forceinline__ int band_gap(int n, int q, int e, int q2, int e2) { if (n <= 0) return 0; const int a = q + e * n, b = q2 + e2 * n; return a < b ? a : b; }

__global__ __launch_bounds__(64)
void k_gapfill_band(const DpJob *__restrict__ jobs, uint32_t n_jobs, PkBases bases, DpParams P,
                    uint32_t *__restrict__ job_counter, uint8_t *__restrict__ slab_all, size_t slab_bytes,
                    DpRes *__restrict__ res, uint32_t *__restrict__ cigar_pool, unsigned long long *__restrict__ pool_cursor, unsigned long long pool_cap)
{
	__shared__ uint8_t s_t[2][BAND_MAXLEN], s_q[2][BAND_MAXLEN];
	__shared__ __align__(16) uint8_t s_win[2][BAND_ROWS * 32];
	__shared__ uint32_t s_cig[2][BAND_MAXCIG + 8];
	const int lane = threadIdx.x, g = lane >> 5, gl = lane & 31;
	uint8_t *slab = slab_all + (size_t)blockIdx.x * slab_bytes + (size_t)g * (slab_bytes / 2);
	const int q = P.q, e = P.e, q2 = P.q2, e2 = P.e2;
	const int sc_mch = P.sc_mch, sc_mis = P.sc_mis, sc_N = P.sc_ambi == 0 ? -e2 : P.sc_ambi;

	for (;;) {
		uint32_t j0 = 0;
		if (lane == 0) j0 = atomicAdd(job_counter, 2u);
		j0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)j0);
		if (j0 >= n_jobs) break;
		const uint32_t jid = j0 + (uint32_t)g;
		const bool on = jid < n_jobs;                          // the second group may be empty at the very end
		DpJob J; memset(&J, 0, sizeof(J));
		if (on) J = jobs[jid];
		const int qlen = on ? J.qlen : 0, tlen = on ? J.tlen : 0;
		// sequences into LDS (orientation resolved here)
		{
			const uint64_t t_base = J.t_off, q_base = J.q_off;          // base positions in the packed store
			for (int i = gl; i < tlen; i += 32) s_t[g][i] = bases.at(t_base + (uint64_t)(J.seq_rev ? tlen - 1 - i : i));
			for (int jx = gl; jx < qlen; jx += 32) {
				const int pj = J.qs + (J.seq_rev ? qlen - 1 - jx : jx);
				int c;
				if (!J.q_rev) c = bases.at(q_base + (uint64_t)(pj)); else { c = bases.at(q_base + (uint64_t)(J.qlen_full - 1 - pj)); c = c < 4 ? 3 - c : 4; }
				s_q[g][jx] = (uint8_t)c;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		const int delta = tlen - qlen;
		const int c = (delta >> 1) & ~1;                       // corridor centre (even, so that both groups shift on the same diagonals)
		const int n_diag = on ? qlen + tlen - 1 : 0;
		const int n_diag_max = max(__builtin_amdgcn_readlane(n_diag, 0), __builtin_amdgcn_readlane(n_diag, 32));
		// state of the cell this lane computed on the previous diagonal / two diagonals ago
		int H1 = BAND_NEG, E1 = BAND_NEG, F1 = BAND_NEG, E21 = BAND_NEG, F21 = BAND_NEG, H2 = BAND_NEG;
		int score = BAND_NEG;
		// (flag is exactly KSW_EZ_APPROX_MAX: without KSW_EZ_APPROX_DROP the greedy H0 walk of ksw2_extd2_sse.c:367-381 only
		// delivers the corner score, which it reaches exactly; there is no Z-drop test to reproduce)
		for (int r = 0; r < n_diag_max; ++r) {
			const int st = (r + c - 30) >> 1;
			const int t = st + gl, jq = r - t;
			const bool shift = (r & 1) == 0 && r > 0;            // st advanced by one since the last diagonal
			// neighbours on diagonal r-1: column t-1 (H, E, E2) and column t (H, F, F2)
			int Hl, El, E2l, Hu, Fu, F2u;
			if (shift) {
				// column t was lane+1's, column t-1 this lane's own
				Hl = H1, El = E1, E2l = E21;
				Hu = wave_shl1(H1, BAND_NEG); Fu = wave_shl1(F1, BAND_NEG); F2u = wave_shl1(F21, BAND_NEG);
				if (gl == 31) Hu = Fu = F2u = BAND_NEG;             // beyond the corridor (lane 31 would read the other group)
			} else {
				Hu = H1, Fu = F1, F2u = F21;
				Hl = wave_shr1(H1, BAND_NEG); El = wave_shr1(E1, BAND_NEG); E2l = wave_shr1(E21, BAND_NEG);
				if (gl == 0) Hl = El = E2l = BAND_NEG;
			}
			int Hd = H2;                                         // (r-2, t-1): always this lane's own
			// matrix borders (global alignment: H(-1,-1) = 0, first row / column pay the cheaper affine gap, no gap state there)
			// (a lane sits on a border only early on: column 0 needs st(r) <= 0, row 0 needs r - st(r) < 32, i.e. r < 40 for |c| <= 6)
			if (r < 48) {
				if (t == 0) { Hl = -band_gap(jq + 1, q, e, q2, e2); El = E2l = BAND_NEG; Hd = -band_gap(jq, q, e, q2, e2); }
				if (jq == 0) { Hu = -band_gap(t + 1, q, e, q2, e2); Fu = F2u = BAND_NEG; Hd = -band_gap(t, q, e, q2, e2); }
			}
			const bool act = on && r < n_diag && t >= 0 && t < tlen && jq >= 0 && jq < qlen;
			int h = BAND_NEG, E = BAND_NEG, F = BAND_NEG, E2 = BAND_NEG, F2 = BAND_NEG, d = 0;
			if (act) {
				const int a0 = s_t[g][t], b0 = s_q[g][jq];
				int sc = a0 == b0 ? sc_mch : sc_mis;
				if (a0 == 4 || b0 == 4) sc = sc_N;
				// gap states entering this cell
				E  = max(El,  Hl - q)  - e;  F  = max(Fu,  Hu - q)  - e;
				E2 = max(E2l, Hl - q2) - e2; F2 = max(F2u, Hu - q2) - e2;
				h = Hd + sc;
				if (E  > h) d = 1, h = E;                          // first maximum wins (ksw2_extd2_sse.c:225-232)
				if (F  > h) d = 2, h = F;
				if (E2 > h) d = 3, h = E2;
				if (F2 > h) d = 4, h = F2;
				if (E  > h - q)  d |= 0x08;                        // the gap would rather extend than reopen (:240-247)
				if (F  > h - q)  d |= 0x10;
				if (E2 > h - q2) d |= 0x20;
				if (F2 > h - q2) d |= 0x40;
				slab[(size_t)r * 32 + gl] = (uint8_t)d;
				if (t == tlen - 1 && jq == qlen - 1) score = h;
			}
			// age the state: what was r-1 becomes r-2 for the cell this lane computes next (same lane by construction)
			H2 = H1;
			H1 = h; E1 = E; F1 = F; E21 = E2; F21 = F2;        // (all -inf when the cell lies outside the matrix)
		}
		// score of the end cell, known to one lane of the group
		{
			const unsigned long long m = __ballot(score > BAND_NEG);
			const unsigned mg = (unsigned)(g ? m >> 32 : m & 0xffffffffULL);
			int src = mg ? (int)(__ffs(mg) - 1) + 32 * g : lane;
			score = __shfl(score, src);
		}
		// ---- the proof obligation: everything outside the corridor is strictly worse ----
		bool ok = on && score > BAND_NEG;
		if (ok) {
			const int i_lo = 32 - c, d_lo = i_lo + delta, d_hi = c + 32, i_hi = d_hi - delta;
			const int u_lo = (i_lo <= qlen && d_lo <= tlen && d_lo >= 0) ? sc_mch * (qlen - i_lo) - band_gap(i_lo, q, e, q2, e2) - band_gap(d_lo, q, e, q2, e2) : BAND_NEG;
			const int u_hi = (d_hi <= tlen && i_hi <= qlen && i_hi >= 0) ? sc_mch * (tlen - d_hi) - band_gap(i_hi, q, e, q2, e2) - band_gap(d_hi, q, e, q2, e2) : BAND_NEG;
			if (i_lo <= 0 || d_hi <= 0 || u_lo >= score || u_hi >= score) ok = false;
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // direction bytes written by the other lanes of the group
		// ---- backtrack (ksw2.h:127-159), both groups at once, each with its 32 lanes ----
		int n_cigar = 0;
		{
			int i = ok ? tlen - 1 : -1, j = ok ? qlen - 1 : -1, state = 0;
			uint32_t last_op = 0xffffffffu;
			uint32_t *cig = s_cig[g];
			long long guard = 0;
			for (;;) {
				const bool walking = i >= 0 && j >= 0;
				if (!__ballot(walking)) break;
				if (++guard > 4096) { n_cigar = -7; break; }
				const int r_hi = i + j;                            // window: diagonals r_hi-63 .. r_hi
				if (walking) {
					// 64 diagonals x 32 bytes are contiguous in the slab: four 16-byte loads per lane, rows stored top-down
#pragma unroll
					for (int it = 0; it < BAND_ROWS * 32 / 16 / 32; ++it) {
						const int idx = it * 32 + gl, r = r_hi - (BAND_ROWS - 1) + (idx >> 1);
						uint4 v = make_uint4(0, 0, 0, 0);
						if (r >= 0) v = *reinterpret_cast<const uint4*>(slab + (size_t)r * 32 + (idx & 1) * 16);
						*reinterpret_cast<uint4*>(&s_win[g][(r_hi - r) * 32 + (idx & 1) * 16]) = v;
					}
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				// the 32 lanes of a group walk together (their i, j, state agree); a straight stretch of the diagonal -- the same corridor
				// column on every other diagonal, direction 0 in every cell -- is taken up to 32 cells at a time
				while (i >= 0 && j >= 0) {
					const int r = i + j, row = r_hi - r;
					if (row >= BAND_ROWS) break;
					const int col = i - ((r + c - 30) >> 1);
					if (state == 0) {
						const int rk = row + 2 * gl;
						const bool inw = rk < BAND_ROWS && i - gl >= 0 && j - gl >= 0 && col >= 0 && col < 32;
						const uint32_t tk = inw ? (uint32_t)s_win[g][rk * 32 + col] : 0xffu;
						const unsigned long long okm = __ballot((tk & 7) == 0);
						const unsigned ok32 = (unsigned)(g ? okm >> 32 : okm & 0xffffffffULL);
						const int run = ok32 == 0xffffffffu ? 32 : __ffs((int)~ok32) - 1;
						if (run > 0) {
							if (0u != last_op) { if (n_cigar >= BAND_MAXCIG) { n_cigar = -9; i = j = -1; break; } if (gl == 0) cig[n_cigar] = (uint32_t)run << 4; ++n_cigar; last_op = 0; }
							else if (gl == 0) cig[n_cigar - 1] += (uint32_t)run << 4;
							i -= run; j -= run;
							continue;
						}
					}
					const uint32_t tmp = (col >= 0 && col < 32) ? s_win[g][row * 32 + col] : 0u;
					if (state == 0) state = tmp & 7;
					else if (!(tmp >> (state + 2) & 1)) state = 0;
					if (state == 0) state = tmp & 7;
					uint32_t op;
					if (state == 0) op = 0, --i, --j;
					else if (state == 1 || state == 3) op = 2, --i;
					else op = 1, --j;
					if (op != last_op) { if (n_cigar >= BAND_MAXCIG) { n_cigar = -9; i = j = -1; break; } if (gl == 0) cig[n_cigar] = 1u << 4 | op; ++n_cigar; last_op = op; }
					else if (gl == 0) cig[n_cigar - 1] += 1u << 4;
				}
				i = __shfl(i, 32 * g); j = __shfl(j, 32 * g);
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			}
			if (gl == 0 && ok && n_cigar >= 0 && n_cigar < BAND_MAXCIG) {
				if (i >= 0) { if (2u != last_op) { cig[n_cigar] = (uint32_t)(i + 1) << 4 | 2u; ++n_cigar; last_op = 2; } else cig[n_cigar - 1] += (uint32_t)(i + 1) << 4; }
				if (j >= 0) { if (1u != last_op) { cig[n_cigar] = (uint32_t)(j + 1) << 4 | 1u; ++n_cigar; last_op = 1; } else cig[n_cigar - 1] += (uint32_t)(j + 1) << 4; }
			}
			n_cigar = __shfl(n_cigar, 32 * g);
			if (n_cigar < 0) ok = false;
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		unsigned long long base = 0;
		if (gl == 0 && ok && n_cigar > 0) base = atomicAdd(pool_cursor, (unsigned long long)n_cigar);
		base = ((unsigned long long)(unsigned)__shfl((int)(base >> 32), 32 * g) << 32) | (unsigned)__shfl((int)(base & 0xffffffffULL), 32 * g);
		if (ok && n_cigar > 0 && base + (unsigned long long)n_cigar <= pool_cap)
			for (int k = gl; k < n_cigar; k += 32) cigar_pool[base + k] = s_cig[g][n_cigar - 1 - k];
		if (gl == 0 && on) {
			DpRes R;
			R.max = 0, R.max_q = -1, R.max_t = -1, R.mqe = -0x40000000, R.mqe_t = -1, R.mte = -0x40000000, R.mte_q = -1;
			R.score = ok ? score : -0x40000000; R.zdropped = 0, R.reach_end = 0;
			R.n_cigar = ok ? n_cigar : -9;                       // -9: not proven inside the corridor, run the full matrix
			R.pad = 0, R.cigar_off = base;
			res[jid] = R;
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	}
}

size_t band_slab_bytes(int max_diag) { return ((size_t)2 * max_diag * 32 + 255) & ~(size_t)255; }

void launch_gapfill_band(unsigned n_waves, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter, uint8_t *slab, size_t slab_bytes,
                         DpRes *res, uint32_t *pool, unsigned long long *cursor, unsigned long long pool_cap, hipStream_t st)
{
	hipLaunchKernelGGL(k_gapfill_band, dim3(n_waves), dim3(64), 0, st, jobs, n_jobs, bases, P, counter, slab, slab_bytes, res, pool, cursor, pool_cap);
}

} // namespace pga
