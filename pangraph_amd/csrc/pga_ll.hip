// pga_ll.hip -- kernel #5c: score and end point of the best LOCAL alignment (single affine gap), the test behind
// minimap2's inversion detection: ksw_ll_i16 (ksw2_ll_sse.c:69-151) as called from mm_test_zdrop (align.c:78-86) and
// mm_align1_inv (align.c:851-858) on windows of up to max_gap x max_gap bases.
//
// The reference sweeps the target row by row over a query striped across 8 int16 lanes (Farrar), with the lazy-F loop.
// Its H matrix is the plain Smith-Waterman matrix (a horizontal gap directly followed by a vertical one costs the same
// as the two in the other order, so computing E from the uncorrected H loses nothing) over the query PADDED to a
// multiple of 8 with zero-score columns; what is specific to it and reproduced here:
//   * te is the LAST target row whose maximum equals the global maximum (ksw2_ll_sse.c:139: >=);
//   * qe is read from that row in the memory order of the striped layout, last hit wins (ksw2_ll_sse.c:145-150):
//     column j sits at memory index (j % slen) * 8 + j / slen.
// One 1024-thread workgroup per problem.  Thread t owns R = ceil(tlen/1024) CONSECUTIVE target rows and sweeps the query
// column by column, one step behind thread t-1: at step s it computes the cells (i, j = s - t) of its rows top to bottom with
// H(i, j-1) and F(i, j-1) of its rows in registers.  What a column needs from the row above the block -- H and E of that row
// in the same column -- is what thread t-1 produced one step earlier: one DPP lane shift inside a wave, a double-buffered LDS
// slot across waves, one barrier per step.  A cell costs ~16 VALU operations and no LDS access (the anti-diagonal version this
// replaces read five int16 values per cell from LDS and took 2.3 us per diagonal of a 10 kb x 10 kb problem; this one 1.0 us per
// column step).  Scores are int32 here; they never leave the int16 range for the sizes accepted (the launcher checks
// a * qlen < 32000), so the reference's saturating arithmetic is the plain one.
#include <mutex>
#include "pga_common.h"
#include "pga_dp.h"
#include "pga_wave.h"

namespace pga {

#define LL_NT 1024          // a problem has a CU to itself: the waves hide each other's latencies
#define LL_RMAX 10          // rows per thread: PGA_LL_MAX_LEN / LL_NT

__global__ __launch_bounds__(LL_NT)
void k_ll_i16(const DpJob *__restrict__ jobs, uint32_t n_jobs, PkBases bases, DpParams P, uint32_t *__restrict__ job_counter,
              unsigned long long *__restrict__ rowkey_all, size_t rowkey_stride, int t_cap, DpRes *__restrict__ res)
{
	extern __shared__ __align__(16) uint8_t dyn[];            // the query, padded to a multiple of 8 columns (t_cap bytes)
	__shared__ uint32_t s_job;
	__shared__ unsigned long long s_part[LL_NT / 64];
	__shared__ int s_xh[2][LL_NT / 64], s_xe[2][LL_NT / 64];  // hand-off across wave boundaries, by step parity
	(void)rowkey_all; (void)rowkey_stride;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint8_t *qb = dyn;
	const int gapoe = P.q + P.e, ge = P.e;                    // the caller passes the single-affine pair in (q, e)
	const int sc_mch = P.sc_mch, sc_mis = P.sc_mis, sc_N = P.sc_ambi;

	for (;;) {
		__syncthreads();
		if (tid == 0) s_job = atomicAdd(job_counter, 1u);
		__syncthreads();
		const uint32_t jid = s_job;
		if (jid >= n_jobs) break;
		const DpJob J = jobs[jid];
		const uint64_t t_base = J.t_off, q_base = J.q_off;          // base positions in the packed store
		const int qlen = J.qlen, tlen = J.tlen;
		const int slen = (qlen + 7) / 8, qlen8 = slen * 8;
		const int R = (tlen + LL_NT - 1) / LL_NT;              // rows per thread (uniform)
		const int n_act = (tlen + R - 1) / R;                  // threads that own at least one row
		const int i0 = tid * R;
		for (int j = tid; j < qlen8; j += LL_NT) {
			int c = 4;
			if (j < qlen) {
				const int pj = J.qs + (J.seq_rev ? qlen - 1 - j : j);
				if (!J.q_rev) c = bases.at(q_base + (uint64_t)(pj));
				else { c = bases.at(q_base + (uint64_t)(J.qlen_full - 1 - pj)); c = c < 4 ? 3 - c : 4; }
			}
			qb[j] = (uint8_t)(j < qlen ? c : 5);                // 5: padding column, score 0
		}
		// this thread's rows: target base, and the mismatch score it pays (N rows pay the ambiguity score against everything)
		int ta[LL_RMAX], tm[LL_RMAX], Hl[LL_RMAX], Fl[LL_RMAX];
#pragma unroll
		for (int k = 0; k < LL_RMAX; ++k) {
			const int i = i0 + k;
			const int a = (k < R && i < tlen) ? (int)bases.at(t_base + (uint64_t)(J.seq_rev ? tlen - 1 - i : i)) : 4;
			ta[k] = a; tm[k] = a == 4 ? sc_N : sc_mis; Hl[k] = 0; Fl[k] = 0;
		}
		if (tid < 2 * (LL_NT / 64)) { (&s_xh[0][0])[tid] = 0; (&s_xe[0][0])[tid] = 0; }
		__syncthreads();
		int out_h = 0, out_e = 0, diag_up = 0;                 // H, E of this thread's last row in the column of the previous step; H above the block one column back
		unsigned long long best = 0;                           // max over (h, row, striped memory index), only cells with h >= 1
		int bh = 1;
		int jm = 0, jd = 0;                                    // j % slen, j / slen of the column this thread is at
		const int n_step = qlen8 + n_act - 1;
		for (int s = 0; s < n_step; ++s) {
			const int j = s - tid;
			// what the row above the block holds in column j: produced by thread tid-1 one step ago
			int up_h = wave_shr1(out_h, 0), up_e = wave_shr1(out_e, 0);
			if (lane == 0 && wave > 0) { up_h = s_xh[(s + 1) & 1][wave - 1]; up_e = s_xe[(s + 1) & 1][wave - 1]; }
			if (tid < n_act && j >= 0 && j < qlen8) {
				const int b = qb[j];
				const bool b_special = b >= 4;
				const int sv = b == 5 ? 0 : sc_N;
				const int midx = jm * 8 + jd;
				int hu = up_h, eu = up_e, hd = diag_up;
#pragma unroll
				for (int k = 0; k < LL_RMAX; ++k) {
					if (k >= R) break;                                 // uniform
					int sc = ta[k] == b ? sc_mch : tm[k];
					if (b_special) sc = sv;
					const int hl = Hl[k];
					int e = max(max(eu - ge, hu - gapoe), 0);
					int f = max(max(Fl[k] - ge, hl - gapoe), 0);
					const int h = max(max(hd + sc, e), f);
					if (h >= bh && i0 + k < tlen) {
						bh = h;
						const unsigned long long key = ((unsigned long long)(unsigned)h << 32) | ((unsigned)(i0 + k) << 16) | (unsigned)midx;
						best = key > best ? key : best;
					}
					hd = hl; Hl[k] = h; Fl[k] = f; hu = h; eu = e;
				}
				diag_up = up_h;
				out_h = hu; out_e = eu;
				if (++jm == slen) jm = 0, ++jd;
			}
			if (lane == 63) { s_xh[s & 1][wave] = out_h; s_xe[s & 1][wave] = out_e; }
			__syncthreads();
		}
		// global maximum: highest score, then the LAST row holding it, then that row's last hit in striped memory order
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) {
			const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(best & 0xffffffffULL), d), hi = (unsigned)__shfl_xor((int)(unsigned)(best >> 32), d);
			const unsigned long long o = ((unsigned long long)hi << 32) | lo;
			best = o > best ? o : best;
		}
		if (lane == 0) s_part[wave] = best;
		__syncthreads();
		if (tid == 0) {
			for (int k = 0; k < LL_NT / 64; ++k) best = s_part[k] > best ? s_part[k] : best;
			const int gmax = (int)(best >> 32);
			DpRes Rr; memset(&Rr, 0, sizeof(Rr));
			Rr.score = gmax;
			if (gmax > 0) {
				const int te = (int)((best >> 16) & 0xffffULL), mi = (int)(best & 0xffffULL);
				Rr.max_t = te; Rr.max_q = mi / 8 + mi % 8 * slen;
			} else { Rr.max_t = tlen - 1; Rr.max_q = qlen8 - 1; }  // all-zero matrix: the last row and the last memory slot win the ties
			res[jid] = Rr;
		}
	}
}

size_t ll_lds_bytes(int t_cap) { return (size_t)t_cap + 64; }

void launch_ll_i16(unsigned n_blocks, int t_cap, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, uint32_t *counter,
                   unsigned long long *rowkey, size_t rowkey_stride, DpRes *res, hipStream_t st)
{
	{	// a per-DEVICE function attribute, set once per device whatever thread comes first
		static std::mutex mu; static bool attr_set[64] = {};
		int dev = 0; PGA_HIP(hipGetDevice(&dev));
		std::lock_guard<std::mutex> lk(mu);
		if (!attr_set[dev & 63]) { PGA_HIP(hipFuncSetAttribute((const void*)k_ll_i16, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024)); attr_set[dev & 63] = true; }
	}
	hipLaunchKernelGGL(k_ll_i16, dim3(n_blocks), dim3(LL_NT), ll_lds_bytes(t_cap), st, jobs, n_jobs, bases, P, counter, rowkey, rowkey_stride, t_cap, res);
}

} // namespace pga
