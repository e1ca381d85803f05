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
// One 256-thread workgroup per problem sweeps anti-diagonals; a thread owns the target rows i = tid (mod 256), so F and
// the row's own H stay private to it; H (three diagonals) and E (two) are exchanged through LDS with one barrier per
// diagonal.  int16 arithmetic never saturates for the sizes accepted (the launcher checks a * qlen < 32000).
#include "pga_common.h"
#include "pga_dp.h"

namespace pga {

#define LL_NT 1024          // a problem has a CU to itself (LDS): the waves are there to hide LDS latency

__global__ __launch_bounds__(LL_NT)
void k_ll_i16(const DpJob *__restrict__ jobs, uint32_t n_jobs, const uint8_t *__restrict__ nt4, DpParams P, uint32_t *__restrict__ job_counter,
              unsigned long long *__restrict__ rowkey_all, size_t rowkey_stride, int t_cap, DpRes *__restrict__ res)
{
	extern __shared__ __align__(16) uint8_t dyn[];
	__shared__ uint32_t s_job;
	__shared__ unsigned long long s_part[LL_NT / 64];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	__builtin_amdgcn_s_setprio(3);
	int16_t *Hb[3] = { (int16_t*)dyn, (int16_t*)dyn + t_cap, (int16_t*)dyn + 2 * t_cap };
	int16_t *Eb[2] = { (int16_t*)dyn + 3 * t_cap, (int16_t*)dyn + 4 * t_cap };
	int16_t *Fr = (int16_t*)dyn + 5 * t_cap;
	uint8_t *tb = (uint8_t*)((int16_t*)dyn + 6 * t_cap), *qb = tb + t_cap;
	unsigned long long *rowkey = rowkey_all + (size_t)blockIdx.x * rowkey_stride;
	const int gapoe = P.q + P.e, ge = P.e;                    // the caller passes the single-affine pair in (q, e)
	const int sc_mch = P.sc_mch, sc_mis = P.sc_mis, sc_N = P.sc_ambi;

	for (;;) {
		__syncthreads();
		if (tid == 0) s_job = atomicAdd(job_counter, 1u);
		__syncthreads();
		const uint32_t jid = s_job;
		if (jid >= n_jobs) break;
		const DpJob J = jobs[jid];
		const uint8_t *t_base = nt4 + J.t_off, *q_base = nt4 + J.q_off;
		const int qlen = J.qlen, tlen = J.tlen;
		const int slen = (qlen + 7) / 8, qlen8 = slen * 8;
		for (int i = tid; i < tlen; i += LL_NT) {
			tb[i] = t_base[J.seq_rev ? tlen - 1 - i : i];
			Hb[0][i] = Hb[1][i] = Hb[2][i] = 0; Eb[0][i] = Eb[1][i] = 0; Fr[i] = 0;
			rowkey[i] = 0;
		}
		for (int j = tid; j < qlen8; j += LL_NT) {
			int c = 4;
			if (j < qlen) {
				const int pj = J.qs + (J.seq_rev ? qlen - 1 - j : j);
				if (!J.q_rev) c = q_base[pj];
				else { c = q_base[J.qlen_full - 1 - pj]; c = c < 4 ? 3 - c : 4; }
			}
			qb[j] = (uint8_t)(j < qlen ? c : 5);                // 5: padding column, score 0
		}
		__syncthreads();
		int tbest = 1;                                         // row keys are only kept for cells that could still be the maximum
		const int n_diag = tlen + qlen8 - 1;
		for (int r = 0; r < n_diag; ++r) {
			const int ilo = r - (qlen8 - 1) > 0 ? r - (qlen8 - 1) : 0, ihi = r < tlen - 1 ? r : tlen - 1;
			const int16_t *H1 = Hb[(r + 2) % 3], *H2 = Hb[(r + 1) % 3]; int16_t *H0 = Hb[r % 3];   // H1: diagonal r-1, H2: diagonal r-2
			const int16_t *E1 = Eb[(r + 1) & 1]; int16_t *E0 = Eb[r & 1];
			int i = ilo + ((tid - ilo) & (LL_NT - 1));            // first row >= ilo owned by this thread
			for (; i <= ihi; i += LL_NT) {
				const int j = r - i;
				const int a = tb[i], b = qb[j];
				int s = b == 5 ? 0 : (a == 4 || b == 4) ? sc_N : (a == b ? sc_mch : sc_mis);
				const int hd = (i > 0 && j > 0) ? (int)H2[i - 1] : 0;                 // H(i-1, j-1)
				const int hu = i > 0 ? (int)H1[i - 1] : 0, eu = i > 0 ? (int)E1[i - 1] : 0;  // H(i-1, j), E(i-1, j)
				const int hl = j > 0 ? (int)H1[i] : 0, fl = j > 0 ? (int)Fr[i] : 0;          // H(i, j-1), F(i, j-1)
				int e = eu - ge; { const int t = hu - gapoe; e = e > t ? e : t; } if (e < 0) e = 0;
				int f = fl - ge; { const int t = hl - gapoe; f = f > t ? f : t; } if (f < 0) f = 0;
				if (i == 0) e = 0;
				if (j == 0) f = 0;
				int h = hd + s; h = h > e ? h : e; h = h > f ? h : f;
				H0[i] = (int16_t)h; E0[i] = (int16_t)e; Fr[i] = (int16_t)f;
				if (h >= tbest) {
					tbest = h;
					const unsigned long long key = ((unsigned long long)(unsigned)h << 32) | (unsigned)((j % slen) * 8 + j / slen);
					if (key > rowkey[i]) rowkey[i] = key;
				}
			}
			__syncthreads();
		}
		// global maximum, last row holding it, and that row's last hit in striped memory order
		unsigned long long best = 0;
		for (int i = tid; i < tlen; i += LL_NT) {
			const unsigned long long k = ((rowkey[i] >> 32) << 32) | (unsigned)i;
			best = k > best ? k : best;
		}
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
			DpRes R; memset(&R, 0, sizeof(R));
			R.score = gmax;
			if (gmax > 0) {
				const int te = (int)(best & 0xffffffffULL);
				const int mi = (int)(rowkey[te] & 0xffffffffULL);
				R.max_t = te; R.max_q = mi / 8 + mi % 8 * slen;
			} else { R.max_t = tlen - 1; R.max_q = qlen8 - 1; }    // all-zero matrix: the last row and the last memory slot win the ties
			res[jid] = R;
		}
	}
}

size_t ll_lds_bytes(int t_cap) { return (size_t)t_cap * 14; }

void launch_ll_i16(unsigned n_blocks, int t_cap, const DpJob *jobs, uint32_t n_jobs, const uint8_t *nt4, const DpParams &P, uint32_t *counter,
                   unsigned long long *rowkey, size_t rowkey_stride, DpRes *res, hipStream_t st)
{
	static bool attr_set = false;
	if (!attr_set) { PGA_HIP(hipFuncSetAttribute((const void*)k_ll_i16, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024)); attr_set = true; }
	hipLaunchKernelGGL(k_ll_i16, dim3(n_blocks), dim3(LL_NT), ll_lds_bytes(t_cap), st, jobs, n_jobs, nt4, P, counter, rowkey, rowkey_stride, t_cap, res);
}

} // namespace pga
