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

// ---- the same matrix over SEVERAL workgroups (round 6) ----
// One workgroup is bound by the VALU of ONE CU: ten rows per thread, ~16 operations a cell, 10^8 cells of a 10 kb x 10 kb window in 11-13 ms -- and a
// launch of this kernel is what a DP round of a leaf batch waits for (55 launches and 0.71 s of kernel time per build step; doubling that kernel time costs
// a step 5 %: the sensitivity experiment of DESIGN.md section 8).  Here the target rows of a problem are dealt to G groups of NT threads (thread g * NT + t
// owns R = ceil(tlen / (G * NT)) consecutive rows), each the systolic array of k_ll_i16 over its own rows; what group g needs from group g - 1 -- H and E of
// that group's LAST row, column by column -- travels through device memory as one 64-bit word per column with a valid bit (relaxed agent-scope stores by the
// last thread as it leaves a column; loads 64 columns at a time by wave 0 of the consumer, one block ahead of their use, so that the sweep never waits for a
// load once it runs: a consumer starts when its producer is NT + 64 columns in and then keeps that distance).  Every group keeps the maximum of its rows in
// k_ll_i16's key (score, LAST row, last slot in striped order); the group that arrives last combines them.  Groups of one problem have consecutive block
// indices, producers first, and the launcher keeps the groups of a launch few enough that waiting groups can never fill an XCD (ll_groups below): a
// producer always finds a slot.  Measured alone on the device (dev/ll_probe.py, wall time of a launch, host path included), 10 kb x 10 kb: 21.8 ms on one
// workgroup, 12.5 ms on 4 x 1024 threads, 9.3 ms on 16 x 512, **8.7 ms on 16 x 256** (three rows per thread: a group of four waves -- one per SIMD -- pays
// a fraction of a sixteen-wave barrier per column; one wave per group, no barrier at all, is slower again: 9.5 ms on 64 x 64, the hand-over lag of 64
// boundaries); 6 kb x 6 kb: 9.2 -> 5.2 ms; 3.5 kb x 3.5 kb: 4.3 -> 3.1 ms.
#define LL_G_MAX 16
#define LL_BND_COLS PGA_LL_MAX_LEN                          // boundary words per consumer
#define LL_HDR_WORDS 32
#define LL_JOB_WORDS (LL_HDR_WORDS + (LL_G_MAX - 1) * LL_BND_COLS)     // per problem: [0, LL_G_MAX) group maxima, [LL_G_MAX] arrivals, then the boundary columns

template <int NT>
__global__ __launch_bounds__(NT)
void k_ll_multi(const DpJob *__restrict__ jobs, uint32_t n_jobs, int G, PkBases bases, DpParams P, unsigned long long *__restrict__ scratch /* zeroed */, DpRes *__restrict__ res)
{
	extern __shared__ __align__(16) uint8_t dyn[];            // the query, padded to a multiple of 8 columns
	__shared__ unsigned long long s_part[NT / 64];
	__shared__ int s_xh[2][NT / 64], s_xe[2][NT / 64];
	__shared__ int s_in_h[2][64], s_in_e[2][64];              // the producer's columns, two blocks of 64
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t jid = blockIdx.x / (uint32_t)G; const int g = (int)(blockIdx.x % (uint32_t)G);
	if (jid >= n_jobs) return;
	uint8_t *qb = dyn;
	const int gapoe = P.q + P.e, ge = P.e;
	const int sc_mch = P.sc_mch, sc_mis = P.sc_mis, sc_N = P.sc_ambi;
	const DpJob J = jobs[jid];
	const uint64_t t_base = J.t_off, q_base = J.q_off;
	const int qlen = J.qlen, tlen = J.tlen;
	const int slen = (qlen + 7) / 8, qlen8 = slen * 8;
	const int R = (tlen + G * NT - 1) / (G * NT);      // rows per thread (uniform over the problem)
	const int n_act_total = (tlen + R - 1) / R;              // threads of the problem that own at least one row
	int n_act = n_act_total - g * NT; n_act = n_act < 0 ? 0 : n_act > NT ? NT : n_act;
	const bool feeds = (g + 1) * NT < n_act_total;         // the next group has rows: this one is full and its last thread publishes
	unsigned long long *part = scratch + (size_t)jid * LL_JOB_WORDS;
	unsigned long long *bnd_in = part + LL_HDR_WORDS + (size_t)(g > 0 ? g - 1 : 0) * LL_BND_COLS, *bnd_out = part + LL_HDR_WORDS + (size_t)g * LL_BND_COLS;
	const unsigned long long VALID = 1ULL << 63;
	unsigned long long best = 0;
	if (n_act > 0) {
		const int i0 = (g * NT + tid) * R;
		for (int j = tid; j < qlen8; j += NT) {
			int c = 4;
			if (j < qlen) {
				const int pj = J.qs + (J.seq_rev ? qlen - 1 - j : j);
				if (!J.q_rev) c = bases.at(q_base + (uint64_t)(pj));
				else { c = bases.at(q_base + (uint64_t)(J.qlen_full - 1 - pj)); c = c < 4 ? 3 - c : 4; }
			}
			qb[j] = (uint8_t)(j < qlen ? c : 5);
		}
		int ta[LL_RMAX], tm[LL_RMAX], Hl[LL_RMAX], Fl[LL_RMAX];
#pragma unroll
		for (int k = 0; k < LL_RMAX; ++k) {
			const int i = i0 + k;
			const int a = (k < R && i < tlen) ? (int)bases.at(t_base + (uint64_t)(J.seq_rev ? tlen - 1 - i : i)) : 4;
			ta[k] = a; tm[k] = a == 4 ? sc_N : sc_mis; Hl[k] = 0; Fl[k] = 0;
		}
		if (tid < 2 * (NT / 64)) { (&s_xh[0][0])[tid] = 0; (&s_xe[0][0])[tid] = 0; }
		// the producer's first block of columns
		unsigned long long pre = VALID;
		if (g > 0 && wave == 0) {
			for (;;) {
				pre = lane < qlen8 ? __hip_atomic_load(bnd_in + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : VALID;
				if (__ballot(!(pre >> 63)) == 0ULL) break;
				__builtin_amdgcn_s_sleep(8);
			}
			s_in_h[0][lane] = (int)((pre >> 20) & 0xfffffULL); s_in_e[0][lane] = (int)(pre & 0xfffffULL);
		}
		__syncthreads();
		int out_h = 0, out_e = 0, diag_up = 0;
		int bh = 1;
		int jm = 0, jd = 0;
		const int n_step = qlen8 + n_act - 1;
		for (int s = 0; s < n_step; ++s) {
			const int j = s - tid;
			if (g > 0 && wave == 0 && (s & 63) == 0) {           // the block after this one: on its way while this one is swept
				const int idx = ((s >> 6) + 1) * 64 + lane;
				pre = idx < qlen8 ? __hip_atomic_load(bnd_in + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : VALID;
			}
			int up_h = wave_shr1(out_h, 0), up_e = wave_shr1(out_e, 0);
			if (lane == 0 && wave > 0) { up_h = s_xh[(s + 1) & 1][wave - 1]; up_e = s_xe[(s + 1) & 1][wave - 1]; }
			if (g > 0 && tid == 0 && s < qlen8) { up_h = s_in_h[(s >> 6) & 1][s & 63]; up_e = s_in_e[(s >> 6) & 1][s & 63]; }
			if (tid < n_act && j >= 0 && j < qlen8) {
				const int b = qb[j];
				const bool b_special = b >= 4;
				const int sv = b == 5 ? 0 : sc_N;
				const int midx = jm * 8 + jd;
				int hu = up_h, eu = up_e, hd = diag_up;
#pragma unroll
				for (int k = 0; k < LL_RMAX; ++k) {
					if (k >= R) break;
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
				if (feeds && tid == NT - 1)
					__hip_atomic_store(bnd_out + j, VALID | (unsigned long long)(unsigned)out_h << 20 | (unsigned long long)(unsigned)out_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			if (lane == 63) { s_xh[s & 1][wave] = out_h; s_xe[s & 1][wave] = out_e; }
			if (g > 0 && wave == 0 && (s & 63) == 63) {          // the next block must be there now (it has been for ~60 steps unless the producer stalled)
				const int idx = ((s >> 6) + 1) * 64 + lane;
				while (__ballot(!(pre >> 63)) != 0ULL) {
					__builtin_amdgcn_s_sleep(4);
					if (!(pre >> 63)) pre = __hip_atomic_load(bnd_in + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				s_in_h[((s >> 6) + 1) & 1][lane] = (int)((pre >> 20) & 0xfffffULL); s_in_e[((s >> 6) + 1) & 1][lane] = (int)(pre & 0xfffffULL);
			}
			__syncthreads();
		}
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) {
			const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(best & 0xffffffffULL), d), hi = (unsigned)__shfl_xor((int)(unsigned)(best >> 32), d);
			const unsigned long long o = ((unsigned long long)hi << 32) | lo;
			best = o > best ? o : best;
		}
		if (lane == 0) s_part[wave] = best;
		__syncthreads();
		if (tid == 0) for (int k = 0; k < NT / 64; ++k) best = s_part[k] > best ? s_part[k] : best;
	}
	if (tid == 0) {
		__hip_atomic_store(part + g, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		const unsigned long long arrived = __hip_atomic_fetch_add(part + LL_G_MAX, 1ULL, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
		if (arrived == (unsigned long long)(G - 1)) {
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			unsigned long long all = 0;
			for (int k = 0; k < G; ++k) { const unsigned long long v = __hip_atomic_load(part + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); all = v > all ? v : all; }
			const int gmax = (int)(all >> 32);
			DpRes Rr; memset(&Rr, 0, sizeof(Rr));
			Rr.score = gmax;
			if (gmax > 0) {
				const int te = (int)((all >> 16) & 0xffffULL), mi = (int)(all & 0xffffULL);
				Rr.max_t = te; Rr.max_q = mi / 8 + mi % 8 * slen;
			} else { Rr.max_t = tlen - 1; Rr.max_q = qlen8 - 1; }
			res[jid] = Rr;
		}
	}
}

size_t ll_lds_bytes(int t_cap) { return (size_t)t_cap + 64; }
size_t ll_multi_scratch_bytes() { return (size_t)LL_JOB_WORDS * 8; }

// threads per group (PGA_LL_NT; read on every call) and groups per problem for a launch of n_jobs problems whose longest target is t_max: three rows per
// thread, as far as the device has room for all groups of all problems at once (a group of 256 threads is four waves, one per SIMD: its per-column barrier
// costs a fraction of that of sixteen waves, which is what bounds a column step once a thread holds only a few rows); 1 group = the single-workgroup kernel
int ll_group_threads() { const char *e = getenv("PGA_LL_NT"); const int v = e ? atoi(e) : 256; return v == 1024 || v == 512 ? v : 256; }
int ll_groups(uint32_t n_jobs, int t_max)
{
	const int g_env = getenv("PGA_LL_GROUPS") ? atoi(getenv("PGA_LL_GROUPS")) : -1;     // (read on every call: tests force every group count inside one process)
	const int nt = ll_group_threads();
	int G = g_env > 0 ? g_env : (t_max + 3 * nt - 1) / (3 * nt);
	const int g_min = (t_max + LL_RMAX * nt - 1) / (LL_RMAX * nt);                      // a thread holds at most LL_RMAX rows
	if (G < g_min) G = g_min;
	if (G > LL_G_MAX) G = LL_G_MAX;
	// All groups of all problems of a launch are resident at once, and a group that waits for its producer holds its slot while it does.  Workgroups of a launch
	// go round the XCDs, so a consumer can be placed before its producer if the producer's XCD is momentarily full; that resolves as soon as anything there
	// finishes -- unless an XCD were full of WAITING groups only.  128 groups per launch are 16 per XCD (64 of its 1 024 wave slots at 256 threads a group): six
	// launches of six batches in flight cannot fill an XCD between them.
	const uint64_t max_groups = 128;
	while (G > g_min && G > 1 && (uint64_t)n_jobs * (uint64_t)G > max_groups) --G;
	if (G < g_min || (uint64_t)n_jobs * (uint64_t)G > max_groups) return 1;   // no room (or a target beyond the groups' reach): the single-workgroup kernel
	return G < 1 ? 1 : G;
}

template <int NT> static void launch_ll_multi_nt(int G, int t_cap, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, unsigned long long *scratch, DpRes *res, hipStream_t st)
{
	{
		static std::mutex mu; static bool attr_set[64] = {};
		int dev = 0; PGA_HIP(hipGetDevice(&dev));
		std::lock_guard<std::mutex> lk(mu);
		if (!attr_set[dev & 63]) { PGA_HIP(hipFuncSetAttribute((const void*)k_ll_multi<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024)); attr_set[dev & 63] = true; }
	}
	hipLaunchKernelGGL(k_ll_multi<NT>, dim3(n_jobs * (unsigned)G), dim3(NT), ll_lds_bytes(t_cap), st, jobs, n_jobs, G, bases, P, scratch, res);
}
void launch_ll_multi(int G, int t_cap, const DpJob *jobs, uint32_t n_jobs, PkBases bases, const DpParams &P, unsigned long long *scratch, DpRes *res, hipStream_t st)
{
	PGA_HIP(hipMemsetAsync(scratch, 0, (size_t)n_jobs * ll_multi_scratch_bytes(), st));
	const int nt = ll_group_threads();
	if (nt == 1024) launch_ll_multi_nt<1024>(G, t_cap, jobs, n_jobs, bases, P, scratch, res, st);
	else if (nt == 512) launch_ll_multi_nt<512>(G, t_cap, jobs, n_jobs, bases, P, scratch, res, st);
	else launch_ll_multi_nt<256>(G, t_cap, jobs, n_jobs, bases, P, scratch, res, st);
}

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
