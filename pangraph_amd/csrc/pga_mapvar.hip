// pga_mapvar.hip -- SURVEY 8(f)-1: every member sequence of a merged block re-aligned onto the anchor consensus.
//
//   MergePromise::solve_promise     packages/pangraph/src/pangraph/reweave.rs:40-94      (the caller: one job per member sequence)
//   map_variations                  packages/pangraph/src/align/map_variations.rs:39-77
//   align_with_nextclade            packages/pangraph/src/align/nextclade/align_with_nextclade.rs:24-75
//   align_nuc_simplestripe          packages/pangraph/src/align/nextclade/align/align.rs:32-71   (band retry loop)
//   simple_stripes                  .../align/band_2d.rs:36-57
//   score_matrix                    .../align/score_matrix.rs:23-199
//   backtrace                       .../align/backtrace.rs:17-85
//   insertions_strip                .../align/insertions_strip.rs:47-97
//   find_nuc_changes                .../analyze/nuc_changes.rs:18-71
//
// ONE WAVE PER JOB, persistent waves on an atomic queue.  The reference fills the band row by row, cell by cell; the only
// dependency inside a row is the horizontal gap (ref_gaps).  Here a row is one step of the wave, a lane per column:
//   * Ht[q] = max(diagonal move, vertical gap) needs the previous row only (scores and qry_gaps live in LDS rings over the column
//     index: a column keeps its slot for as long as it is inside the band);
//   * ref_gaps[q] = max over j < q of Ht[j] - open - e * (q - 1 - j), e = min(extend, open) -- the reference's recurrence
//     max(ref_gaps[q-1] - extend, S[q-1] - open) with S[q-1] = max(Ht[q-1], ref_gaps[q-1]) unrolled -- is ONE prefix maximum over
//     the lanes (six DPP steps); rows wider than 64 columns go chunk by chunk with a carry;
//   * the reference's comparisons (which move wins a tie, the "extend" flags, the BOUNDARY marks) are then evaluated per cell on
//     the exact values, in the reference's order, so the path bytes are the reference's bytes.
// Path bytes go to the wave's slab (row pitch 2 * band_width + 2); scores are never stored (only the corner score is reported).
// Backtrace: the wave looks 64 cells ahead along the current move direction (diagonal, row or column) with one load and consumes
// the whole run: a 10 kb alignment is ~200 trips instead of 20 k dependent loads.  Runs (kind, length) are written back to front,
// then walked front to back twice (count, write) to produce substitutions, deletions and insertions in the reference's order.
// Jobs whose path touched a BOUNDARY cell come back with hit = 1; the host doubles the band (align.rs:55-62) and queues them again.
#include "pga_common.h"
#include "pga_wave.h"
#include "../../include/pga_align.h"
#include <unordered_map>
#include <cstdio>
#include <chrono>
#include <thread>

namespace pga {

#define MV_NO_ALIGN (-1000000000)     // score_matrix.rs:13
enum { MV_MATCH = 1, MV_REF_GAP_MATRIX = 2, MV_QRY_GAP_MATRIX = 4, MV_REF_GAP_EXTEND = 8, MV_QRY_GAP_EXTEND = 16, MV_BOUNDARY = 32 };   // :6-11
#define MV_N 14
#define MV_GAP 15
#define MV_BAD 255

struct MvParams { int32_t match, mismatch, gap_open, ext, left_free, right_free, left_align, min_length, max_attempts; };
struct MvJob { uint64_t ref_off, qry_off; uint32_t ref_len, qry_len; int32_t ms; uint32_t bw, attempt, orig; };
struct MvDevJob { uint64_t ref_off, qry_off; uint32_t ref_len, qry_len; int32_t mean_shift; uint32_t band_width; };   // sequences already in device memory (pga_reconsensus)
struct MvOut { int32_t status, score, attempts, hit; uint32_t n_subs, n_dels, n_inss, n_ib; uint64_t sub_off, del_off, ins_off, ib_off; };
struct MvCursors { unsigned long long subs, dels, inss, ib; };
struct MvCaps { unsigned long long subs, dels, inss, ib; };

// nuc.rs:10-30 / :99-121: the sixteen letters in enum order; anything else is an error of the job.  The gap letter '-' is REJECTED too
// (status 2): to_nuc accepts it, but find_nuc_changes / insertions_strip then read a literal '-' of the input as an alignment gap
// (is_gap()), which the run-based edit extraction here does not reproduce -- and pangraph's block sequences never hold one.
__global__ void k_mv_encode(const char *__restrict__ ascii, uint64_t n, uint8_t *__restrict__ codes)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		uint8_t c;
		switch (ascii[i]) {
		case 'T': c = 0; break; case 'A': c = 1; break; case 'W': c = 2; break; case 'C': c = 3; break;
		case 'Y': c = 4; break; case 'M': c = 5; break; case 'H': c = 6; break; case 'G': c = 7; break;
		case 'K': c = 8; break; case 'R': c = 9; break; case 'D': c = 10; break; case 'S': c = 11; break;
		case 'B': c = 12; break; case 'V': c = 13; break; case 'N': c = 14; break;
		default: c = MV_BAD;
		}
		codes[i] = c;
	}
}

__device__ __forceinline__ char mv_letter(int c) { return "TAWCYMHGKRDSBVN-"[c & 15]; }
// score_matrix_nuc.rs:6-26: letters are the 4-bit sets 1..15 over {T, A, C, G}; two match when the sets intersect; the gap letter matches N and itself
__device__ __forceinline__ bool mv_match(int x, int y) { return (x == MV_GAP || y == MV_GAP) ? (x >= MV_N && y >= MV_N) : (((x + 1) & (y + 1)) != 0); }
__device__ __forceinline__ unsigned long long mv_low(int n) { return n >= 64 ? ~0ULL : ((1ULL << n) - 1); }
__device__ __forceinline__ void mv_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
// between rows: LDS operations of one wave execute in order, so the LDS rings only need the compiler to keep that order
template <int RING> __device__ __forceinline__ void mv_row_fence() { if (RING > 0) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// traceback and edit extraction of one job by the whole wave (the path bytes are in `slab`, written by the score-matrix pass)
__device__ __forceinline__ void mv_finish(const uint8_t *ref, const uint8_t *qry, int ref_len, int qlen, long long ms, long long bw, int attempt, uint8_t *slab,
                                          const MvParams &P, int lane, MvOut &O, MvCursors *cur, const MvCaps &cap, pga_sub_t *subs, pga_del_t *dels, pga_ins_t *inss, char *ins_seq)
{
	auto sbeg = [&](int i) -> int { if (i == 0) return 0; const long long v = (long long)i - ms - bw; return (int)(v < 0 ? 0 : v > qlen ? qlen : v); };
	auto send = [&](int i) -> int { if (i == ref_len) return qlen + 1; const long long v = (long long)i - ms + bw + 1; return (int)(v < 1 ? 1 : v > (long long)qlen + 1 ? qlen + 1 : v); };
	const long long pitch_ll = 2 * bw + 2 < (long long)qlen + 2 ? 2 * bw + 2 : (long long)qlen + 2;
	const size_t pitch = (size_t)pitch_ll;
	const size_t path_bytes = ref_len > 0 ? (size_t)(ref_len - 1) * pitch + (size_t)qlen + 2 : 0;
	uint32_t *runs = (uint32_t*)(slab + ((path_bytes + 15) & ~(size_t)15));
	auto paddr = [&](int ri, int q) -> size_t { return (size_t)(ri - 1) * pitch + (size_t)(q - sbeg(ri)); };
	uint32_t n_runs = 0;
	bool go = true;
	// ---- backtrace.rs:17-85, 64 cells per trip ----
	int r = ref_len, q = qlen, cm = 0, hit = 0, cur_t = -1;
	uint32_t cur_n = 0;
	auto push = [&](int t, uint32_t n) {
		if (t == cur_t) { cur_n += n; return; }
		if (cur_t >= 0) { if (lane == 0) runs[n_runs] = (uint32_t)cur_t << 30 | cur_n; ++n_runs; }
		cur_t = t; cur_n = n;
	};
	while ((r > 0 || q > 0) && O.status == 0) {
		if (r == 0) {                                                       // row 0 is REF_GAP_EXTEND + REF_GAP_MATRIX all the way (:63-64)
			if (q >= send(0)) { O.status = 3; break; }
			push(1, (uint32_t)q); q = 0; break;
		}
		int rr = r - lane, qq = q - lane;
		bool v = rr >= 1 && qq >= 0 && qq >= sbeg(rr) && qq < send(rr);
		int o = v ? (int)slab[paddr(rr, qq)] : 0;
		if (!(__ballot(v) & 1ULL)) { O.status = 3; break; }                 // the reference would panic (band_2d.rs:118-124)
		const int o0 = rl(o, 0);
		if (o0 & MV_BOUNDARY) hit = 1;
		if (cm == 0 && (o0 & MV_MATCH)) {
			const unsigned long long m = __ballot(v && (o & MV_MATCH) && qq >= 1);
			const int n = m == ~0ULL ? 64 : __builtin_ctzll(~m);
			if (__ballot(o & MV_BOUNDARY) & mv_low(n)) hit = 1;
			push(0, (uint32_t)n); r -= n; q -= n;
		} else if ((cm == 0 && (o0 & MV_REF_GAP_MATRIX)) || cm == MV_REF_GAP_MATRIX) {
			qq = q - lane;
			v = qq >= 1 && qq >= sbeg(r) && qq < send(r);
			o = v ? (int)slab[paddr(r, qq)] : 0;
			const unsigned long long vm = __ballot(v), em = __ballot(v && (o & MV_REF_GAP_EXTEND));
			if (!(vm & 1ULL)) { O.status = 3; break; }
			const int k = em == ~0ULL ? 64 : __builtin_ctzll(~em);
			int n;
			if (k < 64 && ((vm >> k) & 1ULL)) { n = k + 1; cm = 0; } else { n = k; cm = MV_REF_GAP_MATRIX; }
			if (__ballot(o & MV_BOUNDARY) & mv_low(n)) hit = 1;
			push(1, (uint32_t)n); q -= n;
		} else if ((cm == 0 && (o0 & MV_QRY_GAP_MATRIX)) || cm == MV_QRY_GAP_MATRIX) {
			rr = r - lane;
			v = rr >= 1 && q >= sbeg(rr) && q < send(rr);
			o = v ? (int)slab[paddr(rr, q)] : 0;
			const unsigned long long vm = __ballot(v), em = __ballot(v && (o & MV_QRY_GAP_EXTEND));
			if (!(vm & 1ULL)) { O.status = 3; break; }
			const int k = em == ~0ULL ? 64 : __builtin_ctzll(~em);
			int n;
			if (k < 64 && ((vm >> k) & 1ULL)) { n = k + 1; cm = 0; } else { n = k; cm = MV_QRY_GAP_MATRIX; }
			if (__ballot(o & MV_BOUNDARY) & mv_low(n)) hit = 1;
			push(2, (uint32_t)n); r -= n;
		} else O.status = 3;                                                 // unreachable!() in the reference
	}
	if (cur_t >= 0) { if (lane == 0) runs[n_runs] = (uint32_t)cur_t << 30 | cur_n; ++n_runs; }
	mv_fence();
	O.hit = hit;
	if (O.status || (hit && attempt < P.max_attempts)) go = false;   // align.rs:55: another attempt with a wider band
	// ---- insertions_strip + find_nuc_changes + the terminal deletions, from the runs (front to back = the list backwards) ----
	const unsigned long long lt = (1ULL << lane) - 1;
	for (int pass = 0; go && pass < 2; ++pass) {
		uint32_t n_subs = 0, n_dels = 0, n_inss = 0, n_ib = 0;
		long long n_del = 0, del_pos = -1, a_start = -1, a_end = -1;
		bool before = true;
		int rp = 0, qp = 0;
		for (int t = (int)n_runs - 1; t >= 0; --t) {
			const uint32_t w = runs[t];
			const int kind = (int)(w >> 30); const int L = (int)(w & 0x3fffffffu);
			if (kind == 0) {
				if (before) { a_start = rp; before = false; }
				else if (n_del > 0) { if (pass && lane == 0) { dels[O.del_off + n_dels].pos = (uint32_t)del_pos; dels[O.del_off + n_dels].len = (uint32_t)n_del; } ++n_dels; n_del = 0; }
				for (int i0 = 0; i0 < L; i0 += 64) {
					const int i = i0 + lane;
					int a = 0, c = 0;
					if (i < L) { a = ref[rp + i]; c = qry[qp + i]; }
					const bool diff = i < L && a != c;
					const unsigned long long dm = __ballot(diff);
					if (pass && diff) { pga_sub_t s; s.pos = (uint32_t)(rp + i); s.alt = (uint32_t)mv_letter(c); subs[O.sub_off + n_subs + (uint32_t)__popcll(dm & lt)] = s; }
					n_subs += (uint32_t)__popcll(dm);
				}
				rp += L; qp += L; a_end = rp;
			} else if (kind == 1) {
				if (pass) {
					if (lane == 0) { pga_ins_t s; s.pos = (uint32_t)rp; s.len = (uint32_t)L; s.seq_off = O.ib_off + n_ib; inss[O.ins_off + n_inss] = s; }   // map_variations.rs:71-74: position + 1
					for (int i = lane; i < L; i += 64) ins_seq[O.ib_off + n_ib + (uint32_t)i] = mv_letter(qry[qp + i]);
				}
				++n_inss; n_ib += (uint32_t)L; qp += L;
			} else {
				if (!before) { if (n_del == 0) del_pos = rp; n_del += L; }
				rp += L;
			}
		}
		// align_with_nextclade.rs:46-64: leading and trailing gaps, behind the internal deletions
		if (a_start >= 0 && a_end >= 0) {
			if (a_start > 0) { if (pass && lane == 0) { dels[O.del_off + n_dels].pos = 0; dels[O.del_off + n_dels].len = (uint32_t)a_start; } ++n_dels; }
			if (a_end < ref_len) { if (pass && lane == 0) { dels[O.del_off + n_dels].pos = (uint32_t)a_end; dels[O.del_off + n_dels].len = (uint32_t)(ref_len - a_end); } ++n_dels; }
		} else { if (pass && lane == 0) { dels[O.del_off + n_dels].pos = 0; dels[O.del_off + n_dels].len = (uint32_t)ref_len; } ++n_dels; }
		if (pass == 0) {
			O.n_subs = n_subs; O.n_dels = n_dels; O.n_inss = n_inss; O.n_ib = n_ib;
			unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
			if (lane == 0) { a0 = atomicAdd(&cur->subs, (unsigned long long)n_subs); a1 = atomicAdd(&cur->dels, (unsigned long long)n_dels); a2 = atomicAdd(&cur->inss, (unsigned long long)n_inss); a3 = atomicAdd(&cur->ib, (unsigned long long)n_ib); }
			O.sub_off = (uint64_t)__shfl((long long)a0, 0); O.del_off = (uint64_t)__shfl((long long)a1, 0); O.ins_off = (uint64_t)__shfl((long long)a2, 0); O.ib_off = (uint64_t)__shfl((long long)a3, 0);
			// the pools are sized for typical divergence, not for the worst case: a job that does not fit comes back (status 4) and runs again
			if (O.sub_off + n_subs > cap.subs || O.del_off + n_dels > cap.dels || O.ins_off + n_inss > cap.inss || O.ib_off + n_ib > cap.ib) { O.status = 4; go = false; }
		}
	}
}

// RING > 0: the rings live in LDS (RING columns); RING == 0: in device memory, ring_n columns per wave
template <int RING>
__global__ __launch_bounds__(64)
void k_mapvar(const MvJob *__restrict__ jobs, int n_jobs, MvParams P, const uint8_t *__restrict__ codes, uint32_t *job_counter,
              uint8_t *slabs, uint64_t slab_bytes, int32_t *gring, int ring_n, MvOut *__restrict__ out, MvCursors *cur, MvCaps cap,
              pga_sub_t *subs, pga_del_t *dels, pga_ins_t *inss, char *ins_seq)
{
	__shared__ int32_t s_ring[RING > 0 ? 3 * RING : 1];
	const int lane = threadIdx.x;
	const int RN = RING > 0 ? RING : ring_n, M = RN - 1;
	int32_t *ring = RING > 0 ? s_ring : gring + (size_t)blockIdx.x * 3 * (size_t)ring_n;
	uint8_t *slab = slabs + (size_t)blockIdx.x * slab_bytes;
	const int la = P.left_align ? 1 : 0;                                   // score_matrix.rs:44-47
	for (;;) {
		// (no `if (lane == 0)` around the atomic: with that branch the compiler split this loop into one for lane 0 and one for the other
		// lanes, which then went round alone and read job 0 for ever.  Every lane adds, lane 0 adds one: one atomic per wave after all.)
		int j = (int)atomicAdd(job_counter, lane == 0 ? 1u : 0u);
		j = __builtin_amdgcn_readfirstlane(j);
		if (j >= n_jobs) break;
		const MvJob J = jobs[j];
		const int ref_len = (int)J.ref_len, qlen = (int)J.qry_len;
		const long long ms = J.ms, bw = J.bw;
		const uint8_t *ref = codes + J.ref_off, *qry = codes + J.qry_off;
		MvOut O; memset(&O, 0, sizeof(O)); O.attempts = (int32_t)J.attempt;
		// ---- to_nuc_seq of both sequences (align_with_nextclade.rs:30-31), then the length test (align.rs:42-46) ----
		{
			bool bad = false;
			for (int i = lane; i < ref_len; i += 64) bad |= ref[i] == MV_BAD;
			for (int i = lane; i < qlen; i += 64) bad |= qry[i] == MV_BAD;
			if (__ballot(bad)) O.status = 2; else if (qlen < P.min_length) O.status = 1;
		}
		// (one way through the body and one back edge: no continue statements)
		bool go = O.status == 0;
		// band_2d.rs:36-57
		auto sbeg = [&](int i) -> int { if (i == 0) return 0; const long long v = (long long)i - ms - bw; return (int)(v < 0 ? 0 : v > qlen ? qlen : v); };
		auto send = [&](int i) -> int { if (i == ref_len) return qlen + 1; const long long v = (long long)i - ms + bw + 1; return (int)(v < 1 ? 1 : v > (long long)qlen + 1 ? qlen + 1 : v); };
		const long long pitch_ll = 2 * bw + 2 < (long long)qlen + 2 ? 2 * bw + 2 : (long long)qlen + 2;
		const size_t pitch = (size_t)pitch_ll;
		const size_t path_bytes = ref_len > 0 ? (size_t)(ref_len - 1) * pitch + (size_t)qlen + 2 : 0;
		uint32_t *runs = (uint32_t*)(slab + ((path_bytes + 15) & ~(size_t)15));
		auto paddr = [&](int ri, int q) -> size_t { return (size_t)(ri - 1) * pitch + (size_t)(q - sbeg(ri)); };

		// ---- score_matrix.rs:23-199 ----
		uint32_t n_runs = 0;
		if (go) {
		int32_t *bufS = ring, *qg = ring + 2 * (size_t)RN;
		for (int i = lane; i < RN; i += 64) qg[i] = MV_NO_ALIGN;                                     // :77
		{
			const int e0 = send(0);                                                                   // row 0 (:63-75): only its last RN columns can be read
			for (int q = (e0 > RN ? e0 - RN : 0) + lane; q < e0; q += 64) bufS[q & M] = (q == 0 || P.left_free) ? 0 : -(P.gap_open + (q - 1) * P.ext);
		}
		mv_fence();
		int32_t final_score = (P.left_free || qlen == 0) ? 0 : -(P.gap_open + (qlen - 1) * P.ext);   // ref_len == 0: the corner lies in row 0
		int refreg = 0;
		int qc_pref = 0;
		// the stripes of the rows ri - 2 (end), ri - 1, ri, ri + 1 roll along; the cell arithmetic is branch-free (a branch costs more than the cell)
		int pb = 0, pe = send(0), ppe = 0, b = ref_len >= 1 ? sbeg(1) : 0, e = ref_len >= 1 ? send(1) : 0;
		{ const int q1 = b + lane; if (ref_len >= 1 && q1 >= 1 && q1 < e) qc_pref = qry[q1 - 1]; }
		for (int ri = 1; ri <= ref_len; ++ri) {
			if (((ri - 1) & 63) == 0) refreg = ri - 1 + lane < ref_len ? ref[ri - 1 + lane] : 0;
			const int r = rl(refreg, (ri - 1) & 63);
			const bool rN = r == MV_N;
			const int mr = rN ? 31 : r + 1;                                                            // letters as sets: mv_match(x, y) = sets intersect
			const bool last = ri == ref_len;
			const int nb = last ? 0 : sbeg(ri + 1), ne = last ? 0 : send(ri + 1);
			const int o_r = (last && P.right_free) ? 0 : P.gap_open, x_r = (last && P.right_free) ? 0 : P.ext;    // :136-143
			const int ep = x_r < o_r ? x_r : o_r;
			const int32_t s0 = P.left_free ? 0 : -(P.gap_open + (ri - 1) * P.ext);                     // column 0 (:96-107)
			const int32_t *prevS = bufS + (size_t)((ri - 1) & 1) * RN;
			int32_t *curS = bufS + (size_t)(ri & 1) * RN;
			int32_t carryA = INT32_MIN, carryS = 0, carryG = 0;
			const size_t rowbase = (size_t)(ri - 1) * pitch;
			const int qc_first = qc_pref;
			{ const int qn = nb + lane; qc_pref = (qn >= 1 && qn < ne) ? qry[qn - 1] : 0; }             // the next row's letters, in flight during this row
			for (int c0 = b; c0 < e; c0 += 64) {
				const int q = c0 + lane;
				const bool on = q < e;
				const int qc = c0 == b ? qc_first : ((on && q >= 1) ? (int)qry[q - 1] : 0);
				const int32_t Sd = prevS[(q - 1) & M], Su = prevS[q & M], QG = qg[q & M];
				const bool q0 = q == 0;
				const bool inner = !last && q < qlen;
				const bool diag_ok = q > pb && q - 1 < pe;                                              // :115-124
				const bool up_ok = !q0 && q < pe;                                                       // :165
				const int mq = qc == MV_N ? 31 : qc + 1;
				const int sc = (qc == MV_N || rN) ? P.match - 1 : ((mq & mr) ? P.match : -P.mismatch);
				int32_t score = diag_ok ? Sd + sc : MV_NO_ALIGN;
				int origin = diag_ok ? MV_MATCH : 0;
				const bool fr = q == qlen && P.right_free;
				const int32_t qe = QG - (fr ? 0 : P.ext), qo = Su - (fr ? 0 : P.gap_open);
				const bool extq = up_ok && qe >= qo && q < ppe;                                         // :175 (ppe = 0 in row 1)
				const int32_t tq = extq ? qe : qo;
				int32_t Ht = (up_ok && tq > score) ? tq : score;
				Ht = q0 ? s0 : Ht;
				// ref_gaps of every column of the chunk: exclusive prefix maximum of Ht[j] + ep * (j - b)
				const int32_t A = on ? Ht + ep * (q - b) : INT32_MIN;
				const int32_t Pin = wave_prefix_max_incl(A);
				int32_t excl = wave_shr1(Pin, INT32_MIN);
				excl = carryA > excl ? carryA : excl;
				{ const int32_t top = rl(Pin, 63); carryA = top > carryA ? top : carryA; }
				const int32_t G = q > b ? excl - o_r - ep * (q - 1 - b) : 0;
				const bool refg = q > b && score - la < G;                                              // :147-150
				score = refg ? G : score; origin = refg ? MV_REF_GAP_MATRIX : origin;
				const bool qryg = up_ok && score - la < tq;                                             // :180-183
				score = qryg ? tq : score; origin = qryg ? MV_QRY_GAP_MATRIX : origin;
				score = q0 ? s0 : score; origin = q0 ? MV_QRY_GAP_MATRIX : origin;
				const int32_t Sl = wave_shr1(score, carryS), Gl = wave_shr1(G, carryG);
				int tmp_path = q0 ? MV_QRY_GAP_EXTEND : ((inner && (!diag_ok || q <= b || !up_ok)) ? MV_BOUNDARY : 0) + (extq ? MV_QRY_GAP_EXTEND : 0);
				tmp_path += (q > b + 1 && Gl - x_r >= Sl - o_r) ? MV_REF_GAP_EXTEND : 0;                // :144-146
				carryS = rl(score, 63); carryG = rl(G, 63);
				if (on) {
					slab[rowbase + (size_t)(q - b)] = (uint8_t)(tmp_path + origin);
					curS[q & M] = score;
					qg[q & M] = up_ok ? tq : MV_NO_ALIGN;                                               // :184-187 (qry_gaps of a column entering the band is NO_ALIGN)
				}
				if (last && qlen >= c0 && qlen < c0 + 64) final_score = rl(score, qlen - c0);
			}
			mv_row_fence<RING>();
			ppe = pe; pb = b; pe = e; b = nb; e = ne;
		}
		mv_fence();                                                         // the path bytes, for the lanes that read them below
		O.score = final_score;

		}
		if (go) mv_finish(ref, qry, ref_len, qlen, ms, bw, (int)J.attempt, slab, P, lane, O, cur, cap, subs, dels, inss, ins_seq);
		if (lane == 0) out[j] = O;
	}
}

// ---- narrow bands: SEVERAL JOBS PER WAVE ----
// A band of 2 * band_width + 1 <= 31 columns leaves half of the lanes of k_mapvar idle (<= 15: three quarters).  Here a wave takes 64 / SEG
// consecutive jobs of the queue (sorted by size: neighbours are members of the same block), a segment of SEG lanes each: every row of
// every job fits one segment (the host checks the last row too, which reaches to the end of the query), so there is no chunk loop and no
// carry; the prefix maximum stops at the segment borders (row-local DPP steps, plus row_bcast15 for 32-lane segments); stripes, letters
// and scores are per-lane values, the rings per segment.  The jobs of a wave share the row loop (it runs to the longest reference), the
// traceback and the edits are then done job by job by the whole wave (mv_finish).
template <int SEG> __device__ __forceinline__ int32_t mv_seg_prefix_max(int32_t v)
{
	v = dpp_max_step<0x111, 0xf>(v); v = dpp_max_step<0x112, 0xf>(v); v = dpp_max_step<0x114, 0xf>(v); v = dpp_max_step<0x118, 0xf>(v);
	if (SEG == 32) v = dpp_max_step<0x142, 0xa>(v);                    // row_bcast15 into rows 1 and 3
	return v;
}

template <int SEG>
__global__ __launch_bounds__(64)
void k_mapvar_packed(const MvJob *__restrict__ jobs, int n_jobs, MvParams P, const uint8_t *__restrict__ codes, uint32_t *job_counter,
                     uint8_t *slabs, uint64_t slab_bytes, MvOut *__restrict__ out, MvCursors *cur, MvCaps cap,
                     pga_sub_t *subs, pga_del_t *dels, pga_ins_t *inss, char *ins_seq)
{
	constexpr int NJ = 64 / SEG, RN = 64, M = RN - 1;
	__shared__ int32_t s_ring[NJ * 3 * RN];
	const int lane = threadIdx.x, seg = lane / SEG, sl = lane % SEG;
	int32_t *bufS = s_ring + seg * 3 * RN, *qg = bufS + 2 * RN;
	const int la = P.left_align ? 1 : 0;
	for (;;) {
		int j0 = (int)atomicAdd(job_counter, lane == 0 ? (uint32_t)NJ : 0u);          // (no lane-0 branch: see k_mapvar)
		j0 = __builtin_amdgcn_readfirstlane(j0);
		if (j0 >= n_jobs) break;
		const int nj = n_jobs - j0 < NJ ? n_jobs - j0 : NJ;
		const bool have = seg < nj;
		const MvJob J = jobs[j0 + (have ? seg : 0)];
		const int ref_len = (int)J.ref_len, qlen = (int)J.qry_len, ms = J.ms, bw = (int)J.bw;
		const uint8_t *ref = codes + J.ref_off, *qry = codes + J.qry_off;
		uint8_t *slab = slabs + ((size_t)blockIdx.x * NJ + (size_t)seg) * slab_bytes;
		int status_l = 0;                                                               // to_nuc_seq and the length test, job by job
		for (int s = 0; s < nj; ++s) {
			const MvJob Js = jobs[j0 + s];
			const uint8_t *r2 = codes + Js.ref_off, *q2 = codes + Js.qry_off;
			bool bad = false;
			for (int i = lane; i < (int)Js.ref_len; i += 64) bad |= r2[i] == MV_BAD;
			for (int i = lane; i < (int)Js.qry_len; i += 64) bad |= q2[i] == MV_BAD;
			const int st = __ballot(bad) ? 2 : ((int)Js.qry_len < P.min_length ? 1 : 0);
			status_l = seg == s ? st : status_l;
		}
		const bool act = have && status_l == 0;
		auto sbeg = [&](int i) -> int { if (i == 0) return 0; const int v = i - ms - bw; return v < 0 ? 0 : v > qlen ? qlen : v; };
		auto send = [&](int i) -> int { if (i == ref_len) return qlen + 1; const int v = i - ms + bw + 1; return v < 1 ? 1 : v > qlen + 1 ? qlen + 1 : v; };
		const int pitch = 2 * bw + 2 < qlen + 2 ? 2 * bw + 2 : qlen + 2;
		for (int i = sl; i < RN; i += SEG) qg[i] = MV_NO_ALIGN;
		{
			const int e0 = send(0);
			for (int q = (e0 > RN ? e0 - RN : 0) + sl; q < e0; q += SEG) bufS[q & M] = (q == 0 || P.left_free) ? 0 : -(P.gap_open + (q - 1) * P.ext);
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		const int max_len = wave_max_i32(act ? ref_len : 0);
		int32_t fin = 0; bool hasfin = false;
		int pb = 0, pe = send(0), ppe = 0, b = sbeg(1), e = send(1);
		int rnext = act ? (int)ref[0] : 0;
		int qc_pref = 0;
		{ const int q1 = b + sl; if (act && q1 >= 1 && q1 < e) qc_pref = qry[q1 - 1]; }
		for (int ri = 1; ri <= max_len; ++ri) {
			const bool rowon = act && ri <= ref_len;
			const bool last = ri == ref_len;
			const bool more = rowon && !last;
			const int r = rnext;
			rnext = more ? (int)ref[ri] : 0;
			const bool rN = r == MV_N;
			const int mr = rN ? 31 : r + 1;
			const int nb = more ? sbeg(ri + 1) : 0, ne = more ? send(ri + 1) : 0;
			const int o_r = (last && P.right_free) ? 0 : P.gap_open, x_r = (last && P.right_free) ? 0 : P.ext;
			const int ep = x_r < o_r ? x_r : o_r;
			const int32_t s0 = P.left_free ? 0 : -(P.gap_open + (ri - 1) * P.ext);
			const int32_t *prevS = bufS + ((ri - 1) & 1) * RN;
			int32_t *curS = bufS + (ri & 1) * RN;
			const int q = b + sl;
			const bool on = rowon && q < e;
			const int qc = qc_pref;
			{ const int qn = nb + sl; qc_pref = (more && qn >= 1 && qn < ne) ? (int)qry[qn - 1] : 0; }
			const int32_t Sd = prevS[(q - 1) & M], Su = prevS[q & M], QG = qg[q & M];
			const bool q0 = q == 0;
			const bool inner = !last && q < qlen;
			const bool diag_ok = q > pb && q - 1 < pe;
			const bool up_ok = !q0 && q < pe;
			const int mq = qc == MV_N ? 31 : qc + 1;
			const int sc = (qc == MV_N || rN) ? P.match - 1 : ((mq & mr) ? P.match : -P.mismatch);
			int32_t score = diag_ok ? Sd + sc : MV_NO_ALIGN;
			int origin = diag_ok ? MV_MATCH : 0;
			const bool fr = q == qlen && P.right_free;
			const int32_t qe = QG - (fr ? 0 : P.ext), qo = Su - (fr ? 0 : P.gap_open);
			const bool extq = up_ok && qe >= qo && q < ppe;
			const int32_t tq = extq ? qe : qo;
			int32_t Ht = (up_ok && tq > score) ? tq : score;
			Ht = q0 ? s0 : Ht;
			const int32_t A = on ? Ht + ep * (q - b) : INT32_MIN;
			const int32_t Pin = mv_seg_prefix_max<SEG>(A);
			int32_t excl = wave_shr1(Pin, INT32_MIN);
			excl = sl == 0 ? INT32_MIN : excl;
			const int32_t G = (q > b && sl > 0) ? excl - o_r - ep * (q - 1 - b) : 0;
			const bool refg = q > b && score - la < G;
			score = refg ? G : score; origin = refg ? MV_REF_GAP_MATRIX : origin;
			const bool qryg = up_ok && score - la < tq;
			score = qryg ? tq : score; origin = qryg ? MV_QRY_GAP_MATRIX : origin;
			score = q0 ? s0 : score; origin = q0 ? MV_QRY_GAP_MATRIX : origin;
			const int32_t Sl = wave_shr1(score, 0), Gl = wave_shr1(G, 0);
			int tmp_path = q0 ? MV_QRY_GAP_EXTEND : ((inner && (!diag_ok || q <= b || !up_ok)) ? MV_BOUNDARY : 0) + (extq ? MV_QRY_GAP_EXTEND : 0);
			tmp_path += (q > b + 1 && Gl - x_r >= Sl - o_r) ? MV_REF_GAP_EXTEND : 0;
			if (on) {
				slab[(size_t)(ri - 1) * (size_t)pitch + (size_t)(q - b)] = (uint8_t)(tmp_path + origin);
				curS[q & M] = score;
				qg[q & M] = up_ok ? tq : MV_NO_ALIGN;
			}
			if (on && last && q == qlen) { fin = score; hasfin = true; }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			if (rowon) { ppe = pe; pb = b; pe = e; b = nb; e = ne; }
		}
		mv_fence();
		const unsigned long long finmask = __ballot(hasfin);
		for (int s = 0; s < nj; ++s) {
			const MvJob Js = jobs[j0 + s];
			MvOut O; memset(&O, 0, sizeof(O)); O.attempts = (int32_t)Js.attempt;
			O.status = rl(status_l, s * SEG);
			if (O.status == 0) {
				const unsigned long long m = finmask & (((SEG == 32 ? 0xffffffffULL : 0xffffULL)) << (s * SEG));
				O.score = m ? rl(fin, __ffsll((long long)m) - 1) : 0;
				mv_finish(codes + Js.ref_off, codes + Js.qry_off, (int)Js.ref_len, (int)Js.qry_len, (long long)Js.ms, (long long)Js.bw, (int)Js.attempt,
				          slabs + ((size_t)blockIdx.x * NJ + (size_t)s) * slab_bytes, P, lane, O, cur, cap, subs, dels, inss, ins_seq);
			}
			if (lane == 0) out[j0 + s] = O;
		}
	}
}

static inline size_t mv_slab_need(const MvJob &J)
{
	const long long p = std::min<long long>(2LL * J.bw + 2, (long long)J.qry_len + 2);
	const size_t path = J.ref_len > 0 ? (size_t)(J.ref_len - 1) * (size_t)p + J.qry_len + 2 : 0;
	return ((path + 15) & ~(size_t)15) + 4 * ((size_t)J.ref_len + J.qry_len + 4);
}
static inline long long mv_ring_cols(const MvJob &J) { return std::min<long long>(2LL * J.bw + 2, (long long)J.qry_len + 2); }
// 16 / 32: every row of the job fits a segment of that many lanes (k_mapvar_packed); 0: one job per wave
static inline int mv_pack_seg(const MvJob &J)
{
	static const bool off = getenv("PGA_MAPVAR_NO_PACK") != nullptr;
	if (off || J.ref_len < 1 || J.bw > 15 || std::llabs((long long)J.ms) > (1LL << 28)) return 0;
	long long lb = (long long)J.ref_len - J.ms - J.bw;
	lb = lb < 0 ? 0 : lb > (long long)J.qry_len ? (long long)J.qry_len : lb;
	const long long need = std::max<long long>(2LL * J.bw + 1, (long long)J.qry_len + 1 - lb);     // the widest row: a band row or the last one
	return need <= 16 ? 16 : need <= 32 ? 32 : 0;
}
static inline int mv_ring_class(const MvJob &J) { const int ps = mv_pack_seg(J); if (ps) return ps; const long long w = mv_ring_cols(J); return w <= 128 ? 128 : w <= 512 ? 512 : w <= 2048 ? 2048 : 0; }

static void mv_run_rounds(std::vector<MvJob> &pending, const uint8_t *d_codes_p, const pga_mapvar_params_t &prm, pga_mapvar_res_t *res,
                          std::vector<pga_sub_t> &h_subs, std::vector<pga_del_t> &h_dels, std::vector<pga_ins_t> &h_inss, std::vector<char> &h_seq, hipStream_t st);

void map_variations_host(int64_t n, const pga_mapvar_job_t *jobs, const pga_mapvar_params_t &prm, pga_mapvar_res_t *res,
                         std::vector<pga_sub_t> &h_subs, std::vector<pga_del_t> &h_dels, std::vector<pga_ins_t> &h_inss, std::vector<char> &h_seq)
{
	const bool verbose = getenv("PGA_VERBOSE") != nullptr;
	if (prm.penalty_gap_open < 0 || prm.penalty_gap_extend < 0 || prm.score_match < 0 || prm.penalty_mismatch < 0) throw std::runtime_error("pga_map_variations: negative score parameter");
	if (prm.max_alignment_attempts < 1) throw std::runtime_error("pga_map_variations: max_alignment_attempts must be at least 1");
	hipStream_t st = 0;
	// sequences: one copy per distinct (pointer, length) -- the members of a block share the anchor consensus
	// (offsets first, then the distinct sequences are copied into ONE pinned staging buffer by a few host threads and leave in one DMA:
	// growing a std::vector by 200 MB and a pageable copy cost more than the kernels)
	struct Placed { uint32_t len; uint64_t off; };
	std::unordered_map<const char*, Placed> seen;
	struct Piece { const char *p; uint32_t len; uint64_t off; };
	std::vector<Piece> pieces;
	uint64_t cat_size = 0;
	auto place = [&](const char *p, uint32_t len) -> uint64_t {
		auto it = seen.find(p);
		if (it != seen.end() && it->second.len == len) return it->second.off;
		const uint64_t off = cat_size;
		pieces.push_back(Piece{p, len, off});
		cat_size += len;
		seen[p] = Placed{len, off};
		return off;
	};
	std::vector<MvJob> pending((size_t)n);
	for (int64_t i = 0; i < n; ++i) {
		const pga_mapvar_job_t &j = jobs[i];
		if ((j.ref_len && !j.ref) || (j.qry_len && !j.qry)) throw std::runtime_error("pga_map_variations: null sequence");
		if (j.ref_len >= (1u << 30) || j.qry_len >= (1u << 30)) throw std::runtime_error("pga_map_variations: sequence longer than 2^30");
		if ((uint64_t)std::min(prm.penalty_gap_extend, prm.penalty_gap_open) * ((uint64_t)j.qry_len + 1) >= (1ULL << 29)) throw std::runtime_error("pga_map_variations: gap extension penalty times query length overflows the score type");
		MvJob &J = pending[(size_t)i];
		J.ref_off = place(j.ref, j.ref_len); J.qry_off = place(j.qry, j.qry_len);
		J.ref_len = j.ref_len; J.qry_len = j.qry_len; J.ms = j.mean_shift;
		const uint64_t cap = (uint64_t)j.ref_len + j.qry_len + (uint64_t)std::llabs((long long)j.mean_shift) + 2;     // a wider band has the same stripes
		J.bw = (uint32_t)std::min<uint64_t>((uint64_t)j.band_width + (uint64_t)std::max(prm.extra_band_width, 0), cap);   // map_variations.rs:51
		J.attempt = 1; J.orig = (uint32_t)i;
	}
	DBuf<char> d_ascii; DBuf<uint8_t> d_codes;
	PinVec<char> stage;
	stage.resize(cat_size + 1);
	{
		const int nt = cat_size > (8u << 20) ? 4 : 1;
		std::vector<std::thread> th;
		for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() { for (size_t i = (size_t)t; i < pieces.size(); i += (size_t)nt) memcpy(stage.data() + pieces[i].off, pieces[i].p, pieces[i].len); });
		for (auto &x : th) x.join();
	}
	d_ascii.upload(stage.data(), cat_size, st);
	d_codes.alloc(cat_size + 64);
	if (cat_size) k_mv_encode<<<(unsigned)std::min<size_t>((cat_size + 255) / 256, 65535), 256, 0, st>>>(d_ascii.p, cat_size, d_codes.p);
	mv_run_rounds(pending, d_codes.p, prm, res, h_subs, h_dels, h_inss, h_seq, st);
}

// The same with the letters already in device memory (reconsensus builds the new consensus and the member sequences there): jobs address
// d_ascii[ref_off .. + ref_len) / d_ascii[qry_off .. + qry_len); band_width is the caller's (extra_band_width is added here as above).
void map_variations_dev(int64_t n, const MvDevJob *jobs, const char *d_ascii, uint64_t cat_size, const pga_mapvar_params_t &prm, pga_mapvar_res_t *res,
                        std::vector<pga_sub_t> &h_subs, std::vector<pga_del_t> &h_dels, std::vector<pga_ins_t> &h_inss, std::vector<char> &h_seq, hipStream_t st)
{
	if (prm.penalty_gap_open < 0 || prm.penalty_gap_extend < 0 || prm.score_match < 0 || prm.penalty_mismatch < 0) throw std::runtime_error("pga_map_variations: negative score parameter");
	if (prm.max_alignment_attempts < 1) throw std::runtime_error("pga_map_variations: max_alignment_attempts must be at least 1");
	std::vector<MvJob> pending((size_t)n);
	for (int64_t i = 0; i < n; ++i) {
		const MvDevJob &j = jobs[i];
		if (j.ref_len >= (1u << 30) || j.qry_len >= (1u << 30)) throw std::runtime_error("pga_map_variations: sequence longer than 2^30");
		if ((uint64_t)std::min(prm.penalty_gap_extend, prm.penalty_gap_open) * ((uint64_t)j.qry_len + 1) >= (1ULL << 29)) throw std::runtime_error("pga_map_variations: gap extension penalty times query length overflows the score type");
		MvJob &J = pending[(size_t)i];
		J.ref_off = j.ref_off; J.qry_off = j.qry_off; J.ref_len = j.ref_len; J.qry_len = j.qry_len; J.ms = j.mean_shift;
		const uint64_t cap = (uint64_t)j.ref_len + j.qry_len + (uint64_t)std::llabs((long long)j.mean_shift) + 2;
		J.bw = (uint32_t)std::min<uint64_t>((uint64_t)j.band_width + (uint64_t)std::max(prm.extra_band_width, 0), cap);
		J.attempt = 1; J.orig = (uint32_t)i;
	}
	DBuf<uint8_t> d_codes(cat_size + 64);
	if (cat_size) k_mv_encode<<<(unsigned)std::min<size_t>((cat_size + 255) / 256, 65535), 256, 0, st>>>(d_ascii, cat_size, d_codes.p);
	mv_run_rounds(pending, d_codes.p, prm, res, h_subs, h_dels, h_inss, h_seq, st);
}

static void mv_run_rounds(std::vector<MvJob> &pending, const uint8_t *d_codes_p, const pga_mapvar_params_t &prm, pga_mapvar_res_t *res,
                          std::vector<pga_sub_t> &h_subs, std::vector<pga_del_t> &h_dels, std::vector<pga_ins_t> &h_inss, std::vector<char> &h_seq, hipStream_t st)
{
	const bool verbose = getenv("PGA_VERBOSE") != nullptr;
	const MvParams P{prm.score_match, prm.penalty_mismatch, prm.penalty_gap_open, prm.penalty_gap_extend, prm.left_terminal_gaps_free != 0, prm.right_terminal_gaps_free != 0,
	                 prm.gap_align_left != 0, prm.min_length, prm.max_alignment_attempts};
	const char *eb = getenv("PGA_MAPVAR_SLAB_GB");
	const size_t budget = (size_t)((eb ? atof(eb) : 16.0) * (double)(1ULL << 30));
	h_subs.clear(); h_dels.clear(); h_inss.clear(); h_seq.clear();
	int round = 0;
	bool overflowed = false;
	while (!pending.empty()) {
		++round;
		// big slabs first; one launch per (ring class, need within a factor of four)
		std::stable_sort(pending.begin(), pending.end(), [](const MvJob &a, const MvJob &b) { const int ca = mv_ring_class(a), cb = mv_ring_class(b); if (ca != cb) return (ca == 0 ? 1 << 30 : ca) > (cb == 0 ? 1 << 30 : cb); return mv_slab_need(a) > mv_slab_need(b); });
		const auto t_round = std::chrono::steady_clock::now();
		DBuf<MvJob> d_jobs; d_jobs.upload(pending, st);
		DBuf<MvOut> d_out(pending.size());
		// output pools: the worst case (every base a substitution, every other base an insertion) would be 40 bytes per base; the first
		// round reserves an eighth of it, jobs that overflow run again in a round whose pools hold their worst case
		uint64_t cap_subs = 0, cap_dels = 0, cap_inss = 0, cap_ib = 0;
		for (const MvJob &J : pending) { const uint64_t m = std::min(J.ref_len, J.qry_len); cap_subs += m; cap_dels += (uint64_t)J.ref_len / 2 + 3; cap_inss += std::min<uint64_t>(J.qry_len, (uint64_t)J.ref_len + 1) + 1; cap_ib += J.qry_len; }
		if (!overflowed) {
			if (getenv("PGA_MAPVAR_TIGHT_POOLS")) { cap_subs = cap_subs / 64 + 8; cap_dels = cap_dels / 64 + 8; cap_inss = cap_inss / 64 + 8; cap_ib = cap_ib / 64 + 8; }   // tests: force the overflow round
			else { cap_subs = cap_subs / 8 + 4096 + 4 * pending.size(); cap_dels = cap_dels / 16 + 4096 + 4 * pending.size(); cap_inss = cap_inss / 16 + 4096 + 4 * pending.size(); cap_ib = cap_ib / 8 + 65536; }
		}
		if ((cap_subs + cap_dels) * 8 + cap_inss * 16 + cap_ib > (64ULL << 30)) throw std::runtime_error("pga_map_variations: more than 64 GB of output pools in one call; split the batch");
		DBuf<pga_sub_t> d_subs(cap_subs + 1); DBuf<pga_del_t> d_dels(cap_dels + 1); DBuf<pga_ins_t> d_inss(cap_inss + 1); DBuf<char> d_seq(cap_ib + 1);
		const MvCaps caps{cap_subs, cap_dels, cap_inss, cap_ib};
		DBuf<MvCursors> d_cur(1); d_cur.zero(st);
		std::vector<DBuf<uint8_t>> keep_slabs; std::vector<DBuf<int32_t>> keep_rings; std::vector<DBuf<uint32_t>> keep_ctr;
		size_t s0 = 0;
		while (s0 < pending.size()) {
			const int cls = mv_ring_class(pending[s0]);
			const size_t need_max = mv_slab_need(pending[s0]);
			size_t s1 = s0 + 1;
			long long cols_max = mv_ring_cols(pending[s0]);
			while (s1 < pending.size() && mv_ring_class(pending[s1]) == cls && mv_slab_need(pending[s1]) * 4 >= need_max) { cols_max = std::max(cols_max, mv_ring_cols(pending[s1])); ++s1; }
			const size_t nj = s1 - s0;
			const size_t slab_bytes = (need_max + 255) & ~(size_t)255;
			const int pack = cls == 16 || cls == 32 ? 64 / cls : 1;                                   // jobs per wave
			int ring_n = 0;
			if (cls == 0) { ring_n = 4096; while ((long long)ring_n < cols_max) ring_n <<= 1; }
			const size_t per_slot = slab_bytes * (size_t)pack + (cls == 0 ? (size_t)ring_n * 12 : 0);
			size_t n_slots = std::min<size_t>((nj + (size_t)pack - 1) / (size_t)pack, std::max<size_t>(1, budget / per_slot));
			n_slots = std::min<size_t>(n_slots, cls == 2048 ? 1536 : 8192);
			keep_slabs.emplace_back(n_slots * slab_bytes * (size_t)pack);
			keep_ctr.emplace_back(1); keep_ctr.back().zero(st);
			int32_t *gr = nullptr;
			if (cls == 0) { keep_rings.emplace_back(n_slots * (size_t)ring_n * 3); gr = keep_rings.back().p; }
			if (verbose) fprintf(stderr, "[pga]   map_variations round %d: %zu jobs, %s %d, slab %.1f KB x %zu waves\n", round, nj, pack > 1 ? "segments of" : "ring", cls ? cls : ring_n, slab_bytes / 1024.0, n_slots);
#define MV_LAUNCH(R) k_mapvar<R><<<(unsigned)n_slots, 64, 0, st>>>(d_jobs.p + s0, (int)nj, P, d_codes_p, keep_ctr.back().p, keep_slabs.back().p, slab_bytes, gr, ring_n, d_out.p + s0, d_cur.p, caps, d_subs.p, d_dels.p, d_inss.p, d_seq.p)
#define MV_LAUNCH_P(SG) k_mapvar_packed<SG><<<(unsigned)n_slots, 64, 0, st>>>(d_jobs.p + s0, (int)nj, P, d_codes_p, keep_ctr.back().p, keep_slabs.back().p, slab_bytes, d_out.p + s0, d_cur.p, caps, d_subs.p, d_dels.p, d_inss.p, d_seq.p)
			if (cls == 16) MV_LAUNCH_P(16); else if (cls == 32) MV_LAUNCH_P(32);
			else if (cls == 128) MV_LAUNCH(128); else if (cls == 512) MV_LAUNCH(512); else if (cls == 2048) MV_LAUNCH(2048); else MV_LAUNCH(0);
#undef MV_LAUNCH
#undef MV_LAUNCH_P
			PGA_HIP(hipGetLastError());
			s0 = s1;
		}
		std::vector<MvOut> ho = d_out.download(st);
		const std::vector<MvCursors> hc = d_cur.download(st);
		const size_t b_subs = h_subs.size(), b_dels = h_dels.size(), b_inss = h_inss.size(), b_seq = h_seq.size();
		std::vector<MvCursors> hcm = hc;                                    // the cursors run past the pools when a job overflowed
		hcm[0].subs = std::min<unsigned long long>(hcm[0].subs, cap_subs); hcm[0].dels = std::min<unsigned long long>(hcm[0].dels, cap_dels);
		hcm[0].inss = std::min<unsigned long long>(hcm[0].inss, cap_inss); hcm[0].ib = std::min<unsigned long long>(hcm[0].ib, cap_ib);
		h_subs.resize(b_subs + hcm[0].subs); h_dels.resize(b_dels + hcm[0].dels); h_inss.resize(b_inss + hcm[0].inss); h_seq.resize(b_seq + hcm[0].ib);
		if (hcm[0].subs) PGA_HIP(hipMemcpyAsync(h_subs.data() + b_subs, d_subs.p, hcm[0].subs * sizeof(pga_sub_t), hipMemcpyDeviceToHost, st));
		if (hcm[0].dels) PGA_HIP(hipMemcpyAsync(h_dels.data() + b_dels, d_dels.p, hcm[0].dels * sizeof(pga_del_t), hipMemcpyDeviceToHost, st));
		if (hcm[0].inss) PGA_HIP(hipMemcpyAsync(h_inss.data() + b_inss, d_inss.p, hcm[0].inss * sizeof(pga_ins_t), hipMemcpyDeviceToHost, st));
		if (hcm[0].ib) PGA_HIP(hipMemcpyAsync(h_seq.data() + b_seq, d_seq.p, hcm[0].ib, hipMemcpyDeviceToHost, st));
		PGA_HIP(sync_stream(st));
		for (size_t k = b_inss; k < h_inss.size(); ++k) h_inss[k].seq_off += b_seq;
		std::vector<MvJob> next;
		size_t n_over = 0;
		for (size_t k = 0; k < pending.size(); ++k) {
			const MvOut &o = ho[k]; const MvJob &J = pending[k];
			if (o.status == 4) { next.push_back(J); n_over++; continue; }                                 // the output pools were full
			if (o.status == 0 && o.hit && (int)J.attempt < P.max_attempts) {                          // align.rs:55-62
				MvJob N = J;
				const uint64_t a = (uint64_t)std::llabs((long long)J.ms);
				const uint64_t cap = (uint64_t)J.ref_len + J.qry_len + a + 2;
				N.bw = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(2ULL * J.bw, std::max<uint64_t>(1, a)), cap);
				N.attempt = J.attempt + 1;
				next.push_back(N);
				continue;
			}
			pga_mapvar_res_t &R = res[J.orig];
			R.status = o.status; R.score = o.score; R.attempts = o.attempts; R.hit_boundary = o.hit;
			R.n_subs = o.n_subs; R.n_dels = o.n_dels; R.n_inss = o.n_inss; R.n_ins_bases = o.n_ib;
			R.sub_off = b_subs + o.sub_off; R.del_off = b_dels + o.del_off; R.ins_off = b_inss + o.ins_off;
		}
		if (verbose) fprintf(stderr, "[pga]   map_variations round %d: %zu of %zu jobs hit the band boundary and go again; %.1f ms (kernels + download of %.1f MB of edits)\n", round, next.size(), pending.size(),
		                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_round).count(), (hcm[0].subs * 8 + hcm[0].dels * 8 + hcm[0].inss * 16 + hcm[0].ib) / 1e6);
		overflowed = n_over > 0;
		if (verbose && n_over) fprintf(stderr, "[pga]   map_variations round %d: %zu jobs did not fit the output pools and run again\n", round, n_over);
		pending.swap(next);
	}
}

} // namespace pga
