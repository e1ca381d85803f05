// pga_dp.h -- DP problem descriptors exchanged between the alignment driver (host) and kernel #5.
#pragma once
#include "pga_common.h"

namespace pga {

// One ksw_extd2 call (reference: align.c:316-344 mm_align_pair).  Sequences are NOT copied: a problem names
// windows of the resident packed store (PkBases); reversal (left extension, align.c:711-713) and reverse-complement
// (align.c:970-975) are index transforms applied by the kernel when it reads a base.
struct DpJob {
	uint64_t t_off;      // offset of the target window's first base (forward coordinates)
	uint64_t q_off;      // offset of the query SEQUENCE's first base
	int32_t qlen_full;   // whole query length
	int32_t qs;          // window start on the query strand being aligned
	int32_t qlen, tlen;  // window lengths
	int32_t w, zdrop, end_bonus, flag;
	uint8_t q_rev;       // 1: the window lies on the reverse complement of the query
	uint8_t seq_rev;     // 1: both windows are read back to front
	uint8_t pad[2];
};

struct DpRes {           // ksw_extz_t (ksw2.h:31-40)
	int32_t max, max_q, max_t, mqe, mqe_t, mte, mte_q, score;
	int32_t zdropped, reach_end, n_cigar, pad;
	uint64_t cigar_off;  // into the CIGAR pool returned with the batch
};

// DpJob.flag bit (not a ksw2 flag): the job is a local-alignment score query (ksw_ll_i16), answered by pga_ll.hip;
// the result comes back as score / max_q (qe) / max_t (te)
#define PGA_JOB_LL 0x8000
#define PGA_LL_MAX_LEN 10240     // longest query (padded to 8) / target the LL kernel holds in LDS

struct DpParams { int32_t q, e, q2, e2, sc_mch, sc_mis, sc_ambi; int32_t lb_mode; }; // sc_* are matrix entries mat[0], mat[1], mat[24]; lb_mode: 0 = the length-bound stop
                                                                                          // (below) is on, 1 = off, 2 = checked (the sweep goes on and the outcome is compared)

// The LENGTH-BOUND STOP of an extension whose target window is the few bases left before a block end (tlen <= 64; the bounds below close up to
// tlen ~ 50 with the asm10 scores) while the query runs on for
// kilobases: the reference sweeps ~w + 2 tlen diagonals until the band has slid past the last target column (ksw2_extd2_sse.c:172: st > en ->
// zdropped), because the cells it still visits are pure gap cells whose drop grows as fast as the threshold (ksw2.h:178).  None of those
// cells can change the answer, and that is decidable after ~3 tlen diagonals:
//   * until the band's left edge has moved sixteen columns (r <= w + 30) every column of [0, T) is computed on every diagonal from computed
//     neighbours, i.e. H(t, j) is the true score of the best alignment of t[0..t] with q[0..j] that starts at the origin; such an alignment has at
//     most t + 1 <= tlen match columns and at least j - t gap bases in the query direction, which cost at least g(j - t) = min(q + e L, q2 + e2 L)
//     in one piece (more in several): H(t, j) <= a tlen - g(j - t), and on every diagonal r' > r >= 2 tlen the cells of the range have
//     j - t >= r + 3 - 2 tlen;
//   * on the last diagonals (tlen > 16: at most 2 tlen - 32 of them run with the fresh-edge values of :176-183) a tracked H grows by one u or v
//     per diagonal, and those stay inside the difference bounds -(q + e) ... a + q + e whatever the edge supplies (each bound follows from
//     x, y >= -(q + e) and the bounds of the inputs alone), so H <= a tlen - g(w + 32 - 2 tlen) + (a + q + e)(2 tlen - 32);
//   * the query is long enough (qlen >= w + 2 tlen) that the range runs empty before the last query row: the sweep ends with zdropped = 1 (by the
//     z-drop test or by the empty range: the same record), mqe and score are never touched, and max / mte only change for an H above them.
// So once both bounds are <= min(ez.max, ez.mte) the record is final.  On the BASELINE build 55 % of all diagonals of the banded classes are of
// this kind (64 k problems per step, 1 540 diagonals each, ~150 with the stop).
struct LbStop { int32_t on, r_lo, r_hi, tail; };
__host__ __device__ inline int32_t lb_gap(int q, int e, int q2, int e2, int L) { const int a = q + e * L, b = q2 + e2 * L; return a < b ? a : b; }
__host__ __device__ inline LbStop lb_stop_of(int qlen, int tlen, int w, int flag, int q, int e, int q2, int e2, int sc_mch, int sc_mis, int sc_N, int lb_mode)
{
	LbStop S; S.on = 0; S.r_lo = S.r_hi = S.tail = 0;
	if (lb_mode == 1 || (flag & (0x08 | 0x8000)) || tlen > 64 || tlen < 1 || w < 64 || qlen < w + 2 * tlen) return S;
	if (sc_mch < 0 || sc_mis > sc_mch || sc_N > sc_mch || q < 0 || e < 0 || q2 < 0 || e2 < 0) return S;
	S.on = 1; S.r_lo = 2 * tlen; S.r_hi = w + 30;
	const int qe1 = q + e < q2 + e2 ? q + e : q2 + e2;
	S.tail = tlen > 16 ? sc_mch * tlen - lb_gap(q, e, q2, e2, w + 32 - 2 * tlen) + (sc_mch + qe1) * (2 * tlen - 32) : (int32_t)0x80000000;
	return S;
}
// after diagonal r: is the record final?  (m = min(ez.max, ez.mte))
__host__ __device__ inline bool lb_final(const LbStop &S, int r, int tlen, int q, int e, int q2, int e2, int sc_mch, int m)
{
	return S.on && r >= S.r_lo && r <= S.r_hi && sc_mch * tlen - lb_gap(q, e, q2, e2, r + 3 - 2 * tlen) <= m && S.tail <= m;
}

int dp_lb_mode();          // PGA_LB=off: 1, PGA_LB=check: 2 (read on every call: tests switch it inside one process), else 0
size_t dp_slab_bytes(int qlen, int tlen, int w);
void dp_run(PkBases d_bases, const std::vector<DpJob> &jobs, const DpParams &P, std::vector<DpRes> &res, PinVec<uint32_t> &cigars, hipStream_t st, Timers *tm = nullptr);

} // namespace pga
