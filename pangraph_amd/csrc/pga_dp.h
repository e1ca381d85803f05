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

struct DpParams { int32_t q, e, q2, e2, sc_mch, sc_mis, sc_ambi; }; // sc_* are matrix entries mat[0], mat[1], mat[24]

size_t dp_slab_bytes(int qlen, int tlen, int w);
void dp_run(PkBases d_bases, const std::vector<DpJob> &jobs, const DpParams &P, std::vector<DpRes> &res, PinVec<uint32_t> &cigars, hipStream_t st, Timers *tm = nullptr);

} // namespace pga
