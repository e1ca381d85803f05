// pga_post.h -- batched device evaluation of the base-reading steps around the DP kernels (pga_post.hip).
#pragma once
#include "pga_common.h"
#include "pga_dp.h"

namespace pga {

// windows are named like DP problems (pga_dp.h): offsets into the resident base array, the query by its sequence start, full
// length, window start on the aligned strand and strand
struct PostProbe { uint64_t t_off, q_off; int32_t qlen_full, qs, n; int32_t q_rev; };                 // equally long windows of n bases
struct PostWalk { uint64_t t_off, q_off; int32_t qlen_full, qs, q_rev; uint32_t n_cigar; uint64_t cig_off; };
struct PostWalkRes { int32_t max_zdrop, t0, t1, q0, q1; };                                              // window of the worst drop: target [t0,t1], query [q0,q1] (align.c:40-43)
struct PostFin { uint64_t t_off, q_off; int32_t qlen_full, q_start, q_rev; uint32_t n_cigar; uint64_t cig_off; };
struct PostFinRes { uint32_t n_cigar; int32_t qshift, tshift, blen, mlen, n_ambi, dp_max, n_gapo, n_gap, q_span, t_span; };

// out[i] = number of mismatches of probe i, or -1 if the windows hold an ambiguous base or differ in more than m_max positions
void post_identity(PkBases d_bases, const PinVec<PostProbe> &probes, int m_max, PinVec<int32_t> &out, hipStream_t st);
void post_zdrop_walk(PkBases d_bases, const std::vector<PostWalk> &reqs, const std::vector<uint32_t> &cig, const DpParams &P, std::vector<PostWalkRes> &out, hipStream_t st);
// cig holds the operation lists of all requests back to back; on return request i's final list is cig[cig_off .. cig_off + out[i].n_cigar)
void post_cigar_finish(PkBases d_bases, const std::vector<PostFin> &reqs, PinVec<uint32_t> &cig, const DpParams &P, std::vector<PostFinRes> &out, hipStream_t st);

} // namespace pga
