// pga_maxocc_hist.h -- mm_idx_cal_max_occ of every group of a batch WITHOUT a sort (the default since round 6; PGA_MAXOCC_HIST=0: the sort of all keys).
//
// mm_idx_cal_max_occ (packages/minimap2-sys/minimap2/index.c:186-207) is an order statistic: of the occurrence counts of a group's n distinct
// minimizers, the (uint32)((1 - f) * n)-th smallest, plus one -- with f = 2e-4 one of the few largest counts.  The sort route
// (index_cal_max_occ, pga_index.hip) sorts group << cbits | count over all keys of the batch: 3.5 radix passes, ~14 dispatches, 172 times per build step.
// Counts are small numbers (1 or 2 for almost every k-mer, dozens for repeats), so a histogram per group answers the question exactly:
//   k_mo_hist     one pass over the keys: count = occ_off[i + 1] - occ_off[i] into bin min(count, MO_BINS - 1) of its group's histogram (keys arrive
//                 group by group: a workgroup aggregates the bins of its first key's group in LDS, the few keys of other groups go to global atomics)
//   k_mo_select   one workgroup per group: n = the histogram's total, the bin in which the cumulative count passes kk = (uint32)((1 - f) * n);
//                 the LAST bin collects every count >= MO_BINS - 1: if the answer lies there the group reports -1 and the caller takes the sort route.
// One fill + two launches.
//
// STATUS: written in round 5 without a device and checked under dev/emu/hip_emu.h against the sorted counts at three fractions (tests/test_routes_emu.py);
// round 6: on the device every one of the 1998 calls of the BASELINE build keeps its digest with it, tests/test_gpu_zz_candidates.py compares mid_occ itself
// against the sort route; alone it leaves a build step where it was (2 078 against 2 073 / 2 078 ms) and takes ~2.7 k dispatches out of it.
#pragma once
#ifndef PGA_EMU
#include "pga_common.h"
#endif

namespace pga {

constexpr uint32_t MO_BINS = 1024;        // bin b < MO_BINS - 1: count == b; the last bin: count >= MO_BINS - 1
constexpr uint32_t MO_NT = 256;
constexpr uint32_t MO_KEYS = 8 * MO_NT;   // keys per workgroup of k_mo_hist

__global__ __launch_bounds__(256)
void k_mo_hist(const uint32_t *__restrict__ occ_off, const uint32_t *__restrict__ key_grp, uint32_t n_keys, uint32_t *__restrict__ hist /* n_grp x MO_BINS, zeroed */)
{
	__shared__ uint32_t h[MO_BINS];
	const uint32_t tid = threadIdx.x, k0 = blockIdx.x * MO_KEYS, k1 = k0 + MO_KEYS < n_keys ? k0 + MO_KEYS : n_keys;
	const uint32_t g0 = key_grp[k0];
	for (uint32_t b = tid; b < MO_BINS; b += MO_NT) h[b] = 0;
	__syncthreads();
	for (uint32_t i = k0 + tid; i < k1; i += MO_NT) {
		const uint32_t c = occ_off[i + 1] - occ_off[i], g = key_grp[i];
		const uint32_t b = c < MO_BINS - 1 ? c : MO_BINS - 1;
		if (g == g0) atomicAdd(&h[b], 1u); else atomicAdd(&hist[(size_t)g * MO_BINS + b], 1u);
	}
	__syncthreads();
	for (uint32_t b = tid; b < MO_BINS; b += MO_NT) { const uint32_t c = h[b]; if (c) atomicAdd(&hist[(size_t)g0 * MO_BINS + b], c); }
}

// out[g] = index.c:186-207 for group g, or -1 when the answer is a count the histogram does not resolve
__global__ __launch_bounds__(256)
void k_mo_select(const uint32_t *__restrict__ hist, int n_grp, float f, int32_t *__restrict__ out)
{
	__shared__ uint32_t sc[MO_NT];
	const uint32_t tid = threadIdx.x, g = blockIdx.x;
	constexpr uint32_t PER = MO_BINS / MO_NT;
	const uint32_t *H = hist + (size_t)g * MO_BINS;
	uint32_t loc[PER], s = 0;
	for (uint32_t j = 0; j < PER; ++j) { loc[j] = H[tid * PER + j]; s += loc[j]; }
	sc[tid] = s;
	__syncthreads();
	for (uint32_t d = 1; d < MO_NT; d <<= 1) {
		const uint32_t v = tid >= d ? sc[tid - d] : 0;
		__syncthreads();
		sc[tid] += v;
		__syncthreads();
	}
	const uint32_t n = sc[MO_NT - 1];
	if (n == 0) { if (tid == 0) out[g] = 1; return; }                       // (uniform: every thread sees the same total)
	const uint32_t kk = (uint32_t)((1. - (double)f) * (double)n);           // index.c:204: double arithmetic on the float fraction; kk < n
	uint32_t run = sc[tid] - s;                                             // keys in the bins before this thread's
	for (uint32_t j = 0; j < PER; ++j) {
		if (run <= kk && kk < run + loc[j]) {                                 // exactly one (thread, bin) holds the kk-th smallest count
			const uint32_t b = tid * PER + j;
			out[g] = b == MO_BINS - 1 ? -1 : (int32_t)(b + 1u);
		}
		run += loc[j];
	}
}

} // namespace pga
