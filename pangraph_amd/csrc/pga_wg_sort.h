// pga_wg_sort.h -- a stable sort of up to WGS_CAP (key, value) pairs in ONE launch of one workgroup (the default for the chaining stage's small sorts since round 6; PGA_WG_SORT=0: rocPRIM).
//
// rocPRIM sorts 1 k ... 1 M elements as a block sort + up to ten merge passes of two kernels each: 6.2 k merge kernels and 0.9 k block sorts per build
// step (profiles/r05_f_c5_kernel_stats.csv), most of them for the small sorts of the chaining stage of the calls above the leaf level (segment
// lengths, chain candidates of a few hundred to a few thousand anchors) -- each a chain of 3-9 dependent launches of ~19 us on a call's critical path.
// Up to WGS_CAP pairs fit LDS: a bitonic network over (key bits [0, end_bit), original index) is a stable sort by those bits, the order
// rocprim::radix_sort_pairs(..., 0, end_bit) gives.
//
// STATUS: written in round 5 without a device and checked under dev/emu/hip_emu.h against std::stable_sort (tests/test_routes_emu.py); round 6: on the device
// every one of the 1998 calls of the BASELINE build keeps its digest with it (bench.py's parity check of the timed step), tests/test_gpu_zz_candidates.py holds it
// against the rocPRIM route, and together with pga_maxocc_hist.h it is worth ~5 % of a build step (ABAB, medians of six steps: 1 965 against 2 079 / 2 197 ms).
#pragma once
#ifndef PGA_EMU
#include "pga_common.h"
#endif

namespace pga {

constexpr uint32_t WGS_CAP = 4096;
constexpr uint32_t WGS_NT = 1024;

template <typename K>
__global__ __launch_bounds__(1024)
void k_wg_sort_pairs(const K *__restrict__ kin, K *__restrict__ kout, const uint32_t *__restrict__ vin, uint32_t *__restrict__ vout, uint32_t n, int end_bit)
{
	__shared__ uint64_t key[WGS_CAP];          // the key bits the sort compares
	__shared__ uint16_t idx[WGS_CAP];          // original position: the tie-break that makes the network stable, and where the full key and the value come from
	const uint32_t tid = threadIdx.x;
	if (n == 0 || n > WGS_CAP) return;
	const uint64_t mask = end_bit >= 64 ? ~0ULL : ((1ULL << end_bit) - 1);
	uint32_t P = 2; while (P < n) P <<= 1;
	for (uint32_t j = tid; j < P; j += WGS_NT) {
		if (j < n) { key[j] = (uint64_t)kin[j] & mask; idx[j] = (uint16_t)j; }
		else { key[j] = ~0ULL; idx[j] = 0xffff; }                 // padding sorts behind every record (equal keys: by the larger index)
	}
	__syncthreads();
	for (uint32_t k = 2; k <= P; k <<= 1) for (uint32_t j = k >> 1; j > 0; j >>= 1) {
		for (uint32_t i = tid; i < P; i += WGS_NT) {
			const uint32_t x = i ^ j;
			if (x > i) {
				const bool up = (i & k) == 0;
				const uint64_t ka = key[i], kb = key[x];
				const uint16_t ia = idx[i], ib = idx[x];
				const bool gt = ka > kb || (ka == kb && ia > ib);
				if (gt == up) { key[i] = kb; key[x] = ka; idx[i] = ib; idx[x] = ia; }
			}
		}
		__syncthreads();
	}
	for (uint32_t j = tid; j < n; j += WGS_NT) { const uint32_t o = idx[j]; kout[j] = kin[o]; vout[j] = vin[o]; }
}

} // namespace pga
