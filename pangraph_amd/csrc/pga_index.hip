// pga_index.hip -- kernel group #2: minimizer index of a batch.
//
// Replaces mm_idx_str()'s bucket/khash index (reference: packages/minimap2-sys/minimap2/index.c:186-270,
// 293-300,408-456).  Only what mm_idx_get() RETURNS is observable (a list of occurrence words
// y = rid<<32|pos<<1|strand in ascending order per minimizer hash, index.c:84-98,252), so the layout is free:
// one device-wide stable radix sort of (x -> y) -- minimizers arrive ordered by (rid,pos), hence y-ascending,
// and a stable sort keeps that order inside each key -- then run-length boundaries give the distinct keys and
// a CSR offset array.  Each minimizer also learns the id of its key group, which is what the all-vs-all
// seeding stage needs instead of a hash probe (queries ARE the indexed sequences in pangraph,
// packages/pangraph/src/align/minimap2_lib/align_with_minimap2_lib.rs:62-74).
#include "pga_common.h"
#include "pga_pipeline.h"
#include <rocprim/rocprim.hpp>

namespace pga {

__global__ void k_split(const u128 *__restrict__ mz, uint64_t n, uint64_t *__restrict__ kx, uint64_t *__restrict__ vy)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { u128 m = mz[i]; kx[i] = m.x; vy[i] = m.y; }
}

__global__ void k_iota(uint32_t *v, uint64_t n)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) v[i] = (uint32_t)i;
}

__global__ void k_head_flags(const uint64_t *__restrict__ kx, uint64_t n, uint32_t *__restrict__ flag)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) flag[i] = (i == 0 || (kx[i] >> 8) != (kx[i - 1] >> 8)) ? 1u : 0u;
}

// gid[i] (inclusive scan of head flags) - 1 = group of sorted position i
__global__ void k_groups(const uint64_t *__restrict__ kx, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ gid_incl,
                         const uint32_t *__restrict__ orig, uint64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ occ_off,
                         uint32_t *__restrict__ grp_of_mz)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t g = gid_incl[i] - 1;
	if (flag[i]) { key[g] = kx[i] >> 8; occ_off[g] = (uint32_t)i; }
	grp_of_mz[orig[i]] = g;
}

__global__ void k_counts(const uint32_t *__restrict__ occ_off, uint64_t n_keys, uint32_t *__restrict__ cnt)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_keys) cnt[i] = occ_off[i + 1] - occ_off[i];
}

void build_index_ex(const Minimizers &M, int w, int k, Index &I, DBuf<uint32_t> &grp_of_mz, hipStream_t st)
{
	I.w = w, I.k = k; I.n_occ = M.n; I.n_keys = 0;
	const uint64_t n = M.n;
	grp_of_mz.alloc(n ? n : 1);
	I.occ.alloc(n ? n : 1);
	if (n == 0) { I.key.alloc(1); I.occ_off.alloc(1); I.occ_off.zero(st); return; }
	if (n >= (1ULL << 32)) throw std::runtime_error("pga: more than 2^32 minimizers in one batch");
	const unsigned nb = (unsigned)((n + 255) / 256);
	DBuf<uint64_t> kx(n), kx2(n), vy(n);
	DBuf<uint32_t> orig(n), orig2(n);
	hipLaunchKernelGGL(k_split, dim3(nb), dim3(256), 0, st, M.mz.p, n, kx.p, vy.p);
	hipLaunchKernelGGL(k_iota, dim3(nb), dim3(256), 0, st, orig.p, n);
	// stable LSD radix sort on the 64-bit minimizer word; the permutation is carried as the value
	size_t tmp_bytes = 0;
	PGA_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kx.p, kx2.p, orig.p, orig2.p, n, 0, 64, st));
	DBuf<uint8_t> tmp(tmp_bytes ? tmp_bytes : 1);
	PGA_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, kx.p, kx2.p, orig.p, orig2.p, n, 0, 64, st));
	// occ[i] = y of the i-th sorted minimizer
	{
		auto gather = rocprim::make_transform_iterator(orig2.p, [vyp = vy.p] __device__ (uint32_t o) { return vyp[o]; });
		PGA_HIP(rocprim::transform(gather, I.occ.p, n, rocprim::identity<uint64_t>(), st));
	}
	DBuf<uint32_t> flag(n), gid(n);
	hipLaunchKernelGGL(k_head_flags, dim3(nb), dim3(256), 0, st, kx2.p, n, flag.p);
	size_t tmp2 = 0;
	PGA_HIP(rocprim::inclusive_scan(nullptr, tmp2, flag.p, gid.p, n, rocprim::plus<uint32_t>(), st));
	DBuf<uint8_t> tmpb(tmp2 ? tmp2 : 1);
	PGA_HIP(rocprim::inclusive_scan(tmpb.p, tmp2, flag.p, gid.p, n, rocprim::plus<uint32_t>(), st));
	uint32_t n_keys = 0;
	PGA_HIP(hipMemcpyAsync(&n_keys, gid.p + (n - 1), 4, hipMemcpyDeviceToHost, st));
	PGA_HIP(hipStreamSynchronize(st));
	I.n_keys = n_keys;
	I.key.alloc(n_keys);
	I.occ_off.alloc((size_t)n_keys + 1);
	hipLaunchKernelGGL(k_groups, dim3(nb), dim3(256), 0, st, kx2.p, flag.p, gid.p, orig2.p, n, I.key.p, I.occ_off.p, grp_of_mz.p);
	uint32_t n32 = (uint32_t)n;
	PGA_HIP(hipMemcpyAsync(I.occ_off.p + n_keys, &n32, 4, hipMemcpyHostToDevice, st));
	PGA_HIP(hipGetLastError());
	PGA_HIP(hipStreamSynchronize(st));
}

// mm_idx_cal_max_occ (index.c:186-207): the (uint32)((1-f)*n)-th smallest occurrence count, plus one.
int32_t index_cal_max_occ(const Index &I, float f, hipStream_t st)
{
	if (f <= 0.) return INT32_MAX;
	const uint64_t n = I.n_keys;
	if (n == 0) return 1;
	DBuf<uint32_t> cnt(n), cnt2(n);
	hipLaunchKernelGGL(k_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, I.occ_off.p, n, cnt.p);
	size_t tmp_bytes = 0;
	PGA_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, cnt.p, cnt2.p, n, 0, 32, st));
	DBuf<uint8_t> tmp(tmp_bytes ? tmp_bytes : 1);
	PGA_HIP(rocprim::radix_sort_keys(tmp.p, tmp_bytes, cnt.p, cnt2.p, n, 0, 32, st));
	size_t kk = (uint32_t)((1. - f) * n);
	uint32_t v = 0;
	PGA_HIP(hipMemcpyAsync(&v, cnt2.p + kk, 4, hipMemcpyDeviceToHost, st));
	PGA_HIP(hipStreamSynchronize(st));
	return (int32_t)(v + 1);
}

} // namespace pga
