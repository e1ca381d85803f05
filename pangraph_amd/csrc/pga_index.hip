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
#include "pga_maxocc_hist.h"
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

__global__ void k_group_keys(const uint64_t *__restrict__ vy, const uint32_t *__restrict__ orig, const uint32_t *__restrict__ grp_of_seq, uint64_t n,
                             uint32_t *__restrict__ gk, uint32_t *__restrict__ idx)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { gk[i] = grp_of_seq[vy[orig[i]] >> 32]; idx[i] = (uint32_t)i; }
}
__global__ void k_apply_perm(const uint64_t *__restrict__ kx, const uint32_t *__restrict__ orig, const uint32_t *__restrict__ perm, uint64_t n,
                             uint64_t *__restrict__ kx_out, uint32_t *__restrict__ orig_out)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { uint32_t p = perm[i]; kx_out[i] = kx[p]; orig_out[i] = orig[p]; }
}
__global__ void k_head_flags_g(const uint64_t *__restrict__ kx, const uint32_t *__restrict__ gk, uint64_t n, uint32_t *__restrict__ flag)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) flag[i] = (i == 0 || (kx[i] >> 8) != (kx[i - 1] >> 8) || (gk && gk[i] != gk[i - 1])) ? 1u : 0u;
}
__global__ void k_groups_g(const uint64_t *__restrict__ kx, const uint32_t *__restrict__ gk, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ gid_incl,
                           const uint32_t *__restrict__ orig, uint64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ occ_off, uint32_t *__restrict__ key_grp,
                           uint32_t *__restrict__ grp_of_mz)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t g = gid_incl[i] - 1;
	if (flag[i]) { key[g] = kx[i] >> 8; occ_off[g] = (uint32_t)i; key_grp[g] = gk ? gk[i] : 0u; }
	grp_of_mz[orig[i]] = g;
}

// composite sort key: group id above the 2k hash bits (minimizers arrive in (group, sequence, position) order: rid is at hand)
__global__ void k_split_ck(const u128 *__restrict__ mz, uint64_t n, const uint32_t *__restrict__ grp_of_seq, int hash_bits, uint64_t *__restrict__ ck, uint64_t *__restrict__ vy, uint32_t *__restrict__ orig)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const u128 m = mz[i];
	const uint64_t mask = hash_bits >= 64 ? ~0ULL : (1ULL << hash_bits) - 1;
	ck[i] = (uint64_t)grp_of_seq[m.y >> 32] << hash_bits | ((m.x >> 8) & mask);
	vy[i] = m.y; orig[i] = (uint32_t)i;
}
__global__ void k_head_flags_ck(const uint64_t *__restrict__ ck, uint64_t n, uint32_t *__restrict__ flag)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) flag[i] = (i == 0 || ck[i] != ck[i - 1]) ? 1u : 0u;
}
__global__ void k_groups_ck(const uint64_t *__restrict__ ck, int hash_bits, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ gid_incl, const uint32_t *__restrict__ orig, const uint64_t *__restrict__ vy, uint64_t n,
                            uint64_t *__restrict__ key, uint32_t *__restrict__ occ_off, uint32_t *__restrict__ key_grp, uint32_t *__restrict__ grp_of_mz, uint64_t *__restrict__ occ)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t g = gid_incl[i] - 1;
	const uint64_t c = ck[i];
	if (flag[i]) { key[g] = c & (hash_bits >= 64 ? ~0ULL : (1ULL << hash_bits) - 1); occ_off[g] = (uint32_t)i; key_grp[g] = hash_bits >= 64 ? 0u : (uint32_t)(c >> hash_bits); }
	const uint32_t o = orig[i];
	grp_of_mz[o] = g;
	occ[i] = vy[o];
	if (i == n - 1) occ_off[g + 1] = (uint32_t)n;                          // the end of the last key's list (was a 4-byte copy from the host and a wait)
}

// Bring the minimizers of equal (group, x) together keeping y ascending inside a key.  ONE stable sort of the composite key group << 2k | hash
// when it fits 64 bits (always with pangraph's k <= 28 and fewer than 2^8 ... 2^26 groups); otherwise a stable
// sort on x, then one on the group id.
void build_index_ex(const SeqSet &S, const Minimizers &M, int w, int k, Index &I, DBuf<uint32_t> &grp_of_mz, hipStream_t st)
{
	I.w = w, I.k = k; I.n_occ = M.n; I.n_keys = 0;
	const uint64_t n = M.n;
	grp_of_mz.alloc(n ? n : 1);
	I.occ.alloc(n ? n : 1);
	if (n == 0) { I.key.alloc(1); I.occ_off.alloc(1); I.occ_off.zero(st); I.key_grp.alloc(1); return; }
	if (n >= (1ULL << 32)) throw std::runtime_error("pga: more than 2^32 minimizers in one batch");
	const unsigned nb = (unsigned)((n + 255) / 256);
	{
		int gbits = 0; while ((1LL << gbits) < S.n_grp) ++gbits;
		const int hash_bits = std::min(64, 2 * k);
		if (hash_bits + gbits <= 64 && !getenv("PGA_INDEX_TWO_SORTS")) {
			DBuf<uint64_t> ck(n), ck2(n), vy(n);
			DBuf<uint32_t> orig(n), orig2(n);
			hipLaunchKernelGGL(k_split_ck, dim3(nb), dim3(256), 0, st, M.mz.p, n, S.d_grp_of_seq.p, hash_bits, ck.p, vy.p, orig.p);
			// The group digit could be left out: the sort is stable and the minimizers arrive group by group, so a sort over the hash bits alone
			// keeps the groups apart inside a run of equal hashes -- the same lists in (hash, group) order, and the order of the KEYS is not
			// observable (a minimizer finds its list through grp_of_mz).  PGA_INDEX_HASH_ORDER=1 does that (one digit pass of six saved; all
			// 1998 calls of the BASELINE build keep their digests).  It is NOT the default: the one time it was measured the step was 11 % slower
			// (medians of four steps, one box: 2 397 against 2 146 ms; the host's CPU time per step was 12 % higher in that run too, so the box
			// may have been disturbed -- not measured again).  A possible mechanism: in (group, hash) order the lists a group's queries read lie
			// together (~2 MB per whole-genome pair: resident in an XCD's L2), in hash order they are spread over the batch's whole array.
			static const bool hash_order = getenv("PGA_INDEX_HASH_ORDER") != nullptr;
			const unsigned sort_end = (unsigned)(hash_bits + (hash_order ? 0 : gbits));
			size_t tmp_bytes = 0;
			PGA_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ck.p, ck2.p, orig.p, orig2.p, n, 0, sort_end, st));
			DBuf<uint8_t> tmp(tmp_bytes ? tmp_bytes : 1);
			PGA_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, ck.p, ck2.p, orig.p, orig2.p, n, 0, sort_end, st));
			DBuf<uint32_t> flag(n), gid(n);
			hipLaunchKernelGGL(k_head_flags_ck, dim3(nb), dim3(256), 0, st, ck2.p, n, flag.p);
			size_t tmp2 = 0;
			PGA_HIP(rocprim::inclusive_scan(nullptr, tmp2, flag.p, gid.p, n, rocprim::plus<uint32_t>(), st));
			DBuf<uint8_t> tmpb(tmp2 ? tmp2 : 1);
			PGA_HIP(rocprim::inclusive_scan(tmpb.p, tmp2, flag.p, gid.p, n, rocprim::plus<uint32_t>(), st));
			uint32_t n_keys = 0;
			PGA_HIP(hipMemcpyAsync(&n_keys, gid.p + (n - 1), 4, hipMemcpyDeviceToHost, st));
			PGA_HIP(sync_stream(st));
			I.n_keys = n_keys;
			I.key.alloc(n_keys);
			I.occ_off.alloc((size_t)n_keys + 1);
			I.key_grp.alloc(n_keys);
			hipLaunchKernelGGL(k_groups_ck, dim3(nb), dim3(256), 0, st, ck2.p, hash_bits, flag.p, gid.p, orig2.p, vy.p, n, I.key.p, I.occ_off.p, I.key_grp.p, grp_of_mz.p, I.occ.p);
			PGA_HIP(hipGetLastError());
			return;                                                              // (no wait: the scratch of this scope goes back to the call's arena, reused in stream order)
		}
	}
	DBuf<uint64_t> kx(n), kx2(n), vy(n);
	DBuf<uint32_t> orig(n), orig2(n), gk;
	hipLaunchKernelGGL(k_split, dim3(nb), dim3(256), 0, st, M.mz.p, n, kx.p, vy.p);
	hipLaunchKernelGGL(k_iota, dim3(nb), dim3(256), 0, st, orig.p, n);
	{
		size_t tmp_bytes = 0;
		// x = hash << 8 | span with a 2k-bit hash and span == k for every minimizer (sketch.c:136 without HPC): only bits [8, 8+2k) vary
		const int lo_bit = 8, hi_bit = std::min(64, 8 + 2 * k);
		PGA_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kx.p, kx2.p, orig.p, orig2.p, n, lo_bit, hi_bit, st));
		DBuf<uint8_t> tmp(tmp_bytes ? tmp_bytes : 1);
		PGA_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, kx.p, kx2.p, orig.p, orig2.p, n, lo_bit, hi_bit, st));
	}
	uint64_t *kxs = kx2.p; uint32_t *origs = orig2.p; const uint32_t *gks = nullptr;
	if (S.n_grp > 1) {
		DBuf<uint32_t> gk0(n), idx0(n), idx1(n);
		gk.alloc(n);
		hipLaunchKernelGGL(k_group_keys, dim3(nb), dim3(256), 0, st, vy.p, orig2.p, S.d_grp_of_seq.p, n, gk0.p, idx0.p);
		int bits = 1; while ((1LL << bits) < S.n_grp) ++bits;
		size_t tmp_bytes = 0;
		PGA_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, gk0.p, gk.p, idx0.p, idx1.p, n, 0, bits, st));
		DBuf<uint8_t> tmp(tmp_bytes ? tmp_bytes : 1);
		PGA_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, gk0.p, gk.p, idx0.p, idx1.p, n, 0, bits, st));
		hipLaunchKernelGGL(k_apply_perm, dim3(nb), dim3(256), 0, st, kx2.p, orig2.p, idx1.p, n, kx.p, orig.p);
		kxs = kx.p, origs = orig.p, gks = gk.p;
		PGA_HIP(sync_stream(st));
	}
	{   // occ[i] = y of the i-th sorted minimizer
		auto gather = rocprim::make_transform_iterator(origs, [vyp = vy.p] __device__ (uint32_t o) { return vyp[o]; });
		PGA_HIP(rocprim::transform(gather, I.occ.p, n, rocprim::identity<uint64_t>(), st));
	}
	DBuf<uint32_t> flag(n), gid(n);
	hipLaunchKernelGGL(k_head_flags_g, dim3(nb), dim3(256), 0, st, kxs, gks, n, flag.p);
	size_t tmp2 = 0;
	PGA_HIP(rocprim::inclusive_scan(nullptr, tmp2, flag.p, gid.p, n, rocprim::plus<uint32_t>(), st));
	DBuf<uint8_t> tmpb(tmp2 ? tmp2 : 1);
	PGA_HIP(rocprim::inclusive_scan(tmpb.p, tmp2, flag.p, gid.p, n, rocprim::plus<uint32_t>(), st));
	uint32_t n_keys = 0;
	PGA_HIP(hipMemcpyAsync(&n_keys, gid.p + (n - 1), 4, hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
	I.n_keys = n_keys;
	I.key.alloc(n_keys);
	I.occ_off.alloc((size_t)n_keys + 1);
	I.key_grp.alloc(n_keys);
	hipLaunchKernelGGL(k_groups_g, dim3(nb), dim3(256), 0, st, kxs, gks, flag.p, gid.p, origs, n, I.key.p, I.occ_off.p, I.key_grp.p, grp_of_mz.p);
	uint32_t n32 = (uint32_t)n;
	PGA_HIP(hipMemcpyAsync(I.occ_off.p + n_keys, &n32, 4, hipMemcpyHostToDevice, st));
	PGA_HIP(hipGetLastError());
	PGA_HIP(sync_stream(st));
}

// comp = group << cbits | occurrence count: a count never exceeds the number of indexed minimizers, so cbits = bit length of n_occ holds it
// and the sort below runs over gbits + cbits bits (four or five digit passes) instead of all 64 (eight)
__global__ void k_grp_cnt_keys(const uint32_t *__restrict__ occ_off, const uint32_t *__restrict__ key_grp, uint64_t n_keys, int cbits, uint64_t *__restrict__ comp)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_keys) comp[i] = (uint64_t)key_grp[i] << cbits | (uint64_t)(occ_off[i + 1] - occ_off[i]);
}
__global__ void k_grp_quantile(const uint64_t *__restrict__ comp, uint64_t n_keys, int n_grp, int cbits, float f, int32_t *__restrict__ out)
{
	int g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_grp) return;
	uint64_t lo = 0, hi = n_keys, a, b;
	while (lo < hi) { uint64_t m = (lo + hi) >> 1; if ((comp[m] >> cbits) < (uint64_t)g) lo = m + 1; else hi = m; }
	a = lo; hi = n_keys;
	while (lo < hi) { uint64_t m = (lo + hi) >> 1; if ((comp[m] >> cbits) < (uint64_t)g + 1) lo = m + 1; else hi = m; }
	b = lo;
	const uint64_t n = b - a;
	if (n == 0) { out[g] = 1; return; }
	const uint64_t kk = (uint32_t)((1. - (double)f) * (double)n);      // index.c:204: double arithmetic on the float fraction
	out[g] = (int32_t)((uint32_t)(comp[a + kk] & ((1ULL << cbits) - 1)) + 1u);
}

// mm_idx_cal_max_occ (index.c:186-207) of every group: the (uint32)((1-f)*n)-th smallest occurrence count, plus one.
std::vector<int32_t> index_cal_max_occ(const SeqSet &S, const Index &I, float f, hipStream_t st)
{
	std::vector<int32_t> out((size_t)S.n_grp, INT32_MAX);
	if (f <= 0.) return out;
	const uint64_t n = I.n_keys;
	if (n == 0) { std::fill(out.begin(), out.end(), 1); return out; }
	// The default since round 6 (pga_maxocc_hist.h): a histogram of the counts per group instead of a sort of all keys -- one fill and two launches for the
	// ~14 dispatches of the key sort; a group whose answer is a count the histogram does not resolve (1 023 or more) sends the batch to the sort below.
	// All 1998 calls of the BASELINE build keep their digests (mid_occ itself is compared in tests/test_gpu_zz_candidates.py); PGA_MAXOCC_HIST=0: the sort.
	static const bool hist_route = !(getenv("PGA_MAXOCC_HIST") && getenv("PGA_MAXOCC_HIST")[0] == '0');
	if (hist_route && n < (1ULL << 32) && (size_t)S.n_grp * MO_BINS * sizeof(uint32_t) <= ((size_t)1 << 30)) {
		DBuf<uint32_t> hist((size_t)S.n_grp * MO_BINS); hist.zero(st);
		DBuf<int32_t> d_sel((size_t)S.n_grp);
		hipLaunchKernelGGL(k_mo_hist, dim3((unsigned)((n + MO_KEYS - 1) / MO_KEYS)), dim3(MO_NT), 0, st, I.occ_off.p, I.key_grp.p, (uint32_t)n, hist.p);
		hipLaunchKernelGGL(k_mo_select, dim3((unsigned)S.n_grp), dim3(MO_NT), 0, st, hist.p, S.n_grp, f, d_sel.p);
		PGA_HIP(hipGetLastError());
		std::vector<int32_t> sel = d_sel.download(st);
		bool resolved = true;
		for (int32_t v : sel) if (v < 0) { resolved = false; break; }
		if (resolved) return sel;
	}
	DBuf<uint64_t> comp(n), comp2(n);
	int cbits = 1; while (cbits < 32 && (I.n_occ >> cbits) != 0) ++cbits;        // counts are <= n_occ < 2^32
	int gbits = 0; while (gbits < 32 && (1LL << gbits) < S.n_grp) ++gbits;       // groups are < n_grp
	const unsigned sort_bits = (unsigned)(getenv("PGA_MAXOCC_SORT64") ? 64 : gbits + cbits);
	hipLaunchKernelGGL(k_grp_cnt_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, I.occ_off.p, I.key_grp.p, n, cbits, comp.p);
	size_t tmp_bytes = 0;
	PGA_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, comp.p, comp2.p, n, 0, sort_bits, st));
	DBuf<uint8_t> tmp(tmp_bytes ? tmp_bytes : 1);
	PGA_HIP(rocprim::radix_sort_keys(tmp.p, tmp_bytes, comp.p, comp2.p, n, 0, sort_bits, st));
	DBuf<int32_t> d_out((size_t)S.n_grp);
	hipLaunchKernelGGL(k_grp_quantile, dim3((unsigned)((S.n_grp + 255) / 256)), dim3(256), 0, st, comp2.p, n, S.n_grp, cbits, f, d_out.p);
	PGA_HIP(hipGetLastError());
	return d_out.download(st);
}

} // namespace pga
