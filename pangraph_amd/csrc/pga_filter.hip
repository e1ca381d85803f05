// pga_filter.hip -- SURVEY 8(f)-2: what pangraph does with the match list right behind the alignment path, on the device, so that
// only ACCEPTED matches leave the GPU (and travel in the multi-GPU gather).
//
//   self_merge: drop self matches, split_matches per match   packages/pangraph/src/pangraph/graph_merging.rs:105-113
//   split_matches / keep_groups / generate_subalignment / side_patches   packages/pangraph/src/pangraph/split_matches.rs:13-237
//   add_flanking_indel                                                    packages/pangraph/src/align/bam/cigar.rs:60-96
//   alignment_energy2                                                     packages/pangraph/src/align/energy.rs:37-54
//   filter_matches (E < 0, stable sort by energy, greedy interval compatibility per block)   graph_merging.rs:187-240
//
// Kernels: k_split<count|write> one thread per match walks its CIGAR (groups between indels >= threshold, positions by running sums;
// a kept group starts and ends with a match operation, so every side patch is one operation put in front or appended);
// energies; two stable rocPRIM sorts (energy, then group) give every group's candidates in the reference's order, ties in input
// order (the reference's own order of ties is that of its parallel aligner, i.e. undefined); k_greedy one WAVE per group walks the
// candidates in order and tests each against the group's accepted intervals 64 at a time; accepted records and their CIGARs are
// compacted in that order.
#include "pga_common.h"
#include "../../include/pga_align.h"
#include <rocprim/rocprim.hpp>
#include <stdexcept>
#include <vector>

namespace pga {

struct FilterParams { int32_t thr, flags; double alpha, beta; };

__device__ __forceinline__ bool op_is_match(uint32_t op) { const uint32_t k = op & 15u; return k == 0u || k == 7u || k == 8u; }

__device__ __forceinline__ double energy2(const pga_match_t &a, double alpha, double beta)
{
	const double L = (double)a.matches;
	const double M = a.divergence * L;
	int C = 4;
	if (a.qry_start == 0) --C;
	if (a.qry_end == a.qry_len) --C;
	if (a.ref_start == 0) --C;
	if (a.ref_end == a.ref_len) --C;
	return -L + (double)C * alpha + M * beta;
}

// WRITE = false: cnt_rec[i], cnt_ops[i] = sub-alignments of match i and their operations; WRITE = true: they are written at off_rec[i] / off_ops[i]
template <bool WRITE>
__global__ void k_split(const pga_match_t *__restrict__ m, const uint32_t *__restrict__ cig, int64_t n, int thr, uint32_t *__restrict__ cnt_rec, uint32_t *__restrict__ cnt_ops,
                        const uint32_t *__restrict__ off_rec, const uint32_t *__restrict__ off_ops, pga_match_t *__restrict__ out_m, uint32_t *__restrict__ out_cig, int *__restrict__ bad)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const pga_match_t a = m[i];
	uint32_t n_rec = 0, n_ops = 0;
	if (a.qry != a.ref) {
		const uint32_t *c = cig + a.cigar_off;
		long g_start = -1, last_match = -1;
		unsigned long long M = 0, I = 0, D = 0;
		long long q = 0, r = 0;                          // bases consumed in front of operation j
		long long gq = 0, gr = 0, lq = 0, lr = 0;        // ... in front of the group, and behind its last match
		auto close = [&]() {
			if (g_start >= 0 && last_match >= 0 && M >= (unsigned long long)thr) {
				const uint32_t core = (uint32_t)(last_match - g_start + 1);
				pga_match_t o = a;
				o.ref_start = a.ref_start + (int32_t)gr; o.ref_end = a.ref_start + (int32_t)lr;
				if (!a.reverse) { o.qry_start = a.qry_start + (int32_t)gq; o.qry_end = a.qry_start + (int32_t)lq; }
				else { o.qry_start = a.qry_end - (int32_t)lq; o.qry_end = a.qry_end - (int32_t)gq; }
				// side patches: which, and how long
				const int p_rl = (o.ref_start > 0 && o.ref_start < thr) ? o.ref_start : 0;
				const int p_rt = (o.ref_end < o.ref_len && o.ref_len - o.ref_end < thr) ? o.ref_len - o.ref_end : 0;
				const int p_qs = (o.qry_start > 0 && o.qry_start < thr) ? o.qry_start : 0;
				const int p_qe = (o.qry_end < o.qry_len && o.qry_len - o.qry_end < thr) ? o.qry_len - o.qry_end : 0;
				const uint32_t total = core + (p_rl > 0) + (p_rt > 0) + (p_qs > 0) + (p_qe > 0);
				if (WRITE) {
					// the query's start patch leads on the forward strand and trails on the reverse one (split_matches.rs:213-217), its end patch
					// the other way round; a reference patch is placed first, so an insertion in front comes before the deletion and one
					// at the back behind it
					const int i_lead = !a.reverse ? p_qs : p_qe, i_trail = !a.reverse ? p_qe : p_qs;
					uint32_t *w = out_cig + off_ops[i] + n_ops;
					uint32_t k = 0;
					if (i_lead > 0) w[k++] = (uint32_t)i_lead << 4 | 1u;
					if (p_rl > 0) w[k++] = (uint32_t)p_rl << 4 | 2u;
					long long ml = 0, tl = 0;
					for (long j = g_start; j <= last_match; ++j) { const uint32_t op = c[j]; w[k++] = op; tl += op >> 4; if (op_is_match(op)) ml += op >> 4; }
					if (p_rt > 0) w[k++] = (uint32_t)p_rt << 4 | 2u;
					if (i_trail > 0) w[k++] = (uint32_t)i_trail << 4 | 1u;
					o.matches = (int32_t)ml; o.length = (int32_t)(tl + p_rl + p_rt + p_qs + p_qe);
					if (p_rl) o.ref_start = 0;
					if (p_rt) o.ref_end = o.ref_len;
					if (p_qs) o.qry_start = 0;
					if (p_qe) o.qry_end = o.qry_len;
					o.cigar_off = (uint64_t)off_ops[i] + n_ops; o.n_cigar = total; o.pad = 0;
					out_m[off_rec[i] + n_rec] = o;
				}
				++n_rec; n_ops += total;
			}
			g_start = -1; last_match = -1; M = I = D = 0;
		};
		for (uint32_t j = 0; j < a.n_cigar; ++j) {
			const uint32_t op = c[j], k = op & 15u, len = op >> 4;
			const bool mt = op_is_match(op);
			if (g_start < 0) {
				if (mt) { g_start = j; gq = q; gr = r; }
			}
			if (g_start >= 0) {
				if (mt) { M += len; I = 0; D = 0; last_match = j; }
				else if (k == 1u) I += len;
				else if (k == 2u) D += len;
				else { atomicMax(bad, 1); return; }
			}
			if (mt || k == 1u) q += len;
			if (mt || k == 2u) r += len;
			if (mt) { lq = q; lr = r; }
			if (g_start >= 0 && (I > D ? I : D) >= (unsigned long long)thr) close();
		}
		close();
	}
	if (!WRITE) { cnt_rec[i] = n_rec; cnt_ops[i] = n_ops; }
}

__global__ void k_energy_keys(const pga_match_t *__restrict__ m, uint32_t n, double alpha, double beta, unsigned long long *__restrict__ key, uint32_t *__restrict__ idx, uint32_t *__restrict__ keep)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double e = energy2(m[i], alpha, beta);
	const unsigned long long b = (unsigned long long)__double_as_longlong(e);
	key[i] = (b >> 63) ? ~b : (b | 0x8000000000000000ULL);      // order-preserving image of a finite double
	idx[i] = i;
	keep[i] = e < 0.0 ? 1u : 0u;                                 // graph_merging.rs:198
}
__global__ void k_gather_group(const pga_match_t *__restrict__ m, const uint32_t *__restrict__ idx, uint32_t n, uint32_t *__restrict__ grp)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) grp[i] = (uint32_t)m[idx[i]].group;
}

// one wave per group: candidates [seg[g], seg[g+1]) of the sorted list, in order
__global__ __launch_bounds__(64)
void k_greedy(const pga_match_t *__restrict__ m, const uint32_t *__restrict__ order, const uint32_t *__restrict__ seg, uint32_t n_grp, const uint32_t *__restrict__ keep,
              int4 *__restrict__ acc, uint32_t *__restrict__ accepted)
{
	const uint32_t g = blockIdx.x;
	if (g >= n_grp) return;
	const int lane = threadIdx.x;
	const uint32_t lo = seg[g], hi = seg[g + 1];
	int4 *A = acc + 2 * (size_t)lo;                          // the group's accepted intervals: (block, start, end, -)
	uint32_t n_acc = 0;
	for (uint32_t c = lo; c < hi; ++c) {
		const uint32_t id = order[c];
		if (!keep[id]) { if (lane == 0) accepted[c] = 0; continue; }
		const pga_match_t a = m[id];
		bool clash = false;
		for (uint32_t j = lane; j < n_acc; j += 64) {       // graph_merging.rs:218-230, interval.rs:42-46
			const int4 v = A[j];
			clash |= (v.x == a.ref && v.z > a.ref_start && v.y < a.ref_end) || (v.x == a.qry && v.z > a.qry_start && v.y < a.qry_end);
		}
		const bool any = __ballot(clash) != 0ull;
		if (!any) {
			if (lane == 0) { A[n_acc] = make_int4(a.ref, a.ref_start, a.ref_end, 0); A[n_acc + 1] = make_int4(a.qry, a.qry_start, a.qry_end, 0); }
			n_acc += 2;
			__threadfence();                                 // (lane 0's stores must be what every lane's next loads see: write back + L1 invalidate)
		}
		if (lane == 0) accepted[c] = any ? 0u : 1u;
	}
}

__global__ void k_heads_u32(const uint32_t *__restrict__ v, uint32_t n, uint32_t *__restrict__ head)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) head[i] = (i == 0 || v[i] != v[i - 1]) ? 1u : 0u;
}
__global__ void k_seg_starts(const uint32_t *__restrict__ head, const uint32_t *__restrict__ rank_incl, uint32_t n, uint32_t *__restrict__ seg)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && head[i]) seg[rank_incl[i] - 1] = i;
	if (i == 0) seg[rank_incl[n - 1]] = n;
}
__global__ void k_acc_sizes(const pga_match_t *__restrict__ m, const uint32_t *__restrict__ order, const uint32_t *__restrict__ accepted, uint32_t n, uint32_t *__restrict__ ops)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) ops[i] = accepted[i] ? m[order[i]].n_cigar : 0u;
}
__global__ void k_compact(const pga_match_t *__restrict__ m, const uint32_t *__restrict__ cig, const uint32_t *__restrict__ order, const uint32_t *__restrict__ accepted,
                          const uint32_t *__restrict__ rec_excl, const uint32_t *__restrict__ ops_excl, uint32_t n, pga_match_t *__restrict__ out_m, uint32_t *__restrict__ out_cig)
{
	const uint32_t i = blockIdx.x;                           // one workgroup per sorted candidate
	if (i >= n || !accepted[i]) return;
	pga_match_t a = m[order[i]];
	const uint32_t *src = cig + a.cigar_off;
	uint32_t *dst = out_cig + ops_excl[i];
	for (uint32_t j = threadIdx.x; j < a.n_cigar; j += blockDim.x) dst[j] = src[j];
	if (threadIdx.x == 0) { a.cigar_off = ops_excl[i]; out_m[rec_excl[i]] = a; }
}

template <class T> static void excl_scan_u32(const T *in, T *out, size_t n, hipStream_t st)
{
	size_t tb = 0;
	PGA_HIP(rocprim::exclusive_scan(nullptr, tb, in, out, (T)0, n, rocprim::plus<T>(), st));
	DBuf<uint8_t> tmp(tb + 16);
	PGA_HIP(rocprim::exclusive_scan(tmp.p, tb, in, out, (T)0, n, rocprim::plus<T>(), st));
}

void filter_matches_dev(int64_t n_in, const pga_match_t *h_m, const uint32_t *h_cig, uint64_t n_ops_in, const FilterParams &fp, std::vector<pga_match_t> &out_m, std::vector<uint32_t> &out_cig)
{
	out_m.clear(); out_cig.clear();
	if (n_in <= 0) return;
	if (n_in > 0x7fffffffLL || n_ops_in > 0xffffffffULL) throw std::runtime_error("pga: match list too large for the filter (2^31 records / 2^32 CIGAR operations)");
	hipStream_t st = 0;
	DBuf<pga_match_t> m0; m0.upload(h_m, (size_t)n_in, st);
	DBuf<uint32_t> c0; c0.upload(h_cig, (size_t)n_ops_in + 1 > 0 ? (size_t)n_ops_in : 0, st);
	DBuf<pga_match_t> m1; DBuf<uint32_t> c1;
	const pga_match_t *m = m0.p; const uint32_t *cig = c0.p; uint32_t n = (uint32_t)n_in;
	if (fp.flags & 1) {
		DBuf<uint32_t> cr((size_t)n_in + 1), co((size_t)n_in + 1), orr((size_t)n_in + 1), oo((size_t)n_in + 1);
		DBuf<int> bad(1); bad.zero(st);
		cr.zero(st); co.zero(st);
		const unsigned nb = (unsigned)((n_in + 127) / 128);
		hipLaunchKernelGGL(k_split<false>, dim3(nb), dim3(128), 0, st, m0.p, c0.p, n_in, fp.thr, cr.p, co.p, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (pga_match_t*)nullptr, (uint32_t*)nullptr, bad.p);
		PGA_HIP(hipGetLastError());
		if (bad.download(st)[0]) throw std::runtime_error("pga: Unexpected CIGAR operation in a match (split_matches.rs:62-65)");
		excl_scan_u32(cr.p, orr.p, (size_t)n_in + 1, st);
		excl_scan_u32(co.p, oo.p, (size_t)n_in + 1, st);
		uint32_t tot[2];
		PGA_HIP(hipMemcpyAsync(&tot[0], orr.p + n_in, 4, hipMemcpyDeviceToHost, st));
		PGA_HIP(hipMemcpyAsync(&tot[1], oo.p + n_in, 4, hipMemcpyDeviceToHost, st));
		PGA_HIP(sync_stream(st));
		m1.alloc((size_t)tot[0] + 1); c1.alloc((size_t)tot[1] + 1);
		hipLaunchKernelGGL(k_split<true>, dim3(nb), dim3(128), 0, st, m0.p, c0.p, n_in, fp.thr, (uint32_t*)nullptr, (uint32_t*)nullptr, orr.p, oo.p, m1.p, c1.p, bad.p);
		PGA_HIP(hipGetLastError());
		PGA_HIP(sync_stream(st));
		m = m1.p; cig = c1.p; n = tot[0];
		if (!(fp.flags & 2)) {
			out_m.resize(n); out_cig.resize(tot[1]);
			if (n) PGA_HIP(hipMemcpyAsync(out_m.data(), m1.p, (size_t)n * sizeof(pga_match_t), hipMemcpyDeviceToHost, st));
			if (tot[1]) PGA_HIP(hipMemcpyAsync(out_cig.data(), c1.p, (size_t)tot[1] * 4, hipMemcpyDeviceToHost, st));
			PGA_HIP(sync_stream(st));
			return;
		}
	} else if (!(fp.flags & 2)) {
		out_m.assign(h_m, h_m + n_in); out_cig.assign(h_cig, h_cig + n_ops_in);
		return;
	}
	if (n == 0) return;
	// ---- filter_matches ----
	DBuf<unsigned long long> key(n), key2(n); DBuf<uint32_t> idx(n), idx2(n), keep(n), grp(n), grp2(n), ord(n);
	const unsigned nb = (n + 255) / 256;
	hipLaunchKernelGGL(k_energy_keys, dim3(nb), dim3(256), 0, st, m, n, fp.alpha, fp.beta, key.p, idx.p, keep.p);
	size_t tb = 0;
	PGA_HIP(rocprim::radix_sort_pairs(nullptr, tb, key.p, key2.p, idx.p, idx2.p, n, 0, 64, st));
	{ DBuf<uint8_t> tmp(tb + 16); PGA_HIP(rocprim::radix_sort_pairs(tmp.p, tb, key.p, key2.p, idx.p, idx2.p, n, 0, 64, st)); }
	hipLaunchKernelGGL(k_gather_group, dim3(nb), dim3(256), 0, st, m, idx2.p, n, grp.p);
	PGA_HIP(rocprim::radix_sort_pairs(nullptr, tb, grp.p, grp2.p, idx2.p, ord.p, n, 0, 32, st));
	{ DBuf<uint8_t> tmp(tb + 16); PGA_HIP(rocprim::radix_sort_pairs(tmp.p, tb, grp.p, grp2.p, idx2.p, ord.p, n, 0, 32, st)); }
	DBuf<uint32_t> head(n), rank(n);
	hipLaunchKernelGGL(k_heads_u32, dim3(nb), dim3(256), 0, st, grp2.p, n, head.p);
	PGA_HIP(rocprim::inclusive_scan(nullptr, tb, head.p, rank.p, n, rocprim::plus<uint32_t>(), st));
	{ DBuf<uint8_t> tmp(tb + 16); PGA_HIP(rocprim::inclusive_scan(tmp.p, tb, head.p, rank.p, n, rocprim::plus<uint32_t>(), st)); }
	uint32_t n_grp = 0;
	PGA_HIP(hipMemcpyAsync(&n_grp, rank.p + (n - 1), 4, hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
	DBuf<uint32_t> seg((size_t)n_grp + 1), accepted(n), aops((size_t)n + 1), rex((size_t)n + 1), oex((size_t)n + 1);
	hipLaunchKernelGGL(k_seg_starts, dim3(nb), dim3(256), 0, st, head.p, rank.p, n, seg.p);
	DBuf<int4> acc(2 * (size_t)n + 2);
	hipLaunchKernelGGL(k_greedy, dim3(n_grp), dim3(64), 0, st, m, ord.p, seg.p, n_grp, keep.p, acc.p, accepted.p);
	PGA_HIP(hipGetLastError());
	// compaction in the sorted order
	DBuf<uint32_t> acc1((size_t)n + 1); acc1.zero(st); aops.zero(st);
	PGA_HIP(hipMemcpyAsync(acc1.p, accepted.p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
	hipLaunchKernelGGL(k_acc_sizes, dim3(nb), dim3(256), 0, st, m, ord.p, accepted.p, n, aops.p);
	excl_scan_u32(acc1.p, rex.p, (size_t)n + 1, st);
	excl_scan_u32(aops.p, oex.p, (size_t)n + 1, st);
	uint32_t tot[2];
	PGA_HIP(hipMemcpyAsync(&tot[0], rex.p + n, 4, hipMemcpyDeviceToHost, st));
	PGA_HIP(hipMemcpyAsync(&tot[1], oex.p + n, 4, hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
	DBuf<pga_match_t> fm((size_t)tot[0] + 1); DBuf<uint32_t> fc((size_t)tot[1] + 1);
	hipLaunchKernelGGL(k_compact, dim3(n), dim3(64), 0, st, m, cig, ord.p, accepted.p, rex.p, oex.p, n, fm.p, fc.p);
	PGA_HIP(hipGetLastError());
	out_m.resize(tot[0]); out_cig.resize(tot[1]);
	if (tot[0]) PGA_HIP(hipMemcpyAsync(out_m.data(), fm.p, (size_t)tot[0] * sizeof(pga_match_t), hipMemcpyDeviceToHost, st));
	if (tot[1]) PGA_HIP(hipMemcpyAsync(out_cig.data(), fc.p, (size_t)tot[1] * 4, hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
}

} // namespace pga
