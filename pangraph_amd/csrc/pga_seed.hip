// pga_seed.hip -- kernel group #3: query-side minimizer filter, seed selection, anchor expansion and the
// anchor sort, for every query of an all-vs-all batch at once.
//
// Replaces mm_seed_mz_flt / mm_seed_collect_all / mm_seed_select / mm_collect_matches
// (reference: packages/minimap2-sys/minimap2/seed.c:5-131) and skip_seed / collect_seed_hits (map.c:78-100,168-204).
// Output per query: the anchor array a[] exactly as minimap2 hands it to chaining --
//   x = rev<<63 | rid<<32 | rpos,  y = flags<<40 | span<<32 | qpos  (lchain.c:140-147),
// sorted by x with radix_sort_128x's tie order (map.c:202).
//
// Every query of the batch is one of the indexed sequences (pangraph maps exactly what it indexed), so the
// "probe" of a minimizer is its key group from the index build, and the number of copies of a hash inside
// the query is the length of the rid-run inside that group's occurrence list (both are lookups, no hashing).
#include "pga_common.h"
#include "pga_sort_exact.h"
#include "pga_sort_wave.h"
#include "pga_pipeline.h"
#include <rocprim/rocprim.hpp>
#include <cstdio>

namespace pga {

#define SEED_TANDEM (1ULL << 42)  // mmpriv.h:20
#define SEED_SELF   (1ULL << 43)  // mmpriv.h:21

struct SeedParams {
	int64_t flag;
	const int32_t *mid_occ_of_grp;      // mm_mapopt_t::mid_occ is per index, i.e. per group
	const uint32_t *grp_of_seq, *grp_base;
	int32_t max_max_occ, occ_dist;
	float q_occ_frac;
	__device__ __forceinline__ int32_t mid_occ(uint32_t qid) const { return mid_occ_of_grp[grp_of_seq[qid]]; }
};

__device__ __forceinline__ uint32_t lower_bound_u64(const uint64_t *a, uint32_t n, uint64_t v)
{
	uint32_t lo = 0, hi = n;
	while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
	return lo;
}

// seed.c:5-28 -- keep[i]=0 for query minimizers whose hash is over-represented inside the query
__global__ void k_mz_keep(const u128 *__restrict__ mz, uint64_t n, const uint64_t *__restrict__ seq_off, const uint32_t *__restrict__ grp,
                          const uint32_t *__restrict__ occ_off, const uint64_t *__restrict__ occ, SeedParams P, const uint8_t *__restrict__ own, uint32_t *__restrict__ keep)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t kp = 1;
	const uint32_t qid = (uint32_t)(mz[i].y >> 32);
	if (own && !own[qid]) { keep[i] = 0; return; }                     // a query of another shard (pga_batch_align_shard): indexed here, mapped there
	const uint64_t n_mv = seq_off[qid + 1] - seq_off[qid];
	const int32_t mid_occ = P.mid_occ(qid);
	if (n_mv > (uint64_t)mid_occ && P.q_occ_frac > 0.0f && mid_occ > 0) {
		const uint32_t g = grp[i], o0 = occ_off[g], cn = occ_off[g + 1] - o0;
		if (cn > (uint32_t)mid_occ) {
			const uint64_t *cr = occ + o0;
			int32_t cnt = (int32_t)(lower_bound_u64(cr, cn, (uint64_t)(qid + 1) << 32) - lower_bound_u64(cr, cn, (uint64_t)qid << 32));
			if (cnt > mid_occ && (float)cnt > (float)n_mv * P.q_occ_frac) kp = 0; // seed.c:17 compares in float
		}
	}
	keep[i] = kp;
}

__global__ void k_new_seq_off(const uint32_t *__restrict__ pos_excl, const uint64_t *__restrict__ seq_off, int n_seq, uint64_t n, uint32_t total, uint64_t *__restrict__ out)
{
	int q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q <= n_seq) { uint64_t o = seq_off[q]; out[q] = o < n ? pos_excl[o] : total; }
}

// seeds in query order (seed.c:30-54).  sd_* arrays are indexed by the compacted minimizer index.
__global__ void k_seed_make(const u128 *__restrict__ mz, const uint32_t *__restrict__ kept_idx, uint64_t n_kept, const uint64_t *__restrict__ seq_off2,
                            const uint32_t *__restrict__ grp, const uint32_t *__restrict__ occ_off, SeedParams P,
                            uint32_t *__restrict__ sd_n, uint32_t *__restrict__ sd_occ, uint32_t *__restrict__ sd_qpos, uint8_t *__restrict__ sd_flag,
                            uint32_t *__restrict__ q_has_high)
{
	uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n_kept) return;
	const uint32_t i = kept_idx ? kept_idx[j] : (uint32_t)j;
	const u128 m = mz[i];
	const uint32_t qid = (uint32_t)(m.y >> 32), g = grp[i], o0 = occ_off[g], cn = occ_off[g + 1] - o0;
	uint8_t fl = 0;
	const uint64_t lo = seq_off2[qid], hi = seq_off2[qid + 1];
	if (j > lo) { uint32_t ip = kept_idx ? kept_idx[j - 1] : (uint32_t)(j - 1); if (mz[ip].x >> 8 == m.x >> 8) fl = 1; }
	if (j + 1 < hi) { uint32_t in = kept_idx ? kept_idx[j + 1] : (uint32_t)(j + 1); if (mz[in].x >> 8 == m.x >> 8) fl = 1; }
	sd_n[j] = cn, sd_occ[j] = o0, sd_qpos[j] = (uint32_t)m.y, sd_flag[j] = fl;   // bit0 = tandem, bit1 = filtered
	if (cn > (uint32_t)P.mid_occ(qid)) atomicOr(&q_has_high[qid], 1u);
}

// seed.c:56-96 + the rep_len accounting of seed.c:107-128; one lane per query that has high-occurrence seeds
__global__ void k_seed_select(int n_seq, const uint64_t *__restrict__ seq_off2, const uint32_t *__restrict__ seq_len, const uint32_t *__restrict__ q_has_high,
                              const uint32_t *__restrict__ sd_n, const uint32_t *__restrict__ sd_qpos, uint8_t *__restrict__ sd_flag, SeedParams P,
                              int32_t k_span, int32_t *__restrict__ rep_len)
{
	int q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n_seq) return;
	rep_len[q] = 0;
	if (!q_has_high[q]) return;
	const int64_t base = (int64_t)seq_off2[q];
	const int32_t n = (int32_t)(seq_off2[q + 1] - seq_off2[q]);
	const uint32_t *a_n = sd_n + base, *a_qp = sd_qpos + base;
	uint8_t *a_fl = sd_flag + base;
	const uint32_t max_occ = (uint32_t)P.mid_occ((uint32_t)q);
	if (P.occ_dist > 0 && P.max_max_occ > (int32_t)max_occ) {
		if (n > 1) {
			uint64_t heap[128];
			int32_t last0 = -1;
			for (int32_t i = 0; i <= n; ++i) {
				if (i != n && a_n[i] > max_occ) continue;
				if (i - last0 > 1) {
					int32_t ps = last0 < 0 ? 0 : (int32_t)(a_qp[last0] >> 1);
					int32_t pe = i == n ? (int32_t)seq_len[q] : (int32_t)(a_qp[i] >> 1);
					int32_t st = last0 + 1, en = i;
					int32_t keep = (int32_t)((double)(pe - ps) / P.occ_dist + .499);
					if (keep > 0) {
						if (keep > 128) keep = 128;
						// the `keep` smallest under (n, index): bounded max-heap, the top is replaced only by a strictly smaller n
						int32_t hs = 0, j;
						for (j = st; j < en && hs < keep; ++j) {   // sift-up insert
							uint64_t v = (uint64_t)a_n[j] << 32 | (uint32_t)j; int32_t c = hs++;
							while (c > 0) { int32_t p = (c - 1) >> 1; if (heap[p] >= v) break; heap[c] = heap[p]; c = p; }
							heap[c] = v;
						}
						for (; j < en; ++j) {
							if (a_n[j] < (uint32_t)(heap[0] >> 32)) {
								uint64_t v = (uint64_t)a_n[j] << 32 | (uint32_t)j; int32_t c = 0;
								for (;;) { int32_t l = 2 * c + 1; if (l >= hs) break; if (l + 1 < hs && heap[l + 1] > heap[l]) ++l; if (heap[l] <= v) break; heap[c] = heap[l]; c = l; }
								heap[c] = v;
							}
						}
						for (j = 0; j < hs; ++j) a_fl[(uint32_t)heap[j]] |= 2;
					}
					for (int32_t j = st; j < en; ++j) a_fl[j] ^= 2;
					for (int32_t j = st; j < en; ++j) if (a_n[j] > (uint32_t)P.max_max_occ) a_fl[j] |= 2;
				}
				last0 = i;
			}
		}
	} else {
		for (int32_t i = 0; i < n; ++i) if (a_n[i] > max_occ) a_fl[i] |= 2;
	}
	int rep_st = 0, rep_en = 0, rl = 0;
	for (int32_t i = 0; i < n; ++i) {
		if (a_fl[i] & 2) {
			int en = (int)(a_qp[i] >> 1) + 1, st = en - k_span;
			if (st > rep_en) { rl += rep_en - rep_st; rep_st = st, rep_en = en; } else rep_en = en;
		}
	}
	rl += rep_en - rep_st;
	rep_len[q] = rl;
}

struct SkipCtx { int64_t flag; const int32_t *name_rank; const uint32_t *seq_len; const uint32_t *grp_base; };

__device__ __forceinline__ bool skip_seed(const SkipCtx &C, uint64_t r, uint32_t q_pos, uint32_t qid, uint32_t qlen, bool *is_self) // map.c:78-100
{
	*is_self = false;
	if (C.flag & (MM_F_NO_DIAG | MM_F_NO_DUAL)) {
		const uint32_t t = (uint32_t)(r >> 32);
		const int32_t cmp = C.name_rank[qid] - C.name_rank[t];           // sign of strcmp(qname, tname)
		if ((C.flag & MM_F_NO_DIAG) && cmp == 0 && C.seq_len[t] == qlen) {
			if ((uint32_t)r >> 1 == (q_pos >> 1)) return true;
			if ((r & 1) == (q_pos & 1)) *is_self = true;
		}
		if ((C.flag & MM_F_NO_DUAL) && cmp > 0) return true;
	}
	if (C.flag & (MM_F_FOR_ONLY | MM_F_REV_ONLY)) {
		if ((r & 1) == (q_pos & 1)) { if (C.flag & MM_F_REV_ONLY) return true; }
		else if (C.flag & MM_F_FOR_ONLY) return true;
	}
	return false;
}

// Anchors of every seed in ONE walk over its occurrence list (map.c:168-204): seed j writes at ub_off[j] -- the offsets of an upper
// bound (every occurrence kept) -- and records how many it kept; k_anchor_pack then packs the kept ones, already split into x / y for the
// sort.  (Counting first and writing second walked every occurrence list twice: the lists are scattered 8-byte reads, one cache line per
// seed and pass -- 0.48 TB of fetches per build for 8 GB of anchors.)
__global__ void k_anchors(const u128 *__restrict__ mz, const uint32_t *__restrict__ kept_idx, uint64_t n_kept,
                          const uint32_t *__restrict__ sd_n, const uint32_t *__restrict__ sd_occ, const uint32_t *__restrict__ sd_qpos, const uint8_t *__restrict__ sd_flag,
                          const uint64_t *__restrict__ occ, SkipCtx C, int32_t k_span,
                          uint32_t *__restrict__ cnt, const uint64_t *__restrict__ ub_off, u128 *__restrict__ ub)
{
	uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n_kept) return;
	const uint8_t fl = sd_flag[j];
	uint32_t c = 0;
	if (!(fl & 2)) {
		const uint32_t i = kept_idx ? kept_idx[j] : (uint32_t)j;
		const uint32_t qid = (uint32_t)(mz[i].y >> 32), qlen = C.seq_len[qid], q_pos = sd_qpos[j], n = sd_n[j];
		const uint64_t *cr = occ + sd_occ[j];
		u128 *out = ub + ub_off[j];
		for (uint32_t t = 0; t < n; ++t) {
			uint64_t r = cr[t];
			bool is_self;
			if (skip_seed(C, r, q_pos, qid, qlen, &is_self)) continue;
			u128 p;
			const uint64_t rpos = (uint32_t)r >> 1;
			r -= (uint64_t)C.grp_base[qid] << 32;                     // target id relative to the group, as in a per-group index
			if ((r & 1) == (q_pos & 1)) {
				p.x = (r & 0xffffffff00000000ULL) | rpos;
				p.y = (uint64_t)k_span << 32 | (q_pos >> 1);
			} else {
				p.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | rpos;
				p.y = (uint64_t)k_span << 32 | (uint32_t)(qlen - ((q_pos >> 1) + 1 - (uint32_t)k_span) - 1);
			}
			if (fl & 1) p.y |= SEED_TANDEM;
			if (is_self) p.y |= SEED_SELF;
			out[c] = p;
			++c;
		}
	}
	cnt[j] = c;
}
// the kept anchors of seed j, from its slots of the upper-bound layout to their final places (x and y apart: what the sort wants)
__global__ void k_anchor_pack(uint64_t n_kept, const uint32_t *__restrict__ cnt, const uint64_t *__restrict__ ub_off, const u128 *__restrict__ ub, const uint64_t *__restrict__ a_off,
                              uint64_t *__restrict__ x, uint64_t *__restrict__ y)
{
	uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n_kept) return;
	const uint32_t c = cnt[j];
	const u128 *in = ub + ub_off[j];
	const uint64_t o = a_off[j];
	for (uint32_t t = 0; t < c; ++t) { const u128 v = in[t]; x[o + t] = v.x; y[o + t] = v.y; }
}

__global__ void k_query_anchor_off(const uint64_t *__restrict__ a_off, const uint64_t *__restrict__ seq_off2, int n_seq, uint64_t n_kept, uint64_t total, uint64_t *__restrict__ q_aoff)
{
	int q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q <= n_seq) { uint64_t o = seq_off2[q]; q_aoff[q] = o < n_kept ? a_off[o] : total; }
}

__global__ void k_iota32(uint32_t *v, uint64_t n)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) v[i] = (uint32_t)i;
}
__global__ void k_query_of_anchor(const uint32_t *__restrict__ idx, const uint64_t *__restrict__ q_aoff, int n_seq, uint64_t n, uint32_t *__restrict__ qk)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t o = idx[i];
	int lo = 0, hi = n_seq;
	while (lo < hi) { int m = (lo + hi) >> 1; if (q_aoff[m + 1] <= o) lo = m + 1; else hi = m; }
	qk[i] = (uint32_t)lo;
}
__global__ void k_gather_xy(const uint64_t *__restrict__ x, const uint64_t *__restrict__ y, const uint32_t *__restrict__ idx, uint64_t n, uint64_t *__restrict__ xo, uint64_t *__restrict__ yo)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { const uint32_t o = idx[i]; xo[i] = x[o]; yo[i] = y[o]; }
}
__global__ void k_split128(const u128 *__restrict__ a, uint64_t n, uint64_t *__restrict__ x, uint64_t *__restrict__ y)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { u128 v = a[i]; x[i] = v.x; y[i] = v.y; }
}
// the sort key of an anchor without its unused bits: strand | target id | target position, packed (same order as x)
__global__ void k_anchor_key(const uint64_t *__restrict__ x, uint64_t n, int rid_bits, int pos_bits, uint64_t *__restrict__ key)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { const uint64_t v = x[i]; key[i] = (v >> 63) << (rid_bits + pos_bits) | ((v >> 32) & 0x7fffffffULL) << pos_bits | (v & 0xffffffffULL); }
}
__global__ void k_join128(const uint64_t *__restrict__ x, const uint64_t *__restrict__ y, uint64_t n, u128 *__restrict__ a)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { u128 v; v.x = x[i]; v.y = y[i]; a[i] = v; }
}
// which queries hold equal keys after the stable sort?  (only those need the sequential replay)
__global__ void k_tie_flags(const uint64_t *__restrict__ xs, const uint64_t *__restrict__ q_aoff, int n_seq, uint64_t n, const u128 *__restrict__ a_unsorted, uint32_t *__restrict__ q_tie)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i + 1 >= n) return;
	if (xs[i] == xs[i + 1]) {
		// both belong to the same query iff no query boundary lies at i+1: x carries no query id, so look it up
		int lo = 0, hi = n_seq;
		while (lo < hi) { int m = (lo + hi) >> 1; if (q_aoff[m + 1] <= i) lo = m + 1; else hi = m; }
		if (i + 1 < q_aoff[lo + 1]) q_tie[lo] = 1;
	}
}
// queries whose anchors hold equal keys restart from the raw order
__global__ void k_copy_tied(int n_seq, const uint32_t *__restrict__ q_tie, const uint64_t *__restrict__ q_aoff, const uint64_t *__restrict__ x_unsorted, const uint64_t *__restrict__ y_unsorted, u128 *__restrict__ a_sorted)
{
	const int q = blockIdx.x;
	if (q >= n_seq || !q_tie[q]) return;
	const uint64_t b = q_aoff[q], e = q_aoff[q + 1];
	for (uint64_t i = b + (uint64_t)blockIdx.y * blockDim.x + threadIdx.x; i < e; i += (uint64_t)gridDim.y * blockDim.x) { u128 v; v.x = x_unsorted[i]; v.y = y_unsorted[i]; a_sorted[i] = v; }
}

// ---- host orchestration ----

template <class T> static void excl_scan(const T *in, uint64_t *out, size_t n, hipStream_t st)
{
	size_t tb = 0;
	auto it = rocprim::make_transform_iterator(in, [] __device__ (T v) { return (uint64_t)v; });
	PGA_HIP(rocprim::exclusive_scan(nullptr, tb, it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), st));
	DBuf<uint8_t> tmp(tb ? tb : 1);
	PGA_HIP(rocprim::exclusive_scan(tmp.p, tb, it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), st));
}

void seed_all(const SeqSet &S, const Minimizers &M, const Index &I, const DBuf<uint32_t> &grp_of_mz, const mm_mapopt_t &opt,
              const DBuf<int32_t> &d_name_rank, const DBuf<int32_t> &d_mid_occ, SeedResult &O, hipStream_t st, Timers *tm, const uint8_t *d_own, bool exact_order)
{
	const int n_seq = S.n_seq;
	const uint64_t n = M.n;
	O.n_a = 0; O.h_q_aoff.assign((size_t)n_seq + 1, 0); O.h_rep_len.assign((size_t)n_seq, 0);
	O.q_aoff.alloc((size_t)n_seq + 1); O.q_aoff.zero(st);
	if (n == 0) { O.a.alloc(1); return; }
	SeedParams P{opt.flag, d_mid_occ.p, S.d_grp_of_seq.p, S.d_grp_base.p, opt.max_max_occ, opt.occ_dist, opt.q_occ_frac};
	const unsigned nb = (unsigned)((n + 255) / 256), nbq = (unsigned)((n_seq + 1 + 255) / 256);

	// 1. query-side filter + order-preserving compaction
	DBuf<uint32_t> keep(n), pos(n + 1);
	hipLaunchKernelGGL(k_mz_keep, dim3(nb), dim3(256), 0, st, M.mz.p, n, M.seq_off.p, grp_of_mz.p, I.occ_off.p, I.occ.p, P, d_own, keep.p);
	DBuf<uint64_t> pos64(n + 1);
	{
		// exclusive scan over n+1 items (the extra item, read as zero, yields the total)
		struct In { const uint32_t *v; uint64_t n; };
		const In in{keep.p, n};
		auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), [in] __device__ (uint64_t i) { return i < in.n ? (uint64_t)in.v[i] : (uint64_t)0; });
		size_t tb = 0;
		PGA_HIP(rocprim::exclusive_scan(nullptr, tb, it, pos64.p, (uint64_t)0, (size_t)n + 1, rocprim::plus<uint64_t>(), st));
		DBuf<uint8_t> tmp(tb ? tb : 1);
		PGA_HIP(rocprim::exclusive_scan(tmp.p, tb, it, pos64.p, (uint64_t)0, (size_t)n + 1, rocprim::plus<uint64_t>(), st));
	}
	uint64_t n_kept = 0;
	PGA_HIP(hipMemcpyAsync(&n_kept, pos64.p + n, 8, hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
	DBuf<uint32_t> kept_idx; DBuf<uint64_t> seq_off2((size_t)n_seq + 1);
	const uint32_t *kept_p = nullptr;
	if (n_kept != n) {
		kept_idx.alloc(n_kept ? n_kept : 1);
		// scatter indices of kept minimizers
		auto scatter = [] __device__ (uint64_t) {};
		(void)scatter;
		struct Sc { const uint32_t *keep; const uint64_t *pos; uint32_t *out; };
		Sc sc{keep.p, pos64.p, kept_idx.p};
		auto idx_it = rocprim::make_counting_iterator<uint64_t>(0);
		PGA_HIP(rocprim::transform(idx_it, rocprim::make_discard_iterator(), n,
		        [sc] __device__ (uint64_t i) { if (sc.keep[i]) sc.out[sc.pos[i]] = (uint32_t)i; return 0; }, st));
		kept_p = kept_idx.p;
		struct So { const uint64_t *pos, *soff; uint64_t *out; uint64_t n, total; };
		So so{pos64.p, M.seq_off.p, seq_off2.p, n, n_kept};
		auto q_it = rocprim::make_counting_iterator<uint64_t>(0);
		PGA_HIP(rocprim::transform(q_it, rocprim::make_discard_iterator(), (size_t)n_seq + 1,
		        [so] __device__ (uint64_t q) { uint64_t o = so.soff[q]; so.out[q] = o < so.n ? so.pos[o] : so.total; return 0; }, st));
	} else {
		PGA_HIP(hipMemcpyAsync(seq_off2.p, M.seq_off.p, ((size_t)n_seq + 1) * 8, hipMemcpyDeviceToDevice, st));
	}
	if (n_kept == 0) { O.a.alloc(1); return; }

	// 2. seeds, selection, rep_len
	DBuf<uint32_t> sd_n(n_kept), sd_occ(n_kept), sd_qpos(n_kept), q_high((size_t)n_seq);
	DBuf<uint8_t> sd_flag(n_kept);
	DBuf<int32_t> rep_len((size_t)n_seq);
	q_high.zero(st);
	const unsigned nbk = (unsigned)((n_kept + 255) / 256);
	hipLaunchKernelGGL(k_seed_make, dim3(nbk), dim3(256), 0, st, M.mz.p, kept_p, n_kept, seq_off2.p, grp_of_mz.p, I.occ_off.p, P,
	                   sd_n.p, sd_occ.p, sd_qpos.p, sd_flag.p, q_high.p);
	hipLaunchKernelGGL(k_seed_select, dim3((unsigned)((n_seq + 63) / 64)), dim3(64), 0, st, n_seq, seq_off2.p, S.d_len.p, q_high.p,
	                   sd_n.p, sd_qpos.p, sd_flag.p, P, (int32_t)I.k, rep_len.p);

	// 3. anchors: one walk over the occurrence lists into an upper-bound layout, scan of the kept counts, pack
	SkipCtx C{opt.flag, d_name_rank.p, S.d_len.p, S.d_grp_base.p};
	DBuf<uint32_t> cnt(n_kept + 1); cnt.zero(st);
	DBuf<uint64_t> a_off(n_kept + 1), ub_off(n_kept + 1);
	{
		struct Ub { const uint32_t *n; const uint8_t *fl; uint64_t n_kept; };
		Ub u{sd_n.p, sd_flag.p, n_kept};
		auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), [u] __device__ (uint64_t j) { return j < u.n_kept && !(u.fl[j] & 2) ? (uint64_t)u.n[j] : (uint64_t)0; });
		size_t tb = 0;
		PGA_HIP(rocprim::exclusive_scan(nullptr, tb, it, ub_off.p, (uint64_t)0, n_kept + 1, rocprim::plus<uint64_t>(), st));
		DBuf<uint8_t> tmp(tb ? tb : 1);
		PGA_HIP(rocprim::exclusive_scan(tmp.p, tb, it, ub_off.p, (uint64_t)0, n_kept + 1, rocprim::plus<uint64_t>(), st));
	}
	uint64_t n_ub = 0;
	PGA_HIP(hipMemcpyAsync(&n_ub, ub_off.p + n_kept, 8, hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
	DBuf<u128> ub(n_ub ? n_ub : 1);
	hipLaunchKernelGGL(k_anchors, dim3(nbk), dim3(256), 0, st, M.mz.p, kept_p, n_kept, sd_n.p, sd_occ.p, sd_qpos.p, sd_flag.p, I.occ.p, C,
	                   (int32_t)I.k, cnt.p, ub_off.p, ub.p);
	excl_scan(cnt.p, a_off.p, n_kept + 1, st);
	uint64_t n_a = 0;
	PGA_HIP(hipMemcpyAsync(&n_a, a_off.p + n_kept, 8, hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
	O.n_a = n_a;
	hipLaunchKernelGGL(k_query_anchor_off, dim3(nbq), dim3(256), 0, st, a_off.p, seq_off2.p, n_seq, n_kept, n_a, O.q_aoff.p);
	{ Downloads dl(st); dl.add(O.h_rep_len, rep_len.p, rep_len.n); dl.add(O.h_q_aoff, O.q_aoff.p, O.q_aoff.n); dl.wait(); }
	O.a.alloc(n_a ? n_a : 1);
	if (n_a == 0) return;

	// 4. sort each query's anchors by x.  Parallel stable segmented sort first; queries that contain equal keys
	//    are then re-sorted from the raw order by the sequential replay of radix_sort_128x (pga_sort_exact.h).
	const unsigned nba = (unsigned)((n_a + 255) / 256);
	DBuf<uint64_t> &x0 = O.raw_x, &y0 = O.raw_y, &x1 = O.srt_x, &y1 = O.srt_y;
	x0.alloc(n_a); y0.alloc(n_a); x1.alloc(n_a); y1.alloc(n_a);
	hipLaunchKernelGGL(k_anchor_pack, dim3(nbk), dim3(256), 0, st, n_kept, cnt.p, ub_off.p, ub.p, a_off.p, x0.p, y0.p);
	ub.release();            // (the upper-bound layout holds every occurrence, about twice the anchors: back to the arena -- in stream order -- before the sort's buffers are taken)
	{
		// anchors are already grouped by query, so a device-wide stable sort by x followed by a stable sort by the
		// query id (LSD order) equals a per-query sort, without the one-block-per-segment cost of a segmented sort
		DBuf<uint32_t> idx0(n_a), idx1(n_a), qk0(n_a), qk1(n_a), idx2(n_a);
		hipLaunchKernelGGL(k_iota32, dim3(nba), dim3(256), 0, st, idx0.p, n_a);
		// x = strand << 63 | target << 32 | position uses 1 + log2(group size) + log2(longest sequence) of its 64 bits: the sort runs over a
		// packed key of just those (25 bits for a pair of 5 Mbp genomes: four digit passes instead of eight)
		uint32_t max_len = 1; int64_t max_grp = 1;
		for (uint32_t l : S.len) max_len = std::max(max_len, l);
		for (int g = 0; g < S.n_grp; ++g) max_grp = std::max<int64_t>(max_grp, S.grp_off[(size_t)g + 1] - S.grp_off[(size_t)g]);
		int pos_bits = 1, rid_bits = 1;
		while (pos_bits < 32 && (1ULL << pos_bits) <= (uint64_t)max_len) ++pos_bits;
		while (rid_bits < 31 && (1LL << rid_bits) < max_grp) ++rid_bits;
		hipLaunchKernelGGL(k_anchor_key, dim3(nba), dim3(256), 0, st, x0.p, n_a, rid_bits, pos_bits, x1.p);
		DBuf<uint64_t> kc2(n_a);
		size_t tb = 0;
		PGA_HIP(rocprim::radix_sort_pairs(nullptr, tb, x1.p, kc2.p, idx0.p, idx1.p, n_a, 0, 1 + rid_bits + pos_bits, st));
		DBuf<uint8_t> tmp(tb ? tb : 1);
		PGA_HIP(rocprim::radix_sort_pairs(tmp.p, tb, x1.p, kc2.p, idx0.p, idx1.p, n_a, 0, 1 + rid_bits + pos_bits, st));
		if (n_seq > 1) {
			hipLaunchKernelGGL(k_query_of_anchor, dim3(nba), dim3(256), 0, st, idx1.p, O.q_aoff.p, n_seq, n_a, qk0.p);
			int bits = 1; while ((1LL << bits) < n_seq) ++bits;
			size_t tb2 = 0;
			PGA_HIP(rocprim::radix_sort_pairs(nullptr, tb2, qk0.p, qk1.p, idx1.p, idx2.p, n_a, 0, bits, st));
			DBuf<uint8_t> tmp2(tb2 ? tb2 : 1);
			PGA_HIP(rocprim::radix_sort_pairs(tmp2.p, tb2, qk0.p, qk1.p, idx1.p, idx2.p, n_a, 0, bits, st));
			hipLaunchKernelGGL(k_gather_xy, dim3(nba), dim3(256), 0, st, x0.p, y0.p, idx2.p, n_a, x1.p, y1.p);
		} else {
			hipLaunchKernelGGL(k_gather_xy, dim3(nba), dim3(256), 0, st, x0.p, y0.p, idx1.p, n_a, x1.p, y1.p);
		}
	}
	hipLaunchKernelGGL(k_join128, dim3(nba), dim3(256), 0, st, x1.p, y1.p, n_a, O.a.p);
	DBuf<uint32_t> &q_tie = O.q_tie; q_tie.alloc((size_t)n_seq); q_tie.zero(st);
	hipLaunchKernelGGL(k_tie_flags, dim3(nba), dim3(256), 0, st, x1.p, O.q_aoff.p, n_seq, n_a, (const u128*)nullptr, q_tie.p);
	// the stable sort above doubles as a hint for the replay: buckets without equal keys are copied from it instead of being walked
	DBuf<uint32_t> &dupc = O.dupc; dupc.alloc(n_a);
	{
		struct Dp { const uint64_t *x; };
		Dp dp{x1.p};
		auto flag_it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), [dp] __device__ (uint64_t i) { return (uint32_t)(i > 0 && dp.x[i] == dp.x[i - 1]); });
		size_t tb = 0;
		PGA_HIP(rocprim::inclusive_scan(nullptr, tb, flag_it, dupc.p, n_a, rocprim::plus<uint32_t>(), st));
		DBuf<uint8_t> tmp(tb ? tb : 1);
		PGA_HIP(rocprim::inclusive_scan(tmp.p, tb, flag_it, dupc.p, n_a, rocprim::plus<uint32_t>(), st));
	}
	O.exact = false;
	if (exact_order) seed_exact_order(O, nullptr, n_seq, st, tm);
	PGA_HIP(hipGetLastError());
	PGA_HIP(sync_stream(st));
	if (getenv("PGA_VERBOSE")) {
		std::vector<uint32_t> tf = q_tie.download(st); size_t nt = 0; for (uint32_t v : tf) nt += v;
		fprintf(stderr, "[pga]   seed: %llu anchors, %zu of %d queries hold equal anchor keys (%s)\n", (unsigned long long)n_a, nt, n_seq,
		        exact_order ? "sequential sort replay" : "stable order for now: the chaining stage asks for the reference's order where it matters");
	}
}

__global__ void k_need_tied(const uint32_t *__restrict__ q_tie, const uint32_t *__restrict__ need, int n_seq, uint32_t *__restrict__ out)
{
	const int q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q < n_seq) out[q] = q_tie[q] && (!need || need[q]) ? 1u : 0u;
}

void seed_exact_order(SeedResult &O, const uint32_t *d_need, int n_seq, hipStream_t st, Timers *tm)
{
	if (O.n_a == 0 || !O.q_tie.p || !O.raw_x.p) return;
	DBuf<uint32_t> fl((size_t)n_seq);
	hipLaunchKernelGGL(k_need_tied, dim3((unsigned)((n_seq + 255) / 256)), dim3(256), 0, st, O.q_tie.p, d_need, n_seq, fl.p);
	// queries whose anchors hold equal keys restart from the raw order and go through the replay of radix_sort_128x (pga_sort_replay.hip); the stable
	// sort doubles as its hint: buckets without equal keys are copied from it instead of being walked
	hipLaunchKernelGGL(k_copy_tied, dim3((unsigned)n_seq, 64), dim3(256), 0, st, n_seq, fl.p, O.q_aoff.p, O.raw_x.p, O.raw_y.p, O.a.p);
	const RsHint hint{O.srt_x.p, O.srt_y.p, O.dupc.p};
	replay_sort_segments(O.a.p, O.n_a, O.q_aoff.p, nullptr, n_seq, fl.p, st, tm, getenv("PGA_NO_SORT_HINT") ? nullptr : &hint);
	PGA_HIP(sync_stream(st));
	if (!d_need) O.exact = true;
}

bool exact_sorts_forced() { const char *e = getenv("PGA_EXACT_SORTS"); return e && *e && *e != '0'; }

} // namespace pga
