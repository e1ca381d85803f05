// pga_mash.hip -- SURVEY 8(f)-3, the guide tree: pangraph's mash-style distance between the input genomes and the
// neighbor-joining tree built from it, on the device.
//
//   minimizers_sketch   packages/pangraph/src/distance/mash/minimizer.rs:49-160  (hash: hash.rs:3-12; k = 15, w = 100)
//   mash_distance       packages/pangraph/src/distance/mash/mash_distance.rs:9-65
//   neighbor joining    packages/pangraph/src/tree/neighbor_joining.rs:16-103
//
// Kernels:
//   k_mash_chunks    one LANE per 4096-base chunk of a sequence runs the reference's window automaton (ring of w hashes in LDS,
//                    lane-interleaved: conflict-free).  A chunk starts w+k bases early without emitting: every threshold the
//                    automaton tests is <= w+k, the window holds w entries and the current minimum is the leftmost minimum of
//                    the window, so after that warm-up the state is the state of the sequential run.  Two passes (count, then
//                    write at the scanned offsets): the output order is the reference's.
//   distance         the reference counts, for every distinct hash value, every pair of sequences that holds it (O(m^2) per value:
//                    5e10 increments for 1000 related genomes).  Here: sort by value (rocPRIM), dense value ranks, one BIT per
//                    (sequence, value) and C = B * B^T over 64-bit words: |M_i and M_j| = sum_w popc(B[i][w] & B[j][w]), tiled
//                    through LDS.  The value axis is processed in slabs so that the bit matrix stays below a fixed budget.
//   k_nj             one workgroup runs the whole joining loop on the n x n matrix in place (alive list instead of compaction).
//                    The f64 sums follow ndarray 0.16.1's orders (see oracle/pgo_mash.c): a row's eight-accumulator unrolled sum
//                    and its left-to-right sum come out of one pass over the row; argmin = first minimum in row-major order.
#include "pga_common.h"
#include <type_traits>
#include <rocprim/rocprim.hpp>
#include <stdexcept>
#include <string>
#include <vector>
#include <chrono>
#include <cstdio>

namespace pga {

#define MASH_CHUNK 4096
#define MASH_MAX 0xffffffffffffffffULL

__device__ __forceinline__ uint64_t mash_hash(uint64_t x, uint64_t mask)
{
	x = (~x + (x << 21)) & mask;
	x = x ^ (x >> 24);
	x = (x + (x << 3) + (x << 8)) & mask;
	x = x ^ (x >> 14);
	x = (x + (x << 2) + (x << 4)) & mask;
	x = x ^ (x >> 28);
	x = (x + (x << 31)) & mask;
	return x;
}

struct MashChunk { uint32_t rid; uint32_t start; };

// MODE 0: cnt[chunk] = number of minimizers the chunk emits; MODE 1: they go to val/pos at off[chunk]; MODE 2: single pass -- they go to
// a staging area of `stage_cap` entries per chunk (cnt still receives the true count: an overflowing chunk sends everybody through
// modes 0 and 1).  RING: uint32_t when a hash and the strand bit fit 32 bits (k <= 15: half the LDS, twice the waves per CU).
template <int MODE, class RING>
__global__ __launch_bounds__(64)
void k_mash_chunks(PkBases bases, const uint64_t *__restrict__ seq_off, const uint32_t *__restrict__ seq_len, const MashChunk *__restrict__ chunks, uint32_t n_chunks,
                   int k, int w, uint32_t *__restrict__ cnt, const uint64_t *__restrict__ off, uint64_t *__restrict__ val, uint64_t *__restrict__ pos, uint32_t stage_cap)
{
	extern __shared__ uint64_t ring_raw[];
	RING *ring = reinterpret_cast<RING*>(ring_raw);          // ring[slot * 64 + lane]
	constexpr bool WRITE = MODE != 0;
	constexpr RING RMAX = (RING)~(RING)0;
	constexpr int SBIT = sizeof(RING) * 8 - 1;                // the strand bit of a stored entry
	const int lane = threadIdx.x;
	const uint32_t c = blockIdx.x * 64 + lane;
	if (c >= n_chunks) return;
	const MashChunk ch = chunks[c];
	const uint32_t len = seq_len[ch.rid];
	const uint64_t s0 = seq_off[ch.rid];                               // the sequence's first base in the packed store
	const uint64_t id = ch.rid;
	const uint64_t mask = (1ULL << (2 * k)) - 1, shift = 2 * (uint64_t)(k - 1);
	const uint32_t end = ch.start + MASH_CHUNK < len ? ch.start + MASH_CHUNK : len;
	const uint32_t warm = (uint32_t)(w + k);
	const uint32_t q0 = ch.start > warm ? ch.start - warm : 0;
	uint64_t fwd = 0, rev = 0;
	uint64_t min_v = MASH_MAX, min_p = MASH_MAX;
	uint32_t l = q0 > 0 ? warm : 0;                          // (>= w+k behaves like any longer run; an N inside the warm-up makes it exact)
	int bi = (int)(q0 % (uint32_t)w), mi = bi;               // (slot = position mod w, as in a run from position 0)
	for (int i = 0; i < w; ++i) ring[i * 64 + lane] = RMAX;
	uint64_t n_out = 0;
	const uint64_t o0 = MODE == 1 ? off[c] : MODE == 2 ? (uint64_t)c * stage_cap : 0;
	// position of the entry in slot i, seen from the current position p (slot bi was just written with p): p - ((bi - i) mod w)
	auto slot_pos = [&](int i, uint32_t p, RING v) -> uint64_t {
		// the position word of a window entry: id << 32 | locus << 1 | strand; the strand bit travels in bit 63 of the stored value
		const int age = bi >= i ? bi - i : bi - i + w;
		return (id << 32) | ((uint64_t)(p - (uint32_t)age + 1) << 1) | (uint64_t)(v >> SBIT);
	};
	auto emit = [&](uint64_t v, uint64_t pp, bool live) {
		if (live) { if (WRITE && (MODE == 1 || n_out < stage_cap)) { val[o0 + n_out] = v; pos[o0 + n_out] = pp; } ++n_out; }
	};
	for (uint32_t p = q0; p < end; ++p) {
		const bool live = p >= ch.start;
		const int cde = bases.at(s0 + (uint64_t)p);
		uint64_t nv = MASH_MAX, np = MASH_MAX;               // (stored values carry the strand in bit 63; MAX stays MAX)
		uint64_t st_bit = 0;
		if (cde >= 4) l = 0;
		else {
			fwd = ((fwd << 2) | (uint64_t)cde) & mask;
			rev = (rev >> 2) | ((uint64_t)(3 ^ cde) << shift);
			++l;
			if (l >= (uint32_t)k) {
				const uint64_t pp = (id << 32) | ((uint64_t)(p + 1) << 1);
				if (fwd <= rev) { nv = mash_hash(fwd, mask); np = pp; }
				else { nv = mash_hash(rev, mask); np = pp | 1; st_bit = 1ULL << 63; }
			}
		}
		ring[bi * 64 + lane] = nv == MASH_MAX ? RMAX : (RING)((RING)nv | (RING)((RING)(st_bit >> 63) << SBIT));   // a hash has at most 2k bits: the top bit of an entry is free
		auto same_as_min = [&](bool lv) {                    // window entries equal to the minimum at another position, oldest first
			for (int t = 1; t <= w; ++t) {
				int i = bi + t; if (i >= w) i -= w;
				const RING e = ring[i * 64 + lane];
				if (e == RMAX) continue;
				const uint64_t ev = (uint64_t)(RING)(e & (RING)~((RING)1 << SBIT));
				if (ev == min_v) { const uint64_t ep = slot_pos(i, p, e); if (ep != min_p) emit(ev, ep, lv); }
			}
		};
		if (l == (uint32_t)(w + k - 1) && min_v != MASH_MAX) same_as_min(live);
		if (nv < min_v) {
			if (l >= (uint32_t)(w + k) && min_v != MASH_MAX) emit(min_v, min_p, live);
			min_v = nv; min_p = np; mi = bi;
		} else if (bi == mi) {
			if (l >= (uint32_t)(w + k - 1) && min_v != MASH_MAX) emit(min_v, min_p, live);
			min_v = MASH_MAX;                                // (keeps its position)
			for (int t = 1; t <= w; ++t) {
				int i = bi + t; if (i >= w) i -= w;
				const RING e = ring[i * 64 + lane];
				if (e == RMAX) continue;
				const uint64_t ev = (uint64_t)(RING)(e & (RING)~((RING)1 << SBIT));
				if (ev < min_v) { mi = i; min_v = ev; min_p = slot_pos(i, p, e); }
			}
			if (l >= (uint32_t)(w + k - 1) && min_v != MASH_MAX) same_as_min(live);
		}
		if (++bi >= w) bi = 0;
	}
	if (end == len && min_v != MASH_MAX) emit(min_v, min_p, true);          // the last minimum of the sequence
	if (MODE != 1) cnt[c] = (uint32_t)n_out;
}

// staging -> final arrays (one wave per chunk)
__global__ void k_mash_unstage(const uint64_t *__restrict__ sval, const uint64_t *__restrict__ spos, const uint32_t *__restrict__ cnt, const uint64_t *__restrict__ off, uint32_t n_chunks, uint32_t stage_cap,
                               uint64_t *__restrict__ val, uint64_t *__restrict__ pos)
{
	const uint32_t c = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
	if (c >= n_chunks) return;
	const uint32_t n = cnt[c]; const uint64_t o = off[c], so = (uint64_t)c * stage_cap;
	for (uint32_t i = threadIdx.x & 63; i < n; i += 64) { val[o + i] = sval[so + i]; pos[o + i] = spos[so + i]; }
}

struct MashSketch { DBuf<uint64_t> val, pos; std::vector<uint64_t> seq_off; uint64_t n = 0; };

static void mash_sketch_all(const SeqSet &S, int k, int w, MashSketch &M, hipStream_t st)
{
	if (k < 1 || k >= 32) throw std::runtime_error("pga: mash sketch needs 0 < k < 32 (minimizer.rs:55)");
	if (w < 1 || w >= 256) throw std::runtime_error("pga: mash sketch needs 0 < w < 256 (minimizer.rs:56)");
	std::vector<MashChunk> chunks; std::vector<uint64_t> first((size_t)S.n_seq + 1, 0);
	for (int r = 0; r < S.n_seq; ++r) {
		first[(size_t)r] = chunks.size();
		for (uint32_t s0 = 0; s0 < S.len[(size_t)r]; s0 += MASH_CHUNK) chunks.push_back({(uint32_t)r, s0});
	}
	first[(size_t)S.n_seq] = chunks.size();
	M.seq_off.assign((size_t)S.n_seq + 1, 0);
	M.n = 0;
	if (chunks.empty()) return;
	const uint32_t nc = (uint32_t)chunks.size();
	DBuf<MashChunk> d_ch; d_ch.upload(chunks, st);
	DBuf<uint32_t> d_cnt(nc);
	const bool small = 2 * k + 1 <= 32;
	const size_t lds = (size_t)w * 64 * (small ? sizeof(uint32_t) : sizeof(uint64_t));
	static bool attr = false;
	if (!attr) {
		for (const void *f : {(const void*)k_mash_chunks<0, uint64_t>, (const void*)k_mash_chunks<1, uint64_t>, (const void*)k_mash_chunks<2, uint64_t>,
		                      (const void*)k_mash_chunks<0, uint32_t>, (const void*)k_mash_chunks<1, uint32_t>, (const void*)k_mash_chunks<2, uint32_t>})
			PGA_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 255 * 64 * 8));
		attr = true;
	}
	auto launch = [&](int mode, uint32_t *cnt, const uint64_t *off, uint64_t *val, uint64_t *pos, uint32_t cap) {
		const dim3 g((nc + 63) / 64), b(64);
#define PGA_MASH_GO(M, R) hipLaunchKernelGGL((k_mash_chunks<M, R>), g, b, lds, st, S.bases(), S.d_off.p, S.d_len.p, d_ch.p, nc, k, w, cnt, off, val, pos, cap)
		if (small) { if (mode == 0) PGA_MASH_GO(0, uint32_t); else if (mode == 1) PGA_MASH_GO(1, uint32_t); else PGA_MASH_GO(2, uint32_t); }
		else { if (mode == 0) PGA_MASH_GO(0, uint64_t); else if (mode == 1) PGA_MASH_GO(1, uint64_t); else PGA_MASH_GO(2, uint64_t); }
#undef PGA_MASH_GO
		PGA_HIP(hipGetLastError());
	};
	// single pass into a staging area of 4x the expected density per chunk; a chunk that overflows it (long arrays of equal hashes)
	// sends the batch through the count + write passes instead
	const uint32_t stage_cap = (uint32_t)std::max(64, 8 * MASH_CHUNK / (w + 1) + 32);
	DBuf<uint64_t> sval, spos;
	const bool staged = getenv("PGA_MASH_TWO_PASS") == nullptr && (uint64_t)nc * stage_cap * 16 <= ((uint64_t)24 << 30);
	if (staged) { sval.alloc((size_t)nc * stage_cap); spos.alloc((size_t)nc * stage_cap); launch(2, d_cnt.p, nullptr, sval.p, spos.p, stage_cap); }
	else launch(0, d_cnt.p, nullptr, nullptr, nullptr, 0);
	std::vector<uint32_t> cnt = d_cnt.download(st);
	std::vector<uint64_t> off((size_t)nc + 1, 0);
	bool overflow = false;
	for (uint32_t i = 0; i < nc; ++i) { off[(size_t)i + 1] = off[i] + cnt[i]; overflow |= cnt[i] > stage_cap; }
	for (int r = 0; r <= S.n_seq; ++r) M.seq_off[(size_t)r] = off[first[(size_t)r]];
	M.n = off[nc];
	DBuf<uint64_t> d_off; d_off.upload(off, st);
	M.val.alloc((size_t)M.n + 1); M.pos.alloc((size_t)M.n + 1);
	if (staged && !overflow) hipLaunchKernelGGL(k_mash_unstage, dim3((nc + 3) / 4), dim3(256), 0, st, sval.p, spos.p, d_cnt.p, d_off.p, nc, stage_cap, M.val.p, M.pos.p);
	else launch(1, nullptr, d_off.p, M.val.p, M.pos.p, 0);
	PGA_HIP(hipGetLastError());
	PGA_HIP(sync_stream(st));
}

// ---- distance ----
__global__ void k_mash_sid(const uint64_t *__restrict__ pos, uint64_t n, uint32_t *__restrict__ sid)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) sid[i] = (uint32_t)(pos[i] >> 32);
}
__global__ void k_mash_heads(const uint64_t *__restrict__ v, uint64_t n, uint32_t *__restrict__ head)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) head[i] = (i == 0 || v[i] != v[i - 1]) ? 1u : 0u;
}
// rank[i] = dense index of the value of sorted record i (inclusive scan of the head flags, minus one)
__global__ void k_mash_set_bits(const uint32_t *__restrict__ rank_incl, const uint32_t *__restrict__ sid, uint64_t n, uint64_t v0, uint64_t v1, unsigned long long *__restrict__ B, uint64_t words)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t v = (uint64_t)rank_incl[i] - 1;
	if (v < v0 || v >= v1) return;
	const uint64_t b = v - v0;
	atomicOr(&B[(uint64_t)sid[i] * words + (b >> 6)], 1ULL << (b & 63));
}
#define MT 16          // sequences per tile side
#define MW 64          // words per LDS step
__global__ __launch_bounds__(MT * MT)
void k_mash_count(const unsigned long long *__restrict__ B, uint64_t words, int n, unsigned long long *__restrict__ C)
{
	__shared__ unsigned long long sa[MT][MW + 1], sb[MT][MW + 1];
	const int ti = blockIdx.y, tj = blockIdx.x;
	if (tj < ti) return;                                  // the upper triangle only (the count is symmetric)
	const int tx = threadIdx.x % MT, ty = threadIdx.x / MT;
	const int i = ti * MT + ty, j = tj * MT + tx;
	unsigned long long acc = 0;
	for (uint64_t w0 = 0; w0 < words; w0 += MW) {
		for (int e = threadIdx.x; e < MT * MW; e += MT * MT) {
			const int r = e / MW, cw = e % MW;
			const uint64_t wd = w0 + cw;
			const int ri = ti * MT + r, rj = tj * MT + r;
			sa[r][cw] = (ri < n && wd < words) ? B[(uint64_t)ri * words + wd] : 0ULL;
			sb[r][cw] = (rj < n && wd < words) ? B[(uint64_t)rj * words + wd] : 0ULL;
		}
		__syncthreads();
#pragma unroll 8
		for (int cw = 0; cw < MW; ++cw) acc += (unsigned long long)__popcll(sa[ty][cw] & sb[tx][cw]);
		__syncthreads();
	}
	if (i < n && j < n && i <= j) C[(uint64_t)i * n + j] += acc;
}
__global__ void k_mash_dist(const unsigned long long *__restrict__ C, int n, double *__restrict__ D, int *__restrict__ bad)
{
	const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= (uint64_t)n * n) return;
	const int i = (int)(e / n), j = (int)(e % n);
	if (i == j) { D[e] = 0.0; if (C[e] == 0) atomicMin(bad, i); return; }
	const int a = i < j ? i : j, b = i < j ? j : i;                     // mash_distance.rs:57-60: 1 - shared / own(a), mirrored
	D[e] = 1.0 - (double)C[(uint64_t)a * n + b] / (double)C[(uint64_t)a * n + a];
}

static void mash_distance_dev(const SeqSet &S, int k, int w, DBuf<double> &D, hipStream_t st, double *t_sketch_ms = nullptr)
{
	const int n = S.n_seq;
	MashSketch M;
	mash_sketch_all(S, k, w, M, st);
	for (int r = 0; r < n; ++r) if (M.seq_off[(size_t)r + 1] == M.seq_off[(size_t)r])
		throw std::runtime_error("pga: no minimizer found for sequence " + std::to_string(r) + " during mash distance evaluation (mash_distance.rs:19-20)");
	const uint64_t N = M.n;
	if (getenv("PGA_VERBOSE")) fprintf(stderr, "[pga]   mash sketch: %llu minimizers (k %d, w %d)\n", (unsigned long long)N, k, w);
	DBuf<uint32_t> sid(N), sid2(N); DBuf<uint64_t> v2(N);
	hipLaunchKernelGGL(k_mash_sid, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, M.pos.p, N, sid.p);
	size_t tmp_bytes = 0;
	PGA_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, M.val.p, v2.p, sid.p, sid2.p, N, 0, 2 * k, st));
	DBuf<uint8_t> tmp(tmp_bytes + 16);
	PGA_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, M.val.p, v2.p, sid.p, sid2.p, N, 0, 2 * k, st));
	DBuf<uint32_t> head(N), rank(N);
	hipLaunchKernelGGL(k_mash_heads, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, v2.p, N, head.p);
	size_t tmp2 = 0;
	PGA_HIP(rocprim::inclusive_scan(nullptr, tmp2, head.p, rank.p, N, rocprim::plus<uint32_t>(), st));
	DBuf<uint8_t> tmpb(tmp2 + 16);
	PGA_HIP(rocprim::inclusive_scan(tmpb.p, tmp2, head.p, rank.p, N, rocprim::plus<uint32_t>(), st));
	uint32_t V = 0;
	PGA_HIP(hipMemcpyAsync(&V, rank.p + (N - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
	// the bit matrix in slabs of the value axis: n x slab bits, at most ~2 GB
	const uint64_t budget_words = (uint64_t)(getenv("PGA_MASH_SLAB_MB") ? atof(getenv("PGA_MASH_SLAB_MB")) * (1 << 20) : 2048.0 * (1 << 20)) / 8;
	uint64_t slab_words = std::max<uint64_t>(MW, budget_words / (uint64_t)std::max(1, n));
	slab_words = slab_words / MW * MW;
	const uint64_t all_words = ((uint64_t)V + 63) / 64;
	if (slab_words > (all_words + MW - 1) / MW * MW) slab_words = (all_words + MW - 1) / MW * MW;
	DBuf<unsigned long long> B((size_t)n * slab_words), C((size_t)n * n);
	C.zero(st);
	const int tiles = (n + MT - 1) / MT;
	if (getenv("PGA_VERBOSE")) fprintf(stderr, "[pga]   mash distance: %u distinct values, bit matrix %d x %llu words per slab\n", V, n, (unsigned long long)slab_words);
	for (uint64_t v0 = 0; v0 < V; v0 += slab_words * 64) {
		B.zero(st);
		hipLaunchKernelGGL(k_mash_set_bits, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, rank.p, sid2.p, N, v0, v0 + slab_words * 64, B.p, slab_words);
		hipLaunchKernelGGL(k_mash_count, dim3(tiles, tiles), dim3(MT * MT), 0, st, B.p, slab_words, n, C.p);
	}
	D.alloc((size_t)n * n);
	DBuf<int> bad(1); { const int big = 0x7fffffff; PGA_HIP(hipMemcpyAsync(bad.p, &big, sizeof(int), hipMemcpyHostToDevice, st)); }
	hipLaunchKernelGGL(k_mash_dist, dim3((unsigned)(((uint64_t)n * n + 255) / 256)), dim3(256), 0, st, C.p, n, D.p, bad.p);
	PGA_HIP(hipGetLastError());
	const int b = bad.download(st)[0];
	if (b != 0x7fffffff) throw std::runtime_error("pga: no self-hit found for sequence " + std::to_string(b) + " (mash_distance.rs:51-54)");
}

// ---- neighbor joining: one workgroup, the matrix in place ----
#define NJ_NT 1024
#define NJ_MAX 2048
// BIG = false: n <= NJ_MAX, the per-index state lives in LDS.  BIG = true: any n -- the same loop with that state in device memory
// (`scratch`: n words of alive, n of node, 2n doubles), still one workgroup: the joins are sequential and a join of n > 2048 rows has
// enough work per step (n*n Q entries) for 1024 threads; the reference (tree/neighbor_joining.rs:16-103) has no limit, so neither has this.
template <bool BIG>
__global__ __launch_bounds__(NJ_NT)
void k_nj(int n, double *__restrict__ D, int32_t *__restrict__ merges, int *__restrict__ status, unsigned char *__restrict__ scratch, double d_max, int32_t *__restrict__ near)
{
	using alive_t = typename std::conditional<BIG, uint32_t, uint16_t>::type;
	__shared__ uint16_t alive_s[BIG ? 1 : NJ_MAX];        // physical index of logical row/column
	__shared__ int32_t node_s[BIG ? 1 : NJ_MAX];          // tree node of logical index
	__shared__ double s0_s[BIG ? 1 : NJ_MAX], s1_s[BIG ? 1 : NJ_MAX]; // per logical index: left-to-right sum (= the column sum of the symmetric matrix) and unrolled sum of its row
	__shared__ double rq[NJ_NT]; __shared__ unsigned long long ri[NJ_NT];
	// The runner-up: the smallest Q of a DIFFERENT unordered pair (the transposed entry of the best is the same join).  When it lies within
	// the error a different summation order could make -- the row and column sums of neighbor_joining.rs:60-75 are ndarray's, restated
	// above, not pinned against ndarray itself -- the join is counted in near[0] (first such join in near[1]): the caller's flag.
	__shared__ double rq2[NJ_NT]; __shared__ unsigned long long ri2[NJ_NT];
	int32_t n_near = 0, first_near = -1;
	double *s0 = BIG ? reinterpret_cast<double*>(scratch) : s0_s;
	double *s1 = BIG ? reinterpret_cast<double*>(scratch) + n : s1_s;
	int32_t *node = BIG ? reinterpret_cast<int32_t*>(scratch + (size_t)16 * n) : node_s;
	alive_t *alive = BIG ? reinterpret_cast<alive_t*>(scratch + (size_t)20 * n) : reinterpret_cast<alive_t*>(alive_s);
	const int tid = threadIdx.x;
	for (int i = tid; i < n; i += NJ_NT) alive[i] = (alive_t)i, node[i] = i;
	__syncthreads();
	int m = n, t = 0;
	while (m > 2) {
		// row sums in both orders
		for (int r = tid; r < m; r += NJ_NT) {
			const double *row = D + (size_t)alive[r] * n;
			const int m8 = m & ~7;
			double a = 0.0, p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0;
			int c = 0;
			for (; c < m8; c += 8) {
				const double x0 = row[alive[c]], x1 = row[alive[c + 1]], x2 = row[alive[c + 2]], x3 = row[alive[c + 3]];
				const double x4 = row[alive[c + 4]], x5 = row[alive[c + 5]], x6 = row[alive[c + 6]], x7 = row[alive[c + 7]];
				a = a + x0; a = a + x1; a = a + x2; a = a + x3; a = a + x4; a = a + x5; a = a + x6; a = a + x7;
				p0 = p0 + x0; p1 = p1 + x1; p2 = p2 + x2; p3 = p3 + x3; p4 = p4 + x4; p5 = p5 + x5; p6 = p6 + x6; p7 = p7 + x7;
			}
			double u = 0.0;
			u = u + (p0 + p4); u = u + (p1 + p5); u = u + (p2 + p6); u = u + (p3 + p7);
			for (; c < m; ++c) { const double x = row[alive[c]]; a = a + x; u = u + x; }
			s0[r] = a; s1[r] = u;
		}
		__syncthreads();
		// Q and its first minimum in row-major order
		// (best, runner-up of another pair) <- (best, runner-up) + candidate (q, idx); by value: nothing here may end up addressable
		struct Two { double b; unsigned long long bi; double s; unsigned long long si; };
		auto same_pair = [](unsigned long long a, unsigned long long b) {
			return a != ~0ULL && b != ~0ULL && (a == b || ((unsigned)(a >> 32) == (unsigned)b && (unsigned)a == (unsigned)(b >> 32)));
		};
		auto take = [&](Two t, double q, unsigned long long idx) -> Two {
			if (idx == ~0ULL) return t;
			if (q < t.b || (q == t.b && idx < t.bi)) { if (!same_pair(idx, t.bi)) { t.s = t.b; t.si = t.bi; } t.b = q; t.bi = idx; }
			else if (q < t.s && !same_pair(idx, t.bi)) { t.s = q; t.si = idx; }
			return t;
		};
		Two mine{INFINITY, ~0ULL, INFINITY, ~0ULL};
		const double mm2 = (double)m - 2.0;
		for (int r = tid; r < m; r += NJ_NT) {
			const double *row = D + (size_t)alive[r] * n;
			const double sr = s1[r];
			for (int c = 0; c < m; ++c) {
				if (c == r) continue;
				const double q = (mm2 * row[alive[c]] - s0[c]) - sr;
				if (q < mine.s) mine = take(mine, q, (unsigned long long)r << 32 | (unsigned)c);     // (mine.s >= mine.b: nothing above the runner-up matters; an equal best comes later in row-major order)
			}
		}
		rq[tid] = mine.b; ri[tid] = mine.bi; rq2[tid] = mine.s; ri2[tid] = mine.si;
		__syncthreads();
		for (int sft = NJ_NT / 2; sft > 0; sft >>= 1) {
			if (tid < sft) {
				Two t{rq[tid], ri[tid], rq2[tid], ri2[tid]};
				t = take(t, rq[tid + sft], ri[tid + sft]);
				t = take(t, rq2[tid + sft], ri2[tid + sft]);
				rq[tid] = t.b; ri[tid] = t.bi; rq2[tid] = t.s; ri2[tid] = t.si;
			}
			__syncthreads();
		}
		const unsigned long long best = ri[0];
		if (best == ~0ULL) { if (tid == 0) *status = -1; return; }
		{
			// |error of a sum of m terms of size <= 1.5 d_max| <= m * 2^-53 * 1.5 m d_max; Q holds two of them and (m - 2) * d
			const double tol = 8.0 * (double)m * (double)m * 1.1102230246251565e-16 * d_max;
			// (m = 4 and m = 3 are not counted: there the Q of complementary pairs -- of all three pairs -- are EQUAL in exact arithmetic,
			// so the last two joins of every tree rest on the rounding of the sums; see include/pga_align.h)
			if (m > 4 && ri2[0] != ~0ULL && rq2[0] - rq[0] <= tol) { if (n_near == 0) first_near = t; ++n_near; }
		}
		int bi = (int)(best >> 32), bj = (int)(unsigned)best;
		const int i = bi < bj ? bi : bj, j = bi < bj ? bj : bi;
		const int pi = alive[i], pj = alive[j];
		const double dij = D[(size_t)pi * n + pj];
		__syncthreads();
		// the joined node takes row and column i: 0.5 * ((D[i][c] + D[j][c]) - D[i][j])
		for (int c = tid; c < m; c += NJ_NT) {
			const int pc = alive[c];
			const double dn = 0.5 * ((D[(size_t)pi * n + pc] + D[(size_t)pj * n + pc]) - dij);
			s0[c] = dn;                                  // (staged: row j must be read before row i changes where c == j)
		}
		__syncthreads();
		for (int c = tid; c < m; c += NJ_NT) { const int pc = alive[c]; D[(size_t)pi * n + pc] = s0[c]; D[(size_t)pc * n + pi] = s0[c]; }
		__syncthreads();
		if (tid == 0) { D[(size_t)pi * n + pi] = 0.0; merges[2 * t] = node[i]; merges[2 * t + 1] = node[j]; node[i] = n + t; }
		__syncthreads();
		// remove logical index j
		alive_t av = 0; int32_t nv = 0;
		for (int base = j; base < m - 1; base += NJ_NT) {
			const int c = base + tid;
			if (c < m - 1) { av = alive[c + 1]; nv = node[c + 1]; }
			__syncthreads();
			if (c < m - 1) { alive[c] = av; node[c] = nv; }
			__syncthreads();
		}
		--m; ++t;
		if (BIG) __threadfence();
		else __threadfence_block();
		__syncthreads();
	}
	if (tid == 0) { merges[2 * t] = node[0]; merges[2 * t + 1] = node[1]; *status = 0; if (near) { near[0] = n_near; near[1] = first_near; } }
}

static thread_local int32_t t_nj_near[2] = {0, -1};
void nj_last_near(int32_t *count, int32_t *first) { *count = t_nj_near[0]; *first = t_nj_near[1]; }

static void nj_dev(int n, double *d_D, std::vector<int32_t> &merges, hipStream_t st)
{
	merges.assign((size_t)std::max(0, n - 1) * 2, 0);
	t_nj_near[0] = 0, t_nj_near[1] = -1;
	if (n < 2) return;
	// the scale of the matrix, for the near-tie bound
	double d_max = 0.0;
	{
		DBuf<double> d_mx(1);
		size_t tb = 0;
		auto it = rocprim::make_transform_iterator(d_D, [] __device__ (double v) { return v < 0 ? -v : v; });
		PGA_HIP(rocprim::reduce(nullptr, tb, it, d_mx.p, 0.0, (size_t)n * n, rocprim::maximum<double>(), st));
		DBuf<uint8_t> tmp(tb ? tb : 1);
		PGA_HIP(rocprim::reduce(tmp.p, tb, it, d_mx.p, 0.0, (size_t)n * n, rocprim::maximum<double>(), st));
		d_max = d_mx.download(st)[0];
	}
	DBuf<int32_t> d_near(2); d_near.zero(st);
	DBuf<int32_t> d_m((size_t)(n - 1) * 2); DBuf<int> d_s(1);
	{ const int one = 1; PGA_HIP(hipMemcpyAsync(d_s.p, &one, sizeof(int), hipMemcpyHostToDevice, st)); }
	DBuf<unsigned char> scratch;
	if (n > NJ_MAX) {
		scratch.alloc((size_t)24 * n + 64);
		hipLaunchKernelGGL(k_nj<true>, dim3(1), dim3(NJ_NT), 0, st, n, d_D, d_m.p, d_s.p, scratch.p, d_max, d_near.p);
	} else hipLaunchKernelGGL(k_nj<false>, dim3(1), dim3(NJ_NT), 0, st, n, d_D, d_m.p, d_s.p, (unsigned char*)nullptr, d_max, d_near.p);
	PGA_HIP(hipGetLastError());
	if (d_s.download(st)[0] != 0) throw std::runtime_error("pga: neighbor joining found no pair to join (the distance matrix holds NaN or infinity)");
	merges = d_m.download(st);
	{ std::vector<int32_t> nr = d_near.download(st); t_nj_near[0] = nr[0]; t_nj_near[1] = nr[1]; }
}

// ---------------------------------------------------------------- entry points used by pga_api.cpp
void mash_stage_sketch(int n, const char *const *seqs, const uint32_t *lens, int k, int w, std::vector<uint64_t> &val, std::vector<uint64_t> &pos, std::vector<uint64_t> &off)
{
	SeqSet S; const int64_t one_grp[2] = {0, n};
	upload_seqs(S, n, seqs, lens, nullptr, 1, one_grp, 0);
	MashSketch M; mash_sketch_all(S, k, w, M, 0);
	val = M.val.download(0); val.resize(M.n);
	pos = M.pos.download(0); pos.resize(M.n);
	off = M.seq_off;
}

void mash_distance_host(int n, const char *const *seqs, const uint32_t *lens, int k, int w, double *dist, int32_t *merges)
{
	if (n <= 0) return;
	const bool verbose = getenv("PGA_VERBOSE") != nullptr;
	auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = now();
	SeqSet S; const int64_t one_grp[2] = {0, n};
	upload_seqs(S, n, seqs, lens, nullptr, 1, one_grp, 0);
	PGA_HIP(sync_stream(0));
	const double t1 = now();
	DBuf<double> D;
	mash_distance_dev(S, k, w, D, 0);
	PGA_HIP(sync_stream(0));
	const double t2 = now();
	if (dist) { PGA_HIP(hipMemcpyAsync(dist, D.p, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, 0)); PGA_HIP(sync_stream(0)); }
	if (merges) { std::vector<int32_t> m; nj_dev(n, D.p, m, 0); if (!m.empty()) memcpy(merges, m.data(), m.size() * sizeof(int32_t)); }
	if (verbose) fprintf(stderr, "[pga] guide tree: %d sequences, %.3f Gbp: hand-over %.3f s, sketch + distance %.3f s, neighbor joining %.3f s\n", n, S.total * 1e-9, t1 - t0, t2 - t1, now() - t2);
}

void nj_host(int n, const double *dist, int32_t *merges)
{
	if (n < 2) return;
	// k_nj takes ndarray's sum_axis(Axis(0)) (column sums) from the row pass, which is the same number only for a symmetric matrix --
	// what mash_distance produces (mash_distance.rs:56-63 fills both halves).  A caller's own matrix is checked instead of trusted.
	for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j)
		if (!(dist[(size_t)i * n + j] == dist[(size_t)j * n + i]))
			throw std::runtime_error("pga_guide_tree_nj: the distance matrix is not symmetric at (" + std::to_string(i) + ", " + std::to_string(j) + ")");
	DBuf<double> D; D.upload(dist, (size_t)n * n, 0);
	std::vector<int32_t> m; nj_dev(n, D.p, m, 0);
	memcpy(merges, m.data(), m.size() * sizeof(int32_t));
}

} // namespace pga
