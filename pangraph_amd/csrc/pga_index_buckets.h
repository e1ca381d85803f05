// pga_index_buckets.h -- the minimizer index of a batch WITHOUT a device-wide sort (candidate route: PGA_INDEX_BUCKETS=1, see the status note below).
//
// What mm_idx_get() returns is all that is observable of minimap2's index (packages/minimap2-sys/minimap2/index.c:84-98,252: the occurrence words
// y = rid<<32|pos<<1|strand of a minimizer hash, ascending); the order of the KEYS is free (DESIGN.md section 9 b).  So the minimizers of a batch do not
// have to be sorted by (group, hash) -- six digit passes of a radix sort over 44-46 bits, 0.4 s of kernel time and ~4.5 k dispatches per build step --
// they only have to be brought together by (group, hash) with y ascending inside a key, and the lists of a group should stay together:
//   1. every group g gets 2^bits[g] buckets by the top hash bits, sized so that a bucket holds ~1 k minimizers (the hash is minimap2's invertible
//      integer hash of the k-mer: its top bits are uniform); the buckets of the batch are numbered group by group;
//   2. k_ixb_count / k_ixb_scan / k_ixb_scatter: one counting pass and one scatter pass over the minimizers in tiles of 16 k (LDS histogram of the
//      tile's buckets, ONE global atomic per non-empty bin and tile), exact bucket offsets in between; the order inside a bucket is whatever the
//      atomics gave;
//   3. k_ixb_sort: one workgroup per bucket sorts its records in LDS by (composite key, original index) -- the minimizers arrive ordered by (rid,
//      pos), so the original index is the y order -- writes the occurrence lists and counts the bucket's keys;
//   4. k_ixb_scan over the key counts, k_ixb_groups: key ids, list offsets, group of every key, key id of every minimizer.
// Three passes over the data instead of six + flags + scan; 7 dispatches + one memset instead of 27.  A bucket of more than IXB_CAP records (a k-mer
// repeated thousands of times inside one group) is reported by the scan and the caller takes the sort route for the batch.
//
// STATUS: written in round 5 without a device to run it on.  Its LOGIC is checked under dev/emu/hip_emu.h (every workgroup as fibers on the host:
// tests/test_index_buckets_emu.py -- random batches with empty, tiny and large groups, repeated k-mers, an overflowing bucket, tiles that span more
// buckets than the LDS histogram holds).  On an MI355X it has run ONCE: __graft_entry__.smoke() under PGA_INDEX_BUCKETS=1, records identical to the
// oracle (two small batches); tests/test_gpu_zz_candidates.py holds it against the sort route at size in the driver's suite (tolerant: xfail, not
// strict).  It is not reachable unless PGA_INDEX_BUCKETS=1 is set, has never been timed, and no parity or speed claim rests on it.
#pragma once
#ifndef PGA_EMU
#include "pga_common.h"
#endif

namespace pga {

struct IxbGroup { uint32_t mz_begin, bucket_base, bits, pad; };     // per group; entry n_grp is the sentinel {n, number of buckets, 0, 0}

constexpr uint32_t IXB_TILE = 16384;       // minimizers per workgroup in the counting and scatter passes
constexpr uint32_t IXB_NT = 256;
#ifndef PGA_IXB_HB
#define PGA_IXB_HB 8192
#endif
constexpr uint32_t IXB_HB = PGA_IXB_HB;    // LDS histogram bins of a tile (a tile whose groups span more buckets uses global atomics per record; the emulation test builds with 16 too)
constexpr uint32_t IXB_CAP = 4096;         // records a bucket may hold
constexpr uint32_t IXB_TARGET = 1024;      // records per bucket aimed at

__device__ __forceinline__ uint64_t ixb_hash(uint64_t x, int hash_bits) { return (x >> 8) & (hash_bits >= 64 ? ~0ULL : (1ULL << hash_bits) - 1); }
__device__ __forceinline__ uint32_t ixb_bucket(const IxbGroup &G, uint64_t h, int hash_bits) { return G.bucket_base + (G.bits ? (uint32_t)(h >> (hash_bits - (int)G.bits)) : 0u); }
__device__ __forceinline__ uint32_t ixb_min(uint32_t a, uint32_t b) { return a < b ? a : b; }

// the buckets a tile's records can fall into: those of the groups of its first and of its last record (groups are consecutive)
struct IxbSpan { uint32_t lo, span; };
__device__ __forceinline__ IxbSpan ixb_tile_span(const u128 *__restrict__ mz, uint32_t t0, uint32_t t1, const uint32_t *__restrict__ grp_of_seq, const IxbGroup *__restrict__ gt)
{
	const uint32_t g0 = grp_of_seq[mz[t0].y >> 32], g1 = grp_of_seq[mz[t1 - 1].y >> 32];
	IxbSpan s; s.lo = gt[g0].bucket_base; s.span = gt[g1 + 1].bucket_base - s.lo;
	return s;
}

// cnt[b] += records of bucket b (cnt zeroed by the caller)
__global__ __launch_bounds__(256)
void k_ixb_count(const u128 *__restrict__ mz, uint32_t n, const uint32_t *__restrict__ grp_of_seq, const IxbGroup *__restrict__ gt, int hash_bits, uint32_t *__restrict__ cnt)
{
	__shared__ uint32_t h[IXB_HB];
	const uint32_t tid = threadIdx.x, t0 = blockIdx.x * IXB_TILE, t1 = ixb_min(n, t0 + IXB_TILE);
	const IxbSpan S = ixb_tile_span(mz, t0, t1, grp_of_seq, gt);
	const bool lds = S.span <= IXB_HB;
	if (lds) for (uint32_t b = tid; b < S.span; b += IXB_NT) h[b] = 0;
	__syncthreads();
	for (uint32_t i = t0 + tid; i < t1; i += IXB_NT) {
		const u128 m = mz[i];
		const uint32_t b = ixb_bucket(gt[grp_of_seq[m.y >> 32]], ixb_hash(m.x, hash_bits), hash_bits);
		if (lds) atomicAdd(&h[b - S.lo], 1u); else atomicAdd(&cnt[b], 1u);
	}
	__syncthreads();
	if (lds) for (uint32_t b = tid; b < S.span; b += IXB_NT) { const uint32_t c = h[b]; if (c) atomicAdd(&cnt[S.lo + b], c); }
}

// off[0 .. nb] = exclusive prefix sums of cnt[0 .. nb); flags[0] = max(flags[0], largest count) when it exceeds cap, flags[1] = total.  ONE workgroup of 1024.
__global__ __launch_bounds__(1024)
void k_ixb_scan(const uint32_t *__restrict__ cnt, uint32_t nb, uint32_t *__restrict__ off, uint32_t cap, uint32_t *__restrict__ flags)
{
	__shared__ uint32_t part[1024];
	const uint32_t tid = threadIdx.x, per = (nb + 1023) / 1024;
	const uint32_t a = ixb_min(nb, tid * per), e = ixb_min(nb, a + per);
	uint32_t s = 0, mx = 0;
	for (uint32_t i = a; i < e; ++i) { const uint32_t c = cnt[i]; s += c; if (c > mx) mx = c; }
	part[tid] = s;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) {
		const uint32_t v = tid >= d ? part[tid - d] : 0;
		__syncthreads();
		part[tid] += v;
		__syncthreads();
	}
	uint32_t run = part[tid] - s;
	for (uint32_t i = a; i < e; ++i) { off[i] = run; run += cnt[i]; }
	if (tid == 1023) { off[nb] = part[1023]; flags[1] = part[1023]; }
	if (mx > cap) atomicMax(&flags[0], mx);
}

// the records of every bucket to [off[b], off[b + 1]) of (sck, sy, so), in the order the atomics give (cursor zeroed by the caller)
__global__ __launch_bounds__(256)
void k_ixb_scatter(const u128 *__restrict__ mz, uint32_t n, const uint32_t *__restrict__ grp_of_seq, const IxbGroup *__restrict__ gt, int hash_bits,
                   const uint32_t *__restrict__ off, uint32_t *__restrict__ cursor, uint64_t *__restrict__ sck, uint64_t *__restrict__ sy, uint32_t *__restrict__ so)
{
	__shared__ uint32_t h[IXB_HB];             // first the tile's count per bin, then the next free position of the bin in the output
	const uint32_t tid = threadIdx.x, t0 = blockIdx.x * IXB_TILE, t1 = ixb_min(n, t0 + IXB_TILE);
	const IxbSpan S = ixb_tile_span(mz, t0, t1, grp_of_seq, gt);
	const bool lds = S.span <= IXB_HB;
	if (lds) {
		for (uint32_t b = tid; b < S.span; b += IXB_NT) h[b] = 0;
		__syncthreads();
		for (uint32_t i = t0 + tid; i < t1; i += IXB_NT) {
			const u128 m = mz[i];
			atomicAdd(&h[ixb_bucket(gt[grp_of_seq[m.y >> 32]], ixb_hash(m.x, hash_bits), hash_bits) - S.lo], 1u);
		}
		__syncthreads();
		for (uint32_t b = tid; b < S.span; b += IXB_NT) { const uint32_t c = h[b]; if (c) h[b] = off[S.lo + b] + atomicAdd(&cursor[S.lo + b], c); }
		__syncthreads();
	}
	for (uint32_t i = t0 + tid; i < t1; i += IXB_NT) {
		const u128 m = mz[i];
		const uint32_t g = grp_of_seq[m.y >> 32];
		const uint64_t hh = ixb_hash(m.x, hash_bits);
		const uint32_t b = ixb_bucket(gt[g], hh, hash_bits);
		const uint32_t pos = lds ? atomicAdd(&h[b - S.lo], 1u) : off[b] + atomicAdd(&cursor[b], 1u);
		sck[pos] = hash_bits >= 64 ? hh : ((uint64_t)g << hash_bits | hh);
		sy[pos] = m.y; so[pos] = i;
	}
}

// one workgroup per bucket: its records sorted by (composite key, original index); sorted keys and original indices, the occurrence lists, the
// number of distinct keys of the bucket.  A bucket over IXB_CAP writes nothing (k_ixb_scan has reported it).
__global__ __launch_bounds__(256)
void k_ixb_sort(const uint32_t *__restrict__ off, const uint64_t *__restrict__ sck, const uint64_t *__restrict__ sy, const uint32_t *__restrict__ so,
                uint64_t *__restrict__ ck2, uint32_t *__restrict__ orig2, uint64_t *__restrict__ occ, uint32_t *__restrict__ nk)
{
	__shared__ uint64_t key[IXB_CAP];
	__shared__ uint32_t org[IXB_CAP];
	__shared__ uint16_t slot[IXB_CAP];
	__shared__ uint32_t heads;
	const uint32_t tid = threadIdx.x, b = blockIdx.x, s = off[b], c = off[b + 1] - s;
	if (c == 0 || c > IXB_CAP) { if (tid == 0) nk[b] = 0; return; }
	uint32_t P = 2; while (P < c) P <<= 1;
	for (uint32_t j = tid; j < P; j += IXB_NT) {
		if (j < c) { key[j] = sck[s + j]; org[j] = so[s + j]; slot[j] = (uint16_t)j; }
		else { key[j] = ~0ULL; org[j] = ~0u; slot[j] = 0; }
	}
	if (tid == 0) heads = 0;
	__syncthreads();
	for (uint32_t k = 2; k <= P; k <<= 1) for (uint32_t j = k >> 1; j > 0; j >>= 1) {
		for (uint32_t i = tid; i < P; i += IXB_NT) {
			const uint32_t x = i ^ j;
			if (x > i) {
				const bool up = (i & k) == 0;
				const uint64_t ka = key[i], kb = key[x];
				const uint32_t oa = org[i], ob = org[x];
				const bool gt = ka > kb || (ka == kb && oa > ob);
				if (gt == up) { key[i] = kb; key[x] = ka; org[i] = ob; org[x] = oa; const uint16_t t = slot[i]; slot[i] = slot[x]; slot[x] = t; }
			}
		}
		__syncthreads();
	}
	uint32_t mine = 0;
	for (uint32_t j = tid; j < c; j += IXB_NT) {
		ck2[s + j] = key[j]; orig2[s + j] = org[j]; occ[s + j] = sy[s + slot[j]];
		if (j == 0 || key[j] != key[j - 1]) ++mine;
	}
	if (mine) atomicAdd(&heads, mine);
	__syncthreads();
	if (tid == 0) nk[b] = heads;
}

// one workgroup per bucket: key ids (kbase[b] + rank of the key inside the bucket), list offsets, the group of every key, the key id of every minimizer
__global__ __launch_bounds__(256)
void k_ixb_groups(const uint32_t *__restrict__ off, const uint32_t *__restrict__ kbase, const uint64_t *__restrict__ ck2, const uint32_t *__restrict__ orig2, int hash_bits,
                  uint32_t n, uint32_t n_keys, uint64_t *__restrict__ key, uint32_t *__restrict__ occ_off, uint32_t *__restrict__ key_grp, uint32_t *__restrict__ grp_of_mz)
{
	__shared__ uint32_t sc[IXB_NT];
	__shared__ uint32_t carry;
	const uint32_t tid = threadIdx.x, b = blockIdx.x, s = off[b], c = off[b + 1] - s;
	if (b == 0 && tid == 0) occ_off[n_keys] = n;                              // the end of the last key's list
	if (c == 0 || c > IXB_CAP) return;
	if (tid == 0) carry = kbase[b];
	__syncthreads();
	for (uint32_t base = 0; base < c; base += IXB_NT) {
		const uint32_t j = base + tid;
		const bool f = j < c && (j == 0 || ck2[s + j] != ck2[s + j - 1]);
		sc[tid] = f ? 1u : 0u;
		__syncthreads();
		for (uint32_t d = 1; d < IXB_NT; d <<= 1) {
			const uint32_t v = tid >= d ? sc[tid - d] : 0;
			__syncthreads();
			sc[tid] += v;
			__syncthreads();
		}
		if (j < c) {
			const uint32_t g = carry + sc[tid] - 1;
			if (f) {
				const uint64_t ck = ck2[s + j];
				key[g] = hash_bits >= 64 ? ck : (ck & ((1ULL << hash_bits) - 1));
				occ_off[g] = s + j;
				key_grp[g] = hash_bits >= 64 ? 0u : (uint32_t)(ck >> hash_bits);
			}
			grp_of_mz[orig2[s + j]] = g;
		}
		__syncthreads();
		if (tid == IXB_NT - 1) carry += sc[IXB_NT - 1];
		__syncthreads();
	}
}

// the bucket table of a batch: group g holds the minimizers [mz_begin[g], mz_begin[g + 1]); returns the number of buckets.  Host code.
inline uint32_t ixb_make_table(int n_grp, const uint64_t *mz_begin /* n_grp + 1 */, int hash_bits, IxbGroup *gt /* n_grp + 1 */)
{
	uint32_t nb = 0;
	for (int g = 0; g < n_grp; ++g) {
		const uint64_t size = mz_begin[g + 1] - mz_begin[g];
		uint32_t bits = 0;
		while (bits < 20 && (int)bits < hash_bits && (size >> bits) > IXB_TARGET) ++bits;
		gt[g].mz_begin = (uint32_t)mz_begin[g]; gt[g].bucket_base = nb; gt[g].bits = bits; gt[g].pad = 0;
		nb += 1u << bits;
	}
	gt[n_grp].mz_begin = (uint32_t)mz_begin[n_grp]; gt[n_grp].bucket_base = nb; gt[n_grp].bits = 0; gt[n_grp].pad = 0;
	return nb;
}

} // namespace pga
