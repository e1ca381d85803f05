// pga_sort_replay.hip -- minimap2's unstable radix_sort_128x (ksort.h:101-151) replayed exactly on many independent
// arrays at once, LEVEL-SYNCHRONOUSLY: the cycle-leader walk of one bucket is sequential (pga_sort_wave.h), but after a
// level has scattered a run into its buckets, the buckets are independent sorts -- the reference merely recurses into
// them one after the other.  Every pass hands each pending run (array slice, digit shift) to its own wave; the wave
// walks the run, insertion-sorts the buckets of <= 64 records and queues the larger ones for the next pass.  The
// critical path of a sort falls from "all levels, one wave" to "the longest run of each level".
#include "pga_common.h"
#include "pga_sort_wave.h"
#include "pga_sort_big.h"
#include "pga_pipeline.h"
#include <cstdio>

namespace pga {

struct RsRun { uint64_t start; uint64_t vary; uint32_t len; int32_t shift; };   // vary: bits that differ somewhere in the whole array (levels without any are skipped)

// highest byte at or below `shift` in which the array's keys differ (-8: none)
__device__ __forceinline__ int rs_next_level(uint64_t vary, int shift)
{
	while (shift >= 0 && ((vary >> shift) & 255) == 0) shift -= 8;
	return shift;
}
// runs of at most RS_SMALL records go to the queue of the LDS-resident sorter
__device__ __forceinline__ void rs_push2(RsRun *out, uint32_t *n_out, RsRun *out_s, uint32_t *n_out_s, uint32_t cap, uint64_t start, uint32_t len, int shift, uint64_t vary, int lane);
__device__ __forceinline__ void rs_push(RsRun *out, uint32_t *n_out, uint32_t cap, uint64_t start, uint32_t len, int shift, uint64_t vary, int lane)
{
	if (lane == 0) { const uint32_t k = atomicAdd(n_out, 1u); if (k < cap) { out[k].start = start; out[k].len = len; out[k].shift = shift; out[k].vary = vary; } }
}

__device__ __forceinline__ void rs_push2(RsRun *out, uint32_t *n_out, RsRun *out_s, uint32_t *n_out_s, uint32_t cap, uint64_t start, uint32_t len, int shift, uint64_t vary, int lane)
{
	if (len <= 1024) rs_push(out_s, n_out_s, cap, start, len, shift, vary, lane); else rs_push(out, n_out, cap, start, len, shift, vary, lane);
}

// one wave per array: small arrays are finished here, the others enter the run queue at their first non-trivial level
__global__ __launch_bounds__(64)
void k_rs_init(u128 *__restrict__ a, const uint64_t *__restrict__ off, const int64_t *__restrict__ len, int n_seg, const uint32_t *__restrict__ flag, RsRun *__restrict__ out, uint32_t *__restrict__ n_out,
               RsRun *__restrict__ out_s, uint32_t *__restrict__ n_out_s, uint32_t cap, RsRun *__restrict__ out_b, uint32_t *__restrict__ n_out_b, uint32_t cap_b, uint32_t big_min)
{
	__shared__ RsLds L;
	const int s = blockIdx.x, lane = threadIdx.x;
	if (s >= n_seg || (flag && !flag[s])) return;
	const uint64_t b = off[s];
	const int64_t n = len ? len[s] : (int64_t)(off[s + 1] - b);
	if (n <= 1) return;
	if (n <= 64) { rs_small_wave(a + b, 0, n, L, lane); return; }
	const uint64_t vary = rs_varying_bits(a + b, n, lane);
	if (vary == 0) return;
	const int shift = (63 - __clzll((long long)vary)) & ~7;
	if (out_b && n >= (int64_t)big_min) rs_push(out_b, n_out_b, cap_b, b, (uint32_t)n, shift, vary, lane);
	else rs_push2(out, n_out, out_s, n_out_s, cap, b, (uint32_t)n, shift, vary, lane);
}

// ---- runs of RSB_MIN records or more: a workgroup per run (pga_sort_big.h) ----
__global__ __launch_bounds__(RSB_NT)
void k_rs_pass_big(u128 *__restrict__ a, const RsRun *__restrict__ in, const uint32_t *__restrict__ n_in, RsRun *__restrict__ out_b, uint32_t *__restrict__ n_out_b, RsRun *__restrict__ out, uint32_t *__restrict__ n_out,
                   RsRun *__restrict__ out_s, uint32_t *__restrict__ n_out_s, uint32_t cap, uint32_t cap_b, uint32_t *__restrict__ work, unsigned long long *__restrict__ prof, uint32_t *__restrict__ rend_all, uint8_t *__restrict__ dig_all,
                   RsHint hint, u128 *__restrict__ tmp_all, uint2 *__restrict__ lg_all, uint4 *__restrict__ blg_all, int run_min, int pass_no, uint32_t big_min)
{
	__shared__ RsBigLds S;
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	if (tid == 0) S.L.run_min = run_min;
	const uint32_t n_runs = *n_in < cap_b ? *n_in : cap_b;
	for (;;) {
		if (tid == 0) S.ctl[0] = atomicAdd(work, 1u);
		__syncthreads();
		const uint32_t r = S.ctl[0];
		__syncthreads();
		if (r >= n_runs) break;
		const RsRun R = in[r];
		if (tid == 0) { S.L.prof[0] = S.L.prof[1] = S.L.prof[2] = S.L.prof[3] = 0; for (int z = 0; z < 8; ++z) S.wp[z] = 0; }
		const unsigned long long tk0 = wall_clock64();
		unsigned long long t_walk = 0;
		uint64_t cur_start = R.start; int64_t n = R.len;
		int shift = R.shift;
		for (;;) {
			u128 *beg = a + cur_start;
			uint8_t *dig = dig_all + cur_start;
			uint32_t *rend = rend_all + cur_start;
			// the first level at or below `shift` that splits the run
			const unsigned long long th0 = wall_clock64();
			while (shift >= 0) {
				rsb_hist(beg, n, shift, dig, S, tid);
				if (S.ctl[1] > 1) break;
				__syncthreads();
				shift = rs_next_level(R.vary, shift - 8);
			}
			if (shift < 0) break;
			const uint32_t n_ne = S.ctl[1];
			unsigned long long nonempty[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) nonempty[k] = __ballot(S.bcnt[lane + 64 * k] > 0);
			const unsigned long long tl0 = wall_clock64();
			bool moved = false;
			if (n_ne == 2 && n >= 1024) {
				// two non-empty buckets: the closed form of pga_sort_wave.h (one wave: three passes with running counts)
				if (wave == 0) {
					int dA = -1; uint32_t cA = 0;
#pragma unroll
					for (int k = 3; k >= 0; --k) if (nonempty[k]) { const int l = __ffsll((long long)nonempty[k]) - 1; dA = l + 64 * k; }
					cA = S.bcnt[dA];
					const bool ok = rs_level_two(beg, n, shift, (uint32_t)dA, cA, tmp_all + cur_start, rend, lane);
					if (lane == 0) S.ctl[2] = ok ? 1u : 0u;
				}
				rs_fence_wg();
				__syncthreads();
				moved = S.ctl[2] != 0;
				__syncthreads();
			}
			if (prof && tid == 0) atomicAdd(&prof[64 + 8 * pass_no + 0], tl0 - th0);
			if (!moved) {
				// digit runs of the original order: long ones -> the run walk (over the resident run table when it fits LDS, else over
				// 64-digit windows and the run ends in memory); short ones -> the walk on digit windows
				rsb_run_count(dig, n, S, tid);
				const uint32_t nb = S.ctl[3];
				const bool by_runs = (uint64_t)n >= (uint64_t)run_min * nb;
				const bool table = by_runs && nb <= RSB_RT_MAX && n < (1LL << 24);
				if (table) rsb_run_table(dig, n, rend, S, tid);
				else if (by_runs) rsb_rend(dig, n, rend, S, tid);
				const unsigned long long tw0 = wall_clock64();
				if (prof && tid == 0) atomicAdd(&prof[64 + 8 * pass_no + 1], tw0 - tl0);
				if (wave == 0) {
#pragma unroll
					for (int k = 0; k < 4; ++k) { const int b = lane + 64 * k; S.L.head[b] = S.bstart[b]; S.L.tail[b] = S.bstart[b] + S.bcnt[b]; }
					rs_fence_wave();
					RsLog G{lg_all + cur_start, blg_all + (cur_start >> 1), 0u, 0u, 0u};
					if (by_runs) rsb_walk_runs(dig, rend, table ? nb : 0u, S.L, S.hw, lane, nonempty, G, S.wp);
					else rsb_walk_digits(dig, S.L, lane, nonempty, n_ne, G);
					if (lane == 0) { S.ctl[4] = G.n1; S.ctl[5] = G.n2; S.ctl[6] = G.tot; }
				}
				rs_fence_wg();
				__syncthreads();
				const unsigned long long ta0 = wall_clock64();
				rsb_apply(beg, tmp_all + cur_start, lg_all + cur_start, blg_all + (cur_start >> 1), S.ctl[4], S.ctl[5], S.ctl[6], tid);
				if (prof && tid == 0) { atomicAdd(&prof[64 + 8 * pass_no + 2], ta0 - tw0); atomicAdd(&prof[64 + 8 * pass_no + 3], wall_clock64() - ta0); }
			}
			t_walk += wall_clock64() - tl0;
			if (shift <= 0) break;                                // nothing below the last byte
			const int next = rs_next_level(R.vary, shift - 8);
			if (next < 0) break;                                  // the keys of a bucket agree in every lower byte: nothing left to order
			// the largest bucket of the level: the workgroup goes on with it when it holds half of the run (see k_rs_pass)
			if (wave == 0) {
				uint32_t big_len = 0, big_off = 0;
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const uint32_t c = S.bcnt[lane + 64 * k], o = S.bstart[lane + 64 * k];
					const uint32_t m = wave_max_u32(c);
					if (m > big_len) { const unsigned long long bm = __ballot(c == m); const int l = __ffsll((long long)bm) - 1; big_len = m; big_off = (uint32_t)__builtin_amdgcn_readlane((int)o, l); }
				}
				if (lane == 0) { S.ctl[7] = big_len; S.ctl[8] = big_off; S.ctl[9] = ((uint64_t)big_len * 2 >= (uint64_t)n && big_len > 4096) ? 1u : 0u; S.ctl[10] = 0u; }
			}
			__syncthreads();
			const uint32_t big_len = S.ctl[7], big_off = S.ctl[8]; const bool follow = S.ctl[9] != 0;
			for (int b = wave; b < 256; b += RSB_NW) {
				const uint32_t c = S.bcnt[b], o = S.bstart[b];
				if (c <= 1) continue;
				const uint64_t g0 = cur_start + o;
				if (c <= 64) { rsb_stable64(a + g0, c, lane); continue; }             // ksort.h:142
				if (hint.dupc && hint.dupc[g0 + c - 1] == hint.dupc[g0]) {
					// no two equal keys in this bucket: its final order is the sorted order
					for (uint32_t i = (uint32_t)lane; i < c; i += 64) { u128 v; v.x = hint.sx[g0 + i]; v.y = hint.sy[g0 + i]; a[g0 + i] = v; }
					continue;
				}
				if (follow && o == big_off && c == big_len) { if (lane == 0) S.ctl[10] = 1u; continue; }
				if (c >= big_min) rs_push(out_b, n_out_b, cap_b, g0, c, next, R.vary, lane);
				else rs_push2(out, n_out, out_s, n_out_s, cap, g0, c, next, R.vary, lane);
			}
			rs_fence_wg();
			__syncthreads();
			const bool go_on = S.ctl[10] != 0;
			__syncthreads();
			if (!go_on) break;
			cur_start += big_off; n = big_len; shift = next;
		}
		if (prof && tid == 0) { for (int z = 0; z < 8; ++z) atomicAdd(&prof[64 + 8 * pass_no + 4 + z], S.wp[z]); }
		if (prof && tid == 0) { const unsigned long long tk2 = wall_clock64(); atomicAdd(&prof[0], t_walk); atomicMax(&prof[1], t_walk); atomicAdd(&prof[32], S.L.prof[0]); atomicAdd(&prof[33], S.L.prof[1]); atomicAdd(&prof[34], S.L.prof[2]); atomicAdd(&prof[35], S.L.prof[3]);
		                         atomicAdd(&prof[2], tk2 - tk0 - t_walk); atomicMax(&prof[3], tk2 - tk0 - t_walk); }
		__syncthreads();
	}
}

// persistent waves over the run queue of this pass
__global__ __launch_bounds__(64)
void k_rs_pass(u128 *__restrict__ a, const RsRun *__restrict__ in, const uint32_t *__restrict__ n_in, RsRun *__restrict__ out, uint32_t *__restrict__ n_out,
               RsRun *__restrict__ out_s, uint32_t *__restrict__ n_out_s, uint32_t cap, uint32_t *__restrict__ work, unsigned long long *__restrict__ prof, uint32_t *__restrict__ rend_all, RsHint hint, u128 *__restrict__ tmp_all, uint2 *__restrict__ lg_all, int run_min)
{
	__shared__ RsLds L;
	const int lane = threadIdx.x;
	if (lane == 0) L.run_min = run_min;
	rs_fence_wave();
	const uint32_t n_runs = *n_in < cap ? *n_in : cap;
	for (;;) {
		uint32_t r = 0;
		if (lane == 0) r = atomicAdd(work, 1u);
		r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
		if (r >= n_runs) break;
		const RsRun R = in[r];
		if (lane == 0) L.prof[0] = L.prof[1] = L.prof[2] = L.prof[3] = 0;
		const unsigned long long tk0 = wall_clock64();
		// A level that leaves one DOMINANT bucket (the strand bit, the target id of a pair: nine records out of ten on one side) does not
		// end the run's turn: the wave goes on with that bucket and queues only the others -- the passes are level-synchronous, and an
		// array whose expensive level sits behind two such levels would otherwise spend it two passes later than its neighbours.
		uint64_t cur_start = R.start; int64_t n = R.len;
		int shift = R.shift;
		unsigned long long t_walk = 0;
		for (;;) {
			u128 *beg = a + cur_start;
			uint32_t cnt[4], off[4];
			uint32_t *rend = rend_all ? rend_all + cur_start : nullptr;    // scratch of the run-length walk, one word per record
			const unsigned long long tl0 = wall_clock64();
			while (shift >= 0 && !rs_level_wave(beg, n, shift, L, lane, cnt, off, rend, tmp_all ? tmp_all + cur_start : nullptr, lg_all ? lg_all + cur_start : nullptr)) shift = rs_next_level(R.vary, shift - 8);   // levels that leave the run in one bucket
			t_walk += wall_clock64() - tl0;
			if (shift <= 0) break;                                // nothing below the last byte
			const int next = rs_next_level(R.vary, shift - 8);
			if (next < 0) break;                                  // the keys of a bucket agree in every lower byte: nothing left to order
			// the largest bucket of the level (uniform)
			uint32_t big_len = 0, big_off = 0;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t m = wave_max_u32(cnt[k]);
				if (m > big_len) { const unsigned long long bm = __ballot(cnt[k] == m); const int l = __ffsll((long long)bm) - 1; big_len = m; big_off = (uint32_t)__builtin_amdgcn_readlane((int)off[k], l); }
			}
			const bool follow = (uint64_t)big_len * 2 >= (uint64_t)n && big_len > 4096;
			bool go_on = false;
			// buckets of this level: <= 64 records are insertion-sorted now (ksort.h:142), larger ones queue for the next level that can split them
			rs_split_buckets(beg, n, shift, cnt, off, L, lane, [&](int64_t rb, int64_t len) {
				const uint64_t g0 = cur_start + (uint64_t)rb;
				if (hint.dupc && hint.dupc[g0 + (uint64_t)len - 1] == hint.dupc[g0]) {
					// no two equal keys in this bucket: its final order is the sorted order
					for (int64_t i = lane; i < len; i += 64) { u128 v; v.x = hint.sx[g0 + (uint64_t)i]; v.y = hint.sy[g0 + (uint64_t)i]; a[g0 + (uint64_t)i] = v; }
					rs_fence_wg();
					return;
				}
				if (follow && (uint32_t)rb == big_off && (uint32_t)len == big_len) { go_on = true; return; }
				rs_push2(out, n_out, out_s, n_out_s, cap, g0, (uint32_t)len, next, R.vary, lane);
			});
			if (!go_on) break;
			rs_fence_wg();
			cur_start += big_off; n = big_len; shift = next;
		}
		if (prof && lane == 0) { const unsigned long long tk2 = wall_clock64(); atomicAdd(&prof[0], t_walk); atomicMax(&prof[1], t_walk); atomicAdd(&prof[32], L.prof[0]); atomicAdd(&prof[33], L.prof[1]); atomicAdd(&prof[34], L.prof[2]); atomicAdd(&prof[35], L.prof[3]);
		                          atomicAdd(&prof[2], tk2 - tk0 - t_walk); atomicMax(&prof[3], tk2 - tk0 - t_walk); }
	}
}

// ---- the same replay DEPENDENCY-DRIVEN: one launch, one queue.  A level-synchronous pass ends when its longest run ends, and an array
// whose expensive level is its second waits behind the first levels of all others; here the buckets of a run enter the queue the
// moment the run is done.  Queue protocol: a producer reserves a slot (atomicAdd on tail), writes the run, then raises the slot's
// ready flag (release: the records it scattered are visible before the flag); `pending` counts runs that are queued or running.
// A consumer claims the next slot (atomicAdd on head) and waits for its flag -- or for pending == 0, which ends the kernel.
struct RsQueue { uint32_t head, tail, pending, overflow; };
__device__ __forceinline__ void rs_apush(RsRun *q, uint32_t *ready, RsQueue *Q, RsRun *out_s, uint32_t *n_out_s, uint32_t cap, uint64_t start, uint32_t len, int shift, uint64_t vary, int lane)
{
	if (len <= 1024) { rs_push(out_s, n_out_s, cap, start, len, shift, vary, lane); return; }
	if (lane == 0) {
		__hip_atomic_fetch_add(&Q->pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint32_t k = __hip_atomic_fetch_add(&Q->tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (k < cap) {
			q[k].start = start; q[k].len = len; q[k].shift = shift; q[k].vary = vary;
			__hip_atomic_store(&ready[k], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		} else { Q->overflow = 1; __hip_atomic_fetch_sub(&Q->pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
	}
}

__global__ __launch_bounds__(64)
void k_rs_async(u128 *__restrict__ a, RsRun *__restrict__ q, uint32_t *__restrict__ ready, RsQueue *__restrict__ Q, RsRun *__restrict__ out_s, uint32_t *__restrict__ n_out_s, uint32_t cap,
                uint32_t *__restrict__ rend_all, RsHint hint, u128 *__restrict__ tmp_all)
{
	__shared__ RsLds L;
	const int lane = threadIdx.x;
	for (;;) {
		uint32_t r = 0;
		if (lane == 0) r = __hip_atomic_fetch_add(&Q->head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
		if (r >= cap) break;
		int state = 0;                                           // 1: the slot is filled, 2: nothing is queued or running any more
		while (state == 0) {
			if (lane == 0) {
				if (__hip_atomic_load(&ready[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) state = 1;
				else if (__hip_atomic_load(&Q->pending, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) state = __hip_atomic_load(&ready[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 2;
			}
			state = __builtin_amdgcn_readfirstlane(state);
			if (state == 0) __builtin_amdgcn_s_sleep(32);
		}
		if (state == 2) break;
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // every lane sees what the producer scattered
		const RsRun R = q[r];
		u128 *beg = a + R.start;
		const int64_t n = R.len;
		int shift = R.shift;
		uint32_t cnt[4], off[4];
		uint32_t *rend = rend_all ? rend_all + R.start : nullptr;
		while (shift >= 0 && !rs_level_wave(beg, n, shift, L, lane, cnt, off, rend, tmp_all ? tmp_all + R.start : nullptr)) shift = rs_next_level(R.vary, shift - 8);
		const int next = shift > 0 ? rs_next_level(R.vary, shift - 8) : -1;
		if (next >= 0) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the scattered records, before any child is announced
			rs_split_buckets(beg, n, shift, cnt, off, L, lane, [&](int64_t rb, int64_t len) {
				const uint64_t g0 = R.start + (uint64_t)rb;
				if (hint.dupc && hint.dupc[g0 + (uint64_t)len - 1] == hint.dupc[g0]) {
					for (int64_t i = lane; i < len; i += 64) { u128 v; v.x = hint.sx[g0 + (uint64_t)i]; v.y = hint.sy[g0 + (uint64_t)i]; a[g0 + (uint64_t)i] = v; }
					rs_fence_wg();
					return;
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // (buckets insertion-sorted in between wrote records too)
				rs_apush(q, ready, Q, out_s, n_out_s, cap, g0, (uint32_t)len, next, R.vary, lane);
			});
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		if (lane == 0) __hip_atomic_fetch_sub(&Q->pending, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
	}
}

// k_rs_init's queue entries become the first slots of the asynchronous queue
__global__ void k_rs_async_seed(const uint32_t *__restrict__ n_in, uint32_t cap, uint32_t *__restrict__ ready, RsQueue *__restrict__ Q)
{
	const uint32_t n = *n_in < cap ? *n_in : cap;
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) ready[i] = 1u;
	if (i == 0) { Q->head = 0; Q->tail = n; Q->pending = n; Q->overflow = *n_in > cap ? 1u : 0u; }
}

// ---- runs of at most RS_SMALL records: the whole remaining sort inside LDS, one wave per run ----
// Same replay (ksort.h:118-149) with the records resident in LDS: a displacement step is two LDS accesses, buckets are
// known from the head/tail arrays (no rescanning), buckets <= 64 are insertion-sorted one lane each, larger ones go on a
// small LDS stack for the next level that can split them.  20 KB of LDS per wave, so eight waves share a CU.
#define RS_SMALL 1024
struct __attribute__((aligned(16))) RsSmallLds {
	u128 buf[RS_SMALL];
	uint32_t head[256], tail[256];
	uint16_t st_start[32], st_len[32]; int8_t st_shift[32];
};

__device__ inline void rs_sort_small(u128 *beg, int n, int shift0, uint64_t vary, RsSmallLds &L, int lane)
{
	for (int i = lane; i < n; i += 64) L.buf[i] = ld128(&beg[i]);
	int top = 0;
	if (lane == 0) { L.st_start[0] = 0; L.st_len[0] = (uint16_t)n; L.st_shift[0] = (int8_t)shift0; }
	top = 1;
	rs_fence_wave();
	while (top > 0) {
		--top;
		const int b0 = L.st_start[top], m = L.st_len[top];
		int shift = L.st_shift[top];
		u128 *B = L.buf + b0;
		uint32_t cnt[4], off[4]; unsigned long long nonempty[4];
		// first level at or below `shift` that splits the run
		for (;;) {
			for (int d = lane; d < 256; d += 64) L.head[d] = 0;
			rs_fence_wave();
			for (int i = lane; i < m; i += 64) atomicAdd(&L.head[(uint32_t)((B[i].x >> shift) & 255)], 1u);
			rs_fence_wave();
			uint32_t run = 0, n_ne = 0;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				cnt[k] = L.head[lane + 64 * k];
				nonempty[k] = __ballot(cnt[k] > 0);
				const uint32_t inc = wave_prefix_sum_incl(cnt[k]);
				off[k] = run + inc - cnt[k];
				run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
				n_ne += (uint32_t)__popcll(nonempty[k]);
			}
			rs_fence_wave();
			if (n_ne > 1) break;
			shift = rs_next_level(vary, shift - 8);
			if (shift < 0) break;
		}
		if (shift < 0) continue;                                 // all keys of the run are equal
#pragma unroll
		for (int k = 0; k < 4; ++k) { L.head[lane + 64 * k] = off[k]; L.tail[lane + 64 * k] = off[k] + cnt[k]; }
		rs_fence_wave();
		// the walk
#pragma unroll 1
		for (int k = 0; k < 4; ++k) {
			unsigned long long todo = nonempty[k];
			while (todo) {
				const int d = 64 * k + (__ffsll((long long)todo) - 1);
				todo &= todo - 1;
				uint32_t h = L.head[d]; const uint32_t tl = L.tail[d];
				while (h < tl) {
					const uint32_t pos = h + (uint32_t)lane;
					const unsigned long long fm = __ballot(pos < tl && (uint32_t)((B[pos < tl ? pos : tl - 1].x >> shift) & 255) != (uint32_t)d);
					if (fm == 0) { h += 64; continue; }
					h += (uint32_t)(__ffsll((long long)fm) - 1);
					u128 carry = B[h];
					int dst = (int)((carry.x >> shift) & 255);
					do {
						const uint32_t hd = L.head[dst];
						const u128 nxt = B[hd];
						rs_fence_wave();
						if (lane == 0) { B[hd] = carry; L.head[dst] = hd + 1; }
						rs_fence_wave();
						carry = nxt;
						dst = (int)((carry.x >> shift) & 255);
					} while (dst != d);
					if (lane == 0) B[h] = carry;
					rs_fence_wave();
					++h;
				}
			}
		}
		if (shift == 0) continue;
		const int next = rs_next_level(vary, shift - 8);
		if (next < 0) continue;
		// buckets: small ones are sorted by their lane now, large ones wait on the stack
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const bool big = cnt[k] > 64;
			if (!big && cnt[k] > 1) rs_insertion(B + off[k], B + off[k] + cnt[k]);
			unsigned long long bm = __ballot(big);
			while (bm) {
				const int src = __ffsll((long long)bm) - 1;
				bm &= bm - 1;
				const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off[k], src), c = (uint32_t)__builtin_amdgcn_readlane((int)cnt[k], src);
				if (lane == 0) { L.st_start[top] = (uint16_t)(b0 + o); L.st_len[top] = (uint16_t)c; L.st_shift[top] = (int8_t)next; }
				++top;
			}
		}
		rs_fence_wave();
	}
	for (int i = lane; i < n; i += 64) beg[i] = L.buf[i];
}

__global__ __launch_bounds__(64)
void k_rs_small(u128 *__restrict__ a, const RsRun *__restrict__ in, const uint32_t *__restrict__ n_in, uint32_t cap, uint32_t *__restrict__ work)
{
	__shared__ RsSmallLds L;
	const int lane = threadIdx.x;
	const uint32_t n_runs = *n_in < cap ? *n_in : cap;
	for (;;) {
		uint32_t r = 0;
		if (lane == 0) r = atomicAdd(work, 1u);
		r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
		if (r >= n_runs) break;
		const RsRun R = in[r];
		rs_sort_small(a + R.start, (int)R.len, R.shift, R.vary, L, lane);
		rs_fence_wg();
	}
}

// sorts every flagged array [off[s], off[s] + len[s]) of `a` by x exactly as radix_sort_128x would (flag == nullptr: all;
// len == nullptr: the arrays are contiguous, off has n_seg+1 entries)
void replay_sort_segments(u128 *a, uint64_t n_total, const uint64_t *d_off, const int64_t *d_len, int n_seg, const uint32_t *d_flag, hipStream_t st, Timers *tm, const RsHint *hint)
{
	if (n_seg <= 0 || n_total == 0) return;
	EventTimer et(st);
	const uint32_t cap = (uint32_t)std::min<uint64_t>(n_total / 65 + (uint64_t)n_seg + 64, 0x7fffffffu);
	DBuf<RsRun> q0(cap), q1(cap), qs(cap);    // two generations of large runs, and the runs small enough for the LDS sorter
	DBuf<uint32_t> ctr(2 * 9 + 2 + 2 * 9);    // per pass: queue length and work counter; [20..38): the same for the queue of BIG runs
	ctr.zero(st);
	// runs of RSB_MIN records or more go to the workgroup kernel (pga_sort_big.h); PGA_RS_NO_BIG=1 keeps everything on single waves
	static const bool use_big = getenv("PGA_RS_NO_BIG") == nullptr && getenv("PGA_NO_RUNWALK") == nullptr && getenv("PGA_NO_TWOBUCKET") == nullptr && getenv("PGA_NO_DIGITWALK") == nullptr && getenv("PGA_RS_ASYNC") == nullptr;
	static const unsigned big_grid = getenv("PGA_RS_BIG_GRID") ? (unsigned)std::max(1, atoi(getenv("PGA_RS_BIG_GRID"))) : 512u;
	static const uint32_t big_min = getenv("PGA_RS_BIG_MIN") ? (uint32_t)std::max(1025, atoi(getenv("PGA_RS_BIG_MIN"))) : (uint32_t)RSB_MIN;
	const uint32_t cap_b = use_big ? (uint32_t)std::min<uint64_t>(n_total / big_min + 2, cap) : 1u;
	DBuf<RsRun> qb0(cap_b), qb1(cap_b);
	hipLaunchKernelGGL(k_rs_init, dim3((unsigned)n_seg), dim3(64), 0, st, a, d_off, d_len, n_seg, d_flag, q0.p, ctr.p + 0, qs.p, ctr.p + 18, cap, use_big ? qb0.p : (RsRun*)nullptr, ctr.p + 20, cap_b, big_min);
	const unsigned grid = 2048;
	RsRun *qin = q0.p, *qout = q1.p;
	const bool verbose = getenv("PGA_VERBOSE") != nullptr;
	DBuf<unsigned long long> dprof(64 + 12 * 8); dprof.zero(st);   // [0, 64): both kernels, slots 4 * pass + {0..3, 32..35}; from 64 on: twelve slots per pass of the workgroup kernel
	DBuf<uint32_t> rend;
	if (!getenv("PGA_NO_RUNWALK")) rend.alloc(n_total);
	DBuf<u128> tmp2;                        // the out-of-place image of the two-bucket levels
	if (rend.p && !getenv("PGA_NO_TWOBUCKET")) tmp2.alloc(n_total);
	DBuf<uint2> lg;                          // the digit walk's record of moves (destination slot, source slot)
	if (tmp2.p && !getenv("PGA_NO_DIGITWALK")) lg.alloc(n_total);
	DBuf<uint8_t> digb; DBuf<uint4> blg;     // workgroup kernel: the digit bytes of a level, the log of moved stretches
	if (use_big) { digb.alloc(n_total + 64); blg.alloc(n_total / 2 + 64); }
	RsRun *qbin = qb0.p, *qbout = qb1.p;
	double pass_ms[9] = {0};
	static const int run_min = getenv("PGA_RS_RUN_MIN") ? std::max(1, atoi(getenv("PGA_RS_RUN_MIN"))) : 64;
	if (verbose) { pass_ms[8] = et.stop(); }
	// (measured: the persistent waves of the dependency-driven variant hold their LDS and slots while they wait and starve the kernels of
	// the other parts -- 7.9 -> 22 s per step; it stays behind PGA_RS_ASYNC=1)
	static const bool sync_passes = getenv("PGA_RS_ASYNC") == nullptr;
	DBuf<uint32_t> ready; DBuf<RsQueue> Qd(1);
	bool async_overflow = false;
	if (!sync_passes) {
		ready.alloc(cap); ready.zero(st);
		EventTimer ep(st);
		hipLaunchKernelGGL(k_rs_async_seed, dim3((cap + 255) / 256), dim3(256), 0, st, ctr.p + 0, cap, ready.p, Qd.p);
		hipLaunchKernelGGL(k_rs_async, dim3(grid), dim3(64), 0, st, a, q0.p, ready.p, Qd.p, qs.p, ctr.p + 18, cap, rend.p, hint ? *hint : RsHint{nullptr, nullptr, nullptr}, tmp2.p);
		if (verbose) pass_ms[0] = ep.stop();
	} else
	for (int pass = 0; pass < 8; ++pass) {      // at most one pass per key byte
		EventTimer ep(st);
		// (a queue that overflowed cap_b is caught below: big runs are at most n_total / RSB_MIN)
		if (use_big) hipLaunchKernelGGL(k_rs_pass_big, dim3(std::min<unsigned>(big_grid, cap_b)), dim3(RSB_NT), 0, st, a, qbin, ctr.p + 20 + 2 * pass, qbout, ctr.p + 20 + 2 * (pass + 1), qout, ctr.p + 2 * (pass + 1), qs.p, ctr.p + 18, cap, cap_b, ctr.p + 20 + 2 * pass + 1,
		                                verbose ? dprof.p + 4 * pass : (unsigned long long*)nullptr, rend.p, digb.p, hint ? *hint : RsHint{nullptr, nullptr, nullptr}, tmp2.p, lg.p, blg.p, run_min, pass, big_min);
		hipLaunchKernelGGL(k_rs_pass, dim3(grid), dim3(64), 0, st, a, qin, ctr.p + 2 * pass, qout, ctr.p + 2 * (pass + 1), qs.p, ctr.p + 18, cap, ctr.p + 2 * pass + 1, verbose ? dprof.p + 4 * pass : (unsigned long long*)nullptr, rend.p, hint ? *hint : RsHint{nullptr, nullptr, nullptr}, tmp2.p, lg.p, run_min);
		if (verbose) {
			pass_ms[pass] = ep.stop();
			uint32_t nr = 0; PGA_HIP(hipMemcpy(&nr, ctr.p + 2 * (pass + 1), 4, hipMemcpyDeviceToHost));
			if (nr > 0 && nr <= cap) {
				std::vector<RsRun> hr(nr); PGA_HIP(hipMemcpy(hr.data(), qout, (size_t)nr * sizeof(RsRun), hipMemcpyDeviceToHost));
				uint64_t sum = 0; uint32_t mx = 0; int sh_min = 64, sh_max = -8;
				for (auto &r : hr) { sum += r.len; mx = std::max(mx, r.len); sh_min = std::min(sh_min, r.shift); sh_max = std::max(sh_max, r.shift); }
				fprintf(stderr, "[pga]     after pass %d: %u runs queued, %llu records, longest %u, shifts %d..%d\n", pass, nr, (unsigned long long)sum, mx, sh_min, sh_max);
			}
			{
				std::vector<unsigned long long> pr = dprof.download(st);
				fprintf(stderr, "[pga]     pass %d: %.1f ms; walks: sum %.1f ms, longest %.2f ms; bucket splitting: sum %.1f ms, longest %.2f ms\n", pass, pass_ms[pass], pr[4 * pass] * 1e-5, pr[4 * pass + 1] * 1e-5, pr[4 * pass + 2] * 1e-5, pr[4 * pass + 3] * 1e-5);
				fprintf(stderr, "[pga]       run-length walk: %llu bulk events moving %llu cycles, %llu token cycles of %llu steps\n", pr[4 * pass + 32], pr[4 * pass + 33], pr[4 * pass + 34], pr[4 * pass + 35]);
				const unsigned long long *pb = pr.data() + 64 + 12 * pass;
				if (use_big) fprintf(stderr, "[pga]       workgroup kernel, summed over its runs: histogram %.2f ms, run ends %.2f ms, walk %.2f ms, apply %.2f ms; in the walk: %llu home skips %.2f ms, lean cycles %.2f ms, rotations %.2f ms (%llu stops; following %.2f ms, logging %.2f ms), token walks %.2f ms\n", pb[0] * 1e-5, pb[1] * 1e-5, pb[2] * 1e-5, pb[3] * 1e-5,
				                     pb[8], pb[4] * 1e-5, pb[5] * 1e-5, pb[6] * 1e-5, pb[9], pb[10] * 1e-5, pb[11] * 1e-5, pb[7] * 1e-5);
			}
		}
		std::swap(qin, qout);
		std::swap(qbin, qbout);
	}
	// the small runs were final the moment they were queued: one launch sorts them all
	EventTimer es(st);
	hipLaunchKernelGGL(k_rs_small, dim3(8192), dim3(64), 0, st, a, qs.p, ctr.p + 18, cap, ctr.p + 19);
	const double ms_small = verbose ? es.stop() : 0.0;
	PGA_HIP(hipGetLastError());
	const double ms = et.stop(K_SORT);
	if (tm) { tm->kern[K_SORT].ms += ms; tm->kern[K_SORT].launches += 1; tm->kern[K_SORT].alg_bytes += 32.0 * (double)n_total; }   // every record read and written once (per level, at least one)
	std::vector<uint32_t> h = ctr.download(st);
	if (!sync_passes) async_overflow = Qd.download(st)[0].overflow != 0;
	if (getenv("PGA_VERBOSE")) fprintf(stderr, "[pga]   sort replay: %llu records in %d arrays, %.3f ms; runs per pass: %u %u %u %u %u %u %u %u; ms: init %.1f, passes %.1f %.1f %.1f %.1f %.1f; %u small runs %.1f ms\n", (unsigned long long)n_total, n_seg, ms, h[0], h[2], h[4], h[6], h[8], h[10], h[12], h[14], pass_ms[8], pass_ms[0], pass_ms[1], pass_ms[2], pass_ms[3], pass_ms[4], h[18], ms_small);
	for (int pass = 0; pass <= 8; ++pass) if (h[2 * pass] > cap) throw std::runtime_error("pga: run queue overflow in the sort replay");
	for (int pass = 0; pass <= 8; ++pass) if (use_big && h[20 + 2 * pass] > cap_b) throw std::runtime_error("pga: big-run queue overflow in the sort replay");
	if (h[18] > cap || async_overflow) throw std::runtime_error("pga: run queue overflow in the sort replay");
}

} // namespace pga
