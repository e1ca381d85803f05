// pga_sort_replay.hip -- minimap2's unstable radix_sort_128x (ksort.h:101-151) replayed exactly on many independent
// arrays at once, LEVEL-SYNCHRONOUSLY: the cycle-leader walk of one bucket is sequential (pga_sort_wave.h), but after a
// level has scattered a run into its buckets, the buckets are independent sorts -- the reference merely recurses into
// them one after the other.  Every pass hands each pending run (array slice, digit shift) to its own wave; the wave
// walks the run, insertion-sorts the buckets of <= 64 records and queues the larger ones for the next pass.  The
// critical path of a sort falls from "all levels, one wave" to "the longest run of each level".
#include "pga_common.h"
#include "pga_sort_wave.h"
#include "pga_pipeline.h"

namespace pga {

struct RsRun { uint64_t start; uint32_t len; int32_t shift; };

__device__ __forceinline__ void rs_push(RsRun *out, uint32_t *n_out, uint32_t cap, uint64_t start, uint32_t len, int shift, int lane)
{
	if (lane == 0) { const uint32_t k = atomicAdd(n_out, 1u); if (k < cap) { out[k].start = start; out[k].len = len; out[k].shift = shift; } }
}

// one wave per array: small arrays are finished here, the others enter the run queue at their first non-trivial level
__global__ __launch_bounds__(64)
void k_rs_init(u128 *__restrict__ a, const uint64_t *__restrict__ off, const int64_t *__restrict__ len, int n_seg, const uint32_t *__restrict__ flag, RsRun *__restrict__ out, uint32_t *__restrict__ n_out, uint32_t cap)
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
	rs_push(out, n_out, cap, b, (uint32_t)n, shift, lane);
}

// persistent waves over the run queue of this pass
__global__ __launch_bounds__(64)
void k_rs_pass(u128 *__restrict__ a, const RsRun *__restrict__ in, const uint32_t *__restrict__ n_in, RsRun *__restrict__ out, uint32_t *__restrict__ n_out, uint32_t cap,
               uint32_t *__restrict__ work)
{
	__shared__ RsLds L;
	const int lane = threadIdx.x;
	const uint32_t n_runs = *n_in < cap ? *n_in : cap;
	for (;;) {
		uint32_t r = 0;
		if (lane == 0) r = atomicAdd(work, 1u);
		r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
		if (r >= n_runs) break;
		const RsRun R = in[r];
		u128 *beg = a + R.start;
		const int64_t n = R.len;
		int shift = R.shift;
		while (!rs_level_wave(beg, n, shift, L, lane) && shift > 0) shift -= 8;     // levels that leave the run in one bucket
		if (shift == 0) continue;
		rs_runs_wave(beg, n, shift, L, lane, [&](int64_t rb, int64_t len) { rs_push(out, n_out, cap, R.start + (uint64_t)rb, (uint32_t)len, shift - 8, lane); });
	}
}

// sorts every flagged array [off[s], off[s] + len[s]) of `a` by x exactly as radix_sort_128x would (flag == nullptr: all;
// len == nullptr: the arrays are contiguous, off has n_seg+1 entries)
void replay_sort_segments(u128 *a, uint64_t n_total, const uint64_t *d_off, const int64_t *d_len, int n_seg, const uint32_t *d_flag, hipStream_t st, Timers *tm)
{
	if (n_seg <= 0 || n_total == 0) return;
	EventTimer et(st);
	const uint32_t cap = (uint32_t)std::min<uint64_t>(n_total / 65 + (uint64_t)n_seg + 64, 0x7fffffffu);
	DBuf<RsRun> q0(cap), q1(cap);
	DBuf<uint32_t> ctr(2 * 9 + 2);            // per pass: queue length and work counter
	ctr.zero(st);
	hipLaunchKernelGGL(k_rs_init, dim3((unsigned)n_seg), dim3(64), 0, st, a, d_off, d_len, n_seg, d_flag, q0.p, ctr.p + 0, cap);
	const unsigned grid = 2048;
	RsRun *qin = q0.p, *qout = q1.p;
	for (int pass = 0; pass < 8; ++pass) {      // at most one pass per key byte
		hipLaunchKernelGGL(k_rs_pass, dim3(grid), dim3(64), 0, st, a, qin, ctr.p + 2 * pass, qout, ctr.p + 2 * (pass + 1), cap, ctr.p + 2 * pass + 1);
		std::swap(qin, qout);
	}
	PGA_HIP(hipGetLastError());
	const double ms = et.stop();
	if (tm) { tm->kern[K_SORT].ms += ms; tm->kern[K_SORT].launches += 1; tm->kern[K_SORT].alg_bytes += 32.0 * (double)n_total; }   // every record read and written once (per level, at least one)
	std::vector<uint32_t> h = ctr.download(st);
	for (int pass = 0; pass <= 8; ++pass) if (h[2 * pass] > cap) throw std::runtime_error("pga: run queue overflow in the sort replay");
}

} // namespace pga
