// pga_sort_big.h -- one level of minimap2's unstable radix sort (ksort.h:118-146) on a BIG run, by a whole workgroup.
//
// pga_sort_wave.h gives a run to ONE wave: that wave streams the run for the histogram, streams it again for the digit-run ends, walks,
// moves the records itself and sorts the small buckets -- the walk is sequential by definition, the rest is not.  Here a run of
// RSB_MIN records or more belongs to a workgroup of RSB_NT threads:
//   streaming phases (all waves)   digit bytes + per-wave histograms (one LDS add per 64 records of a digit run), digit-run ends per
//                                  stripe with a fix-up across stripe borders
//   the walk (wave 0 only)         on the read-only DIGIT BYTES and run ends: where the token goes depends only on the ORIGINAL digit
//                                  sequence (slots at or beyond a head still hold their original records), and every slot is filled
//                                  exactly once from a slot that still held its original record -- so the walk moves nothing; it writes
//                                  down (destination, source, length) for every stretch of slots it fills: `blg` for stretches, `lg` for
//                                  single slots.  A head that advances inside a digit run needs no memory access at all (the cached
//                                  digit stays, the cached remainder drops by one).
//   apply (all waves)              final[dst + m] = original[src + m]: gathered into `tmp` in log order, barrier, scattered back
//   buckets (all waves)            buckets without equal keys are copied from the stable sort (RsHint), buckets of <= 64 records are
//                                  stably rank-sorted by a wave (ksort.h:142: insertion sort is a stable sort), the others are queued
// The arrangement is the reference's: the walk is the one of pga_sort_wave.h (rs_walk_runs / rs_level_sim) with "move" replaced by "log".
#pragma once
#include "pga_sort_wave.h"

namespace pga {

#define RSB_NT 512
#define RSB_NW (RSB_NT / 64)
#define RSB_MIN 16384         // runs from this size on go to the workgroup kernel

// LDS traffic of ONE wave executes in order; what the walk needs between a lane-0 write and the next read is that the compiler keeps the
// order and that earlier LDS operations have returned -- not the completion of the global stores of the log, which a wavefront-scope
// fence (s_waitcnt vmcnt(0)) would wait for on every step
__device__ __forceinline__ void rsb_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct __attribute__((aligned(16))) RsBigLds {
	RsLds L;                         // wave 0's walk state
	uint32_t whist[RSB_NW][256];     // per-wave digit histograms
	uint32_t bcnt[256], bstart[257]; // the level's bucket sizes and offsets
	uint32_t ctl[16];                // broadcast words
	uint32_t s_end[RSB_NW + 1];      // rend: true end of the digit run that crosses the end of stripe w
	uint32_t s_nb[RSB_NW];
	unsigned long long wp[8];        // walk profile (ticks): home skips, lean cycles, followed cycles (simple), token walks
	uint2 hw[256];                   // what sits at a bucket's head: x = the position the entry describes (valid while it equals the head), y = remainder of its digit run << 8 | digit
};
#define RSB_TRY_MIN 8u            // digit-run remainder at a cycle's leader from which the bulk forms (rotations, loops) are tried at all

// ---- digit bytes and histogram of [beg, beg + n) at `shift`: S.bcnt / S.bstart / S.ctl[1] = non-empty buckets ----
__device__ inline void rsb_hist(const u128 *beg, int64_t n, int shift, uint8_t *dig, RsBigLds &S, int tid)
{
	const int wave = tid >> 6, lane = tid & 63;
	for (int d = tid; d < RSB_NW * 256; d += RSB_NT) (&S.whist[0][0])[d] = 0;
	__syncthreads();
	uint32_t *wh = S.whist[wave];
	for (int64_t i0 = 0; i0 < n; i0 += RSB_NT * 4) {
		uint32_t dg[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) { const int64_t i = i0 + (int64_t)k * RSB_NT + tid; const uint32_t d = (uint32_t)((beg[i < n ? i : n - 1].x >> shift) & 255); dg[k] = i < n ? d : 256u; }
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int64_t i = i0 + (int64_t)k * RSB_NT + tid;
			if (i < n) dig[i] = (uint8_t)dg[k];
			const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dg[k]);
			if (__ballot(dg[k] == d0) == ~0ULL) { if (lane == 0 && d0 < 256u) wh[d0] += 64u; }   // 64 records of one digit run: one add
			else if (dg[k] < 256u) atomicAdd(&wh[dg[k]], 1u);
		}
	}
	__syncthreads();
	if (wave == 0) {
		uint32_t run = 0, n_ne = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			uint32_t c = 0;
#pragma unroll
			for (int w = 0; w < RSB_NW; ++w) c += S.whist[w][lane + 64 * k];
			const uint32_t inc = wave_prefix_sum_incl(c);
			S.bcnt[lane + 64 * k] = c;
			S.bstart[lane + 64 * k] = run + inc - c;
			run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
			n_ne += (uint32_t)__popcll(__ballot(c > 0));
		}
		if (lane == 0) { S.bstart[256] = run; S.ctl[1] = n_ne; }
	}
	__syncthreads();
}

// ---- rend[p] = first position after p whose digit differs (digit runs of the original order); S.ctl[3] = number of runs ----
__device__ inline void rsb_rend(const uint8_t *dig, int64_t n, uint32_t *rend, RsBigLds &S, int tid)
{
	const int wave = tid >> 6, lane = tid & 63;
	// stripes of whole 256-record chunks
	const int64_t chunks = (n + 255) >> 8, per = (chunks + RSB_NW - 1) / RSB_NW;
	const int64_t s_lo = (int64_t)wave * per << 8, s_hi_raw = (int64_t)(wave + 1) * per << 8, s_hi = s_hi_raw < n ? s_hi_raw : n;
	uint32_t nb = 0;
	if (s_lo < n) {
		uint32_t carry_end = (uint32_t)s_hi, next_first = s_hi < n ? (uint32_t)dig[s_hi] : 257u;
		for (int64_t g0 = (s_hi - 1) & ~255LL; g0 >= s_lo; g0 -= 256) {
			uint32_t dg[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) { const int64_t p = g0 + 64 * k + lane; const uint32_t d = (uint32_t)dig[p < s_hi ? p : s_hi - 1]; dg[k] = p < s_hi ? d : 256u; }
#pragma unroll
			for (int k = 3; k >= 0; --k) {
				const int64_t p = g0 + 64 * k + lane;
				const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp((int)next_first, (int)dg[k], 0x130, 0xf, 0xf, false);   // wave_shl:1, lane 63 sees the record above
				const bool last_of_run = p < s_hi && nx != dg[k];
				const unsigned long long bm = __ballot(last_of_run);
				nb += (uint32_t)__popcll(bm);
				const unsigned long long up = bm >> lane;
				const uint32_t e = up ? (uint32_t)(p + (__ffsll((long long)up) - 1) + 1) : carry_end;
				if (p < s_hi) rend[p] = e;
				carry_end = (uint32_t)__builtin_amdgcn_readlane((int)e, 0);
				next_first = (uint32_t)__builtin_amdgcn_readlane((int)dg[k], 0);
			}
		}
	}
	if (lane == 0) S.s_nb[wave] = nb;
	rs_fence_wg();
	__syncthreads();
	// a run that crosses a stripe border ends where the next stripe says (back to front: a run may span several stripes)
	if (tid == 0) {
		uint32_t tot = 0;
		for (int w = 0; w < RSB_NW; ++w) tot += S.s_nb[w];
		S.ctl[3] = tot;
		S.s_end[RSB_NW] = (uint32_t)n;
		for (int w = RSB_NW - 1; w >= 0; --w) {
			const int64_t hi_raw = (int64_t)(w + 1) * per << 8;
			if (hi_raw >= n) { S.s_end[w] = (uint32_t)n; continue; }            // the last stripe (or beyond the run): ends at n
			// stripe w ends at hi_raw < n; the run crossing that border (if dig agrees) ends at rend[hi_raw] as the next stripe knows it
			if (dig[hi_raw - 1] != dig[hi_raw]) { S.s_end[w] = (uint32_t)hi_raw; continue; }
			uint32_t e = rend[hi_raw];
			const int64_t nhi_raw = (int64_t)(w + 2) * per << 8, nhi = nhi_raw < n ? nhi_raw : n;
			if ((int64_t)e == nhi && nhi < n) e = S.s_end[w + 1];                 // that run reaches the next border too: already resolved
			S.s_end[w] = e;
		}
	}
	__syncthreads();
	if (s_lo < n && s_hi < n) {
		const uint32_t e = S.s_end[wave];
		if ((int64_t)e != s_hi) {
			// the trailing run of the stripe: positions from s_hi - 1 backwards while the digit stays
			const uint32_t d = (uint32_t)dig[s_hi - 1];
			for (int64_t p0 = s_hi - 64; ; p0 -= 64) {
				const int64_t p = p0 + lane;
				const bool same = p >= s_lo && (uint32_t)dig[p >= s_lo ? p : s_lo] == d;
				// contiguous from the top: lanes above the highest mismatch
				const unsigned long long mis = __ballot(!same);
				const int top_mis = mis ? 63 - __clzll((long long)mis) : -1;
				if (lane > top_mis && p >= s_lo) rend[p] = e;
				if (mis || p0 <= s_lo) break;
			}
		}
	}
	rs_fence_wg();
	__syncthreads();
}

// ---- digit runs of the original order: their number (S.ctl[3], per stripe S.s_nb) ----
__device__ inline void rsb_run_count(const uint8_t *dig, int64_t n, RsBigLds &S, int tid)
{
	const int wave = tid >> 6, lane = tid & 63;
	const int64_t chunks = (n + 255) >> 8, per = (chunks + RSB_NW - 1) / RSB_NW;
	const int64_t s_lo = (int64_t)wave * per << 8, s_hi_raw = (int64_t)(wave + 1) * per << 8, s_hi = s_hi_raw < n ? s_hi_raw : n;
	uint32_t nb = 0;
	if (s_lo < n) {
		uint32_t d_prev = s_lo > 0 ? (uint32_t)dig[s_lo - 1] : 257u;
		for (int64_t g0 = s_lo; g0 < s_hi; g0 += 256) {
			uint32_t dg[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) { const int64_t p = g0 + 64 * k + lane; const uint32_t d = (uint32_t)dig[p < s_hi ? p : s_hi - 1]; dg[k] = p < s_hi ? d : 256u; }
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t left = (uint32_t)wave_shr1((int)dg[k], (int)d_prev);
				nb += (uint32_t)__popcll(__ballot(dg[k] < 256u && dg[k] != left));
				d_prev = (uint32_t)__builtin_amdgcn_readlane((int)dg[k], 63);
			}
		}
	}
	if (lane == 0) S.s_nb[wave] = nb;
	__syncthreads();
	if (tid == 0) { uint32_t tot = 0; for (int w = 0; w < RSB_NW; ++w) tot += S.s_nb[w]; S.ctl[3] = tot; }
	__syncthreads();
}

// ---- the run table: rt[r] = start << 8 | digit of the r-th digit run, rt[nb] = n << 8 (n < 2^24); built in order by all waves ----
__device__ inline void rsb_run_table(const uint8_t *dig, int64_t n, uint32_t *rt, RsBigLds &S, int tid)
{
	const int wave = tid >> 6, lane = tid & 63;
	const int64_t chunks = (n + 255) >> 8, per = (chunks + RSB_NW - 1) / RSB_NW;
	const int64_t s_lo = (int64_t)wave * per << 8, s_hi_raw = (int64_t)(wave + 1) * per << 8, s_hi = s_hi_raw < n ? s_hi_raw : n;
	uint32_t base = 0;
	for (int w = 0; w < wave; ++w) base += S.s_nb[w];
	const unsigned long long lt = (1ULL << lane) - 1;
	if (s_lo < n) {
		uint32_t d_prev = s_lo > 0 ? (uint32_t)dig[s_lo - 1] : 257u;
		for (int64_t g0 = s_lo; g0 < s_hi; g0 += 256) {
			uint32_t dg[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) { const int64_t p = g0 + 64 * k + lane; const uint32_t d = (uint32_t)dig[p < s_hi ? p : s_hi - 1]; dg[k] = p < s_hi ? d : 256u; }
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int64_t p = g0 + 64 * k + lane;
				const uint32_t left = (uint32_t)wave_shr1((int)dg[k], (int)d_prev);
				const bool st = dg[k] < 256u && dg[k] != left;
				const unsigned long long bm = __ballot(st);
				if (st) rt[base + (uint32_t)__popcll(bm & lt)] = (uint32_t)p << 8 | dg[k];
				base += (uint32_t)__popcll(bm);
				d_prev = (uint32_t)__builtin_amdgcn_readlane((int)dg[k], 63);
			}
		}
	}
	if (tid == 0) rt[S.ctl[3]] = (uint32_t)n << 8;
	rs_fence_wg();
	__syncthreads();
	// into LDS (the window pool of the walking wave holds it: 8192 entries)
	uint32_t *RT = (uint32_t*)S.L.win;
	const uint32_t m = S.ctl[3] + 1;
	for (uint32_t r = (uint32_t)tid; r < m; r += RSB_NT) RT[r] = rt[r];
	__syncthreads();
}
#define RSB_RT_MAX 8191u

// ---- the log of the walk ----
struct RsLog {
	uint2 *lg; uint4 *blg;          // single slots (dst, src); stretches (dst, src, len, offset of the stretch in the gather area)
	uint32_t n1, n2, tot;           // entries of lg, of blg, records covered by blg
};
__device__ __forceinline__ void rsl_one(RsLog &G, uint32_t dst, uint32_t src, int lane) { if (lane == 0) G.lg[G.n1] = make_uint2(dst, src); ++G.n1; }
// one stretch (uniform arguments); stretches are cut into pieces of at most 4096 records so that the apply pass can deal them out
__device__ __forceinline__ void rsl_run(RsLog &G, uint32_t dst, uint32_t src, uint32_t len, int lane)
{
	if (len == 0) return;
	if (len < 4) { for (uint32_t m = 0; m < len; ++m) rsl_one(G, dst + m, src + m, lane); return; }
	const uint32_t np = (len + 4095u) >> 12;
	for (uint32_t p0 = 0; p0 < np; p0 += 64) {
		const uint32_t p = p0 + (uint32_t)lane;
		if (p < np) { const uint32_t o = p << 12, l = len - o < 4096u ? len - o : 4096u; G.blg[G.n2 + p] = make_uint4(dst + o, src + o, l, G.tot + o); }
	}
	G.n2 += np; G.tot += len;
}

// ---- rs_walk_runs (pga_sort_wave.h) on digit bytes, logging instead of moving.  Wave 0 only; L.head / L.tail are set. ----
// HW[k] describes the record at bucket k's head (position, digit, remainder of its digit run): one LDS read tells where the token goes
// next; an entry is valid while its position equals the head.  Two forms of a cycle:
//   * the leader's digit run is long (>= RSB_TRY_MIN): the cycle is followed first without logging anything; if it is simple (distinct
//     buckets, no home record met) the next M cycles are M rotations -- logged as stretches; otherwise the token walk with its loop and
//     home-run shortcuts, as in pga_sort_wave.h;
//   * otherwise LEAN token steps: per stop one LDS round trip (head + entry, read together), one log entry, the entry advanced in place
//     while the head stays inside its digit run -- memory is touched only when a digit run at a head is used up.
__device__ inline void rsb_walk_runs(const uint8_t *dig, const uint32_t *rend, uint32_t n_rt, RsLds &L, uint2 *HW, int lane, const unsigned long long (&nonempty)[4], RsLog &G, unsigned long long *wp)
{
	const uint2 none = make_uint2(RS_NONE, 0u);
	// The entry of bucket k for head position pos (uniform arguments; pos < tail), from the bucket's 64-digit LDS window: the digit is the
	// window's, the remainder of its run the distance to the first other digit (one ballot).  Memory is read when the head has left the
	// window (64 digits, one coalesced load) and when the run reaches the window's end (its true end: rend[pos]).
	uint8_t *W = (uint8_t*)L.win;
	const uint32_t *RT = (const uint32_t*)L.win;      // table mode (n_rt > 0): every digit run of the level, resident; L.wbase[k] = the run at bucket k's head
	auto load_entry = [&](uint32_t k, uint32_t pos, uint32_t tk) -> uint2 {
		if (n_rt) {
			uint32_t r = L.wbase[k];
			while ((RT[r + 1] >> 8) <= pos) ++r;                          // heads only move forward: the run index follows
			const uint32_t e = RT[r + 1] >> 8;
			uint32_t rem = (e < tk ? e : tk) - pos;
			if (rem > 0xffffffu) rem = 0xffffffu;
			const uint2 w = make_uint2(pos, rem << 8 | (RT[r] & 255u));
			if (lane == 0) { L.wbase[k] = r; HW[k] = w; }
			return w;
		}
		uint32_t wb = L.wbase[k];
		if (wb == RS_NONE || pos - wb >= 64u) {
			const uint32_t p = pos + (uint32_t)lane;
			W[k * 64u + (uint32_t)lane] = (uint8_t)(p < tk ? dig[p] : 0);
			if (lane == 0) L.wbase[k] = pos;
			wb = pos;
			rsb_sync_lds();
		}
		const uint32_t off = pos - wb, valid = (64u - off) < (tk - pos) ? (64u - off) : (tk - pos);
		const uint32_t b = (uint32_t)lane < valid ? (uint32_t)W[k * 64u + off + (uint32_t)lane] : 256u;
		const uint32_t dd = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
		const unsigned long long mism = __ballot(b != dd);
		uint32_t rem = mism ? (uint32_t)(__ffsll((long long)mism) - 1) : 64u;
		if (rem == valid && pos + rem < tk) { uint32_t r2 = rend[pos]; if (r2 > tk) r2 = tk; rem = r2 - pos; }     // the run goes on beyond the window
		if (rem > 0xffffffu) rem = 0xffffffu;                                 // (a lower bound is enough: the rest of the run is met again)
		const uint2 w = make_uint2(pos, rem << 8 | dd);
		if (lane == 0) HW[k] = w;
		return w;
	};
	auto peek = [&](uint32_t k, uint32_t pos, uint32_t tk, uint32_t &dd, uint32_t &rem) {
		uint2 w = HW[k];
		if (w.x != pos) { w = load_entry(k, pos, tk); rsb_sync_lds(); }
		dd = w.y & 255u; rem = w.y >> 8;
	};
	// the head of bucket k moved from pos by m records (lane-private call: every lane its own bucket, or uniform with lane 0 writing);
	// an entry whose digit run is used up is dropped and rebuilt when the bucket is looked at again
	auto moved_entry = [&](uint32_t k, uint32_t pos, uint32_t m) -> bool {
		const uint2 w = HW[k];
		if (w.x == pos && (w.y >> 8) > m) { HW[k] = make_uint2(pos + m, w.y - (m << 8)); return false; }
		HW[k] = none; return true;
	};
	// entries of the stops pk[q0..q1) whose digit runs were used up, rebuilt by one lane each from the resident run table (window mode: on demand)
	auto refresh = [&](int q0, int q1) {
		if (!n_rt) return;
		for (int q = q0 + lane; q < q1; q += 64) {
			const uint32_t k = L.pk[q], pos = L.head[k], tk = L.tail[k];
			if (pos >= tk) { HW[k] = none; continue; }
			if (HW[k].x == pos) continue;
			uint32_t r = L.wbase[k];
			while ((RT[r + 1] >> 8) <= pos) ++r;
			const uint32_t e = RT[r + 1] >> 8;
			uint32_t rem = (e < tk ? e : tk) - pos;
			if (rem > 0xffffffu) rem = 0xffffffu;
			HW[k] = make_uint2(pos, rem << 8 | (RT[r] & 255u));
			L.wbase[k] = r;
		}
		rsb_sync_lds();
	};
	// The y words of all 256 entries, four per lane, in registers: following a cycle is then a select + v_readlane per stop instead of a
	// dependent LDS round trip (0 = no valid entry: the stop falls back to load_entry).  Reloaded before every path that is followed.
	uint32_t ey0 = 0, ey1 = 0, ey2 = 0, ey3 = 0;
	auto reload_regs = [&]() {
		const uint2 a0 = HW[lane], a1 = HW[lane + 64], a2 = HW[lane + 128], a3 = HW[lane + 192];
		ey0 = a0.x == L.head[lane] ? a0.y : 0u; ey1 = a1.x == L.head[lane + 64] ? a1.y : 0u;
		ey2 = a2.x == L.head[lane + 128] ? a2.y : 0u; ey3 = a3.x == L.head[lane + 192] ? a3.y : 0u;
	};
	auto entry_y = [&](uint32_t k) -> uint32_t {
		const uint32_t sel = k >> 6;
		const uint32_t v = sel == 0 ? ey0 : sel == 1 ? ey1 : sel == 2 ? ey2 : ey3;
		return (uint32_t)rl32((int)v, (int)(k & 63u));
	};
	// positions of the stops pk[0..n) (their heads), lane-parallel
	auto fill_ppos = [&](int nq) { for (int q = lane; q < nq; q += 64) L.ppos[q] = L.head[L.pk[q]]; rsb_sync_lds(); };
	for (int k = lane; k < 256; k += 64) {
		HW[k] = none; L.vmark[k] = 0;
		uint32_t wb = RS_NONE;
		if (n_rt) {                                                      // the run that holds the bucket's first slot
			const uint32_t pos = L.head[k];
			uint32_t lo = 0, hi = n_rt;                                   // last r with start(r) <= pos
			while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((RT[mid] >> 8) <= pos) lo = mid; else hi = mid; }
			wb = lo;
		}
		L.wbase[k] = wb;
	}
	rsb_sync_lds();
#pragma unroll 1
	for (int kk = 0; kk < 4; ++kk) {
		unsigned long long todo = nonempty[kk];
		while (todo) {
			const uint32_t i = (uint32_t)(64 * kk + (__ffsll((long long)todo) - 1));
			todo &= todo - 1;
			uint32_t h = L.head[i]; const uint32_t tl = L.tail[i];
			while (h < tl) {
				uint32_t d0, rem0;
				peek(i, h, tl, d0, rem0);
				const unsigned long long tq0 = wall_clock64();
				if (d0 == i) { h += rem0; if (lane == 0) { L.head[i] = h; wp[0] += wall_clock64() - tq0; wp[4] += 1; } rsb_sync_lds(); continue; }     // a run of records that are home already
				if (rem0 < RSB_TRY_MIN) {
					// ---- lean token cycle (ksort.h:131-138 on indices) ----
					uint32_t carry = h, k = d0, n_steps = 0;
					do {
						const uint32_t pos = L.head[k], tk = L.tail[k];
						uint2 w = HW[k];
						if (w.x != pos) { w = load_entry(k, pos, tk); }
						const uint32_t dd = w.y & 255u, rem = w.y >> 8;
						if (dd == k && rem >= 8 && pos + rem < tk) {
							// the head holds a long run of home records: the token pushes them up by one slot each -- in bulk
							rsl_one(G, pos, carry, lane);
							rsl_run(G, pos + 1, pos, rem, lane);
							carry = pos + rem;
							if (lane == 0) L.head[k] = pos + rem + 1;
							const uint2 wc = load_entry(k, carry, tk);           // the record behind the home run (its entry is stale at once: the head is past it)
							k = wc.y & 255u;
						} else {
							if (lane == 0) {
								G.lg[G.n1] = make_uint2(pos, carry);
								L.head[k] = pos + 1;
								HW[k] = rem > 1 ? make_uint2(pos + 1, w.y - 256u) : none;
							}
							++G.n1;
							carry = pos; k = dd;
						}
						asm volatile("" ::: "memory");                  // (LDS operations of one wave execute in order: the next read sees these writes)
						++n_steps;
					} while (k != i);
					rsl_one(G, h, carry, lane);
					if (lane == 0) { L.prof[2] += 1; L.prof[3] += n_steps; HW[i] = rem0 > 1 ? make_uint2(h + 1, (rem0 - 1) << 8 | d0) : none; L.head[i] = h + 1; wp[1] += wall_clock64() - tq0; }
					++h;
					rsb_sync_lds();
					continue;
				}
				// follow the cycle that starts with the record at h without moving anything
				// (one LDS round trip per stop: head and entry are read together, the visited buckets are a bit set in registers, the
				// stop is written down without waiting)
				uint32_t M = rem0, k = d0; int Lc = 0; bool simple = true;
				unsigned long long va = 0, vb = 0, vc = 0, vd = 0;            // (four scalars, not an indexed array: no scratch)
				rsb_sync_lds();
				reload_regs();
				for (;;) {
					const unsigned long long bit = 1ULL << (k & 63u);
					const uint32_t wi = k >> 6;
					const unsigned long long vw = wi == 0 ? va : wi == 1 ? vb : wi == 2 ? vc : vd;
					if ((vw & bit) || Lc == 256) { simple = false; break; }
					uint32_t y = entry_y(k);
					if (y == 0) { y = load_entry(k, L.head[k], L.tail[k]).y; rsb_sync_lds(); reload_regs(); }
					const uint32_t dd = y & 255u, r2 = y >> 8;
					if (dd == k) { simple = false; break; }
					if (lane == 0) L.pk[Lc] = k;
					if (wi == 0) va |= bit; else if (wi == 1) vb |= bit; else if (wi == 2) vc |= bit; else vd |= bit;
					++Lc;
					if (r2 < M) M = r2;
					if (dd == i) break;
					k = dd;
				}
				rsb_sync_lds();
				if (simple) {
					const unsigned long long tqa = wall_clock64();
					if (lane == 0) { L.prof[0] += 1; L.prof[1] += M; wp[6] += tqa - tq0; }
					// M rotations: the leader's records go to the first stop, every stop's records to the next stop, the last stop's into the
					// leader's slots -- one log entry per stop, written by one lane each
					fill_ppos(Lc);
					if (M >= 4 && M <= 4096) {
						for (int q = lane; q <= Lc; q += 64) {
							const uint32_t dstp = q < Lc ? L.ppos[q] : h, srcp = q == 0 ? h : L.ppos[q - 1];
							G.blg[G.n2 + (uint32_t)q] = make_uint4(dstp, srcp, M, G.tot + (uint32_t)q * M);
						}
						G.n2 += (uint32_t)Lc + 1u; G.tot += ((uint32_t)Lc + 1u) * M;
					} else if (M < 4) {
						for (int q = lane; q <= Lc; q += 64) {
							const uint32_t dstp = q < Lc ? L.ppos[q] : h, srcp = q == 0 ? h : L.ppos[q - 1];
							for (uint32_t m = 0; m < M; ++m) G.lg[G.n1 + (uint32_t)q * M + m] = make_uint2(dstp + m, srcp + m);
						}
						G.n1 += ((uint32_t)Lc + 1u) * M;
					} else {
						rsl_run(G, L.ppos[0], h, M, lane);
						for (int q = 1; q < Lc; ++q) rsl_run(G, L.ppos[q], L.ppos[q - 1], M, lane);
						rsl_run(G, h, L.ppos[Lc - 1], M, lane);
					}
					if (lane == 0) { wp[7] += wall_clock64() - tqa; wp[5] += (unsigned)Lc; }
					// heads move inside their digit runs where the runs are longer than M: the entries follow without a load
					bool stale = false;
					for (int q = lane; q < Lc; q += 64) { const uint32_t kq = L.pk[q]; L.head[kq] = L.ppos[q] + M; stale |= moved_entry(kq, L.ppos[q], M); }
					rsb_sync_lds();
					if (__ballot(stale)) refresh(0, Lc);
					if (lane == 0) { L.head[i] = h + M; (void)moved_entry(i, h, M); wp[2] += wall_clock64() - tq0; }
					h += M;
					rsb_sync_lds();
					continue;
				}
				// the token walk of this one cycle, on indices: the carried record is known by its source slot
				if (lane == 0) L.prof[2] += 1;
				uint32_t carry = h;
				uint32_t dst = d0;
				while (dst != i) {
					int Lc2 = 0, q0 = -1; uint32_t k2 = dst, T = 0xffffffffu; bool home = false;
					unsigned long long va = 0, vb = 0, vc = 0, vd = 0;
					rsb_sync_lds();
					reload_regs();
					for (;;) {
						const unsigned long long bit = 1ULL << (k2 & 63u);
						const uint32_t wi = k2 >> 6;
						const unsigned long long vw = wi == 0 ? va : wi == 1 ? vb : wi == 2 ? vc : vd;
						if (vw & bit) {
							// met before on this path: which stop was it?
							rsb_sync_lds();
							for (int qb = 0; qb < Lc2 && q0 < 0; qb += 64) { const unsigned long long hit = __ballot(qb + lane < Lc2 && L.pk[qb + lane < Lc2 ? qb + lane : 0] == k2); if (hit) q0 = qb + (__ffsll((long long)hit) - 1); }
							break;
						}
						if (Lc2 == 256) break;
						uint32_t y = entry_y(k2);
						if (y == 0) { y = load_entry(k2, L.head[k2], L.tail[k2]).y; rsb_sync_lds(); reload_regs(); }
						const uint32_t dd = y & 255u;
						if (dd == k2) { home = true; break; }
						if (lane == 0) L.pk[Lc2] = k2;
						if (wi == 0) va |= bit; else if (wi == 1) vb |= bit; else if (wi == 2) vc |= bit; else vd |= bit;
						++Lc2;
						if (dd == i) break;
						k2 = dd;
					}
					rsb_sync_lds();
					fill_ppos(Lc2);
					if (q0 >= 0) {
						for (int q = q0; q < Lc2; ++q) { const uint32_t r2 = HW[L.pk[q]].y >> 8; if (r2 < T) T = r2; }     // (peek left every stop's entry valid)
					}
					const int n_plain = (q0 >= 0 && T >= 2) ? q0 : Lc2;
					// plain steps: stop q takes what stop q-1 held, the first one the carry
					bool stale1 = false;
					for (int qb = 0; qb < n_plain; qb += 64) {
						const int q = qb + lane;
						if (q < n_plain) {
							const uint32_t pp = L.ppos[q], kq = L.pk[q];
							G.lg[G.n1 + (uint32_t)q] = make_uint2(pp, q == 0 ? carry : L.ppos[q - 1]);
							L.head[kq] = pp + 1;
							stale1 |= moved_entry(kq, pp, 1u);
						}
					}
					if (n_plain > 0) { G.n1 += (uint32_t)n_plain; carry = L.ppos[n_plain - 1]; }
					if (lane == 0) L.prof[3] += (unsigned long long)n_plain;
					rsb_sync_lds();
					if (__ballot(stale1)) refresh(0, n_plain);
					if (n_plain < Lc2) {
						// T rounds of the loop [q0, Lc2): stop j takes the records of stop j-1, the first stop those of the last stop one round earlier
						if (lane == 0) { L.prof[0] += 1; L.prof[1] += T; }
						const uint32_t p_first = L.ppos[q0], p_last = L.ppos[Lc2 - 1];
						for (int j = Lc2 - 1; j > q0; --j) rsl_run(G, L.ppos[j], L.ppos[j - 1], T, lane);
						rsl_one(G, p_first, carry, lane);
						rsl_run(G, p_first + 1, p_last, T - 1, lane);
						carry = p_last + T - 1;
						bool stale = false;
						for (int q = q0 + lane; q < Lc2; q += 64) { const uint32_t kq = L.pk[q]; L.head[kq] = L.ppos[q] + T; stale |= moved_entry(kq, L.ppos[q], T); }
						rsb_sync_lds();
						if (__ballot(stale)) refresh(q0, Lc2);
					}
					dst = (uint32_t)dig[carry];
					if (!home || dst == i) continue;
					// the head of dst holds a record that is home: one step, or the whole run of home records moved up in bulk
					{
						const uint32_t hd = L.head[dst], tk = L.tail[dst];
						uint32_t dd, rem;
						peek(dst, hd, tk, dd, rem);
						const uint32_t p = hd + rem;                         // end of the digit run at the head (clipped to the tail)
						if (rem >= 8 && p < tk) {
							rsl_run(G, hd + 1, hd, rem, lane);
							rsl_one(G, hd, carry, lane);
							carry = p;
							if (lane == 0) { L.head[dst] = p + 1; HW[dst] = none; }
						} else {
							rsl_one(G, hd, carry, lane);
							carry = hd;
							if (lane == 0) { L.head[dst] = hd + 1; (void)moved_entry(dst, hd, 1u); }
						}
						rsb_sync_lds();
						if (lane == 0) L.prof[3] += 1;
						dst = (uint32_t)dig[carry];
					}
				}
				rsl_one(G, h, carry, lane);
				if (lane == 0) { (void)moved_entry(i, h, 1u); L.head[i] = h + 1; wp[3] += wall_clock64() - tq0; }
				++h;
				rsb_sync_lds();
			}
			if (lane == 0) L.head[i] = h;
			rsb_sync_lds();
		}
	}
}

// ---- rs_level_sim (pga_sort_wave.h) with the shared log: the walk over LDS windows of the digit bytes, for levels with short digit runs ----
__device__ inline void rsb_walk_digits(const uint8_t *dig, RsLds &L, int lane, const unsigned long long (&nonempty)[4], uint32_t n_ne, RsLog &G)
{
	int wdl = 6;
	while (wdl < 12 && (n_ne << (wdl + 1)) <= (uint32_t)(RS_POOL * 16)) ++wdl;
	const uint32_t WD = 1u << wdl;
	uint8_t *dwin = (uint8_t*)L.win;
	{
		uint32_t rank0 = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int b = lane + 64 * k;
			L.wbase[b] = RS_NONE;
			L.wslot[b] = (uint8_t)(rank0 + (uint32_t)__popcll(nonempty[k] & ((1ULL << lane) - 1)));
			rank0 += (uint32_t)__popcll(nonempty[k]);
		}
	}
	rsb_sync_lds();
	auto cover = [&](uint32_t k, uint32_t pos, uint32_t ws) -> uint32_t {
		const uint32_t wb = L.wbase[k];
		if (wb != RS_NONE && pos - wb < WD) return wb;
		const uint32_t tk = L.tail[k];
		for (uint32_t i = (uint32_t)lane; i < WD; i += 64) if (pos + i < tk) dwin[ws + i] = dig[pos + i];
		if (lane == 0) L.wbase[k] = pos;
		rsb_sync_lds();
		return pos;
	};
#pragma unroll 1
	for (int kk = 0; kk < 4; ++kk) {
		unsigned long long todo = nonempty[kk];
		while (todo) {
			const uint32_t i = (uint32_t)(64 * kk + (__ffsll((long long)todo) - 1));
			todo &= todo - 1;
			uint32_t h = L.head[i]; const uint32_t tl = L.tail[i];
			const uint32_t wsi = (uint32_t)L.wslot[i] << wdl;
			while (h < tl) {
				const uint32_t wb = cover(i, h, wsi);
				const uint32_t p = h + (uint32_t)lane;
				const bool inw = p - wb < WD && p < tl;
				const unsigned long long vm = __ballot(inw), fm = __ballot(inw && (uint32_t)dwin[wsi + (p - wb)] != i);
				if (!fm) { h += (uint32_t)__popcll(vm); continue; }
				h += (uint32_t)(__ffsll((long long)fm) - 1);
				uint32_t src = h, k = (uint32_t)dwin[wsi + (h - wb)];
				do {
					const uint32_t pos = L.head[k];
					const uint32_t ws = (uint32_t)L.wslot[k] << wdl;
					const uint32_t wbk = cover(k, pos, ws);
					const uint32_t dd = (uint32_t)dwin[ws + (pos - wbk)];
					if (lane == 0) { G.lg[G.n1] = make_uint2(pos, src); L.head[k] = pos + 1; }
					rsb_sync_lds();
					++G.n1;
					src = pos; k = dd;
				} while (k != i);
				if (lane == 0) G.lg[G.n1] = make_uint2(h, src);
				++G.n1;
				++h;
			}
			if (lane == 0) L.head[i] = h;
			rsb_sync_lds();
		}
	}
	if (lane == 0) { L.prof[2] += 1; L.prof[3] += G.n1; }
}

// ---- final[dst + m] = original[src + m] for everything the walk logged: gather into tmp, barrier, scatter (all waves) ----
__device__ inline void rsb_apply(u128 *beg, u128 *tmp, const uint2 *lg, const uint4 *blg, uint32_t n1, uint32_t n2, uint32_t tot, int tid)
{
	const int wave = tid >> 6, lane = tid & 63;
	(void)tot;
	for (uint32_t e0 = 0; e0 < n1; e0 += RSB_NT * 4) {
		u128 v[4]; uint32_t e[4], src[4];
#pragma unroll
		for (int c = 0; c < 4; ++c) { e[c] = e0 + (uint32_t)tid + (uint32_t)(RSB_NT * c); src[c] = lg[e[c] < n1 ? e[c] : n1 - 1].y; }
#pragma unroll
		for (int c = 0; c < 4; ++c) rs_pin(src[c]);
#pragma unroll
		for (int c = 0; c < 4; ++c) v[c] = ld128(&beg[src[c]]);
#pragma unroll
		for (int c = 0; c < 4; ++c) rs_pin(v[c]);
#pragma unroll
		for (int c = 0; c < 4; ++c) if (e[c] < n1) tmp[e[c]] = v[c];
	}
	for (uint32_t p = (uint32_t)wave; p < n2; p += RSB_NW) {
		const uint4 ev = blg[p];
		for (uint32_t m0 = 0; m0 < ev.z; m0 += 256) {
			u128 v[4];
#pragma unroll
			for (int c = 0; c < 4; ++c) { const uint32_t m = m0 + (uint32_t)lane + 64u * c; v[c] = ld128(&beg[ev.y + (m < ev.z ? m : ev.z - 1)]); }
#pragma unroll
			for (int c = 0; c < 4; ++c) rs_pin(v[c]);
#pragma unroll
			for (int c = 0; c < 4; ++c) { const uint32_t m = m0 + (uint32_t)lane + 64u * c; if (m < ev.z) tmp[n1 + ev.w + m] = v[c]; }
		}
	}
	rs_fence_wg();
	__syncthreads();
	for (uint32_t e0 = 0; e0 < n1; e0 += RSB_NT * 4) {
		u128 v[4]; uint32_t e[4], dst[4];
#pragma unroll
		for (int c = 0; c < 4; ++c) { e[c] = e0 + (uint32_t)tid + (uint32_t)(RSB_NT * c); v[c] = ld128(&tmp[e[c] < n1 ? e[c] : n1 - 1]); dst[c] = lg[e[c] < n1 ? e[c] : n1 - 1].x; }
#pragma unroll
		for (int c = 0; c < 4; ++c) { rs_pin(v[c]); rs_pin(dst[c]); }
#pragma unroll
		for (int c = 0; c < 4; ++c) if (e[c] < n1) beg[dst[c]] = v[c];
	}
	for (uint32_t p = (uint32_t)wave; p < n2; p += RSB_NW) {
		const uint4 ev = blg[p];
		for (uint32_t m0 = 0; m0 < ev.z; m0 += 256) {
			u128 v[4];
#pragma unroll
			for (int c = 0; c < 4; ++c) { const uint32_t m = m0 + (uint32_t)lane + 64u * c; v[c] = ld128(&tmp[n1 + ev.w + (m < ev.z ? m : ev.z - 1)]); }
#pragma unroll
			for (int c = 0; c < 4; ++c) rs_pin(v[c]);
#pragma unroll
			for (int c = 0; c < 4; ++c) { const uint32_t m = m0 + (uint32_t)lane + 64u * c; if (m < ev.z) beg[ev.x + m] = v[c]; }
		}
	}
	rs_fence_wg();
	__syncthreads();
}

// a bucket of 2..64 records, stably sorted by one wave (ksort.h:107-117 is a stable insertion sort: record i lands at rank
// #{j: x_j < x_i or (x_j == x_i and j < i)})
__device__ __forceinline__ void rsb_stable64(u128 *b, uint32_t m, int lane)
{
	u128 mine; mine.x = ~0ULL, mine.y = 0;
	if ((uint32_t)lane < m) mine = ld128(&b[lane]);
	uint32_t rank = 0; bool moved = false;
	for (uint32_t j = 0; j < m; ++j) {
		const uint64_t xj = (uint64_t)(uint32_t)rl32((int)(uint32_t)mine.x, (int)j) | (uint64_t)(uint32_t)rl32((int)(uint32_t)(mine.x >> 32), (int)j) << 32;
		rank += (xj < mine.x) | ((xj == mine.x) & (j < (uint32_t)lane));
	}
	moved = (uint32_t)lane < m && rank != (uint32_t)lane;
	if (__ballot(moved)) { if ((uint32_t)lane < m) b[rank] = mine; }
}

} // namespace pga
