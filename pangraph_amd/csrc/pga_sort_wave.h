// pga_sort_wave.h -- the same permutation as pga_sort_exact.h (minimap2's unstable radix_sort_128x,
// ksort.h:101-151), executed by a whole wavefront (single-wave workgroups only).
//
// The cycle-leader walk of one level is sequential by definition; what the wave buys:
//   * levels whose digit is the same in every key of the array are skipped outright (one OR/AND pass finds them):
//     such a level leaves every run untouched (all records "home", ksort.h:132);
//   * the digit histogram (LDS atomics, 64 records per step);
//   * the long stretches of records that are already home: 64 records are tested per step, so an almost sorted
//     array -- anchors of a co-linear alignment, chain scores along a chain -- costs ~n/64 steps there;
//   * displacement cycles run against LDS: every bucket's head only moves forward and every slot at a head is
//     read once and written once, so each bucket gets a RS_WIN-record write-back window in LDS (filled and
//     flushed with coalesced accesses); one step of a cycle is an LDS exchange instead of a global round trip;
//   * insertion sorts of runs of <= 64 records (ksort.h:142) happen in LDS, one lane per run.
// Every lane of the wave must call these functions with identical arguments.
#pragma once
#include "pga_common.h"
#include "pga_sort_exact.h"
#include "pga_wave.h"

namespace pga {

#define RS_POOL 2048          // records of window space shared by the non-empty buckets of a level (32 KB: four waves per CU)
#define RS_NONE 0xffffffffu

struct __attribute__((aligned(16))) RsLds {
	u128 win[RS_POOL];           // bucket b owns win[wslot[b] << wlog .. +(1 << wlog)): a write-back window over [wbase, wbase + (1 << wlog))
	u128 ins[64];                // insertion-sort staging
	uint32_t head[256], tail[256], wbase[256];
	uint8_t wslot[256];
	uint32_t ppos[256]; uint32_t pk[256];   // the cycle being followed by the run-length walk: slot and bucket of every stop
	uint32_t hpos[256], hrem[256]; uint16_t hdig[256];   // what sits at a bucket's head (digit, remainder of its digit run), valid while head == hpos: following a cycle reads LDS only
	uint32_t vmark[256];         // bucket -> (path stamp << 8 | stop index) of its last visit: "was this bucket a stop of the path being followed" is one read
	unsigned long long prof[4];  // ticks (diagnostics)
	int run_min;                 // see rs_level_wave (PGA_RS_RUN_MIN)
};

__device__ __forceinline__ u128 ld128(const u128 *p) { u128 v; v.x = p->x; v.y = p->y; return v; }
// a use the compiler cannot move: loads issued in front of it stay in front (without it a load whose value is only needed under a
// condition is sunk into that branch with a wait of its own: four loads "in flight" become four round trips)
__device__ __forceinline__ void rs_pin(u128 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
__device__ __forceinline__ void rs_pin(uint32_t &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ int rl32(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
__device__ __forceinline__ void rs_fence_wave() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
__device__ __forceinline__ void rs_fence_wg() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// write the consumed part of bucket d's window back (positions [wbase, head))
__device__ __forceinline__ void rs_flush(u128 *beg, RsLds &L, int d, int wlog, int lane)
{
	const uint32_t wb = L.wbase[d];
	if (wb == RS_NONE) return;
	const uint32_t cnt = L.head[d] - wb;
	if ((uint32_t)lane < cnt) beg[wb + lane] = L.win[((uint32_t)L.wslot[d] << wlog) + lane];
	rs_fence_wave();
	if (lane == 0) L.wbase[d] = RS_NONE;
	rs_fence_wave();
}

// ---- the walk of ksort.h:128-141 over RUNS of records ----
// The reference's walk is one token moving between buckets: at bucket l it drops the record it carries at the bucket's head and
// picks up the record that was there, whose digit is the next bucket.  Slots at or beyond a head still hold their original
// records, so where the token goes depends only on the ORIGINAL digit sequence in front of the heads.  If the records at the
// heads of all buckets of a cycle (leader i -> l1 -> l2 ... -> back to i) are each followed by M-1 records with the same digit,
// the next M-1 cycles visit the same buckets one slot further along: M cycles are M independent rotations of disjoint slots, one
// lane each.  Anchor positions and chain scores grow or fall steadily along an array, so at the upper levels a digit run is
// hundreds to thousands of records long and a level that costs 190 ns per record as a token walk is a few hundred such events.
// rend[p] = end of the digit run of the ORIGINAL record at p.  Cycles that are not simple (a bucket met twice: the token is pushing
// a run of records that are already home) fall back to the token walk of that one cycle, home runs moved in bulk.
__device__ inline void rs_walk_runs(u128 *beg, int shift, const uint32_t *rend, RsLds &L, int lane, const unsigned long long (&nonempty)[4])
{
	auto digit_at = [&](uint32_t p) -> uint32_t { return (uint32_t)((beg[p].x >> shift) & 255); };
	// digit and run remainder of the record at the head of bucket k (uniform arguments): from LDS while the head has not moved since
	// the entry was made, else from memory (and remembered)
	auto peek = [&](uint32_t k, uint32_t pos, uint32_t tk, uint32_t &dd, uint32_t &rem) {
		if (L.hpos[k] == pos) { dd = L.hdig[k]; rem = L.hrem[k]; return; }
		dd = digit_at(pos);
		uint32_t r2 = rend[pos]; if (r2 > tk) r2 = tk;
		rem = r2 - pos;
		if (lane == 0) { L.hpos[k] = pos; L.hdig[k] = (uint16_t)dd; L.hrem[k] = rem; }
		rs_fence_wave();
	};
	// the heads of the buckets pk[q0..q1) have moved: one parallel round of loads brings their entries up to date
	auto refresh = [&](int q0, int q1) {
		for (int q = q0 + lane; q < q1; q += 64) {
			const uint32_t k = L.pk[q], pos = L.head[k], tk = L.tail[k];
			if (pos < tk) { uint32_t r2 = rend[pos]; if (r2 > tk) r2 = tk; L.hpos[k] = pos; L.hdig[k] = (uint16_t)digit_at(pos); L.hrem[k] = r2 - pos; }
			else L.hpos[k] = RS_NONE;
		}
		rs_fence_wave();
	};
	for (int k = lane; k < 256; k += 64) {
		const uint32_t pos = L.head[k], tk = L.tail[k];
		if (pos < tk) { uint32_t r2 = rend[pos]; if (r2 > tk) r2 = tk; L.hpos[k] = pos; L.hdig[k] = (uint16_t)digit_at(pos); L.hrem[k] = r2 - pos; }
		else L.hpos[k] = RS_NONE;
		L.vmark[k] = 0;
	}
	rs_fence_wave();
	uint32_t stamp = 0;                                     // (24 bits: a level of 2^24 paths would clear the marks; arrays are far shorter)
#pragma unroll 1
	for (int kk = 0; kk < 4; ++kk) {
		unsigned long long todo = nonempty[kk];
		while (todo) {
			const uint32_t i = (uint32_t)(64 * kk + (__ffsll((long long)todo) - 1));
			todo &= todo - 1;
			uint32_t h = L.head[i]; const uint32_t tl = L.tail[i];
			while (h < tl) {
				const uint32_t d0 = digit_at(h);
				uint32_t re = rend[h]; if (re > tl) re = tl;
				if (d0 == i) { h = re; continue; }                      // a run of records that are home already
				// follow the cycle that starts with beg[h] without moving anything
				uint32_t M = re - h, k = d0; int Lc = 0; bool simple = true;
				++stamp;
				for (;;) {
					if ((L.vmark[k] >> 8) == stamp || Lc == 256) { simple = false; break; }
					const uint32_t pos = L.head[k], tk = L.tail[k];
					uint32_t dd, r2;
					peek(k, pos, tk, dd, r2);
					if (dd == k) { simple = false; break; }               // the record at the head is home: the token would push it along
					if (lane == 0) { L.ppos[Lc] = pos; L.pk[Lc] = k; L.vmark[k] = stamp << 8 | (uint32_t)Lc; }
					rs_fence_wave();
					++Lc;
					if (r2 < M) M = r2;
					if (dd == i) break;
					k = dd;
				}
				if (simple) {
					if (lane == 0) { L.prof[0] += 1; L.prof[1] += M; }
					// M rotations, one per lane: the record of the leader's slot goes to the first stop, every stop's record to the next
					// stop, the last one (digit i) into the leader's slot
					// (eight stops per trip: their loads are independent of one another, so they are in flight together)
					for (uint32_t m = (uint32_t)lane; m < M; m += 64) {
						u128 t = ld128(&beg[h + m]);
						for (int q0 = 0; q0 < Lc; q0 += 8) {
							u128 v[8];
#pragma unroll
							for (int c = 0; c < 8; ++c) v[c] = ld128(&beg[L.ppos[q0 + c < Lc ? q0 + c : Lc - 1] + m]);   // (unconditional and pinned: see rs_pin)
#pragma unroll
							for (int c = 0; c < 8; ++c) rs_pin(v[c]);
#pragma unroll
							for (int c = 0; c < 8; ++c) if (q0 + c < Lc) { beg[L.ppos[q0 + c] + m] = t; t = v[c]; }
						}
						beg[h + m] = t;
					}
					for (int q = lane; q < Lc; q += 64) L.head[L.pk[q]] += M;   // (distinct buckets: the cycle is simple)
					rs_fence_wave();
					refresh(0, Lc);
					h += M;
					continue;
				}
				// the token walk of this one cycle (every lane follows it; lane 0 owns the stores).  A cycle over mirrored runs bounces
				// between the same few buckets for as long as the digit runs at their heads last (k -> k' -> k -> k' ...): from the
				// token's position the path is followed until it meets a bucket it has already stopped at; the stops from there on
				// form a LOOP that repeats T times, T = the shortest run remainder among them, and T rounds are again independent
				// slot-to-slot copies (stop j takes the records of stop j-1, the first stop those of the last stop one round earlier)
				if (lane == 0) L.prof[2] += 1;
				u128 carry = ld128(&beg[h]);
				uint32_t dst = d0;
				while (dst != i) {
					int Lc = 0, q0 = -1; uint32_t k = dst, T = 0xffffffffu; bool home = false;
					++stamp;
					for (;;) {
						const uint32_t vm = L.vmark[k];
						if ((vm >> 8) == stamp) { q0 = (int)(vm & 255u); break; }
						if (Lc == 256) break;
						const uint32_t pos = L.head[k], tk = L.tail[k];
						uint32_t dd, r2u;
						peek(k, pos, tk, dd, r2u);
						if (dd == k) { home = true; break; }
						if (lane == 0) { L.ppos[Lc] = pos; L.pk[Lc] = k; L.vmark[k] = stamp << 8 | (uint32_t)Lc; }
						rs_fence_wave();
						++Lc;
						if (dd == i) break;
						k = dd;
					}
					if (q0 >= 0) {
						for (int q = q0; q < Lc; ++q) { const uint32_t pos = L.ppos[q], tk = L.tail[L.pk[q]]; uint32_t r2 = rend[pos]; if (r2 > tk) r2 = tk; r2 -= pos; if (r2 < T) T = r2; }
					}
					// the stops before the loop (or all of them, when there is no loop worth taking): ordinary steps
					const int n_plain = (q0 >= 0 && T >= 2) ? q0 : Lc;
					// (the stops are distinct slots and the path is known: stop q takes what stop q-1 held, the first one the carry -- 64 stops per
					// trip with all loads in flight together, instead of one dependent load-store round trip to device memory per stop)
					for (int qb = 0; qb < n_plain; qb += 64) {
						const int q = qb + lane;
						const bool on = q < n_plain;
						u128 old; old.x = 0, old.y = 0;
						uint32_t pp = 0;
						if (on) { pp = L.ppos[q]; old = ld128(&beg[pp]); }
						u128 in;
						in.x = (uint64_t)(uint32_t)wave_shr1((int)(uint32_t)old.x, (int)(uint32_t)carry.x) | (uint64_t)(uint32_t)wave_shr1((int)(uint32_t)(old.x >> 32), (int)(uint32_t)(carry.x >> 32)) << 32;
						in.y = (uint64_t)(uint32_t)wave_shr1((int)(uint32_t)old.y, (int)(uint32_t)carry.y) | (uint64_t)(uint32_t)wave_shr1((int)(uint32_t)(old.y >> 32), (int)(uint32_t)(carry.y >> 32)) << 32;
						if (on) { beg[pp] = in; L.head[L.pk[q]] = pp + 1; }
						const int top = n_plain - qb > 64 ? 63 : n_plain - qb - 1;
						carry.x = (uint64_t)(uint32_t)rl32((int)(uint32_t)old.x, top) | (uint64_t)(uint32_t)rl32((int)(uint32_t)(old.x >> 32), top) << 32;
						carry.y = (uint64_t)(uint32_t)rl32((int)(uint32_t)old.y, top) | (uint64_t)(uint32_t)rl32((int)(uint32_t)(old.y >> 32), top) << 32;
					}
					if (lane == 0) L.prof[3] += (unsigned long long)n_plain;
					rs_fence_wg();
					if (n_plain < Lc) {
						// T rounds of the loop [q0, Lc)
						if (lane == 0) { L.prof[0] += 1; L.prof[1] += T; }
						const uint32_t p_first = L.ppos[q0], p_last = L.ppos[Lc - 1];
						for (uint32_t t0 = 0; t0 < T; t0 += 64) {
							const uint32_t t = t0 + (uint32_t)lane;
							const bool on = t < T;
							u128 last; last.x = 0, last.y = 0;
							if (on) last = ld128(&beg[p_last + t]);
							for (int j = Lc - 1; j > q0; --j) if (on) { const u128 v = ld128(&beg[L.ppos[j - 1] + t]); beg[L.ppos[j] + t] = v; }
							// the first stop takes the last stop's record of the round before: lane l from lane l-1, lane 0 the carry
							u128 prev;
							prev.x = (uint64_t)(uint32_t)wave_shr1((int)(uint32_t)last.x, (int)(uint32_t)carry.x) | (uint64_t)(uint32_t)wave_shr1((int)(uint32_t)(last.x >> 32), (int)(uint32_t)(carry.x >> 32)) << 32;
							prev.y = (uint64_t)(uint32_t)wave_shr1((int)(uint32_t)last.y, (int)(uint32_t)carry.y) | (uint64_t)(uint32_t)wave_shr1((int)(uint32_t)(last.y >> 32), (int)(uint32_t)(carry.y >> 32)) << 32;
							if (on) beg[p_first + t] = prev;
							const int top = (int)(T - t0 > 64 ? 63 : T - t0 - 1);
							carry.x = (uint64_t)(uint32_t)rl32((int)(uint32_t)last.x, top) | (uint64_t)(uint32_t)rl32((int)(uint32_t)(last.x >> 32), top) << 32;
							carry.y = (uint64_t)(uint32_t)rl32((int)(uint32_t)last.y, top) | (uint64_t)(uint32_t)rl32((int)(uint32_t)(last.y >> 32), top) << 32;
						}
						for (int q = q0 + lane; q < Lc; q += 64) L.head[L.pk[q]] = L.ppos[q] + T;
						rs_fence_wg();
						refresh(q0, Lc);
					}
					dst = (uint32_t)((carry.x >> shift) & 255);
					if (!home || dst == i) continue;
					// the head of dst holds a record that is home: one step, or the whole run of home records moved up in bulk
					{
						const uint32_t hd = L.head[dst], tk = L.tail[dst];
						uint32_t p = rend[hd]; if (p > tk) p = tk;
						if (p - hd >= 8 && p < tk) {
							const u128 out = ld128(&beg[p]);
							for (uint32_t hi = p; hi > hd; ) {
								u128 v[4]; uint32_t q[4];
#pragma unroll
								for (int c = 0; c < 4; ++c) { q[c] = hi - (uint32_t)lane - 64u * c; v[c] = ld128(&beg[(int64_t)hi - lane - 64 * c > (int64_t)hd ? q[c] - 1 : hd]); }
#pragma unroll
								for (int c = 0; c < 4; ++c) rs_pin(v[c]);
#pragma unroll
								for (int c = 0; c < 4; ++c) if ((int64_t)hi - lane - 64 * c > (int64_t)hd) beg[q[c]] = v[c];
								hi = hi - hd > 256 ? hi - 256 : hd;
							}
							if (lane == 0) { beg[hd] = carry; L.head[dst] = p + 1; }
							rs_fence_wg();
							carry = out;
						} else {
							const u128 nxt = ld128(&beg[hd]);
							if (lane == 0) { beg[hd] = carry; L.head[dst] = hd + 1; }
							rs_fence_wave();
							carry = nxt;
						}
						if (lane == 0) L.prof[3] += 1;
						dst = (uint32_t)((carry.x >> shift) & 255);
					}
				}
				if (lane == 0) beg[h] = carry;
				++h;
			}
			if (lane == 0) L.head[i] = h;
			rs_fence_wave();
		}
	}
	rs_fence_wg();
}

// one level (ksort.h:118-146) on [beg, beg+n); false if every record has the same digit (the walk is the identity)
// ---- a level with exactly TWO non-empty buckets (the strand bit, the target id of a two-sequence group): closed form ----
// With buckets A < B, region A = [0, cA) and region B = [cA, n), the walk of ksort.h:128-141 does nothing but this: it scans region A;
// a record of bucket A stays; the t-th record of bucket B found there (at i_t) is carried to the head of region B, where it pushes the
// B records one slot to the right until the t-th A record of region B (at q_t) falls out, which lands at i_t.  Heads only move
// forward, so with h_0 = cA, h_t = q_(t-1) + 1:
//     new[i_t] = old[q_t]      new[h_t] = old[i_t]      new[p + 1] = old[p] for the B records p in [h_t, q_t)      everything else stays
// -- three coalesced passes with two running counts instead of one dependent step per record.  `tmp` receives the new arrangement
// (same length as the run), idx holds the 2m positions i_t, q_t (one word per record is enough: m <= n/2).
// Returns false (nothing moved) when only a few records are out of place: the walk skips home records 64 at a time and is faster then.
__device__ inline bool rs_level_two(u128 *beg, int64_t n, int shift, uint32_t dA, uint32_t cA, u128 *tmp, uint32_t *idx, int lane)
{
	const unsigned long long lt = (1ULL << lane) - 1;
	// pass 1: positions of the misplaced records: I[t] = idx[t], Q[t] = idx[m + t]; m is not known yet, so Q is written from the back
	uint32_t mA = 0, mB = 0;
	for (int64_t c0 = 0; c0 < n; c0 += 256) {
		bool mis[4]; int64_t p[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) { p[k] = c0 + lane + 64 * k; const uint64_t x = beg[p[k] < n ? p[k] : n - 1].x; const bool isA = (uint32_t)((x >> shift) & 255) == dA; mis[k] = (p[k] < n) & (p[k] < (int64_t)cA ? !isA : isA); }   // (no short circuit in front of the load's use: the compiler would sink the load into a branch of its own)
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const unsigned long long bm = __ballot(mis[k]);
			const int64_t blk = c0 + 64 * k;                      // a 64-block lies in one region or straddles cA: count per side
			const unsigned long long inA = blk + 64 <= (int64_t)cA ? ~0ULL : blk >= (int64_t)cA ? 0ULL : ((1ULL << (cA - blk)) - 1);
			if (mis[k]) { if (p[k] < (int64_t)cA) idx[mA + (uint32_t)__popcll(bm & inA & lt)] = (uint32_t)p[k]; else idx[(uint32_t)n - 1 - (mB + (uint32_t)__popcll(bm & ~inA & lt))] = (uint32_t)p[k]; }
			mA += (uint32_t)__popcll(bm & inA); mB += (uint32_t)__popcll(bm & ~inA);
		}
	}
	rs_fence_wg();
	const uint32_t m = mA;                                        // == mB
	// few records out of place: the walk skips home records 64 at a time and is a little faster -- as long as the token steps it makes
	// for the misplaced ones (~0.8 us each, one after the other) stay short of what the three passes below cost
	if ((uint64_t)m * 16 < (uint64_t)n && m < 2048) return false;
	auto Qat = [&](uint32_t t) { return idx[(uint32_t)n - 1 - t]; };
	// pass 2: the new arrangement, out of place
	uint32_t tA = 0, tB = 0;
	for (int64_t c0 = 0; c0 < n; c0 += 256) {
		u128 v[4]; int64_t p[4]; bool isA[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) { p[k] = c0 + lane + 64 * k; v[k] = ld128(&beg[p[k] < n ? p[k] : n - 1]); }
#pragma unroll
		for (int k = 0; k < 4; ++k) { rs_pin(v[k]); isA[k] = (p[k] < n) & ((uint32_t)((v[k].x >> shift) & 255) == dA); }
		// (ranks first, then all eight index lookups together, then the stores: a lookup under `if (misplaced)` is a round trip of its own)
		uint32_t tq[4], uq[4], qv[4], iv[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const bool on = p[k] < n, inA = p[k] < (int64_t)cA;
			const bool mis = on & (inA ? !isA[k] : isA[k]);
			const unsigned long long bm = __ballot(mis);
			const int64_t blk = c0 + 64 * k;
			const unsigned long long sideA = blk + 64 <= (int64_t)cA ? ~0ULL : blk >= (int64_t)cA ? 0ULL : ((1ULL << (cA - blk)) - 1);
			tq[k] = tA + (uint32_t)__popcll(bm & sideA & lt);                   // B records of region A in front of p
			uq[k] = tB + (uint32_t)__popcll(bm & ~sideA & lt);                  // A records of region B in front of p
			tA += (uint32_t)__popcll(bm & sideA); tB += (uint32_t)__popcll(bm & ~sideA);
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t t1 = tq[k] ? tq[k] - 1 : 0;
			qv[k] = idx[(uint32_t)n - 1 - (t1 < (uint32_t)n ? t1 : (uint32_t)n - 1)];
			iv[k] = idx[uq[k] < (uint32_t)n ? uq[k] : (uint32_t)n - 1];
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) { rs_pin(qv[k]); rs_pin(iv[k]); }
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const bool on = p[k] < n, inA = p[k] < (int64_t)cA;
			if (on) {
				uint32_t dst;
				if (inA) dst = isA[k] ? (uint32_t)p[k] : (tq[k] == 0 ? cA : qv[k] + 1);
				else dst = isA[k] ? iv[k] : (uq[k] < m ? (uint32_t)p[k] + 1 : (uint32_t)p[k]);   // the u-th A record of region B goes to i_u
				tmp[dst] = v[k];
			}
		}
	}
	rs_fence_wg();
	// pass 3: back in place
	for (int64_t c0 = 0; c0 < n; c0 += 256) {
		u128 v[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) { const int64_t p = c0 + lane + 64 * k; v[k] = ld128(&tmp[p < n ? p : n - 1]); }
#pragma unroll
		for (int k = 0; k < 4; ++k) rs_pin(v[k]);
#pragma unroll
		for (int k = 0; k < 4; ++k) { const int64_t p = c0 + lane + 64 * k; if (p < n) beg[p] = v[k]; }
	}
	rs_fence_wg();
	return true;
}

// ---- a level whose digit runs are SHORT (chain scores, the low bytes of positions): the walk of ksort.h:128-141 simulated on digits ----
// Where the token goes depends only on the ORIGINAL digit sequence (slots at or beyond a head still hold their original records), so the
// walk does not have to move records to know its way: it runs over a byte array of the original digits (read-only, cached per bucket in
// LDS windows of 128-4096 digits instead of 8-64 records: a refill every few hundred steps instead of every eighth), and writes down
// for every slot it fills where the record came from (each slot at a head is filled exactly once, from a slot that still held its
// original record).  One step is two dependent LDS reads (head, digit) instead of a 16-byte exchange through a window with its flush and
// refill traffic; the records then move in two parallel passes (gather into `tmp`, scatter back).
__device__ inline void rs_level_sim(u128 *beg, int64_t n, int shift, RsLds &L, int lane, const uint32_t (&cnt)[4], const uint32_t (&off)[4], const unsigned long long (&nonempty)[4], uint32_t n_ne,
                                    uint8_t *dig, uint2 *lg, u128 *tmp)
{
	for (int64_t i0 = 0; i0 < n; i0 += 256) {
		uint32_t dg[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) { const int64_t i = i0 + lane + 64 * k; dg[k] = (uint32_t)((beg[i < n ? i : n - 1].x >> shift) & 255); }
#pragma unroll
		for (int k = 0; k < 4; ++k) { const int64_t i = i0 + lane + 64 * k; if (i < n) dig[i] = (uint8_t)dg[k]; }
	}
	int wdl = 6;
	while (wdl < 12 && (n_ne << (wdl + 1)) <= (uint32_t)(RS_POOL * 16)) ++wdl;
	const uint32_t WD = 1u << wdl;
	uint8_t *dwin = (uint8_t*)L.win;
	{
		uint32_t rank0 = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int b = lane + 64 * k;
			L.head[b] = off[k]; L.tail[b] = off[k] + cnt[k]; L.wbase[b] = RS_NONE;
			L.wslot[b] = (uint8_t)(rank0 + (uint32_t)__popcll(nonempty[k] & ((1ULL << lane) - 1)));
			rank0 += (uint32_t)__popcll(nonempty[k]);
		}
	}
	rs_fence_wg();                                              // the digit bytes, for the lanes that fill windows from them
	// the window of bucket k covers position pos (uniform arguments); returns the window's base
	auto cover = [&](uint32_t k, uint32_t pos, uint32_t ws) -> uint32_t {
		const uint32_t wb = L.wbase[k];
		if (wb != RS_NONE && pos - wb < WD) return wb;
		const uint32_t tk = L.tail[k];
		for (uint32_t i = (uint32_t)lane; i < WD; i += 64) if (pos + i < tk) dwin[ws + i] = dig[pos + i];
		if (lane == 0) L.wbase[k] = pos;
		rs_fence_wave();
		return pos;
	};
	uint32_t n_log = 0;
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
				// records that are home already: 64 window digits per step
				const uint32_t p = h + (uint32_t)lane;
				const bool inw = p - wb < WD && p < tl;
				const unsigned long long vm = __ballot(inw), fm = __ballot(inw && (uint32_t)dwin[wsi + (p - wb)] != i);
				if (!fm) { h += (uint32_t)__popcll(vm); continue; }
				h += (uint32_t)(__ffsll((long long)fm) - 1);
				// the displacement cycle that starts with the record at h (ksort.h:131-138), on digits
				uint32_t src = h, k = (uint32_t)dwin[wsi + (h - wb)];
				do {
					const uint32_t pos = L.head[k];
					const uint32_t ws = (uint32_t)L.wslot[k] << wdl;
					const uint32_t wbk = cover(k, pos, ws);
					const uint32_t dd = (uint32_t)dwin[ws + (pos - wbk)];
					if (lane == 0) { lg[n_log] = make_uint2(pos, src); L.head[k] = pos + 1; }
					asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (LDS only: a wavefront fence would also wait for the log's global store, every step)
					++n_log;
					src = pos; k = dd;
				} while (k != i);
				if (lane == 0) lg[n_log] = make_uint2(h, src);
				++n_log;
				++h;
			}
			if (lane == 0) L.head[i] = h;
			rs_fence_wave();
		}
	}
	rs_fence_wg();
	if (lane == 0) { L.prof[2] += 1; L.prof[3] += n_log; }
	// the records follow: final[dst] = original[src]
	for (uint32_t e0 = 0; e0 < n_log; e0 += 256) {
		u128 v[4]; uint32_t e[4], src[4];
#pragma unroll
		for (int c = 0; c < 4; ++c) { e[c] = e0 + (uint32_t)lane + 64u * c; src[c] = lg[e[c] < n_log ? e[c] : n_log - 1].y; }
#pragma unroll
		for (int c = 0; c < 4; ++c) rs_pin(src[c]);
#pragma unroll
		for (int c = 0; c < 4; ++c) v[c] = ld128(&beg[src[c]]);
#pragma unroll
		for (int c = 0; c < 4; ++c) rs_pin(v[c]);
#pragma unroll
		for (int c = 0; c < 4; ++c) if (e[c] < n_log) tmp[e[c]] = v[c];
	}
	rs_fence_wg();
	for (uint32_t e0 = 0; e0 < n_log; e0 += 256) {
		u128 v[4]; uint32_t e[4], dst[4];
#pragma unroll
		for (int c = 0; c < 4; ++c) { e[c] = e0 + (uint32_t)lane + 64u * c; v[c] = ld128(&tmp[e[c] < n_log ? e[c] : n_log - 1]); dst[c] = lg[e[c] < n_log ? e[c] : n_log - 1].x; }
#pragma unroll
		for (int c = 0; c < 4; ++c) { rs_pin(v[c]); rs_pin(dst[c]); }
#pragma unroll
		for (int c = 0; c < 4; ++c) if (e[c] < n_log) beg[dst[c]] = v[c];
	}
	rs_fence_wg();
}

__device__ inline bool rs_level_wave(u128 *beg, int64_t n, int shift, RsLds &L, int lane, uint32_t (&cnt)[4], uint32_t (&off)[4], uint32_t *rend = nullptr, u128 *tmp = nullptr, uint2 *lg = nullptr)
{
	for (int d = lane; d < 256; d += 64) L.head[d] = 0, L.wbase[d] = RS_NONE;
	rs_fence_wave();
	// digit histogram, four loads in flight per lane
	uint32_t n_druns = 0, d_prev = 257u;                        // digit runs of the original order (counted here: the run-length walk and its
	for (int64_t i0 = 0; i0 < n; i0 += 256) {                   // backward pass over the array only pay when runs are long)
		uint32_t dg[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) { const int64_t i = i0 + lane + 64 * k; const uint32_t d = (uint32_t)((beg[i < n ? i : n - 1].x >> shift) & 255); dg[k] = i < n ? d : 256u; }
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			if (dg[k] < 256u) atomicAdd(&L.head[dg[k]], 1u);
			const uint32_t left = (uint32_t)wave_shr1((int)dg[k], (int)d_prev);
			n_druns += (uint32_t)__popcll(__ballot(dg[k] < 256u && dg[k] != left));
			d_prev = (uint32_t)__builtin_amdgcn_readlane((int)dg[k], 63);
		}
	}
	rs_fence_wave();
	// counts -> offsets: lane l holds buckets l, l+64, l+128, l+192
	unsigned long long nonempty[4];
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
	if (n_ne <= 1) return false;
	if (n_ne == 2 && tmp && rend && n >= 1024) {
		// which two buckets, and the size of the lower one (uniform)
		int dA = -1; uint32_t cA = 0;
#pragma unroll
		for (int k = 3; k >= 0; --k) if (nonempty[k]) { const int l = __ffsll((long long)nonempty[k]) - 1; dA = l + 64 * k; cA = (uint32_t)__builtin_amdgcn_readlane((int)cnt[k], l); }
		if (rs_level_two(beg, n, shift, (uint32_t)dA, cA, tmp, rend, lane)) return true;
	}
	const uint64_t run_min = (lg && tmp) ? (uint64_t)L.run_min : 64ull;      // mean digit run from which the run-length walk beats the walk on digits
	if (rend && n >= 4096 && (uint64_t)n >= run_min * n_druns) {
		// digit runs of the original order, back to front: rend[p] = first position after p whose digit differs
		// (256 records per trip with their four loads in flight -- unconditional, clamped indices: a load under `p < n ?` becomes a branch with
		// its own wait -- and the digit right above a chunk carried down instead of loaded again)
		uint32_t nb = 0, carry_end = (uint32_t)n, next_first = 257u;
		for (int64_t g0 = (n - 1) & ~255LL; g0 >= 0; g0 -= 256) {
			uint32_t dg[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) { const int64_t p = g0 + 64 * k + lane; const uint32_t d = (uint32_t)((beg[p < n ? p : n - 1].x >> shift) & 255); dg[k] = p < n ? d : 256u; }
#pragma unroll
			for (int k = 3; k >= 0; --k) {
				const int64_t p = g0 + 64 * k + lane;
				const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp((int)next_first, (int)dg[k], 0x130, 0xf, 0xf, false);   // wave_shl:1, lane 63 sees the record above the chunk
				const bool last_of_run = p < n && nx != dg[k];
				const unsigned long long bm = __ballot(last_of_run);
				nb += (uint32_t)__popcll(bm);
				const unsigned long long up = bm >> lane;              // run ends at or after this lane
				const uint32_t e = up ? (uint32_t)(p + (__ffsll((long long)up) - 1) + 1) : carry_end;
				if (p < n) rend[p] = e;
				// the run that reaches beyond this chunk's start continues into the chunk below: its end is the end of lane 0's run
				carry_end = (uint32_t)__builtin_amdgcn_readlane((int)e, 0);
				next_first = (uint32_t)__builtin_amdgcn_readlane((int)dg[k], 0);
			}
		}
		rs_fence_wg();
		if ((uint64_t)n >= run_min * nb) {                          // digit runs of 64+ records on average: below that the token walk through the LDS windows is faster
#pragma unroll
			for (int k = 0; k < 4; ++k) { const int b = lane + 64 * k; L.head[b] = off[k]; L.tail[b] = off[k] + cnt[k]; }
			rs_fence_wave();
			rs_walk_runs(beg, shift, rend, L, lane, nonempty);
			return true;
		}
	}
	if (rend && tmp && lg && n >= 512) { rs_level_sim(beg, n, shift, L, lane, cnt, off, nonempty, n_ne, (uint8_t*)rend, lg, tmp); return true; }
	// window space: the non-empty buckets share the pool, 16 to 64 records each
	int wlog = 3;
	while (wlog < 6 && (n_ne << (wlog + 1)) <= RS_POOL) ++wlog;
	const uint32_t W = 1u << wlog;
	{
		uint32_t rank0 = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int b = lane + 64 * k;
			L.head[b] = off[k]; L.tail[b] = off[k] + cnt[k];
			L.wslot[b] = (uint8_t)(rank0 + (uint32_t)__popcll(nonempty[k] & ((1ULL << lane) - 1)));
			rank0 += (uint32_t)__popcll(nonempty[k]);
		}
	}
	rs_fence_wave();
#pragma unroll 1
	for (int k = 0; k < 4; ++k) {
		unsigned long long todo = nonempty[k];
		while (todo) {
			const int d = 64 * k + (__ffsll((long long)todo) - 1);
			todo &= todo - 1;
			uint32_t h = L.head[d]; const uint32_t tl = L.tail[d];
			rs_flush(beg, L, d, wlog, lane);                        // slots filled while d was a destination
			// bucket d is scanned through a 64-record register window (slots at or beyond a head still hold the original
			// records, and nothing but this loop writes them while d is the current bucket)
			uint32_t cw = h; unsigned long long fm = 0; bool have = false, dirty = false;
			u128 rec; rec.x = 0, rec.y = 0;
			while (h < tl) {
				if (!have || h - cw >= 64) {
					if (dirty) beg[cw + (uint32_t)lane] = rec;
					cw = h; dirty = false; have = true;
					const uint32_t pos = cw + (uint32_t)lane;
					if (pos < tl) rec = ld128(&beg[pos]);
					fm = __ballot(pos < tl && (uint32_t)((rec.x >> shift) & 255) != (uint32_t)d);
				}
				const unsigned long long m = fm >> (h - cw);         // foreign records at positions >= h
				if (m == 0) { h = cw + 64; continue; }
				h += (uint32_t)(__ffsll((long long)m) - 1);
				// displacement chain starting at the foreign record beg[h] (all lanes follow it; lane 0 owns the LDS writes)
				const int sl = (int)(h - cw);
				u128 carry;
				carry.x = (uint64_t)(uint32_t)rl32((int)(uint32_t)rec.x, sl) | (uint64_t)(uint32_t)rl32((int)(uint32_t)(rec.x >> 32), sl) << 32;
				carry.y = (uint64_t)(uint32_t)rl32((int)(uint32_t)rec.y, sl) | (uint64_t)(uint32_t)rl32((int)(uint32_t)(rec.y >> 32), sl) << 32;
				int dst = (int)((carry.x >> shift) & 255);
				int same = 0;
				do {
					if (same >= 8) {
						// The last steps all landed in bucket dst and pushed out records that were already home there: the walk
						// is shifting a run of home records one slot to the right, one record per step (ksort.h:134-137 displaces
						// whatever sits at the head).  Do the whole run at once: find the first record at or after the head
						// that does not belong to dst, move everything before it up by one (top-down, 256 records per round),
						// put the carry at the head and continue with that record.
						same = 0;
						rs_flush(beg, L, dst, wlog, lane);
						const uint32_t hd0 = L.head[dst], tl0 = L.tail[dst];
						uint32_t p = hd0;
						for (;;) {                                        // a foreign record exists before the tail (slot counting)
							const uint32_t q = p + (uint32_t)lane;
							const bool foreign = q < tl0 && (uint32_t)((beg[q].x >> shift) & 255) != (uint32_t)dst;
							const unsigned long long fmk = __ballot(foreign || q >= tl0);
							if (fmk) { p += (uint32_t)(__ffsll((long long)fmk) - 1); break; }
							p += 64;
						}
						if (p < tl0 && p - hd0 >= 16) {
							const u128 out = ld128(&beg[p]);
							for (uint32_t hi = p; hi > hd0; ) {
								u128 v[4]; uint32_t q[4];
#pragma unroll
								for (int k = 0; k < 4; ++k) { q[k] = hi - (uint32_t)lane - 64u * k; v[k] = ld128(&beg[(int64_t)hi - lane - 64 * k > (int64_t)hd0 ? q[k] - 1 : hd0]); }
#pragma unroll
									for (int k = 0; k < 4; ++k) rs_pin(v[k]);
#pragma unroll
								for (int k = 0; k < 4; ++k) if ((int64_t)hi - lane - 64 * k > (int64_t)hd0) beg[q[k]] = v[k];
								hi = hi - hd0 > 256 ? hi - 256 : hd0;
							}
							if (lane == 0) { beg[hd0] = carry; L.head[dst] = p + 1; }
							rs_fence_wg();
							carry = out;
							dst = (int)((carry.x >> shift) & 255);
							continue;
						}
					}
					const uint32_t hd = L.head[dst];
					uint32_t wb = L.wbase[dst];
					const uint32_t ws = (uint32_t)L.wslot[dst] << wlog;
					if (wb == RS_NONE || hd - wb >= W) {
						rs_flush(beg, L, dst, wlog, lane);
						const uint32_t p2 = hd + (uint32_t)lane;
						if ((uint32_t)lane < W && p2 < L.tail[dst]) L.win[ws + lane] = ld128(&beg[p2]);
						if (lane == 0) L.wbase[dst] = hd;
						wb = hd;
						rs_fence_wave();
					}
					const u128 nxt = L.win[ws + (hd - wb)];
					rs_fence_wave();
					if (lane == 0) { L.win[ws + (hd - wb)] = carry; L.head[dst] = hd + 1; }
					rs_fence_wave();
					carry = nxt;
					{ const int nd = (int)((carry.x >> shift) & 255); same = nd == dst ? same + 1 : 0; dst = nd; }
				} while (dst != d);
				if (lane == sl) rec = carry, dirty = true;
				++h;
			}
			if (dirty) beg[cw + (uint32_t)lane] = rec;
			if (lane == 0) L.head[d] = h;
			rs_fence_wave();
		}
	}
	rs_fence_wg();
	return true;
}

// insertion-sort up to 64 records [b, e) through LDS: lane 0 sorts
__device__ __forceinline__ void rs_small_wave(u128 *beg, int64_t b, int64_t e, RsLds &L, int lane)
{
	const int m = (int)(e - b);
	if (m <= 1) return;
	if (lane < m) L.ins[lane] = ld128(&beg[b + lane]);
	rs_fence_wave();
	if (lane == 0) rs_insertion(L.ins, L.ins + m);
	rs_fence_wave();
	if (lane < m) beg[b + lane] = L.ins[lane];
	rs_fence_wg();
}

// After a level: the buckets (lane l holds buckets l, l+64, l+128, l+192: cnt/off from rs_level_wave) of <= 64 records are
// insertion-sorted (ksort.h:142), one lane per bucket, inside LDS -- the 64 buckets of a group are staged together when they
// fit the window pool; larger buckets are handed to big(offset, length).
template <class Big>
__device__ inline void rs_split_buckets(u128 *beg, int64_t n, int shift, const uint32_t (&cnt)[4], const uint32_t (&off)[4], RsLds &L, int lane, Big big)
{
#pragma unroll 1
	for (int k = 0; k < 4; ++k) {
		const uint32_t g_lo = (uint32_t)__builtin_amdgcn_readlane((int)off[k], 0);
		const uint32_t g_hi = (uint32_t)__builtin_amdgcn_readlane((int)(off[k] + cnt[k]), 63);
		if (g_hi == g_lo) continue;
		unsigned long long bm = __ballot(cnt[k] > 64);
		const unsigned long long sm = __ballot(cnt[k] > 1 && cnt[k] <= 64);
		while (bm) {
			const int src = __ffsll((long long)bm) - 1;
			bm &= bm - 1;
			big((int64_t)(uint32_t)__builtin_amdgcn_readlane((int)off[k], src), (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)cnt[k], src));
		}
		if (!sm) continue;
		// the small buckets of the group, staged in the window pool as many lanes' buckets at a time as fit (a 15 k-record run has
		// 64 x 60 records per group: twice the pool -- the generic window scan it used to fall back to cost 3.7 ms per run)
		int lane_lo = 0;
		while (lane_lo < 64) {
			int span = 64 - lane_lo;
			uint32_t s_lo, s_hi;
			for (;;) {
				s_lo = (uint32_t)__builtin_amdgcn_readlane((int)off[k], __builtin_amdgcn_readfirstlane(lane_lo));
				s_hi = (uint32_t)__builtin_amdgcn_readlane((int)(off[k] + cnt[k]), __builtin_amdgcn_readfirstlane(lane_lo + span - 1));
				if (s_hi - s_lo <= RS_POOL || span == 1) break;
				span = (span + 1) / 2;
			}
			const bool mine_in = lane >= lane_lo && lane < lane_lo + span;
			if (s_hi - s_lo <= RS_POOL && __ballot(mine_in && cnt[k] > 1 && cnt[k] <= 64)) {
				for (uint32_t i = s_lo + (uint32_t)lane; i < s_hi; i += 64) L.win[i - s_lo] = ld128(&beg[i]);
				rs_fence_wave();
				bool changed = false;
				if (mine_in && cnt[k] > 1 && cnt[k] <= 64) {
					u128 *b0 = L.win + (off[k] - s_lo), *b1 = b0 + cnt[k];
					for (u128 *q = b0 + 1; q < b1; ++q) if (q->x < (q - 1)->x) { changed = true; break; }
					if (changed && cnt[k] <= 12) rs_insertion(b0, b1);        // a handful of records: one lane each, all buckets at once
				}
				// larger buckets one after the other, the whole wave on each: the insertion sort of ksort.h:101-112 is a STABLE sort, i.e. record i
				// lands at rank #{j: x_j < x_i or (x_j == x_i and j < i)} -- m broadcast reads per lane instead of ~m*m/4 moves of 16 bytes by one
				unsigned long long coop = __ballot(changed && cnt[k] > 12);
				while (coop) {
					const int src = __ffsll((long long)coop) - 1;
					coop &= coop - 1;
					const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)(off[k] - s_lo), src), m = (uint32_t)__builtin_amdgcn_readlane((int)cnt[k], src);
					u128 mine; mine.x = mine.y = 0;
					if ((uint32_t)lane < m) mine = L.win[b + (uint32_t)lane];
					uint32_t rank = 0;
					for (uint32_t j = 0; j < m; ++j) { const uint64_t xj = L.win[b + j].x; rank += (xj < mine.x) | ((xj == mine.x) & (j < (uint32_t)lane)); }
					rs_fence_wave();
					if ((uint32_t)lane < m) L.win[b + rank] = mine;
					rs_fence_wave();
				}
				rs_fence_wave();
				if (__ballot(changed)) {
					for (uint32_t i = s_lo + (uint32_t)lane; i < s_hi; i += 64) beg[i] = L.win[i - s_lo];
					rs_fence_wg();
				}
			}
			lane_lo += span;
		}
	}
	(void)n;
}

// Runs of records that agree on x >> hi_shift, found 64 records at a time: runs of <= 64 records are insertion-sorted
// here (in LDS, one lane per run; ksort.h:142), longer ones are handed to big(offset, length).
template <class Big>
__device__ inline void rs_runs_wave(u128 *beg, int64_t n, int hi_shift, RsLds &L, int lane, Big big)
{
	int64_t rb = 0;
	while (rb < n) {
		// a window of 64 records starting at rb: which of them start a new run?
		const int64_t pos = rb + lane;
		const bool in = pos < n;
		u128 rec; rec.x = ~0ULL, rec.y = 0;
		if (in) rec = ld128(&beg[pos]);
		const uint64_t hk = in ? (hi_shift >= 64 ? 0ULL : rec.x >> hi_shift) : ~0ULL;
		const uint64_t hi0 = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(hk >> 32), 0) << 32) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)hk, 0);
		const uint32_t hp_lo = (uint32_t)__shfl((int)(uint32_t)(hk & 0xffffffffULL), lane > 0 ? lane - 1 : 0), hp_hi = (uint32_t)__shfl((int)(uint32_t)(hk >> 32), lane > 0 ? lane - 1 : 0);
		const uint64_t hp = lane == 0 ? hi0 : ((uint64_t)hp_hi << 32 | (uint64_t)hp_lo);
		const bool start = in && lane > 0 && hk != hp;
		const unsigned long long sm = __ballot(start);
		const int n_in = (int)(n - rb < 64 ? n - rb : 64);
		const bool last_complete = (rb + n_in >= n);
		const unsigned long long starts = sm | 1ULL;                               // bit 0: the run at rb
		const int n_runs = __popcll(starts);
		const int n_sort = last_complete ? n_runs : n_runs - 1;
		if (n_sort == 0) {
			// a single run that continues beyond the window: find its end
			int64_t re = rb + n_in;
			for (;;) {
				const int64_t p2 = re + lane;
				const bool brk = p2 >= n || (hi_shift >= 64 ? 0ULL : beg[p2 < n ? p2 : n - 1].x >> hi_shift) != hi0;
				const unsigned long long bm = __ballot(brk);
				if (bm) { re += __ffsll((long long)bm) - 1; break; }
				re += 64;
			}
			if (re - rb > 64) big(rb, re - rb);
			else rs_small_wave(beg, rb, re, L, lane);
			rb = re;
			continue;
		}
		// complete runs inside the window: run r spans [s_r, s_{r+1}); the last run of the window is complete only
		// if the window reaches the end of the array, otherwise it restarts the next window.  Lane r sorts run r.
		int my_b = -1, my_e = -1; int64_t next_rb = rb + n_in;
		{
			unsigned long long mrest = starts; int r = 0, prev = -1;
			while (mrest) {
				const int bpos = __ffsll((long long)mrest) - 1;
				mrest &= mrest - 1;
				if (prev >= 0 && r - 1 == lane) my_b = prev, my_e = bpos;
				prev = bpos; ++r;
			}
			if (last_complete) { if (n_runs - 1 == lane) my_b = prev, my_e = n_in; }
			else next_rb = rb + prev;
		}
		// only windows holding an out-of-order run are rewritten
		const uint32_t xlo = (uint32_t)__shfl((int)(uint32_t)rec.x, lane > 0 ? lane - 1 : 0), xhi = (uint32_t)__shfl((int)(uint32_t)(rec.x >> 32), lane > 0 ? lane - 1 : 0);
		const uint64_t xprev = (uint64_t)xhi << 32 | xlo;
		const bool desc = in && lane > 0 && !start && rec.x < xprev && pos < next_rb;   // inside a complete run, smaller than its left neighbour
		if (__ballot(desc)) {
			L.ins[lane] = rec;
			rs_fence_wave();
			if (my_b >= 0 && my_e - my_b > 1) rs_insertion(L.ins + my_b, L.ins + my_e);
			rs_fence_wave();
			if (pos < next_rb) beg[pos] = L.ins[lane];
			rs_fence_wg();
		}
		rb = next_rb;
	}
}

// bits that differ somewhere in [beg, beg+n)
__device__ inline uint64_t rs_varying_bits(const u128 *beg, int64_t n, int lane)
{
	uint64_t o = 0, a = ~0ULL;
	for (int64_t i0 = 0; i0 < n; i0 += 256) {                   // (unconditional loads, clamped: four in flight)
		uint64_t x[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) { const int64_t i = i0 + lane + 64 * k; x[k] = beg[i < n ? i : n - 1].x; }
#pragma unroll
		for (int k = 0; k < 4; ++k) { o |= x[k]; a &= x[k]; }
	}
	uint32_t olo = (uint32_t)o, ohi = (uint32_t)(o >> 32), alo = (uint32_t)a, ahi = (uint32_t)(a >> 32);
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		olo |= (uint32_t)__shfl_xor((int)olo, d); ohi |= (uint32_t)__shfl_xor((int)ohi, d);
		alo &= (uint32_t)__shfl_xor((int)alo, d); ahi &= (uint32_t)__shfl_xor((int)ahi, d);
	}
	return ((uint64_t)ohi << 32 | olo) ^ ((uint64_t)ahi << 32 | alo);
}

} // namespace pga
