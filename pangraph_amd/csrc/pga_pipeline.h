// pga_pipeline.h -- stage interfaces of the batch pipeline (host orchestration in pga_api.cpp).
#pragma once
#include "pga_common.h"
#include <memory>
#include <mutex>

namespace pga {

// One alignment record while it is being built (mm_reg1_t + mm_extra_t, minimap.h:94-119).
struct Reg {
	int32_t id = 0, cnt = 0, rid = 0, score = 0, qs = 0, qe = 0, rs = 0, re = 0, parent = 0, subsc = 0, as = 0;
	int32_t mlen = 0, blen = 0, n_sub = 0, score0 = 0;
	uint32_t mapq = 0, split = 0, rev = 0, inv = 0, split_inv = 0, hash = 0;
	bool has_p = false;
	int32_t dp_score = 0, dp_max = 0, dp_max2 = 0; uint32_t n_ambi = 0;
	std::vector<uint32_t> cigar;
};

struct ChainResult {
	std::vector<int32_t> n_u, n_v;   // per query
	PinVec<uint64_t> u;              // chain i of query q at q_aoff[q]+i: score<<32|cnt
	PinVec<u128> a;                  // compacted anchors of query q at q_aoff[q] .. +n_v[q] (only when the host asked for them: want_host_anchors)
	DBuf<u128> d_a;                  // the same on the device: what the alignment stage plans from (pga_plan.hip)
	bool want_host_anchors = true;
};

struct SeedResult {
	DBuf<u128> a; DBuf<uint64_t> q_aoff;
	std::vector<uint64_t> h_q_aoff; std::vector<int32_t> h_rep_len;
	uint64_t n_a = 0;
	// Tie order on demand.  `a` holds every query's anchors STABLY sorted by x.  minimap2's sort (radix_sort_128x, map.c:202) is unstable: where a
	// query holds equal keys (q_tie[q] != 0) the reference's arrangement of those anchors is the outcome of its in-place walk.  The chaining stage
	// proves for almost every query that its result does not depend on that arrangement (pga_chain.hip: chain_all) and asks for the exact order of
	// the others only (seed_exact_order); until then the raw (generation) order and the stable sort stay here as the replay's input and hint.
	bool exact = true;                          // false: tied queries are still in stable order
	DBuf<uint64_t> raw_x, raw_y, srt_x, srt_y;
	DBuf<uint32_t> dupc, q_tie;
};

void build_index_ex(const SeqSet &S, const Minimizers &M, int w, int k, Index &I, DBuf<uint32_t> &grp_of_mz, hipStream_t st);
void seed_all(const SeqSet &S, const Minimizers &M, const Index &I, const DBuf<uint32_t> &grp_of_mz, const mm_mapopt_t &opt,
              const DBuf<int32_t> &d_name_rank, const DBuf<int32_t> &d_mid_occ, SeedResult &O, hipStream_t st, Timers *tm = nullptr,
              const uint8_t *d_own = nullptr, bool exact_order = true);   // d_own[q] == 0: query q is mapped by another shard (its minimizers stay in the index)
// the reference's arrangement of the anchors of the queries with d_need[q] != 0 (all tied queries if null), in place in O.a; the others keep theirs
void seed_exact_order(SeedResult &O, const uint32_t *d_need, int n_seq, hipStream_t st, Timers *tm = nullptr);
// exact_order = false: SR may hold tied queries in stable order (SR.exact == false); queries whose chains could depend on the reference's tie
// order are detected, re-sorted exactly (seed_exact_order) and chained again
void chain_all(const SeqSet &S, SeedResult &SR, const mm_mapopt_t &opt, int k, ChainResult &O, hipStream_t st, Timers *tm = nullptr, bool exact_order = true);
bool exact_sorts_forced();   // PGA_EXACT_SORTS=1: every unstable sort is replayed whether or not the result needs it (round-3 behaviour; the stage taps)
void align_batch(const SeqSet &S, const mm_mapopt_t &opt, int k, const std::vector<uint64_t> &q_aoff, ChainResult &C, const std::vector<int32_t> &rep_len,
                 std::vector<std::vector<Reg>> &out, int n_threads, Timers *tm, hipStream_t st);

// exact replay of minimap2's unstable radix_sort_128x on the flagged arrays [off[s], off[s+1]) of a (pga_sort_replay.hip)
// Optional hint: the same records STABLY sorted (sx, sy: keys and payloads at the same global positions) and dupc[i] = number of
// positions j <= i with sx[j] == sx[j-1].  A bucket of the replay occupies the rank range of its records; if that range holds no two
// equal keys, its final content is the sorted range itself, whatever the walk would have done: it is copied and never queued.
struct RsHint { const uint64_t *sx, *sy; const uint32_t *dupc; };
void replay_sort_segments(u128 *a, uint64_t n_total, const uint64_t *d_off, const int64_t *d_len, int n_seg, const uint32_t *d_flag, hipStream_t st, Timers *tm = nullptr, const RsHint *hint = nullptr);

} // namespace pga
