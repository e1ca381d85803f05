// pga_align.cpp -- the base-level alignment DRIVER of a batch: control flow on the host, every byte of data work on the device.
//
// Replaces (reference: packages/minimap2-sys/minimap2/) hit.c's region bookkeeping (mm_gen_regs, mm_split_reg, mm_filter_regs,
// mm_hit_sort, mm_set_mapq: hit.c:8-88,106-123,188-218,290-329,396-466) and align.c's mm_align_skeleton / mm_align1 /
// mm_align1_inv with their helpers (align.c:355-509,575-1022).  The reference aligns one region after another and calls the DP
// kernel, the z-drop test and the CIGAR post-processing synchronously, reading bases as it goes.  Here
//   * every region of every query of the batch is a small STATE MACHINE over compact records (chain anchors, DP results):
//       plan     what mm_align1 decides before its first DP call: trimmed ends, ignored seeds, DP windows, gap-fill segments
//       advance  consume results in the reference's order: left extension, gap fills with the z-drop test (a second exact pass is
//                requested when it fires), right extension, split, inversion
//   * the host never reads a base.  Everything that does is a batched device request (pga_post.hip, pga_ksw*.hip):
//       identity probes   equally long gap-fill windows: answered "nM" when the main diagonal is provably optimal, else a DP problem
//       DP problems       all problems requested in a round run as a handful of persistent launches (dp_run)
//       z-drop walks      the walk of mm_test_zdrop over a first-pass CIGAR, only when the score cannot rule a drop out
//       CIGAR finishes    mm_fix_cigar + mm_update_extra of a completed region, one wave per region
//     a ROUND is: DP launch -> (advance -> walks / finishes -> advance ...) until every query either finished or waits for DP.
// Region records live in `Reg`; chain anchors stay in the order the chaining stage left them.
#include "pga_common.h"
#include "pga_dp.h"
#include "pga_post.h"
#include "pga_plan.h"
#include "pga_sort_exact.h"
#include "pga_pipeline.h"
#include <cmath>
#include <thread>
#include <atomic>
#include <list>
#include <condition_variable>
#include <deque>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <numeric>

namespace pga {

static inline double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// anchor flag bits (mmpriv.h:18-24) and DP flags (ksw2.h:11-20)
static const uint64_t A_LONG_JOIN = 1ULL << 40, A_IGNORE = 1ULL << 41, A_TANDEM = 1ULL << 42, A_SELF = 1ULL << 43;
static const int DP_NEG_INF = -0x40000000, DP_RIGHT = 0x02, DP_APPROX_MAX = 0x08, DP_EXTZ_ONLY = 0x40, DP_REV_CIGAR = 0x80;

// ---- anchors of one query: x = strand<<63 | target<<32 | target position, y = flags | span<<32 | query position (lchain.c:140-147) ----
struct Anchors {
	u128 *a; int32_t n;
	int32_t tpos(int i) const { return (int32_t)a[i].x; }
	int32_t qpos(int i) const { return (int32_t)a[i].y; }
	int32_t span(int i) const { return (int32_t)(a[i].y >> 32 & 0xff); }
	uint64_t target_key(int i) const { return a[i].x >> 32; }        // strand + target id
	bool flagged(int i, uint64_t f) const { return (a[i].y & f) != 0; }
	void flag(int i, uint64_t f) { a[i].y |= f; }
	// query advance minus target advance between anchor i-1 and i: > 0 insertion, < 0 deletion
	int32_t indel(int i) const { return (qpos(i) - qpos(i - 1)) - (tpos(i) - tpos(i - 1)); }
};

static void sort_by_x(std::vector<u128> &v) { uint32_t head[256], tail[256]; if (!v.empty()) radix_sort_128x_exact(v.data(), v.data() + v.size(), head, tail); }

// ---------------------------------------------------------------- region records from chains
// Coordinates of a chain (hit.c:23-38) and its approximate match / block lengths (hit.c:8-21): one pass over the anchors.
static void chain_extent(Reg &r, int32_t qlen, const Anchors &A)
{
	const int first = r.as, last = r.as + r.cnt - 1;
	const int32_t sp0 = A.span(first);
	r.rev = (uint32_t)(A.a[first].x >> 63);
	r.rid = (int32_t)(A.a[first].x << 1 >> 33);
	r.rs = std::max(0, A.tpos(first) + 1 - sp0);
	r.re = A.tpos(last) + 1;
	const int32_t q_lo = A.qpos(first) + 1 - sp0, q_hi = A.qpos(last) + 1;       // on the aligned strand
	if (r.rev) r.qs = qlen - q_hi, r.qe = qlen - q_lo; else r.qs = q_lo, r.qe = q_hi;
	int32_t covered = 0, block = 0;
	if (r.cnt > 0) {
		covered = block = sp0;
		for (int i = first + 1; i <= last; ++i) {
			const int32_t dt = A.tpos(i) - A.tpos(i - 1), dq = A.qpos(i) - A.qpos(i - 1), sp = A.span(i);
			block += std::max(dt, dq);
			covered += (dt > sp && dq > sp) ? sp : std::min(dt, dq);
		}
	}
	r.mlen = covered, r.blen = block;
}

static inline uint64_t mix64(uint64_t k) // the 64-bit finalizer hit.c:40-50 salts region keys with
{
	k = ~k + (k << 21); k ^= k >> 24;
	k = k + (k << 3) + (k << 8); k ^= k >> 14;
	k = k + (k << 2) + (k << 4); k ^= k >> 28;
	k += k << 31;
	return k;
}
static inline uint32_t name_hash31(const std::string &s) { uint32_t h = 0; bool first = true; for (unsigned char c : s) { h = first ? c : h * 31u + c; first = false; } return h; }   // X31 string hash (khash.h)
static inline uint32_t mix32(uint32_t k) { k += ~(k << 15); k ^= k >> 10; k += k << 3; k ^= k >> 6; k += ~(k << 11); k ^= k >> 16; return k; }   // Wang's 32-bit mix (map.c:246-248)

// One region per chain, ordered by descending (score<<32 | cnt) ^ salt(first anchor, query) -- hit.c:52-88.  The order of equal keys
// is the one minimap2's radix sort leaves, so the keys go through its exact replay.
static void regions_from_chains(uint32_t query_salt, int qlen, int n_chains, const uint64_t *u, const Anchors &A, std::vector<Reg> &regs, const u128 *heads = nullptr)
{
	// heads: the first anchor of every chain, gathered on the device (then A holds no anchors and the extents are left to the planner)
	regs.clear();
	if (n_chains == 0) return;
	std::vector<u128> key((size_t)n_chains);
	int32_t start = 0;
	for (int c = 0; c < n_chains; ++c) {
		const u128 h0 = heads ? heads[c] : A.a[start];
		const uint32_t salt = (uint32_t)mix64((mix64(h0.x) + mix64(h0.y)) ^ query_salt);
		const int32_t cnt = (int32_t)u[c];
		key[(size_t)c].x = u[c] ^ salt;
		key[(size_t)c].y = (uint64_t)start << 32 | (uint32_t)cnt;
		start += cnt;
	}
	sort_by_x(key);
	regs.resize((size_t)n_chains);
	for (int c = 0; c < n_chains; ++c) {
		const u128 &k = key[(size_t)(n_chains - 1 - c)];          // descending
		Reg &r = regs[(size_t)c] = Reg();
		r.id = c, r.parent = -1;                                   // -X: no primary/secondary selection, parents stay unset
		r.score = r.score0 = (int32_t)(k.x >> 32), r.hash = (uint32_t)k.x;
		r.cnt = (int32_t)(uint32_t)k.y, r.as = (int32_t)(k.y >> 32);
		if (!heads) chain_extent(r, qlen, A);
	}
}

// The tail of `head` from its anchor `n_keep` on becomes its own region (hit.c:106-123); scores are shared out by anchor counts.
static void cut_region(Reg &head, Reg &tail, int n_keep, int qlen, const Anchors &A, bool extents = true)
{
	if (n_keep <= 0 || n_keep >= head.cnt) return;
	const int total = head.cnt;
	tail = head;
	tail.id = -1, tail.split_inv = 0;
	tail.has_p = false, tail.cigar.clear(), tail.dp_score = tail.dp_max = tail.dp_max2 = 0, tail.n_ambi = 0;
	tail.as = head.as + n_keep, tail.cnt = total - n_keep;
	tail.score = (int32_t)(head.score * ((float)tail.cnt / total) + .499);
	if (head.parent == head.id) tail.parent = -2;                 // MM_PARENT_TMP_PRI
	head.cnt = n_keep, head.score -= tail.score;
	// (device-side planning: the tail's extent comes back with its plan; the head's is overwritten by what its alignment found)
	if (extents) { chain_extent(tail, qlen, A); chain_extent(head, qlen, A); }
	head.split |= 1, tail.split |= 2;
}

static bool region_survives(const mm_mapopt_t &opt, int qlen, const Reg &r) // hit.c:290-309
{
	if (!r.inv && r.cnt < opt.min_cnt) return false;
	if (!r.has_p) return true;
	if (r.mlen < opt.min_chain_score || r.dp_max < opt.min_dp_max) return false;
	return !(r.qs > qlen * opt.max_clip_ratio && qlen - r.qe > qlen * opt.max_clip_ratio);
}
static void drop_weak_regions(const mm_mapopt_t &opt, int qlen, std::vector<Reg> &regs)
{
	regs.erase(std::remove_if(regs.begin(), regs.end(), [&](const Reg &r) { return !region_survives(opt, qlen, r); }), regs.end());
}

// descending DP score (chain score without a CIGAR), ties by the salted hash, equal keys as minimap2's sort leaves them (hit.c:188-218)
static void order_regions(std::vector<Reg> &regs)
{
	if (regs.size() <= 1) return;
	std::vector<u128> key; key.reserve(regs.size());
	for (size_t i = 0; i < regs.size(); ++i) {
		const Reg &r = regs[i];
		if (!r.inv && r.cnt <= 0) continue;
		key.push_back(u128{(uint64_t)(r.has_p ? r.dp_max : r.score) << 32 | r.hash, (uint64_t)i});
	}
	sort_by_x(key);
	std::vector<Reg> out; out.reserve(key.size());
	for (auto it = key.rbegin(); it != key.rend(); ++it) out.push_back(std::move(regs[it->y]));
	regs.swap(out);
}

// mapping quality of one primary region (hit.c:421-466, long reads); float arithmetic in the reference's order
static uint32_t region_mapq(const Reg &r, float uniq_ratio, int min_chain_sc, int match_sc)
{
	const float coef = 40.0f;
	const float by_score = (r.score > 100 ? 1.0f : 0.01f * r.score) * uniq_ratio, by_cnt = r.cnt > 10 ? 1.0f : 0.1f * r.cnt;
	const float pen = by_score < by_cnt ? by_score : by_cnt;
	const int subsc = std::max(r.subsc, min_chain_sc);
	int mapq;
	if (r.has_p && r.dp_max2 > 0 && r.dp_max > 0) {
		const float identity = (float)r.mlen / r.blen;
		const float x = (float)r.dp_max2 * subsc / r.dp_max / r.score0;
		mapq = (int)(identity * pen * coef * (1.0f - x * x) * logf((float)r.dp_max / match_sc));
		const int alt = (int)(6.02f * identity * identity * (r.dp_max - r.dp_max2) / match_sc + .499f);
		mapq = std::min(mapq, alt);
	} else {
		const float x = (float)subsc / r.score0;
		if (r.has_p) { const float identity = (float)r.mlen / r.blen; mapq = (int)(identity * pen * coef * (1.0f - x) * logf((float)r.dp_max / match_sc)); }
		else mapq = (int)(pen * coef * (1.0f - x) * logf(r.score));
	}
	mapq -= (int)(4.343f * logf(r.n_sub + 1) + .499f);
	uint32_t q = (uint32_t)std::min(60, std::max(0, mapq));
	if (r.has_p && r.dp_max > r.dp_max2 && q == 0) q = 1;
	return q;
}
static void assign_mapq(std::vector<Reg> &regs, int min_chain_sc, int match_sc, int rep_len)
{
	if (regs.empty()) return;
	int64_t primary_sum = 0;
	for (const Reg &r : regs) if (r.parent == r.id) primary_sum += r.score;
	const float uniq_ratio = (float)primary_sum / (primary_sum + rep_len);
	for (Reg &r : regs) r.mapq = (!r.inv && r.parent == r.id) ? region_mapq(r, uniq_ratio, min_chain_sc, match_sc) : 0;
	// an inversion inherits the weaker of its two flanks (hit.c:396-419)
	if (regs.size() < 3 || std::none_of(regs.begin(), regs.end(), [](const Reg &r) { return r.inv != 0; })) return;
	std::vector<u128> by_pos;
	for (int i = 0; i < (int)regs.size(); ++i) if (regs[(size_t)i].parent == i || regs[(size_t)i].parent < 0) by_pos.push_back(u128{(uint64_t)regs[(size_t)i].rid << 32 | (uint32_t)regs[(size_t)i].rs, (uint64_t)i});
	sort_by_x(by_pos);
	for (size_t i = 1; i + 1 < by_pos.size(); ++i) {
		Reg &mid = regs[by_pos[i].y];
		if (mid.inv) mid.mapq = std::min(regs[by_pos[i - 1].y].mapq, regs[by_pos[i + 1].y].mapq);
	}
}

static inline float log2_approx(float x) // mmpriv.h:118-126 (valid for x >= 2)
{
	union { float f; uint32_t i; } z = { x };
	float r = (float)(((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	r += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return r;
}

// Re-scale dp_max of all regions of a query by the divergence of its best one (align.c:897-960).  The operation lists supply the
// gap lengths in order (the reference's double accumulation is order-bound).
static void rescale_dp_max(int qlen, std::vector<Reg> &regs, float frac, int a, int b)
{
	if (regs.size() < 2) return;
	int best = -1, top = -1, second = -1;
	for (int i = 0; i < (int)regs.size(); ++i) {
		const Reg &r = regs[(size_t)i];
		if (!r.has_p) continue;
		if (r.dp_max > top) second = top, top = r.dp_max, best = i;
		else if (r.dp_max > second) second = r.dp_max;
	}
	if (best < 0 || top < 0 || second < 0) return;
	const Reg &lead = regs[(size_t)best];
	if (lead.qe - lead.qs < (double)qlen * frac || second < (double)top * frac) return;
	int32_t n_open = 0, n_base = 0;
	for (uint32_t c : lead.cigar) { const uint32_t op = c & 0xf; if (op == 1 || op == 2) ++n_open, n_base += (int32_t)(c >> 4); }
	const double identity = (double)lead.mlen / (lead.blen + (int32_t)lead.n_ambi - n_base + n_open);     // mm_event_identity
	double div = 1. - identity;
	if (div < 0.02) div = 0.02;
	double b2 = 0.5 / div;
	if (b2 * a < b) b2 = (double)a / b;
	for (Reg &r : regs) {
		if (!r.has_p) continue;
		double gap_cost = 0.0; int32_t gap_bases = 0;
		for (uint32_t c : r.cigar) { const uint32_t op = c & 0xf; if (op == 1 || op == 2) { gap_cost += b2 + (double)log2_approx((float)(1.0 + (c >> 4))); gap_bases += (int32_t)(c >> 4); } }
		const int32_t n_mis = r.blen + (int32_t)r.n_ambi - r.mlen - gap_bases;
		r.dp_max = std::max(0, (int32_t)(a * (r.mlen - b2 * n_mis - gap_cost) + .499));
	}
}

// ---------------------------------------------------------------- seeds of a chain that the DP must not trust (align.c:373-509)
struct LongGaps { std::vector<int> at; };     // chain-relative indices i whose |indel(i)| exceeds a threshold; only used when there are >= 2

static bool long_gaps_of(const Anchors &A, int as1, int cnt1, int min_gap, LongGaps &G)
{
	G.at.clear();
	for (int i = 1; i < cnt1; ++i) { const int32_t g = A.indel(as1 + i); if (g < -min_gap || g > min_gap) G.at.push_back(i); }
	return G.at.size() > 1;
}

// Bursts of compensating indels (an insertion soon undone by a deletion): the seeds inside the heaviest burst are ignored (align.c:392-431).
static void ignore_indel_bursts(Anchors &A, int as1, int cnt1, int min_gap, int weight_min, int reach_bases, int reach_gaps)
{
	LongGaps G;
	if (!long_gaps_of(A, as1, cnt1, min_gap, G)) return;
	const int n = (int)G.at.size();
	struct Burst { int from = -1, to = -1, weight = 0; } cur;
	for (int k = 0;; ++k) {
		if (k == n || k >= cur.to) {
			if (cur.to > 0) for (int i = G.at[(size_t)cur.from]; i < G.at[(size_t)cur.to]; ++i) A.flag(as1 + i, A_IGNORE);
			cur = Burst();
			if (k == n) break;
		}
		// heaviest burst that starts at long gap k: weight = twice the smaller of inserted / deleted bases
		const int i0 = as1 + G.at[(size_t)k];
		int ins = 0, del = 0, best_w = 0, best_l = -1;
		auto add = [&](int g) { if (g > 0) ins += g; else del -= g; };
		add(A.indel(i0));
		const int32_t q0 = A.qpos(i0 - 1), t0 = A.tpos(i0 - 1);
		for (int l = k + 1; l < n && l <= k + reach_gaps; ++l) {
			const int j = as1 + G.at[(size_t)l];
			if (A.qpos(j) - q0 > reach_bases || A.tpos(j) - t0 > reach_bases) break;
			add(A.indel(j));
			const int w = ins + del - std::abs(ins - del);
			if (w > best_w) best_w = w, best_l = l;
		}
		if (best_w > weight_min && best_w > cur.weight) cur.from = k, cur.to = best_l, cur.weight = best_w;
	}
}

// Runs of long gaps packed closer than the gaps are long: everything between is ignored and the run's last seed is marked as the far
// side of one long gap (align.c:433-469).
static void join_crowded_gaps(Anchors &A, int as1, int cnt1, int min_gap, int reach)
{
	LongGaps G;
	if (!long_gaps_of(A, as1, cnt1, min_gap, G)) return;
	const int n = (int)G.at.size();
	for (int k = 0; k < n;) {
		const int i = as1 + G.at[(size_t)k];
		int l = k + 1;
		int32_t t_end = A.tpos(i), q_end = A.qpos(i), g_prev = std::abs(A.indel(i));
		for (; l < n; ++l) {
			const int j = as1 + G.at[(size_t)l];
			if (A.qpos(j) - q_end > reach || A.tpos(j) - t_end > reach) break;
			const int32_t g = std::abs(A.indel(j)), sp = A.span(j - 1);
			const int32_t room = std::min(A.tpos(j - 1) + sp - t_end, A.qpos(j - 1) + sp - q_end);
			if (room > g_prev + g) break;
			t_end = A.tpos(j), q_end = A.qpos(j), g_prev = g;
		}
		if (l > k + 1) {
			const int last = G.at[(size_t)(l - 1)];
			for (int j = G.at[(size_t)k]; j < last; ++j) A.flag(as1 + j, A_IGNORE);
			A.flag(as1 + last, A_LONG_JOIN);
		}
		k = l;
	}
}

// Seeds at either end of a chain that sit off the diagonal of what follows are cut off (align.c:471-509).
static void trim_chain_ends(const Reg &r, const Anchors &A, int bw, int min_match, int32_t &as1, int32_t &cnt1)
{
	as1 = r.as, cnt1 = r.cnt;
	if (r.cnt < 3) return;
	const int last = r.as + r.cnt - 1;
	auto settled = [&](int32_t len, int32_t match) { return len >= bw << 1 || (match >= min_match && match >= bw) || match >= r.mlen >> 1; };
	int32_t len, match;
	len = match = A.span(r.as);
	for (int i = r.as + 1; i < last; ++i) {
		if (A.flagged(i, A_LONG_JOIN)) break;
		const int32_t dt = A.tpos(i) - A.tpos(i - 1), dq = A.qpos(i) - A.qpos(i - 1), lo = std::min(dt, dq), hi = std::max(dt, dq);
		if (hi - lo > len >> 1) as1 = i;
		len += lo, match += std::min(lo, A.span(i));
		if (settled(len, match)) break;
	}
	cnt1 = last + 1 - as1;
	len = match = A.span(last);
	for (int i = last - 1; i > as1; --i) {
		if (A.flagged(i + 1, A_LONG_JOIN)) break;
		const int32_t dt = A.tpos(i + 1) - A.tpos(i), dq = A.qpos(i + 1) - A.qpos(i), lo = std::min(dt, dq), hi = std::max(dt, dq);
		if (hi - lo > len >> 1) cnt1 = i + 1 - as1;
		len += lo, match += std::min(lo, A.span(i + 1));
		if (settled(len, match)) break;
	}
}

static inline bool ll_on_device(const mm_mapopt_t &opt, int q_len, int t_len)
{
	const int q8 = (q_len + 7) / 8 * 8;
	return q_len > 0 && t_len > 0 && q8 <= PGA_LL_MAX_LEN && t_len <= PGA_LL_MAX_LEN && (int64_t)std::abs(opt.a) * q8 < 32000;
}

// A sufficient condition for mm_test_zdrop (align.c:47-89) to return 0 that needs no sequence: every z it tracks is at
// most max - score, i.e. at most the sum of all score decrements along the path.  The global pass's score fixes how
// much the match columns fall short of all-matches (a*L - score - gap costs; the dual-affine gap costs are bounded
// from below by their cheapest form), mm_test_zdrop itself charges q + e*len per gap.  If even that total cannot exceed
// zdrop (nor zdrop_inv, which gates the inversion test), the answer is 0.
static bool zdrop_impossible(const mm_mapopt_t &opt, const DpRes &ez, const uint32_t *cigar)
{
	if (ez.zdropped || ez.n_cigar <= 0 || ez.score <= -0x3fffffff) return false;
	int64_t L = 0, g_dp = 0, g_test = 0;
	for (int k = 0; k < ez.n_cigar; ++k) {
		const int64_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) L += len;
		else if (op == 1 || op == 2 || op == 3) {
			const int64_t c1 = opt.q + (int64_t)opt.e * len, c2 = opt.q2 + (int64_t)opt.e2 * len;
			g_dp += c1 < c2 ? c1 : c2;
			g_test += c1;
		} else return false;
	}
	const int64_t shortfall = (int64_t)opt.a * L - (int64_t)ez.score - g_dp;
	if (shortfall < 0) return false;                 // not a plain match/mismatch matrix: leave it to the full test
	const int64_t lim = opt.zdrop < opt.zdrop_inv ? opt.zdrop : opt.zdrop_inv;
	return shortfall + g_test <= lim;
}

// CIGAR pieces are appended as the reference does it (align.c:291-314): equal operations at the seam are fused
static void append_ops(Reg &r, uint32_t n, const uint32_t *ops)
{
	if (n == 0) return;
	r.has_p = true;
	size_t from = 0;
	if (!r.cigar.empty() && ((r.cigar.back() ^ ops[0]) & 0xf) == 0) { r.cigar.back() += ops[0] >> 4 << 4; from = 1; }
	r.cigar.insert(r.cigar.end(), ops + from, ops + n);
}

// ---------------------------------------------------------------- the per-region state machine
struct Seg {
	int32_t i, rs, qs, re, qe, bw1;
	int job1 = -1, job2 = -1, zcode = -1, ll_job = -1;   // job1 == -2: a first pass that is an identity probe without a problem record (pm: its answer)
	int32_t pm = -1;                    // mismatches the probe counted (-1: not asked yet)
	int32_t i_prev = -1;                // device-side planning: the anchor the segment starts at (chain-relative), for the split after a z-drop
	int walk = 0;                       // z-drop walk: 0 not asked, 1 asked, 2 answered
	bool ll_deferred = false;           // the inversion query only decides split_inv of a split-off region: nobody waits for it here
	int32_t max_zdrop = 0, wt0 = -1, wt1 = -1, wq0 = -1, wq1 = -1;
};

struct RegTask {
	Reg r;
	bool planned = false, done = false, is_inv = false;
	int32_t rid = 0, rev = 0, as1 = 0, cnt1 = 0;
	int32_t rs = 0, qs = 0, re = 0, qe = 0, rs0 = 0, qs0 = 0, re0 = 0, qe0 = 0;
	int32_t bw = 0;
	int left_job = -1, right_job = -1;
	bool left_done = false;
	std::vector<Seg> segs; size_t seg_k = 0;
	int32_t rs1 = 0, qs1 = 0, re1 = 0, qe1 = 0;
	bool dropped = false;
	// CIGAR finish on the device: 0 not yet, 1 asked, 2 applied
	int fin = 0; uint64_t fin_t_off = 0; int32_t fin_q_start = 0, fin_q_rev = 0;
	// inversion test state (mm_align1_inv)
	int inv_state = 0;      // 0 = not evaluated, 1 = waiting for its DP problem, 2 = resolved, 3 = waiting for the local-alignment query
	int inv_ll_job = -1;
	int inv_job = -1; int32_t inv_q_off = 0, inv_t_off = 0, inv_ql = 0, inv_tl = 0;
	int split_inv_ll = -1;  // >= 0: r.split_inv is still to be read off this local-alignment query (mm_test_zdrop's return code 2, align.c:78-86,781)
};

struct WalkAsk { RegTask *T; size_t seg; };

struct QueryCtx {
	int qid = 0; int32_t qlen = 0; int base = 0;   // base: first sequence of the query's group (record rids are group-relative)
	u128 *a = nullptr; int32_t n_a = 0;   // the query's compacted anchors: a view into the chain stage's (pinned) download, flags are set in place
	std::vector<RegTask*> list;      // the reference's regs[] order, grown by insertions
	std::vector<std::unique_ptr<RegTask>> pool;
	int rep_len = 0;
	bool finished = false;
	// DP problems of this query (ids are per query; filled between the threaded phases)
	std::vector<DpJob> jobs; std::vector<DpRes> res; std::vector<const uint32_t*> cig; std::vector<int> pending;   // cig[id] points into a per-round CIGAR pool
	std::unique_ptr<std::deque<uint32_t>> own_cig;   // one-operation CIGARs of the gap fills answered by the identity probe (stable addresses; made on first use: an empty deque already owns 0.5 KB)
	std::vector<RegTask*> probe_tasks; // regions whose plan left segments with job1 == -2 that have not been probed yet
	uint64_t a_off = 0;               // device-side planning: index of the query's first compacted anchor in the device array
	std::vector<RegTask*> to_plan;    // ... regions split off by the last advance pass, waiting for their plan
	std::vector<WalkAsk> walks;       // device requests raised by the last advance pass
	std::vector<RegTask*> fins;
};

struct Driver {
	const SeqSet &S; const mm_mapopt_t &opt; int k; hipStream_t st;
	int8_t mat[25];
	int probe_m_max = -1;             // identity probes: (a+b)*m < a + 2*min(q+e, q2+e2)  <=>  m <= probe_m_max; -1: probes off
	int spec_len = getenv("PGA_SPEC_LEN") ? atoi(getenv("PGA_SPEC_LEN")) : 1500;   // segments at least this long get their second pass speculatively (0: never)
	Driver(const SeqSet &S_, const mm_mapopt_t &o, int k_, hipStream_t st_) : S(S_), opt(o), k(k_), st(st_)
	{
		const int a = std::abs(o.a), b = -std::abs(o.b), amb = -std::abs(o.sc_ambi);     // align.c:9-22
		for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) mat[i * 5 + j] = (int8_t)((i == 4 || j == 4) ? amb : i == j ? a : b);
		const int64_t lim = (int64_t)a + 2 * std::min(o.q + o.e, o.q2 + o.e2);
		if (a > 0 && b < 0) probe_m_max = (int)((lim - 1) / (a - b));
		// a probed segment needs no problem record at all when its answer can never reach the z-drop test: nM with m <= probe_m_max mismatches falls
		// (a + |b|) * m short of all-matches, and zdrop_impossible() says "no drop" iff that is within both thresholds
		lean_probes = probe_m_max >= 0 && (int64_t)(a - b) * probe_m_max <= std::min(o.zdrop, o.zdrop_inv) && !getenv("PGA_NO_LEAN_PROBES");
	}
	bool lean_probes = false;
	// device-side planning (pga_plan.hip): the anchors stay where the chaining stage left them
	u128 *d_anchors = nullptr; bool dev_plan = false;
	PlanParams plan_params() const
	{
		PlanParams P; memset(&P, 0, sizeof(P));
		P.k = k; P.bw = opt.bw; P.bw_long = std::max((int)(opt.bw_long * 1.5 + 1.), (int)(opt.bw * 1.5 + 1.)); P.max_gap = opt.max_gap; P.min_cnt = opt.min_cnt; P.min_chain_score = opt.min_chain_score;
		P.min_ksw_len = opt.min_ksw_len; P.a = opt.a; P.q = opt.q; P.e = opt.e; P.no_end_flt = (opt.flag & MM_F_NO_END_FLT) ? 1 : 0; P.probe_m_max = lean_probes ? probe_m_max : -1;
		P.max_sw_mat = opt.max_sw_mat;
		const int g_max_env = getenv("PGA_PLAN_G_MAX") ? atoi(getenv("PGA_PLAN_G_MAX")) : 4096;
		P.g_max = std::min(4096, std::max(0, g_max_env));      // PLAN_G_MAX of pga_plan.hip
		return P;
	}
	// what plan() leaves in a RegTask, from the planner's records (same order of requests: left extension, segments, right extension)
	void apply_plan(QueryCtx &Q, RegTask &T, const PlanOut &O, const PlanItem *items)
	{
		Reg &r = T.r;
		T.planned = true;
		if (O.status == 3 || r.cnt == 0) { T.done = true; return; }
		r.rev = (uint32_t)O.rev, r.rid = O.rid, r.rs = O.r_rs, r.re = O.r_re, r.qs = O.r_qs, r.qe = O.r_qe, r.mlen = O.r_mlen, r.blen = O.r_blen;
		T.rid = O.rid, T.rev = O.rev;
		int32_t bw = (int)(opt.bw * 1.5 + 1.);
		T.bw = bw;
		T.as1 = O.as1, T.cnt1 = O.cnt1, T.rs = O.rs, T.qs = O.qs, T.rs0 = O.rs0, T.qs0 = O.qs0, T.re0 = O.re0, T.qe0 = O.qe0;
		if (O.qs > 0 && O.rs > 0)
			T.left_job = request(Q, T.rev, T.rid, O.qs0, O.qs - O.qs0, O.rs0, O.rs - O.rs0, 1, bw, opt.end_bonus, r.split_inv ? opt.zdrop_inv : opt.zdrop, DP_EXTZ_ONLY | DP_RIGHT | DP_REV_CIGAR);
		else T.left_done = true, T.rs1 = O.rs, T.qs1 = O.qs;
		T.segs.reserve(O.n_items);
		for (uint32_t k2 = 0; k2 < O.n_items; ++k2) {
			const PlanItem &it = items[k2];
			Seg sg; sg.i = it.i, sg.rs = it.rs, sg.qs = it.qs, sg.re = it.re, sg.qe = it.qe, sg.i_prev = it.i_prev;
			if (it.kind == 0) { sg.bw1 = 0; sg.job1 = -2; sg.pm = it.m; }
			else {
				sg.bw1 = it.bw1;
				sg.job1 = request(Q, T.rev, T.rid, sg.qs, sg.qe - sg.qs, sg.rs, sg.re - sg.rs, 0, sg.bw1, -1, opt.zdrop, DP_APPROX_MAX, false);
				if (it.kind == 1 && spec_len > 0 && opt.zdrop == opt.zdrop_inv && std::max(sg.qe - sg.qs, sg.re - sg.rs) >= spec_len && !have(Q, sg.job1))
					sg.job2 = second_pass(Q, T, sg, opt.zdrop);
			}
			T.segs.push_back(sg);
		}
		T.re = O.T_re, T.qe = O.T_qe;
		if (T.qe < T.qe0 && T.re < T.re0)
			T.right_job = request(Q, T.rev, T.rid, T.qe, T.qe0 - T.qe, T.re, T.re0 - T.re, 0, bw, opt.end_bonus, opt.zdrop, DP_EXTZ_ONLY);
	}
	// target positions of the anchors [from, to) of a query (chain-relative to its first anchor), straight from the device (rare: a z-drop inside a segment)
	void fetch_tpos(const QueryCtx &Q, int from, int to, std::vector<int32_t> &tp) const
	{
		std::vector<u128> tmp((size_t)std::max(0, to - from));
		if (!tmp.empty()) PGA_HIP(hipMemcpy(tmp.data(), d_anchors + Q.a_off + (uint64_t)from, tmp.size() * sizeof(u128), hipMemcpyDeviceToHost));
		tp.resize(tmp.size());
		for (size_t i = 0; i < tmp.size(); ++i) tp[i] = (int32_t)tmp[i].x;
	}

	static bool have(const QueryCtx &Q, int id) { return id >= 0 && (size_t)id < Q.res.size() && Q.res[(size_t)id].pad == 1; }

	int request(QueryCtx &Q, int rev, int rid, int32_t qs, int32_t qlen, int32_t ts, int32_t tlen, int seq_rev, int w, int end_bonus, int zdrop, int flag, bool probe = false)
	{
		DpJob j; memset(&j, 0, sizeof(j));
		j.t_off = S.off[(size_t)(Q.base + rid)] + (uint64_t)ts; j.q_off = S.off[(size_t)Q.qid]; j.qlen_full = Q.qlen;
		j.qs = qs; j.qlen = qlen; j.tlen = tlen; j.w = w; j.zdrop = zdrop; j.end_bonus = end_bonus; j.flag = flag;
		j.q_rev = (uint8_t)rev; j.seq_rev = (uint8_t)seq_rev;
		const int id = (int)Q.jobs.size();
		DpRes r; memset(&r, 0, sizeof(r));
		// problems the reference never hands to the kernel (align.c:326-328, ksw2_extd2_sse.c:82) are resolved here
		r.max_q = r.max_t = r.mqe_t = r.mte_q = -1; r.score = r.mqe = r.mte = DP_NEG_INF;
		if (!(flag & PGA_JOB_LL) && opt.max_sw_mat > 0 && (int64_t)tlen * qlen > opt.max_sw_mat) { r.zdropped = 1; r.pad = 1; }
		else if (qlen <= 0 || tlen <= 0) r.pad = 1;
		// first-pass gap fill of two equally long windows (KSW_EZ_APPROX_MAX, band wide open): the identity probe answers "nM" when so few
		// positions differ that the main diagonal is provably the unique optimum (proof with the tile kernel, pga_ksw_fast.hip)
		if (!r.pad && probe && probe_m_max >= 0) j.pad[0] = 1;
		Q.jobs.push_back(j);
		Q.res.push_back(r); Q.cig.push_back(nullptr);
		if (!r.pad) Q.pending.push_back(id);
		return id;
	}

	// the probe of job `id` came back with m mismatches: the answer is nM with score a*(n-m) - b*m, no matrix
	void answer_probe(QueryCtx &Q, int id, int m)
	{
		const DpJob &j = Q.jobs[(size_t)id];
		DpRes &r = Q.res[(size_t)id];
		r.max = 0; r.max_q = r.max_t = r.mqe_t = r.mte_q = -1; r.mqe = r.mte = DP_NEG_INF;
		r.score = mat[0] * (j.qlen - m) + mat[1] * m; r.n_cigar = 1; r.zdropped = 0; r.pad = 1;
		if (!Q.own_cig) Q.own_cig.reset(new std::deque<uint32_t>());
		Q.own_cig->push_back((uint32_t)j.qlen << 4);
		Q.cig[(size_t)id] = &Q.own_cig->back();
	}

	// ---- plan (align.c:583-700): everything mm_align1 decides before its first DP call ----
	void plan(QueryCtx &Q, RegTask &T)
	{
		Reg &r = T.r; Anchors A{Q.a, Q.n_a}; const int32_t qlen = Q.qlen;
		T.planned = true;
		if (r.cnt == 0) { T.done = true; return; }
		T.rid = (int32_t)(A.a[r.as].x << 1 >> 33), T.rev = (int32_t)(A.a[r.as].x >> 63);
		const int32_t tlen_ref = (int32_t)S.len[(size_t)(Q.base + T.rid)];
		int32_t bw = (int)(opt.bw * 1.5 + 1.), bw_long = (int)(opt.bw_long * 1.5 + 1.);
		if (bw_long < bw) bw_long = bw;
		T.bw = bw;
		int32_t as1, cnt1, rs, qs, re, qe, rs0, qs0, re0, qe0, rs1, qs1, re1, qe1, i, l;
		if (!(opt.flag & MM_F_NO_END_FLT)) trim_chain_ends(r, A, opt.bw, opt.min_chain_score * 2, as1, cnt1);
		else as1 = r.as, cnt1 = r.cnt;
		ignore_indel_bursts(A, as1, cnt1, 10, 40, opt.max_gap >> 1, 10);
		join_crowded_gaps(A, as1, cnt1, 30, opt.max_gap >> 1);
		const int half_k = k >> 1;                                              // mm_adjust_minier, non-HPC: windows start at the middle of a seed
		rs = A.tpos(as1) - half_k, qs = A.qpos(as1) - half_k;
		re = A.tpos(as1 + cnt1 - 1) - half_k, qe = A.qpos(as1 + cnt1 - 1) - half_k;
		rs0 = std::max(0, A.tpos(r.as) + 1 - A.span(r.as));
		qs0 = A.qpos(r.as) + 1 - A.span(r.as);
		rs1 = qs1 = 0;
		for (i = r.as - 1, l = 0; i >= 0 && A.target_key(i) == A.target_key(r.as); --i) {
			const int32_t x = A.tpos(i) + 1 - A.span(i), y = A.qpos(i) + 1 - A.span(i);
			if (x < rs0 && y < qs0) {
				if (++l > opt.min_cnt) { l = std::max(rs0 - x, qs0 - y); rs1 = std::max(0, rs0 - l), qs1 = qs0 - l; break; }
			}
		}
		if (qs > 0 && rs > 0) {
			l = std::min(qs, opt.max_gap);
			qs1 = std::max(qs1, qs - l);
			qs0 = std::min(qs0, qs1);
			l += l * opt.a > opt.q ? (l * opt.a - opt.q) / opt.e : 0;
			l = std::min(std::min(l, opt.max_gap), rs);
			rs1 = std::max(rs1, rs - l);
			rs0 = std::min(std::min(rs0, rs1), rs);
		} else rs0 = rs, qs0 = qs;
		re0 = A.tpos(r.as + r.cnt - 1) + 1, qe0 = A.qpos(r.as + r.cnt - 1) + 1;
		re1 = tlen_ref, qe1 = qlen;
		for (i = r.as + r.cnt, l = 0; i < A.n && A.target_key(i) == A.target_key(r.as); ++i) {
			const int32_t x = A.tpos(i) + 1, y = A.qpos(i) + 1;
			if (x > re0 && y > qe0) {
				if (++l > opt.min_cnt) { l = std::max(x - re0, y - qe0); re1 = re0 + l, qe1 = qe0 + l; break; }
			}
		}
		if (qe < qlen && re < tlen_ref) {
			l = std::min(qlen - qe, opt.max_gap);
			qe1 = std::min(qe1, qe + l);
			qe0 = std::max(qe0, qe1);
			l += l * opt.a > opt.q ? (l * opt.a - opt.q) / opt.e : 0;
			l = std::min(std::min(l, opt.max_gap), tlen_ref - re);
			re1 = std::min(re1, re + l);
			re0 = std::max(re0, re1);
		} else re0 = re, qe0 = qe;
		if (A.flagged(r.as, A_SELF)) {
			int max_ext = std::abs(r.qs - r.rs);
			if (r.rs - rs0 > max_ext) rs0 = r.rs - max_ext;
			if (r.qs - qs0 > max_ext) qs0 = r.qs - max_ext;
			max_ext = std::abs(r.qe - r.re);
			if (re0 - r.re > max_ext) re0 = r.re + max_ext;
			if (qe0 - r.qe > max_ext) qe0 = r.qe + max_ext;
		}
		T.as1 = as1, T.cnt1 = cnt1, T.rs = rs, T.qs = qs, T.rs0 = rs0, T.qs0 = qs0, T.re0 = re0, T.qe0 = qe0;
		// left extension (align.c:702-722): reversed windows, right-aligned gaps, reversed CIGAR
		if (qs > 0 && rs > 0)
			T.left_job = request(Q, T.rev, T.rid, qs0, qs - qs0, rs0, rs - rs0, 1, bw, opt.end_bonus, r.split_inv ? opt.zdrop_inv : opt.zdrop, DP_EXTZ_ONLY | DP_RIGHT | DP_REV_CIGAR);
		else T.left_done = true, T.rs1 = rs, T.qs1 = qs;
		// gap-fill segments (align.c:726-745): determined by the anchors alone
		bool any_lean = false;
		for (i = 1; i < cnt1; ++i) {
			if (A.flagged(as1 + i, A_IGNORE | A_TANDEM) && i != cnt1 - 1) continue;
			re = A.tpos(as1 + i) - half_k, qe = A.qpos(as1 + i) - half_k;
			if (i == cnt1 - 1 || A.flagged(as1 + i, A_LONG_JOIN) || (qe - qs >= opt.min_ksw_len && re - rs >= opt.min_ksw_len)) {
				Seg sg; sg.i = i, sg.rs = rs, sg.qs = qs, sg.re = re, sg.qe = qe, sg.bw1 = bw_long;
				if (A.flagged(as1 + i, A_LONG_JOIN)) sg.bw1 = std::max(qe - qs, re - rs);
				const bool probe = qe - qs == re - rs && sg.bw1 >= qe - qs;
				// 99.9 % of the segments of a whole-genome chain are equally long windows that the identity probe answers "nM": they get no
				// problem record (48 + 56 bytes and three vector pushes each, a million per leaf batch) -- the probe reads the segment itself
				if (probe && lean_probes && qe - qs > 0 && !(opt.max_sw_mat > 0 && (int64_t)(qe - qs) * (re - rs) > opt.max_sw_mat)) { sg.job1 = -2; any_lean = true; }
				else
				sg.job1 = request(Q, T.rev, T.rid, qs, qe - qs, rs, re - rs, 0, sg.bw1, -1, opt.zdrop, DP_APPROX_MAX, probe);
				// a long segment without seeds is where chains break: its exact second pass (needed whenever the z-drop test fires,
				// with the same parameters whatever the test's return code when both thresholds agree) is launched WITH the first
				// pass instead of a round later -- one dependent round less per split, the extra problem runs on idle CUs
				if (!probe && spec_len > 0 && opt.zdrop == opt.zdrop_inv && std::max(qe - qs, re - rs) >= spec_len && !have(Q, sg.job1))
					sg.job2 = second_pass(Q, T, sg, opt.zdrop);
				T.segs.push_back(sg);
				rs = re, qs = qe;
			}
		}
		if (any_lean) Q.probe_tasks.push_back(&T);
		T.re = re, T.qe = qe;   // the last adjusted anchor (what the right extension starts from when nothing dropped)
		if (cnt1 == 1) T.re = A.tpos(as1) - half_k, T.qe = A.qpos(as1) - half_k;
		// right extension (align.c:789-805), speculative: only used when no segment z-drops
		if (T.qe < qe0 && T.re < re0)
			T.right_job = request(Q, T.rev, T.rid, T.qe, qe0 - T.qe, T.re, re0 - T.re, 0, bw, opt.end_bonus, opt.zdrop, DP_EXTZ_ONLY);
	}

	int second_pass(QueryCtx &Q, const RegTask &T, const Seg &sg, int zdrop) { return request(Q, T.rev, T.rid, sg.qs, sg.qe - sg.qs, sg.rs, sg.re - sg.rs, 0, sg.bw1, -1, zdrop, 0); }
	int zcode_of(const Seg &sg, int ll_score) const { return (ll_score >= opt.min_chain_score * opt.a && ll_score >= opt.min_dp_max) ? 2 : (sg.max_zdrop > opt.zdrop ? 1 : 0); }

	// A local alignment (ksw_ll_i16) over windows the device kernel does not hold.  Unreachable from pangraph's options -- both windows are
	// bounded by max_gap = 10 000 under every asm preset (align.c:81-82, 845-855; PGA_LL_MAX_LEN = 10 240) -- so it is refused, loudly,
	// instead of being computed somewhere else: this library has no host path for base work.
	int ll_on_host(QueryCtx &, int, int32_t, int q_len, int, int32_t, int t_len, bool, int *, int *)
	{
		throw std::runtime_error("pga: ksw_ll_i16 over windows of " + std::to_string(q_len) + " x " + std::to_string(t_len) + " bases: the device kernel holds " + std::to_string(PGA_LL_MAX_LEN) +
		                         " (max_gap above 10000 is outside pangraph's presets)");
	}

	void ask_finish(QueryCtx &Q, RegTask &T, int rid, int32_t t_start, int32_t q_start, int q_rev)
	{
		T.fin = 1; T.fin_t_off = S.off[(size_t)(Q.base + rid)] + (uint64_t)t_start; T.fin_q_start = q_start; T.fin_q_rev = q_rev;
		Q.fins.push_back(&T);
	}

	// ---- advance: returns true when the region is complete; r2 receives a split-off region (cnt>0), also on a `false` return ----
	bool advance(QueryCtx &Q, RegTask &T, Reg &r2, int &r2_split_inv_ll)
	{
		r2_split_inv_ll = -1;
		Reg &r = T.r; const Anchors A{Q.a, Q.n_a}; const int32_t qlen = Q.qlen;
		r2.cnt = 0;
		if (T.done) return true;
		if (T.fin == 1) return false;
		if (T.fin == 2) { T.done = true; return true; }
		if (!T.left_done) {
			if (!have(Q, T.left_job)) return false;
			const DpRes &ez = Q.res[(size_t)T.left_job];
			if (ez.n_cigar > 0) { append_ops(r, (uint32_t)ez.n_cigar, Q.cig[(size_t)T.left_job]); r.dp_score += ez.max; }
			T.rs1 = T.rs - (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
			T.qs1 = T.qs - (ez.reach_end ? T.qs - T.qs0 : ez.max_q + 1);
			T.left_done = true;
		}
		if (T.seg_k == 0) T.re1 = T.rs, T.qe1 = T.qs;
		while (T.seg_k < T.segs.size() && !T.dropped) {
			Seg &sg = T.segs[T.seg_k];
			if (sg.job1 == -2) {                                   // a probed segment: nM, no z-drop possible (Driver::lean_probes)
				if (sg.pm < 0) return false;
				T.re1 = sg.re, T.qe1 = sg.qe;
				const uint32_t op = (uint32_t)(sg.qe - sg.qs) << 4;
				append_ops(r, 1u, &op);
				r.dp_score += mat[0] * (sg.qe - sg.qs - sg.pm) + mat[1] * sg.pm;
				sg.zcode = 0;
				++T.seg_k;
				continue;
			}
			if (!have(Q, sg.job1)) return false;
			// results and CIGARs of the following segments lie wherever their kernels finished: start fetching them now
			for (size_t ahead = 2; ahead <= 4; ahead += 2) if (T.seg_k + ahead < T.segs.size()) {
				const int nj = T.segs[T.seg_k + ahead].job1;
				if (nj >= 0 && (size_t)nj < Q.res.size()) { __builtin_prefetch(&Q.res[(size_t)nj]); if (Q.cig[(size_t)nj]) __builtin_prefetch(Q.cig[(size_t)nj]); }
			}
			T.re1 = sg.re, T.qe1 = sg.qe;
			int final_job = sg.job1;
			if (sg.zcode < 0 && sg.ll_job < 0) {
				const DpRes &e1 = Q.res[(size_t)sg.job1];
				if (zdrop_impossible(opt, e1, Q.cig[(size_t)sg.job1])) sg.zcode = 0;
				else if (sg.walk == 0) { sg.walk = 1; Q.walks.push_back(WalkAsk{&T, T.seg_k}); return false; }      // mm_test_zdrop's walk: on the device
				else if (sg.walk == 1) return false;
				else {
					const int q_len = sg.wq1 - sg.wq0, t_len = sg.wt1 - sg.wt0;
					const bool test_inv = !(opt.flag & (MM_F_SPLICE|MM_F_SR|MM_F_FOR_ONLY|MM_F_REV_ONLY)) && sg.max_zdrop > opt.zdrop_inv && q_len < opt.max_gap && t_len < opt.max_gap;   // align.c:78
					if (!test_inv) sg.zcode = sg.max_zdrop > opt.zdrop ? 1 : 0;
					else if (ll_on_device(opt, q_len, t_len)) {
						// the window against its own reverse complement: query = the other strand, [L - (qs+wq1), +q_len)
						sg.ll_job = request(Q, 1 - T.rev, T.rid, qlen - (sg.qs + sg.wq1), q_len, sg.rs + sg.wt0, t_len, 0, 0, -1, 0, PGA_JOB_LL);
						// whatever the answer, the second pass runs when both thresholds agree (they do in every asm preset): the return
						// code is 1 or 2 (max_zdrop > zdrop_inv == zdrop), and 2 only sets split_inv of the piece split off here --
						// so this region goes on with the second pass and leaves the query's answer to that piece
						if (opt.zdrop == opt.zdrop_inv) { if (sg.job2 < 0) sg.job2 = second_pass(Q, T, sg, opt.zdrop); sg.zcode = 1; sg.ll_deferred = true; Q.jobs[(size_t)sg.ll_job].pad[1] = 1; }
					} else {
						int q_end, t_end;
						sg.zcode = zcode_of(sg, ll_on_host(Q, 1 - T.rev, qlen - (sg.qs + sg.wq1), q_len, T.rid, sg.rs + sg.wt0, t_len, false, &q_end, &t_end));
					}
				}
				if (sg.zcode > 0 && sg.job2 < 0) sg.job2 = second_pass(Q, T, sg, sg.zcode == 2 ? opt.zdrop_inv : opt.zdrop);
				if (sg.ll_deferred && !have(Q, sg.job2)) return false;
			}
			if (sg.zcode < 0) {
				if (!have(Q, sg.ll_job)) return false;
				sg.zcode = zcode_of(sg, Q.res[(size_t)sg.ll_job].score);
				if (sg.zcode > 0 && sg.job2 < 0) sg.job2 = second_pass(Q, T, sg, sg.zcode == 2 ? opt.zdrop_inv : opt.zdrop);
			}
			if (sg.zcode != 0) { if (!have(Q, sg.job2)) return false; final_job = sg.job2; }
			const DpRes &ez = Q.res[(size_t)final_job];
			if (ez.n_cigar > 0) append_ops(r, (uint32_t)ez.n_cigar, Q.cig[(size_t)final_job]);
			if (ez.zdropped) {
				r.has_p = true;
				int j = sg.i - 1;
				if (!dev_plan) { while (j >= 0 && A.tpos(T.as1 + j) > sg.rs + ez.max_t) --j; }
				else {
					// the anchors stayed on the device: the walk looks at those inside the segment and, when the maximum sits in its first bases, at a
					// few before it (consecutive anchors of a chain are at least one base apart, and the segment starts half a seed behind its first)
					const int lo = std::max(0, std::max(0, sg.i_prev) - (k >> 1) - 2);
					std::vector<int32_t> tp; fetch_tpos(Q, T.as1 + lo, T.as1 + sg.i, tp);
					while (j >= lo && tp[(size_t)(j - lo)] > sg.rs + ez.max_t) --j;
					if (j < lo && lo > 0) throw std::runtime_error("pga: z-drop split walked past the anchors it fetched");
				}
				T.dropped = true;
				if (j < 0) j = 0;
				r.dp_score += ez.max;
				T.re1 = sg.rs + (ez.max_t + 1), T.qe1 = sg.qs + (ez.max_q + 1);
				if (T.cnt1 - (j + 1) >= opt.min_cnt) {
					cut_region(r, r2, T.as1 + j + 1 - r.as, qlen, A, !dev_plan);
					if (r2.cnt > 0) { if (sg.ll_deferred) r2_split_inv_ll = sg.ll_job; else if (sg.zcode == 2) r2.split_inv = 1; }
				}
				break;
			} else r.dp_score += ez.score;
			++T.seg_k;
		}
		if (!T.dropped && T.qe < T.qe0 && T.re < T.re0) {
			if (!have(Q, T.right_job)) return false;
			const DpRes &ez = Q.res[(size_t)T.right_job];
			if (ez.n_cigar > 0) { append_ops(r, (uint32_t)ez.n_cigar, Q.cig[(size_t)T.right_job]); r.dp_score += ez.max; }
			T.re1 = T.re + (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
			T.qe1 = T.qe + (ez.reach_end ? T.qe0 - T.qe : ez.max_q + 1);
		}
		r.rs = T.rs1, r.re = T.re1;
		if (!T.rev) r.qs = T.qs1, r.qe = T.qe1; else r.qs = qlen - T.qe1, r.qe = qlen - T.qs1;
		if (r.has_p) { ask_finish(Q, T, T.rid, T.rs1, T.qs1, (int)r.rev); return false; }     // mm_update_extra: on the device
		T.done = true;
		return true;
	}

	// ---- mm_align1_inv (align.c:830-885) split at its DP calls ----
	// returns 0 = no inversion, 1 = waiting, 2 = r_inv produced (its CIGAR still has to be finished: fin_* describe the windows)
	int inversion(QueryCtx &Q, RegTask &T, const Reg &r1, Reg &r_inv, int32_t &fin_t_start, int32_t &fin_q_start, int &fin_q_rev)
	{
		const Reg &r2 = T.r; const int32_t qlen = Q.qlen;
		// the query window lies between the two pieces on the OTHER strand; both windows are reversed before the local alignment
		const int q_strand = r1.rev ? 0 : 1; const int32_t q_st = r1.rev ? r2.qe : qlen - r2.qs;
		auto extend_from = [&](int q_off, int t_off) {
			T.inv_q_off = q_off, T.inv_t_off = t_off;
			T.inv_job = request(Q, q_strand, r1.rid, q_st + q_off, T.inv_ql - q_off, r1.re + t_off, T.inv_tl - t_off, 0, (int)(opt.bw * 1.5), -1, opt.zdrop, DP_EXTZ_ONLY);
			T.inv_state = 1;
		};
		if (T.inv_state == 0) {
			T.inv_state = 2;
			if (!(r1.split & 1) || !(r2.split & 2)) return 0;
			if (r1.id != r1.parent && r1.parent != -2) return 0;
			if (r2.id != r2.parent && r2.parent != -2) return 0;
			if (r1.rid != r2.rid || r1.rev != r2.rev) return 0;
			const int ql = r1.rev ? r1.qs - r2.qe : r2.qs - r1.qe, tl = r2.rs - r1.re;
			if (ql < opt.min_chain_score || ql > opt.max_gap) return 0;
			if (tl < opt.min_chain_score || tl > opt.max_gap) return 0;
			T.inv_ql = ql, T.inv_tl = tl;
			if (ll_on_device(opt, ql, tl)) {
				T.inv_ll_job = request(Q, q_strand, r1.rid, q_st, ql, r1.re, tl, 1, 0, -1, 0, PGA_JOB_LL);
				T.inv_state = 3;
				return 1;
			}
			int q_end, t_end;
			const int score = ll_on_host(Q, q_strand, q_st, ql, r1.rid, r1.re, tl, true, &q_end, &t_end);
			if (score < opt.min_dp_max) return 0;
			extend_from(ql - (q_end + 1), tl - (t_end + 1));
		}
		if (T.inv_state == 3) {
			if (!have(Q, T.inv_ll_job)) return 1;
			const DpRes &lr = Q.res[(size_t)T.inv_ll_job];
			T.inv_state = 2;
			if (lr.score < opt.min_dp_max) return 0;
			extend_from(T.inv_ql - (lr.max_q + 1), T.inv_tl - (lr.max_t + 1));
		}
		if (T.inv_state == 1) {
			if (!have(Q, T.inv_job)) return 1;
			T.inv_state = 2;
			const DpRes &ez = Q.res[(size_t)T.inv_job];
			if (ez.n_cigar == 0) return 0;
			r_inv = Reg();
			append_ops(r_inv, (uint32_t)ez.n_cigar, Q.cig[(size_t)T.inv_job]);
			r_inv.dp_score = ez.max;
			r_inv.id = -1, r_inv.parent = -1, r_inv.inv = 1, r_inv.rev = !r1.rev, r_inv.rid = r1.rid;
			const int q_off = T.inv_q_off, t_off = T.inv_t_off;
			if (r_inv.rev == 0) { r_inv.qs = r2.qe + q_off; r_inv.qe = r_inv.qs + ez.max_q + 1; }
			else { r_inv.qe = r2.qs - q_off; r_inv.qs = r_inv.qe - (ez.max_q + 1); }
			r_inv.rs = r1.re + t_off; r_inv.re = r_inv.rs + ez.max_t + 1;
			fin_t_start = r1.re + t_off, fin_q_start = q_st + q_off, fin_q_rev = q_strand;
			return 2;
		}
		return 0;
	}
};

// parallel_for over a pool of persistent helper threads: a round of a small call runs a dozen of these loops, and starting eight
// std::threads for each costs more than the loop (0.3 ms a time).  The caller always takes part, so a loop makes progress even when
// every helper is busy with the loops of other batches; helpers join a loop through tickets and are counted, the caller leaves only when
// the tickets nobody took are withdrawn and the helpers that joined are done.
namespace {
struct PfJob { std::atomic<size_t> next{0}; size_t n = 0, chunk = 1; void (*run)(void*, size_t) = nullptr; void *ctx = nullptr; int active = 0; std::exception_ptr err; };
struct PfPool {
	std::mutex mu; std::condition_variable cv_work, cv_done; std::deque<PfJob*> tickets; std::vector<std::thread> th; bool stop = false;
	void loop(PfJob *j) { try { for (;;) { const size_t i0 = j->next.fetch_add(j->chunk); if (i0 >= j->n) break; const size_t i1 = std::min(j->n, i0 + j->chunk); for (size_t i = i0; i < i1; ++i) j->run(j->ctx, i); } } catch (...) { std::lock_guard<std::mutex> lk(mu); if (!j->err) j->err = std::current_exception(); j->next.store(j->n); } }
	void worker() {
		std::unique_lock<std::mutex> lk(mu);
		for (;;) {
			cv_work.wait(lk, [&] { return stop || !tickets.empty(); });
			if (stop) return;
			PfJob *j = tickets.front(); tickets.pop_front(); ++j->active;
			lk.unlock(); loop(j); lk.lock();
			if (--j->active == 0) cv_done.notify_all();
		}
	}
	void grow(size_t want) { while (th.size() < want) th.emplace_back([this] { worker(); }); }     // (mu held)
	~PfPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv_work.notify_all(); for (auto &t : th) t.join(); }
};
PfPool &pf_pool() { static PfPool *p = new PfPool(); return *p; }       // (leaked on purpose: no destructor order games at exit)
}
// what a call leaves behind on the host (thousands of small heap blocks near the root of a build) is freed by a janitor thread, off the call's path
template <class T> static void scrap_later(std::vector<T> &&v)
{
	struct Janitor {
		std::mutex mu; std::condition_variable cv; std::deque<std::vector<T>> q; std::thread th;
		Janitor() : th([this] { for (;;) { std::vector<T> v; { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !q.empty(); }); v = std::move(q.front()); q.pop_front(); } v.clear(); } }) { th.detach(); }
	};
	static Janitor *J = new Janitor();
	{ std::lock_guard<std::mutex> lk(J->mu); J->q.push_back(std::move(v)); }
	J->cv.notify_one();
}
// (the untyped form is what the other files of the library use: pga_common.h, pool_for)
void pool_for_raw(size_t n, int n_threads, void (*run)(void*, size_t), void *ctx)
{
	if (n_threads <= 1 || n < 2) { for (size_t i = 0; i < n; ++i) run(ctx, i); return; }
	// (items are taken a few at a time once there are thousands: one shared counter)
	PfJob job; job.n = n; job.chunk = std::max<size_t>(1, n / 256); job.ctx = ctx; job.run = run;
	const size_t helpers = std::min<size_t>((size_t)n_threads - 1, n - 1);
	PfPool &P = pf_pool();
	{
		std::lock_guard<std::mutex> lk(P.mu);
		P.grow(std::min<size_t>(64, std::max<size_t>(P.th.size(), (size_t)std::max(usable_cpus(), n_threads))));
		for (size_t h = 0; h < helpers; ++h) P.tickets.push_back(&job);
	}
	P.cv_work.notify_all();
	P.loop(&job);
	{
		std::unique_lock<std::mutex> lk(P.mu);
		for (auto it = P.tickets.begin(); it != P.tickets.end();) it = *it == &job ? P.tickets.erase(it) : it + 1;
		P.cv_done.wait(lk, [&] { return job.active == 0; });
	}
	if (job.err) std::rethrow_exception(job.err);
}
template <class F> static void parallel_for(size_t n, int n_threads, F f) { pool_for_raw(n, n_threads, [](void *c, size_t i) { (*static_cast<F*>(c))(i); }, &f); }

// ---------------------------------------------------------------- device-side planning of a list of regions (pga_plan.hip)
static void plan_list(const SeqSet &S, Driver &D, std::vector<std::pair<QueryCtx*, RegTask*>> &list, hipStream_t st, bool verbose)
{
	const double t0 = wall_s();
	std::vector<PlanIn> in(list.size());
	for (size_t i = 0; i < list.size(); ++i) {
		const QueryCtx &q = *list[i].first; const Reg &r = list[i].second->r;
		PlanIn &p = in[i]; memset(&p, 0, sizeof(p));
		p.a_off = q.a_off; p.n_a = q.n_a; p.as = r.as; p.cnt = r.cnt; p.qlen = q.qlen; p.qid = q.qid; p.base = q.base;
	}
	std::vector<PlanOut> out; std::vector<PlanItem> items;
	plan_regions(in, D.d_anchors, S.bases(), S.d_off.p, S.d_len.p, D.plan_params(), out, items, st);
	size_t off = 0, n_fb = 0, n_it = 0;
	for (size_t i = 0; i < list.size(); ++i) {
		QueryCtx &q = *list[i].first; RegTask &T = *list[i].second;
		if (out[i].status == 2) {
			// more long gaps than the kernel keeps: this one region is planned by the host code on a copy of its query's anchors, and the flags it
			// sets go back to the device (a piece split off later is planned from them)
			std::vector<u128> tmp((size_t)q.n_a);
			PGA_HIP(hipMemcpy(tmp.data(), D.d_anchors + q.a_off, tmp.size() * sizeof(u128), hipMemcpyDeviceToHost));
			q.a = tmp.data();
			// (with the plans on the device nobody has computed this region's extent on the host: the kernel filled it before it gave up)
			{ Reg &r = T.r; const PlanOut &O = out[i]; r.rev = (uint32_t)O.rev, r.rid = O.rid, r.rs = O.r_rs, r.re = O.r_re, r.qs = O.r_qs, r.qe = O.r_qe, r.mlen = O.r_mlen, r.blen = O.r_blen; }
			D.plan(q, T);
			q.a = nullptr;
			PGA_HIP(hipMemcpy(D.d_anchors + q.a_off + (uint64_t)T.r.as, tmp.data() + T.r.as, (size_t)T.r.cnt * sizeof(u128), hipMemcpyHostToDevice));
			++n_fb;
			continue;
		}
		D.apply_plan(q, T, out[i], items.data() + off);
		if (out[i].status == 0) off += out[i].n_items, n_it += out[i].n_items;
	}
	if (verbose) fprintf(stderr, "[pga]   plans on the device: %zu regions, %zu records back (%zu regions by the host code), %.4f s\n", list.size(), n_it, n_fb, wall_s() - t0);
}

// ---------------------------------------------------------------- one set of queries through its rounds
struct RoundRunner {
	const SeqSet &S; const mm_mapopt_t &opt; Driver &D; std::vector<QueryCtx> &Q; std::vector<std::vector<Reg>> &out;
	const std::vector<int> &qs; int set_id, n_threads; hipStream_t st; Timers *tm; DpParams P;
	bool verbose;
	std::list<PinVec<uint32_t>> pools;        // CIGAR pools of the DP rounds: results point into them until the set is done
	// Inversion queries nobody waits for (Seg::ll_deferred: the answer only sets split_inv of the piece split off there, which is read when that piece
	// has been aligned, a round or more later): a 10 kb x 10 kb ksw_ll_i16 takes 13-21 ms, four times a round of end extensions.  They run BESIDE
	// the rounds -- own host thread, stream, arena and launch lanes -- and are collected when they are done or when nothing else is left to do.
	struct AsyncLL { std::thread th; std::atomic<bool> done{false}; std::vector<DpJob> jb; std::vector<std::pair<int,int>> owner; std::vector<DpRes> rs; PinVec<uint32_t> cg; Timers tm; std::string err; };
	std::list<AsyncLL> asyncs;
	void launch_async_ll(std::vector<DpJob> &&jb, std::vector<std::pair<int,int>> &&owner)
	{
		asyncs.emplace_back();
		AsyncLL &A = asyncs.back();
		A.jb = std::move(jb); A.owner = std::move(owner);
		int dev = 0; PGA_HIP(hipGetDevice(&dev));
		const PkBases bases = S.bases(); const DpParams Pc = P; const bool keep_tm = tm != nullptr;
		A.th = std::thread([&A, dev, bases, Pc, keep_tm] {
			hipStream_t ss = nullptr; int arena = -1;
			try {
				PGA_HIP(hipSetDevice(dev));
				arena = dev_lease_arena();
				ArenaScope arena_scope(arena);
				set_thread_budget(1);
				ss = stream_lease();
				dp_run(bases, A.jb, Pc, A.rs, A.cg, ss, keep_tm ? &A.tm : nullptr);
				PGA_HIP(sync_stream(ss));
			} catch (std::exception &e) { A.err = e.what(); if (A.err.empty()) A.err = "unknown error"; }
			if (ss) stream_release(ss);
			if (arena >= 0) dev_release_arena(arena);
			A.done.store(true, std::memory_order_release);
		});
	}
	// results of the finished asynchronous queries into their queries' records (wait = true: of all of them); returns how many arrived
	size_t harvest_async(bool wait)
	{
		size_t got = 0;
		for (auto it = asyncs.begin(); it != asyncs.end();) {
			AsyncLL &A = *it;
			if (!wait && !A.done.load(std::memory_order_acquire)) { ++it; continue; }
			A.th.join();
			if (!A.err.empty()) { const std::string e = A.err; for (auto &B : asyncs) if (B.th.joinable()) B.th.join(); asyncs.clear(); throw std::runtime_error(e); }
			for (size_t i = 0; i < A.rs.size(); ++i) { QueryCtx &q = Q[(size_t)A.owner[i].first]; const int id = A.owner[i].second; q.res[(size_t)id] = A.rs[i]; q.res[(size_t)id].pad = 1; q.cig[(size_t)id] = nullptr; }
			if (tm) { tm->dp_bases += A.tm.dp_bases; for (int i = 0; i < K_COUNT; ++i) { tm->kern[i].ms += A.tm.kern[i].ms; tm->kern[i].launches += A.tm.kern[i].launches; tm->kern[i].alg_bytes += A.tm.kern[i].alg_bytes; tm->kern[i].cells += A.tm.kern[i].cells; } }
			got += A.rs.size();
			it = asyncs.erase(it);
		}
		return got;
	}
	~RoundRunner() { for (auto &A : asyncs) if (A.th.joinable()) A.th.join(); }

	// identity probes: of the pending problems (answered ones leave the pending lists) and of the segments that have no problem record (Seg::job1 == -2:
	// answered ones keep their count, the others become problems now)
	void run_probes()
	{
		const size_t n_q = qs.size();
		std::vector<size_t> off(n_q + 1, 0);
		parallel_for(n_q, n_threads, [&](size_t k) {
			size_t c = 0; const QueryCtx &q = Q[(size_t)qs[k]];
			for (int id : q.pending) c += q.jobs[(size_t)id].pad[0];
			for (const RegTask *T : q.probe_tasks) for (const Seg &sg : T->segs) c += sg.job1 == -2 && sg.pm < 0;
			off[k + 1] = c;
		});
		for (size_t k = 0; k < n_q; ++k) off[k + 1] += off[k];
		const size_t n = off[n_q];
		if (!n) return;
		PinVec<PostProbe> pr; pr.resize(n);
		parallel_for(n_q, n_threads, [&](size_t k) {
			const QueryCtx &q = Q[(size_t)qs[k]]; size_t o = off[k];
			for (int id : q.pending) { const DpJob &j = q.jobs[(size_t)id]; if (j.pad[0]) pr[o++] = PostProbe{j.t_off, j.q_off, j.qlen_full, j.qs, j.qlen, (int32_t)j.q_rev}; }
			const uint64_t q_off = S.off[(size_t)q.qid];
			for (const RegTask *T : q.probe_tasks) {
				const uint64_t t0 = S.off[(size_t)(q.base + T->rid)];
				for (const Seg &sg : T->segs) if (sg.job1 == -2 && sg.pm < 0) pr[o++] = PostProbe{t0 + (uint64_t)sg.rs, q_off, q.qlen, sg.qs, sg.qe - sg.qs, (int32_t)T->rev};
			}
		});
		PinVec<int32_t> m;
		const double t0 = wall_s();
		post_identity(S.bases(), pr, D.probe_m_max, m, st);
		std::atomic<size_t> n_yes(0);
		parallel_for(n_q, n_threads, [&](size_t k) {
			QueryCtx &q = Q[(size_t)qs[k]]; size_t o = off[k], w = 0, yes = 0;
			for (size_t i = 0; i < q.pending.size(); ++i) {
				const int id = q.pending[i];
				DpJob &j = q.jobs[(size_t)id];
				if (j.pad[0]) { j.pad[0] = 0; const int mm = m[o++]; if (mm >= 0) { D.answer_probe(q, id, mm); ++yes; continue; } }
				q.pending[w++] = id;
			}
			q.pending.resize(w);
			for (RegTask *T : q.probe_tasks)
				for (Seg &sg : T->segs) if (sg.job1 == -2 && sg.pm < 0) {
					const int mm = m[o++];
					if (mm >= 0) { sg.pm = mm; ++yes; }
					else sg.job1 = D.request(q, T->rev, T->rid, sg.qs, sg.qe - sg.qs, sg.rs, sg.re - sg.rs, 0, sg.bw1, -1, opt.zdrop, DP_APPROX_MAX, false);    // (the probe has spoken)
				}
			q.probe_tasks.clear();
			n_yes += yes;
		});
		if (verbose) fprintf(stderr, "[pga]   set %d: %zu identity probes, %zu answered without a matrix, %.3f s\n", set_id, n, n_yes.load(), wall_s() - t0);
	}

	void run_dp(int round)
	{
		const size_t n_q = qs.size();
		std::vector<size_t> poff(n_q + 1, 0);
		for (size_t k = 0; k < n_q; ++k) poff[k + 1] = poff[k] + Q[(size_t)qs[k]].pending.size();
		const size_t n_pend = poff[n_q];
		if (!n_pend) return;
		std::vector<DpJob> jb(n_pend); std::vector<std::pair<int,int>> owner(n_pend);
		std::vector<double> cells_of(n_q, 0.0);
		parallel_for(n_q, n_threads, [&](size_t k) {
			const size_t qi = (size_t)qs[k];
			QueryCtx &q = Q[qi];
			size_t o = poff[k]; double cells = 0;
			for (int id : q.pending) { jb[o] = q.jobs[(size_t)id]; owner[o] = std::make_pair((int)qi, id); cells += (double)q.jobs[(size_t)id].qlen * q.jobs[(size_t)id].tlen; ++o; }
			q.pending.clear();
			cells_of[k] = cells;
		});
		static const bool async_ll = getenv("PGA_LL_ASYNC") != nullptr;     // measured: no gain (6.1-5.7 against 6.2-6.5 Gbp/s): one more stream and lane set per batch crowd the hardware queues
		if (async_ll) {
			std::vector<DpJob> ajb; std::vector<std::pair<int,int>> aown; size_t w = 0;
			for (size_t i = 0; i < jb.size(); ++i) {
				if ((jb[i].flag & PGA_JOB_LL) && jb[i].pad[1]) { ajb.push_back(jb[i]); aown.push_back(owner[i]); }
				else { if (w != i) { jb[w] = jb[i]; owner[w] = owner[i]; } ++w; }
			}
			jb.resize(w); owner.resize(w);
			if (!ajb.empty()) { if (verbose) fprintf(stderr, "[pga]   set %d round %d: %zu inversion queries run beside the rounds\n", set_id, round, ajb.size()); launch_async_ll(std::move(ajb), std::move(aown)); }
			if (jb.empty()) return;
		}
		std::vector<DpRes> rs;
		pools.emplace_back();
		PinVec<uint32_t> &cg = pools.back();
		const double t_dp = wall_s();
		dp_run(S.bases(), jb, P, rs, cg, st, tm);
		if (verbose) fprintf(stderr, "[pga]   set %d round %d: %zu DP problems in %.3f s\n", set_id, round, jb.size(), wall_s() - t_dp);
		if (tm) { tm->dp_jobs += (double)jb.size(); for (double c : cells_of) tm->dp_cells += c; }
		const uint32_t *base = cg.data();
		parallel_for((rs.size() + 65535) / 65536, n_threads, [&](size_t blk) {
			const size_t lo = blk * 65536, hi = std::min(rs.size(), lo + 65536);
			for (size_t i = lo; i < hi; ++i) {
				QueryCtx &q = Q[(size_t)owner[i].first]; const int id = owner[i].second;
				if (rs[i].n_cigar < 0) throw std::runtime_error("pga: DP backtrack did not terminate");
				q.res[(size_t)id] = rs[i]; q.res[(size_t)id].pad = 1;
				q.cig[(size_t)id] = base + rs[i].cigar_off;
			}
		});
	}

	// one pass over the queries of the set; returns the number that still wait for something
	int advance_pass()
	{
		std::atomic<int> unfinished(0);
		parallel_for(qs.size(), n_threads, [&](size_t k) {
			const size_t qi = (size_t)qs[k];
			QueryCtx &q = Q[qi];
			if (q.finished) return;
			// walk the list in the reference's order (align.c:981-1010).  Regions are independent, so one that waits does not hold up the
			// others; only the inversion test looks at the previous list element and is deferred while anything before it is open.
			bool waiting = false;
			for (size_t i = 0; i < q.list.size(); ++i) {
				RegTask &T = *q.list[i];
				if (T.is_inv) { if (!T.done) { if (T.fin == 2) T.done = true; else waiting = true; } continue; }
				if (!T.planned) { if (D.dev_plan) { q.to_plan.push_back(&T); waiting = true; continue; } D.plan(q, T); }
				if (!T.done) {
					Reg r2; int r2_ll = -1;
					const bool complete = D.advance(q, T, r2, r2_ll);
					if (r2.cnt > 0) { q.pool.emplace_back(new RegTask()); q.pool.back()->r = r2; q.pool.back()->split_inv_ll = r2_ll; q.list.insert(q.list.begin() + (long)i + 1, q.pool.back().get()); }
					if (!complete) { waiting = true; continue; }
				}
				if (T.split_inv_ll >= 0) {             // the inversion query of the split that made this piece: its answer is split_inv
					if (!Driver::have(q, T.split_inv_ll)) { waiting = true; continue; }
					const int sc = q.res[(size_t)T.split_inv_ll].score;
					T.r.split_inv = (sc >= opt.min_chain_score * opt.a && sc >= opt.min_dp_max) ? 1 : 0;
					T.split_inv_ll = -1;
				}
				if (i > 0 && T.r.split_inv && !(opt.flag & MM_F_NO_INV) && T.inv_state != 2) {
					if (waiting) continue;   // an earlier element is still open: decide later
					Reg r_inv; int32_t ft = 0, fq = 0; int frev = 0;
					const int rc = D.inversion(q, T, q.list[i - 1]->r, r_inv, ft, fq, frev);
					if (rc == 1) { waiting = true; continue; }
					if (rc == 2) {
						q.pool.emplace_back(new RegTask()); RegTask *ti = q.pool.back().get();
						ti->r = r_inv; ti->is_inv = ti->planned = true;
						q.list.insert(q.list.begin() + (long)i + 1, ti);
						D.ask_finish(q, *ti, r_inv.rid, ft, fq, frev);
						waiting = true;
					}
				}
			}
			if (waiting) { ++unfinished; return; }
			// ---- all regions aligned: filters, ranking, mapq (align.c:1013-1021, map.c:340-341) ----
			std::vector<Reg> regs; regs.reserve(q.list.size());
			for (RegTask *t : q.list) regs.push_back(std::move(t->r));
			drop_weak_regions(opt, q.qlen, regs);
			if (q.qlen >= opt.rank_min_len) { rescale_dp_max(q.qlen, regs, opt.rank_frac, opt.a, opt.b); drop_weak_regions(opt, q.qlen, regs); }
			order_regions(regs);
			assign_mapq(regs, opt.min_chain_score, opt.a, q.rep_len);
			out[qi] = std::move(regs);
			q.finished = true; q.pool.clear(); q.list.clear();
		});
		return unfinished.load();
	}

	// device-side planning of the regions the last advance pass split off; returns false if there were none
	bool run_plans()
	{
		if (!D.dev_plan) return false;
		std::vector<std::pair<QueryCtx*, RegTask*>> list;
		for (int qi : qs) { QueryCtx &q = Q[(size_t)qi]; for (RegTask *t : q.to_plan) if (!t->planned) list.emplace_back(&q, t); q.to_plan.clear(); }
		if (list.empty()) return false;
		std::sort(list.begin(), list.end()); list.erase(std::unique(list.begin(), list.end()), list.end());
		plan_list(S, D, list, st, verbose);
		return true;
	}

	// the device requests the last advance pass raised; returns false if there were none
	bool run_post()
	{
		std::vector<PostWalk> walks; std::vector<uint32_t> wcig; std::vector<WalkAsk> wask;
		std::vector<PostFin> fins; std::vector<RegTask*> ftask; size_t fin_ops = 0;
		for (int qi : qs) {
			QueryCtx &q = Q[(size_t)qi];
			for (const WalkAsk &w : q.walks) {
				const Seg &sg = w.T->segs[w.seg]; const DpRes &e1 = q.res[(size_t)sg.job1];
				walks.push_back(PostWalk{S.off[(size_t)(q.base + w.T->rid)] + (uint64_t)sg.rs, S.off[(size_t)q.qid], q.qlen, sg.qs, w.T->rev, (uint32_t)e1.n_cigar, (uint64_t)wcig.size()});
				wcig.insert(wcig.end(), q.cig[(size_t)sg.job1], q.cig[(size_t)sg.job1] + e1.n_cigar);
				wask.push_back(w);
			}
			q.walks.clear();
			for (RegTask *t : q.fins) {
				fins.push_back(PostFin{t->fin_t_off, S.off[(size_t)q.qid], q.qlen, t->fin_q_start, t->fin_q_rev, (uint32_t)t->r.cigar.size(), (uint64_t)fin_ops});
				fin_ops += t->r.cigar.size(); ftask.push_back(t);
			}
			q.fins.clear();
		}
		if (walks.empty() && fins.empty()) return false;
		const double t0 = wall_s();
		if (!walks.empty()) {
			std::vector<PostWalkRes> wr;
			post_zdrop_walk(S.bases(), walks, wcig, P, wr, st);
			for (size_t i = 0; i < wask.size(); ++i) {
				Seg &sg = wask[i].T->segs[wask[i].seg];
				sg.max_zdrop = wr[i].max_zdrop, sg.wt0 = wr[i].t0, sg.wt1 = wr[i].t1, sg.wq0 = wr[i].q0, sg.wq1 = wr[i].q1; sg.walk = 2;
			}
		}
		if (!fins.empty()) {
			PinVec<uint32_t> ops; ops.resize(fin_ops);
			parallel_for(ftask.size(), n_threads, [&](size_t i) { const auto &c = ftask[i]->r.cigar; if (!c.empty()) memcpy(ops.data() + fins[i].cig_off, c.data(), c.size() * 4); });
			std::vector<PostFinRes> fr;
			post_cigar_finish(S.bases(), fins, ops, P, fr, st);
			parallel_for(ftask.size(), n_threads, [&](size_t i) {
				RegTask &T = *ftask[i]; Reg &r = T.r; const PostFinRes &f = fr[i];
				r.cigar.assign(ops.data() + fins[i].cig_off, ops.data() + fins[i].cig_off + f.n_cigar);
				if (f.qshift) { if (r.rev) r.qe -= f.qshift; else r.qs += f.qshift; }      // a leading insertion / deletion left the record (align.c:147-166)
				r.rs += f.tshift;
				r.blen = f.blen, r.mlen = f.mlen, r.n_ambi += (uint32_t)f.n_ambi, r.dp_max = f.dp_max;
				if (f.q_span != r.qe - r.qs || f.t_span != r.re - r.rs) throw std::runtime_error("pga: CIGAR does not span its region");   // align.c:286
				T.fin = 2;
			});
		}
		if (verbose) fprintf(stderr, "[pga]   set %d: %zu z-drop walks, %zu CIGAR finishes (%zu operations) on the device, %.3f s\n", set_id, walks.size(), fins.size(), fin_ops, wall_s() - t0);
		return true;
	}

	// Rounds are a barrier over the queries of the set: round r+1 starts when the slowest problem of round r is done.  After the bulk
	// (round 0) only the few queries with split chains are left, each with its own CHAIN of dependent rounds (one per split point,
	// second pass, inversion test) whose lengths and problem sizes have nothing to do with one another.  From then on every such
	// query runs its rounds on its own (own host thread, stream, arena and launch lanes), a few at a time.
	void run_tails(const std::vector<int> &open)
	{
		std::atomic<size_t> next(0);
		std::vector<std::string> errs(open.size());
		std::vector<Timers> tms(open.size());
		std::vector<std::vector<int>> one(open.size());
		int dev = 0; PGA_HIP(hipGetDevice(&dev));
		auto worker = [&] {
			for (;;) {
				const size_t k = next.fetch_add(1);
				if (k >= open.size()) break;
				hipStream_t ss = nullptr;
				const int arena = dev_lease_arena();
				ArenaScope arena_scope(arena);
				try {
					PGA_HIP(hipSetDevice(dev));
					set_thread_budget(1);
					ss = stream_lease();
					one[k].assign(1, open[k]);
					Driver Dq(S, opt, D.k, ss); Dq.dev_plan = D.dev_plan; Dq.d_anchors = D.d_anchors;
					RoundRunner R{S, opt, Dq, Q, out, one[k], set_id * 1000 + (int)k + 1, 1, ss, tm ? &tms[k] : nullptr, P, verbose, {}};
					R.tail = true;
					R.run();
				} catch (std::exception &e) { errs[k] = e.what(); if (errs[k].empty()) errs[k] = "unknown error"; }
				if (ss) stream_release(ss);
				dev_release_arena(arena);
			}
		};
		static const int conc = getenv("PGA_TAIL_THREADS") ? std::max(1, atoi(getenv("PGA_TAIL_THREADS"))) : 6;
		std::vector<std::thread> th;
		for (int t = 0; t < std::min<int>(conc, (int)open.size()); ++t) th.emplace_back(worker);
		for (auto &t : th) t.join();
		for (auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
		if (tm) for (const Timers &t : tms) {
			tm->dp_jobs += t.dp_jobs; tm->dp_cells += t.dp_cells; tm->dp_bases += t.dp_bases; tm->dp_cigar_ops += t.dp_cigar_ops;
			for (int i = 0; i < K_COUNT; ++i) { tm->kern[i].ms += t.kern[i].ms; tm->kern[i].launches += t.kern[i].launches; tm->kern[i].alg_bytes += t.kern[i].alg_bytes; tm->kern[i].cells += t.kern[i].cells; }
		}
	}

	bool tail = false;

	void run()
	{
		static const int tail_max = getenv("PGA_TAIL_QUERIES") ? atoi(getenv("PGA_TAIL_QUERIES")) : 0;      // off by default: measured slower (the extra streams and lane sets get in the way of the other parts), see DESIGN.md
		for (int round = 0; round < 100000; ++round) {
			const double t_r0 = wall_s();
			run_probes();
			const double t_r1 = wall_s();
			run_dp(round);
			if (verbose) fprintf(stderr, "[pga]   set %d round %d: probes %.4f s, dp %.4f s\n", set_id, round, t_r1 - t_r0, wall_s() - t_r1);
			const double t_adv0 = wall_s();
			int unfinished;
			harvest_async(false);
			for (;;) {
				for (;;) { unfinished = advance_pass(); bool more = run_plans(); more |= run_post(); if (!more) break; }
				if (unfinished == 0) break;
				bool pend = false; for (int qi : qs) pend |= !Q[(size_t)qi].pending.empty() || !Q[(size_t)qi].probe_tasks.empty();
				if (pend || asyncs.empty()) break;
				harvest_async(true);                               // nothing else to do: the queries that are still open wait for an inversion query
			}
			if (verbose) fprintf(stderr, "[pga]   set %d round %d: host advance + device post-processing %.3f s\n", set_id, round, wall_s() - t_adv0);
			if (unfinished == 0) { harvest_async(true); break; }       // (a query whose split left no piece never reads its answer: rare)
			bool any_pending = false; for (int qi : qs) any_pending |= !Q[(size_t)qi].pending.empty() || !Q[(size_t)qi].probe_tasks.empty();
			if (!any_pending) throw std::runtime_error("pga: alignment driver stalled");
			if (!tail && unfinished > 1 && unfinished <= tail_max) {
				std::vector<int> open;
				for (int qi : qs) if (!Q[(size_t)qi].finished) open.push_back(qi);
				if (verbose) fprintf(stderr, "[pga]   set %d: %zu queries continue on their own after round %d\n", set_id, open.size(), round);
				run_tails(open);
				break;
			}
		}
	}
};

// regions + alignment of the whole batch: chains in, final records out
void align_batch(const SeqSet &S, const mm_mapopt_t &opt, int k, const std::vector<uint64_t> &q_aoff, ChainResult &C, const std::vector<int32_t> &rep_len,
                 std::vector<std::vector<Reg>> &out, int n_threads, Timers *tm, hipStream_t st)
{
	const int n_seq = S.n_seq;
	const double t_align0 = wall_s();
	out.assign((size_t)n_seq, {});
	const bool verbose = getenv("PGA_VERBOSE") != nullptr;
	Driver D(S, opt, k, st);
	// planning on the device (pga_plan.hip) whenever the compacted anchors are there and the lean identity probes apply; PGA_HOST_PLAN=1: the
	// round-3 path (anchors downloaded, planned by the host threads)
	D.dev_plan = C.d_a.p != nullptr && !C.want_host_anchors && D.lean_probes && (opt.flag & MM_F_CIGAR);
	D.d_anchors = C.d_a.p;
	std::vector<QueryCtx> Q((size_t)n_seq);
	if (verbose) fprintf(stderr, "[pga]   align: contexts of %d queries %.4f s\n", n_seq, wall_s() - t_align0);
	// ---- regions (mm_gen_regs) and plans ----
	std::vector<u128> heads; std::vector<uint64_t> head_off((size_t)n_seq + 1, 0);
	if (D.dev_plan) {
		// the first anchor of every chain (it salts the region order, hit.c:64-65): gathered on the device
		std::vector<uint64_t> idx;
		for (int qi = 0; qi < n_seq; ++qi) {
			head_off[(size_t)qi] = idx.size();
			uint64_t start = q_aoff[(size_t)qi];
			for (int c = 0; c < C.n_u[(size_t)qi]; ++c) { idx.push_back(start); start += (uint32_t)C.u[q_aoff[(size_t)qi] + (uint64_t)c]; }
		}
		head_off[(size_t)n_seq] = idx.size();
		gather_anchors(idx, C.d_a.p, heads, st);
	}
	parallel_for((size_t)n_seq, n_threads, [&](size_t qi) {
		QueryCtx &q = Q[qi];
		q.qid = (int)qi, q.qlen = (int32_t)S.len[qi], q.rep_len = rep_len[qi]; q.base = (int)S.grp_off[S.grp_of_seq[qi]];
		const int n_u = C.n_u[qi];
		if (q.qlen == 0 || n_u == 0) { q.finished = true; return; }
		const uint64_t b = q_aoff[qi];
		q.a = D.dev_plan ? nullptr : C.a.data() + b;
		q.a_off = b;
		q.n_a = C.n_v[qi];                          // chains are contiguous and every chain becomes a region: nothing to squeeze (hit.c:311-329)
		uint32_t salt = !(opt.flag & MM_F_NO_HASH_NAME) ? name_hash31(S.name[qi]) : 0;
		salt = mix32(salt ^ (mix32((uint32_t)q.qlen) + mix32((uint32_t)opt.seed)));
		std::vector<Reg> regs;
		regions_from_chains(salt, q.qlen, n_u, C.u.data() + b, Anchors{q.a, q.n_a}, regs, D.dev_plan ? heads.data() + head_off[qi] : nullptr);
		for (auto &r : regs) { q.pool.emplace_back(new RegTask()); q.pool.back()->r = r; q.list.push_back(q.pool.back().get()); }
		if (!(opt.flag & MM_F_CIGAR) || D.dev_plan) return;
		for (RegTask *t : q.list) D.plan(q, *t);
	});
	if (D.dev_plan) {
		std::vector<std::pair<QueryCtx*, RegTask*>> list;
		for (int qi = 0; qi < n_seq; ++qi) for (RegTask *t : Q[(size_t)qi].list) list.emplace_back(&Q[(size_t)qi], t);
		plan_list(S, D, list, st, verbose);
	}
	if (verbose) fprintf(stderr, "[pga]   align: regions+plans %.3f s (%d threads%s)\n", wall_s() - t_align0, n_threads, D.dev_plan ? ", plans on the device" : "");
	if (!(opt.flag & MM_F_CIGAR)) {
		for (int qi = 0; qi < n_seq; ++qi) {
			QueryCtx &q = Q[(size_t)qi];
			if (q.finished) continue;
			std::vector<Reg> regs; for (RegTask *t : q.list) regs.push_back(std::move(t->r));
			assign_mapq(regs, opt.min_chain_score, opt.a, q.rep_len);
			out[(size_t)qi] = std::move(regs);
		}
		return;
	}
	// ---- rounds ----
	// The queries are dealt into a few SETS that run their rounds concurrently (one host thread, one stream and one
	// device-memory arena each): while the GPU works on one set's problems the host classifies, collects and advances
	// another's, and the short dependent rounds at the end of one set hide behind the bulk of the next.
	const DpParams P{opt.q, opt.e, opt.q2, opt.e2, D.mat[0], D.mat[1], D.mat[24], dp_lb_mode()};
	// deal the queries by anchor count (largest first, round robin): balanced sets
	// (two sets pay from a few hundred queries on; below that the few long problems of a set only get in each other's way)
	int n_sets = getenv("PGA_ALIGN_SETS") ? atoi(getenv("PGA_ALIGN_SETS")) : (n_seq >= 256 && part_concurrency() < 2 ? 2 : 1);
	if (n_sets < 1) n_sets = 1;
	if (n_seq < 8 * n_sets) n_sets = 1;
	std::vector<int> order((size_t)n_seq);
	std::iota(order.begin(), order.end(), 0);
	std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return Q[(size_t)x].n_a > Q[(size_t)y].n_a; });
	std::vector<std::vector<int>> sets((size_t)n_sets);
	for (int i = 0; i < n_seq; ++i) sets[(size_t)(i % n_sets)].push_back(order[(size_t)i]);
	for (auto &v : sets) std::sort(v.begin(), v.end());
	if (verbose) fprintf(stderr, "[pga]   align: %d set(s) dealt at +%.4f s\n", n_sets, wall_s() - t_align0);
	// the contexts of a call hold a few small heap blocks per query and region (thousands of sequences per call near the root): a janitor
	// thread takes them apart, not this call at its closing brace
	auto scrap = [&] { scrap_later(std::move(Q)); };
	if (n_sets == 1) { RoundRunner R{S, opt, D, Q, out, sets[0], 0, n_threads, st, tm, P, verbose, {}}; R.run(); scrap(); if (verbose) fprintf(stderr, "[pga]   align: rounds done at +%.4f s\n", wall_s() - t_align0); return; }
	int dev = 0; PGA_HIP(hipGetDevice(&dev));
	std::vector<Timers> tms((size_t)n_sets);
	std::vector<std::string> errs((size_t)n_sets);
	std::vector<std::thread> th;
	for (int s = 0; s < n_sets; ++s) th.emplace_back([&, s] {
		hipStream_t ss = nullptr;
		const int arena = dev_lease_arena();      // the set's own stream gets its own arena
		ArenaScope arena_scope(arena);
		try {
			PGA_HIP(hipSetDevice(dev));
			set_thread_budget(std::max(1, n_threads / n_sets));
			ss = stream_lease();
			Driver Ds(S, opt, k, ss); Ds.dev_plan = D.dev_plan; Ds.d_anchors = D.d_anchors;
			RoundRunner R{S, opt, Ds, Q, out, sets[(size_t)s], s, std::max(1, n_threads / n_sets), ss, tm ? &tms[(size_t)s] : nullptr, P, verbose, {}};
			R.run();
		} catch (std::exception &e) { errs[(size_t)s] = e.what(); if (errs[(size_t)s].empty()) errs[(size_t)s] = "unknown error"; }
		if (ss) stream_release(ss);
		dev_release_arena(arena);                 // (the DP lane streams drained inside dp_run; the set's stream just did)
	});
	for (auto &t : th) t.join();
	if (verbose) fprintf(stderr, "[pga]   align: sets joined at +%.4f s\n", wall_s() - t_align0);
	scrap();
	if (verbose) fprintf(stderr, "[pga]   align: contexts taken apart at +%.4f s\n", wall_s() - t_align0);
	for (auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
	if (tm) for (const Timers &t : tms) {
		tm->dp_jobs += t.dp_jobs; tm->dp_cells += t.dp_cells; tm->dp_bases += t.dp_bases; tm->dp_cigar_ops += t.dp_cigar_ops;
		for (int i = 0; i < K_COUNT; ++i) { tm->kern[i].ms += t.kern[i].ms; tm->kern[i].launches += t.kern[i].launches; tm->kern[i].alg_bytes += t.kern[i].alg_bytes; tm->kern[i].cells += t.kern[i].cells; }
	}
}

} // namespace pga
