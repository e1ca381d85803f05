// pga_reconsensus.hip -- SURVEY 8(f)-4: the reconsensus of the blocks a merge updated, on the device.
//
// Replaces, for all updated blocks of a merge at once (reference: packages/pangraph/src/),
//   reconsensus/reconsensus.rs:32-126     analyze_blocks_for_reconsensus + the per-block work of reconsensus_graph
//   pangraph/pangraph_block.rs:191-256    find_majority_substitutions / _deletions / _insertions (is_majority: count > depth / 2)
//   utils/interval.rs:60-86               positions_to_intervals
//   pangraph/pangraph_block.rs:258-291    change_consensus_nucleotide_at_pos, edits.rs:157-238 reconcile_substitution_with_consensus
//   pangraph/pangraph_block.rs:295-332    edit_consensus_and_realign: Edit::apply (edits.rs:307-329) of the majority edits to the consensus and
//                                         of every member's edits to the old consensus, then map_variations per member (pga_mapvar.hip)
// What runs where:
//   device   the counting: substitutions as sorted (block, position, letter) keys with run lengths; deletions as a coverage array over all
//            consensus positions (+1 / -1 at the ends of every deletion, one scan) cut into intervals; insertions as sorted (block, position,
//            hash of the letters) keys, every run verified letter by letter against its first member;
//            Edit::apply as one thread per consensus position (offsets from the sorted edit lists by binary search);
//            the reconciliation of a member's substitutions with changed consensus letters (one thread per member);
//            the re-alignment (k_mapvar*, sequences never leave the device between apply and alignment)
//   host     list bookkeeping: offsets, the classification of a block from its (short) majority lists, BandParameters::from_edits
//            (map_variations.rs:29-37, edits.rs:418-531: integer arithmetic over an edit list), packing of the result
// detach_unaligned_nodes and the graph's node / path maps stay with the caller (reconsensus.rs:76-88): they are not base work.
#include "pga_common.h"
#include "../../include/pga_align.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cmath>
#include <map>
#include <numeric>
#include <string>

namespace pga {

struct MvDevJob { uint64_t ref_off, qry_off; uint32_t ref_len, qry_len; int32_t mean_shift; uint32_t band_width; };
void map_variations_dev(int64_t n, const MvDevJob *jobs, const char *d_ascii, uint64_t cat_size, const pga_mapvar_params_t &prm, pga_mapvar_res_t *res,
                        std::vector<pga_sub_t> &h_subs, std::vector<pga_del_t> &h_dels, std::vector<pga_ins_t> &h_inss, std::vector<char> &h_seq, hipStream_t st);

struct RcMember { uint32_t block; uint32_t n_subs, n_dels, n_inss; uint64_t sub_off, del_off, ins_off; };

// ---- majority counting ----
__global__ void k_rc_keys(const RcMember *__restrict__ mem, int64_t n_mem, const pga_sub_t *__restrict__ subs, const pga_del_t *__restrict__ dels, const pga_ins_t *__restrict__ inss, const char *__restrict__ ins_seq,
                          const uint64_t *__restrict__ cov_base, uint64_t *__restrict__ sub_key, int32_t *__restrict__ cov, uint64_t *__restrict__ ins_key, uint64_t *__restrict__ ins_hash)
{
	const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (m >= n_mem) return;
	const RcMember M = mem[m];
	for (uint32_t t = 0; t < M.n_subs; ++t) { const pga_sub_t s = subs[M.sub_off + t]; sub_key[M.sub_off + t] = (uint64_t)M.block << 40 | (uint64_t)s.pos << 8 | (s.alt & 255u); }
	for (uint32_t t = 0; t < M.n_dels; ++t) { const pga_del_t d = dels[M.del_off + t]; if (d.len) { atomicAdd(&cov[cov_base[M.block] + d.pos], 1); atomicAdd(&cov[cov_base[M.block] + d.pos + d.len], -1); } }
	for (uint32_t t = 0; t < M.n_inss; ++t) {
		const pga_ins_t x = inss[M.ins_off + t];
		uint64_t h = 0xcbf29ce484222325ULL ^ x.len;                     // FNV-1a over the letters (runs of equal hashes are verified letter by letter)
		for (uint32_t c = 0; c < x.len; ++c) { h ^= (uint8_t)ins_seq[x.seq_off + c]; h *= 0x100000001b3ULL; }
		ins_key[M.ins_off + t] = (uint64_t)M.block << 32 | x.pos;
		ins_hash[M.ins_off + t] = h;
	}
}
// runs of equal keys in a sorted array: flag[i] = 1 for the first element of a run whose length is a majority of its block
__global__ void k_rc_sub_major(const uint64_t *__restrict__ key, int64_t n, const uint32_t *__restrict__ depth, uint32_t *__restrict__ flag)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t f = 0;
	const uint64_t k = key[i];
	if (i == 0 || key[i - 1] != k) { int64_t j = i + 1; while (j < n && key[j] == k) ++j; f = (uint32_t)(j - i) > depth[k >> 40] / 2 ? 1u : 0u; }
	flag[i] = f;
}
__global__ void k_rc_ins_major(const uint64_t *__restrict__ key, const uint64_t *__restrict__ hash, const uint32_t *__restrict__ idx, int64_t n, const pga_ins_t *__restrict__ inss, const char *__restrict__ ins_seq,
                               const uint32_t *__restrict__ depth, uint32_t *__restrict__ flag, int *__restrict__ collision)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t f = 0;
	const uint64_t k = key[i], h = hash[i];
	if (i == 0 || key[i - 1] != k || hash[i - 1] != h) {
		const pga_ins_t a = inss[idx[i]];
		int64_t j = i + 1;
		while (j < n && key[j] == k && hash[j] == h) {
			const pga_ins_t b = inss[idx[j]];
			bool same = a.len == b.len;
			for (uint32_t c = 0; same && c < a.len; ++c) same = ins_seq[a.seq_off + c] == ins_seq[b.seq_off + c];
			if (!same) *collision = 1;                                 // equal hashes, different letters: the host recounts this call exactly
			++j;
		}
		f = (uint32_t)(j - i) > depth[k >> 32] / 2 ? 1u : 0u;
	}
	flag[i] = f;
}
__global__ void k_rc_del_flags(const int32_t *__restrict__ covs, uint64_t n_cov, const uint32_t *__restrict__ blk_of_cov_chunk, const uint64_t *__restrict__ cov_base, const uint32_t *__restrict__ depth, int n_blocks,
                               uint32_t *__restrict__ fs, uint32_t *__restrict__ fe)
{
	const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_cov) return;
	// the block of coverage slot g: binary search over the bases
	int lo = 0, hi = n_blocks - 1;
	while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (cov_base[mid] <= g) lo = mid; else hi = mid - 1; }
	const uint32_t half = depth[lo] / 2;
	const bool m = (uint32_t)(covs[g] > 0 ? covs[g] : 0) > half;
	const bool left = g > cov_base[lo] && (uint32_t)(covs[g - 1] > 0 ? covs[g - 1] : 0) > half;
	const bool right = g + 1 < cov_base[lo + 1] && (uint32_t)(covs[g + 1] > 0 ? covs[g + 1] : 0) > half;
	fs[g] = m && !left ? 1u : 0u; fe[g] = m && !right ? 1u : 0u;
	(void)blk_of_cov_chunk;
}
template <class T> __global__ void k_rc_compact(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos, uint64_t n, const T *__restrict__ in, T *__restrict__ out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && flag[i]) out[pos[i]] = in ? in[i] : (T)i;
}

// ---- Edit::apply (edits.rs:307-329) ----
// A job applies one PREPARED edit list to a stretch of `src`: substitutions stably sorted by position (the last of equal positions wins, as
// the sequential loop of the reference lets it), deletions as merged intervals with the count of deleted positions before each, insertions
// sorted by (position, letters) with the count of inserted letters before each.
struct RcApply { uint64_t src_off; uint32_t len; uint64_t out_off; uint32_t n_subs, n_dels, n_inss; uint64_t sub_off, del_off, ins_off; };
struct RcDelIv { uint32_t start, end, before; };
struct RcInsP { uint32_t pos, len, before; uint64_t seq_off; };
__global__ void k_rc_apply(const RcApply *__restrict__ jobs, const uint64_t *__restrict__ job_first, int n_jobs, uint64_t n_threads, const char *__restrict__ src, const pga_sub_t *__restrict__ subs, const RcDelIv *__restrict__ dels,
                           const RcInsP *__restrict__ inss, const char *__restrict__ ins_seq, char *__restrict__ out)
{
	const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_threads) return;
	int lo = 0, hi = n_jobs - 1;
	while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (job_first[mid] <= g) lo = mid; else hi = mid - 1; }
	const RcApply J = jobs[lo];
	const uint32_t p = (uint32_t)(g - job_first[lo]);                 // consensus position 0 .. len (len: insertions behind the last letter)
	// deleted positions before p, and whether p is deleted
	uint32_t del_before = 0; bool deleted = false;
	{
		uint32_t a = 0, b = J.n_dels;                                    // first interval with end > p
		while (a < b) { const uint32_t mid = (a + b) >> 1; if (dels[J.del_off + mid].end <= p) a = mid + 1; else b = mid; }
		if (a < J.n_dels) { const RcDelIv iv = dels[J.del_off + a]; if (iv.start <= p) { deleted = true; del_before = iv.before + (p - iv.start); } else del_before = iv.before; }
		else if (J.n_dels) { const RcDelIv iv = dels[J.del_off + J.n_dels - 1]; del_before = iv.before + (iv.end - iv.start); }
	}
	// insertions at positions < p, and those at p
	uint32_t a = 0, b = J.n_inss;
	while (a < b) { const uint32_t mid = (a + b) >> 1; if (inss[J.ins_off + mid].pos < p) a = mid + 1; else b = mid; }
	uint32_t ins_before = a < J.n_inss ? inss[J.ins_off + a].before : (J.n_inss ? inss[J.ins_off + J.n_inss - 1].before + inss[J.ins_off + J.n_inss - 1].len : 0u);
	uint64_t o = J.out_off + (uint64_t)(p - del_before) + ins_before;
	for (uint32_t t = a; t < J.n_inss && inss[J.ins_off + t].pos == p; ++t) {
		const RcInsP x = inss[J.ins_off + t];
		for (uint32_t c = 0; c < x.len; ++c) out[o + c] = ins_seq[x.seq_off + c];
		o += x.len;
	}
	if (p < J.len && !deleted) {
		char ch = src[J.src_off + p];
		uint32_t sa = 0, sb = J.n_subs;                                  // last substitution with pos == p
		while (sa < sb) { const uint32_t mid = (sa + sb) >> 1; if (subs[J.sub_off + mid].pos <= p) sa = mid + 1; else sb = mid; }
		if (sa > 0 && subs[J.sub_off + sa - 1].pos == p) ch = (char)subs[J.sub_off + sa - 1].alt;
		out[o] = ch;
	}
}

// ---- apply_substitutions_to_block (reconsensus.rs:128-137): one thread per member of a block whose majority edits are substitutions only ----
struct RcRecon { uint32_t n_subs, n_dels, n_msubs; uint64_t sub_off, del_off, msub_off, out_off; };    // out: n_subs + n_msubs slots
struct RcMsub { uint32_t pos, alt, orig; };
__global__ void k_rc_reconcile(const RcRecon *__restrict__ jobs, int64_t n, const pga_sub_t *__restrict__ subs, const pga_del_t *__restrict__ dels, const RcMsub *__restrict__ msubs, pga_sub_t *__restrict__ out,
                               uint32_t *__restrict__ out_n, int32_t *__restrict__ status)
{
	const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (m >= n) return;
	const RcRecon J = jobs[m];
	pga_sub_t *L = out + J.out_off;
	uint32_t cnt = J.n_subs; bool sorted = false; int32_t st = 0;
	for (uint32_t t = 0; t < cnt; ++t) L[t] = subs[J.sub_off + t];
	auto is_deleted = [&](uint32_t pos) { for (uint32_t t = 0; t < J.n_dels; ++t) { const pga_del_t d = dels[J.del_off + t]; if (pos >= d.pos && pos < d.pos + d.len) return true; } return false; };
	for (uint32_t u = 0; u < J.n_msubs && st == 0; ++u) {
		const RcMsub S = msubs[J.msub_off + u];
		uint32_t at = 0; for (uint32_t t = 0; t < cnt; ++t) at += L[t].pos == S.pos;
		if (at == 0) {                                                   // edits.rs:206-209: reversion to the original letter unless the position is deleted
			if (!is_deleted(S.pos)) {
				L[cnt].pos = S.pos; L[cnt].alt = S.orig; ++cnt;
				// subs.sort_by_key(pos): a stable sort of the whole list (edits.rs:165)
				if (!sorted) { for (uint32_t i = 1; i < cnt; ++i) { const pga_sub_t v = L[i]; uint32_t j = i; while (j > 0 && L[j - 1].pos > v.pos) { L[j] = L[j - 1]; --j; } L[j] = v; } sorted = true; }
				else { const pga_sub_t v = L[cnt - 1]; uint32_t j = cnt - 1; while (j > 0 && L[j - 1].pos > v.pos) { L[j] = L[j - 1]; --j; } L[j] = v; }
			}
		} else if (at == 1) {
			if (is_deleted(S.pos)) { st = 5; break; }                      // edits.rs:214-220: a substitution and a deletion at one position
			bool match = false; for (uint32_t t = 0; t < cnt; ++t) if (L[t].pos == S.pos) { match = L[t].alt == S.alt; break; }
			if (match) { uint32_t w = 0; for (uint32_t t = 0; t < cnt; ++t) if (!(L[t].pos == S.pos && L[t].alt == S.alt)) L[w++] = L[t]; cnt = w; }
		} else st = 6;                                                     // edits.rs:224-235: sequence states disagree
	}
	out_n[m] = cnt; status[m] = st;
}

// ---------------------------------------------------------------- host side
static int64_t aligned_count_after(const std::vector<pga_del_t> &dels, uint32_t p, uint32_t cons_len)     // edits.rs:418-440
{
	const int64_t total = cons_len > p ? (int64_t)cons_len - p : 0;
	int64_t overlap = 0;
	for (const pga_del_t &d : dels) if ((uint64_t)d.pos + d.len > p) overlap += (int64_t)((uint64_t)d.pos + d.len) - std::max<int64_t>(p, d.pos);
	return std::max<int64_t>(total - overlap, 0);
}
// BandParameters::from_edits (map_variations.rs:29-37) = (Edit::aln_mean_shift, Edit::aln_bandwidth), edits.rs:442-531; false: no aligned position
static bool band_from_edits(const std::vector<pga_del_t> &dels, const std::vector<std::pair<uint32_t, uint32_t>> &inss /* (pos, len) in list order */, uint32_t cons_len, int64_t &ms, int64_t &bw)
{
	const int64_t ac = aligned_count_after(dels, 0, cons_len);
	if (ac == 0) return false;
	int64_t total = 0;
	for (auto &x : inss) total -= (int64_t)x.second * aligned_count_after(dels, x.first, cons_len);
	for (const pga_del_t &d : dels) total += (int64_t)d.len * aligned_count_after(dels, d.pos, cons_len);
	ms = (int64_t)std::llround((double)total / (double)ac);              // f64::round: half away from zero
	std::vector<std::pair<uint32_t, int64_t>> tp;
	for (auto &x : inss) tp.emplace_back(x.first, -(int64_t)x.second);
	for (const pga_del_t &d : dels) tp.emplace_back(d.pos, (int64_t)d.len);
	std::stable_sort(tp.begin(), tp.end(), [](const std::pair<uint32_t, int64_t> &a, const std::pair<uint32_t, int64_t> &b) { return a.first < b.first; });
	bw = 0; int64_t cur = 0;
	for (size_t i = 0; i < tp.size(); ++i) {
		if (i == 0 && tp[i].first > 0) bw = std::max<int64_t>(bw, std::llabs(cur - ms));
		cur += tp[i].second;
		if (i + 1 == tp.size() && (tp[i].first == cons_len || (tp[i].second > 0 && (int64_t)tp[i].first + tp[i].second == (int64_t)cons_len))) continue;
		bw = std::max<int64_t>(bw, std::llabs(cur - ms));
	}
	return true;
}

template <class T> static T *dup_pool(const std::vector<T> &v) { T *p = (T*)malloc((v.size() ? v.size() : 1) * sizeof(T)); if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T)); return p; }

// prepared lists of one edit (for k_rc_apply); returns the length of the applied sequence
struct PreparedEdit { std::vector<pga_sub_t> subs; std::vector<RcDelIv> dels; std::vector<RcInsP> inss; };
static uint32_t prepare_edit(const pga_sub_t *subs, uint32_t n_subs, const pga_del_t *dels, uint32_t n_dels, const pga_ins_t *inss, uint32_t n_inss, const char *ins_seq, uint32_t cons_len, PreparedEdit &P)
{
	P.subs.assign(subs, subs + n_subs);
	std::stable_sort(P.subs.begin(), P.subs.end(), [](const pga_sub_t &a, const pga_sub_t &b) { return a.pos < b.pos; });
	std::vector<std::pair<uint32_t, uint32_t>> iv;
	for (uint32_t t = 0; t < n_dels; ++t) if (dels[t].len) iv.emplace_back(dels[t].pos, dels[t].pos + dels[t].len);
	std::sort(iv.begin(), iv.end());
	P.dels.clear();
	uint32_t before = 0;
	for (auto &x : iv) {
		if (!P.dels.empty() && x.first <= P.dels.back().end) { if (x.second > P.dels.back().end) { before += x.second - P.dels.back().end; P.dels.back().end = x.second; } continue; }
		P.dels.push_back(RcDelIv{x.first, x.second, before});
		before += x.second - x.first;
	}
	std::vector<uint32_t> ord(n_inss); std::iota(ord.begin(), ord.end(), 0u);
	std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {       // Ins: Ord by (pos, seq), edits.rs:321 `sorted()`
		if (inss[a].pos != inss[b].pos) return inss[a].pos < inss[b].pos;
		const uint32_t la = inss[a].len, lb = inss[b].len; const int c = memcmp(ins_seq + inss[a].seq_off, ins_seq + inss[b].seq_off, std::min(la, lb));
		return c != 0 ? c < 0 : la < lb; });
	P.inss.clear();
	uint32_t ib = 0;
	for (uint32_t o : ord) { P.inss.push_back(RcInsP{inss[o].pos, inss[o].len, ib, inss[o].seq_off}); ib += inss[o].len; }
	return cons_len - before + ib;
}

void reconsensus_host(int64_t n_blocks, const pga_rc_block_t *blocks, const pga_rc_member_t *members, const pga_sub_t *subs, const pga_del_t *dels, const pga_ins_t *inss, const char *ins_seq,
                      const pga_mapvar_params_t &prm, pga_rc_out_t *out)
{
	memset(out, 0, sizeof(*out));
	hipStream_t st = 0;
	if (n_blocks <= 0) return;
	if (n_blocks >= (1 << 24)) throw std::runtime_error("pga_reconsensus: more than 2^24 blocks in one call");
	// ---- offsets ----
	std::vector<uint64_t> mem_first((size_t)n_blocks + 1, 0), cov_base((size_t)n_blocks + 1, 0), cons_off((size_t)n_blocks + 1, 0);
	std::vector<uint32_t> depth((size_t)n_blocks);
	for (int64_t b = 0; b < n_blocks; ++b) {
		mem_first[b + 1] = mem_first[b] + blocks[b].n_members; depth[b] = blocks[b].n_members;
		cov_base[b + 1] = cov_base[b] + blocks[b].cons_len + 1; cons_off[b + 1] = cons_off[b] + blocks[b].cons_len;
		if (blocks[b].cons_len && !blocks[b].consensus) throw std::runtime_error("pga_reconsensus: null consensus");
	}
	const int64_t n_mem = (int64_t)mem_first[n_blocks];
	std::vector<RcMember> mem((size_t)n_mem);
	uint64_t so = 0, dof = 0, io = 0, n_ib = 0;
	std::vector<uint32_t> ins_blk;                                          // block of every input insertion
	for (int64_t b = 0; b < n_blocks; ++b) for (uint64_t m = mem_first[b]; m < mem_first[b + 1]; ++m) {
		RcMember &M = mem[m]; M.block = (uint32_t)b; M.n_subs = members[m].n_subs; M.n_dels = members[m].n_dels; M.n_inss = members[m].n_inss;
		M.sub_off = so; M.del_off = dof; M.ins_off = io; so += M.n_subs; dof += M.n_dels; io += M.n_inss;
		for (uint32_t t = 0; t < M.n_subs; ++t) if (subs[M.sub_off + t].pos >= blocks[b].cons_len) throw std::runtime_error("pga_reconsensus: substitution beyond the consensus");
		for (uint32_t t = 0; t < M.n_dels; ++t) if ((uint64_t)dels[M.del_off + t].pos + dels[M.del_off + t].len > blocks[b].cons_len) throw std::runtime_error("pga_reconsensus: deletion beyond the consensus");
		for (uint32_t t = 0; t < M.n_inss; ++t) { const pga_ins_t &x = inss[M.ins_off + t]; if (x.pos > blocks[b].cons_len) throw std::runtime_error("pga_reconsensus: insertion beyond the consensus"); n_ib = std::max<uint64_t>(n_ib, x.seq_off + x.len); ins_blk.push_back((uint32_t)b); }
	}
	const uint64_t n_subs = so, n_dels = dof, n_inss = io, n_cov = cov_base[n_blocks];
	// ---- upload ----
	DBuf<RcMember> d_mem; d_mem.upload(mem, st);
	DBuf<pga_sub_t> d_subs(n_subs + 1); DBuf<pga_del_t> d_dels(n_dels + 1); DBuf<pga_ins_t> d_inss(n_inss + 1); DBuf<char> d_iseq(n_ib + 1);
	if (n_subs) PGA_HIP(hipMemcpyAsync(d_subs.p, subs, n_subs * sizeof(pga_sub_t), hipMemcpyHostToDevice, st));
	if (n_dels) PGA_HIP(hipMemcpyAsync(d_dels.p, dels, n_dels * sizeof(pga_del_t), hipMemcpyHostToDevice, st));
	if (n_inss) PGA_HIP(hipMemcpyAsync(d_inss.p, inss, n_inss * sizeof(pga_ins_t), hipMemcpyHostToDevice, st));
	if (n_ib) PGA_HIP(hipMemcpyAsync(d_iseq.p, ins_seq, n_ib, hipMemcpyHostToDevice, st));
	DBuf<uint64_t> d_cov_base; d_cov_base.upload(cov_base, st);
	DBuf<uint32_t> d_depth; d_depth.upload(depth, st);
	DBuf<char> d_cons(cons_off[n_blocks] + 1);
	{
		std::vector<char> cat(cons_off[n_blocks] + 1);
		for (int64_t b = 0; b < n_blocks; ++b) if (blocks[b].cons_len) memcpy(cat.data() + cons_off[b], blocks[b].consensus, blocks[b].cons_len);
		PGA_HIP(hipMemcpyAsync(d_cons.p, cat.data(), cons_off[n_blocks], hipMemcpyHostToDevice, st));
		PGA_HIP(sync_stream(st));
	}
	// ---- majority edits ----
	DBuf<uint64_t> sub_key(n_subs + 1), sub_key2(n_subs + 1), ins_key(n_inss + 1), ins_hash(n_inss + 1);
	DBuf<int32_t> cov(n_cov + 1); cov.zero(st);
	if (n_mem) hipLaunchKernelGGL(k_rc_keys, dim3((unsigned)((n_mem + 127) / 128)), dim3(128), 0, st, d_mem.p, n_mem, d_subs.p, d_dels.p, d_inss.p, d_iseq.p, d_cov_base.p, sub_key.p, cov.p, ins_key.p, ins_hash.p);
	std::vector<uint64_t> maj_sub_keys; std::vector<uint64_t> del_starts, del_ends; std::vector<uint32_t> maj_ins_idx;
	auto scan_u32 = [&](const uint32_t *in, uint32_t *o, uint64_t n) { size_t tb = 0; PGA_HIP(rocprim::exclusive_scan(nullptr, tb, in, o, 0u, n, rocprim::plus<uint32_t>(), st)); DBuf<uint8_t> t(tb ? tb : 1); PGA_HIP(rocprim::exclusive_scan(t.p, tb, in, o, 0u, n, rocprim::plus<uint32_t>(), st)); };
	if (n_subs) {
		size_t tb = 0;
		PGA_HIP(rocprim::radix_sort_keys(nullptr, tb, sub_key.p, sub_key2.p, n_subs, 0, 64, st));
		DBuf<uint8_t> tmp(tb ? tb : 1);
		PGA_HIP(rocprim::radix_sort_keys(tmp.p, tb, sub_key.p, sub_key2.p, n_subs, 0, 64, st));
		DBuf<uint32_t> flag(n_subs + 1), pos(n_subs + 1); flag.zero(st);
		hipLaunchKernelGGL(k_rc_sub_major, dim3((unsigned)((n_subs + 255) / 256)), dim3(256), 0, st, sub_key2.p, (int64_t)n_subs, d_depth.p, flag.p);
		scan_u32(flag.p, pos.p, n_subs + 1);
		uint32_t n_maj = 0; PGA_HIP(hipMemcpyAsync(&n_maj, pos.p + n_subs, 4, hipMemcpyDeviceToHost, st)); PGA_HIP(sync_stream(st));
		if (n_maj) { DBuf<uint64_t> o(n_maj); hipLaunchKernelGGL(k_rc_compact<uint64_t>, dim3((unsigned)((n_subs + 255) / 256)), dim3(256), 0, st, flag.p, pos.p, n_subs, sub_key2.p, o.p); maj_sub_keys = o.download(st); }
	}
	if (n_dels) {
		DBuf<int32_t> covs(n_cov + 1);
		size_t tb = 0;
		PGA_HIP(rocprim::inclusive_scan(nullptr, tb, cov.p, covs.p, n_cov, rocprim::plus<int32_t>(), st));
		DBuf<uint8_t> tmp(tb ? tb : 1);
		PGA_HIP(rocprim::inclusive_scan(tmp.p, tb, cov.p, covs.p, n_cov, rocprim::plus<int32_t>(), st));
		DBuf<uint32_t> fs(n_cov + 1), fe(n_cov + 1), ps(n_cov + 1), pe(n_cov + 1); fs.zero(st); fe.zero(st);
		hipLaunchKernelGGL(k_rc_del_flags, dim3((unsigned)((n_cov + 255) / 256)), dim3(256), 0, st, covs.p, n_cov, (const uint32_t*)nullptr, d_cov_base.p, d_depth.p, (int)n_blocks, fs.p, fe.p);
		scan_u32(fs.p, ps.p, n_cov + 1); scan_u32(fe.p, pe.p, n_cov + 1);
		uint32_t ns = 0, ne = 0; PGA_HIP(hipMemcpyAsync(&ns, ps.p + n_cov, 4, hipMemcpyDeviceToHost, st)); PGA_HIP(hipMemcpyAsync(&ne, pe.p + n_cov, 4, hipMemcpyDeviceToHost, st)); PGA_HIP(sync_stream(st));
		if (ns != ne) throw std::runtime_error("pga_reconsensus: interval starts and ends disagree");
		if (ns) {
			DBuf<uint64_t> os(ns), oe(ns);
			hipLaunchKernelGGL(k_rc_compact<uint64_t>, dim3((unsigned)((n_cov + 255) / 256)), dim3(256), 0, st, fs.p, ps.p, n_cov, (const uint64_t*)nullptr, os.p);
			hipLaunchKernelGGL(k_rc_compact<uint64_t>, dim3((unsigned)((n_cov + 255) / 256)), dim3(256), 0, st, fe.p, pe.p, n_cov, (const uint64_t*)nullptr, oe.p);
			del_starts = os.download(st); del_ends = oe.download(st);
		}
	}
	bool ins_exact_on_host = false;
	if (n_inss) {
		// sort by hash, then stably by (block, position): equal (block, position, letters) end up adjacent
		DBuf<uint64_t> h2(n_inss), k1(n_inss), k2(n_inss), hs(n_inss); DBuf<uint32_t> i0(n_inss), i1(n_inss), i2(n_inss);
		{ std::vector<uint32_t> id(n_inss); std::iota(id.begin(), id.end(), 0u); i0.upload(id, st); }
		size_t tb = 0;
		PGA_HIP(rocprim::radix_sort_pairs(nullptr, tb, ins_hash.p, h2.p, i0.p, i1.p, n_inss, 0, 64, st));
		DBuf<uint8_t> tmp(tb ? tb : 1);
		PGA_HIP(rocprim::radix_sort_pairs(tmp.p, tb, ins_hash.p, h2.p, i0.p, i1.p, n_inss, 0, 64, st));
		{ struct G { const uint64_t *k; const uint32_t *i; }; G g{ins_key.p, i1.p};
		  PGA_HIP(rocprim::transform(rocprim::make_counting_iterator<uint64_t>(0), k1.p, n_inss, [g] __device__ (uint64_t j) { return g.k[g.i[j]]; }, st)); }
		PGA_HIP(rocprim::radix_sort_pairs(nullptr, tb, k1.p, k2.p, i1.p, i2.p, n_inss, 0, 56, st));
		DBuf<uint8_t> tmp2(tb ? tb : 1);
		PGA_HIP(rocprim::radix_sort_pairs(tmp2.p, tb, k1.p, k2.p, i1.p, i2.p, n_inss, 0, 56, st));
		{ struct G { const uint64_t *h; const uint32_t *i; }; G g{ins_hash.p, i2.p};
		  PGA_HIP(rocprim::transform(rocprim::make_counting_iterator<uint64_t>(0), hs.p, n_inss, [g] __device__ (uint64_t j) { return g.h[g.i[j]]; }, st)); }
		DBuf<uint32_t> flag(n_inss + 1), pos(n_inss + 1); flag.zero(st);
		DBuf<int> coll(1); coll.zero(st);
		hipLaunchKernelGGL(k_rc_ins_major, dim3((unsigned)((n_inss + 255) / 256)), dim3(256), 0, st, k2.p, hs.p, i2.p, (int64_t)n_inss, d_inss.p, d_iseq.p, d_depth.p, flag.p, coll.p);
		scan_u32(flag.p, pos.p, n_inss + 1);
		uint32_t n_maj = 0; PGA_HIP(hipMemcpyAsync(&n_maj, pos.p + n_inss, 4, hipMemcpyDeviceToHost, st)); PGA_HIP(sync_stream(st));
		if (coll.download(st)[0]) ins_exact_on_host = true;
		else if (n_maj) { DBuf<uint32_t> o(n_maj); hipLaunchKernelGGL(k_rc_compact<uint32_t>, dim3((unsigned)((n_inss + 255) / 256)), dim3(256), 0, st, flag.p, pos.p, n_inss, i2.p, o.p); maj_ins_idx = o.download(st); }
	}
	// ---- per block: the majority edit (insertions ascending by position, deletions ascending, substitutions ascending) ----
	struct Maj { std::vector<pga_sub_t> subs; std::vector<pga_del_t> dels; std::vector<uint32_t> inss; };      // inss: indices into the input pool
	std::vector<Maj> maj((size_t)n_blocks);
	for (uint64_t k : maj_sub_keys) maj[k >> 40].subs.push_back(pga_sub_t{(uint32_t)(k >> 8), (uint32_t)(k & 255)});
	{
		size_t b = 0;
		for (size_t i = 0; i < del_starts.size(); ++i) { while (cov_base[b + 1] <= del_starts[i]) ++b; maj[b].dels.push_back(pga_del_t{(uint32_t)(del_starts[i] - cov_base[b]), (uint32_t)(del_ends[i] + 1 - del_starts[i])}); }
	}
	if (ins_exact_on_host) {
		// two different insertions of one block and position share a 64-bit hash: counted exactly here (block bookkeeping, no base work)
		for (int64_t b = 0; b < n_blocks; ++b) {
			std::map<std::pair<uint32_t, std::string>, std::pair<uint32_t, uint32_t>> cnt;
			for (uint64_t m = mem_first[b]; m < mem_first[b + 1]; ++m) for (uint32_t t = 0; t < mem[m].n_inss; ++t) {
				const pga_ins_t &x = inss[mem[m].ins_off + t];
				auto &e = cnt[{x.pos, std::string(ins_seq + x.seq_off, x.len)}];
				if (e.first++ == 0) e.second = (uint32_t)(mem[m].ins_off + t);
			}
			for (auto &kv : cnt) if (kv.second.first > depth[b] / 2) maj[b].inss.push_back(kv.second.second);
		}
	} else for (uint32_t idx : maj_ins_idx) maj[ins_blk[idx]].inss.push_back(idx);
	for (Maj &M : maj) std::stable_sort(M.inss.begin(), M.inss.end(), [&](uint32_t a, uint32_t b) { return inss[a].pos < inss[b].pos; });
	// ---- result skeleton ----
	std::vector<pga_rc_block_res_t> R((size_t)n_blocks);
	std::vector<pga_mapvar_res_t> MR((size_t)n_mem);
	memset(MR.data(), 0, MR.size() * sizeof(pga_mapvar_res_t));
	std::vector<pga_sub_t> o_subs, m_subs; std::vector<pga_del_t> o_dels, m_dels; std::vector<pga_ins_t> o_inss, m_inss; std::vector<char> o_iseq, m_iseq, o_cons;
	std::vector<int64_t> blocks1, blocks2;
	for (int64_t b = 0; b < n_blocks; ++b) {
		pga_rc_block_res_t &r = R[b]; memset(&r, 0, sizeof(r));
		r.n_subs = (uint32_t)maj[b].subs.size(); r.n_dels = (uint32_t)maj[b].dels.size(); r.n_inss = (uint32_t)maj[b].inss.size();
		r.sub_off = m_subs.size(); r.del_off = m_dels.size(); r.ins_off = m_inss.size();
		m_subs.insert(m_subs.end(), maj[b].subs.begin(), maj[b].subs.end());
		m_dels.insert(m_dels.end(), maj[b].dels.begin(), maj[b].dels.end());
		for (uint32_t idx : maj[b].inss) { const pga_ins_t &x = inss[idx]; m_inss.push_back(pga_ins_t{x.pos, x.len, (uint64_t)m_iseq.size()}); m_iseq.insert(m_iseq.end(), ins_seq + x.seq_off, ins_seq + x.seq_off + x.len); }
		r.kind = (r.n_dels || r.n_inss) ? 2 : r.n_subs ? 1 : 0;
		if (blocks[b].n_members == 0) r.kind = 0;
		if (r.kind == 1) blocks1.push_back(b); else if (r.kind == 2) blocks2.push_back(b);
	}
	// per-member output slots are filled below, in member order, from three sources
	struct MemOut { std::vector<pga_sub_t> subs; std::vector<pga_del_t> dels; std::vector<pga_ins_t> inss; std::vector<char> iseq; };
	// ---- kind 1: consensus letters change, members are reconciled ----
	std::vector<RcRecon> rj; std::vector<RcMsub> rms; std::vector<uint64_t> rj_member;
	uint64_t r_out = 0;
	for (int64_t b : blocks1) {
		const uint64_t mo = rms.size();
		for (const pga_sub_t &s : maj[b].subs) {
			const uint32_t orig = (uint8_t)blocks[b].consensus[s.pos];
			if (orig == s.alt) { R[b].kind = -2; break; }                     // pangraph_block.rs:272-278: the letter is that letter already
			rms.push_back(RcMsub{s.pos, s.alt, orig});
		}
		if (R[b].kind < 0) { rms.resize(mo); continue; }
		for (uint64_t m = mem_first[b]; m < mem_first[b + 1]; ++m) { rj.push_back(RcRecon{mem[m].n_subs, mem[m].n_dels, R[b].n_subs, mem[m].sub_off, mem[m].del_off, mo, r_out}); rj_member.push_back(m); r_out += mem[m].n_subs + R[b].n_subs; }
	}
	std::vector<pga_sub_t> r_subs; std::vector<uint32_t> r_n; std::vector<int32_t> r_st;
	if (!rj.empty()) {
		DBuf<RcRecon> d_rj; d_rj.upload(rj, st); DBuf<RcMsub> d_rms; d_rms.upload(rms, st);
		DBuf<pga_sub_t> d_o(r_out + 1); DBuf<uint32_t> d_n(rj.size()); DBuf<int32_t> d_s(rj.size());
		hipLaunchKernelGGL(k_rc_reconcile, dim3((unsigned)((rj.size() + 63) / 64)), dim3(64), 0, st, d_rj.p, (int64_t)rj.size(), d_subs.p, d_dels.p, d_rms.p, d_o.p, d_n.p, d_s.p);
		r_subs = d_o.download(st); r_n = d_n.download(st); r_st = d_s.download(st);
	}
	// ---- kind 2: Edit::apply on the device (new consensus, every member's sequence), then the re-alignment ----
	std::vector<RcApply> aj; std::vector<pga_sub_t> a_subs; std::vector<RcDelIv> a_dels; std::vector<RcInsP> a_inss; std::vector<uint64_t> a_first;
	std::vector<char> a_iseq;                                               // the letters of the majority insertions (member insertions point into the input pool)
	std::vector<MvDevJob> mvj; std::vector<uint64_t> mvj_member;
	uint64_t a_out = 0, a_threads = 0;
	const uint64_t maj_seq_base = n_ib;                                      // majority letters are appended behind the input letters on the device
	auto add_apply = [&](int64_t b, const PreparedEdit &P, uint32_t out_len) -> uint64_t {
		RcApply J; J.src_off = cons_off[b]; J.len = blocks[b].cons_len; J.out_off = a_out; J.n_subs = (uint32_t)P.subs.size(); J.n_dels = (uint32_t)P.dels.size(); J.n_inss = (uint32_t)P.inss.size();
		J.sub_off = a_subs.size(); J.del_off = a_dels.size(); J.ins_off = a_inss.size();
		a_subs.insert(a_subs.end(), P.subs.begin(), P.subs.end()); a_dels.insert(a_dels.end(), P.dels.begin(), P.dels.end()); a_inss.insert(a_inss.end(), P.inss.begin(), P.inss.end());
		a_first.push_back(a_threads); a_threads += (uint64_t)J.len + 1;
		aj.push_back(J);
		const uint64_t o = a_out; a_out += out_len; return o;
	};
	struct NewCons { uint64_t off; uint32_t len; };
	std::vector<NewCons> nc((size_t)n_blocks, NewCons{0, 0});
	for (int64_t b : blocks2) {
		// the majority edit as lists (its insertion letters live in m_iseq; on the device behind the input letters)
		std::vector<pga_ins_t> mi; for (uint32_t t = 0; t < R[b].n_inss; ++t) { pga_ins_t x = m_inss[R[b].ins_off + t]; mi.push_back(x); }
		PreparedEdit P;
		const uint32_t new_len = prepare_edit(maj[b].subs.data(), R[b].n_subs, maj[b].dels.data(), R[b].n_dels, mi.data(), (uint32_t)mi.size(), m_iseq.data(), blocks[b].cons_len, P);
		for (RcInsP &x : P.inss) x.seq_off += maj_seq_base;
		if (new_len == 0) { R[b].kind = -3; continue; }                       // pangraph_block.rs:298: the consensus cannot be empty
		nc[b] = NewCons{add_apply(b, P, new_len), new_len};
		int64_t bms = 0, bbw = 0;
		std::vector<std::pair<uint32_t, uint32_t>> il; for (const pga_ins_t &x : mi) il.emplace_back(x.pos, x.len);
		if (!band_from_edits(maj[b].dels, il, blocks[b].cons_len, bms, bbw)) { R[b].kind = -4; continue; }   // (the majority edit leaves no aligned position; its apply job runs for nothing)
		for (uint64_t m = mem_first[b]; m < mem_first[b + 1]; ++m) {
			PreparedEdit Q;
			const uint32_t qlen = prepare_edit(subs + mem[m].sub_off, mem[m].n_subs, dels + mem[m].del_off, mem[m].n_dels, inss + mem[m].ins_off, mem[m].n_inss, ins_seq, blocks[b].cons_len, Q);
			const uint64_t qoff = add_apply(b, Q, qlen);
			std::vector<pga_del_t> dl(dels + mem[m].del_off, dels + mem[m].del_off + mem[m].n_dels);
			std::vector<std::pair<uint32_t, uint32_t>> ml; for (uint32_t t = 0; t < mem[m].n_inss; ++t) ml.emplace_back(inss[mem[m].ins_off + t].pos, inss[mem[m].ins_off + t].len);
			int64_t oms = 0, obw = 0;
			if (!band_from_edits(dl, ml, blocks[b].cons_len, oms, obw)) { MR[m].status = 7; continue; }   // map_variations.rs:32: no aligned position (the reference returns an error)
			mvj.push_back(MvDevJob{nc[b].off, qoff, new_len, qlen, (int32_t)(oms - bms), (uint32_t)(obw + bbw)});
			mvj_member.push_back(m);
		}
	}
	std::vector<pga_sub_t> v_subs; std::vector<pga_del_t> v_dels; std::vector<pga_ins_t> v_inss; std::vector<char> v_iseq;
	std::vector<pga_mapvar_res_t> v_res(mvj.size());
	std::vector<char> new_cons_all;
	if (!aj.empty()) {
		DBuf<char> d_iseq2(n_ib + m_iseq.size() + 1);
		if (n_ib) PGA_HIP(hipMemcpyAsync(d_iseq2.p, d_iseq.p, n_ib, hipMemcpyDeviceToDevice, st));
		if (!m_iseq.empty()) PGA_HIP(hipMemcpyAsync(d_iseq2.p + n_ib, m_iseq.data(), m_iseq.size(), hipMemcpyHostToDevice, st));
		DBuf<RcApply> d_aj; d_aj.upload(aj, st); DBuf<uint64_t> d_first; d_first.upload(a_first, st);
		DBuf<pga_sub_t> d_as(a_subs.size() + 1); DBuf<RcDelIv> d_ad(a_dels.size() + 1); DBuf<RcInsP> d_ai(a_inss.size() + 1);
		if (!a_subs.empty()) PGA_HIP(hipMemcpyAsync(d_as.p, a_subs.data(), a_subs.size() * sizeof(pga_sub_t), hipMemcpyHostToDevice, st));
		if (!a_dels.empty()) PGA_HIP(hipMemcpyAsync(d_ad.p, a_dels.data(), a_dels.size() * sizeof(RcDelIv), hipMemcpyHostToDevice, st));
		if (!a_inss.empty()) PGA_HIP(hipMemcpyAsync(d_ai.p, a_inss.data(), a_inss.size() * sizeof(RcInsP), hipMemcpyHostToDevice, st));
		DBuf<char> d_out(a_out + 64);
		hipLaunchKernelGGL(k_rc_apply, dim3((unsigned)((a_threads + 255) / 256)), dim3(256), 0, st, d_aj.p, d_first.p, (int)aj.size(), a_threads, d_cons.p, d_as.p, d_ad.p, d_ai.p, d_iseq2.p, d_out.p);
		PGA_HIP(hipGetLastError());
		// the new consensus sequences go back to the caller; the member sequences only feed the aligner
		new_cons_all.resize(a_out);
		for (int64_t b : blocks2) if (R[b].kind == 2) PGA_HIP(hipMemcpyAsync(new_cons_all.data() + nc[b].off, d_out.p + nc[b].off, nc[b].len, hipMemcpyDeviceToHost, st));
		if (!mvj.empty()) { memset(v_res.data(), 0, v_res.size() * sizeof(pga_mapvar_res_t)); map_variations_dev((int64_t)mvj.size(), mvj.data(), d_out.p, a_out, prm, v_res.data(), v_subs, v_dels, v_inss, v_iseq, st); }
		PGA_HIP(sync_stream(st));
	}
	// ---- pack ----
	std::vector<size_t> rj_of_member((size_t)n_mem, (size_t)-1), mv_of_member((size_t)n_mem, (size_t)-1);
	for (size_t i = 0; i < rj_member.size(); ++i) rj_of_member[rj_member[i]] = i;
	for (size_t i = 0; i < mvj_member.size(); ++i) mv_of_member[mvj_member[i]] = i;
	for (int64_t b = 0; b < n_blocks; ++b) {
		pga_rc_block_res_t &r = R[b];
		// a block whose reconciliation failed for ANY member is an error as a whole (the reference returns Err for the call): decided before
		// anything of the block is packed, so that it comes back with its ORIGINAL consensus and the original edits of every member
		if (r.kind == 1)
			for (uint64_t m = mem_first[b]; m < mem_first[b + 1] && r.kind == 1; ++m)
				if (rj_of_member[m] != (size_t)-1 && r_st[rj_of_member[m]] != 0) r.kind = -5;
		r.cons_off = o_cons.size();
		if (r.kind == 2) { r.cons_len = nc[b].len; o_cons.insert(o_cons.end(), new_cons_all.begin() + nc[b].off, new_cons_all.begin() + nc[b].off + nc[b].len); }
		else {
			r.cons_len = blocks[b].cons_len; o_cons.insert(o_cons.end(), blocks[b].consensus, blocks[b].consensus + blocks[b].cons_len);
			if (r.kind == 1) for (const pga_sub_t &s : maj[b].subs) o_cons[r.cons_off + s.pos] = (char)s.alt;
		}
		for (uint64_t m = mem_first[b]; m < mem_first[b + 1]; ++m) {
			pga_mapvar_res_t &o = MR[m];
			o.sub_off = o_subs.size(); o.del_off = o_dels.size(); o.ins_off = o_inss.size();
			auto copy_in = [&](bool with_subs) {
				if (with_subs) { o.n_subs = mem[m].n_subs; o_subs.insert(o_subs.end(), subs + mem[m].sub_off, subs + mem[m].sub_off + mem[m].n_subs); }
				o.n_dels = mem[m].n_dels; o_dels.insert(o_dels.end(), dels + mem[m].del_off, dels + mem[m].del_off + mem[m].n_dels);
				o.n_inss = mem[m].n_inss;
				for (uint32_t t = 0; t < mem[m].n_inss; ++t) { const pga_ins_t &x = inss[mem[m].ins_off + t]; o_inss.push_back(pga_ins_t{x.pos, x.len, (uint64_t)o_iseq.size()}); o_iseq.insert(o_iseq.end(), ins_seq + x.seq_off, ins_seq + x.seq_off + x.len); o.n_ins_bases += x.len; }
			};
			if (r.kind == 1 && rj_of_member[m] != (size_t)-1) {
				const size_t i = rj_of_member[m];
				o.status = r_st[i];
				o.n_subs = r_n[i]; o_subs.insert(o_subs.end(), r_subs.begin() + rj[i].out_off, r_subs.begin() + rj[i].out_off + r_n[i]);
				copy_in(false);
			} else if (r.kind == 2 && mv_of_member[m] != (size_t)-1) {
				const size_t i = mv_of_member[m];
				const pga_mapvar_res_t &v = v_res[i];
				o.status = v.status; o.score = v.score; o.attempts = v.attempts; o.hit_boundary = v.hit_boundary;
				o.n_subs = v.n_subs; o.n_dels = v.n_dels; o.n_inss = v.n_inss; o.n_ins_bases = v.n_ins_bases;
				o_subs.insert(o_subs.end(), v_subs.begin() + v.sub_off, v_subs.begin() + v.sub_off + v.n_subs);
				o_dels.insert(o_dels.end(), v_dels.begin() + v.del_off, v_dels.begin() + v.del_off + v.n_dels);
				for (uint32_t t = 0; t < v.n_inss; ++t) { const pga_ins_t &x = v_inss[v.ins_off + t]; o_inss.push_back(pga_ins_t{x.pos, x.len, (uint64_t)o_iseq.size()}); o_iseq.insert(o_iseq.end(), v_iseq.begin() + x.seq_off, v_iseq.begin() + x.seq_off + x.len); }
			} else if (r.kind == 2) { /* status already set: the member has no aligned position */ }
			else { if (r.kind == -5 && rj_of_member[m] != (size_t)-1) o.status = r_st[rj_of_member[m]]; copy_in(true); }
		}
	}
	out->blocks = dup_pool(R); out->members = dup_pool(MR);
	out->subs = dup_pool(o_subs); out->dels = dup_pool(o_dels); out->inss = dup_pool(o_inss); out->ins_seq = dup_pool(o_iseq);
	out->m_subs = dup_pool(m_subs); out->m_dels = dup_pool(m_dels); out->m_inss = dup_pool(m_inss); out->m_ins_seq = dup_pool(m_iseq);
	out->cons = dup_pool(o_cons);
}

} // namespace pga
