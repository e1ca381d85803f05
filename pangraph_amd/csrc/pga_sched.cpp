// pga_sched.cpp -- the ready-set schedule of a whole build behind a C-ABI (include/pga_sched.h): host bookkeeping only, no device work.
//
// The reference runs the merges of the guide tree one after the other (packages/pangraph/src/commands/build/build_run.rs:111-128), each with its
// self-merge loop of find_matches calls (packages/pangraph/src/pangraph/graph_merging.rs:26-69,95-128); the only true dependencies are
// "(v, round 0) needs the final round of both children of v" and "(v, round r) needs (v, round r - 1)".  A call becomes ready the moment its
// dependencies are done; up to `slots` batches are in flight, each made of the ready calls of that moment, largest remaining path first.
// The decisions are those of pangraph_amd/schedule.py (class ReadySet) statement by statement -- the same sort (stable, by falling priority),
// the same caps in the same arithmetic -- so that a build driven from Rust or C++ cuts the batches the measured Python host cuts.
#include "../../include/pga_sched.h"
#include "../../include/pga_align.h"
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string t_err;

// schedule.py:cost_estimate (round 4, dev/path_probe.py, every call alone on the device: a spine call of two block sets 12-48 ms, its second round
// 6-10 ms, a whole-genome pair 63 ms).  One operation per statement: the library is built with -ffp-contract=off and the sums must be Python's.
double cost_estimate(int64_t bases, int32_t n_seqs)
{
	const double big = (double)bases / (double)std::max<int32_t>(1, n_seqs);
	double c = 0.006;
	const double per_base = (double)bases * 2.5e-10;
	c = c + per_base;
	c = c + (big > 1e6 ? 0.05 : 0.0);
	return c;
}

struct Ticket { std::vector<int32_t> ids; bool express = false, live = false; };

} // namespace

struct pga_sched {
	int32_t n = 0;
	std::vector<std::vector<int32_t>> deps, users;
	std::vector<int64_t> bases;
	std::vector<int32_t> n_seqs;
	std::vector<double> prio;
	// ---- one run ----
	std::mutex m;
	std::condition_variable cv;
	std::vector<char> want, fin;
	std::vector<int32_t> indeg, ready, unfinished;
	size_t pos_top = 0;
	int64_t left = 0;
	int32_t in_flight = 0, n_express = 0, slots = 1, express = 0;
	double cap_bases = 1.2e9, min_batch_bases = 0.0, express_eps = 0.05, express_cap = 60e6;
	bool aborted = false, started = false;
	std::vector<Ticket> tickets;

	double crit_level()
	{
		while (pos_top < unfinished.size() && fin[(size_t)unfinished[pos_top]]) ++pos_top;
		return pos_top < unfinished.size() ? prio[(size_t)unfinished[pos_top]] : 0.0;
	}
	// 0 nothing may start, 1 bulk, 2 express
	int can_take()
	{
		if (ready.empty()) return 0;
		if (express && n_express < express) {
			double top = prio[(size_t)ready[0]];
			for (int32_t t : ready) top = std::max(top, prio[(size_t)t]);
			if (top >= crit_level() - express_eps) return 2;
		}
		if (in_flight - n_express < slots - express) return 1;
		return 0;
	}
	void take(int kind, std::vector<int32_t> &got)
	{
		got.clear();
		std::stable_sort(ready.begin(), ready.end(), [&](int32_t a, int32_t b) { return prio[(size_t)a] > prio[(size_t)b]; });
		if (kind == 2) {
			const double lvl = crit_level() - express_eps;
			int64_t b = 0;
			size_t k = 0;
			for (; k < ready.size(); ++k) {
				const int32_t t = ready[k];
				if (prio[(size_t)t] < lvl || (!got.empty() && (double)(b + bases[(size_t)t]) > express_cap)) break;
				got.push_back(t); b += bases[(size_t)t];
			}
			ready.erase(ready.begin(), ready.begin() + (ptrdiff_t)k);          // what was taken is a prefix of the sorted list
			++in_flight; ++n_express;
			return;
		}
		int64_t total = 0;
		for (int32_t t : ready) total += bases[(size_t)t];
		const int32_t free_slots = std::max<int32_t>(1, (slots - express) - (in_flight - n_express));
		const double share = free_slots > 1 ? (double)total / (double)free_slots : cap_bases;
		const double cap = std::max(std::max(std::min(cap_bases, share), min_batch_bases), 1.0);
		std::vector<int32_t> rest;
		int64_t b = 0;
		for (int32_t t : ready) {
			if (got.empty() || (double)(b + bases[(size_t)t]) <= cap) { got.push_back(t); b += bases[(size_t)t]; }
			else rest.push_back(t);
		}
		ready.swap(rest);
		++in_flight;
	}
	int32_t hand_out(int kind, int32_t *ids, int32_t cap_ids, int32_t *ticket)
	{
		// the size of the batch is only known after the cut: cut on a copy first when the caller's buffer might be too small
		std::vector<int32_t> got;
		if ((size_t)cap_ids < ready.size()) {
			const std::vector<int32_t> keep = ready; const int32_t f0 = in_flight, e0 = n_express; const size_t p0 = pos_top;
			take(kind, got);
			if ((size_t)cap_ids < got.size()) { ready = keep; in_flight = f0; n_express = e0; pos_top = p0; if (ticket) *ticket = (int32_t)got.size(); return -1; }
		} else take(kind, got);
		size_t k = 0;
		while (k < tickets.size() && tickets[k].live) ++k;
		if (k == tickets.size()) tickets.emplace_back();
		tickets[k].ids = got; tickets[k].express = kind == 2; tickets[k].live = true;
		std::copy(got.begin(), got.end(), ids);
		if (ticket) *ticket = (int32_t)k;
		return (int32_t)got.size();
	}
};

extern "C" {

const char *pga_sched_error(void) { return t_err.c_str(); }
double pga_sched_cost(int64_t bases, int32_t n_seqs) { return cost_estimate(bases, n_seqs); }

pga_sched_t *pga_sched_create(int32_t n_tasks, const int64_t *dep_off, const int32_t *dep, const int64_t *bases, const int32_t *n_seqs)
{
	if (n_tasks < 0 || (n_tasks > 0 && (!dep_off || !bases || !n_seqs))) { t_err = "pga_sched_create: bad arguments"; return nullptr; }
	pga_sched *s = new pga_sched;
	s->n = n_tasks;
	const size_t n = (size_t)n_tasks;
	s->deps.resize(n); s->users.resize(n); s->bases.assign(bases, bases + n); s->n_seqs.assign(n_seqs, n_seqs + n); s->prio.assign(n, 0.0);
	for (size_t i = 0; i < n; ++i) {
		if (dep_off[i + 1] < dep_off[i]) { t_err = "pga_sched_create: dep_off is not ascending"; delete s; return nullptr; }
		for (int64_t k = dep_off[i]; k < dep_off[i + 1]; ++k) {
			const int32_t d = dep[k];
			if (d < 0 || d >= n_tasks || (size_t)d == i) { t_err = "pga_sched_create: task " + std::to_string(i) + " has a bad dependency"; delete s; return nullptr; }
			s->deps[i].push_back(d);
		}
	}
	for (size_t i = 0; i < n; ++i) for (int32_t d : s->deps[i]) s->users[(size_t)d].push_back((int32_t)i);   // users in ascending task order
	// topological order (schedule.py:topo_order), then priorities against it: cost of the task + the longest remaining path of its users
	std::vector<int32_t> indeg(n), stack, order;
	for (size_t i = 0; i < n; ++i) { indeg[i] = (int32_t)s->deps[i].size(); if (indeg[i] == 0) stack.push_back((int32_t)i); }
	while (!stack.empty()) {
		const int32_t x = stack.back(); stack.pop_back(); order.push_back(x);
		for (int32_t u : s->users[(size_t)x]) if (--indeg[(size_t)u] == 0) stack.push_back(u);
	}
	if (order.size() != n) { t_err = "pga_sched_create: dependency cycle"; delete s; return nullptr; }
	for (size_t k = n; k-- > 0;) {
		const size_t t = (size_t)order[k];
		double rest = 0.0;
		for (int32_t u : s->users[t]) rest = std::max(rest, s->prio[(size_t)u]);
		s->prio[t] = cost_estimate(s->bases[t], s->n_seqs[t]) + rest;
	}
	return s;
}

void pga_sched_destroy(pga_sched_t *s) { delete s; }

void pga_sched_prio(const pga_sched_t *s, double *prio) { if (s && prio) std::copy(s->prio.begin(), s->prio.end(), prio); }

int pga_sched_start(pga_sched_t *s, const int32_t *only, int32_t n_only, const int32_t *done, int32_t n_done, int32_t slots, double cap_bases,
                    double min_batch_bases, int32_t express, double express_eps, double express_cap)
{
	if (!s) { t_err = "pga_sched_start: no scheduler"; return -1; }
	std::lock_guard<std::mutex> lk(s->m);
	const size_t n = (size_t)s->n;
	s->want.assign(n, only ? 0 : 1); s->fin.assign(n, 0); s->indeg.assign(n, 0);
	for (int32_t k = 0; only && k < n_only; ++k) { if (only[k] < 0 || only[k] >= s->n) { t_err = "pga_sched_start: task id out of range"; return -1; } s->want[(size_t)only[k]] = 1; }
	for (int32_t k = 0; done && k < n_done; ++k) { if (done[k] < 0 || done[k] >= s->n) { t_err = "pga_sched_start: task id out of range"; return -1; } s->fin[(size_t)done[k]] = 1; }
	if (slots < 1) { t_err = "pga_sched_start: slots must be at least 1 (with none pga_sched_take would wait forever)"; return -1; }
	s->ready.clear(); s->unfinished.clear(); s->tickets.clear();
	s->left = 0;
	for (size_t i = 0; i < n; ++i) {
		if (s->want[i] && s->fin[i]) s->want[i] = 0;          // a task given as done is not handed out again, whether or not `only` names it
		if (!s->want[i]) continue;
		++s->left;
		for (int32_t d : s->deps[i]) {
			if (s->fin[(size_t)d]) continue;
			if (!s->want[(size_t)d]) { t_err = "task " + std::to_string(i) + " depends on " + std::to_string(d) + ", which is neither done nor scheduled"; return -1; }
			++s->indeg[i];
		}
		if (s->indeg[i] == 0) s->ready.push_back((int32_t)i);
		if (!s->fin[i]) s->unfinished.push_back((int32_t)i);
	}
	std::stable_sort(s->unfinished.begin(), s->unfinished.end(), [&](int32_t a, int32_t b) { return s->prio[(size_t)a] > s->prio[(size_t)b]; });
	s->pos_top = 0; s->in_flight = 0; s->n_express = 0;
	s->slots = slots; s->cap_bases = cap_bases; s->min_batch_bases = min_batch_bases;
	s->express = std::max<int32_t>(0, std::min<int32_t>(express, std::max<int32_t>(0, slots - 1)));
	s->express_eps = express_eps; s->express_cap = express_cap;
	s->aborted = false; s->started = true;
	return 0;
}

int32_t pga_sched_take(pga_sched_t *s, int32_t *ids, int32_t cap_ids, int32_t *ticket)
{
	if (!s || !s->started) return 0;
	std::unique_lock<std::mutex> lk(s->m);
	int kind = s->can_take();
	while (kind == 0 && s->left > 0 && !s->aborted) { s->cv.wait(lk); kind = s->can_take(); }
	if (s->left <= 0 || s->aborted) return 0;
	return s->hand_out(kind, ids, cap_ids, ticket);
}

int32_t pga_sched_try_take(pga_sched_t *s, int32_t *ids, int32_t cap_ids, int32_t *ticket)
{
	if (!s || !s->started) return 0;
	std::lock_guard<std::mutex> lk(s->m);
	if (s->left <= 0 || s->aborted) return 0;
	const int kind = s->can_take();
	if (kind == 0) return -2;
	return s->hand_out(kind, ids, cap_ids, ticket);
}

void pga_sched_finish(pga_sched_t *s, int32_t ticket)
{
	if (!s) return;
	{
		std::lock_guard<std::mutex> lk(s->m);
		if (ticket < 0 || (size_t)ticket >= s->tickets.size() || !s->tickets[(size_t)ticket].live) return;
		Ticket &T = s->tickets[(size_t)ticket];
		--s->in_flight;
		if (T.express) --s->n_express;
		for (int32_t i : T.ids) {
			s->fin[(size_t)i] = 1;
			--s->left;
			for (int32_t u : s->users[(size_t)i]) if (s->want[(size_t)u] && --s->indeg[(size_t)u] == 0) s->ready.push_back(u);
		}
		T.live = false; T.ids.clear();
	}
	s->cv.notify_all();
}

void pga_sched_abort(pga_sched_t *s)
{
	if (!s) return;
	{ std::lock_guard<std::mutex> lk(s->m); s->aborted = true; }
	s->cv.notify_all();
}

int32_t pga_sched_left(pga_sched_t *s)
{
	if (!s) return 0;
	std::lock_guard<std::mutex> lk(s->m);
	return (int32_t)s->left;
}

int32_t pga_sched_partition(int32_t n_nodes, const int32_t *child0, const int32_t *child1, int32_t n_tasks, const int32_t *task_node,
                            const int64_t *task_bases, int32_t world, int32_t per_rank, int32_t *owner)
{
	if (n_nodes <= 0 || n_tasks < 0 || !child0 || !child1 || !owner || (n_tasks > 0 && (!task_node || !task_bases))) { t_err = "pga_sched_partition: bad arguments"; return -1; }
	if (world <= 1) { std::fill(owner, owner + n_tasks, 0); return 0; }
	const size_t nn = (size_t)n_nodes;
	for (size_t v = 0; v < nn; ++v) {
		const bool leaf = child0[v] < 0 && child1[v] < 0;
		if (!leaf && (child0[v] <= (int32_t)v || child1[v] <= (int32_t)v || child0[v] >= n_nodes || child1[v] >= n_nodes)) { t_err = "pga_sched_partition: children must have larger ids than their parent"; return -1; }
	}
	// weight of a subtree: the bases of all calls below it (integers, as in schedule.py)
	std::vector<int64_t> weight(nn, 0);
	std::vector<int64_t> own(nn, 0);
	for (int32_t t = 0; t < n_tasks; ++t) { if (task_node[t] < 0 || task_node[t] >= n_nodes) { t_err = "pga_sched_partition: task node out of range"; return -1; } own[(size_t)task_node[t]] += task_bases[t]; }
	for (size_t v = nn; v-- > 0;) {
		int64_t w = own[v];
		if (child0[v] >= 0) w += weight[(size_t)child0[v]] + weight[(size_t)child1[v]];
		weight[v] = w;
	}
	std::vector<int32_t> roots{0};
	std::vector<char> top(nn, 0);
	while ((int64_t)roots.size() < (int64_t)world * per_rank) {
		int32_t r = -1;
		for (int32_t x : roots) {
			if (child0[(size_t)x] < 0) continue;
			if (r < 0 || weight[(size_t)x] > weight[(size_t)r] || (weight[(size_t)x] == weight[(size_t)r] && x < r)) r = x;   // max by (weight, -id)
		}
		if (r < 0) break;
		roots.erase(std::find(roots.begin(), roots.end(), r));
		top[(size_t)r] = 1;
		roots.push_back(child0[(size_t)r]); roots.push_back(child1[(size_t)r]);
	}
	roots.erase(std::remove_if(roots.begin(), roots.end(), [&](int32_t r) { return child0[(size_t)r] < 0; }), roots.end());   // a bare leaf holds no merge
	std::vector<int32_t> order = roots;
	std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return weight[(size_t)a] != weight[(size_t)b] ? weight[(size_t)a] > weight[(size_t)b] : a < b; });
	std::vector<double> load((size_t)world, 0.0);
	std::vector<int32_t> owner_of_node(nn, -2), stack;
	for (int32_t r : order) {
		size_t k = 0;
		for (size_t i = 1; i < (size_t)world; ++i) if (load[i] < load[k]) k = i;                     // min by (load, rank)
		load[k] = load[k] + (double)weight[(size_t)r];
		stack.assign(1, r);
		while (!stack.empty()) {
			const int32_t x = stack.back(); stack.pop_back();
			owner_of_node[(size_t)x] = (int32_t)k;
			if (child0[(size_t)x] >= 0) { stack.push_back(child0[(size_t)x]); stack.push_back(child1[(size_t)x]); }
		}
	}
	int32_t above = 0;
	for (int32_t t = 0; t < n_tasks; ++t) {
		const size_t v = (size_t)task_node[t];
		if (top[v]) { owner[t] = -1; ++above; }
		else if (owner_of_node[v] < 0) { t_err = "pga_sched_partition: a task sits on a node outside every subtree"; return -1; }
		else owner[t] = owner_of_node[v];
	}
	return above;
}

// ---- multi-GPU: the host logic either side of the match-list gather (pangraph_amd/dist.py: shard_groups_balanced, merge_match_lists) ----
void pga_shard_groups_balanced(int32_t n_groups, const double *weights, int32_t world, int32_t *rank_of_group)
{
	if (n_groups <= 0 || !weights || !rank_of_group) return;
	if (world < 1) world = 1;
	std::vector<int32_t> order((size_t)n_groups);
	for (int32_t g = 0; g < n_groups; ++g) order[(size_t)g] = g;
	std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return weights[a] != weights[b] ? weights[a] > weights[b] : a < b; });
	std::vector<double> load((size_t)world, 0.0);
	for (int32_t g : order) {
		size_t k = 0;
		for (size_t i = 1; i < (size_t)world; ++i) if (load[i] < load[k]) k = i;
		rank_of_group[g] = (int32_t)k;
		load[k] = load[k] + weights[g];
	}
}

int pga_merge_match_lists(int32_t n_parts, const pga_match_t *const *matches, const int64_t *n_matches, const uint32_t *const *cigars,
                          const int64_t *n_cigar_words, const int32_t *const *local_to_global, const int32_t *n_local_groups,
                          pga_match_t *out_matches, uint32_t *out_cigars)
{
	if (n_parts < 0 || (n_parts > 0 && (!matches || !n_matches || !cigars || !n_cigar_words))) { t_err = "pga_merge_match_lists: bad arguments"; return -1; }
	int64_t total = 0, base = 0;
	for (int32_t r = 0; r < n_parts; ++r) total += n_matches[r];
	std::vector<pga_match_t> all;
	all.reserve((size_t)total);
	for (int32_t r = 0; r < n_parts; ++r) {
		const int32_t *l2g = local_to_global ? local_to_global[r] : nullptr;
		for (int64_t i = 0; i < n_matches[r]; ++i) {
			pga_match_t m = matches[r][i];
			if (l2g) {
				if (m.group < 0 || (n_local_groups && m.group >= n_local_groups[r])) { t_err = "pga_merge_match_lists: part " + std::to_string(r) + " holds a group id outside its table"; return -1; }
				m.group = l2g[m.group];
			}
			m.cigar_off += (uint64_t)base;
			all.push_back(m);
		}
		if (n_cigar_words[r] > 0) std::copy(cigars[r], cigars[r] + n_cigar_words[r], out_cigars + base);
		base += n_cigar_words[r];
	}
	// a rank's records are already in (group, query, own order) order: a stable sort by (group, query) restores the single-rank order
	std::stable_sort(all.begin(), all.end(), [](const pga_match_t &a, const pga_match_t &b) { return a.group != b.group ? a.group < b.group : a.qry < b.qry; });
	std::copy(all.begin(), all.end(), out_matches);
	return 0;
}

} // extern "C"
