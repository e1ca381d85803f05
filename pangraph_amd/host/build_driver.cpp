// build_driver.cpp -- a whole build driven from a compiled host: what the merge loop of `pangraph build` looks like above the C-ABI when several
// find_matches calls are kept in flight (reference: packages/pangraph/src/commands/build/build_run.rs:111-128 walks the guide tree one merge after
// the other; packages/pangraph/src/pangraph/graph_merging.rs:26-69,95-128 is the self-merge loop of one merge).  No Python in the process:
//   include/pga_sched.h   which calls are ready, which go into the next batch (pga_sched_take / pga_sched_finish)
//   include/pga_align.h   one batch of calls on the device (pga_batch_create -- the hand-over -- and pga_batch_align)
// `slots` worker threads loop over take -> create -> align -> keep the records per call -> finish.  The graph logic between a call's results and
// its parent's inputs (merging blocks, consensus) is the host's and out of scope here: the driver reads every call's block
// sequences from a task file, as bench.py takes them from its simulated build.
//
//   build_driver <tasks.bin> <out.bin> [slots [cap_bases]]
//   PGA_DRIVER_DRY=1: no device work -- every batch "finds" nothing (the host loop alone: task file, schedule, threads, result file; CPU test suite)
//
// SEVERAL RANKS (round 6; one process per GPU) -- the same two phases bench.py runs under torch.distributed,
// without Python and without torch:   PGA_RANK=r PGA_WORLD=W PGA_XDIR=<directory all ranks see>  build_driver <tasks.bin> <out.bin> ...
//   phase 1   pga_sched_partition cuts the guide tree (the task file carries it) into subtrees and deals them to the ranks; a rank runs the calls of its
//             subtrees under its own ready-set schedule and talks to nobody; then ONE exchange: every rank publishes its match list
//   phase 2   the calls above the cut, level by level, by ALL ranks together: every rank hands over the whole level and maps its share of the queries of
//             every group (pga_batch_align_shard); one more exchange; rank 0 puts everything together with pga_merge_match_lists -- the list ONE rank would
//             have produced -- and writes out.bin (the other ranks write nothing)
// The exchange is the HOST's (include/pga_align.h): here a blob per rank and phase in PGA_XDIR, written under a temporary name and renamed, polled by its
// readers -- enough for ranks that share a file system or a node; a production host puts RCCL / MPI point to point (or hipMemcpyPeer) in exchange_put /
// exchange_get and keeps the rest.  PGA_DEVICE=d selects the device (default: rank modulo the visible devices).
//
// tasks.bin (little endian): "PGAB1\0\0\0", i32 n_tasks, i32 sensitivity, i32 n_threads_per_batch, i32 n_nodes (0: no guide tree in the file); if n_nodes > 0:
//   i32 child0[n_nodes], i32 child1[n_nodes] (node 0 the root, -1 -1 a leaf), i32 task_node[n_tasks]; per task: i32 n_deps, i32 dep[n_deps],
//   i32 n_seqs; per sequence: u32 len, u32 name_len, name bytes, bases (ASCII).
// out.bin: "PGAR1\0\0\0", i32 n_tasks, i32 n_batches; per task (in task order): i64 n_matches, pga_match_t[n_matches] (group = task id, cigar_off into the
//   task's own pool), i64 n_cigar_words, u32 words[].
// tests/test_gpu_levels.py::test_native_build_driver_equals_the_python_host holds its records against the Python-driven run of the same build.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>
#include <unistd.h>
#include "../../include/pga_align.h"
#include "../../include/pga_sched.h"

namespace {

struct Task {
	std::vector<int32_t> deps;
	std::vector<std::string> names, seqs;
	int64_t bases = 0;
	// results
	std::vector<pga_match_t> matches;
	std::vector<uint32_t> cigars;
};

bool read_exact(FILE *f, void *p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }

struct Tree { std::vector<int32_t> child0, child1, task_node; };

bool load_tasks(const char *path, std::vector<Task> &tasks, pga_params_t &params, Tree &tree)
{
	FILE *f = fopen(path, "rb");
	if (!f) { fprintf(stderr, "build_driver: cannot open %s\n", path); return false; }
	char magic[8]; int32_t hdr[4];
	bool ok = read_exact(f, magic, 8) && memcmp(magic, "PGAB1\0\0\0", 8) == 0 && read_exact(f, hdr, sizeof hdr) && hdr[0] >= 0;
	if (ok) {
		params.sensitivity = hdr[1]; params.kmer_length = 0; params.indel_len_threshold = 100; params.n_threads = hdr[2];
		tasks.resize((size_t)hdr[0]);
		if (hdr[3] > 0) {
			tree.child0.resize((size_t)hdr[3]); tree.child1.resize((size_t)hdr[3]); tree.task_node.resize((size_t)hdr[0]);
			ok = read_exact(f, tree.child0.data(), 4 * (size_t)hdr[3]) && read_exact(f, tree.child1.data(), 4 * (size_t)hdr[3]) && read_exact(f, tree.task_node.data(), 4 * (size_t)hdr[0]);
		}
		if (ok) for (Task &t : tasks) {
			int32_t nd = 0, ns = 0;
			if (!(ok = read_exact(f, &nd, 4) && nd >= 0)) break;
			t.deps.resize((size_t)nd);
			if (!(ok = read_exact(f, t.deps.data(), 4 * (size_t)nd) && read_exact(f, &ns, 4) && ns >= 0)) break;
			t.names.resize((size_t)ns); t.seqs.resize((size_t)ns);
			for (int32_t i = 0; ok && i < ns; ++i) {
				uint32_t len = 0, nl = 0;
				ok = read_exact(f, &len, 4) && read_exact(f, &nl, 4);
				if (!ok) break;
				t.names[(size_t)i].resize(nl); t.seqs[(size_t)i].resize(len);
				ok = read_exact(f, &t.names[(size_t)i][0], nl) && read_exact(f, &t.seqs[(size_t)i][0], len);
				t.bases += len;
			}
			if (!ok) break;
		}
	}
	fclose(f);
	if (!ok) fprintf(stderr, "build_driver: %s is not a task file\n", path);
	return ok;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- the exchange of several ranks: a blob per (tag, rank) in a directory every rank sees (see the header) ----
bool exchange_put(const std::string &dir, const char *tag, int rank, const std::vector<uint8_t> &blob)
{
	const std::string fin = dir + "/" + tag + "_" + std::to_string(rank) + ".bin", tmp = fin + ".tmp";
	FILE *f = fopen(tmp.c_str(), "wb");
	if (!f) return false;
	const bool ok = blob.empty() || fwrite(blob.data(), 1, blob.size(), f) == blob.size();
	return fclose(f) == 0 && ok && rename(tmp.c_str(), fin.c_str()) == 0;
}
bool exchange_get(const std::string &dir, const char *tag, int rank, std::vector<uint8_t> &blob, double timeout_s)
{
	const std::string fin = dir + "/" + tag + "_" + std::to_string(rank) + ".bin";
	const double t_end = now_s() + timeout_s;
	for (;;) {
		if (FILE *f = fopen(fin.c_str(), "rb")) {
			fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
			blob.resize((size_t)(n > 0 ? n : 0));
			const bool ok = read_exact(f, blob.data(), blob.size());
			fclose(f);
			return ok;
		}
		if (now_s() > t_end) return false;
		usleep(200);
	}
}
// a match list as one blob: i64 n_matches, pga_match_t[], i64 n_cigar_words, u32[]
std::vector<uint8_t> pack_list(const std::vector<pga_match_t> &m, const std::vector<uint32_t> &c)
{
	std::vector<uint8_t> b(16 + m.size() * sizeof(pga_match_t) + c.size() * 4);
	const int64_t nm = (int64_t)m.size(), nc = (int64_t)c.size();
	uint8_t *p = b.data();
	memcpy(p, &nm, 8); p += 8; if (nm) memcpy(p, m.data(), m.size() * sizeof(pga_match_t)); p += m.size() * sizeof(pga_match_t);
	memcpy(p, &nc, 8); p += 8; if (nc) memcpy(p, c.data(), c.size() * 4);
	return b;
}
bool unpack_list(const std::vector<uint8_t> &b, std::vector<pga_match_t> &m, std::vector<uint32_t> &c)
{
	if (b.size() < 16) return false;
	int64_t nm = 0, nc = 0; const uint8_t *p = b.data();
	memcpy(&nm, p, 8); p += 8;
	if (nm < 0 || b.size() < 16 + (size_t)nm * sizeof(pga_match_t)) return false;
	m.resize((size_t)nm); if (nm) memcpy(m.data(), p, (size_t)nm * sizeof(pga_match_t)); p += (size_t)nm * sizeof(pga_match_t);
	memcpy(&nc, p, 8); p += 8;
	if (nc < 0 || b.size() != 16 + (size_t)nm * sizeof(pga_match_t) + (size_t)nc * 4) return false;
	c.resize((size_t)nc); if (nc) memcpy(c.data(), p, (size_t)nc * 4);
	return true;
}

} // namespace

int main(int argc, char **argv)
{
	if (argc < 3) { fprintf(stderr, "usage: build_driver <tasks.bin> <out.bin> [slots [cap_bases]]\n"); return 2; }
	const int slots = argc > 3 ? std::max(1, atoi(argv[3])) : 6;
	const double cap_bases = argc > 4 ? atof(argv[4]) : 1.2e9;
	std::vector<Task> tasks;
	pga_params_t params;
	Tree tree;
	if (!load_tasks(argv[1], tasks, params, tree)) return 2;
	const int32_t n = (int32_t)tasks.size();
	const bool dry = getenv("PGA_DRIVER_DRY") != nullptr;
	const int world = getenv("PGA_WORLD") ? std::max(1, atoi(getenv("PGA_WORLD"))) : 1, rank = getenv("PGA_RANK") ? atoi(getenv("PGA_RANK")) : 0;
	const std::string xdir = getenv("PGA_XDIR") ? getenv("PGA_XDIR") : "";
	if (world > 1 && (rank < 0 || rank >= world || xdir.empty() || tree.child0.empty())) {
		fprintf(stderr, "build_driver: several ranks need PGA_RANK in [0, PGA_WORLD), PGA_XDIR and a task file that carries the guide tree\n"); return 2; }
	if (!dry && pga_device_count() <= 0) { fprintf(stderr, "build_driver: no HIP device (the library has no CPU path)\n"); return 3; }

	// the task graph of the build
	std::vector<int64_t> dep_off((size_t)n + 1, 0), bases((size_t)n);
	std::vector<int32_t> dep, n_seqs((size_t)n);
	for (int32_t i = 0; i < n; ++i) {
		dep.insert(dep.end(), tasks[(size_t)i].deps.begin(), tasks[(size_t)i].deps.end());
		dep_off[(size_t)i + 1] = (int64_t)dep.size();
		bases[(size_t)i] = tasks[(size_t)i].bases; n_seqs[(size_t)i] = (int32_t)tasks[(size_t)i].seqs.size();
	}
	if (dep.empty()) dep.push_back(0);
	pga_sched_t *s = pga_sched_create(n, dep_off.data(), dep.data(), bases.data(), n_seqs.data());
	if (!s) { fprintf(stderr, "build_driver: %s\n", pga_sched_error()); return 2; }
	if (!dry) { const int nd = pga_device_count(); pga_set_device(getenv("PGA_DEVICE") ? atoi(getenv("PGA_DEVICE")) : rank % std::max(1, nd)); pga_warm_streams(slots); }
	// the plan of several ranks: owner[t] = the rank whose subtree holds call t, -1 = above the cut (every rank computes the same plan without talking)
	std::vector<int32_t> owner((size_t)std::max(1, n), 0), mine;
	if (world > 1) {
		const int per_rank = getenv("PGA_SUBTREES_PER_RANK") ? std::max(1, atoi(getenv("PGA_SUBTREES_PER_RANK"))) : 4;
		if (pga_sched_partition((int32_t)tree.child0.size(), tree.child0.data(), tree.child1.data(), n, tree.task_node.data(), bases.data(), world, per_rank, owner.data()) < 0) {
			fprintf(stderr, "build_driver: %s\n", pga_sched_error()); return 2; }
		for (int32_t i = 0; i < n; ++i) if (owner[(size_t)i] == rank) mine.push_back(i);
	}
	const bool have_phase1 = world == 1 || !mine.empty();
	if (have_phase1 && pga_sched_start(s, world > 1 ? mine.data() : nullptr, (int32_t)mine.size(), nullptr, 0, slots, cap_bases, 0.0, 0, 0.05, 60e6) != 0) {
		fprintf(stderr, "build_driver: %s\n", pga_sched_error()); return 2; }

	std::atomic<int> failed{0}, n_batches{0};
	std::atomic<long long> n_matches{0};
	const double t0 = now_s();
	auto worker = [&]() {
		std::vector<int32_t> ids((size_t)std::max(1, n));
		for (;;) {
			int32_t ticket = -1;
			const int32_t k = pga_sched_take(s, ids.data(), (int32_t)ids.size(), &ticket);
			if (k <= 0) return;
			// one group per call; nothing is copied on the host: the batch entry reads the sequences where they lie
			std::vector<int64_t> group_off((size_t)k + 1, 0);
			std::vector<const char*> seqs, names;
			std::vector<uint32_t> lens;
			for (int32_t g = 0; g < k; ++g) {
				const Task &t = tasks[(size_t)ids[(size_t)g]];
				for (size_t i = 0; i < t.seqs.size(); ++i) { seqs.push_back(t.seqs[i].data()); lens.push_back((uint32_t)t.seqs[i].size()); names.push_back(t.names[i].c_str()); }
				group_off[(size_t)g + 1] = (int64_t)seqs.size();
			}
			if (dry) { ++n_batches; std::this_thread::sleep_for(std::chrono::microseconds(200)); pga_sched_finish(s, ticket); continue; }
			pga_batch_t *b = nullptr; pga_result_t *r = nullptr;
			int rc = pga_batch_create(k, group_off.data(), seqs.data(), lens.data(), names.data(), &b);
			if (rc == 0) rc = pga_batch_align(b, &params, &r);
			if (rc != 0) {
				fprintf(stderr, "build_driver: batch of %d calls failed: %s\n", (int)k, pga_last_error());
				if (b) pga_batch_free(b);
				failed = 1; pga_sched_abort(s);
				return;
			}
			// the records come ordered by (group, query, the aligner's order): cut them per call, each with a CIGAR pool of its own
			const int64_t nm = pga_result_n_matches(r);
			const pga_match_t *m = pga_result_matches(r);
			uint64_t n_ops = 0;
			const uint32_t *cg = pga_result_cigars(r, &n_ops);
			for (int64_t i = 0; i < nm; ++i) {
				Task &t = tasks[(size_t)ids[(size_t)m[i].group]];
				pga_match_t rec = m[i];
				rec.group = ids[(size_t)m[i].group];
				rec.cigar_off = (uint64_t)t.cigars.size();
				t.cigars.insert(t.cigars.end(), cg + m[i].cigar_off, cg + m[i].cigar_off + m[i].n_cigar);
				t.matches.push_back(rec);
			}
			n_matches += nm; ++n_batches;
			pga_result_free(r);
			pga_batch_free(b);
			pga_sched_finish(s, ticket);          // the parents' round 0 / the next self-merge round become ready
		}
	};
	std::vector<std::thread> th;
	if (have_phase1) for (int i = 0; i < slots; ++i) th.emplace_back(worker);
	for (std::thread &t : th) t.join();
	const int32_t left = have_phase1 ? pga_sched_left(s) : 0;
	pga_sched_destroy(s);
	if (failed || left != 0) { fprintf(stderr, "build_driver: %d calls did not run\n", (int)left); return 1; }
	const double t_phase1 = now_s() - t0;
	double t_phase2 = 0.0;
	int n_above = 0;
	if (world > 1) {
		const double xto = getenv("PGA_XTIMEOUT_S") ? atof(getenv("PGA_XTIMEOUT_S")) : 600.0;
		// ---- exchange 1: this rank's subtrees (group = global call id, CIGAR offsets into the blob's own pool) ----
		auto flatten = [&](const std::vector<int32_t> &ids, std::vector<pga_match_t> &m, std::vector<uint32_t> &c) {
			for (int32_t id : ids) { const Task &t = tasks[(size_t)id]; for (pga_match_t rec : t.matches) { rec.cigar_off += (uint64_t)c.size(); m.push_back(rec); } c.insert(c.end(), t.cigars.begin(), t.cigars.end()); }
		};
		{ std::vector<pga_match_t> m; std::vector<uint32_t> c; flatten(mine, m, c); if (!exchange_put(xdir, "phase1", rank, pack_list(m, c))) { fprintf(stderr, "build_driver: cannot publish to %s\n", xdir.c_str()); return 2; } }
		// (a real host needs the merged graphs of the subtrees before the calls above the cut can be made: every rank waits for every rank here)
		std::vector<std::vector<uint8_t>> p1((size_t)world), p2((size_t)world);
		for (int r = 0; r < world; ++r) if (!exchange_get(xdir, "phase1", r, p1[(size_t)r], xto)) { fprintf(stderr, "build_driver: rank %d did not publish phase 1\n", r); return 1; }
		// ---- phase 2: the calls above the cut, level by level, the queries of every group split over the ranks ----
		const double t2 = now_s();
		std::vector<uint8_t> done((size_t)n, 0);
		std::vector<int32_t> top;
		for (int32_t i = 0; i < n; ++i) { if (owner[(size_t)i] >= 0) done[(size_t)i] = 1; else top.push_back(i); }
		n_above = (int)top.size();
		std::vector<pga_match_t> m2; std::vector<uint32_t> c2;
		while (!top.empty()) {
			std::vector<int32_t> level, rest;
			for (int32_t id : top) { bool ready = true; for (int32_t d : tasks[(size_t)id].deps) ready = ready && done[(size_t)d]; (ready ? level : rest).push_back(id); }
			if (level.empty()) { fprintf(stderr, "build_driver: the calls above the cut do not resolve\n"); return 1; }
			std::vector<int64_t> group_off((size_t)level.size() + 1, 0);
			std::vector<const char*> seqs, names; std::vector<uint32_t> lens;
			for (size_t g = 0; g < level.size(); ++g) {
				const Task &t = tasks[(size_t)level[g]];
				for (size_t i = 0; i < t.seqs.size(); ++i) { seqs.push_back(t.seqs[i].data()); lens.push_back((uint32_t)t.seqs[i].size()); names.push_back(t.names[i].c_str()); }
				group_off[g + 1] = (int64_t)seqs.size();
			}
			if (!dry) {
				pga_batch_t *b = nullptr; pga_result_t *r = nullptr;
				int rc = pga_batch_create((int32_t)level.size(), group_off.data(), seqs.data(), lens.data(), names.data(), &b);
				if (rc == 0) rc = pga_batch_align_shard(b, &params, rank, world, &r);
				if (rc != 0) { fprintf(stderr, "build_driver: a level of %zu calls failed: %s\n", level.size(), pga_last_error()); if (b) pga_batch_free(b); return 1; }
				const int64_t nm = pga_result_n_matches(r); const pga_match_t *m = pga_result_matches(r);
				uint64_t n_ops = 0; const uint32_t *cg = pga_result_cigars(r, &n_ops);
				for (int64_t i = 0; i < nm; ++i) { pga_match_t rec = m[i]; rec.group = level[(size_t)m[i].group]; rec.cigar_off = (uint64_t)c2.size(); c2.insert(c2.end(), cg + m[i].cigar_off, cg + m[i].cigar_off + m[i].n_cigar); m2.push_back(rec); }
				n_matches += nm;
				pga_result_free(r); pga_batch_free(b);
			}
			++n_batches;
			for (int32_t id : level) done[(size_t)id] = 1;
			top.swap(rest);
		}
		t_phase2 = now_s() - t2;
		// ---- exchange 2, and the owner's side of both: the list ONE rank would have produced ----
		if (!exchange_put(xdir, "phase2", rank, pack_list(m2, c2))) { fprintf(stderr, "build_driver: cannot publish to %s\n", xdir.c_str()); return 2; }
		if (rank != 0) { printf("build_driver: rank %d of %d: %zu calls of its own in %.3f s, %d calls above the cut in %.3f s\n", rank, world, mine.size(), t_phase1, n_above, t_phase2); return 0; }
		for (int r = 0; r < world; ++r) if (!exchange_get(xdir, "phase2", r, p2[(size_t)r], xto)) { fprintf(stderr, "build_driver: rank %d did not publish phase 2\n", r); return 1; }
		std::vector<std::vector<pga_match_t>> pm((size_t)2 * world); std::vector<std::vector<uint32_t>> pc((size_t)2 * world);
		std::vector<const pga_match_t*> mp; std::vector<const uint32_t*> cp; std::vector<int64_t> nmv, ncv;
		int64_t tot_m = 0, tot_c = 0;
		for (int k = 0; k < 2 * world; ++k) {
			if (!unpack_list(k < world ? p1[(size_t)k] : p2[(size_t)(k - world)], pm[(size_t)k], pc[(size_t)k])) { fprintf(stderr, "build_driver: a published list is damaged\n"); return 1; }
			mp.push_back(pm[(size_t)k].data()); cp.push_back(pc[(size_t)k].data()); nmv.push_back((int64_t)pm[(size_t)k].size()); ncv.push_back((int64_t)pc[(size_t)k].size());
			tot_m += nmv.back(); tot_c += ncv.back();
		}
		std::vector<pga_match_t> all((size_t)std::max<int64_t>(1, tot_m)); std::vector<uint32_t> allc((size_t)std::max<int64_t>(1, tot_c));
		if (pga_merge_match_lists(2 * world, mp.data(), nmv.data(), cp.data(), ncv.data(), nullptr, nullptr, all.data(), allc.data()) != 0) { fprintf(stderr, "build_driver: %s\n", pga_last_error()); return 1; }
		for (Task &t : tasks) { t.matches.clear(); t.cigars.clear(); }
		long long total_m = 0;
		for (int64_t i = 0; i < tot_m; ++i) {
			Task &t = tasks[(size_t)all[(size_t)i].group];
			pga_match_t rec = all[(size_t)i];
			rec.cigar_off = (uint64_t)t.cigars.size();
			t.cigars.insert(t.cigars.end(), allc.begin() + (long)all[(size_t)i].cigar_off, allc.begin() + (long)(all[(size_t)i].cigar_off + all[(size_t)i].n_cigar));
			t.matches.push_back(rec); ++total_m;
		}
		n_matches = total_m;
	}
	const double dt = now_s() - t0;

	FILE *o = fopen(argv[2], "wb");
	if (!o) { fprintf(stderr, "build_driver: cannot write %s\n", argv[2]); return 2; }
	const int32_t hdr[2] = {n, n_batches.load()};
	fwrite("PGAR1\0\0\0", 1, 8, o); fwrite(hdr, 4, 2, o);
	for (const Task &t : tasks) {
		const int64_t nm = (int64_t)t.matches.size(), nc = (int64_t)t.cigars.size();
		fwrite(&nm, 8, 1, o); if (nm) fwrite(t.matches.data(), sizeof(pga_match_t), (size_t)nm, o);
		fwrite(&nc, 8, 1, o); if (nc) fwrite(t.cigars.data(), 4, (size_t)nc, o);
	}
	fclose(o);
	long long total = 0; for (const Task &t : tasks) total += t.bases;
	if (world > 1) printf("build_driver: %d ranks: %d calls (%d above the cut), %lld matches, %.3f Gbp in %.3f s (rank 0: phase 1 %.3f s, phase 2 %.3f s)\n", world, (int)n, n_above, n_matches.load(), (double)total * 1e-9, dt, t_phase1, t_phase2);
	else
	printf("build_driver: %d calls in %d batches, %lld matches, %.3f Gbp in %.3f s (%d slots)\n", (int)n, n_batches.load(), n_matches.load(), (double)total * 1e-9, dt, slots);
	return 0;
}
