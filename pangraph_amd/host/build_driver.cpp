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
// tasks.bin (little endian): "PGAB1\0\0\0", i32 n_tasks, i32 sensitivity, i32 n_threads_per_batch, i32 reserved; per task: i32 n_deps, i32 dep[n_deps],
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

bool load_tasks(const char *path, std::vector<Task> &tasks, pga_params_t &params)
{
	FILE *f = fopen(path, "rb");
	if (!f) { fprintf(stderr, "build_driver: cannot open %s\n", path); return false; }
	char magic[8]; int32_t hdr[4];
	bool ok = read_exact(f, magic, 8) && memcmp(magic, "PGAB1\0\0\0", 8) == 0 && read_exact(f, hdr, sizeof hdr) && hdr[0] >= 0;
	if (ok) {
		params.sensitivity = hdr[1]; params.kmer_length = 0; params.indel_len_threshold = 100; params.n_threads = hdr[2];
		tasks.resize((size_t)hdr[0]);
		for (Task &t : tasks) {
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

} // namespace

int main(int argc, char **argv)
{
	if (argc < 3) { fprintf(stderr, "usage: build_driver <tasks.bin> <out.bin> [slots [cap_bases]]\n"); return 2; }
	const int slots = argc > 3 ? std::max(1, atoi(argv[3])) : 6;
	const double cap_bases = argc > 4 ? atof(argv[4]) : 1.2e9;
	std::vector<Task> tasks;
	pga_params_t params;
	if (!load_tasks(argv[1], tasks, params)) return 2;
	const int32_t n = (int32_t)tasks.size();
	const bool dry = getenv("PGA_DRIVER_DRY") != nullptr;
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
	if (!dry) { pga_set_device(0); pga_warm_streams(slots); }
	if (pga_sched_start(s, nullptr, 0, nullptr, 0, slots, cap_bases, 0.0, 0, 0.05, 60e6) != 0) { fprintf(stderr, "build_driver: %s\n", pga_sched_error()); return 2; }

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
	for (int i = 0; i < slots; ++i) th.emplace_back(worker);
	for (std::thread &t : th) t.join();
	const double dt = now_s() - t0;
	const int32_t left = pga_sched_left(s);
	pga_sched_destroy(s);
	if (failed || left != 0) { fprintf(stderr, "build_driver: %d calls did not run\n", (int)left); return 1; }

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
	printf("build_driver: %d calls in %d batches, %lld matches, %.3f Gbp in %.3f s (%d slots)\n", (int)n, n_batches.load(), n_matches.load(), (double)total * 1e-9, dt, slots);
	return 0;
}
