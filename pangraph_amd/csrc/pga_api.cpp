// pga_api.cpp -- the C-ABI of libpgalign.so and the host orchestration of one batch.
//
// Part 1 (include/pga_mm2_abi.h): the minimap2-sys symbols the reference's Rust crate binds.  pangraph always
// queries exactly the sequences it indexed (align_with_minimap2_lib.rs:62-74), so mm_idx_str() uploads,
// sketches and indexes the whole group on the GPU, the FIRST mm_map() against an index runs the complete
// all-vs-all batch on the GPU (options only arrive with mm_map), and every mm_map() hands back its query's
// records as malloc()ed mm_reg1_t[] exactly as minimap2 would (caller frees, packages/minimap2/src/map.rs:407-420).
// Part 2 (include/pga_align.h): the native batch entry for a level-synchronous host.
#include "pga_common.h"
#include <mutex>
#include <array>
#include "pga_pipeline.h"
#include "pga_dp.h"
#include "../../include/pga_align.h"
#include <atomic>
#include <chrono>
#include <map>
#include <thread>
#include <climits>
#include <cstdio>

using namespace pga;

static thread_local std::string g_err;
static void set_err(const std::string &s) { g_err = s; }
static void mem_log(const char *what)
{
	static const bool on = getenv("PGA_MEMLOG") != nullptr;
	if (!on) return;
	size_t fr = 0, tot = 0; (void)hipMemGetInfo(&fr, &tot);
	fprintf(stderr, "[pga-mem] %-28s used %.1f GB of %.1f\n", what, (double)(tot - fr) / 1e9, (double)tot / 1e9);
}
static inline double cpu_s() { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }   // (all threads of the process: clean with one batch in flight)
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---------------------------------------------------------------- options (options.c)
extern "C" void mm_idxopt_init(mm_idxopt_t *opt) // options.c:5-12
{
	memset(opt, 0, sizeof(*opt));
	opt->k = 15, opt->w = 10, opt->flag = 0, opt->bucket_bits = 14;
	opt->mini_batch_size = 50000000; opt->batch_size = 8000000000ULL;
}
extern "C" void mm_mapopt_init(mm_mapopt_t *o) // options.c:14-64
{
	memset(o, 0, sizeof(*o));
	o->seed = 11; o->mid_occ_frac = 2e-4f; o->min_mid_occ = 10; o->max_mid_occ = 1000000; o->sdust_thres = 0; o->q_occ_frac = 0.01f;
	o->min_cnt = 3; o->min_chain_score = 40; o->bw = 500; o->bw_long = 20000; o->max_gap = 5000; o->max_gap_ref = -1;
	o->max_chain_skip = 25; o->max_chain_iter = 5000; o->rmq_inner_dist = 1000; o->rmq_size_cap = 100000; o->rmq_rescue_size = 1000;
	o->rmq_rescue_ratio = 0.1f; o->chain_gap_scale = 0.8f; o->chain_skip_scale = 0.0f; o->max_max_occ = 4095; o->occ_dist = 500;
	o->mask_level = 0.5f; o->mask_len = INT_MAX; o->pri_ratio = 0.8f; o->best_n = 5; o->alt_drop = 0.15f;
	o->a = 2; o->b = 4; o->q = 4; o->e = 2; o->q2 = 24; o->e2 = 1; o->sc_ambi = 1; o->zdrop = 400; o->zdrop_inv = 200; o->end_bonus = -1;
	o->min_dp_max = o->min_chain_score * o->a; o->min_ksw_len = 200; o->anchor_ext_len = 20; o->anchor_ext_shift = 6; o->max_clip_ratio = 1.0f;
	o->mini_batch_size = 500000000; o->max_sw_mat = 100000000; o->cap_kalloc = 1000000000; o->rank_min_len = 500; o->rank_frac = 0.9f;
	o->pe_ori = 0; o->pe_bonus = 33;
}
extern "C" int mm_set_opt(const char *preset, mm_idxopt_t *io, mm_mapopt_t *mo) // options.c:88-162 (asm* only: SURVEY.md section 8a row O)
{
	if (preset == 0) { mm_idxopt_init(io); mm_mapopt_init(mo); return 0; }
	if (strncmp(preset, "asm", 3) == 0) {
		io->flag = 0, io->k = 19, io->w = 19;
		mo->bw = 1000, mo->bw_long = 100000; mo->max_gap = 10000; mo->flag |= MM_F_RMQ;
		mo->min_mid_occ = 50, mo->max_mid_occ = 500; mo->min_dp_max = 200; mo->best_n = 50;
		if (strcmp(preset, "asm5") == 0) mo->a = 1, mo->b = 19, mo->q = 39, mo->q2 = 81, mo->e = 3, mo->e2 = 1, mo->zdrop = mo->zdrop_inv = 200;
		else if (strcmp(preset, "asm10") == 0) mo->a = 1, mo->b = 9, mo->q = 16, mo->q2 = 41, mo->e = 2, mo->e2 = 1, mo->zdrop = mo->zdrop_inv = 200;
		else if (strcmp(preset, "asm20") == 0) mo->a = 1, mo->b = 4, mo->q = 6, mo->q2 = 26, mo->e = 2, mo->e2 = 1, mo->zdrop = mo->zdrop_inv = 200, io->w = 10;
		else return -1;
		return 0;
	}
	return -1;
}
extern "C" int mm_check_opt(const mm_idxopt_t *io, const mm_mapopt_t *mo) // options.c:164-234
{
	if (mo->bw > mo->bw_long) return -8;
	if ((mo->flag & MM_F_RMQ) && (mo->flag & (MM_F_SR | MM_F_SPLICE))) return -7;
	if (io->k <= 0 || io->w <= 0) return -5;
	if (mo->best_n < 0) return -4;
	if (mo->pri_ratio < 0.0f || mo->pri_ratio > 1.0f) return -4;
	if ((mo->flag & MM_F_FOR_ONLY) && (mo->flag & MM_F_REV_ONLY)) return -3;
	if (mo->e <= 0 || mo->q <= 0) return -1;
	if ((mo->q != mo->q2 || mo->e != mo->e2) && !(mo->e > mo->e2 && mo->q + mo->e < mo->q2 + mo->e2)) return -2;
	if ((mo->q + mo->e) + (mo->q2 + mo->e2) > 127) return -1;
	if (mo->zdrop < mo->zdrop_inv) return -5;
	return 0;
}

// ---------------------------------------------------------------- the index handle
struct PgaIdx {
	mm_idx_t hdr;                    // must stay first: the Rust side reads n_seq, seq[i].name, seq[i].len
	SeqSet S; Minimizers M; Index I; DBuf<uint32_t> grp; DBuf<int32_t> d_name_rank, d_mid_occ;
	DBuf<uint8_t> d_own; bool sharded = false;    // pga_batch_align_shard: the queries this call maps (all sequences are indexed)
	std::vector<int32_t> mid_occ_raw;   // mm_idx_cal_max_occ per group (cached per fraction)
	float mid_occ_frac = -1.0f;
	std::vector<mm_idx_seq_t> seq_hdr;
	std::map<std::string, int> by_name;
	std::mutex mtx;
	bool have_results = false, indexed = false; mm_mapopt_t res_opt;
	std::string failure;             // first error of the batch behind this index: sticky for the options it happened with (failure_opt)
	mm_mapopt_t failure_opt;
	std::vector<std::vector<Reg>> results;
	Timers tm;
	hipStream_t st = 0;              // the part's own (non-blocking) stream
	int arena = 0;                   // device-memory arena of the index, leased for its lifetime (pga_mem.cpp)
	PgaIdx() : arena(dev_lease_arena()) {}
	~PgaIdx() { if (st) { (void)sync_stream(st); } release_buffers(); if (st) stream_release(st); dev_release_arena(arena); }
	void release_buffers() { S.d_pk2.release(); S.d_nmask.release(); S.d_off.release(); S.d_len.release(); S.d_grp_of_seq.release(); S.d_grp_base.release(); M.mz.release(); M.seq_off.release();
		I.key.release(); I.occ_off.release(); I.occ.release(); I.key_grp.release(); grp.release(); d_name_rank.release(); d_mid_occ.release(); }
};

// HIP's current device is a property of the HOST THREAD (device 0 until the thread says otherwise).  pga_set_device() therefore also records
// the device as the process default, and every entry point applies it to the thread it is called on: a host that calls the library from
// worker threads (the ready-set schedule of bench.py: one thread per batch in flight) works on the rank's device, not on device 0.
static std::atomic<int> g_default_dev(-1);
static void apply_default_device() noexcept { const int d = g_default_dev.load(); if (d >= 0) { int cur = -1; if (hipGetDevice(&cur) != hipSuccess || cur != d) (void)hipSetDevice(d); } }
static void require_device()
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
		throw std::runtime_error("pga: no HIP device visible -- libpgalign.so is a gfx950 backend and has no CPU fallback");
	const int d = g_default_dev.load();
	if (d >= 0) { int cur = -1; if (hipGetDevice(&cur) != hipSuccess || cur != d) PGA_HIP(hipSetDevice(d)); }
	// How a host thread waits for the device.  The runtime's default SPINS inside hipStreamSynchronize / hipEventSynchronize: the six threads that drive the
	// batches in flight each held a core for the whole step -- 13.7 core-seconds per 2.0 s step, of which 7.5 were that spinning (round 6, ABAB on one box:
	// 6.9 / 6.5 cores busy against 3.0 / 3.1 with blocking waits, the step the same within its spread, 2 005 / 2 112 against 2 010 / 2 023 ms).  A host with eight
	// ranks on one node has two cores per rank, so the library asks for BLOCKING waits on every device it is used on, once; PGA_SYNC=spin | yield | auto
	// leaves the choice to the host.
	{
		static std::mutex mu; static std::vector<int> flagged;
		int cur = 0;
		if (hipGetDevice(&cur) == hipSuccess) {
			std::lock_guard<std::mutex> lk(mu);
			if (std::find(flagged.begin(), flagged.end(), cur) == flagged.end()) {
				flagged.push_back(cur);
				const char *e = getenv("PGA_SYNC");
				const unsigned f = !e || !strcmp(e, "block") ? hipDeviceScheduleBlockingSync : !strcmp(e, "yield") ? hipDeviceScheduleYield : !strcmp(e, "spin") ? hipDeviceScheduleSpin : hipDeviceScheduleAuto;
				const hipError_t r = hipSetDeviceFlags(f);
				if (getenv("PGA_VERBOSE")) fprintf(stderr, "[pga] device %d: hipSetDeviceFlags(%u) -> %s\n", cur, f, hipGetErrorString(r));
				(void)hipGetLastError();
			}
		}
	}
}

static void check_supported(const mm_mapopt_t &o, int k, int w)
{
	if (o.flag & (MM_F_SPLICE | MM_F_SR | MM_F_QSTRAND | MM_F_HEAP_SORT | MM_F_EQX))
		throw std::runtime_error("pga: splice / short-read / qstrand / heap-sort / eqx modes are outside pangraph's path (SURVEY.md section 2)");
	if (!(o.flag & MM_F_RMQ)) throw std::runtime_error("pga: only RMQ chaining (asm presets) is implemented");
	if (!(o.flag & MM_F_NO_LJOIN)) throw std::runtime_error("pga: long-join re-chaining is not implemented (pangraph always passes -X)");
	if (!(o.flag & MM_F_ALL_CHAINS)) throw std::runtime_error("pga: primary/secondary selection is not implemented (pangraph always passes -X)");
	if (o.q == o.q2 && o.e == o.e2) throw std::runtime_error("pga: single-affine scoring (ksw_extz2) is out of scope");
	if (-(-o.b) > 2 * (o.q + o.e) && o.b > 2 * (o.q + o.e)) throw std::runtime_error("pga: mismatch penalty larger than 2*(q+e) disables the reference DP");
	if (k > 28 || k < 1 || w < 1 || w > 255) throw std::runtime_error("pga: k must be in [1,28] and w in [1,255]");
	if (o.sdust_thres > 0) throw std::runtime_error("pga: SDUST masking is not implemented");
	// the inversion test (ksw_ll_i16, align.c:845-855) runs over windows of up to max_gap bases and the device kernel holds 10 240: refused
	// here, by name, instead of in the middle of a batch (every asm preset has max_gap = 5 000 or 10 000)
	if (o.max_gap > 10240) throw std::runtime_error("pga: max_gap = " + std::to_string(o.max_gap) + " is larger than the 10240 bases the ksw_ll_i16 kernel holds (pangraph's presets: at most 10000)");
}

static void idx_sketch_index(PgaIdx &ix)
{
	const int w = ix.hdr.w, k = ix.hdr.k;
	double t1 = now_s(); const double c1 = cpu_s();
	sketch_all(ix.S, w, k, ix.M, ix.st, &ix.tm);
	double t2 = now_s(); const double c2 = cpu_s();
	{
		EventTimer et(ix.st);
		build_index_ex(ix.S, ix.M, w, k, ix.I, ix.grp, ix.st);
		KernelStat &ks = ix.tm.kern[K_INDEX];        // 16 B minimizer read for the sort + sorted write + table write
		ks.ms += et.stop(K_INDEX); ks.launches += 1; ks.alg_bytes += 48.0 * (double)ix.M.n;
	}
	double t3 = now_s();
	if (getenv("PGA_VERBOSE")) fprintf(stderr, "[pga]   process cpu: sketch %.2f index %.2f ms\n", (c2 - c1) * 1e3, (cpu_s() - c2) * 1e3);
	mem_log("after sketch+index");
	ix.tm.sketch = t2 - t1, ix.tm.index = t3 - t2; ix.tm.n_mz = (double)ix.M.n;
	ix.indexed = true;
}

static PgaIdx *idx_build(int w, int k, int n, const char *const *seq, const uint32_t *len, const char *const *name, int n_grp = 1, const int64_t *grp_off = nullptr, bool do_index = true,
                         const SeqFrom *from = nullptr, const uint8_t *const *from_probe = nullptr)
{
	require_device();
	std::unique_ptr<PgaIdx> ix(new PgaIdx());
	ArenaScope arena_scope(ix->arena);
	memset(&ix->hdr, 0, sizeof(ix->hdr));
	ix->st = stream_lease();
	if (w < 1) w = 1;
	double t0 = now_s(); const double c0 = cpu_s();
	const int64_t one_grp[2] = {0, n};
	if (!grp_off) grp_off = one_grp, n_grp = 1;
	upload_seqs(ix->S, n, seq, len, name, n_grp, grp_off, ix->st, from, from_probe);
	const double t_up = now_s();
	const std::vector<std::string> &nm = ix->S.name;                  // (the names live in the sequence set: no second copy)
	ix->seq_hdr.resize((size_t)n);
	for (int i = 0; i < n; ++i) {
		ix->seq_hdr[i].name = name && name[i] ? const_cast<char*>(nm[i].c_str()) : nullptr;
		ix->seq_hdr[i].offset = ix->S.off[i]; ix->seq_hdr[i].len = len[i]; ix->seq_hdr[i].is_alt = 0;
		if (do_index && name && name[i] && n_grp == 1) ix->by_name[nm[i]] = i;     // (mm_map's lookup: a batch handle is never asked by name)
	}
	// rank of every name under strcmp order inside its group (skip_seed compares names as C strings, map.c:84,89)
	std::vector<int32_t> rank((size_t)n);
	for (int g = 0; g < n_grp; ++g) {
		const int b = (int)grp_off[g], m = (int)(grp_off[g + 1] - grp_off[g]);
		std::vector<int> ord((size_t)m); for (int i = 0; i < m; ++i) ord[i] = b + i;
		std::sort(ord.begin(), ord.end(), [&](int a, int c) { return strcmp(nm[a].c_str(), nm[c].c_str()) < 0; });
		for (int i = 0; i < m; ++i) {
			if (i > 0 && nm[ord[i]] == nm[ord[i - 1]] && name)
				throw std::runtime_error("pga: duplicate sequence name '" + nm[ord[i]] + "' inside a group (index.c:436 asserts uniqueness)");
			rank[ord[i]] = i;
		}
	}
	ix->d_name_rank.upload(rank, ix->st);
	ix->hdr.b = 14 < 2 * k ? 14 : 2 * k, ix->hdr.w = w, ix->hdr.k = k, ix->hdr.flag = name ? 0 : MM_I_NO_NAME;
	ix->hdr.n_seq = (uint32_t)n; ix->hdr.seq = ix->seq_hdr.data();
	PGA_HIP(sync_stream(ix->st));
	ix->tm.upload = now_s() - t0;
	if (getenv("PGA_VERBOSE")) fprintf(stderr, "[pga]   process cpu: hand-over %.2f ms\n", (cpu_s() - c0) * 1e3);
	if (getenv("PGA_VERBOSE")) fprintf(stderr, "[pga]   hand-over: sequences %.2f ms, names and ranks %.2f ms\n", (t_up - t0) * 1e3, (now_s() - t_up) * 1e3);
	if (do_index) idx_sketch_index(*ix);
	return ix.release();
}

// mm_mapopt_update (options.c:66-80) for every group of the batch
static std::vector<int32_t> group_mid_occ(PgaIdx &ix, const mm_mapopt_t &opt)
{
	std::vector<int32_t> v((size_t)ix.S.n_grp, opt.mid_occ);
	if (opt.mid_occ > 0) return v;
	if (ix.mid_occ_frac != opt.mid_occ_frac) { ix.mid_occ_raw = index_cal_max_occ(ix.S, ix.I, opt.mid_occ_frac, ix.st); ix.mid_occ_frac = opt.mid_occ_frac; }
	for (int g = 0; g < ix.S.n_grp; ++g) {
		int32_t m = ix.mid_occ_raw[g];
		if (m < opt.min_mid_occ) m = opt.min_mid_occ;
		if (opt.max_mid_occ > opt.min_mid_occ && m > opt.max_mid_occ) m = opt.max_mid_occ;
		v[g] = m;
	}
	return v;
}

static void run_batch(PgaIdx &ix, const mm_mapopt_t &opt, int n_threads)
{
	ArenaScope arena_scope(ix.arena);
	check_supported(opt, ix.I.k, ix.I.w);
	double t0 = now_s(); const double c0 = cpu_s();
	ix.d_mid_occ.upload(group_mid_occ(ix, opt), ix.st);
	SeedResult SR;
	{
		EventTimer et(ix.st);
		seed_all(ix.S, ix.M, ix.I, ix.grp, opt, ix.d_name_rank, ix.d_mid_occ, SR, ix.st, &ix.tm, ix.sharded ? ix.d_own.p : nullptr, exact_sorts_forced());
		KernelStat &ks = ix.tm.kern[K_SEED];         // query minimizers probe the table, anchors written, read and written by the sort
		ks.ms += et.stop(K_SEED); ks.launches += 1; ks.alg_bytes += 16.0 * (double)ix.M.n + 32.0 * (double)SR.n_a;
	}
	double t1 = now_s(); const double c1 = cpu_s();
	mem_log("after seed");
	ChainResult CR;
	CR.want_host_anchors = getenv("PGA_HOST_PLAN") != nullptr;            // (default: the anchors stay on the device and the regions are planned there, pga_plan.hip)
	chain_all(ix.S, SR, opt, ix.I.k, CR, ix.st, &ix.tm, exact_sorts_forced());
	double t2 = now_s(); const double c2 = cpu_s();
	mem_log("after chain");
	if (n_threads <= 0) { n_threads = usable_cpus(); }
	set_thread_budget(n_threads);
	align_batch(ix.S, opt, ix.I.k, SR.h_q_aoff, CR, SR.h_rep_len, ix.results, n_threads, &ix.tm, ix.st);
	double t3 = now_s();
	if (getenv("PGA_VERBOSE")) fprintf(stderr, "[pga]   align_batch returned after %.4f s\n[pga]   process cpu: seed %.2f chain %.2f align %.2f ms\n", t3 - t2, (c1 - c0) * 1e3, (c2 - c1) * 1e3, (cpu_s() - c2) * 1e3);
	mem_log("after align");
	ix.tm.seed = t1 - t0, ix.tm.chain = t2 - t1, ix.tm.align = t3 - t2; ix.tm.n_anchor = (double)SR.n_a;
	ix.have_results = true; ix.res_opt = opt;
	if (getenv("PGA_VERBOSE")) { long long ms_[4]; dev_mem_stats(ms_); fprintf(stderr, "[pga] allocator so far: %lld hipMalloc %.3f s, %lld hipFree %.3f s\n", ms_[0], ms_[1] * 1e-9, ms_[2], ms_[3] * 1e-9); }
	if (getenv("PGA_VERBOSE"))
		fprintf(stderr, "[pga] n_seq=%d bases=%llu mz=%.0f anchors=%.0f | upload %.3f sketch %.3f index %.3f seed %.3f chain %.3f align %.3f s | dp jobs %.0f cells %.3g\n",
		        ix.S.n_seq, (unsigned long long)ix.S.total, ix.tm.n_mz, ix.tm.n_anchor, ix.tm.upload, ix.tm.sketch, ix.tm.index, ix.tm.seed, ix.tm.chain, ix.tm.align, ix.tm.dp_jobs, ix.tm.dp_cells);
}

// ---------------------------------------------------------------- minimap2-sys ABI
extern "C" mm_idx_t *mm_idx_str(int w, int k, int is_hpc, int bucket_bits, int n, const char **seq, const char **name) // index.c:408-456
{
	(void)bucket_bits;
	if (n <= 0) return 0;
	try {
		if (is_hpc) throw std::runtime_error("pga: homopolymer-compressed minimizers are outside pangraph's path");
		std::vector<uint32_t> len((size_t)n);
		for (int i = 0; i < n; ++i) len[i] = (uint32_t)strlen(seq[i]);
		return reinterpret_cast<mm_idx_t*>(idx_build(w, k, n, seq, len.data(), name));
	} catch (std::exception &e) { set_err(e.what()); fprintf(stderr, "[pga] mm_idx_str: %s\n", e.what()); return 0; }
}
extern "C" void mm_idx_destroy(mm_idx_t *mi) { apply_default_device(); delete reinterpret_cast<PgaIdx*>(mi); }

extern "C" void mm_mapopt_update(mm_mapopt_t *opt, const mm_idx_t *mi) // options.c:66-80
{
	PgaIdx *ix = reinterpret_cast<PgaIdx*>(const_cast<mm_idx_t*>(mi));
	if (opt->mid_occ <= 0) {
		ArenaScope arena_scope(ix->arena);
		try { opt->mid_occ = group_mid_occ(*ix, *opt)[0]; }
		catch (std::exception &e) { set_err(e.what()); fprintf(stderr, "[pga] mm_mapopt_update: %s\n", e.what()); return; }
	}
	if (opt->bw_long < opt->bw) opt->bw_long = opt->bw;
}

struct mm_tbuf_s { int unused; };
extern "C" mm_tbuf_t *mm_tbuf_init(void) { return (mm_tbuf_t*)calloc(1, sizeof(mm_tbuf_s)); }
extern "C" void mm_tbuf_destroy(mm_tbuf_t *b) { free(b); }

static mm_reg1_t *regs_to_c(const std::vector<Reg> &regs, int *n_regs)
{
	*n_regs = (int)regs.size();
	if (regs.empty()) return 0;
	mm_reg1_t *out = (mm_reg1_t*)calloc(regs.size(), sizeof(mm_reg1_t));
	for (size_t i = 0; i < regs.size(); ++i) {
		const Reg &r = regs[i]; mm_reg1_t &o = out[i];
		o.id = r.id, o.cnt = r.cnt, o.rid = r.rid, o.score = r.score, o.qs = r.qs, o.qe = r.qe, o.rs = r.rs, o.re = r.re;
		o.parent = r.parent, o.subsc = r.subsc, o.as = r.as, o.mlen = r.mlen, o.blen = r.blen, o.n_sub = r.n_sub, o.score0 = r.score0;
		o.mapq = r.mapq, o.split = r.split, o.rev = r.rev, o.inv = r.inv, o.split_inv = r.split_inv;
		o.hash = r.hash, o.div = -1.0f;
		if (r.has_p) {
			uint32_t cap = (uint32_t)r.cigar.size() + (uint32_t)sizeof(mm_extra_t) / 4;
			--cap, cap |= cap >> 1, cap |= cap >> 2, cap |= cap >> 4, cap |= cap >> 8, cap |= cap >> 16, ++cap; // kroundup32 (align.c:296)
			o.p = (mm_extra_t*)calloc(cap, 4);
			o.p->capacity = cap, o.p->dp_score = r.dp_score, o.p->dp_max = r.dp_max, o.p->dp_max2 = r.dp_max2;
			o.p->n_ambi = r.n_ambi, o.p->trans_strand = 0, o.p->n_cigar = (uint32_t)r.cigar.size();
			if (!r.cigar.empty()) memcpy(o.p->cigar, r.cigar.data(), r.cigar.size() * 4);
		}
	}
	return out;
}

// cheap identity check of a query against the indexed sequence of the same name: the 64 probe positions kept at upload
static bool seqs_same_bases(const SeqSet &S, int qid, const char *seq, int l_seq)
{
	const uint8_t *h = S.probe.data() + (size_t)qid * 64;
	const int step = l_seq > 64 ? l_seq / 64 : 1;
	int k = 0;
	for (int i = 0; i < l_seq && k < 64; i += step, ++k) {
		const char c = seq[i] & 0xdf; uint8_t code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : (c == 'T' || c == 'U') ? 3 : 4;
		if ((uint8_t)seq[i] < 0x40) code = 4;
		if (code != h[k]) return false;
	}
	return true;
}

static bool same_opt(const mm_mapopt_t &a, const mm_mapopt_t &b) { return memcmp(&a, &b, offsetof(mm_mapopt_t, split_prefix)) == 0; }

// The minimap2 ABI has no error channel: NULL with *n_regs = 0 means "no alignments", and the unchanged Rust crate would build
// an unmerged graph from it.  So a failure of the batch (device out of memory, an option check_supported rejects, a query that is
// not one of the indexed sequences) is fatal: the message goes to stderr and the process aborts -- unless PGA_MM_MAP_SOFT_ERRORS=1,
// in which case mm_map returns NULL and pga_last_error() holds the message (tests).  The failure is recorded on the index, so the N
// rayon callers do not re-run a failing batch N times.
static mm_reg1_t *mm_map_fail(PgaIdx *ix, const std::string &msg, int *n_regs)
{
	set_err(msg);
	fprintf(stderr, "[pga] mm_map: %s\n", msg.c_str());
	const char *soft = getenv("PGA_MM_MAP_SOFT_ERRORS");
	if (!(soft && soft[0] == '1')) { fprintf(stderr, "[pga] mm_map cannot report errors through the minimap2 ABI: aborting (PGA_MM_MAP_SOFT_ERRORS=1 returns NULL instead)\n"); abort(); }
	(void)ix; *n_regs = 0; return 0;
}

extern "C" mm_reg1_t *mm_map(const mm_idx_t *mi, int l_seq, const char *seq, int *n_regs, mm_tbuf_t *b, const mm_mapopt_t *opt, const char *name) // map.c:376-381
{
	(void)b;
	*n_regs = 0;
	PgaIdx *ix = reinterpret_cast<PgaIdx*>(const_cast<mm_idx_t*>(mi));
	if (l_seq == 0) return 0;
	try {
		require_device();                      // (the reference calls mm_map from its rayon workers: each thread gets the process's device)
		auto it = name ? ix->by_name.find(name) : ix->by_name.end();
		if (it == ix->by_name.end() || (int)ix->S.len[it->second] != l_seq)
			throw std::runtime_error(std::string("pga: mm_map() query '") + (name ? name : "(null)") + "' is not one of the indexed sequences; this backend aligns a group all-vs-all "
			                         "(what pangraph's find_matches does); mapping foreign queries is not implemented");
		const int qid = it->second;
		if (!seqs_same_bases(ix->S, qid, seq, l_seq)) throw std::runtime_error("pga: mm_map() query bases differ from the indexed sequence of the same name");
		std::lock_guard<std::mutex> lk(ix->mtx);
		// a failure sticks to the OPTIONS it happened with (the N rayon callers of one find_matches do not re-run a failing batch N times);
		// a call with other options gets a fresh attempt
		if (!ix->failure.empty() && same_opt(ix->failure_opt, *opt)) return mm_map_fail(ix, ix->failure, n_regs);
		ix->failure.clear();
		if (!ix->have_results || !same_opt(ix->res_opt, *opt)) {
			for (int attempt = 0; ; ++attempt) {
				try { run_batch(*ix, *opt, 0); break; }
				catch (std::exception &e) {
					std::string msg = e.what(); if (msg.empty()) msg = "unknown error";
					// device out of memory is not a property of the batch: give the allocator's idle blocks back and try once more
					if (attempt == 0 && (msg.find("out of memory") != std::string::npos || msg.find("hipErrorOutOfMemory") != std::string::npos)) { dev_trim(); continue; }
					ix->failure = msg; ix->failure_opt = *opt;
					return mm_map_fail(ix, ix->failure, n_regs);
				}
			}
		}
		return regs_to_c(ix->results[qid], n_regs);
	} catch (std::exception &e) { return mm_map_fail(ix, e.what(), n_regs); }
}

extern "C" double mm_event_identity(const mm_reg1_t *r) // align.c:897-917
{
	int32_t n_gapo = 0, n_gap = 0;
	if (r->p == 0) return -1.0f;
	for (uint32_t i = 0; i < r->p->n_cigar; ++i) {
		int32_t op = r->p->cigar[i] & 0xf, len = (int32_t)(r->p->cigar[i] >> 4);
		if (op == MM_CIGAR_INS || op == MM_CIGAR_DEL) ++n_gapo, n_gap += len;
	}
	return (double)r->mlen / (r->blen + (int32_t)r->p->n_ambi - n_gap + n_gapo);
}

// ---------------------------------------------------------------- native batch entry
struct pga_result_s { std::vector<pga_match_t> m; std::vector<uint32_t> cig; pga_stats_t st; };

static void collect_results(PgaIdx &ixr, int g0, const std::vector<int64_t> &goff, pga_result_s &R)
{
	PgaIdx *ix = &ixr;
	const int n = ix->S.n_seq;
	for (int q = 0; q < n; ++q) for (const Reg &r : ix->results[q]) {
		if (!r.has_p) throw std::runtime_error("Unable to find CIGAR string in the result"); // align_with_minimap2_lib.rs:118
		const int g = (int)ix->S.grp_of_seq[q], gb = (int)goff[g];
		pga_match_t m; memset(&m, 0, sizeof(m));
		m.group = g0 + g, m.qry = q - gb, m.ref = r.rid, m.qry_len = (int32_t)ix->S.len[q], m.qry_start = r.qs, m.qry_end = r.qe;
		m.ref_len = (int32_t)ix->S.len[gb + r.rid], m.ref_start = r.rs, m.ref_end = r.re;
		m.matches = r.mlen, m.length = r.blen, m.quality = (int32_t)r.mapq, m.reverse = (int32_t)r.rev, m.align = r.dp_score;
		m.n_ambi = (int32_t)r.n_ambi, m.inv = (int32_t)r.inv;
		int32_t n_gapo = 0, n_gap = 0;
		for (uint32_t c : r.cigar) { int32_t op = c & 0xf, len = (int32_t)(c >> 4); if (op == 1 || op == 2) ++n_gapo, n_gap += len; }
		m.divergence = 1.0 - (double)r.mlen / (r.blen + (int32_t)r.n_ambi - n_gap + n_gapo);
		m.cigar_off = R.cig.size(), m.n_cigar = (uint32_t)r.cigar.size();
		R.cig.insert(R.cig.end(), r.cigar.begin(), r.cigar.end());
		R.m.push_back(m);
		R.st.aligned_span += (double)(r.qe - r.qs);
	}
	const Timers &t = ix->tm;
	R.st.upload += t.upload, R.st.sketch += t.sketch, R.st.index += t.index, R.st.seed += t.seed, R.st.chain += t.chain, R.st.align += t.align;
	R.st.n_bases += (double)ix->S.total, R.st.n_minimizers += t.n_mz, R.st.n_anchors += t.n_anchor, R.st.n_dp_jobs += t.dp_jobs, R.st.n_dp_cells += t.dp_cells;
	R.st.n_dp_bases += t.dp_bases;
	for (int i = 0; i < K_COUNT; ++i) R.st.kern_ms[i] += t.kern[i].ms, R.st.kern_launches[i] += t.kern[i].launches, R.st.kern_alg_bytes[i] += t.kern[i].alg_bytes, R.st.kern_cells[i] += t.kern[i].cells;
}

static void params_to_opts(const pga_params_t &p, mm_idxopt_t &io, mm_mapopt_t &mo) // align_with_minimap2_lib.rs:35-57 + options_args.rs:273-325
{
	const char *preset = p.sensitivity == 5 ? "asm5" : p.sensitivity == 10 ? "asm10" : p.sensitivity == 20 ? "asm20" : nullptr;
	if (!preset) throw std::runtime_error("Unknown sensitivity preset: " + std::to_string(p.sensitivity));
	mm_set_opt(0, &io, &mo); mm_set_opt(preset, &io, &mo);
	if (p.kmer_length > 0) io.k = (short)p.kmer_length;
	mo.flag |= MM_F_OUT_CG | MM_F_CIGAR;
	int s = p.indel_len_threshold - 10; if (s < 5) s = 5;
	mo.min_dp_max = s;
	mo.flag |= MM_F_ALL_CHAINS | MM_F_NO_DIAG | MM_F_NO_DUAL | MM_F_NO_LJOIN;
	io.bucket_bits = 14;
	if (mm_check_opt(&io, &mo) != 0) throw std::runtime_error("minimap2: mm_check_opt(): options are invalid");
}

extern "C" int pga_align_groups(const pga_params_t *params, int32_t n_groups, const int64_t *group_off,
                                const char *const *seqs, const uint32_t *seq_lens, const char *const *names, pga_result_t **out)
{
	*out = nullptr;
	try {
		require_device();
		mm_idxopt_t io; mm_mapopt_t mo0;
		params_to_opts(*params, io, mo0);
		std::unique_ptr<pga_result_s> R(new pga_result_s());
		memset(&R->st, 0, sizeof(R->st));
		double t_all = now_s();
		// groups are packed into sub-batches of at most max_bases bases (32-bit minimizer / anchor indices per batch)
		const uint64_t max_bases = getenv("PGA_MAX_BATCH_BASES") ? strtoull(getenv("PGA_MAX_BATCH_BASES"), 0, 10) : 3000000000ULL;
		int g0 = 0;
		while (g0 < n_groups) {
			int g1 = g0; uint64_t bases = 0;
			while (g1 < n_groups) {
				uint64_t gb = 0; for (int64_t i = group_off[g1]; i < group_off[g1 + 1]; ++i) gb += seq_lens[i];
				if (g1 > g0 && bases + gb > max_bases) break;
				bases += gb; ++g1;
			}
			const int64_t b = group_off[g0], n = group_off[g1] - b;
			std::vector<int64_t> goff((size_t)(g1 - g0) + 1);
			for (int g = g0; g <= g1; ++g) goff[g - g0] = group_off[g] - b;
			if (n > 0) {
				std::unique_ptr<PgaIdx> ix(idx_build(io.w, io.k, (int)n, seqs + b, seq_lens + b, names + b, g1 - g0, goff.data()));
				mm_mapopt_t mo = mo0;
				if (mo.bw_long < mo.bw) mo.bw_long = mo.bw;
				run_batch(*ix, mo, params->n_threads);
				collect_results(*ix, g0, goff, *R);
			}
			g0 = g1;
		}
		R->st.total = now_s() - t_all; R->st.n_matches = (double)R->m.size();
		*out = R.release();
		return 0;
	} catch (std::exception &e) { set_err(e.what()); return -1; }
}

// ---- resident batches: upload once, align (possibly many times) with the bases already in HBM ----
struct BatchesInFlight { BatchesInFlight() { batch_call_enter(); } ~BatchesInFlight() { batch_call_leave(); } };
struct pga_batch_s { std::vector<std::unique_ptr<PgaIdx>> parts; std::vector<int> g0; std::vector<std::vector<int64_t>> goff; int w = 0, k = 0; uint64_t bases = 0; };

static int batch_create_impl(const pga_batch_s *old, const int64_t *src_index, int32_t n_groups, const int64_t *group_off, const char *const *seqs, const uint32_t *seq_lens, const char *const *names, pga_batch_t **out);
extern "C" int pga_batch_create(int32_t n_groups, const int64_t *group_off, const char *const *seqs, const uint32_t *seq_lens, const char *const *names, pga_batch_t **out)
{
	return batch_create_impl(nullptr, nullptr, n_groups, group_off, seqs, seq_lens, names, out);
}
// The next self-merge round of a merge (graph_merging.rs:26-69) maps the merged graph: most blocks are the blocks of the round before.
// A sequence with seqs[i] == NULL is taken from `old` (its src_index[i]-th sequence, in the order they were handed over) by a device-to-
// device copy of its packed bases; only the others cross PCIe.  The reference re-copies every sequence as an ASCII C string on every
// find_matches call (packages/minimap2/src/index.rs:31-37, map.rs:377-381).  `old` stays valid and may be freed afterwards.
extern "C" int pga_batch_derive(const pga_batch_t *old, int32_t n_groups, const int64_t *group_off, const char *const *seqs, const int64_t *src_index, const uint32_t *seq_lens,
                                const char *const *names, pga_batch_t **out)
{
	if (!old || !src_index) { *out = nullptr; set_err("pga_batch_derive: null batch or index array"); return -1; }
	return batch_create_impl(old, src_index, n_groups, group_off, seqs, seq_lens, names, out);
}
static int batch_create_impl(const pga_batch_s *old, const int64_t *src_index, int32_t n_groups, const int64_t *group_off, const char *const *seqs, const uint32_t *seq_lens, const char *const *names, pga_batch_t **out)
{
	*out = nullptr;
	try {
		// where the sequences of the old batch lie: (part, index in the part) by their global index
		std::vector<int64_t> old_first;
		if (old) { old_first.push_back(0); for (auto &p : old->parts) old_first.push_back(old_first.back() + p->S.n_seq); }
		std::vector<SeqFrom> from; std::vector<const uint8_t*> from_probe;
		if (old) {
			const int64_t n_all = group_off[n_groups];
			from.assign((size_t)n_all, SeqFrom{PkBases{nullptr, nullptr}, 0}); from_probe.assign((size_t)n_all, nullptr);
			for (int64_t i = 0; i < n_all; ++i) if (!seqs[i]) {
				const int64_t g = src_index[i];
				if (g < 0 || g >= old_first.back()) throw std::runtime_error("pga_batch_derive: source index outside the old batch");
				const size_t p = (size_t)(std::upper_bound(old_first.begin(), old_first.end(), g) - old_first.begin()) - 1;
				const SeqSet &S = old->parts[p]->S; const size_t li = (size_t)(g - old_first[p]);
				if (S.len[li] != seq_lens[i]) throw std::runtime_error("pga_batch_derive: length of a derived sequence differs from its source");
				from[(size_t)i] = SeqFrom{S.bases(), S.off[li]}; from_probe[(size_t)i] = S.probe.data() + li * 64;
			}
		}
		std::unique_ptr<pga_batch_s> B(new pga_batch_s());
		// Sub-batches: at most PGA_MAX_BATCH_BASES each (32-bit anchor indices), and -- when the batch is large enough -- at
		// least PGA_PARTS of them (default 1; 2 pays on very large batches): the parts are aligned CONCURRENTLY, each on its own stream, so the
		// dependency-bound phases of one part (sort replay, chain sweep, the last few DP problems) overlap the others' work
		uint64_t max_bases = getenv("PGA_MAX_BATCH_BASES") ? strtoull(getenv("PGA_MAX_BATCH_BASES"), 0, 10) : 3000000000ULL;
		{
			uint64_t total = 0; for (int64_t i = 0; i < group_off[n_groups]; ++i) total += seq_lens[i];
			// about a Gbp per part, up to three: the parts run concurrently, and every stage of this path is bound by some dependency chain
			// (sort replay, chain sweep, DP rounds), so what overlaps is what counts
			int want = getenv("PGA_PARTS") ? atoi(getenv("PGA_PARTS")) : (int)std::min<uint64_t>(3, std::max<uint64_t>(1, total / 1000000000ULL));
			const int need = (int)((total + max_bases - 1) / max_bases);      // parts of EQUAL size (3.5 Gbp -> 1.75 + 1.75, not 3.0 + 0.5)
			if (need > want) want = need;
			if (want > 1 && n_groups >= want && total >= 200000000ULL) max_bases = std::min<uint64_t>(max_bases, (total + want - 1) / want + 1);
		}
		int g0 = 0;
		while (g0 < n_groups) {
			int g1 = g0; uint64_t bases = 0;
			while (g1 < n_groups) {
				uint64_t gb = 0; for (int64_t i = group_off[g1]; i < group_off[g1 + 1]; ++i) gb += seq_lens[i];
				if (g1 > g0 && bases + gb > max_bases) break;
				bases += gb; ++g1;
			}
			const int64_t b = group_off[g0], n = group_off[g1] - b;
			std::vector<int64_t> goff((size_t)(g1 - g0) + 1);
			for (int g = g0; g <= g1; ++g) goff[g - g0] = group_off[g] - b;
			if (n > 0) {
				// w,k are not known yet: upload only (w=k=1 placeholders are overwritten by pga_batch_align)
				B->parts.emplace_back(idx_build(1, 1, (int)n, seqs + b, seq_lens + b, names + b, g1 - g0, goff.data(), false, old ? from.data() + b : nullptr, old ? from_probe.data() + b : nullptr));
				B->g0.push_back(g0); B->goff.push_back(goff); B->bases += bases;
			}
			g0 = g1;
		}
		*out = B.release();
		return 0;
	} catch (std::exception &e) { set_err(e.what()); return -1; }
}

static int batch_align_impl(pga_batch_t *B, const pga_params_t *params, int shard, int n_shards, pga_result_t **out);
extern "C" int pga_batch_align(pga_batch_t *B, const pga_params_t *params, pga_result_t **out) { return batch_align_impl(B, params, 0, 1, out); }
// "replicate the index on each GPU and split the queries into contiguous ranges balanced by their lengths" -- for waves with
// fewer groups than ranks.  Every shard indexes ALL sequences of the batch and maps the queries of its range of every group; the match
// lists of the shards are disjoint and their union, ordered by (group, query), is the list of the unsharded call.
extern "C" int pga_batch_align_shard(pga_batch_t *B, const pga_params_t *params, int32_t shard, int32_t n_shards, pga_result_t **out)
{
	if (n_shards < 1 || shard < 0 || shard >= n_shards) { *out = nullptr; set_err("pga_batch_align_shard: shard index outside [0, n_shards)"); return -1; }
	return batch_align_impl(B, params, shard, n_shards, out);
}
static int batch_align_impl(pga_batch_t *B, const pga_params_t *params, int shard, int n_shards, pga_result_t **out)
{
	*out = nullptr;
	try {
		mm_idxopt_t io; mm_mapopt_t mo0;
		params_to_opts(*params, io, mo0);
		std::unique_ptr<pga_result_s> R(new pga_result_s());
		memset(&R->st, 0, sizeof(R->st));
		double t_all = now_s();
		const int n_parts = (int)B->parts.size();
		BatchesInFlight in_flight;                 // calls of other host threads count as concurrent parts (ready-set schedules keep several batches in flight)
		int threads_each = params->n_threads > 0 ? params->n_threads : usable_cpus();
		threads_each = std::max(1, threads_each / std::max(1, std::min(n_parts, 3)));
		std::vector<std::string> errs((size_t)n_parts);
		require_device();                                    // a fresh host thread sits on device 0: every entry point applies pga_set_device()'s device
		int dev = 0; PGA_HIP(hipGetDevice(&dev));
		auto work = [&](int p) {
			try {
				PGA_HIP(hipSetDevice(dev));
				PgaIdx &ix = *B->parts[p];
				ArenaScope arena_scope(ix.arena);
				set_part_concurrency(std::min(n_parts, 3));
				if (!ix.indexed || ix.hdr.w != io.w || ix.hdr.k != io.k) {
					ix.hdr.w = io.w, ix.hdr.k = io.k, ix.hdr.b = 14 < 2 * io.k ? 14 : 2 * io.k;
					ix.mid_occ_frac = -1.0f; ix.have_results = false;
				}
				const double up = ix.tm.upload; ix.tm = Timers(); ix.tm.upload = up;
				ix.sharded = n_shards > 1;
				if (ix.sharded) {
					// contiguous query ranges per group, balanced by length: query i of a group goes to the shard its midpoint falls into
					std::vector<uint8_t> own((size_t)ix.S.n_seq, 0);
					for (int g = 0; g < ix.S.n_grp; ++g) {
						const int64_t b = ix.S.grp_off[(size_t)g], e = ix.S.grp_off[(size_t)g + 1];
						uint64_t total = 0; for (int64_t i = b; i < e; ++i) total += ix.S.len[(size_t)i];
						uint64_t cum = 0;
						for (int64_t i = b; i < e; ++i) {
							const uint64_t mid = cum + ix.S.len[(size_t)i] / 2;
							const int sh = total ? (int)std::min<uint64_t>((uint64_t)n_shards - 1, (unsigned __int128)mid * (uint64_t)n_shards / total) : 0;
							own[(size_t)i] = sh == shard ? 1 : 0;
							cum += ix.S.len[(size_t)i];
						}
					}
					ix.d_own.upload(own, ix.st);
				}
				idx_sketch_index(ix);          // the index is part of the hot path: rebuilt on every call, like every find_matches does
				mm_mapopt_t mo = mo0;
				if (mo.bw_long < mo.bw) mo.bw_long = mo.bw;
				run_batch(ix, mo, threads_each);
				// minimizers and index are rebuilt by every call: give their memory back before the next part starts
				ix.M.mz.release(); ix.I.key.release(); ix.I.occ_off.release(); ix.I.occ.release(); ix.I.key_grp.release(); ix.grp.release(); ix.indexed = false;
			} catch (std::exception &e) { errs[p] = e.what(); if (errs[p].empty()) errs[p] = "unknown error"; }
		};
		if (n_parts == 1) work(0);
		else {
			// at most three parts at a time: their dependency-bound phases overlap, and the device memory of three parts (resident
			// arrays + DP scratch) stays bounded whatever the number of parts
			std::atomic<int> next(0);
			auto runner = [&] { for (;;) { const int p = next.fetch_add(1); if (p >= n_parts) break; work(p); } };
			std::vector<std::thread> th;
			const int conc = std::min(n_parts, getenv("PGA_PART_CONCURRENCY") ? std::max(1, atoi(getenv("PGA_PART_CONCURRENCY"))) : 3);
			for (int t = 0; t < conc; ++t) th.emplace_back(runner);
			for (auto &t : th) t.join();
		}
		for (int p = 0; p < n_parts; ++p) if (!errs[p].empty()) throw std::runtime_error(errs[p]);
		for (int p = 0; p < n_parts; ++p) {
			PgaIdx &ix = *B->parts[p];
			collect_results(ix, B->g0[p], B->goff[p], *R);
			ix.results.clear(); ix.have_results = false;
		}
		R->st.total = now_s() - t_all; R->st.n_matches = (double)R->m.size();
		*out = R.release();
		return 0;
	} catch (std::exception &e) { set_err(e.what()); return -1; }
}
extern "C" void pga_batch_free(pga_batch_t *B) { apply_default_device(); delete B; }

extern "C" int64_t pga_result_n_matches(const pga_result_t *r) { return (int64_t)r->m.size(); }
extern "C" const pga_match_t *pga_result_matches(const pga_result_t *r) { return r->m.data(); }
extern "C" const uint32_t *pga_result_cigars(const pga_result_t *r, uint64_t *n_ops) { if (n_ops) *n_ops = r->cig.size(); return r->cig.data(); }
extern "C" const pga_stats_t *pga_result_stats(const pga_result_t *r) { return &r->st; }
extern "C" void pga_result_free(pga_result_t *r) { delete r; }
extern "C" const char *pga_last_error(void) { return g_err.c_str(); }
extern "C" int pga_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
extern "C" int pga_set_device(int dev) { if (hipSetDevice(dev) != hipSuccess) { set_err("hipSetDevice failed"); return -1; } g_default_dev.store(dev); return 0; }
extern "C" void pga_free(void *p) { free(p); }
// n streams are created now, one after the other, and left in the pool the batch handles take their streams from: the runtime spreads
// streams over its hardware queues in creation order, so the n handles a host keeps in flight start on n different queues
extern "C" int pga_warm_streams(int32_t n)
{
	try {
		require_device();
		std::vector<hipStream_t> v;
		for (int i = 0; i < n; ++i) v.push_back(pga::stream_lease());
		for (auto it = v.rbegin(); it != v.rend(); ++it) pga::stream_release(*it);
		return 0;
	} catch (std::exception &e) { set_err(e.what()); return -1; }
}
extern "C" int pga_stats_version(void) { return PGA_STATS_VERSION; }
namespace pga { size_t dp_trim_lane_sets(); void dp_lane_dump(); }
extern "C" int64_t pga_trim(void)
{
	apply_default_device();
	long long lv0[2], lv1[2]; dev_mem_levels(lv0);
	const size_t slabs = dp_trim_lane_sets();       // (idle sets only; their slabs go to the block cache, which the next line empties)
	dev_trim();
	dev_mem_levels(lv1);
	(void)slabs;
	return (int64_t)((lv0[0] + lv0[1]) - (lv1[0] + lv1[1]));
}
extern "C" void pga_mem_stats(int64_t out[6])
{
	if (getenv("PGA_MEM_DUMP")) { dev_mem_dump(); dp_lane_dump(); }
	long long a[4], b[2]; pga::dev_mem_stats(a); pga::dev_mem_levels(b);
	for (int i = 0; i < 4; ++i) out[i] = a[i];
	out[4] = b[0], out[5] = b[1];
}

// ---- busy intervals (pga_common.h: busy_note) ----
static_assert(pga::K_COUNT == PGA_N_KERNELS, "pga_stats_t and the busy log number the kernel families alike");
namespace pga {
struct BusyLog { std::mutex mu; std::atomic<bool> open{false}; hipEvent_t ref = nullptr; int dev = -1; std::vector<std::array<float, 3>> iv; };   // (kernel, start ms, end ms) since `ref`
static BusyLog g_busy;
void busy_note(int kern, hipEvent_t a, hipEvent_t b)
{
	if (!g_busy.open.load(std::memory_order_acquire)) return;      // (an interval next to begin / end may be missed, none is corrupted)
	float t0 = 0, t1 = 0;
	std::lock_guard<std::mutex> lk(g_busy.mu);
	if (!g_busy.open || !g_busy.ref) return;
	if (hipEventElapsedTime(&t0, g_busy.ref, a) != hipSuccess || hipEventElapsedTime(&t1, g_busy.ref, b) != hipSuccess) { (void)hipGetLastError(); return; }
	g_busy.iv.push_back({(float)kern, t0, t1});
}
}
extern "C" int pga_busy_begin(void)
{
	apply_default_device();
	std::lock_guard<std::mutex> lk(pga::g_busy.mu);
	if (pga::g_busy.ref) { (void)hipEventDestroy(pga::g_busy.ref); pga::g_busy.ref = nullptr; }
	if (hipEventCreate(&pga::g_busy.ref) != hipSuccess || hipEventRecord(pga::g_busy.ref, nullptr) != hipSuccess || sync_event(pga::g_busy.ref) != hipSuccess) { set_err("pga_busy_begin: no reference event"); return -1; }
	pga::g_busy.iv.clear(); pga::g_busy.open = true;
	return 0;
}
extern "C" int pga_busy_end(double *busy_ms, int32_t n)
{
	std::vector<std::array<float, 3>> iv;
	{ std::lock_guard<std::mutex> lk(pga::g_busy.mu); pga::g_busy.open = false; iv.swap(pga::g_busy.iv); }
	if (!busy_ms || n < PGA_N_KERNELS + 1) { set_err("pga_busy_end: room for PGA_N_KERNELS + 1 values needed"); return -1; }
	auto union_ms = [&](int kern) {
		std::vector<std::pair<float, float>> v;
		for (auto &x : iv) if (kern < 0 || (int)x[0] == kern) v.emplace_back(x[1], x[2]);
		std::sort(v.begin(), v.end());
		double tot = 0; float lo = 0, hi = -1;
		for (auto &p : v) { if (hi < lo || p.first > hi) { if (hi >= lo) tot += hi - lo; lo = p.first; hi = p.second; } else if (p.second > hi) hi = p.second; }
		if (hi >= lo) tot += hi - lo;
		return tot;
	};
	for (int k = 0; k < PGA_N_KERNELS; ++k) busy_ms[k] = union_ms(k);
	busy_ms[PGA_N_KERNELS] = union_ms(-1);
	return (int)iv.size();
}

// ---------------------------------------------------------------- stage taps (parity tests)
template <class T> static T *dup_out(const std::vector<T> &v) { T *p = (T*)malloc((v.size() ? v.size() : 1) * sizeof(T)); if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T)); return p; }

extern "C" int pga_stage_sketch(int32_t n, const char *const *seqs, const uint32_t *lens, int w, int k, uint64_t **mz_xy, uint64_t **seq_off)
{
	try {
		require_device();
		SeqSet S; Minimizers M;
		const int64_t one_grp[2] = {0, n};
		upload_seqs(S, n, seqs, lens, nullptr, 1, one_grp, 0);
		sketch_all(S, w, k, M, 0);
		std::vector<u128> h = M.mz.download(0); h.resize(M.n);
		std::vector<uint64_t> flat(h.size() * 2);
		for (size_t i = 0; i < h.size(); ++i) flat[2 * i] = h[i].x, flat[2 * i + 1] = h[i].y;
		*mz_xy = dup_out(flat); *seq_off = dup_out(M.h_seq_off);
		return 0;
	} catch (std::exception &e) { set_err(e.what()); return -1; }
}

extern "C" int pga_stage_chain(const pga_params_t *params, int32_t n, const char *const *seqs, const uint32_t *lens, const char *const *names,
                               uint64_t **anchors_xy, uint64_t **anchor_off, int32_t **n_u, int32_t **n_v, uint64_t **u, uint64_t **chain_xy, int32_t **rep_len, int32_t *mid_occ)
{
	try {
		mm_idxopt_t io; mm_mapopt_t mo;
		params_to_opts(*params, io, mo);
		std::unique_ptr<PgaIdx> ix(idx_build(io.w, io.k, n, seqs, lens, names));
		mm_mapopt_update(&mo, &ix->hdr);
		*mid_occ = mo.mid_occ;
		check_supported(mo, io.k, io.w);
		ix->d_mid_occ.upload(group_mid_occ(*ix, mo), ix->st);
		// the tap shows the reference's anchor order (PGA_STAGE_SPECULATIVE=1: the anchors in stable order and the chains by the tie-order-independent
		// route of chain_all, as a batch computes them)
		const bool spec_tap = getenv("PGA_STAGE_SPECULATIVE") != nullptr;
		SeedResult SR; seed_all(ix->S, ix->M, ix->I, ix->grp, mo, ix->d_name_rank, ix->d_mid_occ, SR, ix->st, nullptr, nullptr, !spec_tap);
		std::vector<u128> a = SR.a.download(ix->st); a.resize(SR.n_a);
		std::vector<uint64_t> flat(a.size() * 2);
		for (size_t i = 0; i < a.size(); ++i) flat[2 * i] = a[i].x, flat[2 * i + 1] = a[i].y;
		*anchors_xy = dup_out(flat); *anchor_off = dup_out(SR.h_q_aoff); *rep_len = dup_out(SR.h_rep_len);
		ChainResult CR; chain_all(ix->S, SR, mo, io.k, CR, ix->st, nullptr, !spec_tap);
		*n_u = dup_out(CR.n_u); *n_v = dup_out(CR.n_v);
		std::vector<uint64_t> uu(CR.u.begin(), CR.u.end()); uu.resize(SR.n_a);
		*u = dup_out(uu);
		std::vector<uint64_t> cf(SR.n_a * 2);
		for (size_t i = 0; i < SR.n_a && i < CR.a.size(); ++i) cf[2 * i] = CR.a[i].x, cf[2 * i + 1] = CR.a[i].y;
		*chain_xy = dup_out(cf);
		return 0;
	} catch (std::exception &e) { set_err(e.what()); return -1; }
}

extern "C" int pga_stage_extd2(int32_t n_jobs, const uint8_t *const *q, const int32_t *qlen, const uint8_t *const *t, const int32_t *tlen,
                               int a, int b, int sc_ambi, int gapo, int gape, int gapo2, int gape2, const int32_t *w, const int32_t *zdrop, const int32_t *end_bonus, const int32_t *flag,
                               int32_t *ez, uint32_t **cigars, uint64_t *cigar_off)
{
	try {
		require_device();
		// lay the explicit sequences (base codes 0..4) out one behind the other, query i then target i, and pack them as the resident store is packed
		std::vector<uint8_t> buf; std::vector<DpJob> jobs((size_t)n_jobs);
		for (int i = 0; i < n_jobs; ++i) {
			DpJob &j = jobs[i]; memset(&j, 0, sizeof(j));
			j.q_off = buf.size(); buf.insert(buf.end(), q[i], q[i] + qlen[i]);
			j.t_off = buf.size(); buf.insert(buf.end(), t[i], t[i] + tlen[i]);
			j.qlen_full = qlen[i], j.qs = 0, j.qlen = qlen[i], j.tlen = tlen[i], j.w = w[i], j.zdrop = zdrop[i], j.end_bonus = end_bonus[i], j.flag = flag[i];
		}
		buf.resize((buf.size() + 15) / 16 * 16 + 128, 4);
		std::vector<uint32_t> pk(buf.size() / 16, 0u); std::vector<uint16_t> nm(buf.size() / 16, 0);
		for (size_t i = 0; i < buf.size(); ++i) { const uint8_t c = buf[i]; if (c > 3) nm[i >> 4] |= (uint16_t)(1u << (i & 15)); else pk[i >> 4] |= (uint32_t)c << (2 * (i & 15)); }
		DBuf<uint32_t> d_pk; d_pk.upload(pk, 0); DBuf<uint16_t> d_nm; d_nm.upload(nm, 0);
		const PkBases d{d_pk.p, d_nm.p};
		a = a < 0 ? -a : a; b = b > 0 ? -b : b; sc_ambi = sc_ambi > 0 ? -sc_ambi : sc_ambi;
		DpParams P{gapo, gape, gapo2, gape2, a, b, sc_ambi, dp_lb_mode()};
		std::vector<DpJob> run; std::vector<int> idx;
		for (int i = 0; i < n_jobs; ++i) if (qlen[i] > 0 && tlen[i] > 0) run.push_back(jobs[i]), idx.push_back(i);
		std::vector<DpRes> res; PinVec<uint32_t> cg;
		dp_run(d, run, P, res, cg, 0);
		std::vector<uint32_t> all;
		for (int i = 0; i < n_jobs; ++i) { int32_t *e = ez + 12 * i; e[0] = 0; e[1] = e[2] = -1; e[3] = -0x40000000; e[4] = -1; e[5] = -0x40000000; e[6] = -1; e[7] = -0x40000000; e[8] = e[9] = e[10] = e[11] = 0; cigar_off[i] = 0; }
		for (size_t r = 0; r < res.size(); ++r) {
			int i = idx[r]; int32_t *e = ez + 12 * i; const DpRes &R = res[r];
			e[0] = R.max, e[1] = R.max_q, e[2] = R.max_t, e[3] = R.mqe, e[4] = R.mqe_t, e[5] = R.mte, e[6] = R.mte_q, e[7] = R.score, e[8] = R.zdropped, e[9] = R.reach_end, e[10] = R.n_cigar;
			cigar_off[i] = all.size();
			all.insert(all.end(), cg.begin() + R.cigar_off, cg.begin() + R.cigar_off + R.n_cigar);
		}
		*cigars = dup_out(all);
		return 0;
	} catch (std::exception &e) { set_err(e.what()); return -1; }
}

// ---------------------------------------------------------------- split_matches + filter_matches on the device (pga_filter.hip)
namespace pga {
struct FilterParams { int32_t thr, flags; double alpha, beta; };
void filter_matches_dev(int64_t n_in, const pga_match_t *h_m, const uint32_t *h_cig, uint64_t n_ops_in, const FilterParams &fp, std::vector<pga_match_t> &out_m, std::vector<uint32_t> &out_cig);
}

extern "C" int pga_filter_matches(int64_t n, const pga_match_t *matches, const uint32_t *cigars, uint64_t n_ops, const pga_filter_params_t *fp, pga_result_t **out)
{
	try {
		require_device();
		if (!fp || !out) throw std::runtime_error("pga_filter_matches: null argument");
		if (fp->indel_len_threshold < 1) throw std::runtime_error("pga_filter_matches: indel_len_threshold must be positive");
		std::unique_ptr<pga_result_s> R(new pga_result_s());
		memset(&R->st, 0, sizeof(R->st));
		FilterParams p{fp->indel_len_threshold, fp->flags, fp->alpha, fp->beta};
		filter_matches_dev(n, matches, cigars, n_ops, p, R->m, R->cig);
		R->st.n_matches = (double)R->m.size();
		*out = R.release();
		return 0;
	} catch (std::exception &e) { set_err(e.what()); return -1; }
}

extern "C" int pga_result_filter(const pga_result_t *res, const pga_filter_params_t *fp, pga_result_t **out)
{
	if (!res) { set_err("pga_result_filter: null result"); return -1; }
	return pga_filter_matches((int64_t)res->m.size(), res->m.data(), res->cig.data(), (uint64_t)res->cig.size(), fp, out);
}

// ---------------------------------------------------------------- mash distance + neighbor-joining guide tree (pga_mash.hip)
namespace pga {
void mash_stage_sketch(int n, const char *const *seqs, const uint32_t *lens, int k, int w, std::vector<uint64_t> &val, std::vector<uint64_t> &pos, std::vector<uint64_t> &off);
void mash_distance_host(int n, const char *const *seqs, const uint32_t *lens, int k, int w, double *dist, int32_t *merges);
void nj_host(int n, const double *dist, int32_t *merges);
void nj_last_near(int32_t *count, int32_t *first);
}

extern "C" int pga_stage_mash_sketch(int32_t n, const char *const *seqs, const uint32_t *lens, int k, int w, uint64_t **value, uint64_t **position, uint64_t *seq_off)
{
	try {
		require_device();
		std::vector<uint64_t> v, p, o;
		mash_stage_sketch(n, seqs, lens, k, w, v, p, o);
		*value = dup_out(v); *position = dup_out(p);
		for (int i = 0; i <= n; ++i) seq_off[i] = o[(size_t)i];
		return 0;
	} catch (std::exception &e) { set_err(e.what()); return -1; }
}

extern "C" int pga_mash_distance(int32_t n, const char *const *seqs, const uint32_t *lens, int k, int w, double *dist)
{
	try { require_device(); mash_distance_host(n, seqs, lens, k, w, dist, nullptr); return 0; }
	catch (std::exception &e) { set_err(e.what()); return -1; }
}

extern "C" int pga_guide_tree_nj(int32_t n, const double *dist, int32_t *merges)
{
	try { require_device(); nj_host(n, dist, merges); return 0; }
	catch (std::exception &e) { set_err(e.what()); return -1; }
}

extern "C" int pga_guide_tree(int32_t n, const char *const *seqs, const uint32_t *lens, int k, int w, double *dist, int32_t *merges)
{
	try { require_device(); mash_distance_host(n, seqs, lens, k, w, dist, merges); return 0; }
	catch (std::exception &e) { set_err(e.what()); return -1; }
}
extern "C" int32_t pga_nj_near_ties(int32_t *first_join)
{
	int32_t c = 0, f = -1; pga::nj_last_near(&c, &f);
	if (first_join) *first_join = f;
	return c;
}


// ---------------------------------------------------------------- map_variations over all member sequences (pga_mapvar.hip)
namespace pga {
void map_variations_host(int64_t n, const pga_mapvar_job_t *jobs, const pga_mapvar_params_t &prm, pga_mapvar_res_t *res,
                         std::vector<pga_sub_t> &subs, std::vector<pga_del_t> &dels, std::vector<pga_ins_t> &inss, std::vector<char> &seq);
}

extern "C" int pga_map_variations(int64_t n_jobs, const pga_mapvar_job_t *jobs, const pga_mapvar_params_t *params, pga_mapvar_res_t *res,
                                  pga_sub_t **subs, pga_del_t **dels, pga_ins_t **inss, char **ins_seq)
{
	try {
		require_device();
		if (n_jobs < 0 || (n_jobs && (!jobs || !res)) || !params || !subs || !dels || !inss || !ins_seq) throw std::runtime_error("pga_map_variations: null argument");
		std::vector<pga_sub_t> s; std::vector<pga_del_t> d; std::vector<pga_ins_t> i; std::vector<char> q;
		if (n_jobs) { memset(res, 0, sizeof(*res) * (size_t)n_jobs); map_variations_host(n_jobs, jobs, *params, res, s, d, i, q); }
		*subs = dup_out(s); *dels = dup_out(d); *inss = dup_out(i); *ins_seq = dup_out(q);
		return 0;
	} catch (std::exception &e) { set_err(e.what()); return -1; }
}

// ---------------------------------------------------------------- reconsensus (pga_reconsensus.hip)
namespace pga {
void reconsensus_host(int64_t n_blocks, const pga_rc_block_t *blocks, const pga_rc_member_t *members, const pga_sub_t *subs, const pga_del_t *dels, const pga_ins_t *inss, const char *ins_seq,
                      const pga_mapvar_params_t &prm, pga_rc_out_t *out);
}
extern "C" void pga_rc_free(pga_rc_out_t *o)
{
	if (!o) return;
	free(o->blocks); free(o->members); free(o->subs); free(o->dels); free(o->inss); free(o->ins_seq); free(o->m_subs); free(o->m_dels); free(o->m_inss); free(o->m_ins_seq); free(o->cons);
	memset(o, 0, sizeof(*o));
}
extern "C" int pga_reconsensus(int64_t n_blocks, const pga_rc_block_t *blocks, const pga_rc_member_t *members, const pga_sub_t *subs, const pga_del_t *dels,
                               const pga_ins_t *inss, const char *ins_seq, const pga_mapvar_params_t *params, pga_rc_out_t *out)
{
	if (!out) { set_err("pga_reconsensus: null output"); return -1; }
	memset(out, 0, sizeof(*out));
	try { require_device(); reconsensus_host(n_blocks, blocks, members, subs, dels, inss, ins_seq, *params, out); return 0; }
	catch (std::exception &e) { pga_rc_free(out); set_err(e.what()); return -1; }
}

extern "C" int pga_stage_sort(int32_t n_seg, const uint64_t *seg_off, uint64_t *xy)
{
	try {
		require_device();
		if (n_seg <= 0) return 0;
		const uint64_t n = seg_off[n_seg];
		std::vector<int64_t> len((size_t)n_seg); std::vector<uint32_t> flag((size_t)n_seg, 1u);
		for (int s = 0; s < n_seg; ++s) len[s] = (int64_t)(seg_off[s + 1] - seg_off[s]);
		DBuf<u128> a; a.upload(reinterpret_cast<const u128*>(xy), (size_t)n, 0);
		DBuf<uint64_t> off; off.upload(seg_off, (size_t)n_seg + 1, 0);
		DBuf<int64_t> dl; dl.upload(len, 0);
		DBuf<uint32_t> df; df.upload(flag, 0);
		replay_sort_segments(a.p, n, off.p, dl.p, n_seg, df.p, 0);
		if (n) { PGA_HIP(hipMemcpyAsync(xy, a.p, n * sizeof(u128), hipMemcpyDeviceToHost, 0)); PGA_HIP(sync_stream(0)); }
		return 0;
	} catch (std::exception &e) { set_err(e.what()); return -1; }
}
