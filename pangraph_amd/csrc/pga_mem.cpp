// pga_mem.cpp -- caching device allocator behind DBuf.
//
// A batch allocates a few dozen device arrays per stage and the DP stage a scratch slab of tens of GB; hipMalloc and
// hipFree cost from 0.1 ms to tens of ms each and hipFree synchronises the device, so freed blocks are kept in
// per-device, size-rounded free lists and handed out again (a level of `pangraph build` repeats the same sizes call
// after call).  The cache is bounded: idle blocks may fill what the live blocks leave of 90 % of the device memory (and at most PGA_CACHE_GB,
// default 200); beyond that the blocks that have been idle longest (of any pool) are released.  hipFree synchronises the device, so a cache that is too
// small costs far more than the memory it saves.
#include "pga_common.h"
#include <cstdlib>
#include <map>
#include <string>
#include <mutex>
#include <atomic>
#include <chrono>

// The extension stage keeps several independent launches in flight, each on a stream of its own.  HIP multiplexes streams onto
// GPU_MAX_HW_QUEUES hardware queues (default 4) and launches that share a queue run back to back: 6 queues measured 6 % faster per step than
// 4 on MI355X, 12 and more slow every kernel down.  The runtime reads the variable when the process makes its first HIP
// call, so the library sets the default when it is loaded -- for hosts that are not Python too (pangraph_amd/__init__.py does the same);
// a value the host exported itself stays.
// The variable is process-wide: every other HIP user of the host process sees it.  PGA_KEEP_RUNTIME_DEFAULTS=1 leaves the runtime alone.
__attribute__((constructor(101))) static void pga_runtime_defaults() { if (!getenv("PGA_KEEP_RUNTIME_DEFAULTS")) setenv("GPU_MAX_HW_QUEUES", "6", 0); }

namespace pga {

namespace {
struct Pool {
	std::multimap<size_t, void*> idle;              // rounded size -> block
	size_t idle_bytes = 0;
};
struct Live { size_t size; int dev, arena; };
std::mutex g_mu;
std::map<std::pair<int, int>, Pool> g_pools;       // (device, arena)
std::map<void*, Live> g_live;                      // block -> size and home pool
size_t g_idle_total = 0;
thread_local int t_arena = 0;
// idle blocks by age: when the cache is over its limit the blocks that have lain idle longest go first (the hoard of sizes that never came back),
// not the largest (the DP slabs: sixteen GB and a second apiece to buy back)
struct IdleRef { std::pair<int, int> pool; size_t size; void *p; };
std::map<unsigned long long, IdleRef> g_idle_by_age;
std::map<void*, unsigned long long> g_idle_seq;
unsigned long long g_seq = 0;
void idle_note(const std::pair<int, int> &pool, size_t size, void *p) { const unsigned long long q = ++g_seq; g_idle_seq[p] = q; g_idle_by_age[q] = IdleRef{pool, size, p}; }
void idle_forget(void *p) { auto it = g_idle_seq.find(p); if (it != g_idle_seq.end()) { g_idle_by_age.erase(it->second); g_idle_seq.erase(it); } }

size_t round_size(size_t b)
{
	if (b < 256) b = 256;
	size_t p = 256;
	while (p < b) p <<= 1;                           // next power of two ...
	if (b <= (1u << 20)) return p;
	const size_t step = b <= ((size_t)64 << 20) ? p >> 4 : p >> 2;   // ... refined to 1/16 steps above 1 MB, 1/4 steps above 64 MB (batch sizes drift from
	return (b + step - 1) / step * step;                                // call to call: coarse classes keep the large blocks reusable)
}
size_t g_live_total = 0;                            // bytes handed out and not yet freed
std::atomic<long long> g_n_malloc(0), g_ns_malloc(0), g_n_free(0), g_ns_free(0);   // driver calls behind the cache (diagnostics)
struct NsScope { std::atomic<long long> &n, &ns; std::chrono::steady_clock::time_point t0; NsScope(std::atomic<long long> &n_, std::atomic<long long> &ns_) : n(n_), ns(ns_), t0(std::chrono::steady_clock::now()) {}
	~NsScope() { ++n; ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } };
size_t cache_limit()                               // (called with g_mu held)
{
	static size_t cap = [] { const char *e = getenv("PGA_CACHE_GB"); double g = e ? atof(e) : 200.0; return (size_t)(g * (double)(1ull << 30)); }();
	static size_t dev_total = [] { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) != hipSuccess) tot = (size_t)256 << 30; return tot; }();
	// PGA_MEM_SHARE: the part of the device this process may fill (several processes on one device, e.g. the single-device debugging mode of
	// bench.py: each keeps a cache of its own, and the runtime aborts a queue when nothing is left for its own needs)
	static const double share = [] { const char *e = getenv("PGA_MEM_SHARE"); const double v = e ? atof(e) : 1.0; return v > 0.0 && v <= 1.0 ? v : 1.0; }();
	const size_t room = (size_t)((double)(dev_total / 100 * 90) * share);
	const size_t lim = room > g_live_total ? room - g_live_total : 0;
	return lim < cap ? lim : cap;
}
}

// Blocks are recycled inside an ARENA only.  A freed block may still be read by kernels queued on the stream of the code
// that freed it; handing it to the same arena keeps every later use behind those kernels in stream order.  Concurrent
// sub-batches (pga_api.cpp) run with one arena and one stream each.
void dev_set_arena(int arena) { t_arena = arena; }
int dev_get_arena() { return t_arena; }

// Arena ids are LEASED: an id belongs to one worker (one stream) at a time, and it goes back to the free list only after that
// worker has synchronised its stream (ArenaLease's owner does), so whoever leases it next may reuse its idle blocks on any
// stream.  Ids are recycled, so the number of pools is bounded by the peak number of concurrent workers.  Arena 0 is the
// default of threads that never leased one (stage taps, single-threaded tests): not safe for concurrent use on several streams.
namespace { std::vector<int> g_arena_free; int g_arena_next = 1; std::map<int, bool> g_arena_leased; }
int dev_lease_arena()
{
	std::lock_guard<std::mutex> lk(g_mu);
	int a;
	if (!g_arena_free.empty()) { a = g_arena_free.back(); g_arena_free.pop_back(); } else a = g_arena_next++;
	g_arena_leased[a] = true;
	return a;
}
void dev_release_arena(int arena)
{
	if (arena <= 0) return;
	std::lock_guard<std::mutex> lk(g_mu);
	g_arena_free.push_back(arena);
	g_arena_leased[arena] = false;
}

static int spin_us() { static const int v = [] { const char *e = getenv("PGA_SPIN_US"); return e ? atoi(e) : 60; }(); return v; }
static inline long long mono_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
hipError_t sync_stream(hipStream_t s)
{
	const int us = spin_us();
	if (us > 0) {
		const long long t_end = mono_ns() + 1000LL * us;
		for (;;) {
			const hipError_t q = hipStreamQuery(s);
			if (q == hipSuccess) return hipSuccess;
			if (q != hipErrorNotReady) { (void)hipGetLastError(); break; }
			if (mono_ns() >= t_end) break;
			__builtin_ia32_pause();
		}
	}
	return hipStreamSynchronize(s);
}
hipError_t sync_event(hipEvent_t e)
{
	const int us = spin_us();
	if (us > 0) {
		const long long t_end = mono_ns() + 1000LL * us;
		for (;;) {
			const hipError_t q = hipEventQuery(e);
			if (q == hipSuccess) return hipSuccess;
			if (q != hipErrorNotReady) { (void)hipGetLastError(); break; }
			if (mono_ns() >= t_end) break;
			__builtin_ia32_pause();
		}
	}
	return hipEventSynchronize(e);
}

void *dev_alloc(size_t bytes)
{
	int dev = 0;
	PGA_HIP(hipGetDevice(&dev));
	const size_t r = round_size(bytes);
	const int arena = t_arena;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		Pool &P = g_pools[{dev, arena}];
		auto it = P.idle.lower_bound(r);
		if (it != P.idle.end() && (it->first <= r + r / 4 || (r >= ((size_t)16 << 20) && it->first <= 2 * r))) {   // large blocks: up to twice the need beats a hipMalloc (and the hipFree it provokes)
			void *p = it->second; const size_t sz = it->first;
			P.idle.erase(it); P.idle_bytes -= sz; g_idle_total -= sz; g_live[p] = Live{sz, dev, arena}; g_live_total += sz; idle_forget(p);
			return p;
		}
		// nothing in this arena: the idle blocks of arenas that are not leased at the moment are free for all (their owners synchronised
		// their streams before giving the arena back); the block changes its home to this arena
		if (arena != 0) for (auto &kv : g_pools) {
			if (kv.first.first != dev || kv.first.second == 0 || kv.first.second == arena) continue;
			auto ls = g_arena_leased.find(kv.first.second);
			if (ls == g_arena_leased.end() || ls->second) continue;
			Pool &O = kv.second;
			auto jt = O.idle.lower_bound(r);
			if (jt != O.idle.end() && (jt->first <= r + r / 4 || (r >= ((size_t)16 << 20) && jt->first <= 2 * r))) {
				void *p = jt->second; const size_t sz = jt->first;
				O.idle.erase(jt); O.idle_bytes -= sz; g_idle_total -= sz; g_live[p] = Live{sz, dev, arena}; g_live_total += sz; idle_forget(p);
				return p;
			}
		}
	}
	// A miss: the new block counts against the same 90 % of the device as the idle ones -- the cache gives up its oldest blocks FIRST.  (The limit used to
	// be looked at only when a block came back: a process that had filled its cache with the sizes of one workload and then ran another -- the test suite: the
	// level-synchronous C5 build, then the six-slot one -- took the device to its last MB with blocks that all were "live or within the limit when freed",
	// and the runtime aborted the queue whose own allocation failed: HSA_STATUS_ERROR_OUT_OF_RESOURCES.)
	{
		std::vector<void*> drop;
		{
			std::lock_guard<std::mutex> lk(g_mu);
			const size_t lim = cache_limit();                     // what idle blocks may hold beside the live ones
			if (g_idle_total + r > lim) {
				const size_t target = lim > r + ((size_t)2 << 30) ? lim - r - ((size_t)2 << 30) : 0;
				// (oldest first, blocks of THIS device only -- hipFree of another device's block would need that device current -- and at most 64 per miss:
				// every hipFree synchronises the device)
				for (auto at = g_idle_by_age.begin(); at != g_idle_by_age.end() && g_idle_total > target && drop.size() < 64;) {
					const IdleRef v = at->second; ++at;
					if (v.pool.first != dev) continue;
					Pool &V = g_pools[v.pool];
					auto rg = V.idle.equal_range(v.size);
					for (auto jt = rg.first; jt != rg.second; ++jt) if (jt->second == v.p) { V.idle.erase(jt); break; }
					V.idle_bytes -= v.size; g_idle_total -= v.size;
					drop.push_back(v.p);
					idle_forget(v.p);
				}
			}
		}
		for (void *q : drop) { NsScope sc(g_n_free, g_ns_free); (void)hipFree(q); }
	}
	void *p = nullptr;
	hipError_t e;
	{ NsScope sc(g_n_malloc, g_ns_malloc); e = hipMalloc(&p, r); }
	{ static const bool verbose = getenv("PGA_VERBOSE") != nullptr; if (verbose && r >= ((size_t)128 << 20)) fprintf(stderr, "[pga] allocator: hipMalloc of %.0f MB for arena %d (idle in the cache %.1f GB, handed out %.1f GB)\n", r / 1048576.0, arena, g_idle_total / 1073741824.0, g_live_total / 1073741824.0); }
	if (e != hipSuccess) {
		(void)hipGetLastError();
		dev_trim();                                   // give the idle blocks back and retry once
		(void)hipDeviceSynchronize();
		e = hipMalloc(&p, r);
		if (e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " allocating " + std::to_string(r) + " bytes of device memory");
	}
	std::lock_guard<std::mutex> lk(g_mu);
	g_live[p] = Live{r, dev, arena}; g_live_total += r;
	return p;
}

void dev_free(void *p)
{
	if (!p) return;
	std::vector<void*> drop;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		auto it = g_live.find(p);
		if (it == g_live.end()) { drop.push_back(p); }
		else {
			const Live lv = it->second;
			g_live.erase(it); g_live_total -= lv.size;
			Pool &P = g_pools[{lv.dev, lv.arena}];
			P.idle.emplace(lv.size, p); P.idle_bytes += lv.size; g_idle_total += lv.size;
			idle_note({lv.dev, lv.arena}, lv.size, p);
			// over the limit: the blocks that have been idle longest go, whichever pool holds them, until 4 GB below it (hipFree synchronises the
			// device: not on every free from here on)
			if (g_idle_total > cache_limit()) {
				const size_t lim = cache_limit(), target = lim > ((size_t)4 << 30) ? lim - ((size_t)4 << 30) : 0;
				// (oldest first, blocks of THIS device only -- hipFree of another device's block would need that device current -- and at most 64 per call:
				// every hipFree synchronises the device)
				for (auto at = g_idle_by_age.begin(); at != g_idle_by_age.end() && g_idle_total > target && drop.size() < 64;) {
					const IdleRef v = at->second; ++at;
					if (v.pool.first != lv.dev) continue;
					Pool &V = g_pools[v.pool];
					auto rg = V.idle.equal_range(v.size);
					for (auto jt = rg.first; jt != rg.second; ++jt) if (jt->second == v.p) { V.idle.erase(jt); break; }
					V.idle_bytes -= v.size; g_idle_total -= v.size;
					drop.push_back(v.p);
					idle_forget(v.p);
				}
			}
		}
	}
	for (void *q : drop) { NsScope sc(g_n_free, g_ns_free); (void)hipFree(q); }
}

// ---- pooled streams: a batch handle, a query set, a tail query each work on a stream of their own for a few milliseconds; creating and
// destroying a stream costs about as much (hipStreamCreate ... hipStreamDestroy: ~1-3 ms with the synchronisation), so streams are leased.
// A stream goes back drained (the caller synchronises it), so its next user starts on an empty stream.
namespace { std::mutex g_st_mu; std::map<int, std::vector<hipStream_t>> g_st_idle; }
hipStream_t stream_lease()
{
	int dev = 0; PGA_HIP(hipGetDevice(&dev));
	{
		std::lock_guard<std::mutex> lk(g_st_mu);
		auto &v = g_st_idle[dev];
		if (!v.empty()) { hipStream_t s = v.back(); v.pop_back(); return s; }
	}
	hipStream_t s = nullptr;
	static const char *prio = getenv("PGA_MAIN_PRIO");                 // experiment: h / l = the batches' own streams on the high / low priority queue pool
	if (prio && (prio[0] == 'h' || prio[0] == 'l')) {
		int lo = 0, hi = 0; PGA_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
		PGA_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio[0] == 'h' ? hi : lo));
	} else PGA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	return s;
}
void stream_release(hipStream_t s)
{
	if (!s) return;
	(void)hipStreamSynchronize(s);
	int dev = 0; if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamDestroy(s); return; }
	std::lock_guard<std::mutex> lk(g_st_mu);
	g_st_idle[dev].push_back(s);
}

void dev_mem_stats(long long out[4]) { out[0] = g_n_malloc; out[1] = g_ns_malloc; out[2] = g_n_free; out[3] = g_ns_free; }
void dev_mem_levels(long long out[2]) { std::lock_guard<std::mutex> lk(g_mu); out[0] = (long long)g_live_total; out[1] = (long long)g_idle_total; }

// diagnostics (PGA_MEM_DUMP=1, from pga_mem_stats): what every arena holds, live and idle, and the largest idle size classes
void dev_mem_dump()
{
	std::lock_guard<std::mutex> lk(g_mu);
	std::map<std::pair<int, int>, size_t> live;
	for (auto &kv : g_live) live[{kv.second.dev, kv.second.arena}] += kv.second.size;
	fprintf(stderr, "[pga] device memory: %.1f GB live, %.1f GB idle in the cache\n", g_live_total / 1073741824.0, g_idle_total / 1073741824.0);
	for (auto &kv : g_pools) {
		std::map<size_t, int> classes; for (auto &b : kv.second.idle) ++classes[b.first];
		std::string top; int k = 0;
		for (auto it = classes.rbegin(); it != classes.rend() && k < 6; ++it, ++k) top += " " + std::to_string(it->first >> 20) + "MBx" + std::to_string(it->second);
		auto ls = g_arena_leased.find(kv.first.second);
		fprintf(stderr, "[pga]   device %d arena %d%s: live %.2f GB, idle %.2f GB in %zu blocks; largest idle classes:%s\n", kv.first.first, kv.first.second,
		        ls != g_arena_leased.end() && ls->second ? " (leased)" : "", live[kv.first] / 1073741824.0, kv.second.idle_bytes / 1073741824.0, kv.second.idle.size(), top.c_str());
	}
}

void dev_trim()
{
	std::vector<void*> drop;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		for (auto &kv : g_pools) { for (auto &b : kv.second.idle) drop.push_back(b.second); kv.second.idle.clear(); kv.second.idle_bytes = 0; }
		g_idle_total = 0; g_idle_by_age.clear(); g_idle_seq.clear();
	}
	for (void *q : drop) (void)hipFree(q);
}

// ---- pinned host blocks, cached the same way (hipHostMalloc costs milliseconds per hundred MB) ----
namespace {
std::mutex g_pin_mu;
std::multimap<size_t, void*> g_pin_idle;
std::map<void*, size_t> g_pin_live;
size_t g_pin_idle_bytes = 0;
}

void *pin_alloc(size_t bytes)
{
	const size_t r = round_size(bytes);
	{
		std::lock_guard<std::mutex> lk(g_pin_mu);
		auto it = g_pin_idle.lower_bound(r);
		if (it != g_pin_idle.end() && it->first <= r + r / 4) {
			void *p = it->second; const size_t sz = it->first;
			g_pin_idle.erase(it); g_pin_idle_bytes -= sz; g_pin_live[p] = sz;
			return p;
		}
	}
	void *p = nullptr;
	PGA_HIP(hipHostMalloc(&p, r, hipHostMallocDefault));
	std::lock_guard<std::mutex> lk(g_pin_mu);
	g_pin_live[p] = r;
	return p;
}

void pin_free(void *p)
{
	if (!p) return;
	std::vector<void*> drop;
	{
		std::lock_guard<std::mutex> lk(g_pin_mu);
		auto it = g_pin_live.find(p);
		if (it == g_pin_live.end()) drop.push_back(p);
		else {
			const size_t sz = it->second;
			g_pin_live.erase(it);
			g_pin_idle.emplace(sz, p); g_pin_idle_bytes += sz;
			const size_t lim = (size_t)16 << 30;
			while (g_pin_idle_bytes > lim && !g_pin_idle.empty()) {
				auto big = std::prev(g_pin_idle.end());
				drop.push_back(big->second); g_pin_idle_bytes -= big->first; g_pin_idle.erase(big);
			}
		}
	}
	for (void *q : drop) (void)hipHostFree(q);
}

} // namespace pga
