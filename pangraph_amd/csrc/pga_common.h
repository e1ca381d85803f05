// pga_common.h -- shared declarations of libpgalign.so (HIP backend, gfx950 only).
//
// Stage-separated pairwise block-alignment backend for pangraph's `build`:
//   sketch -> index -> seed/anchor -> chain -> regions -> banded dual-affine DP -> records.
// Device data is structure-of-arrays, sized for one whole batch ("level") at a time.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <stdexcept>
#include "../../include/pga_mm2_abi.h"

#define PGA_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
	throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " __FILE__ ":" + std::to_string(__LINE__)); } while (0)

namespace pga {

// ---- waiting for the device (pga_mem.cpp) ----
// The library asks the runtime for BLOCKING waits (pga_api.cpp: require_device): a thread that waits holds no core.  A blocked thread comes back ~100 us
// after its kernel has ended, though, and a call of the upper tree waits two dozen times for kernels of a few microseconds: these two first POLL the stream /
// event for a bounded time (PGA_SPIN_US, default 60 us; 0: never) and block only when the work outlasts it -- the short waits cost what they cost under the
// runtime's spinning default, the long ones no core.
hipError_t sync_stream(hipStream_t s);
hipError_t sync_event(hipEvent_t e);

// ---- device memory: cached blocks (pga_mem.cpp) ----
void *dev_alloc(size_t bytes);
void dev_free(void *p);
void dev_trim();        // release every idle block
void dev_mem_dump();    // diagnostics to stderr: live and idle bytes per arena
void dev_mem_levels(long long out[2]);   // bytes handed out, bytes idle in the cache
void dev_mem_stats(long long out[4]);   // hipMalloc calls, ns spent in them, hipFree calls, ns (since the library was loaded)
void dev_set_arena(int arena);
int dev_get_arena();   // calling thread: recycle device blocks only within this arena (one per concurrent sub-batch)
int dev_lease_arena();                // a globally unique arena id until dev_release_arena(): one worker (one stream) at a time
void dev_release_arena(int arena);    // call after the worker's stream has been synchronised
struct ArenaScope {                   // the calling thread allocates in `arena` while the scope lives
	int prev; explicit ArenaScope(int arena) : prev(dev_get_arena()) { dev_set_arena(arena); } ~ArenaScope() { dev_set_arena(prev); }
	ArenaScope(const ArenaScope&) = delete; ArenaScope &operator=(const ArenaScope&) = delete;
};

// ---- device buffer ----
template <class T> struct DBuf {
	T *p = nullptr; size_t n = 0, cap = 0;
	DBuf() {}
	explicit DBuf(size_t n_) { alloc(n_); }
	DBuf(const DBuf&) = delete; DBuf &operator=(const DBuf&) = delete;
	DBuf(DBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
	DBuf &operator=(DBuf &&o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; } return *this; }
	~DBuf() { release(); }
	void release() { if (p && cap) dev_free(p); p = nullptr; n = cap = 0; }
	void view(T *ptr, size_t n_) { release(); p = ptr; n = n_; cap = 0; }     // a window into somebody else's block (cap == 0: never freed from here)
	void alloc(size_t n_) { // contents undefined
		if (n_ > cap) { release(); p = (T*)dev_alloc(n_ * sizeof(T)); cap = n_; }
		n = n_;
	}
	void zero(hipStream_t s = 0) { if (n) PGA_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
	void upload(const T *h, size_t n_, hipStream_t s = 0) { alloc(n_); if (n_) PGA_HIP(hipMemcpyAsync(p, h, n_ * sizeof(T), hipMemcpyHostToDevice, s)); }
	void upload(const std::vector<T> &h, hipStream_t s = 0) { upload(h.data(), h.size(), s); }
	std::vector<T> download(hipStream_t s = 0) const {
		std::vector<T> h(n);
		if (n) { PGA_HIP(hipMemcpyAsync(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost, s)); PGA_HIP(sync_stream(s)); }
		return h;
	}
};

struct u128 { uint64_t x, y; };

// ---- pinned host memory: cached blocks (pga_mem.cpp); device-to-host copies into it run at full DMA speed ----
void *pin_alloc(size_t bytes);
void pin_free(void *p);
template <class T> struct PinVec {          // the small subset of std::vector the pipeline needs
	T *p = nullptr; size_t n = 0, cap = 0;
	PinVec() {}
	PinVec(const PinVec&) = delete; PinVec &operator=(const PinVec&) = delete;
	PinVec(PinVec &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
	PinVec &operator=(PinVec &&o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; } return *this; }
	~PinVec() { release(); }
	void release() { if (p) pin_free(p); p = nullptr; n = cap = 0; }
	void resize(size_t n_) { if (n_ > cap) { T *q = (T*)pin_alloc((n_ ? n_ : 1) * sizeof(T)); if (n) memcpy(q, p, n * sizeof(T)); if (p) pin_free(p); p = q; cap = n_; } n = n_; }
	void clear() { n = 0; }
	size_t size() const { return n; }
	bool empty() const { return n == 0; }
	T *data() { return p; } const T *data() const { return p; }
	T *begin() { return p; } const T *begin() const { return p; }
	T *end() { return p + n; } const T *end() const { return p + n; }
	T &operator[](size_t i) { return p[i]; } const T &operator[](size_t i) const { return p[i]; }
};
template <class T> static inline void download_to(PinVec<T> &h, const T *d, size_t n, hipStream_t s)
{
	h.resize(n);
	if (n) { PGA_HIP(hipMemcpyAsync(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost, s)); PGA_HIP(sync_stream(s)); }
}

// several device arrays to the host behind ONE synchronisation (each DBuf::download() pays a copy and a synchronisation of its own)
struct Downloads {
	hipStream_t st;
	struct Item { void *dst; void *pin; size_t bytes; };
	std::vector<Item> items;
	explicit Downloads(hipStream_t s) : st(s) {}
	template <class T> void add(std::vector<T> &dst, const T *src, size_t n)
	{
		dst.resize(n);
		if (!n) return;
		Item it{dst.data(), pin_alloc(n * sizeof(T)), n * sizeof(T)};
		PGA_HIP(hipMemcpyAsync(it.pin, src, it.bytes, hipMemcpyDeviceToHost, st));
		items.push_back(it);
	}
	void wait() { if (!items.empty()) PGA_HIP(sync_stream(st)); for (Item &it : items) { memcpy(it.dst, it.pin, it.bytes); pin_free(it.pin); } items.clear(); }
	~Downloads() { for (Item &it : items) pin_free(it.pin); }
};

// ---- the sequence set of one batch, resident in HBM ----

// The resident sequence store (pga_sketch.hip: k_encode_pk): 2 bits per base, sixteen bases per 32-bit word, plus one "not ACGT" bit per
// base -- every kernel of the path reads bases from it (0..3 = ACGT, 4 = anything else: the codes of sketch.c:9-26).
struct PkBases {
	const uint32_t *pk2; const uint16_t *nmask;
	__device__ __forceinline__ int at(uint64_t pos) const
	{
		const uint64_t w = pos >> 4; const uint32_t sft = (uint32_t)pos & 15u;
		const uint32_t b = pk2[w], n = nmask[w];
		return ((n >> sft) & 1u) ? 4 : (int)((b >> (2u * sft)) & 3u);
	}
	// sixteen bases from base position pos on (any alignment): w = their 2-bit codes (base i at bits 2i; 00 where the base is not ACGT),
	// m = their "not ACGT" bits.  (The stores are padded: the word behind the last one exists.)
	__device__ __forceinline__ void window16(uint64_t pos, uint32_t &w, uint32_t &m) const
	{
		const uint64_t i = pos >> 4; const uint32_t sft = (uint32_t)pos & 15u;
		const uint64_t ww = (uint64_t)pk2[i] | (uint64_t)pk2[i + 1] << 32;
		w = (uint32_t)(ww >> (2u * sft));
		const uint32_t mm = (uint32_t)nmask[i] | (uint32_t)nmask[i + 1] << 16;
		m = (mm >> sft) & 0xffffu;
	}
};

struct SeqSet {
	int n_seq = 0;
	uint64_t total = 0;                 // sum of lengths
	std::vector<uint64_t> off;          // n_seq+1 offsets into the concatenation of all sequences (base positions of the packed store)
	std::vector<uint32_t> len;
	std::vector<std::string> name;
	std::vector<uint8_t> probe;         // 64 base codes per sequence at evenly spaced positions (mm_map's identity check); the host keeps no other copy of the bases
	DBuf<uint32_t> d_pk2;               // the packed store: 2 bits per base, sixteen bases per word (word i = bases 16i .. 16i+15 of the concatenation)
	DBuf<uint16_t> d_nmask;             // ... and one bit per base: 1 = not ACGT
	PkBases bases() const { return PkBases{d_pk2.p, d_nmask.p}; }
	DBuf<uint64_t> d_off;
	DBuf<uint32_t> d_len;
	// groups: independent all-vs-all problems sharing the batch (one group == one find_matches call)
	int n_grp = 1;
	std::vector<int64_t> grp_off;       // n_grp+1 offsets into the sequence arrays
	std::vector<uint32_t> grp_of_seq;   // n_seq
	DBuf<uint32_t> d_grp_of_seq, d_grp_base;   // per sequence: its group, and the first sequence of that group
	DBuf<uint8_t> d_tables;                     // d_off, d_len, d_grp_of_seq and d_grp_base are windows into this block: ONE host-to-device copy per batch
	PinVec<uint8_t> h_tables;                   // ... from this pinned image (kept: the copy is asynchronous)
};

// ---- minimizers of a SeqSet (sorted by sequence, then by position == mm_sketch output order) ----
struct Minimizers {
	DBuf<u128> mz;                      // x = hash<<8|k, y = rid<<32|pos<<1|strand
	DBuf<uint64_t> seq_off;             // n_seq+1
	std::vector<uint64_t> h_seq_off;
	uint64_t n = 0;
};

// ---- index: distinct hashes ascending + CSR occurrence lists (y ascending inside a key) ----
struct Index {
	int w = 0, k = 0;
	uint64_t n_keys = 0, n_occ = 0;
	DBuf<uint64_t> key;
	DBuf<uint32_t> occ_off;             // n_keys+1
	DBuf<uint64_t> occ;
	DBuf<uint32_t> key_grp;             // group of every key (keys are sorted by (group, hash); nothing reads the order, but the lists of a group lie together)
};

// device kernels timed with HIP events (order is part of pga_stats_t, include/pga_align.h)
enum { K_SKETCH = 0, K_CHAIN = 1, K_BACKTRACK = 2, K_EXTD2 = 3, K_EXTD2_WIDE = 4 /* <256> */, K_LL = 5, K_SORT = 6, K_BAND = 7, K_WIDE512 = 8, K_WIDE1024 = 9,
       K_INDEX = 10 /* index build: device sorts + CSR kernels */, K_SEED = 11 /* seeding kernels + anchor sort */, K_STRIPS = 12 /* k_wstrips, k_bstrips, k_approx_strips */, K_LANES = 13 /* k_extd2_lanes */, K_PIPE = 14 /* k_ext_pipe */, K_COUNT = 16 };
struct KernelStat { double ms = 0, launches = 0, alg_bytes = 0, cells = 0; };   // cells: DP cells the kernel's loops evaluated
struct Timers {
	double upload = 0, sketch = 0, index = 0, seed = 0, chain = 0, align = 0, total = 0, dp_jobs = 0, dp_cells = 0, n_mz = 0, n_anchor = 0, dp_bases = 0, dp_cigar_ops = 0;
	KernelStat kern[K_COUNT];   // device time of the path's own kernels, measured with HIP events on the launch stream
};
// Busy intervals (pga_busy_begin / pga_busy_end, pga_api.cpp): while a log is open, every event-bracketed launch of kernel family `kern`
// leaves its interval [a, b] on the device's clock; the union per family and over all families is what a step really spent with that
// kernel resident, however many streams and batches overlapped (the sums of kern[].ms do not add up to wall time).  Both events complete.
void busy_note(int kern, hipEvent_t a, hipEvent_t b);
// leased non-blocking streams of the current device (pga_mem.cpp); release drains the stream
hipStream_t stream_lease();
void stream_release(hipStream_t s);
// times everything enqueued on `st` between construction and stop()
// (mark() closes the interval without waiting; finish() reads it -- behind a synchronisation the caller needs anyway it costs nothing, where stop()
// makes the host wait for the kernels it has just queued before it may queue the next ones)
struct EventTimer {
	hipEvent_t a, b; hipStream_t st; bool marked = false;
	explicit EventTimer(hipStream_t s) : st(s) { PGA_HIP(hipEventCreate(&a)); PGA_HIP(hipEventCreate(&b)); PGA_HIP(hipEventRecord(a, st)); }
	void mark() { if (!marked) { PGA_HIP(hipEventRecord(b, st)); marked = true; } }
	double finish(int kern = -1) { float ms = 0; mark(); PGA_HIP(sync_event(b)); PGA_HIP(hipEventElapsedTime(&ms, a, b)); if (kern >= 0) busy_note(kern, a, b); return ms; }
	double stop(int kern = -1) { mark(); return finish(kern); }
	~EventTimer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
};

// stage entry points (each in its own .hip/.cpp)
// number of CPUs this process may run on (the affinity mask, not the machine total)
int usable_cpus();
// host threads the calling thread's batch may use (pga_params_t.n_threads; 0 = usable_cpus())
void set_thread_budget(int n);
int thread_budget();
// how many batches (parts) the calling thread's batch shares the device with: their rounds already overlap, so each runs one query set
void set_part_concurrency(int n);
int part_concurrency();             // concurrent parts of this call, or batch calls in flight in the process, whichever is larger
void batch_call_enter(); void batch_call_leave();
struct SeqFrom { PkBases store; uint64_t pos; };       // where a sequence that is already resident lies (pga_batch_derive)
// a loop over the library's pool of persistent helper threads (pga_align.cpp): the caller takes part, helpers join as they are free.  Starting std::threads
// per loop costs a stack mapping and its removal each (and the address-space lock of the whole process while six batches fault pages in).
void pool_for_raw(size_t n, int n_threads, void (*run)(void*, size_t), void *ctx);
template <class F> static inline void pool_for(size_t n, int n_threads, F f) { pool_for_raw(n, n_threads, [](void *c, size_t i) { (*static_cast<F*>(c))(i); }, &f); }
void upload_seqs(SeqSet &S, int n, const char *const *seq, const uint32_t *len, const char *const *name, int n_grp, const int64_t *grp_off, hipStream_t st,
                 const SeqFrom *from = nullptr, const uint8_t *const *from_probe = nullptr);
void sketch_all(const SeqSet &S, int w, int k, Minimizers &M, hipStream_t st, Timers *tm = nullptr);
std::vector<int32_t> index_cal_max_occ(const SeqSet &S, const Index &I, float f, hipStream_t st);

} // namespace pga
