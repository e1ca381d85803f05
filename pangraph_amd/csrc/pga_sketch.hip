// pga_sketch.hip -- kernel #1: (w,k) symmetric minimizers of every sequence of a batch.
//
// Replaces mm_sketch() (reference: packages/minimap2-sys/minimap2/sketch.c:77-143), which the reference
// calls once per sequence at index time (index.c:449) and again per query (map.c:66).  Bit-exact output:
// the same records x = hash64(min(fwd,rev))<<8 | k, y = rid<<32 | lastPos<<1 | strand, in the same order.
//
// CDNA4 design.  The reference is a streaming state machine; here the emission rules are evaluated per
// base position, independently (the position-parallel definition is spelled out in oracle/pgo_sketch.c):
//   * one 256-thread workgroup per tile of TILE bases (+ a w+k halo), bases read with coalesced 16-byte
//     loads and packed to 2 bits/base + 1 N-bit/base in LDS;
//   * the forward/reverse k-mer words of a position are bit-field extracts of the packed LDS image
//     (reverse strand = complement of the little-endian extract; forward = its 2-bit-group reversal);
//   * the run length of valid bases (reset by N) is a count-leading-zeros on the N-bit window;
//   * the window minimum (rightmost on ties, sketch.c:123,131) is a w-wide scan of 64-bit hashes in LDS;
//   * emitted records are counted, block-scanned (wave64 shuffles) and written in position order to a
//     per-tile staging slab; a second kernel compacts slabs into the final array.
// HBM traffic: 1 B/base in + 16 B/minimizer staged + 16 B/minimizer re-read + 16 B/minimizer out.
// The fast path needs an odd k (no k-mer equals its reverse complement, so every base is a window slot)
// and w+k <= 64; any other (w,k) runs the serial kernel at the bottom (one lane per sequence), exact for
// all inputs, slow, and never hit by pangraph's presets except through -K with an even k.
#include <chrono>
#include "pga_common.h"
#include "pga_wave.h"
#include <sched.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <algorithm>
#include <rocprim/rocprim.hpp>

namespace pga {

__device__ __forceinline__ uint64_t hash64(uint64_t key, uint64_t mask) // sketch.c:28-38
{
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

int usable_cpus()
{
	// the smallest of: PGA_THREADS, the affinity mask, the online CPUs, the cgroup v2 CPU quota
	if (const char *e = getenv("PGA_THREADS")) { const int v = atoi(e); if (v > 0) return v; }
	int c = (int)std::thread::hardware_concurrency();
	if (c <= 0) c = 1;
	cpu_set_t set;
	CPU_ZERO(&set);
	if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0 && a < c) c = a; }
	const long onl = sysconf(_SC_NPROCESSORS_ONLN);
	if (onl > 0 && onl < c) c = (int)onl;
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		long long quota = -1, period = -1;
		char q[64];
		if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) { quota = atoll(q); const int lim = (int)((quota + period - 1) / period); if (lim > 0 && lim < c) c = lim; }
		fclose(f);
	}
	return c;
}

// The gather of a hand-over writes pinned memory that only the DMA engine reads: streaming stores (no read of the destination line, nothing of it left
// in the caches).  PGA_STREAM_COPY=0: plain memcpy.
#if !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
static void stream_copy(uint8_t *dst, const uint8_t *src, size_t n)
{
	static const bool on = !(getenv("PGA_STREAM_COPY") && atoi(getenv("PGA_STREAM_COPY")) == 0);
	if (!on || n < 1024) { memcpy(dst, src, n); return; }
	const size_t head = (64 - ((uintptr_t)dst & 63)) & 63;
	memcpy(dst, src, head); dst += head, src += head, n -= head;
	for (size_t i = n >> 6; i; --i, src += 64, dst += 64) {
		_mm_prefetch((const char*)src + 1024, _MM_HINT_NTA);
		const __m128i a = _mm_loadu_si128((const __m128i*)src), b = _mm_loadu_si128((const __m128i*)(src + 16));
		const __m128i c = _mm_loadu_si128((const __m128i*)(src + 32)), d = _mm_loadu_si128((const __m128i*)(src + 48));
		_mm_stream_si128((__m128i*)dst, a); _mm_stream_si128((__m128i*)(dst + 16), b);
		_mm_stream_si128((__m128i*)(dst + 32), c); _mm_stream_si128((__m128i*)(dst + 48), d);
	}
	memcpy(dst, src, n & 63);
}
static inline void stream_fence() { _mm_sfence(); }
#else
static void stream_copy(uint8_t *dst, const uint8_t *src, size_t n);
static inline void stream_fence() {}
#endif
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static thread_local int t_thread_budget = 0;
void set_thread_budget(int n) { t_thread_budget = n; }
int thread_budget() { return t_thread_budget > 0 ? t_thread_budget : usable_cpus(); }
static thread_local int t_part_conc = 1;
static std::atomic<int> g_batch_calls(0);      // pga_batch_align calls in flight in this process (ready-set schedules keep several going)
void set_part_concurrency(int n) { t_part_conc = n > 0 ? n : 1; }
int part_concurrency() { const int g = g_batch_calls.load(std::memory_order_relaxed); return t_part_conc > g ? t_part_conc : g; }
void batch_call_enter() { g_batch_calls.fetch_add(1); }
void batch_call_leave() { g_batch_calls.fetch_sub(1); }

static inline uint8_t nt4_host(uint8_t r)
{
	uint8_t c = r & 0xdf;
	uint8_t code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : (c == 'T' || c == 'U') ? 3 : 4;
	if (r < 0x40) code = 4;
	return code;
}

// ASCII -> bases (sketch.c:9-26 table: A/a 0, C/c 1, G/g 2, T/t/U/u 3, everything else 4), sixteen bases per thread
__device__ __forceinline__ uint32_t nt4_of(uint32_t r)
{
	const uint32_t c = r & 0xdf;
	uint32_t code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : (c == 'T' || c == 'U') ? 3u : 4u;
	if (r < 0x40) code = 4u;
	return code;
}
// The resident store: 2 bits per base, sixteen bases per 32-bit word, plus one "not ACGT" bit per base (the reference packs its index
// sequences at index time too, four bits per base: index.c:438-446, mmpriv.h:30-31).  Every kernel of the path reads bases from it
// (PkBases, pga_common.h): 0.375 bytes per base written here, no byte-per-base copy exists.
__global__ __launch_bounds__(256)
void k_encode_pk(const uint4 *__restrict__ raw, uint32_t *__restrict__ pk2, uint16_t *__restrict__ nmask, uint64_t n16)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n16) return;
	const uint4 v = raw[i];
	const uint32_t in[4] = {v.x, v.y, v.z, v.w};
	uint32_t bits = 0, nm = 0;
#pragma unroll
	for (int j = 0; j < 16; ++j) { const uint32_t c = nt4_of((in[j >> 2] >> (8 * (j & 3))) & 0xff); bits |= (c & 3u) << (2 * j); nm |= (c > 3 ? 1u : 0u) << j; }
	pk2[i] = bits; nmask[i] = (uint16_t)nm;
}

// The hand-over of a batch: the caller's ASCII sequences go to the device as they are -- host threads gather them into pinned
// staging buffers, chunk by chunk, while the previous chunk is on its way over PCIe -- and are encoded THERE (k_encode_pk).  The
// host keeps no copy of the bases: nothing on the host reads them (pga_align.cpp), except 64 probe positions per sequence that let
// mm_map() check that a query really is the indexed sequence of that name.
// sequences that are already resident (pga_batch_derive): seq[i] == nullptr, their bases are old store `from[i].store` at base position
// `from[i].pos`; they are copied device to device into their place of the new store after the host's sequences are encoded
__global__ void k_repack(uint32_t *__restrict__ pk2, uint16_t *__restrict__ nmask, uint64_t n_words, const uint64_t *__restrict__ off, int n_seq, const SeqFrom *__restrict__ from)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= n_words) return;
	const uint64_t p0 = w << 4;
	if (p0 >= off[n_seq]) return;
	int lo = 0, hi = n_seq - 1;                                        // the sequence that holds base p0
	while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (off[mid] <= p0) lo = mid; else hi = mid - 1; }
	if (p0 + 16 <= off[lo + 1]) {                                      // the whole word lies in one sequence: a shifted copy of two source words
		const SeqFrom F = from[lo];
		if (!F.store.pk2) return;
		uint32_t ww, mm;
		F.store.window16(F.pos + (p0 - off[lo]), ww, mm);
		pk2[w] = ww; nmask[w] = (uint16_t)mm;
		return;
	}
	uint32_t bits = pk2[w], nm = nmask[w];
	bool touched = false;
	for (uint32_t j = 0; j < 16; ++j) {
		const uint64_t p = p0 + j;
		while (lo < n_seq && off[lo + 1] <= p) ++lo;
		if (lo >= n_seq) break;
		const SeqFrom F = from[lo];
		if (!F.store.pk2) continue;
		const int c = F.store.at(F.pos + (p - off[lo]));
		bits = (bits & ~(3u << (2 * j))) | ((uint32_t)(c & 3) << (2 * j));
		nm = (nm & ~(1u << j)) | ((c > 3 ? 1u : 0u) << j);
		touched = true;
	}
	if (touched) { pk2[w] = bits; nmask[w] = (uint16_t)nm; }
}

void upload_seqs(SeqSet &S, int n, const char *const *seq, const uint32_t *len, const char *const *name, int n_grp, const int64_t *grp_off, hipStream_t st, const SeqFrom *from, const uint8_t *const *from_probe)
{
	const double tu0 = now_s(); double tu_g = 0, tu_w = 0;
	S.n_seq = n;
	S.off.assign((size_t)n + 1, 0); S.len.assign(len, len + n); S.name.resize(n);
	for (int i = 0; i < n; ++i) { S.off[i + 1] = S.off[i] + len[i]; S.name[i] = name && name[i] ? name[i] : ""; }
	S.total = S.off[n];
	S.probe.assign((size_t)n * 64, 4);
	// (the probes of a host sequence are read by the thread that gathers it, below: 64 strided reads per sequence touch half of its cache lines, and
	// did so on one thread before anything was on its way to the device -- 7.7 of the 10.7 ms of an 84 Mbp hand-over)
	for (int i = 0; i < n; ++i) if (!seq[i]) {
		if (!from || !from[i].store.pk2) throw std::runtime_error("pga: sequence without bases");
		if (from_probe && from_probe[i]) memcpy(&S.probe[(size_t)i * 64], from_probe[i], 64);
	}
	const double tu1 = now_s();
	// sequences are padded to a 16-byte multiple (and 64 more) so that wide loads never straddle the allocation
	const uint64_t padded = (S.total + 15) / 16 * 16;
	S.d_pk2.alloc((size_t)(padded / 16) + 8); S.d_nmask.alloc((size_t)(padded / 16) + 8);
	const uint64_t chunk = (uint64_t)16 << 20;      // (two staging buffers: a chunk is gathered while the one before it crosses PCIe)
	const uint64_t n_chunks = (S.total + chunk - 1) / chunk;
	if (n_chunks) {
		// Large hand-overs take turns: the gather is bound by the host's memory bandwidth, so six of them side by side (the six leaf batches a build
		// starts with) all finish late and together -- one after the other the first is on the device after a sixth of that time, and the batches no
		// longer move through their stages in lockstep.  Small ones (the calls of the upper tree) never wait.  PGA_UPLOAD_GATE=0: off; =n: n at a time.
		static const int gate_n = getenv("PGA_UPLOAD_GATE") ? atoi(getenv("PGA_UPLOAD_GATE")) : 1;
		struct Gate { std::mutex mu; std::condition_variable cv; int in = 0; };
		static Gate gate;
		const bool gated = gate_n > 0 && S.total >= ((uint64_t)64 << 20);
		if (gated) { std::unique_lock<std::mutex> lk(gate.mu); gate.cv.wait(lk, [&] { return gate.in < gate_n; }); ++gate.in; }
		struct GateLeave { Gate &g; bool on; ~GateLeave() { if (on) { { std::lock_guard<std::mutex> lk(g.mu); --g.in; } g.cv.notify_one(); } } } gate_leave{gate, gated};
		struct Stage { uint8_t *pin = nullptr; uint8_t *dev = nullptr; hipEvent_t sent; bool used = false; };
		Stage sg[2];
		bool staged = false; uint64_t n_staged = 0;
		const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)thread_budget(), 16));
		for (uint64_t c = 0; c < n_chunks; ++c) {
			const uint64_t b = c * chunk, e = std::min<uint64_t>(S.total, b + chunk);
			if (from) {
				// a chunk that holds only sequences resident elsewhere does not cross PCIe: its words start as "not ACGT" and k_repack fills them
				bool host = false;
				for (int i = (int)(std::upper_bound(S.off.begin(), S.off.end(), b) - S.off.begin()) - 1; i < n && S.off[(size_t)i] < e; ++i) if (seq[i] && S.off[(size_t)i + 1] > b && len[i]) { host = true; break; }
				if (!host) {
					const uint64_t nbw = (e - b + 15) / 16;
					PGA_HIP(hipMemsetAsync(S.d_pk2.p + b / 16, 0, nbw * sizeof(uint32_t), st));
					PGA_HIP(hipMemsetAsync(S.d_nmask.p + b / 16, 0xff, nbw * sizeof(uint16_t), st));
					continue;
				}
			}
			if (!staged) { for (Stage &x : sg) { x.pin = (uint8_t*)pin_alloc(chunk); x.dev = (uint8_t*)dev_alloc(chunk); PGA_HIP(hipEventCreateWithFlags(&x.sent, hipEventDisableTiming)); } staged = true; }
			Stage &x = sg[n_staged++ & 1];
			if (x.used) PGA_HIP(sync_event(x.sent));          // the staging buffer's previous chunk has left the host
			x.used = true;
			// gather [b, e) of the concatenation: every thread copies a contiguous slice
			auto gather = [&](uint64_t lo, uint64_t hi) {
				int i = (int)(std::upper_bound(S.off.begin(), S.off.end(), lo) - S.off.begin()) - 1;
				for (uint64_t p = lo; p < hi;) {
					while (S.off[(size_t)i + 1] <= p) ++i;
					const uint64_t stop = std::min<uint64_t>(hi, S.off[(size_t)i + 1]);
					if (seq[i]) {
						const uint64_t s0 = p - S.off[(size_t)i], s1 = stop - S.off[(size_t)i];         // this slice of sequence i, and the probes that lie in it
						stream_copy(x.pin + (p - b), (const uint8_t*)seq[i] + s0, (size_t)(s1 - s0));
						const uint32_t L = len[i], step = L > 64 ? L / 64 : 1;
						for (uint64_t k = (s0 + step - 1) / step; k < 64 && k * step < s1; ++k) S.probe[(size_t)i * 64 + k] = nt4_host((uint8_t)seq[i][k * step]);
					} else memset(x.pin + (p - b), 'N', (size_t)(stop - p));       // resident elsewhere: filled in by k_repack
					p = stop;
				}
				stream_fence();
			};
			const uint64_t per = ((e - b) + nt - 1) / nt;
			const double tg0 = now_s();
			if (nt == 1 || e - b < (1u << 20)) gather(b, e);
			else pool_for((size_t)nt, nt, [&](size_t t) { const uint64_t lo = std::min(e, b + (uint64_t)t * per), hi = std::min(e, lo + per); if (lo < hi) gather(lo, hi); });
			tu_g += now_s() - tg0;
			const uint64_t nb = (e - b + 15) / 16 * 16;
			if (nb > e - b) memset(x.pin + (e - b), 'N', (size_t)(nb - (e - b)));
			PGA_HIP(hipMemcpyAsync(x.dev, x.pin, (size_t)nb, hipMemcpyHostToDevice, st));
			PGA_HIP(hipEventRecord(x.sent, st));
			hipLaunchKernelGGL(k_encode_pk, dim3((unsigned)((nb / 16 + 255) / 256)), dim3(256), 0, st, (const uint4*)x.dev, S.d_pk2.p + b / 16, S.d_nmask.p + b / 16, nb / 16);
		}
		PGA_HIP(hipGetLastError());
		const double tw0 = now_s();
		PGA_HIP(sync_stream(st));
		tu_w = now_s() - tw0;
		if (staged) for (Stage &x : sg) { pin_free(x.pin); dev_free(x.dev); (void)hipEventDestroy(x.sent); }
	}
	if (getenv("PGA_VERBOSE")) fprintf(stderr, "[pga]   upload: %.1f Mbp, %d threads: names+probes %.2f ms, gather %.2f ms, last wait %.2f ms, all %.2f ms\n", S.total * 1e-6, (int)std::min<uint64_t>((uint64_t)thread_budget(), 16), (tu1 - tu0) * 1e3, tu_g * 1e3, tu_w * 1e3, (now_s() - tu0) * 1e3);
	PGA_HIP(hipMemsetAsync(S.d_pk2.p + padded / 16, 0, 8 * sizeof(uint32_t), st));
	PGA_HIP(hipMemsetAsync(S.d_nmask.p + padded / 16, 0xff, 8 * sizeof(uint16_t), st));
	// the per-sequence tables in one block and one copy (offsets, lengths, group of a sequence, first sequence of its group)
	S.n_grp = n_grp;
	S.grp_off.assign(grp_off, grp_off + n_grp + 1);
	S.grp_of_seq.assign((size_t)n, 0);
	{
		std::vector<uint32_t> base((size_t)n, 0);
		for (int g = 0; g < n_grp; ++g) for (int64_t i = grp_off[g]; i < grp_off[g + 1]; ++i) S.grp_of_seq[i] = (uint32_t)g, base[i] = (uint32_t)grp_off[g];
		const size_t o_off = 0, o_len = ((size_t)n + 1) * 8, o_gos = o_len + (((size_t)n * 4 + 15) & ~(size_t)15), o_base = o_gos + (((size_t)n * 4 + 15) & ~(size_t)15);
		const size_t bytes = o_base + (((size_t)n * 4 + 15) & ~(size_t)15) + 16;
		PinVec<uint8_t> &h = S.h_tables; h.resize(bytes);
		memcpy(h.data() + o_off, S.off.data(), ((size_t)n + 1) * 8);
		if (n) { memcpy(h.data() + o_len, S.len.data(), (size_t)n * 4); memcpy(h.data() + o_gos, S.grp_of_seq.data(), (size_t)n * 4); memcpy(h.data() + o_base, base.data(), (size_t)n * 4); }
		S.d_tables.alloc(bytes);
		PGA_HIP(hipMemcpyAsync(S.d_tables.p, h.data(), bytes, hipMemcpyHostToDevice, st));
		S.d_off.view(reinterpret_cast<uint64_t*>(S.d_tables.p + o_off), (size_t)n + 1);
		S.d_len.view(reinterpret_cast<uint32_t*>(S.d_tables.p + o_len), (size_t)n);
		S.d_grp_of_seq.view(reinterpret_cast<uint32_t*>(S.d_tables.p + o_gos), (size_t)n);
		S.d_grp_base.view(reinterpret_cast<uint32_t*>(S.d_tables.p + o_base), (size_t)n);
	}
	if (from) {
		bool any = false; for (int i = 0; i < n; ++i) any |= !seq[i];
		if (any && S.total) {
			std::vector<SeqFrom> f(from, from + n);
			for (int i = 0; i < n; ++i) if (seq[i]) f[(size_t)i].store = PkBases{nullptr, nullptr};
			DBuf<SeqFrom> d_from; d_from.upload(f, st);
			const uint64_t n_words = padded / 16;
			hipLaunchKernelGGL(k_repack, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, S.d_pk2.p, S.d_nmask.p, n_words, S.d_off.p, n, d_from.p);
			PGA_HIP(hipGetLastError());
			PGA_HIP(sync_stream(st));
		}
	}
}

// ------------------------------------------------------------------------------------------------
#define SK_THREADS 256
#define SK_PER 8
#define SK_TILE (SK_THREADS * SK_PER)   // bases per workgroup
#define SK_HALO 64                      // >= w+k on the fast path
#define SK_NONE 0xffffffffffffffffULL

struct SkTile { uint32_t rid; uint32_t start; };

// extract `nbits` (<= 57) bits starting at bit offset `bit` from a little-endian bit string held in 32-bit LDS words
__device__ __forceinline__ uint64_t lds_bits(const uint32_t *w, uint32_t bit, uint32_t nbits)
{
	uint32_t i = bit >> 5, sh = bit & 31;
	uint64_t lo = (uint64_t)w[i] | (uint64_t)w[i + 1] << 32;
	uint64_t v = lo >> sh;
	if (sh) v |= (uint64_t)w[i + 2] << (64 - sh);
	return nbits >= 64 ? v : v & ((1ULL << nbits) - 1);
}

__device__ __forceinline__ uint64_t rev2(uint64_t x) // reverse the order of the 32 two-bit groups
{
	x = __brevll(x);
	return ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
}

template <int W_MAX>
__global__ __launch_bounds__(SK_THREADS)
void k_sketch_tiles(const uint32_t *__restrict__ pk2, const uint16_t *__restrict__ nmask, const uint64_t *__restrict__ seq_off, const uint32_t *__restrict__ seq_len,
                    const SkTile *__restrict__ tiles, int w, int k, u128 *__restrict__ stage, uint32_t stage_cap,
                    uint32_t *__restrict__ tile_cnt, int *__restrict__ overflow)
{
	constexpr int NPOS = SK_TILE + SK_HALO;                 // positions held: [tile_start-HALO, tile_start+TILE)
	__shared__ uint32_t s_bits[NPOS / 16 + 4];              // 2 bits per base
	__shared__ uint32_t s_nmask[NPOS / 32 + 4];             // 1 bit per base: 1 = not ACGT (or outside the sequence)
	__shared__ uint64_t s_hash[SK_TILE + W_MAX + 1];        // hash of the k-mer ENDING at a position, SK_NONE if none; index 0 <-> tile_start-w-... see HB
	__shared__ uint8_t  s_strand[SK_TILE + W_MAX + 1];
	__shared__ uint16_t s_cur[SK_TILE + 1];                 // rightmost window minimum after each position, index 0 <-> tile_start-1
	__shared__ uint32_t s_wsum8[SK_PER][SK_THREADS / 64];
	__shared__ uint32_t s_base;

	const SkTile tl = tiles[blockIdx.x];
	const uint32_t rid = tl.rid, len = seq_len[rid];
	const int64_t t0 = tl.start;                            // first position of the tile (sequence coordinates)
	const uint64_t goff = seq_off[rid];
	const int tid = threadIdx.x;
	const uint64_t mask = (1ULL << 2 * k) - 1;

	// ---- 1. the 2-bit + N-bit image of [t0-HALO, t0+TILE) comes straight from the packed store (6 bytes per sixteen bases): words are
	//         aligned to sixteen bases of the concatenated array, so the LDS image starts at the aligned position `ga` <= goff+t0-HALO;
	//         positions outside the sequence (the previous / next sequence, the padding) count as "not ACGT".
	const int64_t g_lo = (int64_t)goff + t0 - SK_HALO;      // may be negative or belong to the previous sequence
	const int64_t ga = g_lo >= 0 ? (g_lo & ~15LL) : -(((-g_lo) + 15) & ~15LL);
	const int pad = (int)(g_lo - ga);                        // 0..15: LDS position of g_lo
	for (int c = tid; c < NPOS / 16 + 4; c += SK_THREADS) {
		int64_t g = ga + (int64_t)c * 16;
		uint32_t bits = 0, nm = 0xffffu;
		if (g >= 0 && (uint64_t)g < goff + len + 16) { bits = pk2[g >> 4]; nm = nmask[g >> 4]; }      // (the stores are padded)
		// bases of this word inside the sequence: [lo, hi)
		const int64_t lo64 = (int64_t)goff - g, hi64 = (int64_t)goff + (int64_t)len - g;
		const int lo = lo64 < 0 ? 0 : lo64 > 16 ? 16 : (int)lo64, hi = hi64 < 0 ? 0 : hi64 > 16 ? 16 : (int)hi64;
		const uint32_t inside = hi > lo ? ((hi >= 16 ? 0xffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u)) : 0u;
		nm |= ~inside & 0xffffu;
		{   // a base that does not count contributes 00 (what the byte form did): spread the 16-bit mask to 2 bits per base
			uint32_t m = ~nm & 0xffffu;
			m = (m | m << 8) & 0x00ff00ffu; m = (m | m << 4) & 0x0f0f0f0fu; m = (m | m << 2) & 0x33333333u; m = (m | m << 1) & 0x55555555u;
			bits &= m | m << 1;
		}
		s_bits[c] = bits;
		reinterpret_cast<uint16_t*>(s_nmask)[c] = (uint16_t)nm;
	}
	__syncthreads();

	// ---- 2. per position: k-mer words, run length, hash.  Local position q <-> sequence position t0-HALO+q;
	//         hashes are kept for sequence positions [t0-w-1, t0+TILE): HB = index of position t0-w-1.
	const int nb = w + k;                                     // N-bit window width for the run length
	const int first = SK_HALO - w - 1;                        // local q of the first hashed position (>= k-1 since HALO >= w+k)
	for (int h = tid; h < SK_TILE + w + 1; h += SK_THREADS) {
		int q = first + h;                                    // local position (>= k because HALO >= w+k)
		int lq = pad + q;                                     // position inside the LDS image
		uint64_t win = lds_bits(s_bits, 2u * (uint32_t)(lq - k + 1), 2u * (uint32_t)k);   // base j of the k-mer at bits 2j
		const bool valid = lds_bits(s_nmask, (uint32_t)(lq - k + 1), (uint32_t)k) == 0;  // k ACGT bases end here (run >= k)
		uint64_t rv = (~win) & mask;                          // sketch.c:109
		uint64_t fw = rev2(win) >> (64 - 2 * k);              // sketch.c:108
		uint64_t hv = SK_NONE; uint8_t z = 0;
		if (valid) { z = fw < rv ? 0 : 1; hv = hash64(z ? rv : fw, mask); }
		s_hash[h] = hv; s_strand[h] = z;
	}
	__syncthreads();

	// ---- 3. rightmost window minimum after each position t0-1 .. t0+TILE-1 (hash index of position p is p-(t0-w-1))
	for (int c = tid; c < SK_TILE + 1; c += SK_THREADS) {
		int hi = c + w;                                       // hash index of position t0-1+c
		int best = hi - w + 1; uint64_t bv = s_hash[best];
		for (int j = hi - w + 2; j <= hi; ++j) { uint64_t v = s_hash[j]; if (v <= bv) bv = v, best = j; }
		s_cur[c] = (uint16_t)best;
	}
	__syncthreads();

	// ---- 4. emission rules (see oracle/pgo_sketch.c header: rules A-D), evaluated ONCE per position: the eight positions of a thread (one per
	//         256-position slab) keep their rule flags and counts in registers, the eight wave scans run back to back, one barrier publishes
	//         the 8 x 4 wave totals, and every thread knows where its minimizers go (slab by slab, position order).
	const int tile_n = (int)min((int64_t)SK_TILE, (int64_t)len - t0);
	uint32_t cnt[SK_PER], incl[SK_PER]; int prevv[SK_PER], curv[SK_PER]; uint32_t flg[SK_PER];
#pragma unroll
	for (int j = 0; j < SK_PER; ++j) {
		const int c = j * SK_THREADS + tid;                   // position t0+c
		uint32_t n = 0, f = 0; int prev = 0, cur = 0;
		if (c < tile_n) {
			const int hi = c + w + 1;                         // hash index of this position
			prev = s_cur[c], cur = s_cur[c + 1];
			const uint64_t xp = s_hash[hi], xprev = s_hash[prev];
			int run;
			{   // run length again (cheap): needed for the l-thresholds
				const int lq = pad + SK_HALO + c;
				const uint64_t nwin = lds_bits(s_nmask, (uint32_t)(lq - nb + 1), (uint32_t)nb);
				const uint64_t top = nwin << (64 - nb);
				run = top == 0 ? nb : __clzll((long long)top);
			}
			const bool prev_real = xprev != SK_NONE;
			const bool ruleA = run == w + k - 1 && prev_real;
			bool ruleB = false, ruleC1 = false, ruleC2 = false;
			if (xp <= xprev) ruleB = run >= w + k && prev_real;
			else if (prev == hi - w) { ruleC1 = run >= w + k - 1 && prev_real; ruleC2 = run >= w + k - 1 && s_hash[cur] != SK_NONE; }
			const bool ruleD = (t0 + c == (int64_t)len - 1) && s_hash[cur] != SK_NONE;
			if (ruleA) for (int t = hi - w + 1; t < hi; ++t) n += (s_hash[t] == xprev && t != prev);
			n += ruleB + ruleC1;
			if (ruleC2) { const uint64_t xc = s_hash[cur]; for (int t = hi - w + 1; t <= hi; ++t) n += (s_hash[t] == xc && t != cur); }
			n += ruleD;
			f = (ruleA ? 1u : 0u) | (ruleB || ruleC1 ? 2u : 0u) | (ruleC2 ? 4u : 0u) | (ruleD ? 8u : 0u);
		}
		cnt[j] = n; flg[j] = f; prevv[j] = prev; curv[j] = cur;
		incl[j] = wave_prefix_sum_incl(n);
		if ((tid & 63) == 63) s_wsum8[j][tid >> 6] = incl[j];
	}
	__syncthreads();
	uint32_t running = 0, o[SK_PER];
#pragma unroll
	for (int j = 0; j < SK_PER; ++j) {
		uint32_t wbase = 0, total = 0;
#pragma unroll
		for (int wv = 0; wv < SK_THREADS / 64; ++wv) { const uint32_t sv = s_wsum8[j][wv]; if (wv < (tid >> 6)) wbase += sv; total += sv; }
		o[j] = running + wbase + incl[j] - cnt[j];
		running += total;
	}
	if (tid == 0) { tile_cnt[blockIdx.x] = running; if (running > stage_cap) atomicExch(overflow, 1); }
	if (running == 0) return;
	u128 *out = stage + (size_t)blockIdx.x * stage_cap;
	const uint64_t ybase = (uint64_t)rid << 32;
	const int64_t pos0 = t0 - w - 1;                          // sequence position of hash index 0
#pragma unroll
	for (int j = 0; j < SK_PER; ++j) {
		if (!cnt[j]) continue;
		const int c = j * SK_THREADS + tid, hi = c + w + 1, prev = prevv[j], cur = curv[j];
		uint32_t oo = o[j];
		auto emit = [&](int t) {
			if (oo < stage_cap) { u128 r; r.x = s_hash[t] << 8 | (uint64_t)k; r.y = ybase | (uint64_t)(uint32_t)(pos0 + t) << 1 | s_strand[t]; out[oo] = r; }
			++oo;
		};
		if (flg[j] & 1u) { const uint64_t xprev = s_hash[prev]; for (int t = hi - w + 1; t < hi; ++t) if (s_hash[t] == xprev && t != prev) emit(t); }
		if (flg[j] & 2u) emit(prev);
		if (flg[j] & 4u) { const uint64_t xc = s_hash[cur]; for (int t = hi - w + 1; t <= hi; ++t) if (s_hash[t] == xc && t != cur) emit(t); }
		if (flg[j] & 8u) emit(cur);
	}
}

// ---- the same tiles, eight CONSECUTIVE positions per thread (w known at compile time, k odd, 2k + 13 <= 64) ----
// What the kernel above pays per position -- two k-mer extractions from the LDS image, a scan of the w hashes of its window, w-element
// loops whenever one lane of the wave meets rule C2 -- is paid here per EIGHT positions or not at all:
//   * the two k-mer words roll from base to base (sketch.c:104-110), as does the run of ACGT bases;
//   * a minimizer candidate is ONE 64-bit key, hash << 13 | (4095 - local index) << 1 | strand (all ones: no k-mer): the unsigned minimum of
//     a window is its smallest hash at its rightmost position (the `<=` of sketch.c:127,135), and the key holds all an emission needs;
//   * keys go through LDS once (column-major over the eight positions of a thread: conflict-free both ways); a thread fetches the w keys in
//     front of its own eight and gets its nine window minima from one suffix-minimum and one prefix-minimum chain (w + 16 comparisons);
//   * the same two chains over the keys with the index field flipped give the LEFTMOST minimum: the two differ exactly when the window
//     holds its smallest hash twice -- only then are the "identical k-mers in the window" loops of rules A and C2 (sketch.c:121-125,
//     138-142) run at all;
//   * positions of a thread are consecutive, so one wave scan + one barrier order the emissions of the tile.
#define SK8_HALO 96
template <int W>
__global__ __launch_bounds__(SK_THREADS)
void k_sketch_tiles8(const uint32_t *__restrict__ pk2, const uint16_t *__restrict__ nmask, const uint64_t *__restrict__ seq_off, const uint32_t *__restrict__ seq_len,
                     const SkTile *__restrict__ tiles, int k, u128 *__restrict__ stage, uint32_t stage_cap, uint32_t *__restrict__ tile_cnt, int *__restrict__ overflow)
{
	static_assert(W >= 9 && W <= 32, "the two minimum chains meet at element 8");
	constexpr int HB = (W + 7) / 8;                          // chunks of eight positions in front of the tile whose keys the first windows need
	constexpr int NCH = SK_THREADS + HB, KS = NCH + 1;
	constexpr int NPOS = SK_TILE + SK8_HALO;
	constexpr uint64_t NONE = ~0ULL, FLIP = 0x1ffeULL;
	__shared__ uint32_t s_bits[NPOS / 16 + 4];
	__shared__ uint32_t s_nmask[NPOS / 32 + 4];
	__shared__ uint64_t s_key[8 * KS];                       // key of local index L at [(L & 7) * KS + (L >> 3)]; L = 0 <-> position t0 - 8 * HB
	__shared__ uint32_t s_wsum[SK_THREADS / 64];

	const SkTile tl = tiles[blockIdx.x];
	const uint32_t rid = tl.rid, len = seq_len[rid];
	const int64_t t0 = tl.start;
	const uint64_t goff = seq_off[rid];
	const int tid = threadIdx.x;
	const uint64_t mask = (1ULL << 2 * k) - 1;
	const int sh1 = 2 * (k - 1);

	// ---- 1. the 2-bit + N-bit image of [t0 - HALO, t0 + TILE), as in k_sketch_tiles ----
	const int64_t g_lo = (int64_t)goff + t0 - SK8_HALO;
	const int64_t ga = g_lo >= 0 ? (g_lo & ~15LL) : -(((-g_lo) + 15) & ~15LL);
	const int pad = (int)(g_lo - ga);
	for (int c = tid; c < NPOS / 16 + 4; c += SK_THREADS) {
		const int64_t g = ga + (int64_t)c * 16;
		uint32_t bits = 0, nm = 0xffffu;
		if (g >= 0 && (uint64_t)g < goff + len + 16) { bits = pk2[g >> 4]; nm = nmask[g >> 4]; }
		const int64_t lo64 = (int64_t)goff - g, hi64 = (int64_t)goff + (int64_t)len - g;
		const int lo = lo64 < 0 ? 0 : lo64 > 16 ? 16 : (int)lo64, hi = hi64 < 0 ? 0 : hi64 > 16 ? 16 : (int)hi64;
		const uint32_t inside = hi > lo ? ((hi >= 16 ? 0xffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u)) : 0u;
		nm |= ~inside & 0xffffu;
		{
			uint32_t m = ~nm & 0xffffu;
			m = (m | m << 8) & 0x00ff00ffu; m = (m | m << 4) & 0x0f0f0f0fu; m = (m | m << 2) & 0x33333333u; m = (m | m << 1) & 0x55555555u;
			bits &= m | m << 1;
		}
		s_bits[c] = bits;
		reinterpret_cast<uint16_t*>(s_nmask)[c] = (uint16_t)nm;
	}
	__syncthreads();

	// ---- 2. keys of a chunk of eight positions: rolling k-mer words, rolling run ----
	uint64_t own[8]; int run_at[8];
	auto chunk = [&](int c, bool keep) {
		const int L0 = 8 * c;
		const int lq0 = pad + SK8_HALO - 8 * HB + L0;          // image position of the chunk's first base
		const uint64_t win = lds_bits(s_bits, 2u * (uint32_t)(lq0 - k), 2u * (uint32_t)k);     // the k-mer ending just before the chunk, base j at bits 2j
		uint64_t rv = (~win) & mask, fw = rev2(win) >> (64 - 2 * k);
		const int nb = W + k;
		int run;
		{
			const uint64_t nwin = lds_bits(s_nmask, (uint32_t)(lq0 - nb), (uint32_t)nb);
			const uint64_t top = nwin << (64 - nb);
			run = top == 0 ? nb : __clzll((long long)top);
		}
		const uint32_t b16 = (uint32_t)lds_bits(s_bits, 2u * (uint32_t)lq0, 16u), n8 = (uint32_t)lds_bits(s_nmask, (uint32_t)lq0, 8u);
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const uint64_t cb = (b16 >> (2 * j)) & 3u;
			fw = (fw << 2 | cb) & mask;
			rv = rv >> 2 | (3ULL ^ cb) << sh1;
			run = ((n8 >> j) & 1u) ? 0 : run + 1;
			const uint32_t z = fw < rv ? 0u : 1u;
			const uint64_t h = hash64(z ? rv : fw, mask);
			const uint64_t key = run >= k ? (h << 13 | (uint64_t)(4095 - (L0 + j)) << 1 | z) : NONE;
			s_key[j * KS + c] = key;
			if (keep) { own[j] = key; run_at[j] = run; }
		}
	};
	if (tid < HB) chunk(tid, false);
	chunk(tid + HB, true);
	__syncthreads();

	// ---- 3. the nine windows of the thread: window j ends at local index L0 - 1 + j ----
	const int L0 = 8 * (tid + HB);
	uint64_t e[W + 8];
#pragma unroll
	for (int d = 1; d <= W; ++d) e[W - d] = s_key[((8 - (d & 7)) & 7) * KS + (tid + HB - ((d + 7) >> 3))];
#pragma unroll
	for (int j = 0; j < 8; ++j) e[W + j] = own[j];
	uint64_t R[9]; uint32_t dup = 0;                          // rightmost minimum of window j; bit j: the window holds its smallest hash more than once
	{
		uint64_t sx[9]; sx[8] = NONE;
#pragma unroll
		for (int j = 7; j >= 0; --j) sx[j] = e[j] < sx[j + 1] ? e[j] : sx[j + 1];
		uint64_t px = NONE;
#pragma unroll
		for (int t = 8; t < W + 8; ++t) { px = e[t] < px ? e[t] : px; if (t >= W - 1) { const uint64_t a = sx[t - W + 1]; R[t - W + 1] = a < px ? a : px; } }
	}
	{
		uint64_t sx[9]; sx[8] = NONE;
#pragma unroll
		for (int j = 7; j >= 0; --j) { const uint64_t v = e[j] ^ FLIP; sx[j] = v < sx[j + 1] ? v : sx[j + 1]; }
		uint64_t px = NONE;
#pragma unroll
		for (int t = 8; t < W + 8; ++t) {
			const uint64_t v = e[t] ^ FLIP; px = v < px ? v : px;
			if (t >= W - 1) { const uint64_t a = sx[t - W + 1]; const uint64_t l = a < px ? a : px; dup |= ((l ^ FLIP) != R[t - W + 1] ? 1u : 0u) << (t - W + 1); }
		}
	}

	// ---- 4. emission rules (oracle/pgo_sketch.c: rules A-D) for the thread's eight positions, in position order ----
	auto key_at = [&](int L) -> uint64_t { return s_key[(L & 7) * KS + (L >> 3)]; };
	auto idx_of = [](uint64_t key) -> int { return 4095 - (int)((key >> 1) & 4095u); };
	const int tile_n = (int)min((int64_t)SK_TILE, (int64_t)len - t0);
	uint32_t flags = 0, n_mine = 0;                            // four rule bits per position
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const int c = 8 * tid + j;
		if (c >= tile_n) continue;
		const int L = L0 + j, run = run_at[j];
		const uint64_t kp = R[j], kc = R[j + 1], xp = own[j];
		const bool prev_real = kp != NONE, cur_real = kc != NONE;
		const uint64_t hp = kp >> 13, hx = xp >> 13;            // (NONE >> 13 is above every hash: the comparisons below order like the reference's)
		const bool ruleA = run == W + k - 1 && prev_real;
		bool ruleB = false, ruleC1 = false, ruleC2 = false;
		if (hx <= hp) ruleB = run >= W + k && prev_real;
		else if (idx_of(kp) == L - W) { ruleC1 = run >= W + k - 1 && prev_real; ruleC2 = run >= W + k - 1 && cur_real; }
		const bool ruleD = (t0 + c == (int64_t)len - 1) && cur_real;
		uint32_t n = (ruleB || ruleC1 ? 1u : 0u) + (ruleD ? 1u : 0u);
		if (ruleA && ((dup >> j) & 1u)) { const int pi = idx_of(kp); for (int t = L - W + 1; t < L; ++t) { const uint64_t kt = key_at(t); n += (kt != NONE && (kt >> 13) == hp && t != pi) ? 1u : 0u; } }
		if (ruleC2 && ((dup >> (j + 1)) & 1u)) { const int ci = idx_of(kc); const uint64_t hc = kc >> 13; for (int t = L - W + 1; t <= L; ++t) { const uint64_t kt = key_at(t); n += (kt != NONE && (kt >> 13) == hc && t != ci) ? 1u : 0u; } }
		flags |= ((ruleA ? 1u : 0u) | (ruleB || ruleC1 ? 2u : 0u) | (ruleC2 ? 4u : 0u) | (ruleD ? 8u : 0u)) << (4 * j);
		n_mine += n;
	}
	const uint32_t incl = wave_prefix_sum_incl(n_mine);
	if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
	__syncthreads();
	uint32_t wbase = 0, total = 0;
#pragma unroll
	for (int wv = 0; wv < SK_THREADS / 64; ++wv) { const uint32_t sv = s_wsum[wv]; if (wv < (tid >> 6)) wbase += sv; total += sv; }
	if (tid == 0) { tile_cnt[blockIdx.x] = total; if (total > stage_cap) atomicExch(overflow, 1); }
	if (total == 0 || n_mine == 0) return;
	u128 *out = stage + (size_t)blockIdx.x * stage_cap;
	uint32_t oo = wbase + incl - n_mine;
	const uint64_t ybase = (uint64_t)rid << 32;
	const int64_t pos0 = t0 - 8 * HB;                          // sequence position of local index 0
	auto emit = [&](uint64_t key) {
		if (oo < stage_cap) { u128 r; r.x = (key >> 13) << 8 | (uint64_t)k; r.y = ybase | (uint64_t)(uint32_t)(pos0 + idx_of(key)) << 1 | (key & 1u); out[oo] = r; }
		++oo;
	};
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const uint32_t f = (flags >> (4 * j)) & 15u;
		if (!f) continue;
		const int L = L0 + j;
		const uint64_t kp = R[j], kc = R[j + 1];
		if ((f & 1u) && ((dup >> j) & 1u)) { const int pi = idx_of(kp); const uint64_t hp = kp >> 13; for (int t = L - W + 1; t < L; ++t) { const uint64_t kt = key_at(t); if (kt != NONE && (kt >> 13) == hp && t != pi) emit(kt); } }
		if (f & 2u) emit(kp);
		if ((f & 4u) && ((dup >> (j + 1)) & 1u)) { const int ci = idx_of(kc); const uint64_t hc = kc >> 13; for (int t = L - W + 1; t <= L; ++t) { const uint64_t kt = key_at(t); if (kt != NONE && (kt >> 13) == hc && t != ci) emit(kt); } }
		if (f & 8u) emit(kc);
	}
}

// ---- generic serial kernel: one lane per sequence, the streaming formulation (slots, ring of w entries) ----
__global__ void k_sketch_serial(PkBases bases, const uint64_t *__restrict__ seq_off, const uint32_t *__restrict__ seq_len,
                                int n_seq, int w, int k, u128 *__restrict__ out, const uint64_t *__restrict__ out_off, uint64_t *__restrict__ cnt,
                                u128 *__restrict__ ring_all)
{
	int rid = blockIdx.x * blockDim.x + threadIdx.x;
	if (rid >= n_seq) return;
	const uint64_t s0 = seq_off[rid];                                  // the sequence's first base in the packed store
	const int len = (int)seq_len[rid];
	u128 *ring = ring_all + (size_t)rid * 256;
	u128 *o = out ? out + out_off[rid] : nullptr;
	const uint64_t mask = (1ULL << 2 * k) - 1, shift1 = 2 * (k - 1);
	uint64_t fw = 0, rv = 0, n = 0;
	const u128 none = {SK_NONE, SK_NONE};
	u128 mn = none;
	int run = 0, bp = 0, mp = 0;
	for (int j = 0; j < w; ++j) ring[j] = none;
	auto push = [&](u128 v) { if (o) o[n] = v; ++n; };
	for (int i = 0; i < len; ++i) {
		int c = bases.at(s0 + (uint64_t)i);
		u128 info = none;
		if (c < 4) {
			fw = (fw << 2 | (uint64_t)c) & mask;
			rv = (rv >> 2) | (3ULL ^ (uint64_t)c) << shift1;
			if (fw == rv) continue;                            // not a slot
			int z = fw < rv ? 0 : 1;
			++run;
			if (run >= k) { info.x = hash64(z ? rv : fw, mask) << 8 | (uint64_t)k; info.y = (uint64_t)rid << 32 | (uint64_t)(uint32_t)i << 1 | (uint64_t)z; }
		} else run = 0;
		ring[bp] = info;
		if (run == w + k - 1 && mn.x != SK_NONE) {            // rule A
			for (int j = bp + 1; j < w; ++j) if (mn.x == ring[j].x && ring[j].y != mn.y) push(ring[j]);
			for (int j = 0; j < bp; ++j) if (mn.x == ring[j].x && ring[j].y != mn.y) push(ring[j]);
		}
		if (info.x <= mn.x) {                                  // rule B
			if (run >= w + k && mn.x != SK_NONE) push(mn);
			mn = info, mp = bp;
		} else if (bp == mp) {                                 // rule C
			if (run >= w + k - 1 && mn.x != SK_NONE) push(mn);
			mn.x = SK_NONE;
			for (int j = bp + 1; j < w; ++j) if (mn.x >= ring[j].x) mn = ring[j], mp = j;
			for (int j = 0; j <= bp; ++j) if (mn.x >= ring[j].x) mn = ring[j], mp = j;
			if (run >= w + k - 1 && mn.x != SK_NONE) {
				for (int j = bp + 1; j < w; ++j) if (mn.x == ring[j].x && mn.y != ring[j].y) push(ring[j]);
				for (int j = 0; j <= bp; ++j) if (mn.x == ring[j].x && mn.y != ring[j].y) push(ring[j]);
			}
		}
		if (++bp == w) bp = 0;
	}
	if (mn.x != SK_NONE) push(mn);                              // rule D
	if (cnt) cnt[rid] = n;
}

__global__ void k_compact(const u128 *__restrict__ stage, uint32_t stage_cap, const uint32_t *__restrict__ tile_cnt,
                          const uint64_t *__restrict__ tile_off, u128 *__restrict__ out)
{
	const uint32_t n = tile_cnt[blockIdx.x];
	const u128 *src = stage + (size_t)blockIdx.x * stage_cap;
	u128 *dst = out + tile_off[blockIdx.x];
	for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

__global__ void k_seq_off_from_tiles(const uint64_t *__restrict__ tile_off, const uint32_t *__restrict__ first_tile, int n_seq, uint64_t total, uint64_t *__restrict__ seq_off)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_seq) seq_off[i] = tile_off[first_tile[i]];
	if (i == n_seq) seq_off[n_seq] = total;
}

template <class T> static void exclusive_scan_u64(const T *in, uint64_t *out, size_t n, hipStream_t st)
{
	size_t tmp_bytes = 0;
	auto in_it = rocprim::make_transform_iterator(in, [] __device__ (T v) { return (uint64_t)v; });
	PGA_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, in_it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), st));
	DBuf<uint8_t> tmp(tmp_bytes ? tmp_bytes : 1);
	PGA_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, in_it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), st));
	PGA_HIP(sync_stream(st));
}

void sketch_all(const SeqSet &S, int w, int k, Minimizers &M, hipStream_t st, Timers *tm)
{
	const int n = S.n_seq;
	M.n = 0;
	M.h_seq_off.assign((size_t)n + 1, 0);
	M.seq_off.alloc((size_t)n + 1);
	if (n == 0) return;
	const bool fast = (k & 1) && w + k <= 64 && w + k <= SK_HALO && w <= 63;
	if (fast) {
		std::vector<SkTile> tiles; std::vector<uint32_t> first_tile((size_t)n);
		for (int i = 0; i < n; ++i) {
			first_tile[i] = (uint32_t)tiles.size();
			for (uint32_t s = 0; s < S.len[i]; s += SK_TILE) tiles.push_back(SkTile{(uint32_t)i, s});
		}
		// a sequence with no tiles (len 0) points at the next sequence's first tile; pad with a sentinel tile count
		const size_t nt = tiles.size();
		if (nt == 0) { M.seq_off.zero(st); return; }
		DBuf<SkTile> d_tiles; d_tiles.upload(tiles, st);
		DBuf<uint32_t> d_first; d_first.upload(first_tile, st);
		DBuf<uint32_t> d_cnt(nt + 1); d_cnt.zero(st);
		DBuf<uint64_t> d_toff(nt + 1);
		DBuf<int> d_ovf(1); d_ovf.zero(st);
		// expected density is 2/(w+1) per base: 205 per tile of 2048 with w = 19, 372 with w = 10 (asm20).  The staging slab is nt x cap x 16 bytes -- 2.3 GB per leaf
		// batch (0.57 Gbp) at a quarter of a tile, the largest single block of a batch's arena; 1.7x the expectation instead of 2.5x (an overflow retries at 4x)
		uint32_t cap = w >= 16 ? SK_TILE / 6 : SK_TILE / 3;
		for (;;) {
			DBuf<u128> stage(nt * (size_t)cap);
			EventTimer et(st);
			// eight consecutive positions per thread where w is one of the presets' (asm5/asm10: 19, asm20: 10) and a key fits 64 bits
			static const bool no8 = getenv("PGA_SKETCH_STRIDED") != nullptr;
			const bool fit8 = !no8 && 2 * k + 13 <= 64 && 8 * ((w + 7) / 8) + w + k <= SK8_HALO;
			if (fit8 && w == 19) hipLaunchKernelGGL((k_sketch_tiles8<19>), dim3((unsigned)nt), dim3(SK_THREADS), 0, st, S.d_pk2.p, S.d_nmask.p, S.d_off.p, S.d_len.p, d_tiles.p, k, stage.p, cap, d_cnt.p, d_ovf.p);
			else if (fit8 && w == 10) hipLaunchKernelGGL((k_sketch_tiles8<10>), dim3((unsigned)nt), dim3(SK_THREADS), 0, st, S.d_pk2.p, S.d_nmask.p, S.d_off.p, S.d_len.p, d_tiles.p, k, stage.p, cap, d_cnt.p, d_ovf.p);
			else hipLaunchKernelGGL((k_sketch_tiles<63>), dim3((unsigned)nt), dim3(SK_THREADS), 0, st,
			                   S.d_pk2.p, S.d_nmask.p, S.d_off.p, S.d_len.p, d_tiles.p, w, k, stage.p, cap, d_cnt.p, d_ovf.p);
			PGA_HIP(hipGetLastError());
			et.mark();
			// (the overflow flag and the total travel together behind the scan: one wait instead of two; an overflow -- pathological repeats -- wastes a scan)
			exclusive_scan_u64(d_cnt.p, d_toff.p, nt + 1, st);
			std::vector<int> h_ovf; std::vector<uint64_t> h_tot;
			{ Downloads dl(st); dl.add(h_ovf, d_ovf.p, 1); dl.add(h_tot, d_toff.p + nt, 1); dl.wait(); }
			const double k_ms = et.finish(K_SKETCH);
			if (h_ovf[0]) { cap *= 4; d_ovf.zero(st); d_cnt.zero(st); continue; }  // retry with bigger slabs
			const uint64_t total = h_tot[0];
			M.n = total;
			if (tm) { tm->kern[K_SKETCH].ms += k_ms; tm->kern[K_SKETCH].launches += 1; tm->kern[K_SKETCH].alg_bytes += 0.375 * (double)S.total + 16.0 * (double)total; }   // packed bases in (2 bits + the N bit), minimizers out
			M.mz.alloc(total ? total : 1);
			hipLaunchKernelGGL(k_compact, dim3((unsigned)nt), dim3(256), 0, st, stage.p, cap, d_cnt.p, d_toff.p, M.mz.p);
			// sequences of length 0 have no tile: give them the offset of the next tile (or the total)
			std::vector<uint32_t> ft = first_tile;
			for (int i = 0; i < n; ++i) if (S.len[i] == 0) ft[i] = (uint32_t)nt; // d_toff[nt] == total; fixed below for ordering
			for (int i = n - 2; i >= 0; --i) if (S.len[i] == 0) ft[i] = ft[i + 1];
			d_first.upload(ft, st);
			hipLaunchKernelGGL(k_seq_off_from_tiles, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, d_toff.p, d_first.p, n, total, M.seq_off.p);
			PGA_HIP(hipGetLastError());
			M.h_seq_off = M.seq_off.download(st);                 // (waits for everything above: the staging slabs of this scope are done with)
			return;
		}
	} else {
		DBuf<uint64_t> d_cnt((size_t)n + 1); d_cnt.zero(st);
		DBuf<u128> ring((size_t)n * 256);
		hipLaunchKernelGGL(k_sketch_serial, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, S.bases(), S.d_off.p, S.d_len.p, n, w, k,
		                   (u128*)nullptr, (const uint64_t*)nullptr, d_cnt.p, ring.p);
		exclusive_scan_u64(d_cnt.p, M.seq_off.p, (size_t)n + 1, st);
		uint64_t total = 0;
		PGA_HIP(hipMemcpyAsync(&total, M.seq_off.p + n, 8, hipMemcpyDeviceToHost, st));
		PGA_HIP(sync_stream(st));
		M.n = total;
		M.mz.alloc(total ? total : 1);
		hipLaunchKernelGGL(k_sketch_serial, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, S.bases(), S.d_off.p, S.d_len.p, n, w, k,
		                   M.mz.p, M.seq_off.p, (uint64_t*)nullptr, ring.p);
		PGA_HIP(hipGetLastError());
		PGA_HIP(sync_stream(st));
	}
	M.h_seq_off = M.seq_off.download(st);
}

} // namespace pga
