// pga_mem.cpp -- caching device allocator behind DBuf.
//
// A batch allocates a few dozen device arrays per stage and the DP stage a scratch slab of tens of GB; hipMalloc and
// hipFree cost from 0.1 ms to tens of ms each and hipFree synchronises the device, so freed blocks are kept in
// per-device, size-rounded free lists and handed out again (a level of `pangraph build` repeats the same sizes call
// after call).  The cache is bounded: beyond PGA_CACHE_GB (default 96) of idle blocks the largest are released.
#include "pga_common.h"
#include <map>
#include <mutex>

namespace pga {

namespace {
struct Pool {
	std::multimap<size_t, void*> idle;              // rounded size -> block
	std::map<void*, size_t> live;                   // block -> rounded size
	size_t idle_bytes = 0;
};
std::mutex g_mu;
std::map<int, Pool> g_pools;

size_t round_size(size_t b)
{
	if (b < 256) b = 256;
	size_t p = 256;
	while (p < b) p <<= 1;                           // next power of two ...
	if (b <= (1u << 20)) return p;
	const size_t step = p >> 4;                      // ... refined to 1/16 steps above 1 MB
	return (b + step - 1) / step * step;
}
size_t cache_limit()
{
	static size_t lim = [] { const char *e = getenv("PGA_CACHE_GB"); double g = e ? atof(e) : 96.0; return (size_t)(g * (double)(1ull << 30)); }();
	return lim;
}
}

void *dev_alloc(size_t bytes)
{
	int dev = 0;
	PGA_HIP(hipGetDevice(&dev));
	const size_t r = round_size(bytes);
	{
		std::lock_guard<std::mutex> lk(g_mu);
		Pool &P = g_pools[dev];
		auto it = P.idle.lower_bound(r);
		if (it != P.idle.end() && it->first <= r + r / 4) {
			void *p = it->second; const size_t sz = it->first;
			P.idle.erase(it); P.idle_bytes -= sz; P.live[p] = sz;
			return p;
		}
	}
	void *p = nullptr;
	hipError_t e = hipMalloc(&p, r);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		dev_trim();                                   // give the idle blocks back and retry once
		e = hipMalloc(&p, r);
		if (e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " allocating " + std::to_string(r) + " bytes of device memory");
	}
	std::lock_guard<std::mutex> lk(g_mu);
	g_pools[dev].live[p] = r;
	return p;
}

void dev_free(void *p)
{
	if (!p) return;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return;
	std::vector<void*> drop;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		Pool &P = g_pools[dev];
		auto it = P.live.find(p);
		if (it == P.live.end()) { drop.push_back(p); }
		else {
			const size_t sz = it->second;
			P.live.erase(it);
			P.idle.emplace(sz, p); P.idle_bytes += sz;
			while (P.idle_bytes > cache_limit() && !P.idle.empty()) {
				auto big = std::prev(P.idle.end());
				drop.push_back(big->second); P.idle_bytes -= big->first; P.idle.erase(big);
			}
		}
	}
	for (void *q : drop) (void)hipFree(q);
}

void dev_trim()
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return;
	std::vector<void*> drop;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		Pool &P = g_pools[dev];
		for (auto &kv : P.idle) drop.push_back(kv.second);
		P.idle.clear(); P.idle_bytes = 0;
	}
	for (void *q : drop) (void)hipFree(q);
}

} // namespace pga
