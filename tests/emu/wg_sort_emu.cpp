// pga_wg_sort.h under dev/emu/hip_emu.h against std::stable_sort on the compared key bits (what rocprim::radix_sort_pairs(..., 0, end_bit) gives).
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
#include "../../dev/emu/hip_emu.h"
#include "../../pangraph_amd/csrc/pga_wg_sort.h"
using namespace pga;

template <typename K> static bool run(std::mt19937_64 &rng, uint32_t n, int end_bit, uint64_t distinct)
{
	std::vector<K> kin(n), kout(n, (K)~(K)0);
	std::vector<uint32_t> vin(n), vout(n, ~0u);
	for (uint32_t i = 0; i < n; ++i) { kin[i] = (K)(distinct ? rng() % distinct * 0x9E3779B97F4A7C15ULL : rng()); vin[i] = (uint32_t)rng(); }
	emu_launch(dim3(1), dim3(WGS_NT), [&] { k_wg_sort_pairs<K>(kin.data(), kout.data(), vin.data(), vout.data(), n, end_bit); });
	const uint64_t mask = end_bit >= 64 ? ~0ULL : ((1ULL << end_bit) - 1);
	std::vector<uint32_t> ord(n);
	for (uint32_t i = 0; i < n; ++i) ord[i] = i;
	std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return ((uint64_t)kin[a] & mask) < ((uint64_t)kin[b] & mask); });
	for (uint32_t i = 0; i < n; ++i) if (kout[i] != kin[ord[i]] || vout[i] != vin[ord[i]]) { printf("FAIL: n %u, end_bit %d, %zu-byte keys: position %u\n", n, end_bit, sizeof(K), i); return false; }
	return true;
}

int main()
{
	std::mt19937_64 rng(7);
	int n_ok = 0;
	for (uint32_t n : {1u, 2u, 65u, 1000u, 1025u, 4095u, 4096u})
		for (int end_bit : {1, 32, 39, 64})
			for (uint64_t distinct : {0ull, 5ull}) {
				if (!run<uint64_t>(rng, n, end_bit, distinct)) return 1;
				if (end_bit <= 32 && !run<uint32_t>(rng, n, end_bit, distinct)) return 1;
				++n_ok;
			}
	// outside its range the kernel writes nothing (the caller must not have launched it)
	std::vector<uint64_t> kin(WGS_CAP + 1, 5), kout(WGS_CAP + 1, 99); std::vector<uint32_t> vin(WGS_CAP + 1, 1), vout(WGS_CAP + 1, 99);
	emu_launch(dim3(1), dim3(WGS_NT), [&] { k_wg_sort_pairs<uint64_t>(kin.data(), kout.data(), vin.data(), vout.data(), WGS_CAP + 1, 64); });
	if (kout[0] != 99 || vout[WGS_CAP] != 99) { printf("FAIL: wrote beyond its range\n"); return 1; }
	printf("ok: %d configurations\n", n_ok);
	return 0;
}
