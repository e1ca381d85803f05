// pga_maxocc_hist.h (mm_idx_cal_max_occ of every group from per-group histograms of the occurrence counts) with every workgroup on host threads
// (dev/emu/hip_emu.h): the product's kernels on a random key table (keys group by group, as the index build leaves them), against the sorted counts
// (packages/minimap2-sys/minimap2/index.c:186-207), at three fractions.
//   maxocc_hist_emu <seed> <n_groups> <max keys per group> <heavy>   (heavy: one key of group 1 with that many occurrences)
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <algorithm>
#include "../../dev/emu/hip_emu.h"
#include "../../pangraph_amd/csrc/pga_maxocc_hist.h"
using namespace pga;

int main(int argc, char **argv)
{
	const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
	const int n_grp = argc > 2 ? atoi(argv[2]) : 7;
	const int max_keys = argc > 3 ? atoi(argv[3]) : 3000;
	const int heavy = argc > 4 ? atoi(argv[4]) : 0;
	std::mt19937_64 rng(seed);
	std::vector<uint32_t> occ_off{0}, key_grp;
	for (int g = 0; g < n_grp; ++g) {
		const int nk = (g % 5 == 3) ? 0 : (g % 5 == 4) ? (int)(rng() % 3) : 1 + (int)(rng() % (uint64_t)max_keys);    // empty and tiny groups among them
		for (int i = 0; i < nk; ++i) {
			uint32_t c = 1 + (uint32_t)(rng() % 3 == 0 ? rng() % 40 : rng() % 2);      // mostly 1 or 2, some dozens
			if (rng() % 500 == 0) c = 200 + (uint32_t)(rng() % 700);                    // repeats: hundreds
			if (heavy && g == 1 && i == 0) c = (uint32_t)heavy;
			key_grp.push_back((uint32_t)g); occ_off.push_back(occ_off.back() + c);
		}
	}
	const uint32_t n_keys = (uint32_t)key_grp.size();
	if (n_keys == 0) { printf("ok: no keys\n"); return 0; }
	int n_unresolved = 0;
	for (float f : {2e-4f, 0.05f, 0.5f}) {
		std::vector<uint32_t> hist((size_t)n_grp * MO_BINS, 0);
		std::vector<int32_t> got((size_t)n_grp, -7);
		emu_launch(dim3((n_keys + MO_KEYS - 1) / MO_KEYS), dim3(MO_NT), [&] { k_mo_hist(occ_off.data(), key_grp.data(), n_keys, hist.data()); });
		emu_launch(dim3((unsigned)n_grp), dim3(MO_NT), [&] { k_mo_select(hist.data(), n_grp, f, got.data()); });
		std::vector<std::vector<uint32_t>> counts((size_t)n_grp);
		for (uint32_t i = 0; i < n_keys; ++i) counts[key_grp[i]].push_back(occ_off[i + 1] - occ_off[i]);
		for (int g = 0; g < n_grp; ++g) {
			std::vector<uint32_t> &c = counts[(size_t)g];
			int32_t want = 1;
			if (!c.empty()) { std::sort(c.begin(), c.end()); want = (int32_t)(c[(uint32_t)((1. - (double)f) * (double)c.size())] + 1u); }
			if (got[(size_t)g] == -1) { if (want - 1 < (int32_t)MO_BINS - 1) { printf("FAIL: group %d unresolved although its answer is %d\n", g, want); return 1; } ++n_unresolved; continue; }
			if (got[(size_t)g] != want) { printf("FAIL: max_occ of group %d at f = %g: %d, reference %d\n", g, f, got[(size_t)g], want); return 1; }
		}
	}
	printf("ok: %u keys, %d groups\n", n_keys, n_grp);
	if (n_unresolved) printf("(max_occ: %d group answers beyond the histogram: the sort route)\n", n_unresolved);
	return 0;
}
