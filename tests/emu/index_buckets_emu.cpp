// The sort-free index build (pangraph_amd/csrc/pga_index_buckets.h) with every workgroup on host threads (dev/emu/hip_emu.h): the kernels are the
// product's, the flow below is build_index_buckets() of pga_index.hip with plain arrays.  Checked against a std::map: every minimizer must find,
// through its key id, exactly the occurrence words of the minimizers of ITS group with ITS hash, ascending (index.c:84-98,252).
//   index_buckets_emu <seed> <n_groups> <max minimizers per sequence> <distinct hashes per group> <heavy>   (heavy: one k-mer repeated that many times in group 1)
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>
#include "../../dev/emu/hip_emu.h"
struct u128 { uint64_t x, y; };
#include "../../pangraph_amd/csrc/pga_index_buckets.h"
#include "../../pangraph_amd/csrc/pga_maxocc_hist.h"
#include <algorithm>
using namespace pga;

int main(int argc, char **argv)
{
	const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
	const int n_grp = argc > 2 ? atoi(argv[2]) : 7;
	const int max_per_seq = argc > 3 ? atoi(argv[3]) : 3000;
	const int distinct = argc > 4 ? atoi(argv[4]) : 2000;
	const int heavy = argc > 5 ? atoi(argv[5]) : 0;
	const int k = 19, hash_bits = 2 * k;
	std::mt19937_64 rng(seed);
	// sequences group by group; minimizers ordered by (rid, pos)
	std::vector<uint32_t> grp_of_seq;
	std::vector<u128> mz;
	std::vector<uint64_t> mz_begin{0};
	for (int g = 0; g < n_grp; ++g) {
		const int n_seq = (g % 5 == 3) ? 0 : 1 + (int)(rng() % 3);                 // some groups hold nothing
		std::vector<uint64_t> pool((size_t)std::max(1, (int)(rng() % (uint64_t)distinct) + 1));
		for (uint64_t &h : pool) h = rng() & ((1ULL << hash_bits) - 1);
		for (int s = 0; s < n_seq; ++s) {
			const uint32_t rid = (uint32_t)grp_of_seq.size();
			grp_of_seq.push_back((uint32_t)g);
			const int n_mz = (g % 5 == 4) ? (int)(rng() % 3) : (int)(rng() % (uint64_t)max_per_seq);
			uint32_t pos = 0;
			for (int i = 0; i < n_mz; ++i) {
				pos += 1 + (uint32_t)(rng() % 20);
				uint64_t h = pool[rng() % pool.size()];
				if (heavy && g == 1 && i < heavy) h = pool[0];
				mz.push_back(u128{h << 8 | (uint64_t)k, (uint64_t)rid << 32 | (uint64_t)pos << 1 | (rng() & 1)});
			}
		}
		mz_begin.push_back(mz.size());
	}
	if (grp_of_seq.empty()) { grp_of_seq.push_back(0); }
	const uint32_t n = (uint32_t)mz.size();
	if (n == 0) { printf("ok: empty batch\n"); return 0; }

	// ---- the flow of build_index_buckets (pga_index.hip) ----
	std::vector<IxbGroup> gt((size_t)n_grp + 1);
	const uint32_t nb = ixb_make_table(n_grp, mz_begin.data(), hash_bits, gt.data());
	std::vector<uint32_t> cnt(nb, 0), cursor(nb, 0), flags(4, 0), off((size_t)nb + 1, 0xdeadbeef), nk(nb, 0xdeadbeef), kbase((size_t)nb + 1, 0xdeadbeef), flags2(4, 0);
	std::vector<uint64_t> sck(n, ~0ULL), sy(n, ~0ULL), ck2(n, ~0ULL), occ(n, ~0ULL);
	std::vector<uint32_t> so(n, ~0u), orig2(n, ~0u), grp_of_mz(n, ~0u);
	const unsigned tiles = (n + IXB_TILE - 1) / IXB_TILE;
	emu_launch(dim3(tiles), dim3(IXB_NT), [&] { k_ixb_count(mz.data(), n, grp_of_seq.data(), gt.data(), hash_bits, cnt.data()); });
	emu_launch(dim3(1), dim3(1024), [&] { k_ixb_scan(cnt.data(), nb, off.data(), IXB_CAP, flags.data()); });
	if (off[nb] != n || flags[1] != n) { printf("FAIL: the bucket counts add up to %u, not %u\n", off[nb], n); return 1; }
	const bool overflow = flags[0] > IXB_CAP;
	emu_launch(dim3(tiles), dim3(IXB_NT), [&] { k_ixb_scatter(mz.data(), n, grp_of_seq.data(), gt.data(), hash_bits, off.data(), cursor.data(), sck.data(), sy.data(), so.data()); });
	for (uint32_t b = 0; b < nb; ++b) if (cursor[b] != off[b + 1] - off[b]) { printf("FAIL: bucket %u received %u of %u records\n", b, cursor[b], off[b + 1] - off[b]); return 1; }
	emu_launch(dim3(nb), dim3(IXB_NT), [&] { k_ixb_sort(off.data(), sck.data(), sy.data(), so.data(), ck2.data(), orig2.data(), occ.data(), nk.data()); });
	emu_launch(dim3(1), dim3(1024), [&] { k_ixb_scan(nk.data(), nb, kbase.data(), 0xffffffffu, flags2.data()); });
	const uint32_t n_keys = flags2[1];
	if (overflow) {
		uint32_t mx = 0; for (uint32_t b = 0; b < nb; ++b) mx = std::max(mx, off[b + 1] - off[b]);
		if (mx != flags[0]) { printf("FAIL: overflow flag %u, largest bucket %u\n", flags[0], mx); return 1; }
		printf("ok: overflow reported (a bucket of %u records > %u): the caller takes the sort route\n", flags[0], IXB_CAP);
		return heavy > (int)IXB_CAP ? 0 : 1;
	}
	std::vector<uint64_t> key(n_keys, ~0ULL);
	std::vector<uint32_t> occ_off((size_t)n_keys + 1, ~0u), key_grp(n_keys, ~0u);
	emu_launch(dim3(nb), dim3(IXB_NT), [&] { k_ixb_groups(off.data(), kbase.data(), ck2.data(), orig2.data(), hash_bits, n, n_keys, key.data(), occ_off.data(), key_grp.data(), grp_of_mz.data()); });

	// ---- reference: (group, hash) -> y ascending ----
	std::map<std::pair<uint32_t, uint64_t>, std::vector<uint64_t>> ref;
	for (uint32_t i = 0; i < n; ++i) ref[{grp_of_seq[mz[i].y >> 32], mz[i].x >> 8}].push_back(mz[i].y);      // arrival order is y ascending
	if (ref.size() != n_keys) { printf("FAIL: %u keys, reference %zu\n", n_keys, ref.size()); return 1; }
	if (occ_off[n_keys] != n) { printf("FAIL: occ_off[n_keys] = %u\n", occ_off[n_keys]); return 1; }
	std::vector<char> key_seen(n_keys, 0);
	for (uint32_t i = 0; i < n; ++i) {
		const uint32_t g = grp_of_mz[i];
		if (g >= n_keys) { printf("FAIL: minimizer %u has key id %u\n", i, g); return 1; }
		const uint32_t grp = grp_of_seq[mz[i].y >> 32];
		if (key[g] != mz[i].x >> 8 || key_grp[g] != grp) { printf("FAIL: minimizer %u: key id %u holds another (group, hash)\n", i, g); return 1; }
		if (key_seen[g]) continue;
		key_seen[g] = 1;
		const std::vector<uint64_t> &want = ref[{grp, mz[i].x >> 8}];
		const uint32_t o0 = occ_off[g], o1 = occ_off[g + 1];
		if (o1 < o0 || o1 - o0 != want.size()) { printf("FAIL: list of key %u has %u entries, reference %zu\n", g, o1 - o0, want.size()); return 1; }
		for (uint32_t j = 0; j < o1 - o0; ++j) if (occ[o0 + j] != want[j]) { printf("FAIL: list of key %u differs at %u\n", g, j); return 1; }
	}
	for (uint32_t g = 0; g < n_keys; ++g) if (!key_seen[g]) { printf("FAIL: key %u belongs to no minimizer\n", g); return 1; }
	// the lists of a group lie together: key ids (and list offsets) of a group form one range
	for (uint32_t g = 1; g < n_keys; ++g) if (key_grp[g] < key_grp[g - 1]) { printf("FAIL: groups are not consecutive in key order at %u\n", g); return 1; }
	// ---- mid_occ of every group from histograms (pga_maxocc_hist.h) against the sorted counts (index.c:186-207) ----
	int n_unresolved = 0;
	for (float f : {2e-4f, 0.05f, 0.5f}) {
		std::vector<uint32_t> hist((size_t)n_grp * MO_BINS, 0);
		std::vector<int32_t> got((size_t)n_grp, -7);
		emu_launch(dim3((n_keys + MO_KEYS - 1) / MO_KEYS), dim3(MO_NT), [&] { k_mo_hist(occ_off.data(), key_grp.data(), n_keys, hist.data()); });
		emu_launch(dim3((unsigned)n_grp), dim3(MO_NT), [&] { k_mo_select(hist.data(), n_grp, f, got.data()); });
		std::vector<std::vector<uint32_t>> counts((size_t)n_grp);
		for (uint32_t g = 0; g < n_keys; ++g) counts[key_grp[g]].push_back(occ_off[g + 1] - occ_off[g]);
		for (int g = 0; g < n_grp; ++g) {
			std::vector<uint32_t> &c = counts[(size_t)g];
			int32_t want = 1;
			if (!c.empty()) { std::sort(c.begin(), c.end()); want = (int32_t)(c[(uint32_t)((1. - (double)f) * (double)c.size())] + 1u); }
			if (got[(size_t)g] == -1) { if (want - 1 < (int32_t)MO_BINS - 1) { printf("FAIL: group %d unresolved although its answer is %d\n", g, want); return 1; } ++n_unresolved; continue; }
			if (got[(size_t)g] != want) { printf("FAIL: max_occ of group %d at f = %g: %d, reference %d\n", g, f, got[(size_t)g], want); return 1; }
		}
	}
	uint32_t biggest = 0; for (uint32_t b = 0; b < nb; ++b) biggest = std::max(biggest, off[b + 1] - off[b]);

	printf("ok: %u minimizers, %d groups, %u buckets (largest %u), %u keys, %u tiles\n", n, n_grp, nb, biggest, n_keys, tiles);
	if (n_unresolved) printf("(max_occ: %d group answers beyond the histogram: the sort route)\n", n_unresolved);
	return 0;
}
