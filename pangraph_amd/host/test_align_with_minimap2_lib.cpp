// The reference's only known-answer test for this path, restated against the C++ host mirror:
// packages/pangraph/src/align/minimap2_lib/align_with_minimap2_lib.rs:135-204 (test_align_with_minimap2_lib_one_general_case).
// Inputs are read from tests/golden/kat_unit_pair.json (a tiny hand parser: the file holds two sequences and their names).
#include <cstdio>
#include <fstream>
#include <sstream>
#include "minimap2.hpp"

static std::vector<std::string> json_string_array(const std::string &txt, const std::string &key)
{
	std::vector<std::string> out;
	size_t p = txt.find("\"" + key + "\":[");
	if (p == std::string::npos) return out;
	p = txt.find('[', p);
	size_t e = txt.find(']', p);
	while (true) {
		size_t a = txt.find('"', p);
		if (a == std::string::npos || a > e) break;
		size_t b = txt.find('"', a + 1);
		out.push_back(txt.substr(a + 1, b - a - 1));
		p = b + 1;
	}
	return out;
}

int main(int argc, char **argv)
{
	using namespace pangraph;
	std::ifstream f(argc > 1 ? argv[1] : "tests/golden/kat_unit_pair.json");
	std::stringstream ss; ss << f.rdbuf();
	const std::string txt = ss.str();
	std::vector<std::string> names = json_string_array(txt, "names"), seqs = json_string_array(txt, "seqs");
	if (names.size() != 2 || seqs.size() != 2) { fprintf(stderr, "cannot read the fixture\n"); return 2; }
	AlignmentArgs params; params.kmer_length = 10; params.sensitivity = 20;
	std::vector<Alignment> actual;
	try { actual = align_with_minimap2_lib(seqs, names, params); }
	catch (std::exception &e) { fprintf(stderr, "error: %s\n", e.what()); return 3; }
	std::vector<Alignment> expected = { Alignment{ Hit{"0", 998, 0, 996}, Hit{"1", 1000, 0, 998}, 969, 998, 0, '+', "545M1D225M1D226M",
	                                               0.029058116232464903, 845.0 } };
	if (!(actual == expected)) {
		fprintf(stderr, "MISMATCH: got %zu alignments\n", actual.size());
		for (auto &a : actual) fprintf(stderr, "  %s:(%zu,%zu)/%zu %s:(%zu,%zu)/%zu m=%zu l=%zu q=%zu %c %s de=%.17g AS=%g\n", a.qry.name.c_str(), a.qry.start, a.qry.end, a.qry.length,
		                               a.reff.name.c_str(), a.reff.start, a.reff.end, a.reff.length, a.matches, a.length, a.quality, a.orientation, a.cigar.c_str(), a.divergence.value_or(-1), a.align.value_or(-1));
		return 1;
	}
	// error behaviour of the adapter (align_with_minimap2_lib.rs:35-40)
	bool threw = false;
	try { AlignmentArgs bad; bad.sensitivity = 7; align_with_minimap2_lib(seqs, names, bad); } catch (std::runtime_error &) { threw = true; }
	if (!threw) { fprintf(stderr, "unknown sensitivity did not raise\n"); return 1; }
	printf("test_align_with_minimap2_lib_one_general_case ... ok\n");
	return 0;
}
