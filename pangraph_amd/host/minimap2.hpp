// minimap2.hpp -- C++ host-side mirror of the reference's Rust layers above the C-ABI (the Rust toolchain is absent
// in this image, so the mirror is C++ as the reference's host is compiled code):
//   Minimap2Preset / Minimap2Args / Minimap2Options   packages/minimap2/src/options.rs:11-138, options_args.rs:15-551
//   Minimap2Index                                      packages/minimap2/src/index.rs:10-77
//   Minimap2Mapper / Minimap2Result / Minimap2PafRow   packages/minimap2/src/map.rs:24-70,263-421
//   AlignmentArgs / Alignment / Hit                    packages/pangraph/src/align/{alignment_args,alignment}.rs
//   align_with_minimap2_lib                            packages/pangraph/src/align/minimap2_lib/align_with_minimap2_lib.rs:15-122
// Same names, argument meaning and error behaviour (errors are exceptions carrying the Rust messages).
#pragma once
#include <cstdlib>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/pga_mm2_abi.h"

namespace pangraph {

enum class Minimap2Preset { Asm5, Asm10, Asm20 };
inline const char *as_str(Minimap2Preset p) { return p == Minimap2Preset::Asm5 ? "asm5" : p == Minimap2Preset::Asm10 ? "asm10" : "asm20"; }

struct Minimap2Args {                       // options_args.rs:15-270 (the fields pangraph sets)
	std::optional<Minimap2Preset> x;
	std::optional<int> k, w, s;
	bool c = false, X = false;
	std::optional<int> bucket_bits;
};

struct Minimap2Options {                    // options.rs:55-138
	mm_idxopt_t idx_opt; mm_mapopt_t map_opt;
	explicit Minimap2Options(const Minimap2Args &a) {
		if (mm_set_opt(nullptr, &idx_opt, &map_opt) != 0) throw std::runtime_error("minimap2: mm_set_opt(null, ...): failed to set options: incorrect preset");
		if (a.x && mm_set_opt(as_str(*a.x), &idx_opt, &map_opt) != 0) throw std::runtime_error("minimap2: mm_set_opt(preset, ...): failed to set options: incorrect preset");
		if (a.k) idx_opt.k = (short)*a.k;                     // options_args.rs:274-279
		if (a.w) idx_opt.w = (short)*a.w;
		if (a.c) map_opt.flag |= MM_F_OUT_CG | MM_F_CIGAR;    // :305-307
		if (a.s) map_opt.min_dp_max = *a.s;                   // :314-316
		if (a.X) map_opt.flag |= MM_F_ALL_CHAINS | MM_F_NO_DIAG | MM_F_NO_DUAL | MM_F_NO_LJOIN; // :323-325
		if (a.bucket_bits) idx_opt.bucket_bits = (short)*a.bucket_bits;
		if (mm_check_opt(&idx_opt, &map_opt) != 0) throw std::runtime_error("minimap2: mm_check_opt(): options are invalid");
	}
};

class Minimap2Index {                       // index.rs:10-77
	mm_idx_t *idx_ = nullptr;
public:
	Minimap2Options options;
	Minimap2Index(const std::vector<std::string> &seqs, const std::vector<std::string> &names, const Minimap2Args &args) : options(args) {
		std::vector<const char*> s, n;
		for (auto &x : seqs) s.push_back(x.c_str());
		for (auto &x : names) n.push_back(x.c_str());
		idx_ = mm_idx_str(options.idx_opt.w, options.idx_opt.k, options.idx_opt.flag & MM_I_HPC, options.idx_opt.bucket_bits, (int)seqs.size(), s.data(), n.data());
		if (!idx_) throw std::runtime_error("minimap2: failed to create index");
		mm_mapopt_update(&options.map_opt, idx_);
	}
	Minimap2Index(const Minimap2Index&) = delete;
	~Minimap2Index() { mm_idx_destroy(idx_); }
	const mm_idx_t *get() const { return idx_; }
};

struct Minimap2PafRowSeq { std::string name; size_t len; int32_t start, end; };
struct Minimap2PafRow {                     // map.rs:215-261 (fields pangraph reads)
	Minimap2PafRowSeq q, t; char strand; int32_t mlen, blen; uint8_t mapq;
	std::optional<int32_t> AS; std::optional<std::string> cg; std::optional<double> de;
};

class Minimap2Mapper {                      // map.rs:24-43, buf.rs
	mm_tbuf_t *buf_; const Minimap2Index &idx_;
public:
	explicit Minimap2Mapper(const Minimap2Index &idx) : buf_(mm_tbuf_init()), idx_(idx) {}
	~Minimap2Mapper() { mm_tbuf_destroy(buf_); }
	std::vector<Minimap2PafRow> run_map(const std::string &seq, const std::string &name) { // Minimap2Result::new, map.rs:45-70
		int n_regs = 0;
		mm_reg1_t *regs = mm_map(idx_.get(), (int)seq.size(), seq.c_str(), &n_regs, buf_, &idx_.options.map_opt, name.c_str());
		std::vector<Minimap2PafRow> pafs;
		const mm_idx_t *mi = idx_.get();
		for (int i = 0; i < n_regs; ++i) {                     // Minimap2PafRow::from_raw, map.rs:264-355
			const mm_reg1_t &r = regs[i];
			Minimap2PafRow p;
			p.q = {name, seq.size(), r.qs, r.qe};
			p.t = {mi->seq[r.rid].name, mi->seq[r.rid].len, r.rs, r.re};
			p.strand = r.rev ? '-' : '+'; p.mlen = r.mlen, p.blen = r.blen, p.mapq = (uint8_t)r.mapq;
			if (r.p) {
				p.AS = r.p->dp_score;
				std::string cg;
				for (uint32_t j = 0; j < r.p->n_cigar; ++j) { cg += std::to_string(r.p->cigar[j] >> 4); cg += MM_CIGAR_STR[r.p->cigar[j] & 0xf]; }
				p.cg = cg;
				p.de = 1.0 - mm_event_identity(&r);
				free(r.p);                                     // Drop, map.rs:407-420
			}
			pafs.push_back(p);
		}
		free(regs);
		return pafs;
	}
};

struct AlignmentArgs {                      // alignment_args.rs:5-35
	size_t indel_len_threshold = 100; double alpha = 100.0, beta = 10.0; size_t sensitivity = 10; std::optional<size_t> kmer_length;
};
struct Hit { std::string name; size_t length; size_t start, end; bool operator==(const Hit &o) const { return name == o.name && length == o.length && start == o.start && end == o.end; } };
struct Alignment {                          // alignment.rs:40-57
	Hit qry, reff; size_t matches, length, quality; char orientation; std::string cigar; std::optional<double> divergence, align;
	bool operator==(const Alignment &o) const {
		return qry == o.qry && reff == o.reff && matches == o.matches && length == o.length && quality == o.quality && orientation == o.orientation &&
		       cigar == o.cigar && divergence == o.divergence && align == o.align;
	}
};

inline std::vector<Alignment> align_with_minimap2_lib(const std::vector<std::string> &seqs, const std::vector<std::string> &names, const AlignmentArgs &params)
{   // align_with_minimap2_lib.rs:29-122
	if (names.size() != seqs.size())
		throw std::logic_error("Number of sequences and number of sequence names is expected to be the same, but found: " + std::to_string(seqs.size()) + " sequences and " + std::to_string(names.size()) + " names");
	Minimap2Args args;
	switch (params.sensitivity) {
		case 5: args.x = Minimap2Preset::Asm5; break;
		case 10: args.x = Minimap2Preset::Asm10; break;
		case 20: args.x = Minimap2Preset::Asm20; break;
		default: throw std::runtime_error("Unknown sensitivity preset: " + std::to_string(params.sensitivity));
	}
	if (params.kmer_length) args.k = (int)*params.kmer_length;
	args.c = true, args.X = true;
	long s = (long)params.indel_len_threshold - 10; args.s = (int)(s < 5 ? 5 : s);
	args.bucket_bits = 14;
	Minimap2Index idx(seqs, names, args);
	Minimap2Mapper mapper(idx);
	std::vector<Alignment> alns;
	for (size_t i = 0; i < seqs.size(); ++i)
		for (auto &paf : mapper.run_map(seqs[i], names[i])) {
			if (!paf.cg) throw std::logic_error("Unable to find CIGAR string in the result");
			alns.push_back(Alignment{Hit{paf.q.name, paf.q.len, (size_t)paf.q.start, (size_t)paf.q.end}, Hit{paf.t.name, paf.t.len, (size_t)paf.t.start, (size_t)paf.t.end},
			                         (size_t)paf.mlen, (size_t)paf.blen, (size_t)paf.mapq, paf.strand, *paf.cg, paf.de, paf.AS ? std::optional<double>((double)*paf.AS) : std::nullopt});
		}
	return alns;
}

} // namespace pangraph
