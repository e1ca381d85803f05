// pga_align.cpp -- region bookkeeping and the base-level alignment DRIVER of a batch (host side).
//
// Replaces mm_gen_regs / mm_split_reg / mm_filter_regs / mm_hit_sort / mm_squeeze_a / mm_set_mapq
// (reference: packages/minimap2-sys/minimap2/hit.c) and mm_align_skeleton / mm_align1 / mm_align1_inv with their
// helpers (align.c:9-45,47-167,240-314,355-498,575-1022).  The reference aligns one region after another and
// calls the DP kernel synchronously.  Here every region of every query of the batch is a small state machine:
//   plan    (once)  end fixing, bad-seed flags, DP windows, the list of gap-fill segments  -> DP problems
//   advance (rounds) consume DP results in the reference's order: left extension, gap fills with the z-drop
//                    test (a second exact pass is requested when it fires), right extension, split, inversion
// and all DP problems requested in a round run as ONE kernel launch (pga_ksw.hip).  Typical batches need 2-3
// rounds.  CIGAR post-processing (mm_fix_cigar / mm_update_extra) stays on the host, threaded over queries.
#include "pga_common.h"
#include "pga_dp.h"
#include "pga_sort_exact.h"
#include "pga_pipeline.h"
#include <cmath>
#include <thread>
#include <atomic>
#include <cassert>
#include <list>
#include <deque>
#include <chrono>
#include <cstdio>
#include <mutex>

namespace pga {

// host-side phase accounting (PGA_VERBOSE): nanoseconds summed over threads
static std::atomic<long long> g_ns[8];
static bool g_prof = false;
struct ScopeNs { int k; std::chrono::steady_clock::time_point t0; explicit ScopeNs(int k_) : k(k_) { if (g_prof) t0 = std::chrono::steady_clock::now(); }
	~ScopeNs() { if (g_prof) g_ns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } };



#define SEED_LONG_JOIN (1ULL<<40)
#define SEED_IGNORE    (1ULL<<41)
#define SEED_TANDEM    (1ULL<<42)
#define SEED_SELF      (1ULL<<43)
#define NEG_INF (-0x40000000)
#define EZ_RIGHT      0x02
#define EZ_APPROX_MAX 0x08
#define EZ_EXTZ_ONLY  0x40
#define EZ_REV_CIGAR  0x80

static inline float mg_log2_host(float x) // mmpriv.h:118-126
{
	union { float f; uint32_t i; } z = { x };
	float log_2 = (float)(((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	log_2 += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return log_2;
}

static void sort128(std::vector<u128> &v) { uint32_t head[256], tail[256]; if (!v.empty()) radix_sort_128x_exact(v.data(), v.data() + v.size(), head, tail); }

// ---------------- hit.c ----------------
static void fuzzy_len(Reg &r, const u128 *a) // hit.c:8-21
{
	r.mlen = r.blen = 0;
	if (r.cnt <= 0) return;
	r.mlen = r.blen = (int32_t)(a[r.as].y >> 32 & 0xff);
	for (int i = r.as + 1; i < r.as + r.cnt; ++i) {
		int span = (int)(a[i].y >> 32 & 0xff);
		int tl = (int32_t)a[i].x - (int32_t)a[i-1].x, ql = (int32_t)a[i].y - (int32_t)a[i-1].y;
		r.blen += tl > ql ? tl : ql;
		r.mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
	}
}
static void set_coor(Reg &r, int32_t qlen, const u128 *a) // hit.c:23-38
{
	int32_t k = r.as, q_span = (int32_t)(a[k].y >> 32 & 0xff);
	r.rev = (uint32_t)(a[k].x >> 63);
	r.rid = (int32_t)(a[k].x << 1 >> 33);
	r.rs = (int32_t)a[k].x + 1 > q_span ? (int32_t)a[k].x + 1 - q_span : 0;
	r.re = (int32_t)a[k + r.cnt - 1].x + 1;
	if (!r.rev) { r.qs = (int32_t)a[k].y + 1 - q_span; r.qe = (int32_t)a[k + r.cnt - 1].y + 1; }
	else { r.qs = qlen - ((int32_t)a[k + r.cnt - 1].y + 1); r.qe = qlen - ((int32_t)a[k].y + 1 - q_span); }
	fuzzy_len(r, a);
}
static inline uint64_t mix64(uint64_t key) // hit.c:40-50
{
	key = (~key + (key << 21)); key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)); key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)); key = key ^ key >> 28;
	key = (key + (key << 31));
	return key;
}
static inline uint32_t x31_hash(const char *s) { uint32_t h = (uint32_t)*s; if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)*s; return h; }
static inline uint32_t wang_hash(uint32_t key) { key += ~(key << 15); key ^= (key >> 10); key += (key << 3); key ^= (key >> 6); key += ~(key << 11); key ^= (key >> 16); return key; }

static void gen_regs(uint32_t hash, int qlen, int n_u, const uint64_t *u, const u128 *a, std::vector<Reg> &regs) // hit.c:52-88
{
	regs.clear();
	if (n_u == 0) return;
	std::vector<u128> z((size_t)n_u);
	int i, k;
	for (i = k = 0; i < n_u; ++i) {
		uint32_t h = (uint32_t)mix64((mix64(a[k].x) + mix64(a[k].y)) ^ hash);
		z[i].x = u[i] ^ h;
		z[i].y = (uint64_t)k << 32 | (uint64_t)(int32_t)u[i];
		k += (int32_t)u[i];
	}
	sort128(z);
	std::reverse(z.begin(), z.end());
	regs.resize((size_t)n_u);
	for (i = 0; i < n_u; ++i) {
		Reg &r = regs[i];
		r = Reg();
		r.id = i, r.parent = -1;
		r.score = r.score0 = (int32_t)(z[i].x >> 32);
		r.hash = (uint32_t)z[i].x;
		r.cnt = (int32_t)z[i].y, r.as = (int32_t)(z[i].y >> 32);
		set_coor(r, qlen, a);
	}
}

static void split_reg(Reg &r, Reg &r2, int n, int qlen, const u128 *a) // hit.c:106-123
{
	if (n <= 0 || n >= r.cnt) return;
	r2 = r;
	r2.id = -1; r2.has_p = false; r2.cigar.clear(); r2.dp_score = r2.dp_max = r2.dp_max2 = 0; r2.n_ambi = 0;
	r2.split_inv = 0;
	r2.cnt = r.cnt - n;
	r2.score = (int32_t)(r.score * ((float)r2.cnt / r.cnt) + .499);
	r2.as = r.as + n;
	if (r.parent == r.id) r2.parent = -2;
	set_coor(r2, qlen, a);
	r.cnt -= r2.cnt;
	r.score -= r2.score;
	set_coor(r, qlen, a);
	r.split |= 1, r2.split |= 2;
}

static void filter_regs(const mm_mapopt_t &opt, int qlen, std::vector<Reg> &regs) // hit.c:290-309
{
	size_t k = 0;
	for (size_t i = 0; i < regs.size(); ++i) {
		Reg &r = regs[i];
		int flt = 0;
		if (!r.inv && r.cnt < opt.min_cnt) flt = 1;
		if (r.has_p) {
			if (r.mlen < opt.min_chain_score) flt = 1;
			else if (r.dp_max < opt.min_dp_max) flt = 1;
			else if (r.qs > qlen * opt.max_clip_ratio && qlen - r.qe > qlen * opt.max_clip_ratio) flt = 1;
		}
		if (!flt) { if (k < i) regs[k] = std::move(regs[i]); ++k; }
	}
	regs.resize(k);
}

static void hit_sort(std::vector<Reg> &regs) // hit.c:188-218
{
	const int n = (int)regs.size();
	if (n <= 1) return;
	std::vector<u128> aux; aux.reserve((size_t)n);
	for (int i = 0; i < n; ++i)
		if (regs[i].inv || regs[i].cnt > 0) {
			int score = regs[i].has_p ? regs[i].dp_max : regs[i].score;
			aux.push_back(u128{(uint64_t)score << 32 | regs[i].hash, (uint64_t)i});
		}
	sort128(aux);
	std::vector<Reg> t; t.reserve(aux.size());
	for (int i = (int)aux.size() - 1; i >= 0; --i) t.push_back(std::move(regs[aux[i].y]));
	regs.swap(t);
}

static void set_mapq(std::vector<Reg> &regs, int min_chain_sc, int match_sc, int rep_len) // hit.c:396-466 (is_sr = 0)
{
	static const float q_coef = 40.0f;
	int64_t sum_sc = 0;
	const int n_regs = (int)regs.size();
	if (n_regs == 0) return;
	for (auto &r : regs) if (r.parent == r.id) sum_sc += r.score;
	float uniq_ratio = (float)sum_sc / (sum_sc + rep_len);
	for (auto &r : regs) {
		if (r.inv) r.mapq = 0;
		else if (r.parent == r.id) {
			int mapq, subsc;
			float pen_s1 = (r.score > 100 ? 1.0f : 0.01f * r.score) * uniq_ratio;
			float pen_cm = r.cnt > 10 ? 1.0f : 0.1f * r.cnt;
			pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
			subsc = r.subsc > min_chain_sc ? r.subsc : min_chain_sc;
			if (r.has_p && r.dp_max2 > 0 && r.dp_max > 0) {
				float identity = (float)r.mlen / r.blen;
				float x = (float)r.dp_max2 * subsc / r.dp_max / r.score0;
				mapq = (int)(identity * pen_cm * q_coef * (1.0f - x * x) * logf((float)r.dp_max / match_sc));
				int mapq_alt = (int)(6.02f * identity * identity * (r.dp_max - r.dp_max2) / match_sc + .499f);
				mapq = mapq < mapq_alt ? mapq : mapq_alt;
			} else {
				float x = (float)subsc / r.score0;
				if (r.has_p) { float identity = (float)r.mlen / r.blen; mapq = (int)(identity * pen_cm * q_coef * (1.0f - x) * logf((float)r.dp_max / match_sc)); }
				else mapq = (int)(pen_cm * q_coef * (1.0f - x) * logf(r.score));
			}
			mapq -= (int)(4.343f * logf(r.n_sub + 1) + .499f);
			mapq = mapq > 0 ? mapq : 0;
			r.mapq = mapq < 60 ? mapq : 60;
			if (r.has_p && r.dp_max > r.dp_max2 && r.mapq == 0) r.mapq = 1;
		} else r.mapq = 0;
	}
	// mm_set_inv_mapq (hit.c:396-419)
	if (n_regs < 3) return;
	bool any = false; for (auto &r : regs) any |= r.inv != 0;
	if (!any) return;
	std::vector<u128> aux;
	for (int i = 0; i < n_regs; ++i) if (regs[i].parent == i || regs[i].parent < 0) aux.push_back(u128{(uint64_t)regs[i].rid << 32 | (uint64_t)(uint32_t)regs[i].rs, (uint64_t)i});
	sort128(aux);
	for (int i = 1; i + 1 < (int)aux.size(); ++i) {
		Reg &inv = regs[aux[i].y];
		if (inv.inv) { Reg &l = regs[aux[i-1].y], &r = regs[aux[i+1].y]; inv.mapq = l.mapq < r.mapq ? l.mapq : r.mapq; }
	}
}

// ---------------- align.c helpers ----------------
struct SeqAccess {
	const SeqSet *S;
	inline const uint8_t *tptr(int rid) const { return S->h_nt4.data() + S->off[rid]; }
	void target(int rid, int32_t st, int32_t en, std::vector<uint8_t> &out) const { // index.c:152-162
		out.clear();
		int32_t len = (int32_t)S->len[rid];
		if (st >= len || st < 0) return;
		if (en > len) en = len;
		if (en > st) out.assign(tptr(rid) + st, tptr(rid) + en);
	}
	void query(int qid, int rev, int32_t st, int32_t en, std::vector<uint8_t> &out) const { // align.c:970-975
		const uint8_t *q = tptr(qid); const int32_t qlen = (int32_t)S->len[qid];
		out.resize((size_t)(en > st ? en - st : 0));
		if (!rev) { for (int32_t i = st; i < en; ++i) out[i - st] = q[i]; }
		else for (int32_t i = st; i < en; ++i) { uint8_t c = q[qlen - 1 - i]; out[i - st] = c < 4 ? 3 - c : 4; }
	}
};

static void gen_mat(int8_t *mat, int a, int b, int sc_ambi) // align.c:9-22
{
	a = a < 0 ? -a : a; b = b > 0 ? -b : b; sc_ambi = sc_ambi > 0 ? -sc_ambi : sc_ambi;
	for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) mat[i * 5 + j] = (int8_t)(i == j ? a : b); mat[i * 5 + 4] = (int8_t)sc_ambi; }
	for (int j = 0; j < 5; ++j) mat[20 + j] = (int8_t)sc_ambi;
}

// striped int16 local alignment (ksw2_ll_sse.c:37-152), evaluated lane by lane; see oracle/pgo_ksw.c for the derivation
static inline int16_t adds16(int a, int b) { int s = a + b; return (int16_t)(s > 32767 ? 32767 : s < -32768 ? -32768 : s); }
static inline int16_t subsu16(int16_t a, int16_t b) { uint16_t x = (uint16_t)a, y = (uint16_t)b; return (int16_t)(x > y ? x - y : 0); }
static inline int16_t max16(int16_t a, int16_t b) { return a > b ? a : b; }
static int ll_i16(int qlen, const uint8_t *query, const int8_t *mat, int tlen, const uint8_t *target, int gapo, int gape, int *qe, int *te)
{
	const int m = 5, slen = (qlen + 7) / 8, qlen8 = slen * 8;
	int gmax = 0;
	std::vector<int16_t> prof((size_t)m * qlen8), H0v((size_t)qlen8, 0), H1v((size_t)qlen8, 0), E((size_t)qlen8, 0), Hmax((size_t)qlen8, 0);
	int16_t *H0 = H0v.data(), *H1 = H1v.data();
	const int16_t gapoe = (int16_t)(gapo + gape), ge = (int16_t)gape;
	for (int a = 0; a < m; ++a) for (int j = 0; j < slen; ++j) for (int l = 0; l < 8; ++l) {
		int pos = j + l * slen; prof[((size_t)a * slen + j) * 8 + l] = pos >= qlen ? 0 : mat[a * m + query[pos]];
	}
	*qe = *te = -1;
	for (int i = 0; i < tlen; ++i) {
		int16_t f[8], h[8], mx[8], e[8];
		const int16_t *S = prof.data() + (size_t)target[i] * slen * 8;
		bool done = false;
		for (int l = 0; l < 8; ++l) f[l] = 0, mx[l] = 0;
		h[0] = 0; for (int l = 1; l < 8; ++l) h[l] = H0[(slen - 1) * 8 + l - 1];
		for (int j = 0; j < slen; ++j) for (int l = 0; l < 8; ++l) {
			int16_t hh = adds16(h[l], S[j * 8 + l]);
			e[l] = E[j * 8 + l];
			hh = max16(hh, e[l]); hh = max16(hh, f[l]);
			mx[l] = max16(mx[l], hh);
			H1[j * 8 + l] = hh;
			hh = subsu16(hh, gapoe);
			e[l] = max16(subsu16(e[l], ge), hh); E[j * 8 + l] = e[l];
			f[l] = max16(subsu16(f[l], ge), hh);
			h[l] = H0[j * 8 + l];
		}
		for (int k = 0; k < 8 && !done; ++k) {
			for (int l = 7; l > 0; --l) f[l] = f[l - 1];
			f[0] = 0;
			for (int j = 0; j < slen; ++j) {
				bool any = false;
				for (int l = 0; l < 8; ++l) {
					int16_t hh = max16(H1[j * 8 + l], f[l]);
					H1[j * 8 + l] = hh;
					hh = subsu16(hh, gapoe);
					f[l] = subsu16(f[l], ge);
					if (f[l] > hh) any = true;
				}
				if (!any) { done = true; break; }
			}
		}
		int imax = 0; for (int l = 0; l < 8; ++l) if (mx[l] > imax) imax = mx[l];
		if (imax >= gmax) { gmax = imax, *te = i; memcpy(Hmax.data(), H1, (size_t)qlen8 * 2); }
		std::swap(H0, H1);
	}
	for (int i = 0; i < qlen8; ++i) if ((int)(uint16_t)Hmax[i] == gmax) *qe = i / 8 + i % 8 * slen;
	return gmax;
}

static void track_zdrop(int32_t score, int i, int j, int32_t *max, int *max_i, int *max_j, int e, int *max_zdrop, int pos[2][2]) // align.c:32-45
{
	if (score < *max) {
		int li = i - *max_i, lj = j - *max_j, diff = li > lj ? li - lj : lj - li, z = *max - score - diff * e;
		if (z > *max_zdrop) { *max_zdrop = z; pos[0][0] = *max_i, pos[0][1] = i; pos[1][0] = *max_j, pos[1][1] = j; }
	} else *max = score, *max_i = i, *max_j = j;
}
// mm_test_zdrop (align.c:47-89), first half: the walk along the CIGAR.  Returns max_zdrop and, in pos, the window of the
// worst drop.  The second half (the local alignment of the window against its reverse complement, align.c:78-86) is a
// separate problem: on the GPU (pga_ll.hip) or, for windows the kernel does not take, ll_i16 here.
static int zdrop_walk(const mm_mapopt_t &opt, const uint8_t *qseq, const uint8_t *tseq, uint32_t n_cigar, const uint32_t *cigar, const int8_t *mat, int pos[2][2])
{
	int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
	pos[0][0] = pos[0][1] = pos[1][0] = pos[1][1] = -1;
	for (uint32_t k = 0; k < n_cigar; ++k) {
		uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) {
			for (uint32_t l = 0; l < len; ++l) { score += mat[tseq[i + l] * 5 + qseq[j + l]]; track_zdrop(score, i + (int)l, j + (int)l, &max, &max_i, &max_j, opt.e, &max_zdrop, pos); }
			i += len, j += len;
		} else if (op == 1 || op == 2 || op == 3) {
			score -= opt.q + opt.e * (int)len;
			if (op == 1) j += len; else i += len;
			track_zdrop(score, i, j, &max, &max_i, &max_j, opt.e, &max_zdrop, pos);
		}
	}
	return max_zdrop;
}
static inline bool zdrop_wants_inversion_test(const mm_mapopt_t &opt, int max_zdrop, const int pos[2][2])
{
	const int q_len = pos[1][1] - pos[1][0], t_len = pos[0][1] - pos[0][0];
	return !(opt.flag & (MM_F_SPLICE|MM_F_SR|MM_F_FOR_ONLY|MM_F_REV_ONLY)) && max_zdrop > opt.zdrop_inv && q_len < opt.max_gap && t_len < opt.max_gap;
}
static inline bool ll_on_device(const mm_mapopt_t &opt, int q_len, int t_len)
{
	const int q8 = (q_len + 7) / 8 * 8;
	return q_len > 0 && t_len > 0 && q8 <= PGA_LL_MAX_LEN && t_len <= PGA_LL_MAX_LEN && (int64_t)(opt.a > 0 ? opt.a : -opt.a) * q8 < 32000;
}

// A sufficient condition for mm_test_zdrop (align.c:47-89) to return 0 that needs no sequence: every z it tracks is at
// most max - score, i.e. at most the sum of all score decrements along the path.  The global pass's score fixes how
// much the match columns fall short of all-matches (a*L - score - gap costs; the dual-affine gap costs are bounded
// from below by their cheapest form), mm_test_zdrop itself charges q + e*len per gap.  If even that total cannot exceed
// zdrop (nor zdrop_inv, which gates the inversion test), the answer is 0.
static bool zdrop_impossible(const mm_mapopt_t &opt, const DpRes &ez, const uint32_t *cigar)
{
	if (ez.zdropped || ez.n_cigar <= 0 || ez.score <= -0x3fffffff) return false;
	int64_t L = 0, g_dp = 0, g_test = 0;
	for (int k = 0; k < ez.n_cigar; ++k) {
		const int64_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) L += len;
		else if (op == 1 || op == 2 || op == 3) {
			const int64_t c1 = opt.q + (int64_t)opt.e * len, c2 = opt.q2 + (int64_t)opt.e2 * len;
			g_dp += c1 < c2 ? c1 : c2;
			g_test += c1;
		} else return false;
	}
	const int64_t shortfall = (int64_t)opt.a * L - (int64_t)ez.score - g_dp;
	if (shortfall < 0) return false;                 // not a plain match/mismatch matrix: leave it to the full test
	const int64_t lim = opt.zdrop < opt.zdrop_inv ? opt.zdrop : opt.zdrop_inv;
	return shortfall + g_test <= lim;
}

static void cigar_append(Reg &r, uint32_t n_cigar, const uint32_t *cigar) // align.c:291-314
{
	if (n_cigar == 0) return;
	r.has_p = true;
	if (!r.cigar.empty() && (r.cigar.back() & 0xf) == (cigar[0] & 0xf)) {
		r.cigar.back() += cigar[0] >> 4 << 4;
		r.cigar.insert(r.cigar.end(), cigar + 1, cigar + n_cigar);
	} else r.cigar.insert(r.cigar.end(), cigar, cigar + n_cigar);
}

static void fix_cigar(Reg &r, const uint8_t *qseq, const uint8_t *tseq, int *qshift, int *tshift) // align.c:91-167
{
	std::vector<uint32_t> &cg = r.cigar;
	int32_t toff = 0, qoff = 0, to_shrink = 0;
	uint32_t k, n_cigar = (uint32_t)cg.size();
	*qshift = *tshift = 0;
	if (n_cigar <= 1) return;
	for (k = 0; k < n_cigar; ++k) {
		uint32_t op = cg[k] & 0xf, len = cg[k] >> 4;
		if (len == 0) to_shrink = 1;
		if (op == 0) toff += len, qoff += len;
		else if (op == 1 || op == 2) {
			if (k > 0 && k < n_cigar - 1 && (cg[k-1] & 0xf) == 0 && (cg[k+1] & 0xf) == 0) {
				int l, prev_len = (int)(cg[k-1] >> 4);
				const uint8_t *sq = op == 1 ? qseq : tseq; int32_t o = op == 1 ? qoff : toff;
				for (l = 0; l < prev_len; ++l) if (sq[o - 1 - l] != sq[o + (int)len - 1 - l]) break;
				if (l > 0) cg[k-1] -= (uint32_t)l << 4, cg[k+1] += (uint32_t)l << 4, qoff -= l, toff -= l;
				if (l == prev_len) to_shrink = 1;
			}
			if (op == 1) qoff += len; else toff += len;
		} else if (op == 3) toff += len;
	}
	for (k = 0; k + 2 < n_cigar; ++k) {
		if ((cg[k] & 0xf) > 0 && (cg[k] & 0xf) + (cg[k+1] & 0xf) == 3) {
			uint32_t l, s[3] = {0, 0, 0};
			for (l = k; l < n_cigar; ++l) { uint32_t op = cg[l] & 0xf; if (op == 1 || op == 2 || cg[l] >> 4 == 0) s[op] += cg[l] >> 4; else break; }
			if (s[1] > 0 && s[2] > 0 && l - k > 2) {
				cg[k] = s[1] << 4 | 1; cg[k+1] = s[2] << 4 | 2;
				for (k += 2; k < l; ++k) cg[k] &= 0xf;
				to_shrink = 1;
			}
			k = l;
		}
	}
	if (to_shrink) {
		uint32_t l = 0;
		for (k = 0; k < n_cigar; ++k) if (cg[k] >> 4 != 0) cg[l++] = cg[k];
		n_cigar = l;
		for (k = l = 0; k < n_cigar; ++k)
			if (k == n_cigar - 1 || (cg[k] & 0xf) != (cg[k+1] & 0xf)) cg[l++] = cg[k];
			else cg[k+1] += cg[k] >> 4 << 4;
		n_cigar = l;
	}
	if ((cg[0] & 0xf) == 1 || (cg[0] & 0xf) == 2) {
		int32_t l = (int32_t)(cg[0] >> 4);
		if ((cg[0] & 0xf) == 1) { if (r.rev) r.qe -= l; else r.qs += l; *qshift = l; }
		else r.rs += l, *tshift = l;
		--n_cigar;
		memmove(cg.data(), cg.data() + 1, (size_t)n_cigar * 4);
	}
	cg.resize(n_cigar);
}

static void update_extra(Reg &r, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int q, int e) // align.c:240-289
{
	if (!r.has_p) return;
	int32_t qshift, tshift, toff = 0, qoff = 0;
	double s = 0.0, max = 0.0;
	fix_cigar(r, qseq, tseq, &qshift, &tshift);
	qseq += qshift, tseq += tshift;
	r.blen = r.mlen = 0;
	const int a_match = mat[0];
	const bool same_match = a_match > 0 && mat[6] == a_match && mat[12] == a_match && mat[18] == a_match;
	for (uint32_t k = 0; k < r.cigar.size(); ++k) {
		uint32_t op = r.cigar[k] & 0xf, len = r.cigar[k] >> 4;
		if (op == 0) {
			int n_ambi = 0, n_diff = 0;
			// stretches of identical unambiguous bases, eight at a time: every step adds the same positive match score, so s only
			// grows (no clamp) and the running maximum is s at the end of the stretch; all values are integers plus the few
			// fractional bits of earlier gap terms, far inside a double's mantissa, so s + 8a is the stepwise sum exactly
			const uint8_t *pq = qseq + qoff, *pt = tseq + toff;
			for (uint32_t l = 0; l < len;) {
				if (same_match) {
					const uint32_t l0 = l;
					while (l + 8 <= len) {
						uint64_t wq, wt; memcpy(&wq, pq + l, 8); memcpy(&wt, pt + l, 8);
						if (wq != wt || (wq & 0xFCFCFCFCFCFCFCFCULL)) break;
						l += 8;
					}
					if (l > l0) { s += (double)a_match * (double)(l - l0); max = max > s ? max : s; if (l == len) break; }
				}
				int cq = pq[l], ct = pt[l];
				if (ct > 3 || cq > 3) ++n_ambi; else if (ct != cq) ++n_diff;
				s += mat[ct * 5 + cq];
				if (s < 0) s = 0; else max = max > s ? max : s;
				++l;
			}
			r.blen += len - n_ambi, r.mlen += len - (n_ambi + n_diff), r.n_ambi += n_ambi;
			toff += len, qoff += len;
		} else if (op == 1 || op == 2) {
			int n_ambi = 0;
			const uint8_t *sq = op == 1 ? qseq + qoff : tseq + toff;
			for (uint32_t l = 0; l < len; ++l) if (sq[l] > 3) ++n_ambi;
			r.blen += len - n_ambi, r.n_ambi += n_ambi;
			s -= q + (double)e * mg_log2_host((float)(1.0 + len));
			if (s < 0) s = 0;
			if (op == 1) qoff += len; else toff += len;
		} else if (op == 3) toff += len;
	}
	r.dp_max = (int32_t)(max + .499);
}

static int *long_gaps(int as1, int cnt1, const u128 *a, int min_gap, std::vector<int> &K) // align.c:373-390
{
	K.clear();
	int n = 0;
	for (int i = 1; i < cnt1; ++i) {
		int gap = ((int32_t)a[as1+i].y - (int32_t)a[as1+i-1].y) - ((int32_t)a[as1+i].x - (int32_t)a[as1+i-1].x);
		if (gap < -min_gap || gap > min_gap) ++n;
	}
	if (n <= 1) return nullptr;
	for (int i = 1; i < cnt1; ++i) {
		int gap = ((int32_t)a[as1+i].y - (int32_t)a[as1+i-1].y) - ((int32_t)a[as1+i].x - (int32_t)a[as1+i-1].x);
		if (gap < -min_gap || gap > min_gap) K.push_back(i);
	}
	return K.data();
}
static void filter_bad_seeds(int as1, int cnt1, u128 *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt) // align.c:392-431
{
	std::vector<int> Kv; int *K = long_gaps(as1, cnt1, a, min_gap, Kv);
	if (!K) return;
	int n = (int)Kv.size(), max = 0, max_st = -1, max_en = -1, i, k;
	for (k = 0;; ++k) {
		int gap, l, n_ins = 0, n_del = 0, qs, rs, max_diff = 0, max_diff_l = -1;
		if (k == n || k >= max_en) {
			if (max_en > 0) for (i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= SEED_IGNORE;
			max = 0, max_st = max_en = -1;
			if (k == n) break;
		}
		i = K[k];
		gap = ((int32_t)a[as1+i].y - (int32_t)a[as1+i-1].y) - (int32_t)(a[as1+i].x - a[as1+i-1].x);
		if (gap > 0) n_ins += gap; else n_del += -gap;
		qs = (int32_t)a[as1+i-1].y; rs = (int32_t)a[as1+i-1].x;
		for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
			int j = K[l], diff;
			if ((int32_t)a[as1+j].y - qs > max_ext_len || (int32_t)a[as1+j].x - rs > max_ext_len) break;
			gap = ((int32_t)a[as1+j].y - (int32_t)a[as1+j-1].y) - (int32_t)(a[as1+j].x - a[as1+j-1].x);
			if (gap > 0) n_ins += gap; else n_del += -gap;
			diff = n_ins + n_del - abs(n_ins - n_del);
			if (max_diff < diff) max_diff = diff, max_diff_l = l;
		}
		if (max_diff > diff_thres && max_diff > max) max = max_diff, max_st = k, max_en = max_diff_l;
	}
}
static void filter_bad_seeds_alt(int as1, int cnt1, u128 *a, int min_gap, int max_ext) // align.c:433-469
{
	std::vector<int> Kv; int *K = long_gaps(as1, cnt1, a, min_gap, Kv);
	if (!K) return;
	int n = (int)Kv.size();
	for (int k = 0; k < n;) {
		int i = K[k], l;
		int gap1 = ((int32_t)a[as1+i].y - (int32_t)a[as1+i-1].y) - ((int32_t)a[as1+i].x - (int32_t)a[as1+i-1].x);
		int re1 = (int32_t)a[as1+i].x, qe1 = (int32_t)a[as1+i].y;
		gap1 = gap1 > 0 ? gap1 : -gap1;
		for (l = k + 1; l < n; ++l) {
			int j = K[l], gap2, q_span_pre, rs2, qs2, m;
			if ((int32_t)a[as1+j].y - qe1 > max_ext || (int32_t)a[as1+j].x - re1 > max_ext) break;
			gap2 = ((int32_t)a[as1+j].y - (int32_t)a[as1+j-1].y) - (int32_t)(a[as1+j].x - a[as1+j-1].x);
			q_span_pre = (int)(a[as1+j-1].y >> 32 & 0xff);
			rs2 = (int32_t)a[as1+j-1].x + q_span_pre; qs2 = (int32_t)a[as1+j-1].y + q_span_pre;
			m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
			gap2 = gap2 > 0 ? gap2 : -gap2;
			if (m > gap1 + gap2) break;
			re1 = (int32_t)a[as1+j].x, qe1 = (int32_t)a[as1+j].y;
			gap1 = gap2;
		}
		if (l > k + 1) {
			int end = K[l - 1];
			for (int j = K[k]; j < end; ++j) a[as1 + j].y |= SEED_IGNORE;
			a[as1 + end].y |= SEED_LONG_JOIN;
		}
		k = l;
	}
}
static void fix_bad_ends(const Reg &r, const u128 *a, int bw, int min_match, int32_t *as, int32_t *cnt) // align.c:471-509
{
	int32_t i, l, m;
	*as = r.as, *cnt = r.cnt;
	if (r.cnt < 3) return;
	m = l = (int32_t)(a[r.as].y >> 32 & 0xff);
	for (i = r.as + 1; i < r.as + r.cnt - 1; ++i) {
		int32_t lq, lr, min, max, q_span = (int32_t)(a[i].y >> 32 & 0xff);
		if (a[i].y & SEED_LONG_JOIN) break;
		lr = (int32_t)a[i].x - (int32_t)a[i-1].x; lq = (int32_t)a[i].y - (int32_t)a[i-1].y;
		min = lr < lq ? lr : lq, max = lr > lq ? lr : lq;
		if (max - min > l >> 1) *as = i;
		l += min; m += min < q_span ? min : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r.mlen >> 1) break;
	}
	*cnt = r.as + r.cnt - *as;
	m = l = (int32_t)(a[r.as + r.cnt - 1].y >> 32 & 0xff);
	for (i = r.as + r.cnt - 2; i > *as; --i) {
		int32_t lq, lr, min, max, q_span = (int32_t)(a[i+1].y >> 32 & 0xff);
		if (a[i+1].y & SEED_LONG_JOIN) break;
		lr = (int32_t)a[i+1].x - (int32_t)a[i].x; lq = (int32_t)a[i+1].y - (int32_t)a[i].y;
		min = lr < lq ? lr : lq, max = lr > lq ? lr : lq;
		if (max - min > l >> 1) *cnt = i + 1 - *as;
		l += min; m += min < q_span ? min : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r.mlen >> 1) break;
	}
}

// ---------------- the per-region state machine ----------------
struct Seg { int32_t i, rs, qs, re, qe, bw1; int job1 = -1, job2 = -1, zcode = -1; int ll_job = -1; int32_t max_zdrop = 0; };

struct RegTask {
	Reg r;
	bool planned = false, done = false, is_inv = false;
	int32_t rid = 0, rev = 0, as1 = 0, cnt1 = 0;
	int32_t rs = 0, qs = 0, re = 0, qe = 0, rs0 = 0, qs0 = 0, re0 = 0, qe0 = 0;
	int32_t bw = 0;
	int left_job = -1, right_job = -1;
	bool left_done = false;
	std::vector<Seg> segs; size_t seg_k = 0;
	int32_t rs1 = 0, qs1 = 0, re1 = 0, qe1 = 0;
	bool dropped = false;
	// inversion test state (mm_align1_inv)
	int inv_state = 0;      // 0 = not evaluated, 1 = waiting for its DP problem, 2 = resolved
	int inv_ll_job = -1;    // the local-alignment query that precedes it (state 3 = waiting for that)
	int inv_job = -1; int32_t inv_q_off = 0, inv_t_off = 0, inv_ql = 0, inv_tl = 0;
};

struct QueryCtx {
	int qid = 0; int32_t qlen = 0; int base = 0;   // base: first sequence of the query's group (record rids are group-relative)
	std::vector<u128> a; int32_t n_a = 0;
	std::vector<RegTask*> list;      // the reference's regs[] order, grown by insertions
	std::vector<std::unique_ptr<RegTask>> pool;
	int rep_len = 0;
	bool finished = false;
	// DP problems of this query (ids are per query; filled between the threaded phases)
	std::vector<DpJob> jobs; std::vector<DpRes> res; std::vector<const uint32_t*> cig; std::vector<int> pending;   // cig[id] points into a per-round CIGAR pool
	std::deque<uint32_t> own_cig;     // one-operation CIGARs of the gap fills answered without a DP problem (stable addresses)
};

struct Driver {
	const SeqSet &S; const mm_mapopt_t &opt; int k;
	SeqAccess acc; int8_t mat[25];
	Driver(const SeqSet &S_, const mm_mapopt_t &o, int k_) : S(S_), opt(o), k(k_) { acc.S = &S; gen_mat(mat, o.a, o.b, o.sc_ambi); }

	static bool have(const QueryCtx &Q, int id) { return id >= 0 && (size_t)id < Q.res.size() && Q.res[id].pad == 1; }

	int request(QueryCtx &Q, int rev, int rid, int32_t qs, int32_t qlen, int32_t ts, int32_t tlen, int seq_rev, int w, int end_bonus, int zdrop, int flag)
	{
		DpJob j; memset(&j, 0, sizeof(j));
		j.t_off = S.off[Q.base + rid] + (uint64_t)ts; j.q_off = S.off[Q.qid]; j.qlen_full = Q.qlen;
		j.qs = qs; j.qlen = qlen; j.tlen = tlen; j.w = w; j.zdrop = zdrop; j.end_bonus = end_bonus; j.flag = flag;
		j.q_rev = (uint8_t)rev; j.seq_rev = (uint8_t)seq_rev;
		int id = (int)Q.jobs.size();
		Q.jobs.push_back(j);
		DpRes r; memset(&r, 0, sizeof(r));
		// problems the reference never hands to the kernel (align.c:326-328, ksw2_extd2_sse.c:82) are resolved here
		r.max_q = r.max_t = r.mqe_t = r.mte_q = -1; r.score = r.mqe = r.mte = NEG_INF;
		if (!(flag & PGA_JOB_LL) && opt.max_sw_mat > 0 && (int64_t)tlen * qlen > opt.max_sw_mat) { r.zdropped = 1; r.pad = 1; }
		else if (qlen <= 0 || tlen <= 0) r.pad = 1;
		Q.res.push_back(r); Q.cig.push_back(nullptr);
		if (!r.pad) Q.pending.push_back(id);
		return id;
	}

	// First-pass gap fill (KSW_EZ_APPROX_MAX) of two equally long, N-free windows that differ in so few positions that the main
	// diagonal is provably the unique optimum ((a+b)*m < a + 2*min(q+e, q2+e2): every other alignment has an insertion and a
	// deletion -- the proof and its parity test are with the tile kernel, pga_ksw_fast.hip): the answer is "nM" with score
	// a*(n-m) - b*m, and no problem is sent to the GPU at all.  Returns false if the windows do not qualify.
	bool gap_fill_by_identity(QueryCtx &Q, int rev, int rid, int32_t qs, int32_t ts, int32_t n, int &id_out)
	{
		if (n <= 0) return false;
		if (opt.max_sw_mat > 0 && (int64_t)n * n > opt.max_sw_mat) return false;   // align.c:326: the reference never aligns such a window (request() answers z-dropped)
		const int a_ = mat[0], b_ = -mat[1];
		const int g1 = std::min(opt.q + opt.e, opt.q2 + opt.e2);
		const int64_t lim = (int64_t)a_ + 2 * g1;                  // (a+b)*m < lim
		if (a_ <= 0 || b_ <= 0) return false;
		const uint8_t *t = acc.tptr(Q.base + rid) + ts, *q = acc.tptr(Q.qid);
		int m = 0;
		// eight bases per step: codes are 0..3 (anything else disqualifies), a differing byte has bit 0 or bit 1 set in the XOR
		const int64_t m_max = (lim - 1) / (a_ + b_);               // (a+b)*m < lim  <=>  m <= m_max
		int32_t i = 0;
		if (!rev) {
			const uint8_t *qq = q + qs;
			for (; i + 8 <= n; i += 8) {
				uint64_t wt, wq; memcpy(&wt, t + i, 8); memcpy(&wq, qq + i, 8);
				if ((wt | wq) & 0xFCFCFCFCFCFCFCFCULL) return false;
				const uint64_t d = wt ^ wq;
				if (d) { m += __builtin_popcountll((d | d >> 1) & 0x0101010101010101ULL); if (m > m_max) return false; }
			}
			for (; i < n; ++i) {
				const uint8_t x = t[i], y = qq[i];
				if ((x | y) > 3) return false;
				if (x != y && ++m > m_max) return false;
			}
		} else {
			const uint8_t *qq = q + (Q.qlen - 1 - qs);                // base i of the reverse strand window = 3 - q[qlen-1-(qs+i)]
			for (; i + 8 <= n; i += 8) {
				uint64_t wt, wq; memcpy(&wt, t + i, 8); memcpy(&wq, qq - i - 7, 8);
				if ((wt | wq) & 0xFCFCFCFCFCFCFCFCULL) return false;
				const uint64_t d = wt ^ (__builtin_bswap64(wq) ^ 0x0303030303030303ULL);   // reversed, complemented (3 - y == y ^ 3 for 0..3)
				if (d) { m += __builtin_popcountll((d | d >> 1) & 0x0101010101010101ULL); if (m > m_max) return false; }
			}
			for (; i < n; ++i) {
				const uint8_t x = t[i], y = qq[-i];
				if ((x | y) > 3) return false;
				if (x != (uint8_t)(3 - y) && ++m > m_max) return false;
			}
		}
		DpJob j; memset(&j, 0, sizeof(j));
		j.t_off = S.off[Q.base + rid] + (uint64_t)ts; j.q_off = S.off[Q.qid]; j.qlen_full = Q.qlen; j.qs = qs; j.qlen = n; j.tlen = n; j.flag = EZ_APPROX_MAX;
		DpRes r; memset(&r, 0, sizeof(r));
		r.max = 0; r.max_q = r.max_t = r.mqe_t = r.mte_q = -1; r.mqe = r.mte = NEG_INF;
		r.score = a_ * (n - m) - b_ * m; r.n_cigar = 1; r.pad = 1;
		Q.own_cig.push_back((uint32_t)n << 4);
		id_out = (int)Q.jobs.size();
		Q.jobs.push_back(j); Q.res.push_back(r); Q.cig.push_back(&Q.own_cig.back());
		return true;
	}

	// ---- plan (align.c:583-700): everything mm_align1 decides before its first DP call ----
	void plan(QueryCtx &Q, RegTask &T)
	{
		Reg &r = T.r; u128 *a = Q.a.data(); const int32_t qlen = Q.qlen, n_a = Q.n_a;
		T.planned = true;
		if (r.cnt == 0) { T.done = true; return; }
		T.rid = (int32_t)(a[r.as].x << 1 >> 33), T.rev = (int32_t)(a[r.as].x >> 63);
		const int32_t tlen_ref = (int32_t)S.len[Q.base + T.rid];
		int32_t bw = (int)(opt.bw * 1.5 + 1.), bw_long = (int)(opt.bw_long * 1.5 + 1.);
		if (bw_long < bw) bw_long = bw;
		T.bw = bw;
		int32_t as1, cnt1, rs, qs, re, qe, rs0, qs0, re0, qe0, rs1, qs1, re1, qe1, i, l;
		if (!(opt.flag & MM_F_NO_END_FLT)) fix_bad_ends(r, a, opt.bw, opt.min_chain_score * 2, &as1, &cnt1);
		else as1 = r.as, cnt1 = r.cnt;
		filter_bad_seeds(as1, cnt1, a, 10, 40, opt.max_gap >> 1, 10);
		filter_bad_seeds_alt(as1, cnt1, a, 30, opt.max_gap >> 1);
		rs = (int32_t)a[as1].x - (k >> 1), qs = (int32_t)a[as1].y - (k >> 1);                       // mm_adjust_minier, non-HPC
		re = (int32_t)a[as1 + cnt1 - 1].x - (k >> 1), qe = (int32_t)a[as1 + cnt1 - 1].y - (k >> 1);
		rs0 = (int32_t)a[r.as].x + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
		qs0 = (int32_t)a[r.as].y + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
		if (rs0 < 0) rs0 = 0;
		rs1 = qs1 = 0;
		for (i = r.as - 1, l = 0; i >= 0 && a[i].x >> 32 == a[r.as].x >> 32; --i) {
			int32_t x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff), y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
			if (x < rs0 && y < qs0) {
				if (++l > opt.min_cnt) { l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y; rs1 = rs0 - l, qs1 = qs0 - l; if (rs1 < 0) rs1 = 0; break; }
			}
		}
		if (qs > 0 && rs > 0) {
			l = qs < opt.max_gap ? qs : opt.max_gap;
			qs1 = qs1 > qs - l ? qs1 : qs - l;
			qs0 = qs0 < qs1 ? qs0 : qs1;
			l += l * opt.a > opt.q ? (l * opt.a - opt.q) / opt.e : 0;
			l = l < opt.max_gap ? l : opt.max_gap;
			l = l < rs ? l : rs;
			rs1 = rs1 > rs - l ? rs1 : rs - l;
			rs0 = rs0 < rs1 ? rs0 : rs1;
			rs0 = rs0 < rs ? rs0 : rs;
		} else rs0 = rs, qs0 = qs;
		re0 = (int32_t)a[r.as + r.cnt - 1].x + 1, qe0 = (int32_t)a[r.as + r.cnt - 1].y + 1;
		re1 = tlen_ref, qe1 = qlen;
		for (i = r.as + r.cnt, l = 0; i < n_a && a[i].x >> 32 == a[r.as].x >> 32; ++i) {
			int32_t x = (int32_t)a[i].x + 1, y = (int32_t)a[i].y + 1;
			if (x > re0 && y > qe0) {
				if (++l > opt.min_cnt) { l = x - re0 > y - qe0 ? x - re0 : y - qe0; re1 = re0 + l, qe1 = qe0 + l; break; }
			}
		}
		if (qe < qlen && re < tlen_ref) {
			l = qlen - qe < opt.max_gap ? qlen - qe : opt.max_gap;
			qe1 = qe1 < qe + l ? qe1 : qe + l;
			qe0 = qe0 > qe1 ? qe0 : qe1;
			l += l * opt.a > opt.q ? (l * opt.a - opt.q) / opt.e : 0;
			l = l < opt.max_gap ? l : opt.max_gap;
			l = l < tlen_ref - re ? l : tlen_ref - re;
			re1 = re1 < re + l ? re1 : re + l;
			re0 = re0 > re1 ? re0 : re1;
		} else re0 = re, qe0 = qe;
		if (a[r.as].y & SEED_SELF) {
			int max_ext = r.qs > r.rs ? r.qs - r.rs : r.rs - r.qs;
			if (r.rs - rs0 > max_ext) rs0 = r.rs - max_ext;
			if (r.qs - qs0 > max_ext) qs0 = r.qs - max_ext;
			max_ext = r.qe > r.re ? r.qe - r.re : r.re - r.qe;
			if (re0 - r.re > max_ext) re0 = r.re + max_ext;
			if (qe0 - r.qe > max_ext) qe0 = r.qe + max_ext;
		}
		T.as1 = as1, T.cnt1 = cnt1, T.rs = rs, T.qs = qs, T.rs0 = rs0, T.qs0 = qs0, T.re0 = re0, T.qe0 = qe0;
		// left extension (align.c:702-722): reversed windows, right-aligned gaps, reversed CIGAR
		if (qs > 0 && rs > 0)
			T.left_job = request(Q, T.rev, T.rid, qs0, qs - qs0, rs0, rs - rs0, 1, bw, opt.end_bonus, r.split_inv ? opt.zdrop_inv : opt.zdrop, EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR);
		else T.left_done = true, T.rs1 = rs, T.qs1 = qs;
		// gap-fill segments (align.c:726-745): determined by the anchors alone
		for (i = 1; i < cnt1; ++i) {
			if ((a[as1+i].y & (SEED_IGNORE | SEED_TANDEM)) && i != cnt1 - 1) continue;
			re = (int32_t)a[as1 + i].x - (k >> 1), qe = (int32_t)a[as1 + i].y - (k >> 1);
			if (i == cnt1 - 1 || (a[as1+i].y & SEED_LONG_JOIN) || (qe - qs >= opt.min_ksw_len && re - rs >= opt.min_ksw_len)) {
				Seg sg; sg.i = i, sg.rs = rs, sg.qs = qs, sg.re = re, sg.qe = qe, sg.bw1 = bw_long;
				if (a[as1+i].y & SEED_LONG_JOIN) sg.bw1 = qe - qs > re - rs ? qe - qs : re - rs;
				if (!(qe - qs == re - rs && sg.bw1 >= qe - qs && gap_fill_by_identity(Q, T.rev, T.rid, qs, rs, qe - qs, sg.job1)))
					sg.job1 = request(Q, T.rev, T.rid, qs, qe - qs, rs, re - rs, 0, sg.bw1, -1, opt.zdrop, EZ_APPROX_MAX);
				T.segs.push_back(sg);
				rs = re, qs = qe;
			}
		}
		T.re = re, T.qe = qe;   // the last adjusted anchor (what the right extension starts from when nothing dropped)
		if (cnt1 == 1) T.re = (int32_t)a[as1].x - (k >> 1), T.qe = (int32_t)a[as1].y - (k >> 1);
		// right extension (align.c:789-805), speculative: only used when no segment z-drops
		if (T.qe < qe0 && T.re < re0)
			T.right_job = request(Q, T.rev, T.rid, T.qe, qe0 - T.qe, T.re, re0 - T.re, 0, bw, opt.end_bonus, opt.zdrop, EZ_EXTZ_ONLY);
	}

	// ---- advance: returns true when the region is complete; r2 receives a split-off region (cnt>0) ----
	bool advance(QueryCtx &Q, RegTask &T, Reg &r2)
	{
		Reg &r = T.r; u128 *a = Q.a.data(); const int32_t qlen = Q.qlen;
		r2.cnt = 0;
		if (T.done) return true;
		if (!T.left_done) {
			if (!have(Q, T.left_job)) return false;
			const DpRes &ez = Q.res[T.left_job];
			if (ez.n_cigar > 0) { cigar_append(r, (uint32_t)ez.n_cigar, Q.cig[T.left_job]); r.dp_score += ez.max; }
			T.rs1 = T.rs - (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
			T.qs1 = T.qs - (ez.reach_end ? T.qs - T.qs0 : ez.max_q + 1);
			T.left_done = true;
		}
		if (T.seg_k == 0) T.re1 = T.rs, T.qe1 = T.qs;
		std::vector<uint8_t> qw, tw;
		ScopeNs sc_seg(0);
		while (T.seg_k < T.segs.size() && !T.dropped) {
			Seg &sg = T.segs[T.seg_k];
			if (!have(Q, sg.job1)) return false;
			// results and CIGARs of the following segments lie wherever their kernels finished: start fetching them now
			for (size_t ahead = 2; ahead <= 4; ahead += 2) if (T.seg_k + ahead < T.segs.size()) {
				const int nj = T.segs[T.seg_k + ahead].job1;
				if (nj >= 0 && (size_t)nj < Q.res.size()) { __builtin_prefetch(&Q.res[nj]); if (Q.cig[nj]) __builtin_prefetch(Q.cig[nj]); }
			}
			T.re1 = sg.re, T.qe1 = sg.qe;
			int final_job = sg.job1;
			if (sg.zcode < 0 && sg.ll_job < 0) {
				const DpRes &e1 = Q.res[sg.job1];
				if (zdrop_impossible(opt, e1, Q.cig[sg.job1])) { sg.zcode = 0; if (g_prof) g_ns[5] += 1; }
				else {
					if (g_prof) g_ns[6] += 1;
					ScopeNs sc(7);
					acc.query(Q.qid, T.rev, sg.qs, sg.qe, qw); acc.target(Q.base + T.rid, sg.rs, sg.re, tw);
					int pos[2][2];
					sg.max_zdrop = zdrop_walk(opt, qw.data(), tw.data(), (uint32_t)e1.n_cigar, Q.cig[sg.job1], mat, pos);
					const int q_len = pos[1][1] - pos[1][0], t_len = pos[0][1] - pos[0][0];
					if (!zdrop_wants_inversion_test(opt, sg.max_zdrop, pos)) sg.zcode = sg.max_zdrop > opt.zdrop ? 1 : 0;
					else if (ll_on_device(opt, q_len, t_len)) {
						// the window against its own reverse complement: query = the other strand, [L - (qs+pos11), +q_len)
						sg.ll_job = request(Q, 1 - T.rev, T.rid, qlen - (sg.qs + pos[1][1]), q_len, sg.rs + pos[0][0], t_len, 0, 0, -1, 0, PGA_JOB_LL);
						// whatever the answer, the second pass runs when both thresholds agree (they do in every asm preset)
						if (opt.zdrop == opt.zdrop_inv) sg.job2 = request(Q, T.rev, T.rid, sg.qs, sg.qe - sg.qs, sg.rs, sg.re - sg.rs, 0, sg.bw1, -1, opt.zdrop, 0);
					} else {
						std::vector<uint8_t> qseq2((size_t)(q_len > 0 ? q_len : 0));
						int q_off, t_off;
						for (int i = 0; i < q_len; ++i) { int c = qw[pos[1][1] - i - 1]; qseq2[i] = (uint8_t)(c >= 4 ? 4 : 3 - c); }
						const int score = ll_i16(q_len, qseq2.data(), mat, t_len, tw.data() + pos[0][0], opt.q, opt.e, &q_off, &t_off);
						sg.zcode = (score >= opt.min_chain_score * opt.a && score >= opt.min_dp_max) ? 2 : (sg.max_zdrop > opt.zdrop ? 1 : 0);
					}
				}
				if (sg.zcode > 0 && sg.job2 < 0) sg.job2 = request(Q, T.rev, T.rid, sg.qs, sg.qe - sg.qs, sg.rs, sg.re - sg.rs, 0, sg.bw1, -1, sg.zcode == 2 ? opt.zdrop_inv : opt.zdrop, 0);
			}
			if (sg.zcode < 0) {
				if (!have(Q, sg.ll_job)) return false;
				const int score = Q.res[sg.ll_job].score;
				sg.zcode = (score >= opt.min_chain_score * opt.a && score >= opt.min_dp_max) ? 2 : (sg.max_zdrop > opt.zdrop ? 1 : 0);
				if (sg.zcode > 0 && sg.job2 < 0) sg.job2 = request(Q, T.rev, T.rid, sg.qs, sg.qe - sg.qs, sg.rs, sg.re - sg.rs, 0, sg.bw1, -1, sg.zcode == 2 ? opt.zdrop_inv : opt.zdrop, 0);
			}
			if (sg.zcode != 0) { if (!have(Q, sg.job2)) return false; final_job = sg.job2; }
			const DpRes &ez = Q.res[final_job];
			if (ez.n_cigar > 0) cigar_append(r, (uint32_t)ez.n_cigar, Q.cig[final_job]);
			if (ez.zdropped) {
				r.has_p = true;
				int j;
				for (j = sg.i - 1; j >= 0; --j) if ((int32_t)a[T.as1 + j].x <= sg.rs + ez.max_t) break;
				T.dropped = true;
				if (j < 0) j = 0;
				r.dp_score += ez.max;
				T.re1 = sg.rs + (ez.max_t + 1), T.qe1 = sg.qs + (ez.max_q + 1);
				if (T.cnt1 - (j + 1) >= opt.min_cnt) {
					split_reg(r, r2, T.as1 + j + 1 - r.as, qlen, a);
					if (r2.cnt > 0 && sg.zcode == 2) r2.split_inv = 1;
				}
				break;
			} else r.dp_score += ez.score;
			++T.seg_k;
		}
		if (!T.dropped && T.qe < T.qe0 && T.re < T.re0) {
			if (!have(Q, T.right_job)) return false;
			const DpRes &ez = Q.res[T.right_job];
			if (ez.n_cigar > 0) { cigar_append(r, (uint32_t)ez.n_cigar, Q.cig[T.right_job]); r.dp_score += ez.max; }
			T.re1 = T.re + (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
			T.qe1 = T.qe + (ez.reach_end ? T.qe0 - T.qe : ez.max_q + 1);
		}
		r.rs = T.rs1, r.re = T.re1;
		if (!T.rev) r.qs = T.qs1, r.qe = T.qe1; else r.qs = qlen - T.qe1, r.qe = qlen - T.qs1;
		if (r.has_p) {
			{ ScopeNs sc(1); acc.target(Q.base + T.rid, T.rs1, T.re1, tw); acc.query(Q.qid, (int)r.rev, T.qs1, T.qe1, qw); }
			ScopeNs sc(2);
			update_extra(r, qw.data(), tw.data(), mat, opt.q, opt.e);
		}
		T.done = true;
		return true;
	}

	// ---- mm_align1_inv (align.c:830-885) split at its DP call ----
	// returns 0 = no inversion, 1 = waiting, 2 = r_inv produced
	int inversion(QueryCtx &Q, RegTask &T, const Reg &r1, Reg &r_inv)
	{
		const Reg &r2 = T.r; const int32_t qlen = Q.qlen;
		if (T.inv_state == 0) {
			T.inv_state = 2;
			if (!(r1.split & 1) || !(r2.split & 2)) return 0;
			if (r1.id != r1.parent && r1.parent != -2) return 0;
			if (r2.id != r2.parent && r2.parent != -2) return 0;
			if (r1.rid != r2.rid || r1.rev != r2.rev) return 0;
			int ql = r1.rev ? r1.qs - r2.qe : r2.qs - r1.qe, tl = r2.rs - r1.re;
			if (ql < opt.min_chain_score || ql > opt.max_gap) return 0;
			if (tl < opt.min_chain_score || tl > opt.max_gap) return 0;
			// qseq = r1.rev ? &qseq0[0][r2.qe] : &qseq0[1][qlen - r2.qs]; both windows are reversed before the local alignment
			const int q_strand = r1.rev ? 0 : 1; const int32_t q_st = r1.rev ? r2.qe : qlen - r2.qs;
			T.inv_ql = ql, T.inv_tl = tl;
			int score, q_off, t_off;
			if (ll_on_device(opt, ql, tl)) {
				T.inv_ll_job = request(Q, q_strand, r1.rid, q_st, ql, r1.re, tl, 1, 0, -1, 0, PGA_JOB_LL);
				T.inv_state = 3;
				return 1;
			}
			std::vector<uint8_t> tw, qw;
			acc.target(Q.base + r1.rid, r1.re, r2.rs, tw);
			acc.query(Q.qid, q_strand, q_st, q_st + ql, qw);
			std::reverse(qw.begin(), qw.end()); std::reverse(tw.begin(), tw.end());
			score = ll_i16(ql, qw.data(), mat, tl, tw.data(), opt.q, opt.e, &q_off, &t_off);
			if (score < opt.min_dp_max) return 0;
			q_off = ql - (q_off + 1), t_off = tl - (t_off + 1);
			T.inv_q_off = q_off, T.inv_t_off = t_off;
			T.inv_job = request(Q, q_strand, r1.rid, q_st + q_off, ql - q_off, r1.re + t_off, tl - t_off, 0, (int)(opt.bw * 1.5), -1, opt.zdrop, EZ_EXTZ_ONLY);
			T.inv_state = 1;
		}
		if (T.inv_state == 3) {
			if (!have(Q, T.inv_ll_job)) return 1;
			const DpRes &lr = Q.res[T.inv_ll_job];
			T.inv_state = 2;
			if (lr.score < opt.min_dp_max) return 0;
			const int ql = T.inv_ql, tl = T.inv_tl;
			const int q_off = ql - (lr.max_q + 1), t_off = tl - (lr.max_t + 1);
			const int q_strand = r1.rev ? 0 : 1; const int32_t q_st = r1.rev ? r2.qe : qlen - r2.qs;
			T.inv_q_off = q_off, T.inv_t_off = t_off;
			T.inv_job = request(Q, q_strand, r1.rid, q_st + q_off, ql - q_off, r1.re + t_off, tl - t_off, 0, (int)(opt.bw * 1.5), -1, opt.zdrop, EZ_EXTZ_ONLY);
			T.inv_state = 1;
		}
		if (T.inv_state == 1) {
			if (!have(Q, T.inv_job)) return 1;
			T.inv_state = 2;
			const DpRes &ez = Q.res[T.inv_job];
			if (ez.n_cigar == 0) return 0;
			r_inv = Reg();
			cigar_append(r_inv, (uint32_t)ez.n_cigar, Q.cig[T.inv_job]);
			r_inv.dp_score = ez.max;
			r_inv.id = -1, r_inv.parent = -1, r_inv.inv = 1, r_inv.rev = !r1.rev, r_inv.rid = r1.rid;
			const int q_off = T.inv_q_off, t_off = T.inv_t_off;
			if (r_inv.rev == 0) { r_inv.qs = r2.qe + q_off; r_inv.qe = r_inv.qs + ez.max_q + 1; }
			else { r_inv.qe = r2.qs - q_off; r_inv.qs = r_inv.qe - (ez.max_q + 1); }
			r_inv.rs = r1.re + t_off; r_inv.re = r_inv.rs + ez.max_t + 1;
			std::vector<uint8_t> tw, qw;
			const int q_strand = r1.rev ? 0 : 1; const int32_t q_st = r1.rev ? r2.qe : qlen - r2.qs;
			acc.target(Q.base + r1.rid, r1.re + t_off, r2.rs, tw);
			acc.query(Q.qid, q_strand, q_st + q_off, q_st + T.inv_ql, qw);
			update_extra(r_inv, qw.data(), tw.data(), mat, opt.q, opt.e);
			return 2;
		}
		return 0;
	}
};

static int squeeze_a(std::vector<RegTask*> &list, std::vector<u128> &a) // hit.c:311-329
{
	const int n = (int)list.size();
	std::vector<uint64_t> aux((size_t)n);
	for (int i = 0; i < n; ++i) aux[i] = (uint64_t)list[i]->r.as << 32 | (uint64_t)i;
	std::sort(aux.begin(), aux.end());   // keys are unique (distinct `as`), so any sort reproduces radix_sort_64
	int as = 0;
	for (int i = 0; i < n; ++i) {
		Reg &r = list[(int32_t)aux[i]]->r;
		if (r.as != as) { memmove(&a[as], &a[r.as], (size_t)r.cnt * 16); r.as = as; }
		as += r.cnt;
	}
	return as;
}

static double event_identity(const Reg &r) // align.c:897-917
{
	int32_t n_gapo = 0, n_gap = 0;
	if (!r.has_p) return -1.0f;
	for (uint32_t c : r.cigar) { int32_t op = c & 0xf, len = (int32_t)(c >> 4); if (op == 1 || op == 2) ++n_gapo, n_gap += len; }
	return (double)r.mlen / (r.blen + (int32_t)r.n_ambi - n_gap + n_gapo);
}
static int32_t recal_max_dp(const Reg &r, double b2, int32_t match_sc) // align.c:919-934
{
	int32_t n_gap = 0, n_mis; double gap_cost = 0.0;
	if (!r.has_p) return -1;
	for (uint32_t c : r.cigar) { int32_t op = c & 0xf, len = (int32_t)(c >> 4); if (op == 1 || op == 2) { gap_cost += b2 + (double)mg_log2_host((float)(1.0 + len)); n_gap += len; } }
	n_mis = r.blen + (int32_t)r.n_ambi - r.mlen - n_gap;
	return (int32_t)(match_sc * (r.mlen - b2 * n_mis - gap_cost) + .499);
}
static void update_dp_max(int qlen, std::vector<Reg> &regs, float frac, int a, int b) // align.c:936-960
{
	int32_t max = -1, max2 = -1, max_i = -1;
	const int n_regs = (int)regs.size();
	if (n_regs < 2) return;
	for (int i = 0; i < n_regs; ++i) {
		Reg &r = regs[i];
		if (!r.has_p) continue;
		if (r.dp_max > max) max2 = max, max = r.dp_max, max_i = i;
		else if (r.dp_max > max2) max2 = r.dp_max;
	}
	if (max_i < 0 || max < 0 || max2 < 0) return;
	if (regs[max_i].qe - regs[max_i].qs < (double)qlen * frac) return;
	if (max2 < (double)max * frac) return;
	double div = 1. - event_identity(regs[max_i]);
	if (div < 0.02) div = 0.02;
	double b2 = 0.5 / div;
	if (b2 * a < b) b2 = (double)a / b;
	for (auto &r : regs) { if (!r.has_p) continue; r.dp_max = recal_max_dp(r, b2, a); if (r.dp_max < 0) r.dp_max = 0; }
}

template <class F> static void parallel_for(size_t n, int n_threads, F f)
{
	if (n_threads <= 1 || n < 2) { for (size_t i = 0; i < n; ++i) f(i); return; }
	std::atomic<size_t> next(0);
	std::vector<std::thread> th;
	std::exception_ptr err = nullptr; std::mutex em;
	for (int t = 0; t < n_threads; ++t) th.emplace_back([&] {
		try { for (;;) { size_t i = next.fetch_add(1); if (i >= n) break; f(i); } }
		catch (...) { std::lock_guard<std::mutex> lk(em); if (!err) err = std::current_exception(); }
	});
	for (auto &t : th) t.join();
	if (err) std::rethrow_exception(err);
}

// regions + alignment of the whole batch: chains in, final records out
void align_batch(const SeqSet &S, const mm_mapopt_t &opt, int k, const std::vector<uint64_t> &q_aoff, ChainResult &C, const std::vector<int32_t> &rep_len,
                 std::vector<std::vector<Reg>> &out, int n_threads, Timers *tm, hipStream_t st)
{
	const int n_seq = S.n_seq;
	const double t_align0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
	out.assign((size_t)n_seq, {});
	g_prof = getenv("PGA_VERBOSE") != nullptr;
	Driver D(S, opt, k);
	std::vector<QueryCtx> Q((size_t)n_seq);
	// ---- regions (mm_gen_regs) and plans ----
	parallel_for((size_t)n_seq, n_threads, [&](size_t qi) {
		QueryCtx &q = Q[qi];
		q.qid = (int)qi, q.qlen = (int32_t)S.len[qi], q.rep_len = rep_len[qi]; q.base = (int)S.grp_off[S.grp_of_seq[qi]];
		const int n_u = C.n_u[qi];
		if (q.qlen == 0 || n_u == 0) { q.finished = true; return; }
		const uint64_t b = q_aoff[qi];
		q.a.assign(C.a.begin() + b, C.a.begin() + b + C.n_v[qi]);
		uint32_t hash = !(opt.flag & MM_F_NO_HASH_NAME) ? x31_hash(S.name[qi].c_str()) : 0;
		hash ^= wang_hash((uint32_t)q.qlen) + wang_hash((uint32_t)opt.seed);
		hash = wang_hash(hash);
		std::vector<Reg> regs;
		gen_regs(hash, q.qlen, n_u, C.u.data() + b, q.a.data(), regs);
		for (auto &r : regs) { q.pool.emplace_back(new RegTask()); q.pool.back()->r = r; q.list.push_back(q.pool.back().get()); }
		if (!(opt.flag & MM_F_CIGAR)) return;
		q.n_a = squeeze_a(q.list, q.a);
		for (RegTask *t : q.list) D.plan(q, *t);
	});
	if (getenv("PGA_VERBOSE")) fprintf(stderr, "[pga]   align: regions+plans %.3f s (%d threads)\n", std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_align0, n_threads);
	// ---- rounds ----
	// The queries are dealt into a few SETS that run their rounds concurrently (one host thread, one stream and one
	// device-memory arena each): while the GPU works on one set's problems the host classifies, collects and advances
	// another's, and the short dependent rounds at the end of one set hide behind the bulk of the next.
	if (opt.flag & MM_F_CIGAR) {
		DpParams P{opt.q, opt.e, opt.q2, opt.e2, D.mat[0], D.mat[1], D.mat[24]};
		auto run_rounds = [&](const std::vector<int> &qs, int set_id, int n_threads, hipStream_t st, Timers *tm) {
		const size_t n_q = qs.size();
		size_t n_requested = 0;
		std::list<PinVec<uint32_t>> pools;
		for (int round = 0; round < 100000; ++round) {
			// run what was requested
			{
				std::vector<size_t> poff(n_q + 1, 0);
				for (size_t k = 0; k < n_q; ++k) poff[k + 1] = poff[k] + Q[qs[k]].pending.size();
				const size_t n_pend = poff[n_q];
				std::vector<DpJob> jb(n_pend); std::vector<std::pair<int,int>> owner(n_pend);
				std::vector<double> cells_of(n_q, 0.0);
				parallel_for(n_q, n_threads, [&](size_t k) {
					const size_t qi = (size_t)qs[k];
					QueryCtx &q = Q[qi];
					size_t o = poff[k]; double cells = 0;
					for (int id : q.pending) { jb[o] = q.jobs[id]; owner[o] = std::make_pair((int)qi, id); cells += (double)q.jobs[id].qlen * q.jobs[id].tlen; ++o; }
					q.pending.clear();
					cells_of[k] = cells;
				});
				n_requested = jb.size();
				if (!jb.empty()) {
					std::vector<DpRes> rs;
					pools.emplace_back();
					PinVec<uint32_t> &cg = pools.back();          // stays alive until the batch is done: results point into it
					double t_dp = getenv("PGA_VERBOSE") ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0;
					dp_run(S.d_nt4.p, jb, P, rs, cg, st, tm);
					if (t_dp > 0) fprintf(stderr, "[pga]   set %d round %d: %zu DP problems in %.3f s\n", set_id, round, jb.size(), std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_dp);
					if (tm) { tm->dp_jobs += (double)jb.size(); for (double c : cells_of) tm->dp_cells += c; }
					const uint32_t *base = cg.data();
					parallel_for((rs.size() + 65535) / 65536, n_threads, [&](size_t blk) {
						const size_t lo = blk * 65536, hi = std::min(rs.size(), lo + 65536);
						for (size_t i = lo; i < hi; ++i) {
							QueryCtx &q = Q[owner[i].first]; const int id = owner[i].second;
							if (rs[i].n_cigar < 0) throw std::runtime_error("pga: DP backtrack did not terminate");
							q.res[id] = rs[i]; q.res[id].pad = 1;
							q.cig[id] = base + rs[i].cigar_off;
						}
					});
				}
			}
			const double t_adv0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
			std::atomic<int> unfinished(0);
			parallel_for(n_q, n_threads, [&](size_t k) {
				const size_t qi = (size_t)qs[k];
				QueryCtx &q = Q[qi];
				if (q.finished) return;
				// walk the list in the reference's order (align.c:981-1010); stop at the first region that must wait
				// Regions are independent, so one that waits for a DP result does not hold up the others; only the
				// inversion test looks at the previous list element and is deferred while anything before it is open.
				bool waiting = false;
				for (size_t i = 0; i < q.list.size(); ++i) {
					RegTask &T = *q.list[i];
					if (T.is_inv) continue;
					if (!T.planned) { ScopeNs sc(3); D.plan(q, T); }
					if (!T.done) {
						Reg r2;
						if (!D.advance(q, T, r2)) { waiting = true; continue; }
						if (r2.cnt > 0) { q.pool.emplace_back(new RegTask()); q.pool.back()->r = r2; q.list.insert(q.list.begin() + i + 1, q.pool.back().get()); }
					}
					if (i > 0 && T.r.split_inv && !(opt.flag & MM_F_NO_INV) && T.inv_state != 2) {
						if (waiting) continue;   // an earlier element is still open: decide next round
						Reg r_inv;
						int rc = D.inversion(q, T, q.list[i - 1]->r, r_inv);
						if (rc == 1) { waiting = true; continue; }
						if (rc == 2) {
							q.pool.emplace_back(new RegTask()); RegTask *ti = q.pool.back().get();
							ti->r = r_inv; ti->is_inv = ti->planned = ti->done = true;
							q.list.insert(q.list.begin() + i + 1, ti);
						}
					}
				}
				if (waiting) { ++unfinished; return; }
				ScopeNs sc_fin(4);
				// ---- all regions aligned: filters, ranking, mapq (align.c:1013-1021, map.c:340-341) ----
				std::vector<Reg> regs; regs.reserve(q.list.size());
				for (RegTask *t : q.list) regs.push_back(std::move(t->r));
				filter_regs(opt, q.qlen, regs);
				if (q.qlen >= opt.rank_min_len) { update_dp_max(q.qlen, regs, opt.rank_frac, opt.a, opt.b); filter_regs(opt, q.qlen, regs); }
				hit_sort(regs);
				set_mapq(regs, opt.min_chain_score, opt.a, q.rep_len);
				out[qi] = std::move(regs);
				q.finished = true; q.pool.clear(); q.list.clear(); q.a.clear(); q.a.shrink_to_fit();
			});
			if (g_prof) { fprintf(stderr, "[pga]   host phases (thread-summed s): segments %.3f, fetch %.3f, update_extra %.3f, plan %.3f, finish %.3f; z-drop test skipped %lld, run %lld (%.3f s)\n", g_ns[0] * 1e-9, g_ns[1] * 1e-9, g_ns[2] * 1e-9, g_ns[3] * 1e-9, g_ns[4] * 1e-9, (long long)g_ns[5], (long long)g_ns[6], g_ns[7] * 1e-9); for (auto &x : g_ns) x = 0; }
			if (getenv("PGA_VERBOSE")) fprintf(stderr, "[pga]   set %d round %d: host advance %.3f s\n", set_id, round, std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_adv0);
			if (unfinished.load() == 0) break;
			bool any_pending = false; for (int qi : qs) any_pending |= !Q[qi].pending.empty();
			if (!any_pending) throw std::runtime_error("pga: alignment driver stalled");
			(void)n_requested;
		}
		};
		// deal the queries by anchor count (largest first, round robin): balanced sets
		// (two sets pay from a few hundred queries on; below that the few long problems of a set only get in each other's way)
		int n_sets = getenv("PGA_ALIGN_SETS") ? atoi(getenv("PGA_ALIGN_SETS")) : (n_seq >= 256 ? 2 : 1);
		if (n_sets < 1) n_sets = 1;
		if (n_seq < 8 * n_sets) n_sets = 1;
		std::vector<int> order((size_t)n_seq);
		for (int i = 0; i < n_seq; ++i) order[i] = i;
		std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return Q[x].n_a > Q[y].n_a; });
		std::vector<std::vector<int>> sets((size_t)n_sets);
		for (int i = 0; i < n_seq; ++i) sets[(size_t)(i % n_sets)].push_back(order[i]);
		for (auto &v : sets) std::sort(v.begin(), v.end());
		if (n_sets == 1) run_rounds(sets[0], 0, n_threads, st, tm);
		else {
			int dev = 0; PGA_HIP(hipGetDevice(&dev));
			std::vector<Timers> tms((size_t)n_sets);
			std::vector<std::string> errs((size_t)n_sets);
			std::vector<std::thread> th;
			for (int k = 0; k < n_sets; ++k) th.emplace_back([&, k] {
				hipStream_t ss = nullptr;
				const int arena = dev_lease_arena();      // the set's own stream gets its own arena
				ArenaScope arena_scope(arena);
				try {
					PGA_HIP(hipSetDevice(dev));
					set_thread_budget(std::max(1, n_threads / n_sets));
					PGA_HIP(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
					run_rounds(sets[(size_t)k], k, std::max(1, n_threads / n_sets), ss, tm ? &tms[(size_t)k] : nullptr);
				} catch (std::exception &e) { errs[(size_t)k] = e.what(); if (errs[(size_t)k].empty()) errs[(size_t)k] = "unknown error"; }
				if (ss) { (void)hipStreamSynchronize(ss); (void)hipStreamDestroy(ss); }
				(void)hipDeviceSynchronize();            // the DP lane streams of the set have drained too: the arena's blocks are reusable
				dev_release_arena(arena);
			});
			for (auto &t : th) t.join();
			for (auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
			if (tm) for (const Timers &t : tms) {
				tm->dp_jobs += t.dp_jobs; tm->dp_cells += t.dp_cells; tm->dp_bases += t.dp_bases; tm->dp_cigar_ops += t.dp_cigar_ops;
				for (int i = 0; i < K_COUNT; ++i) { tm->kern[i].ms += t.kern[i].ms; tm->kern[i].launches += t.kern[i].launches; tm->kern[i].alg_bytes += t.kern[i].alg_bytes; tm->kern[i].cells += t.kern[i].cells; }
			}
		}

	} else {
		for (int qi = 0; qi < n_seq; ++qi) {
			QueryCtx &q = Q[qi];
			if (q.finished) continue;
			std::vector<Reg> regs; for (RegTask *t : q.list) regs.push_back(std::move(t->r));
			set_mapq(regs, opt.min_chain_score, opt.a, q.rep_len);
			out[qi] = std::move(regs);
		}
	}
}

} // namespace pga
