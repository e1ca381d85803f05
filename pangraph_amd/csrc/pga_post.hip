// pga_post.hip -- everything of the base-level alignment stage that READS BASES outside the DP kernels, as batched device kernels:
// the host driver (pga_align.cpp) keeps control flow over compact records and never touches a base.
//
//   k_seg_identity   first-pass gap fills whose windows are equally long: mismatch count with early exit, deciding whether the main
//                    diagonal is provably the unique optimum (proof: pga_ksw_fast.hip) -- replaces the per-segment DP call
//                    mm_align_pair (align.c:316-344) for two thirds of the segments (nearly all of them between close relatives)
//   k_zdrop_walk     mm_test_zdrop's walk along a CIGAR (align.c:32-45,47-77): the worst score drop and its window
//   k_cigar_finish   mm_fix_cigar + mm_update_extra (align.c:91-167,240-289) of a finished region: one WAVE per region; CIGAR
//                    operations are consumed in order, the bases under an operation 64 at a time across the lanes
//
// Numerics of k_cigar_finish: the reference accumulates s (double) base by base with a clamp at 0 and tracks its maximum.  Inside a
// match run the increments are integers and every partial sum is an integer plus the fractional bits of earlier gap terms
// (e * mg_log2: a float times an int) -- far inside a double's mantissa for any sequence length, so double addition is EXACT here and
// the 64-base block form below (prefix sums and prefix minima in int32, s_i = P_i - min(-s_in, min_j<=i P_j)) returns bit-identical
// values to the stepwise loop.  Gap terms are evaluated exactly as the reference writes them (float mg_log2, double product, no FMA).
#include "pga_common.h"
#include "pga_post.h"
#include "pga_wave.h"

namespace pga {

__device__ __forceinline__ int post_tbase(PkBases bases, uint64_t t_off, int i) { return bases.at(t_off + (uint64_t)i); }
// base j of the aligned query strand, window starting at q_start on that strand (reverse strand = complement read backwards, align.c:970-975)
__device__ __forceinline__ int post_qbase(PkBases bases, uint64_t q_off, int qlen_full, int q_start, int q_rev, int j)
{
	const int pj = q_start + j;
	if (!q_rev) return bases.at(q_off + (uint64_t)pj);
	const int c = bases.at(q_off + (uint64_t)(qlen_full - 1 - pj));
	return c < 4 ? 3 - c : 4;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_prefix_sum_incl(v), 63); }

// ------------------------------------------------------------------------------------------------ identity probes
// One WAVE per probe at a time (grid-stride): the lanes read 64 consecutive bases of both windows per step (coalesced; a lane per
// probe would pull a whole cache line for every byte), the mismatch count is a ballot + popcount, the early exit is uniform.
__global__ __launch_bounds__(256)
void k_seg_identity(const PostProbe *__restrict__ pr, uint32_t n, PkBases bases, int m_max, int32_t *__restrict__ out)
{
	const int lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
	for (uint32_t i = wave; i < n; i += n_waves) {
		const PostProbe P = pr[i];
		const uint64_t t = P.t_off;
		const uint64_t q = P.q_rev ? P.q_off + (uint64_t)(P.qlen_full - 1 - P.qs) : P.q_off + (uint64_t)P.qs;
		int m = 0;
		for (int b = 0; b < P.n; b += 64) {
			const int k = b + lane;
			bool bad = false, diff = false;
			if (k < P.n) {
				const int x = bases.at(t + (uint64_t)k), y = bases.at(P.q_rev ? q - (uint64_t)k : q + (uint64_t)k);
				bad = (x | y) > 3;
				diff = P.q_rev ? x != 3 - y : x != y;
			}
			if (__ballot(bad)) { m = -1; break; }
			m += __popcll(__ballot(diff));
			if (m > m_max) { m = -1; break; }
		}
		if (lane == 0) out[i] = m;
	}
}

void post_identity(PkBases d_bases, const PinVec<PostProbe> &probes, int m_max, PinVec<int32_t> &out, hipStream_t st)
{
	const size_t n = probes.size();
	out.resize(n);
	if (!n) return;
	if (n <= 4096) {                                              // (few probes: the kernel takes them from, and answers into, the caller's pinned lists)
		hipLaunchKernelGGL(k_seg_identity, dim3((unsigned)std::min<size_t>((n + 3) / 4, 256 * 32)), dim3(256), 0, st, probes.data(), (uint32_t)n, d_bases, m_max, out.data());
		PGA_HIP(hipGetLastError());
		PGA_HIP(sync_stream(st));
		return;
	}
	DBuf<PostProbe> d; d.alloc(n);
	DBuf<int32_t> r; r.alloc(n);
	PGA_HIP(hipMemcpyAsync(d.p, probes.data(), n * sizeof(PostProbe), hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_seg_identity, dim3((unsigned)std::min<size_t>((n + 3) / 4, 256 * 32)), dim3(256), 0, st, d.p, (uint32_t)n, d_bases, m_max, r.p);
	PGA_HIP(hipGetLastError());
	PGA_HIP(hipMemcpyAsync(out.data(), r.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
}

// ------------------------------------------------------------------------------------------------ z-drop walk
// One thread per request.  Scores: match run positions add mat[t][q]; a gap of any kind subtracts q + e*len (align.c:64-72); after
// every step the tracker compares with the best prefix so far, discounting the diagonal offset at e per base (align.c:32-45).
__global__ __launch_bounds__(64)
void k_zdrop_walk(const PostWalk *__restrict__ rq, uint32_t n, const uint32_t *__restrict__ cig, PkBases bases,
                  int sc_mch, int sc_mis, int sc_ambi, int gap_q, int gap_e, PostWalkRes *__restrict__ out)
{
	const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= n) return;
	const PostWalk W = rq[id];
	const uint32_t *cg = cig + W.cig_off;
	int score = 0, best = INT32_MIN, best_i = -1, best_j = -1, ti = 0, qj = 0, worst = 0;
	int w_t0 = -1, w_t1 = -1, w_q0 = -1, w_q1 = -1;
	auto track = [&](int ci, int cj) {
		if (score < best) {
			const int li = ci - best_i, lj = cj - best_j, off = li > lj ? li - lj : lj - li, z = best - score - off * gap_e;
			if (z > worst) { worst = z; w_t0 = best_i, w_t1 = ci, w_q0 = best_j, w_q1 = cj; }
		} else best = score, best_i = ci, best_j = cj;
	};
	for (uint32_t k = 0; k < W.n_cigar; ++k) {
		const uint32_t op = cg[k] & 0xf; const int len = (int)(cg[k] >> 4);
		if (op == 0) {
			for (int l = 0; l < len; ++l) {
				const int tb = post_tbase(bases, W.t_off, ti + l), qb = post_qbase(bases, W.q_off, W.qlen_full, W.qs, W.q_rev, qj + l);
				score += (tb > 3 || qb > 3) ? sc_ambi : tb == qb ? sc_mch : sc_mis;
				track(ti + l, qj + l);
			}
			ti += len, qj += len;
		} else if (op == 1 || op == 2 || op == 3) {
			score -= gap_q + gap_e * len;
			if (op == 1) qj += len; else ti += len;
			track(ti, qj);
		}
	}
	PostWalkRes R; R.max_zdrop = worst; R.t0 = w_t0, R.t1 = w_t1, R.q0 = w_q0, R.q1 = w_q1;
	out[id] = R;
}

void post_zdrop_walk(PkBases d_bases, const std::vector<PostWalk> &reqs, const std::vector<uint32_t> &cig, const DpParams &P, std::vector<PostWalkRes> &out, hipStream_t st)
{
	const size_t n = reqs.size();
	out.resize(n);
	if (!n) return;
	DBuf<uint32_t> c; c.alloc(cig.size() ? cig.size() : 1);
	if (!cig.empty()) PGA_HIP(hipMemcpyAsync(c.p, cig.data(), cig.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
	if (n <= 4096) {                                              // (requests and answers through pinned memory; the operation lists are read many times: a device copy)
		PinVec<PostWalk> hd; hd.resize(n); memcpy(hd.data(), reqs.data(), n * sizeof(PostWalk));
		PinVec<PostWalkRes> hr; hr.resize(n);
		hipLaunchKernelGGL(k_zdrop_walk, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, hd.data(), (uint32_t)n, c.p, d_bases, P.sc_mch, P.sc_mis, P.sc_ambi, P.q, P.e, hr.data());
		PGA_HIP(hipGetLastError());
		PGA_HIP(sync_stream(st));
		memcpy(out.data(), hr.data(), n * sizeof(PostWalkRes));
		return;
	}
	DBuf<PostWalk> d; d.upload(reqs, st);
	DBuf<PostWalkRes> r; r.alloc(n);
	hipLaunchKernelGGL(k_zdrop_walk, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d.p, (uint32_t)n, c.p, d_bases, P.sc_mch, P.sc_mis, P.sc_ambi, P.q, P.e, r.p);
	PGA_HIP(hipGetLastError());
	PGA_HIP(hipMemcpyAsync(out.data(), r.p, n * sizeof(PostWalkRes), hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
}

// ------------------------------------------------------------------------------------------------ CIGAR finish
__device__ __forceinline__ float post_log2(float x) // mmpriv.h:118-126
{
	union { float f; uint32_t i; } z = { x };
	float r = (float)(((z.i >> 23) & 255) - 128);   // unsigned arithmetic as in the reference (x >= 2 here: gaps are at least one base long)
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	r += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return r;
}

__device__ __forceinline__ int32_t wave_prefix_min_incl(int32_t v)
{
	v = dpp_min_step_i<0x111, 0xf>(v); v = dpp_min_step_i<0x112, 0xf>(v); v = dpp_min_step_i<0x114, 0xf>(v); v = dpp_min_step_i<0x118, 0xf>(v);
	v = dpp_min_step_i<0x142, 0xa>(v); v = dpp_min_step_i<0x143, 0xc>(v);
	return v;
}

#define FIN_LDS_OPS 6144     // regions with up to this many operations are edited in LDS (24 KB), longer ones in place in HBM

// One wave per region.  Stage 1 (mm_fix_cigar): indels between two match runs slide left as far as the bases allow (the lanes test
// 64 positions at a time), mixed I/D stretches are merged, empty operations dropped, a leading indel is cut off and reported as a
// shift of the region start.  Stage 2 (mm_update_extra): blen / mlen / n_ambi and the clamped running score.
__global__ __launch_bounds__(64)
void k_cigar_finish(const PostFin *__restrict__ rq, uint32_t n, uint32_t *__restrict__ cig_all, PkBases bases,
                    int sc_mch, int sc_mis, int sc_ambi, int gap_q, int gap_e, PostFinRes *__restrict__ out)
{
	__shared__ uint32_t s_ops[FIN_LDS_OPS];
	const int lane = threadIdx.x;
	for (uint32_t id = blockIdx.x; id < n; id += gridDim.x) {
		const PostFin F = rq[id];
		uint32_t *g_ops = cig_all + F.cig_off;
		uint32_t nc = F.n_cigar;
		const bool in_lds = nc <= FIN_LDS_OPS;
		uint32_t *cg = in_lds ? s_ops : g_ops;
		if (in_lds) { for (uint32_t k = lane; k < nc; k += 64) s_ops[k] = g_ops[k]; }
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		int qshift = 0, tshift = 0;
		if (nc > 1) {
			// ---- left-alignment of indels (align.c:100-117) ----
			int toff = 0, qoff = 0; bool squeeze = false;
			for (uint32_t k = 0; k < nc; ++k) {
				const uint32_t c = cg[k], op = c & 0xf; const int len = (int)(c >> 4);
				if (len == 0) squeeze = true;
				if (op == 0) toff += len, qoff += len;
				else if (op == 1 || op == 2) {
					if (k > 0 && k + 1 < nc && (cg[k - 1] & 0xf) == 0 && (cg[k + 1] & 0xf) == 0) {
						const int room = (int)(cg[k - 1] >> 4), o = op == 1 ? qoff : toff;
						int slid = 0;
						for (int b = 0; b < room; b += 64) {
							const int l = b + lane;
							bool same = false;
							if (l < room) {
								const int x = op == 1 ? post_qbase(bases, F.q_off, F.qlen_full, F.q_start, F.q_rev, o - 1 - l) : post_tbase(bases, F.t_off, o - 1 - l);
								const int y = op == 1 ? post_qbase(bases, F.q_off, F.qlen_full, F.q_start, F.q_rev, o + len - 1 - l) : post_tbase(bases, F.t_off, o + len - 1 - l);
								same = x == y;
							}
							const unsigned long long eq = __ballot(same);
							const int run = eq == ~0ULL ? 64 : __builtin_ctzll(~eq);
							slid += run;
							if (run < 64) break;
						}
						if (slid > room) slid = room;
						if (slid > 0) {
							if (lane == 0) { cg[k - 1] -= (uint32_t)slid << 4; cg[k + 1] += (uint32_t)slid << 4; }
							__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
							qoff -= slid, toff -= slid;
						}
						if (slid == room) squeeze = true;
					}
					if (op == 1) qoff += len; else toff += len;
				} else if (op == 3) toff += len;
			}
			// ---- the remaining passes edit the operation list only: lane 0 ----
			if (lane == 0) {
				// mixed insertion/deletion stretches become one I and one D (align.c:118-133)
				for (uint32_t k = 0; k + 2 < nc; ++k) {
					const uint32_t a = cg[k] & 0xf;
					if (a > 0 && a + (cg[k + 1] & 0xf) == 3) {
						uint32_t sum_i = 0, sum_d = 0, e = k;
						for (; e < nc; ++e) {
							const uint32_t op = cg[e] & 0xf, ln = cg[e] >> 4;
							if (op == 1) sum_i += ln; else if (op == 2) sum_d += ln; else if (ln != 0) break;
						}
						if (sum_i > 0 && sum_d > 0 && e - k > 2) {
							cg[k] = sum_i << 4 | 1; cg[k + 1] = sum_d << 4 | 2;
							for (uint32_t z = k + 2; z < e; ++z) cg[z] &= 0xf;
							squeeze = true;
						}
						k = e;
					}
				}
				if (squeeze) {                          // drop empty operations, then fuse neighbours of one kind (align.c:134-146)
					uint32_t w = 0;
					for (uint32_t k = 0; k < nc; ++k) if (cg[k] >> 4) cg[w++] = cg[k];
					nc = w; w = 0;
					for (uint32_t k = 0; k < nc; ++k) {
						if (k + 1 == nc || (cg[k] & 0xf) != (cg[k + 1] & 0xf)) cg[w++] = cg[k];
						else cg[k + 1] += cg[k] >> 4 << 4;
					}
					nc = w;
				}
			}
			nc = (uint32_t)__builtin_amdgcn_readfirstlane((int)nc);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
		uint32_t first = 0;                             // a leading indel leaves the record (align.c:147-166; lists of one operation are left alone, :96)
		if (F.n_cigar > 1 && nc > 0 && ((cg[0] & 0xf) == 1 || (cg[0] & 0xf) == 2)) {
			if ((cg[0] & 0xf) == 1) qshift = (int)(cg[0] >> 4); else tshift = (int)(cg[0] >> 4);
			first = 1;
		}
		// ---- blen / mlen / n_ambi / dp_max over the final list (align.c:249-285) ----
		const int q0 = F.q_start + qshift; const uint64_t t0 = F.t_off + (uint64_t)tshift;
		int toff = 0, qoff = 0, blen = 0, mlen = 0, n_ambi = 0, n_gapo = 0, n_gap = 0;
		double s = 0.0, smax = 0.0;
		for (uint32_t k = first; k < nc; ++k) {
			const uint32_t c = cg[k], op = c & 0xf; const int len = (int)(c >> 4);
			if (op == 0) {
				int ambi = 0, diff = 0;
				// 1024 bases per trip, sixteen CONSECUTIVE bases per lane, straight from the packed store: two 32-bit windows (target, query --
				// the reverse strand is the window read backwards and complemented) XORed give the sixteen comparisons of a lane at once; four
				// word loads per lane and side cover what sixty-four byte loads did.  The clamped running score in its block form (see the
				// header): a lane sums its sixteen scores and takes their prefix minimum, one prefix sum and one prefix minimum across the lanes
				// place them, a second pass over the sixteen positions evaluates s at every position (int32 sums, the reference's doubles).
				for (int b = 0; b < len; b += 1024) {
					const int j0 = b + 16 * lane;
					const int v = len - j0 <= 0 ? 0 : len - j0 >= 16 ? 16 : len - j0;
					const int n_tot = len - b < 1024 ? len - b : 1024;
					uint32_t tw = 0, tm = 0, qw = 0, qm = 0;
					if (v > 0) {
						bases.window16(t0 + (uint64_t)(toff + j0), tw, tm);
						const int pj = q0 + qoff + j0;
						if (!F.q_rev) bases.window16(F.q_off + (uint64_t)pj, qw, qm);
						else {
							const int64_t hi = (int64_t)F.qlen_full - 1 - pj;           // the base of j0; the next fifteen lie below it
							if (hi >= 15 || F.q_off >= 16) {
								uint32_t w, m;
								bases.window16(F.q_off + (uint64_t)(hi - 15), w, m);
								w = __brev(w); w = ((w >> 1) & 0x55555555u) | ((w & 0x55555555u) << 1);     // sixteen 2-bit groups in reverse order
								qw = ~w; qm = __brev(m) >> 16;
							} else {
								for (int i = 0; i < v; ++i) { const int c = post_qbase(bases, F.q_off, F.qlen_full, q0, F.q_rev, qoff + j0 + i); if (c > 3) qm |= 1u << i; else qw |= (uint32_t)c << (2 * i); }
							}
						}
					}
					const uint32_t vm16 = v >= 16 ? 0xffffu : (1u << v) - 1u, vm32 = v >= 16 ? ~0u : (1u << (2 * v)) - 1u;
					uint32_t amb = (tm | qm) & vm16;
					uint32_t sp = amb; sp = (sp | sp << 8) & 0x00ff00ffu; sp = (sp | sp << 4) & 0x0f0f0f0fu; sp = (sp | sp << 2) & 0x33333333u; sp = (sp | sp << 1) & 0x55555555u;
					uint32_t d2 = tw ^ qw; d2 = (d2 | d2 >> 1) & 0x55555555u & vm32 & ~sp;        // bit 2i: base i differs (and neither side is ambiguous)
					const int n_a = (int)wave_sum_u32((uint32_t)__popc(amb)), n_d = (int)wave_sum_u32((uint32_t)__popc(d2));
					ambi += n_a, diff += n_d;
					if (n_a == 0 && n_d == 0 && sc_mch > 0) { s += (double)sc_mch * (double)n_tot; smax = smax > s ? smax : s; continue; }   // all matches: s only grows
					// The trip in block form (round 6).  With s_in the score before the trip, P_i the sum of the first i scores and pmin_i = min(P_1 .. P_i):
					// s_i = max(s_in + P_i, P_i - pmin_i), hence the trip's maximum is max(s_in + A, B) with A = max_i P_i, B = max_i (P_i - pmin_i), and it leaves
					// s_out = max(s_in + T, T - PM) (T = P_n, PM = pmin_n) -- four integers per trip that do not depend on s_in.  A match scores sc_mch > 0, so between
					// two EVENTS (a mismatch or an ambiguous base) P_i rises: minima of P can only sit right behind an event or at the first position, maxima of P and
					// of P - pmin right in front of an event, at one, or at the end -- a lane visits those positions only (one or two of its sixteen bases at 1 %
					// divergence) instead of all sixteen twice.  sc_mch <= 0 (no preset has it) takes the position-by-position form.
					int T = 0, Lm = 0x3fffffff;
					const uint32_t dif16 = (d2 & 0x55555555u);                            // bit 2i -> compress to bit i
					uint32_t dd = dif16; dd = (dd | dd >> 1) & 0x33333333u; dd = (dd | dd >> 2) & 0x0f0f0f0fu; dd = (dd | dd >> 4) & 0x00ff00ffu; dd = (dd | dd >> 8) & 0x0000ffffu;
					const uint32_t ev16 = (amb | dd) & vm16;
					double mx; 
					if (sc_mch > 0) {
						const int d_mis = sc_mis - sc_mch, d_amb = sc_ambi - sc_mch;
						T = v * sc_mch + __popc(dd & ~amb) * d_mis + __popc(amb) * d_amb;
						{	// pass 1: the lane's own prefix minimum (of sums that start at its first base)
							int cum = 0; uint32_t E = ev16;
							if (v > 0 && !(E & 1u)) Lm = sc_mch;
							while (E) {
								const int pz = __ffs((int)E) - 1; E &= E - 1;
								cum += (amb >> pz & 1u) ? d_amb : d_mis;
								const int part = (pz + 1) * sc_mch + cum;
								Lm = Lm < part ? Lm : part;
							}
						}
						const int Bx = (int)wave_prefix_sum_incl((uint32_t)T) - T;
						const int Gi = wave_prefix_min_incl(v > 0 ? Bx + Lm : 0x3fffffff);
						const int Ge = wave_shr1(Gi, 0x3fffffff);
						int A_l = -0x3fffffff, B_l = -0x3fffffff;
						if (v > 0) {	// pass 2: maxima of P and of P - pmin at the candidate positions
							int cum = 0, pmin = Ge; uint32_t E = ev16;
							auto at = [&](int i_) { const int run = Bx + i_ * sc_mch + cum; pmin = pmin < run ? pmin : run; A_l = A_l > run ? A_l : run; const int d_ = run - pmin; B_l = B_l > d_ ? B_l : d_; };
							if (!(E & 1u)) at(1);
							while (E) {
								const int pz = __ffs((int)E) - 1; E &= E - 1;
								if (pz >= 1) at(pz);                                           // right in front of the event
								cum += (amb >> pz & 1u) ? d_amb : d_mis;
								at(pz + 1);                                                    // ... and at it
							}
							at(v);
						}
						const int A = wave_max_i32(A_l), B = wave_max_i32(B_l);
						const int ll = (n_tot - 1) >> 4;                                 // the lane of the trip's last base
						const int Ttot = rl(Bx + T, ll), PM = rl(Gi, ll);
						const double c1 = s + (double)A, c2 = (double)B;
						mx = c1 > c2 ? c1 : c2;
						const double o1 = s + (double)Ttot, o2 = (double)(Ttot - PM);
						s = o1 > o2 ? o1 : o2;
					} else {
					for (int i = 0; i < v; ++i) { const int x = (amb >> i & 1u) ? sc_ambi : (d2 >> (2 * i) & 1u) ? sc_mis : sc_mch; T += x; Lm = Lm < T ? Lm : T; }
					const int Bx = (int)wave_prefix_sum_incl((uint32_t)T) - T;          // sum of the scores in front of this lane
					const int Gi = wave_prefix_min_incl(v > 0 ? Bx + Lm : 0x3fffffff);    // prefix minimum up to and including this lane
					const int Ge = wave_shr1(Gi, 0x3fffffff);                            // ... of the lanes in front
					const double neg_in = -s;
					double lmax = 0.0, slast = 0.0;
					{
						int run = Bx, pmin = Ge;
						for (int i = 0; i < v; ++i) {
							const int x = (amb >> i & 1u) ? sc_ambi : (d2 >> (2 * i) & 1u) ? sc_mis : sc_mch;
							run += x; pmin = pmin < run ? pmin : run;
							const double pm = (double)pmin, floor_ = neg_in < pm ? neg_in : pm;
							const double si = (double)run - floor_;
							lmax = lmax > si ? lmax : si; slast = si;
						}
					}
					mx = -wave_min_f64_key(-(v > 0 ? lmax : 0.0));
					const int ll = (n_tot - 1) >> 4;                                     // the lane of the trip's last base
					const long long bits = __double_as_longlong(slast);
					const int lo = rl((int)(bits & 0xffffffffLL), ll), hi2 = rl((int)(bits >> 32), ll);
					s = __longlong_as_double(((long long)hi2 << 32) | (unsigned)lo);
					}
					smax = smax > mx ? smax : mx;
				}
				blen += len - ambi, mlen += len - (ambi + diff), n_ambi += ambi;
				toff += len, qoff += len;
			} else if (op == 1 || op == 2) {
				int ambi = 0;
				for (int b = 0; b < len; b += 64) {
					const int l = b + lane;
					const bool a = l < len && (op == 1 ? post_qbase(bases, F.q_off, F.qlen_full, q0, F.q_rev, qoff + l) : post_tbase(bases, t0, toff + l)) > 3;
					ambi += __popcll(__ballot(a));
				}
				blen += len - ambi, n_ambi += ambi; ++n_gapo, n_gap += len;
				s -= (double)gap_q + (double)gap_e * (double)post_log2((float)(1.0 + (double)len));
				if (s < 0) s = 0;
				if (op == 1) qoff += len; else toff += len;
			} else if (op == 3) toff += len;
		}
		if (in_lds) { for (uint32_t k = first + lane; k < nc; k += 64) g_ops[k - first] = s_ops[k]; }
		else if (first) {                               // in place in HBM: close the gap left by the leading indel
			for (uint32_t b = first; b < nc; b += 64) {
				const uint32_t k = b + lane; const uint32_t v = k < nc ? g_ops[k] : 0;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				if (k < nc) g_ops[k - first] = v;
			}
		}
		if (lane == 0) {
			PostFinRes R; R.n_cigar = nc - first; R.qshift = qshift; R.tshift = tshift; R.blen = blen; R.mlen = mlen; R.n_ambi = n_ambi;
			R.dp_max = (int32_t)(smax + .499); R.n_gapo = n_gapo; R.n_gap = n_gap; R.q_span = qoff; R.t_span = toff;
			out[id] = R;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
}

void post_cigar_finish(PkBases d_bases, const std::vector<PostFin> &reqs, PinVec<uint32_t> &cig, const DpParams &P, std::vector<PostFinRes> &out, hipStream_t st)
{
	const size_t n = reqs.size();
	out.resize(n);
	if (!n) return;
	DBuf<uint32_t> c; c.alloc(cig.size() ? cig.size() : 1);
	if (cig.size()) PGA_HIP(hipMemcpyAsync(c.p, cig.data(), cig.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
	const unsigned grid = (unsigned)std::min<size_t>(n, 256 * 32);
	if (n <= 4096) {                                              // (requests and records through pinned memory)
		PinVec<PostFin> hd; hd.resize(n); memcpy(hd.data(), reqs.data(), n * sizeof(PostFin));
		PinVec<PostFinRes> hr; hr.resize(n);
		hipLaunchKernelGGL(k_cigar_finish, dim3(grid), dim3(64), 0, st, hd.data(), (uint32_t)n, c.p, d_bases, P.sc_mch, P.sc_mis, P.sc_ambi, P.q, P.e, hr.data());
		PGA_HIP(hipGetLastError());
		if (cig.size()) PGA_HIP(hipMemcpyAsync(cig.data(), c.p, cig.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
		PGA_HIP(sync_stream(st));
		memcpy(out.data(), hr.data(), n * sizeof(PostFinRes));
		return;
	}
	DBuf<PostFin> d; d.upload(reqs, st);
	DBuf<PostFinRes> r; r.alloc(n);
	hipLaunchKernelGGL(k_cigar_finish, dim3(grid), dim3(64), 0, st, d.p, (uint32_t)n, c.p, d_bases, P.sc_mch, P.sc_mis, P.sc_ambi, P.q, P.e, r.p);
	PGA_HIP(hipGetLastError());
	PGA_HIP(hipMemcpyAsync(out.data(), r.p, n * sizeof(PostFinRes), hipMemcpyDeviceToHost, st));
	if (cig.size()) PGA_HIP(hipMemcpyAsync(cig.data(), c.p, cig.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	PGA_HIP(sync_stream(st));
}

// bases of one window, back on the host (only the rare local-alignment windows the LL kernel does not take need them)


} // namespace pga
