/* pga_align.h -- the drop-in boundary, part 2: the native batch entry of libpgalign.so.
 *
 * One call aligns G independent all-vs-all GROUPS (one group = the block set of one `find_matches` call,
 * reference: packages/pangraph/src/pangraph/graph_merging.rs:176-185 ->
 * packages/pangraph/src/align/minimap2_lib/align_with_minimap2_lib.rs:15-85) and returns, per group, exactly the
 * records `align_with_minimap2_lib` would build from minimap2's output (alignment.rs:13-57), in query order
 * (ascending sequence index inside the group, then minimap2's own order inside a query).
 * Plain pointers and sizes only; results are owned by the library until pga_result_free().
 *
 * Errors: negative return + pga_last_error() (thread-local string).  The library needs a gfx950 device; it has
 * no CPU fallback.
 */
#ifndef PGA_ALIGN_H
#define PGA_ALIGN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	int32_t sensitivity;         /* 5 | 10 | 20  -> minimap2 preset asm5/asm10/asm20 (align_with_minimap2_lib.rs:35-40) */
	int32_t kmer_length;         /* <=0: preset default (-K, alignment_args.rs:33-35)                                  */
	int32_t indel_len_threshold; /* -l; minimap2 -s = max(len-10, 5) (align_with_minimap2_lib.rs:47)                  */
	int32_t n_threads;           /* host threads for CIGAR post-processing; <=0: all cores                           */
} pga_params_t;

/* One record == one `Alignment` (alignment.rs:40-57). */
typedef struct {
	int32_t group;               /* index of the group the record belongs to */
	int32_t qry, ref;            /* sequence indices inside the group (-> Hit.name) */
	int32_t qry_len, qry_start, qry_end;
	int32_t ref_len, ref_start, ref_end;
	int32_t matches, length, quality;  /* mlen, blen, mapq */
	int32_t reverse;             /* 0 forward, 1 reverse (orientation) */
	int32_t align;               /* AS = dp_score */
	int32_t n_ambi, inv;
	double divergence;           /* de = 1 - mm_event_identity (packages/minimap2/src/map.rs:320-325) */
	uint64_t cigar_off;          /* first op in the CIGAR pool */
	uint32_t n_cigar;            /* ops are len<<4|op, op in "MIDNSHP=XB" */
	uint32_t pad;
} pga_match_t;

typedef struct pga_result_s pga_result_t;

typedef struct {                 /* stage wall times (s) and work counters of a call */
	double upload, sketch, index, seed, chain, align, total;
	double n_bases, n_minimizers, n_anchors, n_dp_jobs, n_dp_cells, n_matches, n_dp_bases;
	/* the path's own kernels, device time from HIP events on the launch stream (streams overlap: the times do not add up to `total`):
	 * [0] k_sketch_tiles  [1] k_chain_fast (+k_chain_segments)  [2] k_bt_list + k_bt_walk  [3] k_extd2_fast (register tiles)
	 * [4] k_extd2_wide<256>  [5] k_ll_i16  [6] k_rs_init + k_rs_pass + k_rs_small (sort replay)  [7] k_gapfill_band (corridor gap fills)
	 * [8] k_extd2_wide<512>  [9] k_extd2_wide<1024>  [10] index build (device sorts + CSR kernels)  [11] seeding kernels + anchor sort
	 * [12] k_approx_strips (large unbanded gap fills over several workgroups)  [13] k_extd2_lanes (banded problems, rows in registers)
	 * [14..15] unused.  kern_cells: DP cells evaluated by the DP kernels ([3] [4] [5] [7] [8] [9] [12] [13]), 0 elsewhere.
	 * The struct is read through pga_result_stats() only and is sized by PGA_STATS_VERSION: version 2 (round 2) grew the arrays from 10 to
	 * 16 entries and added kern_cells -- a consumer compiled against version 1 must be rebuilt. */
#define PGA_STATS_VERSION 2
	double kern_ms[16], kern_launches[16], kern_alg_bytes[16], kern_cells[16];
	double aligned_span;         /* sum of (qry_end - qry_start) over the emitted matches (secondary metric) */
} pga_stats_t;

/* seqs: n_seqs sequences, ASCII, NOT necessarily NUL-terminated (lengths in seq_lens); names: NUL-terminated
 * decimal BlockId strings (unique inside a group); group_off: n_groups+1 offsets into the sequence arrays. */
int pga_align_groups(const pga_params_t *params, int32_t n_groups, const int64_t *group_off,
                     const char *const *seqs, const uint32_t *seq_lens, const char *const *names, pga_result_t **out);
/* Same work split in two: pga_batch_create() copies the sequences to the device (2-step hand-over for callers
 * that keep block sequences resident across self-merge rounds); pga_batch_align() runs sketch -> index -> seed ->
 * chain -> extend on the resident bases and may be called repeatedly. */
typedef struct pga_batch_s pga_batch_t;
int pga_batch_create(int32_t n_groups, const int64_t *group_off, const char *const *seqs, const uint32_t *seq_lens, const char *const *names, pga_batch_t **out);
int pga_batch_align(pga_batch_t *batch, const pga_params_t *params, pga_result_t **out);
/* The batch of the next self-merge round: a sequence with seqs[i] == NULL is the src_index[i]-th sequence of `old` (numbered in hand-over
 * order) and is copied device to device from its packed store; the others are handed over as in pga_batch_create (block
 * sequences stay resident across rounds; the reference re-copies every block as ASCII per call, minimap2/src/index.rs:31-37).  seq_lens[i]
 * of a derived sequence must equal its source's.  `old` is not modified and may be freed afterwards. */
int pga_batch_derive(const pga_batch_t *old, int32_t n_groups, const int64_t *group_off, const char *const *seqs, const int64_t *src_index,
                     const uint32_t *seq_lens, const char *const *names, pga_batch_t **out);
/* Multi-GPU hosts, waves with fewer groups than ranks (the reference parallelises over the queries of one index,
 * align_with_minimap2_lib.rs:64-74): every shard indexes ALL sequences of the batch and maps, of every group, a contiguous range of the
 * queries (balanced by length).  The shards' match lists are disjoint; their union ordered by (group, query) is pga_batch_align's list. */
int pga_batch_align_shard(pga_batch_t *batch, const pga_params_t *params, int32_t shard, int32_t n_shards, pga_result_t **out);
/* The exchange step of a multi-GPU build is a gather of match lists and nothing else; transport is the host's (RCCL / MPI point to point: the
 * CIGAR pool dominates the payload and only the owner of the graph needs it).  These two are the host logic either side of it (no device work;
 * pangraph_amd/dist.py holds the same for the Python host, tests/test_dist_cpu.py compares them):
 *   pga_shard_groups_balanced  deals n_groups groups to `world` ranks, heaviest first, each to the rank that is lightest so far (ties: lower
 *                              rank); rank_of_group[g] receives the owner.  Deterministic: every rank computes the same plan without talking.
 *   pga_merge_match_lists      puts the lists of n_parts ranks together as ONE rank would have produced them: group ids become global
 *                              (local_to_global[r][g], or unchanged where local_to_global or local_to_global[r] is NULL), CIGAR offsets point into
 *                              the concatenated pool, records ordered by (group, query, the aligner's own order inside a query) -- also when the
 *                              queries of one group were split over the ranks (pga_batch_align_shard).  out_matches holds sum n_matches records,
 *                              out_cigars sum n_cigar_words words; returns 0, -1 on a group id outside its table. */
void pga_shard_groups_balanced(int32_t n_groups, const double *weights, int32_t world, int32_t *rank_of_group);
int pga_merge_match_lists(int32_t n_parts, const pga_match_t *const *matches, const int64_t *n_matches, const uint32_t *const *cigars,
                          const int64_t *n_cigar_words, const int32_t *const *local_to_global, const int32_t *n_local_groups,
                          pga_match_t *out_matches, uint32_t *out_cigars);
void pga_batch_free(pga_batch_t *batch);
int64_t pga_result_n_matches(const pga_result_t *r);
const pga_match_t *pga_result_matches(const pga_result_t *r);
const uint32_t *pga_result_cigars(const pga_result_t *r, uint64_t *n_ops);
const pga_stats_t *pga_result_stats(const pga_result_t *r);
void pga_result_free(pga_result_t *r);
const char *pga_last_error(void);
/* number of visible HIP devices (<=0: none, the library cannot run) and selection of the one to use.  The selection holds for the PROCESS:
 * HIP's current device belongs to the host thread, so every entry point of the library (mm_map included) applies the selected device to
 * the thread it is called on -- worker threads of the host need not call pga_set_device themselves. */
int pga_device_count(void);
int pga_set_device(int dev);
/* optional: create n streams ahead of time for the batch handles a host keeps in flight at once (they land on different hardware queues) */
int pga_warm_streams(int32_t n);
/* Gives back what the library holds of the device between calls: the idle blocks of its device-memory cache and the scratch slabs of the DP launch-lane
 * sets no call is using at the moment (both are bought again, at hipMalloc's price, by the next calls that need them).  For a host that shares the device
 * with another process or is done aligning for a while; hipFree synchronises the device.  Returns the bytes released. */
int64_t pga_trim(void);
/* the library's device-memory cache (diagnostics): out[0..5] = hipMalloc calls behind the cache, ns spent in them, hipFree calls, ns, bytes handed
 * out at the moment, bytes idle in the cache.  A host that sees hipFree calls grow from batch to batch has filled the device: the cache gives
 * the blocks that have been idle longest back, down to 4 GB below its limit (PGA_CACHE_GB, 90 % of the device), and hipFree synchronises the
 * device each time. */
void pga_mem_stats(int64_t out[6]);

/* Stage taps for parity tests (same semantics as the reference functions named in each comment). */
/* mm_sketch (sketch.c:77): minimizers of n sequences; out arrays are malloc()ed, caller frees with pga_free */
int pga_stage_sketch(int32_t n, const char *const *seqs, const uint32_t *lens, int w, int k, uint64_t **mz_xy, uint64_t **seq_off);
/* collect_seed_hits + mg_lchain_rmq (map.c:168-204, lchain.c:250-368) of an all-vs-all group:
 * anchors (x,y pairs) per query, chains u[] and compacted anchors per query, as flat malloc()ed arrays */
int pga_stage_chain(const pga_params_t *params, int32_t n, const char *const *seqs, const uint32_t *lens, const char *const *names,
                    uint64_t **anchors_xy, uint64_t **anchor_off, int32_t **n_u, int32_t **n_v, uint64_t **u, uint64_t **chain_xy, int32_t **rep_len, int32_t *mid_occ);
/* ksw_extd2_sse (ksw2_extd2_sse.c:34) on explicit nt4 sequences; ez[12] = max,max_q,max_t,mqe,mqe_t,mte,mte_q,score,zdropped,reach_end,n_cigar,0 */
int pga_stage_extd2(int32_t n_jobs, const uint8_t *const *q, const int32_t *qlen, const uint8_t *const *t, const int32_t *tlen,
                    int a, int b, int sc_ambi, int gapo, int gape, int gapo2, int gape2, const int32_t *w, const int32_t *zdrop, const int32_t *end_bonus, const int32_t *flag,
                    int32_t *ez, uint32_t **cigars, uint64_t *cigar_off);
/* radix_sort_128x (ksort.h:101-151, misc.c:155-159), the exact replay incl. the arrangement of equal keys: sorts every array
 * [seg_off[s], seg_off[s+1]) of the n_seg arrays in xy (two uint64 per record: x = key, y = payload) in place */
int pga_stage_sort(int32_t n_seg, const uint64_t *seg_off, uint64_t *xy);

/* ---- the step right behind find_matches (packages/pangraph/src/pangraph/graph_merging.rs:95-128 self_merge) ----
 * flags & 1: drop self matches (qry == ref inside a group, :107) and split every match at indels >= indel_len_threshold, with side
 * patches (pangraph/split_matches.rs:13-237; the reference's default threshold is 100, align/alignment_args.rs:8-11);
 * flags & 2: filter_matches per group (:187-216): alignment_energy2 (align/energy.rs:37-54, defaults alpha 100, beta 10) < 0, stable
 * sort by energy, greedy acceptance of matches whose query and reference intervals overlap no accepted interval of the same block.
 * Records come back in the same layout, group by group (ascending), accepted matches in energy order; ties keep the order of the
 * input records (the reference's tie order is that of its parallel aligner: undefined).  A CIGAR operation other than M I D = X
 * inside a kept stretch is an error, as in the reference (:62-65).  The result is freed with pga_result_free(). */
typedef struct { int32_t indel_len_threshold; int32_t flags; double alpha, beta; } pga_filter_params_t;
int pga_filter_matches(int64_t n, const pga_match_t *matches, const uint32_t *cigars, uint64_t n_ops, const pga_filter_params_t *fp, pga_result_t **out);
int pga_result_filter(const pga_result_t *res, const pga_filter_params_t *fp, pga_result_t **out);

/* ---- the guide tree (packages/pangraph/src/commands/build/build_run.rs:100 build_tree_using_neighbor_joining) ----
 * pga_mash_distance replaces distance/mash/mash_distance.rs:9-65 (minimizers_sketch of every sequence, minimizer.rs:49-160, with
 * MinimizersParams k, w -- the reference uses the defaults 15 and 100 -- then 1 - shared / own distinct minimizer values):
 * dist is n x n doubles, row major.  pga_guide_tree_nj replaces tree/neighbor_joining.rs:16-103: leaves are nodes 0..n-1 in input
 * order, join t (0-based) creates node n + t with children merges[2t] and merges[2t+1] (first the node that stood earlier in the
 * reference's node list); the last join is the root.  pga_guide_tree does both without moving the matrix through the host
 * (dist may be NULL).  A sequence without any minimizer is an error, as in the reference (it panics, mash_distance.rs:19-20).
 * No limit on n other than the n x n matrix in device memory (up to 2048 sequences the joining state lives in LDS, above that in
 * device memory).  pga_guide_tree_nj requires a SYMMETRIC matrix (what mash_distance produces) and returns -1 otherwise: the
 * reference's Q uses row sums and column sums, which the kernel takes from one pass.
 * Returns 0, or -1 with the message in pga_last_error(). */
int pga_mash_distance(int32_t n, const char *const *seqs, const uint32_t *lens, int k, int w, double *dist);
int pga_guide_tree_nj(int32_t n, const double *dist, int32_t *merges);
int pga_guide_tree(int32_t n, const char *const *seqs, const uint32_t *lens, int k, int w, double *dist, int32_t *merges);
/* The joining step picks the smallest Q (neighbor_joining.rs:77-96); its row and column sums are f64 sums taken in ndarray's order as
 * restated here (not pinned against ndarray itself).  pga_nj_near_ties() = the number of joins of the calling thread's
 * last pga_guide_tree_nj / pga_guide_tree call in which the Q of ANOTHER pair came within the error a different summation order can make
 * (8 m^2 2^-53 max|d| at m live nodes) of the chosen one; *first_join (may be NULL) = the first such join, -1 if none.  The joins at
 * m = 4 and m = 3 are left out of the count: there the Q of complementary pairs (at m = 3: of all pairs) are equal in exact arithmetic, so
 * the last two joins of EVERY tree -- the root and its children -- are decided by the rounding of these sums, in the reference as here.
 * 0 means no other join depends on the order the sums are taken in; otherwise a caller that needs the reference's tree bit for bit can
 * run its own joining on the distance matrix. */
int32_t pga_nj_near_ties(int32_t *first_join);
/* stage tap: the minimizers of every sequence in the reference's order (value = Minimizer.value, position = Minimizer.position with
 * the sequence's index as id); seq_off has n + 1 entries */
int pga_stage_mash_sketch(int32_t n, const char *const *seqs, const uint32_t *lens, int k, int w, uint64_t **value, uint64_t **position, uint64_t *seq_off);
/* ---- the re-alignment of a merged block's member sequences onto the anchor consensus ----
 * pga_map_variations replaces the loop of MergePromise::solve_promise (packages/pangraph/src/pangraph/reweave.rs:40-94) over
 * map_variations (packages/pangraph/src/align/map_variations.rs:39-77): align_with_nextclade (align/nextclade/align_with_nextclade.rs:
 * 24-75) = banded global alignment with free terminal gaps over simple_stripes(mean_shift, band_width + extra_band_width)
 * (align/nextclade/align/{band_2d.rs:36-57, score_matrix.rs:23-199, backtrace.rs:17-85}), the band doubled while the path touches
 * its boundary (align/nextclade/align/align.rs:55-62), then insertions_strip / find_nuc_changes and the terminal deletions.
 * One job per member sequence; ref and qry are upper-case IUPAC letters (not NUL-terminated); jobs that share a consensus should
 * pass the same pointer (it is uploaded once).  The caller keeps Edit::apply, reverse_complement and BandParameters::from_edits
 * (reweave.rs:53-75).  Per job: status 0, or the reference's error -- 1 the query is shorter than min_length (align.rs:42-46),
 * 2 a letter to_nuc rejects (alphabet/nuc.rs:99-121) or a literal '-' in ref / qry (to_nuc accepts it, the edit extraction of the reference would
 * read it as an alignment gap; block sequences never contain one, so it is rejected instead of reproduced), 3 the traceback left the band (the reference panics).  Substitutions in
 * reference order, deletions as the reference pushes them (internal ones ascending, then the leading, then the trailing one),
 * insertions ascending with pangraph's position convention (map_variations.rs:71-74).  The four arrays are freed with pga_free().
 * Returns 0, or -1 with the message in pga_last_error(). */
typedef struct {
	int32_t score_match, penalty_mismatch, penalty_gap_open, penalty_gap_extend;   /* NextalignParams::default(): 3, 1, 6, 0 (params.rs:142-170) */
	int32_t left_terminal_gaps_free, right_terminal_gaps_free, gap_align_left;     /* 1, 1, 1 */
	int32_t min_length, max_alignment_attempts, extra_band_width;                  /* map_variations.rs:45-52: 1, args (4), args (5) (build_args.rs:76-85) */
} pga_mapvar_params_t;
typedef struct { const char *ref, *qry; uint32_t ref_len, qry_len; int32_t mean_shift; uint32_t band_width; } pga_mapvar_job_t;
typedef struct { uint32_t pos, alt; } pga_sub_t;                    /* Sub { pos, alt }: alt is the query's letter */
typedef struct { uint32_t pos, len; } pga_del_t;                    /* Del { pos, len } */
typedef struct { uint32_t pos, len; uint64_t seq_off; } pga_ins_t;  /* Ins { pos, seq = ins_seq[seq_off .. seq_off + len) } */
typedef struct {
	int32_t status, score, attempts, hit_boundary;
	uint32_t n_subs, n_dels, n_inss, n_ins_bases;
	uint64_t sub_off, del_off, ins_off;                             /* first entry of the job in subs / dels / inss */
} pga_mapvar_res_t;
int pga_map_variations(int64_t n_jobs, const pga_mapvar_job_t *jobs, const pga_mapvar_params_t *params, pga_mapvar_res_t *res,
                       pga_sub_t **subs, pga_del_t **dels, pga_ins_t **inss, char **ins_seq);
void pga_free(void *p);

/* ---- reconsensus of the blocks a merge updated ----
 * pga_reconsensus replaces, for all updated blocks at once, analyze_blocks_for_reconsensus and the per-block work of reconsensus_graph
 * (packages/pangraph/src/reconsensus/reconsensus.rs:32-126): find_majority_edits (pangraph/pangraph_block.rs:191-256: an edit shared by more
 * than depth / 2 members), then per block either nothing (kind 0), apply_substitutions_to_block (kind 1: the consensus letters change, every
 * member's substitutions are reconciled, edits.rs:196-238) or edit_consensus_and_realign (kind 2, pangraph_block.rs:295-332: the majority
 * edits applied to the consensus, every member's sequence rebuilt with Edit::apply and re-aligned by map_variations with the band of
 * BandParameters::from_edits).  Counting, Edit::apply, reconciliation and the re-alignment run on the device; sequences built there feed the
 * aligner without leaving it.  detach_unaligned_nodes and the node / path maps (reconsensus.rs:76-88) stay with the caller.
 * Input: the members of block b are the next blocks[b].n_members entries of members[] (the reference's BTreeMap order), the edits of a
 * member the next n_subs / n_dels / n_inss entries of subs / dels / inss (insertion letters: ins_seq[seq_off .. seq_off + len)).
 * Output (freed with pga_rc_free): per block its kind, its consensus afterwards and its majority edits; per member (input order) its edits
 * against that consensus as a pga_mapvar_res_t (offsets into out->subs / dels / inss; kind 0: the input edits).  A negative kind is the
 * reference's error for that block (-2 a majority letter equals the consensus letter, -3 empty consensus, -4 no aligned position,
 * -5 a member holds a substitution and a deletion, or two substitutions, at one position); a block with a negative kind comes back with its
 * ORIGINAL consensus and the original edits of every member (status: the member's own code).  Member status 7: no aligned position
 * (map_variations.rs:32), otherwise the codes of pga_map_variations -- a caller checks the member status also when kind == 2 (a member
 * without an aligned position does not fail its block).  Returns 0, or -1 with the message in pga_last_error(). */
typedef struct { const char *consensus; uint32_t cons_len, n_members; } pga_rc_block_t;
typedef struct { uint32_t n_subs, n_dels, n_inss; } pga_rc_member_t;
typedef struct {
	int32_t kind; uint32_t cons_len; uint64_t cons_off;                  /* consensus afterwards: out->cons[cons_off .. cons_off + cons_len) */
	uint32_t n_subs, n_dels, n_inss; uint64_t sub_off, del_off, ins_off; /* majority edits: out->m_subs / m_dels / m_inss (letters in out->m_ins_seq) */
} pga_rc_block_res_t;
typedef struct {
	pga_rc_block_res_t *blocks; pga_mapvar_res_t *members;
	pga_sub_t *subs; pga_del_t *dels; pga_ins_t *inss; char *ins_seq;
	pga_sub_t *m_subs; pga_del_t *m_dels; pga_ins_t *m_inss; char *m_ins_seq;
	char *cons;
} pga_rc_out_t;
int pga_reconsensus(int64_t n_blocks, const pga_rc_block_t *blocks, const pga_rc_member_t *members, const pga_sub_t *subs, const pga_del_t *dels,
                    const pga_ins_t *inss, const char *ins_seq, const pga_mapvar_params_t *params, pga_rc_out_t *out);
void pga_rc_free(pga_rc_out_t *out);
int pga_stats_version(void);     /* == PGA_STATS_VERSION of the header the library was built with */
/* Measurement only (no reference interface behind it): the kern_ms sums of pga_stats_t count overlapping launches on different streams
 * and batches several times.  Between pga_busy_begin() and pga_busy_end() every event-bracketed launch of the process leaves its interval
 * on the device clock; pga_busy_end writes, for each of the PGA_N_KERNELS kernel families of pga_stats_t (same order), the length in ms of
 * the UNION of its intervals, and in busy_ms[PGA_N_KERNELS] the union over all families (n >= PGA_N_KERNELS + 1).  Returns the number of
 * intervals seen, -1 on error. */
#define PGA_N_KERNELS 16
int pga_busy_begin(void);
int pga_busy_end(double *busy_ms, int32_t n);
#ifdef __cplusplus
}
#endif
#endif
