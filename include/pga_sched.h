/* pga_sched.h -- the ready-set schedule of a whole build, for hosts that are not Python.
 *
 * What it replaces on the reference side: nothing is called through FFI there -- the reference walks the guide tree in post-order, one merge after
 * the other (packages/pangraph/src/commands/build/build_run.rs:111-128), and every merge runs its self-merge loop of `find_matches` calls
 * (packages/pangraph/src/pangraph/graph_merging.rs:26-69,95-128).  A device wants several calls in flight, so the host has to know which calls
 * are READY: (v, round 0) needs the final round of both children of v, (v, round r) needs (v, round r - 1).  These entry points hold that
 * bookkeeping (no device work, no HIP call): the host creates the task graph once, starts a run, and its worker threads -- one per slot --
 * loop over  pga_sched_take -> build the batch (pga_batch_create / pga_batch_align, include/pga_align.h) -> pga_sched_finish.
 * Decisions are those of pangraph_amd/schedule.py (`ReadySet`); tests/test_schedule_cpu.py steps both through the same simulated builds.
 * INTEGRATION.md section D shows the Rust side.
 */
#ifndef PGA_SCHED_H
#define PGA_SCHED_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pga_sched pga_sched_t;

/* Task graph: task i depends on dep[dep_off[i] .. dep_off[i + 1]) (task ids; a dependency must have a smaller or larger id, no cycles),
 * holds bases[i] bases in n_seqs[i] sequences.  Computes every task's priority: the estimated cost of the dependency chain from the task to
 * the last call that needs it (seconds alone on the device: pga_sched_cost).  NULL on a cycle or a bad dependency (pga_sched_error()). */
pga_sched_t *pga_sched_create(int32_t n_tasks, const int64_t *dep_off, const int32_t *dep, const int64_t *bases, const int32_t *n_seqs);
void pga_sched_destroy(pga_sched_t *s);
const char *pga_sched_error(void);                       /* the calling thread's last error text */

/* seconds a call of `bases` bases in `n_seqs` sequences takes alone on the device (rough; measured on MI355X, see schedule.py:cost_estimate) */
double pga_sched_cost(int64_t bases, int32_t n_seqs);
/* prio[i] of every task (n_tasks doubles) */
void pga_sched_prio(const pga_sched_t *s, double *prio);

/* Starts a run over the tasks `only[0 .. n_only)` (only == NULL: all tasks); `done[0 .. n_done)` count as finished from the start.
 * slots: batches in flight at most; cap_bases: bases per batch at most (a single larger task still goes alone; a large ready set is spread over
 * the free slots); min_batch_bases: lower bound of that cap; express > 0: that many slots are reserved for calls whose remaining path is within
 * express_eps seconds of the longest remaining path of the run (at most express_cap bases per such batch).
 * Returns 0, or -1 when a task depends on one that is neither done nor part of the run. */
int pga_sched_start(pga_sched_t *s, const int32_t *only, int32_t n_only, const int32_t *done, int32_t n_done, int32_t slots, double cap_bases,
                    double min_batch_bases, int32_t express, double express_eps, double express_cap);

/* Blocks until a batch may start (a slot is free and a call is ready) or the run is over.  Writes the task ids of the batch to ids[0 .. return
 * value) -- largest remaining path first -- and a ticket for pga_sched_finish.  Returns 0 when every task of the run has finished or the run was
 * aborted (workers leave their loop), -1 when cap_ids is too small for the batch (nothing is taken; *ticket holds the size needed). */
int32_t pga_sched_take(pga_sched_t *s, int32_t *ids, int32_t cap_ids, int32_t *ticket);
/* The same without blocking: returns -2 when nothing may start right now (a single-threaded host that polls; the simulation of the tests). */
int32_t pga_sched_try_take(pga_sched_t *s, int32_t *ids, int32_t cap_ids, int32_t *ticket);
/* The batch behind `ticket` is done: its tasks count as finished, the calls that waited for them become ready, its slot is free. */
void pga_sched_finish(pga_sched_t *s, int32_t ticket);
/* A batch failed: every blocked pga_sched_take returns 0 from now on. */
void pga_sched_abort(pga_sched_t *s);
/* tasks of the run that have not finished yet */
int32_t pga_sched_left(pga_sched_t *s);

/* Multi-GPU: cuts the guide tree into at least world * per_rank subtrees (the heaviest is split at its root until there are enough), deals them
 * to the ranks heaviest first and writes owner[t] = rank of task t, or -1 for the calls above the cut that all ranks run together after the
 * first gather (pga_batch_align_shard).  Nodes 0 .. n_nodes - 1, node 0 the root, children have larger ids than their parent, child0[v] =
 * child1[v] = -1 for a leaf; task_node[t] = the node whose merge task t belongs to.  Deterministic: every rank computes the same plan without
 * talking.  Returns the number of tasks above the cut, -1 on bad input.  (schedule.py:partition_subtrees) */
int32_t pga_sched_partition(int32_t n_nodes, const int32_t *child0, const int32_t *child1, int32_t n_tasks, const int32_t *task_node,
                            const int64_t *task_bases, int32_t world, int32_t per_rank, int32_t *owner);

#ifdef __cplusplus
}
#endif
#endif
