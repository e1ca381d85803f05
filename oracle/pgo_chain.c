/* pgo_chain.c -- ORACLE (test infrastructure only).
 *
 * Co-linear chaining with range-min queries, restating mg_lchain_rmq / comput_sc_simple /
 * mg_chain_backtrack / mg_chain_bk_end / compact_a (lchain.c:9-111,232-368) and the balanced tree
 * behind it (krmq.h).  minimap2's asm presets set MM_F_RMQ (options.c:119), so this -- not the plain
 * chaining DP -- is what pangraph runs.
 *
 * The range-min structure is an AVL tree keyed by (y, i) whose nodes carry the minimum `pri` of their
 * subtree.  Which of several equal-`pri` nodes a query returns depends on the tree's shape and on the
 * order in which subtree minima were refreshed (krmq.h:110-150,163-166,177-181), so the tree is
 * re-enacted operation for operation on an index-addressed node pool:
 *   - insert: descend, attach a leaf, refresh minima bottom-up while the new node keeps winning,
 *     fix balance factors below the deepest non-balanced ancestor, one single or double rotation;
 *   - erase: unlink (replace by the leftmost node of the right subtree), refresh minima along the
 *     whole path, rebalance upward with single/double rotations;
 *   - rotations recompute the minimum of the node that moves down and hand the old subtree minimum to
 *     the node that moves up (it is NOT recomputed);
 *   - minimum refresh: keep own node unless the left child's minimum is <=, then take the right child's
 *     minimum unless the current one is strictly smaller (krmq.h:110-113).
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "pgo.h"

int pgo_dbg_tie_check = 0; long pgo_dbg_n_rmq = 0, pgo_dbg_n_tie = 0, pgo_dbg_n_inner = 0;

typedef struct {
	int32_t y;
	int64_t i;
	double pri;
	int32_t c[2];   /* children, -1 for none */
	int32_t s;      /* node holding the subtree minimum */
	int8_t bal;
	uint32_t size;
} node_t;

typedef struct {
	node_t *nd;
	int32_t n_nd, m_nd;
	int32_t *free_list; int32_t n_free, m_free;
	int32_t root;
} tree_t;

#define MAXD 64

static int32_t node_alloc(tree_t *t)
{
	if (t->n_free > 0) return t->free_list[--t->n_free];
	if (t->n_nd == t->m_nd) {
		t->m_nd = t->m_nd ? t->m_nd * 2 : 1024;
		t->nd = (node_t*)realloc(t->nd, (size_t)t->m_nd * sizeof(node_t));
	}
	return t->n_nd++;
}
static void node_free(tree_t *t, int32_t x)
{
	if (t->n_free == t->m_free) {
		t->m_free = t->m_free ? t->m_free * 2 : 1024;
		t->free_list = (int32_t*)realloc(t->free_list, (size_t)t->m_free * 4);
	}
	t->free_list[t->n_free++] = x;
}
static inline int key_cmp(int32_t ya, int64_t ia, const node_t *b) /* lchain.c:223 */
{
	return ya < b->y ? -1 : ya > b->y ? 1 : (ia > b->i) - (ia < b->i);
}
static inline uint32_t sz(const tree_t *t, int32_t x) { return x < 0 ? 0 : t->nd[x].size; }

static inline void refresh_min(tree_t *t, int32_t p) /* krmq.h:110-113 */
{
	node_t *nd = t->nd;
	int32_t q = nd[p].c[0], r = nd[p].c[1], s;
	s = (q < 0 || nd[p].pri < nd[nd[q].s].pri) ? p : nd[q].s;
	s = (r < 0 || nd[s].pri < nd[nd[r].s].pri) ? s : nd[r].s;
	nd[p].s = s;
}

static int32_t rotate1(tree_t *t, int32_t p, int dir) /* krmq.h:115-126 */
{
	node_t *nd = t->nd;
	int opp = 1 - dir;
	int32_t q = nd[p].c[opp], s = nd[p].s;
	uint32_t size_p = nd[p].size;
	nd[p].size -= nd[q].size - sz(t, nd[q].c[dir]);
	nd[q].size = size_p;
	/* minimum of p over its new children: its own `dir` child and q's `dir` child */
	{
		int32_t l = nd[p].c[dir], r = nd[q].c[dir], m;
		m = (l < 0 || nd[p].pri < nd[nd[l].s].pri) ? p : nd[l].s;
		m = (r < 0 || nd[m].pri < nd[nd[r].s].pri) ? m : nd[r].s;
		nd[p].s = m;
	}
	nd[q].s = s;
	nd[p].c[opp] = nd[q].c[dir];
	nd[q].c[dir] = p;
	return q;
}

static int32_t rotate2(tree_t *t, int32_t p, int dir) /* krmq.h:128-149 */
{
	node_t *nd = t->nd;
	int opp = 1 - dir, b1;
	int32_t q = nd[p].c[opp], r = nd[q].c[dir], s = nd[p].s;
	uint32_t size_x_dir = sz(t, nd[r].c[dir]);
	nd[r].size = nd[p].size;
	nd[p].size -= nd[q].size - size_x_dir;
	nd[q].size -= size_x_dir + 1;
	{ /* krmq_update_min(p, p->p[dir], r->p[dir]) -- argument order is (node, "left", "right") */
		int32_t l = nd[p].c[dir], rr = nd[r].c[dir], m;
		m = (l < 0 || nd[p].pri < nd[nd[l].s].pri) ? p : nd[l].s;
		m = (rr < 0 || nd[m].pri < nd[nd[rr].s].pri) ? m : nd[rr].s;
		nd[p].s = m;
	}
	{ /* krmq_update_min(q, q->p[opp], r->p[opp]) */
		int32_t l = nd[q].c[opp], rr = nd[r].c[opp], m;
		m = (l < 0 || nd[q].pri < nd[nd[l].s].pri) ? q : nd[l].s;
		m = (rr < 0 || nd[m].pri < nd[nd[rr].s].pri) ? m : nd[rr].s;
		nd[q].s = m;
	}
	nd[r].s = s;
	nd[p].c[opp] = nd[r].c[dir];
	nd[r].c[dir] = p;
	nd[q].c[dir] = nd[r].c[opp];
	nd[r].c[opp] = q;
	b1 = dir == 0 ? +1 : -1;
	if (nd[r].bal == b1) nd[q].bal = 0, nd[p].bal = (int8_t)-b1;
	else if (nd[r].bal == 0) nd[q].bal = nd[p].bal = 0;
	else nd[q].bal = (int8_t)b1, nd[p].bal = 0;
	nd[r].bal = 0;
	return r;
}

static void tree_insert(tree_t *t, int32_t x) /* krmq.h:152-200; keys are unique here */
{
	node_t *nd = t->nd;
	uint8_t stack[MAXD];
	int32_t path[MAXD];
	int32_t bp = t->root, bq = -1, p, q, r;
	int top = 0, path_len = 0, which = 0;
	for (p = bp, q = bq; p >= 0; q = p, p = nd[p].c[which]) {
		int cmp = key_cmp(nd[x].y, nd[x].i, &nd[p]);
		assert(cmp != 0);
		if (nd[p].bal != 0) bq = q, bp = p, top = 0;
		stack[top++] = (uint8_t)(which = (cmp > 0));
		path[path_len++] = p;
	}
	nd[x].bal = 0, nd[x].size = 1, nd[x].c[0] = nd[x].c[1] = -1, nd[x].s = x;
	if (q < 0) t->root = x; else nd[q].c[which] = x;
	if (bp < 0) return;
	for (int i = 0; i < path_len; ++i) ++nd[path[i]].size;
	for (int i = path_len - 1; i >= 0; --i) {
		refresh_min(t, path[i]);
		if (nd[path[i]].s != x) break;
	}
	for (p = bp, top = 0; p != x; p = nd[p].c[stack[top]], ++top) {
		if (stack[top] == 0) --nd[p].bal; else ++nd[p].bal;
	}
	if (nd[bp].bal > -2 && nd[bp].bal < 2) return;
	which = (nd[bp].bal < 0);
	int b1 = which == 0 ? +1 : -1;
	q = nd[bp].c[1 - which];
	if (nd[q].bal == b1) {
		r = rotate1(t, bp, which);
		nd[q].bal = nd[bp].bal = 0;
	} else r = rotate2(t, bp, which);
	if (bq < 0) t->root = r;
	else nd[bq].c[bp != nd[bq].c[0]] = r;
}

static int32_t tree_find(const tree_t *t, int32_t y, int64_t i) /* krmq.h:84-96 */
{
	int32_t p = t->root;
	while (p >= 0) {
		int cmp = key_cmp(y, i, &t->nd[p]);
		if (cmp < 0) p = t->nd[p].c[0]; else if (cmp > 0) p = t->nd[p].c[1]; else break;
	}
	return p;
}

/* krmq.h:203-285.  The reference walks from a stack copy of the root ("fake") whose left child is the
 * real root; here slot `FAKE` of path[] stands for it and child links of the fake are redirected to t->root. */
#define FAKE (-2)
static inline int32_t get_child(tree_t *t, int32_t p, int d) { return p == FAKE ? (d == 0 ? t->root : -1) : t->nd[p].c[d]; }
static inline void set_child(tree_t *t, int32_t p, int d, int32_t v) { if (p == FAKE) { if (d == 0) t->root = v; } else t->nd[p].c[d] = v; }

static void tree_erase(tree_t *t, int32_t x)
{
	node_t *nd = t->nd;
	int32_t path[MAXD], p;
	uint8_t dir[MAXD];
	int d = 0, i;
	{ /* locate x, recording the path from the fake root */
		int cmp = -1;
		p = FAKE;
		while (cmp) {
			int which = (cmp > 0);
			dir[d] = (uint8_t)which, path[d++] = p;
			p = get_child(t, p, which);
			assert(p >= 0);
			cmp = key_cmp(nd[x].y, nd[x].i, &nd[p]);
		}
		assert(p == x);
	}
	for (i = 1; i < d; ++i) --nd[path[i]].size;
	if (nd[p].c[1] < 0) {
		set_child(t, path[d-1], dir[d-1], nd[p].c[0]);
	} else {
		int32_t q = nd[p].c[1];
		if (nd[q].c[0] < 0) {
			nd[q].c[0] = nd[p].c[0];
			nd[q].bal = nd[p].bal;
			set_child(t, path[d-1], dir[d-1], q);
			path[d] = q, dir[d++] = 1;
			nd[q].size = nd[p].size - 1;
		} else {
			int32_t r;
			int e = d++;
			for (;;) {
				dir[d] = 0, path[d++] = q;
				r = nd[q].c[0];
				if (nd[r].c[0] < 0) break;
				q = r;
			}
			nd[r].c[0] = nd[p].c[0];
			nd[q].c[0] = nd[r].c[1];
			nd[r].c[1] = nd[p].c[1];
			nd[r].bal = nd[p].bal;
			set_child(t, path[e-1], dir[e-1], r);
			path[e] = r, dir[e] = 1;
			for (i = e + 1; i < d; ++i) --nd[path[i]].size;
			nd[r].size = nd[p].size - 1;
		}
	}
	for (i = d - 1; i >= 1; --i) refresh_min(t, path[i]); /* path[0] is the fake root: its minimum is never read */
	while (--d > 0) {
		int32_t q = path[d];
		int which = dir[d], other = 1 - which, b1 = 1, b2 = 2;
		if (which) b1 = -b1, b2 = -b2;
		nd[q].bal = (int8_t)(nd[q].bal + b1);
		if (nd[q].bal == b1) break;
		else if (nd[q].bal == b2) {
			int32_t r = nd[q].c[other];
			if (nd[r].bal == -b1) {
				set_child(t, path[d-1], dir[d-1], rotate2(t, q, which));
			} else {
				set_child(t, path[d-1], dir[d-1], rotate1(t, q, which));
				if (nd[r].bal == 0) {
					nd[r].bal = (int8_t)-b1;
					nd[q].bal = (int8_t)b1;
					break;
				} else nd[r].bal = nd[q].bal = 0;
			}
		}
	}
}

/* krmq.h:98-140: minimum-pri node with key in the CLOSED interval [(lo_y, lo_i), (hi_y, hi_i)] */
static int32_t tree_rmq(const tree_t *t, int32_t lo_y, int64_t lo_i, int32_t hi_y, int64_t hi_i)
{
	const node_t *nd = t->nd;
	int32_t path[2][MAXD], p, min;
	int plen[2] = {0, 0}, pcmp[2][MAXD], i, cmp, lca;
	if (t->root < 0) return -1;
	for (p = t->root; p >= 0;) {
		cmp = key_cmp(lo_y, lo_i, &nd[p]);
		path[0][plen[0]] = p, pcmp[0][plen[0]++] = cmp;
		if (cmp < 0) p = nd[p].c[0]; else if (cmp > 0) p = nd[p].c[1]; else break;
	}
	for (p = t->root; p >= 0;) {
		cmp = key_cmp(hi_y, hi_i, &nd[p]);
		path[1][plen[1]] = p, pcmp[1][plen[1]++] = cmp;
		if (cmp < 0) p = nd[p].c[0]; else if (cmp > 0) p = nd[p].c[1]; else break;
	}
	for (i = 0; i < plen[0] && i < plen[1]; ++i)
		if (path[0][i] == path[1][i] && pcmp[0][i] <= 0 && pcmp[1][i] >= 0) break;
	if (i == plen[0] || i == plen[1]) return -1;
	lca = i, min = path[0][lca];
	for (i = lca + 1; i < plen[0]; ++i) {
		if (pcmp[0][i] <= 0) {
			int32_t u = path[0][i], r = nd[u].c[1];
			if (nd[u].pri < nd[min].pri) min = u;
			if (r >= 0 && nd[nd[r].s].pri < nd[min].pri) min = nd[r].s;
		}
	}
	for (i = lca + 1; i < plen[1]; ++i) {
		if (pcmp[1][i] >= 0) {
			int32_t u = path[1][i], l = nd[u].c[0];
			if (nd[u].pri < nd[min].pri) min = u;
			if (l >= 0 && nd[nd[l].s].pri < nd[min].pri) min = nd[l].s;
		}
	}
	return min;
}

/* iterator over keys in descending order starting from the largest key <= (y, i)
 * (krmq_interval + krmq_itr_find + krmq_itr_prev, krmq.h:97-109,306-340) */
typedef struct { int32_t stack[MAXD]; int top; } iter_t;
static int iter_seek_le(const tree_t *t, int32_t y, int64_t i, iter_t *it)
{
	const node_t *nd = t->nd;
	int32_t p = t->root, lower = -1;
	while (p >= 0) {
		int cmp = key_cmp(y, i, &nd[p]);
		if (cmp < 0) p = nd[p].c[0];
		else if (cmp > 0) lower = p, p = nd[p].c[1];
		else { lower = p; break; }
	}
	if (lower < 0) return 0;
	it->top = -1;
	for (p = t->root; p >= 0;) {
		int cmp = key_cmp(nd[lower].y, nd[lower].i, &nd[p]);
		it->stack[++it->top] = p;
		if (cmp < 0) p = nd[p].c[0]; else if (cmp > 0) p = nd[p].c[1]; else break;
	}
	return 1;
}
static int iter_prev(const tree_t *t, iter_t *it)
{
	const node_t *nd = t->nd;
	int32_t p;
	if (it->top < 0) return 0;
	p = nd[it->stack[it->top]].c[0];
	if (p >= 0) {
		for (; p >= 0; p = nd[p].c[1]) it->stack[++it->top] = p;
		return 1;
	}
	do { p = it->stack[it->top--]; } while (it->top >= 0 && p == nd[it->stack[it->top]].c[0]);
	return it->top < 0 ? 0 : 1;
}

static inline int32_t score_pair(const pg128 *ai, const pg128 *aj, float pen_gap, float pen_skip, int32_t *exact, int32_t *width) /* lchain.c:232-248 */
{
	int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, q_span, sc;
	dr = (int32_t)(ai->x - aj->x);
	*width = dd = dr > dq ? dr - dq : dq - dr;
	dg = dr < dq ? dr : dq;
	q_span = aj->y >> 32 & 0xff;
	sc = q_span < dg ? q_span : dg;
	if (exact) *exact = (dd == 0 && dg <= q_span);
	if (dd || dq > q_span) {
		float lin_pen, log_pen;
		lin_pen = pen_gap * (float)dd + pen_skip * (float)dg;
		log_pen = dd >= 1 ? pgo_log2f_approx((float)(dd + 1)) : 0.0f;
		sc -= (int)(lin_pen + .5f * log_pen);
	}
	return sc;
}

static int64_t bk_end(int32_t max_drop, const pg128 *z, const int32_t *f, const int64_t *p, int32_t *t, int64_t k) /* lchain.c:9-25 */
{
	int64_t i = (int64_t)z[k].y, end_i = -1, max_i = i;
	int32_t max_s = 0;
	if (i < 0 || t[i] != 0) return i;
	do {
		int32_t s;
		t[i] = 2;
		end_i = i = p[i];
		s = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
		if (s > max_s) max_s = s, max_i = i;
		else if (max_s - s > max_drop) break;
	} while (i >= 0 && t[i] == 0);
	for (i = (int64_t)z[k].y; i >= 0 && i != end_i; i = p[i]) t[i] = 0;
	return max_i;
}

pg128 *pgo_lchain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc,
                      float chn_pen_gap, float chn_pen_skip, int64_t n, pg128 *a, int *n_u_, uint64_t **u_)
{
	int32_t *f, *t, *v, max_drop = bw;
	int64_t *p, i, i0, st = 0, st_inner = 0;
	tree_t T = {0}, Ti = {0};
	int32_t *slot = 0, *slot_i = 0; /* anchor index -> node id in each tree (-1 when absent) */
	T.root = Ti.root = -1;
	*u_ = 0, *n_u_ = 0;
	if (n == 0 || a == 0) { free(a); return 0; }
	if (max_dist < bw) max_dist = bw;
	if (max_dist_inner <= 0 || max_dist_inner >= max_dist) max_dist_inner = 0;
	p = (int64_t*)malloc((size_t)n * 8);
	f = (int32_t*)malloc((size_t)n * 4);
	t = (int32_t*)calloc((size_t)n, 4);
	v = (int32_t*)malloc((size_t)n * 4);
	slot = (int32_t*)malloc((size_t)n * 4);
	slot_i = (int32_t*)malloc((size_t)n * 4);
	for (i = 0; i < n; ++i) slot[i] = slot_i[i] = -1;

	for (i = i0 = 0; i < n; ++i) { /* lchain.c:276-357 */
		int64_t max_j = -1;
		int32_t q_span = a[i].y >> 32 & 0xff, max_f = q_span;
		if (i0 < i && a[i0].x != a[i].x) { /* anchors sharing x never chain to each other: insert them late */
			for (int64_t j = i0; j < i; ++j) {
				int32_t q = node_alloc(&T);
				T.nd[q].y = (int32_t)a[j].y, T.nd[q].i = j;
				T.nd[q].pri = -(f[j] + 0.5 * chn_pen_gap * ((int32_t)a[j].x + (int32_t)a[j].y));
				tree_insert(&T, q);
				slot[j] = q;
				if (max_dist_inner > 0) {
					int32_t r = node_alloc(&Ti);
					Ti.nd[r].y = T.nd[q].y, Ti.nd[r].i = j, Ti.nd[r].pri = T.nd[q].pri;
					tree_insert(&Ti, r);
					slot_i[j] = r;
				}
			}
			i0 = i;
		}
		while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + (uint64_t)max_dist || (int)sz(&T, T.root) > cap_rmq_size)) {
			if (slot[st] >= 0) {
				assert(tree_find(&T, (int32_t)a[st].y, st) == slot[st]);
				tree_erase(&T, slot[st]); node_free(&T, slot[st]); slot[st] = -1;
			}
			++st;
		}
		if (max_dist_inner > 0) {
			while (st_inner < i && (a[i].x >> 32 != a[st_inner].x >> 32 || a[i].x > a[st_inner].x + (uint64_t)max_dist_inner || (int)sz(&Ti, Ti.root) > cap_rmq_size)) {
				if (slot_i[st_inner] >= 0) {
					tree_erase(&Ti, slot_i[st_inner]); node_free(&Ti, slot_i[st_inner]); slot_i[st_inner] = -1;
				}
				++st_inner;
			}
		}
		{
			int32_t q = tree_rmq(&T, (int32_t)a[i].y - max_dist, INT32_MAX, (int32_t)a[i].y, 0);
			if (pgo_dbg_tie_check && q >= 0) { /* debug: brute-force count of candidates tied with the returned minimum */
				int64_t jj, n_tie = 0;
				for (jj = st; jj < i0; ++jj) {
					if (slot[jj] < 0) continue;
					int32_t yj = (int32_t)a[jj].y;
					if (yj <= (int32_t)a[i].y - max_dist || yj > (int32_t)a[i].y || (yj == (int32_t)a[i].y && jj > 0)) continue;
					if (T.nd[slot[jj]].pri == T.nd[q].pri) ++n_tie;
				}
				++pgo_dbg_n_rmq;
				if (n_tie > 1) ++pgo_dbg_n_tie;
			}
			if (q >= 0) {
				int32_t sc, exact, width, n_skip = 0;
				int64_t j = T.nd[q].i;
				sc = f[j] + score_pair(&a[i], &a[j], chn_pen_gap, chn_pen_skip, &exact, &width);
				if (width <= bw && sc > max_f) max_f = sc, max_j = j;
				if (!exact && Ti.root >= 0 && (int32_t)a[i].y > 0) {
					++pgo_dbg_n_inner;
					iter_t it;
					if (iter_seek_le(&Ti, (int32_t)a[i].y - 1, n, &it)) {
						for (;;) {
							const node_t *e = &Ti.nd[it.stack[it.top]];
							if (e->y < (int32_t)a[i].y - max_dist_inner) break;
							j = e->i;
							sc = f[j] + score_pair(&a[i], &a[j], chn_pen_gap, chn_pen_skip, 0, &width);
							if (width <= bw) {
								if (sc > max_f) {
									max_f = sc, max_j = j;
									if (n_skip > 0) --n_skip;
								} else if (t[j] == (int32_t)i) {
									if (++n_skip > max_chn_skip) break;
								}
								if (p[j] >= 0) t[p[j]] = (int32_t)i;
							}
							if (!iter_prev(&Ti, &it)) break;
						}
					}
				}
			}
		}
		f[i] = max_f, p[i] = max_j;
		v[i] = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f;
	}
	free(T.nd); free(T.free_list); free(Ti.nd); free(Ti.free_list); free(slot); free(slot_i);

	/* ---- backtrack (lchain.c:27-76): candidate ends sorted by score with the unstable radix sort ---- */
	int64_t n_z = 0, k, n_v = 0;
	int32_t n_u = 0;
	for (i = 0; i < n; ++i) if (f[i] >= min_sc) ++n_z;
	uint64_t *u = 0;
	if (n_z > 0) {
		pg128 *z = (pg128*)malloc((size_t)n_z * sizeof(pg128));
		for (i = 0, k = 0; i < n; ++i) if (f[i] >= min_sc) z[k].x = (uint64_t)f[i], z[k++].y = (uint64_t)i;
		pgo_radix_sort_128x(z, z + n_z);
		memset(t, 0, (size_t)n * 4);
		u = (uint64_t*)malloc((size_t)n_z * 8);
		for (k = n_z - 1; k >= 0; --k) {
			if (t[z[k].y] != 0) continue;
			int64_t n_v0 = n_v, end_i;
			int32_t sc;
			end_i = bk_end(max_drop, z, f, p, t, k);
			for (i = (int64_t)z[k].y; i != end_i; i = p[i]) v[n_v++] = (int32_t)i, t[i] = 1;
			sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
			if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) u[n_u++] = (uint64_t)sc << 32 | (uint64_t)(n_v - n_v0);
			else n_v = n_v0;
		}
		free(z);
	}
	free(p); free(f); free(t);
	*n_u_ = n_u, *u_ = u;
	if (n_u == 0) { free(a); free(v); free(u); *u_ = 0; return 0; }

	/* ---- compact (lchain.c:78-111): chains in ascending anchor order, sorted by first target position ---- */
	pg128 *b = (pg128*)malloc((size_t)n_v * sizeof(pg128));
	pg128 *wv = (pg128*)malloc((size_t)n_u * sizeof(pg128));
	for (i = 0, k = 0; i < n_u; ++i) {
		int32_t k0 = (int32_t)k, ni = (int32_t)u[i];
		for (int32_t j = 0; j < ni; ++j) b[k++] = a[v[k0 + (ni - j - 1)]];
	}
	for (i = k = 0; i < n_u; ++i) {
		wv[i].x = b[k].x, wv[i].y = (uint64_t)k << 32 | (uint64_t)i;
		k += (int32_t)u[i];
	}
	pgo_radix_sort_128x(wv, wv + n_u);
	uint64_t *u2 = (uint64_t*)malloc((size_t)n_u * 8);
	pg128 *out = (pg128*)malloc((size_t)n_v * sizeof(pg128));
	for (i = k = 0; i < n_u; ++i) {
		int32_t j = (int32_t)wv[i].y, nn = (int32_t)u[j];
		u2[i] = u[j];
		memcpy(&out[k], &b[wv[i].y >> 32], (size_t)nn * sizeof(pg128));
		k += nn;
	}
	memcpy(u, u2, (size_t)n_u * 8);
	free(a); free(b); free(wv); free(u2); free(v);
	return out;
}
