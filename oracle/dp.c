/*
 * dp.c -- ORACLE (test infrastructure only): the dynamic programming over the belief graph,
 *   conditional_dijkstra          src/belief_graph.rs:89-175
 *   extract_policy                src/belief_graph.rs:177-212
 *   get_best_expected_children    src/belief_graph.rs:214-263
 *   transition_probability        src/common.rs:187-190
 *   PTO::compute_expected_costs_to_goals  src/pto.rs:261-275
 * on explicit graphs (what the reference's own tests build by hand, belief_graph.rs:278-500; a node carries a belief
 * id -- the clustering key of the policy -- and, separately, the row of its belief vector: the reference's second
 * test graph gives ten nodes id 2 with the vector of id 1) and on the belief
 * graph of the context (belief.c).  cost_evaluator is norm2 (the PTOFuncs default, pto_graph.rs:150-152).
 * The priority queue is an indexed binary heap; `push` of a queued item replaces its priority like the
 * priority_queue crate.  Ties pop in heap order, not in the crate's: the result does not depend on it -- every
 * operation of the relaxation is monotone in f64, so any pop order ends in the same fixpoint.
 * Pinned by the two known-answer tests of belief_graph.rs:502-567 (tests/test_oracle_dp.py).
 */
#include "orc_internal.h"
#include <math.h>

/* common.rs:187-190 */
static double transition_probability(const double *parent_bs, const double *child_bs, uint32_t nw) {
    double s = 0.0;
    for (uint32_t w = 0; w < nw; ++w) s = s + (child_bs[w] > 0.0 ? parent_bs[w] : 0.0);
    return s;
}

typedef struct {
    uint64_t *item;     /* heap of node ids */
    double *prio;
    int64_t *pos;       /* node id -> heap index, -1 = not queued */
    size_t n;
} pq;

static int pq_better(const pq *q, size_t a, size_t b) { return q->prio[a] < q->prio[b]; }   /* Priority::cmp: smaller prio = greater */
static void pq_swap(pq *q, size_t a, size_t b) {
    uint64_t ti = q->item[a]; q->item[a] = q->item[b]; q->item[b] = ti;
    double tp = q->prio[a]; q->prio[a] = q->prio[b]; q->prio[b] = tp;
    q->pos[q->item[a]] = (int64_t)a; q->pos[q->item[b]] = (int64_t)b;
}
static void pq_up(pq *q, size_t i) {
    while (i && pq_better(q, i, (i - 1) / 2)) { pq_swap(q, i, (i - 1) / 2); i = (i - 1) / 2; }
}
static void pq_down(pq *q, size_t i) {
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < q->n && pq_better(q, l, m)) m = l;
        if (r < q->n && pq_better(q, r, m)) m = r;
        if (m == i) return;
        pq_swap(q, i, m);
        i = m;
    }
}
static void pq_push(pq *q, uint64_t id, double prio) {
    if (q->pos[id] >= 0) {
        size_t i = (size_t)q->pos[id];
        q->prio[i] = prio;
        pq_up(q, i);
        pq_down(q, (size_t)q->pos[id]);
        return;
    }
    q->item[q->n] = id; q->prio[q->n] = prio; q->pos[id] = (int64_t)q->n;
    pq_up(q, q->n++);
}
static uint64_t pq_pop(pq *q) {
    uint64_t id = q->item[0];
    pq_swap(q, 0, q->n - 1);
    q->n--;
    q->pos[id] = -1;
    if (q->n) pq_down(q, 0);
    return id;
}

/* 0 ok; -1 "node type should be know at this stage!"; -2 assert!(p > 0.0) */
int orc_conditional_dijkstra(uint64_t n, const double *xy, const uint32_t *belief_vec, const double *beliefs, uint32_t nw,
                             const uint8_t *types, const uint64_t *coff, const uint32_t *cid, const uint64_t *poff, const uint32_t *pid,
                             const uint64_t *finals, uint64_t n_final, double *dist) {
    pq q;
    q.item = malloc((n + 1) * sizeof(uint64_t)); q.prio = malloc((n + 1) * sizeof(double)); q.pos = malloc((n + 1) * sizeof(int64_t));
    q.n = 0;
    for (uint64_t i = 0; i < n; ++i) { dist[i] = INFINITY; q.pos[i] = -1; }
    for (uint64_t k = 0; k < n_final; ++k) { dist[finals[k]] = 0.0; pq_push(&q, finals[k], 0.0); }
    int rc = 0;
    while (q.n && !rc) {
        const uint64_t v = pq_pop(&q);
        for (uint64_t e = poff[v]; e < poff[v + 1] && !rc; ++e) {
            const uint64_t u = pid[e];
            double alternative = 0.0;
            if (types[u] == 1) {
                alternative += orc_norm2(xy + 2 * u, xy + 2 * v) + dist[v];
            } else if (types[u] == 2) {
                for (uint64_t c = coff[u]; c < coff[u + 1]; ++c) {
                    const uint64_t vv = cid[c];
                    const double p = transition_probability(beliefs + (size_t)belief_vec[u] * nw, beliefs + (size_t)belief_vec[vv] * nw, nw);
                    if (!(p > 0.0)) { rc = -2; break; }
                    alternative += p * (orc_norm2(xy + 2 * u, xy + 2 * vv) + dist[vv]);
                }
            } else {
                rc = -1;
            }
            if (!rc && alternative < dist[u]) {
                dist[u] = alternative;
                pq_push(&q, u, alternative);
            }
        }
    }
    free(q.item); free(q.prio); free(q.pos);
    return rc;
}

/* extract_policy: fills original_id / parent (-1 for the root) / is_leaf per policy node in add_node order; returns the
 * number of policy nodes, -1 empty graph, -2 a failed assert of get_best_expected_children, -3 cap too small */
int64_t orc_extract_policy(uint64_t n, const double *xy, const uint32_t *belief_id, const uint32_t *belief_vec, const double *beliefs, uint32_t nw,
                           const uint64_t *coff, const uint32_t *cid, const double *dist,
                           uint64_t *original_id, int64_t *parent, uint8_t *is_leaf, uint64_t cap) {
    if (!n) return -1;
    uint64_t np = 0;
    size_t L = 0, Lcap = 64;
    uint64_t *lp = malloc(Lcap * sizeof(uint64_t)), *lb = malloc(Lcap * sizeof(uint64_t));
    if (cap < 1) { free(lp); free(lb); return -3; }
    original_id[0] = 0; parent[0] = -1; is_leaf[0] = 0; np = 1;
    lp[0] = 0; lb[0] = 0; L = 1;
    int64_t rc = 0;
    uint32_t *keys = NULL;
    size_t kcap = 0;
    while (L && rc >= 0) {
        --L;
        const uint64_t pol = lp[L], bn = lb[L];
        /* get_best_expected_children: children clustered by their belief id (BTreeMap: ascending), best of each cluster */
        const uint64_t c0 = coff[bn], c1 = coff[bn + 1];
        if (c1 - c0 > kcap) { kcap = (size_t)(c1 - c0) + 16; keys = realloc(keys, kcap * sizeof(uint32_t)); }
        size_t nk = 0;
        for (uint64_t c = c0; c < c1; ++c) {
            const uint32_t key = belief_id[cid[c]];
            size_t at = 0;
            while (at < nk && keys[at] < key) ++at;
            if (at < nk && keys[at] == key) continue;
            for (size_t m = nk; m > at; --m) keys[m] = keys[m - 1];
            keys[at] = key;
            ++nk;
        }
        for (size_t kk = 0; kk < nk && rc >= 0; ++kk) {
            uint64_t best = 0;
            int have = 0;
            double p = 0.0, best_cost = INFINITY;
            for (uint64_t c = c0; c < c1; ++c) {
                const uint64_t child = cid[c];
                if (belief_id[child] != keys[kk]) continue;
                if (!have) {
                    best = child;
                    have = 1;
                    p = transition_probability(beliefs + (size_t)belief_vec[bn] * nw, beliefs + (size_t)belief_vec[best] * nw, nw);
                    if (!(p > 0.0)) { rc = -2; break; }
                }
                const double cost = p * (orc_norm2(xy + 2 * bn, xy + 2 * child) + dist[child]);
                if (cost < best_cost) { best_cost = cost; best = child; }
            }
            if (rc < 0) break;
            if (!(p * dist[best] <= dist[bn])) { rc = -2; break; }
            if (np == cap) { rc = -3; break; }
            const uint8_t leaf = dist[best] == 0.0;
            original_id[np] = best; parent[np] = (int64_t)pol; is_leaf[np] = leaf;
            if (!leaf) {
                if (L == Lcap) { Lcap *= 2; lp = realloc(lp, Lcap * sizeof(uint64_t)); lb = realloc(lb, Lcap * sizeof(uint64_t)); }
                lp[L] = np; lb[L] = best; ++L;
            }
            ++np;
        }
    }
    free(lp); free(lb); free(keys);
    return rc < 0 ? rc : (int64_t)np;
}

/* ---- on the context's belief graph (belief.c): PTO::compute_expected_costs_to_goals, PTO::extract_policy */
struct orc_bg_view { size_t B, NB; uint32_t nw; const double *beliefs; const uint8_t *types; const uint64_t *coff, *poff; const uint32_t *cid, *pid; };
int orc_bg_view_get(const orc_ctx *c, struct orc_bg_view *v);      /* belief.c */

static int bg_arrays(const orc_ctx *c, struct orc_bg_view *v, double **xy, uint32_t **bid) {
    if (orc_bg_view_get(c, v)) return -1;
    *xy = malloc(v->NB * 2 * sizeof(double));
    *bid = malloc(v->NB * sizeof(uint32_t));
    for (size_t i = 0; i < v->NB; ++i) {
        (*xy)[2 * i] = c->nx[i / v->B]; (*xy)[2 * i + 1] = c->ny[i / v->B];
        (*bid)[i] = (uint32_t)(i % v->B);
    }
    return 0;
}

int orc_bg_expected_costs(const orc_ctx *c, double *dist) {
    struct orc_bg_view v;
    double *xy; uint32_t *bid;
    if (bg_arrays(c, &v, &xy, &bid)) return -1;
    /* pto.rs:263-271: final belief nodes = belief nodes of the final graph nodes whose belief is compatible with the finality;
     * node_to_belief_nodes[final_id][b] is Some only where the belief is compatible with the node's validity */
    uint64_t *finals = malloc((c->n_final * v.B + 1) * sizeof(uint64_t)), nf = 0;
    for (uint64_t k = 0; k < c->n_final; ++k) {
        const uint64_t id = c->final_ids[k];
        for (size_t b = 0; b < v.B; ++b) {
            int some = 1, compat = 1;
            for (uint32_t w = 0; w < v.nw; ++w) {
                const double p = v.beliefs[b * v.nw + w];
                if (p > 0.0 && !((c->validities[c->node_validity[id]] >> w) & 1)) some = 0;
                if (p > 0.0 && !((c->final_masks[k] >> w) & 1)) compat = 0;
            }
            if (some && compat) finals[nf++] = id * v.B + b;
        }
    }
    int rc = orc_conditional_dijkstra(v.NB, xy, bid, v.beliefs, v.nw, v.types, v.coff, v.cid, v.poff, v.pid, finals, nf, dist);
    free(finals); free(xy); free(bid);
    return rc;
}

int64_t orc_bg_extract_policy(const orc_ctx *c, const double *dist, uint64_t *original_id, int64_t *parent, uint8_t *is_leaf, uint64_t cap) {
    struct orc_bg_view v;
    double *xy; uint32_t *bid;
    if (bg_arrays(c, &v, &xy, &bid)) return -1;
    int64_t r = orc_extract_policy(v.NB, xy, bid, bid, v.beliefs, v.nw, v.coff, v.cid, dist, original_id, parent, is_leaf, cap);
    free(xy); free(bid);
    return r;
}
