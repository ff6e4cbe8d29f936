/*
 * grow.c -- ORACLE (test infrastructure only): the growth loops of the reference.
 *
 *   rrt_seq      RRT::grow_tree             src/rrt.rs:102-174 (RRT* with rewire)
 *   pto_seq      PTO::grow_graph            src/pto.rs:55-139  (belief-space RRG)
 *   *_batched    the batched contract of the GPU engine (no reference line: the
 *                reference is strictly sequential).  Definition:
 *     - a step takes the next nb = min(K, iterations left before the loop
 *       condition has to be looked at again) iterations; their samples are drawn
 *       in iteration order (goal bias on the global iteration index, rrt.rs:176-181);
 *     - every sample of the step is evaluated against the SNAPSHOT of the tree at
 *       step start: nearest neighbour = lexicographic min of (norm2, id), radius
 *       = heuristic_radius(snapshot size [+1 for PTO, pto.rs:88]), neighbour set
 *       in ascending id, best parent = minimum of dist_root+cost, equal costs
 *       resolved in kd-tree pre-order of the snapshot tree (that is the order
 *       the reference's radius search lists neighbours in, so Iterator::min_by
 *       keeps exactly that one, rrt.rs:123,143-145; such ties are systematic:
 *       every 100th iteration re-adds the goal point, rrt.rs:176-181);
 *     - valid samples are committed in sample order (ids = snapshot size + rank);
 *       rewires are applied sequentially in that order with the reference's
 *       strict `<` against the current dist_root of the target;
 *     - PTO: reach[new] = OR over edges of reach_snapshot[nbr] & validity, then
 *       reach[nbr] |= reach[new] & validity (pto.rs:111-120 order).
 *     With K = 1 this is the reference loop itself; tests assert
 *     batched(K=1) == seq bit for bit.
 *   ORC_ALGO_BATCHED_KD (2) computes the same batched result with the kd-tree as
 *   a candidate generator (exact: the radius set is a set, and the NN tie-break is
 *   re-done over all nodes at the minimal distance) so that 100k-node cases
 *   finish in seconds; tests assert it equals the brute-force definition.
 */
#include "orc_internal.h"
#include <math.h>

static void reset_outputs(orc_ctx *c, int mode) {
    c->mode = mode;
    c->n_nodes = 0;
    c->n_iter = 0;
    c->n_final = 0;
    c->n_edges = 0;
    c->complete = 0;
    c->oob = 0;
    c->err[0] = 0;
}

static uint64_t add_node(orc_ctx *c, const double s[2], int64_t parent, double dist, uint64_t reach, uint32_t vid) {
    orc_ctx_reserve_nodes(c, c->n_nodes + 1);
    uint64_t id = c->n_nodes++;
    c->nx[id] = s[0];
    c->ny[id] = s[1];
    c->parent[id] = parent;
    c->dist[id] = dist;
    c->reach[id] = reach;
    c->node_validity[id] = vid;
    return id;
}

static void node_state(const orc_ctx *c, uint64_t id, double s[2]) {
    s[0] = c->nx[id];
    s[1] = c->ny[id];
}

/* rrt.rs:176-181 */
static int rrt_sample(orc_ctx *c, uint64_t iteration, double s[2]) {
    if (iteration % 100 == 0) return orc_goal_example(c, 0, s);
    return orc_sample(c, s);
}

typedef struct {
    uint64_t *v;
    size_t n, cap;
} idvec;
static void idvec_push(idvec *a, uint64_t x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? 2 * a->cap : 256;
        a->v = (uint64_t *)realloc(a->v, a->cap * sizeof(uint64_t));
    }
    a->v[a->n++] = x;
}

/* all ids within `radius` by the kd-tree, unbounded */
static void kd_radius_all(const orc_kdtree *kd, const double q[2], double radius, idvec *out) {
    out->n = 0;
    if (out->cap == 0) {
        out->cap = 256;
        out->v = (uint64_t *)malloc(out->cap * sizeof(uint64_t));
    }
    size_t need = orc_kd_radius(kd, q, radius, out->v, out->cap);
    if (need > out->cap) {
        out->cap = need;
        out->v = (uint64_t *)realloc(out->v, out->cap * sizeof(uint64_t));
        need = orc_kd_radius(kd, q, radius, out->v, out->cap);
    }
    out->n = need;
}

/* ======================================================== RRT*, sequential */
static int rrt_seq(orc_ctx *c, const double start[2], double max_step, double search_radius,
                   uint64_t n_iter_min, uint64_t n_iter_max) {
    reset_outputs(c, ORC_MODE_RRT);
    add_node(c, start, -1, 0.0, 0, 0);         /* rrt.rs:105 */
    orc_kdtree *kd = orc_kd_new(start, 0);     /* rrt.rs:106 */
    idvec nb = {0}, nbv = {0};
    int rc = 0;

    uint64_t i = 0;
    while (i < n_iter_min || (c->n_final == 0 && i < n_iter_max)) { /* rrt.rs:109 */
        i += 1;
        double ns[2];
        if (rrt_sample(c, i, ns)) { rc = -1; break; }            /* rrt.rs:112 */
        uint64_t kd_from = orc_kd_nearest(kd, ns, NULL, 0);      /* rrt.rs:113 */
        double from[2];
        node_state(c, kd_from, from);
        orc_steer(from, ns, max_step);                           /* rrt.rs:115 */
        if (!orc_rrt_state_valid(c, ns)) continue;               /* rrt.rs:117 */

        double radius = orc_heuristic_radius(c->n_nodes, max_step, search_radius, 2); /* rrt.rs:121 */
        kd_radius_all(kd, ns, radius, &nb);                      /* rrt.rs:123 */
        nbv.n = 0;
        for (size_t a = 0; a < nb.n; ++a) {                      /* rrt.rs:124 */
            double s[2];
            node_state(c, nb.v[a], s);
            if (orc_rrt_transition_valid(c, s, ns)) idvec_push(&nbv, nb.v[a]);
        }
        if (nbv.n == 0) idvec_push(&nbv, kd_from);               /* rrt.rs:132-134 */

        /* rrt.rs:137-145: first minimum of dist_root + cost */
        uint64_t best = nbv.v[0];
        double best_total = 0.0, best_cost = 0.0;
        for (size_t a = 0; a < nbv.n; ++a) {
            double s[2];
            node_state(c, nbv.v[a], s);
            double cost = orc_norm2(s, ns);
            double total = c->dist[nbv.v[a]] + cost;
            if (a == 0 || total < best_total) {
                best = nbv.v[a];
                best_total = total;
                best_cost = cost;
            }
        }
        uint64_t id = add_node(c, ns, (int64_t)best, c->dist[best] + best_cost, 0, 0); /* rrt.rs:148, 30-37 */
        double new_dist = c->dist[id];

        for (size_t a = 0; a < nbv.n; ++a) {                     /* rrt.rs:152-161 */
            uint64_t j = nbv.v[a];
            if (j == best) continue;
            double s[2];
            node_state(c, j, s);
            double d = orc_norm2(ns, s);
            double via = new_dist + d;
            if (via < c->dist[j]) {                              /* rrt.rs:40-46 */
                c->parent[j] = (int64_t)id;
                c->dist[j] = new_dist + d;
            }
        }
        orc_kd_add(kd, ns, id);                                  /* rrt.rs:163 */
        uint64_t mask = 0;
        if (orc_goal(c, ns, &mask)) orc_ctx_push_final(c, id, mask); /* rrt.rs:165-167 */
    }
    c->n_iter = i;
    orc_kd_free(kd);
    free(nb.v);
    free(nbv.v);
    return rc;
}

/* ----------------------------------------- snapshot neighbour search helpers */
/* lexicographic min of (norm2(node_j, q), j) over j < n_snap that pass the
 * world filter; returns 0 (the root, nearest_neighbor.rs:90) when none passes */
static uint64_t nn_brute(const orc_ctx *c, uint64_t n_snap, const double q[2], const uint64_t *reach, uint32_t world) {
    double dmin = INFINITY;
    uint64_t best = 0;
    for (uint64_t j = 0; j < n_snap; ++j) {
        double s[2] = {c->nx[j], c->ny[j]};
        double d = orc_norm2(s, q);
        if (d < dmin && (!reach || ((reach[j] >> world) & 1))) {
            dmin = d;
            best = j;
        }
    }
    return best;
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static uint64_t nn_kd(const orc_ctx *c, const orc_kdtree *kd, const double q[2], const uint64_t *reach, uint32_t world, idvec *tmp) {
    uint64_t cand = orc_kd_nearest(kd, q, reach, world);
    if (reach && !((reach[cand] >> world) & 1)) return 0; /* nothing passed: root */
    double s[2] = {c->nx[cand], c->ny[cand]};
    double dstar = orc_norm2(s, q);
    /* every node at exactly the minimal rounded distance; lowest passing id wins */
    kd_radius_all(kd, q, dstar, tmp);
    uint64_t best = cand;
    for (size_t a = 0; a < tmp->n; ++a) {
        uint64_t j = tmp->v[a];
        double t[2] = {c->nx[j], c->ny[j]};
        if (orc_norm2(t, q) == dstar && j < best && (!reach || ((reach[j] >> world) & 1))) best = j;
    }
    return best;
}

static void radius_brute(const orc_ctx *c, uint64_t n_snap, const double q[2], double radius, idvec *out) {
    out->n = 0;
    for (uint64_t j = 0; j < n_snap; ++j) {
        double s[2] = {c->nx[j], c->ny[j]};
        if (orc_norm2(s, q) <= radius) idvec_push(out, j);
    }
}

typedef struct {
    double ns[2];
    int valid;
    int vid;          /* PTO validity id */
    uint32_t world;   /* PTO sampled world */
    uint64_t nn;
    uint64_t best;
    double best_cost;
    size_t nb_off, nb_n; /* validated neighbours in the step's pool */
} step_item;

/* ============================================================ RRT*, batched */
static int rrt_batched(orc_ctx *c, const double start[2], double max_step, double search_radius,
                       uint64_t n_iter_min, uint64_t n_iter_max, uint32_t K, int use_kd) {
    if (K == 0) {
        snprintf(c->err, sizeof c->err, "batch_K must be >= 1");
        return -1;
    }
    reset_outputs(c, ORC_MODE_RRT);
    add_node(c, start, -1, 0.0, 0, 0);
    orc_kdtree *kd = orc_kd_new(start, 0); /* always kept: it defines the tie order */
    step_item *items = (step_item *)malloc(sizeof(step_item) * K);
    idvec pool = {0}, pool_tv = {0}, nb = {0}, tmp = {0};
    double *dist_snap = NULL;
    size_t dist_cap = 0;
    int rc = 0;

    uint64_t i = 0;
    while (i < n_iter_min || (c->n_final == 0 && i < n_iter_max)) {
        uint64_t limit = i < n_iter_min ? n_iter_min : n_iter_max;
        uint64_t nbatch = limit - i < K ? limit - i : K;
        uint64_t n_snap = c->n_nodes;
        if (dist_cap < n_snap) {
            dist_cap = 2 * n_snap + 1024;
            dist_snap = (double *)realloc(dist_snap, dist_cap * sizeof(double));
        }
        memcpy(dist_snap, c->dist, n_snap * sizeof(double));
        double radius = orc_heuristic_radius(n_snap, max_step, search_radius, 2);

        /* samples in iteration order */
        for (uint64_t k = 0; k < nbatch && !rc; ++k)
            if (rrt_sample(c, i + k + 1, items[k].ns)) rc = -1;
        if (rc) break;

        /* evaluate every sample against the snapshot */
        pool.n = 0;
        for (uint64_t k = 0; k < nbatch; ++k) {
            step_item *it = &items[k];
            it->nn = use_kd ? nn_kd(c, kd, it->ns, NULL, 0, &tmp) : nn_brute(c, n_snap, it->ns, NULL, 0);
            double from[2];
            node_state(c, it->nn, from);
            orc_steer(from, it->ns, max_step);
            it->valid = orc_rrt_state_valid(c, it->ns);
            it->nb_off = pool.n;
            it->nb_n = 0;
            if (!it->valid) continue;
            if (use_kd) {
                kd_radius_all(kd, it->ns, radius, &nb);
                qsort(nb.v, nb.n, sizeof(uint64_t), cmp_u64);
            } else {
                radius_brute(c, n_snap, it->ns, radius, &nb);
            }
            for (size_t a = 0; a < nb.n; ++a) {
                double s[2];
                node_state(c, nb.v[a], s);
                if (orc_rrt_transition_valid(c, s, it->ns)) idvec_push(&pool, nb.v[a]);
            }
            if (pool.n == it->nb_off) idvec_push(&pool, it->nn);
            it->nb_n = pool.n - it->nb_off;
            for (size_t a = 0; a < it->nb_n; ++a) {
                uint64_t j = pool.v[it->nb_off + a];
                double s[2];
                node_state(c, j, s);
                double cost = orc_norm2(s, it->ns);
                double total = dist_snap[j] + cost;
                double best_total = a ? dist_snap[it->best] + it->best_cost : 0.0;
                if (a == 0 || total < best_total || (total == best_total && orc_kd_preorder_less(kd, j, it->best))) {
                    it->best = j;
                    it->best_cost = cost;
                }
            }
        }

        /* commit in sample order */
        for (uint64_t k = 0; k < nbatch; ++k) {
            step_item *it = &items[k];
            if (!it->valid) continue;
            uint64_t id = add_node(c, it->ns, (int64_t)it->best, dist_snap[it->best] + it->best_cost, 0, 0);
            double new_dist = c->dist[id];
            for (size_t a = 0; a < it->nb_n; ++a) {
                uint64_t j = pool.v[it->nb_off + a];
                if (j == it->best) continue;
                double s[2];
                node_state(c, j, s);
                double d = orc_norm2(it->ns, s);
                double via = new_dist + d;
                if (via < c->dist[j]) {
                    c->parent[j] = (int64_t)id;
                    c->dist[j] = via;
                }
            }
            orc_kd_add(kd, it->ns, id);
            uint64_t mask = 0;
            if (orc_goal(c, it->ns, &mask)) orc_ctx_push_final(c, id, mask);
        }
        i += nbatch;
    }
    c->n_iter = i;
    orc_kd_free(kd);
    free(items);
    free(pool.v);
    free(pool_tv.v);
    free(nb.v);
    free(tmp.v);
    free(dist_snap);
    return rc;
}

/* ===================================================== PTO RRG, sequential */
static int final_set_complete(const orc_ctx *c) {
    /* stateless form of pto_reachability.rs:81-101: reach only grows, so the
     * lazily refreshed `finality` equals this OR at every call */
    if (c->n_final == 0) return 0;
    uint64_t fin = 0;
    for (uint64_t k = 0; k < c->n_final; ++k) fin |= c->reach[c->final_ids[k]] & c->final_masks[k];
    uint64_t all = c->n_worlds >= 64 ? ~0ULL : ((1ULL << c->n_worlds) - 1);
    return (fin & all) == all;
}

/* pto.rs:141-149 */
static int pto_sample(orc_ctx *c, uint64_t iteration, uint32_t *world, double s[2]) {
    *world = (uint32_t)orc_sample_discrete(c, (uint64_t)c->n_worlds);
    if (c->err[0]) return -1;
    if (iteration % 100 == 0) return orc_goal_example(c, *world, s);
    return orc_sample(c, s);
}

static int pto_seq(orc_ctx *c, const double start[2], double max_step, double search_radius,
                   uint64_t n_iter_min, uint64_t n_iter_max) {
    reset_outputs(c, ORC_MODE_PTO);
    if (!c->has_grid) {
        snprintf(c->err, sizeof c->err, "PTO mode needs a grid");
        return -1;
    }
    int root_vid = orc_state_validity(c, start);   /* pto.rs:61 */
    if (root_vid < 0) {
        snprintf(c->err, sizeof c->err, "Start from a valid state!");
        return -2;
    }
    orc_reach *reach = orc_reach_new();
    add_node(c, start, -1, 0.0, 0, (uint32_t)root_vid);               /* pto.rs:62 */
    orc_reach_set_root(reach, c->validities[root_vid], (uint32_t)c->n_worlds); /* pto.rs:63 */
    c->reach[0] = orc_reach_get(reach, 0);
    orc_kdtree *kd = orc_kd_new(start, 0);                             /* pto.rs:64 */
    idvec nb = {0};
    int rc = 0;

    uint64_t i = 0;
    for (;;) {
        /* pto.rs:67: the completeness test is only evaluated once i >= n_iter_min */
        if (!(i < n_iter_min || (!orc_reach_is_final_set_complete(reach) && i < n_iter_max))) break;
        i += 1;
        uint32_t world;
        double ns[2];
        if (pto_sample(c, i, &world, ns)) { rc = -1; break; }          /* pto.rs:71 */
        uint64_t kd_from = orc_kd_nearest(kd, ns, c->reach, world);    /* pto.rs:74-77 */
        double from[2];
        node_state(c, kd_from, from);
        orc_steer(from, ns, max_step);                                 /* pto.rs:79 */
        int vid = orc_state_validity(c, ns);                           /* pto.rs:81 */
        if (vid < 0) continue;
        uint64_t id = add_node(c, ns, -1, 0.0, 0, (uint32_t)vid);      /* pto.rs:83 */
        orc_reach_add_node(reach, c->validities[vid]);                 /* pto.rs:85 */
        double radius = orc_heuristic_radius(c->n_nodes, max_step, search_radius, 2); /* pto.rs:88 */
        kd_radius_all(kd, ns, radius, &nb);                            /* pto.rs:95-97 */
        if (nb.n == 0) idvec_push(&nb, kd_from);                       /* pto.rs:99 */
        uint64_t e0 = c->n_edges;
        for (size_t a = 0; a < nb.n; ++a) {                            /* pto.rs:103-108 */
            double s[2];
            node_state(c, nb.v[a], s);
            int tv = orc_transition_validity(c, s, ns);
            if (tv >= 0) orc_ctx_push_edge(c, (uint32_t)nb.v[a], (uint32_t)id, (uint32_t)tv);
        }
        for (uint64_t e = e0; e < c->n_edges; ++e) {                   /* pto.rs:111-114 */
            orc_reach_add_edge(reach, c->edges[e].from, id, c->validities[c->edges[e].validity_id]);
        }
        for (uint64_t e = e0; e < c->n_edges; ++e) {                   /* pto.rs:117-120 */
            orc_reach_add_edge(reach, id, c->edges[e].from, c->validities[c->edges[e].validity_id]);
            c->reach[c->edges[e].from] = orc_reach_get(reach, c->edges[e].from);
        }
        c->reach[id] = orc_reach_get(reach, id);
        uint64_t fin = 0;
        if (orc_goal(c, ns, &fin)) {                                   /* pto.rs:122-124 */
            orc_reach_add_final_node(reach, id, fin);
            orc_ctx_push_final(c, id, fin);
        }
        orc_kd_add(kd, ns, id);                                        /* pto.rs:126 */
    }
    c->n_iter = i;
    c->complete = orc_reach_is_final_set_complete(reach);              /* pto.rs:134 */
    for (uint64_t j = 0; j < c->n_nodes; ++j) c->reach[j] = orc_reach_get(reach, j);
    orc_reach_free(reach);
    orc_kd_free(kd);
    free(nb.v);
    if (rc) return rc;
    return c->complete ? 0 : 1;
}

/* ========================================================= PTO RRG, batched */
static int pto_batched(orc_ctx *c, const double start[2], double max_step, double search_radius,
                       uint64_t n_iter_min, uint64_t n_iter_max, uint32_t K, int use_kd) {
    if (K == 0) {
        snprintf(c->err, sizeof c->err, "batch_K must be >= 1");
        return -1;
    }
    reset_outputs(c, ORC_MODE_PTO);
    if (!c->has_grid) {
        snprintf(c->err, sizeof c->err, "PTO mode needs a grid");
        return -1;
    }
    int root_vid = orc_state_validity(c, start);
    if (root_vid < 0) {
        snprintf(c->err, sizeof c->err, "Start from a valid state!");
        return -2;
    }
    add_node(c, start, -1, 0.0, c->validities[root_vid], (uint32_t)root_vid);
    orc_kdtree *kd = use_kd ? orc_kd_new(start, 0) : NULL;
    step_item *items = (step_item *)malloc(sizeof(step_item) * K);
    idvec pool = {0}, pool_tv = {0}, nb = {0}, tmp = {0};
    uint64_t *reach_snap = NULL;
    size_t snap_cap = 0;
    int rc = 0;

    uint64_t i = 0;
    while (i < n_iter_min || (!final_set_complete(c) && i < n_iter_max)) {
        uint64_t limit = i < n_iter_min ? n_iter_min : n_iter_max;
        uint64_t nbatch = limit - i < K ? limit - i : K;
        uint64_t n_snap = c->n_nodes;
        if (snap_cap < n_snap) {
            snap_cap = 2 * n_snap + 1024;
            reach_snap = (uint64_t *)realloc(reach_snap, snap_cap * sizeof(uint64_t));
        }
        memcpy(reach_snap, c->reach, n_snap * sizeof(uint64_t));
        double radius = orc_heuristic_radius(n_snap + 1, max_step, search_radius, 2);

        for (uint64_t k = 0; k < nbatch && !rc; ++k)
            if (pto_sample(c, i + k + 1, &items[k].world, items[k].ns)) rc = -1;
        if (rc) break;

        pool.n = 0;
        pool_tv.n = 0;
        for (uint64_t k = 0; k < nbatch; ++k) {
            step_item *it = &items[k];
            it->nn = use_kd ? nn_kd(c, kd, it->ns, reach_snap, it->world, &tmp)
                            : nn_brute(c, n_snap, it->ns, reach_snap, it->world);
            double from[2];
            node_state(c, it->nn, from);
            orc_steer(from, it->ns, max_step);
            it->vid = orc_state_validity(c, it->ns);
            it->valid = it->vid >= 0;
            it->nb_off = pool.n;
            it->nb_n = 0;
            if (!it->valid) continue;
            if (use_kd) {
                kd_radius_all(kd, it->ns, radius, &nb);
                qsort(nb.v, nb.n, sizeof(uint64_t), cmp_u64);
            } else {
                radius_brute(c, n_snap, it->ns, radius, &nb);
            }
            if (nb.n == 0) idvec_push(&nb, it->nn);
            for (size_t a = 0; a < nb.n; ++a) {
                double s[2];
                node_state(c, nb.v[a], s);
                int tv = orc_transition_validity(c, s, it->ns);
                if (tv >= 0) {
                    idvec_push(&pool, nb.v[a]);
                    idvec_push(&pool_tv, (uint64_t)tv);
                }
            }
            it->nb_n = pool.n - it->nb_off;
        }

        /* commit: nodes and forward edges in sample order, reach phase 1 from the
         * snapshot, then phase 2 */
        uint64_t first_new = c->n_nodes;
        for (uint64_t k = 0; k < nbatch; ++k) {
            step_item *it = &items[k];
            if (!it->valid) continue;
            uint64_t r = 0;
            for (size_t a = 0; a < it->nb_n; ++a)
                r |= reach_snap[pool.v[it->nb_off + a]] & c->validities[pool_tv.v[it->nb_off + a]];
            uint64_t id = add_node(c, it->ns, -1, 0.0, r, (uint32_t)it->vid);
            for (size_t a = 0; a < it->nb_n; ++a)
                orc_ctx_push_edge(c, (uint32_t)pool.v[it->nb_off + a], (uint32_t)id, (uint32_t)pool_tv.v[it->nb_off + a]);
            uint64_t fin = 0;
            if (orc_goal(c, it->ns, &fin)) orc_ctx_push_final(c, id, fin);
            if (kd) orc_kd_add(kd, it->ns, id);
        }
        uint64_t id = first_new;
        for (uint64_t k = 0; k < nbatch; ++k) {
            step_item *it = &items[k];
            if (!it->valid) continue;
            for (size_t a = 0; a < it->nb_n; ++a)
                c->reach[pool.v[it->nb_off + a]] |= c->reach[id] & c->validities[pool_tv.v[it->nb_off + a]];
            ++id;
        }
        i += nbatch;
    }
    c->n_iter = i;
    c->complete = final_set_complete(c);
    orc_kd_free(kd);
    free(items);
    free(pool.v);
    free(pool_tv.v);
    free(nb.v);
    free(tmp.v);
    free(reach_snap);
    if (rc) return rc;
    return c->complete ? 0 : 1;
}

int orc_grow(orc_ctx *c, const double start[2], double max_step, double search_radius,
             uint64_t n_iter_min, uint64_t n_iter_max, uint32_t batch_K, int mode, int algo) {
    int rc;
    orc_bg_release(c);
    if (mode == ORC_MODE_RRT) {
        if (algo == ORC_ALGO_SEQ) rc = rrt_seq(c, start, max_step, search_radius, n_iter_min, n_iter_max);
        else rc = rrt_batched(c, start, max_step, search_radius, n_iter_min, n_iter_max, batch_K, algo == ORC_ALGO_BATCHED_KD);
    } else if (mode == ORC_MODE_PTO) {
        if (algo == ORC_ALGO_SEQ) rc = pto_seq(c, start, max_step, search_radius, n_iter_min, n_iter_max);
        else rc = pto_batched(c, start, max_step, search_radius, n_iter_min, n_iter_max, batch_K, algo == ORC_ALGO_BATCHED_KD);
    } else {
        snprintf(c->err, sizeof c->err, "bad mode");
        return -1;
    }
    if (rc >= 0 && c->oob) {
        snprintf(c->err, sizeof c->err, "raster access outside the map (the reference panics here)");
        return -3;
    }
    return rc;
}

/* ------------------------------------------------------------------ getters */
uint64_t orc_num_nodes(const orc_ctx *c) { return c->n_nodes; }
uint64_t orc_num_iterations(const orc_ctx *c) { return c->n_iter; }

int orc_get_tree(const orc_ctx *c, double *xy, int64_t *parent, double *dist_root) {
    for (uint64_t j = 0; j < c->n_nodes; ++j) {
        if (xy) {
            xy[2 * j] = c->nx[j];
            xy[2 * j + 1] = c->ny[j];
        }
        if (parent) parent[j] = c->parent[j];
        if (dist_root) dist_root[j] = c->dist[j];
    }
    return 0;
}

uint64_t orc_num_final(const orc_ctx *c) { return c->n_final; }
int orc_get_final_ids(const orc_ctx *c, uint64_t *ids) {
    memcpy(ids, c->final_ids, c->n_final * sizeof(uint64_t));
    return 0;
}
int orc_get_final_masks(const orc_ctx *c, uint64_t *masks) {
    memcpy(masks, c->final_masks, c->n_final * sizeof(uint64_t));
    return 0;
}
int orc_get_reach(const orc_ctx *c, uint64_t *masks) {
    memcpy(masks, c->reach, c->n_nodes * sizeof(uint64_t));
    return 0;
}
int orc_get_node_validity(const orc_ctx *c, uint32_t *v) {
    memcpy(v, c->node_validity, c->n_nodes * sizeof(uint32_t));
    return 0;
}
uint64_t orc_num_edges(const orc_ctx *c) { return c->n_edges; }
int orc_get_edges(const orc_ctx *c, uint32_t *from, uint32_t *to, uint32_t *validity_id) {
    for (uint64_t e = 0; e < c->n_edges; ++e) {
        from[e] = c->edges[e].from;
        to[e] = c->edges[e].to;
        validity_id[e] = c->edges[e].validity_id;
    }
    return 0;
}
int orc_is_final_set_complete(const orc_ctx *c) { return c->mode == ORC_MODE_PTO ? c->complete : (c->n_final > 0); }

/* rrt.rs:48-61 get_path_to, 223-227 get_path_cost, 183-193 get_best_solution:
 * the FIRST final node of minimal path cost wins (Iterator::min_by) */
uint64_t orc_best_solution(const orc_ctx *c, double *path_xy, uint64_t cap, double *cost) {
    if (c->n_final == 0) return 0; /* Err("No solution found") */
    uint64_t best_len = 0, best_id = 0;
    double best_cost = 0.0;
    int have = 0;
    for (uint64_t k = 0; k < c->n_final; ++k) {
        uint64_t id = c->final_ids[k];
        /* path root -> id; cost summed over consecutive pairs from the root */
        uint64_t len = 1;
        for (int64_t p = c->parent[id]; p >= 0; p = c->parent[p]) ++len;
        uint64_t *ids = (uint64_t *)malloc(len * sizeof(uint64_t));
        uint64_t pos = len;
        ids[--pos] = id;
        for (int64_t p = c->parent[id]; p >= 0; p = c->parent[p]) ids[--pos] = (uint64_t)p;
        double sum = 0.0;
        for (uint64_t a = 0; a + 1 < len; ++a) {
            double s0[2] = {c->nx[ids[a]], c->ny[ids[a]]}, s1[2] = {c->nx[ids[a + 1]], c->ny[ids[a + 1]]};
            sum += orc_norm2(s0, s1);
        }
        free(ids);
        if (!have || sum < best_cost) {
            have = 1;
            best_cost = sum;
            best_id = id;
            best_len = len;
        }
    }
    if (cost) *cost = best_cost;
    if (path_xy && cap >= best_len) {
        uint64_t pos = best_len;
        int64_t p = (int64_t)best_id;
        while (p >= 0) {
            --pos;
            path_xy[2 * pos] = c->nx[p];
            path_xy[2 * pos + 1] = c->ny[p];
            p = c->parent[p];
        }
    }
    return best_len;
}

/* rrt.rs:229-246: walk up while the parent is itself final (root counts as
 * parent id 0 through unwrap_or(0)); the set is returned in ascending id here */
uint64_t orc_firstly_final_ids(const orc_ctx *c, uint64_t *ids, uint64_t cap) {
    if (c->n_nodes == 0) return 0;
    uint8_t *is_final = (uint8_t *)calloc(c->n_nodes, 1), *is_first = (uint8_t *)calloc(c->n_nodes, 1);
    for (uint64_t k = 0; k < c->n_final; ++k) is_final[c->final_ids[k]] = 1;
    for (uint64_t k = 0; k < c->n_final; ++k) {
        uint64_t first = c->final_ids[k];
        for (;;) {
            int64_t p = c->parent[first];
            uint64_t pid = p < 0 ? 0 : (uint64_t)p;
            if (!is_final[pid]) break;
            if (p < 0) break; /* unwrap() on the root would panic in the reference */
            first = (uint64_t)p;
        }
        is_first[first] = 1;
    }
    uint64_t n = 0;
    for (uint64_t j = 0; j < c->n_nodes; ++j)
        if (is_first[j]) {
            if (n < cap && ids) ids[n] = j;
            ++n;
        }
    free(is_final);
    free(is_first);
    return n;
}

/* PRM::init + PRM::grow_graph (prm.rs:33-109): every sample is a node (validity id 0, no steering, no state check),
 * connected both ways to the kd-tree's radius neighbours whose transition is valid.  Edges are logged neighbour -> new
 * with what transition_validator returned (the graph itself stores validity 0 for both directions, prm.rs:96-103). */
int orc_prm_grow(orc_ctx *c, const double start[2], double max_step, double search_radius, uint64_t n_iter) {
    orc_bg_release(c);
    reset_outputs(c, ORC_MODE_PTO);
    c->oob = 0;
    if (!c->has_grid) {
        snprintf(c->err, sizeof c->err, "PRM needs a grid");
        return -1;
    }
    add_node(c, start, -1, 0.0, 0, 0);                                 /* prm.rs:33-36 */
    orc_kdtree *kd = orc_kd_new(start, 0);
    idvec nb = {0};
    int rc = 0;
    for (uint64_t i = 0; i < n_iter; ++i) {                            /* prm.rs:38-51 */
        double ns[2];
        if (orc_sample(c, ns)) { rc = -1; break; }                     /* continuous_sampler.sample() */
        uint64_t id = add_node(c, ns, -1, 0.0, 0, 0);                  /* prm.rs:62 */
        double radius = orc_heuristic_radius(c->n_nodes, max_step, search_radius, 2);   /* prm.rs:66 */
        kd_radius_all(kd, ns, radius, &nb);                            /* prm.rs:73-75 */
        orc_kd_add(kd, ns, id);                                        /* prm.rs:77 */
        for (size_t a = 0; a < nb.n; ++a) {                            /* prm.rs:90-95 */
            double s[2];
            node_state(c, nb.v[a], s);
            int tv = orc_transition_validity(c, s, ns);
            if (c->oob) { rc = -1; snprintf(c->err, sizeof c->err, "raster access the reference would panic on"); break; }
            if (tv >= 0) orc_ctx_push_edge(c, (uint32_t)nb.v[a], (uint32_t)id, (uint32_t)tv);
        }
        if (rc) break;
    }
    orc_kd_free(kd);
    free(nb.v);
    c->n_iter = n_iter;
    return rc;
}

/* PRM::plan_path (prm.rs:111-123): nearest roadmap nodes of start and goal, dijkstra from the goal (pto_graph.rs:275-303),
 * extract_path (pto_graph.rs:305-326: from the start, always to the first parent of least cost-to-goal + edge).
 * Works on the roadmap of the last orc_prm_grow.  Returns the number of states written to path_xy (0 = no path: the
 * reference returns an empty Vec), -1 on error, the needed count if cap is too small. */
int64_t orc_prm_plan_path(orc_ctx *c, const double start[2], const double goal[2], double *path_xy, uint64_t cap) {
    const uint64_t N = c->n_nodes;
    if (!N) return -1;
    const double root[2] = {c->nx[0], c->ny[0]};
    orc_kdtree *kd = orc_kd_new(root, 0);
    for (uint64_t i = 1; i < N; ++i) { const double s[2] = {c->nx[i], c->ny[i]}; orc_kd_add(kd, s, i); }
    const uint64_t kd_start = orc_kd_nearest(kd, start, NULL, 0), kd_goal = orc_kd_nearest(kd, goal, NULL, 0);
    orc_kd_free(kd);
    /* PTOGraph parents lists in push order (prm.rs:96-103): for one new node, add_edge(nbr, new) for all, then add_edge(new, nbr) */
    uint64_t *off = calloc(N + 1, sizeof(uint64_t));
    for (uint64_t e = 0; e < c->n_edges; ++e) { off[c->edges[e].from + 1]++; off[c->edges[e].to + 1]++; }
    for (uint64_t i = 0; i < N; ++i) off[i + 1] += off[i];
    uint32_t *par = malloc((2 * c->n_edges + 1) * sizeof(uint32_t));
    uint64_t *fill = malloc((N + 1) * sizeof(uint64_t));
    memcpy(fill, off, (N + 1) * sizeof(uint64_t));
    for (uint64_t e = 0; e < c->n_edges;) {
        uint64_t e1 = e;
        while (e1 < c->n_edges && c->edges[e1].to == c->edges[e].to) ++e1;
        for (uint64_t k = e; k < e1; ++k) par[fill[c->edges[k].to]++] = c->edges[k].from;      /* parents of new: its neighbours */
        for (uint64_t k = e; k < e1; ++k) par[fill[c->edges[k].from]++] = c->edges[k].to;      /* parents of each neighbour: new */
        e = e1;
    }
    free(fill);
    double *xy = malloc(N * 2 * sizeof(double)), *dist = malloc(N * sizeof(double));
    uint32_t *row = calloc(N, sizeof(uint32_t));
    uint8_t *types = malloc(N);
    for (uint64_t i = 0; i < N; ++i) { xy[2 * i] = c->nx[i]; xy[2 * i + 1] = c->ny[i]; types[i] = 1; }
    const double one = 1.0;
    const uint64_t fin = kd_goal;
    /* dijkstra == conditional_dijkstra without observation nodes; the adjacency is symmetric, children == parents */
    int rc = orc_conditional_dijkstra(N, xy, row, &one, 1, types, off, par, off, par, &fin, 1, dist);
    int64_t n_path = -1;
    if (!rc) {
        if (isinf(dist[kd_start])) n_path = 0;                          /* prm.rs:117-119 */
        else {
            uint64_t node = kd_start;
            n_path = 0;
            for (uint64_t guard = 0; guard <= N; ++guard) {
                if ((uint64_t)n_path < cap) { path_xy[2 * n_path] = c->nx[node]; path_xy[2 * n_path + 1] = c->ny[node]; }
                ++n_path;
                if (dist[node] == 0.0) break;
                uint64_t best = 0;
                double best_cost = 0.0;
                int have = 0;
                for (uint64_t k = off[node]; k < off[node + 1]; ++k) {
                    const uint64_t p = par[k];
                    const double cost = dist[p] + orc_norm2(xy + 2 * p, xy + 2 * node);
                    if (!have || cost < best_cost) { best = p; best_cost = cost; have = 1; }     /* min_by: the first minimum */
                }
                if (!have) { n_path = -1; break; }
                node = best;
                if (guard == N) n_path = -1;                            /* zero-length cycle: the reference would not terminate */
            }
        }
    }
    free(off); free(par); free(xy); free(dist); free(row); free(types);
    return n_path;
}
