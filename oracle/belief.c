/*
 * belief.c -- ORACLE (test infrastructure only): the belief-space expansion
 * PTO::build_belief_graph (src/pto.rs:185-259) with everything it calls:
 *   get_successor_belief_states  src/map_io.rs:244-279, src/map_shelves_io.rs:206-240
 *   observe_impl                 src/map_io.rs:281-300, src/map_shelves_io.rs:242-265
 *   reachable_belief_states      src/map_io.rs:515-546 == src/map_shelves_io.rs:490-520
 *   hash                         src/common.rs:352-355
 *   is_compatible / compute_compatibility  src/common.rs:256-276
 *   BeliefGraph::add_node/add_edge/belief_id  src/belief_graph.rs:44-71
 * A literal restatement: the same loops in the same order; every add_edge call
 * is logged, the children / parents lists are that log bucketed by `from` / `to`
 * (stable), i.e. the reference's Vec::push order.
 * Pinned by the observation-model / reachable-belief tests of map_io.rs:666-723
 * and map_shelves_io.rs:595-660 restated on the synthetic maps
 * (tests/test_oracle_belief.py).
 */
#include "orc_internal.h"
#include <math.h>

/* common.rs:352-355.  usize arithmetic; release builds wrap on overflow (10^i leaves 64 bits at i >= 20). */
uint64_t orc_belief_hash(const double *bs, uint32_t n) {
    uint64_t h = 0, p10 = 1;
    for (uint32_t i = 0; i < n; ++i) {
        double r = round(bs[i] * 1000.0);          /* f64::round: half away from zero, like C round() */
        uint64_t q = r != r ? 0 : (r <= 0.0 ? 0 : (r >= 18446744073709551615.0 ? UINT64_MAX : (uint64_t)r)); /* `as usize` saturates */
        h += (p10 + 1) * q;
        p10 *= 10;
    }
    return h;
}

static void normalize(double *b, uint32_t n) {     /* the nested fn of both get_successor_belief_states */
    double sum = 0.0;
    for (uint32_t w = 0; w < n; ++w) sum = sum + b[w];
    for (uint32_t w = 0; w < n; ++w) b[w] /= sum;
}
static int any_nan(const double *b, uint32_t n) {
    for (uint32_t w = 0; w < n; ++w)
        if (b[w] != b[w]) return 1;
    return 0;
}

/* out: up to 2 vectors of n_worlds; returns how many (the NaN ones are dropped) */
int orc_belief_successors(const orc_ctx *c, const double *belief, uint32_t zone, double *out) {
    const uint32_t n = (uint32_t)c->n_worlds;
    int k = 0;
    if (c->domain == ORC_DOMAIN_DOOR) {            /* map_io.rs:244-279; zones_to_worlds[z] == world_validities[z] (113-128) */
        const uint64_t mask = c->validities[zone];
        double *b = out + (size_t)k * n;           /* assume closed */
        for (uint32_t w = 0; w < n; ++w) b[w] = ((mask >> w) & 1) ? 0.0 : belief[w];
        normalize(b, n);
        if (!any_nan(b, n)) ++k;
        b = out + (size_t)k * n;                   /* assume open */
        for (uint32_t w = 0; w < n; ++w) b[w] = ((mask >> w) & 1) ? belief[w] : 0.0;
        normalize(b, n);
        if (!any_nan(b, n)) ++k;
    } else {                                       /* map_shelves_io.rs:206-240 */
        double *b = out + (size_t)k * n;           /* object there */
        for (uint32_t w = 0; w < n; ++w) b[w] = w == zone ? belief[w] : 0.0;
        normalize(b, n);
        if (!any_nan(b, n)) ++k;
        b = out + (size_t)k * n;                   /* object not there */
        for (uint32_t w = 0; w < n; ++w) b[w] = w == zone ? 0.0 : belief[w];
        normalize(b, n);
        if (!any_nan(b, n)) ++k;
    }
    return k;
}

/* map_io.rs:283-297 / map_shelves_io.rs:259-265; 1 seen, 0 not, < 0 the reference panics */
int orc_zone_observable(const orc_ctx *c, const double xy[2], uint32_t zone) {
    if (!(orc_norm2(xy, c->zone_pos[zone]) < c->visibility)) return 0;
    int cls = orc_traversed_class(c, xy, c->zone_pos[zone]);
    if (cls < 0) return cls;
    return cls != ORC_HIGH_OBSTACLE;
}

/* observe_impl: out receives the posterior vectors; returns their number, -1 raster panic, -2 cap */
int64_t orc_observe(const orc_ctx *c, const double xy[2], const double *belief, double *out, size_t cap) {
    const uint32_t n = (uint32_t)c->n_worlds;
    int seen[ORC_MAX_ZONES], n_seen = 0;
    for (int z = 0; z < c->n_zones; ++z) {
        seen[z] = orc_zone_observable(c, xy, (uint32_t)z);
        if (seen[z] < 0) return -1;
        n_seen += seen[z];
    }
    if (n_seen > 20) return -2;
    const size_t room = (size_t)1 << n_seen;                 /* each seen zone at most doubles the list */
    double *cur = malloc(room * n * sizeof(double)), *nxt = malloc(room * n * sizeof(double));
    size_t cnt = 1;
    memcpy(cur, belief, n * sizeof(double));
    for (int z = 0; z < c->n_zones; ++z) {
        if (!seen[z]) continue;
        size_t m = 0;
        for (size_t i = 0; i < cnt; ++i) m += (size_t)orc_belief_successors(c, cur + i * n, (uint32_t)z, nxt + m * n);
        double *t = cur; cur = nxt; nxt = t;
        cnt = m;
    }
    int64_t ret = (int64_t)cnt;
    if (cnt > cap) ret = -2;
    else memcpy(out, cur, cnt * n * sizeof(double));
    free(cur); free(nxt);
    return ret;
}

/* reachable_belief_states: literal LIFO; `contains` is exact f64 equality, the hash set decides what is new.
 * out may be NULL to count. */
int64_t orc_reachable_beliefs(const orc_ctx *c, const double *start, double *out, size_t cap) {
    const uint32_t n = (uint32_t)c->n_worlds;
    const int nz = c->n_zones;
    size_t R = 0, Rcap = 64, L = 0, Lcap = 64;
    double *reach = malloc(Rcap * n * sizeof(double));
    uint64_t *hashes = malloc(Rcap * sizeof(uint64_t));
    double *lb = malloc(Lcap * n * sizeof(double));          /* lifo: belief */
    uint64_t *lz = malloc(Lcap * sizeof(uint64_t));          /* lifo: zones still to check (bit set; order = ascending ids, as the Vec keeps it) */
    double succ[2 * 64];
    double *cur = malloc(n * sizeof(double));
    memcpy(reach, start, n * sizeof(double));
    R = 1;                                                   /* NB: the start's hash is not inserted (map_io.rs:520) */
    size_t H = 0;
    memcpy(lb, start, n * sizeof(double));
    lz[0] = nz >= 64 ? ~0ull : ((1ull << nz) - 1);
    L = 1;
    while (L) {
        --L;
        memcpy(cur, lb + L * n, n * sizeof(double));
        const uint64_t zones = lz[L];
        for (int z = 0; z < nz; ++z) {
            if (!((zones >> z) & 1)) continue;
            const uint64_t remaining = zones & ~(1ull << z);
            int k = orc_belief_successors(c, cur, (uint32_t)z, succ);
            for (int s = 0; s < k; ++s) {
                const double *v = succ + (size_t)s * n;
                int found = 0;
                for (size_t r = 0; r < R && !found; ++r) {
                    int eq = 1;
                    for (uint32_t w = 0; w < n && eq; ++w) eq = reach[r * n + w] == v[w];
                    found = eq;
                }
                if (found) continue;
                const uint64_t hv = orc_belief_hash(v, n);
                int known = 0;
                for (size_t r = 0; r < H && !known; ++r) known = hashes[r] == hv;
                if (!known) {
                    if (R == Rcap) { Rcap *= 2; reach = realloc(reach, Rcap * n * sizeof(double)); hashes = realloc(hashes, Rcap * sizeof(uint64_t)); }
                    memcpy(reach + R * n, v, n * sizeof(double));
                    ++R;
                    hashes[H++] = hv;
                }
                if (L == Lcap) { Lcap *= 2; lb = realloc(lb, Lcap * n * sizeof(double)); lz = realloc(lz, Lcap * sizeof(uint64_t)); }
                memcpy(lb + L * n, v, n * sizeof(double));
                lz[L++] = remaining;
            }
        }
    }
    int64_t ret = (int64_t)R;
    if (out) {
        if (R > cap) ret = -2;
        else memcpy(out, reach, R * n * sizeof(double));
    }
    free(reach); free(hashes); free(lb); free(lz); free(cur);
    return ret;
}

/* common.rs:256-264 */
static int is_compatible(const double *b, uint64_t validity, uint32_t n) {
    for (uint32_t w = 0; w < n; ++w)
        if (b[w] > 0.0 && !((validity >> w) & 1)) return 0;
    return 1;
}

struct orc_bg {
    uint32_t nw;
    size_t B, NB;
    double *beliefs;
    uint8_t *types;                 /* 0 Unknown, 1 Action, 2 Observation (belief_graph.rs:13-17) */
    uint64_t n_edges;
    uint64_t *coff, *poff;
    uint32_t *cid, *pid;
};

static void bg_free(struct orc_bg *g) {
    if (!g) return;
    free(g->beliefs); free(g->types); free(g->coff); free(g->poff); free(g->cid); free(g->pid);
    free(g);
}
void orc_bg_release(orc_ctx *c) { bg_free(c->bg); c->bg = NULL; }

typedef struct { uint32_t from, to; } bedge;

/* PTO::build_belief_graph (pto.rs:185-259) on the graph of the last orc_grow(mode PTO). 0 ok, < 0 reference panics */
int orc_build_belief_graph(orc_ctx *c, const double *start_belief) {
    const uint32_t n = (uint32_t)c->n_worlds;
    orc_bg_release(c);
    if (!c->n_nodes || !c->node_validity) { snprintf(c->err, sizeof c->err, "no PTO graph"); return -1; }
    struct orc_bg *g = calloc(1, sizeof *g);
    g->nw = n;
    int64_t B = orc_reachable_beliefs(c, start_belief, NULL, 0);
    g->B = (size_t)B;
    g->beliefs = malloc(g->B * n * sizeof(double));
    orc_reachable_beliefs(c, start_belief, g->beliefs, g->B);
    uint64_t *bh = malloc(g->B * sizeof(uint64_t));
    for (size_t b = 0; b < g->B; ++b) bh[b] = orc_belief_hash(g->beliefs + b * n, n);
    for (size_t a = 0; a < g->B; ++a)                       /* create_belief_states_hash_map asserts no collision (belief_graph.rs:82) */
        for (size_t b = a + 1; b < g->B; ++b)
            if (bh[a] == bh[b]) { free(bh); bg_free(g); snprintf(c->err, sizeof c->err, "collision when hashing the belief states!"); return -2; }
    const int V = c->n_validities;
    uint8_t *compat = malloc(g->B * (size_t)V);
    for (size_t b = 0; b < g->B; ++b)
        for (int v = 0; v < V; ++v) compat[b * V + v] = (uint8_t)is_compatible(g->beliefs + b * n, c->validities[v], n);

    const size_t N = c->n_nodes;
    g->NB = N * g->B;
    g->types = calloc(g->NB, 1);
    /* PTOGraph adjacency in push order (pto.rs:111-120): for one new node first add_edge(nbr, new) for all nbr, then add_edge(new, nbr) */
    uint64_t *aoff = calloc(N + 1, sizeof(uint64_t));
    for (uint64_t e = 0; e < c->n_edges; ++e) { aoff[c->edges[e].from + 1]++; aoff[c->edges[e].to + 1]++; }
    for (size_t i = 0; i < N; ++i) aoff[i + 1] += aoff[i];
    uint32_t *aid = malloc((2 * c->n_edges + 1) * sizeof(uint32_t)), *aval = malloc((2 * c->n_edges + 1) * sizeof(uint32_t));
    uint64_t *fill = malloc((N + 1) * sizeof(uint64_t));
    memcpy(fill, aoff, (N + 1) * sizeof(uint64_t));
    /* KdTree::nearest_neighbors lists the neighbours of one new node in kd pre-order (nearest_neighbor.rs:101-117);
     * the sequential algorithm already logged its edges that way (no-op below), the batched ones logged the same
     * sets in search order, so each group is put into pre-order of the kd-tree of all nodes (KdTree::add in id order;
     * the relative pre-order of two nodes never changes once both are in the tree). */
    {
        uint64_t *pre = malloc(N * sizeof(uint64_t)), *rank = malloc(N * sizeof(uint64_t));
        const double root[2] = {c->nx[0], c->ny[0]};
        orc_kdtree *kd = orc_kd_new(root, 0);
        for (size_t i = 1; i < N; ++i) { const double s[2] = {c->nx[i], c->ny[i]}; orc_kd_add(kd, s, i); }
        size_t got = orc_kd_radius(kd, root, 1e300, pre, N);
        orc_kd_free(kd);
        if (got != N) { free(pre); free(rank); free(fill); free(aoff); free(aid); free(aval); free(bh); free(compat); bg_free(g); snprintf(c->err, sizeof c->err, "kd pre-order failed"); return -5; }
        for (size_t k = 0; k < N; ++k) rank[pre[k]] = k;
        for (uint64_t e = 0; e < c->n_edges;) {
            uint64_t e1 = e;
            while (e1 < c->n_edges && c->edges[e1].to == c->edges[e].to) ++e1;
            for (uint64_t a = e + 1; a < e1; ++a) {            /* insertion sort of one neighbour list */
                orc_edge x = c->edges[a];
                uint64_t b = a;
                while (b > e && rank[c->edges[b - 1].from] > rank[x.from]) { c->edges[b] = c->edges[b - 1]; --b; }
                c->edges[b] = x;
            }
            e = e1;
        }
        free(pre); free(rank);
    }
    for (uint64_t e = 0; e < c->n_edges;) {
        uint64_t e1 = e;
        while (e1 < c->n_edges && c->edges[e1].to == c->edges[e].to) ++e1;
        for (uint64_t k = e; k < e1; ++k) { uint64_t p = fill[c->edges[k].from]++; aid[p] = c->edges[k].to; aval[p] = c->edges[k].validity_id; }
        for (uint64_t k = e; k < e1; ++k) { uint64_t p = fill[c->edges[k].to]++; aid[p] = c->edges[k].from; aval[p] = c->edges[k].validity_id; }
        e = e1;
    }
    free(fill);

    /* node_to_belief_nodes[id][belief] = Some(id*B + belief) iff compatible (pto.rs:198-208) */
#define N2B(id, b) (compat[(size_t)(b) * V + c->node_validity[id]] ? (int64_t)((size_t)(id) * g->B + (b)) : -1)
    size_t E = 0, Ecap = 1 << 16;
    bedge *log = malloc(Ecap * sizeof(bedge));
#define ADD_EDGE(f, t) do { if (E == Ecap) { Ecap *= 2; log = realloc(log, Ecap * sizeof(bedge)); } log[E].from = (uint32_t)(f); log[E].to = (uint32_t)(t); ++E; } while (0)
    int rc = 0;
    size_t ocap = 1024;
    double *obs = malloc(ocap * n * sizeof(double));
    /* observation edges (pto.rs:211-233) */
    for (size_t id = 0; id < N && !rc; ++id) {
        const double xy[2] = {c->nx[id], c->ny[id]};
        for (size_t b = 0; b < g->B && !rc; ++b) {
            int64_t k = orc_observe(c, xy, g->beliefs + b * n, obs, ocap);
            while (k == -2 && ocap < (1u << 20)) { ocap *= 4; obs = realloc(obs, ocap * n * sizeof(double)); k = orc_observe(c, xy, g->beliefs + b * n, obs, ocap); }
            if (k < 0) { rc = -3; snprintf(c->err, sizeof c->err, "observe: raster access the reference would panic on"); break; }
            const int64_t parent = N2B(id, b);
            for (int64_t s = 0; s < k; ++s) {
                const uint64_t hs = orc_belief_hash(obs + (size_t)s * n, n);
                if (hs == bh[b]) continue;
                size_t cb = g->B;
                for (size_t q = 0; q < g->B; ++q) if (bh[q] == hs) { cb = q; break; }
                if (cb == g->B) { rc = -4; snprintf(c->err, sizeof c->err, "no id corresponding to this belief state!"); break; }
                const int64_t child = N2B(id, cb);
                if (parent >= 0 && child >= 0) { g->types[parent] = 2; ADD_EDGE(parent, child); }
            }
        }
    }
    /* action edges (pto.rs:235-257) */
    for (size_t id = 0; id < N && !rc; ++id)
        for (size_t b = 0; b < g->B; ++b) {
            const int64_t parent = N2B(id, b);
            if (parent < 0 || g->types[parent] == 2) continue;
            for (uint64_t k = aoff[id]; k < aoff[id + 1]; ++k) {
                const int64_t child = N2B(aid[k], b);
                if (child < 0) continue;
                if (compat[b * V + aval[k]]) { g->types[parent] = 1; ADD_EDGE(parent, child); }
            }
        }
    free(obs); free(aoff); free(aid); free(aval); free(bh); free(compat);
    if (rc) { free(log); bg_free(g); return rc; }
    g->n_edges = E;
    g->coff = calloc(g->NB + 1, sizeof(uint64_t));
    g->poff = calloc(g->NB + 1, sizeof(uint64_t));
    g->cid = malloc((E + 1) * sizeof(uint32_t));
    g->pid = malloc((E + 1) * sizeof(uint32_t));
    for (size_t e = 0; e < E; ++e) { g->coff[log[e].from + 1]++; g->poff[log[e].to + 1]++; }
    for (size_t i = 0; i < g->NB; ++i) { g->coff[i + 1] += g->coff[i]; g->poff[i + 1] += g->poff[i]; }
    uint64_t *cf = malloc((g->NB + 1) * sizeof(uint64_t)), *pf = malloc((g->NB + 1) * sizeof(uint64_t));
    memcpy(cf, g->coff, (g->NB + 1) * sizeof(uint64_t));
    memcpy(pf, g->poff, (g->NB + 1) * sizeof(uint64_t));
    for (size_t e = 0; e < E; ++e) { g->cid[cf[log[e].from]++] = log[e].to; g->pid[pf[log[e].to]++] = log[e].from; }
    free(cf); free(pf); free(log);
    c->bg = g;
    return 0;
}

uint64_t orc_bg_num_beliefs(const orc_ctx *c) { return c->bg ? c->bg->B : 0; }
uint64_t orc_bg_num_nodes(const orc_ctx *c) { return c->bg ? c->bg->NB : 0; }
uint64_t orc_bg_num_edges(const orc_ctx *c) { return c->bg ? c->bg->n_edges : 0; }
int orc_bg_get_beliefs(const orc_ctx *c, double *out) {
    if (!c->bg) return -1;
    memcpy(out, c->bg->beliefs, c->bg->B * c->bg->nw * sizeof(double));
    return 0;
}
int orc_bg_get_types(const orc_ctx *c, uint8_t *out) {
    if (!c->bg) return -1;
    memcpy(out, c->bg->types, c->bg->NB);
    return 0;
}
int orc_bg_get_children(const orc_ctx *c, uint64_t *off, uint32_t *ids) {
    if (!c->bg) return -1;
    memcpy(off, c->bg->coff, (c->bg->NB + 1) * sizeof(uint64_t));
    memcpy(ids, c->bg->cid, c->bg->n_edges * sizeof(uint32_t));
    return 0;
}
int orc_bg_get_parents(const orc_ctx *c, uint64_t *off, uint32_t *ids) {
    if (!c->bg) return -1;
    memcpy(off, c->bg->poff, (c->bg->NB + 1) * sizeof(uint64_t));
    memcpy(ids, c->bg->pid, c->bg->n_edges * sizeof(uint32_t));
    return 0;
}

/* read-only view for dp.c */
struct orc_bg_view { size_t B, NB; uint32_t nw; const double *beliefs; const uint8_t *types; const uint64_t *coff, *poff; const uint32_t *cid, *pid; };
int orc_bg_view_get(const orc_ctx *c, struct orc_bg_view *v) {
    if (!c->bg) return -1;
    v->B = c->bg->B; v->NB = c->bg->NB; v->nw = c->bg->nw; v->beliefs = c->bg->beliefs; v->types = c->bg->types;
    v->coff = c->bg->coff; v->poff = c->bg->poff; v->cid = c->bg->cid; v->pid = c->bg->pid;
    return 0;
}
