/* mmprm.c -- ORACLE (test infrastructure only): CPU restatement of the multi-modal PRM growth
 *   MapShelfDomainTampPRM::grow_mm_prm       src/map_shelves_tamp_prm.rs:328-393
 *   ModeTree::{add_mode, add_transition, get_transitions}   :135-283
 *   sample_observation_of_zone                :482-493
 *   is_final, normalize_belief                :19-26
 *   PRM::{grow_graph, add_sample}             src/prm.rs:38-109 (one PRM, one kd-tree and one sampler clone per mode)
 * with its quirks kept: every mode's sampler is a clone of the planner's never-advanced sampler (:212,249: all modes draw the
 * same sample sequence), both transitions of a zone are created with observation = true (:227,268), the reaching probability of
 * the "not there" successor is taken before normalisation (:241).
 * Parity unpinned against reference vectors: the reference's tests of this planner need its LFS rasters (:511-597). */
#define _GNU_SOURCE        /* sincos */
#include <math.h>
#include "orc_internal.h"

typedef struct { size_t n, cap; uint64_t *v; } u64vec;
static void u64_push(u64vec *a, uint64_t x) {
    if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 16; a->v = (uint64_t *)realloc(a->v, a->cap * sizeof(uint64_t)); }
    a->v[a->n++] = x;
}

typedef struct {
    double belief[64];
    double reaching_probability;
    uint64_t hash;                  /* of the belief (the key of mode_hash_map) */
    int remaining[64], n_remaining;
    int64_t there[64], not_there[64];          /* zone -> transition index (the two hash maps), -1 = none */
    orc_pcg64 sampler;                         /* PRM::continuous_sampler */
    orc_kdtree *kd;
    uint64_t n_nodes, cap_nodes;
    double *xy;
    u64vec efrom, eto, finals;
} mm_mode;

typedef struct {
    uint32_t zone, from, to;
    int observation;
    u64vec pairs;
} mm_transition;

struct orc_mm {
    int nw;
    size_t n_modes, cap_modes, n_tr, cap_tr;
    mm_mode *modes;
    mm_transition *tr;
    uint64_t n_beliefs;
};

static uint64_t mm_hash(const double *b, int n) {               /* common.rs:352-355 */
    uint64_t h = 0, p10 = 1;
    for (int i = 0; i < n; ++i, p10 *= 10) {
        double x = b[i] * 1000.0;
        if (!(x > 0.0)) continue;
        uint64_t q = (uint64_t)x;
        if (x - (double)q >= 0.5) ++q;                           /* f64::round: half away from zero */
        h += (p10 + 1) * q;
    }
    return h;
}
static double mm_transition_probability(const double *parent, const double *child, int n) {     /* common.rs:187-190 */
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + (child[i] > 0.0 ? parent[i] : 0.0);
    return s;
}
static int mm_is_final(const double *b, int n) {                 /* :19-21 */
    double m = b[0];
    for (int i = 1; i < n; ++i) if (b[i] >= m) m = b[i];         /* max_by keeps the last maximum; only the value matters */
    return m >= 0.999;
}
static void mm_normalize(double *b, int n) {                     /* :23-26 */
    double sum = 0.0;
    for (int i = 0; i < n; ++i) sum = sum + b[i];
    for (int i = 0; i < n; ++i) b[i] = b[i] / sum;
}

/* PRM::add_sample (prm.rs:52-109) on one mode */
static uint64_t mode_add_sample(orc_ctx *c, mm_mode *m, const double s[2], double max_step, double search_radius, int *err) {
    if (m->n_nodes == m->cap_nodes) { m->cap_nodes = m->cap_nodes ? 2 * m->cap_nodes : 256; m->xy = (double *)realloc(m->xy, 2 * m->cap_nodes * sizeof(double)); }
    const uint64_t id = m->n_nodes++;
    m->xy[2 * id] = s[0]; m->xy[2 * id + 1] = s[1];
    if (id == 0) { m->kd = orc_kd_new(s, 0); return 0; }        /* prm.rs:54-58 */
    const double radius = orc_heuristic_radius(m->n_nodes, max_step, search_radius, 2);          /* prm.rs:66 */
    size_t cap = 64, n;
    uint64_t *nb = (uint64_t *)malloc(cap * sizeof(uint64_t));
    while ((n = orc_kd_radius(m->kd, s, radius, nb, cap)) > cap) { cap = n; nb = (uint64_t *)realloc(nb, cap * sizeof(uint64_t)); }
    orc_kd_add(m->kd, s, id);                                    /* prm.rs:77 */
    for (size_t a = 0; a < n; ++a) {                             /* prm.rs:90-103 */
        const double from[2] = {m->xy[2 * nb[a]], m->xy[2 * nb[a] + 1]};
        const int tv = orc_transition_validity(c, from, s);
        if (c->oob) { *err = 1; break; }
        if (tv >= 0) { u64_push(&m->efrom, nb[a]); u64_push(&m->eto, id); }
    }
    free(nb);
    return id;
}

static size_t mm_add_mode(orc_mm *g, const int *remaining, int n_remaining, double reach_p, const double *belief, const orc_pcg64 *sampler) {   /* :135-164 */
    if (g->n_modes == g->cap_modes) { g->cap_modes = g->cap_modes ? 2 * g->cap_modes : 16; g->modes = (mm_mode *)realloc(g->modes, g->cap_modes * sizeof(mm_mode)); }
    mm_mode *m = &g->modes[g->n_modes];
    memset(m, 0, sizeof *m);
    memcpy(m->belief, belief, (size_t)g->nw * sizeof(double));
    m->reaching_probability = reach_p;
    m->hash = mm_hash(m->belief, g->nw);
    memcpy(m->remaining, remaining, (size_t)n_remaining * sizeof(int));
    m->n_remaining = n_remaining;
    for (int z = 0; z < 64; ++z) m->there[z] = m->not_there[z] = -1;
    m->sampler = *sampler;
    return g->n_modes++;
}
static int64_t mm_mode_of_hash(const orc_mm *g, uint64_t h) {    /* mode_hash_map: the last mode inserted with that hash */
    for (size_t m = g->n_modes; m-- > 0;) if (g->modes[m].hash == h) return (int64_t)m;
    return -1;
}
static size_t mm_add_transition(orc_mm *g, uint32_t zone, uint32_t from, uint32_t to, int observation) {     /* :166-181 */
    if (g->n_tr == g->cap_tr) { g->cap_tr = g->cap_tr ? 2 * g->cap_tr : 16; g->tr = (mm_transition *)realloc(g->tr, g->cap_tr * sizeof(mm_transition)); }
    mm_transition *t = &g->tr[g->n_tr];
    memset(t, 0, sizeof *t);
    t->zone = zone; t->from = from; t->to = to; t->observation = observation;
    return g->n_tr++;
}

/* ModeTree::get_transitions (:183-282) */
static int mm_get_transitions(orc_ctx *c, orc_mm *g, size_t mode_id, int zone, const orc_pcg64 *planner_sampler, size_t out[2], int *err) {
    int n_out = 0;
    const int nw = g->nw;
    if (mm_is_final(g->modes[mode_id].belief, nw)) return 0;
    for (int pass = 0; pass < 2; ++pass) {                       /* object there, then object not there */
        mm_mode *mode = &g->modes[mode_id];
        int64_t *map = pass == 0 ? mode->there : mode->not_there;
        if (map[zone] >= 0) { out[n_out++] = (size_t)map[zone]; continue; }
        double sb[64];
        double reach_p;
        if (pass == 0) {
            for (int w = 0; w < nw; ++w) sb[w] = 0.0;
            sb[zone] = 1.0;
            mm_normalize(sb, nw);
            reach_p = mode->reaching_probability * mm_transition_probability(mode->belief, sb, nw);
        } else {
            memcpy(sb, mode->belief, (size_t)nw * sizeof(double));
            sb[zone] = 0.0;
            reach_p = mode->reaching_probability * mm_transition_probability(mode->belief, sb, nw);
            double sum = 0.0;
            for (int w = 0; w < nw; ++w) sum = sum + sb[w];
            if (!(sum > 0.0)) { *err = 2; return n_out; }        /* the reference asserts */
            mm_normalize(sb, nw);
        }
        int64_t succ = mm_mode_of_hash(g, mm_hash(sb, nw));
        if (succ < 0) {
            int remaining[64], nr = 0;
            for (int k = 0; k < mode->n_remaining; ++k) if (mode->remaining[k] != zone) remaining[nr++] = mode->remaining[k];
            succ = (int64_t)mm_add_mode(g, remaining, nr, reach_p, sb, planner_sampler);
            int goal_zone = -1;
            if (pass == 0) goal_zone = zone;
            else for (int w = 0; w < nw; ++w) if (sb[w] == 1.0) { goal_zone = w; break; }
            if (goal_zone >= 0) {                                /* initial goal state of the new mode */
                const uint64_t gid = mode_add_sample(c, &g->modes[succ], c->zone_pos[goal_zone], 0.0, 0.0, err);
                u64_push(&g->modes[succ].finals, gid);
            }
        }
        const size_t t = mm_add_transition(g, (uint32_t)zone, (uint32_t)mode_id, (uint32_t)succ, 1);     /* sic: `true` in both branches */
        mode = &g->modes[mode_id];                               /* (the array may have moved) */
        (pass == 0 ? mode->there : mode->not_there)[zone] = (int64_t)t;
        out[n_out++] = t;
    }
    return n_out;
}

void orc_mm_free(orc_mm *g) {
    if (!g) return;
    for (size_t m = 0; m < g->n_modes; ++m) {
        if (g->modes[m].kd) orc_kd_free(g->modes[m].kd);
        free(g->modes[m].xy); free(g->modes[m].efrom.v); free(g->modes[m].eto.v); free(g->modes[m].finals.v);
    }
    for (size_t t = 0; t < g->n_tr; ++t) free(g->tr[t].pairs.v);
    free(g->modes); free(g->tr); free(g);
}

orc_mm *orc_mm_prm_grow(orc_ctx *c, const double start[2], const double *initial_belief, double max_step, double search_radius, uint64_t n_iter_per_belief) {
    if (!c->has_grid || !c->zones || c->domain != 0) { snprintf(c->err, sizeof c->err, "multi-modal PRM: a shelf domain with zones"); return NULL; }
    const int nw = c->n_worlds;
    double bsum = 0.0;
    for (int w = 0; w < nw; ++w) bsum += initial_belief[w];
    if (fabs(bsum - 1.0) > 1e-6) { snprintf(c->err, sizeof c->err, "belief state does not sum to 1"); return NULL; }
    orc_mm *g = (orc_mm *)calloc(1, sizeof *g);
    g->nw = nw;
    const int64_t B = orc_reachable_beliefs(c, initial_belief, NULL, 0);              /* :331-335 */
    if (B < 0) { free(g); return NULL; }
    g->n_beliefs = (uint64_t)B;
    const orc_pcg64 planner_sampler = c->crng;                   /* never advanced by the planner: every mode clones this state */
    orc_pcg64 zone_sampler;                                      /* ContinuousSampler::new([0, 0], [visibility, 2 pi]) (:302) */
    orc_pcg64_seed_from_u64(&zone_sampler, 0);
    int remaining[64], err = 0;
    for (int z = 0; z < c->n_zones; ++z) remaining[z] = z;
    mm_add_mode(g, remaining, c->n_zones, 1.0, initial_belief, &planner_sampler);
    mode_add_sample(c, &g->modes[0], start, 0.0, 0.0, &err);                          /* :339 */
    const uint64_t total = n_iter_per_belief * (uint64_t)B;
    const uint64_t n_outer = (uint64_t)((double)total / 200.0);
    for (uint64_t i = 0; i < n_outer && !err; ++i) {
        const size_t mode_id = (size_t)orc_gen_range_usize(&c->drng, 0, g->n_modes);  /* :351 */
        for (int s = 0; s < 190 && !err; ++s) {                                       /* :357 grow_graph(.., 190) */
            mm_mode *m = &g->modes[mode_id];
            double p[2];
            p[0] = orc_gen_range_f64(&m->sampler, c->s_low[0], c->s_up[0]);
            p[1] = orc_gen_range_f64(&m->sampler, c->s_low[1], c->s_up[1]);
            mode_add_sample(c, m, p, max_step, search_radius, &err);
        }
        for (int j = 0; j < 10 && !err; ++j) {                                        /* :360-389 */
            if (g->modes[mode_id].n_remaining == 0) continue;
            const size_t zi = (size_t)orc_gen_range_usize(&c->drng, 0, (uint64_t)g->modes[mode_id].n_remaining);
            const int zone = g->modes[mode_id].remaining[zi];
            size_t tids[2];
            const int nt = mm_get_transitions(c, g, mode_id, zone, &planner_sampler, tids, &err);
            /* sample_observation_of_zone (:482-493): the radius draw is made and discarded */
            (void)orc_gen_range_f64(&zone_sampler, 0.0, c->visibility);
            const double angle = orc_gen_range_f64(&zone_sampler, 0.0, 2.0 * M_PI);
            /* angle.cos() and angle.sin() of one operand: rustc (LLVM) merges the two intrinsics into ONE sincos call on glibc targets,
             * whose results can differ from separate sin / cos calls in the last bit -- so sincos it is (libm, not correctly rounded:
             * parity unpinned like the radius heuristic's ln / powf) */
            double sn, cs;
            sincos(angle, &sn, &cs);
            double ts[2];
            ts[0] = c->zone_pos[zone][0] + c->visibility * cs;
            ts[1] = c->zone_pos[zone][1] + c->visibility * sn;
            for (int d = 0; d < 2; ++d) {                                             /* f64::clamp(low, up - 0.0001) */
                const double lo = c->s_low[d], hi = c->s_up[d] - 0.0001;
                if (ts[d] < lo) ts[d] = lo;
                if (ts[d] > hi) ts[d] = hi;
            }
            const uint64_t obs = mode_add_sample(c, &g->modes[mode_id], ts, max_step, search_radius, &err);
            for (int k = 0; k < nt; ++k) {
                mm_transition *t = &g->tr[tids[k]];
                const uint64_t dst = mode_add_sample(c, &g->modes[t->to], ts, max_step, search_radius, &err);
                u64_push(&t->pairs, obs); u64_push(&t->pairs, dst);
            }
        }
    }
    if (err) { snprintf(c->err, sizeof c->err, err == 1 ? "raster access the reference would panic on" : "belief without mass (the reference asserts)"); orc_mm_free(g); return NULL; }
    return g;
}

uint64_t orc_mm_num_modes(const orc_mm *g) { return g->n_modes; }
uint64_t orc_mm_num_transitions(const orc_mm *g) { return g->n_tr; }
uint64_t orc_mm_num_beliefs(const orc_mm *g) { return g->n_beliefs; }
void orc_mm_mode_info(const orc_mm *g, uint64_t m, double *belief, double *reach_p, uint64_t *n_nodes, uint64_t *n_edges, uint64_t *n_final) {
    const mm_mode *md = &g->modes[m];
    memcpy(belief, md->belief, (size_t)g->nw * sizeof(double));
    *reach_p = md->reaching_probability; *n_nodes = md->n_nodes; *n_edges = md->efrom.n; *n_final = md->finals.n;
}
void orc_mm_mode_graph(const orc_mm *g, uint64_t m, double *xy, uint64_t *efrom, uint64_t *eto, uint64_t *finals) {
    const mm_mode *md = &g->modes[m];
    memcpy(xy, md->xy, 2 * md->n_nodes * sizeof(double));
    memcpy(efrom, md->efrom.v, md->efrom.n * sizeof(uint64_t));
    memcpy(eto, md->eto.v, md->eto.n * sizeof(uint64_t));
    memcpy(finals, md->finals.v, md->finals.n * sizeof(uint64_t));
}
void orc_mm_transition(const orc_mm *g, uint64_t t, uint32_t *zone, uint32_t *from, uint32_t *to, int *observation, uint64_t *n_pairs) {
    *zone = g->tr[t].zone; *from = g->tr[t].from; *to = g->tr[t].to; *observation = g->tr[t].observation; *n_pairs = g->tr[t].pairs.n / 2;
}
void orc_mm_transition_pairs(const orc_mm *g, uint64_t t, uint64_t *pairs) { memcpy(pairs, g->tr[t].pairs.v, g->tr[t].pairs.n * sizeof(uint64_t)); }
