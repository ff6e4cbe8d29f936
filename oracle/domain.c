/*
 * domain.c -- ORACLE (test infrastructure only): geometry primitives, the two
 * grid-backed domains, goals and samplers of the reference, restated in C.
 *
 *   norm1 / norm2 / steer            src/common.rs:192-225
 *   heuristic_radius                 src/common.rs:357-369
 *   SquareGoal                       src/common.rs:304-350
 *   ObservationGoal                  src/rrt.rs:325-341, src/map_shelves_tamp_rrt.rs:49-65
 *   MapShelfDomain                   src/map_shelves_io.rs:88-203, 259-265, 459-488
 *   Map (doors)                      src/map_io.rs:90-241, 482-513
 *   ContinuousSampler/DiscreteSampler src/sample_space.rs:6-60
 *   Bresenham                        line_drawing 0.8 (Cargo.toml:20), used at
 *                                    map_shelves_io.rs:196 and map_io.rs:225 --
 *                                    restated from the crate's published
 *                                    algorithm, PARITY UNPINNED (the reference's
 *                                    raycast tests need the LFS maps); pinned only
 *                                    by the crate's documented example
 *                                    (0,0)->(5,6), see tests/test_oracle_kat.py.
 */
#include "orc_internal.h"
#include <math.h>

/* ---- common.rs:192-201 */
double orc_norm1(const double a[2], const double b[2]) {
    double d = 0.0;
    for (int i = 0; i < 2; ++i) d += fabs(b[i] - a[i]);
    return d;
}

/* ---- common.rs:203-213 */
double orc_norm2(const double a[2], const double b[2]) {
    double d2 = 0.0;
    for (int i = 0; i < 2; ++i) {
        double dx = b[i] - a[i];
        d2 += dx * dx;
    }
    return sqrt(d2);
}

/* ---- common.rs:215-225 (the step length is the L1 norm) */
void orc_steer(const double from[2], double to[2], double max_step) {
    double step = orc_norm1(from, to);
    if (step > max_step) {
        double lambda = max_step / step;
        for (int i = 0; i < 2; ++i) to[i] = from[i] + (to[i] - from[i]) * lambda;
    }
}

/* ---- common.rs:357-369 (f64::ln / f64::powf are the platform libm) */
double orc_heuristic_radius(uint64_t n_nodes, double max_step, double search_radius, uint64_t dim) {
    double n = (double)n_nodes;
    double s = search_radius * pow(log(n) / n, 1.0 / (double)dim);
    return s < max_step ? s : max_step;
}

/* Rust `f64 as u32`: truncate toward zero, saturate, NaN -> 0 */
uint32_t orc_f64_as_u32(double v) {
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}

/* ---- line_drawing 0.8 Bresenham<i32> with its Octant transform; both end
 * points inclusive; error = dy - dx; per step: emit, if error >= 0 {y += 1;
 * error -= dx}, x += 1, error += dy. */
static int octant_of(int32_t x0, int32_t y0, int32_t x1, int32_t y1) {
    int value = 0;
    int32_t dx = x1 - x0, dy = y1 - y0;
    if (dy < 0) {
        dx = -dx;
        dy = -dy;
        value += 4;
    }
    if (dx < 0) {
        int32_t tmp = dx;
        dx = dy;
        dy = -tmp;
        value += 2;
    }
    if (dx < dy) value += 1;
    return value;
}
static void octant_to(int o, int32_t x, int32_t y, int32_t *ox, int32_t *oy) {
    switch (o) {
    case 0: *ox = x; *oy = y; break;
    case 1: *ox = y; *oy = x; break;
    case 2: *ox = y; *oy = -x; break;
    case 3: *ox = -x; *oy = y; break;
    case 4: *ox = -x; *oy = -y; break;
    case 5: *ox = -y; *oy = -x; break;
    case 6: *ox = -y; *oy = x; break;
    default: *ox = x; *oy = -y; break;
    }
}
static void octant_from(int o, int32_t x, int32_t y, int32_t *ox, int32_t *oy) {
    switch (o) {
    case 0: *ox = x; *oy = y; break;
    case 1: *ox = y; *oy = x; break;
    case 2: *ox = -y; *oy = x; break;
    case 3: *ox = -x; *oy = y; break;
    case 4: *ox = -x; *oy = -y; break;
    case 5: *ox = -y; *oy = -x; break;
    case 6: *ox = y; *oy = -x; break;
    default: *ox = x; *oy = -y; break;
    }
}

typedef struct {
    int32_t x, y, end_x, dx, dy, error;
    int octant;
} bres_it;

static void bres_init(bres_it *b, int32_t x0, int32_t y0, int32_t x1, int32_t y1) {
    b->octant = octant_of(x0, y0, x1, y1);
    int32_t sx, sy, ex, ey;
    octant_to(b->octant, x0, y0, &sx, &sy);
    octant_to(b->octant, x1, y1, &ex, &ey);
    b->dx = ex - sx;
    b->dy = ey - sy;
    b->x = sx;
    b->y = sy;
    b->end_x = ex;
    b->error = b->dy - b->dx;
}
static int bres_next(bres_it *b, int32_t *px, int32_t *py) {
    if (b->x > b->end_x) return 0;
    octant_from(b->octant, b->x, b->y, px, py);
    if (b->error >= 0) {
        b->y += 1;
        b->error -= b->dx;
    }
    b->x += 1;
    b->error += b->dy;
    return 1;
}

size_t orc_bresenham(int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t *out_xy, size_t cap) {
    bres_it b;
    bres_init(&b, x0, y0, x1, y1);
    size_t n = 0;
    int32_t px, py;
    while (bres_next(&b, &px, &py)) {
        if (n < cap) {
            out_xy[2 * n] = px;
            out_xy[2 * n + 1] = py;
        }
        ++n;
    }
    return n;
}

/* ------------------------------------------------------------------ context */
orc_ctx *orc_create(void) {
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    if (!c) return NULL;
    c->s_low[0] = c->s_low[1] = -1.0;
    c->s_up[0] = c->s_up[1] = 1.0;
    orc_pcg64_seed_from_u64(&c->crng, 0); /* sample_space.rs:18 */
    orc_pcg64_seed_from_u64(&c->drng, 0); /* sample_space.rs:47 */
    c->n_worlds = 1;
    c->n_validities = 1;
    c->validities[0] = 1; /* map_io.rs:108-111 init_without_zones */
    return c;
}

void orc_destroy(orc_ctx *c) {
    if (!c) return;
    orc_bg_release(c);
    free(c->occ); free(c->zones); free(c->inj_xy); free(c->inj_worlds);
    free(c->nx); free(c->ny); free(c->dist); free(c->parent); free(c->reach);
    free(c->node_validity); free(c->final_ids); free(c->final_masks); free(c->edges);
    free(c);
}

const char *orc_last_error(const orc_ctx *c) { return c ? c->err : "null context"; }

static int fail(orc_ctx *c, const char *msg) {
    snprintf(c->err, sizeof c->err, "%s", msg);
    return -1;
}

/* map_shelves_io.rs:88-94 / map_io.rs:90-96: ppm = width / (up0 - low0) */
int orc_set_grid(orc_ctx *c, const uint8_t *occ, uint32_t W, uint32_t H, const double low[2], const double up[2], int domain) {
    if (!occ || W == 0 || H == 0) return fail(c, "empty grid");
    if (domain != ORC_DOMAIN_SHELF && domain != ORC_DOMAIN_DOOR) return fail(c, "bad domain");
    free(c->occ);
    free(c->zones);
    c->zones = NULL;
    c->occ = (uint8_t *)malloc((size_t)W * H);
    memcpy(c->occ, occ, (size_t)W * H);
    c->W = W;
    c->H = H;
    c->low[0] = low[0];
    c->low[1] = low[1];
    c->ppm = (double)W / (up[0] - low[0]);
    c->domain = domain;
    c->has_grid = 1;
    c->n_zones = 0;
    c->n_worlds = 1;
    c->n_validities = 1;
    c->validities[0] = 1;
    c->visibility = 0.0;
    return 0;
}

static uint64_t ones(int n) { return n >= 64 ? ~0ULL : ((1ULL << n) - 1); }

/* map_shelves_io.rs:106-148 / map_io.rs:113-161 */
int orc_set_zones(orc_ctx *c, const uint8_t *zone_ids, double visibility) {
    if (!c->has_grid) return fail(c, "set_grid first");
    size_t n = (size_t)c->W * c->H;
    free(c->zones);
    c->zones = (uint8_t *)malloc(n);
    memcpy(c->zones, zone_ids, n);
    /* init_zone_ids: n_zones = max id + 1 (pixels equal to 255 carry no zone) */
    int max_id = 0;
    for (size_t p = 0; p < n; ++p)
        if (c->zones[p] != 255 && c->zones[p] > max_id) max_id = c->zones[p];
    c->n_zones = max_id + 1;
    if (c->n_zones > ORC_MAX_ZONES) return fail(c, "too many zones");
    /* init_zone_positions: integer centroid in u32 arithmetic, then to_coordinates
     * (which adds low[1] to x and low[0] to y, map_shelves_io.rs:172-177) */
    for (int z = 0; z < c->n_zones; ++z) {
        uint32_t si = 0, sj = 0, cnt = 0;
        for (uint32_t i = 0; i < c->H; ++i)
            for (uint32_t j = 0; j < c->W; ++j)
                if (c->zones[(size_t)i * c->W + j] == z) {
                    si += i;
                    sj += j;
                    ++cnt;
                }
        if (cnt == 0) return fail(c, "zone id without pixels (reference divides by zero)");
        uint32_t ci = si / cnt, cj = sj / cnt;
        c->zone_pos[z][0] = (double)cj / c->ppm + c->low[1];
        c->zone_pos[z][1] = (double)(c->H - 1 - ci) / c->ppm + c->low[0];
    }
    c->visibility = visibility;
    if (c->domain == ORC_DOMAIN_SHELF) {
        /* one world per zone; a single validity = all ones (map_shelves_io.rs:113) */
        c->n_worlds = c->n_zones;
        c->n_validities = 1;
        c->validities[0] = ones(c->n_worlds);
    } else {
        /* 2^n_zones worlds; zone k is traversable in worlds with bit k set; last
         * validity = all ones (map_io.rs:121-126, 198-214) */
        if (c->n_zones > 6) return fail(c, "door domain supports at most 6 zones (64 worlds)");
        c->n_worlds = 1 << c->n_zones;
        for (int z = 0; z < c->n_zones; ++z) {
            uint64_t m = 0;
            for (int w = 0; w < c->n_worlds; ++w)
                if (w & (1 << z)) m |= 1ULL << w;
            c->validities[z] = m;
        }
        c->validities[c->n_zones] = ones(c->n_worlds);
        c->n_validities = c->n_zones + 1;
    }
    return 0;
}

int orc_set_sampler(orc_ctx *c, const double low[2], const double up[2], uint64_t seed) {
    for (int i = 0; i < 2; ++i) {
        if (!(low[i] < up[i])) return fail(c, "sampler: low >= up");
        c->s_low[i] = low[i];
        c->s_up[i] = up[i];
    }
    orc_pcg64_seed_from_u64(&c->crng, seed);
    orc_pcg64_seed_from_u64(&c->drng, seed);
    free(c->inj_xy);
    c->inj_xy = NULL;
    c->inj_n = c->inj_pos = 0;
    free(c->inj_worlds);
    c->inj_worlds = NULL;
    c->inj_wn = c->inj_wpos = 0;
    return 0;
}

int orc_set_discrete_seed(orc_ctx *c, uint64_t seed) {
    orc_pcg64_seed_from_u64(&c->drng, seed);
    return 0;
}

int orc_set_samples(orc_ctx *c, const double *xy, size_t n) {
    free(c->inj_xy);
    c->inj_xy = (double *)malloc(sizeof(double) * 2 * (n ? n : 1));
    memcpy(c->inj_xy, xy, sizeof(double) * 2 * n);
    c->inj_n = n;
    c->inj_pos = 0;
    return 0;
}

int orc_set_worlds(orc_ctx *c, const uint32_t *worlds, size_t n) {
    free(c->inj_worlds);
    c->inj_worlds = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    memcpy(c->inj_worlds, worlds, sizeof(uint32_t) * n);
    c->inj_wn = n;
    c->inj_wpos = 0;
    return 0;
}

/* sample_space.rs:30-36 */
int orc_sample(orc_ctx *c, double xy[2]) {
    if (c->inj_xy) {
        if (c->inj_pos >= c->inj_n) return fail(c, "injected sample stream exhausted");
        xy[0] = c->inj_xy[2 * c->inj_pos];
        xy[1] = c->inj_xy[2 * c->inj_pos + 1];
        c->inj_pos++;
        return 0;
    }
    for (int i = 0; i < 2; ++i) xy[i] = orc_gen_range_f64(&c->crng, c->s_low[i], c->s_up[i]);
    return 0;
}

/* sample_space.rs:57-59 */
uint64_t orc_sample_discrete(orc_ctx *c, uint64_t n) {
    if (c->inj_worlds) {
        if (c->inj_wpos >= c->inj_wn) {
            fail(c, "injected world stream exhausted");
            return 0;
        }
        return c->inj_worlds[c->inj_wpos++];
    }
    return orc_gen_range_usize(&c->drng, 0, n);
}

/* common.rs:310-333 */
int orc_set_square_goal(orc_ctx *c, const double *centers, const uint64_t *masks, uint32_t G, double l1_radius) {
    if (G == 0 || G > ORC_MAX_GOALS) return fail(c, "square goal: need 1..64 goals");
    c->goal_kind = 1;
    c->G = G;
    c->goal_l1 = l1_radius;
    for (uint32_t g = 0; g < G; ++g) {
        c->goal_centers[g][0] = centers[2 * g];
        c->goal_centers[g][1] = centers[2 * g + 1];
        c->goal_masks[g] = masks[g];
    }
    for (int w = 0; w < 64; ++w) {
        c->world_to_goal[w][0] = c->world_to_goal[w][1] = 0.0;
        int has = 0;
        for (uint32_t g = 0; g < G; ++g)
            if ((masks[g] >> w) & 1) {
                if (has) return fail(c, "square goal: validities overlap");
                c->world_to_goal[w][0] = centers[2 * g];
                c->world_to_goal[w][1] = centers[2 * g + 1];
                has = 1;
            }
    }
    return 0;
}

int orc_set_observation_goal(orc_ctx *c, uint32_t zone_id) {
    if (!c->zones || (int)zone_id >= c->n_zones) return fail(c, "observation goal: unknown zone");
    c->goal_kind = 2;
    c->obs_zone = zone_id;
    return 0;
}

/* ------------------------------------------------------------- grid queries */
/* map_shelves_io.rs:165-170 == map_io.rs:176-181 */
int orc_to_pixel(const orc_ctx *c, const double xy[2], uint32_t ij[2]) {
    ij[0] = orc_f64_as_u32(((double)(c->H - 1)) - (xy[1] - c->low[1]) * c->ppm);
    ij[1] = orc_f64_as_u32((xy[0] - c->low[0]) * c->ppm);
    return 0;
}

/* class of pixel (row i, column j) == img.get_pixel(j, i)
 * shelves: map_shelves_io.rs:150-156; doors: map_io.rs:165-174, 190-196 */
int orc_pixel_class(const orc_ctx *c, uint32_t i, uint32_t j) {
    if (i >= c->H || j >= c->W) return -1; /* image::get_pixel would panic */
    uint8_t p = c->occ[(size_t)i * c->W + j];
    if (c->domain == ORC_DOMAIN_SHELF) {
        if (p == 255) return ORC_FREE;
        if (p >= 127) return ORC_LOW_OBSTACLE;
        return ORC_HIGH_OBSTACLE;
    }
    if (p == 255) return ORC_FREE;
    if (p == 0) return ORC_HIGH_OBSTACLE;
    if (!c->zones) return -2; /* "Zones missing" panic */
    uint8_t z = c->zones[(size_t)i * c->W + j];
    if (z == 255) return -2; /* unwrap() on None panics */
    return ORC_ZONE_BASE + z;
}

int orc_state_class(const orc_ctx *c, const double xy[2]) {
    uint32_t ij[2];
    orc_to_pixel(c, xy, ij);
    return orc_pixel_class(c, ij[0], ij[1]);
}

/* map_shelves_io.rs:187-203 and map_io.rs:216-241 */
int orc_traversed_class(const orc_ctx *c, const double a[2], const double b[2]) {
    uint32_t aij[2], bij[2];
    orc_to_pixel(c, a, aij);
    orc_to_pixel(c, b, bij);
    bres_it it;
    bres_init(&it, (int32_t)aij[0], (int32_t)aij[1], (int32_t)bij[0], (int32_t)bij[1]);
    int32_t i, j;
    if (c->domain == ORC_DOMAIN_SHELF) {
        uint8_t lowest = 255;
        while (bres_next(&it, &i, &j)) {
            if (i < 0 || j < 0 || (uint32_t)i >= c->H || (uint32_t)j >= c->W) return -1;
            uint8_t p = c->occ[(size_t)i * c->W + (size_t)j];
            if (p < lowest) lowest = p;
            if (lowest == 0) return ORC_HIGH_OBSTACLE;
        }
        if (lowest == 255) return ORC_FREE;
        if (lowest >= 127) return ORC_LOW_OBSTACLE;
        return ORC_HIGH_OBSTACLE;
    }
    int traversed = ORC_FREE;
    while (bres_next(&it, &i, &j)) {
        if (i < 0 || j < 0 || (uint32_t)i >= c->H || (uint32_t)j >= c->W) return -1;
        int cls = orc_pixel_class(c, (uint32_t)i, (uint32_t)j);
        if (cls == ORC_FREE) continue;
        if (cls == ORC_HIGH_OBSTACLE) return ORC_HIGH_OBSTACLE;
        if (cls < 0) return cls;
        if (traversed >= ORC_ZONE_BASE && traversed != cls) return -3; /* "multiple zone traversal not supported" */
        traversed = cls;
    }
    return traversed;
}

/* PTOFuncs::state_validity: map_shelves_io.rs:464-469, map_io.rs:487-493 */
int orc_state_validity(orc_ctx *c, const double xy[2]) {
    int cls = orc_state_class(c, xy);
    if (cls < 0) {
        c->oob = 1;
        return -1;
    }
    if (cls == ORC_FREE) return c->n_validities - 1;
    if (cls >= ORC_ZONE_BASE) return cls - ORC_ZONE_BASE;
    return -1;
}

/* PTOFuncs::transition_validator: map_shelves_io.rs:471-488, map_io.rs:495-513 */
int orc_transition_validity(orc_ctx *c, const double a[2], const double b[2]) {
    int cls = orc_traversed_class(c, a, b);
    if (cls < 0) {
        c->oob = 1;
        return -1;
    }
    if (cls == ORC_FREE) return c->n_validities - 1;
    if (cls >= ORC_ZONE_BASE) return cls - ORC_ZONE_BASE;
    return -1;
}

/* RTTFuncs adapter: map_shelves_tamp_rrt.rs:35-47; defaults rrt.rs:64-72 */
int orc_rrt_state_valid(orc_ctx *c, const double xy[2]) {
    if (!c->has_grid) return 1;
    int cls = orc_state_class(c, xy);
    if (cls < 0) c->oob = 1;
    return cls == ORC_FREE;
}
int orc_rrt_transition_valid(orc_ctx *c, const double a[2], const double b[2]) {
    if (!c->has_grid) return 1;
    int cls = orc_traversed_class(c, a, b);
    if (cls < 0) c->oob = 1;
    return cls == ORC_FREE;
}

/* GoalFuncs::goal: SquareGoal common.rs:336-345; ObservationGoal rrt.rs:330-336
 * with is_zone_observable map_shelves_io.rs:259-265 */
int orc_goal(const orc_ctx *c, const double xy[2], uint64_t *mask) {
    if (c->goal_kind == 1) {
        for (uint32_t g = 0; g < c->G; ++g)
            if (orc_norm1(xy, c->goal_centers[g]) < c->goal_l1) {
                if (mask) *mask = c->goal_masks[g];
                return 1;
            }
        return 0;
    }
    if (c->goal_kind == 2) {
        const double *zp = c->zone_pos[c->obs_zone];
        if (orc_norm2(xy, zp) < c->visibility) {
            int cls = orc_traversed_class(c, xy, zp);
            if (cls != ORC_HIGH_OBSTACLE && cls >= 0) {
                if (mask) *mask = 1;
                return 1;
            }
        }
        return 0;
    }
    return 0;
}

/* GoalFuncs::goal_example: common.rs:347-349; rrt.rs:338-340 */
int orc_goal_example(const orc_ctx *c, uint32_t world, double xy[2]) {
    if (c->goal_kind == 1) {
        xy[0] = c->world_to_goal[world & 63][0];
        xy[1] = c->world_to_goal[world & 63][1];
        return 0;
    }
    if (c->goal_kind == 2) {
        xy[0] = c->zone_pos[c->obs_zone][0];
        xy[1] = c->zone_pos[c->obs_zone][1];
        return 0;
    }
    xy[0] = xy[1] = 0.0; /* common.rs:299-301 default */
    return 0;
}

int orc_n_zones(const orc_ctx *c) { return c->n_zones; }
int orc_n_worlds(const orc_ctx *c) { return c->n_worlds; }
int orc_n_validities(const orc_ctx *c) { return c->n_validities; }
int orc_get_validities(const orc_ctx *c, uint64_t *out) {
    for (int i = 0; i < c->n_validities; ++i) out[i] = c->validities[i];
    return c->n_validities;
}
int orc_get_zone_positions(const orc_ctx *c, double *xy) {
    for (int z = 0; z < c->n_zones; ++z) {
        xy[2 * z] = c->zone_pos[z][0];
        xy[2 * z + 1] = c->zone_pos[z][1];
    }
    return c->n_zones;
}

/* ----------------------------------------------------------- output buffers */
void orc_ctx_reserve_nodes(orc_ctx *c, uint64_t n) {
    if (n <= c->cap_nodes) return;
    uint64_t cap = c->cap_nodes ? c->cap_nodes : 1024;
    while (cap < n) cap *= 2;
    c->nx = (double *)realloc(c->nx, cap * sizeof(double));
    c->ny = (double *)realloc(c->ny, cap * sizeof(double));
    c->dist = (double *)realloc(c->dist, cap * sizeof(double));
    c->parent = (int64_t *)realloc(c->parent, cap * sizeof(int64_t));
    c->reach = (uint64_t *)realloc(c->reach, cap * sizeof(uint64_t));
    c->node_validity = (uint32_t *)realloc(c->node_validity, cap * sizeof(uint32_t));
    c->cap_nodes = cap;
}

void orc_ctx_push_final(orc_ctx *c, uint64_t id, uint64_t mask) {
    if (c->n_final == c->cap_final) {
        c->cap_final = c->cap_final ? 2 * c->cap_final : 64;
        c->final_ids = (uint64_t *)realloc(c->final_ids, c->cap_final * sizeof(uint64_t));
        c->final_masks = (uint64_t *)realloc(c->final_masks, c->cap_final * sizeof(uint64_t));
    }
    c->final_ids[c->n_final] = id;
    c->final_masks[c->n_final] = mask;
    c->n_final++;
}

void orc_ctx_push_edge(orc_ctx *c, uint32_t from, uint32_t to, uint32_t v) {
    if (c->n_edges == c->cap_edges) {
        c->cap_edges = c->cap_edges ? 2 * c->cap_edges : 4096;
        c->edges = (orc_edge *)realloc(c->edges, c->cap_edges * sizeof(orc_edge));
    }
    c->edges[c->n_edges].from = from;
    c->edges[c->n_edges].to = to;
    c->edges[c->n_edges].validity_id = v;
    c->n_edges++;
}
