/* orc_internal.h -- ORACLE internals (test infrastructure only). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "porrt_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_ZONES 64
#define ORC_MAX_VALIDITIES 65
#define ORC_MAX_GOALS 64

typedef struct {
    uint32_t from, to, validity_id;
} orc_edge;

struct orc_ctx {
    char err[256];
    /* grid (map_shelves_io.rs:65-94 / map_io.rs:67-96) */
    uint8_t *occ;
    uint8_t *zones;
    uint32_t W, H;
    double low[2];
    double ppm;
    int domain;
    int has_grid;
    double visibility;
    int n_zones;
    double zone_pos[ORC_MAX_ZONES][2];
    /* worlds (map_shelves_io.rs:106-113, map_io.rs:113-128) */
    int n_worlds;
    int n_validities;
    uint64_t validities[ORC_MAX_VALIDITIES];
    /* samplers (sample_space.rs:6-60) */
    double s_low[2], s_up[2];
    orc_pcg64 crng, drng;
    double *inj_xy;
    size_t inj_n, inj_pos;
    uint32_t *inj_worlds;
    size_t inj_wn, inj_wpos;
    /* goal (common.rs:304-350, rrt.rs:325-341) */
    int goal_kind; /* 0 none, 1 square, 2 observation */
    uint32_t G;
    double goal_centers[ORC_MAX_GOALS][2];
    uint64_t goal_masks[ORC_MAX_GOALS];
    double goal_l1;
    double world_to_goal[64][2];
    uint32_t obs_zone;
    /* outputs of the last grow */
    int mode;
    uint64_t n_nodes, cap_nodes, n_iter;
    double *nx, *ny, *dist;
    int64_t *parent;
    uint64_t *reach;
    uint32_t *node_validity;
    uint64_t *final_ids, *final_masks;
    uint64_t n_final, cap_final;
    orc_edge *edges;
    uint64_t n_edges, cap_edges;
    int complete;
    int oob; /* a raster access outside W x H happened (reference would panic) */
    struct orc_bg *bg; /* belief.c: result of the last orc_build_belief_graph */
};

/* class of one pixel / point / segment in the ORC_* encoding; -1 = outside raster */
int orc_pixel_class(const orc_ctx *c, uint32_t i, uint32_t j);
void orc_ctx_reserve_nodes(orc_ctx *c, uint64_t n);
void orc_ctx_push_final(orc_ctx *c, uint64_t id, uint64_t mask);
void orc_ctx_push_edge(orc_ctx *c, uint32_t from, uint32_t to, uint32_t v);
/* validity id of a state / transition as PTOFuncs (map_shelves_io.rs:464-488,
 * map_io.rs:487-513); -1 = None */
int orc_state_validity(orc_ctx *c, const double xy[2]);
int orc_transition_validity(orc_ctx *c, const double a[2], const double b[2]);
/* RTTFuncs adapter (map_shelves_tamp_rrt.rs:35-47); no grid = empty space (rrt.rs:64-72) */
int orc_rrt_state_valid(orc_ctx *c, const double xy[2]);
int orc_rrt_transition_valid(orc_ctx *c, const double a[2], const double b[2]);

#endif
