/*
 * reach.c -- ORACLE (test infrastructure only): restatement of
 * src/pto_reachability.rs:6-102 (per-node world-reachability masks, conservative
 * one-hop propagation, lazily refreshed completeness of the final set).
 * A WorldMask (BitVec, bit w <-> world w) is held in one u64; the reference
 * never uses more than 16 worlds (src/map_io.rs:144, src/pto.rs:347).
 */
#include "orc_internal.h"

struct orc_reach {
    uint64_t *validities, *reach;
    size_t n, cap;
    uint64_t *final_ids, *finalities;
    size_t n_final, cap_final;
    uint64_t finality;
    uint32_t n_worlds;
    int dirty;
};

orc_reach *orc_reach_new(void) { return (orc_reach *)calloc(1, sizeof(orc_reach)); }

void orc_reach_free(orc_reach *r) {
    if (!r) return;
    free(r->validities);
    free(r->reach);
    free(r->final_ids);
    free(r->finalities);
    free(r);
}

static void push_node(orc_reach *r, uint64_t validity, uint64_t reach) {
    if (r->n == r->cap) {
        r->cap = r->cap ? 2 * r->cap : 1024;
        r->validities = (uint64_t *)realloc(r->validities, r->cap * sizeof(uint64_t));
        r->reach = (uint64_t *)realloc(r->reach, r->cap * sizeof(uint64_t));
    }
    r->validities[r->n] = validity;
    r->reach[r->n] = reach;
    r->n++;
}

/* pto_reachability.rs:23-28: the root is reachable in the worlds it is valid in */
void orc_reach_set_root(orc_reach *r, uint64_t validity, uint32_t n_worlds) {
    r->n_worlds = n_worlds;
    push_node(r, validity, validity);
    r->finality = 0;
}

/* pto_reachability.rs:30-33: a new node starts unreachable */
void orc_reach_add_node(orc_reach *r, uint64_t validity) { push_node(r, validity, 0); }

/* pto_reachability.rs:35-40 */
void orc_reach_add_final_node(orc_reach *r, uint64_t id, uint64_t finality) {
    if (r->n_final == r->cap_final) {
        r->cap_final = r->cap_final ? 2 * r->cap_final : 64;
        r->final_ids = (uint64_t *)realloc(r->final_ids, r->cap_final * sizeof(uint64_t));
        r->finalities = (uint64_t *)realloc(r->finalities, r->cap_final * sizeof(uint64_t));
    }
    r->final_ids[r->n_final] = id;
    r->finalities[r->n_final] = finality;
    r->n_final++;
    r->dirty = 1;
}

static int is_final(const orc_reach *r, uint64_t id) {
    for (size_t i = 0; i < r->n_final; ++i)
        if (r->final_ids[i] == id) return 1;
    return 0;
}

/* pto_reachability.rs:42-52: reach[to][w] |= reach[from][w] & edge[w], bit by bit */
void orc_reach_add_edge(orc_reach *r, uint64_t from, uint64_t to, uint64_t edge_validity) {
    for (uint32_t i = 0; i < r->n_worlds; ++i) {
        uint64_t r_to = (r->reach[to] >> i) & 1;
        uint64_t r_from = (r->reach[from] >> i) & 1;
        uint64_t v = (edge_validity >> i) & 1;
        uint64_t nv = r_to | (r_from & v);
        r->reach[to] = (r->reach[to] & ~(1ULL << i)) | (nv << i);
    }
    if (r->n_worlds && is_final(r, to)) r->dirty = 1;
}

uint64_t orc_reach_get(const orc_reach *r, uint64_t id) { return r->reach[id]; }

/* pto_reachability.rs:92-101 */
static void update_finality(orc_reach *r) {
    for (size_t k = 0; k < r->n_final; ++k) {
        uint64_t node_reach = r->reach[r->final_ids[k]];
        r->finality |= node_reach & r->finalities[k];
    }
}

/* pto_reachability.rs:81-90 */
int orc_reach_is_final_set_complete(orc_reach *r) {
    if (r->n_final == 0) return 0;
    if (r->dirty) {
        update_finality(r);
        r->dirty = 0;
    }
    uint64_t all = r->n_worlds >= 64 ? ~0ULL : ((1ULL << r->n_worlds) - 1);
    return (r->finality & all) == all;
}

/* pto_reachability.rs:58-63 */
size_t orc_reach_final_nodes_for_world(const orc_reach *r, uint32_t world, uint64_t *out, size_t cap) {
    size_t n = 0;
    for (size_t k = 0; k < r->n_final; ++k) {
        uint64_t id = r->final_ids[k];
        if (((r->reach[id] >> world) & 1) && ((r->finalities[k] >> world) & 1)) {
            if (n < cap) out[n] = id;
            n++;
        }
    }
    return n;
}
