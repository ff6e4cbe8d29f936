/*
 * kdtree.c -- ORACLE (test infrastructure only): restatement of the reference's
 * incremental, unbalanced kd-tree, src/nearest_neighbor.rs:3-127.  Nodes are kept
 * in a growable array instead of Box links; insertion rule, visiting order and
 * pruning tests are the reference's, so ties resolve exactly as they do there.
 */
#include "orc_internal.h"
#include <math.h>

typedef struct {
    uint64_t id;
    double s[2];
    int64_t left, right;
    int64_t up;     /* slot of the parent, -1 for the root */
    uint32_t depth;
    uint8_t is_right;
} kd_node;

struct orc_kdtree {
    kd_node *n;
    size_t len, cap;
    /* id -> slot, for the structure probes of the KATs */
    int64_t *slot_of_id;
    size_t slot_cap;
};

typedef int (*kd_filter)(uint64_t id, const void *env);

static void kd_push(orc_kdtree *t, const double s[2], uint64_t id) {
    if (t->len == t->cap) {
        t->cap = t->cap ? 2 * t->cap : 1024;
        t->n = (kd_node *)realloc(t->n, t->cap * sizeof(kd_node));
    }
    kd_node *k = &t->n[t->len];
    k->id = id;
    k->s[0] = s[0];
    k->s[1] = s[1];
    k->left = k->right = -1;
    k->up = -1;
    k->depth = 0;
    k->is_right = 0;
    if (id >= t->slot_cap) {
        size_t nc = t->slot_cap ? t->slot_cap : 1024;
        while (nc <= id) nc *= 2;
        t->slot_of_id = (int64_t *)realloc(t->slot_of_id, nc * sizeof(int64_t));
        for (size_t i = t->slot_cap; i < nc; ++i) t->slot_of_id[i] = -1;
        t->slot_cap = nc;
    }
    t->slot_of_id[id] = (int64_t)t->len;
    t->len++;
}

/* nearest_neighbor.rs:19-27 */
orc_kdtree *orc_kd_new(const double root[2], uint64_t root_id) {
    orc_kdtree *t = (orc_kdtree *)calloc(1, sizeof(orc_kdtree));
    kd_push(t, root, root_id);
    return t;
}

void orc_kd_free(orc_kdtree *t) {
    if (!t) return;
    free(t->n);
    free(t->slot_of_id);
    free(t);
}

/* nearest_neighbor.rs:29-46: cycle the axes; strictly smaller goes left, equal
 * or larger goes right */
void orc_kd_add(orc_kdtree *t, const double s[2], uint64_t id) {
    size_t cur = 0;
    for (int axis = 0;; axis = (axis + 1) % 2) {
        int go_left = s[axis] < t->n[cur].s[axis];
        int64_t next = go_left ? t->n[cur].left : t->n[cur].right;
        if (next >= 0) {
            cur = (size_t)next;
        } else {
            int64_t slot = (int64_t)t->len;
            kd_push(t, s, id); /* may realloc: index, do not keep pointers */
            if (go_left) t->n[cur].left = slot;
            else t->n[cur].right = slot;
            t->n[slot].up = (int64_t)cur;
            t->n[slot].depth = t->n[cur].depth + 1;
            t->n[slot].is_right = (uint8_t)!go_left;
            return;
        }
    }
}

typedef struct {
    const orc_kdtree *t;
    double q[2];
    double dmin;
    size_t nearest;
    kd_filter filter;
    const void *env;
} nn_args;

/* nearest_neighbor.rs:59-88 */
static void nn_inner(nn_args *a, size_t from, int axis) {
    const kd_node *f = &a->t->n[from];
    {
        double d = orc_norm2(f->s, a->q);
        if (d < a->dmin && (!a->filter || a->filter(f->id, a->env))) {
            a->dmin = d;
            a->nearest = from;
        }
    }
    int next_axis = (axis + 1) % 2;
    if (a->q[axis] < f->s[axis]) {
        /* left first */
        if (a->q[axis] - a->dmin < f->s[axis] && f->left >= 0) nn_inner(a, (size_t)f->left, next_axis);
        f = &a->t->n[from];
        if (a->q[axis] + a->dmin >= f->s[axis] && f->right >= 0) nn_inner(a, (size_t)f->right, next_axis);
    } else {
        /* right first */
        if (a->q[axis] + a->dmin >= f->s[axis] && f->right >= 0) nn_inner(a, (size_t)f->right, next_axis);
        f = &a->t->n[from];
        if (a->q[axis] - a->dmin < f->s[axis] && f->left >= 0) nn_inner(a, (size_t)f->left, next_axis);
    }
}

/* nearest_neighbor.rs:52-92: starts from dmin = +inf and nearest = root, so the
 * root is returned when no node passes the filter */
static uint64_t kd_nearest_generic(const orc_kdtree *t, const double q[2], kd_filter filter, const void *env) {
    nn_args a;
    a.t = t;
    a.q[0] = q[0];
    a.q[1] = q[1];
    a.dmin = INFINITY;
    a.nearest = 0;
    a.filter = filter;
    a.env = env;
    nn_inner(&a, 0, 0);
    return t->n[a.nearest].id;
}

typedef struct {
    const uint64_t *reach;
    uint32_t world;
} reach_env;
static int reach_filter(uint64_t id, const void *env) {
    const reach_env *e = (const reach_env *)env;
    return (int)((e->reach[id] >> e->world) & 1);
}

uint64_t orc_kd_nearest(const orc_kdtree *t, const double q[2], const uint64_t *reach, uint32_t world) {
    if (!reach) return kd_nearest_generic(t, q, NULL, NULL);
    reach_env e = {reach, world};
    return kd_nearest_generic(t, q, reach_filter, &e);
}

typedef struct {
    const uint64_t *excl;
    size_t n;
} excl_env;
static int excl_filter(uint64_t id, const void *env) {
    const excl_env *e = (const excl_env *)env;
    for (size_t i = 0; i < e->n; ++i)
        if (e->excl[i] == id) return 0;
    return 1;
}

uint64_t orc_kd_nearest_excluding(const orc_kdtree *t, const double q[2], const uint64_t *excl, size_t n_excl) {
    excl_env e = {excl, n_excl};
    return kd_nearest_generic(t, q, excl_filter, &e);
}

typedef struct {
    const orc_kdtree *t;
    double q[2];
    double radius;
    uint64_t *out;
    size_t cap, n;
} rad_args;

/* nearest_neighbor.rs:101-117: pre-order, left subtree then right */
static void rad_inner(rad_args *a, size_t from, int axis) {
    const kd_node *f = &a->t->n[from];
    {
        double d = orc_norm2(f->s, a->q);
        if (d <= a->radius) {
            if (a->n < a->cap) a->out[a->n] = f->id;
            a->n++;
        }
    }
    int next_axis = (axis + 1) % 2;
    if (a->q[axis] - a->radius <= f->s[axis] && f->left >= 0) rad_inner(a, (size_t)f->left, next_axis);
    f = &a->t->n[from];
    if (a->q[axis] + a->radius >= f->s[axis] && f->right >= 0) rad_inner(a, (size_t)f->right, next_axis);
}

size_t orc_kd_radius(const orc_kdtree *t, const double q[2], double radius, uint64_t *out_ids, size_t cap) {
    rad_args a;
    a.t = t;
    a.q[0] = q[0];
    a.q[1] = q[1];
    a.radius = radius;
    a.out = out_ids;
    a.cap = cap;
    a.n = 0;
    rad_inner(&a, 0, 0);
    return a.n;
}

int64_t orc_kd_child(const orc_kdtree *t, uint64_t node_id, int right) {
    if (node_id >= t->slot_cap || t->slot_of_id[node_id] < 0) return -2;
    const kd_node *k = &t->n[t->slot_of_id[node_id]];
    int64_t c = right ? k->right : k->left;
    return c < 0 ? -1 : (int64_t)t->n[c].id;
}

int orc_kd_state(const orc_kdtree *t, uint64_t node_id, double s[2]) {
    if (node_id >= t->slot_cap || t->slot_of_id[node_id] < 0) return -1;
    const kd_node *k = &t->n[t->slot_of_id[node_id]];
    s[0] = k->s[0];
    s[1] = k->s[1];
    return 0;
}

/* 1 iff node id_u is visited before node id_v by a pre-order walk (node, left
 * subtree, right subtree) -- the order in which nearest_neighbors() lists its
 * results (nearest_neighbor.rs:101-117) and therefore the order in which
 * Iterator::min_by resolves equal costs (rrt.rs:143-145). */
int orc_kd_preorder_less(const orc_kdtree *t, uint64_t id_u, uint64_t id_v) {
    if (id_u == id_v) return 0;
    int64_t u = t->slot_of_id[id_u], v = t->slot_of_id[id_v];
    int u_right = -1, v_right = -1; /* side of the child we came from */
    while (t->n[u].depth > t->n[v].depth) {
        u_right = t->n[u].is_right;
        u = t->n[u].up;
    }
    while (t->n[v].depth > t->n[u].depth) {
        v_right = t->n[v].is_right;
        v = t->n[v].up;
    }
    if (u == v) return u_right >= 0 ? 0 : 1; /* one is an ancestor of the other: ancestor first */
    while (u != v) {
        u_right = t->n[u].is_right;
        v_right = t->n[v].is_right;
        u = t->n[u].up;
        v = t->n[v].up;
    }
    return u_right < v_right; /* left subtree before right subtree */
}
