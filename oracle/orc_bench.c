/*
 * orc_bench.c -- ORACLE (test infrastructure only): times the sequential CPU
 * restatement of RRT::grow_tree (src/rrt.rs:102-174) on one thread.  Used by
 * bench.py's cpu_baseline leg ("port": the Rust reference cannot be built here).
 * usage: orc_bench <map.pgm|-> <n_iter> <seed> [algo=0] [K=1]
 */
#include "porrt_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static uint8_t *read_p5(const char *path, uint32_t *W, uint32_t *H) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    char magic[3] = {0};
    int w, h, maxv;
    if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "P5")) { fclose(f); return NULL; }
    int c;
    int vals[3], n = 0;
    while (n < 3) {
        c = fgetc(f);
        if (c == '#') { while ((c = fgetc(f)) != '\n' && c != EOF) {} continue; }
        if (c == ' ' || c == '\n' || c == '\r' || c == '\t') continue;
        ungetc(c, f);
        if (fscanf(f, "%d", &vals[n]) != 1) { fclose(f); return NULL; }
        n++;
    }
    fgetc(f);
    w = vals[0]; h = vals[1]; maxv = vals[2];
    (void)maxv;
    uint8_t *d = (uint8_t *)malloc((size_t)w * h);
    if (fread(d, 1, (size_t)w * h, f) != (size_t)w * h) { free(d); fclose(f); return NULL; }
    fclose(f);
    *W = (uint32_t)w; *H = (uint32_t)h;
    return d;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s map.pgm|- n_iter seed [algo] [K]\n", argv[0]); return 2; }
    uint64_t n_iter = strtoull(argv[2], NULL, 10), seed = strtoull(argv[3], NULL, 10);
    int algo = argc > 4 ? atoi(argv[4]) : 0;
    uint32_t K = argc > 5 ? (uint32_t)atoi(argv[5]) : 1;
    orc_ctx *c = orc_create();
    double low[2] = {-1, -1}, up[2] = {1, 1};
    if (strcmp(argv[1], "-")) {
        uint32_t W, H;
        uint8_t *g = read_p5(argv[1], &W, &H);
        if (!g) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
        orc_set_grid(c, g, W, H, low, up, ORC_DOMAIN_SHELF);
        free(g);
    }
    orc_set_sampler(c, low, up, seed);
    double centers[2] = {0.9, 0.0};
    uint64_t mask = 1;
    orc_set_square_goal(c, centers, &mask, 1, 0.05);
    double start[2] = {0.0, -1.0};
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int rc = orc_grow(c, start, 0.1, 2.0, n_iter, n_iter, K, ORC_MODE_RRT, algo);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double s = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    uint64_t n = orc_num_nodes(c);
    printf("{\"rc\": %d, \"n_iter\": %llu, \"n_nodes\": %llu, \"n_final\": %llu, \"seconds\": %.6f, \"expansions_per_s\": %.1f}\n",
           rc, (unsigned long long)orc_num_iterations(c), (unsigned long long)n,
           (unsigned long long)orc_num_final(c), s, (double)(n - 1) / s);
    orc_destroy(c);
    return rc < 0;
}
