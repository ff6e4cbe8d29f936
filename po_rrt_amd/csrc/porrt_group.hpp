// porrt_group.hpp -- the RRT* step kernels with several samples per wave (gfx950).
//
// A grow step of RRT::grow_tree (src/rrt.rs:109-168) is two kernels:
//   k_nn2    GL lanes per sample: nearest neighbour (KdTree::nearest_neighbor, nearest_neighbor.rs:48-92, as the exact
//            minimum of (norm2, id) over the region pages), steer (common.rs:215-225), point validity
//            (map_shelves_io.rs:158-170).  Extra workgroups run the previous step's rewire phase 2.
//   k_conn2  GL lanes per sample: radius search around the steered state (nearest_neighbor.rs:94-126) whose hits
//            never leave the CU -- they are compacted into an LDS list, raycast (map_shelves_io.rs:187-203), reduced to
//            the best parent (rrt.rs:137-145), and only the handful of actual rewire candidates (rrt.rs:152-161) go to
//            memory for the commit pass.  A sample with more hits than the LDS list holds (the dense start of a tree,
//            the copies of the goal point) is served afterwards by the whole workgroup through the global lists, the
//            way k_connect_rrt does it.  One extra workgroup files the step's new nodes into the region pages.
// Between the two lies the only global dependency of a step: a new node's id is N + its rank among the step's valid
// samples.
//
// Why groups: with one wave per sample a launch for 128 queries is 131 072 waves of ~1200 vector instructions each, a
// third of them busy; with 16 lanes per sample it is a quarter of the waves and the index arithmetic is shared by
// four samples.  The neighbour lists used to make a round trip through HBM (20 B per neighbour written by the search,
// read by the connect pass, 8 B more written and read again by the commit pass); now 12 B per actual rewire go out.
#pragma once
#include "porrt_device.hpp"

namespace porrt {


// visit(x, y, id, ok) for every node whose region meets the box of the disc (q, rho), called by all lanes of the group
// together; q, rho, N are uniform over the group.  See scan_disc for the page layout; here a group walks the 64-slot
// pages of R regions per round (lane <-> region for the counts), 64 / GL coalesced loads per page in flight.
// (A dense walk -- all regions as one virtual array, slots found by a binary search over the prefix sums held across
// the group -- was measured: the dependent cross-lane reads cost more than the idle lanes at region ends.)
template <int GL, int R, bool WITHD, class Visit>
__device__ __forceinline__ void gscan_disc(const RunConst &rc, uint32_t b, const GTeam<GL> &tm, double qx, double qy, double rho, uint32_t N, Visit visit,
                                           uint32_t skip_region = 0xFFFFFFFFu) {
    const uint32_t gl = tm.gl;
    int cx0, cy0, cx1, cy1;
    rep_cell(rc, qx - rho, qy - rho, kRG, cx0, cy0);
    rep_cell(rc, qx + rho, qy + rho, kRG, cx1, cy1);
    const uint32_t x0 = (uint32_t)cx0, y0 = (uint32_t)cy0;
    const uint32_t w = (uint32_t)(cx1 - cx0 + 1), nreg = w * (uint32_t)(cy1 - cy0 + 1);
    constexpr int U = (int)kPage / GL;
    if (nreg * 16u > N) {           // young tree: streaming the id-ordered arrays is cheaper than walking empty regions
        auto gx = as_global(rc.nx), gy = as_global(rc.ny), gdA = as_global(rc.distA);
        for (uint32_t j0 = 0; j0 < N; j0 += (uint32_t)(U * GL)) {
            double x[U], y[U], d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t j = j0 + (uint32_t)(GL * u) + gl;
                x[u] = gx[j < N ? j : 0u];
                y[u] = gy[j < N ? j : 0u];
                d[u] = WITHD ? gdA[j < N ? j : 0u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (j0 + (uint32_t)(GL * u) < N) visit(x[u], y[u], d[u], (int)(j0 + (uint32_t)(GL * u) + gl), j0 + (uint32_t)(GL * u) + gl < N);
        }
        return;
    }
    auto gcnt = as_global(rc.rg_cnt) + (b & 1u) * kRegions;
    auto gdir = as_global(rc.rg_dir);
    auto gxy = as_global(reinterpret_cast<const dbl2 *>(rc.pg_xy));
    auto gid = as_global(rc.pg_id);
    auto gpd = as_global(rc.pg_d);
    for (uint32_t r0 = 0; r0 < nreg; r0 += (uint32_t)GL) {
        const uint32_t r = r0 + gl;
        uint32_t reg = 0, cnt = 0;
        if (r < nreg) {
            const uint32_t ry = r / w;
            reg = (y0 + ry) * kRG + x0 + (r - ry * w);
            cnt = reg == skip_region ? 0u : gcnt[reg];
        }
        uint32_t page = reg;                                     // the first page of a region is static
        for (uint32_t lvl = 0;; ++lvl) {                         // page level: slots [64 lvl, 64 lvl + 64) of every region
            const uint32_t have = cnt > lvl * kPage ? cnt - lvl * kPage : 0u;
            unsigned long long m = tm.ballot(have > 0u);
            if (!m) break;
            // the next level's page ids are fetched while this level is walked
            uint32_t page_next = 0;
            if (have > kPage) page_next = gdir[(size_t)reg * rc.rg_maxp + lvl + 1u];
            while (m) {
                uint32_t pg[R], pc[R];
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    pg[q] = 0; pc[q] = 0;
                    if (m) {
                        const int l = (int)__builtin_ctzll(m);
                        m &= m - 1;
                        pg[q] = tm.shfl(page, l);
                        const uint32_t h = tm.shfl(have, l);
                        pc[q] = h < kPage ? h : kPage;
                    }
                }
                dbl2 v[R][U];
                double d[R][U];
                int id[R][U];
#pragma unroll
                for (int q = 0; q < R; ++q)
#pragma unroll
                    for (int u = 0; u < U; ++u) {                 // only the filled slots are fetched
                        const uint32_t sl = (uint32_t)(u * GL) + gl;
                        const bool ld = sl < pc[q];
                        v[q][u] = ld ? gxy[(size_t)pg[q] * kPage + sl] : dbl2{0.0, 0.0};
                        id[q][u] = ld ? gid[(size_t)pg[q] * kPage + sl] : -1;
                        d[q][u] = (WITHD && ld) ? gpd[(size_t)pg[q] * kPage + sl] : 0.0;
                    }
#pragma unroll
                for (int q = 0; q < R; ++q)
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if ((uint32_t)(u * GL) < pc[q]) visit(v[q][u].x, v[q][u].y, d[q][u], id[q][u], (uint32_t)(u * GL) + gl < pc[q]);
            }
            page = page_next;
        }
    }
}

// ---- where a workgroup works
// Workgroups are dealt to the 8 XCDs round-robin in launch order, and each XCD has its own 4 MiB L2.  A launch serves Q
// queries (grid rows) whose trees share nothing, so the (x, row) pair a workgroup works on is rearranged: XCD c takes
// the rows c, c + 8, ... one after the other, all workgroups of a row on one XCD -- a query's pages are fetched into one
// L2 instead of eight.  (Which XCD gets launch index 0 does not matter; only that i and i + 8 share one.)
__device__ __forceinline__ void xcd_swizzle(uint32_t &bx, uint32_t &by) {
    const uint32_t gx = gridDim.x, Q = gridDim.y;
    if (Q & 7u) return;
    const uint32_t L = by * gx + bx, slot = L >> 3;
    by = (slot / gx) * 8u + (L & 7u);
    bx = slot % gx;
}
// The same with `ns` single workgroups per row (page filing, clone) taken out first: launch indices 0 .. ns Q - 1 are
// row i / ns, role i % ns -- they take longest and start before everything else -- the rest are the rows' gx - ns
// ordinary workgroups.  Returns the role (< ns) or ns for an ordinary workgroup, whose index is left in bx.
__device__ __forceinline__ uint32_t xcd_swizzle_roles(uint32_t &bx, uint32_t &by, uint32_t ns) {
    const uint32_t gx = gridDim.x, Q = gridDim.y;
    const uint32_t L = by * gx + bx;
    if (L < ns * Q) { by = L / ns; bx = 0; return L % ns; }
    const uint32_t t = L - ns * Q, g2 = gx - ns;
    if (Q & 7u) { by = t / g2; bx = t % g2; return ns; }
    const uint32_t slot = t >> 3;                      // ns Q is a multiple of 8: t and L agree on the XCD
    by = (slot / g2) * 8u + (t & 7u);
    bx = slot % g2;
    return ns;
}

// One workgroup per step: the step's sample indices, ordered by the region grid cell the sample lies in (8 x 8-cell
// tiles, row-major inside).  Samples served by one workgroup -- and by workgroups that run at the same time -- then
// read the same few region pages.  Any order is a correct order: results are indexed by the sample.
__global__ __launch_bounds__(256) void k_sort_samples(const RunConst *__restrict__ rcp, uint32_t b0, unsigned long long it0, unsigned long long n, uint32_t K) {
    const RunConst &rc = rcp[blockIdx.y];      // one context per grid row
    __shared__ uint32_t s_bin[kRegions];
    __shared__ uint32_t s_tot[256];
    const uint32_t s = blockIdx.x, b = b0 + s;
    unsigned long long i0 = it0 + (unsigned long long)s * K;
    uint32_t nb = (unsigned long long)s * K < n ? (uint32_t)(n - (unsigned long long)s * K < K ? n - (unsigned long long)s * K : K) : 0u;
    if (rc.sched_nb) {                          // the row's own plan (k_sched_init): step b may start anywhere and be shorter
        if (b >= rc.sched_steps) return;
        i0 = rc.sched_i0[b]; nb = rc.sched_nb[b];
    }
    auto key_of = [&](uint32_t k) {
        int cx, cy;
        rep_cell(rc, rc.sx[i0 + k], rc.sy[i0 + k], kRG, cx, cy);
        constexpr int TPR = (kRG + 7) / 8;
        const uint32_t tile = (uint32_t)((cy >> 3) * TPR + (cx >> 3)), in = (uint32_t)((cy & 7) * 8 + (cx & 7));
        // tiles are 64 keys apart; the last tile row / column of a 40 x 40 grid is only partly used
        return tile * 64u + in;
    };
    constexpr uint32_t NK = ((kRG + 7) / 8) * ((kRG + 7) / 8) * 64u;
    static_assert(NK <= kRegions, "key space");
    for (uint32_t r = threadIdx.x; r < NK; r += 256u) s_bin[r] = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nb; k += 256u) atomicAdd(&s_bin[key_of(k)], 1u);
    __syncthreads();
    // exclusive scan of the bins: per-thread runs, then the run totals
    constexpr uint32_t PER = (NK + 255u) / 256u;
    uint32_t run = 0;
    for (uint32_t q = 0; q < PER; ++q) { const uint32_t r = threadIdx.x * PER + q; if (r < NK) run += s_bin[r]; }
    s_tot[threadIdx.x] = run;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t acc = 0; for (uint32_t t = 0; t < 256u; ++t) { const uint32_t v = s_tot[t]; s_tot[t] = acc; acc += v; } }
    __syncthreads();
    uint32_t acc = s_tot[threadIdx.x];
    for (uint32_t q = 0; q < PER; ++q) { const uint32_t r = threadIdx.x * PER + q; if (r < NK) { const uint32_t v = s_bin[r]; s_bin[r] = acc; acc += v; } }
    __syncthreads();
    uint16_t *out = rc.perm + (size_t)b * rc.part_stride;
    double *ox = rc.ssx + (size_t)b * rc.part_stride, *oy = rc.ssy + (size_t)b * rc.part_stride;
    for (uint32_t k = threadIdx.x; k < nb; k += 256u) {
        const uint32_t pos = atomicAdd(&s_bin[key_of(k)], 1u);
        out[pos] = (uint16_t)k;
        ox[pos] = rc.sx[i0 + k];
        oy[pos] = rc.sy[i0 + k];
    }
}

// nn_bound_wave for a group (GL >= 16): lanes 0..8 of the group take the 3x3 cells.
template <int GL>
__device__ __forceinline__ double nn_bound_group(const RunConst &rc, const GTeam<GL> &tm, uint32_t N, double qx, double qy) {
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    double m = INF;
    for (int l = 0; l < kRepLevels; ++l) {
        const int G = rep_dim(l);
        int cx, cy;
        rep_cell(rc, qx, qy, G, cx, cy);
        int r = -1;
        if (tm.gl < 9u) {
            const int x = cx + (int)(tm.gl % 3u) - 1, y = cy + (int)(tm.gl / 3u) - 1;
            if (x >= 0 && y >= 0 && x < G && y < G) r = as_global(rc.rep)[rep_off(l) + y * G + x];
        }
        double d2 = INF;
        if (r >= 0 && (uint32_t)r < N) d2 = dist2(as_global(rc.nx)[r], as_global(rc.ny)[r], qx, qy);
        if constexpr (GL == 16 && PORRT_DPP) {
            double o;
            o = dpp_f64<kDppRowRor + 8>(d2); d2 = o < d2 ? o : d2;
            o = dpp_f64<kDppRowRor + 4>(d2); d2 = o < d2 ? o : d2;
            o = dpp_f64<kDppRowRor + 2>(d2); d2 = o < d2 ? o : d2;
            o = dpp_f64<kDppRowRor + 1>(d2); d2 = o < d2 ? o : d2;
            m = d2;                                        // (every lane of the row holds the minimum)
        } else {
            for (int off = 8; off > 0; off >>= 1) {       // lanes 0..15 of the group hold everything
                const double o = __shfl_xor(d2, off);
                d2 = o < d2 ? o : d2;
            }
            m = tm.shfl(d2, 0);
        }
        if (m < INF) break;
    }
    if (m == INF) m = dist2(as_global(rc.nx)[0], as_global(rc.ny)[0], qx, qy);      // the root always exists
    return m;
}

// one region's pages for a group: visit(x, y, id, ok) for its `cnt` nodes.  kRegPages pages in flight: a young tree is a dense
// blob, its regions hold many pages each, and the samples far from it -- most of them, then -- walk those pages one trip at a time
// otherwise.
#ifndef PORRT_REG_PAGES
#define PORRT_REG_PAGES 1
#endif
template <int GL, class Visit>
__device__ __forceinline__ void gscan_region(const RunConst &rc, const GTeam<GL> &tm, uint32_t reg, uint32_t cnt, Visit visit) {
    auto gdir = as_global(rc.rg_dir);
    auto gxy = as_global(reinterpret_cast<const dbl2 *>(rc.pg_xy));
    auto gid = as_global(rc.pg_id);
    constexpr int U = (int)kPage / GL, RP = PORRT_REG_PAGES;
    const uint32_t npages = (cnt + kPage - 1u) / kPage;
    for (uint32_t j0 = 0; j0 < npages; j0 += (uint32_t)RP) {
        uint32_t page[RP], pc[RP];
#pragma unroll
        for (int q = 0; q < RP; ++q) {
            const uint32_t j = j0 + (uint32_t)q;
            page[q] = 0; pc[q] = 0;
            if (j < npages) {
                page[q] = j ? gdir[(size_t)reg * rc.rg_maxp + j] : reg;        // the first page of a region is static
                pc[q] = cnt - j * kPage < kPage ? cnt - j * kPage : kPage;
            }
        }
        dbl2 v[RP][U];
        int id[RP][U];
#pragma unroll
        for (int q = 0; q < RP; ++q)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t sl = (uint32_t)(u * GL) + tm.gl;
                const bool ld = sl < pc[q];
                v[q][u] = ld ? gxy[(size_t)page[q] * kPage + sl] : dbl2{0.0, 0.0};
                id[q][u] = ld ? gid[(size_t)page[q] * kPage + sl] : -1;
            }
#pragma unroll
        for (int q = 0; q < RP; ++q)
#pragma unroll
            for (int u = 0; u < U; ++u)
                if ((uint32_t)(u * GL) < pc[q]) visit(v[q][u].x, v[q][u].y, id[q][u], (uint32_t)(u * GL) + tm.gl < pc[q]);
    }
}

// squared distance from q to the part of the plane region (cx, cy) takes its nodes from (rep_cell: border regions reach to
// infinity), shrunk by a margin that covers the roundings of the cell function: a lower bound for every node of the region
__device__ __forceinline__ double region_gap2(const RunConst &rc, double qx, double qy, int cx, int cy, double wx, double wy) {
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    const double xlo = cx == 0 ? -INF : rc.bx0 + (double)cx * wx, xhi = cx == kRG - 1 ? INF : rc.bx0 + (double)(cx + 1) * wx;
    const double ylo = cy == 0 ? -INF : rc.by0 + (double)cy * wy, yhi = cy == kRG - 1 ? INF : rc.by0 + (double)(cy + 1) * wy;
    double gx = qx < xlo ? xlo - qx : (qx > xhi ? qx - xhi : 0.0), gy = qy < ylo ? ylo - qy : (qy > yhi ? qy - yhi : 0.0);
    const double eps = 1e-9 * (1.0 + fabs(qx) + fabs(qy));
    gx = gx > eps ? gx - eps : 0.0;
    gy = gy > eps ? gy - eps : 0.0;
    return gx * gx + gy * gy;
}

// KdTree::nearest_neighbor (nearest_neighbor.rs:48-92) for a group: the exact minimum of (norm2, id) over the region pages.
// Regions are taken nearest first (by the distance from the sample to the region's rectangle) and the walk ends when the
// next one cannot hold a node as near as the best so far.  Stage A: the 4 x 4 block of regions around the sample, one
// lane each (their counts are one load); it settles the search whenever the best node is nearer than the block's border --
// always, once the tree covers the map.  Stage B (a thin tree, or a sample far from it): the occupied regions of the whole
// grid from the occupancy bitmap, 128 regions per lane.
template <int GL>
__device__ __forceinline__ void group_nn(const RunConst &rc, uint32_t b, const GTeam<GL> &tm, uint32_t N, double sqx, double sqy, int &nn, double &fx,
                                         double &fy) {
    static_assert(GL >= 16, "a 4 x 4 block of regions, one lane each (8 x 8 on a fine grid: four each)");
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    double bestD = INF, bestx = 0.0, besty = 0.0;
    int best = 0x7FFFFFFF;
    // thr: no node with d2 above it can win or tie (sqrt is monotone; the factor keeps rounded ties in), so the
    // sqrt -- the expensive part -- is only taken for the few nodes that may improve the lane's best
    double thr = INF;
    auto visit = [&](double x, double y, int id, bool ok) {
        if (!ok) return;
        const double d2 = dist2(x, y, sqx, sqy);
        if (d2 > thr) return;
        const double D = sqrt(d2);                           // the reference compares rounded distances
        if (D < bestD || (D == bestD && id < best)) { bestD = D; best = id; bestx = x; besty = y; thr = d2 * (1.0 + 1e-15); }
    };
    auto group_best = [&]() {
        double rd = bestD;
        int ri = best;
        tm.argmin(rd, ri);
        // the lane that holds the winner hands over its threshold and the node's coordinates
        const unsigned long long own = tm.ballot(best == ri && bestD == rd);
        const int src = own ? (int)__builtin_ctzll(own) : 0;
        thr = tm.shfl(thr, src);
        bestx = tm.shfl(bestx, src); besty = tm.shfl(besty, src);
        bestD = rd; best = ri;
    };
    auto gcnt = as_global(rc.rg_cnt) + (b & 1u) * kRegions;
    int rx, ry;
    rep_cell(rc, sqx, sqy, kRG, rx, ry);
    const double rwx = 1.0 / (rc.binv_w * (double)kRG), rwy = 1.0 / (rc.binv_h * (double)kRG);      // a region's width and height (two divisions, once)
    // ---- stage A: the block [ax0, ax0 + SB) x [ay0, ay0 + SB) of regions with the sample's region in its middle; the block
    // covers about the same area whatever the grid (a finer grid makes the radius searches cheaper and would otherwise
    // send most nearest-neighbour searches on to stage B)
    constexpr int SB = kRG >= 56 ? 8 : 4, NBLK = SB * SB, RPL = (NBLK + GL - 1) / GL;         // regions per lane
    const double fcx = (sqx - rc.bx0) * rc.binv_w * (double)kRG - (double)rx, fcy = (sqy - rc.by0) * rc.binv_h * (double)kRG - (double)ry;
    int ax0 = rx - (fcx < 0.5 ? SB / 2 : SB / 2 - 1), ay0 = ry - (fcy < 0.5 ? SB / 2 : SB / 2 - 1);
    ax0 = ax0 < 0 ? 0 : (ax0 > kRG - SB ? kRG - SB : ax0);
    ay0 = ay0 < 0 ? 0 : (ay0 > kRG - SB ? kRG - SB : ay0);
    {
        uint32_t reg[RPL], cnt[RPL];
        double gap[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            const int idx = (int)tm.gl + q * GL;
            const bool in = idx < NBLK;
            const int cx = ax0 + (in ? idx % SB : 0), cy = ay0 + (in ? idx / SB : 0);
            reg[q] = (uint32_t)(cy * kRG + cx);
            cnt[q] = in ? gcnt[reg[q]] : 0u;
            gap[q] = cnt[q] ? region_gap2(rc, sqx, sqy, cx, cy, rwx, rwy) : INF;
        }
        for (;;) {
            double g = gap[0];
            uint32_t mreg = reg[0], mcnt = cnt[0];
            int mq = 0;
#pragma unroll
            for (int q = 1; q < RPL; ++q) if (gap[q] < g) { g = gap[q]; mreg = reg[q]; mcnt = cnt[q]; mq = q; }
            int who = (int)tm.gl;
            tm.argmin(g, who);
            if (!(g <= thr) || g == INF) break;              // nothing left in the block that could hold a node as near
#if defined(PORRT_TIMING) && PORRT_TIMING == 1
            if (tm.gl == 0) atomicAdd(&rc.cnt->tim[5], 100ull);
#endif
            gscan_region<GL>(rc, tm, tm.shfl(mreg, who), tm.shfl(mcnt, who), visit);
            group_best();
            if ((int)tm.gl == who) {
#pragma unroll
                for (int q = 0; q < RPL; ++q) if (q == mq) gap[q] = INF;
            }
        }
    }
    // settled if no node outside the block can be as near: the gap to the block's complement (border regions reach to infinity)
    {
        const double wx = rwx, wy = rwy;
        const double eps = 1e-9 * (1.0 + fabs(sqx) + fabs(sqy));
        double out = INF;
        if (ax0 > 0) { const double d = sqx - (rc.bx0 + (double)ax0 * wx) - eps; out = d < out ? d : out; }
        if (ax0 + SB < kRG) { const double d = (rc.bx0 + (double)(ax0 + SB) * wx) - sqx - eps; out = d < out ? d : out; }
        if (ay0 > 0) { const double d = sqy - (rc.by0 + (double)ay0 * wy) - eps; out = d < out ? d : out; }
        if (ay0 + SB < kRG) { const double d = (rc.by0 + (double)(ay0 + SB) * wy) - sqy - eps; out = d < out ? d : out; }
        out = out > 0.0 ? out : 0.0;
#if defined(PORRT_TIMING) && PORRT_TIMING == 1
        if (tm.gl == 0) { atomicAdd(&rc.cnt->tim[13], 1ull); if (!(out * out > thr)) atomicAdd(&rc.cnt->tim[14], 1ull); }
#endif
        if (out * out > thr) { nn = best; fx = bestx; fy = besty; return; }
    }
    // ---- stage B: the occupied regions outside the block, a row of regions at a time (lane l: rows l, l + GL, ...).  Along a row
    // the gap grows with the distance in x from the sample's column, so a row's nearest pending region is one of two bits of its
    // occupancy: the first at or right of the column, the last left of it.  A lane keeps the better of the two per row and looks
    // again only when its row lost a region; a side whose nearest region is already too far is dropped whole (thr only shrinks).
    auto gocc = as_global(rc.rg_occ) + (b & 1u) * kOccWords;
    constexpr int ROWS = (kRG + GL - 1) / GL;
    static_assert(kRG <= 64, "a row of regions is one word of bits");
    unsigned long long rowbits[ROWS];
    double rgap[ROWS];
    int rreg[ROWS];
    const unsigned long long below = (1ull << rx) - 1ull;       // the columns left of the sample's
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        const int y = (int)tm.gl + q * GL;
        unsigned long long m = 0ull;
        if (y < kRG) {
            const uint32_t o = (uint32_t)y * (uint32_t)kRG, wi = o >> 6, sh = o & 63u;
            m = gocc[wi] >> sh;
            if (sh + (uint32_t)kRG > 64u) m |= gocc[wi + 1u] << (64u - sh);
            m &= (1ull << kRG) - 1ull;
            if (y >= ay0 && y < ay0 + SB) m &= ~((((1ull << SB) - 1ull)) << ax0);      // (the block's regions are done)
        }
        rowbits[q] = m;
    }
    auto row_best = [&](int q, unsigned long long &m) {
        const int y = (int)tm.gl + q * GL;
        double g = INF;
        int r = -1;
        const unsigned long long hi = m >> rx;
        if (hi) {
            const int cx = rx + (int)__builtin_ctzll(hi);
            const double gg = region_gap2(rc, sqx, sqy, cx, y, rwx, rwy);
            if (gg <= thr) { g = gg; r = y * kRG + cx; }
            else m &= below;
        }
        const unsigned long long lo = m & below;
        if (lo) {
            const int cx = 63 - (int)__builtin_clzll(lo);
            const double gg = region_gap2(rc, sqx, sqy, cx, y, rwx, rwy);
            if (!(gg <= thr)) m &= ~below;
            else if (gg < g) { g = gg; r = y * kRG + cx; }
        }
        rgap[q] = g; rreg[q] = r;
    };
#pragma unroll
    for (int q = 0; q < ROWS; ++q) row_best(q, rowbits[q]);
    for (;;) {
        double g = rgap[0];
        int mine = rreg[0], mq = 0;
#pragma unroll
        for (int q = 1; q < ROWS; ++q) if (rgap[q] < g) { g = rgap[q]; mine = rreg[q]; mq = q; }
        int who = (int)tm.gl;
        tm.argmin(g, who);
        if (!(g <= thr) || g == INF) break;
#if defined(PORRT_TIMING) && PORRT_TIMING == 1
        if (tm.gl == 0) atomicAdd(&rc.cnt->tim[6], 100ull);
#endif
        const uint32_t reg = (uint32_t)tm.shfl(mine, who);
        gscan_region<GL>(rc, tm, reg, gcnt[reg], visit);
        group_best();
        if ((int)tm.gl == who) {
#pragma unroll
            for (int q = 0; q < ROWS; ++q)
                if (q == mq) { rowbits[q] &= ~(1ull << (reg % (uint32_t)kRG)); row_best(q, rowbits[q]); }
        }
    }
    nn = best; fx = bestx; fy = besty;
    if (best == 0x7FFFFFFF) { nn = 0; fx = as_global(rc.nx)[0]; fy = as_global(rc.ny)[0]; }   // (cannot happen without a filter: the root exists)
}

// RRT* step, first kernel: GL lanes per sample.  grid.x = ceil(nb / SPB) search workgroups + ceil(cnb / SPB) workgroups
// running the rewire phase 2 of step cb (commit_rrt_sample), SPB = 256 / GL samples per workgroup.
#ifndef PORRT_NN2_WAVES
#define PORRT_NN2_WAVES 5
#endif
template <int GL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PORRT_NN2_WAVES, 8))) void k_nn2(const RunConst *__restrict__ rcp, uint32_t b, uint32_t i0, uint32_t nb, uint32_t vwords, uint32_t cb,
                                             uint32_t cnb) {
    static_assert(GL == 16 || GL == 32 || GL == 64, "group size");
    constexpr uint32_t SPB = 256u / GL;
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    xcd_swizzle(bx, by);
    const RunConst &rc = rcp[by];               // one context per grid row (porrt_grow_batch)
    GTeam<GL> tm;
    tm.gl = threadIdx.x % GL;
    tm.base = (threadIdx.x & 63u) - tm.gl;
    const uint32_t near_blocks = (nb + SPB - 1u) / SPB;
    if (bx >= near_blocks) {
        const uint32_t ck = (bx - near_blocks) * SPB + threadIdx.x / GL;
        PORRT_T0();
        if (ck < cnb && ck < row_nb(rc, cb, cnb)) commit_rrt_sample(rc, cb, vwords, ck, tm.gl, GL);
        PORRT_TACC_A(rc, 4);
        return;
    }
    const uint32_t slot = bx * SPB + threadIdx.x / GL;
    if (slot >= row_nb(rc, b, nb)) return;
    PORRT_T0();
    const size_t so = (size_t)b * rc.part_stride + slot;
    const uint32_t k = as_global(rc.perm)[so];
    const uint32_t N = as_global(rc.n_at)[b];
    const double sqx = as_global(rc.ssx)[so], sqy = as_global(rc.ssy)[so];
    (void)i0;
    // Most samples of a grown tree are not steered: steer() (common.rs:215-225) moves the sample only if the L1 distance to
    // its nearest node (by L2) exceeds max_step, and L1 <= sqrt(2) L2 <= sqrt(2) * (L2 distance to ANY node).  So a node
    // within 0.7 max_step (< max_step / sqrt(2), margin for the roundings) proves that the new state is the sample itself,
    // whichever node is nearest -- and that node is only needed if no neighbour turns out to be connectable (rrt.rs:132-134;
    // k_conn2 searches it then).  The bound pyramid names a node in the 3x3 finest cells around the sample, if there is one.
    int nn = -1;
    double fx = 0.0, fy = 0.0;
    bool easy = false;
    // ... and within the step's search radius, so that the radius search of k_conn2 finds it too (a sample without any neighbour
    // needs the nearest node after all)
    const double lim = 0.7 * rc.max_step, t2b = as_global(rc.t2_at)[b];
    const double lim2 = lim * lim < t2b ? lim * lim : t2b;
    // the class of the sample's own pixel, asked for now: most samples are not steered, and their validity would otherwise be a
    // trip to the raster of its own at the end
    uint32_t err = 0;
    int cls_own = CLS_FREE;
    if (rc.has_grid) cls_own = state_class(rc, sqx, sqy, &err);
    // The cells carry their node's position rounded to f32 (rep_f): every node they name is older than this step (the pyramid is
    // written by the connect kernels of the steps before), so a cell that is not empty is a node of the snapshot, and its true position
    // lies within perr of the rounded one -- the test is made that much stricter.  (An empty cell is a NaN: no comparison holds.)
    const double perr = 1.3e-7 * (fabs(sqx) + fabs(sqy) + 2.0 * lim);
    const double limf = lim - perr, limf2 = limf > 0.0 ? (limf * limf < t2b ? limf * limf : t2b * (1.0 - 1e-6)) : -1.0;
    (void)lim2;
    for (int l = 0; l < 2 && !easy; ++l) {           // the finest cells; then, for a thin tree, the 3x3 cells of the next level
        int cx, cy;
        const int G = rep_dim(l);
        rep_cell(rc, sqx, sqy, G, cx, cy);
        bool near = false;
        if (tm.gl < 9u) {
            const int x = cx + (int)(tm.gl % 3u) - 1, y = cy + (int)(tm.gl / 3u) - 1;
            if (x >= 0 && y >= 0 && x < G && y < G) {
                const flt2 pf = as_global(rc.rep_f)[rep_off(l) + y * G + x];
                near = dist2((double)pf.x, (double)pf.y, sqx, sqy) <= limf2;
            }
        }
        easy = tm.ballot(near) != 0ull;
    }
    PORRT_TACC_A(rc, 0);
    if (!easy) group_nn<GL>(rc, b, tm, N, sqx, sqy, nn, fx, fy);
    PORRT_TACC_A(rc, 1);
    double tx = sqx, ty = sqy;
    if (!easy) {
        // common.rs:215-225
        double step = fabs(tx - fx);
        step += fabs(ty - fy);
        if (step > rc.max_step) {
            const double lambda = rc.max_step / step;
            double ux = (tx - fx) * lambda, uy = (ty - fy) * lambda;
            tx = fx + ux;
            ty = fy + uy;
        }
    }
    bool valid = true;
    if (rc.has_grid) {
        int cls = cls_own;
        if (tx != sqx || ty != sqy) { err = 0; cls = state_class(rc, tx, ty, &err); }      // steered: the new state's pixel
        valid = cls == CLS_FREE && !err;                         // RTTFuncs adapter (tamp_rrt.rs:40-42)
    }
    PORRT_TACC_A(rc, 2);
    if (tm.gl == 0) {
        as_global(rc.q_x)[k] = tx;
        as_global(rc.q_y)[k] = ty;
        // copy for the kd insertion, which runs beside the following steps (one slice per step)
        const size_t o2 = (size_t)b * rc.part_stride + k;
        as_global(rc.kq_x)[o2] = tx; as_global(rc.kq_y)[o2] = ty; as_global(rc.kq_vid)[o2] = valid ? 0 : -1;
        as_global(rc.q_nn)[k] = nn;
        // A node exactly on the goal point (every 100th iteration re-adds it once it is reached, rrt.rs:176-181) has every
        // earlier copy as a neighbour, and a step's copies are identical searches: they are served together, once
        // (q_vid = 1: the clone workgroup of k_conn2).
        int qv = valid ? 0 : -1;
        if (valid && rc.goal_kind != 0 && tx == rc.gp_x && ty == rc.gp_y) {
            const uint32_t slot = atomicAdd(&rc.cnt->clone_n, 1u);
            if (slot < 64u) { rc.cnt->clone_k[slot] = k; qv = 1; }
        }
        as_global(rc.q_vid)[k] = qv;
        // what the connect groups read, in slot order
        as_global(rc.bq_x)[slot] = tx; as_global(rc.bq_y)[slot] = ty; as_global(rc.bq_k)[slot] = qv == 0 ? (uint16_t)k : (uint16_t)0xFFFFu;
        if (valid) atomicOr(&rc.valid_mask[(size_t)b * vwords + (k >> 6)], 1ull << (k & 63u));
        if (err) atomicOr(&rc.cnt->err, err);
    }
}

// dynamic LDS of k_conn2: per sample the LDS part of its hit list; the page-filing workgroup uses the same bytes as its
// scratch
__host__ __device__ inline size_t conn2_lds_bytes(uint32_t GL) {
    const size_t a = 4u * (64u / 16u) * kHitBytes;      // per workgroup whatever the group size: a wave's lists hold 4 x kLdsHits hits between them
    static_assert(kGTrackLds <= 4u * (64u / 16u) * kHitBytes, "g_track_step uses the same bytes");
    return a > kInsertLds ? a : kInsertLds;
}

// (the connect pass needs these rarely: functions of their own, so that their registers are not the step kernel's)
template <int GL>
__device__ __attribute__((noinline)) int group_nn_call(const RunConst &rc, uint32_t b, uint32_t gl, uint32_t base, uint32_t N, double px, double py) {
    GTeam<GL> tm;
    tm.gl = gl; tm.base = base;
    int nn;
    double fx, fy;
    group_nn<GL>(rc, b, tm, N, px, py, nn, fx, fy);
    return nn;
}
// nearest node of (px, py) by all 64 lanes of the calling wave (px, py wave-uniform)
__device__ __attribute__((noinline)) int wave_nn(const RunConst &rc, uint32_t b, uint32_t N, double px, double py) {
    GTeam<64> t64;
    t64.gl = threadIdx.x & 63u;
    t64.base = 0;
    int nn;
    double fx, fy;
    group_nn<64>(rc, b, t64, N, px, py, nn, fx, fy);
    return nn;
}

// one wave: radius search around (pxh, pyh) into the global list of sample kh (the form k_near wrote for every sample)
__device__ __forceinline__ void list_scan_wave(const RunConst &rc, uint32_t b, uint32_t N, double T2, uint32_t kh, double pxh, double pyh, uint32_t lane,
                                               uint32_t &err) {
    auto cid = as_global(rc.cand_id) + cand_off(rc, b, kh);
    auto cxy = as_global(reinterpret_cast<dbl2 *>(rc.cand_xy)) + cand_off(rc, b, kh);
    const uint32_t cap = rc.cand_cap;
    uint32_t tot = 0;
    bool over = false;
    scan_disc(rc, b, pxh, pyh, disc_radius(T2, pxh, pyh), uni(N), lane, [&](double x, double y, int jd, bool ok) {
        const bool in = ok && dist2(x, y, pxh, pyh) <= T2;
        const unsigned long long hm = __ballot(in);
        const uint32_t pos = tot + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
        if (in) {
            if (pos < cap) {
                cid[pos] = jd;
                dbl2 v;
                v.x = x; v.y = y;
                cxy[pos] = v;
            } else {
                over = true;
            }
        }
        tot += (uint32_t)__popcll(hm);
    });
    if (lane == 0) as_global(rc.cand_cnt)[cand_cnt_at(rc, b, kh)] = tot;
    if (over) err |= (uint32_t)ERR_CAND_OVERFLOW;
}

// The clone workgroup of k_conn2: the step's copies of the goal point, one search for all of them by the 4-wave team.
// A function of its own (not inlined): its registers are not the step kernel's.
__device__ __attribute__((noinline)) void clone_workgroup(const RunConst &rc, uint32_t b, uint32_t vwords, uint32_t N, double T2) {
    __shared__ double s_d[kConnectWaves];
    __shared__ int s_i[kConnectWaves];
    __shared__ uint32_t s_cid[64], s_ck[64];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t err = 0;
    uint32_t n = rc.cnt->clone_n;
    n = n < 64u ? n : 64u;
    if (n == 0) return;
    if (threadIdx.x < n) {
        const uint32_t kc = rc.cnt->clone_k[threadIdx.x];
        s_ck[threadIdx.x] = kc;
        s_cid[threadIdx.x] = N + rank_before(rc, b, vwords, kc);
    }
    __syncthreads();
    if (threadIdx.x == 0) {                     // the lowest id leads (its rewires are the strict improvements)
        uint32_t lead = 0;
        for (uint32_t t = 1; t < n; ++t) if (s_cid[t] < s_cid[lead]) lead = t;
        const uint32_t ti = s_cid[0], tk = s_ck[0];
        s_cid[0] = s_cid[lead]; s_ck[0] = s_ck[lead]; s_cid[lead] = ti; s_ck[lead] = tk;
        rc.cnt->clone_n = 0;                    // for the next step
    }
    __syncthreads();
    const uint32_t kh = uni(s_ck[0]), idh = uni(s_cid[0]);
    if (threadIdx.x >= 1 && threadIdx.x < n) as_global(rc.cand_cnt)[cand_cnt_at(rc, b, s_ck[threadIdx.x])] = 0u;     // no rewire candidates of their own
    const double pxh = uni_d(as_global(rc.q_x)[kh]), pyh = uni_d(as_global(rc.q_y)[kh]);
    if (wv == 0) {
        list_scan_wave(rc, b, N, T2, kh, pxh, pyh, lane, err);
        __threadfence();
    }
    __syncthreads();
    const uint32_t hc = cand_count(rc, b, kh);
    TableGrid grid;            // the few rays outside the clearance window read the raster in memory
    grid.p = rc.cls; grid.W = rc.W;
    Team<kConnectWaves> tmh;
    tmh.scr_d = s_d; tmh.scr_i = s_i; tmh.wave = wv; tmh.lane = lane;
    connect_rrt_sample(rc, tmh, global_list(rc, b, kh), grid, b, kh, idh, pxh, pyh, hc, err, s_cid + 1, n - 1u, -1,
                       [&]() { return wave_nn(rc, b, N, pxh, pyh); });       // (every wave of the team finds the same node)
    if (err) atomicOr(&rc.cnt->err, err);
}

__device__ __forceinline__ MemHits mem_hits(const RunConst &rc, uint32_t b, uint32_t k) {
    MemHits M;
    M.sid = as_global(rc.cand_id) + cand_off(rc, b, k);
    M.sxy = as_global(reinterpret_cast<dbl2 *>(rc.cand_xy)) + cand_off(rc, b, k);
    M.sd = as_global(rc.cand_val) + cand_val_off(rc, b, k);
    M.out_id = M.sid;
    M.out_val = M.sd;
    M.out_cnt = as_global(rc.cand_cnt) + cand_cnt_at(rc, b, k);
    M.out_cap = rc.cand_cap;
    return M;
}

// One sample with more hits than its LDS list holds, served by the whole wave (a function of its own: its registers are not the
// step kernel's).  Its hits are in its slice of the lists in memory; when the wave's four LDS lists together hold them (4 x kLdsHits:
// most such samples) they are read back into those lists in one coalesced pass and the three passes of the connect run on LDS.
template <int GL>
__device__ __attribute__((noinline)) void heavy_sample_wave(const RunConst &rc, uint32_t b, uint32_t N, uint32_t kh, uint32_t idh, uint32_t toth, int clrh,
                                                            double pxh, double pyh, uint8_t *wb, uint32_t &err) {
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t kWaveHits = (64u / (uint32_t)GL) * (kLdsHits * (uint32_t)(GL / 16));
    Team<1> tw;
    tw.scr_d = nullptr; tw.scr_i = nullptr; tw.wave = 0; tw.lane = lane;
    TableGrid grid;
    grid.p = rc.cls; grid.W = rc.W;
    const uint32_t cap = rc.cand_cap;
    const MemHits Mh = mem_hits(rc, b, kh);
    PORRT_T0();
    if (toth <= kWaveHits) {
        LdsHits Wl;
        Wl.hx = reinterpret_cast<double *>(wb);
        Wl.hy = Wl.hx + kWaveHits;
        Wl.hd = Wl.hy + kWaveHits;
        Wl.hid = reinterpret_cast<int *>(Wl.hd + kWaveHits);
        Wl.out_id = Mh.out_id; Wl.out_val = Mh.out_val; Wl.out_cnt = Mh.out_cnt; Wl.out_cap = Mh.out_cap;
        for (uint32_t a0 = 0; a0 < toth; a0 += 256u) {              // four entries per lane in flight
            int jd[4];
            dbl2 v[4];
            double d[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const uint32_t a = a0 + u * 64u + lane;
                jd[u] = a < toth ? Mh.sid[a] : 0;
                v[u] = a < toth ? Mh.sxy[a] : dbl2{0.0, 0.0};
                d[u] = a < toth ? Mh.sd[a] : 0.0;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const uint32_t a = a0 + u * 64u + lane;
                if (a < toth) { Wl.hid[a] = jd[u]; Wl.hx[a] = v[u].x; Wl.hy[a] = v[u].y; Wl.hd[a] = d[u]; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        PORRT_TACC_H(rc, 0);
        connect_rrt_sample(rc, tw, Wl, grid, b, kh, idh, pxh, pyh, toth, err, nullptr, 0, clrh, [&]() { return wave_nn(rc, b, N, pxh, pyh); });
        __builtin_amdgcn_wave_barrier();
        PORRT_TACC_H(rc, 3);
    } else {
        connect_rrt_sample(rc, tw, Mh, grid, b, kh, idh, pxh, pyh, toth < cap ? toth : cap, err, nullptr, 0, clrh,
                           [&]() { return wave_nn(rc, b, N, pxh, pyh); });
    }
}

#ifndef PORRT_CONN2_WAVES
#define PORRT_CONN2_WAVES 4
#endif
template <int GL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PORRT_CONN2_WAVES, 8))) void k_conn2(const RunConst *__restrict__ rcp, uint32_t b, uint32_t nb, uint32_t vwords, uint32_t lazy) {
    static_assert(GL == 16 || GL == 32 || GL == 64, "group size");
    constexpr uint32_t SPB = 256u / GL;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_dyn[];
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    const uint32_t ns = lazy ? 3u : 2u;         // the rows' special workgroups: page filing, goal-point copies, (lazy) the goal path of the kd order
    const uint32_t role = xcd_swizzle_roles(bx, by, ns);
    const RunConst &rc = rcp[by];               // one context per grid row (porrt_grow_batch)
    nb = row_nb(rc, b, nb);
    if (nb == 0) return;                        // (a row that has stopped, or does not run this step)
    if (role == 0) { insert_step_pages(rc, b, nb, vwords, lds_dyn); return; }    // the page-filing workgroup
    if (lazy && role == 2u) { g_track_step<1>(rc, b, nb, vwords, lds_dyn); return; }
    const uint32_t lane = threadIdx.x & 63u, si = threadIdx.x / GL;
    const uint32_t slot = bx * SPB + si;
    // first round trip: everything that depends on nothing
    const uint32_t N = as_global(rc.n_at)[b];
    const double T2 = as_global(rc.t2_at)[b];              // rad_T2[N], rrt.rs:121: the size before insertion
    uint32_t k = 0xFFFFu;
    double px = 0.0, py = 0.0;
    if (role == ns && slot < nb) { k = as_global(rc.bq_k)[slot]; px = as_global(rc.bq_x)[slot]; py = as_global(rc.bq_y)[slot]; }
    if (role == 1) { clone_workgroup(rc, b, vwords, N, T2); return; }
    GTeam<GL> tm;
    tm.gl = threadIdx.x % GL;
    tm.base = lane - tm.gl;
    const bool act = k != 0xFFFFu;                         // else: no sample, an invalid one, or a copy of the goal point (clone workgroup)
    uint32_t err = 0;
    PORRT_T0();
    uint32_t id = 0, tot = 0;
    int clr_b = 0;
    bool heavy = false;
    const uint32_t cap = rc.cand_cap;
    TableGrid grid;            // the few rays outside the clearance window read the raster in memory
    grid.p = rc.cls; grid.W = rc.W;
    if (act) {
        // second round trip: the new node's id and the clearance around its pixel, beside the region counts of the search
        id = N + rank_before_lanes(rc, b, vwords, k, tm.gl, (uint32_t)GL);
        if (rc.has_grid) {
            uint32_t bi, bj;
            to_pixel(rc, px, py, bi, bj);
            if (bi < rc.H && bj < rc.W) clr_b = as_global(rc.clr)[bi * rc.W + bj];
        }
        // a sample's list: the wave's LDS share split between its 64 / GL samples (80 hits at 16 lanes per sample, 320 at 64)
        constexpr uint32_t kHits = kLdsHits * (uint32_t)(GL / 16);
        uint8_t *hb = lds_dyn + si * (kHits * 28u);
        LdsHits L;
        L.hx = reinterpret_cast<double *>(hb);
        L.hy = L.hx + kHits;
        L.hd = L.hy + kHits;
        L.hid = reinterpret_cast<int *>(L.hd + kHits);
        const MemHits M = mem_hits(rc, b, k);
        L.out_id = M.out_id; L.out_val = M.out_val; L.out_cnt = M.out_cnt; L.out_cap = M.out_cap;
        PORRT_TACC_B(rc, 0);
        gscan_disc<GL, 1, true>(rc, b, tm, px, py, disc_radius(T2, px, py), N, [&](double x, double y, double dA, int jd, bool ok) {
            const bool in = ok && dist2(x, y, px, py) <= T2;
            const unsigned long long hm = tm.ballot(in);
            const uint32_t pos = tot + (uint32_t)__popcll(hm & ((1ull << tm.gl) - 1ull));
            if (in) {
                if (pos < kHits) { L.hid[pos] = jd; L.hx[pos] = x; L.hy[pos] = y; L.hd[pos] = dA; }
                else if (pos < cap) { M.sid[pos] = jd; dbl2 v; v.x = x; v.y = y; M.sxy[pos] = v; M.sd[pos] = dA; }
                else err |= (uint32_t)ERR_CAND_OVERFLOW;
            }
            tot += (uint32_t)__popcll(hm);
        });
        __builtin_amdgcn_wave_barrier();
        PORRT_TACC_B(rc, 1);
        if (tot <= kHits) {
            connect_rrt_sample(rc, tm, L, grid, b, k, id, px, py, tot, err, nullptr, 0, clr_b,
                               [&]() { return group_nn_call<GL>(rc, b, tm.gl, tm.base, N, px, py); });
            PORRT_TACC_B(rc, 2);
        } else {
            // more hits than LDS holds (the dense start of a tree, the neighbourhood of the goal point): the first ones
            // join the rest in memory, and the whole wave serves the sample below
            heavy = true;
            if (tm.gl == 0) atomicAdd(&rc.cnt->n_heavy, 1u);
            for (uint32_t a = tm.gl; a < kHits; a += (uint32_t)GL) {
                M.sid[a] = L.hid[a];
                dbl2 v;
                v.x = L.hx[a]; v.y = L.hy[a];
                M.sxy[a] = v;
                M.sd[a] = L.hd[a];
            }
        }
    }
    unsigned long long hv = __ballot(heavy && tm.gl == 0);
    if (hv) {
        // written by this wave: its own loads see the entries once its stores are counted down
        __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
        __builtin_amdgcn_wave_barrier();
        uint8_t *wb = lds_dyn + (threadIdx.x >> 6) * (64u / (uint32_t)GL) * ((kLdsHits * (uint32_t)(GL / 16)) * 28u);      // the wave's lists as one (their samples are through)
        while (hv) {
            const int l = (int)__builtin_ctzll(hv);
            hv &= hv - 1;
            const uint32_t kh = uni((uint32_t)__shfl((int)k, l)), idh = uni((uint32_t)__shfl((int)id, l)), toth = uni((uint32_t)__shfl((int)tot, l));
            const int clrh = (int)uni((uint32_t)__shfl(clr_b, l));
            const double pxh = uni_d(__shfl(px, l)), pyh = uni_d(__shfl(py, l));
            heavy_sample_wave<GL>(rc, b, N, kh, idh, toth, clrh, pxh, pyh, wb, err);
        }
        PORRT_TACC_B(rc, 3);
    }
    if (err) atomicOr(&rc.cnt->err, err);
}

// stand-alone rewire phase 2 for the last step of a launch sequence, GL lanes per sample
template <int GL>
__global__ __launch_bounds__(256) void k_commit2(const RunConst *__restrict__ rcp, uint32_t b, uint32_t nb, uint32_t vwords) {
    const uint32_t k = (blockIdx.x * 256u + threadIdx.x) / GL;
    if (k < nb && k < row_nb(rcp[blockIdx.y], b, nb)) commit_rrt_sample(rcp[blockIdx.y], b, vwords, k, threadIdx.x % GL, GL);
}

} // namespace porrt
