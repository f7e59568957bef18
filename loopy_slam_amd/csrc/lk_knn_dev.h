// Device-side exact radius-limited top-8 search over the uniform grid (used by the
// standalone query kernel and inlined into the sample/interpolate kernel).
#pragma once
#include "lk_common.h"

__device__ __forceinline__ int lk_cell_coord(float x, float o, float inv, int d) {
    float f = floorf((x - o) * inv);
    f = fminf(fmaxf(f, 0.0f), (float)(d - 1));
    return (int)f;
}

// keep the 8 smallest (d2, idx) pairs, ascending; fully unrolled so both arrays stay in VGPRs
__device__ __forceinline__ void lk_top8_insert(float (&d)[LK_K], int (&id)[LK_K], float nd, int ni) {
    if (nd < d[LK_K - 1] || (nd == d[LK_K - 1] && ni < id[LK_K - 1])) {
        d[LK_K - 1] = nd;
        id[LK_K - 1] = ni;
#pragma unroll
        for (int s = LK_K - 1; s > 0; --s) {
            const bool sw = (d[s] < d[s - 1]) || (d[s] == d[s - 1] && id[s] < id[s - 1]);
            const float td = sw ? d[s - 1] : d[s];
            const int ti = sw ? id[s - 1] : id[s];
            d[s - 1] = sw ? d[s] : d[s - 1];
            id[s - 1] = sw ? id[s] : id[s - 1];
            d[s] = td;
            id[s] = ti;
        }
    }
}

// Scan the cells overlapping the query's radius box.  x is the fastest-varying cell
// coordinate, so the cells [ix0..ix1] of one (iy,iz) row are ONE contiguous range of the
// cell-sorted point array: at most (2r/cell+2)^2 ranges per query instead of ^3 cells.
__device__ __forceinline__ void lk_knn_scan(const LkGrid* __restrict__ G, const float4* __restrict__ sorted,
                                            const int32_t* __restrict__ cell_start,
                                            float qx, float qy, float qz, float r2,
                                            float (&d)[LK_K], int (&id)[LK_K]) {
#pragma unroll
    for (int j = 0; j < LK_K; ++j) { d[j] = LK_FLT_MAX; id[j] = -1; }
    if (G->n <= 0) return;
    const float ox = G->ox, oy = G->oy, oz = G->oz, inv = G->inv_cell;
    const int dx = G->dx, dy = G->dy, dz = G->dz;
    const float r = sqrtf(r2) * 1.0001f + 1e-6f;       // box slightly inflated: never misses a cell
    // query box entirely outside the grid -> no neighbour
    if ((qx + r - ox) * inv < 0.0f || (qx - r - ox) * inv >= (float)dx) return;
    if ((qy + r - oy) * inv < 0.0f || (qy - r - oy) * inv >= (float)dy) return;
    if ((qz + r - oz) * inv < 0.0f || (qz - r - oz) * inv >= (float)dz) return;
    const int ix0 = lk_cell_coord(qx - r, ox, inv, dx), ix1 = lk_cell_coord(qx + r, ox, inv, dx);
    const int iy0 = lk_cell_coord(qy - r, oy, inv, dy), iy1 = lk_cell_coord(qy + r, oy, inv, dy);
    const int iz0 = lk_cell_coord(qz - r, oz, inv, dz), iz1 = lk_cell_coord(qz + r, oz, inv, dz);
    for (int iz = iz0; iz <= iz1; ++iz) {
        for (int iy = iy0; iy <= iy1; ++iy) {
            const int row = (iz * dy + iy) * dx;
            const int s = cell_start[row + ix0];
            const int e = cell_start[row + ix1 + 1];
            for (int t = s; t < e; ++t) {
                const float4 p = sorted[t];
                const float d2 = lk_dist2(qx, qy, qz, p.x, p.y, p.z);
                if (d2 <= r2) lk_top8_insert(d, id, d2, __float_as_int(p.w));
            }
        }
    }
}
