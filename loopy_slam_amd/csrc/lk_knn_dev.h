// Device-side exact radius-limited top-8 search over the uniform grid (used by the
// standalone query kernel and inlined into the sample/interpolate kernel).
//
// T consecutive lanes (T = 8 or 16, a power of two <= 64) cooperate on ONE query: the candidate
// ranges are walked T points at a time (T x 16 B = one or two full 128-B lines per step, coalesced),
// every lane keeps a private sorted top-8 of the candidates it saw, and log2(T) xor-butterfly merge
// rounds leave the identical global top-8 in all T lanes.  The order is the strict total order
// (d2, index), so the result does not depend on which lane met which candidate.
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

// All T lanes of a group must call this convergently with the same query (qx,qy,qz,r2).
// x is the fastest-varying cell coordinate, so the cells [ix0..ix1] of one (iy,iz) row are ONE
// contiguous range of the cell-sorted point array.
template <int T>
__device__ __forceinline__ void lk_knn_scan_coop(const LkGrid* __restrict__ G, const float4* __restrict__ sorted,
                                                 const int32_t* __restrict__ cell_start,
                                                 float qx, float qy, float qz, float r2, int sub,
                                                 float (&d)[LK_K], int (&id)[LK_K]) {
#pragma unroll
    for (int j = 0; j < LK_K; ++j) { d[j] = LK_FLT_MAX; id[j] = -1; }
    const float ox = G->ox, oy = G->oy, oz = G->oz, inv = G->inv_cell;
    const int dx = G->dx, dy = G->dy, dz = G->dz;
    const float r = sqrtf(r2) * 1.0001f + 1e-6f;       // box slightly inflated: never misses a cell
    bool any = G->n > 0;
    // query box entirely outside the grid -> no neighbour
    any = any && !((qx + r - ox) * inv < 0.0f || (qx - r - ox) * inv >= (float)dx);
    any = any && !((qy + r - oy) * inv < 0.0f || (qy - r - oy) * inv >= (float)dy);
    any = any && !((qz + r - oz) * inv < 0.0f || (qz - r - oz) * inv >= (float)dz);
    if (any) {
        const int ix0 = lk_cell_coord(qx - r, ox, inv, dx), ix1 = lk_cell_coord(qx + r, ox, inv, dx);
        const int iy0 = lk_cell_coord(qy - r, oy, inv, dy), iy1 = lk_cell_coord(qy + r, oy, inv, dy);
        const int iz0 = lk_cell_coord(qz - r, oz, inv, dz), iz1 = lk_cell_coord(qz + r, oz, inv, dz);
        for (int iz = iz0; iz <= iz1; ++iz) {
            for (int iy = iy0; iy <= iy1; ++iy) {
                const int row = (iz * dy + iy) * dx;
                const int s = cell_start[row + ix0];
                const int e = cell_start[row + ix1 + 1];
                for (int t = s + sub; t < e; t += T) {
                    const float4 p = sorted[t];
                    const float d2 = lk_dist2(qx, qy, qz, p.x, p.y, p.z);
                    if (d2 <= r2) lk_top8_insert(d, id, d2, __float_as_int(p.w));
                }
            }
        }
    }
    // butterfly merge: after round m every lane holds the top-8 of its 2m-lane subgroup
#pragma unroll
    for (int m = 1; m < T; m <<= 1) {
        float od[LK_K];
        int oi[LK_K];
#pragma unroll
        for (int j = 0; j < LK_K; ++j) { od[j] = __shfl_xor(d[j], m); oi[j] = __shfl_xor(id[j], m); }
#pragma unroll
        for (int j = 0; j < LK_K; ++j)
            if (oi[j] >= 0) lk_top8_insert(d, id, od[j], oi[j]);
    }
}
