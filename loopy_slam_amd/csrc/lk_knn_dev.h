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
#ifndef LK_KNN_STAMP
#define LK_KNN_STAMP(I) do {} while (0)
#define LK_KNN_STAMPW(I) do {} while (0)
#endif

__device__ __forceinline__ int lk_cell_coord(float x, float o, float inv, int d) {
    float f = floorf((x - o) * inv);
    f = fminf(fmaxf(f, 0.0f), (float)(d - 1));
    return (int)f;
}

template <int V> struct LkInt { static constexpr int value = V; };
// A candidate is ONE 64-bit key: (bits of d2) << 32 | index.  d2 >= +0, so the unsigned order of the float bits is the
// float order and the key order is the strict total order (d2, index) in a single v_cmp_lt_u64.
// The empty slot is (FLT_MAX, -1) = the largest key any list ever holds.
#define LK_KEY_EMPTY 0x7f7fffffffffffffull
__device__ __forceinline__ uint64_t lk_key(float d2, int idx) {
    return ((uint64_t)__float_as_uint(d2) << 32) | (uint32_t)idx;
}

// insert into the ascending 8-list, dropping the largest: k[s] <- max(k[s-1], min(key, k[s])); branch-free
// FILL: the list holds at most FILL real entries when the call is made (the rest are LK_KEY_EMPTY) - the stages above slot FILL would move
// EMPTY over EMPTY and are left out: the j-th candidate a lane meets costs j stages, not 7
template <int FILL = LK_K>
__device__ __forceinline__ void lk_top8_insert(uint64_t (&k)[LK_K], uint64_t key) {
#pragma unroll
    for (int s = (FILL < LK_K - 1 ? FILL : LK_K - 1); s > 0; --s) {
        const uint64_t lo = key < k[s] ? key : k[s];
        k[s] = key < k[s - 1] ? k[s - 1] : lo;
    }
    k[0] = key < k[0] ? key : k[0];
}

// a candidate enters the insertion network only if it is inside the radius AND ahead of the list's current last entry (the order is
// total, so this is exactly "belongs to the 8 smallest so far"): in a dense cloud most in-radius candidates arrive when the list is
// already full of nearer ones, and the 8-stage network is the search's largest VALU item
template <int FILL = LK_K>
__device__ __forceinline__ void lk_top8_offer(uint64_t (&k)[LK_K], float d2, float r2, int idx) {
    const uint64_t key = lk_key(d2, idx);
    if (d2 <= r2 && key < k[LK_K - 1]) lk_top8_insert<FILL>(k, key);
}

// one butterfly round: my ascending 8-list against the partner's, keep the 8 smallest, re-sort (see the merge note below)
template <int CTRL>
__device__ __forceinline__ void lk_knn_merge_round(uint64_t (&k)[LK_K]) {
#pragma unroll
    for (int j = 0; j < LK_K / 2; ++j) {           // partner's list reversed against mine, in place
        const uint64_t hi = lk_dpp_u64<CTRL>(k[LK_K - 1 - j]), lo = lk_dpp_u64<CTRL>(k[j]);
        k[j] = hi < k[j] ? hi : k[j];
        k[LK_K - 1 - j] = lo < k[LK_K - 1 - j] ? lo : k[LK_K - 1 - j];
    }
#pragma unroll
    for (int st = LK_K / 2; st > 0; st >>= 1) {
#pragma unroll
        for (int j = 0; j < LK_K; ++j) {
            if ((j & st) == 0) {
                const uint64_t x = k[j], y = k[j + st];
                k[j] = y < x ? y : x;
                k[j + st] = y < x ? x : y;
            }
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
    uint64_t k[LK_K];
#pragma unroll
    for (int j = 0; j < LK_K; ++j) k[j] = LK_KEY_EMPTY;
    const float ox = G->ox, oy = G->oy, oz = G->oz, inv = G->inv_cell;
    const int dx = G->dx, dy = G->dy, dz = G->dz;
    const float rfull = sqrtf(r2) * 1.0001f + 1e-6f;   // box slightly inflated: never misses a cell
    // Radius well above the cell edge (the dynamic query radius of the TUM / ScanNet configs goes up to 0.16 over 0.08-m cells: 5 x 5 rows
    // of cells, 125 cells): TWO PHASES, still exact.  Phase 1 scans the box of half-width b just under one cell edge - at most 3 x 3 rows,
    // the fast path below - which holds every point within distance b of the query; if the list is full and its last entry is nearer
    // than b, no point outside that box can enter it (its squared distance exceeds b^2 in one coordinate alone) and the search is over -
    // the usual case next to a surface.  Otherwise phase 2 walks the rest of the full box (the cells phase 1 has seen are skipped).
    const float cellf = G->cell;
    const bool big = rfull > cellf * 1.05f;
    const float r = big ? cellf * 0.9999f - 1e-6f : rfull;
    bool any = G->n > 0;
    // query box entirely outside the grid -> no neighbour
    any = any && !((qx + r - ox) * inv < 0.0f || (qx - r - ox) * inv >= (float)dx);
    any = any && !((qy + r - oy) * inv < 0.0f || (qy - r - oy) * inv >= (float)dy);
    any = any && !((qz + r - oz) * inv < 0.0f || (qz - r - oz) * inv >= (float)dz);
    // Row table: the query box covers ny x nz rows of cells.  The usual case (cell size >= radius) is at most 3 x 3
    // rows: lane `sub` fetches the [start,end) of rows sub, sub+T (independent loads, one latency), the group
    // exchanges them by shuffle, and the first T candidates of EVERY row are fetched before any of them is examined
    // (nine independent 16-B loads in flight instead of nine dependent cell_start -> point round trips).
    int ix0 = 0, ix1 = 0, iy0 = 0, iy1 = -1, iz0 = 0, iz1 = -1;
    if (any) {
        ix0 = lk_cell_coord(qx - r, ox, inv, dx); ix1 = lk_cell_coord(qx + r, ox, inv, dx);
        iy0 = lk_cell_coord(qy - r, oy, inv, dy); iy1 = lk_cell_coord(qy + r, oy, inv, dy);
        iz0 = lk_cell_coord(qz - r, oz, inv, dz); iz1 = lk_cell_coord(qz + r, oz, inv, dz);
    }
    const int ny = iy1 - iy0 + 1, nz = iz1 - iz0 + 1;
    const int nrows = any ? ny * nz : 0;
    constexpr int LK_ROWS = 9, NH = (LK_ROWS + T - 1) / T;
    const int nfast = nrows <= LK_ROWS ? nrows : 0;
    int my_s[NH], my_e[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int rr = sub + h * T;
        my_s[h] = 0; my_e[h] = 0;
        if (rr < nfast) {
            const int zz = rr / ny, yy = rr - zz * ny;
            const int row = ((iz0 + zz) * dy + iy0 + yy) * dx;
            my_s[h] = cell_start[row + ix0];
            my_e[h] = cell_start[row + ix1 + 1];
        }
    }
    LK_KNN_STAMPW(3);                                // the row table (cell_start) has arrived
    int rs[LK_ROWS], re[LK_ROWS];
    float4 c[LK_ROWS];
#pragma unroll
    for (int i = 0; i < LK_ROWS; ++i) {
        rs[i] = __shfl(my_s[i / T], i & (T - 1), T) + sub;
        re[i] = __shfl(my_e[i / T], i & (T - 1), T);
        if (rs[i] < re[i]) c[i] = sorted[rs[i]];
    }
    // (row i's candidate is at most the (i + 1)-th this lane has met)
    {
        auto first = [&](auto I) {
            constexpr int i = decltype(I)::value;
            if (rs[i] < re[i]) {
                const float d2 = lk_dist2(qx, qy, qz, c[i].x, c[i].y, c[i].z);
                lk_top8_offer<i>(k, d2, r2, __float_as_int(c[i].w));
            }
        };
        first(LkInt<0>()); first(LkInt<1>()); first(LkInt<2>()); first(LkInt<3>()); first(LkInt<4>());
        first(LkInt<5>()); first(LkInt<6>()); first(LkInt<7>()); first(LkInt<8>());
    }
    LK_KNN_STAMPW(4);                                // the first T candidates of every row are ranked
#pragma unroll
    for (int i = 0; i < LK_ROWS; ++i) {
#pragma unroll 1
        for (int t = rs[i] + T; t < re[i]; t += T) {
            const float4 p = sorted[t];
            const float d2 = lk_dist2(qx, qy, qz, p.x, p.y, p.z);
            lk_top8_offer(k, d2, r2, __float_as_int(p.w));
        }
    }
    LK_KNN_STAMPW(5);                                // ... and the rest of the rows
    if (nrows > LK_ROWS) {          // cells smaller than the radius: plain row walk
#pragma unroll 1
        for (int iz = iz0; iz <= iz1; ++iz) {
#pragma unroll 1
            for (int iy = iy0; iy <= iy1; ++iy) {
                const int row = (iz * dy + iy) * dx;
                const int s = cell_start[row + ix0];
                const int e = cell_start[row + ix1 + 1];
#pragma unroll 1
                for (int t = s + sub; t < e; t += T) {
                    const float4 p = sorted[t];
                    const float d2 = lk_dist2(qx, qy, qz, p.x, p.y, p.z);
                    lk_top8_offer(k, d2, r2, __float_as_int(p.w));
                }
            }
        }
    }
    // butterfly merge: after round m every lane holds the top-8 of its 2m-lane subgroup.  Two ascending 8-lists
    // A (mine) and B (partner's): L[i] = min(A[i], B[7-i]) is the 8 smallest of the union as a bitonic sequence,
    // which three compare-exchange stages sort.  Straight-line code (the order is total, so both partners end
    // with the identical list).
    // round r pairs every lane with one lane of the OTHER half of its 2^(r+1)-lane group: quad swaps for r = 0, 1, then
    // the mirrors of the 8- and 16-lane groups (DPP moves, no LDS traffic)
    lk_knn_merge_round<0xB1>(k);
    lk_knn_merge_round<0x4E>(k);
    lk_knn_merge_round<0x141>(k);
    if (T == 16) lk_knn_merge_round<0x140>(k);
    LK_KNN_STAMP(6);                                 // merged: every lane of the group holds the top-8
    if (big) {
        // every lane of the group holds the same list here, so the decision is group-uniform
        const float rho2 = r * r * (1.0f - 1e-6f);
        const bool done = k[LK_K - 1] != LK_KEY_EMPTY && __uint_as_float((uint32_t)(k[LK_K - 1] >> 32)) <= rho2;
        if (!done) {
            if (sub != 0) {          // one copy of the phase-1 list survives: the merge below must not meet a candidate twice
#pragma unroll
                for (int j = 0; j < LK_K; ++j) k[j] = LK_KEY_EMPTY;
            }
            bool anyf = G->n > 0;
            anyf = anyf && !((qx + rfull - ox) * inv < 0.0f || (qx - rfull - ox) * inv >= (float)dx);
            anyf = anyf && !((qy + rfull - oy) * inv < 0.0f || (qy - rfull - oy) * inv >= (float)dy);
            anyf = anyf && !((qz + rfull - oz) * inv < 0.0f || (qz - rfull - oz) * inv >= (float)dz);
            if (anyf) {
                const int fx0 = lk_cell_coord(qx - rfull, ox, inv, dx), fx1 = lk_cell_coord(qx + rfull, ox, inv, dx);
                const int fy0 = lk_cell_coord(qy - rfull, oy, inv, dy), fy1 = lk_cell_coord(qy + rfull, oy, inv, dy);
                const int fz0 = lk_cell_coord(qz - rfull, oz, inv, dz), fz1 = lk_cell_coord(qz + rfull, oz, inv, dz);
                auto walk = [&](int row, int xa, int xb) {
                    if (xa > xb) return;
                    const int s = cell_start[row + xa];
                    const int e = cell_start[row + xb + 1];
#pragma unroll 1
                    for (int t = s + sub; t < e; t += T) {
                        const float4 p = sorted[t];
                        const float d2 = lk_dist2(qx, qy, qz, p.x, p.y, p.z);
                        lk_top8_offer(k, d2, r2, __float_as_int(p.w));
                    }
                };
#pragma unroll 1
                for (int iz = fz0; iz <= fz1; ++iz) {
#pragma unroll 1
                    for (int iy = fy0; iy <= fy1; ++iy) {
                        const int row = (iz * dy + iy) * dx;
                        const bool seen = any && iz >= iz0 && iz <= iz1 && iy >= iy0 && iy <= iy1;     // phase 1 scanned [ix0, ix1] of this row
                        if (!seen) walk(row, fx0, fx1);
                        else { walk(row, fx0, ix0 - 1); walk(row, ix1 + 1, fx1); }
                    }
                }
            }
            lk_knn_merge_round<0xB1>(k);
            lk_knn_merge_round<0x4E>(k);
            lk_knn_merge_round<0x141>(k);
            if (T == 16) lk_knn_merge_round<0x140>(k);
        }
    }
#pragma unroll
    for (int j = 0; j < LK_K; ++j) { d[j] = __uint_as_float((uint32_t)(k[j] >> 32)); id[j] = (int)(uint32_t)k[j]; }
}
