// Per-ray alpha composite and its backward as device functions (raw2outputs_nerf_color, src/common.py:382-422;
// Renderer.py:184-200), shared by the many-workgroup kernels (k_composite, k_composite_bwd) and the one-workgroup
// fused forms of the tracking loop (lk_loop.hip).
#pragma once
#include "lk_common.h"

struct LkRayOut { float depth, var, c0, c1, c2; bool valid; };

// occupancy of unsupported samples := -100, alpha composite, depth / variance / colour, validity (decoder.py:259-260); q = raw rows of the
// ray's samples, has = the sample has its neighbours, z = its depth along the ray
__device__ __forceinline__ LkRayOut lk_composite_vals(const float4 (&q)[LK_S_MAX], const bool (&has)[LK_S_MAX], const float (&z)[LK_S_MAX],
                                                      int S, float coef, float gt_depth) {
    float T = 1.0f, wsum = 0.0f, dsum = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    float wv[LK_S_MAX], zv[LK_S_MAX];
    int nhas = 0;
#pragma unroll
    for (int s = 0; s < LK_S_MAX; ++s) {
        if (s < S) {
            nhas += has[s] ? 1 : 0;
            const float occ = has[s] ? q[s].w : -100.0f;
            const float alpha = lk_sigmoid(coef * occ);
            const float w = alpha * T;
            T *= (1.0f - alpha + 1e-10f);
            wv[s] = w; zv[s] = z[s];
            wsum += w; dsum += w * z[s];
            c0 += w * q[s].x; c1 += w * q[s].y; c2 += w * q[s].z;
        } else { wv[s] = 0.0f; zv[s] = 0.0f; }
    }
    const float ws = wsum + 1e-10f;
    const float depth = dsum / ws;
    float var = 0.0f;
#pragma unroll
    for (int s = 0; s < LK_S_MAX; ++s) { const float t = zv[s] - depth; var += wv[s] * t * t; }
    LkRayOut o;
    o.depth = (gt_depth > 0.0f) ? depth : 0.0f;              // Renderer.py:197-198
    o.var = var;
    o.c0 = c0 / ws; o.c1 = c1 / ws; o.c2 = c2 / ws;
    o.valid = nhas >= S / 2 + 1;
    return o;
}
__device__ __forceinline__ LkRayOut lk_composite_ray(const float* __restrict__ raw, const float* __restrict__ zbuf,
                                                     const int32_t* __restrict__ nbr_count, int r, int S, int min_nn, float coef,
                                                     float gt_depth) {
    float4 q[LK_S_MAX];
    bool has[LK_S_MAX];
    float z[LK_S_MAX];
#pragma unroll
    for (int s = 0; s < LK_S_MAX; ++s) {
        q[s] = make_float4(0.f, 0.f, 0.f, 0.f); has[s] = false; z[s] = 0.0f;
        if (s < S) {
            const int p = r * S + s;
            q[s] = *reinterpret_cast<const float4*>(raw + (size_t)p * 4);
            has[s] = nbr_count[p] >= min_nn;
            z[s] = zbuf[p];
        }
    }
    return lk_composite_vals(q, has, z, S, coef, gt_depth);
}

// d(depth, var, colour) -> d raw[S][4] of one ray (the forward is recomputed from raw), left in out[s]
__device__ __forceinline__ void lk_composite_bwd_ray_core(const float* __restrict__ raw_, const float* __restrict__ zbuf,
                                                          const int32_t* __restrict__ nbr_count, int r, int S, int min_nn, float coef,
                                                          float gt_depth, float d_depth, float gvar, float g0, float g1, float g2,
                                                          float4 (&out)[LK_S_MAX]) {
    float al[LK_S_MAX], be[LK_S_MAX], Tt[LK_S_MAX], wv[LK_S_MAX], zv[LK_S_MAX], cr[LK_S_MAX], cg[LK_S_MAX], cb[LK_S_MAX];
    float T = 1.0f, wsum = 0.0f, dsum = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int s = 0; s < LK_S_MAX; ++s) {
        al[s] = be[s] = Tt[s] = wv[s] = zv[s] = cr[s] = cg[s] = cb[s] = 0.0f;
        if (s < S) {
            const int p = r * S + s;
            const float4 raw = *reinterpret_cast<const float4*>(raw_ + (size_t)p * 4);
            const bool has = nbr_count[p] >= min_nn;
            const float occ = has ? raw.w : -100.0f;
            const float alpha = lk_sigmoid(coef * occ);
            al[s] = alpha; Tt[s] = T; be[s] = 1.0f - alpha + 1e-10f;
            const float w = alpha * T;
            T *= be[s];
            wv[s] = w; zv[s] = zbuf[p];
            cr[s] = raw.x; cg[s] = raw.y; cb[s] = raw.z;
            wsum += w; dsum += w * zv[s];
            c0 += w * raw.x; c1 += w * raw.y; c2 += w * raw.z;
        }
    }
    const float W = wsum + 1e-10f;
    const float depth = dsum / W;
    const float col0 = c0 / W, col1 = c1 / W, col2 = c2 / W;
    float gdep = (gt_depth > 0.0f) ? d_depth : 0.0f;                    // depth of zero-depth rays is overwritten
    float dvar_ddepth = 0.0f;
#pragma unroll
    for (int s = 0; s < LK_S_MAX; ++s) dvar_ddepth += -2.0f * wv[s] * (zv[s] - depth);
    gdep += gvar * dvar_ddepth;
    float gw[LK_S_MAX];
#pragma unroll
    for (int s = 0; s < LK_S_MAX; ++s) {
        const float dz = zv[s] - depth;
        gw[s] = (gdep * dz + g0 * (cr[s] - col0) + g1 * (cg[s] - col1) + g2 * (cb[s] - col2)) / W + gvar * dz * dz;
    }
    float suffix = 0.0f;                                                // sum_{u>s} gw_u w_u
#pragma unroll
    for (int s = LK_S_MAX - 1; s >= 0; --s) {
        if (s < S) {
            const float galpha = gw[s] * Tt[s] - suffix / be[s];
            const float gocc = galpha * al[s] * (1.0f - al[s]) * coef;
            suffix += gw[s] * wv[s];
            const float k = wv[s] / W;
            out[s] = make_float4(g0 * k, g1 * k, g2 * k, gocc);
        } else out[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void lk_composite_bwd_ray(const float* __restrict__ raw_, const float* __restrict__ zbuf,
                                                     const int32_t* __restrict__ nbr_count, int r, int S, int min_nn, float coef,
                                                     float gt_depth, float d_depth, float gvar, float g0, float g1, float g2,
                                                     float* __restrict__ d_raw) {
    float4 out[LK_S_MAX];
    lk_composite_bwd_ray_core(raw_, zbuf, nbr_count, r, S, min_nn, coef, gt_depth, d_depth, gvar, g0, g1, g2, out);
#pragma unroll
    for (int s = 0; s < LK_S_MAX; ++s)
        if (s < S) *reinterpret_cast<float4*>(d_raw + (size_t)(r * S + s) * 4) = out[s];
}
// the same for ONE sample of the ray (the caller is that sample's lane: the tracking loop's decoder backward)
__device__ __forceinline__ float4 lk_composite_bwd_sample(const float* __restrict__ raw_, const float* __restrict__ zbuf,
                                                          const int32_t* __restrict__ nbr_count, int r, int S, int s_own, int min_nn, float coef,
                                                          float gt_depth, float d_depth, float gvar, float g0, float g1, float g2) {
    float4 out[LK_S_MAX];
    lk_composite_bwd_ray_core(raw_, zbuf, nbr_count, r, S, min_nn, coef, gt_depth, d_depth, gvar, g0, g1, g2, out);
    float4 o = out[0];
#pragma unroll
    for (int s = 1; s < LK_S_MAX; ++s)
        if (s == s_own) o = out[s];
    return o;
}
