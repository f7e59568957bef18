// Per-ray alpha composite and its backward as device functions (raw2outputs_nerf_color, src/common.py:382-422;
// Renderer.py:184-200), shared by the many-workgroup kernels (k_composite, k_composite_bwd) and the one-workgroup
// fused forms of the tracking loop (lk_loop.hip).
#pragma once
#include "lk_common.h"
#include "lk_kernels.h"

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

// The composite of one ray as its backward needs it (recomputed from raw): per-sample alpha / transmittance / weight, the normalised outputs
struct LkRayState {
    float al[LK_S_MAX], be[LK_S_MAX], Tt[LK_S_MAX], wv[LK_S_MAX], zv[LK_S_MAX], cr[LK_S_MAX], cg[LK_S_MAX], cb[LK_S_MAX];
    float W, depth, col0, col1, col2;
    int nhas;
};
__device__ __forceinline__ void lk_ray_state(const float* __restrict__ raw_, const float* __restrict__ zbuf, const int32_t* __restrict__ nbr_count,
                                             int r, int S, int min_nn, float coef, LkRayState& st) {
    float T = 1.0f, wsum = 0.0f, dsum = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    st.nhas = 0;
#pragma unroll
    for (int s = 0; s < LK_S_MAX; ++s) {
        st.al[s] = st.be[s] = st.Tt[s] = st.wv[s] = st.zv[s] = st.cr[s] = st.cg[s] = st.cb[s] = 0.0f;
        if (s < S) {
            const int p = r * S + s;
            const float4 raw = *reinterpret_cast<const float4*>(raw_ + (size_t)p * 4);
            const bool has = nbr_count[p] >= min_nn;
            st.nhas += has ? 1 : 0;
            const float occ = has ? raw.w : -100.0f;
            const float alpha = lk_sigmoid(coef * occ);
            st.al[s] = alpha; st.Tt[s] = T; st.be[s] = 1.0f - alpha + 1e-10f;
            const float w = alpha * T;
            T *= st.be[s];
            st.wv[s] = w; st.zv[s] = zbuf[p];
            st.cr[s] = raw.x; st.cg[s] = raw.y; st.cb[s] = raw.z;
            wsum += w; dsum += w * st.zv[s];
            c0 += w * raw.x; c1 += w * raw.y; c2 += w * raw.z;
        }
    }
    st.W = wsum + 1e-10f;
    st.depth = dsum / st.W;
    st.col0 = c0 / st.W; st.col1 = c1 / st.W; st.col2 = c2 / st.W;
}
__device__ __forceinline__ float lk_ray_var(const LkRayState& st) {
    float var = 0.0f;
#pragma unroll
    for (int s = 0; s < LK_S_MAX; ++s) { const float t = st.zv[s] - st.depth; var += st.wv[s] * t * t; }
    return var;
}
// d(depth, var, colour) -> d raw[S][4] of the ray, left in out[s]
__device__ __forceinline__ void lk_ray_grad(const LkRayState& st, int S, float coef, float gt_depth, float d_depth, float gvar, float g0, float g1, float g2,
                                            float4 (&out)[LK_S_MAX]) {
    const float W = st.W, depth = st.depth;
    float gdep = (gt_depth > 0.0f) ? d_depth : 0.0f;                    // depth of zero-depth rays is overwritten
    float dvar_ddepth = 0.0f;
#pragma unroll
    for (int s = 0; s < LK_S_MAX; ++s) dvar_ddepth += -2.0f * st.wv[s] * (st.zv[s] - depth);
    gdep += gvar * dvar_ddepth;
    float gw[LK_S_MAX];
#pragma unroll
    for (int s = 0; s < LK_S_MAX; ++s) {
        const float dz = st.zv[s] - depth;
        gw[s] = (gdep * dz + g0 * (st.cr[s] - st.col0) + g1 * (st.cg[s] - st.col1) + g2 * (st.cb[s] - st.col2)) / W + gvar * dz * dz;
    }
    float suffix = 0.0f;                                                // sum_{u>s} gw_u w_u
#pragma unroll
    for (int s = LK_S_MAX - 1; s >= 0; --s) {
        if (s < S) {
            const float galpha = gw[s] * st.Tt[s] - suffix / st.be[s];
            const float gocc = galpha * st.al[s] * (1.0f - st.al[s]) * coef;
            suffix += gw[s] * st.wv[s];
            const float k = st.wv[s] / W;
            out[s] = make_float4(g0 * k, g1 * k, g2 * k, gocc);
        } else out[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// d(depth, var, colour) -> d raw[S][4] of one ray (the forward is recomputed from raw), left in out[s]
__device__ __forceinline__ void lk_composite_bwd_ray_core(const float* __restrict__ raw_, const float* __restrict__ zbuf,
                                                          const int32_t* __restrict__ nbr_count, int r, int S, int min_nn, float coef,
                                                          float gt_depth, float d_depth, float gvar, float g0, float g1, float g2,
                                                          float4 (&out)[LK_S_MAX]) {
    LkRayState st;
    lk_ray_state(raw_, zbuf, nbr_count, r, S, min_nn, coef, st);
    lk_ray_grad(st, S, coef, gt_depth, d_depth, gvar, g0, g1, g2, out);
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

// The mapper's loss (Mapper.py:691-720) of the ray of sample sp and the composite backward of it for that sample - k_composite's arithmetic (forward,
// mask, terms, gradients) with the ray recomputed by the sample's own lane.  write_ray: this lane also stores the ray's outputs (one lane per ray does);
// *geo / *col / *cnt: the ray's terms of the loss row.
__device__ __forceinline__ float4 lk_map_draw(const LkCompositeArgs& a, int sp, bool write_ray, float* geo, float* col, float* cnt) {
    const int S = a.S;
    const int r = sp / S, s_own = sp - r * S;
    const float gd = a.gt_depth[r];
    LkRayState st;
    lk_ray_state(a.raw, a.z, a.nbr_count, r, S, a.min_nn, a.coef, st);
    const float dout = ((a.keep_depth ? 1.0f : gd) > 0.0f) ? st.depth : 0.0f;             // Renderer.py:197-198
    const bool valid = st.nhas >= S / 2 + 1;
    const bool m = (gd > 0.0f) && valid && !(dout != dout);
    float dd = 0.0f, d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, l_geo = 0.0f, l_col = 0.0f, l_cnt = 0.0f;
    if (m) {
        l_geo = fabsf(gd - dout);
        dd = (dout > gd) ? 1.0f : ((dout < gd) ? -1.0f : 0.0f);
        l_cnt = 1.0f;
        if (a.use_color) {
            const float e0 = st.col0 - a.gt_color[3 * r], e1 = st.col1 - a.gt_color[3 * r + 1], e2 = st.col2 - a.gt_color[3 * r + 2];
            l_col = fabsf(e0) + fabsf(e1) + fabsf(e2);
            d0 = a.w_color * ((e0 > 0.0f) ? 1.0f : ((e0 < 0.0f) ? -1.0f : 0.0f));
            d1 = a.w_color * ((e1 > 0.0f) ? 1.0f : ((e1 < 0.0f) ? -1.0f : 0.0f));
            d2 = a.w_color * ((e2 > 0.0f) ? 1.0f : ((e2 < 0.0f) ? -1.0f : 0.0f));
        }
    }
    if (write_ray) {
        a.depth[r] = dout; a.var[r] = lk_ray_var(st);
        a.color[3 * r] = st.col0; a.color[3 * r + 1] = st.col1; a.color[3 * r + 2] = st.col2;
        a.valid_ray[r] = valid ? 1 : 0;
        a.d_depth[r] = dd;
        a.d_color[3 * r] = d0; a.d_color[3 * r + 1] = d1; a.d_color[3 * r + 2] = d2;
    }
    *geo = l_geo; *col = l_col; *cnt = l_cnt;
    float4 out[LK_S_MAX];
    lk_ray_grad(st, S, a.coef, gd, dd, 0.0f, d0, d1, d2, out);
    float4 o = out[0];
#pragma unroll
    for (int s = 1; s < LK_S_MAX; ++s)
        if (s == s_own) o = out[s];
    return o;
}

// The composite backward of sample sp from the per-ray loss gradients a caller left in d_depth / d_var / d_color (k_composite_bwd's arithmetic
// for one sample; the sample's own lane calls)
__device__ __forceinline__ float4 lk_cb_draw(const LkCompositeBwdArgs& a, int sp) {
    const int r = sp / a.S;
    return lk_composite_bwd_sample(a.raw, a.z, a.nbr_count, r, a.S, sp - r * a.S, a.min_nn, a.coef, a.keep_depth ? 1.0f : a.gt_depth[r], a.d_depth[r],
                                   a.d_var ? a.d_var[r] : 0.0f, a.d_color ? a.d_color[3 * r] : 0.0f, a.d_color ? a.d_color[3 * r + 1] : 0.0f,
                                   a.d_color ? a.d_color[3 * r + 2] : 0.0f);
}
