// The exposure step of the per-frame loops as a device function (lk_optim.hip: k_exposure_step; lk_loop.hip: second workgroup of k_track_final).
#pragma once
#include "lk_common.h"
#include "lk_adam_dev.h"
#include "lk_kernels.h"

// One launch per iteration of the per-frame loops (lk_exposure_desc, include/loopy_hip.h): backward of the exposure MLP from g_aff
// (the body of k_exposure_bwd, gradients kept in g), Adam on the MLP's tensors and on the trainable features, forward with the stepped
// values (k_exposure_fwd) for the next iteration, g_aff cleared.  mode bit 0: backward + step, bit 1: forward (+ clear).
// (struct ExposureStepArgs: lk_kernels.h - it also travels inside LkFeatScatterArgs)
// Called by EVERY thread of a workgroup of >= 256 threads (barriers inside); the first 256 do the work.
// part / n_part (or NULL): per-tile sums [n_part][12] of the tracked frame's d affine as k_decode_bwd leaves them (LkDecodeBwdArgs::g_affine_part,
// F = 1): summed here, on top of g_aff, instead of by a launch of their own.
__device__ __forceinline__ void lk_exposure_step_body(const ExposureStepArgs& a, const float* __restrict__ part, int n_part) {
    __shared__ float s_ga[LK_EXPOSURE_MAX_F * 12];
    __shared__ float s_dp[LK_EXPOSURE_MAX_F * 128];
    __shared__ float s_pw[16][12];
    const int F = a.F;
    const int t = (int)threadIdx.x < 256 ? (int)threadIdx.x : (1 << 24);          // threads past the first 256: every loop below is empty for them
    if (a.mode & 1) {
        for (int e = t; e < F * 12; e += 256) s_ga[e] = a.g_aff[e];
        if (part) {
            // 12 column sums over the tiles: thread -> tiles t, t + blockDim, ...; waves -> LDS -> twelve threads
            float v[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = 0.0f;
            for (int p = (int)threadIdx.x; p < n_part; p += (int)blockDim.x) {
                const float4* q = reinterpret_cast<const float4*>(part + (size_t)p * 12);
                const float4 q0 = q[0], q1 = q[1], q2 = q[2];
                v[0] += q0.x; v[1] += q0.y; v[2] += q0.z; v[3] += q0.w; v[4] += q1.x; v[5] += q1.y; v[6] += q1.z; v[7] += q1.w;
                v[8] += q2.x; v[9] += q2.y; v[10] += q2.z; v[11] += q2.w;
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
            }
            const int wv = (int)threadIdx.x >> 6;
            if (((int)threadIdx.x & 63) == 0 && wv < 16) {
#pragma unroll
                for (int k = 0; k < 12; ++k) s_pw[wv][k] = v[k];
            }
            __syncthreads();
            if (t < 12) {
                float s = s_ga[t];
                const int nw = ((int)blockDim.x + 63) >> 6;
                for (int w = 0; w < nw && w < 16; ++w) s += s_pw[w][t];
                s_ga[t] = s;
            }
        }
        __syncthreads();
        for (int e = t; e < F * 128; e += 256) {
            const int f = e >> 7, u = e & 127;
            float dh = 0.0f;
#pragma unroll
            for (int o = 0; o < 12; ++o) dh = fmaf(a.W2[o * 128 + u], s_ga[f * 12 + o], dh);
            s_dp[e] = dh * lk_softplus100_grad_from_out(a.hid[e]);
        }
        __syncthreads();
        // every gradient element is formed and consumed by the same thread; the feature gradients read W1 BEFORE it is stepped
        float gfeat[(LK_EXPOSURE_MAX_F * 8 + 255) / 256];
#pragma unroll
        for (int q = 0; q < (LK_EXPOSURE_MAX_F * 8 + 255) / 256; ++q) {
            const int e = t + 256 * q;
            float acc = 0.0f;
            if (e < F * 8) {
                const int f = e >> 3, k = e & 7;
                for (int u = 0; u < 128; ++u) acc = fmaf(a.W1[u * 8 + k], s_dp[f * 128 + u], acc);
                a.g[2700 + e] = acc;
            }
            gfeat[q] = acc;
        }
        __syncthreads();
        const bool mlp = a.step_mlp >= 0.0f;
        auto step = [&](float* p, int gi, float gval, float step_size) {
            float m = a.m[gi], v = a.v[gi];
            *p = lk_adam_elem(*p, gval, m, v, a.beta1, a.beta2, a.eps, step_size, a.bc2_sqrt);
            a.m[gi] = m; a.v[gi] = v;
        };
        for (int e = t; e < 1024; e += 256) {                       // W1 [128][8]
            const int u = e >> 3, k = e & 7;
            float acc = 0.0f;
            for (int f = 0; f < F; ++f) acc = fmaf(s_dp[f * 128 + u], a.feats[f * 8 + k], acc);
            a.g[e] = acc;
            if (mlp) step(a.W1 + e, e, acc, a.step_mlp);
        }
        for (int u = t; u < 128; u += 256) {                        // b1
            float acc = 0.0f;
            for (int f = 0; f < F; ++f) acc += s_dp[f * 128 + u];
            a.g[1024 + u] = acc;
            if (mlp) step(a.b1 + u, 1024 + u, acc, a.step_mlp);
        }
        for (int e = t; e < 1536; e += 256) {                       // W2 [12][128]
            const int o = e >> 7, u = e & 127;
            float acc = 0.0f;
            for (int f = 0; f < F; ++f) acc = fmaf(s_ga[f * 12 + o], a.hid[f * 128 + u], acc);
            a.g[1152 + e] = acc;
            if (mlp) step(a.W2 + e, 1152 + e, acc, a.step_mlp);
        }
        for (int o = t; o < 12; o += 256) {                         // b2
            float acc = 0.0f;
            for (int f = 0; f < F; ++f) acc += s_ga[f * 12 + o];
            a.g[2688 + o] = acc;
            if (mlp) step(a.b2 + o, 2688 + o, acc, a.step_mlp);
        }
        __syncthreads();                                            // W1's gradient read the features: step them last
#pragma unroll
        for (int q = 0; q < (LK_EXPOSURE_MAX_F * 8 + 255) / 256; ++q) {
            const int e = t + 256 * q;
            if (e < F * 8 && (e >> 3) >= a.feat_first && (e >> 3) < a.feat_first + a.feat_count) step(a.feats + e, 2700 + e, gfeat[q], a.step_feat);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (a.mode & 2) {
        __shared__ unsigned s_amax;
        if (t == 0) s_amax = 0u;
        float* s_h = s_dp;
        for (int e = t; e < F * 128; e += 256) {
            const int f = e >> 7, u = e & 127;
            float acc = a.b1[u];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc = fmaf(a.W1[u * 8 + k], a.feats[f * 8 + k], acc);
            const float h = lk_softplus100(acc);
            s_h[e] = h;
            a.hid[e] = h;
        }
        __syncthreads();
        for (int e = t; e < F * 12; e += 256) {
            const int f = e / 12, o = e - f * 12;
            float acc = a.b2[o];
            for (int u = 0; u < 128; ++u) acc = fmaf(a.W2[o * 128 + u], s_h[f * 128 + u], acc);
            a.aff[e] = acc;
            a.g_aff[e] = 0.0f;
            if (o < 9 && a.bwd_scale) atomicMax(&s_amax, __float_as_uint(fabsf(acc)));       // the 3 x 3 part (positive floats order like their bits)
        }
        if (a.bwd_scale) {
            __syncthreads();
            if (t == 0) {
                // |d out| <= (0.25 w_color) x (row / column sum of |A|) <= (0.25 w_color) x 3 max|A|: the power of two that brings 3 max|A| into (0.5, 1]
                const float bound = 3.0f * __uint_as_float(s_amax);
                int ex = 0;
                if (bound > 0.0f && bound < 3.0e38f) { (void)frexpf(bound, &ex); }       // bound = m 2^ex, m in [0.5, 1)
                ex = ex > 40 ? 40 : (ex < -40 ? -40 : ex);
                *a.bwd_scale = ldexpf(1.0f, -ex);
            }
        }
    }
}
