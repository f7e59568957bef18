// torch.optim.Adam (amsgrad=False, weight_decay=0) element arithmetic, shared by k_adam (lk_optim.hip) and the optimiser step that
// rides in k_bwd_reduce (lk_bwd2.hip, LkStepRider).
#pragma once
#include "lk_common.h"
#include "lk_kernels.h"

// one element: returns the new parameter value, updates the moments in place
__device__ __forceinline__ float lk_adam_elem(float p, float g, float& m, float& v, float b1, float b2, float eps, float step_size, float bc2_sqrt) {
    m = m * b1 + (1.0f - b1) * g;                       // exp_avg.mul_(beta1).add_(grad, alpha=1-beta1)
    v = v * b2 + (1.0f - b2) * (g * g);                 // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps;      // (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps)
    return p - step_size * (m / denom);                 // param.addcdiv_(exp_avg, denom, value=-lr/bias_correction1)
}

// Element i of a segment is p[i] - or, with a row index (frustum-selected feature rows optimised in place in
// the full table, Mapper.py:498-512,578-586), p[row_index[i / row_len] * row_len + i % row_len]; m and v are
// always compact.  zero_grad clears the consumed gradient so the next iteration's scatter-add starts from 0.
__device__ __forceinline__ void lk_adam_seg_block(const AdamSegDev& S, float b1, float b2, float eps, int bx, int gx) {
    for (long long i = (long long)bx * 256 + threadIdx.x; i < S.n; i += (long long)gx * 256) {
        long long e = i;
        if (S.row_index) {
            const long long row = i / S.row_len;
            e = (long long)S.row_index[row] * S.row_len + (i - row * S.row_len);
        }
        const float g = S.g[e];
        float m = S.m[i], v = S.v[i];
        if (S.p_f16) {                                      // half table: fp32 step, stored rounded to nearest
            _Float16* ph = reinterpret_cast<_Float16*>(S.p) + e;
            *ph = (_Float16)lk_adam_elem((float)*ph, g, m, v, b1, b2, eps, S.step_size, S.bc2_sqrt);
        } else {
            S.p[e] = lk_adam_elem(S.p[e], g, m, v, b1, b2, eps, S.step_size, S.bc2_sqrt);
        }
        S.m[i] = m;
        S.v[i] = v;
        if (S.zero_grad) S.g[e] = 0.0f;
    }
}
