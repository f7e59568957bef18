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
__device__ __forceinline__ void lk_adam_one(const AdamSegDev& S, long long i, long long e, float b1, float b2, float eps) {
    const float g = S.g[S.g_compact ? i : e];
    float m = S.m[i], v = S.v[i];
    if (S.p_f16) {                                      // half table: fp32 step, stored rounded to nearest
        _Float16* ph = reinterpret_cast<_Float16*>(S.p) + e;
        *ph = (_Float16)lk_adam_elem((float)*ph, g, m, v, b1, b2, eps, S.step_size, S.bc2_sqrt);
    } else {
        S.p[e] = lk_adam_elem(S.p[e], g, m, v, b1, b2, eps, S.step_size, S.bc2_sqrt);
    }
    S.m[i] = m;
    S.v[i] = v;
    if (S.zero_grad && !S.g_compact) S.g[e] = 0.0f;
}
// Flagged rows (lk_adam_seg::row_flags): a wave reads the flags of 64 consecutive rows with one coalesced load, then walks the set bits
// of the ballot - rows of up to 64 elements, one per half-wave when row_len <= 32 (the feature tables: 32).  A 5 M-row table of which a
// refinement call has touched 5 % costs 5 MB of flags + the touched rows instead of 28 bytes for every element.
__device__ __forceinline__ void lk_adam_seg_flagged(const AdamSegDev& S, float b1, float b2, float eps, int bx, int gx) {
    const int lane = (int)threadIdx.x & 63;
    const long long n_rows = S.n / S.row_len;
    const long long wave = (long long)bx * 4 + ((int)threadIdx.x >> 6), n_waves = (long long)gx * 4;
    const bool pair = S.row_len <= 32;
    // rows per wave and pass: 64 (one coalesced flag load) where the table has at least that many rows per wave of the launch; fewer on smaller
    // tables - a wave walks its flagged rows two at a time, each step a dependent load -> step -> store round trip, and with DENSE flags (the
    // end-of-sequence refinement of a 20 000-point map: every row touched) 64 rows per wave were 32 such round trips on 312 of the launch's
    // 8 192 waves: 29 us for what the row-list form does in 6
    int RW = 64;
    while (RW > 2 && n_rows < n_waves * RW) RW >>= 1;
    for (long long r0 = wave * RW; r0 < n_rows; r0 += n_waves * RW) {
        const bool on = lane < RW && r0 + lane < n_rows && S.row_flags[r0 + lane] != 0;
        unsigned long long mask = __ballot(on);
        while (mask) {
            const int ra = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            int rb = -1;
            if (pair && mask) { rb = __ffsll((long long)mask) - 1; mask &= mask - 1; }
            const int half = lane >> 5;
            const int row = pair ? (half ? rb : ra) : ra;
            const int k = pair ? (lane & 31) : lane;
            if (row >= 0 && k < S.row_len) {
                const long long e = (r0 + row) * S.row_len + k;
                lk_adam_one(S, e, e, b1, b2, eps);
            }
        }
    }
}
__device__ __forceinline__ void lk_adam_seg_block(const AdamSegDev& S, float b1, float b2, float eps, int bx, int gx) {
    if (S.row_flags) { lk_adam_seg_flagged(S, b1, b2, eps, bx, gx); return; }
    for (long long i = (long long)bx * 256 + threadIdx.x; i < S.n; i += (long long)gx * 256) {
        long long e = i;
        if (S.row_index) {
            const long long row = i / S.row_len;
            e = (long long)S.row_index[row] * S.row_len + (i - row * S.row_len);
        }
        lk_adam_one(S, i, e, b1, b2, eps);
    }
}
