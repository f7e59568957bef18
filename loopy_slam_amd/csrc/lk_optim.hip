// Everything around the render call inside one optimisation iteration:
//   losses fused with their output gradients (Mapper.py:691-720, Tracker.py:169-191),
//   multi-tensor Adam (torch.optim.Adam semantics; Mapper.py:570,723, Tracker.py:352,194),
//   pose -> rays and its backward (common.py:301-343, 104-120),
//   inside-mask threshold (median / max) and stable ballot + prefix-sum compaction
//   (Tracker.py:153-160, Mapper.py:674-681, common.py:249-255).
#include "lk_common.h"
#include "lk_mask_dev.h"
#include "lk_adam_dev.h"
#include "lk_exposure_dev.h"
#include "lk_composite_dev.h"

#include <math.h>
#include <string.h>

// ------------------------------------------------------------------ block reduction helper
__device__ __forceinline__ float block_sum_256(float v, float* sh /*[4]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lk_lane() == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ float sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

// ------------------------------------------------------------------ mapper loss
__global__ __launch_bounds__(256) void k_loss_mapper(int R, const float* __restrict__ depth, const float* __restrict__ color,
                                                     const uint8_t* __restrict__ valid, const float* __restrict__ gt_depth,
                                                     const float* __restrict__ gt_color, float w_color, int use_color,
                                                     float* __restrict__ d_depth, float* __restrict__ d_color,
                                                     float* __restrict__ out) {
    __shared__ float sh[4];
    const int r = blockIdx.x * 256 + (int)threadIdx.x;
    float geo = 0.0f, col = 0.0f, cnt = 0.0f;
    if (r < R) {
        const float d = depth[r], g = gt_depth[r];
        const bool m = (g > 0.0f) && valid[r] && !(d != d);
        float dd = 0.0f, dc0 = 0.0f, dc1 = 0.0f, dc2 = 0.0f;
        if (m) {
            geo = fabsf(g - d);
            dd = sgn(d - g);
            cnt = 1.0f;
            if (use_color) {
                const float e0 = color[3 * r] - gt_color[3 * r], e1 = color[3 * r + 1] - gt_color[3 * r + 1],
                            e2 = color[3 * r + 2] - gt_color[3 * r + 2];
                col = fabsf(e0) + fabsf(e1) + fabsf(e2);
                dc0 = w_color * sgn(e0); dc1 = w_color * sgn(e1); dc2 = w_color * sgn(e2);
            }
        }
        d_depth[r] = dd;
        d_color[3 * r] = dc0; d_color[3 * r + 1] = dc1; d_color[3 * r + 2] = dc2;
    }
    geo = block_sum_256(geo, sh);
    col = block_sum_256(col, sh);
    cnt = block_sum_256(cnt, sh);
    if (threadIdx.x == 0) {
        atomicAdd(out + 0, geo + (use_color ? w_color * col : 0.0f));
        atomicAdd(out + 1, geo);
        atomicAdd(out + 2, col);
        atomicAdd(out + 3, cnt);
    }
}

// ------------------------------------------------------------------ tracker loss (two passes: mean of the normalised residual)
// MEDIAN (tracking.handle_dynamic: False, Tracker.py:177-179): scratch[r] = |gt - depth| (sign bit set for an absent ray), no sums;
// k_loss_tracker_median then leaves 10 x the median in scratch[R] and pass 2 masks by it
template <bool MEDIAN>
__global__ __launch_bounds__(256) void k_loss_tracker_pass1(int R, const float* __restrict__ depth, const float* __restrict__ var,
                                                            const float* __restrict__ gt_depth, float* __restrict__ scratch) {
    __shared__ float sh[4];
    const int r = blockIdx.x * 256 + (int)threadIdx.x;
    if (MEDIAN) {
        if (r < R) scratch[r] = (gt_depth[r] > 0.0f) ? fabsf(gt_depth[r] - depth[r]) : -1.0f;
        return;
    }
    // Rays with gt_depth <= 0 are "absent": the reference filters them out BEFORE rendering
    // (get_samples depth_filter + inside mask, Tracker.py:142-160), so they take no part in the mean.
    float t = 0.0f, c = 0.0f;
    if (r < R) {
        const bool present = gt_depth[r] > 0.0f;
        t = present ? fabsf(gt_depth[r] - depth[r]) / sqrtf(var[r] + 1e-10f) : 0.0f;
        c = present ? 1.0f : 0.0f;
        scratch[r] = t;
    }
    t = block_sum_256(t, sh);
    c = block_sum_256(c, sh);
    if (threadIdx.x == 0) { atomicAdd(scratch + R, t); atomicAdd(scratch + R + 1, c); }
}

__global__ __launch_bounds__(1024) void k_loss_tracker_median(int R, float* __restrict__ scratch) {
    __shared__ LkMedianShared S;
    const float thr = lk_block_median10(scratch, R, S);
    if (threadIdx.x == 0) scratch[R] = thr;
}

template <bool MEDIAN>
__global__ __launch_bounds__(256) void k_loss_tracker_pass2(int R, const float* __restrict__ depth, const float* __restrict__ var,
                                                            const float* __restrict__ color, const float* __restrict__ gt_depth,
                                                            const float* __restrict__ gt_color, float w_color, int use_color,
                                                            const float* __restrict__ scratch, float* __restrict__ d_depth,
                                                            float* __restrict__ d_color, float* __restrict__ out) {
    __shared__ float sh[4];
    const int r = blockIdx.x * 256 + (int)threadIdx.x;
    const float thr = MEDIAN ? scratch[R] : 10.0f * (scratch[R] / fmaxf(scratch[R + 1], 1.0f));
    float geo = 0.0f, col = 0.0f, cnt = 0.0f;
    if (r < R) {
        const float d = depth[r], v = var[r], g = gt_depth[r];
        const float t = MEDIAN ? fabsf(g - d) / sqrtf(v + 1e-10f) : scratch[r];     // the loss term stays uncertainty-normalised
        const float tm = MEDIAN ? scratch[r] : t;                                   // what the mask compares
        const bool m = (tm < thr) && (g > 0.0f) && !(d != d) && !(v != v);
        float dd = 0.0f, dc0 = 0.0f, dc1 = 0.0f, dc2 = 0.0f;
        if (m) {
            geo = fminf(fmaxf(t, 0.0f), 1e3f);
            if (t <= 1e3f) dd = sgn(d - g) / sqrtf(v + 1e-10f);
            cnt = 1.0f;
            const float e0 = color[3 * r] - gt_color[3 * r], e1 = color[3 * r + 1] - gt_color[3 * r + 1],
                        e2 = color[3 * r + 2] - gt_color[3 * r + 2];
            col = fabsf(e0) + fabsf(e1) + fabsf(e2);
            if (use_color) { dc0 = w_color * sgn(e0); dc1 = w_color * sgn(e1); dc2 = w_color * sgn(e2); }
        }
        d_depth[r] = dd;
        d_color[3 * r] = dc0; d_color[3 * r + 1] = dc1; d_color[3 * r + 2] = dc2;
    }
    geo = block_sum_256(geo, sh);
    col = block_sum_256(col, sh);
    cnt = block_sum_256(cnt, sh);
    if (threadIdx.x == 0) {
        atomicAdd(out + 0, geo + (use_color ? w_color * col : 0.0f));
        atomicAdd(out + 1, geo);
        atomicAdd(out + 2, col);
        atomicAdd(out + 3, cnt);
    }
}

// ------------------------------------------------------------------ one-workgroup forms for training batches
// R <= LK_LOSS_1WG_MAX rays: a single 1024-thread workgroup owns the whole batch, so the sums need neither atomics
// nor a zero-fill launch, the tracker's two passes become one kernel, and the result is order-deterministic.
#define LK_LOSS_1WG_MAX 16384
__device__ __forceinline__ float block_sum_1024(float v, float* sh /*[16]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lk_lane() == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += sh[i];
    return s;
}

__global__ __launch_bounds__(1024) void k_loss_mapper_1wg(int R, const float* __restrict__ depth, const float* __restrict__ color,
                                                          const uint8_t* __restrict__ valid, const float* __restrict__ gt_depth,
                                                          const float* __restrict__ gt_color, float w_color, int use_color,
                                                          float* __restrict__ d_depth, float* __restrict__ d_color,
                                                          float* __restrict__ out) {
    __shared__ float sh[16];
    float geo = 0.0f, col = 0.0f, cnt = 0.0f;
    for (int r = threadIdx.x; r < R; r += 1024) {
        const float d = depth[r], g = gt_depth[r];
        const bool m = (g > 0.0f) && valid[r] && !(d != d);
        float dd = 0.0f, dc0 = 0.0f, dc1 = 0.0f, dc2 = 0.0f;
        if (m) {
            geo += fabsf(g - d);
            dd = sgn(d - g);
            cnt += 1.0f;
            if (use_color) {
                const float e0 = color[3 * r] - gt_color[3 * r], e1 = color[3 * r + 1] - gt_color[3 * r + 1],
                            e2 = color[3 * r + 2] - gt_color[3 * r + 2];
                col += fabsf(e0) + fabsf(e1) + fabsf(e2);
                dc0 = w_color * sgn(e0); dc1 = w_color * sgn(e1); dc2 = w_color * sgn(e2);
            }
        }
        d_depth[r] = dd;
        d_color[3 * r] = dc0; d_color[3 * r + 1] = dc1; d_color[3 * r + 2] = dc2;
    }
    geo = block_sum_1024(geo, sh);
    col = block_sum_1024(col, sh);
    cnt = block_sum_1024(cnt, sh);
    if (threadIdx.x == 0) {
        out[0] = geo + (use_color ? w_color * col : 0.0f);
        out[1] = geo; out[2] = col; out[3] = cnt;
    }
}

// MEDIAN: the mask compares |gt - depth| with 10 x its median (handle_dynamic: False); the residuals go through `scratch` [R] for the select
template <bool MEDIAN>
__global__ __launch_bounds__(1024) void k_loss_tracker_1wg(int R, const float* __restrict__ depth, const float* __restrict__ var,
                                                           const float* __restrict__ color, const float* __restrict__ gt_depth,
                                                           const float* __restrict__ gt_color, float w_color, int use_color,
                                                           float* __restrict__ d_depth, float* __restrict__ d_color,
                                                           float* __restrict__ out, float* __restrict__ scratch) {
    __shared__ float sh[16];
    __shared__ LkMedianShared S;
    constexpr int VPT = LK_LOSS_1WG_MAX / 1024;
    float tv[VPT], tm[VPT];
    float tsum = 0.0f, csum = 0.0f;
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        const int r = (int)threadIdx.x + 1024 * q;
        tv[q] = 0.0f; tm[q] = 0.0f;
        if (r < R) {
            const bool present = gt_depth[r] > 0.0f;     // absent rays take no part in the mean (see pass1 above)
            tv[q] = present ? fabsf(gt_depth[r] - depth[r]) / sqrtf(var[r] + 1e-10f) : 0.0f;
            tsum += tv[q];
            csum += present ? 1.0f : 0.0f;
            if (MEDIAN) { tm[q] = present ? fabsf(gt_depth[r] - depth[r]) : -1.0f; scratch[r] = tm[q]; }
        }
    }
    float thr;
    if (MEDIAN) {
        __syncthreads();                                   // the workgroup's own stores to scratch
        thr = lk_block_median10(scratch, R, S);
    } else {
        tsum = block_sum_1024(tsum, sh);
        csum = block_sum_1024(csum, sh);
        thr = 10.0f * (tsum / fmaxf(csum, 1.0f));
    }
    float geo = 0.0f, col = 0.0f, cnt = 0.0f;
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        const int r = (int)threadIdx.x + 1024 * q;
        if (r < R) {
            const float d = depth[r], v = var[r], g = gt_depth[r], t = tv[q];
            const bool m = ((MEDIAN ? tm[q] : t) < thr) && (g > 0.0f) && !(d != d) && !(v != v);
            float dd = 0.0f, dc0 = 0.0f, dc1 = 0.0f, dc2 = 0.0f;
            if (m) {
                geo += fminf(fmaxf(t, 0.0f), 1e3f);
                if (t <= 1e3f) dd = sgn(d - g) / sqrtf(v + 1e-10f);
                cnt += 1.0f;
                const float e0 = color[3 * r] - gt_color[3 * r], e1 = color[3 * r + 1] - gt_color[3 * r + 1],
                            e2 = color[3 * r + 2] - gt_color[3 * r + 2];
                col += fabsf(e0) + fabsf(e1) + fabsf(e2);
                if (use_color) { dc0 = w_color * sgn(e0); dc1 = w_color * sgn(e1); dc2 = w_color * sgn(e2); }
            }
            d_depth[r] = dd;
            d_color[3 * r] = dc0; d_color[3 * r + 1] = dc1; d_color[3 * r + 2] = dc2;
        }
    }
    geo = block_sum_1024(geo, sh);
    col = block_sum_1024(col, sh);
    cnt = block_sum_1024(cnt, sh);
    if (threadIdx.x == 0) {
        out[0] = geo + (use_color ? w_color * col : 0.0f);
        out[1] = geo; out[2] = col; out[3] = cnt;
    }
}

// ------------------------------------------------------------------ Adam
struct AdamArgs { AdamSegDev s[LK_ADAM_MAX_SEG]; int n_seg; float beta1, beta2, eps; };

// (element arithmetic and the per-segment block body: lk_adam_dev.h - k_bwd_reduce carries the same step as a rider)
__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
    lk_adam_seg_block(a.s[blockIdx.y], a.beta1, a.beta2, a.eps, (int)blockIdx.x, (int)gridDim.x);
}
// the same launch with one more row of blocks: block (0, n_seg) is the exposure step of the mapping iteration (lk_map_frame with
// exposure encoding: it depends on the loss kernel's d affine only, and was a launch of its own behind this one)
__global__ __launch_bounds__(256) void k_adam_x(AdamArgs a, ExposureStepArgs xa) {
    if ((int)blockIdx.y == a.n_seg) {
        if (blockIdx.x == 0) lk_exposure_step_body(xa, nullptr, 0);
        return;
    }
    lk_adam_seg_block(a.s[blockIdx.y], a.beta1, a.beta2, a.eps, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------ pose -> rays
__device__ __forceinline__ void quat_rot(const float* __restrict__ cam, float (&Rm)[9]) {
    const float qr = cam[0], qi = cam[1], qj = cam[2], qk = cam[3];
    const float s = 2.0f / (qr * qr + qi * qi + qj * qj + qk * qk);
    Rm[0] = 1.0f - s * (qj * qj + qk * qk); Rm[1] = s * (qi * qj - qk * qr); Rm[2] = s * (qi * qk + qj * qr);
    Rm[3] = s * (qi * qj + qk * qr); Rm[4] = 1.0f - s * (qi * qi + qk * qk); Rm[5] = s * (qj * qk - qi * qr);
    Rm[6] = s * (qi * qk - qj * qr); Rm[7] = s * (qj * qk + qi * qr); Rm[8] = 1.0f - s * (qi * qi + qj * qj);
}

__global__ __launch_bounds__(256) void k_rays_from_pose(const float* __restrict__ cam, const float* __restrict__ pi,
                                                        const float* __restrict__ pj, int R, float fx, float fy, float cx,
                                                        float cy, float* __restrict__ ro, float* __restrict__ rd) {
    const int r = blockIdx.x * 256 + (int)threadIdx.x;
    if (r >= R) return;
    float Rm[9];
    quat_rot(cam, Rm);
    const float d0 = (pi[r] - cx) / fx, d1 = -(pj[r] - cy) / fy, d2 = -1.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        rd[3 * r + a] = (d0 * Rm[3 * a] + d1 * Rm[3 * a + 1]) + d2 * Rm[3 * a + 2];
        ro[3 * r + a] = cam[4 + a];
    }
}

// d loss / d cam7 from d rays: single block (R is a ray batch, <= ~1e4 in the tracker)
__global__ __launch_bounds__(256) void k_pose_bwd(const float* __restrict__ cam, const float* __restrict__ pi,
                                                  const float* __restrict__ pj, int R, float fx, float fy, float cx, float cy,
                                                  const float* __restrict__ g_ro, const float* __restrict__ g_rd,
                                                  float* __restrict__ g_cam) {
    __shared__ float sh[4];
    __shared__ float acc[12];
    float G[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gT[3] = {0.f, 0.f, 0.f};
    for (int r = threadIdx.x; r < R; r += 256) {
        const float dir[3] = {(pi[r] - cx) / fx, -(pj[r] - cy) / fy, -1.0f};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float g = g_rd[3 * r + a];
            G[3 * a] = fmaf(g, dir[0], G[3 * a]); G[3 * a + 1] = fmaf(g, dir[1], G[3 * a + 1]); G[3 * a + 2] = fmaf(g, dir[2], G[3 * a + 2]);
            gT[a] += g_ro[3 * r + a];
        }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) { const float s = block_sum_256(G[q], sh); if (threadIdx.x == 0) acc[q] = s; }
#pragma unroll
    for (int q = 0; q < 3; ++q) { const float s = block_sum_256(gT[q], sh); if (threadIdx.x == 0) acc[9 + q] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float qr = cam[0], qi = cam[1], qj = cam[2], qk = cam[3];
        const float N = qr * qr + qi * qi + qj * qj + qk * qk, s = 2.0f / N;
        const float P[9] = {-(qj * qj + qk * qk), qi * qj - qk * qr, qi * qk + qj * qr,
                            qi * qj + qk * qr, -(qi * qi + qk * qk), qj * qk - qi * qr,
                            qi * qk - qj * qr, qj * qk + qi * qr, -(qi * qi + qj * qj)};
        float gp = 0.0f;
#pragma unroll
        for (int q = 0; q < 9; ++q) gp += acc[q] * P[q];
        const float* g = acc;
        const float dPr = g[1] * (-qk) + g[2] * qj + g[3] * qk + g[5] * (-qi) + g[6] * (-qj) + g[7] * qi;
        const float dPi = g[1] * qj + g[2] * qk + g[3] * qj + g[4] * (-2.0f * qi) + g[5] * (-qr) + g[6] * qk + g[7] * qr + g[8] * (-2.0f * qi);
        const float dPj = g[0] * (-2.0f * qj) + g[1] * qi + g[2] * qr + g[3] * qi + g[5] * qk + g[6] * (-qr) + g[7] * qk + g[8] * (-2.0f * qj);
        const float dPk = g[0] * (-2.0f * qk) + g[1] * (-qr) + g[2] * qi + g[3] * qr + g[4] * (-2.0f * qk) + g[5] * qj + g[6] * qi + g[7] * qj;
        const float ds = -s * s;            // d s / d q_x = -s^2 q_x
        g_cam[0] = ds * qr * gp + s * dPr;
        g_cam[1] = ds * qi * gp + s * dPi;
        g_cam[2] = ds * qj * gp + s * dPj;
        g_cam[3] = ds * qk * gp + s * dPk;
        g_cam[4] = acc[9]; g_cam[5] = acc[10]; g_cam[6] = acc[11];
    }
}

// ------------------------------------------------------------------ stable compaction (single block, 1024 threads)
__global__ __launch_bounds__(1024) void k_compact(const uint8_t* __restrict__ mask, int n, int32_t* __restrict__ out_index,
                                                  int32_t* __restrict__ out_count) {
    __shared__ int wcount[16];
    __shared__ int base;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + t;
        const bool keep = (i < n) && mask[i] != 0;
        const unsigned long long b = __ballot(keep);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[w] = __popcll(b);
        __syncthreads();
        int woff = base;
        for (int q = 0; q < w; ++q) woff += wcount[q];
        if (keep) out_index[woff + before] = i;
        __syncthreads();
        if (t == 1023) base = woff + __popcll(b);
        __syncthreads();
    }
    if (t == 0) *out_count = base;
}

// ------------------------------------------------------------------ inside mask: thr = min(10*median(d>0), 1.2*max)
// (single 1024-thread workgroup; the threshold search itself is lk_mask_dev.h::lk_inside_thr)
template <bool REG>
__global__ __launch_bounds__(1024) void k_inside_mask(const float* depth, int n, uint8_t* __restrict__ mask, float* depth_filtered,
                                                      float* __restrict__ out_thr, uint32_t* __restrict__ scratch) {
    __shared__ LkMaskShared S;
    constexpr int VPT = LK_MASK_VPT;
    const int t = threadIdx.x;
    unsigned u[VPT];
    unsigned mycnt = 0, mymax = 0;
    if (REG) {
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int i = t + 1024 * q;
            const float d = (i < n) ? depth[i] : 0.0f;
            u[q] = (d > 0.0f) ? __float_as_uint(d) : 0u;            // positive floats order like their bit patterns
            if (u[q]) { ++mycnt; mymax = max(mymax, u[q]); }
        }
    } else {
#pragma unroll
        for (int q = 0; q < VPT; ++q) u[q] = 0u;
        for (int i = t; i < n; i += 1024) {
            const float d = depth[i];
            const unsigned v = (d > 0.0f) ? __float_as_uint(d) : 0u;
            scratch[i] = v;
            if (v) { ++mycnt; mymax = max(mymax, v); }
        }
    }
    bool any;
    const float thr = lk_inside_thr<REG>(u, scratch, n, mycnt, mymax, S, &any);
    if (!any) {
        for (int i = t; i < n; i += 1024) { if (mask) mask[i] = 0; if (depth_filtered) depth_filtered[i] = 0.0f; }
        if (t == 0) *out_thr = 0.0f;
        return;
    }
    if (REG) {
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int i = t + 1024 * q;
            if (i < n) {
                const float d = __uint_as_float(u[q]);                  // 0 for non-positive depths
                const bool in = u[q] && d <= thr;
                if (mask) mask[i] = in ? 1 : 0;
                if (depth_filtered) depth_filtered[i] = in ? d : 0.0f;  // rejected rays become "absent" (gt_depth = 0)
            }
        }
    } else {
        for (int i = t; i < n; i += 1024) {
            const float d = depth[i];
            const bool in = d > 0.0f && d <= thr;
            if (mask) mask[i] = in ? 1 : 0;
            if (depth_filtered) depth_filtered[i] = in ? d : 0.0f;
        }
    }
    if (t == 0) *out_thr = thr;
}


// ------------------------------------------------------------------ exposure encoding (ScanNet, model.encode_exposure)
// mlp_exposure = Linear(8,128) -> Softplus(beta=100) -> Linear(128,12) (decoder.py:534-540): one affine (3x3 | 3) per
// exposure feature.  F <= LK_EXPOSURE_MAX_F features (keyframes of the mapping window, or the tracker's single frame).
__global__ __launch_bounds__(256) void k_exposure_fwd(const float* __restrict__ feats, const float* __restrict__ W1,
                                                      const float* __restrict__ b1, const float* __restrict__ W2,
                                                      const float* __restrict__ b2, int F, float* __restrict__ aff,
                                                      float* __restrict__ hid) {
    __shared__ float s_h[LK_EXPOSURE_MAX_F * 128];
    for (int e = threadIdx.x; e < F * 128; e += 256) {
        const int f = e >> 7, u = e & 127;
        float acc = b1[u];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = fmaf(W1[u * 8 + k], feats[f * 8 + k], acc);
        const float h = lk_softplus100(acc);
        s_h[e] = h;
        hid[e] = h;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < F * 12; e += 256) {
        const int f = e / 12, o = e - f * 12;
        float acc = b2[o];
        for (int u = 0; u < 128; ++u) acc = fmaf(W2[o * 128 + u], s_h[f * 128 + u], acc);
        aff[e] = acc;
    }
}

// gradients of the exposure MLP and features from d loss / d affine [F,12]: g = [W1 1024 | b1 128 | W2 1536 | b2 12 | feats F*8]
__global__ __launch_bounds__(256) void k_exposure_bwd(const float* __restrict__ feats, const float* __restrict__ W1,
                                                      const float* __restrict__ W2, const float* __restrict__ hid,
                                                      const float* __restrict__ g_aff, int F, float* __restrict__ g) {
    __shared__ float s_ga[LK_EXPOSURE_MAX_F * 12];
    __shared__ float s_dp[LK_EXPOSURE_MAX_F * 128];      // d loss / d pre-activation
    for (int e = threadIdx.x; e < F * 12; e += 256) s_ga[e] = g_aff[e];
    __syncthreads();
    for (int e = threadIdx.x; e < F * 128; e += 256) {
        const int f = e >> 7, u = e & 127;
        float dh = 0.0f;
#pragma unroll
        for (int o = 0; o < 12; ++o) dh = fmaf(W2[o * 128 + u], s_ga[f * 12 + o], dh);
        s_dp[e] = dh * lk_softplus100_grad_from_out(hid[e]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += 256) {                  // g W1 [128][8]
        const int u = e >> 3, k = e & 7;
        float acc = 0.0f;
        for (int f = 0; f < F; ++f) acc = fmaf(s_dp[f * 128 + u], feats[f * 8 + k], acc);
        g[e] = acc;
    }
    for (int u = threadIdx.x; u < 128; u += 256) {                   // g b1
        float acc = 0.0f;
        for (int f = 0; f < F; ++f) acc += s_dp[f * 128 + u];
        g[1024 + u] = acc;
    }
    for (int e = threadIdx.x; e < 1536; e += 256) {                  // g W2 [12][128]
        const int o = e >> 7, u = e & 127;
        float acc = 0.0f;
        for (int f = 0; f < F; ++f) acc = fmaf(s_ga[f * 12 + o], hid[f * 128 + u], acc);
        g[1152 + e] = acc;
    }
    for (int o = threadIdx.x; o < 12; o += 256) {                    // g b2
        float acc = 0.0f;
        for (int f = 0; f < F; ++f) acc += s_ga[f * 12 + o];
        g[2688 + o] = acc;
    }
    for (int e = threadIdx.x; e < F * 8; e += 256) {                 // g feats
        const int f = e >> 3, k = e & 7;
        float acc = 0.0f;
        for (int u = 0; u < 128; ++u) acc = fmaf(W1[u * 8 + k], s_dp[f * 128 + u], acc);
        g[2700 + e] = acc;
    }
}

// Mapper loss of the colour stage with exposure encoding (Mapper.py:691-720): the rays of keyframe f get
// sigmoid(logits @ rot_f + trans_f); returns d depth, d logits, [loss, geo, colour, #masked] and d loss / d affine [F,12].
// COMP (the mapping loop): the composite of the ray (k_composite's arithmetic and outputs) in the same thread - one launch instead of two.
// d affine: the rays of a batch are grouped by keyframe, so the lanes of a wave share one or two keyframes - the twelve terms are summed over
// the lanes of a keyframe by shuffles and added to the workgroup's LDS table once per wave and keyframe (one LDS float atomic per LANE and
// term retired a lane every two cycles with all 64 on the same address: most of the kernel's 14 us).
template <bool COMP>
__global__ __launch_bounds__(256) void k_loss_mapper_exposure(LkCompositeArgs ca, int R, const float* __restrict__ depth, const float* __restrict__ logits,
                                                              const uint8_t* __restrict__ valid, const float* __restrict__ gt_depth,
                                                              const float* __restrict__ gt_color, const int32_t* __restrict__ frame_id,
                                                              const float* __restrict__ aff, int F, float w_color,
                                                              float* __restrict__ d_depth, float* __restrict__ d_logits,
                                                              float* __restrict__ out, float* __restrict__ g_aff) {
    __shared__ float s_aff[LK_EXPOSURE_MAX_F * 12], s_g[LK_EXPOSURE_MAX_F * 12], s_sum[3];
    for (int e = threadIdx.x; e < F * 12; e += 256) { s_aff[e] = aff[e]; s_g[e] = 0.0f; }
    if (threadIdx.x < 3) s_sum[threadIdx.x] = 0.0f;
    __syncthreads();
    const int lane = (int)threadIdx.x & 63;
    float geo = 0.0f, col = 0.0f, cnt = 0.0f;
    for (int r0 = blockIdx.x * 256; r0 < R; r0 += gridDim.x * 256) {      // (whole waves stay in the loop: wave collectives inside)
        const int r = r0 + (int)threadIdx.x;
        const bool in = r < R;
        float d = 0.0f, gd = 0.0f, l0 = 0.0f, l1 = 0.0f, l2 = 0.0f;
        bool val = false;
        if (in) {
            gd = gt_depth[r];
            if (COMP) {
                const LkRayOut ro = lk_composite_ray(ca.raw, ca.z, ca.nbr_count, r, ca.S, ca.min_nn, ca.coef, ca.keep_depth ? 1.0f : gd);
                ca.depth[r] = ro.depth; ca.var[r] = ro.var;
                ca.color[3 * r] = ro.c0; ca.color[3 * r + 1] = ro.c1; ca.color[3 * r + 2] = ro.c2;
                ca.valid_ray[r] = ro.valid ? 1 : 0;
                d = ro.depth; l0 = ro.c0; l1 = ro.c1; l2 = ro.c2; val = ro.valid;
            } else {
                d = depth[r]; val = valid[r] != 0;
            }
        }
        const bool m = in && (gd > 0.0f) && val && !(d != d);
        const int f = (m && frame_id) ? frame_id[r] : 0;
        float dd = 0.0f, dl[3] = {0.0f, 0.0f, 0.0f}, ga[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) ga[k] = 0.0f;
        if (m) {
            const float* A = s_aff + f * 12;
            if (!COMP) { l0 = logits[3 * r]; l1 = logits[3 * r + 1]; l2 = logits[3 * r + 2]; }
            float dp[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float pre = l0 * A[c] + l1 * A[3 + c] + l2 * A[6 + c] + A[9 + c];      // (logits @ rot)[c] + trans[c]
                const float y = lk_sigmoid(pre);
                const float e = y - gt_color[3 * r + c];
                col += fabsf(e);
                dp[c] = w_color * sgn(e) * y * (1.0f - y);
            }
            geo += fabsf(gd - d);
            dd = sgn(d - gd);
            cnt += 1.0f;
            const float lv[3] = {l0, l1, l2};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                dl[i] = A[3 * i] * dp[0] + A[3 * i + 1] * dp[1] + A[3 * i + 2] * dp[2];
#pragma unroll
                for (int c = 0; c < 3; ++c) ga[3 * i + c] = lv[i] * dp[c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) ga[9 + c] = dp[c];
        }
        // d affine of the wave's rays, keyframe by keyframe
        unsigned long long todo = __ballot(m);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int fl = __shfl(f, leader);
            const bool mine = m && f == fl;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                float v = mine ? ga[k] : 0.0f;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                if (lane == leader) atomicAdd(&s_g[fl * 12 + k], v);
            }
            todo &= ~__ballot(mine);
        }
        if (in) {
            d_depth[r] = dd;
            d_logits[3 * r] = dl[0]; d_logits[3 * r + 1] = dl[1]; d_logits[3 * r + 2] = dl[2];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { geo += __shfl_xor(geo, o); col += __shfl_xor(col, o); cnt += __shfl_xor(cnt, o); }
    if (lane == 0) { atomicAdd(&s_sum[0], geo); atomicAdd(&s_sum[1], col); atomicAdd(&s_sum[2], cnt); }
    __syncthreads();
    for (int e = threadIdx.x; e < F * 12; e += 256) atomicAdd(g_aff + e, s_g[e]);
    if (threadIdx.x == 0) {
        atomicAdd(out + 0, s_sum[0] + w_color * s_sum[1]);
        atomicAdd(out + 1, s_sum[0]); atomicAdd(out + 2, s_sum[1]); atomicAdd(out + 3, s_sum[2]);
    }
}

__global__ __launch_bounds__(256) void k_exposure_step(ExposureStepArgs a) { lk_exposure_step_body(a, nullptr, 0); }
// step: 1-based Adam step of the exposure groups (they first step in the first iteration that uses them)
int lk_exposure_step_args(const lk_exposure_desc& x, int mode, int step, float beta1, float beta2, float eps, ExposureStepArgs* out) {
    LK_REQUIRE(x.F >= 1 && x.F <= LK_EXPOSURE_MAX_F, "exposure: F out of range");
    LK_REQUIRE(x.feats && x.W1 && x.b1 && x.W2 && x.b2 && x.aff && x.hid && x.g_aff && x.g && x.adam, "exposure: NULL buffer in lk_exposure_desc");
    LK_REQUIRE(x.feat_first >= 0 && x.feat_count >= 0 && x.feat_first + x.feat_count <= x.F, "exposure: bad trainable feature range");
    ExposureStepArgs a;
    a.feats = x.feats; a.W1 = x.W1; a.b1 = x.b1; a.W2 = x.W2; a.b2 = x.b2; a.F = x.F;
    a.aff = x.aff; a.hid = x.hid; a.g_aff = x.g_aff; a.g = x.g; a.m = x.adam; a.v = x.adam + LK_EXPOSURE_GRAD_FLOATS; a.bwd_scale = x.bwd_scale;
    const int s1 = step < 1 ? 1 : step;
    const double bc1 = 1.0 - pow((double)beta1, (double)s1), bc2 = 1.0 - pow((double)beta2, (double)s1);
    a.step_mlp = x.lr_mlp < 0.0f ? -1.0f : (float)((double)x.lr_mlp / bc1);
    a.step_feat = (float)((double)x.lr_feat / bc1);
    a.bc2_sqrt = (float)sqrt(bc2); a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.feat_first = x.feat_first; a.feat_count = x.feat_count; a.mode = mode;
    *out = a;
    return LK_OK;
}
int lk_launch_exposure_step(const lk_exposure_desc& x, int mode, int step, float beta1, float beta2, float eps, hipStream_t st) {
    ExposureStepArgs a;
    const int rc = lk_exposure_step_args(x, mode, step, beta1, beta2, eps, &a);
    if (rc != LK_OK) return rc;
    hipLaunchKernelGGL(k_exposure_step, dim3(1), dim3(256), 0, st, a);
    return LK_OK;
}
// lk_loss_mapper_exposure without its two memsets (the loops clear out_loss and g_aff elsewhere)
int lk_launch_loss_mapper_exposure(int R, const float* depth, const float* logits, const uint8_t* valid_ray, const float* gt_depth,
                                   const float* gt_color, const int32_t* frame_id, const float* aff, int F, float w_color,
                                   float* d_depth, float* d_logits, float* out_loss, float* g_aff, hipStream_t st) {
    if (R == 0) return LK_OK;
    int gx = lk_cdiv(R, 256);
    if (gx > 64) gx = 64;
    LkCompositeArgs none;
    memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL(k_loss_mapper_exposure<false>, dim3(gx), dim3(256), 0, st, none, R, depth, logits, valid_ray, gt_depth, gt_color,
                       frame_id, aff, F, w_color, d_depth, d_logits, out_loss, g_aff);
    return LK_OK;
}
// ... with the composite of the rays inside (the mapping loop: no k_composite launch in front); ca carries raw / z / nbr_count and the outputs
int lk_launch_composite_loss_exposure(const LkCompositeArgs& ca, const float* gt_color, const int32_t* frame_id, const float* aff, int F, float w_color,
                                      float* d_depth, float* d_logits, float* out_loss, float* g_aff, hipStream_t st) {
    if (ca.R == 0) return LK_OK;
    int gx = lk_cdiv(ca.R, 256);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(k_loss_mapper_exposure<true>, dim3(gx), dim3(256), 0, st, ca, ca.R, (const float*)nullptr, (const float*)nullptr,
                       (const uint8_t*)nullptr, ca.gt_depth, gt_color, frame_id, aff, F, w_color, d_depth, d_logits, out_loss, g_aff);
    return LK_OK;
}

// ------------------------------------------------------------------ host API
extern "C" int lk_loss_mapper(int32_t R, const float* depth, const float* color, const uint8_t* valid_ray,
                              const float* gt_depth, const float* gt_color, float w_color, int32_t use_color,
                              float* d_depth, float* d_color, float* out_loss, void* stream_) {
    LK_REQUIRE(R >= 0 && out_loss, "lk_loss_mapper: bad arguments");
    hipStream_t st = (hipStream_t)stream_;
    LK_REQUIRE(R == 0 || (depth && color && valid_ray && gt_depth && gt_color && d_depth && d_color), "lk_loss_mapper: NULL buffer");
    if (R > 0 && R <= LK_LOSS_1WG_MAX) {
        hipLaunchKernelGGL(k_loss_mapper_1wg, dim3(1), dim3(1024), 0, st, (int)R, depth, color, valid_ray, gt_depth, gt_color,
                           w_color, (int)use_color, d_depth, d_color, out_loss);
        LK_LAUNCH_CHECK();
        return LK_OK;
    }
    LK_HIP_TRY(hipMemsetAsync(out_loss, 0, 4 * sizeof(float), st));
    if (R == 0) return LK_OK;
    hipLaunchKernelGGL(k_loss_mapper, dim3(lk_cdiv(R, 256)), dim3(256), 0, st, (int)R, depth, color, valid_ray, gt_depth,
                       gt_color, w_color, (int)use_color, d_depth, d_color, out_loss);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

extern "C" int lk_loss_tracker(int32_t R, const float* depth, const float* var, const float* color,
                               const float* gt_depth, const float* gt_color, float w_color, int32_t use_color,
                               float* d_depth, float* d_color, float* out_loss, float* scratch, void* stream_) {
    LK_REQUIRE(R >= 0 && out_loss && scratch, "lk_loss_tracker: bad arguments");
    hipStream_t st = (hipStream_t)stream_;
    LK_REQUIRE(R == 0 || (depth && var && color && gt_depth && gt_color && d_depth && d_color), "lk_loss_tracker: NULL buffer");
    const bool median = (use_color & LK_TRACK_MEDIAN_MASK) != 0;
    use_color &= LK_TRACK_USE_COLOR;
    if (R > 0 && R <= LK_LOSS_1WG_MAX) {
        if (median)
            hipLaunchKernelGGL(k_loss_tracker_1wg<true>, dim3(1), dim3(1024), 0, st, (int)R, depth, var, color, gt_depth, gt_color,
                               w_color, (int)use_color, d_depth, d_color, out_loss, scratch);
        else
            hipLaunchKernelGGL(k_loss_tracker_1wg<false>, dim3(1), dim3(1024), 0, st, (int)R, depth, var, color, gt_depth, gt_color,
                               w_color, (int)use_color, d_depth, d_color, out_loss, scratch);
        LK_LAUNCH_CHECK();
        return LK_OK;
    }
    LK_HIP_TRY(hipMemsetAsync(out_loss, 0, 4 * sizeof(float), st));
    if (R == 0) return LK_OK;
    if (median) {
        hipLaunchKernelGGL(k_loss_tracker_pass1<true>, dim3(lk_cdiv(R, 256)), dim3(256), 0, st, (int)R, depth, var, gt_depth, scratch);
        hipLaunchKernelGGL(k_loss_tracker_median, dim3(1), dim3(1024), 0, st, (int)R, scratch);
        hipLaunchKernelGGL(k_loss_tracker_pass2<true>, dim3(lk_cdiv(R, 256)), dim3(256), 0, st, (int)R, depth, var, color, gt_depth,
                           gt_color, w_color, (int)use_color, (const float*)scratch, d_depth, d_color, out_loss);
    } else {
        LK_HIP_TRY(hipMemsetAsync(scratch + R, 0, 2 * sizeof(float), st));
        hipLaunchKernelGGL(k_loss_tracker_pass1<false>, dim3(lk_cdiv(R, 256)), dim3(256), 0, st, (int)R, depth, var, gt_depth, scratch);
        hipLaunchKernelGGL(k_loss_tracker_pass2<false>, dim3(lk_cdiv(R, 256)), dim3(256), 0, st, (int)R, depth, var, color, gt_depth,
                           gt_color, w_color, (int)use_color, (const float*)scratch, d_depth, d_color, out_loss);
    }
    LK_LAUNCH_CHECK();
    return LK_OK;
}


extern "C" int lk_exposure_fwd(const float* feats, const float* W1, const float* b1, const float* W2, const float* b2, int32_t F,
                               float* aff, float* hid, void* stream_) {
    LK_REQUIRE(F >= 1 && F <= LK_EXPOSURE_MAX_F, "lk_exposure_fwd: F out of range");
    LK_REQUIRE(feats && W1 && b1 && W2 && b2 && aff && hid, "lk_exposure_fwd: NULL buffer");
    hipLaunchKernelGGL(k_exposure_fwd, dim3(1), dim3(256), 0, (hipStream_t)stream_, feats, W1, b1, W2, b2, (int)F, aff, hid);
    LK_LAUNCH_CHECK();
    return LK_OK;
}
extern "C" int lk_exposure_bwd(const float* feats, const float* W1, const float* W2, const float* hid, const float* g_aff, int32_t F,
                               float* g, void* stream_) {
    LK_REQUIRE(F >= 1 && F <= LK_EXPOSURE_MAX_F, "lk_exposure_bwd: F out of range");
    LK_REQUIRE(feats && W1 && W2 && hid && g_aff && g, "lk_exposure_bwd: NULL buffer");
    hipLaunchKernelGGL(k_exposure_bwd, dim3(1), dim3(256), 0, (hipStream_t)stream_, feats, W1, W2, hid, g_aff, (int)F, g);
    LK_LAUNCH_CHECK();
    return LK_OK;
}
extern "C" int lk_loss_mapper_exposure(int32_t R, const float* depth, const float* logits, const uint8_t* valid_ray,
                                       const float* gt_depth, const float* gt_color, const int32_t* frame_id, const float* aff,
                                       int32_t F, float w_color, float* d_depth, float* d_logits, float* out_loss, float* g_aff,
                                       void* stream_) {
    LK_REQUIRE(R >= 0 && F >= 1 && F <= LK_EXPOSURE_MAX_F && out_loss && g_aff && aff, "lk_loss_mapper_exposure: bad arguments");
    LK_REQUIRE(R == 0 || (depth && logits && valid_ray && gt_depth && gt_color && d_depth && d_logits), "lk_loss_mapper_exposure: NULL buffer");
    hipStream_t st = (hipStream_t)stream_;
    LK_HIP_TRY(hipMemsetAsync(out_loss, 0, 4 * sizeof(float), st));
    LK_HIP_TRY(hipMemsetAsync(g_aff, 0, (size_t)F * 12 * sizeof(float), st));
    if (R == 0) return LK_OK;
    const int rc = lk_launch_loss_mapper_exposure((int)R, depth, logits, valid_ray, gt_depth, gt_color, frame_id, aff, (int)F, w_color, d_depth, d_logits,
                                                  out_loss, g_aff, st);
    if (rc != LK_OK) return rc;
    LK_LAUNCH_CHECK();
    return LK_OK;
}

int lk_adam_step_x(const lk_adam_seg* segs, int32_t n_seg, float beta1, float beta2, float eps, const ExposureStepArgs* xa, void* stream_);
extern "C" int lk_adam_step(const lk_adam_seg* segs, int32_t n_seg, float beta1, float beta2, float eps, void* stream_) {
    return lk_adam_step_x(segs, n_seg, beta1, beta2, eps, nullptr, stream_);
}
int lk_adam_step_x(const lk_adam_seg* segs, int32_t n_seg, float beta1, float beta2, float eps, const ExposureStepArgs* xa, void* stream_) {
    LK_REQUIRE(n_seg >= 0 && n_seg <= LK_ADAM_MAX_SEG, "lk_adam_step: too many segments");
    if (n_seg == 0 && !xa) return LK_OK;
    LK_REQUIRE(segs != nullptr, "lk_adam_step: NULL segments");
    AdamArgs a;
    memset(&a, 0, sizeof(a));
    long long nmax = 0;
    for (int i = 0; i < n_seg; ++i) {
        LK_REQUIRE(segs[i].n >= 0 && segs[i].step >= 1, "lk_adam_step: bad segment");
        LK_REQUIRE(segs[i].n == 0 || (segs[i].p && segs[i].g && segs[i].m && segs[i].v), "lk_adam_step: NULL tensor");
        const double bc1 = 1.0 - pow((double)beta1, (double)segs[i].step);
        const double bc2 = 1.0 - pow((double)beta2, (double)segs[i].step);
        a.s[i].p = segs[i].p; a.s[i].g = segs[i].g; a.s[i].m = segs[i].m; a.s[i].v = segs[i].v; a.s[i].n = segs[i].n;
        a.s[i].row_index = segs[i].row_index; a.s[i].row_len = segs[i].row_len > 0 ? segs[i].row_len : 1;
        a.s[i].zero_grad = segs[i].zero_grad; a.s[i].g_compact = segs[i].g_compact;
        a.s[i].p_f16 = segs[i].p_f16;
        a.s[i].row_flags = segs[i].row_flags;
        LK_REQUIRE(!segs[i].row_flags || (!segs[i].row_index && a.s[i].row_len <= 64 && segs[i].n % a.s[i].row_len == 0),
                   "lk_adam_step: row_flags need row_index == NULL, row_len <= 64 and n a multiple of row_len");
        a.s[i].step_size = (float)((double)segs[i].lr / bc1);
        a.s[i].bc2_sqrt = (float)sqrt(bc2);
        if (segs[i].n > nmax) nmax = segs[i].n;
    }
    a.n_seg = n_seg; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    if (nmax == 0 && !xa) return LK_OK;
    int gx = lk_cdiv(nmax, 256);
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    if (xa) hipLaunchKernelGGL(k_adam_x, dim3(gx, n_seg + 1), dim3(256), 0, (hipStream_t)stream_, a, *xa);
    else hipLaunchKernelGGL(k_adam, dim3(gx, n_seg), dim3(256), 0, (hipStream_t)stream_, a);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

// ------------------------------------------------------------------ all-reduce bucket pack / unpack
struct CopySeg { float* data; long long n, off; const int32_t* row_index; int row_len; };
struct CopyArgs { CopySeg s[LK_ADAM_MAX_SEG]; int n_seg; float* bucket; int unpack; };
__global__ __launch_bounds__(256) void k_bucket_copy(CopyArgs a) {
    const CopySeg sg = a.s[blockIdx.y];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < sg.n; i += (long long)gridDim.x * 256) {
        long long src = i;
        if (sg.row_index) { const long long r = i / sg.row_len; src = (long long)sg.row_index[r] * sg.row_len + (i - r * sg.row_len); }
        if (a.unpack == 1) sg.data[src] = a.bucket[sg.off + i];
        else {
            a.bucket[sg.off + i] = sg.data[src];
            if (a.unpack == 2) sg.data[src] = 0.0f;
        }
    }
}
extern "C" int lk_bucket_copy(const lk_copy_seg* segs, int32_t n_seg, float* bucket, int32_t unpack, void* stream_) {
    LK_REQUIRE(n_seg >= 0 && n_seg <= LK_ADAM_MAX_SEG, "lk_bucket_copy: too many segments");
    if (n_seg == 0) return LK_OK;
    LK_REQUIRE(segs != nullptr && bucket != nullptr, "lk_bucket_copy: NULL argument");
    CopyArgs a;
    memset(&a, 0, sizeof(a));
    long long off = 0, nmax = 0;
    for (int i = 0; i < n_seg; ++i) {
        LK_REQUIRE(segs[i].n >= 0 && (segs[i].n == 0 || segs[i].data), "lk_bucket_copy: bad segment");
        a.s[i].data = segs[i].data; a.s[i].n = segs[i].n; a.s[i].off = off;
        a.s[i].row_index = segs[i].row_index; a.s[i].row_len = segs[i].row_len > 0 ? segs[i].row_len : 1;
        off += segs[i].n;
        if (segs[i].n > nmax) nmax = segs[i].n;
    }
    a.n_seg = n_seg; a.bucket = bucket; a.unpack = unpack;
    if (nmax == 0) return LK_OK;
    int gx = lk_cdiv(nmax, 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_bucket_copy, dim3(gx, n_seg), dim3(256), 0, (hipStream_t)stream_, a);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

extern "C" int lk_rays_from_pose(const float* cam7, const float* pix_i, const float* pix_j, int32_t R,
                                 float fx, float fy, float cx, float cy, float* rays_o, float* rays_d, void* stream_) {
    LK_REQUIRE(R >= 0, "lk_rays_from_pose: R < 0");
    if (R == 0) return LK_OK;
    LK_REQUIRE(cam7 && pix_i && pix_j && rays_o && rays_d, "lk_rays_from_pose: NULL buffer");
    hipLaunchKernelGGL(k_rays_from_pose, dim3(lk_cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream_, cam7, pix_i, pix_j, (int)R,
                       fx, fy, cx, cy, rays_o, rays_d);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

extern "C" int lk_pose_bwd(const float* cam7, const float* pix_i, const float* pix_j, int32_t R,
                           float fx, float fy, float cx, float cy, const float* g_rays_o, const float* g_rays_d,
                           float* g_cam7, void* stream_) {
    LK_REQUIRE(R >= 0 && cam7 && g_cam7, "lk_pose_bwd: bad arguments");
    LK_REQUIRE(R == 0 || (pix_i && pix_j && g_rays_o && g_rays_d), "lk_pose_bwd: NULL buffer");
    hipLaunchKernelGGL(k_pose_bwd, dim3(1), dim3(256), 0, (hipStream_t)stream_, cam7, pix_i, pix_j, (int)R, fx, fy, cx, cy,
                       g_rays_o, g_rays_d, g_cam7);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

// ------------------------------------------------------------------ stable compaction, many blocks (n up to millions)
// counts per 256-element block -> exclusive scan of the block counts (one workgroup) -> every block re-derives its
// local ranks with ballots and writes its indices behind its offset.  block_scratch: ceil(n / 256) ints.
__global__ __launch_bounds__(256) void k_compact_count(const uint8_t* __restrict__ mask, int n, int32_t* __restrict__ block_count) {
    __shared__ int wc[4];
    const int i = blockIdx.x * 256 + (int)threadIdx.x;
    const unsigned long long b = __ballot(i < n && mask[i] != 0);
    if (lk_lane() == 0) wc[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}
__global__ __launch_bounds__(1024) void k_compact_scan(int32_t* __restrict__ block_count, int nb, int32_t* __restrict__ out_count) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < nb; c0 += 1024) {
        const int i = c0 + t;
        const int v = (i < nb) ? block_count[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int nv = __shfl_up(incl, o); if (lane >= o) incl += nv; }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int base = carry;
        for (int q = 0; q < w; ++q) base += wsum[q];
        if (i < nb) block_count[i] = base + incl - v;           // exclusive offset of block i
        __syncthreads();
        if (t == 1023) carry = base + incl;
        __syncthreads();
    }
    if (t == 0) *out_count = carry;
}
__global__ __launch_bounds__(256) void k_compact_scatter(const uint8_t* __restrict__ mask, int n, const int32_t* __restrict__ block_off,
                                                         int32_t* __restrict__ out_index) {
    __shared__ int wc[4];
    const int i = blockIdx.x * 256 + (int)threadIdx.x, lane = lk_lane(), w = (int)threadIdx.x >> 6;
    const bool keep = i < n && mask[i] != 0;
    const unsigned long long b = __ballot(keep);
    if (lane == 0) wc[w] = __popcll(b);
    __syncthreads();
    int off = block_off[blockIdx.x];
    for (int q = 0; q < w; ++q) off += wc[q];
    if (keep) out_index[off + __popcll(b & ((1ull << lane) - 1ull))] = i;
}

int lk_launch_compact_mb(const uint8_t* mask, int n, int32_t* out_index, int32_t* out_count, int32_t* block_scratch, hipStream_t st) {
    const int nb = lk_cdiv(n, 256);
    hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(256), 0, st, mask, n, block_scratch);
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, st, block_scratch, nb, out_count);
    hipLaunchKernelGGL(k_compact_scatter, dim3(nb), dim3(256), 0, st, mask, n, (const int32_t*)block_scratch, out_index);
    return LK_OK;
}

int lk_launch_compact(const uint8_t* mask, int n, int32_t* out_index, int32_t* out_count, hipStream_t st) {
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, st, mask, n, out_index, out_count);
    return LK_OK;
}

extern "C" int lk_compact(const uint8_t* mask, int32_t n, int32_t* out_index, int32_t* out_count, void* stream_) {
    LK_REQUIRE(n >= 0 && out_count, "lk_compact: bad arguments");
    LK_REQUIRE(n == 0 || (mask && out_index), "lk_compact: NULL buffer");
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, (hipStream_t)stream_, mask, (int)n, out_index, out_count);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

// ------------------------------------------------------------------ rows a batch touches (whole-map optimisation, data parallel)
// flags[i] = 1 for every point i that appears in the batch's neighbour lists: with every row of the map a parameter (final
// refinement, Mapper.py:884-897) a batch of R rays touches <= 8 R S of the N rows; the ranks of a ray-sharded step exchange the
// gradient rows of the UNION of their flags only (loopy_slam_amd/parallel.py).  The caller clears `flags`.
__global__ __launch_bounds__(256) void k_touch_rows(const int32_t* __restrict__ nbr_idx, long long n, uint8_t* __restrict__ flags, int N) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int j = nbr_idx[i];
        if (j >= 0 && j < N) flags[j] = 1;
    }
}
extern "C" int lk_touch_rows(const int32_t* nbr_idx, int64_t n, uint8_t* flags, int32_t N, void* stream_) {
    LK_REQUIRE(n >= 0 && N >= 0, "lk_touch_rows: bad sizes");
    if (n == 0 || N == 0) return LK_OK;
    LK_REQUIRE(nbr_idx && flags, "lk_touch_rows: NULL buffer");
    int gx = lk_cdiv(n, 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_touch_rows, dim3(gx), dim3(256), 0, (hipStream_t)stream_, nbr_idx, (long long)n, flags, (int)N);
    LK_LAUNCH_CHECK();
    return LK_OK;
}
__global__ __launch_bounds__(256) void k_or_flags(const uint8_t* __restrict__ src, long long n, uint8_t* __restrict__ dst) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        if (src[i]) dst[i] = 1;
}
extern "C" int lk_knn_flag_rows(lk_knn_t knn, const uint8_t* flags, int64_t n, void* stream_) {
    LK_REQUIRE(knn != nullptr && n >= 0 && n <= knn->capacity, "lk_knn_flag_rows: bad arguments");
    if (n == 0) return LK_OK;
    LK_REQUIRE(flags != nullptr && knn->act_flag != nullptr, "lk_knn_flag_rows: NULL buffer");
    int gx = lk_cdiv(n, 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_or_flags, dim3(gx), dim3(256), 0, (hipStream_t)stream_, flags, (long long)n, knn->act_flag);
    LK_LAUNCH_CHECK();
    return LK_OK;
}
extern "C" int lk_compact_large(const uint8_t* mask, int32_t n, int32_t* out_index, int32_t* out_count, int32_t* block_scratch, void* stream_) {
    LK_REQUIRE(n >= 0 && out_count, "lk_compact_large: bad arguments");
    LK_REQUIRE(n == 0 || (mask && out_index && block_scratch), "lk_compact_large: NULL buffer");
    if (n <= 16384) lk_launch_compact(mask, n, out_index, out_count, (hipStream_t)stream_);
    else lk_launch_compact_mb(mask, n, out_index, out_count, block_scratch, (hipStream_t)stream_);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

extern "C" int lk_inside_mask(const float* depth, int32_t n, uint8_t* mask, float* depth_filtered, float* out_thr,
                              uint32_t* scratch, void* stream_) {
    LK_REQUIRE(n >= 0 && out_thr, "lk_inside_mask: bad arguments");
    if (n == 0) return LK_OK;
    LK_REQUIRE(depth && scratch && (mask || depth_filtered), "lk_inside_mask: NULL buffer");
    if (n <= LK_MASK_REG_MAX)
        hipLaunchKernelGGL((k_inside_mask<true>), dim3(1), dim3(1024), 0, (hipStream_t)stream_, depth, (int)n, mask, depth_filtered, out_thr, scratch);
    else
        hipLaunchKernelGGL((k_inside_mask<false>), dim3(1), dim3(1024), 0, (hipStream_t)stream_, depth, (int)n, mask, depth_filtered, out_thr, scratch);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

// ------------------------------------------------------------------ ray batch from stacked keyframes
// get_samples / get_sample_uv / select_uv / get_rays_from_uv (common.py:104-172, 237-259) for a whole
// multi-keyframe batch in one launch: ray r looks at frame frame_id[r], window pixel rnd[r] (row-major over
// [H0,H0+h) x [W0,W0+w)), and gets its depth, colour, (optional) squared query radius and world ray.
struct GatherArgs {
    const float* depth; const float* color; const float* c2w; const float* r2_map;
    const int32_t* frame_id; const int32_t* rnd;
    int R, H, W, H0, W0, w, c2w_stride;
    float fx, fy, cx, cy;
    float* rays_o; float* rays_d; float* gt_depth; float* gt_color; float* pix_i; float* pix_j; float* r2_ray;
};

__global__ __launch_bounds__(256) void k_gather_rays(GatherArgs a) {
    const int r = blockIdx.x * 256 + (int)threadIdx.x;
    if (r >= a.R) return;
    const int f = a.frame_id ? a.frame_id[r] : 0;
    const int q = a.rnd[r];
    const int i = a.W0 + q % a.w, j = a.H0 + q / a.w;
    const size_t pix = ((size_t)f * a.H + j) * a.W + i;
    a.gt_depth[r] = a.depth[pix];
    a.gt_color[3 * r] = a.color[3 * pix]; a.gt_color[3 * r + 1] = a.color[3 * pix + 1]; a.gt_color[3 * r + 2] = a.color[3 * pix + 2];
    if (a.r2_ray) a.r2_ray[r] = a.r2_map ? a.r2_map[pix] : 0.0f;
    if (a.pix_i) { a.pix_i[r] = (float)i; a.pix_j[r] = (float)j; }
    const float* M = a.c2w + (size_t)f * a.c2w_stride;             // row-major [3 or 4][4]
    const float d0 = ((float)i - a.cx) / a.fx, d1 = -((float)j - a.cy) / a.fy, d2 = -1.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a.rays_d[3 * r + c] = (d0 * M[4 * c] + d1 * M[4 * c + 1]) + d2 * M[4 * c + 2];
        a.rays_o[3 * r + c] = M[4 * c + 3];
    }
}

extern "C" int lk_gather_rays(const float* depth_stack, const float* color_stack, const float* c2w_stack, int32_t c2w_stride,
                              const float* r2_map_stack, const int32_t* frame_id, const int32_t* rnd, int32_t R,
                              int32_t H, int32_t W, int32_t H0, int32_t W0, int32_t w, float fx, float fy, float cx, float cy,
                              float* rays_o, float* rays_d, float* gt_depth, float* gt_color, float* pix_i, float* pix_j,
                              float* r2_ray, void* stream_) {
    LK_REQUIRE(R >= 0 && w > 0, "lk_gather_rays: bad sizes");
    if (R == 0) return LK_OK;
    LK_REQUIRE(depth_stack && color_stack && c2w_stack && rnd && rays_o && rays_d && gt_depth && gt_color, "lk_gather_rays: NULL buffer");
    GatherArgs a;
    a.depth = depth_stack; a.color = color_stack; a.c2w = c2w_stack; a.r2_map = r2_map_stack; a.frame_id = frame_id; a.rnd = rnd;
    a.R = R; a.H = H; a.W = W; a.H0 = H0; a.W0 = W0; a.w = w; a.c2w_stride = c2w_stride;
    a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy;
    a.rays_o = rays_o; a.rays_d = rays_d; a.gt_depth = gt_depth; a.gt_color = gt_color; a.pix_i = pix_i; a.pix_j = pix_j; a.r2_ray = r2_ray;
    hipLaunchKernelGGL(k_gather_rays, dim3(lk_cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream_, a);
    LK_LAUNCH_CHECK();
    return LK_OK;
}
