// Device code of the tracking loop's pose step (lk_loop.hip): quaternion -> rotation, gradient from the ray moments, Adam,
// rays of the next batch.
#pragma once
#include "lk_common.h"
#include "lk_kernels.h"
#include "lk_composite_dev.h"

__device__ __forceinline__ float lp_sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

// ---- the tracker's loss (Tracker.py:169-191) per ray, in the prologue of the tracking loop's k_decode_bwd
// mask threshold 10 x the batch mean of the normalised residuals from pass 1's block sums: a whole wave calls, every wave of every workgroup
// adds the same pairs in the same order - the same threshold everywhere
// (split in two: the first four pairs of every lane - all of them up to 256 partial pairs, the fused loop's 250 tiles - are FETCHED by lk_track_parts_load
// and summed later by lk_track_threshold, so that the caller can issue its own loads in between: with the sum taken first, the ray's operands
// were a second cold round trip behind the partials' - 5-6 us of prologue in every tile of the tracking loop's k_decode_bwd,
// profiles/r6_track_chain_before.md.  The order of the additions is the one it always was.)
struct LkTrackParts { float t[4], c[4]; };
__device__ __forceinline__ LkTrackParts lk_track_parts_load(const LkTrackLossArgs& a, int n_part) {
    LkTrackParts p;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = lk_lane() + 64 * q;
        p.t[q] = 0.0f; p.c[q] = 0.0f;
        if (!a.median && b < n_part) { const float2 v = *reinterpret_cast<const float2*>(a.part + 2 * b); p.t[q] = v.x; p.c[q] = v.y; }
    }
    return p;
}
__device__ __forceinline__ float lk_track_threshold(const LkTrackLossArgs& a, int n_part, const LkTrackParts& p) {
    if (a.median) return a.part[0];
    float ts = 0.0f, cs = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (lk_lane() + 64 * q < n_part) { ts += p.t[q]; cs += p.c[q]; }
    }
    for (int b = lk_lane() + 256; b < n_part; b += 64) { ts += a.part[2 * b]; cs += a.part[2 * b + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ts += __shfl_xor(ts, o); cs += __shfl_xor(cs, o); }
    return 10.0f * (ts / fmaxf(cs, 1.0f));
}
struct LkTrackRayLoss { float dd, dc0, dc1, dc2, geo, col, cnt, gt; };
struct LkTrackRayIn { float d, v, g, tm, c0, c1, c2, g0, g1, g2; };
__device__ __forceinline__ LkTrackRayIn lk_track_ray_in(const LkTrackLossArgs& a, int r) {
    LkTrackRayIn i;
    i.d = a.depth[r]; i.v = a.var[r]; i.g = a.gt_depth[r]; i.tm = a.resid[r];
    i.c0 = a.color[3 * r]; i.c1 = a.color[3 * r + 1]; i.c2 = a.color[3 * r + 2];
    i.g0 = a.gt_color[3 * r]; i.g1 = a.gt_color[3 * r + 1]; i.g2 = a.gt_color[3 * r + 2];
    return i;
}
__device__ __forceinline__ LkTrackRayLoss lk_track_ray_loss(const LkTrackLossArgs& a, const LkTrackRayIn& in, float thr) {
    LkTrackRayLoss o;
    const float d = in.d, v = in.v, g = in.g, tm = in.tm;
    const float tt = a.median ? fabsf(g - d) / sqrtf(v + 1e-10f) : tm;          // the loss term stays uncertainty-normalised
    const bool m = (tm < thr) && (g > 0.0f) && !(d != d) && !(v != v);
    o.dd = 0.0f; o.dc0 = 0.0f; o.dc1 = 0.0f; o.dc2 = 0.0f; o.geo = 0.0f; o.col = 0.0f; o.cnt = 0.0f; o.gt = g;
    if (m) {
        o.geo = fminf(fmaxf(tt, 0.0f), 1e3f);
        if (tt <= 1e3f) o.dd = lp_sgn(d - g) / sqrtf(v + 1e-10f);
        o.cnt = 1.0f;
        const float e0 = in.c0 - in.g0, e1 = in.c1 - in.g1, e2 = in.c2 - in.g2;
        o.col = fabsf(e0) + fabsf(e1) + fabsf(e2);
        if (a.use_color) { o.dc0 = a.w_color * lp_sgn(e0); o.dc1 = a.w_color * lp_sgn(e1); o.dc2 = a.w_color * lp_sgn(e2); }
    }
    return o;
}
// d raw of sample sp (its lane calls; all 64 lanes of the wave must call - the threshold is a wave sum); *row: the ray's loss terms
__device__ __forceinline__ float4 lk_track_draw(const LkTrackLossArgs& a, int n_part, int sp, LkTrackRayLoss* row) {
    const int r = sp / a.S;
    // three independent sets of operands - the partial pairs, the ray's outputs and readings, its samples' raw values - in flight together
    const LkTrackParts parts = lk_track_parts_load(a, n_part);
    const LkTrackRayIn in = lk_track_ray_in(a, r);
    LkRayState st;
    lk_ray_state(a.raw, a.z, a.nbr_count, r, a.S, a.min_nn, a.coef, st);
    const float thr = lk_track_threshold(a, n_part, parts);
    const LkTrackRayLoss o = lk_track_ray_loss(a, in, thr);
    if (row) *row = o;
    float4 out[LK_S_MAX];
    lk_ray_grad(st, a.S, a.coef, o.gt, o.dd, 0.0f, o.dc0, o.dc1, o.dc2, out);
    const int s_own = sp - r * a.S;
    float4 res = out[0];
#pragma unroll
    for (int s = 1; s < LK_S_MAX; ++s)
        if (s == s_own) res = out[s];
    return res;
}

__device__ __forceinline__ void lp_quat_rot(const float* __restrict__ cam, float (&Rm)[9]) {       // common.py:301-324
    const float qr = cam[0], qi = cam[1], qj = cam[2], qk = cam[3];
    const float s = 2.0f / (qr * qr + qi * qi + qj * qj + qk * qk);
    Rm[0] = 1.0f - s * (qj * qj + qk * qk); Rm[1] = s * (qi * qj - qk * qr); Rm[2] = s * (qi * qk + qj * qr);
    Rm[3] = s * (qi * qj + qk * qr); Rm[4] = 1.0f - s * (qi * qi + qk * qk); Rm[5] = s * (qj * qk - qi * qr);
    Rm[6] = s * (qi * qk - qj * qr); Rm[7] = s * (qj * qk + qi * qr); Rm[8] = 1.0f - s * (qi * qi + qj * qj);
}


// pose gradient from the ray moments (k_pose_bwd's formulas), Adam on (T | q) (Tracker.py:317-352), the candidate pose log - by a whole
// workgroup of NW waves (all of its threads call); the stepped pose (do_update = 0: the pose as it is) is left in s_cam[7] (LDS) behind a barrier.
// write: store the pose, its moments, the gradient and the log row (exactly one workgroup of a launch does).
template <int NW>
__device__ __forceinline__ void lk_track_pose_step(const LkTrackFinalArgs& a, float* __restrict__ s_cam, bool write) {
    __shared__ float s_w[NW][16];
    __shared__ float acc[16];
    const int t = threadIdx.x;
    float cam0[7], mv0[14];
    if (t == 0) {       // issued first: these do not depend on the partial sums
        const float* ci = a.cam_in ? a.cam_in : a.cam;
        const float* mi = a.mv_in ? a.mv_in : a.adam_mv;
#pragma unroll
        for (int e = 0; e < 7; ++e) cam0[e] = ci[e];
#pragma unroll
        for (int e = 0; e < 14; ++e) mv0[e] = a.do_update ? mi[e] : 0.0f;
    }
    if (a.do_update) {
        float v12[16];                       // 12 ray moments + the 4 terms of the iteration's loss row (a.loss_part)
#pragma unroll
        for (int q = 0; q < 16; ++q) v12[q] = 0.0f;
        for (int b = t; b < a.n_part; b += 64 * NW) {
#pragma unroll
            for (int q = 0; q < 12; ++q) v12[q] += a.pose_part[(size_t)b * 12 + q];
        }
        if (a.loss_part) {
            for (int b = t; b < a.n_loss_part; b += 64 * NW) {
                const float4 v = *reinterpret_cast<const float4*>(a.loss_part + (size_t)b * 4);
                v12[12] += v.x; v12[13] += v.y; v12[14] += v.z; v12[15] += v.w;
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v12[q] += __shfl_xor(v12[q], o);
        }
        if (lk_lane() == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) s_w[t >> 6][q] = v12[q];
        }
        __syncthreads();
        if (t < 16) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += s_w[w][t];
            acc[t] = s;
            if (write && a.log_row && a.loss_part && t >= 12) a.log_row[t - 12] = s;
        }
        __syncthreads();
        if (t == 0) {
            const float qr = cam0[0], qi = cam0[1], qj = cam0[2], qk = cam0[3];
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
            const float ds = -s * s;
            float gc[7];
            gc[0] = ds * qr * gp + s * dPr; gc[1] = ds * qi * gp + s * dPi; gc[2] = ds * qj * gp + s * dPj; gc[3] = ds * qk * gp + s * dPk;
            gc[4] = acc[9]; gc[5] = acc[10]; gc[6] = acc[11];
#pragma unroll
            for (int e = 0; e < 7; ++e) {        // torch.optim.Adam, group T: elements 4..6, group q: 0..3 (as k_adam)
                if (write && a.hist_pre) a.hist_pre[e] = cam0[e];
                if (write) a.g_cam[e] = gc[e];
                const float m = mv0[e] * a.beta1 + (1.0f - a.beta1) * gc[e];
                const float v = mv0[7 + e] * a.beta2 + (1.0f - a.beta2) * (gc[e] * gc[e]);
                const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
                if (write) { a.adam_mv[e] = m; a.adam_mv[7 + e] = v; }
                const float p = cam0[e] - (e < 4 ? a.step_q : a.step_T) * (m / denom);
                if (write) a.cam[e] = p;
                cam0[e] = p;
                if (write && a.hist_post) a.hist_post[e] = p;
            }
        }
    }
    if (t == 0) {
#pragma unroll
        for (int e = 0; e < 7; ++e) s_cam[e] = cam0[e];        // the stepped pose (do_update = 0: the pose as it is)
    }
    __syncthreads();
}

// ... and the rays of that pose for the next iteration's pixels (get_rays_from_uv): k_track_final, one workgroup, a few microseconds
template <int NW>
__device__ __forceinline__ void lk_track_final_body(const LkTrackFinalArgs& a) {
    __shared__ float s_cam[7];
    const int t = threadIdx.x;
    // Issued first, used last: the pixels of the next batch do not depend on the partial sums - their round trip runs beside the
    // reduction instead of behind it (the launch is a chain of dependent round trips, nothing else)
    constexpr int RPT = 8;                                 // rays per thread: R <= 8192 in the fused loop (LK_MASK_REG_MAX)
    float npi[RPT], npj[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = t + 64 * NW * q;
        const bool on = a.rays_o != nullptr && r < a.R;
        npi[q] = on ? a.next_pix_i[r] : 0.0f; npj[q] = on ? a.next_pix_j[r] : 0.0f;
    }
    lk_track_pose_step<NW>(a, s_cam, true);
    if (!a.rays_o) return;
    float Rm[9];
    lp_quat_rot(s_cam, Rm);
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = t + 64 * NW * q;
        if (r < a.R) {
            const float d0 = (npi[q] - a.cx) / a.fx, d1 = -(npj[q] - a.cy) / a.fy, d2 = -1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a.rays_d[3 * r + c] = (d0 * Rm[3 * c] + d1 * Rm[3 * c + 1]) + d2 * Rm[3 * c + 2];
                a.rays_o[3 * r + c] = s_cam[4 + c];
            }
        }
    }
}
