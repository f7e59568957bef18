// Per-frame optimisation loops as single C-ABI calls (include/loopy_hip.h: lk_track_frame, lk_map_frame).
//   reference: Tracker.run loop body + optimize_cam_in_batch (src/Tracker.py:102-197, 313-401),
//              Mapper.optimize_map's joint iterations (src/Mapper.py:576-735).
// The host enqueues a whole frame's launches from C++ (no interpreter between kernels), and for tracking batches the small
// steps between the five heavy kernels are fused into three one-workgroup kernels:
//   k_track_prep    pose copy + pixel gather + rays of the current pose + inside mask        (was 4 launches)
//   k_track_loss    alpha composite + tracker loss + composite backward                        (was 3 launches)
//   k_track_update  d p -> d rays -> d pose, Adam on the 7 pose parameters                     (was 3 launches)
// (a launch of a trivial kernel costs ~5 us on the GPU's front end however little it does: 16 -> 9 launches per iteration).
#include "lk_common.h"
#include "lk_kernels.h"
#include "lk_composite_dev.h"
#include "lk_mask_dev.h"

#include <math.h>
#include <string.h>

using namespace lkw;

#define LK_TRACK_FUSED_MAX_R LK_MASK_REG_MAX          // rays a single workgroup keeps in registers (8 per thread)

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ float lp_block_sum_1024(float v, float* sh /*[16]*/) {
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
__device__ __forceinline__ float lp_sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
__device__ __forceinline__ void lp_quat_rot(const float* __restrict__ cam, float (&Rm)[9]) {       // common.py:301-324
    const float qr = cam[0], qi = cam[1], qj = cam[2], qk = cam[3];
    const float s = 2.0f / (qr * qr + qi * qi + qj * qj + qk * qk);
    Rm[0] = 1.0f - s * (qj * qj + qk * qk); Rm[1] = s * (qi * qj - qk * qr); Rm[2] = s * (qi * qk + qj * qr);
    Rm[3] = s * (qi * qj + qk * qr); Rm[4] = 1.0f - s * (qi * qi + qk * qk); Rm[5] = s * (qj * qk - qi * qr);
    Rm[6] = s * (qi * qk - qj * qr); Rm[7] = s * (qj * qk + qi * qr); Rm[8] = 1.0f - s * (qi * qi + qj * qj);
}

// ------------------------------------------------------------------ k_track_prep
struct LkTrackPrepArgs {
    const float* cam; float* hist_row;                  // hist_row NULL: no candidate recorded here
    const float* depth_img; const float* color_img; const float* r2_map; const int32_t* rnd;
    int R, W, H0, W0, w;
    float fx, fy, cx, cy;
    float* rays_o; float* rays_d; float* gt_depth; float* gt_color; float* pix_i; float* pix_j; float* r2_ray; float* thr;
};
// get_samples (common.py:237-259) + get_rays_from_uv of the CURRENT pose (common.py:104-120, 327-343) + inside mask
// (Tracker.py:153-160: rejected rays become absent, gt_depth = 0) for R <= 8192 rays in one workgroup
__global__ __launch_bounds__(1024) void k_track_prep(LkTrackPrepArgs a) {
    __shared__ LkMaskShared S;
    constexpr int VPT = LK_MASK_VPT;
    const int t = threadIdx.x;
    if (a.hist_row && t < 7) a.hist_row[t] = a.cam[t];
    float Rm[9];
    lp_quat_rot(a.cam, Rm);
    const float tx = a.cam[4], ty = a.cam[5], tz = a.cam[6];
    unsigned u[VPT];
    unsigned mycnt = 0, mymax = 0;
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        const int r = t + 1024 * q;
        u[q] = 0u;
        if (r < a.R) {
            const int px = a.rnd[r];
            const int i = a.W0 + px % a.w, j = a.H0 + px / a.w;
            const size_t pix = (size_t)j * a.W + i;
            const float d = a.depth_img[pix];
            u[q] = (d > 0.0f) ? __float_as_uint(d) : 0u;
            if (u[q]) { ++mycnt; mymax = max(mymax, u[q]); }
            a.gt_color[3 * r] = a.color_img[3 * pix]; a.gt_color[3 * r + 1] = a.color_img[3 * pix + 1]; a.gt_color[3 * r + 2] = a.color_img[3 * pix + 2];
            if (a.r2_ray) a.r2_ray[r] = a.r2_map ? a.r2_map[pix] : 0.0f;
            a.pix_i[r] = (float)i; a.pix_j[r] = (float)j;
            const float d0 = ((float)i - a.cx) / a.fx, d1 = -((float)j - a.cy) / a.fy, d2 = -1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) a.rays_d[3 * r + c] = (d0 * Rm[3 * c] + d1 * Rm[3 * c + 1]) + d2 * Rm[3 * c + 2];
            a.rays_o[3 * r] = tx; a.rays_o[3 * r + 1] = ty; a.rays_o[3 * r + 2] = tz;
        }
    }
    bool any;
    const float thr = lk_inside_thr<true>(u, nullptr, a.R, mycnt, mymax, S, &any);
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        const int r = t + 1024 * q;
        if (r < a.R) {
            const float d = __uint_as_float(u[q]);
            a.gt_depth[r] = (any && u[q] && d <= thr) ? d : 0.0f;
        }
    }
    if (t == 0) *a.thr = any ? thr : 0.0f;
}

// ------------------------------------------------------------------ k_track_loss
struct LkTrackLossArgs {
    int R, S, min_nn;
    float coef, w_color;
    int use_color;
    const float* raw; const float* z; const int32_t* nbr_count; const float* gt_depth; const float* gt_color;
    float* depth; float* var; float* color; uint8_t* valid_ray;
    float* d_depth; float* d_color; float* d_raw; float* out4;
};
// raw2outputs_nerf_color (common.py:382-422) + tracker loss (Tracker.py:169-191) + the composite's backward for the
// resulting d depth / d colour, R <= 8192 rays in one workgroup (the loss mask needs the batch mean of the residual)
__global__ __launch_bounds__(1024) void k_track_loss(LkTrackLossArgs a) {
    __shared__ float sh[16];
    constexpr int VPT = LK_TRACK_FUSED_MAX_R / 1024;
    const int t = threadIdx.x;
    float tv[VPT], dep[VPT], var[VPT], c0[VPT], c1[VPT], c2[VPT];
    float tsum = 0.0f, csum = 0.0f;
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        const int r = t + 1024 * q;
        tv[q] = dep[q] = var[q] = c0[q] = c1[q] = c2[q] = 0.0f;
        if (r < a.R) {
            const float gd = a.gt_depth[r];
            const LkRayOut o = lk_composite_ray(a.raw, a.z, a.nbr_count, r, a.S, a.min_nn, a.coef, gd);
            dep[q] = o.depth; var[q] = o.var; c0[q] = o.c0; c1[q] = o.c1; c2[q] = o.c2;
            a.depth[r] = o.depth; a.var[r] = o.var;
            a.color[3 * r] = o.c0; a.color[3 * r + 1] = o.c1; a.color[3 * r + 2] = o.c2;
            a.valid_ray[r] = o.valid ? 1 : 0;
            const bool present = gd > 0.0f;                // absent rays take no part in the mean (filtered before the render)
            tv[q] = present ? fabsf(gd - o.depth) / sqrtf(o.var + 1e-10f) : 0.0f;
            tsum += tv[q];
            csum += present ? 1.0f : 0.0f;
        }
    }
    tsum = lp_block_sum_1024(tsum, sh);
    csum = lp_block_sum_1024(csum, sh);
    const float thr = 10.0f * (tsum / fmaxf(csum, 1.0f));
    float geo = 0.0f, col = 0.0f, cnt = 0.0f;
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        const int r = t + 1024 * q;
        if (r < a.R) {
            const float d = dep[q], v = var[q], g = a.gt_depth[r], tt = tv[q];
            const bool m = (tt < thr) && (g > 0.0f) && !(d != d) && !(v != v);
            float dd = 0.0f, dc0 = 0.0f, dc1 = 0.0f, dc2 = 0.0f;
            if (m) {
                geo += fminf(fmaxf(tt, 0.0f), 1e3f);
                if (tt <= 1e3f) dd = lp_sgn(d - g) / sqrtf(v + 1e-10f);
                cnt += 1.0f;
                const float e0 = c0[q] - a.gt_color[3 * r], e1 = c1[q] - a.gt_color[3 * r + 1], e2 = c2[q] - a.gt_color[3 * r + 2];
                col += fabsf(e0) + fabsf(e1) + fabsf(e2);
                if (a.use_color) { dc0 = a.w_color * lp_sgn(e0); dc1 = a.w_color * lp_sgn(e1); dc2 = a.w_color * lp_sgn(e2); }
            }
            a.d_depth[r] = dd;
            a.d_color[3 * r] = dc0; a.d_color[3 * r + 1] = dc1; a.d_color[3 * r + 2] = dc2;
            lk_composite_bwd_ray(a.raw, a.z, a.nbr_count, r, a.S, a.min_nn, a.coef, g, dd, 0.0f, dc0, dc1, dc2, a.d_raw);
        }
    }
    geo = lp_block_sum_1024(geo, sh);
    col = lp_block_sum_1024(col, sh);
    cnt = lp_block_sum_1024(cnt, sh);
    if (t == 0) {
        a.out4[0] = geo + (a.use_color ? a.w_color * col : 0.0f);
        a.out4[1] = geo; a.out4[2] = col; a.out4[3] = cnt;
    }
}

// ------------------------------------------------------------------ k_track_update
struct LkTrackUpdateArgs {
    int R, S;
    float fx, fy, cx, cy;
    const float* z; const float* dp_total; const float* pix_i; const float* pix_j;
    float* g_rays_o; float* g_rays_d;
    float* cam; float* g_cam; float* adam_mv; float* hist_row;     // hist_row NULL: no candidate recorded here
    float step_T, step_q, bc2_sqrt, beta1, beta2, eps;            // lr / bias_correction1 per group, sqrt(bias_correction2)
};
// d p -> d rays (sum over the samples of a ray) -> d pose (k_pose_bwd's formulas) -> Adam on (T | q), in one workgroup
__global__ __launch_bounds__(1024) void k_track_update(LkTrackUpdateArgs a) {
    __shared__ float sh[16];
    __shared__ float acc[12];
    const int t = threadIdx.x;
    float G[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gT[3] = {0.f, 0.f, 0.f};
    for (int r = t; r < a.R; r += 1024) {
        float o[3] = {0.f, 0.f, 0.f}, dd[3] = {0.f, 0.f, 0.f};
        for (int s = 0; s < a.S; ++s) {
            const int p = r * a.S + s;
            const float4 g = *reinterpret_cast<const float4*>(a.dp_total + (size_t)p * 4);
            const float z = a.z[p];
            o[0] += g.x; o[1] += g.y; o[2] += g.z;
            dd[0] = fmaf(g.x, z, dd[0]); dd[1] = fmaf(g.y, z, dd[1]); dd[2] = fmaf(g.z, z, dd[2]);
        }
        const float dir[3] = {(a.pix_i[r] - a.cx) / a.fx, -(a.pix_j[r] - a.cy) / a.fy, -1.0f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a.g_rays_o[3 * r + c] = o[c]; a.g_rays_d[3 * r + c] = dd[c];
            G[3 * c] = fmaf(dd[c], dir[0], G[3 * c]); G[3 * c + 1] = fmaf(dd[c], dir[1], G[3 * c + 1]); G[3 * c + 2] = fmaf(dd[c], dir[2], G[3 * c + 2]);
            gT[c] += o[c];
        }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) { const float s = lp_block_sum_1024(G[q], sh); if (t == 0) acc[q] = s; }
#pragma unroll
    for (int q = 0; q < 3; ++q) { const float s = lp_block_sum_1024(gT[q], sh); if (t == 0) acc[9 + q] = s; }
    __syncthreads();
    if (t == 0) {
        float* cam = a.cam;
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
        const float ds = -s * s;
        float gc[7];
        gc[0] = ds * qr * gp + s * dPr; gc[1] = ds * qi * gp + s * dPi; gc[2] = ds * qj * gp + s * dPj; gc[3] = ds * qk * gp + s * dPk;
        gc[4] = acc[9]; gc[5] = acc[10]; gc[6] = acc[11];
        // torch.optim.Adam on the 7 parameters (group T: elements 4..6, group q: 0..3), as k_adam
#pragma unroll
        for (int e = 0; e < 7; ++e) {
            a.g_cam[e] = gc[e];
            const float m = a.adam_mv[e] * a.beta1 + (1.0f - a.beta1) * gc[e];
            const float v = a.adam_mv[7 + e] * a.beta2 + (1.0f - a.beta2) * (gc[e] * gc[e]);
            const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
            a.adam_mv[e] = m; a.adam_mv[7 + e] = v;
            const float p = cam[e] - (e < 4 ? a.step_q : a.step_T) * (m / denom);
            cam[e] = p;
            if (a.hist_row) a.hist_row[e] = p;
        }
    }
}

// ------------------------------------------------------------------ lk_track_frame
namespace {
void adam_scalars(float lr, int step, float beta1, float beta2, float* step_size, float* bc2_sqrt) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    *step_size = (float)((double)lr / bc1);
    *bc2_sqrt = (float)sqrt(bc2);
}
}  // namespace

extern "C" int lk_track_frame(const lk_track_desc* d, void* stream_) {
    LK_REQUIRE(d != nullptr, "lk_track_frame: NULL descriptor");
    LK_REQUIRE(d->iters >= 0 && d->render.R >= 0 && d->w > 0, "lk_track_frame: bad sizes");
    if (d->iters == 0 || d->render.R == 0) return LK_OK;
    LK_REQUIRE(d->depth_img && d->color_img && d->rnd && d->gt_color && d->pix_i && d->pix_j && d->thr && d->scratch_u32 && d->loss_scratch &&
               d->cam7 && d->g_cam7 && d->adam_mv && d->hist && d->log, "lk_track_frame: NULL buffer");
    LK_REQUIRE(d->render.g_rays_o && d->render.g_rays_d && d->render.d_depth && d->render.d_color && d->render.bwd_scratch && d->render.act,
               "lk_track_frame: the render descriptor needs g_rays_o/g_rays_d, d_depth/d_color, bwd_scratch and act");
    hipStream_t st = (hipStream_t)stream_;
    lk_render_desc rd = d->render;
    rd.flags = (d->render.flags & LK_FLAG_REL_POS) | LK_FLAG_STAGE_COLOR | LK_FLAG_TRACKER | LK_FLAG_SAVE_ACT | LK_FLAG_GRAD_RAYS | LK_FLAG_ZERO_ABSENT;
    rd.stats_chunk = rd.R > 0 ? rd.R : 1;
    rd.g_geo_feats = nullptr; rd.g_col_feats = nullptr; rd.g_weights = nullptr;
    const int R = rd.R, S = rd.S;
    const bool fused = R <= LK_TRACK_FUSED_MAX_R;
    const LkBwdOffsets off = lk_bwd_offsets((int64_t)R * S, rd.flags);
    const float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;
    LK_HIP_TRY(hipMemsetAsync(d->adam_mv, 0, 14 * sizeof(float), st));
    for (int it = 0; it < d->iters; ++it) {
        const int32_t* rnd = d->rnd + (size_t)it * R;
        float* hist_row = d->hist + (size_t)it * 7;
        float* log_row = d->log + (size_t)it * 4;
        // ---- batch assembly
        if (fused) {
            LkTrackPrepArgs pa;
            pa.cam = d->cam7; pa.hist_row = d->hist_post ? nullptr : hist_row;
            pa.depth_img = d->depth_img; pa.color_img = d->color_img; pa.r2_map = d->r2_map; pa.rnd = rnd;
            pa.R = R; pa.W = d->W; pa.H0 = d->H0; pa.W0 = d->W0; pa.w = d->w;
            pa.fx = d->fx; pa.fy = d->fy; pa.cx = d->cx; pa.cy = d->cy;
            pa.rays_o = const_cast<float*>(rd.rays_o); pa.rays_d = const_cast<float*>(rd.rays_d); pa.gt_depth = const_cast<float*>(rd.gt_depth);
            pa.gt_color = d->gt_color; pa.pix_i = d->pix_i; pa.pix_j = d->pix_j; pa.r2_ray = const_cast<float*>(rd.r2_ray); pa.thr = d->thr;
            hipLaunchKernelGGL(k_track_prep, dim3(1), dim3(1024), 0, st, pa);
        } else {
            static const float eye[12] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f};
            (void)eye;
            if (!d->hist_post) LK_HIP_TRY(hipMemcpyAsync(hist_row, d->cam7, 7 * sizeof(float), hipMemcpyDeviceToDevice, st));
            // the image gathers do not depend on the pose: gather with the pose slot pointing at the (unused) previous rays,
            // then overwrite the rays with those of the current pose
            int rc = lk_gather_rays(d->depth_img, d->color_img, d->cam7 /* any 12 floats: rays are recomputed below */, 0, d->r2_map, nullptr, rnd, R,
                                    d->H, d->W, d->H0, d->W0, d->w, d->fx, d->fy, d->cx, d->cy, const_cast<float*>(rd.rays_o),
                                    const_cast<float*>(rd.rays_d), const_cast<float*>(rd.gt_depth), d->gt_color, d->pix_i, d->pix_j,
                                    const_cast<float*>(rd.r2_ray), st);
            if (rc != LK_OK) return rc;
            rc = lk_rays_from_pose(d->cam7, d->pix_i, d->pix_j, R, d->fx, d->fy, d->cx, d->cy, const_cast<float*>(rd.rays_o), const_cast<float*>(rd.rays_d), st);
            if (rc != LK_OK) return rc;
            rc = lk_inside_mask(rd.gt_depth, R, nullptr, const_cast<float*>(rd.gt_depth), d->thr, d->scratch_u32, st);
            if (rc != LK_OK) return rc;
        }
        // ---- forward, loss, backward
        int rc = lk_render_fwd_impl(&rd, st, fused ? LK_SKIP_COMPOSITE : 0);
        if (rc != LK_OK) return rc;
        if (fused) {
            LkTrackLossArgs la;
            la.R = R; la.S = S; la.min_nn = rd.min_nn; la.coef = rd.coef; la.w_color = d->w_color; la.use_color = d->use_color;
            la.raw = rd.raw; la.z = rd.z; la.nbr_count = rd.nbr_count; la.gt_depth = rd.gt_depth; la.gt_color = d->gt_color;
            la.depth = rd.depth; la.var = rd.var; la.color = rd.color; la.valid_ray = rd.valid_ray;
            la.d_depth = const_cast<float*>(rd.d_depth); la.d_color = const_cast<float*>(rd.d_color);
            la.d_raw = rd.bwd_scratch + off.d_raw; la.out4 = log_row;
            hipLaunchKernelGGL(k_track_loss, dim3(1), dim3(1024), 0, st, la);
        } else {
            rc = lk_loss_tracker(R, rd.depth, rd.var, rd.color, rd.gt_depth, d->gt_color, d->w_color, d->use_color,
                                 const_cast<float*>(rd.d_depth), const_cast<float*>(rd.d_color), log_row, d->loss_scratch, st);
            if (rc != LK_OK) return rc;
        }
        rc = lk_render_bwd_impl(&rd, st, fused ? (LK_SKIP_COMPOSITE_BWD | LK_SKIP_RAYS_BWD) : 0);
        if (rc != LK_OK) return rc;
        // ---- pose gradient + Adam (Tracker.py:317-352: group T at cam_lr, group q at 0.2 cam_lr when separate_LR)
        float step_T, step_q, bc2s;
        adam_scalars(d->lr_T, it + 1, beta1, beta2, &step_T, &bc2s);
        adam_scalars(d->lr_q, it + 1, beta1, beta2, &step_q, &bc2s);
        if (fused) {
            LkTrackUpdateArgs ua;
            ua.R = R; ua.S = S; ua.fx = d->fx; ua.fy = d->fy; ua.cx = d->cx; ua.cy = d->cy;
            ua.z = rd.z; ua.dp_total = rd.bwd_scratch + off.dp_total; ua.pix_i = d->pix_i; ua.pix_j = d->pix_j;
            ua.g_rays_o = rd.g_rays_o; ua.g_rays_d = rd.g_rays_d;
            ua.cam = d->cam7; ua.g_cam = d->g_cam7; ua.adam_mv = d->adam_mv; ua.hist_row = d->hist_post ? hist_row : nullptr;
            ua.step_T = step_T; ua.step_q = step_q; ua.bc2_sqrt = bc2s; ua.beta1 = beta1; ua.beta2 = beta2; ua.eps = eps;
            hipLaunchKernelGGL(k_track_update, dim3(1), dim3(1024), 0, st, ua);
        } else {
            rc = lk_pose_bwd(d->cam7, d->pix_i, d->pix_j, R, d->fx, d->fy, d->cx, d->cy, rd.g_rays_o, rd.g_rays_d, d->g_cam7, st);
            if (rc != LK_OK) return rc;
            lk_adam_seg seg[2];
            memset(seg, 0, sizeof(seg));
            seg[0].p = d->cam7 + 4; seg[0].g = d->g_cam7 + 4; seg[0].m = d->adam_mv + 4; seg[0].v = d->adam_mv + 11; seg[0].n = 3; seg[0].lr = d->lr_T; seg[0].step = it + 1;
            seg[1].p = d->cam7; seg[1].g = d->g_cam7; seg[1].m = d->adam_mv; seg[1].v = d->adam_mv + 7; seg[1].n = 4; seg[1].lr = d->lr_q; seg[1].step = it + 1;
            rc = lk_adam_step(seg, 2, beta1, beta2, eps, st);
            if (rc != LK_OK) return rc;
            if (d->hist_post) LK_HIP_TRY(hipMemcpyAsync(hist_row, d->cam7, 7 * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
    }
    LK_LAUNCH_CHECK();
    return LK_OK;
}

// ------------------------------------------------------------------ lk_map_frame
extern "C" int lk_map_frame(const lk_map_desc* d, int32_t it_begin, int32_t it_end, int32_t phases, void* stream_) {
    LK_REQUIRE(d != nullptr, "lk_map_frame: NULL descriptor");
    LK_REQUIRE(it_begin >= 0 && it_end <= d->iters && it_begin <= it_end && (phases & 3), "lk_map_frame: bad iteration range / phases");
    LK_REQUIRE(d->n_geo_dec >= 0 && d->n_geo_dec <= LK_MAX_SPANS && d->n_col_dec >= 0 && d->n_col_dec <= LK_MAX_SPANS, "lk_map_frame: too many spans");
    if (it_begin == it_end || d->render.R == 0) return LK_OK;
    LK_REQUIRE(d->depth_stack && d->color_stack && d->c2w_stack && d->rnd && d->gt_color && d->thr && d->scratch_u32 && d->log,
               "lk_map_frame: NULL batch buffer");
    LK_REQUIRE(d->weights_rw && d->weights_frag_rw && d->geo_feats_rw && d->col_feats_rw && d->adam_rows && d->adam_dec && d->n_rows >= 0,
               "lk_map_frame: NULL optimiser buffer");
    LK_REQUIRE(d->render.g_geo_feats && d->render.g_col_feats && d->render.g_weights && d->render.d_depth && d->render.d_color &&
               d->render.bwd_scratch && d->render.act, "lk_map_frame: the render descriptor needs gradient buffers, d_depth/d_color, bwd_scratch and act");
    hipStream_t st = (hipStream_t)stream_;
    const int R = d->render.R;
    const float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;
    const int64_t nb = lk_weight_blob_floats();
    const int64_t nrow = d->n_rows * LK_C;
    for (int it = it_begin; it < it_end; ++it) {
        const bool color = it >= d->n_geo_iters;
        lk_render_desc rd = d->render;
        rd.flags = (d->render.flags & (LK_FLAG_REL_POS | LK_FLAG_UNIT_LOSS_GRADS)) | (color ? LK_FLAG_STAGE_COLOR : 0) | LK_FLAG_SAVE_ACT |
                   LK_FLAG_GRAD_FEATS | LK_FLAG_GRAD_WEIGHTS | LK_FLAG_ZERO_ABSENT | LK_FLAG_MAPPER_LOSS;
        rd.stats_chunk = R;
        rd.loss_gt_color = d->gt_color; rd.loss_w_color = d->w_color; rd.loss_out4 = d->log + (size_t)it * 4;
        if (phases & 1) {
            int rc = lk_gather_rays(d->depth_stack, d->color_stack, d->c2w_stack, d->c2w_stride, d->r2_map_stack, d->frame_id, d->rnd + (size_t)it * R, R,
                                    d->H, d->W, d->H0, d->W0, d->w, d->fx, d->fy, d->cx, d->cy, const_cast<float*>(rd.rays_o),
                                    const_cast<float*>(rd.rays_d), const_cast<float*>(rd.gt_depth), d->gt_color, nullptr, nullptr,
                                    const_cast<float*>(rd.r2_ray), st);
            if (rc != LK_OK) return rc;
            rc = lk_inside_mask(rd.gt_depth, R, nullptr, const_cast<float*>(rd.gt_depth), d->thr, d->scratch_u32, st);
            if (rc != LK_OK) return rc;
            // the loss gradient is final when the composite kernel has written it: its backward rides in the same launch
            rc = lk_render_fwd_impl(&rd, st, LK_FUSE_COMPOSITE_BWD);
            if (rc != LK_OK) return rc;
            rc = lk_render_bwd_impl(&rd, st, LK_SKIP_COMPOSITE_BWD);
            if (rc != LK_OK) return rc;
        }
        if (phases & 2) {
            // torch.optim.Adam: parameters without a gradient are skipped and keep their own step count - the colour decoder and
            // the colour rows first step in the first 'color' iteration (Mapper.py:588-607, 722-724)
            const float* lr = d->lr[color ? 1 : 0];
            lk_adam_seg seg[LK_ADAM_MAX_SEG];
            memset(seg, 0, sizeof(seg));
            int ns = 0;
            auto dec_seg = [&](const lk_blob_span& sp, int step) {
                lk_adam_seg& s = seg[ns++];
                s.p = d->weights_rw + sp.offset; s.g = rd.g_weights + sp.offset; s.m = d->adam_dec + sp.offset; s.v = d->adam_dec + nb + sp.offset;
                s.n = sp.n; s.lr = lr[0]; s.step = step; s.zero_grad = 1;
            };
            for (int k = 0; k < d->n_geo_dec; ++k) dec_seg(d->geo_dec[k], it + 1);
            if (color) for (int k = 0; k < d->n_col_dec; ++k) dec_seg(d->col_dec[k], it - d->n_geo_iters + 1);
            LK_REQUIRE(ns + 2 <= LK_ADAM_MAX_SEG, "lk_map_frame: too many optimiser segments");
            {
                lk_adam_seg& s = seg[ns++];
                s.p = d->geo_feats_rw; s.g = rd.g_geo_feats; s.m = d->adam_rows; s.v = d->adam_rows + nrow; s.n = nrow; s.lr = lr[1]; s.step = it + 1;
                s.row_index = d->rows; s.row_len = d->rows ? LK_C : 1; s.zero_grad = 1;
            }
            if (color) {
                lk_adam_seg& s = seg[ns++];
                s.p = d->col_feats_rw; s.g = rd.g_col_feats; s.m = d->adam_rows + 2 * nrow; s.v = d->adam_rows + 3 * nrow; s.n = nrow; s.lr = lr[2];
                s.step = it - d->n_geo_iters + 1; s.row_index = d->rows; s.row_len = d->rows ? LK_C : 1; s.zero_grad = 1;
            }
            int rc = lk_adam_step(seg, ns, beta1, beta2, eps, st);
            if (rc != LK_OK) return rc;
            if (color) {        // the matrix fragments are copies of the colour-decoder matrices, which only move in this stage
                rc = lk_weights_repack(d->weights_rw, d->weights_frag_rw, st);
                if (rc != LK_OK) return rc;
            }
        }
    }
    LK_LAUNCH_CHECK();
    return LK_OK;
}
