// Depth-guided sampling + radius-kNN + inverse-distance feature interpolation (HBM-bound half of
// the render forward), and the per-ray alpha composite.
//   reference: Renderer.render_batch_ray z-sampling (src/utils/Renderer.py:98-176),
//              NeuralPointCloud.find_neighbors_faiss (src/neural_point.py:1659-1708),
//              MLP_*.get_feature_at_pos (src/conv_onet/models/decoder.py:180-231, 431-492),
//              raw2outputs_nerf_color (src/common.py:382-422), Renderer.py:184-200.
#include "lk_common.h"
#include "lk_weights_dev.h"
#define LK_SEARCH_WGS 512
LK_CHAIN_DEFINE(sample)
#define LK_KNN_STAMP(I) LK_STAMP(I)          // (probe build: the search's phases inside k_sample_interp_pose)
#define LK_KNN_STAMPW(I) LK_STAMPW(I)
#include "lk_knn_dev.h"
#include "lk_kernels.h"
#include "lk_track_dev.h"
#include "lk_composite_dev.h"

// torch.linspace(start, end, steps)[i] (symmetric evaluation, aten RangeFactories)
__device__ __forceinline__ float lk_linspace(float start, float end, int steps, int i) {
    if (steps <= 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

// far_bb per group of `chunk` rays (Renderer.py:102-121): far = clamp(min(5*mean, 1.2*max), 0, 1.2*max)
__global__ __launch_bounds__(256) void k_depth_stats(const float* __restrict__ gt, int R, int chunk,
                                                     float* __restrict__ far_out) {
    __shared__ float ssum[4], smax[4];
    const int b = blockIdx.x;
    const int lo = b * chunk, hi = min(R, lo + chunk);
    float sum = 0.0f, mx = -LK_FLT_MAX;
    for (int i = lo + (int)threadIdx.x; i < hi; i += 256) { const float v = gt[i]; sum += v; mx = fmaxf(mx, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if (lk_lane() == 0) { ssum[threadIdx.x >> 6] = sum; smax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sum = ssum[0] + ssum[1] + ssum[2] + ssum[3];
        mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        const float mean = sum / (float)(hi - lo);
        const float mx12 = mx * 1.2f;
        const float far_bb = fminf(5.0f * mean, mx12);
        far_out[b] = (mx > 0.0f) ? fminf(fmaxf(far_bb, 0.0f), mx12) : far_bb;
    }
}

// Eight lanes per sample point (32 points per 256-thread block): z and p (every lane, redundantly),
// cooperative exact top-8 within the radius, normalised weights, then lane `sub` gathers its float4
// of each of the 8 neighbour rows (8 lanes x 16 B = one 128-B feature row per load instruction).
// T = lanes per sample point (8: large batches; 16: training batches, halves each lane's serial candidate chain)
// MODE 0: search + interpolation.  The search depends on the rays and the positions only - not on what a mapping call optimises -
// so lk_map_frame runs it for ALL its iterations ahead of time on a third stream (MODE 1: lists only) and every iteration starts
// with MODE 2 (lists given: interpolation of the current features, the backward's row count).
// POSE (tracking loop, MODE 0): the rays are not read but derived from the pose s_cam[7] (LDS: the pose step of the iteration before ran as
// this launch's prologue, k_sample_interp_pose) and the batch's pixels, with k_track_final's arithmetic; the group of a ray's first
// sample also stores them for the kernels behind this one.
// (POSE) what a sample group reads that does NOT depend on the pose: fetched by k_sample_interp_pose in front of the pose step, whose three
// barriers the compiler moves no load across - behind it the ray's reading and pixel were one more cold round trip of every group's chain
struct LkPoseEarly { float gt, pix_i, pix_j, r2; LkGrid grid; };
template <int T, int MODE, bool POSE = false>
__device__ __forceinline__ void sample_interp_block(const LkSampleArgs& a, int block, const LkTrackFinalArgs* f = nullptr, const float* s_cam = nullptr,
                                                    const LkPoseEarly* early = nullptr) {
    const int sub = (int)threadIdx.x & (T - 1);
    const int p_raw = block * (256 / T) + (int)threadIdx.x / T;
    const bool live = p_raw < a.P;
    const int pidx = live ? p_raw : a.P - 1;           // dead groups shadow the last point, never store
    const int r = pidx / a.S, s = pidx - r * a.S;
    float d[LK_K], w[LK_K];
    int id[LK_K];
    int count = 0;
    float z = 0.0f;
    if (MODE == 2) {
        const int4 i0 = *reinterpret_cast<const int4*>(a.nbr_idx + (size_t)pidx * LK_K);
        const int4 i1 = *reinterpret_cast<const int4*>(a.nbr_idx + (size_t)pidx * LK_K + 4);
        const float4 w0 = *reinterpret_cast<const float4*>(a.nbr_w + (size_t)pidx * LK_K);
        const float4 w1 = *reinterpret_cast<const float4*>(a.nbr_w + (size_t)pidx * LK_K + 4);
        id[0] = i0.x; id[1] = i0.y; id[2] = i0.z; id[3] = i0.w; id[4] = i1.x; id[5] = i1.y; id[6] = i1.z; id[7] = i1.w;
        w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w; w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
        count = a.nbr_count[pidx];
    } else {
    const float gt = POSE ? early->gt : a.gt_depth[r];
    if (gt > 0.0f) {
        const float t = lk_linspace(0.0f, 1.0f, a.S, s);
        z = __fadd_rn(__fmul_rn(__fmul_rn(a.near_surface, gt), __fsub_rn(1.0f, t)),
                      __fmul_rn(__fmul_rn(a.far_surface, gt), t));
    } else if (a.flags & LK_FLAG_Z_GIVEN) {
        z = a.z[pidx];                                  // placed by the caller (sample_near_pcl, Renderer.py:152-160)
    } else {
        const float far = a.far_stats ? a.far_stats[r / a.stats_chunk] : 0.0f;
        z = lk_linspace(a.near_end, far, a.S, s);
    }
    float o3[3], d3[3];
    if (POSE) {
        float Rm[9];
        lp_quat_rot(s_cam, Rm);
        const float d0 = (early->pix_i - f->cx) / f->fx, d1 = -(early->pix_j - f->cy) / f->fy, d2 = -1.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d3[c] = (d0 * Rm[3 * c] + d1 * Rm[3 * c + 1]) + d2 * Rm[3 * c + 2];
            o3[c] = s_cam[4 + c];
        }
        if (live && s == 0 && sub == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { f->rays_d[3 * r + c] = d3[c]; f->rays_o[3 * r + c] = o3[c]; }
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) { o3[c] = a.rays_o[3 * r + c]; d3[c] = a.rays_d[3 * r + c]; }
    }
    const float qx = lk_madd_rn(o3[0], d3[0], z);
    const float qy = lk_madd_rn(o3[1], d3[1], z);
    const float qz = lk_madd_rn(o3[2], d3[2], z);
    const float r2 = POSE ? early->r2 : (a.r2_ray ? a.r2_ray[r] : a.r2_static);
    if (POSE) LK_STAMPW(2);                          // the ray's reading, pixel and radius have arrived: the query point is known
    lk_knn_scan_coop<T>(POSE ? &early->grid : a.grid, a.sorted, a.cell_start, qx, qy, qz, r2, sub, d, id);
    if (POSE) LK_STAMP(7);
    // w = 1/(D+1e-10), zero outside the radius, L1-normalised (decoder.py:210-220)
    float wsum = 0.0f;
#pragma unroll
    for (int j = 0; j < LK_K; ++j) {
        const bool in = id[j] >= 0 && d[j] <= r2;
        w[j] = in ? 1.0f / (d[j] + 1e-10f) : 0.0f;
        wsum += w[j];
        count += (id[j] >= 0 && d[j] < r2) ? 1 : 0;
    }
    const float inv = 1.0f / fmaxf(wsum, 1e-12f);
#pragma unroll
    for (int j = 0; j < LK_K; ++j) w[j] = w[j] * inv;
    }
    if (!live) return;
    // lane `sub` publishes neighbour slot `sub`
    {
        float wj = w[0];
        int ij = id[0];
#pragma unroll
        for (int j = 1; j < LK_K; ++j) { wj = (sub == j) ? w[j] : wj; ij = (sub == j) ? id[j] : ij; }
        if (MODE != 2) {
            if (POSE) LK_STAMP(8);
            if (sub < LK_K) {
                a.nbr_idx[(size_t)pidx * LK_K + sub] = ij;
                a.nbr_w[(size_t)pidx * LK_K + sub] = wj;
            }
            if (sub == 0) { a.nbr_count[pidx] = count; a.z[pidx] = z; }
        }
        if (MODE == 1) {      // lists only; the rows of every iteration of the chunk are counted per point on the way (k_seg_count's test)
            if (a.seg_cnt && sub < LK_K) {
                const int y = a.seg_P ? pidx / a.seg_P : 0;
                const bool skipped = a.seg_live && r - y * (a.seg_P / a.S) >= a.seg_live[y];
                int rk = -1;
                if (!skipped && ij >= 0 && wj != 0.0f && count >= a.min_nn && (!a.row_mask || a.row_mask[ij])) {
                    const int key = a.seg_key ? a.seg_key[ij] : ij;
                    if (key >= 0) rk = atomicAdd(a.seg_cnt + (size_t)y * a.seg_cnt_stride + key, 1);
                }
                a.seg_rank[(size_t)pidx * LK_K + sub] = rk;
            }
            return;
        }
        // first pass of the backward's counting sort of the rows by point (k_seg_count, lk_bwd2.hip), while the indices are here
        if (a.seg_cnt && sub < LK_K) {
            int rk = -1;
            if (ij >= 0 && wj != 0.0f && count >= a.min_nn && (!a.row_mask || a.row_mask[ij])) rk = atomicAdd(a.seg_cnt + ij, 1);
            a.seg_rank[(size_t)pidx * LK_K + sub] = rk;
        }
    }
    const bool do_col = (a.flags & LK_FLAG_STAGE_COLOR) && !(a.flags & LK_FLAG_REL_POS);
    // gather: with 8 lanes per point every lane serves one float4 of BOTH tables; with 16 lanes the low 8 lanes
    // serve the geometry row, the high 8 the colour row
    const int f4 = sub & 7;
    const bool lane_geo = (T == 8) || sub < 8;
    const bool lane_col = do_col && ((T == 8) || sub >= 8);
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ac = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool f16 = (a.flags & LK_FLAG_FEATS_F16) != 0;
    if (count >= a.min_nn) {
#pragma unroll
        for (int j = 0; j < LK_K; ++j) {
            if (w[j] != 0.0f) {
                const float wj = w[j];
                if (lane_geo) {
                    const float4 g = lk_feat4(a.geo_feats, f16, (size_t)id[j] * LK_C + f4 * 4);
                    ag.x = fmaf(wj, g.x, ag.x); ag.y = fmaf(wj, g.y, ag.y); ag.z = fmaf(wj, g.z, ag.z); ag.w = fmaf(wj, g.w, ag.w);
                }
                if (lane_col) {
                    const float4 c = lk_feat4(a.col_feats, f16, (size_t)id[j] * LK_C + f4 * 4);
                    ac.x = fmaf(wj, c.x, ac.x); ac.y = fmaf(wj, c.y, ac.y); ac.z = fmaf(wj, c.z, ac.z); ac.w = fmaf(wj, c.w, ac.w);
                }
            }
        }
    } else {            // no usable neighbourhood: the shared noise vector (decoder.py:228-229)
        if (a.noise_geo) ag = *reinterpret_cast<const float4*>(a.noise_geo + f4 * 4);
        if (a.noise_col) ac = *reinterpret_cast<const float4*>(a.noise_col + f4 * 4);
    }
    if (POSE) LK_STAMPW(9);                          // the feature rows have arrived and are summed
    if (lane_geo) *reinterpret_cast<float4*>(a.c_geo + (size_t)pidx * LK_C + f4 * 4) = ag;
    if (lane_col) *reinterpret_cast<float4*>(a.c_col + (size_t)pidx * LK_C + f4 * 4) = ac;
    // rel-pos colour features come from k_relpos_fwd, which does not visit the rays behind the live prefix: give their samples a defined
    // (finite) feature here - the noise vector of unsupported samples - so that what the decoder computes for them stays finite
    if (a.live_rays && (a.flags & LK_FLAG_STAGE_COLOR) && (a.flags & LK_FLAG_REL_POS) && r >= *a.live_rays && lane_geo)
        *reinterpret_cast<float4*>(a.c_col + (size_t)pidx * LK_C + f4 * 4) =
            a.noise_col ? *reinterpret_cast<const float4*>(a.noise_col + f4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// MODE 1 (the look-ahead search of lk_map_frame, beside the iterations of the previous chunk) runs as a BOUNDED grid that walks its
// sample groups: a chunk is 150 000 samples = 18 750 waves, and launched as such it takes every wave slot of the chip for ~110 us -
// the iteration running beside it took 170 us instead of 78.  With at most LK_SEARCH_WGS workgroups (two per compute unit) the
// search leaves three quarters of the slots to the loop's kernels, which are small ('geometry': 782 waves per launch).
// Tracking loop: the pose step of the iteration BEFORE as the prologue of this iteration's search - every workgroup reduces the ray
// moments and steps the pose for itself (the same arithmetic in the same order: the same pose everywhere), workgroup 0 also stores
// pose, moments, gradient and log row (into the OTHER pose buffer: the others are still reading this one).  One launch (8-11 us of
// dependent latency) less per tracking iteration.
template <int T>
__global__ __launch_bounds__(256) void k_sample_interp_pose(LkSampleArgs a, LkTrackFinalArgs f) {
    __shared__ float s_cam[7];
    LK_STAMP(0);
    LkPoseEarly e;
    {       // (the group's ray, as sample_interp_block derives it)
        const int p_raw = (int)blockIdx.x * (256 / T) + (int)threadIdx.x / T;
        const int r = (p_raw < a.P ? p_raw : a.P - 1) / a.S;
        e.gt = a.gt_depth[r]; e.pix_i = f.next_pix_i[r]; e.pix_j = f.next_pix_j[r];
        e.r2 = a.r2_ray ? a.r2_ray[r] : a.r2_static;
        e.grid = *a.grid;
    }
    lk_track_pose_step<4>(f, s_cam, blockIdx.x == 0);
    LK_STAMP(1);                                     // the stepped pose is in LDS
    sample_interp_block<T, 0, true>(a, (int)blockIdx.x, &f, s_cam, &e);
    LK_STAMPW(10);
}
template <int T, int MODE>
__global__ __launch_bounds__(256) void k_sample_interp(LkSampleArgs a) {
    if (MODE == 1) {
        const int nblk = (a.P + 256 / T - 1) / (256 / T);
        for (int b = (int)blockIdx.x; b < nblk; b += (int)gridDim.x) sample_interp_block<T, MODE>(a, b);
    } else {
        sample_interp_block<T, MODE>(a, (int)blockIdx.x);
    }
}

// The interpolation launch of a mapping iteration with the fragment repack of the iteration BEFORE as a rider (blocks >= rp_block0):
// both depend on that iteration's Adam step and on nothing of each other, and the repack's consumers come later in the stream - one
// dispatch (9 us of kernel plus its gap) less per 'color' iteration.
template <int T>
__global__ __launch_bounds__(256) void k_interp_repack(LkSampleArgs a, FragTable tb) {
    if (a.rp_copy_dst && (int)blockIdx.x >= a.rp_block1) {      // the stepped blob over the master blob (nobody reads the master in this launch)
        const int i = (((int)blockIdx.x - a.rp_block1) * 256 + (int)threadIdx.x) * 4;
        if (i >= a.rp_skip_lo && i < a.rp_skip_hi) return;          // (range boundaries are multiples of 64 floats)
        if (i + 3 < a.rp_copy_n) *reinterpret_cast<float4*>(a.rp_copy_dst + i) = *reinterpret_cast<const float4*>(a.rp_plain + i);
        else for (int k = i; k < a.rp_copy_n; ++k) a.rp_copy_dst[k] = a.rp_plain[k];
        return;
    }
    if ((int)blockIdx.x >= a.rp_block0) {
        const int u = ((int)blockIdx.x - a.rp_block0) * 256 + (int)threadIdx.x;
        if (u < LK_REPACK_UNITS) {
            if (a.rp_m_hi > a.rp_m_lo) {            // all but the matrices [rp_m_lo, rp_m_hi)
                repack_unit(a.rp_plain, reinterpret_cast<u32x4*>(a.rp_frag), tb, u, 0, a.rp_m_lo);
                repack_unit(a.rp_plain, reinterpret_cast<u32x4*>(a.rp_frag), tb, u, a.rp_m_hi, N_FRAG_MATS);
            } else repack_unit(a.rp_plain, reinterpret_cast<u32x4*>(a.rp_frag), tb, u);
        }
        return;
    }
    sample_interp_block<T, 2>(a, (int)blockIdx.x);
}

// One thread per ray: occupancy of unsupported samples := -100, alpha composite, validity.  With a.loss_out the mapper loss
// of the batch (Mapper.py:691-720) rides along: per-ray terms and gradients here, one block sum and four atomics per block
// (a launch at the latency floor leaves every mapping iteration).
__global__ __launch_bounds__(256) void k_composite(LkCompositeArgs a) {
    __shared__ float s_red[3][4];
    const int r = blockIdx.x * 256 + (int)threadIdx.x;
    float l_geo = 0.0f, l_col = 0.0f, l_cnt = 0.0f;
    if (r < a.R) {
        const float gd = a.gt_depth[r];
        const LkRayOut ro = lk_composite_ray(a.raw, a.z, a.nbr_count, r, a.S, a.min_nn, a.coef, a.keep_depth ? 1.0f : gd);
        const float dout = ro.depth, o0 = ro.c0, o1 = ro.c1, o2 = ro.c2;
        const bool valid = ro.valid;
        a.depth[r] = dout;
        a.var[r] = ro.var;
        a.color[3 * r] = o0; a.color[3 * r + 1] = o1; a.color[3 * r + 2] = o2;
        a.valid_ray[r] = valid ? 1 : 0;
        if (a.loss_out) {
            const bool m = (gd > 0.0f) && valid && !(dout != dout);
            float dd = 0.0f, d0 = 0.0f, d1 = 0.0f, d2 = 0.0f;
            if (m) {
                l_geo = fabsf(gd - dout);
                dd = (dout > gd) ? 1.0f : ((dout < gd) ? -1.0f : 0.0f);
                l_cnt = 1.0f;
                if (a.use_color) {
                    const float e0 = o0 - a.gt_color[3 * r], e1 = o1 - a.gt_color[3 * r + 1], e2 = o2 - a.gt_color[3 * r + 2];
                    l_col = fabsf(e0) + fabsf(e1) + fabsf(e2);
                    d0 = a.w_color * ((e0 > 0.0f) ? 1.0f : ((e0 < 0.0f) ? -1.0f : 0.0f));
                    d1 = a.w_color * ((e1 > 0.0f) ? 1.0f : ((e1 < 0.0f) ? -1.0f : 0.0f));
                    d2 = a.w_color * ((e2 > 0.0f) ? 1.0f : ((e2 < 0.0f) ? -1.0f : 0.0f));
                }
            }
            a.d_depth[r] = dd;
            a.d_color[3 * r] = d0; a.d_color[3 * r + 1] = d1; a.d_color[3 * r + 2] = d2;
            if (a.d_raw) lk_composite_bwd_ray(a.raw, a.z, a.nbr_count, r, a.S, a.min_nn, a.coef, gd, dd, 0.0f, d0, d1, d2, a.d_raw);
        }
    }
    if (a.loss_out) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { l_geo += __shfl_xor(l_geo, o); l_col += __shfl_xor(l_col, o); l_cnt += __shfl_xor(l_cnt, o); }
        const int w = (int)threadIdx.x >> 6;
        if (lk_lane() == 0) { s_red[0][w] = l_geo; s_red[1][w] = l_col; s_red[2][w] = l_cnt; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float geo = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
            const float col = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
            const float cnt = (s_red[2][0] + s_red[2][1]) + (s_red[2][2] + s_red[2][3]);
            atomicAdd(a.loss_out + 0, geo + (a.use_color ? a.w_color * col : 0.0f));
            atomicAdd(a.loss_out + 1, geo); atomicAdd(a.loss_out + 2, col); atomicAdd(a.loss_out + 3, cnt);
        }
    }
}

int lk_launch_depth_stats(const float* gt, int R, int chunk, float* far_out, hipStream_t st) {
    LkProfScope prof_(LKK_DEPTH_STATS, st);
    hipLaunchKernelGGL(k_depth_stats, dim3(lk_cdiv(R, chunk)), dim3(256), 0, st, gt, R, chunk, far_out);
    return LK_OK;
}
// samples up to which the search + interpolation launch gives a query 16 lanes (above: 8).  Measured on the 5 000-ray tracker batches of the
// TUM / ScanNet configs (25 000 samples): 8 lanes 33.2-33.9 ms per tracked frame, 16 lanes 34.1-35.2; the 1 500-ray batches (7 500 samples) keep 16
#ifndef LK_T16_MAX_P
#define LK_T16_MAX_P (1 << 14)
#endif
int lk_launch_sample_interp(const LkSampleArgs& a, hipStream_t st, int mode, const LkTrackFinalArgs* pose) {
    if (pose) {         // tracking loop, mode 0
        LkProfScope prof_(LKK_SAMPLE_INTERP, st);
        if (a.P <= LK_T16_MAX_P) hipLaunchKernelGGL((k_sample_interp_pose<16>), dim3(lk_cdiv(a.P, 16)), dim3(256), 0, st, a, *pose);
        else hipLaunchKernelGGL((k_sample_interp_pose<8>), dim3(lk_cdiv(a.P, 32)), dim3(256), 0, st, a, *pose);
        return LK_OK;
    }
    if (mode == 1) {         // lists only, ahead of time: not one of the timed per-iteration launches
        const int cap = LK_SEARCH_WGS;      // measured: 256 / 512 / 1024 / unbounded -> map call 15.4 / 15.0 / 15.4 / 15.3 ms
        auto grid = [&](int per) { const int n = lk_cdiv(a.P, per); return n < cap ? n : cap; };
        if (a.P <= (1 << 16)) hipLaunchKernelGGL((k_sample_interp<16, 1>), dim3(grid(16)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_sample_interp<8, 1>), dim3(grid(32)), dim3(256), 0, st, a);
        return LK_OK;
    }
    LkProfScope prof_(LKK_SAMPLE_INTERP, st);
    if (mode == 2) {
        // with rel-pos colour features only the geometry rows are interpolated here: 8 lanes per point are enough
        const bool two = (a.flags & LK_FLAG_STAGE_COLOR) && !(a.flags & LK_FLAG_REL_POS);
        if (a.rp_plain) {
            LkSampleArgs b = a;
            const bool wide = two && a.P <= (1 << 16);
            b.rp_block0 = lk_cdiv(a.P, wide ? 16 : 32);
            b.rp_block1 = b.rp_block0 + lk_cdiv(LK_REPACK_UNITS, 256);
            const dim3 grid(b.rp_block1 + (b.rp_copy_dst ? lk_cdiv(b.rp_copy_n, 1024) : 0));
            if (wide) hipLaunchKernelGGL((k_interp_repack<16>), grid, dim3(256), 0, st, b, lk_frag_table());
            else hipLaunchKernelGGL((k_interp_repack<8>), grid, dim3(256), 0, st, b, lk_frag_table());
            return LK_OK;
        }
        if (two && a.P <= (1 << 16)) hipLaunchKernelGGL((k_sample_interp<16, 2>), dim3(lk_cdiv(a.P, 16)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_sample_interp<8, 2>), dim3(lk_cdiv(a.P, 32)), dim3(256), 0, st, a);
        return LK_OK;
    }
    if (a.P <= LK_T16_MAX_P) hipLaunchKernelGGL((k_sample_interp<16, 0>), dim3(lk_cdiv(a.P, 16)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_sample_interp<8, 0>), dim3(lk_cdiv(a.P, 32)), dim3(256), 0, st, a);
    return LK_OK;
}
int lk_launch_composite(const LkCompositeArgs& a, hipStream_t st) {
    LkProfScope prof_(LKK_COMPOSITE, st);
    hipLaunchKernelGGL(k_composite, dim3(lk_cdiv(a.R, 256)), dim3(256), 0, st, a);
    return LK_OK;
}
