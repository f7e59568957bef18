// Map maintenance around the hot loop, device-resident:
//   frustum row selection   Mapper.get_mask_from_c2w (src/Mapper.py:165-217)   -> the row index Adam optimises in place
//   point insertion         NeuralPointCloud.add_neural_points, geometry part (src/neural_point.py:1557-1631)
// Both are HBM-bound index work: one thread per point / ray, coalesced 12-byte position reads, the radius test on
// the same uniform grid as the renderer's kNN, stable ballot + prefix-sum compaction (k_compact).
#include "lk_common.h"
#include "lk_knn_dev.h"

#include <math.h>

int lk_launch_compact(const uint8_t* mask, int n, int32_t* out_index, int32_t* out_count, hipStream_t st);
int lk_launch_compact_mb(const uint8_t* mask, int n, int32_t* out_index, int32_t* out_count, int32_t* block_scratch, hipStream_t st);

// ------------------------------------------------------------------ frustum rows
struct LkFrustumArgs {
    const float* pos; int N;
    double w[12];                          // w2c rows 0..2 (float32 values)
    const float* depth; int H, W;
    double fx, fy, cx, cy;
    int edge;
    float* d_samp; uint8_t* mask; unsigned* dmax_bits;
};

// projection of point i exactly as the reference's float64 numpy code (sums in the order ((a+b)+c)+d)
__device__ __forceinline__ void frustum_project(const LkFrustumArgs& a, int i, float& u, float& v, double& zz) {
    const double x = (double)a.pos[3 * (size_t)i], y = (double)a.pos[3 * (size_t)i + 1], z = (double)a.pos[3 * (size_t)i + 2];
    const double c0 = ((a.w[0] * x + a.w[1] * y) + a.w[2] * z) + a.w[3];
    const double c1 = ((a.w[4] * x + a.w[5] * y) + a.w[6] * z) + a.w[7];
    const double c2 = ((a.w[8] * x + a.w[9] * y) + a.w[10] * z) + a.w[11];
    const double xc = -c0;                                      // cam_cord[:, 0] *= -1
    zz = c2 + 1e-5;
    u = (float)((a.fx * xc + a.cx * c2) / zz);
    v = (float)((a.fy * c1 + a.cy * c2) / zz);
}

// cv2.remap INTER_LINEAR / BORDER_CONSTANT(0) on a float32 image (oracle.remap_linear_zero states the algorithm)
__device__ __forceinline__ float frustum_sample(const float* __restrict__ img, int H, int W, float u, float v) {
    const bool big = !(fabsf(u) <= 1e7f) || !(fabsf(v) <= 1e7f);           // also catches NaN / inf
    const float us = big ? -1e6f : u, vs = big ? -1e6f : v;
    const long long sx = (long long)rint((double)us * 32.0), sy = (long long)rint((double)vs * 32.0);
    const long long ix = sx >> 5, iy = sy >> 5;
    const float ax = (float)(sx & 31) / 32.0f, ay = (float)(sy & 31) / 32.0f;
    auto tap = [&](long long yy, long long xx) -> float {
        return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? img[yy * W + xx] : 0.0f;
    };
    const float w00 = __fmul_rn(1.0f - ay, 1.0f - ax), w01 = __fmul_rn(1.0f - ay, ax), w10 = __fmul_rn(ay, 1.0f - ax), w11 = __fmul_rn(ay, ax);
    float out = __fmul_rn(tap(iy, ix), w00);
    out = __fadd_rn(out, __fmul_rn(tap(iy, ix + 1), w01));
    out = __fadd_rn(out, __fmul_rn(tap(iy + 1, ix), w10));
    out = __fadd_rn(out, __fmul_rn(tap(iy + 1, ix + 1), w11));
    return out;
}

__global__ __launch_bounds__(256) void k_frustum_sample(LkFrustumArgs a) {
    __shared__ unsigned smax[4];
    const int i = blockIdx.x * 256 + (int)threadIdx.x;
    unsigned bits = 0;
    if (i < a.N) {
        float u, v; double zz;
        frustum_project(a, i, u, v, zz);
        const float d = frustum_sample(a.depth, a.H, a.W, u, v);
        a.d_samp[i] = d;
        bits = (d > 0.0f) ? __float_as_uint(d) : 0u;           // sampled depths are >= 0: bit order = value order
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o));
    if (lk_lane() == 0) smax[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(a.dmax_bits, max(max(smax[0], smax[1]), max(smax[2], smax[3])));
}

__global__ __launch_bounds__(256) void k_frustum_mask(LkFrustumArgs a) {
    const int i = blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= a.N) return;
    float u, v; double zz;
    frustum_project(a, i, u, v, zz);
    float d = a.d_samp[i];
    if (d == 0.0f) d = __uint_as_float(*a.dmax_bits);          // depths[zero_mask] = max(sampled depths)
    const float e = (float)a.edge;
    const bool in_img = (u < (float)a.W - e) && (u > e) && (v < (float)a.H - e) && (v > e);
    const bool ok = in_img && (0.0 <= -zz) && (-zz <= (double)__fadd_rn(d, 0.5f));
    a.mask[i] = ok ? 1 : 0;
}

extern "C" int lk_frustum_rows(const float* pos, int32_t N, const float* w2c12_host, const float* depth, int32_t H, int32_t W,
                               float fx, float fy, float cx, float cy, int32_t edge, float* scratch_depth, uint8_t* scratch_mask,
                               uint32_t* scratch_max, int32_t* out_index, int32_t* out_count, void* stream_) {
    LK_REQUIRE(N >= 0 && w2c12_host && out_count, "lk_frustum_rows: bad arguments");
    hipStream_t st = (hipStream_t)stream_;
    if (N == 0) { LK_HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(int32_t), st)); return LK_OK; }
    LK_REQUIRE(pos && depth && H > 0 && W > 0 && scratch_depth && scratch_mask && scratch_max && out_index, "lk_frustum_rows: NULL buffer");
    LkFrustumArgs a;
    a.pos = pos; a.N = N;
    for (int q = 0; q < 12; ++q) a.w[q] = (double)w2c12_host[q];
    a.depth = depth; a.H = H; a.W = W;
    a.fx = (double)fx; a.fy = (double)fy; a.cx = (double)cx; a.cy = (double)cy;
    a.edge = edge; a.d_samp = scratch_depth; a.mask = scratch_mask; a.dmax_bits = scratch_max;
    LK_HIP_TRY(hipMemsetAsync(scratch_max, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_frustum_sample, dim3(lk_cdiv(N, 256)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_frustum_mask, dim3(lk_cdiv(N, 256)), dim3(256), 0, st, a);
    // the sampled-depth scratch is dead after the mask pass: reuse it for the per-block counts of the compaction
    if (N > 16384) lk_launch_compact_mb(scratch_mask, N, out_index, out_count, reinterpret_cast<int32_t*>(scratch_depth), st);
    else lk_launch_compact(scratch_mask, N, out_index, out_count, st);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

// ------------------------------------------------------------------ point insertion
struct LkAddArgs {
    int n;
    const float* rays_o; const float* rays_d; const float* gt_depth; const float* r2_ray; float r2_static;
    const LkGrid* grid; const float4* sorted; const int32_t* cell_start; int have_cloud;
    uint8_t* mask;
    const int32_t* idx; const int32_t* count;
    float near_surface, far_surface; int n_add;
    float* pos_out;
};

// accept ray i iff depth > 0 and no cloud point lies strictly inside the add radius of its surface point
__global__ __launch_bounds__(256) void k_add_test(LkAddArgs a) {
    const int i = blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= a.n) return;
    const float gd = a.gt_depth[i];
    bool ok = gd > 0.0f;
    if (ok && a.have_cloud) {
        const float qx = lk_madd_rn(a.rays_o[3 * i], a.rays_d[3 * i], gd);
        const float qy = lk_madd_rn(a.rays_o[3 * i + 1], a.rays_d[3 * i + 1], gd);
        const float qz = lk_madd_rn(a.rays_o[3 * i + 2], a.rays_d[3 * i + 2], gd);
        const float r2 = a.r2_ray ? a.r2_ray[i] : a.r2_static;
        const LkGrid* G = a.grid;
        const float ox = G->ox, oy = G->oy, oz = G->oz, inv = G->inv_cell;
        const int dx = G->dx, dy = G->dy, dz = G->dz;
        const float r = sqrtf(r2) * 1.0001f + 1e-6f;
        bool any = G->n > 0;
        any = any && !((qx + r - ox) * inv < 0.0f || (qx - r - ox) * inv >= (float)dx);
        any = any && !((qy + r - oy) * inv < 0.0f || (qy - r - oy) * inv >= (float)dy);
        any = any && !((qz + r - oz) * inv < 0.0f || (qz - r - oz) * inv >= (float)dz);
        if (any) {
            const int ix0 = lk_cell_coord(qx - r, ox, inv, dx), ix1 = lk_cell_coord(qx + r, ox, inv, dx);
            const int iy0 = lk_cell_coord(qy - r, oy, inv, dy), iy1 = lk_cell_coord(qy + r, oy, inv, dy);
            const int iz0 = lk_cell_coord(qz - r, oz, inv, dz), iz1 = lk_cell_coord(qz + r, oz, inv, dz);
            for (int iz = iz0; iz <= iz1 && ok; ++iz)
                for (int iy = iy0; iy <= iy1 && ok; ++iy) {
                    const int row = (iz * dy + iy) * dx;
                    const int s = a.cell_start[row + ix0], e = a.cell_start[row + ix1 + 1];
                    for (int t = s; t < e; ++t) {
                        const float4 p = a.sorted[t];
                        if (lk_dist2(qx, qy, qz, p.x, p.y, p.z) < r2) { ok = false; break; }   // neighbor_num = sum(D < r^2)
                    }
                }
        }
    }
    a.mask[i] = ok ? 1 : 0;
}

// accepted ray j -> n_add points at z = near*d*(1-t) + far*d*t, t = linspace(0,1,n_add)  (torch arithmetic, f32)
__global__ __launch_bounds__(256) void k_add_emit(LkAddArgs a) {
    const int j = blockIdx.x * 256 + (int)threadIdx.x;
    if (j >= *a.count) return;
    const int i = a.idx[j];
    const float gd = a.gt_depth[i];
    for (int q = 0; q < a.n_add; ++q) {
        float t = 0.0f;                                            // torch.linspace(0, 1, steps)[q], symmetric evaluation
        if (a.n_add > 1) {
            const float step = 1.0f / (float)(a.n_add - 1);
            t = (q < a.n_add / 2) ? step * (float)q : 1.0f - step * (float)(a.n_add - 1 - q);
        }
        const float z = __fadd_rn(__fmul_rn(__fmul_rn(a.near_surface, gd), __fsub_rn(1.0f, t)),
                                  __fmul_rn(__fmul_rn(a.far_surface, gd), t));
        float* o = a.pos_out + ((size_t)j * a.n_add + q) * 3;
        o[0] = lk_madd_rn(a.rays_o[3 * i], a.rays_d[3 * i], z);
        o[1] = lk_madd_rn(a.rays_o[3 * i + 1], a.rays_d[3 * i + 1], z);
        o[2] = lk_madd_rn(a.rays_o[3 * i + 2], a.rays_d[3 * i + 2], z);
    }
}

extern "C" int lk_add_points(lk_knn_t knn, const float* rays_o, const float* rays_d, const float* gt_depth, int32_t n,
                             float r2_static, const float* r2_per_ray, float near_surface, float far_surface, int32_t n_add,
                             uint8_t* scratch_mask, int32_t* out_ray_index, int32_t* out_count, float* out_points, void* stream_) {
    LK_REQUIRE(n >= 0 && n_add >= 1 && out_count, "lk_add_points: bad arguments");
    hipStream_t st = (hipStream_t)stream_;
    if (n == 0) { LK_HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(int32_t), st)); return LK_OK; }
    LK_REQUIRE(rays_o && rays_d && gt_depth && scratch_mask && out_ray_index && out_points, "lk_add_points: NULL buffer");
    LkAddArgs a;
    a.n = n; a.rays_o = rays_o; a.rays_d = rays_d; a.gt_depth = gt_depth; a.r2_ray = r2_per_ray; a.r2_static = r2_static;
    a.have_cloud = (knn != nullptr && knn->n > 0) ? 1 : 0;
    a.grid = knn ? knn->grid : nullptr; a.sorted = knn ? knn->sorted : nullptr; a.cell_start = knn ? knn->cell_start : nullptr;
    a.mask = scratch_mask; a.idx = out_ray_index; a.count = out_count;
    a.near_surface = near_surface; a.far_surface = far_surface; a.n_add = n_add; a.pos_out = out_points;
    hipLaunchKernelGGL(k_add_test, dim3(lk_cdiv(n, 256)), dim3(256), 0, st, a);
    lk_launch_compact(scratch_mask, n, out_ray_index, out_count, st);
    hipLaunchKernelGGL(k_add_emit, dim3(lk_cdiv(n, 256)), dim3(256), 0, st, a);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

// ------------------------------------------------------------------ per-frame image pre-passes
// Colour-gradient magnitude and the dynamic radius maps (Tracker.py:243-258, Mapper.py:854-872), in float64 like the
// reference (its images are float64): grey = rgb2gray, Sobel with reflected borders, magnitude, radius = piecewise-
// linear map of the clipped magnitude; the magnitude and the SQUARED radii are stored as float32 (the ABI's r^2
// convention).
struct LkRadiusArgs {
    const float* color; int H, W;
    double thr, add_max, slope_add, query_max, slope_query;
    float* grad_mag; float* r2_add; float* r2_query;
};

__device__ __forceinline__ double rm_grey(const float* __restrict__ c, int H, int W, int y, int x) {
    y = y < 0 ? 0 : (y >= H ? H - 1 : y);                       // 'reflect' with a radius-1 kernel = repeat the border pixel
    x = x < 0 ? 0 : (x >= W ? W - 1 : x);
    const float* p = c + ((size_t)y * W + x) * 3;
    return ((double)p[0] * 0.2125 + (double)p[1] * 0.7154) + (double)p[2] * 0.0721;
}

__global__ __launch_bounds__(256) void k_radius_maps(LkRadiusArgs a) {
    const int i = blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= a.H * a.W) return;
    const int y = i / a.W, x = i - y * a.W;
    double g[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) g[dy][dx] = rm_grey(a.color, a.H, a.W, y + dy - 1, x + dx - 1);
    const double gy = 0.25 * (g[2][0] - g[0][0]) + 0.5 * (g[2][1] - g[0][1]) + 0.25 * (g[2][2] - g[0][2]);
    const double gx = 0.25 * (g[0][2] - g[0][0]) + 0.5 * (g[1][2] - g[1][0]) + 0.25 * (g[2][2] - g[2][0]);
    const double mag = sqrt(gx * gx + gy * gy);
    a.grad_mag[i] = (float)mag;
    const double xc = fmin(fmax(mag, 0.0), a.thr);
    const double ra = (xc <= 0.01) ? a.add_max : a.slope_add * (xc - 0.01) + a.add_max;
    const double rq = (xc <= 0.01) ? a.query_max : a.slope_query * (xc - 0.01) + a.query_max;
    if (a.r2_add) a.r2_add[i] = (float)(ra * ra);
    if (a.r2_query) a.r2_query[i] = (float)(rq * rq);
}

extern "C" int lk_radius_maps(const float* color, int32_t H, int32_t W, double color_grad_threshold, double radius_add_max,
                              double radius_add_min, double radius_query_ratio, float* grad_mag, float* r2_add, float* r2_query,
                              void* stream_) {
    LK_REQUIRE(H > 0 && W > 0 && color && grad_mag, "lk_radius_maps: bad arguments");
    LK_REQUIRE(color_grad_threshold > 0.01, "lk_radius_maps: color_grad_threshold must exceed 0.01");
    LkRadiusArgs a;
    a.color = color; a.H = H; a.W = W;
    a.thr = color_grad_threshold;                       // config scalars are python floats (float64) in the reference
    a.add_max = radius_add_max;
    a.slope_add = (radius_add_min - radius_add_max) / (a.thr - 0.01);
    a.query_max = radius_query_ratio * radius_add_max;
    a.slope_query = (radius_query_ratio * radius_add_min - a.query_max) / (a.thr - 0.01);
    a.grad_mag = grad_mag; a.r2_add = r2_add; a.r2_query = r2_query;
    hipLaunchKernelGGL(k_radius_maps, dim3(lk_cdiv((int64_t)H * W, 256)), dim3(256), 0, (hipStream_t)stream_, a);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

// Pixel pool with the largest colour gradients (get_selected_index_with_grad, common.py:198-234): the K largest
// magnitudes of the whole image (K-th value by a 4-pass radix select on the bit patterns; ties at the cut in ascending
// pixel order), then only pixels inside the window with a valid depth are kept.  One 1024-thread workgroup (the
// image is 1.2 MB and stays in L2; this runs once per frame).
__global__ __launch_bounds__(1024) void k_top_grad(const float* __restrict__ grad, int n, int K, int W, int H0, int H1, int W0, int W1,
                                                   const float* __restrict__ depth, int depth_limit,
                                                   int32_t* __restrict__ out_index, int32_t* __restrict__ out_count) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_rank, s_above;
    __shared__ int wa[16], wb[16];
    __shared__ int s_ties_seen, s_out;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) { s_prefix = 0; s_rank = (unsigned)(n - K); s_above = 0; s_ties_seen = 0; s_out = 0; }   // ascending rank of the K-th largest
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (t < 256) hist[t] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned himask = (shift == 24) ? 0u : (0xffffffffu << (shift + 8));
        for (int c0 = 0; c0 < n; c0 += 1024) {
            const int i = c0 + t;
            const unsigned u = (i < n) ? __float_as_uint(grad[i]) : 0u;
            const bool on = (i < n) && (u & himask) == prefix;
            const unsigned digit = (u >> shift) & 255u;
            if (shift >= 16) {          // few distinct high bytes: one LDS add per (wave, byte)
                unsigned long long pending = __ballot(on);
                while (pending) {
                    const int leader = __ffsll((long long)pending) - 1;
                    const unsigned dl = __shfl(digit, leader);
                    const unsigned long long same = __ballot(on && digit == dl);
                    if (lane == leader) atomicAdd(&hist[dl], (unsigned)__popcll(same));
                    pending &= ~same;
                }
            } else if (on) {
                atomicAdd(&hist[digit], 1u);
            }
        }
        __syncthreads();
        if (t < 64) {
            const unsigned rank = s_rank;
            const unsigned h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2], h3 = hist[4 * t + 3];
            const unsigned tot = h0 + h1 + h2 + h3;
            unsigned incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned nbv = __shfl_up(incl, o);
                if (t >= o) incl += nbv;
            }
            const unsigned excl = incl - tot;
            if (rank >= excl && rank < incl) {
                unsigned r = rank - excl, b = 4 * t;
                if (r >= h0) { r -= h0; ++b; if (r >= h1) { r -= h1; ++b; if (r >= h2) { r -= h2; ++b; } } }
                s_rank = r;
                s_prefix = prefix | (b << shift);
            }
        }
        __syncthreads();
    }
    const unsigned kth = s_prefix;                                  // bit pattern of the K-th largest magnitude
    // how many are strictly above it -> how many ties at the cut belong to the pool
    {
        unsigned cnt = 0;
        for (int i = t; i < n; i += 1024) cnt += (__float_as_uint(grad[i]) > kth) ? 1u : 0u;
        atomicAdd(&s_above, cnt);
    }
    __syncthreads();
    const int ties_needed = K - (int)s_above;
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + t;
        const unsigned u = (i < n) ? __float_as_uint(grad[i]) : 0u;
        const bool tie = (i < n) && u == kth;
        const unsigned long long bt = __ballot(tie);
        if (lane == 0) wa[w] = __popcll(bt);
        __syncthreads();
        int tie_rank = s_ties_seen + __popcll(bt & ((1ull << lane) - 1ull));
        for (int q = 0; q < w; ++q) tie_rank += wa[q];
        bool member = (i < n) && (u > kth || (tie && tie_rank < ties_needed));
        if (member) {
            const int y = i / W, x = i - y * W;
            member = y >= H0 && y < H1 && x >= W0 && x < W1;
            if (member && depth) { const float d = depth[i]; member = d > 0.0f && (!depth_limit || d <= 5.0f); }
        }
        const unsigned long long bm = __ballot(member);
        if (lane == 0) wb[w] = __popcll(bm);
        __syncthreads();
        int off = s_out + __popcll(bm & ((1ull << lane) - 1ull));
        for (int q = 0; q < w; ++q) off += wb[q];
        if (member) out_index[off] = i;
        __syncthreads();
        if (t == 0) {
            int ta = 0, tb = 0;
            for (int q = 0; q < 16; ++q) { ta += wa[q]; tb += wb[q]; }
            s_ties_seen += ta; s_out += tb;
        }
        __syncthreads();
    }
    if (t == 0) *out_count = s_out;
}

// ---- the same pool with MANY workgroups (images above 16 384 pixels).  One workgroup walks a 640 x 480 image six times and pays an LDS atomic
// per pixel and pass on ONE compute unit: 0.9 ms per tracked frame of the TUM / ScanNet configs.  Here every pass is a launch over
// LK_TG_WGS slices: per-slice LDS histograms added into a global 256-bin histogram per pass (every later launch re-derives the
// selected prefix from the histograms itself - 64 lanes, a 256-entry scan per pass), per-slice counts of "above the K-th value" /
// "equal to it", the membership mask with the ties ranked in pixel order across the slices, and the library's many-block compaction
// (ascending indices).  Same pool, bit for bit.
#define LK_TG_WGS 256
struct TgScratch { unsigned* hist; unsigned* above; unsigned* ties; uint8_t* mask; int32_t* blocks; size_t mask_cap; };
// one scratch set PER DEVICE (a process that drives several devices must not hand one device's pointers to another's kernels); calls on
// ONE device share it: lk_top_grad_pixels is one-call-at-a-time per device (include/loopy_hip.h)
static TgScratch& tg_scratch() {
    static TgScratch s[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    return s[dev];
}

// prefix (bit pattern so far) and rank (ascending rank inside the candidates) after `passes` radix passes; called by wave 0, all 64 lanes
__device__ __forceinline__ void tg_select(const unsigned* __restrict__ hist_all, int passes, unsigned rank0, unsigned& prefix, unsigned& rank) {
    const int t = lk_lane();
    prefix = 0u; rank = rank0;
    for (int p = 0; p < passes; ++p) {
        const int shift = 24 - 8 * p;
        const unsigned* hist = hist_all + 256 * p;
        const unsigned h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2], h3 = hist[4 * t + 3];
        const unsigned tot = h0 + h1 + h2 + h3;
        unsigned incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned nbv = __shfl_up(incl, o);
            if (t >= o) incl += nbv;
        }
        const unsigned excl = incl - tot;
        unsigned r = 0, b = 0;
        const bool mine = rank >= excl && rank < incl;
        if (mine) {
            r = rank - excl; b = 4 * t;
            if (r >= h0) { r -= h0; ++b; if (r >= h1) { r -= h1; ++b; if (r >= h2) { r -= h2; ++b; } } }
        }
        const unsigned long long who = __ballot(mine);
        const int src = __ffsll((long long)who) - 1;
        rank = __shfl(r, src);
        prefix |= __shfl(b, src) << shift;
    }
}
__global__ __launch_bounds__(256) void k_tg_hist(const float* __restrict__ grad, int n, int K, int pass, int slice, unsigned* __restrict__ hist_all) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix;
    const int t = threadIdx.x;
    hist[t] = 0;
    if (t < 64) {
        unsigned prefix, rank;
        tg_select(hist_all, pass, (unsigned)(n - K), prefix, rank);
        if (t == 0) s_prefix = prefix;
    }
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const unsigned prefix = s_prefix, himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    const int i0 = blockIdx.x * slice, i1 = min(n, i0 + slice);
    for (int i = i0 + t; i < i1; i += 256) {
        const unsigned u = __float_as_uint(grad[i]);
        if ((u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (hist[t]) atomicAdd(hist_all + 256 * pass + t, hist[t]);
}
__global__ __launch_bounds__(256) void k_tg_count(const float* __restrict__ grad, int n, int K, int slice, const unsigned* __restrict__ hist_all,
                                                  unsigned* __restrict__ above, unsigned* __restrict__ ties) {
    __shared__ unsigned s_kth, s_a, s_t;
    const int t = threadIdx.x;
    if (t < 64) {
        unsigned prefix, rank;
        tg_select(hist_all, 4, (unsigned)(n - K), prefix, rank);
        if (t == 0) { s_kth = prefix; s_a = 0; s_t = 0; }
    }
    __syncthreads();
    const unsigned kth = s_kth;
    const int i0 = blockIdx.x * slice, i1 = min(n, i0 + slice);
    unsigned ca = 0, ct = 0;
    for (int i = i0 + t; i < i1; i += 256) {
        const unsigned u = __float_as_uint(grad[i]);
        ca += u > kth ? 1u : 0u; ct += u == kth ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ca += __shfl_xor(ca, o); ct += __shfl_xor(ct, o); }
    if (lk_lane() == 0) { atomicAdd(&s_a, ca); atomicAdd(&s_t, ct); }
    __syncthreads();
    if (t == 0) { above[blockIdx.x] = s_a; ties[blockIdx.x] = s_t; }
}
__global__ __launch_bounds__(256) void k_tg_mask(const float* __restrict__ grad, int n, int K, int W, int H0, int H1, int W0, int W1,
                                                 const float* __restrict__ depth, int depth_limit, int slice, const unsigned* __restrict__ hist_all,
                                                 const unsigned* __restrict__ above, const unsigned* __restrict__ ties, uint8_t* __restrict__ mask) {
    __shared__ unsigned s_kth, s_above, s_before;
    __shared__ int wa[4];
    __shared__ int s_seen;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t < 64) {
        unsigned prefix, rank;
        tg_select(hist_all, 4, (unsigned)(n - K), prefix, rank);
        // pixels above the cut in the whole image; ties in the slices before this one
        unsigned a = 0, b = 0;
        for (int q = t; q < (int)gridDim.x; q += 64) { a += above[q]; if (q < (int)blockIdx.x) b += ties[q]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (t == 0) { s_kth = prefix; s_above = a; s_before = b; s_seen = 0; }
    }
    __syncthreads();
    const unsigned kth = s_kth;
    const int ties_needed = K - (int)s_above;
    const int i0 = blockIdx.x * slice, i1 = min(n, i0 + slice);
    for (int c0 = i0; c0 < i1; c0 += 256) {             // `slice` is a multiple of 256: the chunks keep pixel order
        const int i = c0 + t;
        const unsigned u = (i < i1) ? __float_as_uint(grad[i]) : 0u;
        const bool tie = (i < i1) && u == kth;
        const unsigned long long bt = __ballot(tie);
        if (lane == 0) wa[w] = __popcll(bt);
        __syncthreads();
        int tie_rank = (int)s_before + s_seen + __popcll(bt & ((1ull << lane) - 1ull));
        for (int q = 0; q < w; ++q) tie_rank += wa[q];
        bool member = (i < i1) && (u > kth || (tie && tie_rank < ties_needed));
        if (member) {
            const int y = i / W, x = i - y * W;
            member = y >= H0 && y < H1 && x >= W0 && x < W1;
            if (member && depth) { const float d = depth[i]; member = d > 0.0f && (!depth_limit || d <= 5.0f); }
        }
        if (i < i1) mask[i] = member ? 1 : 0;
        __syncthreads();
        if (t == 0) s_seen += wa[0] + wa[1] + wa[2] + wa[3];
        __syncthreads();
    }
}

extern "C" int lk_top_grad_pixels(const float* grad_mag, int32_t H, int32_t W, int32_t K, int32_t H0, int32_t H1, int32_t W0, int32_t W1,
                                  const float* depth, int32_t depth_limit, int32_t* out_index, int32_t* out_count, void* stream_) {
    LK_REQUIRE(H > 0 && W > 0 && grad_mag && out_index && out_count && K >= 0, "lk_top_grad_pixels: bad arguments");
    const int n = H * W;
    if (K > n) K = n;
    hipStream_t st = (hipStream_t)stream_;
    if (K == 0) { LK_HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(int32_t), st)); return LK_OK; }
    if (n <= 16384) {
        hipLaunchKernelGGL(k_top_grad, dim3(1), dim3(1024), 0, st, grad_mag, n, (int)K, (int)W, (int)H0, (int)H1, (int)W0, (int)W1,
                           depth, (int)depth_limit, out_index, out_count);
        LK_LAUNCH_CHECK();
        return LK_OK;
    }
    // library-owned scratch (grown on demand; like the library's streams it is shared state: one call at a time)
    TgScratch& ts = tg_scratch();
    if (!ts.hist) {
        LK_HIP_TRY(hipMalloc((void**)&ts.hist, sizeof(unsigned) * (4 * 256 + 2 * LK_TG_WGS)));
        ts.above = ts.hist + 4 * 256; ts.ties = ts.above + LK_TG_WGS;
    }
    if (ts.mask_cap < (size_t)n) {
        LK_HIP_TRY(hipStreamSynchronize(st));           // (first frame of a resolution only) nobody may still read the old buffers
        if (ts.mask) (void)hipFree(ts.mask);
        if (ts.blocks) (void)hipFree(ts.blocks);
        ts.mask = nullptr; ts.blocks = nullptr; ts.mask_cap = 0;
        LK_HIP_TRY(hipMalloc((void**)&ts.mask, (size_t)n + 256));
        LK_HIP_TRY(hipMalloc((void**)&ts.blocks, sizeof(int32_t) * ((size_t)lk_cdiv(n, 256) + 2)));
        ts.mask_cap = (size_t)n;
    }
    const int slice = lk_cdiv(lk_cdiv(n, LK_TG_WGS), 256) * 256;
    const int G = lk_cdiv(n, slice);
    LK_HIP_TRY(hipMemsetAsync(ts.hist, 0, sizeof(unsigned) * 4 * 256, st));
    for (int pass = 0; pass < 4; ++pass) hipLaunchKernelGGL(k_tg_hist, dim3(G), dim3(256), 0, st, grad_mag, n, (int)K, pass, slice, ts.hist);
    hipLaunchKernelGGL(k_tg_count, dim3(G), dim3(256), 0, st, grad_mag, n, (int)K, slice, ts.hist, ts.above, ts.ties);
    hipLaunchKernelGGL(k_tg_mask, dim3(G), dim3(256), 0, st, grad_mag, n, (int)K, (int)W, (int)H0, (int)H1, (int)W0, (int)W1, depth, (int)depth_limit,
                       slice, ts.hist, ts.above, ts.ties, ts.mask);
    lk_launch_compact_mb(ts.mask, n, out_index, out_count, ts.blocks, st);
    LK_LAUNCH_CHECK();
    return LK_OK;
}
