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
