// Neighbour index: exact uniform-grid radius-kNN over the dynamic neural point cloud.
// Replaces faiss-gpu IndexIVFFlat train/add/search (reference src/neural_point.py:67-72,
// 1623-1627, 1659-1708).  Build = device-only counting sort (AABB reduce -> cell histogram
// -> exclusive scan -> scatter), no host synchronisation, graph-capturable.
#include "lk_common.h"
#include "lk_knn_dev.h"

#include <limits.h>

#define SCAN_ITEMS 1024   // cells scanned per block (256 threads x 4)

__device__ __forceinline__ int enc_f(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float dec_f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void k_grid_reset(LkGrid* g) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int a = 0; a < 3; ++a) { g->min_enc[a] = INT_MAX; g->max_enc[a] = INT_MIN; }
    }
}

__global__ __launch_bounds__(256) void k_aabb(const float* __restrict__ pos, int n, LkGrid* g) {
    float mn0 = LK_FLT_MAX, mn1 = LK_FLT_MAX, mn2 = LK_FLT_MAX;
    float mx0 = -LK_FLT_MAX, mx1 = -LK_FLT_MAX, mx2 = -LK_FLT_MAX;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
        mn0 = fminf(mn0, x); mx0 = fmaxf(mx0, x);
        mn1 = fminf(mn1, y); mx1 = fmaxf(mx1, y);
        mn2 = fminf(mn2, z); mx2 = fmaxf(mx2, z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn0 = fminf(mn0, __shfl_xor(mn0, o)); mx0 = fmaxf(mx0, __shfl_xor(mx0, o));
        mn1 = fminf(mn1, __shfl_xor(mn1, o)); mx1 = fmaxf(mx1, __shfl_xor(mx1, o));
        mn2 = fminf(mn2, __shfl_xor(mn2, o)); mx2 = fmaxf(mx2, __shfl_xor(mx2, o));
    }
    if (lk_lane() == 0) {
        atomicMin(&g->min_enc[0], enc_f(mn0)); atomicMax(&g->max_enc[0], enc_f(mx0));
        atomicMin(&g->min_enc[1], enc_f(mn1)); atomicMax(&g->max_enc[1], enc_f(mx1));
        atomicMin(&g->min_enc[2], enc_f(mn2)); atomicMax(&g->max_enc[2], enc_f(mx2));
    }
}

__global__ void k_grid_finalize(LkGrid* g, float base_cell, long long max_cells, int n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (n <= 0) {       // empty cloud: one empty cell, queries return no neighbours
        g->ox = g->oy = g->oz = 0.0f; g->cell = base_cell; g->inv_cell = 1.0f / base_cell;
        g->dx = g->dy = g->dz = 1; g->ncells = 1; g->n = 0;
        return;
    }
    const float mnx = dec_f(g->min_enc[0]), mny = dec_f(g->min_enc[1]), mnz = dec_f(g->min_enc[2]);
    const float mxx = dec_f(g->max_enc[0]), mxy = dec_f(g->max_enc[1]), mxz = dec_f(g->max_enc[2]);
    float cell = base_cell;
    int dx = 1, dy = 1, dz = 1;
    for (int it = 0; it < 200; ++it) {
        const float fx = floorf((mxx - mnx) / cell) + 1.0f, fy = floorf((mxy - mny) / cell) + 1.0f,
                    fz = floorf((mxz - mnz) / cell) + 1.0f;
        if (fx * fy * fz <= (float)max_cells && fx < 2.0e9f && fy < 2.0e9f && fz < 2.0e9f) {
            dx = (int)fx; dy = (int)fy; dz = (int)fz;
            if ((long long)dx * dy * dz <= max_cells) break;
        }
        cell *= 1.25f;
    }
    g->ox = mnx; g->oy = mny; g->oz = mnz;
    g->cell = cell; g->inv_cell = 1.0f / cell;
    g->dx = dx; g->dy = dy; g->dz = dz;
    g->ncells = dx * dy * dz;
    g->n = n;
}

__global__ __launch_bounds__(256) void k_zero_counts(int32_t* __restrict__ cell_start, const LkGrid* __restrict__ g) {
    const int total = g->ncells + 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) cell_start[i] = 0;
}

__global__ __launch_bounds__(256) void k_count(const float* __restrict__ pos, int n, const LkGrid* __restrict__ g,
                                               int32_t* __restrict__ counts, int32_t* __restrict__ cell_of,
                                               int32_t* __restrict__ rank_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cx = lk_cell_coord(pos[3 * i], g->ox, g->inv_cell, g->dx);
    const int cy = lk_cell_coord(pos[3 * i + 1], g->oy, g->inv_cell, g->dy);
    const int cz = lk_cell_coord(pos[3 * i + 2], g->oz, g->inv_cell, g->dz);
    const int c = (cz * g->dy + cy) * g->dx + cx;
    cell_of[i] = c;
    rank_of[i] = atomicAdd(&counts[c], 1);
}

// in-place exclusive scan of (ncells+1) counts: block-local scan + block totals
// (g == NULL: the length is the host-known `total_host` - lk_launch_scan_i32)
// (out != data: the offsets go to `out` and the counts in `data` are cleared on the way - a counter array that is zero again
// when its offsets exist needs no memset before its next use)
// (gridDim.y > 1: a batch of independent scans, member y at data + y * dstride / block_sums + y * sstride)
__global__ __launch_bounds__(256) void k_scan_block(int32_t* data, int32_t* out, int32_t* __restrict__ block_sums,
                                                    const LkGrid* __restrict__ g, int total_host, int dstride, int sstride) {
    __shared__ int wsum[4];
    data += (size_t)blockIdx.y * dstride; out += (size_t)blockIdx.y * dstride; block_sums += (size_t)blockIdx.y * sstride;
    const int total = g ? g->ncells + 1 : total_host;
    const int base = blockIdx.x * SCAN_ITEMS;
    if (base >= total) return;                       // uniform per block
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int idx = base + t * 4;
    int v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (idx + q < total) ? data[idx + q] : 0;
    const int s = v[0] + v[1] + v[2] + v[3];
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int nb = __shfl_up(incl, o);
        if (lane >= o) incl += nb;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < w; ++q) woff += wsum[q];
    int run = woff + incl - s;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (idx + q < total) {
            out[idx + q] = run;
            if (out != data) data[idx + q] = 0;
        }
        run += v[q];
    }
    if (t == 255) block_sums[blockIdx.x] = woff + incl;
}

// exclusive scan of the block totals (single block, 256 at a time with a running carry)
__global__ __launch_bounds__(256) void k_scan_sums(int32_t* __restrict__ block_sums, const LkGrid* __restrict__ g, int total_host, int sstride) {
    __shared__ int wsum[4];
    __shared__ int carry;
    block_sums += (size_t)blockIdx.x * sstride;              // one workgroup per batch member
    const int nb = ((g ? g->ncells + 1 : total_host) + SCAN_ITEMS - 1) / SCAN_ITEMS;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 256) {
        const int i = base + t;
        const int s = (i < nb) ? block_sums[i] : 0;
        int incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int nbv = __shfl_up(incl, o);
            if (lane >= o) incl += nbv;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int woff = carry;
        for (int q = 0; q < w; ++q) woff += wsum[q];
        if (i < nb) block_sums[i] = woff + incl - s;
        __syncthreads();
        if (t == 255) carry = woff + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_scan_add(int32_t* __restrict__ data, const int32_t* __restrict__ block_sums,
                                                  const LkGrid* __restrict__ g, int total_host, int dstride, int sstride) {
    data += (size_t)blockIdx.y * dstride; block_sums += (size_t)blockIdx.y * sstride;
    const int total = g ? g->ncells + 1 : total_host;
    const int base = blockIdx.x * SCAN_ITEMS;
    if (base >= total) return;
    const int off = block_sums[blockIdx.x];
    const int idx = base + threadIdx.x * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (idx + q < total) data[idx + q] += off;
}

__global__ __launch_bounds__(256) void k_scatter(const float* __restrict__ pos, int n,
                                                 const int32_t* __restrict__ cell_start,
                                                 const int32_t* __restrict__ cell_of, const int32_t* __restrict__ rank_of,
                                                 float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int dst = cell_start[cell_of[i]] + rank_of[i];
    sorted[dst] = make_float4(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], __int_as_float(i));
}

template <int T>
__global__ __launch_bounds__(256) void k_knn_query(const LkGrid* __restrict__ g, const float4* __restrict__ sorted,
                                                   const int32_t* __restrict__ cell_start,
                                                   const float* __restrict__ q, int P, float r2_scalar,
                                                   const float* __restrict__ r2_per_query,
                                                   float* __restrict__ out_d2, int32_t* __restrict__ out_idx,
                                                   int32_t* __restrict__ out_count) {
    const int qi_raw = blockIdx.x * (256 / T) + (int)threadIdx.x / T;
    const int sub = (int)threadIdx.x % T;
    const bool live = qi_raw < P;
    const int i = live ? qi_raw : P - 1;             // dead groups shadow the last query (shuffles stay convergent)
    const float r2 = r2_per_query ? r2_per_query[i] : r2_scalar;
    float d[LK_K];
    int id[LK_K];
    lk_knn_scan_coop<T>(g, sorted, cell_start, q[3 * i], q[3 * i + 1], q[3 * i + 2], r2, sub, d, id);
    if (!live) return;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < LK_K; ++j) {
        cnt += (id[j] >= 0 && d[j] < r2) ? 1 : 0;
        if (sub == (j % T)) {
            out_d2[(size_t)i * LK_K + j] = d[j];
            out_idx[(size_t)i * LK_K + j] = id[j];
        }
    }
    if (sub == 0) out_count[i] = cnt;
}

// ------------------------------------------------------------------ host API
// in-place exclusive scan of `total` int32 counts (block_sums: lk_cdiv(total, 1024) + 256 ints of scratch)
int lk_launch_scan_i32(int32_t* data, int32_t* out, int32_t* block_sums, int total, hipStream_t st, int batch, int dstride, int sstride) {
    const int nb = lk_cdiv(total, SCAN_ITEMS);
    hipLaunchKernelGGL(k_scan_block, dim3(nb, batch), dim3(256), 0, st, data, out, block_sums, (const LkGrid*)nullptr, total, dstride, sstride);
    if (nb > 1) {
        hipLaunchKernelGGL(k_scan_sums, dim3(batch), dim3(256), 0, st, block_sums, (const LkGrid*)nullptr, total, sstride);
        hipLaunchKernelGGL(k_scan_add, dim3(nb, batch), dim3(256), 0, st, out, (const int32_t*)block_sums, (const LkGrid*)nullptr, total, dstride, sstride);
    }
    return LK_OK;
}

extern "C" int lk_knn_create(float cell_size, int64_t capacity_points, int64_t max_cells, lk_knn_t* out) {
    LK_REQUIRE(out != nullptr, "lk_knn_create: out is NULL");
    LK_REQUIRE(cell_size > 0.0f, "lk_knn_create: cell_size must be > 0");
    LK_REQUIRE(capacity_points > 0 && capacity_points < (1ll << 31), "lk_knn_create: capacity out of range");
    if (max_cells <= 0) max_cells = 1ll << 24;
    LK_REQUIRE(max_cells < (1ll << 30), "lk_knn_create: max_cells too large");
    lk_knn_s* h = new lk_knn_s();
    h->cell_size = cell_size;
    h->capacity = capacity_points;
    h->max_cells = max_cells;
    h->n = 0;
    h->n_scan_blocks = lk_cdiv(max_cells + 1, SCAN_ITEMS);
    h->seg_stride = (int)(capacity_points + 1 + SCAN_ITEMS); h->seg_sums_stride = (int)(lk_cdiv(capacity_points + 1, SCAN_ITEMS) + 256);
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc((void**)&h->grid, sizeof(LkGrid));
    if (e == hipSuccess) e = hipMalloc((void**)&h->sorted, sizeof(float4) * (size_t)capacity_points);
    if (e == hipSuccess) e = hipMalloc((void**)&h->cell_start, sizeof(int32_t) * (size_t)(max_cells + 1 + SCAN_ITEMS));
    if (e == hipSuccess) e = hipMalloc((void**)&h->cell_of, sizeof(int32_t) * (size_t)capacity_points);
    if (e == hipSuccess) e = hipMalloc((void**)&h->rank_of, sizeof(int32_t) * (size_t)capacity_points);
    if (e == hipSuccess) e = hipMalloc((void**)&h->block_sums, sizeof(int32_t) * (size_t)(h->n_scan_blocks + 256));
    if (e == hipSuccess) e = hipMalloc((void**)&h->seg_cnt, sizeof(int32_t) * (size_t)h->seg_stride * LK_SEG_BATCH);
    if (e == hipSuccess) e = hipMemset(h->seg_cnt, 0, sizeof(int32_t) * (size_t)h->seg_stride * LK_SEG_BATCH);
    if (e == hipSuccess) e = hipMalloc((void**)&h->seg_off, sizeof(int32_t) * (size_t)h->seg_stride * LK_SEG_BATCH);
    if (e == hipSuccess) e = hipMalloc((void**)&h->seg_sums, sizeof(int32_t) * (size_t)h->seg_sums_stride * LK_SEG_BATCH);
    if (e == hipSuccess) e = hipMalloc((void**)&h->row_rank, sizeof(int32_t) * (size_t)capacity_points);
    if (e == hipSuccess) e = hipMalloc((void**)&h->act_flag, (size_t)capacity_points + 64);
    if (e == hipSuccess) e = hipMemset(h->act_flag, 0, (size_t)capacity_points + 64);
    if (e == hipSuccess) e = hipMemset(h->grid, 0, sizeof(LkGrid));
    if (e != hipSuccess) {
        lk_set_error("lk_knn_create: allocation failed: %s", hipGetErrorString(e));
        lk_knn_destroy(h);
        return LK_ERR_HIP;
    }
    *out = h;
    return LK_OK;
}

extern "C" int lk_knn_destroy(lk_knn_t h) {
    if (!h) return LK_OK;
    if (h->grid) (void)hipFree(h->grid);
    if (h->sorted) (void)hipFree(h->sorted);
    if (h->cell_start) (void)hipFree(h->cell_start);
    if (h->cell_of) (void)hipFree(h->cell_of);
    if (h->rank_of) (void)hipFree(h->rank_of);
    if (h->block_sums) (void)hipFree(h->block_sums);
    if (h->pos_own) (void)hipFree(h->pos_own);
    if (h->seg_cnt) (void)hipFree(h->seg_cnt);
    if (h->seg_off) (void)hipFree(h->seg_off);
    if (h->seg_sums) (void)hipFree(h->seg_sums);
    if (h->act_flag) (void)hipFree(h->act_flag);
    if (h->row_rank) (void)hipFree(h->row_rank);
    delete h;
    return LK_OK;
}

extern "C" int64_t lk_knn_size(lk_knn_t h) { return h ? h->n : -1; }

extern "C" int lk_knn_build(lk_knn_t h, const float* pos, int64_t N, void* stream_) {
    LK_REQUIRE(h != nullptr, "lk_knn_build: NULL handle");
    LK_REQUIRE(N >= 0 && N <= h->capacity, "lk_knn_build: N exceeds the capacity given to lk_knn_create");
    LK_REQUIRE(N == 0 || pos != nullptr, "lk_knn_build: pos is NULL");
    hipStream_t st = (hipStream_t)stream_;
    const int n = (int)N;
    // The per-point row counters of the feature-gradient sort are zero BETWEEN sorts by construction (the scan clears what a sort counted).
    // A sort that never reached its scan - a launch that failed midway, an lk_map_frame sequence abandoned before its look-ahead chunk was
    // consumed - would leave counts behind that the next sort silently adds to: a rebuild of the index is the point where no sort is in
    // flight in the caller's stream order, so the counters of the rows in use are cleared here (a few tens of microseconds at 5 M points).
    if (h->seg_cnt) {
        const int64_t used = (int64_t)(h->n > N ? h->n : N) + 1;
        for (int y = 0; y < LK_SEG_BATCH; ++y)
            LK_HIP_TRY(hipMemsetAsync(h->seg_cnt + (size_t)y * h->seg_stride, 0, sizeof(int32_t) * (size_t)used, st));
    }
    h->n = N;
    hipLaunchKernelGGL(k_grid_reset, dim3(1), dim3(64), 0, st, h->grid);
    if (n > 0) {
        const int nb = lk_cdiv(n, 256);
        hipLaunchKernelGGL(k_aabb, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, st, pos, n, h->grid);
    }
    hipLaunchKernelGGL(k_grid_finalize, dim3(1), dim3(64), 0, st, h->grid, h->cell_size, (long long)h->max_cells, n);
    if (n > 0) {
        hipLaunchKernelGGL(k_zero_counts, dim3(2048), dim3(256), 0, st, h->cell_start, h->grid);
        hipLaunchKernelGGL(k_count, dim3(lk_cdiv(n, 256)), dim3(256), 0, st, pos, n, h->grid, h->cell_start,
                           h->cell_of, h->rank_of);
        hipLaunchKernelGGL(k_scan_block, dim3(h->n_scan_blocks), dim3(256), 0, st, h->cell_start, h->cell_start, h->block_sums, h->grid, 0, 0, 0);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, st, h->block_sums, h->grid, 0, 0);
        hipLaunchKernelGGL(k_scan_add, dim3(h->n_scan_blocks), dim3(256), 0, st, h->cell_start, h->block_sums, h->grid, 0, 0, 0);
        hipLaunchKernelGGL(k_scatter, dim3(lk_cdiv(n, 256)), dim3(256), 0, st, pos, n, h->cell_start, h->cell_of,
                           h->rank_of, h->sorted);
    }
    LK_LAUNCH_CHECK();
    return LK_OK;
}

// un-sort: original-order positions back out of the grid's sorted copy (for lk_knn_append)
__global__ __launch_bounds__(256) void k_unsort(const float4* __restrict__ sorted, int n, float* __restrict__ pos) {
    const int i = blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= n) return;
    const float4 p = sorted[i];
    const int idx = __float_as_int(p.w);
    pos[3 * (size_t)idx] = p.x; pos[3 * (size_t)idx + 1] = p.y; pos[3 * (size_t)idx + 2] = p.z;
}

extern "C" int lk_knn_append(lk_knn_t h, const float* pos_new, int64_t M, void* stream_) {
    LK_REQUIRE(h != nullptr, "lk_knn_append: NULL handle");
    LK_REQUIRE(M >= 0 && h->n + M <= h->capacity, "lk_knn_append: the grown cloud exceeds the capacity given to lk_knn_create");
    if (M == 0) return LK_OK;
    LK_REQUIRE(pos_new != nullptr, "lk_knn_append: pos_new is NULL");
    hipStream_t st = (hipStream_t)stream_;
    if (!h->pos_own) LK_HIP_TRY(hipMalloc((void**)&h->pos_own, sizeof(float) * 3 * (size_t)h->capacity));
    const int n = (int)h->n;
    // the grid keeps no original-order copy: recover it from the sorted one (x, y, z, index), put the new points behind it
    // (indices n .. n + M - 1) and run the O(N) counting-sort build over the grown array
    if (n > 0) hipLaunchKernelGGL(k_unsort, dim3(lk_cdiv(n, 256)), dim3(256), 0, st, h->sorted, n, h->pos_own);
    LK_HIP_TRY(hipMemcpyAsync(h->pos_own + 3 * (size_t)n, pos_new, sizeof(float) * 3 * (size_t)M, hipMemcpyDeviceToDevice, st));
    return lk_knn_build(h, h->pos_own, h->n + M, st);
}

extern "C" int lk_knn_query(lk_knn_t h, const float* q, int64_t P, float r2_scalar, const float* r2_per_query,
                            float* out_d2, int32_t* out_idx, int32_t* out_count, void* stream_) {
    LK_REQUIRE(h != nullptr, "lk_knn_query: NULL handle");
    LK_REQUIRE(P >= 0 && P < (1ll << 31), "lk_knn_query: P out of range");
    if (P == 0) return LK_OK;
    LK_REQUIRE(q && out_d2 && out_idx && out_count, "lk_knn_query: NULL buffer");
    if (P < (1 << 16))      // small batches: more lanes per query to shorten each lane's serial chain
        hipLaunchKernelGGL((k_knn_query<16>), dim3(lk_cdiv(P, 16)), dim3(256), 0, (hipStream_t)stream_, h->grid, h->sorted,
                           h->cell_start, q, (int)P, r2_scalar, r2_per_query, out_d2, out_idx, out_count);
    else
        hipLaunchKernelGGL((k_knn_query<8>), dim3(lk_cdiv(P, 32)), dim3(256), 0, (hipStream_t)stream_, h->grid, h->sorted,
                           h->cell_start, q, (int)P, r2_scalar, r2_per_query, out_d2, out_idx, out_count);
    LK_LAUNCH_CHECK();
    return LK_OK;
}
