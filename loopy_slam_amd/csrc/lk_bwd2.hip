// Backward, part 2:
//   k_interp_bwd    tracker mode: gradient through the interpolation weights to the sample position (decoder.py:191-229)
//   k_seg_* / k_feat_gather  d c_geo / d c_col / per-neighbour rows -> feature-row gradients (rows sorted by point, one atomic per run)
//   k_rays_bwd      d p -> d rays_o, d rays_d
//   k_relpos_bwd    backward of the relative-position neighbour MLP (decoder.py:477-488)
//   k_relpos_bwd_fused / k_dw2_hbar   mapper mode: the same backward with linear1's / linear2's weight gradients inside
//   k_wgrad(+_reduce)  colour-trunk weight gradients as streamed MFMA reductions over the sample rows
//   k_bwd_reduce    every partial-sum reduction of a mapper 'color' backward + the fc_c products (fc_post_body) in one launch
#include "lk_common.h"
#include "lk_kernels.h"
#include "lk_adam_dev.h"
#include "lk_exposure_dev.h"

using namespace lkw;

LK_CHAIN_DEFINE(bwd2)

// ---------------------------------------------------------------------------------------------
// one workgroup-sized block of 32 samples, 256 threads (8 per sample); `block` = index of the 32-sample block
__device__ __forceinline__ void interp_bwd_block(const LkInterpBwdArgs& a, int block) {
    const int sub = (int)threadIdx.x & 7;
    const int p_raw = block * 32 + ((int)threadIdx.x >> 3);
    const bool live = p_raw < a.P;
    const int pidx = live ? p_raw : a.P - 1;
    const bool has = a.nbr_count[pidx] >= a.min_nn;
    const bool color = (a.flags & LK_FLAG_STAGE_COLOR) != 0;
    const bool relpos = (a.flags & LK_FLAG_REL_POS) != 0;
    const bool do_col = color && !relpos;
    int id[LK_K];
    float w[LK_K];
    {
        const int4 i0 = *reinterpret_cast<const int4*>(a.nbr_idx + (size_t)pidx * LK_K);
        const int4 i1 = *reinterpret_cast<const int4*>(a.nbr_idx + (size_t)pidx * LK_K + 4);
        const float4 w0 = *reinterpret_cast<const float4*>(a.nbr_w + (size_t)pidx * LK_K);
        const float4 w1 = *reinterpret_cast<const float4*>(a.nbr_w + (size_t)pidx * LK_K + 4);
        id[0] = i0.x; id[1] = i0.y; id[2] = i0.z; id[3] = i0.w; id[4] = i1.x; id[5] = i1.y; id[6] = i1.z; id[7] = i1.w;
        w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w; w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
    }
    const float4 dcg = *reinterpret_cast<const float4*>(a.dc_geo + (size_t)pidx * LK_C + sub * 4);
    float4 dcc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (do_col) dcc = *reinterpret_cast<const float4*>(a.dc_col + (size_t)pidx * LK_C + sub * 4);
    if (!(a.flags & LK_FLAG_GRAD_RAYS)) return;
    // ---- tracker: d loss / d normalised weight_j = dc . feat_j  (+ the rel-pos branch's share)
    float dwn[LK_K];
#pragma unroll
    for (int j = 0; j < LK_K; ++j) {
        float part = 0.0f;
        if (has && w[j] != 0.0f) {
            const float4 g = lk_feat4(a.geo_feats, (a.flags & LK_FLAG_FEATS_F16) != 0, (size_t)id[j] * LK_C + sub * 4);
            part = dcg.x * g.x + dcg.y * g.y + dcg.z * g.z + dcg.w * g.w;
            if (do_col) {
                const float4 c = lk_feat4(a.col_feats, (a.flags & LK_FLAG_FEATS_F16) != 0, (size_t)id[j] * LK_C + sub * 4);
                part += dcc.x * c.x + dcc.y * c.y + dcc.z * c.z + dcc.w * c.w;
            }
        }
        part = lk_sum8<true>(part);
        if (color && relpos && a.dw_rel && has) part += a.dw_rel[(size_t)pidx * LK_K + j];
        dwn[j] = part;
    }
    LK_STAMPW(7);                                    // (probe build) lists, d c rows and the geometry rows arrived: d loss / d weight
    const int r = pidx / a.S;
    const float z = a.z[pidx];
    const float px = lk_madd_rn(a.rays_o[3 * r], a.rays_d[3 * r], z);
    const float py = lk_madd_rn(a.rays_o[3 * r + 1], a.rays_d[3 * r + 1], z);
    const float pz = lk_madd_rn(a.rays_o[3 * r + 2], a.rays_d[3 * r + 2], z);
    const float r2 = a.r2_ray ? a.r2_ray[r] : a.r2_static;
    float dpx = 0.0f, dpy = 0.0f, dpz = 0.0f;
    if (has) {
        float wr[LK_K], ex[LK_K], ey[LK_K], ez[LK_K];
        float S = 0.0f, dot = 0.0f;
#pragma unroll
        for (int j = 0; j < LK_K; ++j) {
            wr[j] = 0.0f; ex[j] = ey[j] = ez[j] = 0.0f;
            if (id[j] >= 0) {
                const float x = a.pos[3 * (size_t)id[j]], y = a.pos[3 * (size_t)id[j] + 1], zz = a.pos[3 * (size_t)id[j] + 2];
                const float D = lk_dist2(px, py, pz, x, y, zz);
                if (D <= r2) { wr[j] = 1.0f / (D + 1e-10f); ex[j] = x - px; ey[j] = y - py; ez[j] = zz - pz; }
            }
            S += wr[j];
            dot += dwn[j] * w[j];
        }
        const float invS = 1.0f / fmaxf(S, 1e-12f);
#pragma unroll
        for (int j = 0; j < LK_K; ++j) {
            if (wr[j] != 0.0f) {
                const float dw = (dwn[j] - dot) * invS;          // through the L1 normalisation
                const float dD = -wr[j] * wr[j] * dw;            // through 1/(D+eps)
                dpx += dD * (-2.0f) * ex[j]; dpy += dD * (-2.0f) * ey[j]; dpz += dD * (-2.0f) * ez[j];
            }
        }
    }
    if (live && sub == 0) {
        if (a.dp_embed) { const float4 e = *reinterpret_cast<const float4*>(a.dp_embed + (size_t)pidx * 4); dpx += e.x; dpy += e.y; dpz += e.z; }
        if (color && a.dp_embed_col) { const float4 e = *reinterpret_cast<const float4*>(a.dp_embed_col + (size_t)pidx * 4); dpx += e.x; dpy += e.y; dpz += e.z; }
        if (color && relpos && a.dp_rel) { const float4 e = *reinterpret_cast<const float4*>(a.dp_rel + (size_t)pidx * 4); dpx += e.x; dpy += e.y; dpz += e.z; }
        *reinterpret_cast<float4*>(a.dp_total + (size_t)pidx * 4) = make_float4(dpx, dpy, dpz, 0.0f);
    }
    LK_STAMPW(8);                                    // positions arrived, d p of the sample complete
    if (a.pose_part) {
        // tracking loop: the pose gradient needs G[c][k] = sum_rays (sum_s d p_c z) dir_k and T[c] = sum d p_c (k_pose_bwd); a sample
        // contributes d p_c z dir_k / d p_c, summed here over the 32 samples of the workgroup (one lane per sample carries it)
        __shared__ float s_pp[4][12];
        float v[12];
        const bool mine = live && sub == 0;
        const float d0 = mine ? (a.pix_i[r] - a.cx) / a.fx : 0.0f, d1 = mine ? -(a.pix_j[r] - a.cy) / a.fy : 0.0f, d2 = mine ? -1.0f : 0.0f;
        const float gx = mine ? dpx : 0.0f, gy = mine ? dpy : 0.0f, gz = mine ? dpz : 0.0f;
        const float zx = gx * z, zy = gy * z, zz = gz * z;
        v[0] = zx * d0; v[1] = zx * d1; v[2] = zx * d2; v[3] = zy * d0; v[4] = zy * d1; v[5] = zy * d2; v[6] = zz * d0; v[7] = zz * d1; v[8] = zz * d2;
        v[9] = gx; v[10] = gy; v[11] = gz;
#pragma unroll
        for (int q = 0; q < 12; ++q) {
#pragma unroll
            for (int o = 32; o >= 8; o >>= 1) v[q] += __shfl_xor(v[q], o);          // lanes 0, 8, .., 56 hold the samples of the wave
        }
        const int wv = (int)threadIdx.x >> 6;
        if (lk_lane() == 0) {
#pragma unroll
            for (int q = 0; q < 12; ++q) s_pp[wv][q] = v[q];
        }
        __syncthreads();
        if (threadIdx.x < 12) a.pose_part[(size_t)block * 12 + threadIdx.x] = (s_pp[0][threadIdx.x] + s_pp[1][threadIdx.x]) + (s_pp[2][threadIdx.x] + s_pp[3][threadIdx.x]);
    }
    LK_STAMPW(9);
}
__global__ __launch_bounds__(256) void k_interp_bwd(LkInterpBwdArgs a) { interp_bwd_block(a, (int)blockIdx.x); }
// the same launch with one more workgroup: the tracking iteration's exposure step (backward of the exposure MLP from the per-tile sums of d affine
// the decoder backward left, Adam, forward for the next iteration) - nothing between the decoder backward and the next forward reads what it
// writes, and in a launch of its own (k_track_final's second workgroup) it kept the pose step out of the next search launch
__global__ __launch_bounds__(256) void k_interp_bwd_x(LkInterpBwdArgs a, ExposureStepArgs xa, const float* __restrict__ part, int n_part) {
    if (blockIdx.x + 1 == gridDim.x) { lk_exposure_step_body(xa, part, n_part); return; }
    interp_bwd_block(a, (int)blockIdx.x);
}

// Feature-row gradients:
//   geometry rows:            g_geo[idx] += w * d c_geo[sample]
//   colour rows, no rel-pos:  g_col[idx] += w * d c_col[sample]
//   colour rows, rel-pos:     g_col[idx] += d feat[sample, neighbour]   (from k_relpos_bwd)
// Round 1 scattered them with one half-wave (32 channels = one 128-byte line) per (sample, neighbour) and an atomic per lane.
// Now: the same sums with a fraction of the atomics.  A mapper batch touches few points many times (5 000 rays x 5 samples x 8
// neighbours = 197 k rows on 15 k points of the benchmark frame: 12.7 rows per point), and the atomic scatter pays for every
// row twice (two tables) at the memory-side atomic rate, with the adds of a point serialised on its line.  The rows are
// counting-sorted by point - k_seg_count (rank of the row among the rows of its point; inside k_sample_interp when the forward
// knows that this backward follows), an exclusive scan of the per-point counts that also clears them, k_seg_place - which only
// needs the neighbour indices: it runs on the second stream beside the decoders.
// k_feat_gather then gives every half-wave (32 channels) 16 consecutive rows of the sorted list: the rows of a point are
// added in registers and flushed with ONE atomic per run (runs can continue in the next chunk) - about 27 k flushes instead
// of 197 k, every half-wave with the same amount of work (a per-point linked list walked by its first row was slower than
// the atomics: the longest chain sets the time).
__global__ __launch_bounds__(256) void k_seg_count(LkFeatScatterArgs a) {
    const long long row = (long long)blockIdx.x * 256 + (int)threadIdx.x;
    if (row >= (long long)a.P * LK_K) return;
    const int y = (int)blockIdx.y;                           // batch member (consecutive iterations)
    const long long grow = (long long)y * a.P * LK_K + row;
    const int s = (int)(row >> 3);
    const int idx = a.nbr_idx[grow];
    int rk = -1;
    const bool skipped = a.live_rays && s >= a.live_rays[y] * a.S;              // ray without a reading: its rows carry no gradient
    if (!skipped && idx >= 0 && a.nbr_w[grow] != 0.0f && a.nbr_count[(size_t)y * a.P + s] >= a.min_nn && (!a.row_mask || a.row_mask[idx])) {
        const int key = a.key_of ? a.key_of[idx] : idx;
        if (key >= 0) rk = atomicAdd(a.seg_cnt + (size_t)y * a.cnt_stride + key, 1);
    }
    a.seg_rank[grow] = rk;
}
__global__ __launch_bounds__(256) void k_seg_place(LkFeatScatterArgs a) {
    const long long row = (long long)blockIdx.x * 256 + (int)threadIdx.x;
    if (row >= (long long)a.P * LK_K) return;
    const int y = (int)blockIdx.y;
    const long long base = (long long)y * a.P * LK_K;
    const int32_t* __restrict__ off = a.seg_off + (size_t)y * a.cnt_stride;
    const int rk = a.seg_rank[base + row];
    if (rk >= 0) {
        const int idx = a.nbr_idx[base + row];
        a.seg_list[base + off[a.key_of ? a.key_of[idx] : idx] + rk] = (int)row;
    }
    if (row == 0 && a.seg_total) a.seg_total[y] = off[a.N];
}

// linear2 of the rel-pos MLP (see k_dw2_hbar below) as a body: block bx of nb.  LDS for 32 samples at a time (20 KB: as a rider of k_feat_gather
// it must not cost that kernel its eight workgroups per compute unit), a block's 64 samples in two passes.
#define LK_DW2_LDS_SAMPLES 32
__device__ __forceinline__ void dw2_body(const float* __restrict__ dc, const float* __restrict__ w_sum, const float* __restrict__ hbar,
                                         int P_all, const int32_t* __restrict__ live_rays, int S, float* __restrict__ part, int bx, int nb) {
    const int P = live_rays ? min(P_all, *live_rays * S) : P_all;
    __shared__ __attribute__((aligned(16))) float s_h[LK_DW2_LDS_SAMPLES][128];
    __shared__ float s_a[LK_DW2_LDS_SAMPLES][33];
    const int t = (int)threadIdx.x, n = t >> 3, k0 = 16 * (t & 7);
    float acc[16], bsum = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int s0 = bx * 64; s0 < P; s0 += nb * 64) {
        for (int half = 0; half < 64; half += LK_DW2_LDS_SAMPLES) {
            const int ns = min(LK_DW2_LDS_SAMPLES, P - s0 - half);
            if (ns <= 0) break;
            __syncthreads();
            // Hbar rows: ns x 32 float4, thread t takes every 256th; A' = wsum * d c: ns x 32 floats
#pragma unroll
            for (int q = 0; q < LK_DW2_LDS_SAMPLES * 32 / 256; ++q) {
                const int e = q * 256 + t, sm = e >> 5;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                float av = 0.0f;
                if (sm < ns) {
                    v = reinterpret_cast<const float4*>(hbar + (size_t)(s0 + half + sm) * 128)[e & 31];
                    av = w_sum[s0 + half + sm] * dc[(size_t)(s0 + half + sm) * LK_C + (e & 31)];
                }
                reinterpret_cast<float4*>(&s_h[sm][0])[e & 31] = v;
                s_a[sm][e & 31] = av;
            }
            __syncthreads();
#pragma unroll 4
            for (int sm = 0; sm < LK_DW2_LDS_SAMPLES; ++sm) {
                const float av = s_a[sm][n];
                const float4* __restrict__ hb = reinterpret_cast<const float4*>(&s_h[sm][k0]);
                bsum += av;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = hb[q];
                    acc[4 * q] = fmaf(av, v.x, acc[4 * q]); acc[4 * q + 1] = fmaf(av, v.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(av, v.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(av, v.w, acc[4 * q + 3]);
                }
            }
        }
    }
    float* __restrict__ out = part + (size_t)bx * (32 * 129) + n * 129;
#pragma unroll
    for (int i = 0; i < 16; ++i) out[k0 + i] = acc[i];
    if (k0 == 0) out[128] = bsum;
}
#define LK_GATHER_CHUNK 16
__device__ __forceinline__ void col_reduce_body(const float* __restrict__ part, int n_parts, int width, float* __restrict__ out, int bx, float (*sh)[32],
                                                const LkStepRider& sr);
__global__ __launch_bounds__(256) void k_feat_gather(LkFeatScatterArgs a) {
    // rider, FIRST in the grid: linear2 of the rel-pos MLP (k_dw2_hbar's blocks; it reads what the rel-pos backward left, as the gather does) -
    // as a launch of its own behind the gather it ran alone on the chip for 16 us of every 'color' iteration
    const int xb = a.x_on ? 1 : 0;                 // rider in block 0: the exposure step's backward + Adam half (LkFeatScatterArgs::x)
    if (xb && blockIdx.x == 0) { lk_exposure_step_body(a.x, nullptr, 0); return; }
    if ((int)blockIdx.x - xb < a.dw2_blocks) {
        dw2_body(a.dw2_dc, a.dw2_w_sum, a.dw2_hbar, a.P, a.dw2_live, a.dw2_S, a.dw2_part, (int)blockIdx.x - xb, a.dw2_blocks);
        return;
    }
    const int bx = (int)blockIdx.x - xb - a.dw2_blocks;
    if (a.red_part && bx >= a.red_block0) {     // rider: column sums of a partial table of the kernel before (one launch less)
        __shared__ float sh[8][32];
        LkStepRider none; none.n_span = 0;
        col_reduce_body(a.red_part, a.red_n, a.red_width, a.red_out, bx - a.red_block0, sh, none);
        return;
    }
    const int c = (int)threadIdx.x & 31;
    const long long i0 = ((long long)bx * 8 + ((int)threadIdx.x >> 5)) * LK_GATHER_CHUNK;
    const int total = a.seg_total ? *a.seg_total : a.seg_off[a.N];
    if (i0 >= total) return;
    const int n = min(LK_GATHER_CHUNK, (int)(total - i0));
    const bool col = a.dfeat != nullptr || a.dc_col != nullptr;
    int cur = -1;
    float sg = 0.0f, sc = 0.0f;
    for (int b = 0; b < n; b += 4) {                       // four rows per step: their loads are independent
        int row[4], idx[4];
        float vg[4], vc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) row[u] = a.seg_list[i0 + min(b + u, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            idx[u] = a.nbr_idx[row[u]];
            const float w = a.nbr_w[row[u]];
            const int s = row[u] >> 3;
            vg[u] = w * a.dc_geo[(size_t)s * LK_C + c];
            vc[u] = a.dfeat ? a.dfeat[(size_t)row[u] * LK_C + c] : (a.dc_col ? w * a.dc_col[(size_t)s * LK_C + c] : 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (b + u >= n) break;
            if (idx[u] != cur) {
                if (cur >= 0) {
                    atomicAdd(a.g_geo_feats + (size_t)cur * LK_C + c, sg);
                    if (col) atomicAdd(a.g_col_feats + (size_t)cur * LK_C + c, sc);
                    if (a.act_flag && c == 0) a.act_flag[cur] = 1;
                }
                cur = idx[u]; sg = 0.0f; sc = 0.0f;
            }
            sg += vg[u]; sc += vc[u];
        }
    }
    atomicAdd(a.g_geo_feats + (size_t)cur * LK_C + c, sg);
    if (col) atomicAdd(a.g_col_feats + (size_t)cur * LK_C + c, sc);
    if (a.act_flag && c == 0) a.act_flag[cur] = 1;
}

__global__ __launch_bounds__(256) void k_rays_bwd(LkRaysBwdArgs a) {
    const int r = blockIdx.x * 256 + (int)threadIdx.x;
    if (r >= a.R) return;
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
    for (int s = 0; s < a.S; ++s) {
        const int p = r * a.S + s;
        const float4 g = *reinterpret_cast<const float4*>(a.dp_total + (size_t)p * 4);
        const float z = a.z[p];
        ox += g.x; oy += g.y; oz += g.z;
        dx = fmaf(g.x, z, dx); dy = fmaf(g.y, z, dy); dz = fmaf(g.z, z, dz);
    }
    a.g_rays_o[3 * r] = ox; a.g_rays_o[3 * r + 1] = oy; a.g_rays_o[3 * r + 2] = oz;
    a.g_rays_d[3 * r] = dx; a.g_rays_d[3 * r + 1] = dy; a.g_rays_d[3 * r + 2] = dz;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float rp_embed_unit(const float* __restrict__ B, int u, float a0, float a1, float a2) {
    const int xi = (u < 10) ? u : u - 10;
    const float x = lk_fourier_arg(a0, a1, a2, B[xi], B[10 + xi], B[20 + xi]);
    return (u < 10) ? lk_sinf(x) : lk_cosf(x);
}

// H16 (the tracking loop's launch, k_relpos_interp_bwd: unit-scale colour loss gradients, no weight gradients): the two backward products on
// fp16 pieces of the 2^10-scaled chain instead of bf16 pieces, as decode_bwd_col_wg<true> and the mapper's fused variant below - d c_col is
// what the colour trunk's scaled chain delivered, so the same pre-scale puts d out, d hid in fp16's normal range; half the matrix
// instructions (72 instead of 144 per 32 rows) and 3 instead of 5.5 vector instructions per split value; d x is scaled back where it leaves.
template <bool F16, bool H16 = false>          // F16: half feature tables (LK_FLAG_FEATS_F16) - a template, the kernel has no register to spare for a branch
__device__ __forceinline__ void relpos_bwd_wave(const LkRelposBwdArgs& a, int sample0, float* __restrict__ part) {
    typedef BwdPiece<H16> PC;
    typedef typename PC::T Piece;
    constexpr float SC = H16 ? 1024.0f : 1.0f, ISC = H16 ? 1.0f / 1024.0f : 1.0f;
    const int lane = lk_lane();
    const int h = lane >> 5;
    const int j = lane & 31;
    const int sample = sample0 + (j >> 3);
    const bool live = sample < a.P;
    const int sp = live ? sample : a.P - 1;
    const int nb_i = j & 7;
    const int r = sp / a.S;
    const float z = a.z[sp];
    const float px = lk_madd_rn(a.rays_o[3 * r], a.rays_d[3 * r], z);
    const float py = lk_madd_rn(a.rays_o[3 * r + 1], a.rays_d[3 * r + 1], z);
    const float pz = lk_madd_rn(a.rays_o[3 * r + 2], a.rays_d[3 * r + 2], z);
    int idx = a.nbr_idx[(size_t)sp * LK_K + nb_i];
    const bool has = a.nbr_count[sp] >= a.min_nn;
    float wgt = (idx >= 0 && has && live) ? a.nbr_w[(size_t)sp * LK_K + nb_i] : 0.0f;
    if (idx < 0) idx = 0;
    const float a0 = __fmul_rn(LK_TWO_PI, __fsub_rn(a.pos[3 * (size_t)idx], px));
    const float a1 = __fmul_rn(LK_TWO_PI, __fsub_rn(a.pos[3 * (size_t)idx + 1], py));
    const float a2 = __fmul_rn(LK_TWO_PI, __fsub_rn(a.pos[3 * (size_t)idx + 2], pz));
    const float* __restrict__ W = a.W;
    const u32x4* __restrict__ FB = reinterpret_cast<const u32x4*>(a.Wfrag);
    const u32x4* __restrict__ FP = FB + (H16 ? FRAGB_U4 : 0);      // the backward products' pieces
    const size_t frow = (size_t)idx * LK_C;
    constexpr bool f16 = F16;
    const bool want_w = !H16 && (a.flags & LK_FLAG_GRAD_WEIGHTS) != 0;
    const bool want_p = (a.flags & LK_FLAG_GRAD_RAYS) != 0;
    // ---- recompute the forward of this tile
    f32x16 x0, x1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int u0 = 8 * g + 4 * h;
        if (u0 < ER) {
#pragma unroll
            for (int t = 0; t < 4; ++t) x0[4 * g + t] = rp_embed_unit(W + R_EB, u0 + t, a0, a1, a2);
        } else {
            const float4 v = lk_feat4(a.col_feats, f16, frow + (u0 - ER));
            x0[4 * g] = v.x; x0[4 * g + 1] = v.y; x0[4 * g + 2] = v.z; x0[4 * g + 3] = v.w;
        }
    }
    x1 = lk_zero16();
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int u0 = 32 + 8 * g + 4 * h;
        if (u0 < KR) {
            const float4 v = lk_feat4(a.col_feats, f16, frow + (u0 - ER));
            x1[4 * g] = v.x; x1[4 * g + 1] = v.y; x1[4 * g + 2] = v.z; x1[4 * g + 3] = v.w;
        }
    }
    LK_STAMPW(1);                                    // (probe build) lists, positions, feature rows arrived; embedding evaluated
    f32x16 hid[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) hid[nb] = lk_rowvec_tile(W + R_B1, nb * 32, lane);      // accumulators start from the bias
    // the recomputed forward uses the forward kernels' fp16x3 products (operands of O(1): embedding, features, activations)
    const u32x4* __restrict__ FH = FB + FRAGB_U4;
    lk_gemm_h3<4, 2>(hid, FH + FM20_FWDH, 4, 0, 0, x0, 0, lane);
    lk_gemm_h3<4, 2>(hid, FH + FM20_FWDH, 4, 2, 0, x1, 0, lane);       // units 32..55; registers 12..15 of x1 are zero
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) hid[nb][q] = lk_softplus100(hid[nb][q]);
    }
    LK_STAMP(2);                                     // hidden layer recomputed
    // ---- d out = w * dc ; (tracker) d w = dc . out
    f32x16 dout[1];
    const float* dcrow = a.dc_col + (size_t)sp * LK_C;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(dcrow + 8 * g + 4 * h);
        dout[0][4 * g] = v.x; dout[0][4 * g + 1] = v.y; dout[0][4 * g + 2] = v.z; dout[0][4 * g + 3] = v.w;
    }
    if (want_p) {
        f32x16 out[1];
        out[0] = lk_rowvec_tile(W + R_B2, 0, lane);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) lk_gemm_h3<1, 2>(out, FH + FM21_FWDH, 1, 2 * kb, 0, hid[kb], 0, lane);
        float part = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) part = fmaf(dout[0][q], out[0][q], part);
        part += __shfl_xor(part, 32);
        if (live && h == 0) a.dw_rel[(size_t)sp * LK_K + nb_i] = (has && a.nbr_idx[(size_t)sp * LK_K + nb_i] >= 0) ? part : 0.0f;
    }
    {
        const float ws = H16 ? wgt * SC : wgt;
#pragma unroll
        for (int q = 0; q < 16; ++q) dout[0][q] *= ws;
    }
    // Loads and stores share one in-order counter: a fragment fetched after a store cannot be waited for before that store
    // has landed.  Every product's fragments are therefore fetched BEFORE the stores that precede it in the data flow
    // (one block ahead, pinned with scheduling barriers), and the big row stores go last.
    // ---- operands of the streamed weight-gradient reductions that need hid itself
    //   linear2: dW2 = sum_rows (w d c) hid^T = sum_samples d c (sum_j w_j hid_j)^T  -> only the per-SAMPLE weighted
    //            hidden vector Hbar [P][128] and the per-sample weight sum are needed (8x fewer rows, no hid rows)
    if (want_w) {
        float wsum = wgt;
        wsum = lk_sum8<true>(wsum);
        if (live && h == 0 && nb_i == 0) a.w_sum[sp] = wsum;
    }
    // ---- d hid = (W2^T d out) * softplus'(hid), block by block IN PLACE of hid (64 fewer live registers)
    f32x16 (&dhid)[4] = hid;
    const Piece db0 = PC::split(dout[0], 0), db1 = PC::split(dout[0], 1);
    Piece fa0 = PC::load(FP + PC::tr(21), 4, 0, 0, lane), fa1 = PC::load(FP + PC::tr(21), 4, 1, 0, lane);
    Piece fx[4];                                  // head of the d x product (block 0 of d hid), fetched before the last stores
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        f32x16 t = lk_zero16();
        t = PC::mma(fa0, db0, t);
        t = PC::mma(fa1, db1, t);
        if (nb < 3) {
            fa0 = PC::load(FP + PC::tr(21), 4, 0, nb + 1, lane);
            fa1 = PC::load(FP + PC::tr(21), 4, 1, nb + 1, lane);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) fx[q] = PC::load(FP + PC::tr(20), 2, q >> 1, q & 1, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (want_w) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    float sred = wgt * hid[nb][4 * g + tt];
                    sred = lk_sum8<true>(sred);
                    v[tt] = sred;
                }
                if (live && nb_i == 0)
                    *reinterpret_cast<float4*>(a.hbar + (size_t)sp * 128 + nb * 32 + 8 * g + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) dhid[nb][q] = t[q] * lk_softplus100_grad_from_out(hid[nb][q]);
    }
    LK_STAMP(3);                                     // d hid
    // ---- d x = W1^T d hid   (virtual 64 input units: 0..19 embedding, 20..51 feature channels)
    f32x16 dx[2];
    dx[0] = lk_zero16(); dx[1] = lk_zero16();
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int G = 0; G < 2; ++G) {
            const Piece b = PC::split(dhid[nb], G);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                dx[kb] = PC::mma(nb == 0 ? fx[2 * G + kb] : PC::load(FP + PC::tr(20), 2, 2 * nb + G, kb, lane), b, dx[kb]);
        }
    }
    if (H16) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int q = 0; q < 16; ++q) dx[kb][q] *= ISC;
    }
    __builtin_amdgcn_sched_barrier(0);
    LK_STAMP(4);                                     // d x
    if (want_w) {
        //   linear1: rows [8P][192] = d hid (128) | x (64)
        if (live) {
            if (h == 0) a.w_eff[(size_t)sp * 8 + nb_i] = wgt;
            float* row = a.rows + ((size_t)sp * 8 + nb_i) * 192;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(row + nb * 32 + 8 * g + 4 * h) =
                        make_float4(dhid[nb][4 * g], dhid[nb][4 * g + 1], dhid[nb][4 * g + 2], dhid[nb][4 * g + 3]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<float4*>(row + 128 + 8 * g + 4 * h) = make_float4(x0[4 * g], x0[4 * g + 1], x0[4 * g + 2], x0[4 * g + 3]);
                *reinterpret_cast<float4*>(row + 160 + 8 * g + 4 * h) = make_float4(x1[4 * g], x1[4 * g + 1], x1[4 * g + 2], x1[4 * g + 3]);
            }
        }
    }
    float dax = 0.0f, day = 0.0f, daz = 0.0f;        // d loss / d (x_I - p), this lane's share
    const float* B = W + R_EB;
    // d sin = cos, d cos = -sin: the derivative of embedding unit u is (+/-) the FORWARD value of its partner unit u +/- 10,
    // which x0 already holds - in this lane for 12 of the 20 units, in the lane of the other half-wave for units
    // 2, 3, 6, 7 / 12, 13, 16, 17 (four exchanges) - instead of twelve more sin / cos evaluations per lane.
    float fder[12];
    {
        const float r0 = __shfl_xor(h ? x0[4] : x0[8], 32), r1 = __shfl_xor(h ? x0[5] : x0[9], 32);
        const float r2 = __shfl_xor(x0[2], 32), r3 = __shfl_xor(x0[3], 32);
        fder[0] = x0[6]; fder[1] = x0[7]; fder[2] = r0; fder[3] = r1;                                        // units 0-3 | 4-7
        fder[4] = h ? -r2 : x0[10]; fder[5] = h ? -r3 : x0[11]; fder[6] = -x0[0]; fder[7] = -x0[1];          // units 8-11 | 12-15
        fder[8] = -r2; fder[9] = -r3; fder[10] = -x0[4]; fder[11] = -x0[5];                                  // units 16-19 (low half)
    }
#pragma unroll
    for (int tile = 0; tile < 2; ++tile)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int u0 = 32 * tile + 8 * g + 4 * h;
            const bool is_emb = u0 < ER;                 // uniform per half-wave (group 2 of tile 0 is mixed)
            if (tile == 0 && g < 3) {                    // groups that hold embedding units in at least one half
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int u = u0 + t;
                    const int xi = is_emb ? ((u < 10) ? u : u - 10) : 0;
                    const float b0 = B[xi], b1 = B[10 + xi], b2 = B[20 + xi];
                    float gx = 0.0f;
                    if (is_emb) gx = dx[tile][4 * g + t] * fder[4 * g + t];
                    if (want_p) {
                        dax = fmaf(gx * LK_TWO_PI, b0, dax); day = fmaf(gx * LK_TWO_PI, b1, day); daz = fmaf(gx * LK_TWO_PI, b2, daz);
                    }
                    if (want_w) {                        // every lane takes part in the reductions (convergent shuffles)
                        const float s0 = lk_half_wave_sum(gx * a0), s1 = lk_half_wave_sum(gx * a1), s2 = lk_half_wave_sum(gx * a2);
                        if ((lane & 31) == LK_HWS_LANE && is_emb) {   // two units (sin, cos) and both half-waves meet in one slot: LDS atomics
                            atomicAdd(part + xi, s0);
                            atomicAdd(part + 10 + xi, s1);
                            atomicAdd(part + 20 + xi, s2);
                        }
                    }
                }
            }
            if (!is_emb && u0 < KR) {     // d feature row of this neighbour: summed per point by k_feat_gather
                if ((a.flags & LK_FLAG_GRAD_FEATS) && live)
                    *reinterpret_cast<float4*>(a.dfeat + ((size_t)sp * 8 + nb_i) * LK_C + (u0 - ER)) =
                        make_float4(dx[tile][4 * g], dx[tile][4 * g + 1], dx[tile][4 * g + 2], dx[tile][4 * g + 3]);
            }
        }
    if (want_p) {
        // both halves of a row, then the 8 neighbour rows of the sample; d p = - d (x_I - p)
        dax += __shfl_xor(dax, 32); day += __shfl_xor(day, 32); daz += __shfl_xor(daz, 32);
        dax = lk_sum8<true>(dax);
        day = lk_sum8<true>(day);
        daz = lk_sum8<true>(daz);
        if (live && h == 0 && nb_i == 0) *reinterpret_cast<float4*>(a.dp_rel + (size_t)sp * 4) = make_float4(-dax, -day, -daz, 0.0f);
    }
}

template <bool F16>
__global__ __launch_bounds__(256, 2) void k_relpos_bwd(LkRelposBwdArgs a) {
    __shared__ float s_part[4][32];
    const bool want_w = (a.flags & LK_FLAG_GRAD_WEIGHTS) != 0;
    const int w = (int)threadIdx.x >> 6;
    if (want_w) {
        if (threadIdx.x < 128) (&s_part[0][0])[threadIdx.x] = 0.0f;
        __syncthreads();
    }
    const int sample0 = (blockIdx.x * 4 + w) * 4;
    if (sample0 < a.P) relpos_bwd_wave<F16>(a, sample0, s_part[w]);
    if (want_w) {
        __syncthreads();
        if (threadIdx.x < 32)
            a.part_br[(size_t)blockIdx.x * 32 + threadIdx.x] = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] +
                                                                 s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
    }
}

// Tracker-sized batches (lk_track_frame): the rel-pos backward and the interpolation backward of a 32-sample block in ONE launch - eight
// waves run the rel-pos backward (4 samples each, as k_relpos_bwd), one barrier, waves 4..7 leave, waves 0..3 are k_interp_bwd's block.
// No weight gradients here (the tracker optimises the pose only).
template <bool F16, bool H16>
__global__ __launch_bounds__(512) void k_relpos_interp_bwd(LkRelposBwdArgs rb, LkInterpBwdArgs ib) {
    __shared__ float s_dummy[32];
    const int w = (int)threadIdx.x >> 6;
    const int sample0 = (int)blockIdx.x * 32 + 4 * w;
    LK_STAMP(0);
    if (sample0 < rb.P) relpos_bwd_wave<F16, H16>(rb, sample0, s_dummy);
    LK_STAMP(5);
    __syncthreads();                                   // d w_rel / d p_rel of the block are written
    LK_STAMP(6);
    if (w >= 4) return;
    interp_bwd_block(ib, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// FUSED MAPPER VARIANT (lk_relpos_fused(flags): weight gradients wanted, unit-scale loss gradients, no ray gradients).
// The plain kernel above hands linear1's weight gradient to k_wgrad through `rows` - 6 KB written per sample and read back
// 1.4 times: two thirds of the HBM traffic of a mapper 'color' iteration.  Here
//     dW1[u][i] = sum_rows d hid[u][row] * x[i][row]
// is taken inside the kernel.  Both operands live as CT tiles (lane = row), the reduction index of a matrix instruction
// is the one index that is NOT a lane, so the tile has to be turned: the fp16 piece registers that the other products
// need anyway (x: the forward recompute, d hid: the d x product) are also written to a per-wave LDS image [piece][unit][row]
// with two-byte stores (no extra VALU work) and read back as 16-byte A / B operands - 8 consecutive rows of one unit per
// lane.  The 16-byte chunk of rows c of unit u sits at chunk c ^ (u >> 2 & 3): the 16 lanes an LDS read serves together then
// hit 16 different 16-byte slots of the 256-byte bank row.  Input unit 52 of the x image is the constant 1: column 52 of the
// product is the bias gradient.  The images of the four waves of a workgroup (4 x 32 rows) are consumed TOGETHER: after a
// barrier wave w owns one 32 x 32 block of the product - hidden block 2 p + (w >> 1), input block w & 1, in the two phases
// p = 0, 1 (d hid is staged two blocks at a time: that is what fits twice into the 160 KB of a compute unit) - and walks the
// 128 staged rows with 24 matrix instructions per phase into two accumulators it keeps in REGISTERS for the whole kernel.
// (A first version added every wave's tile into one LDS tile with float atomics: ds_add_f32 retires about one lane every two
// cycles, the kernel ran three times slower than the one it replaces.)  Workgroups are persistent (grid <= 2 per compute
// unit) and store their 128 x 64 tile once, k_bwd_reduce adds the <= 512 partial tiles: fixed summation order, no atomics.
// All products run on scaled fp16 pieces (d out * 2^10, as in decode_bwd_col_wg<true>): half the matrix instructions and
// 3 instead of 5.5 VALU instructions per split value of the bf16 path.
#define RPF_XU 56                                            // staged input units: 0..51 real, 52 = 1, 53..55 = 0
#define RPF_DU 64                                            // staged d hid units: two 32-unit blocks at a time
#define RPF_STAGE_HALVES ((2 * RPF_XU + 2 * RPF_DU) * 32)    // per wave: x pieces [2][56][32] + d hid pieces [2][64][32]
#define RPF_SC 1024.0f
#define RPF_ISC (1.0f / 1024.0f)

// lane-dependent parts of the image addresses (in halves), computed once per wave: everything else is a compile-time constant
struct RpfLane {
    int wr[2];        // store slot of (unit 4 h, this lane's row) for units with (unit >> 2 & 3) == (2 k + h) & 3, k = 0, 1
    int rd_j[2];      // operand slot of unit (lane & 31), row half kk = 0, 1
    int rd_x[2];      // operand slot of input unit 32 + (lane & 31), clamped to the zero unit 55
};
__device__ __forceinline__ RpfLane rpf_lane(int lane) {
    const int h = lane >> 5, row = lane & 31;
    RpfLane L;
#pragma unroll
    for (int k = 0; k < 2; ++k) L.wr[k] = 4 * h * 32 + ((((row >> 3) ^ (2 * k + h)) & 3) << 3) + (row & 7);
    const int xu = (32 + row < RPF_XU) ? 32 + row : RPF_XU - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        L.rd_j[kk] = row * 32 + ((((2 * kk + h) ^ (row >> 2)) & 3) << 3);
        L.rd_x[kk] = xu * 32 + ((((2 * kk + h) ^ (xu >> 2)) & 3) << 3);
    }
    return L;
}
// the piece registers of CT-tile registers 8G..8G+7 (lk_split_cth: register j = units u, u + 1 of this lane's row) -> images;
// unit u = unit0 + 2 (j & 1) + 8 (2 G + (j >> 1)) + 4 h sits at chunk (row >> 3) ^ (u >> 2 & 3) = (row >> 3) ^ (2 (j >> 1) + h & 3)
__device__ __forceinline__ void rpf_stage(uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, const LkH8& b, int G, int unit0,
                                          int n_units, const RpfLane& L, int h) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = unit0 + 2 * (j & 1) + 8 * (2 * G + (j >> 1));                  // compile-time part of the unit
        if (c + 4 < n_units || (c < n_units && h == 0)) {                            // unit c + 4 h (and its odd partner) exists
            const int s = c * 32 + L.wr[j >> 1];
            hi[s] = (uint16_t)(b.p[0][j] & 0xffffu); hi[s + 32] = (uint16_t)(b.p[0][j] >> 16);
            lo[s] = (uint16_t)(b.p[1][j] & 0xffffu); lo[s + 32] = (uint16_t)(b.p[1][j] >> 16);
        }
    }
}
// matrix-instruction operand at a lane slot of RpfLane: 8 consecutive rows of one unit
__device__ __forceinline__ u32x4 rpf_operand(const uint16_t* __restrict__ img, int slot) {
    return *reinterpret_cast<const u32x4*>(img + slot);
}

// WG = false (LK_FLAG_EMBED_GRADS_ONLY: the MLP's matrices are frozen, only the Fourier matrix and the feature rows get gradients): no
// LDS images, no linear1 blocks, no per-sample Hbar / weight-sum rows, no barriers - the same d x / d feature / d B arithmetic.
template <bool WG, bool F16>     // F16: half feature tables, a template as in relpos_bwd_wave (the run-time branch cost ~100 instructions of moves per tile)
__device__ __forceinline__ void relpos_bwd_wave_fused(const LkRelposBwdArgs& a, int sample0, int P_live, float* __restrict__ part,
                                                      uint16_t* __restrict__ stage_wg, int w, f32x16 (&acc)[2]) {
    const int lane = lk_opaque(lk_lane());            // per tile: see lk_opaque
    const int h = lane >> 5;
    const int j = lane & 31;
    const int sample = sample0 + (j >> 3);
    const bool live = sample < P_live;
    const int sp = live ? sample : P_live - 1;
    const int nb_i = j & 7;
    const int r = sp / a.S;
    const float z = a.z[sp];
    const float px = lk_madd_rn(a.rays_o[3 * r], a.rays_d[3 * r], z);
    const float py = lk_madd_rn(a.rays_o[3 * r + 1], a.rays_d[3 * r + 1], z);
    const float pz = lk_madd_rn(a.rays_o[3 * r + 2], a.rays_d[3 * r + 2], z);
    int idx = a.nbr_idx[(size_t)sp * LK_K + nb_i];
    const bool has = a.nbr_count[sp] >= a.min_nn;
    const float wgt = (idx >= 0 && has && live) ? a.nbr_w[(size_t)sp * LK_K + nb_i] : 0.0f;
    if (idx < 0) idx = 0;
    const float a0 = __fmul_rn(LK_TWO_PI, __fsub_rn(a.pos[3 * (size_t)idx], px));
    const float a1 = __fmul_rn(LK_TWO_PI, __fsub_rn(a.pos[3 * (size_t)idx + 1], py));
    const float a2 = __fmul_rn(LK_TWO_PI, __fsub_rn(a.pos[3 * (size_t)idx + 2], pz));
    const float* __restrict__ W = a.W;
    const u32x4* __restrict__ FH = reinterpret_cast<const u32x4*>(a.Wfrag) + FRAGB_U4;
    const size_t frow = (size_t)idx * LK_C;
    uint16_t* __restrict__ stage = stage_wg + w * RPF_STAGE_HALVES;
    uint16_t* __restrict__ xt_hi = stage;
    uint16_t* __restrict__ xt_lo = stage + RPF_XU * 32;
    uint16_t* __restrict__ dt_hi = stage + 2 * RPF_XU * 32;
    uint16_t* __restrict__ dt_lo = dt_hi + RPF_DU * 32;
    const RpfLane RL = rpf_lane(lane);
    // ---- recompute the forward of this tile; the fp16 pieces of x go to the LDS image on the way
    f32x16 x0, x1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int u0 = 8 * g + 4 * h;
        if (u0 < ER) {
#pragma unroll
            for (int t = 0; t < 4; ++t) x0[4 * g + t] = rp_embed_unit(W + R_EB, u0 + t, a0, a1, a2);
        } else {
            const float4 v = lk_feat4(a.col_feats, F16, frow + (u0 - ER));
            x0[4 * g] = v.x; x0[4 * g + 1] = v.y; x0[4 * g + 2] = v.z; x0[4 * g + 3] = v.w;
        }
    }
    x1 = lk_zero16();
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int u0 = 32 + 8 * g + 4 * h;
        if (u0 < KR) {
            const float4 v = lk_feat4(a.col_feats, F16, frow + (u0 - ER));
            x1[4 * g] = v.x; x1[4 * g + 1] = v.y; x1[4 * g + 2] = v.z; x1[4 * g + 3] = v.w;
        }
    }
    // derivative of embedding unit u = (+/-) the forward value of its partner unit u +/- 10 (see relpos_bwd_wave): taken now,
    // x0 is dead after the forward products
    float fder[12];
    {
        const float r0 = __shfl_xor(h ? x0[4] : x0[8], 32), r1 = __shfl_xor(h ? x0[5] : x0[9], 32);
        const float r2 = __shfl_xor(x0[2], 32), r3 = __shfl_xor(x0[3], 32);
        fder[0] = x0[6]; fder[1] = x0[7]; fder[2] = r0; fder[3] = r1;
        fder[4] = h ? -r2 : x0[10]; fder[5] = h ? -r3 : x0[11]; fder[6] = -x0[0]; fder[7] = -x0[1];
        fder[8] = -r2; fder[9] = -r3; fder[10] = -x0[4]; fder[11] = -x0[5];
    }
    f32x16 hid[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) hid[nb] = lk_rowvec_tile(W + R_B1, nb * 32, lane);
    {   // weight fragments one 16-unit block ahead of the products (pinned: the scheduler would otherwise fetch all four at once)
        LkH8 fr[4], nx[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) fr[nb] = lk_fragh_load(FH + FM20_FWDH, 4, 0, nb, lane);
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            const LkH8 b = lk_split_cth(G < 2 ? x0 : x1, G & 1);
            if (WG) rpf_stage(xt_hi, xt_lo, b, G & 1, G < 2 ? 0 : 32, KR, RL, h);
            if (G < 3) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) nx[nb] = lk_fragh_load(FH + FM20_FWDH, 4, G + 1, nb, lane);
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) hid[nb] = lk_mma3h(fr[nb], b, hid[nb]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) fr[nb] = nx[nb];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) hid[nb][q] = lk_softplus100(hid[nb][q]);
    }
    // ---- d out = w * d c * 2^10
    f32x16 dout;
    const float* dcrow = a.dc_col + (size_t)sp * LK_C;
    const float wsc = wgt * RPF_SC;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(dcrow + 8 * g + 4 * h);
        dout[4 * g] = v.x * wsc; dout[4 * g + 1] = v.y * wsc; dout[4 * g + 2] = v.z * wsc; dout[4 * g + 3] = v.w * wsc;
    }
    if (WG) {   // linear2 is reduced per SAMPLE (see relpos_bwd_wave): weight sum and weighted hidden vector
        const float wsum = lk_sum8<true>(wgt);
        if (live && h == 0 && nb_i == 0) a.w_sum[sp] = wsum;
    }
    // ---- d hid = (W2^T d out) * softplus'(hid), block by block in place of hid
    f32x16 (&dhid)[4] = hid;
    const LkH8 db0 = lk_split_cth(dout, 0), db1 = lk_split_cth(dout, 1);
    LkH8 fa0 = lk_fragh_load(FH + FM21_TRH, 4, 0, 0, lane), fa1 = lk_fragh_load(FH + FM21_TRH, 4, 1, 0, lane);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        f32x16 t = lk_zero16();
        t = lk_mma3h(fa0, db0, t);
        t = lk_mma3h(fa1, db1, t);
        if (nb < 3) {
            fa0 = lk_fragh_load(FH + FM21_TRH, 4, 0, nb + 1, lane);
            fa1 = lk_fragh_load(FH + FM21_TRH, 4, 1, nb + 1, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (WG) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) v[tt] = lk_sum8<true>(wgt * hid[nb][4 * g + tt]);
                if (live && nb_i == 0)
                    *reinterpret_cast<float4*>(a.hbar + (size_t)sp * 128 + nb * 32 + 8 * g + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) dhid[nb][q] = t[q] * lk_softplus100_grad_from_out(hid[nb][q]);
    }
    // ---- d x = W1^T d hid and, block by block of d hid, dW1[block] += d hid[block] (x)^T through the LDS images
    f32x16 dx[2];
    dx[0] = lk_zero16(); dx[1] = lk_zero16();
    LkH8 fx[4];                                           // W1^T fragments of one d hid block, fetched one block ahead
#pragma unroll
    for (int q = 0; q < 4; ++q) fx[q] = lk_fragh_load(FH + FM20_TRH, 2, q >> 1, q & 1, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int G = 0; G < 2; ++G) {
            const LkH8 b = lk_split_cth(dhid[nb], G);
            if (WG) rpf_stage(dt_hi, dt_lo, b, G, 32 * (nb & 1), RPF_DU, RL, h);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) dx[kb] = lk_mma3h(fx[2 * G + kb], b, dx[kb]);
        }
        if (nb < 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) fx[q] = lk_fragh_load(FH + FM20_TRH, 2, 2 * (nb + 1) + (q >> 1), q & 1, lane);
        }
        if (WG && (nb & 1)) {
            // phase p = nb >> 1: blocks 2 p, 2 p + 1 of every wave's d hid are staged.  This wave's block of the product:
            // hidden units 32 (2 p + (w >> 1)) .., input units 32 (w & 1) .., over the rows of all four waves.
            __syncthreads();
            const int p = nb >> 1;
            const int da = 32 * 32 * (w >> 1);                               // second staged block = units 32..63 of the image
#pragma unroll
            for (int src = 0; src < 4; ++src) {
                const uint16_t* __restrict__ sx = stage_wg + src * RPF_STAGE_HALVES;
                const uint16_t* __restrict__ sd = sx + 2 * RPF_XU * 32;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int xs = (w & 1) ? RL.rd_x[kk] : RL.rd_j[kk];     // input units 56..63 do not exist: the zero unit 55
                    const u32x4 ah = rpf_operand(sd, da + RL.rd_j[kk]), al = rpf_operand(sd + RPF_DU * 32, da + RL.rd_j[kk]);
                    const u32x4 bh = rpf_operand(sx, xs), bl = rpf_operand(sx + RPF_XU * 32, xs);
                    acc[p] = lk_mfma_f16(al, bh, acc[p]);
                    acc[p] = lk_mfma_f16(ah, bl, acc[p]);
                    acc[p] = lk_mfma_f16(ah, bh, acc[p]);
                }
            }
            __syncthreads();                             // everyone has read the images before they are overwritten
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- d B (Fourier matrix) and the d feature rows, from d x / 2^10
#pragma unroll
    for (int tile = 0; tile < 2; ++tile)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int u0 = 32 * tile + 8 * g + 4 * h;
            const bool is_emb = u0 < ER;
            if (tile == 0 && g < 3) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int u = u0 + t;
                    const int xi = is_emb ? ((u < 10) ? u : u - 10) : 0;
                    float gx = 0.0f;
                    if (is_emb) gx = dx[tile][4 * g + t] * RPF_ISC * fder[4 * g + t];
                    const float s0 = lk_half_wave_sum(gx * a0), s1 = lk_half_wave_sum(gx * a1), s2 = lk_half_wave_sum(gx * a2);
                    if ((lane & 31) == LK_HWS_LANE && is_emb) {
                        atomicAdd(part + xi, s0);
                        atomicAdd(part + 10 + xi, s1);
                        atomicAdd(part + 20 + xi, s2);
                    }
                }
            }
            if (!is_emb && u0 < KR) {
                if ((a.flags & LK_FLAG_GRAD_FEATS) && live)
                    *reinterpret_cast<float4*>(a.dfeat + ((size_t)sp * 8 + nb_i) * LK_C + (u0 - ER)) =
                        make_float4(dx[tile][4 * g] * RPF_ISC, dx[tile][4 * g + 1] * RPF_ISC, dx[tile][4 * g + 2] * RPF_ISC, dx[tile][4 * g + 3] * RPF_ISC);
            }
        }
}

template <bool WG, bool F16>
__global__ __launch_bounds__(256, 2) void k_relpos_bwd_fused(LkRelposBwdArgs a) {
    __shared__ float s_part[4][32];
    __shared__ __attribute__((aligned(16))) uint16_t s_stage[WG ? 4 * RPF_STAGE_HALVES : 8];
    const int w = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    if (threadIdx.x < 128) (&s_part[0][0])[threadIdx.x] = 0.0f;
    if (WG) {   // input units 52..55 of the wave's x image: the constant 1 (bias column) and three zero units; never rewritten
        uint16_t* xt = s_stage + w * RPF_STAGE_HALVES;
        for (int i = lane; i < 2 * 4 * 32; i += 64) {
            const int piece = i >> 7, u = KR + ((i >> 5) & 3);
            xt[piece * RPF_XU * 32 + u * 32 + (i & 31)] = (piece == 0 && u == KR) ? (uint16_t)0x3C00u : (uint16_t)0;
        }
    }
    f32x16 acc[2];
    acc[0] = lk_zero16(); acc[1] = lk_zero16();
    __syncthreads();
    // every wave runs every tile of the workgroup (barriers inside); rows past the end are dead lanes with weight 0
    const int P_live = a.live_rays ? min(a.P, *a.live_rays * a.S) : a.P;       // the rays without a reading sit behind the prefix
    for (int t = (int)blockIdx.x; t * 16 < P_live; t += (int)gridDim.x)
        relpos_bwd_wave_fused<WG, F16>(a, (t * 4 + w) * 4, P_live, s_part[w], s_stage, w, acc);
    __syncthreads();
    if (threadIdx.x < 32)
        a.part_br[(size_t)blockIdx.x * 32 + threadIdx.x] = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] +
                                                             s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
    if (!WG) return;
    // accumulator p of wave w: hidden units 32 (2 p + (w >> 1)) + row(q, h), input units 32 (w & 1) + (lane & 31)
    float* __restrict__ out = a.dw1_part + (size_t)blockIdx.x * (128 * 64);
    const int h = lane >> 5, j = lane & 31;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 16; ++q) out[(32 * (2 * p + (w >> 1)) + lk_frag_row(q, h)) * 64 + 32 * (w & 1) + j] = acc[p][q] * RPF_ISC;
}

// linear2 of the rel-pos MLP in the fused variant: dW2[n][k] = sum_samples (wsum_s d c_s[n]) Hbar_s[k], db2[n] = sum_s wsum_s d c_s[n]
// ([32][128] + [32] outputs, P samples: 0.2 GFLOP - too small for the k_wgrad + k_wgrad_reduce pair, whose two launches cost
// 50 us on the weight-gradient stream).  A workgroup stages 64 samples in LDS with coalesced 16-byte loads (the walk over the
// samples is otherwise a chain of dependent load latencies), then thread t adds them into output row n = t >> 3, columns
// 16 (t & 7) .. + 15, and stores a [32][129] partial tile (column 128 = bias), summed by k_bwd_reduce.
#define LK_DW2_PARTS 512
#define LK_DW2_TILE (32 * 129)
#define LK_DW2_SAMPLES 64
__global__ __launch_bounds__(256) void k_dw2_hbar(const float* __restrict__ dc, const float* __restrict__ w_sum, const float* __restrict__ hbar,
                                                  int P_all, const int32_t* __restrict__ live_rays, int S, float* __restrict__ part) {
    dw2_body(dc, w_sum, hbar, P_all, live_rays, S, part, (int)blockIdx.x, (int)gridDim.x);
}

// Sums of the partial tiles of the fused variant (a part of k_bwd_reduce): linear1 [n1][128][64] (column 52 = bias) from
// k_relpos_bwd_fused, linear2 [n2][32][129] (column 128 = bias) from k_dw2_hbar.  32 consecutive elements x 8 partial lanes per workgroup; every output has one
// owner: no atomics, fixed order.
// Final write of a decoder gradient element by its one owner: plain accumulation, or (step rider, lk_kernels.h) the Adam step of the
// element right here when it belongs to a stepped span
__device__ __forceinline__ void grad_commit(float* __restrict__ gptr, float s, const LkStepRider& sr) {
    if (sr.n_span > 0) {
        const long long off = gptr - sr.g;
        for (int q = 0; q < sr.n_span; ++q) {
            const LkStepSpan sp = sr.span[q];
            if (off >= sp.off && off < (long long)sp.off + sp.n) {
                const float g = *gptr + s;
                float m = sr.m[off], v = sr.v[off];
                sr.w_next[off] = lk_adam_elem(sr.p[off], g, m, v, sr.beta1, sr.beta2, sr.eps, sp.step_size, sp.bc2_sqrt);
                sr.m[off] = m; sr.v[off] = v;
                *gptr = 0.0f;
                return;
            }
        }
    }
    *gptr += s;
}
__device__ __forceinline__ void rp_reduce_body(const float* __restrict__ part1, int n1, const float* __restrict__ part2, int n2,
                                               float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2,
                                               int bx, float (*sh)[32], const LkStepRider& sr) {
    const int e = (int)threadIdx.x & 31, q = (int)threadIdx.x >> 5;
    const bool second = bx >= 128 * 64 / 32;
    const int o = (bx - (second ? 128 * 64 / 32 : 0)) * 32 + e;
    const int tile = second ? LK_DW2_TILE : 128 * 64, n_parts = second ? n2 : n1;
    const bool in = o < tile;
    const float* __restrict__ src = (second ? part2 : part1) + (in ? o : 0);
    float s = 0.0f;
#pragma unroll 4
    for (int y = q; y < n_parts; y += 8) s += src[(size_t)y * tile];
    sh[q][e] = s;
    __syncthreads();
    if (q != 0 || !in) return;
    s = ((sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e])) + ((sh[4][e] + sh[5][e]) + (sh[6][e] + sh[7][e]));
    if (!second) {
        const int n = o >> 6, k = o & 63;
        if (k < KR) grad_commit(dW1 + (size_t)n * KRP + k, s, sr);
        else if (k == KR) grad_commit(db1 + n, s, sr);
    } else {
        const int n = o / 129, k = o - n * 129;
        if (k < 128) grad_commit(dW2 + (size_t)n * 128 + k, s, sr);
        else grad_commit(db2 + n, s, sr);
    }
}
#define LK_RP_REDUCE_BLOCKS (128 * 64 / 32 + (LK_DW2_TILE + 31) / 32)

// ---------------------------------------------------------------------------------------------
// dW[n][k] += sum_rows A'[row][n] * B[row][k]  for every decoder matrix (one "job" each): a reduction GEMM with a
// small output (<= 128 x 168) and a very long reduction (the rows = samples or neighbour rows).
//
// One WAVE per (unit, chunk of rows), no LDS, no barriers.  A unit is an (N piece) x (K piece) of a job, a piece being
// 32*V consecutive columns (V = 1 or 2) handled as V interleaved 32x32 MFMA blocks: lane i of a half-wave owns
// columns piece0 + V*i .. +V-1, so ONE V-float load per lane is, for a pair of rows (half-wave = row), at the same
// time a fully coalesced 128*V-byte row segment and the A (resp. B) operand of v_mfma_f32_32x32x2_f32 for V blocks
// (block b = the columns congruent to b mod V — any column permutation is as good as another for an outer product).
// Per pair of rows a (2,2) unit issues 3 loads (A, the element-wise factor source A2, B) and 4 MFMAs; loads run
// WG_STEPS - 1 row pairs ahead of the MFMAs in a register ring.  The bias gradient is the running column sum of the A' operand.
// Global float atomics are the scarce resource here (they execute memory-side: ~85 G lane-adds/s measured - flushing
// a 64x64 unit costs as much as 100 row pairs of MFMAs), so inside lk_render_bwd every wave writes its accumulator
// tile to a partial buffer with plain coalesced stores and k_wgrad_reduce sums the tiles (no atomics, reproducible).
// The four waves of a workgroup are four consecutive UNITS of the same row chunk, and the grid walks units fastest:
// the pieces of A / A2 / B that several units need are fetched from HBM once and re-read from L1/L2.
#ifndef WG_STEPS
#define WG_STEPS 16
#endif
#define WG_CHUNK (2 * WG_STEPS)                     // rows per chunk = one ring of row pairs

template <int V> struct WgVec;
template <> struct WgVec<1> { float v[1]; };
template <> struct WgVec<2> { float v[2]; };
template <int V>
__device__ __forceinline__ WgVec<V> wg_load(const float* __restrict__ p) {
    WgVec<V> r;
    if (V == 2) { const float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[V - 1] = t.y; }
    else r.v[0] = *p;
    return r;
}

// H16 (mapper mode, LkWgradArgs::h16): the product runs on the 16-bit matrix pipe.  Eight consecutive ring slots ARE the
// operands of one v_mfma_f32_32x32x16_f16 as they stand: the lane of half h holds rows 2 j + h, j = 0..7, of its column - any
// order of the 16 reduction indices is as good as another when A and B agree - so the loaded values are only cut into fp16
// pieces (hi + lo, lk_split8h; A' times 2^10 first: unit-scale loss gradients put d h around 1e-4) and multiplied with three
// instructions of 32 cycles per 16 rows instead of eight fp32 instructions of 64 cycles, which did not overlap with the
// VALU work either.  The tile is scaled back when it is stored.
// Tile mode inside a workgroup (LkWgLds): the waves of a workgroup that work on the SAME unit (a "run", ~3 waves) leave ONE tile - the followers park
// theirs in LDS (park), the first of the run (the leader) adds them to its own behind the workgroup's one barrier, in order, and stores the sum.
// 2 048 tiles of 16.6 KB per launch were 34 MB written and read back by the reduction launch; about a third are left.  The leader reads ALL THREE
// follower slots of every element unconditionally (in[f]: a follower's tile, or any valid tile where the run is shorter - the value is then
// dropped by a select): with a loop over the run's length per element the sum was a chain of dependent LDS round trips at the very end of
// every wave's life, 6.4 us of a 48-us launch (knock-out build), now one batch of reads per block.  (Sharing the sum out over the run's waves was
// 1 us faster alone but needs all four tiles in LDS - 66 KB per workgroup, and k_feat_gather's workgroups no longer fit beside two of them.)
struct LkWgLds { float* park; const float* in[3]; int n_in; bool on; };
template <int NV, int KV, int MODE, bool H16>
__device__ __forceinline__ void wgrad_unit(const LkWgradJob& J, int n0, int k0, int c0, int stride, int lane,
                                           float* __restrict__ tile, int rows, bool aux_t = false, float dsc = 1.0f, LkWgLds wl = LkWgLds{nullptr, {nullptr, nullptr, nullptr}, 0, false}) {
    const int i = lane & 31, h = lane >> 5;
    // this lane's columns; out-of-range columns read a legal address and are zeroed (A) / never flushed (B)
    const int ncol = n0 + NV * i, kcol = k0 + KV * i;
    bool nok[NV], kok[KV];
#pragma unroll
    for (int b = 0; b < NV; ++b) nok[b] = ncol + b < J.N;
#pragma unroll
    for (int b = 0; b < KV; ++b) kok[b] = kcol + b < J.K;
    const int ncl = nok[0] ? ncol : 0;
    const float* __restrict__ pA = J.A + ncl;
    const float* __restrict__ pA2 = (MODE == 1) ? J.A2 + ncl : J.A2;
    const float* __restrict__ pB;
    int ldb;
    if (J.B3 && kcol >= J.k_split2) { pB = J.B3 + (kok[0] ? kcol - J.k_split2 : 0); ldb = J.ldb3; }
    else if (J.B2 && kcol >= J.k_split) { pB = J.B2 + (kcol - J.k_split); ldb = J.ldb2; }
    else { pB = J.B + (kok[0] ? kcol : 0); ldb = J.ldb; }
    const int lda = J.lda, lda2 = J.lda2;
    constexpr int mode = MODE;
    f32x16 acc[NV][KV];
#pragma unroll
    for (int bn = 0; bn < NV; ++bn)
#pragma unroll
        for (int bk = 0; bk < KV; ++bk) acc[bn][bk] = lk_zero16();
    float bsum[NV];
#pragma unroll
    for (int b = 0; b < NV; ++b) bsum[b] = 0.0f;
    // this wave's rows: chunks c0, c0 + stride, c0 + 2 stride, ... of WG_CHUNK = 2 * WG_STEPS rows each
    const int last = rows > 0 ? rows - 1 : 0;            // rows: J.rows, or the live prefix of a partitioned batch
    const int n_chunks = (rows + WG_CHUNK - 1) / WG_CHUNK;
    const int my_chunks = (c0 < n_chunks) ? (n_chunks - c0 + stride - 1) / stride : 0;
    WgVec<NV> ra[WG_STEPS], ra2[WG_STEPS];
    WgVec<KV> rb[WG_STEPS];
    float rw[WG_STEPS];
    auto fetch1 = [&](int s, int n) {              // slot s <- row pair s of this wave's n-th chunk (clamped past the end)
        int row = (c0 + n * stride) * WG_CHUNK + 2 * s + h;
        row = (row < last && n < my_chunks) ? row : last;
        ra[s] = wg_load<NV>(pA + (size_t)row * lda);
        if (mode == 1) ra2[s] = wg_load<NV>(pA2 + (size_t)row * lda2);
        if (mode == 2) rw[s] = pA2[row];
        rb[s] = wg_load<KV>(pB + (size_t)row * ldb);
    };
    // Register ring of WG_STEPS row pairs = one chunk: slot s is consumed (its 3 loads are the oldest outstanding ones)
    // and immediately refilled with the same pair of the wave's NEXT chunk, so WG_STEPS - 1 pairs are always in
    // flight behind the MFMAs.  Refills past the end re-read the last row (clamped address) and are masked when consumed.
#pragma unroll
    for (int s = 0; s < WG_STEPS; ++s) { fetch1(s, 0); __builtin_amdgcn_sched_barrier(0); }    // issue in slot order
    const float SCALE = H16 ? 1024.0f * dsc : 1.0f, ISCALE = H16 ? (1.0f / 1024.0f) / dsc : 1.0f;       // dsc: LkWgradArgs::dscale, a power of two
    for (int n = 0; n < my_chunks; ++n) {
        const int row0 = (c0 + n * stride) * WG_CHUNK;
        if (H16) {
#pragma unroll
            for (int g8 = 0; g8 < WG_STEPS / 8; ++g8) {
                LkH8 ap[NV], bp[KV];
#ifndef WGK_NOMATH
#pragma unroll
                for (int b = 0; b < NV; ++b) {
                    float v8[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int sl = 8 * g8 + jj;
                        const bool ok = row0 + 2 * sl + h < rows;
                        float f = SCALE;
                        if (mode == 1) f = SCALE * lk_softplus100_grad_from_out(ra2[sl].v[b]);
                        if (mode == 2) f = SCALE * rw[sl];
                        v8[jj] = (ok && nok[b]) ? ra[sl].v[b] * f : 0.0f;
                        bsum[b] += v8[jj];
                    }
                    ap[b] = lk_split8h(v8);
                }
#pragma unroll
                for (int b = 0; b < KV; ++b) {
                    float v8[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) v8[jj] = rb[8 * g8 + jj].v[b];
                    bp[b] = lk_split8h(v8);
                }
#endif
#ifdef WGK_NOMATH
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
#pragma unroll
                    for (int b = 0; b < NV; ++b) acc[b][0][jj] += ra[8 * g8 + jj].v[b];
#pragma unroll
                    for (int b = 0; b < KV; ++b) acc[0][b][8 + jj] += rb[8 * g8 + jj].v[b];
                }
#else
#pragma unroll
                for (int bn = 0; bn < NV; ++bn)
#pragma unroll
                    for (int bk = 0; bk < KV; ++bk) acc[bn][bk] = lk_mma3h(ap[bn], bp[bk], acc[bn][bk]);
#endif
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) { fetch1(8 * g8 + jj, n + 1); __builtin_amdgcn_sched_barrier(0); }
            }
            continue;
        }
#pragma unroll
        for (int s = 0; s < WG_STEPS; ++s) {
            const bool ok = row0 + 2 * s + h < rows;
            float av[NV], bv[KV];
#pragma unroll
            for (int b = 0; b < NV; ++b) {
                float f = 1.0f;
                if (mode == 1) f = lk_softplus100_grad_from_out(ra2[s].v[b]);
                if (mode == 2) f = rw[s];
                av[b] = (ok && nok[b]) ? ra[s].v[b] * f : 0.0f;
                bsum[b] += av[b];
            }
#pragma unroll
            for (int b = 0; b < KV; ++b) bv[b] = rb[s].v[b];
#pragma unroll
            for (int bn = 0; bn < NV; ++bn)
#pragma unroll
                for (int bk = 0; bk < KV; ++bk) acc[bn][bk] = lk_mfma(av[bn], bv[bk], acc[bn][bk]);
            fetch1(s, n + 1);
            __builtin_amdgcn_sched_barrier(0);          // keep consume(s) -> refill(s) order: the ring IS the schedule
        }
    }
    // tile mode: a follower parks element idx in LDS, the leader (or a wave alone) adds the run's and stores
    if (tile) {
        const bool follower = wl.park != nullptr;
        if (wl.on && !follower) __syncthreads();              // the followers' tiles are in LDS
        auto put = [&](int idx, float v) {
            if (follower) { wl.park[idx] = v; return; }
            if (wl.on) {
                const float v0 = wl.in[0][idx], v1 = wl.in[1][idx], v2 = wl.in[2][idx];
                v += (wl.n_in > 0) ? v0 : 0.0f;
                v += (wl.n_in > 1) ? v1 : 0.0f;
                v += (wl.n_in > 2) ? v2 : 0.0f;
            }
            tile[idx] = v;
        };
        if (aux_t && KV == 1) {
            // the auxiliary columns (M = d y^T c, LkFcPost): stored TRANSPOSED, [column kc][row n of the unit], because their only reader
            // (fc_post_body) sums ONE column over the tiles - in the lane-major layout below that was one float out of every 128-byte line,
            // and the 33 column blocks of a layer fetched the same lines 33 times (86 MB per launch for 2.6 MB of tiles)
#pragma unroll
            for (int bn = 0; bn < NV; ++bn)
#pragma unroll
                for (int r = 0; r < 16; ++r) put(i * 64 + NV * lk_frag_row(r, h) + bn, acc[bn][0][r] * ISCALE);
        } else {       // partial tile [block (bn,bk)][register r][lane] + bias sums: 256-byte coalesced stores
#pragma unroll
            for (int bn = 0; bn < NV; ++bn)
#pragma unroll
                for (int bk = 0; bk < KV; ++bk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) put(((bn * KV + bk) * 16 + r) * 64 + lane, acc[bn][bk][r] * ISCALE);
            if (k0 == 0) {
#pragma unroll
                for (int b = 0; b < NV; ++b) {
                    const float v = (bsum[b] + __shfl_xor(bsum[b], 32)) * ISCALE;
                    if (h == 0) put(4 * 16 * 64 + NV * i + b, v);
                }
            }
        }
        if (wl.on && follower) __syncthreads();
        return;
    }
    // atomic flush: block (bn, bk), register r of lane (j = lane&31, h): n = n0 + NV*frag_row(r,h) + bn, k = k0 + KV*j + bk
#pragma unroll
    for (int bn = 0; bn < NV; ++bn)
#pragma unroll
        for (int bk = 0; bk < KV; ++bk) {
            if (!kok[bk] || (J.k_aux && kcol + bk >= J.k_aux)) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + NV * lk_frag_row(r, h) + bn;
                if (nn < J.N) atomicAdd(J.dW + (size_t)nn * J.ldw + kcol + bk, acc[bn][bk][r] * ISCALE);
            }
        }
    if (J.db && k0 == 0) {
#pragma unroll
        for (int b = 0; b < NV; ++b) {
            const float v = (bsum[b] + __shfl_xor(bsum[b], 32)) * ISCALE;
            if (h == 0 && nok[b]) atomicAdd(J.db + ncol + b, v);
        }
    }
}

template <int NV, int KV, bool H16>
__device__ __forceinline__ void wgrad_unit_mode(const LkWgradJob& J, int n0, int k0, int c0, int stride, int lane, float* tile, int rows, float dsc, LkWgLds wl) {
    const bool aux_t = J.k_aux > 0 && k0 >= J.k_aux;
    if (J.a_mode == 0) wgrad_unit<NV, KV, 0, H16>(J, n0, k0, c0, stride, lane, tile, rows, aux_t, dsc, wl);
    else if (J.a_mode == 1) wgrad_unit<NV, KV, 1, H16>(J, n0, k0, c0, stride, lane, tile, rows, aux_t, dsc, wl);
    else wgrad_unit<NV, KV, 2, H16>(J, n0, k0, c0, stride, lane, tile, rows, aux_t, dsc, wl);
}
// tile y = 8 jl + x of a unit was stored iff its wave led a run: the first wave of the unit on its XCD, or the first wave of a workgroup
// (wave slots are handed out four to a workgroup: slot & 3 == 0)
__device__ __forceinline__ bool lk_wg_tile_stored(const LkWgradUnit& U, int y) { const int jl = y >> 3; return jl == 0 || ((U.wave0 + jl) & 3) == 0; }

// XCD-AWARE ROW OWNERSHIP.  Several units read the same rows (the 64-column pieces of one operand, or two jobs sharing
// d h), and every XCD has its own L2: with a unit's waves spread over the chip a row chunk was fetched from HBM by up to
// eight L2s (417 MB fetched per launch for 185 MB of operands).  Block b runs on XCD b % 8, so chunk c of the rows is
// handled ON XCD c % 8 BY EVERY UNIT: each XCD gets the same partition of its 256 wave slots into units (waves in
// proportion to the unit's work), local wave jl of a unit takes the chunks x + 8 (jl + k W), k = 0, 1, ...  All waves of
// the launch are co-resident (two per SIMD) and sweep the rows at the same speed, so the re-reads hit the XCD's L2.
template <bool H16>
__global__ __launch_bounds__(256) void k_wgrad(LkWgradArgs a) {
    __shared__ float s_tile[3][LK_WG_TILE];          // the followers' tiles of the workgroup's runs (tile mode): wave w parks in s_tile[w - 1]
    const int lane = lk_lane();
    const int x = lk_uniform((int)blockIdx.x & 7);
    const int w = lk_uniform((int)threadIdx.x >> 6);
    const int l = lk_uniform(((int)blockIdx.x >> 3) * 4 + w);      // wave slot inside the XCD
    const bool tiles = a.part != nullptr;
    if (l >= a.n_waves) { if (tiles) __syncthreads(); return; }       // (every wave of a tile-mode workgroup meets the one barrier)
    // the two device-side scalars of the launch first: their round trips run beside the unit look-up (they were two dependent loads behind it)
    const int live_rays = a.live_rays ? lk_uniform(*a.live_rays) : 0;
    const float dsc = (H16 && a.dscale) ? *a.dscale : 1.0f;
    int u = 0;                                                          // the unit whose slots [wave0, wave0 + n_waves) hold l: bisection over <= 48 entries
    for (int step = 32; step > 0; step >>= 1)
        if (u + step < a.n_units && l >= a.unit[u + step].wave0) u += step;
    const LkWgradUnit& U = a.unit[u];
    const LkWgradJob& J = a.job[U.job];
    const int jl = l - U.wave0;
    const int c0 = x + 8 * jl, stride = 8 * U.n_waves;
    // the run of this unit's waves inside the workgroup: its first wave (the unit's first on this XCD, or the workgroup's wave 0) leads
    LkWgLds wl{nullptr, {s_tile[0], s_tile[0], s_tile[0]}, 0, tiles};
    const bool leader = jl == 0 || w == 0;
    if (tiles && !leader) wl.park = s_tile[w - 1];
    if (tiles && leader) {
        int nf = 0;
        while (w + 1 + nf < 4 && jl + 1 + nf < U.n_waves && l + 1 + nf < a.n_waves) ++nf;
        wl.n_in = nf;                                                   // followers w + 1 .. w + nf parked in s_tile[w] .. s_tile[w + nf - 1]
        for (int f = 0; f < 3; ++f) wl.in[f] = s_tile[f < nf ? w + f : 0];
    }
    // a wave without a chunk (tiny problems) still contributes its (zero) tile
    float* tile = tiles ? a.part + ((size_t)8 * U.wave0 + 8 * jl + x) * LK_WG_TILE : nullptr;
    // rows behind the live prefix of a partitioned batch were not written by their producers (k_decode_bwd skips those tiles)
    const int rows = a.live_rays ? min(J.rows, live_rays * a.S) : J.rows;
    if (!tile && c0 >= (rows + WG_CHUNK - 1) / WG_CHUNK) return;
    if (U.nv == 2 && U.kv == 2) wgrad_unit_mode<2, 2, H16>(J, U.n0, U.k0, c0, stride, lane, tile, rows, dsc, wl);
    else if (U.nv == 2) wgrad_unit_mode<2, 1, H16>(J, U.n0, U.k0, c0, stride, lane, tile, rows, dsc, wl);
    else if (U.kv == 2) wgrad_unit_mode<1, 2, H16>(J, U.n0, U.k0, c0, stride, lane, tile, rows, dsc, wl);
    else wgrad_unit_mode<1, 1, H16>(J, U.n0, U.k0, c0, stride, lane, tile, rows, dsc, wl);
}

// dW += sum over the unit's waves of the partial tiles (tile order: contiguous reads; every output element is owned by
// exactly one thread, so the read-modify-write of dW needs no atomics and the result is run-to-run reproducible).
__device__ __forceinline__ void wgrad_reduce_body(const LkWgradArgs& a, int bx, int by, float (*sh)[32], const LkStepRider& sr) {
    const LkWgradUnit& U = a.unit[bx];
    const LkWgradJob& J = a.job[U.job];
    const int nv = U.nv, kv = U.kv;
    if (J.k_aux > 0 && U.k0 >= J.k_aux) return;          // auxiliary columns: not part of dW, summed by fc_post_body (transposed tiles)
    // 32 consecutive tile elements x 8 row-block lanes per workgroup
    const int e = (int)threadIdx.x & 31, q = (int)threadIdx.x >> 5;
    const int idx = by * 32 + e;
    const bool in_acc = idx < nv * kv * 1024, in_bias = idx >= 4 * 16 * 64 && idx < 4 * 16 * 64 + 32 * nv;
    if (!in_acc && !in_bias && by * 32 + 31 >= nv * kv * 1024 && by * 32 < 4 * 16 * 64) return;   // unused blocks of a narrow unit
    const int nblk = 8 * U.n_waves;                                    // every wave of the unit stored a tile
    const size_t stride = LK_WG_TILE;
    const float* __restrict__ src = a.part + (size_t)8 * U.wave0 * LK_WG_TILE + idx;
    float s = 0.0f;
    if (in_acc || in_bias) {
#pragma unroll 4
        for (int y = q; y < nblk; y += 8) s += lk_wg_tile_stored(U, y) ? src[(size_t)y * stride] : 0.0f;      // (the other waves' tiles went into their leader's)
    }
    sh[q][e] = s;
    __syncthreads();
    if (q != 0) return;
    s = ((sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e])) + ((sh[4][e] + sh[5][e]) + (sh[6][e] + sh[7][e]));
    if (in_acc) {
        const int blk = idx >> 10, r = (idx >> 6) & 15, lane = idx & 63, h = lane >> 5, j = lane & 31;
        const int bn = blk / kv, bk = blk - bn * kv;
        const int nn = U.n0 + nv * lk_frag_row(r, h) + bn, k = U.k0 + kv * j + bk;
        if (nn < J.N && k < (J.k_aux ? J.k_aux : J.K)) grad_commit(J.dW + (size_t)nn * J.ldw + k, s, sr);     // auxiliary columns: fc_post_body
    } else if (in_bias) {
        const int tcol = idx - 4 * 16 * 64;
        if (J.db && U.k0 == 0 && U.n0 + tcol < J.N) grad_commit(J.db + U.n0 + tcol, s, sr);
    }
}
__global__ __launch_bounds__(256) void k_wgrad_reduce(LkWgradArgs a) {
    __shared__ float sh[8][32];
    LkStepRider none; none.n_span = 0;
    wgrad_reduce_body(a, (int)blockIdx.x, (int)blockIdx.y, sh, none);
}

// fc_c gradients from the auxiliary columns (lk_kernels.h LkFcPost): block (f, kc) sums the partial tiles of column kc of
// M = d y^T c (block kc = 32: of db = sum d y) of job fc[f].src_job itself - the tile sums of the same launch are not visible
// yet - and adds W^T M[:, kc] (W^T db): one owner per output element, no atomics, reproducible.  The block is a chain of two
// memory round trips (the tiles sit 16 KB apart, W's column walks its rows), so every load of a phase is issued before the first
// is used: thread (e, g) sums the tiles y = g, g + 4, ... of element e; thread (v, uh) takes 64 rows of W.
#define LK_FC_POST_COLS 33
#define LK_FC_MAX_TILES 16            // loads in flight per thread: 8 n_waves <= 64 tiles of a unit over four thread groups
struct LkFcPostLds { float part[4][128]; float m[128]; float half[2][128]; };
__device__ __forceinline__ void fc_post_body(const LkWgradArgs& a, int f, int kc, LkFcPostLds& sh, const LkStepRider& sr) {
    const LkFcPost& F = a.fc[f];
    const LkWgradJob& J = a.job[F.src_job];
    const int t = (int)threadIdx.x;
    const bool bias = kc == 32;
    // phase 1: m[n] = column kc of M (or db), n = 0 .. 127
    {
        const int g = t >> 6, e = t & 63;                       // e = (bn, r, h): row n0 + nv * frag_row(r, h) + bn of the unit
        float s2[2] = {0.0f, 0.0f};
        int which = 0;
        for (int u = 0; u < a.n_units && which < 2; ++u) {
            const LkWgradUnit& U = a.unit[u];
            if (U.job != F.src_job || U.k0 != (bias ? 0 : J.k_aux)) continue;
            const int nblk = 8 * U.n_waves, nv = U.nv;
            const float* __restrict__ src = a.part + (size_t)8 * U.wave0 * LK_WG_TILE;
            // bias: element e = column e of the tile's bias section (row n0 + e); column kc of M: the transposed tile [kc][row e of the unit]
            const int off = bias ? 4 * 16 * 64 + e : kc * 64 + e;
            float s = 0.0f;
            for (int y0 = 0; y0 < nblk; y0 += 4 * LK_FC_MAX_TILES) {          // one pass at the benchmark's sizes
                float v[LK_FC_MAX_TILES];
#pragma unroll
                for (int q = 0; q < LK_FC_MAX_TILES; ++q) {
                    const int y = y0 + g + 4 * q;
                    v[q] = (y < nblk && e < 32 * nv && lk_wg_tile_stored(U, y)) ? src[(size_t)y * LK_WG_TILE + off] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < LK_FC_MAX_TILES; ++q) s += v[q];
            }
            s2[which++] = s;
        }
        sh.part[g][e] = s2[0]; sh.part[g][64 + e] = s2[1];
    }
    __syncthreads();
    if (t < 128) {
        // units in table order: n0 = 0 then n0 = 64 (nv = 2), or the single nv = 1 unit of the output job
        const int which = t >> 6, e = t & 63;
        const float s = ((sh.part[0][t] + sh.part[1][t]) + sh.part[2][t]) + sh.part[3][t];
        const int nv = J.N > 32 ? 2 : 1;
        const int n = 64 * which + e;
        const bool ok = (nv == 2 || (which == 0 && e < 32)) && n < 128;
        if (ok) sh.m[n] = (n < J.N) ? s : 0.0f;
    }
    if (J.N <= 32 && t >= 32 && t < 128) sh.m[t] = 0.0f;
    __syncthreads();
    // phase 2: out[v] += sum_u W[u][off + v] m[u]
    const int v = t & 127, uh = t >> 7;
    const int u0 = uh * 64, u1 = F.rows_u < u0 + 64 ? F.rows_u : u0 + 64;
    const float* __restrict__ Wv = F.W + F.off + v;
    float w[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) w[q] = (u0 + q < u1) ? Wv[(size_t)(u0 + q) * F.ldw] : 0.0f;
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < 64; ++q) acc = fmaf(w[q], sh.m[u0 + q], acc);
    sh.half[uh][v] = acc;
    __syncthreads();
    if (t < 128) {
        const float r = sh.half[0][t] + sh.half[1][t];
        if (bias) grad_commit(F.du + t, r, sr); else grad_commit(F.dU + (size_t)t * 32 + kc, r, sr);
    }
}
__global__ __launch_bounds__(256) void k_fc_post(LkWgradArgs a) {
    __shared__ LkFcPostLds sh;
    LkStepRider none; none.n_span = 0;
    fc_post_body(a, (int)blockIdx.x / LK_FC_POST_COLS, (int)blockIdx.x % LK_FC_POST_COLS, sh, none);
}

// column sums of a partial table [n_parts][width] into out[width] (+=), 32 columns per workgroup, one owner per column
__device__ __forceinline__ void col_reduce_body(const float* __restrict__ part, int n_parts, int width, float* __restrict__ out, int bx, float (*sh)[32],
                                                const LkStepRider& sr) {
    const int e = (int)threadIdx.x & 31, q = (int)threadIdx.x >> 5;
    const int col = bx * 32 + e;
    float s = 0.0f;
    if (col < width) {
#pragma unroll 4
        for (int y = q; y < n_parts; y += 8) s += part[(size_t)y * width + col];
    }
    sh[q][e] = s;
    __syncthreads();
    if (q == 0 && col < width) grad_commit(out + col, ((sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e])) + ((sh[4][e] + sh[5][e]) + (sh[6][e] + sh[7][e])), sr);
}

// EVERY partial-sum reduction of a mapper 'color' backward in one launch after the two streams have joined (they were five
// launches of 5-10 us each on the critical path): the k_wgrad tiles, the linear1 / linear2 tiles of the fused rel-pos variant,
// the Fourier-matrix partials of the two decoders.  Blocks [0, b_wg) | [b_wg, b_rp) | [b_rp, b_pg) | [b_pg, b_pr).
__global__ __launch_bounds__(256) void k_bwd_reduce(LkWgradArgs wa, LkBwdReduceArgs r, LkStepRider step) {
    __shared__ float sh[8][32];
    __shared__ LkFcPostLds sh_fc;
    const LkStepRider& sr = step;
    // the fc_c blocks are two dependent memory round trips long: first in the grid, or they start when the tile sums retire
    const int n_fc = r.b_fc - r.b_pr;
    if ((int)blockIdx.x < n_fc) { fc_post_body(wa, (int)blockIdx.x / LK_FC_POST_COLS, (int)blockIdx.x % LK_FC_POST_COLS, sh_fc, sr); return; }
    const int b = (int)blockIdx.x - n_fc;
    if (b < r.b_wg) wgrad_reduce_body(wa, b / r.ny, b % r.ny, sh, sr);
    else if (b < r.b_rp) rp_reduce_body(r.part1, r.n1, r.part2, r.n2, r.dW1, r.db1, r.dW2, r.db2, b - r.b_wg, sh, sr);
    else if (b < r.b_pg) col_reduce_body(r.part_bg, r.n_bg, 288, r.out_bg, b - r.b_rp, sh, sr);
    else if (b < r.b_pr) col_reduce_body(r.part_br, r.n_br, 32, r.out_br, b - r.b_pg, sh, sr);
    else {      // feature-row Adam segments (final since the gather): the step rider's extra blocks
        const int q = b - r.b_pr;
        if (q < step.feat_gx) lk_adam_seg_block(step.feat[0], step.beta1, step.beta2, step.eps, q, step.feat_gx);
        else lk_adam_seg_block(step.feat[1], step.beta1, step.beta2, step.eps, q - step.feat_gx, step.feat_gx);
    }
}
// feat_rows false: without the step rider's feature-row blocks (the trunk's half of a split step)
int lk_launch_bwd_reduce(const LkWgradArgs& wa, LkBwdReduceArgs r, bool with_rp, hipStream_t st, const LkStepRider* step, bool feat_rows) {
    r.ny = lk_cdiv(LK_WG_TILE, 32);
    r.b_wg = wa.part && wa.n_units > 0 ? wa.n_units * r.ny : 0;
    r.b_rp = r.b_wg + (with_rp ? LK_RP_REDUCE_BLOCKS : 0);
    r.b_pg = r.b_rp + (r.part_bg ? lk_cdiv(288, 32) : 0);
    r.b_pr = r.b_pg + (r.part_br ? 1 : 0);
    r.b_fc = r.b_pr + (r.b_wg > 0 ? wa.n_fc * LK_FC_POST_COLS : 0);
    LkStepRider sr;
    if (step) sr = *step; else sr = LkStepRider{};
    r.b_ad = r.b_fc + ((sr.n_span > 0 && feat_rows) ? sr.n_feat * sr.feat_gx : 0);
    if (r.b_ad > 0) hipLaunchKernelGGL(k_bwd_reduce, dim3(r.b_ad), dim3(256), 0, st, wa, r, sr);
    return LK_OK;
}

// h16: d c_col comes from the colour trunk's pre-scaled fp16 chain with unit-scale loss gradients and no exposure affine in between
// (relpos_bwd_wave<.., true>)
int lk_launch_relpos_interp_bwd(const LkRelposBwdArgs& rb, const LkInterpBwdArgs& ib, hipStream_t st, bool h16) {
    LkProfScope prof_(LKK_RELPOS_BWD, st);
    const dim3 grid(lk_cdiv(rb.P, 32));
    if (rb.flags & LK_FLAG_FEATS_F16) {
        if (h16) hipLaunchKernelGGL((k_relpos_interp_bwd<true, true>), grid, dim3(512), 0, st, rb, ib);
        else hipLaunchKernelGGL((k_relpos_interp_bwd<true, false>), grid, dim3(512), 0, st, rb, ib);
    } else {
        if (h16) hipLaunchKernelGGL((k_relpos_interp_bwd<false, true>), grid, dim3(512), 0, st, rb, ib);
        else hipLaunchKernelGGL((k_relpos_interp_bwd<false, false>), grid, dim3(512), 0, st, rb, ib);
    }
    return LK_OK;
}
int lk_launch_interp_bwd(const LkInterpBwdArgs& a, hipStream_t st, const ExposureStepArgs* xstep, const float* xstep_part, int xstep_n_part) {
    LkProfScope prof_(LKK_INTERP_BWD, st);
    if (xstep) hipLaunchKernelGGL(k_interp_bwd_x, dim3(lk_cdiv(a.P, 32) + 1), dim3(256), 0, st, a, *xstep, xstep_part, xstep_n_part);
    else hipLaunchKernelGGL(k_interp_bwd, dim3(lk_cdiv(a.P, 32)), dim3(256), 0, st, a);
    return LK_OK;
}
int lk_launch_feat_scatter(const LkFeatScatterArgs& a, hipStream_t st) {
    LkProfScope prof_(LKK_FEAT_SCATTER, st);
    LkFeatScatterArgs b = a;
    b.red_block0 = lk_cdiv((long long)a.P * LK_K, 8 * LK_GATHER_CHUNK);
    if (!b.dw2_part) b.dw2_blocks = 0;
    hipLaunchKernelGGL(k_feat_gather, dim3((b.x_on ? 1 : 0) + b.dw2_blocks + b.red_block0 + (a.red_part ? lk_cdiv(a.red_width, 32) : 0)), dim3(256), 0, st, b);
    return LK_OK;
}
int lk_launch_seg_sort(const LkFeatScatterArgs& a, bool counted, hipStream_t st, int batch) {
    const int nb = lk_cdiv((long long)a.P * LK_K, 256);
    // seg_cnt is zero on entry: the scan of the previous sort cleared what that sort had counted
    if (!counted) hipLaunchKernelGGL(k_seg_count, dim3(nb, batch), dim3(256), 0, st, a);
    lk_launch_scan_i32(a.seg_cnt, a.seg_off, a.seg_sums, a.N + 1, st, batch, a.cnt_stride, a.sums_stride);
    hipLaunchKernelGGL(k_seg_place, dim3(nb, batch), dim3(256), 0, st, a);
    return LK_OK;
}
int lk_launch_rays_bwd(const LkRaysBwdArgs& a, hipStream_t st) {
    LkProfScope prof_(LKK_RAYS_BWD, st);
    hipLaunchKernelGGL(k_rays_bwd, dim3(lk_cdiv(a.R, 256)), dim3(256), 0, st, a);
    return LK_OK;
}
int lk_relpos_bwd_parts(int P) { const int n = lk_cdiv(lk_cdiv(P, 4), 4); return n < LK_RPF_MAX_PARTS ? n : LK_RPF_MAX_PARTS; }
int lk_launch_relpos_bwd(const LkRelposBwdArgs& a, hipStream_t st) {
    LkProfScope prof_(LKK_RELPOS_BWD, st);
    const int waves = lk_cdiv(a.P, 4);
    if (lk_relpos_fused(a.flags)) {
        const bool wg = !(a.flags & LK_FLAG_EMBED_GRADS_ONLY), f16 = (a.flags & LK_FLAG_FEATS_F16) != 0;
        const dim3 grid(lk_relpos_bwd_parts(a.P));
        if (wg && f16) hipLaunchKernelGGL((k_relpos_bwd_fused<true, true>), grid, dim3(256), 0, st, a);
        else if (wg) hipLaunchKernelGGL((k_relpos_bwd_fused<true, false>), grid, dim3(256), 0, st, a);
        else if (f16) hipLaunchKernelGGL((k_relpos_bwd_fused<false, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_relpos_bwd_fused<false, false>), grid, dim3(256), 0, st, a);
    }
    else if (a.flags & LK_FLAG_FEATS_F16) hipLaunchKernelGGL(k_relpos_bwd<true>, dim3(lk_cdiv(waves, 4)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_relpos_bwd<false>, dim3(lk_cdiv(waves, 4)), dim3(256), 0, st, a);
    return LK_OK;
}
int lk_dw2_parts(int P) { const int n = lk_cdiv(P, LK_DW2_SAMPLES); return n < 1 ? 1 : (n < LK_DW2_PARTS ? n : LK_DW2_PARTS); }
int64_t lk_dw2_part_floats(int P) { return (int64_t)lk_dw2_parts(P) * LK_DW2_TILE; }
int lk_launch_dw2_hbar(const LkRelposBwdArgs& a, float* dw2_part, hipStream_t st) {       // its tiles are summed by k_bwd_reduce
    hipLaunchKernelGGL(k_dw2_hbar, dim3(lk_dw2_parts(a.P)), dim3(256), 0, st, a.dc_col, a.w_sum, a.hbar, a.P, a.live_rays, a.S, dw2_part);
    return LK_OK;
}
int lk_occupancy_relpos_bwd_fused() {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_relpos_bwd_fused<true, false>, 256, 0);
    return n;
}
int lk_occupancy_wgrad() {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_wgrad<true>, 256, 0);
    return n;
}
int lk_launch_wgrad(const LkWgradArgs& a_in, int max_rows, hipStream_t st, LkWgradArgs* deferred) {
    if (deferred) deferred->n_units = 0;
    if (a_in.n_jobs == 0 || max_rows <= 0) return LK_OK;
    if (a_in.n_fc > 0 && !a_in.part) return LK_ERR_ARG;     // the auxiliary columns exist as partial tiles only
    LkWgradArgs a = a_in;
    // cut every job into (N piece) x (K piece) units of 32 or 64 columns
    a.n_units = 0;
    double work[LK_WGRAD_MAX_UNITS], total = 0.0;
    for (int j = 0; j < a.n_jobs; ++j) {
        const LkWgradJob& J = a.job[j];
        for (int n0 = 0; n0 < J.N; n0 += 64) {
            const int nv = (J.N - n0 > 32) ? 2 : 1;
            const int k_main = J.k_aux ? J.k_aux : J.K;     // the auxiliary columns are a k piece of their own (fc_post_body finds it by k0)
            for (int k0 = 0; k0 < J.K; k0 += (k0 < k_main && k0 + 64 > k_main) ? k_main - k0 : 64) {
                const int k_end = k0 < k_main ? k_main : J.K;
                const int kv = (k_end - k0 > 32) ? 2 : 1;
                if (a.n_units >= LK_WGRAD_MAX_UNITS) return LK_ERR_ARG;
                LkWgradUnit& U = a.unit[a.n_units];
                U.job = j; U.n0 = n0; U.k0 = k0; U.nv = nv; U.kv = kv;
                // cost of a row pair: the MFMAs plus about one MFMA's worth of loads / element-wise work
                // cost of a row pair = the columns the wave loads (measured: with the wave slots in proportion to nv * kv + 0.5,
                // "the matrix instructions", the narrow units trailed the sweep of the wide ones that read the same rows, their
                // re-reads missed the XCD's L2 and the launch took 62 us instead of 50)
                work[a.n_units] = (double)J.rows * (nv + kv);
                total += work[a.n_units++];
            }
        }
    }
    // wave slots PER XCD (LK_WG_MAX_WAVES / 8 = two per SIMD, all co-resident) in proportion to the unit's work, at most
    // one per 8 x 32-row chunks (chunk c lives on XCD c % 8, see k_wgrad); wave0 / n_waves are per-XCD numbers
    int next = 0;
    for (int u = 0; u < a.n_units; ++u) {
        LkWgradUnit& U = a.unit[u];
        const int n_chunks_x = lk_cdiv(lk_cdiv(a.job[U.job].rows, WG_CHUNK), 8);
        int w = (int)((LK_WG_MAX_WAVES / 8 - a.n_units) * (work[u] / total)) + 1;
        if (w > n_chunks_x) w = n_chunks_x > 0 ? n_chunks_x : 1;
        U.wave0 = next; U.n_waves = w;
        next += w;
    }
    a.n_waves = next;
    {
        LkProfScope prof_(LKK_WGRAD, st);                              // timing scope = k_wgrad alone (as rocprof reports it)
        if (a.h16) hipLaunchKernelGGL(k_wgrad<true>, dim3(8 * lk_cdiv(a.n_waves, 4)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_wgrad<false>, dim3(8 * lk_cdiv(a.n_waves, 4)), dim3(256), 0, st, a);
    }
    if (deferred) { *deferred = a; return LK_OK; }     // the tiles are summed later (k_bwd_reduce)
    if (a.part) hipLaunchKernelGGL(k_wgrad_reduce, dim3(a.n_units, lk_cdiv(LK_WG_TILE, 32)), dim3(256), 0, st, a);
    if (a.part && a.n_fc > 0) hipLaunchKernelGGL(k_fc_post, dim3(a.n_fc * LK_FC_POST_COLS), dim3(256), 0, st, a);
    return LK_OK;
}
