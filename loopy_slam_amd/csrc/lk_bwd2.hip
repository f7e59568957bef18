// Backward, part 2:
//   k_interp_bwd  d c_geo / d c_col -> scatter-add into the feature-row gradients; tracker mode: gradient
//                 through the interpolation weights to the sample position (decoder.py:191-229)
//   k_rays_bwd    d p -> d rays_o, d rays_d
//   k_relpos_bwd  backward of the relative-position neighbour MLP (decoder.py:477-488)
//   k_wgrad       all decoder weight gradients as streamed MFMA reductions over the sample rows
#include "lk_common.h"
#include "lk_kernels.h"

using namespace lkw;

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_interp_bwd(LkInterpBwdArgs a) {
    const int sub = (int)threadIdx.x & 7;
    const int p_raw = blockIdx.x * 32 + ((int)threadIdx.x >> 3);
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
            const float4 g = *reinterpret_cast<const float4*>(a.geo_feats + (size_t)id[j] * LK_C + sub * 4);
            part = dcg.x * g.x + dcg.y * g.y + dcg.z * g.z + dcg.w * g.w;
            if (do_col) {
                const float4 c = *reinterpret_cast<const float4*>(a.col_feats + (size_t)id[j] * LK_C + sub * 4);
                part += dcc.x * c.x + dcc.y * c.y + dcc.z * c.z + dcc.w * c.w;
            }
        }
        part += __shfl_xor(part, 1); part += __shfl_xor(part, 2); part += __shfl_xor(part, 4);
        if (color && relpos && a.dw_rel && has) part += a.dw_rel[(size_t)pidx * LK_K + j];
        dwn[j] = part;
    }
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
}

// Feature-gradient scatter: one half-wave (32 lanes = the 32 channels = one 128-B line) per (sample, neighbour),
// so every wave-wide atomic touches two full lines instead of up to 64 scattered ones (measured 2x-5x faster).
//   geometry rows:            g_geo[idx] += w * d c_geo[sample]
//   colour rows, no rel-pos:  g_col[idx] += w * d c_col[sample]
//   colour rows, rel-pos:     g_col[idx] += d feat[sample, neighbour]   (from k_relpos_bwd)
__global__ __launch_bounds__(256) void k_feat_scatter(LkFeatScatterArgs a) {
    const long long row = (long long)blockIdx.x * 8 + ((int)threadIdx.x >> 5);
    const int c = (int)threadIdx.x & 31;
    if (row >= (long long)a.P * LK_K) return;
    const int s = (int)(row >> 3);
    const int idx = a.nbr_idx[row];
    const float w = a.nbr_w[row];
    if (idx < 0 || w == 0.0f || a.nbr_count[s] < a.min_nn) return;
    atomicAdd(a.g_geo_feats + (size_t)idx * LK_C + c, w * a.dc_geo[(size_t)s * LK_C + c]);
    if (a.dfeat) atomicAdd(a.g_col_feats + (size_t)idx * LK_C + c, a.dfeat[(size_t)row * LK_C + c]);
    else if (a.dc_col) atomicAdd(a.g_col_feats + (size_t)idx * LK_C + c, w * a.dc_col[(size_t)s * LK_C + c]);
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

__device__ __forceinline__ void relpos_bwd_wave(const LkRelposBwdArgs& a, int sample0, float* __restrict__ part) {
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
    const float* __restrict__ F = a.Wfrag;
    const float* __restrict__ frow = a.col_feats + (size_t)idx * LK_C;
    const bool want_w = (a.flags & LK_FLAG_GRAD_WEIGHTS) != 0;
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
            const float4 v = *reinterpret_cast<const float4*>(frow + (u0 - ER));
            x0[4 * g] = v.x; x0[4 * g + 1] = v.y; x0[4 * g + 2] = v.z; x0[4 * g + 3] = v.w;
        }
    }
    x1 = lk_zero16();
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int u0 = 32 + 8 * g + 4 * h;
        if (u0 < KR) {
            const float4 v = *reinterpret_cast<const float4*>(frow + (u0 - ER));
            x1[4 * g] = v.x; x1[4 * g + 1] = v.y; x1[4 * g + 2] = v.z; x1[4 * g + 3] = v.w;
        }
    }
    f32x16 hid[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) hid[nb] = lk_zero16();
    lk_gemm_frag<4, 4>(hid, F + FM20_FWD, 4, 0, 0, x0, lane);
    lk_gemm_frag<4, 3>(hid, F + FM20_FWD, 4, 4, 0, x1, lane);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        lk_add_rowvec(hid[nb], W + R_B1, nb * 32, lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) hid[nb][q] = lk_softplus100(hid[nb][q]);
    }
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
        out[0] = lk_zero16();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) lk_gemm_frag<1, 4>(out, F + FM21_FWD, 1, 4 * kb, 0, hid[kb], lane);
        lk_add_rowvec(out[0], W + R_B2, 0, lane);
        float part = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) part = fmaf(dout[0][q], out[0][q], part);
        part += __shfl_xor(part, 32);
        if (live && h == 0) a.dw_rel[(size_t)sp * LK_K + nb_i] = (has && a.nbr_idx[(size_t)sp * LK_K + nb_i] >= 0) ? part : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) dout[0][q] *= wgt;
    // ---- d hid = (W2^T d out) * softplus'(hid)
    f32x16 dhid[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) dhid[kb] = lk_zero16();
    lk_gemm_frag<4, 4>(dhid, F + FM21_TR, 4, 0, 0, dout[0], lane);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) dhid[nb][q] *= lk_softplus100_grad_from_out(hid[nb][q]);
    if (want_w) {
        // operands of the streamed weight-gradient reductions.
        //   linear1: rows [8P][192] = d hid (128) | x (64)
        //   linear2: dW2 = sum_rows (w d c) hid^T = sum_samples d c (sum_j w_j hid_j)^T  -> only the per-SAMPLE weighted
        //            hidden vector Hbar [P][128] and the per-sample weight sum are needed (8x fewer rows, no hid rows)
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
        {
        float wsum = wgt;
        wsum += __shfl_xor(wsum, 1); wsum += __shfl_xor(wsum, 2); wsum += __shfl_xor(wsum, 4);
        if (live && h == 0 && nb_i == 0) a.w_sum[sp] = wsum;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float sred = wgt * hid[nb][4 * g + t];
                    sred += __shfl_xor(sred, 1); sred += __shfl_xor(sred, 2); sred += __shfl_xor(sred, 4);
                    v[t] = sred;
                }
                if (live && nb_i == 0)
                    *reinterpret_cast<float4*>(a.hbar + (size_t)sp * 128 + nb * 32 + 8 * g + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    // ---- d x = W1^T d hid   (virtual 64 input units: 0..19 embedding, 20..51 feature channels)
    f32x16 dx[2];
    dx[0] = lk_zero16(); dx[1] = lk_zero16();
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) lk_gemm_frag<2, 4>(dx, F + FM20_TR, 2, 4 * nb, 0, dhid[nb], lane);
    float dax = 0.0f, day = 0.0f, daz = 0.0f;        // d loss / d (x_I - p), this lane's share
    const float* B = W + R_EB;
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
                    if (is_emb) {
                        const float x = lk_fourier_arg(a0, a1, a2, b0, b1, b2);
                        const float f = (u < 10) ? lk_cosf(x) : -lk_sinf(x);
                        gx = dx[tile][4 * g + t] * f;
                    }
                    if (want_p) {
                        dax = fmaf(gx * LK_TWO_PI, b0, dax); day = fmaf(gx * LK_TWO_PI, b1, day); daz = fmaf(gx * LK_TWO_PI, b2, daz);
                    }
                    if (want_w) {                        // every lane takes part in the reductions (convergent shuffles)
                        const float s0 = lk_half_wave_sum(gx * a0), s1 = lk_half_wave_sum(gx * a1), s2 = lk_half_wave_sum(gx * a2);
                        if ((lane & 31) == 0 && is_emb) {   // two units (sin, cos) and both half-waves meet in one slot: LDS atomics
                            atomicAdd(part + xi, s0);
                            atomicAdd(part + 10 + xi, s1);
                            atomicAdd(part + 20 + xi, s2);
                        }
                    }
                }
            }
            if (!is_emb && u0 < KR) {     // d feature row of this neighbour: scattered by k_feat_scatter (line-coalesced atomics)
                if ((a.flags & LK_FLAG_GRAD_FEATS) && live)
                    *reinterpret_cast<float4*>(a.dfeat + ((size_t)sp * 8 + nb_i) * LK_C + (u0 - ER)) =
                        make_float4(dx[tile][4 * g], dx[tile][4 * g + 1], dx[tile][4 * g + 2], dx[tile][4 * g + 3]);
            }
        }
    if (want_p) {
        // both halves of a row, then the 8 neighbour rows of the sample; d p = - d (x_I - p)
        dax += __shfl_xor(dax, 32); day += __shfl_xor(day, 32); daz += __shfl_xor(daz, 32);
        dax += __shfl_xor(dax, 1); dax += __shfl_xor(dax, 2); dax += __shfl_xor(dax, 4);
        day += __shfl_xor(day, 1); day += __shfl_xor(day, 2); day += __shfl_xor(day, 4);
        daz += __shfl_xor(daz, 1); daz += __shfl_xor(daz, 2); daz += __shfl_xor(daz, 4);
        if (live && h == 0 && nb_i == 0) *reinterpret_cast<float4*>(a.dp_rel + (size_t)sp * 4) = make_float4(-dax, -day, -daz, 0.0f);
    }
}

__global__ __launch_bounds__(256) void k_relpos_bwd(LkRelposBwdArgs a) {
    __shared__ float s_part[4][32];
    const bool want_w = (a.flags & LK_FLAG_GRAD_WEIGHTS) != 0;
    const int w = (int)threadIdx.x >> 6;
    if (want_w) {
        if (threadIdx.x < 128) (&s_part[0][0])[threadIdx.x] = 0.0f;
        __syncthreads();
    }
    const int sample0 = (blockIdx.x * 4 + w) * 4;
    if (sample0 < a.P) relpos_bwd_wave(a, sample0, s_part[w]);
    if (want_w) {
        __syncthreads();
        if (threadIdx.x < 32)
            a.part_br[(size_t)blockIdx.x * 32 + threadIdx.x] = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] +
                                                                 s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
    }
}

// ---------------------------------------------------------------------------------------------
// dW[n][k] += sum_rows A[row][n] * B[row][k]  for every decoder matrix (one "job" each).
// One 4-wave workgroup per (job, chunk of rows).  Rows are streamed 32 at a time: all 256 threads fetch the
// A and B row tiles with coalesced 16-byte loads (register prefetch of the next tile overlaps the MFMAs of the
// current one), the element-wise factor of A (softplus', or the rel-pos row weight) is applied on the fly, the
// tiles go through LDS once, and each wave owns a subset of the 32x32 output blocks: 16 k-steps of
// v_mfma_f32_32x32x2_f32 per tile and block, operands read conflict-free from LDS.  Accumulators stay in
// registers for the whole chunk and are flushed once with line-coalesced atomics; bias gradients are the
// column sums of the A tiles.
#define WG_RT 32
#define WG_LDA 132
#define WG_LDB 196

// Raw operand fetches.  Nothing may CONSUME a prefetched register before the MFMA phase of the current tile has
// been issued (a select or multiply right after the load makes the compiler wait for the data first, which
// serialised load latency and MFMAs: 5.3 us per tile instead of 2): addresses are clamped instead of
// predicated, and masking / the softplus' factor / the rel-pos row weight are applied when the tile is written to LDS.
__device__ __forceinline__ void wg_fetch_a(const LkWgradJob& J, long long row, int c4, float4& v, float4& v2) {
    if (J.a_mode == 2) {
        v = *reinterpret_cast<const float4*>(J.A + (size_t)row * J.lda + 4 * c4);
        const float w = J.A2[row];
        v2 = make_float4(w, w, w, w);
    } else {
        v = *reinterpret_cast<const float4*>(J.A + (size_t)row * J.lda + 4 * c4);
        if (J.a_mode == 1) v2 = *reinterpret_cast<const float4*>(J.A2 + (size_t)row * J.lda2 + 4 * c4);
    }
}
__device__ __forceinline__ float4 wg_finish_a(const LkWgradJob& J, float4 v, float4 v2, bool ok) {
    if (!ok) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (J.a_mode == 1) {
        v.x *= lk_softplus100_grad_from_out(v2.x); v.y *= lk_softplus100_grad_from_out(v2.y);
        v.z *= lk_softplus100_grad_from_out(v2.z); v.w *= lk_softplus100_grad_from_out(v2.w);
    } else if (J.a_mode == 2) {
        v.x *= v2.x; v.y *= v2.x; v.z *= v2.x; v.w *= v2.x;
    }
    return v;
}
__device__ __forceinline__ float4 wg_fetch_b(const LkWgradJob& J, long long row, int c4) {
    const int k = 4 * c4;
    if (J.B2 && k >= J.k_split) return *reinterpret_cast<const float4*>(J.B2 + (size_t)row * J.ldb2 + (k - J.k_split));
    return *reinterpret_cast<const float4*>(J.B + (size_t)row * J.ldb + k);
}

__global__ __launch_bounds__(256) void k_wgrad(LkWgradArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[WG_RT * WG_LDA];
    __shared__ __attribute__((aligned(16))) float sB[WG_RT * WG_LDB];
    const LkWgradJob& J = a.job[blockIdx.y];
    const long long c0 = (long long)blockIdx.x * a.chunk;
    if (c0 >= J.rows) return;                                          // uniform per block
    const long long c1 = (c0 + a.chunk < J.rows) ? c0 + a.chunk : J.rows;
    const int t = (int)threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, hh = lane >> 5;
    const int N4 = (J.N + 3) >> 2, K4 = (J.K + 3) >> 2;                // float4 columns of the A / B tiles
    const int NB = (J.N + 31) >> 5, KB = (J.K + 31) >> 5, U = NB * KB; // 32x32 output blocks, U <= 24
    const int nA = WG_RT * N4, nB = WG_RT * K4;                        // float4 elements per tile
    f32x16 acc[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) acc[q] = lk_zero16();
    float bsum = 0.0f;
    float4 ra[4], ra2[4], rb[6];
    // zero the padding columns once (tiles narrower than a multiple of 32 leave stale LDS otherwise)
    for (int e = t; e < WG_RT * WG_LDA; e += 256) sA[e] = 0.0f;
    for (int e = t; e < WG_RT * WG_LDB; e += 256) sB[e] = 0.0f;
    // element -> (row-in-tile, float4 column), clamped so that every thread always has a legal address
    int ar[4], ac[4], br[6], bc[6];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int e = min(t + 256 * q, nA - 1); ar[q] = e / N4; ac[q] = e - ar[q] * N4; ra2[q] = make_float4(1.f, 1.f, 1.f, 1.f); }
#pragma unroll
    for (int q = 0; q < 6; ++q) { const int e = min(t + 256 * q, nB - 1); br[q] = e / K4; bc[q] = e - br[q] * K4; }
    const long long last = c1 - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) wg_fetch_a(J, min(c0 + ar[q], last), ac[q], ra[q], ra2[q]);
#pragma unroll
    for (int q = 0; q < 6; ++q) rb[q] = wg_fetch_b(J, min(c0 + br[q], last), bc[q]);
    for (long long tile = c0; tile < c1; tile += WG_RT) {
        __syncthreads();                                               // previous tile fully consumed
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (t + 256 * q < nA) *reinterpret_cast<float4*>(sA + ar[q] * WG_LDA + 4 * ac[q]) = wg_finish_a(J, ra[q], ra2[q], tile + ar[q] < c1);
#pragma unroll
        for (int q = 0; q < 6; ++q)
            if (t + 256 * q < nB) *reinterpret_cast<float4*>(sB + br[q] * WG_LDB + 4 * bc[q]) = (tile + br[q] < c1) ? rb[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        const long long nxt = tile + WG_RT;
        if (nxt < c1) {                                                // prefetch the next tile: raw loads only
#pragma unroll
            for (int q = 0; q < 4; ++q) wg_fetch_a(J, min(nxt + ar[q], last), ac[q], ra[q], ra2[q]);
#pragma unroll
            for (int q = 0; q < 6; ++q) rb[q] = wg_fetch_b(J, min(nxt + br[q], last), bc[q]);
        }
        if (J.db && t < J.N) {
#pragma unroll 8
            for (int r = 0; r < WG_RT; ++r) bsum += sA[r * WG_LDA + t];
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int u = w + 4 * q;
            if (u < U) {                                               // wave-uniform
                const int nb = u / KB, kb = u - nb * KB;
                const float* pa = sA + hh * WG_LDA + nb * 32 + l31;
                const float* pb = sB + hh * WG_LDB + kb * 32 + l31;
                // all 32 operand reads of this block are issued before its 16 MFMAs: with one wave per SIMD nothing
                // else hides the LDS latency (an interleaved read->MFMA chain measured 2.7x the MFMA time)
                float av[WG_RT / 2], bv[WG_RT / 2];
#pragma unroll
                for (int s2 = 0; s2 < WG_RT / 2; ++s2) { av[s2] = pa[2 * s2 * WG_LDA]; bv[s2] = pb[2 * s2 * WG_LDB]; }
#pragma unroll
                for (int s2 = 0; s2 < WG_RT / 2; ++s2) acc[q] = lk_mfma(av[s2], bv[s2], acc[q]);
            }
        }
    }
    // flush: lane holds column k = kb*32 + (lane&31), rows n = nb*32 + frag_row(r, half)
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int u = w + 4 * q;
        if (u < U) {
            const int nb = u / KB, kb = u - nb * KB;
            const int k = kb * 32 + l31;
            if (k < J.K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nn = nb * 32 + lk_frag_row(r, hh);
                    if (nn < J.N) atomicAdd(J.dW + (size_t)nn * J.ldw + k, acc[q][r]);
                }
            }
        }
    }
    if (J.db && t < J.N) atomicAdd(J.db + t, bsum);
}

int lk_launch_interp_bwd(const LkInterpBwdArgs& a, hipStream_t st) {
    LkProfScope prof_(LKK_INTERP_BWD, st);
    hipLaunchKernelGGL(k_interp_bwd, dim3(lk_cdiv(a.P, 32)), dim3(256), 0, st, a);
    return LK_OK;
}
int lk_launch_feat_scatter(const LkFeatScatterArgs& a, hipStream_t st) {
    LkProfScope prof_(LKK_INTERP_BWD, st);
    hipLaunchKernelGGL(k_feat_scatter, dim3(lk_cdiv((long long)a.P * LK_K, 8)), dim3(256), 0, st, a);
    return LK_OK;
}
int lk_launch_rays_bwd(const LkRaysBwdArgs& a, hipStream_t st) {
    LkProfScope prof_(LKK_RAYS_BWD, st);
    hipLaunchKernelGGL(k_rays_bwd, dim3(lk_cdiv(a.R, 256)), dim3(256), 0, st, a);
    return LK_OK;
}
int lk_launch_relpos_bwd(const LkRelposBwdArgs& a, hipStream_t st) {
    LkProfScope prof_(LKK_RELPOS_BWD, st);
    const int waves = lk_cdiv(a.P, 4);
    hipLaunchKernelGGL(k_relpos_bwd, dim3(lk_cdiv(waves, 4)), dim3(256), 0, st, a);
    return LK_OK;
}
int lk_launch_wgrad(const LkWgradArgs& a, int max_rows, hipStream_t st) {
    LkProfScope prof_(LKK_WGRAD, st);
    if (a.n_jobs == 0 || max_rows <= 0) return LK_OK;
    hipLaunchKernelGGL(k_wgrad, dim3(lk_cdiv(max_rows, a.chunk), a.n_jobs), dim3(256), 0, st, a);
    return LK_OK;
}
