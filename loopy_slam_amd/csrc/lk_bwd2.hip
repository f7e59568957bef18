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
    const bool act = live && has;
    if ((a.flags & LK_FLAG_GRAD_FEATS) && act) {
#pragma unroll
        for (int j = 0; j < LK_K; ++j) {
            if (w[j] != 0.0f) {
                float* gg = a.g_geo_feats + (size_t)id[j] * LK_C + sub * 4;
                atomicAdd(gg + 0, w[j] * dcg.x); atomicAdd(gg + 1, w[j] * dcg.y);
                atomicAdd(gg + 2, w[j] * dcg.z); atomicAdd(gg + 3, w[j] * dcg.w);
                if (do_col) {
                    float* gc = a.g_col_feats + (size_t)id[j] * LK_C + sub * 4;
                    atomicAdd(gc + 0, w[j] * dcc.x); atomicAdd(gc + 1, w[j] * dcc.y);
                    atomicAdd(gc + 2, w[j] * dcc.z); atomicAdd(gc + 3, w[j] * dcc.w);
                }
            }
        }
    }
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
        if (color && relpos && a.dp_rel) { const float4 e = *reinterpret_cast<const float4*>(a.dp_rel + (size_t)pidx * 4); dpx += e.x; dpy += e.y; dpz += e.z; }
        *reinterpret_cast<float4*>(a.dp_total + (size_t)pidx * 4) = make_float4(dpx, dpy, dpz, 0.0f);
    }
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

__global__ __launch_bounds__(256) void k_relpos_bwd(LkRelposBwdArgs a) {
    const int lane = lk_lane();
    const int wave = blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    const int sample0 = wave * 4;
    if (sample0 >= a.P) return;
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
    if (want_w && live) {     // rows for the streamed weight-gradient reductions: hid | dhid | x
        if (h == 0) a.w_eff[(size_t)sp * 8 + nb_i] = wgt;
        float* row = a.rows + ((size_t)sp * 8 + nb_i) * 320;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<float4*>(row + nb * 32 + 8 * g + 4 * h) =
                    make_float4(hid[nb][4 * g], hid[nb][4 * g + 1], hid[nb][4 * g + 2], hid[nb][4 * g + 3]);
                *reinterpret_cast<float4*>(row + 128 + nb * 32 + 8 * g + 4 * h) =
                    make_float4(dhid[nb][4 * g], dhid[nb][4 * g + 1], dhid[nb][4 * g + 2], dhid[nb][4 * g + 3]);
            }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4*>(row + 256 + 8 * g + 4 * h) = make_float4(x0[4 * g], x0[4 * g + 1], x0[4 * g + 2], x0[4 * g + 3]);
            *reinterpret_cast<float4*>(row + 288 + 8 * g + 4 * h) = make_float4(x1[4 * g], x1[4 * g + 1], x1[4 * g + 2], x1[4 * g + 3]);
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
            if (u0 < ER) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int u = u0 + t;
                    const int xi = (u < 10) ? u : u - 10;
                    const float b0 = B[xi], b1 = B[10 + xi], b2 = B[20 + xi];
                    const float x = lk_fourier_arg(a0, a1, a2, b0, b1, b2);
                    const float f = (u < 10) ? lk_cosf(x) : -lk_sinf(x);
                    const float gx = dx[tile][4 * g + t] * f;
                    if (want_p) {
                        dax = fmaf(gx * LK_TWO_PI, b0, dax); day = fmaf(gx * LK_TWO_PI, b1, day); daz = fmaf(gx * LK_TWO_PI, b2, daz);
                    }
                    if (want_w) {
                        const float s0 = lk_half_wave_sum(gx * a0), s1 = lk_half_wave_sum(gx * a1), s2 = lk_half_wave_sum(gx * a2);
                        if ((lane & 31) == 0) {
                            atomicAdd(a.g_weights + R_EB + xi, s0);
                            atomicAdd(a.g_weights + R_EB + 10 + xi, s1);
                            atomicAdd(a.g_weights + R_EB + 20 + xi, s2);
                        }
                    }
                }
            } else if (u0 < KR) {
                if ((a.flags & LK_FLAG_GRAD_FEATS) && wgt != 0.0f) {
                    float* gc = a.g_col_feats + (size_t)idx * LK_C + (u0 - ER);
                    atomicAdd(gc + 0, dx[tile][4 * g]); atomicAdd(gc + 1, dx[tile][4 * g + 1]);
                    atomicAdd(gc + 2, dx[tile][4 * g + 2]); atomicAdd(gc + 3, dx[tile][4 * g + 3]);
                }
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

// ---------------------------------------------------------------------------------------------
// dW[n][k] += sum_rows A[row][n] * B[row][k].  One wave per (job, 32-row block of dW, chunk of rows):
// A and B are read straight from the row-major activation scratch (a half-wave reads one 128-B line),
// two rows per v_mfma_f32_32x32x2_f32, accumulators live in registers for the whole chunk and are
// flushed once with coalesced atomics.  Bias gradients ride along as a running sum of the A operand.
__device__ __forceinline__ float wg_load_a(const LkWgradJob& J, size_t row, int n) {
    if (n >= J.N) return 0.0f;
    if (J.a_mode == 0) return J.A[row * J.lda + n];
    if (J.a_mode == 1) return J.A[row * J.lda + n] * lk_softplus100_grad_from_out(J.A2[row * J.lda2 + n]);
    return J.A2[row] * J.A[(row >> 3) * J.lda + n];           // rel-pos: w[row] * dc[sample][n]
}
__device__ __forceinline__ float wg_load_b(const LkWgradJob& J, size_t row, int k) {
    if (k >= J.K) return 0.0f;
    if (J.B2 && k >= J.k_split) return J.B2[row * J.ldb2 + (k - J.k_split)];
    return J.B[row * J.ldb + k];
}

__global__ __launch_bounds__(64) void k_wgrad(LkWgradArgs a) {
    const int lane = lk_lane();
    // decode the work item: blockIdx.y enumerates (job, n-block)
    int item = blockIdx.y, ji = 0;
    for (; ji < a.n_jobs; ++ji) {
        const int nbj = (a.job[ji].N + 31) >> 5;
        if (item < nbj) break;
        item -= nbj;
    }
    if (ji >= a.n_jobs) return;
    const LkWgradJob& J = a.job[ji];
    const int nb = item;
    const long long c0 = (long long)blockIdx.x * a.chunk;
    if (c0 >= J.rows) return;
    const long long c1 = (c0 + a.chunk < J.rows) ? c0 + a.chunk : J.rows;
    const int KB = (J.K + 31) >> 5;
    const int n = nb * 32 + (lane & 31);
    const int kl = lane & 31;
    const int hh = lane >> 5;
    f32x16 acc[6];
#pragma unroll
    for (int kb = 0; kb < 6; ++kb) acc[kb] = lk_zero16();
    float bsum = 0.0f;
    for (long long row = c0 + hh; row < c1 + hh; row += 2) {     // both halves iterate the same count
        const bool ok = row < c1;
        const float av = ok ? wg_load_a(J, (size_t)row, n) : 0.0f;
        bsum += av;
#pragma unroll
        for (int kb = 0; kb < 6; ++kb) {
            if (kb < KB) {
                const float bv = ok ? wg_load_b(J, (size_t)row, kb * 32 + kl) : 0.0f;
                acc[kb] = lk_mfma(av, bv, acc[kb]);
            }
        }
    }
    // flush: lane holds column k = kb*32 + (lane&31), rows n = nb*32 + frag_row(r, half)
#pragma unroll
    for (int kb = 0; kb < 6; ++kb) {
        if (kb < KB) {
            const int k = kb * 32 + kl;
            if (k < J.K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nn = nb * 32 + lk_frag_row(r, hh);
                    if (nn < J.N) atomicAdd(J.dW + (size_t)nn * J.ldw + k, acc[kb][r]);
                }
            }
        }
    }
    if (J.db) {
        bsum += __shfl_xor(bsum, 32);
        if (hh == 0 && n < J.N) atomicAdd(J.db + n, bsum);
    }
}

int lk_launch_interp_bwd(const LkInterpBwdArgs& a, hipStream_t st) {
    LkProfScope prof_(LKK_INTERP_BWD, st);
    hipLaunchKernelGGL(k_interp_bwd, dim3(lk_cdiv(a.P, 32)), dim3(256), 0, st, a);
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
    int items = 0;
    for (int j = 0; j < a.n_jobs; ++j) items += (a.job[j].N + 31) / 32;
    if (items == 0 || max_rows <= 0) return LK_OK;
    hipLaunchKernelGGL(k_wgrad, dim3(lk_cdiv(max_rows, a.chunk), items), dim3(64), 0, st, a);
    return LK_OK;
}
