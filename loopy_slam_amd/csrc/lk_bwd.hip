// Backward of the render graph (autograd of Renderer.render_batch_ray: Mapper.py:722, Tracker.py:193).
//   k_composite_bwd   d(depth,var,color) -> d raw[P,4]                        (common.py:408-421)
//   k_decode_bwd      d raw -> d c_geo, d c_col, d y_i rows (for k_wgrad), d p, d B_g  (decoder.py:263-288, 513-546)
// Same register-chained scheme as the forward (CT tiles, bf16x6 products: lk_common.h) on the TRANSPOSED
// weight fragments: dX^T = W^T dY^T, the dY CT tile being the B operand.
#include "lk_common.h"
#include "lk_kernels.h"
#include "lk_composite_dev.h"
#include "lk_track_dev.h"

using namespace lkw;

LK_CHAIN_DEFINE(bwd)

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_composite_bwd(LkCompositeBwdArgs a) {
    const int r = blockIdx.x * 256 + (int)threadIdx.x;
    if (r >= a.R) return;
    lk_composite_bwd_ray(a.raw, a.z, a.nbr_count, r, a.S, a.min_nn, a.coef, a.keep_depth ? 1.0f : a.gt_depth[r], a.d_depth[r], a.d_var ? a.d_var[r] : 0.0f,
                         a.d_color ? a.d_color[3 * r] : 0.0f, a.d_color ? a.d_color[3 * r + 1] : 0.0f, a.d_color ? a.d_color[3 * r + 2] : 0.0f,
                         a.d_raw);
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 ct_load32(const float* __restrict__ row, int lane) {
    const int h = lane >> 5;
    f32x16 t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(row + 8 * g + 4 * h);
        t[4 * g + 0] = v.x; t[4 * g + 1] = v.y; t[4 * g + 2] = v.z; t[4 * g + 3] = v.w;
    }
    return t;
}
__device__ __forceinline__ void ct_store32(float* __restrict__ row, const f32x16& t, bool live, int lane) {
    const int h = lane >> 5;
    if (!live) return;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(row + 8 * g + 4 * h) = make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
}

// Sample coordinates shared by both roles
struct BwdSample {
    int sample, sp, h;
    bool live, store;       // live: carries gradient (loads come from sample sp);  store: a sample of the batch - its rows are written (zeros if !live)
    float a0, a1, a2;
};
__device__ __forceinline__ BwdSample bwd_sample(const LkDecodeBwdArgs& a, int tile, int lane) {
    BwdSample d;
    d.sample = tile * 32 + (lane & 31);
    // live = carries gradient: samples of rays behind the live prefix of a partitioned batch sit in the last processed tile too.  Their loss
    // gradient is zero, but their d raw is formed from what the forward left in raw for them - for a ray whose samples straddle into a SKIPPED
    // tile that is stale memory, and 0 x NaN is NaN: such a lane took the tile's Fourier-matrix partial sums with it (found by poisoning the
    // test buffers: a -1 index table freed and reused as a float buffer is NaN bit patterns)
    const int P_live = a.live_rays ? min(a.P, *a.live_rays * a.S) : a.P;
    d.live = d.sample < P_live;
    d.store = d.sample < a.P;
    d.h = lane >> 5;
    d.sp = d.live ? d.sample : P_live - 1;            // (a processed sample: what the dead lanes load is defined)
    const int r = d.sp / a.S;
    const float z = a.z[d.sp];
    const float px = lk_madd_rn(a.rays_o[3 * r], a.rays_d[3 * r], z);
    const float py = lk_madd_rn(a.rays_o[3 * r + 1], a.rays_d[3 * r + 1], z);
    const float pz = lk_madd_rn(a.rays_o[3 * r + 2], a.rays_d[3 * r + 2], z);
    d.a0 = __fmul_rn(LK_TWO_PI, px); d.a1 = __fmul_rn(LK_TWO_PI, py); d.a2 = __fmul_rn(LK_TWO_PI, pz);
    return d;
}

// ================= colour decoder backward-data: FOUR waves = one 32-sample tile =================
// Mirror of decode_col_wg: wave w owns units [32w, 32w+32) of every layer's gradient.  Per layer i = 4..0:
//   d c  += U_i^T[:, own units] d h_i[own]                (partial over the wave's units; summed over waves at the end)
//   d y_i = d h_i * softplus'(a_i)                        (own units)  -> parked in LDS (same lane-chunk scheme)
//   d h_{i-1}[own] = W_i^T[own, :] d y_i                  (B operand = all four parked blocks)
// The embedding gradient tiles (tracker mode) are spread evenly: layer 3's two tiles on waves 0,1, layer 0's on
// waves 2,3; every wave turns its tile into a d p partial and wave 0 adds the four.
// H16: products on fp16 pieces (lk_mma3h) instead of bf16 pieces (lk_mma6).  Only for unit-scale loss gradients
// (LK_FLAG_UNIT_LOSS_GRADS, mapper mode): the whole chain d h_4 .. d h_0 is linear in d out, so d out is multiplied by 2^10
// once (median |d h| 1.5e-4 -> 0.15: both fp16 pieces normal numbers), everything in between is scaled with it, and the
// stored d h rows / d c are scaled back where they are written (in the copy resp. the final sum: no extra instruction).
// DEEP: as in the forward (lk_decode.hip) - launches whose tiles are all resident at once fetch more blocks of W_i^T ahead
// (all eight; the register count of the kernel is set by its geometry role)
// TL: the launch may belong to the tracking loop (LkDecodeBwdArgs::tl_n_part) - d raw formed in the prologue instead of read
template <bool H16, bool DEEP, bool TL>
__device__ __forceinline__ void decode_bwd_col_wg(const LkDecodeBwdArgs& a, int tile, int w, int lane,
                                                  u32x4* __restrict__ s_x /* [2][24*64] */, float (*s_o)[3 * 32]) {
    typedef BwdPiece<H16> PC;
    typedef typename PC::T Piece;
    constexpr int NP = PC::NP;
    // (a.dscale: a power of two from the device, exact in both directions)
    const float ds = (H16 && a.dscale) ? *a.dscale : 1.0f;
    const float SC = H16 ? 1024.0f * ds : 1.0f, ISC = H16 ? (1.0f / 1024.0f) / ds : 1.0f;
    const BwdSample d = bwd_sample(a, tile, lane);
    const int h = d.h, sp = d.sp;
    const bool live = d.live;
    const float a0 = d.a0, a1 = d.a1, a2 = d.a2;
    const float* __restrict__ W = a.W;
    const u32x4* __restrict__ FB = reinterpret_cast<const u32x4*>(a.Wfrag) + (H16 ? FRAGB_U4 : 0);
    // (the colour trunk has no trainable Fourier matrix: with LK_FLAG_EMBED_GRADS_ONLY nothing here is a weight-gradient operand)
    const bool want_w = (a.flags & LK_FLAG_GRAD_WEIGHTS) != 0 && !(a.flags & LK_FLAG_EMBED_GRADS_ONLY);
    const bool want_p = (a.flags & LK_FLAG_GRAD_RAYS) != 0;
    // the derivative mask softplus'(z_i) as the forward stored it (unorm16 pairs, this lane's 16 values contiguous): + LK_COL_SLAYER(P, layer)
    const unsigned* act_col_s = reinterpret_cast<const unsigned*>(a.act + (size_t)a.P * LK_ACT_GEO_A) + (size_t)sp * 64 + 16 * w + 8 * h;
    // ... or, behind a TRACKER-mode forward (lk_kernels.h: LK_ACT_COL_A), the fp32 a_i rows.  Only the TL instantiations can meet one (the
    // launcher keeps tracker-mode descriptors out of the mapper-loop form), so the mapper form carries no second load path
    const bool a32 = TL && (a.flags & LK_FLAG_TRACKER) != 0;
    const float* act_col_a = a.act + (size_t)a.P * LK_ACT_GEO_A + (size_t)sp * 128 + w * 32;
    const float* act_col_h = a.act + (size_t)a.P * (LK_ACT_GEO_A + LK_ACT_COL_A) + (size_t)sp * 128;
    float4 draw;
    if (a.ml_on) { float t0, t1, t2; draw = lk_map_draw(a.ml, sp, false, &t0, &t1, &t2); }
    else if (a.cb_on) draw = lk_cb_draw(a.cb, sp);
    else if (TL && a.tl_n_part > 0) draw = lk_track_draw(a.tl, a.tl_n_part, sp, nullptr);
    else draw = *reinterpret_cast<const float4*>(a.d_raw + (size_t)sp * 4);
    if (!live) draw = make_float4(0.f, 0.f, 0.f, 0.f);            // dead lanes contribute nothing to reductions
    float g0 = draw.x, g1 = draw.y, g2 = draw.z;
    const float4 yo = *reinterpret_cast<const float4*>(a.raw + (size_t)sp * 4);
    LK_STAMPW(1);                                    // (probe build) threshold, loss term and composite backward of the lane's sample
    if (!(a.flags & LK_FLAG_COLOR_LOGITS)) {                     // through the sigmoid
        g0 *= yo.x * (1.0f - yo.x); g1 *= yo.y * (1.0f - yo.y); g2 *= yo.z * (1.0f - yo.z);
    }
    // output-layer weights of the wave's own units
    float4 wo0[4], wo1[4], wo2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int u = 32 * w + 8 * g + 4 * h;
        wo0[g] = *reinterpret_cast<const float4*>(W + C_WO + u);
        wo1[g] = *reinterpret_cast<const float4*>(W + C_WO + HC + u);
        wo2[g] = *reinterpret_cast<const float4*>(W + C_WO + 2 * HC + u);
    }
    if (a.affine) {                                              // out' = out @ A + t  (decoder.py:536-539)
        if (a.g_affine) {
            // recompute the pre-affine output o = Wo h4 + bo: per-wave partial over its units, summed by wave 0
            const f32x16 h4 = ct_load32(act_col_h + LK_COL_LAYER(a.P, 4) + w * 32, lane);
            float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v0 = h4[4 * g], v1 = h4[4 * g + 1], v2 = h4[4 * g + 2], v3 = h4[4 * g + 3];
                o0 = fmaf(wo0[g].x, v0, o0); o0 = fmaf(wo0[g].y, v1, o0); o0 = fmaf(wo0[g].z, v2, o0); o0 = fmaf(wo0[g].w, v3, o0);
                o1 = fmaf(wo1[g].x, v0, o1); o1 = fmaf(wo1[g].y, v1, o1); o1 = fmaf(wo1[g].z, v2, o1); o1 = fmaf(wo1[g].w, v3, o1);
                o2 = fmaf(wo2[g].x, v0, o2); o2 = fmaf(wo2[g].y, v1, o2); o2 = fmaf(wo2[g].z, v2, o2); o2 = fmaf(wo2[g].w, v3, o2);
            }
            o0 += __shfl_xor(o0, 32); o1 += __shfl_xor(o1, 32); o2 += __shfl_xor(o2, 32);
            if (h == 0) { s_o[w][lane] = o0; s_o[w][32 + lane] = o1; s_o[w][64 + lane] = o2; }
            __syncthreads();
            if (w == 0) {
                const int c = lane & 31;
                const float oo[3] = {((s_o[0][c] + s_o[1][c]) + s_o[2][c]) + s_o[3][c] + W[C_BO],
                                     ((s_o[0][32 + c] + s_o[1][32 + c]) + s_o[2][32 + c]) + s_o[3][32 + c] + W[C_BO + 1],
                                     ((s_o[0][64 + c] + s_o[1][64 + c]) + s_o[2][64 + c]) + s_o[3][64 + c] + W[C_BO + 2]};
                const float gm[3] = {g0, g1, g2};
#pragma unroll
                for (int c3 = 0; c3 < 3; ++c3)
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const float v = lk_half_wave_sum(oo[c3] * gm[m]);
                        if (lane == LK_HWS_LANE) a.g_affine_part[(size_t)tile * 12 + c3 * 3 + m] = v;
                    }
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const float v = lk_half_wave_sum(gm[m]);
                    if (lane == LK_HWS_LANE) a.g_affine_part[(size_t)tile * 12 + 9 + m] = v;
                }
            }
            __syncthreads();                                     // s_o is reused for the d p partials
        }
        const float* A = a.affine;
        const float t0 = g0 * A[0] + g1 * A[1] + g2 * A[2];
        const float t1 = g0 * A[3] + g1 * A[4] + g2 * A[5];
        const float t2 = g0 * A[6] + g1 * A[7] + g2 * A[8];
        g0 = t0; g1 = t1; g2 = t2;
    }
    if (want_w && d.store && h == 0 && w == 0) *reinterpret_cast<float4*>(a.dlogit + (size_t)d.sample * 4) = make_float4(g0, g1, g2, 0.0f);
    // dh4[own] = Wo^T d out (H16: times 2^10 from here on)
    g0 *= SC; g1 *= SC; g2 *= SC;
    f32x16 dh, dy;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        dh[4 * g + 0] = wo0[g].x * g0 + wo1[g].x * g1 + wo2[g].x * g2;
        dh[4 * g + 1] = wo0[g].y * g0 + wo1[g].y * g1 + wo2[g].y * g2;
        dh[4 * g + 2] = wo0[g].z * g0 + wo1[g].z * g1 + wo2[g].z * g2;
        dh[4 * g + 3] = wo0[g].w * g0 + wo1[g].w * g1 + wo2[g].w * g2;
    }
    f32x16 dc = lk_zero16(), de = lk_zero16();
    int buf = 0;
    // Split-bf16 products (lk_common.h::lk_mma6).  Loads and stores share one in-order counter, so what a layer needs
    // right after its d y store - un = U_i^T blocks of the own units, sv = the saved derivative mask, wn = first four blocks of W_i^T for
    // the own output block - is fetched BEFORE that store, at the end of the previous layer; blocks 4..7 come in line.
    constexpr int NPF = DEEP ? 8 : 4;
    Piece un[2], wn[NPF];
    u32x4 sv0, sv1;
    f32x16 av;
    auto prefetch = [&](int i) {
        const u32x4* ut = FB + PC::tr(15 + i);
#pragma unroll
        for (int G = 0; G < 2; ++G) un[G] = PC::load(ut, 1, 2 * w + G, 0, lane);
        if (TL && a32) av = ct_load32(act_col_a + LK_COL_LAYER(a.P, i), lane);
        else { sv0 = *reinterpret_cast<const u32x4*>(act_col_s + LK_COL_SLAYER(a.P, i)); sv1 = *reinterpret_cast<const u32x4*>(act_col_s + LK_COL_SLAYER(a.P, i) + 4); }
        if (i >= 1) {
            const u32x4* wt = FB + PC::tr(10 + i);
#pragma unroll
            for (int G = 0; G < NPF; ++G) wn[G] = PC::load(wt, i == 3 ? 6 : 4, G, i == 3 ? 2 + w : w, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto lds_b = [&](const u32x4* xs, int G) {
        Piece b;
#pragma unroll
        for (int q = 0; q < NP; ++q) b.p[q] = xs[(G * NP + q) * 64 + lane];
        return b;
    };
    prefetch(4);
    LK_STAMPW(2);                                    // d h_4 formed, layer 4's operands arrived
#pragma unroll
    for (int i = 4; i >= 0; --i) {
        if (TL && a32) {
#pragma unroll
            for (int q = 0; q < 16; ++q) dy[q] = dh[q] * lk_softplus100_grad_from_out(av[q]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                dy[2 * q] = dh[2 * q] * lk_unorm16_lo(sv0[q]); dy[2 * q + 1] = dh[2 * q + 1] * lk_unorm16_hi(sv0[q]);
                dy[8 + 2 * q] = dh[8 + 2 * q] * lk_unorm16_lo(sv1[q]); dy[8 + 2 * q + 1] = dh[8 + 2 * q + 1] * lk_unorm16_hi(sv1[q]);
            }
        }
        // d y_i rows for the weight-gradient jobs (W_i and, through the auxiliary columns of job i, U_{i-1}: lk_kernels.h LkFcPost).
        // (Round 1 stored a register COPY of its rows, believing that the stores must not read registers the next product
        // overwrites.  The cause was elsewhere: the copy happened to stop the SLP vectoriser from turning d h = Wo^T d out into
        // packed-fp32 instructions, and it is those that corrupt lanes 48-63 of a register when two workgroups share a compute
        // unit - see build.py and DESIGN.md §3.  The library is built without them.)
        if (want_w) {
            f32x16 dys = dy;
            if (H16) {
#pragma unroll
                for (int q = 0; q < 16; ++q) dys[q] = dy[q] * ISC;
            }
            ct_store32(a.dy_col + LK_COL_LAYER(a.P, i) + (size_t)d.sample * 128 + w * 32, dys, d.store, lane);
        }
#pragma unroll
        for (int G = 0; G < 2; ++G) dc = PC::mma(un[G], PC::split(dh, G), dc);
        if (i == 0 && !want_p) break;
        u32x4* xs = s_x + buf * (24 * 64);
#pragma unroll
        for (int G = 0; G < 2; ++G) {
            const Piece b = PC::split(dy, G);
#pragma unroll
            for (int q = 0; q < NP; ++q) xs[((w * 2 + G) * NP + q) * 64 + lane] = b.p[q];
        }
        __syncthreads();
        LK_STAMP(7 - i);                             // slots 3..7: the d y pieces of layers 4..0 are parked
        buf ^= 1;
        if (i >= 1) {
            const u32x4* wt = FB + PC::tr(10 + i);
            const int nbt = i == 3 ? 6 : 4, nb = i == 3 ? 2 + w : w;
            dh = lk_zero16();
#pragma unroll
            for (int G = 0; G < 8; ++G) dh = PC::mma(G < NPF ? wn[G] : PC::load(wt, nbt, G, nb, lane), lds_b(xs, G), dh);
            if (i == 3 && want_p && w < 2) {
#pragma unroll
                for (int G = 0; G < 8; ++G) de = PC::mma(PC::load(FB + PC::tr(13), 6, G, w, lane), lds_b(xs, G), de);
            }
            prefetch(i - 1);
        } else {       // i == 0: only the embedding receives gradient
            if (w >= 2) {
#pragma unroll
                for (int G = 0; G < 8; ++G) de = PC::mma(PC::load(FB + PC::tr(10), 2, G, w - 2, lane), lds_b(xs, G), de);
            }
        }
    }
    // d c: park the per-wave partials, wave w sums register chunk w of all four -> one float4 per lane
    {
        float4* xs = reinterpret_cast<float4*>(s_x + buf * (24 * 64));
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[(w * 4 + j) * 64 + lane] = make_float4(dc[4 * j], dc[4 * j + 1], dc[4 * j + 2], dc[4 * j + 3]);
        float dpx = 0.0f, dpy = 0.0f, dpz = 0.0f;
        if (want_p) {    // e_u = sin(x_u) (u<20) | cos(x_{u-20});  dp_i += de_u * f'(x) * 2 pi * B[i][xi]
            const float* B = W + C_EB;
            const int et = w & 1;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int u = 32 * et + lk_frag_row(q, h);
                if (u < EC) {
                    const int xi = (u < 20) ? u : u - 20;
                    const float b0 = B[xi], b1 = B[20 + xi], b2 = B[40 + xi];
                    const float x = lk_fourier_arg(a0, a1, a2, b0, b1, b2);
                    const float f = (u < 20) ? lk_cosf(x) : -lk_sinf(x);
                    const float gx = de[q] * f * LK_TWO_PI;
                    dpx = fmaf(gx, b0, dpx); dpy = fmaf(gx, b1, dpy); dpz = fmaf(gx, b2, dpz);
                }
            }
            dpx += __shfl_xor(dpx, 32); dpy += __shfl_xor(dpy, 32); dpz += __shfl_xor(dpz, 32);
            if (h == 0) { s_o[w][lane] = dpx; s_o[w][32 + lane] = dpy; s_o[w][64 + lane] = dpz; }
        }
        __syncthreads();
        LK_STAMP(8);
        float4 c0 = xs[(0 * 4 + w) * 64 + lane];
        const float4 c1 = xs[(1 * 4 + w) * 64 + lane], c2 = xs[(2 * 4 + w) * 64 + lane], c3 = xs[(3 * 4 + w) * 64 + lane];
        c0.x = ((c0.x + c1.x) + c2.x) + c3.x; c0.y = ((c0.y + c1.y) + c2.y) + c3.y;
        c0.z = ((c0.z + c1.z) + c2.z) + c3.z; c0.w = ((c0.w + c1.w) + c2.w) + c3.w;
        if (H16) { c0.x *= ISC; c0.y *= ISC; c0.z *= ISC; c0.w *= ISC; }
        if (d.store) *reinterpret_cast<float4*>(a.dc_col + (size_t)d.sample * LK_C + 8 * w + 4 * h) = c0;
        if (want_p && w == 0 && h == 0 && d.store) {
            const float x = ((s_o[0][lane] + s_o[1][lane]) + s_o[2][lane]) + s_o[3][lane];
            const float y = ((s_o[0][32 + lane] + s_o[1][32 + lane]) + s_o[2][32 + lane]) + s_o[3][32 + lane];
            const float z = ((s_o[0][64 + lane] + s_o[1][64 + lane]) + s_o[2][64 + lane]) + s_o[3][64 + lane];
            *reinterpret_cast<float4*>(a.dp_embed_col + (size_t)d.sample * 4) = make_float4(x * ISC, y * ISC, z * ISC, 0.0f);      // d e was formed from the scaled d y
        }
    }
    LK_STAMPW(9);
}

// Embedding gradient of the geometry decoder, d e = W_3[:, embedding]^T d y_3 + W_0^T d y_0, one 32-unit block at a time (one accumulator
// tile alive; the loop is not unrolled - three copies of the 48 cosines and 144 reductions are code nobody needs):
// e_u = sin(x_u): ge_u = de_u cos(x_u);  WW: dB[i][u] += sum_s ge_u a_i(s) (-> part);  WP: dp_i += ge_u 2 pi B[i][u]
// GH16: the pieces are fp16 pairs of the 2^10-scaled chain (decode_bwd_geo_wave); isc scales d e back
template <bool WP, bool WW, bool GH16>
__device__ __forceinline__ void geo_embed_bwd(const u32x4* __restrict__ FB, const float* __restrict__ B, const u32x4* __restrict__ park,
                                              const typename BwdPiece<GH16>::T (&y0)[2], float isc, float a0, float a1, float a2,
                                              float& dpx, float& dpy, float& dpz, float* __restrict__ part, int lane) {
    typedef BwdPiece<GH16> PC;
    constexpr int NP = PC::NP;
    const int h = lane >> 5;
#pragma unroll 1
    for (int tile = 0; tile < 3; ++tile) {
        f32x16 de = lk_zero16();
#pragma unroll
        for (int G = 0; G < 2; ++G) {
            typename PC::T b;
#pragma unroll
            for (int q = 0; q < NP; ++q) b.p[q] = park[(G * NP + q) * 64 + lane];
            de = PC::mma(PC::load(FB + PC::tr(3), 4, G, tile, lane), b, de);
        }
#pragma unroll
        for (int G = 0; G < 2; ++G) de = PC::mma(PC::load(FB + PC::tr(0), 3, G, tile, lane), y0[G], de);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int u0 = 32 * tile + 8 * g + 4 * h;
            const float4 b0 = *reinterpret_cast<const float4*>(B + u0);
            const float4 b1 = *reinterpret_cast<const float4*>(B + EGP + u0);
            const float4 b2 = *reinterpret_cast<const float4*>(B + 2 * EGP + u0);
            const float bb0[4] = {b0.x, b0.y, b0.z, b0.w}, bb1[4] = {b1.x, b1.y, b1.z, b1.w}, bb2[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int u = u0 + t;
                float ge = (GH16 ? de[4 * g + t] * isc : de[4 * g + t]) * lk_cosf(lk_fourier_arg(a0, a1, a2, bb0[t], bb1[t], bb2[t]));
                if (u >= EG) ge = 0.0f;                      // padding units of the last block
                if (WP) {
                    const float gx = ge * LK_TWO_PI;
                    dpx = fmaf(gx, bb0[t], dpx); dpy = fmaf(gx, bb1[t], dpy); dpz = fmaf(gx, bb2[t], dpz);
                }
                if (WW) {
                    const float s0 = lk_half_wave_sum(ge * a0), s1 = lk_half_wave_sum(ge * a1), s2 = lk_half_wave_sum(ge * a2);
                    if ((lane & 31) == LK_HWS_LANE) { part[u] = s0; part[EGP + u] = s1; part[2 * EGP + u] = s2; }
                }
            }
        }
    }
}

// ================= geometry decoder backward: one wave = one 32-sample tile =================
// part = this wave's [3][96] slice of the workgroup's d B_g partial sums (LDS)
// park = this wave's 6 x 64 u32x4 of LDS for the pieces of d y_3
// GH16: products on fp16 pieces, as the colour trunk's H16 - only where d occ is bounded: the mapper's L1 depth term has unit
// gradients (LK_FLAG_UNIT_LOSS_GRADS without ray gradients), so d occ = d depth . d depth / d occ + d colour . d colour / d occ is at most
// of the order of the sample spacing.  The chain is linear in d occ: d occ is multiplied by 2^10 once, d c and d e are scaled back where
// they leave the chain.  Half the matrix instructions of the wave's dependent chain (three per product instead of six).
template <bool GH16>
__device__ __forceinline__ void decode_bwd_geo_wave(const LkDecodeBwdArgs& a, int tile, float* __restrict__ part, u32x4* __restrict__ park) {
    typedef BwdPiece<GH16> PC;
    typedef typename PC::T Piece;
    constexpr int NP = PC::NP;
    constexpr float SC = GH16 ? 1024.0f : 1.0f, ISC = GH16 ? 1.0f / 1024.0f : 1.0f;
    const int lane = lk_lane();
    const BwdSample d = bwd_sample(a, tile, lane);
    const int h = d.h, sp = d.sp;
    const bool live = d.live;
    const float a0 = d.a0, a1 = d.a1, a2 = d.a2;
    const float* __restrict__ W = a.W;
    const u32x4* __restrict__ FB = reinterpret_cast<const u32x4*>(a.Wfrag) + (GH16 ? FRAGB_U4 : 0);
    const bool want_w = (a.flags & LK_FLAG_GRAD_WEIGHTS) != 0;
    const bool want_p = (a.flags & LK_FLAG_GRAD_RAYS) != 0;
    const float* act_geo = a.act + (size_t)sp * LK_ACT_GEO_A;
    float4 draw;
    if (a.ml_on) {
        // mapping loop: composite, loss term and composite backward here instead of in k_composite (7-9 us of every iteration); the lane of a
        // ray's first sample stores the ray's outputs, the tile's terms of the loss row go to ml_row_part
        const bool first = live && h == 0 && sp % a.S == 0;
        float G, C, N;
        draw = lk_map_draw(a.ml, sp, first, &G, &C, &N);
        if (!first) { G = 0.0f; C = 0.0f; N = 0.0f; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { G += __shfl_xor(G, o); C += __shfl_xor(C, o); N += __shfl_xor(N, o); }
        if (lane == 0) *reinterpret_cast<float4*>(a.ml_row_part + (size_t)tile * 4) = make_float4(G + (a.ml.use_color ? a.ml.w_color * C : 0.0f), G, C, N);
    } else if (a.cb_on) {
        draw = lk_cb_draw(a.cb, sp);
    } else if (!GH16 && a.tl_n_part > 0) {
        // tracking loop: the loss term of the sample's ray and the composite backward of it, here instead of in a launch of its own
        // (k_track_loss2: 6 us of an iteration of 117); the ray's first sample carries its terms to the loss row (Tracker.py:183-191)
        LkTrackRayLoss row;
        draw = lk_track_draw(a.tl, a.tl_n_part, sp, &row);
        const bool first = live && h == 0 && sp % a.S == 0;
        float G = first ? row.geo : 0.0f, C = first ? row.col : 0.0f, N = first ? row.cnt : 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { G += __shfl_xor(G, o); C += __shfl_xor(C, o); N += __shfl_xor(N, o); }
        if (lane == 0) *reinterpret_cast<float4*>(a.tl.row_part + (size_t)tile * 4) = make_float4(G + (a.tl.use_color ? a.tl.w_color * C : 0.0f), G, C, N);
    } else draw = *reinterpret_cast<const float4*>(a.d_raw + (size_t)sp * 4);
    if (!live) draw = make_float4(0.f, 0.f, 0.f, 0.f);            // dead lanes contribute nothing to reductions
    LK_STAMPW(1);
    float dpx = 0.0f, dpy = 0.0f, dpz = 0.0f;                     // this lane's share of dL/dp (embedding path)
    // ================= geometry decoder =================
    {
        const float docc = GH16 ? draw.w * SC : draw.w;
        f32x16 dh, dy, dcg, acc1;
        auto gemm2 = [&](f32x16& acc, const u32x4* fr, const f32x16& x) {        // acc += M^T x over the two 16-k blocks of a 32-wide x
#pragma unroll
            for (int G = 0; G < 2; ++G) acc = PC::mma(PC::load(fr, 1, G, 0, lane), PC::split(x, G), acc);
        };
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 wo = *reinterpret_cast<const float4*>(W + G_WO + 8 * g + 4 * h);
            dh[4 * g] = wo.x * docc; dh[4 * g + 1] = wo.y * docc; dh[4 * g + 2] = wo.z * docc; dh[4 * g + 3] = wo.w * docc;
        }
        dcg = lk_zero16();
#pragma unroll
        for (int i = 4; i >= 0; --i) {
            const u32x4* Utr = FB + PC::tr(5 + i);
            gemm2(dcg, Utr, dh);
            const f32x16 av = ct_load32(act_geo + i * 32, lane);
#pragma unroll
            for (int q = 0; q < 16; ++q) dy[q] = (av[q] > 0.0f) ? dh[q] : 0.0f;
            if (i == 4 || i == 2 || i == 1) {
                acc1 = lk_zero16();
                gemm2(acc1, FB + PC::tr(i), dy);
                dh = acc1;
            } else if (i == 3) {
                // layer 3 reads [embedding | h_2]: only the h_2 block of W_3^T d y_3 is needed now.  The three embedding blocks are
                // formed at the end, in front of layer 0's (same products in the same order - bit-identical), from the pieces of d y_3
                // parked in LDS: three accumulator tiles less are alive through layers 2..0, which is what lets the kernel run at
                // three waves per SIMD
                acc1 = lk_zero16();
#pragma unroll
                for (int G = 0; G < 2; ++G) {
                    const Piece b = PC::split(dy, G);
#pragma unroll
                    for (int q = 0; q < NP; ++q) park[(G * NP + q) * 64 + lane] = b.p[q];
                    acc1 = PC::mma(PC::load(FB + PC::tr(3), 4, G, 3, lane), b, acc1);
                }
                dh = acc1;
            }
            // i == 0: only the embedding receives gradient (below)
            LK_STAMP(7 - i);
            __builtin_amdgcn_sched_barrier(0);      // keeps the fragment loads of the layers below from being hoisted to the top (registers)
        }
        if (GH16) {
#pragma unroll
            for (int q = 0; q < 16; ++q) dcg[q] *= ISC;
        }
        ct_store32(a.dc_geo + (size_t)d.sample * LK_C, dcg, d.store, lane);
        LK_STAMP(8);
        // Embedding gradient d e = W_3[:, embedding]^T d y_3 + W_0^T d y_0, one 32-unit block at a time (one accumulator tile alive):
        // e_u = sin(x_u): ge_u = de_u cos(x_u);  dB[i][u] += sum_s ge_u a_i(s);  dp_i += ge_u 2 pi B[i][u]
        const Piece y0[2] = {PC::split(dy, 0), PC::split(dy, 1)};
        // (the flags select the form once, outside the loop: with `if (want_p)` per value the compiler sinks the whole d p chain - and with it
        // all 48 ge values and the 144 entries of B they need - to the end of the kernel: 216 registers instead of ~130)
        if (want_p && !want_w) geo_embed_bwd<true, false, GH16>(FB, W + G_EB, park, y0, ISC, a0, a1, a2, dpx, dpy, dpz, part, lane);
        else if (want_w && !want_p) geo_embed_bwd<false, true, GH16>(FB, W + G_EB, park, y0, ISC, a0, a1, a2, dpx, dpy, dpz, part, lane);
        else geo_embed_bwd<true, true, GH16>(FB, W + G_EB, park, y0, ISC, a0, a1, a2, dpx, dpy, dpz, part, lane);
    }
    if (want_p) {
        dpx += __shfl_xor(dpx, 32); dpy += __shfl_xor(dpy, 32); dpz += __shfl_xor(dpz, 32);
        if (d.store && h == 0) *reinterpret_cast<float4*>(a.dp_embed + (size_t)d.sample * 4) = make_float4(dpx, dpy, dpz, 0.0f);
    }
    LK_STAMPW(9);
}

// Hot-address atomics are the slowest thing this chip does (a few hundred distinct addresses hit by every
// wave serialise in the memory-side atomic unit): the embedding-matrix gradient is therefore reduced
// wave -> LDS -> one partial row per workgroup, and summed by k_reduce_partials.
// Block roles as in k_decode_fwd: n_col_blocks colour tiles and the geometry workgroups (4 tiles each), geometry first in the grid.
#ifndef LK_DBWD_MINB
#define LK_DBWD_MINB 2
#endif
// GH16: the geometry role on fp16 pieces too (launcher: unit-scale loss gradients and no ray gradients - the mapper's iterations)
template <bool H16, bool DEEP, bool GH16 = false>
__global__ __launch_bounds__(256, LK_DBWD_MINB) void k_decode_bwd(LkDecodeBwdArgs a, int n_col_blocks) {
    __shared__ u32x4 s_x[2 * 24 * 64];
    __shared__ float s_o[4][3 * 32];
    const int w = (int)threadIdx.x >> 6;
    LK_STAMP(0);
    const int P_live = a.live_rays ? min(a.P, *a.live_rays * a.S) : a.P;       // as k_decode_fwd
    // geometry workgroups FIRST in the grid: a geometry tile is one wave's 20-us chain - dispatched behind the colour tiles (as in rounds 1-2)
    // those chains were the launch's tail on a nearly empty chip; in front they run beside the colour tiles (5 000 rays: 60 -> 55 us)
    const int n_geo_blocks = (int)gridDim.x - n_col_blocks;
    const int bid = (int)blockIdx.x < n_geo_blocks ? n_col_blocks + (int)blockIdx.x : (int)blockIdx.x - n_geo_blocks;
    if (bid < n_col_blocks) {
        if (bid * 32 >= P_live) {       // a skipped tile contributes zero to the per-tile sums of d affine
            if (a.affine && a.g_affine && threadIdx.x < 12) a.g_affine_part[(size_t)bid * 12 + threadIdx.x] = 0.0f;
            return;
        }
        decode_bwd_col_wg<H16, DEEP, !GH16>(a, bid, w, lk_lane(), s_x, s_o);
        return;
    }
    float (*s_part)[3 * EGP] = reinterpret_cast<float (*)[3 * EGP]>(s_x);
    const bool want_w = (a.flags & LK_FLAG_GRAD_WEIGHTS) != 0;
    const int gb = bid - n_col_blocks;
    if (want_w) {
        for (int e = threadIdx.x; e < 4 * 3 * EGP; e += 256) (&s_part[0][0])[e] = 0.0f;
        __syncthreads();
    }
    const int tile = gb * 4 + w;
    if (tile * 32 < P_live) decode_bwd_geo_wave<GH16>(a, tile, s_part[w], s_x + 512 + w * (6 * 64));       // (s_part ends at s_x[288])
    if (want_w) {
        __syncthreads();
        for (int e = threadIdx.x; e < 3 * EGP; e += 256)
            a.part_bg[(size_t)gb * (3 * EGP) + e] = s_part[0][e] + s_part[1][e] + s_part[2][e] + s_part[3][e];
    }
}

// out[j] += sum_p part[p][j].  One workgroup per (32-column group, 16 partial rows): 8 row lanes x 32 columns, two
// rows per lane read coalesced, combined through LDS, then ONE atomic per (workgroup, column) - a few tens to a
// hundred adds per address in total (~13 ns each when they collide), and no serial loop over the rows.
__global__ __launch_bounds__(256) void k_reduce_partials(const float* __restrict__ part, int n_parts, int width, float* __restrict__ out) {
    __shared__ float sh[8][32];
    const int e = (int)threadIdx.x & 31, q = (int)threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + e;
    const int p0 = blockIdx.y * 16 + q;
    float s = 0.0f;
    if (col < width) {
        if (p0 < n_parts) s = part[(size_t)p0 * width + col];
        if (p0 + 8 < n_parts) s += part[(size_t)(p0 + 8) * width + col];
    }
    sh[q][e] = s;
    __syncthreads();
    if (q == 0 && col < width)
        atomicAdd(out + col, ((sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e])) + ((sh[4][e] + sh[5][e]) + (sh[6][e] + sh[7][e])));
}

int lk_launch_reduce_partials(const float* part, int n_parts, int width, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_partials, dim3(lk_cdiv(width, 32), lk_cdiv(n_parts, 16)), dim3(256), 0, st, part, n_parts, width, out);
    return LK_OK;
}

int lk_launch_composite_bwd(const LkCompositeBwdArgs& a, hipStream_t st) {
    LkProfScope prof_(LKK_COMPOSITE_BWD, st);
    hipLaunchKernelGGL(k_composite_bwd, dim3(lk_cdiv(a.R, 256)), dim3(256), 0, st, a);
    return LK_OK;
}
int lk_launch_decode_bwd(const LkDecodeBwdArgs& a, hipStream_t st) {
    // two timer names: launches WITH ray gradients (the tracker's, and a BA-mode mapper's) are "k_decode_bwd_track", launches without
    // them (the mapper's) "k_decode_bwd" - whatever pieces they run on: profile.work_per_step books every mapping iteration under the
    // mapper's name, and a mapper launch without unit-scale loss gradients (exposure encoding with the rel-pos MLP, a standalone
    // lk_render_bwd) used to land under the tracker's (round-5 advisor)
    LkProfScope prof_((a.flags & LK_FLAG_GRAD_RAYS) ? LKK_DECODE_BWD_TRACK : LKK_DECODE_BWD, st);
    const int tiles = lk_cdiv(a.P, 32);
    const int n_col = (a.flags & LK_FLAG_STAGE_COLOR) ? tiles : 0;
    // fp16 pieces only where the COLOUR loss gradients have unit scale (see decode_bwd_col_wg): the mapper's L1 sums, and the tracker's
    // loss too - its d colour is w_color sgn(.), only its d depth = 1 / sqrt(var) is unbounded, and that reaches the geometry decoder
    // alone (d raw[:, 3]), whose backward stays on bf16 pieces
    const bool h16 = (a.flags & LK_FLAG_UNIT_LOSS_GRADS) != 0;
    const bool deep = n_col > 0 && n_col <= LK_DEEP_MAX_TILES;
    const dim3 grid(n_col + lk_cdiv(tiles, 4));
    // the geometry decoder's backward follows where d depth is bounded as well: unit-scale loss gradients WITHOUT ray gradients (with them
    // the caller is the tracker, whose d depth = 1 / sqrt(var) is not)
    // (and not behind a tracker-mode forward, whose saved activations have the fp32 layout only the other instantiations read)
    const bool gh16 = h16 && !(a.flags & (LK_FLAG_GRAD_RAYS | LK_FLAG_TRACKER));
    if (gh16 && deep) hipLaunchKernelGGL((k_decode_bwd<true, true, true>), grid, dim3(256), 0, st, a, n_col);
    else if (gh16) hipLaunchKernelGGL((k_decode_bwd<true, false, true>), grid, dim3(256), 0, st, a, n_col);
    else if (h16 && deep) hipLaunchKernelGGL((k_decode_bwd<true, true>), grid, dim3(256), 0, st, a, n_col);
    else if (h16) hipLaunchKernelGGL((k_decode_bwd<true, false>), grid, dim3(256), 0, st, a, n_col);
    else if (deep) hipLaunchKernelGGL((k_decode_bwd<false, true>), grid, dim3(256), 0, st, a, n_col);
    else hipLaunchKernelGGL((k_decode_bwd<false, false>), grid, dim3(256), 0, st, a, n_col);
    return LK_OK;
}
int lk_occupancy_decode_bwd() {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_decode_bwd<true, false, true>, 256, 0);
    return n;
}
