// Fused decoder forward on the matrix cores (compute-bound half of the render forward).
//   reference: GaussianFourierFeatureTransform (src/conv_onet/models/decoder.py:12-43),
//              MLP_geometry.forward (decoder.py:263-288), MLP_color.forward (decoder.py:513-546),
//              MLP_col_neighbor / rel-pos branch of MLP_color.get_feature_at_pos (decoder.py:477-490),
//              NICER.forward stage dispatch (decoder.py:573-610).
//
// Design (CDNA4): a tile is 32 sample points.  Activations are kept TRANSPOSED in the MFMA C/D layout ("CT tile":
// 32 units x 32 samples, lane = sample column) so that each layer Y^T = W X^T takes A = weights (16-byte loads of the
// split-bf16 fragment blob) and B = the previous layer's accumulator registers, eight registers per 16-k matrix
// instruction; every fp32 product is six bf16 piece products on the bf16 pipe (lk_common.h: lk_mma6, fp32-class accuracy).
// Geometry decoder and rel-pos MLP: one wave per tile, whole network through registers, no LDS, no barrier.  Colour
// trunk (128 wide): the four waves of a workgroup own 32 output units each and exchange their tiles through LDS as
// lane-contiguous split pieces (decode_col_wg).
#include "lk_common.h"
#include "lk_kernels.h"
#include "lk_composite_dev.h"
#include <string.h>

using namespace lkw;

LK_CHAIN_DEFINE(fwd)

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 ct_load_rows32(const float* __restrict__ row /* 32 floats of this sample */,
                                                 bool live, int lane) {
    const int h = lane >> 5;
    f32x16 t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) v = *reinterpret_cast<const float4*>(row + 8 * g + 4 * h);
        t[4 * g + 0] = v.x; t[4 * g + 1] = v.y; t[4 * g + 2] = v.z; t[4 * g + 3] = v.w;
    }
    return t;
}

__device__ __forceinline__ void ct_store_rows32(float* __restrict__ row, const f32x16& t, bool live, int lane) {
    const int h = lane >> 5;
    if (!live) return;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(row + 8 * g + 4 * h) = make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
}

// LK_FLAG_CHECK_RANGE (status != NULL, wave-uniform): a tile that is about to be cut into fp16 pieces (lk_split_cth) is tested against fp16's
// ceiling - the pack-convert saturates silently there (loopy_hip.h: LK_STATUS_ACT_RANGE)
// CHECK is a template parameter of the kernels: the test costs the default forward 8 registers otherwise (173 > the 168 of three workgroups
// per compute unit); launches with the flag take the <.., true> instantiations (and never the fused tracker launch)
template <bool CHECK>
__device__ __forceinline__ void ct_check_range(const f32x16& t, unsigned* status) {
    if (!CHECK || !status) return;
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) bad = bad || lk_out_of_range(t[r], 65504.0f);
    if (bad) lk_status_raise(status, LK_STATUS_ACT_RANGE);
}

// geometry embedding tile b (units 32b..32b+31 of sin((2*pi*p) @ B_g), B_g padded [3][96])
__device__ __forceinline__ f32x16 geo_embed_tile(const float* __restrict__ B, int b, float a0, float a1, float a2, int lane) {
    const int h = lane >> 5;
    f32x16 e;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int u0 = 32 * b + 8 * g + 4 * h;
        const float4 b0 = *reinterpret_cast<const float4*>(B + u0);
        const float4 b1 = *reinterpret_cast<const float4*>(B + EGP + u0);
        const float4 b2 = *reinterpret_cast<const float4*>(B + 2 * EGP + u0);
        e[4 * g + 0] = (u0 + 0 < EG) ? lk_sinf(lk_fourier_arg(a0, a1, a2, b0.x, b1.x, b2.x)) : 0.0f;
        e[4 * g + 1] = (u0 + 1 < EG) ? lk_sinf(lk_fourier_arg(a0, a1, a2, b0.y, b1.y, b2.y)) : 0.0f;
        e[4 * g + 2] = (u0 + 2 < EG) ? lk_sinf(lk_fourier_arg(a0, a1, a2, b0.z, b1.z, b2.z)) : 0.0f;
        e[4 * g + 3] = (u0 + 3 < EG) ? lk_sinf(lk_fourier_arg(a0, a1, a2, b0.w, b1.w, b2.w)) : 0.0f;
    }
    return e;
}

// [sin(x_0..n-1), cos(x_0..n-1)] embedding unit u of a [3][n] matrix (colour: n = 20, rel-pos: n = 10)
__device__ __forceinline__ float sincos_embed_unit(const float* __restrict__ B, int n, int u, float a0, float a1, float a2) {
    if (u >= 2 * n) return 0.0f;
    const int xi = (u < n) ? u : u - n;
    const float x = lk_fourier_arg(a0, a1, a2, B[xi], B[n + xi], B[2 * n + xi]);
    return (u < n) ? lk_sinf(x) : lk_cosf(x);
}

template <int NG>
__device__ __forceinline__ f32x16 sincos_embed_tile(const float* __restrict__ B, int n, int b, float a0, float a1, float a2, int lane) {
    const int h = lane >> 5;
    f32x16 e = lk_zero16();
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            e[4 * g + t] = sincos_embed_unit(B, n, 32 * b + 8 * g + 4 * h + t, a0, a1, a2);
    return e;
}

// bias + activation (+ optional save) + fc_c(c): h = act(acc + b) + (U c + u); c arrives as its two split blocks
template <int NB, bool SOFTPLUS>
__device__ __forceinline__ void layer_finish(f32x16 (&acc)[NB],
                                             const u32x4* __restrict__ UfragB, const float* __restrict__ ubias,
                                             const LkH8 (&cb)[2], float* __restrict__ save_a, bool live, int lane) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {          // acc started from the layer's bias (lk_rowvec_tile)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = SOFTPLUS ? lk_softplus100(acc[nb][r]) : fmaxf(acc[nb][r], 0.0f);
        if (save_a) ct_store_rows32(save_a + nb * 32, acc[nb], live, lane);
        lk_add_rowvec(acc[nb], ubias, nb * 32, lane);
#pragma unroll
        for (int G = 0; G < 2; ++G) acc[nb] = lk_mma3h(lk_fragh_load(UfragB, NB, G, nb, lane), cb[G], acc[nb]);
    }
}

// ---------------------------------------------------------------------------------------------
// Sample coordinates shared by both decoder roles: lane -> sample of the tile, p = o + d z, a = fl(2 pi p)
struct DecSample {
    int sample, sp, h;
    bool live;
    float a0, a1, a2;
};
__device__ __forceinline__ DecSample dec_sample(const LkDecodeArgs& a, int tile, int lane) {
    DecSample d;
    const int ts = a.tile_stride ? a.tile_stride : 32;           // (a tile of whole rays: LkDecodeArgs::tile_stride)
    d.sample = tile * ts + (lane & 31);
    d.live = d.sample < a.P && (lane & 31) < ts;
    d.h = lane >> 5;
    d.sp = d.live ? d.sample : a.P - 1;                          // clamp: dead lanes compute, never store
    const int r = d.sp / a.S;
    const float z = a.z[d.sp];
    const float px = lk_madd_rn(a.rays_o[3 * r], a.rays_d[3 * r], z);
    const float py = lk_madd_rn(a.rays_o[3 * r + 1], a.rays_d[3 * r + 1], z);
    const float pz = lk_madd_rn(a.rays_o[3 * r + 2], a.rays_d[3 * r + 2], z);
    d.a0 = __fmul_rn(LK_TWO_PI, px); d.a1 = __fmul_rn(LK_TWO_PI, py); d.a2 = __fmul_rn(LK_TWO_PI, pz);
    return d;
}

// ================= geometry decoder (hidden 32, relu): one wave = one 32-sample tile, registers only =================
template <bool CHECK = false>
__device__ __forceinline__ float decode_geo_wave(const LkDecodeArgs& a, int tile, int lane) {      // returns the occupancy of lane & 31's sample
    const DecSample d = dec_sample(a, tile, lane);
    const int h = d.h, sp = d.sp;
    const bool live = d.live;
    const float a0 = d.a0, a1 = d.a1, a2 = d.a2;
    const float* __restrict__ W = a.W;
    const u32x4* __restrict__ FB = reinterpret_cast<const u32x4*>(a.Wfrag) + FRAGB_U4;      // fp16 forward fragments
    const bool save = (a.flags & LK_FLAG_SAVE_ACT) && a.act != nullptr;
    float* act_geo = save ? a.act + (size_t)sp * LK_ACT_GEO_A : nullptr;
    // the 96 embedding units and the interpolated feature are B operands twice / five times: split once
    LkH8 eb[6], cb[2];
    {
        const f32x16 e0 = geo_embed_tile(W + G_EB, 0, a0, a1, a2, lane);
        eb[0] = lk_split_cth(e0, 0); eb[1] = lk_split_cth(e0, 1);
        const f32x16 e1 = geo_embed_tile(W + G_EB, 1, a0, a1, a2, lane);
        eb[2] = lk_split_cth(e1, 0); eb[3] = lk_split_cth(e1, 1);
        const f32x16 e2 = geo_embed_tile(W + G_EB, 2, a0, a1, a2, lane);
        eb[4] = lk_split_cth(e2, 0); eb[5] = lk_split_cth(e2, 1);
        const f32x16 cg = ct_load_rows32(a.c_geo + (size_t)sp * LK_C, true, lane);
        ct_check_range<CHECK>(cg, a.status);
        cb[0] = lk_split_cth(cg, 0); cb[1] = lk_split_cth(cg, 1);
    }
    LK_STAMPW(5);                                    // (probe build) geometry wave: embedding evaluated, c_geo arrived
    f32x16 acc[1], hh;
    // layer 0: 93 -> 32
    acc[0] = lk_rowvec_tile(W + G_B0, 0, lane);
#pragma unroll
    for (int G = 0; G < 6; ++G) acc[0] = lk_mma3h(lk_fragh_load(FB + FM0_FWDH, 1, G, 0, lane), eb[G], acc[0]);
    layer_finish<1, false>(acc, FB + FM5_FWDH, W + G_U0 + a64(HG * CF), cb, act_geo, live, lane);
    hh = acc[0]; ct_check_range<CHECK>(hh, a.status);
    LK_STAMP(6);
    // layers 1, 2: 32 -> 32
    acc[0] = lk_rowvec_tile(W + G_B1, 0, lane);
    lk_gemm_h3<1, 2>(acc, FB + FM1_FWDH, 1, 0, 0, hh, 0, lane);
    layer_finish<1, false>(acc, FB + FM6_FWDH, W + G_U0 + G_USTRIDE + a64(HG * CF), cb,
                           act_geo ? act_geo + 32 : nullptr, live, lane);
    hh = acc[0]; ct_check_range<CHECK>(hh, a.status);
    LK_STAMP(7);
    acc[0] = lk_rowvec_tile(W + G_B2, 0, lane);
    lk_gemm_h3<1, 2>(acc, FB + FM2_FWDH, 1, 0, 0, hh, 0, lane);
    layer_finish<1, false>(acc, FB + FM7_FWDH, W + G_U0 + 2 * G_USTRIDE + a64(HG * CF), cb,
                           act_geo ? act_geo + 64 : nullptr, live, lane);
    hh = acc[0]; ct_check_range<CHECK>(hh, a.status);
    LK_STAMP(8);
    // layer 3 (skip): [e(93) | h(32)] -> 32, packed as [96 | 32]
    acc[0] = lk_rowvec_tile(W + G_B3, 0, lane);
#pragma unroll
    for (int G = 0; G < 6; ++G) acc[0] = lk_mma3h(lk_fragh_load(FB + FM3_FWDH, 1, G, 0, lane), eb[G], acc[0]);
    lk_gemm_h3<1, 2>(acc, FB + FM3_FWDH, 1, 6, 0, hh, 0, lane);
    layer_finish<1, false>(acc, FB + FM8_FWDH, W + G_U0 + 3 * G_USTRIDE + a64(HG * CF), cb,
                           act_geo ? act_geo + 96 : nullptr, live, lane);
    hh = acc[0]; ct_check_range<CHECK>(hh, a.status);
    LK_STAMP(9);
    // layer 4
    acc[0] = lk_rowvec_tile(W + G_B4, 0, lane);
    lk_gemm_h3<1, 2>(acc, FB + FM4_FWDH, 1, 0, 0, hh, 0, lane);
    layer_finish<1, false>(acc, FB + FM9_FWDH, W + G_U0 + 4 * G_USTRIDE + a64(HG * CF), cb,
                           act_geo ? act_geo + 128 : nullptr, live, lane);
    // output 32 -> 1 on the VALU: each half-wave holds 16 of the 32 units of its sample
    float part = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 wo = *reinterpret_cast<const float4*>(W + G_WO + 8 * g + 4 * h);
        part = fmaf(wo.x, acc[0][4 * g], part); part = fmaf(wo.y, acc[0][4 * g + 1], part);
        part = fmaf(wo.z, acc[0][4 * g + 2], part); part = fmaf(wo.w, acc[0][4 * g + 3], part);
    }
    part += __shfl_xor(part, 32);
    if (live && h == 0) a.raw[(size_t)d.sample * 4 + 3] = part + W[G_BO];
    LK_STAMP(10);
    return part + W[G_BO];
}

// The same decoder with its operands in LDS (the tracker's fused launch, k_relpos_decode_fwd: one tile per compute unit, and the geometry wave's
// chain is what the tile's composite waits for).  In decode_geo_wave a layer is two dependent global round trips - its W fragments and bias,
// then (behind the activation store: loads issued after a store are not moved above it) its fc_c fragments and bias - 2.3-2.6 us per 32-wide
// layer whose matrix work is 0.3 us, 7 us for layer 0 (profiles/r6_track_chain_before.md).  The launch's eight waves copy the decoder's 30
// forward fragment blocks (60 KB, fp16 pieces) and ten bias vectors into LDS while they wait for the rel-pos MLP's inputs (lk_geo_stage:
// global_load_lds, no register in between), and this wave reads them from there.  Same products in the same order: bit-identical.
// s_gw: blocks in the order of use - W0 (6), U0 (2), W1, U1, W2, U2 (2 each), W3 (8), U3, W4, U4; s_gb: b_0..b_4, u_0..u_4
#define LK_GEO_STAGE_BLOCKS 30
#define LK_GLDS_MAX_TILES 256           // compute units of the chip: one tile each
__device__ __forceinline__ void lk_geo_stage(const LkDecodeArgs& a, int w, int lane, u32x4* __restrict__ s_gw, float (*s_gb)[32]) {
    const u32x4* __restrict__ FB = reinterpret_cast<const u32x4*>(a.Wfrag) + FRAGB_U4;
    constexpr int SRC[10] = {FM0_FWDH, FM5_FWDH, FM1_FWDH, FM6_FWDH, FM2_FWDH, FM7_FWDH, FM3_FWDH, FM8_FWDH, FM4_FWDH, FM9_FWDH};
    constexpr int NBLK[10] = {6, 2, 2, 2, 2, 2, 8, 2, 2, 2};
    int chunk = 0;                      // 1-KB chunks (one piece of one block = one wave-wide 16-byte access), dealt round-robin to the 8 waves
#pragma unroll
    for (int m = 0; m < 10; ++m) {
#pragma unroll
        for (int c = 0; c < 2 * NBLK[m]; ++c, ++chunk) {
            if ((chunk & 7) == w) {
                __builtin_amdgcn_global_load_lds(FB + SRC[m] + c * 64 + lane, (__attribute__((address_space(3))) u32x4*)(s_gw + chunk * 64), 16, 0, 0);
            }
        }
    }
    const int t = w * 64 + lane;
    if (t < 320) {
        const int j = t >> 5, u = t & 31;
        const int b_off[5] = {G_B0, G_B1, G_B2, G_B3, G_B4};
        s_gb[j][u] = a.W[j < 5 ? b_off[j] + u : G_U0 + (j - 5) * G_USTRIDE + a64(HG * CF) + u];
    }
}
template <bool CHECK = false>
__device__ __forceinline__ float decode_geo_wave_lds(const LkDecodeArgs& a, int tile, int lane, const u32x4* __restrict__ s_gw, const float (*s_gb)[32]) {
    const DecSample d = dec_sample(a, tile, lane);
    const int h = d.h, sp = d.sp;
    const bool live = d.live;
    const float a0 = d.a0, a1 = d.a1, a2 = d.a2;
    const float* __restrict__ W = a.W;
    const bool save = (a.flags & LK_FLAG_SAVE_ACT) && a.act != nullptr;
    float* act_geo = save ? a.act + (size_t)sp * LK_ACT_GEO_A : nullptr;
    constexpr int WB[5] = {0, 8, 12, 16, 26}, UB[5] = {6, 10, 14, 24, 28};      // first block of W_i / U_i in s_gw
    auto frag = [&](int block) {
        LkH8 f;
        f.p[0] = s_gw[(block * 2 + 0) * 64 + lane]; f.p[1] = s_gw[(block * 2 + 1) * 64 + lane];
        return f;
    };
    auto bias = [&](int j) {
        f32x16 t;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b = *reinterpret_cast<const float4*>(&s_gb[j][8 * g + 4 * h]);
            t[4 * g + 0] = b.x; t[4 * g + 1] = b.y; t[4 * g + 2] = b.z; t[4 * g + 3] = b.w;
        }
        return t;
    };
    LkH8 eb[6], cb[2];
    {
        const f32x16 e0 = geo_embed_tile(W + G_EB, 0, a0, a1, a2, lane);
        eb[0] = lk_split_cth(e0, 0); eb[1] = lk_split_cth(e0, 1);
        const f32x16 e1 = geo_embed_tile(W + G_EB, 1, a0, a1, a2, lane);
        eb[2] = lk_split_cth(e1, 0); eb[3] = lk_split_cth(e1, 1);
        const f32x16 e2 = geo_embed_tile(W + G_EB, 2, a0, a1, a2, lane);
        eb[4] = lk_split_cth(e2, 0); eb[5] = lk_split_cth(e2, 1);
        const f32x16 cg = ct_load_rows32(a.c_geo + (size_t)sp * LK_C, true, lane);
        ct_check_range<CHECK>(cg, a.status);
        cb[0] = lk_split_cth(cg, 0); cb[1] = lk_split_cth(cg, 1);
    }
    LK_STAMPW(5);
    f32x16 acc, hh = lk_zero16();
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        acc = bias(i);
        if (i == 0 || i == 3) {         // [embedding (6 blocks) | hidden (layer 3: 2 blocks)]
#pragma unroll
            for (int G = 0; G < 6; ++G) acc = lk_mma3h(frag(WB[i] + G), eb[G], acc);
            if (i == 3) {
#pragma unroll
                for (int G = 0; G < 2; ++G) acc = lk_mma3h(frag(WB[i] + 6 + G), lk_split_cth(hh, G), acc);
            }
        } else {
#pragma unroll
            for (int G = 0; G < 2; ++G) acc = lk_mma3h(frag(WB[i] + G), lk_split_cth(hh, G), acc);
        }
        // epilogue: relu, saved row, + u_i, + U_i c
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.0f);
        if (act_geo) ct_store_rows32(act_geo + 32 * i, acc, live, lane);
        {
            const f32x16 ub = bias(5 + i);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += ub[r];
        }
#pragma unroll
        for (int G = 0; G < 2; ++G) acc = lk_mma3h(frag(UB[i] + G), cb[G], acc);
        LK_STAMP(6 + i);
        hh = acc;
        if (i < 4) ct_check_range<CHECK>(hh, a.status);
    }
    // output 32 -> 1 on the VALU: each half-wave holds 16 of the 32 units of its sample
    float part = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 wo = *reinterpret_cast<const float4*>(W + G_WO + 8 * g + 4 * h);
        part = fmaf(wo.x, acc[4 * g], part); part = fmaf(wo.y, acc[4 * g + 1], part);
        part = fmaf(wo.z, acc[4 * g + 2], part); part = fmaf(wo.w, acc[4 * g + 3], part);
    }
    part += __shfl_xor(part, 32);
    if (live && h == 0) a.raw[(size_t)d.sample * 4 + 3] = part + W[G_BO];
    return part + W[G_BO];
}

// ================= colour decoder (hidden 128, softplus beta=100): FOUR waves = one 32-sample tile =================
// Wave w owns output units [32w, 32w+32) of every layer (one accumulator, 48 matrix instructions per 128-wide layer), so
// a tile's serial chain is a quarter of the one-wave form and a training batch (a few hundred tiles) spreads over four
// times as many SIMDs.  The next layer needs all 128 units as its B operand: each wave SPLITS its activated CT tile and
// parks the bf16 pieces in LDS, [wave][16-k block][piece][lane] — the reader of block (w', G) is the SAME lane id in every
// wave (the C/D-row walk), so writes and reads are both lane-contiguous (conflict-free ds_write/read_b128).
// Double-buffered: one barrier per layer.
// s_bias: the ten bias vectors of the trunk (b_0..b_4, u_0..u_4), staged once per workgroup: fetched in line they were two global
// loads per layer, each waiting behind the weight fragments prefetched just before them (loads return in order) - 1-2 k cycles of
// a 9 k-cycle layer (tools/probe/decode_clock.py).
__device__ __forceinline__ f32x16 ct_bias_lds(const float* __restrict__ v, int unit0, int lane) {
    const int h = lane >> 5;
    f32x16 t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(v + unit0 + 8 * g + 4 * h);
        t[4 * g + 0] = b.x; t[4 * g + 1] = b.y; t[4 * g + 2] = b.z; t[4 * g + 3] = b.w;
    }
    return t;
}
// DEEP (launches whose tiles are all resident at two workgroups per compute unit - the tracker's 235 tiles): ALL eight hidden blocks of
// the next layer (and the three embedding blocks of layer 3) are fetched before the stores of the current one, 40 registers more;
// with one tile per compute unit nothing else hides the L2 round trip of the in-line blocks (decode_clock.py: 2.5-5 k cycles per
// product phase for 1 k cycles of matrix instructions).
// SOFTBAR (s_cnt: an LDS word, zero on entry): the four waves meet at a barrier of their own (lk_soft_barrier) instead of s_barrier - for
// workgroups in which a fifth wave runs something else meanwhile (k_relpos_decode_fwd: the geometry decoder on wave 4)
// s_raw (or NULL; wave 0 only reads it): the tile's colours are also left there as raw rows [32][4] (k_relpos_decode_fwd's composite epilogue)
// The part of a colour tile's set-up that does NOT depend on the interpolated colour feature: the ten bias vectors into s_bias, the wave's
// share of the forty sin / cos values as fp16 pieces into s_x[1] (and as rows for the weight gradients).  decode_col_wg runs it first thing;
// the tracker's fused launch (k_relpos_decode_fwd) runs it IN FRONT of the rel-pos MLP, whose output is the colour feature - behind it, as
// until round 6, its cold fetches (biases, Fourier matrix, the sample's depth and ray) were 2.9 us of every tile's chain
// (profiles/r6_track_chain_before.md, "decoder set-up").  Its LDS writes are ordered before their readers by decode_col_wg's first barrier.
__device__ __forceinline__ void decode_col_setup(const LkDecodeArgs& a, int tile, int w, int lane, u32x4 (*s_x)[16 * 64], float (*s_bias)[128]) {
    {
        const int t = (int)threadIdx.x;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int e = q * 256 + t, j = e >> 7, u = e & 127;           // vector j: b_j (j < 5) or u_(j-5)
            const int b_off[5] = {C_B0, C_B1, C_B2, C_B3, C_B4};
            s_bias[j][u] = a.W[j < 5 ? b_off[j] + u : C_U0 + (j - 5) * C_USTRIDE + a64(HC * CF) + u];
        }
    }
    const DecSample d = dec_sample(a, tile, lane);
    const int h = d.h, sp = d.sp;
    const bool live = d.live;
    const float a0 = d.a0, a1 = d.a1, a2 = d.a2;
    const float* __restrict__ W = a.W;
    const bool save = (a.flags & LK_FLAG_SAVE_ACT) && a.act != nullptr;
    // embedding (40 units = blocks 0, 1 and a quarter of 2): B operand of two products, split once.
    // The forty sin / cos values are the same for the four waves: wave w evaluates register group g = w of block 0 (units 8 w + 4 h + t),
    // wave 3 also block 1's four, and the fp16 pieces travel through s_x[1] (an activation buffer from layer 1's epilogue on, two
    // barriers later) - 4 to 8 evaluations per lane instead of 20, a fifth of the kernel's VALU instructions.
    float ev[4], e1v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < 4; ++t) ev[t] = sincos_embed_unit(W + C_EB, 20, 8 * w + 4 * h + t, a0, a1, a2);
    if (w == 3) {
#pragma unroll
        for (int t = 0; t < 4; ++t) e1v[t] = sincos_embed_unit(W + C_EB, 20, 32 + 4 * h + t, a0, a1, a2);
    }
    if (save && live) {              // embedding rows 0..39 (input of layers 0 and 3) for the weight gradients
        float* erow = a.act + (size_t)a.P * (LK_ACT_GEO_A + LK_ACT_COL_A + LK_ACT_COL_H) + (size_t)sp * LK_ACT_COL_E;
        *reinterpret_cast<float4*>(erow + 8 * w + 4 * h) = make_float4(ev[0], ev[1], ev[2], ev[3]);
        if (w == 3) *reinterpret_cast<float4*>(erow + 32 + 4 * h) = make_float4(e1v[0], e1v[1], e1v[2], e1v[3]);
    }
    // the same cut as lk_split8h: hi = rtz_f16 of the pair, lo = rtz_f16 of the remainders
    auto cut2 = [&](float x, float y, unsigned& hi, unsigned& lo) {
        const lk_f16x2 hh = __builtin_amdgcn_cvt_pkrtz(x, y);
        const lk_f16x2 ll = __builtin_amdgcn_cvt_pkrtz(x - (float)hh[0], y - (float)hh[1]);
        hi = __builtin_bit_cast(unsigned, hh); lo = __builtin_bit_cast(unsigned, ll);
    };
    unsigned* se = reinterpret_cast<unsigned*>(s_x[1]);
    unsigned hi0, lo0, hi1, lo1;
    cut2(ev[0], ev[1], hi0, lo0); cut2(ev[2], ev[3], hi1, lo1);
    const int G = w >> 1, c0 = 2 * (w & 1);
    *reinterpret_cast<uint2*>(se + ((G * 2 + 0) * 64 + lane) * 4 + c0) = make_uint2(hi0, hi1);
    *reinterpret_cast<uint2*>(se + ((G * 2 + 1) * 64 + lane) * 4 + c0) = make_uint2(lo0, lo1);
    if (w == 3) {
        cut2(e1v[0], e1v[1], hi0, lo0); cut2(e1v[2], e1v[3], hi1, lo1);
        s_x[1][(2 * 2 + 0) * 64 + lane] = u32x4{hi0, hi1, 0u, 0u};
        s_x[1][(2 * 2 + 1) * 64 + lane] = u32x4{lo0, lo1, 0u, 0u};
    }
}

// PRESET: the caller has run decode_col_setup for this tile already (k_relpos_decode_fwd)
template <bool DEEP, bool SOFTBAR = false, bool CHECK = false, bool PRESET = false>
__device__ __forceinline__ void decode_col_wg(const LkDecodeArgs& a, int tile, int w, int lane,
                                              u32x4 (*s_x)[16 * 64] /* [2][16*64] */, float (*s_o)[3 * 32] /* [4][96] */,
                                              float (*s_bias)[128] /* [10][128] */, unsigned* s_cnt = nullptr, float* s_raw = nullptr) {
    unsigned n_bar = 0;
    auto wg_barrier = [&]() {
        if (SOFTBAR) lk_soft_barrier(s_cnt, 4u * (++n_bar));
        else __syncthreads();
    };
    if (!PRESET) decode_col_setup(a, tile, w, lane, s_x, s_bias);
    const DecSample d = dec_sample(a, tile, lane);
    const int h = d.h, sp = d.sp;
    const bool live = d.live;
    const float a0 = d.a0, a1 = d.a1, a2 = d.a2;
    const float* __restrict__ W = a.W;
    const u32x4* __restrict__ FB = reinterpret_cast<const u32x4*>(a.Wfrag) + FRAGB_U4;      // fp16 forward fragments
    const bool save = (a.flags & LK_FLAG_SAVE_ACT) && a.act != nullptr;
    // this sample's row of layer 0; layer L is LK_COL_LAYER(P, L) floats further (layer-major, lk_kernels.h)
    // (the derivative mask: 64 words per sample and layer; wave w, lane half h own words [16 w + 8 h, + 8) - a lane's 16 values are contiguous)
    unsigned* act_col_s = save ? reinterpret_cast<unsigned*>(a.act + (size_t)a.P * LK_ACT_GEO_A) + (size_t)sp * 64 + 16 * w + 8 * h : nullptr;
    // tracker mode (lk_kernels.h: LK_ACT_COL_A): the fp32 a_i rows instead, [layer][P][128] in the same region
    const bool a32 = (a.flags & LK_FLAG_TRACKER) != 0;
    float* act_col_a = save ? a.act + (size_t)a.P * LK_ACT_GEO_A + (size_t)sp * 128 + w * 32 : nullptr;
    float* act_col_h = save ? a.act + (size_t)a.P * (LK_ACT_GEO_A + LK_ACT_COL_A) + (size_t)sp * 128 : nullptr;
    // the embedding pieces are in s_x[1] (decode_col_setup); the interpolated feature is the B operand of five products: split once
    LkH8 eb[3], cb[2];
    {
        const f32x16 cc = ct_load_rows32(a.c_col + (size_t)sp * LK_C, true, lane);
        ct_check_range<CHECK>(cc, a.status);
        cb[0] = lk_split_cth(cc, 0); cb[1] = lk_split_cth(cc, 1);
    }
    LK_STAMPW(14);                                   // (probe build) this wave's set-up: sample, embedding share, c_col arrived
    wg_barrier();                                   // the embedding pieces and s_bias are complete
    LK_STAMP(5);
#pragma unroll
    for (int G = 0; G < 3; ++G) {
        eb[G].p[0] = s_x[1][(G * 2 + 0) * 64 + lane];
        eb[G].p[1] = s_x[1][(G * 2 + 1) * 64 + lane];
    }
    // a block travels to the other waves through LDS as SPLIT pieces (the producer splits once, the four consumers read
    // bf16): block (w, G), piece p at [((w*2 + G)*3 + p)*64 + lane], lane-contiguous 16-byte accesses both ways
    // Loads and stores share one in-order counter on this chip: a weight fragment fetched AFTER the activation stores of
    // a layer cannot be waited for without waiting for those stores too.  So everything a layer's epilogue and the head
    // of the next product need is fetched BEFORE the stores (wn: first four hidden blocks of the next layer, un: fc_c),
    // pinned with scheduling barriers; the tail of the product (blocks 4..7) is fetched in line, long after the stores.
    constexpr int NPF = DEEP ? 8 : 4;
    LkH8 wn[NPF], un[2], we[DEEP ? 3 : 1];
    auto prefetch_hidden = [&](const u32x4* fragb, int G0) {
#pragma unroll
        for (int G = 0; G < NPF; ++G) wn[G] = lk_fragh_load(fragb, 4, G0 + G, w, lane);
        if (DEEP && G0 == 3) {
#pragma unroll
            for (int G = 0; G < 3; ++G) we[G] = lk_fragh_load(fragb, 4, G, w, lane);
        }
    };
    auto prefetch_u = [&](const u32x4* ufragb) {
#pragma unroll
        for (int G = 0; G < 2; ++G) un[G] = lk_fragh_load(ufragb, 4, G, w, lane);
    };
    // acc += W[own block][hidden 128] h with h read from LDS; G0 = first 16-k block of the hidden part in the matrix
    auto hidden = [&](f32x16& acc, const u32x4* fragb, int G0, int buf) {
#pragma unroll
        for (int G = 0; G < 8; ++G) {
            LkH8 b;
#pragma unroll
            for (int q = 0; q < 2; ++q) b.p[q] = s_x[buf][(G * 2 + q) * 64 + lane];
            acc = lk_mma3h(G < NPF ? wn[G] : lk_fragh_load(fragb, 4, G0 + G, w, lane), b, acc);
        }
    };
    auto embed = [&](f32x16& acc, const u32x4* fragb, bool fetched) {
#pragma unroll
        for (int G = 0; G < 3; ++G) acc = lk_mma3h((DEEP && fetched) ? we[G] : lk_fragh_load(fragb, 4, G, w, lane), eb[G], acc);
    };
    // bias + softplus + fc_c(c) for the wave's own 32-unit block, then ALL stores of the layer: saved a / h rows, LDS park
    auto finish = [&](f32x16& acc, unsigned* save_s, int L, int buf) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = lk_softplus100(acc[r]);          // acc started from the layer's bias
        u32x4 sg0 = {0u, 0u, 0u, 0u}, sg1 = {0u, 0u, 0u, 0u};
        f32x16 act;
        if (save_s && a32) act = acc;
        else if (save_s) {        // softplus'(z) = sigmoid(100 z) = 1 - exp(-100 a): all the backward wants of a (lk_common.h: lk_pack_unorm16)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sg0[q] = lk_pack_unorm16(lk_softplus100_grad_from_out(acc[2 * q]), lk_softplus100_grad_from_out(acc[2 * q + 1]));
                sg1[q] = lk_pack_unorm16(lk_softplus100_grad_from_out(acc[8 + 2 * q]), lk_softplus100_grad_from_out(acc[8 + 2 * q + 1]));
            }
        }
        {
            const f32x16 ub = ct_bias_lds(s_bias[5 + L], w * 32, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += ub[r];
        }
#pragma unroll
        for (int G = 0; G < 2; ++G) acc = lk_mma3h(un[G], cb[G], acc);
        __builtin_amdgcn_sched_barrier(0);
        if (save_s && a32) ct_store_rows32(act_col_a + LK_COL_LAYER(a.P, L), act, live, lane);
        else if (save_s && live) { *reinterpret_cast<u32x4*>(save_s) = sg0; *reinterpret_cast<u32x4*>(save_s + 4) = sg1; }
        if (save) ct_store_rows32(act_col_h + LK_COL_LAYER(a.P, L) + w * 32, acc, live, lane);
        if (buf >= 0) {
            ct_check_range<CHECK>(acc, a.status);
#pragma unroll
            for (int G = 0; G < 2; ++G) {
                const LkH8 b = lk_split_cth(acc, G);
#pragma unroll
                for (int q = 0; q < 2; ++q) s_x[buf][((w * 2 + G) * 2 + q) * 64 + lane] = b.p[q];
            }
        }
    };
    f32x16 acc;
    // layer 0: 40 -> 128
    prefetch_u(FB + FM15_FWDH);
    acc = ct_bias_lds(s_bias[0], w * 32, lane);
    embed(acc, FB + FM10_FWDH, false);
    prefetch_hidden(FB + FM11_FWDH, 0);
    __builtin_amdgcn_sched_barrier(0);
    finish(acc, act_col_s, 0, 0);
    wg_barrier();
    LK_STAMP(6);
    // layers 1, 2: 128 -> 128
#pragma unroll
    for (int L = 1; L <= 2; ++L) {
        prefetch_u(FB + (L == 1 ? FM16_FWDH : FM17_FWDH));
        acc = ct_bias_lds(s_bias[L], w * 32, lane);
        hidden(acc, FB + (L == 1 ? FM11_FWDH : FM12_FWDH), 0, (L - 1) & 1);
        if (L == 1) prefetch_hidden(FB + FM12_FWDH, 0);
        else prefetch_hidden(FB + FM13_FWDH, 3);
        __builtin_amdgcn_sched_barrier(0);
        finish(acc, act_col_s ? act_col_s + LK_COL_SLAYER(a.P, L) : nullptr, L, L & 1);
        wg_barrier();
        if (L == 1) LK_STAMP(7); else LK_STAMP(8);
    }
    // layer 3 (skip): [e(40) | h(128)] -> 128
    prefetch_u(FB + FM18_FWDH);
    acc = ct_bias_lds(s_bias[3], w * 32, lane);
    embed(acc, FB + FM13_FWDH, true);
    hidden(acc, FB + FM13_FWDH, 3, 0);
    prefetch_hidden(FB + FM14_FWDH, 0);
    __builtin_amdgcn_sched_barrier(0);
    finish(acc, act_col_s ? act_col_s + LK_COL_SLAYER(a.P, 3) : nullptr, 3, 1);
    wg_barrier();
    LK_STAMP(9);
    // layer 4
    prefetch_u(FB + FM19_FWDH);
    acc = ct_bias_lds(s_bias[4], w * 32, lane);
    hidden(acc, FB + FM14_FWDH, 0, 1);
    finish(acc, act_col_s ? act_col_s + LK_COL_SLAYER(a.P, 4) : nullptr, 4, -1);
    LK_STAMP(10);
    // output 128 -> 3 on the VALU: per-wave partial over its 32 units, summed over the waves in fixed order
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int u = 32 * w + 8 * g + 4 * h;
        const float4 w0 = *reinterpret_cast<const float4*>(W + C_WO + u);
        const float4 w1 = *reinterpret_cast<const float4*>(W + C_WO + HC + u);
        const float4 w2 = *reinterpret_cast<const float4*>(W + C_WO + 2 * HC + u);
        const float v0 = acc[4 * g], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
        o0 = fmaf(w0.x, v0, o0); o0 = fmaf(w0.y, v1, o0); o0 = fmaf(w0.z, v2, o0); o0 = fmaf(w0.w, v3, o0);
        o1 = fmaf(w1.x, v0, o1); o1 = fmaf(w1.y, v1, o1); o1 = fmaf(w1.z, v2, o1); o1 = fmaf(w1.w, v3, o1);
        o2 = fmaf(w2.x, v0, o2); o2 = fmaf(w2.y, v1, o2); o2 = fmaf(w2.z, v2, o2); o2 = fmaf(w2.w, v3, o2);
    }
    o0 += __shfl_xor(o0, 32); o1 += __shfl_xor(o1, 32); o2 += __shfl_xor(o2, 32);
    if (h == 0) { s_o[w][lane] = o0; s_o[w][32 + lane] = o1; s_o[w][64 + lane] = o2; }
    wg_barrier();
    LK_STAMP(11);
    if (w == 0 && h == 0) {
        o0 = ((s_o[0][lane] + s_o[1][lane]) + s_o[2][lane]) + s_o[3][lane];
        o1 = ((s_o[0][32 + lane] + s_o[1][32 + lane]) + s_o[2][32 + lane]) + s_o[3][32 + lane];
        o2 = ((s_o[0][64 + lane] + s_o[1][64 + lane]) + s_o[2][64 + lane]) + s_o[3][64 + lane];
        o0 += W[C_BO]; o1 += W[C_BO + 1]; o2 += W[C_BO + 2];
        if (a.affine) {                      // out @ A + t, A = affine[:9].reshape(3,3) (decoder.py:536-539)
            const float* A = a.affine;
            const float t0 = o0 * A[0] + o1 * A[3] + o2 * A[6] + A[9];
            const float t1 = o0 * A[1] + o1 * A[4] + o2 * A[7] + A[10];
            const float t2 = o0 * A[2] + o1 * A[5] + o2 * A[8] + A[11];
            o0 = t0; o1 = t1; o2 = t2;
        }
        if (!(a.flags & LK_FLAG_COLOR_LOGITS)) { o0 = lk_sigmoid(o0); o1 = lk_sigmoid(o1); o2 = lk_sigmoid(o2); }
        if (live) {
            float* out = a.raw + (size_t)d.sample * 4;
            out[0] = o0; out[1] = o1; out[2] = o2;
        }
        if (s_raw) { s_raw[4 * lane] = o0; s_raw[4 * lane + 1] = o1; s_raw[4 * lane + 2] = o2; }
    }
}

// Block roles: `n_col_blocks` workgroups are colour tiles (4 waves per tile), the others run the geometry
// decoder (4 independent tiles per workgroup) and come FIRST in the grid.  raw[:, 0:3] and raw[:, 3] are written by the two roles separately;
// in the geometry stage there are no colour blocks and raw[:, 0:3] is zero-filled by the geometry wave.
template <bool DEEP, bool CHECK = false>
__global__ __launch_bounds__(256, 2) void k_decode_fwd(LkDecodeArgs a, int n_col_blocks) {
    __shared__ u32x4 s_x[2][16 * 64];
    __shared__ float s_o[4][3 * 32];
    __shared__ float s_bias[10][128];
    const int lane = lk_lane();
    const int w = (int)threadIdx.x >> 6;
    const int P_live = a.live_rays ? min(a.P, *a.live_rays * a.S) : a.P;       // samples of rays with a depth reading (they come first)
    // geometry workgroups first in the grid (one-wave chains, the longest of the launch: behind the colour tiles they were its tail)
    const int n_geo_blocks = (int)gridDim.x - n_col_blocks;
    const int bid = (int)blockIdx.x < n_geo_blocks ? n_col_blocks + (int)blockIdx.x : (int)blockIdx.x - n_geo_blocks;
    if (bid < n_col_blocks) {
        if (bid * 32 >= P_live) return;
        decode_col_wg<DEEP, false, CHECK>(a, bid, w, lane, s_x, s_o, s_bias);
        return;
    }
    const int tile = (bid - n_col_blocks) * 4 + w;
    if (tile * 32 >= P_live) return;
    (void)decode_geo_wave<CHECK>(a, tile, lane);
    if (n_col_blocks == 0) {
        const int sample = tile * 32 + (lane & 31);
        if (sample < a.P && lane < 32) { float* out = a.raw + (size_t)sample * 4; out[0] = 0.0f; out[1] = 0.0f; out[2] = 0.0f; }
    }
}

// ---------------------------------------------------------------------------------------------
// Relative-position neighbour MLP: one wave = 32 neighbour rows = 4 samples x 8 neighbours.
//   x_j = [sin(2 pi D_j B_r), cos(2 pi D_j B_r), F[I_j]] (52) -> 128 softplus100 -> 32;  c = sum_j w_j f_j
template <bool CHECK = false>
__device__ __forceinline__ void relpos_fwd_wave(const LkRelposArgs& a, int sample0, int P) {
    const int lane = lk_lane();
    const int h = lane >> 5;
    const int j = lane & 31;                    // row of the tile: sample j>>3, neighbour j&7
    const int sample = sample0 + (j >> 3);
    const bool live = sample < P;
    const int sp = live ? sample : P - 1;
    const int nb_i = j & 7;
    const int r = sp / a.S;
    const float z = a.z[sp];
    const float px = lk_madd_rn(a.rays_o[3 * r], a.rays_d[3 * r], z);
    const float py = lk_madd_rn(a.rays_o[3 * r + 1], a.rays_d[3 * r + 1], z);
    const float pz = lk_madd_rn(a.rays_o[3 * r + 2], a.rays_d[3 * r + 2], z);
    int idx = a.nbr_idx[(size_t)sp * LK_K + nb_i];
    const float wgt = (idx >= 0) ? a.nbr_w[(size_t)sp * LK_K + nb_i] : 0.0f;
    if (idx < 0) idx = 0;                       // padded slot: finite dummy row, weight 0
    // D = x_I - p (decoder.py:478-479); arguments a_i = fl(2 pi D_i)
    const float a0 = __fmul_rn(LK_TWO_PI, __fsub_rn(a.pos[3 * (size_t)idx], px));
    const float a1 = __fmul_rn(LK_TWO_PI, __fsub_rn(a.pos[3 * (size_t)idx + 1], py));
    const float a2 = __fmul_rn(LK_TWO_PI, __fsub_rn(a.pos[3 * (size_t)idx + 2], pz));
    const float* __restrict__ W = a.W;
    const u32x4* __restrict__ FB = reinterpret_cast<const u32x4*>(a.Wfrag) + FRAGB_U4;      // fp16 forward fragments
    const size_t frow = (size_t)idx * LK_C;
    const bool f16 = a.feats_f16 != 0;
    // X^T tiles: units 0..19 embedding, 20..51 feature channels 0..31, 52..55 zero
    f32x16 x0, x1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int u0 = 8 * g + 4 * h;
        if (u0 < ER) {
#pragma unroll
            for (int t = 0; t < 4; ++t) x0[4 * g + t] = sincos_embed_unit(W + R_EB, 10, u0 + t, a0, a1, a2);
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
    ct_check_range<CHECK>(x0, a.status); ct_check_range<CHECK>(x1, a.status);
    LK_STAMPW(1);                                    // (probe build) lists, positions and feature rows arrived, embedding evaluated
    f32x16 hid[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) hid[nb] = lk_rowvec_tile(W + R_B1, nb * 32, lane);      // accumulators start from the bias
    lk_gemm_h3<4, 2>(hid, FB + FM20_FWDH, 4, 0, 0, x0, 0, lane);
    lk_gemm_h3<4, 2>(hid, FB + FM20_FWDH, 4, 2, 0, x1, 0, lane);       // units 32..55; registers 12..15 of x1 are zero
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) hid[nb][q] = lk_softplus100(hid[nb][q]);
        ct_check_range<CHECK>(hid[nb], a.status);
    }
    LK_STAMP(2);
    f32x16 out[1];
    out[0] = lk_rowvec_tile(W + R_B2, 0, lane);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) lk_gemm_h3<1, 2>(out, FB + FM21_FWDH, 1, 2 * kb, 0, hid[kb], 0, lane);
    // c[ch] = sum over the 8 neighbour rows of a sample (8 consecutive lanes) of w * f[ch]
    const bool has = a.nbr_count[sp] >= a.min_nn;
    float* __restrict__ crow = a.c_col + (size_t)sp * LK_C;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float s = wgt * out[0][4 * g + t];
            s = lk_sum8<true>(s);
            v[t] = s;
        }
        if (live && nb_i == 0) {
            const int ch = 8 * g + 4 * h;
            float4 o = make_float4(v[0], v[1], v[2], v[3]);
            if (!has) o = a.noise_col ? *reinterpret_cast<const float4*>(a.noise_col + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(crow + ch) = o;
        }
    }
}
template <bool CHECK>
__device__ __forceinline__ void relpos_fwd_kernel(const LkRelposArgs& a) {
    const int wave = blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    const int sample0 = wave * 4;
    const int P = a.live_rays ? min(a.P, *a.live_rays * a.S) : a.P;      // rays without a reading sit behind the live prefix: skipped
    if (sample0 >= P) return;
    relpos_fwd_wave<CHECK>(a, sample0, P);
}
__global__ __launch_bounds__(256) void k_relpos_fwd(LkRelposArgs a) { relpos_fwd_kernel<false>(a); }
__global__ __launch_bounds__(256) void k_relpos_fwd_checked(LkRelposArgs a) { relpos_fwd_kernel<true>(a); }       // LK_FLAG_CHECK_RANGE

// Rel-pos neighbour MLP + both decoders in ONE launch (the tracker's batches: every kernel of an iteration is a single partial round
// of the chip, so a launch boundary costs its fixed ~6 us - dispatch, an L2-cold start, the drain - for nothing).  A colour workgroup is
// EIGHT waves: all eight run the rel-pos MLP of the tile's 32 samples (4 samples = 32 neighbour rows per wave, as k_relpos_fwd), meet at
// one barrier, then waves 0..3 decode the tile's colour as decode_col_wg does, wave 4 runs the geometry decoder of the SAME tile (no
// barriers in it) and waves 5..7 leave - a barrier only counts the waves still alive.  One workgroup per tile and nothing else in the
// grid: an eight-wave workgroup at this register count fills a compute unit, and a tracking batch (235 tiles) must not need a second round.
// The four colour waves use a barrier of their own (decode_col_wg<.., SOFTBAR>).  With s_barrier they stood at their FIRST barrier until wave 4 - the geometry
// decoder, no barrier in it - had ended: the hardware barrier waits for every wave of the workgroup that is still alive (shader-clock
// stamps, profiles/r4_decode_clock32.txt: 23 k cycles in the decode's set-up phase of the tracker's launch against 7.7 k for the same
// code in the mapper's, where the geometry tiles are workgroups of their own).
// COMP (the tracking loop): pass 1 of the tracker's loss is the launch's epilogue.  Tiles hold WHOLE rays (a.tile_stride = (32 / S) S
// samples, 30 of the 32 lanes at S = 5: 250 tiles instead of 235 for 1 500 rays, still one per compute unit); wave 4 leaves its occupancies and
// wave 0 its colours in LDS as raw rows, wave 0 waits for wave 4 (a counter in LDS, as the soft barrier) and its first lanes composite one ray
// each (lk_composite_ray on the LDS rows: the arithmetic of k_track_composite), write the ray's outputs and residual, and the tile's
// (sum of residuals, #present rays) pair - the mask threshold's partial sums are per tile instead of per 256 rays (5 us of an iteration of 117)
// GLDS (launches of at most one tile per compute unit - the 104 KB of LDS keep a second workgroup off the unit): the geometry decoder's operands
// staged in LDS, decode_geo_wave_lds
template <bool DEEP, bool COMP, bool GLDS>
__global__ __launch_bounds__(512) void k_relpos_decode_fwd(LkRelposArgs ra, LkDecodeArgs a, LkTrackLossArgs tl) {
    __shared__ u32x4 s_x[2][16 * 64];
    __shared__ float s_o[4][3 * 32];
    __shared__ float s_bias[10][128];
    __shared__ unsigned s_cnt;
    __shared__ unsigned s_geo_done;
    __shared__ __attribute__((aligned(16))) float s_raw[32 * 4];
    __shared__ u32x4 s_gw[GLDS ? LK_GEO_STAGE_BLOCKS * 128 : 1];
    __shared__ __attribute__((aligned(16))) float s_gb[GLDS ? 10 : 1][32];
    const int lane = lk_lane();
    const int w = (int)threadIdx.x >> 6;
    const int tile = (int)blockIdx.x;
    const int ts = COMP ? a.tile_stride : 32;
    const int base = tile * ts;
    const int lim = COMP ? min(ra.P, base + ts) : ra.P;      // the rel-pos rows of the tile's own samples only
    const int sample0 = base + 4 * w;
    if (threadIdx.x == 0) { s_cnt = 0u; s_geo_done = 0u; }
    LK_STAMP(0);
    if (w < 4) decode_col_setup(a, tile, w, lane, s_x, s_bias);      // (in front of the rel-pos MLP: its fetches are cold, and independent of it)
    if (GLDS) lk_geo_stage(a, w, lane, s_gw, s_gb);                  // the geometry decoder's operands into LDS (decode_geo_wave_lds)
    if (sample0 < lim) relpos_fwd_wave(ra, sample0, lim);
    LK_STAMP(3);
    __syncthreads();                               // the tile's c_col rows are written
    LK_STAMP(4);
    if (w > 4) return;
    if (w == 4) {
        const float occ = GLDS ? decode_geo_wave_lds(a, tile, lane, s_gw, s_gb) : decode_geo_wave(a, tile, lane);
        if (COMP) {
            if (lane < 32) s_raw[4 * lane + 3] = occ;
            __builtin_amdgcn_wave_barrier();       // (the host emulation runs lanes as fibers: every lane's row before lane 0's signal)
            __threadfence_block();
            if (lane == 0) atomicAdd(&s_geo_done, 1u);
        }
        return;
    }
    // the composite's global operands (the ray's reading, its samples' depths and neighbour counts) are fetched here, five layers ahead
    // of their use: behind the decoder they were a round trip at the end of every tile's chain
    const int S = COMP ? tl.S : 1;
    const int n_rays = COMP ? (lim - base) / S : 0;       // P and the tile stride are multiples of S
    float cz[LK_S_MAX], gd = 0.0f;
    bool chas[LK_S_MAX];
#pragma unroll
    for (int s = 0; s < LK_S_MAX; ++s) { cz[s] = 0.0f; chas[s] = false; }
    if (COMP && w == 0 && lane < n_rays) {
        gd = tl.gt_depth[base / S + lane];
#pragma unroll
        for (int s = 0; s < LK_S_MAX; ++s) {
            if (s < S) { cz[s] = tl.z[base + lane * S + s]; chas[s] = tl.nbr_count[base + lane * S + s] >= tl.min_nn; }
        }
    }
    decode_col_wg<DEEP, true, false, true>(a, tile, w, lane, s_x, s_o, s_bias, &s_cnt, COMP ? s_raw : nullptr);
    if (COMP && w == 0) {
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        while (*reinterpret_cast<volatile unsigned*>(&s_geo_done) < 1u) __builtin_amdgcn_s_sleep(1);
        __threadfence_block();
        LK_STAMP(12);
        float tv = 0.0f, cv = 0.0f;
        if (lane < n_rays) {
            const int r = base / S + lane;
            float4 q[LK_S_MAX];
#pragma unroll
            for (int s = 0; s < LK_S_MAX; ++s) q[s] = (s < S) ? *reinterpret_cast<const float4*>(s_raw + (lane * S + s) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const LkRayOut o = lk_composite_vals(q, chas, cz, S, tl.coef, gd);
            tl.depth[r] = o.depth; tl.var[r] = o.var;
            tl.color[3 * r] = o.c0; tl.color[3 * r + 1] = o.c1; tl.color[3 * r + 2] = o.c2;
            tl.valid_ray[r] = o.valid ? 1 : 0;
            const bool present = gd > 0.0f;
            tv = present ? fabsf(gd - o.depth) / sqrtf(o.var + 1e-10f) : 0.0f;
            cv = present ? 1.0f : 0.0f;
            tl.resid[r] = tl.median ? (present ? fabsf(gd - o.depth) : -1.0f) : tv;
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { tv += __shfl_xor(tv, o); cv += __shfl_xor(cv, o); }       // n_rays <= 8 (S >= 4)
        if (lane == 0) { tl.part[2 * tile] = tv; tl.part[2 * tile + 1] = cv; }
        if (tile == 0 && lane < 4) tl.out4[lane] = 0.0f;        // the loss row pass 2 accumulates into
        LK_STAMPW(13);
    }
}
int lk_launch_decode_fwd(const LkDecodeArgs& a, hipStream_t st) {
    LkProfScope prof_(LKK_DECODE_FWD, st);
    const int tiles = lk_cdiv(a.P, 32);
    const int n_col = (a.flags & LK_FLAG_STAGE_COLOR) ? tiles : 0;
    if (a.status) hipLaunchKernelGGL((k_decode_fwd<false, true>), dim3(n_col + lk_cdiv(tiles, 4)), dim3(256), 0, st, a, n_col);      // LK_FLAG_CHECK_RANGE
    else if (n_col > 0 && n_col <= LK_DEEP_MAX_TILES_FWD) hipLaunchKernelGGL(k_decode_fwd<true>, dim3(n_col + lk_cdiv(tiles, 4)), dim3(256), 0, st, a, n_col);
    else hipLaunchKernelGGL(k_decode_fwd<false>, dim3(n_col + lk_cdiv(tiles, 4)), dim3(256), 0, st, a, n_col);
    return LK_OK;
}
int lk_occupancy_decode_fwd() {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_decode_fwd<false>, 256, 0);
    return n;
}
int lk_occupancy_relpos_fwd() {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_relpos_fwd, 256, 0);
    return n;
}
// tracker-sized colour batches with the rel-pos MLP (lk_track_frame): see k_relpos_decode_fwd
bool lk_relpos_decode_fusable(const LkDecodeArgs& a) {
    const int tiles = lk_cdiv(a.P, 32);
    // (not with LK_FLAG_CHECK_RANGE: the operand tests live in the two separate kernels' checked instantiations)
    return (a.flags & LK_FLAG_STAGE_COLOR) && (a.flags & LK_FLAG_REL_POS) && tiles > 0 && tiles <= LK_DEEP_MAX_TILES_FWD && a.live_rays == nullptr && a.status == nullptr;
}
// comp: pass 1 of the tracker's loss as the launch's epilogue where the tiles can hold whole rays (4 <= S <= 32, one round of tiles);
// *comp_tiles = 0 otherwise and the caller launches k_track_composite
int lk_launch_relpos_decode_fwd(const LkRelposArgs& ra, const LkDecodeArgs& a, hipStream_t st, const LkTrackLossArgs* comp, int* comp_tiles) {
    LkProfScope prof_(LKK_DECODE_FWD, st);
    const int tiles = lk_cdiv(a.P, 32);
    LkTrackLossArgs tl;
    memset(&tl, 0, sizeof(tl));
    const int ts = a.S > 0 ? (32 / a.S) * a.S : 0;
    if (comp && a.S >= 4 && a.S <= 32 && a.P % a.S == 0 && comp->S == a.S && lk_cdiv(a.P, ts) <= LK_DEEP_MAX_TILES_FWD) {
        LkDecodeArgs b = a;
        b.tile_stride = ts;
        const int ctiles = lk_cdiv(a.P, ts);
        if (ctiles <= LK_GLDS_MAX_TILES) hipLaunchKernelGGL((k_relpos_decode_fwd<true, true, true>), dim3(ctiles), dim3(512), 0, st, ra, b, *comp);
        else hipLaunchKernelGGL((k_relpos_decode_fwd<true, true, false>), dim3(ctiles), dim3(512), 0, st, ra, b, *comp);
        if (comp_tiles) *comp_tiles = ctiles;
        return LK_OK;
    }
    if (tiles <= LK_GLDS_MAX_TILES) hipLaunchKernelGGL((k_relpos_decode_fwd<true, false, true>), dim3(tiles), dim3(512), 0, st, ra, a, tl);
    else hipLaunchKernelGGL((k_relpos_decode_fwd<true, false, false>), dim3(tiles), dim3(512), 0, st, ra, a, tl);
    return LK_OK;
}
int lk_launch_relpos_fwd(const LkRelposArgs& a, hipStream_t st) {
    LkProfScope prof_(LKK_RELPOS_FWD, st);
    const int waves = lk_cdiv(a.P, 4);
    if (a.status) hipLaunchKernelGGL(k_relpos_fwd_checked, dim3(lk_cdiv(waves, 4)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_relpos_fwd, dim3(lk_cdiv(waves, 4)), dim3(256), 0, st, a);
    return LK_OK;
}
