// Weight gradients of the GEOMETRY decoder (mapping.fix_geo_decoder: False - Mapper.py:524-526 puts geo_decoder.parameters() into
// the decoder group; every reference config keeps it True, where only embedder._B is trained and this file is never launched).
//   reference: MLP_geometry.forward (src/conv_onet/models/decoder.py:263-288):
//     h_0 = relu(W_0 e + b_0) + U_0 c + u_0;  h_i = relu(W_i x_i + b_i) + U_i c + u_i,  x_3 = [e ; h_2], x_i = h_{i-1} otherwise;
//     occ = w_o . h_4 + b_o;  e = sin(2 pi p B_g)
//
// An option that is off in every shipped config does not get to touch k_decode_bwd (its registers are budgeted to the last one,
// DESIGN.md section 3): this is a stand-alone launch that takes what the forward saved - the relu outputs a_i (act), the interpolated
// feature c (c_geo), d loss / d occ (d_raw[.,3]) - and redoes the 32-wide chain in plain fp32:
//     d h_4 = w_o d occ;   d y_i = d h_i (a_i > 0);   d h_{i-1} = W_i[:, hidden]^T d y_i
//     d W_i += d y_i x_i^T,  d b_i += d y_i,  d U_i += d h_i c^T,  d u_i += d h_i,  d w_o += d occ h_4,  d b_o += d occ
// HBM-bound by construction (<= 1.3 KB read per sample, 16 128 gradient floats per workgroup written once): a workgroup walks
// 16-sample tiles, stages each sample's vectors in LDS (one record per sample), and every thread keeps 63 gradient entries of the
// geometry decoder's span of the blob [G_W0, G_END) in registers - entry e of the span is a product of two record slots, decoded once.
// Partial spans per workgroup, summed by lk_launch_reduce_partials into g_weights (which accumulates, like every other gradient).
#include "lk_common.h"
#include "lk_kernels.h"

using namespace lkw;

namespace {
constexpr int GW_TILE = 16;                       // samples per tile (16 threads stage one sample)
constexpr int GW_SPAN = G_END - G_W0;             // floats of the blob that belong to the geometry decoder's matrices and biases
constexpr int GW_EPT = (GW_SPAN + 255) / 256;     // gradient entries per thread
// record of one sample in LDS
constexpr int RO_E = 0;                           // e          [96]  (93 real)
constexpr int RO_A = RO_E + EGP;                  // a_0..a_4   [160] relu outputs
constexpr int RO_H = RO_A + 5 * HG;               // h_0..h_4   [160]
constexpr int RO_C = RO_H + 5 * HG;               // c          [32]
constexpr int RO_DY = RO_C + CF;                  // d y_0..4   [160]
constexpr int RO_DH = RO_DY + 5 * HG;             // d h_0..4   [160]
constexpr int RO_DOCC = RO_DH + 5 * HG;           // d occ
constexpr int RO_ONE = RO_DOCC + 1;               // 1.0 (bias entries)
constexpr int RO_ZERO = RO_ONE + 1;               // 0.0 (padding entries of the blob)
constexpr int RO_STRIDE = RO_ZERO + 2;            // 772: even, and 772 mod 32 = 4 spreads the 16 records over the banks

// blob offset b in [G_W0, G_END) -> the two record slots whose product is that entry's per-sample gradient
__device__ __forceinline__ void gw_entry(int b, int& ia, int& ib) {
    ia = RO_ZERO; ib = RO_ZERO;
    auto in = [&](int base, int n) { return b >= base && b < base + n; };
    if (in(G_W0, HG * EGP)) { const int r = b - G_W0; ia = RO_DY + r / EGP; ib = RO_E + r % EGP; return; }
    if (in(G_W1, HG * HG)) { const int r = b - G_W1; ia = RO_DY + HG + r / HG; ib = RO_H + r % HG; return; }
    if (in(G_W2, HG * HG)) { const int r = b - G_W2; ia = RO_DY + 2 * HG + r / HG; ib = RO_H + HG + r % HG; return; }
    if (in(G_W3, HG * (EGP + HG))) {
        const int r = b - G_W3, o = r / (EGP + HG), k = r % (EGP + HG);
        ia = RO_DY + 3 * HG + o; ib = k < EGP ? RO_E + k : RO_H + 2 * HG + (k - EGP); return;
    }
    if (in(G_W4, HG * HG)) { const int r = b - G_W4; ia = RO_DY + 4 * HG + r / HG; ib = RO_H + 3 * HG + r % HG; return; }
    const int b_off[5] = {G_B0, G_B1, G_B2, G_B3, G_B4};
#pragma unroll
    for (int i = 0; i < 5; ++i)
        if (b >= b_off[i] && b < b_off[i] + HG) { ia = RO_DY + i * HG + (b - b_off[i]); ib = RO_ONE; return; }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int u0 = G_U0 + i * G_USTRIDE, ub = u0 + a64(HG * CF);
        if (b >= u0 && b < u0 + HG * CF) { const int r = b - u0; ia = RO_DH + i * HG + r / CF; ib = RO_C + r % CF; return; }
        if (b >= ub && b < ub + HG) { ia = RO_DH + i * HG + (b - ub); ib = RO_ONE; return; }
    }
    if (b >= G_WO && b < G_WO + HG) { ia = RO_DOCC; ib = RO_H + 4 * HG + (b - G_WO); return; }
    if (b == G_BO) { ia = RO_DOCC; ib = RO_ONE; return; }
}
}  // namespace

// weights are re-read per tile (L1 / L2 hits): a plain load is loop-invariant, and the compiler then hoists all ~600 of them out of the
// tile loop into registers (512 VGPRs + 361 spilled)
__device__ __forceinline__ float gw_ld(const float* p) { return *reinterpret_cast<const volatile float*>(p); }

struct LkGeoWgradArgs {
    int P, S;
    const float* rays_o; const float* rays_d; const float* z;
    const float* W;            // master blob
    const float* act;          // [P][LK_ACT_GEO_A] relu outputs of the five trunk layers (sample-major head of the activation buffer)
    const float* c_geo;        // [P][32]
    const float* d_raw;        // [P][4], .w = d loss / d occ
    float* part;               // [gridDim.x][GW_SPAN]
    const int32_t* live_rays;  // [1] or NULL: only the samples of the first *live_rays rays were rendered (partitioned batches of lk_map_frame)
};

__global__ __launch_bounds__(256) void k_geo_wgrad(LkGeoWgradArgs a) {
    __shared__ float rec[GW_TILE * RO_STRIDE];
    const int t = (int)threadIdx.x, s = t >> 4, q = t & 15;
    const float* __restrict__ W = a.W;
    float acc[GW_EPT];
    int slot[GW_EPT];
#pragma unroll
    for (int j = 0; j < GW_EPT; ++j) {
        acc[j] = 0.0f;
        int ia = RO_ZERO, ib = RO_ZERO;
        const int e = t + 256 * j;
        if (e < GW_SPAN) gw_entry(G_W0 + e, ia, ib);
        slot[j] = ia | (ib << 16);
    }
    float* __restrict__ R = rec + s * RO_STRIDE;
    const int P_live = a.live_rays ? min(a.P, *a.live_rays * a.S) : a.P;      // the forward saved nothing for the samples behind the live prefix
    const int n_tiles = (P_live + GW_TILE - 1) / GW_TILE;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int sp = tile * GW_TILE + s;
        const bool live = sp < P_live;
        __syncthreads();                                                   // the accumulate phase of the tile before
        // ---- stage: d occ, constants, embedding, relu outputs, c
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
        if (live) {
            const int r = sp / a.S;
            const float z = a.z[sp];
            a0 = __fmul_rn(LK_TWO_PI, lk_madd_rn(a.rays_o[3 * r], a.rays_d[3 * r], z));
            a1 = __fmul_rn(LK_TWO_PI, lk_madd_rn(a.rays_o[3 * r + 1], a.rays_d[3 * r + 1], z));
            a2 = __fmul_rn(LK_TWO_PI, lk_madd_rn(a.rays_o[3 * r + 2], a.rays_d[3 * r + 2], z));
        }
        if (q == 0) { R[RO_DOCC] = live ? a.d_raw[(size_t)sp * 4 + 3] : 0.0f; R[RO_ONE] = 1.0f; R[RO_ZERO] = 0.0f; R[RO_ZERO + 1] = 0.0f; }
#pragma unroll
        for (int m = 0; m < EGP / 16; ++m) {
            const int u = q + 16 * m;
            R[RO_E + u] = (live && u < EG) ? lk_sinf(lk_fourier_arg(a0, a1, a2, gw_ld(W + G_EB + u), gw_ld(W + G_EB + EGP + u), gw_ld(W + G_EB + 2 * EGP + u))) : 0.0f;
        }
#pragma unroll
        for (int m = 0; m < 5 * HG / 16; ++m) R[RO_A + q + 16 * m] = live ? a.act[(size_t)sp * LK_ACT_GEO_A + q + 16 * m] : 0.0f;
#pragma unroll
        for (int m = 0; m < CF / 16; ++m) R[RO_C + q + 16 * m] = live ? a.c_geo[(size_t)sp * LK_C + q + 16 * m] : 0.0f;
        __syncthreads();
        // ---- h_i = a_i + U_i c + u_i
#pragma unroll 1
        for (int m = 0; m < 5 * HG / 16; ++m) {
            const int idx = q + 16 * m, i = idx / HG, o = idx % HG;
            const float* U = W + G_U0 + i * G_USTRIDE;
            float v = gw_ld(U + a64(HG * CF) + o);
#pragma unroll 4
            for (int k = 0; k < CF; ++k) v += gw_ld(U + o * CF + k) * R[RO_C + k];
            R[RO_H + idx] = R[RO_A + idx] + v;
        }
        // ---- the backward chain, two units per thread and layer
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int o = q + 16 * m;
            const float dh = gw_ld(W + G_WO + o) * R[RO_DOCC];
            R[RO_DH + 4 * HG + o] = dh;
            R[RO_DY + 4 * HG + o] = (R[RO_A + 4 * HG + o] > 0.0f) ? dh : 0.0f;
        }
#pragma unroll 1
        for (int i = 4; i >= 1; --i) {
            __syncthreads();
            const int base = (i == 4) ? G_W4 : (i == 3) ? G_W3 : (i == 2) ? G_W2 : G_W1;
            const int ld = (i == 3) ? EGP + HG : HG, h0 = (i == 3) ? EGP : 0;          // hidden columns of W_i
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int k = q + 16 * m;
                float dh = 0.0f;
#pragma unroll 4
                for (int o = 0; o < HG; ++o) dh += gw_ld(W + base + o * ld + h0 + k) * R[RO_DY + i * HG + o];
                R[RO_DH + (i - 1) * HG + k] = dh;
                R[RO_DY + (i - 1) * HG + k] = (R[RO_A + (i - 1) * HG + k] > 0.0f) ? dh : 0.0f;
            }
        }
        __syncthreads();
        // ---- accumulate: entry = product of two record slots, summed over the tile's samples
        for (int ss = 0; ss < GW_TILE; ++ss) {
            const float* __restrict__ Rs = rec + ss * RO_STRIDE;
#pragma unroll
            for (int j = 0; j < GW_EPT; ++j) acc[j] += Rs[slot[j] & 0xffff] * Rs[slot[j] >> 16];
        }
    }
    float* __restrict__ out = a.part + (size_t)blockIdx.x * GW_SPAN;
#pragma unroll
    for (int j = 0; j < GW_EPT; ++j) {
        const int e = t + 256 * j;
        if (e < GW_SPAN) out[e] = acc[j];
    }
}

// up to three workgroups per compute unit (49 KB of LDS each): with 256 - one per unit, four waves - the launch measured 670 us per 25 000
// samples on the chip (profiles/r4_iteration_timeline_geofree.md), bound by the dependent LDS reads of its accumulate phase
int lk_geo_wgrad_parts(int P) {
    const int tiles = lk_cdiv(P, GW_TILE);
    return tiles < 768 ? (tiles < 1 ? 1 : tiles) : 768;
}
int64_t lk_geo_wgrad_part_floats(int P) { return (int64_t)lk_geo_wgrad_parts(P) * GW_SPAN; }

// d_raw must be complete (composite backward); g_weights accumulates
int lk_launch_geo_wgrad(int P, int S, const float* rays_o, const float* rays_d, const float* z, const float* W, const float* act,
                        const float* c_geo, const float* d_raw, float* part, float* g_weights, hipStream_t st, const int32_t* live_rays) {
    if (P <= 0) return LK_OK;
    LkGeoWgradArgs a;
    a.live_rays = live_rays;
    a.P = P; a.S = S; a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.W = W; a.act = act; a.c_geo = c_geo; a.d_raw = d_raw; a.part = part;
    const int n = lk_geo_wgrad_parts(P);
    hipLaunchKernelGGL(k_geo_wgrad, dim3(n), dim3(256), 0, st, a);
    return lk_launch_reduce_partials(part, n, GW_SPAN, g_weights + G_W0, st);
}
