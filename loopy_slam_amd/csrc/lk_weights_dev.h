// Device side of the fragment repack (lk_weights.hip): shared with the interpolation kernel of lk_map_frame's iterations, which
// carries the repack of the iteration before as a rider (lk_sample.hip: k_interp_repack).
#pragma once
#include "lk_common.h"

using namespace lkw;

struct FragTable { FragMat m[N_FRAG_MATS]; unsigned* status; };      // status: lk_status_dev() - the range check of the fp16 forms

// one lane-block element of the split-bf16 fragments: unit u = (matrix form, G, blk, lane), 8 weights -> 3 x 16 B
// (m_lo, m_hi: only the matrices [m_lo, m_hi) of the table - a repack split between two launches, LK_FRAG_* below)
__device__ __forceinline__ void repack_split_unit(const float* __restrict__ plain, u32x4* __restrict__ fragb, const FragTable& tb, int u, int m_lo, int m_hi) {
    const int blk_all = u >> 6, lane = u & 63;
    const int o4 = blk_all * 192;                      // uint4 offset of the block
    int mi = 0;
#pragma unroll 1
    for (int q = 1; q < N_FRAG_MATS; ++q) mi = (o4 >= tb.m[q].fwdb) ? q : mi;
    if (mi < m_lo || mi >= m_hi) return;
    const FragMat M = tb.m[mi];
    const int h = lane >> 5, j = lane & 31;
    float v[8];
    if (o4 < M.trb) {                                  // forward: [G][nb], embedding run padded to 16 columns
        const int blk = (o4 - M.fwdb) / 192;
        const int NBT = M.rows >> 5;
        const int G = blk / NBT, nb = blk - G * NBT;
        const int ksplit = M.e_real < M.ld ? M.e_real : M.ld;
        const int GE = kb16(ksplit);
        const int kbase = G < GE ? 16 * G : ksplit + 16 * (G - GE);
        const int kend = G < GE ? ksplit : M.ld;
        const int row = nb * 32 + j;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int col = kbase + (i < 4 ? 4 * h + i : 8 + 4 * h + (i - 4));
            v[i] = col < kend ? plain[M.plain + row * M.ld + col] : 0.0f;
        }
    } else {                                           // transposed: [G over the layer's outputs][kb over virtual inputs]
        const int blk = (o4 - M.trb) / 192;
        const int KB = M.kv >> 5;
        const int G = blk / KB, kb = blk - G * KB;
        const int vc = kb * 32 + j;
        int col = -1;
        if (vc < M.e_real) col = vc;
        else if (vc >= M.e_virt) col = vc - (M.e_virt - M.e_real);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = 16 * G + (i < 4 ? 4 * h + i : 8 + 4 * h + (i - 4));
            v[i] = (col >= 0 && col < M.ld) ? plain[M.plain + n * M.ld + col] : 0.0f;
        }
    }
    const LkB8 s = lk_split8(v);
    u32x4* __restrict__ q = fragb + o4 + lane;
    q[0] = s.p[0]; q[64] = s.p[1]; q[128] = s.p[2];
}

// both forms once more as two fp16 pieces: unit u = (matrix form, G, blk, lane)
__device__ __forceinline__ void repack_half_unit(const float* __restrict__ plain, u32x4* __restrict__ fragh, const FragTable& tb, int u, int m_lo, int m_hi) {
    const int blk_all = u >> 6, lane = u & 63;
    const int o4 = blk_all * 128;
    int mi = 0;
#pragma unroll 1
    for (int q = 1; q < N_FRAG_MATS; ++q) mi = (o4 >= tb.m[q].fwdh) ? q : mi;
    if (mi < m_lo || mi >= m_hi) return;
    const FragMat M = tb.m[mi];
    const int h = lane >> 5, j = lane & 31;
    float v[8];
    if (o4 < M.trh) {
        const int blk = (o4 - M.fwdh) / 128;
        const int NBT = M.rows >> 5;
        const int G = blk / NBT, nb = blk - G * NBT;
        const int ksplit = M.e_real < M.ld ? M.e_real : M.ld;
        const int GE = kb16(ksplit);
        const int kbase = G < GE ? 16 * G : ksplit + 16 * (G - GE);
        const int kend = G < GE ? ksplit : M.ld;
        const int row = nb * 32 + j;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int col = kbase + (i < 4 ? 4 * h + i : 8 + 4 * h + (i - 4));
            v[i] = col < kend ? plain[M.plain + row * M.ld + col] : 0.0f;
        }
    } else {
        const int blk = (o4 - M.trh) / 128;
        const int KB = M.kv >> 5;
        const int G = blk / KB, kb = blk - G * KB;
        const int vc = kb * 32 + j;
        int col = -1;
        if (vc < M.e_real) col = vc;
        else if (vc >= M.e_virt) col = vc - (M.e_virt - M.e_real);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = 16 * G + (i < 4 ? 4 * h + i : 8 + 4 * h + (i - 4));
            v[i] = (col >= 0 && col < M.ld) ? plain[M.plain + n * M.ld + col] : 0.0f;
        }
    }
    // fp16 pieces saturate where bf16 pieces do not: a weight at or above 2^15 (or a non-finite one) is reported, not silently clipped
    // (LK_STATUS_WEIGHT_RANGE; the reference's fp32 decoder has no ceiling, decoder.py:513-546)
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) bad = bad || lk_out_of_range(v[i], 32768.0f);
    if (bad) lk_status_raise(tb.status, LK_STATUS_WEIGHT_RANGE);
    const LkH8 s = lk_split8h(v);
    u32x4* __restrict__ q = fragh + o4 + lane;
    q[0] = s.p[0]; q[64] = s.p[1];
}

// unit u of the whole repack: split-bf16 forms first, then the fp16 forms
#define LK_REPACK_UNITS (FRAGB_U4 / 3 + FRAGH_U4 / 2)
__device__ __forceinline__ void repack_unit(const float* __restrict__ plain, u32x4* __restrict__ fragb, const FragTable& tb, int u, int m_lo = 0, int m_hi = N_FRAG_MATS) {
    if (u < FRAGB_U4 / 3) repack_split_unit(plain, fragb, tb, u, m_lo, m_hi);
    else repack_half_unit(plain, fragb + FRAGB_U4, tb, u - FRAGB_U4 / 3, m_lo, m_hi);
}
inline FragTable lk_frag_table() {
    static const FragMat rows[N_FRAG_MATS] = {LKW_FRAG_TABLE};
    FragTable tb;
    for (int i = 0; i < N_FRAG_MATS; ++i) tb.m[i] = rows[i];
    tb.status = lk_status_dev();
    return tb;
}
