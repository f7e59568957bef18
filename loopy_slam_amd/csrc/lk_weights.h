// Packed decoder-weight blob layout (see include/loopy_hip.h, lk_weight_layout).
// Offsets are compile-time constants shared by the kernels and the host table.
// Reference shapes: SURVEY.md Appendix C / src/conv_onet/models/decoder.py:125-170,364-420.
#pragma once
#include <stdint.h>

namespace lkw {

constexpr int a64(int x) { return (x + 63) / 64 * 64; }

constexpr int CF = 32;     // feature channels (c_dim)
constexpr int HG = 32;     // geometry hidden width
constexpr int EG = 93;     // geometry Fourier features (sin only)
constexpr int EGP = 96;    // padded
constexpr int HC = 128;    // colour hidden width
constexpr int EC = 40;     // colour Fourier features ([sin 20, cos 20])
constexpr int ER = 20;     // rel-pos Fourier features ([sin 10, cos 10])
constexpr int KR = 52;     // rel-pos MLP input (20 + 32)
constexpr int KRP = 56;    // padded

// ---- geometry decoder
constexpr int G_EB = 0;                              // [3][96]   embedder._B (cols 93..95 = 0)
constexpr int G_W0 = G_EB + a64(3 * EGP);            // [32][96]
constexpr int G_B0 = G_W0 + a64(HG * EGP);
constexpr int G_W1 = G_B0 + a64(HG);                 // [32][32]
constexpr int G_B1 = G_W1 + a64(HG * HG);
constexpr int G_W2 = G_B1 + a64(HG);
constexpr int G_B2 = G_W2 + a64(HG * HG);
constexpr int G_W3 = G_B2 + a64(HG);                 // [32][128] = [e(93) 0 0 0 | h(32)]
constexpr int G_B3 = G_W3 + a64(HG * (EGP + HG));
constexpr int G_W4 = G_B3 + a64(HG);
constexpr int G_B4 = G_W4 + a64(HG * HG);
constexpr int G_U0 = G_B4 + a64(HG);                 // fc_c.i [32][32], bias [32], i = 0..4, stride G_USTRIDE
constexpr int G_USTRIDE = a64(HG * CF) + a64(HG);
constexpr int G_WO = G_U0 + 5 * G_USTRIDE;           // [32]
constexpr int G_BO = G_WO + a64(HG);                 // [1]
constexpr int G_END = G_BO + 64;

// ---- colour decoder
constexpr int C_EB = G_END;                          // [3][20]   embedder._B (fixed, not a Parameter)
constexpr int C_W0 = C_EB + a64(3 * 20);             // [128][40]
constexpr int C_B0 = C_W0 + a64(HC * EC);
constexpr int C_W1 = C_B0 + a64(HC);                 // [128][128]
constexpr int C_B1 = C_W1 + a64(HC * HC);
constexpr int C_W2 = C_B1 + a64(HC);
constexpr int C_B2 = C_W2 + a64(HC * HC);
constexpr int C_W3 = C_B2 + a64(HC);                 // [128][168] = [e(40) | h(128)]
constexpr int C_B3 = C_W3 + a64(HC * (EC + HC));
constexpr int C_W4 = C_B3 + a64(HC);
constexpr int C_B4 = C_W4 + a64(HC * HC);
constexpr int C_U0 = C_B4 + a64(HC);                 // fc_c.i [128][32], bias [128], stride C_USTRIDE
constexpr int C_USTRIDE = a64(HC * CF) + a64(HC);
constexpr int C_WO = C_U0 + 5 * C_USTRIDE;           // [3][128]
constexpr int C_BO = C_WO + a64(3 * HC);             // [3]
// ---- colour decoder: relative-position neighbour MLP
constexpr int R_EB = C_BO + 64;                      // [3][10]  embedder_rel_pos._B
constexpr int R_W1 = R_EB + a64(3 * 10);             // [128][56] (cols 52..55 = 0)
constexpr int R_B1 = R_W1 + a64(HC * KRP);
constexpr int R_W2 = R_B1 + a64(HC);                 // [32][128]
constexpr int R_B2 = R_W2 + a64(32 * HC);
constexpr int BLOB_FLOATS = R_B2 + 64;


// ---------------------------------------------------------------------------------------------
// Derived "fragment" blob (lk_weights_repack): every GEMM matrix re-laid in the operand order of the matrix
// instruction, every weight cut into three bf16 pieces hi + mid + lo (exact; lk_common.h::lk_mma6), in two forms:
//   forward    block (G, nb): lane l, element i -> W[nb*32 + (l&31)][k(G, l>>5, i)]
//   transposed block (G, kb): lane l, element i -> W[k(G, l>>5, i)][vcol = kb*32 + (l&31)]
// with k(G, h, i) = 16 G + 4 h + i (i < 4), 16 G + 8 + 4 h + (i - 4) (i >= 4): the two C/D-row-walk groups (2G, 2G+1) of
// a CT tile, i.e. registers 8G..8G+7 of the previous layer's accumulators are the matching B operand.
// One block = the 8 k-values a lane feeds to ONE v_mfma_f32_32x32x16_bf16, for the 3 pieces:
//   block = [piece 0..2][lane 0..63] uint4  (3 KiB contiguous, three fully used 1-KiB wave loads);  offsets in uint4 units.
// "vcol" is a virtual input column: the embedding part of a skip/first layer is padded to a multiple of 32 (colour:
// 40 -> 64) so that every 32-wide block of dX^T is either embedding or hidden.  In the forward form a skip / first
// layer's embedding columns form their own run of blocks, zero-padded to a multiple of 16 columns, so that a block never
// straddles the embedding and the hidden part (their B operands live in different CT tiles).
struct FragMat { int plain, rows, ld, e_real, e_virt, kv, fwdb, trb, fwdh, trh; };

constexpr int kb16(int k) { return (k + 15) / 16; }
constexpr int fwd_blocks16(int ld, int e_real) { return e_real < ld ? kb16(e_real) + kb16(ld - e_real) : kb16(ld); }

#define LKW_FM(idx, plain_, rows_, ld_, ereal_, evirt_, kv_, prev_)                              \
    constexpr int FM##idx##_FWDB = FM##prev_##_ENDB;                                             \
    constexpr int FM##idx##_TRB = FM##idx##_FWDB + fwd_blocks16(ld_, ereal_) * ((rows_) / 32) * 192; \
    constexpr int FM##idx##_ENDB = FM##idx##_TRB + ((rows_) / 16) * ((kv_) / 32) * 192;        \
    constexpr int FM##idx##_FWDH = FM##prev_##_ENDH;                                             \
    constexpr int FM##idx##_TRH = FM##idx##_FWDH + fwd_blocks16(ld_, ereal_) * ((rows_) / 32) * 128; \
    constexpr int FM##idx##_ENDH = FM##idx##_TRH + ((rows_) / 16) * ((kv_) / 32) * 128;

constexpr int FMS_ENDB = 0, FMS_ENDH = 0;

// index:            plain   rows ld            e_real e_virt kv
LKW_FM(0,  G_W0, HG, EGP,        EGP, EGP, 96,  S)
LKW_FM(1,  G_W1, HG, HG,         HG,  HG,  32,  0)
LKW_FM(2,  G_W2, HG, HG,         HG,  HG,  32,  1)
LKW_FM(3,  G_W3, HG, EGP + HG,   128, 128, 128, 2)
LKW_FM(4,  G_W4, HG, HG,         HG,  HG,  32,  3)
LKW_FM(5,  G_U0 + 0 * G_USTRIDE, HG, CF, CF, CF, 32, 4)
LKW_FM(6,  G_U0 + 1 * G_USTRIDE, HG, CF, CF, CF, 32, 5)
LKW_FM(7,  G_U0 + 2 * G_USTRIDE, HG, CF, CF, CF, 32, 6)
LKW_FM(8,  G_U0 + 3 * G_USTRIDE, HG, CF, CF, CF, 32, 7)
LKW_FM(9,  G_U0 + 4 * G_USTRIDE, HG, CF, CF, CF, 32, 8)
LKW_FM(10, C_W0, HC, EC,         EC,  64,  64,  9)
LKW_FM(11, C_W1, HC, HC,         HC,  HC,  128, 10)
LKW_FM(12, C_W2, HC, HC,         HC,  HC,  128, 11)
LKW_FM(13, C_W3, HC, EC + HC,    EC,  64,  192, 12)
LKW_FM(14, C_W4, HC, HC,         HC,  HC,  128, 13)
LKW_FM(15, C_U0 + 0 * C_USTRIDE, HC, CF, CF, CF, 32, 14)
LKW_FM(16, C_U0 + 1 * C_USTRIDE, HC, CF, CF, CF, 32, 15)
LKW_FM(17, C_U0 + 2 * C_USTRIDE, HC, CF, CF, CF, 32, 16)
LKW_FM(18, C_U0 + 3 * C_USTRIDE, HC, CF, CF, CF, 32, 17)
LKW_FM(19, C_U0 + 4 * C_USTRIDE, HC, CF, CF, CF, 32, 18)
LKW_FM(20, R_W1, HC, KRP,        KRP, 64,  64,  19)
LKW_FM(21, R_W2, CF, HC,         HC,  HC,  128, 20)
constexpr int FRAGB_U4 = FM21_ENDB;         // uint4 units
// Both forms once more as TWO fp16 pieces (lk_common.h::lk_mma3h: the forward kernels, and the mapper's backward whose
// loss gradients have unit scale): block = [piece 0..1][lane] uint4 = 2 KiB, same block order; they follow the bf16 blob.
constexpr int FRAGH_U4 = FM21_ENDH;
constexpr int N_FRAG_MATS = 22;
// matrix indices of the table below: geometry decoder [0, 10), colour trunk [LK_FRAG_COL_LO, LK_FRAG_COL_HI), rel-pos MLP [20, 22);
// the trunk's parameters are the blob range [C_EB, R_EB)
#define LK_FRAG_COL_LO 10
#define LK_FRAG_COL_HI 20
// transposed-form offsets by matrix index (bf16 pieces / fp16 pieces), for code templated on the piece type
constexpr int FRAG_TRB[N_FRAG_MATS] = {FM0_TRB, FM1_TRB, FM2_TRB, FM3_TRB, FM4_TRB, FM5_TRB, FM6_TRB, FM7_TRB, FM8_TRB, FM9_TRB, FM10_TRB,
                                       FM11_TRB, FM12_TRB, FM13_TRB, FM14_TRB, FM15_TRB, FM16_TRB, FM17_TRB, FM18_TRB, FM19_TRB, FM20_TRB, FM21_TRB};
constexpr int FRAG_TRH[N_FRAG_MATS] = {FM0_TRH, FM1_TRH, FM2_TRH, FM3_TRH, FM4_TRH, FM5_TRH, FM6_TRH, FM7_TRH, FM8_TRH, FM9_TRH, FM10_TRH,
                                       FM11_TRH, FM12_TRH, FM13_TRH, FM14_TRH, FM15_TRH, FM16_TRH, FM17_TRH, FM18_TRH, FM19_TRH, FM20_TRH, FM21_TRH};

#define LKW_FM_ROW(idx, plain_, rows_, ld_, ereal_, evirt_, kv_) {plain_, rows_, ld_, ereal_, evirt_, kv_, FM##idx##_FWDB, FM##idx##_TRB, FM##idx##_FWDH, FM##idx##_TRH}
#define LKW_FRAG_TABLE                                                                 \
    LKW_FM_ROW(0,  G_W0, HG, EGP,        EGP, EGP, 96),                                \
    LKW_FM_ROW(1,  G_W1, HG, HG,         HG,  HG,  32),                                \
    LKW_FM_ROW(2,  G_W2, HG, HG,         HG,  HG,  32),                                \
    LKW_FM_ROW(3,  G_W3, HG, EGP + HG,   128, 128, 128),                               \
    LKW_FM_ROW(4,  G_W4, HG, HG,         HG,  HG,  32),                                \
    LKW_FM_ROW(5,  G_U0 + 0 * G_USTRIDE, HG, CF, CF, CF, 32),                          \
    LKW_FM_ROW(6,  G_U0 + 1 * G_USTRIDE, HG, CF, CF, CF, 32),                          \
    LKW_FM_ROW(7,  G_U0 + 2 * G_USTRIDE, HG, CF, CF, CF, 32),                          \
    LKW_FM_ROW(8,  G_U0 + 3 * G_USTRIDE, HG, CF, CF, CF, 32),                          \
    LKW_FM_ROW(9,  G_U0 + 4 * G_USTRIDE, HG, CF, CF, CF, 32),                          \
    LKW_FM_ROW(10, C_W0, HC, EC,         EC,  64,  64),                                \
    LKW_FM_ROW(11, C_W1, HC, HC,         HC,  HC,  128),                               \
    LKW_FM_ROW(12, C_W2, HC, HC,         HC,  HC,  128),                               \
    LKW_FM_ROW(13, C_W3, HC, EC + HC,    EC,  64,  192),                               \
    LKW_FM_ROW(14, C_W4, HC, HC,         HC,  HC,  128),                               \
    LKW_FM_ROW(15, C_U0 + 0 * C_USTRIDE, HC, CF, CF, CF, 32),                          \
    LKW_FM_ROW(16, C_U0 + 1 * C_USTRIDE, HC, CF, CF, CF, 32),                          \
    LKW_FM_ROW(17, C_U0 + 2 * C_USTRIDE, HC, CF, CF, CF, 32),                          \
    LKW_FM_ROW(18, C_U0 + 3 * C_USTRIDE, HC, CF, CF, CF, 32),                          \
    LKW_FM_ROW(19, C_U0 + 4 * C_USTRIDE, HC, CF, CF, CF, 32),                          \
    LKW_FM_ROW(20, R_W1, HC, KRP,        KRP, 64,  64),                                \
    LKW_FM_ROW(21, R_W2, CF, HC,         HC,  HC,  128)

}  // namespace lkw
