// Internal kernel argument blocks and launch prototypes (one translation unit per kernel family).
#pragma once
#include "lk_common.h"

struct LkSampleArgs {
    int R, S, P, stats_chunk;
    unsigned flags;
    const float* rays_o; const float* rays_d; const float* gt_depth; const float* r2_ray; const float* far_stats;
    const LkGrid* grid; const float4* sorted; const int32_t* cell_start;
    const float* geo_feats; const float* col_feats; const float* noise_geo; const float* noise_col;
    float near_surface, far_surface, near_end, r2_static;
    int min_nn;
    float* z; int32_t* nbr_idx; float* nbr_w; int32_t* nbr_count; float* c_geo; float* c_col;
};

struct LkCompositeArgs {
    int R, S, min_nn;
    float coef;
    const float* raw; const float* z; const int32_t* nbr_count; const float* gt_depth;
    float* depth; float* var; float* color; uint8_t* valid_ray;
};

// fused decoder forward (register-chained MFMA): raw[P,4] = (rgb | logits, occ)
struct LkDecodeArgs {
    int R, S, P;
    unsigned flags;
    const float* rays_o; const float* rays_d; const float* z;
    const float* c_geo; const float* c_col;       // [P,32]
    const float* W;                               // packed blob (plain master)
    const float* Wfrag;                           // fragment blob (lk_weights_repack)
    const float* affine;                          // [12] or NULL
    float* raw;                                   // [P,4]
    float* act;                                   // SAVE_ACT scratch or NULL
};

// relative-position neighbour MLP (colour features, Replica config)
struct LkRelposArgs {
    int R, S, P, min_nn;
    const float* rays_o; const float* rays_d; const float* z;
    const float4* sorted_unused;
    const float* pos;                             // [N,3] cloud positions (original order)
    const float* col_feats;                       // [N,32]
    const int32_t* nbr_idx; const float* nbr_w; const int32_t* nbr_count;
    const float* W; const float* Wfrag; const float* noise_col;
    float* c_col;                                 // [P,32]
};

int lk_launch_depth_stats(const float* gt, int R, int chunk, float* far_out, hipStream_t st);
int lk_launch_sample_interp(const LkSampleArgs& a, hipStream_t st);
int lk_launch_composite(const LkCompositeArgs& a, hipStream_t st);
int lk_launch_decode_fwd(const LkDecodeArgs& a, hipStream_t st);
int lk_launch_relpos_fwd(const LkRelposArgs& a, hipStream_t st);

// activation scratch layout (floats per sample), SAVE_ACT
//   [P][160] geometry a_i | [P][640] colour a_i | [P][640] colour h_i   (i = 0..4, row-major per sample)
//   a_i = act(W_i x_i + b_i), h_i = a_i + fc_c_i(c)
#define LK_ACT_GEO_A (5 * 32)
#define LK_ACT_COL_A (5 * 128)
#define LK_ACT_COL_H (5 * 128)
#define LK_ACT_FLOATS_PER_SAMPLE (LK_ACT_GEO_A + LK_ACT_COL_A + LK_ACT_COL_H)
