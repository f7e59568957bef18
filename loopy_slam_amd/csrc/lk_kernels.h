// Internal kernel argument blocks and launch prototypes (one translation unit per kernel family).
#pragma once
#include "lk_common.h"

// the piece type of a backward product chain: three bf16 pieces (lk_mma6: any scale) or two fp16 pieces of a PRE-SCALED chain (lk_mma3h: unit-scale
// loss gradients times 2^10) - decode_bwd_col_wg, decode_bwd_geo_wave (lk_bwd.hip), relpos_bwd_wave (lk_bwd2.hip)
template <bool H16> struct BwdPiece;
template <> struct BwdPiece<false> {
    typedef LkB8 T;
    static constexpr int NP = 3;
    static __device__ __forceinline__ T split(const f32x16& x, int G) { return lk_split_ct(x, G); }
    static __device__ __forceinline__ f32x16 mma(const T& a, const T& b, f32x16 c) { return lk_mma6(a, b, c); }
    static __device__ __forceinline__ T load(const u32x4* f, int NBT, int G, int nb, int lane) { return lk_fragb_load(f, NBT, G, nb, lane); }
    static __device__ __forceinline__ int tr(int idx) { return lkw::FRAG_TRB[idx]; }
};
template <> struct BwdPiece<true> {
    typedef LkH8 T;
    static constexpr int NP = 2;
    static __device__ __forceinline__ T split(const f32x16& x, int G) { return lk_split_cth(x, G); }
    static __device__ __forceinline__ f32x16 mma(const T& a, const T& b, f32x16 c) { return lk_mma3h(a, b, c); }
    static __device__ __forceinline__ T load(const u32x4* f, int NBT, int G, int nb, int lane) { return lk_fragh_load(f, NBT, G, nb, lane); }
    static __device__ __forceinline__ int tr(int idx) { return lkw::FRAG_TRH[idx]; }
};


// ---- optional per-kernel timing with HIP events on the launch stream (lk_profile_begin / lk_profile_end)
enum LkKernelId { LKK_DEPTH_STATS = 0, LKK_SAMPLE_INTERP, LKK_RELPOS_FWD, LKK_DECODE_FWD, LKK_COMPOSITE, LKK_COMPOSITE_BWD,
                  LKK_DECODE_BWD, LKK_RELPOS_BWD, LKK_INTERP_BWD, LKK_RAYS_BWD, LKK_WGRAD, LKK_FEAT_SCATTER, LKK_DECODE_BWD_TRACK, LKK_COUNT };
void lk_prof_before(int kid, hipStream_t st);
void lk_prof_after(int kid, hipStream_t st);
struct LkProfScope {
    int kid; hipStream_t st;
    LkProfScope(int k, hipStream_t s) : kid(k), st(s) { lk_prof_before(kid, st); }
    ~LkProfScope() { lk_prof_after(kid, st); }
};

struct LkSampleArgs {
    int R, S, P, stats_chunk;
    unsigned flags;
    const float* rays_o; const float* rays_d; const float* gt_depth; const float* r2_ray; const float* far_stats;
    const LkGrid* grid; const float4* sorted; const int32_t* cell_start;
    const float* geo_feats; const float* col_feats; const float* noise_geo; const float* noise_col;
    float near_surface, far_surface, near_end, r2_static;
    int min_nn;
    float* z; int32_t* nbr_idx; float* nbr_w; int32_t* nbr_count; float* c_geo; float* c_col;
    // rows counted per point on the way (first pass of the backward's counting sort, see k_seg_count); seg_cnt == NULL: off
    int32_t* seg_cnt; int32_t* seg_rank; const uint8_t* row_mask;
    // search-only launches over several iterations (lk_map_frame's chunks): sample p belongs to iteration y = p / seg_P and counts into
    // seg_cnt + y * seg_cnt_stride; rays behind the live prefix seg_live[y] of their iteration are left out (0 / NULL: one batch)
    int seg_P, seg_cnt_stride; const int32_t* seg_live;
    const int32_t* seg_key;                        // or NULL: as LkFeatScatterArgs::key_of
    // interpolation-only launches (mode 2): fragment repack of plain -> frag as a rider (k_interp_repack); NULL: none
    const float* rp_plain; float* rp_frag; int rp_block0;
    float* rp_copy_dst; int rp_copy_n, rp_block1;   // blocks >= rp_block1: rp_copy_dst[0 .. n) = rp_plain[..] (the stepped blob of the step rider, LkStepRider::w_next)
    int rp_m_lo, rp_m_hi, rp_skip_lo, rp_skip_hi;   // split step: the matrices [rp_m_lo, rp_m_hi) of the table are NOT repacked, the floats [rp_skip_lo, rp_skip_hi) not copied (0, 0: all)
    const int32_t* live_rays;                     // see LkRelposArgs: with it the sampler gives the skipped samples their colour feature (noise)
};

struct LkCompositeArgs {
    int R, S, min_nn;
    float coef;
    const float* raw; const float* z; const int32_t* nbr_count; const float* gt_depth;
    float* depth; float* var; float* color; uint8_t* valid_ray;
    // fused mapper loss (LK_FLAG_MAPPER_LOSS): NULL loss_out = off
    const float* gt_color; float w_color; int use_color;
    float* d_depth; float* d_color; float* loss_out;
    float* d_raw;                                  // with loss_out: also the composite BACKWARD of the loss gradient (d raw [P,4]), or NULL
    int keep_depth;                                // LK_FLAG_Z_GIVEN: the depth of rays without a reading is NOT zeroed (Renderer.py:197-198)
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
    const int32_t* live_rays;                     // as LkRelposArgs: tiles behind the live prefix of a partitioned batch are not decoded (the
                                                  // reference never renders those rays: they are filtered out before the render, Mapper.py:645-681)
    int tile_stride;                              // 0 = 32; k_relpos_decode_fwd with the composite inside: (32 / S) S - a tile holds whole rays,
                                                  // its lanes >= tile_stride idle
    unsigned* status;                             // LK_FLAG_CHECK_RANGE: lk_status_dev(), else NULL (no operand checks)
};

// relative-position neighbour MLP (colour features, Replica config)
struct LkRelposArgs {
    int R, S, P, min_nn;
    const float* rays_o; const float* rays_d; const float* z;
    const float* pos;                             // [N,3] cloud positions (original order)
    const float* col_feats;                       // [N,32]
    const int32_t* nbr_idx; const float* nbr_w; const int32_t* nbr_count;
    const float* W; const float* Wfrag; const float* noise_col;
    float* c_col;                                 // [P,32]
    int feats_f16;                                // LK_FLAG_FEATS_F16
    const int32_t* live_rays;                     // [1] or NULL: only the first *live_rays rays of the batch have a depth reading (k_pregather
                                                  // partitions the mapper's batches); the samples of the others are not processed
    unsigned* status;                             // as LkDecodeArgs::status
};

struct LkCompositeBwdArgs {
    int R, S, min_nn;
    float coef;
    const float* raw; const float* z; const int32_t* nbr_count; const float* gt_depth;
    const float* d_depth; const float* d_var; const float* d_color;
    float* d_raw;                                  // [P,4]
    int keep_depth;                                // as LkCompositeArgs
};

// tracker loss in two passes (lk_loop.hip: k_track_composite or the epilogue of k_relpos_decode_fwd, then - LkDecodeBwdArgs::tl_n_part - the prologue of k_decode_bwd)
struct LkTrackLossArgs {
    int R, S, min_nn;
    float coef, w_color;
    int use_color;
    const float* raw; const float* z; const int32_t* nbr_count; const float* gt_depth; const float* gt_color;
    float* depth; float* var; float* color; uint8_t* valid_ray;
    float* d_depth; float* d_color; float* d_raw; float* out4;
    float* resid;                 // [R] normalised residual of every ray (median: |gt - depth|, sign bit set for an absent ray)
    float* part;                  // [blocks][2] per-workgroup (sum of residuals, #present rays); median: part[0] = 10 x the median
    int median;                   // tracking.handle_dynamic: False (LK_TRACK_MEDIAN_MASK): k_track_median runs between the two passes
    float* row_part;              // [decoder-backward tiles][4] the loss row's terms per tile (pass 2 as the prologue of k_decode_bwd: summed by the
                                  // pose step, LkTrackFinalArgs::loss_part - 235 tiles x 4 atomics on one line stalled the geometry waves for 14 us)
};

struct LkDecodeBwdArgs {
    int R, S, P;
    unsigned flags;
    const float* rays_o; const float* rays_d; const float* z;
    const float* W; const float* Wfrag; const float* affine;
    const float* act; const float* raw; const float* d_raw;
    float* dc_geo; float* dc_col;                  // [P,32]
    float* dy_col;                                 // [5][P][128] d y_i = d h_i * softplus'(a_i): the rows the weight-gradient jobs stream (GRAD_WEIGHTS)
    float* dlogit;                                 // [P,4]   (GRAD_WEIGHTS)
    float* dp_embed;                               // [P,4]   (GRAD_RAYS) geometry-decoder embedding path
    float* dp_embed_col;                           // [P,4]   (GRAD_RAYS, colour stage) colour-decoder embedding path
    float* g_weights; float* g_affine;
    float* g_affine_part;           // [colour tiles][12] per-tile sums of d affine (g_affine != NULL): summed by the caller of the launch
                                    // (lk_launch_reduce_partials, or k_exposure_step in the tracking loop) - 782 tiles x 12 atomics on
                                    // twelve addresses took 108 us per 5 000-ray launch
    float* part_bg;                                // [n_blocks][288] per-workgroup partial sums of d embedder._B (GRAD_WEIGHTS)
    const int32_t* live_rays;                      // as LkDecodeArgs
    const float* dscale;                           // [1] device-side power of two on top of the fp16-piece form's 2^10 pre-scale (exposure encoding:
                                                   // the loss gradient is scaled by a LEARNED affine, lk_exposure_desc::bwd_scale), or NULL = 1
    int cb_on;                                     // 1: d_raw is NOT read and k_composite_bwd was NOT launched - every lane forms the composite backward
    LkCompositeBwdArgs cb;                         // of its own sample from the per-ray loss gradients (cb.d_depth / d_var / d_color; cb.d_raw unused)
    int ml_on;                                     // 1 (mapping loop, LK_COMPOSITE_IN_BWD): d_raw is NOT read and k_composite was NOT launched - every
    LkCompositeArgs ml;                            // lane composites its sample's ray from raw, forms the mapper's loss term (Mapper.py:691-720) and the
    float* ml_row_part;                            // composite backward of it; the geometry role writes the ray's outputs and the loss row's terms of
                                                   // its tile ([tiles][4], summed per iteration at the end of the lk_map_frame call)
    int tl_n_part;                                 // > 0 (tracking loop): d_raw is NOT read - every lane forms the tracker's loss term of its sample's ray
    LkTrackLossArgs tl;                            // and the composite backward of it from pass 1's per-ray outputs (tl.part: tl_n_part pairs); the
                                                   // geometry role adds the loss row (tl.out4)
};

struct LkTrackFinalArgs {
    int R, n_part;
    float fx, fy, cx, cy;
    const float* pose_part;                                        // [n_part][12] from k_interp_bwd
    float* cam; float* g_cam; float* adam_mv; float* hist_pre; float* hist_post;      // hist rows or NULL
    float step_T, step_q, bc2_sqrt, beta1, beta2, eps;            // lr / bias_correction1 per group, sqrt(bias_correction2)
    const float* next_pix_i; const float* next_pix_j; float* rays_o; float* rays_d;   // rays of the NEXT iteration's pixels, or NULL
    int do_update;                                                 // 0: only the rays of `cam` (before the first iteration)
    const float* cam_in; const float* mv_in;                       // or NULL: the pose / moments are read from here and written to cam / adam_mv
                                                                   // (the step as the prologue of the next iteration's search: every workgroup reads, one writes)
    const float* loss_part; int n_loss_part; float* log_row;       // or NULL: the stepped iteration's loss row = sum of [n_loss_part][4] (LkTrackLossArgs::row_part)
};
// interpolation backward: feature-row scatter (+ tracker: weights -> distances -> positions)
struct LkInterpBwdArgs {
    int R, S, P, min_nn;
    unsigned flags;
    const float* rays_o; const float* rays_d; const float* z; const float* r2_ray; float r2_static;
    const float* pos; const float* geo_feats; const float* col_feats;
    const int32_t* nbr_idx; const float* nbr_w; const int32_t* nbr_count;
    const float* dc_geo; const float* dc_col;
    const float* dw_rel;                           // [P,8] d loss / d normalised weight from the rel-pos branch, or NULL
    const float* dp_embed; const float* dp_embed_col; const float* dp_rel;    // [P,4] or NULL
    float* g_geo_feats; float* g_col_feats;        // [N,32] accumulated
    float* dp_total;                               // [P,4] (GRAD_RAYS)
    // optional (tracking loop): per-workgroup partial sums of the pose gradient's 12 ray moments, see LkBwdExtra
    float* pose_part; const float* pix_i; const float* pix_j; float fx, fy, cx, cy;
};
// Extras of the fused tracking loop for lk_render_bwd_impl: with pose_part the interpolation backward also reduces, per
// workgroup of 32 samples, G[c][k] = sum d p_c z dir_k (9) and T[c] = sum d p_c (3) - what k_pose_bwd sums over the rays -
// into pose_part[block][12]; lk_bwd_pose_parts(P) blocks.
// ---- optimiser step (lk_optim.hip / lk_adam_dev.h)
struct AdamSegDev {
    float* p; float* g; float* m; float* v; long long n; float step_size, bc2_sqrt;
    const int32_t* row_index; int row_len; int zero_grad; int p_f16;
    const uint8_t* row_flags;                      // or NULL: only flagged rows are stepped (lk_adam_seg::row_flags)
    int g_compact;                                 // g indexed like m / v (lk_adam_seg::g_compact)
};
// The Adam step of a mapper 'color' iteration as a rider of its reduction launch (lk_map_frame with no gradient exchange between the
// backward and the step): every decoder gradient element has exactly ONE owner thread in k_bwd_reduce, which steps the element as
// soon as it has its sum (new value -> w_next: the master blob is read by the fc_c blocks of the same launch and stays as it was
// until the next iteration's interpolation launch copies w_next over it), and the feature-row segments - final since the gather -
// run as extra blocks.  One dispatch (11 us + its gap) less per iteration.
struct LkStepSpan { int off, n; float step_size, bc2_sqrt; };
struct LkStepRider {
    int n_span;                                    // decoder spans (offsets into the blob); 0 = no rider
    LkStepSpan span[16];
    float* g; const float* p; float* w_next; float* m; float* v;     // blob-shaped arrays
    float beta1, beta2, eps;
    AdamSegDev feat[2]; int n_feat, feat_gx;       // feature-row segments: n_feat * feat_gx extra blocks
};
struct ExposureStepArgs;
struct LkBwdExtra { float* pose_part; const float* pix_i; const float* pix_j; float fx, fy, cx, cy;
                    int32_t* seg_list; int32_t* seg_total; const int32_t* live_rays; const LkStepRider* step; const float* dscale;
                    uint8_t* act_flag; int signal_rows; const LkTrackLossArgs* track_loss; int track_n_part;
                    float* loss_rows;              // LK_COMPOSITE_IN_BWD: LkDecodeBwdArgs::ml_row_part of this iteration
                    // SPLIT STEP (with `step`, a forked backward): the colour trunk's reduction + Adam + copy-back + fragment repack run on the
                    // weight-gradient stream behind k_wgrad and the JOIN IS LEFT TO THE CALLER (lk_render_fwd_impl(.., join_side_before_decode) of the
                    // next iteration, in front of its decoder launch); the launch stream only reduces / steps what its own kernels produced (rel-pos
                    // MLP, Fourier matrices, feature rows).  weights_frag_rw: the fragment buffer the side launch repacks into; *split_done = 1 if taken
                    int split_reduce; float* split_frag; float* split_master; int* split_done;
                    const ExposureStepArgs* xstep;      // or NULL: an exposure step that rides in this backward - in the gather launch (mapper:
                    const float* xstep_part; int xstep_n_part; };   // LkFeatScatterArgs::x) or, without feature gradients (tracker), as the last workgroup
                                                        // of the interpolation backward's launch, which then also sums the per-tile d affine (xstep_part)     // mapper loop: the iteration's sorted row list (lk_map_frame sorts ahead); signal_rows: lk_map_desc::signal_rows
inline int lk_bwd_pose_parts(int64_t P) { return (int)((P + 31) / 32); }

// One exposure step of the per-frame loops (lk_exposure_dev.h: lk_exposure_step_body)
struct ExposureStepArgs {
    float* feats; float* W1; float* b1; float* W2; float* b2; int F;
    float* aff; float* hid; float* g_aff; float* g; float* m; float* v; float* bwd_scale;
    float step_mlp, step_feat, bc2_sqrt, beta1, beta2, eps;      // lr / bias_correction1 per group (step_mlp < 0: frozen), sqrt(bias_correction2)
    int feat_first, feat_count, mode;
};

struct LkFeatScatterArgs {
    int P, min_nn;
    const int32_t* nbr_idx; const float* nbr_w; const int32_t* nbr_count;
    const float* dc_geo; const float* dc_col; const float* dfeat;
    float* g_geo_feats; float* g_col_feats;
    const uint8_t* row_mask;                       // [N] or NULL: scatter only into rows flagged non-zero
    // counting sort of the rows by point (lk_launch_seg_sort) -> gather without per-row atomics
    int32_t* seg_cnt;                              // [N + 1] (lk_knn_s::seg_cnt) rows per point: zero between calls (the scan clears it)
    int32_t* seg_off;                              // [N + 1] (lk_knn_s::seg_off) exclusive offsets of the points' rows; [N] = rows in the list
    int32_t* seg_sums;                             // scan scratch (lk_knn_s::seg_sums)
    int32_t* seg_rank;                             // [8P] rank of the row among the rows of its point, -1 = row takes no part
    int32_t* seg_list;                             // [8P] rows ordered by point
    int32_t* seg_total;                            // [1] number of rows in seg_list, written by k_seg_place (NULL: seg_off[N])
    // a batch of sorts in one launch (blockIdx.y; lk_map_frame sorts the rows of several iterations ahead of its loop): member y reads
    // nbr_* / live_rays / seg_total of iteration y (consecutive arrays) and uses seg_cnt / seg_off + y * cnt_stride, seg_sums + y * sums_stride
    int cnt_stride, sums_stride;
    // k_feat_gather rider: out[width] += column sums of part[n][width] (the geometry Fourier-matrix partials of k_decode_bwd), blocks >= red_block0
    const float* red_part; int red_n, red_width, red_block0; float* red_out;
    // k_feat_gather rider, first in the grid: k_dw2_hbar's blocks (dw2_part = NULL: none); samples = P, live prefix = *dw2_live rays of dw2_S samples (or NULL)
    // k_feat_gather rider, block 0 when x_on: the BACKWARD + Adam half of the mapping iteration's exposure step (x.mode = 1) - it needs d affine
    // only, final since the loss kernel; its 20-us chain of one workgroup then runs beside the gather instead of in the Adam launch
    int x_on; ExposureStepArgs x;
    const float* dw2_dc; const float* dw2_w_sum; const float* dw2_hbar; float* dw2_part; int dw2_blocks; const int32_t* dw2_live; int dw2_S;
    const int32_t* live_rays; int S;               // rows of rays >= *live_rays take no part (NULL: all)
    int N;
    uint8_t* act_flag;                             // or NULL: k_feat_gather flags every point it adds a gradient to (lk_knn_s::act_flag)
    const int32_t* key_of;                         // or NULL (key = point index, N = points): sort key of a point, < 0 = the point's rows take no part;
                                                   // N = number of keys (lk_knn_s::row_rank: the optimised rows of lk_map_frame)
};
int lk_launch_seg_sort(const LkFeatScatterArgs& a, bool counted, hipStream_t st, int batch = 1);        // counted: k_sample_interp already ran the count pass
int lk_launch_scan_i32(int32_t* data, int32_t* out, int32_t* block_sums, int total, hipStream_t st,
                       int batch = 1, int dstride = 0, int sstride = 0);   // out != data: data is cleared; batch members y at + y * stride

struct LkRaysBwdArgs { int R, S; const float* z; const float* dp_total; float* g_rays_o; float* g_rays_d; };

// rel-pos neighbour MLP backward
struct LkRelposBwdArgs {
    int R, S, P, min_nn;
    unsigned flags;
    const float* rays_o; const float* rays_d; const float* z;
    const float* pos; const float* col_feats;
    const int32_t* nbr_idx; const float* nbr_w; const int32_t* nbr_count;
    const float* W; const float* Wfrag;
    const float* dc_col;                           // [P,32]
    float* g_col_feats; float* g_weights;
    float* dw_rel;                                 // [P,8]  (GRAD_RAYS)
    float* dp_rel;                                 // [P,4]  (GRAD_RAYS)
    float* rows;                                   // [8P][192]: dhid(128) | x(64)  (GRAD_WEIGHTS)
    float* hbar;                                   // [P][128] sum_j w_j hid_j
    float* w_sum;                                  // [P] sum_j w_j
    float* dfeat;                                  // [8P][32] d loss / d feature row per neighbour (GRAD_FEATS)
    float* w_eff;                                  // [8P] weight actually applied to each neighbour row
    float* part_br;                                // [n_blocks][32] per-workgroup partial sums of d embedder_rel_pos._B
    float* dw1_part;                               // [n_blocks][128][64] per-workgroup d linear1 tiles (fused variant, lk_relpos_fused)
    const int32_t* live_rays;                      // as LkRelposArgs (fused variant only)
};
// k_relpos_bwd_fused (lk_bwd2.hip): linear1's weight gradient inside the rel-pos backward - mapper mode only (scaled fp16 pieces
// need unit-scale loss gradients; the ray-gradient products stay on the plain kernel)
#ifndef LK_RPF_MAX_PARTS
#define LK_RPF_MAX_PARTS 512                       // persistent workgroups: two per compute unit
#endif
static inline bool lk_relpos_fused(unsigned flags) {
    return (flags & LK_FLAG_GRAD_WEIGHTS) && (flags & LK_FLAG_UNIT_LOSS_GRADS) && !(flags & LK_FLAG_GRAD_RAYS);
}
int lk_relpos_bwd_parts(int P);                    // workgroups = partial tiles of the fused variant
// linear2 partial tiles (k_dw2_hbar); dw2_part: lk_dw2_part_floats(P) floats
int64_t lk_dw2_part_floats(int P);
int lk_launch_dw2_hbar(const LkRelposBwdArgs& a, float* dw2_part, hipStream_t st);

// weight gradients: dW[n][k] += sum_rows A[row][n] * B[row][k]  (one wave per (job, column unit, row chunk))
struct LkWgradJob {
    const float* A; int lda; int a_mode;           // 0 plain, 1 A*softplus'(A2), 2 A2[row]*A[row][n]
    const float* A2; int lda2;
    const float* B; int ldb;
    const float* B2; int ldb2; int k_split;        // optional second source for columns k >= k_split
    const float* B3; int ldb3; int k_split2;       // optional third source for columns k >= k_split2 (> k_split)
    int k_aux;                                     // > 0: columns k >= k_aux are not part of dW (their tiles feed fc_post_body only)
    int N, K;                                      // logical sizes (N <= 128, K <= 200 with the auxiliary columns)
    int rows;
    float* dW; int ldw;                            // plain blob matrix [N][ldw]
    float* db;                                     // bias gradient [N] or NULL
};
#define LK_WGRAD_MAX_JOBS 12
#define LK_WGRAD_MAX_UNITS 48
// fc_c weight gradients of the colour trunk WITHOUT their own reduction over the samples.  h_i = a_i + U_i c + u_i and
// d h_i = W_{i+1}^T d y_{i+1} (hidden columns of W_{i+1}; d h_4 = Wo^T d out), so
//   dU_i = sum_s d h_i[s] (x) c[s] = W_{i+1}^T (sum_s d y_{i+1}[s] (x) c[s]) = W_{i+1}^T M_{i+1},   du_i = W_{i+1}^T db_{i+1}:
// the job that streams d y_{i+1} anyway carries c as 32 auxiliary B columns (M_{i+1}, [128][32]) and a 128 x 128 x 32 product per
// layer finishes the job (fc_post_body) - the d h_i rows are neither stored by k_decode_bwd nor streamed a second time
// (64 MB written and read per 5 000-ray batch, a third of the weight-gradient kernel's HBM bytes).
struct LkFcPost {
    int src_job, rows_u, ldw, off;                 // M / db of job src_job ([rows_u][32], [rows_u]); W[u][off + v], row stride ldw
    const float* W;
    float* dU; float* du;                          // [128][32] += W^T M, [128] += W^T db
};
struct LkWgradUnit { int job, n0, k0; short nv, kv; int wave0, n_waves; };   // columns [n0, n0 + 32 nv) x [k0, k0 + 32 kv) of job; its wave slots PER XCD
struct LkWgradArgs {
    LkWgradJob job[LK_WGRAD_MAX_JOBS]; int n_jobs;
    int chunk;                                     // unused (kept for lk_wgrad_single's signature)
    LkWgradUnit unit[LK_WGRAD_MAX_UNITS]; int n_units, n_waves;   // filled by the launcher (n_waves: used slots per XCD)
    float* part;                                   // [n_waves][LK_WG_TILE] partial tiles (one per wave), or NULL (atomic flush)
    int h16;                                       // products on scaled fp16 pieces (unit-scale loss gradients only)
    LkFcPost fc[5]; int n_fc;                      // fc_c gradients finished from the auxiliary columns (needs `part`)
    const int32_t* live_rays; int S;               // rows >= *live_rays * S of every job are not read (their producers skipped them), or NULL
    const float* dscale;                           // as LkDecodeBwdArgs (h16 only)
};
static_assert(sizeof(LkWgradArgs) <= 3600, "LkWgradArgs travels as a kernel argument next to LkBwdReduceArgs (4 KB limit)");
#ifndef LK_WG_MAX_WAVES
#define LK_WG_MAX_WAVES 2048                       // waves of one weight-gradient launch: two per SIMD, all co-resident
#endif
#define LK_WG_TILE (4 * 16 * 64 + 64)              // floats per tile: accumulators [block][reg][lane] + bias sums
// floats of LkWgradArgs::part (one tile per wave, whatever the problem size)
inline int64_t lk_wgrad_part_floats(int64_t, bool) { return (int64_t)LK_WG_MAX_WAVES * LK_WG_TILE; }

// lk_render_fwd / lk_render_bwd with parts of their launch sequence left to the caller (the fused per-frame loops, lk_loop.hip)
enum { LK_SKIP_COMPOSITE = 1, LK_SKIP_COMPOSITE_BWD = 2, LK_SKIP_RAYS_BWD = 4, LK_FUSE_COMPOSITE_BWD = 8, LK_LOSS_PREZEROED = 16,
       LK_SEG_SORTED = 32 /* bwd: the forward (LK_FUSE_COMPOSITE_BWD + GRAD_FEATS) already sorted the rows by point */,
       LK_PRESAMPLED = 64 /* fwd: z / nbr_idx / nbr_w / nbr_count are given (lk_presample), only interpolate */,
       LK_SKIP_AFF_REDUCE = 256 /* bwd: the caller sums the per-tile d affine partials itself (k_track_final's exposure workgroup) */,
       LK_COMPOSITE_IN_BWD = 512 /* mapping loop (MAPPER_LOSS): fwd launches no composite; bwd forms d raw in the prologue of k_decode_bwd (LkBwdExtra::loss_rows) */,
       LK_FUSE_SMALL = 128 /* tracker-sized batches: rel-pos MLP + decoders in one launch (fwd), rel-pos backward + interpolation backward in one (bwd) */ };
// cnt: the batch holds n iterations of P_iter samples each (n <= LK_SEG_BATCH); their rows are counted per point on the way
struct LkPresampleCount { int P_iter; int32_t* seg_rank; const int32_t* live_rays; const int32_t* key_of; };
int lk_presample(const lk_render_desc* d, hipStream_t st, const LkPresampleCount* cnt = nullptr);
int lk_wait_rows_event(hipStream_t st);                     // lk_map_wait_rows
bool lk_serial_mode();                                      // LK_SERIAL / lk_set_serial: one stream only
// library-owned third stream (lk_map_frame's search ahead of the loop; small independent launches of the backward)
#define LK_PRE_CHUNKS 16
struct LkAuxStream { hipStream_t st = nullptr; hipEvent_t e0 = nullptr; hipEvent_t ev[LK_PRE_CHUNKS] = {}; hipEvent_t e1 = nullptr, e2 = nullptr; bool ok = false; };
LkAuxStream& lk_aux_stream();      // the search of a batch: z and the neighbour lists
// rider of the interpolation launch (LK_PRESAMPLED only): repack `src` (NULL: d->weights) into the fragment buffer `frag` (= d->weights_frag,
// writable) and, with copy_dst, copy src[0 .. copy_n) over copy_dst (= d->weights, writable) - the blob stepped by the step rider
struct LkRepackRider { float* frag; const float* src; float* copy_dst; int copy_n;
                       int skip_trunk; };      // non-zero: the colour trunk's spans and fragments were taken care of on the weight-gradient stream (split step)
int lk_launch_repack_trunk(const float* src, float* dst, float* frag, hipStream_t st);      // lk_weights.hip
int lk_render_fwd_impl(const lk_render_desc* d, hipStream_t st, unsigned skip, const int32_t* live_rays = nullptr, const LkRepackRider* repack = nullptr,
                       const LkTrackFinalArgs* pose = nullptr, const LkTrackLossArgs* comp = nullptr, int* comp_tiles = nullptr,
                       bool join_side_before_decode = false);      // the last: a split step of the iteration before is still running on the weight-gradient stream       // pose: the tracking loop's pose step as the prologue of the search launch
int lk_render_bwd_impl(const lk_render_desc* d, hipStream_t st, unsigned skip, const LkBwdExtra* ex = nullptr);
struct LkBwdOffsets { int64_t d_raw, dp_total, aff_part; };
LkBwdOffsets lk_bwd_offsets(int64_t P, uint32_t flags);      // float offsets of two regions of lk_render_desc::bwd_scratch

int lk_launch_composite_bwd(const LkCompositeBwdArgs& a, hipStream_t st);
int lk_launch_decode_bwd(const LkDecodeBwdArgs& a, hipStream_t st);
int lk_launch_interp_bwd(const LkInterpBwdArgs& a, hipStream_t st, const ExposureStepArgs* xstep = nullptr, const float* xstep_part = nullptr, int xstep_n_part = 0);
int lk_launch_rays_bwd(const LkRaysBwdArgs& a, hipStream_t st);
int lk_launch_feat_scatter(const LkFeatScatterArgs& a, hipStream_t st);
int lk_launch_relpos_bwd(const LkRelposBwdArgs& a, hipStream_t st);
int lk_launch_wgrad(const LkWgradArgs& a, int max_rows, hipStream_t st, LkWgradArgs* deferred = nullptr);   // deferred: skip the tile sums, return the unit table
// all partial-sum reductions of one backward in one launch (k_bwd_reduce, lk_bwd2.hip); block ranges are filled by the launcher
struct LkBwdReduceArgs {
    int b_wg, ny, b_rp, b_pg, b_pr, b_fc, b_ad;
    const float* part1; int n1; const float* part2; int n2; float* dW1; float* db1; float* dW2; float* db2;
    const float* part_bg; int n_bg; float* out_bg;
    const float* part_br; int n_br; float* out_br;
};
int lk_launch_bwd_reduce(const LkWgradArgs& wa, LkBwdReduceArgs r, bool with_rp, hipStream_t st, const LkStepRider* step = nullptr, bool feat_rows = true);
int lk_dw2_parts(int P);
int lk_launch_reduce_partials(const float* part, int n_parts, int width, float* out, hipStream_t st);
// geometry decoder weight gradients (LK_FLAG_GRAD_GEO_DECODER; lk_geo_wgrad.hip)
int64_t lk_geo_wgrad_part_floats(int P);
int lk_launch_geo_wgrad(int P, int S, const float* rays_o, const float* rays_d, const float* z, const float* W, const float* act,
                        const float* c_geo, const float* d_raw, float* part, float* g_weights, hipStream_t st, const int32_t* live_rays = nullptr);

int lk_launch_depth_stats(const float* gt, int R, int chunk, float* far_out, hipStream_t st);
int lk_launch_sample_interp(const LkSampleArgs& a, hipStream_t st, int mode = 0, const LkTrackFinalArgs* pose = nullptr);     // 1: search only, 2: interpolation of given lists; pose: k_sample_interp_pose
int lk_launch_composite(const LkCompositeArgs& a, hipStream_t st);
int lk_launch_decode_fwd(const LkDecodeArgs& a, hipStream_t st);
int lk_launch_relpos_fwd(const LkRelposArgs& a, hipStream_t st);
bool lk_relpos_decode_fusable(const LkDecodeArgs& a);
struct LkTrackLossArgs;
// comp (tracking loop): pass 1 of the tracker's loss - composite, residuals, per-TILE sums - as the epilogue of the launch; *comp_tiles = the
// number of (sum, count) pairs it left in comp->part, or 0 where the launch cannot carry it (the caller launches k_track_composite then)
int lk_launch_relpos_decode_fwd(const LkRelposArgs& ra, const LkDecodeArgs& a, hipStream_t st, const LkTrackLossArgs* comp = nullptr, int* comp_tiles = nullptr);       // k_relpos_fwd + k_decode_fwd in one launch
int lk_launch_relpos_interp_bwd(const LkRelposBwdArgs& rb, const LkInterpBwdArgs& ib, hipStream_t st, bool h16);  // k_relpos_bwd + k_interp_bwd in one launch

// activation scratch layout (floats per sample), SAVE_ACT
//   [P][160] geometry a_i | [P][320] colour softplus'(z_i) as unorm16 pairs | [P][640] colour h_i | [P][40] colour embedding  (i = 0..4)
//   a_i = act(z_i), z_i = W_i x_i + b_i, h_i = a_i + fc_c_i(c).  The colour trunk's backward needs a_i only for d y_i = d h_i softplus'(z_i):
//   the forward stores that factor in 16 bits (lk_pack_unorm16; round 5 stored the a_i rows, 64 MB per 5 000-ray batch written and read back)
#define LK_ACT_GEO_A (5 * 32)
#define LK_ACT_COL_A (5 * 128)       // mapper mode uses the first 5 * 64 words: 128 unorm16 per sample and layer, lane-contiguous (LK_COL_SLAYER,
                                     // decode_col_wg::finish); TRACKER MODE (LK_FLAG_TRACKER: the tracking loop, bundle adjustment) keeps the fp32 a_i rows
                                     // [layer][P][128] - the pose gradient goes through 2 pi B cos(2 pi p B), |B| ~ 25-32, and amplifies a
                                     // 7.6e-6 quantisation of the mask into visibly different pose trajectories (tests/test_steps_parity.py, BA case:
                                     // 0.75 % instead of 0.006 % of the geometry rows more than 1e-4 from the oracle's after four iterations),
                                     // and its 235-tile launches are latency-bound, not store-bound
#define LK_ACT_COL_H (5 * 128)
#define LK_ACT_COL_E 40       // colour Fourier embedding (input of layers 0 and 3)
#define LK_ACT_FLOATS_PER_SAMPLE (LK_ACT_GEO_A + LK_ACT_COL_A + LK_ACT_COL_H + LK_ACT_COL_E)
// The colour trunk's saved activations (a_i, h_i) and its d h_i rows are LAYER-MAJOR, [layer][P][128]: a job of the weight-gradient
// reduction then streams one contiguous [P][128] array (512-byte pieces of 2.5-KB rows cost it a third of its bandwidth), and the 32
// samples x 128 bytes a wave of the decoders stores per layer sit 512 bytes apart instead of 2 560.
// colour tiles up to which the decoders take their deep-prefetch form (lk_decode.hip / lk_bwd.hip)
#ifndef LK_DEEP_MAX_TILES
#define LK_DEEP_MAX_TILES 512
#endif
#ifndef LK_DEEP_MAX_TILES_FWD
#define LK_DEEP_MAX_TILES_FWD 512
#endif
#define LK_COL_LAYER(P, layer) ((size_t)(layer) * (size_t)(P) * 128)
#define LK_COL_SLAYER(P, layer) ((size_t)(layer) * (size_t)(P) * 64)       // the derivative mask: 64 words per sample and layer

// resident 256-thread workgroups per compute unit as the runtime computes them (registers, LDS): lk_debug_occupancy
int lk_occupancy_decode_fwd();
int lk_occupancy_relpos_fwd();
int lk_occupancy_decode_bwd();
int lk_occupancy_relpos_bwd_fused();
int lk_occupancy_wgrad();
