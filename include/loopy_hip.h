/* loopy_hip.h — C ABI of libloopyhip.so: the MI355X (gfx950) implementation of
 * Loopy-SLAM's per-frame neural-point render / optimise hot path.
 *
 * The reference (eriksandstroem/Loopy-SLAM) has no FFI; its seam for this path is the
 * Python API (Renderer.render_batch_ray, NICER.forward, NeuralPointCloud.find_neighbors_faiss,
 * torch.optim.Adam ...).  Each entry point below names the reference code it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless named host_*; the caller (PyTorch) owns all
 *    buffers; the library allocates only what an lk_knn_t handle owns;
 *  - every call returns 0 on success, <0 on error (lk_last_error() gives the message,
 *    thread-local); nothing throws across the ABI;
 *  - kernels are enqueued on `stream` (a hipStream_t; torch.cuda.current_stream().cuda_stream)
 *    and never synchronise the device; lk_render_bwd (and lk_render_fwd inside lk_map_frame) additionally fork part of their
 *    work onto ONE library-owned non-blocking stream, and lk_map_frame runs the neighbour search of its iterations ahead on a
 *    library-owned third stream; both are joined back into `stream` with events, so results are complete in stream
 *    order (see lk_set_serial to switch them off).  These streams and the per-point scratch of an lk_knn_t are shared state:
 *    concurrent calls from several host threads must be ordered by the caller;
 *  - fp32 everywhere; indices int32; R rays, S samples/ray (<= 8), P = R*S points in
 *    ray-major order (point r*S+s), k = 8 neighbours, C = 32 channels, N cloud points.
 */
#ifndef LOOPY_HIP_H
#define LOOPY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LK_ABI_VERSION 1
#define LK_K 8          /* pointcloud.nn_num   (configs/point_slam.yaml:136) */
#define LK_C 32         /* model.c_dim         (configs/point_slam.yaml:11)  */
#define LK_S_MAX 8      /* rendering.N_surface is 5 in every config          */

#define LK_OK 0
#define LK_ERR_ARG (-1)
#define LK_ERR_HIP (-2)
#define LK_ERR_STATE (-3)
#define LK_ERR_RANGE (-4)   /* an operand left the range of the split fp16 products (see lk_status_peek) */

typedef struct lk_knn_s* lk_knn_t;

int lk_version(void);
const char* lk_last_error(void);

/* ---------------------------------------------------------------- neighbour index
 * Replaces the FAISS-GPU IVF index of NeuralPointCloud (src/neural_point.py:67-72 create,
 * :1623-1627 train+add, :1659-1708 find_neighbors_faiss) by an exact uniform-grid search.
 * Contract (oracle/hotpath.py::knn_exact): d2 = (dx*dx+dy*dy)+dz*dz in fp32, candidates
 * d2 <= r2, the k=8 smallest by (d2, index) ascending, empty slots idx=-1 / d2=FLT_MAX,
 * count = #returned with d2 < r2.
 */
int lk_knn_create(float cell_size, int64_t capacity_points, int64_t max_cells, lk_knn_t* out);
int lk_knn_destroy(lk_knn_t h);
/* (Re)build the grid over pos[0..N).  Appending points = rebuilding with the larger N
 * (counting sort, O(N), device only, no host sync).  pos must stay alive and unchanged
 * until the next build (the grid keeps its own sorted copy; pos itself is not read by queries). */
int lk_knn_build(lk_knn_t h, const float* pos, int64_t N, void* stream);
int64_t lk_knn_size(lk_knn_t h);
/* Grow the indexed cloud by M points (they get the indices lk_knn_size() .. + M - 1): NeuralPointCloud.add_neural_points'
 * index.add (src/neural_point.py:1623-1627).  A uniform grid has no cheap in-place insert - this is lk_knn_build over
 * (points recovered from the grid's sorted copy | pos_new): O(N + M), device only, the handle keeps its own position buffer
 * from the first call on.  A caller that holds the grown position array anyway (NeuralPointCloud does) calls lk_knn_build. */
int lk_knn_append(lk_knn_t h, const float* pos_new, int64_t M, void* stream);
/* r2_per_query may be NULL (then r2_scalar is used for every query). */
int lk_knn_query(lk_knn_t h, const float* q, int64_t P, float r2_scalar, const float* r2_per_query,
                 float* out_d2 /*[P,8]*/, int32_t* out_idx /*[P,8]*/, int32_t* out_count /*[P]*/,
                 void* stream);

/* ---------------------------------------------------------------- decoder weights
 * NICER = MLP_geometry (hidden 32, relu) + MLP_color (hidden 128, softplus beta=100)
 * (src/conv_onet/models/decoder.py:106-288, 345-546).  The kernels read ONE packed fp32
 * blob; lk_weight_layout() describes it so the host can pack/unpack a state_dict.
 * Every matrix is stored [out][in_padded] (torch layout, input dim zero-padded to a
 * multiple of 8; the geometry skip layer's input is [e(93) pad 3 | h(32)]).
 */
typedef struct {
    char name[64];      /* reference state_dict key, e.g. "color_decoder.pts_linears.3.weight" */
    int64_t offset;     /* in floats, multiple of 64 */
    int32_t rows, cols; /* logical shape (cols = 1 for vectors) */
    int32_t ld;         /* padded row length in the blob */
    int32_t col_split;  /* if >0: logical cols [col_split..) start at padded col `col_shift` */
    int32_t col_shift;
} lk_weight_entry;
int lk_weight_layout(lk_weight_entry* out, int max_entries); /* returns #entries */
int64_t lk_weight_blob_floats(void);
/* The GEMM operands are streamed from a derived "fragment" copy of the blob (MFMA operand order, forward + transposed,
 * every weight as three bf16 pieces, the forward form also as two fp16 pieces; opaque to the caller: allocate lk_weight_frag_floats() floats, 16-byte aligned);
 * refresh it after every change of the master blob (optimiser step, load). */
int64_t lk_weight_frag_floats(void);
int lk_weights_repack(const float* blob, float* frag, void* stream);
/* The same, then `stream` is synchronised and the range check of the repack (LK_STATUS_WEIGHT_RANGE below) is reported: LK_ERR_RANGE with
 * the message in lk_last_error() when a matrix entry is non-finite or |w| >= 32768.  For the places where weights enter from outside
 * (checkpoint load, DecoderBlob.pack); the per-step repacks stay asynchronous and report through the sticky status word. */
int lk_weights_repack_checked(const float* blob, float* frag, void* stream);

/* ---------------------------------------------------------------- operand-range status (sticky)
 * The decoders' fp32 products run as split fp16 products on the 16-bit matrix pipe: exact to fp32 class for operands
 * below fp16's ceiling, SATURATING above it - the reference (src/conv_onet/models/decoder.py:513-546, plain fp32) has no such ceiling.
 * The library therefore keeps one status word (host-mapped memory, written by the kernels, read by the host without synchronisation):
 *   LK_STATUS_WEIGHT_RANGE  a repack (lk_weights_repack, the repack riders of lk_map_frame) met a matrix entry that is non-finite or
 *                           has |w| >= 32768;
 *   LK_STATUS_ACT_RANGE     a forward launched with LK_FLAG_CHECK_RANGE met an operand (interpolated feature, activation) that is
 *                           non-finite or has |x| >= 65504.
 * The word is STICKY: once a bit is set every lk_render_fwd / lk_render_bwd / lk_track_frame / lk_map_frame / lk_weights_repack call
 * fails with LK_ERR_RANGE (nothing is launched) until lk_status_clear().  A bit set by a launch that is still running shows at a later
 * call; lk_status_sync waits for `stream` first. */
#define LK_STATUS_WEIGHT_RANGE 1u
#define LK_STATUS_ACT_RANGE    2u
int lk_status_peek(uint32_t* bits);
int lk_status_sync(void* stream, uint32_t* bits);
int lk_status_clear(void);

/* ---------------------------------------------------------------- render forward / backward */
#define LK_FLAG_STAGE_COLOR   (1u << 0)  /* NICER stage 'color' (else 'geometry': rgb = 0)          */
#define LK_FLAG_TRACKER       (1u << 1)  /* is_tracker: gradient flows to the sample positions      */
#define LK_FLAG_REL_POS       (1u << 2)  /* model.encode_rel_pos_in_col                             */
#define LK_FLAG_COLOR_LOGITS  (1u << 3)  /* colour decoder returns pre-sigmoid logits (decoder.py:541-542) */
#define LK_FLAG_SAVE_ACT      (1u << 4)  /* forward keeps the activations lk_render_bwd needs        */
#define LK_FLAG_GRAD_FEATS    (1u << 5)  /* backward: d/d geo_feats, d/d col_feats                   */
#define LK_FLAG_GRAD_WEIGHTS  (1u << 6)  /* backward: d/d decoder blob                               */
#define LK_FLAG_GRAD_RAYS     (1u << 7)  /* backward: d/d rays_o, d/d rays_d (tracker / BA)          */
#define LK_FLAG_ALL_DEPTH_POS (1u << 8)  /* caller guarantees gt_depth > 0 for every ray            */
#define LK_FLAG_ZERO_ABSENT   (1u << 9)  /* rays with gt_depth <= 0 are ABSENT (filtered rays of a training batch kept for a
                                           * static shape: the losses ignore them): no far_bb statistics, their samples sit
                                           * between near_end and 0 */
#define LK_FLAG_MAPPER_LOSS   (1u << 10) /* lk_render_fwd also evaluates the mapper loss of the batch (Mapper.py:691-720, what
                                           * lk_loss_mapper computes) inside the composite kernel: reads loss_gt_color /
                                           * loss_w_color, writes d_depth, d_color and loss_out4 = [loss, geo, colour, #masked] */
#define LK_FLAG_UNIT_LOSS_GRADS (1u << 11) /* lk_render_bwd: the caller guarantees |d_color| <= ~1 (sum-type losses such as the mapper's
                                           * and the tracker's L1 colour terms; without ray gradients also |d_depth| <= ~1): the colour
                                           * decoder's backward may then run its products on fp16 pieces with a fixed 2^10 pre-scale
                                           * instead of bf16 pieces.  d_depth reaches the geometry decoder only, whose backward ignores
                                           * the flag */
#define LK_FLAG_Z_GIVEN       (1u << 12) /* lk_render_fwd: rays with gt_depth <= 0 take their S sample depths from `z` as the caller filled
                                         * it (rendering.sample_near_pcl, Renderer.py:152-160) instead of linspace(near_end, far_bb) */
#define LK_FLAG_FEATS_F16     (1u << 13) /* opt-in storage format: geo_feats / col_feats point at IEEE half tables [N,32] (64-byte rows; BASELINE
                                         * config 5 'fp16 features').  Everything computed from them, and their gradients, stays fp32 */
#define LK_FLAG_EMBED_GRADS_ONLY (1u << 14) /* lk_render_bwd, with LK_FLAG_GRAD_WEIGHTS: of the decoder blob only the Fourier matrices
                                           geo_decoder.embedder._B and (REL_POS) color_decoder.embedder_rel_pos._B receive gradients - the
                                           colour decoder's and the rel-pos MLP's matrices are frozen (Mapper.py:531-541 with
                                           fix_color_decoder, the end-of-sequence refinement): no weight-gradient rows are written and no
                                           weight-gradient reduction is launched; the rest of g_weights is left untouched */

#define LK_FLAG_CHECK_RANGE   (1u << 16) /* lk_render_fwd (debug): the decoder kernels test every operand they cut into fp16 pieces (interpolated
                                           features, activations) and set LK_STATUS_ACT_RANGE when one is non-finite or >= 65504 in magnitude */
#define LK_FLAG_GRAD_GEO_DECODER (1u << 15) /* lk_render_bwd, with LK_FLAG_GRAD_WEIGHTS: the geometry decoder's matrices and biases receive
                                           gradients too (mapping.fix_geo_decoder: False, Mapper.py:524-526; every reference config keeps
                                           them frozen and trains geo_decoder.embedder._B alone) - one more launch that redoes the 32-wide
                                           backward chain per sample in plain fp32 from the saved activations (lk_geo_wgrad.hip) */

typedef struct {
    /* ---- sizes */
    int32_t R, S;
    int32_t stats_chunk;        /* rays per far_bb group: R for a training batch, ray_batch_size (3000)
                                   for render_img (Renderer.py:102-121 is evaluated per batch) */
    uint32_t flags;
    /* ---- inputs */
    const float* rays_o;        /* [R,3] */
    const float* rays_d;        /* [R,3] */
    const float* gt_depth;      /* [R]   */
    const float* r2_ray;        /* [R] squared query radius per ray, or NULL -> r2_static */
    lk_knn_t knn;
    const float* pos;           /* [N,3] cloud positions in original order (rel-pos MLP, tracker gradients) */
    const float* geo_feats;     /* [N,32] */
    const float* col_feats;     /* [N,32] */
    const float* weights;       /* packed blob (master, plain [out][in_padded] matrices) */
    const float* weights_frag;  /* lk_weights_repack(weights) */
    const float* affine;        /* [12] exposure transform applied before the sigmoid, or NULL */
    const float* noise_geo;     /* [32] feature of samples without neighbours, or NULL (zeros) */
    const float* noise_col;     /* [32] */
    float near_surface, far_surface, near_end, coef, r2_static;
    int32_t min_nn;
    /* ---- outputs (Renderer.render_batch_ray: src/utils/Renderer.py:71-201) */
    float* depth;               /* [R]   */
    float* var;                 /* [R]   */
    float* color;               /* [R,3] */
    uint8_t* valid_ray;         /* [R]   */
    /* ---- state shared by forward and backward (caller allocated) */
    float* z;                   /* [R,S]   */
    int32_t* nbr_idx;           /* [P,8]   */
    float* nbr_w;               /* [P,8] normalised interpolation weights */
    int32_t* nbr_count;         /* [P]     */
    float* c_geo;               /* [P,32]  */
    float* c_col;               /* [P,32]  */
    float* raw;                 /* [P,4] rgb(or logits), occ */
    float* far_stats;           /* [ceil(R/stats_chunk)] */
    float* act;                 /* SAVE_ACT: lk_render_act_floats(R,S,flags) floats, else NULL */
    /* ---- backward inputs */
    const float* d_depth;       /* [R]   */
    const float* d_var;         /* [R] or NULL */
    const float* d_color;       /* [R,3] */
    /* ---- backward outputs (ACCUMULATED into: caller zeroes) */
    float* g_geo_feats;         /* [N,32] */
    float* g_col_feats;         /* [N,32] */
    float* g_weights;           /* blob-shaped */
    float* g_rays_o;            /* [R,3] (overwritten) */
    float* g_rays_d;            /* [R,3] (overwritten) */
    float* g_affine;            /* [12] accumulated, or NULL */
    const uint8_t* grad_row_mask; /* [N] or NULL: feature-row gradients are produced only for rows with a non-zero byte
                                  * (the frustum rows being optimised, Mapper.py:498-512) - the others are never read */
    /* ---- backward scratch */
    float* bwd_scratch;         /* lk_render_bwd_scratch_floats(R,S,flags) floats - the layout depends on `flags`: size it with the
                                 * flags of the call that uses it */
    int64_t bwd_scratch_cap;    /* floats available at bwd_scratch; checked against the layout of the call (0 = unchecked) */
    /* ---- LK_FLAG_MAPPER_LOSS */
    const float* loss_gt_color;  /* [R,3] */
    float* loss_out4;            /* [4] */
    float loss_w_color;
} lk_render_desc;

int64_t lk_render_act_floats(int32_t R, int32_t S, uint32_t flags);
int64_t lk_render_bwd_scratch_floats(int32_t R, int32_t S, uint32_t flags);
/* Renderer.render_batch_ray + NICER.forward + raw2outputs_nerf_color
 * (Renderer.py:71-201, decoder.py:573-610, common.py:382-422). */
int lk_render_fwd(const lk_render_desc* d, void* stream);
/* autograd backward of the same graph (Mapper.py:722, Tracker.py:193). */
int lk_render_bwd(const lk_render_desc* d, void* stream);

/* ---------------------------------------------------------------- losses (fused with d/d outputs)
 * Mapper (src/Mapper.py:691-720, non-exposure branch): mask = gt>0 & valid_ray & !nan(depth);
 *   loss = sum|gt-depth| + w_color*sum|gt_color-color| (colour term only if use_color).
 * Tracker (src/Tracker.py:169-191): u = |gt-depth|/sqrt(var+1e-10); mask = u < 10*mean(u) & gt>0 & !nan;
 *   loss = sum clamp(u,0,1e3) + w_color*sum|gt_color-color|.
 *   lk_loss_tracker's use_color is a flag word: LK_TRACK_USE_COLOR (the colour term takes part in the loss and its gradient) and
 *   LK_TRACK_MEDIAN_MASK (tracking.handle_dynamic: False, Tracker.py:177-179: mask = |gt-depth| < 10*median(|gt-depth|) & gt>0 & !nan,
 *   torch.median = the lower middle value, NaN if any residual is NaN; the loss terms are unchanged).
 * out_loss[0..3] = {loss, geo_loss, color_loss, #masked rays}; d_depth/d_color are overwritten. */
#define LK_TRACK_USE_COLOR   1
#define LK_TRACK_MEDIAN_MASK 2
int lk_loss_mapper(int32_t R, const float* depth, const float* color, const uint8_t* valid_ray,
                   const float* gt_depth, const float* gt_color, float w_color, int32_t use_color,
                   float* d_depth, float* d_color, float* out_loss, void* stream);
int lk_loss_tracker(int32_t R, const float* depth, const float* var, const float* color,
                    const float* gt_depth, const float* gt_color, float w_color, int32_t use_color,
                    float* d_depth, float* d_color, float* out_loss, float* scratch /*[R+8]*/, void* stream);

/* ---------------------------------------------------------------- Adam
 * torch.optim.Adam (amsgrad=False, weight_decay=0) on up to LK_ADAM_MAX_SEG tensors in one launch
 * (Mapper.py:570,723; Tracker.py:352,194).  `step` is the 1-based step count of that tensor. */
#define LK_ADAM_MAX_SEG 16
typedef struct {
    float* p; float* g; float* m; float* v;   /* m, v: compact optimiser state of n floats */
    int64_t n;                                /* elements updated by this segment */
    float lr; int32_t step;
    const int32_t* row_index;                 /* optional: element i lives at p[row_index[i/row_len]*row_len + i%row_len]
                                                 (frustum-selected feature rows updated in place in the full table) */
    int32_t row_len;
    int32_t zero_grad;                        /* clear the consumed gradient entries */
    int32_t p_f16;                            /* p is an IEEE half array (LK_FLAG_FEATS_F16 tables): read as fp32, stepped, rounded to nearest */
    const uint8_t* row_flags;                 /* optional (row_index must be NULL, n a multiple of row_len): [n / row_len] bytes, only the rows
                                                 with a non-zero flag are stepped.  Exact whenever the skipped rows have a zero gradient and zero
                                                 moments (Adam leaves such an element bit for bit where it is): whole-map refinement over millions
                                                 of rows of which an optimize_map call touches a few per cent (lk_map_frame sets the flags) */
    int32_t g_compact;                        /* with row_index: g is indexed like m and v (element i), not like p - the gradient rows arrive
                                                 compact in row-list order (a data-parallel caller's all-reduced bucket); zero_grad is ignored */
} lk_adam_seg;
int lk_adam_step(const lk_adam_seg* host_segs, int32_t n_seg, float beta1, float beta2, float eps, void* stream);

/* Exposure encoding (model.encode_exposure, ScanNet): mlp_exposure = Linear(8,128) -> Softplus(100) -> Linear(128,12)
 * (src/conv_onet/models/decoder.py:534-540) evaluated for F exposure features at once (the keyframes of the mapping window,
 * Mapper.py:588-607, or the tracker's frame, Tracker.py:329-344): aff [F,12] = (3x3 rot | 3 trans), hid [F,128] kept for
 * the backward.  lk_exposure_bwd turns d loss / d aff into g = [W1 1024 | b1 128 | W2 1536 | b2 12 | feats F*8].
 * lk_loss_mapper_exposure is lk_loss_mapper of the colour stage with the per-keyframe affine on the rendered colour LOGITS
 * (Mapper.py:697-715): out d_logits and g_aff [F,12] besides d_depth and [loss, geo, colour, #masked]. */
#define LK_EXPOSURE_MAX_F 32
#define LK_EXPOSURE_GRAD_FLOATS (1024 + 128 + 1536 + 12 + LK_EXPOSURE_MAX_F * 8)
int lk_exposure_fwd(const float* feats, const float* W1, const float* b1, const float* W2, const float* b2, int32_t F,
                    float* aff, float* hid, void* stream);
int lk_exposure_bwd(const float* feats, const float* W1, const float* W2, const float* hid, const float* g_aff, int32_t F,
                    float* g, void* stream);
int lk_loss_mapper_exposure(int32_t R, const float* depth, const float* logits, const uint8_t* valid_ray, const float* gt_depth,
                            const float* gt_color, const int32_t* frame_id, const float* aff, int32_t F, float w_color,
                            float* d_depth, float* d_logits, float* out_loss, float* g_aff, void* stream);

/* Gradient exchange bucket of the ray-sharded data-parallel step (the all-reduce payload of Mapper.py:722-724's
 * step = decoder-gradient spans + the feature-gradient rows being optimised): segment i is `n` floats at `data`
 * (row_index NULL) or the rows data[row_index[k]] of a [*, row_len] table (n = rows * row_len); the bucket is the
 * concatenation of the segments.  unpack = 0: bucket <- segments, 1: segments <- bucket, 2: bucket <- segments and the copied source
 * elements are cleared (for lk_map_desc::grad_bucket: the step then reads the bucket and nothing is unpacked).  One launch either way. */
typedef struct lk_copy_seg {
    float* data;
    int64_t n;
    const int32_t* row_index;
    int32_t row_len;
} lk_copy_seg;
int lk_bucket_copy(const lk_copy_seg* host_segs, int32_t n_seg, float* bucket, int32_t unpack, void* stream);

/* ---------------------------------------------------------------- rays / pose / compaction
 * get_camera_from_tensor + get_rays_from_uv (src/common.py:301-343,104-120): cam = (qw,qx,qy,qz,tx,ty,tz). */
int lk_rays_from_pose(const float* cam7, const float* pix_i, const float* pix_j, int32_t R,
                      float fx, float fy, float cx, float cy, float* rays_o, float* rays_d, void* stream);
/* One launch for a multi-keyframe ray batch: get_samples/get_sample_uv/select_uv/get_rays_from_uv
 * (src/common.py:104-172,237-259; Mapper.py:625-665, Tracker.py:142-146).  Ray r reads frame frame_id[r]
 * (NULL: frame 0) of the stacked images at window pixel rnd[r] in [0, h*w) (row-major over the window
 * [H0,H0+h) x [W0,W0+w)); c2w_stack holds row-major [3|4][4] matrices, c2w_stride floats apart.
 * pix_i/pix_j/r2_ray/r2_map_stack/frame_id may be NULL. */
int lk_gather_rays(const float* depth_stack, const float* color_stack, const float* c2w_stack, int32_t c2w_stride,
                   const float* r2_map_stack, const int32_t* frame_id, const int32_t* rnd, int32_t R,
                   int32_t H, int32_t W, int32_t H0, int32_t W0, int32_t w, float fx, float fy, float cx, float cy,
                   float* rays_o, float* rays_d, float* gt_depth, float* gt_color, float* pix_i, float* pix_j,
                   float* r2_ray, void* stream);
/* d loss / d cam7 from d rays (overwrites g_cam7[7]). */
int lk_pose_bwd(const float* cam7, const float* pix_i, const float* pix_j, int32_t R,
                float fx, float fy, float cx, float cy, const float* g_rays_o, const float* g_rays_d,
                float* g_cam7, void* stream);
/* Stable stream compaction (wave ballot + prefix sum): out_index[0..count) = i with mask[i]!=0. */
int lk_compact(const uint8_t* mask, int32_t n, int32_t* out_index, int32_t* out_count, void* stream);
/* lk_compact for masks of millions of entries (many-workgroup count / scan / scatter; same result).  block_scratch:
 * ceil(n / 256) ints. */
int lk_compact_large(const uint8_t* mask, int32_t n, int32_t* out_index, int32_t* out_count, int32_t* block_scratch, void* stream);
/* Rows of the [N,32] feature tables a batch's neighbour lists refer to: flags[i] = 1 for every 0 <= nbr_idx[j] = i < N, j < n
 * (flags is cleared by the caller).  When the whole map is optimised (final refinement, Mapper.py:884-897; BASELINE config 5)
 * a ray-sharded multi-GPU step exchanges the gradient rows of the union of the ranks' flags only. */
int lk_touch_rows(const int32_t* nbr_idx, int64_t n, uint8_t* flags, int32_t N, void* stream);
/* ORs flags[0..n) into the index's per-row flags of the running optimize_map call (rows whose Adam state may be non-zero: lk_map_frame
 * with rows == NULL steps only these; it clears them at the call's first iteration and sets the rows its own batches touch).  A
 * data-parallel caller passes the MAX-reduced flags of lk_touch_rows: rows only another rank touched receive gradient through the exchange. */
int lk_knn_flag_rows(lk_knn_t knn, const uint8_t* flags, int64_t n, void* stream);
/* ---------------------------------------------------------------- map maintenance around the hot loop
 * Frustum row selection, Mapper.get_mask_from_c2w (src/Mapper.py:165-217): out_index[0..*out_count) = ascending indices
 * of the points of pos[N,3] that project inside the image of the pose (cropped by `edge` pixels, negative = enlarged)
 * and lie no more than 0.5 m behind the observed depth (bilinear lookup as cv2.remap INTER_LINEAR / constant 0 border;
 * zero lookups are replaced by the largest lookup).  w2c12_host = rows 0..2 of inv(c2w) (HOST pointer, 12 floats;
 * the projection itself runs in float64 like the reference's numpy code).  Scratch: float[N], uint8[N], uint32[1]. */
int lk_frustum_rows(const float* pos, int32_t N, const float* w2c12_host, const float* depth, int32_t H, int32_t W,
                    float fx, float fy, float cx, float cy, int32_t edge, float* scratch_depth, uint8_t* scratch_mask,
                    uint32_t* scratch_max, int32_t* out_index, int32_t* out_count, void* stream);
/* Point insertion, geometry part of NeuralPointCloud.add_neural_points (src/neural_point.py:1557-1631): a ray with
 * gt_depth > 0 is accepted iff NO point of the index `knn` (NULL or empty: accept all) lies at squared distance
 * < r2 (r2_per_ray[i] if given, else r2_static) from o + d*gt_depth; accepted ray j (ascending ray order) emits n_add
 * points o + d*z, z = linspace(near_surface*depth, far_surface*depth, n_add), at out_points[(j*n_add + q)*3].
 * out_ray_index[0..*out_count) = accepted rays; out_points must hold 3*n_add*n floats.  Rays of one call are not
 * de-duplicated against each other (as in the reference).  Feature rows of the new points are the caller's
 * (N(0, 0.1), neural_point.py:1614-1617), followed by lk_knn_build on the grown cloud. */
int lk_add_points(lk_knn_t knn, const float* rays_o, const float* rays_d, const float* gt_depth, int32_t n,
                  float r2_static, const float* r2_per_ray, float near_surface, float far_surface, int32_t n_add,
                  uint8_t* scratch_mask, int32_t* out_ray_index, int32_t* out_count, float* out_points, void* stream);
/* Per-frame image pre-passes (Tracker.py:243-268, Mapper.py:854-872, common.py:175-234).
 * lk_radius_maps: grad_mag[H,W] = |Sobel(rgb2gray(color))| (reflected borders), and the dynamic radii as SQUARED
 * float32 maps: r_add = lerp over [0, 0.01, thr] -> [max, max, min] of the clipped magnitude, r_query = ratio * r_add
 * (r2_add / r2_query may be NULL).
 * lk_top_grad_pixels: the K pixels of largest grad_mag over the whole image (ties at the cut in ascending flat index),
 * kept only inside the window [H0,H1) x [W0,W1) and where depth > 0 (depth may be NULL; depth_limit: also <= 5):
 * out_index[0..*out_count) ascending flat indices (out_index must hold K ints).  Images above 16 384 pixels use a library-owned
 * scratch, one per DEVICE (grown with a stream synchronisation when a larger image first arrives): on one device the calls must be
 * ordered - one stream, or streams ordered by events - like the calls on one lk_knn_t handle. */
int lk_radius_maps(const float* color, int32_t H, int32_t W, double color_grad_threshold, double radius_add_max,
                   double radius_add_min, double radius_query_ratio, float* grad_mag, float* r2_add, float* r2_query, void* stream);
int lk_top_grad_pixels(const float* grad_mag, int32_t H, int32_t W, int32_t K, int32_t H0, int32_t H1, int32_t W0, int32_t W1,
                       const float* depth, int32_t depth_limit, int32_t* out_index, int32_t* out_count, void* stream);
/* thr = min(10*median(depth), 1.2*max(depth)) over depth>0 (Tracker.py:153-155, Mapper.py:674-676);
 * mask[i] = depth[i] > 0 && depth[i] <= thr (mask may be NULL); depth_filtered[i] = mask ? depth : 0 (may be
 * NULL or alias depth): a ray with gt_depth 0 is "absent" for the losses, which keeps the batch shape static
 * (no host sync) while matching the reference's boolean-index filtering.  scratch: n uint32. */
int lk_inside_mask(const float* depth, int32_t n, uint8_t* mask, float* depth_filtered, float* out_thr,
                   uint32_t* scratch, void* stream);

/* Exposure encoding inside the per-frame loops (lk_track_frame / lk_map_frame): everything lk_exposure_fwd / _bwd and the Adam groups of
 * mlp_exposure and the exposure features need (Tracker.py:329-344; Mapper.py:524-570, 588-607), all caller-owned.  Per iteration the
 * loops run ONE small launch for it: backward of the MLP from g_aff, Adam on its tensors and on the trainable features, forward with
 * the stepped values for the next iteration, g_aff cleared. */
typedef struct {
    float* feats;               /* [F,8] exposure features (tracker: F = 1; mapper: the keyframes of the window, the current frame last) */
    float* W1; float* b1; float* W2; float* b2;      /* mlp_exposure: [128,8], [128], [12,128], [12] - stepped in place */
    int32_t F;                  /* 1 .. LK_EXPOSURE_MAX_F */
    float* aff;                 /* [F,12] */
    float* hid;                 /* [F,128] */
    float* g_aff;               /* [F,12] d loss / d aff of the iteration (accumulated by the loss / decoder kernels) */
    float* g;                   /* [LK_EXPOSURE_GRAD_FLOATS] layout of lk_exposure_bwd */
    float* adam;                /* [2][LK_EXPOSURE_GRAD_FLOATS] exp_avg | exp_avg_sq in the layout of g; zeroed by the caller (fresh optimiser) */
    float lr_mlp;               /* Adam rate of the four MLP tensors (tracker 1e-3; mapper: decoders_lr of stage 'color'); < 0: frozen */
    float lr_feat;              /* Adam rate of the trainable features (1e-3) */
    int32_t feat_first, feat_count;      /* features [feat_first, feat_first + feat_count) are Adam parameters (mapper: only the last) */
    float* bwd_scale;           /* [1] or NULL.  With it the colour decoder's backward and the weight-gradient reductions of the loops run on
                                   pre-scaled fp16 pieces as under LK_FLAG_UNIT_LOSS_GRADS although the loss gradient passes through the
                                   LEARNED affines: every forward of the MLP stores here the power of two that maps 3 max|A| into (0.5, 1]
                                   - the bound of |d out| relative to the plain colour loss - and the kernels apply it on top of their 2^10 */
} lk_exposure_desc;

/* ---------------------------------------------------------------- per-frame optimisation loops
 * The launch sequences of the reference's two inner loops as ONE call each, so that the host enqueues a frame's work in
 * microseconds instead of interpreting ~20 Python statements per iteration (at 1 500-ray tracking batches the Python loop
 * was slower than the GPU).  Static shapes, no host synchronisation inside; every buffer is the caller's.
 *
 * lk_track_frame: Tracker.run's loop body for one frame (src/Tracker.py:313-401) = iters x { pixel gather + rays of the
 * current pose + inside mask (Tracker.py:142-160), lk_render_fwd in tracker mode, tracker loss (Tracker.py:169-191),
 * lk_render_bwd w.r.t. the rays, pose gradient, Adam on (T | quaternion) (Tracker.py:317-352) }.  For R <= 8192 the small
 * steps run as three one-workgroup kernels (batch assembly; composite + loss + composite backward; ray / pose gradient +
 * Adam) - 9 launches per iteration instead of 16.
 * render: R, S, the map (knn, pos, tables, weights), scalars, and ALL per-call buffers of lk_render_fwd / lk_render_bwd
 * (rays_o, rays_d, gt_depth, r2_ray or NULL, outputs, state, act, d_depth, d_color, g_rays_o, g_rays_d, bwd_scratch sized
 * for flags | TRACKER | STAGE_COLOR | SAVE_ACT | GRAD_RAYS | ZERO_ABSENT); render.flags carries LK_FLAG_REL_POS only. */
typedef struct {
    lk_render_desc render;
    const float* depth_img;     /* [H,W]   */
    const float* color_img;     /* [H,W,3] */
    const float* r2_map;        /* [H,W] squared dynamic query radius per pixel, or NULL */
    int32_t H, W, H0, W0, w;    /* draw q of rnd -> pixel (row H0 + q / w, column W0 + q % w) */
    float fx, fy, cx, cy;
    const int32_t* rnd;         /* [iters][R] pixel draws (window pixels, or flat image indices with H0 = W0 = 0, w = W) */
    float* gt_color;            /* [R,3] */
    float* pix_i; float* pix_j; /* [R]   */
    float* thr;                 /* [1]   */
    uint32_t* scratch_u32;      /* [R]   */
    float* loss_scratch;        /* [R+8] */
    float* cam7;                /* [7] in: initial pose (qw,qx,qy,qz,tx,ty,tz); out: pose after the last iteration */
    float* g_cam7;              /* [7] */
    float* adam_mv;             /* [14] exp_avg | exp_avg_sq of the pose; zeroed by the call (fresh optimiser per frame) */
    float lr_T, lr_q;           /* separate_LR: cam_lr and 0.2 cam_lr (Tracker.py:317-333); otherwise both cam_lr */
    float w_color; int32_t use_color;   /* flag word as for lk_loss_tracker: LK_TRACK_USE_COLOR | LK_TRACK_MEDIAN_MASK */
    int32_t hist_post;          /* 0: hist[it] = pose BEFORE iteration it (separate_LR: the candidate is a detached copy);
                                   1: pose AFTER its update (one leaf tensor stepped in place, Tracker.py:375-377) */
    float* hist;                /* [iters][7] candidate poses */
    float* log;                 /* [iters][4] loss, geo, colour, #masked rays per iteration; complete when the call's work has
                                 * completed (a row may be written by the iteration AFTER its own: the pose step sums its terms) */
    int32_t iters;
    float* work;                /* lk_track_work_floats(R, S, iters) floats: with it (and R <= 8192) every iteration's pixels, colours,
                                   radii and inside mask are assembled by ONE launch up front and the small steps of an iteration run
                                   fused (4 launches per iteration instead of 16); gt_color / pix_i / pix_j / thr / scratch_u32 /
                                   loss_scratch and render.g_rays_o / g_rays_d are then not used and may be NULL.  NULL: the
                                   per-iteration launch sequence */
    const lk_exposure_desc* exposure;   /* model.encode_exposure (HOST pointer) or NULL: the frame's affine is applied per sample inside the
                                   colour decoder (decoder.py:534-540); feature and MLP are stepped every iteration (Tracker.py:329-344) */
} lk_track_desc;
int64_t lk_track_work_floats(int32_t R, int32_t S, int32_t iters);
int lk_track_frame(const lk_track_desc* d, void* stream);

/* lk_map_frame: the joint iterations [it_begin, it_end) of one Mapper.optimize_map call (src/Mapper.py:576-735) = per iteration { multi-keyframe ray gather, inside mask, lk_render_fwd with the fused mapper loss, lk_render_bwd,
 * Adam over {decoder spans, geometry rows, colour rows}, fragment repack in stage 'color' }.  Iteration it runs stage
 * 'geometry' iff it < n_geo_iters.  phases: 1 = forward/backward only, 2 = optimiser step only, 3 = both (a ray-sharded
 * multi-GPU caller all-reduces the gradients between phase 1 and phase 2 of every iteration).
 * render: as for lk_track_frame (flags: LK_FLAG_REL_POS, + LK_FLAG_UNIT_LOSS_GRADS if wanted; g_geo_feats, g_col_feats,
 * g_weights, grad_row_mask, bwd_scratch sized for STAGE_COLOR | SAVE_ACT | GRAD_FEATS | GRAD_WEIGHTS). */
#define LK_MAX_SPANS 8
typedef struct { int64_t offset, n; } lk_blob_span;       /* floats [offset, offset + n) of the weight blob */
typedef struct {
    lk_render_desc render;
    const float* depth_stack; const float* color_stack; const float* c2w_stack; int32_t c2w_stride;
    const float* r2_map_stack;  /* or NULL */
    const int32_t* frame_id;    /* [R] keyframe of every ray */
    const int32_t* rnd;         /* [iters][R] */
    int32_t H, W, H0, W0, w;
    float fx, fy, cx, cy;
    float* gt_color; float* thr; uint32_t* scratch_u32;
    float w_color;
    float* log;                 /* [iters][4] as lk_track_desc::log; the rows of a call's iterations are complete at the END of that
                                 * lk_map_frame call (one launch sums the per-tile terms the decoder backward left) */
    /* optimiser (a fresh Adam per optimize_map call, Mapper.py:570: the caller zeroes the state buffers) */
    float* weights_rw;          /* = render.weights, writable */
    float* weights_frag_rw;     /* = render.weights_frag, writable */
    float* geo_feats_rw; float* col_feats_rw;     /* = render.geo_feats / col_feats, writable */
    const int32_t* rows;        /* [n_rows] rows being optimised, or NULL = all N rows (then n_rows = N) */
    int64_t n_rows;
    float* adam_rows;           /* [4][n_rows*32]: exp_avg, exp_avg_sq of the geometry rows, then of the colour rows */
    lk_blob_span geo_dec[LK_MAX_SPANS]; int32_t n_geo_dec;   /* decoder spans stepped in both stages (embedder._B) */
    lk_blob_span col_dec[LK_MAX_SPANS]; int32_t n_col_dec;   /* decoder spans stepped in stage 'color' */
    float* adam_dec;            /* [2][lk_weight_blob_floats()]: exp_avg | exp_avg_sq, blob-shaped */
    float lr[2][3];             /* [stage: geometry, colour][decoders, geometry rows, colour rows] (configs mapping.stage.*) */
    int32_t iters, n_geo_iters;
    float* work;                /* lk_map_work_floats(R, S, iters) floats, or NULL: as lk_track_desc.work - the batches (pixels, rays, colours,
                                   radii, inside masks) of all `iters` iterations are assembled by one launch of the call that starts at
                                   it_begin = 0 (gt_color / thr / scratch_u32 and render.rays_o / rays_d / gt_depth are then unused) */
    const lk_exposure_desc* exposure;   /* model.encode_exposure (HOST pointer) or NULL: the 'color' iterations render colour LOGITS and the loss
                                   applies sigmoid(logits @ rot_f + trans_f) of the ray's keyframe f = frame_id (Mapper.py:697-715);
                                   needs `work`; bwd_scratch sized WITHOUT LK_FLAG_UNIT_LOSS_GRADS (d logits = w sigma' A is unbounded) */
    const float* grad_bucket;   /* phase-2 calls of a data-parallel caller (rows != NULL, no exposure): the all-reduced gradient bucket or NULL.  With
                                   it the step reads its gradients from the bucket itself - decoder span k of the geometry / colour list at float
                                   offset bucket_geo_dec[k] / bucket_col_dec[k], the optimised rows of the two tables compact (row-list order) at
                                   bucket_geo_rows / bucket_col_rows - and no unpack launch is needed; the caller packs with lk_bucket_copy mode 2
                                   (copy + clear the source: what the step's zero_grad would have done) */
    int64_t bucket_geo_dec[LK_MAX_SPANS], bucket_col_dec[LK_MAX_SPANS];
    int64_t bucket_geo_rows, bucket_col_rows;
    int32_t union_rows_flagged; /* rows == NULL, phase-split (data-parallel) callers: non-zero = between phase 1 and phase 2 of every iteration
                                   the caller ORs the union of the rows ALL ranks touched into the index's row flags (lk_knn_flag_rows); the
                                   step then visits the flagged rows only, as the single-process call does on its own (0: dense step) */
    int32_t it_offset;          /* a long optimize_map call may be issued as consecutive SEGMENTS, one descriptor each: this one covers the
                                   iterations it_offset .. it_offset + iters - 1 of the call (rnd, log and work are the segment's own; iters
                                   sizes them), n_geo_iters stays the call's GLOBAL count, the optimiser state (adam_rows, adam_dec) carries
                                   over and the step counts continue.  The work buffer then holds one segment's batches and neighbour
                                   lists instead of the whole call's (26 floats per sample and iteration).  0 for an unsegmented call */
    int32_t batches_ready;      /* non-zero: lk_map_prepare has already assembled this descriptor's batches (see there) */
    int32_t train_geo_decoder;  /* mapping.fix_geo_decoder: False (Mapper.py:524-526): the geometry decoder's own matrices and biases are parameters too - they are
                                   listed in geo_dec (all of the decoder's used tensors), every backward also forms their gradients
                                   (LK_FLAG_GRAD_GEO_DECODER: size render.bwd_scratch with that flag), and the matrix fragments are refreshed after the
                                   'geometry' steps as well */
    int32_t signal_rows;        /* phase-1 calls of a data-parallel caller: non-zero = the backward records a library-owned event on the launch stream
                                   right behind the feature-row gather - the point from which g_geo_feats / g_col_feats of the iteration are final,
                                   while the weight-gradient launch and the reduction of the decoder gradients are still to run.  lk_map_wait_rows
                                   makes another stream wait for it: the caller exchanges the row part of its gradient bucket there, beside the
                                   tail of the backward (loopy_slam_amd/parallel.py) */
} lk_map_desc;
int64_t lk_map_work_floats(int32_t R, int32_t S, int32_t iters);
/* The batch assembly of lk_map_frame's first call (pixels, rays, colours, radii, inside masks of all `iters` iterations: one launch) AHEAD of
 * that call: it reads the batch inputs of the descriptor only (stacks, frame_id, rnd, window, intrinsics, render.R / S / r2_ray, work, iters,
 * log, exposure) - not the row list, not the map.  A caller whose row list comes out of a device-side selection with a count read-back
 * (Mapper.get_mask_from_c2w -> lk_frustum_rows) enqueues this BEFORE the read-back, so that the device assembles batches while the host
 * waits and fills in the rest of the descriptor; the lk_map_frame call then carries batches_ready = 1. */
int lk_map_prepare(const lk_map_desc* d, void* stream);
/* float offset inside `work` of the neighbour lists lk_map_frame keeps per iteration: int32 [iters][R*S][8] (what a data-parallel
 * caller needs to agree on the touched rows of an iteration, loopy_slam_amd/parallel.py) */
int64_t lk_map_work_nbr_idx(int32_t R, int32_t S, int32_t iters);
/* One render / backward sequence per kNN handle at a time: the handle owns the row counters of the gradient sort (zero between calls: the
 * scan that consumes them clears them) and lk_map_frame's look-ahead search and sort run on a library-owned stream.  The call with
 * it_begin = 0 joins that stream before it touches `work`; after an ERROR return from lk_map_frame / lk_render_bwd the counters may be
 * stale - lk_knn_build (which clears them) before the handle renders a backward again. */
/* Phase-split callers (phases 1, then 2, per iteration): the phase-2 call of a 'color' iteration that is not the call's last leaves the
 * fragment repack to the phase-1 call of the NEXT iteration (it rides in that call's interpolation launch, as in the unsplit loop) - the
 * iterations of an optimize_map call must be issued in order, each phase 1 followed by its phase 2. */
int lk_map_frame(const lk_map_desc* d, int32_t it_begin, int32_t it_end, int32_t phases, void* stream);
/* Makes `stream` wait until the neighbour lists of iteration `it` (work + lk_map_work_nbr_idx) are written: lk_map_frame
 * searches ahead of its loop on a library-owned stream.  Valid after the phase-1 call of iteration it - 1 (or it) of the same
 * optimize_map call has returned. */
int lk_map_wait_lists(const lk_map_desc* d, int32_t it, void* stream);
/* `stream` waits until the feature-row gradients of the LAST phase-1 call issued with lk_map_desc::signal_rows are final.  The event is
 * recorded on that call's launch stream whatever the library's stream mode (with LK_SERIAL / lk_set_serial, or after a failed side-stream
 * creation, too: the waiter always gets a real dependency on the backward); a no-op only if no call has signalled yet. */
int lk_map_wait_rows(const lk_map_desc* d, void* stream);

/* ---------------------------------------------------------------- weight-gradient building block
 * dW[n][k] += sum_rows A'[row][n] * B[row][k], db[n] += sum_rows A'[row][n] (db may be NULL); row-major operands.
 * a_mode 0: A' = A;  1: A' = A * softplus100'(A2) with A2 the activation OUTPUT;  2: A' = A2[row] * A[row>>3]
 * (rel-pos: per-neighbour weight times the per-sample gradient).  This is what lk_render_bwd runs for every
 * decoder matrix (torch: grad of nn.Linear weights, decoder.py:265-288,480-546). */
int lk_wgrad_single(const float* A, int32_t lda, int32_t a_mode, const float* A2, int32_t lda2,
                    const float* B, int32_t ldb, int32_t N, int32_t K, int64_t rows,
                    float* dW, int32_t ldw, float* db, int32_t chunk, void* stream);

/* ---------------------------------------------------------------- measurement
 * Per-kernel GPU time with HIP events recorded on the launch stream around the selected kernels
 * (names: comma-separated, e.g. "k_decode_bwd", or "*").  lk_profile_end synchronises those events and writes
 * "name calls total_ms" lines into buf.  Used by bench.py for the roofline figure.  A name stands for the instantiations rocprofv3 lists
 * under it, with one split: the decoder backward of launches WITH ray gradients (the tracker's: other instantiations, bf16 pieces in the
 * geometry role) is timed as "k_decode_bwd_track", the mapper's as "k_decode_bwd". */
int lk_profile_begin(const char* names);
/* lk_render_bwd runs the decoder weight-gradient reductions on a second, library-owned HIP stream beside the rel-pos backward and
 * the feature scatter (fork / join with events on the caller's stream: the caller sees one ordered stream).  on != 0 keeps
 * everything on the caller's stream - per-kernel durations are then those of a kernel running alone (measurement; the
 * environment variable LK_SERIAL sets the initial state). */
int lk_set_serial(int32_t on);
/* Creates the library's side streams now (they are otherwise created at their first use).  Call right after choosing the device and BEFORE
 * anything that creates many streams of its own (torch.distributed process groups, RCCL): the runtime multiplexes streams onto a few hardware
 * queues in creation order, and a side stream that shares the launch stream's queue serialises the backward's fork. */
int lk_streams_init(void);
int lk_profile_end(char* buf, int cap);
/* Measurement: resident 256-thread workgroups per compute unit of the five MLP kernels, as the runtime computes them from the
 * registers and LDS of the loaded code objects (hipOccupancyMaxActiveBlocksPerMultiprocessor): out[0..4] = k_decode_fwd,
 * k_decode_bwd (mapper form), k_relpos_fwd, k_relpos_bwd_fused, k_wgrad.  Needs a device. */
int lk_debug_occupancy(int32_t out[5]);
/* Test hook: every weight-gradient launch forked onto the library's side stream is preceded there by a kernel that spins for `us`
 * microseconds (0 = off, the default) - a missing ordering between the side stream and the launch stream then fails deterministically
 * instead of by timing (tests/test_split_step_order.py).  Results must not depend on it. */
int lk_debug_side_delay(int32_t us);

#ifdef __cplusplus
}
#endif
#endif /* LOOPY_HIP_H */
