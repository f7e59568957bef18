// Per-frame optimisation loops as single C-ABI calls (include/loopy_hip.h: lk_track_frame, lk_map_frame).
//   reference: Tracker.run loop body + optimize_cam_in_batch (src/Tracker.py:102-197, 313-401),
//              Mapper.optimize_map's joint iterations (src/Mapper.py:576-735).
// The host enqueues a whole frame's launches from C++ (no interpreter between kernels), and the steps that do not depend on
// what the iterations optimise leave the per-iteration sequence:
//   k_pregather        pixel gather + inside mask (+ the mapper's rays) of ALL iterations in one launch
//   k_track_composite  alpha composite + residuals; the tracker loss + composite backward ride in k_decode_bwd   (two many-workgroup
//                      launches around the one global quantity, the batch mean of the residual)
//   k_interp_bwd       also reduces the pose gradient's ray moments per workgroup (lk_bwd2.hip)
//   k_track_final      pose gradient, Adam on the 7 pose parameters, candidate log, rays of the next iteration
// 16 -> 9 launches per tracking iteration, 11 -> 8 per geometry iteration (a launch of a trivial kernel costs ~5 us on the
// GPU's front end however little it does; single-workgroup fusions of the same steps were measured SLOWER: one compute unit
// cannot keep enough scattered loads in flight - DESIGN.md section 7).
#include "lk_common.h"
#include "lk_kernels.h"
#include "lk_composite_dev.h"
#include "lk_mask_dev.h"
#include "lk_track_dev.h"
#include "lk_exposure_dev.h"

#include <math.h>
#include <string.h>

using namespace lkw;

int lk_launch_exposure_step(const lk_exposure_desc& x, int mode, int step, float beta1, float beta2, float eps, hipStream_t st);      // lk_optim.hip
int lk_exposure_step_args(const lk_exposure_desc& x, int mode, int step, float beta1, float beta2, float eps, ExposureStepArgs* out);
int lk_adam_step_x(const lk_adam_seg* segs, int32_t n_seg, float beta1, float beta2, float eps, const ExposureStepArgs* xa, void* stream_);
int lk_launch_composite_loss_exposure(const LkCompositeArgs& ca, const float* gt_color, const int32_t* frame_id, const float* aff, int F, float w_color,
                                      float* d_depth, float* d_logits, float* out_loss, float* g_aff, hipStream_t st);

#define LK_TRACK_FUSED_MAX_R LK_MASK_REG_MAX          // rays a single workgroup keeps in registers (8 per thread)

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ float lp_block_sum_1024(float v, float* sh /*[16]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lk_lane() == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += sh[i];
    return s;
}
// ------------------------------------------------------------------ k_pregather
// Batch assembly of ALL iterations of a frame in one launch (workgroup b = iteration b): pixel gather (get_samples,
// common.py:237-259), inside mask (Tracker.py:153-160 / Mapper.py:674-681: rejected rays become absent, gt_depth = 0) and - for
// the mapper, whose keyframe poses are fixed - the rays (get_rays_from_uv, common.py:104-120).  None of this depends on what the
// iterations optimise, so it leaves the per-iteration launch sequence (two to four launches at the ~5 us floor each).
struct LkPregatherArgs {
    const float* depth; const float* color; const float* c2w; int c2w_stride; const float* r2_map; const int32_t* frame_id;
    const int32_t* rnd;                                   // [iters][R]
    int R, H, W, H0, W0, w;
    float fx, fy, cx, cy;
    float* rays_o; float* rays_d;                         // [iters][R][3], or NULL (tracker: the rays follow the pose)
    float* gt_depth; float* gt_color; float* pix_i; float* pix_j; float* r2_ray; float* thr;       // [iters][R] (x3), thr [iters]
    float* zero4;                                         // [iters][4] rows to clear (loss sums accumulated with atomics), or NULL
    int32_t* frame_out;                                   // [iters][R] keyframe of every ray of the assembled (partitioned) batch, or NULL
    int32_t* n_live;                                      // [iters] or NULL.  With it the batch is PARTITIONED: the rays that keep a depth reading
                                                          // (in draw order) come first, the rejected ones behind them, n_live[it] = their number
};
// One ray of the batch: everything but the depth (which the inside mask decides) written to slot `dst`
__device__ __forceinline__ void pregather_ray(const LkPregatherArgs& a, size_t base, int r, int dst) {
    const int f = a.frame_id ? a.frame_id[r] : 0;
    const int px = a.rnd[base + r];
    const int i = a.W0 + px % a.w, j = a.H0 + px / a.w;
    const size_t pix = ((size_t)f * a.H + j) * a.W + i;
    float* gc = a.gt_color + (base + dst) * 3;
    gc[0] = a.color[3 * pix]; gc[1] = a.color[3 * pix + 1]; gc[2] = a.color[3 * pix + 2];
    if (a.r2_ray) a.r2_ray[base + dst] = a.r2_map ? a.r2_map[pix] : 0.0f;
    if (a.pix_i) { a.pix_i[base + dst] = (float)i; a.pix_j[base + dst] = (float)j; }
    if (a.frame_out) a.frame_out[base + dst] = f;
    if (a.rays_o) {
        const float* M = a.c2w + (size_t)f * a.c2w_stride;             // row-major [3 or 4][4]
        const float d0 = ((float)i - a.cx) / a.fx, d1 = -((float)j - a.cy) / a.fy, d2 = -1.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a.rays_d[(base + dst) * 3 + c] = (d0 * M[4 * c] + d1 * M[4 * c + 1]) + d2 * M[4 * c + 2];
            a.rays_o[(base + dst) * 3 + c] = M[4 * c + 3];
        }
    }
}
template <int VPT>
__global__ __launch_bounds__(1024) void k_pregather(LkPregatherArgs a) {
    __shared__ LkMaskShared S;
    __shared__ int s_wave[VPT][16];                       // kept rays per (ray group q, wave)
    const int t = threadIdx.x, it = blockIdx.x;
    const size_t base = (size_t)it * a.R;
    if (a.zero4 && t < 4) a.zero4[(size_t)it * 4 + t] = 0.0f;
    unsigned u[VPT];
    unsigned mycnt = 0, mymax = 0;
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        const int r = t + 1024 * q;
        u[q] = 0u;
        if (r < a.R) {
            const int f = a.frame_id ? a.frame_id[r] : 0;
            const int px = a.rnd[base + r];
            const int i = a.W0 + px % a.w, j = a.H0 + px / a.w;
            const float d = a.depth[((size_t)f * a.H + j) * a.W + i];
            u[q] = (d > 0.0f) ? __float_as_uint(d) : 0u;
            if (u[q]) { ++mycnt; mymax = max(mymax, u[q]); }
        }
    }
    bool any;
    const float thr = lk_inside_thr<true, VPT>(u, nullptr, a.R, mycnt, mymax, S, &any);
    if (t == 0 && a.thr) a.thr[it] = any ? thr : 0.0f;
    bool keep[VPT];
#pragma unroll
    for (int q = 0; q < VPT; ++q) keep[q] = any && u[q] && __uint_as_float(u[q]) <= thr;
    if (!a.n_live) {                                      // batch in draw order (tracker)
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int r = t + 1024 * q;
            if (r < a.R) {
                pregather_ray(a, base, r, r);
                a.gt_depth[base + r] = keep[q] ? __uint_as_float(u[q]) : 0.0f;
            }
        }
        return;
    }
    // Stable partition, kept rays first: the kernels whose cost is per SAMPLE (rel-pos MLP forward / backward) then work on a prefix
    // and skip the rays without a reading altogether - 2 % of a synthetic frame, 5-15 % of a sensor frame; at the reference's batch of
    // 5 000 rays that is also the difference between 3 and 4 rounds of workgroup tiles.  Ray r = t + 1024 q: order = q-major.
    const int lane = t & 63, wv = t >> 6;
    int pre[VPT];                                         // kept rays before this one inside its (q, wave)
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        const unsigned long long bal = __ballot(keep[q]);
        pre[q] = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[q][wv] = __popcll(bal);
    }
    __syncthreads();
    int before_q = 0, n_keep = 0;                         // kept rays in the (q', wave') pairs before (q, wv); total
    int off[VPT];
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        int mine = 0;
#pragma unroll
        for (int w2 = 0; w2 < 16; ++w2) {
            const int c = s_wave[q][w2];
            if (w2 < wv) mine += c;
            n_keep += c;
        }
        off[q] = before_q + mine;
        before_q = n_keep;
    }
    if (t == 0) a.n_live[it] = n_keep;
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        const int r = t + 1024 * q;
        if (r < a.R) {
            const int kb = off[q] + pre[q];               // kept rays before r
            const int dst = keep[q] ? kb : n_keep + (r - kb);
            pregather_ray(a, base, r, dst);
            a.gt_depth[base + dst] = keep[q] ? __uint_as_float(u[q]) : 0.0f;
        }
    }
}

// ------------------------------------------------------------------ tracker loss in two many-workgroup launches
// pass 1: raw2outputs_nerf_color (common.py:382-422) + the uncertainty-normalised residual of every ray and its block sums
__global__ __launch_bounds__(256) void k_track_composite(LkTrackLossArgs a) {
    __shared__ float sh[2][4];
    const int r = blockIdx.x * 256 + (int)threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 4) a.out4[threadIdx.x] = 0.0f;        // pass 2 accumulates into it
    float tv = 0.0f, cv = 0.0f;
    if (r < a.R) {
        const float gd = a.gt_depth[r];
        const LkRayOut o = lk_composite_ray(a.raw, a.z, a.nbr_count, r, a.S, a.min_nn, a.coef, gd);
        a.depth[r] = o.depth; a.var[r] = o.var;
        a.color[3 * r] = o.c0; a.color[3 * r + 1] = o.c1; a.color[3 * r + 2] = o.c2;
        a.valid_ray[r] = o.valid ? 1 : 0;
        const bool present = gd > 0.0f;                    // absent rays take no part in the mean (filtered before the render)
        tv = present ? fabsf(gd - o.depth) / sqrtf(o.var + 1e-10f) : 0.0f;
        cv = present ? 1.0f : 0.0f;
        a.resid[r] = a.median ? (present ? fabsf(gd - o.depth) : -1.0f) : tv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { tv += __shfl_xor(tv, o); cv += __shfl_xor(cv, o); }
    if (lk_lane() == 0) { sh[0][threadIdx.x >> 6] = tv; sh[1][threadIdx.x >> 6] = cv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.part[2 * blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        a.part[2 * blockIdx.x + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}
// pass 2 - mask by 10 x the batch mean, loss terms (Tracker.py:169-191), d depth / d colour and the composite's backward for them - is the
// prologue of k_decode_bwd (lk_track_draw, lk_track_dev.h); with tracking.handle_dynamic: False the threshold is the median's:
__global__ __launch_bounds__(1024) void k_track_median(LkTrackLossArgs a) {
    __shared__ LkMedianShared S;
    const float thr = lk_block_median10(a.resid, a.R, S);
    if (threadIdx.x == 0) a.part[0] = thr;
}

// loss rows of the iterations [it0, it0 + gridDim.x) of an lk_map_frame call from the per-tile terms k_decode_bwd left (LK_COMPOSITE_IN_BWD):
// one workgroup per iteration, fixed order, no atomics; tiles behind the live prefix of a partitioned batch were not processed
__global__ __launch_bounds__(256) void k_loss_rows_sum(const float* __restrict__ rows, int tiles, int P, int S, const int32_t* __restrict__ n_live, int it0,
                                                        float* __restrict__ log) {
    __shared__ float4 sh[4];
    const int it = it0 + (int)blockIdx.x;
    const int P_live = n_live ? min(P, n_live[it] * S) : P;
    const int nt = (P_live + 31) / 32;
    const float* base = rows + (size_t)it * tiles * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = threadIdx.x; b < nt; b += 256) {
        const float4 q = *reinterpret_cast<const float4*>(base + (size_t)b * 4);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); v.z += __shfl_xor(v.z, o); v.w += __shfl_xor(v.w, o); }
    if (lk_lane() == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = log + (size_t)it * 4;
        o[0] = (sh[0].x + sh[1].x) + (sh[2].x + sh[3].x); o[1] = (sh[0].y + sh[1].y) + (sh[2].y + sh[3].y);
        o[2] = (sh[0].z + sh[1].z) + (sh[2].z + sh[3].z); o[3] = (sh[0].w + sh[1].w) + (sh[2].w + sh[3].w);
    }
}
// position of every optimised row in the row list of an lk_map_frame call (lk_knn_s::row_rank, cleared to -1 before)
__global__ __launch_bounds__(256) void k_row_rank(const int32_t* __restrict__ rows, int n_rows, int N, int32_t* __restrict__ rank) {
    const int i = blockIdx.x * 256 + (int)threadIdx.x;
    if (i < n_rows) { const int r = rows[i]; if (r >= 0 && r < N) rank[r] = i; }
}
// ------------------------------------------------------------------ k_track_final (body: lk_track_dev.h)
// With exposure encoding the launch has a second workgroup: the exposure step of the iteration (backward of the 8 -> 128 -> 12 MLP from the
// per-tile sums of d affine, Adam, forward with the stepped values) - it depends on the decoder backward only, as the pose step does on the
// interpolation backward, and two more launches on the iteration's chain (13 + 5 us) are gone
__global__ __launch_bounds__(1024) void k_track_final(LkTrackFinalArgs a, ExposureStepArgs xa, const float* __restrict__ aff_part, int n_aff_part) {
    if (blockIdx.x == 1) { lk_exposure_step_body(xa, aff_part, n_aff_part); return; }
    lk_track_final_body<16>(a);
}

// ------------------------------------------------------------------ lk_track_frame
namespace {
void adam_scalars(float lr, int step, float beta1, float beta2, float* step_size, float* bc2_sqrt) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    *step_size = (float)((double)lr / bc1);
    *bc2_sqrt = (float)sqrt(bc2);
}
int64_t al4(int64_t x) { return (x + 3) / 4 * 4; }
// pass 1's (sum, count) pairs: one per 256 rays (k_track_composite) or one per tile of whole rays (k_relpos_decode_fwd's epilogue)
static int64_t track_loss_parts(int64_t R, int64_t S) {
    const int64_t s = S < 1 ? 1 : (S > 32 ? 32 : S), ts = (32 / s) * s;
    const int64_t a = (R + 255) / 256, b = (R * S + ts - 1) / ts;
    return a > b ? a : b;
}
struct TrackWork { int64_t gt_depth, gt_color, pix_i, pix_j, r2_ray, thr, resid, loss_part, pose_part, row_part, ring, total; };
TrackWork track_work(int64_t R, int64_t S, int64_t iters) {
    TrackWork w;
    int64_t o = 0;
    w.gt_depth = o; o += al4(iters * R);
    w.gt_color = o; o += al4(iters * R * 3);
    w.pix_i = o; o += al4(iters * R);
    w.pix_j = o; o += al4(iters * R);
    w.r2_ray = o; o += al4(iters * R);
    w.thr = o; o += al4(iters);
    w.resid = o; o += al4(R);
    w.loss_part = o; o += al4(2 * track_loss_parts(R, S));
    w.pose_part = o; o += al4(12 * (int64_t)lk_bwd_pose_parts(R * S));
    w.row_part = o; o += 4 * (int64_t)lk_bwd_pose_parts(R * S);      // LkTrackLossArgs::row_part
    w.ring = o; o += 48;              // two poses [2][8] and their Adam moments [2][16]: the pose step as the prologue of the next search launch
    w.total = o;
    return w;
}
struct MapWork { int64_t rays_o, rays_d, gt_depth, gt_color, r2_ray, thr, n_live, frame_id, z, nbr_idx, nbr_w, nbr_count, seg_list, seg_total, seg_rank, w_next, loss_rows, c_col_alt, total; };
MapWork map_work(int64_t R, int64_t S, int64_t iters) {
    MapWork w;
    int64_t o = 0;
    w.rays_o = o; o += al4(iters * R * 3);
    w.rays_d = o; o += al4(iters * R * 3);
    w.gt_depth = o; o += al4(iters * R);
    w.gt_color = o; o += al4(iters * R * 3);
    w.r2_ray = o; o += al4(iters * R);
    w.thr = o; o += al4(iters);
    w.n_live = o; o += al4(iters);                       // int32: rays of the iteration that keep a depth reading (they come first)
    w.frame_id = o; o += al4(iters * R);                 // int32: keyframe of every ray AFTER the partition (exposure encoding: affine per keyframe)
    // the search results of every iteration (lk_presample on the third stream): [iters][P], [iters][P][8]
    const int64_t P = R * S;
    w.z = o; o += al4(iters * P);
    w.nbr_idx = o; o += al4(iters * P * LK_K);
    w.nbr_w = o; o += al4(iters * P * LK_K);
    w.nbr_count = o; o += al4(iters * P);
    // the rows of every iteration sorted by point (feature-gradient gather), also ahead of the loop: [iters][8P] + [iters] + scratch [8P]
    w.seg_list = o; o += al4(iters * P * LK_K);
    w.seg_total = o; o += al4(iters);
    w.seg_rank = o; o += al4(P * LK_K * LK_SEG_BATCH);        // the rows of up to LK_SEG_BATCH iterations are sorted per launch
    w.w_next = o; o += al4(lk_weight_blob_floats());          // the decoder blob as stepped by the step rider (LkStepRider::w_next)
    w.loss_rows = o; o += al4(iters * 4 * ((P + 31) / 32));   // the loss row's terms per decoder-backward tile (LK_COMPOSITE_IN_BWD): [iters][tiles][4]
    // second interpolated-colour-feature buffer [P][32]: odd iterations of the call write this one, even ones lk_render_desc::c_col.  The
    // weight-gradient launch of a SPLIT step (LkBwdExtra::split_reduce) streams c_col as the auxiliary columns of its fc_c jobs on the side
    // stream while the launch stream is already in the next iteration, whose interpolation / rel-pos MLP writes c_col BEFORE the join in
    // front of its decoder launch - with one buffer that overwrite raced with k_wgrad (round-5 advisor: the TUM budget, no rel-pos MLP,
    // rewrites it a few microseconds into the next iteration).  With two, iteration it + 2 is the next writer of iteration it's buffer, and
    // it starts behind the join of iteration it + 1, which is behind iteration it's k_wgrad on the side stream.
    w.c_col_alt = o; o += al4(P * LK_C);
    w.total = o;
    return w;
}
// Third stream (lk_aux_stream, lk_api.hip): the neighbour search of a mapping call's iterations runs ahead of the iterations
// themselves (it reads the rays and the positions, nothing the iterations write), in up to LK_PRE_CHUNKS launches with an event each.
typedef LkAuxStream PreStream;
int map_pre_chunk(int iters) { const int c = lk_cdiv(iters, LK_PRE_CHUNKS - 1); return c < 6 ? 6 : c; }
PreStream& pre_stream() { return lk_aux_stream(); }
}  // namespace

extern "C" int64_t lk_track_work_floats(int32_t R, int32_t S, int32_t iters) { return track_work(R, S, iters).total; }
extern "C" int64_t lk_map_work_floats(int32_t R, int32_t S, int32_t iters) { return map_work(R, S, iters).total; }
extern "C" int64_t lk_map_work_nbr_idx(int32_t R, int32_t S, int32_t iters) { return map_work(R, S, iters).nbr_idx; }

extern "C" int lk_track_frame(const lk_track_desc* d, void* stream_) {
    LK_REQUIRE(d != nullptr, "lk_track_frame: NULL descriptor");
    { const int rcg = lk_status_gate("lk_track_frame"); if (rcg != LK_OK) return rcg; }
    LK_REQUIRE(d->iters >= 0 && d->render.R >= 0 && d->w > 0, "lk_track_frame: bad sizes");
    if (d->iters == 0 || d->render.R == 0) return LK_OK;
    LK_REQUIRE(d->depth_img && d->color_img && d->rnd && d->cam7 && d->g_cam7 && d->adam_mv && d->hist && d->log, "lk_track_frame: NULL buffer");
    LK_REQUIRE(d->render.d_depth && d->render.d_color && d->render.bwd_scratch && d->render.act,
               "lk_track_frame: the render descriptor needs d_depth/d_color, bwd_scratch and act");
    hipStream_t st = (hipStream_t)stream_;
    lk_render_desc rd = d->render;
    rd.flags = (d->render.flags & (LK_FLAG_REL_POS | LK_FLAG_FEATS_F16)) | LK_FLAG_STAGE_COLOR | LK_FLAG_TRACKER | LK_FLAG_SAVE_ACT | LK_FLAG_GRAD_RAYS | LK_FLAG_ZERO_ABSENT;
    // the tracker's colour loss gradient is w_color sgn(.) (Tracker.py:183-191): unit scale, so the colour decoder's backward may run on
    // pre-scaled fp16 pieces (LK_FLAG_UNIT_LOSS_GRADS is added below, once `fused` is known)
    rd.stats_chunk = rd.R > 0 ? rd.R : 1;
    rd.g_geo_feats = nullptr; rd.g_col_feats = nullptr; rd.g_weights = nullptr;
    // exposure encoding (decoder.py:534-540, Tracker.py:329-344): the frame's affine inside the colour decoder, its gradient from the
    // decoder backward, feature + MLP stepped once per iteration
    const lk_exposure_desc* xd = d->exposure;
    LK_REQUIRE(!xd || xd->F == 1, "lk_track_frame: the tracker has ONE exposure feature (this frame's)");
    if (xd) { rd.affine = xd->aff; rd.g_affine = xd->g_aff; }
    const int R = rd.R, S = rd.S, iters = d->iters;
    const bool fused = R <= LK_TRACK_FUSED_MAX_R && d->work != nullptr;
    // the pose step of an iteration as the prologue of the NEXT iteration's search launch (with exposure encoding the step keeps its own launch:
    // the exposure workgroup rides in it)
    // (round 4: with exposure encoding too - the exposure step then rides as the last workgroup of the interpolation backward's launch;
    // only exposure encoding WITH the rel-pos MLP - no shipped config - keeps the step's own launch k_track_final with the exposure workgroup in it)
    const bool x_in_bwd = fused && d->exposure != nullptr && !(rd.flags & LK_FLAG_REL_POS);
    const bool prologue = fused && (d->exposure == nullptr || x_in_bwd);
    // (with exposure encoding d out passes through the learned affine first: unit scale only relative to xd->bwd_scale, which the fused
    // sequence hands to the kernels)
    if ((xd == nullptr || (xd->bwd_scale && fused)) && fabsf(d->w_color) <= 4.0f) rd.flags |= LK_FLAG_UNIT_LOSS_GRADS;
    if (!fused) LK_REQUIRE(d->gt_color && d->pix_i && d->pix_j && d->thr && d->scratch_u32 && d->loss_scratch && d->render.g_rays_o && d->render.g_rays_d,
                           "lk_track_frame: without `work` (or above 8192 rays) the per-iteration batch buffers and g_rays_o/g_rays_d are needed");
    const LkBwdOffsets off = lk_bwd_offsets((int64_t)R * S, rd.flags);
    const float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;
    LK_HIP_TRY(hipMemsetAsync(d->adam_mv, 0, 14 * sizeof(float), st));
    const TrackWork wk = track_work(R, S, iters);
    float* W0 = d->work;
    const int n_pose = lk_bwd_pose_parts((int64_t)R * S), n_lp = lk_cdiv(R, 256);
    // (fused: the loss terms and the composite backward are the prologue of the decoder backward, lk_track_draw)
    LkTrackFinalArgs fa;
    memset(&fa, 0, sizeof(fa));
    if (fused) {
        // every iteration's pixels, colours, radii and inside mask in one launch; then the rays of the initial pose
        LkPregatherArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.depth = d->depth_img; pa.color = d->color_img; pa.r2_map = d->r2_map; pa.rnd = d->rnd;
        pa.R = R; pa.H = d->H; pa.W = d->W; pa.H0 = d->H0; pa.W0 = d->W0; pa.w = d->w;
        pa.fx = d->fx; pa.fy = d->fy; pa.cx = d->cx; pa.cy = d->cy;
        pa.gt_depth = W0 + wk.gt_depth; pa.gt_color = W0 + wk.gt_color; pa.pix_i = W0 + wk.pix_i; pa.pix_j = W0 + wk.pix_j;
        pa.r2_ray = rd.r2_ray ? W0 + wk.r2_ray : nullptr; pa.thr = W0 + wk.thr;
        hipLaunchKernelGGL(k_pregather<LK_MASK_VPT>, dim3(iters), dim3(1024), 0, st, pa);
        fa.R = R; fa.n_part = n_pose; fa.fx = d->fx; fa.fy = d->fy; fa.cx = d->cx; fa.cy = d->cy;
        fa.pose_part = W0 + wk.pose_part; fa.cam = d->cam7; fa.g_cam = d->g_cam7; fa.adam_mv = d->adam_mv;
        fa.beta1 = beta1; fa.beta2 = beta2; fa.eps = eps;
        fa.rays_o = const_cast<float*>(rd.rays_o); fa.rays_d = const_cast<float*>(rd.rays_d);
        fa.next_pix_i = W0 + wk.pix_i; fa.next_pix_j = W0 + wk.pix_j; fa.do_update = 0;
        ExposureStepArgs xa;
        memset(&xa, 0, sizeof(xa));
        if (xd) {       // the exposure forward of the first iteration rides as the launch's second workgroup
            const int rc = lk_exposure_step_args(*xd, 2, 1, beta1, beta2, eps, &xa);
            if (rc != LK_OK) return rc;
        }
        hipLaunchKernelGGL(k_track_final, dim3(xd ? 2 : 1), dim3(1024), 0, st, fa, xa, (const float*)nullptr, 0);
        if (prologue) {      // pose ring: slot 0 = the initial pose with zero moments
            LK_HIP_TRY(hipMemcpyAsync(W0 + wk.ring, d->cam7, 7 * sizeof(float), hipMemcpyDeviceToDevice, st));
            LK_HIP_TRY(hipMemsetAsync(W0 + wk.ring + 16, 0, 32 * sizeof(float), st));
        }
    } else if (xd) {
        const int rc = lk_launch_exposure_step(*xd, 2, 1, beta1, beta2, eps, st);
        if (rc != LK_OK) return rc;
    }
    for (int it = 0; it < iters; ++it) {
        const int32_t* rnd = d->rnd + (size_t)it * R;
        float* hist_row = d->hist + (size_t)it * 7;
        float* log_row = d->log + (size_t)it * 4;
        float* gt_color = d->gt_color;
        if (fused) {
            rd.gt_depth = W0 + wk.gt_depth + (size_t)it * R;
            if (rd.r2_ray) rd.r2_ray = W0 + wk.r2_ray + (size_t)it * R;
            gt_color = W0 + wk.gt_color + (size_t)it * R * 3;
        } else {
            if (!d->hist_post) LK_HIP_TRY(hipMemcpyAsync(hist_row, d->cam7, 7 * sizeof(float), hipMemcpyDeviceToDevice, st));
            // the image gathers do not depend on the pose (the pose slot only feeds rays that are recomputed right after)
            int rc = lk_gather_rays(d->depth_img, d->color_img, d->cam7, 0, d->r2_map, nullptr, rnd, R, d->H, d->W, d->H0, d->W0, d->w,
                                    d->fx, d->fy, d->cx, d->cy, const_cast<float*>(rd.rays_o), const_cast<float*>(rd.rays_d),
                                    const_cast<float*>(rd.gt_depth), d->gt_color, d->pix_i, d->pix_j, const_cast<float*>(rd.r2_ray), st);
            if (rc != LK_OK) return rc;
            rc = lk_rays_from_pose(d->cam7, d->pix_i, d->pix_j, R, d->fx, d->fy, d->cx, d->cy, const_cast<float*>(rd.rays_o), const_cast<float*>(rd.rays_d), st);
            if (rc != LK_OK) return rc;
            rc = lk_inside_mask(rd.gt_depth, R, nullptr, const_cast<float*>(rd.gt_depth), d->thr, d->scratch_u32, st);
            if (rc != LK_OK) return rc;
        }
        // ---- forward, loss, backward
        // the pose step of the iteration before as the prologue of this iteration's search (k_sample_interp_pose): pose it - 1 -> it
        LkTrackFinalArgs pa = fa;
        if (prologue && it > 0) {
            float sT, sq, b2;
            adam_scalars(d->lr_T, it, beta1, beta2, &sT, &b2);
            adam_scalars(d->lr_q, it, beta1, beta2, &sq, &b2);
            float* hrow = d->hist + (size_t)(it - 1) * 7;
            pa.step_T = sT; pa.step_q = sq; pa.bc2_sqrt = b2; pa.do_update = 1;
            pa.hist_pre = d->hist_post ? nullptr : hrow; pa.hist_post = d->hist_post ? hrow : nullptr;
            pa.cam_in = W0 + wk.ring + 8 * ((it - 1) & 1); pa.cam = W0 + wk.ring + 8 * (it & 1);
            pa.mv_in = W0 + wk.ring + 16 + 16 * ((it - 1) & 1); pa.adam_mv = W0 + wk.ring + 16 + 16 * (it & 1);
            pa.rays_o = const_cast<float*>(rd.rays_o); pa.rays_d = const_cast<float*>(rd.rays_d);
            pa.next_pix_i = W0 + wk.pix_i + (size_t)it * R; pa.next_pix_j = W0 + wk.pix_j + (size_t)it * R;
            pa.loss_part = W0 + wk.row_part; pa.n_loss_part = n_pose; pa.log_row = d->log + (size_t)(it - 1) * 4;
        }
        LkTrackLossArgs la;
        memset(&la, 0, sizeof(la));
        if (fused) {
            la.R = R; la.S = S; la.min_nn = rd.min_nn; la.coef = rd.coef; la.w_color = d->w_color;
            la.use_color = d->use_color & LK_TRACK_USE_COLOR; la.median = (d->use_color & LK_TRACK_MEDIAN_MASK) ? 1 : 0;
            la.raw = rd.raw; la.z = rd.z; la.nbr_count = rd.nbr_count; la.gt_depth = rd.gt_depth; la.gt_color = gt_color;
            la.depth = rd.depth; la.var = rd.var; la.color = rd.color; la.valid_ray = rd.valid_ray;
            la.d_depth = const_cast<float*>(rd.d_depth); la.d_color = const_cast<float*>(rd.d_color);
            la.d_raw = rd.bwd_scratch + off.d_raw; la.out4 = log_row; la.resid = W0 + wk.resid; la.part = W0 + wk.loss_part;
            la.row_part = W0 + wk.row_part;
        }
        // (pass 1 of the loss rides in the forward's launch where it can: comp_tiles = its number of partial pairs)
        int comp_tiles = 0;
        int rc = lk_render_fwd_impl(&rd, st, fused ? (LK_SKIP_COMPOSITE | LK_FUSE_SMALL) : 0, nullptr, nullptr, (prologue && it > 0) ? &pa : nullptr,
                                    fused ? &la : nullptr, &comp_tiles);
        if (rc != LK_OK) return rc;
        const int n_part = comp_tiles > 0 ? comp_tiles : n_lp;
        if (fused) {
            if (comp_tiles == 0) hipLaunchKernelGGL(k_track_composite, dim3(n_lp), dim3(256), 0, st, la);
            if (la.median) hipLaunchKernelGGL(k_track_median, dim3(1), dim3(1024), 0, st, la);
        } else {
            rc = lk_loss_tracker(R, rd.depth, rd.var, rd.color, rd.gt_depth, d->gt_color, d->w_color, d->use_color,
                                 const_cast<float*>(rd.d_depth), const_cast<float*>(rd.d_color), log_row, d->loss_scratch, st);
            if (rc != LK_OK) return rc;
        }
        LkBwdExtra ex;
        memset(&ex, 0, sizeof(ex));
        if (fused) { ex.track_loss = &la; ex.track_n_part = n_part; }
        ex.pose_part = fused ? W0 + wk.pose_part : nullptr;
        ex.dscale = (xd && (rd.flags & LK_FLAG_UNIT_LOSS_GRADS)) ? xd->bwd_scale : nullptr;
        ex.pix_i = W0 ? W0 + wk.pix_i + (size_t)it * R : nullptr; ex.pix_j = W0 ? W0 + wk.pix_j + (size_t)it * R : nullptr;
        ex.fx = d->fx; ex.fy = d->fy; ex.cx = d->cx; ex.cy = d->cy;
        ExposureStepArgs xa_bwd;
        if (x_in_bwd) {
            rc = lk_exposure_step_args(*xd, 3, it + 1, beta1, beta2, eps, &xa_bwd);
            if (rc != LK_OK) return rc;
            ex.xstep = &xa_bwd; ex.xstep_part = rd.bwd_scratch + off.aff_part; ex.xstep_n_part = lk_cdiv(R * S, 32);
        }
        rc = lk_render_bwd_impl(&rd, st, fused ? (LK_SKIP_COMPOSITE_BWD | LK_SKIP_RAYS_BWD | LK_FUSE_SMALL | (xd ? LK_SKIP_AFF_REDUCE : 0)) : 0, fused ? &ex : nullptr);
        if (rc != LK_OK) return rc;
        if (xd && !fused) {
            rc = lk_launch_exposure_step(*xd, 3, it + 1, beta1, beta2, eps, st);
            if (rc != LK_OK) return rc;
        }
        // ---- pose gradient + Adam (Tracker.py:317-352: group T at cam_lr, group q at 0.2 cam_lr when separate_LR)
        // (letting the workgroup of k_interp_bwd that finishes last run this step saved the launch and cost more: every workgroup
        // pays a device-scope release fence for the hand-over, 139 -> 151 us per iteration)
        float step_T, step_q, bc2s;
        adam_scalars(d->lr_T, it + 1, beta1, beta2, &step_T, &bc2s);
        adam_scalars(d->lr_q, it + 1, beta1, beta2, &step_q, &bc2s);
        if (fused && prologue && it + 1 < iters) {
            // (the step of this iteration runs inside the next iteration's search launch)
        } else if (fused) {
            if (prologue) {      // the last step: from the ring into the caller's pose and moments
                fa.cam_in = W0 + wk.ring + 8 * (it & 1); fa.mv_in = W0 + wk.ring + 16 + 16 * (it & 1);
            }
            fa.step_T = step_T; fa.step_q = step_q; fa.bc2_sqrt = bc2s; fa.do_update = 1;
            fa.loss_part = W0 + wk.row_part; fa.n_loss_part = n_pose; fa.log_row = log_row;
            fa.hist_pre = d->hist_post ? nullptr : hist_row; fa.hist_post = d->hist_post ? hist_row : nullptr;
            const bool more = it + 1 < iters;
            fa.rays_o = more ? const_cast<float*>(rd.rays_o) : nullptr; fa.rays_d = const_cast<float*>(rd.rays_d);
            fa.next_pix_i = W0 + wk.pix_i + (size_t)(it + 1) * R * (more ? 1 : 0); fa.next_pix_j = W0 + wk.pix_j + (size_t)(it + 1) * R * (more ? 1 : 0);
            ExposureStepArgs xa;
            memset(&xa, 0, sizeof(xa));
            const bool x_here = xd != nullptr && !x_in_bwd;       // (else the exposure step of this iteration rode in the backward above)
            if (x_here) {
                rc = lk_exposure_step_args(*xd, 3, it + 1, beta1, beta2, eps, &xa);
                if (rc != LK_OK) return rc;
            }
            hipLaunchKernelGGL(k_track_final, dim3(x_here ? 2 : 1), dim3(1024), 0, st, fa, xa,
                               x_here ? (const float*)(rd.bwd_scratch + off.aff_part) : (const float*)nullptr, x_here ? lk_cdiv(R * S, 32) : 0);
        }
        if (!fused) {
            rc = lk_pose_bwd(d->cam7, d->pix_i, d->pix_j, R, d->fx, d->fy, d->cx, d->cy, rd.g_rays_o, rd.g_rays_d, d->g_cam7, st);
            if (rc != LK_OK) return rc;
            lk_adam_seg seg[2];
            memset(seg, 0, sizeof(seg));
            seg[0].p = d->cam7 + 4; seg[0].g = d->g_cam7 + 4; seg[0].m = d->adam_mv + 4; seg[0].v = d->adam_mv + 11; seg[0].n = 3; seg[0].lr = d->lr_T; seg[0].step = it + 1;
            seg[1].p = d->cam7; seg[1].g = d->g_cam7; seg[1].m = d->adam_mv; seg[1].v = d->adam_mv + 7; seg[1].n = 4; seg[1].lr = d->lr_q; seg[1].step = it + 1;
            rc = lk_adam_step(seg, 2, beta1, beta2, eps, st);
            if (rc != LK_OK) return rc;
            if (d->hist_post) LK_HIP_TRY(hipMemcpyAsync(hist_row, d->cam7, 7 * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
    }
    LK_LAUNCH_CHECK();
    return LK_OK;
}

// ------------------------------------------------------------------ lk_map_frame
// Start of an optimize_map call (or of a segment of one): join the look-ahead stream, then assemble the batches of every iteration.
// Neither step reads the optimised rows, the map or the decoders - lk_map_prepare can run it before the caller knows its row list.
static void map_join_and_pregather(const lk_map_desc* d, const MapWork& wk, int R, const lk_exposure_desc* xd, hipStream_t st) {
    float* W0 = d->work;
    {
        // A call that starts an optimize_map call joins the look-ahead stream first: an earlier call that stopped before its last enqueued
        // chunk was consumed (an error between the phases of a data-parallel caller, a range that ends early) may still be searching and
        // sorting there - reading the `work` buffer this call's k_pregather overwrites and holding the index's row counters mid-sort.
        // With an idle look-ahead stream the wait is satisfied at once.
        PreStream& pj = pre_stream();
        if (pj.ok) { (void)hipEventRecord(pj.e1, pj.st); (void)hipStreamWaitEvent(st, pj.e1, 0); }
    }
    {
        // pixels, rays, colours, radii and the inside mask of EVERY iteration of this optimize_map call in one launch; it also
        // clears the loss rows the composite kernels accumulate into
        LkPregatherArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.depth = d->depth_stack; pa.color = d->color_stack; pa.c2w = d->c2w_stack; pa.c2w_stride = d->c2w_stride; pa.r2_map = d->r2_map_stack;
        pa.frame_id = d->frame_id; pa.rnd = d->rnd;
        pa.R = R; pa.H = d->H; pa.W = d->W; pa.H0 = d->H0; pa.W0 = d->W0; pa.w = d->w;
        pa.fx = d->fx; pa.fy = d->fy; pa.cx = d->cx; pa.cy = d->cy;
        pa.rays_o = W0 + wk.rays_o; pa.rays_d = W0 + wk.rays_d; pa.gt_depth = W0 + wk.gt_depth; pa.gt_color = W0 + wk.gt_color;
        pa.r2_ray = d->render.r2_ray ? W0 + wk.r2_ray : nullptr; pa.thr = W0 + wk.thr; pa.zero4 = d->log;
        pa.n_live = reinterpret_cast<int32_t*>(W0 + wk.n_live);
        if (xd) pa.frame_out = reinterpret_cast<int32_t*>(W0 + wk.frame_id);
        if (R <= LK_MASK_REG_MAX) hipLaunchKernelGGL(k_pregather<LK_MASK_VPT>, dim3(d->iters), dim3(1024), 0, st, pa);
        else hipLaunchKernelGGL(k_pregather<LK_LOOP_MAX_R / 1024>, dim3(d->iters), dim3(1024), 0, st, pa);
    }
}
// The batch assembly of an optimize_map call (or of its first segment) ahead of the call itself: d needs the batch inputs (stacks, frame_id,
// rnd, window, intrinsics), render.R / S / r2_ray, work, iters, log and exposure - NOT rows, the optimiser buffers or the gradient buffers.
// The lk_map_frame call that follows (same descriptor values, batches_ready = 1, it_begin = 0) skips its own assembly.
extern "C" int lk_map_prepare(const lk_map_desc* d, void* stream_) {
    LK_REQUIRE(d != nullptr && d->iters >= 0 && d->render.R >= 0, "lk_map_prepare: bad descriptor");
    if (d->iters == 0 || d->render.R == 0) return LK_OK;
    LK_REQUIRE(d->work != nullptr && d->render.R <= LK_LOOP_MAX_R, "lk_map_prepare: needs `work` and at most 16384 rays");
    LK_REQUIRE(d->depth_stack && d->color_stack && d->c2w_stack && d->rnd && d->log, "lk_map_prepare: NULL batch buffer");
    LK_REQUIRE(!d->exposure || d->frame_id, "lk_map_prepare: exposure encoding needs frame_id");
    const MapWork wk = map_work(d->render.R, d->render.S, d->iters);
    map_join_and_pregather(d, wk, d->render.R, d->exposure, (hipStream_t)stream_);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

extern "C" int lk_map_frame(const lk_map_desc* d, int32_t it_begin, int32_t it_end, int32_t phases, void* stream_) {
    LK_REQUIRE(d != nullptr, "lk_map_frame: NULL descriptor");
    { const int rcg = lk_status_gate("lk_map_frame"); if (rcg != LK_OK) return rcg; }
    LK_REQUIRE(it_begin >= 0 && it_end <= d->iters && it_begin <= it_end && (phases & 3), "lk_map_frame: bad iteration range / phases");
    LK_REQUIRE(d->n_geo_dec >= 0 && d->n_geo_dec <= LK_MAX_SPANS && d->n_col_dec >= 0 && d->n_col_dec <= LK_MAX_SPANS, "lk_map_frame: too many spans");
    if (it_begin == it_end || d->render.R == 0) return LK_OK;
    // a segment of a longer optimize_map call (lk_map_desc::it_offset): local iteration `it` is the call's iteration goff + it; the first
    // 'color' iteration has the local index n_geo_l (<= 0: the whole segment is 'color')
    LK_REQUIRE(d->it_offset >= 0, "lk_map_frame: negative it_offset");
    const int goff = d->it_offset, n_geo_l = d->n_geo_iters - goff;
    LK_REQUIRE(d->depth_stack && d->color_stack && d->c2w_stack && d->rnd && d->log, "lk_map_frame: NULL batch buffer");
    LK_REQUIRE(d->weights_rw && d->weights_frag_rw && d->geo_feats_rw && d->col_feats_rw && d->adam_rows && d->adam_dec && d->n_rows >= 0,
               "lk_map_frame: NULL optimiser buffer");
    LK_REQUIRE(d->render.g_geo_feats && d->render.g_col_feats && d->render.g_weights && d->render.d_depth && d->render.d_color &&
               d->render.bwd_scratch && d->render.act, "lk_map_frame: the render descriptor needs gradient buffers, d_depth/d_color, bwd_scratch and act");
    hipStream_t st = (hipStream_t)stream_;
    const int R = d->render.R;
    const bool pre = d->work != nullptr && R <= LK_LOOP_MAX_R;
    if (!pre) LK_REQUIRE(d->gt_color && d->thr && d->scratch_u32, "lk_map_frame: without `work` (or above 16384 rays) the per-iteration batch buffers are needed");
    // exposure encoding (Mapper.py:588-607, 697-715): the 'color' iterations render logits, the loss applies the keyframe's affine
    const lk_exposure_desc* xd = d->exposure;
    LK_REQUIRE(!xd || (pre && d->frame_id), "lk_map_frame: exposure encoding needs `work` (<= 16384 rays) and frame_id");
    const float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;
    const int64_t nb = lk_weight_blob_floats();
    const int64_t nrow = d->n_rows * LK_C;
    // Whole-map refinement (rows = NULL: every row of the tables is a parameter, Mapper.py:884-897): Adam only visits the rows that have
    // received a gradient since the call's first iteration - the gather flags them (lk_knn_s::act_flag), the step reads the flags.  A row
    // that never got a gradient has zero moments, and Adam leaves such a row bit for bit where it is: the result is the dense step's, at
    // 5 M points for 5 MB of flags + the touched rows instead of 10 GB per iteration.  (Between the phases of a data-parallel caller only
    // if it adds the rows the OTHER ranks touched, lk_knn_flag_rows / union_rows_flagged: they receive gradient through the exchange.)
    lk_knn_s* kn_h = d->render.knn;
    const bool act_rows = d->rows == nullptr && ((phases & 3) == 3 || d->union_rows_flagged) && kn_h->act_flag != nullptr && d->n_rows <= kn_h->capacity;
    if (act_rows && it_begin == 0 && goff == 0 && (phases & 1)) LK_HIP_TRY(hipMemsetAsync(kn_h->act_flag, 0, (size_t)d->n_rows, st));
    const MapWork wk = map_work(R, d->render.S, d->iters);
    const int64_t Pn = (int64_t)R * d->render.S;
    // iterations per chunk of the work that runs ahead; chunk 0 is the first iteration alone (the loop waits for it), chunk c >= 1
    // starts at iteration 1 + (c - 1) pre_chunk
    const int pre_chunk = map_pre_chunk(d->iters);
    auto chunk_start = [&](int c) { return c == 0 ? 0 : 1 + (c - 1) * pre_chunk; };
    auto chunk_of = [&](int it) { return it == 0 ? 0 : 1 + (it - 1) / pre_chunk; };
    float* W0 = d->work;
    if (pre && it_begin == 0 && (phases & 1) && !d->batches_ready) map_join_and_pregather(d, wk, R, xd, st);
    // Ahead of the loop, on the third stream, a few iterations per chunk: the neighbour search (one launch per chunk) and,
    // per iteration, the counting sort of its rows by point for the feature-gradient gather (count, scan, place: lk_bwd2.hip) - both read
    // the rays, the positions and the row mask only.  Chunk c + 1 is enqueued when the loop reaches chunk c.
    const bool sort_ahead = pre && d->render.g_geo_feats != nullptr;
    // with a row list the sort is keyed by the row's POSITION in that list (lk_knn_s::row_rank): the counters and the scans of an iteration
    // cover n_rows keys instead of all N points (5 M points: three scans of 40 M counters per chunk, 0.7 ms of HBM-bound launches)
    const bool key_rows = sort_ahead && d->rows != nullptr && kn_h->row_rank != nullptr && d->n_rows > 0;
    auto enqueue_chunk = [&](int c) -> int {
        const int c0 = chunk_start(c);
        if (c0 >= d->iters) return LK_OK;
        PreStream& ps = pre_stream();
        hipStream_t pst = ps.ok ? ps.st : st;
        if (ps.ok && c == 0) { (void)hipEventRecord(ps.e0, st); (void)hipStreamWaitEvent(pst, ps.e0, 0); }     // after k_pregather
        const int c1 = chunk_start(c + 1) < d->iters ? chunk_start(c + 1) : d->iters, nc = c1 - c0;
        lk_render_desc sd = d->render;
        sd.flags = (d->render.flags & LK_FLAG_REL_POS) | LK_FLAG_ZERO_ABSENT;
        sd.R = nc * R; sd.stats_chunk = sd.R;
        sd.rays_o = W0 + wk.rays_o + (size_t)c0 * R * 3; sd.rays_d = W0 + wk.rays_d + (size_t)c0 * R * 3;
        sd.gt_depth = W0 + wk.gt_depth + (size_t)c0 * R;
        sd.r2_ray = d->render.r2_ray ? W0 + wk.r2_ray + (size_t)c0 * R : nullptr;
        sd.z = W0 + wk.z + (size_t)c0 * Pn; sd.nbr_count = reinterpret_cast<int32_t*>(W0 + wk.nbr_count) + (size_t)c0 * Pn;
        sd.nbr_idx = reinterpret_cast<int32_t*>(W0 + wk.nbr_idx) + (size_t)c0 * Pn * LK_K; sd.nbr_w = W0 + wk.nbr_w + (size_t)c0 * Pn * LK_K;
        const bool counted = sort_ahead && nc <= LK_SEG_BATCH;       // the search counts the rows per point of its iterations on the way
        LkPresampleCount pc;
        pc.P_iter = (int)Pn; pc.seg_rank = reinterpret_cast<int32_t*>(W0 + wk.seg_rank); pc.live_rays = reinterpret_cast<const int32_t*>(W0 + wk.n_live) + c0;
        pc.key_of = key_rows ? kn_h->row_rank : nullptr;
        sd.grad_row_mask = d->render.grad_row_mask;
        int rc = lk_presample(&sd, pst, counted ? &pc : nullptr);
        if (rc != LK_OK) return rc;
        for (int it = c0; sort_ahead && it < c0 + nc; it += LK_SEG_BATCH) {       // batches of consecutive iterations: five launches each
            const int nbatch = c0 + nc - it < LK_SEG_BATCH ? c0 + nc - it : LK_SEG_BATCH;
            const lk_knn_s* kn = d->render.knn;
            LkFeatScatterArgs fs;
            memset(&fs, 0, sizeof(fs));
            fs.P = (int)Pn; fs.min_nn = d->render.min_nn; fs.row_mask = d->render.grad_row_mask; fs.N = key_rows ? (int)d->n_rows : (int)kn->n;
            fs.key_of = key_rows ? kn_h->row_rank : nullptr;
            fs.nbr_idx = reinterpret_cast<int32_t*>(W0 + wk.nbr_idx) + (size_t)it * Pn * LK_K; fs.nbr_w = W0 + wk.nbr_w + (size_t)it * Pn * LK_K;
            fs.nbr_count = reinterpret_cast<int32_t*>(W0 + wk.nbr_count) + (size_t)it * Pn;
            fs.seg_cnt = kn->seg_cnt; fs.seg_off = kn->seg_off; fs.seg_sums = kn->seg_sums;
            fs.cnt_stride = kn->seg_stride; fs.sums_stride = kn->seg_sums_stride;
            fs.seg_rank = reinterpret_cast<int32_t*>(W0 + wk.seg_rank);
            fs.seg_list = reinterpret_cast<int32_t*>(W0 + wk.seg_list) + (size_t)it * Pn * LK_K;
            fs.seg_total = reinterpret_cast<int32_t*>(W0 + wk.seg_total) + it;
            fs.live_rays = reinterpret_cast<const int32_t*>(W0 + wk.n_live) + it; fs.S = d->render.S;
            rc = lk_launch_seg_sort(fs, counted, pst, nbatch);
            if (rc != LK_OK) return rc;
        }
        if (ps.ok) (void)hipEventRecord(ps.ev[c % LK_PRE_CHUNKS], pst);
        return LK_OK;
    };
    if (key_rows && it_begin == 0 && (phases & 1)) {
        LK_HIP_TRY(hipMemsetAsync(kn_h->row_rank, 0xFF, sizeof(int32_t) * (size_t)kn_h->n, st));        // -1: not optimised
        hipLaunchKernelGGL(k_row_rank, dim3(lk_cdiv(d->n_rows, 256)), dim3(256), 0, st, d->rows, (int)d->n_rows, (int)kn_h->n, kn_h->row_rank);
    }
    if (pre && it_begin == 0 && (phases & 1)) {
        const int rc = enqueue_chunk(0);
        if (rc != LK_OK) return rc;
    }
    // the optimiser segments of iteration `it` (torch.optim.Adam: parameters without a gradient are skipped and keep their own step
    // count - the colour decoder and the colour rows first step in the first 'color' iteration, Mapper.py:588-607, 722-724)
    auto build_segs = [&](int it, bool color, lk_adam_seg* seg, int& ns) -> bool {
        const float* lr = d->lr[color ? 1 : 0];
        ns = 0;
        auto dec_seg = [&](const lk_blob_span& sp, int step) {
            lk_adam_seg& s = seg[ns++];
            s.p = d->weights_rw + sp.offset; s.g = d->render.g_weights + sp.offset; s.m = d->adam_dec + sp.offset; s.v = d->adam_dec + nb + sp.offset;
            s.n = sp.n; s.lr = lr[0]; s.step = step; s.zero_grad = 1;
        };
        for (int k = 0; k < d->n_geo_dec; ++k) dec_seg(d->geo_dec[k], goff + it + 1);
        if (color) for (int k = 0; k < d->n_col_dec; ++k) dec_seg(d->col_dec[k], it - n_geo_l + 1);
        if (ns + 2 > LK_ADAM_MAX_SEG) return false;
        {
            lk_adam_seg& s = seg[ns++];
            s.p = d->geo_feats_rw; s.g = d->render.g_geo_feats; s.m = d->adam_rows; s.v = d->adam_rows + nrow; s.n = nrow; s.lr = lr[1]; s.step = goff + it + 1;
            s.row_index = d->rows; s.row_len = d->rows ? LK_C : 1; s.zero_grad = 1; s.p_f16 = (d->render.flags & LK_FLAG_FEATS_F16) ? 1 : 0;
            if (act_rows) { s.row_len = LK_C; s.row_flags = kn_h->act_flag; }
        }
        if (color) {
            lk_adam_seg& s = seg[ns++];
            s.p = d->col_feats_rw; s.g = d->render.g_col_feats; s.m = d->adam_rows + 2 * nrow; s.v = d->adam_rows + 3 * nrow; s.n = nrow; s.lr = lr[2];
            s.step = it - n_geo_l + 1; s.row_index = d->rows; s.row_len = d->rows ? LK_C : 1; s.zero_grad = 1;
            s.p_f16 = (d->render.flags & LK_FLAG_FEATS_F16) ? 1 : 0;
            if (act_rows) { s.row_len = LK_C; s.row_flags = kn_h->act_flag; }
        }
        return true;
    };
    bool repack_pending = false, stepped_pending = false;
    int split_pending = 0;          // the step of the iteration before was split: its trunk half is (maybe still) running on the weight-gradient stream
    // Step rider (LkStepRider, lk_kernels.h): in a 'color' iteration that is followed by another one in this call, with no gradient
    // exchange in between (phases == 3), the Adam step happens inside the reduction launch of the backward
    const bool train_geo = d->train_geo_decoder != 0;      // (its gradients come from a launch of their own, k_geo_wgrad: no owner thread in the reduction launch -> no rider)
    const bool rider_ok = pre && !xd && !train_geo && (phases & 3) == 3 && d->render.weights == d->weights_rw && d->render.g_weights != nullptr &&
                          (!(d->render.flags & LK_FLAG_REL_POS) || lk_relpos_fused(d->render.flags | LK_FLAG_GRAD_WEIGHTS)) &&
                          d->n_geo_dec + d->n_col_dec <= 16 && nb < (1ll << 31);
    // fix_color_decoder (the end-of-sequence refinement, Mapper.py:531-535): of the colour-stage spans only embedder_rel_pos._B is left (or
    // nothing) - the backward then skips the weight-gradient rows and reductions, and the matrix fragments never change
    bool embed_only = true;
    for (int k = 0; k < d->n_col_dec; ++k)
        embed_only = embed_only && d->col_dec[k].offset >= R_EB && d->col_dec[k].offset + d->col_dec[k].n <= R_EB + 3 * 10;
    if (train_geo) embed_only = false;          // (the geometry decoder's weight gradients need the full backward)
    bool w_next_ready = false;
    bool x_early = false;                          // this iteration's exposure backward + Adam already ran (in the gather launch)
    int sum_lo = -1, sum_hi = -1;                  // iterations whose loss rows are summed at the end of the call (the exposure variant's are not)
    bool x_fwd_done = it_begin > n_geo_l;          // a later call of a phase-split sequence: the step launch of the iteration before did the forward
    for (int it = it_begin; it < it_end; ++it) {
        const bool color = it >= n_geo_l;
        const bool use_rider = rider_ok && !embed_only && color && it + 1 < it_end && d->n_col_dec > 0;
        lk_render_desc rd = d->render;
        const bool xit = xd != nullptr && color;            // this iteration's loss is the exposure variant (its own launch after the composite)
        rd.flags = (d->render.flags & (LK_FLAG_REL_POS | LK_FLAG_UNIT_LOSS_GRADS | LK_FLAG_FEATS_F16)) | (color ? LK_FLAG_STAGE_COLOR : 0) | LK_FLAG_SAVE_ACT |
                   LK_FLAG_GRAD_FEATS | LK_FLAG_GRAD_WEIGHTS | LK_FLAG_ZERO_ABSENT | LK_FLAG_MAPPER_LOSS | (embed_only && color ? LK_FLAG_EMBED_GRADS_ONLY : 0) |
                   (train_geo ? LK_FLAG_GRAD_GEO_DECODER : 0);
        // d logits = w sigma' A with a LEARNED A: unit scale only relative to the power of two the exposure step keeps in xd->bwd_scale (the
        // kernels apply it on top of their 2^10); without that cell, or with the rel-pos MLP (its fused backward has no such hook): bf16 pieces
        const bool x_unit = xd && xd->bwd_scale && !(d->render.flags & LK_FLAG_REL_POS);
        if (xd && !x_unit) rd.flags &= ~LK_FLAG_UNIT_LOSS_GRADS;
        if (x_unit) rd.flags |= LK_FLAG_UNIT_LOSS_GRADS;
        if (xit) rd.flags = (rd.flags & ~LK_FLAG_MAPPER_LOSS) | LK_FLAG_COLOR_LOGITS;
        rd.stats_chunk = R;
        rd.loss_gt_color = d->gt_color; rd.loss_w_color = d->w_color; rd.loss_out4 = d->log + (size_t)it * 4;
        if (pre) {
            rd.rays_o = W0 + wk.rays_o + (size_t)it * R * 3; rd.rays_d = W0 + wk.rays_d + (size_t)it * R * 3;
            rd.gt_depth = W0 + wk.gt_depth + (size_t)it * R; rd.loss_gt_color = W0 + wk.gt_color + (size_t)it * R * 3;
            if (rd.r2_ray) rd.r2_ray = W0 + wk.r2_ray + (size_t)it * R;
            rd.z = W0 + wk.z + (size_t)it * Pn; rd.nbr_count = reinterpret_cast<int32_t*>(W0 + wk.nbr_count) + (size_t)it * Pn;
            rd.nbr_idx = reinterpret_cast<int32_t*>(W0 + wk.nbr_idx) + (size_t)it * Pn * LK_K; rd.nbr_w = W0 + wk.nbr_w + (size_t)it * Pn * LK_K;
            if ((goff + it) & 1) rd.c_col = W0 + wk.c_col_alt;          // see MapWork::c_col_alt: the split step's k_wgrad may still be reading the other one
            const int ck = chunk_of(it);
            if ((phases & 1) && it == chunk_start(ck)) {     // entering chunk ck: enqueue chunk ck + 1, then wait for chunk ck
                const int rc = enqueue_chunk(ck + 1);
                if (rc != LK_OK) return rc;
            }
            if ((phases & 1) && (it == chunk_start(ck) || it == it_begin)) {
                PreStream& ps = pre_stream();
                if (ps.ok) (void)hipStreamWaitEvent(st, ps.ev[ck % LK_PRE_CHUNKS], 0);
            }
        }
        if (phases & 1) {
            int rc;
            if (!pre) {
                rc = lk_gather_rays(d->depth_stack, d->color_stack, d->c2w_stack, d->c2w_stride, d->r2_map_stack, d->frame_id, d->rnd + (size_t)it * R, R,
                                    d->H, d->W, d->H0, d->W0, d->w, d->fx, d->fy, d->cx, d->cy, const_cast<float*>(rd.rays_o),
                                    const_cast<float*>(rd.rays_d), const_cast<float*>(rd.gt_depth), d->gt_color, nullptr, nullptr,
                                    const_cast<float*>(rd.r2_ray), st);
                if (rc != LK_OK) return rc;
                rc = lk_inside_mask(rd.gt_depth, R, nullptr, const_cast<float*>(rd.gt_depth), d->thr, d->scratch_u32, st);
                if (rc != LK_OK) return rc;
            }
            // the loss gradient is final when the composite kernel has written it: its backward rides in the same launch
            const int32_t* live = pre ? reinterpret_cast<const int32_t*>(W0 + wk.n_live) + it : nullptr;
            // the fragment repack of the iteration before rides in this iteration's interpolation launch (see the end of the loop body)
            LkRepackRider rr;
            rr.frag = d->weights_frag_rw; rr.src = stepped_pending ? W0 + wk.w_next : nullptr;
            rr.copy_dst = stepped_pending ? d->weights_rw : nullptr; rr.copy_n = (int)nb;
            rr.skip_trunk = (stepped_pending && split_pending) ? 1 : 0;
            const bool join_side = rr.skip_trunk != 0;
            if (xit && it == (n_geo_l > it_begin ? n_geo_l : it_begin) && !x_fwd_done) {
                // affines of the window's keyframes for the first 'color' iteration of this call (later ones: the step launch below)
                rc = lk_launch_exposure_step(*xd, 2, 1, beta1, beta2, eps, st);
                if (rc != LK_OK) return rc;
            }
            x_fwd_done = x_fwd_done || xit;
            // phase-split caller: the step of 'color' iteration it - 1 (its own call) moved the colour decoder and left the repack to this launch
            const bool split_repack = pre && (phases & 3) == 1 && it == it_begin && it > n_geo_l && !embed_only && d->n_col_dec > 0 &&
                                      d->render.weights == d->weights_rw;
            // the composite, the loss and its backward as the prologue of the decoder backward (no k_composite launch)
            const bool comp_bwd = pre && !xit && !train_geo;
            if (comp_bwd) { if (sum_lo < 0) sum_lo = it; sum_hi = it + 1; }
            // exposure variant: the composite runs inside the loss kernel below (one launch instead of two)
            rc = lk_render_fwd_impl(&rd, st, (xit ? LK_SKIP_COMPOSITE : LK_FUSE_COMPOSITE_BWD) | (pre ? (LK_LOSS_PREZEROED | LK_PRESAMPLED) : 0) |
                                    (sort_ahead ? LK_SEG_SORTED : 0) | (comp_bwd ? LK_COMPOSITE_IN_BWD : 0), live,
                                    (repack_pending || stepped_pending || split_repack) ? &rr : nullptr, nullptr, nullptr, nullptr, join_side);
            repack_pending = false; stepped_pending = false; split_pending = 0;
            if (rc != LK_OK) return rc;
            if (xit) {          // composite + Mapper.py:697-715 on the rendered logits in one launch: d depth, d logits, loss row, d loss / d affine
                LkCompositeArgs ca;
                memset(&ca, 0, sizeof(ca));
                ca.R = R; ca.S = rd.S; ca.min_nn = rd.min_nn; ca.coef = rd.coef;
                ca.raw = rd.raw; ca.z = rd.z; ca.nbr_count = rd.nbr_count; ca.gt_depth = rd.gt_depth;
                ca.depth = rd.depth; ca.var = rd.var; ca.color = rd.color; ca.valid_ray = rd.valid_ray;
                ca.keep_depth = (rd.flags & LK_FLAG_Z_GIVEN) ? 1 : 0;
                rc = lk_launch_composite_loss_exposure(ca, rd.loss_gt_color, reinterpret_cast<const int32_t*>(W0 + wk.frame_id) + (size_t)it * R, xd->aff, xd->F,
                                                       d->w_color, const_cast<float*>(rd.d_depth), const_cast<float*>(rd.d_color), rd.loss_out4, xd->g_aff, st);
                if (rc != LK_OK) return rc;
            }
            LkBwdExtra ex;
            memset(&ex, 0, sizeof(ex));
            if (sort_ahead) {
                ex.seg_list = reinterpret_cast<int32_t*>(W0 + wk.seg_list) + (size_t)it * Pn * LK_K;
                ex.seg_total = reinterpret_cast<int32_t*>(W0 + wk.seg_total) + it;
            }
            ex.live_rays = live;
            ex.act_flag = act_rows ? kn_h->act_flag : nullptr;
            ex.signal_rows = ((phases & 3) == 1) ? d->signal_rows : 0;
            ex.dscale = (xd && (rd.flags & LK_FLAG_UNIT_LOSS_GRADS)) ? xd->bwd_scale : nullptr;
            LkStepRider sr;
            if (use_rider) {
                lk_adam_seg seg[LK_ADAM_MAX_SEG];
                memset(seg, 0, sizeof(seg));
                int ns = 0;
                LK_REQUIRE(build_segs(it, color, seg, ns), "lk_map_frame: too many optimiser segments");
                memset(&sr, 0, sizeof(sr));
                sr.g = d->render.g_weights; sr.p = d->weights_rw; sr.w_next = W0 + wk.w_next; sr.m = d->adam_dec; sr.v = d->adam_dec + nb;
                sr.beta1 = beta1; sr.beta2 = beta2; sr.eps = eps;
                long long nmax = 0;
                for (int q = 0; q < ns; ++q) {
                    const double bc1 = 1.0 - pow((double)beta1, (double)seg[q].step), bc2 = 1.0 - pow((double)beta2, (double)seg[q].step);
                    const float step_size = (float)((double)seg[q].lr / bc1), bc2_sqrt = (float)sqrt(bc2);
                    if (q < ns - 2) {       // decoder span
                        LkStepSpan& sp = sr.span[sr.n_span++];
                        sp.off = (int)(seg[q].p - d->weights_rw); sp.n = (int)seg[q].n; sp.step_size = step_size; sp.bc2_sqrt = bc2_sqrt;
                    } else {                // the two feature-row segments
                        AdamSegDev& f = sr.feat[sr.n_feat++];
                        f.p = seg[q].p; f.g = seg[q].g; f.m = seg[q].m; f.v = seg[q].v; f.n = seg[q].n; f.step_size = step_size; f.bc2_sqrt = bc2_sqrt;
                        f.row_index = seg[q].row_index; f.row_len = seg[q].row_len > 0 ? seg[q].row_len : 1; f.zero_grad = seg[q].zero_grad; f.p_f16 = seg[q].p_f16;
                        f.row_flags = seg[q].row_flags; f.g_compact = seg[q].g_compact;
                        if (seg[q].n > nmax) nmax = seg[q].n;
                    }
                }
                sr.feat_gx = (int)(nmax > 0 ? (lk_cdiv(nmax, 256) > 2048 ? 2048 : lk_cdiv(nmax, 256)) : 1);
                if (!w_next_ready) {        // elements outside the stepped spans ride through the copy-back unchanged
                    LK_HIP_TRY(hipMemcpyAsync(W0 + wk.w_next, d->weights_rw, sizeof(float) * (size_t)nb, hipMemcpyDeviceToDevice, st));
                    w_next_ready = true;
                }
                ex.step = &sr;
                // the colour trunk's half of this step on the weight-gradient stream, joined in front of the next iteration's decoder launch
                ex.split_reduce = 1; ex.split_frag = d->weights_frag_rw; ex.split_master = d->weights_rw; ex.split_done = &split_pending;
            }
            if (comp_bwd) ex.loss_rows = W0 + wk.loss_rows + (size_t)it * 4 * lk_cdiv(Pn, 32);
            // exposure encoding, one process: the backward + Adam half of the exposure step rides in this backward's gather launch (d affine is
            // final since the loss kernel above; with ranks it is exchanged first and the whole step stays in the Adam launch)
            ExposureStepArgs xa_early;
            x_early = false;
            if (xit && pre && (phases & 3) == 3) {
                rc = lk_exposure_step_args(*xd, 1, it - n_geo_l + 1, beta1, beta2, eps, &xa_early);
                if (rc != LK_OK) return rc;
                ex.xstep = &xa_early;
                x_early = true;
            }
            rc = lk_render_bwd_impl(&rd, st, (xit ? 0 : LK_SKIP_COMPOSITE_BWD) | LK_SEG_SORTED | (comp_bwd ? LK_COMPOSITE_IN_BWD : 0), pre ? &ex : nullptr);
            if (rc != LK_OK) return rc;
        }
        if (phases & 2) {
            lk_adam_seg seg[LK_ADAM_MAX_SEG];
            memset(seg, 0, sizeof(seg));
            int ns = 0;
            LK_REQUIRE(build_segs(it, color, seg, ns), "lk_map_frame: too many optimiser segments");
            if (d->grad_bucket && (phases & 3) == 2) {
                // data-parallel caller: the summed gradients are in its all-reduce bucket - decoder spans at their bucket offsets, the optimised
                // rows compact in row-list order - and the step reads them there (no unpack launch; the pack cleared the sources)
                LK_REQUIRE(d->rows != nullptr && !xd, "lk_map_frame: grad_bucket needs a row list and no exposure encoding");
                int q = 0;
                for (int k = 0; k < d->n_geo_dec; ++k, ++q) { seg[q].g = const_cast<float*>(d->grad_bucket) + d->bucket_geo_dec[k]; seg[q].zero_grad = 0; }
                if (color) for (int k = 0; k < d->n_col_dec; ++k, ++q) { seg[q].g = const_cast<float*>(d->grad_bucket) + d->bucket_col_dec[k]; seg[q].zero_grad = 0; }
                seg[q].g = const_cast<float*>(d->grad_bucket) + d->bucket_geo_rows; seg[q].g_compact = 1; seg[q].zero_grad = 0; ++q;
                if (color) { seg[q].g = const_cast<float*>(d->grad_bucket) + d->bucket_col_rows; seg[q].g_compact = 1; seg[q].zero_grad = 0; ++q; }
                LK_REQUIRE(q == ns, "lk_map_frame: segment order changed under grad_bucket");
            }
            if (use_rider) {        // stepped inside k_bwd_reduce; the next iteration's interpolation launch copies the blob back and repacks
                stepped_pending = true;
                continue;
            }
            int rc;
            if (xit) {          // exposure MLP backward + its Adam groups + the affines of the next iteration: one more block of the Adam launch
                ExposureStepArgs xa;
                rc = lk_exposure_step_args(*xd, x_early ? 2 : 3, it - n_geo_l + 1, beta1, beta2, eps, &xa);
                if (rc != LK_OK) return rc;
                rc = lk_adam_step_x(seg, ns, beta1, beta2, eps, &xa, st);
            } else {
                rc = lk_adam_step(seg, ns, beta1, beta2, eps, st);
            }
            if (rc != LK_OK) return rc;
            if ((color && !embed_only) || train_geo) {  // the matrix fragments are copies of the colour-decoder matrices, which only move in this stage
                // (and of the geometry decoder's, which move in both stages when they are trained)
                // (rd.weights == weights_rw: the next iteration's forward reads what this step wrote)
                if (pre && (phases & 3) == 3 && it + 1 < it_end && d->render.weights == d->weights_rw) repack_pending = true;
                else if (color && !embed_only && pre && (phases & 3) == 2 && it + 1 < d->iters && d->render.weights == d->weights_rw) {
                    // phase-split caller: the phase-1 call of iteration it + 1 repacks in its interpolation launch (split_repack below)
                } else {
                    rc = lk_weights_repack(d->weights_rw, d->weights_frag_rw, st);
                    if (rc != LK_OK) return rc;
                }
            }
        }
    }
    if ((phases & 3) != 3) {
        // a phase-split caller (one call per iteration around its gradient exchange) gets ONE sum launch, behind the backward of the sequence's
        // last iteration: the rows of every iteration that left per-tile terms - all of them, or the 'geometry' ones with exposure encoding
        // (decided from the SEQUENCE, not from this call's own iterations: with exposure encoding the last call is a 'color' iteration
        // that leaves no terms itself, and the 'geometry' rows [0, n_geo) would never be summed)
        sum_lo = -1;
        if ((phases & 1) && it_end == d->iters && pre && !train_geo) {
            sum_lo = 0;
            sum_hi = xd ? (n_geo_l < 0 ? 0 : (n_geo_l > d->iters ? d->iters : n_geo_l)) : d->iters;
        }
    }
    if (sum_lo >= 0) {
        if (sum_hi > sum_lo)
            hipLaunchKernelGGL(k_loss_rows_sum, dim3(sum_hi - sum_lo), dim3(256), 0, st, W0 + wk.loss_rows, lk_cdiv(Pn, 32), (int)Pn, d->render.S,
                               reinterpret_cast<const int32_t*>(W0 + wk.n_live), sum_lo, d->log);
    }
    LK_LAUNCH_CHECK();
    return LK_OK;
}

extern "C" int lk_map_wait_rows(const lk_map_desc* d, void* stream_) {
    LK_REQUIRE(d != nullptr, "lk_map_wait_rows: NULL descriptor");
    return lk_wait_rows_event((hipStream_t)stream_);
}

// The look-ahead chunk that holds iteration `it` has been enqueued (see the header): `stream` waits for its event.
extern "C" int lk_map_wait_lists(const lk_map_desc* d, int32_t it, void* stream_) {
    LK_REQUIRE(d != nullptr && it >= 0 && it < d->iters, "lk_map_wait_lists: bad arguments");
    PreStream& ps = pre_stream();
    if (!ps.ok || d->work == nullptr || d->render.R > LK_LOOP_MAX_R) return LK_OK;      // no look-ahead: the lists are written in the call's stream order
    const int pre_chunk = map_pre_chunk(d->iters);
    const int ck = it == 0 ? 0 : 1 + (it - 1) / pre_chunk;
    LK_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream_, ps.ev[ck % LK_PRE_CHUNKS], 0));
    return LK_OK;
}
