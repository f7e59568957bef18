// extern "C" entry points of libloopyhip.so (see include/loopy_hip.h).
#include "lk_common.h"
#include "lk_kernels.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

using namespace lkw;

static thread_local char g_err[512] = "";

void lk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int lk_version(void) { return LK_ABI_VERSION; }

// ------------------------------------------------------------------ operand-range status word (loopy_hip.h)
namespace {
struct StatusWord { volatile unsigned* host = nullptr; unsigned* dev = nullptr; bool tried = false; };
StatusWord& status_word() {
    static StatusWord w;
    if (!w.tried) {
        w.tried = true;
        void* h = nullptr; void* d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            memset(h, 0, 64);
            w.host = static_cast<volatile unsigned*>(h); w.dev = static_cast<unsigned*>(d);
        } else (void)hipGetLastError();          // no word: the checks are off (lk_status_peek says so), nothing else changes
    }
    return w;
}
}  // namespace
unsigned* lk_status_dev() { return status_word().dev; }
int lk_status_gate(const char* who) {
    StatusWord& w = status_word();
    const unsigned bits = w.host ? *w.host : 0u;
    if (!bits) return LK_OK;
    lk_set_error("%s: refused - an earlier launch left the range of the split fp16 products (status %u:%s%s); the results since then are not "
                 "the reference's fp32 results.  lk_status_clear() after fixing the inputs", who, bits,
                 (bits & LK_STATUS_WEIGHT_RANGE) ? " a decoder matrix entry is non-finite or |w| >= 32768" : "",
                 (bits & LK_STATUS_ACT_RANGE) ? " a forward operand (feature / activation) is non-finite or |x| >= 65504" : "");
    return LK_ERR_RANGE;
}
extern "C" int lk_status_peek(uint32_t* bits) {
    LK_REQUIRE(bits != nullptr, "lk_status_peek: NULL bits");
    StatusWord& w = status_word();
    if (!w.host) { *bits = 0; lk_set_error("lk_status_peek: the status word could not be allocated - range checks are off"); return LK_ERR_STATE; }
    *bits = *w.host;
    return LK_OK;
}
extern "C" int lk_status_sync(void* stream_, uint32_t* bits) {
    LK_HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    return lk_status_peek(bits);
}
extern "C" int lk_status_clear(void) {
    StatusWord& w = status_word();
    if (w.host) *w.host = 0u;
    return LK_OK;
}
extern "C" const char* lk_last_error(void) { return g_err; }

// ------------------------------------------------------------------ weight layout table
namespace {
struct WEntry { const char* name; int off, rows, cols, ld, split, shift; };
#define GU(i) (G_U0 + (i) * G_USTRIDE)
#define CU(i) (C_U0 + (i) * C_USTRIDE)
const WEntry kEntries[] = {
    {"geo_decoder.embedder._B", G_EB, 3, EG, EGP, 0, 0},
    {"geo_decoder.pts_linears.0.weight", G_W0, HG, EG, EGP, 0, 0},
    {"geo_decoder.pts_linears.0.bias", G_B0, HG, 1, 1, 0, 0},
    {"geo_decoder.pts_linears.1.weight", G_W1, HG, HG, HG, 0, 0},
    {"geo_decoder.pts_linears.1.bias", G_B1, HG, 1, 1, 0, 0},
    {"geo_decoder.pts_linears.2.weight", G_W2, HG, HG, HG, 0, 0},
    {"geo_decoder.pts_linears.2.bias", G_B2, HG, 1, 1, 0, 0},
    {"geo_decoder.pts_linears.3.weight", G_W3, HG, EG + HG, EGP + HG, EG, EGP},
    {"geo_decoder.pts_linears.3.bias", G_B3, HG, 1, 1, 0, 0},
    {"geo_decoder.pts_linears.4.weight", G_W4, HG, HG, HG, 0, 0},
    {"geo_decoder.pts_linears.4.bias", G_B4, HG, 1, 1, 0, 0},
    {"geo_decoder.fc_c.0.weight", GU(0), HG, CF, CF, 0, 0}, {"geo_decoder.fc_c.0.bias", GU(0) + a64(HG * CF), HG, 1, 1, 0, 0},
    {"geo_decoder.fc_c.1.weight", GU(1), HG, CF, CF, 0, 0}, {"geo_decoder.fc_c.1.bias", GU(1) + a64(HG * CF), HG, 1, 1, 0, 0},
    {"geo_decoder.fc_c.2.weight", GU(2), HG, CF, CF, 0, 0}, {"geo_decoder.fc_c.2.bias", GU(2) + a64(HG * CF), HG, 1, 1, 0, 0},
    {"geo_decoder.fc_c.3.weight", GU(3), HG, CF, CF, 0, 0}, {"geo_decoder.fc_c.3.bias", GU(3) + a64(HG * CF), HG, 1, 1, 0, 0},
    {"geo_decoder.fc_c.4.weight", GU(4), HG, CF, CF, 0, 0}, {"geo_decoder.fc_c.4.bias", GU(4) + a64(HG * CF), HG, 1, 1, 0, 0},
    {"geo_decoder.output_linear.weight", G_WO, 1, HG, HG, 0, 0},
    {"geo_decoder.output_linear.bias", G_BO, 1, 1, 1, 0, 0},
    {"color_decoder.embedder._B", C_EB, 3, 20, 20, 0, 0},
    {"color_decoder.pts_linears.0.weight", C_W0, HC, EC, EC, 0, 0},
    {"color_decoder.pts_linears.0.bias", C_B0, HC, 1, 1, 0, 0},
    {"color_decoder.pts_linears.1.weight", C_W1, HC, HC, HC, 0, 0},
    {"color_decoder.pts_linears.1.bias", C_B1, HC, 1, 1, 0, 0},
    {"color_decoder.pts_linears.2.weight", C_W2, HC, HC, HC, 0, 0},
    {"color_decoder.pts_linears.2.bias", C_B2, HC, 1, 1, 0, 0},
    {"color_decoder.pts_linears.3.weight", C_W3, HC, EC + HC, EC + HC, 0, 0},
    {"color_decoder.pts_linears.3.bias", C_B3, HC, 1, 1, 0, 0},
    {"color_decoder.pts_linears.4.weight", C_W4, HC, HC, HC, 0, 0},
    {"color_decoder.pts_linears.4.bias", C_B4, HC, 1, 1, 0, 0},
    {"color_decoder.fc_c.0.weight", CU(0), HC, CF, CF, 0, 0}, {"color_decoder.fc_c.0.bias", CU(0) + a64(HC * CF), HC, 1, 1, 0, 0},
    {"color_decoder.fc_c.1.weight", CU(1), HC, CF, CF, 0, 0}, {"color_decoder.fc_c.1.bias", CU(1) + a64(HC * CF), HC, 1, 1, 0, 0},
    {"color_decoder.fc_c.2.weight", CU(2), HC, CF, CF, 0, 0}, {"color_decoder.fc_c.2.bias", CU(2) + a64(HC * CF), HC, 1, 1, 0, 0},
    {"color_decoder.fc_c.3.weight", CU(3), HC, CF, CF, 0, 0}, {"color_decoder.fc_c.3.bias", CU(3) + a64(HC * CF), HC, 1, 1, 0, 0},
    {"color_decoder.fc_c.4.weight", CU(4), HC, CF, CF, 0, 0}, {"color_decoder.fc_c.4.bias", CU(4) + a64(HC * CF), HC, 1, 1, 0, 0},
    {"color_decoder.output_linear.weight", C_WO, 3, HC, HC, 0, 0},
    {"color_decoder.output_linear.bias", C_BO, 3, 1, 1, 0, 0},
    {"color_decoder.embedder_rel_pos._B", R_EB, 3, 10, 10, 0, 0},
    {"color_decoder.mlp_col_neighbor.linear1.weight", R_W1, HC, KR, KRP, 0, 0},
    {"color_decoder.mlp_col_neighbor.linear1.bias", R_B1, HC, 1, 1, 0, 0},
    {"color_decoder.mlp_col_neighbor.linear2.weight", R_W2, CF, HC, HC, 0, 0},
    {"color_decoder.mlp_col_neighbor.linear2.bias", R_B2, CF, 1, 1, 0, 0},
};
const int kNumEntries = (int)(sizeof(kEntries) / sizeof(kEntries[0]));
}  // namespace

extern "C" int lk_weight_layout(lk_weight_entry* out, int max_entries) {
    if (out) {
        for (int i = 0; i < kNumEntries && i < max_entries; ++i) {
            memset(&out[i], 0, sizeof(lk_weight_entry));
            strncpy(out[i].name, kEntries[i].name, sizeof(out[i].name) - 1);
            out[i].offset = kEntries[i].off;
            out[i].rows = kEntries[i].rows;
            out[i].cols = kEntries[i].cols;
            out[i].ld = kEntries[i].ld;
            out[i].col_split = kEntries[i].split;
            out[i].col_shift = kEntries[i].shift;
        }
    }
    return kNumEntries;
}
extern "C" int64_t lk_weight_blob_floats(void) { return BLOB_FLOATS; }

extern "C" int64_t lk_render_act_floats(int32_t R, int32_t S, uint32_t flags) {
    (void)flags;
    return (int64_t)R * S * LK_ACT_FLOATS_PER_SAMPLE;
}

// ------------------------------------------------------------------ backward scratch layout
namespace {
struct BwdLayout { int64_t dfeat, d_raw, dc_geo, dc_col, dp_embed, dp_embed_col, dp_rel, dp_total, dw_rel, w_eff, dlogit, aff_part, part_bg, part_br, hbar, w_sum, dy_col, rows, dw1_part, dw2_part, wg_part, geo_part, seg_rank, seg_list, total; };
BwdLayout bwd_layout(int64_t P, uint32_t flags) {
    BwdLayout L;
    int64_t o = 0;
    // every region starts on a 16-byte boundary whatever P is (float4 accesses; P = R*S need not be a multiple of 4)
    auto al = [](int64_t x) { return (x + 3) / 4 * 4; };
    L.d_raw = o; o += al(4 * P);
    L.dc_geo = o; o += al(32 * P);
    L.dc_col = o; o += al(32 * P);
    L.dp_embed = o; o += al(4 * P);
    L.dp_embed_col = o; o += al(4 * P);
    L.dp_rel = o; o += al(4 * P);
    L.dp_total = o; o += al(4 * P);
    L.dw_rel = o; o += al(8 * P);
    L.w_eff = o; o += al(8 * P);
    L.dlogit = o; o += al(4 * P);
    L.aff_part = o; o += al((int64_t)lk_cdiv(P, 32) * 12);         // per-tile sums of the exposure affine's gradient (k_decode_bwd)
    L.part_bg = o; o += al((int64_t)lk_cdiv(lk_cdiv(P, 32), 4) * 288);
    L.part_br = o; o += al((int64_t)lk_cdiv(lk_cdiv(P, 4), 4) * 32);
    const bool color = (flags & LK_FLAG_STAGE_COLOR) != 0, gw = (flags & LK_FLAG_GRAD_WEIGHTS) != 0;
    L.hbar = o; if (color && gw && (flags & LK_FLAG_REL_POS)) o += al(128 * P);
    L.dfeat = o; if (color && (flags & LK_FLAG_REL_POS) && (flags & LK_FLAG_GRAD_FEATS)) o += al(8 * 32 * P);
    L.w_sum = o; o += al(P);
    L.dy_col = o; if (color && gw) o += al(640 * P);
    // linear1 of the rel-pos MLP: neighbour rows for k_wgrad, or (mapper mode) the workgroup tiles of k_relpos_bwd_fused
    const bool rp_w = color && gw && (flags & LK_FLAG_REL_POS), fused = rp_w && lk_relpos_fused(flags);
    L.rows = o; if (rp_w && !fused) o += al(8 * 192 * P);
    L.dw1_part = o; if (fused) o += al((int64_t)lk_relpos_bwd_parts((int)P) * 128 * 64);
    L.dw2_part = o; if (fused) o += al(lk_dw2_part_floats((int)P));
    L.wg_part = o; if (color && gw) o += al(lk_wgrad_part_floats(P, (flags & LK_FLAG_REL_POS) != 0));
    L.geo_part = o; if (gw && (flags & LK_FLAG_GRAD_GEO_DECODER)) o += al(lk_geo_wgrad_part_floats((int)P));
    L.seg_rank = o; if (flags & LK_FLAG_GRAD_FEATS) o += al(8 * P);
    L.seg_list = o; if (flags & LK_FLAG_GRAD_FEATS) o += al(8 * P);
    L.total = o;
    return L;
}
}  // namespace

// ------------------------------------------------------------------ render forward
static int check_desc(const lk_render_desc* d, const char* who) {
    if (!d) { lk_set_error("%s: NULL descriptor", who); return LK_ERR_ARG; }
    if (d->R < 0 || d->S < 1 || d->S > LK_S_MAX) { lk_set_error("%s: bad R/S (%d, %d)", who, d->R, d->S); return LK_ERR_ARG; }
    if ((int64_t)d->R * d->S >= (1ll << 31)) { lk_set_error("%s: R*S too large", who); return LK_ERR_ARG; }
    if (!d->knn) { lk_set_error("%s: NULL knn handle", who); return LK_ERR_ARG; }
    if (d->R > 0 && (!d->rays_o || !d->rays_d || !d->gt_depth || !d->geo_feats || !d->weights || !d->weights_frag || !d->z || !d->nbr_idx ||
                     !d->nbr_w || !d->nbr_count || !d->c_geo || !d->raw)) {
        lk_set_error("%s: NULL buffer in descriptor", who); return LK_ERR_ARG;
    }
    if ((d->flags & LK_FLAG_STAGE_COLOR) && d->R > 0 && (!d->col_feats || !d->c_col)) {
        lk_set_error("%s: colour stage needs col_feats and c_col", who); return LK_ERR_ARG;
    }
    if ((d->flags & (LK_FLAG_REL_POS | LK_FLAG_TRACKER)) && d->R > 0 && !d->pos) {
        lk_set_error("%s: rel-pos / tracker mode needs pos", who); return LK_ERR_ARG;
    }
    if (d->stats_chunk < 1) { lk_set_error("%s: stats_chunk must be >= 1", who); return LK_ERR_ARG; }
    return LK_OK;
}

// rows of the batch counting-sorted by point for the feature-gradient gather (k_seg_count .. k_seg_place, lk_bwd2.hip): on the
// second stream when there is one (it needs nothing but the neighbour indices), the caller's stream waits for `link` later
namespace { struct SideStream; SideStream& side_stream(); }
static int lk_wait_side_join(hipStream_t st);        // `st` waits for the weight-gradient stream's last join event (defined behind SideStream)
static void seg_args(const lk_render_desc* d, int P, LkFeatScatterArgs& fs);
static int seg_sort_async(const lk_render_desc* d, int P, bool counted, hipStream_t st);

static void fill_sample_args(const lk_render_desc* d, int P, bool all_pos, LkSampleArgs& sa) {
    sa.R = d->R; sa.S = d->S; sa.P = P; sa.stats_chunk = d->stats_chunk; sa.flags = d->flags;
    sa.rays_o = d->rays_o; sa.rays_d = d->rays_d; sa.gt_depth = d->gt_depth; sa.r2_ray = d->r2_ray;
    sa.far_stats = all_pos ? nullptr : d->far_stats;
    sa.grid = d->knn->grid; sa.sorted = d->knn->sorted; sa.cell_start = d->knn->cell_start;
    sa.geo_feats = d->geo_feats; sa.col_feats = d->col_feats; sa.noise_geo = d->noise_geo; sa.noise_col = d->noise_col;
    sa.near_surface = d->near_surface; sa.far_surface = d->far_surface; sa.near_end = d->near_end; sa.r2_static = d->r2_static;
    sa.min_nn = d->min_nn;
    sa.z = d->z; sa.nbr_idx = d->nbr_idx; sa.nbr_w = d->nbr_w; sa.nbr_count = d->nbr_count; sa.c_geo = d->c_geo; sa.c_col = d->c_col;
    sa.seg_cnt = nullptr; sa.seg_rank = nullptr; sa.row_mask = nullptr; sa.live_rays = nullptr;
    sa.seg_P = 0; sa.seg_cnt_stride = 0; sa.seg_live = nullptr; sa.seg_key = nullptr;
    sa.rp_plain = nullptr; sa.rp_frag = nullptr; sa.rp_block0 = 0; sa.rp_copy_dst = nullptr; sa.rp_copy_n = 0; sa.rp_block1 = 0;
    sa.rp_m_lo = sa.rp_m_hi = sa.rp_skip_lo = sa.rp_skip_hi = 0;
}
// z and the neighbour lists of a batch (the part of the sampler that does not read the feature tables); needs ZERO_ABSENT /
// ALL_DEPTH_POS batches (no far_bb statistics)
int lk_presample(const lk_render_desc* d, hipStream_t st, const LkPresampleCount* cnt) {
    LK_REQUIRE(d && d->knn && d->rays_o && d->rays_d && d->gt_depth && d->z && d->nbr_idx && d->nbr_w && d->nbr_count, "lk_presample: NULL buffer");
    LK_REQUIRE(d->flags & (LK_FLAG_ALL_DEPTH_POS | LK_FLAG_ZERO_ABSENT), "lk_presample: needs ALL_DEPTH_POS / ZERO_ABSENT");
    LK_REQUIRE((int64_t)d->R * d->S < (1ll << 31), "lk_presample: R*S too large");
    if (d->R == 0) return LK_OK;
    LkSampleArgs sa;
    fill_sample_args(d, d->R * d->S, true, sa);
    if (cnt) {
        sa.seg_cnt = d->knn->seg_cnt; sa.seg_cnt_stride = d->knn->seg_stride; sa.seg_rank = cnt->seg_rank; sa.row_mask = d->grad_row_mask;
        sa.seg_P = cnt->P_iter; sa.seg_live = cnt->live_rays; sa.seg_key = cnt->key_of;
    }
    return lk_launch_sample_interp(sa, st, 1);
}

extern "C" int lk_render_fwd(const lk_render_desc* d, void* stream_) {
    const int rcg = lk_status_gate("lk_render_fwd");
    return rcg != LK_OK ? rcg : lk_render_fwd_impl(d, (hipStream_t)stream_, 0);
}

int lk_render_fwd_impl(const lk_render_desc* d, hipStream_t st, unsigned skip, const int32_t* live_rays, const LkRepackRider* repack, const LkTrackFinalArgs* pose,
                       const LkTrackLossArgs* comp, int* comp_tiles, bool join_side_before_decode) {
    int rc = check_desc(d, "lk_render_fwd");
    if (rc != LK_OK) return rc;
    if (d->R == 0) return LK_OK;
    LK_REQUIRE(d->depth && d->var && d->color && d->valid_ray, "lk_render_fwd: NULL output buffer");
    const int P = d->R * d->S;
    const bool all_pos = (d->flags & (LK_FLAG_ALL_DEPTH_POS | LK_FLAG_ZERO_ABSENT)) != 0;
    if (!all_pos) {
        LK_REQUIRE(d->far_stats != nullptr, "lk_render_fwd: far_stats required unless ALL_DEPTH_POS / ZERO_ABSENT");
        lk_launch_depth_stats(d->gt_depth, d->R, d->stats_chunk, d->far_stats, st);
    }
    LkSampleArgs sa;
    fill_sample_args(d, P, all_pos, sa);
    const bool presort = (skip & LK_FUSE_COMPOSITE_BWD) && (d->flags & LK_FLAG_GRAD_FEATS) && d->bwd_scratch && !(skip & LK_SEG_SORTED);
    if (presort) {       // the backward of this forward follows (lk_map_frame): its rows are counted per point by the sampler ...
        LkFeatScatterArgs fs;
        seg_args(d, P, fs);
        sa.seg_cnt = fs.seg_cnt; sa.seg_rank = fs.seg_rank; sa.row_mask = fs.row_mask;
    }
    sa.live_rays = live_rays;
    LK_REQUIRE(!repack || (skip & LK_PRESAMPLED), "lk_render_fwd: the repack rider needs a presampled batch");
    if (repack) {
        sa.rp_plain = repack->src ? repack->src : d->weights; sa.rp_frag = repack->frag;
        if (repack->copy_dst && repack->src) { sa.rp_copy_dst = repack->copy_dst; sa.rp_copy_n = repack->copy_n; }
        if (repack->skip_trunk) { sa.rp_m_lo = LK_FRAG_COL_LO; sa.rp_m_hi = LK_FRAG_COL_HI; sa.rp_skip_lo = C_EB; sa.rp_skip_hi = R_EB; }
    }
    LK_REQUIRE(!pose || !(skip & LK_PRESAMPLED), "lk_render_fwd: the pose prologue belongs to the search launch");
    lk_launch_sample_interp(sa, st, (skip & LK_PRESAMPLED) ? 2 : 0, pose);
    if (presort) {       // ... and sorted beside the decoders
        const int rc2 = seg_sort_async(d, P, true, st);
        if (rc2 != LK_OK) return rc2;
    }

    const bool color = (d->flags & LK_FLAG_STAGE_COLOR) != 0;
    LkDecodeArgs da;
    da.R = d->R; da.S = d->S; da.P = P; da.flags = d->flags;
    da.rays_o = d->rays_o; da.rays_d = d->rays_d; da.z = d->z;
    da.c_geo = d->c_geo; da.c_col = d->c_col; da.W = d->weights; da.Wfrag = d->weights_frag; da.affine = d->affine;
    da.raw = d->raw; da.act = d->act; da.live_rays = live_rays; da.tile_stride = 0;
    da.status = (d->flags & LK_FLAG_CHECK_RANGE) ? lk_status_dev() : nullptr;
    if (comp_tiles) *comp_tiles = 0;
    const bool fuse_small = (skip & LK_FUSE_SMALL) && lk_relpos_decode_fusable(da);
    if (color && (d->flags & LK_FLAG_REL_POS)) {
        LkRelposArgs ra;
        ra.R = d->R; ra.S = d->S; ra.P = P; ra.min_nn = d->min_nn;
        ra.rays_o = d->rays_o; ra.rays_d = d->rays_d; ra.z = d->z;
        ra.pos = d->pos; ra.col_feats = d->col_feats; ra.feats_f16 = (d->flags & LK_FLAG_FEATS_F16) ? 1 : 0; ra.live_rays = live_rays;
        ra.nbr_idx = d->nbr_idx; ra.nbr_w = d->nbr_w; ra.nbr_count = d->nbr_count;
        ra.W = d->weights; ra.Wfrag = d->weights_frag; ra.noise_col = d->noise_col; ra.c_col = d->c_col;
        ra.status = da.status;
        if (fuse_small) lk_launch_relpos_decode_fwd(ra, da, st, comp, comp_tiles);
        else lk_launch_relpos_fwd(ra, st);
    }
    if (join_side_before_decode) {
        // split step of the iteration before (LkBwdExtra::split_reduce): the colour trunk's stepped weights and fragments come from the
        // weight-gradient stream - first needed here; the interpolation and the rel-pos MLP above read rows, the blob's other spans and
        // the rel-pos fragments, which the launch stream itself stepped and repacked
        // (a correctness dependency ACROSS calls since the split step: a failed wait must not let the decoder read half-written fragments)
        const int rcj = lk_wait_side_join(st);
        if (rcj != LK_OK) return rcj;
    }
    if (!fuse_small) lk_launch_decode_fwd(da, st);

    if (skip & LK_SKIP_COMPOSITE) { LK_LAUNCH_CHECK(); return LK_OK; }     // the caller composites (fused loss kernel, lk_loop.hip)
    if ((skip & LK_COMPOSITE_IN_BWD) && (d->flags & LK_FLAG_MAPPER_LOSS)) { LK_LAUNCH_CHECK(); return LK_OK; }     // ... or the decoder backward does
    LkCompositeArgs ca;
    ca.R = d->R; ca.S = d->S; ca.min_nn = d->min_nn; ca.coef = d->coef;
    ca.raw = d->raw; ca.z = d->z; ca.nbr_count = d->nbr_count; ca.gt_depth = d->gt_depth;
    ca.depth = d->depth; ca.var = d->var; ca.color = d->color; ca.valid_ray = d->valid_ray;
    ca.gt_color = nullptr; ca.w_color = 0.0f; ca.use_color = 0; ca.d_depth = nullptr; ca.d_color = nullptr; ca.loss_out = nullptr; ca.d_raw = nullptr;
    ca.keep_depth = (d->flags & LK_FLAG_Z_GIVEN) ? 1 : 0;
    if (d->flags & LK_FLAG_MAPPER_LOSS) {
        LK_REQUIRE(d->loss_out4 && d->d_depth && d->d_color && d->loss_gt_color, "lk_render_fwd: MAPPER_LOSS needs loss_gt_color, loss_out4, d_depth, d_color");
        if (!(skip & LK_LOSS_PREZEROED)) LK_HIP_TRY(hipMemsetAsync(d->loss_out4, 0, 4 * sizeof(float), st));
        if (skip & LK_FUSE_COMPOSITE_BWD) {       // the loss gradient is final: d raw right away (one launch less per iteration)
            LK_REQUIRE(d->bwd_scratch != nullptr, "lk_render_fwd: fused composite backward needs bwd_scratch");
            ca.d_raw = d->bwd_scratch + bwd_layout((int64_t)P, d->flags).d_raw;
        }
        ca.gt_color = d->loss_gt_color; ca.w_color = d->loss_w_color; ca.use_color = (d->flags & LK_FLAG_STAGE_COLOR) ? 1 : 0;
        ca.d_depth = const_cast<float*>(d->d_depth); ca.d_color = const_cast<float*>(d->d_color); ca.loss_out = d->loss_out4;
    }
    lk_launch_composite(ca, st);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

// ------------------------------------------------------------------ render backward


extern "C" int64_t lk_render_bwd_scratch_floats(int32_t R, int32_t S, uint32_t flags) {
    return bwd_layout((int64_t)R * S, flags).total;
}

// Second stream for the weight-gradient reductions: they only consume what decode_bwd / relpos_bwd saved, so they run
// beside the rel-pos backward and the feature scatter (fork / join with events; created once per process).
namespace {
int g_serial = -1;            // -1: not decided yet (environment LK_SERIAL), 0 / 1: set by lk_set_serial
struct SideStream { hipStream_t st = nullptr; hipEvent_t fork = nullptr, mid = nullptr, join = nullptr, fork0 = nullptr, link = nullptr; bool ok = false; };
SideStream& side_stream() {
    static SideStream s, none;
    if (g_serial < 0) g_serial = getenv("LK_SERIAL") != nullptr ? 1 : 0;
    if (g_serial) return none;                                       // one stream: clean per-kernel timing
    if (!s.st) {
        s.ok = hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.mid, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.join, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.fork0, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.link, hipEventDisableTiming) == hipSuccess;
    }
    return s;
}
}  // namespace

static int lk_wait_side_join(hipStream_t st) {
    SideStream& s = side_stream();
    // the caller skipped the trunk's repack and copy-back because a split step recorded `join`: without the side stream that state cannot exist
    LK_REQUIRE(s.ok, "lk_render_fwd: a split step is pending but the weight-gradient stream is gone");
    LK_HIP_TRY(hipStreamWaitEvent(st, s.join, 0));
    return LK_OK;
}
// Test hook (tests/test_split_step_order.py): every weight-gradient launch that is forked onto the side stream is preceded there by a kernel
// that spins for `us` microseconds - the side stream then trails the launch stream by that much, which turns any missing ordering between
// the two (round 5: iteration it's k_wgrad against iteration it + 1's c_col writes) from a timing accident into a deterministic failure.
namespace { int g_side_delay_us = 0; }
__global__ void k_spin_us(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
extern "C" int lk_debug_side_delay(int32_t us) { g_side_delay_us = us > 0 ? us : 0; return LK_OK; }
extern "C" int lk_set_serial(int32_t on) { g_serial = on ? 1 : 0; return LK_OK; }
// Creates the library's two streams NOW instead of at their first use.  The runtime hands its few hardware queues to streams as they are
// created: a process that first creates dozens of other streams (torch's stream pool comes into being with the first collective of a
// process group, RCCL brings its own) and only then renders can find the library's side stream on the SAME hardware queue as the launch
// stream - the weight-gradient fork of every 'color' iteration then runs serialised behind 12-us barrier packets (measured: 370 instead of
// 308 us per iteration, tools/trace_window.py).  Call it right after the device is chosen; core.Engine does.
extern "C" int lk_streams_init(void) {
    if (lk_serial_mode()) return LK_OK;
    SideStream& s = side_stream();
    LkAuxStream& a = lk_aux_stream();
    // a failed creation costs the overlap, not the results: every user of the side streams checks `ok` and falls back to the launch
    // stream - so it is not an error of this call either; the process then runs as with LK_SERIAL=1
    if (!(s.ok && a.ok)) { g_serial = 1; lk_set_error("lk_streams_init: stream / event creation failed - running on the launch stream only"); }
    return LK_OK;
}
// "the feature-row gradients of the backward are final" (lk_map_desc::signal_rows -> lk_map_wait_rows): an event of its own, recorded on
// the LAUNCH stream right behind the gather - it does not depend on the side streams, so a caller that exchanges the rows on a
// communication stream is ordered behind the backward in the serial mode too (LK_SERIAL, or side-stream creation failed); if the event
// cannot be created the backward that was asked to signal fails instead of leaving the waiter without a dependency
namespace { hipEvent_t g_rows_ev = nullptr; bool g_rows_set = false; }
static int rows_event_record(hipStream_t st) {
    if (g_rows_ev == nullptr) LK_HIP_TRY(hipEventCreateWithFlags(&g_rows_ev, hipEventDisableTiming));
    LK_HIP_TRY(hipEventRecord(g_rows_ev, st));
    g_rows_set = true;
    return LK_OK;
}
int lk_wait_rows_event(hipStream_t st) {
    if (g_rows_set) LK_HIP_TRY(hipStreamWaitEvent(st, g_rows_ev, 0));
    return LK_OK;
}
extern "C" int lk_debug_occupancy(int32_t out[5]) {
    LK_REQUIRE(out != nullptr, "lk_debug_occupancy: out is null");
    out[0] = lk_occupancy_decode_fwd(); out[1] = lk_occupancy_decode_bwd(); out[2] = lk_occupancy_relpos_fwd();
    out[3] = lk_occupancy_relpos_bwd_fused(); out[4] = lk_occupancy_wgrad();
    return LK_OK;
}
bool lk_serial_mode() { if (g_serial < 0) g_serial = getenv("LK_SERIAL") != nullptr ? 1 : 0; return g_serial != 0; }
LkAuxStream& lk_aux_stream() {
    static LkAuxStream s, none;
    if (lk_serial_mode()) return none;
    if (!s.st) {
        // default priority: on a low-priority queue every launch of the chain waited 40-180 us for its dispatch while the chip sat idle
        // (the loop's first iterations wait for this stream)
        s.ok = hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&s.e0, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.e1, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.e2, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < LK_PRE_CHUNKS && s.ok; ++i) s.ok = hipEventCreateWithFlags(&s.ev[i], hipEventDisableTiming) == hipSuccess;
    }
    return s;
}

static void seg_args(const lk_render_desc* d, int P, LkFeatScatterArgs& fs) {
    const BwdLayout L = bwd_layout(P, d->flags);
    memset(&fs, 0, sizeof(fs));
    fs.P = P; fs.min_nn = d->min_nn; fs.nbr_idx = d->nbr_idx; fs.nbr_w = d->nbr_w; fs.nbr_count = d->nbr_count;
    fs.row_mask = d->grad_row_mask; fs.seg_cnt = d->knn->seg_cnt; fs.seg_off = d->knn->seg_off; fs.seg_sums = d->knn->seg_sums; fs.N = (int)d->knn->n;
    fs.seg_rank = reinterpret_cast<int32_t*>(d->bwd_scratch + L.seg_rank); fs.seg_list = reinterpret_cast<int32_t*>(d->bwd_scratch + L.seg_list);
}
static int seg_sort_async(const lk_render_desc* d, int P, bool counted, hipStream_t st) {
    LkFeatScatterArgs fs;
    seg_args(d, P, fs);
    SideStream& ss = side_stream();
    if (!ss.ok) return lk_launch_seg_sort(fs, counted, st);
    (void)hipEventRecord(ss.fork0, st);
    (void)hipStreamWaitEvent(ss.st, ss.fork0, 0);
    const int rc = lk_launch_seg_sort(fs, counted, ss.st);
    (void)hipEventRecord(ss.link, ss.st);
    return rc;
}

extern "C" int lk_render_bwd(const lk_render_desc* d, void* stream_) {
    const int rcg = lk_status_gate("lk_render_bwd");
    return rcg != LK_OK ? rcg : lk_render_bwd_impl(d, (hipStream_t)stream_, 0, nullptr);
}

LkBwdOffsets lk_bwd_offsets(int64_t P, uint32_t flags) {
    const BwdLayout L = bwd_layout(P, flags);
    LkBwdOffsets o;
    o.d_raw = L.d_raw; o.dp_total = L.dp_total; o.aff_part = L.aff_part;
    return o;
}

int lk_render_bwd_impl(const lk_render_desc* d, hipStream_t st, unsigned skip, const LkBwdExtra* ex) {
    int rc = check_desc(d, "lk_render_bwd");
    if (rc != LK_OK) return rc;
    if (d->R == 0) return LK_OK;
    LK_REQUIRE(d->act != nullptr && (d->flags & LK_FLAG_SAVE_ACT), "lk_render_bwd: the forward must run with SAVE_ACT and act");
    LK_REQUIRE(d->bwd_scratch != nullptr && d->d_depth != nullptr, "lk_render_bwd: bwd_scratch / d_depth missing");
    const uint32_t flags = d->flags;
    const bool color = (flags & LK_FLAG_STAGE_COLOR) != 0, relpos = color && (flags & LK_FLAG_REL_POS);
    const bool gf = (flags & LK_FLAG_GRAD_FEATS) != 0, gw = (flags & LK_FLAG_GRAD_WEIGHTS) != 0, gr = (flags & LK_FLAG_GRAD_RAYS) != 0;
    LK_REQUIRE(!gf || (d->g_geo_feats && (!color || d->g_col_feats)), "lk_render_bwd: GRAD_FEATS needs g_geo_feats/g_col_feats");
    LK_REQUIRE(!gw || d->g_weights, "lk_render_bwd: GRAD_WEIGHTS needs g_weights");
    LK_REQUIRE(!gr || (((skip & LK_SKIP_RAYS_BWD) || (d->g_rays_o && d->g_rays_d)) && d->pos), "lk_render_bwd: GRAD_RAYS needs g_rays_o/g_rays_d/pos");
    LK_REQUIRE(!color || d->d_color, "lk_render_bwd: colour stage needs d_color");
    const int P = d->R * d->S;
    const BwdLayout L = bwd_layout(P, flags);
    LK_REQUIRE(d->bwd_scratch_cap == 0 || d->bwd_scratch_cap >= L.total, "lk_render_bwd: bwd_scratch is smaller than lk_render_bwd_scratch_floats(R, S, flags) for the flags of this call");
    float* S0 = d->bwd_scratch;

    SideStream& ss = side_stream();
    LkFeatScatterArgs fs;
    seg_args(d, P, fs);
    if (gf && !(skip & LK_SEG_SORTED)) {
        const int rc2 = seg_sort_async(d, P, false, st);
        if (rc2 != LK_OK) return rc2;
    }

    LkCompositeBwdArgs cb;
    cb.R = d->R; cb.S = d->S; cb.min_nn = d->min_nn; cb.coef = d->coef;
    cb.raw = d->raw; cb.z = d->z; cb.nbr_count = d->nbr_count; cb.gt_depth = d->gt_depth;
    cb.d_depth = d->d_depth; cb.d_var = d->d_var; cb.d_color = color ? d->d_color : nullptr;
    cb.d_raw = S0 + L.d_raw; cb.keep_depth = (flags & LK_FLAG_Z_GIVEN) ? 1 : 0;
    // the composite backward as the prologue of the decoder backward (every lane: its own sample) instead of a launch in front of it;
    // k_geo_wgrad reads the d raw array (LK_FLAG_GRAD_GEO_DECODER: the launch stays)
    const bool cb_inline = !(skip & LK_SKIP_COMPOSITE_BWD) && !(gw && (flags & LK_FLAG_GRAD_GEO_DECODER));
    if (!(skip & LK_SKIP_COMPOSITE_BWD) && !cb_inline) lk_launch_composite_bwd(cb, st);

    // tracker-sized batches: rel-pos backward + interpolation backward in one launch (k_relpos_interp_bwd)
    const bool fuse_small = (skip & LK_FUSE_SMALL) && relpos && gr && !gw && !gf && lk_cdiv(P, 32) <= LK_DEEP_MAX_TILES;
    bool fuse_rb = false;
    LkRelposBwdArgs rb_fused;
    LkDecodeBwdArgs db;
    db.R = d->R; db.S = d->S; db.P = P; db.flags = flags;
    db.rays_o = d->rays_o; db.rays_d = d->rays_d; db.z = d->z;
    db.W = d->weights; db.Wfrag = d->weights_frag; db.affine = d->affine;
    db.act = d->act; db.raw = d->raw; db.d_raw = S0 + L.d_raw;
    db.dc_geo = S0 + L.dc_geo; db.dc_col = S0 + L.dc_col; db.dy_col = S0 + L.dy_col; db.dlogit = S0 + L.dlogit;
    db.dp_embed = S0 + L.dp_embed; db.dp_embed_col = S0 + L.dp_embed_col; db.g_weights = d->g_weights; db.g_affine = d->g_affine; db.g_affine_part = S0 + L.aff_part; db.part_bg = S0 + L.part_bg;
    db.live_rays = ex ? ex->live_rays : nullptr;
    db.dscale = ex ? ex->dscale : nullptr;
    db.tl_n_part = 0;
    memset(&db.tl, 0, sizeof(db.tl));
    db.cb_on = cb_inline ? 1 : 0; db.cb = cb;
    db.ml_on = 0; db.ml_row_part = nullptr;
    memset(&db.ml, 0, sizeof(db.ml));
    if ((skip & LK_COMPOSITE_IN_BWD) && (flags & LK_FLAG_MAPPER_LOSS)) {      // as the composite launch of lk_render_fwd_impl would have been set up
        LK_REQUIRE(ex && ex->loss_rows && d->d_depth && d->d_color && d->loss_gt_color && (skip & LK_SKIP_COMPOSITE_BWD),
                   "lk_render_bwd: the composite inside the decoder backward needs loss_rows, loss_gt_color, d_depth, d_color");
        LkCompositeArgs& ca = db.ml;
        ca.R = d->R; ca.S = d->S; ca.min_nn = d->min_nn; ca.coef = d->coef;
        ca.raw = d->raw; ca.z = d->z; ca.nbr_count = d->nbr_count; ca.gt_depth = d->gt_depth;
        ca.depth = d->depth; ca.var = d->var; ca.color = d->color; ca.valid_ray = d->valid_ray;
        ca.keep_depth = (flags & LK_FLAG_Z_GIVEN) ? 1 : 0;
        ca.gt_color = d->loss_gt_color; ca.w_color = d->loss_w_color; ca.use_color = color ? 1 : 0;
        ca.d_depth = const_cast<float*>(d->d_depth); ca.d_color = const_cast<float*>(d->d_color);
        db.ml_on = 1; db.ml_row_part = ex->loss_rows;
    }
    if (ex && ex->track_loss) {      // tracking loop: the loss and the composite backward are the launch's prologue (no d_raw array)
        LK_REQUIRE((flags & LK_FLAG_GRAD_RAYS) != 0 && ex->track_n_part > 0, "lk_render_bwd: the inline tracker loss needs ray gradients");
        db.tl = *ex->track_loss; db.tl_n_part = ex->track_n_part;
    }
    lk_launch_decode_bwd(db, st);
    if (gw && (flags & LK_FLAG_GRAD_GEO_DECODER)) {      // mapping.fix_geo_decoder: False - the geometry decoder's own matrices and biases
        LK_REQUIRE(!(flags & LK_FLAG_EMBED_GRADS_ONLY), "lk_render_bwd: GRAD_GEO_DECODER does not combine with EMBED_GRADS_ONLY");
        rc = lk_launch_geo_wgrad(P, d->S, d->rays_o, d->rays_d, d->z, d->weights, d->act, d->c_geo, S0 + L.d_raw, S0 + L.geo_part, d->g_weights, st,
                                 ex ? ex->live_rays : nullptr);
        if (rc != LK_OK) return rc;
    }
    // d affine: the colour tiles stored their 12 sums, one small launch adds them into g_affine (49 adds per address instead of 782)
    if (color && d->affine && d->g_affine && !(skip & LK_SKIP_AFF_REDUCE)) lk_launch_reduce_partials(S0 + L.aff_part, lk_cdiv(P, 32), 12, d->g_affine, st);
    // mapper 'color' backward with one weight-gradient launch: every partial-sum reduction is deferred to ONE launch at the end
    // gwf: the colour decoder's / rel-pos MLP's matrices want gradients too (not only the Fourier matrices, LK_FLAG_EMBED_GRADS_ONLY)
    const bool gwf = gw && !(flags & LK_FLAG_EMBED_GRADS_ONLY);
    const bool defer = gwf && color && (!relpos || lk_relpos_fused(flags));
    LkWgradArgs wdef;
    wdef.n_units = 0; wdef.part = nullptr;
    // the Fourier-matrix partials of k_decode_bwd: summed by a rider of the gather launch when there is one, else by their own launch
    const bool bg_rides = gw && !defer && gf;
    if (gw && !defer && !bg_rides) lk_launch_reduce_partials(S0 + L.part_bg, lk_cdiv(lk_cdiv(P, 32), 4), 288, d->g_weights + G_EB, st);

    const bool dw2_rides = gf && gwf && relpos && lk_relpos_fused(flags);
    const bool forked = gwf && color && ss.ok;
    // split step (LkBwdExtra::split_reduce): see the struct; needs the fork, the deferred reduction and the step rider
    const bool split = forked && defer && ex && ex->split_reduce && ex->step && ex->step->n_span > 0 && ex->split_frag && ex->split_master;
    if (ex && ex->split_done) *ex->split_done = split ? 1 : 0;
    hipStream_t wst = st;                      // stream of the weight-gradient launches
    if (forked) {
        (void)hipEventRecord(ss.fork, st);
        (void)hipStreamWaitEvent(ss.st, ss.fork, 0);
        wst = ss.st;
    }
    if (gwf && color) {
        // colour decoder weight gradients as streamed reductions over the saved rows (geometry decoder weights
        // other than embedder._B are frozen in every reference config: mapping.fix_geo_decoder = True)
        const float* act_h = d->act + (size_t)P * (LK_ACT_GEO_A + LK_ACT_COL_A);
        const float* act_e = d->act + (size_t)P * (LK_ACT_GEO_A + LK_ACT_COL_A + LK_ACT_COL_H);
        const float* dy = S0 + L.dy_col;           // d y_i = d h_i * softplus'(a_i), [layer][P][128]
        float* G = d->g_weights;
        LkWgradArgs wa;
        memset(&wa, 0, sizeof(wa));
        int nj = 0;
        const int w_off[5] = {C_W0, C_W1, C_W2, C_W3, C_W4}, b_off[5] = {C_B0, C_B1, C_B2, C_B3, C_B4};
        const int w_ld[5] = {EC, HC, HC, EC + HC, HC};
        // c as auxiliary columns of job `src`: M = A^T c and the bias sums finish fc_c layer `layer` (LkFcPost, lk_kernels.h)
        auto fc_from = [&](LkWgradJob& J, int src, int layer, const float* Wsrc, int rows_u, int ldw, int off) {
            J.k_aux = J.K;
            if (J.B2) { J.B3 = d->c_col; J.ldb3 = LK_C; J.k_split2 = J.K; }
            else { J.B2 = d->c_col; J.ldb2 = LK_C; J.k_split = J.K; }
            J.K += CF;
            LkFcPost& F = wa.fc[wa.n_fc++];
            F.src_job = src; F.rows_u = rows_u; F.ldw = ldw; F.off = off; F.W = Wsrc;
            F.dU = G + C_U0 + layer * C_USTRIDE; F.du = F.dU + a64(HC * CF);
        };
        for (int i = 0; i < 5; ++i) {
            LkWgradJob& J = wa.job[nj++];
            J.A = dy + LK_COL_LAYER(P, i); J.lda = 128; J.a_mode = 0;
            if (i == 0) { J.B = act_e; J.ldb = LK_ACT_COL_E; }
            else if (i == 3) { J.B = act_e; J.ldb = LK_ACT_COL_E; J.B2 = act_h + LK_COL_LAYER(P, 2); J.ldb2 = 128; J.k_split = EC; }
            else { J.B = act_h + LK_COL_LAYER(P, i - 1); J.ldb = 128; }
            J.N = HC; J.K = w_ld[i]; J.rows = P; J.dW = G + w_off[i]; J.ldw = w_ld[i]; J.db = G + b_off[i];
            // d h_{i-1} = (hidden columns of W_i)^T d y_i
            if (i >= 1) fc_from(J, nj - 1, i - 1, d->weights + w_off[i], HC, w_ld[i], i == 3 ? EC : 0);
        }
        {
            LkWgradJob& J = wa.job[nj++];
            J.A = S0 + L.dlogit; J.lda = 4; J.a_mode = 0;
            J.B = act_h + LK_COL_LAYER(P, 4); J.ldb = 128;
            J.N = 3; J.K = HC; J.rows = P; J.dW = G + C_WO; J.ldw = HC; J.db = G + C_BO;
            fc_from(J, nj - 1, 4, d->weights + C_WO, 3, HC, 0);      // d h_4 = Wo^T d out
        }
        wa.n_jobs = nj; wa.chunk = 0; wa.part = S0 + L.wg_part;
        wa.h16 = (flags & LK_FLAG_UNIT_LOSS_GRADS) && !gr ? 1 : 0;
        wa.live_rays = ex ? ex->live_rays : nullptr; wa.S = d->S; wa.dscale = ex ? ex->dscale : nullptr;
        if (forked && g_side_delay_us > 0) hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(64), 0, wst, (long long)g_side_delay_us * 100);   // wall_clock64: 100 MHz
        lk_launch_wgrad(wa, P, wst, defer ? &wdef : nullptr);
        if (split) {
            // the trunk's half of the step, right behind its weight gradients on their stream: tile sums + fc_c products with the Adam rider
            // (new values -> w_next), then the stepped trunk over the master blob and its fragments.  Everything it reads was written before
            // the fork or by k_wgrad; everything it writes (trunk spans of w_next / master / Adam state / g_weights, trunk fragments) is read
            // again only behind the join, which the NEXT iteration's forward places in front of its decoder launch.
            LkBwdReduceArgs rt;
            memset(&rt, 0, sizeof(rt));
            lk_launch_bwd_reduce(wdef, rt, false, wst, ex->step, false);
            lk_launch_repack_trunk(ex->step->w_next, ex->split_master, ex->split_frag, wst);
        }
    }

    if (relpos) {
        LkRelposBwdArgs rb;
        rb.R = d->R; rb.S = d->S; rb.P = P; rb.min_nn = d->min_nn; rb.flags = flags;
        rb.rays_o = d->rays_o; rb.rays_d = d->rays_d; rb.z = d->z; rb.pos = d->pos; rb.col_feats = d->col_feats;
        rb.nbr_idx = d->nbr_idx; rb.nbr_w = d->nbr_w; rb.nbr_count = d->nbr_count;
        rb.W = d->weights; rb.Wfrag = d->weights_frag; rb.dc_col = S0 + L.dc_col;
        rb.g_col_feats = d->g_col_feats; rb.g_weights = d->g_weights;
        rb.dw_rel = S0 + L.dw_rel; rb.dp_rel = S0 + L.dp_rel; rb.rows = S0 + L.rows; rb.w_eff = S0 + L.w_eff;
        rb.dfeat = S0 + L.dfeat;
        rb.part_br = S0 + L.part_br; rb.hbar = S0 + L.hbar; rb.w_sum = S0 + L.w_sum; rb.dw1_part = S0 + L.dw1_part;
        rb.live_rays = (ex && lk_relpos_fused(flags)) ? ex->live_rays : nullptr;
        fuse_rb = fuse_small;                   // launched together with the interpolation backward below
        if (fuse_rb) rb_fused = rb;
        else lk_launch_relpos_bwd(rb, st);
        // the weight-gradient stream waits for the rel-pos backward only where it consumes its rows (the unfused linear1 / linear2 jobs below): in
        // the fused form nothing on that stream depends on it, and the event record on the caller's stream is a 6-7 us bubble between the rel-pos
        // backward and the gather of every 'color' iteration (tools/trace_window.py)
        if (forked && !lk_relpos_fused(flags)) { (void)hipEventRecord(ss.mid, st); (void)hipStreamWaitEvent(wst, ss.mid, 0); }
        // (the fused kernel - kept in the embedding-only mode, where its linear1 tiles go unused - leaves one partial row per workgroup)
        if (gw && !defer) lk_launch_reduce_partials(S0 + L.part_br, lk_relpos_fused(flags) ? lk_relpos_bwd_parts(P) : lk_cdiv(lk_cdiv(P, 4), 4), 32,
                                                    d->g_weights + R_EB, st);
    }

    if (gf) {
        fs.dc_geo = S0 + L.dc_geo; fs.dc_col = (color && !relpos) ? S0 + L.dc_col : nullptr; fs.dfeat = relpos ? S0 + L.dfeat : nullptr;
        fs.g_geo_feats = d->g_geo_feats; fs.g_col_feats = d->g_col_feats;
        if (ex && ex->seg_list) { fs.seg_list = ex->seg_list; fs.seg_total = ex->seg_total; }      // sorted ahead of the loop (lk_map_frame)
        else if (ss.ok) (void)hipStreamWaitEvent(st, ss.link, 0);
        fs.act_flag = ex ? ex->act_flag : nullptr;
        if (bg_rides) { fs.red_part = S0 + L.part_bg; fs.red_n = lk_cdiv(lk_cdiv(P, 32), 4); fs.red_width = 288; fs.red_out = d->g_weights + G_EB; }
        if (dw2_rides) {       // linear2 of the rel-pos MLP: k_dw2_hbar's blocks in front of the gather's (without feature gradients: a launch of its own, below)
            fs.dw2_dc = S0 + L.dc_col; fs.dw2_w_sum = S0 + L.w_sum; fs.dw2_hbar = S0 + L.hbar; fs.dw2_part = S0 + L.dw2_part;
            fs.dw2_blocks = lk_dw2_parts(P); fs.dw2_live = ex ? ex->live_rays : nullptr; fs.dw2_S = d->S;
        }
        if (ex && ex->xstep) { fs.x_on = 1; fs.x = *ex->xstep; }
        lk_launch_feat_scatter(fs, st);
        // data-parallel caller: the feature-row gradients are final from here (lk_map_desc::signal_rows)
        if (ex && ex->signal_rows) {
            const int rc3 = rows_event_record(st);
            if (rc3 != LK_OK) return rc3;
        }
    }
    if (gr) {
        LkInterpBwdArgs ib;
        ib.R = d->R; ib.S = d->S; ib.P = P; ib.min_nn = d->min_nn; ib.flags = flags;
        ib.rays_o = d->rays_o; ib.rays_d = d->rays_d; ib.z = d->z; ib.r2_ray = d->r2_ray; ib.r2_static = d->r2_static;
        ib.pos = d->pos; ib.geo_feats = d->geo_feats; ib.col_feats = d->col_feats;
        ib.nbr_idx = d->nbr_idx; ib.nbr_w = d->nbr_w; ib.nbr_count = d->nbr_count;
        ib.dc_geo = S0 + L.dc_geo; ib.dc_col = S0 + L.dc_col;
        ib.dw_rel = relpos ? S0 + L.dw_rel : nullptr;
        ib.dp_embed = S0 + L.dp_embed; ib.dp_embed_col = color ? S0 + L.dp_embed_col : nullptr; ib.dp_rel = relpos ? S0 + L.dp_rel : nullptr;
        ib.g_geo_feats = d->g_geo_feats; ib.g_col_feats = d->g_col_feats; ib.dp_total = S0 + L.dp_total;
        ib.pose_part = ex ? ex->pose_part : nullptr;
        if (ex) { ib.pix_i = ex->pix_i; ib.pix_j = ex->pix_j; ib.fx = ex->fx; ib.fy = ex->fy; ib.cx = ex->cx; ib.cy = ex->cy; }
        else { ib.pix_i = ib.pix_j = nullptr; ib.fx = ib.fy = 1.0f; ib.cx = ib.cy = 0.0f; }
        if (fuse_rb) lk_launch_relpos_interp_bwd(rb_fused, ib, st, (flags & LK_FLAG_UNIT_LOSS_GRADS) && d->affine == nullptr);
        else lk_launch_interp_bwd(ib, st, (ex && !gf) ? ex->xstep : nullptr, ex ? ex->xstep_part : nullptr, ex ? ex->xstep_n_part : 0);
    }
    if (gr && !(skip & LK_SKIP_RAYS_BWD)) {
        LkRaysBwdArgs rr;
        rr.R = d->R; rr.S = d->S; rr.z = d->z; rr.dp_total = S0 + L.dp_total; rr.g_rays_o = d->g_rays_o; rr.g_rays_d = d->g_rays_d;
        lk_launch_rays_bwd(rr, st);
    }
    if (gwf && relpos) {
        float* G = d->g_weights;
        if (lk_relpos_fused(flags)) {
            // linear1 was reduced inside k_relpos_bwd_fused (workgroup tiles), linear2 = samples x (wsum d c) (x) Hbar: two small launches
            LkRelposBwdArgs rb;
            memset(&rb, 0, sizeof(rb));
            rb.P = P; rb.S = d->S; rb.dc_col = S0 + L.dc_col; rb.w_sum = S0 + L.w_sum; rb.hbar = S0 + L.hbar; rb.dw1_part = S0 + L.dw1_part;
            rb.live_rays = ex ? ex->live_rays : nullptr;
            // on the caller's stream: after the fused kernel and the gather it has room, the weight-gradient stream is the longer one
            // (on the third stream beside the gather: no gain, 399 -> 409 us per colour iteration)
            if (!dw2_rides) lk_launch_dw2_hbar(rb, S0 + L.dw2_part, st);
        } else {
            LkWgradArgs wr;
            memset(&wr, 0, sizeof(wr));
            const float* rows = S0 + L.rows;
            LkWgradJob& J1 = wr.job[0];       // linear1 [128][52]: rows = neighbour rows, A = d hid, B = x
            J1.A = rows; J1.lda = 192; J1.a_mode = 0; J1.B = rows + 128; J1.ldb = 192;
            J1.N = HC; J1.K = KR; J1.rows = 8 * P; J1.dW = G + R_W1; J1.ldw = KRP; J1.db = G + R_B1;
            LkWgradJob& J2 = wr.job[1];       // linear2 [32][128]: rows = SAMPLES, A = (sum_j w_j) * d c, B = sum_j w_j hid_j
            J2.A = S0 + L.dc_col; J2.lda = LK_C; J2.a_mode = 2; J2.A2 = S0 + L.w_sum; J2.lda2 = 1;
            J2.B = S0 + L.hbar; J2.ldb = 128;
            J2.N = CF; J2.K = HC; J2.rows = P; J2.dW = G + R_W2; J2.ldw = HC; J2.db = G + R_B2;
            wr.n_jobs = 2; wr.chunk = 0; wr.part = S0 + L.wg_part;
            lk_launch_wgrad(wr, 8 * P, wst);
        }
    }
    if (forked) {
        // (since the split step the join orders the NEXT call's decoder behind this call's trunk step: its failure is an error, not a lost overlap)
        LK_HIP_TRY(hipEventRecord(ss.join, wst));
        if (!split) LK_HIP_TRY(hipStreamWaitEvent(st, ss.join, 0));
    }
    if (defer) {
        float* G = d->g_weights;
        LkBwdReduceArgs r;
        memset(&r, 0, sizeof(r));
        const bool with_rp = relpos;          // (defer && relpos) == the fused variant
        if (with_rp) {
            r.part1 = S0 + L.dw1_part; r.n1 = lk_relpos_bwd_parts(P); r.part2 = S0 + L.dw2_part; r.n2 = lk_dw2_parts(P);
            r.dW1 = G + R_W1; r.db1 = G + R_B1; r.dW2 = G + R_W2; r.db2 = G + R_B2;
            r.part_br = S0 + L.part_br; r.n_br = lk_relpos_bwd_parts(P); r.out_br = G + R_EB;
        }
        r.part_bg = S0 + L.part_bg; r.n_bg = lk_cdiv(lk_cdiv(P, 32), 4); r.out_bg = G + G_EB;
        if (split) {        // the launch stream's half: what its own kernels left (rel-pos tiles, Fourier partials) + the feature rows
            LkWgradArgs none;
            memset(&none, 0, sizeof(none));
            lk_launch_bwd_reduce(none, r, with_rp, st, ex->step, true);
        } else lk_launch_bwd_reduce(wdef, r, with_rp, st, ex ? ex->step : nullptr);
    }
    LK_LAUNCH_CHECK();
    return LK_OK;
}

// ------------------------------------------------------------------ building block: one weight-gradient reduction
extern "C" int lk_wgrad_single(const float* A, int32_t lda, int32_t a_mode, const float* A2, int32_t lda2,
                               const float* B, int32_t ldb, int32_t N, int32_t K, int64_t rows,
                               float* dW, int32_t ldw, float* db, int32_t chunk, void* stream_) {
    LK_REQUIRE(A && B && dW && N > 0 && N <= 128 && K > 0 && K <= 192 && rows >= 0 && chunk >= 32, "lk_wgrad_single: bad arguments");
    LK_REQUIRE((N % 4 == 0 || lda >= ((N + 3) / 4) * 4) && K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0, "lk_wgrad_single: rows must be float4-addressable");
    if (rows == 0) return LK_OK;
    LkWgradArgs wa;
    memset(&wa, 0, sizeof(wa));
    LkWgradJob& J = wa.job[0];
    J.A = A; J.lda = lda; J.a_mode = a_mode; J.A2 = A2; J.lda2 = lda2; J.B = B; J.ldb = ldb;
    J.N = N; J.K = K; J.rows = (int)rows; J.dW = dW; J.ldw = ldw; J.db = db;
    wa.n_jobs = 1; wa.chunk = chunk;
    lk_launch_wgrad(wa, (int)rows, (hipStream_t)stream_);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

// ------------------------------------------------------------------ per-kernel timing (HIP events on the launch stream)
#include <vector>
#include <string>
namespace {
const char* kKernelNames[LKK_COUNT] = {"k_depth_stats", "k_sample_interp", "k_relpos_fwd", "k_decode_fwd", "k_composite",
                                       "k_composite_bwd", "k_decode_bwd", "k_relpos_bwd", "k_interp_bwd", "k_rays_bwd", "k_wgrad", "k_feat_gather", "k_decode_bwd_track"};
struct ProfState {
    bool on = false;
    bool enabled[LKK_COUNT] = {};
    std::vector<hipEvent_t> ev0[LKK_COUNT], ev1[LKK_COUNT];
};
ProfState g_prof;
}  // namespace

void lk_prof_before(int kid, hipStream_t st) {
    if (!g_prof.on || !g_prof.enabled[kid]) return;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return;
    g_prof.ev0[kid].push_back(e0);
    g_prof.ev1[kid].push_back(e1);
    (void)hipEventRecord(e0, st);
}
void lk_prof_after(int kid, hipStream_t st) {
    if (!g_prof.on || !g_prof.enabled[kid] || g_prof.ev1[kid].empty()) return;
    (void)hipEventRecord(g_prof.ev1[kid].back(), st);
}

/* names: comma-separated kernel names, or "*" for all */
extern "C" int lk_profile_begin(const char* names) {
    LK_REQUIRE(names != nullptr, "lk_profile_begin: NULL names");
    const std::string s(names);
    for (int k = 0; k < LKK_COUNT; ++k) {
        g_prof.enabled[k] = (s == "*") || (("," + s + ",").find(std::string(",") + kKernelNames[k] + ",") != std::string::npos);
        g_prof.ev0[k].clear();
        g_prof.ev1[k].clear();
    }
    g_prof.on = true;
    return LK_OK;
}

/* Synchronises the recorded events; writes "name calls total_ms\n" lines into buf. */
extern "C" int lk_profile_end(char* buf, int cap) {
    g_prof.on = false;
    std::string out;
    for (int k = 0; k < LKK_COUNT; ++k) {
        double total = 0.0;
        const size_t n = g_prof.ev0[k].size();
        for (size_t i = 0; i < n; ++i) {
            float ms = 0.0f;
            (void)hipEventSynchronize(g_prof.ev1[k][i]);
            if (hipEventElapsedTime(&ms, g_prof.ev0[k][i], g_prof.ev1[k][i]) == hipSuccess) total += ms;
            (void)hipEventDestroy(g_prof.ev0[k][i]);
            (void)hipEventDestroy(g_prof.ev1[k][i]);
        }
        g_prof.ev0[k].clear();
        g_prof.ev1[k].clear();
        if (n) {
            char line[160];
            snprintf(line, sizeof(line), "%s %zu %.6f\n", kKernelNames[k], n, total);
            out += line;
        }
    }
    if (buf && cap > 0) {
        strncpy(buf, out.c_str(), (size_t)cap - 1);
        buf[cap - 1] = 0;
    }
    return LK_OK;
}
