// extern "C" entry points of libloopyhip.so (see include/loopy_hip.h).
#include "lk_common.h"
#include "lk_kernels.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

using namespace lkw;

static thread_local char g_err[512] = "";

void lk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int lk_version(void) { return LK_ABI_VERSION; }
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

extern "C" int lk_render_fwd(const lk_render_desc* d, void* stream_) {
    int rc = check_desc(d, "lk_render_fwd");
    if (rc != LK_OK) return rc;
    if (d->R == 0) return LK_OK;
    LK_REQUIRE(d->depth && d->var && d->color && d->valid_ray, "lk_render_fwd: NULL output buffer");
    hipStream_t st = (hipStream_t)stream_;
    const int P = d->R * d->S;
    const bool all_pos = (d->flags & LK_FLAG_ALL_DEPTH_POS) != 0;
    if (!all_pos) {
        LK_REQUIRE(d->far_stats != nullptr, "lk_render_fwd: far_stats required unless ALL_DEPTH_POS");
        lk_launch_depth_stats(d->gt_depth, d->R, d->stats_chunk, d->far_stats, st);
    }
    LkSampleArgs sa;
    sa.R = d->R; sa.S = d->S; sa.P = P; sa.stats_chunk = d->stats_chunk; sa.flags = d->flags;
    sa.rays_o = d->rays_o; sa.rays_d = d->rays_d; sa.gt_depth = d->gt_depth; sa.r2_ray = d->r2_ray;
    sa.far_stats = all_pos ? nullptr : d->far_stats;
    sa.grid = d->knn->grid; sa.sorted = d->knn->sorted; sa.cell_start = d->knn->cell_start;
    sa.geo_feats = d->geo_feats; sa.col_feats = d->col_feats; sa.noise_geo = d->noise_geo; sa.noise_col = d->noise_col;
    sa.near_surface = d->near_surface; sa.far_surface = d->far_surface; sa.near_end = d->near_end; sa.r2_static = d->r2_static;
    sa.min_nn = d->min_nn;
    sa.z = d->z; sa.nbr_idx = d->nbr_idx; sa.nbr_w = d->nbr_w; sa.nbr_count = d->nbr_count; sa.c_geo = d->c_geo; sa.c_col = d->c_col;
    lk_launch_sample_interp(sa, st);

    const bool color = (d->flags & LK_FLAG_STAGE_COLOR) != 0;
    if (color && (d->flags & LK_FLAG_REL_POS)) {
        LkRelposArgs ra;
        ra.R = d->R; ra.S = d->S; ra.P = P; ra.min_nn = d->min_nn;
        ra.rays_o = d->rays_o; ra.rays_d = d->rays_d; ra.z = d->z; ra.sorted_unused = nullptr;
        ra.pos = d->pos; ra.col_feats = d->col_feats;
        ra.nbr_idx = d->nbr_idx; ra.nbr_w = d->nbr_w; ra.nbr_count = d->nbr_count;
        ra.W = d->weights; ra.Wfrag = d->weights_frag; ra.noise_col = d->noise_col; ra.c_col = d->c_col;
        lk_launch_relpos_fwd(ra, st);
    }
    LkDecodeArgs da;
    da.R = d->R; da.S = d->S; da.P = P; da.flags = d->flags;
    da.rays_o = d->rays_o; da.rays_d = d->rays_d; da.z = d->z;
    da.c_geo = d->c_geo; da.c_col = d->c_col; da.W = d->weights; da.Wfrag = d->weights_frag; da.affine = d->affine;
    da.raw = d->raw; da.act = d->act;
    lk_launch_decode_fwd(da, st);

    LkCompositeArgs ca;
    ca.R = d->R; ca.S = d->S; ca.min_nn = d->min_nn; ca.coef = d->coef;
    ca.raw = d->raw; ca.z = d->z; ca.nbr_count = d->nbr_count; ca.gt_depth = d->gt_depth;
    ca.depth = d->depth; ca.var = d->var; ca.color = d->color; ca.valid_ray = d->valid_ray;
    lk_launch_composite(ca, st);
    LK_LAUNCH_CHECK();
    return LK_OK;
}
