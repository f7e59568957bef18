// Fragment repack of the decoder weights (see lk_weights.h): plain master blob -> MFMA-operand-ordered split-bf16 copies
// (forward and transposed) that the render kernels stream with fully coalesced 16-byte loads.  Runs once per optimiser
// step (108k parameters; a few microseconds).
#include "lk_weights_dev.h"

__global__ __launch_bounds__(256) void k_weights_repack(const float* __restrict__ plain, u32x4* __restrict__ fragb, FragTable tb) {
    for (int u = blockIdx.x * 256 + (int)threadIdx.x; u < LK_REPACK_UNITS; u += gridDim.x * 256) repack_unit(plain, fragb, tb, u);
}

// The colour trunk's half of a split step (lk_render_bwd_impl: LkBwdExtra::split_reduce): the stepped values of the blob range [lo, hi) over the
// master blob, and the fragments of the trunk's matrices from them - on the weight-gradient stream, behind the trunk's reduction + Adam.
__global__ __launch_bounds__(256) void k_repack_trunk(const float* __restrict__ src, float* __restrict__ dst, int lo, int hi, u32x4* __restrict__ fragb,
                                                      FragTable tb, int copy_block0) {
    if ((int)blockIdx.x >= copy_block0) {
        const int i = lo + (((int)blockIdx.x - copy_block0) * 256 + (int)threadIdx.x) * 4;      // lo, hi: multiples of 64 floats
        if (i + 3 < hi) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
        return;
    }
    const int u = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (u < LK_REPACK_UNITS) repack_unit(src, fragb, tb, u, LK_FRAG_COL_LO, LK_FRAG_COL_HI);
}
int lk_launch_repack_trunk(const float* src, float* dst, float* frag, hipStream_t st) {
    static_assert(C_EB % 4 == 0 && R_EB % 4 == 0, "trunk range must be float4-addressable");
    const int b0 = lk_cdiv(LK_REPACK_UNITS, 256);
    hipLaunchKernelGGL(k_repack_trunk, dim3(b0 + lk_cdiv(R_EB - C_EB, 1024)), dim3(256), 0, st, src, dst, (int)C_EB, (int)R_EB,
                       reinterpret_cast<u32x4*>(frag), lk_frag_table(), b0);
    return LK_OK;
}

// floats of the fragment buffer the caller allocates (opaque derived data: 3 x 16 B per lane-block element)
extern "C" int64_t lk_weight_frag_floats(void) { return 4 * ((int64_t)FRAGB_U4 + FRAGH_U4); }

extern "C" int lk_weights_repack(const float* plain, float* frag, void* stream_) {
    LK_REQUIRE(plain && frag, "lk_weights_repack: NULL buffer");
    { const int rcg = lk_status_gate("lk_weights_repack"); if (rcg != LK_OK) return rcg; }
    const FragTable tb = lk_frag_table();
    hipLaunchKernelGGL(k_weights_repack, dim3(lk_cdiv(LK_REPACK_UNITS, 256)), dim3(256), 0, (hipStream_t)stream_, plain,
                       reinterpret_cast<u32x4*>(frag), tb);
    LK_LAUNCH_CHECK();
    return LK_OK;
}

extern "C" int lk_weights_repack_checked(const float* plain, float* frag, void* stream_) {
    const int rc = lk_weights_repack(plain, frag, stream_);
    if (rc != LK_OK) return rc;
    if (!lk_status_dev()) return LK_OK;          // no status word (allocation failed): nothing to report
    uint32_t bits = 0;
    const int rc2 = lk_status_sync(stream_, &bits);
    if (rc2 != LK_OK) return rc2;
    return lk_status_gate("lk_weights_repack_checked");
}
