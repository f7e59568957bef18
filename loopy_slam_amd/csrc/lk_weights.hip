// Fragment repack of the decoder weights (see lk_weights.h): plain master blob -> MFMA-operand-ordered split-bf16 copies
// (forward and transposed) that the render kernels stream with fully coalesced 16-byte loads.  Runs once per optimiser
// step (108k parameters; a few microseconds).
#include "lk_weights_dev.h"

__global__ __launch_bounds__(256) void k_weights_repack(const float* __restrict__ plain, u32x4* __restrict__ fragb, FragTable tb) {
    for (int u = blockIdx.x * 256 + (int)threadIdx.x; u < LK_REPACK_UNITS; u += gridDim.x * 256) repack_unit(plain, fragb, tb, u);
}

// floats of the fragment buffer the caller allocates (opaque derived data: 3 x 16 B per lane-block element)
extern "C" int64_t lk_weight_frag_floats(void) { return 4 * ((int64_t)FRAGB_U4 + FRAGH_U4); }

extern "C" int lk_weights_repack(const float* plain, float* frag, void* stream_) {
    LK_REQUIRE(plain && frag, "lk_weights_repack: NULL buffer");
    const FragTable tb = lk_frag_table();
    hipLaunchKernelGGL(k_weights_repack, dim3(lk_cdiv(LK_REPACK_UNITS, 256)), dim3(256), 0, (hipStream_t)stream_, plain,
                       reinterpret_cast<u32x4*>(frag), tb);
    LK_LAUNCH_CHECK();
    return LK_OK;
}
