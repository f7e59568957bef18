// Fragment repack of the decoder weights (see lk_weights.h): plain master blob -> MFMA-operand-
// ordered copies (forward and transposed) that the render kernels stream with fully coalesced
// 16-byte loads.  Runs once per optimiser step (108k parameters; a few microseconds).
#include "lk_common.h"

using namespace lkw;

struct FragTable { FragMat m[N_FRAG_MATS]; };

__global__ __launch_bounds__(256) void k_weights_repack(const float* __restrict__ plain, float* __restrict__ frag,
                                                        FragTable tb) {
    for (int idx = blockIdx.x * 256 + (int)threadIdx.x; idx < FRAG_FLOATS; idx += gridDim.x * 256) {
        int mi = 0;
#pragma unroll 1
        for (int q = 1; q < N_FRAG_MATS; ++q) mi = (idx >= tb.m[q].fwd) ? q : mi;
        const FragMat M = tb.m[mi];
        float val = 0.0f;
        if (idx < M.tr) {                    // forward fragments, kg-major
            const int o = idx - M.fwd;
            const int blk = o >> 8, l = (o & 255) >> 2, t = o & 3;
            const int NBT = M.rows >> 5;
            const int kg = blk / NBT, nb = blk - kg * NBT;
            const int row = nb * 32 + (l & 31), col = 8 * kg + 4 * (l >> 5) + t;
            val = plain[M.plain + row * M.ld + col];
        } else {                             // transposed fragments, ng-major, virtual input columns
            const int o = idx - M.tr;
            const int blk = o >> 8, l = (o & 255) >> 2, t = o & 3;
            const int KB = M.kv >> 5;
            const int ng = blk / KB, kb = blk - ng * KB;
            const int n = 8 * ng + 4 * (l >> 5) + t;
            const int v = kb * 32 + (l & 31);
            int col = -1;
            if (v < M.e_real) col = v;
            else if (v >= M.e_virt) col = v - (M.e_virt - M.e_real);
            if (col >= 0 && col < M.ld) val = plain[M.plain + n * M.ld + col];
        }
        frag[idx] = val;
    }
}

extern "C" int64_t lk_weight_frag_floats(void) { return FRAG_FLOATS; }

extern "C" int lk_weights_repack(const float* plain, float* frag, void* stream_) {
    LK_REQUIRE(plain && frag, "lk_weights_repack: NULL buffer");
    static const FragMat rows[N_FRAG_MATS] = {LKW_FRAG_TABLE};
    FragTable tb;
    for (int i = 0; i < N_FRAG_MATS; ++i) tb.m[i] = rows[i];
    hipLaunchKernelGGL(k_weights_repack, dim3(lk_cdiv(FRAG_FLOATS, 256)), dim3(256), 0, (hipStream_t)stream_, plain, frag, tb);
    LK_LAUNCH_CHECK();
    return LK_OK;
}
