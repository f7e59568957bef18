// Probe (not product code): how many wait states does a VALU read of an MFMA result need, alone and with other waves of
// the SIMD keeping the matrix pipe busy?  N dependent v_mfma_f32_32x32x16_bf16, then `s_nop` x WAIT, then the compiler's
// own code reads the accumulator (it cannot see the MFMAs inside the asm block, so it adds no wait states of its own).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NM, int WAIT>
__global__ __launch_bounds__(256) void k_raw(float* out, int iters) {
    const unsigned one = 0x3f803f80u;
    u32x4 A = {one, one, one, one}, B = {one, one, one, one};
    float bad = 0.f, total = 0.f;
    for (int it = 0; it < iters; ++it) {
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        asm volatile(
            ".rept %[nm]\n"
            "v_mfma_f32_32x32x16_bf16 %[acc], %[a], %[b], %[acc]\n"
            ".endr\n"
            ".rept %[wt]\n"
            "s_nop 0\n"
            ".endr\n"
            : [acc] "+v"(acc)
            : [a] "v"(A), [b] "v"(B), [nm] "n"(NM), [wt] "n"(WAIT));
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += acc[r];
        bad += (s != 16.0f * 16.0f * NM) ? 1.f : 0.f;
        total += s;
    }
    if (bad != 0.f) atomicAdd(out, bad);
    if (total == 12345.f) out[1] = 1.f;
}

template <int NM, int WAIT>
static void run(float* d, int blocks, int iters) {
    CK(hipMemset(d, 0, 8));
    hipLaunchKernelGGL((k_raw<NM, WAIT>), dim3(blocks), dim3(256), 0, 0, d, iters);
    CK(hipDeviceSynchronize());
    float h[2];
    CK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
    printf("  %d MFMAs + %2d wait states: wrong reads %8.0f of %d", NM, WAIT, h[0], blocks * 256 * iters);
    printf("\n");
}

int main() {
    float* d;
    CK(hipMalloc(&d, 8));
    const int iters = 500;
    for (int wps = 1; wps <= 3; ++wps) {
        const int blocks = 256 * wps;
        printf("%d waves per SIMD\n", wps);
        run<6, 0>(d, blocks, iters);
        run<6, 4>(d, blocks, iters);
        run<6, 8>(d, blocks, iters);
        run<6, 12>(d, blocks, iters);
        run<6, 16>(d, blocks, iters);
        run<6, 24>(d, blocks, iters);
        run<6, 48>(d, blocks, iters);
        run<1, 12>(d, blocks, iters);
    }
    return 0;
}
