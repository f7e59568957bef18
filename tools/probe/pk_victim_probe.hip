// Probe (not product code): are packed-fp32 VALU ops (v_pk_mul_f32 / v_pk_add_f32: what the SLP vectoriser makes of the
// d h = Wo^T d out expression of k_decode_bwd) safe when ANOTHER wave of the same SIMD issues matrix / transcendental / LDS /
// memory instructions in between?  The round-1 failure (DESIGN.md "store-data rule") needs two workgroups per compute unit,
// is limited to lanes 48-63 of one register, goes away with -fno-slp-vectorize and is NOT cured by wait states inside the
// wave (tools/probe/store_hazard_probe.hip modes 6-9 are clean; s_nop after every packed op still fails).
//
// Even workgroups are victims: the packed chain of the failing code next to its scalar twin, compared bit for bit in
// registers (no memory involved).  Odd workgroups are aggressors of one kind.  Mismatches are counted per 16-lane quarter.
//   hipcc --offload-arch=gfx950 -O2 -o pk_victim_probe pk_victim_probe.hip && ./pk_victim_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// AGG: 0 none (aggressor workgroups idle), 1 MFMA chains, 2 v_exp_f32 (trans unit), 3 LDS read/write, 4 global loads + stores,
//      5 the victim's own code (packed chains on both waves of the SIMD)
template <int AGG>
__global__ __launch_bounds__(256) void k_pk(unsigned long long* bad /*[4] per quarter*/, float* sink, int iters) {
    __shared__ u32x4 lds[256];
    const int lane = threadIdx.x & 63;
    const bool victim = (blockIdx.x & 1) == 0 || AGG == 5;
    const float base = 0.001f * (float)(threadIdx.x + 1) + (float)(blockIdx.x & 7);
    if (victim) {
        unsigned long long nbad = 0;
        for (int it = 0; it < iters; ++it) {
            const float a0 = base + it, a1 = base * 1.5f - it, b0 = 0.37f * base, b1 = 1.0f / (base + 1.0f), c0 = base - 2.0f, c1 = 3.0f - base, g0 = 0.11f + it, g1 = 0.013f * it;
            unsigned m0, m1;
            asm volatile(
                // inputs -> fixed registers
                "v_mov_b32 v100, %[a0]\n v_mov_b32 v101, %[a1]\n v_mov_b32 v104, %[b0]\n v_mov_b32 v105, %[b1]\n"
                "v_mov_b32 v106, %[c0]\n v_mov_b32 v107, %[c1]\n v_mov_b32 v102, %[g0]\n v_mov_b32 v103, %[g1]\n v_mov_b32 v108, %[g1]\n v_mov_b32 v109, %[g0]\n"
                "s_nop 4\n"
                // packed chain exactly as compiled in the failing build (op_sel broadcasts of g0 / g1)
                "v_pk_mul_f32 v[110:111], v[100:101], v[102:103] op_sel:[0,1]\n"
                "v_pk_mul_f32 v[112:113], v[104:105], v[102:103] op_sel_hi:[1,0]\n"
                "v_pk_mul_f32 v[118:119], v[106:107], v[108:109] op_sel_hi:[1,0]\n"
                "v_pk_add_f32 v[110:111], v[112:113], v[110:111]\n"
                "v_pk_add_f32 v[114:115], v[118:119], v[110:111]\n"
                "s_nop 7\n"
                // scalar twin
                "v_mul_f32 v120, v100, v103\n v_mul_f32 v121, v101, v103\n"
                "v_mul_f32 v122, v104, v102\n v_mul_f32 v123, v105, v102\n"
                "v_mul_f32 v124, v106, v108\n v_mul_f32 v125, v107, v108\n"
                "v_add_f32 v120, v122, v120\n v_add_f32 v121, v123, v121\n"
                "v_add_f32 v116, v124, v120\n v_add_f32 v117, v125, v121\n"
                "s_nop 4\n"
                "v_xor_b32 %[m0], v114, v116\n v_xor_b32 %[m1], v115, v117\n"
                : [m0] "=v"(m0), [m1] "=v"(m1)
                : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [c0] "v"(c0), [c1] "v"(c1), [g0] "v"(g0), [g1] "v"(g1)
                : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
                  "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125");
            nbad += (m0 != 0) + (m1 != 0);
        }
        if (nbad) atomicAdd(bad + (lane >> 4), nbad);
        return;
    }
    // ---- aggressors
    const unsigned one = 0x3f803f80u;
    u32x4 A = {one, one, one, one}, B = {one, one, one, one};
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float x = base, y = 0.f;
    lds[threadIdx.x] = A;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (AGG == 1) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n"
                         "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n"
                         : [c] "+v"(acc) : [a] "v"(A), [b] "v"(B));
        } else if (AGG == 2) {
            asm volatile("v_exp_f32 %[y], %[x]\n v_log_f32 %[x], %[y]\n v_exp_f32 %[y], %[x]\n v_log_f32 %[x], %[y]\n s_nop 1\n" : [x] "+v"(x), [y] "+v"(y));
        } else if (AGG == 3) {
            u32x4 t = lds[(threadIdx.x + it) & 255];
            lds[(threadIdx.x * 7 + it) & 255] = t;
            y += __uint_as_float(t[0]);
        } else if (AGG == 4) {
            const float v = sink[(size_t)(blockIdx.x * 256 + threadIdx.x) * 16 + (it & 15)];
            sink[(size_t)(blockIdx.x * 256 + threadIdx.x) * 16 + ((it + 5) & 15)] = v + 1.0f;
            y += v;
        }
    }
    float s = y + x;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) sink[0] = s;
}

template <int AGG>
static void run(unsigned long long* bad, float* sink, int blocks, const char* what) {
    CK(hipMemset(bad, 0, 32));
    hipLaunchKernelGGL((k_pk<AGG>), dim3(blocks), dim3(256), 0, 0, bad, sink, 4000);
    CK(hipDeviceSynchronize());
    unsigned long long h[4];
    CK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost));
    printf("  aggressor %-28s blocks %4d: packed != scalar in lanes 0-15: %llu, 16-31: %llu, 32-47: %llu, 48-63: %llu\n", what, blocks, h[0], h[1], h[2], h[3]);
    fflush(stdout);
}

int main() {
    unsigned long long* bad;
    float* sink;
    CK(hipMalloc(&bad, 32));
    CK(hipMalloc(&sink, (size_t)2048 * 256 * 16 * sizeof(float)));
    CK(hipMemset(sink, 0, (size_t)2048 * 256 * 16 * sizeof(float)));
    for (int blocks = 512; blocks <= 2048; blocks *= 2) {       // 2, 4, 8 workgroups per compute unit
        run<0>(bad, sink, blocks, "none");
        run<1>(bad, sink, blocks, "MFMA chains");
        run<2>(bad, sink, blocks, "v_exp / v_log (trans)");
        run<3>(bad, sink, blocks, "LDS read + write");
        run<4>(bad, sink, blocks, "global load + store");
        run<5>(bad, sink, blocks, "packed chains (all waves)");
    }
    return 0;
}
