// Probe (not product code): can the fp32 matrix products of the decoders run on the bf16 matrix pipe with fp32-class
// accuracy?  x = hi + mid + lo (three bf16 pieces by truncation, exact), a.b ~ the 6 piece products with i + j <= 2
// ("bf16x6"), each a v_mfma_f32_32x32x16_bf16 (32 cycles, 16 k) instead of v_mfma_f32_32x32x2_f32 (64 cycles, 2 k).
//   part 1: accuracy of fp32-MFMA / bf16x3 / bf16x6 against an fp64 host reference (32x32 tile, K = 128)
//   part 2: throughput of a register-chained layer loop  X <- 0.5 * W.X  (NB output blocks per 32-row input tile)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 f16x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bfloat(unsigned u) { return __builtin_bit_cast(float, u); }
// upper halves of (b, a) -> one register: low 16 = a's bf16, high 16 = b's bf16
__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

struct Split8 { bf16x8 p[3]; };
// eight fp32 values (registers 4g..4g+3 of k-groups g = 2G, 2G+1 of a CT tile) -> three bf16x8 operands
__device__ __forceinline__ Split8 split8(const float (&v)[8]) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned hb = fbits(v[i]) & 0xffff0000u;
        const float r1 = v[i] - bfloat(hb);
        const unsigned mb = fbits(r1) & 0xffff0000u;
        const float r2 = r1 - bfloat(mb);
        h[i] = hb; m[i] = mb; l[i] = fbits(r2);
    }
    Split8 s;
    u32x4 a, b, c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = pack_hi(h[2 * i], h[2 * i + 1]);
        b[i] = pack_hi(m[2 * i], m[2 * i + 1]);
        c[i] = pack_hi(l[2 * i], l[2 * i + 1]);
    }
    s.p[0] = __builtin_bit_cast(bf16x8, a); s.p[1] = __builtin_bit_cast(bf16x8, b); s.p[2] = __builtin_bit_cast(bf16x8, c);
    return s;
}

struct SplitH { u32x4 p[2]; };
// fp16 pieces: hi = rtz_f16(x) (pack-convert, 2 values per instruction), lo = rtz_f16(x - hi): 22 significand bits
__device__ __forceinline__ SplitH split8h(const float (&v)[8]) {
    SplitH s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f16x2 h = __builtin_amdgcn_cvt_pkrtz(v[2 * i], v[2 * i + 1]);
        const float r0 = v[2 * i] - (float)h[0], r1 = v[2 * i + 1] - (float)h[1];
        const f16x2 l = __builtin_amdgcn_cvt_pkrtz(r0, r1);
        s.p[0][i] = __builtin_bit_cast(unsigned, h);
        s.p[1][i] = __builtin_bit_cast(unsigned, l);
    }
    return s;
}
__device__ __forceinline__ f32x16 mma_h3(const SplitH& a, const SplitH& b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.p[1]), __builtin_bit_cast(f16x8, b.p[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.p[0]), __builtin_bit_cast(f16x8, b.p[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.p[0]), __builtin_bit_cast(f16x8, b.p[0]), acc, 0, 0, 0);
    return acc;
}

template <int TERMS>
__device__ __forceinline__ f32x16 mma_split(const Split8& a, const Split8& b, f32x16 acc) {
    // smallest terms first
    if (TERMS >= 6) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], b.p[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[1], acc, 0, 0, 0);
    }
    if (TERMS >= 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[1], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[0], acc, 0, 0, 0);
    return acc;
}

// ---- part 1: C[32x32] = A[32xK] . B[Kx32], one wave.  mode 0: fp32 MFMA, 3 / 6: split terms
__global__ void k_acc(const float* A, const float* B, float* C, int K, int mode) {
    const int lane = threadIdx.x, n = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[n * K + k + h], B[(k + h) * 32 + n], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            float av[8], bv[8];
            for (int i = 0; i < 8; ++i) { av[i] = A[n * K + k0 + 8 * h + i]; bv[i] = B[(k0 + 8 * h + i) * 32 + n]; }
            if (mode == 2) { acc = mma_h3(split8h(av), split8h(bv), acc); continue; }
            const Split8 a = split8(av), b = split8(bv);
            acc = (mode == 6) ? mma_split<6>(a, b, acc) : (mode == 3) ? mma_split<3>(a, b, acc) : mma_split<1>(a, b, acc);
        }
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + n] = acc[r];
}

// ---- part 2: layer chain.  X (NT CT tiles = 32*NT units x 32 samples) <- 0.5 * W . X with W fragments from global
// (L2-resident), NT input tiles, NT output blocks: a square 32*NT-wide layer, all in one wave (the geometry / rel-pos form).
template <int NT, int MODE>
__global__ __launch_bounds__(256) void k_chain(const float* __restrict__ Wf, const unsigned* __restrict__ Wb, float* out, int layers) {
    const int lane = threadIdx.x & 63;
    f32x16 x[NT];
    for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) x[t][i] = 0.01f * (float)((lane * 7 + i * 3 + t) % 17) - 0.08f;
    for (int L = 0; L < layers; ++L) {
        f32x16 y[NT];
        for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) y[t][i] = 0.f;
        if (MODE == 0) {
            // fp32: fragment blob [kt][g][nb][lane] float4 = four k-steps
            const float4* __restrict__ F = reinterpret_cast<const float4*>(Wf) + lane;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
                        const float4 a = F[((kt * 4 + g) * NT + nb) * 64];
                        y[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, x[kt][4 * g + 0], y[nb], 0, 0, 0);
                        y[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, x[kt][4 * g + 1], y[nb], 0, 0, 0);
                        y[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, x[kt][4 * g + 2], y[nb], 0, 0, 0);
                        y[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, x[kt][4 * g + 3], y[nb], 0, 0, 0);
                    }
        } else if (MODE == 2) {
            const u32x4* __restrict__ F = reinterpret_cast<const u32x4*>(Wb) + lane;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int G = 0; G < 2; ++G) {
                    float bv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) bv[i] = x[kt][8 * G + i];
                    const SplitH b = split8h(bv);
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
                        SplitH a;
#pragma unroll
                        for (int p = 0; p < 2; ++p) a.p[p] = F[(((kt * 2 + G) * NT + nb) * 3 + p) * 64];
                        y[nb] = mma_h3(a, b, y[nb]);
                    }
                }
        } else {
            // split: blob [kt][G][nb][piece][lane] uint4 (8 bf16), pieces pre-split on the host side of the product
            const u32x4* __restrict__ F = reinterpret_cast<const u32x4*>(Wb) + lane;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int G = 0; G < 2; ++G) {
                    float bv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) bv[i] = x[kt][8 * G + i];
                    const Split8 b = split8(bv);
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
                        Split8 a;
#pragma unroll
                        for (int p = 0; p < 3; ++p) a.p[p] = __builtin_bit_cast(bf16x8, F[(((kt * 2 + G) * NT + nb) * 3 + p) * 64]);
                        y[nb] = mma_split<MODE>(a, b, y[nb]);
                    }
                }
        }
        for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) x[t][i] = fmaxf(0.5f * y[t][i], -1.0f);
    }
    float s = 0.f;
    for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) s += x[t][i];
    if (s == 123.456f) out[0] = s;
}

template <int NT, int MODE>
static float time_chain(const float* Wf, const unsigned* Wb, float* out, int blocks, int layers) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_chain<NT, MODE>), dim3(blocks), dim3(256), 0, 0, Wf, Wb, out, layers);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_chain<NT, MODE>), dim3(blocks), dim3(256), 0, 0, Wf, Wb, out, layers);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5.f;
}

int main() {
    // ---- accuracy
    const int K = 128;
    std::vector<float> A(32 * K), B(K * 32), C(32 * 32);
    srand(7);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * expf(3.f * ((float)rand() / RAND_MAX - 0.5f));
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    const int modes[5] = {0, 1, 3, 6, 2};
    for (int mi = 0; mi < 5; ++mi) {
        hipLaunchKernelGGL(k_acc, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, modes[mi]);
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0.0, worst_f = 0.0;
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
            double ref = 0.0, mag = 0.0;
            float f = 0.f;
            for (int k = 0; k < K; ++k) { ref += (double)A[m * K + k] * B[k * 32 + n]; mag += fabs((double)A[m * K + k] * B[k * 32 + n]); f = fmaf(A[m * K + k], B[k * 32 + n], f); }
            worst = fmax(worst, fabs(C[m * 32 + n] - ref) / mag);
            worst_f = fmax(worst_f, fabs((double)C[m * 32 + n] - (double)f) / mag);
        }
        printf("accuracy mode %d: max |C - fp64| / sum|ab| = %.3e   max |C - fmaf chain| / sum|ab| = %.3e\n", modes[mi], worst, worst_f);
    }
    // ---- throughput
    float* Wf; unsigned* Wb; float* out;
    const size_t nW = 4 * 4 * 4 * 64 * 4;           // NT = 4: [kt 4][g 4][nb 4][lane 64] float4
    std::vector<float> w(nW);
    for (auto& v : w) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    std::vector<unsigned> wb(4 * 2 * 4 * 3 * 64 * 4);
    for (auto& v : wb) v = 0x3c003c00u + (rand() & 0x00ff00ff);
    CK(hipMalloc(&Wf, w.size() * 4)); CK(hipMalloc(&Wb, wb.size() * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemcpy(Wf, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Wb, wb.data(), wb.size() * 4, hipMemcpyHostToDevice));
    const int layers = 64;
    for (int wps = 1; wps <= 3; ++wps) {
        const int blocks = 256 * wps;                   // 4 waves per block -> wps waves per SIMD
        const float a1 = time_chain<1, 0>(Wf, Wb, out, blocks, layers), b1 = time_chain<1, 6>(Wf, Wb, out, blocks, layers), c1 = time_chain<1, 3>(Wf, Wb, out, blocks, layers);
        const float a4 = time_chain<4, 0>(Wf, Wb, out, blocks, layers), b4 = time_chain<4, 6>(Wf, Wb, out, blocks, layers), c4 = time_chain<4, 3>(Wf, Wb, out, blocks, layers);
        const float h1 = time_chain<1, 2>(Wf, Wb, out, blocks, layers), h4 = time_chain<4, 2>(Wf, Wb, out, blocks, layers);
        printf("   f16x3: 32-wide %.1f us   128-wide %.1f us\n", h1 * 1e3, h4 * 1e3);
        // cycles per layer per SIMD at 2.4 GHz nominal
        printf("waves/SIMD %d  32-wide layer: fp32 %.1f us  x6 %.1f us  x3 %.1f us   128-wide layer: fp32 %.1f us  x6 %.1f us  x3 %.1f us  (%d layers)\n",
               wps, a1 * 1e3, b1 * 1e3, c1 * 1e3, a4 * 1e3, b4 * 1e3, c4 * 1e3, layers);
    }
    return 0;
}
