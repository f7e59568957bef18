// Shared host/device helpers for libloopyhip (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "../../include/loopy_hip.h"
#include "lk_weights.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 lk_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 lk_f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 lk_f16x2 __attribute__((ext_vector_type(2)));

#define LK_TWO_PI 6.283185307179586f
#define LK_FLT_MAX 3.402823466e+38f

// ------------------------------------------------------------------ host side: errors
void lk_set_error(const char* fmt, ...);
#define LK_HIP_TRY(expr)                                                                   \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            lk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return LK_ERR_HIP;                                                             \
        }                                                                                  \
    } while (0)
#define LK_LAUNCH_CHECK()                                                                  \
    do {                                                                                   \
        hipError_t _e = hipGetLastError();                                                 \
        if (_e != hipSuccess) {                                                            \
            lk_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return LK_ERR_HIP;                                                             \
        }                                                                                  \
    } while (0)
#define LK_REQUIRE(cond, msg)                                                              \
    do {                                                                                   \
        if (!(cond)) { lk_set_error("%s (%s:%d)", msg, __FILE__, __LINE__); return LK_ERR_ARG; } \
    } while (0)

static inline int lk_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------ operand-range status word (loopy_hip.h: lk_status_peek)
unsigned* lk_status_dev();                  // device address of the word (host-mapped memory), or NULL if it could not be allocated
int lk_status_gate(const char* who);        // LK_ERR_RANGE (+ message) while a bit is set, else LK_OK
__device__ __forceinline__ void lk_status_raise(unsigned* status, unsigned bits) {
    if (status) __hip_atomic_fetch_or(status, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// fp16x3 operand ceiling: |x| < lim and finite (NaN fails the comparison)
__device__ __forceinline__ bool lk_out_of_range(float x, float lim) { return !(fabsf(x) < lim); }

// ------------------------------------------------------------------ neighbour grid (device resident)
struct LkGrid {
    float ox, oy, oz;      // origin = min corner of the point AABB
    float cell, inv_cell;  // cell edge (>= requested; grown until dx*dy*dz <= max_cells)
    int dx, dy, dz;
    int ncells;
    int n;                 // points in the grid
    int min_enc[3], max_enc[3];   // order-preserving int encodings of the AABB (reduction scratch)
};

#define LK_SEG_BATCH 8
struct lk_knn_s {
    float cell_size;
    int64_t capacity;      // points
    int64_t max_cells;
    int64_t n;             // points of the last build (host copy)
    LkGrid* grid;          // device
    float4* sorted;        // [capacity] (x,y,z,bitcast original index), cell order
    int32_t* cell_start;   // [max_cells + 1] exclusive prefix of per-cell counts
    int32_t* cell_of;      // [capacity]
    int32_t* rank_of;      // [capacity]
    int32_t* block_sums;   // scan scratch
    float* pos_own = nullptr;   // [capacity][3] original-order copy, allocated by the first lk_knn_append
    int seg_stride = 0, seg_sums_stride = 0;   // ints per batch member of the three arrays below (LK_SEG_BATCH members: lk_map_frame sorts the rows of several iterations per launch)
    int32_t* seg_cnt = nullptr;    // [capacity + 1] rows per point (zero between calls): the feature-gradient gather of lk_render_bwd
    int32_t* seg_off = nullptr;    // [capacity + 1] their exclusive offsets
    int32_t* seg_sums = nullptr;   // scan scratch of seg_cnt
    int32_t* row_rank = nullptr;   // [capacity] lk_map_frame with a row list: position of a point in that list, -1 = not optimised - the KEY of the
                                   // look-ahead row sort (its counters and scans then cover n_rows keys instead of all N points)
    uint8_t* act_flag = nullptr;   // [capacity] rows that received a gradient in the running optimize_map call (lk_map_frame, rows = NULL:
                                   // whole-map refinement steps only these rows, lk_adam_seg::row_flags); cleared at the call's first iteration
    int32_t n_scan_blocks;
};

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ int lk_lane() { return (int)(threadIdx.x & 63u); }
// value known to be the same in every lane of the wave -> scalar register (lets the compiler use scalar loads and
// real branches for everything derived from it)
__device__ __forceinline__ int lk_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Probe build only (tools/ab_build.sh chain -DLK_PROBE_CHAIN; never in the shipped library: tests/test_product_hygiene.py): the four
// launches of a tracking iteration leave 100-MHz wall-clock stamps (s_memrealtime: one clock for every compute unit and every launch) at
// their phase boundaries, [workgroup][wave][slot], in a buffer of their translation unit; lk_debug_chain_<k> copies it out and
// tools/probe/track_chain.py lays the last iteration's chain out on one time axis.  LK_STAMPW waits for the wave's outstanding memory
// operations first ("the data has arrived"), LK_STAMP does not.
#ifdef LK_PROBE_CHAIN
#define LK_CHAIN_WGS 512
#define LK_CHAIN_WAVES 8
#define LK_CHAIN_SLOTS 16
#define LK_CHAIN_DEFINE(NAME)                                                                                                   \
    __device__ unsigned long long g_lk_chain[LK_CHAIN_WGS][LK_CHAIN_WAVES][LK_CHAIN_SLOTS];                                     \
    extern "C" int lk_debug_chain_##NAME(unsigned long long* out) {                                                             \
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lk_chain), sizeof(g_lk_chain)) == hipSuccess ? 0 : 1;                      \
    }
#define LK_STAMP(I)                                                                                                             \
    do {                                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        if (lk_lane() == 0 && blockIdx.x < LK_CHAIN_WGS) g_lk_chain[blockIdx.x][threadIdx.x >> 6][I] = wall_clock64();          \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
    } while (0)
#define LK_STAMPW(I)                                                                                                            \
    do {                                                                                                                        \
        __builtin_amdgcn_s_waitcnt(0);                                                                                          \
        LK_STAMP(I);                                                                                                            \
    } while (0)
#else
#define LK_CHAIN_DEFINE(NAME)
#define LK_STAMP(I) do {} while (0)
#define LK_STAMPW(I) do {} while (0)
#endif

// squared distance exactly as the contract states: (dx*dx + dy*dy) + dz*dz, one rounding per op
__device__ __forceinline__ float lk_dist2(float qx, float qy, float qz, float px, float py, float pz) {
    const float dx = __fsub_rn(px, qx), dy = __fsub_rn(py, qy), dz = __fsub_rn(pz, qz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// sample position p = o + d*z with a rounded multiply then a rounded add (Renderer.py:167-168)
__device__ __forceinline__ float lk_madd_rn(float o, float d, float z) { return __fadd_rn(o, __fmul_rn(d, z)); }

// Fourier argument (2*pi*x) @ B[:,u] evaluated as torch-CPU does for K=3: a0*b0, then fma, fma
// (pinned by tests/test_oracle_golden.py::test_embed_fma_order).  a_i = fl(2*pi*x_i).
__device__ __forceinline__ float lk_fourier_arg(float a0, float a1, float a2, float b0, float b1, float b2) {
    return fmaf(a2, b2, fmaf(a1, b1, __fmul_rn(a0, b0)));
}

// ---- compact transcendental helpers.  The decoders evaluate ~500 activations and ~140 Fourier
// features per sample; OCML's fully accurate expf/log1pf/sinf (with the inlined Payne-Hanek path
// for large arguments) made the fused kernel ~400 KB of code, i.e. instruction-cache bound.  These
// versions are branch-free, a dozen instructions each, and accurate to ~1e-7 ABSOLUTE, which is
// what matters for O(1) activations (validated against the oracle in tests/).
// Raw v_exp_f32 / v_log_f32 (base 2, 1 ulp): the libm wrappers add a denormal-range rescue (compare, select, ldexp)
// around each of them, which doubles the VALU cost of an activation; here the argument of log2 is >= 1 and an
// underflowing 2^y may flush to 0 (absolute error < 1e-38 on an O(1) activation).
__device__ __forceinline__ float lk_exp2_raw(float y) { return __builtin_amdgcn_exp2f(y); }
__device__ __forceinline__ float lk_log2_raw(float y) { return __builtin_amdgcn_logf(y); }
__device__ __forceinline__ float lk_softplus100(float x) {
    // torch softplus(beta=100, threshold=20): log1p(exp(100x))/100, linear above the threshold (x > 0.2).
    // 2^(x * 100 log2 e), log2(1 + .) * (ln 2 / 100): two multiplies folded into the constants (the activation is the
    // largest VALU item of the colour path: ~2 700 evaluations per sample); |error| <= ~3e-9 absolute on the result.
    const float soft = lk_log2_raw(1.0f + lk_exp2_raw(x * 144.26950408889634f)) * 0.006931471805599453f;
    return x > 0.2f ? x : soft;
}
// d softplus100 / dx expressed through the OUTPUT a = softplus100(x): sigmoid(100x) = 1 - exp(-100a)
__device__ __forceinline__ float lk_softplus100_grad_from_out(float a) { return 1.0f - lk_exp2_raw(a * -144.26950408889634f); }
__device__ __forceinline__ float lk_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
// The colour trunk's SAVED derivative mask: softplus'(z_i) in [0, 1] as unorm16 (v_cvt_pknorm_u16_f32 packs two values per instruction),
// |error| <= 2^-17 = 7.6e-6 absolute.  The backward needs a_i = softplus(z_i) for nothing but this factor (d y_i = d h_i softplus'(z_i));
// the other way to drop the a_i rows - recovering a_i = h_i - (U_i c + u_i) from the h_i rows the weight gradients stream anyway - has
// the same error (half an ulp of h_i times the factor 100 of d softplus' / d a) at the price of a 128 x 32 product per layer in the
// backward.  Stored: 256 instead of 512 bytes per sample and layer (DESIGN.md §2 "activation scratch").
__device__ __forceinline__ unsigned lk_pack_unorm16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(lo, hi));
}
__device__ __forceinline__ float lk_unorm16_lo(unsigned w) { return (float)(w & 0xffffu) * (1.0f / 65535.0f); }
__device__ __forceinline__ float lk_unorm16_hi(unsigned w) { return (float)(w >> 16) * (1.0f / 65535.0f); }

// sin / cos for |x| < ~1e5: n = rint(x * 2/pi); r = x - n*pi/2 by a 3-term Cody-Waite reduction with
// fma (pi/2 = P1 + P2 + P3, P1 has 8 significant bits so n*P1 is exact), then the fdlibm minimax
// kernels on [-pi/4, pi/4] and a quadrant select.  ~1-2 ulp.
__device__ __forceinline__ void lk_sincos_core(float x, float& s, float& c, int& q) {
    const float n = rintf(x * 0.63661977236758134f);
    float r = fmaf(-n, 1.5703125f, x);
    r = fmaf(-n, 4.837512969970703125e-4f, r);
    r = fmaf(-n, 7.54978995489188e-8f, r);
    const float r2 = r * r;
    const float ps = fmaf(r2, fmaf(r2, fmaf(r2, 2.7183114939898219064e-6f, -1.98393348360966317347e-4f),
                                   8.3333293858894631756e-3f), -1.66666666416265235595e-1f);
    s = fmaf(r * r2, ps, r);
    const float pc = fmaf(r2, fmaf(r2, fmaf(r2, 2.43904487962774090654e-5f, -1.38867637746099294692e-3f),
                                   4.16666233237390631894e-2f), -4.99999997251031003120e-1f);
    c = fmaf(r2, pc, 1.0f);
    q = (int)n;
}
__device__ __forceinline__ float lk_sinf(float x) {
    float s, c; int q;
    lk_sincos_core(x, s, c, q);
    const float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
__device__ __forceinline__ float lk_cosf(float x) {
    float s, c; int q;
    lk_sincos_core(x, s, c, q);
    const float v = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -v : v;
}

// A value the optimiser cannot see through (constraint "v": a VGPR on the device, a vector register in the CPU test build).
// Inside a persistent-workgroup loop it makes what is derived from it loop-VARIANT: the 64-bit addresses of a few dozen weight
// fragment blocks are then formed where they are used instead of being hoisted out of the loop and spilled.
__device__ __forceinline__ int lk_opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// Four consecutive channels of a feature row, element index e (a multiple of 4) into the table: fp32 tables, or - opt-in,
// LK_FLAG_FEATS_F16 - IEEE half tables (8 bytes, exact conversion).  `f16` is uniform over the launch.
__device__ __forceinline__ float4 lk_feat4(const float* __restrict__ table, bool f16, size_t e) {
    if (!f16) return *reinterpret_cast<const float4*>(table + e);
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(table) + e);
    const lk_f16x2 a = __builtin_bit_cast(lk_f16x2, u.x), b = __builtin_bit_cast(lk_f16x2, u.y);
    return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
}

// C/D-fragment bookkeeping of v_mfma_f32_32x32x2_f32: lane l, register r holds
// row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
__device__ __forceinline__ int lk_frag_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 lk_mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 lk_zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}

// ------------------------------------------------------------------ register-chained transposed GEMM
// Activations live as "CT tiles": a [32 units x 32 samples] block held in the MFMA C/D layout
// (lane = sample column, registers = unit rows).  One layer is Y^T = W * X^T with
//   A operand = W[out = nb*32 + (lane&31)][k],  B operand = X^T[k][sample = lane&31].
// The reduction index k is walked in the order the C/D layout stores rows: lane half h of register 4g+t holds row
// 8g + 4h + t, so the registers of the previous layer's accumulator ARE the next layer's B operand (no LDS, no
// barrier): eight consecutive registers (two row groups) feed one 16-k matrix instruction, the weight fragments are laid
// out in the same k order (lk_weights.h), and a wave-wide 16-byte fragment load is one contiguous, fully used 1-KiB block.

// ------------------------------------------------------------------ fp32 products on the bf16 matrix pipe ("bf16x6")
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (64 cycles for 2 k) and - measured - does not overlap with the VALU
// work of the other waves of the SIMD: kernel time = MFMA cycles + VALU cycles.  The bf16 pipe is 16x faster per k and
// co-issues with the VALU.  An fp32 value is EXACTLY hi + mid + lo with three bf16 pieces cut by truncation (8 + 8 + 8
// significand bits); a.b is taken as the six piece products with i + j <= 2 (the dropped ones are <= 2^-24 |a b|),
// accumulated in fp32 smallest first: measured max error 1.8e-7 sum|a b| against fp64 (the fp32 MFMA: 2.0e-7).
// Six v_mfma_f32_32x32x16_bf16 (32 cycles, 16 k) replace eight fp32 MFMAs (64 cycles, 2 k): 2.67x fewer pipe cycles.
// The k order inside one instruction is free as long as A and B agree, so the 8 values of a lane are the two C/D-row-walk
// groups (2G, 2G+1) of the CT tile: registers 8G..8G+7, straight from the previous layer's accumulators.
struct LkB8 { u32x4 p[3]; };
// upper halves of (a, b) -> one register (low 16 bits = a's bf16)
__device__ __forceinline__ unsigned lk_pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ LkB8 lk_split8(const float (&v)[8]) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned hb = __float_as_uint(v[i]) & 0xffff0000u;
        const float r1 = v[i] - __uint_as_float(hb);
        const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(mb);
        h[i] = hb; m[i] = mb; l[i] = __float_as_uint(r2);
    }
    LkB8 s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s.p[0][i] = lk_pack_hi(h[2 * i], h[2 * i + 1]);
        s.p[1][i] = lk_pack_hi(m[2 * i], m[2 * i + 1]);
        s.p[2][i] = lk_pack_hi(l[2 * i], l[2 * i + 1]);
    }
    return s;
}
// registers 8G..8G+7 of a CT tile
__device__ __forceinline__ LkB8 lk_split_ct(const f32x16& x, int G) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = x[8 * G + i];
    return lk_split8(v);
}
__device__ __forceinline__ f32x16 lk_mfma_b16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(lk_bf16x8, a), __builtin_bit_cast(lk_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 lk_mma6(const LkB8& a, const LkB8& b, f32x16 acc) {
    acc = lk_mfma_b16(a.p[2], b.p[0], acc);
    acc = lk_mfma_b16(a.p[0], b.p[2], acc);
    acc = lk_mfma_b16(a.p[1], b.p[1], acc);
    acc = lk_mfma_b16(a.p[1], b.p[0], acc);
    acc = lk_mfma_b16(a.p[0], b.p[1], acc);
    acc = lk_mfma_b16(a.p[0], b.p[0], acc);
    return acc;
}
// split-fragment block (G, nb) of a matrix: fragb = first block of the matrix form (lkw::FMxx_FWDB / _TRB, uint4 units
// behind the fp32 fragments), NBT = blocks per G
__device__ __forceinline__ LkB8 lk_fragb_load(const u32x4* __restrict__ fragb, int NBT, int G, int nb, int lane) {
    const u32x4* __restrict__ q = fragb + ((size_t)G * NBT + nb) * 192 + lane;
    LkB8 a;
    a.p[0] = q[0]; a.p[1] = q[64]; a.p[2] = q[128];
    return a;
}
// acc[nb] += W[nb0 + nb][G0 .. G0+NGG) X  for the NGG 16-k blocks taken from registers of x starting at block XG0
template <int NB, int NGG>
__device__ __forceinline__ void lk_gemm_b6(f32x16 (&acc)[NB], const u32x4* __restrict__ fragb, int NBT, int G0, int nb0,
                                           const f32x16& x, int XG0, int lane) {
#pragma unroll
    for (int G = 0; G < NGG; ++G) {
        const LkB8 b = lk_split_ct(x, XG0 + G);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = lk_mma6(lk_fragb_load(fragb, NBT, G0 + G, nb0 + nb, lane), b, acc[nb]);
    }
}

// ------------------------------------------------------------------ forward products as fp16x3
// Two fp16 pieces per value (hi = rtz_f16(x) by pack-convert, lo = rtz_f16(x - hi): 22 significand bits, 3 VALU
// instructions per value instead of 5.5) and three products hi.hi + hi.lo + lo.hi: measured 1.7e-7 sum|ab| against fp64 on
// O(1) data - the same as bf16x6 - at half the matrix instructions.  fp16 has a narrow exponent: the low piece of
// |x| < 0.1 is a subnormal (absolute error <= 3e-8 per element instead of relative 2^-22) and |x| > 65 504 saturates, which
// is harmless for the forward operands (Fourier features, interpolated point features, activations, weights: O(1e-3 .. 10))
// and is why the BACKWARD kernels, whose operands are gradients of arbitrary scale, stay on the bf16 pieces.
struct LkH8 { u32x4 p[2]; };
__device__ __forceinline__ LkH8 lk_split8h(const float (&v)[8]) {
    LkH8 s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const lk_f16x2 h = __builtin_amdgcn_cvt_pkrtz(v[2 * i], v[2 * i + 1]);
        const float r0 = v[2 * i] - (float)h[0], r1 = v[2 * i + 1] - (float)h[1];
        const lk_f16x2 l = __builtin_amdgcn_cvt_pkrtz(r0, r1);
        s.p[0][i] = __builtin_bit_cast(unsigned, h);
        s.p[1][i] = __builtin_bit_cast(unsigned, l);
    }
    return s;
}
__device__ __forceinline__ LkH8 lk_split_cth(const f32x16& x, int G) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = x[8 * G + i];
    return lk_split8h(v);
}
__device__ __forceinline__ f32x16 lk_mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(lk_f16x8, a), __builtin_bit_cast(lk_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 lk_mma3h(const LkH8& a, const LkH8& b, f32x16 acc) {
    acc = lk_mfma_f16(a.p[1], b.p[0], acc);
    acc = lk_mfma_f16(a.p[0], b.p[1], acc);
    acc = lk_mfma_f16(a.p[0], b.p[0], acc);
    return acc;
}
// fp16 forward-fragment block (G, nb): fragh = first block of the matrix (lkw::FMxx_FWDH, uint4 units behind the bf16 blob)
__device__ __forceinline__ LkH8 lk_fragh_load(const u32x4* __restrict__ fragh, int NBT, int G, int nb, int lane) {
    const u32x4* __restrict__ q = fragh + ((size_t)G * NBT + nb) * 128 + lane;
    LkH8 a;
    a.p[0] = q[0]; a.p[1] = q[64];
    return a;
}
template <int NB, int NGG>
__device__ __forceinline__ void lk_gemm_h3(f32x16 (&acc)[NB], const u32x4* __restrict__ fragh, int NBT, int G0, int nb0,
                                           const f32x16& x, int XG0, int lane) {
#pragma unroll
    for (int G = 0; G < NGG; ++G) {
        const LkH8 b = lk_split_cth(x, XG0 + G);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = lk_mma3h(lk_fragh_load(fragh, NBT, G0 + G, nb0 + nb, lane), b, acc[nb]);
    }
}

// Barrier among SOME waves of a workgroup (s_barrier waits for every wave of the workgroup that has not ended: a wave that runs a long chain
// of its own - the geometry decoder on wave 4 of the tracker's fused forward - holds the others at their first barrier until it is through).
// cnt: an LDS word, zero when the participants start; target = participants x (ordinal of this barrier, from 1).  Every participant calls
// it the same number of times; LDS writes before it are visible to the others' reads after it.
__device__ __forceinline__ void lk_soft_barrier(unsigned* cnt, unsigned target) {
    __threadfence_block();
    if (lk_lane() == 0) atomicAdd(cnt, 1u);
    while (*reinterpret_cast<volatile unsigned*>(cnt) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence_block();
}

// Sum over the 32 lanes of each half-wave with DPP adds (plain VALU, no LDS traffic: the ds_bpermute form of the same
// reduction cost 0.8 ms per benchmark step in the embedding-gradient sections).  quad swaps, then the two mirrors give
// every lane its 16-lane row total; row_bcast15 adds row 0 (2) into row 1 (3).
// THE TOTAL IS VALID IN LANES 16..31 AND 48..63 ONLY (lanes with bit 4 set); LK_HWS_LANE is the lane of a half to read.
#define LK_HWS_LANE 16
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float lk_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// PIN: an empty asm behind the last add keeps it beside its DPP move.  Where the consumer sits in a one-lane branch the compiler otherwise sinks
// the add into the branch, and the step is mov 0 + mov_dpp + add instead of ONE add_dpp (rel-pos backward kernels: 128 instructions per tile);
// where the sums feed straight-line code (k_decode_bwd) the pin only costs hazard nops - chosen per call site.
template <bool PIN = false>
__device__ __forceinline__ float lk_half_wave_sum(float v) {
    v += lk_dpp<0xB1, 0xF>(v);          // quad_perm [1,0,3,2]
    v += lk_dpp<0x4E, 0xF>(v);          // quad_perm [2,3,0,1]
    v += lk_dpp<0x141, 0xF>(v);         // row_half_mirror: the other quad of the 8-group
    v += lk_dpp<0x140, 0xF>(v);         // row_mirror: the other 8-group of the row
    v += lk_dpp<0x142, 0xA>(v);         // row_bcast15 into rows 1 and 3
    if (PIN) asm("" : "+v"(v));
    return v;
}

// sum over the 8 lanes of an aligned 8-group (the 8 neighbour rows of a sample): every lane gets the total
template <bool PIN = false>
__device__ __forceinline__ float lk_sum8(float v) {
    v += lk_dpp<0xB1, 0xF>(v);
    v += lk_dpp<0x4E, 0xF>(v);
    v += lk_dpp<0x141, 0xF>(v);
    if (PIN) asm("" : "+v"(v));
    return v;
}
// 64-bit value of the partner lane in DPP pairing `CTRL` (quad swaps and the row mirrors are perfect matchings between
// the two halves of every 2-, 4-, 8- and 16-lane group)
template <int CTRL>
__device__ __forceinline__ uint64_t lk_dpp_u64(uint64_t v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xF, 0xF, false);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

// CT tile whose 16 registers are a per-row vector (bias): the start value of an accumulator - the bias add then costs no
// VALU instruction at all (four 16-byte loads land in the accumulator registers)
__device__ __forceinline__ f32x16 lk_rowvec_tile(const float* __restrict__ v, int unit0, int lane) {
    const int h = lane >> 5;
    f32x16 t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(v + unit0 + 8 * g + 4 * h);
        t[4 * g + 0] = b.x; t[4 * g + 1] = b.y; t[4 * g + 2] = b.z; t[4 * g + 3] = b.w;
    }
    return t;
}
// per-row vector (bias) of a CT tile: v[unit(r,h)] for r = 0..15, unit0 = first unit of the tile
__device__ __forceinline__ void lk_add_rowvec(f32x16& acc, const float* __restrict__ v, int unit0, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(v + unit0 + 8 * g + 4 * h);
        acc[4 * g + 0] += b.x; acc[4 * g + 1] += b.y; acc[4 * g + 2] += b.z; acc[4 * g + 3] += b.w;
    }
}
