// TEST-ONLY host emulation of the slice of HIP that loopy_slam_amd/csrc uses.
//
// Purpose: this container has no GPU and GPU minutes are scarce, so the kernel
// *logic* (indexing, MFMA fragment bookkeeping, barriers, atomics) is exercised on
// the CPU by compiling the unmodified .hip sources against this header
// (`-I tests/hipemu`, which shadows <hip/hip_runtime.h>) into
// tests/hipemu/_build/libloopyhip_emu.so.  Only tests load that library.  The
// product package loads libloopyhip.so (gfx950 code object) and nothing else; there
// is no CPU fallback in the product path.
//
// Model: blocks run one after another on the calling OS thread; the threads of a
// block are fibers (hand-rolled x86-64 context switch) scheduled round-robin and
// switched only at __syncthreads() and at wave collectives (shuffles, ballots,
// MFMA).  A wave is 64 consecutive threads.  MFMA follows the gfx950 lane layout
// documented in /opt/skills/guides/cdna_hip_programming.md §3:
//   32x32x2 f32:  A: lane l holds A[i=l&31][k=l>>5];  B: lane l holds B[k=l>>5][j=l&31];
//                 C/D: lane l, reg r holds D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31];
//                 D = fma(a_k1, b_k1, fma(a_k0, b_k0, C))  (k-ordered fp32 fma chain).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <tuple>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

typedef int hipError_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

namespace hipemu {
struct Lane;
struct LaneView { uint3 tid; uint3 bid; dim3 bdim; dim3 gdim; };
extern LaneView* cur_view;
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void syncthreads();
// wave collectives (all ALIVE lanes of the wave must call them convergently)
uint32_t wave_exchange_u32(uint32_t v, int src_lane);          // returns v of src_lane (garbage if exited)
uint64_t wave_ballot(bool p);
void wave_mfma32x32x2(float a, float b, const float* c, float* d);
void wave_mfma16x16x4(float a, float b, const float* c, float* d);
void wave_mfma32x32x16_bf16(const float* a8, const float* b8, const float* c, float* d);
void wave_mfma16x16x32(const float* a8, const float* b8, const float* c, float* d);
void spin_yield();
int lane_id();
}  // namespace hipemu

#define threadIdx (hipemu::cur_view->tid)
#define blockIdx (hipemu::cur_view->bid)
#define blockDim (hipemu::cur_view->bdim)
#define gridDim (hipemu::cur_view->gdim)
#define warpSize 64

namespace hipemu {
template <typename K, typename... A>
static inline void launch_k(dim3 grid, dim3 block, K kernel, A... args) {
    // arguments are evaluated at the launch site and passed by value, as on the device
    std::function<void()> body = [=]() { kernel(args...); };
    launch(grid, block, body);
}
}  // namespace hipemu
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch_k(dim3(grid), dim3(block), kernel, __VA_ARGS__)

static inline void __syncthreads() { hipemu::syncthreads(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---- wave intrinsics
static inline int __lane_id() { return hipemu::lane_id(); }
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "emu shuffles are 32- or 64-bit (two 32-bit exchanges)");
    uint32_t u[2] = {0, 0}; std::memcpy(u, &v, sizeof(T));
    int lane = hipemu::lane_id();
    int s = (lane & ~(width - 1)) | (src & (width - 1));
    u[0] = hipemu::wave_exchange_u32(u[0], s);
    if (sizeof(T) == 8) u[1] = hipemu::wave_exchange_u32(u[1], s);
    T r; std::memcpy(&r, u, sizeof(T)); return r;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    return __shfl(v, (hipemu::lane_id() ^ mask) & (width - 1), width);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = hipemu::lane_id(); int l = lane & (width - 1);
    return __shfl(v, (l + (int)d < width) ? l + (int)d : l, width);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = hipemu::lane_id(); int l = lane & (width - 1);
    return __shfl(v, (l - (int)d >= 0) ? l - (int)d : l, width);
}
static inline unsigned long long __ballot(int p) { return hipemu::wave_ballot(p != 0); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
static inline int __any(int p) { return __ballot(p) != 0ull; }

// ---- MFMA
typedef float f32x16_emu __attribute__((ext_vector_type(16)));
typedef float f32x4_emu __attribute__((ext_vector_type(4)));
static inline f32x16_emu __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, f32x16_emu c, int, int, int) {
    float ci[16], di[16];
    for (int r = 0; r < 16; ++r) ci[r] = c[r];
    hipemu::wave_mfma32x32x2(a, b, ci, di);
    f32x16_emu d;
    for (int r = 0; r < 16; ++r) d[r] = di[r];
    return d;
}
static inline f32x4_emu __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4_emu c, int, int, int) {
    float ci[4], di[4];
    for (int r = 0; r < 4; ++r) ci[r] = c[r];
    hipemu::wave_mfma16x16x4(a, b, ci, di);
    f32x4_emu d;
    for (int r = 0; r < 4; ++r) d[r] = di[r];
    return d;
}
// bf16 MFMA: operands are 8 bf16 per lane (passed as the product's own vector type, widened here)
typedef __bf16 bf16x8_emu __attribute__((ext_vector_type(8)));
static inline f32x16_emu __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf16x8_emu a, bf16x8_emu b, f32x16_emu c, int, int, int) {
    unsigned short ua[8], ub[8];
    std::memcpy(ua, &a, 16); std::memcpy(ub, &b, 16);
    float fa[8], fb[8], ci[16], di[16];
    for (int e = 0; e < 8; ++e) {
        unsigned x = (unsigned)ua[e] << 16, y = (unsigned)ub[e] << 16;
        std::memcpy(&fa[e], &x, 4); std::memcpy(&fb[e], &y, 4);
    }
    for (int r = 0; r < 16; ++r) ci[r] = c[r];
    hipemu::wave_mfma32x32x16_bf16(fa, fb, ci, di);
    f32x16_emu d;
    for (int r = 0; r < 16; ++r) d[r] = di[r];
    return d;
}
// fp16 MFMA (same lane layout as the bf16 form) and the pack-convert used to cut fp32 values into fp16 pieces
typedef _Float16 f16x8_emu __attribute__((ext_vector_type(8)));
static inline f32x16_emu __builtin_amdgcn_mfma_f32_32x32x16_f16(f16x8_emu a, f16x8_emu b, f32x16_emu c, int, int, int) {
    float fa[8], fb[8], ci[16], di[16];
    for (int e = 0; e < 8; ++e) { fa[e] = (float)a[e]; fb[e] = (float)b[e]; }
    for (int r = 0; r < 16; ++r) ci[r] = c[r];
    hipemu::wave_mfma32x32x16_bf16(fa, fb, ci, di);
    f32x16_emu d;
    for (int r = 0; r < 16; ++r) d[r] = di[r];
    return d;
}
static inline f32x4_emu __builtin_amdgcn_mfma_f32_16x16x32_f16(f16x8_emu a, f16x8_emu b, f32x4_emu c, int, int, int) {
    float fa[8], fb[8], ci[4], di[4];
    for (int e = 0; e < 8; ++e) { fa[e] = (float)a[e]; fb[e] = (float)b[e]; }
    for (int r = 0; r < 4; ++r) ci[r] = c[r];
    hipemu::wave_mfma16x16x32(fa, fb, ci, di);
    f32x4_emu d;
    for (int r = 0; r < 4; ++r) d[r] = di[r];
    return d;
}
static inline f32x4_emu __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf16x8_emu a, bf16x8_emu b, f32x4_emu c, int, int, int) {
    unsigned short ua[8], ub[8];
    std::memcpy(ua, &a, 16); std::memcpy(ub, &b, 16);
    float fa[8], fb[8], ci[4], di[4];
    for (int e = 0; e < 8; ++e) {
        unsigned x = (unsigned)ua[e] << 16, y = (unsigned)ub[e] << 16;
        std::memcpy(&fa[e], &x, 4); std::memcpy(&fb[e], &y, 4);
    }
    for (int r = 0; r < 4; ++r) ci[r] = c[r];
    hipemu::wave_mfma16x16x32(fa, fb, ci, di);
    f32x4_emu d;
    for (int r = 0; r < 4; ++r) d[r] = di[r];
    return d;
}
typedef __fp16 f16x2_emu __attribute__((ext_vector_type(2)));
static inline _Float16 hipemu_rtz_f16(float x) {        // round toward zero, saturating (v_cvt_pkrtz_f16_f32)
    if (x != x) return (_Float16)x;
    _Float16 h = (_Float16)x;                            // nearest
    if (std::fabs((float)h) > std::fabs(x) || std::isinf((float)h)) {
        unsigned short b; std::memcpy(&b, &h, 2);
        b = (unsigned short)(b - 1);                     // one step toward zero (sign-magnitude encoding)
        std::memcpy(&h, &b, 2);
    }
    return h;
}
static inline f16x2_emu __builtin_amdgcn_cvt_pkrtz(float a, float b) {
    const _Float16 x = hipemu_rtz_f16(a), y = hipemu_rtz_f16(b);
    f16x2_emu r;
    std::memcpy(reinterpret_cast<char*>(&r), &x, 2); std::memcpy(reinterpret_cast<char*>(&r) + 2, &y, 2);
    return r;
}
// v_perm_b32: byte pool {S0 = bytes 7..4, S1 = bytes 3..0}, selector byte n of `sel` picks pool byte (0..7)
typedef unsigned short u16x2_emu __attribute__((ext_vector_type(2)));
static inline u16x2_emu __builtin_amdgcn_cvt_pknorm_u16(float a, float b) {        // v_cvt_pknorm_u16_f32: round-to-nearest-even of clamp(x, 0, 1) * 65535
    auto q = [](float x) { x = x != x ? 0.0f : (x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x)); return (unsigned short)::nearbyintf(x * 65535.0f); };
    u16x2_emu r; r[0] = q(a); r[1] = q(b); return r;
}
static inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel) {
    const unsigned long long pool = ((unsigned long long)s0 << 32) | s1;
    unsigned r = 0;
    for (int n = 0; n < 4; ++n) {
        const unsigned c = (sel >> (8 * n)) & 0xff;
        const unsigned byte = c < 8 ? (unsigned)((pool >> (8 * c)) & 0xff) : (c == 0x0c ? 0u : 0xffu);
        r |= byte << (8 * n);
    }
    return r;
}
// DPP data movement (v_mov_b32 dpp): quad_perm, row_shl/shr/ror, row_mirror, row_half_mirror, row_bcast15/31.
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int lane = hipemu::lane_id(), row = lane >> 4, l16 = lane & 15;
    int from = lane;
    bool valid = true;
    if (ctrl < 0x100) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
    else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; valid = l16 + n < 16; from = lane + n; }
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; valid = l16 >= n; from = lane - n; }
    else if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; from = (row << 4) | ((l16 - n + 16) & 15); }
    else if (ctrl == 0x140) from = (row << 4) | (15 - l16);
    else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));
    else if (ctrl == 0x142) { valid = row >= 1; from = ((row - 1) << 4) | 15; }
    else if (ctrl == 0x143) { valid = row >= 2; from = 31; }
    else valid = false;
    const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> (l16 >> 2)) & 1);
    const int got = (int)hipemu::wave_exchange_u32((uint32_t)src, valid ? from : lane);     // every lane takes part
    if (!enabled) return old;
    if (!valid) return bound_ctrl ? 0 : old;
    return got;
}
static inline float __builtin_amdgcn_exp2f(float v) { return ::exp2f(v); }
static inline float __builtin_amdgcn_logf(float v) { return ::log2f(v); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // only ever applied to wave-uniform values
static inline void __builtin_amdgcn_s_setprio(int) {}
// global_load_lds_dwordx4 (gfx950): every lane's 16 bytes go straight to LDS at (uniform base) + 16 * lane, no register in between
template <class G, class L>
static inline void __builtin_amdgcn_global_load_lds(const G* g, L lds_base, int size, int, int) {
    char* dst = (char*)(__UINTPTR_TYPE__)lds_base + (size_t)size * hipemu::lane_id();
    __builtin_memcpy(dst, (const void*)g, size);
}
static inline void __builtin_amdgcn_sched_barrier(int) {}
// compiler-level wave barrier on the device; here the point where every lane's LDS accesses so far have happened
static inline void __builtin_amdgcn_s_sleep(int) { hipemu::spin_yield(); }
static inline void __builtin_amdgcn_wave_barrier() { (void)hipemu::wave_ballot(true); }
// 100 MHz wall clock of the device (k_spin_us, the side-stream delay test hook): the emulator's streams run in order on one host thread,
// so a delay has nothing to reorder - it returns at once
static inline long long wall_clock64() { static long long t = 0; return t += (1ll << 40); }

// ---- math (round-to-nearest, never contracted)
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fmaf_rn(float a, float b, float c) { return std::fmaf(a, b, c); }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; std::memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; std::memcpy(&f, &i, 4); return f; }
#define __expf(x) (::expf(x))
#define __logf(x) (::logf(x))
// floorf/sqrtf/fabsf/fminf/fmaxf/expf/logf/log1pf/expm1f/sinf/cosf/fmaf come from <cmath> (global namespace)
template <typename T> static inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> static inline T max(T a, T b) { return a > b ? a : b; }

// ---- atomics (single OS thread => plain read-modify-write)
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline float unsafeAtomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
// (__hip_atomic_fetch_or / __HIP_MEMORY_SCOPE_SYSTEM are clang builtins on the host too: lk_status_raise compiles as it stands)
#ifndef __HIP_MEMORY_SCOPE_SYSTEM
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif

// ---- host API
namespace hipemu { void* guarded_alloc(size_t n); void guarded_free(void* p); }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = hipemu::guarded_alloc(n); return *p ? hipSuccess : 2; }
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { hipemu::guarded_free(p); return hipSuccess; }
#define hipHostMallocMapped 2
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = hipemu::guarded_alloc(n); return *p ? hipSuccess : 2; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)1; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)1; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
template <class F> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 1; return hipSuccess; }   // no registers, no LDS limit on the host
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
