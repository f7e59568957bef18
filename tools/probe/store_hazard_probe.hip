// Probe (not product code): the "store-data rule" of DESIGN.md, reduced to the instruction level.
//
// Round 1 saw wrong half-tiles of d h in k_decode_bwd when the 16-byte row stores of a layer's gradient read the very
// accumulator registers that the next product overwrites.  Two candidate hazards, neither visible to the compiler's
// hazard recogniser when it matters, both testable with hand-placed instructions (inline asm is opaque to the recogniser,
// so NOTHING is inserted between the instructions below):
//
//   RAW  v_mfma_f32_32x32x16_bf16 writes v[100:115]  ->  global_store_dwordx4 reads them WAIT wait states later.
//        Stale data (the sentinel written before the MFMA) in memory = the store read its data before the matrix pipe
//        had written the result back.
//   WAR  global_store_dwordx4 reads v[100:115]  ->  GAP wait states later a VALU (or an MFMA) overwrites them.
//        The overwriting value in memory = the store fetched its data after the instruction that follows it in
//        program order had already written the registers.
//
// Four 16-byte stores per lane and iteration, rows 2 560 B apart (the layout of the d h rows: 640 floats per sample), run
// with 1, 2 and 3 waves per SIMD so that the memory pipe of a compute unit is shared the way two workgroups of the real
// kernel share it.  Every element of the buffer is written exactly once and checked on the device afterwards.
//
//   hipcc --offload-arch=gfx950 -O2 -o store_hazard_probe store_hazard_probe.hip && ./store_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define ROW_FLOATS 640
#define ITERS 20                      // 20 x 32 floats (two half-waves x 16) = one 640-float row per sample column
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115"

#define STORES                                              \
    "global_store_dwordx4 %[p], v[100:103], off\n"          \
    "global_store_dwordx4 %[p], v[104:107], off offset:32\n"\
    "global_store_dwordx4 %[p], v[108:111], off offset:64\n"\
    "global_store_dwordx4 %[p], v[112:115], off offset:96\n"
#define FILL(src)                                           \
    ".irp r,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115\n" \
    "v_mov_b32 v\\r, " src "\n"                             \
    ".endr\n"

// MODE 0: RAW (MFMA -> store).  MODE 1: WAR (store -> VALU).  MODE 2: WAR (store -> MFMA).  MODE 3: as 1 with s_waitcnt vmcnt(0)
// instead of wait states (the fully safe form, must be clean).
template <int MODE, int WAIT>
__global__ __launch_bounds__(256) void k_probe(float* buf) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float* row = buf + (wave * 32 + (lane & 31)) * ROW_FLOATS + (lane >> 5) * 4;
    const unsigned one = 0x3f803f80u;                 // packed bf16 1.0
    const u32x4 A = {one, one, one, one};
    const float neg = -1.0f;
    for (int it = 0; it < ITERS; ++it) {
        float* p = row + it * 32;
        const unsigned short bb = (unsigned short)(__float_as_uint((float)(it + 1)) >> 16);     // exact in bf16 (<= 20)
        const unsigned bp = bb | ((unsigned)bb << 16);
        const u32x4 B = {bp, bp, bp, bp};
        const float good = (float)(it + 1);
        if (MODE == 0) {
            asm volatile(
                FILL("%[neg]")
                "s_nop 7\n"
                "v_mfma_f32_32x32x16_bf16 v[100:115], %[a], %[b], 0\n"
                ".rept %[wt]\n"
                "s_nop 0\n"
                ".endr\n"
                STORES
                "s_waitcnt vmcnt(0)\n"
                :: [p] "v"(p), [a] "v"(A), [b] "v"(B), [neg] "v"(neg), [wt] "n"(WAIT)
                : "memory", CLOB);
        } else if (MODE == 1) {
            asm volatile(
                FILL("%[good]")
                "s_nop 7\n"
                STORES
                ".rept %[wt]\n"
                "s_nop 0\n"
                ".endr\n"
                FILL("%[neg]")
                :: [p] "v"(p), [good] "v"(good), [neg] "v"(neg), [wt] "n"(WAIT)
                : "memory", CLOB);
        } else if (MODE == 2) {
            asm volatile(
                FILL("%[good]")
                "s_nop 7\n"
                STORES
                ".rept %[wt]\n"
                "s_nop 0\n"
                ".endr\n"
                "v_mfma_f32_32x32x16_bf16 v[100:115], %[a], %[b], 0\n"
                "s_nop 15\n"
                :: [p] "v"(p), [a] "v"(A), [b] "v"(B), [good] "v"(good), [wt] "n"(WAIT)
                : "memory", CLOB);
        } else if (MODE == 4 || MODE == 5) {
            // RAW behind a CHAIN of four dependent MFMAs (the shape of a layer product): the matrix pipe of the SIMD is busy
            // for 4 x 32 cycles per wave, so with 2-3 waves per SIMD an MFMA regularly queues behind another wave's.
            // MODE 5 = the remedy under test: a VALU copy of the result (v_mov, WAIT wait states after the MFMA) is what the
            // stores read.
            if (MODE == 4) {
                asm volatile(
                    FILL("%[neg]")
                    "s_nop 7\n"
                    "v_mfma_f32_32x32x16_bf16 v[100:115], %[a], %[b], 0\n"
                    "v_mfma_f32_32x32x16_bf16 v[100:115], %[a], %[b], v[100:115]\n"
                    "v_mfma_f32_32x32x16_bf16 v[100:115], %[a], %[b], v[100:115]\n"
                    "v_mfma_f32_32x32x16_bf16 v[100:115], %[a], %[b], v[100:115]\n"
                    ".rept %[wt]\n"
                    "s_nop 0\n"
                    ".endr\n"
                    STORES
                    "s_waitcnt vmcnt(0)\n"
                    :: [p] "v"(p), [a] "v"(A), [b] "v"(B), [neg] "v"(neg), [wt] "n"(WAIT)
                    : "memory", CLOB);
            } else {
                asm volatile(
                    FILL("%[neg]")
                    ".irp r,116,117,118,119,120,121,122,123,124,125,126,127,128,129,130,131\n"
                    "v_mov_b32 v\\r, %[neg]\n"
                    ".endr\n"
                    "s_nop 7\n"
                    "v_mfma_f32_32x32x16_bf16 v[100:115], %[a], %[b], 0\n"
                    "v_mfma_f32_32x32x16_bf16 v[100:115], %[a], %[b], v[100:115]\n"
                    "v_mfma_f32_32x32x16_bf16 v[100:115], %[a], %[b], v[100:115]\n"
                    "v_mfma_f32_32x32x16_bf16 v[100:115], %[a], %[b], v[100:115]\n"
                    ".rept %[wt]\n"
                    "s_nop 0\n"
                    ".endr\n"
                    ".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n"
                    "v_mov_b32 v[116+\\r], v[100+\\r]\n"
                    ".endr\n"
                    "global_store_dwordx4 %[p], v[116:119], off\n"
                    "global_store_dwordx4 %[p], v[120:123], off offset:32\n"
                    "global_store_dwordx4 %[p], v[124:127], off offset:64\n"
                    "global_store_dwordx4 %[p], v[128:131], off offset:96\n"
                    "s_waitcnt vmcnt(0)\n"
                    :: [p] "v"(p), [a] "v"(A), [b] "v"(B), [neg] "v"(neg), [wt] "n"(WAIT)
                    : "memory", CLOB, "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127",
                      "v128", "v129", "v130", "v131");
            }
        } else if (MODE >= 6) {
            // Packed fp32 VALU ops (v_pk_add_f32 & co.: TWO passes over the wave, the SLP vectoriser creates them).  The failing
            // build computed d h of the last layer with chains of them (built without -fno-slp-vectorize) and showed ONE wrong
            // register in lanes 48-63.
            //   MODE 6 / 7: WAR on the SOURCES of a packed op - v_pk_add_f32 reads v[120:121]; WAIT wait states later a plain
            //               VALU (6: two v_mov, 7: v_lshl_add_u64, the instruction of the failing code) overwrites them.
            //   MODE 8:     RAW packed result -> global_store data.   MODE 9: RAW packed result -> v_mov (VALU) -> store.
            const float onef = 1.0f;
            const unsigned long long junk = 0x7fc000007fc00000ull, zero64 = 0ull;
            if (MODE == 6 || MODE == 7) {
                asm volatile(
                    FILL("%[neg]")
                    "v_mov_b32 v120, %[good]\n v_mov_b32 v121, %[good]\n v_mov_b32 v122, %[one]\n v_mov_b32 v123, %[one]\n"
                    "s_nop 7\n"
                    "v_pk_add_f32 v[124:125], v[122:123], v[120:121]\n"
                    ".rept %[wt]\n"
                    "s_nop 0\n"
                    ".endr\n"
                    ".if %[mode] == 6\n"
                    "v_mov_b32 v121, %[neg]\n v_mov_b32 v120, %[neg]\n"
                    ".else\n"
                    "v_lshl_add_u64 v[120:121], %[junk], 0, %[zero]\n"
                    ".endif\n"
                    "s_nop 15\n"
                    ".irp r,100,101,102,103,104,105,106,107\n v_mov_b32 v\\r, v124\n .endr\n"
                    ".irp r,108,109,110,111,112,113,114,115\n v_mov_b32 v\\r, v125\n .endr\n"
                    "s_nop 7\n"
                    STORES
                    "s_waitcnt vmcnt(0)\n"
                    :: [p] "v"(p), [good] "v"(good), [neg] "v"(neg), [one] "v"(onef), [junk] "v"(junk), [zero] "v"(zero64), [wt] "n"(WAIT), [mode] "n"(MODE)
                    : "memory", CLOB, "v120", "v121", "v122", "v123", "v124", "v125");
            } else if (MODE == 8) {
                asm volatile(
                    FILL("%[neg]")
                    "v_mov_b32 v120, %[good]\n v_mov_b32 v121, %[good]\n v_mov_b32 v122, %[one]\n v_mov_b32 v123, %[one]\n"
                    "s_nop 7\n"
                    ".irp r,100,102,104,106,108,110,112,114\n v_pk_add_f32 v[\\r:\\r+1], v[122:123], v[120:121]\n .endr\n"
                    ".rept %[wt]\n"
                    "s_nop 0\n"
                    ".endr\n"
                    "global_store_dwordx4 %[p], v[112:115], off offset:96\n"
                    "global_store_dwordx4 %[p], v[108:111], off offset:64\n"
                    "global_store_dwordx4 %[p], v[104:107], off offset:32\n"
                    "global_store_dwordx4 %[p], v[100:103], off\n"
                    "s_waitcnt vmcnt(0)\n"
                    :: [p] "v"(p), [good] "v"(good), [neg] "v"(neg), [one] "v"(onef), [wt] "n"(WAIT)
                    : "memory", CLOB, "v120", "v121", "v122", "v123");
            } else {
                asm volatile(
                    FILL("%[neg]")
                    ".irp r,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147\n v_mov_b32 v\\r, %[neg]\n .endr\n"
                    "v_mov_b32 v120, %[good]\n v_mov_b32 v121, %[good]\n v_mov_b32 v122, %[one]\n v_mov_b32 v123, %[one]\n"
                    "s_nop 7\n"
                    ".irp r,100,102,104,106,108,110,112,114\n v_pk_add_f32 v[\\r:\\r+1], v[122:123], v[120:121]\n .endr\n"
                    ".rept %[wt]\n"
                    "s_nop 0\n"
                    ".endr\n"
                    ".irp r,15,14,13,12,11,10,9,8,7,6,5,4,3,2,1,0\n v_mov_b32 v[132+\\r], v[100+\\r]\n .endr\n"
                    "s_nop 7\n"
                    "global_store_dwordx4 %[p], v[132:135], off\n"
                    "global_store_dwordx4 %[p], v[136:139], off offset:32\n"
                    "global_store_dwordx4 %[p], v[140:143], off offset:64\n"
                    "global_store_dwordx4 %[p], v[144:147], off offset:96\n"
                    "s_waitcnt vmcnt(0)\n"
                    :: [p] "v"(p), [good] "v"(good), [neg] "v"(neg), [one] "v"(onef), [wt] "n"(WAIT)
                    : "memory", CLOB, "v120", "v121", "v122", "v123", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139",
                      "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147");
            }
        } else {
            asm volatile(
                FILL("%[good]")
                "s_nop 7\n"
                STORES
                "s_waitcnt vmcnt(0)\n"
                FILL("%[neg]")
                :: [p] "v"(p), [good] "v"(good), [neg] "v"(neg)
                : "memory", CLOB);
        }
    }
}

// expected: MODE 0 -> 16 * (it + 1); else it + 1
__global__ void k_check(const float* buf, long n, int mode, unsigned long long* bad, unsigned long long* bad_hi_lanes) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % ROW_FLOATS);
        const int it = col / 32;
        const float expect = mode >= 6 ? (float)(it + 2) : (mode == 0 ? 16.0f : mode >= 4 ? 64.0f : 1.0f) * (float)(it + 1);
        if (buf[i] != expect) {
            atomicAdd(bad, 1ull);
            // which sample column (= lane & 31) of the wave's 32-row tile was it?
            const long r = i / ROW_FLOATS;
            if ((r & 31) >= 16) atomicAdd(bad_hi_lanes, 1ull);
        }
    }
}

template <int MODE, int WAIT>
static void run(float* buf, unsigned long long* cnt, int wps) {
    const int blocks = 256 * wps;                               // 4 waves per block: wps waves per SIMD, all co-resident
    const long n = (long)blocks * 4 * 32 * ROW_FLOATS;
    CK(hipMemset(buf, 0xff, n * sizeof(float)));
    CK(hipMemset(cnt, 0, 16));
    hipLaunchKernelGGL((k_probe<MODE, WAIT>), dim3(blocks), dim3(256), 0, 0, buf);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_check, dim3(1024), dim3(256), 0, 0, buf, n, MODE, cnt, cnt + 1);
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost));
    const char* names[10] = {"RAW mfma->store", "WAR store->valu", "WAR store->mfma", "WAR store->vmcnt(0)->valu", "RAW 4-mfma chain->store", "RAW 4-mfma->v_mov->store",
                             "WAR pk_add srcs<-v_mov", "WAR pk_add srcs<-lshl_add_u64", "RAW pk_add->store", "RAW pk_add->v_mov->store"};
    printf("  %-26s wait %2d, %d waves/SIMD: wrong elements %10llu of %ld (%.4f %%), in sample columns 16-31: %llu\n",
           names[MODE], WAIT, wps, h[0], n, 100.0 * (double)h[0] / (double)n, h[1]);
    fflush(stdout);
}

int main() {
    float* buf;
    unsigned long long* cnt;
    CK(hipMalloc(&buf, (size_t)256 * 3 * 4 * 32 * ROW_FLOATS * sizeof(float) + (1 << 20)));
    CK(hipMalloc(&cnt, 16));
    for (int wps = 1; wps <= 3; ++wps) {
        printf("%d waves per SIMD\n", wps);
        run<0, 0>(buf, cnt, wps); run<0, 2>(buf, cnt, wps); run<0, 4>(buf, cnt, wps); run<0, 6>(buf, cnt, wps); run<0, 8>(buf, cnt, wps);
        run<0, 10>(buf, cnt, wps); run<0, 12>(buf, cnt, wps); run<0, 16>(buf, cnt, wps); run<0, 20>(buf, cnt, wps); run<0, 32>(buf, cnt, wps);
        run<1, 0>(buf, cnt, wps); run<1, 1>(buf, cnt, wps); run<1, 2>(buf, cnt, wps); run<1, 4>(buf, cnt, wps); run<1, 8>(buf, cnt, wps);
        run<1, 16>(buf, cnt, wps); run<1, 64>(buf, cnt, wps); run<1, 256>(buf, cnt, wps);
        run<2, 0>(buf, cnt, wps); run<2, 1>(buf, cnt, wps); run<2, 2>(buf, cnt, wps); run<2, 4>(buf, cnt, wps); run<2, 8>(buf, cnt, wps);
        run<2, 16>(buf, cnt, wps); run<2, 64>(buf, cnt, wps);
        run<3, 0>(buf, cnt, wps);
        run<4, 0>(buf, cnt, wps); run<4, 4>(buf, cnt, wps); run<4, 8>(buf, cnt, wps); run<4, 12>(buf, cnt, wps); run<4, 16>(buf, cnt, wps);
        run<4, 20>(buf, cnt, wps); run<4, 32>(buf, cnt, wps); run<4, 64>(buf, cnt, wps); run<4, 128>(buf, cnt, wps);
        run<5, 0>(buf, cnt, wps); run<5, 4>(buf, cnt, wps); run<5, 12>(buf, cnt, wps);
        run<6, 0>(buf, cnt, wps); run<6, 1>(buf, cnt, wps); run<6, 2>(buf, cnt, wps); run<6, 4>(buf, cnt, wps);
        run<7, 0>(buf, cnt, wps); run<7, 1>(buf, cnt, wps); run<7, 2>(buf, cnt, wps); run<7, 4>(buf, cnt, wps);
        run<8, 0>(buf, cnt, wps); run<8, 1>(buf, cnt, wps); run<8, 2>(buf, cnt, wps); run<8, 4>(buf, cnt, wps);
        run<9, 0>(buf, cnt, wps); run<9, 1>(buf, cnt, wps); run<9, 2>(buf, cnt, wps); run<9, 4>(buf, cnt, wps);
    }
    return 0;
}
