// tools/probe/stream_ceiling.hip (round 6): what a launch-sized streaming kernel can reach on this chip - the ceiling k_wgrad (148 MB read per launch) and the training
// forward (115 MB written per launch) are priced against.  hipcc --offload-arch=gfx950 -O3 -o stream_ceiling stream_ceiling.hip && ./stream_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// contiguous: a wave walks 1-KiB pieces (64 lanes x 16 B), DEPTH loads in flight
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ src, size_t n4, float* __restrict__ out) {
    const size_t wave = (size_t)blockIdx.x * 4 + threadIdx.x / 64, nw = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    size_t i = wave * 64 + lane;
    const size_t step = nw * 64;
    for (; i + (DEPTH - 1) * step < n4; i += DEPTH * step) {
        float4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = src[i + d * step];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d].x + v[d].y + v[d].z + v[d].w;
    }
    for (; i < n4; i += step) { float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
// half rows: rows of 512 B, a wave reads the 256-B half `h` of 2 x DEPTH consecutive... (8 B per lane, 32 lanes per row half, two rows per load) - k_wgrad's pattern
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read_half(const float2* __restrict__ src, size_t rows, float* __restrict__ out) {
    const size_t wave = (size_t)blockIdx.x * 4 + threadIdx.x / 64, nw = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    const int h = (int)(wave & 1);                      // which half of the row
    const size_t w2 = wave >> 1, nw2 = nw >> 1;
    float acc = 0.f;
    // row pair p covers rows 2p, 2p+1: lane l reads row 2p + (l >> 5), float2 index (l & 31) + 32 h of the 64 float2 of a row
    const size_t pairs = rows / 2;
    for (size_t p = w2 * DEPTH; p + DEPTH <= pairs; p += nw2 * DEPTH) {
        float2 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = src[(2 * (p + d) + (lane >> 5)) * 64 + (lane & 31) + 32 * h];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d].x + v[d].y;
    }
    if (acc == 123.456f) out[0] = acc;
}
// ... and k_wgrad's sharing: the four waves of a workgroup are four units over the SAME rows, every half row is requested by two of them
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read_half_dup(const float2* __restrict__ src, size_t rows, float* __restrict__ out) {
    const int w = threadIdx.x / 64, lane = threadIdx.x & 63;
    const int h = w & 1;                                 // waves 0, 2 -> half 0; waves 1, 3 -> half 1 (each half twice per workgroup)
    float acc = 0.f;
    const size_t pairs = rows / 2;
    for (size_t p = (size_t)blockIdx.x * DEPTH; p + DEPTH <= pairs; p += (size_t)gridDim.x * DEPTH) {
        float2 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = src[(2 * (p + d) + (lane >> 5)) * 64 + (lane & 31) + 32 * h];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d].x + v[d].y;
    }
    if (acc == 123.456f) out[0] = acc;
}
// ... and as k_wgrad has it since the XCD-aware ownership: the two waves that request a row half sit in DIFFERENT workgroups of the same XCD (b and b + gridDim / 2)
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read_half_dup_far(const float2* __restrict__ src, size_t rows, float* __restrict__ out) {
    const int w = threadIdx.x / 64, lane = threadIdx.x & 63;
    const size_t half_grid = gridDim.x / 2;
    const size_t b = blockIdx.x % half_grid;              // both copies walk the same row pairs
    const int h = (w & 1);
    float acc = 0.f;
    const size_t pairs = rows / 2;
    // a workgroup takes two consecutive groups of DEPTH row pairs, waves (0, 1) the two halves of the first, (2, 3) of the second; the far copy (b + gridDim / 2) reads the same
    for (size_t p = (b * 2 + (w >> 1)) * DEPTH; p + DEPTH <= pairs; p += half_grid * 2 * DEPTH) {
        float2 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = src[(2 * (p + d) + (lane >> 5)) * 64 + (lane & 31) + 32 * h];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d].x + v[d].y;
    }
    if (acc == 123.456f) out[0] = acc;
}
// the same rows staged ONCE per workgroup: every wave loads a quarter of the chunk (contiguous 16-B loads), LDS hands the halves out
template <int ROWS>
__global__ __launch_bounds__(256) void k_read_lds(const float4* __restrict__ src, size_t rows, float* __restrict__ out) {
    __shared__ float4 buf[2][ROWS * 32];
    const int t = threadIdx.x, w = t / 64, lane = t & 63, h = w & 1;
    float acc = 0.f;
    const size_t chunks = rows / ROWS;
    size_t c = blockIdx.x;
    float4 v[ROWS * 32 / 256];
    auto fetch = [&](size_t cc) {
#pragma unroll
        for (int k = 0; k < ROWS * 32 / 256; ++k) v[k] = src[cc * (ROWS * 32) + k * 256 + t];
    };
    if (c < chunks) fetch(c);
    int par = 0;
    for (; c < chunks; c += gridDim.x, par ^= 1) {
#pragma unroll
        for (int k = 0; k < ROWS * 32 / 256; ++k) buf[par][k * 256 + t] = v[k];
        if (c + gridDim.x < chunks) fetch(c + gridDim.x);
        __syncthreads();
        const float2* b2 = reinterpret_cast<const float2*>(buf[par]);
#pragma unroll
        for (int p = 0; p < ROWS / 2; ++p) { const float2 x = b2[(2 * p + (lane >> 5)) * 64 + (lane & 31) + 32 * h]; acc += x.x + x.y; }
    }
    if (acc == 123.456f) out[0] = acc;
}
template <int DEPTH>
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ dst, size_t n4) {
    const size_t wave = (size_t)blockIdx.x * 4 + threadIdx.x / 64, nw = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    const size_t step = nw * 64;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)lane);
    for (size_t i = wave * 64 + lane; i < n4; i += step) dst[i] = v;
}
// pieces: as the decoder's saves - a wave stores 128-B pieces (32 lanes x 4 B ... here 8 lanes x 16 B) 512 B apart
__global__ __launch_bounds__(256) void k_write_pieces(float4* __restrict__ dst, size_t n4) {
    const size_t wave = (size_t)blockIdx.x * 4 + threadIdx.x / 64, nw = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)lane);
    // a wave owns 32 rows x one 128-B quarter per store: lane l -> row (l >> 3), float4 (l & 7) of quarter q; 8 rows per instruction
    const size_t rows = n4 / 32;                     // rows of 512 B = 32 float4
    for (size_t r0 = (wave >> 2) * 32; r0 + 32 <= rows; r0 += (nw >> 2) * 32) {
        const int q = (int)(wave & 3);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[(r0 + 8 * k + (lane >> 3)) * 32 + q * 8 + (lane & 7)] = v;
    }
}

int main() {
    const size_t MB = 1 << 20;
    const size_t big = 1024 * MB;
    float4* buf; float* out;
    CK(hipMalloc(&buf, big)); CK(hipMalloc(&out, 4));
    CK(hipMemset(buf, 0, big));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, size_t bytes, int nbuf, auto launch) {
        // nbuf buffers of `bytes` walked round robin: nbuf x bytes > 256 MB = colder than the Infinity Cache
        for (int w = 0; w < 3; ++w) launch(buf + (size_t)(w % nbuf) * (bytes / 16));
        hipDeviceSynchronize();
        const int N = 24;
        hipEventRecord(e0);
        for (int k = 0; k < N; ++k) launch(buf + (size_t)(k % nbuf) * (bytes / 16));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-46s %6.1f MB x %d buffers: %7.1f us / launch  %6.2f TB/s\n", name, bytes / 1e6, nbuf, ms * 1e3 / N, bytes / (ms * 1e-3 / N) / 1e12);
    };
    for (size_t mb : {148}) {
        const size_t bytes = mb * 1000 * 1000 / 4096 * 4096;
        for (int nbuf : {1, (int)(big / bytes)}) {
            for (int wgs : {512, 1024, 2048, 8192}) {
                char nm[96];
                snprintf(nm, sizeof nm, "read contiguous depth 8, %d workgroups", wgs);
                run(nm, bytes, nbuf, [&](float4* p) { hipLaunchKernelGGL(k_read<8>, dim3(wgs), dim3(256), 0, 0, p, bytes / 16, out); });
            }
            run("read contiguous depth 16, 512 workgroups", bytes, nbuf, [&](float4* p) { hipLaunchKernelGGL(k_read<16>, dim3(512), dim3(256), 0, 0, p, bytes / 16, out); });
            run("read half rows depth 16, 512 workgroups", bytes, nbuf, [&](float4* p) { hipLaunchKernelGGL(k_read_half<16>, dim3(512), dim3(256), 0, 0, (const float2*)p, bytes / 512, out); });
            run("read half rows depth 30, 512 workgroups", bytes, nbuf, [&](float4* p) { hipLaunchKernelGGL(k_read_half<30>, dim3(512), dim3(256), 0, 0, (const float2*)p, bytes / 512, out); });
            run("read half rows TWICE per workgroup depth 16, 512", bytes, nbuf, [&](float4* p) { hipLaunchKernelGGL(k_read_half_dup<16>, dim3(512), dim3(256), 0, 0, (const float2*)p, bytes / 512, out); });
            run("read half rows TWICE per workgroup depth 8, 512", bytes, nbuf, [&](float4* p) { hipLaunchKernelGGL(k_read_half_dup<8>, dim3(512), dim3(256), 0, 0, (const float2*)p, bytes / 512, out); });
            run("read half rows ONCE, 1024 workgroups (each half by one wave)", bytes, nbuf, [&](float4* p) { hipLaunchKernelGGL(k_read_half<16>, dim3(1024), dim3(256), 0, 0, (const float2*)p, bytes / 512, out); });
            run("read every half row TWICE from far workgroups depth 16, 512", bytes, nbuf, [&](float4* p) { hipLaunchKernelGGL(k_read_half_dup_far<16>, dim3(512), dim3(256), 0, 0, (const float2*)p, bytes / 512, out); });
            run("read once into LDS (32-row chunks), 512", bytes, nbuf, [&](float4* p) { hipLaunchKernelGGL(k_read_lds<32>, dim3(512), dim3(256), 0, 0, p, bytes / 512, out); });
            run("read once into LDS (32-row chunks), 768", bytes, nbuf, [&](float4* p) { hipLaunchKernelGGL(k_read_lds<32>, dim3(768), dim3(256), 0, 0, p, bytes / 512, out); });
            run("read once into LDS (64-row chunks), 512", bytes, nbuf, [&](float4* p) { hipLaunchKernelGGL(k_read_lds<64>, dim3(512), dim3(256), 0, 0, p, bytes / 512, out); });
            for (int wgs : {512, 768, 2048, 8192}) {
                char nm[96];
                snprintf(nm, sizeof nm, "write contiguous, %d workgroups", wgs);
                run(nm, bytes, nbuf, [&](float4* p) { hipLaunchKernelGGL(k_write<1>, dim3(wgs), dim3(256), 0, 0, p, bytes / 16); });
            }
            run("write 128-B pieces 512 B apart, 768 workgroups", bytes, nbuf, [&](float4* p) { hipLaunchKernelGGL(k_write_pieces, dim3(768), dim3(256), 0, 0, p, bytes / 16); });
        }
    }
    return 0;
}
