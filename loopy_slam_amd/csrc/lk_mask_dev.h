// Inside-mask threshold thr = min(10 * median(d > 0), 1.2 * max(d)) of a ray batch (Tracker.py:153-155, Mapper.py:674-676)
// as a device function of ONE 1024-thread workgroup, shared by k_inside_mask (lk_optim.hip) and the fused batch-assembly
// kernel of the tracking loop (lk_loop.hip).  Median = 4-pass radix select over the bit patterns of the positive depths.
// REG = true (n <= LK_MASK_REG_MAX): the values stay in registers between the passes; otherwise they are re-read from
// `scratch`.  The top byte of a depth takes a handful of values, so in the first pass the histogram is built with
// one LDS add per (wave, distinct byte) instead of one same-address add per ray.
#pragma once
#include "lk_common.h"

#define LK_MASK_REG_MAX 8192
#define LK_MASK_VPT (LK_MASK_REG_MAX / 1024)
// the batch-assembly kernel of the per-frame loops also comes in a 16-values-per-thread form (k_pregather<16>: the 10 000-ray mapping
// batches of the TUM / ScanNet configs)
#define LK_LOOP_MAX_R 16384
struct LkMaskShared { unsigned hist[256]; unsigned s_prefix, s_rank, s_cnt, s_maxbits; };

// u[q] = bit pattern of depth (t + 1024 q) if positive else 0 (REG) / scratch[i] likewise (!REG); mycnt / mymax = this
// thread's count of positive depths and their largest bit pattern.  Returns thr; *any = false if no depth is positive.
template <bool REG, int VPT = LK_MASK_VPT>
__device__ __forceinline__ float lk_inside_thr(const unsigned (&u)[VPT], const uint32_t* __restrict__ scratch, int n,
                                               unsigned mycnt, unsigned mymax, LkMaskShared& S, bool* any) {
    const int t = threadIdx.x, lane = t & 63;
    if (t == 0) { S.s_cnt = 0; S.s_maxbits = 0; }
    __syncthreads();
    // one LDS atomic per WAVE (1024 same-address atomics cost microseconds)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mycnt += __shfl_xor(mycnt, o); mymax = max(mymax, (unsigned)__shfl_xor((int)mymax, o)); }
    if (lane == 0) { atomicAdd(&S.s_cnt, mycnt); atomicMax(&S.s_maxbits, mymax); }
    __syncthreads();
    const unsigned m = S.s_cnt;
    *any = m != 0;
    if (m == 0) return 0.0f;
    // Shortcut: thr = min(10*median, 1.2*max) is 1.2*max unless the median itself satisfies fl(10 v) < fl(1.2 max); that
    // predicate is monotone in v, so the (lower) median at rank (m-1)/2 satisfies it iff MORE than (m-1)/2 values do -
    // one counting pass instead of the 4-pass radix select (depth images: median ~ max/2, the select almost never runs).
    const float mx12 = __fmul_rn(1.2f, __uint_as_float(S.s_maxbits));
    {
        unsigned below = 0;
        if (REG) {
#pragma unroll
            for (int q = 0; q < VPT; ++q) below += (u[q] && __fmul_rn(10.0f, __uint_as_float(u[q])) < mx12) ? 1u : 0u;
        } else {
            for (int i = t; i < n; i += 1024) { const unsigned v = scratch[i]; below += (v && __fmul_rn(10.0f, __uint_as_float(v)) < mx12) ? 1u : 0u; }
        }
        if (t == 0) S.s_rank = 0;
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
        if (lane == 0) atomicAdd(&S.s_rank, below);
        __syncthreads();
    }
    const bool need_median = S.s_rank > (m - 1) / 2;                    // block-uniform
    __syncthreads();
    if (t == 0) { S.s_prefix = 0; S.s_rank = (m - 1) / 2; }              // torch.median: lower of the two middle values
    __syncthreads();
    for (int shift = 24; need_median && shift >= 0; shift -= 8) {
        if (t < 256) S.hist[t] = 0;
        __syncthreads();
        const unsigned prefix = S.s_prefix;
        const unsigned himask = (shift == 24) ? 0u : (0xffffffffu << (shift + 8));
        if (REG) {
#pragma unroll
            for (int q = 0; q < VPT; ++q) {
                const bool on = u[q] && (u[q] & himask) == prefix;
                const unsigned digit = (u[q] >> shift) & 255u;
                if (shift == 24) {      // all lanes walk the loop together (wave-uniform trip count)
                    unsigned long long pending = __ballot(on);
                    while (pending) {
                        const int leader = __ffsll((long long)pending) - 1;
                        const unsigned dl = __shfl(digit, leader);
                        const unsigned long long same = __ballot(on && digit == dl);
                        if (lane == leader) atomicAdd(&S.hist[dl], (unsigned)__popcll(same));
                        pending &= ~same;
                    }
                } else if (on) {
                    atomicAdd(&S.hist[digit], 1u);
                }
            }
        } else {
            for (int i = t; i < n; i += 1024) {
                const unsigned v = scratch[i];
                if (v && (v & himask) == prefix) atomicAdd(&S.hist[(v >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        // locate the bin that holds the wanted rank: wave 0 scans the 256-bin histogram (4 bins per lane)
        if (t < 64) {
            const unsigned rank = S.s_rank;                   // every lane reads before the (later) single write
            const unsigned h0 = S.hist[4 * t], h1 = S.hist[4 * t + 1], h2 = S.hist[4 * t + 2], h3 = S.hist[4 * t + 3];
            const unsigned tot = h0 + h1 + h2 + h3;
            unsigned incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned nbv = __shfl_up(incl, o);
                if (t >= o) incl += nbv;
            }
            const unsigned excl = incl - tot;
            if (rank >= excl && rank < incl) {              // exactly one lane
                unsigned r = rank - excl, b = 4 * t;
                if (r >= h0) { r -= h0; ++b; if (r >= h1) { r -= h1; ++b; if (r >= h2) { r -= h2; ++b; } } }
                S.s_rank = r;
                S.s_prefix = prefix | (b << shift);
            }
        }
        __syncthreads();
    }
    const float med = __uint_as_float(S.s_prefix);
    return need_median ? fminf(__fmul_rn(10.0f, med), mx12) : mx12;
}

// ------------------------------------------------------------------------------------------------------------------------
// tracking.handle_dynamic: False (Tracker.py:177-179): the outlier mask of the tracker loss is |gt - depth| < 10 * median(|gt - depth|)
// instead of the uncertainty-normalised residual against 10 x its mean.  resid[i] >= 0 (or NaN) for a present ray, sign bit set for
// an absent one (gt_depth <= 0: the reference filters those before it renders).  ONE 1024-thread workgroup: 4-pass radix select
// over the bit patterns (non-negative floats order as unsigned integers), values re-read from memory in every pass.
// torch.median: the LOWER of the two middle values (rank (m - 1) / 2), NaN if any element is NaN (the mask is then empty).
// No present ray: 0 (empty mask).
struct LkMedianShared { unsigned hist[256]; unsigned s_prefix, s_rank, s_cnt, s_nan; };
__device__ __forceinline__ float lk_block_median10(const float* __restrict__ resid, int n, LkMedianShared& S) {
    const int t = threadIdx.x, lane = t & 63, nt = blockDim.x;
    if (t == 0) { S.s_cnt = 0; S.s_nan = 0; }
    __syncthreads();
    unsigned cnt = 0, nan = 0;
    for (int i = t; i < n; i += nt) {
        const unsigned v = __float_as_uint(resid[i]);
        if (!(v & 0x80000000u)) { ++cnt; nan += (v > 0x7f800000u) ? 1u : 0u; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o); nan += __shfl_xor(nan, o); }
    if (lane == 0) { atomicAdd(&S.s_cnt, cnt); atomicAdd(&S.s_nan, nan); }
    __syncthreads();
    const unsigned m = S.s_cnt;
    if (m == 0) return 0.0f;
    if (S.s_nan != 0) return __uint_as_float(0x7fc00000u);
    __syncthreads();
    if (t == 0) { S.s_prefix = 0; S.s_rank = (m - 1) / 2; }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = t; b < 256; b += nt) S.hist[b] = 0;
        __syncthreads();
        const unsigned prefix = S.s_prefix;
        const unsigned himask = (shift == 24) ? 0u : (0xffffffffu << (shift + 8));
        for (int i0 = 0; i0 < n; i0 += nt) {         // workgroup-uniform trip count: every lane takes part in the ballots
            const unsigned v = (i0 + t < n) ? __float_as_uint(resid[i0 + t]) : 0x80000000u;
            const bool on = !(v & 0x80000000u) && (v & himask) == prefix;
            const unsigned digit = (v >> shift) & 255u;
            // one LDS add per (wave, distinct digit): residuals share their top bytes, same-address adds would serialise
            unsigned long long pending = __ballot(on);
            while (pending) {
                const int leader = __ffsll((long long)pending) - 1;
                const unsigned dl = __shfl(digit, leader);
                const unsigned long long same = __ballot(on && digit == dl);
                if (lane == leader) atomicAdd(&S.hist[dl], (unsigned)__popcll(same));
                pending &= ~same;
            }
        }
        __syncthreads();
        if (t < 64) {
            const unsigned rank = S.s_rank;
            const unsigned h0 = S.hist[4 * t], h1 = S.hist[4 * t + 1], h2 = S.hist[4 * t + 2], h3 = S.hist[4 * t + 3];
            const unsigned tot = h0 + h1 + h2 + h3;
            unsigned incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned nbv = __shfl_up(incl, o);
                if (t >= o) incl += nbv;
            }
            const unsigned excl = incl - tot;
            if (rank >= excl && rank < incl) {
                unsigned r = rank - excl, b = 4 * t;
                if (r >= h0) { r -= h0; ++b; if (r >= h1) { r -= h1; ++b; if (r >= h2) { r -= h2; ++b; } } }
                S.s_rank = r;
                S.s_prefix = prefix | (b << shift);
            }
        }
        __syncthreads();
    }
    return __fmul_rn(10.0f, __uint_as_float(S.s_prefix));
}
