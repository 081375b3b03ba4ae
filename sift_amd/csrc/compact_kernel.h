// compact_kernel.h -- 7x7 cull on the sparse candidate list, border filter and raster-ordered
// compaction into Keypoint_t[max_kp].
//
// Restates culling_hls + IsBestKeypoint (DetectKeypoints.cpp:567-750) and combine_keypoints_hls
// (:372-439).  Pre-cull candidates are ~0.6 % of the frame, so everything after k_detect works
// on its list and bit masks instead of on dense planes.  The compaction is rank based (per-row counts, then an in-row wave scan), which
// reproduces the reference's raster order and its "first max_kp in raster order" cap without a
// sort.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "detect_kernel.h"

namespace sift {


struct kp_t { uint16_t x, y; uint32_t contrast; };   // Keypoint_t, SIFTParameters.hpp:79-85

// bits [x0, x0+7) of row mask `m` (S segments), out-of-range bits read 0.  Branch-free: one
// unaligned 2-byte load from a clamped byte offset (the mask is little-endian, so bit b of the
// row is bit b & 7 of byte b >> 3); bits left of the row are shifted in as zeros, bits right of
// it shifted out of the 16 loaded ones.  Calls for several rows overlap.
__device__ __forceinline__ uint32_t mask_field7(const u64 *__restrict__ m, int S, int x0)
{
    const int bo = min(max(x0 >> 3, 0), S * 8 - 2);  // x0 in [-3, 64 S + 2]
    const int sh = x0 - bo * 8;                      // -3 .. 18
    uint16_t v16;
    __builtin_memcpy(&v16, reinterpret_cast<const char *>(m) + bo, 2);
    const uint32_t v = v16;
    return (sh >= 0 ? v >> sh : v << -sh) & 0x7fu;
}

// (:393) keep only 71 < y < H-71 and 71 < x < W-71
__device__ __forceinline__ u64 border_mask(int y, int seg, int W, int H)
{
    if (!(y > 71 && y < H - 71)) return 0ull;
    const int lo = 72 - seg * 64;                    // first kept bit
    const int hi = (W - 71) - seg * 64;              // one past the last kept bit
    if (hi <= 0 || lo >= 64) return 0ull;
    u64 m = ~0ull;
    if (lo > 0) m &= (~0ull) << lo;
    if (hi < 64) m &= (~0ull) >> (64 - hi);
    return m;
}

// 7x7 cull, one lane per pre-cull candidate of k_detect's list.  A candidate survives iff its
// Q15 edge ratio is strictly smaller than that of every OTHER candidate in the 7x7 window
// (:728-731, :684; neighbours are pre-cull values).  Survivors get their bit in survmask and,
// inside the border, count in rowcount[y] (both zeroed by k_detect).
// grid: x = blocks of 256 (grid-stride over the list), y = frames.
__global__ __launch_bounds__(256) void k_cull(const u64 *__restrict__ candmask, const uint32_t *__restrict__ keyplane,
                                              const uint32_t *__restrict__ list,
                                              const uint32_t *__restrict__ listcount, u64 *__restrict__ survmask,
                                              uint32_t *__restrict__ rowcount, int W, int H, int S, int list_cap, int Wl)
{
    const int frame = blockIdx.y;
    const size_t fpix = (size_t)frame * (size_t)W * (size_t)H;
    const uint32_t *__restrict__ kp = keyplane + fpix;
    const size_t frow = (size_t)frame * H;
    const u64 *__restrict__ cm = candmask + frow * S;
    const uint32_t n = min(listcount[frame], (uint32_t)list_cap);
    const uint32_t *__restrict__ fl = list + (size_t)frame * list_cap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t e = fl[i];
        const int y = (int)(e >> 16), x = (int)(e & 0xffffu);
        const uint32_t ec = kp[(size_t)y * W + x] >> 8;
        // candidates in the 7x7 window: 7 bits per row, row dy+3 at bits 7*(dy+3)
        u64 nb = 0ull;
#pragma unroll
        for (int dy = -3; dy <= 3; dy++) {
            const int yy = y + dy;
            const bool ok = yy >= 0 && yy < H;
            const uint32_t f = mask_field7(cm + (size_t)(ok ? yy : 0) * S, S, x - 3);
            nb |= (u64)(ok ? f : 0u) << (7 * (dy + 3));
        }
        nb &= ~(1ull << 24);                                         // not the keypoint itself (:728)
        // the neighbours' keys, four loads in flight (a lane rarely has more neighbours)
        bool best = true;
        while (nb && best) {
            uint32_t en[4];
            bool has[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                has[u] = nb != 0ull;
                const int k = has[u] ? __ffsll((long long)nb) - 1 : 24;   // 24: the keypoint itself
                nb &= nb - 1;
                en[u] = kp[(size_t)(y + k / 7 - 3) * W + (x + k % 7 - 3)] >> 8;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (has[u] && !(ec < en[u])) best = false;           // strict (:730)
        }
        if (best) {
            const u64 bit = 1ull << (x & 63);
            atomicOr(survmask + (frow + y) * S + (x >> 6), bit);
            if (border_mask(y, x >> 6, Wl, H) & bit) atomicAdd(rowcount + frow + y, 1u);   // the border rule counts from the logical width
        }
    }
}

// Exclusive prefix of the row counts: rowbase[y] = keypoints in rows above y, rowbase[H] = the frame's total.
// One block per frame; chunks of 1024 rows, block scan per chunk.
// grid: x = frames; block = 1024
__global__ __launch_bounds__(1024) void k_rowprefix(const uint32_t *__restrict__ rowcount, uint32_t *__restrict__ rowbase, int H)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t *__restrict__ rc = rowcount + (size_t)blockIdx.x * H;
    uint32_t *__restrict__ rb = rowbase + (size_t)blockIdx.x * (H + 1);
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    for (int y0 = 0; y0 <= H; y0 += 1024) {
        const int y = y0 + (int)threadIdx.x;
        const uint32_t n = y < H ? rc[y] : 0u;
        uint32_t inc = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t before = carry;
        for (int w = 0; w < wave; w++) before += wsum[w];
        if (y <= H) rb[y] = before + inc - n;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + inc;
        __syncthreads();
    }
}

// One wave per image row (+ one extra wave per frame for padding / totals: total_ctx is the
// context's own copy of the uncapped count, which the fused describe kernel reads; total_out the
// caller's, optional).  rowbase: k_rowprefix's exclusive prefix of the row counts.
// grid: x = ceil((H+1)/4) blocks of 256, y = frames.
__global__ __launch_bounds__(256) void k_emit(const u64 *__restrict__ survmask,
                                              const uint32_t *__restrict__ keyplane,
                                              const uint32_t *__restrict__ rowbase, kp_t *__restrict__ kps,
                                              uint32_t *__restrict__ total_ctx, uint32_t *__restrict__ total_out,
                                              int W, int H, int S, int max_kp, int Wl)
{
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (y > H) return;
    const size_t frow = (size_t)blockIdx.y * H;
    const uint32_t *__restrict__ rb = rowbase + (size_t)blockIdx.y * (H + 1);
    kp_t *__restrict__ out = kps + (size_t)blockIdx.y * max_kp;

    // keypoints in the rows above (k_rowprefix; y == H: the frame total) and, requested together with
    // them, the row's first 64 mask segments, so that the wave waits for memory once before it reads the keys
    uint32_t base = rb[y];
    u64 m0 = 0ull;
    if (y < H) {
        if (lane < S) m0 = survmask[(frow + y) * S + lane];
        if (rb[y + 1] == base) return;               // nothing to emit from this row
    }

    if (y == H) {
        // padding (:428-436) and the uncapped total (nKeypointsTotal, :405-406)
        if (lane == 0) {
            total_ctx[blockIdx.y] = base;
            if (total_out) total_out[blockIdx.y] = base;
        }
        const uint32_t first = base < (uint32_t)max_kp ? base : (uint32_t)max_kp;
        u64 *o64 = reinterpret_cast<u64 *>(out);
        for (uint32_t i = first + lane; i < (uint32_t)max_kp; i += 64) o64[i] = 0ull;
        return;
    }
    if (base >= (uint32_t)max_kp) return;            // cap reached above this row (:409)

    const uint32_t *__restrict__ kp = keyplane + (size_t)blockIdx.y * (size_t)W * (size_t)H + (size_t)y * W;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int seg = s0 + lane;
        u64 m = 0ull;
        if (seg < S) m = (s0 == 0 ? m0 : survmask[(frow + y) * S + seg]) & border_mask(y, seg, Wl, H);
        const uint32_t n = (uint32_t)__popcll(m);
        // inclusive wave scan of n
        uint32_t inc = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        uint32_t idx = base + inc - n;
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (idx < (uint32_t)max_kp) {
                const int x = seg * 64 + b;
                kp_t k;
                k.x = (uint16_t)x;
                k.y = (uint16_t)y;
                k.contrast = kp[x] & 0xffu;
                out[idx] = k;
            }
            idx++;
        }
        base += __shfl(inc, 63);
    }
}

// Debug: expand masks + keys to dense maps (sift_hip_debug_planes).
__global__ void k_expand_maps(const u64 *__restrict__ candmask, const u64 *__restrict__ survmask,
                              const uint32_t *__restrict__ keyplane, uint16_t *__restrict__ cand,
                              int32_t *__restrict__ eratio, uint16_t *__restrict__ surv, int W, int H, int S, int Wl)
{
    // W: pitch of the key plane; Wl: logical width = pitch of the dense maps written here
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= Wl || y >= H) return;
    const size_t i = (size_t)y * Wl + x;
    const bool c = (candmask[(size_t)y * S + (x >> 6)] >> (x & 63)) & 1ull;
    const bool s = (survmask[(size_t)y * S + (x >> 6)] >> (x & 63)) & 1ull;
    const uint32_t key = c ? keyplane[(size_t)y * W + x] : 0u;
    if (cand) cand[i] = c ? (uint16_t)(key & 0xffu) : 0;
    if (eratio) eratio[i] = c ? (int32_t)(key >> 8) : 0x7fffffff;
    if (surv) surv[i] = s ? (uint16_t)(key & 0xffu) : 0;
}

}  // namespace sift
