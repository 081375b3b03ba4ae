// compact_kernel.h -- edge test and 7x7 cull on the sparse candidate list, border filter and
// raster-ordered compaction into Keypoint_t[max_kp].
//
// Restates IsOnEdge (DetectKeypoints.cpp:314-352), culling_hls + IsBestKeypoint (:567-750) and
// combine_keypoints_hls (:372-439).  The pixels that pass the extremum and contrast tests are
// ~1 % of the frame, so everything after k_detect works on its list and bit masks instead of on
// dense planes.  The compaction is rank based (per-row counts, then an in-row wave scan), which
// reproduces the reference's raster order and its "first max_kp in raster order" cap without a
// sort.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "detect_kernel.h"

namespace sift {


struct kp_t { uint16_t x, y; uint32_t contrast; };   // Keypoint_t, SIFTParameters.hpp:79-85

// bits [x0, x0+7) of row mask `m` (S segments), out-of-range bits read 0
__device__ __forceinline__ uint32_t mask_field7(const u64 *__restrict__ m, int S, int x0)
{
    const int sa = (x0 >= 0) ? (x0 >> 6) : -1;
    const int off = x0 - sa * 64;                    // 0..63
    const u64 a = (sa >= 0 && sa < S) ? m[sa] : 0ull;
    u64 f = a >> off;
    if (off > 57) {                                  // field straddles into the next segment
        const u64 b = (sa + 1 >= 0 && sa + 1 < S) ? m[sa + 1] : 0ull;
        f |= b << (64 - off);
    }
    return (uint32_t)(f & 0x7full);
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

// D2 = G2 - G1 at (y, x) from the planar pyr plane, 0 outside the image
__device__ __forceinline__ int pyr_d2_at(const uint4 *__restrict__ fp, int W4, int W, int H, int y, int x)
{
    if (y < 0 || y >= H || x < 0 || x >= W) return 0;
    const uint32_t *q = reinterpret_cast<const uint32_t *>(fp + (size_t)y * W4 + (x >> 2));
    const int sh = (x & 3) * 8;
    return (int)((q[2] >> sh) & 0xffu) - (int)((q[1] >> sh) & 0xffu);
}

// IsOnEdge (:314-352) + float -> ap_fixed<32,17,AP_RND_ZERO>: Q15 ratio or -1.
__device__ __forceinline__ int edge_ratio_q15(int L, int R, int U, int Dn, int UL, int UR, int DL, int DR, int c)
{
    const int fxx = L + R - 2 * c;
    const int fyy = U + Dn - 2 * c;
    const int fxy = UL + DR - DL - UR;
    const int tr = fxx + fyy;
    const int det = fxx * fyy - fxy * fxy;
    if (det == 0) return -1;
    // both operands are exact in binary32 (|.| < 2^24); __fdiv_rn is the IEEE
    // correctly rounded divide whatever the fast-math flags are
    const float r = __fdiv_rn((float)(tr * tr), (float)det);
    if (r < 0.0f || r > 12.0f) return -1;            // EIGEN_RATIO_TH, DetectKeypoints.hpp:18
    // r * 2^15 is exact in binary32 (r <= 12); round to nearest, ties toward zero
    const float v = r * 32768.0f;
    const float fl = floorf(v);
    return (int)fl + ((v - fl) > 0.5f ? 1 : 0);
}

// Q15 edge ratio of the pixel (y, x), -1 if the edge test rejects it; *contrast = |D2|
__device__ __forceinline__ int edge_at(const uint4 *__restrict__ fp, int W4, int W, int H, int y, int x, int *contrast)
{
    const int c = pyr_d2_at(fp, W4, W, H, y, x);
    if (contrast) *contrast = c < 0 ? -c : c;
    return edge_ratio_q15(pyr_d2_at(fp, W4, W, H, y, x - 1), pyr_d2_at(fp, W4, W, H, y, x + 1),
                          pyr_d2_at(fp, W4, W, H, y - 1, x), pyr_d2_at(fp, W4, W, H, y + 1, x),
                          pyr_d2_at(fp, W4, W, H, y - 1, x - 1), pyr_d2_at(fp, W4, W, H, y - 1, x + 1),
                          pyr_d2_at(fp, W4, W, H, y + 1, x - 1), pyr_d2_at(fp, W4, W, H, y + 1, x + 1), c);
}

// Edge test + 7x7 cull, one lane per pixel of k_detect's list (strict extrema above the contrast
// threshold, ~1 % of the pixels).  A listed pixel whose edge ratio is valid is a pre-cull
// candidate (kpMap1 != 0, :190-207): its bit is set in candmask and keyplane gets
// (E_q15 << 8) | contrast.  It survives the cull iff its ratio is strictly smaller than that of
// every OTHER pre-cull candidate in the 7x7 window (:728-731, :684); the ratios of the listed
// pixels in the window are recomputed here instead of being exchanged between lanes.
// Survivors get their bit in survmask and, inside the border, count in rowcount[y].
// grid: x = blocks of 256 (grid-stride over the list), y = frames.
__global__ __launch_bounds__(256) void k_cand(const uint4 *__restrict__ pyr, const u64 *__restrict__ premask,
                                              const uint32_t *__restrict__ list,
                                              const uint32_t *__restrict__ listcount, u64 *__restrict__ candmask,
                                              u64 *__restrict__ survmask, uint32_t *__restrict__ keyplane,
                                              uint32_t *__restrict__ rowcount, int W, int H, int S, int list_cap)
{
    const int frame = blockIdx.y;
    const int W4 = W >> 2;
    const size_t fpix = (size_t)frame * (size_t)W * (size_t)H;
    const uint4 *__restrict__ fp = pyr + (fpix >> 2);
    const size_t frow = (size_t)frame * H;
    const u64 *__restrict__ pm = premask + frow * S;
    const uint32_t n = min(listcount[frame], (uint32_t)list_cap);
    const uint32_t *__restrict__ fl = list + (size_t)frame * list_cap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t e = fl[i];
        const int y = (int)(e >> 16), x = (int)(e & 0xffffu);
        int ac;
        const int ec = edge_at(fp, W4, W, H, y, x, &ac);
        if (ec < 0) continue;                                        // :200 -- not a candidate
        const u64 bit = 1ull << (x & 63);
        atomicOr(candmask + (frow + y) * S + (x >> 6), bit);
        keyplane[fpix + (size_t)y * W + x] = ((uint32_t)ec << 8) | (uint32_t)ac;
        bool best = true;
        for (int dy = -3; dy <= 3 && best; dy++) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            uint32_t f = mask_field7(pm + (size_t)yy * S, S, x - 3);
            if (dy == 0) f &= ~8u;                                   // not the keypoint itself (:728)
            while (f) {
                const int k = __ffs((int)f) - 1;
                f &= f - 1;
                const int en = edge_at(fp, W4, W, H, yy, x - 3 + k, nullptr);
                if (en >= 0 && !(ec < en)) { best = false; break; }  // strict (:730)
            }
        }
        if (best) {
            atomicOr(survmask + (frow + y) * S + (x >> 6), bit);
            if (border_mask(y, x >> 6, W, H) & bit) atomicAdd(rowcount + frow + y, 1u);
        }
    }
}

// One wave per image row (+ one extra wave per frame for padding / totals).
// grid: x = ceil((H+1)/4) blocks of 256, y = frames.
__global__ __launch_bounds__(256) void k_emit(const u64 *__restrict__ survmask,
                                              const uint32_t *__restrict__ keyplane,
                                              const uint32_t *__restrict__ rowcount, kp_t *__restrict__ kps,
                                              uint32_t *__restrict__ total_out, int W, int H, int S, int max_kp)
{
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y > H) return;
    const int lane = threadIdx.x & 63;
    const size_t frow = (size_t)blockIdx.y * H;
    const uint32_t *__restrict__ rc = rowcount + frow;
    kp_t *__restrict__ out = kps + (size_t)blockIdx.y * max_kp;

    if (y < H && rc[y] == 0) return;                 // nothing to emit from this row
    // keypoints in rows above (y == H: the frame total)
    uint32_t base = 0;
    for (int r = lane; r < y; r += 64) base += rc[r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) base += __shfl_xor(base, o);

    if (y == H) {
        // padding (:428-436) and the uncapped total (nKeypointsTotal, :405-406)
        if (lane == 0 && total_out) total_out[blockIdx.y] = base;
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
        if (seg < S) m = survmask[(frow + y) * S + seg] & border_mask(y, seg, W, H);
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
                              int32_t *__restrict__ eratio, uint16_t *__restrict__ surv, int W, int H, int S)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const size_t i = (size_t)y * W + x;
    const bool c = (candmask[(size_t)y * S + (x >> 6)] >> (x & 63)) & 1ull;
    const bool s = (survmask[(size_t)y * S + (x >> 6)] >> (x & 63)) & 1ull;
    const uint32_t key = c ? keyplane[i] : 0u;
    if (cand) cand[i] = c ? (uint16_t)(key & 0xffu) : 0;
    if (eratio) eratio[i] = c ? (int32_t)(key >> 8) : 0x7fffffff;
    if (surv) surv[i] = s ? (uint16_t)(key & 0xffu) : 0;
}

}  // namespace sift
