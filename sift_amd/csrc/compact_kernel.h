// compact_kernel.h -- 7x7 cull on the sparse candidate set, border filter and
// raster-ordered compaction into Keypoint_t[max_kp].
//
// Restates culling_hls + IsBestKeypoint (DetectKeypoints.cpp:567-750) and
// combine_keypoints_hls (:372-439).  Candidates are ~0.5 % of the pixels, so
// the cull works on the ballot masks of k_detect instead of on dense planes:
// a lane owns one 64-pixel segment, finds the candidates in the 7x7 window of
// each of its candidates by shifting the neighbouring masks, and only then
// reads their keys.  The compaction is rank based (per-row counts, then an
// in-row wave scan), which reproduces the reference's raster order and its
// "first max_kp in raster order" cap without a sort.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sift {

typedef unsigned long long u64;

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

// One wave per image row, lane = segment.  grid: x = ceil(H/4) blocks of 256, y = frames.
__global__ __launch_bounds__(256) void k_cull(const u64 *__restrict__ candmask,
                                              const uint32_t *__restrict__ keyplane,
                                              u64 *__restrict__ survmask, uint32_t *__restrict__ rowcount,
                                              int W, int H, int S)
{
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= H) return;
    const int lane = threadIdx.x & 63;
    const size_t frow = (size_t)blockIdx.y * H;
    const u64 *__restrict__ cm = candmask + frow * S;
    const uint32_t *__restrict__ kp = keyplane + (size_t)blockIdx.y * (size_t)W * (size_t)H;
    uint32_t cnt = 0;
    for (int seg = lane; seg < S; seg += 64) {
        const u64 m = cm[(size_t)y * S + seg];
        u64 keep = m;
        u64 it = m;
        while (it) {
            const int b = __ffsll((long long)it) - 1;
            it &= it - 1;
            const int x = seg * 64 + b;
            const uint32_t ec = kp[(size_t)y * W + x] >> 8;
            bool best = true;
            for (int dy = -3; dy <= 3 && best; dy++) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                uint32_t f = mask_field7(cm + (size_t)yy * S, S, x - 3);
                if (dy == 0) f &= ~8u;               // not the keypoint itself (:728)
                while (f) {
                    const int k = __ffs((int)f) - 1;
                    f &= f - 1;
                    const uint32_t en = kp[(size_t)yy * W + (x - 3 + k)] >> 8;
                    if (!(ec < en)) { best = false; break; }   // strict (:730)
                }
            }
            if (!best) keep &= ~(1ull << b);
        }
        survmask[(frow + y) * S + seg] = keep;
        cnt += (uint32_t)__popcll(keep & border_mask(y, seg, W, H));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) rowcount[frow + y] = cnt;
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
