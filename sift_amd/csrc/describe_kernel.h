// describe_kernel.h -- main orientation + 9x8 circular-region descriptor,
// one wave per keypoint.
//
// Restates compute_main_orientation + ComputeMainOrientation
// (SIFT_Descriptor.cpp:18-103, :107-130), compute_histograms_hls (:151-278)
// and reorder_histograms_hls (:295-398).
//
// MaskAll (mask_all.hpp:8-78) is the union of the nine 25x25 circles
// (circle_mask.hpp:16-40) placed at (XStart, YStart); its column 70 is empty,
// so the reference's 71x70 orientation window (:70) sees the whole union.
// Hence, with Hin[n][a] the nine circle histograms and Ov[a] the histogram of
// the pixels that lie in two circles (no pixel lies in three):
//        orientation histogram  Hist[a] = sum_n Hin[n][a] - Ov[a]      (exact integers)
// and each of the 3809 union pixels of the 71x71 window is loaded once.
//
// A host-built table lists the union pixels (those in two circles first):
//   entry = dx | dy<<8 | c1<<16 | c2<<20          (c = 15: none)
// Lane l accumulates into its own LDS column hist[row][l] (bank = lane, no
// conflicts, ds_add_u32 without return); rows are circle*8 + (ang-1), then 8
// rows for Ov.  Codes outside 1..8 are dropped (code 0 is accumulated and then
// discarded by the reference, :245-252).  The 80 row sums are then read with a
// per-lane skew so that the 64 lanes hit 64 different banks.
//
// The gradient plane is addressed linearly like the reference does
// (pGrad[y*WIDTH + x - 35 + col], :64/:204); indices outside the plane read
// {0,0} (the reference would read out of bounds).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "compact_kernel.h"

namespace sift {

enum { DESC_FUSED = 0, DESC_ORI_ONLY = 1, DESC_GIVEN_ORI = 2 };
enum { GRAD_PACKED16 = 0, GRAD_FULL32 = 1 };

constexpr int DESC_ROWS = 80;   // 72 circle bins + 8 overlap bins

// grid: x = max_kp keypoints, y = frames; block = 64 (one wave)
template <int GFMT, int MODE>
__global__ __launch_bounds__(64) void k_describe(const void *__restrict__ gradv, const kp_t *__restrict__ kps,
                                                 const uint8_t *__restrict__ ori_in, uint8_t *__restrict__ ori_out,
                                                 uint32_t *__restrict__ desc, const uint32_t *__restrict__ tab,
                                                 int ntab, int W, int H, int max_kp)
{
    __shared__ uint32_t hist[DESC_ROWS][64];
    __shared__ uint32_t fin[DESC_ROWS];
    __shared__ uint32_t s_m;

    const int lane = threadIdx.x;
    const size_t slot = (size_t)blockIdx.y * max_kp + blockIdx.x;
    const kp_t kp = kps[slot];
    const int px = kp.x, py = kp.y;
    uint32_t *__restrict__ out = desc ? desc + slot * 72 : nullptr;

    if (!(px > 0 && py > 0)) {                       // :52 / :196 -- padding slot
        if (MODE != DESC_ORI_ONLY) {
            out[lane] = 0;
            if (lane < 8) out[64 + lane] = 0;
        }
        if (MODE != DESC_GIVEN_ORI && ori_out && lane == 0) ori_out[slot] = 0;
        return;
    }

#pragma unroll
    for (int r = 0; r < DESC_ROWS; r++) hist[r][lane] = 0;
    // same-lane LDS accesses are ordered; no barrier needed before the adds

    const long npix = (long)W * H;
    const size_t fpix = (size_t)blockIdx.y * (size_t)npix;
    const long origin = (long)(py - 35) * W + (px - 35);
    for (int t = lane; t < ntab; t += 64) {
        const uint32_t e = tab[t];
        const int dx = e & 0xff, dy = (e >> 8) & 0xff;
        const int c1 = (e >> 16) & 15, c2 = (e >> 20) & 15;
        const long idx = origin + (long)dy * W + dx;
        int ang = 0;
        uint32_t mag = 0;
        if (idx >= 0 && idx < npix) {
            if (GFMT == GRAD_PACKED16) {
                const uint32_t g = reinterpret_cast<const uint16_t *>(gradv)[fpix + idx];
                ang = g >> 12;
                mag = g & 0xfffu;
            } else {
                const uint32_t g = reinterpret_cast<const uint32_t *>(gradv)[fpix + idx];
                ang = g & 0xffffu;
                mag = g >> 16;
            }
        }
        if (ang >= 1 && ang <= 8 && c1 != 15) {
            atomicAdd(&hist[c1 * 8 + ang - 1][lane], mag);
            if (c2 != 15) {
                atomicAdd(&hist[c2 * 8 + ang - 1][lane], mag);
                atomicAdd(&hist[72 + ang - 1][lane], mag);
            }
        }
    }
    __syncthreads();

    // row sums: lane l takes rows l and l+64, reading column (c+l)&63 at step c
    {
        uint32_t s0 = 0, s1 = 0;
        const bool second = lane + 64 < DESC_ROWS;
#pragma unroll 8
        for (int c = 0; c < 64; c++) {
            const int col = (c + lane) & 63;
            s0 += hist[lane][col];
            if (second) s1 += hist[lane + 64][col];
        }
        fin[lane] = s0;
        if (second) fin[lane + 64] = s1;
    }
    __syncthreads();

    // main orientation (:110-126): first bin 1..8 with the strictly largest value
    if (lane == 0) {
        uint32_t m = 0;
        if (MODE == DESC_GIVEN_ORI) {
            m = ori_in[slot];
        } else {
            int best = 0;
            for (int a = 0; a < 8; a++) {
                int v = -(int)fin[72 + a];
                for (int n = 0; n < 9; n++) v += (int)fin[n * 8 + a];
                if (v > best) { best = v; m = (uint32_t)(a + 1); }
            }
            if (ori_out) ori_out[slot] = (uint8_t)m;
        }
        s_m = m;
    }
    __syncthreads();
    if (MODE == DESC_ORI_ONLY) return;

    // rotation (:346-393): out[c][j] = Hin[c<8 ? (c+m)&7 : 8][(j+m)&7]
    const uint32_t m = s_m;
    {
        const int c = lane >> 3, j = lane & 7;       // circles 0..7
        out[lane] = fin[((c + m) & 7) * 8 + ((j + m) & 7)];
        if (lane < 8) out[64 + lane] = fin[64 + ((lane + m) & 7)];
    }
}

}  // namespace sift
