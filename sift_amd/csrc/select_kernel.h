// select_kernel.h -- the keypoint post-processing of main.cpp:82-130 on the device.
//
// The reference sorts all MAX_KEYPOINTS slots by contrast, descending (std::sort with
// CompareContrast, main.cpp:8-11, :82), keeps the first MAX_FINAL_KEYPOINTS (:84) and counts
// those with x > 0 && y > 0 (:112-129).  Only the first nfinal of the sorted order are ever
// used, so this kernel selects them instead of sorting 10,000 slots:
//   1. radix select (4 passes of 8 bits, LDS histogram): T = the nfinal-th largest contrast,
//      and how many slots with contrast == T belong to the selection;
//   2. ordered compaction (block scan, slot order) of the slots with contrast > T and of the
//      first slots with contrast == T into LDS;
//   3. rank of every selected slot by counting (contrast descending, then slot ascending).
// std::sort is not stable: among equal contrasts the reference's order is whatever the
// library's introsort leaves.  Here ties are ordered by slot -- the raster order sift_detect
// emits -- and the tests' CPU restatement does the same.
//
// Padding slots (0, 0, 0) take part like in the reference: with fewer than nfinal valid
// keypoints the tail of the output is zeros.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "compact_kernel.h"

namespace sift {

constexpr int SEL_THREADS = 1024;
constexpr int SEL_MAX_FINAL = 4096;      // selected slots live in LDS (3 x 16 KB)

// grid: x = frames; block = 1024.
// kps [frames][max_kp] -> final [frames][nfinal]; optional: index [frames][nfinal] (slot of
// each selected keypoint), nvalid [frames], desc_in [frames][max_kp][72] -> desc_out
// [frames][nfinal][72] (rows of the selected keypoints: the reference leaves the descriptor
// matrix in slot order, main.cpp:134-148).
__global__ __launch_bounds__(SEL_THREADS) void k_select_top(const kp_t *__restrict__ kps, int max_kp, int nfinal,
                                                            kp_t *__restrict__ final, uint32_t *__restrict__ index,
                                                            uint32_t *__restrict__ nvalid,
                                                            const uint32_t *__restrict__ desc_in,
                                                            uint32_t *__restrict__ desc_out)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t s_digit, s_above, s_valid;
    __shared__ uint32_t sel_key[SEL_MAX_FINAL], sel_slot[SEL_MAX_FINAL], fin_slot[SEL_MAX_FINAL];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const kp_t *__restrict__ fk = kps + (size_t)blockIdx.x * max_kp;

    // 1. radix select: after the pass over byte b, `prefix` holds the top (4-b) bytes of T and
    //    `remaining` the rank of T among the slots that share them
    uint32_t prefix = 0, remaining = (uint32_t)nfinal;
    for (int b = 3; b >= 0; --b) {
        if (t < 256) hist[t] = 0;
        if (t == 0) s_valid = 0;
        __syncthreads();
        for (int i = t; i < max_kp; i += SEL_THREADS) {
            const uint32_t k = fk[i].contrast;
            if (b == 3 || (k >> (8 * (b + 1))) == prefix) atomicAdd(&hist[(k >> (8 * b)) & 255u], 1u);
        }
        __syncthreads();
        // thread d' = 255 - d scans from the largest digit down: inc = slots with digit >= d
        uint32_t c = 0, inc = 0;
        if (t < 256) {
            c = hist[255 - t];
            inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
                if (lane >= o) inc += v;
            }
            if (lane == 63) wsum[wave] = inc;
        }
        __syncthreads();
        if (t < 256) {
            for (int w = 0; w < wave; w++) inc += wsum[w];
            const uint32_t above = inc - c;                  // slots with a larger digit
            if (above < remaining && remaining <= inc) { s_digit = 255u - (uint32_t)t; s_above = above; }
        }
        __syncthreads();
        prefix = (prefix << 8) | s_digit;
        remaining -= s_above;
        __syncthreads();
    }
    const uint32_t T = prefix, need_eq = remaining, count_gt = (uint32_t)nfinal - need_eq;

    // 2. ordered compaction: contrast > T in slot order, then the first need_eq slots with contrast == T
    uint32_t run_gt = 0, run_eq = 0;
    for (int i0 = 0; i0 < max_kp; i0 += SEL_THREADS) {
        const int i = i0 + t;
        const uint32_t k = i < max_kp ? fk[i].contrast : 0u;
        const bool gt = i < max_kp && k > T, eq = i < max_kp && k == T;
        const uint32_t v = (gt ? 1u : 0u) | (eq ? 0x10000u : 0u);   // two counters of <= 1024 each
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t base = 0, all = 0;
        for (int w = 0; w < 16; w++) {
            const uint32_t s = wsum[w];
            if (w < wave) base += s;
            all += s;
        }
        const uint32_t excl = base + inc - v;
        if (gt) {
            const uint32_t pos = run_gt + (excl & 0xffffu);
            sel_key[pos] = k;
            sel_slot[pos] = (uint32_t)i;
        }
        if (eq) {
            const uint32_t r = run_eq + (excl >> 16);
            if (r < need_eq) {
                sel_key[count_gt + r] = k;
                sel_slot[count_gt + r] = (uint32_t)i;
            }
        }
        run_gt += all & 0xffffu;
        run_eq += all >> 16;
        __syncthreads();
    }

    // 3. ranks.  The == T group already stands in its final place (all of it ranks below every
    //    larger contrast, in slot order); a slot of the > T group counts the larger contrasts and
    //    the equal ones before it (the group is in slot order).
    uint32_t valid = 0;
    for (int e = t; e < nfinal; e += SEL_THREADS) {
        uint32_t rank = (uint32_t)e;
        const uint32_t ke = sel_key[e], se = sel_slot[e];
        if ((uint32_t)e < count_gt) {
            rank = 0;
            for (uint32_t j = 0; j < count_gt; j++) {
                const uint32_t kj = sel_key[j];
                rank += (kj > ke || (kj == ke && j < (uint32_t)e)) ? 1u : 0u;
            }
        }
        const kp_t k = fk[se];
        final[(size_t)blockIdx.x * nfinal + rank] = k;
        if (index) index[(size_t)blockIdx.x * nfinal + rank] = se;
        fin_slot[rank] = se;
        valid += (k.x > 0 && k.y > 0) ? 1u : 0u;            // main.cpp:125
    }
    if (nvalid) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) valid += (uint32_t)__shfl_xor((int)valid, o);
        if (lane == 0 && valid) atomicAdd(&s_valid, valid);
    }
    __syncthreads();
    if (nvalid && t == 0) nvalid[blockIdx.x] = s_valid;

    // rows of the selected keypoints
    if (desc_in && desc_out) {
        const uint32_t *__restrict__ din = desc_in + (size_t)blockIdx.x * max_kp * 72;
        uint32_t *__restrict__ dout = desc_out + (size_t)blockIdx.x * nfinal * 72;
        for (int w = t; w < nfinal * 72; w += SEL_THREADS) {
            const int r = w / 72, c = w - r * 72;
            dout[w] = din[(size_t)fin_slot[r] * 72 + c];
        }
    }
}

}  // namespace sift
