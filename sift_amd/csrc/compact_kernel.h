// compact_kernel.h -- 7x7 cull on the candidate bit mask, border filter and raster-ordered
// compaction into Keypoint_t[max_kp].
//
// Restates culling_hls + IsBestKeypoint (DetectKeypoints.cpp:567-750) and combine_keypoints_hls
// (:372-439).  Pre-cull candidates are ~0.6 % of the frame, so everything after k_detect works
// on bit masks (one word per 64 pixels) and on the keys at the candidates instead of on dense planes.
// The compaction is rank based (per-row counts, their prefix, then an in-row wave scan), which
// reproduces the reference's raster order and its "first max_kp in raster order" cap without a
// sort.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "detect_kernel.h"

namespace sift {


struct kp_t { uint16_t x, y; uint32_t contrast; };   // Keypoint_t, SIFTParameters.hpp:79-85

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

// 7x7 cull by position: one lane per 64-pixel word of the candidate mask, a wave = 64 consecutive words (two
// rows of a 1080p frame).  A candidate survives iff its Q15 edge ratio is strictly smaller than that of every OTHER
// candidate in the 7x7 window (:728-731, :684; neighbours are pre-cull values).
//   * the seven rows of words around the lane's word come through a bounds-checked descriptor of the frame's
//     mask (rows above and below the image read 0), each with the last dword of the word to its left and the
//     first of the word to its right: 21 loads that neighbouring lanes share cache lines of;
//   * per row three dwords P0..P2 = bits -3..28, 29..60, 61..92 of that window, so that the 7 neighbours of bit b in
//     a row are v_alignbit(P1, P0, b) for b < 32 and v_alignbit(P2, P1, b) for b >= 32 (the shift takes b mod 32):
//     two instructions per row and candidate, no memory access;
//   * only candidates that HAVE a neighbour (few) read keys: they are parked in four slots, and after the bit
//     loops their own key and their first neighbour's are requested together -- one round trip to the key
//     plane per word, not one per candidate; further neighbours are looked at only while the candidate is
//     still the best.
// Every word of survmask is stored once (no atomics, nothing to clear beforehand); rowcount (zeroed by k_detect)
// gets one atomic add per word with a survivor inside the border.
// grid: x = ceil(H * S / 256), y = frames; block = 256
__global__ __launch_bounds__(256) void k_cull(const u64 *__restrict__ candmask, const uint32_t *__restrict__ keyplane,
                                              u64 *__restrict__ survmask, uint32_t *__restrict__ rowcount,
                                              int W, int H, int S, int Wl)
{
    const int frame = blockIdx.y;
    const int nwords = H * S;
    const int w = blockIdx.x * 256 + (int)threadIdx.x;
    if (w >= nwords) return;
    const size_t frow = (size_t)frame * H;
    const int y = w / S, seg = w - y * S;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u64 *>(candmask + frow * S), 0, nwords * 8, 0x00020000);
    const uint32_t lmask = seg > 0 ? ~0u : 0u, rmask = seg < S - 1 ? ~0u : 0u;   // no word to the left / right of the row
    uint32_t P0[7], P1[7], P2[7];
    u64 c0 = 0ull;
#pragma unroll
    for (int r = 0; r < 7; r++) {
        // row y + r - 3: an offset before the mask wraps to 4 GB, one past it lies beyond the descriptor: both read 0
        const uint32_t off = (uint32_t)(w + (r - 3) * S) * 8u;
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 m = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
        const uint32_t l = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off - 4u), 0, 0) & lmask;
        const uint32_t rr = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off + 8u), 0, 0) & rmask;
        P0[r] = __builtin_amdgcn_alignbit(m.x, l, 29);
        P1[r] = __builtin_amdgcn_alignbit(m.y, m.x, 29);
        P2[r] = __builtin_amdgcn_alignbit(rr, m.y, 29);
        if (r == 3) c0 = (u64)m.x | (u64)m.y << 32;
    }
    u64 surv = 0ull;
    if (c0) {                                                            // a third of the words hold a candidate
        const uint32_t *__restrict__ kp = keyplane + (size_t)frame * (size_t)W * (size_t)H + (size_t)y * W + seg * 64;   // the word's bit 0
        // a candidate with neighbours: nb = its 7x7 field (row r at bits 7r, bit 7r + c = column b - 3 + c, itself cleared)
        auto resolve = [&](int b, u64 nb) -> bool {
            const uint32_t ec = kp[b] >> 8;
            bool best = true;
            while (nb && best) {                                         // four keys in flight (a candidate rarely has more neighbours)
                uint32_t en[4];
                bool hn[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    hn[u] = nb != 0ull;
                    const int k = hn[u] ? __ffsll((long long)nb) - 1 : 24;   // 24: the keypoint itself
                    nb &= nb - 1;
                    en[u] = kp[(k / 7 - 3) * W + (b + k % 7 - 3)] >> 8;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (hn[u] && !(ec < en[u])) best = false;            // strict (:730)
            }
            return best;
        };
        int qb[4] = {0, 0, 0, 0};
        u64 qn[4] = {0ull, 0ull, 0ull, 0ull};
        int nq = 0;
        // the slots are written through a switch so that they stay in registers
        auto park = [&](int b, u64 nb) {
            switch (nq) {
            case 0: qb[0] = b; qn[0] = nb; break;
            case 1: qb[1] = b; qn[1] = nb; break;
            case 2: qb[2] = b; qn[2] = nb; break;
            default: qb[3] = b; qn[3] = nb; break;
            }
            nq++;
        };
        auto bits = [&](uint32_t c, auto high) {
            constexpr bool HIGH = decltype(high)::value;
            while (c) {
                const int b5 = __ffs((int)c) - 1;
                c &= c - 1;
                uint32_t f[7];
#pragma unroll
                for (int r = 0; r < 7; r++) f[r] = __builtin_amdgcn_alignbit(HIGH ? P2[r] : P1[r], HIGH ? P1[r] : P0[r], (uint32_t)b5) & 0x7fu;
                f[3] &= ~8u;                                             // not the keypoint itself (:728)
                const int b = b5 + (HIGH ? 32 : 0);
                if ((f[0] | f[1] | f[2]) | (f[3] | f[4] | f[5]) | f[6]) {
                    const u64 nb = (u64)(f[0] | f[1] << 7 | f[2] << 14 | f[3] << 21) | (u64)(f[4] | f[5] << 7 | f[6] << 14) << 28;
                    if (nq < 4) {
                        park(b, nb);
                    } else if (resolve(b, nb)) {
                        surv |= 1ull << b;
                    }
                } else {
                    surv |= 1ull << b;                                   // alone in its window
                }
            }
        };
        bits((uint32_t)c0, std::false_type{});
        bits((uint32_t)(c0 >> 32), std::true_type{});
        // parked candidates: own key and first neighbour's key of all of them in one round trip
        uint32_t ec[4], e1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            ec[j] = e1[j] = 0u;
            if (j < nq) {
                const int k = __ffsll((long long)qn[j]) - 1;
                ec[j] = kp[qb[j]] >> 8;
                e1[j] = kp[(k / 7 - 3) * W + (qb[j] + k % 7 - 3)] >> 8;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j < nq) {
                bool best = ec[j] < e1[j];
                const u64 rest = qn[j] & (qn[j] - 1);
                if (best && rest) best = resolve(qb[j], rest);
                if (best) surv |= 1ull << qb[j];
            }
        }
    }
    survmask[frow * S + w] = surv;
    const int n = __popcll(surv & border_mask(y, seg, Wl, H));          // the border rule counts from the logical width
    if (n) atomicAdd(rowcount + frow + y, (uint32_t)n);
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

// Keypoints in raster order: rank = keypoints in the rows above (rowbase: k_rowprefix's exclusive prefix of the
// row counts) + survivors inside the border before the pixel in its row (wave scan over the row's mask words).
// One extra wave per frame pads the array and writes the totals: total_ctx is the context's own copy of the
// uncapped count, which the fused describe kernel reads; total_out the caller's, optional.
// A wave takes R = 64 / S whole rows when a row's mask has at most 32 words (two rows of a 1080p frame, six of a
// 640-pixel one), else one row in passes of 64 words; lane -> (row r = lane / S, word lane - r S).
// grid: x = ceil((ceil(H / R) + 1) / 4) blocks of 256, y = frames.
__global__ __launch_bounds__(256) void k_emit(const u64 *__restrict__ survmask,
                                              const uint32_t *__restrict__ keyplane,
                                              const uint32_t *__restrict__ rowbase, kp_t *__restrict__ kps,
                                              uint32_t *__restrict__ total_ctx, uint32_t *__restrict__ total_out,
                                              int W, int H, int S, int max_kp, int Wl, int R)
{
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), nw = (H + R - 1) / R;
    const int lane = threadIdx.x & 63;
    if (wv > nw) return;
    const size_t frow = (size_t)blockIdx.y * H;
    const uint32_t *__restrict__ rb = rowbase + (size_t)blockIdx.y * (H + 1);
    kp_t *__restrict__ out = kps + (size_t)blockIdx.y * max_kp;

    if (wv == nw) {
        // one more wave per frame: padding (:428-436) and the uncapped total (nKeypointsTotal, :405-406)
        const uint32_t total = rb[H];
        if (lane == 0) {
            total_ctx[blockIdx.y] = total;
            if (total_out) total_out[blockIdx.y] = total;
        }
        const uint32_t first = total < (uint32_t)max_kp ? total : (uint32_t)max_kp;
        u64 *o64 = reinterpret_cast<u64 *>(out);
        for (uint32_t i = first + lane; i < (uint32_t)max_kp; i += 64) o64[i] = 0ull;
        return;
    }
    const uint32_t *__restrict__ kpf = keyplane + (size_t)blockIdx.y * (size_t)W * (size_t)H;
    auto emit = [&](u64 m, uint32_t idx, int y, int seg) {
        const uint32_t *__restrict__ kp = kpf + (size_t)y * W + seg * 64;
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (idx < (uint32_t)max_kp) {                                // the first max_kp in raster order (:409)
                kp_t k;
                k.x = (uint16_t)(seg * 64 + b);
                k.y = (uint16_t)y;
                k.contrast = kp[b] & 0xffu;
                out[idx] = k;
            }
            idx++;
        }
    };
    // inclusive wave scan
    auto scan = [&](uint32_t n) {
        uint32_t inc = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        return inc;
    };

    if (R > 1) {                                                         // wave-uniform: S <= 32
        int r = 0;
        for (int k = 1; k < R; k++) r += lane >= k * S;
        const int seg = lane - r * S, y = wv * R + r;
        const bool live = lane < R * S && y < H;
        // keypoints in the rows above (k_rowprefix) and the row's mask word, requested together
        const uint32_t base = rb[live ? y : 0];
        u64 m = live ? survmask[(frow + y) * S + seg] : 0ull;
        m &= border_mask(y, seg, Wl, H);
        const uint32_t n = (uint32_t)__popcll(m);
        const uint32_t inc = scan(n);
        const uint32_t before = __shfl(inc, max(r * S - 1, 0));          // inclusive count at the end of the row before, in this wave
        emit(m, base + inc - n - (r ? before : 0u), y, seg);
        return;
    }
    const int y = wv;
    uint32_t base = rb[y];
    if (rb[y + 1] == base || base >= (uint32_t)max_kp) return;           // nothing to emit from this row / cap reached above it
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int seg = s0 + lane;
        u64 m = 0ull;
        if (seg < S) m = survmask[(frow + y) * S + seg] & border_mask(y, seg, Wl, H);
        const uint32_t n = (uint32_t)__popcll(m);
        const uint32_t inc = scan(n);
        emit(m, base + inc - n, y, seg);
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
