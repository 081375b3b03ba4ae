// blur_kernel.h -- Q15 Gaussian blur x3 (5x5, 11x11, 19x19) for gfx950.
//
// Restates GaussianBlurHLS + FilterKernelOperator1/2/3 + RoundToInt
// (DetectKeypoints.cpp:925-1043, :832-912, :810-827) as one streaming VALU
// kernel.  The quantised 2-D tables are not separable (ranks 3/6/10), so the
// exact 2-D integer sum is computed, with v_dot4_u32_u8 on byte-packed pixels:
//
//   * a thread owns a strip 4 pixels wide and `rows` tall and walks DOWN it;
//   * per input row it loads 7 aligned dwords (28 pixels around its 4) and
//     builds, with v_alignbyte_b32, the five window dwords W_j (pixels
//     x-9+4j .. x-9+4j+3) of each of its 4 output pixels -- one set of windows
//     feeds all three kernels (11x11 uses W_1..W_3, 5x5 uses W_1..W_2);
//   * every kernel keeps one accumulator per (output row in flight, pixel):
//     19 + 11 + 5 "slots".  Input row i updates slot s with table row
//     (n-1-s) and the result moves to slot s-1 in the same instruction
//     (dot4's destination is a different register from its addend), so the
//     vertical slide costs no moves; slot 0 is a finished output row;
//   * table values >255 are split K = lo + 256*hi; the hi plane is sparse
//     (21 / 45 / 13 taps) and is folded with one v_lshl_add per table row;
//   * the accumulator starts at 16384, so G = acc >> 15 is RoundToInt's
//     floor(S + 0.5); the clamp to 255 (:848-851) can never fire because
//     255 * sum(Kq) + 16384 < 256 * 32768 for all three tables (static_assert).
//
// Output: one interleaved word per pixel  src | G1<<8 | G2<<16 | G3<<24
// ("pyr" plane) -- 16-byte stores per lane, and the detect kernel gets every
// plane of a neighbour with a single load.  G1/G2 rows finish 7/4 input rows
// before the G3 row of the same image row; they wait in packed delay lines.
//
// No LDS, no barriers: neighbouring threads' loads overlap and are served by
// the vector L1.  Out-of-image taps read 0 (DetectKeypoints.cpp:968-991; the
// top rows, uninitialised in the reference, are defined as 0).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "sift_tables.h"

namespace sift {

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// full-table lookup from the unique octant (see sift_tables.h)
constexpr int oct_at(const unsigned short *oct, int n, int r, int c)
{
    if (r < 0 || r >= n || c < 0 || c >= n) return 0;
    int o = n / 2;
    int rr = r <= o ? r : n - 1 - r, cc = c <= o ? c : n - 1 - c;
    int a = rr < cc ? rr : cc, b = rr < cc ? cc : rr;
    return oct[a * (o + 1) - a * (a - 1) / 2 + (b - a)];
}

template <int WHICH> constexpr int ksize() { return WHICH == 1 ? 5 : WHICH == 2 ? 11 : 19; }

// tap of table WHICH at table row r and WINDOW column wc (0..19; the window
// is the 19x19 one, the smaller tables sit centred in it: offsets 7 and 4)
template <int WHICH>
constexpr int ktap(int r, int wc)
{
    return WHICH == 3 ? oct_at(SIFT_G3_OCT, 19, r, wc)
         : WHICH == 2 ? oct_at(SIFT_G2_OCT, 11, r, wc - 4)
                      : oct_at(SIFT_G1_OCT, 5, r, wc - 7);
}

// coefficient dword for window dword j: byte b multiplies pixel x-9+4j+b
template <int WHICH>
constexpr uint32_t kdw(int r, int j, int plane)
{
    uint32_t v = 0;
    for (int b = 0; b < 4; b++) {
        uint32_t t = (uint32_t)ktap<WHICH>(r, 4 * j + b);
        v |= (plane == 0 ? (t & 255u) : (t >> 8)) << (8 * b);
    }
    return v;
}

template <int WHICH>
constexpr long ksum()
{
    long s = 0;
    for (int r = 0; r < ksize<WHICH>(); r++)
        for (int c = 0; c < 20; c++) s += ktap<WHICH>(r, c);
    return s;
}
static_assert(ksum<1>() == 32765 && ksum<2>() == 32759 && ksum<3>() == 32744, "table sums (SURVEY.md A.1)");
static_assert(255 * ksum<1>() + 16384 < 256 * 32768L, "clamp cannot fire");
static_assert(kdw<3>(9, 2, 1) != 0 && kdw<3>(0, 0, 1) == 0, "hi plane only near the centre");

// acc + sum_j W_j . K[R][j]   (lo plane into acc, hi plane into a temp, folded <<8)
template <int WHICH, int R>
__device__ __forceinline__ uint32_t row_mac(const uint32_t (&w)[5], uint32_t acc)
{
    static_for<0, 5>([&](auto jj) {
        constexpr int j = jj;
        constexpr uint32_t k = kdw<WHICH>(R, j, 0);
        if constexpr (k != 0) acc = __builtin_amdgcn_udot4(w[j], k, acc, false);
    });
    constexpr bool any_hi = (kdw<WHICH>(R, 0, 1) | kdw<WHICH>(R, 1, 1) | kdw<WHICH>(R, 2, 1) |
                             kdw<WHICH>(R, 3, 1) | kdw<WHICH>(R, 4, 1)) != 0;
    if constexpr (any_hi) {
        uint32_t h = 0;
        static_for<0, 5>([&](auto jj) {
            constexpr int j = jj;
            constexpr uint32_t k = kdw<WHICH>(R, j, 1);
            if constexpr (k != 0) h = __builtin_amdgcn_udot4(w[j], k, h, false);
        });
        acc = (h << 8) + acc;
    }
    return acc;
}

constexpr uint32_t BLUR_INIT = 16384u;   // the +0.5 of RoundToInt in Q15

// One input row into the slots [LO,HI) of table WHICH (N = table size).
template <int WHICH, int LO, int HI>
__device__ __forceinline__ void slide(uint32_t (&a)[ksize<WHICH>()][4], const uint32_t (&w)[4][5])
{
    constexpr int N = ksize<WHICH>();
    static_for<LO, HI>([&](auto ss) {
        constexpr int s = ss;
#pragma unroll
        for (int p = 0; p < 4; p++)
            a[s][p] = row_mac<WHICH, N - 1 - s>(w[p], s == N - 1 ? BLUR_INIT : a[s + 1 < N ? s + 1 : s][p]);
    });
}

__device__ __forceinline__ uint32_t pack4_q15(const uint32_t (&a)[4])
{
    // (a>>15) of four finished accumulators into one dword, pixel p in byte p
    uint32_t lo = (a[0] >> 15) | ((a[1] >> 15) << 8);
    uint32_t hi = (a[2] >> 15) | ((a[3] >> 15) << 8);
    return lo | (hi << 16);
}

// grid: x = ceil(W/4 / 256) blocks of 256 threads, y = strips, z = frames
template <bool GUARD>
__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ imgs, uint32_t *__restrict__ pyr,
                                              int W, int H, int rows)
{
    const int W4 = W >> 2;
    const int xq = blockIdx.x * blockDim.x + threadIdx.x;
    if (xq >= W4) return;
    const int ys = blockIdx.y * rows;
    const int ye = min(H, ys + rows);
    const size_t fofs = (size_t)blockIdx.z * (size_t)W * (size_t)H;
    const uint32_t *__restrict__ img = reinterpret_cast<const uint32_t *>(imgs + fofs);
    uint32_t *__restrict__ out = pyr + fofs;

    // column part of the 7 raw-dword addresses and their validity
    int qc[7];
    bool cv[7];
#pragma unroll
    for (int k = 0; k < 7; k++) {
        int q = xq - 3 + k;
        cv[k] = (q >= 0) && (q < W4);
        qc[k] = cv[k] ? q : xq;
    }
    auto load_row = [&](int i, uint32_t (&A)[7]) {
        const bool rv = (i >= 0) && (i < H);
        const uint32_t *rp = img + (size_t)(rv ? i : 0) * W4;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            uint32_t v = rp[qc[k]];
            A[k] = (rv && cv[k]) ? v : 0u;
        }
    };

    uint32_t a3[19][4], a2[11][4], a1[5][4];
#pragma unroll
    for (int s = 0; s < 19; s++)
#pragma unroll
        for (int p = 0; p < 4; p++) a3[s][p] = 0;
#pragma unroll
    for (int s = 0; s < 11; s++)
#pragma unroll
        for (int p = 0; p < 4; p++) a2[s][p] = 0;
#pragma unroll
    for (int s = 0; s < 5; s++)
#pragma unroll
        for (int p = 0; p < 4; p++) a1[s][p] = 0;
    uint32_t d1[7] = {0, 0, 0, 0, 0, 0, 0}, d2[4] = {0, 0, 0, 0};

    uint32_t cur[7], nxt[7];
    load_row(ys - 9, cur);
    for (int i = ys - 9; i < ye + 9; ++i) {
        load_row(i + 1, nxt);                 // prefetch; a row past the strip is harmless (clamped / zero)

        // window dwords: pixel p's window starts at byte 3+p of cur[]
        uint32_t w[4][5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            w[0][j] = __builtin_amdgcn_alignbyte(cur[j + 1], cur[j], 3);
            w[1][j] = cur[j + 1];
            w[2][j] = __builtin_amdgcn_alignbyte(cur[j + 2], cur[j + 1], 1);
            w[3][j] = __builtin_amdgcn_alignbyte(cur[j + 2], cur[j + 1], 2);
        }

        if constexpr (GUARD) {
            // slot s of table n holds output row  i - off + s  (off = 9/5/2);
            // a group of slots is skipped when none of its rows is in [ys,ye)
            auto live = [&](int off, int lo, int hi) {   // slots [lo,hi)
                return (i - off + hi - 1 >= ys) && (i - off + lo < ye);
            };
            if (live(9, 0, 5)) slide<3, 0, 5>(a3, w);
            if (live(9, 5, 10)) slide<3, 5, 10>(a3, w);
            if (live(9, 10, 15)) slide<3, 10, 15>(a3, w);
            if (live(9, 15, 19)) slide<3, 15, 19>(a3, w);
            if (live(5, 0, 6)) slide<2, 0, 6>(a2, w);
            if (live(5, 6, 11)) slide<2, 6, 11>(a2, w);
            if (live(2, 0, 5)) slide<1, 0, 5>(a1, w);
        } else {
            slide<3, 0, 19>(a3, w);
            slide<2, 0, 11>(a2, w);
            slide<1, 0, 5>(a1, w);
        }

        // finished rows: G1 row i-2, G2 row i-5, G3 row i-9
        const int y3 = i - 9;
        if (y3 >= ys) {                       // y3 < ye holds by the loop bound
            const uint32_t g1 = d1[6], g2 = d2[3];
            const uint32_t sp = img[(size_t)y3 * W4 + xq];
            uint4 o;
            o.x = (sp & 0xffu) | ((g1 & 0xffu) << 8) | ((g2 & 0xffu) << 16) | ((a3[0][0] >> 15) << 24);
            o.y = ((sp >> 8) & 0xffu) | (((g1 >> 8) & 0xffu) << 8) | (((g2 >> 8) & 0xffu) << 16) | ((a3[0][1] >> 15) << 24);
            o.z = ((sp >> 16) & 0xffu) | (((g1 >> 16) & 0xffu) << 8) | (((g2 >> 16) & 0xffu) << 16) | ((a3[0][2] >> 15) << 24);
            o.w = (sp >> 24) | ((g1 >> 24) << 8) | ((g2 >> 24) << 16) | ((a3[0][3] >> 15) << 24);
            *reinterpret_cast<uint4 *>(out + (size_t)y3 * W + 4 * (size_t)xq) = o;
        }
#pragma unroll
        for (int k = 6; k > 0; k--) d1[k] = d1[k - 1];
        d1[0] = pack4_q15(a1[0]);
#pragma unroll
        for (int k = 3; k > 0; k--) d2[k] = d2[k - 1];
        d2[0] = pack4_q15(a2[0]);

#pragma unroll
        for (int k = 0; k < 7; k++) cur[k] = nxt[k];
    }
}

}  // namespace sift
