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
// Output ("pyr" plane): one uint4 per group of 4 pixels = { src, G1, G2, G3 },
// each a dword with pixel p of the group in byte p -- a 16-byte store per lane
// with no byte shuffling, and the detect kernel (4 pixels per lane as well)
// gets all four planes of its pixels with one load.  G1/G2 rows finish 7/4
// input rows before the G3 row of the same image row; they wait in packed
// delay lines.
//
// No LDS, no barriers: neighbouring threads' loads overlap and are served by
// the vector L1; the next input row is loaded one iteration ahead.  Out-of-image
// taps read 0 (DetectKeypoints.cpp:968-991; the top rows, uninitialised in the
// reference, are defined as 0): every row is read through a buffer descriptor of
// exactly its W bytes (none for rows outside the image), so those taps are the
// zeros of the hardware's bounds check.
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

constexpr uint32_t BLUR_INIT = 16384u;   // the +0.5 of RoundToInt in Q15

template <int WHICH, int R>
constexpr bool row_has_hi()
{
    return (kdw<WHICH>(R, 0, 1) | kdw<WHICH>(R, 1, 1) | kdw<WHICH>(R, 2, 1) | kdw<WHICH>(R, 3, 1) |
            kdw<WHICH>(R, 4, 1)) != 0;
}

// One input row into the slots [LO,HI) of table WHICH (N = table size): slot s takes table row
// N-1-s and the old slot s+1 as its addend (ascending s, so the old value is still there).
// The four pixels of the thread are interleaved instruction by instruction: a v_dot4 that feeds
// the next one's accumulator back to back costs a wait state (the compiler pads with s_nop).
template <int WHICH, int LO, int HI>
__device__ __forceinline__ void slide(uint32_t (&a)[ksize<WHICH>()][4], const uint32_t (&w)[4][5])
{
    constexpr int N = ksize<WHICH>();
    static_for<LO, HI>([&](auto ss) {
        constexpr int s = ss;
        constexpr int R = N - 1 - s;
        uint32_t acc[4];
#pragma unroll
        for (int p = 0; p < 4; p++) acc[p] = (s == N - 1) ? BLUR_INIT : a[s + 1 < N ? s + 1 : s][p];
        static_for<0, 5>([&](auto jj) {
            constexpr int j = jj;
            constexpr uint32_t k = kdw<WHICH>(R, j, 0);
            if constexpr (k != 0) {
#pragma unroll
                for (int p = 0; p < 4; p++) acc[p] = __builtin_amdgcn_udot4(w[p][j], k, acc[p], false);
            }
        });
        if constexpr (row_has_hi<WHICH, R>()) {
            uint32_t h[4] = {0, 0, 0, 0};
            static_for<0, 5>([&](auto jj) {
                constexpr int j = jj;
                constexpr uint32_t k = kdw<WHICH>(R, j, 1);
                if constexpr (k != 0) {
#pragma unroll
                    for (int p = 0; p < 4; p++) h[p] = __builtin_amdgcn_udot4(w[p][j], k, h[p], false);
                }
            });
#pragma unroll
            for (int p = 0; p < 4; p++) acc[p] = (h[p] << 8) + acc[p];
        }
#pragma unroll
        for (int p = 0; p < 4; p++) a[s][p] = acc[p];
    });
}

// The tables are symmetric top to bottom (row R = row N-1-R), so one input row contributes the
// same value to slot s and to slot N-1-s: it is computed once (dot4 chains from 0, hi plane
// folded) and added to both with plain adds, which issue at twice the rate of a dot4.  The slots
// are still updated in ascending order (slot s takes the old slot s+1), so the contributions of
// the lower half stay in registers until the upper half has used them.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// two dwords as one 64-bit value: a register pair, no instruction (as a shift-and-or it costs moves)
__device__ __forceinline__ uint64_t pair64(uint32_t lo, uint32_t hi)
{
    u32x2 v = {lo, hi};
    return __builtin_bit_cast(uint64_t, v);
}
__device__ __forceinline__ uint32_t lo32(uint64_t v) { return __builtin_bit_cast(u32x2, v).x; }
__device__ __forceinline__ uint32_t hi32(uint64_t v) { return __builtin_bit_cast(u32x2, v).y; }

template <int WHICH, int R>
__device__ __forceinline__ void row_contrib(uint32_t (&c)[4], const uint32_t (&w)[4][5], const uint32_t (&init)[4])
{
#pragma unroll
    for (int p = 0; p < 4; p++) c[p] = init[p];
    static_for<0, 5>([&](auto jj) {
        constexpr int j = jj;
        constexpr uint32_t k = kdw<WHICH>(R, j, 0);
        if constexpr (k != 0) {
#pragma unroll
            for (int p = 0; p < 4; p++) c[p] = __builtin_amdgcn_udot4(w[p][j], k, c[p], false);
        }
    });
    if constexpr (row_has_hi<WHICH, R>()) {
        uint32_t h[4] = {0, 0, 0, 0};
        static_for<0, 5>([&](auto jj) {
            constexpr int j = jj;
            constexpr uint32_t k = kdw<WHICH>(R, j, 1);
            if constexpr (k != 0) {
#pragma unroll
                for (int p = 0; p < 4; p++) h[p] = __builtin_amdgcn_udot4(w[p][j], k, h[p], false);
            }
        });
        // (h << 8) + c for two pixels at once: h < 2^16, so the shift does not cross the dwords
        const uint64_t r0 = (pair64(h[0], h[1]) << 8) + pair64(c[0], c[1]);
        const uint64_t r1 = (pair64(h[2], h[3]) << 8) + pair64(c[2], c[3]);
        c[0] = lo32(r0); c[1] = hi32(r0);
        c[2] = lo32(r1); c[3] = hi32(r1);
    }
}

template <int WHICH>
constexpr bool table_is_symmetric()
{
    constexpr int N = ksize<WHICH>();
    for (int r = 0; r < N; r++)
        for (int j = 0; j < 5; j++)
            for (int pl = 0; pl < 2; pl++)
                if (kdw<WHICH>(r, j, pl) != kdw<WHICH>(N - 1 - r, j, pl)) return false;
    return true;
}

// Two accumulators per 64-bit add: an accumulator stays below 2^24 (255 * 32765 + 16384), so adding
// the pixel pairs (0,1) and (2,3) as 64-bit values never carries from the low dword into the high
// one, and v_lshl_add_u64 issues at the rate of one 32-bit add (tools/ubench/valu_rate.hip).
// Two accumulators per 64-bit add: an accumulator stays below 2^24 (255 * 32765 + 16384), so adding
// the pixel pairs (0,1) and (2,3) as 64-bit values never carries from the low dword into the high
// one, and v_lshl_add_u64 issues at the rate of one 32-bit add (tools/ubench/valu_rate.hip).  The
// state of this variant is therefore held as pairs: a[slot][0] = pixels 0|1, a[slot][1] = pixels 2|3.
template <int WHICH>
__device__ __forceinline__ void slide_sym(uint64_t (&a)[ksize<WHICH>()][2], const uint32_t (&w)[4][5])
{
    constexpr int N = ksize<WHICH>(), C = N / 2;
    static_assert(table_is_symmetric<WHICH>(), "rows R and N-1-R must be equal");
    const uint32_t zero[4] = {0, 0, 0, 0};
    uint64_t c[C][2];
    // slots 0..C-1 (table rows N-1-s = rows s)
    static_for<0, C>([&](auto ss) {
        constexpr int s = ss;
        uint32_t t[4];
        row_contrib<WHICH, s>(t, w, zero);
        c[s][0] = pair64(t[0], t[1]);
        c[s][1] = pair64(t[2], t[3]);
        a[s][0] = a[s + 1][0] + c[s][0];
        a[s][1] = a[s + 1][1] + c[s][1];
    });
    // the centre row has no partner: the chain starts from the old slot C+1
    {
        const uint32_t init[4] = {lo32(a[C + 1][0]), hi32(a[C + 1][0]), lo32(a[C + 1][1]), hi32(a[C + 1][1])};
        uint32_t t[4];
        row_contrib<WHICH, C>(t, w, init);
        a[C][0] = pair64(t[0], t[1]);
        a[C][1] = pair64(t[2], t[3]);
    }
    // slots C+1..N-1 (table rows N-1-s) reuse the contribution of slot N-1-s
    static_for<C + 1, N>([&](auto ss) {
        constexpr int s = ss;
        constexpr int nx = s + 1 < N ? s + 1 : s;
        const uint64_t init = pair64(BLUR_INIT, BLUR_INIT);
        a[s][0] = (s == N - 1 ? init : a[nx][0]) + c[N - 1 - s][0];
        a[s][1] = (s == N - 1 ? init : a[nx][1]) + c[N - 1 - s][1];
    });
}

__device__ __forceinline__ uint32_t pack4_q15(const uint32_t (&a)[4])
{
    // (a>>15) of four finished accumulators into one dword, pixel p in byte p
    uint32_t lo = (a[0] >> 15) | ((a[1] >> 15) << 8);
    uint32_t hi = (a[2] >> 15) | ((a[3] >> 15) << 8);
    return lo | (hi << 16);
}

__device__ __forceinline__ uint32_t pack4_q15(const uint64_t (&a)[2])
{
    const uint32_t t[4] = {lo32(a[0]), hi32(a[0]), lo32(a[1]), hi32(a[1])};
    return pack4_q15(t);
}

// "pyr" plane: one uint4 per group of 4 pixels = { src, G1, G2, G3 }, each a dword with pixel p
// of the group in byte p.
//
// grid: x = ceil(W/4 / blockDim) blocks, y = strips, z = frames.
// GUARD: skip slot groups whose rows lie outside the strip (short strips: small batches).  Without
// it every slot is updated in every iteration, through the symmetric scheme above (long strips).
// W is the row pitch of the frames, a multiple of 4; Wl <= W the logical width: columns Wl .. W-1 hold zeros
// on input and get zeros in G1..G3 (to every later stage they are "outside the image").
template <bool GUARD>
__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ imgs, uint4 *__restrict__ pyr,
                                              int W, int H, int rows, int Wl)
{
    const int W4 = W >> 2;
    const int xq = blockIdx.x * blockDim.x + threadIdx.x;
    if (xq >= W4) return;
    const int ys = blockIdx.y * rows;
    const int ye = min(H, ys + rows);
    const size_t fofs = (size_t)blockIdx.z * (size_t)W * (size_t)H;
    const uint32_t *__restrict__ img = reinterpret_cast<const uint32_t *>(imgs + fofs);
    uint4 *__restrict__ out = pyr + (fofs >> 2);
    const int inside = Wl - 4 * xq;                  // pixels of this group inside the logical width
    const uint32_t gmask = inside >= 4 ? ~0u : inside <= 0 ? 0u : (1u << (8 * inside)) - 1u;

    // A row is read through a buffer descriptor of exactly its W bytes (none for rows outside the
    // image): the dwords left and right of the row (offsets below 0 wrap to 4 GB) and whole rows
    // above / below the image come back as zeros from the bounds check -- no masks, no row of zeros,
    // no address arithmetic in the loop.
    uint32_t qoff[7];
#pragma unroll
    for (int k = 0; k < 7; k++) qoff[k] = (uint32_t)(xq - 3 + k) * 4u;
    auto row_rsrc = [&](int i) {                     // wave-uniform
        const bool in = i >= 0 && i < H;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(img + (size_t)(in ? i : 0) * W4), 0, in ? W : 0, 0x00020000);
    };
    auto load_row = [&](int i, uint32_t (&A)[7]) {
        const __amdgpu_buffer_rsrc_t rs = row_rsrc(i);
#pragma unroll
        for (int k = 0; k < 7; k++) A[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)qoff[k], 0, 0);
    };

    // per slot 4 x u32 (GUARD) or 2 x u64 (pixel pairs, see slide_sym)
    constexpr int NP = GUARD ? 4 : 2;
    using acc_t = std::conditional_t<GUARD, uint32_t, uint64_t>;
    acc_t a3[19][NP], a2[11][NP], a1[5][NP];
#pragma unroll
    for (int s = 0; s < 19; s++)
#pragma unroll
        for (int p = 0; p < NP; p++) a3[s][p] = 0;
#pragma unroll
    for (int s = 0; s < 11; s++)
#pragma unroll
        for (int p = 0; p < NP; p++) a2[s][p] = 0;
#pragma unroll
    for (int s = 0; s < 5; s++)
#pragma unroll
        for (int p = 0; p < NP; p++) a1[s][p] = 0;
    uint32_t d1[7] = {0, 0, 0, 0, 0, 0, 0}, d2[4] = {0, 0, 0, 0};

    // c[]: the dwords of the current row; the next row is requested as soon as the windows are
    // built from them (one iteration ahead) and lands in the same registers
    uint32_t c[7];
    load_row(ys - 9, c);
    // two rows per trip: half the copies of the delay lines and of the row hand-over (four rows
    // need more than 256 VGPRs in the long-strip variant and were slower in the short-strip one)
#pragma unroll 2
    for (int i = ys - 9; i < ye + 9; ++i) {
        uint32_t w[4][5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            w[0][j] = __builtin_amdgcn_alignbyte(c[j + 1], c[j], 3);
            w[1][j] = c[j + 1];
            w[2][j] = __builtin_amdgcn_alignbyte(c[j + 2], c[j + 1], 1);
            w[3][j] = __builtin_amdgcn_alignbyte(c[j + 2], c[j + 1], 2);
        }
        load_row(i + 1, c);                   // prefetch: consumed in the next iteration
        // source dword of the row that finishes in this iteration (used at the bottom)
        const int y3 = i - 9;
        const uint32_t sp = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(row_rsrc(y3), (int)qoff[3], 0, 0);

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
            slide_sym<3>(a3, w);
            slide_sym<2>(a2, w);
            slide_sym<1>(a1, w);
        }

        // finished rows: G1 row i-2, G2 row i-5, G3 row i-9
        if (y3 >= ys) {                       // y3 < ye holds by the loop bound
            uint4 o;
            o.x = sp;
            o.y = d1[6] & gmask;
            o.z = d2[3] & gmask;
            o.w = pack4_q15(a3[0]) & gmask;
            out[(size_t)y3 * W4 + xq] = o;
        }
#pragma unroll
        for (int k = 6; k > 0; k--) d1[k] = d1[k - 1];
        d1[0] = pack4_q15(a1[0]);
#pragma unroll
        for (int k = 3; k > 0; k--) d2[k] = d2[k - 1];
        d2[0] = pack4_q15(a2[0]);

    }
}

}  // namespace sift
