// blur_mfma_kernel.h -- the three Q15 Gaussian blurs on the matrix cores (v_mfma_i32_16x16x64_i8).
//
// Restates GaussianBlurHLS + FilterKernelOperator1/2/3 + RoundToInt (DetectKeypoints.cpp:925-1043,
// :832-912, :810-827) like blur_kernel.h, bit for bit: the exact 2-D integer sum of every table.
//
// A table applied to a block of an image is a product with a banded matrix.  One MFMA computes
//     D[m][unit] = sum over k of A[m][k] * P[k][unit]          m = 0..15, k = 0..63, 16 units
// and the three index spaces are cut so that little of the band's zero area is multiplied:
//   * a unit is a block of 8 adjacent pixels (columns 8 xb .. 8 xb + 7) of a strip of rows -- any 16
//     blocks, neighbours in a row or blocks of other strips;
//   * m = 8 a + n: output pixel n = 0..7 of output row y + a, a = 0, 1 (an accumulator tile is a PAIR
//     of output rows);
//   * k = 32 b + kk: input pixel at column 8 xb - 12 + kk, kk = 0..31, of input row i + b, b = 0, 1 (an
//     operand is a PAIR of input rows; the 32-pixel window covers the 8 + 2 * 9 pixels the widest
//     table needs);
//   * A = A_d[m][k] = Kq[d + b - a + o][kk - 12 - n + o] (0 outside the table; o = 2, 5, 9), which
//     depends only on d = i - y, the distance of the input pair from the output pair: even, |d| <= 10,
//     6, 2 for the three tables.  The fragments are built at compile time, loaded once, and stay in
//     registers.
// So one MFMA applies four (output row, input row) combinations to 8 pixels of 16 units, where round 2's
// kernel (16 pixels x one row, a 64-pixel window) applied one to 16 pixels: the 19x19 table costs
// 11 MFMAs per pair of input rows and 128 pixels instead of 19 per row and 256 pixels.  Per 256 pixels
// and input row: 11 + 7 (19x19: both coefficient planes), 7 + 5 (11x11), 3 + 3 (5x5) = 36 MFMAs
// instead of 60.
//
//   * operand layout of the instruction (tools/ubench/mfma_i8_layout.hip): A: lane l holds row m = l & 15,
//     k = 16 (l >> 4) + byte.  P: lane l holds unit l & 15, k = 16 (l >> 4) + byte, i.e. the 16 bytes at
//     column 8 xb - 12 + 16 (g & 1) of input row i + (g >> 1), g = l >> 4 -- one 16-byte load per lane and
//     pair of input rows, straight from global memory (no LDS).  D: lane l, register v = m = 4 g + v:
//     pixels 4 (g & 1) + v of output row y + (g >> 1) -- four ADJACENT pixels, so the packed {src, G1, G2,
//     G3} store of blur_kernel.h's "pyr" plane needs no transposition.
//   * i8 operands are signed: pixels enter as p ^ 0x80 = p - 128, and the accumulators start at
//     16384 + 128 * sum(Kq) (RoundToInt's + 0.5 and the offset back).  A tap outside the image must read 0:
//     rows outside the image are zeros of the bounds-checked load, dwords left or right of the row are
//     masked to zero, and the same xor turns a zero into -128 = "pixel 0".
//   * table values above 127 are split K = lo + 128 hi.  The hi products run as a second accumulator
//     chain over the distances that have any (the middle ones); it is folded into the main chain with one
//     v_lshl_add per register when its last step is done -- on values finished a step earlier, so no
//     VALU instruction waits for an MFMA.
//   * vertical direction: an accumulator tile per output pair in flight (11 + 7, 7 + 5, 3 + 3), input pair
//     i updates the tile at distance d with A_d and the result moves one slot on in the same instruction
//     (D != C); slot 0 is a finished pair of output rows.
//   * two wave roles per group of 16 units so that two waves per SIMD hide each other's memory latency and
//     the work is balanced: role 0 runs the 19x19 table (18 MFMAs per step), role 1 the 11x11 and 5x5
//     tables (18).  A workgroup is 2 + 2 such waves; role 0 stores the G3 dword of a group of four pixels,
//     role 1 its {src, G1, G2} dwords (the L2 merges the two pieces of a line).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "blur_kernel.h"

namespace sift {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int BM_UNIT = 8;        // output pixels of a unit
constexpr int BM_LEFT = 12;       // the operand window starts this many pixels left of the unit (a multiple of 4)
constexpr int BM_WINDOW = 32;     // pixels of one input row in an operand

template <int WHICH> constexpr const unsigned short *koct() { return WHICH == 3 ? SIFT_G3_OCT : WHICH == 2 ? SIFT_G2_OCT : SIFT_G1_OCT; }
template <int WHICH> constexpr int ktab(int r, int c) { return oct_at(koct<WHICH>(), ksize<WHICH>(), r, c); }
// coefficient plane PL (0: K & 127, 1: K >> 7) of a tap
template <int WHICH, int PL> constexpr int kplane(int r, int c) { return PL == 0 ? (ktab<WHICH>(r, c) & 127) : (ktab<WHICH>(r, c) >> 7); }

// first table row with a value above 127 (rows F .. N-1-F have one: the tables are unimodal)
template <int WHICH>
constexpr int hi_first()
{
    constexpr int N = ksize<WHICH>();
    for (int r = 0; r < N; r++)
        for (int c = 0; c < N; c++)
            if (ktab<WHICH>(r, c) > 127) return r;
    return N;
}
template <int WHICH>
constexpr bool hi_rows_are_a_centred_range()
{
    constexpr int N = ksize<WHICH>(), F = hi_first<WHICH>();
    for (int r = 0; r < N; r++) {
        bool any = false;
        for (int c = 0; c < N; c++) {
            any |= ktab<WHICH>(r, c) > 127;
            if (ktab<WHICH>(r, c) > 127 * 128 + 127) return false;
        }
        if (any != (r >= F && r <= N - 1 - F)) return false;
    }
    return true;
}
static_assert(hi_rows_are_a_centred_range<1>() && hi_rows_are_a_centred_range<2>() && hi_rows_are_a_centred_range<3>(), "lo + 128 hi split");
static_assert(hi_first<1>() == 0 && hi_first<2>() == 1 && hi_first<3>() == 4, "rows with a second coefficient plane");
static_assert(BM_UNIT + 2 * (ksize<3>() / 2) <= BM_WINDOW && BM_LEFT >= ksize<3>() / 2 && BM_WINDOW - BM_LEFT >= BM_UNIT + ksize<3>() / 2,
              "the window holds every tap of the widest table");

// Distances d = i - y (input pair minus output pair, even) at which plane PL of table WHICH has a tap:
// |d + b - a| <= h for some a, b in {0, 1}, h = half height of the plane's rows.  Slot s holds the tile at
// distance dmax - 2 s: slot 0 takes its last product, slot nslots - 1 its first.
template <int WHICH, int PL> constexpr int half_rows() { return ksize<WHICH>() / 2 - (PL ? hi_first<WHICH>() : 0); }
template <int WHICH, int PL> constexpr int dmax() { return (half_rows<WHICH, PL>() + 1) & ~1; }
template <int WHICH, int PL> constexpr int nslots() { return dmax<WHICH, PL>() + 1; }
static_assert(nslots<3, 0>() == 11 && nslots<3, 1>() == 7 && nslots<2, 0>() == 7 && nslots<2, 1>() == 5 && nslots<1, 0>() == 3 && nslots<1, 1>() == 3,
              "36 MFMAs per step and group of units");

template <int WHICH, int PL> constexpr int frag_base()
{
    int n = 0;
    if (WHICH == 3 && PL == 0) return n;
    n += nslots<3, 0>();
    if (WHICH == 3) return n;
    n += nslots<3, 1>();
    if (WHICH == 2 && PL == 0) return n;
    n += nslots<2, 0>();
    if (WHICH == 2) return n;
    n += nslots<2, 1>();
    if (WHICH == 1 && PL == 0) return n;
    return n + nslots<1, 0>();
}
constexpr int BLUR_NFRAG = frag_base<1, 1>() + nslots<1, 1>();
static_assert(BLUR_NFRAG == 36, "one fragment per MFMA of a step");

struct BlurFrags { uint32_t w[BLUR_NFRAG][64][4]; };

template <int WHICH, int PL>
constexpr void fill_frags(BlurFrags &t)
{
    constexpr int N = ksize<WHICH>(), O = N / 2, BASE = frag_base<WHICH, PL>(), NS = nslots<WHICH, PL>(), DMAX = dmax<WHICH, PL>();
    for (int s = 0; s < NS; s++) {
        const int d = DMAX - 2 * s;
        for (int l = 0; l < 64; l++) {
            const int m = l & 15, a = m >> 3, n = m & 7, g = l >> 4;
            for (int dw = 0; dw < 4; dw++) {
                uint32_t v = 0;
                for (int by = 0; by < 4; by++) {
                    const int k = 16 * g + 4 * dw + by, b = k >> 5, kk = k & 31;
                    const int r = d + b - a + O;                     // table row of (output row y + a, input row i + b)
                    const int c = kk - BM_LEFT - n + O;              // tap of input column 8 xb - 12 + kk for output pixel n
                    const uint32_t tap = (r >= 0 && r < N && c >= 0 && c < N) ? (uint32_t)kplane<WHICH, PL>(r, c) : 0u;
                    v |= tap << (8 * by);
                }
                t.w[BASE + s][l][dw] = v;
            }
        }
    }
}
constexpr BlurFrags make_blur_frags()
{
    BlurFrags t = {};
    fill_frags<3, 0>(t); fill_frags<3, 1>(t);
    fill_frags<2, 0>(t); fill_frags<2, 1>(t);
    fill_frags<1, 0>(t); fill_frags<1, 1>(t);
    return t;
}
__device__ constexpr BlurFrags g_blur_frags = make_blur_frags();

// every tap of every table sits in exactly one fragment entry per output pixel (checked for one pixel of each row of a tile)
template <int WHICH, int PL>
constexpr long frag_sum(const BlurFrags &t, int m)
{
    long s = 0;
    for (int f = frag_base<WHICH, PL>(); f < frag_base<WHICH, PL>() + nslots<WHICH, PL>(); f++)
        for (int g = 0; g < 4; g++)
            for (int dw = 0; dw < 4; dw++)
                for (int by = 0; by < 4; by++) s += (t.w[f][16 * g + m][dw] >> (8 * by)) & 255u;
    return s;
}
template <int WHICH>
constexpr bool frags_hold_the_table()
{
    constexpr BlurFrags t = make_blur_frags();
    for (int m = 0; m < 16; m++)
        if (frag_sum<WHICH, 0>(t, m) + 128 * frag_sum<WHICH, 1>(t, m) != ksum<WHICH>()) return false;
    return true;
}
static_assert(frags_hold_the_table<1>() && frags_hold_the_table<2>() && frags_hold_the_table<3>(), "fragments = tables");

#ifndef BLUR_MFMA_WG_GROUPS
#define BLUR_MFMA_WG_GROUPS 2    // unit groups per workgroup (x 2 roles x 64 lanes)
#endif
#ifndef BLUR_STORE_AUX
#define BLUR_STORE_AUX 0     // cache policy bits of the pyr stores (bit 0 sc0, bit 1 nt, bit 4 sc1)
#endif
#ifndef BLUR_MFMA_WAVES
#define BLUR_MFMA_WAVES 2
#endif
#ifndef BLUR_MFMA_AHEAD
#define BLUR_MFMA_AHEAD 3
#endif
template <int WHICH> constexpr int blur_init() { return 16384 + 128 * (int)ksum<WHICH>(); }

template <int WHICH>
struct BlurState {
    static constexpr int JL = nslots<WHICH, 0>(), JH = nslots<WHICH, 1>();
    static constexpr int FOLD = (JL - JH) / 2;      // lo slot of the tile whose hi chain takes its last product
    v4i acc[JL];        // slot s: the output pair at distance dmax - 2 s from the current input pair
    v4i hi[JH];         // hi chain of lo slot FOLD + t
    v4i flo[JL], fhi[JH];
};

// an empty use of a register: what is in flight into it has to land first
template <class T> __device__ __forceinline__ void use_now(T &x) { asm volatile("" : "+v"(x)); }

template <int WHICH>
__device__ __forceinline__ void load_frags(BlurState<WHICH> &st, int lane)
{
    // constexpr locals: a call to a constexpr function in a loop bound is evaluated at run time
    constexpr int JL = BlurState<WHICH>::JL, JH = BlurState<WHICH>::JH, BL = frag_base<WHICH, 0>(), BH = frag_base<WHICH, 1>();
    const uint4 *__restrict__ t = reinterpret_cast<const uint4 *>(&g_blur_frags.w[0][0][0]);
    static_for<0, JL>([&](auto ss) {
        constexpr int s = ss;
        const uint4 q = t[(BL + s) * 64 + lane];
        st.flo[s] = v4i{(int)q.x, (int)q.y, (int)q.z, (int)q.w};
    });
    static_for<0, JH>([&](auto ss) {
        constexpr int s = ss;
        const uint4 q = t[(BH + s) * 64 + lane];
        st.fhi[s] = v4i{(int)q.x, (int)q.y, (int)q.z, (int)q.w};
    });
    // a use of every fragment: the compiler waits for these loads here, before the loop, instead of counting them
    // into the waits inside it (which run in every iteration)
    static_for<0, JL>([&](auto ss) { use_now(st.flo[decltype(ss)::value]); });
    static_for<0, JH>([&](auto ss) { use_now(st.fhi[decltype(ss)::value]); });
    const v4i zero = v4i{0, 0, 0, 0};
    static_for<0, JL>([&](auto ss) { st.acc[decltype(ss)::value] = zero; });
    static_for<0, JH>([&](auto ss) { st.hi[decltype(ss)::value] = zero; });
}

// one pair of input rows (operand P) into every tile of table WHICH; slot 0 is then a finished pair of output rows
template <int WHICH>
__device__ __forceinline__ void slide_mfma(BlurState<WHICH> &st, const v4i P)
{
    constexpr int JL = BlurState<WHICH>::JL, JH = BlurState<WHICH>::JH, FOLD = BlurState<WHICH>::FOLD;
    static_for<0, JL>([&](auto ss) {
        constexpr int s = ss;
        v4i cin;
        if constexpr (s == JL - 1) {
            constexpr int init = blur_init<WHICH>();
            cin = v4i{init, init, init, init};
        } else if constexpr (FOLD > 0 && s == FOLD - 1) {
            cin = st.acc[s + 1] + (st.hi[0] << 7);                       // the hi chain of slot FOLD ended a step ago
        } else {
            cin = st.acc[s + 1];
        }
        st.acc[s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.flo[s], P, cin, 0, 0, 0);
        if constexpr (s >= FOLD && s < FOLD + JH) {
            constexpr int t = s - FOLD;
            v4i hin = v4i{0, 0, 0, 0};
            if constexpr (t < JH - 1) hin = st.hi[t + 1];
            st.hi[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.fhi[t], P, hin, 0, 0, 0);
        }
    });
}

// the lane's four pixels of the finished pair of table WHICH (slot 0) as one dword, pixel p in byte p
template <int WHICH>
__device__ __forceinline__ uint32_t finished_row(const BlurState<WHICH> &st)
{
    v4i a = st.acc[0];
    if constexpr (BlurState<WHICH>::FOLD == 0) a += st.hi[0] << 7;
    const uint32_t t[4] = {(uint32_t)a[0], (uint32_t)a[1], (uint32_t)a[2], (uint32_t)a[3]};
    return pack4_q15(t);
}

// Strips of `rows` rows (even, except that one strip may be the whole frame).  A frame's units are (strip,
// block of 8 pixels) pairs, strip-major; unit group w is units 16 w .. 16 w + 15.
__host__ __device__ constexpr int blur_mfma_blocks(int W) { return (W + BM_UNIT - 1) / BM_UNIT; }

template <int ROLE>
__device__ __forceinline__ void blur_mfma_role(const uint8_t *__restrict__ imgs, uint4 *__restrict__ pyr, int W, int H, int rows,
                                               int nstrips, int group, int frame, int lane, int Wl)
{
    constexpr int AHEAD = BLUR_MFMA_AHEAD;
    const int W4 = W >> 2, NB = blur_mfma_blocks(W);
    const int u = group * 16 + (lane & 15), g = lane >> 4;
    const bool uvalid = u < NB * nstrips;
    const int strip = uvalid ? u / NB : 0, xb = uvalid ? u - strip * NB : 0;
    const int ys = strip * rows, ye = min(H, ys + rows);
    const size_t fofs = (size_t)frame * (size_t)W * (size_t)H;
    // The frame and its pyr plane through bounds-checked descriptors: offsets at or beyond the end read 0,
    // and a store to OUTSIDE is dropped, so neither loads nor stores need a branch (behind a branch the
    // compiler cannot count a store, and its waits for the row requests then wait for everything).
    // The host takes this kernel only for frames whose pyr plane stays below 2^32 - 16 bytes (OUTSIDE).
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(imgs + fofs), 0, (int)((uint32_t)W * (uint32_t)H), 0x00020000);
    const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(pyr + (fofs >> 2), 0, (int)((uint32_t)W * (uint32_t)H * 4u), 0x00020000);
    constexpr uint32_t OUTSIDE = 0xfffffff0u;

    // operand chunk of this lane: 16 bytes at column col0 of input row i + rb.  Dwords left of the row are not
    // requested (the chunk is moved right by `sh` dwords and put back in place after the load), dwords right of
    // it belong to the next row and are masked; both happen in few waves, behind wave-uniform tests.
    const int rb = g >> 1;
    const int col0 = BM_UNIT * xb - BM_LEFT + 16 * (g & 1);
    const int sh = col0 < 0 ? (-col0) >> 2 : 0;                          // 0, 1 or 3 (col0 = -4, -12)
    const int ndw = max(0, min(4, (W - col0) >> 2));                     // dwords of the chunk left of the row's end
    const bool colok = uvalid && ndw > 0;
    const bool edge_wave = __any(sh != 0 || ndw < 4);
    // the lane's own four pixels (group q of the row): source dword and the store
    const int q = 2 * xb + (g & 1);
    const bool qok = uvalid && q < W4;
    // W is the row pitch (a multiple of 4), Wl <= W the logical width: G1..G3 of columns Wl .. W-1 are zeros
    const int inside = Wl - 4 * q;
    const uint32_t gmask = inside >= 4 ? ~0u : inside <= 0 ? 0u : (1u << (8 * inside)) - 1u;

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // Offsets are plain 32-bit arithmetic, no selects: a row above the image gives a "negative" offset, i.e.
    // one near 2^32, a row below it one at or beyond the frame's size, and the bounds check answers both with
    // zeros; lanes without a valid chunk (units past the last one, chunks right of the row) read somewhere
    // inside the frame and their values are masked or never stored.
#ifdef BLUR_EXP_HOTLOAD
    const uint32_t W2 = 0u;    // timing only: every request reads the same pair of rows
#else
    const uint32_t W2 = 2u * (uint32_t)W;
#endif
    auto operand = [&](u32x4 p) -> v4i {
        if (edge_wave) {                                                 // wave-uniform
            if (sh == 1) p = u32x4{0u, p.x, p.y, p.z};
            if (sh == 3) p = u32x4{0u, 0u, 0u, p.x};
            p.x = ndw > 0 ? p.x : 0u;
            p.y = ndw > 1 ? p.y : 0u;
            p.z = ndw > 2 ? p.z : 0u;
            p.w = ndw > 3 ? p.w : 0u;
        }
        return v4i{(int)(p.x ^ 0x80808080u), (int)(p.y ^ 0x80808080u), (int)(p.z ^ 0x80808080u), (int)(p.w ^ 0x80808080u)};
    };
    // Steps over pairs of input rows, first pair ys - DMAX: rows / 2 + DMAX steps finish every pair of the strip,
    // rounded up to whole rounds of the request slots (wave-uniform): the extra ones read rows past the strip's
    // last input row and finish rows past its end, which are not stored.  Pairs are requested AHEAD steps before
    // they are used, into slots whose roles rotate: the loop body exists once per slot, so no register is copied
    // when the rows move up (a copy would have to wait for its load), and there is no guard inside a round (with
    // one the compiler copies the slots at the loop's end).
    // The compiler counts memory operations to place its waits (loads and stores share one in-order counter)
    // and takes the smallest count over every path into the loop; coming from the prologue only the requests
    // are in flight, which would make the wait before a pair's first use wait for the younger request as well.
    // The prologue therefore waits for its own requests (once per wave): the loop is then entered with nothing
    // pending and its waits are the ones its own back edge needs -- the younger requests and stores stay in
    // flight.  (Peeling the first round off instead makes the compiler rotate the request registers with copies
    // at the loop's end, and a copy waits for a load that has only just been issued.)
    const int npairs = (rows + 1) >> 1;
    // a lane stores row rel = 2 (step - DMAX) + rb of its strip while rel < len (unsigned: rel < 0 is a large number)
#ifdef BLUR_EXP_HOTSTORE
#define BLUR_SOFF(x) ((x) & 0xfffffu)     // timing only: every store lands in the frame's first megabyte (stays in the L2)
#else
#define BLUR_SOFF(x) (x)
#endif
#ifdef BLUR_EXP_NOSTORE
    const uint32_t len = 0u;   // timing only: nothing is stored
#else
    const uint32_t len = qok ? (uint32_t)(ye - ys) : 0u;
#endif

    if constexpr (ROLE == 0) {
        constexpr int DMAX = dmax<3, 0>();
        BlurState<3> s3;
        load_frags<3>(s3, lane);
        const int n = npairs + DMAX;
        uint32_t roff = (uint32_t)((ys - DMAX + rb) * W) + (colok ? (uint32_t)(col0 + 4 * sh) : 0u);   // next request
        // the step that takes input pair i finishes output pair i - DMAX: the first one ys - 2 DMAX
        uint32_t soff = ((uint32_t)((ys - 2 * DMAX + rb) * W4) + (uint32_t)q) * 16u + 12u;              // G3 dword of the row finished next
        uint32_t rel = (uint32_t)(rb - 2 * DMAX);
        u32x4 R[AHEAD];
#pragma unroll
        for (int k = 0; k < AHEAD; k++) {
            R[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)roff, 0, 0);
            roff += W2;
        }
#pragma unroll
        for (int k = 0; k < AHEAD; k++) use_now(R[k]);                   // a use: the compiler waits for the requests here
        auto body = [&](auto slot) {
            constexpr int k = decltype(slot)::value;
            const v4i P = operand(R[k]);
            R[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)roff, 0, 0);
            roff += W2;
            slide_mfma<3>(s3, P);
            __builtin_amdgcn_raw_buffer_store_b32(finished_row<3>(s3) & gmask, ws, rel < len ? (int)BLUR_SOFF(soff) : (int)OUTSIDE, 0, BLUR_STORE_AUX);
            soff += 4u * W2;                                           // two rows of W4 groups of 16 bytes
            rel += 2u;
        };
        for (int t = 0; t < n; t += AHEAD)
            static_for<0, AHEAD>([&](auto slot) { body(slot); });
    } else {
        constexpr int DMAX = dmax<2, 0>(), LAG = (dmax<2, 0>() - dmax<1, 0>()) / 2;   // G1 pairs wait LAG steps for their G2 pair
        static_assert(LAG == 2, "delay line of the 5x5 table's rows");
        BlurState<2> s2;
        BlurState<1> s1;
        load_frags<2>(s2, lane);
        load_frags<1>(s1, lane);
        constexpr int SLAG = DMAX / 2;                                  // the source pixels of a G2 pair entered SLAG steps before it is finished
        static_assert(SLAG == 3, "delay line of the source dwords");
        uint32_t d1[LAG] = {0, 0}, sq[SLAG] = {0, 0, 0};
        const int n = npairs + DMAX;
        uint32_t roff = (uint32_t)((ys - DMAX + rb) * W) + (colok ? (uint32_t)(col0 + 4 * sh) : 0u);
        uint32_t soff = ((uint32_t)((ys - 2 * DMAX + rb) * W4) + (uint32_t)q) * 16u;
        uint32_t rel = (uint32_t)(rb - 2 * DMAX);
        u32x4 R[AHEAD];
#pragma unroll
        for (int k = 0; k < AHEAD; k++) {
            R[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)roff, 0, 0);
            roff += W2;
        }
#pragma unroll
        for (int k = 0; k < AHEAD; k++) use_now(R[k]);
        typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
        auto body = [&](auto slot) {
            constexpr int k = decltype(slot)::value;
            const v4i P = operand(R[k]);
            // The lane's own four source pixels are part of its operand chunk: the chunk of an even half (g & 1 = 0)
            // starts 12 pixels left of the unit, so they are its last dword; that of an odd half starts at the
            // unit's fifth pixel, so they are its first.  No separate load of the source plane: the dword waits in a
            // delay line until the G2 pair of its rows is finished, and {src, G1, G2} leave as ONE 12-byte store.
            uint32_t src_now = (uint32_t)((g & 1) ? P[0] : P[3]) ^ 0x80808080u;
            use_now(src_now);                                          // formed here, before the slot is requested again: left to the
                                                                       // compiler the request lands in another register, is copied at the
                                                                       // loop's end, and the copy waits for a load that has only just left
            R[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)roff, 0, 0);
            roff += W2;
            slide_mfma<2>(s2, P);
            slide_mfma<1>(s1, P);
            // the G2 pair (i - 6, i - 5) and the G1 pair (i - 2, i - 1) are finished by this step
            const uint32_t so = rel < len ? BLUR_SOFF(soff) : OUTSIDE;
            // (as a 4-byte and an 8-byte store the same data takes 1.107 ms per 256 frames, as one 12-byte store 1.029)
            __builtin_amdgcn_raw_buffer_store_b96(u32x3{sq[SLAG - 1], d1[LAG - 1] & gmask, finished_row<2>(s2) & gmask}, ws, (int)so, 0, BLUR_STORE_AUX);
            soff += 4u * W2;                                           // two rows of W4 groups of 16 bytes
            rel += 2u;
            d1[1] = d1[0];
            d1[0] = finished_row<1>(s1);
            sq[2] = sq[1];
            sq[1] = sq[0];
            sq[0] = src_now;
        };
        for (int t = 0; t < n; t += AHEAD)
            static_for<0, AHEAD>([&](auto slot) { body(slot); });
    }
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(BLUR_MFMA_WAVES, BLUR_MFMA_WAVES)))
void k_blur_mfma(const uint8_t *__restrict__ imgs, uint4 *__restrict__ pyr, int W, int H, int rows, int nstrips, int Wl, int xcd_frames)
{
    // a workgroup of 256 threads takes two unit groups: its first two waves run role 0, the other two role 1
    // (two workgroups per CU, every SIMD holds two waves)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = (int)blockDim.x >> 7;
    // xcd_frames = 0: workgroup (x, y) takes unit groups of frame y.  xcd_frames = the number of frames (large
    // batches; gridDim.y is that number rounded up to a multiple of 8): the workgroups of a launch are dealt
    // round-robin over the 8 XCDs, so workgroup L runs on XCD L % 8, and XCD k takes the frames k, k + 8, ... whole --
    // the windows of neighbouring unit groups and strips then overlap in ONE L2 (1.104 -> 1.071 ms per 256 frames)
    int bx = (int)blockIdx.x, frame = (int)blockIdx.y;
    if (xcd_frames) {
        const int L = (int)(blockIdx.x + gridDim.x * blockIdx.y), idx = L >> 3;
        bx = idx % (int)gridDim.x;
        frame = (idx / (int)gridDim.x) * 8 + (L & 7);
        if (frame >= xcd_frames) return;
    }
    const int group = bx * half + (wave % half);
    const bool gvalid = group * 16 < blur_mfma_blocks(W) * nstrips;     // wave-uniform
    if (!gvalid) return;
    if (wave < half) blur_mfma_role<0>(imgs, pyr, W, H, rows, nstrips, group, frame, lane, Wl);
    else blur_mfma_role<1>(imgs, pyr, W, H, rows, nstrips, group, frame, lane, Wl);
}

}  // namespace sift
