// blur_mfma_kernel.h -- the three Q15 Gaussian blurs on the matrix cores (v_mfma_i32_16x16x64_i8).
//
// Restates GaussianBlurHLS + FilterKernelOperator1/2/3 + RoundToInt (DetectKeypoints.cpp:925-1043,
// :832-912, :810-827) like blur_kernel.h, bit for bit: the exact 2-D integer sum of every table.
//
// A table row applied to an image row is a 1-D convolution, i.e. a product with a banded (Toeplitz)
// matrix.  One MFMA takes 16 "units" -- blocks of 16 adjacent output pixels of one image row each,
// any 16 blocks: neighbours in a row, or blocks of other strips -- and computes, for every unit,
//     D[n][unit] = sum over k of T_r[n][k] * P[unit][k]          n = 0..15 output pixels, k = 0..63
// where P[unit][k] is the input pixel at column 16 xb - 16 + k of the unit's current input row and
// T_r[n][k] = Kq[r][k - 16 - n + o] is table row r shifted to output pixel n (o = 2, 5, 9; 0 outside
// the table).  The 64-pixel window covers the 16 + 2 * 9 pixels the widest table needs.  The
// Toeplitz fragments depend on the lane and the table row only: they are built at compile time, loaded
// once and stay in registers.  The pixel operand is one 16-byte load per lane and input row, straight
// from global memory (no LDS: the operand of every table row of an input row is the same registers).
//
//   * A = Toeplitz: lane l holds row n = l & 15, k = 16 (l >> 4) + byte.  B = pixels: lane l holds unit
//     l & 15, k = 16 (l >> 4) + byte, i.e. the 16 bytes at column 16 (xb + (l >> 4) - 1).
//     D: lane l, register v = output pixel 4 (l >> 4) + v of unit l & 15 -- four ADJACENT pixels, so the
//     packed {src, G1, G2, G3} store of blur_kernel.h's "pyr" plane needs no transposition.
//   * i8 operands are signed: pixels enter as p ^ 0x80 = p - 128, and the accumulators start at
//     16384 + 128 * sum(Kq) (RoundToInt's + 0.5 and the offset back).  A tap outside the image must read 0:
//     the bounds-checked buffer load returns 0 there, which the same xor turns into -128 = "pixel 0".
//   * table values above 127 are split K = lo + 128 hi.  The hi products run as a second accumulator
//     chain over the rows that have any (the middle ones); it is folded into the main chain with one
//     v_lshl_add per register when its last row is done -- on values finished an iteration earlier,
//     so no VALU instruction waits for an MFMA.
//   * vertical direction as in blur_kernel.h: an accumulator tile per output row in flight (19 + 11 + 5),
//     input row i updates slot s with table row N-1-s and the result moves to slot s-1 in the same
//     instruction (D != C), slot 0 is a finished output row.
//   * 60 MFMAs of 16 cycles per 256 pixels and input row = 3.75 matrix-core cycles per pixel and SIMD,
//     against ~7 VALU cycles of the dot4 kernel.  The accumulators (240 registers) and the Toeplitz
//     fragments (132) of all three tables would leave one wave per SIMD with nothing to hide memory
//     latency behind, so the work of a group of units is split over two waves: role 0 runs the 19x19
//     table (30 MFMAs per row, 184 registers), role 1 the 11x11 and 5x5 tables (30 MFMAs, 188 registers).
//     A workgroup is 2 + 2 such waves and two workgroups fill a CU, two waves per SIMD; role 0 stores the
//     G3 dword of a group of four pixels, role 1 its {src, G1, G2} dwords (the L2 merges the two pieces of a line).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "blur_kernel.h"

namespace sift {

typedef int v4i __attribute__((ext_vector_type(4)));

template <int WHICH> constexpr const unsigned short *koct() { return WHICH == 3 ? SIFT_G3_OCT : WHICH == 2 ? SIFT_G2_OCT : SIFT_G1_OCT; }
template <int WHICH> constexpr int ktab(int r, int c) { return oct_at(koct<WHICH>(), ksize<WHICH>(), r, c); }

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

// Toeplitz fragments, one per distinct table row (row r = row N-1-r) and coefficient plane
template <int WHICH> constexpr int nlo() { return ksize<WHICH>() / 2 + 1; }
template <int WHICH> constexpr int nhi() { return ksize<WHICH>() / 2 + 1 - hi_first<WHICH>(); }
template <int WHICH> constexpr int frag_base() { return WHICH == 3 ? 0 : WHICH == 2 ? nlo<3>() + nhi<3>() : nlo<3>() + nhi<3>() + nlo<2>() + nhi<2>(); }
constexpr int BLUR_NFRAG = frag_base<1>() + nlo<1>() + nhi<1>();
static_assert(BLUR_NFRAG == 33, "19 lo + 14 hi fragments");

struct BlurFrags { uint32_t w[BLUR_NFRAG][64][4]; };

template <int WHICH>
constexpr void fill_frags(BlurFrags &t)
{
    constexpr int N = ksize<WHICH>(), O = N / 2, F = hi_first<WHICH>(), BASE = frag_base<WHICH>(), NLO = nlo<WHICH>();
    for (int r = 0; r <= O; r++)
        for (int pl = 0; pl < 2; pl++) {
            if (pl == 1 && r < F) continue;
            const int f = BASE + (pl == 0 ? r : NLO + r - F);
            for (int l = 0; l < 64; l++) {
                const int n = l & 15, g = l >> 4;
                for (int d = 0; d < 4; d++) {
                    uint32_t v = 0;
                    for (int b = 0; b < 4; b++) {
                        const int k = 16 * g + 4 * d + b;            // input pixel at column 16 xb - 16 + k
                        const int c = k - 16 - n + O;                // its tap for output pixel n
                        const uint32_t tap = (c >= 0 && c < N) ? (uint32_t)ktab<WHICH>(r, c) : 0u;
                        v |= (pl == 0 ? (tap & 127u) : (tap >> 7)) << (8 * b);
                    }
                    t.w[f][l][d] = v;
                }
            }
        }
}
constexpr BlurFrags make_blur_frags()
{
    BlurFrags t = {};
    fill_frags<3>(t);
    fill_frags<2>(t);
    fill_frags<1>(t);
    return t;
}
__device__ constexpr BlurFrags g_blur_frags = make_blur_frags();

#ifndef BLUR_MFMA_AHEAD
#define BLUR_MFMA_AHEAD 2
#endif
template <int WHICH> constexpr int blur_init() { return 16384 + 128 * (int)ksum<WHICH>(); }

template <int WHICH>
struct BlurState {
    static constexpr int N = ksize<WHICH>(), F = hi_first<WHICH>(), NH = N - 2 * F;
    v4i acc[N];        // slot s: output row (current input row) - N/2 + s
    v4i hi[NH];        // hi chain of slot F + j
    v4i flo[nlo<WHICH>()], fhi[nhi<WHICH>()];
};

template <int WHICH>
__device__ __forceinline__ void load_frags(BlurState<WHICH> &st, int lane)
{
    // constexpr locals: a call to a constexpr function in a loop bound is evaluated at run time
    constexpr int NLO = nlo<WHICH>(), NHI = nhi<WHICH>(), BASE = frag_base<WHICH>();
    const uint4 *__restrict__ t = reinterpret_cast<const uint4 *>(&g_blur_frags.w[0][0][0]);
    static_for<0, NLO>([&](auto rr) {
        constexpr int r = rr;
        const uint4 q = t[(BASE + r) * 64 + lane];
        st.flo[r] = v4i{(int)q.x, (int)q.y, (int)q.z, (int)q.w};
    });
    static_for<0, NHI>([&](auto rr) {
        constexpr int r = rr;
        const uint4 q = t[(BASE + NLO + r) * 64 + lane];
        st.fhi[r] = v4i{(int)q.x, (int)q.y, (int)q.z, (int)q.w};
    });
}

// one input row (operand P) into every slot of table WHICH; returns nothing: slot 0 is then finished
template <int WHICH>
__device__ __forceinline__ void slide_mfma(BlurState<WHICH> &st, const v4i P)
{
    constexpr int N = BlurState<WHICH>::N, F = BlurState<WHICH>::F, O = N / 2;
    static_for<0, N>([&](auto ss) {
        constexpr int s = ss;
        constexpr int r = N - 1 - s, rr = r <= O ? r : N - 1 - r;       // table row and its fragment
        v4i cin;
        if constexpr (s == N - 1) {
            constexpr int init = blur_init<WHICH>();
            cin = v4i{init, init, init, init};
        } else if constexpr (F > 0 && s == F - 1) {
            cin = st.acc[s + 1] + (st.hi[0] << 7);                       // the hi chain of slot F ended an iteration ago
        } else {
            cin = st.acc[s + 1];
        }
        st.acc[s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.flo[rr], P, cin, 0, 0, 0);
        if constexpr (s >= F && s <= N - 1 - F) {
            v4i hin = v4i{0, 0, 0, 0};
            if constexpr (s < N - 1 - F) hin = st.hi[s + 1 - F];
            st.hi[s - F] = __builtin_amdgcn_mfma_i32_16x16x64_i8(st.fhi[rr - F], P, hin, 0, 0, 0);
        }
    });
}

// the finished row of table WHICH (slot 0) as one dword, pixel p of the lane's four in byte p
template <int WHICH>
__device__ __forceinline__ uint32_t finished_row(const BlurState<WHICH> &st)
{
    v4i a = st.acc[0];
    if constexpr (BlurState<WHICH>::F == 0) a += st.hi[0] << 7;
    const uint32_t t[4] = {(uint32_t)a[0], (uint32_t)a[1], (uint32_t)a[2], (uint32_t)a[3]};
    return pack4_q15(t);
}

// grid: x = ceil(unit groups per frame / g) blocks of 128 g threads (g unit groups x 2 roles), y = frames.
// A frame's units are (strip, block of 16 pixels) pairs, strip-major; unit group w is units 16 w .. 16 w + 15.
template <int ROLE>
__device__ __forceinline__ void blur_mfma_role(const uint8_t *__restrict__ imgs, uint4 *__restrict__ pyr, int W, int H, int rows,
                                               int nstrips, int group, int lane, int Wl)
{
    constexpr int AHEAD = BLUR_MFMA_AHEAD;
    const int W4 = W >> 2, NB = (W + 15) >> 4;
    const int u = group * 16 + (lane & 15), g = lane >> 4;
    const bool uvalid = u < NB * nstrips;
    const int strip = uvalid ? u / NB : 0, xb = uvalid ? u - strip * NB : 0;
    const int ys = strip * rows, ye = min(H, ys + rows);
    const size_t fofs = (size_t)blockIdx.y * (size_t)W * (size_t)H;
    // The frame and its pyr plane through bounds-checked descriptors: offsets at or beyond the end read 0,
    // and a store to OUTSIDE is dropped, so neither loads nor stores need a branch (behind a branch the
    // compiler cannot count a store, and its waits for the row requests then wait for everything).
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(imgs + fofs), 0, W * H, 0x00020000);
    const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(pyr + (fofs >> 2), 0, W * H * 4, 0x00020000);
    constexpr uint32_t OUTSIDE = 0xfffffff0u;

    // operand chunk of this lane: 16 bytes at column 16 c of the unit's input row
    const int c = xb + g - 1;
    const bool colok = uvalid && c >= 0 && 16 * c < W;
    const int ndw = min(4, (W - 16 * c) >> 2);                           // dwords of the chunk inside the row (W % 16 != 0: the last chunk is cut)
    // the lane's own four pixels (group q of the row): source dword and the store
    const int q = 4 * xb + g;
    const bool qok = uvalid && q < W4;
    // W is the row pitch (a multiple of 4), Wl <= W the logical width: G1..G3 of columns Wl .. W-1 are zeros
    const int inside = Wl - 4 * q;
    const uint32_t gmask = inside >= 4 ? ~0u : inside <= 0 ? 0u : (1u << (8 * inside)) - 1u;

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto request_row = [&](int y) -> u32x4 {                             // y differs between the lanes of a wave (strips)
        const bool ok = colok && (unsigned)y < (unsigned)H;
        const uint32_t off = ok ? (uint32_t)y * (uint32_t)W + 16u * (uint32_t)c : OUTSIDE;
        return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    };
    auto operand = [&](u32x4 p) -> v4i {
        if (W & 15) {                                                    // wave-uniform
            p.y = ndw > 1 ? p.y : 0u;
            p.z = ndw > 2 ? p.z : 0u;
            p.w = ndw > 3 ? p.w : 0u;
        }
        return v4i{(int)(p.x ^ 0x80808080u), (int)(p.y ^ 0x80808080u), (int)(p.z ^ 0x80808080u), (int)(p.w ^ 0x80808080u)};
    };
    const v4i zero = v4i{0, 0, 0, 0};
    // rows + 18 iterations (input row ys - 9 + i), rounded up to whole rounds of the request slots
    // (wave-uniform): the extra ones read rows past the strip's last input row and finish rows past its
    // end, which are not stored.  Rows are requested AHEAD iterations before they are used, into slots
    // whose roles rotate: the loop body exists once per slot, so no register is copied when the rows move
    // up (a copy would have to wait for its load), and there is no guard inside a round (with one the
    // compiler copies the slots at the loop's end).
    const int n = rows + 18;

    if constexpr (ROLE == 0) {
        BlurState<3> s3;
        load_frags<3>(s3, lane);
#pragma unroll
        for (int s = 0; s < 19; s++) s3.acc[s] = zero;
#pragma unroll
        for (int s = 0; s < BlurState<3>::NH; s++) s3.hi[s] = zero;
        u32x4 R[AHEAD];
#pragma unroll
        for (int k = 0; k < AHEAD; k++) R[k] = request_row(ys - 9 + k);
        auto body = [&](int i, auto slot) {
            constexpr int k = decltype(slot)::value;
            const int y = ys - 9 + i;
            const v4i P = operand(R[k]);
            R[k] = request_row(y + AHEAD);
            slide_mfma<3>(s3, P);
            const int y3 = y - 9;                                        // G3 row y - 9 is finished
            const bool st_ok = qok && y3 >= ys && y3 < ye;
            __builtin_amdgcn_raw_buffer_store_b32(finished_row<3>(s3) & gmask, ws, st_ok ? (int)(((uint32_t)y3 * (uint32_t)W4 + (uint32_t)q) * 16u + 12u) : (int)OUTSIDE, 0, 0);
        };
        for (int i = 0; i < n; i += AHEAD)
            static_for<0, AHEAD>([&](auto slot) { body(i + decltype(slot)::value, slot); });
    } else {
        BlurState<2> s2;
        BlurState<1> s1;
        load_frags<2>(s2, lane);
        load_frags<1>(s1, lane);
#pragma unroll
        for (int s = 0; s < 11; s++) s2.acc[s] = zero;
#pragma unroll
        for (int s = 0; s < BlurState<2>::NH; s++) s2.hi[s] = zero;
#pragma unroll
        for (int s = 0; s < 5; s++) s1.acc[s] = zero;
#pragma unroll
        for (int s = 0; s < BlurState<1>::NH; s++) s1.hi[s] = zero;
        uint32_t d1[3] = {0, 0, 0};                                      // G1 rows wait three iterations for their G2 row
        auto request_src = [&](int y5) -> uint32_t {
            const bool ok = qok && y5 >= ys && y5 < ye;
            return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (int)((uint32_t)y5 * (uint32_t)W + 4u * (uint32_t)q) : (int)OUTSIDE, 0, 0);
        };
        u32x4 R[AHEAD];
        uint32_t S[AHEAD];
#pragma unroll
        for (int k = 0; k < AHEAD; k++) {
            R[k] = request_row(ys - 9 + k);
            S[k] = request_src(ys - 14 + k);
        }
        auto body = [&](int i, auto slot) {
            constexpr int k = decltype(slot)::value;
            const int y = ys - 9 + i;
            const v4i P = operand(R[k]);
            const uint32_t sp = S[k];
            R[k] = request_row(y + AHEAD);
            S[k] = request_src(y - 5 + AHEAD);
            slide_mfma<2>(s2, P);
            slide_mfma<1>(s1, P);
            const int y5 = y - 5;                                        // G2 row y - 5 and G1 row y - 2 are finished
            const bool st_ok = qok && y5 >= ys && y5 < ye;
            typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
            __builtin_amdgcn_raw_buffer_store_b96(u32x3{sp, d1[2] & gmask, finished_row<2>(s2) & gmask}, ws,
                                                  st_ok ? (int)(((uint32_t)y5 * (uint32_t)W4 + (uint32_t)q) * 16u) : (int)OUTSIDE, 0, 0);
            d1[2] = d1[1];
            d1[1] = d1[0];
            d1[0] = finished_row<1>(s1);
        };
        for (int i = 0; i < n; i += AHEAD)
            static_for<0, AHEAD>([&](auto slot) { body(i + decltype(slot)::value, slot); });
    }
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_blur_mfma(const uint8_t *__restrict__ imgs, uint4 *__restrict__ pyr, int W, int H, int rows, int nstrips, int Wl)
{
    // a workgroup of 128 g threads takes g unit groups: its first g waves run role 0, the other g role 1
    // (launched with g = 2: two workgroups per CU, every SIMD holds two waves)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = (int)blockDim.x >> 7;
    const int group = blockIdx.x * half + (wave % half);
    const int NB = (W + 15) >> 4;
    if (group * 16 >= NB * nstrips) return;                              // wave-uniform
    if (wave < half) blur_mfma_role<0>(imgs, pyr, W, H, rows, nstrips, group, lane, Wl);
    else blur_mfma_role<1>(imgs, pyr, W, H, rows, nstrips, group, lane, Wl);
}

}  // namespace sift
