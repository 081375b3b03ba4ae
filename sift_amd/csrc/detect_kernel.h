// detect_kernel.h -- DoG, 3x3x3 extremum, contrast and edge tests and the gradient plane, from
// the planar "pyr" words written by k_blur.
//
// Restates build_DOG_hls (DetectKeypoints.cpp:511-559), detect_extrema_hls + IsLocalExtremum +
// IsOnEdge (:44-223, :234-308, :314-352) and ComputeGradientsHLS + ComputeGradientKernel
// (:1273-1363, :1186-1225).
//
// A lane owns 4 adjacent pixels (one uint4 of pyr = src, G1, G2, G3 dwords) and walks down
// `rows` image rows; a wave covers 256 columns.  The DoG planes are never materialised:
// D1 = G1-src, D2 = G2-G1, D3 = G3-G2 are byte differences, and every out-of-image neighbour
// reads as 0, which gives D = 0 and G2 = 0 exactly as the reference's zero fill does.
//
// The 26-neighbour test is separable: per column and row keep e = max(D1,D3), f = min(D1,D3)
// and b = D2; per column the max/min over the three rows; then a pixel's neighbourhood maximum is
//   max( E[x-1], E[x], E[x+1], Ball[x-1], Ball[x+1], max(b[y-1][x], b[y+1][x]) ).
// Left/right columns are in the lane for the inner pixels and come from the neighbouring lane by
// DPP wave shifts for the outer ones; lanes 0 and 63 compute the one column beyond the wave
// themselves (the "halo" column; the DPP shift leaves their `old` value in place).
//
// Pixels that pass the extremum and contrast tests (~1 %) are queued in LDS; at the end of the
// strip the wave runs the edge test (the one float divide of the path) on the queue, one lane per
// entry, and what passes is a pre-cull candidate (kpMap1 != 0, :190-207).
//
// Outputs per frame
//   candmask[y][seg]  u64, bit = pre-cull candidate (raster order; every word of the strip written)
//   keyplane[y][x]    u32, written ONLY at candidates: (E_q15 << 8) | |D2|
//   gradp             gradp_pack(code, mag) per pair of pixels, rows interleaved by four (gradp_quad_word; internal, 2 B/pixel)
//   gradfull[y][x]    Grad_t {ang, mag}                        (optional, sift_detect API)
// and it zeroes rowcount, which k_cull fills with atomics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "blur_kernel.h"     // static_for

namespace sift {

typedef unsigned long long u64;

// Internal gradient plane (written by k_detect, read by k_describe): 2 bytes per pixel, one dword
// per pair of pixels (E = even column, O = odd column), a = code - 1, dir = a >> 1, b = a & 1:
//   bits 0-3 zero | 4 b_E | 5-6 dir_E | 7-8 dir_O | 9-17 mag_O | 18-21 zero | 22 b_O | 23-31 mag_E
// and all fields of a pixel 0 for code 0, whose magnitude never reaches a descriptor bin.  The
// fields stand where k_describe needs them: the dword itself (E) or dword >> 18 (O) is the shift
// count 16 b (v_lshlrev looks at five bits) that puts the magnitude into its half of the LDS bin
// pair, dword & 0x180 is O's LDS row offset, dword >> 23 is E's magnitude.
__device__ __forceinline__ uint32_t gradp_pack(int ang, int mag, bool odd)
{
    const uint32_t a = (uint32_t)(ang - 1), dir = a >> 1, b = a & 1u;
    const uint32_t w = odd ? ((uint32_t)mag << 9 | dir << 7 | b << 22) : ((uint32_t)mag << 23 | dir << 5 | b << 4);
    return ang ? w : 0u;
}
// the pixel's fields in the order mag | dir << 9 | b << 15 (the pixel sweep of k_describe)
__device__ __forceinline__ uint32_t gradp_unpack(uint32_t pair, bool odd)
{
    return odd ? (((pair >> 9) & 0x1ffu) | ((pair >> 7) & 3u) << 9 | ((pair >> 22) & 1u) << 15)
               : ((pair >> 23) | ((pair >> 5) & 3u) << 9 | ((pair >> 4) & 1u) << 15);
}

// ComputeGradientKernel (:1186-1225), branch-free.
//   mag = floor(sqrt(s)), s = gx^2+gy^2 <= 130050: with s exact in binary32,
//   floor(sqrt(s + 0.5)) equals the integer square root even with a 1-ulp sqrt (the distance of
//   sqrt(s + 0.5) to the nearest integer is >= 0.25/361).
//   code (:1206-1213): 0 if gx*gy == 0; otherwise with sy = [gy<0], x = [gx*gy<0] the quadrant
//   pair is 2*(2 sy + x) + 1 and the second code of the pair is taken unless |gx| > |gy| (pairs
//   1/2, 5/6: x = 0) resp. |gx| < |gy| (pairs 3/4, 7/8: x = 1).
__device__ __forceinline__ void grad_code(int gx, int gy, int &ang, int &mag)
{
    const int gx2 = gx * gx, gy2 = gy * gy;
    mag = (int)__builtin_amdgcn_sqrtf((float)(gx2 + gy2) + 0.5f);      // v_sqrt_f32
    const int p = gx * gy;
    const int xm = p >> 31;                                            // -1 where gx, gy differ in sign
    const int dd = ((gx2 - gy2) ^ xm) - xm;                            // |gx|-|gy| signed so that "first code" is dd > 0
    const uint32_t t = (uint32_t)(dd - 1) >> 31;                       // 1 unless dd > 0
    const uint32_t sy = (uint32_t)gy >> 31;
    const int a = (int)((sy << 2) + ((uint32_t)xm & 2u) + 1u + t);
    ang = p ? a : 0;
}

// Layout of the internal gradient plane: four image rows are interleaved quad by quad, so the
// 4-pixel quad g of rows 4k .. 4k+3 is one 32-byte piece and a descriptor window touches a third
// fewer cache lines than in a row-major plane (describe_kernel.h).  Offsets in 16-bit words.
__host__ __device__ __forceinline__ size_t gradp_frame_words(int W, int H) { return (size_t)((H + 3) & ~3) * (size_t)W; }
__device__ __forceinline__ size_t gradp_quad_word(int y, int g, int W4)
{
    return ((size_t)(y >> 2) * (size_t)W4 + (size_t)g) * 16u + (size_t)(y & 3) * 4u;
}

// gradp_pack(grad_code(gx, gy)) in one go: with a - 1 = 4 sy + 2 x + t (see grad_code), dir is
// 2 sy + x and b is t; 0 where gx * gy == 0
__device__ __forceinline__ uint32_t grad_word(int gx, int gy, bool odd)
{
    const int gx2 = gx * gx, gy2 = gy * gy;
    const uint32_t mag = (uint32_t)(int)__builtin_amdgcn_sqrtf((float)(gx2 + gy2) + 0.5f);
    const int p = gx * gy;
    const int xm = p >> 31;
    const int dd = ((gx2 - gy2) ^ xm) - xm;
    const uint32_t t = (uint32_t)(dd - 1) >> 31;
    const uint32_t sy = (uint32_t)gy >> 31;
    const uint32_t w = odd ? (mag << 9 | (sy << 8 | ((uint32_t)xm & 0x80u)) | t << 22)
                           : (mag << 23 | (sy << 6 | ((uint32_t)xm & 0x20u)) | t << 4);
    return p ? w : 0u;
}

__device__ __forceinline__ int byte_of(uint32_t w, int p) { return (int)((w >> (8 * p)) & 0xffu); }

// lane i <- lane i-1 (wave_shr:1) / lane i+1 (wave_shl:1); lanes without a source keep `old`
__device__ __forceinline__ int from_left(int old, int v)
{
    return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int from_right(int old, int v)
{
    return __builtin_amdgcn_update_dpp(old, v, 0x130, 0xf, 0xf, false);
}
// OR over aligned groups of 8 lanes (result in every lane of the group)
__device__ __forceinline__ uint32_t or8(uint32_t x)
{
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, true);   // row_half_mirror
    return x;
}

// ---- two pixels per instruction: packed binary16 --------------------------------------------------
// The byte planes and their differences (|D| <= 255) are integers that binary16 holds exactly, and gfx950
// has three-input packed maxima and minima (v_pk_maximum3_f16 / v_pk_minimum3_f16: one wave-instruction
// per four cycles like every other VOP3, tools/ubench/valu_rate.hip), so the 26-neighbour test runs on
// pairs of pixels.  A byte b enters as the half 0x6400 | b = 1024 + b (one v_perm_b32 per pair and plane);
// the bias drops out of every difference.  Pairs are (pixel 0, pixel 2) = "even" and (pixel 1, pixel 3) =
// "odd" of a lane's four pixels, low half first: the right-hand neighbours of the even pair are the odd pair
// as it stands, the left-hand ones the odd pair moved up one half (v_alignbit with the neighbouring lane's).
typedef uint32_t h2;
constexpr uint32_t H2_BIAS = 0x64646464u;             // v_perm source: the exponent byte of 1024 + b
constexpr uint32_t H2_SEL_EVEN = 0x04020400u;         // -> {0x64, byte 2, 0x64, byte 0}
constexpr uint32_t H2_SEL_ODD = 0x04030401u;          // -> {0x64, byte 3, 0x64, byte 1}
__device__ __forceinline__ h2 h2_sub(h2 a, h2 b) { h2 d; asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ h2 h2_max(h2 a, h2 b) { h2 d; asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ h2 h2_min(h2 a, h2 b) { h2 d; asm("v_pk_min_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ h2 h2_max3(h2 a, h2 b, h2 c) { h2 d; asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ h2 h2_min3(h2 a, h2 b, h2 c) { h2 d; asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ h2 h2_add(h2 a, h2 b) { h2 d; asm("v_pk_add_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// a + b - 2 c
__device__ __forceinline__ h2 h2_second_diff(h2 a, h2 b, h2 c) { h2 d; asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(d) : "v"(c), "v"(0xc000c000u), "v"(h2_add(a, b))); return d; }
// half HI of x in the low half, half HI of y in the high half of the result
template <int HI> __device__ __forceinline__ uint32_t h2_pick(h2 x, h2 y) { return __builtin_amdgcn_perm(y, x, HI ? 0x07060302u : 0x05040100u); }
__device__ __forceinline__ float h2_lo_float(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float h2_hi_float(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
// lane masks of a comparison of one half (HI = 0: low, 1: high) of two pairs; inactive lanes give 0
template <int HI> __device__ __forceinline__ u64 h2_gt(h2 a, h2 b)
{
    u64 m;
    if constexpr (HI) asm("v_cmp_gt_f16_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:WORD_1" : "=s"(m) : "v"(a), "v"(b));
    else asm("v_cmp_gt_f16_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:WORD_0" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
template <int HI> __device__ __forceinline__ u64 h2_abs_ge(h2 a, h2 b)     // |a| >= b
{
    u64 m;
    if constexpr (HI) asm("v_cmp_ge_f16_sdwa %0, |%1|, %2 src0_sel:WORD_1 src1_sel:WORD_1" : "=s"(m) : "v"(a), "v"(b));
    else asm("v_cmp_ge_f16_sdwa %0, |%1|, %2 src0_sel:WORD_0 src1_sel:WORD_0" : "=s"(m) : "v"(a), "v"(b));
    return m;
}

// D2 = G2 - G1 at (y, x) from the planar pyr plane, 0 outside the image.  Branch-free (clamped
// address, masked result) so that the nine calls of one edge test load in parallel.
__device__ __forceinline__ int pyr_d2_at(const uint4 *__restrict__ fp, int W4, int W, int H, int y, int x)
{
    const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const int yc = ok ? y : 0, xc = ok ? x : 0;
    const uint32_t *q = reinterpret_cast<const uint32_t *>(fp + (size_t)yc * W4 + (xc >> 2));
    const uint32_t g1 = q[1], g2 = q[2];
    const int sh = (xc & 3) * 8;
    const int d = (int)((g2 >> sh) & 0xffu) - (int)((g1 >> sh) & 0xffu);
    return ok ? d : 0;
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

// The same from the three second differences as exact binary32 integers (|fxx|, |fyy|, |fxy| <= 1020, so the trace
// squared and the determinant stay below 2^24: every product and difference is exact, fused or not).
__device__ __forceinline__ int edge_ratio_q15_f(float fxx, float fyy, float fxy)
{
    const float tr = fxx + fyy;
    const float det = fxx * fyy - fxy * fxy;
    if (det == 0.0f) return -1;
    const float r = __fdiv_rn(tr * tr, det);
    if (r < 0.0f || r > 12.0f) return -1;            // EIGEN_RATIO_TH, DetectKeypoints.hpp:18
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

constexpr int DETECT_ROWS = 32;                       // most rows per strip (LDS is sized for it)
// LDS queue of the pixels that passed the extremum and contrast tests: position, the three second differences
// of the edge test (:314-352) and D2 itself, 12 bytes.  A row adds at most 256 entries (4 pixels x 64 lanes); the
// queue is drained (edge test) at the strip's end and after any row that leaves fewer than 256 free entries, so no
// density of extrema can overflow it.  (A strict maximum and a strict minimum may be neighbours, so "never
// 8-adjacent" bounds only each kind by 1/4; real frames stay near 1 %, about 80 entries per strip.)
#ifndef DETECT_QUEUE_ENTRIES
#define DETECT_QUEUE_ENTRIES 512
#endif
constexpr int DETECT_QUEUE = DETECT_QUEUE_ENTRIES;
static_assert(DETECT_QUEUE >= 2 * 256, "a row must fit after a drain");


// grid: x = ceil(W/256) spans, y = strips of `rows` <= DETECT_ROWS, z = frames; block = 64 (one wave)
template <bool FULLGRAD>
__global__ __launch_bounds__(64) void k_detect(const uint4 *__restrict__ pyr,
                                               u64 *__restrict__ candmask, uint32_t *__restrict__ keyplane,
                                               uint32_t *__restrict__ rowcount,
                                               uint16_t *__restrict__ gradp, uint32_t *__restrict__ gradfull,
                                               int W, int H, int S, int contrastTh, int rows, int qcap, int Wl)
{
    __shared__ uint32_t queue[3][DETECT_QUEUE];       // [0]: y << 16 | x, [1]: fxx | fyy << 16, [2]: fxy | D2 << 16 (binary16)
    __shared__ u64 cmap[DETECT_ROWS * 4];             // candidate bits of the strip: row-major, 4 words per row

    const int lane = threadIdx.x;
    const int W4 = W >> 2;
    const int g = blockIdx.x * 64 + lane;             // 4-pixel group of this lane
    const int ys = blockIdx.y * rows;
    const int ye = min(H, ys + rows);
    const int frame = blockIdx.z;
    const size_t fpix = (size_t)frame * (size_t)W * (size_t)H;
    const uint4 *__restrict__ fp = pyr + (fpix >> 2);

    cmap[lane] = 0ull;
    cmap[lane + 64] = 0ull;
    __syncthreads();
    const bool colvalid = g < W4;
    // halo column: the pixel left of the wave for lane 0, right of it for lane 63
    const int gh = lane == 0 ? g - 1 : g + 1;

    // A row is read through a buffer descriptor of exactly its W4 groups (none for rows outside the
    // image): lanes beyond the row, the halo lanes at the frame's left and right edge (offset -16
    // wraps to 4 GB) and rows above / below the image get zeros from the bounds check, which is what
    // "outside the image reads 0" needs -- no masks, no address arithmetic in the loop.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // only lanes 0 and 63 use their halo group: the others ask for an offset beyond the row, which the bounds check
    // answers with zeros without a trip to the cache
    const uint32_t goff = (uint32_t)g * 16u, hoff = (lane == 0 || lane == 63) ? (uint32_t)gh * 16u : 0xfffffff0u;
    auto load_row = [&](int r, uint4 &Q, uint4 &Qh) {
#ifdef DET_EXP_HOTLOAD
        r = ys;                                                                  // timing only: every request reads the strip's first row
#endif
        const bool in = r >= 0 && r < H;                                         // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint4 *>(fp + (size_t)(in ? r : 0) * W4), 0, in ? W4 * 16 : 0, 0x00020000);
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)goff, 0, 0);
#ifdef DET_EXP_NOHALO
        const u32x4 b = a;                                                       // timing only
#else
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)hoff, 0, 0);
#endif
        Q = make_uint4(a.x, a.y, a.z, a.w);
        Qh = make_uint4(b.x, b.y, b.z, b.w);
    };

    // Per pair of columns and row: e = max(D1, D3), f = min(D1, D3), m = max(D1, D2, D3), n = min(D1, D2, D3)
    // and b = D2, as packed halves (even pair, odd pair).  Three rows are held (y-1, y, y+1) in three slots whose
    // roles rotate: the loop body exists once per rotation phase, so no register is copied when the rows move up.
    h2 se[3][2], sf[3][2], sm[3][2], sn[3][2], sb[3][2], hm[3], hn[3], hb[3];
    // for the edge test (:314-352): the D2 pairs left of the even pair and right of the odd pair, and the horizontal
    // differences left - right of both pairs (fxy = that of the row above minus that of the row below)
    h2 bl[3], br[3], hd[3][2];
    uint32_t g2[3], hg2[3];                           // G2 dwords of the lane's group and of the halo group
    // the halo pair: lane 0 needs pixel 3 of the group to its left (the odd pair's high half), lane 63 pixel 0 of
    // the group to its right (the even pair's low half)
    const uint32_t hsel = lane == 0 ? H2_SEL_ODD : H2_SEL_EVEN;
    // row r into slot k
    auto push = [&](const uint4 &Q, const uint4 &Qh, int k) {
        const uint32_t sel[2] = {H2_SEL_EVEN, H2_SEL_ODD};
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const h2 ps = __builtin_amdgcn_perm(H2_BIAS, Q.x, sel[j]), p1 = __builtin_amdgcn_perm(H2_BIAS, Q.y, sel[j]);
            const h2 p2 = __builtin_amdgcn_perm(H2_BIAS, Q.z, sel[j]), p3 = __builtin_amdgcn_perm(H2_BIAS, Q.w, sel[j]);
            const h2 d1 = h2_sub(p1, ps), d2 = h2_sub(p2, p1), d3 = h2_sub(p3, p2);
            se[k][j] = h2_max(d1, d3);
            sf[k][j] = h2_min(d1, d3);
            sm[k][j] = h2_max3(d1, d2, d3);
            sn[k][j] = h2_min3(d1, d2, d3);
            sb[k][j] = d2;
        }
        {
            const h2 ps = __builtin_amdgcn_perm(H2_BIAS, Qh.x, hsel), p1 = __builtin_amdgcn_perm(H2_BIAS, Qh.y, hsel);
            const h2 p2 = __builtin_amdgcn_perm(H2_BIAS, Qh.z, hsel), p3 = __builtin_amdgcn_perm(H2_BIAS, Qh.w, hsel);
            const h2 d1 = h2_sub(p1, ps), d2 = h2_sub(p2, p1), d3 = h2_sub(p3, p2);
            hm[k] = h2_max3(d1, d2, d3);
            hn[k] = h2_min3(d1, d2, d3);
            hb[k] = d2;
        }
        // (DPP shifts: every lane is active here)
        bl[k] = __builtin_amdgcn_alignbit(sb[k][1], (uint32_t)from_left((int)hb[k], (int)sb[k][1]), 16);     // columns -1, 1
        br[k] = __builtin_amdgcn_alignbit((uint32_t)from_right((int)hb[k], (int)sb[k][0]), sb[k][0], 16);    // columns 2, 4
        hd[k][0] = h2_sub(bl[k], sb[k][1]);
        hd[k][1] = h2_sub(sb[k][0], br[k]);
        g2[k] = Q.z;
        hg2[k] = Qh.z;
    };

    // prologue: rows ys-1 (slot 0) and ys (slot 1)
    // Rows are requested two iterations ahead into two request slots that alternate (no register is copied, so no
    // wait is forced early): the kernel is bound by the memory system, and the second row in flight is worth 2.2 %
    // (0.922 -> 0.901 ms per 256 frames).  In round 1, when VALU issue bound it, deeper requests were 1.3 % slower.
    uint4 Q, Qh, Qs[2], Qhs[2];
    load_row(ys - 1, Q, Qh);
    push(Q, Qh, 0);
    load_row(ys, Q, Qh);
    push(Q, Qh, 1);
    load_row(ys + 1, Qs[0], Qhs[0]);
    load_row(ys + 2, Qs[1], Qhs[1]);

    uint2 gb0 = make_uint2(0u, 0u), gb1 = gb0, gb2 = gb0;   // gradient words of rows 4k .. 4k+2 until row 4k+3 is done
    uint32_t qcount = 0;                              // wave-uniform
    const size_t frow = (size_t)frame * H;
    // Edge test on the queue, one lane per entry; what passes is a pre-cull candidate: its key and its bit in cmap
    // are written.  Everything the test needs was queued from registers: no pixel is read again.
    auto drain = [&]() {
#ifdef DET_EXP_NODRAIN
        qcount = 0;                                                              // timing only: no edge test, no candidates
        return;
#endif
        __syncthreads();
        for (uint32_t i0 = 0; i0 < qcount; i0 += 64) {
            const bool live = i0 + lane < qcount;
            const uint32_t i = live ? i0 + lane : 0u;
            const uint32_t e = queue[0][i], w1 = queue[1][i], w2 = queue[2][i];
            const int y = (int)(e >> 16), x = (int)(e & 0xffffu);
            const int ec = edge_ratio_q15_f(h2_lo_float(w1), h2_hi_float(w1), h2_lo_float(w2));
            const int ac = (int)fabsf(h2_hi_float(w2));
            if (live && ec >= 0) {                                       // :200
                keyplane[fpix + (size_t)y * W + x] = ((uint32_t)ec << 8) | (uint32_t)ac;
                atomicOr(&cmap[(y - ys) * 4 + ((x >> 6) & 3)], 1ull << (x & 63));
            }
        }
        __syncthreads();
        qcount = 0;
    };
    // |D2| >= contrastTh (:194) and |D2| != 0 (:684) in one test; |D2| <= 255, so any threshold above 256 is 256
    const uint32_t thbits = 0x6400u | (uint32_t)(min(max(contrastTh, 1), 256) & 0x3ff);               // 1024 + th ...
    const h2 thh = h2_sub(thbits | thbits << 16, 0x64006400u);                                          // ... - 1024, both halves
    const u64 colmask = __ballot(colvalid);
    // one image row; A, B, C = slots of rows y-1, y, y+1 (C is overwritten with row y+1)
    auto row = [&](int y, auto ka, auto kb, auto kc, auto ls) {
        constexpr int A = decltype(ka)::value, B = decltype(kb)::value, C = decltype(kc)::value;
        constexpr int LS = decltype(ls)::value;                          // request slot that holds row y+1
        push(Qs[LS], Qhs[LS], C);                                        // row y+1, requested two iterations ago
        load_row(y + 3, Qs[LS], Qhs[LS]);                                // (a row past the strip is harmless)

        // Neighbourhood maximum of a pixel (the 26 values of :245-304): its own column without the centre,
        // Y = max(e[y], m[y-1], m[y+1]), and the whole columns left and right of it, X = max(m[y-1], m[y], m[y+1]);
        // minima alike.  The odd pair is the even pair's right-hand columns and the even pair the odd pair's
        // left-hand ones; the other two come from the neighbouring lanes (halo pairs at the wave's ends).
        const h2 XE = h2_max3(sm[A][0], sm[B][0], sm[C][0]), XO = h2_max3(sm[A][1], sm[B][1], sm[C][1]);
        const h2 NE = h2_min3(sn[A][0], sn[B][0], sn[C][0]), NO = h2_min3(sn[A][1], sn[B][1], sn[C][1]);
        const h2 hX = h2_max3(hm[A], hm[B], hm[C]), hN = h2_min3(hn[A], hn[B], hn[C]);
        const h2 XlE = __builtin_amdgcn_alignbit(XO, (uint32_t)from_left((int)hX, (int)XO), 16);    // columns -1, 1
        const h2 NlE = __builtin_amdgcn_alignbit(NO, (uint32_t)from_left((int)hN, (int)NO), 16);
        const h2 XrO = __builtin_amdgcn_alignbit((uint32_t)from_right((int)hX, (int)XE), XE, 16);   // columns 2, 4
        const h2 NrO = __builtin_amdgcn_alignbit((uint32_t)from_right((int)hN, (int)NE), NE, 16);
        const h2 mxE = h2_max3(XlE, h2_max3(se[B][0], sm[A][0], sm[C][0]), XO);
        const h2 mxO = h2_max3(XE, h2_max3(se[B][1], sm[A][1], sm[C][1]), XrO);
        const h2 mnE = h2_min3(NlE, h2_min3(sf[B][0], sn[A][0], sn[C][0]), NO);
        const h2 mnO = h2_min3(NE, h2_min3(sf[B][1], sn[A][1], sn[C][1]), NrO);
        // strict 26-neighbour extremum (:245-304), |D2| >= contrastTh (:194), kpOut = |c| non-zero (:684): lane
        // masks straight from the comparisons (pixel p: even pair low, odd low, even high, odd high)
        const h2 cE = sb[B][0], cO = sb[B][1];
        const u64 pm[4] = {
            (h2_gt<0>(cE, mxE) | h2_gt<0>(mnE, cE)) & h2_abs_ge<0>(cE, thh) & colmask,
            (h2_gt<0>(cO, mxO) | h2_gt<0>(mnO, cO)) & h2_abs_ge<0>(cO, thh) & colmask,
            (h2_gt<1>(cE, mxE) | h2_gt<1>(mnE, cE)) & h2_abs_ge<1>(cE, thh) & colmask,
            (h2_gt<1>(cO, mxO) | h2_gt<1>(mnO, cO)) & h2_abs_ge<1>(cO, thh) & colmask,
        };
        if ((pm[0] | pm[1] | pm[2] | pm[3]) != 0ull) {
            // second differences of D2 at every pixel of the row (:318-322): fxx = L + R - 2c, fyy = U + Dn - 2c,
            // fxy = UL + DR - DL - UR -- integers of at most 1020, exact in binary16
            const h2 fxx[2] = {h2_second_diff(bl[B], sb[B][1], cE), h2_second_diff(sb[B][0], br[B], cO)};
            const h2 fyy[2] = {h2_second_diff(sb[A][0], sb[C][0], cE), h2_second_diff(sb[A][1], sb[C][1], cO)};
            const h2 fxy[2] = {h2_sub(hd[A][0], hd[C][0]), h2_sub(hd[A][1], hd[C][1])};
            static_for<0, 4>([&](auto pp) {
                constexpr int p = decltype(pp)::value, j = p & 1, hi = p >> 1;
                const u64 bal = pm[p];
                if (bal) {
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    if (((bal >> lane) & 1ull) && qcount + below < (uint32_t)DETECT_QUEUE) {
                        queue[0][qcount + below] = ((uint32_t)y << 16) | (uint32_t)(g * 4 + p);
                        queue[1][qcount + below] = h2_pick<hi>(fxx[j], fyy[j]);
                        queue[2][qcount + below] = h2_pick<hi>(fxy[j], sb[B][j]);
                    }
                    qcount += (uint32_t)__popcll(bal);
                }
            });
        }
        if (qcount + 256u > (uint32_t)qcap) drain();                     // wave-uniform: the next row might not fit

        // gradient of G2 at row y (:1189-1190): gx from the row itself, gy from rows y-1 and y+1
        // (the DPP shifts run with every lane active: a disabled source lane would leave `old`)
        const uint32_t g2b = g2[B];
        const uint32_t lw = (uint32_t)from_left((int)hg2[B], (int)g2b);       // G2 dword of the group to the left
        const uint32_t rw = (uint32_t)from_right((int)hg2[B], (int)g2b);
        if (colvalid) {
            uint32_t gp[4], gf[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int right = p < 3 ? byte_of(g2b, p + 1) : byte_of(rw, 0);
                const int left = p ? byte_of(g2b, p - 1) : byte_of(lw, 3);
                const int gx = right - left;
                const int gy = byte_of(g2[C], p) - byte_of(g2[A], p);
                if constexpr (FULLGRAD) {
                    int ang, mag;
                    grad_code(gx, gy, ang, mag);
                    gp[p] = gradp_pack(ang, mag, p & 1);
                    gf[p] = (uint32_t)ang | ((uint32_t)mag << 16);        // Grad_t {angGrad, magGrad}
                } else {
                    gp[p] = grad_word(gx, gy, p & 1);
                    gf[p] = 0;
                }
            }
            if constexpr (FULLGRAD) {
                // the caller's Grad_t plane has the logical width Wl as its pitch (W is the internal, padded one)
                uint32_t *const gdst = gradfull + ((size_t)frame * H + (size_t)y) * (size_t)Wl + (size_t)g * 4;
                if (Wl == W) {                                           // wave-uniform
                    *reinterpret_cast<uint4 *>(gdst) = make_uint4(gf[0], gf[1], gf[2], gf[3]);
                } else {
#pragma unroll
                    for (int p = 0; p < 4; p++)
                        if (g * 4 + p < Wl) gdst[p] = gf[p];
                }
            }
            uint2 o;
            o.x = gp[0] | gp[1];
            o.y = gp[2] | gp[3];
            // the four rows of a group are 32 neighbouring bytes per quad: they are kept until the
            // group is complete and leave as one piece (a strip that starts or ends inside a group
            // stores its rows one by one); y, ys and ye are wave-uniform
            uint16_t *const gq = gradp + gradp_frame_words(W, H) * (size_t)frame;
            const int yr = y & 3;
#ifdef DET_EXP_NOGRADSTORE
            if (yr == 3 && y - 3 >= ys) {                                // timing only: the gradient words are computed, not stored
                asm volatile("" :: "v"(gb0.x ^ gb0.y ^ gb1.x ^ gb1.y ^ gb2.x ^ gb2.y ^ o.x ^ o.y));
            } else if (y - yr < ys || (y | 3) >= ye) {
                asm volatile("" :: "v"(o.x ^ o.y));
#else
            if (yr == 3 && y - 3 >= ys) {
                uint4 *dst = reinterpret_cast<uint4 *>(gq + gradp_quad_word(y - 3, g, W4));
                dst[0] = make_uint4(gb0.x, gb0.y, gb1.x, gb1.y);
                dst[1] = make_uint4(gb2.x, gb2.y, o.x, o.y);
            } else if (y - yr < ys || (y | 3) >= ye) {
                *reinterpret_cast<uint2 *>(gq + gradp_quad_word(y, g, W4)) = o;
#endif
            } else if (yr == 0) {
                gb0 = o;
            } else if (yr == 1) {
                gb1 = o;
            } else {
                gb2 = o;
            }
        }
    };
    {
        using K0 = std::integral_constant<int, 0>;
        using K1 = std::integral_constant<int, 1>;
        using K2 = std::integral_constant<int, 2>;
        // (qcap is DETECT_QUEUE except in the test that forces drains, sift_hip_set_detect_queue)
        // the row slots rotate with period 3, the request slots with period 2: six bodies
        for (int y = ys; y < ye; y += 6) {
            row(y, K0{}, K1{}, K2{}, K0{});
            if (y + 1 < ye) row(y + 1, K1{}, K2{}, K0{}, K1{});
            if (y + 2 < ye) row(y + 2, K2{}, K0{}, K1{}, K0{});
            if (y + 3 < ye) row(y + 3, K0{}, K1{}, K2{}, K1{});
            if (y + 4 < ye) row(y + 4, K1{}, K2{}, K0{}, K0{});
            if (y + 5 < ye) row(y + 5, K2{}, K0{}, K1{}, K1{});
        }
    }

    drain();
    // every candidate mask word of the strip (k_cull writes every word of the survivors' mask)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int wi = lane + 64 * k, row = wi >> 2, seg = blockIdx.x * 4 + (wi & 3);
        if (ys + row < ye && seg < S) candmask[(frow + ys + row) * S + seg] = cmap[wi];
    }
    if (blockIdx.x == 0 && lane < ye - ys) rowcount[frow + ys + lane] = 0u;
}

}  // namespace sift
