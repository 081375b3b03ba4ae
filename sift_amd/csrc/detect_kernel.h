// detect_kernel.h -- DoG, 3x3x3 extremum, contrast, edge ratio and the gradient
// plane, from the interleaved "pyr" words written by k_blur.
//
// Restates build_DOG_hls (DetectKeypoints.cpp:511-559), detect_extrema_hls +
// IsLocalExtremum + IsOnEdge (:44-223, :234-308, :314-352) and
// ComputeGradientsHLS + ComputeGradientKernel (:1273-1363, :1186-1225).
//
// A wave owns a 64-pixel row segment and walks down `rows` image rows keeping
// the three most recent rows of (x-1, x, x+1) words in registers, so each
// image row costs three cached dword loads per lane and no LDS.  The DoG
// planes are never materialised: D1 = G1-src, D2 = G2-G1, D3 = G3-G2 are byte
// differences inside a word, and every out-of-image neighbour is the word 0,
// which gives D = 0 and G2 = 0 exactly as the reference's zero fill does.
//
// Outputs per frame
//   candmask[y][seg]  u64 ballot of pre-cull candidates (kpMap1 != 0)
//   keyplane[y][x]    u32, written ONLY at candidates: (E_q15 << 8) | |D2|
//   gradp[y][x]       u16 mag | (ang-1) << 9, 0 for ang 0  (internal, 2 B/pixel; describe_kernel.h)
//   gradfull[y][x]    Grad_t {ang, mag}              (optional, sift_detect API)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sift {

__device__ __forceinline__ int pyr_d1(uint32_t w) { return (int)((w >> 8) & 0xff) - (int)(w & 0xff); }
__device__ __forceinline__ int pyr_d2(uint32_t w) { return (int)((w >> 16) & 0xff) - (int)((w >> 8) & 0xff); }
__device__ __forceinline__ int pyr_d3(uint32_t w) { return (int)(w >> 24) - (int)((w >> 16) & 0xff); }
__device__ __forceinline__ int pyr_g2(uint32_t w) { return (int)((w >> 16) & 0xff); }

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

// internal 2-byte gradient word (written by k_detect, read by k_describe): mag | (code-1) << 9,
// and 0 for code 0, whose magnitude never reaches a descriptor bin
__device__ __forceinline__ uint32_t gradp_pack(int ang, int mag)
{
    return ang ? ((uint32_t)mag | ((uint32_t)(ang - 1) << 9)) : 0u;
}

// ComputeGradientKernel (:1186-1225)
__device__ __forceinline__ void grad_code(int gx, int gy, int &ang, int &mag)
{
    const int s = gx * gx + gy * gy;                 // <= 130050
    int m = (int)__fsqrt_rn((float)s);
    m -= (m * m > s) ? 1 : 0;                        // guards: exact integer sqrt whatever
    m += ((m + 1) * (m + 1) <= s) ? 1 : 0;           // the rounding of the float root
    const int ax = gx < 0 ? -gx : gx, ay = gy < 0 ? -gy : gy;
    int a = 0;
    if (gx > 0 && gy > 0) a = ax > ay ? 1 : 2;
    else if (gx < 0 && gy > 0) a = ax < ay ? 3 : 4;
    else if (gx < 0 && gy < 0) a = ax > ay ? 5 : 6;
    else if (gx > 0 && gy < 0) a = ax < ay ? 7 : 8;
    ang = a;
    mag = m;
}

// grid: x = ceil(W/256), y = strips, z = frames; block 256 (4 segments)
template <bool FULLGRAD>
__global__ __launch_bounds__(256) void k_detect(const uint32_t *__restrict__ pyr,
                                                unsigned long long *__restrict__ candmask,
                                                uint32_t *__restrict__ keyplane, uint16_t *__restrict__ gradp,
                                                uint32_t *__restrict__ gradfull, int W, int H, int S, int rows,
                                                int contrastTh)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int seg = x >> 6;
    if (seg >= S) return;                            // whole wave out (wave-uniform)
    const int ys = blockIdx.y * rows;
    const int ye = min(H, ys + rows);
    const size_t fpix = (size_t)blockIdx.z * (size_t)W * (size_t)H;
    const uint32_t *__restrict__ p = pyr + fpix;
    const bool xin = x < W;
    const bool lin = xin && x >= 1, rin = x + 1 < W;
    const int xc = xin ? x : 0;

    auto load3 = [&](int y, uint32_t (&r)[3]) {
        const bool rv = y >= 0 && y < H;
        const uint32_t *rp = p + (size_t)(rv ? y : 0) * W;
        const uint32_t a = rp[lin ? xc - 1 : xc], b = rp[xc], c = rp[rin ? xc + 1 : xc];
        r[0] = (rv && lin) ? a : 0u;
        r[1] = (rv && xin) ? b : 0u;
        r[2] = (rv && rin) ? c : 0u;
    };

    uint32_t r0[3], r1[3], r2[3];
    load3(ys - 1, r0);
    load3(ys, r1);
    for (int y = ys; y < ye; ++y) {
        load3(y + 1, r2);
        const int c = pyr_d2(r1[1]);
        const int ac = c < 0 ? -c : c;

        // strict 26-neighbour extremum (:245-304)
        int mx = -1024, mn = 1024;
        auto acc3 = [&](uint32_t w) {
            const int a = pyr_d1(w), b = pyr_d3(w);
            mx = max(mx, max(a, b));
            mn = min(mn, min(a, b));
        };
        auto acc2 = [&](uint32_t w) {
            const int m = pyr_d2(w);
            mx = max(mx, m);
            mn = min(mn, m);
        };
#pragma unroll
        for (int k = 0; k < 3; k++) { acc3(r0[k]); acc3(r1[k]); acc3(r2[k]); }
        acc2(r0[0]); acc2(r0[1]); acc2(r0[2]); acc2(r1[0]); acc2(r1[2]); acc2(r2[0]); acc2(r2[1]); acc2(r2[2]);
        const bool ext = (c > mx) || (c < mn);

        bool cand = false;
        int e = -1;
        if (xin && ext && ac >= contrastTh) {        // :194, :200
            e = edge_ratio_q15(pyr_d2(r1[0]), pyr_d2(r1[2]), pyr_d2(r0[1]), pyr_d2(r2[1]),
                               pyr_d2(r0[0]), pyr_d2(r0[2]), pyr_d2(r2[0]), pyr_d2(r2[2]), c);
            cand = (e >= 0) && (ac > 0);             // kpOut = |c| must be non-zero to count (:684)
        }
        const unsigned long long bal = __ballot(cand);
        const size_t row = (size_t)blockIdx.z * H + y;
        if ((threadIdx.x & 63) == 0) candmask[row * S + seg] = bal;
        if (cand) keyplane[fpix + (size_t)y * W + x] = ((uint32_t)e << 8) | (uint32_t)ac;

        // gradient of G2 (:1189-1190)
        if (xin) {
            int ang, mag;
            grad_code(pyr_g2(r1[2]) - pyr_g2(r1[0]), pyr_g2(r2[1]) - pyr_g2(r0[1]), ang, mag);
            gradp[fpix + (size_t)y * W + x] = (uint16_t)gradp_pack(ang, mag);
            if constexpr (FULLGRAD) gradfull[fpix + (size_t)y * W + x] = (uint32_t)ang | ((uint32_t)mag << 16);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { r0[k] = r1[k]; r1[k] = r2[k]; }
    }
}

}  // namespace sift
