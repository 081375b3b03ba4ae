// ingest_kernel.h -- colour frame -> 8-bit gray on the device (the ingest step of main.cpp:51-58).
//
// The reference reads every frame with cv::imread(path, CV_LOAD_IMAGE_GRAYSCALE) (main.cpp:51) and
// copies the gray bytes into the hardware buffer (:55-58).  For a 24-bit BMP OpenCV's reader turns
// BGR into gray with fixed-point weights (imgcodecs utils.cpp, icvCvt_BGR2Gray_8u_C3C1R, 2.4/3.x):
//     gray = (B * 1868 + G * 9617 + R * 4899 + 8192) >> 14
// OpenCV is not part of the reference tree and is not installed here, so this formula is restated
// from its published source and is unpinned.  Rows may be stored bottom-up (BMP) and padded.
//
// HBM-bound: 3 bytes read, 1 written per pixel.  A lane converts 4 pixels from three aligned dwords.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sift {

__device__ __forceinline__ uint32_t bgr_gray(uint32_t b, uint32_t g, uint32_t r)
{
    return (b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14;
}

// grid: x = ceil(ceil(W/4) / 256), y = rows, z = frames; block = 256.  src: frames of `rows` x src_pitch
// bytes (pitch a multiple of 4, >= 3*W); flip: source row H-1-y feeds output row y (bottom-up files).
// dst: gray rows of W bytes, any W (one dword store per lane when W is a multiple of 4, bytes otherwise).
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int W, int H,
                                                  long src_pitch, long src_frame, int flip)
{
    const int xq = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (4 * xq >= W) return;
    const int sy = flip ? H - 1 - y : y;
    const uint8_t *__restrict__ row = src + (size_t)blockIdx.z * src_frame + (size_t)sy * src_pitch;
    uint8_t *__restrict__ out = dst + ((size_t)blockIdx.z * H + y) * W;
    if (4 * xq + 4 <= W) {
        const uint32_t *__restrict__ p = reinterpret_cast<const uint32_t *>(row) + (size_t)xq * 3;
        const uint32_t a = p[0], b = p[1], c = p[2];      // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
        const uint32_t g0 = bgr_gray(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u);
        const uint32_t g1 = bgr_gray(a >> 24, b & 255u, (b >> 8) & 255u);
        const uint32_t g2 = bgr_gray((b >> 16) & 255u, b >> 24, c & 255u);
        const uint32_t g3 = bgr_gray((c >> 8) & 255u, (c >> 16) & 255u, c >> 24);
        if ((W & 3) == 0) {                               // block-uniform
            reinterpret_cast<uint32_t *>(out)[xq] = g0 | g1 << 8 | g2 << 16 | g3 << 24;
        } else {
            out[4 * xq] = (uint8_t)g0; out[4 * xq + 1] = (uint8_t)g1; out[4 * xq + 2] = (uint8_t)g2; out[4 * xq + 3] = (uint8_t)g3;
        }
    } else {
        for (int x = 4 * xq; x < W; x++) out[x] = (uint8_t)bgr_gray(row[3 * x], row[3 * x + 1], row[3 * x + 2]);
    }
}

// Frames of any width into the internal layout: rows of `W` bytes (a multiple of 4), the columns from Wl on zero.
// grid: x = ceil(W/4 / 256), y = rows, z = frames; block = 256
__global__ __launch_bounds__(256) void k_pad_rows(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int Wl, int W, int H)
{
    const int xq = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (4 * xq >= W) return;
    const uint8_t *__restrict__ row = src + ((size_t)blockIdx.z * H + y) * Wl;
    uint32_t v = 0;
#pragma unroll
    for (int p = 0; p < 4; p++)
        if (4 * xq + p < Wl) v |= (uint32_t)row[4 * xq + p] << (8 * p);
    reinterpret_cast<uint32_t *>(dst + ((size_t)blockIdx.z * H + y) * W)[xq] = v;
}

// Extension (the reference runs one octave; SURVEY.md 8(f)4 lists a multi-octave mode as optional): the next
// octave's input frame = every second pixel of every second row of one blurred plane (1 = 5x5, 2 = 11x11,
// 3 = 19x19 table) of the frames a context has just processed, read from its pyr words.
// W4: groups per internal row; the output is (Wl/2) x (H/2) with pitch Wl/2.
// grid: x = ceil(Wl/2 / 256), y = H/2, z = frames; block = 256
__global__ __launch_bounds__(256) void k_next_octave(const uint4 *__restrict__ pyr, uint8_t *__restrict__ dst, int W4, int H, int W2, int H2,
                                                     int plane)
{
    const int x2 = blockIdx.x * blockDim.x + threadIdx.x, y2 = blockIdx.y;
    if (x2 >= W2) return;
    const uint4 q = pyr[((size_t)blockIdx.z * H + 2 * y2) * W4 + (x2 >> 1)];       // pixel 2 x2 lies in group x2 / 2, byte 2 (x2 & 1)
    const uint32_t w = plane == 1 ? q.y : plane == 2 ? q.z : q.w;
    dst[((size_t)blockIdx.z * H2 + y2) * W2 + x2] = (uint8_t)(w >> (16 * (x2 & 1)));
}

}  // namespace sift
