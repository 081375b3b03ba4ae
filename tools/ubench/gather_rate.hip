// TA cost of window gathers on gfx950: each wave reads a 71-row window of a 2-byte plane either as
// 8-byte quads (19 per row) or as 16-byte chunks (10 per row), lanes = consecutive pieces, rows W*2
// bytes apart; windows at random aligned positions.  Prints ns per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// NEAR > 0: the waves of a block read windows within NEAR pixels of the block's origin (spatial neighbours)
template <int BYTES, int NEAR = 0>
__global__ void k(const char *plane, const uint32_t *origins, uint32_t *out, int W, int iters)
{
    constexpr int PER_ROW = BYTES == 8 ? 19 : 10;
    constexpr int N = 71 * PER_ROW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const char *base = plane + origins[(blockIdx.x * iters + it) & 0xffff];
        if (NEAR) base += ((wave * 5) % NEAR) * W * 2 + ((wave * 12) % NEAR) * 2 / 8 * 8;
        for (int i0 = 0; i0 < N; i0 += 256) {
            uint32_t v[4] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int i = i0 + u * 64 + lane;
                i = i < N ? i : N - 1;
                const int r = i / PER_ROW, q = i - r * PER_ROW;
                const char *p = base + (size_t)r * W * 2 + q * BYTES;
                if (BYTES == 8) { uint2 d = *(const uint2 *)p; v[u] = d.x ^ d.y; }
                else { uint4 d = *(const uint4 *)p; v[u] = d.x ^ d.y ^ d.z ^ d.w; }
            }
            acc += v[0] + v[1] + v[2] + v[3];
        }
    }
    out[(blockIdx.x * (blockDim.x >> 6) + wave) * 64 + lane] = acc;
}

// window as ROWS segments of PER_ROW 8-byte quads, segments PITCHMUL plane rows apart: the same bytes
// in fewer, longer contiguous runs (what a row-interleaved plane layout would give)
template <int ROWS, int PER_ROW, int PITCHMUL, int NEAR>
__global__ void kshape(const char *plane, const uint32_t *origins, uint32_t *out, int W, int iters)
{
    constexpr int N = ROWS * PER_ROW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const char *base = plane + origins[(blockIdx.x * iters + it) & 0xffff];
        if (NEAR) base += ((wave * 5) % NEAR) * W * 2 + ((wave * 12) % NEAR) * 2 / 8 * 8;
        for (int i0 = 0; i0 < N; i0 += 256) {
            uint32_t v[4] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int i = i0 + u * 64 + lane;
                i = i < N ? i : N - 1;
                const int r = i / PER_ROW, q = i - r * PER_ROW;
                uint2 d = *(const uint2 *)(base + (size_t)r * W * 2 * PITCHMUL + q * 8);
                v[u] = d.x ^ d.y;
            }
            acc += v[0] + v[1] + v[2] + v[3];
        }
    }
    out[(blockIdx.x * (blockDim.x >> 6) + wave) * 64 + lane] = acc;
}

// the same window from a split format: 1 byte per pixel in one plane (8-byte pieces = 8 pixels) and
// 4 bits per pixel in a second one (4-byte pieces = 8 pixels): 1.5 instead of 2 bytes per pixel
template <int NEAR>
__global__ void ksplit(const char *plane, const uint32_t *origins, uint32_t *out, int W, int iters)
{
    constexpr int PER_ROW = 10, N = 71 * PER_ROW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char *plane2 = plane + (size_t)1920 * 1080 * 32;            // second plane further on in the buffer
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint32_t o = origins[(blockIdx.x * iters + it) & 0xffff] >> 1;      // pixel index
        const char *base = plane + (o & ~7u), *base2 = plane2 + ((o >> 1) & ~3u);
        if (NEAR) { const int dy = (wave * 5) % NEAR, dx = ((wave * 12) % NEAR) & ~7; base += dy * W + dx; base2 += dy * (W / 2) + dx / 2; }
        for (int i0 = 0; i0 < N; i0 += 256) {
            uint32_t v[4] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int i = i0 + u * 64 + lane;
                i = i < N ? i : N - 1;
                const int r = i / PER_ROW, q = i - r * PER_ROW;
                uint2 d = *(const uint2 *)(base + (size_t)r * W + q * 8);
                uint32_t e = *(const uint32_t *)(base2 + (size_t)r * (W / 2) + q * 4);
                v[u] = d.x ^ d.y ^ e;
            }
            acc += v[0] + v[1] + v[2] + v[3];
        }
    }
    out[(blockIdx.x * (blockDim.x >> 6) + wave) * 64 + lane] = acc;
}

template <int ROWS, int PER_ROW, int PITCHMUL>
int shape_case(const char *plane, const uint32_t *orig, uint32_t *out, int W, int blocks, int iters)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float m1 = 0, m8 = 0;
    for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0));
        kshape<ROWS, PER_ROW, PITCHMUL, 0><<<blocks, 64>>>(plane, orig, out, W, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&m1, e0, e1));
        CHECK(hipEventRecord(e0));
        kshape<ROWS, PER_ROW, PITCHMUL, 40><<<blocks / 8, 512>>>(plane, orig, out, W, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&m8, e0, e1));
    }
    printf("shape %2d segments x %3d bytes, pitch x%d: random windows %.1f ns, neighbouring (8 waves/block) %.1f ns per window per CU\n",
           ROWS, PER_ROW * 8, PITCHMUL, m1 * 1e6 / ((double)blocks * iters / 256), m8 * 1e6 / ((double)blocks * iters / 256));
    return 0;
}

int main()
{
    const int W = 1920, H = 1080, F = 32;
    char *plane; uint32_t *orig, *out;
    CHECK(hipMalloc(&plane, (size_t)W * H * 2 * F));
    CHECK(hipMemset(plane, 1, (size_t)W * H * 2 * F));
    std::vector<uint32_t> o(65536);
    uint32_t s = 12345;
    const int blocks = 256 * 24, iters = 8;
    for (int al = 0; al < 2; al++) {
        for (auto &v : o) {                       // raster-ordered-ish: random frame, row band, column
            s = s * 1664525u + 1013904223u; uint32_t f = (s >> 8) % F;
            s = s * 1664525u + 1013904223u; uint32_t y = 40 + (s >> 8) % (H - 120);
            s = s * 1664525u + 1013904223u; uint32_t x = 40 + (s >> 8) % (W - 160);
            if (al) x &= ~7u;                     // 16-byte aligned windows
            else x &= ~3u;                        // 8-byte aligned windows
            v = ((f * H + y) * W + x) * 2;
        }
        CHECK(hipMalloc(&orig, o.size() * 4)); CHECK(hipMalloc(&out, blocks * 64 * 4 * 2));
        CHECK(hipMemcpy(orig, o.data(), o.size() * 4, hipMemcpyHostToDevice));
        if (al == 0) {
            if (shape_case<71, 19, 1>(plane, orig, out, W, blocks, iters)) return 1;
            {   // windows in raster order per frame (2048 per frame), as k_describe's slots are
                std::vector<uint32_t> so(o);
                std::sort(so.begin(), so.end());
                uint32_t *sorig; CHECK(hipMalloc(&sorig, so.size() * 4));
                CHECK(hipMemcpy(sorig, so.data(), so.size() * 4, hipMemcpyHostToDevice));
                printf("raster-ordered windows (the 'random' column; 'neighbouring' as before):\n");
                if (shape_case<71, 19, 1>(plane, sorig, out, W, blocks, iters)) return 1;
                if (shape_case<36, 38, 2>(plane, sorig, out, W, blocks, iters)) return 1;
                if (shape_case<18, 76, 4>(plane, sorig, out, W, blocks, iters)) return 1;
                if (shape_case<9, 152, 8>(plane, sorig, out, W, blocks, iters)) return 1;
                printf("random windows again:\n");
            }
            {
                hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                float m1 = 0, m8 = 0;
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipEventRecord(e0));
                    ksplit<0><<<blocks, 64>>>(plane, orig, out, W, iters);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&m1, e0, e1));
                    CHECK(hipEventRecord(e0));
                    ksplit<40><<<blocks / 8, 512>>>(plane, orig, out, W, iters);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&m8, e0, e1));
                }
                printf("split format (1 B + 4 bit per pixel, 8 + 4 byte pieces): random windows %.1f ns, neighbouring (8 waves/block) %.1f ns per window per CU\n",
                       m1 * 1e6 / ((double)blocks * iters / 256), m8 * 1e6 / ((double)blocks * iters / 256));
            }
            if (shape_case<36, 38, 2>(plane, orig, out, W, blocks, iters)) return 1;
            if (shape_case<18, 76, 4>(plane, orig, out, W, blocks, iters)) return 1;
            if (shape_case<9, 152, 8>(plane, orig, out, W, blocks, iters)) return 1;
        }
        for (int b = 0; b < 2; b++) {
            if (b == 1 && al == 0) continue;      // 16-byte loads need 16-byte alignment
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0));
                if (b == 0) k<8><<<blocks, 64>>>(plane, orig, out, W, iters);
                else k<16><<<blocks, 64>>>(plane, orig, out, W, iters);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            }
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const int per_row = b == 0 ? 19 : 10;
            const double ninstr = (double)blocks * iters * ((71 * per_row + 255) / 256) * 4;
            if (b == 0 && al == 0) {
                for (int wpb = 4; wpb <= 16; wpb *= 2) {
                    for (int rep = 0; rep < 2; rep++) {
                        CHECK(hipEventRecord(e0));
                        k<8, 40><<<blocks / wpb, 64 * wpb>>>(plane, orig, out, W, iters);
                        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    }
                    float m2; CHECK(hipEventElapsedTime(&m2, e0, e1));
                    printf("  %2d waves per block on neighbouring windows, 8-byte pieces: %.3f ms, %.1f ns per window per CU\n", wpb, m2,
                           m2 * 1e6 / ((double)blocks * iters / 256));
                    for (int rep = 0; rep < 2; rep++) {
                        CHECK(hipEventRecord(e0));
                        k<16, 40><<<blocks / wpb, 64 * wpb>>>(plane, orig, out, W, iters);
                        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    }
                    CHECK(hipEventElapsedTime(&m2, e0, e1));
                    printf("  %2d waves per block on neighbouring windows, 16-byte pieces (unaligned): %.3f ms, %.1f ns per window per CU\n", wpb, m2,
                           m2 * 1e6 / ((double)blocks * iters / 256));
                }
            }
            printf("window as %2d-byte pieces (%s-aligned): %.3f ms, %.1f ns per window per CU, %.1f cycles@2.4GHz per wave-load per CU\n",
                   b == 0 ? 8 : 16, al ? "16B" : "8B", ms, ms * 1e6 / ((double)blocks * iters / 256), ms * 1e-3 * 2.4e9 / (ninstr / 256));
        }
    }
    return 0;
}
