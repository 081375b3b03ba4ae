// match_kernel.h -- brute-force descriptor matcher that fills DMatch_t (SIFTParameters.hpp:122-128).
//
// The reference declares only the type: { ap_uint<32> distance; ap_int<16> querryIdx, trainIdx }, with the
// remark that the distance "is float in opencv" and is kept integer for the FPGA; it links OpenCV's
// features2d / flann (.cproject:93-94) but calls no matcher.  This is therefore an extension with no
// reference code to restate ("parity unpinned").  Semantics, chosen so that the struct's
// fields hold them exactly: cv::BFMatcher(NORM_L1).match() on the raw 72-word descriptors -- for every
// query row the train row of smallest sum of absolute differences (at most 72 * 517 * 360 < 2^24... x 72:
// it fits 32 bits), ties to the smaller train index; -1 / 0xffffffff when there is no train row.
//
// A lane owns one query row (72 dwords in registers); the four waves of a workgroup take every fourth train
// row, whose 72 dwords are wave-uniform (scalar loads) and go straight into v_sad_u32 as its scalar operand;
// the waves' minima meet in LDS.  Not a contraction (|a - b| does not factor), so no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sift {

struct dmatch_t { uint32_t distance; int16_t queryIdx, trainIdx; };   // DMatch_t, SIFTParameters.hpp:122-128

constexpr int MATCH_QUERIES = 64;       // per workgroup
constexpr int MATCH_WAVES = 4;

// grid: x = ceil(nquery / 64), y = pairs of descriptor sets; block = 256
__global__ __launch_bounds__(256) void k_match(const uint32_t *__restrict__ query, int nquery, const uint32_t *__restrict__ train,
                                               int ntrain, dmatch_t *__restrict__ out)
{
    __shared__ unsigned long long best[MATCH_WAVES][MATCH_QUERIES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int q = blockIdx.x * MATCH_QUERIES + lane;
    const uint32_t *__restrict__ qd = query + ((size_t)blockIdx.y * nquery + (size_t)min(q, nquery - 1)) * 72;
    const uint32_t *__restrict__ td = train + (size_t)blockIdx.y * ntrain * 72;
    uint32_t r[72];
#pragma unroll
    for (int j = 0; j < 72; j += 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(qd + j);
        r[j] = v.x; r[j + 1] = v.y; r[j + 2] = v.z; r[j + 3] = v.w;
    }
    // key = distance << 16 | train index: the minimum is the smallest distance, then the smallest index
    // (distance < 2^32 and index < 2^15, so the key fits 48 bits)
    unsigned long long key = ~0ull;
    for (int t = wave; t < ntrain; t += MATCH_WAVES) {
        const uint32_t *__restrict__ tr = td + (size_t)t * 72;          // wave-uniform: scalar loads
        uint32_t d = 0;
#pragma unroll
        for (int j = 0; j < 72; j++)                                   // d += |r[j] - tr[j]|: one instruction (hipcc has no builtin
            asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(r[j]), "s"(tr[j]), "v"(d));   // for it and compiles max - min + add to three)
        const unsigned long long k = (unsigned long long)d << 16 | (unsigned)t;
        key = k < key ? k : key;
    }
    best[wave][lane] = key;
    __syncthreads();
    if (wave == 0 && q < nquery) {
#pragma unroll
        for (int w = 1; w < MATCH_WAVES; w++) key = best[w][lane] < key ? best[w][lane] : key;
        dmatch_t m;
        m.distance = ntrain > 0 ? (uint32_t)(key >> 16) : 0xffffffffu;
        m.queryIdx = (int16_t)q;
        m.trainIdx = ntrain > 0 ? (int16_t)(key & 0xffffu) : (int16_t)-1;
        out[(size_t)blockIdx.y * nquery + q] = m;
    }
}

}  // namespace sift
