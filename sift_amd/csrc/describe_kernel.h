// describe_kernel.h -- main orientation + 9x8 circular-region descriptor,
// one wave per keypoint.
//
// Restates compute_main_orientation + ComputeMainOrientation
// (SIFT_Descriptor.cpp:18-103, :107-130), compute_histograms_hls (:151-278)
// and reorder_histograms_hls (:295-398).
//
// MaskAll (mask_all.hpp:8-78) is the union of the nine 25x25 circles
// (circle_mask.hpp:16-40) placed at (XStart, YStart); its column 70 is empty,
// so the reference's 71x70 orientation window (:70) sees the whole union.
// Hence, with Hin[n][a] the nine circle histograms and Ov[a] the histogram of
// the pixels that lie in two circles (no pixel lies in three):
//        orientation histogram  Hist[a] = sum_n Hin[n][a] - Ov[a]      (exact integers)
// and each of the 3809 union pixels of the 71x71 window is loaded once.
//
// A host-built table lists the union pixels in raster order, those in two
// circles first (part A, n2 entries, padded to a multiple of 64), then the
// rest (part B).  One entry is two dwords:
//   lo = (dy*W + dx) * 2          byte offset of the pixel in a 2-byte plane
//   hi = c1 << 9 | c2 << 25       circle ids, pre-shifted to LDS row offsets
// Consecutive lanes take consecutive entries, i.e. neighbouring pixels of a
// window row, so one gather instruction touches 2-3 rows of 142 bytes.
//
// Lanes l and l+32 accumulate into LDS column l of hist[row][32] (bank =
// column: no conflicts inside a half wave; ds_add_u32 without return).
//   * internal gradient format (k_detect's 2-byte plane, detect_kernel.h: gradp_pack): a pixel
//     carries mag, dir = a >> 1 and b = a & 1 with a = code-1, all 0 for code 0, so
//     every pixel is added without a test (code 0 is accumulated and then discarded by the
//     reference, :245-252).  Two direction bins share a dword as 16-bit
//     halves: row = circle*4 + dir, half = b (40 rows, 5 KB per wave).
//     A half holds at most 2 lanes x 61 pixels x 360 < 2^16.
//   * caller-supplied Grad_t planes (the stand-alone entry points): a bin is a
//     full dword, row = circle*8 + a (80 rows); codes outside 1..8 are dropped.
// The column sums are formed with 16-byte LDS reads (lane -> row i*8 + lane/8,
// columns 4*(lane%8)..+3) and an 8-lane DPP reduction.
//
// Fast path of the internal format ("quad sweep").  Gathering single 2-byte
// pixels keeps the texture addresser busy for the whole kernel (measured:
// TA_BUSY = 100 %, ~30 cache accesses per gather).  Keypoints whose window lies
// inside the plane therefore read the window as aligned 8-byte quads (4 pixels
// per lane).  The internal plane interleaves four image rows quad by quad
// (detect_kernel.h: gradp_quad_word), so the window's quads of four rows lie in
// one run of 19 x 32 bytes and a window touches 139 instead of 228 cache lines.
// A second table has one variant per alignment of the window in that plane
// ((py-35) & 3 rows into a group, (px-35) & 3 columns into a quad); its entries
// are ordered by address, consecutive lanes = consecutive 8-byte pieces:
//   lo = byte offset of the quad from the aligned window base | notov bits << 24
//   hi = 4 x (c1 | c2 << 4), one byte per pixel of the quad
// with circle id 9 = the overlap bins and 10, 11 = "not in the union" / "not an
// overlap pixel" (eight spare LDS rows that are never read), so every pixel of a quad is added without a
// test.  Quads with a two-circle pixel come first (part A, 3 adds per pixel),
// then the quads whose four pixels lie in one circle (2-byte entries, the circle's
// LDS offset formed once per entry): pairs of them in rows 2j, 2j + 1 of a group of
// four rows -- 16 neighbouring bytes of the plane, ONE gather for eight pixels (part
// U2) -- and single ones (part U), then the rest (part B, 1 add per pixel); A and B
// are padded to multiples of 64 with quads that only feed the spare rows, U2 and U
// hold whole steps only.  (Measured, timing only: a gather and its table load less
// are worth 2 % of the kernel, an LDS add less 1 %.)
//
// The gradient plane is addressed linearly like the reference does
// (pGrad[y*WIDTH + x - 35 + col], :64/:204); indices outside the plane read
// {0,0} (the reference would read out of bounds).  Keypoints whose window lies
// inside the plane -- all that sift_detect emits -- take a path without checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "compact_kernel.h"
#include "detect_kernel.h"

namespace sift {

enum { DESC_FUSED = 0, DESC_ORI_ONLY = 1, DESC_GIVEN_ORI = 2 };
enum { GRAD_PACKED16 = 0, GRAD_FULL32 = 1 };

// The quad table of one window alignment is 64 records, one per lane: the lane's entries of every step, part A
// (8 bytes each) first, then U2 and U (2 bytes each), then B (4 bytes each, on a 4-byte boundary), the record
// padded to a multiple of 16 bytes.  A lane fetches its record with a few 16-byte loads (the reference's masks:
// 5, 4 + 1 and 2 steps = 64 bytes = four loads; one load per entry costs 0.9 % of the kernel each, measured).
// In memory the records are interleaved by 16-byte piece -- piece i of lane l at (i * 64 + l) * 16 -- so that a
// load of the wave reads 1 KB of consecutive bytes.
struct QuadRecord {
    int offU2, offU, offB, size;
    __host__ __device__ constexpr QuadRecord(int na, int nu2, int nu, int nb)
        : offU2(na * 8), offU(na * 8 + nu2 * 2), offB((na * 8 + (nu2 + nu) * 2 + 3) & ~3), size((((na * 8 + (nu2 + nu) * 2 + 3) & ~3) + nb * 4 + 15) & ~15) {}
    // byte `ofs` of lane `lane`'s record, from the start of the alignment's table
    __host__ __device__ static constexpr uint32_t at(uint32_t lane, uint32_t ofs) { return ((ofs >> 4) * 64u + lane) * 16u + (ofs & 15u); }
};

constexpr int DESC_MAX_MAG_PACKED = 360;     // floor(sqrt(2*255^2)), the largest k_detect can produce
constexpr int DESC_UNROLL = 4;               // table entries a lane has in flight

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t x)   // 0 where the source lies outside the row of 16
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)   // v_pk_add_u16
{
    us2 r = __builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b);
    return __builtin_bit_cast(uint32_t, r);
}

// Caller-supplied Grad_t planes (the stand-alone entry points) into the internal plane, so that
// those calls take the quad sweep too.  The internal format holds magnitudes up to
// DESC_MAX_MAG_PACKED in 16-bit bin halves; a larger one (sift_detect produces none) raises
// bit 0 of *flag, and the call then runs the Grad_t kernel instead; one above 32767, which the
// reference's ap_int<16> MagGrad (:81) would count as negative in the orientation sum, raises bit 1 as
// well: the host-buffer entry points refuse such a plane (SIFT_HIP_EINVAL) instead of returning an
// orientation the reference would not.  Codes outside 1..8 carry no weight (:245-252), like code 0.
// grid: x = ceil(W/4 / 256), y = ceil(H/4) groups of rows, z = frames; block = 256
__global__ __launch_bounds__(256) void k_grad_pairs(const uint4 *__restrict__ grad, uint16_t *__restrict__ gradp,
                                                    uint32_t *__restrict__ flag, int W, int H)
{
    const int W4 = W >> 2, g = blockIdx.x * 256 + threadIdx.x;
    if (g >= W4) return;
    const uint4 *__restrict__ fin = grad + (size_t)blockIdx.z * (size_t)W4 * (size_t)H;
    uint32_t o[8];
    bool big = false, huge = false;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int y = blockIdx.y * 4 + r;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (y < H) q = fin[(size_t)y * W4 + g];
        const uint32_t px[4] = {q.x, q.y, q.z, q.w};
        uint32_t w[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t ang = px[p] & 0xffffu, mag = px[p] >> 16;
            const bool counts = ang >= 1u && ang <= 8u;
            big |= counts && mag > (uint32_t)DESC_MAX_MAG_PACKED;
            huge |= counts && mag > 32767u;                            // the reference's ap_int<16> MagGrad (:81) would go negative
            w[p] = counts ? gradp_pack((int)ang, (int)(mag & 0x1ffu), p & 1) : 0u;
        }
        o[2 * r] = w[0] | w[1];
        o[2 * r + 1] = w[2] | w[3];
    }
    uint4 *dst = reinterpret_cast<uint4 *>(gradp + gradp_frame_words(W, H) * (size_t)blockIdx.z + gradp_quad_word(blockIdx.y * 4, g, W4));
    dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
    dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    if (big) atomicOr(flag, huge ? 3u : 1u);
}

// The same check for planes that are swept in their own format (widths that are not a multiple of 4 skip
// k_grad_pairs): a counted magnitude above 32767 raises both bits of *flag.  grid: blocks of 256, grid-stride.
__global__ __launch_bounds__(256) void k_grad_check(const uint32_t *__restrict__ grad, uint32_t *__restrict__ flag, size_t n)
{
    bool huge = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t g = grad[i], ang = g & 0xffffu;
        huge |= ang >= 1u && ang <= 8u && (g >> 16) > 32767u;
    }
    if (huge) atomicOr(flag, 3u);
}

// grid: x = max_kp keypoints, y = frames; block = 64 (one wave)
// NA, NU2, NU, NB > 0: the quad table has exactly NA, NU2, NU and NB steps of 64 entries (the reference's masks:
// 5, 4, 1 and 2) and the sweep is straight-line code with every load of the window in flight at once
// SPEC: the geometry has orientation corrections (spec, nspec); the reference's geometry has none, and its
// kernels carry no code for them (with it the straight-line variant no longer fits its 80 registers)
template <int GFMT, int MODE, bool LOOP = false, int NA = 0, int NU2 = 0, int NU = 0, int NB = 0, bool SPEC = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GFMT == 0 ? 6 : 4, 8))) void k_describe(const void *__restrict__ gradv, const kp_t *__restrict__ kps,
                                                 const uint8_t *__restrict__ ori_in, uint8_t *__restrict__ ori_out,
                                                 uint32_t *__restrict__ desc, const uint2 *__restrict__ tab,
                                                 int n2, int n2pad, int ntab, const void *__restrict__ qtab,
                                                 int qa, int qu2, int qu, int qb, const uint32_t *__restrict__ total, int W, int H,
                                                 int max_kp, const uint32_t *__restrict__ gate, uint32_t gate_want,
                                                 int radius, int qrow4, const uint2 *__restrict__ spec, int nspec, int Wl)
{
    // W: row pitch of the internal gradient plane (a multiple of 4).  Wl <= W: the logical frame width, which is
    // the pitch of the reference's linear addressing (pGrad[y*WIDTH + x ...]) and of a caller's Grad_t plane.
    // stand-alone calls launch this kernel once per gradient format; k_grad_pairs' flag says which runs
    if (gate && (*gate & 1u) != gate_want) return;
    constexpr bool PACK = GFMT == GRAD_PACKED16;
    constexpr int ROWS = PACK ? 40 : 80;         // rows that hold bins
    constexpr int LROWS = PACK ? 48 : 80;        // + circle id 10: spare rows of the quad sweep (never read)
    constexpr int U = DESC_UNROLL;
    __shared__ __align__(16) uint32_t hist[LROWS][32];
    __shared__ __align__(8) uint32_t fin[80];
    __shared__ uint32_t corr[8];                 // orientation corrections of the geometry (spec), per direction code

    const int lane = threadIdx.x;
    // slots blockIdx.x, + gridDim.x, ...; in the fused mode `total` (k_emit's count) tells which
    // slots are padding without reading them
    const int nfilled = total ? (int)min(total[blockIdx.y], (uint32_t)max_kp) : max_kp;
    // LOOP = false: the grid has one workgroup per slot.  LOOP = true (gridDim.x a multiple of 8):
    // workgroups are dealt round-robin over the 8 XCDs, so workgroup x runs on XCD x & 7; XCD k takes
    // the k-th eighth of the filled slots -- a band of the raster-ordered keypoints, so that its L2
    // holds one band of the gradient plane instead of all of it -- and then the k-th eighth of the
    // padding slots; the workgroups of an XCD stride through its list.
    const int bs = (nfilled + 7) >> 3, pb = (max_kp - nfilled + 7) >> 3;
#ifdef DESC_EXP_HOIST
    // timing only: the wave's first keypoint and table record serve all its slots -- no dependent keypoint -> record ->
    // gather chain after the first slot
    bool hoisted = false;
    kp_t kp_keep = {};
    uint32_t rec_keep[16] = {};
#endif
    for (int v = blockIdx.x; v < (LOOP ? 8 * (bs + pb) : max_kp); v += LOOP ? (int)gridDim.x : max_kp) {
    int kslot = v;
    if constexpr (LOOP) {
        const int xcd = v & 7;
        int r = v >> 3;
        if (r < bs) {
            kslot = xcd * bs + r;
            if (kslot >= nfilled) continue;
        } else {
            r -= bs;
            kslot = nfilled + xcd * pb + r;
            if (r >= pb || kslot >= max_kp) continue;
        }
    }
    const size_t slot = (size_t)blockIdx.y * max_kp + kslot;
    if (kslot >= nfilled) {                          // padding slot: descriptor defined as 0
        desc[slot * 72 + lane] = 0;
        if (lane < 8) desc[slot * 72 + 64 + lane] = 0;
        continue;
    }
#ifdef DESC_EXP_HOIST
    if (!hoisted) kp_keep = kps[slot];
    const kp_t kp = kp_keep;
#else
    const kp_t kp = kps[slot];
#endif
#ifdef DESC_EXP_EMPTY
    const int px = 0, py = kp.y;                      // timing only: every slot takes the padding path (load, two stores)
#else
    const int px = kp.x, py = kp.y;
#endif
    uint32_t *__restrict__ out = desc ? desc + slot * 72 : nullptr;

    if (!(px > 0 && py > 0)) {                       // :52 / :196 -- padding slot
        if (MODE != DESC_ORI_ONLY) {
            out[lane] = 0;
            if (lane < 8) out[64 + lane] = 0;
        }
        if (MODE != DESC_GIVEN_ORI && ori_out && lane == 0) ori_out[slot] = 0;
        continue;
    }

    {
        // hist = 0 with ds_write_addtid_b32 (address = M0 + offset + 4 * lane: no address register, 2 cycles per
        // 256 bytes against 13 per 1024 for ds_write_b128).
        const uint32_t hist_lds = (uint32_t)(uintptr_t)&hist[0][0];   // low half of the flat address = LDS offset
        // One asm statement for the M0 write and every store that reads it: the compiler has no place between them
        // to put an M0 user of its own (indexed register moves, lane selects, LDS-direct); M0 is a reserved
        // register that it sets anew before each of its own uses, so nothing of its own relies on the value left
        // here (a clobber entry for it is refused as "reserved").  An M0 write needs one wait state before an
        // add-TID access.
        static_assert(ROWS * 32 * 4 / 256 == 20 || ROWS * 32 * 4 / 256 == 40, "the clears below");
        if constexpr (ROWS * 32 * 4 / 256 == 20)                         // the spare rows are never read: they need no clearing
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                         "ds_write_addtid_b32 %1 offset:0\n\tds_write_addtid_b32 %1 offset:256\n\tds_write_addtid_b32 %1 offset:512\n\tds_write_addtid_b32 %1 offset:768\n\t"
                         "ds_write_addtid_b32 %1 offset:1024\n\tds_write_addtid_b32 %1 offset:1280\n\tds_write_addtid_b32 %1 offset:1536\n\tds_write_addtid_b32 %1 offset:1792\n\t"
                         "ds_write_addtid_b32 %1 offset:2048\n\tds_write_addtid_b32 %1 offset:2304\n\tds_write_addtid_b32 %1 offset:2560\n\tds_write_addtid_b32 %1 offset:2816\n\t"
                         "ds_write_addtid_b32 %1 offset:3072\n\tds_write_addtid_b32 %1 offset:3328\n\tds_write_addtid_b32 %1 offset:3584\n\tds_write_addtid_b32 %1 offset:3840\n\t"
                         "ds_write_addtid_b32 %1 offset:4096\n\tds_write_addtid_b32 %1 offset:4352\n\tds_write_addtid_b32 %1 offset:4608\n\tds_write_addtid_b32 %1 offset:4864"
                         :: "s"(hist_lds), "v"(0u) : "memory");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                         "ds_write_addtid_b32 %1 offset:0\n\tds_write_addtid_b32 %1 offset:256\n\tds_write_addtid_b32 %1 offset:512\n\tds_write_addtid_b32 %1 offset:768\n\t"
                         "ds_write_addtid_b32 %1 offset:1024\n\tds_write_addtid_b32 %1 offset:1280\n\tds_write_addtid_b32 %1 offset:1536\n\tds_write_addtid_b32 %1 offset:1792\n\t"
                         "ds_write_addtid_b32 %1 offset:2048\n\tds_write_addtid_b32 %1 offset:2304\n\tds_write_addtid_b32 %1 offset:2560\n\tds_write_addtid_b32 %1 offset:2816\n\t"
                         "ds_write_addtid_b32 %1 offset:3072\n\tds_write_addtid_b32 %1 offset:3328\n\tds_write_addtid_b32 %1 offset:3584\n\tds_write_addtid_b32 %1 offset:3840\n\t"
                         "ds_write_addtid_b32 %1 offset:4096\n\tds_write_addtid_b32 %1 offset:4352\n\tds_write_addtid_b32 %1 offset:4608\n\tds_write_addtid_b32 %1 offset:4864\n\t"
                         "ds_write_addtid_b32 %1 offset:5120\n\tds_write_addtid_b32 %1 offset:5376\n\tds_write_addtid_b32 %1 offset:5632\n\tds_write_addtid_b32 %1 offset:5888\n\t"
                         "ds_write_addtid_b32 %1 offset:6144\n\tds_write_addtid_b32 %1 offset:6400\n\tds_write_addtid_b32 %1 offset:6656\n\tds_write_addtid_b32 %1 offset:6912\n\t"
                         "ds_write_addtid_b32 %1 offset:7168\n\tds_write_addtid_b32 %1 offset:7424\n\tds_write_addtid_b32 %1 offset:7680\n\tds_write_addtid_b32 %1 offset:7936\n\t"
                         "ds_write_addtid_b32 %1 offset:8192\n\tds_write_addtid_b32 %1 offset:8448\n\tds_write_addtid_b32 %1 offset:8704\n\tds_write_addtid_b32 %1 offset:8960\n\t"
                         "ds_write_addtid_b32 %1 offset:9216\n\tds_write_addtid_b32 %1 offset:9472\n\tds_write_addtid_b32 %1 offset:9728\n\tds_write_addtid_b32 %1 offset:9984"
                         :: "s"(hist_lds), "v"(0u) : "memory");
        if (SPEC && lane < 8) corr[lane] = 0u;
    }
    __syncthreads();

    const long npix = (long)Wl * H;
    const long origin = (long)(py - radius) * Wl + (px - radius);
    const int W4 = W >> 2;
    const char *__restrict__ gplane = reinterpret_cast<const char *>(gradv) +
                                      (size_t)blockIdx.y * (PACK ? gradp_frame_words(W, H) * 2 : (size_t)npix * 4);   // Grad_t planes: logical pitch
    const uint32_t col4 = (uint32_t)(lane & 31) * 4u;
    char *const hbase = reinterpret_cast<char *>(&hist[0][0]);

    // pixel sweep.  Unchecked (Grad_t planes only): every window pixel lies in the plane, so a gather
    // is base(SGPR) + 32-bit offset.  The internal plane is read pixel by pixel only for windows the
    // quad sweep cannot take (none that sift_detect emits), through its interleaved addressing.
    const char *__restrict__ gwin = gplane + origin * 4;
    auto gather = [&](auto chk, uint32_t off2, bool live) -> uint32_t {
        if constexpr (decltype(chk)::value) {
            long idx = origin + (long)(off2 >> 1);
            const bool ok = live && idx >= 0 && idx < npix;
            idx = ok ? idx : 0;
            uint32_t g;
            if constexpr (PACK) {
                const uint32_t y = (uint32_t)idx / (uint32_t)Wl, x = (uint32_t)idx - y * (uint32_t)Wl;
                const uint32_t pair = *reinterpret_cast<const uint32_t *>(gplane + gradp_quad_word((int)y, (int)(x >> 2), W4) * 2 + (x & 2u) * 2u);
                g = gradp_unpack(pair, x & 1u);
            } else {
                g = *reinterpret_cast<const uint32_t *>(gplane + idx * 4);
            }
            return ok ? g : 0u;
        } else {
            static_assert(!PACK || decltype(chk)::value, "the internal plane has no unchecked pixel sweep");
            const uint32_t g = *reinterpret_cast<const uint32_t *>(gwin + (off2 << 1));
            return live ? g : 0u;
        }
    };
    auto add = [&](uint32_t byteofs, uint32_t v) {
#ifdef DESC_EXP_NOADDS
        asm volatile("" :: "v"(byteofs), "v"(v));                       // timing only: address and value formed, no LDS instruction
#else
        atomicAdd(reinterpret_cast<uint32_t *>(hbase + byteofs), v);
#endif
    };
    auto accumulate = [&](auto two, uint32_t hi, uint32_t g) {
        if constexpr (PACK) {
            // g = gradp word (0 for code 0): value into half a&1 of row circle*4 + (a>>1)
            const uint32_t v = (g & 0x1ffu) << (g >> 11);
            const uint32_t rl = (((g >> 9) & 3u) << 7) | col4;         // (a>>1) * 128 + column
            add((hi & 0x1e00u) | rl, v);                               // circle * 512
            if constexpr (decltype(two)::value) {
                add(((hi >> 16) & 0x1e00u) | rl, v);
                add(36u * 128u + rl, v);
            }
        } else {
            const uint32_t a = (g & 0xffffu) - 1u;
            const uint32_t v = a < 8u ? (g >> 16) : 0u;
            const uint32_t rl = ((a & 7u) << 7) | col4;
            add(((hi & 0x1e00u) << 1) + rl, v);                        // circle * 1024
            if constexpr (decltype(two)::value) {
                add((((hi >> 16) & 0x1e00u) << 1) + rl, v);
                add(72u * 128u + rl, v);
            }
        }
    };
    // entries [begin, end): lane takes begin + lane, + 64, ...; U in flight.  Whole chunks run
    // without per-entry guards, the last partial chunk with them.
    const char *__restrict__ tabc = reinterpret_cast<const char *>(tab);
    auto chunk = [&](auto chk, auto two, auto guard, int t0, int end) {
        uint2 e[U];
        uint32_t g[U];
        bool live[U];
        const uint32_t tb = (uint32_t)(t0 + lane) * 8u;
#pragma unroll
        for (int u = 0; u < U; u++) {
            live[u] = decltype(guard)::value ? (t0 + u * 64 + lane < end) : true;
            e[u] = *reinterpret_cast<const uint2 *>(tabc + (live[u] ? tb + (uint32_t)u * 512u : tb));
        }
#pragma unroll
        for (int u = 0; u < U; u++) g[u] = gather(chk, e[u].x, live[u]);
#pragma unroll
        for (int u = 0; u < U; u++) accumulate(two, e[u].y, g[u]);
    };
    auto part = [&](auto chk, auto two, int begin, int end) {
        int t0 = begin;
        for (; t0 + 64 * U <= end; t0 += 64 * U) chunk(chk, two, std::false_type{}, t0, end);
        for (; t0 < end; t0 += 64 * U) chunk(chk, two, std::true_type{}, t0, end);
    };
    auto sweep = [&](auto chk) {
        part(chk, std::true_type{}, 0, n2);
        part(chk, std::false_type{}, n2pad, ntab);
    };
    // quad sweep (internal format, window inside the plane)
    // value (magnitude in its half of the bin pair) and LDS row offset of the even / odd pixel of a
    // pair dword (detect_kernel.h: gradp_pack); the shift counts rely on five-bit shifts
    auto pair_v = [](auto odd, uint32_t d) {
        if constexpr (decltype(odd)::value) return ((d >> 9) & 0x1ffu) << ((d >> 18) & 31u);
        else return (d >> 23) << (d & 31u);
    };
    auto pair_rl_at = [](auto odd, uint32_t d, uint32_t base) {
        if constexpr (decltype(odd)::value) return (d & 0x180u) | base;
        else return ((d << 2) & 0x180u) | base;
    };
    auto pair_rl = [&](auto odd, uint32_t d) { return pair_rl_at(odd, d, col4); };
    auto quad_pixel = [&](auto two, auto odd, uint32_t d, uint32_t cc, uint32_t ovsel) {
        // d = pair dword; cc = c1 | c2 << 4 in its low byte; ovsel bit 10 set = not an overlap
        // pixel: circle id 9 | 2 = 11, a spare row like id 10
        const uint32_t v = pair_v(odd, d);
        const uint32_t rl = pair_rl(odd, d);
        add(((cc & 15u) << 9) | rl, v);
#ifdef DESC_EXP_ONEADD
        if constexpr (false) {                                          // timing only: one add per pixel of a two-circle quad
#else
        if constexpr (decltype(two)::value) {
#endif
            add((((cc >> 4) & 15u) << 9) | rl, v);
            add((ovsel & 0x400u) | (rl | (9u << 9)), v);
        }
    };
    // part B entries are one dword: group of four plane rows | 4 x c1 << 9 | (quad of the row * 4 +
    // row in the group) << 25; a group of rows is W * 8 bytes, a quad's four rows 32 bytes.  The
    // first pixel's circle id stands where the LDS row offset needs it (one v_and_or).
    const uint32_t groupbytes = (uint32_t)W * 8u;
    auto b_offset = [&](uint32_t e) { return __umul24(e & 0x1fu, groupbytes) + ((e >> 22) & 0x3f8u); };   // W * 8 < 2^24
    auto b_pixels = [&](uint32_t e, uint2 q) {
        const std::false_type one;
        const std::false_type even;
        const std::true_type odd;
        add((e & 0x1e00u) | pair_rl(even, q.x), pair_v(even, q.x));
        quad_pixel(one, odd, q.x, e >> 13, 0u);
        quad_pixel(one, even, q.y, e >> 17, 0u);
        quad_pixel(one, odd, q.y, e >> 21, 0u);
    };
    // part U entries are 16 bits, for quads whose four pixels lie in one circle and in no other (most
    // of the window): (quad * 4 + row in the group) | group << 7 | circle id << 12.  The circle's LDS
    // offset is formed once per quad; an odd pixel's address is then one v_and_or.
    auto u_offset = [&](uint32_t e) { return __umul24((e >> 7) & 0x1fu, groupbytes) + ((e << 3) & 0x3f8u); };
    auto u_pixels = [&](uint32_t e, uint2 q) {
        const std::false_type even;
        const std::true_type odd;
        const uint32_t cbase = ((e >> 3) & 0x1e00u) | col4;
        add(pair_rl_at(even, q.x, cbase), pair_v(even, q.x));
        add(pair_rl_at(odd, q.x, cbase), pair_v(odd, q.x));
        add(pair_rl_at(even, q.y, cbase), pair_v(even, q.y));
        add(pair_rl_at(odd, q.y, cbase), pair_v(odd, q.y));
    };
    auto u2_pixels = [&](uint32_t e, uint4 q) {                          // rows 2j and 2j + 1 of the group, one circle
        const std::false_type even;
        const std::true_type odd;
        const uint32_t cbase = ((e >> 3) & 0x1e00u) | col4;
        add(pair_rl_at(even, q.x, cbase), pair_v(even, q.x));
        add(pair_rl_at(odd, q.x, cbase), pair_v(odd, q.x));
        add(pair_rl_at(even, q.y, cbase), pair_v(even, q.y));
        add(pair_rl_at(odd, q.y, cbase), pair_v(odd, q.y));
        add(pair_rl_at(even, q.z, cbase), pair_v(even, q.z));
        add(pair_rl_at(odd, q.z, cbase), pair_v(odd, q.z));
        add(pair_rl_at(even, q.w, cbase), pair_v(even, q.w));
        add(pair_rl_at(odd, q.w, cbase), pair_v(odd, q.w));
    };
    auto a_pixels = [&](uint2 e, uint2 q) {
        const std::true_type two;
        const std::false_type even;
        const std::true_type odd;
        quad_pixel(two, even, q.x, e.y, e.x >> 14);
        quad_pixel(two, odd, q.x, e.y >> 8, e.x >> 15);
        quad_pixel(two, even, q.y, e.y >> 16, e.x >> 16);
        quad_pixel(two, odd, q.y, e.y >> 24, e.x >> 17);
    };
    // tq: this alignment's table, 64 records (QuadRecord)
    // generic path: tq = the alignment's table, part = byte offset of the part in a record, t0 = first step of the chunk
    auto chunk_a = [&](auto nu, const char *gq, const char *tq, int t0) {
        constexpr int N = decltype(nu)::value;
        uint2 e[N], q[N];
#pragma unroll
        for (int u = 0; u < N; u++) e[u] = *reinterpret_cast<const uint2 *>(tq + QuadRecord::at((uint32_t)lane, (uint32_t)(t0 + u) * 8u));
#pragma unroll
        for (int u = 0; u < N; u++) q[u] = *reinterpret_cast<const uint2 *>(gq + (e[u].x & 0xffffffu));
#pragma unroll
        for (int u = 0; u < N; u++) a_pixels(e[u], q[u]);
    };
    auto chunk_b = [&](auto nu, const char *gq, const char *tq, int part, int t0) {
        constexpr int N = decltype(nu)::value;
        uint32_t e[N];
        uint2 q[N];
#pragma unroll
        for (int u = 0; u < N; u++) e[u] = *reinterpret_cast<const uint32_t *>(tq + QuadRecord::at((uint32_t)lane, (uint32_t)part + (uint32_t)(t0 + u) * 4u));
#pragma unroll
        for (int u = 0; u < N; u++) q[u] = *reinterpret_cast<const uint2 *>(gq + b_offset(e[u]));
#pragma unroll
        for (int u = 0; u < N; u++) b_pixels(e[u], q[u]);
    };
    auto chunk_u2 = [&](auto nu, const char *gq, const char *tq, int part, int t0) {
        constexpr int N = decltype(nu)::value;
        uint32_t e[N];
        uint4 q[N];
#pragma unroll
        for (int u = 0; u < N; u++) e[u] = *reinterpret_cast<const uint16_t *>(tq + QuadRecord::at((uint32_t)lane, (uint32_t)part + (uint32_t)(t0 + u) * 2u));
#pragma unroll
        for (int u = 0; u < N; u++) q[u] = *reinterpret_cast<const uint4 *>(gq + u_offset(e[u]));
#pragma unroll
        for (int u = 0; u < N; u++) u2_pixels(e[u], q[u]);
    };
    auto chunk_u = [&](auto nu, const char *gq, const char *tq, int part, int t0) {
        constexpr int N = decltype(nu)::value;
        uint32_t e[N];
        uint2 q[N];
#pragma unroll
        for (int u = 0; u < N; u++) e[u] = *reinterpret_cast<const uint16_t *>(tq + QuadRecord::at((uint32_t)lane, (uint32_t)part + (uint32_t)(t0 + u) * 2u));
#pragma unroll
        for (int u = 0; u < N; u++) q[u] = *reinterpret_cast<const uint2 *>(gq + u_offset(e[u]));
#pragma unroll
        for (int u = 0; u < N; u++) u_pixels(e[u], q[u]);
    };
    // A wave's time is mostly load latency (table entry -> quad -> adds), so the static variant
    // issues all NA + NU + NB table loads, then all gathers, and only then starts adding.
    auto quad_static = [&](const char *gq, const char *tq) {
        constexpr int MA = NA > 0 ? NA : 1, MU2 = NU2 > 0 ? NU2 : 1, MU = NU > 0 ? NU : 1, MB = NB > 0 ? NB : 1;
        uint2 ea[MA], qa_[MA], qu_[MU], qb_[MB];
        uint4 qu2_[MU2];
        uint32_t eu2[MU2], eu[MU], eb[MB];
        // the lane's record in 16-byte loads; the entries are dwords / halves of dwords of it
        constexpr QuadRecord R(NA, NU2, NU, NB);
        constexpr int ND = R.size > 0 ? R.size / 4 : 4;                  // (instantiated, never run, for the table-driven variants)
        uint32_t rec[ND];
#pragma unroll
        for (int i = 0; i < ND / 4; i++) {
#ifdef DESC_EXP_FEWER
            if (i == ND / 4 - 1) { rec[4 * i] = rec[0] * 3u; rec[4 * i + 1] = rec[1] * 5u; rec[4 * i + 2] = rec[2] * 7u; rec[4 * i + 3] = rec[3] * 9u; continue; }   // timing only: three table loads
#endif
#ifdef DESC_EXP_HOIST
            if (hoisted) { rec[4 * i] = rec_keep[4 * i]; rec[4 * i + 1] = rec_keep[4 * i + 1]; rec[4 * i + 2] = rec_keep[4 * i + 2]; rec[4 * i + 3] = rec_keep[4 * i + 3]; continue; }
#endif
            const uint4 v = *reinterpret_cast<const uint4 *>(tq + QuadRecord::at((uint32_t)lane, (uint32_t)i * 16u));
            rec[4 * i] = v.x; rec[4 * i + 1] = v.y; rec[4 * i + 2] = v.z; rec[4 * i + 3] = v.w;
#ifdef DESC_EXP_HOIST
            rec_keep[4 * i] = v.x; rec_keep[4 * i + 1] = v.y; rec_keep[4 * i + 2] = v.z; rec_keep[4 * i + 3] = v.w;
#endif
        }
#ifdef DESC_EXP_HOIST
        hoisted = true;
#endif
        auto half = [&](int byteofs) { return (byteofs & 2) ? rec[byteofs >> 2] >> 16 : rec[byteofs >> 2]; };   // the fields' masks drop a low half's upper bits
#pragma unroll
        for (int u = 0; u < NA; u++) ea[u] = make_uint2(rec[2 * u], rec[2 * u + 1]);
#pragma unroll
        for (int u = 0; u < NU2; u++) eu2[u] = half(R.offU2 + 2 * u);
#pragma unroll
        for (int u = 0; u < NU; u++) eu[u] = half(R.offU + 2 * u);
#pragma unroll
        for (int u = 0; u < NB; u++) eb[u] = rec[R.offB / 4 + u];
#pragma unroll
#ifdef DESC_EXP_NOGATHER
        // timing only: the quads' words are made up from the table entries, nothing is read from the plane
        for (int u = 0; u < NA; u++) qa_[u] = make_uint2(ea[u].x * 0x9E3779B1u, ea[u].y * 0x85EBCA6Bu);
#pragma unroll
        for (int u = 0; u < NU2; u++) qu2_[u] = make_uint4(eu2[u] * 0x9E3779B1u, eu2[u] * 0x85EBCA6Bu, eu2[u] * 0xC2B2AE35u, eu2[u] * 0x27D4EB2Fu);
#pragma unroll
        for (int u = 0; u < NU; u++) qu_[u] = make_uint2(eu[u] * 0x9E3779B1u, eu[u] * 0x85EBCA6Bu);
#pragma unroll
        for (int u = 0; u < NB; u++) qb_[u] = make_uint2(eb[u] * 0x9E3779B1u, eb[u] * 0x85EBCA6Bu);
#else
#ifdef DESC_EXP_FEWER
        // timing only: what a table with paired two-circle quads would cost -- two of the five A gathers made up
        for (int u = 0; u < NA; u++) qa_[u] = u < 3 ? *reinterpret_cast<const uint2 *>(gq + (ea[u].x & 0xffffffu)) : make_uint2(ea[u].x * 0x9E3779B1u, ea[u].y * 0x85EBCA6Bu);
#else
        for (int u = 0; u < NA; u++) qa_[u] = *reinterpret_cast<const uint2 *>(gq + (ea[u].x & 0xffffffu));
#endif
#pragma unroll
        for (int u = 0; u < NU2; u++) qu2_[u] = *reinterpret_cast<const uint4 *>(gq + u_offset(eu2[u]));
#pragma unroll
        for (int u = 0; u < NU; u++) qu_[u] = *reinterpret_cast<const uint2 *>(gq + u_offset(eu[u]));
#pragma unroll
        for (int u = 0; u < NB; u++) qb_[u] = *reinterpret_cast<const uint2 *>(gq + b_offset(eb[u]));
#endif
#ifdef DESC_EXP_EXTRAVALU
        {   // timing only: 64 more VALU instructions per slot (four independent chains of v_and_or_b32), no other work
            uint32_t x0 = rec[0], x1 = rec[1], x2 = rec[2], x3 = rec[3];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x1) : "v"(x2), "v"(x3));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x2) : "v"(x3), "v"(x0));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x3) : "v"(x0), "v"(x1));
            }
            asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3));
        }
#endif
#ifdef DESC_EXP_NOPIXELS
        // timing only: the window is loaded and waited for, no pixel is processed
        for (int u = 0; u < NA; u++) asm volatile("" :: "v"(qa_[u].x), "v"(qa_[u].y));
        for (int u = 0; u < NU2; u++) asm volatile("" :: "v"(qu2_[u].x), "v"(qu2_[u].y), "v"(qu2_[u].z), "v"(qu2_[u].w));
        for (int u = 0; u < NU; u++) asm volatile("" :: "v"(qu_[u].x), "v"(qu_[u].y));
        for (int u = 0; u < NB; u++) asm volatile("" :: "v"(qb_[u].x), "v"(qb_[u].y));
#else
#pragma unroll
        for (int u = 0; u < NA; u++) a_pixels(ea[u], qa_[u]);
#pragma unroll
        for (int u = 0; u < NU2; u++) u2_pixels(eu2[u], qu2_[u]);
#pragma unroll
        for (int u = 0; u < NU; u++) u_pixels(eu[u], qu_[u]);
#pragma unroll
        for (int u = 0; u < NB; u++) b_pixels(eb[u], qb_[u]);
#endif
    };
    auto quad_loops = [&](const char *gq, const char *tq) {
        const QuadRecord R(qa >> 6, qu2 >> 6, qu >> 6, qb >> 6);
        const int na = qa >> 6, nu2 = qu2 >> 6, nu = qu >> 6, nb = qb >> 6;
        int t0 = 0;
        for (; t0 + U <= na; t0 += U) chunk_a(std::integral_constant<int, U>{}, gq, tq, t0);
        for (; t0 < na; t0++) chunk_a(std::integral_constant<int, 1>{}, gq, tq, t0);
        t0 = 0;
        for (; t0 + 2 <= nu2; t0 += 2) chunk_u2(std::integral_constant<int, 2>{}, gq, tq, R.offU2, t0);
        for (; t0 < nu2; t0++) chunk_u2(std::integral_constant<int, 1>{}, gq, tq, R.offU2, t0);
        t0 = 0;
        for (; t0 + U <= nu; t0 += U) chunk_u(std::integral_constant<int, U>{}, gq, tq, R.offU, t0);
        for (; t0 < nu; t0++) chunk_u(std::integral_constant<int, 1>{}, gq, tq, R.offU, t0);
        t0 = 0;
        for (; t0 + U <= nb; t0 += U) chunk_b(std::integral_constant<int, U>{}, gq, tq, R.offB, t0);
        for (; t0 < nb; t0++) chunk_b(std::integral_constant<int, 1>{}, gq, tq, R.offB, t0);
    };

    // the window is rows py-35..py+35 at columns px-35..px+35 of a linearly addressed plane
    if constexpr (PACK) {
        // quad sweep: the 19 aligned quads per row that cover the window must lie in their rows
        const int x0 = px - radius, y0 = py - radius;
        const int xph = x0 & 3, yph = y0 & 3;        // one table per alignment of the window in the plane
        if (qtab && x0 - xph >= 0 && x0 - xph + qrow4 <= W && px + radius < Wl && y0 >= 0 && py + radius < H) {
            const char *gq = gplane + gradp_quad_word(y0 - yph, (x0 - xph) >> 2, W4) * 2;
            const char *tq = reinterpret_cast<const char *>(qtab) + (size_t)(yph * 4 + xph) * 64u * (size_t)QuadRecord(qa >> 6, qu2 >> 6, qu >> 6, qb >> 6).size;
            if constexpr (NA + NU2 + NU + NB > 0) quad_static(gq, tq);
            else quad_loops(gq, tq);
        } else {
            sweep(std::true_type{});
        }
    } else {
        const bool inside = origin >= 0 && origin + 2L * radius * Wl + 2 * radius < npix;
        if (inside) sweep(std::false_type{});
        else sweep(std::true_type{});
    }
    // Orientation corrections (none for the reference's geometry): the few pixels whose weight in the
    // orientation sum is not "once per union pixel" -- union pixels in the column the 2 * radius loop bound
    // (:70) leaves out count -1, pixels of the orientation mask outside the union +1 -- one lane each.
    if constexpr (SPEC && MODE != DESC_GIVEN_ORI) {
        if (lane < nspec) {
            const uint2 e = spec[lane];
            const uint32_t g = gather(std::true_type{}, e.x, true);
            uint32_t a, v;
            if constexpr (PACK) { v = g & 0x1ffu; a = ((g >> 9) & 3u) * 2u + ((g >> 15) & 1u); }
            else { a = (g & 0xffffu) - 1u; v = a < 8u ? (g >> 16) : 0u; a &= 7u; }
            atomicAdd(&corr[a], e.y ? 0u - v : v);
        }
    }
    __syncthreads();

    // column sums -> fin[bin], bin = circle*8 + a (72..79: overlap)
#ifndef DESC_EXP_NOREDUCE
    {
        const int cg = lane & 7, rsub = lane >> 3;
#pragma unroll
        for (int i = 0; i < ROWS / 8; i++) {
            const int row = i * 8 + rsub;
            const uint4 q = *reinterpret_cast<const uint4 *>(&hist[row][cg * 4]);
            if constexpr (PACK) {
                // 16-bit halves: 2 columns x K pixels x 360 stays below 2^16 (K = pixels of one circle
                // that fall to one column: 27 for the pixel table, 47 for the quad table of the
                // reference's masks; checked on the host); widen after that
                const uint32_t p01 = pk_add(q.x, q.y), p23 = pk_add(q.z, q.w);
                uint32_t lo = (p01 & 0xffffu) + (p23 & 0xffffu), hi = (p01 >> 16) + (p23 >> 16);
                lo += dpp_mov<0x111>(lo); hi += dpp_mov<0x111>(hi);    // row_shr:1
                lo += dpp_mov<0x112>(lo); hi += dpp_mov<0x112>(hi);    // row_shr:2
                lo += dpp_mov<0x114>(lo); hi += dpp_mov<0x114>(hi);    // row_shr:4
                if (cg == 7) *reinterpret_cast<uint2 *>(&fin[row * 2]) = make_uint2(lo, hi);
            } else {
                uint32_t s = q.x + q.y + q.z + q.w;
                s += dpp_mov<0x111>(s);
                s += dpp_mov<0x112>(s);
                s += dpp_mov<0x114>(s);
                if (cg == 7) fin[row] = s;
            }
        }
    }
#endif
    __syncthreads();

    // main orientation (:110-126): first bin 1..8 with the strictly largest value, 0 if none > 0
    uint32_t m;
    if (MODE == DESC_GIVEN_ORI) {
        m = ori_in[slot];
    } else {
        const int a = lane & 7;
        uint32_t v = 0u - fin[72 + a];
#pragma unroll
        for (int n = 0; n < 9; n++) v += fin[n * 8 + a];
        if constexpr (SPEC) v += corr[a];
        // (int)v > 0 is the reference's test; the key orders by value, then by smaller a
        uint32_t key = ((int)v > 0) ? ((v << 3) | (uint32_t)(7 - a)) : 0u;
        // max over the 8 lanes of a group: lanes ^1 and ^2 by quad permutes; after those the two
        // quads of a group are uniform, so the mirror of the 8 lanes supplies the other quad
        key = max(key, dpp_mov<0xB1>(key));              // quad_perm:[1,0,3,2]
        key = max(key, dpp_mov<0x4E>(key));              // quad_perm:[2,3,0,1]
        key = max(key, dpp_mov<0x141>(key));             // row_half_mirror
        m = key ? 8u - (key & 7u) : 0u;
        if (ori_out && lane == 0) ori_out[slot] = (uint8_t)m;
    }
    // rotation (:346-393): out[c][j] = Hin[c<8 ? (c+m)&7 : 8][(j+m)&7]
    if (MODE != DESC_ORI_ONLY) {
        const int c = lane >> 3, j = lane & 7;       // circles 0..7
        out[lane] = fin[((c + m) & 7) * 8 + ((j + m) & 7)];
        if (lane < 8) out[64 + lane] = fin[64 + ((lane + m) & 7)];
    }
    if (LOOP) __syncthreads();                       // hist and fin are reused by the next slot
    }
}

}  // namespace sift
