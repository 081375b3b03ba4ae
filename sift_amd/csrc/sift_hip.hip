// sift_hip.hip -- C ABI (include/sift_hip.h) over the gfx950 kernels.
//
// Host side of the drop-in boundary: context = device + stream + persistent
// scratch; the four reference hardware functions (HW_SIFT.hpp:11-15,
// DetectKeypoints.hpp:109-113, SIFT_Descriptor.hpp:12-14, :59-62) as host- and
// device-buffer entry points.  No CPU fallback exists: without a HIP device
// every entry point fails with SIFT_HIP_ENODEV / SIFT_HIP_EHIP.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <optional>
#include <vector>
#include <thread>
#include <mutex>
#include <atomic>
#include <algorithm>
#include <cmath>

#include "../../include/sift_hip.h"
#include "blur_kernel.h"
#include "blur_mfma_kernel.h"
#include "detect_kernel.h"
#include "compact_kernel.h"
#include "describe_kernel.h"
#include "select_kernel.h"
#include "ingest_kernel.h"
#include "match_kernel.h"

static_assert(sizeof(sift_keypoint) == 8 && sizeof(sift::kp_t) == 8, "Keypoint_t is 8 bytes");
static_assert(sizeof(sift_grad) == 4, "Grad_t is 4 bytes");
static_assert(sizeof(sift_dmatch) == 8 && sizeof(sift::dmatch_t) == 8, "DMatch_t is 8 bytes");

namespace {

thread_local std::string g_err;

int fail(int code, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                 \
    do {                                                              \
        hipError_t e_ = (expr);                                       \
        if (e_ != hipSuccess) return fail(SIFT_HIP_EHIP, #expr, e_); \
    } while (0)

// Everything the library creates on a device goes through these three, which count: a caller (and
// tests/test_parity_gpu.py) can check through sift_hip_debug_created that a steady-state call creates nothing.
std::atomic<long> g_created[3];           // streams, events, device allocations since the library was loaded
hipError_t make_stream(hipStream_t *s) { g_created[0]++; return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
hipError_t make_event(hipEvent_t *e, unsigned flags) { g_created[1]++; return hipEventCreateWithFlags(e, flags); }
template <class T> hipError_t dev_alloc(T **p, size_t bytes) { g_created[2]++; return hipMalloc((void **)p, bytes); }

enum { ST_BLUR = 0, ST_DETECT, ST_CULL, ST_EMIT, ST_DESCRIBE };
const char *const kStageNames[SIFT_HIP_NSTAGES] = {"k_blur", "k_detect", "k_cull", "k_emit", "k_describe"};

// Descriptor geometry.  0 = Cam0, the one the reference's code uses (window 71, radius 35, 25x25 circle
// mask, XStart/YStart, MaskAll = the union of the nine circles).  1 = Cam1: the reference ships only its
// tables (SIFTParameters.hpp:46-55, circle_mask.hpp:12-13, :43-71, mask_all.hpp:161-241); its semantics
// are defined as the Cam0 loops with those constants substituted, so the orientation window
// is 81 rows x 80 columns (2 * radius, SIFT_Descriptor.cpp:70) of MaskAll_81, which is the union of the
// nine Cam1 circles plus 8 pixels.
struct Geometry {
    int window, radius, msize;
    const unsigned char *xs, *ys, *lo, *hi;
    int nextra;
    const unsigned char (*extra)[2];      // {row, column} of orientation-mask pixels outside the union
};
const Geometry kGeometry[2] = {
    {71, 35, 25, SIFT_XSTART, SIFT_YSTART, SIFT_CIRCLE_LO, SIFT_CIRCLE_HI, 0, nullptr},
    {81, 40, 29, SIFT_XSTART_CAM1, SIFT_YSTART_CAM1, SIFT_CIRCLE_CAM1_LO, SIFT_CIRCLE_CAM1_HI, 8, SIFT_MASKALL81_EXTRA},
};
constexpr int MAXWIN = 81;

// circle ids of every window pixel: c1 = first circle, c2 = second (10 = none); false if a pixel lies in three
bool circle_ids(const Geometry &g, int (&c1)[MAXWIN][MAXWIN], int (&c2)[MAXWIN][MAXWIN])
{
    for (int y = 0; y < MAXWIN; y++)
        for (int x = 0; x < MAXWIN; x++) c1[y][x] = c2[y][x] = 10;
    for (int n = 0; n < 9; n++)
        for (int i = 0; i < g.msize; i++)
            for (int j = g.lo[i]; j < g.hi[i]; j++) {
                const int y = g.ys[n] + i, x = g.xs[n] + j;
                if (c1[y][x] == 10) c1[y][x] = n;
                else if (c2[y][x] == 10) c2[y][x] = n;
                else return false;
            }
    return true;
}

// Union-pixel table of the descriptor window for frames `W` wide (see describe_kernel.h): two-circle
// pixels first (n2, padded to n2pad), then the rest.
// kmax = the most pixels of one circle (or of the overlap set) that fall to one lane.
struct WindowTable { std::vector<uint2> e; int n2 = 0, n2pad = 0, kmax = 0; };
WindowTable build_window_table(int W, const Geometry &g)
{
    int c1[MAXWIN][MAXWIN], c2[MAXWIN][MAXWIN];
    if (!circle_ids(g, c1, c2)) return WindowTable{};   // would break Hist = sum Hin - Ov; checked in create()
    WindowTable t;
    std::vector<uint2> one;
    for (int y = 0; y < g.window; y++)
        for (int x = 0; x < g.window; x++) {
            const uint32_t off2 = (uint32_t)(y * W + x) * 2u;
            if (c2[y][x] != 10)
                t.e.push_back(make_uint2(off2, (uint32_t)c1[y][x] << 9 | (uint32_t)c2[y][x] << 25));
            else if (c1[y][x] != 10)
                one.push_back(make_uint2(off2, (uint32_t)c1[y][x] << 9));
        }
    t.n2 = (int)t.e.size();
    while (t.e.size() % 64) t.e.push_back(make_uint2(0, 0));   // never read: the loops stop at n2
    t.n2pad = (int)t.e.size();
    t.e.insert(t.e.end(), one.begin(), one.end());
    // lanes l and l+32 share LDS column l: count per (column, circle | overlap)
    int k[32][10];
    memset(k, 0, sizeof k);
    for (size_t i = 0; i < t.e.size(); i++) {
        if ((int)i >= t.n2 && (int)i < t.n2pad) continue;
        const int col = (int)(i % 32);
        k[col][(t.e[i].y >> 9) & 15]++;
        if ((int)i < t.n2) { k[col][(t.e[i].y >> 25) & 15]++; k[col][9]++; }
    }
    for (int c = 0; c < 32; c++)
        for (int n = 0; n < 10; n++) t.kmax = std::max(t.kmax, k[c][n]);
    return t;
}

// Pixels whose weight in the orientation sum differs from "once per union pixel" (Hist = sum Hin - Ov):
// union pixels in columns the orientation loop does not reach (x >= 2 * radius, SIFT_Descriptor.cpp:70) count
// -1, orientation-mask pixels outside the union +1.  None for Cam0 (column 70 of MaskAll is empty); 22 + 8
// for Cam1.  Entry: x = byte offset in a 2-byte plane as in the window table, y = 1 to subtract.
std::vector<uint2> build_special_table(int W, const Geometry &g)
{
    int c1[MAXWIN][MAXWIN], c2[MAXWIN][MAXWIN];
    std::vector<uint2> t;
    if (!circle_ids(g, c1, c2)) return t;
    for (int y = 0; y < g.window; y++)
        for (int x = 2 * g.radius; x < g.window; x++)
            if (c1[y][x] != 10) t.push_back(make_uint2((uint32_t)(y * W + x) * 2u, 1u));
    for (int i = 0; i < g.nextra; i++)
        if (g.extra[i][1] < 2 * g.radius) t.push_back(make_uint2((uint32_t)(g.extra[i][0] * W + g.extra[i][1]) * 2u, 0u));
    return t;
}

// Quad table of the describe kernel's fast path (describe_kernel.h): for each alignment of the
// window in the plane (variant = (row of the window's first row in its group of four) * 4 + column
// phase), the aligned 4-pixel quads of the window's rows that hold a union pixel; quads with a
// two-circle pixel first (part A: qa entries after padding, 8 bytes each), then quads that lie in one
// circle with all four pixels: pairs of them in rows 2j, 2j + 1 of a group of four rows -- 16 neighbouring
// bytes of the row-interleaved plane, one gather for eight pixels -- (part U2: qu2 entries) and single ones
// (part U: qu entries), 2 bytes each: (quad * 4 + row in the group) | group of rows << 7 | circle id << 12;
// then the rest (part B: qb entries, 4 bytes each: group of rows | 4 x circle id << 9 | (quad * 4 + row
// in the group) << 25).
// All parts are ordered by address in the row-interleaved plane (detect_kernel.h), so consecutive
// lanes read consecutive pieces.  The entries of a window alignment are stored as 64 records, one per lane
// (sift::QuadRecord), which a lane fetches with a few 16-byte loads.
struct QuadTable { std::vector<uint8_t> bytes; int qa = 0, qu2 = 0, qu = 0, qb = 0, kmax = 0; };
QuadTable build_quad_table(int W, const Geometry &g)
{
    int c1[MAXWIN][MAXWIN], c2[MAXWIN][MAXWIN];
    if (!circle_ids(g, c1, c2)) return QuadTable{};
    const int WS = g.window, QR = (WS + 2) / 4 + 1;     // aligned quads that cover a window row at any phase: 19 / 21
    struct Quad { int y, q, a[4], b[4]; };          // y = window row + row phase: row from the base group's first row
    std::vector<Quad> A[16], U2[16], U[16], B[16];
    size_t na = 0, nu = SIZE_MAX, nu2 = SIZE_MAX;
    for (int v = 0; v < 16; v++) {
        const int ph = v & 3, yph = v >> 2;
        for (int y = 0; y < WS; y++)
            for (int q = 0; q < QR; q++) {
                Quad e;
                e.y = y + yph; e.q = q;
                bool any = false, two = false, uniform = true;
                for (int p = 0; p < 4; p++) {
                    const int x = q * 4 + p - ph;                  // window column of this pixel
                    e.a[p] = e.b[p] = 10;
                    if (x >= 0 && x < WS) { e.a[p] = c1[y][x]; e.b[p] = c2[y][x]; }
                    any |= e.a[p] != 10;
                    two |= e.b[p] != 10;
                    uniform &= e.a[p] != 10 && e.a[p] == e.a[0];
                }
                if (two) A[v].push_back(e);
                else if (uniform) U[v].push_back(e);
                else if (any) B[v].push_back(e);
            }
        na = std::max(na, A[v].size());
    }
    // bytes from the base: (group of rows) * W * 8 + quad * 32 + (row in the group) * 8
    auto low = [](const Quad &e) { return (e.q * 4 + (e.y & 3)) * 8; };
    auto before = [&](const Quad &l, const Quad &r) { return (l.y >> 2) != (r.y >> 2) ? (l.y >> 2) < (r.y >> 2) : low(l) < low(r); };
    QuadTable t;
    t.qa = (int)((na + 63) / 64 * 64);
    // one-circle quads: the upper quad of a pair in rows (2j, 2j + 1) of a group goes to U2, its partner is dropped
    // from U (it travels with the upper one); whole steps only: what is left of the pairs is split up again and what
    // is left of the single quads goes to part B
    for (int v = 0; v < 16; v++) {
        std::sort(U[v].begin(), U[v].end(), before);
        std::vector<Quad> single;
        std::vector<char> taken(U[v].size(), 0);
        for (size_t i = 0; i < U[v].size(); i++) {
            if (taken[i]) continue;
            const Quad &e = U[v][i];
            size_t j = U[v].size();
            if ((e.y & 1) == 0)
                for (size_t k = i + 1; k < U[v].size(); k++)
                    if (!taken[k] && U[v][k].y == e.y + 1 && U[v][k].q == e.q && U[v][k].a[0] == e.a[0]) { j = k; break; }
            if (j < U[v].size()) { taken[j] = 1; U2[v].push_back(e); }
            else single.push_back(e);
        }
        U[v] = single;
        nu2 = std::min(nu2, U2[v].size());
    }
    t.qu2 = (int)(nu2 / 64 * 64);
    for (int v = 0; v < 16; v++) {
        for (size_t i = (size_t)t.qu2; i < U2[v].size(); i++) {       // pairs beyond the whole steps: two single quads again
            Quad lower = U2[v][i];
            lower.y += 1;
            U[v].push_back(U2[v][i]);
            U[v].push_back(lower);
        }
        U2[v].resize((size_t)t.qu2);
        std::sort(U[v].begin(), U[v].end(), before);
        nu = std::min(nu, U[v].size());
    }
    t.qu = (int)(nu / 64 * 64);
    size_t nb = 0;
    for (int v = 0; v < 16; v++) {
        B[v].insert(B[v].end(), U[v].begin() + t.qu, U[v].end());
        U[v].resize((size_t)t.qu);
        nb = std::max(nb, B[v].size());
    }
    t.qb = (int)((nb + 63) / 64 * 64);
    const Quad nop = {0, 0, {10, 10, 10, 10}, {10, 10, 10, 10}};   // offset 0, every pixel -> spare rows
    // a phase = 64 records, one per lane, interleaved by 16-byte piece (sift::QuadRecord): entry i of a part belongs to
    // lane i % 64, step i / 64
    const sift::QuadRecord R(t.qa / 64, t.qu2 / 64, t.qu / 64, t.qb / 64);
    t.bytes.assign((size_t)16 * 64 * (size_t)R.size, 0);
    uint8_t *phase = nullptr;
    auto put = [&](int part_off, int entry_bytes, size_t i, uint32_t w0, uint32_t w1 = 0) {
        uint8_t *d = phase + sift::QuadRecord::at((uint32_t)(i % 64), (uint32_t)part_off + (uint32_t)(i / 64) * (uint32_t)entry_bytes);   // an entry never straddles a 16-byte piece
        for (int b = 0; b < entry_bytes; b++) d[b] = (uint8_t)((b < 4 ? w0 >> (8 * b) : w1 >> (8 * (b - 4))));
    };
    for (int ph = 0; ph < 16; ph++) {
        std::sort(A[ph].begin(), A[ph].end(), before);
        std::sort(B[ph].begin(), B[ph].end(), before);
        A[ph].resize((size_t)t.qa, nop);
        B[ph].resize((size_t)t.qb, nop);
        phase = t.bytes.data() + (size_t)ph * 64 * (size_t)R.size;
        // pixels per (LDS column, circle | overlap) of this phase; entry i falls to column i % 32
        int k[32][10];
        memset(k, 0, sizeof k);
        for (size_t i = 0; i < A[ph].size(); i++) {
            const Quad &e = A[ph][i];
            uint32_t lo = (uint32_t)((e.y >> 2) * W * 8 + low(e)), hi = 0;
            for (int p = 0; p < 4; p++) {
                if (e.a[p] < 9) k[i % 32][e.a[p]]++;
                if (e.b[p] < 9) { k[i % 32][e.b[p]]++; k[i % 32][9]++; }
                hi |= (uint32_t)(e.a[p] | e.b[p] << 4) << (8 * p);
                if (e.b[p] == 10) lo |= 1u << (24 + p);            // notov: the third add goes to the spare rows
            }
            put(0, 8, i, lo, hi);
        }
        for (size_t i = 0; i < U2[ph].size(); i++) {
            const Quad &e = U2[ph][i];
            k[i % 32][e.a[0]] += 8;
            put(R.offU2, 2, i, (uint32_t)(e.q * 4 + (e.y & 3)) | (uint32_t)(e.y >> 2) << 7 | (uint32_t)e.a[0] << 12);
        }
        for (size_t i = 0; i < U[ph].size(); i++) {
            const Quad &e = U[ph][i];
            k[i % 32][e.a[0]] += 4;
            put(R.offU, 2, i, (uint32_t)(e.q * 4 + (e.y & 3)) | (uint32_t)(e.y >> 2) << 7 | (uint32_t)e.a[0] << 12);
        }
        for (size_t i = 0; i < B[ph].size(); i++) {
            const Quad &e = B[ph][i];
            uint32_t w = (uint32_t)(e.y >> 2) | (uint32_t)(e.q * 4 + (e.y & 3)) << 25;
            for (int p = 0; p < 4; p++) {
                if (e.a[p] < 9) k[i % 32][e.a[p]]++;
                w |= (uint32_t)e.a[p] << (9 + 4 * p);
            }
            put(R.offB, 4, i, w);
        }
        for (int c = 0; c < 32; c++)
            for (int n = 0; n < 10; n++) t.kmax = std::max(t.kmax, k[c][n]);
    }
    return t;
}

}  // namespace

struct sift_hip_ctx {
    int device = 0, W = 0, H = 0, S = 0, max_kp = 0, max_batch = 0;   // W: internal row pitch, a multiple of 4
    int Wl = 0;                           // the caller's (logical) frame width; columns Wl .. W-1 of the internal planes are padding
    size_t npix = 0;                      // W * H
    hipStream_t stream = nullptr;
    // persistent scratch, max_batch frames each
    uint8_t *d_img = nullptr;             // staging for host-buffer entry points
    uint4 *d_pyr = nullptr;               // {src, G1, G2, G3} dwords per group of 4 pixels
    unsigned long long *d_cand = nullptr, *d_surv = nullptr;   // [H][S] bit masks
    uint32_t *d_key = nullptr;            // (E<<8)|contrast at candidates
    uint32_t *d_rowcount = nullptr;       // [H]
    uint32_t *d_rowbase = nullptr;        // [H + 1] exclusive prefix of d_rowcount (k_rowprefix)
    uint16_t *d_gradp = nullptr;          // internal gradient plane (gradp_pack, detect_kernel.h)
    uint32_t *d_gradfull = nullptr;       // Grad_t staging (1 frame, lazily allocated)
    sift::kp_t *d_kps = nullptr;          // staging for host-buffer entry points
    uint32_t *d_desc = nullptr;
    uint32_t *d_total = nullptr;          // staging for host-buffer entry points
    uint32_t *d_total_ctx = nullptr;      // k_emit's uncapped count per frame, read by the fused describe kernel
    uint8_t *d_ori = nullptr;
    sift::kp_t *d_final = nullptr;        // staging of sift_hip_select_top (1 frame, lazily allocated)
    uint32_t *d_findex = nullptr, *d_nvalid = nullptr;
    sift::kp_t *d_tfinal = nullptr;       // sift_hip_batch_top: selected keypoints of the chunks in flight (lazily allocated)
    uint32_t *d_tindex = nullptr, *d_tdesc = nullptr, *d_tnvalid = nullptr;
    size_t top_cap = 0;
    uint8_t *d_match = nullptr;           // staging of sift_hip_match (lazily allocated)
    size_t match_cap = 0;
    uint8_t *d_bgr = nullptr;             // staging of sift_hip_gray_from_bgr (lazily allocated)
    size_t bgr_cap = 0;
    uint8_t *d_oct = nullptr;             // staging of sift_hip_next_octave (lazily allocated, one frame)
    // sift_hip_batch[_top]: the two copy streams and the hand-over events of the chunks in flight; made by the
    // first batch call on this context and kept (ensure_batch_streams)
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t in_done[2] = {}, k_done[2] = {}, out_done[2] = {};
    uint2 *d_tab = nullptr;
    uint32_t *d_qtab = nullptr;           // quad table (build_quad_table)
    int qu2 = 0;                          // its part U2 (pairs of one-circle quads)
    uint2 *d_spec = nullptr;              // orientation corrections of the geometry (build_special_table)
    int nspec = 0, radius = 35, qrow4 = 76; // window radius and bytes^-1: 4 x aligned quads per window row (describe kernel)
    uint32_t *d_gate = nullptr;           // k_grad_pairs' flag: which of the two describe kernels of a stand-alone call runs
    int ntab = 0, n2 = 0, n2pad = 0, qa = 0, qu = 0, qb = 0;
    bool quad_sweep = true;               // false: every window through the pixel sweep (sift_hip_set_quad_sweep)
    int blur_rows = 0;
    bool blur_mfma = true;                // k_blur_mfma (matrix cores) instead of k_blur (dot4 VALU); sift_hip_set_blur_mfma
    // graph of the last fused call (small batches: launch gaps are a third of a single frame's time)
    struct GraphKey { const void *imgs = nullptr, *kps = nullptr, *desc = nullptr, *total = nullptr; int nframes = 0, th = 0, brows = 0, drows = 0; } gkey;
    hipGraphExec_t gexec = nullptr;
    int gseen = 0;                        // consecutive calls of the shape in gkey
    int graph_max_frames = 4;             // calls with more frames are launched kernel by kernel (0: never use a graph)
    int detect_rows = 0;                  // rows per k_detect strip (0 = heuristic)
    int detect_queue = sift::DETECT_QUEUE; // LDS queue entries k_detect uses before it drains (test knob)
    int blur_sym_rows = 48;               // strips at least this long run k_blur<false>
    // profiling: a ring of `nslots` calls, one HIP event pair per kernel of a call
    struct ProfSlot { hipEvent_t ev[SIFT_HIP_NSTAGES][2] = {}; bool used[SIFT_HIP_NSTAGES] = {}; };
    std::vector<ProfSlot> prof;
    long ncalls = 0;                      // calls recorded since profiling was switched on
    int cur = -1;                         // slot of the call being enqueued (-1: profiling off)
    int last_frames = 0;
};

namespace {

struct DevGuard {
    int prev = -1;
    explicit DevGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct StageTimer {
    sift_hip_ctx *c; int st; hipStream_t s;
    StageTimer(sift_hip_ctx *c_, int st_, hipStream_t s_) : c(c_), st(st_), s(s_)
    {
        if (c->cur >= 0) { hipEventRecord(c->prof[c->cur].ev[st][0], s); c->prof[c->cur].used[st] = true; }
    }
    ~StageTimer() { if (c->cur >= 0) hipEventRecord(c->prof[c->cur].ev[st][1], s); }
};

// every entry point that enqueues kernels opens one profiling record
void begin_call(sift_hip_ctx *c)
{
    if (c->prof.empty()) { c->cur = -1; return; }
    c->cur = (int)(c->ncalls % (long)c->prof.size());
    c->ncalls++;
    memset(c->prof[c->cur].used, 0, sizeof c->prof[c->cur].used);
}

int pick_blur_rows(const sift_hip_ctx *c, int nframes)
{
    if (c->blur_rows > 0) return c->blur_rows;
    // Strip length by a cost model fitted to measurements (tools/blur_rows_sweep.py): a wave
    // runs rows + 18 warm-up iterations; k_blur<false> (long strips) pays 0.66 of k_blur<true>'s
    // cost per iteration but all of the warm-up, k_blur<true> about half of it; the chip takes 1024
    // waves at one per SIMD, a second wave per SIMD costs half as much again (measured), and what
    // does not fit into 2048 waves runs after them.
    const double waves_per_strip = (double)(((c->W / 4) + 63) / 64);
    double best = 1e300;
    int best_rows = c->H;
    for (int k = 1; k <= (c->H + 5) / 6; k++) {
        const int rows = (c->H + k - 1) / k;
        if (rows < 6) break;
        const double strips = (double)((c->H + rows - 1) / rows);
        const double waves = waves_per_strip * strips * nframes;
        // long strips run a little slower per row (measured with tools/blur_rows_big.py: 1080-row
        // strips 4 % behind 360-row ones although they have the smallest warm-up share)
        const double per_wave = rows >= c->blur_sym_rows ? 0.66 * (rows + 18) * (1.0 + rows / 10000.0) : rows + 9.0;
        const double full = std::floor(waves / 2048.0), rem = waves - 2048.0 * full;
        // more than four rounds of 2048 waves: the rounds blur into a steady stream
        const double cost = per_wave * (waves > 4.0 * 2048.0 ? 1.5 * waves / 2048.0
                                                             : 1.5 * full + (rem <= 0.0 ? 0.0 : rem <= 1024.0 ? 1.0 : 1.5));
        if (cost < best * 0.999) { best = cost; best_rows = rows; }
    }
    return best_rows;
}

// Strip length of k_blur_mfma: a wave takes 16 (strip, 8-pixel block) units and runs rows / 2 + 10 steps of
// 18 MFMAs; the chip holds 2048 such waves (two per SIMD, one of each role).  Few strips keep the 10 warm-up
// steps small, many fill the chip when the batch is small: the cheapest of all strip counts by that model.
int pick_blur_mfma_rows(const sift_hip_ctx *c, int nframes)
{
    if (c->blur_rows > 0) return std::min(c->blur_rows, c->H);
    const int NB = sift::blur_mfma_blocks(c->W);
    double best = 1e300;
    int best_rows = c->H;
    for (int k = 1; k <= c->H; k++) {
        int rows = (c->H + k - 1) / k;
        if (k > 1) rows = (rows + 1) & ~1;           // pairs of rows: an odd strip would compute one row twice
        if (rows < 8) break;
        const int strips = (c->H + rows - 1) / rows;
        const double waves = (double)((NB * strips + 15) / 16) * nframes;
        const double rounds = waves > 4096.0 ? waves / 1024.0 : std::ceil(waves / 1024.0);
        const double cost = ((rows + 1) / 2 + 10) * rounds;
        if (cost < best * 0.999) { best = cost; best_rows = rows; }
    }
    return best_rows;
}

// Frames of the caller's width into the internal pitch (columns Wl .. W-1 zero), or the caller's pointer itself
// when the width is a multiple of 4 already.  `dst`: nframes * npix bytes of this context.
const uint8_t *pad_frames(sift_hip_ctx *c, const uint8_t *d_imgs, int nframes, uint8_t *dst, hipStream_t s)
{
    if (c->Wl == c->W) return d_imgs;
    dim3 grid((c->W / 4 + 255) / 256, c->H, nframes);
    hipLaunchKernelGGL(sift::k_pad_rows, grid, dim3(256), 0, s, d_imgs, dst, c->Wl, c->W, c->H);
    return dst;
}

// host frame -> internal staging (rows of W bytes; the padding columns were zeroed at create and are never written)
hipError_t upload_frame(sift_hip_ctx *c, uint8_t *dst, const uint8_t *img, hipStream_t s)
{
    if (c->Wl == c->W) return hipMemcpyAsync(dst, img, c->npix, hipMemcpyHostToDevice, s);
    return hipMemcpy2DAsync(dst, (size_t)c->W, img, (size_t)c->Wl, (size_t)c->Wl, (size_t)c->H, hipMemcpyHostToDevice, s);
}

// blur -> detect -> cull -> emit on device-resident frames in the internal pitch (pad_frames / upload_frame)
int run_detect(sift_hip_ctx *c, const uint8_t *d_imgs, int nframes, uint32_t *d_gradfull, sift::kp_t *d_kps,
               uint32_t *d_total, int th, hipStream_t s)
{
    const int W = c->W, H = c->H, S = c->S;
    c->last_frames = nframes;
    // k_blur_mfma addresses a frame's pyr plane (4 bytes per pixel) through one 32-bit buffer descriptor, whose
    // size has to stay at or below the offset the kernel uses for "dropped" (0xfffffff0)
    if (c->blur_mfma && c->npix * 4 <= 0xfffffff0ull) {
        StageTimer t(c, ST_BLUR, s);
        const int rows = pick_blur_mfma_rows(c, nframes);
        const int nstrips = (H + rows - 1) / rows, NB = sift::blur_mfma_blocks(W);
        const int waves = (NB * nstrips + 15) / 16;
        // workgroups of two unit groups (two waves of each role; two workgroups fill a CU's register file):
        // measured against one and four groups per workgroup, 1.42 - 1.49 ms against 1.52 and 1.54 - 1.61 per 256 frames
        // batches of 64 frames and more: every XCD takes whole frames (k_blur_mfma, xcd_frames); the grid's frame
        // dimension is rounded up to a multiple of 8 and the workgroups of the frames that do not exist return at once
        const int xcd_frames = nframes >= 64 ? nframes : 0;
        const int gy = xcd_frames ? (nframes + 7) / 8 * 8 : nframes;
        hipLaunchKernelGGL(sift::k_blur_mfma, dim3((waves + BLUR_MFMA_WG_GROUPS - 1) / BLUR_MFMA_WG_GROUPS, gy), dim3(128 * BLUR_MFMA_WG_GROUPS), 0, s, d_imgs, c->d_pyr, W, H, rows, nstrips, c->Wl, xcd_frames);
    } else {
        StageTimer t(c, ST_BLUR, s);
        const int rows = pick_blur_rows(c, nframes);
        // 64-thread blocks when a row has few dword columns, so small frames still spread over CUs
        const int bs = (W / 4) >= 256 ? 256 : 64;
        dim3 grid(((W / 4) + bs - 1) / bs, (H + rows - 1) / rows, nframes);
        // long strips: the 18 warm-up rows are a small share, every slot is updated in every
        // iteration with the symmetric scheme; short strips skip the slots outside the strip
        if (rows >= c->blur_sym_rows)
            hipLaunchKernelGGL(sift::k_blur<false>, grid, dim3(bs), 0, s, d_imgs, c->d_pyr, W, H, rows, c->Wl);
        else
            hipLaunchKernelGGL(sift::k_blur<true>, grid, dim3(bs), 0, s, d_imgs, c->d_pyr, W, H, rows, c->Wl);
    }
    {
        StageTimer t(c, ST_DETECT, s);
        // strips of up to DETECT_ROWS rows; shorter ones (2 extra rows of work per strip) when a
        // small batch would otherwise leave most of the chip idle
        const int spans = (W + 255) / 256;
        int drows = c->detect_rows;
        if (drows <= 0) {
            const long strips = (768L + (long)spans * nframes - 1) / ((long)spans * nframes);   // measured: ~3 waves per CU suffice
            drows = (int)((H + strips - 1) / strips);
            // large batches: strips of 12 rows (measured with tools/detect_sweep.py at 256 frames: 0.920 ms against 0.926 for
            // 16, 0.942 for 24, 0.959 for 32 rows -- two more rows read per strip cost less than the shorter waves gain)
            drows = std::min(drows, 12);
        }
        drows = std::max(4, std::min(drows, (int)sift::DETECT_ROWS));
        dim3 grid(spans, (H + drows - 1) / drows, nframes);
        if (d_gradfull)
            hipLaunchKernelGGL(sift::k_detect<true>, grid, dim3(64), 0, s, c->d_pyr, c->d_cand, c->d_key,
                               c->d_rowcount, c->d_gradp, d_gradfull, W, H, S, th, drows, c->detect_queue, c->Wl);
        else
            hipLaunchKernelGGL(sift::k_detect<false>, grid, dim3(64), 0, s, c->d_pyr, c->d_cand, c->d_key,
                               c->d_rowcount, c->d_gradp, (uint32_t *)nullptr, W, H, S, th, drows, c->detect_queue, c->Wl);
    }
    {
        StageTimer t(c, ST_CULL, s);
        // one lane per 64-pixel word of the candidate mask
        hipLaunchKernelGGL(sift::k_cull, dim3((H * S + 255) / 256, nframes), dim3(256), 0, s, c->d_cand, c->d_key,
                           c->d_surv, c->d_rowcount, W, H, S, c->Wl);
    }
    {
        StageTimer t(c, ST_EMIT, s);
        const int R = S <= 32 ? 64 / S : 1;            // whole rows per wave
        dim3 grid(((H + R - 1) / R + 1 + 3) / 4, nframes);
        hipLaunchKernelGGL(sift::k_rowprefix, dim3(nframes), dim3(1024), 0, s, c->d_rowcount, c->d_rowbase, H);
        hipLaunchKernelGGL(sift::k_emit, grid, dim3(256), 0, s, c->d_surv, c->d_key, c->d_rowbase, d_kps,
                           c->d_total_ctx, d_total, W, H, S, c->max_kp, c->Wl, R);
    }
    HIP_TRY(hipGetLastError());
    return SIFT_HIP_OK;
}

template <int GFMT, int MODE>
int run_describe(sift_hip_ctx *c, const void *d_grad, int nframes, const sift::kp_t *d_kps, const uint8_t *ori_in,
                 uint8_t *ori_out, uint32_t *d_desc, hipStream_t s, const uint32_t *gate = nullptr, uint32_t gate_want = 0)
{
    std::optional<StageTimer> t;
    if (!gate) t.emplace(c, ST_DESCRIBE, s);         // gated pairs are timed as one by run_describe_planes
    // one wave per slot, or per group of slots when the batch is so large that the launch would
    // exceed ~330k waves (sparse frames: most of them would only write zeros; dense frames lose
    // nothing: measured 0.684 against 0.691 ms per 64 frames with two slots per wave)
    const bool fused = MODE == sift::DESC_FUSED;
    const int gx = std::max(1024, std::min(c->max_kp, (327680 + nframes - 1) / nframes));
    dim3 grid(gx >= c->max_kp ? c->max_kp : (gx & ~7), nframes);   // grouped slots: a multiple of 8 (one share per XCD)
    const uint32_t *tot = fused ? (const uint32_t *)c->d_total_ctx : (const uint32_t *)nullptr;
    // geometries with orientation corrections (Cam1) run the table-driven variants with the correction step
    if (c->nspec) {
        if ((int)grid.x == c->max_kp)
            hipLaunchKernelGGL((sift::k_describe<GFMT, MODE, false, 0, 0, 0, 0, true>), grid, dim3(64), 0, s, d_grad, d_kps, ori_in, ori_out,
                               d_desc, c->d_tab, c->n2, c->n2pad, c->ntab, c->quad_sweep ? c->d_qtab : nullptr, c->qa, c->qu2, c->qu, c->qb, tot, c->W, c->H, c->max_kp, gate, gate_want, c->radius, c->qrow4, c->d_spec, c->nspec, c->Wl);
        else
            hipLaunchKernelGGL((sift::k_describe<GFMT, MODE, true, 0, 0, 0, 0, true>), grid, dim3(64), 0, s, d_grad, d_kps, ori_in, ori_out,
                               d_desc, c->d_tab, c->n2, c->n2pad, c->ntab, c->quad_sweep ? c->d_qtab : nullptr, c->qa, c->qu2, c->qu, c->qb, tot, c->W, c->H, c->max_kp, gate, gate_want, c->radius, c->qrow4, c->d_spec, c->nspec, c->Wl);
        HIP_TRY(hipGetLastError());
        return SIFT_HIP_OK;
    }
    // the reference's masks give 5 + 4 + 1 + 2 steps of 64 entries: straight-line variant
    const bool ref_steps = c->qa == 5 * 64 && c->qu2 == 4 * 64 && c->qu == 64 && c->qb == 2 * 64;
    if (GFMT == sift::GRAD_PACKED16 && (int)grid.x == c->max_kp && ref_steps)
        hipLaunchKernelGGL((sift::k_describe<GFMT, MODE, false, 5, 4, 1, 2>), grid, dim3(64), 0, s, d_grad, d_kps, ori_in, ori_out,
                           d_desc, c->d_tab, c->n2, c->n2pad, c->ntab, c->quad_sweep ? c->d_qtab : nullptr, c->qa, c->qu2, c->qu, c->qb, tot, c->W, c->H, c->max_kp, gate, gate_want, c->radius, c->qrow4, c->d_spec, c->nspec, c->Wl);
    else if ((int)grid.x == c->max_kp)
        hipLaunchKernelGGL((sift::k_describe<GFMT, MODE, false>), grid, dim3(64), 0, s, d_grad, d_kps, ori_in, ori_out,
                           d_desc, c->d_tab, c->n2, c->n2pad, c->ntab, c->quad_sweep ? c->d_qtab : nullptr, c->qa, c->qu2, c->qu, c->qb, tot, c->W, c->H, c->max_kp, gate, gate_want, c->radius, c->qrow4, c->d_spec, c->nspec, c->Wl);
    else if (GFMT == sift::GRAD_PACKED16 && ref_steps)
        hipLaunchKernelGGL((sift::k_describe<GFMT, MODE, true, 5, 4, 1, 2>), grid, dim3(64), 0, s, d_grad, d_kps, ori_in, ori_out,
                           d_desc, c->d_tab, c->n2, c->n2pad, c->ntab, c->quad_sweep ? c->d_qtab : nullptr, c->qa, c->qu2, c->qu, c->qb, tot, c->W, c->H, c->max_kp, gate, gate_want, c->radius, c->qrow4, c->d_spec, c->nspec, c->Wl);
    else
        hipLaunchKernelGGL((sift::k_describe<GFMT, MODE, true>), grid, dim3(64), 0, s, d_grad, d_kps, ori_in, ori_out,
                           d_desc, c->d_tab, c->n2, c->n2pad, c->ntab, c->quad_sweep ? c->d_qtab : nullptr, c->qa, c->qu2, c->qu, c->qb, tot, c->W, c->H, c->max_kp, gate, gate_want, c->radius, c->qrow4, c->d_spec, c->nspec, c->Wl);
    HIP_TRY(hipGetLastError());
    return SIFT_HIP_OK;
}

// Stand-alone orientation / descriptor calls: the caller's Grad_t planes are rewritten into the
// internal plane (a streaming kernel) and swept by the fast kernel; if a magnitude does not fit
// that format the flag set by the conversion lets the Grad_t kernel run instead.  Both kernels are
// enqueued, one of them returns at once.
template <int MODE>
int run_describe_planes(sift_hip_ctx *c, const sift_grad *d_grad, int nframes, const sift::kp_t *d_kps, const uint8_t *ori_in,
                        uint8_t *ori_out, uint32_t *d_desc, hipStream_t s)
{
    StageTimer t(c, ST_DESCRIBE, s);
    HIP_TRY(hipMemsetAsync(c->d_gate, 0, 4, s));
    // a caller's plane whose width is not a multiple of 4 is swept in its own format (its rows are not
    // 16-byte aligned, which the conversion kernel reads them by)
    if (c->Wl != c->W) {
        const size_t n = (size_t)nframes * (size_t)c->Wl * (size_t)c->H;   // the magnitude check k_grad_pairs would have made
        hipLaunchKernelGGL(sift::k_grad_check, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, (const uint32_t *)d_grad, c->d_gate, n);
        HIP_TRY(hipGetLastError());
        return run_describe<sift::GRAD_FULL32, MODE>(c, d_grad, nframes, d_kps, ori_in, ori_out, d_desc, s);
    }
    const dim3 grid((c->W / 4 + 255) / 256, (c->H + 3) / 4, nframes);
    hipLaunchKernelGGL(sift::k_grad_pairs, grid, dim3(256), 0, s, (const uint4 *)d_grad, c->d_gradp, c->d_gate, c->W, c->H);
    HIP_TRY(hipGetLastError());
    if (int r = run_describe<sift::GRAD_PACKED16, MODE>(c, c->d_gradp, nframes, d_kps, ori_in, ori_out, d_desc, s, c->d_gate, 0u)) return r;
    return run_describe<sift::GRAD_FULL32, MODE>(c, d_grad, nframes, d_kps, ori_in, ori_out, d_desc, s, c->d_gate, 1u);
}

// end of a host-buffer orientation / descriptor call: wait, then look at k_grad_pairs' flag
int finish_planes_call(sift_hip_ctx *c, hipStream_t s)
{
    uint32_t flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, c->d_gate, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (flag & 2u)
        return fail(SIFT_HIP_EINVAL, "gradient plane holds a counted magnitude above 32767: the reference's ap_int<16> MagGrad "
                                     "(SIFT_Descriptor.cpp:81) would count it as negative; such planes are not supported");
    return SIFT_HIP_OK;
}

int check_ctx(const sift_hip_ctx *c)
{
    if (!c) return fail(SIFT_HIP_EINVAL, "null context");
    return SIFT_HIP_OK;
}

int ensure_select(sift_hip_ctx *c)
{
    const size_t n = (size_t)std::min(c->max_kp, (int)sift::SEL_MAX_FINAL);
    if (!c->d_final) HIP_TRY(dev_alloc(&c->d_final, n * sizeof(sift::kp_t)));
    if (!c->d_findex) HIP_TRY(dev_alloc(&c->d_findex, n * 4));
    if (!c->d_nvalid) HIP_TRY(dev_alloc(&c->d_nvalid, 4));
    return SIFT_HIP_OK;
}

int ensure_batch_streams(sift_hip_ctx *c)
{
    if (!c->s_in) HIP_TRY(make_stream(&c->s_in));
    if (!c->s_out) HIP_TRY(make_stream(&c->s_out));
    for (int i = 0; i < 2; i++) {
        if (!c->in_done[i]) HIP_TRY(make_event(&c->in_done[i], hipEventDisableTiming));
        if (!c->k_done[i]) HIP_TRY(make_event(&c->k_done[i], hipEventDisableTiming));
        if (!c->out_done[i]) HIP_TRY(make_event(&c->out_done[i], hipEventDisableTiming));
    }
    return SIFT_HIP_OK;
}

int ensure_gradfull(sift_hip_ctx *c)
{
    if (!c->d_gradfull) HIP_TRY(dev_alloc(&c->d_gradfull, c->npix * sizeof(uint32_t)));
    return SIFT_HIP_OK;
}

}  // namespace

static int fused_dev(sift_hip_ctx *c, const uint8_t *d_imgs, bool padded, int nframes, sift_keypoint *d_kps, uint32_t *d_desc,
                     uint32_t *d_total, int contrastTh, void *stream);

extern "C" {

const char *sift_hip_last_error(void) { return g_err.c_str(); }

int sift_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void *sift_hip_host_alloc(unsigned long bytes)
{
    void *p = nullptr;
    if (bytes == 0) bytes = 1;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { fail(SIFT_HIP_EHIP, "hipHostMalloc", e); return nullptr; }
    return p;
}

void sift_hip_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

const char *sift_hip_stage_name(int stage)
{
    return (stage >= 0 && stage < SIFT_HIP_NSTAGES) ? kStageNames[stage] : "";
}

int sift_hip_create(sift_hip_ctx **out, int device, int width, int height, int max_keypoints, int max_batch)
{
    return sift_hip_create_geometry(out, device, width, height, max_keypoints, max_batch, SIFT_HIP_GEOMETRY_CAM0);
}

int sift_hip_create_geometry(sift_hip_ctx **out, int device, int width, int height, int max_keypoints, int max_batch, int geometry)
{
    if (!out) return fail(SIFT_HIP_EINVAL, "out is null");
    *out = nullptr;
    if (geometry != SIFT_HIP_GEOMETRY_CAM0 && geometry != SIFT_HIP_GEOMETRY_CAM1) return fail(SIFT_HIP_EINVAL, "unknown descriptor geometry");
    const Geometry &geo = kGeometry[geometry];
    if (width < 1 || height < 1 || width > 65532 || height > 65535)
        return fail(SIFT_HIP_EINVAL, "width must lie in [1,65532], height in [1,65535]");
    if (max_keypoints < 1 || max_batch < 1) return fail(SIFT_HIP_EINVAL, "max_keypoints and max_batch must be >= 1");
    const int logical_width = width;
    width = (width + 3) & ~3;                        // internal row pitch; the extra columns are zeros = "outside the image"
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SIFT_HIP_ENODEV, "no HIP device");
    if (device < 0 || device >= ndev) return fail(SIFT_HIP_EINVAL, "device index out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SIFT_HIP_ENODEV, "device is not gfx950 (the library carries gfx950 code only)");

    // the pixel tables hold offsets of the reference's linear addressing (pitch = the logical width), the
    // quad table byte offsets in the internal plane (pitch = the padded width)
    const WindowTable wt = build_window_table(logical_width, geo);
    const std::vector<uint2> &tab = wt.e;
    if (tab.empty()) return fail(SIFT_HIP_EINVAL, "circle tables overlap three-fold");
    const QuadTable qt = build_quad_table(width, geo);
    const std::vector<uint2> spec = build_special_table(logical_width, geo);
    if (spec.size() > 64) return fail(SIFT_HIP_EINVAL, "more than 64 orientation corrections");
    // 16-bit bins of the describe kernel: a column's bin takes kmax pixels, and the reduction adds
    // 2 columns before widening (describe_kernel.h)
    if (2L * std::max(wt.kmax, qt.kmax) * sift::DESC_MAX_MAG_PACKED >= 65536)
        return fail(SIFT_HIP_EINVAL, "descriptor window too large for 16-bit lane bins");

    DevGuard g(device);
    sift_hip_ctx *c = new sift_hip_ctx;
    c->device = device; c->W = width; c->Wl = logical_width; c->H = height; c->S = (width + 63) / 64;
    c->max_kp = max_keypoints; c->max_batch = max_batch;
    c->npix = (size_t)width * height;
    const size_t B = (size_t)max_batch, np = c->npix, HS = (size_t)height * c->S;
    hipError_t e = hipSuccess;
    auto A = [&](void **p, size_t bytes) { if (e == hipSuccess) e = dev_alloc(p, bytes); };
    if (e == hipSuccess) e = make_stream(&c->stream);
    A((void **)&c->d_img, B * np);
    if (e == hipSuccess && logical_width != width) e = hipMemset(c->d_img, 0, B * np);   // the padding columns stay zero
    A((void **)&c->d_pyr, B * np * 4);
    A((void **)&c->d_cand, B * HS * 8);
    A((void **)&c->d_surv, B * HS * 8);
    A((void **)&c->d_key, B * np * 4);
    A((void **)&c->d_rowcount, B * (size_t)height * 4);
    A((void **)&c->d_rowbase, B * ((size_t)height + 1) * 4);
    A((void **)&c->d_gradp, B * sift::gradp_frame_words(width, height) * 2);
    A((void **)&c->d_kps, B * (size_t)max_keypoints * 8);
    A((void **)&c->d_desc, B * (size_t)max_keypoints * 72 * 4);
    A((void **)&c->d_total, B * 4);
    A((void **)&c->d_total_ctx, B * 4);
    A((void **)&c->d_ori, B * (size_t)max_keypoints);
    A((void **)&c->d_tab, tab.size() * sizeof(uint2));
    c->ntab = (int)tab.size(); c->n2 = wt.n2; c->n2pad = wt.n2pad;
    c->qa = qt.qa; c->qu2 = qt.qu2; c->qu = qt.qu; c->qb = qt.qb;
    c->radius = geo.radius; c->qrow4 = ((geo.window + 2) / 4 + 1) * 4; c->nspec = (int)spec.size();
    A((void **)&c->d_spec, std::max<size_t>(1, spec.size()) * sizeof(uint2));
    if (e == hipSuccess && !spec.empty()) e = hipMemcpy(c->d_spec, spec.data(), spec.size() * sizeof(uint2), hipMemcpyHostToDevice);
    A((void **)&c->d_gate, 4);
    A((void **)&c->d_qtab, qt.bytes.size());
    if (e == hipSuccess) e = hipMemcpy(c->d_qtab, qt.bytes.data(), qt.bytes.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_tab, tab.data(), tab.size() * sizeof(uint2), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        sift_hip_destroy(c);
        return fail(SIFT_HIP_EHIP, "sift_hip_create", e);
    }
    *out = c;
    return SIFT_HIP_OK;
}

void sift_hip_destroy(sift_hip_ctx *c)
{
    if (!c) return;
    DevGuard g(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    void *ptrs[] = {c->d_img, c->d_pyr, c->d_cand, c->d_surv, c->d_key, c->d_rowcount, c->d_rowbase, c->d_gradp,
                    c->d_gradfull, c->d_final, c->d_findex, c->d_nvalid, c->d_tfinal, c->d_tindex, c->d_tdesc, c->d_tnvalid, c->d_bgr, c->d_oct, c->d_match, c->d_kps, c->d_desc, c->d_total, c->d_total_ctx, c->d_ori, c->d_tab, c->d_qtab, c->d_gate, c->d_spec};
    for (void *p : ptrs) if (p) hipFree(p);
    for (auto &ps : c->prof)
        for (int i = 0; i < SIFT_HIP_NSTAGES; i++)
            for (int k = 0; k < 2; k++) if (ps.ev[i][k]) hipEventDestroy(ps.ev[i][k]);
    if (c->gexec) (void)hipGraphExecDestroy(c->gexec);
    for (hipStream_t bs : {c->s_in, c->s_out}) if (bs) { (void)hipStreamSynchronize(bs); (void)hipStreamDestroy(bs); }
    for (int i = 0; i < 2; i++)
        for (hipEvent_t ev : {c->in_done[i], c->k_done[i], c->out_done[i]}) if (ev) (void)hipEventDestroy(ev);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int sift_hip_sync(sift_hip_ctx *c)
{
    if (int r = check_ctx(c)) return r;
    DevGuard g(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SIFT_HIP_OK;
}

int sift_hip_set_profiling(sift_hip_ctx *c, int nslots)
{
    if (int r = check_ctx(c)) return r;
    if (nslots < 0 || nslots > 4096) return fail(SIFT_HIP_EINVAL, "nslots outside [0, 4096]");
    DevGuard g(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto &ps : c->prof)
        for (int i = 0; i < SIFT_HIP_NSTAGES; i++)
            for (int k = 0; k < 2; k++) if (ps.ev[i][k]) hipEventDestroy(ps.ev[i][k]);
    c->prof.clear();
    c->ncalls = 0;
    c->cur = -1;
    c->prof.resize((size_t)nslots);
    for (auto &ps : c->prof)
        for (int i = 0; i < SIFT_HIP_NSTAGES; i++)
            for (int k = 0; k < 2; k++) HIP_TRY(make_event(&ps.ev[i][k], hipEventDefault));
    return SIFT_HIP_OK;
}

int sift_hip_set_blur_rows(sift_hip_ctx *c, int rows)
{
    if (int r = check_ctx(c)) return r;
    if (rows < 0) return fail(SIFT_HIP_EINVAL, "rows must be >= 0");
    c->blur_rows = rows;
    return SIFT_HIP_OK;
}

int sift_hip_set_blur_mfma(sift_hip_ctx *c, int on)
{
    if (int r = check_ctx(c)) return r;
    c->blur_mfma = on != 0;
    if (c->gexec) { (void)hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }   // a captured graph holds the old choice
    c->gkey = sift_hip_ctx::GraphKey{};
    c->gseen = 0;
    return SIFT_HIP_OK;
}

int sift_hip_set_graph_frames(sift_hip_ctx *c, int max_frames)
{
    if (int r = check_ctx(c)) return r;
    if (max_frames < 0) return fail(SIFT_HIP_EINVAL, "max_frames must be >= 0");
    c->graph_max_frames = max_frames;
    return SIFT_HIP_OK;
}

int sift_hip_set_quad_sweep(sift_hip_ctx *c, int on)
{
    if (int r = check_ctx(c)) return r;
    c->quad_sweep = on != 0;
    if (c->gexec) { (void)hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }   // a captured graph holds the old choice
    c->gkey = sift_hip_ctx::GraphKey{};
    c->gseen = 0;
    return SIFT_HIP_OK;
}

int sift_hip_set_detect_rows(sift_hip_ctx *c, int rows)
{
    if (int r = check_ctx(c)) return r;
    if (rows < 0) return fail(SIFT_HIP_EINVAL, "rows must be >= 0");
    c->detect_rows = rows;
    return SIFT_HIP_OK;
}

int sift_hip_set_detect_queue(sift_hip_ctx *c, int entries)
{
    if (int r = check_ctx(c)) return r;
    if (entries < 0) return fail(SIFT_HIP_EINVAL, "entries must be >= 0");
    c->detect_queue = entries == 0 ? (int)sift::DETECT_QUEUE : std::max(256, std::min(entries, (int)sift::DETECT_QUEUE));
    if (c->gexec) { (void)hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }   // a captured graph holds the old value
    c->gkey = sift_hip_ctx::GraphKey{};
    c->gseen = 0;
    return SIFT_HIP_OK;
}

int sift_hip_stage_ms(sift_hip_ctx *c, int back, float ms[SIFT_HIP_NSTAGES])
{
    if (int r = check_ctx(c)) return r;
    if (!ms) return fail(SIFT_HIP_EINVAL, "ms is null");
    for (int i = 0; i < SIFT_HIP_NSTAGES; i++) ms[i] = -1.0f;
    const long n = (long)c->prof.size();
    const long have = c->ncalls < n ? c->ncalls : n;
    if (back < 0 || back >= have) return fail(SIFT_HIP_EINVAL, "no profiling record that far back");
    DevGuard g(c->device);
    const sift_hip_ctx::ProfSlot &ps = c->prof[(size_t)((c->ncalls - 1 - back) % n)];
    for (int i = 0; i < SIFT_HIP_NSTAGES; i++)
        if (ps.used[i]) {
            HIP_TRY(hipEventSynchronize(ps.ev[i][1]));
            HIP_TRY(hipEventElapsedTime(&ms[i], ps.ev[i][0], ps.ev[i][1]));
        }
    return (int)have;
}

/* ---- device-buffer entry points ---------------------------------------- */

int sift_hip_detect_dev(sift_hip_ctx *c, const uint8_t *d_imgs, int nframes, sift_grad *d_grad, sift_keypoint *d_kps,
                        uint32_t *d_total, int contrastTh, void *stream)
{
    if (int r = check_ctx(c)) return r;
    if (!d_imgs || !d_kps) return fail(SIFT_HIP_EINVAL, "null buffer");
    if (nframes < 1 || nframes > c->max_batch) return fail(SIFT_HIP_EINVAL, "nframes outside [1, max_batch]");
    DevGuard g(c->device);
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    begin_call(c);
    return run_detect(c, pad_frames(c, d_imgs, nframes, c->d_img, s), nframes, (uint32_t *)d_grad, (sift::kp_t *)d_kps, d_total,
                      contrastTh, s);
}

int sift_hip_detect_and_compute_dev(sift_hip_ctx *c, const uint8_t *d_imgs, int nframes, sift_keypoint *d_kps,
                                    uint32_t *d_desc, uint32_t *d_total, int contrastTh, void *stream)
{
    return fused_dev(c, d_imgs, false, nframes, d_kps, d_desc, d_total, contrastTh, stream);
}

}  // extern "C"

// padded: d_imgs already has the internal pitch (the host-buffer entry points' staging)
static int fused_dev(sift_hip_ctx *c, const uint8_t *d_imgs, bool padded, int nframes, sift_keypoint *d_kps, uint32_t *d_desc,
                     uint32_t *d_total, int contrastTh, void *stream)
{
    if (int r = check_ctx(c)) return r;
    if (!d_imgs || !d_kps || !d_desc) return fail(SIFT_HIP_EINVAL, "null buffer");
    if (nframes < 1 || nframes > c->max_batch) return fail(SIFT_HIP_EINVAL, "nframes outside [1, max_batch]");
    DevGuard g(c->device);
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    begin_call(c);
    auto enqueue = [&]() -> int {
        const uint8_t *src = padded ? d_imgs : pad_frames(c, d_imgs, nframes, c->d_img, s);
        if (int r = run_detect(c, src, nframes, nullptr, (sift::kp_t *)d_kps, d_total, contrastTh, s)) return r;
        return run_describe<sift::GRAD_PACKED16, sift::DESC_FUSED>(c, c->d_gradp, nframes, (const sift::kp_t *)d_kps,
                                                                   nullptr, nullptr, d_desc, s);
    };
    // Small batches: the memset and the five kernels are replayed as one graph when the call repeats
    // the previous one (same buffers, batch and parameters), which takes the launch gaps out.  Not
    // with profiling on (its events belong to single launches) and not on the legacy default stream.
    if (c->cur >= 0 || nframes > c->graph_max_frames || s == nullptr) return enqueue();
    const sift_hip_ctx::GraphKey key = {d_imgs, d_kps, d_desc, d_total, nframes, contrastTh, c->blur_rows, c->detect_rows};
    const sift_hip_ctx::GraphKey &k = c->gkey;
    const bool same = k.imgs == key.imgs && k.kps == key.kps && k.desc == key.desc && k.total == key.total &&
                      k.nframes == key.nframes && k.th == key.th && k.brows == key.brows && k.drows == key.drows;
    if (!same) {
        // a new call shape: plain launches; the graph is built when the shape has come three times in a
        // row (building and first launching a graph can cost milliseconds, a one-off call must not pay that)
        if (c->gexec) { (void)hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
        c->gkey = key;
        c->gseen = 1;
        return enqueue();
    }
    if (!c->gexec) {
        if (++c->gseen < 3) return enqueue();
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return enqueue(); }
        const int r = enqueue();
        const hipError_t e = hipStreamEndCapture(s, &graph);
        if (r) { if (graph) (void)hipGraphDestroy(graph); return r; }
        if (e != hipSuccess || !graph) return fail(SIFT_HIP_EHIP, "hipStreamEndCapture", e);
        const hipError_t ei = hipGraphInstantiate(&c->gexec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) { c->gexec = nullptr; return fail(SIFT_HIP_EHIP, "hipGraphInstantiate", ei); }
        (void)hipGraphUpload(c->gexec, s);      // otherwise the first replay pays for it
    }
    HIP_TRY(hipGraphLaunch(c->gexec, s));
    return SIFT_HIP_OK;
}

extern "C" {

int sift_hip_main_orientation_dev(sift_hip_ctx *c, const sift_grad *d_grad, int nframes, const sift_keypoint *d_kps,
                                  uint8_t *d_ori, void *stream)
{
    if (int r = check_ctx(c)) return r;
    if (!d_grad || !d_kps || !d_ori) return fail(SIFT_HIP_EINVAL, "null buffer");
    if (nframes < 1 || nframes > c->max_batch) return fail(SIFT_HIP_EINVAL, "nframes outside [1, max_batch]");
    DevGuard g(c->device);
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    begin_call(c);
    return run_describe_planes<sift::DESC_ORI_ONLY>(c, d_grad, nframes, (const sift::kp_t *)d_kps, nullptr, d_ori, nullptr, s);
}

int sift_hip_descriptor_dev(sift_hip_ctx *c, const sift_grad *d_grad, int nframes, const sift_keypoint *d_kps,
                            const uint8_t *d_ori, uint32_t *d_desc, void *stream)
{
    if (int r = check_ctx(c)) return r;
    if (!d_grad || !d_kps || !d_ori || !d_desc) return fail(SIFT_HIP_EINVAL, "null buffer");
    if (nframes < 1 || nframes > c->max_batch) return fail(SIFT_HIP_EINVAL, "nframes outside [1, max_batch]");
    DevGuard g(c->device);
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    begin_call(c);
    return run_describe_planes<sift::DESC_GIVEN_ORI>(c, d_grad, nframes, (const sift::kp_t *)d_kps, d_ori, nullptr, d_desc, s);
}

int sift_hip_select_top_dev(sift_hip_ctx *c, const sift_keypoint *d_kps, int nframes, int nfinal, sift_keypoint *d_final,
                            uint32_t *d_index, uint32_t *d_nvalid, const uint32_t *d_desc, uint32_t *d_final_desc,
                            void *stream)
{
    if (int r = check_ctx(c)) return r;
    if (!d_kps || !d_final) return fail(SIFT_HIP_EINVAL, "null buffer");
    if (nframes < 1) return fail(SIFT_HIP_EINVAL, "nframes must be >= 1");
    if (nfinal < 1 || nfinal > c->max_kp || nfinal > sift::SEL_MAX_FINAL)
        return fail(SIFT_HIP_EINVAL, "nfinal outside [1, min(max_keypoints, 4096)]");
    if ((d_desc == nullptr) != (d_final_desc == nullptr))
        return fail(SIFT_HIP_EINVAL, "d_desc and d_final_desc go together");
    DevGuard g(c->device);
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    hipLaunchKernelGGL(sift::k_select_top, dim3(nframes), dim3(sift::SEL_THREADS), 0, s, (const sift::kp_t *)d_kps,
                       c->max_kp, nfinal, (sift::kp_t *)d_final, d_index, d_nvalid, d_desc, d_final_desc);
    HIP_TRY(hipGetLastError());
    return SIFT_HIP_OK;
}

int sift_hip_match_dev(sift_hip_ctx *c, const uint32_t *d_query, int nquery, const uint32_t *d_train, int ntrain, int npairs,
                       sift_dmatch *d_matches, void *stream)
{
    if (int r = check_ctx(c)) return r;
    if (nquery < 0 || ntrain < 0 || nquery > 32767 || ntrain > 32767)
        return fail(SIFT_HIP_EINVAL, "nquery and ntrain must lie in [0, 32767] (DMatch_t holds 16-bit indices)");
    if (npairs < 1) return fail(SIFT_HIP_EINVAL, "npairs must be >= 1");
    if (nquery == 0) return SIFT_HIP_OK;
    if (!d_query || !d_matches || (ntrain > 0 && !d_train)) return fail(SIFT_HIP_EINVAL, "null buffer");
    if (((uintptr_t)d_query | (uintptr_t)d_train) & 15) return fail(SIFT_HIP_EINVAL, "descriptor arrays must be 16-byte aligned");
    DevGuard g(c->device);
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    dim3 grid((nquery + sift::MATCH_QUERIES - 1) / sift::MATCH_QUERIES, npairs);
    hipLaunchKernelGGL(sift::k_match, grid, dim3(64 * sift::MATCH_WAVES), 0, s, d_query, nquery, d_train, ntrain,
                       (sift::dmatch_t *)d_matches);
    HIP_TRY(hipGetLastError());
    return SIFT_HIP_OK;
}

int sift_hip_match(sift_hip_ctx *c, const uint32_t *query, int nquery, const uint32_t *train, int ntrain, sift_dmatch *matches)
{
    if (int r = check_ctx(c)) return r;
    if (nquery < 0 || ntrain < 0 || nquery > 32767 || ntrain > 32767)
        return fail(SIFT_HIP_EINVAL, "nquery and ntrain must lie in [0, 32767] (DMatch_t holds 16-bit indices)");
    if (nquery == 0) return SIFT_HIP_OK;
    if (!query || !matches || (ntrain > 0 && !train)) return fail(SIFT_HIP_EINVAL, "null buffer");
    DevGuard g(c->device);
    hipStream_t s = c->stream;
    // staging: both descriptor sets and the matches (grown on demand; at most 2 x 32767 rows)
    const size_t qb = (size_t)nquery * 288, tb = (size_t)ntrain * 288, mb = (size_t)nquery * sizeof(sift_dmatch);
    const size_t need = ((qb + 255) & ~(size_t)255) + ((tb + 255) & ~(size_t)255) + mb;
    if (c->match_cap < need) {
        if (c->d_match) (void)hipFree(c->d_match);
        c->d_match = nullptr; c->match_cap = 0;
        HIP_TRY(dev_alloc(&c->d_match, need));
        c->match_cap = need;
    }
    uint8_t *d_q = c->d_match, *d_t = d_q + ((qb + 255) & ~(size_t)255), *d_m = d_t + ((tb + 255) & ~(size_t)255);
    HIP_TRY(hipMemcpyAsync(d_q, query, qb, hipMemcpyHostToDevice, s));
    if (tb) HIP_TRY(hipMemcpyAsync(d_t, train, tb, hipMemcpyHostToDevice, s));
    if (int r = sift_hip_match_dev(c, (const uint32_t *)d_q, nquery, (const uint32_t *)d_t, ntrain, 1, (sift_dmatch *)d_m, s)) return r;
    HIP_TRY(hipMemcpyAsync(matches, d_m, mb, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SIFT_HIP_OK;
}

int sift_hip_gray_from_bgr_dev(sift_hip_ctx *c, const uint8_t *d_bgr, int nframes, long pitch, int bottom_up,
                               uint8_t *d_gray, void *stream)
{
    if (int r = check_ctx(c)) return r;
    if (!d_bgr || !d_gray) return fail(SIFT_HIP_EINVAL, "null buffer");
    if (nframes < 1) return fail(SIFT_HIP_EINVAL, "nframes must be >= 1");
    if (pitch == 0) pitch = (3L * c->Wl + 3) & ~3L;
    if (pitch < 3L * c->Wl || (pitch & 3)) return fail(SIFT_HIP_EINVAL, "pitch must be a multiple of 4 and >= 3 * width");
    if ((uintptr_t)d_bgr & 3) return fail(SIFT_HIP_EINVAL, "d_bgr must be 4-byte aligned");
    DevGuard g(c->device);
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    dim3 grid(((c->Wl + 3) / 4 + 255) / 256, c->H, nframes);
    hipLaunchKernelGGL(sift::k_bgr2gray, grid, dim3(256), 0, s, d_bgr, d_gray, c->Wl, c->H, pitch, pitch * c->H,
                       bottom_up ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return SIFT_HIP_OK;
}

/* ---- host-buffer entry points ------------------------------------------ */

int sift_hip_gray_from_bgr(sift_hip_ctx *c, const uint8_t *bgr, long pitch, int bottom_up, uint8_t *gray)
{
    if (int r = check_ctx(c)) return r;
    if (!bgr || !gray) return fail(SIFT_HIP_EINVAL, "null buffer");
    if (pitch == 0) pitch = (3L * c->Wl + 3) & ~3L;
    if (pitch < 3L * c->Wl || (pitch & 3)) return fail(SIFT_HIP_EINVAL, "pitch must be a multiple of 4 and >= 3 * width");
    DevGuard g(c->device);
    hipStream_t s = c->stream;
    // staging: the colour rows, then the gray frame (rows of the logical width)
    const size_t bytes = (size_t)pitch * c->H, gbytes = (size_t)c->Wl * c->H;
    if (c->bgr_cap < bytes + gbytes) {
        if (c->d_bgr) (void)hipFree(c->d_bgr);
        c->d_bgr = nullptr; c->bgr_cap = 0;
        HIP_TRY(dev_alloc(&c->d_bgr, bytes + gbytes));
        c->bgr_cap = bytes + gbytes;
    }
    HIP_TRY(hipMemcpyAsync(c->d_bgr, bgr, bytes, hipMemcpyHostToDevice, s));
    if (int r = sift_hip_gray_from_bgr_dev(c, c->d_bgr, 1, pitch, bottom_up, c->d_bgr + bytes, s)) return r;
    HIP_TRY(hipMemcpyAsync(gray, c->d_bgr + bytes, gbytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SIFT_HIP_OK;
}

int sift_hip_select_top(sift_hip_ctx *c, const sift_keypoint *kps, int nfinal, sift_keypoint *final_kps, uint32_t *index,
                        uint32_t *nvalid)
{
    if (int r = check_ctx(c)) return r;
    if (!kps || !final_kps) return fail(SIFT_HIP_EINVAL, "null buffer");
    if (nfinal < 1 || nfinal > c->max_kp || nfinal > sift::SEL_MAX_FINAL)
        return fail(SIFT_HIP_EINVAL, "nfinal outside [1, min(max_keypoints, 4096)]");
    DevGuard g(c->device);
    hipStream_t s = c->stream;
    if (int r = ensure_select(c)) return r;
    HIP_TRY(hipMemcpyAsync(c->d_kps, kps, (size_t)c->max_kp * 8, hipMemcpyHostToDevice, s));
    if (int r = sift_hip_select_top_dev(c, (const sift_keypoint *)c->d_kps, 1, nfinal, (sift_keypoint *)c->d_final,
                                        c->d_findex, c->d_nvalid, nullptr, nullptr, s))
        return r;
    HIP_TRY(hipMemcpyAsync(final_kps, c->d_final, (size_t)nfinal * 8, hipMemcpyDeviceToHost, s));
    if (index) HIP_TRY(hipMemcpyAsync(index, c->d_findex, (size_t)nfinal * 4, hipMemcpyDeviceToHost, s));
    if (nvalid) HIP_TRY(hipMemcpyAsync(nvalid, c->d_nvalid, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SIFT_HIP_OK;
}

int sift_hip_detect_and_compute(sift_hip_ctx *c, const uint8_t *img, sift_keypoint *kps, uint32_t *desc, int contrastTh)
{
    if (int r = check_ctx(c)) return r;
    if (!img || !kps || !desc) return fail(SIFT_HIP_EINVAL, "null buffer");
    DevGuard g(c->device);
    hipStream_t s = c->stream;
    HIP_TRY(upload_frame(c, c->d_img, img, s));
    if (int r = fused_dev(c, c->d_img, true, 1, (sift_keypoint *)c->d_kps, c->d_desc, c->d_total, contrastTh, s)) return r;
    HIP_TRY(hipMemcpyAsync(kps, c->d_kps, (size_t)c->max_kp * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(desc, c->d_desc, (size_t)c->max_kp * 72 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SIFT_HIP_OK;
}

int sift_hip_detect(sift_hip_ctx *c, const uint8_t *img, sift_grad *grad, sift_keypoint *kps, int contrastTh)
{
    if (int r = check_ctx(c)) return r;
    if (!img || !kps) return fail(SIFT_HIP_EINVAL, "null buffer");
    DevGuard g(c->device);
    hipStream_t s = c->stream;
    if (grad) if (int r = ensure_gradfull(c)) return r;
    HIP_TRY(upload_frame(c, c->d_img, img, s));
    begin_call(c);
    if (int r = run_detect(c, c->d_img, 1, grad ? c->d_gradfull : nullptr, c->d_kps, c->d_total, contrastTh, s)) return r;
    HIP_TRY(hipMemcpyAsync(kps, c->d_kps, (size_t)c->max_kp * 8, hipMemcpyDeviceToHost, s));
    if (grad) HIP_TRY(hipMemcpyAsync(grad, c->d_gradfull, (size_t)c->Wl * c->H * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SIFT_HIP_OK;
}

int sift_hip_main_orientation(sift_hip_ctx *c, const sift_grad *grad, const sift_keypoint *kps, uint8_t *ori)
{
    if (int r = check_ctx(c)) return r;
    if (!grad || !kps || !ori) return fail(SIFT_HIP_EINVAL, "null buffer");
    DevGuard g(c->device);
    hipStream_t s = c->stream;
    if (int r = ensure_gradfull(c)) return r;
    HIP_TRY(hipMemcpyAsync(c->d_gradfull, grad, (size_t)c->Wl * c->H * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->d_kps, kps, (size_t)c->max_kp * 8, hipMemcpyHostToDevice, s));
    if (int r = sift_hip_main_orientation_dev(c, (const sift_grad *)c->d_gradfull, 1, (const sift_keypoint *)c->d_kps,
                                              c->d_ori, s))
        return r;
    HIP_TRY(hipMemcpyAsync(ori, c->d_ori, (size_t)c->max_kp, hipMemcpyDeviceToHost, s));
    return finish_planes_call(c, s);
}

int sift_hip_descriptor(sift_hip_ctx *c, const sift_grad *grad, const sift_keypoint *kps, const uint8_t *ori,
                        uint32_t *desc)
{
    if (int r = check_ctx(c)) return r;
    if (!grad || !kps || !ori || !desc) return fail(SIFT_HIP_EINVAL, "null buffer");
    DevGuard g(c->device);
    hipStream_t s = c->stream;
    if (int r = ensure_gradfull(c)) return r;
    HIP_TRY(hipMemcpyAsync(c->d_gradfull, grad, (size_t)c->Wl * c->H * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->d_kps, kps, (size_t)c->max_kp * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->d_ori, ori, (size_t)c->max_kp, hipMemcpyHostToDevice, s));
    if (int r = sift_hip_descriptor_dev(c, (const sift_grad *)c->d_gradfull, 1, (const sift_keypoint *)c->d_kps,
                                        c->d_ori, c->d_desc, s))
        return r;
    HIP_TRY(hipMemcpyAsync(desc, c->d_desc, (size_t)c->max_kp * 72 * 4, hipMemcpyDeviceToHost, s));
    return finish_planes_call(c, s);
}

}  // extern "C"

namespace {

// per-chunk device buffers of the selected keypoints (sift_hip_batch_top), two chunks in flight
int ensure_top_buffers(sift_hip_ctx *c, int nfinal)
{
    const size_t n = (size_t)c->max_batch * (size_t)nfinal;
    if (c->top_cap >= n) return SIFT_HIP_OK;
    for (void *p : {(void *)c->d_tfinal, (void *)c->d_tindex, (void *)c->d_tdesc, (void *)c->d_tnvalid}) if (p) (void)hipFree(p);
    c->d_tfinal = nullptr; c->d_tindex = nullptr; c->d_tdesc = nullptr; c->d_tnvalid = nullptr; c->top_cap = 0;
    HIP_TRY(dev_alloc(&c->d_tfinal, n * sizeof(sift::kp_t)));
    HIP_TRY(dev_alloc(&c->d_tindex, n * 4));
    HIP_TRY(dev_alloc(&c->d_tdesc, n * 72 * 4));
    HIP_TRY(dev_alloc(&c->d_tnvalid, (size_t)c->max_batch * 4));
    c->top_cap = n;
    return SIFT_HIP_OK;
}

// nfinal == 0: every keypoint slot and descriptor row goes back to the host (sift_hip_batch);
// nfinal > 0: the frames' nfinal strongest keypoints, their slots and descriptor rows (sift_hip_batch_top)
int batch_impl(sift_hip_ctx *const *ctxs, int nctx, const uint8_t *const *imgs, int nframes, sift_keypoint *kps,
               uint32_t *desc, int contrastTh, int nfinal, sift_keypoint *final_kps, uint32_t *final_index,
               uint32_t *final_desc, uint32_t *nvalid)
{
    if (!ctxs || nctx < 1 || !imgs || nframes < 0) return fail(SIFT_HIP_EINVAL, "bad argument");
    if (nfinal == 0 && (!kps || !desc)) return fail(SIFT_HIP_EINVAL, "bad argument");
    if (nfinal != 0) {
        if (!final_kps) return fail(SIFT_HIP_EINVAL, "final_kps is null");
        if (nfinal < 1 || nfinal > ctxs[0]->max_kp || nfinal > sift::SEL_MAX_FINAL)
            return fail(SIFT_HIP_EINVAL, "nfinal outside [1, min(max_keypoints, 4096)]");
    }
    for (int i = 0; i < nctx; i++) {
        if (!ctxs[i]) return fail(SIFT_HIP_EINVAL, "null context");
        if (ctxs[i]->W != ctxs[0]->W || ctxs[i]->H != ctxs[0]->H || ctxs[i]->max_kp != ctxs[0]->max_kp)
            return fail(SIFT_HIP_EINVAL, "contexts differ in frame size or max_keypoints");
    }
    for (int f = 0; f < nframes; f++) if (!imgs[f]) return fail(SIFT_HIP_EINVAL, "null frame");
    for (int i = 0; i < nctx; i++) {
        DevGuard g(ctxs[i]->device);
        if (int r = ensure_batch_streams(ctxs[i])) return r;
        if (nfinal) if (int r = ensure_top_buffers(ctxs[i], nfinal)) return r;
    }
    // One host thread per context (= per GPU).  Context i takes the contiguous range of frames
    // sift_hip_batch_range gives it (ceil(nframes / nctx) each), so whatever the number of
    // contexts a chunk's frames are neighbours in the caller's arrays: its results go back as one copy per
    // array, and frames that lie back to back in host memory go up as one copy.  No exchange between
    // contexts.  Inside a context the frames go through in chunks of half its max_batch, double buffered:
    // the H2D copy of chunk k+1 and the D2H copy of chunk k-1 run on the context's two copy streams beside the
    // kernels of chunk k (the overlapped ingest of main.cpp:41-58 / :82-184).  Streams and events belong to
    // the context: a call creates nothing.
    std::vector<int> rc(nctx, SIFT_HIP_OK);
    std::vector<std::string> msg(nctx);
    auto worker = [&](int ci) {
        sift_hip_ctx *c = ctxs[ci];
        DevGuard g(c->device);
        const size_t np = c->npix, K = (size_t)c->max_kp;
        int first = 0, last = 0;
        sift_hip_batch_range(nframes, nctx, ci, &first, &last);
        if (first >= last) return;
        auto step = [&](hipError_t e, const char *what) {
            if (e != hipSuccess && rc[ci] == SIFT_HIP_OK) { rc[ci] = SIFT_HIP_EHIP; msg[ci] = std::string(what) + ": " + hipGetErrorString(e); }
        };
        const int half = c->max_batch >= 2 ? c->max_batch / 2 : c->max_batch;   // frames per chunk
        const int nbuf = c->max_batch >= 2 ? 2 : 1;
        hipStream_t s_in = c->s_in, s_out = c->s_out, s = c->stream;
        hipEvent_t *in_done = c->in_done, *k_done = c->k_done, *out_done = c->out_done;
        size_t chunk = 0;
        for (int b0 = first; b0 < last && rc[ci] == SIFT_HIP_OK; b0 += half, chunk++) {
            const int nb = std::min(half, last - b0);
            const int h = (int)(chunk % (size_t)nbuf);
            uint8_t *d_img = c->d_img + (size_t)h * half * np;
            sift::kp_t *d_kps = c->d_kps + (size_t)h * half * K;
            uint32_t *d_desc = c->d_desc + (size_t)h * half * K * 72;
            // H2D into buffer h once the kernels that last read it (chunk - nbuf) are done
            if (chunk >= (size_t)nbuf) step(hipStreamWaitEvent(s_in, k_done[h], 0), "wait");
            // frames that lie back to back in host memory travel as one copy (fewer, larger transfers)
            const size_t host_frame = (size_t)c->Wl * c->H;
            for (int k = 0; k < nb;) {
                int run = 1;
                if (c->Wl == c->W)
                    while (k + run < nb && imgs[b0 + k + run] == imgs[b0 + k] + (size_t)run * host_frame) run++;
                if (run > 1)
                    step(hipMemcpyAsync(d_img + (size_t)k * np, imgs[b0 + k], (size_t)run * np, hipMemcpyHostToDevice, s_in), "H2D frames");
                else
                    step(upload_frame(c, d_img + (size_t)k * np, imgs[b0 + k], s_in), "H2D frame");
                k += run;
            }
            step(hipEventRecord(in_done[h], s_in), "record");
            // kernels: after the frames have arrived and the previous outputs of buffer h have left
            step(hipStreamWaitEvent(s, in_done[h], 0), "wait");
            if (chunk >= (size_t)nbuf) step(hipStreamWaitEvent(s, out_done[h], 0), "wait");
            if (rc[ci] == SIFT_HIP_OK) {
                int r = fused_dev(c, d_img, true, nb, (sift_keypoint *)d_kps, d_desc, c->d_total + (size_t)h * half, contrastTh, s);
                if (r) { rc[ci] = r; msg[ci] = g_err; }
                if (!r && nfinal) {
                    r = sift_hip_select_top_dev(c, (sift_keypoint *)d_kps, nb, nfinal,
                                                (sift_keypoint *)(c->d_tfinal + (size_t)h * half * nfinal),
                                                c->d_tindex + (size_t)h * half * nfinal, c->d_tnvalid + (size_t)h * half,
                                                d_desc, c->d_tdesc + (size_t)h * half * nfinal * 72, s);
                    if (r) { rc[ci] = r; msg[ci] = g_err; }
                }
            }
            step(hipEventRecord(k_done[h], s), "record");
            // D2H: the chunk's frames are consecutive in the caller's arrays, one copy per array
            step(hipStreamWaitEvent(s_out, k_done[h], 0), "wait");
            if (rc[ci] == SIFT_HIP_OK) {
                const size_t f = (size_t)b0, n = (size_t)nb;
                if (!nfinal) {
                    step(hipMemcpyAsync(kps + f * K, d_kps, n * K * 8, hipMemcpyDeviceToHost, s_out), "D2H keypoints");
                    step(hipMemcpyAsync(desc + f * K * 72, d_desc, n * K * 72 * 4, hipMemcpyDeviceToHost, s_out), "D2H descriptors");
                } else {
                    const size_t NF = (size_t)nfinal, o = (size_t)h * half * NF;
                    step(hipMemcpyAsync(final_kps + f * NF, c->d_tfinal + o, n * NF * 8, hipMemcpyDeviceToHost, s_out), "D2H final keypoints");
                    if (final_index) step(hipMemcpyAsync(final_index + f * NF, c->d_tindex + o, n * NF * 4, hipMemcpyDeviceToHost, s_out), "D2H slots");
                    if (final_desc) step(hipMemcpyAsync(final_desc + f * NF * 72, c->d_tdesc + o * 72, n * NF * 72 * 4, hipMemcpyDeviceToHost, s_out), "D2H descriptor rows");
                    if (nvalid) step(hipMemcpyAsync(nvalid + f, c->d_tnvalid + (size_t)h * half, n * 4, hipMemcpyDeviceToHost, s_out), "D2H counts");
                }
            }
            step(hipEventRecord(out_done[h], s_out), "record");
        }
        // all three streams drain before the call returns, whatever happened above
        step(hipStreamSynchronize(s_in), "sync");
        step(hipStreamSynchronize(s), "sync");
        step(hipStreamSynchronize(s_out), "sync");
    };
    if (nctx == 1) {
        worker(0);
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < nctx; i++) th.emplace_back(worker, i);
        for (auto &t : th) t.join();
    }
    for (int i = 0; i < nctx; i++)
        if (rc[i] != SIFT_HIP_OK) return fail(rc[i], msg[i].c_str());
    return SIFT_HIP_OK;
}

}  // namespace

extern "C" {

int sift_hip_batch_range(int nframes, int nctx, int i, int *first, int *last)
{
    if (nframes < 0 || nctx < 1 || i < 0 || i >= nctx || !first || !last) return fail(SIFT_HIP_EINVAL, "bad partition");
    const long per = ((long)nframes + nctx - 1) / nctx;
    *first = (int)std::min((long)nframes, (long)i * per);
    *last = (int)std::min((long)nframes, (long)*first + per);
    return SIFT_HIP_OK;
}

long sift_hip_debug_created(int what)
{
    return (what >= 0 && what < 3) ? g_created[what].load() : -1;
}

int sift_hip_batch(sift_hip_ctx *const *ctxs, int nctx, const uint8_t *const *imgs, int nframes, sift_keypoint *kps,
                   uint32_t *desc, int contrastTh)
{
    return batch_impl(ctxs, nctx, imgs, nframes, kps, desc, contrastTh, 0, nullptr, nullptr, nullptr, nullptr);
}

int sift_hip_batch_top(sift_hip_ctx *const *ctxs, int nctx, const uint8_t *const *imgs, int nframes, int nfinal,
                       sift_keypoint *final_kps, uint32_t *final_index, uint32_t *final_desc, uint32_t *nvalid,
                       int contrastTh)
{
    if (nfinal < 1) return fail(SIFT_HIP_EINVAL, "nfinal must be >= 1");
    return batch_impl(ctxs, nctx, imgs, nframes, nullptr, nullptr, contrastTh, nfinal, final_kps, final_index, final_desc,
                      nvalid);
}

/* ---- the entry point that allocates nothing ------------------------------------------------ */
}  // extern "C"

// Scratch of sift_hip_detect_and_compute_static: every GPU the library runs on holds its own copy
// of this array until the process ends.  In host code the name is only a stand-in for the symbol:
// the address kernels and hipMemsetAsync get is the one hipGetSymbolAddress returns per device.
static __device__ __align__(256) unsigned char scratch[SIFT_HIP_STATIC_SCRATCH_BYTES];

namespace {

constexpr int STATIC_MAX_DEVICES = 64;
constexpr size_t STATIC_TABLE_BYTES = 256u << 10;     // front of the array: window + quad tables of the current width

constexpr int STATIC_TABLE_SLOTS = 2;                 // widths whose tables stay in the array (a caller that alternates two sizes rebuilds nothing)
struct StaticTables {
    int width = 0;                        // frame width the tables of this slot were built for (0: empty)
    long used = 0;                        // last call that used the slot
    int ntab = 0, n2 = 0, n2pad = 0, qa = 0, qu2 = 0, qu = 0, qb = 0;
    size_t qoff = 0;                      // byte offset of the quad table inside the slot
};
struct StaticState {
    std::mutex mu;                        // calls on one device are serialised on the host; other devices do not wait
    unsigned char *base = nullptr;        // this device's address of `scratch` (hipGetSymbolAddress)
    StaticTables tables[STATIC_TABLE_SLOTS];
    long ncalls = 0;
    sift_hip_ctx ctx;                     // views into the array; owns nothing
};
StaticState g_static[STATIC_MAX_DEVICES];

// the buffers of a call, carved out of the array behind the tables (256-byte aligned)
struct StaticLayout {
    size_t pyr, cand, surv, key, rowcount, rowbase, gradp, total, end;
    StaticLayout(int W, int H, int nframes)
    {
        const size_t B = (size_t)nframes, np = (size_t)W * H, HS = (size_t)H * (size_t)((W + 63) / 64);
        size_t o = STATIC_TABLE_BYTES;
        auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
        pyr = take(B * np * 4);
        cand = take(B * HS * 8);
        surv = take(B * HS * 8);
        key = take(B * np * 4);
        rowcount = take(B * (size_t)H * 4);
        rowbase = take(B * ((size_t)H + 1) * 4);
        gradp = take(B * sift::gradp_frame_words(W, H) * 2);
        total = take(B * 4);
        end = o;
    }
};

}  // namespace

extern "C" {

unsigned long sift_hip_static_scratch_needed(int width, int height, int nframes)
{
    if (width < 4 || (width & 3) || height < 1 || width > 65532 || height > 65535 || nframes < 1) return 0;
    return (unsigned long)StaticLayout(width, height, nframes).end;
}

int sift_hip_detect_and_compute_static(int device, const uint8_t *d_imgs, int nframes, int width, int height,
                                       sift_keypoint *d_kps, uint32_t *d_desc, uint32_t *d_total, int contrastTh,
                                       void *stream)
{
    if (!d_imgs || !d_kps || !d_desc) return fail(SIFT_HIP_EINVAL, "null buffer");
    if (width < 4 || (width & 3) || height < 1 || width > 65532 || height > 65535)
        return fail(SIFT_HIP_EINVAL, "width must be a multiple of 4 in [4,65532], height in [1,65535]");
    if (nframes < 1) return fail(SIFT_HIP_EINVAL, "nframes must be >= 1");
    if (device < 0 || device >= STATIC_MAX_DEVICES) return fail(SIFT_HIP_EINVAL, "device index out of range");
    const StaticLayout L(width, height, nframes);
    if (L.end > (size_t)SIFT_HIP_STATIC_SCRATCH_BYTES) {
        char buf[200];
        snprintf(buf, sizeof buf, "%d frame(s) of %dx%d need %zu bytes of scratch, the static array holds %zu "
                 "(use a context: sift_hip_create)", nframes, width, height, L.end, (size_t)SIFT_HIP_STATIC_SCRATCH_BYTES);
        return fail(SIFT_HIP_EINVAL, buf);
    }
    StaticState &st = g_static[device];
    std::lock_guard<std::mutex> lock(st.mu);
    DevGuard g(device);
    hipStream_t s = (hipStream_t)stream;
    if (!st.base) {
        // once per device, with that device current: its own address of the array
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SIFT_HIP_ENODEV, "no HIP device");
        if (device >= ndev) return fail(SIFT_HIP_EINVAL, "device index out of range");
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(SIFT_HIP_ENODEV, "device is not gfx950 (the library carries gfx950 code only)");
        unsigned char *p = nullptr;
        HIP_TRY(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(SIFT_HIP_EHIP, "hipGetSymbolAddress gave no address");
        st.base = p;
        st.ctx.device = device;
        st.ctx.max_kp = SIFT_HIP_MAX_KEYPOINTS;
        st.ctx.graph_max_frames = 0;
    }
    sift_hip_ctx *c = &st.ctx;
    st.ncalls++;
    StaticTables *tb = nullptr;
    for (StaticTables &t : st.tables) if (t.width == width) tb = &t;
    if (!tb) {
        // set-up for a frame width whose tables are not in the array (first call, or a third width): the descriptor
        // tables hold plane offsets, so they are built on the host and copied into the slot used least recently.
        // The copy is ordered on `stream` after whatever earlier call still reads that slot, and the host waits for
        // it (the vectors go away); this step cannot run under stream capture.  Two widths stay resident, so a caller
        // that alternates two frame sizes rebuilds nothing.
        const WindowTable wt = build_window_table(width, kGeometry[0]);
        const QuadTable qt = build_quad_table(width, kGeometry[0]);
        if (wt.e.empty()) return fail(SIFT_HIP_EINVAL, "circle tables overlap three-fold");
        if (2L * std::max(wt.kmax, qt.kmax) * sift::DESC_MAX_MAG_PACKED >= 65536)
            return fail(SIFT_HIP_EINVAL, "descriptor window too large for 16-bit lane bins");
        const size_t tab_bytes = wt.e.size() * sizeof(uint2), qoff = (tab_bytes + 255) & ~(size_t)255;
        if (qoff + qt.bytes.size() > STATIC_TABLE_BYTES / STATIC_TABLE_SLOTS) return fail(SIFT_HIP_EINVAL, "descriptor tables exceed their share of the static array");
        tb = &st.tables[0];
        for (StaticTables &t : st.tables) if (t.used < tb->used) tb = &t;
        unsigned char *slot = st.base + (size_t)(tb - st.tables) * (STATIC_TABLE_BYTES / STATIC_TABLE_SLOTS);
        tb->width = 0;
        HIP_TRY(hipMemcpyAsync(slot, wt.e.data(), tab_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(slot + qoff, qt.bytes.data(), qt.bytes.size(), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        tb->ntab = (int)wt.e.size(); tb->n2 = wt.n2; tb->n2pad = wt.n2pad;
        tb->qa = qt.qa; tb->qu2 = qt.qu2; tb->qu = qt.qu; tb->qb = qt.qb;
        tb->qoff = qoff;
        tb->width = width;
    }
    tb->used = st.ncalls;
    {
        unsigned char *slot = st.base + (size_t)(tb - st.tables) * (STATIC_TABLE_BYTES / STATIC_TABLE_SLOTS);
        c->d_tab = (uint2 *)slot;
        c->d_qtab = (uint32_t *)(slot + tb->qoff);
        c->ntab = tb->ntab; c->n2 = tb->n2; c->n2pad = tb->n2pad;
        c->qa = tb->qa; c->qu2 = tb->qu2; c->qu = tb->qu; c->qb = tb->qb;
    }
    c->W = width; c->Wl = width; c->H = height; c->S = (width + 63) / 64;
    c->npix = (size_t)width * height;
    c->max_batch = nframes;
    c->d_pyr = (uint4 *)(st.base + L.pyr);
    c->d_cand = (unsigned long long *)(st.base + L.cand);
    c->d_surv = (unsigned long long *)(st.base + L.surv);
    c->d_key = (uint32_t *)(st.base + L.key);
    c->d_rowcount = (uint32_t *)(st.base + L.rowcount);
    c->d_rowbase = (uint32_t *)(st.base + L.rowbase);
    c->d_gradp = (uint16_t *)(st.base + L.gradp);
    c->d_total_ctx = (uint32_t *)(st.base + L.total);
    c->cur = -1;                                      // no profiling records: they would need events
    // The array keeps the previous call's contents.  What must start at zero is cleared by the
    // kernels themselves, as for a context's scratch: the row counts by k_detect; everything else is
    // written before it is read.
    if (int r = run_detect(c, d_imgs, nframes, nullptr, (sift::kp_t *)d_kps, d_total, contrastTh, s)) return r;
    return run_describe<sift::GRAD_PACKED16, sift::DESC_FUSED>(c, c->d_gradp, nframes, (const sift::kp_t *)d_kps, nullptr,
                                                               nullptr, d_desc, s);
}

int sift_hip_next_octave_dev(sift_hip_ctx *c, int frame_first, int nframes, int plane, uint8_t *d_dst, void *stream)
{
    if (int r = check_ctx(c)) return r;
    if (!d_dst) return fail(SIFT_HIP_EINVAL, "null buffer");
    if (plane < 1 || plane > 3) return fail(SIFT_HIP_EINVAL, "plane must be 1, 2 or 3");
    if (nframes < 1 || frame_first < 0 || frame_first + nframes > c->last_frames)
        return fail(SIFT_HIP_EINVAL, "frames outside the last call's batch");
    const int W2 = c->Wl / 2, H2 = c->H / 2;
    if (W2 < 1 || H2 < 1) return fail(SIFT_HIP_EINVAL, "frame too small for another octave");
    DevGuard g(c->device);
    hipLaunchKernelGGL(sift::k_next_octave, dim3((W2 + 255) / 256, H2, nframes), dim3(256), 0, stream ? (hipStream_t)stream : c->stream,
                       c->d_pyr + (size_t)frame_first * (c->npix / 4), d_dst, c->W / 4, c->H, W2, H2, plane);
    HIP_TRY(hipGetLastError());
    return SIFT_HIP_OK;
}

int sift_hip_next_octave(sift_hip_ctx *c, int frame, int plane, uint8_t *dst)
{
    if (int r = check_ctx(c)) return r;
    if (!dst) return fail(SIFT_HIP_EINVAL, "null buffer");
    const size_t n = (size_t)(c->Wl / 2) * (size_t)(c->H / 2);
    if (!n) return fail(SIFT_HIP_EINVAL, "frame too small for another octave");
    DevGuard g(c->device);
    if (!c->d_oct) HIP_TRY(dev_alloc(&c->d_oct, n));      // the size is the context's: allocated once, kept
    if (int r = sift_hip_next_octave_dev(c, frame, 1, plane, c->d_oct, c->stream)) return r;
    HIP_TRY(hipMemcpyAsync(dst, c->d_oct, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SIFT_HIP_OK;
}

int sift_hip_debug_planes(sift_hip_ctx *c, int frame, uint8_t *g1, uint8_t *g2, uint8_t *g3, uint16_t *cand,
                          int32_t *eratio, uint16_t *surv)
{
    if (int r = check_ctx(c)) return r;
    if (frame < 0 || frame >= c->last_frames) return fail(SIFT_HIP_EINVAL, "frame outside the last call's batch");
    DevGuard g(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipDeviceSynchronize());
    const size_t np = c->npix, nl = (size_t)c->Wl * c->H;   // internal and caller-side plane sizes
    if (g1 || g2 || g3) {
        std::vector<uint32_t> pyr(np);    // per group of 4 pixels: src, G1, G2, G3 dwords
        HIP_TRY(hipMemcpy(pyr.data(), c->d_pyr + (size_t)frame * (np / 4), np * 4, hipMemcpyDeviceToHost));
        for (int y = 0; y < c->H; y++)
            for (int x = 0; x < c->Wl; x++) {
                const size_t i = (size_t)y * c->W + x, o = (size_t)y * c->Wl + x;
                const size_t q = (i >> 2) * 4;
                const int sh = (int)(i & 3) * 8;
                if (g1) g1[o] = (uint8_t)(pyr[q + 1] >> sh);
                if (g2) g2[o] = (uint8_t)(pyr[q + 2] >> sh);
                if (g3) g3[o] = (uint8_t)(pyr[q + 3] >> sh);
            }
    }
    if (cand || eratio || surv) {
        uint16_t *d_c = nullptr, *d_s = nullptr;
        int32_t *d_e = nullptr;
        HIP_TRY(dev_alloc(&d_c, nl * 2));
        HIP_TRY(dev_alloc(&d_s, nl * 2));
        HIP_TRY(dev_alloc(&d_e, nl * 4));
        const size_t HS = (size_t)c->H * c->S;
        dim3 grid((c->Wl + 255) / 256, c->H);
        hipLaunchKernelGGL(sift::k_expand_maps, grid, dim3(256), 0, c->stream, c->d_cand + frame * HS,
                           c->d_surv + frame * HS, c->d_key + (size_t)frame * np, d_c, d_e, d_s, c->W, c->H, c->S, c->Wl);
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess && cand) e = hipMemcpy(cand, d_c, nl * 2, hipMemcpyDeviceToHost);
        if (e == hipSuccess && surv) e = hipMemcpy(surv, d_s, nl * 2, hipMemcpyDeviceToHost);
        if (e == hipSuccess && eratio) e = hipMemcpy(eratio, d_e, nl * 4, hipMemcpyDeviceToHost);
        hipFree(d_c); hipFree(d_s); hipFree(d_e);
        if (e != hipSuccess) return fail(SIFT_HIP_EHIP, "debug planes", e);
    }
    return SIFT_HIP_OK;
}

}  // extern "C"
