/*
 * sift_oracle.c -- CPU restatement of the reference hot path (TEST INFRASTRUCTURE).
 *
 * This file is the parity oracle for the HIP path.  It is NOT part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product (sift_amd/) never links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned" in the strict sense.  The reference
 * (/root/reference, a Xilinx SDSoC/HLS project) ships no tests, golden vectors
 * or recorded outputs, and its sources need Xilinx headers (ap_int.h,
 * ap_fixed.h, hls_stream.h, sds_lib.h) that are absent from the image, so it is
 * unbuildable here without writing stand-ins for them, which this project does
 * not do.  The only anchors are facts recorded in SURVEY.md from the survey's
 * own run of the reference: quantised-table sums/maxima/centre rows and the
 * keypoint counts of four named inputs (tests/test_oracle.py checks them).
 *
 * Every function cites the reference lines it restates (paths under
 * /root/reference/src).  All arithmetic is plain integer except one IEEE
 * binary32 divide (edge ratio), exactly as in the reference.
 *
 * Border convention (the reference's streaming stages, e.g.
 * DetectKeypoints.cpp:968-991): every neighbourhood access outside the plane
 * reads 0 -- right/bottom by explicit zero fill, left because the trailing zero
 * columns of a row are the next row's leading window columns.  The top rows of
 * the reference's line buffers are uninitialised; they are defined as 0 here.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "oracle_tables.h"

#define ORC_MAX_KEYPOINTS 10000 /* SIFTParameters.hpp:22 */
#define ORC_DESC_SIZE 72        /* SIFTParameters.hpp:26 */
#define ORC_WINDOW 71           /* SIFTParameters.hpp:34 */
#define ORC_RADIUS 35           /* SIFT_Descriptor.cpp:28 */
#define ORC_EIGEN_TH 12         /* DetectKeypoints.hpp:18 */

typedef struct { uint16_t x, y; uint32_t contrast; } orc_keypoint; /* SIFTParameters.hpp:79-85 */
typedef struct { uint16_t ang, mag; } orc_grad;                    /* SIFTParameters.hpp:88-92 */

/* ------------------------------------------------------------------ tables */

/* Expand the unique octant to the full n x n Q15 table
 * (DetectKeypoints.cpp:756-801 after Mask_t quantisation). */
void orc_gauss_table(int which, int32_t *out)
{
    const unsigned short *oct = which == 1 ? ORC_G1_OCT : which == 2 ? ORC_G2_OCT : ORC_G3_OCT;
    int n = which == 1 ? ORC_G1_SIZE : which == 2 ? ORC_G2_SIZE : ORC_G3_SIZE;
    int o = n / 2;
    for (int r = 0; r < n; r++)
        for (int c = 0; c < n; c++) {
            int rr = r <= o ? r : n - 1 - r, cc = c <= o ? c : n - 1 - c;
            int a = rr < cc ? rr : cc, b = rr < cc ? cc : rr;
            /* row a of the octant starts after sum_{i<a} (o+1-i) entries */
            int base = a * (o + 1) - a * (a - 1) / 2;
            out[r * n + c] = oct[base + (b - a)];
        }
}

/* circle_mask.hpp:16-40 */
void orc_circle_mask(uint8_t out[625])
{
    for (int i = 0; i < 25; i++)
        for (int j = 0; j < 25; j++)
            out[i * 25 + j] = (j >= ORC_CIRCLE_LO[i] && j < ORC_CIRCLE_HI[i]) ? 1 : 0;
}

/* mask_all.hpp:8-78 (= union of the nine circles; tools/derive_tables.py checks) */
void orc_mask_all(uint8_t out[71 * 71])
{
    uint8_t cm[625];
    orc_circle_mask(cm);
    memset(out, 0, 71 * 71);
    for (int n = 0; n < 9; n++)
        for (int i = 0; i < 25; i++)
            for (int j = 0; j < 25; j++)
                if (cm[i * 25 + j])
                    out[(ORC_YSTART[n] + i) * 71 + ORC_XSTART[n] + j] = 1;
}

/* Descriptor geometry.  0 = Cam0, the only one the reference's code uses (window 71, radius 35,
 * 25x25 circle mask, XStart/YStart, MaskAll).  1 = Cam1: the reference only ships its tables
 * (DESC_WIN_SIZE_1 81, MASK_SIZE_1 29, SIFTParameters.hpp:46-55; CircleMaskCam1, XStartCam1/YStartCam1,
 * circle_mask.hpp:12-13, :43-71; MaskAll_81, mask_all.hpp:161-241) and an empty "CAM1" section
 * (SIFT_Descriptor.cpp:496), so its semantics are DEFINED here as the Cam0 loops below with those
 * constants substituted -- in particular the orientation loop still runs over 2 * radius columns (:70),
 * i.e. 80 of the 81, although MaskAll_81's last column is not empty.  Parity unpinned by construction. */
typedef struct {
    int window, radius, msize;
    const unsigned char *xs, *ys, *lo, *hi;
} orc_geometry;
static const orc_geometry ORC_GEO[2] = {
    {71, 35, 25, ORC_XSTART, ORC_YSTART, ORC_CIRCLE_LO, ORC_CIRCLE_HI},
    {81, 40, 29, ORC_XSTART_CAM1, ORC_YSTART_CAM1, ORC_CIRCLE_CAM1_LO, ORC_CIRCLE_CAM1_HI},
};

/* circle mask of a geometry as msize x msize bytes */
static void geo_circle_mask(const orc_geometry *g, uint8_t *out)
{
    for (int i = 0; i < g->msize; i++)
        for (int j = 0; j < g->msize; j++)
            out[i * g->msize + j] = (j >= g->lo[i] && j < g->hi[i]) ? 1 : 0;
}

/* MaskAll (geometry 0, mask_all.hpp:8-78) / MaskAll_81 (geometry 1, mask_all.hpp:161-241) as window x window bytes */
void orc_mask_all_geo(int geometry, uint8_t *out)
{
    if (geometry == 0) { orc_mask_all(out); return; }
    for (int r = 0; r < 81; r++)
        for (int c = 0; c < 81; c++)
            out[r * 81 + c] = (uint8_t)((ORC_MASKALL81[r][c >> 5] >> (c & 31)) & 1u);
}

void orc_circle_mask_geo(int geometry, uint8_t *out)
{
    geo_circle_mask(&ORC_GEO[geometry != 0], out);
}

/* ------------------------------------------------------------------- blur */

/* GaussianBlurHLS + FilterKernelOperator1/2/3 + RoundToInt
 * (DetectKeypoints.cpp:925-1043, :832-912, :810-827).
 * S = sum P * Kq (exact in Fixed_t, 15 fraction bits); RoundToInt adds 0.5 and
 * truncates (operand >= 0); the result is clamped to 255 (:848-851). */
static void blur_one(const uint8_t *pad, int W, int H, int PW, int which, uint8_t *out)
{
    int n = which == 1 ? 5 : which == 2 ? 11 : 19, o = n / 2;
    int32_t k[19 * 19];
    orc_gauss_table(which, k);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int32_t s = 0;
            for (int r = 0; r < n; r++) {
                const uint8_t *p = pad + (size_t)(y + 9 - o + r) * PW + (x + 9 - o);
                const int32_t *kr = k + r * n;
                for (int c = 0; c < n; c++) s += (int32_t)p[c] * kr[c];
            }
            int32_t v = (s + 16384) >> 15;
            out[(size_t)y * W + x] = (uint8_t)(v < 255 ? v : 255);
        }
}

void orc_blur(const uint8_t *img, int W, int H, uint8_t *g1, uint8_t *g2, uint8_t *g3)
{
    int PW = W + 18, PH = H + 18;
    uint8_t *pad = (uint8_t *)calloc((size_t)PW * PH, 1);
    for (int y = 0; y < H; y++) memcpy(pad + (size_t)(y + 9) * PW + 9, img + (size_t)y * W, W);
    blur_one(pad, W, H, PW, 1, g1);
    blur_one(pad, W, H, PW, 2, g2);
    blur_one(pad, W, H, PW, 3, g3);
    free(pad);
}

/* -------------------------------------------------------------------- DoG */

/* build_DOG_hls (DetectKeypoints.cpp:511-559, :543-545) */
void orc_dog(const uint8_t *src, const uint8_t *g1, const uint8_t *g2, const uint8_t *g3,
             int W, int H, int16_t *d1, int16_t *d2, int16_t *d3)
{
    size_t n = (size_t)W * H;
    for (size_t i = 0; i < n; i++) {
        d1[i] = (int16_t)((int)g1[i] - (int)src[i]);
        d2[i] = (int16_t)((int)g2[i] - (int)g1[i]);
        d3[i] = (int16_t)((int)g3[i] - (int)g2[i]);
    }
}

/* ---------------------------------------------------------------- extrema */

static inline int at16(const int16_t *p, int W, int H, int y, int x)
{
    return (y < 0 || y >= H || x < 0 || x >= W) ? 0 : p[(size_t)y * W + x];
}

/* IsLocalExtremum (DetectKeypoints.cpp:234-308): strictly above (or strictly
 * below) all 26 neighbours in the 3x3x3 block; the centres of DoG1 and DoG3
 * are included. */
static int is_local_extremum(const int16_t *d1, const int16_t *d2, const int16_t *d3,
                             int W, int H, int y, int x)
{
    int c = at16(d2, W, H, y, x);
    int is_max = 1, is_min = 1;
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            int a = at16(d1, W, H, y + dy, x + dx);
            int b = at16(d3, W, H, y + dy, x + dx);
            is_max &= (c > a) & (c > b);
            is_min &= (c < a) & (c < b);
            if (dy || dx) {
                int m = at16(d2, W, H, y + dy, x + dx);
                is_max &= (c > m);
                is_min &= (c < m);
            }
        }
    return is_max | is_min;
}

/* IsOnEdge (DetectKeypoints.cpp:314-352) followed by float -> Fixed_t
 * (ap_fixed<32,17,AP_RND_ZERO>, SIFTParameters.hpp:95): returns the Q15 integer
 * of the eigen ratio, or -32768 (the Q15 image of -1) when rejected. */
int32_t orc_edge_ratio_q15(int L, int R, int U, int Dn, int UL, int UR, int DL, int DR, int c)
{
    int fxx = L + R - 2 * c;
    int fyy = U + Dn - 2 * c;
    int fxy = UL + DR - DL - UR;
    int trace = fxx + fyy;
    int deter = fxx * fyy - fxy * fxy;
    if (deter == 0) return -32768;
    volatile float num = (float)(trace * trace), den = (float)deter;
    volatile float ratio = num / den;              /* IEEE binary32 divide (:338) */
    float r = ratio;
    if (r < 0 || r > (float)ORC_EIGEN_TH) return -32768;   /* :342 (-0.0f passes) */
    /* float -> Q15, nearest, ties toward zero; r >= 0 here.  r*32768 is exact
     * in double. */
    double v = (double)r * 32768.0;
    double fl = floor(v);
    double frac = v - fl;
    return (int32_t)fl + (frac > 0.5 ? 1 : 0);
}

/* detect_extrema_hls (DetectKeypoints.cpp:44-223): kp1 = |DoG2| where the
 * pixel is a strict 3x3x3 extremum, |DoG2| >= contrastTh (:194) and the edge
 * ratio is valid (:200); eratio is written for every pixel (:209). */
void orc_extrema(const int16_t *d1, const int16_t *d2, const int16_t *d3, int W, int H,
                 int contrastTh, uint16_t *kp1, int32_t *eratio)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int c = at16(d2, W, H, y, x);
            int ac = c < 0 ? -c : c;                                  /* ABS :1257-1267 */
            int ext = is_local_extremum(d1, d2, d3, W, H, y, x);
            int low = ac < contrastTh;
            int32_t e = orc_edge_ratio_q15(
                at16(d2, W, H, y, x - 1), at16(d2, W, H, y, x + 1),
                at16(d2, W, H, y - 1, x), at16(d2, W, H, y + 1, x),
                at16(d2, W, H, y - 1, x - 1), at16(d2, W, H, y - 1, x + 1),
                at16(d2, W, H, y + 1, x - 1), at16(d2, W, H, y + 1, x + 1), c);
            size_t i = (size_t)y * W + x;
            eratio[i] = e;
            kp1[i] = (uint16_t)((!ext || e == -32768 || low) ? 0 : ac);
        }
}

/* ------------------------------------------------------------------- cull */

/* culling_hls + IsBestKeypoint (DetectKeypoints.cpp:567-750): a candidate
 * survives iff its ratio is strictly smaller than that of every OTHER
 * pre-cull candidate (kp1 != 0) in the 7x7 window (:728-731, :684). */
void orc_cull(const uint16_t *kp1, const int32_t *eratio, int W, int H, uint16_t *kp2)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t i = (size_t)y * W + x;
            uint16_t v = kp1[i];
            if (v) {
                int32_t ec = eratio[i];
                int best = 1;
                for (int dy = -3; dy <= 3 && best; dy++)
                    for (int dx = -3; dx <= 3; dx++) {
                        int yy = y + dy, xx = x + dx;
                        if ((!dy && !dx) || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                        size_t j = (size_t)yy * W + xx;
                        if (kp1[j] != 0 && !(ec < eratio[j])) { best = 0; break; }
                    }
                if (!best) v = 0;
            }
            kp2[i] = v;
        }
}

/* ---------------------------------------------------------------- combine */

/* combine_keypoints_hls (DetectKeypoints.cpp:372-439): border filter (:393),
 * raster scan, first MAX_KEYPOINTS kept (:409), zero padding (:428-436).
 * Returns the uncapped survivor count (nKeypointsTotal). */
int orc_combine(const uint16_t *kp2, int W, int H, int max_kp, orc_keypoint *out)
{
    int n = 0, total = 0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            uint16_t v = 0;
            if ((y > ORC_WINDOW && x > ORC_WINDOW) && (y < H - ORC_WINDOW && x < W - ORC_WINDOW))
                v = kp2[(size_t)y * W + x];
            if (v > 0) total++;
            if (v > 0 && n < max_kp) {
                out[n].x = (uint16_t)x; out[n].y = (uint16_t)y; out[n].contrast = v;
                n++;
            }
        }
    for (int i = n; i < max_kp; i++) { out[i].x = 0; out[i].y = 0; out[i].contrast = 0; }
    return total;
}

/* --------------------------------------------------------------- gradient */

static inline int at8(const uint8_t *p, int W, int H, int y, int x)
{
    return (y < 0 || y >= H || x < 0 || x >= W) ? 0 : p[(size_t)y * W + x];
}

/* ComputeGradientsHLS + ComputeGradientKernel (DetectKeypoints.cpp:1273-1363,
 * :1186-1225).  mag = (int)sqrtf(gx^2+gy^2) which equals the integer square
 * root on this range; the direction code follows :1206-1213. */
void orc_gradient(const uint8_t *g2, int W, int H, orc_grad *out)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int gx = at8(g2, W, H, y, x + 1) - at8(g2, W, H, y, x - 1);
            int gy = at8(g2, W, H, y + 1, x) - at8(g2, W, H, y - 1, x);
            int s = gx * gx + gy * gy;
            int m = (int)sqrt((double)s);
            while (m * m > s) m--;
            while ((m + 1) * (m + 1) <= s) m++;
            int ax = gx < 0 ? -gx : gx, ay = gy < 0 ? -gy : gy;
            int sx = (gx > 0) - (gx < 0), sy = (gy > 0) - (gy < 0);
            int ang = (sx > 0 && sy > 0 && ax >  ay) * 1 + (sx > 0 && sy > 0 && ax <= ay) * 2 +
                      (sx < 0 && sy > 0 && ax <  ay) * 3 + (sx < 0 && sy > 0 && ax >= ay) * 4 +
                      (sx < 0 && sy < 0 && ax >  ay) * 5 + (sx < 0 && sy < 0 && ax <= ay) * 6 +
                      (sx > 0 && sy < 0 && ax <  ay) * 7 + (sx > 0 && sy < 0 && ax >= ay) * 8;
            out[(size_t)y * W + x].ang = (uint16_t)ang;
            out[(size_t)y * W + x].mag = (uint16_t)m;
        }
}

/* ------------------------------------------------------------ sift_detect */

/* Intermediate planes kept for stage-level parity tests. */
typedef struct {
    uint8_t *g1, *g2, *g3;
    int16_t *d1, *d2, *d3;
    uint16_t *kp1, *kp2;
    int32_t *eratio;
} orc_planes;

/* sift_detect (DetectKeypoints.cpp:1073-1175).  Any of the plane pointers in
 * `keep` may be NULL; non-NULL ones receive a copy of that stage's output.
 * Returns the uncapped keypoint count. */
int orc_sift_detect_planes(const uint8_t *img, int W, int H, orc_grad *grad, orc_keypoint *kps,
                           int contrastTh, int max_kp, const orc_planes *keep)
{
    size_t n = (size_t)W * H;
    uint8_t *g1 = malloc(n), *g2 = malloc(n), *g3 = malloc(n);
    int16_t *d1 = malloc(2 * n), *d2 = malloc(2 * n), *d3 = malloc(2 * n);
    uint16_t *kp1 = malloc(2 * n), *kp2 = malloc(2 * n);
    int32_t *er = malloc(4 * n);
    orc_blur(img, W, H, g1, g2, g3);                 /* :1129 */
    orc_dog(img, g1, g2, g3, W, H, d1, d2, d3);      /* :1131 */
    orc_extrema(d1, d2, d3, W, H, contrastTh, kp1, er); /* :1133 */
    orc_cull(kp1, er, W, H, kp2);                    /* :1135 */
    int total = orc_combine(kp2, W, H, max_kp, kps); /* :1138 */
    if (grad) orc_gradient(g2, W, H, grad);          /* :1140 */
    if (keep) {
        if (keep->g1) memcpy(keep->g1, g1, n);
        if (keep->g2) memcpy(keep->g2, g2, n);
        if (keep->g3) memcpy(keep->g3, g3, n);
        if (keep->d1) memcpy(keep->d1, d1, 2 * n);
        if (keep->d2) memcpy(keep->d2, d2, 2 * n);
        if (keep->d3) memcpy(keep->d3, d3, 2 * n);
        if (keep->kp1) memcpy(keep->kp1, kp1, 2 * n);
        if (keep->kp2) memcpy(keep->kp2, kp2, 2 * n);
        if (keep->eratio) memcpy(keep->eratio, er, 4 * n);
    }
    free(g1); free(g2); free(g3); free(d1); free(d2); free(d3); free(kp1); free(kp2); free(er);
    return total;
}

int orc_sift_detect(const uint8_t *img, int W, int H, orc_grad *grad, orc_keypoint *kps, int contrastTh)
{
    return orc_sift_detect_planes(img, W, H, grad, kps, contrastTh, ORC_MAX_KEYPOINTS, NULL);
}

/* ---------------------------------------------- orientation + descriptor */

/* The reference addresses the gradient plane linearly,
 * pGrad[y*WIDTH + ptX - radius + col] (SIFT_Descriptor.cpp:64, :204).  For the
 * keypoints sift_detect emits (>71 px from every edge) that is an ordinary 2-D
 * window.  For caller-supplied keypoints nearer the edge the reference would
 * read outside the array (undefined); here and in the HIP path such reads are
 * defined as {0,0} while in-array linear indices wrap across rows as they do in
 * the reference. */
static inline orc_grad grad_lin(const orc_grad *g, int W, int H, long idx)
{
    orc_grad z = {0, 0};
    return (idx < 0 || idx >= (long)W * H) ? z : g[idx];
}

/* compute_main_orientation + ComputeMainOrientation
 * (SIFT_Descriptor.cpp:18-103, :107-130): window rows x 2 * radius columns (:70) -- 71 x 70 for Cam0. */
void orc_main_orientation_geo(const orc_grad *grad, int W, int H, const orc_keypoint *kps, int n, uint8_t *ori, int geometry)
{
    const orc_geometry *g = &ORC_GEO[geometry != 0];
    const int WS = g->window, R = g->radius;
    uint8_t mask[81 * 81];
    orc_mask_all_geo(geometry != 0, mask);
    for (int k = 0; k < n; k++) {
        int px = kps[k].x, py = kps[k].y;
        uint8_t m = 0;
        if (px > 0 && py > 0) {                                     /* :52 */
            int hist[9] = {0};
            for (int y = py - R; y <= py + R; y++) {
                int line = y - py + R;
                for (int x = 0; x < 2 * R; x++) {                   /* :70 */
                    orc_grad a = grad_lin(grad, W, H, (long)y * W + px - R + x);
                    if (a.ang < 9) hist[a.ang] += (int)(int16_t)a.mag * mask[line * WS + x];
                }
            }
            int best = 0;                                            /* :110-126 */
            for (int i = 1; i < 9; i++)
                if (hist[i] > best) { best = hist[i]; m = (uint8_t)i; }
        }
        ori[k] = m;                                                  /* :100 */
    }
}

void orc_main_orientation(const orc_grad *grad, int W, int H, const orc_keypoint *kps, int n, uint8_t *ori)
{
    orc_main_orientation_geo(grad, W, H, kps, n, ori, 0);
}

/* compute_descriptor = compute_histograms_hls + reorder_histograms_hls
 * (SIFT_Descriptor.cpp:419-478, :151-278, :295-398).  Keypoints with x==0 or
 * y==0 (the padding) have an uninitialised descriptor in the reference
 * (:194-196, :266-274); it is defined as all-zero here. */
void orc_descriptor_geo(const orc_grad *grad, int W, int H, const orc_keypoint *kps, const uint8_t *ori,
                        int n, uint32_t *desc, int geometry)
{
    const orc_geometry *g = &ORC_GEO[geometry != 0];
    const int MS = g->msize, R = g->radius;
    uint8_t cm[29 * 29];
    geo_circle_mask(g, cm);
    for (int k = 0; k < n; k++) {
        int px = kps[k].x, py = kps[k].y;
        uint32_t hin[9][9];
        memset(hin, 0, sizeof hin);
        uint32_t *out = desc + (size_t)k * ORC_DESC_SIZE;
        if (!(px > 0 && py > 0)) { memset(out, 0, ORC_DESC_SIZE * 4); continue; }   /* :196 */
        for (int i = 0; i < MS; i++)                                 /* :211-234 */
            for (int j = 0; j < MS; j++)
                for (int c = 0; c < 9; c++) {
                    int wy = g->ys[c] + i, wx = g->xs[c] + j;
                    orc_grad a = grad_lin(grad, W, H, (long)(py - R + wy) * W + px - R + wx);
                    if (a.ang < 9) hin[c][a.ang] += (uint32_t)cm[i * MS + j] * a.mag;
                }
        /* :245-252 keep codes 1..8 as bins 0..7; :346-380 rotate by the main
         * orientation with 3-bit wrap: out[n][(b - m) & 7] = in[(m + n) & 7][b] */
        unsigned m = ori[k];
        for (int c = 0; c < 9; c++) {
            int srcc = c < 8 ? (int)((m + (unsigned)c) & 7u) : 8;
            for (int b = 0; b < 8; b++)
                out[c * 8 + (int)(((unsigned)b - m) & 7u)] = hin[srcc][b + 1];
        }
    }
}

void orc_descriptor(const orc_grad *grad, int W, int H, const orc_keypoint *kps, const uint8_t *ori,
                    int n, uint32_t *desc)
{
    orc_descriptor_geo(grad, W, H, kps, ori, n, desc, 0);
}

/* sift_detect_and_compute (HW_SIFT.cpp:16-50) */
int orc_detect_and_compute_geo(const uint8_t *img, int W, int H, orc_keypoint *kps, uint32_t *desc, int contrastTh, int geometry)
{
    orc_grad *grad = malloc((size_t)W * H * sizeof(orc_grad));
    uint8_t *ori = malloc(ORC_MAX_KEYPOINTS);
    int total = orc_sift_detect(img, W, H, grad, kps, contrastTh);                 /* :29 */
    orc_main_orientation_geo(grad, W, H, kps, ORC_MAX_KEYPOINTS, ori, geometry);   /* :40 */
    orc_descriptor_geo(grad, W, H, kps, ori, ORC_MAX_KEYPOINTS, desc, geometry);   /* :41 */
    free(grad); free(ori);
    return total;
}

int orc_detect_and_compute(const uint8_t *img, int W, int H, orc_keypoint *kps, uint32_t *desc, int contrastTh)
{
    return orc_detect_and_compute_geo(img, W, H, kps, desc, contrastTh, 0);
}

/* ------------------------------------------------------- synthetic frames */

/* SURVEY.md 8(d) generator (not reference code): integer-only textured frame. */
static inline uint32_t h32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

void orc_synth_frame(uint32_t seed, int W, int H, uint8_t *out)
{
    static const int S[5] = {5, 4, 3, 2, 0}, Wt[5] = {4, 3, 2, 2, 1};
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            uint32_t acc = 0;
            for (int i = 0; i < 5; i++) {
                uint32_t s = (uint32_t)S[i];
                uint32_t n = h32(seed ^ h32(((uint32_t)y >> s) * 0x9E3779B1U + ((uint32_t)x >> s)) ^ (s * 0x85EBCA6BU)) & 0xFF;
                acc += (uint32_t)Wt[i] * n;
            }
            out[(size_t)y * W + x] = (uint8_t)(acc / 12);
        }
}

/* 64-bit FNV-1a, used by the tests to compare large outputs with the hashes in
 * tests/golden/synth_expected.json (not reference code). */
uint64_t orc_fnv1a(const uint8_t *p, size_t n)
{
    uint64_t h = 0xCBF29CE484222325ULL;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001B3ULL;
    return h;
}

/* ---- post-processing of main.cpp:82-130 --------------------------------------------------
 * std::sort(pKeypoints, pKeypoints + MAX_KEYPOINTS, CompareContrast) (:82, CompareContrast =
 * contrast descending, :8-11), memcpy of the first MAX_FINAL_KEYPOINTS (:84), and the count
 * of those with x > 0 && y > 0 (:112-129).  std::sort is not stable, so the reference leaves the
 * order among equal contrasts to the library; this restatement (and the GPU path) orders
 * ties by slot, i.e. by the raster order sift_detect emits.  index[r] = slot of final[r]. */
typedef struct { uint32_t contrast, slot; } sel_key;
static int sel_cmp(const void *a, const void *b)
{
    const sel_key *p = (const sel_key *)a, *q = (const sel_key *)b;
    if (p->contrast != q->contrast) return p->contrast > q->contrast ? -1 : 1;
    return p->slot < q->slot ? -1 : (p->slot > q->slot);
}
int orc_select_top(const orc_keypoint *kps, int n, int nfinal, orc_keypoint *final, uint32_t *index)
{
    sel_key *k = (sel_key *)malloc((size_t)n * sizeof(sel_key));
    int nvalid = 0;
    for (int i = 0; i < n; i++) { k[i].contrast = kps[i].contrast; k[i].slot = (uint32_t)i; }
    qsort(k, (size_t)n, sizeof(sel_key), sel_cmp);
    for (int r = 0; r < nfinal && r < n; r++) {
        final[r] = kps[k[r].slot];
        if (index) index[r] = k[r].slot;
        if (final[r].x > 0 && final[r].y > 0) nvalid++;
    }
    free(k);
    return nvalid;
}

/* ---- ingest of main.cpp:51-58 for colour frames ---------------------------------------------
 * cv::imread(path, CV_LOAD_IMAGE_GRAYSCALE) on a 24-bit BMP: OpenCV (not in the reference tree, not
 * installed here; imgcodecs utils.cpp, icvCvt_BGR2Gray_8u_C3C1R) uses cR = 4899, cG = 9617,
 * cB = 16384 - cR - cG = 1868 and descale(x, 14) = (x + 8192) >> 14.  Unpinned restatement. */
void orc_bgr2gray(const uint8_t *bgr, int W, int H, long pitch, int bottom_up, uint8_t *gray)
{
    for (int y = 0; y < H; y++) {
        const uint8_t *row = bgr + (size_t)(bottom_up ? H - 1 - y : y) * (size_t)pitch;
        for (int x = 0; x < W; x++)
            gray[(size_t)y * W + x] = (uint8_t)((row[3 * x] * 1868 + row[3 * x + 1] * 9617 + row[3 * x + 2] * 4899 + 8192) >> 14);
    }
}

/* ---- matcher for DMatch_t (SIFTParameters.hpp:122-128) ------------------------------------------
 * The reference has the type only.  Defined semantics (see include/sift_hip.h): per query row the train
 * row of smallest L1 distance, ties to the smaller index; {0xffffffff, q, -1} without train rows. */
typedef struct { uint32_t distance; int16_t queryIdx, trainIdx; } orc_dmatch;
void orc_match(const uint32_t *query, int nquery, const uint32_t *train, int ntrain, orc_dmatch *out)
{
    for (int q = 0; q < nquery; q++) {
        uint64_t best = UINT64_MAX;
        int bt = -1;
        for (int t = 0; t < ntrain; t++) {
            uint64_t d = 0;
            for (int j = 0; j < ORC_DESC_SIZE; j++) {
                uint32_t a = query[(size_t)q * ORC_DESC_SIZE + j], b = train[(size_t)t * ORC_DESC_SIZE + j];
                d += a > b ? a - b : b - a;
            }
            if (d < best) { best = d; bt = t; }
        }
        out[q].distance = bt < 0 ? 0xffffffffu : (uint32_t)best;
        out[q].queryIdx = (int16_t)q;
        out[q].trainIdx = (int16_t)bt;
    }
}
