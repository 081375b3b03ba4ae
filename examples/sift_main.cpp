// sift_main.cpp -- the reference's host flow (main.cpp:37-210) on this build: read an 8-bit gray
// frame (binary PGM, an 8- or 24-bit BMP like the reference's Data/Bot/bot_N.bmp, or a synthetic
// frame), time one sift_detect_and_compute call, sort the
// keypoints by contrast, keep MAX_FINAL_KEYPOINTS, write keypoints and descriptors as text.
//   sift_main <in.pgm | in.bmp | synth:W:H:seed> [out_prefix] [contrastTh=15]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../include/HW_SIFT.hpp"

static uint32_t h32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// Uncompressed BMP, 8 bits (palette) or 24 bits per pixel.  8-bit frames are converted here through
// the palette; 24-bit frames keep their BGR rows in `bgr` (pitch, bottom-up flag) and are converted on
// the device by sift_gray_from_bgr once the frame size is known.
static uint32_t rd32(const uint8_t *p) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }
static bool load_bmp(FILE *f, std::vector<uint8_t> &img, std::vector<uint8_t> &bgr, long &pitch, int &bottom_up, int &W, int &H)
{
    uint8_t h[54];
    if (fread(h, 1, 54, f) != 54 || h[0] != 'B' || h[1] != 'M') return false;
    const uint32_t off = rd32(h + 10), hdr = rd32(h + 14), comp = rd32(h + 30);
    const int w = (int)rd32(h + 18), hh = (int)rd32(h + 22), bpp = h[28] | h[29] << 8;
    if (hdr < 40 || comp != 0 || w <= 0 || hh == 0 || (bpp != 8 && bpp != 24)) return false;
    W = w; H = hh < 0 ? -hh : hh; bottom_up = hh > 0;
    pitch = ((long)W * (bpp / 8) + 3) / 4 * 4;
    uint8_t pal[256][4] = {};
    if (bpp == 8) {
        uint32_t ncol = rd32(h + 46);
        if (ncol == 0 || ncol > 256) ncol = 256;
        if (fseek(f, 14 + (long)hdr, SEEK_SET) != 0 || fread(pal, 4, ncol, f) != ncol) return false;
    }
    std::vector<uint8_t> raw((size_t)pitch * H);
    if (fseek(f, (long)off, SEEK_SET) != 0 || fread(raw.data(), 1, raw.size(), f) != raw.size()) return false;
    if (bpp == 24) { bgr.swap(raw); return true; }
    img.resize((size_t)W * H);
    for (int y = 0; y < H; y++) {
        const uint8_t *row = raw.data() + (size_t)(bottom_up ? H - 1 - y : y) * pitch;
        for (int x = 0; x < W; x++) {
            const uint8_t *c = pal[row[x]];                        // B, G, R, reserved
            img[(size_t)y * W + x] = (uint8_t)((c[0] * 1868 + c[1] * 9617 + c[2] * 4899 + 8192) >> 14);
        }
    }
    return true;
}

static bool load(const char *arg, std::vector<uint8_t> &img, std::vector<uint8_t> &bgr, long &pitch, int &bottom_up, int &W, int &H)
{
    unsigned w, h, seed;
    if (sscanf(arg, "synth:%u:%u:%u", &w, &h, &seed) == 3) {       // SURVEY.md 8(d) generator
        static const int S[5] = {5, 4, 3, 2, 0}, Wt[5] = {4, 3, 2, 2, 1};
        W = (int)w; H = (int)h; img.resize((size_t)w * h);
        for (unsigned y = 0; y < h; y++)
            for (unsigned x = 0; x < w; x++) {
                uint32_t acc = 0;
                for (int i = 0; i < 5; i++)
                    acc += Wt[i] * (h32(seed ^ h32((y >> S[i]) * 0x9E3779B1U + (x >> S[i])) ^ (S[i] * 0x85EBCA6BU)) & 0xFF);
                img[(size_t)y * w + x] = (uint8_t)(acc / 12);
            }
        return true;
    }
    FILE *f = fopen(arg, "rb");
    if (!f) return false;
    const size_t n = strlen(arg);
    if (n > 4 && (!strcmp(arg + n - 4, ".bmp") || !strcmp(arg + n - 4, ".BMP"))) {
        const bool ok = load_bmp(f, img, bgr, pitch, bottom_up, W, H);
        fclose(f);
        return ok;
    }
    char magic[3] = {0};
    int maxv = 0;
    bool ok = fscanf(f, "%2s %d %d %d", magic, &W, &H, &maxv) == 4 && !strcmp(magic, "P5") && maxv == 255;
    if (ok) {
        fgetc(f);
        img.resize((size_t)W * H);
        ok = fread(img.data(), 1, img.size(), f) == img.size();
    }
    fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s <in.pgm | synth:W:H:seed> [out_prefix] [contrastTh]\n", argv[0]); return 2; }
    std::vector<uint8_t> img, bgr;
    long pitch = 0;
    int W = 0, H = 0, bottom_up = 0;
    if (!load(argv[1], img, bgr, pitch, bottom_up, W, H)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    if (W % 4) { fprintf(stderr, "frame width must be a multiple of 4\n"); return 1; }
    const int th = argc > 3 ? atoi(argv[3]) : 15;                  // main.cpp:73
    sift_set_frame_size(W, H);

    Pixel_t *src = (Pixel_t *)sift_alloc((size_t)W * H);           // main.cpp:18-32
    Keypoint_t *kps = (Keypoint_t *)sift_alloc(MAX_KEYPOINTS * sizeof(Keypoint_t));
    Desc_t *desc = (Desc_t *)sift_alloc((size_t)MAX_KEYPOINTS * DESC_SIZE * sizeof(Desc_t));
    if (!src || !kps || !desc) { fprintf(stderr, "pinned allocation failed\n"); return 1; }
    if (!bgr.empty()) sift_gray_from_bgr(bgr.data(), pitch, bottom_up, src);   // main.cpp:51: colour file read as gray
    else memcpy(src, img.data(), img.size());                      // main.cpp:55-58

    sift_detect_and_compute(src, kps, desc, th);                   // warm-up: creates the context
    auto t0 = std::chrono::steady_clock::now();                    // main.cpp:63-77
    sift_detect_and_compute(src, kps, desc, th);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

    int n = 0;
    while (n < MAX_KEYPOINTS && kps[n].contrast) n++;
    printf("%dx%d: %d keypoints, %.3f ms (host buffers in, host buffers out)\n", W, H, n, ms);

    // main.cpp:82-130: the MAX_FINAL_KEYPOINTS strongest keypoints in descending order of contrast,
    // selected on the device.  The reference's std::sort is unstable; ties keep raster order here.
    // Descriptors stay in raster order (main.cpp:82 does not permute them), so each final keypoint
    // carries its slot.  The same selection on the host checks the device result.
    std::vector<Keypoint_t> fin(MAX_FINAL_KEYPOINTS);
    std::vector<uint32_t> slot(MAX_FINAL_KEYPOINTS);
    const int nkp = sift_select_final_keypoints(kps, fin.data(), slot.data());
    printf("number of keypoints = %d\n", nkp);                    // main.cpp:130
    std::vector<int> order(MAX_KEYPOINTS);
    for (int i = 0; i < MAX_KEYPOINTS; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return kps[a].contrast > kps[b].contrast; });
    for (int i = 0; i < MAX_FINAL_KEYPOINTS; i++)
        if ((int)slot[i] != order[i] || memcmp(&fin[i], &kps[order[i]], sizeof(Keypoint_t)) != 0) {
            fprintf(stderr, "device selection differs from the host sort at rank %d\n", i);
            return 1;
        }
    if (argc > 2) {
        std::string p = argv[2];
        FILE *f = fopen((p + "_keypoints.txt").c_str(), "w");
        FILE *g = fopen((p + "_descriptors.txt").c_str(), "w");
        if (!f || !g) { fprintf(stderr, "cannot write outputs\n"); return 1; }
        for (int i = 0; i < nkp; i++) {
            const int k = (int)slot[i];
            fprintf(f, "%d %u %u %u\n", k, (unsigned)fin[i].x, (unsigned)fin[i].y, (unsigned)fin[i].contrast);
            for (int j = 0; j < DESC_SIZE; j++) fprintf(g, "%u%c", (unsigned)desc[(size_t)k * DESC_SIZE + j], j + 1 < DESC_SIZE ? ' ' : '\n');
        }
        fclose(f); fclose(g);
    }
    sift_free(src); sift_free(kps); sift_free(desc);
    sift_shutdown();
    return 0;
}
