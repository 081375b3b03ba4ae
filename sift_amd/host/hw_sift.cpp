// hw_sift.cpp -- the reference's four entry points (HW_SIFT.cpp:16-50, DetectKeypoints.cpp:1073,
// SIFT_Descriptor.cpp:18, :419) as thin C++ wrappers over the C ABI (include/sift_hip.h).
// Plain C++ (g++); no HIP types here.  One cached context per process replaces the per-call
// sds_alloc/sds_free scratch of HW_SIFT.cpp:23-24,44-45.
#include <stdio.h>
#include <stdlib.h>
#include <mutex>

#include "../../include/HW_SIFT.hpp"
#include "../../include/DetectKeypoints.hpp"
#include "../../include/SIFT_Descriptor.hpp"
#include "../../include/sift_hip.h"

namespace {

std::mutex g_mu;
sift_hip_ctx *g_ctx = nullptr;
int g_w = WIDTH, g_h = HEIGHT, g_dev = 0, g_geo = SIFT_HIP_GEOMETRY_CAM0;

[[noreturn]] void die(const char *what)
{
    fprintf(stderr, "hw_sift: %s failed: %s\n", what, sift_hip_last_error());
    abort();
}

sift_hip_ctx *ctx()
{
    if (!g_ctx) {
        if (sift_hip_create_geometry(&g_ctx, g_dev, g_w, g_h, MAX_KEYPOINTS, 1, g_geo) != SIFT_HIP_OK) die("sift_hip_create");
        static bool registered = false;              // once per process, however often the context is recreated
        if (!registered) { atexit(sift_shutdown); registered = true; }
    }
    return g_ctx;
}

static_assert(sizeof(Keypoint_t) == sizeof(sift_keypoint) && sizeof(Grad_t) == sizeof(sift_grad) && sizeof(DMatch_t) == sizeof(sift_dmatch), "C ABI layouts");

}  // namespace

void sift_set_frame_size(int width, int height)
{
    std::lock_guard<std::mutex> l(g_mu);
    if (width != g_w || height != g_h) {
        if (g_ctx) { sift_hip_destroy(g_ctx); g_ctx = nullptr; }
        g_w = width; g_h = height;
    }
}

void sift_set_device(int device)
{
    std::lock_guard<std::mutex> l(g_mu);
    if (device != g_dev) {
        if (g_ctx) { sift_hip_destroy(g_ctx); g_ctx = nullptr; }
        g_dev = device;
    }
}

void sift_set_geometry(int geometry)
{
    std::lock_guard<std::mutex> l(g_mu);
    if (geometry != g_geo) {
        if (g_ctx) { sift_hip_destroy(g_ctx); g_ctx = nullptr; }
        g_geo = geometry;
    }
}

void sift_shutdown(void)
{
    std::lock_guard<std::mutex> l(g_mu);
    if (g_ctx) { sift_hip_destroy(g_ctx); g_ctx = nullptr; }
}

void *sift_alloc(unsigned long bytes) { return sift_hip_host_alloc(bytes); }
void sift_free(void *p) { sift_hip_host_free(p); }

void sift_detect_and_compute(Pixel_t *pSrcImageCam0, Keypoint_t *pKeypoints, Desc_t *pDesc, int contrastTh)
{
    std::lock_guard<std::mutex> l(g_mu);
    // HW_SIFT.cpp:29-41: detect -> main orientation -> descriptor, here one device pass
    if (sift_hip_detect_and_compute(ctx(), pSrcImageCam0, (sift_keypoint *)pKeypoints, pDesc, contrastTh) != SIFT_HIP_OK)
        die("sift_detect_and_compute");
}

void sift_gray_from_bgr(const uint8_t *pBgr, long pitch, int bottomUp, Pixel_t *pGray)
{
    std::lock_guard<std::mutex> l(g_mu);
    if (sift_hip_gray_from_bgr(ctx(), pBgr, pitch, bottomUp, pGray) != SIFT_HIP_OK) die("sift_gray_from_bgr");
}

int sift_select_final_keypoints(const Keypoint_t *pKeypoints, Keypoint_t *pFinalKeypoints, uint32_t *pSlot)
{
    std::lock_guard<std::mutex> l(g_mu);
    uint32_t nkp = 0;
    if (sift_hip_select_top(ctx(), (const sift_keypoint *)pKeypoints, MAX_FINAL_KEYPOINTS, (sift_keypoint *)pFinalKeypoints,
                            pSlot, &nkp) != SIFT_HIP_OK)
        die("sift_select_final_keypoints");
    return (int)nkp;
}

void sift_detect(Pixel_t *pSrcImage, Grad_t *pGradients, Keypoint_t *pKeypoints, int contrastTh)
{
    std::lock_guard<std::mutex> l(g_mu);
    if (sift_hip_detect(ctx(), pSrcImage, (sift_grad *)pGradients, (sift_keypoint *)pKeypoints, contrastTh) != SIFT_HIP_OK)
        die("sift_detect");
}

void compute_main_orientation(Grad_t *pGrad, Keypoint_t *pKeypoint, uint8_t *pMainOrientation)
{
    std::lock_guard<std::mutex> l(g_mu);
    if (sift_hip_main_orientation(ctx(), (const sift_grad *)pGrad, (const sift_keypoint *)pKeypoint, pMainOrientation) != SIFT_HIP_OK)
        die("compute_main_orientation");
}

void compute_descriptor(Grad_t *pGrad, Keypoint_t *pKeypoint, uint8_t *pMainOrientation, Desc_t *pDescriptorOut)
{
    std::lock_guard<std::mutex> l(g_mu);
    if (sift_hip_descriptor(ctx(), (const sift_grad *)pGrad, (const sift_keypoint *)pKeypoint, pMainOrientation,
                            pDescriptorOut) != SIFT_HIP_OK)
        die("compute_descriptor");
}

void sift_match(const Desc_t *pQueryDesc, int nQuery, const Desc_t *pTrainDesc, int nTrain, DMatch_t *pMatches)
{
    std::lock_guard<std::mutex> l(g_mu);
    if (sift_hip_match(ctx(), pQueryDesc, nQuery, pTrainDesc, nTrain, (sift_dmatch *)pMatches) != SIFT_HIP_OK) die("sift_match");
}
