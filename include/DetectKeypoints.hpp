// DetectKeypoints.hpp -- sift_detect with the reference's signature
// (/root/reference/src/DetectKeypoints.hpp:109-113, DetectKeypoints.cpp:1073-1175).
#pragma once
#include "SIFTParameters.hpp"

#define EIGEN_RATIO_TH 12      /* DetectKeypoints.hpp:18 */
#define CULL_SIZE 7            /* 7x7 cull window, DetectKeypoints.cpp:567-750 */

// frame -> Grad_t plane of G2 (WIDTH*HEIGHT) + MAX_KEYPOINTS keypoints in raster order, zero padded
void sift_detect(Pixel_t *pSrcImage, Grad_t *pGradients, Keypoint_t *pKeypoints, int contrastTh);
