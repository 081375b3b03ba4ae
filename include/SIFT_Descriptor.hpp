// SIFT_Descriptor.hpp -- compute_main_orientation / compute_descriptor with the reference's
// signatures (/root/reference/src/SIFT_Descriptor.hpp:12-14, :59-62; SIFT_Descriptor.cpp:18-103,
// :419-478).  Supported domain of caller-supplied gradients: angGrad 0..8, magGrad <= 32767 (what
// sift_detect produces is 0..8 and <= 360; outside it the reference indexes out of bounds or
// reinterprets the magnitude as a signed 16-bit value).
#pragma once
#include "SIFTParameters.hpp"

void compute_main_orientation(Grad_t *pGrad, Keypoint_t *pKeypoint, uint8_t *pMainOrientation);
void compute_descriptor(Grad_t *pGrad, Keypoint_t *pKeypoint, uint8_t *pMainOrientation, Desc_t *pDescriptorOut);
