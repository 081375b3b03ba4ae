// HW_SIFT.hpp -- sift_detect_and_compute with the reference's signature
// (/root/reference/src/HW_SIFT.hpp:11-15, HW_SIFT.cpp:16-50), served by the MI355X library
// through the C ABI of sift_hip.h.  Link with -lhw_sift -lsift_hip.
#pragma once
#include "SIFTParameters.hpp"

// frame -> MAX_KEYPOINTS keypoints (zero padded) + MAX_KEYPOINTS x DESC_SIZE descriptor words.
// Like the reference it returns nothing; a failure of the device path prints the library's error
// text and aborts (the reference has no error channel).
void sift_detect_and_compute(Pixel_t *pSrcImageCam0, Keypoint_t *pKeypoints, Desc_t *pDesc, int contrastTh);

// ---- additions of this build (no counterpart in the reference) ---------------------------------
// Frame size used by the four entry points; defaults to WIDTH x HEIGHT.  Must be called while no
// other thread is inside an entry point; it drops the cached device context.
void sift_set_frame_size(int width, int height);
// HIP device the entry points run on (default 0).
void sift_set_device(int device);
// Releases the cached device context (optional; done at process exit otherwise).
void sift_shutdown(void);
// Pinned host buffers, the counterpart of sds_alloc / sds_free (main.cpp:18-32, :214-217).
void *sift_alloc(unsigned long bytes);
void sift_free(void *p);
// The keypoint post-processing of main.cpp:82-130 on the device: the MAX_FINAL_KEYPOINTS keypoints of
// largest contrast in descending order (what std::sort with CompareContrast + memcpy leave in
// pFinalKeypoints), equal contrasts in slot order.  pSlot (may be NULL) receives the slot of each
// final keypoint in pKeypoints / pDesc.  Returns nKp, the number with x > 0 && y > 0 (:112-129).
// pKeypoints itself is left as it is (the reference sorts it in place and never reads it again).
int sift_select_final_keypoints(const Keypoint_t *pKeypoints, Keypoint_t *pFinalKeypoints, uint32_t *pSlot);
// The ingest step of main.cpp:51-58 for a colour frame: 24-bit BGR rows (`pitch` bytes apart, a multiple
// of 4; bottomUp != 0 for BMP's row order) to WIDTH x HEIGHT gray bytes on the device, with the weights
// of cv::imread(..., CV_LOAD_IMAGE_GRAYSCALE).
void sift_gray_from_bgr(const uint8_t *pBgr, long pitch, int bottomUp, Pixel_t *pGray);
// Descriptor geometry of the entry points: 0 = Cam0 (default; what the reference's code computes),
// 1 = Cam1 (the reference's second set of tables, which its code never reads: DESC_WIN_SIZE_1, MASK_SIZE_1,
// CircleMaskCam1, XStartCam1 / YStartCam1, MaskAll_81).  Drops the cached device context.
void sift_set_geometry(int geometry);
// Fills DMatch_t (SIFTParameters.hpp:122-128; the reference declares the type and has no matcher): for each of the
// nQuery descriptor rows the train row of smallest L1 distance (integer, as the struct's comment asks), ties to
// the smaller index; nQuery, nTrain <= 32767.
void sift_match(const Desc_t *pQueryDesc, int nQuery, const Desc_t *pTrainDesc, int nTrain, DMatch_t *pMatches);
