// SIFTParameters.hpp -- configuration contract of the hot path, with the reference's names.
//
// Mirrors /root/reference/src/SIFTParameters.hpp:17-128 for the host side of the MI355X build:
// same macro names and values, same struct names, field names, sizes and layouts, but plain
// fixed-width integers where the reference uses Xilinx ap_uint<N> (ap_uint<8/16/32> are 1/2/4
// byte PODs), so buffers are interchangeable byte for byte.  WIDTH and HEIGHT
// can be overridden with -DWIDTH=... -DHEIGHT=... or at run time with sift_set_frame_size()
// (HW_SIFT.hpp); the reference fixes them at compile time (:17-18).
#pragma once
#include <stdint.h>

#ifndef WIDTH
#define WIDTH 1600                     /* :17 */
#endif
#ifndef HEIGHT
#define HEIGHT 1300                    /* :18 */
#endif

#define N_SCALES 3                     /* :20 */
#define MAX_KEYPOINTS 10000            /* :22 */
#define MAX_FINAL_KEYPOINTS 1000       /* :24 */
#define DESC_SIZE 72                   /* :26 */
#define NCIRCLES 9                     /* :28 */
#define NDIRECTIONS 8                  /* :30 */
#define MASK_SIZE 25                   /* :32 */
#define WINDOW_SIZE 71                 /* :34 */
#define WINDOW_OFFS (WINDOW_SIZE / 2)  /* :38 */
#define FILTER_SIZE1 5                 /* :60 */
#define FILTER_SIZE2 11                /* :64 */
#define FILTER_SIZE3 19                /* :68 */

typedef struct {
    uint16_t x;        /* Col       (:81) */
    uint16_t y;        /* Row       (:82) */
    uint32_t contrast; /* Contrast  (:83) */
} Keypoint_t;          /* :79-85, 8 bytes */

typedef struct {
    uint16_t angGrad;  /* direction code 0..8 (:90) */
    uint16_t magGrad;  /* magnitude          (:91) */
} Grad_t;              /* :88-92, 4 bytes */

typedef uint8_t Pixel_t;   /* :97  */
typedef uint32_t Desc_t;   /* :103 */

typedef struct {
    uint32_t bin0, bin1, bin2, bin3, bin4, bin5, bin6, bin7;
} Hist_t;              /* :108-119 */

typedef struct {
    uint32_t distance;
    int16_t querryIdx;
    int16_t trainIdx;
} DMatch_t;            /* :122-128 (type only; the reference has no matcher) */

#ifdef __cplusplus
static_assert(sizeof(Keypoint_t) == 8 && sizeof(Grad_t) == 4 && sizeof(Hist_t) == 32, "reference layouts");
#endif
