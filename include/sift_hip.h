/*
 * sift_hip.h -- C ABI of the MI355X (gfx950) SIFT hot path.
 *
 * This is the drop-in boundary: the entry points are what the reference's
 * SDSoC hardware-function calls would bind to.  Plain pointers and sizes only.
 * All citations are into the reference tree (/root/reference/src).
 *
 * Conventions
 *   - every function returns 0 on success, a negative code otherwise
 *     (SIFT_HIP_E*); sift_hip_last_error() gives the text of the last failure
 *     on the calling thread.  The reference has no error channel (all `void`,
 *     HW_SIFT.hpp:11-15); the C++ wrappers in include/HW_SIFT.hpp abort on a
 *     non-zero code.
 *   - a context owns one device, one stream and persistent device scratch for
 *     `max_batch` frames (this replaces the per-call sds_alloc/sds_free of
 *     HW_SIFT.cpp:23-24,44-45).  A context is not thread-safe; distinct
 *     contexts are independent.  All calls on one context share that scratch, so
 *     they must be ordered with respect to each other on the device: use one
 *     stream per context (the *_dev calls take a stream argument; two calls of one
 *     context on two streams race), or two contexts for two streams.
 *   - frame size is a run-time parameter of the context (the reference fixes
 *     WIDTH/HEIGHT at compile time, SIFTParameters.hpp:17-18): any width in
 *     [1, 65532] and height in [1, 65535].  Frames, Grad_t planes and every
 *     other caller-side array are dense rows of `width` elements; internally rows
 *     are padded to a multiple of 4 with columns that read as "outside the image".
 *     (Widths that are multiples of 4 are taken as they are; others cost one
 *     repacking pass per call, and their stand-alone orientation / descriptor
 *     calls run the slower Grad_t sweep.)
 *   - outputs are always fully written: exactly max_keypoints keypoint slots
 *     (zero padded, DetectKeypoints.cpp:428-436) and max_keypoints*72
 *     descriptor words per frame.  Descriptors of padding slots are 0 (the
 *     reference leaves them uninitialised, SIFT_Descriptor.cpp:194-196).
 */
#ifndef SIFT_HIP_H
#define SIFT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIFT_HIP_MAX_KEYPOINTS 10000 /* MAX_KEYPOINTS, SIFTParameters.hpp:22 */
#define SIFT_HIP_DESC_SIZE 72        /* DESC_SIZE,     SIFTParameters.hpp:26 */

#define SIFT_HIP_OK 0
#define SIFT_HIP_EINVAL (-1)  /* bad argument (null pointer, size, batch) */
#define SIFT_HIP_EHIP (-2)    /* a HIP runtime call failed */
#define SIFT_HIP_ENODEV (-3)  /* no usable gfx950 device / code object */

/* Keypoint_t, SIFTParameters.hpp:79-85 (8 bytes) */
typedef struct {
    uint16_t x;        /* column */
    uint16_t y;        /* row */
    uint32_t contrast; /* |DoG2| at the keypoint */
} sift_keypoint;

/* Grad_t, SIFTParameters.hpp:88-92 (4 bytes) */
typedef struct {
    uint16_t angGrad; /* direction code 0..8 */
    uint16_t magGrad; /* floor(sqrt(gx^2+gy^2)) */
} sift_grad;

/* DMatch_t, SIFTParameters.hpp:122-128 (8 bytes) */
typedef struct {
    uint32_t distance; /* integer where OpenCV has a float */
    int16_t queryIdx;  /* "querryIdx" in the reference */
    int16_t trainIdx;
} sift_dmatch;

typedef struct sift_hip_ctx sift_hip_ctx;

/* ---- life cycle -------------------------------------------------------- */

/* Replaces the sds_alloc calls of main.cpp:18-32 / HW_SIFT.cpp:23-24: creates
 * the device-side state for frames of width x height on HIP device `device`.
 * max_keypoints is the per-frame cap (reference: 10000); max_batch is the
 * largest number of frames one call may carry (>= 1). */
int sift_hip_create(sift_hip_ctx **out, int device, int width, int height, int max_keypoints, int max_batch);

/* The same with the descriptor geometry chosen.  SIFT_HIP_GEOMETRY_CAM0 is what the reference's code
 * uses (WINDOW_SIZE 71, MASK_SIZE 25, XStart/YStart, MaskAll; SIFTParameters.hpp:34-44, circle_mask.hpp:7-40,
 * mask_all.hpp:8-78) and what sift_hip_create gives.  SIFT_HIP_GEOMETRY_CAM1 uses the second set of tables the
 * reference ships but never reads (DESC_WIN_SIZE_1 81, MASK_SIZE_1 29, XStartCam1/YStartCam1,
 * CircleMaskCam1, MaskAll_81; SIFTParameters.hpp:46-55, circle_mask.hpp:12-13, :43-71, mask_all.hpp:161-241;
 * SIFT_Descriptor.cpp:496 is an empty "CAM1" section).  Its semantics are DEFINED here as the Cam0 loops
 * (SIFT_Descriptor.cpp:52-100, :196-253) with those constants substituted: orientation over 81 rows x 80
 * columns (2 * radius, :70) of MaskAll_81, nine 29 x 29 circle histograms at (XStartCam1, YStartCam1).
 * Only the orientation / descriptor stage depends on the geometry; keypoints do not. */
#define SIFT_HIP_GEOMETRY_CAM0 0
#define SIFT_HIP_GEOMETRY_CAM1 1
int sift_hip_create_geometry(sift_hip_ctx **out, int device, int width, int height, int max_keypoints, int max_batch,
                             int geometry);

/* Replaces sds_free (main.cpp:214-217, HW_SIFT.cpp:44-45). */
void sift_hip_destroy(sift_hip_ctx *ctx);

/* Text of the last error on this thread ("" if none). */
const char *sift_hip_last_error(void);

/* Number of HIP devices visible (0 if none / runtime unusable). */
int sift_hip_device_count(void);

/* Pinned host memory for image / keypoint / descriptor buffers: the counterpart of
 * sds_alloc and sds_free (main.cpp:18-32, :214-217).  NULL on failure. */
void *sift_hip_host_alloc(unsigned long bytes);
void sift_hip_host_free(void *p);

/* ---- host-buffer entry points (the reference's four hardware functions) -- */

/* sift_detect_and_compute, HW_SIFT.hpp:11-15 / HW_SIFT.cpp:16-50.
 * img: width*height bytes; kps: max_keypoints slots; desc: max_keypoints*72. */
int sift_hip_detect_and_compute(sift_hip_ctx *ctx, const uint8_t *img, sift_keypoint *kps, uint32_t *desc,
                                int contrastTh);

/* sift_detect, DetectKeypoints.hpp:109-113 / DetectKeypoints.cpp:1073-1175.
 * grad (width*height Grad_t) may be NULL when the caller does not want it. */
int sift_hip_detect(sift_hip_ctx *ctx, const uint8_t *img, sift_grad *grad, sift_keypoint *kps, int contrastTh);

/* compute_main_orientation, SIFT_Descriptor.hpp:12-14 / SIFT_Descriptor.cpp:18-103.
 * Supported planes: magGrad <= 32767 wherever angGrad is 1..8 (sift_detect produces at most 360).  The
 * reference accumulates ap_int<16> MagGrad (:81), so a larger magnitude would count as negative there;
 * this call and sift_hip_descriptor return SIFT_HIP_EINVAL for such a plane (the *_dev variants, which do
 * not synchronise, leave the result unspecified). */
int sift_hip_main_orientation(sift_hip_ctx *ctx, const sift_grad *grad, const sift_keypoint *kps, uint8_t *ori);

/* compute_descriptor, SIFT_Descriptor.hpp:59-62 / SIFT_Descriptor.cpp:419-478. */
int sift_hip_descriptor(sift_hip_ctx *ctx, const sift_grad *grad, const sift_keypoint *kps, const uint8_t *ori,
                        uint32_t *desc);

/* The keypoint post-processing of main.cpp:82-130: the nfinal (reference:
 * MAX_FINAL_KEYPOINTS = 1000, SIFTParameters.hpp:24) slots of largest contrast of kps
 * (max_keypoints slots, padding included), in descending order of contrast -- what
 * std::sort(pKeypoints, ..., CompareContrast) + memcpy leave in pFinalKeypoints (:82-84).
 * std::sort is not stable; here equal contrasts keep slot (= raster) order.
 * index (optional): slot of final_kps[r] in kps / in the descriptor matrix;
 * nvalid (optional): how many of them have x > 0 && y > 0 (nKp, main.cpp:112-129).
 * 1 <= nfinal <= min(max_keypoints, 4096). */
#define SIFT_HIP_MAX_FINAL_KEYPOINTS 1000
int sift_hip_select_top(sift_hip_ctx *ctx, const sift_keypoint *kps, int nfinal, sift_keypoint *final_kps,
                        uint32_t *index, uint32_t *nvalid);

/* The image loop of main.cpp:37-210 as one call over one context per GPU: context i takes the contiguous
 * range of frames sift_hip_batch_range gives it (ceil(nframes / nctx) frames each, the last ranges
 * shorter or empty), on a host thread of its own; outputs land in the caller's per-frame slots
 * kps[f*max_keypoints], desc[f*max_keypoints*72].  Frames are independent: no exchange between
 * contexts.  Inside a context the frames travel in chunks of max_batch / 2, double buffered on the
 * context's own copy streams (created by its first batch call, kept until sift_hip_destroy: a later
 * call creates no stream, event or device buffer); frames that are neighbours in host memory go up
 * as one copy, a chunk's results come back as one copy per array. */
int sift_hip_batch(sift_hip_ctx *const *ctxs, int nctx, const uint8_t *const *imgs, int nframes,
                   sift_keypoint *kps, uint32_t *desc, int contrastTh);

/* The partition rule of sift_hip_batch[_top] (needs no device): context i of nctx
 * owns frames [*first, *last) of nframes. */
int sift_hip_batch_range(int nframes, int nctx, int i, int *first, int *last);

/* The same loop with the post-processing of main.cpp:82-130 on the device: per frame only the
 * nfinal strongest keypoints (descending contrast, ties in slot order), their slots, their
 * descriptor rows and the count nKp travel back to the host -- 0.3 MB instead of 2.96 MB per frame
 * at nfinal = 1000.  final_kps: nframes*nfinal; final_index, final_desc (nframes*nfinal*72), nvalid
 * (nframes) may be NULL. */
int sift_hip_batch_top(sift_hip_ctx *const *ctxs, int nctx, const uint8_t *const *imgs, int nframes, int nfinal,
                       sift_keypoint *final_kps, uint32_t *final_index, uint32_t *final_desc, uint32_t *nvalid,
                       int contrastTh);

/* Brute-force matcher for DMatch_t (SIFTParameters.hpp:122-128).  NO REFERENCE CODE: the reference declares
 * the type (and links OpenCV's features2d / flann, .cproject:93-94) but never matches; SURVEY.md 8(f)4 lists a
 * matcher as an unpinned extension.  Semantics chosen to fit the struct: cv::BFMatcher(NORM_L1).match() on raw
 * 72-word descriptor rows -- matches[q] = { sum over j of |query[q][j] - train[t][j]|, q, t } for the train row t
 * of smallest distance (ties: smaller t); { 0xffffffff, q, -1 } when ntrain is 0.  nquery, ntrain <= 32767.
 * Distances are 32-bit like the field: supported are rows whose distances stay below 2^32 (the words
 * compute_descriptor produces stay below 2^18, so 72 of them cannot reach that). */
int sift_hip_match(sift_hip_ctx *ctx, const uint32_t *query_desc, int nquery, const uint32_t *train_desc, int ntrain,
                   sift_dmatch *matches);

/* The ingest step of main.cpp:51-58 for colour frames: 24-bit BGR rows (as in a BMP: `pitch` bytes
 * apart, a multiple of 4, 0 = 3*width; bottom_up != 0 when the first stored row is the bottom one)
 * to the width*height gray bytes the entry points above take (pitch 0 = 3*width rounded up to 4), with the fixed-point weights of
 * cv::imread(..., CV_LOAD_IMAGE_GRAYSCALE): gray = (1868 B + 9617 G + 4899 R + 8192) >> 14.
 * OpenCV is not part of the reference tree: the formula is restated from its published source. */
int sift_hip_gray_from_bgr(sift_hip_ctx *ctx, const uint8_t *bgr, long pitch, int bottom_up, uint8_t *gray);

/* ---- device-buffer entry points ------------------------------------------
 * Same work with inputs and outputs already resident in this context's device
 * memory (what the SDSoC data movers of DetectKeypoints.cpp:1054-1069 would
 * otherwise copy).  d_imgs: nframes contiguous width*height frames;
 * d_kps: nframes*max_keypoints; d_desc: nframes*max_keypoints*72;
 * nframes <= max_batch.  `stream` is a hipStream_t (NULL = the context's own
 * stream); the call only enqueues work, it does not synchronise.
 * d_total (optional, may be NULL) receives per frame the uncapped survivor
 * count (nKeypointsTotal, DetectKeypoints.cpp:380,405-406). */
int sift_hip_detect_and_compute_dev(sift_hip_ctx *ctx, const uint8_t *d_imgs, int nframes, sift_keypoint *d_kps,
                                    uint32_t *d_desc, uint32_t *d_total, int contrastTh, void *stream);

/* sift_detect on device buffers; d_grad (nframes*width*height Grad_t) may be NULL. */
int sift_hip_detect_dev(sift_hip_ctx *ctx, const uint8_t *d_imgs, int nframes, sift_grad *d_grad,
                        sift_keypoint *d_kps, uint32_t *d_total, int contrastTh, void *stream);

/* compute_main_orientation / compute_descriptor on device buffers (one frame's
 * Grad_t plane per frame).  The planes are first rewritten into the context's
 * internal gradient plane (so d_grad is only read, and the context's scratch of
 * the fused call is overwritten); planes with a magnitude above 360 -- more than
 * sift_detect can produce -- are swept in their own format instead, at about a
 * third of the speed, with the same results. */
int sift_hip_main_orientation_dev(sift_hip_ctx *ctx, const sift_grad *d_grad, int nframes,
                                  const sift_keypoint *d_kps, uint8_t *d_ori, void *stream);
int sift_hip_descriptor_dev(sift_hip_ctx *ctx, const sift_grad *d_grad, int nframes, const sift_keypoint *d_kps,
                            const uint8_t *d_ori, uint32_t *d_desc, void *stream);

/* sift_hip_select_top on device buffers: d_kps nframes*max_keypoints -> d_final
 * nframes*nfinal; optional d_index nframes*nfinal, d_nvalid nframes.  When d_desc
 * (nframes*max_keypoints*72) and d_final_desc (nframes*nfinal*72) are given, the
 * descriptor rows of the selected keypoints are gathered as well (the reference leaves the
 * descriptor matrix in slot order, main.cpp:134-148, so row r of its file does not belong to
 * final keypoint r; `index` is what links the two).  nframes is not bound by max_batch. */
int sift_hip_select_top_dev(sift_hip_ctx *ctx, const sift_keypoint *d_kps, int nframes, int nfinal,
                            sift_keypoint *d_final, uint32_t *d_index, uint32_t *d_nvalid, const uint32_t *d_desc,
                            uint32_t *d_final_desc, void *stream);

/* sift_hip_match on device buffers, for `npairs` independent pairs of descriptor sets: pair p matches rows
 * d_query[p * nquery ...] against d_train[p * ntrain ...] into d_matches[p * nquery ...] (16-byte aligned arrays). */
int sift_hip_match_dev(sift_hip_ctx *ctx, const uint32_t *d_query_desc, int nquery, const uint32_t *d_train_desc, int ntrain,
                       int npairs, sift_dmatch *d_matches, void *stream);

/* sift_hip_gray_from_bgr on device buffers: d_bgr nframes*height*pitch bytes (4-byte aligned) ->
 * d_gray nframes*width*height, e.g. straight into the d_imgs of the calls above. */
int sift_hip_gray_from_bgr_dev(sift_hip_ctx *ctx, const uint8_t *d_bgr, int nframes, long pitch, int bottom_up,
                               uint8_t *d_gray, void *stream);

/* sift_hip_detect_and_compute_dev for callers that cannot allocate: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE
 * (its sift_detect_and_compute allocates scratch on every call, HW_SIFT.cpp:23-24).  It takes no
 * context: all device scratch lives in one static __device__ array of
 * SIFT_HIP_STATIC_SCRATCH_BYTES inside the library (each GPU the process touches holds its own
 * copy until the process ends), and the call makes no hipMalloc / hipFree and creates no stream,
 * event or graph.  d_imgs, d_kps (nframes*10000 slots), d_desc (nframes*10000*72) and the optional
 * d_total (nframes) are the caller's device buffers on `device`; max_keypoints is fixed at
 * SIFT_HIP_MAX_KEYPOINTS; the work is enqueued on `stream` (NULL = the legacy default stream) and
 * not waited for.
 * Limit: sift_hip_static_scratch_needed(width, height, nframes) (about 10.3 bytes per pixel and
 * frame + 0.26 MB) must not exceed SIFT_HIP_STATIC_SCRATCH_BYTES -- one 1920x1080 frame or ten
 * 640x480 frames fit; larger calls fail with SIFT_HIP_EINVAL and a message that says so.
 * A call for a frame width whose descriptor tables are not in the array builds them on the host, copies
 * them on `stream` and waits for that copy: make such a call outside stream capture.  The tables of the two
 * most recently used widths stay in the array, so only the first call for a width -- and calls that rotate
 * through three or more widths -- pay that; everything else only enqueues.
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two
 * threads) race; order them on one stream or with events.  On the host, calls for one GPU are serialised by a
 * lock of that GPU's own (held across the table copy's wait); calls for different GPUs do not wait for each other. */
#define SIFT_HIP_STATIC_SCRATCH_BYTES (32u << 20)
int sift_hip_detect_and_compute_static(int device, const uint8_t *d_imgs, int nframes, int width, int height,
                                       sift_keypoint *d_kps, uint32_t *d_desc, uint32_t *d_total, int contrastTh,
                                       void *stream);
/* Bytes of the static array a call of that shape uses (0 for an invalid shape). */
unsigned long sift_hip_static_scratch_needed(int width, int height, int nframes);

/* Block until everything this context enqueued has finished. */
int sift_hip_sync(sift_hip_ctx *ctx);

/* ---- further octaves (extension) ------------------------------------------ */

/* The reference processes one octave (sift_detect's three blurs of the input frame, DetectKeypoints.cpp:1073-1175);
 * SURVEY.md 8(f)4 lists a multi-octave mode as an optional extension, with nothing in the reference to pin it to.
 * This build defines it by composition: the input frame of the next octave is every second pixel of every second
 * row -- pixel (2x, 2y) -- of one blurred plane of the frame just processed (plane 1 = the 5x5 table's G1, 2 = the
 * 11x11 table's G2, which keypoints are detected on, 3 = the 19x19 table's G3), (width/2) x (height/2) pixels with
 * pitch width/2, and that frame goes through the unchanged single-octave entry points of a context created for that
 * size.  Keypoint coordinates are in the octave's own pixels (multiply by 2^octave for the input frame's).
 *
 * sift_hip_next_octave_dev: frames frame_first .. frame_first + nframes - 1 of the LAST detect call on `ctx`, into
 * d_dst (device, nframes * (width/2) * (height/2) bytes); enqueued on `stream` (NULL = the context's own), which must be the stream of that
 * call (one stream per context).  sift_hip_next_octave: one frame into a host buffer, blocking. */
int sift_hip_next_octave_dev(sift_hip_ctx *ctx, int frame_first, int nframes, int plane, uint8_t *d_dst, void *stream);
int sift_hip_next_octave(sift_hip_ctx *ctx, int frame, int plane, uint8_t *dst);

/* ---- introspection (tests, bench) ---------------------------------------- */

/* Stage planes of frame `frame` of the LAST detect call on this context,
 * copied to host.  Any pointer may be NULL.  g1/g2/g3: blurred planes
 * (DetectKeypoints.cpp:1034-1037), width*height bytes each; cand: pre-cull
 * candidate map |DoG2| (kpMap1, :207), u16 per pixel; eratio: Q15 eigen ratio
 * at candidate pixels, 0x7fffffff elsewhere (the reference writes it for every
 * pixel, :209; only candidate values are ever read, :728); surv: post-cull map
 * before the border filter (kpMap2, :693). */
int sift_hip_debug_planes(sift_hip_ctx *ctx, int frame, uint8_t *g1, uint8_t *g2, uint8_t *g3, uint16_t *cand,
                          int32_t *eratio, uint16_t *surv);

/* How many streams (what = 0), events (1) and device buffers (2) the library has created in this
 * process so far (-1 for another `what`).  Test hook: two readings around a call show what it created. */
long sift_hip_debug_created(int what);

/* Per-kernel HIP-event timing.  sift_hip_set_profiling(ctx, nslots) keeps a ring
 * of the last `nslots` calls (0 switches profiling off); every *_dev / host
 * call then records one event pair per kernel on the stream it launches on.
 * sift_hip_stage_ms(ctx, back, ms) waits for and returns the kernel times (ms,
 * -1 for a kernel that call did not launch) of the call `back` calls before the
 * most recent one (0 = most recent); its return value is the number of calls
 * the ring holds.  sift_hip_stage_name gives the kernel names in launch order.
 * This replaces the sds_clock_counter() bracket of main.cpp:63-77. */
#define SIFT_HIP_NSTAGES 5
int sift_hip_set_profiling(sift_hip_ctx *ctx, int nslots);
int sift_hip_stage_ms(sift_hip_ctx *ctx, int back, float ms[SIFT_HIP_NSTAGES]);
const char *sift_hip_stage_name(int stage);

/* Tuning knob: rows per thread strip of the blur kernel (0 = heuristic). */
int sift_hip_set_blur_rows(sift_hip_ctx *ctx, int rows);
/* Test / comparison knob: on = 0 runs the blurs on the vector ALUs (v_dot4_u32_u8, k_blur) instead of
 * the matrix cores (v_mfma_i32_16x16x64_i8, k_blur_mfma).  Same bits either way.  Default: on. */
int sift_hip_set_blur_mfma(sift_hip_ctx *ctx, int on);
/* Tuning knob: sift_hip_detect_and_compute_dev calls of up to `max_frames` frames (default 4) that
 * repeat the previous call's buffers and parameters are replayed as one HIP graph; 0 = never. */
int sift_hip_set_graph_frames(sift_hip_ctx *ctx, int max_frames);
/* Test knob: on = 0 sends every descriptor window of sift_hip_detect_and_compute[_dev] through the
 * pixel-by-pixel sweep that otherwise only serves windows the quad sweep cannot take (none that
 * sift_detect emits); results are identical, the kernel is several times slower.  Default: on. */
int sift_hip_set_quad_sweep(sift_hip_ctx *ctx, int on);
/* Tuning knob: rows per strip of the detect kernel (0 = heuristic; clamped to [4, 32]). */
int sift_hip_set_detect_rows(sift_hip_ctx *ctx, int rows);
/* Test knob: LDS queue entries the detect kernel may hold (0 = the full queue, 512; clamped to [256, 512]); it
 * drains the queue in the middle of a strip after any row that leaves fewer than 256 of them free.  Results
 * do not depend on it. */
int sift_hip_set_detect_queue(sift_hip_ctx *ctx, int entries);

#ifdef __cplusplus
}
#endif
#endif /* SIFT_HIP_H */
