/*
 * lanecv.h -- C ABI of the MI355X lane-pipeline library (liblanecv.so).
 *
 * Every entry point replaces one call the reference's lane scripts make into
 * the third-party cv2 module; the citation beside each declaration names the
 * reference call site (file:line under the reference repository).  Plain C:
 * pointers and sizes only, no exceptions cross this boundary.  Every function
 * returns LC_OK (0) or a negative LC_E* code; lc_last_error() returns the
 * message of the calling thread's last failure.
 *
 * Images are row-major uint8.  Colour frames are H x W x 3 interleaved in
 * memory order (what cv2.imread / VideoCapture.read return); gray images,
 * edge maps and masks are H x W.  A stride is the distance between rows in
 * bytes.  All per-op functions take HOST pointers and are synchronous; the
 * lc_pipe_* functions are the batched, asynchronous, device-resident path.
 */
#ifndef LANECV_H
#define LANECV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC_OK            0
#define LC_EINVAL       -1   /* bad argument (cv2 would raise cv2.error) */
#define LC_ENOMEM       -2
#define LC_ECAP         -3   /* caller-provided output capacity too small */
#define LC_EHIP         -4   /* HIP runtime failure (message has the HIP error) */
#define LC_EUNSUPPORTED -5   /* valid cv2 call outside the supported parameter domain */

/* cv2.COLOR_BGR2GRAY / cv2.COLOR_RGB2GRAY */
#define LC_COLOR_BGR2GRAY 6
#define LC_COLOR_RGB2GRAY 7
/* cv2.COLOR_BGR2HSV / cv2.COLOR_RGB2HSV (8-bit, hue in [0, 180)) */
#define LC_COLOR_BGR2HSV 40
#define LC_COLOR_RGB2HSV 41

typedef struct lc_ctx lc_ctx;     /* bound to one HIP device; calls on one ctx are serialised by the caller */
typedef struct lc_pipe lc_pipe;   /* a configured batched pipeline living on a ctx */

int lc_version(void);
const char *lc_last_error(void);

/* One context per GPU / worker thread.  Fails (LC_EHIP) when no HIP device is usable. */
int lc_create(int device, lc_ctx **out);
int lc_destroy(lc_ctx *ctx);

/* Pinned host memory for frame rings fed to lc_pipe_submit_host. */
int lc_host_alloc_pinned(size_t bytes, void **out);
int lc_host_free(void *p);

/* ---- per-op entry points (host pointers, synchronous) -------------------- */

/* cv2.cvtColor(src, COLOR_BGR2GRAY | COLOR_RGB2GRAY), 8UC3 -> 8UC1.
 * Replaces: Simple Lane Detection with OpenCV_2(images).py:69 (RGB2GRAY),
 *           F1_vo_Lane_Tracking.py:186 (BGR2GRAY). */
int lc_cvt_gray(lc_ctx *ctx, const uint8_t *src, int w, int h, size_t src_stride,
                int code, uint8_t *dst, size_t dst_stride);

/* cv2.cvtColor(src, COLOR_BGR2HSV | COLOR_RGB2HSV), 8UC3 -> 8UC3 (H in [0,180), S, V).
 * Replaces: Simple Lane Detection with OpenCV_6(mp4).py:65. */
int lc_cvt_hsv(lc_ctx *ctx, const uint8_t *src, int w, int h, size_t src_stride,
               int code, uint8_t *dst, size_t dst_stride);

/* cv2.inRange(src, lowerb, upperb) on an 8UC3 image with three integer bounds each
 * (any integers: bounds outside [0, 255] behave as in cv2); dst is 8UC1, 0 / 255.
 * Replaces: Simple Lane Detection with OpenCV_6(mp4).py:77-78. */
int lc_in_range3(lc_ctx *ctx, const uint8_t *src, int w, int h, size_t src_stride,
                 const int *lowerb, const int *upperb, uint8_t *dst, size_t dst_stride);

/* cv2.bitwise_or(a, b) on 8UC1 images.
 * Replaces: Simple Lane Detection with OpenCV_6(mp4).py:81. */
int lc_or(lc_ctx *ctx, const uint8_t *a, size_t a_stride, const uint8_t *b, size_t b_stride,
          uint8_t *dst, size_t dst_stride, int w, int h);

/* cv2.bitwise_and(src, src, mask=mask) on an 8UC3 image with an 8UC1 mask: pixels whose
 * mask byte is 0 come out black (cv2 allocates a zeroed destination).
 * Replaces: Simple Lane Detection with OpenCV_6(mp4).py:84. */
int lc_and3_mask(lc_ctx *ctx, const uint8_t *src, size_t src_stride, const uint8_t *mask, size_t mask_stride,
                 uint8_t *dst, size_t dst_stride, int w, int h);

/* cv2.resize(src, (dw, dh)) with the default INTER_LINEAR on an 8-bit image of 1 or 3 channels
 * (OpenCV's 11-bit fixed-point bilinear; an exact 2x2 decimation takes its area average instead).
 * Replaces: F1_vo_Lane_Tracking.py:185. */
int lc_resize_linear(lc_ctx *ctx, const uint8_t *src, int sw, int sh, size_t src_stride, int channels,
                     uint8_t *dst, int dw, int dh, size_t dst_stride);

/* cv2.addWeighted(src1, alpha, src2, beta, gamma) on 8-bit images of one size (any channel count: rows of
 * row_bytes = width * channels bytes): dst = saturate(round(src1 * alpha + src2 * beta + gamma)), computed in
 * float as OpenCV does.
 * Replaces: Simple Lane Detection with OpenCV_2(images).py:50, ..._5(mp4).py:50, F1_vo_Lane_Tracking.py:196. */
int lc_add_weighted(lc_ctx *ctx, const uint8_t *a, size_t a_stride, double alpha, const uint8_t *b, size_t b_stride,
                    double beta, double gamma, uint8_t *dst, size_t dst_stride, int row_bytes, int h);

/* cv2.GaussianBlur(src, (5,5), 0), 8UC1, BORDER_REFLECT_101.
 * Replaces: F1_vo_Lane_Tracking.py:97. */
int lc_gaussian5(lc_ctx *ctx, const uint8_t *src, int w, int h, size_t src_stride,
                 uint8_t *dst, size_t dst_stride);

/* cv2.Canny(image, threshold1, threshold2) with apertureSize 3, L2gradient False.
 * Replaces: Simple Lane Detection with OpenCV_2(images).py:72 (100, 200),
 *           F1_vo_Lane_Tracking.py:98 (50, 150). */
int lc_canny(lc_ctx *ctx, const uint8_t *src, int w, int h, size_t src_stride,
             double threshold1, double threshold2, uint8_t *dst, size_t dst_stride);

/* No counterpart in the reference: cv2.Canny (same arithmetic and result as lc_canny) for callers that cannot
 * allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * The call makes no hipMalloc / hipFree and no allocating library call: its working memory (two class bit planes,
 * the hysteresis work lists) is a fixed array of 2.4 MB inside the library's code object, one copy per GPU.
 *   dev_image  DEVICE pointer, h rows of w gray bytes, dense
 *   dev_edges  DEVICE pointer, h rows of w bytes (0 / 255), dense, 16-byte aligned
 *   stream     a hipStream_t of the context's device (NULL: the device's null stream).  The kernels are enqueued
 *              there and the call returns without waiting; the caller synchronises the stream.
 * Limit: w <= 4096 and h <= 2304 (LC_EUNSUPPORTED beyond; lc_canny has no such limit).
 * All calls on one GPU share the one array: calls that overlap in time on a GPU (different streams, threads or
 * contexts) race -- keep them on one stream or order them with events. */
int lc_canny_noalloc(lc_ctx *ctx, const void *dev_image, int w, int h, double threshold1, double threshold2,
                     void *dev_edges, void *stream);

/* cv2.fillPoly(img, pts, color) with LINE_8, shift 0, on an 8UC1 image, in place.
 * pts holds the (x, y) int32 pairs of all contours back to back, npts[i] vertices each.
 * Replaces: Simple Lane Detection with OpenCV_2(images).py:20,
 *           F1_vo_Lane_Tracking.py:111, :175. */
int lc_fill_poly(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride,
                 const int32_t *pts, const int *npts, int ncontours, int color);

/* cv2.rectangle(img, pt1, pt2, color, -1) (filled) on an 8UC1 image, in place.
 * Replaces: F1_vo_Lane_Tracking.py:179-180. */
int lc_rectangle_filled(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride,
                        int x1, int y1, int x2, int y2, int color);

/* cv2.line(img, pt1, pt2, color, thickness) with LINE_8, shift 0, on an 8-bit image of 1..4 channels, in place.
 * color holds one value per channel (clamped to 0..255).  thickness 1 is the 8-connected line; above it
 * OpenCV's thick line: the convex quad around the segment, filled and outlined, with round caps at both ends.
 * Replaces: Simple Lane Detection with OpenCV_2(images).py:47 (drawn with thickness 5 from :146). */
int lc_line(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride, int channels, int x1, int y1, int x2, int y2,
            const int *color, int thickness);

/* cv2.polylines(img, pts, isClosed, color, thickness): every edge of every contour as a thick line with a round
 * cap at its far end (an open polyline caps its first point too).  pts / npts as in lc_fill_poly.
 * Replaces: Simple Lane Detection with OpenCV_5(mp4).py:41, :48 (thickness 8), F1_vo_Lane_Tracking.py:138, :140 (5). */
int lc_polylines(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride, int channels, const int32_t *pts, const int *npts,
                 int ncontours, int is_closed, const int *color, int thickness);

/* cv2.bitwise_and(a, b[, mask=mask]) on 8UC1 images; mask may be NULL.
 * Replaces: Simple Lane Detection with OpenCV_2(images).py:23,
 *           F1_vo_Lane_Tracking.py:102 (mask=), :112. */
int lc_and(lc_ctx *ctx, const uint8_t *a, size_t a_stride, const uint8_t *b, size_t b_stride,
           const uint8_t *mask, size_t mask_stride, uint8_t *dst, size_t dst_stride, int w, int h);

/* Geometry HoughLinesP derives from (w, h, rho, theta): number of angles and of
 * rho bins.  Either output may be NULL. */
int lc_hough_geometry(int w, int h, double rho, double theta, int *numangle, int *numrho);

/* Full-vote Hough accumulator of an 8UC1 edge map: every non-zero pixel adds one
 * vote per angle at the bin HoughLinesP would use; acc is [numangle][numrho] int32.
 * (cv2.HoughLinesP never exposes its accumulator; this is the object the
 * accumulator parity test compares.) */
int lc_hough_accum(lc_ctx *ctx, const uint8_t *img, int w, int h, size_t stride,
                   double rho, double theta, int32_t *acc, int acc_capacity,
                   int *numangle, int *numrho);

/* Measurement aid with no counterpart in the reference: the rate at which this GPU performs int32 atomic adds
 * in LDS, in 1e9 lane-level adds per second, for the ceiling the vote kernels are reported against.
 * run = 1: every lane its own word; run = r (2..64): r neighbouring lanes share a word; run = 0: pseudo-random
 * words.  returning != 0 measures the form that hands the old value back (HoughLinesP's vote). */
int lc_lds_atomic_rate(lc_ctx *ctx, int run, int returning, double *gadds_per_s);

/* cv2.HoughLinesP(image, rho, theta, threshold, minLineLength=, maxLineGap=).
 * out receives up to `capacity` rows of (x1, y1, x2, y2) in OpenCV's detection
 * order; *n is the number found (LC_ECAP if it exceeds capacity).
 * Replaces: Simple Lane Detection with OpenCV_2(images).py:84-92,
 *           Simple Lane Detection with OpenCV_5(mp4).py:76-84. */
int lc_hough_lines_p(lc_ctx *ctx, const uint8_t *img, int w, int h, size_t stride,
                     double rho, double theta, int threshold,
                     double min_line_length, double max_line_gap,
                     int32_t *out, int capacity, int *n);

/* ---- batched pipeline (device-resident, asynchronous) -------------------- */

typedef struct lc_pipe_params {
    int width, height;        /* frame size */
    int max_batch;            /* frames per submit, upper bound */
    int nslots;               /* independent in-flight batches.  A pipeline's streams do not grow with it; slots of small
                                 batches (2 * max_batch <= CUs, four slots or more) share banks of buffers and are
                                 launched a bank at a time: see lc_pipe_layout */
    int color_code;           /* LC_COLOR_BGR2GRAY or LC_COLOR_RGB2GRAY */
    int blur;                 /* 1: GaussianBlur 5x5 between gray and Canny (F1 recipe) */
    double canny_lo, canny_hi;
    const uint8_t *roi_mask;  /* host, w*h, result of the script's fillPoly; NULL = no ROI */
    const uint8_t *car_mask;  /* host, w*h, applied as bitwise_and(e, e, mask=); NULL = none */
    double rho, theta;
    int threshold;
    double min_line_length, max_line_gap;
    int seg_capacity;         /* segments kept per frame */
    int want_accumulator;     /* compute the full-vote accumulator */
    int want_segments;        /* run HoughLinesP */
    int keep_intermediates;   /* also materialise gray, blurred and pre-ROI edge images */
    /* Colour gate of Simple Lane Detection with OpenCV_6(mp4).py:65-88 in front of the gray conversion:
     * hsv = cvtColor(frame, BGR2HSV); m = inRange(hsv, lo0, hi0) | inRange(hsv, lo1, hi1);
     * gray = cvtColor(bitwise_and(frame, frame, mask=m), color_code).  LC_BUF_GRAY is that gated gray. */
    int hsv_gate;             /* 1: apply the gate */
    int hsv_bounds[12];       /* lo0[3], hi0[3], lo1[3], hi1[3], each in H, S, V order */
    int want_points;          /* keep the ordered list of edge points even without a Hough stage (lc_pipe_read_points) */
    /* cv2.resize(frame, (width, height)) in front of the recipe (F1_vo_Lane_Tracking.py:185): submitted frames
     * are src_width x src_height; 0 = the pipeline's own size, no resize. */
    int src_width, src_height;
} lc_pipe_params;

/* What lc_pipe_read copies out of a finished slot. */
#define LC_BUF_GRAY     1     /* uint8  [n][h][w]   (keep_intermediates) */
#define LC_BUF_BLUR     2     /* uint8  [n][h][w]   (keep_intermediates, blur) */
#define LC_BUF_EDGES    3     /* uint8  [n][h][w]   Canny output before the masks (keep_intermediates) */
#define LC_BUF_MASKED   4     /* uint8  [n][h][w]   edge map after car mask and ROI */
#define LC_BUF_ACC      5     /* int32  [n][numangle][numrho] (want_accumulator) */
#define LC_BUF_NPOINTS  6     /* int32  [n]         non-zero pixels of the masked edge map */
#define LC_BUF_RESIZED  7     /* uint8  [n][h][w][3] the frames after the leading resize (src_width / src_height set;
                                 an exact reduction by three is read by the edge kernel directly and only
                                 materialised with keep_intermediates) */

/* Stages timed with HIP events on the slot's stream (lc_pipe_stage_ms). */
#define LC_STAGE_UPLOAD 0
#define LC_STAGE_EDGE   1     /* gray (+blur) + Sobel + NMS + hysteresis + masks */
#define LC_STAGE_POINTS 2     /* ordered compaction of the edge points */
#define LC_STAGE_VOTE   3     /* full-vote accumulator */
#define LC_STAGE_LINES  4     /* HoughLinesP */
#define LC_STAGE_TOTAL  5
#define LC_STAGE_FRONT  6     /* part of EDGE: the fused gray/blur/Sobel/NMS kernel */
#define LC_STAGE_HYST   7     /* part of EDGE: the hysteresis rounds */
#define LC_STAGE_FINAL  8     /* part of EDGE: masks + 0/255 store */
#define LC_NUM_STAGES   9

int lc_pipe_create(lc_ctx *ctx, const lc_pipe_params *params, lc_pipe **out);
int lc_pipe_destroy(lc_pipe *pipe);

/* Copies nframes BGR frames from host memory (pinned for true overlap) to the
 * slot on its stream and enqueues the whole pipeline behind the copy.  host_bgr must
 * stay valid and unchanged until lc_pipe_wait on the slot returns: the copy is
 * asynchronous, and from pinned memory, with an ROI, only what the kernels read is
 * copied at first (the rest of a frame follows in the rare case that it is needed
 * after all).  Frames in ordinary (pageable) memory are copied whole, as one linear
 * copy: partial copies from such memory run row by row and cost more than they save. */
int lc_pipe_submit_host(lc_pipe *pipe, int slot, const uint8_t *host_bgr, int nframes);
/* Same, but the frames already sit in device memory (no copy is made; the
 * buffer must stay valid and unchanged until lc_pipe_wait on the slot returns: the kernels run some time after
 * this call, and a frame the pipeline's short cuts cannot settle is read a second time). */
int lc_pipe_submit_device(lc_pipe *pipe, int slot, const void *dev_bgr, int nframes);
/* Blocks until the slot is done, then copies the segments out:
 * seg_xyxy [nframes][seg_capacity][4] int32, seg_count [nframes] int32 (either may be NULL).
 * A frame with more than seg_capacity segments reports its true count. */
int lc_pipe_wait(lc_pipe *pipe, int slot, int32_t *seg_xyxy, int32_t *seg_count);
int lc_pipe_read(lc_pipe *pipe, int slot, int what, void *host_dst, size_t bytes);
/* The non-zero pixels of one frame's masked edge map in row-major order, as x | y << 16 -- what
 * masked_edges.nonzero() enumerates (F1_vo_Lane_Tracking.py:117-119).  The pipeline must have been
 * configured with want_points, want_accumulator or want_segments (with want_segments, call it before the
 * frame's HoughLinesP fallback could have permuted the list: it is exact whenever the status is clean).
 * *n receives the count; LC_ECAP if it exceeds `capacity`. */
int lc_pipe_read_points(lc_pipe *pipe, int slot, int frame, uint32_t *xy, int capacity, int *n);
/* Stage times of the slot's last batch, from HIP events on the streams the stages ran on.  The events are only
 * recorded for batches submitted after lc_pipe_set_timing(pipe, 1) (a dozen more packets per batch on the pipeline's
 * streams: worth leaving off for small batches, whose rate the host's launch calls bound); LC_EINVAL otherwise.
 * Diagnostics with no counterpart in the reference. */
int lc_pipe_stage_ms(lc_pipe *pipe, int slot, int stage, float *ms);
int lc_pipe_set_timing(lc_pipe *pipe, int on);
/* Diagnostics: when the twelve events of the slot's last batch fired, in milliseconds since the pipeline was created
 * (START UPLOADED FRONT HYST EDGE POINTS VOTE LINES END ESTART LSTART VSTART; a negative value = not recorded for that
 * batch).  Needs lc_pipe_set_timing(pipe, 1). */
int lc_pipe_event_times(lc_pipe *pipe, int slot, float *ms12);
/* Device pointer of a slot's own input buffer (for callers that fill it on the device). */
int lc_pipe_input_device_ptr(lc_pipe *pipe, int slot, void **dev_ptr);
/* Hysteresis diagnostics of the slot's last batch: rounds until the fixed point (the first
 * included), maximum over the frames and total. */
int lc_pipe_hyst_rounds(lc_pipe *pipe, int slot, int *rounds_max, int *rounds_total);
/* Diagnostics with no counterpart in the reference.  A pipeline with an ROI classifies only the parts of a frame within 15
 * pixels of it (the result inside the ROI is proven independent of the rest per frame, and a frame for which it cannot be
 * proven is classified again in full): *first_row = the first row of which anything is classified (0: from the top),
 * *pixels_classified = pixels per frame the edge kernel classifies (width x height without an ROI), *frames_redone = frames
 * of the slot's last batch that needed the full pass.  Any pointer may be NULL. */
int lc_pipe_roi_skip(lc_pipe *pipe, int slot, int *first_row, long long *pixels_classified, int *frames_redone);
/* Diagnostics with no counterpart in the reference: the first row of every frame of the slot's last batch that was
 * copied to the GPU by lc_pipe_submit_host (0: whole frames, also for batches submitted from device memory).  With an
 * ROI the rows above the first one the kernels read are not copied. */
int lc_pipe_rows_uploaded(lc_pipe *pipe, int slot, int *first_row);
/* Diagnostics with no counterpart in the reference: bytes per frame of the slot's last batch that lc_pipe_submit_host
 * (and whatever the launch fetched after it) copied to the GPU.  With an ROI, while the pipeline classifies only the
 * units near it, of the rows from *first_row down only the columns those units read are copied; 0 for batches
 * submitted from device memory.  (Frames that had to be classified in full are fetched whole when the slot is waited
 * for; they are not counted here.) */
int lc_pipe_bytes_uploaded(lc_pipe *pipe, int slot, long long *bytes_per_frame);
/* Diagnostics with no counterpart in the reference: how many of the pipeline's streams found no hardware queue of
 * their own when it was created (HIP folds a process's streams onto GPU_MAX_HW_QUEUES queues; the pipeline probes
 * which ones share).  0 = every stage overlaps as designed; otherwise stages that were meant to run side by side
 * were put on one stream. */
int lc_pipe_shared_queues(lc_pipe *pipe, int *count);
/* Diagnostics with no counterpart in the reference: how the pipeline laid itself out.  *group_size = consecutive slots
 * that share a bank of buffers and ONE launch of every kernel, queued when the bank's last slot is submitted or one of
 * its slots is waited for (1: a launch per slot; more for batches that leave most CUs free), *lines_streams = streams
 * HoughLinesP launches take in turn, *roi_levels = levels of the ROI scheme (0: every frame is classified in full).
 * Any pointer may be NULL. */
int lc_pipe_layout(lc_pipe *pipe, int *group_size, int *lines_streams, int *roi_levels);
/* Synchronous copy of nframes host frames into the slot's own input buffer, nothing else:
 * a later lc_pipe_submit_device on that pointer then runs with the input resident in HBM. */
int lc_pipe_upload(lc_pipe *pipe, int slot, const uint8_t *host_bgr, int nframes);

#ifdef __cplusplus
}
#endif
#endif /* LANECV_H */
