// Internal declarations shared by the HIP translation units of liblanecv.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <vector>

namespace lanecv {

// ---- geometry of the bit-plane representation ------------------------------
// The Canny classification and the final edge map live in HBM as bit planes,
// one uint64 per 64 consecutive pixels of a row.
// All planes are row-major, P[frame][y][tx] with tx = x / 64 (TX words per row): the strong and
// weak Canny classes, the ROI mask and the masked edge map the Hough stage consumes.  The
// hysteresis and finalize kernels work on 64x64-pixel tiles of them (lane = row).
constexpr int TILE = 64;

struct Geo {
    int W, H;       // pixels
    int TX, TY;     // tiles: ceil(W/64), ceil(H/64)
    __host__ __device__ size_t plane_words() const { return (size_t)TX * H; }      // per frame
    __host__ __device__ size_t rowplane_words() const { return (size_t)TX * H; }
};

inline Geo make_geo(int W, int H) {
    Geo g; g.W = W; g.H = H; g.TX = (W + TILE - 1) / TILE; g.TY = (H + TILE - 1) / TILE; return g;
}

// Which frames of a launch take part in it.  A pipeline slot holds the batches of up to MAX_PARTS callers' submits
// back to back (part j at frames [j * per_part, (j + 1) * per_part)); a launch covers the slot's frames from 0 up to
// the last part that takes part, and every kernel leaves the frames outside the set alone -- no reads, no writes:
// another launch may be working on them.  per_part == 0: every frame of the launch takes part.
constexpr int MAX_PARTS = 16;
struct FrameSet {
    int per_part;
    int count[MAX_PARTS];       // frames of part j that take part (0: none)
    __host__ __device__ int part(int f) const { return per_part ? f / per_part : 0; }
    __host__ __device__ bool has(int f) const {
        if (!per_part) return true;
        const int j = f / per_part;
        return f - j * per_part < count[j];
    }
    __host__ __device__ bool first_of_part(int f) const { return per_part ? (f % per_part == 0 && count[f / per_part] > 0) : f == 0; }
};

// The frames of a launch that does flagged frames only, by number (relative to the launch's first frame): workgroup i does
// frame f[i].  n = 0: a workgroup per frame, each looking its flag up.
struct FlagList {
    int n;
    uint16_t f[62];
};

// Per-angle constants of HoughLinesProbabilistic, computed on the host exactly
// as OpenCV's CPU code does (double cos/sin scaled by the float 1/rho, narrowed
// to float; the walk step uses float division) and uploaded once.
struct AngleEntry {
    float c, s;      // trigtab[2n], trigtab[2n+1]
    int xflag;       // 1: x is the walk's major axis
    int dmaj;        // +-1 step of the major axis
    int dmin;        // 16.16 step of the minor axis
    int pad;
};

// Colour gate in front of the gray conversion: a pixel is kept when its 8-bit HSV value (hue range 180)
// lies in either of two boxes.  A channel test is (x - lo) <= span on unsigned numbers; an empty range
// has lo = 1000, span = 0.  Box 0 takes lo[0..2] / span[0..2] in H, S, V order, box 1 the rest.
// hue_any: bit k = box k accepts every hue; vmin: the smallest v at which a pixel can pass either box (256: none)
struct GateDev { int lo[6]; uint32_t span[6]; int hue_any, vmin; };
// bounds: lo0[3], hi0[3], lo1[3], hi1[3] as the inRange calls receive them (any integers);
// nboxes = 1 leaves the second box empty
GateDev make_gate(const int *bounds, int nboxes);
// the two reciprocal tables of the HSV conversion, sdiv[256] then hdiv[256]
void hsv_tables_host(int *tab512);
// the streaming kernel's tables for one gate: the two reciprocal tables, then per v the diff intervals (bytes
// dmin0, dmax0, dmin1, dmax1) inside which a pixel passes the saturation and value bounds of either box (gate_keep4)
void gate_tables_host(GateDev &gate, int *tab768);   // also sets gate.vmin

// cv::resize INTER_LINEAR tables for (sw, sh) -> (dw, dh): xofs[dw], ia[dw], yofs[dh], ib[dh] (weights packed
// a0 | a1 << 16); returns true when the size pair is the exact 2x2 decimation OpenCV reroutes to its area code
bool resize_tables_host(int sw, int sh, int dw, int dh, int *xofs, int *ia, int *yofs, int *ib);

// ---- launchers (each enqueues on `st`; no synchronisation inside) ----------
// dst = saturate(round(a * alpha + (b * beta + gamma))) over n bytes (cv::addWeighted, 8-bit)
void launch_add_weighted(const uint8_t *a, float alpha, const uint8_t *b, float beta, float gamma, uint8_t *dst, size_t n,
                         hipStream_t st);
// dense frames [n][sh][sw][cn] -> [n][dh][dw][cn]; tabs = the four tables back to back in device memory
// decimate3: resize_is_decimate3 said yes for these tables (3-channel frames with a width that is a multiple of 4 then
// take a copy kernel)
void launch_resize_linear(const uint8_t *src, int sw, int sh, int cn, int nframes, uint8_t *dst, int dw, int dh,
                          const int *tabs, bool area2, hipStream_t st, bool decimate3 = false);
bool resize_is_decimate3(int sw, int sh, int dw, int dh, const int *xofs, const int *ia, const int *yofs, const int *ib);
// colour gate (kernels_gate.hip); tab = device copy of hsv_tables_host
void launch_hsv(const uint8_t *src, size_t npix, int code, const int *tab, uint8_t *dst, hipStream_t st);
void launch_in_range3(const uint8_t *src, size_t npix, GateDev box0, uint8_t *dst, hipStream_t st);
void launch_or(const uint8_t *a, const uint8_t *b, uint8_t *dst, size_t n, hipStream_t st);
void launch_and3_mask(const uint8_t *src, const uint8_t *mask, uint8_t *dst, size_t npix, hipStream_t st);
void launch_gate_gray(const uint8_t *bgr, size_t npix, int gray_code, const int *tab, GateDev gate, uint8_t *gray,
                      hipStream_t st);
// edge stage
void launch_gray(const uint8_t *bgr, int W, int H, int nframes, int code, uint8_t *gray, hipStream_t st);
// streaming front kernel (kernels_front_stream.hip): interleaved 3-channel input, W % 4 == 0
bool front_stream_supported(Geo g);
int front_stream_strips(Geo g);     // 248-column strips per frame row
int front_stream_rows_per_unit(Geo g, int max_batch, int rows_classified);   // rows_classified: 0 = all of them
// Pixels the final edge map cannot depend on need not be classified.  With an ROI a pipeline classifies only the units
// (strip x row segment) that come within RING pixels of it: a launch in mode 1 does those (`needed`, or without a list all
// units from row_split down); launch_hysteresis then decides per frame whether that was enough; a launch in mode 0 with
// only_frames redoes the frames for which it was not, all units; and before the buffers are next used at a level that
// skips units a launch in mode 2 clears the plane rows of the other units (`unneeded`) that the full pass stored, so
// that the next batch finds them empty again.
// Strips need not start at multiples of 248 columns in every row segment: segment sg's strips start at column
// 248 * strip - 8 * by8[sg], so that the first strip an ROI needs in that segment begins where the ROI's neighbourhood
// does (one strip fewer in most segments of a triangle).  nstrips: strips per segment with the shifts applied (one more
// than front_stream_strips wherever a shifted segment needs it to reach the end of the row).  A pipeline fixes the
// shifts once: the change tracking behind rowflags is per strip and row.
constexpr int FRONT_SHIFT_SEGS = 64;
struct FrontShift {
    int nstrips;
    uint8_t by8[FRONT_SHIFT_SEGS];
};
struct FrontSplit {
    int row_split;              // > 0: rows above it form a region of units of their own (never needed)
    int mode;                   // 0: all units; 1: the needed units only; 2: clear the plane rows of the unneeded units
    const int *only_frames;     // device flags, one per frame, or null: frames whose flag is 0 are left alone
    const uint16_t *needed, *unneeded;    // device lists of units (segment | strip << 8), or null
    int nneeded, nunneeded;
    const FrontShift *shift;    // or null: no segment is shifted
};
// the unit grid of a frame for a unit height and a split row (what launch_front_stream uses; for building unit lists)
struct FrontGrid { int nseg_top, rows_top, nseg_low, rows_low, nstrips, strip_cols; };
FrontGrid front_stream_grid(Geo g, int rows_per_unit, int row_split);
// A hysteresis launch (the arguments of launch_hysteresis below) that a streaming front-kernel launch carries as its
// first workgroups: the hysteresis of the batch BEFORE the one the front kernel classifies, one workgroup of 128 threads
// per frame, beside the front kernel's own units instead of between two front kernels.  px_scratch: RING_LIST_WORDS
// words per frame of device memory (the ring check's pixel list, which the stand-alone kernel keeps in LDS).
constexpr int RING_LIST_WORDS = 256;
struct HystLaunch {
    Geo g;
    int nframes;
    uint64_t *planeS;
    const uint64_t *planeW;
    uint32_t *wake;
    int *rounds, *status;
    const uint32_t *ring_words;
    int nring_words;
    const uint64_t *ring_code;
    int *redo;
    uint32_t *px_scratch;
    FrameSet fs;
    int wake_frames;         // see launch_hysteresis
};
// whether launch_front would hand a launch of this kind to the streaming kernel (the only one that can carry a lead)
bool front_takes_lead(Geo g, int in_kind);
void launch_front_stream(const uint8_t *src, int code, bool blur, Geo g, int nframes, int low, int high,
                         uint64_t *planeS, uint64_t *planeW, uint32_t *wake, uint8_t *gray_out, uint8_t *blur_out,
                         int rows_per_unit, const GateDev *gate, const int *tab, hipStream_t st, uint8_t *rowflags = nullptr,
                         bool dec3 = false, const FrontSplit *split = nullptr, const FrameSet *fs = nullptr,
                         const uint8_t *const *src_parts = nullptr, const HystLaunch *lead = nullptr);
// fs / src_parts (may be null): the frames that take part, and where each part's frames start (src is ignored then)
// dec3: src holds frames of 3 W x 3 H pixels; the kernel takes source pixel (3x + 1, 3y + 1) for pixel (x, y), which is
// what cv2.resize's bilinear reduction by exactly three yields (no gate, no intermediate images in this form)
void launch_blur5(const uint8_t *src, int W, int H, int nframes, uint8_t *dst, hipStream_t st);
// in_kind: 0 = gray bytes in, 1 = interleaved 3-channel bytes in
// gate (may be null, 3-channel input only): colour gate in front of the gray conversion.  Frames the
// streaming kernel cannot take go through a gated gray image first: gate_gray (nframes * W * H bytes)
// is where it is put, and it may be the same buffer as gray_out.
// rowflags (may be null; streaming kernel only): nframes * (front_stream_strips + 1) * H bytes, all zero before the first
// call like the two planes; with it only strip rows that hold or held a candidate pixel are stored.
// wake: nframes * TX * TY flags, all zero on entry; the tiles that hold a weak pixel are set (launch_hysteresis
// consumes and clears them).
void launch_front(const uint8_t *src, size_t src_bytes, int in_kind, int code, bool blur, Geo g, int nframes,
                  int low, int high, uint64_t *planeS, uint64_t *planeW, uint32_t *wake,
                  uint8_t *gray_out, uint8_t *blur_out, hipStream_t st,
                  const GateDev *gate = nullptr, const int *tab = nullptr, uint8_t *gate_gray = nullptr,
                  uint8_t *rowflags = nullptr, bool dec3 = false, int rows_per_unit = 0, const FrontSplit *split = nullptr,
                  const FrameSet *fs = nullptr, const uint8_t *const *src_parts = nullptr, const HystLaunch *lead = nullptr);
// lead (streaming kernel, no intermediates, not a flag-checking launch): see HystLaunch
// rows_per_unit (streaming kernel; 0 = derived from nframes): pipelines pass the value of their largest batch, so that
// the unit grid -- and with it the per-unit change tracking behind rowflags -- does not move with the batch size
// hysteresis to the fixed point: one workgroup per frame floods the tiles flagged in wake (by the front
// kernel: those with weak pixels), then repeats over the tiles whose neighbours gained a strong border pixel
// until a round flags none; rounds[f] receives the rounds frame f needed.  wake holds two arrays of
// nframes * tiles flags (the second all zero on entry); both are all zero on exit.
// status (may be null): five words a launch over all frames zeroes; this kernel and the kernels behind it on the stream
// raise them with plain stores -- [0] a frame holds more edge points than its list keeps (launch_points, HoughLinesP),
// [1] a vote left the accumulator, [2] a frame did not fit the LDS the fast HoughLinesP kernel was launched with,
// [3] a frame must be classified again in full (below), [4] a frame is left to the general HoughLinesP kernel
// only_frames (may be null): device flags, one per frame; frames whose flag is 0 are left alone
// ring_code (may be null): after a pass over the units near an ROI only -- does anything outside them matter to frame
// f?  Ring k = the pixels at Chebyshev distance k from the ROI (k = 1..15; the classified units cover them all).  A
// chain of candidate pixels from an ROI pixel to a strong pixel further out passes through every ring (the distance
// changes by at most one per step), and if the ROI pixel is not an edge already, every pixel of the chain inside the
// rings is a weak pixel that stayed weak.  redo[f] = 1 iff such pixels connect ring 1 to ring 15 (8-neighbours, flooded
// from all of ring 1's), or there are too many of them to tell; bit 30 of rounds[f] is set then (the host reads that, frame by frame: a word
// for the whole batch would be written by this launch's workgroups on several XCDs, the clearing one among them).
// ring_words: indices of the plane words that hold ring pixels; ring_code[b]: bit plane b of the ring number (0 = no ring).
void launch_hysteresis(Geo g, int nframes, uint64_t *planeS, const uint64_t *planeW, uint32_t *wake, int *rounds,
                       hipStream_t st, int *status = nullptr, const int *only_frames = nullptr,
                       const uint32_t *ring_words = nullptr, int nring_words = 0, const uint64_t *ring_code = nullptr,
                       int *redo = nullptr, const FrameSet *fs = nullptr, int wake_frames = 0);
// wake_frames: the frames the wake buffer was sized for (the second flag array starts behind that many frames' flags);
// 0 = nframes.  Launches over a part of a buffer's frames must pass it: the second array of a shorter launch would
// otherwise lie where the front kernel of another launch sets the flags of the frames behind.
// fs (may be null): the frames that take part; every part then has status words of its own, status + 8 * part
// roi_bits: one plane shared by all frames, or nullptr
// prevbits (may be null): the masked bits of the previous call on the same masked_out / rowplane buffers, all
// zero before the first (like those buffers); with it, and without edges_out, only what changed is written
void launch_finalize(Geo g, int nframes, const uint64_t *planeS, const uint64_t *roi_bits,
                     uint8_t *edges_out, uint8_t *masked_out, uint64_t *rowplane, int *rowcount, hipStream_t st,
                     uint64_t *prevbits = nullptr, const FrameSet *fs = nullptr,   // fs: with prevbits only
                     // with prevbits and roi_bits: the plane words that hold ROI pixels (row << 8 | word of the row) -- only
                     // those are then looked at
                     const uint32_t *roi_words = nullptr, int n_roi_words = 0);
void launch_and(const uint8_t *a, const uint8_t *b, const uint8_t *mask, uint8_t *dst, size_t n, hipStream_t st);
void launch_pack_nonzero(const uint8_t *img, Geo g, int nframes, uint64_t *rowplane, int *rowcount, hipStream_t st);
// points + Hough
// edge points of each frame in scan order from its bit plane and per-row counts (which it clears).  status (may be
// null) are the words an earlier kernel on the stream zeroed (see launch_hysteresis) and this kernel and the Hough
// kernels behind it raise with plain stores: status[0] = a frame holds more points than the list's capacity (raised here)
void launch_points(Geo g, int nframes, const uint64_t *rowplane, int *rowcount, uint32_t *nz, int nz_cap, int *npoints,
                   int *status, hipStream_t st, const FrameSet *fs = nullptr);
void launch_vote(Geo g, int nframes, const uint32_t *nz, int nz_cap, const int *npoints, const AngleEntry *tab,
                 int numangle, int numrho, int32_t *acc, const int *flags, hipStream_t st, const FrameSet *fs = nullptr);
// Opt-in to more than 64 KiB of dynamic LDS for a kernel.  The attribute holds for the current device only, so the
// largest request is remembered per (call site, device); thread-safe.  Returns the HIP error of a failed opt-in.
struct LdsOptIn {
    size_t have[64] = {};
    hipError_t ensure(const void *kernel, size_t bytes);
};
// Streams that do not share a hardware queue (kernels_probe.hip).  pick() creates streams until it finds one whose
// kernels run beside those of every stream in `others` (at most 12 tries, a few hundred microseconds each; the GPU
// should be idle); failing that it returns the last stream tried and sets *own_queue to false.  Streams it tried and
// dropped are destroyed with the picker.  LANECV_QUEUE_PROBE=0 switches the experiment off.
struct StreamPicker {
    int *flags_host = nullptr, *flags_dev = nullptr;
    std::vector<hipStream_t> rejected;
    bool side_by_side(hipStream_t a, hipStream_t b);
    hipStream_t pick(const std::vector<hipStream_t> &others, bool *own_queue);
    ~StreamPicker();
};
// LDS atomic-add rate probe (kernels_probe.hip): best of three launches; 0 on success
int lds_atomic_probe(int run, bool returning, int blocks, int iters, int *sink, hipStream_t st, float *ms, double *adds);
bool launch_ppht_fast(Geo g, int nframes, uint32_t *nz, int nz_cap, const int *npoints, uint64_t *rowplane,
                      const AngleEntry *tab, int numangle, int numrho, int threshold, int line_length, int line_gap,
                      int32_t *seg, int seg_cap, int *seg_count, int *status, int *flags, const uint32_t *rng_out,
                      int rng_len, int32_t *acc_full, hipStream_t st, int lds_bytes = 0, bool only_flagged = false, int waves = 0,
                      int status_frames = 0, const FrameSet *fs = nullptr, int device_state = 0, const FlagList *list = nullptr);
// device_state 1: the form of the kernel that leaves the edge bits in the bit plane in device memory, 2: the point list as
// well, permuted in place (with only_flagged, in this order between a launch of the ordinary form and the general kernel:
// they take the frames whose mask words, or points, did not fit the LDS beside the accumulator).  A frame the kernel
// flags gets seg_count -1 until another kernel has done it: from those the caller makes `list`, so that a launch of
// these forms is as many workgroups as there are frames to do -- each needs a whole CU's LDS, i.e. a CU that the
// kernels of the batches in flight have left empty, whether it then finds its frame flagged or not.
// status_frames: 0 = one block of status words for the launch; n = frames [k n, (k + 1) n) raise status + 8 k (a launch
// over the batches of several pipeline slots, each with status words of its own)
// waves: 8 or 4 wavefronts per frame (0: 4 with ppht_lds_half(), else 8).  Four is for launches that fill every CU twice
// over; a batch that leaves CUs free is better served by eight even with half the LDS.
// lds_bytes: dynamic LDS per frame (0 = the most a CU gives one workgroup); with ppht_lds_half() two frames share a
// CU, and a second launch with only_flagged and the full size takes the frames that needed more
int ppht_lds_half();
void launch_ppht_general(Geo g, int nframes, uint32_t *nz, int nz_cap, const int *npoints, uint64_t *rowplane,
                         const AngleEntry *tab, int numangle, int numrho, int32_t *acc, int threshold, int line_length,
                         int line_gap, int32_t *seg, int seg_cap, int *seg_count, int *status, const int *flags,
                         hipStream_t st, int32_t *vote_acc = nullptr, int status_frames = 0);   // vote_acc: also the full-vote accumulator of the frames flagged with bit 1
// polygons
struct PolyEdgeDev { int y0, y1; long long x, dx; };
struct LineDev { int x0, y0, count, vert, dmaj, dmin, dmajor_len, dminor_len; };
void launch_fill_spans(uint8_t *img, int W, int H, const PolyEdgeDev *edges, int nedges, int ymin, int ymax,
                       int color, hipStream_t st);
void launch_draw_lines(uint8_t *img, int W, int H, const LineDev *lines, int nlines, int color, hipStream_t st);
void launch_fill_rect(uint8_t *img, int W, int H, int x1, int y1, int x2, int y2, int color, hipStream_t st);
// thick lines on dense images of cn channels (kernels_poly.hip); color = one byte per channel, channel 0 lowest
struct TrapDev { int y0, y1; long long xa, dxa, xb, dxb; };              // rows [y0, y1) of FillConvexPoly's edge walk
struct Line2Dev { int xmajor, ecount; long long major0, minor0, step, endx, endy; };   // Line2 after its set-up
void launch_paint_traps(uint8_t *img, int W, int H, int cn, const TrapDev *traps, int ntraps, uint32_t color, hipStream_t st);
void launch_paint_line2(uint8_t *img, int W, int H, int cn, const Line2Dev *lines, int nlines, int maxcount, uint32_t color,
                        hipStream_t st);
void launch_paint_circles(uint8_t *img, int W, int H, int cn, const int2 *centres, int ncircles, const int *hw, int radius,
                          uint32_t color, hipStream_t st);
void launch_draw_lines_cn(uint8_t *img, int W, int H, int cn, const LineDev *lines, int nlines, uint32_t color, hipStream_t st);

}  // namespace lanecv
