// The batched pipeline behind lc_pipe_*: a slot's buffers and events, the pipeline object, and what pipe_create.hip
// (set-up) and pipe_run.hip (submit / wait / read) share.
#pragma once
#include "host.h"

using lanecv::Geo;
using lanecv::GateDev;
using lanecv::FrontSplit;

// ---------------------------------------------------------------------------
// What a caller submits into and waits for.  A slot owns no buffers: it is a place in a bank.
struct Slot {
    int bank = 0, pos = 0;            // its bank and its place there: frames [pos * max_batch, (pos + 1) * max_batch) of the bank's buffers
    int nframes = 0;
    const uint8_t *input = nullptr;   // device pointer of the submitted frames (inside the bank's own input buffer for host submits)
    bool from_host = false;
    const uint8_t *host_src = nullptr;  // host submits: where the frames came from (stays valid until the slot has been waited for)
    int rows_from = 0;                // host submits: the first row of every frame that has been uploaded so far
    bool windowed = false;            // ... and of those rows only the column windows the tightest level reads (lc_pipe::up_windows)
    bool eager_redo = true;           // its launch classified flagged frames again itself (it could: they were there in full)
    bool busy = false;                // submitted and not yet waited for
    bool pending = false;             // submitted, but the bank's kernels for it are not queued yet
    bool timed = false;               // the launch that covers it recorded the stage events
    bool fast_lines = false;          // its HoughLinesP ran in the LDS-resident kernel, which may leave frames to the general one
    int level = 0;                    // ROI-scheme level of the launch that covers it; the pipeline's nlevels = classified in full
    bool tail_queued = true;          // what follows hysteresis is queued (false while its launch's hysteresis waits to lead the next front launch)
    hipEvent_t done = nullptr;        // fires when the launch that covers the slot has finished
    // its part of the bank's pinned result block
    int *h_rounds = nullptr, *h_status = nullptr, *h_segcount = nullptr;
    int32_t *h_seg = nullptr;
};

// A set of buffers for `parts` slots back to back, and the events of its launches.  Batches that fill the GPU get a
// bank each (parts = 1).  Batches that leave most CUs free -- every kernel behind the front kernel is one workgroup
// per frame, and what the host pays per launch does not shrink with the batch -- share one: the kernels are queued
// when the bank's last slot has been submitted (or one of its slots is waited for), ONE launch each over the frames
// of all the slots submitted since the last time (a FrameSet says which; the kernels leave every other frame alone,
// an earlier launch over other slots of the bank may still be running).
struct Bank {
    Work w;
    int parts = 1;
    hipEvent_t ev[12] = {};           // of the bank's latest launch (stage times, and the waits between its streams)
    int planes_level = 0;             // loosest level a launch of this bank ran at since its planes were last cleared down
    bool redone = false;              // ... and some of its frames were classified again in full since then (their redo flags say which)
    int32_t *h_res = nullptr;         // pinned host mirror of the result block: [segcount][rounds][status 8 * parts][segments]
    bool zero_copy = false;           // the kernels write the result block straight into h_res (mapped pinned memory)
};

// One launch of a bank: which of its slots take part, and over how many of the bank's frames the kernels are launched
// (from frame 0 up to the last frame of the last slot taking part; the frame set makes them skip the rest).
struct Launch {
    lanecv::FrameSet fs{};
    const uint8_t *src[lanecv::MAX_PARTS] = {};
    int nframes = 0;
    int level = 0;
};

// A launch whose front kernel is queued and whose hysteresis is not yet: it will lead the next front launch (or, if the
// launch's slots are waited for first, run as a kernel of its own), and everything behind it follows then.
struct PendingTail {
    bool active = false;
    int bank = 0;
    Launch L;
    hipStream_t ls = nullptr;         // the launch's tail stream
    bool skipping = false, eager = false, fused_resize = false;
    size_t src_bytes = 0;
};

struct lc_pipe {
    lc_ctx *ctx = nullptr;
    lc_pipe_params p{};
    Geo g{};
    int low = 0, high = 0, line_length = 0, line_gap = 0;
    DevBuf<uint64_t> roi_bits;
    DevBuf<uint32_t> roi_words;   // the plane words that hold ROI pixels, row << 8 | word of the row (what finalize looks at)
    int n_roi_words = 0;
    bool has_roi = false;
    bool has_gate = false;        // colour gate in front of the gray conversion
    GateDev gate{};
    DevBuf<int> gate_tab;     // gate_tables_host: the HSV reciprocal tables and this gate's s/v intervals
    // cv2.resize in front of everything: submitted frames are src_w x src_h, the pipeline works on width x height
    bool resizes = false, resize_area2 = false, resize_dec3 = false;
    int src_w = 0, src_h = 0;
    size_t src_frame_bytes = 0;   // bytes of one submitted frame
    // the reduction by exactly three is read by the front kernel itself (no resized frame exists then): without
    // intermediates, without the colour gate, widths the streaming kernel takes
    bool fused_dec3() const {
        return resizes && resize_dec3 && !p.keep_intermediates && !has_gate && front_stream_supported(g) && !getenv("LANECV_FRONT_TILED") &&
               !getenv("LANECV_NO_FUSED_RESIZE");
    }
    DevBuf<int> resize_tab;
    // A pipeline owns three or four streams of equal priority (more HoughLinesP streams for small batches, below),
    // whatever its number of slots; a slot is a set of buffers and events.  HIP maps streams onto a few hardware
    // queues, so the count stays small, and the waits between streams run one way only -- st_up -> st_edge ->
    // st_post -> st_lines.
    //   st_up     host-to-device copies of submitted frames (created by the first host submit).
    //   st_edge   the front kernels (and the leading resize) of all slots, in submission order, back to back.
    //             They fill the GPU by themselves, so two of them gain nothing from running side by side.
    //   st_post   the short latency-bound kernels behind a front kernel: hysteresis, finalize, compaction (and the
    //             vote kernel of accumulator-only recipes); with a single slot it is st_edge itself.
    //   st_lines  HoughLinesP and the copy of the batch's results.
    hipStream_t st_up = nullptr, st_edge = nullptr, st_post = nullptr, st_lines = nullptr;
    // More HoughLinesP streams: the kernel is one sequential workgroup per frame, so a launch over fewer frames than
    // the GPU has room for leaves the rest to whatever may run beside it -- and on a single stream that is never
    // another bank's HoughLinesP.  Launches take the streams in turn: as many as launches fit the GPU side by side
    // ((two frames per CU ? 2 : 1) * CUs / frames of a bank), at most three and no more than there are banks.
    // LANECV_LINES_STREAMS overrides the count.
    std::vector<hipStream_t> st_lines_more;
    unsigned lines_turn = 0;
    // Tail streams: the short kernels behind a front kernel run on the HoughLinesP stream of their own launch instead of
    // a post stream shared by all launches (they precede that launch's HoughLinesP anyway), which leaves a hardware queue
    // for a third HoughLinesP stream: while one launch's short kernels wait for room beside a front kernel, the other
    // streams' HoughLinesP kernels run on, and a third launch fills the CUs the slowest frames of the first leave idle.
    bool tail_streams = false;
    bool hyst_on_edge = false;    // with tail streams: hysteresis stays on the edge stream, behind its front kernel
    // ... and, for launches the streaming front kernel takes, not as a kernel of its own but as the first workgroups of the
    // NEXT front launch (k_front_lead): the edge stream then holds front kernels only, back to back.  The tail of a launch
    // is therefore queued one submit later (or when its slots are waited for): `pend`.
    bool hyst_fused = false;
    PendingTail pend;
    // The streaming front kernel's unit height (fixed for the pipeline's largest batch: its change tracking is per unit)
    // and, for pipelines with an ROI, which units it classifies at all.  The final edge map is the hysteresis result
    // INSIDE the ROI, and that depends on pixels outside it only through weak pixels that connect the two.  The front
    // kernel classifies the units (strip x row segment) that come within 15 pixels of the ROI; the hysteresis kernel proves per
    // frame that nothing further out can matter (one of the 15 rings around the ROI without leftover weak pixels), and the
    // rare frame it cannot prove it for is classified again in full.  Not with intermediates (the pre-ROI edge map
    // must then be complete).  row_split: the first row of the lower region of the unit grid (rows above hold no
    // needed unit).
    int front_rows = 0, row_split = 0, roi_first_row = 0;
    // Levels of the scheme, tightest first: [0] the units within 15 pixels of the ROI, rings around the ROI; [1] all units
    // from row_split down, "rings" = the rows between row_split and the ROI's first row (the same proof around the band of
    // rows the ROI spans: for recipes whose weak pixels connect everything near the ROI, as the F1 tracker's low Canny
    // thresholds do, but not the sky above it).  `level` is what the next batch uses; nlevels = classify in full.  A batch
    // in which more than a quarter of the frames needed the full pass after all moves the pipeline one level out for
    // the next 256 batches, after which it tries one level in again.  A slot that goes back to a tighter level has the
    // plane rows of the units that level does not need cleared first (for every frame: `ones`).
    struct SkipLevel {
        DevBuf<uint16_t> needed, unneeded;      // units: segment | strip << 8
        int n_needed = 0, n_unneeded = 0, n_ring_words = 0;
        DevBuf<uint32_t> ring_words;            // plane words that hold ring pixels
        DevBuf<uint64_t> ring_code;             // four bit planes: the ring number (1..15) of every pixel, 0 = in no ring
        long long pixels = 0;                   // pixels per frame of the needed units
    };
    SkipLevel levels[2];
    int nlevels = 0, level = 0, level_hold = 0;
    // Launches queue the second pass over flagged frames (the front kernel again, hysteresis again: two launches that
    // find nothing to do when no frame is flagged) only while flagged frames are about -- for this many launches after
    // lc_pipe_wait saw the second batch with one within a few launches.  Otherwise a flagged frame is made up for when its
    // slot is waited for (enqueue_repair).
    int redo_hold = 0;
    int redo_recent = 0;          // launches left in which another flagged frame starts the hold (lc_pipe_wait)
    DevBuf<int> ones;
    // What the front kernel reads of a frame at the tightest level, as a few windows (rows [r0, r1), bytes [b0, b1) of each):
    // per row segment the columns of its needed strips and the rows the units warm up and drain on.  Frames submitted from
    // host memory while the pipeline is at that level are uploaded window by window (a 3-D copy each); whatever else turns out
    // to be needed follows as whole rows or whole frames.
    struct UpWindow { int r0, r1, b0, b1; };
    std::vector<UpWindow> up_windows;
    long long up_window_bytes = 0;    // per frame
    // where each row segment's strips start (FrontShift): chosen from the tightest level's units, the same for all levels
    lanecv::FrontShift shift{};
    bool shifted = false;
    long long frame_pixels = 0;
    FrontSplit split(int lvl, int mode, const int *only) const {
        const lanecv::FrontShift *sh = shifted ? &shift : nullptr;
        if (lvl >= nlevels) return FrontSplit{row_split, 0, only, nullptr, nullptr, 0, 0, sh};
        const SkipLevel &L = levels[lvl];
        return FrontSplit{row_split, mode, only, L.needed.p, L.unneeded.p, L.n_needed, L.n_unneeded, sh};
    }
    bool fuse_vote = true;        // HoughLinesP's kernel also produces the full-vote accumulator (in LDS, before its own)
    // HoughLinesP with half a CU's LDS per frame, so that two frames -- of two batches, on two streams -- share a CU:
    // the kernel is one sequential wavefront per frame and leaves most of a CU idle.  For pipelines of two banks or more
    // and frame sizes whose working set usually fits (up to about 2.5 megapixels; 1080p road frames take 47 - 49 KiB).
    // A frame that needs more is done again when its slot is waited for (lc_pipe_wait sees status word 2), with a whole
    // CU's LDS; when that happens twice within a few launches the next `ppht_full_for` batches get a whole CU's LDS per
    // frame from the start, after which the half size is tried anew.
    bool ppht_two_per_cu = false;
    int ppht_full_for = 0;
    int ppht_flag_recent = 0;     // launches left in which a second flagged frame switches to whole-CU launches (lc_pipe_wait)
    int cu_count = 256;
    int shared_queues = 0;        // streams of this layout that found no hardware queue of their own (lc_pipe_create)
    hipEvent_t ev_base = nullptr; // recorded when stage times were first switched on
    bool timing = false;          // record the events lc_pipe_stage_ms reads (lc_pipe_set_timing); a dozen more packets per batch

    // slots per bank (see Bank), frames a bank holds, and the banks
    int parts = 1, bank_frames = 0;
    std::vector<Bank> banks;
    std::vector<Slot> slots;
};

namespace lanecv {

enum { EV_START = 0, EV_UPLOADED, EV_FRONT, EV_HYST, EV_EDGE, EV_POINTS, EV_VOTE, EV_LINES, EV_END, EV_ESTART, EV_LSTART, EV_VSTART };

// queues what is still to be queued of the pipeline's batches (slots submitted into banks that are not full yet)
void pipe_flush_pending(lc_pipe *pipe);

}  // namespace lanecv
