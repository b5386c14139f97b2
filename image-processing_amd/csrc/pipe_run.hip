// The batched pipeline at run time: what a submit queues on the pipeline's streams, the wait, and the read-backs.
#include "pipe.h"

#include <chrono>
#include <thread>

using namespace lanecv;

namespace {

// an event that only the stage times need
inline void stamp(lc_pipe *pp, Bank &bk, int ev, hipStream_t st) {
    if (pp->timing) (void)hipEventRecord(bk.ev[ev], st);
}

bool vote_is_fused(const lc_pipe *pp) { return pp->p.want_accumulator && pp->p.want_segments && pp->fuse_vote; }

// What follows hysteresis on stream es: the masked edge map, the ordered point list and, for recipes that want the
// accumulator without segments (or with LANECV_VOTE_FUSED=0), the vote kernel.
void enqueue_post(lc_pipe *pp, Bank &bk, const Launch &L, hipStream_t es) {
    Work &w = bk.w;
    const lc_pipe_params &p = pp->p;
    const int n = L.nframes;
    const bool hough = p.want_accumulator || p.want_segments;
    const bool points = hough || p.want_points;
    // nobody consumes (and clears) the row counts: the slot's own rows, then
    if (!points)
        for (int j = 0; j < bk.parts; j++)
            if (L.fs.count[j])
                (void)hipMemsetAsync(w.rowcount.p + (size_t)j * p.max_batch * w.g.H, 0, sizeof(int) * (size_t)L.fs.count[j] * w.g.H, es);
    launch_finalize(w.g, n, w.planeS.p, pp->has_roi ? pp->roi_bits.p : nullptr, w.inter ? w.edges.p : nullptr, w.masked.p,
                    w.rowplane.p, w.rowcount.p, es, w.inter ? nullptr : w.prevbits.p, &L.fs, pp->roi_words.p, pp->n_roi_words);
    stamp(pp, bk, EV_EDGE, es);
    if (points) launch_points(w.g, n, w.rowplane.p, w.rowcount.p, w.nz.p, w.nz_cap, w.npoints.p, w.r_status, es, &L.fs);
    // The full-vote accumulator depends only on the point list.  With segments wanted, the HoughLinesP kernel
    // produces it on the side (in the LDS array it then counts in); otherwise the vote kernel does, here.
    const bool own_vote = p.want_accumulator && !vote_is_fused(pp);
    if (own_vote) {
        stamp(pp, bk, EV_POINTS, es);
        launch_vote(w.g, n, w.nz.p, w.nz_cap, w.npoints.p, w.tab.p, w.hg.numangle, w.hg.numrho, w.acc_full.p, nullptr, es, &L.fs);
    }
}

// HoughLinesP of a launch's frames on stream ls, and the events that mark its slots done
void enqueue_lines(lc_pipe *pp, Bank &bk, Slot *first, const Launch &L, hipStream_t ls) {
    Work &w = bk.w;
    const lc_pipe_params &p = pp->p;
    const bool vote_fused = vote_is_fused(pp);
    stamp(pp, bk, EV_LSTART, ls);
    bool fast = false;
    if (p.want_segments) {
        fast = launch_ppht_fast(w.g, L.nframes, w.nz.p, w.nz_cap, w.npoints.p, w.rowplane.p, w.tab.p, w.hg.numangle, w.hg.numrho,
                                p.threshold, pp->line_length, pp->line_gap, w.r_seg, w.seg_cap, w.r_segcount, w.r_status, w.fallback.p,
                                w.rng_out.p, w.rng_len, vote_fused ? w.acc_full.p : nullptr, ls,
                                (pp->ppht_two_per_cu && pp->ppht_full_for == 0) ? ppht_lds_half() : 0, false,
                                2 * L.nframes > pp->cu_count ? 0 : 8, p.max_batch, &L.fs);
        if (pp->ppht_full_for > 0) pp->ppht_full_for--;
        if (pp->ppht_flag_recent > 0) pp->ppht_flag_recent--;
        // Frames the fast kernel cannot do (more points than fit a CU's LDS) are rare: it raises a status word of their
        // slot and lc_pipe_wait launches the general kernel then.  No packet that is not needed: every one on this
        // stream is on the critical path.  (More than 64 angles or frames wider than 4096: the general kernel for all,
        // slot by slot -- it knows no frame sets.)
        if (!fast)
            for (int j = 0; j < bk.parts; j++) {
                if (!L.fs.count[j]) continue;
                const size_t fo = (size_t)j * p.max_batch, accn = (size_t)w.hg.numangle * w.hg.numrho;
                launch_ppht_general(w.g, L.fs.count[j], w.nz.p + fo * w.nz_cap, w.nz_cap, w.npoints.p + fo, w.rowplane.p + fo * w.g.rowplane_words(),
                                    w.tab.p, w.hg.numangle, w.hg.numrho, w.acc_ppht.p + fo * accn, p.threshold, pp->line_length, pp->line_gap,
                                    w.r_seg + fo * w.seg_cap * 4, w.seg_cap, w.r_segcount + fo, w.r_status + 8 * j, nullptr, ls,
                                    vote_fused ? w.acc_full.p + fo * accn : nullptr);
            }
    }
    stamp(pp, bk, EV_LINES, ls);
    if (!bk.zero_copy) {   // (banks of one slot) one copy: counts, rounds, status and (if wanted) the segments of the n frames
        const size_t res_ints = w.res_head + (p.want_segments ? (size_t)L.nframes * w.seg_cap * 4 : 0);
        (void)hipMemcpyAsync(bk.h_res, w.resblock.p, sizeof(int32_t) * res_ints, hipMemcpyDeviceToHost, ls);
        stamp(pp, bk, EV_END, ls);
    }
    for (int j = 0; j < bk.parts; j++) {
        if (!L.fs.count[j]) continue;
        Slot &s = first[j];
        (void)hipEventRecord(s.done, ls);
        s.fast_lines = fast;
    }
}

// First row of a frame the kernels of a launch at `level` read: a few rows above the first row classified (the front
// kernel warms its blur and Sobel windows up on them); 0 at the full level.
int first_row_read(const lc_pipe *pp, int level) {
    if (level >= pp->nlevels || pp->resizes) return 0;
    const int r0 = pp->row_split - 8;
    return r0 < 16 ? 0 : r0;
}

// copies rows [r0, r1) of each of the slot's n frames from the host buffer it was submitted from, on the upload stream
int upload_rows(lc_pipe *pipe, Slot &s, int r0, int r1) {
    if (r1 <= r0) return LC_OK;
    const size_t row_bytes = (size_t)3 * pipe->src_w, off = (size_t)r0 * row_bytes, bytes = (size_t)(r1 - r0) * row_bytes;
    uint8_t *dst = const_cast<uint8_t *>(s.input);
    if (r0 == 0 && r1 == pipe->src_h) {
        HIP_TRY(hipMemcpyAsync(dst, s.host_src, (size_t)s.nframes * pipe->src_frame_bytes, hipMemcpyHostToDevice, pipe->st_up));
        return LC_OK;
    }
    // (one row of the 2-D copy = the rows [r0, r1) of one frame, which lie back to back; its pitch = a whole frame)
    HIP_TRY(hipMemcpy2DAsync(dst + off, pipe->src_frame_bytes, s.host_src + off, pipe->src_frame_bytes, bytes, s.nframes,
                             hipMemcpyHostToDevice, pipe->st_up));
    return LC_OK;
}

// copies the tightest level's windows (lc_pipe::up_windows) of each of the slot's n frames: a 3-D copy per window (bytes of a
// row x rows x frames)
int upload_windows(lc_pipe *pipe, Slot &s) {
    const size_t pitch = (size_t)3 * pipe->src_w;
    for (const lc_pipe::UpWindow &uw : pipe->up_windows) {
        hipMemcpy3DParms c = {};
        c.srcPtr = make_hipPitchedPtr(const_cast<uint8_t *>(s.host_src), pitch, pitch, (size_t)pipe->src_h);
        c.dstPtr = make_hipPitchedPtr(const_cast<uint8_t *>(s.input), pitch, pitch, (size_t)pipe->src_h);
        c.srcPos = make_hipPos((size_t)uw.b0, (size_t)uw.r0, 0);
        c.dstPos = c.srcPos;
        c.extent = make_hipExtent((size_t)(uw.b1 - uw.b0), (size_t)(uw.r1 - uw.r0), (size_t)s.nframes);
        c.kind = hipMemcpyHostToDevice;
        HIP_TRY(hipMemcpy3DAsync(&c, pipe->st_up));
    }
    return LC_OK;
}

hipStream_t next_lines_stream(lc_pipe *pp) {
    if (!pp->p.want_segments) return pp->st_post;
    const unsigned turn = pp->lines_turn++ % (unsigned)(1 + pp->st_lines_more.size());
    return turn ? pp->st_lines_more[turn - 1] : pp->st_lines;
}

// the hysteresis of a launch as the arguments a front launch carries (see HystLaunch)
HystLaunch hyst_launch(lc_pipe *pp, const PendingTail &T) {
    Bank &bk = pp->banks[T.bank];
    Work &w = bk.w;
    const lc_pipe::SkipLevel *SL = T.skipping ? &pp->levels[T.L.level] : nullptr;
    HystLaunch h{};
    h.g = w.g; h.nframes = T.L.nframes;
    h.planeS = w.planeS.p; h.planeW = w.planeW.p; h.wake = w.wake.p;
    h.rounds = w.r_rounds; h.status = w.r_status;
    h.ring_words = SL ? SL->ring_words.p : nullptr; h.nring_words = SL ? SL->n_ring_words : 0;
    h.ring_code = SL ? SL->ring_code.p : nullptr;
    h.redo = w.redo.p;
    h.px_scratch = w.ring_px.p;
    h.fs = T.L.fs;
    h.wake_frames = w.B;
    return h;
}

// Everything of the pending launch (pp->pend) behind its front kernel: hysteresis -- unless the front launch just queued
// carried it (hysteresis_led) --, the second pass over flagged frames while such frames are about, finalize, point
// lists, HoughLinesP and the events that mark its slots done.
int finish_tail(lc_pipe *pp, bool hysteresis_led) {
    PendingTail T = pp->pend;
    pp->pend.active = false;
    if (!T.active) return LC_OK;
    Bank &bk = pp->banks[T.bank];
    Work &w = bk.w;
    const lc_pipe_params &p = pp->p;
    Slot *first = &pp->slots[(size_t)T.bank * pp->parts];
    const Launch &L = T.L;
    const int n = L.nframes;
    hipStream_t es = pp->st_edge, ps = pp->st_post, ls = T.ls;
    if (pp->tail_streams) ps = ls;            // the launch's short kernels share its HoughLinesP stream
    // hysteresis runs on hs: the post (or tail) stream, or the edge stream itself (behind its front kernel, before the next)
    hipStream_t hs = pp->hyst_on_edge ? es : ps;
    if (hs != es) HIP_TRY(hipStreamWaitEvent(hs, bk.ev[EV_FRONT], 0));
    // Hysteresis; with an ROI level active the frame's workgroup then proves that the units classified settle the ROI's
    // edges, or flags the frame (rare).  Flagged frames: all units again, and their hysteresis -- two launches that
    // leave at once for the other frames, queued only while flagged frames are about (lc_pipe::redo_hold).  (What the full
    // passes leave in the plane rows of units the level does not classify is cleared before the bank's next launch:
    // lc_pipe_wait sees the flag.)
    if (!hysteresis_led) {
        const lc_pipe::SkipLevel *SL = T.skipping ? &pp->levels[L.level] : nullptr;
        launch_hysteresis(w.g, n, w.planeS.p, w.planeW.p, w.wake.p, w.r_rounds, hs, w.r_status, nullptr, SL ? SL->ring_words.p : nullptr,
                          SL ? SL->n_ring_words : 0, SL ? SL->ring_code.p : nullptr, w.redo.p, &L.fs, w.B);
    }
    // The second pass runs where the launch's short kernels run -- with tail streams NOT on the edge stream, whose next
    // front kernel (another bank's) need not wait for it: the full hysteresis of a frame with a long weak chain is a
    // quarter of a millisecond of rounds on one workgroup.
    const bool redo_on_tail = pp->tail_streams && hs != ps && !getenv("LANECV_REDO_ON_EDGE");
    auto second_pass = [&](hipStream_t st) {
        const FrontSplit again = pp->split(pp->nlevels, 0, w.redo.p);
        launch_front(L.src[0], T.src_bytes, 1, p.color_code, p.blur != 0, w.g, n, pp->low, pp->high, w.planeS.p, w.planeW.p, w.wake.p,
                     nullptr, nullptr, st, pp->has_gate ? &pp->gate : nullptr, pp->gate_tab.p, w.gray.p, w.rowflags.p, T.fused_resize,
                     pp->front_rows, &again, &L.fs, L.src);
        launch_hysteresis(w.g, n, w.planeS.p, w.planeW.p, w.wake.p, w.r_rounds, st, nullptr, w.redo.p, nullptr, 0, nullptr, nullptr, &L.fs, w.B);
    };
    if (T.skipping && T.eager && !redo_on_tail) second_pass(hs);
    if (hs != ps || pp->timing) HIP_TRY(hipEventRecord(bk.ev[EV_HYST], hs));
    if (hs != ps) HIP_TRY(hipStreamWaitEvent(ps, bk.ev[EV_HYST], 0));
    if (T.skipping && T.eager && redo_on_tail) second_pass(ps);
    enqueue_post(pp, bk, L, ps);
    const int ready = (p.want_accumulator && !vote_is_fused(pp)) ? EV_VOTE : EV_POINTS;
    if (ls != ps || pp->timing) HIP_TRY(hipEventRecord(bk.ev[ready], ps));
    if (pp->timing && ready == EV_POINTS && !vote_is_fused(pp)) (void)hipEventRecord(bk.ev[EV_VOTE], ps);
    // (a stand-alone vote kernel reads the point list the general HoughLinesP kernel permutes: HoughLinesP behind it, then)
    if (ls != ps) HIP_TRY(hipStreamWaitEvent(ls, bk.ev[ready], 0));
    enqueue_lines(pp, bk, first, L, ls);
    HIP_TRY(hipGetLastError());
    for (int j = 0; j < bk.parts; j++)
        if (L.fs.count[j]) first[j].tail_queued = true;
    return LC_OK;
}

// Queues the kernels for the bank's pending slots: one launch each over all their frames.
int launch_bank(lc_pipe *pp, int bi) {
    Bank &bk = pp->banks[bi];
    Work &w = bk.w;
    const lc_pipe_params &p = pp->p;
    Slot *first = &pp->slots[(size_t)bi * pp->parts];
    Launch L;
    L.fs.per_part = p.max_batch;
    bool older = false, from_host = false;          // an earlier launch of this bank may still be running / frames are being uploaded
    for (int j = 0; j < bk.parts; j++) {
        const Slot &s = first[j];
        L.src[j] = s.input;
        older = older || (s.busy && !s.pending);
        if (!s.pending) continue;
        L.fs.count[j] = s.nframes;
        L.nframes = j * p.max_batch + s.nframes;
        from_host = from_host || s.from_host;
    }
    if (L.nframes == 0) return LC_OK;
    const int n = L.nframes;
    hipStream_t es = pp->st_edge;
    hipStream_t ls = next_lines_stream(pp);   // the launch's HoughLinesP stream (with tail streams: of its short kernels too)
    if (pp->level > 0 && pp->nlevels > 0) {
        if (pp->level_hold > 0) pp->level_hold--;
        else { pp->level--; pp->level_hold = 0; }     // try one level in again
    }
    L.level = pp->nlevels > 0 ? pp->level : 0;
    // (no clearing of plane rows beside a launch that may still be reading them: stay at the looser level this once)
    const int dirty_level = bk.redone ? pp->nlevels : bk.planes_level;
    if (older && L.level < dirty_level) L.level = std::min(dirty_level, pp->nlevels);
    // Frames submitted from host memory were uploaded from the first row the level expected at the time reads; if the
    // launch's level reads more, the rows above follow now.  Frames that are not there in full cannot be classified
    // again in the stream: lc_pipe_wait does that for the frames flagged, after fetching the rest of them.
    bool eager = true;
    for (int j = 0; j < bk.parts; j++) {
        Slot &s = first[j];
        if (!s.pending || !s.from_host) continue;
        const int need_from = first_row_read(pp, L.level);
        if (s.windowed && L.level != 0) {            // the level moved out since the submit: whole rows after all
            int rc;
            if ((rc = upload_rows(pp, s, need_from, pp->src_h))) return rc;
            s.rows_from = need_from;
            s.windowed = false;
        } else if (s.rows_from > need_from) {
            int rc;
            if ((rc = upload_rows(pp, s, need_from, s.rows_from))) return rc;
            s.rows_from = need_from;
        }
        eager = eager && s.rows_from == 0 && !s.windowed;
    }
    static const bool always_eager = getenv("LANECV_EAGER_REDO") && getenv("LANECV_EAGER_REDO")[0] == '1';   // diagnostics
    if (pp->redo_recent > 0) pp->redo_recent--;
    if (pp->redo_hold > 0) pp->redo_hold--;
    else if (!always_eager) eager = false;
    if (from_host) {
        HIP_TRY(hipEventRecord(bk.ev[EV_UPLOADED], pp->st_up));     // behind the copies of every pending slot
        HIP_TRY(hipStreamWaitEvent(es, bk.ev[EV_UPLOADED], 0));
    } else {
        stamp(pp, bk, EV_START, es);
        stamp(pp, bk, EV_UPLOADED, es);
    }
    stamp(pp, bk, EV_ESTART, es);
    // the exact reduction by three needs no pass of its own: the front kernel reads the source pixels it keeps
    const bool fused_resize = pp->fused_dec3();
    if (pp->resizes && !fused_resize) {             // (banks of one slot)
        launch_resize_linear(L.src[0], pp->src_w, pp->src_h, 3, n, w.bgr.p, w.g.W, w.g.H, pp->resize_tab.p, pp->resize_area2, es, pp->resize_dec3);
        L.src[0] = first[0].input = w.bgr.p;
    }
    const bool skipping = L.level < pp->nlevels;
    const FrontSplit skip = pp->split(L.level, 1, nullptr);
    // (what the tiled kernel may read of a one-slot bank's frames)
    const size_t src_bytes = (size_t)n * (pp->resizes && !fused_resize ? 3 * w.npix : pp->src_frame_bytes);
    if (pp->nlevels > 0) {
        if (L.level < dirty_level) {
            // Back at a tighter level, or after a batch some of whose frames had to be classified in full (see
            // lc_pipe_wait): the plane rows of the units this level does not need go first.  For every frame the bank
            // holds, not just this launch's: an earlier, larger batch left its bits in the frames behind this one's
            // too, and a later batch at this level would flood from them.  If all that happened since the last clearing
            // is frames classified again, only those frames: their redo flags still stand (a frame's flag is rewritten
            // by the next launch that covers the frame, which this clearing precedes on the edge stream).
            const bool flagged_only = bk.redone && L.level >= bk.planes_level;
            const FrontSplit clear_all = pp->split(L.level, 2, flagged_only ? w.redo.p : pp->ones.p);
            launch_front(L.src[0], src_bytes, 1, p.color_code, p.blur != 0, w.g, bk.parts * p.max_batch, pp->low, pp->high,
                         w.planeS.p, w.planeW.p, w.wake.p, nullptr, nullptr, es, nullptr, nullptr, nullptr, w.rowflags.p, fused_resize,
                         pp->front_rows, &clear_all);
        }
        bk.planes_level = L.level;
        bk.redone = false;
    }
    // The front kernel -- with the hysteresis of the launch before as its first workgroups, if that is still to come
    HystLaunch lead{};
    const bool with_lead = pp->pend.active && pp->hyst_fused && !w.inter;
    if (with_lead) lead = hyst_launch(pp, pp->pend);
    launch_front(L.src[0], src_bytes, 1, p.color_code, p.blur != 0, w.g, n, pp->low, pp->high, w.planeS.p, w.planeW.p,
                 w.wake.p, w.inter ? w.gray.p : nullptr, (w.inter && p.blur) ? w.blur.p : nullptr, es,
                 pp->has_gate ? &pp->gate : nullptr, pp->gate_tab.p, w.gray.p, w.rowflags.p, fused_resize, pp->front_rows,
                 pp->nlevels > 0 ? &skip : nullptr, &L.fs, L.src, with_lead ? &lead : nullptr);
    if (pp->pend.active) {
        int rc;
        if ((rc = finish_tail(pp, with_lead))) return rc;
    }
    PendingTail T;
    T.active = true; T.bank = bi; T.L = L; T.ls = ls; T.skipping = skipping; T.eager = eager; T.fused_resize = fused_resize; T.src_bytes = src_bytes;
    if (!pp->hyst_on_edge || pp->timing) HIP_TRY(hipEventRecord(bk.ev[EV_FRONT], es));     // (what a hysteresis kernel on another stream waits for)
    for (int j = 0; j < bk.parts; j++) {
        if (!L.fs.count[j]) continue;
        first[j].pending = false;
        first[j].level = L.level;
        first[j].timed = pp->timing;
        first[j].eager_redo = eager;
        first[j].tail_queued = false;
    }
    pp->pend = T;
    // Launches whose hysteresis can lead the next front launch leave their tail for then (or for lc_pipe_wait); the
    // others queue it now
    if (pp->hyst_fused && !w.inter) return LC_OK;
    return finish_tail(pp, false);
}

// The copy stream, created by the first host submit (a pipeline fed device frames never owns it).  Copies run on the
// DMA engines, but their packets still queue behind whatever shares the stream's hardware queue: picked like the rest.
int upload_stream(lc_pipe *pipe) {
    if (pipe->st_up) return LC_OK;
    for (Slot &s : pipe->slots)
        if (s.busy) { HIP_TRY(hipStreamCreateWithFlags(&pipe->st_up, hipStreamNonBlocking)); return LC_OK; }   // no experiments beside work in flight
    StreamPicker picker;
    std::vector<hipStream_t> mine = {pipe->st_edge};
    if (pipe->st_post != pipe->st_edge && pipe->st_post != pipe->st_lines) mine.push_back(pipe->st_post);
    if (pipe->st_lines != pipe->st_edge) mine.push_back(pipe->st_lines);
    for (hipStream_t st : pipe->st_lines_more) mine.push_back(st);
    bool own = true;
    pipe->st_up = picker.pick(mine, &own);
    if (!pipe->st_up) return fail(LC_EHIP, "hipStreamCreate failed for the upload stream");
    if (!own) pipe->shared_queues++;
    return LC_OK;
}

int get_slot(lc_pipe *pipe, int slot, Slot **out) {
    if (!pipe) return fail(LC_EINVAL, "null pipeline");
    if (slot < 0 || slot >= (int)pipe->slots.size()) return fail(LC_EINVAL, "slot %d out of range", slot);
    HIP_TRY(hipSetDevice(pipe->ctx->device));
    *out = &pipe->slots[slot];
    return LC_OK;
}

// the slot's own part of its bank's input buffer
uint8_t *own_input(lc_pipe *pipe, const Slot &s) {
    Work &w = pipe->banks[s.bank].w;
    return (pipe->resizes ? w.src_in.p : w.bgr.p) + (size_t)s.pos * pipe->p.max_batch * pipe->src_frame_bytes;
}

// a submit: the slot joins its bank's pending ones, and the bank's kernels are queued once they are all there
int submit(lc_pipe *pipe, Slot &s) {
    s.busy = s.pending = true;
    const Bank &bk = pipe->banks[s.bank];
    const Slot *first = &pipe->slots[(size_t)s.bank * pipe->parts];
    int pending = 0;
    for (int j = 0; j < bk.parts; j++) pending += first[j].pending ? 1 : 0;
    if (pending < bk.parts) return LC_OK;
    const int rc = launch_bank(pipe, s.bank);
    if (rc) s.busy = s.pending = false;
    return rc;
}

// Polls `ev` and gives up after LANECV_WAIT_TIMEOUT_S seconds (default 120; 0 = block in hipEventSynchronize for ever)
// with a report of which of the pipeline's slots have finished: a pipeline whose streams starve each other (more
// streams in the process than hardware queues, see lc_pipe_create) then fails loudly instead of hanging its caller.
int wait_event(lc_pipe *pipe, int slot, hipEvent_t ev) {
    static const double wait_limit = getenv("LANECV_WAIT_TIMEOUT_S") ? atof(getenv("LANECV_WAIT_TIMEOUT_S")) : 120.0;
    if (!(wait_limit > 0)) {
        HIP_TRY(hipEventSynchronize(ev));
        return LC_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return LC_OK;
        if (q != hipErrorNotReady) HIP_TRY(q);
        if ((spins & 63) != 63) continue;                     // the clock is read every 64th query
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > wait_limit) {
            std::string rep;
            for (size_t k = 0; k < pipe->slots.size(); k++) {
                Slot &o = pipe->slots[k];
                rep += " slot" + std::to_string(k) + (o.busy ? (o.pending ? "(pending)" : "*") : "") + ":";
                rep += hipEventQuery(o.done) == hipSuccess ? '1' : '0';
            }
            if (pipe->timing)
                for (size_t b = 0; b < pipe->banks.size(); b++) {
                    rep += " bank" + std::to_string(b) + ":";
                    for (auto &e : pipe->banks[b].ev) rep += hipEventQuery(e) == hipSuccess ? '1' : '0';
                }
            (void)hipGetLastError();
            return fail(LC_EHIP, "wait: slot %d not finished after %.0f s; done events fired per slot (* = in flight)%s%s", slot, wait_limit,
                        pipe->timing ? ", stage events per bank (START UPLOADED FRONT HYST EDGE POINTS VOTE LINES END ESTART LSTART VSTART):" : ":",
                        rep.c_str());
        }
        if (waited > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(50));   // a batch takes about a millisecond
    }
}

// Frames of slot s whose edges the units near the ROI did not settle, when the launch could not classify them again
// itself (they were not in device memory in full): the rows that were not uploaded are fetched for the flagged frames
// from the host buffer the slot was submitted from, then all units of those frames once more, their hysteresis, and
// everything behind it for the slot's whole batch -- finalize rewrites what changed, the point lists and HoughLinesP
// are made again.  On the post stream, in order behind whatever else is queued there.
int enqueue_repair(lc_pipe *pp, Slot &s) {
    Bank &bk = pp->banks[s.bank];
    Work &w = bk.w;
    const lc_pipe_params &p = pp->p;
    Slot *first = &pp->slots[(size_t)s.bank * pp->parts];
    hipStream_t ps = pp->st_post;
    if (s.from_host && (s.rows_from > 0 || s.windowed)) {
        const size_t bytes = s.windowed ? pp->src_frame_bytes : (size_t)s.rows_from * 3 * pp->src_w;
        for (int i = 0; i < s.nframes; i++)
            if ((s.h_rounds[i] >> 30) & 1)
                HIP_TRY(hipMemcpyAsync(const_cast<uint8_t *>(s.input) + (size_t)i * pp->src_frame_bytes, s.host_src + (size_t)i * pp->src_frame_bytes,
                                       bytes, hipMemcpyHostToDevice, pp->st_up));
        HIP_TRY(hipEventRecord(bk.ev[EV_UPLOADED], pp->st_up));
        HIP_TRY(hipStreamWaitEvent(ps, bk.ev[EV_UPLOADED], 0));
    }
    Launch L;
    L.fs.per_part = p.max_batch;
    for (int j = 0; j < bk.parts; j++) L.src[j] = first[j].input;
    L.fs.count[s.pos] = s.nframes;
    L.nframes = s.pos * p.max_batch + s.nframes;
    L.level = pp->nlevels;
    const FrontSplit again = pp->split(pp->nlevels, 0, w.redo.p);
    const size_t src_bytes = (size_t)L.nframes * (pp->resizes && !pp->fused_dec3() ? 3 * w.npix : pp->src_frame_bytes);
    launch_front(L.src[0], src_bytes, 1, p.color_code, p.blur != 0, w.g, L.nframes, pp->low, pp->high,
                 w.planeS.p, w.planeW.p, w.wake.p, nullptr, nullptr, ps, pp->has_gate ? &pp->gate : nullptr, pp->gate_tab.p, w.gray.p,
                 w.rowflags.p, pp->fused_dec3(), pp->front_rows, &again, &L.fs, L.src);
    launch_hysteresis(w.g, L.nframes, w.planeS.p, w.planeW.p, w.wake.p, w.r_rounds, ps, nullptr, w.redo.p, nullptr, 0, nullptr, nullptr, &L.fs, w.B);
    enqueue_post(pp, bk, L, ps);
    enqueue_lines(pp, bk, first, L, ps);
    HIP_TRY(hipGetLastError());
    return LC_OK;
}

}  // namespace

namespace lanecv {
void pipe_flush_pending(lc_pipe *pipe) {
    for (size_t b = 0; b < pipe->banks.size(); b++) (void)launch_bank(pipe, (int)b);
    (void)finish_tail(pipe, false);
}
}  // namespace lanecv

extern "C" {

int lc_pipe_submit_host(lc_pipe *pipe, int slot, const uint8_t *host_bgr, int nframes) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!host_bgr || nframes <= 0 || nframes > pipe->p.max_batch) return fail(LC_EINVAL, "submit: bad frames argument");
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight; call lc_pipe_wait first", slot);
    s->nframes = nframes;
    s->input = own_input(pipe, *s);
    s->from_host = true;
    s->host_src = host_bgr;
    if ((rc = upload_stream(pipe))) return rc;
    if (pipe->timing) HIP_TRY(hipEventRecord(pipe->banks[s->bank].ev[EV_START], pipe->st_up));
    // Only the rows the kernels will read: with an ROI level active, those from a little above the first row classified
    // (the level is the one the pipeline is at now; launch_bank fetches the rest should the launch need it, lc_pipe_wait
    // for the rare frame that must be classified in full after all).  LANECV_FULL_UPLOAD=1: whole frames always.
    static const bool full_upload_env = getenv("LANECV_FULL_UPLOAD") && getenv("LANECV_FULL_UPLOAD")[0] == '1';
    // Frames in ordinary (pageable) host memory go up whole, as one linear copy: 2-D and 3-D copies from such memory run
    // row by row (milliseconds for a single 960 x 540 frame), which costs far more than the bytes they save.
    bool pinned = false;
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, host_bgr) == hipSuccess) pinned = at.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    const bool full_upload = full_upload_env || !pinned;
    s->rows_from = full_upload ? 0 : first_row_read(pipe, pipe->nlevels > 0 ? pipe->level : 0);
    s->windowed = !full_upload && pipe->nlevels > 0 && pipe->level == 0 && !pipe->up_windows.empty() && s->rows_from > 0 &&
                  !getenv("LANECV_NO_UPLOAD_WINDOWS");
    if (s->windowed) rc = upload_windows(pipe, *s);
    else rc = upload_rows(pipe, *s, s->rows_from, pipe->src_h);
    if (rc) return rc;
    return submit(pipe, *s);
}

int lc_pipe_submit_device(lc_pipe *pipe, int slot, const void *dev_bgr, int nframes) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!dev_bgr || nframes <= 0 || nframes > pipe->p.max_batch) return fail(LC_EINVAL, "submit: bad frames argument");
    // (the slot's own input buffer, lc_pipe_input_device_ptr, is taken as it is: slots of a bank lie back to back in one buffer,
    // and with an odd frame size the later ones start on a multiple of 4 only -- which is all the kernels need)
    if (((uintptr_t)dev_bgr & 15) != 0 && dev_bgr != own_input(pipe, *s)) return fail(LC_EINVAL, "submit: device frames must be 16-byte aligned");
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight; call lc_pipe_wait first", slot);
    s->nframes = nframes;
    s->input = (const uint8_t *)dev_bgr;
    s->from_host = false;
    s->windowed = false;
    s->rows_from = 0;
    return submit(pipe, *s);
}

int lc_pipe_wait(lc_pipe *pipe, int slot, int32_t *seg_xyxy, int32_t *seg_count) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!s->busy) return fail(LC_EINVAL, "slot %d has nothing in flight", slot);
    const lc_pipe_params &p = pipe->p;
    if (s->pending && (rc = launch_bank(pipe, s->bank))) return rc;      // its bank is not full: what is there goes now
    if (!s->tail_queued && (rc = finish_tail(pipe, false))) return rc;   // its hysteresis was waiting for a front launch to lead
    if ((rc = wait_event(pipe, slot, s->done))) return rc;
    // Frames the ROI scheme could not settle from the units near the ROI: every frame's workgroup says so in its own word
    // of the result block (bit 30 of the rounds word).  Not one word for the batch: the hysteresis launch also clears the
    // batch's status words, from the first frame's workgroup, and two workgroups on different XCDs writing one word of
    // device memory in one launch reach it in no particular order -- the flag of a later frame was lost that way.
    int redone = 0;
    if (s->level < pipe->nlevels)
        for (int i = 0; i < s->nframes; i++) redone += (s->h_rounds[i] >> 30) & 1;
    if (redone > 0) {
        // They were (or are now) classified again in full.  Their full passes store plane rows in units the level does not
        // classify: cleared before the bank's next launch.  And a batch with many of them moves the pipeline a level out.
        pipe->banks[s->bank].redone = true;
        // the launches queue the second pass themselves only when such frames come again within a few launches: the two
        // extra launches cost every batch a tenth of its tail (the hysteresis workgroups that look the flags up wait for
        // room like any other), an isolated frame is cheaper made up for here
        if (pipe->redo_recent > 0) pipe->redo_hold = 64;
        pipe->redo_recent = 24;
        const char *ob = getenv("LANECV_SKIP_OFF_BATCHES");         // diagnostics / tests; 0 = never move a level out
        const int off_batches = ob ? atoi(ob) : 256;
        if (4 * redone > s->nframes && off_batches > 0 && pipe->level <= s->level) {
            pipe->level = s->level + 1;
            pipe->level_hold = off_batches;
        }
        if (!s->eager_redo) {            // the launch could not do it: the frames were not in device memory in full
            if ((rc = enqueue_repair(pipe, *s))) return rc;
            if ((rc = wait_event(pipe, slot, s->done))) return rc;
        }
    }
    if (p.want_segments && s->fast_lines && s->h_status[4]) {
        // Frames with more edge points than the LDS-resident HoughLinesP kernel takes (it raised a status word of the
        // slot): the general kernel, for them.  Left out of the submit, where nearly always it would have been a packet
        // on the HoughLinesP stream that finds nothing to do.
        Work &w = pipe->banks[s->bank].w;
        const bool vote_fused = vote_is_fused(pipe);
        hipStream_t ps = pipe->st_post;
        const size_t fo = (size_t)s->pos * p.max_batch, accn = (size_t)w.hg.numangle * w.hg.numrho;
        // first the LDS kernel's forms that read the mask, then the points as well, from device memory (frames whose points
        // are too scattered for their mask words to fit beside points and accumulator, or too many), then the general
        // kernel for what is still flagged
        // (the frames to do, by number, from the marks the kernel left in their segment counts: a launch of as many
        // workgroups, not one per frame of the batch -- each of them must find an empty CU first)
        FlagList list{};
        int nflagged = 0;
        for (int i = 0; i < s->nframes; i++)
            if (s->h_segcount[i] < 0) {
                if (nflagged < (int)(sizeof(list.f) / sizeof(list.f[0]))) list.f[nflagged] = (uint16_t)i;
                nflagged++;
            }
        list.n = nflagged <= (int)(sizeof(list.f) / sizeof(list.f[0])) ? nflagged : 0;
        // Three tries, each with a whole CU's LDS: everything in LDS as in the ordinary launch (a frame that only outgrew HALF
        // a CU's LDS: walks along lines through a mask in LDS are the fastest), the mask in device memory, the points too.
        // (A first try with the 79 KiB that are free beside one frame of a shared-CU launch, so that the workgroup need not
        // wait for an empty CU, changed nothing measurable: the frame's own HoughLinesP is the time.)
        for (int state = (list.n && pipe->ppht_two_per_cu) ? 0 : 1; state <= 2; state++)
            (void)launch_ppht_fast(w.g, s->nframes, w.nz.p + fo * w.nz_cap, w.nz_cap, w.npoints.p + fo, w.rowplane.p + fo * w.g.rowplane_words(), w.tab.p,
                                   w.hg.numangle, w.hg.numrho, p.threshold, pipe->line_length, pipe->line_gap, w.r_seg + fo * w.seg_cap * 4, w.seg_cap,
                                   w.r_segcount + fo, w.r_status + 8 * s->pos, w.fallback.p + fo, w.rng_out.p, w.rng_len,
                                   vote_fused ? w.acc_full.p + fo * accn : nullptr, ps, 0, true, 8, 0, nullptr, state, &list);
        launch_ppht_general(w.g, s->nframes, w.nz.p + fo * w.nz_cap, w.nz_cap, w.npoints.p + fo, w.rowplane.p + fo * w.g.rowplane_words(), w.tab.p,
                            w.hg.numangle, w.hg.numrho, w.acc_ppht.p + fo * accn, p.threshold, pipe->line_length, pipe->line_gap,
                            w.r_seg + fo * w.seg_cap * 4, w.seg_cap, w.r_segcount + fo, w.r_status + 8 * s->pos, w.fallback.p + fo, ps,
                            vote_fused ? w.acc_full.p + fo * accn : nullptr);
        if (!pipe->banks[s->bank].zero_copy) {
            const size_t res_ints = w.res_head + (size_t)s->nframes * w.seg_cap * 4;
            HIP_TRY(hipMemcpyAsync(pipe->banks[s->bank].h_res, w.resblock.p, sizeof(int32_t) * res_ints, hipMemcpyDeviceToHost, ps));
        }
        HIP_TRY(hipEventRecord(s->done, ps));
        HIP_TRY(hipGetLastError());
        if ((rc = wait_event(pipe, slot, s->done))) return rc;
    }
    s->busy = false;
    if (p.want_segments && s->h_status[2]) {       // a frame outgrew half a CU's LDS (see lc_pipe)
        // Once in a while: the frame was done again just now, with a whole CU's LDS, and that was that.  Twice within a few
        // launches: the content is like that, and every batch would pay for its flagged frames when it is waited for --
        // whole-CU launches for the next 64, which cost every batch a little instead.  LANECV_PPHT_FULL_AT_ONCE=1: at the first.
        static const bool at_once = getenv("LANECV_PPHT_FULL_AT_ONCE") != nullptr;
        if (pipe->ppht_flag_recent > 0 || at_once) pipe->ppht_full_for = 64;
        pipe->ppht_flag_recent = 24;
    }
    if (p.want_segments || p.want_accumulator || p.want_points) {
        if (s->h_status[0])
            return fail(LC_ECAP, "a frame holds more edge points than the pipeline keeps per frame (%d): its point list, accumulator and "
                        "segments are incomplete", pipe->banks[s->bank].w.nz_cap);
        if (s->h_status[1])
            return fail(LC_EINVAL, "HoughLinesP: a vote fell outside the accumulator (the angle table does not fit the image size)");
    }
    if (p.want_segments) {
        if (seg_count) memcpy(seg_count, s->h_segcount, sizeof(int) * s->nframes);
        if (seg_xyxy) memcpy(seg_xyxy, s->h_seg, sizeof(int32_t) * 4 * (size_t)s->nframes * pipe->banks[s->bank].w.seg_cap);
    }
    return LC_OK;
}

int lc_pipe_set_timing(lc_pipe *pipe, int on) {
    if (!pipe) return fail(LC_EINVAL, "null pipeline");
    if (on && !pipe->ev_base) {          // the origin of lc_pipe_event_times
        HIP_TRY(hipSetDevice(pipe->ctx->device));
        HIP_TRY(hipEventCreate(&pipe->ev_base));
        HIP_TRY(hipEventRecord(pipe->ev_base, pipe->st_edge));
        HIP_TRY(hipEventSynchronize(pipe->ev_base));
    }
    pipe->timing = on != 0;
    return LC_OK;
}

int lc_pipe_read(lc_pipe *pipe, int slot, int what, void *host_dst, size_t bytes) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight", slot);
    if (!host_dst) return fail(LC_EINVAL, "null destination");
    Work &w = pipe->banks[s->bank].w;
    const uint8_t *src = nullptr;
    size_t need = 0, off = 0;
    const size_t img = (size_t)s->nframes * w.npix, fo = (size_t)s->pos * pipe->p.max_batch;     // fo: the slot's first frame in the bank
    const size_t accn = (size_t)w.hg.numangle * w.hg.numrho;
    switch (what) {
        case LC_BUF_GRAY: src = w.inter ? w.gray.p : nullptr; need = img; off = fo * w.npix; break;
        case LC_BUF_BLUR: src = (w.inter && pipe->p.blur) ? w.blur.p : nullptr; need = img; off = fo * w.npix; break;
        case LC_BUF_EDGES: src = w.inter ? w.edges.p : nullptr; need = img; off = fo * w.npix; break;
        case LC_BUF_MASKED: src = w.masked.p; need = img; off = fo * w.npix; break;
        case LC_BUF_ACC: src = pipe->p.want_accumulator ? reinterpret_cast<const uint8_t *>(w.acc_full.p) : nullptr;
            need = sizeof(int32_t) * (size_t)s->nframes * accn; off = sizeof(int32_t) * fo * accn; break;
        case LC_BUF_NPOINTS:
            src = (pipe->p.want_accumulator || pipe->p.want_segments || pipe->p.want_points) ? reinterpret_cast<const uint8_t *>(w.npoints.p) : nullptr;
            need = sizeof(int) * s->nframes; off = sizeof(int) * fo; break;
        case LC_BUF_RESIZED:
            if (pipe->fused_dec3())
                return fail(LC_EINVAL, "the resized frames are not materialised when the reduction by three is fused into the edge "
                            "kernel; configure keep_intermediates to read them");
            src = pipe->resizes ? w.bgr.p : nullptr; need = 3 * img; off = 3 * fo * w.npix; break;
        default: return fail(LC_EINVAL, "unknown buffer id %d", what);
    }
    if (!src) return fail(LC_EINVAL, "buffer %d is not produced by this pipeline's configuration", what);
    if (bytes < need) return fail(LC_ECAP, "destination holds %zu bytes, %zu needed", bytes, need);
    HIP_TRY(hipMemcpy(host_dst, src + off, need, hipMemcpyDeviceToHost));
    return LC_OK;
}

int lc_pipe_read_points(lc_pipe *pipe, int slot, int frame, uint32_t *xy, int capacity, int *n) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (s->busy) return fail(LC_EINVAL, "slot %d has not been waited for", slot);
    if (!n) return fail(LC_EINVAL, "null count pointer");
    if (!(pipe->p.want_accumulator || pipe->p.want_segments || pipe->p.want_points))
        return fail(LC_EINVAL, "this pipeline's configuration does not produce the point list");
    if (frame < 0 || frame >= s->nframes) return fail(LC_EINVAL, "frame %d out of range [0, %d)", frame, s->nframes);
    Work &w = pipe->banks[s->bank].w;
    const size_t f = (size_t)s->pos * pipe->p.max_batch + frame;
    int np = 0;
    HIP_TRY(hipMemcpy(&np, w.npoints.p + f, sizeof(int), hipMemcpyDeviceToHost));
    *n = np;
    if (np > w.nz_cap) return fail(LC_ECAP, "frame %d has %d edge points, the slot keeps %d", frame, np, w.nz_cap);
    if (np > capacity) return fail(LC_ECAP, "frame %d has %d edge points, destination holds %d", frame, np, capacity);
    if (np > 0) {
        if (!xy) return fail(LC_EINVAL, "null destination");
        HIP_TRY(hipMemcpy(xy, w.nz.p + f * w.nz_cap, sizeof(uint32_t) * np, hipMemcpyDeviceToHost));
    }
    return LC_OK;
}

int lc_pipe_stage_ms(lc_pipe *pipe, int slot, int stage, float *ms) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!ms) return fail(LC_EINVAL, "null output");
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight", slot);
    if (!s->timed) return fail(LC_EINVAL, "slot %d: the stage times of its last batch were not recorded (lc_pipe_set_timing)", slot);
    Bank &bk = pipe->banks[s->bank];
    int a, b;
    switch (stage) {
        case LC_STAGE_UPLOAD: a = EV_START; b = EV_UPLOADED; break;
        case LC_STAGE_EDGE: a = EV_ESTART; b = EV_EDGE; break;
        case LC_STAGE_POINTS: a = EV_EDGE; b = EV_POINTS; break;
        case LC_STAGE_VOTE:   // with segments wanted the accumulator comes out of the HoughLinesP kernel: counted under LINES
            if (pipe->p.want_accumulator && pipe->p.want_segments) { *ms = 0.f; return LC_OK; }
            a = EV_POINTS; b = EV_VOTE;
            break;
        case LC_STAGE_LINES: a = EV_LSTART; b = EV_LINES; break;
        case LC_STAGE_TOTAL: a = EV_START; b = bk.zero_copy ? EV_LINES : EV_END; break;
        case LC_STAGE_FRONT: a = EV_ESTART; b = EV_FRONT; break;
        case LC_STAGE_HYST: a = EV_FRONT; b = EV_HYST; break;
        case LC_STAGE_FINAL: a = EV_HYST; b = EV_EDGE; break;
        default: return fail(LC_EINVAL, "unknown stage %d", stage);
    }
    // (the events are those of the bank's latest launch: with several slots to a bank, of the launch that covered them all)
    HIP_TRY(hipEventElapsedTime(ms, bk.ev[a], bk.ev[b]));
    return LC_OK;
}

int lc_pipe_event_times(lc_pipe *pipe, int slot, float *ms12) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!ms12) return fail(LC_EINVAL, "null output");
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight", slot);
    if (!s->timed || !pipe->ev_base) return fail(LC_EINVAL, "slot %d: the events of its last batch were not recorded (lc_pipe_set_timing)", slot);
    for (int k = 0; k < 12; k++) {
        float ms = -1.f;
        if (hipEventElapsedTime(&ms, pipe->ev_base, pipe->banks[s->bank].ev[k]) != hipSuccess) { ms = -1.f; (void)hipGetLastError(); }
        ms12[k] = ms;
    }
    return LC_OK;
}

int lc_pipe_hyst_rounds(lc_pipe *pipe, int slot, int *rounds_max, int *rounds_total) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight", slot);
    int mx = 0, tot = 0;
    for (int i = 0; i < s->nframes; i++) { const int r = s->h_rounds[i] & 0x3FFFFFFF; mx = std::max(mx, r); tot += r; }
    if (rounds_max) *rounds_max = mx;
    if (rounds_total) *rounds_total = tot;
    return LC_OK;
}

int lc_pipe_roi_skip(lc_pipe *pipe, int slot, int *first_row, long long *pixels_classified, int *frames_redone) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight", slot);
    if (first_row) *first_row = pipe->row_split;
    if (pixels_classified) *pixels_classified = s->level < pipe->nlevels ? pipe->levels[s->level].pixels : pipe->frame_pixels;
    if (frames_redone) {
        *frames_redone = 0;
        if (pipe->nlevels > 0 && s->level >= pipe->nlevels) *frames_redone = s->nframes;     // classified in full from the start
        else if (pipe->nlevels > 0)
            for (int i = 0; i < s->nframes; i++) *frames_redone += (s->h_rounds[i] >> 30) & 1;
    }
    return LC_OK;
}

int lc_pipe_rows_uploaded(lc_pipe *pipe, int slot, int *first_row) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!first_row) return fail(LC_EINVAL, "null output");
    *first_row = s->from_host ? s->rows_from : 0;
    return LC_OK;
}

int lc_pipe_bytes_uploaded(lc_pipe *pipe, int slot, long long *bytes_per_frame) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!bytes_per_frame) return fail(LC_EINVAL, "null output");
    if (!s->from_host) *bytes_per_frame = 0;
    else if (s->windowed) *bytes_per_frame = pipe->up_window_bytes;
    else *bytes_per_frame = (long long)(pipe->src_h - s->rows_from) * 3 * pipe->src_w;
    return LC_OK;
}

int lc_pipe_shared_queues(lc_pipe *pipe, int *count) {
    if (!pipe || !count) return fail(LC_EINVAL, "null argument");
    *count = pipe->shared_queues;
    return LC_OK;
}

int lc_pipe_layout(lc_pipe *pipe, int *group_size, int *lines_streams, int *roi_levels) {
    if (!pipe) return fail(LC_EINVAL, "null pipeline");
    if (group_size) *group_size = pipe->parts;
    if (lines_streams) *lines_streams = pipe->p.want_segments ? 1 + (int)pipe->st_lines_more.size() : 0;
    if (roi_levels) *roi_levels = pipe->nlevels;
    return LC_OK;
}

int lc_pipe_upload(lc_pipe *pipe, int slot, const uint8_t *host_bgr, int nframes) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!host_bgr || nframes <= 0 || nframes > pipe->p.max_batch) return fail(LC_EINVAL, "upload: bad frames argument");
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight; call lc_pipe_wait first", slot);
    if ((rc = upload_stream(pipe))) return rc;
    HIP_TRY(hipMemcpyAsync(own_input(pipe, *s), host_bgr, (size_t)nframes * pipe->src_frame_bytes, hipMemcpyHostToDevice, pipe->st_up));
    HIP_TRY(hipStreamSynchronize(pipe->st_up));
    return LC_OK;
}

int lc_pipe_input_device_ptr(lc_pipe *pipe, int slot, void **dev_ptr) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!dev_ptr) return fail(LC_EINVAL, "null output");
    *dev_ptr = own_input(pipe, *s);
    return LC_OK;
}

}  // extern "C"
