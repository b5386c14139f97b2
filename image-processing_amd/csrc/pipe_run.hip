// The batched pipeline at run time: what a submit queues on the pipeline's streams, the wait, and the read-backs.
#include "pipe.h"

#include <chrono>
#include <thread>

using namespace lanecv;

namespace {

void enqueue_lines(lc_pipe *pp, Slot &s, hipEvent_t after);
void flush_group(lc_pipe *pp, int g);

void enqueue_back(lc_pipe *pp, Slot &s) {
    Work &w = s.w;
    const lc_pipe_params &p = pp->p;
    const int n = s.nframes;
    hipStream_t es = pp->st_post;
    const bool hough = p.want_accumulator || p.want_segments;
    const bool points = hough || p.want_points;
    if (!points) (void)hipMemsetAsync(w.rowcount.p, 0, sizeof(int) * (size_t)n * w.g.H, es);   // nobody consumes (and clears) the counts
    launch_finalize(w.g, n, w.planeS.p, pp->has_roi ? pp->roi_bits.p : nullptr, w.inter ? w.edges.p : nullptr, w.masked.p,
                    w.rowplane.p, w.rowcount.p, es, w.inter ? nullptr : w.prevbits.p);
    if (s.level < pp->nlevels) {
        // frames that were classified in full: the plane rows of the units no frame needs are cleared again, now that
        // they have been read (a few workgroups that look at the flags and, nearly always, leave)
        const FrontSplit clear = pp->split(s.level, 2, w.redo.p);
        launch_front(s.input, s.input_bytes, 1, p.color_code, p.blur != 0, w.g, n, pp->low, pp->high, w.planeS.p, w.planeW.p, w.wake.p,
                     nullptr, nullptr, es, nullptr, nullptr, nullptr, w.rowflags.p, pp->fused_dec3(), pp->front_rows, &clear);
    }
    (void)hipEventRecord(s.ev[EV_EDGE], es);
    if (points) launch_points(w.g, n, w.rowplane.p, w.rowcount.p, w.nz.p, w.nz_cap, w.npoints.p, w.r_status, es);
    // grouped slots: the group's HoughLinesP launch covers every slot's full capacity, and waits for EV_POINTS only --
    // the frames beyond this batch must read as empty by then
    if (p.want_segments && pp->group_size > 1 && n < p.max_batch)
        (void)hipMemsetAsync(w.npoints.p + n, 0, sizeof(int) * (size_t)(p.max_batch - n), es);
    (void)hipEventRecord(s.ev[EV_POINTS], es);
    // The full-vote accumulator depends only on the point list.  With segments wanted, the HoughLinesP kernel
    // produces it on the side (its helper wavefronts, from the same LDS-resident set-up) and the vote kernel
    // only runs for the frames that kernel flags; it sits between the two HoughLinesP steps because the
    // general one permutes the point list.  Stage times: VOTE = that flagged launch, LINES = both steps.
    const bool vote_fused = p.want_accumulator && p.want_segments && pp->fuse_vote;
    if (p.want_accumulator && !vote_fused)
        launch_vote(w.g, n, w.nz.p, w.nz_cap, w.npoints.p, w.tab.p, w.hg.numangle, w.hg.numrho, w.acc_full.p, nullptr, es);
    if (!vote_fused) (void)hipEventRecord(s.ev[EV_VOTE], es);
    if (p.want_segments && pp->group_size > 1) {
        // grouped slots: the HoughLinesP launch waits for the group's last slot (or for a wait on one of its slots)
        lc_pipe::Group &G = pp->groups[s.group];
        G.pending.push_back((int)(&s - &pp->slots[0]));
        s.lines_pending = true;
        const int first = s.group * pp->group_size, gs = std::min(pp->group_size, (int)pp->slots.size() - first);
        if ((int)G.pending.size() >= gs) flush_group(pp, s.group);
        return;
    }
    if (p.want_segments && pp->st_lines != es && pp->defer_after && pp->slots.size() > 1) pp->deferred = (int)(&s - &pp->slots[0]);
    else enqueue_lines(pp, s, nullptr);
}

// One HoughLinesP launch for the slots of a group whose points are queued: their frames lie back to back in the
// group's buffers (slot j at frame j * max_batch), the launch covers them all.  What the launch has to report about a
// frame it raises in the status words of the frame's own slot (the slots' status blocks lie back to back too), which
// that slot's hysteresis kernel zeroed earlier on in the batch: a slot never sees another slot's flags, and nothing is
// cleared from the host.
void flush_group(lc_pipe *pp, int g) {
    lc_pipe::Group &G = pp->groups[g];
    if (G.pending.empty()) return;
    const lc_pipe_params &p = pp->p;
    const int first = g * pp->group_size, m = (int)G.pending.size();
    bool prefix = true;
    for (int i = 0; i < m; i++) prefix = prefix && G.pending[i] == first + i;
    if (!prefix) {      // slots used out of order: a launch each
        for (int k : G.pending) { pp->slots[k].lines_pending = false; enqueue_lines(pp, pp->slots[k], nullptr); }
        G.pending.clear();
        return;
    }
    Slot &lead = pp->slots[first];
    Work &w = lead.w;
    const int n_total = (m - 1) * p.max_batch + pp->slots[first + m - 1].nframes;
    const bool vote_fused = p.want_accumulator && p.want_segments && pp->fuse_vote;
    const unsigned turn = pp->lines_turn++ % (unsigned)(1 + pp->st_lines_more.size());
    hipStream_t ls = turn ? pp->st_lines_more[turn - 1] : pp->st_lines;
    for (int i = 0; i < m; i++) {
        Slot &s = pp->slots[first + i];
        if (ls != pp->st_post) (void)hipStreamWaitEvent(ls, s.ev[(p.want_accumulator && !vote_fused) ? EV_VOTE : EV_POINTS], 0);
    }
    for (int i = 0; i < m; i++) (void)hipEventRecord(pp->slots[first + i].ev[EV_LSTART], ls);
    const bool fast = launch_ppht_fast(w.g, n_total, w.nz.p, w.nz_cap, w.npoints.p, w.rowplane.p, w.tab.p, w.hg.numangle, w.hg.numrho,
                                       p.threshold, pp->line_length, pp->line_gap, w.r_seg, w.seg_cap, w.r_segcount, w.r_status,
                                       w.fallback.p, w.rng_out.p, RNG_TABLE, vote_fused ? w.acc_full.p : nullptr, ls,
                                       (pp->ppht_two_per_cu && pp->ppht_full_for == 0) ? ppht_lds_half() : 0, false,
                                       2 * n_total > pp->cu_count ? 0 : 8, p.max_batch);
    if (pp->ppht_full_for > 0) pp->ppht_full_for--;
    launch_ppht_general(w.g, n_total, w.nz.p, w.nz_cap, w.npoints.p, w.rowplane.p, w.tab.p, w.hg.numangle, w.hg.numrho, w.acc_ppht.p,
                        p.threshold, pp->line_length, pp->line_gap, w.r_seg, w.seg_cap, w.r_segcount, w.r_status,
                        fast ? w.fallback.p : nullptr, ls, vote_fused ? w.acc_full.p : nullptr, p.max_batch);
    for (int i = 0; i < m; i++) {
        Slot &s = pp->slots[first + i];
        (void)hipEventRecord(s.ev[EV_LINES], ls);
        s.end_ev = EV_LINES;
        s.lines_pending = false;
    }
    G.pending.clear();
}

// HoughLinesP step and the copy of the batch's results; `after` (may be null) is a later batch's event to start behind
void enqueue_lines(lc_pipe *pp, Slot &s, hipEvent_t after) {
    Work &w = s.w;
    const lc_pipe_params &p = pp->p;
    const int n = s.nframes;
    hipStream_t es = pp->st_post;
    const bool vote_fused = p.want_accumulator && p.want_segments && pp->fuse_vote;
    hipStream_t ls = es;   // the stream the batch finishes on
    if (p.want_segments) {
        const unsigned turn = pp->lines_turn++ % (unsigned)(1 + pp->st_lines_more.size());
        ls = turn ? pp->st_lines_more[turn - 1] : pp->st_lines;
        // (a stand-alone vote kernel reads the point list the general kernel permutes: behind it, then)
        if (ls != es) (void)hipStreamWaitEvent(ls, s.ev[(p.want_accumulator && !vote_fused) ? EV_VOTE : EV_POINTS], 0);
        if (ls != es && after) (void)hipStreamWaitEvent(ls, after, 0);
        (void)hipEventRecord(s.ev[EV_LSTART], ls);
        const bool fast = launch_ppht_fast(w.g, n, w.nz.p, w.nz_cap, w.npoints.p, w.rowplane.p, w.tab.p, w.hg.numangle, w.hg.numrho,
                                           p.threshold, pp->line_length, pp->line_gap, w.r_seg, w.seg_cap, w.r_segcount, w.r_status,
                                           w.fallback.p, w.rng_out.p, RNG_TABLE, vote_fused ? w.acc_full.p : nullptr, ls,
                                           (pp->ppht_two_per_cu && pp->ppht_full_for == 0) ? ppht_lds_half() : 0, false,
                                           2 * p.max_batch > pp->cu_count ? 0 : 8);
        if (pp->ppht_full_for > 0) pp->ppht_full_for--;
        // one launch for the frames the fast kernel flagged: their full-vote accumulator first (the general kernel
        // permutes the point list), then HoughLinesP with all state in global memory.  No events in between: every
        // packet on this stream is on the critical path.
        launch_ppht_general(w.g, n, w.nz.p, w.nz_cap, w.npoints.p, w.rowplane.p, w.tab.p, w.hg.numangle, w.hg.numrho, w.acc_ppht.p,
                            p.threshold, pp->line_length, pp->line_gap, w.r_seg, w.seg_cap, w.r_segcount, w.r_status,
                            fast ? w.fallback.p : nullptr, ls, vote_fused ? w.acc_full.p : nullptr);
        (void)hipEventRecord(s.ev[EV_LINES], ls);
    } else {
        (void)hipEventRecord(s.ev[EV_LSTART], ls);
        (void)hipEventRecord(s.ev[EV_LINES], ls);
    }
    s.end_ev = EV_LINES;
    if (!s.zero_copy) {   // one copy: counts, rounds, status and (if wanted) the segments of the n frames
        const size_t res_ints = w.res_head + (p.want_segments ? (size_t)n * w.seg_cap * 4 : 0);
        (void)hipMemcpyAsync(s.h_res, w.resblock.p, sizeof(int32_t) * res_ints, hipMemcpyDeviceToHost, ls);
        (void)hipEventRecord(s.ev[EV_END], ls);
        s.end_ev = EV_END;
    }
}

// from_host: the frames were just queued for copy on st_up (EV_START is recorded there)
int enqueue_all(lc_pipe *pp, Slot &s, bool from_host) {
    Work &w = s.w;
    const lc_pipe_params &p = pp->p;
    const int n = s.nframes;
    hipStream_t es = pp->st_edge;
    if (from_host) {
        HIP_TRY(hipEventRecord(s.ev[EV_UPLOADED], pp->st_up));
        HIP_TRY(hipStreamWaitEvent(es, s.ev[EV_UPLOADED], 0));
    } else {
        HIP_TRY(hipEventRecord(s.ev[EV_START], es));
        HIP_TRY(hipEventRecord(s.ev[EV_UPLOADED], es));
    }
    HIP_TRY(hipEventRecord(s.ev[EV_ESTART], es));
    // the exact reduction by three needs no pass of its own: the front kernel reads the source pixels it keeps
    const bool fused_resize = pp->fused_dec3();
    if (pp->resizes && !fused_resize) {
        launch_resize_linear(s.input, pp->src_w, pp->src_h, 3, n, w.bgr.p, w.g.W, w.g.H, pp->resize_tab.p, pp->resize_area2, es, pp->resize_dec3);
        s.input = w.bgr.p;
        s.input_bytes = (size_t)n * 3 * w.npix;
    }
    if (pp->level > 0 && pp->nlevels > 0) {
        if (pp->level_hold > 0) pp->level_hold--;
        else { pp->level--; pp->level_hold = 0; }     // try one level in again
    }
    s.level = pp->nlevels > 0 ? pp->level : 0;
    const bool skipping = s.level < pp->nlevels;
    const FrontSplit skip = pp->split(s.level, 1, nullptr), again = pp->split(pp->nlevels, 0, w.redo.p);
    if (pp->nlevels > 0) {
        if (s.level < s.planes_level) {      // back at a tighter level: the plane rows of the units it does not need go first
            // (for every frame the slot can hold, not just this batch's: an earlier, larger batch at the looser level
            // left its bits in the frames behind this batch's too, and a later batch at this level would flood from them)
            const FrontSplit clear_all = pp->split(s.level, 2, pp->ones.p);
            launch_front(s.input, s.input_bytes, 1, p.color_code, p.blur != 0, w.g, p.max_batch, pp->low, pp->high, w.planeS.p, w.planeW.p,
                         w.wake.p, nullptr, nullptr, es, nullptr, nullptr, nullptr, w.rowflags.p, pp->fused_dec3(), pp->front_rows,
                         &clear_all);
        }
        s.planes_level = s.level;
    }
    launch_front(s.input, s.input_bytes, 1, p.color_code, p.blur != 0, w.g, n, pp->low, pp->high, w.planeS.p, w.planeW.p,
                 w.wake.p, w.inter ? w.gray.p : nullptr, (w.inter && p.blur) ? w.blur.p : nullptr, es,
                 pp->has_gate ? &pp->gate : nullptr, pp->gate_tab.p, w.gray.p, w.rowflags.p, fused_resize, pp->front_rows,
                 pp->nlevels > 0 ? &skip : nullptr);
    HIP_TRY(hipEventRecord(s.ev[EV_FRONT], es));
    if (pp->deferred >= 0 && pp->defer_after == 1) { enqueue_lines(pp, pp->slots[pp->deferred], s.ev[EV_FRONT]); pp->deferred = -1; }
    hipStream_t ps = pp->st_post;
    if (ps != es) HIP_TRY(hipStreamWaitEvent(ps, s.ev[EV_FRONT], 0));
    launch_hysteresis(w.g, n, w.planeS.p, w.planeW.p, w.wake.p, w.r_rounds, ps, w.r_status);
    if (skipping) {
        // frames for which the units near the ROI alone do not settle its edges: all units again, and their hysteresis
        // (both launches leave at once for the other frames, which is nearly always all of them)
        const lc_pipe::SkipLevel &L = pp->levels[s.level];
        launch_ring_check(w.g, n, w.planeS.p, w.planeW.p, L.ring_words.p, L.n_ring_words, L.ring_code.p, w.redo.p, w.r_status, ps,
                          w.r_rounds);
        launch_front(s.input, s.input_bytes, 1, p.color_code, p.blur != 0, w.g, n, pp->low, pp->high, w.planeS.p, w.planeW.p,
                     w.wake.p, nullptr, nullptr, ps, pp->has_gate ? &pp->gate : nullptr, pp->gate_tab.p, w.gray.p, w.rowflags.p,
                     fused_resize, pp->front_rows, &again);
        launch_hysteresis(w.g, n, w.planeS.p, w.planeW.p, w.wake.p, w.r_rounds, ps, nullptr, w.redo.p);
    }
    HIP_TRY(hipEventRecord(s.ev[EV_HYST], ps));
    if (pp->deferred >= 0) { enqueue_lines(pp, pp->slots[pp->deferred], s.ev[EV_HYST]); pp->deferred = -1; }
    enqueue_back(pp, s);
    HIP_TRY(hipGetLastError());
    s.busy = true;
    s.timed = true;
    return LC_OK;
}

// The copy stream, created by the first host submit (a pipeline fed device frames never owns it).  Copies run on the
// DMA engines, but their packets still queue behind whatever shares the stream's hardware queue: picked like the rest.
int upload_stream(lc_pipe *pipe) {
    if (pipe->st_up) return LC_OK;
    for (Slot &s : pipe->slots)
        if (s.busy) { HIP_TRY(hipStreamCreateWithFlags(&pipe->st_up, hipStreamNonBlocking)); return LC_OK; }   // no experiments beside work in flight
    StreamPicker picker;
    std::vector<hipStream_t> mine = {pipe->st_edge};
    if (pipe->st_post != pipe->st_edge) mine.push_back(pipe->st_post);
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

}  // namespace

namespace lanecv {
void pipe_flush_pending(lc_pipe *pipe) {
    if (pipe->deferred >= 0) { enqueue_lines(pipe, pipe->slots[pipe->deferred], nullptr); pipe->deferred = -1; }
    for (size_t g = 0; g < pipe->groups.size(); g++) flush_group(pipe, (int)g);
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
    s->input_bytes = (size_t)nframes * pipe->src_frame_bytes;
    uint8_t *own = pipe->resizes ? s->w.src_in.p : s->w.bgr.p;
    s->input = own;
    if ((rc = upload_stream(pipe))) return rc;
    HIP_TRY(hipEventRecord(s->ev[EV_START], pipe->st_up));
    HIP_TRY(hipMemcpyAsync(own, host_bgr, s->input_bytes, hipMemcpyHostToDevice, pipe->st_up));
    return enqueue_all(pipe, *s, true);
}

int lc_pipe_submit_device(lc_pipe *pipe, int slot, const void *dev_bgr, int nframes) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!dev_bgr || nframes <= 0 || nframes > pipe->p.max_batch) return fail(LC_EINVAL, "submit: bad frames argument");
    if (((uintptr_t)dev_bgr & 15) != 0) return fail(LC_EINVAL, "submit: device frames must be 16-byte aligned");
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight; call lc_pipe_wait first", slot);
    s->nframes = nframes;
    s->input_bytes = (size_t)nframes * pipe->src_frame_bytes;
    s->input = (const uint8_t *)dev_bgr;
    return enqueue_all(pipe, *s, false);
}

int lc_pipe_wait(lc_pipe *pipe, int slot, int32_t *seg_xyxy, int32_t *seg_count) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!s->busy) return fail(LC_EINVAL, "slot %d has nothing in flight", slot);
    if (pipe->deferred == slot) { enqueue_lines(pipe, *s, nullptr); pipe->deferred = -1; }
    if (s->lines_pending) flush_group(pipe, s->group);
    // The wait polls the batch's last event and gives up after LANECV_WAIT_TIMEOUT_S seconds (default 120; 0 = block
    // in hipEventSynchronize for ever) with a report of which of the pipeline's events have fired, per slot, in the
    // order of the EV_ enumeration: a pipeline whose streams starve each other (more streams in the process than
    // hardware queues, see lc_pipe_create) then fails loudly instead of hanging its caller.
    static const double wait_limit = getenv("LANECV_WAIT_TIMEOUT_S") ? atof(getenv("LANECV_WAIT_TIMEOUT_S")) : 120.0;
    if (wait_limit > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; spins++) {
            const hipError_t q = hipEventQuery(s->ev[s->end_ev]);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) HIP_TRY(q);
            if ((spins & 63) != 63) continue;                     // the clock is read every 64th query
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > wait_limit) {
                std::string rep;
                for (size_t k = 0; k < pipe->slots.size(); k++) {
                    Slot &o = pipe->slots[k];
                    rep += " slot" + std::to_string(k) + (o.busy ? "*:" : ":");
                    for (auto &ev : o.ev) rep += hipEventQuery(ev) == hipSuccess ? '1' : '0';
                }
                (void)hipGetLastError();
                return fail(LC_EHIP, "wait: slot %d not finished after %.0f s; events fired (START UPLOADED FRONT HYST EDGE POINTS VOTE "
                            "LINES END ESTART LSTART VSTART):%s", slot, wait_limit, rep.c_str());
            }
            if (waited > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(50));   // a batch takes about a millisecond
        }
    } else {
        HIP_TRY(hipEventSynchronize(s->ev[s->end_ev]));
    }
    s->busy = false;
    const lc_pipe_params &p = pipe->p;
    if (p.want_segments && s->h_status[2]) pipe->ppht_full_for = 64;   // a frame outgrew half a CU's LDS (see lc_pipe)
    if (s->level < pipe->nlevels && s->h_status[3]) {                   // frames that needed the full pass after all
        int redone = 0;
        for (int i = 0; i < s->nframes; i++) redone += (s->h_rounds[i] >> 30) & 1;
        const char *ob = getenv("LANECV_SKIP_OFF_BATCHES");         // diagnostics / tests; 0 = never move a level out
        const int off_batches = ob ? atoi(ob) : 256;
        if (16 * redone > s->nframes && off_batches > 0 && pipe->level <= s->level) {
            pipe->level = s->level + 1;
            pipe->level_hold = off_batches;
        }
    }
    if (p.want_segments || p.want_accumulator || p.want_points) {
        if (s->h_status[0])
            return fail(LC_ECAP, "a frame holds more edge points than the pipeline keeps per frame (%d): its point list, accumulator and "
                        "segments are incomplete", s->w.nz_cap);
        if (s->h_status[1])
            return fail(LC_EINVAL, "HoughLinesP: a vote fell outside the accumulator (the angle table does not fit the image size)");
    }
    if (p.want_segments) {
        if (seg_count) memcpy(seg_count, s->h_segcount, sizeof(int) * s->nframes);
        if (seg_xyxy) memcpy(seg_xyxy, s->h_seg, sizeof(int32_t) * 4 * (size_t)s->nframes * s->w.seg_cap);
    }
    return LC_OK;
}

int lc_pipe_read(lc_pipe *pipe, int slot, int what, void *host_dst, size_t bytes) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight", slot);
    if (!host_dst) return fail(LC_EINVAL, "null destination");
    Work &w = s->w;
    const void *src = nullptr;
    size_t need = 0;
    const size_t img = (size_t)s->nframes * w.npix;
    switch (what) {
        case LC_BUF_GRAY: src = w.inter ? w.gray.p : nullptr; need = img; break;
        case LC_BUF_BLUR: src = (w.inter && pipe->p.blur) ? w.blur.p : nullptr; need = img; break;
        case LC_BUF_EDGES: src = w.inter ? w.edges.p : nullptr; need = img; break;
        case LC_BUF_MASKED: src = w.masked.p; need = img; break;
        case LC_BUF_ACC: src = pipe->p.want_accumulator ? w.acc_full.p : nullptr;
            need = sizeof(int32_t) * (size_t)s->nframes * w.hg.numangle * w.hg.numrho; break;
        case LC_BUF_NPOINTS: src = (pipe->p.want_accumulator || pipe->p.want_segments || pipe->p.want_points) ? w.npoints.p : nullptr;
            need = sizeof(int) * s->nframes; break;
        case LC_BUF_RESIZED:
            if (pipe->fused_dec3())
                return fail(LC_EINVAL, "the resized frames are not materialised when the reduction by three is fused into the edge "
                            "kernel; configure keep_intermediates to read them");
            src = pipe->resizes ? w.bgr.p : nullptr; need = 3 * img; break;
        default: return fail(LC_EINVAL, "unknown buffer id %d", what);
    }
    if (!src) return fail(LC_EINVAL, "buffer %d is not produced by this pipeline's configuration", what);
    if (bytes < need) return fail(LC_ECAP, "destination holds %zu bytes, %zu needed", bytes, need);
    HIP_TRY(hipMemcpy(host_dst, src, need, hipMemcpyDeviceToHost));
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
    Work &w = s->w;
    int np = 0;
    HIP_TRY(hipMemcpy(&np, w.npoints.p + frame, sizeof(int), hipMemcpyDeviceToHost));
    *n = np;
    if (np > w.nz_cap) return fail(LC_ECAP, "frame %d has %d edge points, the slot keeps %d", frame, np, w.nz_cap);
    if (np > capacity) return fail(LC_ECAP, "frame %d has %d edge points, destination holds %d", frame, np, capacity);
    if (np > 0) {
        if (!xy) return fail(LC_EINVAL, "null destination");
        HIP_TRY(hipMemcpy(xy, w.nz.p + (size_t)frame * w.nz_cap, sizeof(uint32_t) * np, hipMemcpyDeviceToHost));
    }
    return LC_OK;
}

int lc_pipe_stage_ms(lc_pipe *pipe, int slot, int stage, float *ms) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!ms) return fail(LC_EINVAL, "null output");
    if (s->busy || !s->timed) return fail(LC_EINVAL, "slot %d has no complete timing", slot);
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
        case LC_STAGE_TOTAL: a = EV_START; b = s->end_ev; break;
        case LC_STAGE_FRONT: a = EV_ESTART; b = EV_FRONT; break;
        case LC_STAGE_HYST: a = EV_FRONT; b = EV_HYST; break;
        case LC_STAGE_FINAL: a = EV_HYST; b = EV_EDGE; break;
        default: return fail(LC_EINVAL, "unknown stage %d", stage);
    }
    HIP_TRY(hipEventElapsedTime(ms, s->ev[a], s->ev[b]));
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
        else if (pipe->nlevels > 0 && s->nframes > 0) {
            std::vector<int> redo(s->nframes);
            HIP_TRY(hipMemcpy(redo.data(), s->w.redo.p, sizeof(int) * s->nframes, hipMemcpyDeviceToHost));
            for (int v : redo) *frames_redone += v != 0;
        }
    }
    return LC_OK;
}

int lc_pipe_shared_queues(lc_pipe *pipe, int *count) {
    if (!pipe || !count) return fail(LC_EINVAL, "null argument");
    *count = pipe->shared_queues;
    return LC_OK;
}

int lc_pipe_layout(lc_pipe *pipe, int *group_size, int *lines_streams, int *roi_levels) {
    if (!pipe) return fail(LC_EINVAL, "null pipeline");
    if (group_size) *group_size = pipe->group_size;
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
    HIP_TRY(hipMemcpyAsync(pipe->resizes ? s->w.src_in.p : s->w.bgr.p, host_bgr, (size_t)nframes * pipe->src_frame_bytes,
                           hipMemcpyHostToDevice, pipe->st_up));
    HIP_TRY(hipStreamSynchronize(pipe->st_up));
    return LC_OK;
}

int lc_pipe_input_device_ptr(lc_pipe *pipe, int slot, void **dev_ptr) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!dev_ptr) return fail(LC_EINVAL, "null output");
    *dev_ptr = pipe->resizes ? s->w.src_in.p : s->w.bgr.p;
    return LC_OK;
}

}  // extern "C"
