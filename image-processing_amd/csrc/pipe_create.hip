// The batched pipeline's set-up: the ROI plane and the levels of the ROI scheme, the streams (each with a hardware
// queue of its own, found by experiment), the banks of buffers and the slots' places in them; and its tear-down.
#include "pipe.h"

using namespace lanecv;

namespace {

// How many slots share a bank of buffers (see Bank): batches that leave most CUs free are queued `parts` at a time.
// For what the kernels' frame sets cover: the streaming front kernel on frames as submitted (or its fused reduction by
// three), no intermediates.  LANECV_NO_GROUPS=1 / LANECV_GROUP_SIZE=n: diagnostics.
void setup_layout(lc_pipe *pp) {
    const lc_pipe_params &p = pp->p;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, pp->ctx->device) == hipSuccess && prop.multiProcessorCount > 0) pp->cu_count = prop.multiProcessorCount;
    else (void)hipGetLastError();
    const bool can = !getenv("LANECV_NO_GROUPS") && !p.keep_intermediates && front_stream_supported(pp->g) && !getenv("LANECV_FRONT_TILED") &&
                     (!pp->resizes || pp->fused_dec3()) && !(getenv("LANECV_RESULT_COPY") && getenv("LANECV_RESULT_COPY")[0] == '1');
    int parts = 1;
    if (can && 2 * p.max_batch <= pp->cu_count && p.nslots >= 4) parts = std::min(p.nslots / 2, pp->cu_count / p.max_batch);
    if (const char *gs = getenv("LANECV_GROUP_SIZE")) {
        const int v = atoi(gs);
        if (can && v >= 1 && v <= p.nslots) parts = v;
    }
    pp->parts = std::max(1, std::min(parts, (int)MAX_PARTS));
    pp->bank_frames = pp->parts * p.max_batch;
}

// cv2.resize in front (tables as OpenCV builds them) and the colour gate's tables
int setup_front_tables(lc_pipe *pp) {
    const lc_pipe_params &p = pp->p;
    int rc;
    pp->src_w = p.src_width > 0 ? p.src_width : p.width;
    pp->src_h = p.src_height > 0 ? p.src_height : p.height;
    if (pp->src_w > MAX_DIM || pp->src_h > MAX_DIM) return fail(LC_EUNSUPPORTED, "source frames above %d pixels a side are not supported", MAX_DIM);
    pp->resizes = pp->src_w != p.width || pp->src_h != p.height;
    pp->src_frame_bytes = (size_t)3 * pp->src_w * pp->src_h;
    if (pp->resizes) {
        const size_t tn = 2 * ((size_t)p.width + p.height);
        std::vector<int> tab(tn);
        pp->resize_area2 = resize_tables_host(pp->src_w, pp->src_h, p.width, p.height, tab.data(), tab.data() + p.width,
                                              tab.data() + 2 * p.width, tab.data() + 2 * p.width + p.height);
        pp->resize_dec3 = !pp->resize_area2 && resize_is_decimate3(pp->src_w, pp->src_h, p.width, p.height, tab.data(), tab.data() + p.width,
                                                                   tab.data() + 2 * p.width, tab.data() + 2 * p.width + p.height);
        if ((rc = pp->resize_tab.alloc(tn))) return rc;
        hipError_t e = hipMemcpy(pp->resize_tab.p, tab.data(), sizeof(int) * tn, hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(LC_EHIP, "resize table upload failed: %s", hipGetErrorString(e));
    }
    if (p.hsv_gate) {
        pp->has_gate = true;
        pp->gate = make_gate(p.hsv_bounds, 2);
        std::vector<int> tab(768);
        gate_tables_host(pp->gate, tab.data());
        if ((rc = pp->gate_tab.alloc(tab.size()))) return rc;
        hipError_t e = hipMemcpy(pp->gate_tab.p, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(LC_EHIP, "gate table upload failed: %s", hipGetErrorString(e));
    }
    return LC_OK;
}

// ROI and car mask fold into one bit plane (also returned in `bits`): keep = (roi == 255) && (car != 0)
int setup_roi(lc_pipe *pp, const lc_pipe_params &p, std::vector<uint64_t> &bits) {
    int rc;
    if (p.roi_mask || p.car_mask) {
        const Geo g = pp->g;
        bits.assign(g.plane_words(), 0);
        int roi_first = -1;
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                size_t i = (size_t)y * g.W + x;
                bool keep = true;
                if (p.roi_mask) {
                    uint8_t v = p.roi_mask[i];
                    if (v != 0 && v != 255) {
                        return fail(LC_EUNSUPPORTED, "roi_mask must hold only 0 and 255 (found %d at %d,%d)", v, x, y);
                    }
                    keep = v != 0;
                }
                if (keep && p.car_mask) keep = p.car_mask[i] != 0;
                if (keep) {
                    bits[(size_t)y * g.TX + (x >> 6)] |= 1ull << (x & 63);
                    if (roi_first < 0) roi_first = y;
                }
            }
        pp->roi_first_row = roi_first < 0 ? g.H : roi_first;
        if ((rc = pp->roi_bits.alloc(bits.size()))) return rc;
        hipError_t e = hipMemcpy(pp->roi_bits.p, bits.data(), bits.size() * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(LC_EHIP, "roi upload failed: %s", hipGetErrorString(e));
        std::vector<uint32_t> words;
        for (int y = 0; y < g.H; y++)
            for (int tx = 0; tx < g.TX; tx++)
                if (bits[(size_t)y * g.TX + tx]) words.push_back((uint32_t)y << 8 | (uint32_t)tx);
        if (!words.empty() && g.TX <= 256) {
            if ((rc = pp->roi_words.alloc(words.size()))) return rc;
            e = hipMemcpy(pp->roi_words.p, words.data(), words.size() * 4, hipMemcpyHostToDevice);
            if (e != hipSuccess) return fail(LC_EHIP, "roi upload failed: %s", hipGetErrorString(e));
            pp->n_roi_words = (int)words.size();
        }
        pp->has_roi = true;
    }
    return LC_OK;
}

// The levels of the ROI scheme (see lc_pipe): which units of the front kernel's grid each level classifies, and the
// rings around the region the per-frame proof looks at.  Also fixes the front kernel's unit height.
int setup_skip_levels(lc_pipe *pp, const std::vector<uint64_t> &bits) {
    const lc_pipe_params &p = pp->p;
    int rc;
    {
        const char *ns = getenv("LANECV_NO_ROW_SKIP");
        const Geo g = pp->g;
        // tuning: rows per unit of the streaming front kernel (read here, once: the unit lists below are built for it)
        int rows_forced = 0;
        if (const char *re = getenv("LANECV_FRONT_ROWS")) {
            rows_forced = atoi(re);
            if (rows_forced < 8 || rows_forced > MAX_DIM) return fail(LC_EINVAL, "LANECV_FRONT_ROWS=%s: expected 8 .. %d rows", re, MAX_DIM);
        }
        auto unit_rows = [&](int rows_classified) { return rows_forced ? rows_forced : front_stream_rows_per_unit(g, pp->bank_frames, rows_classified); };
        // (the per-frame proof packs pixel coordinates into 13 bits each)
        if (pp->has_roi && !p.keep_intermediates && front_stream_supported(g) && !getenv("LANECV_FRONT_TILED") && !ns && g.W <= 8191 &&
            g.H <= 8191) {
            // rings 1..15 around the ROI by fifteen 3x3 dilations of its bit plane
            constexpr int NRING = 15;
            const int TX = g.TX, H = g.H;
            const uint64_t last_mask = (g.W & 63) ? ((1ull << (g.W & 63)) - 1) : ~0ull;
            auto dilate = [&](const std::vector<uint64_t> &in) {
                std::vector<uint64_t> hz(in.size()), out(in.size());
                for (int y = 0; y < H; y++)
                    for (int tx = 0; tx < TX; tx++) {
                        const uint64_t c = in[(size_t)y * TX + tx];
                        const uint64_t l = tx > 0 ? in[(size_t)y * TX + tx - 1] >> 63 : 0, r = tx + 1 < TX ? in[(size_t)y * TX + tx + 1] << 63 : 0;
                        uint64_t v = c | (c << 1) | (c >> 1) | l | r;
                        if (tx == TX - 1) v &= last_mask;
                        hz[(size_t)y * TX + tx] = v;
                    }
                for (int y = 0; y < H; y++)
                    for (int tx = 0; tx < TX; tx++) {
                        uint64_t v = hz[(size_t)y * TX + tx];
                        if (y > 0) v |= hz[(size_t)(y - 1) * TX + tx];
                        if (y + 1 < H) v |= hz[(size_t)(y + 1) * TX + tx];
                        out[(size_t)y * TX + tx] = v;
                    }
                return out;
            };
            std::vector<uint64_t> cur = bits, code(4 * bits.size(), 0);
            for (int k = 1; k <= NRING; k++) {
                std::vector<uint64_t> nxt = dilate(cur);
                for (size_t i = 0; i < bits.size(); i++) {
                    const uint64_t ring = nxt[i] & ~cur[i];
                    for (int b = 0; b < 4; b++)
                        if (k & (1 << b)) code[b * bits.size() + i] |= ring;
                }
                cur.swap(nxt);
            }
            int first = -1;
            for (int y = 0; y < H && first < 0; y++)
                for (int tx = 0; tx < TX; tx++)
                    if (cur[(size_t)y * TX + tx]) { first = y; break; }
            if (first >= 0) {
                pp->row_split = first >= 64 ? first : 0;
                pp->front_rows = unit_rows(H - pp->row_split);
                const FrontGrid fg = front_stream_grid(g, pp->front_rows, pp->row_split);
                // Where each lower segment's strips start: at the first column (rounded down to a plane byte) that comes within
                // the rings of the ROI in the segment's rows, so that the segment takes the fewest strips that cover its share
                // of the region -- with a fixed grid a triangle's slanted sides cost about one strip more in most segments.
                pp->shift = lanecv::FrontShift{};
                pp->shift.nstrips = fg.nstrips;
                pp->shifted = false;
                if (fg.nseg_top + fg.nseg_low <= lanecv::FRONT_SHIFT_SEGS && !getenv("LANECV_NO_STRIP_SHIFT")) {
                    for (int sg = fg.nseg_top; sg < fg.nseg_top + fg.nseg_low; sg++) {
                        const int y0 = pp->row_split + (sg - fg.nseg_top) * fg.rows_low, y1 = std::min(y0 + fg.rows_low, H);
                        int first_col = -1;
                        for (int tx = 0; tx < TX && first_col < 0; tx++) {
                            uint64_t any = 0;
                            for (int y = y0; y < y1; y++) any |= cur[(size_t)y * TX + tx];
                            if (any) first_col = tx * 64 + __builtin_ctzll(any);
                        }
                        if (first_col < 0) continue;
                        const int a8 = first_col & ~7;
                        const int d = (fg.strip_cols - a8 % fg.strip_cols) % fg.strip_cols;      // 248 * k - d == a8 for some k
                        pp->shift.by8[sg] = (uint8_t)(d / 8);
                        if (d) pp->shifted = true;
                        pp->shift.nstrips = std::max(pp->shift.nstrips, (lanecv::TILE * TX + d + fg.strip_cols - 1) / fg.strip_cols);
                    }
                }
                const int nst = pp->shift.nstrips;
                // uploads one level: the units a region `reach` (bit plane) touches, and the ring code planes around it
                auto add_level = [&](const std::vector<uint64_t> &reach, const std::vector<uint64_t> &rcode) -> int {
                    std::vector<uint16_t> needed, unneeded;
                    long long npix = 0;
                    for (int sg = 0; sg < fg.nseg_top + fg.nseg_low; sg++)
                        for (int st = 0; st < nst; st++) {
                            bool need = false;
                            const int xs = st * fg.strip_cols - 8 * (sg < lanecv::FRONT_SHIFT_SEGS ? pp->shift.by8[sg] : 0);
                            if (xs >= lanecv::TILE * TX) continue;       // (a shifted segment's spare strip: no unit)
                            if (sg >= fg.nseg_top) {
                                const int y0 = pp->row_split + (sg - fg.nseg_top) * fg.rows_low, y1 = std::min(y0 + fg.rows_low, H);
                                const int x0 = std::max(xs, 0), x1 = std::min(xs + fg.strip_cols, g.W);
                                for (int y = y0; y < y1 && !need; y++)
                                    for (int tx = x0 >> 6; tx <= (x1 - 1) >> 6 && tx < TX && !need; tx++) {
                                        uint64_t v = reach[(size_t)y * TX + tx];
                                        const int lo = std::max(x0 - tx * 64, 0), hi = std::min(x1 - tx * 64, 64);   // bits [lo, hi) of the word
                                        if (lo > 0) v &= ~((1ull << lo) - 1);
                                        if (hi < 64) v &= (1ull << hi) - 1;
                                        need = v != 0;
                                    }
                                if (need) npix += (long long)(y1 - y0) * (x1 - x0);
                            }
                            (need ? needed : unneeded).push_back((uint16_t)(sg | (st << 8)));
                        }
                    if (needed.empty() || unneeded.empty()) return LC_OK;          // nothing to skip at this level
                    for (int lv = 0; lv < pp->nlevels; lv++)
                        if (pp->levels[lv].n_needed == (int)needed.size()) return LC_OK;   // no looser than the level before it
                    std::vector<uint32_t> rw;
                    const size_t nw = bits.size();
                    for (size_t i = 0; i < nw; i++)
                        if (rcode[i] | rcode[nw + i] | rcode[2 * nw + i] | rcode[3 * nw + i]) rw.push_back((uint32_t)i);
                    lc_pipe::SkipLevel &L = pp->levels[pp->nlevels];
                    int rc2;
                    if ((rc2 = L.needed.alloc(needed.size())) || (rc2 = L.unneeded.alloc(unneeded.size())) ||
                        (rc2 = L.ring_words.alloc(rw.size())) || (rc2 = L.ring_code.alloc(rcode.size()))) return rc2;
                    hipError_t e = hipMemcpy(L.needed.p, needed.data(), needed.size() * 2, hipMemcpyHostToDevice);
                    if (e == hipSuccess) e = hipMemcpy(L.unneeded.p, unneeded.data(), unneeded.size() * 2, hipMemcpyHostToDevice);
                    if (e == hipSuccess && !rw.empty()) e = hipMemcpy(L.ring_words.p, rw.data(), rw.size() * 4, hipMemcpyHostToDevice);
                    if (e == hipSuccess) e = hipMemcpy(L.ring_code.p, rcode.data(), rcode.size() * 8, hipMemcpyHostToDevice);
                    if (e != hipSuccess) return fail(LC_EHIP, "unit list upload failed: %s", hipGetErrorString(e));
                    L.n_needed = (int)needed.size(); L.n_unneeded = (int)unneeded.size(); L.n_ring_words = (int)rw.size();
                    L.pixels = npix;
                    if (pp->nlevels == 0 && !pp->resizes) {
                        // the windows of a frame this level's units read: per lower segment the columns from its first needed
                        // strip's first lane to its last one's last (a strip is 256 columns wide, 4 before its first output
                        // column), the rows from 4 above the segment to 4 below (blur, Sobel and NMS reach that far)
                        pp->up_windows.clear();
                        pp->up_window_bytes = 0;
                        for (int sg = fg.nseg_top; sg < fg.nseg_top + fg.nseg_low; sg++) {
                            int smin = INT_MAX, smax = -1;
                            for (uint16_t u : needed)
                                if ((u & 255) == sg) { smin = std::min(smin, (int)(u >> 8)); smax = std::max(smax, (int)(u >> 8)); }
                            if (smax < 0) continue;
                            const int y0 = pp->row_split + (sg - fg.nseg_top) * fg.rows_low, y1 = std::min(y0 + fg.rows_low, H);
                            const int shift_cols = 8 * (sg < lanecv::FRONT_SHIFT_SEGS ? pp->shift.by8[sg] : 0);
                            const int c0 = std::max(0, smin * fg.strip_cols - shift_cols - 4), c1 = std::min(g.W, smax * fg.strip_cols - shift_cols + 252);
                            lc_pipe::UpWindow uw{std::max(0, y0 - 4), std::min(H, y1 + 4), 3 * c0, 3 * c1};
                            pp->up_windows.push_back(uw);
                            pp->up_window_bytes += (long long)(uw.r1 - uw.r0) * (uw.b1 - uw.b0);
                        }
                    }
                    pp->nlevels++;
                    return LC_OK;
                };
                if (fg.nseg_top + fg.nseg_low < 256 && nst < 256) {
                    if ((rc = add_level(cur, code))) return rc;
                    if (pp->row_split > 0) {
                        // the band of rows the ROI spans as the region: ring k = row (first ROI row) - k, all columns
                        std::vector<uint64_t> band(bits.size(), 0), bcode(4 * bits.size(), 0);
                        for (int y = pp->row_split; y < H; y++)
                            for (int tx = 0; tx < TX; tx++) band[(size_t)y * TX + tx] = tx == TX - 1 ? last_mask : ~0ull;
                        for (int k = 1; k <= NRING; k++) {
                            const int y = pp->roi_first_row - k;
                            if (y < 0) break;
                            for (int tx = 0; tx < TX; tx++)
                                for (int b = 0; b < 4; b++)
                                    if (k & (1 << b)) bcode[b * bits.size() + (size_t)y * TX + tx] = tx == TX - 1 ? last_mask : ~0ull;
                        }
                        if ((rc = add_level(band, bcode))) return rc;
                    }
                }
                if (pp->nlevels > 0) {
                    std::vector<int> one(pp->bank_frames, 1);
                    if ((rc = pp->ones.alloc(one.size()))) return rc;
                    if (hipMemcpy(pp->ones.p, one.data(), one.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                        return fail(LC_EHIP, "flag upload failed");
                    }
                } else {
                    pp->row_split = 0;
                    pp->shift = lanecv::FrontShift{};
                    pp->shifted = false;
                }
            }
        }
        if (pp->nlevels == 0) pp->front_rows = unit_rows(g.H);
        pp->frame_pixels = (long long)g.W * g.H;
    }
    return LC_OK;
}

int setup_streams(lc_pipe *pp) {
    const lc_pipe_params &p = pp->p;
    {
        // All streams have the same priority.  With the post and HoughLinesP streams at high priority their few
        // workgroups were placed ahead of a front kernel's thousands a little sooner (327 k against 320 k frames/s at
        // 1080p x 256), but both streams spend most of their time with a wait for another stream at the head of their
        // queue, and whenever the process owned more hardware queues than this pipeline's (RCCL initialised, a second
        // pipeline process on the GPU) the front kernel's queue was never served again: a hang at the first batch.
        // st_up is created by the first host submit: a pipeline that is only ever handed device frames keeps its
        // stream count (and with it the number of hardware queues it wants) at three
        //
        // HIP folds a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and kernels of
        // streams that share one run one after the other -- which undoes the overlap this layout exists for (198 k
        // instead of 327 k frames/s at 1080p x 256 when the post stream shared a queue).  Which streams share is
        // HIP's business and depends on what else the process created, so every stream below is picked by
        // experiment (StreamPicker): candidates are created until one runs side by side with the streams picked so
        // far.  In the order of what the overlap is worth: HoughLinesP beside the front kernels, then the short
        // kernels, then further HoughLinesP streams for small batches; a stream for which no queue of its own is
        // left falls back to the stream it would otherwise have run beside (counted in `shared_queues`).
        StreamPicker picker;
        std::vector<hipStream_t> mine;
        bool own = true;
        hipError_t e = hipStreamCreateWithFlags(&pp->st_edge, hipStreamNonBlocking);
        if (e == hipSuccess) mine.push_back(pp->st_edge);
        const char *pr = getenv("LANECV_LINES_PRIORITY");   // diagnostics: 0 keeps HoughLinesP on the edge stream
        if (e == hipSuccess && p.want_segments && !(pr && pr[0] == '0')) {
            pp->st_lines = picker.pick(mine, &own);
            if (!pp->st_lines) e = hipErrorOutOfMemory;
            else { mine.push_back(pp->st_lines); pp->shared_queues += own ? 0 : 1; }
        }
        if (e != hipSuccess) return fail(LC_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        const bool own_lines = pp->st_lines != nullptr;
        if (!pp->st_lines) pp->st_lines = pp->st_edge;
        // the short kernels get a stream of their own when batches are pipelined: the next batch's
        // front kernel then starts at once and they run beside it (LANECV_POST_STREAM=0: keep them on st_edge)
        pp->st_post = pp->st_edge;
        const char *po = getenv("LANECV_POST_STREAM");
        // Pipelines with three banks or more run the short kernels of a launch on that launch's own HoughLinesP stream
        // (see lc_pipe::tail_streams) and hysteresis on the edge stream.  LANECV_TAIL_STREAMS=0: the post stream
        // layout for them too; =1: hysteresis on the tail streams as well (it then waits for room beside the next
        // front kernel: measured slower).  (Set to 1 or 2 it also applies to pipelines of two banks, where the layout
        // makes no difference: 505 k frames/s either way at 720p x 32 with 16 slots.)
        const char *ts = getenv("LANECV_TAIL_STREAMS");
        // (Not for launches whose front kernel is no longer than the hysteresis kernel behind it -- 960 x 540 frames in
        // batches of 256: 1.12 M frames/s with the post stream, 1.01 M with this; 1280 x 720: 0.78 M against 0.85 M.)
        const long long classified = (pp->nlevels > 0 ? pp->levels[0].pixels : pp->frame_pixels) * (long long)pp->bank_frames;
        const bool long_front = pp->parts > 1 || classified >= 64ll << 20;
        pp->tail_streams = own_lines && pp->banks.size() >= (ts ? 2u : 3u) && (ts ? ts[0] != '0' : long_front);
        pp->hyst_on_edge = pp->tail_streams && !(ts && ts[0] == '1');
        // LANECV_HYST_LEAD=1: hysteresis not as a kernel of its own on the edge stream but as the first workgroups of the
        // next front launch (lc_pipe::hyst_fused).  Off by default: the edge stream then holds front kernels only (0.33 ms
        // a batch at 1080p x 256 instead of 0.29 + 0.11), but beside front kernels that never pause the HoughLinesP
        // kernels take 1.2 ms instead of 1.06, the tail streams become the limit, and a batch's tail starts a submit
        // later: 0.49 ms a batch against 0.445.
        const char *hl = getenv("LANECV_HYST_LEAD");
        pp->hyst_fused = pp->hyst_on_edge && front_takes_lead(pp->g, 1) && !p.keep_intermediates && hl && hl[0] == '1';
        if (pp->tail_streams) pp->st_post = pp->st_lines;
        else if (pp->banks.size() > 1 && !(po && po[0] == '0')) {
            hipStream_t st = picker.pick(mine, &own);
            if (st && own) { pp->st_post = st; mine.push_back(st); }
            else {                                  // no queue of its own: behind the front kernels, as with one slot
                if (st) picker.rejected.push_back(st);
                pp->shared_queues++;
            }
        }
        if (own_lines) {
            int want = 1;
            if (pp->cu_count > 0)
                want = std::max(1, std::min(std::min((pp->ppht_two_per_cu ? 2 : 1) * pp->cu_count / pp->bank_frames + (pp->tail_streams ? 1 : 0),
                                                     (int)pp->banks.size()), 3));
            const char *ns = getenv("LANECV_LINES_STREAMS");
            if (ns && ns[0] >= '1' && ns[0] <= '6') want = ns[0] - '0';   // (only streams that find a hardware queue of their own are taken)
            for (int k = 1; k < want; k++) {
                hipStream_t st = picker.pick(mine, &own);
                if (st && own) { pp->st_lines_more.push_back(st); mine.push_back(st); }
                else { if (st) picker.rejected.push_back(st); pp->shared_queues++; break; }
            }
        }
    }
    if (getenv("LANECV_QUEUE_DEBUG")) {      // diagnostics: which of the pipeline's streams run side by side, after the rejected candidates are gone
        std::vector<hipStream_t> all{pp->st_edge, pp->st_post, pp->st_lines};
        for (hipStream_t st : pp->st_lines_more) all.push_back(st);
        StreamPicker check;
        for (size_t i = 0; i < all.size(); i++) {
            for (size_t j = 0; j < all.size(); j++)
                fprintf(stderr, "%c", i == j || all[i] == all[j] ? '.' : (check.side_by_side(all[i], all[j]) ? '1' : '0'));
            fprintf(stderr, "  (row spins, column runs beside it: edge, post, lines, more lines)\n");
        }
    }
    return LC_OK;
}

int setup_banks(lc_pipe *pp) {
    const lc_pipe_params &p = pp->p;
    const bool hough = p.want_accumulator || p.want_segments;
    const int B = p.max_batch, G = pp->parts, segcap = std::max(p.seg_capacity, 1);
    int rc;
    pp->banks.resize((p.nslots + G - 1) / G);
    pp->slots.resize(p.nslots);
    for (size_t bi = 0; bi < pp->banks.size(); bi++) {
        Bank &bk = pp->banks[bi];
        bk.parts = std::min(G, p.nslots - (int)bi * G);
        const int frames = bk.parts * B;
        hipError_t e;
        for (auto &ev : bk.ev) {
            e = hipEventCreate(&ev);
            if (e != hipSuccess) return fail(LC_EHIP, "hipEventCreate failed: %s", hipGetErrorString(e));
        }
        Work &w = bk.w;
        if ((rc = w.init_images(p.width, p.height, frames, true, p.keep_intermediates != 0, true))) return rc;
        if (pp->resizes && (rc = w.src_in.alloc((size_t)frames * pp->src_frame_bytes))) return rc;
        // frames the streaming front kernel cannot take pass through a gated gray image
        if (pp->has_gate && !w.gray.p && !front_stream_supported(pp->g) && (rc = w.gray.alloc((size_t)frames * p.width * p.height))) return rc;
        if (hough || p.want_points) {
            // point list capacity per frame: an eighth of the pixels (a Canny map inside an ROI stays far below)
            int cap = std::max(1024, (int)std::min<size_t>((size_t)p.width * p.height, (size_t)p.width * p.height / 8 + 1024));
            if ((rc = w.init_points(cap))) return rc;
            HIP_TRY(hipMemset(w.npoints.p, 0, sizeof(int) * (size_t)frames));
        }
        if (hough && (rc = w.init_hough(p.rho, p.theta, segcap, p.want_accumulator != 0, p.want_segments != 0, pp->st_edge))) return rc;
        w.seg_cap = segcap;
        if ((rc = w.init_results(segcap, bk.parts))) return rc;
        void *hp = nullptr;
        const size_t res_ints = w.res_head + (size_t)frames * segcap * 4;
        // mapped and coherent: the kernels store a batch's results into it directly (see below), and the host reads
        // them after the batch's last event
        if (hipHostMalloc(&hp, sizeof(int32_t) * res_ints, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            hp = nullptr;
            (void)hipGetLastError();
            if ((rc = lc_host_alloc_pinned(sizeof(int32_t) * res_ints, &hp))) return rc;
        }
        bk.h_res = (int32_t *)hp;
        memset(bk.h_res, 0, sizeof(int32_t) * w.res_head);
        // What goes back per batch is a few hundred bytes per frame, written once: the kernels store it into the
        // pinned block directly, which takes a copy and its launch off the HoughLinesP stream
        // (LANECV_RESULT_COPY=1 keeps the device block and the copy; banks then hold one slot each).
        const char *rc_env = getenv("LANECV_RESULT_COPY");
        void *dp = nullptr;
        if (!(rc_env && rc_env[0] == '1') && hipHostGetDevicePointer(&dp, bk.h_res, 0) == hipSuccess && dp) {
            int32_t *d = (int32_t *)dp;
            w.r_segcount = d; w.r_rounds = d + frames; w.r_status = d + 2 * (size_t)frames; w.r_seg = d + w.res_head;
            bk.zero_copy = true;
        } else if (bk.parts > 1) {
            return fail(LC_EHIP, "the pinned result block of a bank of %d slots cannot be mapped into the GPU's address space "
                        "(LANECV_NO_GROUPS=1 configures one slot per bank)", bk.parts);
        }
        for (int j = 0; j < bk.parts; j++) {
            Slot &s = pp->slots[bi * G + j];
            s.bank = (int)bi; s.pos = j;
            const size_t fo = (size_t)j * B;       // the slot's first frame in the bank's buffers
            s.h_segcount = bk.h_res + fo; s.h_rounds = bk.h_res + frames + fo; s.h_status = bk.h_res + 2 * (size_t)frames + 8 * (size_t)j;
            s.h_seg = bk.h_res + w.res_head + fo * segcap * 4;
            e = hipEventCreate(&s.done);
            if (e != hipSuccess) return fail(LC_EHIP, "hipEventCreate failed: %s", hipGetErrorString(e));
        }
    }
    return LC_OK;
}

}  // namespace

extern "C" {

int lc_pipe_create(lc_ctx *ctx, const lc_pipe_params *params, lc_pipe **out) {
    int rc;
    if (!out || !params) return fail(LC_EINVAL, "lc_pipe_create: null argument");
    *out = nullptr;
    if ((rc = ctx_bind(ctx))) return rc;
    const lc_pipe_params &p = *params;
    if (p.width <= 0 || p.height <= 0 || p.max_batch <= 0 || p.nslots <= 0) return fail(LC_EINVAL, "lc_pipe_create: bad sizes");
    if (p.width > MAX_DIM || p.height > MAX_DIM) return fail(LC_EUNSUPPORTED, "frames above %d pixels a side are not supported", MAX_DIM);
    if (p.color_code != LC_COLOR_BGR2GRAY && p.color_code != LC_COLOR_RGB2GRAY)
        return fail(LC_EUNSUPPORTED, "color_code must be COLOR_BGR2GRAY (6) or COLOR_RGB2GRAY (7)");
    if (!(p.canny_lo == p.canny_lo) || !(p.canny_hi == p.canny_hi)) return fail(LC_EINVAL, "NaN Canny threshold");
    if (p.want_segments && p.seg_capacity <= 0) return fail(LC_EINVAL, "seg_capacity must be positive");

    lc_pipe *pp = new lc_pipe();
    pp->ctx = ctx;
    pp->p = p;
    pp->p.roi_mask = pp->p.car_mask = nullptr;   // host pointers are not kept
    pp->g = make_geo(p.width, p.height);
    double t1 = std::min(p.canny_lo, p.canny_hi), t2 = std::max(p.canny_lo, p.canny_hi);
    t1 = std::max(-1e9, std::min(1e9, t1));
    t2 = std::max(-1e9, std::min(1e9, t2));
    pp->low = (int)std::floor(t1);
    pp->high = (int)std::floor(t2);
    pp->line_length = (int)lrint(p.min_line_length);
    pp->line_gap = (int)lrint(p.max_line_gap);
    if (const char *fv = getenv("LANECV_VOTE_FUSED")) pp->fuse_vote = fv[0] != '0';
    pp->ppht_two_per_cu = p.want_segments && (size_t)p.width * p.height <= 2600000;    // (and two banks or more, below)
    if (const char *tp = getenv("LANECV_PPHT_TWO_PER_CU")) pp->ppht_two_per_cu = tp[0] != '0';   // diagnostics

    std::vector<uint64_t> bits;
    if ((rc = setup_front_tables(pp))) { lc_pipe_destroy(pp); return rc; }
    setup_layout(pp);
    pp->banks.resize((p.nslots + pp->parts - 1) / pp->parts);     // (their number shapes the stream layout; filled in by setup_banks)
    // (With two banks as with more: launches of 256 frames that take a whole CU each cannot overlap at all -- 720p x 32 with
    // 16 slots 420 k -> 505 k frames/s, 1080p x 256 with two slots 315 k -> 404 k, 960 x 540 x 256 with two 555 k -> 779 k.
    // A single bank has nothing to overlap with, and a frame alone on a CU is served by eight wavefronts instead of four.)
    if (!getenv("LANECV_PPHT_TWO_PER_CU")) pp->ppht_two_per_cu = pp->ppht_two_per_cu && pp->banks.size() >= 2;
    if ((rc = setup_roi(pp, p, bits)) || (rc = setup_skip_levels(pp, bits)) || (rc = setup_streams(pp)) || (rc = setup_banks(pp))) {
        lc_pipe_destroy(pp);
        return rc;
    }
    // the buffers were cleared with hipMemset, which may still be running on the null stream, and the pipeline's
    // streams do not wait for that stream: nothing may be submitted before the clears have landed
    {
        const hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { lc_pipe_destroy(pp); return fail(LC_EHIP, "pipeline set-up failed: %s", hipGetErrorString(e)); }
    }
    *out = pp;
    return LC_OK;
}

int lc_pipe_destroy(lc_pipe *pipe) {
    if (!pipe) return LC_OK;
    (void)hipSetDevice(pipe->ctx->device);
    if (pipe->st_lines) pipe_flush_pending(pipe);
    if (pipe->st_up) (void)hipStreamSynchronize(pipe->st_up);
    if (pipe->st_edge) (void)hipStreamSynchronize(pipe->st_edge);
    if (pipe->st_lines) (void)hipStreamSynchronize(pipe->st_lines);
    for (hipStream_t st : pipe->st_lines_more) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (pipe->st_post && pipe->st_post != pipe->st_edge && pipe->st_post != pipe->st_lines) { (void)hipStreamSynchronize(pipe->st_post); (void)hipStreamDestroy(pipe->st_post); }
    for (Slot &s : pipe->slots) if (s.done) (void)hipEventDestroy(s.done);
    for (Bank &bk : pipe->banks) {
        for (auto &ev : bk.ev) if (ev) (void)hipEventDestroy(ev);
        if (bk.h_res) (void)hipHostFree(bk.h_res);
    }
    if (pipe->ev_base) (void)hipEventDestroy(pipe->ev_base);
    if (pipe->st_lines && pipe->st_lines != pipe->st_edge) (void)hipStreamDestroy(pipe->st_lines);
    if (pipe->st_edge) (void)hipStreamDestroy(pipe->st_edge);
    if (pipe->st_up) (void)hipStreamDestroy(pipe->st_up);
    delete pipe;
    return LC_OK;
}

}  // extern "C"

