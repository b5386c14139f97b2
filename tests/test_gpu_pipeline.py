"""GPU parity of the batched pipeline (C ABI lc_pipe_*) against the CPU oracle,
recipe by recipe: gray, blurred, edge, masked images, the full-vote accumulator
and the HoughLinesP segments are all compared bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_frame(oracle, f, rec, roi, car=None):
    return oracle.lane_frame(f, rec.color_code, rec.blur, rec.canny[0], rec.canny[1], roi, rec.rho, rec.theta,
                             rec.threshold, rec.min_line_length, rec.max_line_gap, car_mask=car,
                             want_acc=True, want_segments=True)


def _pinned(pipe, frames):
    """The frames in page-locked memory of the pipeline's (only from such memory are frames uploaded in part)."""
    buf = pipe.pinned_frames(len(frames))
    buf[:] = frames
    return buf


def _check_batch(oracle, pipe, frames, rec, slot=0, check_inter=True):
    from lanecv import _abi
    pipe.submit(slot, frames)
    segs = pipe.collect(slot) if rec.segments else (pipe.wait(slot), None)[1]
    masked = pipe.read(slot, _abi.BUF_MASKED)
    inter = {}
    if check_inter:
        inter["gray"] = pipe.read(slot, _abi.BUF_GRAY)
        inter["edges"] = pipe.read(slot, _abi.BUF_EDGES)
        if rec.blur:
            inter["blur"] = pipe.read(slot, _abi.BUF_BLUR)
    acc = pipe.read(slot, _abi.BUF_ACC) if rec.accumulator else None
    npts = pipe.read(slot, _abi.BUF_NPOINTS) if (rec.accumulator or rec.segments) else None
    for i, f in enumerate(frames):
        r = _oracle_frame(oracle, f, rec, pipe.roi, rec.car_mask)
        for k, v in inter.items():
            assert (v[i] == r[k]).all(), (i, k)
        assert (masked[i] == r["masked"]).all(), i
        if acc is not None:
            assert (acc[i] == r["acc"]).all(), i
        if npts is not None:
            assert npts[i] == np.count_nonzero(r["masked"])
        if rec.segments:
            want = r["segments"]
            got = np.zeros((0, 4), np.int32) if segs[i] is None else segs[i].reshape(-1, 4)
            assert got.shape == want.shape and (got == want).all(), i


@pytest.mark.parametrize("w,h,name", [(960, 540, "simple"), (1280, 720, "curve"), (1920, 1080, "north_star"),
                                      (640, 360, "f1"), (641, 359, "north_star"), (130, 70, "simple")])
def test_recipe_parity(oracle, w, h, name):
    from lanecv import synth
    from lanecv.pipeline import LanePipeline, Recipe
    rec = getattr(Recipe, name)(w, h)
    if name != "f1":
        rec.accumulator = True
    if w < 200:
        rec.threshold, rec.min_line_length = 20, 10
    pipe = LanePipeline(w, h, rec, max_batch=4, nslots=2, seg_capacity=2048, keep_intermediates=True)
    frames = np.stack([synth.synth_road(w, h, seed=9, t=t) for t in range(3)])
    _check_batch(oracle, pipe, frames, rec, slot=0)
    adv = synth.adversarial_frames(w, h)
    frames = np.stack([adv["ramps"], adv["border_impulses"], adv["black"], adv["checker3"]])
    _check_batch(oracle, pipe, frames, rec, slot=1)
    pipe.close()


def test_roi_matches_fillpoly_of_the_scripts(oracle):
    from lanecv.pipeline import LanePipeline, Recipe
    for (w, h) in ((960, 540), (1920, 1080), (641, 359)):
        pipe = LanePipeline(w, h, Recipe.simple(w, h), max_batch=1, nslots=1)
        roi = np.zeros((h, w), np.uint8)
        oracle.fill_poly(roi, [np.array([(0, h), (w / 2, h / 2), (w, h)]).astype(np.int32)], 255)
        assert (pipe.roi == roi).all()
        pipe.close()


@pytest.mark.parametrize("nslots", [3, 5])
def test_slots_overlap_and_reuse(oracle, nslots):
    """All slots in flight at once, each reused: results must not depend on the slot or the order.  (Batches this
    small take two HoughLinesP streams in turn with three slots, three with five.)"""
    from lanecv import synth
    from lanecv.pipeline import LanePipeline, Recipe
    w, h = 960, 540
    rec = Recipe.north_star(w, h)
    pipe = LanePipeline(w, h, rec, max_batch=8, nslots=nslots, timing=True)
    pinned = [pipe.pinned_frames(8) for _ in range(nslots)]
    batches = []
    for b in range(2 * nslots):
        fr = np.stack([synth.synth_road(w, h, seed=20 + b, t=t) for t in range(8 if b != 4 else 3)])
        batches.append(fr)
    results = {}
    inflight = []
    for b, fr in enumerate(batches):
        slot = b % nslots
        if len(inflight) == nslots:
            ob, oslot = inflight.pop(0)
            results[ob] = pipe.collect(oslot)
        pinned[slot][:len(fr)] = fr
        pipe.submit(slot, pinned[slot][:len(fr)])
        inflight.append((b, slot))
    for ob, oslot in inflight:
        results[ob] = pipe.collect(oslot)
    for b, fr in enumerate(batches):
        assert len(results[b]) == len(fr)
        for i, f in enumerate(fr):
            want = _oracle_frame(oracle, f, rec, pipe.roi)["segments"]
            got = np.zeros((0, 4), np.int32) if results[b][i] is None else results[b][i].reshape(-1, 4)
            assert got.shape == want.shape and (got == want).all(), (b, i)
    assert pipe.stage_ms(0, 1) > 0
    pipe.close()


def test_full_size_properties():
    """BASELINE sizes beyond what the oracle checks in seconds: size-independent properties.
    1080p batch of 32 and 4K batch of 4: accumulator rows each sum to the number of edge points;
    every segment ends on edge pixels inside the ROI and obeys the length rule; identical frames
    give identical results; masked edges are 0/255 and vanish outside the ROI."""
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    for (w, h, n) in ((1920, 1080, 32), (3840, 2160, 4)):
        rec = Recipe.north_star(w, h)
        pipe = LanePipeline(w, h, rec, max_batch=n, nslots=1)
        base = [synth.synth_road(w, h, seed=3, t=t) for t in range(4)]
        frames = np.stack([base[i % 4] for i in range(n)])
        pipe.submit(0, frames)
        segs = pipe.collect(0)
        masked = pipe.read(0, _abi.BUF_MASKED)
        acc = pipe.read(0, _abi.BUF_ACC)
        npts = pipe.read(0, _abi.BUF_NPOINTS)
        assert set(np.unique(masked)) <= {0, 255}
        assert not masked[:, pipe.roi == 0].any()
        for i in range(n):
            assert npts[i] == np.count_nonzero(masked[i]) and npts[i] > 0
            assert (acc[i].sum(axis=1) == npts[i]).all()
            assert segs[i] is not None
            for x1, y1, x2, y2 in segs[i].reshape(-1, 4):
                assert masked[i, y1, x1] and masked[i, y2, x2]
                assert max(abs(x2 - x1), abs(y2 - y1)) >= 40
            j = i % 4
            assert (masked[i] == masked[j]).all() and (acc[i] == acc[j]).all()
            assert (segs[i] == segs[j]).all()
        pipe.close()


@pytest.mark.parametrize("w,h", [(644, 359), (248, 100), (200, 37), (8, 5), (252, 64), (1000, 130), (64, 1), (12, 300)])
@pytest.mark.parametrize("blur", [False, True])
def test_streaming_front_sizes(oracle, w, h, blur):
    """Widths that are multiples of 4 take the streaming front kernel: strips that end inside a
    64-pixel word, strips wholly outside the image, frames shorter than the vertical halo."""
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    rec = Recipe(color_code=7 if not blur else 6, blur=blur, canny=(40, 90), roi_vertices=None,
                 segments=False, accumulator=False)
    pipe = LanePipeline(w, h, rec, max_batch=5, nslots=1, keep_intermediates=True)
    adv = synth.adversarial_frames(w, h)
    frames = np.stack([synth.synth_road(w, h, seed=4, t=1), adv["noise"], adv["ramps"], adv["border_impulses"],
                       adv["checker2"]])
    pipe.submit(0, frames)
    pipe.wait(0)
    gray, edges = pipe.read(0, _abi.BUF_GRAY), pipe.read(0, _abi.BUF_EDGES)
    bl = pipe.read(0, _abi.BUF_BLUR) if blur else None
    for i, f in enumerate(frames):
        g = oracle.cvt_gray(f, rec.color_code)
        assert (gray[i] == g).all(), i
        if blur:
            g = oracle.gaussian5(g)
            assert (bl[i] == g).all(), i
        assert (edges[i] == oracle.canny(g, 40, 90)).all(), i
    pipe.close()


@pytest.mark.parametrize("keep", [False, True])
def test_one_slot_many_batches(oracle, keep):
    """One slot, batch after batch of very different frames and frame counts: the pipeline keeps state between
    batches (it rewrites only what changed in the planes and the masked map, its flags clean themselves), so
    every batch must still come out exactly as from a fresh pipeline."""
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    w, h = 640, 360
    rec = Recipe.north_star(w, h)
    rec.threshold, rec.min_line_length = 60, 20
    pipe = LanePipeline(w, h, rec, max_batch=4, nslots=1, seg_capacity=4096, keep_intermediates=keep)
    adv = synth.adversarial_frames(w, h)
    road = [synth.synth_road(w, h, seed=12, t=t) for t in range(4)]
    batches = [
        [adv["noise"], road[0], adv["checker2"], road[1]],
        [road[2], adv["black"], road[3], adv["white"]],
        [adv["black"]],
        [road[1], adv["ramps"]],
        [adv["noise"], adv["noise"], adv["black"], road[0]],
        [adv["black"], adv["black"], adv["black"], adv["black"]],
        [road[3], road[2], road[1], road[0]],
    ]
    for b, frames in enumerate(batches):
        frames = np.stack(frames)
        pipe.submit(0, frames)
        segs = pipe.collect(0)
        masked, acc = pipe.read(0, _abi.BUF_MASKED), pipe.read(0, _abi.BUF_ACC)
        edges = pipe.read(0, _abi.BUF_EDGES) if keep else None
        for i, f in enumerate(frames):
            r = _oracle_frame(oracle, f, rec, pipe.roi)
            assert (masked[i] == r["masked"]).all(), (b, i)
            assert (acc[i] == r["acc"]).all(), (b, i)
            if keep:
                assert (edges[i] == r["edges"]).all(), (b, i)
            got = np.zeros((0, 4), np.int32) if segs[i] is None else segs[i].reshape(-1, 4)
            assert got.shape == r["segments"].shape and (got == r["segments"]).all(), (b, i)
    pipe.close()


@pytest.mark.parametrize("w,h,lo,hi", [(640, 360, 17500, 27000), (1280, 720, 36000, 90000)])
def test_dense_frames_take_the_flagged_paths(oracle, w, h, lo, hi):
    """Frames whose points, accumulator and mask words do not fit the LDS share a launch leaves a frame are flagged by
    the HoughLinesP kernel and done again when the slot is collected: with the mask left in device memory if points and
    accumulator alone fit a whole CU's LDS (the smaller case), else (the larger case) with the point list left there as
    well; either way the kernel also makes their full-vote accumulator.  Flagged and ordinary frames share a batch.
    (The global-memory kernel behind these two takes recipes of more than 64 angles -- tests/test_gpu_ops.py -- and, with
    LANECV_NO_SCATTER_KERNEL=1, the flagged frames of tests/test_gpu_switches.py.)
    The frame is noise above a flat lower part, sized so that the count lands in the wanted range and under the
    pipeline's point capacity (an eighth of the pixels)."""
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    rec = Recipe(color_code=6, blur=False, canny=(100, 200), roi_vertices=None, accumulator=True,
                 threshold=120, min_line_length=30, max_line_gap=5)
    noise = synth.adversarial_frames(w, h)["noise"]
    dense = None
    for rows in range(60, h, 20):
        f = np.full((h, w, 3), 90, np.uint8)
        f[:rows] = noise[:rows]
        n = np.count_nonzero(_oracle_frame(oracle, f, rec, None)["masked"])
        if lo <= n <= hi:
            dense = f
            break
    assert dense is not None, "no noise height gives an edge count in the wanted range"
    frames = np.stack([dense, synth.synth_road(w, h, seed=9, t=1), dense[:, ::-1].copy()])
    pipe = LanePipeline(w, h, rec, max_batch=3, nslots=2, seg_capacity=16384)
    for slot in (0, 1, 0):
        _check_batch(oracle, pipe, frames, rec, slot=slot, check_inter=False)
    assert lo <= pipe.read(0, _abi.BUF_NPOINTS).max() <= hi
    pipe.close()


def test_4k_parity(oracle):
    """Four distinct 3840x2160 frames through the north-star recipe: masked map, full-vote accumulator (60 x 2000
    bins) and segments against the oracle, bit for bit -- the size at which HoughLinesP's LDS row map and the
    reachable-bin ranges are widest."""
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    w, h = 3840, 2160
    rec = Recipe.north_star(w, h)
    pipe = LanePipeline(w, h, rec, max_batch=4, nslots=1, seg_capacity=1024)
    frames = np.stack([synth.synth_road(w, h, seed=4, t=t) for t in (0, 37, 90, 151)])
    _check_batch(oracle, pipe, frames, rec, slot=0, check_inter=False)
    assert (pipe.numangle, pipe.numrho) == (60, 2000)
    pipe.close()


def test_bench_launch_shape_parity(oracle):
    """The benchmark's own launch shape -- 1080p, batches of 256 frames resident in HBM, three slots in flight,
    submitted with lc_pipe_submit_device -- on CHANGING content: four device batches hold sixteen distinct frames
    in four rotations, so every slot sees different frames each time it is reused (the pipeline rewrites only what
    changed since the slot's previous batch).  Every frame's segments, and masked map and accumulator of every
    batch, are compared with the oracle."""
    from concurrent.futures import ThreadPoolExecutor
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    import hiprt
    w, h, B, nslots, ndistinct = 1920, 1080, 256, 3, 16
    rec = Recipe.north_star(w, h)
    pipe = LanePipeline(w, h, rec, max_batch=B, nslots=nslots, seg_capacity=128)
    distinct = [synth.synth_road(w, h, seed=100 + (t % 2), t=11 * t) for t in range(ndistinct)]
    with ThreadPoolExecutor(8) as ex:
        want = list(ex.map(lambda f: _oracle_frame(oracle, f, rec, pipe.roi), distinct))
    rotations = [0, 5, 10, 3]
    # One of the four device batches also holds frames the ROI scheme cannot settle from the units near the ROI -- faint
    # vertical edges made edges only by their strong upper ends far above it, and a frame of noise (which also has more
    # edge points than the LDS-resident HoughLinesP kernel takes): the second pass over flagged frames and the general
    # HoughLinesP kernel at the benchmark's launch shape, between clean batches in the same slots.
    def faint_edge(x, ytop, flip=False):
        g = np.full((h, w), 60, np.int32)
        g[:ytop, x:] += 120
        g[ytop:, x:] += 60
        f = np.repeat(g[:, :, None], 3, axis=2).astype(np.uint8)
        return f[:, ::-1].copy() if flip else f

    special = {7: faint_edge(w // 2 + 60, 150), 100: faint_edge(w // 2 + 200, 40, flip=True), 200: faint_edge(w // 2 + 90, 300),
               255: synth.adversarial_frames(w, h)["noise"]}
    with ThreadPoolExecutor(4) as ex:
        want_special = dict(zip(special, ex.map(lambda f: _oracle_frame(oracle, f, rec, pipe.roi), special.values())))
    assert np.count_nonzero(want_special[255]["masked"]) > 16383
    dirty = 2                                       # the device batch that holds them
    dev = []
    try:
        for b, rot in enumerate(rotations):
            host = np.stack([distinct[(i + rot) % ndistinct] for i in range(B)])
            if b == dirty:
                for i, f in special.items():
                    host[i] = f
            d = hiprt.DeviceBuffer(host.nbytes)
            d.upload(host)
            dev.append(d)
            del host

        def check(k, slot):
            seg, cnt = pipe.wait(slot)
            masked = pipe.read(slot, _abi.BUF_MASKED)
            acc = pipe.read(slot, _abi.BUF_ACC)
            b = k % len(rotations)
            rot = rotations[b]
            assert pipe.roi_skip(slot)[2] == (len(special) if b == dirty else 0), (k, pipe.roi_skip(slot))
            for i in range(B):
                r = want_special[i] if (b == dirty and i in special) else want[(i + rot) % ndistinct]
                assert cnt[i] == len(r["segments"]), (k, i)
                m = min(cnt[i], seg.shape[1])
                assert np.array_equal(seg[i, :m], r["segments"][:m]), (k, i)
                assert np.array_equal(masked[i], r["masked"]), (k, i)
                assert np.array_equal(acc[i], r["acc"]), (k, i)

        inflight = []
        for k in range(9):                        # every slot is used three times, each time on another rotation
            slot = k % nslots
            if len(inflight) == nslots:
                check(*inflight.pop(0))
            pipe.submit_device(slot, dev[k % len(rotations)].ptr.value, B)
            inflight.append((k, slot))
        for it in inflight:
            check(*it)
        # a smaller batch into a used slot: the kernel's per-unit change tracking must not depend on the batch size
        pipe.submit_device(0, dev[1].ptr.value, 100)
        seg, cnt = pipe.wait(0)
        masked = pipe.read(0, _abi.BUF_MASKED)
        for i in range(100):
            r = want[(i + rotations[1]) % ndistinct]
            assert cnt[i] == len(r["segments"]) and np.array_equal(seg[i, :cnt[i]], r["segments"]), i
            assert np.array_equal(masked[i], r["masked"]), i
    finally:
        pipe.close()
        for d in dev:
            d.free()


def test_temporaries_stay_alive_per_slot(oracle):
    """Frames handed over as lists or strided views are copied to a temporary before the asynchronous upload;
    each slot keeps its own until it has been waited for."""
    from lanecv import synth
    from lanecv.pipeline import LanePipeline, Recipe
    w, h = 640, 360
    rec = Recipe.simple(w, h)
    rec.threshold, rec.min_line_length = 60, 20
    pipe = LanePipeline(w, h, rec, max_batch=4, nslots=3, seg_capacity=2048)
    wide = [np.stack([synth.synth_road(2 * w, h, seed=30 + s, t=t) for t in range(4)]) for s in range(3)]
    for s in range(3):
        pipe.submit(s, wide[s][:, :, ::2])        # non-contiguous: submit makes a copy
    for s in range(3):
        segs = pipe.collect(s)
        for i in range(4):
            r = _oracle_frame(oracle, np.ascontiguousarray(wide[s][i, :, ::2]), rec, pipe.roi)["segments"]
            got = np.zeros((0, 4), np.int32) if segs[i] is None else segs[i].reshape(-1, 4)
            assert got.shape == r.shape and (got == r).all(), (s, i)
    pipe.close()


def test_rows_above_the_roi_are_skipped_exactly(oracle, monkeypatch):
    """A pipeline whose ROI begins far down the frame classifies only the rows from a little above the ROI: per frame
    it proves that nothing above can reach the ROI through weak pixels, and classifies the frame again from row 0
    where it cannot.  Road frames must take the short way, frames built to need the rows above must be caught
    (a faint vertical edge that is only made an edge by its strong upper end, far above the ROI), and a slot that
    saw such a frame must be clean again for the next batch."""
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    monkeypatch.setenv("LANECV_SKIP_OFF_BATCHES", "0")      # stay at the tightest level whatever the share of frames redone
    w, h = 960, 540
    rec = Recipe.north_star(w, h)
    pipe = LanePipeline(w, h, rec, max_batch=6, nslots=2, seg_capacity=512)

    def faint_edge(x, ytop, flip=False):
        g = np.full((h, w), 60, np.int32)
        g[:ytop, x:] += 120                      # strong: 2.5 * 120 > 200
        g[ytop:, x:] += 60                       # weak alone: 100 < 2.5 * 60 <= 200
        f = np.repeat(g[:, :, None], 3, axis=2).astype(np.uint8)
        return f[:, ::-1].copy() if flip else f

    road = [synth.synth_road(w, h, seed=41, t=t) for t in range(4)]
    first, redone = pipe.row_skip(0)
    assert first == h // 2 - 15 and redone == 0
    batches = [
        np.stack(road),
        np.stack([road[0], faint_edge(w // 2 + 40, 100), road[1], faint_edge(w // 2 + 90, 30, flip=True), road[2]]),
        np.stack(road[::-1] + road[:2]),         # the same slot after frames that were redone: the rows above are clean again
        np.stack([faint_edge(w // 2 + 40, 100)] * 3),
        np.stack(road[:3]),
    ]
    want_redone = [0, 2, 0, 3, 0]
    for b, frames in enumerate(batches):
        assert np.count_nonzero(_oracle_frame(oracle, frames[1 if b in (1,) else 0], rec, pipe.roi)["masked"]) > 0
        pipe.submit(0, _pinned(pipe, frames))
        segs = pipe.collect(0)
        masked, acc = pipe.read(0, _abi.BUF_MASKED), pipe.read(0, _abi.BUF_ACC)
        assert pipe.row_skip(0)[1] == want_redone[b], b
        # (submitted from pinned host memory: only the rows from a little above the first one classified were copied, and the
        # frames that needed the rows above were fetched in full when the slot was waited for)
        assert 0 < pipe.rows_uploaded(0) <= first
        for i, f in enumerate(frames):
            r = _oracle_frame(oracle, f, rec, pipe.roi)
            assert (masked[i] == r["masked"]).all(), (b, i)
            assert (acc[i] == r["acc"]).all(), (b, i)
            got = np.zeros((0, 4), np.int32) if segs[i] is None else segs[i].reshape(-1, 4)
            assert got.shape == r["segments"].shape and (got == r["segments"]).all(), (b, i)
    pipe.close()


@pytest.mark.parametrize("blur", [False, True])
def test_a_later_frames_need_for_the_full_pass_is_not_lost(oracle, blur):
    """The one frame of a batch that must be classified again in full sits BEHIND frames that need many hysteresis rounds
    (1080p road frames without the blur): its workgroup finishes long before the first frame's, on another XCD.  A flag
    word shared by the batch, cleared by the first frame's workgroup in the same launch, lost that frame's flag -- the
    host now reads every frame's own word.  The frame's edges inside the ROI exist only through the rows far above it."""
    import dataclasses
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    w, h = 1920, 1080
    rec = dataclasses.replace(Recipe.simple(w, h), blur=blur, accumulator=True)
    g = np.full((h, w), 60, np.int32)
    g[:440, 1168:] += 120
    g[440:, 1168:] += 45 if not blur else 60
    faint = np.repeat(g[:, :, None], 3, axis=2).astype(np.uint8)
    road = [synth.synth_road(w, h, seed=5, t=t) for t in (3, 4)]
    for frames in ([road[0], faint], [road[0], road[1], faint], [faint, road[0]]):
        pipe = LanePipeline(w, h, rec, max_batch=len(frames), nslots=1, seg_capacity=512)
        frames = np.stack(frames)
        pipe.submit(0, frames)
        segs = pipe.collect(0)
        masked = pipe.read(0, _abi.BUF_MASKED)
        assert pipe.row_skip(0)[1] == 1
        for i, f in enumerate(frames):
            r = _oracle_frame(oracle, f, rec, pipe.roi)
            assert (masked[i] == r["masked"]).all(), (len(frames), i)
            got = np.zeros((0, 4), np.int32) if segs[i] is None else segs[i].reshape(-1, 4)
            assert got.shape == r["segments"].shape and (got == r["segments"]).all(), (len(frames), i)
        pipe.close()


def test_roi_skipping_switches_off_and_on_again(oracle, monkeypatch):
    """A batch in which more than a quarter of the frames needed all units after all switches the ROI scheme off for a
    number of batches (here 2 instead of 256), which are classified in full from the start; the slot then skips again,
    after the plane rows such batches left outside the needed units have been cleared.  Results never change."""
    monkeypatch.setenv("LANECV_SKIP_OFF_BATCHES", "2")
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    w, h = 960, 540
    rec = Recipe.north_star(w, h)
    pipe = LanePipeline(w, h, rec, max_batch=4, nslots=1, seg_capacity=512)
    g = np.full((h, w), 60, np.int32)
    g[:100, w // 2 + 40:] += 120
    g[100:, w // 2 + 40:] += 60
    faint = np.repeat(g[:, :, None], 3, axis=2).astype(np.uint8)
    noise = synth.adversarial_frames(w, h)["noise"]
    road = [synth.synth_road(w, h, seed=43, t=t) for t in range(4)]
    batches = [np.stack([faint, road[0], faint, road[1]]),     # half of it redone at level 0 (units near the ROI): one level out
               np.stack([faint, noise, faint, faint]),         # level 1 (all rows from above the ROI down): all redone, out again
               np.stack([noise, road[2], road[3], road[0]]),   # in full from the start
               np.stack(road),                                  # in full
               np.stack(road[::-1]),                            # level 1 again: the rows above must be clean
               np.stack(road),                                  # level 1
               np.stack(road[::-1]),                            # level 0 again: the units away from the ROI must be clean
               np.stack(road)]
    want_redone = [2, 4, 4, 4, 0, 0, 0, 0]
    want_pixels_less = [True, True, False, False, True, True, True, True]
    for b, frames in enumerate(batches):
        pipe.submit(0, frames)
        segs = pipe.collect(0)
        masked, acc = pipe.read(0, _abi.BUF_MASKED), pipe.read(0, _abi.BUF_ACC)
        first, pixels, redone = pipe.roi_skip(0)
        assert redone == want_redone[b] and (pixels < w * h) == want_pixels_less[b], (b, first, pixels, redone)
        for i, f in enumerate(frames):
            r = _oracle_frame(oracle, f, rec, pipe.roi)
            assert (masked[i] == r["masked"]).all(), (b, i)
            assert (acc[i] == r["acc"]).all(), (b, i)
            got = np.zeros((0, 4), np.int32) if segs[i] is None else segs[i].reshape(-1, 4)
            assert got.shape == r["segments"].shape and (got == r["segments"]).all(), (b, i)
    pipe.close()


def test_short_batch_at_a_level_change_clears_every_frame(oracle, monkeypatch):
    """A slot that comes back to a tighter level of the ROI scheme has the plane rows outside that level's units cleared
    -- for every frame it can hold, not only for the frames of the batch that happens to come first: here that batch
    is short (the end of a clip), and a later full batch at the same level must not flood from what an earlier full
    batch, classified in full, left behind in the frames the short one did not touch.  The last batch is built to
    show it: a faint vertical edge, weak everywhere (so not an edge at all), exactly under where the earlier frames
    had a strong one above the rows this level classifies."""
    monkeypatch.setenv("LANECV_SKIP_OFF_BATCHES", "3")
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    w, h, B = 960, 540, 6
    rec = Recipe.north_star(w, h)
    pipe = LanePipeline(w, h, rec, max_batch=B, nslots=2, seg_capacity=512)
    x = w // 2 + 40

    def column(ytop):
        g = np.full((h, w), 60, np.int32)
        g[:ytop, x:] += 120                      # strong above ytop
        g[ytop:, x:] += 60                       # weak below
        return np.repeat(g[:, :, None], 3, axis=2).astype(np.uint8)

    first = pipe.row_skip(0)[0]
    assert first == h // 2 - 15
    strong_top, strong_to_split, weak_only = column(100), column(first), column(0)
    road = [synth.synth_road(w, h, seed=47, t=t) for t in range(B)]
    assert not _oracle_frame(oracle, weak_only, rec, pipe.roi)["masked"].any()
    plan = [                                     # (slot, frames, pixels classified < whole frame, frames redone)
        (0, [strong_top] * B, True, B),                      # units near the ROI: all redone -> one level out for 3 batches
        (0, [strong_top] * B, True, B),                      # rows from above the ROI down: all redone -> in full
        (0, [strong_to_split] * B, False, B),                # in full: strong bits just above the first row classified
        (0, [strong_to_split] * B, False, B),
        (0, [strong_to_split] * B, False, B),
        (0, road[:2], True, 0),                              # back at the row level with a SHORT batch: the clear
        (1, [strong_top] * B, True, B),                      # units level on the other slot: redone -> out to the row level
        (0, [weak_only] * B, True, B),                       # the row level again, full batch, no clear in between: the weak
                                                             # column crosses every ring row and stays weak -> redone, no edge
        (0, road, False, B),                                 # (in full: the batch before was redone whole)
    ]
    for b, (slot, frames, less, redone) in enumerate(plan):
        frames = np.stack(frames)
        pipe.submit(slot, frames)
        segs = pipe.collect(slot)
        masked, acc = pipe.read(slot, _abi.BUF_MASKED), pipe.read(slot, _abi.BUF_ACC)
        _, pixels, got_redone = pipe.roi_skip(slot)
        assert (pixels < w * h) == less and got_redone == redone, (b, pixels, got_redone)
        want = {}
        for i, f in enumerate(frames):
            key = hash(f.tobytes())
            r = want.get(key) or _oracle_frame(oracle, f, rec, pipe.roi)
            want[key] = r
            assert (masked[i] == r["masked"]).all(), (b, i)
            assert (acc[i] == r["acc"]).all(), (b, i)
            got = np.zeros((0, 4), np.int32) if segs[i] is None else segs[i].reshape(-1, 4)
            assert got.shape == r["segments"].shape and (got == r["segments"]).all(), (b, i)
    pipe.close()


def test_rows_follow_when_the_level_moves_out_before_the_launch(oracle, monkeypatch):
    """Frames submitted from host memory are copied from the first row the ROI level of the moment reads.  With two
    slots to a bank the kernels are queued later, when the bank is full -- and by then the pipeline may have moved to
    classifying whole frames: the rows that were left out must follow before the launch."""
    monkeypatch.setenv("LANECV_SKIP_OFF_BATCHES", "50")
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    w, h, B = 960, 540, 4
    rec = Recipe.north_star(w, h)
    pipe = LanePipeline(w, h, rec, max_batch=B, nslots=4, seg_capacity=512)
    assert pipe.layout()[0] == 2
    g = np.full((h, w), 60, np.int32)
    g[:100, w // 2 + 40:] += 120
    g[100:, w // 2 + 40:] += 60
    faint = np.repeat(g[:, :, None], 3, axis=2).astype(np.uint8)
    road = [synth.synth_road(w, h, seed=51, t=t) for t in range(2 * B)]

    def check(slot, frames):
        segs = pipe.collect(slot)
        masked, acc = pipe.read(slot, _abi.BUF_MASKED), pipe.read(slot, _abi.BUF_ACC)
        for i, f in enumerate(frames):
            r = _oracle_frame(oracle, f, rec, pipe.roi)
            assert (masked[i] == r["masked"]).all() and (acc[i] == r["acc"]).all(), (slot, i)
            got = np.zeros((0, 4), np.int32) if segs[i] is None else segs[i].reshape(-1, 4)
            assert got.shape == r["segments"].shape and (got == r["segments"]).all(), (slot, i)

    first = pipe.row_skip(0)[0]
    a, b_ = np.stack(road[:B]), np.stack(road[B:])
    a, b_ = _pinned(pipe, a), _pinned(pipe, b_)
    pipe.submit(0, a)                              # bank 0, first slot: copied from a little above row `first`, waits for slot 1
    assert 0 < pipe.rows_uploaded(0) <= first
    for rnd in range(2):                           # bank 1: frames that need the second pass, twice -> out to whole frames
        fr = np.stack([faint] * B)
        pipe.submit(2, fr)
        pipe.submit(3, fr)
        check(2, fr)
        check(3, fr)
    pipe.submit(1, b_)                             # fills bank 0: its launch classifies whole frames
    assert pipe.rows_uploaded(1) == 0
    check(0, a)
    check(1, b_)
    assert pipe.rows_uploaded(0) == 0 and pipe.roi_skip(0)[1] == w * h
    pipe.close()


def test_host_frames_are_uploaded_in_windows(oracle, monkeypatch):
    """Frames submitted from host memory while the pipeline classifies only the units near the ROI are copied window by
    window: per row segment the columns its units read.  The slot's device buffer is filled with noise first, so a read
    outside the windows would show; a frame that needs the full pass after all is fetched whole; and while the pipeline
    is a level out whole rows are sent."""
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    w, h, B = 960, 540, 4
    rec = Recipe.north_star(w, h)
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (B, h, w, 3), dtype=np.uint8)
    g = np.full((h, w), 60, np.int32)
    g[:100, w // 2 + 40:] += 120
    g[100:, w // 2 + 40:] += 60
    faint = np.repeat(g[:, :, None], 3, axis=2).astype(np.uint8)
    road = [synth.synth_road(w, h, seed=61, t=t) for t in range(B)]

    def run(pipe, slot, frames, redone):
        pipe.upload(slot, noise[:len(frames)])           # what an earlier batch might have left in the slot's buffer
        pipe.submit(slot, _pinned(pipe, frames))
        segs = pipe.collect(slot)
        masked, acc = pipe.read(slot, _abi.BUF_MASKED), pipe.read(slot, _abi.BUF_ACC)
        assert pipe.roi_skip(slot)[2] == redone
        for i, f in enumerate(frames):
            r = _oracle_frame(oracle, f, rec, pipe.roi)
            assert (masked[i] == r["masked"]).all() and (acc[i] == r["acc"]).all(), (slot, i)
            got = np.zeros((0, 4), np.int32) if segs[i] is None else segs[i].reshape(-1, 4)
            assert got.shape == r["segments"].shape and (got == r["segments"]).all(), (slot, i)

    monkeypatch.setenv("LANECV_SKIP_OFF_BATCHES", "0")      # stay at the tightest level whatever is redone
    pipe = LanePipeline(w, h, rec, max_batch=B, nslots=2, seg_capacity=512)
    first = pipe.row_skip(0)[0]
    rows_bytes = 3 * w * (h - (first - 8))
    run(pipe, 0, np.stack(road), 0)
    windows_bytes = pipe.bytes_uploaded(0)
    assert 0 < windows_bytes < rows_bytes and 0 < pipe.rows_uploaded(0) <= first
    run(pipe, 1, np.stack([road[0], faint, road[1]]), 1)   # the flagged frame is fetched whole when the slot is waited for
    assert pipe.bytes_uploaded(1) == windows_bytes
    run(pipe, 0, np.stack([faint] * B), B)
    run(pipe, 1, np.stack(road[::-1]), 0)
    monkeypatch.setenv("LANECV_NO_UPLOAD_WINDOWS", "1")
    run(pipe, 0, np.stack(road), 0)
    assert pipe.bytes_uploaded(0) == rows_bytes
    monkeypatch.delenv("LANECV_NO_UPLOAD_WINDOWS")
    pipe.submit(1, np.stack(road))                       # ordinary (pageable) memory: whole frames, one linear copy
    pipe.wait(1)
    assert pipe.rows_uploaded(1) == 0 and pipe.bytes_uploaded(1) == 3 * w * h
    pipe.close()

    monkeypatch.setenv("LANECV_SKIP_OFF_BATCHES", "2")      # a batch with many frames redone: a level out for two batches
    pipe = LanePipeline(w, h, rec, max_batch=B, nslots=2, seg_capacity=512)
    run(pipe, 0, np.stack([faint] * B), B)
    assert pipe.bytes_uploaded(0) == windows_bytes
    run(pipe, 1, np.stack(road[::-1]), 0)                   # all columns from above the ROI down: whole rows
    assert pipe.bytes_uploaded(1) == rows_bytes
    run(pipe, 0, np.stack(road), 0)
    run(pipe, 1, np.stack(road), 0)                         # (its launch is back at the units near the ROI; its submit was not yet)
    run(pipe, 0, np.stack(road[::-1]), 0)
    assert pipe.bytes_uploaded(0) == windows_bytes
    pipe.close()


@pytest.mark.parametrize("layout", [{"LANECV_TAIL_STREAMS": "0"}, {"LANECV_TAIL_STREAMS": "1"}, {"LANECV_TAIL_STREAMS": "2"},
                                    {"LANECV_TAIL_STREAMS": "2", "LANECV_HYST_LEAD": "1"},
                                    {"LANECV_TAIL_STREAMS": "2", "LANECV_HYST_LEAD": "1", "LANECV_EAGER_REDO": "1"}],
                         ids=["post-stream", "tails", "tails+edge-hysteresis", "hysteresis-leads", "hysteresis-leads+eager"])
@pytest.mark.parametrize("group", [1, 2])
def test_stream_layouts_agree(oracle, monkeypatch, layout, group):
    """The same batches through every stream layout of a pipeline with six slots, as banks of one slot and of two (post stream; the short kernels on the
    HoughLinesP streams; hysteresis on the edge stream; hysteresis as the first workgroups of the next front launch),
    all slots in flight, device-resident frames, with frames among them that the units near the ROI cannot settle
    (classified again in the stream once such frames are about, else when the slot is waited for) and a short batch:
    masked map, accumulator and segments must equal the oracle's whatever the layout, the frames redone be counted,
    and a slot reused after them be clean."""
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    for k, v in layout.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("LANECV_SKIP_OFF_BATCHES", "0")
    monkeypatch.setenv("LANECV_GROUP_SIZE", str(group))
    w, h, nslots = 960, 540, 6
    rec = Recipe.north_star(w, h)
    pipe = LanePipeline(w, h, rec, max_batch=6, nslots=nslots, seg_capacity=512)
    assert pipe.layout()[0] == group

    def faint_edge(x, ytop):
        g = np.full((h, w), 60, np.int32)
        g[:ytop, x:] += 120
        g[ytop:, x:] += 60
        return np.repeat(g[:, :, None], 3, axis=2).astype(np.uint8)

    road = [synth.synth_road(w, h, seed=77, t=t) for t in range(6)]
    batches, want_redone = [], []
    for b in range(3 * nslots):
        fr = [road[(b + i) % 6] for i in range(6 if b != 5 else 2)]
        n_bad = 0
        if b in (2, 3, 7):
            fr[1] = faint_edge(w // 2 + 40 + 10 * b, 60 + 5 * b)
            n_bad = 1
        batches.append(np.stack(fr))
        want_redone.append(n_bad)
    ref = {}

    def check(b, slot):
        segs = pipe.collect(slot)
        masked, acc = pipe.read(slot, _abi.BUF_MASKED), pipe.read(slot, _abi.BUF_ACC)
        assert pipe.row_skip(slot)[1] == want_redone[b], b
        for i, f in enumerate(batches[b]):
            key = f.tobytes()
            if key not in ref:
                ref[key] = _oracle_frame(oracle, f, rec, pipe.roi)
            r = ref[key]
            assert (masked[i] == r["masked"]).all(), (b, i)
            assert (acc[i] == r["acc"]).all(), (b, i)
            got = np.zeros((0, 4), np.int32) if segs[i] is None else segs[i].reshape(-1, 4)
            assert got.shape == r["segments"].shape and (got == r["segments"]).all(), (b, i)

    inflight = []
    for b, fr in enumerate(batches):
        slot = b % nslots
        if len(inflight) == nslots:
            check(*inflight.pop(0))
        ptr = pipe.upload(slot, fr)
        pipe.submit_device(slot, ptr, len(fr))
        inflight.append((b, slot))
    for ob, oslot in inflight:
        check(ob, oslot)
    pipe.close()
