"""GPU parity of pipelines whose batches leave most CUs free (BASELINE configs[1]: 720p x 32, configs[3]: 4K x 64).
HoughLinesP is one workgroup per frame, so such a pipeline runs ONE HoughLinesP launch for a group of consecutive
slots.  What can go wrong there: slots used out of order, a wait in the middle of a group, a short batch in a slot
that is not the group's last, and a frame's trouble (point-list overflow, too many segments, too many points for the
LDS kernel) reported to another slot of its group.  Everything is compared with the oracle bit for bit, through the
C ABI, at the frame sizes and batch sizes BASELINE.json names."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_frame(oracle, f, rec, roi):
    return oracle.lane_frame(f, rec.color_code, rec.blur, rec.canny[0], rec.canny[1], roi, rec.rho, rec.theta,
                             rec.threshold, rec.min_line_length, rec.max_line_gap, want_acc=True, want_segments=True)


def _oracle_all(oracle, frames, rec, roi):
    with ThreadPoolExecutor(8) as ex:
        return list(ex.map(lambda f: _oracle_frame(oracle, f, rec, roi), frames))


def _check(pipe, slot, want, idx, accumulator):
    """waits for `slot` and compares frame i of its batch with want[idx[i]]"""
    from lanecv import _abi
    seg, cnt = pipe.wait(slot)
    masked = pipe.read(slot, _abi.BUF_MASKED)
    acc = pipe.read(slot, _abi.BUF_ACC) if accumulator else None
    assert len(cnt) == len(idx)
    for i, k in enumerate(idx):
        r = want[k]
        assert cnt[i] == len(r["segments"]), (slot, i, cnt[i], len(r["segments"]))
        assert np.array_equal(seg[i, :cnt[i]], r["segments"]), (slot, i)
        assert np.array_equal(masked[i], r["masked"]), (slot, i)
        if acc is not None:
            assert np.array_equal(acc[i], r["acc"]), (slot, i)


def test_groups_720p_curve_batch32(oracle):
    """BASELINE configs[1] as configured: 1280x720, `Simple Lane Detection with OpenCV_5(mp4).py:66-84`'s recipe,
    batches of 32, eight slots -> two groups of four slots, 128 frames per HoughLinesP launch."""
    from lanecv import synth
    from lanecv.pipeline import LanePipeline, Recipe
    w, h, B, nslots, nd = 1280, 720, 32, 8, 12
    rec = Recipe.curve(w, h)
    pipe = LanePipeline(w, h, rec, max_batch=B, nslots=nslots, seg_capacity=256)
    try:
        assert pipe.layout()[0] == 4
        distinct = [synth.synth_road(w, h, seed=2 + (t % 3), t=17 * t) for t in range(nd)]
        want = _oracle_all(oracle, distinct, rec, pipe.roi)

        def batch(rot, n=B):
            idx = [(i + rot) % nd for i in range(n)]
            return np.stack([distinct[k] for k in idx]), idx

        # in order, every slot full, all in flight
        idxs = {}
        for s in range(nslots):
            fr, idxs[s] = batch(s)
            pipe.submit(s, fr)
        for s in range(nslots):
            _check(pipe, s, want, idxs[s], False)
        # out of order inside both groups (each slot then gets a launch of its own)
        for s in (1, 0, 3, 2, 6, 7, 5, 4):
            fr, idxs[s] = batch(5 + s)
            pipe.submit(s, fr)
        for s in (3, 1, 2, 0, 4, 5, 6, 7):
            _check(pipe, s, want, idxs[s], False)
        # a wait on slot 0 before slot 1 is submitted: slot 0's frames get a launch now, the rest of the group later
        fr, idxs[0] = batch(3)
        pipe.submit(0, fr)
        _check(pipe, 0, want, idxs[0], False)
        for s in (1, 2, 3):
            fr, idxs[s] = batch(7 + s)
            pipe.submit(s, fr)
        fr, idxs[0] = batch(9)
        pipe.submit(0, fr)               # slot 0 again while 1..3 still wait for their launch: out of order now
        for s in (2, 0, 1, 3):
            _check(pipe, s, want, idxs[s], False)
        # short batches in the middle of a group, in slots that held full batches before: the frames behind them
        # must read as empty to the group's launch
        sizes = {0: B, 1: 5, 2: 1, 3: B, 4: 7, 5: B, 6: 31, 7: 2}
        for s in range(nslots):
            fr, idxs[s] = batch(2 * s + 1, sizes[s])
            pipe.submit(s, fr)
        for s in range(nslots):
            _check(pipe, s, want, idxs[s], False)
        # and full batches again behind them
        for s in range(nslots):
            fr, idxs[s] = batch(11 - s)
            pipe.submit(s, fr)
        for s in reversed(range(nslots)):
            _check(pipe, s, want, idxs[s], False)
    finally:
        pipe.close()


def test_groups_4k_north_star_batch64(oracle):
    """BASELINE configs[3] as configured: 3840x2160, the north-star recipe with its full-vote accumulator, batches of
    64, four slots -> two groups of two slots, 128 frames per HoughLinesP launch."""
    from lanecv import synth
    from lanecv.pipeline import LanePipeline, Recipe
    w, h, B, nslots = 3840, 2160, 64, 4
    rec = Recipe.north_star(w, h)
    pipe = LanePipeline(w, h, rec, max_batch=B, nslots=nslots, seg_capacity=256)
    try:
        assert pipe.layout()[0] == 2
        distinct = [synth.synth_road(w, h, seed=4, t=t) for t in (0, 37, 90, 151)]
        want = _oracle_all(oracle, distinct, rec, pipe.roi)
        nd = len(distinct)
        host = [np.empty((B, h, w, 3), np.uint8) for _ in range(nslots)]     # one per slot: uploads are asynchronous
        for rnd, sizes in enumerate(([B] * nslots, [B, 9, B, B], [3, B, 64, 1])):
            idxs = {}
            for s in range(nslots):
                idxs[s] = [(i + s + rnd) % nd for i in range(sizes[s])]
                for i, k in enumerate(idxs[s]):
                    host[s][i] = distinct[k]
                pipe.submit(s, host[s][:sizes[s]])
            for s in ((0, 1, 2, 3), (2, 3, 1, 0), (1, 0, 3, 2))[rnd]:
                _check(pipe, s, want, idxs[s], True)
    finally:
        pipe.close()


def _dense_frame(oracle, rec, w, h, lo, hi):
    from lanecv import synth
    noise = synth.adversarial_frames(w, h)["noise"]
    for rows in range(40, h + 1, 20):
        f = np.full((h, w, 3), 90, np.uint8)
        f[:rows] = noise[:rows]
        n = np.count_nonzero(_oracle_frame(oracle, f, rec, None)["masked"])
        if lo <= n <= hi:
            return f, n
    raise AssertionError("no noise height gives an edge count in [%d, %d]" % (lo, hi))


def test_group_reports_reach_the_right_slot(oracle):
    """Inside one group: a frame with more points than the LDS kernel takes (done by the general kernel), a frame with
    more points than the pipeline keeps per frame (LC_ECAP for ITS slot only), and a frame with more segments than
    seg_capacity (its count says so; the other frames are untouched)."""
    from lanecv import synth, _abi
    from lanecv.pipeline import LanePipeline, Recipe
    w, h, B = 640, 360, 3
    rec = Recipe(color_code=6, blur=False, canny=(100, 200), roi_vertices=None, accumulator=True,
                 threshold=120, min_line_length=30, max_line_gap=5)
    cap_points = w * h // 8 + 1024
    dense, nd = _dense_frame(oracle, rec, w, h, 17500, 27000)
    over, no = _dense_frame(oracle, rec, w, h, cap_points + 500, w * h)
    road = [synth.synth_road(w, h, seed=9, t=t) for t in range(3)]
    want_road = _oracle_all(oracle, road, rec, None)
    want_dense = _oracle_frame(oracle, dense, rec, None)
    segcap = 8
    assert len(want_dense["segments"]) > segcap or any(len(r["segments"]) > segcap for r in want_road)
    pipe = LanePipeline(w, h, rec, max_batch=B, nslots=4, seg_capacity=segcap)
    try:
        assert pipe.layout()[0] == 2
        for rnd in range(2):
            pipe.submit(0, np.stack([road[0], dense, road[1]]))
            pipe.submit(1, np.stack([road[2], over]))             # short batch, overflowing frame: group 0's last slot
            pipe.submit(2, np.stack([over, road[1], road[0]]))
            pipe.submit(3, np.stack(road))
            # slot 0: fine, whatever its neighbour in the group did
            seg, cnt = pipe.wait(0)
            masked, acc = pipe.read(0, _abi.BUF_MASKED), pipe.read(0, _abi.BUF_ACC)
            for i, r in enumerate((want_road[0], want_dense, want_road[1])):
                assert cnt[i] == len(r["segments"]), (rnd, i)
                k = min(cnt[i], segcap)
                assert np.array_equal(seg[i, :k], r["segments"][:k]), (rnd, i)
                assert np.array_equal(masked[i], r["masked"]) and np.array_equal(acc[i], r["acc"]), (rnd, i)
            assert pipe.read(0, _abi.BUF_NPOINTS)[1] == nd > 16383
            for s in (1, 2):
                with pytest.raises(_abi.error) as e:
                    pipe.wait(s)
                assert e.value.code == _abi.LC_ECAP, (rnd, s)
            seg, cnt = pipe.wait(3)
            for i, r in enumerate(want_road):
                assert cnt[i] == len(r["segments"]), (rnd, i)
                k = min(cnt[i], segcap)
                assert np.array_equal(seg[i, :k], r["segments"][:k]), (rnd, i)
    finally:
        pipe.close()
