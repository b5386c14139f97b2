#!/usr/bin/env python3
"""Lane-pipeline benchmark (BASELINE.json metric: 1080p lane-pipeline frames/sec + achieved HBM GB/s).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the hot path (BGR2GRAY -> 5x5 blur -> Canny -> triangle ROI -> full-vote
accumulator -> HoughLinesP) over one batch of 256 synthetic 1080p road frames that already sit in HBM
when the timed region starts.  Each rank owns one GPU and its own clip (seeds 100 + rank); there is no
data-path collective, only a final gather of the per-frame segment lists to rank 0 (weak scaling).
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

# GPU_MAX_HW_QUEUES is left as the box sets it (4, HIP's own default): the pipeline picks streams that do not share a
# hardware queue by itself (lc_pipe_create), and the bench line says how many could not ("streams_sharing_a_queue").
# lc_pipe_wait gives up after this many seconds with a report of the events that fired instead of hanging a rank.
os.environ.setdefault("LANECV_WAIT_TIMEOUT_S", "60")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "image-processing_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H, BATCH, NSLOTS, SEG_CAP = 1920, 1080, 256, 6, 128
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak


def _gen(args):
    from lanecv import synth
    seed, t = args
    return synth.synth_road(W, H, seed, t)


def make_frames(seed, n, workers):
    """n distinct frames of one clip, generated before anything touches the GPU."""
    import numpy as np
    import multiprocessing as mp
    jobs = [(seed, t) for t in range(n)]
    if workers > 1:
        # close + join, not the pool's context manager: that ends the workers with SIGTERM, and under rocprofv3 --pmc
        # (whose library is already loaded, with its signal handler, when the workers are forked) a worker that catches
        # it starts the profiler's finalisation and may never end
        pool = mp.get_context("fork").Pool(workers)
        try:
            frames = pool.map(_gen, jobs, chunksize=4)
        finally:
            pool.close()
            pool.join()
    else:
        frames = [_gen(j) for j in jobs]
    return np.stack(frames)


def pmc_traffic(batch):
    """HBM bytes per k_front_stream launch from the committed rocprofv3 PMC passes (separate FETCH_SIZE and
    WRITE_SIZE runs, FETCH_SIZE doubled as gfx950 needs); None when the launch shape differs from the profiled one."""
    path = os.path.join(ROOT, "profiles", "round3_pmc_edge_stage.json")
    if batch != 256 or not os.path.exists(path):
        return None
    with open(path) as fh:
        return round(json.load(fh)["kernels"]["k_front_stream<true, 0, false, false>"]["traffic_bytes"])


def valu_issue(batch, launch_ms, device):
    """The bound the front kernel actually runs against: vector-instruction issue.  Instructions per launch from the
    committed PMC pass (SQ_INSTS_VALU), one wave instruction per SIMD every 4 cycles (what the packed 16-bit, v_dot4,
    v_perm, v_alignbyte and DPP forms the kernel consists of issue at on gfx950: tools/valu_rate.hip), the launch
    duration measured here."""
    import torch
    path = os.path.join(ROOT, "profiles", "round3_pmc_edge_stage.json")
    if batch != 256 or not os.path.exists(path):
        return None
    with open(path) as fh:
        insts = json.load(fh)["kernels"]["k_front_stream<true, 0, false, false>"]["SQ_INSTS_VALU"]
    props = torch.cuda.get_device_properties(device)
    mhz = (getattr(props, "clock_rate", 0) or 2400000) / 1e3
    peak = props.multi_processor_count * 4 * mhz * 1e6 / 4 / 1e9            # 1e9 wave instructions per second
    achieved = insts / (launch_ms * 1e-3) / 1e9
    return {"bound": "vector issue", "instructions_per_launch": insts, "cycles_per_instruction": 4, "clock_mhz": round(mhz),
            "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "1e9 wave instructions/s", "frac": round(achieved / peak, 4)}


def vote_rates(dev_ptr, batch, device):
    """LDS-atomic throughput of the Hough vote: the same frames through a pipeline that stops at the full-vote
    accumulator (k_vote on every frame; in the timed recipe HoughLinesP's kernel produces the accumulator on the
    side), one batch in flight, against this GPU's LDS atomic-add rate measured by the library's probe kernel."""
    import numpy as np
    from lanecv import _abi
    from lanecv.pipeline import LanePipeline, Recipe
    rec = Recipe.north_star(W, H)
    rec.segments = False
    pipe = LanePipeline(W, H, rec, max_batch=batch, nslots=1, device=device, timing=True)
    ms = []
    for _ in range(6):
        pipe.submit_device(0, dev_ptr, batch)
        pipe.wait(0)
        ms.append(pipe.stage_ms(0, _abi.STAGE_VOTE))
    ms = float(np.median(ms[1:]))
    points = int(pipe.read(0, _abi.BUF_NPOINTS).astype(np.int64).sum())
    numangle, numrho = pipe.numangle, pipe.numrho
    pipe.close()
    votes = float(points) * numangle
    peak = {"own_word": _abi.lds_atomic_rate(1, False, device), "run_of_8": _abi.lds_atomic_rate(8, False, device),
            "one_word_per_wave": _abi.lds_atomic_rate(64, False, device), "random_words": _abi.lds_atomic_rate(0, False, device),
            "own_word_returning": _abi.lds_atomic_rate(1, True, device),
            "random_words_returning": _abi.lds_atomic_rate(0, True, device)}
    ach = votes / (ms * 1e-3) / 1e9
    return {"kernel": "k_vote, one batch in flight", "launch_ms": round(ms, 4), "edge_points_per_launch": points,
            "votes_per_launch": votes, "accumulator": [numangle, numrho],
            "achieved": round(ach, 1), "unit": "1e9 votes/s (LDS atomic increments the accumulator receives: points x angles)",
            "lds_atomic_peak": {k: round(v, 1) for k, v in peak.items()},
            "frac_of_own_word_peak": round(ach / peak["own_word"], 4),
            "note": "peaks = int32 LDS atomic adds per second of the probe kernel (lc_lds_atomic_rate) on this GPU, 8 "
                    "workgroups of 256 lanes per CU, by address pattern; k_vote issues one lane-level atomic per vote (the lanes of a wavefront "
                    "take stretches of the sorted point list that lie n/64 points apart, so that they rarely meet in a bin)"}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def probe_cv2():
    """What `import cv2` did on this host, as observed: the reference's arithmetic lives in that module."""
    try:
        import cv2
    except Exception as e:          # ModuleNotFoundError here; anything else is worth reading too
        return None, {"cv2": "absent", "import_error": "%s: %s" % (type(e).__name__, e)}
    info = {"cv2": cv2.__version__, "threads": cv2.getNumThreads()}
    try:
        build = cv2.getBuildInformation()
        info["build_lines"] = [ln.strip() for ln in build.splitlines()
                               if any(k in ln for k in ("Version control", "CPU/HW features", "Baseline", "Intel IPP", "Parallel framework"))][:8]
    except Exception:
        pass
    try:
        if hasattr(cv2, "ipp"):
            cv2.ipp.setUseIPP(False)          # the generic code path is the contract
            info["ipp"] = "switched off"
    except Exception:
        pass
    return cv2, info


def _median_rate(fn, frames, repeats):
    """frames/s of fn over `frames`: a warm-up over a tenth of them, then the median of `repeats` timed passes."""
    import numpy as np
    fn(frames[:max(1, len(frames) // 10)])
    rates = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn(frames)
        rates.append(len(frames) / (time.perf_counter() - t0))
    return float(np.median(rates)), [round(r, 2) for r in rates]


def single_image():
    """BASELINE configs[0]: ONE 960 x 540 road image through gray -> GaussianBlur -> Canny -> ROI -> HoughLinesP, call by
    call through the cv2-shaped module (a NumPy array up and down per call), as the reference's image script runs it; the
    oracle on one host core beside it.  Milliseconds per image, median of 20."""
    import numpy as np
    import lanecv as cv2
    from lanecv import synth
    from lanecv.pipeline import triangle_roi
    from oracle import lane_oracle
    w, h = 960, 540
    img = synth.synth_road(w, h, seed=3, t=0)
    verts = triangle_roi(w, h)

    def script():
        gray = cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)
        blur = cv2.GaussianBlur(gray, (5, 5), 0)
        edges = cv2.Canny(blur, 100, 200)
        mask = np.zeros_like(edges)
        cv2.fillPoly(mask, verts, 255)
        masked = cv2.bitwise_and(edges, mask)
        return cv2.HoughLinesP(masked, rho=6, theta=np.pi / 60, threshold=160, lines=np.array([]), minLineLength=40, maxLineGap=25)

    def med(fn, n):
        fn()
        t = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            t.append(time.perf_counter() - t0)
        return float(np.median(t)) * 1e3

    got = script()
    roi = np.zeros((h, w), np.uint8)
    lane_oracle.fill_poly(roi, verts, 255)
    ref = lane_oracle.lane_frame(img, 6, True, 100, 200, roi, 6, np.pi / 60, 160, 40, 25, want_acc=False)["segments"]
    same = (got is None and len(ref) == 0) or (got is not None and got.reshape(-1, 4).shape == ref.shape and (got.reshape(-1, 4) == ref).all())
    return {"workload": "BASELINE configs[0]: one 960x540 image, cvtColor -> GaussianBlur -> Canny -> fillPoly/bitwise_and -> HoughLinesP, "
                        "six cv2-shaped calls on NumPy arrays",
            "ms_per_image": round(med(script, 20), 3),
            "oracle_one_core_ms": round(med(lambda: lane_oracle.lane_frame(img, 6, True, 100, 200, roi, 6, np.pi / 60, 160, 40, 25,
                                                                             want_acc=False), 5), 3),
            "segments": 0 if got is None else len(got), "segments_equal_oracle": bool(same)}


def cpu_baseline(frames, rec, roi_vertices, roi, threads, budget_s=24.0):
    """The CPU side of the same recipe on the same frames, on this host: the reference's own cv2 call sequence
    (`Simple Lane Detection with OpenCV_2(images).py:69-92` with the 5x5 blur of `F1_vo_Lane_Tracking.py:97` in front of
    Canny, per frame, the ROI mask rebuilt every frame as the scripts do) when cv2 can be imported, and in any case the
    oracle -- a scalar C port of the OpenCV arithmetic -- frame-parallel over `threads` host threads.  Warm-up over a
    tenth of the sample, median of five timed passes, stage by stage for one thread as well."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import lane_oracle as lo
    lo.lib()
    cv2, cv2_info = probe_cv2()

    def oracle_frame(f):
        lo.lane_frame(f, rec.color_code, rec.blur, rec.canny[0], rec.canny[1], roi, rec.rho, rec.theta,
                      rec.threshold, rec.min_line_length, rec.max_line_gap, want_acc=True, want_segments=True)

    # sample size: about budget_s seconds of CPU work in all (six passes over the sample on `threads` threads)
    t0 = time.perf_counter()
    oracle_frame(frames[0])
    one = time.perf_counter() - t0
    n = int(max(threads, min(len(frames), budget_s * threads / (6.6 * one))))
    sample = frames[:n]

    def oracle_all(fr):
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(oracle_frame, fr))

    rate, rates = _median_rate(oracle_all, sample, 5)
    # stage by stage, one thread, a few frames: where the port's time goes
    k = min(4, len(sample))
    stage = {"gray": 0.0, "blur": 0.0, "canny": 0.0, "roi_and": 0.0, "accumulator": 0.0, "hough_lines_p": 0.0}
    for f in sample[:k]:
        t = [time.perf_counter()]
        g = lo.cvt_gray(f, rec.color_code); t.append(time.perf_counter())
        if rec.blur:
            g = lo.gaussian5(g)
        t.append(time.perf_counter())
        e = lo.canny(g, rec.canny[0], rec.canny[1]); t.append(time.perf_counter())
        m = np.zeros_like(e)
        lo.fill_poly(m, roi_vertices, 255)
        e = lo.bitwise_and(e, m); t.append(time.perf_counter())
        lo.hough_accum(e, rec.rho, rec.theta); t.append(time.perf_counter())
        lo.hough_lines_p(e, rec.rho, rec.theta, rec.threshold, rec.min_line_length, rec.max_line_gap); t.append(time.perf_counter())
        for name, dt in zip(stage, np.diff(t)):
            stage[name] += dt * 1e3 / k
    out = {"value": round(rate, 2), "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": "%d of the same 1080p frames through the same recipe with the scalar C oracle, frame-parallel over %d threads; "
                     "warm-up over a tenth of them, median of 5 passes" % (len(sample), threads),
           "passes_fps": rates, "cpu_model": cpu_model(), "host_cpus": os.cpu_count(),
           "stage_ms_per_frame_one_thread": {k_: round(v, 3) for k_, v in stage.items()}}
    out.update(cv2_info)
    if cv2 is not None:
        # the literal reference sequence through cv2, per frame, (i) one cv2 thread, (ii) cv2's own threads, one process
        def cv2_frame(f):
            g = cv2.cvtColor(f, cv2.COLOR_BGR2GRAY if rec.color_code == 6 else cv2.COLOR_RGB2GRAY)
            if rec.blur:
                g = cv2.GaussianBlur(g, (5, 5), 0)
            e = cv2.Canny(g, rec.canny[0], rec.canny[1])
            m = np.zeros_like(e)
            cv2.fillPoly(m, roi_vertices, 255)
            e = cv2.bitwise_and(e, m)
            cv2.HoughLinesP(e, rec.rho, rec.theta, rec.threshold, lines=np.array([]), minLineLength=rec.min_line_length,
                            maxLineGap=rec.max_line_gap)

        def cv2_all(fr):
            for f in fr:
                cv2_frame(f)

        nthreads = cv2.getNumThreads()
        cv2.setNumThreads(1)
        r1, _ = _median_rate(cv2_all, sample[:max(8, n // threads)], 5)
        cv2.setNumThreads(nthreads)
        rn, _ = _median_rate(cv2_all, sample[:max(8, n // 4)], 5)

        def cv2_pool(fr):
            with ThreadPoolExecutor(threads) as ex:      # cv2 releases the GIL inside its calls
                list(ex.map(cv2_frame, fr))

        cv2.setNumThreads(1)
        rp, _ = _median_rate(cv2_pool, sample, 5)
        cv2.setNumThreads(nthreads)
        out["cv2_fps"] = {"one_thread": round(r1, 2), "cv2_threads": round(rn, 2), "frame_parallel_%d_threads" % threads: round(rp, 2)}
        out["value"] = round(max(rn, rp), 2)
        out["kind"] = "reference"
        out["port_fps"] = round(rate, 2)
        out["sample"] = out["sample"].replace("with the scalar C oracle", "through cv2 (value: the better of cv2's own threads and "
                                              "frame-parallel cv2) and with the scalar C oracle (port_fps)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--slots", type=int, default=NSLOTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streaming", action="store_true")
    ap.add_argument("--same-input", action="store_true", help="resubmit one device batch every step (the round-1 figure)")
    ap.add_argument("--clips", type=int, default=0,
                    help="BASELINE configs[4] shape: this many clips sharded clip-major over the ranks (0: one clip of "
                         "steps x batch frames per rank)")
    ap.add_argument("--frames-per-clip", type=int, default=2048)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world
    batch = args.batch

    import numpy as np
    host_threads = min(16, os.cpu_count() or 1)
    sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
    from lanecv import shard
    # Default: one clip per rank (clip 100 + rank), `steps` batches long.  --clips C: C clips of --frames-per-clip
    # frames, clip c on rank c mod world (shard.clip_assignment), each rank working through its clips one after
    # the other.  Either way a clip is fed from two distinct device batches that alternate, so that a slot never sees
    # the frames it processed last time: nothing of the pipeline's change tracking (it rewrites only what differs
    # from the slot's previous batch) is flattered by identical input.
    nbatches = 1 if args.same_input else 2
    if args.clips > 0:
        my_clips = shard.clip_assignment(args.clips, world, rank)
        frames_per_clip = args.frames_per_clip
        distinct = min(64, nbatches * batch)       # frames generated per clip; tiled to the two batches on the device
    else:
        my_clips = [rank]
        frames_per_clip = args.steps * batch
        distinct = nbatches * batch
    host_clips = [make_frames(100 + c, distinct, host_threads) for c in my_clips]
    frames = host_clips[0][:batch] if my_clips else make_frames(100, batch, host_threads)

    import torch                       # first: liblanecv.so then binds to the HIP runtime torch loaded
    import torch.distributed as dist
    # LANECV_BENCH_REHEARSE=1: every rank on GPU 0 and gloo between them -- runs the multi-rank code on a one-GPU
    # box (sharding, barriers, max over ranks, the gather); its numbers mean nothing
    rehearse = os.environ.get("LANECV_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run the process group is set up even for a single rank: the one-GPU line then runs
    # with RCCL's queues beside the pipeline's, like the ranks of a larger job
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    coll_device = "cpu" if rehearse else "cuda"

    if rehearse:
        import faulthandler
        faulthandler.dump_traceback_later(40, exit=True)      # a stuck rank says where, then ends the run

    def note(msg):
        if rehearse:
            print("[rank %d] %s" % (rank, msg), file=sys.stderr, flush=True)

    from lanecv import _abi
    from lanecv.pipeline import LanePipeline, Recipe
    rec = Recipe.north_star(W, H)
    if os.environ.get("LANECV_BENCH_START_AT"):       # diagnostics: several benches on one GPU starting in step
        while time.time() < float(os.environ["LANECV_BENCH_START_AT"]):
            time.sleep(0.0005)
    pipe = LanePipeline(W, H, rec, max_batch=batch, nslots=args.slots, seg_capacity=SEG_CAP, device=local_rank, timing=True)

    # inputs resident in HBM: per clip `nbatches` batches of `batch` frames
    frame_bytes = 3 * W * H
    dev_clips = []
    for hc in host_clips:
        d = torch.from_numpy(hc).cuda()
        if len(hc) < nbatches * batch:           # tile the distinct frames, each batch in another order
            idx = torch.cat([(torch.arange(batch) * 5 + 11 * b) % len(hc) for b in range(nbatches)]).cuda()
            d = d[idx].contiguous()
        dev_clips.append(d)
    torch.cuda.synchronize()
    ptr = dev_clips[0].data_ptr() if dev_clips else None
    # the jobs of this rank in order: (device pointer, frames) batch after batch, clip after clip
    jobs = []
    for d in dev_clips:
        for k in range((frames_per_clip + batch - 1) // batch):
            jobs.append((d.data_ptr() + (k % nbatches) * batch * frame_bytes, min(batch, frames_per_clip - k * batch)))
    note("pipeline and frames ready")

    results = {}
    stage_names = {"front": _abi.STAGE_FRONT, "hyst": _abi.STAGE_HYST, "final": _abi.STAGE_FINAL,
                   "edge": _abi.STAGE_EDGE, "points": _abi.STAGE_POINTS, "vote": _abi.STAGE_VOTE,
                   "lines": _abi.STAGE_LINES, "total": _abi.STAGE_TOTAL}

    trace = [] if os.environ.get("LANECV_BENCH_TRACE") == "1" else None    # diagnostics: when each wait of the timed run returned

    def run(job_list, collect_stages=None, keep=None):
        """Pushes the jobs through the slots in order; keep (a shard.RowSink) receives each batch's segments unpadded."""
        inflight = []

        def finish(s):
            seg, cnt = pipe.wait(s)
            if trace is not None and keep is not None:
                trace.append(time.perf_counter())
            results[s] = (seg, cnt)
            if keep is not None:
                keep.append(seg, cnt)
            if collect_stages is not None:
                collect_stages.append({n: pipe.stage_ms(s, i) for n, i in stage_names.items()})

        for k, (p_dev, n) in enumerate(job_list):
            slot = k % args.slots
            if len(inflight) == args.slots:
                finish(inflight.pop(0))
            pipe.submit_device(slot, p_dev, n)
            inflight.append(slot)
        for s in inflight:
            finish(s)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    if jobs:
        run([jobs[k % len(jobs)] for k in range(args.warmup)])
    note("warm-up done")
    stages, kept = [], shard.RowSink(sum(n for _, n in jobs), SEG_CAP)      # the rank's result buffer: allocated before the clock starts
    # (as timeit does: Python's collector off while the clock runs -- a full collection is a pause of tens of milliseconds that
    # has nothing to do with the path measured)
    import gc
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    run(jobs, stages, kept)
    note("timed batches done")
    # the final result gather, the path's only exchange between ranks: every frame's segments of every clip, unpadded
    gathered = shard.gather_compact(kept.rows(), kept.counts(), dst=0, device=coll_device if use_dist else None)
    note("gathered")
    if trace is not None:
        t_g = time.perf_counter()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if trace is not None:
        print("[trace] waits returned at (ms): %s; gathered at %.3f; end %.3f"
              % (" ".join("%.3f" % ((t - t0) * 1e3) for t in trace), (t_g - t0) * 1e3, dt * 1e3), file=sys.stderr, flush=True)
        t_s = time.perf_counter()
        torch.cuda.synchronize()
        print("[trace] a synchronize of the idle device: %.3f ms" % ((time.perf_counter() - t_s) * 1e3), file=sys.stderr, flush=True)
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    nclips_total = args.clips if args.clips > 0 else world
    total_frames = nclips_total * frames_per_clip
    fps = total_frames / dt
    timed_steps = len(jobs)
    if rank == 0:
        # what arrived must be every frame of every clip, in each rank's own clip order
        got_frames = sum(len(c) for _, c in gathered)
        got_clips = sum(len(shard.clip_assignment(args.clips, world, r)) if args.clips > 0 else 1 for r in range(len(gathered)))
        if got_frames != total_frames or got_clips != nclips_total or any(len(r) != int(c.sum()) for r, c in gathered):
            sys.exit("final gather is incomplete: %d frames of %d clips arrived, %d of %d expected"
                     % (got_frames, got_clips, total_frames, nclips_total))
    seg, cnt = results[(timed_steps - 1) % args.slots] if jobs else (None, np.zeros(1, np.int32))

    out = None
    if rank == 0:
        mean = {n: float(np.mean([s[n] for s in stages])) for n in stage_names}
        # dominant HBM kernel: k_front_stream: 3 B/px of BGR read and two class bit planes (2/8 B/px) written = 3.25 B per
        # pixel it classifies.  With the triangle ROI it classifies only the strip x row-segment units within 15 pixels of
        # the ROI (the pipeline proves per frame that the rest cannot matter, and classifies a frame it cannot prove it for
        # again in full): the algorithmic bytes of a launch are 3.25 B x those pixels x frames.  The rate the same launch
        # would have at 3.25 B/px of whole frames is reported beside it, as a rate only.
        first_row, pixels_classified, redone = pipe.roi_skip((timed_steps - 1) % args.slots)
        front_bytes = 3.25 * pixels_classified * batch
        frame_bytes_equiv = 3.25 * W * H * batch
        front_ms = mean["front"]
        achieved = front_bytes / (front_ms * 1e-3) / 1e9
        stage_bytes = (3.0 * pixels_classified + 1.0 * W * H) * batch      # BGR of the pixels classified in + the masked edge map out
        out = {
            "metric": "1080p lane-pipeline frames/sec", "value": round(fps, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps if args.clips == 0 else timed_steps, "warmup": args.warmup,
            "ms_per_step": round(dt / max(timed_steps, 1) * 1e3, 3), "higher_is_better": True,
            "scaling": "weak" if args.clips == 0 else "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("synthetic 1080p road frames (BASELINE configs[2]), one clip of %d frames per GPU" % frames_per_clip
                                    if args.clips == 0 else
                                    "%d synthetic 1080p clips x %d frames (BASELINE configs[4] with %d instead of 2,000 frames a clip), "
                                    "clip c on GPU c mod %d" % (args.clips, frames_per_clip, frames_per_clip, world))
                                   + ", batch=%d per step (two distinct batches of a clip alternating), recipe north_star: "
                                     "BGR2GRAY -> GaussianBlur 5x5 -> Canny(100,200) -> triangle ROI -> full-vote accumulator "
                                     "-> HoughLinesP(6, pi/60, 160, 40, 25); inputs resident in HBM; every frame's segments "
                                     "gathered to rank 0 at the end" % batch,
                       "width": W, "height": H, "batch": batch, "slots_in_flight": args.slots,
                       "sharding": "clip-major (clip c on GPU c mod N), no data-path collective, one final unpadded segment gather",
                       "clips": nclips_total, "frames_per_clip": frames_per_clip},
            "roofline": {"bound": "hbm", "kernel": "k_front_stream<true, 0, false, false>",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(batch),
                         "algorithmic_bytes_per_launch": front_bytes, "launch_ms": round(front_ms, 4),
                         "pixels_classified_per_frame": pixels_classified, "first_row_classified": first_row,
                         "frames_classified_again_in_full_last_batch": redone,
                         "whole_frame_equivalent_GBs": round(frame_bytes_equiv / (front_ms * 1e-3) / 1e9, 1),
                         "note": "achieved = algorithmic bytes (3.25 B x pixels_classified_per_frame x frames of a launch) / launch "
                                 "duration; whole_frame_equivalent_GBs = the same launch priced at 3.25 B/px of whole 1920 x 1080 frames (a "
                                 "rate of frames, not of bytes: the kernel neither reads nor classifies the rest); "
                                 "duration = HIP events around the kernel on the stream it runs on, inside the timed region "
                                 "(HoughLinesP of the previous batch runs beside it); traffic = FETCH_SIZE (x2, gfx950) + "
                                 "WRITE_SIZE per launch from the committed rocprofv3 PMC passes of the same launch shape"},
            "stage_ms": {k: round(v, 4) for k, v in mean.items()},
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "unset (HIP default 4)"),
            "streams_sharing_a_queue": pipe.shared_queues(),
            "segments_frame0": int(cnt[0]), "gathered_ranks": len(gathered), "gathered_frames": got_frames,
            "gathered_segments": int(sum(len(r) for r, _ in gathered)), "hyst_closure_rounds_max_total": list(pipe.hyst_rounds(0)),
        }

        # exclusive timings: one batch in flight, nothing else on the GPU
        excl = []
        for _ in range(5):
            pipe.submit_device(0, ptr, batch)
            pipe.wait(0)
            excl.append({n: pipe.stage_ms(0, i) for n, i in stage_names.items()})
        em = {n: float(np.median([s[n] for s in excl])) for n in stage_names}
        # whole edge stage (front + hysteresis + finalize): 3 B/px in, the 1 B/px masked edge map out.  Taken from the
        # exclusive run: in the timed region its short kernels sit on a stream of their own and their event-to-event
        # times are mostly waiting for room beside the next batch's front kernel.
        edge_x = em["front"] + em["hyst"] + em["final"]
        out["roofline_edge_stage"] = {"kernels": "k_front_stream + k_hysteresis + k_finalize_roi, one batch in flight",
                                      "algorithmic_bytes_per_launch": stage_bytes, "launch_ms": round(edge_x, 4),
                                      "achieved": round(stage_bytes / (edge_x * 1e-3) / 1e9, 1),
                                      "frac": round(stage_bytes / (edge_x * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        out["roofline_exclusive"] = {"kernel": "k_front_stream<true, 0, false, false>, nothing else on the GPU", "launch_ms": round(em["front"], 4),
                                     "achieved": round(front_bytes / (em["front"] * 1e-3) / 1e9, 1),
                                     "frac": round(front_bytes / (em["front"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "whole_frame_equivalent_GBs": round(frame_bytes_equiv / (em["front"] * 1e-3) / 1e9, 1),
                                     "stage_ms": {k: round(v, 4) for k, v in em.items()}}
        out["roofline_exclusive"]["valu_issue"] = valu_issue(batch, em["front"], local_rank)

        out["vote"] = vote_rates(ptr, batch, local_rank)

        if not args.no_streaming:
            # BASELINE configs[2] as written -- "streaming with pinned async upload": pinned host ring -> hipMemcpyAsync
            # on the upload stream -> pipeline, slots overlapped.  PCIe-bound (6.2 MB per frame), so it carries its own
            # roofline: the host-to-device rate of a plain pinned copy of one batch, measured here.  Never `value`.
            pinned = [pipe.pinned_frames(batch) for _ in range(args.slots)]
            reps = (batch + len(frames) - 1) // len(frames)
            for p in pinned:
                p[:] = np.concatenate([frames] * reps)[:batch]
            ssteps = max(args.slots * 3, 9)
            for slot in range(args.slots):           # first touch of the upload stream outside the timed loop
                pipe.submit(slot, pinned[slot])
            for slot in range(args.slots):
                pipe.wait(slot)
            t1 = time.perf_counter()
            inflight = []
            for k in range(ssteps):
                slot = k % args.slots
                if len(inflight) == args.slots:
                    pipe.wait(inflight.pop(0))
                pipe.submit(slot, pinned[slot])
                inflight.append(slot)
            for s in inflight:
                pipe.wait(s)
            sfps = ssteps * batch / (time.perf_counter() - t1)
            # with an ROI only what the kernels read is copied: of the rows from a little above the ROI down, per row
            # segment the columns of the units the front kernel classifies
            rows_from = pipe.rows_uploaded((ssteps - 1) % args.slots)
            sent_bytes = pipe.bytes_uploaded((ssteps - 1) % args.slots)
            src = torch.from_numpy(pinned[0].reshape(-1))           # the same pinned block, copied by torch alone
            dst = torch.empty(src.numel(), dtype=torch.uint8, device="cuda")
            best = 0.0
            for _ in range(4):
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                dst.copy_(src, non_blocking=True)
                torch.cuda.synchronize()
                best = max(best, src.numel() / (time.perf_counter() - t2) / 1e9)
            out["streaming_fps_pcie_inclusive"] = round(sfps, 1)
            out["streaming_roofline"] = {"bound": "pcie", "achieved": round(sfps * sent_bytes / 1e9, 2), "peak": round(best, 2),
                                         "unit": "GB/s", "frac": round(sfps * sent_bytes / 1e9 / best, 4),
                                         "bytes_sent_per_frame": sent_bytes, "first_row_sent": rows_from,
                                         "whole_frame_equivalent_GBs": round(sfps * frame_bytes / 1e9, 2),
                                         "note": "achieved = bytes actually copied (of rows %d..%d of every frame the column windows "
                                                 "the front kernel's units read with the triangle ROI, a 3-D copy per row segment; a "
                                                 "frame that needs the rest after all is fetched when its slot is waited for) x "
                                                 "frames/s; peak = host-to-device rate of one pinned %d-frame batch copied alone, "
                                                 "measured in this run" % (rows_from, H - 1, batch)}
            del dst

        if world == 1 and not args.no_cpu_baseline:
            out["single_image"] = single_image()
            out["cpu_baseline"] = cpu_baseline(frames[:min(len(frames), 256)], rec, rec.roi_vertices, pipe.roi, host_threads)
        print(json.dumps(out), flush=True)

    pipe.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
