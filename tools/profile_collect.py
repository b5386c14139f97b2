"""Turns the output of tools/profile_round.sh (gpurun_out/<dir>) into the files kept under profiles/:
   python tools/profile_collect.py gpurun_out/prof_r2 round2"""
import collections, csv, glob, json, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
KERNEL = "k_front_stream<"


def counters(sub):
    f = glob.glob(src + '/' + sub + '/**/*_counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if KERNEL in r['Kernel_Name']:
            agg[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
            agg[r['Dispatch_Id']]['_dur'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    return [agg[k] for k in sorted(agg, key=int)]


def mean(rows, key, lo=1, hi=6):        # tools/front_probe.py: launches 0-5 with Canny (100, 200), 6-11 with thresholds nothing passes
    v = [r[key] for r in rows[lo:hi]]
    return sum(v) / len(v)


fetch, write, sq = counters('fetch'), counters('write'), counters('sq')
W, H, B = 1920, 1080, 256
roof = json.load(open(src + '/bench.json'))["roofline"]
pixels = int(roof["pixels_classified_per_frame"])
unit_rows = 64
fk, wk = mean(fetch, 'FETCH_SIZE'), mean(write, 'WRITE_SIZE')
out = {
    "what": "HBM traffic and instruction counts of k_front_stream from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE and the SQ counters "
            "in separate runs), workload = 256 synthetic 1080p frames per launch, recipe north_star without the Hough stages (command: "
            "rocprofv3 --pmc <counters> -- python3 tools/front_probe.py; launches 2-6 of each pass, Canny thresholds (100, 200), are "
            "averaged). With the triangle ROI the kernel classifies the %d pixels per frame of the units within 15 pixels of it "
            "(of %d)." % (pixels, W * H),
    "units": "FETCH_SIZE / WRITE_SIZE are KiB per launch; bytes = KiB * 1024",
    "gfx950_correction": "FETCH_SIZE is doubled (it tallies the kernel's 128-B read requests at 64 B); compare own_read_bytes, the kernel's "
                         "own count: 3 B per pixel classified * 256/248 (strip halo) * (unit rows + 8)/unit rows (halo rows of a unit).",
    "kernels": {"k_front_stream<true, 0, false, false>": {
        "FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
        "fetch_bytes_corrected": round(2 * fk * 1024), "write_bytes": round(wk * 1024),
        "traffic_bytes": round(2 * fk * 1024 + wk * 1024),
        "own_read_bytes": round(3 * pixels * B * (256 / 248) * ((unit_rows + 8) / unit_rows)),
        "algorithmic_bytes_pixels_classified": 3.25 * pixels * B, "algorithmic_bytes_whole_frames": 3.25 * W * H * B,
        "SQ_INSTS_VALU": round(mean(sq, 'SQ_INSTS_VALU')), "SQ_INSTS_SALU": round(mean(sq, 'SQ_INSTS_SALU')),
        "SQ_INSTS_VALU_thresholds_nothing_passes": round(mean(sq, 'SQ_INSTS_VALU', 7, 12)),
        "SQ_ACTIVE_INST_VALU": round(mean(sq, 'SQ_ACTIVE_INST_VALU')), "SQ_WAVE_CYCLES": round(mean(sq, 'SQ_WAVE_CYCLES')),
        "SQ_WAIT_INST_ANY": round(mean(sq, 'SQ_WAIT_INST_ANY')), "SQ_BUSY_CYCLES": round(mean(sq, 'SQ_BUSY_CYCLES')),
        "launch_us_under_pmc": round(mean(sq, '_dur'), 1),
        "note": "the first version of the kernel (whole frames, before the instruction work): SQ_INSTS_VALU 259.7 M, SQ_INSTS_SALU 127.3 M per launch; the "
                "whole-frame form of the present kernel (LANECV_NO_ROW_SKIP=1): 225.8 M / 46.0 M; rows 524..1079 only: 138.3 M / 27.2 M"}}}
json.dump(out, open('profiles/%s_pmc_edge_stage.json' % tag, 'w'), indent=1)
shutil.copy(glob.glob(src + '/bench/**/*kernel_stats.csv', recursive=True)[0], 'profiles/%s_kernel_stats.csv' % tag)
for a, b in (("bench_under_rocprof.json", "bench_under_rocprof.json"), ("bench.json", "bench.json"),
             ("bench_b512_under_rocprof.json", "bench_b512_under_rocprof.json")):
    shutil.copy(src + '/' + a, 'profiles/%s_%s' % (tag, b))
print(json.dumps(out["kernels"], indent=1))
