"""Turns the passes of tools/pmc_more.sh (gpurun_out/<dir>) into the two files kept under profiles/:
   python tools/pmc_collect_more.py gpurun_out/r3f/pmc round3"""
import collections, csv, glob, json, sys

src, tag = sys.argv[1], sys.argv[2]


def rows(sub, pat):
    f = glob.glob(src + '/' + sub + '/**/*_counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            agg[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
            agg[r['Dispatch_Id']]['_us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            agg[r['Dispatch_Id']]['_name'] = r['Kernel_Name']
    return [agg[k] for k in sorted(agg, key=int)]


def mean(rs, key, lo, hi):
    v = [r[key] for r in rs[lo:hi] if key in r]
    return sum(v) / len(v) if v else None


front = {}
for sub in "abc":
    rs = rows(sub, "k_front_stream<")
    for key in sorted(set(k for r in rs for k in r) - {"_name"}):
        # tools/front_probe.py: launches 0-5 with Canny (100, 200), 6-11 with thresholds nothing passes; the first of each is left out
        front.setdefault(key if key != "_us" else "launch_us_pass_" + sub, {})
        front[key if key != "_us" else "launch_us_pass_" + sub] = {"canny_100_200": round(mean(rs, key, 1, 6), 1),
                                                                  "thresholds_nothing_passes": round(mean(rs, key, 7, 12), 1)}
c = {k: v["canny_100_200"] for k, v in front.items()}
simds, cus, ses, clock_ghz = 1024, 256, 32, 2.4
launch_cycles = c["SQ_CYCLES"] / ses
derived = {
    "launch_cycles_per_shader_engine": round(launch_cycles),
    "waves": round(c["SQ_WAVES"]),
    "mean_resident_waves": round(4 * c["SQ_WAVE_CYCLES"] / launch_cycles),
    "share_of_wave_time_waiting_on_a_counter": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
    "cu_busy_share": round(c["SQ_BUSY_CU_CYCLES"] / cus / launch_cycles * 4, 3) if c["SQ_BUSY_CU_CYCLES"] / cus / launch_cycles < 0.5 else round(c["SQ_BUSY_CU_CYCLES"] / cus / launch_cycles, 3),
    "l2_read_requests": round(c["TCP_TCC_READ_REQ"]),
    "l2_read_latency_cycles": round(c["TCP_TCC_READ_REQ_LATENCY"] / c["TCP_TCC_READ_REQ"]),
    "vmem_instructions_in_flight_mean": round(c["SQ_INST_LEVEL_VMEM"] / launch_cycles * 4, 1),
}
json.dump({"what": "further counter passes over k_front_stream alone (tools/pmc_more.sh passes a, b, c: rocprofv3 --pmc <counters> -- python3 "
                   "tools/front_probe.py; 256 x 1080p frames per launch, the units near the triangle ROI), means of launches 2-6 "
                   "(Canny 100, 200) and 8-12 (thresholds nothing passes)",
           "units": "SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_BUSY_CU_CYCLES, SQ_INST_LEVEL_VMEM count in units of four cycles (summed over waves / CUs); "
                    "SQ_CYCLES in cycles summed over the 32 shader engines",
           "counters": front, "derived_canny_100_200": derived}, open("profiles/%s_pmc_front_occupancy.json" % tag, "w"), indent=1)

per = collections.OrderedDict()
f = glob.glob(src + '/d/**/*_counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('lanecv::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if not k.startswith("k_") or "probe" in k or "queue" in k: continue
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[k][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, v in agg.items():
    d = list(dur[k].values())
    if sum(d) / len(d) < 2: continue
    per[k] = {"launches": len(d), "mean_us_serialised": round(sum(d) / len(d), 1)}
    per[k].update({cn: round(sum(x) / len(x)) for cn, x in sorted(v.items())})
json.dump({"what": "instruction counts per launch of every kernel of the bench loop (tools/pmc_more.sh pass d: rocprofv3 --pmc ... -- python3 bench.py "
                   "--steps 8 --warmup 4; kernels are serialised under --pmc, so the times are those of a kernel alone on the GPU)",
           "kernels": per}, open("profiles/%s_pmc_all_kernels.json" % tag, "w"), indent=1)
print(json.dumps(derived, indent=1)); print(json.dumps(per, indent=1)[:3000])
