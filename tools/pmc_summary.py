"""Per-kernel means of the counters in a rocprofv3 --pmc counter_collection.csv (first argument: directory)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_counter_collection.csv', recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else 'k_front_stream'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if pat in r['Kernel_Name']:
        k = r['Kernel_Name'][:80]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, v in agg.items():
    print(k, 'durations us:', [round(d) for d in dur[k].values()])
    for c, vals in sorted(v.items()):
        print('   %-22s' % c, [round(x / 1e6, 1) for x in vals])
