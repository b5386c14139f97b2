"""Per-kernel means of the counters in a rocprofv3 --pmc counter_collection.csv (first argument: directory; second: a
part of the kernel names wanted, default k_front_stream; third: 'all' prints every launch instead of the mean)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_counter_collection.csv', recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else 'k_front_stream'
every = len(sys.argv) > 3 and sys.argv[3] == 'all'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if pat in r['Kernel_Name']:
        k = r['Kernel_Name'].replace('lanecv::(anonymous namespace)::', '').replace('void ', '')[:60]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, v in agg.items():
    d = list(dur[k].values())
    print(k, 'launches', len(d), 'mean us %.1f' % (sum(d) / len(d)), [round(x) for x in d] if every else '')
    for c, vals in sorted(v.items()):
        print('   %-26s' % c, [round(x / 1e6, 2) for x in vals] if every else 'mean %.3f M' % (sum(vals) / len(vals) / 1e6))
