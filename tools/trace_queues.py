"""Per queue: which kernels ran, how many, mean duration, and the share of the traced interval the queue was busy.
usage: trace_queues.py <dir> [first_fraction last_fraction]  (fractions of the dispatch list, default 0.3 0.8)"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
rows = [r for r in csv.DictReader(open(f)) if 'queue_' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[int(len(rows) * lo):int(len(rows) * hi)]
t0, t1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
print('window %.2f ms, %d dispatches' % ((t1 - t0) / 1e6, len(rows)))
q = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    m = re.search(r'(k_[a-z0-9_]+)', r['Kernel_Name'])
    q[r['Queue_Id']][m.group(1) if m else '?'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for qi, ks in sorted(q.items()):
    busy = sum(sum(v) for v in ks.values())
    print('queue %s busy %.0f%%' % (qi, 100 * busy * 1e3 / (t1 - t0)))
    for k, v in ks.items():
        print('    %-22s n=%3d mean %7.1f us  sum %8.1f' % (k, len(v), sum(v) / len(v), sum(v)))
