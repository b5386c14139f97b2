"""Start / end of the kernels in a rocprofv3 --kernel-trace csv, relative to the first, for the last N dispatches:
shows which kernels run side by side.  usage: trace_overlap.py <dir> [N] [pattern]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-N:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    m = re.search(r'(k_[a-z0-9_]+|__amd[a-zA-Z_]+)', r['Kernel_Name']); name = (m.group(1) if m else r['Kernel_Name'])[:28]
    print('%-28s q%-3s %9.1f %9.1f  %7.1f us' % (name, r.get('Queue_Id', '?'), (int(r['Start_Timestamp']) - t0) / 1e3,
                                                 (int(r['End_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
