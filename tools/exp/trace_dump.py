#!/usr/bin/env python3
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rd = csv.DictReader(open(f))
    for r in rd:
        rows.append(r)
print(list(rows[0].keys()))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fits = [i for i, r in enumerate(rows) if 'fit_grad_kernel' in r['Kernel_Name']]
mid = fits[len(fits) // 2]
t0 = int(rows[mid]['Start_Timestamp'])
for r in rows[mid:mid + int(sys.argv[2]) if len(sys.argv) > 2 else mid + 60]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s / 1e3:10.1f} {e / 1e3:10.1f} {(e - s) / 1e3:8.1f} us  q={r.get('Queue_Id')} {r['Kernel_Name'][:60]}")
