#!/usr/bin/env python3
"""Union of kernel intervals from a rocprofv3 kernel trace: how busy was the GPU between the first and the last fit launch?"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
fits = [r for r in rows if 'fit_grad_kernel' in r[2]]
t0, t1 = fits[0][0], max(r[1] for r in fits)
busy = 0; cur_s = cur_e = None
fit_busy = 0
for s, e, n in rows:
    if e < t0 or s > t1: continue
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
cur_s = cur_e = None
for s, e, n in fits:
    if cur_e is None or s > cur_e:
        if cur_e is not None: fit_busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
fit_busy += cur_e - cur_s
print(f'window {1e-9 * (t1 - t0):.3f} s, any kernel running {busy / (t1 - t0):.3f}, a fit kernel running {fit_busy / (t1 - t0):.3f}, '
      f'fit launches {len(fits)}, sum of fit durations {1e-9 * sum(e - s for s, e, _ in fits):.3f} s')
# gaps > 1 ms with no kernel at all
gaps = []
cur_e = None
for s, e, n in rows:
    if e < t0 or s > t1: continue
    if cur_e is not None and s - cur_e > 1_000_000: gaps.append((cur_e - t0, s - cur_e))
    cur_e = e if cur_e is None else max(cur_e, e)
print('idle gaps > 1 ms:', len(gaps), 'total', sum(g for _, g in gaps) * 1e-9, 's; first:', [(round(a * 1e-9, 3), round(g * 1e-6, 1)) for a, g in gaps[:12]])
