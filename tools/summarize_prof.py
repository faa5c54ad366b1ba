#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into one text summary for profiles/."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.split('(')[0]
    for pre in ('void sucre::', 'sucre::'):
        name = name.replace(pre, '')
    return name[:70]


for f in glob.glob(f'{out}/trace/**/*kernel_stats.csv', recursive=True):
    print(f'== kernel stats ({f})')
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"{short(r['Name']):70s} calls={r['Calls']:>6s} total_ns={r['TotalDurationNs']:>12s} avg_ns={float(r['AverageNs']):>12.1f} "
              f"min={r['MinNs']:>9s} max={r['MaxNs']:>9s} pct={r['Percentage']}")

for d in ('pmc_sq', 'pmc_sq2', 'pmc_fetch', 'pmc_write'):
    for f in glob.glob(f'{out}/{d}/**/*counter_collection.csv', recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(lambda: defaultdict(int))
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[k][r['Counter_Name']] += 1
        print(f'== {d}: per-dispatch averages')
        for k in acc:
            if not any(s in k for s in ('fit_grad', 'fit_closed', 'light_grad', 'match_kernel', 'update_J', 'gather_kernel')):
                continue
            print(' ', k)
            for c in sorted(acc[k]):
                print(f'      {c:28s} {acc[k][c] / cnt[k][c]:18.1f}   (n={cnt[k][c]})')
