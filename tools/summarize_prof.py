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


for f in glob.glob(f'{out}/trace/**/*kernel_stats.csv', recursive=True) + glob.glob(f'{out}/trace_inflight2/**/*kernel_stats.csv', recursive=True):
    print(f'== kernel stats ({f})')
    rows = list(csv.DictReader(open(f)))
    ours = ('sucre::',)
    for i, r in enumerate(rows):
        if i >= 6 and not any(o in r['Name'] for o in ours):
            continue   # beyond the top rows only this library's kernels (the rest is torch building the synthetic scene)
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
            if not any(s in k for s in ('fit_grad', 'fit_closed', 'light_grad', 'match_kernel', 'update_J', 'scatter_kernel', 'plan_kernel', 'group_iter', 'select_')):
                continue
            print(' ', k)
            for c in sorted(acc[k]):
                print(f'      {c:28s} {acc[k][c] / cnt[k][c]:18.1f}   (n={cnt[k][c]})')


# HBM bytes per launch of the dominant kernel from the FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md: FETCH_SIZE
# is in KiB-like units of 1024 B and, on gfx950, tallies the 128-B requests of wide streaming reads at 64 B -> x2)
import json  # noqa: E402


def pmc_avg(d, counter, kernel):
    for f in glob.glob(f'{out}/{d}/**/*counter_collection.csv', recursive=True):
        tot, n = 0.0, 0
        for r in csv.DictReader(open(f)):
            if kernel in r['Kernel_Name'] and r['Counter_Name'] == counter:
                tot += float(r['Counter_Value']); n += 1
        if n:
            return tot / n
    return None


kernel = sys.argv[2] if len(sys.argv) > 2 else 'fit_grad_kernel'
fetch, write = pmc_avg('pmc_fetch', 'FETCH_SIZE', kernel), pmc_avg('pmc_write', 'WRITE_SIZE', kernel)
if fetch is not None and write is not None:
    rec = {'kernel': kernel, 'fetch_size_kb': fetch, 'write_size_kb': write,
           'hbm_bytes_per_launch': int(round((2 * fetch + write) * 1024))}
    for f in glob.glob(f'{out}/trace/**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r['Name']:
                rec['rocprofv3_avg_ns'] = float(r['AverageNs'])
                break
    import re
    try:   # the workload the numbers belong to: bench.py prints n_obs in its JSON line
        rec['n_obs'] = int(re.search(r'"n_obs": (\d+)', open(f'{out}/bench_trace.log').read()).group(1))
    except (OSError, AttributeError):
        pass
    rec['_comment'] = ('HBM traffic of the dominant kernel from rocprofv3 PMC passes (tools/profile.sh; separate --pmc runs, never '
                       'combined with tracing). FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests of wide '
                       'streaming reads at 64 B); WRITE_SIZE is exact. Workload: bench.py default, --images-in-flight 1.')
    json.dump(rec, open(f'{out}/traffic.json', 'w'), indent=1)
    print('== traffic', json.dumps(rec))
