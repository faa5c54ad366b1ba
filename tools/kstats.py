#!/usr/bin/env python3
"""Prints this library's rows of a rocprofv3 `--kernel-trace --stats` output directory (kernel_stats.csv), in time order.
usage: tools/kstats.py <dir> [substring ...]"""
import csv
import glob
import sys

for f in sorted(glob.glob(f'{sys.argv[1]}/**/*kernel_stats.csv', recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if 'sucre::' in r['Name']]
    if len(sys.argv) > 2:
        rows = [r for r in rows if any(s in r['Name'] for s in sys.argv[2:])]
    print('==', f)
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
        name = r['Name'].split('(')[0].replace('void sucre::', '').replace('sucre::', '')
        print(f"{name[:44]:44s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs']) / 1e3:9.1f} min_us={float(r['MinNs']) / 1e3:9.1f} "
              f"max_us={float(r['MaxNs']) / 1e3:9.1f}")
