#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile.sh for one mode into the three files that are committed under
profiles/:  <tag>_<mode>_kernel_stats.csv (this library's kernels only), <tag>_<mode>_summary.txt (stats + per-dispatch
PMC averages) and <tag>_<mode>_traffic.json (HBM bytes per launch of the mode's dominant kernel and of the match /
finalize stage, with the rocprofv3 average duration of the same kernel -- what bench.py reads for `roofline.traffic` and
what a reader needs to recompute `roofline.frac`).
usage: summarize_prof.py <prof dir> <mode> <tag>"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

root, mode, tag = Path(sys.argv[1]), sys.argv[2], sys.argv[3]
out = root / mode
dest = root / 'summaries'
dest.mkdir(exist_ok=True)
DOMINANT = {'jparam': 'fit_grad_kernel<true, 0>', 'jparam_inflight2': 'fit_grad_kernel<true, 0>', 'closed': 'fit_closed_kernel<true, 0, false>',
            'light': 'light_grad_kernel<false, false', 'light_closed': 'light_grad_kernel<true, false', 'u16mm_4k': 'fit_grad_kernel<true, 1>',
            'shared4': 'group_iter_kernel<0, 0>', 'jparam_batch32': 'batch_iter_kernel<0, 0>', 'closed_batch32': 'batch_iter_kernel<1, 0>',
            'jparam_f32plain': 'fit_grad_kernel<true, 0>', 'jparam_f32z26': 'fit_grad_kernel<true, 0>', 'jparam_deep': 'fit_grad_kernel<true, 0>', 'jparam_deep_f32z26': 'fit_grad_kernel<true, 0>'}
MATCH_STAGE = ('match_kernel', 'view_partial_kernel', 'view_total_kernel', 'pixel_count_kernel', 'bin_scan_kernel',
               'permute_kernel', 'strip_table_kernel', 'strip_levels_kernel', 'tile_offset_kernel', 'strip_offset_kernel',
               'scatter_kernel', 'tail_bits_clear_kernel', 'plan_kernel')


def short(name):
    return name.split('(')[0].replace('void sucre::', '').replace('sucre::', '')


lines = []
stats = {}
for f in glob.glob(f'{out}/trace/**/*kernel_stats.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'sucre::' in r['Name']]
    with open(dest / f'{tag}_{mode}_kernel_stats.csv', 'w', newline='') as g:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys()) if rows else ['Name'])
        w.writeheader()
        for r in rows:
            w.writerow(r)
    lines.append(f'== kernel stats, mode {mode} (rocprofv3 --kernel-trace --stats; this library\'s kernels)')
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
        stats[short(r['Name'])] = r
        lines.append(f"{short(r['Name'])[:60]:60s} calls={r['Calls']:>6s} avg_ns={float(r['AverageNs']):>12.1f} min={r['MinNs']:>9s} "
                     f"max={r['MaxNs']:>9s}")

pmc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))   # kernel -> counter -> [sum, n]
for d in ('pmc_sq', 'pmc_sq2', 'pmc_fetch', 'pmc_write'):
    for f in glob.glob(f'{out}/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'sucre::' not in r['Kernel_Name']:
                continue
            e = pmc[short(r['Kernel_Name'])][r['Counter_Name']]
            e[0] += float(r['Counter_Value']); e[1] += 1
if pmc:
    lines.append('== PMC, per-dispatch averages (separate --pmc passes)')
    for k in sorted(pmc):
        lines.append('  ' + k)
        for c in sorted(pmc[k]):
            lines.append(f'      {c:28s} {pmc[k][c][0] / pmc[k][c][1]:18.1f}   (n={pmc[k][c][1]})')
(dest / f'{tag}_{mode}_summary.txt').write_text('\n'.join(lines) + '\n')


def avg(kernel_sub, counter):
    for k, cs in pmc.items():
        if k.startswith(kernel_sub) and counter in cs:
            return cs[counter][0] / cs[counter][1]
    return None


def hbm_bytes(kernel_sub):
    """FETCH_SIZE is in 1024-byte units and, on gfx950, tallies the 128-B requests of wide streaming reads at 64 B: doubled
    (MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is exact."""
    f, w = avg(kernel_sub, 'FETCH_SIZE'), avg(kernel_sub, 'WRITE_SIZE')
    return None if f is None or w is None else int(round((2 * f + w) * 1024)), f, w


kernel = DOMINANT[mode]
rec = {'mode': mode, 'kernel': kernel}
b = hbm_bytes(kernel)
if b[0] is not None:
    rec.update(fetch_size_kb=b[1], write_size_kb=b[2], hbm_bytes_per_launch=b[0])
for k, r in stats.items():
    if k.startswith(kernel):
        rec['rocprofv3_avg_ns'] = float(r['AverageNs'])
        rec['rocprofv3_calls'] = int(r['Calls'])
stage = {}
for k in pmc:
    if any(k.startswith(m) for m in MATCH_STAGE):
        bb = hbm_bytes(k)
        if bb[0] is not None:
            stage[k] = {'hbm_bytes_per_launch': bb[0], 'avg_ns': float(stats[k]['AverageNs']) if k in stats else None}
if stage:
    rec['match_stage'] = stage
    rec['match_stage_hbm_bytes'] = sum(v['hbm_bytes_per_launch'] for v in stage.values())
    if all(v['avg_ns'] is not None for v in stage.values()):
        rec['match_stage_kernel_ns'] = sum(v['avg_ns'] for v in stage.values())
try:   # the workload the numbers belong to: bench.py prints it in its JSON line
    line = [ln for ln in open(out / 'bench_trace.log') if ln.startswith('{')][-1]
    j = json.loads(line)
    rec['n_obs'] = j['config']['n_obs']
    rec['workload'] = j['config']['workload']
    rec['bench_ms_per_launch_solo'] = j['roofline']['ms_per_launch']
    rec['algorithmic_bytes_per_launch'] = j['roofline']['algorithmic_bytes_per_launch']
except (OSError, IndexError, KeyError, ValueError) as e:
    rec['workload_error'] = repr(e)
rec['_comment'] = ('HBM traffic from rocprofv3 PMC passes (tools/profile.sh; separate --pmc runs, never combined with tracing): '
                   '2 x FETCH_SIZE + WRITE_SIZE, in bytes per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies the '
                   '128-B requests of wide streaming reads at 64 B).  rocprofv3_avg_ns: --kernel-trace --stats, one image at a time.')
(dest / f'{tag}_{mode}_traffic.json').write_text(json.dumps(rec, indent=1) + '\n')
print(f'== {mode}:', json.dumps({k: v for k, v in rec.items() if k not in ('_comment', 'match_stage')}))
