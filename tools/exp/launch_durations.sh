# every fit launch's duration in order (rocprofv3 --kernel-trace, one image at a time): where the spread between the fastest and the
# average launch comes from.   gpurun -- bash tools/exp/launch_durations.sh [bench args]
export TMPDIR=/tmp
rm -rf /tmp/ld; rocprofv3 --kernel-trace --output-format csv -d /tmp/ld -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --images-in-flight 1 "$@" > /tmp/ld.log 2>&1
python3 - <<'PY'
import csv, glob
import numpy as np
rows = []
for f in glob.glob('/tmp/ld/**/*kernel_trace.csv', recursive=True):
    rows += [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fit = [(int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows if 'fit_grad_kernel' in r['Kernel_Name']]
d = np.array([x[1] for x in fit]) / 1e3
t = np.array([x[0] for x in fit])
gap = np.diff(t) / 1e3 - d[:-1]
print('launches', len(d), 'avg', d.mean().round(1), 'min', d.min().round(1), 'p10', np.percentile(d, 10).round(1), 'median', np.median(d).round(1), 'p90', np.percentile(d, 90).round(1), 'max', d.max().round(1))
n = len(d) // 200
for i in range(n):
    seg = d[i * 200:(i + 1) * 200]
    print(f'image {i}: first 5', seg[:5].round(1), 'launches 5-50 avg', seg[5:50].mean().round(1), '50-200 avg', seg[50:].mean().round(1), 'min', seg.min().round(1))
g = gap[(gap > 0) & (gap < 50)]
print('gap between consecutive fit launches (us): median', np.median(g).round(2), 'mean', g.mean().round(2))
print('by position within an image (avg over images), every 20th:', np.array([d[k::200].mean() for k in range(0, 200, 20)]).round(1))
PY
