#!/bin/bash
# kernel-trace stats only (no PMC): tools/trace_kernels.sh <tag> [bench args]
TAG=${1:-t}; shift || true
OUT=gpurun_out/trace_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $* > $OUT/bench.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/trace/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f))):
        n = r['Name']
        if 'sucre' in n:
            print(f"{n.split('(')[0].replace('void sucre::','')[:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f}")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
tail -1 $OUT/bench.log | cut -c1-300
