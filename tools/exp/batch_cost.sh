#!/bin/bash
# What a batch iteration of b config-1 images costs alone:  gpurun -- bash tools/exp/batch_cost.sh
for b in 1 2 4 6 8 16 32; do
  python3 bench.py --config 1 --fit-batch $b --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('b =', $b, ' iteration alone', round(r['ms_per_launch']*1e3,1), 'us  frac', round(r['frac'],3), ' whole job', round(d['value'],1), 'Mpix/s')"
done
