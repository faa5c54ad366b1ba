#!/bin/bash
# same-box A/B of the product against other builds of the library (sucre_amd/libsucre_hip_<name>.so), one process each, interleaved:
#   gpurun -- bash tools/exp/ab_vs.sh "<bench args>" name [name ...]
ARGS=$1; shift
for rep in 1 2; do
for which in product "$@"; do
  if [ $which = product ]; then unset SUCRE_HIP_LIB; else export SUCRE_HIP_LIB=$PWD/sucre_amd/libsucre_hip_$which.so; fi
  python3 bench.py --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$which'.ljust(10), round(d['value'],2), 'Mpix/s', round(d['config']['ms_per_image'],2), 'ms/img; launch alone', round(r['ms_per_launch']*1e3,1), 'us frac', round(r['frac'],3), 'match ms', round(d.get('roofline_match',{}).get('ms') or 0,3))"
done
done
