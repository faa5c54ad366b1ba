#!/bin/bash
# same-box A/B of library builds on the headline workload: tools/exp/ab_solo.sh <variant> [<variant> ...]  ("product" = the shipped build)
# (interleaved, two rounds; prints bench.py's solo-kernel time, the timed-region time and the headline value)
for round in 1 2; do
  for v in "$@"; do
    if [ $v = product ]; then unset SUCRE_HIP_LIB; else export SUCRE_HIP_LIB=$PWD/sucre_amd/libsucre_hip_$v.so; fi
    python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline ${AB_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s round $round: ms/image %.2f  us/launch alone %.1f  in flight %.1f  frac %.3f  value %.1f Mpix/s  match+finalize %.2f ms' % ('$v', d['ms_per_step'], r['ms_per_launch']*1e3, r['timed_region_ms_per_launch']*1e3, r['frac'], d['value'], d.get('roofline_match',{}).get('ms', float('nan'))))"
  done
done
