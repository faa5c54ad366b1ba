#!/bin/bash
# times bench.py (solo kernel + overlapped throughput) for every libsucre_hip_*.so variant given
for lib in "$@"; do
  SUCRE_HIP_LIB=$PWD/sucre_amd/$lib python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-26s ms/image %.2f  us/launch alone %.1f  timed %.1f  frac %.3f  match+finalize %.2f ms' % ('$lib', d['ms_per_step'], r['ms_per_launch']*1e3, r['timed_region_ms_per_launch']*1e3, r['frac'], d['roofline_match']['ms']))"
done
