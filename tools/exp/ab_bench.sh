# same-box A/B of the whole bench (two images in flight) between the product library and a variant build:
#   tools/exp/ab_bench.sh <variant> [bench.py flags]      (interleaved, three rounds each)
set -u
VAR=$1; shift
for round in 1 2 3; do
  for which in cur $VAR; do
    if [ $which = cur ]; then unset SUCRE_HIP_LIB; else export SUCRE_HIP_LIB=$PWD/sucre_amd/libsucre_hip_$VAR.so; fi
    python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$which round $round: %.2f Mpix/s  %.3f ms/image  launch alone %.1f us  timed-region %.1f us' % (d['value'], d['ms_per_step'], r['ms_per_launch']*1e3, r['timed_region_ms_per_launch']*1e3))"
  done
done
