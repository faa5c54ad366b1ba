#!/bin/bash
# experiment batch 1: full-size parity, levels dump, snake deal
mkdir -p gpurun_out/exp1
python -m pytest tests/test_gpu_fullsize.py -x -q -s -m gpu > gpurun_out/exp1/fullsize.log 2>&1
tail -15 gpurun_out/exp1/fullsize.log
python tools/exp/levels_dump.py gpurun_out/exp1/levels_c2.npz > gpurun_out/exp1/levels.log 2>&1; tail -2 gpurun_out/exp1/levels.log
for lib in libsucre_hip.so libsucre_hip_snake.so libsucre_hip.so libsucre_hip_snake.so; do
  for S in 1 2; do
    echo "== $lib in-flight $S"
    SUCRE_HIP_LIB=$PWD/sucre_amd/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --images-in-flight $S 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/image %.2f  ms/launch alone %.4f  timed %.4f  frac %.3f' % (d['ms_per_step'], r['ms_per_launch'], r['timed_region_ms_per_launch'], r['frac']))"
  done
done
