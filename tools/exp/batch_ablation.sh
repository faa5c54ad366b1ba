#!/bin/bash
# Where the time of an image inside a batch launch goes (BASELINE config 1, 16 images per launch), same box, one process each:
#   gpurun -- bash tools/exp/batch_ablation.sh        (build the variants first: see below)
# for v in 1 2; do make -C sucre_amd/csrc VARIANT=batch$v EXTRA=-DSUCRE_EXP_BATCH=$v; done
set -u
ARGS="--config 1 --steps 32 --warmup 16 --fit-batch 16 --images-in-flight 1 --no-cpu-baseline --solo-images 4"
for which in product batch1 batch2 product; do
  if [ $which = product ]; then unset SUCRE_HIP_LIB; else export SUCRE_HIP_LIB=$PWD/sucre_amd/libsucre_hip_$which.so; fi
  python3 bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$which'.ljust(8), 'launch of 16 images', round(r['ms_per_launch']*1e3,1), 'us =', round(r['ms_per_launch']*1e3/16,2), 'us per image')"
done
