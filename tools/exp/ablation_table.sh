#!/bin/bash
# Four-row ablation table per fit kernel (VERDICT r03 task 6i): the product build, the arithmetic compiled out
# (SUCRE_EXP_NOCOMPUTE: data still loaded and touched), the loads compiled out (SUCRE_EXP_NOLOAD: arithmetic on whatever
# LDS / constants hold), both compiled out (the launch's skeleton: bookkeeping, waits, reductions, tail).
#   build here:  bash tools/exp/build_variants.sh "nocompute -DSUCRE_EXP_NOCOMPUTE" "noload -DSUCRE_EXP_NOLOAD" "neither -DSUCRE_EXP_NOCOMPUTE -DSUCRE_EXP_NOLOAD"
#   on the box:  gpurun -- bash tools/exp/ablation_table.sh
# Times are bench.py's own roofline block: HIP events around the launches of images restored strictly one at a time.
set -u
OUT=gpurun_out/ablation; mkdir -p $OUT
TABLE=$OUT/table.txt; : > $TABLE
for MODE in jparam closed light light_closed; do
  case $MODE in
    jparam) M="" ;; closed) M="--use-closed-form" ;; light) M="--light-model" ;; light_closed) M="--light-model --use-closed-form" ;;
  esac
  echo "## $MODE  (bench.py --images-in-flight 1 --solo-images 3 $M)" >> $TABLE
  for LIB in product nocompute noload neither; do
    if [ $LIB = product ]; then unset SUCRE_HIP_LIB; else export SUCRE_HIP_LIB=$PWD/sucre_amd/libsucre_hip_$LIB.so; fi
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --images-in-flight 1 --solo-images 3 $M 2> $OUT/${MODE}_$LIB.err | tail -1 > $OUT/${MODE}_$LIB.json
    python3 - $OUT/${MODE}_$LIB.json $LIB >> $TABLE <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); r = d['roofline']
    print('%-10s %8.1f us per launch   (algorithmic %.1f MB -> %.3f of 8 TB/s)' % (sys.argv[2], r['ms_per_launch'] * 1e3, r['algorithmic_bytes_per_launch'] / 1e6, r['frac']))
except Exception as e:
    print('%-10s failed: %r' % (sys.argv[2], e))
PY
  done
done
cat $TABLE
