# same-box A/B of the light model: round-2 tree (_r02 worktree) vs the current build
set -u
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --images-in-flight 1 --solo-images 2 --light-model"
for extra in "" "--use-closed-form"; do
  if [ -d _r02 ]; then (cd _r02 && python3 bench.py $ARGS $extra 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r02  $extra ms/image %.2f  ms/iteration alone %.4f' % (d['ms_per_step'], d['roofline']['ms_per_launch']))"); fi
  python3 bench.py $ARGS $extra 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cur  $extra ms/image %.2f  ms/iteration alone %.4f' % (d['ms_per_step'], d['roofline']['ms_per_launch']))"
done
