# same-box A/B of closed-form mode: round-2 tree (_r02 worktree) vs the current build
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --images-in-flight 1 --solo-images 3 --use-closed-form"
if [ -d _r02 ]; then (cd _r02 && python3 bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r02 closed ms/image %.2f  ms/iteration alone %.4f' % (d['ms_per_step'], d['roofline']['ms_per_launch']))"); fi
python3 bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cur closed ms/image %.2f  ms/iteration alone %.4f' % (d['ms_per_step'], d['roofline']['ms_per_launch']))"
