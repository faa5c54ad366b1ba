# bench.py in every mode quoted in DESIGN.md section 6 / README.md (default settings: two images in flight)
set -u
mkdir -p gpurun_out/modes
run() { name=$1; shift; python3 bench.py --no-cpu-baseline "$@" > gpurun_out/modes/$name.json 2> gpurun_out/modes/$name.err; python3 - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.loads(open(f'gpurun_out/modes/{name}.json').read().strip().splitlines()[-1])
    r = d['roofline']
    print(f"{name:14s} {d['value']:8.2f} Mpix/s  {d['config']['ms_per_image']:8.2f} ms/image  launch alone {r['ms_per_launch'] * 1e3:7.1f} us  "
          f"frac {r['frac']:.3f}  timed-region frac {r['timed_region_frac']:.3f}  n_obs {d['config']['n_obs']}  match+finalize "
          f"{d.get('roofline_match', {}).get('ms', float('nan')):.2f} ms")
except Exception as e:
    print(name, 'FAILED', repr(e))
PY
}
run default
run inflight1 --images-in-flight 1
run closed --use-closed-form
run u16mm --obs-format u16mm
run light --light-model
run light_closed --light-model --use-closed-form
run config3 --batch-images 32 --steps 2 --warmup 1
run shared4 --shared-water --batch-images 4 --steps 4 --warmup 1
run shared4_closed --shared-water --batch-images 4 --use-closed-form --steps 4 --warmup 1
run config5 --width 3840 --height 2160 --neighbours 256 --obs-format u16mm --steps 3 --warmup 1 --solo-images 1
run numiter10 --num-iter 10 --steps 20 --warmup 3
