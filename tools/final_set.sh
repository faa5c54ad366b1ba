#!/bin/bash
# The round's evidence on ONE box, the final build, one gpurun call:
#   gpurun --timeout 5000 -- bash tools/final_set.sh r06
# every mode of tools/profile.sh (rocprofv3 kernel trace + PMC passes -> gpurun_out/prof_<tag>/summaries) and every bench.py line
# quoted in DESIGN.md section 6 (-> gpurun_out/final_<tag>/bench_<name>.json), then one table of both
# (gpurun_out/final_<tag>/<tag>_final_build_one_box.txt).  Copy the summaries, the bench lines and the table into profiles/.
set -u
TAG=${1:-r06}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
# (SKIP_PROFILE=1: the bench lines only -- e.g. once more after the profiles have been committed, so that every line's
# `profile_frac` / `traffic` come from this round's files)
[ -n "${SKIP_PROFILE:-}" ] || bash tools/profile.sh $TAG jparam jparam_inflight2 closed light light_closed u16mm_4k shared4 jparam_batch32 closed_batch32 \
     jparam_f32plain jparam_f32z26 jparam_deep jparam_deep_f32z26 > $OUT/profile.log 2>&1
run() { name=$1; shift; python3 bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
run default
run config1 --config 1 --no-cpu-baseline
run config1_closed --config 1 --use-closed-form --no-cpu-baseline
run config1_one_launch_per_image --config 1 --fit-batch 1 --steps 20 --warmup 4 --no-cpu-baseline
run config2 --config 2 --no-cpu-baseline
run config3 --config 3 --steps 2 --warmup 1 --no-cpu-baseline
run config4 --config 4 --steps 2 --warmup 1 --no-cpu-baseline
run config5 --config 5 --steps 3 --warmup 1 --solo-images 1 --no-cpu-baseline
run closed --use-closed-form --no-cpu-baseline
run light --light-model --no-cpu-baseline
run light_closed --light-model --use-closed-form --no-cpu-baseline
run f32plain --obs-format f32plain --no-cpu-baseline
run f32z26 --obs-format f32z26 --no-cpu-baseline
run deep --scene deep --no-cpu-baseline
run deep_f32z26 --scene deep --obs-format f32z26 --no-cpu-baseline
python3 - $TAG <<'PY' | tee $OUT/${TAG}_final_build_one_box.txt
import glob, json, sys
tag = sys.argv[1]
print(f'One box, the final build of round {tag[1:]} (every mode profiled and every bench line run in ONE gpurun call: tools/final_set.sh).\n')
print(f"{'mode':20s} {'dominant kernel':38s} {'us/launch (rocprofv3)':>22s} {'PMC MB/launch':>14s} {'algorithmic MB':>15s} {'algorithmic / (us * 8 TB/s)':>28s}")
for f in sorted(glob.glob(f'gpurun_out/prof_{tag}/summaries/{tag}_*_traffic.json')):
    r = json.load(open(f))
    us = r.get('rocprofv3_avg_ns', float('nan')) / 1e3
    a = r.get('algorithmic_bytes_per_launch', float('nan'))
    print(f"{r['mode']:20s} {r['kernel'][:38]:38s} {us:22.1f} {r.get('hbm_bytes_per_launch', float('nan')) / 1e6:14.1f} {a / 1e6:15.1f} {a / (us * 1e-6) / 8e12:28.3f}")
print(f"\n{'bench line':32s} {'Mpix/s':>8s} {'prepacked':>10s} {'one alone':>10s} {'ms/step':>9s} {'frac alone':>11s} {'timed region':>13s} {'us/launch':>10s}  store")
for f in sorted(glob.glob(f'gpurun_out/final_{tag}/bench_*.json')):
    try:
        d = json.loads([ln for ln in open(f) if ln.startswith('{')][-1])
    except Exception as e:
        print(f.split('bench_')[-1][:-5], 'FAILED', repr(e)); continue
    r, c = d['roofline'], d['config']
    fmt = lambda x: f'{x:10.2f}' if x is not None else f"{'-':>10s}"
    print(f"{f.split('bench_')[-1][:-5]:32s} {d['value']:8.2f} {fmt(c.get('value_views_prepacked'))} {fmt(c.get('value_one_image_alone'))} {d['ms_per_step']:9.2f} "
          f"{r['frac']:11.3f} {r['timed_region_frac']:13.3f} {r['ms_per_launch'] * 1e3:10.1f}  {r['store_format']}")
PY
