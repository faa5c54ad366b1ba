#!/bin/bash
# Same-box comparison of this tree with an older round's (kernel durations move +-5 % between boxes, so the committed profiles of
# two rounds are not comparable by themselves):
#   git worktree add /tmp/rNN <round-NN commit> && make -C /tmp/rNN/sucre_amd/csrc && mkdir -p tools/exp/_rNN
#   cp -r /tmp/rNN/{sucre_amd,bench.py,profiles,tools} tools/exp/_rNN/        (git-excluded; gpurun ships it)
#   gpurun -- bash tools/exp/ab_rounds.sh tools/exp/_rNN > profiles/rMM_same_box_vs_rNN.txt
# Per tree and mode: the dominant kernel's rocprofv3 --kernel-trace --stats average, one image at a time, two rounds interleaved.
OLD=$1
export TMPDIR=/tmp
mkdir -p gpurun_out/ab_rounds
for round in 1 2; do
  for tree in . $OLD; do
    for mode in "" "--use-closed-form" "--light-model"; do
      tag=$(echo "$tree$mode$round" | tr -c 'a-zA-Z0-9' '_')
      rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_rounds/$tag -- python3 $tree/bench.py --steps 2 --warmup 1 --no-cpu-baseline --images-in-flight 1 $mode > gpurun_out/ab_rounds/$tag.log 2>&1
      f=$(find gpurun_out/ab_rounds/$tag -name "*kernel_stats.csv" | head -1)
      python3 - "$f" "$tree" "$mode" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'sucre::' in r['Name']]
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
r = rows[0]
name = r['Name'].split('(')[0].replace('void sucre::', '')
print(f"{sys.argv[2]:18s} {sys.argv[3] or '(default)':18s} {name[:40]:40s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
      find gpurun_out/ab_rounds/$tag -name "*kernel_trace.csv" -delete; find gpurun_out/ab_rounds/$tag -name "*.db" -delete
    done
  done
done
