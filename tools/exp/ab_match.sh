# same-box A/B of the match stage: round-2 tree (_r02 worktree), current build
set -u
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --images-in-flight 1 --num-iter 20 --solo-images 2"
mkdir -p gpurun_out/ab
if [ -d _r02 ]; then (cd _r02 && rocprofv3 --kernel-trace --stats --output-format csv -d ../gpurun_out/ab/r02 -- python3 bench.py $ARGS > ../gpurun_out/ab/r02.log 2>&1); python3 tools/kstats.py gpurun_out/ab/r02 match_kernel scatter fit_grad pixel_count; fi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab/cur -- python3 bench.py $ARGS > gpurun_out/ab/cur.log 2>&1
python3 tools/kstats.py gpurun_out/ab/cur match_kernel scatter fit_grad tile_cull
find gpurun_out/ab -name "*kernel_trace.csv" -delete; find gpurun_out/ab -name "*.db" -delete
