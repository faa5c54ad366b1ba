# same-box A/B of two builds of the library: tools/exp/ab_lib.sh <variant> [kernel substrings ...]
# (make -C sucre_amd/csrc VARIANT=<variant> EXTRA=-D... first: ../libsucre_hip_<variant>.so); interleaved, two rounds each
set -u
export TMPDIR=/tmp
VAR=$1; shift
KERNELS=${@:-match_kernel scatter fit_grad}
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --images-in-flight 1 --num-iter 20 --solo-images 2 ${AB_ARGS:-}"   # AB_ARGS: e.g. "--light-model --use-closed-form"
mkdir -p gpurun_out/ab
for round in 1 2; do
  for which in cur $VAR; do
    if [ $which = cur ]; then unset SUCRE_HIP_LIB; else export SUCRE_HIP_LIB=$PWD/sucre_amd/libsucre_hip_$VAR.so; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab/${which}_$round -- python3 bench.py $ARGS > gpurun_out/ab/${which}_$round.log 2>&1
    echo "## $which round $round"; python3 tools/kstats.py gpurun_out/ab/${which}_$round $KERNELS | grep -v '^=='
  done
done
find gpurun_out/ab -name "*kernel_trace.csv" -delete; find gpurun_out/ab -name "*.db" -delete
