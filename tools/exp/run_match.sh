# parity + result digests + per-kernel times of the match / finalize stage (gpurun: bash tools/exp/run_match.sh <tag>)
set -u
TAG=${1:-m}
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
python tools/exp/digest.py > gpurun_out/$TAG/digest.txt 2>&1
if [ -f gpurun_out/digest_before.txt ]; then diff <(grep DIGEST gpurun_out/digest_before.txt) <(grep DIGEST gpurun_out/$TAG/digest.txt) && echo "DIGESTS EQUAL"; else cat gpurun_out/$TAG/digest.txt; fi
python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --images-in-flight 1 --num-iter 20 --solo-images 2 > gpurun_out/$TAG/bench.log 2>&1
python3 tools/kstats.py gpurun_out/$TAG/trace
grep -o '"roofline_match".*' gpurun_out/$TAG/bench.log | cut -c1-330
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/$TAG -name "*.db" -delete
