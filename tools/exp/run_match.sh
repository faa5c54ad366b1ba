set -u
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_api.py -x -q 2>&1 | tail -3
mkdir -p gpurun_out/m1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/m1/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --images-in-flight 1 --num-iter 20 --solo-images 2 > gpurun_out/m1/bench.log 2>&1
python3 tools/kstats.py gpurun_out/m1/trace
grep -o '"roofline_match".*' gpurun_out/m1/bench.log | cut -c1-400
find gpurun_out/m1 -name "*kernel_trace.csv" -delete; find gpurun_out/m1 -name "*.db" -delete
