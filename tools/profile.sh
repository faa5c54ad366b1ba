#!/bin/bash
# Profiles bench.py on the GPU box: kernel-trace stats + separate PMC passes (never combined, see task notes).
# usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# one image at a time, so that a kernel's duration is its own (bench.py overlaps two images by default)
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --images-in-flight 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
# and the default command (two images in flight: kernel durations overlap)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_inflight2 -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline $* > $OUT/bench_trace_inflight2.log 2>&1
PMCARGS="--steps 1 --warmup 0 --no-cpu-baseline --images-in-flight 1 --num-iter 10 $*"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py $PMCARGS > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $PMCARGS > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 bench.py $PMCARGS > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $PMCARGS > $OUT/bench_pmc_sq2.log 2>&1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
# keep the copy-back small: raw per-dispatch traces and counter dumps stay on the box
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/summary.txt
