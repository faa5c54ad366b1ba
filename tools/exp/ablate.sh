#!/bin/bash
bash tools/exp/variants.sh libsucre_hip_w4r3.so libsucre_hip_noload.so libsucre_hip_nocompute.so libsucre_hip_neither.so
export TMPDIR=/tmp
OUT=gpurun_out/exp2; mkdir -p $OUT
PMCARGS="--steps 1 --warmup 0 --no-cpu-baseline --images-in-flight 1 --num-iter 10 --solo-images 1"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py $PMCARGS > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $PMCARGS > $OUT/bench_pmc_sq2.log 2>&1
python3 tools/summarize_prof.py $OUT 2>&1 | grep -A12 "fit_grad"
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
