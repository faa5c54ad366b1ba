#!/bin/bash
# rocprofv3 evidence for every mode of the engine, on the GPU box:
#   gpurun -- bash tools/profile.sh <round tag, e.g. r03> [mode ...]      (default: every mode)
# Per mode: one --kernel-trace --stats run with ONE image at a time (a kernel's duration is then its own: bench.py keeps
# two images in flight by default), and separate --pmc runs (never combined with tracing) for FETCH_SIZE, WRITE_SIZE and
# the SQ counters.  tools/summarize_prof.py condenses them into profiles/<tag>_<mode>_{kernel_stats.csv,traffic.json,
# summary.txt}; raw per-dispatch dumps stay on the box.
set -u
TAG=${1:-r06}; shift || true
MODES=${*:-"jparam jparam_inflight2 closed light light_closed u16mm_4k shared4 jparam_batch32"}
export TMPDIR=/tmp
mode_args() {
  case $1 in
    jparam) echo "" ;;
    jparam_inflight2) echo "" ;;
    closed) echo "--use-closed-form" ;;
    light) echo "--light-model" ;;
    light_closed) echo "--light-model --use-closed-form" ;;
    u16mm_4k) echo "--width 3840 --height 2160 --neighbours 256 --obs-format u16mm" ;;
    shared4) echo "--shared-water --batch-images 4" ;;
    jparam_batch32) echo "--config 1" ;;   # BASELINE config 1 (640x480, 4 neighbours), 32 consecutive images per fit launch
    closed_batch32) echo "--config 1 --use-closed-form" ;;
    jparam_f32plain) echo "--obs-format f32plain" ;;   # the float32 words themselves (7 B/obs): what an image outside every code window pays
    jparam_f32z26) echo "--obs-format f32z26" ;;       # the 26-bit range codes forced on the default scene (A/B against the 24-bit ones)
    jparam_deep) echo "--scene deep" ;;                # ranges 0.7-8 m: outside the 24-bit range codes, the store keeps the float32 words
    jparam_deep_f32z26) echo "--scene deep --obs-format f32z26" ;;   # ... as 26-bit range codes (6.25 B/obs): slower to decode than the words are to read
    *) echo "unknown mode $1" >&2; exit 2 ;;
  esac
}
for MODE in $MODES; do   # an unknown mode stops the script before anything is profiled (exit inside $(...) would not)
  mode_args $MODE > /dev/null || exit 2
done
for MODE in $MODES; do
  OUT=gpurun_out/prof_${TAG}/$MODE
  mkdir -p $OUT
  M=$(mode_args $MODE)
  if [ $MODE = jparam_inflight2 ]; then   # the default command: two images in flight, kernel durations overlap
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
    # bytes per launch in THIS configuration (counter collection serialises the dispatches, so the window is bench.py's own
    # HIP-event union, not the profiler's: what the counters add is that a launch moves the same bytes with a neighbour in flight)
    PMC="--steps 2 --warmup 1 --no-cpu-baseline --num-iter 10 --solo-images 1"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $PMC > $OUT/bench_pmc_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 bench.py $PMC > $OUT/bench_pmc_write.log 2>&1
  else
    STEPS="--steps 2 --warmup 1"; [ $MODE = u16mm_4k ] && STEPS="--steps 1 --warmup 1 --solo-images 1"
    case $MODE in *_batch32) STEPS="--steps 64 --warmup 32" ;; esac
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $STEPS --no-cpu-baseline --images-in-flight 1 $M > $OUT/bench_trace.log 2>&1
    PMC="--steps 1 --warmup 0 --no-cpu-baseline --images-in-flight 1 --num-iter 10 --solo-images 1 $M"
    # (shared4: with --warmup 0 bench.py's set-up pass fits ONE image through the group kernel; --warmup 1 skips it, so that
    # every group_iter_kernel dispatch in the averages walks all four images)
    [ $MODE = shared4 ] && PMC="--steps 1 --warmup 1 --no-cpu-baseline --images-in-flight 1 --num-iter 10 $M"
    case $MODE in *_batch32) PMC="--steps 32 --warmup 32 --no-cpu-baseline --images-in-flight 1 --num-iter 10 --solo-images 1 $M" ;; esac
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $PMC > $OUT/bench_pmc_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 bench.py $PMC > $OUT/bench_pmc_write.log 2>&1
    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py $PMC > $OUT/bench_pmc_sq.log 2>&1
    if [ $MODE = jparam ]; then
      rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $PMC > $OUT/bench_pmc_sq2.log 2>&1
    fi
  fi
  python3 tools/summarize_prof.py gpurun_out/prof_${TAG} $MODE $TAG
  # keep the copy-back small: raw per-dispatch traces and counter dumps stay on the box
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
  grep -h '"metric"' $OUT/bench_trace.log | cut -c1-300
done
ls gpurun_out/prof_${TAG}/summaries
