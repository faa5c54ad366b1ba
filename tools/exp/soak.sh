#!/bin/bash
# Soak: three processes fitting the same image at the same time, many repetitions each; every digest of every process
# must be the digest of a run alone on the GPU (tests/concurrency_worker.py).  usage: tools/exp/soak.sh [reps] [T] [fixture]
REPS=${1:-200}; T=${2:-40}
FIX=${3:-relief_96x64_n6}
REF=$(python3 tests/concurrency_worker.py $FIX 3 $T | grep ^DIGESTS | cut -d' ' -f2)
for i in 1 2 3; do python3 tests/concurrency_worker.py $FIX $REPS $T > /tmp/soak_$i.log 2>&1 & done
wait
for i in 1 2 3; do
  grep ^DIGESTS /tmp/soak_$i.log | tr ' ' '\n' | tail -n +2 | sort | uniq -c
done
echo reference $REF
