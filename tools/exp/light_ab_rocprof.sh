# same-box A/B of the light kernels under rocprofv3 (kernel-only durations): the product against other builds of the library
#   gpurun -- bash tools/exp/light_ab_rocprof.sh name [name ...]     (sucre_amd/libsucre_hip_<name>.so)
export TMPDIR=/tmp
for rep in 1 2; do
for v in product "$@"; do
  if [ $v = product ]; then unset SUCRE_HIP_LIB; else export SUCRE_HIP_LIB=$PWD/sucre_amd/libsucre_hip_$v.so; fi
  for m in "" "--use-closed-form"; do
  rm -rf /tmp/lp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --images-in-flight 1 --light-model $m > /tmp/lp.log 2>&1
  python3 - "$v" "$m" <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/lp/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'light_grad_kernel<' in r['Name'] and int(r['Calls']) > 100 or 'light_tail' in r['Name']:
            print(sys.argv[1].ljust(8), sys.argv[2].ljust(18), r['Name'].split('(')[0].replace('void sucre::','')[:48].ljust(48), 'calls', r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1))
PY
  done
done
done
