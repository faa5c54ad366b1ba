for rep in 1 2; do
for which in product cw5 cdeal cw5deal; do
  if [ $which = product ]; then unset SUCRE_HIP_LIB; else export SUCRE_HIP_LIB=$PWD/sucre_amd/libsucre_hip_$which.so; fi
  for fl in 1 2; do
  python3 bench.py --use-closed-form --no-cpu-baseline --steps 6 --warmup 2 --images-in-flight $fl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$which'.ljust(8), 'in flight $fl:', round(d['value'],2), 'Mpix/s', round(d['config']['ms_per_image'],2), 'ms/img; launch alone', round(r['ms_per_launch']*1e3,1), 'us frac', round(r['frac'],3))"
  done
done
done
