#!/bin/bash
# usage: tools/exp/build_variants.sh "name -DFLAG ..." ["name2 ..."]   -> sucre_amd/libsucre_hip_<name>.so (parallel builds)
cd /root/repo/sucre_amd/csrc || exit 1
for v in "$@"; do
  set -- $v; n=$1; shift
  rm -f ../libsucre_hip_$n.so *_$n.o
  make VARIANT=$n EXTRA="$*" -j2 > /tmp/build_$n.log 2>&1 &
done
wait
ls -la /root/repo/sucre_amd/*.so
grep -l " error" /tmp/build_*.log 2>/dev/null
