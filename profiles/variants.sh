#!/bin/bash
# bench.py's headline numbers for several builds of the library (make -C subzero_amd/csrc team_variant ...):
# usage: bash profiles/variants.sh <out file> <name> [<name> ...]   ("base" = the product build)
out=$1; shift
for v in "$@"; do
  lib=subzero_amd/csrc/libsubzero_hip_$v.so
  [ "$v" = base ] && lib=subzero_amd/csrc/libsubzero_hip.so
  for rep in 1 2; do
    SUBZERO_HIP_LIB=$PWD/$lib python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-clip-line 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-10s ms/step %.4f device %.4f team kernel %.4f ms  pairs/launch %.0f' % ('$v', d['ms_per_step'], d['device_ms_per_step'], r['kernel_ms'], r.get('pairs_per_launch', 0)))" >> $out
  done
done
cat $out
