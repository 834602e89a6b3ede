#!/bin/bash
# kernel statistics of the two workloads beside the headline that bench.py reports: the field of evolved outlines
# (profiles/evolved_step.py) and the B1 clip batch (profiles/clip_batch.py).  usage: bash profiles/collect_other.sh r04
set -e
tag=${1:-r04}; root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
for w in evolved_step clip_batch; do
  rm -rf $out/${tag}_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_$w -- python3 $root/profiles/$w.py > $out/${tag}_$w.txt 2> $out/${tag}_$w.err
  cp $(ls $out/${tag}_$w/*/*kernel_stats.csv | head -1) $out/${tag}_${w}_kernel_stats.csv
  rm -rf $out/${tag}_$w
done
