#!/bin/bash
# One measurement session on the GPU box: kernel trace + stats of the bench command, then FETCH_SIZE / WRITE_SIZE in
# separate --pmc passes (MI355X_MICROARCH.md), all under gpurun_out/<tag>_*; profiles/make_summaries.py turns them into
# the committed summaries.  usage: bash profiles/collect.sh r04
set -e
tag=${1:-r04}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
rm -rf $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $root/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-clip-line > $out/${tag}_stats_bench.json 2> $out/${tag}_stats.err
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_fetch -- python3 $root/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-clip-line > /dev/null 2> $out/${tag}_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_write -- python3 $root/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-clip-line > /dev/null 2> $out/${tag}_write.err
echo "write pass done"
python3 profiles/make_summaries.py $tag $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write
cp profiles/${tag}_* $out/ 2>/dev/null || true
python3 profiles/timed_region.py $out/${tag}_stats 100 $out/${tag}_kernel_stats_timed_region.csv > /dev/null
# keep the merged-back output small: the raw traces are not needed once summarised
rm -rf $out/${tag}_fetch $out/${tag}_write
find $out/${tag}_stats -name "*kernel_trace.csv" -size +40M -delete
