#!/bin/bash
# The kernel-trace part of tools/profile_round.sh alone: tools/trace_round.sh <tag>
set -u
TAG=${1:-r3}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; rm -rf $OUT/trace
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-other-content > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
cd $ROOT
python3 tools/timeline.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) 3 > $OUT/timeline.txt
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
tail -3 $OUT/timeline.txt
