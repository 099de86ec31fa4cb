#!/bin/bash
# Kernel trace of a short bench run and its timeline.  Run on the GPU box from the repository root.
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/tq
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/trace.err || exit 1
cd $ROOT
python3 tools/timeline.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) 3 > $OUT/timeline.txt
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
cat $OUT/timeline.txt
