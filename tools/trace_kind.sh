#!/bin/bash
# Kernel trace of a short 1080p bench on one content class: tools/trace_kind.sh <kind>  (run on the GPU box from the repository root)
R=$PWD; K=${1:-ramp}
O=$R/gpurun_out/trk; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --kind $K > $O/b.json 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
cd $R; head -12 $(ls $O/t/*/*kernel_stats.csv | head -1) | cut -c1-200
