#!/bin/bash
# Every dispatch of one serialized 1080p clip with its duration (kernel trace).  Run on the GPU box from the repository root.
R=$PWD; export TMPDIR=/tmp
rm -rf gpurun_out/st
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/st -- python3 $R/tools/serial_trace.py run ${1:-natural} > /dev/null 2>&1) || exit 1
python3 tools/serial_trace.py show $(ls gpurun_out/st/*/*kernel_trace.csv | head -1) > gpurun_out/serial_trace.txt; rm -rf gpurun_out/st
cat gpurun_out/serial_trace.txt
