#!/bin/bash
# GPU tests with the suppression forced onto the flooded set, a serialized trace of one clip, then the bench per mode.
set -u
export TMPDIR=/tmp
SIFT_TEXTURE_CANNY=sparse timeout -k 10 400 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputests_sparse.log 2>&1
rc=$?; tail -n 4 gpurun_out/gputests_sparse.log; [ $rc -ne 0 ] && exit $rc
R=$PWD
rm -rf gpurun_out/st
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/st -- python3 $R/tools/serial_trace.py run > /dev/null 2>&1) || exit 1
python3 tools/serial_trace.py show $(ls gpurun_out/st/*/*kernel_trace.csv | head -1) > gpurun_out/serial_trace.txt; rm -rf gpurun_out/st
cat gpurun_out/serial_trace.txt
bash tools/envab.sh SIFT_TEXTURE_CANNY dense auto dense auto
