#!/bin/bash
# serialized trace of one clip per value of SIFT_TEXTURE_NMSDBG (timing experiment: results are wrong for non-zero values)
R=$PWD; export TMPDIR=/tmp
for v in "$@"; do
  export SIFT_TEXTURE_NMSDBG=$v
  rm -rf gpurun_out/st
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/st -- python3 $R/tools/serial_trace.py run > /dev/null 2>&1) || exit 1
  echo "== NMSDBG=$v"; python3 tools/serial_trace.py show $(ls gpurun_out/st/*/*kernel_trace.csv | head -1) | grep -i "nms\|stencil"; rm -rf gpurun_out/st
done
