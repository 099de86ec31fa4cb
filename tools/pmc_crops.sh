#!/bin/bash
# SQ counters of the 256x256-crop kernels (1024 crops per dispatch).  Run on the GPU box from the repository root.
R=$PWD
O=$R/gpurun_out/pmcc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/p1 -- python3 $R/tools/prof_crops.py > $O/p1.log 2>&1 || { tail -5 $O/p1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/prof_crops.py > $O/t.log 2>&1 || { tail -5 $O/t.log; exit 1; }
cd $R && python3 tools/pmc_summary.py $(ls $O/p1/*/*counter_collection.csv | head -1) > $O/p1.txt
grep -A9 "rows_kernel\|cols_kernel" $O/p1.txt
cat $(ls $O/t/*/*kernel_stats.csv | head -1) | cut -c1-200 | head -8
