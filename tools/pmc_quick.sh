#!/bin/bash
# One SQ counter pass over tools/prof1.py (30-frame dispatches); prints the stencil kernel's counters.
# usage: tools/pmc_quick.sh [library]   (run on the GPU box from the repository root)
R=$PWD
[ -n "${1:-}" ] && export SIFT_TEXTURE_LIB=$R/$1
O=$R/gpurun_out/pmcq
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -- python3 $R/tools/prof1.py > $O/p1.log 2>&1 || { tail -5 $O/p1.log; exit 1; }
cd $R && python3 tools/pmc_summary.py $(ls $O/p1/*/*counter_collection.csv | head -1) > $O/p1.txt
grep -A9 "stencil" $O/p1.txt
