#!/bin/bash
# SQ counter passes over tools/prof1.py; prints the FFT kernels' counters.  Run on the GPU box from the repository root.
R=$PWD
O=$R/gpurun_out/pmcf
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -- python3 $R/tools/prof1.py > $O/p1.log 2>&1 || { tail -5 $O/p1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $O/p2 -- python3 $R/tools/prof1.py > $O/p2.log 2>&1 || { tail -5 $O/p2.log; }
cd $R && python3 tools/pmc_summary.py $(ls $O/p1/*/*counter_collection.csv | head -1) > $O/p1.txt
python3 tools/pmc_summary.py $(ls $O/p2/*/*counter_collection.csv | head -1) > $O/p2.txt 2>/dev/null
grep -A9 -E "rows_kernel|cols_kernel" $O/p1.txt; grep -A9 -E "rows_kernel|cols_kernel" $O/p2.txt
