#!/bin/bash
# Everything profiles/ holds for one round, run on the GPU box from the repository root:
#   tools/profile_round.sh <tag>        -> gpurun_out/<tag>/
set -u
TAG=${1:-r3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
step() { echo "== $*" >&2; }
step bench 1080p
timeout -k 10 400 python3 bench.py > $OUT/bench_1080p.json 2> $OUT/bench_1080p.err || exit 1
for c in crops256 4k_multiradius 720p; do
  step bench $c
  timeout -k 10 300 python3 bench.py --config $c --steps 20 --warmup 2 --cpu-seconds 8 > $OUT/bench_$c.json 2> $OUT/bench_$c.err || exit 1
done
step bench clips10k
timeout -k 10 300 python3 bench.py --config clips10k --warmup 2 > $OUT/bench_clips10k.json 2> $OUT/bench_clips10k.err || exit 1
step host-fed
timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --host-fed > $OUT/bench_hostfed.json 2> $OUT/bench_hostfed.err || exit 1
step latency
timeout -k 10 200 python3 tools/latency.py > $OUT/latency.txt 2>&1 || exit 1
cd /tmp
step kernel trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-other-content > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
step pmc fetch
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/prof1.py > $OUT/pmc_fetch.log 2>&1 || exit 1
step pmc write
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/prof1.py > $OUT/pmc_write.log 2>&1 || exit 1
step pmc sq
timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/tools/prof1.py > $OUT/pmc_sq.log 2>&1 || exit 1
cd $ROOT
step summaries
python3 tools/timeline.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) 3 > $OUT/timeline.txt
python3 tools/pmc_summary.py $(ls $OUT/pmc_sq/*/*counter_collection.csv | head -1) > $OUT/pmc_sq_summary.txt
python3 profiles/traffic_from_pmc.py $(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $OUT/pmc_write/*/*counter_collection.csv | head -1) 30 $OUT/traffic.json > /dev/null
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
echo done
