#!/bin/bash
# 1080p bench at several step counts: tools/stepsab.sh 8 40 ...
for k in "$@"; do
  timeout -k 10 200 python3 bench.py --steps $k --warmup 2 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$k failed"; tail -3 gpurun_out/ab.err; continue; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); print('steps', sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()})" $k
done
