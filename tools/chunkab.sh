#!/bin/bash
# 1080p bench at several internal chunk sizes: tools/chunkab.sh 0 75 50 ...
for c in "$@"; do
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --chunk $c > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$c failed"; tail -3 gpurun_out/ab.err; continue; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); print('chunk', sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()}, {k:round(v,4) for k,v in d['kernel_ms_per_step_serialized'].items()})" $c
done
