#!/bin/bash
# 1080p bench once per value of an environment variable: tools/envab.sh NAME v1 v2 ...
name=$1; shift
for v in "$@"; do
  export $name=$v
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$v failed"; tail -3 gpurun_out/ab.err; continue; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()}, {k:round(v,4) for k,v in d['kernel_ms_per_step_serialized'].items()})" $name=$v
done
