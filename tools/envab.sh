#!/bin/bash
# The 1080p bench once per value of an environment variable: tools/envab.sh VAR v1 v2 ...  ("-" = unset)
var=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  timeout -k 10 200 python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --time-all-kernels > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$v failed"; tail -3 gpurun_out/ab.err; continue; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()})" $v
done
