#!/bin/bash
# Serialized kernel times of the 1080p bench for each library given on the command line (A/B of kernel variants).
for lib in "$@"; do
  SIFT_TEXTURE_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$lib failed"; tail -3 gpurun_out/ab.err; continue; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()}, {k:round(v,4) for k,v in d['kernel_ms_per_step_serialized'].items()})" $lib
done
