#!/bin/bash
# bench.py --config clips10k (a shortened run) once per value of an environment variable: tools/clips_ab.sh VAR v1 v2 ... ("-" = unset)
var=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  timeout -k 10 200 python3 bench.py --config clips10k --steps 400 --warmup 2 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$v failed"; tail -3 gpurun_out/ab.err; continue; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); s=d['streaming']; print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), round(s['frames_per_s_analysis_only_rank0']), round(s['synth_kernel_ms_per_clip_in_situ'],3))" $v
done
