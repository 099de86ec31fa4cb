#!/bin/bash
for mode in "" "--blocking" "--chunk 75" "--blocking --chunk 75"; do
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline $mode > gpurun_out/sp.json 2> gpurun_out/sp.err || { echo "$mode failed"; tail -3 gpurun_out/sp.err; continue; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/sp.json')); print('mode',sys.argv[1:], round(d['value']), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" $mode
done
