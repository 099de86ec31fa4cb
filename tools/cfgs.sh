#!/bin/bash
# Short bench of every configuration; one line each.  Run on the GPU box from the repository root.
for cfg in 1080p 720p crops256 4k_multiradius; do
  timeout -k 10 300 python3 bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/cfg_$cfg.json 2> gpurun_out/cfg_$cfg.err || { echo "$cfg failed"; tail -3 gpurun_out/cfg_$cfg.err; continue; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/cfg_'+sys.argv[1]+'.json')); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['path']['frac'],4), {k:round(v,3) for k,v in d['kernel_ms_per_step_serialized'].items()})" $cfg
done
