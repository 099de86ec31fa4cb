#!/bin/bash
timeout -k 10 300 python3 bench.py --config 4k_multiradius --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/c5.json 2> gpurun_out/c5.err || { tail -3 gpurun_out/c5.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/c5.json')); print(round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step_serialized'].items()})"
