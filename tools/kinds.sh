#!/bin/bash
# Serialized kernel times of the 1080p bench for each synthetic content class.
for kind in natural noise flat ramp step tie; do
  timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --kind $kind > gpurun_out/k.json 2> gpurun_out/k.err || { echo "$kind failed"; tail -3 gpurun_out/k.err; continue; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/k.json')); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step_serialized'].items()}, 'fft64 frames', d.get('fft64_frames_in_timed_region'))" $kind
done
