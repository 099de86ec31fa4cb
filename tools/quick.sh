#!/bin/bash
# GPU tests, then a short bench; prints the headline and the serialized kernel times.  Run on the GPU box from the repository root.
set -u
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputest.log 2>&1
rc=$?
tail -4 gpurun_out/gputest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err || exit 1
python3 -c "
import json; d=json.load(open('gpurun_out/bench_q.json')); print(round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()}, {k:round(v,4) for k,v in d['kernel_ms_per_step_serialized'].items()}, d.get('parity_ok'))"
