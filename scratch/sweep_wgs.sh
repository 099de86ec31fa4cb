#!/bin/bash
for n in 0 3 2 1; do
  for p in "" "--pipeline"; do
    echo "stencil wgs/CU cap=$n $p"
    STX_STENCIL_WGS=$n timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline $p | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  fps %.0f ms/step %.3f' % (r['value'], r['ms_per_step']), {k: round(v,3) for k,v in r['kernel_ms_per_step'].items()})
print('     serialized', {k: round(v,3) for k,v in r['kernel_ms_per_step_serialized'].items()})
" || exit 1
  done
done
