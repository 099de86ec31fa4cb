#!/bin/bash
# chunk-size / submission sweep of the 1080p bench (no CPU baseline)
for c in 0 75 50 38 30 25; do
  for p in "" "--pipeline"; do
    echo "chunk=$c $p"
    timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --chunk $c $p | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  fps %.0f ms/step %.3f' % (r['value'], r['ms_per_step']), {k: round(v,3) for k,v in r['kernel_ms_per_step'].items()})
" || exit 1
  done
done
