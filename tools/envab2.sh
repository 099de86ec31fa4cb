# The 1080p bench as it is timed by default, alternating between values of an environment variable:
#   tools/envab2.sh VAR <rounds> v1 v2 ...   ("-" = unset)
var=$1; n=$2; shift; shift
for r in $(seq 1 $n); do for v in "$@"; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  timeout -k 10 200 python3 bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-other-content > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$v failed"; tail -3 gpurun_out/ab.err; continue; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" $v
done; done
