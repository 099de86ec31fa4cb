for v in "$@"; do
  if [ "$v" = "-" ]; then unset SIFT_TEXTURE_LIB; else export SIFT_TEXTURE_LIB=$v; fi
  timeout -k 10 200 python3 bench.py --config crops256 --steps 30 --warmup 2 --no-cpu-baseline --no-other-content --time-all-kernels > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$v failed"; tail -3 gpurun_out/ab.err; continue; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms_per_step'].items()})" $v
done
