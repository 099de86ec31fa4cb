#!/bin/bash
# FETCH_SIZE and WRITE_SIZE passes over tools/prof1.py for each library given; prints the stencil kernel's per-frame bytes.
# usage: tools/pmc_traffic.sh lib1 [lib2 ...]   (run on the GPU box from the repository root)
R=$PWD
export TMPDIR=/tmp
for lib in "$@"; do
  export SIFT_TEXTURE_LIB=$R/$lib
  O=$R/gpurun_out/pmct; rm -rf $O; mkdir -p $O
  cd /tmp
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/tools/prof1.py > $O/f.log 2>&1 || { tail -5 $O/f.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $R/tools/prof1.py > $O/w.log 2>&1 || { tail -5 $O/w.log; exit 1; }
  cd $R
  python3 profiles/traffic_from_pmc.py $(ls $O/f/*/*counter_collection.csv | head -1) $(ls $O/w/*/*counter_collection.csv | head -1) 30 $O/t.json > /dev/null
  python3 -c "
import json,sys; d=json.load(open('$O/t.json'))['per_frame_bytes']; print(sys.argv[1], {k:(round(v['read']/1e6,2), round(v['write']/1e6,2)) for k,v in d.items() if 'read' in v})" $lib
done
