#!/bin/bash
# throughput mode under several builds on the same box: scratch/ab_batch.sh lib1.so lib2.so ...
for r in 1 2; do
for lib in "$@"; do
  v=$(GH_LIB=$lib python3 bench.py --batch 256 --steps 1 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %.1f rw %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms_hip_events']))")
  echo "$(basename $lib) $v"
done; done
