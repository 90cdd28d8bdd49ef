#!/bin/bash
Q="--gpus 2 --backend gloo --share-gpu --steps 5 --warmup 2 --no-cpu-baseline --no-throughput-leg --no-e2e --no-spec-matrix"
for r in 1 2; do
for d in scratch/ab/wt_fb9018c_1 scratch/ab/wt_161ca73 scratch/ab/wt_b938f46 .; do
  v=$(cd $d && python3 bench.py $Q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f hap/s %.1f ms/step' % (d['value'], d['ms_per_step']))")
  echo "$d: $v"
done; done
