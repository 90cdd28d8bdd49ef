#!/bin/bash
# two ranks on one GPU over gloo under several settings (the multi-rank code path on a one-GPU box)
Q="--gpus 2 --backend gloo --share-gpu --steps 5 --warmup 2 --no-cpu-baseline --no-throughput-leg --no-e2e --no-spec-matrix"
for e in "GH_RWSEG=1" "GH_RWSEG=0" "GH_RWSEG=1" "GH_RWSEG=0"; do
  v=$(env $e python3 bench.py $Q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f hap/s %.1f ms/step' % (d['value'], d['ms_per_step']))")
  echo "$e: $v"
done
