#!/bin/bash
# A/B of two builds of libgretel_hip.so on the same box: scratch/ab_bench.sh <libA.so> [bench args...]; B = the in-tree build.
# Order A B B A per round (a run right after another one is not quite the same as the first).
A=$1; shift
Q="--no-cpu-baseline --no-throughput-leg --no-e2e --no-spec-matrix --steps 20 --warmup 3 $*"
one() { python3 bench.py $Q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])"; }
for i in 1 2 3; do
  a1=$(GH_LIB=$A one); b1=$(one); b2=$(one); a2=$(GH_LIB=$A one)
  echo "A $a1 $a2 | B $b1 $b2"
done
