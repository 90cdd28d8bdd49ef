#!/bin/bash
# scratch/ab_env.sh "<ENV=VAL for variant A>" [bench args...]: default build settings against the same with an environment variable set, three rounds
A=$1; shift
Q="--no-cpu-baseline --no-throughput-leg --no-e2e --no-spec-matrix --steps 20 --warmup 3 $*"
one() { python3 bench.py $Q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])"; }
for i in 1 2 3; do
  a1=$(env $A python3 bench.py $Q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])"); c1=$(one); c2=$(one)
  a2=$(env $A python3 bench.py $Q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])")
  echo "[$A] $a1 $a2 | default $c1 $c2"
done
