#!/bin/bash
# three builds on the same box: scratch/ab3.sh <libA.so> <libB.so> [bench args...]; C = the in-tree build.  Order A B C C B A per round.
A=$1; B=$2; shift; shift
Q="--no-cpu-baseline --no-throughput-leg --no-e2e --no-spec-matrix --steps 20 --warmup 3 $*"
one() { python3 bench.py $Q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f/%.2f' % (d['value'], 1e3*(d['kernels_ms_per_launch'].get('rwseg') or 0)))"; }
for i in 1 2 3; do
  a1=$(GH_LIB=$A one); b1=$(GH_LIB=$B one); c1=$(one); c2=$(one); b2=$(GH_LIB=$B one); a2=$(GH_LIB=$A one)
  echo "A $a1 $a2 | B $b1 $b2 | C $c1 $c2"
done
