#!/bin/bash
# scratch/ab_cfg.sh "<bench args>" lib1 lib2 ... ("-" = the in-tree build): value of each build for one bench configuration, three rounds
ARGS=$1; shift
for i in 1 2 3; do
  line=""
  for l in "$@"; do
    if [ "$l" = "-" ]; then v=$(python3 bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])");
    else v=$(GH_LIB=$l python3 bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])"); fi
    line="$line $l=$v"
  done
  echo "$line"
done
