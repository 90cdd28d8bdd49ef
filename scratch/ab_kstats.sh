#!/bin/bash
# kernel averages of two builds on the same box: scratch/ab_kstats.sh <libA.so> <libB.so> [bench args]
A=$1; B=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
Q="--steps 3 --warmup 1 --no-spec-matrix --no-cpu-baseline --no-throughput-leg --no-e2e $*"
for x in A B; do
  lib=$A; [ $x = B ] && lib=$B
  export GH_LIB=$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$x -o ks -- python3 $GRAFT_REPO_ROOT/bench.py $Q > /dev/null 2>&1
  echo "== $x"; python3 - /tmp/ks_$x/ks_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print("%-44s calls %5s avg %9.1f ns" % (r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"])))
PY
done
