#!/bin/bash
# kernel averages (rocprofv3 --kernel-trace --stats) of one python script: scratch/kstats_script.sh <tag> <script.py> [args...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
SCRIPT=$GRAFT_REPO_ROOT/$1; shift
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -o ks -- python3 $SCRIPT "$@" > /tmp/ks_$TAG.log 2>&1
tail -2 /tmp/ks_$TAG.log
f=$(find /tmp/ks_$TAG -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("%-60s calls %5s avg %9.1f ns" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"])))
PY
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/ks_$TAG && cp $f $GRAFT_REPO_ROOT/gpurun_out/ks_$TAG/
