#!/bin/bash
# kernel averages (rocprofv3 --kernel-trace --stats) of a bench.py run: scratch/kstats_bench.sh <tag> [bench args...]
TAG=$1; shift
export TMPDIR=/tmp
Q="--no-spec-matrix --no-cpu-baseline --no-throughput-leg --no-e2e $*"
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -o ks -- python3 bench.py $Q > /tmp/ks_$TAG.json 2> /tmp/ks_$TAG.log
python3 -c "import sys,json; d=json.loads(open('/tmp/ks_$TAG.json').read().strip().splitlines()[-1]); print('value %.0f ms/step %.3f' % (d['value'], d['ms_per_step']))"
f=$(find /tmp/ks_$TAG -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("%-64s calls %6s avg %10.1f ns  %5s%%" % (r["Name"].split("(")[0][:64], r["Calls"], float(r["AverageNs"]), r["Percentage"][:5]))
PY
mkdir -p gpurun_out/ks_$TAG && cp $f gpurun_out/ks_$TAG/
