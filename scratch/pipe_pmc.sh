#!/bin/bash
# PMC traffic of the pipeline kernel: bash scratch/pipe_pmc.sh <windows> <paths> <tag>   (on the GPU box, from the repo root)
set -u
W=${1:-256}; P=${2:-10}; TAG=${3:-pipe}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $out
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o pmc -- python3 $GRAFT_REPO_ROOT/scratch/pipe_bench.py $W $P 1 C3 > $out/$c.log 2> $out/$c.err
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob("$out/%s/**/*counter_collection.csv" % ctr, recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == ctr:
                k = row["Kernel_Name"].split("(")[0][:60]
                acc[k][ctr] += float(row["Counter_Value"]); n[k][ctr] += 1
for k in sorted(acc, key=lambda k: -acc[k]["FETCH_SIZE"])[:8]:
    print("%-62s launches %4d  FETCH %10.1f MB (x2 = %10.1f)  WRITE %10.1f MB" % (k, max(n[k].values()), acc[k]["FETCH_SIZE"]/1024, 2*acc[k]["FETCH_SIZE"]/1024, acc[k]["WRITE_SIZE"]/1024))
PY
