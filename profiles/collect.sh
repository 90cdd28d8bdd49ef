#!/bin/bash
# Profiles of round 1, to be run on the GPU box from the repo root:
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/collect.sh'
# Kernel trace and the two PMC passes are separate runs (never combined with other trace domains).
set -u
out=gpurun_out/prof_r1
mkdir -p $out/kt $out/kt_batch $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-leg > $out/bench_under_rocprof.json 2> $out/kt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_batch -- python3 bench.py --batch 256 --steps 1 --warmup 1 > $out/bench_batch256_under_rocprof.json 2> $out/kt_batch.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-throughput-leg > $out/pmc_$c.json 2> $out/pmc_$c.err
done
find $out -name "*.csv" | head -40
