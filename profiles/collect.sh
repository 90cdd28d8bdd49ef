#!/bin/bash
# Profiles of one round, to be run on the GPU box from the repo root:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash profiles/collect.sh r2'
# Kernel trace and the two PMC passes are separate runs (never combined with other trace domains); rocprofv3 is
# given python3 directly.  Summaries are then copied from gpurun_out/prof_<round>/ into profiles/ by hand
# (profiles/pmc_summarize.py for the counters).
set -u
R=${1:-r2}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --config C2 > $out/bench_c2.json 2> $out/bench_c2.err
python3 bench.py --config C5 --steps 2 --warmup 1 > $out/bench_c5.json 2> $out/bench_c5.err
python3 bench.py --gpus 2 --backend gloo --share-gpu --steps 3 --no-cpu-baseline --no-throughput-leg --no-e2e > $out/bench_2ranks_one_gpu_gloo.json 2> $out/bench_2ranks.err
python3 bench.py --batch 256 --steps 1 --warmup 1 > $out/bench_batch256.json 2> $out/bench_batch256.err
if [ "${2:-}" = "cpu-full" ]; then
  python3 bench.py --config C2 --steps 2 --cpu-full --no-throughput-leg --no-e2e > $out/bench_c2_cpu_full.json 2> $out/bench_c2_cpu_full.err
  python3 bench.py --steps 2 --cpu-full --no-throughput-leg --no-e2e > $out/bench_c3_cpu_full.json 2> $out/bench_c3_cpu_full.err
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-leg --no-e2e > $out/bench_under_rocprof.json 2> $out/kt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c5 -o kt_c5 -- python3 $GRAFT_REPO_ROOT/bench.py --config C5 --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-leg --no-e2e > $out/bench_c5_under_rocprof.json 2> $out/kt_c5.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-throughput-leg --no-e2e > $out/pmc_$c.json 2> $out/pmc_$c.err
done
find $out -name "*.csv" | head -40
