#!/bin/bash
# Profiles of one round, to be run on the GPU box from the repo root:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash profiles/collect.sh r6'      (COMMIT FIRST: the summaries record whether the tree was dirty)
# Kernel trace and the PMC passes are separate runs (never combined with other trace domains); rocprofv3 is given python3
# directly.  Back in the build container the summaries are made from gpurun_out/prof_<round>/ (where git is):
#   scratch/copy_profiles.sh r6          (copies the bench lines and kernel statistics, summarises the PMC passes, recounts the ISA)
# then commit and: gpurun ... 'bash profiles/collect.sh r5 lines'; scratch/copy_profiles.sh r6 lines    (the bench lines with `traffic`)
set -u
R=${1:-r6}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $out
cd $GRAFT_REPO_ROOT
if [ "${2:-}" = "lines" ]; then
  # second call, once scratch/copy_profiles.sh has put the counter summaries of THIS build into profiles/ (committed): the bench
  # lines again, which then quote `roofline.traffic` from them (the first call's lines cannot: the summaries did not exist yet)
  python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
  python3 bench.py --config C2 > $out/bench_c2.json 2> $out/bench_c2.err
  python3 bench.py --config C5 --steps 2 --warmup 1 > $out/bench_c5.json 2> $out/bench_c5.err
  python3 bench.py --batch 256 --steps 1 --warmup 1 > $out/bench_batch256.json 2> $out/bench_batch256.err
  exit 0
fi
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --config C2 > $out/bench_c2.json 2> $out/bench_c2.err
python3 bench.py --config C5 --steps 2 --warmup 1 > $out/bench_c5.json 2> $out/bench_c5.err
python3 bench.py --gpus 2 --backend gloo --share-gpu --steps 3 --no-cpu-baseline --no-throughput-leg --no-e2e --no-spec-matrix > $out/bench_2ranks_one_gpu_gloo.json 2> $out/bench_2ranks.err
python3 bench.py --batch 256 --steps 1 --warmup 1 > $out/bench_batch256.json 2> $out/bench_batch256.err
GH_FUSE=1 python3 bench.py --no-cpu-baseline --no-throughput-leg --no-e2e --no-spec-matrix > $out/bench_three_launches.json 2> $out/bench_three_launches.err
GH_RWSEG=0 python3 bench.py --no-cpu-baseline --no-throughput-leg --no-e2e --no-spec-matrix > $out/bench_four_launches.json 2> $out/bench_four_launches.err
python3 scratch/l_sweep.py > $out/l_sweep.txt 2>&1
# ... and the same window with '-' at 1 % of the positions beside it (no lag count may cost more than 1.5x)
python3 scratch/l_sweep_del.py > $out/l_sweep_del.txt 2>&1
# the fills alone (HIP events), and the sparse-deletion window (mixed radix) beside the plain one and the five-symbol radix
for c in C2 C3 C5; do python3 scratch/fill_time.py $c 2>&1 | tail -2; done > $out/fill_times.txt
(python3 scratch/mixed_try.py 0.01; GH_MIXED=0 python3 scratch/mixed_try.py 0.01; python3 scratch/mixed_try.py 0) > $out/mixed_radix.txt 2>&1
# the window pipeline (csrc/wpipe.hpp): windows per GPU, the batched launches of rounds 1-4 beside it, two workgroups per CU, and
# the per-role cycle accounting of the diagnostic build (scratch/lib_pipe_prof.so: -DPIPE_PROF, built in the container)
export PYTHONPATH=$GRAFT_REPO_ROOT
(for w in 32 64 128 256 512; do python3 scratch/pipe_bench.py $w 100 2 2>&1 | grep -v amdgpu.ids | tail -1; done
 echo "--- GH_PIPE=0 (batched launches)"; GH_PIPE=0 python3 scratch/pipe_bench.py 256 100 2 2>&1 | grep -v amdgpu.ids | tail -1
 echo "--- GH_PIPE_NT=512 (two workgroups per CU), 512 windows"; GH_PIPE_NT=512 python3 scratch/pipe_bench.py 512 100 2 2>&1 | grep -v amdgpu.ids | tail -1
 echo "--- C2 (1k SNPs, L = 3), 256 windows"; python3 scratch/pipe_bench.py 256 100 2 C2 2>&1 | grep -v amdgpu.ids | tail -1) > $out/pipe_windows.txt 2>&1
# windows with five-candidate positions through the pipeline's WIDE launch (round 6): the sparse-deletion window ('-' at 1 % of the
# positions), the narrow window beside it in the same call, the batched launches such windows took until round 5, the published spec
(for w in 64 128 256; do PB_DEL=0.01 python3 scratch/pipe_bench.py $w 100 3 2>&1 | grep -v amdgpu.ids | tail -1; python3 scratch/pipe_bench.py $w 100 3 2>&1 | grep -v amdgpu.ids | tail -1; done
 echo "--- GH_PIPE_WIDE=0 (the batched launches of rounds 1-4 for such windows)"; PB_DEL=0.01 GH_PIPE_WIDE=0 python3 scratch/pipe_bench.py 256 100 2 2>&1 | grep -v amdgpu.ids | tail -1
 echo "--- conditional E + marginal term"; PB_DEL=0.01 PB_COND=E PB_MT=1 python3 scratch/pipe_bench.py 256 100 3 2>&1 | grep -v amdgpu.ids | tail -1; PB_COND=E PB_MT=1 python3 scratch/pipe_bench.py 256 100 3 2>&1 | grep -v amdgpu.ids | tail -1
 echo "--- a batch of narrow and wide windows (every 4th / every 2nd window has the deletion columns): the two launches side by side"; for m in 4 2; do PB_DEL=0.01 PB_MIX=$m python3 scratch/pipe_bench.py 256 100 2 2>&1 | grep -v amdgpu.ids | tail -1; done
 echo "--- '-' at 0.02 % / 0.2 % / 3 % of the positions, 256 windows"; for f in 0.0002 0.002 0.03; do PB_DEL=$f python3 scratch/pipe_bench.py 256 100 2 2>&1 | grep -v amdgpu.ids | tail -1; done) > $out/pipe_wide.txt 2>&1
if [ -f scratch/lib_pipe_prof.so ]; then
  (GH_LIB=$GRAFT_REPO_ROOT/scratch/lib_pipe_prof.so GH_PIPE_STAMPS=1 python3 scratch/pipe_bench.py 256 21 1 2>&1 | grep -v "amdgpu.ids\|gh_batch_spin" | tail -4
   GH_LIB=$GRAFT_REPO_ROOT/scratch/lib_pipe_prof.so GH_PIPE_STAMPS=1 python3 scratch/pipe_bench.py 64 21 1 2>&1 | grep -v "amdgpu.ids\|gh_batch_spin" | tail -4
   echo "--- the sparse-deletion window (WIDE launch), 256 windows"
   GH_LIB=$GRAFT_REPO_ROOT/scratch/lib_pipe_prof.so GH_PIPE_STAMPS=1 PB_DEL=0.01 python3 scratch/pipe_bench.py 256 21 1 2>&1 | grep -v "amdgpu.ids\|gh_batch_spin" | grep -v " 0.00 0.00 0.00  loader 0.00\|walker: 0.00\|in 0 groups\|waves 0..15: 0 0 0 0 0 0 0 0 0" | tail -5) > $out/pipe_roles.txt 2>&1
fi
Q0="--no-cpu-baseline --no-throughput-leg --no-e2e --no-spec-matrix --steps 30 --warmup 3"
for a in "" "--force-dist" "--force-dist --blocking-gather" "" "--force-dist" "--force-dist --blocking-gather"; do
  python3 bench.py $Q0 $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %.0f hap/s %.4f ms/step' % (sys.argv[1], d['value'], d['ms_per_step']))" "[$a]"
done > $out/gather_cost.txt 2>&1
if [ "${2:-}" = "cpu-full" ]; then
  python3 bench.py --config C2 --steps 2 --cpu-full --no-throughput-leg --no-e2e --no-spec-matrix > $out/bench_c2_cpu_full.json 2> $out/bench_c2_cpu_full.err
  python3 bench.py --steps 2 --cpu-full --no-throughput-leg --no-e2e --no-spec-matrix > $out/bench_c3_cpu_full.json 2> $out/bench_c3_cpu_full.err
fi
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-throughput-leg --no-e2e --no-spec-matrix"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $Q > $out/bench_under_rocprof.json 2> $out/kt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c5 -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --config C5 --steps 1 --warmup 1 $Q > $out/bench_c5_under_rocprof.json 2> $out/kt_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_b256 -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --batch 256 --steps 1 --warmup 1 > $out/bench_batch256_under_rocprof.json 2> $out/kt_b256.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_sparse -o kt -- python3 $GRAFT_REPO_ROOT/scratch/mixed_try.py 0.01 > $out/kt_sparse.txt 2> $out/kt_sparse.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c2 -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --config C2 --steps 3 --warmup 1 $Q > $out/bench_c2_under_rocprof.json 2> $out/kt_c2.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 $Q > $out/pmc_$c.json 2> $out/pmc_$c.err
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_c5_$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --config C5 --paths 100 --steps 1 --warmup 0 $Q > $out/pmc_c5_$c.json 2> $out/pmc_c5_$c.err
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_b256_$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --batch 256 --paths 100 --steps 1 --warmup 0 > $out/pmc_b256_$c.json 2> $out/pmc_b256_$c.err
done
# what travels back is capped at 64 MiB: the per-dispatch traces are not needed for the summaries (kernel statistics, counter sums)
find $out -name "*_kernel_trace.csv" -delete
find $out \( -name "*.db" -o -name "*_agent_info.csv" -o -name "*_domain_stats.csv" \) -delete
du -sh $out
find $out -name "*.csv" | head -60
