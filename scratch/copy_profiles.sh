#!/bin/bash
# copies what profiles/collect.sh left under gpurun_out/prof_<round>/ into profiles/<round>_* (run in the build container, from the repo root)
R=${1:-r6}; D=gpurun_out/prof_$R
if [ "${2:-}" = "lines" ]; then
  for f in bench_default bench_c2 bench_c5 bench_batch256; do cp $D/$f.json profiles/${R}_$f.json; done
  exit 0
fi
for f in bench_default bench_c2 bench_c5 bench_batch256 bench_three_launches bench_four_launches bench_under_rocprof bench_2ranks_one_gpu_gloo; do cp $D/$f.json profiles/${R}_$f.json; done
cp $D/kt/kt_kernel_stats.csv profiles/${R}_kernel_stats_bench_c3.csv
cp $D/kt_c5/kt_kernel_stats.csv profiles/${R}_kernel_stats_bench_c5.csv
cp $D/kt_b256/kt_kernel_stats.csv profiles/${R}_kernel_stats_batch256_c3.csv
cp $D/kt_c2/kt_kernel_stats.csv profiles/${R}_kernel_stats_bench_c2.csv
cp $D/kt_sparse/kt_kernel_stats.csv profiles/${R}_kernel_stats_sparse_deletions_c3.csv
for f in l_sweep l_sweep_del fill_times mixed_radix gather_cost pipe_windows pipe_wide pipe_roles; do cp $D/$f.txt profiles/${R}_$f.txt; done
python profiles/pmc_summarize.py $D pmc > profiles/${R}_pmc_traffic.json
python profiles/pmc_summarize.py $D pmc_c5 > profiles/${R}_pmc_traffic_c5.json
python profiles/pmc_summarize.py $D pmc_b256 > profiles/${R}_pmc_traffic_batch256.json
python profiles/seg_isa_count.py > profiles/${R}_seg_isa.json
grep -h kernel_source_sha profiles/${R}_pmc_traffic*.json
