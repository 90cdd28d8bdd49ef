#!/bin/bash
# The layout of the pipeline's table (wpipe.hpp: pipe_gp_piece) against the two it was measured against, and the loaders making
# the far lags against reading them -- times in ONE call (boxes differ by 10 %), then the counters per variant.
# On the GPU box, from the repo root:   bash scratch/table_layout.sh > gpurun_out/table_layout.txt
# Needs scratch/lib_gp_rowmajor.so and scratch/lib_gp_lagmajor.so (hipcc ... -DPIPE_GP_ROWMAJOR / -DPIPE_GP_LAGMAJOR, see scratch/README.md).
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
echo "kernel sources $(python3 -c 'import bench; print(bench.kernel_source_sha())'), git $(cat .git/HEAD 2>/dev/null || echo '(snapshot)')"
echo "== kernel ms per 100 paths x 256 windows of the C3 contig (A f32 | E + marginal term f32), three rounds, same call"
python3 scratch/pipe_bench.py 256 100 2 > /dev/null 2>&1
for r in 1 2 3; do
  for v in "rowmajor:scratch/lib_gp_rowmajor.so:0" "rowmajor+made:scratch/lib_gp_rowmajor.so:1" "lagmajor:scratch/lib_gp_lagmajor.so:0" "lagmajor+made:scratch/lib_gp_lagmajor.so:1" "whole-lines:gretel_amd/libgretel_hip.so:0" "whole-lines+made:gretel_amd/libgretel_hip.so:1"; do
    n=${v%%:*}; rest=${v#*:}; l=${rest%%:*}; s=${rest##*:}
    a=$(GH_PIPE_SYNTH=$s GH_LIB=$PWD/$l python3 scratch/pipe_bench.py 256 100 2 2>&1 | tail -1 | sed "s/.*kernel \([0-9.]*\) ms.*/\1/")
    e=$(PB_COND=E PB_MT=1 GH_PIPE_SYNTH=$s GH_LIB=$PWD/$l python3 scratch/pipe_bench.py 256 100 2 2>&1 | tail -1 | sed "s/.*kernel \([0-9.]*\) ms.*/\1/")
    printf "%-18s A %7s   E+mt %7s\n" $n $a $e
  done
done
echo "== counters of k_wpipe, 256 windows x 40 paths, A f32 (FETCH_SIZE x 2 + WRITE_SIZE = HBM bytes, MI355X_MICROARCH.md)"
for v in "rowmajor:scratch/lib_gp_rowmajor.so:0" "rowmajor+made:scratch/lib_gp_rowmajor.so:1" "lagmajor+made:scratch/lib_gp_lagmajor.so:1" "whole-lines:gretel_amd/libgretel_hip.so:0" "whole-lines+made:gretel_amd/libgretel_hip.so:1"; do
  n=${v%%:*}; rest=${v#*:}; l=${rest%%:*}; s=${rest##*:}
  printf "%-18s " $n
  GH_PIPE_SYNTH=$s GH_LIB=$GRAFT_REPO_ROOT/$l bash scratch/pipe_pmc.sh 256 40 tl_$n 2>&1 | grep wpipe | sed 's/void k_wpipe<float, 5, 1024> *//'
done
