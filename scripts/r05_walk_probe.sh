#!/bin/bash
# Phase ablation of the walk-only kernel (directory path) on one C4 shard and the HRC shape: profiling build, timing only.
cd $GRAFT_REPO_ROOT; export O=$PWD/gpurun_out/r05_walk; rm -rf $O; mkdir -p $O; Q=$PWD/scripts/quick_times.py
export BGT_AMD_LIB=$PWD/bgt_amd/lib/libbgt_hip_ablate.so
for shape in c4 hrc; do
  echo "== $shape phase ticks" >> $O/ablate.log
  BGTH_DEBUG_TIMES=1 python $Q $shape 2>&1 | grep "memtime\|^$shape" | tail -2 >> $O/ablate.log
  for sk in 0 65536 524288 1048576 589824 1114112 1638400; do echo "== $shape skip $sk" >> $O/ablate.log; BGTH_DEBUG_SKIP=$sk python $Q $shape 2>/dev/null >> $O/ablate.log; done
done
cat $O/ablate.log
