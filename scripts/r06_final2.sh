#!/bin/bash
# Round-6 closing run, part 2: kernel traces (rocprofv3 --kernel-trace --stats, 25 calls each) of the five bench workloads and the
# PMC passes of C2 (C3's were taken when its start ranks changed: profiles/r06_c3).
cd $GRAFT_REPO_ROOT
bash scripts/profile.sh r06_c2 > /dev/null 2>&1
export TRACE_ONLY=1
bash scripts/profile.sh r06_hrc --workload hrc --sites 142000 > /dev/null 2>&1
bash scripts/profile.sh r06_hrcsub --workload hrc --sites 142000 --every 13 > /dev/null 2>&1
bash scripts/profile.sh r06_c4shard --workload c4 --sites 1253376 > /dev/null 2>&1
ls gpurun_out/prof_r06_*/trace/*/ | head -40
