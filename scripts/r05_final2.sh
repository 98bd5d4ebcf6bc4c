#!/bin/bash
# Round-5 closing run, part 2: kernel traces of the five bench workloads (25 calls each), the width table, the groups table.
cd $GRAFT_REPO_ROOT
export TRACE_ONLY=1
bash scripts/profile.sh r05_c2 > /dev/null 2>&1
bash scripts/profile.sh r05_c3 --workload c3 --every 20 > /dev/null 2>&1
bash scripts/profile.sh r05_hrc --workload hrc --sites 142000 > /dev/null 2>&1
bash scripts/profile.sh r05_hrcsub --workload hrc --sites 142000 --every 13 > /dev/null 2>&1
bash scripts/profile.sh r05_c4shard --workload c4 --sites 1253376 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/width_sweep.py 1000,2504,5000,10000,12000,13000,15000,17000,20000,25000,32488,35000,50000,70000,100000 > gpurun_out/width_sweep.log 2>&1; tail -16 gpurun_out/width_sweep.log
python scripts/groups_ab.py 2>/dev/null | tail -6
