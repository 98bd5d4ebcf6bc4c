#!/bin/bash
# The metric's own command line on one C4 shard as a DATABASE on disk (100,000 samples x 153 file blocks = 1,253,376
# sites, 2.4 GB of .pbf): `bgt view -G -f'AC>0'`, process start to last byte, with the stage times of BGT_TRACE /
# BGTH_TRACE.  GPU box: bash scripts/cli_time_c4shard.sh [blocks]
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
B=${1:-153}
S=$(( B * 8192 ))
make -s -C bgt_amd/host
T=$(mktemp -d)
s=$(date +%s%N); ./bgt_amd/bin/bgt synth $T/c4 100000 $S 4 > /dev/null; e=$(date +%s%N); echo "synth $S sites: $(( (e-s)/1000000 )) ms"; ls -la $T | tail -4
for i in $(seq ${REPS:-4}); do
  sleep ${GAP:-0}
  s=$(date +%s%N); BGT_TRACE=1 BGTH_TRACE=1 ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/c4 2> $T/trace.txt | wc -l; e=$(date +%s%N)
  echo "view -G -f AC>0: $(( (e-s)/1000000 )) ms   [$(grep -E 'upload|arena|prepare|sites:' $T/trace.txt | sed 's/.*\] //' | tr '\n' ';')]"
done
cat $T/trace.txt
if [ -n "$EXTRA" ]; then
  for v in "BGTH_DIR_ARENA_MB=40000" "BGT_CLEAN_EXIT=1" "BGTH_OPEN_HINT=none"; do
    echo "--- $v"
    for i in 1 2 3; do
      s=$(date +%s%N); env $v BGT_TRACE=1 BGTH_TRACE=1 ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/c4 2> $T/trace.txt | wc -l; e=$(date +%s%N)
      echo "view -G -f AC>0: $(( (e-s)/1000000 )) ms   [$(grep -E 'upload|arena|sub-check|prepare|sites:' $T/trace.txt | sed 's/.*\] //' | tr '\n' ';')]"
    done
  done
fi
rm -rf $T
