#!/bin/bash
# Back-to-back `bgt view` processes: does handing the tear-down to a child (default) cost the NEXT command what it saves this one?
# GPU box: bash scripts/cli_loop_time.sh
cd ${GRAFT_REPO_ROOT:-$(pwd)}; make -s -C bgt_amd/host
T=$(mktemp -d); ./bgt_amd/bin/bgt synth $T/db 10000 262144 2 >/dev/null
for mode in fork nofork fork nofork; do
  if [ $mode = nofork ]; then export BGT_NO_FORK=1; else unset BGT_NO_FORK; fi
  s=$(date +%s%N); for i in $(seq 20); do ./bgt_amd/bin/bgt view -G -C -r 11:5000000-5001000 $T/db > /dev/null; done; e=$(date +%s%N)
  echo "$mode: 20 region queries back to back: $(( (e-s)/1000000 )) ms"
  s=$(date +%s%N); for i in $(seq 8); do ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/db > /dev/null; done; e=$(date +%s%N)
  echo "$mode: 8 whole-file walks back to back: $(( (e-s)/1000000 )) ms"
done
sleep 1; rm -rf $T
