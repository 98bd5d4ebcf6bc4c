#!/bin/bash
# Configuration C3 shape through the CLI on one GPU: a 100,000-sample database, `-s` picks every 20th sample
# (5,000 of 100,000), AC/AN; and the whole 200,000-column cohort on the first sites.  This repo's `bgt view` next to
# the compiled reference.  usage: bash scripts/cli_time_c3.sh [sites]
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
S=${1:-100000}
make -s -C bgt_amd/host
T=$(mktemp -d)
./bgt_amd/bin/bgt synth $T/c3 100000 $S 3 > /dev/null
run() { local bin=$1; shift; [ -x $bin ] || return 0
  local s=$(date +%s%N); local sum=$($bin "$@" $T/c3 | md5sum | cut -c1-8); local e=$(date +%s%N)
  echo "$sum $(( (e - s) / 1000000 )) ms  $bin $*"; }
for bin in bgt_amd/bin/bgt oracle/_ref/bgt; do
  run $bin view -G -C -s 'idx%20==0'
  run $bin view -G -C -n 4000
done
rm -rf $T
