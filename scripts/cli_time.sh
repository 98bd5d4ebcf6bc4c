#!/bin/bash
# Wall time of `bgt view` variants, this repo's CLI next to the compiled reference (oracle/_ref/bgt), on a
# C1-shaped database written by `bgt synth`.  Run on the GPU box: bash scripts/cli_time.sh [samples] [sites]
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-2504}; S=${2:-50000}
make -s -C bgt_amd/host
T=$(mktemp -d)
./bgt_amd/bin/bgt synth $T/db $N $S 1 > /dev/null
run() {   # binary, args...
  local bin=$1; shift
  [ -x $bin ] || return 0
  local sum=$($bin "$@" $T/db | md5sum | cut -c1-8)          # identity check; timed separately without the md5 pipe
  local s=$(date +%s%N); $bin "$@" $T/db > /dev/null; local e=$(date +%s%N)
  echo "$sum $(( (e - s) / 1000000 )) ms  $bin $*"
}
for bin in oracle/_ref/bgt bgt_amd/bin/bgt; do
  run $bin view -G
  run $bin view -G -f 'AC>0'
  run $bin view -G -C
  run $bin view
  run $bin view -b
  run $bin view -s 'pop=="A"'
  run $bin view -G -s 'pop=="A"' -s 'pop=="B"' -f 'AC1>0&&AC2==0'
done
rm -rf $T
