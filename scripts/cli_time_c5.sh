#!/bin/bash
# Configuration C5 shape through the CLI on one GPU: two 50,000-sample databases, two sample groups across both,
# `-f 'AC1>0&&AC2==0'`, this repo's `bgt view` next to the compiled reference.  usage: bash scripts/cli_time_c5.sh [sites]
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
S=${1:-100000}
make -s -C bgt_amd/host
T=$(mktemp -d)
./bgt_amd/bin/bgt synth $T/a 50000 $S 5 > /dev/null
./bgt_amd/bin/bgt synth $T/b 50000 $S 6 > /dev/null
ls -la $T | awk '{print $5, $9}' | tail -8
for bin in bgt_amd/bin/bgt oracle/_ref/bgt; do
  [ -x $bin ] || continue
  s=$(date +%s%N); sum=$($bin view -G -s 'pop=="A"' -s 'pop=="B"' -f 'AC1>0&&AC2==0' $T/a $T/b | md5sum | cut -c1-8); e=$(date +%s%N)
  echo "$sum $(( (e - s) / 1000000 )) ms  $bin view -G -s pop==A -s pop==B -f AC1>0&&AC2==0 a b   ($S sites per database)"
done
rm -rf $T
