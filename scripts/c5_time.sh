#!/bin/bash
# configs[4] of BASELINE.json (C5) in its product form: two databases of 50,000 samples each (seeds 5 and 6: the same
# positions, independent alleles), two sample groups, `bgt view -G -s'pop=="A"' -s'pop=="B"' -f'AC1>0&&AC2==0' a b`.
# This repo's CLI -- one image per database on the device, and with BGT_GPUS every database dealt over shards -- next
# to the compiled reference (oracle/_ref/bgt) on the same files; outputs compared by md5.
# Run on the GPU box: bash scripts/c5_time.sh [samples-per-db] [sites-per-db]
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-50000}; S=${2:-65536}
make -s -C bgt_amd/host
T=$(mktemp -d)
./bgt_amd/bin/bgt synth $T/a $N $S 5 > /dev/null
./bgt_amd/bin/bgt synth $T/b $N $S 6 > /dev/null
ARGS=(view -G -s 'pop=="A"' -s 'pop=="B"' -f 'AC1>0&&AC2==0' $T/a $T/b)
run() {   # label, binary
  [ -x $2 ] || { echo "$1: $2 not built"; return 0; }
  local s=$(date +%s%N); local out=$($2 "${ARGS[@]}" | tee >(wc -l > $T/lines) | md5sum | cut -c1-12); local e=$(date +%s%N)
  echo "$1: $(( (e - s) / 1000000 )) ms  md5 $out  lines $(cat $T/lines)"
}
echo "two databases x $N samples x $S sites"
run "this repo, one device image per database" bgt_amd/bin/bgt
BGT_GPUS=0,0,0,0 run "this repo, BGT_GPUS=0,0,0,0 (4 shards per database)" bgt_amd/bin/bgt
run "reference (1 core)" oracle/_ref/bgt
rm -rf $T
