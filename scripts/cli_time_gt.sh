#!/bin/bash
# `bgt view` WITH genotype columns (VCF text / BCF) at 10,000 samples x 100,000 sites, this repo's CLI and the compiled
# reference (25.9 s for the reference).  Run on the GPU box: bash scripts/cli_time_gt.sh [samples] [sites]
cd $GRAFT_REPO_ROOT; make -s -C bgt_amd/host
S=${1:-10000}; N=${2:-100000}
T=$(mktemp -d); ./bgt_amd/bin/bgt synth $T/db $S $N 2 >/dev/null
t() { local s=$(date +%s%N); "$@"; local e=$(date +%s%N); echo "$(( (e-s)/1000000 )) ms"; }
echo -n "view (VCF text, all samples) > /dev/null: "; t sh -c "BGT_TRACE=1 ./bgt_amd/bin/bgt view $T/db 2>$T/err > /dev/null"; grep -E "sites:|prepare" $T/err
echo -n "view | wc -c: "; t sh -c "./bgt_amd/bin/bgt view $T/db | wc -c"
echo -n "view -b (BCF) > /dev/null: "; t sh -c "./bgt_amd/bin/bgt view -b $T/db > /dev/null"
echo -n "view -s idx%20==0 (VCF, 5% of the samples): "; t sh -c "./bgt_amd/bin/bgt view -s 'idx%20==0' $T/db > /dev/null"
if [ -x oracle/_ref/bgt ]; then
  M=$(( N / 10 ))
  echo -n "REF view -i1 -n$M (VCF text) > /dev/null [x10 for the whole file]: "; t sh -c "oracle/_ref/bgt view -n $M $T/db > /dev/null"
  cmp <(./bgt_amd/bin/bgt view -n 2000 $T/db) <(oracle/_ref/bgt view -n 2000 $T/db) && echo "first 2000 records identical"
fi
rm -rf $T
