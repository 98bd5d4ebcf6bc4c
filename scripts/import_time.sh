#!/bin/bash
# `bgt import` wall time, this build (host atomizer + device PBWT encoder) vs the compiled reference (all CPU), on a VCF with
# genotypes exported from a synthetic database.  Run on the GPU box: bash scripts/import_time.sh [samples] [sites]
cd $GRAFT_REPO_ROOT; make -s -C bgt_amd/host
NS=${1:-2504}; NR=${2:-50000}
T=$(mktemp -d)
BGTH_SYNTH_MISSING_PPM=0 ./bgt_amd/bin/bgt synth $T/db $NS $NR 1 > /dev/null    # (no missing calls: with them the generator puts code 3 on two-allele sites, which `view` prints as allele 2 -- not a valid VCF)
./bgt_amd/bin/bgt view $T/db > $T/in.vcf; ls -la $T/in.vcf | awk '{print "VCF bytes", $5}'
for i in 1 2; do s=$(date +%s%N); ./bgt_amd/bin/bgt import -S $T/mine $T/in.vcf; e=$(date +%s%N); echo "this build: import $(( (e-s)/1000000 )) ms"; done
s=$(date +%s%N); oracle/_ref/bgt import -S $T/want $T/in.vcf; e=$(date +%s%N); echo "reference : import $(( (e-s)/1000000 )) ms"
cmp $T/mine.pbf $T/want.pbf && cmp $T/mine.bcf $T/want.bcf && cmp $T/mine.spl $T/want.spl && echo "files identical"
rm -rf $T
