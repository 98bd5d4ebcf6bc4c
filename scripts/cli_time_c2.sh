#!/bin/bash
# CLI wall time at C2 scale (10,000 samples x 1,000,000 sites written by `bgt synth`): whole-file filter scan,
# `view -G`, and a small region query with and without counts, this repo's CLI and the compiled reference.
# Run on the GPU box: bash scripts/cli_time_c2.sh   (BGTH_TRACE=1 prints the image-open stages)
cd $GRAFT_REPO_ROOT; make -s -C bgt_amd/host
T=$(mktemp -d); s=$(date +%s%N); ./bgt_amd/bin/bgt synth $T/db 10000 1000000 2 >/dev/null; e=$(date +%s%N); echo "synth $(( (e-s)/1000000 )) ms"; ls -la $T
for i in 1 2; do
s=$(date +%s%N); BGTH_TRACE=1 ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/db > /dev/null; e=$(date +%s%N); echo "view -G -f AC>0: $(( (e-s)/1000000 )) ms"
done
s=$(date +%s%N); ./bgt_amd/bin/bgt view -G $T/db > /dev/null; e=$(date +%s%N); echo "view -G (no device): $(( (e-s)/1000000 )) ms"
s=$(date +%s%N); ./bgt_amd/bin/bgt view -G -C -r 11:5000000-5001000 $T/db | wc -l; e=$(date +%s%N); echo "view -GC -r small: $(( (e-s)/1000000 )) ms"
s=$(date +%s%N); oracle/_ref/bgt view -G -C -r 11:5000000-5001000 $T/db | wc -l; e=$(date +%s%N); echo "REF view -GC -r small: $(( (e-s)/1000000 )) ms"
s=$(date +%s%N); ./bgt_amd/bin/bgt view -G -r 11:5000000-5001000 $T/db | wc -l; e=$(date +%s%N); echo "view -G -r small (no device): $(( (e-s)/1000000 )) ms"
s=$(date +%s%N); oracle/_ref/bgt view -G -r 11:5000000-5001000 $T/db | wc -l; e=$(date +%s%N); echo "REF view -G -r small: $(( (e-s)/1000000 )) ms"
rm -rf $T
