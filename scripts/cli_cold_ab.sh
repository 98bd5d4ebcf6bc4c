#!/bin/bash
# Cold `bgt view -G -f 'AC>0'` at C2 scale: the image parsed beside the HIP runtime's start (default) against waiting for the
# runtime first (BGTH_OPEN_WAIT_FIRST=1), 8 runs each, interleaved.  Run on the GPU box: bash scripts/cli_cold_ab.sh
cd $GRAFT_REPO_ROOT; make -s -C bgt_amd/host
T=$(mktemp -d); ./bgt_amd/bin/bgt synth $T/db 10000 1000000 2 >/dev/null
A=(); B=()
for i in 1 2 3 4 5 6 7 8; do
  s=$(date +%s%N); BGTH_OPEN_WAIT_FIRST=1 ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/db | cat > /dev/null; e=$(date +%s%N); A+=($(( (e-s)/1000000 )))
  s=$(date +%s%N); ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/db | cat > /dev/null; e=$(date +%s%N); B+=($(( (e-s)/1000000 )))
done
echo "wait-first  ms: ${A[*]}"
echo "beside-init ms: ${B[*]}"
BGTH_TRACE=1 BGT_TRACE=1 ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/db 2>&1 >/dev/null | grep trace
rm -rf $T
