#!/bin/bash
# Cold `bgt view -G -f 'AC>0'` at C2 scale and `view -GC` at the HRC shape: the image's strings uploaded by one hipMemcpy from
# pageable memory (BGTH_PLAIN_UPLOAD=1: until round 5) against the threaded pinned-chunk upload (round 6), 8 runs each, interleaved;
# then the stage times of one run of each.  Run on the GPU box: bash scripts/cold_upload_ab.sh
cd $GRAFT_REPO_ROOT; make -s -C bgt_amd/host
T=$(mktemp -d)
./bgt_amd/bin/bgt synth $T/c2 10000 1000000 2 >/dev/null
./bgt_amd/bin/bgt synth $T/hrc 32488 142000 7 >/dev/null
run() { s=$(date +%s%N); "$@" | cat > /dev/null; e=$(date +%s%N); echo $(( (e-s)/1000000 )); }
for DB in "c2 -G -f AC>0" "hrc -GC"; do
  set -- $DB; N=$1; shift
  A=(); B=()
  for i in 1 2 3 4 5 6 7 8; do
    A+=($(BGTH_PLAIN_UPLOAD=1 run ./bgt_amd/bin/bgt view "$@" $T/$N))
    B+=($(run ./bgt_amd/bin/bgt view "$@" $T/$N))
  done
  echo "$N view $*: plain hipMemcpy ms: ${A[*]}"
  echo "$N view $*: pinned chunks   ms: ${B[*]}"
done
echo "--- stages, plain"; BGTH_PLAIN_UPLOAD=1 BGTH_TRACE=1 BGT_TRACE=1 ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/c2 2>&1 >/dev/null | grep trace
echo "--- stages, pinned chunks"; BGTH_TRACE=1 BGT_TRACE=1 ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/c2 2>&1 >/dev/null | grep trace
rm -rf $T
