#!/bin/bash
# Timeline of the bulk walk inside a resident host (BGT_TRACE marks of bgtm_write_vcf_bulk) for the metric's command at C2
# scale.  Run on the GPU box: bash scripts/e2e_timeline.sh [threads]
cd $GRAFT_REPO_ROOT; make -s -C bgt_amd/host
T=$(mktemp -d); ./bgt_amd/bin/bgt synth $T/db 10000 1000000 2 >/dev/null
[ -n "$1" ] && export BGT_THREADS=$1
BGT_TRACE=1 ./bgt_amd/bin/bgt-server -u $T/sock $T/db 2> $T/srv.err &
SRV=$!
for i in $(seq 1 600); do [ -S $T/sock ] && break; sleep 0.05; done
export BGT_SERVER=$T/sock
for i in 1 2 3 4; do
  s=$(date +%s%N); ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/db | cat > /dev/null; e=$(date +%s%N); echo "client wall $(( (e-s)/1000000 )) ms"
done
unset BGT_SERVER
kill $SRV; wait $SRV 2>/dev/null
tail -32 $T/srv.err
rm -rf $T
