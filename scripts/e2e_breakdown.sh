#!/bin/bash
# Where the resident command line's time goes at C2 scale (10,000 samples x 1,000,000 sites, `view -G -f 'AC>0'`,
# image held by `bgt-server -u`): the same query with its output into a pipe, /dev/null and a file in /dev/shm,
# with 8 ... 128 formatter threads.  Run on the GPU box: bash scripts/e2e_breakdown.sh [sites]
cd $GRAFT_REPO_ROOT; make -s -C bgt_amd/host
SITES=${1:-1000000}
T=$(mktemp -d); ./bgt_amd/bin/bgt synth $T/db 10000 $SITES 2 >/dev/null
echo "host cores: $(nproc)"
ms() { echo $(( ($2 - $1) / 1000000 )); }
for TH in "" 8 16 32 64 128; do
  if [ -n "$TH" ]; then export BGT_THREADS=$TH; else unset BGT_THREADS; fi
  BGT_TRACE=1 ./bgt_amd/bin/bgt-server -u $T/sock $T/db 2> $T/srv.err &
  SRV=$!
  for i in $(seq 1 600); do [ -S $T/sock ] && break; sleep 0.05; done
  export BGT_SERVER=$T/sock
  ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/db > /dev/null      # warm
  best_pipe=999999; best_null=999999; best_file=999999
  for i in 1 2 3 4 5; do
    s=$(date +%s%N); ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/db | cat > /dev/null; e=$(date +%s%N); d=$(ms $s $e); [ $d -lt $best_pipe ] && best_pipe=$d
    s=$(date +%s%N); ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/db > /dev/null; e=$(date +%s%N); d=$(ms $s $e); [ $d -lt $best_null ] && best_null=$d
    s=$(date +%s%N); ./bgt_amd/bin/bgt view -G -f 'AC>0' $T/db > /dev/shm/e2e_out.vcf; e=$(date +%s%N); d=$(ms $s $e); [ $d -lt $best_file ] && best_file=$d
  done
  echo "threads=${TH:-default}: pipe $best_pipe ms, /dev/null $best_null ms, /dev/shm file $best_file ms ($(wc -c < /dev/shm/e2e_out.vcf) bytes)"
  unset BGT_SERVER
  kill $SRV; wait $SRV 2>/dev/null
  rm -f $T/sock
done
tail -30 $T/srv.err
rm -rf $T /dev/shm/e2e_out.vcf
