cd $GRAFT_REPO_ROOT; make -s -C bgt_amd/host
T=$(mktemp -d); ./bgt_amd/bin/bgt synth $T/db 10000 1000000 2 >/dev/null
for i in 1 2 3; do
s=$(date +%s%N); BGT_TRACE=1 ./bgt_amd/bin/bgt view -G -C -r 11:5000000-5001000 $T/db 2>$T/err >/dev/null; e=$(date +%s%N)
echo "start @$((s/1000000)) end @$((e/1000000)) wall $(( (e-s)/1000000 )) ms"; grep "@" $T/err | sed 's/.*\] //'
done
rm -rf $T
