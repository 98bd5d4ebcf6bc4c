#!/bin/bash
# How long do short processes over libbgt.so / the bgt binary take, start to exit?  (GPU box)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,os,subprocess,time,sys
root=os.getcwd()
sys.path.insert(0,root)
import bgt_amd
subprocess.check_call(["make","-s","-C","bgt_amd/host"])
exe="/tmp/api_mine"; lib=os.path.join(root,"bgt_amd","lib")
subprocess.check_call(["gcc","-O1","-I","include","tests/integration/api_dump.c","-o",exe,"-L",lib,"-lbgt","-Wl,-rpath,"+lib])
cases=json.load(open("tests/golden/folds.json"))
for c in cases[:3]:
    for env in ({}, {"BGTH_TRACE":"1","BGT_TRACE":"1"}):
        t=time.time(); p=subprocess.run([exe]+c["args"],cwd="tests/golden/bgt",stdout=subprocess.PIPE,stderr=subprocess.PIPE,env=dict(os.environ,**env)); dt=time.time()-t
        print("api_dump %-40s %.2f s rc %d" % (" ".join(c["args"])[:40], dt, p.returncode)); 
        if env: print(p.stderr.decode()[-1500:])
BGT=os.path.join(root,"bgt_amd","bin","bgt")
for args in (["pbfview","tests/golden/ex1.pbf"],["view","-C","tests/golden/bgt/ex2"]):
    t=time.time(); p=subprocess.run([BGT]+args,stdout=subprocess.PIPE,stderr=subprocess.PIPE); print("bgt %-30s %.2f s rc %d"%(" ".join(args),time.time()-t,p.returncode))
PY
