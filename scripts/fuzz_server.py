#!/usr/bin/env python3
"""Differential fuzz of bgt-server on the GPU box: random query strings answered by this repo's server and by the same
source linked with the compiled reference library (one-shot mode, -q), bodies and exit statuses compared; every
20th query also goes over HTTP to a resident server whose body must equal the one-shot answer.
usage: python scripts/fuzz_server.py [seconds] [seed]"""
import os
import random
import socket
import subprocess
import sys
import tempfile
import time
import urllib.error
import urllib.parse
import urllib.request

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "bgt")
MINE = os.path.join(ROOT, "bgt_amd", "bin", "bgt-server")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bgt_amd", "host")])
refdir = os.path.join(ROOT, "oracle", "_ref")
REF = os.path.join(tempfile.mkdtemp(), "bgt-server-ref")
subprocess.check_call(["gcc", "-O1", "-DBGS_REFERENCE_LIB", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "bgt_amd", "host", "server.c"),
                       "-o", REF, "-L", refdir, "-l:libbgt_ref.so", "-Wl,-rpath," + refdir, "-lz", "-lm", "-lpthread"])

REGIONS = ["11", "12", "11:1000-1100", "11:1050-1051", "11:1,100-1,300", "12:500-510", "11:1101", "13", "11:1-999", "12:503", "zz"]
SAMPLES = ['pop=="X"', 'pop=="Y"', 'pop=="Z"', "idx%5==0", "idx<10", "idx>=30", ",A001,A010,A049", ",B002,B039", ",A000,A001,A002,A003,A004,B000,B001", 'idx%7==3.or.pop=="X"', "pop=="]
FILTERS = ["AC>0", "AC==0", "AN>90", "AC/AN>0.2", "(AC1>0.and.AC2==0)", "AC1/AN1>=0.1&&AC2<5", "AC3>0", "AC>1.AND.AC<10", "AC%2==1", "AN-AC>80", "AC>"]
TABLES = ["CHROM,POS,AC,AN", "POS,REF,ALT,END", "AC/AN,AC1,AN1", "POS,(AC+1)*2,AC//3", "CHROM,POS,AC2,AC3"]
ALLELES = [",11:1010:1:A", ",11:1010:1:A,11:1010:1:C", ",11:1060:1:G,11:1040:1:G", ",11:1100:CAG:C,12:500:CAG:C",
           ",11:1060::C", ",11:1020:1:T,11:1030:1:C,11:1050:1:A", ",13:5:1:A", "impact>=2", "cadd>10.5", 'gene=="ABC"', "impact>=99"]
DBS = [["synA"], ["synB"], ["synA", "synB"], ["synB", "synA"], ["ex2"], ["ex3"],
       ["mgsA"], ["mgsA", "mgsB"], ["mgsB", "synA"], ["mgsZ"], ["mgsZ", "mgsA"]]      # `_mgs:i:` tags: tests/golden/make_mgs_golden.py
enc = lambda v: urllib.parse.quote(v, safe=rnd.choice(["", "(),:=<>/*%"]) if "%" not in v else "")


def make():
    dbs = rnd.choice(DBS)
    syn = dbs[0][0] in "sm"
    opts, q = [], []
    if rnd.random() < 0.3:
        opts += ["-m", str(rnd.choice([50, 1500, 20000]))]
    if rnd.random() < 0.15:
        opts += ["-g", str(rnd.choice([1, 3, 12]))]
    if rnd.random() < 0.3:
        q.append("g")
    if rnd.random() < 0.4:
        q.append("C")
    if rnd.random() < 0.35:
        q.append("r=" + enc(rnd.choice(REGIONS)))
    if rnd.random() < 0.2:
        q.append("i=" + str(rnd.choice([0, 1, 2, 17, 35, "x"])))
    if rnd.random() < 0.25:
        q.append("n=" + str(rnd.choice([0, 1, 5, 12, 1000, "q"])))
    n_grp = 0
    if syn:
        n_grp = rnd.choice([0, 0, 1, 2, 3])
        for _ in range(n_grp):
            q.append("s=" + enc(rnd.choice(SAMPLES)))
    grp_ok = lambda e: n_grp >= 2 or not any(v in e for v in ("AC1", "AN1", "AC2", "AC3"))   # (uninitialised in the reference otherwise)
    if rnd.random() < 0.4:
        f = rnd.choice([f for f in FILTERS if grp_ok(f)])
        q.append("f=" + (f if "&&" in f else enc(f)))                       # a literal && must survive the parameter split
    if rnd.random() < 0.15:
        q.append("t=" + enc(rnd.choice([t for t in TABLES if grp_ok(t)])))
    if syn and rnd.random() < 0.35:
        al = rnd.choice(ALLELES)
        if al[0] != ",":
            opts += ["-d", "vardb.fmf"]
        q.append("a=" + enc(al))
        r = rnd.random()
        q += ["S"] if r < 0.3 else ["H"] if r < 0.6 else ["S", "H"] if r < 0.7 else []
    rnd.shuffle(q)
    return dbs, opts, "&".join(q)


def one_shot(exe, dbs, opts, query):
    p = subprocess.run([exe] + opts + ["-q", query] + dbs, cwd=GOLD, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    return p.returncode, p.stdout


s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
srv = subprocess.Popen([MINE, "-p", str(port), "synA", "synB"], cwd=GOLD, stderr=subprocess.DEVNULL)
for _ in range(600):
    try:
        socket.create_connection(("127.0.0.1", port), timeout=1).close()
        break
    except OSError:
        time.sleep(0.1)
t_end = time.time() + budget
n = bad = n_http = 0
while time.time() < t_end:
    dbs, opts, query = make()
    mine, ref = one_shot(MINE, dbs, opts, query), one_shot(REF, dbs, opts, query)
    n += 1
    if mine != ref:
        bad += 1
        print("DIFF", dbs, opts, repr(query), mine[0], ref[0], len(mine[1]), len(ref[1]))
    if n % 20 == 0 or (dbs == ["synA", "synB"] and not opts and query):
        dbs2, _, q2 = (dbs, opts, query) if dbs == ["synA", "synB"] and not opts and query else (["synA", "synB"], [], "C&r=11:1000-1100")
        want = one_shot(MINE, dbs2, [], q2)
        try:
            body = urllib.request.urlopen("http://127.0.0.1:%d/?%s" % (port, q2), timeout=120).read()
        except urllib.error.HTTPError as e:
            body = e.read()
        n_http += 1
        if body != want[1]:
            bad += 1
            print("HTTP DIFF", repr(q2), len(body), len(want[1]))
# concurrency: sixteen clients draw from a set of queries with known answers and hit the resident server at once
import threading
known = {}
while len(known) < 40:
    dbs, opts, query = make()
    if dbs == ["synA", "synB"] and not opts and query:
        known[query] = one_shot(MINE, dbs, [], query)[1]
keys = sorted(known)
conc_bad = []


def client(k):
    r = random.Random(1000 + k)
    for _ in range(60):
        q = r.choice(keys)
        try:
            body = urllib.request.urlopen("http://127.0.0.1:%d/?%s" % (port, q), timeout=120).read()
        except urllib.error.HTTPError as e:
            body = e.read()
        if body != known[q]:
            conc_bad.append(q)


th = [threading.Thread(target=client, args=(k,)) for k in range(16)]
for t in th:
    t.start()
for t in th:
    t.join()
for q in conc_bad[:5]:
    print("CONCURRENT DIFF", repr(q))
bad += len(conc_bad)
print("concurrent phase: 16 clients x 60 queries, %d wrong answers" % len(conc_bad))
srv.terminate()
srv.wait()
print("server fuzz: %d queries (%d also over HTTP), %d differences" % (n, n_http, bad))
sys.exit(1 if bad else 0)
