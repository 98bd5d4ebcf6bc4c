#!/usr/bin/env python3
"""Differential fuzz of `bgt view`: random option combinations on the golden databases (tests/golden/bgt), this
repo's CLI next to the compiled reference (oracle/_ref/bgt): same exit status and the same bytes on stdout.
usage (GPU box): python scripts/fuzz_cli.py [seconds] [seed]"""
import os
import random
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "bgt")
MINE = os.path.join(ROOT, "bgt_amd", "bin", "bgt")
REF = os.path.join(ROOT, "oracle", "_ref", "bgt")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bgt_amd", "host")])

SYNTH = len(sys.argv) > 3 and sys.argv[3] == "synth"        # third argument: fuzz two generated multi-block databases instead
REGIONS = ["11", "12", "11:1000-1100", "11:1050-1051", "11:1,100-1,300", "12:500-510", "11:1101", "13", "11:1-999", "12:503"]
SAMPLES = ['pop=="X"', 'pop=="Y"', 'pop=="Z"', "idx%5==0", "idx<10", "idx>=30", ",A001,A010,A049", ",B002,B039", ":A003,B003", ",A000,A001,A002,A003,A004,B000,B001", "idx%7==3||pop==\"X\""]
FILTERS = ["AC>0", "AC==0", "AN>90", "AC/AN>0.2", "AC1>0&&AC2==0", "AC1/AN1>=0.1&&AC2<5", "AC3>0", "AC>1&&AC<10", "AC%2==1", "AN-AC>80"]
TABLES = ["CHROM,POS,AC,AN", "POS,REF,ALT,END", "AC/AN,AC1,AN1", "POS,(AC+1)*2,AC//3", "CHROM,POS,AC2,AC3"]
ALLELES = [",11:1010:1:A", ",11:1010:1:A,11:1010:1:C", ",11:1060:1:G,11:1040:1:G", "alleles.txt", ",11:1100:CAG:C,12:500:CAG:C",
           ",11:1060::C", ",11:1020:1:T,11:1030:1:C,11:1050:1:A", ",13:5:1:A"]
DBS = [["synA"], ["synB"], ["synA", "synB"], ["synB", "synA"], ["ex2"], ["ex3"],
       ["mgsA"], ["mgsA", "mgsB"], ["mgsB", "synA"], ["mgsZ"], ["mgsZ", "mgsA"]]      # `_mgs:i:` tags: tests/golden/make_mgs_golden.py


if SYNTH:
    import tempfile
    TMP = tempfile.mkdtemp()
    GOLD = TMP
    subprocess.check_call([MINE, "synth", os.path.join(TMP, "s1"), "300", "20000", "7"], stdout=subprocess.DEVNULL)
    subprocess.check_call([MINE, "synth", os.path.join(TMP, "s2"), "200", "20000", "8"], stdout=subprocess.DEVNULL)
    for f in ("regions.bed", "points.bed"):
        with open(os.path.join(TMP, f), "w") as fh:
            for k in range(40):
                b = rnd.randint(1000, 200000)
                fh.write("11\t%d\t%d\n" % (b, b + rnd.choice([1, 5, 50, 3000])) if f[0] == "r" else "11\t%d\n" % b)
    REGIONS = ["11"] + ["11:%d-%d" % (b, b + w) for b, w in ((rnd.randint(1000, 200000), rnd.choice([1, 30, 500, 9000, 90000])) for _ in range(30))]
    SAMPLES = ['pop=="A"', 'pop=="B"', 'pop=="C"', "idx%5==0", "idx<10", "idx>=150", "idx%7==3||pop==\"A\""]
    sites = subprocess.run([REF, "view", "-G", "-t", "CHROM,POS,REF,ALT", os.path.join(TMP, "s1")], stdout=subprocess.PIPE,
                           check=True).stdout.decode().split("\n")[:-1]
    pick = [x.split("\t") for x in rnd.sample(sites, 60)]
    ALLELES = ["," + ",".join("%s:%s:%s:%s" % tuple(x) for x in rnd.sample(pick, rnd.randint(1, 6))) for _ in range(25)]
    open(os.path.join(TMP, "alleles.txt"), "w").write("\n".join("%s:%s:%s:%s" % tuple(x) for x in pick[:4]) + "\n")
    ALLELES.append("alleles.txt")
    open(os.path.join(TMP, "vardb.fmf"), "w").write("".join("%s:%s:%s:%s\tgene:Z:%s\tcadd:f:%.1f\timpact:i:%d\n" %
                                                    (x[0], x[1], x[2], x[3], rnd.choice(["ABC", "XYZ"]), rnd.random() * 30, rnd.randint(0, 5)) for x in pick))
    DBS = [["s1"], ["s2"], ["s1", "s2"], ["s2", "s1"]]


def make():
    a = []
    dbs = rnd.choice(DBS)
    syn = dbs[0][0] in "sm"
    if rnd.random() < 0.5:
        a += ["-G"]
    if rnd.random() < 0.4:
        a += ["-C"]
    if rnd.random() < 0.35:
        a += ["-r", rnd.choice(REGIONS)]
    if rnd.random() < 0.2:
        a += ["-i", str(rnd.randint(1, 25000 if SYNTH else 35))]
    if rnd.random() < 0.2:
        a += ["-n", str(rnd.randint(0, 3000 if SYNTH else 12))]
    n_grp = 0
    if syn:
        n_grp = rnd.choice([0, 0, 1, 2, 3])
        for _ in range(n_grp):
            a += ["-s", rnd.choice(SAMPLES)]
    # AC1/AN1/... with fewer than two groups read uninitialised memory in the reference (bgtm_cal_info fills the
    # per-group counts only for n_groups > 1, bgt.c:735-757, but bgtm_assign_expr binds them anyway): not comparable
    grp_ok = lambda e: n_grp >= 2 or not any(v in e for v in ("AC1", "AN1", "AC2", "AC3"))
    if rnd.random() < 0.4:
        a += ["-f", rnd.choice([f for f in FILTERS if grp_ok(f)])]
    mode = rnd.random()
    if mode < 0.12:
        a += ["-t", rnd.choice([t for t in TABLES if grp_ok(t)])]
    elif mode < 0.22:
        a += ["-b"] + (["-l", str(rnd.randint(0, 9))] if rnd.random() < 0.3 else [])
    elif mode < 0.26:
        a += ["-u"]
    if syn and rnd.random() < 0.2:
        a += ["-B", rnd.choice(["regions.bed", "points.bed"])] + (["-e"] if rnd.random() < 0.4 else [])
    if syn and rnd.random() < 0.3:
        a += ["-a", rnd.choice(ALLELES)]
        r = rnd.random()
        if r < 0.3:
            a += ["-S"]
        elif r < 0.6:
            a += ["-H"]
        elif r < 0.7:
            a += ["-S", "-H"]
    if syn and rnd.random() < 0.08:
        a = [x for x in a if x not in ("-S", "-H")]
        if "-a" in a:
            k = a.index("-a"); del a[k:k + 2]
        a += ["-d", "vardb.fmf", "-a", rnd.choice(["impact>=2", "cadd>10.5", 'gene=="ABC"'])] + (["-M"] if rnd.random() < 0.5 else [])
    return a, dbs


t_end = time.time() + budget
n = bad = 0
while time.time() < t_end:
    args, dbs = make()
    out = []
    for exe in (MINE, REF):
        try:
            p = subprocess.run([exe, "view"] + args + dbs, cwd=GOLD, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
            out.append((p.returncode, p.stdout))
        except subprocess.TimeoutExpired:
            out.append(("timeout", b""))
    n += 1
    ok = out[0] == out[1]
    if not ok and isinstance(out[1][0], int) and out[1][0] < 0:
        ok = out[0][0] != 0            # the reference crashed (signal): any failure of ours counts as agreement
    if not ok:
        bad += 1
        print("DIFF rc %s vs %s, %d vs %d bytes: view %s %s" % (out[0][0], out[1][0], len(out[0][1]), len(out[1][1]),
                                                               " ".join("'%s'" % x for x in args), " ".join(dbs)), flush=True)
        if bad >= 10:
            break
print("cli fuzz: %d commands, %d differences" % (n, bad))
sys.exit(1 if bad else 0)
