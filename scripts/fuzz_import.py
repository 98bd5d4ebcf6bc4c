#!/usr/bin/env python3
"""Differential fuzz of `bgt import` on the GPU box: random VCFs (multi-allelic records, CIGARs, symbolic alleles, repeated
and overlapping positions, missing alleles, filters, three contigs) through this repo's importer and the compiled
reference's; the three output files must be identical byte for byte, also from gzip and BCF input and with -F.
usage: python scripts/fuzz_import.py [seconds] [seed]"""
import gzip
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_import import random_vcf  # noqa: E402

BGT = os.path.join(ROOT, "bgt_amd", "bin", "bgt")
REF = os.path.join(ROOT, "oracle", "_ref", "bgt")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bgt_amd", "host")])
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(seed0)
t_end = time.time() + budget
n = bad = 0
while time.time() < t_end:
    seed = int(rng.integers(1, 1 << 30))
    n_samples = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 64, 100, 257, 700]))
    n_records = int(rng.choice([1, 5, 60, 400, 2000, 9000])) if n_samples < 200 else int(rng.choice([5, 200, 1200]))
    vcf = os.path.join(tmp, "in.vcf")
    random_vcf(np.random.default_rng(seed), n_samples, n_records, vcf)
    opts = ["-F"] if rng.random() < 0.4 else []
    src, in_opts = vcf, ["-S"]
    form = rng.random()
    if form < 0.2:                                   # gzip'ed text
        src = vcf + ".gz"
        with open(vcf, "rb") as f, gzip.open(src, "wb") as g:
            g.write(f.read())
                                                     # (BCF input is covered by tests/test_import.py; text and gzip here)
    mine, want = os.path.join(tmp, "mine"), os.path.join(tmp, "want")
    r1 = subprocess.run([BGT, "import"] + in_opts + opts + [mine, src], stderr=subprocess.PIPE, timeout=600)
    r2 = subprocess.run([REF, "import"] + in_opts + opts + [want, src], stderr=subprocess.PIPE, timeout=600)
    n += 1
    same = r1.returncode == r2.returncode
    if same and r1.returncode == 0:
        for ext in ("spl", "pbf", "bcf"):
            if open(mine + "." + ext, "rb").read() != open(want + "." + ext, "rb").read():
                same = False
                print("DIFF in .%s" % ext)
    if not same:
        bad += 1
        print("DIFF seed=%d samples=%d records=%d opts=%s src=%s rc=%d/%d" % (seed, n_samples, n_records, opts, os.path.basename(src), r1.returncode, r2.returncode))
print("import fuzz: %d VCFs, %d differences" % (n, bad))
sys.exit(1 if bad else 0)
