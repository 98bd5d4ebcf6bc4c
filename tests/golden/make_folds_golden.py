#!/usr/bin/env python3
"""Fixture: what the COMPILED REFERENCE leaves in bgtm_t::alcnt / ::hap (reference bgt.c:859-876, the `-S` / `-H` reductions) and
prints from them, for allele sets over the golden databases -- tests/integration/api_dump.c (the call sequence of
bgt-server.go) linked with oracle/_ref/libbgt_ref.so.  Writes tests/golden/folds.json = [{"args": [...], "stdout": "..."}].
Run in the build container (needs oracle/_ref): python tests/golden/make_folds_golden.py"""
import json
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
GOLD = os.path.join(HERE, "bgt")
REFDIR = os.path.join(ROOT, "oracle", "_ref")

CASES = [
    ["server", "100000000", "synA", "--", "-S", "-a", ",11:1010:1:A,11:1050:1:C"],
    ["server", "100000000", "synA", "--", "-H", "-a", ",11:1010:1:A,11:1010:1:C,11:1050:1:A,11:1060:1:G,11:1020:1:T,11:1030:1:C,11:1040:1:G"],
    ["server", "100000000", "synA", "--", "-S", "-H", "-a", ",11:1010:T:A,11:1060:C:G,12:500:CAG:C,11:1100:CAG:C"],
    ["server", "100000000", "synA", "--", "-S", "-a", ",11:1060::C"],                         # a reference-allele query
    ["server", "100000000", "synA", "--", "-S", "-H", "-a", ":11:1010:T:A,11:1100:CAG:C", "-s", "idx<25"],
    ["server", "100000000", "synA", "--", "-H", "-a", ",11:1060:1:G,11:1040:1:G,11:1080:1:G", "-s", ",A049,A003,A010,A001"],   # samples in any order
    ["server", "100000000", "synA", "synB", "--", "-S", "-H", "-a", ",11:1060:1:G,11:1040:1:G,11:1080:1:G", "-s", 'pop=="X"', "-s", 'pop=="Y"', "-s", "idx<10"],
    ["server", "100000000", "synB", "synA", "--", "-S", "-a", ",11:1000:C:G,11:1030:TAG:T,12:500:CAG:C"],
    ["server", "100000000", "synA", "--", "-S", "-H", "-a", ",13:5:1:A,nonsense"],            # nothing matches
]


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "api_ref")
        subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "integration", "api_dump.c"),
                               "-o", exe, "-L", REFDIR, "-l:libbgt_ref.so", "-Wl,-rpath," + REFDIR, "-lz", "-lm", "-lpthread"])
        out = []
        for args in CASES:
            p = subprocess.run([exe] + args, cwd=GOLD, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
            out.append({"args": args, "stdout": p.stdout.decode()})
    with open(os.path.join(HERE, "folds.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote %d cases" % len(out))


if __name__ == "__main__":
    main()
