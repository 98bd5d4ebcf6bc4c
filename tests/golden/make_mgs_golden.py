#!/usr/bin/env python3
"""Adds the `_mgs` goldens (minimal group size: the per-sample privacy mask of the reference, bgt.c:24-38 the tag,
:151-152 naming a sample in a list, :290-313 genotype columns dropped, :610-653 the header's sample list and BGT_F_NO_GT,
:678-688 bgtm_test_mgs, :964 the `SP` lines of -S).  Runs the COMPILED REFERENCE (oracle/_ref/bgt) and stores stdout + rc.

Trios written here (genotypes and sites are those of synA / synB: the .pbf / .bcf / .bcf.csi are copies, only the .spl differs):
  mgsA   synA with `_mgs:i:` cycling over 0, 1, 2, 5, <absent> by sample index (a real-valued and a negative tag once each:
         both count as unset, bgt.c:35)
  mgsB   synB with `_mgs:i:` 2 on every third sample
  mgsZ   synA with `_mgs:i:3` on EVERY sample: no sample may be shown, the header has no FORMAT column (bgt.c:621-623)
Run in the build container only:  python tests/golden/make_mgs_golden.py"""
import hashlib
import json
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BGT = os.path.join(ROOT, "oracle", "_ref", "bgt")
X, Y = 'pop=="X"', 'pop=="Y"'
NAMES5 = ",A000,A001,A002,A003,A004,A005,A006,A007,A008,A009"      # two of each tag value

MGS_CMDS = {
    "mgsA_view": ([], ["mgsA"]),
    "mgsA_C": (["-C"], ["mgsA"]),
    "mgsA_CG": (["-CG"], ["mgsA"]),
    "mgsA_bcf": (["-b", "-C"], ["mgsA"]),
    "mgsA_region": (["-C", "-r", "11:1000-1100"], ["mgsA"]),
    "mgsA_grp": (["-s", X, "-s", Y], ["mgsA"]),                      # expressions select whatever the tag says
    "mgsA_grp_f": (["-s", X, "-s", Y, "-f", "AC1>0&&AC2==0"], ["mgsA"]),
    "mgsA_names": (["-s", NAMES5], ["mgsA"]),                        # a list names only samples whose tag allows it
    "mgsA_names_C": (["-C", "-s", NAMES5, "-s", "idx>=40"], ["mgsA"]),
    "mgsA_names_only_hidden": (["-C", "-s", ",A002,A003,A007"], ["mgsA"]),   # nobody qualifies: an empty group
    "mgsA_expr_hidden": (["-C", "-s", "idx==2||idx==3||idx==7"], ["mgsA"]),  # selected and counted, never shown
    "mgsA_al_S": (["-S", "-a", "alleles.txt"], ["mgsA"]),            # SP lines skip masked samples (bgt.c:964)
    "mgsA_al_S2": (["-S", "-a", ":11:1010:T:A,11:1100:CAG:C", "-s", "idx<25"], ["mgsA"]),
    "mgsA_al_S1": (["-S", "-a", ",11:1010:1:A"], ["mgsA"]),
    "mgsAB_al_S1": (["-S", "-a", ",11:1060:1:G", "-s", X, "-s", Y], ["mgsA", "mgsB"]),
    "mgsA_hap": (["-H", "-a", "alleles.txt"], ["mgsA"]),
    "mgsA_hap_S_grp": (["-H", "-S", "-a", ",11:1060:1:G,11:1040:1:G", "-s", X, "-s", Y], ["mgsA"]),
    "mgsA_t": (["-t", "CHROM,POS,AC,AN", "-s", NAMES5], ["mgsA"]),
    "mgsAB_view": ([], ["mgsA", "mgsB"]),
    "mgsAB_grp3": (["-s", X, "-s", Y, "-s", "idx<10", "-C"], ["mgsA", "mgsB"]),
    "mgsBA_f": (["-f", "AN>150&&AC>3"], ["mgsB", "mgsA"]),
    "mgsAB_names": (["-C", "-s", ",A001,A002,B002,B003,B039"], ["mgsA", "mgsB"]),
    "mgsA_synB": (["-C"], ["mgsA", "synB"]),                          # one masked database beside an open one
    "mgsZ_view": ([], ["mgsZ"]),                                      # nobody may be shown: no FORMAT, no genotypes
    "mgsZ_C": (["-C"], ["mgsZ"]),
    "mgsZ_grp": (["-s", X, "-s", Y, "-f", "AC1>0"], ["mgsZ"]),
    "mgsZ_S": (["-S", "-a", "alleles.txt"], ["mgsZ"]),
    "mgsZA": (["-C"], ["mgsZ", "mgsA"]),
}


def tag_for_A(i):
    v = (0, 1, 2, 5, None)[i % 5]
    if i == 44:
        return "\t_mgs:r:2.5"                                        # not FMF_INT: unset
    if i == 49:
        return "\t_mgs:i:-3"                                         # negative: unset
    return "" if v is None else "\t_mgs:i:%d" % v


def write_trio(out_dir, name, src, tag):
    for ext in ("pbf", "bcf", "bcf.csi"):
        shutil.copyfile(os.path.join(out_dir, src + "." + ext), os.path.join(out_dir, name + "." + ext))
    rows = open(os.path.join(out_dir, src + ".spl")).read().splitlines()
    with open(os.path.join(out_dir, name + ".spl"), "w") as f:
        for i, row in enumerate(rows):
            f.write(row + tag(i) + "\n")


def main():
    out_dir = os.path.join(HERE, "bgt")
    exp = os.path.join(out_dir, "expected")
    write_trio(out_dir, "mgsA", "synA", tag_for_A)
    write_trio(out_dir, "mgsB", "synB", lambda i: "\t_mgs:i:2" if i % 3 == 0 else "")
    write_trio(out_dir, "mgsZ", "synA", lambda i: "\t_mgs:i:3")
    manifest = json.load(open(os.path.join(out_dir, "manifest.json")))
    for name in ("mgsA", "mgsB", "mgsZ"):
        manifest["inputs"][name] = {k: hashlib.md5(open(os.path.join(out_dir, name + "." + k), "rb").read()).hexdigest()
                                    for k in ("pbf", "bcf", "bcf.csi", "spl")}
    for name, (args, prefixes) in MGS_CMDS.items():
        res = subprocess.run([BGT, "view"] + args + prefixes, cwd=out_dir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        open(os.path.join(exp, name + ".out"), "wb").write(res.stdout)
        manifest["views"][name] = {"args": args, "prefixes": prefixes, "rc": res.returncode,
                                   "md5": hashlib.md5(res.stdout).hexdigest(), "bytes": len(res.stdout)}
        print("view %-24s rc=%d %6d B  %s" % (name, res.returncode, len(res.stdout), res.stderr.decode()[:80].strip()))
    json.dump(manifest, open(os.path.join(out_dir, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
