#!/usr/bin/env python3
"""Adds the `bgt view -t` (tabular output, reference bgt.c:547-593,775-795, view.c:43,120,153) goldens: runs the
COMPILED REFERENCE (oracle/_ref/bgt) on the committed trios of tests/golden/bgt and stores stdout + rc.
Run in the build container only:  python tests/golden/make_table_golden.py"""
import hashlib
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BGT = os.path.join(ROOT, "oracle", "_ref", "bgt")
X, Y = 'pop=="X"', 'pop=="Y"'
TABLE_CMDS = {
    "ex2_t": (["-t", "CHROM,POS,END,REF,ALT,AC,AN"], ["ex2"]),
    "ex3_t_G": (["-G", "-t", "POS,AC,AN,AC/AN"], ["ex3"]),
    "synA_t_ratio": (["-G", "-t", "CHROM,POS,AC/AN,AC//2,AN%7,1.5*AC", "-f", "AC>0"], ["synA"]),
    "synA_t_grp": (["-G", "-s", X, "-s", Y, "-t", "POS,AC1,AN1,AC2/AN2,AC3"], ["synA"]),
    "synA_t_str": (["-G", "-t", 'REF=="A",ALT,(AC+1)*2,abs(AC-AN),END-POS'], ["synA"]),
    "synA_t_region": (["-t", "CHROM,POS,AC", "-r", "11:1000-1100", "-n", "5"], ["synA"]),
    "synAB_t": (["-t", "CHROM,POS,REF,ALT,AC,AN"], ["synA", "synB"]),
    "synA_t_paren": (["-G", "-t", "POS,(AC,AN)"], ["synA"]),
    "synA_t_bad": (["-G", "-t", "AC+"], ["synA"]),
    # -B / -e: BED overlap (reference bedidx.c); regions.bed and points.bed are fixtures of this repo
    "synA_bed": (["-G", "-B", "regions.bed"], ["synA"]),
    "synA_bed_excl": (["-CG", "-B", "regions.bed", "-e"], ["synA"]),
    "synA_bed_points_gt": (["-B", "points.bed", "-s", 'pop=="X"'], ["synA"]),
    "synAB_bed_f": (["-G", "-B", "regions.bed", "-f", "AC>2"], ["synA", "synB"]),
    # -a / -S: allele sets (reference bgt.c:477-544, :843-876, :957-970); alleles.txt is a fixture of this repo
    "synA_al": (["-a", ",11:1010:1:A,11:1050:1:C"], ["synA"]),
    "synA_al_G": (["-G", "-C", "-a", "alleles.txt"], ["synA"]),
    "synA_al_S": (["-S", "-a", "alleles.txt"], ["synA"]),
    "synA_al_S2": (["-S", "-a", ":11:1010:T:A,11:1100:CAG:C"], ["synA"]),
    "synA_al_S1": (["-S", "-a", ",11:1010:1:C", "-s", "idx<25"], ["synA"]),
    "synAB_al_S": (["-S", "-a", ",11:1060:1:G,11:1040:1:G", "-s", 'pop=="X"', "-s", 'pop=="Y"'], ["synA", "synB"]),
    "synA_al_refquery": (["-S", "-a", ",11:1060::C"], ["synA"]),
    # -H: haplotype counts over the allele set (reference bgt.c:896-955)
    "synA_hap": (["-H", "-a", "alleles.txt"], ["synA"]),
    "synA_hap2": (["-H", "-a", ",11:1010:1:A,11:1010:1:C,11:1050:1:A,11:1060:1:G,11:1020:1:T,11:1030:1:C,11:1040:1:G"], ["synA"]),
    "synAB_hap_grp": (["-H", "-S", "-a", ",11:1060:1:G,11:1040:1:G,11:1080:1:G", "-s", 'pop=="X"', "-s", 'pop=="Y"', "-s", "idx<10"],
                      ["synA", "synB"]),
    # -d / -M: alleles selected from a variant annotation file (vardb.fmf, a fixture of this repo) by expression
    "synA_db_impact": (["-G", "-C", "-d", "vardb.fmf", "-a", "impact>=2"], ["synA"]),
    "synA_db_mem": (["-G", "-C", "-d", "vardb.fmf", "-M", "-a", "impact>=2"], ["synA"]),
    "synA_db_real": (["-G", "-C", "-d", "vardb.fmf", "-a", "cadd>10.5"], ["synA"]),
    "synA_db_real_mem": (["-G", "-C", "-d", "vardb.fmf", "-M", "-a", "cadd>10.5"], ["synA"]),
    "synA_db_gene_S": (["-S", "-d", "vardb.fmf", "-a", 'gene=="ABC"'], ["synA"]),
    "synA_db_row_H": (["-H", "-d", "vardb.fmf", "-M", "-a", '_ROW_!="13:7:1:A"&&impact<9'], ["synA"]),
    "synA_db_bad": (["-G", "-d", "vardb.fmf", "-a", "impact>="], ["synA"]),
    "synA_al_none": (["-G", "-a", ",13:5:1:A,nonsense"], ["synA"]),
    "synA_bed_t": (["-B", "points.bed", "-e", "-t", "CHROM,POS,END,AC", "-n", "6"], ["synA"]),
}

out_dir = os.path.join(HERE, "bgt")
exp = os.path.join(out_dir, "expected")
manifest = json.load(open(os.path.join(out_dir, "manifest.json")))
for name, (args, prefixes) in TABLE_CMDS.items():
    res = subprocess.run([BGT, "view"] + args + prefixes, cwd=out_dir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    open(os.path.join(exp, name + ".out"), "wb").write(res.stdout)
    manifest["views"][name] = {"args": args, "prefixes": prefixes, "rc": res.returncode,
                               "md5": hashlib.md5(res.stdout).hexdigest(), "bytes": len(res.stdout)}
    print("view %-16s rc=%d %6d B  %s" % (name, res.returncode, len(res.stdout), res.stderr.decode()[:80].strip()))
json.dump(manifest, open(os.path.join(out_dir, "manifest.json"), "w"), indent=1, sort_keys=True)
