#!/usr/bin/env python3
"""Regenerate the golden fixtures from the COMPILED REFERENCE (oracle/_ref, built by oracle/Makefile from
/root/reference).  Run in the build container only:  python tests/golden/make_golden.py

Writes (all small, committed):
  tests/golden/codec.npz        PBF images written by the reference's pbf_open_w/pbf_write/pbf_close for
                                the seeded matrices of scenarios.py, and the byte planes returned by the
                                reference's pbf_subset/pbf_seek/pbf_read for every scenario.
  tests/golden/ex1.pbf          `pbfview -Sb ex1.pim` (ex1.pim is a data file of the reference).
  tests/golden/bgt/*            BGT trios written by `bgt import` + expected `bgt view ...` outputs.
Only DATA is stored: inputs, and outputs of the reference run on them.
"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, HERE)
import scenarios  # noqa: E402

from make_golden_lib import ref_write_pbf, ref_replay  # noqa: E402


def codec_goldens(tmp):
    store = {}
    for name, (mat, shift) in scenarios.cases().items():
        path = os.path.join(tmp, name + ".pbf")
        ref_write_pbf(path, mat, shift)
        data = open(path, "rb").read()
        store[name + "/pbf"] = np.frombuffer(data, np.uint8)
        store[name + "/shift"] = np.int32(shift)
        rows, m = mat.shape
        for i, ops in enumerate(scenarios.scenarios(name, rows, m, shift)):
            res = ref_replay(path, m, 2, ops)
            if name == "longrun":      # keep the fixture small: packed bits instead of byte planes
                store["%s/s%d" % (name, i)] = np.packbits(res, axis=-1)
                store["%s/s%d_w" % (name, i)] = np.int32(res.shape[-1])
            else:
                store["%s/s%d" % (name, i)] = res
        print("codec case %-8s m=%d rows=%d shift=%d pbf=%d B md5=%s" %
              (name, m, rows, shift, len(data), hashlib.md5(data).hexdigest()))
    np.savez_compressed(os.path.join(HERE, "codec.npz"), **store)
    # the reference's own toy fixture through its own CLI
    ex1_path = os.path.join(tmp, "ex1_cli.pbf")      # stdout must be a real file: the footer uses ftell()
    with open(ex1_path, "wb") as f:
        subprocess.check_call([os.path.join(REF, "pbfview"), "-Sb", "/root/reference/ex1.pim"], stdout=f)
    ex1 = open(ex1_path, "rb").read()
    assert hashlib.md5(ex1).hexdigest() == "ffeac837ea3d039ec92a2da801901bd5"
    assert ex1 == bytes(store["ex1/pbf"])
    open(os.path.join(HERE, "ex1.pbf"), "wb").write(ex1)


# ------------------------------------------------------------------------------------------------------
# API-level goldens: `bgt import` a few tiny VCFs, run `bgt view` variants
# ------------------------------------------------------------------------------------------------------
def synth_vcf(rng, n_samples, positions, prefix_name, multi_every=4):
    lines = ["##fileformat=VCFv4.1",
             '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
             "##contig=<ID=11,length=135006516>", "##contig=<ID=12,length=133851895>",
             "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" +
             "\t".join("%s%03d" % (prefix_name, i) for i in range(n_samples))]
    nuc = "ACGT"
    for k, (chrom, pos) in enumerate(positions):
        r = nuc[(pos * 7 + 1) % 4]
        alts = [nuc[(pos * 7 + 2) % 4]]
        if k % multi_every == 1:
            alts.append(nuc[(pos * 7 + 3) % 4])
        if k % 7 == 3:                                   # an indel so that END / rlen paths are hit
            r, alts = r + "AG", [r]
        na = len(alts)
        freq = rng.random() * 0.5
        gts = []
        for _ in range(n_samples):
            al = []
            for _h in range(2):
                x = rng.random()
                if x < 0.03:
                    al.append(".")
                elif x < 0.03 + freq:
                    al.append(str(int(rng.integers(1, na + 1))))
                else:
                    al.append("0")
            gts.append(al[0] + ("|" if rng.random() < 0.5 else "/") + al[1])
        lines.append("\t".join([chrom, str(pos), ".", r, ",".join(alts), "50", "PASS", ".", "GT"] + gts))
    return "\n".join(lines) + "\n"


VIEW_CMDS = {
    # name: (args before prefixes, prefixes)
    "ex2_view": ([], ["ex2"]),
    "ex2_C": (["-C"], ["ex2"]),
    "ex2_G_f": (["-G", "-f", "AC>0"], ["ex2"]),
    "ex3_view": ([], ["ex3"]),
    "ex3_C": (["-C"], ["ex3"]),
    "synA_view": ([], ["synA"]),
    "synA_CG": (["-C", "-G"], ["synA"]),
    "synA_G_f": (["-G", "-f", "AC>0"], ["synA"]),
    "synA_G": (["-G"], ["synA"]),
    "synA_sub": (["-s", ",A003,A010,A011,A049", "-f", "AC>0"], ["synA"]),
    "synA_sub_C": (["-C", "-s", "idx%5==0"], ["synA"]),
    "synA_grp": (["-G", "-s", "pop==\"X\"", "-s", "pop==\"Y\"", "-f", "AC1>0&&AC2==0"], ["synA"]),
    "synA_grp_gt": (["-s", "pop==\"X\"", "-s", "pop==\"Y\""], ["synA"]),
    "synA_grp_ratio": (["-G", "-s", "pop==\"X\"", "-s", "pop==\"Z\"", "-f", "AC1/AN1>=0.1&&AC2<5"], ["synA"]),
    "synA_region": (["-C", "-r", "11:1000-1200"], ["synA"]),
    "synA_region12": (["-CG", "-r", "12"], ["synA"]),
    "synA_i_n": (["-CG", "-i", "5", "-n", "7"], ["synA"]),
    "synA_unbound": (["-G", "-f", "AC3>0"], ["synA"]),
    "synAB_view": ([], ["synA", "synB"]),
    "synAB_CG": (["-CG"], ["synA", "synB"]),
    "synAB_grp": (["-G", "-s", "pop==\"X\"", "-s", "pop==\"Y\"", "-f", "AC1>0&&AC2==0"], ["synA", "synB"]),
    "synAB_grp3": (["-G", "-s", "pop==\"X\"", "-s", "pop==\"Y\"", "-s", "idx<10"], ["synA", "synB"]),
    "synAB_sub_gt": (["-s", ",A001,B002,B039", "-C"], ["synA", "synB"]),
    "synBA_f": (["-f", "AN>150&&AC>3", "-G"], ["synB", "synA"]),
    "synA_bcf": (["-bG", "-C"], ["synA"]),
}


def bgt_goldens(tmp):
    out_dir = os.path.join(HERE, "bgt")
    os.makedirs(out_dir, exist_ok=True)
    bgt = os.path.join(REF, "bgt")
    rng = np.random.default_rng(7)
    posA = [("11", 1000 + 10 * i) for i in range(24)] + [("12", 500 + 3 * i) for i in range(6)]
    posB = [("11", 1000 + 10 * i) for i in range(4, 30, 2)] + [("11", 1003), ("11", 1500)] + \
           [("12", 500 + 3 * i) for i in range(3, 9)]
    posB = sorted(posB, key=lambda x: (x[0], x[1]))
    inputs = {
        "ex2": open("/root/reference/ex2.vcf").read(),        # data files of the reference
        "ex3": open("/root/reference/ex3.vcf").read(),
        "synA": synth_vcf(rng, 50, posA, "A"),
        "synB": synth_vcf(rng, 40, posB, "B", multi_every=3),
    }
    manifest = {"inputs": {}, "views": {}}
    for name, text in inputs.items():
        vcf = os.path.join(tmp, name + ".vcf")
        open(vcf, "w").write(text)
        open(os.path.join(out_dir, name + ".vcf"), "w").write(text)
        subprocess.check_call([bgt, "import", "-S", os.path.join(out_dir, name), vcf])
        if name.startswith("syn"):                       # add metadata to the sample file
            spl = open(os.path.join(out_dir, name + ".spl")).read().split()
            with open(os.path.join(out_dir, name + ".spl"), "w") as f:
                for i, s in enumerate(spl):
                    f.write("%s\tpop:Z:%s\tidx:i:%d\n" % (s, "XYZ"[i % 3], i))
        manifest["inputs"][name] = {
            k: hashlib.md5(open(os.path.join(out_dir, name + "." + k), "rb").read()).hexdigest()
            for k in ("pbf", "bcf", "bcf.csi", "spl")}
    exp = os.path.join(out_dir, "expected")
    os.makedirs(exp, exist_ok=True)
    for name, (args, prefixes) in VIEW_CMDS.items():
        res = subprocess.run([bgt, "view"] + args + prefixes, cwd=out_dir, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE)
        open(os.path.join(exp, name + ".out"), "wb").write(res.stdout)
        manifest["views"][name] = {"args": args, "prefixes": prefixes, "rc": res.returncode,
                                   "md5": hashlib.md5(res.stdout).hexdigest(), "bytes": len(res.stdout)}
        print("view %-16s rc=%d %6d B" % (name, res.returncode, len(res.stdout)))
    json.dump(manifest, open(os.path.join(out_dir, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as tmp:
        codec_goldens(tmp)
        bgt_goldens(tmp)
