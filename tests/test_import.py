"""`bgt import`: VCF -> .pbf / .bcf / .bcf.csi / .spl with the atomizer on the host and the PBWT encoder
on the MI355X.  (1) the committed inputs of the golden trios re-imported: .pbf and .bcf byte-identical to what the
compiled reference wrote (the .spl of the syn* goldens carries metadata added afterwards: names compared); (2) random
VCFs -- multi-allelic sites, MNPs, insertions, deletions, INFO/CIGAR, overlapping records, repeated alleles, filtered
records, missing calls, several contigs -- imported by BOTH this build and the reference binary: every output file byte
for byte, the .csi through region queries of both front ends on both indexes."""
import os
import subprocess

import numpy as np
import pytest

from conftest import require_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "bgt")
BGT = os.path.join(ROOT, "bgt_amd", "bin", "bgt")


@pytest.fixture(scope="module", autouse=True)
def built():
    import bgt_amd
    bgt_amd.build_library()
    __import__("bgt_amd").build_host_shell()


def test_import_usage_and_bad_input(tmp_path):
    p = subprocess.run([BGT, "import"], stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"Usage: bgt import" in p.stderr
    assert subprocess.run([BGT, "import", "-S", str(tmp_path / "x"), str(tmp_path / "nosuch.vcf")], stderr=subprocess.PIPE).returncode != 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ex2", "ex3", "synA", "synB"])
def test_golden_inputs_reimport_to_the_reference_bytes(tmp_path, name):
    out = str(tmp_path / name)
    subprocess.check_call([BGT, "import", "-S", out, os.path.join(GOLD, name + ".vcf")], timeout=300)
    for ext in ("pbf", "bcf"):
        assert open(out + "." + ext, "rb").read() == open(os.path.join(GOLD, name + "." + ext), "rb").read(), ext
    mine = open(out + ".spl").read().split("\n")
    gold = [line.split("\t")[0] for line in open(os.path.join(GOLD, name + ".spl")).read().split("\n")]
    assert mine == gold
    # the index this build writes serves region queries of both front ends
    ref = require_ref("bgt")
    for exe in (BGT, ref):
        a = subprocess.run([exe, "view", "-G", "-r", "11:1000-1100", out], stdout=subprocess.PIPE, check=True).stdout
        b = subprocess.run([exe, "view", "-G", "-r", "11:1000-1100", os.path.join(GOLD, name)], stdout=subprocess.PIPE, check=True).stdout
        assert a == b


def random_vcf(rng, n_samples, n_records, path):
    bases = "ACGT"
    lines = ["##fileformat=VCFv4.1", '##INFO=<ID=CIGAR,Number=A,Type=String,Description="">',
             '##INFO=<ID=DP,Number=1,Type=Integer,Description="">', '##FILTER=<ID=q10,Description="">',
             '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">', '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="">',
             "##contig=<ID=11,length=135006516>", "##contig=<ID=12,length=133851895>", "##contig=<ID=X,length=155270560>",
             "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("S%03d" % i for i in range(n_samples))]
    pos, ctg = 100, 0
    for _ in range(n_records):
        if rng.random() < 0.01 and ctg < 2:
            ctg += 1; pos = 50
        pos += int(rng.integers(0, 6))                                  # 0: same position as the record before (overlaps, repeats)
        l_ref = int(rng.choice([1, 1, 1, 2, 3, 5]))
        ref = "".join(rng.choice(list(bases), l_ref))
        alts, cig = [], []
        for _ in range(int(rng.choice([1, 1, 2, 3]))):
            kind = rng.integers(0, 5)
            if kind == 0 or l_ref == 1 and kind in (2,):                  # substitution(s) over the whole REF
                alt = "".join(bases[(bases.index(c) + int(rng.integers(1, 4))) % 4] if rng.random() < 0.7 else c for c in ref)
                if alt == ref:
                    alt = bases[(bases.index(ref[0]) + 1) % 4] + ref[1:]
                c = "%dM" % l_ref
            elif kind == 1:                                               # insertion after the first base
                ins = "".join(rng.choice(list(bases), int(rng.integers(1, 4))))
                alt = ref[0] + ins + ref[1:]
                c = "1M%dI" % len(ins) + ("%dM" % (l_ref - 1) if l_ref > 1 else "")
            elif kind == 2:                                               # deletion keeping the first base
                d = int(rng.integers(1, l_ref))
                alt = ref[0] + ref[1 + d:]
                c = "1M%dD" % d + ("%dM" % (l_ref - 1 - d) if l_ref - 1 - d > 0 else "")
            elif kind == 3:
                alt, c = "<DEL>", None
            else:                                                         # one substituted base somewhere
                k = int(rng.integers(0, l_ref))
                alt = ref[:k] + bases[(bases.index(ref[k]) + 2) % 4] + ref[k + 1:]
                c = "%dM" % l_ref
            if alt not in alts or rng.random() < 0.1:
                alts.append(alt); cig.append(c)
        use_cigar = rng.random() < 0.3 and all(c is not None for c in cig)
        info = "DP=%d" % rng.integers(1, 99) + (";CIGAR=" + ",".join(cig) if use_cigar else "")
        flt = rng.choice(["PASS", "PASS", ".", "q10"])
        gts = []
        for _ in range(n_samples):
            a = [str(x) if x >= 0 else "." for x in rng.integers(-1, len(alts) + 1, 2)]
            gts.append(a[0] + rng.choice(["/", "|"]) + a[1] + (":%d" % rng.integers(0, 50)))
        lines.append("\t".join([["11", "12", "X"][ctg], str(pos), ".", ref, ",".join(alts), "%d" % rng.integers(1, 99), flt, info, "GT:DP"] + gts))
    open(path, "w").write("\n".join(lines) + "\n")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_samples,n_records,opts", [(1, 7, 400, []), (2, 40, 1500, ["-F"]), (3, 3, 9000, []), (4, 300, 600, ["-F"]),
                                                           (5, 60, 9500, ["-1"]), (6, 11, 300, ["-1", "-F"])])
def test_random_vcfs_import_like_the_reference_binary(tmp_path, seed, n_samples, n_records, opts):
    ref = require_ref("bgt")
    vcf = str(tmp_path / "in.vcf")
    random_vcf(np.random.default_rng(seed), n_samples, n_records, vcf)
    mine, want = str(tmp_path / "mine"), str(tmp_path / "want")
    subprocess.check_call([BGT, "import", "-S"] + opts + [mine, vcf], timeout=600, stderr=subprocess.DEVNULL)
    subprocess.check_call([ref, "import", "-S"] + opts + [want, vcf], timeout=600, stderr=subprocess.DEVNULL)
    for ext in ("spl", "pbf", "bcf") + (("pb1",) if "-1" in opts else ()):   # -1: the one-plane file (import.c:72-74, 98-101)
        assert open(mine + "." + ext, "rb").read() == open(want + "." + ext, "rb").read(), ext
    if "-1" in opts:                                                    # and the one-plane file decodes (as a two-plane image whose
        ref_pv = require_ref("pbfview")                                  # second plane is empty) like the reference's codec tool reads it
        a = subprocess.run([BGT, "pbfview", mine + ".pb1"], stdout=subprocess.PIPE, check=True).stdout
        b = subprocess.run([ref_pv, want + ".pb1"], stdout=subprocess.PIPE, check=True).stdout
        assert a == b and a.count(b"\n") > 100
    for reg in ("11:100-400", "12", "X:60-90", "11:1,000-1,200"):       # either front end, either index: the same sites
        outs = [subprocess.run([exe, "view", "-C", "-r", reg, db], stdout=subprocess.PIPE, check=True).stdout
                for exe in (BGT, ref) for db in (mine, want)]
        assert outs[0] == outs[1] == outs[2] == outs[3], reg


@pytest.mark.gpu
def test_bcf_input_imports_like_the_reference_binary(tmp_path):
    """binary input: a BCF with genotypes (written by the reference's `view -b` from a database) imported by both builds"""
    ref = require_ref("bgt")
    vcf = str(tmp_path / "in.vcf")
    random_vcf(np.random.default_rng(9), 25, 700, vcf)
    db = str(tmp_path / "db")
    subprocess.check_call([ref, "import", "-S", "-F", db, vcf], stderr=subprocess.DEVNULL)
    bcf = str(tmp_path / "geno.bcf")
    with open(bcf, "wb") as f:
        subprocess.check_call([ref, "view", "-b", db], stdout=f)                # sites x 25 samples, GT as int8 pairs
    mine, want = str(tmp_path / "mine"), str(tmp_path / "want")
    subprocess.check_call([BGT, "import", mine, bcf], timeout=600, stderr=subprocess.DEVNULL)
    subprocess.check_call([ref, "import", want, bcf], timeout=600, stderr=subprocess.DEVNULL)
    for ext in ("spl", "pbf", "bcf"):
        assert open(mine + "." + ext, "rb").read() == open(want + "." + ext, "rb").read(), ext
    a = subprocess.run([BGT, "view", "-C", mine], stdout=subprocess.PIPE, check=True).stdout
    b = subprocess.run([ref, "view", "-C", want], stdout=subprocess.PIPE, check=True).stdout
    assert a == b and a.count(b"\n") > 500


@pytest.mark.gpu
@pytest.mark.timeout(2400)
def test_biobank_width_through_both_command_lines(tmp_path):
    """400,000 samples = 800,000 haplotypes (round 6: beyond every LDS width -- the writer's directories and the reader's rows live
    in memory) through the COMMAND LINES: one VCF of 60 records (multi-allelic sites, an indel, missing calls) imported by this
    repo's `bgt import` (device encoder) and by the compiled reference's; the .pbf / .bcf must be the same bytes.  Then `bgt view`
    of both on this repo's database: counts of the whole cohort, a filter, two sample groups, a sparse subset WITH genotype
    columns, a region -- every stdout byte for byte (reference pbwt.c:92-105 / 199-219 take any width; bgt.c as it is)."""
    ref = require_ref("bgt")
    n, n_rec = 400000, 60
    rng = np.random.default_rng(400000)
    vcf = str(tmp_path / "wide.vcf")
    gts = np.array([b"0|0", b"0|1", b"1|0", b"1|1", b"./.", b"0|2", b"2|1", b".|1"], dtype=object)
    founder = rng.integers(0, 16, n)                                   # columns that share a founder share most calls: runs for the PBWT
    with open(vcf, "wb") as f:
        f.write(b"##fileformat=VCFv4.1\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n##contig=<ID=11,length=135006516>\n")
        f.write(b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + b"\t".join(b"S%06d" % i for i in range(n)) + b"\n")
        for r in range(n_rec):
            multi = r % 5 == 2 and r % 7 != 3
            per_founder = rng.integers(0, 4, 16) if rng.random() < 0.7 else np.zeros(16, np.int64)
            call = per_founder[founder]
            noise = rng.random(n)
            call = np.where(noise < 0.002, 4, call)                    # missing
            if multi:
                call = np.where((noise > 0.002) & (noise < 0.01), rng.integers(5, 8, n), call)
            ref_al, alt = (b"CAG", b"C") if r % 7 == 3 else (b"A", b"G,T" if multi else b"G")
            f.write(b"11\t%d\t.\t%s\t%s\t50\tPASS\t.\tGT\t" % (1000 + 10 * r, ref_al, alt) + b"\t".join(gts[call]) + b"\n")
    mine, want = str(tmp_path / "mine"), str(tmp_path / "want")
    for exe, out in ((BGT, mine), (ref, want)):
        p = subprocess.run([exe, "import", "-S", out, vcf], timeout=1200, stderr=subprocess.PIPE)
        assert p.returncode == 0, (exe, p.stderr.decode()[-600:])
    for ext in ("pbf", "bcf", "spl"):
        assert open(mine + "." + ext, "rb").read() == open(want + "." + ext, "rb").read(), ext
    with open(mine + ".spl", "w") as f:                                # metadata for the expressions below
        for i in range(n):
            f.write("S%06d\tpop:Z:%s\tidx:i:%d\n" % (i, "XYZ"[i % 3], i))
    import hashlib
    for args in (["-C", "-G"], ["-G", "-f", "AC>100"], ["-G", "-s", 'pop=="X"', "-s", 'pop=="Y"', "-f", "AC1>0&&AC2>0"],
                 ["-C", "-s", "idx%40000==7"], ["-s", ",S000001,S399999,S123456", "-r", "11:1000-1200"]):
        out = [subprocess.run([exe, "view"] + args + [mine], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200) for exe in (BGT, ref)]
        assert out[0].returncode == out[1].returncode == 0, (args, out[0].stderr.decode()[-300:], out[1].stderr.decode()[-300:])
        assert len(out[0].stdout) > 500 and hashlib.md5(out[0].stdout).hexdigest() == hashlib.md5(out[1].stdout).hexdigest(), args
