"""End-to-end drop-in check at the shape of BASELINE.json configs[0] (2,504 samples x 50,000 sites): the
synthetic database written by this repo's generator is read by BOTH the compiled reference
(oracle/_ref/bgt, CPU) and this repo's `bgt view` (MI355X); stdout must be byte-identical."""
import hashlib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BGT = os.path.join(ROOT, "bgt_amd", "bin", "bgt")
REF = os.path.join(ROOT, "oracle", "_ref", "bgt")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def reference_binary_present():
    from conftest import require_ref
    require_ref("bgt")
    require_ref("pbfview")


@pytest.fixture(scope="module")
def c1(tmp_path_factory):
    import bgt_amd
    bgt_amd.build_library()
    __import__("bgt_amd").build_host_shell()
    d = tmp_path_factory.mktemp("c1")
    prefix = str(d / "c1")
    subprocess.check_call([BGT, "synth", prefix, "2504", "50000", "1"])
    for ext in ("pbf", "bcf", "bcf.csi", "spl"):
        assert os.path.getsize(prefix + "." + ext) > 0
    return prefix


def md5_of(cmd):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    return p.returncode, hashlib.md5(p.stdout).hexdigest(), len(p.stdout), p.stderr.decode()[-300:]


@pytest.mark.parametrize("args", [["-C", "-G"], ["-G", "-f", "AC>0"], ["-G", "-f", "AC>0", "-r", "11:20000-300000"],
                                  ["-C", "-G", "-i", "40000", "-n", "5000"],
                                  ["-G", "-s", "pop==\"A\"", "-s", "pop==\"B\"", "-f", "AC1>0&&AC2==0"],
                                  ["-C", "-G", "-s", "idx%20==0"], ["-C", "-r", "11:100000-101000", "-s", "idx<40"],
                                  ["-G", "-s", "pop==\"A\"", "-s", "pop==\"C\"", "-f", "AC1/AN1>=0.1&&AC2<5"],
                                  ["-C", "-r", "11:1000-1500"], ["-bCG", "-n", "3000"],
                                  ["-b", "-i", "100", "-n", "2500"], ["-u", "-n", "700", "-s", "idx<500"],
                                  ["-b", "-l", "9", "-n", "1500", "-C"]])
def test_same_bytes_as_reference_binary(c1, args):
    mine = md5_of([BGT, "view"] + args + [c1])
    ref = md5_of([REF, "view"] + args + [c1])
    assert mine[0] == ref[0] == 0, (mine, ref)
    assert mine[2] > 0 and mine[1:3] == ref[1:3], (args, mine, ref)


def test_generated_pbf_is_what_the_reference_encoder_writes(c1, tmp_path):
    """decode the synthetic .pbf with the reference (pbfview) and re-encode it with the reference: same bytes,
    i.e. the PBWT-domain generator + GPU checkpoints produce a canonical file."""
    pbfview = os.path.join(ROOT, "oracle", "_ref", "pbfview")
    small = str(tmp_path / "small")
    subprocess.check_call([BGT, "synth", small, "300", "20000", "7"])
    pim = tmp_path / "x.pim"
    with open(pim, "wb") as f:
        subprocess.check_call([pbfview, small + ".pbf"], stdout=f)
    re = tmp_path / "re.pbf"
    with open(re, "wb") as f:
        subprocess.check_call([pbfview, "-Sb", str(pim)], stdout=f)
    assert open(re, "rb").read() == open(small + ".pbf", "rb").read()


def test_two_database_group_join_like_config5(tmp_path):
    """Shape of BASELINE.json configs[4]: two databases over the same positions with independent alleles, two
    sample groups, `-f'AC1>0&&AC2==0'`; plus the variants with genotypes and with three groups."""
    import bgt_amd
    bgt_amd.build_library()
    __import__("bgt_amd").build_host_shell()
    a, b = str(tmp_path / "dba"), str(tmp_path / "dbb")
    subprocess.check_call([BGT, "synth", a, "5000", "30000", "5"])
    subprocess.check_call([BGT, "synth", b, "4000", "30000", "6"])
    for args in (["-G", "-s", "pop==\"A\"", "-s", "pop==\"B\"", "-f", "AC1>0&&AC2==0"],
                 ["-G", "-C"], ["-G", "-s", "idx<100", "-s", "idx>=100&&idx<300", "-s", "pop==\"C\"", "-f", "AC3>AC1"],
                 ["-s", "idx%1000==0", "-r", "11:100000-140000", "-C"]):
        mine = md5_of([BGT, "view"] + args + [a, b])
        ref = md5_of([REF, "view"] + args + [a, b])
        assert mine[0] == ref[0] == 0, (mine, ref)
        assert mine[2] > 0 and mine[1:3] == ref[1:3], (args, mine, ref)


def test_tables_bed_and_allele_sets_like_reference(c1, tmp_path):
    """-t tables, -B/-e BED filters and -a/-S allele sets on the C1-shaped database; the allele strings are taken
    from sites the reference itself reports (every 997th site with AC > 20), in REF:ALT and in rlen:ALT form."""
    bed = tmp_path / "r.bed"
    bed.write_text("".join("11\t%d\t%d\n" % (1000 + 977 * k, 1000 + 977 * k + 300 * (k % 7)) for k in range(1, 400)) +
                   "11\t250000\n12\t1\t1000\n")
    sites = subprocess.run([REF, "view", "-G", "-f", "AC>20", "-t", "CHROM,POS,REF,ALT,AC", c1], stdout=subprocess.PIPE,
                           check=True).stdout.decode().splitlines()
    assert len(sites) > 5000
    picks = [s.split("\t") for s in sites[::997]][:12]
    als = ",".join(["%s:%s:%s:%s" % (c, p, r, a) for c, p, r, a, _ in picks[:6]] +
                   ["%s:%s:%d:%s" % (c, p, len(r), a) for c, p, r, a, _ in picks[6:]])
    alfile = tmp_path / "alleles.txt"
    alfile.write_text("\n".join("%s:%s:%s:%s\tnote" % (c, p, r, a) for c, p, r, a, _ in picks[:3]) + "\n")
    for args in (["-G", "-t", "CHROM,POS,END,REF,ALT,AC,AN,AC/AN,AC//3,AN%11", "-f", "AC>0", "-n", "20000"],
                 ["-G", "-s", "pop==\"A\"", "-s", "pop==\"B\"", "-t", "POS,AC1,AN1,AC2,AN2,AC1/AN1-AC2/AN2,AC3"],
                 ["-CG", "-B", str(bed)], ["-CG", "-B", str(bed), "-e", "-r", "11:1000-200000"],
                 ["-G", "-a", "," + als, "-C"], ["-a", str(alfile), "-s", "idx<50"],
                 ["-S", "-a", "," + ",".join(als.split(",")[:2])], ["-S", "-a", str(alfile), "-s", "pop==\"A\""],
                 ["-S", "-a", ",%s:%s::%s" % (picks[0][0], picks[0][1], picks[0][2])],
                 ["-H", "-a", "," + ",".join(als.split(",")[:7])],                       # > 16 distinct haplotypes, many ties
                 ["-H", "-a", "," + als, "-s", "pop==\"A\"", "-s", "pop==\"B\"", "-s", "idx%3==0"],
                 ["-H", "-S", "-a", str(alfile)]):
        mine = md5_of([BGT, "view"] + args + [c1])
        ref = md5_of([REF, "view"] + args + [c1])
        assert mine[0] == ref[0] == 0, (args, mine, ref)
        assert mine[1:3] == ref[1:3], (args, mine, ref)


def test_wide_cohort_subset_like_config3(tmp_path):
    """Shape of BASELINE.json configs[2] at reduced length: 100,000 samples (m = 200,000: the team-mode kernels, row
    index, sub-checkpoints), `-s` picks every 20th sample (5,000 of 100,000), AC/AN; plus a region with genotypes of
    a few samples.  The reference decodes 200,000-wide rows on the CPU: ~0.5 ms per site."""
    import bgt_amd
    bgt_amd.build_library()
    __import__("bgt_amd").build_host_shell()
    db = str(tmp_path / "c3")
    subprocess.check_call([BGT, "synth", db, "100000", "12000", "3"], timeout=600)
    for args in (["-G", "-C", "-s", "idx%20==0", "-n", "9000"],
                 ["-G", "-s", "idx%20==0", "-s", "idx%20==1", "-f", "AC1>0&&AC2==0", "-i", "8000"],
                 ["-C", "-s", "idx%25000==7", "-r", "11:90000-95000"],
                 ["-G", "-C", "-i", "8100", "-n", "1500"],                       # the whole 200,000-column cohort
                 ["-G", "-f", "AC/AN>0.3", "-r", "11:60000-70000"]):
        mine = md5_of([BGT, "view"] + args + [db])
        ref = md5_of([REF, "view"] + args + [db])
        assert mine[0] == ref[0] == 0, (args, mine, ref)
        assert mine[2] > 0 and mine[1:3] == ref[1:3], (args, mine, ref)


def test_one_shot_walk_of_a_wide_cohort(tmp_path, monkeypatch):
    """`bgt view -G -f` over a whole wide database opens its image with BGTH_OPEN_HINT=walk: no sub-checkpoints, a directory
    arena of one round of workgroups walked in several passes, the strings streamed to HBM while the host parses.  The
    hint only applies when file blocks x column slices fill the chip (one C4 shard: 153 x 4); BGTH_OPEN_HINT_MIN lowers that
    bar so that 100,000 samples x 12,000 sites (2 blocks) take the same code path.  Same bytes as the compiled reference, as
    without the hint, and as with an arena too small for one pass."""
    import bgt_amd
    bgt_amd.build_library()
    __import__("bgt_amd").build_host_shell()
    db = str(tmp_path / "w")
    subprocess.check_call([BGT, "synth", db, "100000", "12000", "4"], timeout=600)
    args = ["-G", "-f", "AC>0"]
    ref = md5_of([REF, "view"] + args + [db])
    plain = md5_of([BGT, "view"] + args + [db])
    monkeypatch.setenv("BGTH_OPEN_HINT_MIN", "1")
    monkeypatch.setenv("BGTH_TRACE", "1")
    hinted = md5_of([BGT, "view"] + args + [db])
    err = subprocess.run([BGT, "view"] + args + [db], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300).stderr.decode()
    import re                                                     # (a short image takes more, narrower column slices to fill the chip)
    assert re.search(r"one-shot walk: 2 file blocks x \d+ column slices", err) and "directory arena" in err, err
    monkeypatch.setenv("BGTH_DIR_ARENA_MB", "1000")                   # 8192-row units of 100 KB rows: one unit a pass
    small = md5_of([BGT, "view"] + args + [db])
    assert ref[0] == plain[0] == hinted[0] == small[0] == 0, (ref, plain, hinted, small)
    assert ref[2] > 0 and plain[1:3] == ref[1:3] and hinted[1:3] == ref[1:3] and small[1:3] == ref[1:3], (ref, plain, hinted, small)

