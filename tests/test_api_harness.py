"""The reader API driven from C, as a caller of the reference drives it (tests/integration/api_dump.c): the same harness
source is linked once with bgt_amd/lib/libbgt.so and once with the compiled reference (oracle/_ref/libbgt_ref.so); the two
print-outs must be identical.  Covers bgt_read() (reference bgt.c:347-356: row number, the site record as the file holds
it, GT vector) and the per-query call sequence of bgt-server.go:220-373."""
import os
import subprocess

import pytest

from conftest import require_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "bgt")
SRC = os.path.join(ROOT, "tests", "integration", "api_dump.c")
BGT = os.path.join(ROOT, "bgt_amd", "bin", "bgt")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    import bgt_amd
    bgt_amd.build_library()
    __import__("bgt_amd").build_host_shell()
    require_ref("libbgt_ref.so")
    d = tmp_path_factory.mktemp("api")
    mine, ref = str(d / "api_mine"), str(d / "api_ref")
    inc = ["-I", os.path.join(ROOT, "include")]
    lib = os.path.join(ROOT, "bgt_amd", "lib")
    subprocess.check_call(["gcc", "-O1", "-Wall"] + inc + [SRC, "-o", mine, "-L", lib, "-lbgt", "-Wl,-rpath," + lib])
    refdir = os.path.join(ROOT, "oracle", "_ref")
    subprocess.check_call(["gcc", "-O1", "-Wall"] + inc + [SRC, "-o", ref, "-L", refdir, "-l:libbgt_ref.so",
                           "-Wl,-rpath," + refdir, "-lz", "-lm", "-lpthread"])
    return mine, ref


def both(harness, args, cwd=GOLD):
    out = []
    for exe in harness:
        p = subprocess.run([exe] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, (exe, args, p.stderr.decode()[-400:])
        out.append(p.stdout)
    return out


def test_server_sequence_without_genotype_work(harness):
    """no -C / -s / -f: nothing depends on a genotype, the device is never opened (runs on the CPU box too)"""
    for args in (["server", "1000000", "synA", "synB", "--"], ["server", "300", "synA", "--", "-r", "11:1000-1100"],
                 ["server", "1000000", "synB", "synA", "--", "-i", "5", "-n", "7"]):
        mine, ref = both(harness, args)
        assert mine == ref and len(mine) > 100, args


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["read", "synA"], ["read", "synB", "11:1,035-1,120"], ["read", "synA", "-", "7"],
                                  ["read", "ex2"], ["read", "ex3"], ["read", "synA", "12"], ["read", "synA", "13"]])
def test_bgt_read_to_eof_like_the_reference(harness, args):
    mine, ref = both(harness, args)
    assert mine == ref, args
    assert b"end=-1" in mine


@pytest.mark.gpu
def test_bgt_read_on_a_multi_block_database(harness, tmp_path):
    db = str(tmp_path / "db")
    subprocess.check_call([BGT, "synth", db, "300", "20000", "7"])
    for args in (["read", db], ["read", db, "11:90000-95000"], ["read", db, "-", "16380"]):
        mine, ref = both(harness, args, cwd=str(tmp_path))
        assert mine == ref and mine.count(b"\nrow=") > 5, args


@pytest.mark.gpu
@pytest.mark.parametrize("args", [
    ["server", "100000", "synA", "synB", "--", "-s", "pop==\"X\"", "-s", "pop==\"Y\"", "-f", "AC1>0"],
    ["server", "1500", "synA", "synB", "--", "-C"],                                   # stops on the n_gt_read cap
    ["server", "100000", "synA", "--", "-g", "-C", "-r", "11:1000-1200", "-n", "9"],
    ["server", "100000", "synB", "synA", "--", "-g", "-s", "idx<10", "-i", "4"],
    ["server", "100000", "synA", "synB", "--", "-s", "pop==\"X\"", "-t", "CHROM,POS,REF,ALT,AC1,AN1"],
    ["server", "100000", "synA", "--", "-S", "-a", ",11:1000:C:G,11:1030:TAG:T"],
    ["server", "100000", "synA", "--", "-H", "-a", ",11:1000:C:G,11:1030:TAG:T", "-s", "pop==\"X\"", "-s", "pop==\"Y\""],
    # with genotypes on the harness also prints bm->a[0] / bm->a[1] after every bgtm_read: the merged site's two byte planes
    # (bgt.h:70, bgt.c:829-842) -- two databases with sites only one of them has (the code-2 fill), and `_mgs` tags that hide
    # samples from the record (bgt.c:290-313) but not from the planes
    ["server", "100000", "synA", "synB", "--", "-g"],
    ["server", "100000", "synB", "synA", "--", "-g", "-s", "pop==\"X\"", "-s", "pop==\"Y\"", "-f", "AC1>0"],
    ["server", "100000", "mgsA", "--", "-g", "-C"],
    ["server", "100000", "mgsA", "mgsB", "--", "-g", "-s", "idx<12"],
    ["server", "100000", "mgsZ", "mgsA", "--", "-g", "-C", "-r", "11:1000-1100"],
])
def test_server_call_sequence_like_the_reference(harness, args):
    mine, ref = both(harness, args)
    assert mine == ref, args
    assert mine.count(b"\n") > 3
