"""`bgt pbfview` (bgt_amd/host/pbfview_cli.c): the reference's codec-level tool (pbfview.c) on the device codec, run side by
side with the compiled reference tool (oracle/_ref/pbfview) on random matrices: PIM -> PBF, PBF -> PIM, column subsets in
any order, seeks inside and across checkpoint blocks and behind the end, PBF -> PBF recoding; one to eight bit planes (pbfview.c
is g-agnostic: VERDICT r5 missing #3)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import require_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BGT = os.path.join(ROOT, "bgt_amd", "bin", "bgt")


@pytest.fixture(scope="module", autouse=True)
def built():
    import bgt_amd
    bgt_amd.build_library()
    __import__("bgt_amd").build_host_shell()


def run(cmd):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=30)
    return p.returncode, p.stdout


def run_to_file(cmd, path):
    """PBF output goes to a FILE: the reference writes ftell() into the footer, which is -1 on a pipe"""
    with open(path, "wb") as fh:
        rc = subprocess.run(cmd, stdout=fh, stderr=subprocess.PIPE, timeout=30).returncode
    return rc, open(path, "rb").read()


def test_usage():
    p = subprocess.run([BGT, "pbfview"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"Usage: pbfview" in p.stderr


def write_pim(path, mat, g):
    with open(path, "w") as f:
        f.write("PIM1 %d %d\n" % (mat.shape[1], g))
        for row in mat:
            f.write(" ".join(str(int(x)) for x in row) + "\n")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,m,rows,g,shift", [(1, 4, 9, 2, 13), (2, 70, 40, 2, 3), (3, 333, 150, 2, 4), (4, 64, 33, 1, 3),
                                                 (5, 2000, 70, 2, 5), (6, 1, 20, 2, 2), (7, 700, 90, 1, 4),
                                                 (8, 50, 40, 3, 3), (9, 333, 70, 5, 4), (10, 70, 30, 8, 2), (11, 1000, 50, 4, 3)])
def test_side_by_side_with_the_reference_tool(tmp_path, seed, m, rows, g, shift):
    ref = require_ref("pbfview")
    rng = np.random.default_rng(seed)
    mat = rng.integers(0, 1 << g, (rows, m))
    mat[rng.random((rows, m)) < 0.6] = 0                               # runs
    if rows > 8:
        mat[3] = 0; mat[5] = (1 << g) - 1; mat[7, : m // 2] = 1       # all-zero, all-ones, one long run
    pim = str(tmp_path / "in.pim")
    write_pim(pim, mat, g)
    # PIM -> PBF
    pbf, o1, o2 = str(tmp_path / "x.pbf"), str(tmp_path / "o1.pbf"), str(tmp_path / "o2.pbf")
    mine, want = run_to_file([BGT, "pbfview", "-S", "-b", "-s", str(shift), pim], o1), run_to_file([ref, "-S", "-b", "-s", str(shift), pim], pbf)
    assert mine == want and mine[0] == 0 and len(mine[1]) > 16
    # PIM -> PIM (the echo) and PBF -> PIM
    assert run([BGT, "pbfview", "-S", pim]) == run([ref, "-S", pim])
    mine, want = run([BGT, "pbfview", pbf]), run([ref, pbf])
    assert mine == want                                                # (any g: a file of more than two planes opens as a bundle of plane pairs)
    # (with ONE column the reference's PIM reader notices the end of the file a read late and writes the last value once
    #  more as an extra row; the encoder here reads the same way, so the files agree and only this check looks at `rows`)
    assert mine[1].decode().split("\n")[1:-1][:rows] == [" ".join(str(int(x)) for x in row) for row in mat]
    # seeks: inside a block, at and across checkpoints, the last row, behind the end; with and without -n
    blk = 1 << shift
    # (a row BEHIND the end inside the forward window of 1 << shift rows makes the reference spin on pbf_read: not run)
    for r0 in sorted(r for r in {1, blk - 1, blk, blk + 1, 2 * blk + 3, rows - 1, rows, rows + blk + 1} if 0 < r <= rows or r > max(rows, blk)):
        for n in (None, 1, 3):
            args = ["-r", str(r0)] + (["-n", str(n)] if n else [])
            assert run([BGT, "pbfview"] + args + [pbf]) == run([ref] + args + [pbf]), args
    # column subsets, any order, with a seek; and all columns named
    for k in sorted({1, 2, min(5, m), m}):
        cols = rng.permutation(m)[:k]
        args = sum((["-c", str(int(c))] for c in cols), [])
        assert run([BGT, "pbfview"] + args + [pbf]) == run([ref] + args + [pbf]), args
        assert run([BGT, "pbfview", "-r", str(min(rows - 1, blk + 2))] + args + [pbf]) == run([ref, "-r", str(min(rows - 1, blk + 2))] + args + [pbf]), args
    # PBF -> PBF, whole and subset
    assert run_to_file([BGT, "pbfview", "-b", "-s", str(shift), pbf], o1) == run_to_file([ref, "-b", "-s", str(shift), pbf], o2)
    if m > 3:
        args = ["-c", "2", "-c", "0", "-c", str(m - 1), "-b", "-s", "2"]
        assert run_to_file([BGT, "pbfview"] + args + [pbf], o1) == run_to_file([ref] + args + [pbf], o2)
