"""BASELINE.json configs[1] and configs[2] at their FULL length, FROM ROW 0, against the compiled reference and the CPU oracle.

The deep windows of test_full_size.py re-base the oracle onto ranks the device image holds (bench.oracle_window): past the first
blocks nothing there starts from the identity order.  Here the whole database is decoded sequentially on the CPU, from the
identity order of row 0 through every one of its 123 checkpoint blocks, twice over:

  * the COMPILED REFERENCE (oracle/_ref/bgt) runs the metric's own command line on the database file this repo's generator
    wrote; stdout must be this repo's `bgt view` byte for byte (md5 over ~50 MB of VCF text: every AN / AC of every site);
  * the CPU oracle (oracle/liborc.so: pbwt.c:69-170 + bgt.c:735-757 restated) scans rows [0, n) of the same file in one
    sequential pass and its int32 counts must equal the device's -- the image opened from the FILE, and the image bench.py
    times (built from the generator's strings, bgt_amd.synth_rows + HipPbf.from_rle): the benchmark's own launch.

C2: 10,000 samples x 1,000,000 sites, whole cohort, `-G -f'AC>0'` (reference: ~23 s, oracle: ~22 s).
C3: 100,000 samples x 1,000,000 sites, `-s 'idx%20==0'` (5,000 samples; reference ~38 s, oracle ~40 s)."""
import hashlib
import os
import subprocess
import time

import numpy as np
import pytest

import orc
from conftest import require_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BGT = os.path.join(ROOT, "bgt_amd", "bin", "bgt")

pytestmark = pytest.mark.gpu


def md5_stdout(cmd, timeout):
    """md5 + size of a command's stdout, streamed (the VCF text of 1,000,000 sites need not sit in memory twice)"""
    h, n = hashlib.md5(), 0
    t0 = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        while True:
            blk = p.stdout.read(1 << 22)
            if not blk:
                break
            h.update(blk)
            n += len(blk)
            assert time.time() - t0 < timeout, "timeout: " + " ".join(cmd)
        rc = p.wait(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    return rc, h.hexdigest(), n, p.stderr.read().decode()[-300:]


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name,n_samples,sites,seed,every,view_args", [
    ("C2", 10000, 1000000, 2, 0, ["-G", "-f", "AC>0"]),
    ("C3", 100000, 1000000, 3, 20, ["-G", "-f", "AC>0", "-s", "idx%20==0"]),
])
def test_full_length_from_row_zero(name, n_samples, sites, seed, every, view_args, tmp_path):
    import bgt_amd
    bgt_amd.build_library()
    bgt_amd.build_host_shell()
    ref = require_ref("bgt")
    m = 2 * n_samples
    prefix = str(tmp_path / name.lower())
    subprocess.check_call([BGT, "synth", prefix, str(n_samples), str(sites), str(seed)], timeout=900)

    # (1) the command line: compiled reference against this repo, every byte of stdout
    r = md5_stdout([ref, "view"] + view_args + [prefix], 900)
    mine = md5_stdout([BGT, "view"] + view_args + [prefix], 600)
    assert r[0] == mine[0] == 0, (r, mine)
    assert r[2] > 30 * sites and r[1:3] == mine[1:3], (name, r, mine)

    # (2) the counts: one sequential oracle pass over the FILE from the identity order of row 0 ...
    sel = np.arange(0, n_samples, every) if every else None
    cols = np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1).astype(np.int32) if every else None
    ora = orc.Pbf(np.fromfile(prefix + ".pbf", np.uint8))
    assert (ora.m, ora.n, ora.shift) == (m, sites, 13)
    if cols is not None:
        ora.subset(cols)
    want = ora.scan(0, sites).reshape(sites, 1, 3)
    ora.close()
    # ... against the image opened from the file
    pbf = bgt_amd.HipPbf.open(prefix + ".pbf")
    rd = bgt_amd.HipReader(pbf)
    rd.select(cols)
    got = rd.scan(0, sites)
    assert np.array_equal(got, want), (name, "image from the file", rd.path(), rd.geometry(), np.nonzero((got != want).any((1, 2)))[0][:5])
    rd.close()
    pbf.close()
    # ... and against the launch bench.py times: the image built from the generator's strings (HipPbf.from_rle)
    rle, lens = bgt_amd.synth_rows(m, 0, sites, seed)
    pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
    del rle, lens
    rd = bgt_amd.HipReader(pbf)
    rd.select(cols)
    got = rd.scan(0, sites)
    assert np.array_equal(got, want), (name, "the benchmark's image", rd.path(), rd.geometry(), np.nonzero((got != want).any((1, 2)))[0][:5])
    if every:
        assert rd.path()["plane_split"], rd.path()
    rd.close()
    pbf.close()
