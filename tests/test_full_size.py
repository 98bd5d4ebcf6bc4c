"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle would need minutes
there).  For every site the RLE strings alone determine the number of ones in each bit plane, whatever the
permutation:  ones(plane 0) = n(ALT) + n(<M>)  and  ones(plane 1) = n(missing) + n(<M>).  The device reports
AN = m - n(missing), AC = n(ALT), AC<M> = n(<M>), so for the whole cohort
      AC + ACM == ones0        and        (m - AN) + ACM == ones1            for every one of the sites,
and sums over sample groups / column slices must reproduce the whole-cohort counts.  A sampled sub-range
is additionally compared with the CPU oracle."""
import os

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


def ones_per_string(rle, lens):
    code = rle >> 1
    length = (code & 15).astype(np.int64) << (4 * (code >> 4)).astype(np.int64)
    ones = np.where(rle & 1, length, 0)
    off = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])[:-1]
    tot = np.add.reduceat(ones, np.minimum(off, max(rle.size - 1, 0)))
    tot[lens == 0] = 0
    return tot.reshape(-1, 2)


@pytest.mark.parametrize("n_samples,sites,seed", [(10000, 1000000, 2), (100000, 60000, 3), (2504, 50000, 1)])
def test_plane_popcounts_match_counts_everywhere(n_samples, sites, seed):
    import bgt_amd
    m = 2 * n_samples
    rle, lens = bgt_amd.synth_rows(m, 0, sites, seed)
    pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
    rd = bgt_amd.HipReader(pbf)
    counts = rd.scan(0, sites)[:, 0, :].astype(np.int64)
    ones = ones_per_string(rle, lens)
    an, ac, acm = counts[:, 0], counts[:, 1], counts[:, 2]
    assert np.array_equal(ac + acm, ones[:, 0])
    assert np.array_equal((m - an) + acm, ones[:, 1])
    assert (an >= 0).all() and (an <= m).all()

    # three sample groups partition the cohort: their counts add up to the whole-cohort counts
    group = (1 + np.arange(n_samples) % 3).astype(np.uint32)
    rd.select(np.arange(m), group=group, n_groups=3)
    lo, hi = sites // 2, min(sites, sites // 2 + 20000)
    g = rd.scan(lo, hi).astype(np.int64)
    assert np.array_equal(g[:, 0, :], counts[lo:hi])
    assert np.array_equal(g[:, 1:, :].sum(1), counts[lo:hi])

    # a sample subset decoded through a different launch geometry agrees with the group it equals
    sel = np.nonzero(group == 2)[0]
    cols = np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1)
    rd.select(cols)
    sub = rd.scan(lo, hi).astype(np.int64)
    assert np.array_equal(sub[:, 0, :], g[:, 2, :])

    # a window deep inside the file against the CPU oracle (needs the checkpoints: save and reload)
    import tempfile
    w0 = (sites // 8192 // 2) * 8192
    w1 = min(sites, w0 + (2000 if m > 50000 else 12000))
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "x.pbf")
        pbf.save(path)
        data = open(path, "rb").read()
    oc = orc.Pbf(data).scan(w0, w1).astype(np.int64)
    assert np.array_equal(oc, counts[w0:w1])


@pytest.mark.parametrize("name,n_samples,sites,seed,every", [("C3", 100000, 1000000, 3, 20), ("C4 shard", 100000, 153 * 8192, 4, 7), ("C2", 10000, 1000000, 2, 3)])
def test_subsets_and_groups_at_width_against_the_oracle(name, n_samples, sites, seed, every, tmp_path):
    """Subsets and sample groups AT WIDTH and AT LENGTH against the CPU oracle, not against themselves.  Three windows per shape:
    across the FIRST block boundary, across a boundary in the MIDDLE of the image, and the last rows of the image with its
    ragged last block (C3, C2: 1,000,000 = 122 x 8192 + 576 sites).  A window starts 2,048 rows (narrow cohorts: 8,192) before
    its boundary; the strings are drawn again, built into a small image, re-based onto the ranks the big image holds there --
    bench.oracle_window -- and decoded by the oracle sequentially across the boundary.  C3 = BASELINE configs[2] at its full
    1,000,000 sites: every 20th of 100,000 samples (the plane-split kernels), one and three groups; one C4 shard (153 file
    blocks, 1,253,376 sites): a subset with three groups and the whole cohort in three groups (directory path); C2: the same
    at 10,000 samples."""
    import bgt_amd
    import bench
    m = 2 * n_samples
    rle, lens = bgt_amd.synth_rows(m, 0, sites, seed)
    pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
    del rle
    rd = bgt_amd.HipReader(pbf)
    back, ahead = (2048, 512) if m > 20000 else (8192, 1024)
    last_blk = (sites - 1) // 8192 * 8192                              # first row of the last (ragged or whole) block
    windows = [("first boundary", 8192 - back if 8192 - back >= 4096 else 2 * 8192 - back, None),
               ("middle", (sites // 2) // 8192 * 8192 - back, None),
               ("end of the image", last_blk - back, sites)]
    sel = np.arange(0, n_samples, every)
    cols = np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1).astype(np.int32)
    whole3 = (1 + (np.arange(n_samples) % 3)).astype(np.uint32)
    for label, lo, hi in windows:
        hi = lo + back + ahead if hi is None else hi
        assert 4096 <= lo < hi <= sites
        for n_groups in (1, 3):
            group = (1 + (np.arange(sel.size) % n_groups)).astype(np.uint32) if n_groups > 1 else None
            rd.select(cols, group=group, n_groups=n_groups)
            got = rd.scan(lo, hi)
            oc, _ = bench.oracle_window(bgt_amd, np, pbf, m, 13, seed, lo, lo, hi - lo, str(tmp_path), 0, cols, group, n_groups)
            assert np.array_equal(got, oc), (name, label, n_groups, rd.path(), rd.geometry())
        # the whole cohort in three groups
        rd.select(np.arange(m, dtype=np.int32), group=whole3, n_groups=3)
        got = rd.scan(lo, hi)
        oc, _ = bench.oracle_window(bgt_amd, np, pbf, m, 13, seed, lo, lo, hi - lo, str(tmp_path), 0, None, whole3, 3)
        assert np.array_equal(got, oc), (name, label, "whole cohort, 3 groups", rd.path(), rd.geometry())
    if name == "C3":
        # ... and the scan the benchmark times, over ALL 1,000,000 sites in one launch: its rows inside the three windows are the
        # windows' (a scan that starts at row 0 and one that starts at a sub-checkpoint must agree on every row they share)
        rd.select(cols)
        full = rd.scan(0, sites)
        assert rd.path()["plane_split"], rd.path()
        for label, lo, hi in windows:
            hi = lo + back + ahead if hi is None else hi
            assert np.array_equal(full[lo:hi], rd.scan(lo, hi)), (name, label, "full-length scan against the window's")
    rd.close()
    pbf.close()


@pytest.mark.parametrize("n_samples,sites,seed", [(10000, 131072, 2), (2504, 100000, 1), (100000, 6000, 3)])
def test_decode_encode_round_trip_at_scale(n_samples, sites, seed, tmp_path):
    """decode -> encode -> decode: the synthetic cohort's run-length strings come from the CPU generator; scanning them
    gives the genotype matrix, the device WRITER turns the matrix back into a .pbf, and (1) that image scans back to the
    same counts and genotypes, (2) its run-length strings are the generator's, byte for byte -- two independent encoders,
    one on each side of the PCIe bus, agreeing on every row."""
    import bgt_amd
    m = 2 * n_samples
    rle, lens = bgt_amd.synth_rows(m, 0, sites, seed)
    src = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
    counts, gt = bgt_amd.HipReader(src).scan(0, sites, want_gt=True)
    codes = np.empty((sites, m), np.uint8)
    for k in range(4):
        codes[:, k::4] = ((gt >> (2 * k)) & 3)[:, :(m - k + 3) // 4]
    enc = bgt_amd.HipEncoder(m, 2, 13)
    half = sites // 2 + 7
    enc.write(codes[:half])
    enc.write_packed(gt[half:])                               # the reader's 2-bit rows go straight back in
    image = enc.finish()
    enc.close()
    back = bgt_amd.HipPbf.from_bytes(image)
    counts2, gt2 = bgt_amd.HipReader(back).scan(0, sites, want_gt=True)
    assert np.array_equal(counts2, counts) and np.array_equal(gt2, gt)
    path = str(tmp_path / "src.pbf")
    src.save(path)
    assert open(path, "rb").read() == image
