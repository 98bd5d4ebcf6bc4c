"""GPU parity: the HIP path (through the C ABI of libbgt_hip.so) against the CPU oracle and against the
golden fixtures the compiled reference produced.  Bit-exact: genotype bytes and AC/AN integers."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import orc
import scenarios

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "codec.npz"))


@pytest.fixture(scope="module")
def hip():
    import bgt_amd
    assert os.path.exists(bgt_amd.library_path()), "libbgt_hip.so must be built in-tree"
    assert bgt_amd.device_count() > 0, bgt_amd.last_error()
    return bgt_amd


def unpack_gt(gt, width):
    codes = np.stack([(gt >> (2 * k)) & 3 for k in range(4)], -1).reshape(gt.shape[0], -1)
    return codes[:, :width]


def oracle_scan(data, row0, row1, cols=None, group=None, n_groups=1):
    p = orc.Pbf(data)
    if cols is not None:
        p.subset(cols)
    counts, gt = p.scan(row0, row1, group=group, n_groups=n_groups, want_gt=True)
    return counts.reshape(row1 - row0, -1, 3), gt


def replay_hip(rd, ops):
    out = []
    for op in ops:
        if op[0] == "subset":
            rd.select(op[1])
        elif op[0] == "seek":
            rd.seek(op[1])
        else:
            for _ in range(op[1]):
                a = rd.read()
                if a is None:
                    break
                out.append(a)
    return np.stack(out) if out else np.zeros((0, 2, rd.width), np.uint8)


@pytest.mark.parametrize("name", list(scenarios.cases().keys()))
def test_golden_scenarios_pull_interface(hip, name):
    """pbf_subset / pbf_seek / pbf_read semantics, expected planes produced by the compiled reference."""
    mat, shift = scenarios.cases()[name]
    rows, m = mat.shape
    data = bytes(GOLD[name + "/pbf"])
    pbf = hip.HipPbf.from_bytes(data)
    assert (pbf.m, pbf.g, pbf.shift, pbf.n) == (m, 2, shift, rows)
    for i, ops in enumerate(scenarios.scenarios(name, rows, m, shift)):
        rd = hip.HipReader(pbf)
        got = replay_hip(rd, ops)
        exp = GOLD["%s/s%d" % (name, i)]
        if name == "longrun":
            exp = np.unpackbits(exp, axis=-1)[..., :int(GOLD["%s/s%d_w" % (name, i)])]
        assert got.shape == exp.shape, (name, i)
        assert np.array_equal(got, exp), (name, i)
        rd.close()
    pbf.close()


@pytest.mark.parametrize("name", list(scenarios.cases().keys()))
def test_golden_scan_counts_and_genotypes(hip, name):
    mat, shift = scenarios.cases()[name]
    rows, m = mat.shape
    data = bytes(GOLD[name + "/pbf"])
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    counts, gt = rd.scan(0, rows, want_gt=True)
    assert np.array_equal(unpack_gt(gt, m), mat)
    assert np.array_equal(counts[:, 0, 0], (mat != 2).sum(1))
    assert np.array_equal(counts[:, 0, 1], (mat == 1).sum(1))
    assert np.array_equal(counts[:, 0, 2], (mat == 3).sum(1))
    oc, ogt = oracle_scan(data, 0, rows)
    assert np.array_equal(counts, oc) and np.array_equal(gt, ogt)


def make_case(seed, m, rows, shift, **kw):
    rng = np.random.default_rng(seed)
    mat = scenarios.ld_matrix(rng, rows, m, **kw)
    return mat, orc.encode_pbf(mat, 2, shift), rng


@pytest.mark.parametrize("seed,m,rows,shift", [(11, 64, 50, 3), (12, 63, 70, 4), (13, 65, 40, 13), (14, 1, 20, 2),
                                               (15, 2, 33, 3), (16, 1000, 300, 6), (17, 5008, 200, 5),
                                               (18, 4097, 100, 13), (19, 20000, 64, 4)])
def test_random_cohorts_full(hip, seed, m, rows, shift):
    mat, data, rng = make_case(seed, m, rows, shift, n_founders=8, switch=0.02)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    counts, gt = rd.scan(0, rows, want_gt=True)
    assert np.array_equal(unpack_gt(gt, m), mat)
    oc, ogt = oracle_scan(data, 0, rows)
    assert np.array_equal(counts, oc)
    assert np.array_equal(gt, ogt)
    # counts-only path (no genotype planes) must agree with the genotype path
    assert np.array_equal(rd.scan(0, rows), oc)
    # a range that starts and ends inside blocks
    a, b = rows // 3, rows - rows // 4
    c2, g2 = rd.scan(a, b, want_gt=True)
    assert np.array_equal(c2, oc[a:b]) and np.array_equal(g2, ogt[a:b])
    assert rd.scan(5 if rows > 5 else 0, 5 if rows > 5 else 0).shape[0] == 0      # empty range


@pytest.mark.parametrize("seed,m,rows,shift,frac", [(21, 200, 120, 4, 0.05), (22, 5008, 150, 5, 0.2),
                                                    (23, 5008, 90, 13, 0.5), (24, 20000, 40, 3, 0.05),
                                                    (25, 130, 64, 3, 0.9)])
def test_sample_subset(hip, seed, m, rows, shift, frac):
    """-s sample subset: columns {2s, 2s+1} of the chosen samples in ascending order (ref bgt.c:239-242)."""
    mat, data, rng = make_case(seed, m, rows, shift, n_founders=5, switch=0.05)
    ns = m // 2
    samples = np.sort(rng.choice(ns, size=max(1, int(ns * frac)), replace=False))
    cols = np.stack([2 * samples, 2 * samples + 1], 1).reshape(-1)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    rd.select(cols)
    assert rd.width == cols.size
    counts, gt = rd.scan(0, rows, want_gt=True)
    assert np.array_equal(unpack_gt(gt, cols.size), mat[:, cols])
    oc, ogt = oracle_scan(data, 0, rows, cols=cols)
    assert np.array_equal(counts, oc) and np.array_equal(gt, ogt)


@pytest.mark.parametrize("seed,m,rows,shift,G", [(31, 120, 80, 4, 2), (32, 5008, 100, 5, 3), (33, 2000, 60, 13, 32),
                                                 (34, 20000, 30, 3, 2), (35, 300, 50, 4, 5)])
def test_sample_groups(hip, seed, m, rows, shift, G):
    """several -s groups: per-group AN/AC (ref bgt.c:740-750); samples outside every group are not decoded."""
    mat, data, rng = make_case(seed, m, rows, shift, n_founders=6, switch=0.03)
    ns = m // 2
    tag = rng.integers(0, G + 1, ns)             # 0 = not selected
    samples = np.nonzero(tag)[0]
    cols = np.stack([2 * samples, 2 * samples + 1], 1).reshape(-1)
    group = tag[samples].astype(np.uint32)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    rd.select(cols, group=group, n_groups=G)
    counts, gt = rd.scan(0, rows, want_gt=True)
    assert counts.shape == (rows, 1 + G, 3)
    oc, ogt = oracle_scan(data, 0, rows, cols=cols if cols.size < m else None, group=group, n_groups=G)
    assert np.array_equal(counts, oc)
    assert np.array_equal(gt, ogt)
    sub = mat[:, cols]
    for g in range(1, G + 1):
        sel = np.repeat(group == g, 2)
        assert np.array_equal(counts[:, g, 1], (sub[:, sel] == 1).sum(1))
        assert np.array_equal(counts[:, g, 0], (sub[:, sel] != 2).sum(1))


@pytest.mark.parametrize("threads,cpt,K", [(256, 2, 1), (256, 4, 3), (256, 8, 16), (256, 12, 4), (256, 16, 2),
                                           (256, 20, 1), (512, 4, 8), (512, 8, 4), (512, 10, 3), (512, 12, 2),
                                           (512, 16, 5), (512, 20, 8), (1024, 4, 16), (1024, 8, 2), (1024, 10, 9),
                                           (1024, 12, 3), (1024, 16, 7), (1024, 20, 5), (1024, 24, 1), (512, 24, 1),
                                           (512, 32, 2), (512, 40, 4), (512, 48, 1), (256, 2, 2), (1024, 8, 4),
                                           (512, 64, 1), (512, 80, 1), (512, 98, 1), (1024, 28, 6), (1024, 32, 5),
                                           (1024, 36, 4), (1024, 40, 2)])
def test_every_launch_geometry(hip, threads, cpt, K):
    """Force each kernel instantiation (and multi-slice launches) on one cohort."""
    mat, data, rng = make_case(41, 5008, 130, 5, n_founders=7, switch=0.04)
    oc, ogt = oracle_scan(data, 0, 130)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    rd.tune(threads, cpt, K)
    counts, gt = rd.scan(0, 130, want_gt=True)
    geo = rd.geometry()
    assert (geo["threads"], geo["cols_per_thread"]) == (threads, cpt)
    assert np.array_equal(counts, oc) and np.array_equal(gt, ogt), geo
    assert np.array_equal(rd.scan(3, 127), oc[3:127])
    # the same with two groups (the MULTI kernels)
    ns = 5008 // 2
    group = (1 + (np.arange(ns) % 2)).astype(np.uint32)
    rd.select(np.arange(5008), group=group, n_groups=2)
    c2 = rd.scan(0, 130)
    o2, _ = oracle_scan(data, 0, 130, group=group, n_groups=2)
    assert np.array_equal(c2, o2)


@pytest.mark.parametrize("threads,cpt,K", [(0, 0, 0), (512, 80, 1), (512, 64, 1), (512, 98, 1), (1024, 24, 1),
                                           (1024, 8, 1), (512, 20, 1), (256, 20, 1), (1024, 16, 2)])
@pytest.mark.parametrize("in_place", [False, True])
def test_wide_cohort_team_mode(hip, threads, cpt, K, in_place, monkeypatch):
    """Team mode on a cohort wide enough for several 256-byte chunks per string, several directory trips and
    several 8192-position segments of the row index -- noisy rows (many runs), rows of one long run, a row
    whose string stops short of m -- whole cohort, a sparse subset and two groups.  in_place: the variant for
    cohorts too wide for a separate toggle array in LDS (forced here through the debug knob)."""
    if in_place:
        hip.force_kernels(1)
    rng = np.random.default_rng(77)
    m, rows, shift = 41000, 24, 3
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=40, switch=0.2)
    mat[3] = 0; mat[4] = 1; mat[5] = 3; mat[6, :20000] = 2; mat[6, 20000:] = 0
    mat[7] = rng.integers(0, 4, m)                              # every nibble boundary, ~30k runs per plane
    data = orc.encode_pbf(mat, 2, shift)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    rd.tune(threads, cpt, K)
    oc, ogt = oracle_scan(data, 0, rows)
    counts, gt = rd.scan(0, rows, want_gt=True)
    assert np.array_equal(counts, oc) and np.array_equal(gt, ogt), rd.geometry()
    assert np.array_equal(unpack_gt(gt, m), mat)
    assert np.array_equal(rd.scan(5, 21), oc[5:21])
    cols = np.sort(rng.choice(m // 2, 700, replace=False))
    cols = np.stack([2 * cols, 2 * cols + 1], 1).reshape(-1)
    group = (1 + (np.arange(700) % 3)).astype(np.uint32)
    rd.select(cols, group=group, n_groups=3)
    rd.tune(threads if threads != 512 or cpt < 64 else 512, cpt if cpt < 64 else 8, K)
    c2, g2 = rd.scan(0, rows, want_gt=True)
    o2, og2 = oracle_scan(data, 0, rows, cols=cols, group=group, n_groups=3)
    assert np.array_equal(c2, o2) and np.array_equal(g2, og2)


def test_genotype_vector_and_text_from_the_device(hip):
    """BGTH_WANT_GT8 / BGTH_WANT_GTTEXT: what bgt_gen_gt (bgt.c:290-313, table :250) and the GT branch of
    vcf_format1 (vcf.c:940-969) make of the two planes, compared with the same mapping applied to the oracle's codes."""
    mat, data, rng = make_case(71, 1200, 300, 6, n_founders=9, switch=0.05)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    samples = np.sort(rng.choice(600, 77, replace=False))
    cols = np.stack([2 * samples, 2 * samples + 1], 1).reshape(-1).astype(np.int32)
    bits2gt = np.array([2, 4, 0, 6], np.int8)
    chars = np.frombuffer(b"01.2", np.uint8)
    for sel in (None, cols):
        if sel is not None:
            rd.select(sel)
        sub = mat if sel is None else mat[:, sel]
        for want in (rd.WANT_GT8, rd.WANT_GTTEXT, rd.WANT_GT8 | rd.WANT_GTTEXT, rd.WANT_PLANES | rd.WANT_GT8):
            rd.config(want, 64)
            for row in (0, 63, 64, 130, 299):
                rd.seek(row)
                got = rd.read()
                assert got is not None
                if want & rd.WANT_PLANES:
                    assert np.array_equal(got[0] | (got[1] << 1), sub[row])
                if want & rd.WANT_GT8:
                    assert np.array_equal(rd.last_gt8(), bits2gt[sub[row]])
                else:
                    assert rd.last_gt8() is None
                if want & rd.WANT_GTTEXT:
                    c = chars[sub[row]].reshape(-1, 2)
                    want_txt = b"".join(b"\t" + bytes([a]) + b"/" + bytes([b]) for a, b in c)
                    assert rd.last_gt_text() == want_txt
    rd.select(np.array([0, 1, 2], np.int32))                      # an odd number of columns cannot form samples
    rd.config(rd.WANT_GT8, 0)
    rd.seek(0)
    assert rd.read() is None and "whole samples" in hip.last_error()


@pytest.mark.parametrize("sub", [None, "9", "13"])
def test_sub_checkpoints(hip, tmp_path, monkeypatch, sub):
    """The image keeps rank-form checkpoints every 2^11 rows (derived on the device from the file's 'S' records
    every 2^13): scans and seeks that start inside a file block, and the saved file, must not notice."""
    if sub:
        monkeypatch.setenv("BGTH_SUB_SHIFT", sub)
    mat, data, rng = make_case(61, 300, 20000, 13, n_founders=12, switch=0.03)
    oc, ogt = oracle_scan(data, 0, 20000)
    for maker in ("bytes", "rle"):
        if maker == "bytes":
            pbf = hip.HipPbf.from_bytes(data)
        else:
            m_, shift_, strings = split_rle(data)
            pbf = hip.HipPbf.from_rle(300, 13, np.frombuffer(b"".join(strings), np.uint8),
                                      np.array([len(x) for x in strings], np.uint32))
        rd = hip.HipReader(pbf)
        counts, gt = rd.scan(0, 20000, want_gt=True)
        assert np.array_equal(counts, oc) and np.array_equal(gt, ogt)
        for a, b in [(2047, 2050), (2048, 4096), (5000, 5001), (8191, 8193), (10240, 19999), (16384, 20000)]:
            c2, g2 = rd.scan(a, b, want_gt=True)
            assert np.array_equal(c2, oc[a:b]) and np.array_equal(g2, ogt[a:b]), (a, b)
        cols = np.array([5, 4, 299, 0, 17, 16], np.int32)
        rd.select(cols)
        rd.seek(12345)
        got = np.stack([rd.read() for _ in range(3)])
        assert np.array_equal(got[:, 0] | (got[:, 1] << 1), mat[12345:12348][:, cols])
        out = str(tmp_path / ("re_%s.pbf" % maker))
        pbf.save(out)
        assert open(out, "rb").read() == data


@pytest.mark.parametrize("force", [None, "2", "4"])
@pytest.mark.parametrize("threads,cpt,K", [(0, 0, 0), (1024, 20, 0), (512, 10, 3), (256, 8, 2), (512, 80, 1), (512, 98, 1), (1024, 8, 1)])
def test_rows_with_an_empty_plane(hip, monkeypatch, force, threads, cpt, K):
    """Rows whose plane 1 (missing / <M>) is all zero take the shortcut of the ZP kernels (reference pbwt.c:135-138);
    a cohort that mixes such rows with ordinary ones, rows of one repeated code, and a plane-0-empty row, through both
    kernel families (bgth_force_kernels 2 = never, 4 = always use the ZP kernels), whole cohort, groups, genotypes."""
    if force:
        hip.force_kernels(int(force))
    rng = np.random.default_rng(123)
    m, rows, shift = 9000, 160, 5
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=12, switch=0.03)
    called = rng.random(rows) < 0.7
    mat[called] &= 1                                                   # fully called rows: codes 0/1 only
    mat[5] = 0; mat[6] = 1; mat[7] = 2; mat[8] = 3
    mat[9] = (rng.random(m) < 0.01).astype(np.uint8) * 2               # plane 0 empty, plane 1 sparse
    data = orc.encode_pbf(mat, 2, shift)
    rd = hip.HipReader(hip.HipPbf.from_bytes(data))
    rd.tune(threads, cpt, K)
    oc, ogt = oracle_scan(data, 0, rows)
    counts, gt = rd.scan(0, rows, want_gt=True)
    assert np.array_equal(counts, oc) and np.array_equal(gt, ogt), rd.geometry()
    assert np.array_equal(rd.scan(3, 150), oc[3:150])
    ns = m // 2
    group = (1 + (np.arange(ns) % 3)).astype(np.uint32)
    rd.select(np.arange(m, dtype=np.int32), group=group, n_groups=3)
    c3, g3 = rd.scan(0, rows, want_gt=True)
    o3, og3 = oracle_scan(data, 0, rows, group=group, n_groups=3)
    assert np.array_equal(c3, o3) and np.array_equal(g3, og3)


def split_rle(data):
    m, g, shift = struct.unpack("<iii", data[4:16])
    pos, strings = 16, []
    while data[pos:pos + 1] != b"I":
        if data[pos:pos + 1] == b"S":
            pos += 1 + 4 * g * m
        pos += 1
        for _ in range(g):
            (l,) = struct.unpack("<i", data[pos:pos + 4])
            strings.append(data[pos + 4:pos + 4 + l])
            pos += 4 + l
    return m, shift, strings


@pytest.mark.parametrize("sequential", [False, True])
@pytest.mark.parametrize("seed,m,rows,shift", [(51, 300, 200, 4), (52, 5008, 100, 5), (53, 70, 33, 3), (54, 41000, 50, 3),
                                               (55, 9, 16, 4), (56, 2600, 8192 + 4096 + 17, 12)])
def test_checkpoints_rebuilt_on_device(hip, tmp_path, monkeypatch, seed, m, rows, shift, sequential):
    """bgth_pbf_from_rle derives every 'S' record on the GPU; saved file == the encoder's file, byte for byte.  Default:
    every block at once from the identity order + composition of the blocks' rank maps; sequential (BGTH_FORCE_SEQUENTIAL_CHECKPOINTS): one
    launch per block.  Cases: many blocks, a ragged last block, exactly one block, team-mode width, sub-checkpoints
    inside the file blocks (shift 12 > the sub-block shift 11)."""
    if sequential:
        hip.force_kernels(512)
    mat, data, rng = make_case(seed, m, rows, shift)
    m_, shift_, strings = split_rle(data)
    rle = np.frombuffer(b"".join(strings), np.uint8)
    lens = np.array([len(s) for s in strings], np.uint32)
    pbf = hip.HipPbf.from_rle(m, shift, rle, lens)
    out = str(tmp_path / "re.pbf")
    pbf.save(out)
    assert open(out, "rb").read() == data
    fin = pbf.final_ranks()                                          # (checked against a chained shard in the next test)
    assert sorted(fin[0].tolist()) == list(range(m)) and sorted(fin[1].tolist()) == list(range(m))
    rd = hip.HipReader(pbf)
    oc, _ = oracle_scan(data, 0, rows)
    assert np.array_equal(rd.scan(0, rows), oc)


@pytest.mark.parametrize("seed,m,rows,shift,sub,force", [(41, 5008, 300, 5, None, 0), (42, 20000, 70, 4, "2", 0), (43, 41000, 40, 3, None, 0),
                                                         (44, 6400, 130, 6, "3", 32), (45, 1, 12, 2, None, 0), (46, 63, 40, 13, None, 32),
                                                         (47, 9000, 160, 5, "4", 4), (48, 7000, 150, 5, "3", 32 | 8192),
                                                         (49, 130, 60, 3, None, 32 | 8192), (50, 30000, 50, 4, None, 32 | 8192),
                                                         (51, 200000, 20, 3, None, 0), (52, 41000, 45, 3, None, 32),
                                                         (53, 70000, 37, 4, "2", 32), (54, 27000, 50, 4, None, 0),
                                                         (55, 32768, 40, 3, None, 0), (56, 39000, 34, 3, None, 0)])
def test_slots_in_rank_order_count_what_slots_in_column_order_count(hip, monkeypatch, seed, m, rows, shift, sub, force):
    """Whole cohort, one group, counts only: the slots of a sub-block are its columns in the order of their plane-0 ranks at its
    checkpoint (round 5: fewer LDS bank conflicts in the walk's gather -- profiles/r05_lds), and only n(code 3) is counted per
    column, the planes' ones being the rows' own (3 instead of 7 scalar instructions per column); BGTH_FORCE_COLUMN_ORDER keeps
    the general path.  Both against the oracle: narrow, pipelined, team and forced directory-path kernels (with four plane-row
    buffers, with sixteen and eight -- four and two rows per barrier, m <= 38,000 / 78,000 --, and with three -- BGTH_FORCE_THREE_PLANE_BUFFERS,
    as at m = 200,000), the empty-plane-1
    shortcut, sub-checkpoints inside the blocks, scans that start and end inside sub-blocks, a reader that leaves the whole
    cohort for a subset and comes back, and an image whose checkpoints change under the table (bgth_pbf_rebase)."""
    if sub:
        monkeypatch.setenv("BGTH_SUB_SHIFT", sub)
    mat, data, rng = make_case(seed, m, rows, shift, n_founders=9, switch=0.05)
    if force == 4:
        mat[::3] &= 1                                                # rows whose plane 1 is empty
        data = orc.encode_pbf(mat, 2, shift)
    oc, _ = oracle_scan(data, 0, rows)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    for order in (0, hip.hip.FORCE_COLUMN_ORDER, 0):
        hip.force_kernels(force | order)
        assert np.array_equal(rd.scan(0, rows), oc), (order, rd.geometry(), rd.path())
        for a, b in ((1, rows), (rows // 3, rows - 2), (rows - 1, rows)):
            assert np.array_equal(rd.scan(a, b), oc[a:b]), (order, a, b)
    if m >= 64:
        cols = np.arange(0, m, 3, dtype=np.int32)
        rd.select(cols)
        sc, _ = oracle_scan(data, 0, rows, cols=cols)
        assert np.array_equal(rd.scan(0, rows), sc)
        rd.select(None)
        assert np.array_equal(rd.scan(0, rows), oc)
    # genotype planes are wanted: the slot order is the selection's again (the planes are in slot order)
    c, g = rd.scan(0, rows, want_gt=True)
    assert np.array_equal(c, oc) and np.array_equal(g, oracle_scan(data, 0, rows)[1])
    rd.close(); pbf.close()
    # checkpoints that change under the table: an image from bare strings, scanned, re-based, scanned again
    _, _, strings = split_rle(data)
    rle = np.frombuffer(b"".join(strings), np.uint8)
    lens = np.array([len(x) for x in strings], np.uint32)
    img = hip.HipPbf.from_rle(m, shift, rle, lens)
    r2 = hip.HipReader(img)
    hip.force_kernels(force)
    assert np.array_equal(r2.scan(0, rows), oc)
    start = np.stack([rng.permutation(m).astype(np.int32), rng.permutation(m).astype(np.int32)])
    img.rebase(start)
    after = r2.scan(0, rows)
    hip.force_kernels(force | hip.hip.FORCE_COLUMN_ORDER)
    assert np.array_equal(r2.scan(0, rows), after)
    r2.close(); img.close()


def test_shards_of_one_database_opened_side_by_side(hip, tmp_path):
    """SURVEY 8e with ONE database: every shard is built from the identity order on its own (bgth_pbf_from_rle), the
    final ranks are chained (shard r starts from final(r-1) o ... o final(0), bgth_pbf_rebase) and the shards' scans
    concatenate to the scan of the whole image; every shard re-saved equals the corresponding records of the file."""
    rng = np.random.default_rng(2024)
    m, shift, rows = 3000, 5, 32 * 9 + 11                            # 10 blocks, the last one ragged
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=12, switch=0.1)
    mat[rng.integers(0, rows, 40), rng.integers(0, m, 40)] = 2
    mat[rng.integers(0, rows, 40), rng.integers(0, m, 40)] = 3
    data = orc.encode_pbf(mat, 2, shift)
    _, _, strings = split_rle(data)
    whole = hip.HipPbf.from_bytes(data)
    wr = hip.HipReader(whole)
    want, want_gt = wr.scan(0, rows, want_gt=True)
    oc, ogt = oracle_scan(data, 0, rows)
    assert np.array_equal(want, oc) and np.array_equal(want_gt, ogt)
    bounds = [0, 32 * 3, 32 * 4, 32 * 8, rows]                       # shards of 3, 1, 4 and 1+ blocks
    start = np.stack([np.arange(m, dtype=np.int32)] * 2)
    got, got_gt = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        ss = strings[2 * a: 2 * b]
        rle = np.frombuffer(b"".join(ss), np.uint8)
        lens = np.array([len(x) for x in ss], np.uint32)
        sh = hip.HipPbf.from_rle(m, shift, rle, lens)
        own_final = sh.final_ranks()
        sh.rebase(start)
        # final ranks after re-basing = own final map through the start order
        nxt = sh.final_ranks()
        assert np.array_equal(nxt, np.stack([own_final[p][start[p]] for p in range(2)]))
        rd = hip.HipReader(sh)
        c, g = rd.scan(0, b - a, want_gt=True)
        got.append(c); got_gt.append(g)
        assert np.array_equal(rd.scan(5, b - a - 1), want[a + 5: b - 1])   # a scan that starts inside a block
        rd.close(); sh.close()
        start = nxt
    assert np.array_equal(np.concatenate(got), want)
    assert np.array_equal(np.concatenate(got_gt), want_gt)
    with pytest.raises(RuntimeError):
        whole.final_ranks()                                          # an image read from a file keeps none
    wr.close(); whole.close()


def test_bad_inputs_fail_loudly(hip):
    with pytest.raises(RuntimeError):
        hip.HipPbf.from_bytes(b"not a pbf image at all")
    mat, data, rng = make_case(61, 64, 10, 3)
    pbf = hip.HipPbf.from_bytes(data)
    rd = hip.HipReader(pbf)
    with pytest.raises(RuntimeError):
        rd.scan(0, 11)
    with pytest.raises(RuntimeError):
        rd.select([0, 64])
    with pytest.raises(RuntimeError):
        rd.select([0, 1], group=[3], n_groups=2)
    with pytest.raises(RuntimeError):
        rd.seek(10)


def test_edge_shapes(hip, tmp_path):
    """Empty image, one row, one column, a selection of one sample, 32 groups of one sample each, the last
    sub-block shorter than the others, truncated images."""
    # no rows at all: header + footer only (what the reference writer leaves for an empty input, pbwt.c:264-276)
    empty = orc.encode_pbf(np.zeros((0, 6), np.uint8), 2, 13)
    pbf = hip.HipPbf.from_bytes(empty)
    rd = hip.HipReader(pbf)
    assert rd.scan(0, 0).shape[0] == 0 and rd.read() is None
    out = str(tmp_path / "empty.pbf")
    pbf.save(out)
    assert open(out, "rb").read() == empty
    # a single row / a single column
    for m, rows in ((1, 1), (2, 1), (1, 40), (3, 2049)):
        rng = np.random.default_rng(m * 100 + rows)
        mat = rng.integers(0, 4, (rows, m)).astype(np.uint8)
        data = orc.encode_pbf(mat, 2, 13)
        rd = hip.HipReader(hip.HipPbf.from_bytes(data))
        counts, gt = rd.scan(0, rows, want_gt=True)
        oc, ogt = oracle_scan(data, 0, rows)
        assert np.array_equal(counts, oc) and np.array_equal(gt, ogt), (m, rows)
    # one sample out of many; 32 groups of one sample each (BGT_MAX_GROUPS, bgt.h:13)
    mat, data, rng = make_case(91, 200, 2100, 13, n_founders=6, switch=0.05)
    rd = hip.HipReader(hip.HipPbf.from_bytes(data))
    rd.select(np.array([198, 199], np.int32))
    oc, ogt = oracle_scan(data, 0, 2100, cols=np.array([198, 199], np.int32))
    counts, gt = rd.scan(0, 2100, want_gt=True)
    assert np.array_equal(counts, oc) and np.array_equal(gt, ogt)
    cols = np.arange(64, dtype=np.int32)
    group = np.arange(1, 33, dtype=np.uint32)
    rd.select(cols, group=group, n_groups=32)
    o32, _ = oracle_scan(data, 0, 2100, cols=cols, group=group, n_groups=32)
    assert np.array_equal(rd.scan(0, 2100), o32)
    assert np.array_equal(rd.scan(2047, 2100), o32[2047:])           # starts one row before the second sub-block
    # truncated images fail at open, whatever the cut
    for cut in (15, 17, len(data) // 2, len(data) - 9):
        with pytest.raises(RuntimeError):
            hip.HipPbf.from_bytes(data[:cut])


def test_readers_on_threads_share_one_image(hip):
    """SURVEY 8b threading: distinct readers over one image may run on different OS threads (the Go server's
    per-request readers).  Four threads scan different selections of one image -- narrow and wide enough for the
    lazily built row index -- repeatedly; every result must equal the single-threaded one."""
    import threading
    rng = np.random.default_rng(5)
    mat = scenarios.ld_matrix(rng, 600, 41000, n_founders=20, switch=0.05)
    data = orc.encode_pbf(mat, 2, 7)
    pbf = hip.HipPbf.from_bytes(data)
    sels = [None] + [np.sort(rng.choice(20500, k, replace=False)) for k in (50, 700, 5000)]
    sels = [None if s is None else np.stack([2 * s, 2 * s + 1], 1).reshape(-1).astype(np.int32) for s in sels]
    want = []
    for s in sels:
        sub = mat if s is None else mat[:, s]
        want.append(np.stack([(sub != 2).sum(1), (sub == 1).sum(1), (sub == 3).sum(1)], 1).astype(np.int32))
    errors = []

    def work(k):
        try:
            rd = hip.HipReader(pbf)
            if sels[k] is not None:
                rd.select(sels[k])
            if k == 0:
                rd.tune(512, 80, 1)                       # team mode: builds the row index while others scan
            for _ in range(4):
                got = rd.scan(0, 600)[:, 0]
                if not np.array_equal(got, want[k]):
                    errors.append("thread %d: counts differ" % k)
            rd.close()
        except Exception as e:                            # noqa: BLE001
            errors.append("thread %d: %r" % (k, e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(sels))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("shift,rows,lo,hi", [(13, 20000, 9000, 9500), (13, 20000, 8192, 16384), (13, 20000, 0, 1),
                                               (13, 20000, 19999, 20000), (5, 1000, 37, 911), (6, 640, 0, 640)])
def test_partial_image_of_a_row_range(hip, tmp_path, shift, rows, lo, hi):
    """bgth_pbf_open_rows loads only the file blocks that cover [lo,hi): scans, seeks and reads inside the loaded
    blocks give what the whole file gives, file row numbers are kept, anything outside fails loudly."""
    mat, data, rng = make_case(81, 260, rows, shift, n_founders=10, switch=0.04)
    path = str(tmp_path / "x.pbf")
    open(path, "wb").write(data)
    oc, ogt = oracle_scan(data, 0, rows)
    pbf = hip.HipPbf.open_rows(path, lo, hi)
    assert pbf.n == rows                                         # rows of the FILE
    first = hip.lib().bgth_pbf_first_row
    first.restype = C.c_int64
    first.argtypes = [C.c_void_p]
    assert first(pbf.h) == (lo >> shift) << shift
    rd = hip.HipReader(pbf)
    counts, gt = rd.scan(lo, hi, want_gt=True)
    assert np.array_equal(counts, oc[lo:hi]) and np.array_equal(gt, ogt[lo:hi])
    rd.select(np.array([7, 6, 259, 0], np.int32))
    rd.seek(hi - 1)
    got = rd.read()
    assert np.array_equal(got[0] | (got[1] << 1), mat[hi - 1][[7, 6, 259, 0]])
    blk = 1 << shift
    loaded_end = min(rows, ((hi - 1) // blk + 1) * blk)
    if loaded_end < rows:
        with pytest.raises(RuntimeError):
            rd.scan(lo, loaded_end + 1)
        with pytest.raises(RuntimeError):
            rd.seek(loaded_end)
    if (lo // blk) * blk > 0:
        with pytest.raises(RuntimeError):
            rd.scan((lo // blk) * blk - 1, hi)
    if (lo // blk) * blk > 0 or loaded_end < rows:               # a partial image cannot be written back
        with pytest.raises(RuntimeError):
            pbf.save(str(tmp_path / "no.pbf"))
    else:
        pbf.save(str(tmp_path / "re.pbf"))
        assert open(str(tmp_path / "re.pbf"), "rb").read() == data
    with pytest.raises(RuntimeError):
        hip.HipPbf.open_rows(path, hi, lo)


@pytest.mark.gpu
@pytest.mark.parametrize("g,m,rows,shift", [(3, 100, 60, 3), (4, 5000, 80, 4), (7, 333, 40, 2), (8, 41000, 24, 3)])
def test_more_than_two_bit_planes(hip, tmp_path, g, m, rows, shift):
    """pbf_open_w / pbf_read loop over any number of planes (pbwt.c:211-213, 325-334); every plane is a PBWT of its own.  Such a file
    opens as a bundle of two-plane images and serves the codec interface: whole rows, a column subset in any order, a seek across
    checkpoints, the file written back byte for byte; what is defined for BGT's two planes only -- counts -- fails with a message.
    Against the oracle reader on the oracle writer's file."""
    rng = np.random.default_rng(g * 1000 + m)
    mat = rng.integers(0, 1 << g, (rows, m)).astype(np.uint8)
    mat[rng.random((rows, m)) < 0.7] = 0
    mat[3] = 0; mat[4] = (1 << g) - 1
    data = orc.encode_pbf(mat, g, shift)
    pbf = hip.HipPbf.from_bytes(data)
    assert (pbf.m, pbf.g, pbf.n) == (m, g, rows)
    rd = hip.HipReader(pbf)
    for r in range(rows):
        a = rd.read()
        assert a.shape == (g, m) and np.array_equal(a, np.stack([(mat[r] >> k) & 1 for k in range(g)])), r
    assert rd.read() is None
    cols = rng.permutation(m)[: max(1, m // 7)].astype(np.int32)
    rd.select(cols)
    ora = orc.Pbf(data)
    ora.subset(cols)
    start = (1 << shift) + 1
    rd.seek(start); ora.seek(start)
    for r in range(start, rows):
        assert np.array_equal(rd.read(), ora.read()), r
    out = str(tmp_path / "planes.pbf")
    pbf.save(out)
    assert open(out, "rb").read() == data
    with pytest.raises(RuntimeError, match="two planes"):
        rd.scan(0, rows)
    rd.close()
    pbf.close()
    # partial and sharded images are for BGT's two planes: a clean refusal, nothing half-opened
    with pytest.raises(RuntimeError, match="bit planes"):
        hip.HipPbf.open_rows(out, 0, min(rows, 8))
    with pytest.raises(RuntimeError, match="bit planes"):
        hip.HipPbf.open_sharded(out, [0, 0])

