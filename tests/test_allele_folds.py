"""`bgt view -a ALLELES -S / -H` reductions on the device (reference bgt.c:859-876):
bgth_reader_fold_last / bgth_reader_take_folds against the same loops in numpy over the oracle's decoded rows --
single image and sharded, subsets in any sample order, every code, bits up to 63, take-and-restart."""
import numpy as np
import pytest

import orc
import scenarios


def host_folds(codes, rows, want, bits):
    """codes uint8[rows_total][width]; the reference's two loops (bgt.c:862-874)."""
    width = codes.shape[1]
    car = np.zeros(width // 2, np.int32)
    hap = np.zeros(width, np.uint64)
    for r, w, b in zip(rows, want, bits):
        c = codes[r]
        if w >= 0:
            car += ((c[0::2] == w) | (c[1::2] == w)).astype(np.int32)
        if b >= 0:
            hap |= np.where(c == 1, np.uint64(1) << np.uint64(b), np.uint64(0))
    return car, hap


@pytest.mark.gpu
@pytest.mark.parametrize("n_shards", [1, 3])
def test_folds_match_the_reference_loops(tmp_path, n_shards):
    import bgt_amd
    rng = np.random.default_rng(23 + n_shards)
    m, n_rows, shift = 1300, 200, 4
    mat = scenarios.ld_matrix(rng, n_rows, m, n_founders=11, switch=0.04)
    mat[rng.random(mat.shape) < 0.02] = 2
    mat[rng.random(mat.shape) < 0.01] = 3
    data = orc.encode_pbf(mat, 2, shift)
    path = str(tmp_path / "x.pbf")
    open(path, "wb").write(data)
    pbf = bgt_amd.HipPbf.from_bytes(data) if n_shards == 1 else bgt_amd.HipPbf.open_sharded(path, [0] * n_shards)
    rd = bgt_amd.HipReader(pbf)
    for n_pick in (m // 2, 77, 1):
        # samples in any order; selecting all of them means "no subset" as in pbf_subset (pbwt.c:377): file order
        pick = rng.permutation(m // 2)[:n_pick] if n_pick < m // 2 else np.arange(m // 2)
        cols = np.stack([2 * pick, 2 * pick + 1], 1).reshape(-1)
        rd.select(cols)
        rd.config(rd.WANT_BITS, 0)
        codes = mat[:, cols]
        rows = np.sort(rng.choice(n_rows, 64, replace=False))
        want = rng.choice([0, 1, 1, 1, -1, 2, 3], 64)
        bits = np.arange(64)
        bits[rng.random(64) < 0.1] = -1
        for r, w, b in zip(rows, want, bits):
            rd.seek(int(r))
            assert rd.read() is True
            rd.fold_last(int(w), int(b))
        car, hap = rd.take_folds()
        ecar, ehap = host_folds(codes, rows, want, bits)
        assert np.array_equal(car, ecar) and np.array_equal(hap, ehap)
        # after a take the accumulators start at zero; consecutive rows without a seek
        rd.seek(5)
        for k in range(3):
            rd.read(); rd.fold_last(1, k)
        car, hap = rd.take_folds()
        ecar, ehap = host_folds(codes, [5, 6, 7], [1, 1, 1], [0, 1, 2])
        assert np.array_equal(car, ecar) and np.array_equal(hap, ehap)
        car, hap = rd.take_folds()                                    # nothing folded: zeros
        assert not car.any() and not hap.any()


@pytest.mark.gpu
def test_fold_errors_are_reported():
    import bgt_amd
    rng = np.random.default_rng(5)
    mat = scenarios.ld_matrix(rng, 20, 64, n_founders=4, switch=0.1)
    rd = bgt_amd.HipReader(bgt_amd.HipPbf.from_bytes(orc.encode_pbf(mat, 2, 13)))
    with pytest.raises(RuntimeError):                                 # nothing read yet
        rd.fold_last(1, 0)
    rd.config(rd.WANT_PLANES, 0)
    rd.read()
    with pytest.raises(RuntimeError):                                 # the bit planes were not kept
        rd.fold_last(1, 0)
    rd.config(rd.WANT_BITS, 0)
    rd.seek(0); rd.read()
    with pytest.raises(RuntimeError):
        rd.fold_last(1, 64)
    rd.select(np.array([0, 1, 2], np.int32))                          # half a sample
    rd.config(rd.WANT_BITS, 0)
    rd.read()
    with pytest.raises(RuntimeError):
        rd.fold_last(1, 0)


@pytest.mark.gpu
def test_fold_arrays_equal_the_compiled_reference(tmp_path):
    """bgtm_t::alcnt / ::hap as the COMPILED REFERENCE leaves them (bgt.c:859-876) -- tests/golden/folds.json, written by
    tests/golden/make_folds_golden.py from oracle/_ref/libbgt_ref.so -- against the same call sequence (tests/integration/api_dump.c,
    the order of bgt-server.go) over libbgt.so, where the two reductions run on the device (bgth_reader_fold_last per matched
    site, bgth_reader_take_folds at the end): `-S`, `-H`, both, a reference-allele query, a subset, samples in any order, two
    databases with groups, an allele set that matches nothing."""
    import json
    import os
    import subprocess
    import bgt_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bgt_amd.build_library()
    __import__("bgt_amd").build_host_shell()
    exe, lib = str(tmp_path / "api_mine"), os.path.join(root, "bgt_amd", "lib")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "integration", "api_dump.c"),
                           "-o", exe, "-L", lib, "-lbgt", "-Wl,-rpath," + lib])
    cases = json.load(open(os.path.join(root, "tests", "golden", "folds.json")))
    assert len(cases) >= 9
    for c in cases:
        p = subprocess.run([exe] + c["args"], cwd=os.path.join(root, "tests", "golden", "bgt"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, (c["args"], p.stderr.decode()[-300:])
        assert p.stdout.decode() == c["stdout"], c["args"]
