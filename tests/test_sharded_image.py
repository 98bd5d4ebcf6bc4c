"""Site-range sharding BEHIND the C ABI (bgth_pbf_open_sharded): one database dealt out as block-aligned
shards, each a partial image with its own reader, stream and host thread.  The box has one GPU, so the shards of the GPU
tests all live on device 0 ("virtual shards"): what is pinned is everything but the device numbers -- the dealing of
blocks, row offsets of the partial images, concurrent scans into one host array, pull-interface windows that straddle
shard boundaries, the selection replicated per shard -- and, through `BGT_GPUS`, the whole `bgt view` on top of it,
two-database merge included (configs[3] and [4] of BASELINE.json in their product form)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import orc
import scenarios
from bgt_amd.shard import block_shards
from conftest import require_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BGT = os.path.join(ROOT, "bgt_amd", "bin", "bgt")


def test_shard_ranges_of_the_c_abi_match_the_python_sharding():
    import bgt_amd
    for n_rows, shift, world in [(1000000, 13, 8), (10000000, 13, 8), (50000, 13, 8), (8192, 13, 2), (8193, 13, 2),
                                 (1, 13, 4), (150, 4, 3), (0, 13, 2), (128072, 13, 8)]:
        assert bgt_amd.shard_ranges(n_rows, shift, world) == block_shards(n_rows, shift, world)
    sh = bgt_amd.shard_ranges(10000000, 13, 8)                       # configs[3]: 153 blocks per GPU, the last shorter
    assert [(b - a) // 8192 for a, b in sh[:7]] == [153] * 7 and sh[7] == (7 * 153 * 8192, 10000000)


@pytest.mark.gpu
@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_sharded_image_equals_the_single_image(tmp_path, n_shards):
    import bgt_amd
    rng = np.random.default_rng(11)
    m, rows, shift = 700, 150, 4                                      # 10 blocks of 16 rows (the last one ragged)
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=9, switch=0.05)
    data = orc.encode_pbf(mat, 2, shift)
    path = str(tmp_path / "x.pbf")
    open(path, "wb").write(data)
    one = bgt_amd.HipReader(bgt_amd.HipPbf.from_bytes(data))
    pbf = bgt_amd.HipPbf.open_sharded(path, [0] * n_shards)
    assert pbf.n == rows and pbf.n_shards == min(n_shards, len([r for r in block_shards(rows, shift, n_shards) if r[1] > r[0]]))
    rd = bgt_amd.HipReader(pbf)
    c1, g1 = one.scan(0, rows, want_gt=True)
    c2, g2 = rd.scan(0, rows, want_gt=True)
    assert np.array_equal(c1, c2) and np.array_equal(g1, g2)
    assert np.array_equal(rd.scan(7, 131), c1[7:131])                 # starts and ends inside shards
    # a subset with groups, replicated on every shard
    pick = np.sort(rng.choice(m // 2, 90, replace=False))
    cols = np.stack([2 * pick, 2 * pick + 1], 1).reshape(-1)
    grp = (1 + np.arange(90) % 3).astype(np.uint32)
    for r in (one, rd):
        r.select(cols, group=grp, n_groups=3)
    assert np.array_equal(one.scan(0, rows), rd.scan(0, rows))
    # the pull interface: windows straddle shard boundaries; planes, genotype vector and counts per row
    for r in (one, rd):
        r.config(r.WANT_PLANES | r.WANT_GT8, 0)
    for start in (0, 15, 16, 77, 149):
        one.seek(start); rd.seek(start)
        for _ in range(min(40, rows - start)):
            a, b = one.read(), rd.read()
            assert np.array_equal(a, b) and np.array_equal(one.last_gt8(), rd.last_gt8())
            assert np.array_equal(one.last_counts(), rd.last_counts())
    rd.seek(rows - 1); assert rd.read() is not None and rd.read() is None
    # counts that STAY on the device: the shards' pieces are gathered on shard 0's device (device copies between shards of
    # one device; BGTH_FORCE_RCCL_TO_SELF sends the same bytes through RCCL, rank to itself -- the code path of real multi-GPU
    # nodes, exercised here on one device), and the device filter runs on the gathered array
    import torch
    want = one.scan(0, rows)
    for variant in (None, "1024"):
        if variant:
            bgt_amd.force_kernels(int(variant))
        try:
            for a, b in ((0, rows), (7, rows - 9), (16, 17)):
                d = torch.full((b - a, 4, 3), -1, dtype=torch.int32, device="cuda")
                assert rd.scan_device(a, b, d.data_ptr()) == b - a
                torch.cuda.synchronize()
                assert np.array_equal(d.cpu().numpy(), want[a:b]), (variant, a, b)
            st = torch.cuda.Stream()
            d = torch.zeros((rows, 4, 3), dtype=torch.int32, device="cuda")
            flags = torch.zeros(rows, dtype=torch.uint8, device="cuda")
            n_pass = torch.zeros(1, dtype=torch.int64, device="cuda")
            flt = bgt_amd.HipFilter("AC1>0&&AC2==0", n_groups=3)
            rd.scan_device(0, rows, d.data_ptr(), stream=st.cuda_stream)          # caller's stream: nothing waits on the host
            flt.apply_device(d.data_ptr(), rows, 12, flags.data_ptr(), n_pass.data_ptr(), st.cuda_stream)
            st.synchronize()
            exp = (want[:, 1, 1] > 0) & (want[:, 2, 1] == 0)
            assert np.array_equal(flags.cpu().numpy() != 0, exp) and int(n_pass.item()) == int(exp.sum())
        finally:
            bgt_amd.force_kernels(0)
    with pytest.raises(RuntimeError):                                 # bit planes are not gathered
        rd.scan_device(0, rows, d.data_ptr(), d.data_ptr(), d.data_ptr())


@pytest.mark.gpu
def test_eight_shards_at_the_width_of_configs_3_share_one_device(tmp_path, monkeypatch):
    """The dry run a first 8-GPU node deserves: a database at the WIDTH of BASELINE configs[3] (100,000
    samples) dealt over EIGHT shards that all live on device 0, every shard on the directory path with its arena capped at
    its share (an eighth) of the device's arena budget -- set small here, so that every shard walks its rows in several
    passes -- double-buffered device scans on a caller's stream (the pattern of bench.py; the second scan of a shard may not
    overwrite counts the first gather has not copied yet: ADVICE r4), gathered counts equal to the single image's on every row,
    and the HBM the eight readers keep within the budget."""
    import torch
    import bgt_amd
    m, shift = 200000, 13
    rows = 8 * 3 * 8192 - 1000                                       # 24 file blocks, the last one ragged: 3 per shard
    rle, lens = bgt_amd.synth_rows(m, 0, rows, 4)                    # seed 4 = C4
    one_img = bgt_amd.HipPbf.from_rle(m, shift, rle, lens)
    del rle
    path = str(tmp_path / "c4w.pbf")
    one_img.save(path)
    one = bgt_amd.HipReader(one_img)
    want = one.scan(0, rows)
    assert one.path()["directory_path"]
    one.close(); one_img.close()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    budget_mb = 6000
    monkeypatch.setenv("BGTH_DIR_ARENA_MB", str(budget_mb))          # per shard: 750 MB = 7,680 rows of 100 KB: three sub-blocks a pass
    free0 = torch.cuda.mem_get_info()[0]
    pbf = bgt_amd.HipPbf.open_sharded(path, [0] * 8)
    assert pbf.n == rows and pbf.n_shards == 8
    rd = bgt_amd.HipReader(pbf)
    st = torch.cuda.Stream()
    bufs = [torch.full((rows, 1, 3), -1, dtype=torch.int32, device="cuda") for _ in range(2)]
    with bgt_amd.forced_kernels(bgt_amd.hip.FORCE_REBUILD_ROWS):
        for k in range(4):                                           # back to back, nothing waits on the host in between
            assert rd.scan_device(0, rows, bufs[k & 1].data_ptr(), stream=st.cuda_stream) == rows
    st.synchronize()
    for b in bufs:
        assert np.array_equal(b.cpu().numpy(), want)
    p = rd.path()
    assert p["directory_path"] and p["passes"] >= 2, p               # (of shard 0: its arena holds a part of its rows at a time)
    used = free0 - torch.cuda.mem_get_info()[0]
    image = pbf.hbm_bytes
    assert used <= image + (budget_mb << 20) + (1 << 30), (used, image)   # images + the eight arenas (<= the budget) + readers' buffers
    a, b = 2 * 8192 + 5, 21 * 8192 - 3                               # a range that starts and ends inside shards
    d = torch.zeros((b - a, 1, 3), dtype=torch.int32, device="cuda")
    rd.scan_device(a, b, d.data_ptr())
    assert np.array_equal(d.cpu().numpy(), want[a:b])
    rd.close(); pbf.close()


def md5_of(cmd, env=None):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    return p.returncode, hashlib.md5(p.stdout).hexdigest(), len(p.stdout), p.stderr.decode()[-300:]


@pytest.mark.gpu
def test_bgt_view_over_shards_like_configs_3_and_4(tmp_path):
    """`bgt view` with BGT_GPUS: one database over 4 shards (configs[3] in its product form) and the two-database,
    two-group merge over 3 shards each (configs[4]); byte-identical to the compiled reference on the same files."""
    import bgt_amd
    ref = require_ref("bgt")
    bgt_amd.build_library()
    __import__("bgt_amd").build_host_shell()
    a, b = str(tmp_path / "dba"), str(tmp_path / "dbb")
    subprocess.check_call([BGT, "synth", a, "3000", "40000", "5"])       # 5 file blocks
    subprocess.check_call([BGT, "synth", b, "2000", "40000", "6"])
    for gpus, args, dbs in (("0,0,0,0", ["-G", "-f", "AC>0"], [a]), ("0,0,0", ["-G", "-s", "pop==\"A\"", "-s", "pop==\"B\"", "-f", "AC1>0&&AC2==0"], [a, b]),
                            ("0,0", ["-C", "-s", "idx%500==1", "-i", "8000", "-n", "9000"], [a]), ("0,0,0,0,0,0,0,0", ["-G", "-C", "-r", "11:90000-330000"], [b, a])):
        env = dict(os.environ, BGT_GPUS=gpus)
        mine = md5_of([BGT, "view"] + args + dbs, env=env)
        want = md5_of([ref, "view"] + args + dbs)
        assert mine[0] == want[0] == 0, (gpus, args, mine, want)
        assert mine[2] > 0 and mine[1:3] == want[1:3], (gpus, args, mine, want)
