"""BASELINE.json configs[3] and configs[4] at their own shapes, on ONE GPU.

C4  100,000 samples x 10,000,000 sites, site-range sharded over 8 GPUs (153 file blocks of 8192 rows per GPU,
    the last GPU shorter) with a gather of the per-shard counts.  What a rank computes does not depend on the
    other ranks, so the HIP side of the sharding is pinned here without a second GPU: every block_shards() range
    of a 100,000-sample database (>= 2 full blocks per "rank", the last one ragged) is scanned on its own --
    once from the whole image and once from a PARTIAL image that holds only that rank's file blocks, which is
    what a rank of C4 loads -- and the concatenation must equal the single whole scan; plane-popcount identities
    on every site; an oracle window straddling a shard boundary.
C5  two 50,000-sample databases (m = 100,000 each: the team-mode kernels with several sample groups), two groups
    that span both, `-f'AC1>0&&AC2==0'`, through the two-database merge of `bgt view`, byte-identical to the
    compiled reference.
"""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import orc
from bgt_amd.shard import block_shards
from conftest import require_ref
from test_full_size import ones_per_string

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BGT = os.path.join(ROOT, "bgt_amd", "bin", "bgt")

pytestmark = pytest.mark.gpu


def test_c4_shard_ranges_equal_the_whole_scan(tmp_path, monkeypatch):
    import torch
    import bgt_amd
    n_samples, shift, world = 100000, 13, 8
    m = 2 * n_samples
    sites = world * 2 * 8192 - 3000                      # 16 file blocks, the last one ragged: 2 blocks per rank
    rle, lens = bgt_amd.synth_rows(m, 0, sites, 4)       # seed 4 = C4
    pbf = bgt_amd.HipPbf.from_rle(m, shift, rle, lens)
    rd = bgt_amd.HipReader(pbf)
    whole = rd.scan(0, sites)
    # the per-GPU shape of C4 takes the directory path (rows built once, walk-only slices); the team kernels agree
    assert rd.path()["directory_path"] and rd.geometry()["slices"] >= 4, (rd.path(), rd.geometry())
    bgt_amd.force_kernels(64)
    assert np.array_equal(rd.scan(0, sites), whole) and rd.geometry()["threads"] == 512 and not rd.path()["directory_path"]
    bgt_amd.force_kernels(0)
    # plane popcounts of every site (independent of any permutation state)
    ones = ones_per_string(rle, lens)
    c = whole[:, 0, :].astype(np.int64)
    assert np.array_equal(c[:, 1] + c[:, 2], ones[:, 0]) and np.array_equal((m - c[:, 0]) + c[:, 2], ones[:, 1])

    shards = block_shards(sites, shift, world)
    assert [(b - a) // 8192 for a, b in shards[:-1]] == [2] * (world - 1) and 0 < shards[-1][1] - shards[-1][0] < 2 * 8192
    # (1) each rank's range from the whole image, results left in HBM as the multi-GPU path does
    dev = torch.device("cuda", 0)
    parts = []
    for r0, r1 in shards:
        d = torch.empty((r1 - r0, 1, 3), dtype=torch.int32, device=dev)
        rd.scan_device(r0, r1, d.data_ptr())
        torch.cuda.synchronize()
        parts.append(d.cpu().numpy())
    assert np.array_equal(np.concatenate(parts, 0), whole)

    # (2) each rank loads ONLY its own file blocks (bgth_pbf_open_rows through the footer's block index)
    path = str(tmp_path / "c4.pbf")
    pbf.save(path)
    parts = []
    for r0, r1 in shards:
        part = bgt_amd.HipPbf.open_rows(path, r0, r1)
        assert part.n == sites                                           # still reports the file's rows
        prd = bgt_amd.HipReader(part)
        parts.append(prd.scan(r0, r1))
        with pytest.raises(RuntimeError):
            prd.scan(max(0, r0 - 1), r1) if r0 > 0 else prd.scan(r0, min(sites, r1 + 8192) if r1 < sites else r1 + 1)
        prd.close(); part.close()
    assert np.array_equal(np.concatenate(parts, 0), whole)

    # (3) an oracle window across the boundary between rank 2 and rank 3 (rows 49,152 +- 150)
    edge = shards[3][0]
    data = open(path, "rb").read()
    oc = orc.Pbf(data).scan(edge - 150, edge + 150)
    assert np.array_equal(oc.reshape(300, 1, 3), whole[edge - 150: edge + 150])

    # (4) the same shards with a sample subset and groups (slot tables are replicated per rank)
    sel = np.arange(0, n_samples, 20)
    cols = np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1)
    rd.select(cols, group=(1 + np.arange(sel.size) % 2).astype(np.uint32), n_groups=2)
    sub_whole = rd.scan(0, sites)
    sub_parts = [rd.scan(r0, r1) for r0, r1 in shards]
    assert np.array_equal(np.concatenate(sub_parts, 0), sub_whole)
    assert np.array_equal(sub_whole[:, 1:, :].sum(1), sub_whole[:, 0, :])


def md5_of(cmd, timeout=900):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return p.returncode, hashlib.md5(p.stdout).hexdigest(), len(p.stdout), p.stderr.decode()[-300:]


def test_c5_two_wide_databases_two_groups(tmp_path):
    import bgt_amd
    ref = require_ref("bgt")
    bgt_amd.build_library()
    __import__("bgt_amd").build_host_shell()
    a, b = str(tmp_path / "dba"), str(tmp_path / "dbb")
    subprocess.check_call([BGT, "synth", a, "50000", "12000", "5"], timeout=900)     # seeds 5 / 6 = C5
    subprocess.check_call([BGT, "synth", b, "50000", "12000", "6"], timeout=900)
    for args in (["-G", "-s", "pop==\"A\"", "-s", "pop==\"B\"", "-f", "AC1>0&&AC2==0"],           # the configuration itself
                 ["-G", "-s", "pop==\"A\"", "-s", "pop==\"B\"", "-s", "pop==\"C\"", "-f", "AC1>AC2&&AN3>0", "-i", "9000"],
                 ["-G", "-C", "-r", "11:80000-90000"],
                 ["-s", "idx%10000==3", "-s", "idx%10000==4", "-r", "11:100000-101000"]):      # genotypes of a few samples
        mine = md5_of([BGT, "view"] + args + [a, b])
        want = md5_of([ref, "view"] + args + [a, b])
        assert mine[0] == want[0] == 0, (args, mine, want)
        assert mine[2] > 0 and mine[1:3] == want[1:3], (args, mine, want)
