"""Multi-rank path on CPU: world_size-2 gloo.  The genotype kernels need a GPU, so the ranks exchange
counts produced by the CPU oracle for their own block-aligned shard; what is tested is the sharding
arithmetic and that the gathered array is exactly the single-process scan, in site order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import orc
import scenarios
from bgt_amd.shard import block_shards, gather_counts


def test_block_shards_cover_rows_once():
    for n_rows, shift, world in [(1000000, 13, 8), (10000000, 13, 8), (50000, 13, 8), (8192, 13, 2), (8193, 13, 2),
                                 (1, 13, 4), (150, 4, 3), (0, 13, 2)]:
        sh = block_shards(n_rows, shift, world)
        assert len(sh) == world
        assert sh[0][0] == 0 and sh[-1][1] == n_rows
        for (a0, a1), (b0, b1) in zip(sh, sh[1:]):
            assert a1 == b0 and a0 <= a1
        for r0, r1 in sh:
            assert r0 % (1 << shift) == 0 or r0 == n_rows
    # BASELINE config 4: 10M sites, 1221 blocks -> 153 blocks per GPU, the last GPU shorter
    sh = block_shards(10000000, 13, 8)
    assert [(b - a) // 8192 for a, b in sh[:7]] == [153] * 7 and sh[7][1] - sh[7][0] == 10000000 - 7 * 153 * 8192


def _worker(rank, world, port, data, n_rows, shift, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shards = block_shards(n_rows, shift, world)
    r0, r1 = shards[rank]
    local = orc.Pbf(data).scan(r0, r1).reshape(r1 - r0, 1, 3) if r1 > r0 else np.zeros((0, 1, 3), np.int32)
    got = gather_counts(dist, torch.from_numpy(np.ascontiguousarray(local)), shards, rank)
    if rank == 0:
        ret.put(got.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gathered_shards_equal_single_scan(world):
    rng = np.random.default_rng(3)
    mat = scenarios.ld_matrix(rng, 150, 64)
    shift = 4                                                     # 16-row blocks -> 10 blocks over the ranks
    data = orc.encode_pbf(mat, 2, shift)
    full = orc.Pbf(data).scan(0, 150).reshape(150, 1, 3)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, data, 150, shift, ret)) for r in range(world)]
    for p in procs:
        p.start()
    got = ret.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(got, full)
