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


def _s_records(data):
    """rank form (column -> rank, per plane) of every 'S' record of a .pbf image (reference pbwt.c:292-301 writes the
    permutation rank -> column; :343 inverts it)."""
    import struct
    m, g, shift = struct.unpack("<iii", data[4:16])
    pos, out = 16, []
    while data[pos:pos + 1] != b"I":
        if data[pos:pos + 1] == b"S":
            perm = np.frombuffer(data, np.int32, g * m, pos + 1).reshape(g, m)
            inv = np.empty_like(perm)
            for p in range(g):
                inv[p][perm[p]] = np.arange(m, dtype=np.int32)
            out.append(inv)
            pos += 1 + 4 * g * m
        pos += 1
        for _ in range(g):
            (l,) = struct.unpack("<i", data[pos:pos + 4])
            pos += 4 + l
    return out


def test_rank_map_composition_on_the_oracle():
    """The arithmetic behind bgth_pbf_rebase and the parallel checkpoint derivation, pinned to the oracle writer's 'S'
    records on the CPU: a row moves whatever sits at position R to LF(R) = bit ? n0 + rank1(R) : R - rank1(R)
    (reference pbwt.c:76-88), so what a block does to the order is a map F_b of POSITIONS -- the ranks it produces from
    the identity order -- and the checkpoint after the block is F_b o (the checkpoint before it), whatever that was."""
    rng = np.random.default_rng(11)
    m, shift, rows = 97, 3, 8 * 9
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=6, switch=0.2)
    mat[rng.integers(0, rows, 30), rng.integers(0, m, 30)] = 3
    mat[rng.integers(0, rows, 30), rng.integers(0, m, 30)] = 2
    whole = _s_records(orc.encode_pbf(mat, 2, shift))              # column -> rank before rows 0, 8, 16, ...
    assert len(whole) == 9
    for p in range(2):
        bits = (mat >> p) & 1
        for b in range(8):
            # the block's rows in PBWT order (what its strings spell), from the order the file holds before the block
            rank = whole[b][p].copy()
            fmap = np.arange(m)                                      # F_b so far: position before the block -> position now
            for r in range(8 * b, 8 * b + 8):
                row = np.empty(m, np.int64)
                row[rank] = bits[r]                                  # B[rank of column c] = bit of column c
                n0 = int((row == 0).sum())
                before = np.concatenate([[0], np.cumsum(row)])[:-1]  # rank1(R)
                lf = np.where(row == 1, n0 + before, np.arange(m) - before)
                rank = lf[rank]
                fmap = lf[fmap]
            assert np.array_equal(rank, whole[b + 1][p]), (p, b)    # the numpy restatement agrees with the oracle writer
            assert np.array_equal(fmap[whole[b][p]], whole[b + 1][p])   # checkpoint(b+1) = F_b o checkpoint(b)
