"""Site-range sharding of one BGT database over the GPUs of a node.

Rows are independent across 8192-row checkpoint blocks ('S' records, reference pbwt.c:292-301), so rank p of
P gets a contiguous range of whole blocks; there is no halo and no collective on the data path.  The only
exchange is one all-gather of the per-shard allele counts (12 B per site and count entry) to the rank that
applies the filter and emits the sites -- RCCL over xGMI on the GPUs, gloo in the CPU tests.
"""
import numpy as np


def block_shards(n_rows, shift, world):
    """[(row0, row1)] per rank: ceil(B/P) blocks each, the last ranks may be shorter or empty."""
    blk = 1 << shift
    n_blk = (n_rows + blk - 1) >> shift
    per = (n_blk + world - 1) // world
    out = []
    for p in range(world):
        b0, b1 = min(n_blk, p * per), min(n_blk, (p + 1) * per)
        out.append((min(n_rows, b0 << shift), min(n_rows, b1 << shift)))
    return out


def gather_counts(dist, local_counts, shards, rank):
    """All-gather variable-length shard results.  local_counts: torch int32 [rows_p, E, 3] on the rank's
    device.  Every rank contributes a buffer padded to the longest shard (all_gather_into_tensor needs equal
    sizes); returns on every rank the concatenation in site order, [n_rows, E, 3]."""
    import torch
    world = len(shards)
    longest = max(r1 - r0 for r0, r1 in shards)
    e = local_counts.shape[1]
    send = torch.zeros((longest, e, 3), dtype=torch.int32, device=local_counts.device)
    send[: local_counts.shape[0]] = local_counts
    recv = torch.empty((world * longest, e, 3), dtype=torch.int32, device=local_counts.device)
    dist.all_gather_into_tensor(recv, send)
    parts = [recv[p * longest: p * longest + (shards[p][1] - shards[p][0])] for p in range(world)]
    return torch.cat(parts, 0)


def merge_shard_arrays(parts):
    """host-side equivalent for numpy arrays (used by tests)"""
    return np.concatenate(parts, 0)
