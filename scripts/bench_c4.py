#!/usr/bin/env python3
"""Configuration C4 of BASELINE.json: one 100,000-sample x 10,000,000-site cohort, site-range sharded by checkpoint
block over the GPUs of a node, per-shard allele counts gathered over RCCL (strong scaling: the total is fixed).

  python scripts/bench_c4.py                                            # one GPU holds all 1,221 blocks (~30 GB HBM)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/bench_c4.py

Prints one JSON line on rank 0.  Same kernels, generator and ABI as bench.py; shards as bgt_amd/shard.py."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=100000)
    ap.add_argument("--sites", type=int, default=10000000)
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import bgt_amd
    from bgt_amd import shard
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    m = 2 * args.samples
    shards = shard.block_shards(args.sites, 13, world)                 # whole 8192-row blocks; the last may be shorter
    r0, r1 = shards[rank]
    t0 = time.time()
    rle, lens = bgt_amd.synth_rows(m, r0, r1 - r0, args.seed)
    pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens, device=local)
    t_setup = time.time() - t0
    del rle
    rd = bgt_amd.HipReader(pbf)
    flt = bgt_amd.HipFilter("AC>0", device=local)
    counts = torch.empty((r1 - r0, 1, 3), dtype=torch.int32, device=dev)
    flags = torch.empty(max(r1 - r0, 1), dtype=torch.uint8, device=dev)
    n_pass = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()

    def step():
        rd.scan_device(0, r1 - r0, counts.data_ptr(), stream=stream.cuda_stream)
        n_pass.zero_()
        flt.apply_device(counts.data_ptr(), r1 - r0, 3, flags.data_ptr(), n_pass.data_ptr(), stream.cuda_stream)
        allc = shard.gather_counts(dist, counts, shards, rank) if world > 1 else counts
        if world > 1:
            dist.all_reduce(n_pass)
        return allc

    step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        allc = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        assert allc.shape[0] == args.sites
        print(json.dumps({"workload": "C4: %d samples x %d sites, %d GPU(s), block shards + gather of counts" %
                          (args.samples, args.sites, world), "sites_per_s": args.sites / dt, "ms_per_pass": dt * 1e3,
                          "kernel_ms_rank0": rd.timing()["scan_ms"], "sites_passing_filter": int(n_pass.item()),
                          "shards": [b - a for a, b in shards], "setup_s_rank0": round(t_setup, 1),
                          "hbm_resident_bytes_rank0": pbf.hbm_bytes, "launch": rd.geometry()}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
