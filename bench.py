#!/usr/bin/env python3
"""bench.py -- sites/sec of the `bgt view -G -f'AC>0'` whole-cohort scan on MI355X.

One "step" = one pass of the hot path over the rank's whole shard: PBWT run-length decode of both bit
planes of every site, rank-tracking column reconstruction, AC/AN reduction (all on the GPU, through the
C ABI of libbgt_hip.so), the site filter AC>0 on the device, counts + pass flags delivered to the host.  Inputs (RLE
strings, row directory, checkpoints) are resident in HBM before the timed region starts.

  N = 1   workload C2 of BASELINE.json: synthetic 10,000 samples (m = 20,000 haplotypes) x 1,000,000 sites.
  N > 1   weak scaling: rank r scans sites [r*1M, (r+1)*1M) of the same cohort (site-range sharding, no
          data-path collective), then an all_gather over RCCL/xGMI of the per-shard allele counts and pass flags.

Prints one JSON line (rank 0).  `roofline.achieved` prices the decode kernel with the reference's
ALGORITHMIC bytes (16*T + r + 12 per site); the kernel keeps that permutation state in
registers, so its real HBM traffic (`traffic`, from rocprofv3 PMC counters when profiles/ holds them) is
~3 orders of magnitude lower -- see DESIGN.md.  `cpu_baseline` times the CPU oracle (a port of the
reference path; the reference itself is not on the GPU box) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLES = {"c2": 10000, "c3": 100000, "small": 2504}
HBM_PEAK_GBS = 8000.0                      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(SAMPLES))
    ap.add_argument("--sites", type=int, default=1000000, help="sites per GPU")
    ap.add_argument("--every", type=int, default=0, help="select every N-th sample only (C3: --workload c3 --every 20)")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--cpu-sample", type=int, default=262144, help="sites for the CPU baseline (0 = skip)")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--cpt", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run" % args.gpus)

    import numpy as np
    import torch
    import bgt_amd

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)       # "nccl" is RCCL on ROCm

    n_samples = SAMPLES[args.workload]
    m = 2 * n_samples
    sites = args.sites
    shift = 13

    # ---- synthetic shard: rows [rank*sites, (rank+1)*sites) of cohort (seed, m), drawn on the host cores
    t0 = time.time()
    rle, lens = bgt_amd.synth_rows(m, rank * sites, sites, args.seed)
    t_gen = time.time() - t0
    t0 = time.time()
    pbf = bgt_amd.HipPbf.from_rle(m, shift, rle, lens, device=local)   # upload + checkpoints on the GPU
    t_load = time.time() - t0
    rle_bytes_per_site = rle.size / sites
    rd = bgt_amd.HipReader(pbf)
    if args.every > 1:                                          # sample subset (-s): fewer tracked columns, same rows
        sel = np.arange(0, n_samples, args.every)
        rd.select(np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1))
    rd.tune(args.threads, args.cpt, args.batch)
    T = rd.width

    # Two result buffers: the D2H copy of step i (rank 0, side stream) overlaps the scan of step i+1.
    flt = bgt_amd.HipFilter("AC>0", n_groups=1, device=local)            # -f'AC>0', evaluated on the device
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream(device=dev)
    counts = [torch.empty((sites, 1, 3), dtype=torch.int32, device=dev) for _ in range(2)]
    flags = [torch.empty(sites, dtype=torch.uint8, device=dev) for _ in range(2)]
    n_pass_d = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2)]
    if world > 1:
        g_counts = [torch.empty((world * sites, 1, 3), dtype=torch.int32, device=dev) for _ in range(2)]
        g_flags = [torch.empty(world * sites, dtype=torch.uint8, device=dev) for _ in range(2)]
    else:
        g_counts, g_flags = counts, flags
    if rank == 0:
        host = [torch.empty((world * sites, 1, 3), dtype=torch.int32).pin_memory() for _ in range(2)]
        host_flags = [torch.empty(world * sites, dtype=torch.uint8).pin_memory() for _ in range(2)]
        host_n_pass = [torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    for e in copied:
        e.record(main)
    kernel_ms = []
    n_steps_done = [0]

    def step():
        b = n_steps_done[0] & 1
        n_steps_done[0] += 1
        main.wait_event(copied[b])                                       # buffer b has left the device
        rd.scan_device(0, sites, counts[b].data_ptr(), stream=main.cuda_stream)
        n_pass_d[b].zero_()
        flt.apply_device(counts[b].data_ptr(), sites, 3, flags[b].data_ptr(), n_pass_d[b].data_ptr(),
                         main.cuda_stream)
        if world > 1:                                                    # per-shard AN/AC + flags over xGMI
            dist.all_gather_into_tensor(g_counts[b], counts[b])
            dist.all_gather_into_tensor(g_flags[b], flags[b])
            dist.all_reduce(n_pass_d[b])
        ready[b].record(main)
        if rank == 0:
            with torch.cuda.stream(side):
                side.wait_event(ready[b])
                host[b].copy_(g_counts[b], non_blocking=True)
                host_flags[b].copy_(g_flags[b], non_blocking=True)
                host_n_pass[b].copy_(n_pass_d[b], non_blocking=True)
                copied[b].record(side)
        return b                                                         # (nothing here waits: steps are enqueued back to back)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    kernel_ms.clear()
    barrier()
    t0 = time.perf_counter()
    last = 0
    for _ in range(args.steps):
        last = step()
    barrier()                                                            # includes the side stream: all results on the host
    dt = time.perf_counter() - t0
    kernel_ms.append(rd.timing()["scan_ms"])                             # HIP events around the scan kernel of the last step
    n_pass = int(host_n_pass[last].item()) if rank == 0 else 0
    if rank == 0:
        host = host[last]
        assert n_pass == int(host_flags[last].numpy().sum())
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    ms_per_step = dt / args.steps * 1e3
    value = world * sites / (dt / args.steps)
    k_ms = sum(kernel_ms) / len(kernel_ms)
    alg_bytes_per_site = 16.0 * T + rle_bytes_per_site + 12.0
    achieved = alg_bytes_per_site * sites / (k_ms * 1e-3) / 1e9
    geo = rd.geometry()

    out = None
    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload and tj.get("sites") == sites:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "sites/sec `bgt view -G -f'AC>0'` whole-cohort scan",
            "value": value, "unit": "sites/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "C2: synthetic %d samples x %d sites per GPU, whole cohort, -G -f'AC>0'"
                                   % (n_samples, sites) if args.workload == "c2" else
                                   "%s: %d samples x %d sites per GPU%s" % (args.workload, n_samples, sites,
                                   ", every %d-th sample selected" % args.every if args.every > 1 else ""),
                       "haplotypes": m, "tracked_columns": T, "sites_per_gpu": sites,
                       "sharding": "site-range x%d + all_gather(counts)" % world if world > 1 else "single GPU",
                       "rle_bytes_per_site": round(rle_bytes_per_site, 1), "sites_passing_filter": n_pass,
                       "filter": "AC>0 evaluated on the device (bgth_filter_apply_device); counts + flags copied "
                                 "to pinned host memory, the copy of step i overlapping the scan of step i+1",
                       "launch": geo},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "scan_kernel<%d,%d>" % (geo["threads"], geo["cols_per_thread"]),
                         "kernel_ms": k_ms, "algorithmic_bytes_per_site": alg_bytes_per_site,
                         "note": "achieved = reference-algorithm bytes / kernel time; the kernel keeps the "
                                 "permutation in registers, real HBM traffic is `traffic` (see DESIGN.md)"},
            "setup": {"generate_s": round(t_gen, 2), "upload_and_checkpoints_s": round(t_load, 2),
                      "hbm_resident_bytes": pbf.hbm_bytes},
        }

    # ---- CPU baseline + on-box parity check (rank 0, N=1 only) on a bounded sample of the same cohort:
    # the first `cpu_sample` sites.  Preferred: the COMPILED REFERENCE (oracle/_ref/bgt, built from
    # /root/reference in the build container and shipped with the repo) running the metric's own command
    # line on a database this repo writes; its stdout is also compared with this repo's `bgt view`.
    # Always: the CPU oracle (port) on the same rows, compared with the counts the GPU delivered.
    if rank == 0 and world == 1 and args.cpu_sample > 0 and args.every <= 1:
        import hashlib
        import subprocess
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc                                            # CPU oracle: checker / baseline only
        ns = min(sites, args.cpu_sample)
        nstr = 2 * ns
        nbytes = int(lens[:nstr].sum(dtype=np.int64))
        sample = bgt_amd.HipPbf.from_rle(m, shift, rle[:nbytes], lens[:nstr], device=local)
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "sample.pbf")
            sample.save(path)
            data = open(path, "rb").read()
            sample.close()
            p = orc.Pbf(data)
            t0 = time.perf_counter()
            oc = p.scan(0, ns)
            n_pass_cpu = int((oc[:, 1] > 0).sum())
            t_cpu = time.perf_counter() - t0
            same = bool(np.array_equal(oc.reshape(ns, 1, 3), host.numpy()[:ns])) and \
                bool(np.array_equal(oc[:, 1] > 0, host_flags[last].numpy()[:ns] != 0))
            port = {"value": ns / t_cpu, "unit": "sites/s", "cores": 1, "kind": "port",
                    "sample": "first %d sites of the same cohort (oracle/liborc.so: decode both planes + AC/AN + "
                              "AC>0, one thread, %.1f s)" % (ns, t_cpu),
                    "gpu_matches_cpu_on_sample": same, "sites_passing_filter": n_pass_cpu}
            out["cpu_baseline"] = port
            if not same:
                out["parity_error"] = "GPU counts differ from the CPU oracle on the sample"
            ref_bin = os.path.join(ROOT, "oracle", "_ref", "bgt")
            my_bin = os.path.join(ROOT, "bgt_amd", "bin", "bgt")
            if os.path.exists(ref_bin) and args.workload == "c2":
                try:
                    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bgt_amd", "host")])
                    prefix = os.path.join(tmp, "db")
                    subprocess.check_call([my_bin, "synth", prefix, str(n_samples), str(ns), str(args.seed)])
                    cmd = ["view", "-G", "-f", "AC>0", prefix]
                    t0 = time.perf_counter()
                    ref_out = subprocess.run([ref_bin] + cmd, stdout=subprocess.PIPE, check=True).stdout
                    t_ref = time.perf_counter() - t0
                    t0 = time.perf_counter()
                    my_out = subprocess.run([my_bin] + cmd, stdout=subprocess.PIPE, check=True).stdout
                    t_mine = time.perf_counter() - t0
                    cli_same = hashlib.md5(ref_out).hexdigest() == hashlib.md5(my_out).hexdigest()
                    out["cpu_baseline"] = {
                        "value": ns / t_ref, "unit": "sites/s", "cores": 1, "kind": "reference",
                        "sample": "reference `bgt view -G -f'AC>0'` (oracle/_ref/bgt, gcc -O2, one thread, %.1f s wall incl. "
                                  "open) on a %d-sample x %d-site database = first sites of the same cohort"
                                  % (t_ref, n_samples, ns),
                        "cli_stdout_identical_to_reference": cli_same, "cli_stdout_bytes": len(ref_out),
                        "this_repo_cli_same_command_s": round(t_mine, 2),
                        "gpu_matches_cpu_on_sample": same, "port": port}
                    # the reference is single-threaded; the whole box = one process per 8192-site block range
                    # (disjoint -r ranges), as many at a time as there are cores
                    try:
                        n_blk = (ns + 8191) // 8192
                        procs = min(n_blk, os.cpu_count() or 1)
                        t0 = time.perf_counter()
                        running = []
                        for k in range(n_blk):
                            reg = "11:%d-%d" % (1000 + 10 * k * 8192, 1000 + 10 * min(ns, (k + 1) * 8192) - 1)
                            running.append(subprocess.Popen([ref_bin, "view", "-G", "-f", "AC>0", "-r", reg, prefix],
                                                            stdout=subprocess.DEVNULL))
                            if len(running) >= procs:
                                running.pop(0).wait()
                        for pr in running:
                            pr.wait()
                        t_all = time.perf_counter() - t0
                        out["cpu_baseline"]["all_cores"] = {"value": ns / t_all, "unit": "sites/s", "processes": procs,
                                                            "sample": "%d reference processes over disjoint 8192-site regions, %.1f s wall" % (n_blk, t_all)}
                    except Exception as e:
                        out["cpu_baseline"]["all_cores"] = {"error": repr(e)[:120]}
                    if not cli_same:
                        out["parity_error"] = "`bgt view` stdout differs from the reference binary"
                except Exception as e:                        # keep the port numbers, say why
                    out["cpu_baseline"]["reference_leg_error"] = repr(e)[:200]
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
