"""The exact command the driver uses for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), dry-run on the one
GPU of the test box: two ranks, both on device 0 (BENCH_ALL_RANKS_ON_DEVICE0), collectives over gloo instead of RCCL.
Everything else is the real path: C4 shape (100,000 samples) sharded by file blocks, every rank's shard built from the
identity order and re-based onto the composition of the earlier shards (ONE database), per-shard scans on the device,
gather, and rank 0's on-box parity checks (plane-popcount identity on every site of every shard + a CPU-oracle window
across the boundary between shard 0 and shard 1)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workload,sites,extra", [("c4", 6 * 8192 - 1000, []), ("c2", 40000, ["--no-secondary"])])
def test_two_ranks_on_one_device(workload, sites, extra, tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--sites", str(sites), "--detail", str(tmp_path / "detail.json")] + ["--workload", workload] + extra
    env = dict(os.environ, BENCH_ALL_RANKS_ON_DEVICE0="1", BGTH_DIR_ARENA_MB="6000")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1200, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = p.stdout.decode().splitlines()[-1]                 # the driver reads the LAST line: compact, strict JSON
    assert len(line) < 4096
    short = json.loads(line, parse_constant=lambda c: pytest.fail("non-finite number in the bench line: " + c))
    out = json.load(open(tmp_path / "detail.json"))           # the full record
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "roofline", "config", "parity_ok"):
        assert k in short, k
    assert short["n_gpus"] == 2 and short["parity_ok"] is True and abs(short["value"] - out["value"]) <= 1e-5 * out["value"]
    assert short["parity"]["sites_checked_popcount_identity"] == short["config"]["sites_total"] and len(short["per_rank_kernel_ms"]) == 2
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert out["scaling"] == ("strong" if workload == "c4" else "weak")
    if workload == "c4":
        assert "C4" in out["config"]["workload"] and out["config"]["haplotypes"] == 200000
    assert out["parity_ok"] is True, out["parity"]
    assert out["parity"]["popcount_identity_ok"] is True and out["parity"]["oracle_window"]["matches"] is True
    assert out["parity"]["sites_checked_popcount_identity"] == out["config"]["sites_total"]
    assert len(out["per_rank_kernel_ms"]) == 2 and "parity_error" not in out


def test_default_run_of_two_ranks_carries_the_sharded_c4_record(tmp_path):
    """The driver's command without --workload: the headline is the per-GPU C2 workload (weak scaling: a value the N = 1 value
    compares with), BASELINE configs[3] block-sharded over the ranks rides along as secondary record, both checked on the box."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--sites", "40000", "--secondary-sites", str(6 * 8192 - 1000), "--secondary-steps", "1",
           "--detail", str(tmp_path / "detail.json")]
    env = dict(os.environ, BENCH_ALL_RANKS_ON_DEVICE0="1", BGTH_DIR_ARENA_MB="6000")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1200, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    short = json.loads(p.stdout.decode().splitlines()[-1])
    assert len(p.stdout.decode().splitlines()[-1]) < 4096 and short["scaling"] == "weak" and short["ranks"]["device_of_rank"] == [0, 0]
    s0 = short["secondary"][0]
    assert s0["name"] == "C4-sharded" and s0["scaling"] == "strong" and s0["haplotypes"] == 200000 and s0["parity_ok"] is True
    assert s0["sites_checked_popcount_identity"] == s0["sites_total"] and s0["sites_per_s"] > 0
    out = json.load(open(tmp_path / "detail.json"))
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["haplotypes"] == 20000 and out["parity_ok"] is True
    assert out["ranks"]["world_size"] == 2 and out["ranks"]["device_of_rank"] == [0, 0]
    sec = out["secondary"][0]
    assert sec["name"] == "C4-sharded" and "error" not in sec, sec
    assert sec["scaling"] == "strong" and sec["config"]["haplotypes"] == 200000 and sec["value"] > 0
    assert sec["parity_ok"] is True and sec["parity"]["sites_checked_popcount_identity"] == sec["config"]["sites_total"]
    assert "parity_error" not in out
    # ... and the PRODUCT's multi-GPU path, timed by rank 0 alone after the ranks have left: one process, the same database file
    # dealt over the devices by bgth_pbf_open_sharded, counts gathered by bgth_reader_scan_device; `BGT_GPUS=... bgt view` beside it
    ps = out["product_sharded"]
    assert "error" not in ps, ps
    assert ps["devices"] == [0, 0] and ps["parity_ok"] is True and ps["parity"]["sites"] == 80000 and len(ps["per_shard_kernel_ms"]) == 2
    assert ps["ms_per_step"] > 0 and ps["gather_ms"] >= 0 and ps["bgt_view"]["stdout_identical"] is True
    assert short["product_sharded"]["parity_ok"] is True and short["product_sharded"]["bgt_view_stdout_identical"] is True


def test_product_sharded_record_on_one_rank(tmp_path):
    """`--product-devices 0,0,0` on a plain N = 1 run: three shards of one database on the box's one device, gathered through RCCL
    rank-to-itself as well (BGTH_FORCE_RCCL_TO_SELF is the library's test knob for the send / receive pairs)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--sites", "50000", "--no-secondary",
           "--cpu-sample", "0", "--no-counters", "--product-devices", "0,0,0", "--detail", str(tmp_path / "detail.json")]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    short = json.loads(p.stdout.decode().splitlines()[-1])
    ps = json.load(open(tmp_path / "detail.json"))["product_sharded"]
    assert "error" not in ps, ps
    assert ps["devices"] == [0, 0, 0] and ps["parity_ok"] is True and len(ps["per_shard_kernel_ms"]) == 3 and ps["parity"]["sites"] == 50000
    assert short["product_sharded"]["parity_ok"] is True
