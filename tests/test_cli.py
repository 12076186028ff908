import re
import subprocess
import sys
import os

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(script, *args):
    cmd = [sys.executable, os.path.join(ROOT, script), "--device", "cpu", "--iters", "40",
           "--batch-size", "2", "--train-size", "96", "--test-size", "8", *args]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)


def test_part1_output_contract():
    r = _run("part1/main.py")
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "Total params: 9,225,610" in out
    assert re.search(r"^Loss at 20th batch is \d+\.\d+", out, re.M)
    assert re.search(r"^Loss at 40th batch is \d+\.\d+", out, re.M)
    assert re.search(r"^Total execution time is : [\d.e-]+ seconds$", out, re.M)
    assert re.search(r"^Average execution time is  : [\d.e-]+ seconds$", out, re.M)
    assert re.search(r"^Test set: Average loss: \d+\.\d{4}, Accuracy: \d+/8 \(\d+%\)$", out, re.M)
    assert "master_ip" not in out          # part1 has no process group


def test_part3_flags_and_lines():
    r = _run("part3/main.py", "--master-ip", "127.0.0.1:29517", "--rank", "0", "--num-nodes", "1")
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "master_ip = 127.0.0.1:29517 and nodes = 1" in out
    assert re.search(r"^\[\d+\] world_size = 1, rank = 0, backend=none$", out, re.M)
    assert "Epoch 0 triggered" in out
    assert "BatchNorm2d-2" in out
    assert re.search(r"^Average execution time is : [\d.e-]+ seconds$", out, re.M)


def test_reference_flag_defaults():
    from distributed_machine_learning_amd.cli import build_parser
    from distributed_machine_learning_amd.config import part_defaults
    ns = build_parser(part_defaults("part2a")).parse_args([])
    assert (ns.master_ip, ns.rank, ns.num_nodes) == ("127.0.1.1:8000", 0, 1)
    assert ns.batch_size == 64 and ns.sync == "gather_scatter" and not ns.batch_norm
    assert part_defaults("part3").batch_norm and part_defaults("part1").batch_size == 256


def test_rendezvous_failure_exits_nonzero():
    cmd = [sys.executable, os.path.join(ROOT, "part2/2b/main.py"), "--device", "cpu",
           "--master-ip", "127.0.0.1:29518", "--rank", "1", "--num-nodes", "2",
           "--comm-timeout", "3", "--no-summary"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0
    assert "could not join the process group" in r.stderr


def test_bench_self_launch_dry_run():
    """`bench.py --gpus N` without a launcher environment starts its own N ranks (the parent makes
    no GPU call, so this runs anywhere): each child must see the right rank triple and one shared
    rendezvous address."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--dry-launch"], capture_output=True, text=True, timeout=120, cwd=ROOT,
                       env=env)
    assert r.returncode == 0, r.stderr
    recs = sorted((json.loads(ln) for ln in r.stdout.splitlines() if ln.strip()),
                  key=lambda d: d["rank"])
    assert [d["rank"] for d in recs] == [0, 1] and [d["local_rank"] for d in recs] == [0, 1]
    assert all(d["world_size"] == 2 and d["self_launched"] for d in recs)
    assert recs[0]["master"] == recs[1]["master"] and recs[0]["master"].startswith("127.0.0.1:")
    # the communicator id rank 0 publishes through the native store reaches rank 1 unchanged
    assert recs[0]["bootstrap_id"] and recs[0]["bootstrap_id"] == recs[1]["bootstrap_id"]
    assert not recs[0]["agent_store"]


def test_bench_under_torchrun_dry_run():
    """The harness' multi-GPU command line: `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N`.  Every rank
    must take its rank triple from the launcher and exchange the communicator id through the
    launcher's own store (MASTER_PORT is already bound by it)."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4",
                        "--steps", "3", "--warmup", "1", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr
    recs = sorted((json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")),
                  key=lambda d: d["rank"])
    assert [d["rank"] for d in recs] == [0, 1, 2, 3]
    assert [d["local_rank"] for d in recs] == [0, 1, 2, 3]
    assert all(d["world_size"] == 4 and not d["self_launched"] and d["agent_store"] for d in recs)
    assert all(d["master"] == f"127.0.0.1:{port}" for d in recs)
    assert recs[0]["bootstrap_id"] and len({d["bootstrap_id"] for d in recs}) == 1


def test_bench_launcher_env_must_match_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--dry-launch"], capture_output=True, text=True, timeout=120, cwd=ROOT,
                       env=env)
    assert r.returncode != 0 and "does not match --gpus" in r.stderr


def test_cpu_device_has_no_multi_rank_transport():
    """The only transport is RCCL: a multi-rank run on the CPU device says so and exits non-zero."""
    cmd = [sys.executable, os.path.join(ROOT, "part2/2b/main.py"), "--device", "cpu",
           "--master-ip", "127.0.0.1:29519", "--rank", "0", "--num-nodes", "2", "--no-summary"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0
    assert "only transport is RCCL" in r.stderr


def test_scaling_study_table():
    """tools/scaling_study.py: efficiency is per-GPU throughput against the mode's first row, the
    reference column comes from the report's published s/iter."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("scaling_study",
                                                  os.path.join(ROOT, "tools", "scaling_study.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def rec(n, ms, overlap=None, vol=None, sync=None):
        return {"n_gpus": n, "ms_per_step": ms, "s_per_iter": ms / 1e3,
                "value": 256 * n / (ms / 1e3), "params_in_sync": sync,
                "comm_bytes_per_step": vol,
                "config": {"mode": "part3", "overlap": overlap}}

    text = mod.table([rec(1, 0.6), rec(4, 0.75, {"exposed_comm_ms": 0.15, "overlap_fraction": 0.5},
                                       {"tx": 55_400_000, "rx": 55_400_000}, True)])
    lines = text.splitlines()
    assert len(lines) == 4
    one, four = lines[2].split(), lines[3].split()
    assert one[0] == "part3" and one[1] == "1" and one[4] == "100.0%"
    assert four[1] == "4" and four[4] == "80.0%"          # 0.6 / 0.75
    assert "0.150 ms" in lines[3] and "50 %" in lines[3] and "55.4" in lines[3] and "yes" in lines[3]
    assert "0.838" in lines[3]                            # the report's part3 time at four nodes
    assert lines[3].rstrip().endswith("1118x")            # 0.83813 s against 0.75 ms


def _bench_cpu(*extra):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "part1", "--device",
                           "cpu", "--backend", "torch", "--steps", "2", "--warmup", "1", "--batch-size",
                           "16", *extra], capture_output=True, text=True, timeout=600, cwd=ROOT)


def test_bench_cpu_part1_prints_a_record():
    """The reference's part1 is a CPU run of stock PyTorch (/root/reference/part1/main.py:15-16):
    bench.py reproduces that set-up for the baseline table instead of refusing a host without GPU."""
    import json
    r = _bench_cpu("--windows", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["config"]["device"] == "cpu" and rec["config"]["backend"] == "torch"
    assert rec["dtype"] == "fp32" and rec["n_gpus"] == 1 and rec["value"] > 0
    assert rec["extras_error"] is None and rec["windows"]["n"] == 1
    # a CPU run with the native backend or a communicator is refused up front
    assert _bench_cpu("--force-comm").returncode != 0


def test_bench_keeps_its_record_when_the_extras_overrun_their_deadline():
    """Work after the measured region (further windows, cross-rank checks) that does not return in
    time must not cost the run its record: rank 0 prints it with the phase it was in, exit code 4."""
    import json
    r = _bench_cpu("--windows", "50", "--extras-deadline", "0.05")
    assert r.returncode == 4, (r.returncode, r.stderr[-2000:])
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert rec["ms_per_step"] > 0 and "further windows" in rec["extras_error"]
