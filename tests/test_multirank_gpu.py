"""The native engine with MORE THAN ONE RANK on a single-GPU machine.

RCCL refuses two ranks on one device, so these tests run the ranks over the native loopback
test transport (csrc/comm/loopback_comm.cpp): same C++ engine, bucket reducer, manual sync,
event choreography and bench self-checks as an RCCL run, with only the collective calls
themselves replaced.  Reference behaviour being checked: identical parameters on every rank
(sanity check 2 of /root/reference/group25.pdf p3) and mean-reduced data parallelism equal
to the single-process run of the global batch (/root/reference/part1/main.py vs part2/part3).
"""
import json
import os
import subprocess
import sys

import pytest
import torch

from distributed_machine_learning_amd.launch import launch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(__file__)
ROOT = os.path.abspath(os.path.join(HERE, ".."))
WORKER = os.path.join(HERE, "_sync_worker.py")
NATIVE = ["--device", "cuda", "--backend", "native", "--dtype", "bf16", "--transport", "loopback"]


def _run(tmp_path, tag, nproc, extra, dtype="bf16"):
    out = str(tmp_path / tag)
    base = [a if a != "bf16" else dtype for a in NATIVE]
    if nproc == 1:
        rc = subprocess.call([sys.executable, WORKER, "--out", out] + base + extra, timeout=300)
    else:
        rc = launch(WORKER, ["--out", out] + base + extra, nproc, timeout_s=300)
    assert rc == 0, tag
    return [torch.load(f"{out}.rank{r}", weights_only=True) for r in range(nproc)]


def _same(a, b):
    bad = [(k, float((a[k].double() - b[k].double()).abs().max())) for k in a
           if not torch.equal(a[k], b[k])]
    if bad:
        print("tensors that differ (max abs difference):", bad)
    return not bad


def _rel(a, b, keys=None):
    worst = 0.0
    for k in (keys or a):
        if a[k].dim() < 2:
            continue
        d = float((a[k].double() - b[k].double()).norm() / b[k].double().norm().clamp_min(1e-12))
        worst = max(worst, d)
    return worst


def test_loopback_transport_collectives():
    rc = launch(os.path.join(HERE, "_collective_worker.py"), ["--transport", "loopback"], 3,
                timeout_s=240)
    assert rc == 0


@pytest.mark.parametrize("sync", ["gather_scatter", "all_reduce", "ddp"])
def test_two_ranks_mean_match_single_process_global_batch(tmp_path, sync):
    single = _run(tmp_path, "single", 1, ["--sync", "none", "--batch-size", "16", "--iters", "4"])[0]
    ranks = _run(tmp_path, sync, 2, ["--sync", sync, "--grad-reduce", "mean", "--batch-size", "8",
                                     "--iters", "4", "--bucket-cap-mb", "4"])
    assert _same(ranks[0], ranks[1])
    # per-sample gradients are scaled by an exact power of two (1/8 vs 1/16), so bf16 roundings
    # agree and only the fp32 summation order differs
    assert _rel(ranks[0], single) < 1e-4


@pytest.mark.parametrize("sync", ["gather_scatter", "ddp"])
def test_four_ranks_mean_match_single_process_global_batch(tmp_path, sync):
    """The reference's own world size (four nodes, /root/reference/group25.pdf p1): four ranks of
    4 images reproduce the single-process run on the 16-image global batch."""
    single = _run(tmp_path, "single", 1, ["--sync", "none", "--batch-size", "16", "--iters", "3"])[0]
    ranks = _run(tmp_path, sync, 4, ["--sync", sync, "--grad-reduce", "mean", "--batch-size", "4",
                                     "--iters", "3", "--bucket-cap-mb", "4"])
    for r in ranks[1:]:
        assert _same(ranks[0], r)
    assert _rel(ranks[0], single) < 1e-4


@pytest.mark.parametrize("sync", ["gather_scatter", "ddp"])
def test_fp32_engine_two_ranks_match_single_process(tmp_path, sync):
    """The fp32 engine (the reference's precision) under the same protocol."""
    extra = ["--batch-norm"] if sync == "ddp" else []
    single = _run(tmp_path, "single", 1, ["--sync", "none", "--batch-size", "16", "--iters", "3"]
                  + ([] if sync == "ddp" else []), dtype="fp32")[0]
    ranks = _run(tmp_path, sync, 2, ["--sync", sync, "--grad-reduce", "mean", "--batch-size", "8",
                                     "--iters", "3", "--bucket-cap-mb", "4"] + extra, dtype="fp32")
    assert _same(ranks[0], ranks[1])
    if sync != "ddp":       # with BatchNorm the two half-batches normalise separately
        assert _rel(ranks[0], single) < 1e-4


def test_manual_modes_default_to_sum(tmp_path):
    gs = _run(tmp_path, "gs", 2, ["--sync", "gather_scatter", "--iters", "2", "--batch-size", "8"])
    ar = _run(tmp_path, "ar", 2, ["--sync", "all_reduce", "--iters", "2", "--batch-size", "8"])
    mean = _run(tmp_path, "arm", 2, ["--sync", "all_reduce", "--grad-reduce", "mean", "--iters", "2",
                                     "--batch-size", "8"])
    assert _same(gs[0], gs[1]) and _same(ar[0], ar[1])
    assert _rel(gs[0], ar[0]) < 1e-6            # same sums, formed in the same rank order
    assert _rel(mean[0], ar[0]) > 1e-4          # SUM is not MEAN


@pytest.mark.parametrize("rep", range(int(os.environ.get("DML_TEST_REPS", "1"))))
def test_ddp_batchnorm_two_ranks_overlap_variants_agree_bitwise(tmp_path, rep):
    """Buckets leaving from the side stream, the in-line last bucket and the fully serialized
    debug schedule must give the same bits on every rank, running statistics included (they
    ride in the first bucket from rank 0)."""
    common = ["--sync", "ddp", "--batch-norm", "--batch-size", "8", "--iters", "4",
              "--bucket-cap-mb", "2"]
    base = _run(tmp_path, "ddp", 2, common)
    inline = _run(tmp_path, "ddp_inline", 2, common + ["--inline-last-bucket", "1"])
    serial = _run(tmp_path, "ddp_serial", 2, common + ["--serialize-streams"])
    assert _same(base[0], base[1])
    assert _same(inline[0], inline[1]) and _same(serial[0], serial[1])
    assert _same(base[0], inline[0])
    assert _same(base[0], serial[0])
    assert any("running_mean" in k for k in base[0])


def test_ddp_bf16_wire_two_ranks(tmp_path):
    common = ["--sync", "ddp", "--batch-norm", "--batch-size", "8", "--iters", "3",
              "--bucket-cap-mb", "2"]
    f32 = _run(tmp_path, "f32", 2, common)
    b16 = _run(tmp_path, "b16", 2, common + ["--grad-comm-dtype", "bf16"])
    assert _same(b16[0], b16[1])
    assert not _same(b16[0], f32[0])
    assert _rel(b16[0], f32[0]) < 2e-2
    # the running statistics travel as fp32 next to the bf16 gradients
    for k in f32[0]:
        if "running" in k:      # means sit near zero: absolute, against the tensor's scale
            scale = max(1.0, float(f32[0][k].abs().max()))
            assert float((b16[0][k].double() - f32[0][k].double()).abs().max()) < 3e-2 * scale, k


def test_ddp_bf16_wire_without_batchnorm_two_ranks(tmp_path):
    """Without the BatchNorm staging tail a bucket's slice can have any length (the first one ends
    with fc1.bias, 10 floats): the bf16 part is cut at a multiple of four elements and the rest
    travels as fp32 in the same group."""
    common = ["--sync", "ddp", "--batch-size", "8", "--iters", "3", "--bucket-cap-mb", "2"]
    f32 = _run(tmp_path, "f32", 2, common)
    b16 = _run(tmp_path, "b16", 2, common + ["--grad-comm-dtype", "bf16"])
    assert _same(b16[0], b16[1])
    assert not _same(b16[0], f32[0])
    assert _rel(b16[0], f32[0]) < 2e-2


def test_mismatched_replicas_fail_loudly(tmp_path):
    """Rank 1 builds a different network: the construction-time check must stop the job
    (what DistributedDataParallel's constructor does for the reference, part3/main.py:137)."""
    script = str(tmp_path / "odd.py")
    with open(script, "w") as f:
        f.write("import sys, runpy\n"
                "if '--rank' in sys.argv and sys.argv[sys.argv.index('--rank') + 1] == '1':\n"
                "    sys.argv += ['--arch', 'VGG13']\n"
                f"sys.argv[0] = {WORKER!r}\n"
                f"runpy.run_path({WORKER!r}, run_name='__main__')\n")
    out = str(tmp_path / "odd")
    rc = launch(script, ["--out", out] + NATIVE + ["--sync", "all_reduce", "--iters", "1"], 2,
                timeout_s=240)
    assert rc != 0
    assert not os.path.exists(out + ".rank0")


def _bench(args, timeout=420, extra_env=None):
    env = dict(os.environ, **(extra_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args,
                         capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["part2a", "part2b", "part3"])
def test_bench_self_launches_two_ranks_and_proves_it(mode):
    rec = _bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--windows", "2", "--mode", mode,
                  "--batch-size", "32", "--transport", "loopback"])
    assert rec["n_gpus"] == 2 and rec["rccl_nranks"] == 2 and rec["transport"] == "loopback"
    assert rec["params_in_sync"] is True
    assert len(rec["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in rec["per_rank_ms_per_step"])
    assert rec["ms_per_step"] == pytest.approx(max(rec["per_rank_ms_per_step"]), rel=1e-3)
    assert rec["config"]["global_batch"] == 64 and rec["config"]["parallelism"] == "dp2"
    assert rec["config"]["overlap"] is not None and rec["config"]["overlap"]["comm_only_ms"] > 0
    assert rec["windows"]["n"] == 2
    plan = rec["config"]["sync_plan"]
    assert plan["mode"] == {"part2a": "gather_scatter", "part2b": "all_reduce", "part3": "ddp"}[mode]
    if mode == "part3":
        assert len(plan["buckets_mb"]) >= 3 and plan["inline_last_bucket"] is False
        assert plan["bucket_departure"] == "value"      # MI355X / ROCm 7 can wait for a value
    assert rec["extras_error"] is None
    # both convolution plan policies were timed on this job and the faster one ran
    pol = rec["config"]["plan_policy"]
    assert pol["how"] == "timed on this job" and set(pol["ms_per_step"]) == {"tall", "tall128", "streaming"}
    assert pol["chosen"] == min(pol["ms_per_step"], key=pol["ms_per_step"].get)
    # the numerical self-checks of a multi-rank run: against rank 0 alone on the global batch in
    # all three modes, and (bucketed mode) flag words against event pairs bit for bit
    assert rec["self_checks_failed"] is None
    dp = rec["dp_vs_single_rel_err"]
    assert set(dp) == {"ddp", "gather_scatter", "all_reduce"} and all(v < 1e-4 for v in dp.values())
    assert all(m["ranks_identical"] for m in rec["self_checks"]["data_parallel"]["modes"].values())
    assert rec["extras"]["fp32_ms_per_step"] > 0 and rec["extras"]["transport"]["rccl_version"] > 20000
    if mode == "part3":
        assert rec["handoff_bitwise_ok"] is True
        assert rec["extras"]["bf16_wire_ms_per_step"] > 0
    else:
        assert rec["handoff_bitwise_ok"] is None
    # what rank 0 moved per step, from the collectives issued inside the measured region: with
    # two ranks the root of gather/scatter receives and sends one model of gradients, a ring
    # all-reduce moves 2 (W-1)/W = 1 model each way (the report's per-node byte counts,
    # /root/reference/group25.pdf p4 and p6, follow the same arithmetic at four nodes)
    vol = rec["comm_bytes_per_step"]
    model_bytes = 9_225_610 * 4
    if mode == "part3":                 # + BatchNorm weights, + running statistics in a bucket
        model_bytes += (2 * 2752 + 2 * 2752) * 4
    assert vol["tx"] == pytest.approx(model_bytes, rel=1e-3)
    assert vol["rx"] == pytest.approx(model_bytes, rel=1e-3)
    per_step = {k: v / 4 for k, v in vol["calls"].items()}
    if mode == "part2a":
        assert per_step["gather"] == 18 and per_step["scatter"] == 18
    elif mode == "part2b":
        assert per_step["all_reduce"] == pytest.approx(18, abs=1)     # + the two barriers' tokens


def test_bench_two_ranks_with_every_fallback_forced():
    """A device without wait-value support (or a graph capture) falls back to event pairs for the
    hand-offs, the joins and the bucket departures, and the gather/scatter root can use a landing
    slab for its own copy: those paths must keep working."""
    env = {"DML_HANDOFF": "event", "DML_REDUCER_HANDOFF": "event", "DML_JOIN": "event",
           "DML_EVENT_SCOPE": "system", "DML_GATHER_IN_PLACE": "0"}
    for mode in ("part3", "part2a"):
        rec = _bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--windows", "1", "--mode", mode,
                      "--batch-size", "32", "--transport", "loopback"], extra_env=env)
        assert rec["params_in_sync"] is True and rec["rccl_nranks"] == 2
        assert rec["extras_error"] is None
        if mode == "part3":
            assert rec["config"]["sync_plan"]["bucket_departure"] == "event"


def test_bench_one_rank_self_checks_on_rccl():
    """The same checks forced on with one rank over RCCL itself (communicator, reducer, manual
    sync and all record fields; the data-parallel run then IS the single-process run)."""
    rec = _bench(["--steps", "5", "--warmup", "2", "--force-comm", "--self-checks", "on",
                  "--inline-last-bucket", "0", "--windows", "1"])
    assert rec["transport"] == "rccl" and rec["self_checks_failed"] is None
    assert rec["handoff_bitwise_ok"] is True
    assert all(v < 1e-6 for v in rec["dp_vs_single_rel_err"].values())
    assert rec["config"]["sync_plan"]["bucket_departure"] == "value"
    assert rec["comm_bytes_per_step"]["model"].startswith("ring")


def test_bench_one_rank_rccl_record_fields():
    rec = _bench(["--steps", "5", "--warmup", "2", "--force-comm", "--report-overlap"])
    assert rec["rccl_nranks"] == 1 and rec["transport"] == "rccl"
    assert rec["params_in_sync"] is True and len(rec["per_rank_ms_per_step"]) == 1
    assert rec["config"]["overlap"] is not None
    assert rec["windows"]["min_ms_per_step"] <= rec["windows"]["median_ms_per_step"]
    assert rec["config"]["sync_plan"]["inline_last_bucket"] is True


def test_scaling_study_runs_one_and_two_ranks(tmp_path):
    """tools/scaling_study.py drives bench.py over GPU counts and modes (here: the test transport,
    two ranks on one GPU) and prints the report's scalability table."""
    out = str(tmp_path / "records.json")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scaling_study.py"), "--gpus", "1",
                        "2", "--modes", "part2a", "part3", "--transport", "loopback", "--batch-size",
                        "32", "--steps", "4", "--warmup", "2", "--windows", "1", "--json", out],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("part")]
    assert [(x[0], x[1]) for x in rows] == [("part2a", "1"), ("part2a", "2"), ("part3", "1"), ("part3", "2")]
    assert rows[0][4] == "100.0%" and rows[2][4] == "100.0%"
    recs = json.load(open(out))
    assert len(recs) == 4 and recs[1]["params_in_sync"] is True and recs[3]["params_in_sync"] is True
    assert recs[3]["config"]["overlap"] is not None
