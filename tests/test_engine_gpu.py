"""The native engine against the stock-torch fp32 model built from the same state_dict:
forward logits, loss, every parameter gradient, several optimizer steps, evaluation,
checkpoint round trip, the W=1 RCCL paths of all three sync modes and hipGraph replay."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda", 0)


def _mk(batch_norm, batch=32, sync="none", comm=None, **kw):
    from distributed_machine_learning_amd.config import TrainConfig
    from distributed_machine_learning_amd.engine.native_backend import NativeBackend
    from distributed_machine_learning_amd.models import VGG
    torch.manual_seed(1234)
    ref = VGG("VGG11", batch_norm).to(DEV)
    cfg = TrainConfig(batch_size=batch, batch_norm=batch_norm, sync=sync, dtype="bf16",
                      backend="native", lr=0.05, **kw)
    nb = NativeBackend(cfg, DEV, comm, init_state=ref.state_dict())
    return nb, ref


def _batch(batch, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 3, 32, 32, generator=g).to(torch.bfloat16)
    t = torch.randint(0, 10, (batch,), generator=g)
    x4 = torch.zeros(batch, 32, 32, 4, dtype=torch.bfloat16)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    return x.float().to(DEV), x4.to(DEV).contiguous(), t.to(DEV)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-12))


@pytest.mark.parametrize("batch_norm", [False, True])
def test_forward_backward_matches_torch(batch_norm):
    nb, ref = _mk(batch_norm)
    x, x4, t = _batch(32)
    ref.train()
    logits_ref = ref(x)
    loss_ref = F.cross_entropy(logits_ref, t)
    loss_ref.backward()
    s = int(torch.cuda.current_stream().cuda_stream)
    nb.engine.forward_backward(x4.data_ptr(), t.data_ptr(), 32, s)
    torch.cuda.synchronize()
    assert _rel(nb.logits(32), logits_ref.detach()) < 3e-2
    assert abs(float(nb.workspace[nb._loss_off]) - float(loss_ref)) < 2e-2
    grads = {}
    for slot in nb.slots:
        g = nb.grads[slot.offset:slot.offset + slot.numel].view(*slot.shape)
        grads[slot.name] = g.permute(0, 3, 1, 2) if len(slot.shape) == 4 else g
    worst = 0.0
    for name, p in ref.named_parameters():
        r = _rel(grads[name], p.grad)
        worst = max(worst, r)
        if batch_norm and name.endswith(".bias") and p.dim() == 1 and \
                name.replace(".bias", ".weight") in grads and grads[name.replace(".bias", ".weight")].dim() == 4:
            continue     # conv bias under BatchNorm: true gradient is ~0, pure rounding noise
        assert r < 6e-2, f"{name}: relative gradient error {r:.3g}"
    print("worst relative gradient error", worst)


@pytest.mark.parametrize("batch_norm", [False, True])
def test_training_steps_track_torch(batch_norm):
    nb, ref = _mk(batch_norm)
    opt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    losses_n, losses_r = [], []
    for step in range(5):
        x, x4, t = _batch(32, seed=step)
        opt.zero_grad()
        loss = F.cross_entropy(ref(x), t)
        loss.backward()
        opt.step()
        losses_r.append(float(loss))
        losses_n.append(float(nb.train_step(x4, t).item()))
    for a, b in zip(losses_n, losses_r):
        assert abs(a - b) < 0.08 * max(1.0, abs(b)), (losses_n, losses_r)
    sd = nb.state_dict()
    for name, p in ref.state_dict().items():
        if "num_batches" in name:
            assert int(sd[name]) == int(p)
            continue
        assert _rel(sd[name], p) < 5e-2, name


def test_eval_and_checkpoint_roundtrip(tmp_path):
    from distributed_machine_learning_amd.models import VGG
    from distributed_machine_learning_amd.models.checkpoint import load_checkpoint, save_checkpoint
    nb, ref = _mk(True)
    x, x4, t = _batch(32, 5)
    for _ in range(2):
        nb.train_step(x4, t)
    path = str(tmp_path / "ck.pt")
    save_checkpoint(path, nb.state_dict(), ddp_prefix=True)
    plain = VGG("VGG11", True).to(DEV)
    plain.load_state_dict(load_checkpoint(path))
    plain.eval()
    with torch.no_grad():
        logits_ref = plain(x)
    loss, correct = nb.eval_batch(x4, t)
    assert _rel(nb.logits(32), logits_ref) < 3e-2
    assert abs(float(loss) - float(F.cross_entropy(logits_ref, t))) < 3e-2
    assert abs(int(correct) - int((logits_ref.argmax(1) == t).sum())) <= 2
    # partial batch through an engine built for 32
    loss16, _ = nb.eval_batch(x4[:16].contiguous(), t[:16].contiguous())
    assert abs(float(loss16) - float(F.cross_entropy(logits_ref[:16], t[:16]))) < 3e-2
    # load back: identical parameters
    nb2, _ = _mk(True)
    nb2.load_state_dict(load_checkpoint(path))
    assert torch.equal(nb2.params, nb.params)
    assert torch.equal(nb2.bn_buffers, nb.bn_buffers)


@pytest.mark.parametrize("sync", ["gather_scatter", "all_reduce", "ddp"])
def test_sync_modes_world1_match_no_sync(sync):
    """With one rank every strategy must be the identity on the gradients: exercises the
    native TCP store, the RCCL communicator, grouped send/recv, the slab-sum kernel and
    the stream-overlapped bucket reducer end to end."""
    from distributed_machine_learning_amd.parallel.comm import RcclCommunicator
    port = 29600 + ["gather_scatter", "all_reduce", "ddp"].index(sync)
    comm = RcclCommunicator(f"127.0.0.1:{port}", 0, 1, 0, timeout_s=60)
    try:
        base, _ = _mk(False)
        nb, _ = _mk(False, sync=sync, comm=comm)
        for step in range(3):
            _, x4, t = _batch(32, seed=step)
            la = float(base.train_step(x4, t).item())
            lb = float(nb.train_step(x4, t).item())
            assert abs(la - lb) < 1e-3
        comm.guard_sync(30)
        assert _rel(nb.params, base.params) < 1e-3
        nb.close()
    finally:
        comm.destroy()


def test_graph_replay_matches_eager():
    eager, _ = _mk(False)
    graph, _ = _mk(False, use_graph=True)
    for step in range(4):
        _, x4, t = _batch(32, seed=step)
        la = float(eager.train_step(x4, t).item())
        lb = float(graph.train_step(x4, t).item())
        assert abs(la - lb) < 1e-3, (step, la, lb)
    assert graph.engine.has_graph
    assert _rel(graph.params, eager.params) < 1e-3


def test_cli_on_gpu_prints_reference_lines(capsys):
    from distributed_machine_learning_amd.cli import main
    res = main(["--iters", "40", "--batch-size", "64", "--train-size", "4096", "--test-size", "256",
                "--no-summary"], part="part1")
    out = capsys.readouterr().out
    assert "Loss at 20th batch is" in out and "Loss at 40th batch is" in out
    assert "Average execution time is  :" in out
    assert "Test set: Average loss:" in out
    assert res["avg_s_per_iter"] > 0
