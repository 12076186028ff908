"""The fp32 engine (--dtype fp32, the reference's precision) against the stock-torch model
built from the same state_dict and evaluated in float64: logits, loss, every parameter
gradient, optimizer steps, evaluation and checkpoints, and the reference's 40-iteration
protocol (/root/reference/part1/main.py:30-55: lr 0.1, batch 256, loss printed at 20 and 40)
for both engines."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda", 0)


def _mk(batch_norm, batch, dtype="fp32", lr=0.01, **kw):
    from distributed_machine_learning_amd.config import TrainConfig
    from distributed_machine_learning_amd.engine.native_backend import NativeBackend
    from distributed_machine_learning_amd.models import VGG
    torch.manual_seed(1234)
    ref = VGG("VGG11", batch_norm).to(DEV)
    comm = kw.pop("comm", None)
    cfg = TrainConfig(batch_size=batch, batch_norm=batch_norm, dtype=dtype, backend="native", lr=lr,
                      **kw)
    nb = NativeBackend(cfg, DEV, comm, init_state=ref.state_dict())
    return nb, ref


def _batch(batch, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 3, 32, 32, generator=g)
    if dtype == torch.bfloat16:
        x = x.to(torch.bfloat16).float()
    t = torch.randint(0, 10, (batch,), generator=g)
    x4 = torch.zeros(batch, 32, 32, 4, dtype=dtype)
    x4[..., :3] = x.permute(0, 2, 3, 1).to(dtype)
    return x.to(DEV), x4.to(DEV).contiguous(), t.to(DEV)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("batch_norm", [False, True])
def test_fp32_engine_matches_float64_autograd(batch_norm):
    """One forward + backward at batch 32: logits, loss and all 18 / 34 parameter gradients at
    1e-4 of float64 autograd.  (At this size no activation sits within fp32 rounding of a ReLU
    boundary; the batch-256 test below removes that effect instead of relying on it.)"""
    batch = 32
    nb, ref = _mk(batch_norm, batch)
    assert nb.fp32 and nb.arena.dtype == torch.float32 and nb.loader_layout == "nhwc4_fp32"
    x, x4, t = _batch(batch)
    s = int(torch.cuda.current_stream().cuda_stream)
    nb.engine.forward_backward(x4.data_ptr(), t.data_ptr(), batch, s)
    torch.cuda.synchronize()
    ref64 = ref.double()
    logits = ref64(x.double())
    loss = F.cross_entropy(logits, t)
    loss.backward()
    assert _rel(nb.logits(batch), logits.detach()) < 1e-4
    assert abs(float(nb.workspace[nb._loss_off]) - float(loss.detach())) < 1e-5
    named = dict(ref64.named_parameters())
    problems = []
    for slot in nb.slots:
        g = nb.grads[slot.offset:slot.offset + slot.numel].view(*slot.shape)
        if len(slot.shape) == 4:
            g = g.permute(0, 3, 1, 2)
        twin = named.get(slot.name.replace(".bias", ".weight"))
        if batch_norm and slot.name.endswith(".bias") and twin is not None and twin.dim() == 4:
            # conv bias under BatchNorm: exactly zero here, rounding noise in autograd
            assert float(g.abs().max()) == 0.0, slot.name
            continue
        r = _rel(g, named[slot.name].grad)
        if not r < 1e-4:
            problems.append(f"{slot.name}: {r:.3g}")
    assert not problems, "; ".join(problems)


@pytest.mark.parametrize("batch", [32, 256])
@pytest.mark.parametrize("batch_norm", [False, True])
def test_fp32_engine_layer_by_layer(batch_norm, batch):
    """Every kernel of the benchmarked step (batch 256, default plans) in place, at 1e-4: each
    layer's forward output against the float64 op on the engine's own input of that layer, and
    each backward kernel against the float64 formula on the engine's own tensors, taking the
    ReLU / pooling decisions the engine took (an activation within fp32 rounding of zero may
    fall on either side in another arithmetic; that is not what is being checked)."""
    nb, ref = _mk(batch_norm, batch)
    x, x4, t = _batch(batch)
    s = int(torch.cuda.current_stream().cuda_stream)
    nb.engine.forward_backward(x4.data_ptr(), t.data_ptr(), batch, s)
    torch.cuda.synchronize()
    table = nb.engine.layer_table
    saves = nb.engine.bn_save_offsets
    B = batch

    def act(off, hw, c):        # NHWC arena tensor as float64 NCHW
        return nb.arena[off:off + B * hw * hw * c].view(B, hw, hw, c).double().permute(0, 3, 1, 2)

    convs = [m for m in ref.layers if isinstance(m, torch.nn.Conv2d)]
    bns = [m for m in ref.layers if isinstance(m, torch.nn.BatchNorm2d)]
    grads = {}
    for slot in nb.slots:
        g = nb.grads[slot.offset:slot.offset + slot.numel].view(*slot.shape).double()
        grads[slot.name] = g.permute(0, 3, 1, 2) if len(slot.shape) == 4 else g
    names = [k for k, _ in ref.named_parameters()]
    conv_names = [k[:-7] for k in names if k.endswith(".weight") and grads[k].dim() == 4]
    bn_names = [k[:-7] for k in names if k.endswith(".weight") and grads[k].dim() == 1]
    problems = []

    def expect(name, got, want, tol=1e-4):
        r = _rel(got, want)
        if not r < tol:
            problems.append(f"{name}: rel err {r:.3g}")

    # ------------------------------------------------------------------ forward
    inp = x.double()
    layer_in = []
    rebuilt, sure_windows = {}, {}
    for L, (cin, cout, hw, pool, pre_off, out_off, pool_off, gout_off) in enumerate(table):
        layer_in.append(inp)
        w = convs[L].weight.detach().double()
        y = F.conv2d(inp, w, convs[L].bias.detach().double(), padding=1)
        out = act(out_off, hw, cout)
        if batch_norm:
            pre = act(pre_off, hw, cout)
            expect(f"conv{L} fwd", pre, y)
            mean = pre.mean((0, 2, 3))
            var = pre.var((0, 2, 3), unbiased=False)
            so = saves[L]
            expect(f"bn{L} mean", nb.workspace[so:so + cout], mean, 1e-5)
            expect(f"bn{L} rstd", nb.workspace[so + cout:so + 2 * cout], (var + 1e-5).rsqrt(), 1e-5)
            z = F.batch_norm(pre, None, None, bns[L].weight.detach().double(),
                             bns[L].bias.detach().double(), True, 0.1, 1e-5)
            if pool:
                # BatchNorm + ReLU + pool run fused: the full-resolution activation is never
                # stored.  It is rebuilt here; windows whose winner (or whose sign) another
                # arithmetic could decide differently are left out of the backward comparison.
                out = z.relu()
                rebuilt[L] = z
                win = z.view(B, cout, hw // 2, 2, hw // 2, 2).permute(0, 1, 2, 4, 3, 5)
                top = win.reshape(B, cout, hw // 2, hw // 2, 4).topk(2, dim=-1).values
                ok = (top[..., 0].abs() > 1e-4) & ((top[..., 0] - top[..., 1] > 1e-4) | (top[..., 0] < 0))
                sure_windows[L] = ok.repeat_interleave(2, 2).repeat_interleave(2, 3)
                expect(f"bn{L} + pool fwd", act(pool_off, hw // 2, cout), F.max_pool2d(out, 2))
                inp = act(pool_off, hw // 2, cout)
                continue
            sure = z.abs() > 1e-5
            expect(f"bn{L} fwd", out * sure, z.relu() * sure)
        else:
            sure = y.abs() > 1e-5
            expect(f"conv{L} fwd", out * sure, y.relu() * sure)
        inp = out
        if pool:
            nxt = act(pool_off, hw // 2, cout)
            if not torch.equal(nxt, F.max_pool2d(out, 2)):
                problems.append(f"pool{L} fwd differs")
            inp = nxt
    fc_in = inp.flatten(1)
    logits = nb.logits(B).double()
    expect("fc fwd", logits, F.linear(fc_in, ref.fc1.weight.detach().double(),
                                      ref.fc1.bias.detach().double()))

    # ----------------------------------------------------------------- backward
    dl = (torch.softmax(logits, 1) - F.one_hot(t, 10).double()) / B
    expect("fc dW", grads["fc1.weight"], dl.t() @ fc_in)
    expect("fc db", grads["fc1.bias"], dl.sum(0))
    arriving = (dl @ ref.fc1.weight.detach().double()).view(B, -1, 1, 1)
    for L in range(len(table) - 1, -1, -1):
        cin, cout, hw, pool, pre_off, out_off, pool_off, gout_off = table[L]
        if L in rebuilt:
            # the gradient goes to each window's winner where the engine's pooled output says
            # the window was alive
            _, idx = F.max_pool2d(rebuilt[L], 2, return_indices=True)
            g = F.max_unpool2d(arriving * (act(pool_off, hw // 2, cout) > 0), idx, 2)
        else:
            out = act(out_off, hw, cout)
            g = arriving
            if pool:
                o = out.clone().requires_grad_(True)
                F.max_pool2d(o, 2).backward(g)
                g = o.grad
            g = g * (out > 0)                   # the ReLU decisions the engine's forward took
        if batch_norm:
            pre = act(pre_off, hw, cout)
            so = saves[L]
            mean = nb.workspace[so:so + cout].double().view(1, -1, 1, 1)
            rstd = nb.workspace[so + cout:so + 2 * cout].double().view(1, -1, 1, 1)
            gam = bns[L].weight.detach().double().view(1, -1, 1, 1)
            xh = (pre - mean) * rstd
            n = B * hw * hw
            dbeta, dgamma = g.sum((0, 2, 3)), (g * xh).sum((0, 2, 3))
            expect(f"bn{L} dbeta", grads[bn_names[L] + ".bias"], dbeta)
            expect(f"bn{L} dgamma", grads[bn_names[L] + ".weight"], dgamma)
            g = gam * rstd * (g - dbeta.view(1, -1, 1, 1) / n - xh * dgamma.view(1, -1, 1, 1) / n)
            assert float(grads[conv_names[L] + ".bias"].abs().max()) == 0.0
        else:
            expect(f"conv{L} db", grads[conv_names[L] + ".bias"], g.sum((0, 2, 3)))
        gout = act(gout_off, hw, cout)
        if L in sure_windows:
            expect(f"layer{L} dL/d(conv out)", gout * sure_windows[L], g * sure_windows[L])
        else:
            expect(f"layer{L} dL/d(conv out)", gout, g)
        # from here on the engine's own gradient tensor: each kernel is checked in isolation
        dw = torch.nn.grad.conv2d_weight(layer_in[L], (cout, cin, 3, 3), gout, padding=1)
        expect(f"conv{L} dW", grads[conv_names[L] + ".weight"], dw)
        if L > 0:
            arriving = torch.nn.grad.conv2d_input(layer_in[L].shape,
                                                  convs[L].weight.detach().double(), gout, padding=1)
    assert not problems, "; ".join(problems)


@pytest.mark.parametrize("batch_norm", [False, True])
def test_fp32_training_steps_match_torch_sgd(batch_norm):
    nb, ref = _mk(batch_norm, 64)
    ref = ref.double()
    opt = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    for step in range(5):
        x, x4, t = _batch(64, seed=step)
        opt.zero_grad()
        loss = F.cross_entropy(ref(x.double()), t)
        loss.backward()
        opt.step()
        got = float(nb.train_step(x4, t).item())
        # (with BatchNorm one activation on the other side of a ReLU moves later losses by ~1e-3
        # and the difference compounds over steps at this batch size)
        tol = (1e-2 if batch_norm else 2e-4) * max(1.0, abs(float(loss)))
        assert abs(got - float(loss.detach())) < tol, (step, got, float(loss.detach()))
    sd = nb.state_dict()
    for name, p in ref.state_dict().items():
        if "num_batches" in name:
            assert int(sd[name]) == int(p)
        elif "running" in name:     # means sit near zero: absolute, against the tensor's scale
            scale = max(1.0, float(p.abs().max()))
            assert float((sd[name].double() - p).abs().max()) < 2e-2 * scale, name
        elif p.dim() >= 2:
            assert _rel(sd[name], p) < (2e-2 if batch_norm else 1e-3), name


def test_fp32_eval_checkpoint_and_bit_reproducibility(tmp_path):
    from distributed_machine_learning_amd.models import VGG
    from distributed_machine_learning_amd.models.checkpoint import load_checkpoint, save_checkpoint
    a, _ = _mk(True, 32)
    b, _ = _mk(True, 32)
    for step in range(3):
        _, x4, t = _batch(32, seed=step)
        la, lb = a.train_step(x4, t).clone(), b.train_step(x4, t).clone()
        assert torch.equal(la, lb)
    assert torch.equal(a.params, b.params) and torch.equal(a.bn_buffers, b.bn_buffers)
    path = str(tmp_path / "ck.pt")
    save_checkpoint(path, a.state_dict(), ddp_prefix=True)
    plain = VGG("VGG11", True).to(DEV)
    plain.load_state_dict(load_checkpoint(path))
    plain.eval()
    x, x4, t = _batch(32, 7)
    with torch.no_grad():
        logits_ref = plain(x)
    loss, correct = a.eval_batch(x4, t)
    assert _rel(a.logits(32), logits_ref) < 1e-4
    assert abs(float(loss) - float(F.cross_entropy(logits_ref, t))) < 1e-4
    assert int(correct) == int((logits_ref.argmax(1) == t).sum())


@pytest.mark.parametrize("batch_norm", [False, True])
def test_fp32_wgrad_side_stream_matches_single_stream(batch_norm):
    """Weight gradients on the side stream (the default), with and without the bucket reducer,
    against the serialized single-stream schedule: the race detector of the fp32 engine.  Every
    kernel is deterministic, so any difference is a missing dependency."""
    from distributed_machine_learning_amd.parallel.comm import RcclCommunicator
    comm = RcclCommunicator("127.0.0.1:29660" if batch_norm else "127.0.0.1:29661", 0, 1, 0,
                            timeout_s=60)
    try:
        serial, _ = _mk(batch_norm, 64, wgrad_overlap=False)
        side, _ = _mk(batch_norm, 64, wgrad_overlap=True)
        ddp, _ = _mk(batch_norm, 64, wgrad_overlap=True, sync="ddp", comm=comm, bucket_cap_mb=2.0)
        ddp_serial, _ = _mk(batch_norm, 64, sync="ddp", comm=comm, bucket_cap_mb=2.0,
                            serialize_streams=True)
        assert side.engine.wgrad_overlap and not serial.engine.wgrad_overlap
        assert not ddp_serial.engine.wgrad_overlap
        for step in range(4):
            _, x4, t = _batch(64, seed=step)
            want = float(serial.train_step(x4, t).item())
            for other in (side, ddp, ddp_serial):
                assert float(other.train_step(x4, t).item()) == want
        comm.guard_sync(30)
        for other in (side, ddp, ddp_serial):
            assert torch.equal(other.params, serial.params)
            assert torch.equal(other.bn_buffers, serial.bn_buffers)
        ddp.close()
        ddp_serial.close()
    finally:
        comm.destroy()


@pytest.mark.parametrize("sync", ["gather_scatter", "all_reduce", "ddp"])
def test_fp32_sync_modes_world1(sync):
    from distributed_machine_learning_amd.parallel.comm import RcclCommunicator
    port = 29650 + ["gather_scatter", "all_reduce", "ddp"].index(sync)
    comm = RcclCommunicator(f"127.0.0.1:{port}", 0, 1, 0, timeout_s=60)
    try:
        base, _ = _mk(True, 32)
        nb, _ = _mk(True, 32, sync=sync, comm=comm, bucket_cap_mb=2.0)
        for step in range(3):
            _, x4, t = _batch(32, seed=step)
            assert float(base.train_step(x4, t).item()) == float(nb.train_step(x4, t).item())
        comm.guard_sync(30)
        assert torch.equal(nb.params, base.params) and torch.equal(nb.bn_buffers, base.bn_buffers)
        nb.close()
    finally:
        comm.destroy()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("part", ["part1", "part3"])
def test_reference_protocol_40_iterations_tracks_fp32_torch(part, dtype):
    """The reference's run: 40 iterations at lr 0.1, momentum 0.9, weight decay 1e-4, batch 256
    (part1: no BatchNorm; part3: BatchNorm), loss read at iterations 20 and 40 - here on fixed
    synthetic batches, against stock torch in fp32 from the same initial weights."""
    bn = part == "part3"
    nb, ref = _mk(bn, 256, dtype=dtype, lr=0.1)
    opt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    act = torch.float32 if dtype == "fp32" else torch.bfloat16
    batches = [_batch(256, seed=100 + i, dtype=act) for i in range(8)]
    got, want = {}, {}
    for it in range(1, 41):
        x, x4, t = batches[(it - 1) % len(batches)]
        opt.zero_grad()
        loss = F.cross_entropy(ref(x), t)
        loss.backward()
        opt.step()
        mine = nb.train_step(x4, t)
        if it in (20, 40):
            got[it], want[it] = float(mine.item()), float(loss.detach())
    for it in (20, 40):
        assert abs(got[it] - want[it]) < 0.02 * abs(want[it]), (part, dtype, got, want)
