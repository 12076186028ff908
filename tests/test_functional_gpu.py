"""The autograd layer API (ops/functional.py, models/native_vgg.py, optim/fused_sgd.py):
an nn.Module assembled from the native ops must agree with the C++ engine (same kernels,
different driver) and with the stock fp32 torch model built from the same state_dict."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda", 0)


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
def test_native_module_matches_engine_and_torch(batch_norm):
    from distributed_machine_learning_amd.config import TrainConfig
    from distributed_machine_learning_amd.engine.native_backend import NativeBackend
    from distributed_machine_learning_amd.models import VGG, NativeVGG
    from distributed_machine_learning_amd.ops import functional as NF
    B = 32
    torch.manual_seed(7)
    ref = VGG("VGG11", batch_norm).to(DEV)
    net = NativeVGG("VGG11", batch_norm).to(DEV)
    net.load_state_dict(ref.state_dict())
    cfg = TrainConfig(batch_size=B, batch_norm=batch_norm, dtype="bf16", backend="native", lr=0.01)
    eng = NativeBackend(cfg, DEV, None, init_state=ref.state_dict())
    x, x4, t = _batch(B)

    logits = net(x4)
    loss = NF.softmax_cross_entropy(logits, t)
    loss.backward()
    with torch.no_grad():
        ref_loss = F.cross_entropy(ref(x), t)
    assert abs(float(loss.detach()) - float(ref_loss)) < 3e-2 * max(1.0, float(ref_loss))

    s = int(torch.cuda.current_stream().cuda_stream)
    eng.engine.forward_backward(x4.data_ptr(), t.data_ptr(), B, s)
    torch.cuda.synchronize()
    # with BatchNorm the engine pools the unrounded activation (fused kernel), the module
    # pools the stored bf16 one: equal up to bf16 rounding of the pooled values
    assert abs(float(eng.workspace[eng._loss_off]) - float(loss.detach())) < (1e-2 if batch_norm else 1e-4)
    grads = dict(net.named_parameters())
    for slot in eng.slots:
        g = eng.grads[slot.offset:slot.offset + slot.numel].view(*slot.shape)
        if len(slot.shape) == 4:
            g = g.permute(0, 3, 1, 2)
        mine = grads[slot.name].grad
        if batch_norm and slot.name.endswith(".bias") and mine.dim() == 1 and \
                slot.name.replace(".bias", ".weight") in grads and \
                grads[slot.name.replace(".bias", ".weight")].dim() == 4:
            continue        # conv bias under BatchNorm: zero in exact arithmetic, noise in practice
        # Same kernels, but with BatchNorm the engine takes its statistics from the conv
        # accumulators and pools the unrounded activation, the module from the stored bf16
        # tensors; with 32-image batches (128 samples per channel in the last layers) the
        # end-to-end gradient is that sensitive to 1e-3 differences in the statistics.  The
        # tight per-layer comparison is test_engine_layerwise_against_torch.
        assert _rel(mine, g) < (0.35 if batch_norm else 2e-2), slot.name
    if batch_norm:
        bn = [m for m in net.layers if isinstance(m, torch.nn.BatchNorm2d)][0]
        assert int(bn.num_batches_tracked) == 1


def test_fused_sgd_trains_native_module():
    from distributed_machine_learning_amd.models import NativeVGG
    from distributed_machine_learning_amd.ops import functional as NF
    from distributed_machine_learning_amd.optim import FusedSGD
    torch.manual_seed(3)
    net = NativeVGG("VGG11", True).to(DEV)
    twin = NativeVGG("VGG11", True).to(DEV)
    twin.load_state_dict(net.state_dict())
    opt = FusedSGD(net, lr=0.01, momentum=0.9, weight_decay=1e-4)
    ref_opt = torch.optim.SGD(twin.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    _, x4, t = _batch(64, seed=1)
    first = None
    for _ in range(10):
        opt.zero_grad()
        loss = NF.softmax_cross_entropy(net(x4), t)
        loss.backward()
        opt.step()
        ref_opt.zero_grad()
        ref_loss = NF.softmax_cross_entropy(twin(x4), t)
        ref_loss.backward()
        ref_opt.step()
        first = float(loss) if first is None else first
        assert abs(float(loss) - float(ref_loss)) < 5e-2 * max(1.0, float(ref_loss))
    assert float(loss) < first - 0.2, (first, float(loss))        # it learns the fixed batch
    # same trajectory as torch.optim.SGD driving the same kernels (atomics: rounding-level drift)
    for (n, a), (_, b) in zip(net.named_parameters(), twin.named_parameters()):
        if a.dim() >= 2:
            assert _rel(a, b) < 5e-2, n
    assert int(NF.top1_correct(net(x4).detach(), t)) >= 0
