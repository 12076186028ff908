"""Numerics of every HIP kernel against a plain PyTorch fp32 reference of the same op.

Inputs are bf16-representable values; references are computed in fp32 from exactly
those values, so the only differences are accumulation order and the final bf16 rounding
of bf16 outputs (relative 2^-8).  Shapes cover the eight VGG-11 conv layers.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"

# (cin, cout, hw) of VGG-11 conv2..conv8 (conv1 has its own kernels), then the two extra
# shapes the VGG13/16/19 tables need
GENERIC_LAYERS = [(64, 128, 16), (128, 256, 8), (256, 256, 8), (256, 512, 4), (512, 512, 4),
                  (512, 512, 2), (64, 64, 32), (128, 128, 16)]


def native():
    from distributed_machine_learning_amd.ops import native as n
    n.lib()
    return n


def rand_bf16(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


def nchw(x_nhwc):
    return x_nhwc.float().permute(0, 3, 1, 2).contiguous()


def oihw(w_krsc):
    return w_krsc.float().permute(0, 3, 1, 2).contiguous()


def check(name, got, ref, rtol=1.5e-2, atol_scale=1.5e-2):
    got = got.float()
    ref = ref.float()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    scale = float(ref.abs().max().clamp_min(1e-6))
    err = (got - ref).abs()
    tol = atol_scale * scale * 0.25 + rtol * ref.abs()
    bad = err > tol
    if bool(bad.any()):
        idx = bad.nonzero()[:5].tolist()
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{bad.numel()} elements off; max err {float(err.max()):.4g} "
            f"(ref scale {scale:.4g}); first bad idx {idx}; got {got[bad][:5].tolist()} "
            f"ref {ref[bad][:5].tolist()}")


@pytest.mark.parametrize("batch", [2, 32])
@pytest.mark.parametrize("cin,cout,hw", GENERIC_LAYERS)
@pytest.mark.parametrize("relu", [True, False])
def test_conv3x3_fwd(batch, cin, cout, hw, relu):
    n = native()
    x = rand_bf16(batch, hw, hw, cin, seed=1)
    w = rand_bf16(cout, 3, 3, cin, scale=(2.0 / (9 * cin)) ** 0.5, seed=2)
    b = torch.randn(cout, device=DEV) * 0.1
    y = n.conv3x3_fwd(x, w, b, relu=relu)
    ref = F.conv2d(nchw(x), oihw(w), b, padding=1)
    if relu:
        ref = ref.relu()
    check(f"fwd {cin}->{cout}@{hw} B{batch}", nchw(y), ref)


def test_conv3x3_fwd_b256_conv4_and_conv8():
    n = native()
    for cin, cout, hw in [(256, 256, 8), (512, 512, 2), (64, 128, 16)]:
        x = rand_bf16(256, hw, hw, cin, seed=3)
        w = rand_bf16(cout, 3, 3, cin, scale=(2.0 / (9 * cin)) ** 0.5, seed=4)
        b = torch.randn(cout, device=DEV) * 0.1
        y = n.conv3x3_fwd(x, w, b, relu=True)
        ref = F.conv2d(nchw(x), oihw(w), b, padding=1).relu()
        check(f"fwd {cin}->{cout}@{hw} B256", nchw(y), ref)


@pytest.mark.parametrize("batch", [64, 96, 256])
@pytest.mark.parametrize("cin,cout,hw", [(128, 256, 8), (256, 512, 4), (512, 512, 2), (64, 64, 2)])
@pytest.mark.parametrize("plan", [(64, 64, 1), (64, 64, 3), (64, 64, 9), (128, 64, 1), (128, 128, 1)])
def test_conv3x3_position_major_tiles(batch, cin, cout, hw, plan):
    """Tiles of BM images at one spatial position (padded taps skipped as whole K steps)
    must give the same result as the pixel-major tiling, for full and ragged image blocks,
    with and without the tap split."""
    n = native()
    bm, bn, splits = plan
    if cout % bn:
        pytest.skip("tile wider than the layer")
    x = rand_bf16(batch, hw, hw, cin, seed=31)
    w = rand_bf16(cout, 3, 3, cin, scale=(2.0 / (9 * cin)) ** 0.5, seed=32)
    b = torch.randn(cout, device=DEV) * 0.1
    ws = torch.empty(9 * batch * hw * hw * cout, device=DEV)
    y = torch.empty(batch, hw, hw, cout, dtype=torch.bfloat16, device=DEV)
    s = int(torch.cuda.current_stream().cuda_stream)
    n.lib().conv3x3_igemm(x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), 0, ws.data_ptr(),
                          ws.numel(), batch, hw, cin, cout, n.EPI_BIAS_RELU, s, bm, bn, splits, 1)
    ref = F.conv2d(nchw(x), oihw(w), b, padding=1).relu()
    check(f"pm fwd {cin}->{cout}@{hw} B{batch} {plan}", nchw(y), ref)


@pytest.mark.parametrize("waves", [8, 9])
@pytest.mark.parametrize("pm", [0, 1])
@pytest.mark.parametrize("cin,cout,hw,batch", [(256, 256, 8, 64), (64, 128, 16, 33), (512, 512, 4, 128)])
def test_conv3x3_eight_wave_tiles(cin, cout, hw, batch, pm, waves):
    """128x128 tiles run by eight waves (two per SIMD), 3 and 4 LDS stages."""
    n = native()
    if pm and (hw > 8 or batch % 128):
        pytest.skip("position-major tiles need whole image blocks")
    x = rand_bf16(batch, hw, hw, cin, seed=41)
    w = rand_bf16(cout, 3, 3, cin, scale=(2.0 / (9 * cin)) ** 0.5, seed=42)
    b = torch.randn(cout, device=DEV) * 0.1
    y = torch.empty(batch, hw, hw, cout, dtype=torch.bfloat16, device=DEV)
    s = int(torch.cuda.current_stream().cuda_stream)
    n.lib().conv3x3_igemm(x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), 0, 0, 0, batch, hw,
                          cin, cout, n.EPI_BIAS, s, 128, 128, 1, pm, waves)
    check(f"8w fwd {cin}->{cout}@{hw} B{batch}", nchw(y), F.conv2d(nchw(x), oihw(w), b, padding=1))


@pytest.mark.parametrize("batch", [3, 32, 256])
@pytest.mark.parametrize("cin,cout,hw", [(64, 128, 16), (128, 256, 8), (256, 256, 8), (256, 512, 4),
                                         (512, 512, 4), (512, 512, 2), (128, 128, 16)])
@pytest.mark.parametrize("tile", [(256, 64, 20), (256, 64, 21), (128, 64, 20), (128, 64, 21), (128, 64, 25)])
@pytest.mark.parametrize("kind", ["fwd", "fwd_relu", "dgrad", "dgrad_mask"])
def test_conv3x3_tall_patch_tiles(batch, cin, cout, hw, tile, kind):
    """Tall patch-resident tiles (conv_patch.hip): one LDS copy of a tile's pixels serves the nine
    taps through the shift-invariant slot permutation, padding through the zero rows; forward with
    both epilogues and backward-data with and without the ReLU mask, ragged last tiles included."""
    n = native()
    bm, bn, waves = tile
    s = int(torch.cuda.current_stream().cuda_stream)
    x = rand_bf16(batch, hw, hw, cin, seed=71)
    w = rand_bf16(cout, 3, 3, cin, scale=(2.0 / (9 * cin)) ** 0.5, seed=72)
    if kind.startswith("fwd"):
        b = torch.randn(cout, device=DEV) * 0.1
        y = torch.full((batch, hw, hw, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
        epi = n.EPI_BIAS_RELU if kind == "fwd_relu" else n.EPI_BIAS
        n.lib().conv3x3_igemm(x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), 0, 0, 0, batch,
                              hw, cin, cout, epi, s, bm, bn, 1, 0, waves)
        ref = F.conv2d(nchw(x), oihw(w), b, padding=1)
        check(f"tall {kind} {cin}->{cout}@{hw} B{batch} {tile}", nchw(y),
              ref.relu() if kind == "fwd_relu" else ref)
    else:
        dy = rand_bf16(batch, hw, hw, cout, seed=73)
        wd = n.repack_dgrad(w)
        dx = torch.full((batch, hw, hw, cin), float("nan"), dtype=torch.bfloat16, device=DEV)
        mask = rand_bf16(batch, hw, hw, cin, seed=74).relu() if kind == "dgrad_mask" else None
        n.lib().conv3x3_igemm(dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), 0,
                              mask.data_ptr() if mask is not None else 0, 0, 0, batch, hw, cout, cin,
                              n.EPI_MASK if mask is not None else n.EPI_NONE, s, bm, bn, 1, 0, waves)
        ref = torch.nn.grad.conv2d_input((batch, cin, hw, hw), oihw(w), nchw(dy), padding=1)
        if mask is not None:
            ref = ref * (nchw(mask) > 0)
        check(f"tall {kind} {cin}->{cout}@{hw} B{batch} {tile}", nchw(dx), ref)


@pytest.mark.parametrize("batch", [5, 256])
@pytest.mark.parametrize("cin,cout,hw", [(128, 256, 8), (512, 256, 4), (512, 512, 2)])
@pytest.mark.parametrize("tile", [(256, 64), (128, 64)])
@pytest.mark.parametrize("splits", [2, 4])
def test_conv3x3_tall_tiles_chunk_split(batch, cin, cout, hw, tile, splits):
    """Tall tiles with the channel chunks divided between `splits` blocks per tile (fp32 slabs
    summed by the finishing kernel, which also applies the epilogue)."""
    n = native()
    bm, bn = tile
    if splits > cin // 64:
        pytest.skip("more splits than channel chunks")
    s = int(torch.cuda.current_stream().cuda_stream)
    x = rand_bf16(batch, hw, hw, cin, seed=81)
    w = rand_bf16(cout, 3, 3, cin, scale=(2.0 / (9 * cin)) ** 0.5, seed=82)
    b = torch.randn(cout, device=DEV) * 0.1
    ws = torch.empty(9 * batch * hw * hw * cout, device=DEV)
    y = torch.full((batch, hw, hw, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    n.lib().conv3x3_igemm(x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), 0, ws.data_ptr(),
                          ws.numel(), batch, hw, cin, cout, n.EPI_BIAS_RELU, s, bm, bn, splits, 0, 20)
    ref = F.conv2d(nchw(x), oihw(w), b, padding=1).relu()
    check(f"tall split fwd {cin}->{cout}@{hw} B{batch} {tile}/{splits}", nchw(y), ref)


@pytest.mark.parametrize("batch", [3, 32, 128])
@pytest.mark.parametrize("cin,cout,hw", GENERIC_LAYERS)
@pytest.mark.parametrize("tile", [(128, 128), (128, 64), (64, 64)])
def test_conv3x3_patch_resident(batch, cin, cout, hw, tile):
    """Patch-resident kernel: one LDS copy of the tile's pixels (plus halo) serves all nine
    taps at nine row shifts, padding handled by per-lane tap masks."""
    n = native()
    bm, bn = tile
    if cout % bn:
        pytest.skip("tile wider than the layer")
    x = rand_bf16(batch, hw, hw, cin, seed=61)
    w = rand_bf16(cout, 3, 3, cin, scale=(2.0 / (9 * cin)) ** 0.5, seed=62)
    b = torch.randn(cout, device=DEV) * 0.1
    y = torch.empty(batch, hw, hw, cout, dtype=torch.bfloat16, device=DEV)
    s = int(torch.cuda.current_stream().cuda_stream)
    n.lib().conv3x3_igemm(x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), 0, 0, 0, batch, hw,
                          cin, cout, n.EPI_BIAS_RELU, s, bm, bn, 1, 0, 16)
    ref = F.conv2d(nchw(x), oihw(w), b, padding=1).relu()
    check(f"patch fwd {cin}->{cout}@{hw} B{batch} {tile}", nchw(y), ref)


@pytest.mark.parametrize("batch", [3, 32, 256])
@pytest.mark.parametrize("cin,cout,hw", GENERIC_LAYERS + [(3, 64, 32)])
def test_conv_epilogue_statistics(batch, cin, cout, hw):
    """The BatchNorm statistics partials a conv leaves behind: per-channel sum and sum of
    squares of the bias-free output, over all tiles / finishing blocks."""
    n = native()
    C = n.lib()
    s = int(torch.cuda.current_stream().cuda_stream)
    b = torch.randn(cout, device=DEV) * 0.1
    y = torch.empty(batch, hw, hw, cout, dtype=torch.bfloat16, device=DEV)
    if cin == 3:
        x = torch.zeros(batch, 32, 32, 4, dtype=torch.bfloat16, device=DEV)
        x[..., :3] = rand_bf16(batch, 32, 32, 3, seed=51)
        w = torch.randn(64, 3, 3, 3, device=DEV).to(torch.bfloat16).float() * 0.2
        rows = C.conv1_fprop_stats_rows(batch)
        stats = torch.full((rows, 2, cout), float("nan"), device=DEV)
        C.conv1_fprop(x.data_ptr(), n.pack_conv1_weights(w.contiguous()).data_ptr(), b.data_ptr(),
                      y.data_ptr(), batch, False, s, stats.data_ptr())
        ref = F.conv2d(nchw(x[..., :3].contiguous()), w.to(torch.bfloat16).float().permute(0, 3, 1, 2),
                       None, padding=1)
    else:
        x = rand_bf16(batch, hw, hw, cin, seed=51)
        w = rand_bf16(cout, 3, 3, cin, scale=(2.0 / (9 * cin)) ** 0.5, seed=52)
        rows = C.conv3x3_igemm_stats_rows(batch, hw, cin, cout)
        stats = torch.full((rows, 2, cout), float("nan"), device=DEV)
        need = C.conv3x3_igemm_workspace_elems(batch, hw, cin, cout)
        ws = torch.empty(max(need, 1), device=DEV)
        C.conv3x3_igemm(x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), 0, ws.data_ptr(),
                        ws.numel(), batch, hw, cin, cout, n.EPI_BIAS, s, 0, 0, 0, -1, 4,
                        stats.data_ptr())
        ref = F.conv2d(nchw(x), oihw(w), None, padding=1)
    check("conv out", nchw(y), ref + b.view(1, -1, 1, 1))
    tot = stats.sum(0)
    check("sum", tot[0], ref.sum((0, 2, 3)), rtol=2e-3, atol_scale=2e-3)
    check("sum of squares", tot[1], (ref * ref).sum((0, 2, 3)), rtol=2e-3, atol_scale=2e-3)


@pytest.mark.parametrize("batch", [2, 32])
@pytest.mark.parametrize("cin,cout,hw", GENERIC_LAYERS)
def test_conv3x3_dgrad(batch, cin, cout, hw):
    n = native()
    dy = rand_bf16(batch, hw, hw, cout, seed=5)
    w = rand_bf16(cout, 3, 3, cin, scale=(2.0 / (9 * cout)) ** 0.5, seed=6)
    wd = n.repack_dgrad(w)
    # repack is exact
    ref_wd = w.float().flip(1, 2).permute(3, 1, 2, 0).contiguous()
    assert torch.equal(wd.float(), ref_wd), "repack_dgrad mismatch"
    dx = n.conv3x3_dgrad(dy, wd)
    ref = torch.nn.grad.conv2d_input((batch, cin, hw, hw), oihw(w), nchw(dy), padding=1)
    check(f"dgrad {cout}->{cin}@{hw} B{batch}", nchw(dx), ref)
    # fused ReLU backward
    act = rand_bf16(batch, hw, hw, cin, seed=7).relu()
    dxm = n.conv3x3_dgrad(dy, wd, relu_mask=act)
    check(f"dgrad+mask {cout}->{cin}@{hw} B{batch}", nchw(dxm), ref * (nchw(act) > 0))


@pytest.mark.parametrize("batch", [2, 32])
@pytest.mark.parametrize("cin,cout,hw", GENERIC_LAYERS)
def test_conv3x3_wgrad(batch, cin, cout, hw):
    n = native()
    x = rand_bf16(batch, hw, hw, cin, seed=8)
    dy = rand_bf16(batch, hw, hw, cout, seed=9)
    ref = torch.nn.grad.conv2d_weight(nchw(x), (cout, cin, 3, 3), nchw(dy), padding=1)
    for fn in (n.conv3x3_wgrad, n.conv3x3_wgrad_per_tap):
        dw = fn(x, dy)
        check(f"wgrad {cin}->{cout}@{hw} B{batch}", dw.permute(0, 3, 1, 2), ref, rtol=2e-3,
              atol_scale=2e-3)


@pytest.mark.parametrize("cin,cout,hw,batch", [(64, 128, 16, 64), (256, 256, 8, 256), (512, 512, 4, 256)])
def test_conv3x3_wgrad_slab_mode_is_deterministic(cin, cout, hw, batch):
    """Per-split partial results + the batched slab sum: same numbers as the atomic path,
    and bit-identical from run to run."""
    n = native()
    C = n.lib()
    s = int(torch.cuda.current_stream().cuda_stream)
    x = rand_bf16(batch, hw, hw, cin, seed=71)
    dy = rand_bf16(batch, hw, hw, cout, seed=72)
    splits = C.conv3x3_wgrad_splits(batch, hw, cin, cout)
    assert splits > 1
    count = cout * 9 * cin
    outs = []
    for _ in range(2):
        slab = torch.full((splits * count,), float("nan"), device=DEV)
        dw = torch.full((count + 64,), float("nan"), device=DEV)
        C.conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), batch, hw, cin, cout, s, 0,
                        slab.data_ptr())
        C.sum_slabs_batch(slab.data_ptr(), dw.data_ptr(), [0], [0], [count], [splits], s)
        assert bool(torch.isnan(dw[count:]).all())
        outs.append(dw[:count].view(cout, 3, 3, cin).clone())
    assert torch.equal(outs[0], outs[1])
    ref = torch.nn.grad.conv2d_weight(nchw(x), (cout, cin, 3, 3), nchw(dy), padding=1)
    check("wgrad slab", outs[0].permute(0, 3, 1, 2), ref, rtol=2e-3, atol_scale=2e-3)


@pytest.mark.parametrize("hw,cin,cout,batch", [
    (2, 512, 512, 256), (2, 128, 64, 5), (4, 256, 512, 256), (4, 64, 128, 3), (8, 128, 256, 256),
    (8, 256, 256, 7), (16, 64, 128, 64), (16, 128, 64, 2), (32, 64, 64, 16), (32, 64, 128, 3)])
@pytest.mark.parametrize("bco,stages", [(64, 3), (128, 3), (64, 2), (128, 2)])
def test_conv3x3_wgrad_tap_resident(hw, cin, cout, batch, bco, stages):
    """All nine taps from one dy tile and one zero-bordered x tile: every geometry (position-major
    2x2, whole images, image-row bands with halo rows), both co tiles, both ring depths, the
    kernel's own split plan and a single block per tile, against fp32 autograd."""
    n = native()
    if cout % bco:
        pytest.skip("co tile does not divide cout")
    x = rand_bf16(batch, hw, hw, cin, seed=81)
    dy = rand_bf16(batch, hw, hw, cout, seed=82)
    ref = torch.nn.grad.conv2d_weight(nchw(x), (cout, cin, 3, 3), nchw(dy), padding=1)
    steps = (batch * hw * hw + 63) // 64
    for splits in sorted({0, 1, min(3, steps)}):
        dw = n.conv3x3_wgrad9(x, dy, bco=bco, splits=splits, stages=stages)
        check(f"wgrad9 {cin}->{cout}@{hw} B{batch} bco{bco} st{stages} sp{splits}",
              dw.permute(0, 3, 1, 2), ref, rtol=2e-3, atol_scale=2e-3)


@pytest.mark.parametrize("hw,cin,cout,batch", [
    (2, 512, 512, 256), (4, 256, 512, 256), (8, 128, 256, 256), (8, 256, 256, 7), (16, 64, 128, 64),
    (16, 128, 64, 2), (32, 64, 64, 16), (4, 64, 128, 3)])
@pytest.mark.parametrize("variant", ["kstep32", "waves4"])
def test_conv3x3_wgrad_per_tap_tuning_variants(hw, cin, cout, batch, variant):
    """The per-tap kernel with 32-pixel K steps (half the LDS per block) and with four waves per
    block instead of the default eight: same numbers."""
    n = native()
    x = rand_bf16(batch, hw, hw, cin, seed=91)
    dy = rand_bf16(batch, hw, hw, cout, seed=92)
    ref = torch.nn.grad.conv2d_weight(nchw(x), (cout, cin, 3, 3), nchw(dy), padding=1)
    if variant == "kstep32":
        n.lib().set_wgrad_kstep(32)
    else:
        n.lib().set_wgrad_waves(4)
    try:
        dw = n.conv3x3_wgrad_per_tap(x, dy)
    finally:
        n.lib().set_wgrad_kstep(64)
        n.lib().set_wgrad_waves(8)
    check(f"wgrad {variant} {cin}->{cout}@{hw} B{batch}", dw.permute(0, 3, 1, 2), ref, rtol=2e-3,
          atol_scale=2e-3)


def test_conv3x3_wgrad_b256():
    n = native()
    for cin, cout, hw in [(64, 128, 16), (512, 512, 2)]:
        x = rand_bf16(256, hw, hw, cin, seed=10)
        dy = rand_bf16(256, hw, hw, cout, seed=11)
        ref = torch.nn.grad.conv2d_weight(nchw(x), (cout, cin, 3, 3), nchw(dy), padding=1)
        for fn in (n.conv3x3_wgrad, n.conv3x3_wgrad_per_tap):
            dw = fn(x, dy)
            check(f"wgrad {cin}->{cout}@{hw} B256", dw.permute(0, 3, 1, 2), ref, rtol=2e-3,
                  atol_scale=2e-3)


@pytest.mark.parametrize("batch", [2, 64, 256])
def test_conv1_fwd_and_wgrad(batch):
    n = native()
    x3 = rand_bf16(batch, 32, 32, 3, seed=12)
    x4 = torch.zeros(batch, 32, 32, 4, dtype=torch.bfloat16, device=DEV)
    x4[..., :3] = x3
    w = (torch.randn(64, 3, 3, 3, device=DEV) * 0.2).to(torch.bfloat16).float()
    b = torch.randn(64, device=DEV) * 0.1
    w1 = n.pack_conv1_weights(w)
    y = n.conv1_fwd(x4, w1, b, relu=True)
    ref = F.conv2d(nchw(x3), w.permute(0, 3, 1, 2).contiguous(), b, padding=1).relu()
    check(f"conv1 fwd B{batch}", nchw(y), ref)
    dy = rand_bf16(batch, 32, 32, 64, seed=13)
    dw, db = n.conv1_wgrad(x4, dy)
    refw = torch.nn.grad.conv2d_weight(nchw(x3), (64, 3, 3, 3), nchw(dy), padding=1)
    check(f"conv1 wgrad B{batch}", dw.permute(0, 3, 1, 2), refw, rtol=2e-3, atol_scale=2e-3)
    check(f"conv1 bias grad B{batch}", db, dy.float().sum(dim=(0, 1, 2)), rtol=2e-3,
          atol_scale=2e-3)


@pytest.mark.parametrize("c,hw", [(64, 32), (128, 16), (512, 2)])
def test_maxpool(c, hw):
    n = native()
    x = rand_bf16(8, hw, hw, c, seed=14).relu()       # post-ReLU: plenty of ties at zero
    y = n.maxpool2x2_fwd(x)
    xr = nchw(x).requires_grad_(True)
    ref = F.max_pool2d(xr, 2, 2)
    check("maxpool fwd", nchw(y), ref.detach(), rtol=0, atol_scale=0)
    dy = rand_bf16(8, hw // 2, hw // 2, c, seed=15)
    ref.backward(nchw(dy))
    dx = n.maxpool2x2_bwd(x, dy, relu_mask=False)
    check("maxpool bwd", nchw(dx), xr.grad, rtol=0, atol_scale=0)
    dxm = n.maxpool2x2_bwd(x, dy, relu_mask=True)
    check("maxpool bwd + relu mask", nchw(dxm), xr.grad * (nchw(x) > 0), rtol=0, atol_scale=0)


@pytest.mark.parametrize("c,rows_shape", [(64, (4, 32, 32)), (256, (16, 8, 8)), (512, (64, 2, 2))])
def test_bn_relu(c, rows_shape):
    n = native()
    x = rand_bf16(*rows_shape, c, scale=2.0, seed=16) + 0.5
    x = x.to(torch.bfloat16)
    gamma = torch.rand(c, device=DEV) + 0.5
    beta = torch.randn(c, device=DEV) * 0.2
    rm = torch.zeros(c, device=DEV)
    rv = torch.ones(c, device=DEV)
    xr = nchw(x).requires_grad_(True)
    g_ref = gamma.clone().requires_grad_(True)
    b_ref = beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.relu(F.batch_norm(xr, rm_ref, rv_ref, g_ref, b_ref, True, 0.1, 1e-5))
    y, mean, rstd = n.bn_relu_fwd_train(x, gamma, beta, rm, rv)
    check("bn fwd", nchw(y), ref.detach())
    check("bn running_mean", rm, rm_ref, rtol=1e-4, atol_scale=1e-4)
    check("bn running_var", rv, rv_ref, rtol=1e-3, atol_scale=1e-3)
    dy = rand_bf16(*rows_shape, c, seed=17)
    ref.backward(nchw(dy))
    dx, dgamma, dbeta = n.bn_relu_bwd(x, dy, gamma, beta, mean, rstd)
    check("bn dgamma", dgamma, g_ref.grad, rtol=5e-3, atol_scale=5e-3)
    check("bn dbeta", dbeta, b_ref.grad, rtol=5e-3, atol_scale=5e-3)
    check("bn dx", nchw(dx), xr.grad, rtol=2e-2, atol_scale=2e-2)
    ye = n.bn_relu_fwd_eval(x, gamma, beta, rm, rv)
    refe = F.relu(F.batch_norm(nchw(x), rm, rv, gamma, beta, False, 0.1, 1e-5))
    check("bn eval", nchw(ye), refe)


@pytest.mark.parametrize("c,shape", [(64, (4, 32, 32)), (256, (16, 8, 8)), (512, (64, 2, 2))])
def test_bn_relu_pool_fused(c, shape):
    """BatchNorm + ReLU + 2x2 max-pool in one pass each way against the three torch ops."""
    n = native()
    b, hw, _ = shape
    x = (rand_bf16(b, hw, hw, c, scale=2.0, seed=26) + 0.5).to(torch.bfloat16)
    gamma = torch.rand(c, device=DEV) + 0.5
    beta = torch.randn(c, device=DEV) * 0.2
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    xr = nchw(x).requires_grad_(True)
    g_ref = gamma.clone().requires_grad_(True)
    b_ref = beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.max_pool2d(F.relu(F.batch_norm(xr, rm_ref, rv_ref, g_ref, b_ref, True, 0.1, 1e-5)), 2)
    y, mean, rstd = n.bn_relu_pool_fwd_train(x, gamma, beta, rm, rv)
    check("bn+pool fwd", nchw(y), ref.detach())
    check("bn+pool running_mean", rm, rm_ref, rtol=1e-4, atol_scale=1e-4)
    check("bn+pool running_var", rv, rv_ref, rtol=1e-3, atol_scale=1e-3)
    dy = rand_bf16(b, hw // 2, hw // 2, c, seed=27)
    ref.backward(nchw(dy))
    dx, dgamma, dbeta = n.bn_relu_pool_bwd(x, dy, gamma, beta, mean, rstd)
    check("bn+pool dgamma", dgamma, g_ref.grad, rtol=5e-3, atol_scale=5e-3)
    check("bn+pool dbeta", dbeta, b_ref.grad, rtol=5e-3, atol_scale=5e-3)
    check("bn+pool dx", nchw(dx), xr.grad, rtol=2e-2, atol_scale=2e-2)
    ye = n.bn_relu_pool_fwd_eval(x, gamma, beta, rm, rv)
    refe = F.max_pool2d(F.relu(F.batch_norm(nchw(x), rm, rv, gamma, beta, False, 0.1, 1e-5)), 2)
    check("bn+pool eval", nchw(ye), refe)


@pytest.mark.parametrize("pooled", [True, False])
@pytest.mark.parametrize("batch,hw,cin,cout", [(64, 16, 128, 64), (256, 8, 256, 128),
                                               (256, 8, 256, 256), (256, 4, 512, 512)])
def test_dgrad_epilogue_leaves_bn_pool_backward_sums(batch, hw, cin, cout, pooled):
    """A backward-data convolution whose output is the gradient of a BatchNorm + ReLU (+ pool)
    layer's (pooled) activation: the dbeta / dgamma partial rows of its epilogue must fold to what
    the stand-alone reduction pass computes from the stored gradient (same per-element terms,
    another summation order), and the gradient itself must be unchanged."""
    n = native()
    C = n.lib()
    st = int(torch.cuda.current_stream().cuda_stream)
    # consumer layer: `cout` channels at 2*hw, pooled to hw (un-pooled: at hw); this conv maps
    # cin -> cout at hw
    side = 2 * hw if pooled else hw
    pre = (rand_bf16(batch, side, side, cout, scale=2.0, seed=71) + 0.3).to(torch.bfloat16)
    gamma = torch.rand(cout, device=DEV) + 0.5
    beta = torch.randn(cout, device=DEV) * 0.2
    rm, rv = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
    if pooled:
        _, mean, rstd = n.bn_relu_pool_fwd_train(pre, gamma, beta, rm, rv)
    else:
        _, mean, rstd = n.bn_relu_fwd_train(pre, gamma, beta, rm, rv)
    x = rand_bf16(batch, hw, hw, cin, seed=72)
    w = (torch.randn(cout, 3, 3, cin, device=DEV) * 0.05).to(torch.bfloat16)
    ws_elems = C.conv3x3_igemm_workspace_elems(batch, hw, cin, cout)
    ws = torch.empty(max(ws_elems, 1), device=DEV)
    y_ref = torch.empty(batch, hw, hw, cout, dtype=torch.bfloat16, device=DEV)
    C.conv3x3_igemm(x.data_ptr(), w.data_ptr(), y_ref.data_ptr(), 0, 0, ws.data_ptr(), ws_elems,
                    batch, hw, cin, cout, 0, st)
    y = torch.empty_like(y_ref)
    partials = torch.zeros(batch * hw * hw // 64 * 2 * cout, device=DEV)
    rows = C.conv3x3_igemm_bn_pool(x.data_ptr(), w.data_ptr(), y.data_ptr(), ws.data_ptr(), ws_elems,
                                   batch, hw, cin, cout, st, pre.data_ptr(), gamma.data_ptr(),
                                   beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                   partials.data_ptr(), pooled)
    if rows == 0:
        pytest.skip("this shape takes a split / position-major plan: no sums in the epilogue")
    assert torch.equal(y, y_ref)
    sums = partials[:rows * 2 * cout].view(rows, 2, cout).double().sum(0)
    if pooled:
        _, dgamma, dbeta = n.bn_relu_pool_bwd(pre, y_ref, gamma, beta, mean, rstd)
    else:
        _, dgamma, dbeta = n.bn_relu_bwd(pre, y_ref, gamma, beta, mean, rstd)
    check("dbeta from the dgrad epilogue", sums[0].float(), dbeta, rtol=1e-4, atol_scale=1e-4)
    check("dgamma from the dgrad epilogue", sums[1].float(), dgamma, rtol=1e-4, atol_scale=1e-4)


@pytest.mark.parametrize("pooled", [False, True])
def test_bn_backward_sums_split_slabs_itself(pooled):
    """BatchNorm backward fed the unsummed split-K slabs of the convolution that produced its
    incoming gradient: bit-identical to summing the slabs first, and the summed bf16 gradient is
    left where the apply pass (and the caller) expects it."""
    n = native()
    C = n.lib()
    b, hw, c, splits = 16, 4, 128, 3      # (narrow enough that both runs take the three-kernel path)
    x = (rand_bf16(b, hw, hw, c, scale=2.0, seed=61) + 0.3).to(torch.bfloat16)
    gamma = torch.rand(c, device=DEV) + 0.5
    beta = torch.randn(c, device=DEV) * 0.2
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    ohw = hw // 2 if pooled else hw
    slabs = torch.randn(splits, b, ohw, ohw, c, device=DEV)
    dy_ref = ((slabs[0] + slabs[1]) + slabs[2]).to(torch.bfloat16)
    if pooled:
        _, mean, rstd = n.bn_relu_pool_fwd_train(x, gamma, beta, rm, rv)
        want = n.bn_relu_pool_bwd(x, dy_ref, gamma, beta, mean, rstd)
    else:
        _, mean, rstd = n.bn_relu_fwd_train(x, gamma, beta, rm, rv)
        want = n.bn_relu_bwd(x, dy_ref, gamma, beta, mean, rstd)
    dy = torch.zeros_like(dy_ref)
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, device=DEV)
    dbeta = torch.empty(c, device=DEV)
    ws = torch.empty(C.column_reduce_workspace_elems(), device=DEV)
    st = int(torch.cuda.current_stream().cuda_stream)
    args = (x.data_ptr(), dy.data_ptr(), dx.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
            mean.data_ptr(), rstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr())
    if pooled:
        C.bn_relu_pool_bwd(*args, b, hw, c, st, slabs.data_ptr(), splits, slabs[0].numel())
    else:
        C.bn_relu_bwd(*args, b * hw * hw, c, st, slabs.data_ptr(), splits, slabs[0].numel())
    assert torch.equal(dy, dy_ref)
    assert torch.equal(dx, want[0]) and torch.equal(dgamma, want[1]) and torch.equal(dbeta, want[2])


@pytest.mark.parametrize("batch", [2, 64])
def test_conv1_wgrad_with_bn_pool_backward_on_the_fly(batch):
    """First layer under BatchNorm + ReLU + max-pool: the weight-gradient kernel that forms
    dL/d(conv output) on the fly must give what the stored-gradient path gives (same formula,
    same bf16 rounding of the gradient), and match torch autograd through conv -> BN -> ReLU
    -> pool."""
    n = native()
    torch.manual_seed(5)
    x = rand_bf16(batch, 32, 32, 3, seed=41)
    x4 = torch.zeros(batch, 32, 32, 4, dtype=torch.bfloat16, device=DEV)
    x4[..., :3] = x
    w = (torch.randn(64, 3, 3, 3, device=DEV) * 0.2)        # [cout][kh][kw][cin]
    bias = torch.randn(64, device=DEV) * 0.1
    gamma = torch.rand(64, device=DEV) + 0.5
    beta = torch.randn(64, device=DEV) * 0.2
    pre = n.conv1_fwd(x4, n.pack_conv1_weights(w), bias, relu=False)
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    y, mean, rstd = n.bn_relu_pool_fwd_train(pre, gamma, beta, rm, rv)
    dy = rand_bf16(batch, 16, 16, 64, seed=42)
    g, dgamma_a, dbeta_a = n.bn_relu_pool_bwd(pre, dy, gamma, beta, mean, rstd)
    dw_a, _ = n.conv1_wgrad(x4, g)
    dw_b, dgamma_b, dbeta_b = n.conv1_wgrad_bn_pool(x4, pre, dy, gamma, beta, mean, rstd)
    # (the same partial sums folded in two orders: inside the apply pass / by the fold kernel)
    torch.testing.assert_close(dgamma_a, dgamma_b, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(dbeta_a, dbeta_b, rtol=1e-5, atol=1e-4)
    check("conv1 wgrad fused vs stored gradient", dw_b, dw_a, rtol=1e-4, atol_scale=1e-4)
    # against autograd on the bf16-rounded operands
    wr = w.to(torch.bfloat16).float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    conv = F.conv2d(nchw(x).float(), wr, bias, padding=1)
    # forward values = the bf16 conv output the kernels saw (pooling winners and ReLU signs
    # flip on rounding), gradient path = the fp32 convolution
    conv = conv + (nchw(pre).float() - conv).detach()
    ref = F.max_pool2d(F.relu(F.batch_norm(conv, None, None, gamma, beta, True, 0.1, 1e-5)), 2)
    ref.backward(nchw(dy).float())
    check("conv1 wgrad fused vs torch", dw_b, wr.grad.permute(0, 2, 3, 1), rtol=3e-2, atol_scale=3e-2)


@pytest.mark.parametrize("batch", [2, 64])
def test_conv1_wgrad_with_relu_pool_backward_on_the_fly(batch):
    """First layer followed by ReLU + max-pool: forming the routed gradient inside the
    weight-gradient kernel equals storing it with maxpool2x2_bwd first (exactly: the routed
    values are copies of the incoming gradient)."""
    n = native()
    x = rand_bf16(batch, 32, 32, 3, seed=43)
    x4 = torch.zeros(batch, 32, 32, 4, dtype=torch.bfloat16, device=DEV)
    x4[..., :3] = x
    w = torch.randn(64, 3, 3, 3, device=DEV) * 0.2
    bias = torch.randn(64, device=DEV) * 0.1
    act = n.conv1_fwd(x4, n.pack_conv1_weights(w), bias, relu=True)
    dy = rand_bf16(batch, 16, 16, 64, seed=44)
    g = n.maxpool2x2_bwd(act, dy, relu_mask=True)
    dw_a, db_a = n.conv1_wgrad(x4, g)
    dw_b, db_b = n.conv1_wgrad_relu_pool(x4, act, dy)
    assert torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b)


@pytest.mark.parametrize("batch", [3, 256])
def test_linear_and_softmax_xent(batch):
    n = native()
    x = rand_bf16(batch, 512, seed=18)
    w = torch.randn(10, 512, device=DEV) * 0.05
    b = torch.randn(10, device=DEV) * 0.1
    t = torch.randint(0, 10, (batch,), device=DEV)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    xr = x.float().requires_grad_(True)
    logits_ref = F.linear(xr, wr, br)
    loss_ref = F.cross_entropy(logits_ref, t)
    loss_ref.backward()
    logits = n.linear_fwd(x, w, b)
    check("linear fwd", logits, logits_ref.detach(), rtol=1e-4, atol_scale=1e-4)
    loss, dl, correct = n.softmax_xent(logits, t)
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * max(1.0, abs(float(loss_ref)))
    assert int(correct) == int((logits_ref.argmax(1) == t).sum())
    dx, dw, db = n.linear_bwd(x, w, dl)
    check("linear dw", dw, wr.grad, rtol=1e-3, atol_scale=1e-3)
    check("linear db", db, br.grad, rtol=1e-3, atol_scale=1e-3)
    check("linear dx", dx, xr.grad)


def test_fused_sgd_matches_torch():
    n = native()
    torch.manual_seed(0)
    size = 4 * 25013
    p = torch.randn(size, device=DEV)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([ref], lr=0.1, momentum=0.9, weight_decay=1e-4)
    m = torch.zeros_like(p)
    shadow = torch.empty(size, dtype=torch.bfloat16, device=DEV)
    for step in range(4):
        g = torch.randn(size, device=DEV)
        ref.grad = g.clone()
        opt.step()
        n.sgd_momentum_(p, g, m, 0.1, 0.9, 1e-4, shadow=shadow)
        assert torch.allclose(p, ref.detach(), rtol=1e-6, atol=1e-6), f"step {step}"
    assert torch.equal(shadow, p.to(torch.bfloat16))
    # grad_scale folds the 1/world_size of a mean reduction into the update
    p2, m2 = p.clone(), torch.zeros_like(p)
    g = torch.randn(size, device=DEV)
    n.sgd_momentum_(p2, g * 4, m2, 0.1, 0.9, 0.0, grad_scale=0.25)
    assert torch.allclose(p2, p - 0.1 * g, rtol=1e-6, atol=1e-6)


def test_column_sum_and_slab_sum():
    n = native()
    x = rand_bf16(37, 8, 8, 128, seed=19)
    check("column_sum", n.column_sum(x), x.float().sum(dim=(0, 1, 2)), rtol=1e-4, atol_scale=1e-4)
    slabs = torch.randn(5, 4 * 1001, device=DEV)
    ref = slabs.sum(0)
    n.sum_slabs_(slabs)
    assert torch.allclose(slabs[0], ref, rtol=1e-6, atol=1e-5)


def test_copy_batch_copies_every_job_and_nothing_else():
    """One launch copies many tensors (the gather / scatter root's own 18 gradient tensors):
    VGG-11's tensor sizes, plus odd sizes and misaligned views for the narrow paths."""
    n = native()
    sizes = [1728, 64, 73728, 128, 294912, 256, 589824, 256, 1179648, 512, 2359296, 512,
             5120, 10, 7, 1, 33]
    g = torch.Generator(device="cpu").manual_seed(5)
    pairs, checks = [], []
    for i, sz in enumerate(sizes):
        src = torch.randn(sz + 2, generator=g).to(DEV)
        dst = torch.full((sz + 4,), -7.0, device=DEV)
        off = 1 if i % 3 == 2 else 0                      # 4-byte aligned only
        pairs.append((dst[1 + off:1 + off + sz], src[off:off + sz]))
        checks.append((dst, src, off, sz))
    raw = torch.arange(0, 251, dtype=torch.uint8, device=DEV)      # byte path, odd length
    raw_dst = torch.zeros(260, dtype=torch.uint8, device=DEV)
    pairs.append((raw_dst[3:3 + 251], raw))
    n.copy_batch_(pairs)
    torch.cuda.synchronize()
    for dst, src, off, sz in checks:
        assert torch.equal(dst[1 + off:1 + off + sz], src[off:off + sz])
        assert float(dst[off]) == -7.0 and float(dst[1 + off + sz]) == -7.0    # borders intact
    assert torch.equal(raw_dst[3:254], raw) and int(raw_dst[:3].sum()) == 0 and int(raw_dst[254:].sum()) == 0


def test_augment_kernel_matches_reference_transform():
    from distributed_machine_learning_amd.data import synthetic_cifar
    from distributed_machine_learning_amd.data.transforms import (_augment_reference,
                                                                   draw_augmentation)
    n = native()
    ds = synthetic_cifar(16, 3).to(DEV)
    aug = draw_augmentation(torch.arange(16, device=DEV), 7)
    out = n.augment_normalize(ds.images, aug)
    ref = _augment_reference(ds.images, aug).permute(0, 2, 3, 1)
    assert torch.allclose(out[..., :3].float(), ref, rtol=1e-2, atol=1e-2)
    assert float(out[..., 3].abs().max()) == 0.0
    plain = n.augment_normalize(ds.images, None)
    assert torch.allclose(plain[..., :3].float(),
                          _augment_reference(ds.images, None).permute(0, 2, 3, 1),
                          rtol=1e-2, atol=1e-2)
    x = torch.randn(4, 3, 32, 32, device=DEV)
    assert torch.equal(n.nchw_to_nhwc4(x)[..., :3], x.permute(0, 2, 3, 1).to(torch.bfloat16))


@pytest.mark.parametrize("batch", [3, 64, 256])
def test_fc_train_fused(batch):
    """Classifier head: logits, loss, top-1 count and all three gradients in two launches."""
    n = native()
    x = rand_bf16(batch, 512, seed=81)
    w = (torch.randn(10, 512, device=DEV) * 0.05)
    b = torch.randn(10, device=DEV) * 0.1
    t = torch.randint(0, 10, (batch,), device=DEV)
    logits, loss, correct, dx, dw, db = n.fc_train(x, w, b, t)
    xr = x.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    ref_logits = F.linear(xr, wr, br)
    ref_loss = F.cross_entropy(ref_logits, t)
    ref_loss.backward()
    check("fc logits", logits, ref_logits.detach(), rtol=1e-4, atol_scale=1e-4)
    assert abs(float(loss) - float(ref_loss)) < 1e-4
    assert int(correct) == int((ref_logits.argmax(1) == t).sum())
    check("fc dx", dx, xr.grad)
    check("fc dw", dw, wr.grad, rtol=1e-3, atol_scale=1e-3)
    check("fc db", db, br.grad, rtol=1e-3, atol_scale=1e-3)
    again = n.fc_train(x, w, b, t)
    assert torch.equal(again[4], dw) and torch.equal(again[1], loss)     # bit-reproducible
