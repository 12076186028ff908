"""Timing of the BatchNorm+ReLU kernels per VGG-11 layer shape (batch 256)."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from distributed_machine_learning_amd.ops import native  # noqa: E402

DEV = "cuda"
SHAPES = [(32, 64), (16, 128), (8, 256), (4, 512), (2, 512)]
B = 256


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for hw, c in SHAPES:
    x = native.fill_uniform_(torch.empty(B, hw, hw, c, dtype=torch.bfloat16, device=DEV), 1)
    dy = native.fill_uniform_(torch.empty(B, hw, hw, c, dtype=torch.bfloat16, device=DEV), 2)
    g = torch.ones(c, device=DEV)
    b = torch.zeros(c, device=DEV)
    rm = torch.zeros(c, device=DEV)
    rv = torch.ones(c, device=DEV)
    mb = x.numel() * 2 / 1e6
    y, mean, rstd = native.bn_relu_fwd_train(x, g, b, rm, rv)
    t_train = timeit(lambda: native.bn_relu_fwd_train(x, g, b, rm, rv))
    t_eval = timeit(lambda: native.bn_relu_fwd_eval(x, g, b, rm, rv))
    t_bwd = timeit(lambda: native.bn_relu_bwd(x, dy, g, b, mean, rstd))
    t_cs = timeit(lambda: native.column_sum(x))
    t_pool = timeit(lambda: native.maxpool2x2_fwd(x))
    t_poolb = timeit(lambda: native.maxpool2x2_bwd(x, dy[:, :hw // 2, :hw // 2].contiguous(), True)) \
        if hw > 1 else 0
    print(f"hw={hw:2d} c={c:3d} {mb:6.1f} MB | fwd_train {t_train:6.1f} us  eval(apply only) "
          f"{t_eval:6.1f} us  bwd {t_bwd:6.1f} us  colsum {t_cs:6.1f} us  pool {t_pool:5.1f} "
          f"poolbwd {t_poolb:5.1f}", flush=True)
