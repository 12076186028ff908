"""Per-tap weight-gradient kernel, four against eight waves per block, at the split counts the
bf16 engine uses under BatchNorm (block target 288) and their neighbours; batch 256, stand-alone."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from distributed_machine_learning_amd.ops import native  # noqa: E402

LAYERS = [(64, 128, 16), (128, 256, 8), (256, 256, 8), (256, 512, 4), (512, 512, 4), (512, 512, 2)]
B = 256


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    C = native.lib()
    s = int(torch.cuda.current_stream().cuda_stream)
    C.set_wgrad_variant(1)
    for cin, cout, hw in LAYERS:
        x = native.fill_uniform_(torch.empty(B, hw, hw, cin, dtype=torch.bfloat16, device="cuda"), 1)
        dy = native.fill_uniform_(torch.empty(B, hw, hw, cout, dtype=torch.bfloat16, device="cuda"), 2)
        dw = torch.zeros(cout, 3, 3, cin, device="cuda")
        slab = torch.empty(max(64, C.conv3x3_wgrad_max_splits(B, hw, cin, cout)) * cout * 9 * cin,
                           device="cuda")
        eng = C.conv3x3_wgrad_splits(B, hw, cin, cout, 288)
        for sp in sorted({max(1, eng // 2), eng, eng * 2}):
            row = []
            for nw in (4, 8):
                C.set_wgrad_waves(nw)
                row.append(timeit(lambda: C.conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(),
                                                          B, hw, cin, cout, s, sp, slab.data_ptr())))
            C.set_wgrad_waves(8)
            mark = " <- engine" if sp == eng else ""
            print(f"wgrad {cin:4d}->{cout:4d}@{hw:2d} splits={sp:2d}  4 waves {row[0]:7.1f} us   "
                  f"8 waves {row[1]:7.1f} us{mark}", flush=True)


if __name__ == "__main__":
    main()
