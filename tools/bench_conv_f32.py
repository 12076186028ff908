"""Per-layer timing of the fp32 conv kernels at batch 256 for every tile / row order / split
choice (forward and backward-data through conv3x3_igemm_f32, backward-weight through
conv3x3_wgrad_f32).  Used to pick the launch rules in csrc/kernels/conv_f32.hip."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from distributed_machine_learning_amd.ops import native  # noqa: E402

LAYERS = [(64, 128, 16), (128, 256, 8), (256, 256, 8), (256, 512, 4), (512, 512, 4), (512, 512, 2)]
B = int(os.environ.get("BATCH", "256"))


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    which = sys.argv[1:] or ["fwd", "dgrad", "wgrad"]
    for cin, cout, hw in LAYERS:
        flops = 2.0 * B * hw * hw * cin * cout * 9
        x = torch.randn(B, hw, hw, cin, device="cuda")
        dy = torch.randn(B, hw, hw, cout, device="cuda")
        w = torch.randn(cout, 3, 3, cin, device="cuda") * 0.05
        bias = torch.zeros(cout, device="cuda")
        wd = native.repack_dgrad_f32(w)
        for kind in which:
            if kind == "wgrad":
                for pm in (1,):
                    for splits in (0, 4, 8, 12, 16, 24, 32, 48, 64):
                        try:
                            t = timeit(lambda: native.conv3x3_wgrad_f32(x, dy, splits=splits, pm=pm))
                        except Exception as e:      # noqa: BLE001
                            print(f"wgrad {cin}->{cout}@{hw} pm={pm} splits={splits}: {e}")
                            continue
                        print(f"wgrad {cin:4d}->{cout:4d}@{hw:2d} pm={pm} splits={splits:2d} {t:8.1f} us "
                              f"{flops / t / 1e6:6.1f} TF/s (incl. the slab sum)", flush=True)
                continue
            for tile in (64, 128):
                for pm in (0, 1):
                    for split in (0, 1):
                        n_out = cout if kind == "fwd" else cin
                        if (tile == 128 and n_out % 128) or (split and not pm) or (pm and hw < 2):
                            continue
                        if kind == "fwd":
                            fn = lambda: native.conv3x3_fwd_f32(x, w, bias, relu=False, tile=tile,
                                                                pm=pm, split=split)
                        else:
                            fn = lambda: native.conv3x3_dgrad_f32(dy, wd, tile=tile, pm=pm, split=split)
                        t = timeit(fn)
                        print(f"{kind:5s} {cin:4d}->{cout:4d}@{hw:2d} tile={tile:3d} pm={pm} split={split} "
                              f"{t:8.1f} us {flops / t / 1e6:6.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
