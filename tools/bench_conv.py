"""Per-layer timing of the conv kernels at the flagship batch (256) for every tile / split
plan, on random bf16 data.  Used to pick the plan table in csrc/kernels/conv_igemm.hip and
the split heuristic of conv_wgrad.hip.  Writes a table to stdout."""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from distributed_machine_learning_amd.ops import native  # noqa: E402

LAYERS = [(64, 128, 16), (128, 256, 8), (256, 256, 8), (256, 512, 4), (512, 512, 4),
          (512, 512, 2)]
B = int(os.environ.get("BATCH", "256"))
DEV = "cuda"


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
    return e0.elapsed_time(e1) / iters * 1e3      # us


def main():
    C = native.lib()
    s = int(torch.cuda.current_stream().cuda_stream)
    which = sys.argv[1:] or ["fwd", "dgrad", "wgrad"]
    for cin, cout, hw in LAYERS:
        M = B * hw * hw
        flops = 2.0 * M * cout * cin * 9
        x = native.fill_uniform_(torch.empty(B, hw, hw, cin, dtype=torch.bfloat16, device=DEV), 1)
        dy = native.fill_uniform_(torch.empty(B, hw, hw, cout, dtype=torch.bfloat16, device=DEV), 2)
        w = native.fill_uniform_(torch.empty(cout, 3, 3, cin, dtype=torch.bfloat16, device=DEV), 3,
                                 0.05)
        wd = native.repack_dgrad(w)
        bias = torch.zeros(max(cin, cout), device=DEV)
        y = torch.empty(B, hw, hw, cout, dtype=torch.bfloat16, device=DEV)
        dx = torch.empty(B, hw, hw, cin, dtype=torch.bfloat16, device=DEV)
        dw = torch.zeros(cout, 3, 3, cin, device=DEV)
        ws = torch.empty(9 * M * max(cin, cout), device=DEV)
        plans = [(128, 128, 1), (128, 64, 1), (64, 64, 1), (128, 128, 3), (128, 64, 3), (64, 64, 3),
                 (64, 64, 9), (0, 0, 0), (256, 64, 1)]
        waves_opts = (4, 8, 9, 10, 16, 20, 21)
        if os.environ.get("TALL_ONLY"):
            plans = [(256, 64, 1), (128, 64, 1), (0, 0, 0)]
            waves_opts = (4, 20, 21, 22, 23, 24, 25)
        for kind in which:
            if kind == "wgrad":
                C.set_wgrad_variant(1)      # the one-tap-per-block kernel
                slab = torch.empty(max(64, C.conv3x3_wgrad_max_splits(B, hw, cin, cout)) *
                                   cout * 9 * cin, device=DEV)
                for sp in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64):
                    if sp > (M + 63) // 64:
                        continue
                    t = timeit(lambda: C.conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(),
                                                       B, hw, cin, cout, s, sp, slab.data_ptr()))
                    print(f"wgrad {cin:4d}->{cout:4d}@{hw:2d} splits={sp:2d} {t:8.1f} us "
                          f"{flops / t / 1e6:7.1f} TF/s (slab mode, sum kernel not included)",
                          flush=True)
                del slab
                C.set_wgrad_variant(0)
                # tap-resident kernel: co tile x ring depth x block target (slab sum not included)
                for bco in (64, 128):
                    if cout % bco:
                        continue
                    for stages in (2, 3):
                        for target in (64, 128, 256, 512):
                            n = C.conv3x3_wgrad9_splits(B, hw, cin, cout, bco, 0, target)
                            slab = torch.empty(max(n, 1) * cout * 9 * cin, device=DEV)
                            t = timeit(lambda: C.conv3x3_wgrad9(x.data_ptr(), dy.data_ptr(),
                                                                dw.data_ptr(), B, hw, cin, cout, s,
                                                                slab.data_ptr(), bco, n, stages))
                            print(f"wgrad9 {cin:4d}->{cout:4d}@{hw:2d} bco={bco:3d} stages={stages} "
                                  f"target={target:3d} splits={n:3d} {t:8.1f} us "
                                  f"{flops / t / 1e6:7.1f} TF/s", flush=True)
                            del slab
                continue
            for (bm, bn, sp), pm, nw in itertools.product(plans, (0, 1), waves_opts):
                if nw in (8, 9, 10) and (bm, bn) != (128, 128):
                    continue
                if nw == 16 and (sp != 1 or pm or bm == 0 or bm == 256):
                    continue
                if nw >= 20 and (sp != 1 or pm or bn != 64 or hw > 16 or (bm == 256 and nw == 25)):
                    continue
                if bm == 256 and nw < 20:
                    continue
                if os.environ.get("TALL_ONLY") and nw == 4 and bm:
                    continue
                n_out = cout if kind == "fwd" else cin
                if bn and n_out % bn:
                    continue
                if pm and (hw > 8 or (bm and B % bm)):
                    continue
                if bm == 0:
                    pm = -1
                if kind == "fwd":
                    fn = lambda: C.conv3x3_igemm(x.data_ptr(), w.data_ptr(), y.data_ptr(),
                                                 bias.data_ptr(), 0, ws.data_ptr(), ws.numel(), B,
                                                 hw, cin, cout, 2, s, bm, bn, sp, pm, nw)
                else:
                    fn = lambda: C.conv3x3_igemm(dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), 0, 0,
                                                 ws.data_ptr(), ws.numel(), B, hw, cout, cin, 0, s,
                                                 bm, bn, sp, pm, nw)
                t = timeit(fn)
                print(f"{kind:5s} {cin:4d}->{cout:4d}@{hw:2d} plan={bm:3d}x{bn:3d}/{sp} pm={pm:2d} w={nw} {t:8.1f} us "
                      f"{flops / t / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
