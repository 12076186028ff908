"""Workload for rocprofv3 --pmc: every conv kernel of the flagship step once per layer
(default plans, batch 256), three repetitions so cold-cache effects can be dropped.

    rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES ... -d out -o pmc -- python3 tools/pmc_conv.py
    python tools/rocpd_pmc.py out/pmc_results.db
"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from distributed_machine_learning_amd.ops import native  # noqa: E402

LAYERS = [(64, 128, 16), (128, 256, 8), (256, 256, 8), (256, 512, 4), (512, 512, 4), (512, 512, 2)]
B = int(os.environ.get("BATCH", "256"))


def main():
    for rep in range(3):
        for cin, cout, hw in LAYERS:
            x = native.fill_uniform_(torch.empty(B, hw, hw, cin, dtype=torch.bfloat16, device="cuda"), 1)
            dy = native.fill_uniform_(torch.empty(B, hw, hw, cout, dtype=torch.bfloat16, device="cuda"), 2)
            w = native.fill_uniform_(torch.empty(cout, 3, 3, cin, dtype=torch.bfloat16, device="cuda"), 3, 0.05)
            bias = torch.zeros(cout, device="cuda")
            native.conv3x3_fwd(x, w, bias, relu=True)
            native.conv3x3_dgrad(dy, native.repack_dgrad(w))
            native.conv3x3_wgrad_per_tap(x, dy)       # one tap per block
            native.conv3x3_wgrad9(x, dy, bco=64, stages=2)        # tap-resident, both co tiles
            if cout % 128 == 0:
                native.conv3x3_wgrad9(x, dy, bco=128, stages=3)
    x4 = native.fill_uniform_(torch.empty(B, 32, 32, 4, dtype=torch.bfloat16, device="cuda"), 4)
    w1 = native.pack_conv1_weights(torch.randn(64, 3, 3, 3, device="cuda"))
    native.conv1_fwd(x4, w1, torch.zeros(64, device="cuda"))
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
