"""Workload for rocprofv3 --pmc: every fp32 conv kernel of the fp32 step once per layer
(default plans, batch 256), three repetitions so cold-cache effects can be dropped.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY \
        --kernel-trace -d out -o pmc -- python3 tools/pmc_conv_f32.py
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
            x = torch.rand(B, hw, hw, cin, device="cuda") - 0.5
            dy = torch.rand(B, hw, hw, cout, device="cuda") - 0.5
            w = (torch.rand(cout, 3, 3, cin, device="cuda") - 0.5) * 0.1
            bias = torch.zeros(cout, device="cuda")
            native.conv3x3_fwd_f32(x, w, bias, relu=False)
            native.conv3x3_dgrad_f32(dy, native.repack_dgrad_f32(w))
            native.conv3x3_wgrad_f32(x, dy)
    x4 = torch.rand(B, 32, 32, 4, device="cuda")
    native.conv1_fwd_f32(x4, torch.randn(64, 3, 3, 3, device="cuda"), torch.zeros(64, device="cuda"))
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
