"""One conv kernel, three launches, for a rocprofv3 --pmc pass.

    LAYER=2 KIND=fwd PLAN=128,128,1,0,8 rocprofv3 --pmc ... -- python3 tools/pmc_one.py
LAYER indexes (cin, cout, hw) of VGG-11 conv2..conv8; PLAN = BM,BN,splits,pm,waves (0 = auto).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from distributed_machine_learning_amd.ops import native  # noqa: E402

LAYERS = [(64, 128, 16), (128, 256, 8), (256, 256, 8), (256, 512, 4), (512, 512, 4), (512, 512, 2)]
B = int(os.environ.get("BATCH", "256"))
cin, cout, hw = LAYERS[int(os.environ.get("LAYER", "2"))]
kind = os.environ.get("KIND", "fwd")
plan = [int(v) for v in os.environ.get("PLAN", "0,0,0,-1,4").split(",")]
C = native.lib()
s = int(torch.cuda.current_stream().cuda_stream)
x = native.fill_uniform_(torch.empty(B, hw, hw, cin, dtype=torch.bfloat16, device="cuda"), 1)
dy = native.fill_uniform_(torch.empty(B, hw, hw, cout, dtype=torch.bfloat16, device="cuda"), 2)
w = native.fill_uniform_(torch.empty(cout, 3, 3, cin, dtype=torch.bfloat16, device="cuda"), 3, 0.05)
bias = torch.zeros(cout, device="cuda")
y = torch.empty(B, hw, hw, cout, dtype=torch.bfloat16, device="cuda")
ws = torch.empty(9 * B * hw * hw * max(cin, cout), device="cuda")
dw = torch.zeros(cout, 3, 3, cin, device="cuda")
for _ in range(3):
    if kind == "fwd":
        C.conv3x3_igemm(x.data_ptr(), w.data_ptr(), y.data_ptr(), bias.data_ptr(), 0, ws.data_ptr(),
                        ws.numel(), B, hw, cin, cout, 2, s, *plan)
    else:
        C.conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, hw, cin, cout, s, plan[2])
torch.cuda.synchronize()
