"""one deep layer's forward convolution in a loop, split-limb GEMM mode (for rocprofv3 counter passes over the rb16s
kernel): python tools/split_gemm_one.py [HW=25] [Ci=512] [Co=512] [views=8] [mode=1] [reps=30]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops
a = [int(v) for v in sys.argv[1:]] + [None] * 6
HW, Ci, Co, B, mode, reps = a[0] or 25, a[1] or 512, a[2] or 512, a[3] or 8, 1 if a[4] is None else a[4], a[5] or 30
x = torch.relu(torch.randn(B, HW, HW, Ci, device="cuda")); w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.03
b = torch.zeros(Co, device="cuda"); wf = ops.conv3x3_pack(w, 0)
ops.gemm_mode(mode)
for _ in range(reps):
    ops.conv3x3_fwd(x, wf, b, Co, True)
torch.cuda.synchronize()
