"""the deep VGG layers at the row counts of a one- / two-view step and of BASELINE configs[1] (100^3, one view): forward and
data gradient per layer, warm and behind a cache-polluting fill (what a layer sees inside the step: its filters come from
HBM).  The measured floor of the three-kernel path at these row counts (round 5: a GEMM with every operand in flight at
once ran at exactly these times -- the launches stream their transformed filters at the HBM rate already)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neural_flow_style_amd.ops as ops


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cold = torch.empty(192 << 20, dtype=torch.float32, device="cuda")          # 768 MB: evicts L2 + Infinity Cache between calls
t_fill = timed(lambda: cold.zero_())
for B, H, Ci, Co, name in [(1, 50, 256, 256, "conv3_2 @200^2 x1 (100 tiles: not few-row)"),
                           (1, 25, 256, 512, "conv4_1 @200^2 x1"), (1, 25, 512, 512, "conv4_2 @200^2 x1"),
                           (2, 25, 512, 512, "conv4_2 @200^2 x2"), (1, 12, 512, 512, "conv5_1 @200^2 x1"),
                           (1, 25, 256, 256, "conv3_2 @100^2 x1"), (1, 12, 512, 512, "conv4_2 @100^2 x1"),
                           (1, 6, 512, 512, "conv5_1 @100^2 x1")]:
    x = torch.relu(torch.randn(B, H, H, Ci, device="cuda")); w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
    b = torch.zeros(Co, device="cuda"); wf = ops.conv3x3_pack(w, 0); wd = ops.conv3x3_pack(w, 1)
    out = torch.empty(B, H, H, Co, device="cuda"); gy = torch.randn(B, H, H, Co, device="cuda")
    tf = timed(lambda: ops.conv3x3_fwd(x, wf, b, Co, True, out=out))
    tb = timed(lambda: ops.conv3x3_dgrad(gy, wd, Ci))
    tfc = timed(lambda: (cold.zero_(), ops.conv3x3_fwd(x, wf, b, Co, True, out=out))) - t_fill
    tbc = timed(lambda: (cold.zero_(), ops.conv3x3_dgrad(gy, wd, Ci))) - t_fill
    print("%-44s  warm: fwd %6.1f us dgrad %6.1f us   cold: fwd %6.1f us dgrad %6.1f us" % (name, tf, tb, tfc, tbc), flush=True)
