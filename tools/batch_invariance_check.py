import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import neural_flow_style_amd.ops as ops
rng = np.random.RandomState(0)
for mode in (0, 1):
    ops.gemm_mode(mode)
    for (H, Ci, Co) in [(14, 128, 256), (14, 256, 256), (7, 256, 512), (7, 512, 512), (3, 512, 512), (28, 64, 128), (25, 512, 512), (12, 512, 512)]:
        x = torch.tensor(np.maximum(rng.randn(6, H, H, Ci), 0).astype(np.float32)).cuda()
        w = torch.tensor((rng.randn(3, 3, Ci, Co) * 0.05).astype(np.float32)).cuda()
        b = torch.zeros(Co, device="cuda")
        wf = ops.conv3x3_pack(w, 0)
        full = ops.conv3x3_fwd(x, wf, b, Co, True).clone()
        parts = torch.cat([ops.conv3x3_fwd(x[i:i + 2].contiguous(), wf, b, Co, True).clone() for i in (0, 2, 4)])
        one = torch.cat([ops.conv3x3_fwd(x[i:i + 1].contiguous(), wf, b, Co, True).clone() for i in range(6)])
        print("mode", mode, (H, Ci, Co), "6 vs 3x2 equal:", bool(torch.equal(full, parts)), "6 vs 6x1:", bool(torch.equal(full, one)),
              float((full - parts).abs().max()), float((full - one).abs().max()))
