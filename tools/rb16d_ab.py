"""A/B of the split-limb GEMM with A moved global -> LDS as limb planes (NFS_RB16D=1, winograd_gemm_rb16d_kernel) against
the shipped rb16s kernel: bit-identity of the conv result and the GEMM-only time (library event pairs) per deep layer.
    python tools/rb16d_ab.py           (spawns itself once per setting)"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import hashlib, json
    import torch
    import neural_flow_style_amd.ops as ops
    from neural_flow_style_amd import _lib
    L = _lib.lib()
    ops.gemm_mode(1)
    out = {}
    layers = [("conv3_1", 50, 128, 256), ("conv3_2", 50, 256, 256), ("conv4_1", 25, 256, 512), ("conv4_2", 25, 512, 512),
              ("conv5_1", 12, 512, 512), ("conv3_2 dgrad", 50, 256, 256), ("conv4_2 dgrad", 25, 512, 512)]
    for name, HW, Ci, Co in layers:
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        x = torch.relu(torch.randn(8, HW, HW, Ci, device="cuda", generator=g))
        w = torch.randn(3, 3, Ci, Co, device="cuda", generator=g) * 0.03
        b = torch.zeros(Co, device="cuda")
        wf = ops.conv3x3_pack(w, 0)
        if "dgrad" in name:
            wd = ops.conv3x3_pack(w, 1)
            gy = torch.randn(8, HW, HW, Co, device="cuda", generator=g)
            f = lambda: ops.conv3x3_dgrad(gy, wd, Ci, x_in=x)
        else:
            f = lambda: ops.conv3x3_fwd(x, wf, b, Co, True)
        for _ in range(3):
            y = f()
        torch.cuda.synchronize()
        L.nfs_gemm_timer(1)
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
        L.nfs_gemm_timer_read(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
        ms1, fl1, n1, by1 = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
        L.nfs_gemm_timer_read_kind(1, ctypes.byref(ms1), ctypes.byref(fl1), ctypes.byref(n1), ctypes.byref(by1))
        L.nfs_gemm_timer(0)
        out[name] = {"sha": hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16],
                     "gemm_us": 1e3 * (ms.value + ms1.value) / max(n.value + n1.value, 1), "launches": n.value + n1.value}
    print(json.dumps(out))
else:
    import json
    res = {}
    for v in ("0", "1"):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, NFS_RB16D=v), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("NFS_RB16D=%s failed:\n%s" % (v, r.stderr[-2000:])); continue
        res[v] = json.loads(line[-1])
    for name in res.get("0", {}):
        a, b = res["0"][name], res.get("1", {}).get(name)
        print("%-8s rb16s %6.1f us | rb16d %s | bit-identical: %s" % (name, a["gemm_us"], "%6.1f us" % b["gemm_us"] if b else "-",
                                                                    (a["sha"] == b["sha"]) if b else "-"))
