"""Dev probe: is the fused forward bitwise reproducible run to run (integer splat => it should be)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dpc_amd
for (B, N, D, K, sig) in [(4, 8000, 64, 11, 1.6), (4, 8000, 64, 21, 3.0), (2, 16000, 64, 5, 1.0), (2, 9000, 128, 21, 3.0), (2, 40000, 64, 11, 1.6)]:
    inp = dpc_amd.synthetic.make_inputs(B, N, 5150)
    cfg = dpc_amd.default_config(vox_size=D, pc_gauss_kernel_size=K)
    kern = dpc_amd.smoothing_kernel(cfg, sig, device="cuda")
    t = lambda a: torch.tensor(a, device="cuda", requires_grad=True)
    pc, pose, scale = t(inp["pc"]), t(inp["pose"]), t(inp["scale"])
    outs = [dpc_amd.pointcloud_project_fast(cfg, pc, pose, None, None, kern, scaling_factor=scale) for _ in range(5)]
    d = [float((o["proj"].detach() - outs[0]["proj"].detach()).abs().max()) for o in outs]
    print((B, N, D, K), "proj", d)
