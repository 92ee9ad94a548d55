"""Times the forward conv kernels at the C3 shapes with torch events: python scripts/gpu_conv_micro.py [label]"""
import sys
import torch
sys.path.insert(0, ".")
from tianshou_amd import dqn as D

shapes = [("conv1", 512, 84, 84, 4, 8, 4, 32), ("conv2", 512, 20, 20, 32, 4, 2, 64), ("conv3", 512, 9, 9, 64, 3, 1, 64),
          ("fc1", 512, 1, 1, 3136, 1, 1, 512), ("sacL2", 4096, 1, 1, 256, 1, 1, 256)]
out = []
for name, B, IH, IW, IC, K, S, OC in shapes:
    x = torch.randn(B, IH, IW, IC, device="cuda")
    wb = torch.randn(K * K * IC + 1, OC, device="cuda") * 0.05
    for _ in range(3):
        D.conv_forward(x, wb, K, K, S, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        D.conv_forward(x, wb, K, K, S, True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    oh, ow = (IH - K) // S + 1, (IW - K) // S + 1
    gf = 2.0 * B * oh * ow * OC * K * K * IC / 1e9
    out.append(f"{name} {us:6.1f}us {gf / us * 1e-3 * 1e3:5.1f}TF/s")
print((sys.argv[1] if len(sys.argv) > 1 else "base") + ": " + " | ".join(out))
