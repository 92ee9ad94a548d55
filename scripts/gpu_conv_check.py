"""Row-GEMM kernels against torch (fp64 reference) at the C3 / C5 layer shapes: max error relative to the output scale."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import torch.nn.functional as F
from tianshou_amd import dqn as D

torch.manual_seed(0)
shapes = [("conv2m", 512, 20, 20, 32, 4, 2, 64), ("conv3m", 512, 9, 9, 64, 3, 1, 64), ("conv1", 96, 84, 84, 4, 8, 4, 32), ("conv2", 96, 20, 20, 32, 4, 2, 64), ("conv3", 96, 9, 9, 64, 3, 1, 64),
          ("fc1", 96, 1, 1, 3136, 1, 1, 512), ("sacL2", 4096, 1, 1, 256, 1, 1, 256), ("sacL1", 4096, 1, 1, 416, 1, 1, 256),
          ("head", 4096, 1, 1, 256, 1, 1, 32), ("big", 70000, 1, 1, 64, 1, 1, 64)]
for name, B, IH, IW, IC, K, S, OC in shapes:
    x = torch.randn(B, IH, IW, IC, device="cuda")
    wb = torch.randn(K * K * IC + 1, OC, device="cuda") * 0.05
    y = D.conv_forward(x, wb, K, K, S, False)
    w = wb[:-1].double().reshape(K, K, IC, OC).permute(3, 2, 0, 1)
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, wb[-1].double(), stride=S)
    e_f = float((y.double() - yr.permute(0, 2, 3, 1)).abs().max() / yr.abs().max())
    dy = torch.randn_like(y)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    if IC % 32 == 0:
        mask = (torch.rand_like(x) > 0.5).float() if name.endswith("m") else None
        d_wb, dx = D.conv_backward(x, wb, dy, K, K, S, mask=mask)
        gx = xr.grad.permute(0, 2, 3, 1) * (mask.double() if mask is not None else 1.0)
        e_d = float((dx.double() - gx).abs().max() / gx.abs().max())
    else:
        d_wb, _ = D.conv_backward(x, wb, dy, K, K, S, need_dx=False)
        e_d = float("nan")
    gw = wr.grad.permute(2, 3, 1, 0).reshape(K * K * IC, OC)
    e_w = float((d_wb[:-1].double() - gw).abs().max() / gw.abs().max())
    print(f"{name:6s} fwd {e_f:.2e}  dgrad {e_d:.2e}  wgrad {e_w:.2e}")

