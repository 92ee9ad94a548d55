import os, sys
sys.path.insert(0, os.getcwd())
import torch
from tianshou_amd import _lib
from tianshou_amd import dqn as D
lib = _lib.load(); lib.ts_conv_set_generation(1)
def timeit(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 65536
for name, IH, IC, K, S, OC in (("conv2", 20, 32, 4, 2, 64), ("conv3", 9, 64, 3, 1, 64), ("fc1", 1, 3136, 1, 1, 512)):
    x = torch.randn(B, IH, IH, IC, device="cuda").clamp_(min=0)
    wb = torch.randn(K * K * IC + 1, OC, device="cuda") * 0.05
    oh = (IH - K) // S + 1
    dy = torch.randn(B, oh, oh, OC, device="cuda")
    t_w = timeit(lambda: D.conv_backward(x, wb, dy, K, K, S, need_dx=False))
    t_m = timeit(lambda: D.conv_backward(x, wb, dy, K, K, S, mask=x, need_dx=True)) - t_w
    t_n = timeit(lambda: D.conv_backward(x, wb, dy, K, K, S, mask=None, need_dx=True)) - t_w
    print(f"{name}: dgrad with mask {t_m:.0f} us, without {t_n:.0f} us (upper bound of what a 1-bit mask can save: {t_m - t_n:.0f} us)")
    del x, dy
