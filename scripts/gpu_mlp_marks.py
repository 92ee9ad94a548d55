"""Phase marks of one fused-MLP launch (workgroup 0, thread 0; s_memtime ticks, ~2.36 per ns measured against HIP events).
Builds its own copy of the library with -DTS_MLP_MARKS (the shipped one carries neither the marks nor the entry point)."""
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
if "TS_LIB_PATH" not in os.environ:
    from tianshou_amd import build as _b
    os.environ["TS_EXTRA_FLAGS"] = "ts_mlp.hip:-DTS_MLP_MARKS"
    os.environ["TS_LIB_PATH"] = _b.build_library(out=os.path.join(_b.LIBDIR, "libtsengine_mlpmarks.so"))
import torch
from tianshou_amd import sac as S, _lib

lib = _lib.load()
cfg = S.SACConfig()
lay = S.layout(376, 17)
g = torch.Generator().manual_seed(0)
eng = S.SACEngine(376, 17, (torch.randn(lay["actor_count"], generator=g) * 0.05).cuda(), (torch.randn(lay["critic_count"], generator=g) * 0.05).cuda(),
                  (torch.randn(lay["critic_count"], generator=g) * 0.05).cuda(), cfg)
obs = torch.randn(4096, 376, device="cuda")
noise = torch.randn(4096, 17, device="cuda")
for _ in range(3):
    eng.policy_forward(obs, noise)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    eng.policy_forward(obs, noise)
e1.record()
torch.cuda.synchronize()
print("policy_forward (pack + fused MLP + policy kernel):", round(e0.elapsed_time(e1) * 1e3 / 20, 1), "us per call")
out = (C.c_uint64 * 8)()
lib.ts_debug_mlp_marks(out)
t = list(out)
names = ["start", "x in LDS", "layer 1", "layer 2", "head"]
print("raw ticks:", [t[k] - t[0] for k in range(5)])
print(" ".join(f"{names[k]}=+{(t[k] - t[k - 1]) / 2.36:.0f}ns" for k in range(1, 5)), "total", round((t[4] - t[0]) / 2.36), "ns")
tr = (C.c_uint64 * 320)()
lib.ts_debug_mlp_trace(tr)
tr = list(tr)
print("per wave: clock (ns after the kernel's first mark) at the start of each 16-deep block: layer 1 slots 0.., layer 2 slots 16.., head 32..")
for w in range(8):
    row = tr[40 * w: 40 * w + 36]
    print(f"wave {w}:", " ".join(f"{(v - t[0]) / 2.36:6.0f}" if v else "     -" for v in row))

# the same marks after one whole SAC update: the last fused launch is the actor's backward pass (no input gradient)
act = torch.tanh(torch.randn(4096, 17, device="cuda"))
ret = torch.randn(4096, device="cuda")
for _ in range(3):
    eng.update_with_batch(obs, act, ret, noise)
torch.cuda.synchronize()
lib.ts_debug_mlp_marks(out)
t = list(out)
names = ["start", "d_out in LDS", "layer 3^T", "layer 2^T"]
print("backward:", " ".join(f"{names[k]}=+{(t[k] - t[k - 1]) / 2.36:.0f}ns" for k in range(1, 4)), "total", round((t[3] - t[0]) / 2.36), "ns")
tr2 = (C.c_uint64 * 320)()
lib.ts_debug_mlp_trace(tr2)
trl = list(tr2)
for w in range(8):
    row = trl[40 * w: 40 * w + 36]
    print(f"wave {w}:", " ".join(f"{(v - t[0]) / 2.36:6.0f}" if v > t[0] else "     -" for v in row))
