"""Second-generation implicit-GEMM kernels (ts_conv2.hip) on the GPU box:
  1. v_permlane32_swap / layout self-check through the public layers: generation 2 against generation 1, bit for bit
     (forward, input gradient) and to rounding (weight gradient: different slab boundaries), plus an fp64 torch reference;
  2. timings of both generations at the Atari-shape PPO minibatch (65,536 samples) and at the C3 / C5 batch sizes.
Usage: python scripts/gpu_conv2_check.py [check|bench|all] [batch]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
import torch.nn.functional as F

from tianshou_amd import _lib
from tianshou_amd import dqn as D

lib = _lib.load()


def gen(mode):
    return lib.ts_conv_set_generation(int(mode))


def ref64(x, wb, K, S, dy, mask):
    IC, OC = x.shape[-1], wb.shape[1]
    w = wb[:-1].double().reshape(K, K, IC, OC).permute(3, 2, 0, 1)
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, wb[-1].double(), stride=S)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    gx = xr.grad.permute(0, 2, 3, 1)
    if mask is not None:
        gx = gx * (mask > 0).double()
    gw = torch.cat([wr.grad.permute(2, 3, 1, 0).reshape(K * K * IC, OC), dy.double().sum((0, 1, 2))[None]], 0)
    return yr.permute(0, 2, 3, 1), gx, gw


def check():
    torch.manual_seed(0)
    shapes = [  # name, B, IH, IW, IC, K, S, OC, u8
        ("conv1u8", 37, 84, 84, 4, 8, 4, 32, True), ("conv1", 37, 84, 84, 4, 8, 4, 32, False),
        ("conv2", 53, 20, 20, 32, 4, 2, 64, False), ("conv3", 96, 9, 9, 64, 3, 1, 64, False),
        ("fc1", 300, 1, 1, 3136, 1, 1, 512, False), ("head", 1000, 1, 1, 512, 1, 1, 32, False),
        ("sacL2", 4096, 1, 1, 256, 1, 1, 256, False), ("sacL1", 700, 1, 1, 384, 1, 1, 256, False),
        ("wide", 513, 1, 1, 64, 1, 1, 3136, False), ("odd19", 19, 19, 19, 32, 4, 2, 32, False),
        ("k1", 5000, 1, 1, 32, 1, 1, 96, False), ("conv2big", 1500, 20, 20, 32, 4, 2, 64, False),
    ]
    bad = 0
    for name, B, IH, IW, IC, K, S, OC, u8 in shapes:
        if u8:
            x = torch.randint(0, 256, (B, IH, IW, IC), device="cuda", dtype=torch.uint8)
        else:
            x = torch.randn(B, IH, IW, IC, device="cuda")
        wb = torch.randn(K * K * IC + 1, OC, device="cuda") * 0.05
        res = {}
        for g in (-1, 1):
            gen(g)
            y = D.conv_forward(x, wb, K, K, S, True)
            dy = torch.randn(y.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
            can_dx = IC % 32 == 0 and K % S == 0
            mask = (torch.rand(x.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(2)) > 0.5).float() \
                if can_dx and name in ("conv2", "fc1", "conv3") else None
            d_wb, dx = D.conv_backward(x, wb, dy, K, K, S, mask=mask, need_dx=can_dx)
            torch.cuda.synchronize()
            res[g] = (y, d_wb, dx, dy, mask)
        y1, w1, dx1, dy, mask = res[-1]
        y2, w2, dx2, _, _ = res[1]
        f_eq = bool(torch.equal(y1, y2))
        d_eq = dx1 is None or bool(torch.equal(dx1, dx2))
        yr, gx, gw = ref64(x.float(), wb, K, S, dy, mask)
        # relu applied by the layer
        e_f = float((y2.double() - yr.clamp(min=0)).abs().max() / yr.abs().max())
        e_w1 = float((w1.double() - gw).abs().max() / gw.abs().max())
        e_w2 = float((w2.double() - gw).abs().max() / gw.abs().max())
        e_d = float((dx2.double() - gx).abs().max() / gx.abs().max()) if dx2 is not None else float("nan")
        # (bit equality holds only for the k-sequential variants, TS_R2_KSEQ=1, where generation 1 does not split K)
        ok = e_f < 5e-6 and e_w2 < max(2e-6, 3 * e_w1) and (dx2 is None or e_d < 5e-6)
        bad += not ok
        print(f"{name:9s} fwd bit-equal {f_eq}  dgrad bit-equal {d_eq}  err64: fwd {e_f:.1e} wgrad v1 {e_w1:.1e} v2 {e_w2:.1e} "
              f"dgrad {e_d:.1e}  {'OK' if ok else 'FAIL'}", flush=True)
    gen(0)
    print("CHECK", "PASSED" if bad == 0 else f"FAILED ({bad})")
    return bad


def timeit(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def bench(B):
    layers = [("conv1u8", 84, 84, 4, 8, 4, 32, True), ("conv2", 20, 20, 32, 4, 2, 64, False),
              ("conv3", 9, 9, 64, 3, 1, 64, False), ("fc1", 1, 1, 3136, 1, 1, 512, False),
              ("head", 1, 1, 512, 1, 1, 32, False)]
    if B <= 8192:
        layers += [("sacL2", 1, 1, 256, 1, 1, 256, False), ("sacL1", 1, 1, 384, 1, 1, 256, False)]
    tot = {-1: [0.0, 0.0, 0.0], 1: [0.0, 0.0, 0.0]}
    flop = [0.0, 0.0, 0.0]
    n = 5 if B >= 16384 else 20
    for name, IH, IW, IC, K, S, OC, u8 in layers:
        if u8:
            x = torch.randint(0, 256, (B, IH, IW, IC), device="cuda", dtype=torch.uint8)
        else:
            x = torch.randn(B, IH, IW, IC, device="cuda").clamp_(min=0)
        wb = torch.randn(K * K * IC + 1, OC, device="cuda") * 0.05
        oh, ow = (IH - K) // S + 1, (IW - K) // S + 1
        gf = 2.0 * B * oh * ow * OC * K * K * IC / 1e9
        dy = torch.randn(B, oh, ow, OC, device="cuda")
        can_dx = IC % 32 == 0
        line = f"B={B} {name:8s} {gf:8.1f} GF |"
        for g in (-1, 1):
            gen(g)
            t_f = timeit(lambda: D.conv_forward(x, wb, K, K, S, True), n)
            t_w = timeit(lambda: D.conv_backward(x, wb, dy, K, K, S, need_dx=False), n)
            t_b = timeit(lambda: D.conv_backward(x, wb, dy, K, K, S, mask=x if can_dx else None, need_dx=can_dx), n)
            t_d = t_b - t_w if can_dx else 0.0
            tot[g][0] += t_f; tot[g][1] += t_w; tot[g][2] += t_d
            line += f" v{1 if g < 0 else 2}: fwd {t_f:8.1f}us {gf / t_f * 1e3:6.1f}TF  wgrad {t_w:8.1f}us {gf / t_w * 1e3:6.1f}TF"
            line += f"  dgrad {t_d:8.1f}us {gf / t_d * 1e3:6.1f}TF |" if can_dx else "  dgrad      -- |"
        flop[0] += gf; flop[1] += gf; flop[2] += gf if can_dx else 0.0
        print(line, flush=True)
        del x, dy
        torch.cuda.empty_cache()
    for g in (-1, 1):
        t = tot[g]
        print(f"B={B} total v{1 if g < 0 else 2}: fwd {t[0] / 1e3:.2f} ms ({flop[0] / t[0] * 1e3:.1f} TF)  wgrad {t[1] / 1e3:.2f} ms "
              f"({flop[1] / t[1] * 1e3:.1f} TF)  dgrad {t[2] / 1e3:.2f} ms ({flop[2] / t[2] * 1e3:.1f} TF)  "
              f"all {sum(t) / 1e3:.2f} ms = {sum(flop) / sum(t) * 1e3:.1f} TF/s = {sum(flop) / sum(t) * 1e3 / 157.3:.3f} of peak")
    gen(0)


def sweep(B):
    """rows2 variants (waves per workgroup, row tiles per wave, k order) per layer, forward and input gradient."""
    layers = [("conv1u8", 84, 84, 4, 8, 4, 32, True), ("conv2", 20, 20, 32, 4, 2, 64, False),
              ("conv3", 9, 9, 64, 3, 1, 64, False), ("fc1", 1, 1, 3136, 1, 1, 512, False)]
    gen(1)
    for name, IH, IW, IC, K, S, OC, u8 in layers:
        if u8:
            x = torch.randint(0, 256, (B, IH, IW, IC), device="cuda", dtype=torch.uint8)
        else:
            x = torch.randn(B, IH, IW, IC, device="cuda").clamp_(min=0)
        wb = torch.randn(K * K * IC + 1, OC, device="cuda") * 0.05
        oh, ow = (IH - K) // S + 1, (IW - K) // S + 1
        gf = 2.0 * B * oh * ow * OC * K * K * IC / 1e9
        dy = torch.randn(B, oh, ow, OC, device="cuda")
        can_dx = IC % 32 == 0
        t_w = timeit(lambda: D.conv_backward(x, wb, dy, K, K, S, need_dx=False), 5)
        for waves, tm in ((8, 2), (16, 1), (16, 2)):
            for kseq in (0,):
                os.environ["TS_R2_WAVES"], os.environ["TS_R2_TM"], os.environ["TS_R2_KSEQ"] = str(waves), str(tm), str(kseq)
                t_f = t_d = float("nan")
                try:
                    t_f = timeit(lambda: D.conv_forward(x, wb, K, K, S, True), 5)
                except Exception:  # variant not instantiated for this shape
                    pass
                try:
                    if can_dx:
                        t_d = timeit(lambda: D.conv_backward(x, wb, dy, K, K, S, mask=x, need_dx=True), 5) - t_w
                except Exception:
                    pass
                print(f"B={B} {name:8s} waves {waves:2d} tm {tm} kseq {kseq}: fwd {t_f:8.1f}us {gf / t_f * 1e3:6.1f}TF  "
                      f"dgrad {t_d:8.1f}us {gf / t_d * 1e3:6.1f}TF", flush=True)
        for k in ("TS_R2_WAVES", "TS_R2_TM", "TS_R2_KSEQ"):
            os.environ.pop(k, None)
        del x, dy
        torch.cuda.empty_cache()
    gen(0)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    rc = 0
    if what in ("check", "all"):
        rc = check()
    if what == "sweep":
        for b in ([int(a) for a in sys.argv[2:]] or [65536]):
            sweep(b)
    if what in ("bench", "all"):
        for b in ([int(a) for a in sys.argv[2:]] or [65536, 4096, 512]):
            bench(b)
    sys.exit(1 if rc else 0)
