"""Phase stamps of the feature-split step kernel (ts_ppo_q.h) from a -DTS_PHASE_MARKS build of the library: shader-clock cycles
of wave 0 of pair 0's actor and critic workgroups behind every barrier, and every workgroup's start / end on the 100 MHz clock.
    NROWS=8192 TS_PPO_STEPQ=2 [TS_PPO_STEPQ_PAIRS=..] python scripts/gpu_stepq_phases.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
if "TS_LIB_PATH" not in os.environ:
    from tianshou_amd import build as _b
    os.environ["TS_EXTRA_FLAGS"] = "ts_ppo.hip:-DTS_PHASE_MARKS"
    os.environ["TS_LIB_PATH"] = _b.build_library(out=os.path.join(_b.LIBDIR, "libtsengine_marks.so"))
import numpy as np, torch
import bench
from tianshou_amd import _lib
from tianshou_amd.ppo import pack_batch

dev = torch.device("cuda", 0)
L = bench.Learner(dev, 0, 1)
b = L.preprocess()
lib = _lib.load()
hp = L.cfg.to_c()
hp.nets = int(os.environ.get("NETS", "0"))
rec = pack_batch(b, 17, 6)
NROWS = int(os.environ.get("NROWS", "65536"))
for trial in range(3):
    rows = torch.as_tensor(np.random.default_rng(trial).permutation(bench.N_TRANS)[:NROWS], device=dev)
    out = (C.c_int64 * 2048)()
    _lib.check(lib.ts_debug_ppo_step_cycles(L.ws.handle, _lib.ptr(L.eng.params), _lib.i64(17), _lib.i64(6),
        _lib.ptr(rec), _lib.ptr(rows), _lib.i64(NROWS), C.byref(hp), out, _lib.i64(2048), _lib.current_stream(dev)))
    t = np.array(list(out), dtype=np.int64)
    if trial == 0:
        continue
    for net, name in ((0, "actor"), (1, "critic")):
        m = t[64 * net:64 * net + 64]
        k_end = int(np.nonzero(m)[0].max())
        n_tiles = (k_end - 2) // 4
        print(f"trial {trial} {name} workgroup 0 wave 0: total {m[k_end] - m[0]} cycles; prologue (entry -> first barrier) {m[1] - m[0]}")
        for i in range(n_tiles):
            q = m[2 + 4 * i:6 + 4 * i]
            prev = m[1] if i == 0 else m[1 + 4 * i]
            print(f"   tile {i}: P1 {q[0] - prev:6d}  P2 {q[1] - q[0]:6d}  P3 {q[2] - q[1]:6d}  P4 {q[3] - q[2]:6d}   sum {q[3] - prev:6d}")
        print(f"   epilogue {m[k_end] - m[k_end - 1]}")
    se = t[128:128 + 2 * 960].reshape(960, 2).astype(np.float64) * 10.0
    live = se[:, 1] > 0
    se = se[live]
    t00 = se[:, 0].min()
    st, en = se[:, 0] - t00, se[:, 1] - t00
    print(f"   {live.sum()} workgroups: start min/median/max {st.min():.0f}/{np.median(st):.0f}/{st.max():.0f} ns, body min/median/max "
          f"{(en - st).min():.0f}/{np.median(en - st):.0f}/{(en - st).max():.0f} ns, end min/median/max {en.min():.0f}/{np.median(en):.0f}/{en.max():.0f} ns")
