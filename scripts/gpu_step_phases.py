"""Phase timing of one ppo_step_kernel launch (workgroup 0, wave 0), shader-clock cycles.
NETS=1 | 2 in the environment: the one-network kernel (ts_ppo_hparams.nets) on the actor / the critic alone -- only that half's
marks are written (the other half's slots stay 0, read the "(at ...)" column); NROWS: rows per launch.
Builds its own copy of the library with -DTS_PHASE_MARKS (the shipped one carries neither the marks nor the entry point)."""
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
if "TS_LIB_PATH" not in os.environ:
    from tianshou_amd import build as _b
    os.environ["TS_EXTRA_FLAGS"] = "ts_ppo.hip:-DTS_PHASE_MARKS"
    os.environ["TS_LIB_PATH"] = _b.build_library(out=os.path.join(_b.LIBDIR, "libtsengine_marks.so"))
import numpy as np, torch
import bench
from tianshou_amd import _lib

dev = torch.device("cuda", 0)
L = bench.Learner(dev, 0, 1)
b = L.preprocess()
lib = _lib.load()
hp = L.cfg.to_c()
hp.nets = int(os.environ.get("NETS", "0"))      # NETS=1 / 2: ppo_step1_kernel on the actor / the critic alone (one half's phases)
from tianshou_amd.ppo import pack_batch
rec = pack_batch(b, 17, 6)
V1 = os.environ.get("TS_PPO_STEP_V1", "0") not in ("", "0")
if V1:
    names = ["start", "staged", "A:trunk", "A:loss", "A:headgrad", "A:dH1", "A:dW2", "A:dW1", "A:pass_end", "A:flushed",
             "C:trunk", "C:loss", "C:headgrad", "C:dH1", "C:dW2", "C:dW1", "C:pass_end", "end"]
else:
    names = ["start", "staged", "A:trunk", "A:loss", "A:headgrad", "A:dH1", "A:tiles_A", "A:dW2", "A:dW1/misc", "C:staged",
             "C:trunk", "C:loss", "C:headgrad", "C:dH1", "C:tiles_A", "C:dW2", "C:dW1/misc", "end"]
NROWS = int(os.environ.get('NROWS', '65536'))
for trial in range(2):
    rows = torch.as_tensor(np.random.default_rng(trial).permutation(bench.N_TRANS)[:NROWS], device=dev)
    out = (C.c_int64 * 2048)()
    _lib.check(lib.ts_debug_ppo_step_cycles(L.ws.handle, _lib.ptr(L.eng.params), _lib.i64(17), _lib.i64(6),
        _lib.ptr(rec), _lib.ptr(rows), _lib.i64(NROWS), C.byref(hp), out, _lib.i64(2048), _lib.current_stream(dev)))
    t = np.array(list(out), dtype=np.int64)
    print("trial", trial, "total cycles", t[17] - t[0])
    for k in range(1, 18):
        print(f"   {names[k]:12s} +{t[k]-t[k-1]:8d}  (at {t[k]-t[0]:8d})")

    # per-workgroup (start, end) on the chip-wide 100 MHz clock: dispatch skew, body length, drain
    n_wg = min(512, (NROWS + 127) // 128)
    se = t[64:64 + 2 * n_wg].reshape(n_wg, 2).astype(np.float64) * 10.0      # ns
    t00 = se[:, 0].min()
    st, en = se[:, 0] - t00, se[:, 1] - t00
    dur = en - st
    print(f"   {n_wg} workgroups: start min/median/max {st.min():.0f}/{np.median(st):.0f}/{st.max():.0f} ns, "
          f"body min/median/max {dur.min():.0f}/{np.median(dur):.0f}/{dur.max():.0f} ns, end min/median/max {en.min():.0f}/{np.median(en):.0f}/{en.max():.0f} ns")
    order = np.argsort(st)
    print("   start deciles (ns):", " ".join(f"{v:.0f}" for v in np.percentile(st, np.arange(0, 101, 10))))
    print("   end   deciles (ns):", " ".join(f"{v:.0f}" for v in np.percentile(en, np.arange(0, 101, 10))))
    print("   body  deciles (ns):", " ".join(f"{v:.0f}" for v in np.percentile(dur, np.arange(0, 101, 10))))

    hw = t[64 + 1024:64 + 1024 + n_wg]
    hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
    cu, sh, se = (hwid >> 8) & 0xf, (hwid >> 12) & 0x1, (hwid >> 13) & 0x7      # gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    place = {}
    for b in range(n_wg):
        place.setdefault((int(xcc[b]), int(se[b]), int(sh[b]), int(cu[b])), []).append(b)
    sizes = sorted(set(len(v) for v in place.values()))
    diffs = sorted(set(abs(v[1] - v[0]) for v in place.values() if len(v) == 2))
    print(f"   placement: {len(place)} distinct (xcc, se, sh, cu), workgroups per CU {sizes}, block-index distance of the two on a CU {diffs[:8]}")
    slow = dur > np.median(dur)
    print("   slow half are blocks >= n/2:", float(np.mean(slow[n_wg // 2:])), " (share of the upper block indices that are slow)")
    print("   xcc of blocks 0..15:", [int(x) for x in xcc[:16]])
