"""Where sample_random_kernel's time goes: ts_sample_indices_seeded timed with HIP events over batch sizes and sub-buffer
counts (one workgroup: the launch, the serial cdf / prefix passes of thread 0, the Philox draws and searches)."""
import ctypes as C

import torch

from tianshou_amd import _lib

lib = _lib.load()
dev = torch.device("cuda")
err = torch.zeros(1, dtype=torch.int32, device=dev)
for E in (1, 16, 64, 65, 512, 1024):
    for bs in (64, 1024, 4096, 16384):
        T = (1 << 21) // E
        offset = torch.arange(E, dtype=torch.int64, device=dev) * T
        lengths = torch.full((E,), T, dtype=torch.int64, device=dev)
        out = torch.empty(bs, dtype=torch.int64, device=dev)

        def run(n):
            for i in range(n):
                _lib.check(lib.ts_sample_indices_seeded(_lib.ptr(offset), _lib.i64(E), _lib.ptr(lengths), C.c_uint64(7),
                                                       C.c_uint64(i), _lib.i64(bs), _lib.ptr(out), _lib.ptr(err),
                                                       _lib.current_stream(dev)))
        run(5)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run(200)
        b.record()
        torch.cuda.synchronize()
        print(f"E {E:5d} bs {bs:6d}: {a.elapsed_time(b) / 200 * 1e3:7.2f} us per call (back to back)", flush=True)
