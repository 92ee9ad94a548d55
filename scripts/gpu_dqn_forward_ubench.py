"""How much the three B = 512 forward passes of a C3 DQN update (Q_online(s), Q_online(s'), Q_target(s') on three streams) would
gain as ONE B = 1,024 pass of the online network + one B = 512 pass of the target network on two streams."""
import torch

import bench_dqn as BD
from tianshou_amd import dqn as D

dev = torch.device("cuda")
tensors = [t for m in BD.torch_layers() for t in (m.weight, m.bias)]
cfg = D.DQNConfig(gamma=0.99, n_step=3, target_update_freq=500, is_double=True, huber_delta=1.0, lr=1e-4)
eng = D.DQNEngine(BD.C, BD.H, BD.W, BD.N_ACT, D.flat_from_torch(tensors, BD.C, BD.H, BD.W, BD.N_ACT), cfg)
g = torch.Generator(device=dev).manual_seed(0)
x1024 = torch.randint(0, 256, (1024, BD.H, BD.W, BD.C), generator=g, device=dev, dtype=torch.uint8)
xa, xb = x1024[:512].contiguous(), x1024[512:].contiguous()
streams = [torch.cuda.Stream() for _ in range(3)]


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def fan(jobs):
    def run():
        main = torch.cuda.current_stream()
        for st, (x, p) in zip(streams, jobs):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                eng.forward(x, params=p, want_act=False)
        for st, _ in zip(streams, jobs):
            main.wait_stream(st)
    return run


print("one pass B=512            : %7.1f us" % timed(lambda: eng.forward(xa, want_act=False)))
print("one pass B=1024           : %7.1f us" % timed(lambda: eng.forward(x1024, want_act=False)))
print("3 x B=512 on three streams: %7.1f us" % timed(fan([(xa, eng.params), (xb, eng.params), (xb, eng.params_old)])))
print("B=1024 + B=512 on two     : %7.1f us" % timed(fan([(x1024, eng.params), (xb, eng.params_old)])))
print("3 x B=512 on one stream   : %7.1f us" % timed(lambda: [eng.forward(xa, want_act=False), eng.forward(xb, want_act=False), eng.forward(xb, params=eng.params_old, want_act=False)]))
