"""CPU, world_size 2 (gloo): the data-parallel DQN step of tianshou_amd.distributed.DataParallelDQN.

As in test_dp_gloo.py the two device steps (`_local_grad`, `_apply`) are oracle-backed test doubles with the
contract of DQNEngine.gradient / apply_gradient; under test is the shipped host logic: one all-reduce of P + 1
floats, the 1/world scaling, identical replicas - against a single-process oracle update on the union batch."""
import os
import socket
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_dqn as OD  # noqa: E402
from tianshou_amd.distributed import DataParallelDQN  # noqa: E402

C, H, W, A, B_LOCAL, STEPS = 2, 44, 36, 3, 12, 2
CFG = OD.DQNConfig(huber_delta=1.0, lr=3e-4, target_update_freq=0, max_grad_norm=0.7)


def make_problem():
    rng = np.random.default_rng(5)
    obs = rng.integers(0, 256, size=(STEPS, 2 * B_LOCAL, C, H, W), dtype=np.uint8)
    act = rng.integers(0, A, size=(STEPS, 2 * B_LOCAL))
    ret = rng.normal(size=(STEPS, 2 * B_LOCAL)).astype(np.float32) * 2
    return OD.init_params(C, H, W, A, seed=9), obs, act, ret


class OracleBackedDP(DataParallelDQN):
    def __init__(self, eng, state):
        super().__init__(eng)
        self.state = state

    def _local_grad(self, obs, act, returns, weight, out):
        p = {k: v.clone().requires_grad_(True) for k, v in self.state.params.items()}
        q = OD.forward(p, obs)
        q = q[torch.arange(len(act)), torch.as_tensor(act)]
        r = torch.as_tensor(returns)
        loss = torch.nn.functional.huber_loss(q.reshape(-1, 1), r.reshape(-1, 1), delta=CFG.huber_delta)
        loss.backward()
        out[: self.eng.P] = torch.cat([p[k].grad.reshape(-1) for k in OD.PARAM_ORDER])
        out[self.eng.P:] = loss.detach()
        return (r - q).detach()

    def _apply(self, grad):
        shapes = OD.param_shapes(C, H, W, A)
        grads, off = {}, 0
        for k in OD.PARAM_ORDER:
            n = int(np.prod(shapes[k]))
            grads[k] = grad[off:off + n].reshape(shapes[k]).clone()
            off += n
        OD._adam(self.state, CFG, grads)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        p0, obs, act, ret = make_problem()
        st = OD.DQNState.create(p0, CFG)
        eng = SimpleNamespace(P=OD.param_count(C, H, W, A), device=torch.device("cpu"))
        dp = OracleBackedDP(eng, st)
        lo, hi = rank * B_LOCAL, (rank + 1) * B_LOCAL
        losses = []
        for s in range(STEPS):
            loss, td = dp.update_with_batch(obs[s, lo:hi], act[s, lo:hi], ret[s, lo:hi])
            losses.append(float(loss))
        q.put((rank, OD.flatten_params(st.params).numpy().copy(), losses))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_dp_dqn_matches_single_process_union_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, params, losses = q.get(timeout=240)
        res[r] = (params, losses)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][0], res[1][0])            # replicas identical
    assert res[0][1] == res[1][1]

    p0, obs, act, ret = make_problem()
    st = OD.DQNState.create(p0, CFG)
    ref_losses = []
    for s in range(STEPS):
        loss, _ = OD.update_with_batch(st, CFG, obs[s], act[s], ret[s])
        ref_losses.append(loss)
    np.testing.assert_allclose(res[0][1], ref_losses, rtol=1e-5)
    np.testing.assert_allclose(res[0][0], OD.flatten_params(st.params).numpy(), rtol=1e-4, atol=0.05 * CFG.lr)
