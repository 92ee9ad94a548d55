"""CPU, world_size 2 (gloo): the data-parallel SAC update of tianshou_amd.distributed.DataParallelSAC.

As in test_dp_dqn_gloo.py the device steps (`_begin`, `_phase`, `_sizes`) are oracle-backed test doubles with the
contract of SACEngine.begin_phased_update / update_phase (ts_sac_update_phase: critic grad, critic apply, actor grad,
actor apply); under test is the shipped host logic: the two exchanges per update, the 1/world scaling, the alpha step
from the reduced mean log-probability, identical replicas - against a single-process oracle update (sac.py:298-336) on
the union batch.  The phase order / bit-identity of the real kernels is tests/test_gpu_sac.py."""
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

from oracle import oracle_sac as OS  # noqa: E402
from tianshou_amd.distributed import DataParallelSAC  # noqa: E402

OBS, ACT, HID, B_LOCAL, STEPS = 11, 3, 32, 24, 3
CFG = OS.SACConfig(auto_alpha=True, target_entropy=-3.0, log_alpha0=-0.5, actor_lr=1e-3, critic_lr=1e-3, alpha_lr=3e-3, tau=0.01)


def make_problem():
    rng = np.random.default_rng(11)
    n = 2 * B_LOCAL
    obs = rng.normal(size=(STEPS, n, OBS)).astype(np.float32)
    act = np.tanh(rng.normal(size=(STEPS, n, ACT))).astype(np.float32)
    ret = rng.normal(size=(STEPS, n)).astype(np.float32) * 2
    noise = rng.normal(size=(STEPS, n, ACT)).astype(np.float32)
    weight = rng.random(size=(STEPS, n)).astype(np.float32) + 0.5
    return OS.init_sac_params(OBS, ACT, seed=3, hidden=HID), obs, act, ret, noise, weight


def _unflat(vec, shapes, order):
    out, off = {}, 0
    for k in order:
        n = int(np.prod(shapes[k]))
        out[k] = vec[off:off + n].reshape(shapes[k]).clone()
        off += n
    return out


class OracleBackedDP(DataParallelSAC):
    PHASES = SimpleNamespace(PHASE_CRITIC_GRAD=1, PHASE_CRITIC_APPLY=2, PHASE_ACTOR_GRAD=4, PHASE_ACTOR_APPLY=8)

    def __init__(self, state, allreduce=None):
        eng = SimpleNamespace(device=torch.device("cpu"), **vars(self.PHASES))
        super().__init__(eng, allreduce=allreduce)
        self.st = state
        self.pc = sum(int(np.prod(s)) for s in OS.critic_shapes(OBS, ACT, HID).values())
        self.pa = sum(int(np.prod(s)) for s in OS.actor_shapes(OBS, ACT, HID).values())

    def _sizes(self):
        return 2 * self.pc, self.pa + 1

    def _begin(self, obs, act, returns, noise, weight):
        t = lambda x: torch.as_tensor(x, dtype=torch.float32)  # noqa: E731
        return {"obs": t(obs), "act": t(act), "returns": t(returns).flatten(), "noise": t(noise), "weight": t(weight),
                "stats": torch.zeros(5), "w_out": torch.empty(len(obs))}

    def _phase(self, ctx, phase, buf):
        st, pc, pa = self.st, self.pc, self.pa
        if phase == 1:                                               # ddpg.py:279-284 on the local batch
            tds = []
            for k, name in enumerate(("critic1", "critic2")):
                p = {n: v.clone().requires_grad_(True) for n, v in getattr(st, name).items()}
                td = OS.critic_forward(p, ctx["obs"], ctx["act"]).flatten() - ctx["returns"]
                loss = (td.pow(2) * ctx["weight"]).mean()
                g = OS._grads(loss, p)
                buf[k * pc:(k + 1) * pc] = OS.flatten(g, OS.CRITIC_ORDER)
                ctx["stats"][1 + k] = loss.detach()
                tds.append(td.detach())
            ctx["tds"] = tds
        elif phase == 2:
            for k, (name, opt) in enumerate((("critic1", st.opt_c1), ("critic2", st.opt_c2))):
                g = _unflat(buf[k * pc:(k + 1) * pc], OS.critic_shapes(OBS, ACT, HID), OS.CRITIC_ORDER)
                setattr(st, name, opt.apply(getattr(st, name), g))
        elif phase == 4:                                             # sac.py:308-314 with the updated critics
            alpha = OS.alpha_value(st, CFG)
            p = {n: v.clone().requires_grad_(True) for n, v in st.actor.items()}
            a, logp, _, _ = OS.policy_forward(p, ctx["obs"], ctx["noise"])
            q = torch.min(OS.critic_forward(st.critic1, ctx["obs"], a).flatten(), OS.critic_forward(st.critic2, ctx["obs"], a).flatten())
            loss = (alpha * logp.flatten() - q).mean()
            buf[:pa] = OS.flatten(OS._grads(loss, p), OS.ACTOR_ORDER)
            buf[pa] = -logp.detach().mean()
            ctx["stats"][0] = loss.detach()
        else:
            st.actor = st.opt_actor.apply(st.actor, _unflat(buf[:pa], OS.actor_shapes(OBS, ACT, HID), OS.ACTOR_ORDER))
            mean_def = CFG.target_entropy - buf[pa]                  # mean(target_entropy + log_prob), global
            ctx["stats"][4] = -(st.log_alpha * mean_def)
            st.log_alpha = st.opt_alpha.apply({"a": st.log_alpha}, {"a": (-mean_def).reshape(())})["a"]
            ctx["stats"][3] = st.log_alpha.exp()
            ctx["w_out"] = (ctx["tds"][0] + ctx["tds"][1]) / 2.0
            for old, new in ((st.critic1_old, st.critic1), (st.critic2_old, st.critic2)):
                for k in old:
                    old[k] = CFG.tau * new[k] + (1 - CFG.tau) * old[k]


def _flat_state(st):
    return torch.cat([OS.flatten(st.actor, OS.ACTOR_ORDER), OS.flatten(st.critic1, OS.CRITIC_ORDER),
                      OS.flatten(st.critic2, OS.CRITIC_ORDER), OS.flatten(st.critic1_old, OS.CRITIC_ORDER),
                      st.log_alpha.reshape(1)]).numpy().copy()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        (actor, c1, c2), obs, act, ret, noise, weight = make_problem()
        st = OS.SACState.create(actor, c1, c2, CFG)
        sizes = []

        def counted(buf):                    # the `allreduce=` hook (tianshou_amd.collective.NativeAllReduce on a GPU)
            sizes.append(buf.numel())
            dist.all_reduce(buf)

        dp = OracleBackedDP(st, allreduce=counted if rank == 0 else None)
        lo, hi = rank * B_LOCAL, (rank + 1) * B_LOCAL
        stats = []
        for s in range(STEPS):
            out, w = dp.update_with_batch(obs[s, lo:hi], act[s, lo:hi], ret[s, lo:hi], noise[s, lo:hi], weight[s, lo:hi])
            stats.append(out.numpy().copy())
        if rank == 0:                        # two exchanges per update: critics (+2 losses), actor (+log-prob mean, +loss)
            assert sizes == [2 * dp.pc + 2, dp.pa + 2] * STEPS, sizes
        q.put((rank, _flat_state(st), np.stack(stats), w.numpy().copy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_dp_sac_matches_single_process_union_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, flat, stats, w = q.get(timeout=240)
        res[r] = (flat, stats, w)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][0], res[1][0])            # replicas identical, bit for bit
    assert np.array_equal(res[0][1], res[1][1])

    (actor, c1, c2), obs, act, ret, noise, weight = make_problem()
    st = OS.SACState.create(actor, c1, c2, CFG)
    ref_stats = []
    for s in range(STEPS):
        o = OS.update_with_batch(st, CFG, obs[s], act[s], ret[s], noise[s], weight[s])
        ref_stats.append([o["actor_loss"], o["critic1_loss"], o["critic2_loss"], o["alpha"], o["alpha_loss"]])
    # mean of the two local means == mean over the union batch (equal shard sizes), up to fp32 summation order;
    # parameters on the scale of an Adam step
    np.testing.assert_allclose(res[0][1], np.asarray(ref_stats, dtype=np.float32), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(res[0][0], _flat_state(st), rtol=1e-5, atol=0.02 * CFG.actor_lr)
    w_ref = o["weight"].numpy()
    np.testing.assert_allclose(np.concatenate([res[0][2], res[1][2]]), w_ref, rtol=1e-5, atol=1e-6)
