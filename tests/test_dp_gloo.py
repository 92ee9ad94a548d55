"""CPU, world_size 2 (gloo): the data-parallel update loop of tianshou_amd.distributed.

The HIP kernels cannot run here, so the two device steps around the collective
(`_local_grad`, `_apply`) are replaced by oracle-backed test doubles with the same contract as
ts_ppo_grad / ts_ppo_apply.  What is under test is the host logic that ships: minibatch line-up
across ranks, 1/global_batch scaling, the single all-reduce per step, global advantage statistics,
loss-part fix-up, identical replicas afterwards - against a single-process oracle run on the
union batch."""
import os
import socket
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_ppo as OP  # noqa: E402
from tianshou_amd.distributed import DataParallelPPO, shard_envs  # noqa: E402
from tianshou_amd.ppo import PPOConfig, split_offsets  # noqa: E402

OBS, ACT, N_LOCAL, BATCH, REPEAT = 17, 6, 96, 40, 2   # ragged: 96 = 40 + 56 (merge_last)


def make_problem(adv_norm: bool):
    rng = np.random.default_rng(3)
    params = OP.init_params(OBS, ACT, seed=1)
    g = torch.Generator().manual_seed(2)
    for k in params:
        params[k] = params[k] + 0.05 * torch.randn(params[k].shape, generator=g)
    n = 2 * N_LOCAL
    data = dict(obs=torch.from_numpy(rng.normal(size=(n, OBS)).astype(np.float32)),
                act=torch.from_numpy(rng.normal(size=(n, ACT)).astype(np.float32)),
                adv=torch.from_numpy(rng.normal(size=n).astype(np.float32)),
                returns=torch.from_numpy(rng.normal(size=n).astype(np.float32)),
                logp_old=torch.from_numpy((rng.normal(size=n) * 0.3 - 8.0).astype(np.float32)),
                v_s=torch.from_numpy(rng.normal(size=n).astype(np.float32)))
    perms = [[rng.permutation(N_LOCAL) for _ in range(REPEAT)] for _ in range(2)]
    kw = dict(eps_clip=0.2, vf_coef=0.25, ent_coef=0.01, max_grad_norm=0.5, value_clip=True,
              advantage_normalization=adv_norm, lr=3e-4)
    return params, data, perms, kw


class OracleBackedDP(DataParallelPPO):
    """Test double: same contract as ts_ppo_grad / ts_ppo_apply, computed by the CPU oracle."""

    def __init__(self, eng, ocfg, group=None):
        super().__init__(eng, group)
        self.ocfg = ocfg
        self.state = OP.PPOState(params=OP.unflatten_params(eng.params, OBS, ACT))

    def _pack(self, b):
        return b

    def _begin_update(self):
        pass

    def _local_grad(self, rec, rows, global_batch, adv_stats, out):
        import dataclasses

        cfg = dataclasses.replace(self.ocfg, advantage_normalization=False)
        adv = rec["adv"][rows]
        if adv_stats is not None:
            adv = (adv - adv_stats[0]) / (adv_stats[1] + 1e-8)
        p = {k: v.detach().clone().requires_grad_(True) for k, v in self.state.params.items()}
        loss, clip, vf, ent = OP.ppo_minibatch_loss(p, cfg, rec["obs"][rows], rec["act"][rows], adv,
                                                    rec["returns"][rows], rec["logp_old"][rows], rec["v_s"][rows])
        scale = rows.numel() / global_batch      # local mean -> local sum / global_batch
        (loss * scale).backward()
        grads = [p[k].grad if p[k].grad is not None else torch.zeros_like(p[k]) for k in OP.PARAM_ORDER]
        out[: self.eng.P] = torch.cat([g.reshape(-1) for g in grads])
        # the entropy part of d loss / d sigma must not be scaled per rank twice: it already is
        out[self.eng.P:] = torch.stack([loss * scale, clip * scale, vf * scale, ent]).detach()

    def _apply(self, grad):
        flat = grad[: self.eng.P]
        shapes = OP.param_shapes(OBS, ACT)
        grads, off = {}, 0
        for k in OP.PARAM_ORDER:
            n = int(np.prod(shapes[k]))
            grads[k] = flat[off:off + n].reshape(shapes[k]).clone()
            off += n
        if self.ocfg.max_grad_norm is not None:
            norm = torch.sqrt(sum((g * g).sum() for g in grads.values()))
            coef = torch.clamp(self.ocfg.max_grad_norm / (norm + 1e-6), max=1.0)
            grads = {k: g * coef for k, g in grads.items()}
        OP._adam_step(self.state, self.ocfg, grads)
        self.eng.adam_step += 1
        self.eng.params = OP.flatten_params(self.state.params)


def _worker(rank, world, port, adv_norm, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        params, data, perms, kw = make_problem(adv_norm)
        lo, hi = rank * N_LOCAL, (rank + 1) * N_LOCAL
        local = {k: v[lo:hi] for k, v in data.items()}
        eng = SimpleNamespace(params=OP.flatten_params(params), adam_step=0, P=OP.flatten_params(params).numel(),
                              cfg=PPOConfig(**kw), obs_dim=OBS, act_dim=ACT, device=torch.device("cpu"))
        dp = OracleBackedDP(eng, OP.PPOConfig(**kw))
        losses, steps = dp.update(local, BATCH, REPEAT, perms[rank])
        q.put((rank, eng.params.numpy().copy(), losses.numpy().copy(), steps))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("adv_norm", [False, True])
def test_dp_update_matches_single_process_union_batch(adv_norm):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, adv_norm, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, params, losses, steps = q.get(timeout=180)
        res[r] = (params, losses, steps)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # replicas identical
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1])

    # single process on the union batch: global minibatch k = rank0 rows_k ++ (N_LOCAL + rank1 rows_k)
    params, data, perms, kw = make_problem(adv_norm)
    ocfg = OP.PPOConfig(**kw)
    st = OP.PPOState(params={k: v.clone() for k, v in params.items()})
    offs = split_offsets(N_LOCAL, BATCH)
    ref_losses = []
    for r in range(REPEAT):
        for lo, hi in zip(offs[:-1], offs[1:]):
            rows = np.concatenate([perms[0][r][lo:hi], N_LOCAL + perms[1][r][lo:hi]])
            pre = {k: data[k][rows] for k in ("v_s", "returns", "adv", "logp_old")}
            out = OP.update(st, ocfg, {"obs": data["obs"][rows], "act": data["act"][rows]}, pre, None, 1,
                            [np.arange(len(rows))])
            ref_losses.append(out[0])
    assert res[0][2] == len(ref_losses)
    np.testing.assert_allclose(res[0][1], np.asarray(ref_losses), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(res[0][0], OP.flatten_params(st.params).numpy(), rtol=1e-4, atol=2e-6)


def test_shard_envs_partitions_everything():
    for n_env, world in [(512, 8), (10, 4), (3, 8), (7, 1)]:
        spans = [shard_envs(n_env, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n_env
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
