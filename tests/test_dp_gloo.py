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
N_SHARDS = {"equal": (N_LOCAL, N_LOCAL), "uneven": (N_LOCAL, N_LOCAL - 9)}    # shard_envs: sizes may differ by one env


def make_problem(adv_norm: bool, shards=(N_LOCAL, N_LOCAL)):
    rng = np.random.default_rng(3)
    params = OP.init_params(OBS, ACT, seed=1)
    g = torch.Generator().manual_seed(2)
    for k in params:
        params[k] = params[k] + 0.05 * torch.randn(params[k].shape, generator=g)
    n = sum(shards)
    data = dict(obs=torch.from_numpy(rng.normal(size=(n, OBS)).astype(np.float32)),
                act=torch.from_numpy(rng.normal(size=(n, ACT)).astype(np.float32)),
                adv=torch.from_numpy(rng.normal(size=n).astype(np.float32)),
                returns=torch.from_numpy(rng.normal(size=n).astype(np.float32)),
                logp_old=torch.from_numpy((rng.normal(size=n) * 0.3 - 8.0).astype(np.float32)),
                v_s=torch.from_numpy(rng.normal(size=n).astype(np.float32)))
    perms = [[rng.permutation(shards[r]) for _ in range(REPEAT)] for r in range(2)]
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


def _worker(rank, world, port, adv_norm, q, shards=(N_LOCAL, N_LOCAL)):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        params, data, perms, kw = make_problem(adv_norm, shards)
        lo = sum(shards[:rank])
        hi = lo + shards[rank]
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


@pytest.mark.parametrize("adv_norm,layout", [(False, "equal"), (True, "equal"), (True, "uneven")])
def test_dp_update_matches_single_process_union_batch(adv_norm, layout):
    shards = N_SHARDS[layout]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, adv_norm, q, shards)) for r in range(2)]
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

    # single process on the union batch: global minibatch k = rank0 rows_k ++ (n_0 + rank1 rows_k); on uneven shards the
    # boundaries of Batch.split are taken on the largest shard and scaled to each local size (DataParallelPPO._line_up)
    params, data, perms, kw = make_problem(adv_norm, shards)
    ocfg = OP.PPOConfig(**kw)
    st = OP.PPOState(params={k: v.clone() for k, v in params.items()})
    ref = split_offsets(max(shards), BATCH)
    offs_r = [[0] + [min(n, (o * n + max(shards) // 2) // max(shards)) for o in ref[1:-1]] + [n] for n in shards]
    assert offs_r[0] == ref                                   # the largest shard splits exactly like Batch.split
    ref_losses = []
    for r in range(REPEAT):
        for c in range(len(ref) - 1):
            rows = np.concatenate([perms[0][r][offs_r[0][c]:offs_r[0][c + 1]],
                                   shards[0] + perms[1][r][offs_r[1][c]:offs_r[1][c + 1]]])
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


# ------------------------------------------------------------------------------------ global return statistics
def _rms_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tianshou_amd.ppo import rms_merge

        rng = np.random.default_rng(5)
        chunks = [rng.normal(loc=2.0, scale=3.0, size=n) for n in (1000, 731, 12, 2048)]     # 2 updates x 2 ranks
        eng = SimpleNamespace(device=torch.device("cpu"))
        dp = DataParallelPPO(eng)
        rms = [0.0, 1.0, 0.0]                                  # RunningMeanStd initial state (statistics.py:81-91)
        for u in range(2):
            x = chunks[2 * u + rank]
            s1, s2, n = dp._reduce_stats(float(x.sum()), float((x * x).sum()), float(x.size))
            rms = rms_merge(rms, s1, s2, n)
        q.put((rank, rms))
    finally:
        dist.destroy_process_group()


def test_dp_ret_rms_is_the_union_batch_statistics():
    """a2c.py:146-148 on sharded data: after every preprocess both replicas hold the RunningMeanStd a single process
    would have after `ret_rms.update(unnormalized_returns)` on the union of the shards."""
    from oracle import oracle as O

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rms_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1]                                    # bit-identical replicas
    rng = np.random.default_rng(5)
    chunks = [rng.normal(loc=2.0, scale=3.0, size=n) for n in (1000, 731, 12, 2048)]
    m, v, c = 0.0, 1.0, 0.0
    for u in range(2):
        m, v, c = O.rms_update(m, v, c, np.concatenate(chunks[2 * u:2 * u + 2]))
    np.testing.assert_allclose(got[0], [m, v, c], rtol=1e-12)


def test_dp_recompute_advantage_runs_per_repeat():
    """ppo.py:174-178 on the data-parallel path (world 1, no process group): advantages are recomputed before every
    repeat but the first, the records are re-packed, and the step count is repeat x minibatches."""
    calls = []

    class Probe(DataParallelPPO):
        def _pack(self, b):
            calls.append(("pack", float(b["adv"][0])))
            return b

        def _begin_update(self):
            pass

        def _recompute(self, b):
            calls.append(("recompute",))
            return dict(b, adv=b["adv"] + 1.0)

        def _local_grad(self, rec, rows, global_batch, adv_stats, out):
            out.zero_()
            out[self.eng.P + 1] = global_batch            # the clip-loss part survives the loss fix-up

        def _apply(self, grad):
            self.eng.adam_step += 1

    eng = SimpleNamespace(params=torch.zeros(5), adam_step=0, P=5, device=torch.device("cpu"),
                          cfg=PPOConfig(recompute_advantage=True, advantage_normalization=False))
    b = {"obs": torch.zeros(10, 3), "adv": torch.zeros(10)}
    losses, steps = Probe(eng).update(b, 4, 3, [np.arange(10)] * 3)
    assert steps == 6 and eng.adam_step == 6 and losses.shape == (6, 4)
    assert calls == [("pack", 0.0), ("recompute",), ("pack", 1.0), ("recompute",), ("pack", 2.0)]
    assert losses[:, 1].tolist() == [4.0, 6.0] * 3            # global batch sizes of the two chunks (merge_last)
