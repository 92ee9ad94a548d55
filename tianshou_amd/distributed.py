"""Data-parallel PPO update: one process per GPU, replay buffer sharded by sub-buffer (env id),
RCCL all-reduce of the flat fp32 gradient per minibatch step.

The reference has no distributed path (its only multi-GPU mechanism is single-process
``nn.DataParallel``, tianshou/utils/net/common.py:473-515).  Sharding follows SURVEY 8e: episodes
never span sub-buffers (ReplayBufferManager offsets, manager.py:50; the end-flag cut at each
sub-buffer tail, algorithm_base.py:715), so sampling, index math, GAE and logp_old are shard-local
and need no exchange.  Per gradient step exactly one collective moves P + 4 floats (gradient +
loss parts); with advantage normalisation one more tiny collective per update() makes the
per-minibatch mean / std global (ppo.py:184-186 semantics over the global minibatch).

Semantics: global minibatch k = union over ranks of each rank's k-th local minibatch; the loss is
the mean over the global minibatch, so every rank scales its local sums by 1 / global_batch and
the all-reduce (sum) yields exactly the single-process gradient.  Clip + Adam then run
identically on every replica.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .ppo import PPOEngine, pack_batch, split_offsets


def shard_envs(n_env: int, rank: int, world: int) -> tuple[int, int]:
    """Sub-buffers [lo, hi) owned by `rank`: contiguous, sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, rem = divmod(n_env, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_adv_stats(adv: torch.Tensor, perm_rows: list[torch.Tensor], group=None) -> torch.Tensor:
    """{mean, unbiased std} of every GLOBAL minibatch -> float32 [n_chunks, 2] on adv's device.
    One all-reduce of 3 * n_chunks float64 values per repeat."""
    acc = torch.zeros((len(perm_rows), 3), dtype=torch.float64, device=adv.device)
    for k, rows in enumerate(perm_rows):
        a = adv[rows].double()
        acc[k, 0] = a.sum()
        acc[k, 1] = (a * a).sum()
        acc[k, 2] = a.numel()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(acc, group=group)
    n = acc[:, 2]
    mean = acc[:, 0] / n
    var = (acc[:, 1] - n * mean * mean) / (n - 1.0)       # torch.std(): unbiased
    return torch.stack([mean, var.clamp_min(0).sqrt()], dim=1).float().contiguous()


class DataParallelPPO:
    """PPO._update_with_batch (ppo.py:164-224) over `world` replicas of a PPOEngine."""

    def __init__(self, engine: PPOEngine, group=None, allreduce=None):
        """`allreduce` (optional): in-place sum over the ranks of a flat float32 device tensor, e.g.
        `tianshou_amd.collective.NativeAllReduce` (the C-ABI ts_allreduce); default torch.distributed (RCCL)."""
        self.eng = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._allreduce = allreduce
        self._buf = None

    def _exchange(self, buf):
        if self._allreduce is not None:
            self._allreduce(buf)
        else:
            dist.all_reduce(buf, group=self.group)

    # -- the two device steps around the collective (overridden by the CPU test double) --------
    def _local_grad(self, rec, rows, global_batch, adv_stats, out):
        """out[:P] = sum over local rows of d loss_i / global_batch; out[P:P+4] = loss parts."""
        eng, lib = self.eng, _lib.load()
        hp = eng.cfg.to_c()
        _lib.check(lib.ts_ppo_grad(
            eng._ws.handle, _lib.ptr(eng.params), _lib.i64(eng.obs_dim), _lib.i64(eng.act_dim),
            _lib.ptr(rec), _lib.i64(rec.shape[0]), _lib.ptr(rows), _lib.i64(rows.numel()),
            _lib.i64(global_batch), _lib.ptr(adv_stats), C.byref(hp), _lib.ptr(out),
            C.c_void_p(out.data_ptr() + 4 * eng.P), _lib.current_stream(eng.device)))

    def _apply(self, grad):
        eng, lib = self.eng, _lib.load()
        hp = eng.cfg.to_c()
        eng.adam_step += 1
        _lib.check(lib.ts_ppo_apply(
            eng._ws.handle, _lib.ptr(eng.params), _lib.ptr(eng.adam_m), _lib.ptr(eng.adam_v), _lib.i64(eng.adam_step),
            _lib.i64(eng.obs_dim), _lib.i64(eng.act_dim), _lib.ptr(grad), C.byref(hp),
            _lib.current_stream(eng.device)))

    def _native_comm(self):
        """The RCCL communicator handle for ts_ppo_dp_step: a NativeAllReduce's, NULL on one rank, or None when the
        exchange goes through torch.distributed / a test double (then the three calls stay separate)."""
        comm = getattr(self._allreduce, "_comm", None)
        if comm is not None:
            if not comm and self.world > 1:       # NativeAllReduce.close() was called: a NULL handle would skip the exchange
                raise RuntimeError("DataParallelPPO: the NativeAllReduce communicator has been closed")
            return comm
        if self.world == 1 and self._allreduce is None and type(self)._local_grad is DataParallelPPO._local_grad:
            return C.c_void_p()
        return None

    def _dp_step(self, comm, rec, rows, global_batch, adv_stats, out):
        """Gradient + exchange + optimizer step of one minibatch behind one C call (ts_ppo_dp_step)."""
        eng, lib = self.eng, _lib.load()
        hp = eng.cfg.to_c()
        eng.adam_step += 1
        _lib.check(lib.ts_ppo_dp_step(
            eng._ws.handle, comm, _lib.ptr(eng.params), _lib.ptr(eng.adam_m), _lib.ptr(eng.adam_v), _lib.i64(eng.adam_step),
            _lib.i64(eng.obs_dim), _lib.i64(eng.act_dim), _lib.ptr(rec), _lib.i64(rec.shape[0]), _lib.ptr(rows),
            _lib.i64(rows.numel()), _lib.i64(global_batch), _lib.ptr(adv_stats), C.byref(hp), _lib.ptr(out),
            _lib.current_stream(eng.device)))

    def _pack(self, b):
        return pack_batch(b, self.eng.obs_dim, self.eng.act_dim)

    def _begin_update(self):
        # the parameters may have been changed outside ts_ppo_apply since the last update (fused path,
        # load_state_dict): make ts_ppo_grad rebuild its cached weight images once
        _lib.check(_lib.load().ts_ppo_invalidate_image(self.eng._ws.handle))

    # -- preprocessing with global return statistics ----------------------------------------------
    def _reduce_stats(self, s1: float, s2: float, n: float):
        """(sum, sum of squares, count) of the unnormalised returns over ALL ranks: one 3-double all-reduce per
        preprocess, after which every replica performs the same `RunningMeanStd.update` (a2c.py:148)."""
        if self.world == 1:
            return s1, s2, n
        t = torch.tensor([s1, s2, n], dtype=torch.float64, device=self._coll_device())
        dist.all_reduce(t, group=self.group)
        s1, s2, n = (float(x) for x in t.tolist())
        return s1, s2, n

    def _coll_device(self):
        return self.eng.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")

    def preprocess(self, obs, obs_next, act, rew, terminated, truncated, cut_pos, d_n_cut=None):
        """PPO._preprocess_batch (ppo.py:146-162) on this rank's shard: values, GAE and logp_old are shard-local
        (episodes never span sub-buffers); only the three moments that feed `ret_rms` are exchanged."""
        return self.eng.preprocess(obs, obs_next, act, rew, terminated, truncated, cut_pos, d_n_cut,
                                   reduce_stats=self._reduce_stats)

    def _recompute(self, b: dict) -> dict:
        """ppo.py:174-178 (recompute_advantage) on the shard, again with global return statistics."""
        v_s, returns, adv = self.eng.add_returns_and_advantages(
            b["obs"], b["obs_next"], b["rew"], b["terminated"], b["truncated"], b["cut_pos"], b.get("d_n_cut"),
            reduce_stats=self._reduce_stats)
        return dict(b, v_s=v_s, returns=returns, adv=adv)

    # -- minibatch line-up across ranks -------------------------------------------------------------
    def _line_up(self, n_local: int, batch_size: int | None, dev) -> list[int]:
        """Chunk boundaries of this rank's permutation.  Every rank must run the same number of gradient steps (one
        all-reduce each), but `shard_envs` may give shards that differ by one sub-buffer: the boundaries of
        Batch.split (batch.py:1205-1215) are taken on the LARGEST shard and scaled to the local size, so that the k-th
        local minibatches line up and stay proportional.  Equal shards get exactly Batch.split's boundaries.
        One int64 MAX all-reduce per update()."""
        n_ref = n_local
        if self.world > 1:
            t = torch.tensor([n_local], dtype=torch.int64, device=self._coll_device())
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            n_ref = int(t.item())
        # Whether the split is feasible is decided COLLECTIVELY: a rank that raised on its own would leave the others
        # waiting in the next all-reduce.  One MIN all-reduce of an ok flag, then every rank raises the same error.
        problem, offs = None, None
        if n_local < 1:
            problem = "every rank needs at least one transition"
        else:
            ref = split_offsets(n_ref, batch_size, merge_last=True)
            if n_ref == n_local:
                offs = ref
            else:
                offs = [min(n_local, (o * n_local + n_ref // 2) // n_ref) for o in ref]
                offs[0], offs[-1] = 0, n_local
                if any(b <= a for a, b in zip(offs[:-1], offs[1:])):
                    problem = "a shard is too small for the requested number of minibatches"
        if self.world > 1:
            ok = torch.tensor([0 if problem else 1], dtype=torch.int64, device=self._coll_device())
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if int(ok.item()) == 0 and problem is None:
                problem = "another rank cannot split its shard (too few transitions for the requested minibatches)"
        if problem:
            raise ValueError(problem)
        return offs

    def _global_counts(self, counts: list[int], dev) -> list[int]:
        """Rows of every GLOBAL minibatch = sum over ranks of the local chunk sizes (one all-reduce per update())."""
        if self.world == 1:
            return counts
        t = torch.tensor(counts, dtype=torch.int64, device=self._coll_device())
        dist.all_reduce(t, group=self.group)
        return [int(x) for x in t.tolist()]

    # -- update loop ----------------------------------------------------------------------------
    def update(self, b: dict, batch_size: int | None, repeat: int, perms):
        """Every rank passes its LOCAL batch `b` and LOCAL permutations.  Returns (losses [steps, 4] with global loss
        values, steps)."""
        eng, cfg = self.eng, self.eng.cfg
        n = b["obs"].shape[0]
        dev = b["obs"].device
        offs = self._line_up(n, batch_size, dev)
        chunks = list(zip(offs[:-1], offs[1:]))
        g_count = self._global_counts([hi - lo for lo, hi in chunks], dev)
        self._begin_update()
        perm_t = [p.to(device=dev, dtype=torch.int64) if isinstance(p, torch.Tensor)
                  else torch.as_tensor(np.asarray(p, dtype=np.int64), device=dev) for p in perms]
        n_steps = repeat * len(chunks)
        # one [P + 4] row per gradient step: no copy of the loss parts, no reuse hazard between steps
        width = (eng.P + 4 + 3) // 4 * 4                      # 16-byte aligned rows
        if self._buf is None or self._buf.device != dev or self._buf.shape[0] < n_steps:
            self._buf = torch.empty((n_steps, width), dtype=torch.float32, device=dev)
        comm = self._native_comm()
        k = 0
        for r in range(repeat):
            if cfg.recompute_advantage and r > 0:                  # ppo.py:174-178
                b = self._recompute(b)
            rec = self._pack(b)
            rows_r = [perm_t[r][lo:hi] for lo, hi in chunks]
            stats = global_adv_stats(b["adv"], rows_r, self.group) if cfg.advantage_normalization else None
            for c, rows in enumerate(rows_r):
                out = self._buf[k, : eng.P + 4]
                st = None if stats is None else stats[c]
                if comm is not None:
                    self._dp_step(comm, rec, rows, g_count[c], st, out)
                else:
                    self._local_grad(rec, rows, g_count[c], st, out)
                    if self.world > 1:
                        self._exchange(out)                      # RCCL: gradient + loss parts in one call
                    self._apply(out)
                k += 1
        res = self._buf[:n_steps, eng.P:eng.P + 4].clone()
        # the entropy term depends on the parameters only: every rank added the same value (a conditioned-sigma actor's
        # entropy is a per-sample quantity: its parts are sums over the local rows / global batch like the other two)
        if not getattr(eng, "entropy_is_batch_sum", False):
            res[:, 3] /= self.world
        res[:, 0] = res[:, 1] + cfg.vf_coef * res[:, 2] - cfg.ent_coef * res[:, 3]
        return res, n_steps


class DataParallelWidePPO(DataParallelPPO):
    """The same data-parallel update for `tianshou_amd.ppo_wide.WidePPOEngine` (Net[h, h] beyond the fused 64 x 64 kernels,
    e.g. Humanoid's 376 / 17 / 256 x 256): the local step is ts_ppo_wide_step in gradient-only mode with the global batch
    size (its loss sums and gradients are already scaled by 1 / global_batch), the exchange moves [gradient | 4 loss
    parts], clip + Adam run on the summed gradient (ts_adam_step: joint norm over actor + critic, a2c.py:103-107)."""

    def _pack(self, b):
        return b

    def _begin_update(self):
        pass

    def _native_comm(self):
        return None                                   # three calls per step: gradient, exchange, apply

    def _local_grad(self, b, rows, global_batch, adv_stats, out):
        eng = self.eng
        eng.step(b, rows, out[eng.P:eng.P + 4], grad_out=out[:eng.P], apply=False, global_batch=global_batch,
                 adv_stats=adv_stats)

    def _apply(self, grad):
        eng, cfg = self.eng, self.eng.cfg
        eng.adam_step += 1
        _lib.check(_lib.load().ts_adam_step(
            eng._ws.handle, _lib.ptr(eng.params), _lib.ptr(eng.adam_m), _lib.ptr(eng.adam_v), _lib.ptr(grad), _lib.i64(eng.P),
            _lib.i64(eng.adam_step), _lib.f64(cfg.lr), _lib.f64(cfg.betas[0]), _lib.f64(cfg.betas[1]), _lib.f64(cfg.adam_eps),
            _lib.f64(cfg.max_grad_norm or 0.0), _lib.current_stream(eng.device)))


class DataParallelDQN:
    """DQN._update_with_batch (dqn.py:381-404) over `world` replicas of a DQNEngine.

    Every rank samples from its own shard of the replay buffer (sub-buffers never share episodes),
    computes the gradient of the mean loss over its local minibatch, and one RCCL all-reduce (sum) of the flat
    fp32 gradient + the loss scalar (P + 1 floats, 6.75 MB at C3) followed by * 1/world gives the gradient of the
    mean over the global minibatch (equal local batch sizes).  Clip + Adam and the periodic target sync then run
    identically on every replica; PER priorities (td errors) stay shard-local."""

    def __init__(self, engine, group=None, allreduce=None):
        self.eng = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._allreduce = allreduce
        self._buf = None

    def _exchange(self, buf):
        if self._allreduce is not None:
            self._allreduce(buf)
        else:
            dist.all_reduce(buf, group=self.group)

    # -- the two device steps around the collective (overridden by the CPU test double) ------------------
    def _local_grad(self, obs, act, returns, weight, out):
        loss, td = self.eng.gradient(obs, act, returns, weight, out[: self.eng.P])
        out[self.eng.P:] = loss
        return td

    def _apply(self, grad):
        self.eng.apply_gradient(grad)

    def update_with_batch(self, obs, act, returns, weight=None):
        """-> (global mean loss [1], local td errors [B_local])."""
        eng = self.eng
        dev = eng.device
        if self._buf is None or self._buf.device != dev:
            self._buf = torch.empty(eng.P + 1, dtype=torch.float32, device=dev)
        out = self._buf
        td = self._local_grad(obs, act, returns, weight, out)
        if self.world > 1:
            self._exchange(out)
            out.mul_(1.0 / self.world)
        self._apply(out[: eng.P])
        return out[eng.P:].clone(), td


class DataParallelSAC:
    """SAC._update_with_batch (sac.py:298-336) over `world` replicas of a SACEngine.

    Every rank samples its own minibatch from its shard of the replay buffer (equal local batch sizes) and computes
    its n-step targets locally (the lagged critics and the actor are replicated).  Per update two collectives:
    [critic1 | critic2 | loss1 | loss2] after the critics' backward passes and [actor | -mean(log_prob) | actor loss]
    after the actor's -- the sums * 1/world are the gradients of the mean losses over the global minibatch, so the
    three Adam steps, the alpha step (sac.py:203-209 needs only the mean log-probability) and the Polyak update run
    identically on every replica.  PER weights (td errors) stay shard-local."""

    def __init__(self, engine, group=None, allreduce=None):
        self.eng = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._allreduce = allreduce
        self._bufs = None

    # -- device steps around the collectives (overridden by the CPU test double) -------------------------------
    def _begin(self, obs, act, returns, noise, weight):
        return self.eng.begin_phased_update(obs, act, returns, noise, weight)

    def _phase(self, ctx, phase, buf):
        self.eng.update_phase(ctx, phase, buf)

    def _sizes(self):
        return self.eng.exchange_floats()

    def _reduce(self, buf):
        if self.world > 1:
            if self._allreduce is not None:
                self._allreduce(buf)
            else:
                dist.all_reduce(buf, group=self.group)
            buf.mul_(1.0 / self.world)

    def update_with_batch(self, obs, act, returns, noise, weight=None):
        """-> (stats float32[5] with GLOBAL losses, local new PER weights float32[B_local])."""
        eng = self.eng
        n_c, n_a = self._sizes()
        if self._bufs is None:
            dev = eng.device
            self._bufs = (torch.empty(n_c + 2, dtype=torch.float32, device=dev),
                          torch.empty(n_a + 1, dtype=torch.float32, device=dev))
        buf_c, buf_a = self._bufs
        ctx = self._begin(obs, act, returns, noise, weight)
        stats = ctx["stats"]
        self._phase(ctx, eng.PHASE_CRITIC_GRAD, buf_c)
        buf_c[n_c:] = stats[1:3]                               # the two critic losses ride along
        self._reduce(buf_c)
        self._phase(ctx, eng.PHASE_CRITIC_APPLY, buf_c)
        self._phase(ctx, eng.PHASE_ACTOR_GRAD, buf_a)
        buf_a[n_a:] = stats[0:1]
        self._reduce(buf_a)
        self._phase(ctx, eng.PHASE_ACTOR_APPLY, buf_a)
        out = stats.clone()
        out[1:3] = buf_c[n_c:]
        out[0] = buf_a[n_a]
        return out, ctx["w_out"]
