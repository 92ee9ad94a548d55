"""PPO learn() path for the Atari actor-critic on the MI355X engine.

Mirrors PPO._preprocess_batch / _update_with_batch (tianshou/algorithm/modelfree/ppo.py:146-224, a2c.py:115-153)
for the networks of examples/atari/atari_ppo.py:106-118: DQNet(features_only=True, output_dim_added_layer=512)
shared by DiscreteActor(softmax_output=False) and DiscreteCritic, Categorical policy
(utils/net/discrete.py:20-123).  Frames stay uint8 in HBM; every use gathers (and frame-stacks) them straight
into the float32 NHWC batch the conv kernels consume.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .buffer import DeviceReplayBuffer, _i64_dev
from .dqn import _u8_flag, gather_obs_nhwc
from .ppo import PPOConfig, split_offsets
from .returns import cut_positions, gae_scan

TRUNK_KEYS = ["preprocess.net.0.0.weight", "preprocess.net.0.0.bias", "preprocess.net.0.2.weight",
              "preprocess.net.0.2.bias", "preprocess.net.0.4.weight", "preprocess.net.0.4.bias",
              "preprocess.net.1.weight", "preprocess.net.1.bias"]
HEAD_KEYS = ["last.model.0.weight", "last.model.0.bias"]
HEAD = 32


def layer_layout(c: int, h: int, w: int, n_act: int):
    off = (C.c_int64 * 7)()
    geom = (C.c_int64 * 50)()
    _lib.check(_lib.load().ts_cnn_ac_layer_offsets(_lib.i64(c), _lib.i64(h), _lib.i64(w), _lib.i64(n_act), off, geom))
    return np.array(off[:6], np.int64), np.array(geom[:], np.int64).reshape(5, 10)


def flat_from_torch(t: list[torch.Tensor], c: int, h: int, w: int, n_act: int, device="cuda") -> torch.Tensor:
    """[conv1.w, conv1.b, conv2.w, conv2.b, conv3.w, conv3.b, fc.w, fc.b, actor.w, actor.b, critic.w, critic.b]
    in torch layout (also valid for the matching Adam moments) -> the engine's flat vector."""
    _, geom = layer_layout(c, h, w, n_act)
    oh3, ow3 = int(geom[2, 7]), int(geom[2, 8])
    f = lambda x: x.detach().float().cpu()  # noqa: E731
    parts = []
    for i in range(3):
        parts += [f(t[2 * i]).permute(2, 3, 1, 0).reshape(-1), f(t[2 * i + 1]).reshape(-1)]
    parts += [f(t[6]).reshape(512, 64, oh3, ow3).permute(2, 3, 1, 0).reshape(-1), f(t[7]).reshape(-1)]
    head = torch.zeros((513, HEAD), dtype=torch.float32)
    head[:512, :n_act], head[512, :n_act] = f(t[8]).t(), f(t[9])
    head[:512, n_act], head[512, n_act] = f(t[10]).reshape(-1), f(t[11]).reshape(())
    parts.append(head.reshape(-1))
    return torch.cat(parts).to(device).contiguous()


def flat_to_torch(flat: torch.Tensor, c: int, h: int, w: int, n_act: int) -> list[torch.Tensor]:
    off, geom = layer_layout(c, h, w, n_act)
    out = []
    for i in range(3):
        ic, kh, kw, oc = (int(geom[i, j]) for j in (3, 4, 5, 9))
        k = kh * kw * ic
        wb = flat[off[i]:off[i + 1]].reshape(k + 1, oc)
        out += [wb[:k].reshape(kh, kw, ic, oc).permute(3, 2, 0, 1).contiguous(), wb[k].clone()]
    oh3, ow3 = int(geom[2, 7]), int(geom[2, 8])
    fdim = 64 * oh3 * ow3
    wb = flat[off[3]:off[4]].reshape(fdim + 1, 512)
    out += [wb[:fdim].reshape(oh3, ow3, 64, 512).permute(3, 2, 0, 1).reshape(512, fdim).contiguous(), wb[fdim].clone()]
    hd = flat[off[4]:off[5]].reshape(513, HEAD)
    out += [hd[:512, :n_act].t().contiguous(), hd[512, :n_act].clone(),
            hd[:512, n_act].reshape(1, 512).clone(), hd[512, n_act].reshape(1).clone()]
    return out


def gae_and_return_scaling(engine, buffer: DeviceReplayBuffer, idx: torch.Tensor, v_s, v_next) -> dict:
    """GAE over the whole batch + optional return scaling / RunningMeanStd update (a2c.py:131-153,
    statistics.py:99-114) for an engine with `.cfg` and `.ret_rms`.  -> {"returns", "adv"}."""
    cfg, n = engine.cfg, idx.numel()
    cut_pos, d_n_cut = cut_positions(buffer, idx)
    scale = math.sqrt(engine.ret_rms[1] + 1e-8) if cfg.return_scaling else 1.0
    out = gae_scan(v_s, v_next, buffer.rew[idx], buffer.terminated[idx], buffer.truncated[idx], cut_pos,
                   gamma=cfg.gamma, gae_lambda=cfg.gae_lambda, v_scale=scale, ret_div=scale,
                   want_ret_stats=cfg.return_scaling, d_n_cut=d_n_cut)
    if cfg.return_scaling:
        s1, s2 = float(out["ret_sum"]), float(out["ret_sumsq"])
        b_mean = s1 / n
        b_var = max(s2 / n - b_mean * b_mean, 0.0)
        mean, var, count = engine.ret_rms
        delta, tot = b_mean - mean, count + n
        engine.ret_rms = [mean + delta * n / tot, (var * count + b_var * n + delta * delta * count * n / tot) / tot, tot]
    return out


def adv_stats_of(cfg: PPOConfig, adv: torch.Tensor):
    """{mean, unbiased std} of a minibatch's advantages (ppo.py:184-186), or None."""
    if not cfg.advantage_normalization:
        return None
    a64 = adv.double()
    return torch.stack([a64.mean(), a64.std()]).float().contiguous()


def run_minibatches(device, n: int, batch_size: int | None, repeat: int, perms, step_rows, recompute=None):
    """The loop of PPO._update_with_batch (ppo.py:174-178): `repeat` passes over Batch.split(batch_size,
    merge_last=True); step_rows(rows int64 device) -> losses[4]; `recompute()` (recompute_advantage, ppo.py:175-176) runs
    before every repeat after the first.  -> (losses float32[steps, 4], steps)."""
    if perms is None:
        perms = [np.random.permutation(n) for _ in range(repeat)]
    offs = split_offsets(n, batch_size, merge_last=True)
    out = []
    for r in range(repeat):
        if recompute is not None and r > 0:
            recompute()
        perm = _i64_dev(perms[r], device)
        for lo, hi in zip(offs[:-1], offs[1:]):
            out.append(step_rows(perm[lo:hi]))
    return torch.stack(out), len(out)


class CnnPPOEngine:
    """State of one PPO learner (Atari actor-critic) on one GPU."""

    def __init__(self, c: int, h: int, w: int, n_act: int, flat_params: torch.Tensor, cfg: PPOConfig):
        if not flat_params.is_cuda:
            raise RuntimeError("CnnPPOEngine needs parameters on an MI355X (no CPU fallback)")
        if cfg.algo not in ("ppo", "a2c"):
            raise NotImplementedError("CnnPPOEngine: PPO or A2C objective")
        off, _ = layer_layout(c, h, w, n_act)
        self.c, self.h, self.w, self.n_act, self.cfg = c, h, w, n_act, cfg
        self.P = int(off[5])
        if flat_params.numel() != self.P:
            raise ValueError(f"expected {self.P} parameters, got {flat_params.numel()}")
        self.device = flat_params.device
        self.params = flat_params.detach().float().contiguous().clone()
        self.adam_m, self.adam_v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        self.adam_step = 0
        self.ret_rms = [0.0, 1.0, 0.0]                        # RunningMeanStd: mean, var, count
        self._ws = _lib.default_workspace(self.device.index or 0)

    def infer(self, obs_nhwc: torch.Tensor, act=None, want_logits: bool = False):
        """-> (V float32[B], log_prob float32[B] or None[, logits float32[B, A]])."""
        b = obs_nhwc.shape[0]
        obs_nhwc = obs_nhwc.contiguous()
        v = torch.empty(b, dtype=torch.float32, device=self.device)
        act = None if act is None else _i64_dev(act, self.device).reshape(-1)
        logp = torch.empty(b, dtype=torch.float32, device=self.device) if act is not None else None
        logits = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device) if want_logits else None
        _lib.check(_lib.load().ts_cnn_ac_infer(
            self._ws.handle, _lib.ptr(self.params), _lib.i64(self.c), _lib.i64(self.h), _lib.i64(self.w),
            _lib.i64(self.n_act), _lib.ptr(obs_nhwc), _u8_flag(obs_nhwc), _lib.ptr(act), _lib.i64(b), _lib.ptr(v),
            _lib.ptr(logp), _lib.ptr(logits), _lib.current_stream(self.device)))
        return (v, logp, logits) if want_logits else (v, logp)

    # -- PPO._preprocess_batch -------------------------------------------------------------------------------
    def _values(self, buffer: DeviceReplayBuffer, frames: torch.Tensor, idx: torch.Tensor, act_b, stack_num: int,
                obs_next_frames: torch.Tensor | None, chunk: int):
        """V(s), V(s') (and log pi(a|s) when act_b is given) of the transitions idx, one trunk pass per observation."""
        n = idx.numel()
        v_s = torch.empty(n, dtype=torch.float32, device=self.device)
        v_next = torch.empty_like(v_s)
        logp = torch.empty_like(v_s) if act_b is not None else None
        for lo in range(0, n, chunk):
            sl = slice(lo, min(lo + chunk, n))
            v, lp = self.infer(gather_obs_nhwc(frames, buffer, idx[sl], stack_num, as_u8=True), None if act_b is None else act_b[sl])
            v_s[sl] = v
            if lp is not None:
                logp[sl] = lp
            if obs_next_frames is not None:
                v_next[sl] = self.infer(gather_obs_nhwc(obs_next_frames, buffer, idx[sl], stack_num, as_u8=True))[0]
        if obs_next_frames is None:
            # A buffer that does not store obs_next reads it as the (stacked) observation at next(index)
            # (buffer_base.py:624-626) -- and next(index) is itself one of the sampled indices (sample_indices(0) yields every
            # valid slot; next() stays inside the filled part of its sub-buffer).  So V(s') of transition i IS V(s) of
            # transition next(i): the same network on the same input rows.  The reference evaluates the critic a second time
            # on those rows (a2c.py:126-128); a row's value does not depend on which batch it sits in, so gathering it is
            # bit-identical and saves the whole second pass over the rollout.
            pos = torch.empty(buffer.maxsize, dtype=torch.int64, device=self.device)
            pos[idx] = torch.arange(n, dtype=torch.int64, device=self.device)
            v_next = v_s[pos[buffer.next(idx)]]
        return v_s, v_next, logp

    def preprocess(self, buffer: DeviceReplayBuffer, frames: torch.Tensor, act: torch.Tensor, stack_num: int,
                   obs_next_frames: torch.Tensor | None = None, chunk: int = 65536) -> dict:
        """Whole-buffer pass in sample_indices(0) order: V(s), V(s'), log pi_old(a|s) (one trunk pass per
        observation), GAE, optional return scaling (a2c.py:115-153, ppo.py:146-162)."""
        idx = buffer.sample_indices(0)
        act_b = act[idx]
        v_s, v_next, logp_old = self._values(buffer, frames, idx, act_b, stack_num, obs_next_frames, chunk)
        out = gae_and_return_scaling(self, buffer, idx, v_s, v_next)
        return {"indices": idx, "act": act_b, "v_s": v_s, "returns": out["returns"], "adv": out["adv"],
                "logp_old": logp_old, "obs_next_frames": obs_next_frames, "chunk": chunk}

    def recompute(self, buffer: DeviceReplayBuffer, frames: torch.Tensor, pre: dict, stack_num: int) -> None:
        """recompute_advantage (ppo.py:174-178): `_add_returns_and_advantages` again with the current parameters -- V(s),
        V(s'), GAE, return scaling incl. another RunningMeanStd update (a2c.py:148); log pi_old stays.  Updates `pre`."""
        v_s, v_next, _ = self._values(buffer, frames, pre["indices"], None, stack_num, pre.get("obs_next_frames"),
                                      pre.get("chunk", 65536))
        out = gae_and_return_scaling(self, buffer, pre["indices"], v_s, v_next)
        pre["v_s"], pre["returns"], pre["adv"] = v_s, out["returns"], out["adv"]

    # -- one minibatch step ---------------------------------------------------------------------------------------
    def step(self, obs_nhwc, act, adv, returns, logp_old=None, v_old=None, grad_out=None, apply: bool = True) -> torch.Tensor:
        """-> losses float32[4] = {loss, clip / actor, vf, ent} (device).  logp_old / v_old: PPO only."""
        cfg = self.cfg
        b = obs_nhwc.shape[0]
        adv = torch.as_tensor(adv, device=self.device)
        stats = adv_stats_of(cfg, adv)
        if apply:
            self.adam_step += 1
        hp = cfg.to_c()
        if not apply:
            hp.lr = -1.0
        losses = torch.empty(4, dtype=torch.float32, device=self.device)
        f32 = lambda t: None if t is None else t.to(torch.float32).contiguous()  # noqa: E731
        obs_nhwc = obs_nhwc.contiguous()
        _lib.check(_lib.load().ts_cnn_ppo_step(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v),
            _lib.i64(max(self.adam_step, 1)), _lib.i64(self.c), _lib.i64(self.h), _lib.i64(self.w),
            _lib.i64(self.n_act), _lib.ptr(obs_nhwc), _u8_flag(obs_nhwc), _lib.ptr(_i64_dev(act, self.device).reshape(-1)),
            _lib.ptr(f32(adv)), _lib.ptr(f32(returns)), _lib.ptr(f32(logp_old)), _lib.ptr(f32(v_old)), _lib.i64(b),
            _lib.ptr(stats), C.byref(hp), _lib.ptr(losses), _lib.ptr(grad_out), _lib.current_stream(self.device)))
        return losses

    # -- PPO._update_with_batch ---------------------------------------------------------------------------------
    def update(self, buffer: DeviceReplayBuffer, frames: torch.Tensor, pre: dict, stack_num: int,
               batch_size: int | None, repeat: int, perms=None):
        """ppo.py:164-224.  `perms`: `repeat` permutations of range(N) (NumPy arrays for seed-exact parity with
        Batch.split, batch.py:1209, or int64 device tensors).  -> (losses float32[steps, 4], steps)."""
        def step_rows(rows):
            obs = gather_obs_nhwc(frames, buffer, pre["indices"][rows], stack_num, as_u8=True)
            return self.step(obs, pre["act"][rows], pre["adv"][rows], pre["returns"][rows], pre["logp_old"][rows],
                             pre["v_s"][rows])

        rec = (lambda: self.recompute(buffer, frames, pre, stack_num)) if self.cfg.recompute_advantage else None
        return run_minibatches(self.device, pre["indices"].numel(), batch_size, repeat, perms, step_rows, rec)
