"""PPO learn() path for a discrete-action MLP actor-critic on the MI355X engine -- BASELINE.json configs[0]
(CartPole shape: obs 4, MLP[64, 64], 2 actions).

Mirrors PPO._preprocess_batch / _update_with_batch (tianshou/algorithm/modelfree/ppo.py:146-224, a2c.py:115-153) for
the networks of test/discrete/test_ppo_discrete.py:88-98: Net(obs, [h, h]) ReLU shared by DiscreteActor and
DiscreteCritic (utils/net/discrete.py:27-123), Categorical policy.  `DiscreteActor(softmax_output=True)` with
`dist_fn=Categorical` (probabilities) and `softmax_output=False` with the logits dist_fn describe the same
distribution; the kernels work on the pre-softmax outputs.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .buffer import DeviceReplayBuffer, _i64_dev, gather_rows
from .ppo import PPOConfig, split_offsets
from .ppo_cnn import adv_stats_of, gae_and_return_scaling, run_minibatches

TRUNK_KEYS = ["preprocess.model.model.0.weight", "preprocess.model.model.0.bias",
              "preprocess.model.model.2.weight", "preprocess.model.model.2.bias"]
HEAD_KEYS = ["last.model.0.weight", "last.model.0.bias"]


def layout(obs_dim: int, hidden: int, n_act: int) -> dict[str, int]:
    out = (C.c_int64 * 3)()
    _lib.check(_lib.load().ts_mlp_ac_layout(_lib.i64(obs_dim), _lib.i64(hidden), _lib.i64(n_act), out))
    return dict(zip(["k0", "head", "count"], (int(v) for v in out)))


def flat_from_torch(t: list[torch.Tensor], obs_dim: int, hidden: int, n_act: int, device="cuda") -> torch.Tensor:
    """[l1.w, l1.b, l2.w, l2.b, actor.w, actor.b, critic.w, critic.b] in torch nn.Linear layout (also valid for the
    matching Adam moments) -> the engine's flat vector."""
    lay = layout(obs_dim, hidden, n_act)
    f = lambda x: x.detach().float().cpu()  # noqa: E731
    l1 = torch.zeros((lay["k0"] + 1, hidden), dtype=torch.float32)
    l1[:obs_dim], l1[lay["k0"]] = f(t[0]).t(), f(t[1])
    l2 = torch.cat([f(t[2]).t(), f(t[3])[None, :]])
    head = torch.zeros((hidden + 1, lay["head"]), dtype=torch.float32)
    head[:hidden, :n_act], head[hidden, :n_act] = f(t[4]).t(), f(t[5])
    head[:hidden, n_act], head[hidden, n_act] = f(t[6]).reshape(-1), f(t[7]).reshape(())
    return torch.cat([l1.reshape(-1), l2.reshape(-1), head.reshape(-1)]).to(device).contiguous()


def flat_to_torch(flat: torch.Tensor, obs_dim: int, hidden: int, n_act: int) -> list[torch.Tensor]:
    lay = layout(obs_dim, hidden, n_act)
    n1, n2 = (lay["k0"] + 1) * hidden, (hidden + 1) * hidden
    f = flat.detach()
    l1 = f[:n1].reshape(lay["k0"] + 1, hidden)
    l2 = f[n1:n1 + n2].reshape(hidden + 1, hidden)
    hd = f[n1 + n2:].reshape(hidden + 1, lay["head"])
    return [l1[:obs_dim].t().contiguous(), l1[lay["k0"]].clone(), l2[:hidden].t().contiguous(), l2[hidden].clone(),
            hd[:hidden, :n_act].t().contiguous(), hd[hidden, :n_act].clone(),
            hd[:hidden, n_act].reshape(1, hidden).clone(), hd[hidden, n_act].reshape(1).clone()]


class DiscretePPOEngine:
    """State of one PPO learner (shared-trunk MLP, Categorical policy) on one GPU."""

    def __init__(self, obs_dim: int, hidden: int, n_act: int, flat_params: torch.Tensor, cfg: PPOConfig):
        if not flat_params.is_cuda:
            raise RuntimeError("DiscretePPOEngine needs parameters on an MI355X (no CPU fallback)")
        if cfg.algo not in ("ppo", "a2c"):
            raise NotImplementedError("DiscretePPOEngine: PPO or A2C objective")
        self.obs_dim, self.hidden, self.n_act, self.cfg = obs_dim, hidden, n_act, cfg
        self.P = layout(obs_dim, hidden, n_act)["count"]
        if flat_params.numel() != self.P:
            raise ValueError(f"expected {self.P} parameters, got {flat_params.numel()}")
        self.device = flat_params.device
        self.params = flat_params.detach().float().contiguous().clone()
        self.adam_m, self.adam_v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        self.adam_step = 0
        self.ret_rms = [0.0, 1.0, 0.0]                        # RunningMeanStd: mean, var, count
        self._ws = _lib.default_workspace(self.device.index or 0)

    def _dims(self):
        return _lib.i64(self.obs_dim), _lib.i64(self.hidden), _lib.i64(self.n_act)

    def _obs(self, obs) -> torch.Tensor:
        obs = torch.as_tensor(obs, device=self.device).to(torch.float32).reshape(-1, self.obs_dim).contiguous()
        return obs

    def infer(self, obs, act=None, want_logits: bool = False):
        """-> (V float32[B], log_prob float32[B] or None[, logits float32[B, A]])."""
        obs = self._obs(obs)
        b = obs.shape[0]
        v = torch.empty(b, dtype=torch.float32, device=self.device)
        act = None if act is None else _i64_dev(act, self.device).reshape(-1)
        logp = torch.empty(b, dtype=torch.float32, device=self.device) if act is not None else None
        logits = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device) if want_logits else None
        _lib.check(_lib.load().ts_mlp_ac_infer(
            self._ws.handle, _lib.ptr(self.params), *self._dims(), _lib.ptr(obs), _lib.ptr(act), _lib.i64(b), _lib.ptr(v),
            _lib.ptr(logp), _lib.ptr(logits), _lib.current_stream(self.device)))
        return (v, logp, logits) if want_logits else (v, logp)

    # -- PPO._preprocess_batch -------------------------------------------------------------------------------
    def preprocess(self, buffer: DeviceReplayBuffer) -> dict:
        """Whole-buffer pass in sample_indices(0) order: V(s), V(s'), log pi_old(a|s), GAE, optional return scaling
        (a2c.py:115-153, ppo.py:146-162).  Needs buffer.obs and buffer.act; obs_next is the stored column or obs[next(index)] (buffer_base.py:622-626)."""
        if buffer.obs is None or buffer.act is None:
            raise ValueError("the device buffer must hold obs and act")
        idx = buffer.sample_indices(0)
        act_b = buffer.act[idx].reshape(-1)
        v_s, logp_old = self.infer(gather_rows(buffer.obs, idx), act_b)
        v_next, _ = self.infer(buffer.obs_next_rows(idx))
        out = gae_and_return_scaling(self, buffer, idx, v_s, v_next)
        return {"indices": idx, "act": act_b, "v_s": v_s, "returns": out["returns"], "adv": out["adv"],
                "logp_old": logp_old}

    def recompute(self, buffer: DeviceReplayBuffer, pre: dict) -> None:
        """recompute_advantage (ppo.py:174-178): V(s), V(s'), GAE and the return scaling (incl. another RunningMeanStd update,
        a2c.py:148) again with the current parameters; log pi_old stays.  Updates `pre` in place."""
        idx = pre["indices"]
        v_s, _ = self.infer(gather_rows(buffer.obs, idx))
        v_next, _ = self.infer(buffer.obs_next_rows(idx))
        out = gae_and_return_scaling(self, buffer, idx, v_s, v_next)
        pre["v_s"], pre["returns"], pre["adv"] = v_s, out["returns"], out["adv"]

    # -- one minibatch step ---------------------------------------------------------------------------------------
    def step(self, obs, act, adv, returns, logp_old=None, v_old=None, grad_out=None, apply: bool = True) -> torch.Tensor:
        """-> losses float32[4] = {loss, clip / actor, vf, ent} (device).  logp_old / v_old: PPO only."""
        obs = self._obs(obs)
        b = obs.shape[0]
        f32 = lambda t: None if t is None else torch.as_tensor(t, device=self.device).to(torch.float32).reshape(-1).contiguous()  # noqa: E731
        act, adv, returns, logp_old, v_old = _i64_dev(act, self.device).reshape(-1), f32(adv), f32(returns), f32(logp_old), f32(v_old)
        if self.cfg.algo == "ppo" and (logp_old is None or v_old is None):
            raise ValueError("the PPO objective needs logp_old and v_old")
        if any(t is not None and t.numel() != b for t in (act, adv, returns, logp_old, v_old)):
            raise ValueError("minibatch tensors differ in length")
        stats = adv_stats_of(self.cfg, adv)                  # on the device copy
        if apply:
            self.adam_step += 1
        hp = self.cfg.to_c()
        if not apply:
            hp.lr = -1.0
        losses = torch.empty(4, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().ts_mlp_ppo_step(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v),
            _lib.i64(max(self.adam_step, 1)), *self._dims(), _lib.ptr(obs), _lib.ptr(act), _lib.ptr(adv), _lib.ptr(returns),
            _lib.ptr(logp_old), _lib.ptr(v_old), _lib.i64(b), _lib.ptr(stats), C.byref(hp), _lib.ptr(losses),
            _lib.ptr(grad_out), _lib.current_stream(self.device)))
        return losses

    # -- PPO._update_with_batch ---------------------------------------------------------------------------------
    def update(self, buffer: DeviceReplayBuffer, pre: dict, batch_size: int | None, repeat: int, perms=None):
        """ppo.py:164-224.  `perms`: `repeat` permutations of range(N) (NumPy arrays for seed-exact parity with
        Batch.split, batch.py:1209, or int64 device tensors).  -> (losses float32[steps, 4], steps)."""
        obs_all = gather_rows(buffer.obs, pre["indices"])
        n = pre["indices"].numel()
        lib = _lib.load()
        # (the one-launch kernel keeps the Adam moments in registers and has torch.optim.Adam's step built in: RMSprop /
        # weight decay run on the per-step path, whose optimizer step is ts_optim's general kernel)
        if (lib.ts_mlp_ppo_update_supported(*self._dims()) and not os.environ.get("TS_MLP_PPO_PER_STEP")
                and self.cfg.plain_adam):
            # small network: the whole loop (every minibatch of every repeat, clip + Adam included) is ONE launch
            if perms is None:
                perms = [np.random.permutation(n) for _ in range(repeat)]
            offs = split_offsets(n, batch_size, merge_last=True)
            per = len(offs) - 1
            n_steps = repeat * per
            f32 = lambda t: None if t is None else t.to(torch.float32).reshape(-1).contiguous()  # noqa: E731
            losses = torch.empty((n_steps, 4), dtype=torch.float32, device=self.device)
            hp = self.cfg.to_c()
            obs_f, act_i = self._obs(obs_all), _i64_dev(pre["act"], self.device).reshape(-1).contiguous()
            # recompute_advantage: the batch changes between the repeats, so every repeat is its own launch
            groups = [[r] for r in range(repeat)] if self.cfg.recompute_advantage else [list(range(repeat))]
            for grp in groups:
                if grp[0] > 0 and self.cfg.recompute_advantage:
                    self.recompute(buffer, pre)                                       # ppo.py:175-176
                rows = torch.cat([_i64_dev(perms[r], self.device).reshape(-1) for r in grp]).contiguous()
                h_off = np.asarray([k * n + o for k in range(len(grp)) for o in offs[:-1]] + [len(grp) * n], dtype=np.int64)
                k0 = grp[0] * per
                # converted copies stay bound until the call has enqueued its kernels (a temporary's block could be handed
                # to the next conversion before the launch)
                adv_f, ret_f, lp_f, vs_f = f32(pre["adv"]), f32(pre["returns"]), f32(pre["logp_old"]), f32(pre["v_s"])
                _lib.check(lib.ts_mlp_ppo_update(
                    self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v), _lib.i64(self.adam_step),
                    *self._dims(), _lib.ptr(obs_f), _lib.ptr(act_i), _lib.ptr(adv_f), _lib.ptr(ret_f),
                    _lib.ptr(lp_f), _lib.ptr(vs_f), _lib.i64(n), _lib.ptr(rows),
                    h_off.ctypes.data_as(C.c_void_p), _lib.i64(len(grp) * per), C.byref(hp), _lib.ptr(losses[k0:k0 + len(grp) * per]),
                    _lib.current_stream(self.device)))
                self.adam_step += len(grp) * per
            return losses, n_steps

        def step_rows(rows):
            return self.step(obs_all[rows], pre["act"][rows], pre["adv"][rows], pre["returns"][rows],
                             pre["logp_old"][rows], pre["v_s"][rows])

        rec = (lambda: self.recompute(buffer, pre)) if self.cfg.recompute_advantage else None
        return run_minibatches(self.device, pre["indices"].numel(), batch_size, repeat, perms, step_rows, rec)
