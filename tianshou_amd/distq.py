"""Distributional Q-learning (QRDQN, C51) learn() path on the MI355X engine.

Mirrors, on device tensors:
    QRDQNet.forward / C51Net.forward      tianshou/env/atari/atari_network.py:227-235 / :141-151
    QRDQNPolicy / C51Policy.compute_q_value   tianshou/algorithm/modelfree/qrdqn.py:19-21, c51.py:66-67
    QRDQN._target_q / C51._target_q       qrdqn.py:93-104 / c51.py:120-121 (n-step via tianshou_amd.returns)
    C51._target_dist                      c51.py:123-141
    QRDQN / C51._update_with_batch        qrdqn.py:106-131 / c51.py:143-160 (+ periodic hard sync dqn.py:277-285)
There is no CPU path: every function calls libtsengine.so and raises when it is missing.

Parameter layout: the DQN engine's (tianshou_amd.dqn.flat_from_torch; the reference nets are DQNet with
n_act * n_atoms outputs, head column a * n_atoms + j) with the head matrix zero-padded to a multiple of 32 columns:
`flat_from_torch` / `flat_to_torch` below.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib
from .buffer import DeviceReplayBuffer, _i64_dev
from .dqn import _u8_flag, gather_obs_nhwc, gather_obs_pair
from .dqn import flat_from_torch as flat_from_torch_dqn
from .dqn import flat_to_torch as flat_to_torch_dqn
from .lagged import full_parameter_update
from .returns import compute_nstep_return, nstep_indices, nstep_return_from_target_q

QR, C51 = "qr", "c51"
_KIND = {QR: 0, C51: 1}          # TS_DISTQ_QR / TS_DISTQ_C51


class DistQHParams(C.Structure):
    """struct ts_distq_hparams (include/tsengine.h)."""

    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("adam_eps", C.c_double),
                ("max_grad_norm", C.c_double), ("v_min", C.c_double), ("v_max", C.c_double)]


def param_count(c: int, h: int, w: int, n_act: int, n_atoms: int) -> int:
    lib = _lib.load()
    lib.ts_distq_param_count.restype = C.c_int64
    lib.ts_distq_param_count.argtypes = [C.c_int64] * 5
    n = int(lib.ts_distq_param_count(c, h, w, n_act, n_atoms))
    if n < 0:
        raise ValueError("unsupported network dimensions (n_act <= 64, 2 <= n_atoms <= 256)")
    return n


def head_width(n_act: int, n_atoms: int) -> int:
    return (n_act * n_atoms + 31) // 32 * 32


def flat_from_torch(tensors: list[torch.Tensor], c: int, h: int, w: int, n_act: int, n_atoms: int,
                    device="cuda") -> torch.Tensor:
    """QRDQNet / C51Net state_dict tensors (DQNet order; also valid for Adam moments) -> the engine's flat vector."""
    n_out, wd = n_act * n_atoms, head_width(n_act, n_atoms)
    flat = flat_from_torch_dqn(tensors, c, h, w, n_out, device="cpu")
    head = torch.zeros((513, wd), dtype=torch.float32)
    head[:, :n_out] = flat[-513 * n_out:].reshape(513, n_out)
    return torch.cat([flat[:-513 * n_out], head.reshape(-1)]).to(device).contiguous()


def flat_to_torch(flat: torch.Tensor, c: int, h: int, w: int, n_act: int, n_atoms: int) -> list[torch.Tensor]:
    """Inverse of flat_from_torch -> ten tensors in torch layout (on flat's device)."""
    n_out, wd = n_act * n_atoms, head_width(n_act, n_atoms)
    trunk = flat.numel() - 513 * wd
    head = flat[trunk:].reshape(513, wd)[:, :n_out]
    return flat_to_torch_dqn(torch.cat([flat[:trunk], head.reshape(-1)]), c, h, w, n_out)


@dataclass
class DistQConfig:
    """Hyper-parameters of the reference QRDQN (qrdqn.py:33-92) / C51 (c51.py:17-118) + Adam (optim.py:89-110)."""

    kind: str = QR
    n_atoms: int = 200            # num_quantiles / num_atoms
    v_min: float = -10.0          # C51 only
    v_max: float = 10.0
    gamma: float = 0.99
    n_step: int = 1
    target_update_freq: int = 0
    lr: float = 1e-3
    betas: tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8
    max_grad_norm: float | None = None

    def to_c(self, grad_only: bool = False) -> DistQHParams:
        return DistQHParams(-1.0 if grad_only else self.lr, self.betas[0], self.betas[1], self.adam_eps,
                            self.max_grad_norm or 0.0, self.v_min, self.v_max)


class DistQEngine:
    """State of one QRDQN / C51 learner on one GPU: flat parameters, lagged copy, Adam moments, counters."""

    def __init__(self, c: int, h: int, w: int, n_act: int, flat_params: torch.Tensor, cfg: DistQConfig):
        if not flat_params.is_cuda:
            raise RuntimeError("DistQEngine needs parameters on an MI355X (no CPU fallback)")
        if cfg.kind not in _KIND:
            raise ValueError("kind must be 'qr' or 'c51'")
        self.c, self.h, self.w, self.n_act, self.cfg = c, h, w, n_act, cfg
        self.P = param_count(c, h, w, n_act, cfg.n_atoms)
        if flat_params.numel() != self.P:
            raise ValueError(f"expected {self.P} parameters, got {flat_params.numel()}")
        self.device = flat_params.device
        self.params = flat_params.detach().float().contiguous().clone()
        self.params_old = self.params.clone() if cfg.target_update_freq > 0 else None    # dqn.py:240-246
        self.adam_m = torch.zeros_like(self.params)
        self.adam_v = torch.zeros_like(self.params)
        self.adam_step = 0
        self.iter = 0
        if cfg.kind == QR:                                   # tau_hat, qrdqn.py:87-91 (evaluated by torch, as there)
            tau = torch.linspace(0, 1, cfg.n_atoms + 1)
            aux = (tau[:-1] + tau[1:]) / 2
        else:                                                # support, c51.py:61-64
            aux = torch.linspace(cfg.v_min, cfg.v_max, cfg.n_atoms)
        self.aux = aux.to(self.device).contiguous()
        self._ws = _lib.default_workspace(self.device.index or 0)

    def _dims(self):
        return (_lib.i64(self.c), _lib.i64(self.h), _lib.i64(self.w), _lib.i64(self.n_act), _lib.i64(self.cfg.n_atoms),
                C.c_int(_KIND[self.cfg.kind]), _lib.ptr(self.aux))

    def _check_obs(self, obs: torch.Tensor) -> torch.Tensor:
        if tuple(obs.shape[1:]) != (self.h, self.w, self.c) or obs.dtype not in (torch.float32, torch.uint8):
            raise ValueError(f"obs must be float32 or uint8 [B, {self.h}, {self.w}, {self.c}] (NHWC)")
        return obs.contiguous()

    # -- policy forward -------------------------------------------------------------------------------
    def forward(self, obs_nhwc: torch.Tensor, params: torch.Tensor | None = None, want_dist: bool = True):
        """-> (dist float32[B, A, N] or None, q float32[B, A], act int64[B] = argmax_a q)."""
        obs_nhwc = self._check_obs(obs_nhwc)
        b = obs_nhwc.shape[0]
        dist = torch.empty((b, self.n_act, self.cfg.n_atoms), dtype=torch.float32, device=self.device) if want_dist else None
        q = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device)
        act = torch.empty(b, dtype=torch.int64, device=self.device)
        p = self.params if params is None else params
        _lib.check(_lib.load().ts_distq_forward(
            self._ws.handle, _lib.ptr(p), *self._dims(), _lib.ptr(obs_nhwc), _u8_flag(obs_nhwc), _lib.i64(b),
            _lib.ptr(dist), _lib.ptr(q), _lib.ptr(act), _lib.current_stream(self.device)))
        return dist, q, act

    def next_dist(self, obs_next_nhwc: torch.Tensor) -> torch.Tensor:
        """The lagged net's distribution of the online net's greedy action -> float32[B, N]."""
        obs_next_nhwc = self._check_obs(obs_next_nhwc)
        b = obs_next_nhwc.shape[0]
        out = torch.empty((b, self.cfg.n_atoms), dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().ts_distq_next_dist(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.params_old), *self._dims(), _lib.ptr(obs_next_nhwc),
            _u8_flag(obs_next_nhwc), _lib.i64(b), _lib.ptr(out), _lib.current_stream(self.device)))
        return out

    # -- _preprocess_batch (dqn.py:257-275 with the subclass's _target_q) ---------------------------------
    def preprocess(self, buffer: DeviceReplayBuffer, frames: torch.Tensor, indices, stack_num: int,
                   obs_next_frames: torch.Tensor | None = None) -> torch.Tensor:
        """n-step returns float32[I, N]: of the next-state quantiles (QRDQN) or of the support (C51)."""

        def tq_fn(buf, after):
            if self.cfg.kind == C51:                                              # c51.py:120-121
                return self.aux.repeat(after.numel(), 1)
            if obs_next_frames is None:
                on = gather_obs_nhwc(frames, buf, buf.next(after), stack_num, as_u8=True)
            else:
                on = gather_obs_nhwc(obs_next_frames, buf, after, stack_num, as_u8=True)
            return self.next_dist(on)

        class _B:
            pass

        return compute_nstep_return(_B(), buffer, indices, tq_fn, self.cfg.gamma, self.cfg.n_step).returns

    def support_returns(self, buffer: DeviceReplayBuffer, indices) -> torch.Tensor:
        """C51's `preprocess` without the frames: n-step returns of the support need no network (c51.py:120-121)."""
        if self.cfg.kind != C51:
            raise ValueError("support_returns: C51 only (QRDQN's target is the lagged net's output)")
        return self.preprocess(buffer, None, indices, 0)

    def returns_from_obs_next(self, buffer: DeviceReplayBuffer, indices, obs_next_nhwc: torch.Tensor) -> torch.Tensor:
        """QRDQN's `preprocess` for a caller that already holds the observations `_target_q` reads
        (buffer[indices_after_n].obs_next, e.g. dqn.gather_obs_pair's second tensor): next_dist + the arithmetic half of
        compute_nstep_return (algorithm_base.py:793-812) -> float32[I, N]."""
        if self.cfg.kind != QR:
            raise ValueError("returns_from_obs_next: QRDQN only")
        return nstep_return_from_target_q(buffer, indices, self.next_dist(obs_next_nhwc), self.cfg.gamma, self.cfg.n_step)

    def wait_td(self, stream: torch.cuda.Stream) -> None:
        """`stream` waits for the new priorities and the loss of the last `update_with_batch`, not for its backward pass and
        Adam step (ts_dqn_wait_td; see dqn.ReplayStream)."""
        _lib.check(_lib.load().ts_dqn_wait_td(self._ws.handle, C.c_void_p(stream.cuda_stream)))

    # -- _update_with_batch ------------------------------------------------------------------------------
    def update_with_batch(self, obs_nhwc, act, returns, weight=None, obs_next_nhwc=None,
                          grad_out: torch.Tensor | None = None, apply: bool = True, want_target: bool = False):
        """-> (loss float32[1] device tensor, new batch.weight float32[B][, target_dist float32[B, N]])."""
        cfg = self.cfg
        if apply:
            if self.params_old is not None and self.iter % cfg.target_update_freq == 0:    # dqn.py:283-285
                full_parameter_update(self.params_old, self.params)
            self.iter += 1
            self.adam_step += 1
        obs_nhwc = self._check_obs(obs_nhwc)
        b, n = obs_nhwc.shape[0], cfg.n_atoms
        act = _i64_dev(act, self.device).reshape(-1)
        returns = torch.as_tensor(returns, dtype=torch.float32, device=self.device).contiguous()
        if weight is not None:
            weight = torch.as_tensor(weight, device=self.device).to(torch.float32).reshape(-1).contiguous()
        if act.numel() != b or tuple(returns.shape) != (b, n) or (weight is not None and weight.numel() != b):
            raise ValueError("obs / act / returns / weight batch sizes differ")
        nd = None
        if cfg.kind == C51:                                                       # c51.py:148-149
            if obs_next_nhwc is None:
                raise ValueError("C51 needs batch.obs_next")
            nd = self.next_dist(obs_next_nhwc)
        prio = torch.empty(b, dtype=torch.float32, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        tgt = torch.empty((b, n), dtype=torch.float32, device=self.device) if want_target and cfg.kind == C51 else None
        hp = cfg.to_c(grad_only=not apply)
        _lib.check(_lib.load().ts_distq_update(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v),
            _lib.i64(max(self.adam_step, 1)), *self._dims(), _lib.ptr(obs_nhwc), _u8_flag(obs_nhwc), _lib.ptr(act),
            _lib.ptr(returns), _lib.ptr(nd), _lib.ptr(weight), _lib.i64(b), C.byref(hp), _lib.ptr(prio), _lib.ptr(loss),
            _lib.ptr(tgt), _lib.ptr(grad_out), _lib.current_stream(self.device)))
        return (loss, prio, tgt) if want_target else (loss, prio)


def replay_prepare(eng, buffer: DeviceReplayBuffer, frames: torch.Tensor, stack_num: int):
    """`prepare` callable of dqn.ReplayStream for DistQEngine / rainbow.RainbowEngine: what the next batch needs from the
    replay buffer alone -> (obs, obs_next, returns or None), uint8 NHWC observations.
    C51 / Rainbow: obs_next = batch.obs_next (buffer_base.py:624-626) and the n-step returns of the support (c51.py:120-121).
    QRDQN: obs_next = the observations `_target_q` reads n steps on (algorithm_base.py:772-791); the returns follow on
    the caller's stream (`DistQEngine.returns_from_obs_next`: they need both networks)."""
    c51 = eng.cfg.kind == C51
    n = 1 if c51 else eng.cfg.n_step

    def prepare(idx):
        pair = gather_obs_pair(frames, buffer, idx, n, stack_num)
        if pair is None:                                    # layouts outside the pair kernel: the index kernels + two gathers
            after = buffer.next(idx) if c51 else buffer.next(nstep_indices(buffer, idx, n))
            pair = (gather_obs_nhwc(frames, buffer, idx, stack_num, as_u8=True),
                    gather_obs_nhwc(frames, buffer, after, stack_num, as_u8=True))
        return pair[0], pair[1], (eng.support_returns(buffer, idx) if c51 else None)

    return prepare
