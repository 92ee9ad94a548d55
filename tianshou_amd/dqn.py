"""DQN learn() path on the MI355X engine (NatureCNN / DQNet on fp32 MFMA).

Mirrors, on device tensors:
    DQNet.forward                        tianshou/env/atari/atari_network.py:111-122
    DiscreteQLearningPolicy.forward      tianshou/algorithm/modelfree/dqn.py:101-143
    DQN._target_q / _preprocess_batch    dqn.py:257-275, 365-379 (n-step via tianshou_amd.returns)
    DQN._update_with_batch               dqn.py:381-404 (+ periodic hard sync :277-285)
    ReplayBuffer.get with stack_num      data/buffer/buffer_base.py:557-603
There is no CPU path: every function calls libtsengine.so and raises when it is missing.

Internal layouts (include/tsengine.h): observations NHWC float32; per layer one matrix
wb[(kh, kw, ic) + bias row, oc]; fc1 rows in (h, w, c) order.  `flat_from_torch` / `flat_to_torch`
convert from / to the reference's state_dict tensors (and Adam moments) exactly (pure permutations).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .buffer import DeviceReplayBuffer, _i64_dev
from .lagged import full_parameter_update
from .returns import compute_nstep_return, nstep_coefficients, nstep_return_from_target_q

TIANSHOU_KEYS = ["net.0.0.weight", "net.0.0.bias", "net.0.2.weight", "net.0.2.bias",
                 "net.0.4.weight", "net.0.4.bias", "net.1.weight", "net.1.bias",
                 "net.3.weight", "net.3.bias"]


class DQNHParams(C.Structure):
    """struct ts_dqn_hparams (include/tsengine.h)."""

    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("adam_eps", C.c_double),
                ("huber_delta", C.c_double), ("max_grad_norm", C.c_double)]


def _lib_dqn():
    lib = _lib.load()
    lib.ts_dqn_param_count.restype = C.c_int64
    lib.ts_dqn_param_count.argtypes = [C.c_int64] * 4
    return lib


def layer_layout(c: int, h: int, w: int, n_act: int):
    """-> (offsets int64[6], geoms int64[4, 10] rows {B, IH, IW, IC, KH, KW, S, OH, OW, OC})."""
    off = (C.c_int64 * 6)()
    geom = (C.c_int64 * 40)()
    _lib.check(_lib_dqn().ts_dqn_layer_offsets(_lib.i64(c), _lib.i64(h), _lib.i64(w), _lib.i64(n_act), off, geom))
    return np.array(off[:], np.int64), np.array(geom[:], np.int64).reshape(4, 10)


def param_count(c: int, h: int, w: int, n_act: int) -> int:
    n = int(_lib_dqn().ts_dqn_param_count(c, h, w, n_act))
    if n < 0:
        raise ValueError("unsupported DQNet dimensions")
    return n


def flat_from_torch(tensors: list[torch.Tensor], c: int, h: int, w: int, n_act: int, device="cuda") -> torch.Tensor:
    """[conv1.w, conv1.b, conv2.w, ..., fc2.w, fc2.b] in torch layout (DQNet state_dict order, also valid
    for the matching Adam moments) -> the engine's flat vector.  `n_act` = the number of head outputs (any
    width: the trunk geometry does not depend on it)."""
    _, geom = layer_layout(c, h, w, 1)
    oh3, ow3 = int(geom[2, 7]), int(geom[2, 8])
    parts = []
    for i in range(3):
        wt, b = tensors[2 * i].detach().float(), tensors[2 * i + 1].detach().float()
        parts += [wt.permute(2, 3, 1, 0).reshape(-1), b.reshape(-1)]          # [oc, ic, kh, kw] -> [(kh, kw, ic), oc]
    w4, b4 = tensors[6].detach().float(), tensors[7].detach().float()
    parts += [w4.reshape(512, 64, oh3, ow3).permute(2, 3, 1, 0).reshape(-1), b4.reshape(-1)]
    w5, b5 = tensors[8].detach().float(), tensors[9].detach().float()
    parts += [w5.t().reshape(-1), b5.reshape(-1)]
    return torch.cat(parts).to(device).contiguous()


def flat_to_torch(flat: torch.Tensor, c: int, h: int, w: int, n_act: int) -> list[torch.Tensor]:
    """Inverse of flat_from_torch -> ten tensors in torch layout (on flat's device)."""
    off, geom = layer_layout(c, h, w, 1)
    out = []
    for i in range(3):
        ic, kh, kw, oc = (int(geom[i, j]) for j in (3, 4, 5, 9))
        k = kh * kw * ic
        wb = flat[off[i]:off[i + 1]].reshape(k + 1, oc)
        out += [wb[:k].reshape(kh, kw, ic, oc).permute(3, 2, 0, 1).contiguous(), wb[k].clone()]
    oh3, ow3 = int(geom[2, 7]), int(geom[2, 8])
    f = 64 * oh3 * ow3
    wb = flat[off[3]:off[4]].reshape(f + 1, 512)
    out += [wb[:f].reshape(oh3, ow3, 64, 512).permute(3, 2, 0, 1).reshape(512, f).contiguous(), wb[f].clone()]
    wb = flat[off[4]:off[4] + 513 * n_act].reshape(513, n_act)
    out += [wb[:512].t().contiguous(), wb[512].clone()]
    return out


def _u8_flag(obs: torch.Tensor) -> C.c_int:
    """obs_u8 argument of the network entry points: float32 or uint8 NHWC observations."""
    if obs.dtype == torch.uint8:
        return C.c_int(1)
    if obs.dtype == torch.float32:
        return C.c_int(0)
    raise ValueError("observations must be float32 or uint8")


# ---- single layers (tests, other shapes) ---------------------------------------------------------
def _dims(x: torch.Tensor, kh: int, kw: int, stride: int, oc: int):
    b, ih, iw, ic = x.shape
    return (C.c_int64 * 8)(b, ih, iw, ic, kh, kw, stride, oc), ((ih - kh) // stride + 1, (iw - kw) // stride + 1)


def conv_forward(x: torch.Tensor, wb: torch.Tensor, kh: int, kw: int, stride: int, relu: bool) -> torch.Tensor:
    """x float32[B, IH, IW, IC] (NHWC), wb float32[kh*kw*IC + 1, OC] -> y float32[B, OH, OW, OC]."""
    oc = wb.shape[1]
    dims, (oh, ow) = _dims(x, kh, kw, stride, oc)
    y = torch.empty((x.shape[0], oh, ow, oc), dtype=torch.float32, device=x.device)
    ws = _lib.default_workspace(x.device.index or 0)
    _lib.check(_lib.load().ts_conv_forward(ws.handle, _lib.ptr(x), _u8_flag(x), _lib.ptr(wb), _lib.ptr(y), dims,
                                           C.c_int(int(relu)), _lib.current_stream(x.device)))
    return y


def conv_backward(x, wb, dy, kh: int, kw: int, stride: int, mask=None, need_dx: bool = True):
    """-> (d_wb, dx or None); dx is multiplied by (mask > 0) when mask is given."""
    oc = wb.shape[1]
    dims, _ = _dims(x, kh, kw, stride, oc)
    d_wb = torch.empty_like(wb)
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device) if need_dx else None
    ws = _lib.default_workspace(x.device.index or 0)
    _lib.check(_lib.load().ts_conv_backward(ws.handle, _lib.ptr(x), _u8_flag(x), _lib.ptr(wb), _lib.ptr(dy), _lib.ptr(mask),
                                            _lib.ptr(d_wb), _lib.ptr(dx), dims, _lib.current_stream(x.device)))
    return d_wb, dx


# ---- observations ----------------------------------------------------------------------------------
def stack_indices(buffer: DeviceReplayBuffer, index, stack_num: int) -> torch.Tensor:
    """int64[I, stack_num]: column stack_num-1-j = prev^j(index) (buffer_base.py:586-596)."""
    index = _i64_dev(index, buffer.device).reshape(-1)
    out = torch.empty((index.numel(), stack_num), dtype=torch.int64, device=buffer.device)
    _lib.check(_lib.load().ts_stack_indices(
        _lib.ptr(index), _lib.i64(index.numel()), _lib.i64(stack_num), _lib.ptr(buffer.offset),
        _lib.i64(buffer.buffer_num), _lib.ptr(buffer.done), _lib.ptr(buffer.last_index), _lib.ptr(buffer.lengths),
        _lib.ptr(out), _lib.current_stream(buffer.device)))
    return out


def gather_obs_nhwc(frames: torch.Tensor, buffer: DeviceReplayBuffer, index, stack_num: int,
                    as_u8: bool = False) -> torch.Tensor:
    """buffer.get(index, "obs") as the NHWC tensor the network entry points consume: float32, or uint8 with
    as_u8 (the kernels then convert on load and the float32 copy -- 4x the bytes -- never exists).

    frames: uint8 [slots, H, W] with stack_num > 1 (save_only_last_obs layout,
    examples/atari/atari_dqn.py:137-142) or uint8 [slots, C, H, W] with stack_num == 1."""
    index = _i64_dev(index, buffer.device).reshape(-1)
    if stack_num > 1:
        if frames.dim() != 3:
            raise ValueError("stacked layout expects frames [slots, H, W]")
        planes = stack_indices(buffer, index, stack_num)
        c, (hh, ww) = stack_num, frames.shape[1:]
        n_planes = frames.shape[0]
    else:
        if frames.dim() != 4:
            raise ValueError("unstacked layout expects frames [slots, C, H, W]")
        c, hh, ww = frames.shape[1:]
        planes = (index[:, None] * c + torch.arange(c, device=index.device)[None, :]).contiguous()
        n_planes = frames.shape[0] * c
    if frames.dtype != torch.uint8 or not frames.is_contiguous():
        raise ValueError("frames must be a contiguous uint8 tensor")
    out = torch.empty((index.numel(), hh, ww, c), dtype=torch.uint8 if as_u8 else torch.float32, device=frames.device)
    step = 32768
    fn = _lib.load().ts_gather_planes_nhwc_u8 if as_u8 else _lib.load().ts_gather_planes_nhwc
    for lo in range(0, index.numel(), step):
        hi = min(index.numel(), lo + step)
        _lib.check(fn(
            _lib.ptr(frames), _lib.i64(n_planes), _lib.i64(hh * ww), _lib.ptr(planes[lo:hi]), _lib.i64(hi - lo),
            _lib.i64(c), _lib.ptr(out[lo:hi]), _lib.current_stream(frames.device)))
    return out


def gather_obs_pair(frames: torch.Tensor, buffer: DeviceReplayBuffer, index, n_step: int, stack_num: int):
    """(buffer.get(index, "obs"), the obs_next that `_target_q` reads at next(indices_after_n)) as two uint8 NHWC
    tensors from ONE launch (ts_dqn_gather_pair), for a buffer that stores single frames and no obs_next
    (examples/atari/atari_dqn.py:137-142; buffer_base.py:586-596, 624-626; algorithm_base.py:772-791).
    None when the layout is outside the kernel's (stack_num 4, frame size a multiple of 16 bytes): the caller then
    takes the index kernels + gather_obs_nhwc."""
    if (stack_num != 4 or frames.dim() != 3 or frames.dtype != torch.uint8 or not frames.is_contiguous()
            or (frames.shape[1] * frames.shape[2]) % 16 or frames.data_ptr() % 16 or os.environ.get("TS_DQN_NO_PAIR")):
        return None
    index = _i64_dev(index, buffer.device).reshape(-1)
    b, (hh, ww) = index.numel(), frames.shape[1:]
    obs = torch.empty((b, hh, ww, 4), dtype=torch.uint8, device=frames.device)
    obs_next = torch.empty_like(obs)
    _lib.check(_lib.load().ts_dqn_gather_pair(
        _lib.ptr(frames), _lib.i64(frames.shape[0]), _lib.i64(hh * ww), _lib.ptr(index), _lib.i64(b), _lib.i64(n_step),
        _lib.i64(stack_num), _lib.ptr(buffer.offset), _lib.i64(buffer.buffer_num), _lib.ptr(buffer.done),
        _lib.ptr(buffer.last_index), _lib.ptr(buffer.lengths), _lib.ptr(obs), _lib.ptr(obs_next),
        _lib.current_stream(frames.device)))
    return obs, obs_next


@dataclass
class DQNConfig:
    """Hyper-parameters of the reference DQN (dqn.py:309-363) + Adam (optim.py:89-110)."""

    gamma: float = 0.99
    n_step: int = 1
    target_update_freq: int = 0
    is_double: bool = True
    huber_delta: float | None = None
    lr: float = 1e-3
    betas: tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8
    max_grad_norm: float | None = None

    def to_c(self, grad_only: bool = False) -> DQNHParams:
        return DQNHParams(-1.0 if grad_only else self.lr, self.betas[0], self.betas[1], self.adam_eps,
                          self.huber_delta if self.huber_delta is not None else -1.0, self.max_grad_norm or 0.0)


class DQNEngine:
    """State of one DQN learner on one GPU: flat parameters, lagged copy, Adam moments, counters."""

    def __init__(self, c: int, h: int, w: int, n_act: int, flat_params: torch.Tensor, cfg: DQNConfig):
        if not flat_params.is_cuda:
            raise RuntimeError("DQNEngine needs parameters on an MI355X (no CPU fallback)")
        self.c, self.h, self.w, self.n_act, self.cfg = c, h, w, n_act, cfg
        self.P = param_count(c, h, w, n_act)
        if flat_params.numel() != self.P:
            raise ValueError(f"expected {self.P} parameters, got {flat_params.numel()}")
        self.device = flat_params.device
        self.params = flat_params.detach().float().contiguous().clone()
        self.params_old = self.params.clone() if cfg.target_update_freq > 0 else None    # dqn.py:240-246
        self.adam_m = torch.zeros_like(self.params)
        self.adam_v = torch.zeros_like(self.params)
        self.adam_step = 0
        self.iter = 0
        self._ws = _lib.default_workspace(self.device.index or 0)
        self._pre = None               # (obs tensor, cache, done event, params version) of a prefetched forward pass
        self._side = None
        self._rows = None              # state of `learn_rows` (scratch, replay view)
        self._learn = None             # state of `learn_step` (scratch, replay view, the seed of the batch prepared ahead)

    # -- the forward pass on batch.obs, ahead of time ------------------------------------------------
    def prefetch_forward(self, obs_nhwc: torch.Tensor) -> None:
        """Q_online(batch.obs) of the coming `update_with_batch(obs_nhwc, ...)` on a side stream, beside the two obs_next
        passes of `_target_q`: it needs nothing from them (same online parameters, dqn.py:257-275 vs 381-404), and at B = 512
        no single pass fills the chip.  The activations wait in a cache owned by the engine; `update_with_batch` picks them up
        when it is handed the SAME tensor and the parameters have not been written since (ts_dqn_update_cached), and falls
        back to its own forward pass otherwise."""
        lib = _lib.load()
        lib.ts_dqn_cache_bytes.restype = C.c_int64
        b = obs_nhwc.shape[0]
        need = int(lib.ts_dqn_cache_bytes(_lib.i64(self.c), _lib.i64(self.h), _lib.i64(self.w), _lib.i64(self.n_act), _lib.i64(b)))
        if need <= 0 or not obs_nhwc.is_contiguous():
            return
        if self._side is None:
            # the workspace's second side stream (idle during the forward passes; the backward chain uses it later): a stream
            # of our own would be the fifth of the update on four hardware queues (ts_workspace_side_stream)
            h = C.c_void_p()
            _lib.check(lib.ts_workspace_side_stream(self._ws.handle, C.c_int(1), C.byref(h)))
            self._side = torch.cuda.ExternalStream(h.value, device=self.device)
            self._cache = None
        if self._cache is None or self._cache.numel() < need + 256:       # + 256: the pointer is aligned up below
            if self._cache is not None:
                self._cache.record_stream(self._side)                     # a side-stream kernel may still be reading it
            self._cache = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        base = self._cache.data_ptr()
        cache_ptr = C.c_void_p((base + 255) & ~255)
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)                                  # obs (and anything else enqueued so far) is ready
        done = torch.cuda.Event()
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            ws = _lib.default_workspace(self.device.index or 0)          # the side stream's own workspace
            _lib.check(lib.ts_dqn_forward_cache(
                ws.handle, _lib.ptr(self.params), _lib.i64(self.c), _lib.i64(self.h), _lib.i64(self.w), _lib.i64(self.n_act),
                _lib.ptr(obs_nhwc), _u8_flag(obs_nhwc), _lib.i64(b), cache_ptr, _lib.i64(need),
                _lib.current_stream(self.device)))
            done.record(self._side)
        obs_nhwc.record_stream(self._side)
        self._pre = (obs_nhwc, cache_ptr, done, (self.params._version, self.adam_step), b)

    def wait_td(self, stream: torch.cuda.Stream) -> None:
        """`stream` waits for the TD errors and the loss of the last `update_with_batch` -- not for its backward pass and
        Adam step (ts_dqn_wait_td).  What `_postprocess_batch` does with them (PrioritizedReplayBuffer.update_weight,
        prio.py:89-100) and the sampling of the next batch can then run on `stream` beside the rest of the update."""
        _lib.check(_lib.load().ts_dqn_wait_td(self._ws.handle, C.c_void_p(stream.cuda_stream)))

    # -- DiscreteQLearningPolicy.forward ---------------------------------------------------------
    def forward(self, obs_nhwc: torch.Tensor, params: torch.Tensor | None = None, want_act: bool = True):
        """-> (logits float32[B, A], act int64[B] = argmax)."""
        b = obs_nhwc.shape[0]
        if tuple(obs_nhwc.shape[1:]) != (self.h, self.w, self.c) or obs_nhwc.dtype not in (torch.float32, torch.uint8):
            raise ValueError(f"obs must be float32 or uint8 [B, {self.h}, {self.w}, {self.c}] (NHWC)")
        obs_nhwc = obs_nhwc.contiguous()
        q = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device)
        act = torch.empty(b, dtype=torch.int64, device=self.device) if want_act else None
        p = self.params if params is None else params
        _lib.check(_lib.load().ts_dqn_forward(
            self._ws.handle, _lib.ptr(p), _lib.i64(self.c), _lib.i64(self.h), _lib.i64(self.w), _lib.i64(self.n_act),
            _lib.ptr(obs_nhwc), _u8_flag(obs_nhwc), _lib.i64(b), _lib.ptr(q), _lib.ptr(act),
            _lib.current_stream(self.device)))
        return q, act

    # -- DQN._target_q ---------------------------------------------------------------------------------
    def target_q(self, obs_next_nhwc: torch.Tensor) -> torch.Tensor:
        b = obs_next_nhwc.shape[0]
        obs_next_nhwc = obs_next_nhwc.contiguous()
        out = torch.empty(b, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().ts_dqn_target_q_fused(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.params_old), _lib.i64(self.c), _lib.i64(self.h),
            _lib.i64(self.w), _lib.i64(self.n_act), _lib.ptr(obs_next_nhwc), _u8_flag(obs_next_nhwc), _lib.i64(b),
            C.c_int(int(self.cfg.is_double)), _lib.ptr(out), _lib.current_stream(self.device)))
        return out

    def target_returns(self, obs_next_nhwc: torch.Tensor, coef) -> torch.Tensor:
        """_target_q followed by the n-step return arithmetic in the same final kernel (ts_dqn_target_returns); coef =
        returns.nstep_coefficients(buffer, indices, gamma, n_step)."""
        b = obs_next_nhwc.shape[0]
        obs_next_nhwc = obs_next_nhwc.contiguous()
        mask, gpow, mc = coef
        if mask.numel() != b or gpow.numel() != b or mc.numel() != b:
            raise ValueError("n-step coefficients / obs_next batch sizes differ")
        out = torch.empty(b, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().ts_dqn_target_returns(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.params_old), _lib.i64(self.c), _lib.i64(self.h),
            _lib.i64(self.w), _lib.i64(self.n_act), _lib.ptr(obs_next_nhwc), _u8_flag(obs_next_nhwc), _lib.i64(b),
            C.c_int(int(self.cfg.is_double)), _lib.ptr(mask), _lib.ptr(gpow), _lib.ptr(mc), _lib.ptr(out),
            _lib.current_stream(self.device)))
        return out

    # -- DQN._preprocess_batch -----------------------------------------------------------------------
    def preprocess(self, buffer: DeviceReplayBuffer, frames: torch.Tensor, indices, stack_num: int,
                   obs_next_frames: torch.Tensor | None = None) -> torch.Tensor:
        """n-step returns float32[I] with target_q_fn = _target_q (dqn.py:257-275).  When the buffer does
        not store obs_next, s_{t+n} is read at next(indices_after_n) (buffer_base.py:624-626)."""

        def tq_fn(buf, after):
            if obs_next_frames is None:
                on = gather_obs_nhwc(frames, buf, buf.next(after), stack_num, as_u8=True)
            else:
                on = gather_obs_nhwc(obs_next_frames, buf, after, stack_num, as_u8=True)
            return self.target_q(on)

        class _B:
            pass

        b = compute_nstep_return(_B(), buffer, indices, tq_fn, self.cfg.gamma, self.cfg.n_step)
        return b.returns.reshape(-1)

    def preprocess_with_obs(self, buffer: DeviceReplayBuffer, frames: torch.Tensor, indices, stack_num: int,
                            obs_next_frames: torch.Tensor | None = None, prefetch: bool = True, pair=None, coef=None):
        """-> (obs uint8 NHWC [I, H, W, C], returns float32[I]): `preprocess` plus the batch's own observations, which
        `update_with_batch` wants next.  On a frame buffer without obs_next both stacked gathers come from one launch
        (gather_obs_pair) and the forward pass on obs is started on the side stream before the target passes
        (prefetch_forward); otherwise this is gather_obs_nhwc + preprocess.  `pair`: the two gathers, already done by the
        caller (ReplayStream), as are the n-step coefficients `coef` (returns.nstep_coefficients)."""
        if pair is None and obs_next_frames is None:
            pair = gather_obs_pair(frames, buffer, indices, self.cfg.n_step, stack_num)
        if pair is None:
            obs = gather_obs_nhwc(frames, buffer, indices, stack_num, as_u8=True)
            if prefetch:
                self.prefetch_forward(obs)
            return obs, self.preprocess(buffer, frames, indices, stack_num, obs_next_frames)
        obs, obs_next = pair
        if prefetch:
            self.prefetch_forward(obs)
        if coef is None:
            coef = nstep_coefficients(buffer, indices, self.cfg.gamma, self.cfg.n_step)
        return obs, self.target_returns(obs_next, coef)

    # -- the two halves of an update, for the data-parallel path (tianshou_amd.distributed.DataParallelDQN) ----
    def gradient(self, obs_nhwc, act, returns, weight, grad_out: torch.Tensor):
        """Periodic target sync + forward + loss + backward; grad_out[:P] = d loss / d params (mean over THIS
        batch), no optimizer step.  -> (loss, td_error)."""
        if self.params_old is not None and self.iter % self.cfg.target_update_freq == 0:    # dqn.py:283-285
            full_parameter_update(self.params_old, self.params)
        self.iter += 1
        return self.update_with_batch(obs_nhwc, act, returns, weight, grad_out=grad_out, apply=False)

    def apply_gradient(self, grad: torch.Tensor) -> None:
        """clip_grad_norm_ + Adam on a flat gradient (algorithm_base.py:496-500)."""
        cfg = self.cfg
        self.adam_step += 1
        _lib.check(_lib.load().ts_adam_step(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v), _lib.ptr(grad),
            _lib.i64(self.P), _lib.i64(self.adam_step), _lib.f64(cfg.lr), _lib.f64(cfg.betas[0]),
            _lib.f64(cfg.betas[1]), _lib.f64(cfg.adam_eps), _lib.f64(cfg.max_grad_norm or 0.0),
            _lib.current_stream(self.device)))

    # -- DQN._update_with_batch -----------------------------------------------------------------------
    def update_with_batch(self, obs_nhwc, act, returns, weight=None, grad_out: torch.Tensor | None = None,
                          apply: bool = True):
        """-> (loss float32[1] device tensor, td_error float32[B]); td_error is the new batch.weight."""
        cfg = self.cfg
        params_state = (self.params._version, self.adam_step)          # what a prefetched forward pass was computed with
        if apply:
            if self.params_old is not None and self.iter % cfg.target_update_freq == 0:    # dqn.py:283-285
                full_parameter_update(self.params_old, self.params)
            self.iter += 1
            self.adam_step += 1
        b = obs_nhwc.shape[0]
        act = _i64_dev(act, self.device).reshape(-1)
        returns = torch.as_tensor(returns, dtype=torch.float32, device=self.device).reshape(-1).contiguous()
        if weight is not None:
            weight = torch.as_tensor(weight, device=self.device).to(torch.float32).reshape(-1).contiguous()
        if act.numel() != b or returns.numel() != b or (weight is not None and weight.numel() != b):
            raise ValueError("obs / act / returns / weight batch sizes differ")
        td = torch.empty(b, dtype=torch.float32, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        hp = cfg.to_c(grad_only=not apply)
        pre, self._pre = self._pre, None
        if pre is not None and pre[0] is obs_nhwc and pre[3] == params_state and pre[4] == b:
            torch.cuda.current_stream(self.device).wait_event(pre[2])          # the prefetched activations are complete
            _lib.check(_lib.load().ts_dqn_update_cached(
                self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v),
                _lib.i64(max(self.adam_step, 1)), _lib.i64(self.c), _lib.i64(self.h), _lib.i64(self.w),
                _lib.i64(self.n_act), _lib.ptr(obs_nhwc), _u8_flag(obs_nhwc), _lib.ptr(act), _lib.ptr(returns), _lib.ptr(weight),
                _lib.i64(b), C.byref(hp), pre[1], _lib.ptr(td), _lib.ptr(loss), _lib.ptr(grad_out),
                _lib.current_stream(self.device)))
            return loss, td
        obs_nhwc = obs_nhwc.contiguous()
        _lib.check(_lib.load().ts_dqn_update(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v),
            _lib.i64(max(self.adam_step, 1)), _lib.i64(self.c), _lib.i64(self.h), _lib.i64(self.w),
            _lib.i64(self.n_act), _lib.ptr(obs_nhwc), _u8_flag(obs_nhwc), _lib.ptr(act), _lib.ptr(returns), _lib.ptr(weight),
            _lib.i64(b), C.byref(hp), _lib.ptr(td), _lib.ptr(loss), _lib.ptr(grad_out),
            _lib.current_stream(self.device)))
        return loss, td

    # -- sample + preprocess + update + priority update in one call (device-resident Atari-layout buffer) -------------------------
    def learn_step(self, buffer: DeviceReplayBuffer, frames: torch.Tensor, act_col: torch.Tensor, per, batch_size: int, seed,
                   want_td: bool = False):
        """OffPolicyAlgorithm.update (algorithm_base.py:583-631) as ONE library call (ts_dqn_learn_step): what `ReplayStream.take`
        -> `preprocess_with_obs(prefetch=True)` -> `update_with_batch` -> `ReplayStream.give` do -- the same kernels on the same
        values in the same order -- with the draws of update number `counter` taken from the engine's own Philox stream
        (`uniform_draws(batch_size, seed)`; without priorities `buffer.sample_indices(batch_size, seed=seed)`).
        -> (loss float32[1], td_error float32[B] or None).
        per: segtree.PrioritizedWeights or None.  seed = (key, counter), counter advancing by one per call; call `learn_reset()`
        after writing to the buffer or its priorities (the batch prepared ahead of time predates the write).  uint8 contiguous
        frames [slots, h, w], frame stack = the network's channel count = 4, int64 actions (no fallback: use the separate calls
        otherwise)."""
        lib = _lib.load()
        key, counter = int(seed[0]) & (2**64 - 1), int(seed[1]) & (2**64 - 1)
        b = int(batch_size)
        st = self._learn
        ident = (id(buffer), frames.data_ptr(), act_col.data_ptr(), id(per), b)
        if st is None or st["ident"] != ident:
            if not (frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous() and frames.dim() == 3
                    and tuple(frames.shape[1:]) == (self.h, self.w)):
                raise ValueError(f"learn_step: frames must be uint8 contiguous [slots, {self.h}, {self.w}] on the device")
            if not (act_col.is_cuda and act_col.dim() == 1 and act_col.dtype == torch.int64 and act_col.is_contiguous()):
                raise ValueError("learn_step: actions must be an int64 contiguous device column")
            if self.c != 4 or (self.h * self.w) % 16:
                raise NotImplementedError("learn_step: frame stack 4 and planes of a multiple of 16 bytes (ts_dqn_gather_pair)")
            if len(buffer) == 0:                             # buffer_base.py:512-513
                raise ValueError("learn_step: empty buffer")
            lib.ts_dqn_learn_scratch_bytes.restype = C.c_int64
            need = int(lib.ts_dqn_learn_scratch_bytes(_lib.i64(self.c), _lib.i64(self.h), _lib.i64(self.w), _lib.i64(self.n_act),
                                                      _lib.i64(b)))
            if need <= 0:
                raise ValueError("learn_step: unsupported network / batch dimensions")
            scratch = torch.zeros(need + 256, dtype=torch.uint8, device=self.device)
            tree = per.weight._value if per is not None else None
            view = _lib.FrameReplay(
                _lib.ptr(buffer.offset), buffer.buffer_num, _lib.ptr(buffer.lengths), _lib.ptr(buffer.last_index),
                _lib.ptr(buffer.done), _lib.ptr(buffer.terminated), _lib.ptr(buffer.rew), _lib.ptr(frames), self.h * self.w,
                _lib.ptr(act_col), frames.shape[0], _lib.ptr(tree), per.weight._bound if per is not None else 0,
                _lib.ptr(per.prio_minmax) if per is not None else None, per._alpha if per is not None else 0.0,
                per._beta if per is not None else 0.0, int(per._weight_norm) if per is not None else 0, 0)
            st = self._learn = {"ident": ident, "scratch": scratch, "ptr": C.c_void_p((scratch.data_ptr() + 255) & ~255),
                                "bytes": _lib.i64(need), "view": view, "aux": _lib.aux_workspace(self.device.index or 0),
                                "last": None, "keep": (buffer, frames, act_col, per), "B": _lib.i64(b),
                                "dims": (_lib.i64(self.c), _lib.i64(self.h), _lib.i64(self.w), _lib.i64(self.n_act))}
        cfg = self.cfg
        sync = self.params_old is not None and self.iter % cfg.target_update_freq == 0    # dqn.py:283-285, applied inside the call
        self.iter += 1
        self.adam_step += 1
        self._pre = None
        td = torch.empty(b, dtype=torch.float32, device=self.device) if want_td else None
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        hp_of = (cfg.lr, cfg.betas, cfg.adam_eps, cfg.huber_delta, cfg.max_grad_norm)
        if st.get("hp_of") != hp_of:
            st["hp"], st["hp_of"] = cfg.to_c(), hp_of
        prepared = st["last"] == (key, (counter - 1) & (2**64 - 1))
        dims = st["dims"]
        _lib.check(lib.ts_dqn_learn_step(
            self._ws.handle, st["aux"].handle, _lib.ptr(self.params), _lib.ptr(self.params_old), C.c_int(int(sync)),
            _lib.ptr(self.adam_m), _lib.ptr(self.adam_v), _lib.i64(self.adam_step), dims[0], dims[1], dims[2], dims[3],
            C.byref(st["view"]), st["B"],
            _lib.i64(cfg.n_step), _lib.f64(cfg.gamma), C.c_int(int(cfg.is_double)), C.byref(st["hp"]), C.c_uint64(key),
            C.c_uint64(counter), C.c_int(int(prepared)), st["ptr"], st["bytes"], _lib.ptr(td), _lib.ptr(loss), None,
            _lib.current_stream(self.device)))
        st["last"] = (key, counter)
        return loss, td

    def rows_ok(self, buffer: DeviceReplayBuffer, frames, act_col) -> bool:
        """Whether `learn_rows` applies: uint8 contiguous single frames, frame stack 4 = the network's channels, no stored obs_next
        (ReplayBuffer(stack_num=4, ignore_obs_next=True, save_only_last_obs=True), examples/atari/atari_dqn.py), int64 actions."""
        return (self.c == 4 and (self.h * self.w) % 16 == 0 and frames is not None and frames.is_cuda and frames.dtype == torch.uint8
                and frames.dim() == 3 and tuple(frames.shape[1:]) == (self.h, self.w) and frames.is_contiguous()
                and buffer.obs_next is None and act_col is not None and act_col.is_cuda and act_col.dim() == 1
                and act_col.dtype == torch.int64 and act_col.is_contiguous())

    def learn_rows(self, buffer: DeviceReplayBuffer, frames: torch.Tensor, act_col: torch.Tensor, indices: torch.Tensor, weight=None,
                   want_returns: bool = True):
        """`preprocess_with_obs(prefetch=True)` -> `update_with_batch` for a batch the CALLER drew, as ONE library call
        (ts_dqn_learn_rows): the two hooks of `HipDQN.update()` over a host buffer whose sample_indices / get_weight made `indices`
        int64[B] and `weight` (importance weights, or None).  Same kernels on the same values as the two calls.
        -> (loss float32[1], td_error float32[B], returns float32[B] or None).  Requires `rows_ok`."""
        lib = _lib.load()
        b = int(indices.numel())
        st = self._rows
        ident = (id(buffer), frames.data_ptr(), act_col.data_ptr(), b, buffer.offset.data_ptr(), buffer.rew.data_ptr(), buffer.done.data_ptr())
        if st is None or st["ident"] != ident:
            if not self.rows_ok(buffer, frames, act_col):
                raise ValueError("learn_rows: uint8 contiguous [slots, h, w] frames, frame stack 4, no stored obs_next, int64 actions")
            lib.ts_dqn_learn_scratch_bytes.restype = C.c_int64
            need = int(lib.ts_dqn_learn_scratch_bytes(_lib.i64(self.c), _lib.i64(self.h), _lib.i64(self.w), _lib.i64(self.n_act), _lib.i64(b)))
            if need <= 0:
                raise ValueError("learn_rows: unsupported network / batch dimensions")
            scratch = torch.zeros(need + 256, dtype=torch.uint8, device=self.device)
            view = _lib.FrameReplay(
                _lib.ptr(buffer.offset), buffer.buffer_num, _lib.ptr(buffer.lengths), _lib.ptr(buffer.last_index),
                _lib.ptr(buffer.done), _lib.ptr(buffer.terminated), _lib.ptr(buffer.rew), _lib.ptr(frames), self.h * self.w,
                _lib.ptr(act_col), frames.shape[0], None, 0, None, 0.0, 0.0, 0, 0)
            st = self._rows = {"ident": ident, "scratch": scratch, "ptr": C.c_void_p((scratch.data_ptr() + 255) & ~255),
                               "bytes": _lib.i64(need), "view": view, "aux": _lib.aux_workspace(self.device.index or 0),
                               "keep": (buffer, frames, act_col), "B": _lib.i64(b),
                               "dims": (_lib.i64(self.c), _lib.i64(self.h), _lib.i64(self.w), _lib.i64(self.n_act))}
        if not (indices.is_cuda and indices.dtype == torch.int64 and indices.is_contiguous()):
            raise ValueError("learn_rows: indices must be an int64 contiguous device tensor")
        if weight is not None:
            weight = weight.to(device=self.device, dtype=torch.float32).contiguous().reshape(b)
        cfg = self.cfg
        sync = self.params_old is not None and self.iter % cfg.target_update_freq == 0    # dqn.py:283-285, applied inside the call
        self.iter += 1
        self.adam_step += 1
        self._pre = None
        td = torch.empty(b, dtype=torch.float32, device=self.device)
        ret = torch.empty(b, dtype=torch.float32, device=self.device) if want_returns else None
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        hp_of = (cfg.lr, cfg.betas, cfg.adam_eps, cfg.huber_delta, cfg.max_grad_norm)
        if st.get("hp_of") != hp_of:
            st["hp"], st["hp_of"] = cfg.to_c(), hp_of
        dims = st["dims"]
        _lib.check(lib.ts_dqn_learn_rows(
            self._ws.handle, st["aux"].handle, _lib.ptr(self.params), _lib.ptr(self.params_old), C.c_int(int(sync)),
            _lib.ptr(self.adam_m), _lib.ptr(self.adam_v), _lib.i64(self.adam_step), dims[0], dims[1], dims[2], dims[3],
            C.byref(st["view"]), _lib.ptr(indices), _lib.ptr(weight), st["B"], _lib.i64(cfg.n_step), _lib.f64(cfg.gamma),
            C.c_int(int(cfg.is_double)), C.byref(st["hp"]), st["ptr"], st["bytes"], _lib.ptr(ret), _lib.ptr(td), _lib.ptr(loss),
            _lib.current_stream(self.device)))
        return loss, td, ret

    def learn_reset(self) -> None:
        """Drops the batch `learn_step` prepared ahead of time (call it after transitions were written to the buffer)."""
        if self._learn is not None:
            self._learn["last"] = None

    def learn_graph_launches(self) -> int:
        """Updates `learn_step` replayed from a captured HIP graph so far (-1: a capture failed, the streams are used)."""
        lib = _lib.load()
        lib.ts_dqn_learn_graph_launches.restype = C.c_int64
        return int(lib.ts_dqn_learn_graph_launches(self._ws.handle))


def uniform_draws(n: int, seed, device="cuda") -> torch.Tensor:
    """float64[n] in [0, 1): the draws `DQNEngine.learn_step` makes for update seed = (key, counter) in place of the reference's
    np.random.rand(batch_size) (prio.py:65) -- ts_uniform_fill_f64."""
    out = torch.empty(int(n), dtype=torch.float64, device=device)
    _lib.check(_lib.load().ts_uniform_fill_f64(_lib.ptr(out), _lib.i64(n), C.c_uint64(int(seed[0]) & (2**64 - 1)),
                                               C.c_uint64(int(seed[1]) & (2**64 - 1)), _lib.current_stream(out.device)))
    return out


class ReplayStream:
    """The replay half of a prioritized off-policy cycle on its own stream.

    Reference order per update (trainer.py:1093 -> algorithm_base.py:583-631): buffer.sample -> _preprocess_batch ->
    _update_with_batch -> _postprocess_batch (PrioritizedReplayBuffer.update_weight with the TD errors, prio.py:89-100).
    The TD errors exist right after the forward pass and the loss of `_update_with_batch`; the priority update, the next
    batch's draws, its sum-tree descent and its frame gathers need nothing from the backward pass or the optimizer step
    that follow.  Here they are issued on a second stream behind `DQNEngine.wait_td`, so they run beside the backward
    pass; the caller's stream picks the finished batch up with an event.  Same operations on the same values in the same
    order as the sequential loop -- only their placement in time differs.

    draw: () -> float64[batch] uniform draws on the device (prio.py:65); act_of: index tensor -> actions (or None).
    per: the device-resident priorities (segtree.PrioritizedWeights) or None for a uniform buffer -- `draw` then returns the
    sampled indices themselves (ReplayBuffer.sample_indices) and `give` only prepares the next batch.
    prepare: index tensor -> tuple of (nested tuples of) device tensors or None, everything else of the batch that needs
    neither network (default: DQN's observation pair and n-step coefficients; distq.replay_prepare for QRDQN / C51 /
    Rainbow).  `eng` is any engine with `wait_td` (its update records the event behind its priority kernel)."""

    def __init__(self, eng, buffer: DeviceReplayBuffer, frames: torch.Tensor, per, stack_num: int, draw, act_of, prepare=None):
        if per is not None and not hasattr(eng, "wait_td"):
            raise NotImplementedError(f"{type(eng).__name__} records no TD-error event (wait_td): prioritized replay needs "
                                      "an engine whose update calls ts::record_td (DQN / QRDQN / C51 / Rainbow)")
        self.eng, self.buffer, self.frames, self.per, self.stack, self.draw, self.act_of = eng, buffer, frames, per, stack_num, draw, act_of
        self.prepare = prepare if prepare is not None else self._dqn_prepare
        self.stream = torch.cuda.Stream(device=eng.device)
        self._next = None
        self._ready = None

    def _dqn_prepare(self, idx):
        cfg = self.eng.cfg
        pair = gather_obs_pair(self.frames, self.buffer, idx, cfg.n_step, self.stack)
        coef = nstep_coefficients(self.buffer, idx, cfg.gamma, cfg.n_step) if pair is not None else None
        return pair, coef

    def _sample(self):
        if self.per is None:                        # uniform sampling: `draw` returns the indices themselves
            idx, wt = self.draw(), None
        else:
            idx, wt = self.per.sample(self.draw())
            wt = wt.to(torch.float32)               # what update_with_batch converts the importance weights to
        self._next = (idx, wt, self.act_of(idx) if self.act_of is not None else None) + tuple(self.prepare(idx))
        self._ready = torch.cuda.Event()
        self._ready.record(self.stream)

    @staticmethod
    def _tensors(x):
        if isinstance(x, torch.Tensor):
            yield x
        elif isinstance(x, (tuple, list)):
            for y in x:
                yield from ReplayStream._tensors(y)

    def reset(self) -> None:
        """Drops the batch prepared ahead of time (call it after transitions were added to the buffer: the prepared indices
        and priorities predate them); the next `take` samples afresh."""
        self._next = None

    def take(self):
        """-> (indices, IS weights float32, actions, *prepare(indices)) of the next batch, ready on the caller's stream; with
        the default `prepare`: (..., (obs, obs_next) or None, n-step coefficients or None)."""
        main = torch.cuda.current_stream(self.eng.device)
        if self._next is None:
            self.stream.wait_stream(main)
            with torch.cuda.stream(self.stream):
                self._sample()
        main.wait_event(self._ready)
        out, self._next = self._next, None
        for t in self._tensors(out):
            t.record_stream(main)
        return out

    def give(self, indices: torch.Tensor, td: torch.Tensor) -> None:
        """After `update_with_batch`: priority update with its TD errors and the next batch, beside the rest of the update.
        Without priorities (`per` None) only the next batch: it depends on nothing of the update."""
        if self.per is None:
            self.stream.wait_stream(torch.cuda.current_stream(self.eng.device))    # buffer writes enqueued so far
            with torch.cuda.stream(self.stream):
                self._sample()
            return
        self.eng.wait_td(self.stream)
        indices.record_stream(self.stream)
        td.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            self.per.update_weight(indices, td)
            self._sample()
