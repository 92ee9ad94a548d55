"""DQN on a recurrent Q network (DRQN) on the MI355X engine.

Mirrors, on device tensors:
    Recurrent.forward                    tianshou/utils/net/common.py:400-452 (fc1 -> nn.LSTM -> fc2 on the last step)
    DiscreteQLearningPolicy.forward      tianshou/algorithm/modelfree/dqn.py:101-143
    DQN._target_q / _preprocess_batch    dqn.py:257-275, 365-379 (n-step via tianshou_amd.returns)
    DQN._update_with_batch               dqn.py:381-404 (+ periodic hard sync :277-285)
    ReplayBuffer.get with stack_num      data/buffer/buffer_base.py:586-596 on vector observations
Setup: test/discrete/test_drqn.py:79-108.  There is no CPU path: every function calls libtsengine.so.

Flat layout (include/tsengine.h, ts_rnnq_layout): fc1 [k0 + 1, H] | per layer W_ih [H + 1, 4H] | W_hh [H + 1, 4H] |
fc2 [H + 1, 32]; `flat_from_torch` / `flat_to_torch` convert from / to Recurrent.state_dict() order (and Adam moments).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .buffer import DeviceReplayBuffer, _i64_dev, gather_rows
from .dqn import DQNConfig, stack_indices
from .lagged import full_parameter_update
from .returns import compute_nstep_return, nstep_coefficients, nstep_indices

HEAD = 32


def state_dict_keys(layers: int) -> list[str]:
    """Recurrent.state_dict() order (the LSTM is constructed first, common.py:386-393)."""
    ks = []
    for k in range(layers):
        ks += [f"nn.weight_ih_l{k}", f"nn.weight_hh_l{k}", f"nn.bias_ih_l{k}", f"nn.bias_hh_l{k}"]
    return ks + ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]


def layout(obs_dim: int, hidden: int, layers: int, n_act: int) -> dict:
    out = (C.c_int64 * (4 + 2 * layers))()
    _lib.check(_lib.load().ts_rnnq_layout(_lib.i64(obs_dim), _lib.i64(hidden), _lib.i64(layers), _lib.i64(n_act), out))
    return {"k0": int(out[0]), "count": int(out[1]), "fc1": int(out[2]), "ih": [int(out[3 + 2 * l]) for l in range(layers)],
            "hh": [int(out[4 + 2 * l]) for l in range(layers)], "fc2": int(out[3 + 2 * layers])}


def _block(w: torch.Tensor, b: torch.Tensor, k_pad: int, n_pad: int) -> torch.Tensor:
    """nn.Linear-layout weight [n, k] + bias [n] -> [k_pad + 1, n_pad] (zero padding, bias in the last row)."""
    n, k = w.shape
    m = torch.zeros((k_pad + 1, n_pad), dtype=torch.float32)
    m[:k, :n] = w.detach().float().cpu().t()
    m[k_pad, :n] = b.detach().float().cpu().reshape(-1)
    return m.reshape(-1)


def flat_from_torch(t: list[torch.Tensor], obs_dim: int, hidden: int, layers: int, n_act: int, device="cuda") -> torch.Tensor:
    """Tensors in state_dict_keys(layers) order -> the engine's flat vector."""
    k0 = (obs_dim + 31) // 32 * 32
    parts = [_block(t[4 * layers], t[4 * layers + 1], k0, hidden)]
    for k in range(layers):
        w_ih, w_hh, b_ih, b_hh = t[4 * k:4 * k + 4]
        parts += [_block(w_ih, b_ih, hidden, 4 * hidden), _block(w_hh, b_hh, hidden, 4 * hidden)]
    parts.append(_block(t[4 * layers + 2], t[4 * layers + 3], hidden, HEAD))
    return torch.cat(parts).to(device).contiguous()


def flat_to_torch(flat: torch.Tensor, obs_dim: int, hidden: int, layers: int, n_act: int) -> list[torch.Tensor]:
    """Inverse of flat_from_torch -> tensors in state_dict_keys(layers) order (on flat's device)."""
    k0 = (obs_dim + 31) // 32 * 32
    f, off = flat.detach(), 0

    def take(k_pad, n_pad, k, n):
        nonlocal off
        m = f[off:off + (k_pad + 1) * n_pad].reshape(k_pad + 1, n_pad)
        off += (k_pad + 1) * n_pad
        return m[:k, :n].t().contiguous(), m[k_pad, :n].clone()

    fc1 = take(k0, hidden, obs_dim, hidden)
    lstm = []
    for _ in range(layers):
        (w_ih, b_ih), (w_hh, b_hh) = take(hidden, 4 * hidden, hidden, 4 * hidden), take(hidden, 4 * hidden, hidden, 4 * hidden)
        lstm += [w_ih, w_hh, b_ih, b_hh]
    fc2 = take(hidden, HEAD, hidden, n_act)
    return lstm + [*fc1, *fc2]


def gather_stacked_obs(obs_rows: torch.Tensor, buffer: DeviceReplayBuffer, index, stack_num: int) -> torch.Tensor:
    """buffer.get(index, "obs") for vector observations: float32 [I, stack_num, obs_dim], oldest step first
    (buffer_base.py:586-596)."""
    index = _i64_dev(index, buffer.device).reshape(-1)
    rows = stack_indices(buffer, index, stack_num) if stack_num > 1 else index.reshape(-1, 1)
    out = gather_rows(obs_rows, rows.reshape(-1))
    return out.reshape(index.numel(), stack_num, -1)


class RowsReplay(C.Structure):
    """struct ts_rows_replay (include/tsengine.h)."""

    _fields_ = [("offset", C.c_void_p), ("E", C.c_int64), ("lengths", C.c_void_p), ("last_index", C.c_void_p),
                ("done", C.c_void_p), ("terminated", C.c_void_p), ("rew", C.c_void_p), ("obs_rows", C.c_void_p),
                ("obs_next_rows", C.c_void_p), ("act_col", C.c_void_p), ("slots", C.c_int64)]


def gather_stacked_obs_pair(obs_rows: torch.Tensor, buffer: DeviceReplayBuffer, index, n_step: int, stack_num: int,
                            obs_next_rows: torch.Tensor | None = None, act_col: torch.Tensor | None = None):
    """(buffer.get(index, "obs"), the stacked obs_next `_target_q` reads n steps on, batch.act or None) from ONE launch
    (ts_stacked_rows_pair; buffer_base.py:586-596, 624-626, algorithm_base.py:772-791), each observation tensor float32
    [I, stack_num, obs_dim].  None when the layout is outside the kernel's (float32 contiguous [slots, obs_dim] rows, int64
    action column, stack_num <= 16): the caller then takes the index kernels + gather_stacked_obs."""

    def rows_ok(t):
        return t.dim() == 2 and t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda

    if (not rows_ok(obs_rows) or stack_num > 16 or os.environ.get("TS_DRQN_NO_PAIR")
            or (obs_next_rows is not None and (not rows_ok(obs_next_rows) or obs_next_rows.shape != obs_rows.shape))
            or (act_col is not None and not (act_col.dim() == 1 and act_col.dtype == torch.int64 and act_col.is_contiguous()))):
        return None
    index = _i64_dev(index, buffer.device).reshape(-1)
    b, d = index.numel(), obs_rows.shape[1]
    obs = torch.empty((b, stack_num, d), dtype=torch.float32, device=obs_rows.device)
    obs_next = torch.empty_like(obs)
    act = torch.empty(b, dtype=torch.int64, device=obs_rows.device) if act_col is not None else None
    _lib.check(_lib.load().ts_stacked_rows_pair(
        _lib.ptr(obs_rows), _lib.ptr(obs_next_rows), _lib.i64(obs_rows.shape[0]), _lib.i64(d), _lib.ptr(index), _lib.i64(b),
        _lib.i64(n_step), _lib.i64(stack_num), _lib.ptr(buffer.offset), _lib.i64(buffer.buffer_num), _lib.ptr(buffer.done),
        _lib.ptr(buffer.last_index), _lib.ptr(buffer.lengths), _lib.ptr(act_col), _lib.ptr(obs), _lib.ptr(obs_next),
        _lib.ptr(act), _lib.current_stream(obs_rows.device)))
    return obs, obs_next, act


def replay_prepare(eng, buffer: DeviceReplayBuffer, obs_rows: torch.Tensor, stack_num: int, act_col: torch.Tensor,
                   obs_next_rows: torch.Tensor | None = None):
    """`prepare` callable of dqn.ReplayStream for RecurrentDQNEngine: what the next batch needs from the replay buffer alone
    -> ((obs, obs_next, act), n-step coefficients)."""
    cfg = eng.cfg

    def prepare(idx):
        pair = gather_stacked_obs_pair(obs_rows, buffer, idx, cfg.n_step, stack_num, obs_next_rows, act_col)
        if pair is None:
            after = nstep_indices(buffer, idx, cfg.n_step)
            nxt = (gather_stacked_obs(obs_rows, buffer, buffer.next(after), stack_num) if obs_next_rows is None
                   else gather_stacked_obs(obs_next_rows, buffer, after, stack_num))
            pair = (gather_stacked_obs(obs_rows, buffer, idx, stack_num), nxt, act_col[idx])
        return pair, nstep_coefficients(buffer, idx, cfg.gamma, cfg.n_step)

    return prepare


class RecurrentDQNEngine:
    """State of one DRQN learner on one GPU: flat parameters, lagged copy, Adam moments, counters."""

    def __init__(self, obs_dim: int, hidden: int, layers: int, n_act: int, flat_params: torch.Tensor, cfg: DQNConfig):
        if not flat_params.is_cuda:
            raise RuntimeError("RecurrentDQNEngine needs parameters on an MI355X (no CPU fallback)")
        self.obs_dim, self.hidden, self.layers, self.n_act, self.cfg = obs_dim, hidden, layers, n_act, cfg
        self.lay = layout(obs_dim, hidden, layers, n_act)
        self.P = self.lay["count"]
        if flat_params.numel() != self.P:
            raise ValueError(f"expected {self.P} parameters, got {flat_params.numel()}")
        self.device = flat_params.device
        self.params = flat_params.detach().float().contiguous().clone()
        self.params_old = self.params.clone() if cfg.target_update_freq > 0 else None    # dqn.py:240-246
        self.adam_m, self.adam_v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        self.adam_step = 0
        self.iter = 0
        self._ws = _lib.default_workspace(self.device.index or 0)
        self._streams = {}
        self._learn = None             # learn_step's scratch, replay view and second workspace
        self._pre = None               # (obs tensor, cache pointer, done event, parameter state, B, T) of a prefetched forward pass
        self._cache = None

    def _side(self, which: int) -> torch.cuda.Stream:
        """The workspace's side stream `which` (ts_workspace_side_stream) as a torch stream: independent passes go there
        instead of on streams of our own (four hardware queues)."""
        st = self._streams.get(which)
        if st is None:
            h = C.c_void_p()
            _lib.check(_lib.load().ts_workspace_side_stream(self._ws.handle, C.c_int(which), C.byref(h)))
            st = self._streams[which] = torch.cuda.ExternalStream(h.value, device=self.device)
        return st

    def _dims(self):
        return _lib.i64(self.obs_dim), _lib.i64(self.hidden), _lib.i64(self.layers), _lib.i64(self.n_act)

    def _obs(self, obs) -> torch.Tensor:
        obs = torch.as_tensor(obs, device=self.device).to(torch.float32)
        if obs.dim() == 2:
            obs = obs.unsqueeze(1)                                   # evaluation mode, common.py:425-426
        if obs.dim() != 3 or obs.shape[2] != self.obs_dim:
            raise ValueError(f"obs must be [B, T, {self.obs_dim}] or [B, {self.obs_dim}]")
        return obs.contiguous()

    # -- Recurrent.forward + DiscreteQLearningPolicy.forward ---------------------------------------------------------------
    def forward(self, obs, state=None, params: torch.Tensor | None = None, want_state: bool = False):
        """-> (q float32[B, A], act int64[B][, (hidden, cell) each float32[B, L, H] as the reference carries them])."""
        obs = self._obs(obs)
        b, t = obs.shape[:2]
        q = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device)
        act = torch.empty(b, dtype=torch.int64, device=self.device)
        h_in = c_in = h_out = c_out = None
        if state is not None:
            h_in, c_in = (torch.as_tensor(s, device=self.device).to(torch.float32).transpose(0, 1).contiguous() for s in state)
            if h_in.shape != (self.layers, b, self.hidden) or c_in.shape != h_in.shape:
                raise ValueError("state tensors must be [B, layers, hidden]")
        if want_state:
            h_out = torch.empty((self.layers, b, self.hidden), dtype=torch.float32, device=self.device)
            c_out = torch.empty_like(h_out)
        p = self.params if params is None else params
        ws = _lib.default_workspace(self.device.index or 0)          # the CURRENT stream's scratch (target_q runs two passes at once)
        _lib.check(_lib.load().ts_rnnq_forward(
            ws.handle, _lib.ptr(p), *self._dims(), _lib.ptr(obs), _lib.i64(b), _lib.i64(t), _lib.ptr(h_in), _lib.ptr(c_in),
            _lib.ptr(q), _lib.ptr(act), _lib.ptr(h_out), _lib.ptr(c_out), _lib.current_stream(self.device)))
        if want_state:
            return q, act, (h_out.transpose(0, 1).contiguous(), c_out.transpose(0, 1).contiguous())
        return q, act

    # -- DQN._target_q ---------------------------------------------------------------------------------------------------------
    def target_q(self, obs_next) -> torch.Tensor:
        """Both passes on obs_next (the lagged network's on the workspace's side stream) and the arg-max / gather in one call
        (ts_rnnq_target_q_fused)."""
        obs_next = self._obs(obs_next)
        b, t = obs_next.shape[:2]
        out = torch.empty(b, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().ts_rnnq_target_q_fused(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.params_old), *self._dims(), _lib.ptr(obs_next), _lib.i64(b),
            _lib.i64(t), C.c_int(int(self.cfg.is_double)), _lib.ptr(out), _lib.current_stream(self.device)))
        return out

    def target_returns(self, obs_next, coef) -> torch.Tensor:
        """`preprocess` for a caller that holds the stacked obs_next and the n-step coefficients (returns.nstep_coefficients):
        `_target_q` with float(double(target_q * mask) * gamma^n + sum gamma^k r) in its last kernel (ts_rnnq_target_returns)."""
        obs_next = self._obs(obs_next)
        b, t = obs_next.shape[:2]
        mask, gpow, mc = coef
        out = torch.empty(b, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().ts_rnnq_target_returns(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.params_old), *self._dims(), _lib.ptr(obs_next), _lib.i64(b),
            _lib.i64(t), C.c_int(int(self.cfg.is_double)), _lib.ptr(mask), _lib.ptr(gpow), _lib.ptr(mc), _lib.ptr(out),
            _lib.current_stream(self.device)))
        return out

    # -- the update's own forward pass, ahead of time -----------------------------------------------------------------------------
    def prefetch_forward(self, obs) -> torch.Tensor:
        """Q_online(batch.obs) of the coming `update_with_batch(obs, ...)` on the workspace's second side stream, beside the two
        obs_next passes of `_target_q` (as DQNEngine.prefetch_forward).  Returns the tensor to hand to `update_with_batch`
        (the SAME object: the cached activations are used only then, and only while the parameters are unchanged)."""
        lib = _lib.load()
        lib.ts_rnnq_cache_bytes.restype = C.c_int64
        obs = self._obs(obs)
        b, t = obs.shape[:2]
        need = int(lib.ts_rnnq_cache_bytes(*self._dims(), _lib.i64(b), _lib.i64(t)))
        if need <= 0:
            return obs
        if self._cache is None or self._cache.numel() < need + 256:
            self._cache = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        cache_ptr = C.c_void_p((self._cache.data_ptr() + 255) & ~255)
        main, side = torch.cuda.current_stream(self.device), self._side(1)
        ready, done = torch.cuda.Event(), torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            ws = _lib.default_workspace(self.device.index or 0)
            _lib.check(lib.ts_rnnq_forward_cache(ws.handle, _lib.ptr(self.params), *self._dims(), _lib.ptr(obs), _lib.i64(b),
                                                 _lib.i64(t), cache_ptr, _lib.i64(need), _lib.current_stream(self.device)))
            done.record(side)
        obs.record_stream(side)
        self._pre = (obs, cache_ptr, done, (self.params._version, self.adam_step), b, t)
        return obs

    def preprocess_with_obs(self, buffer: DeviceReplayBuffer, obs_rows: torch.Tensor, indices, stack_num: int,
                            obs_next_rows: torch.Tensor | None = None, prefetch: bool = True, pair=None, coef=None):
        """-> (batch.obs float32[I, T, obs_dim], returns float32[I]): the batch's own stacked observations first, their forward
        pass started on a side stream (prefetch_forward), then `_target_q` and the n-step returns.  Both stacked gathers come
        from one launch (gather_stacked_obs_pair) and the returns from the target passes' last kernel (target_returns) where
        the layout allows; `pair` / `coef`: those results when the caller already holds them (dqn.ReplayStream with
        drqn.replay_prepare)."""
        if pair is None:
            pair = gather_stacked_obs_pair(obs_rows, buffer, indices, self.cfg.n_step, stack_num, obs_next_rows)
        if pair is None:
            obs = self._obs(gather_stacked_obs(obs_rows, buffer, indices, stack_num))
            if prefetch:
                obs = self.prefetch_forward(obs)
            return obs, self.preprocess(buffer, obs_rows, indices, stack_num, obs_next_rows)
        obs = self._obs(pair[0])
        if prefetch:
            obs = self.prefetch_forward(obs)
        if coef is None:
            coef = nstep_coefficients(buffer, indices, self.cfg.gamma, self.cfg.n_step)
        return obs, self.target_returns(pair[1], coef)

    # -- sample + preprocess + update in one call (uniform device-resident buffer) ---------------------------------------------
    def learn_step(self, buffer: DeviceReplayBuffer, obs_rows: torch.Tensor, act_col: torch.Tensor, batch_size: int,
                   stack_num: int, seed, obs_next_rows: torch.Tensor | None = None):
        """OffPolicyAlgorithm.update (algorithm_base.py:583-631) on a buffer without priorities as ONE library call
        (ts_rnnq_learn_step): indices = buffer.sample_indices(batch_size, seed=seed), then what `preprocess_with_obs` and
        `update_with_batch` do -- the same kernels on the same values -- with the batch of seed (key, counter + 1) prepared
        beside the update for the next call.  -> (loss float32[1], td_error float32[B]).
        seed = (key, counter), counter advancing by one per call; call `learn_reset()` after writing to the buffer (the
        prepared batch predates the write).  float32 contiguous rows / int64 actions only (no fallback: use the separate
        calls otherwise)."""
        lib = _lib.load()
        key, counter = int(seed[0]) & (2**64 - 1), int(seed[1]) & (2**64 - 1)
        b, t = int(batch_size), int(stack_num)
        st = self._learn
        ident = (id(buffer), obs_rows.data_ptr(), act_col.data_ptr(), None if obs_next_rows is None else obs_next_rows.data_ptr(), b, t)
        if st is None or st["ident"] != ident:
            for r in (obs_rows, obs_next_rows):
                if r is not None and not (r.is_cuda and r.dim() == 2 and r.dtype == torch.float32 and r.is_contiguous()
                                          and r.shape[1] == self.obs_dim):
                    raise ValueError(f"learn_step: observation rows must be float32 contiguous [slots, {self.obs_dim}] on the device")
            if not (act_col.is_cuda and act_col.dim() == 1 and act_col.dtype == torch.int64 and act_col.is_contiguous()):
                raise ValueError("learn_step: actions must be an int64 contiguous device column")
            if len(buffer) == 0:                             # buffer_base.py:512-513
                raise ValueError("learn_step: empty buffer")
            lib.ts_rnnq_learn_scratch_bytes.restype = C.c_int64
            need = int(lib.ts_rnnq_learn_scratch_bytes(*self._dims(), _lib.i64(b), _lib.i64(t)))
            if need <= 0:
                raise ValueError("learn_step: unsupported network / batch dimensions")
            scratch = torch.zeros(need + 256, dtype=torch.uint8, device=self.device)
            view = RowsReplay(_lib.ptr(buffer.offset), buffer.buffer_num, _lib.ptr(buffer.lengths), _lib.ptr(buffer.last_index),
                              _lib.ptr(buffer.done), _lib.ptr(buffer.terminated), _lib.ptr(buffer.rew), _lib.ptr(obs_rows),
                              _lib.ptr(obs_next_rows), _lib.ptr(act_col), obs_rows.shape[0])
            st = self._learn = {"ident": ident, "scratch": scratch, "ptr": C.c_void_p((scratch.data_ptr() + 255) & ~255),
                                "bytes": _lib.i64(need), "view": view, "aux": _lib.aux_workspace(self.device.index or 0), "last": None,
                                "keep": (buffer, obs_rows, act_col, obs_next_rows), "B": _lib.i64(b), "T": _lib.i64(t),
                                }
        cfg = self.cfg
        sync = self.params_old is not None and self.iter % cfg.target_update_freq == 0    # dqn.py:283-285, applied inside the call
        self.iter += 1
        self.adam_step += 1
        self._pre = None
        td = torch.empty(b, dtype=torch.float32, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        hp = st.get("hp")
        if hp is None or st.get("hp_of") != (cfg.lr, cfg.betas, cfg.adam_eps, cfg.huber_delta, cfg.max_grad_norm):
            hp = st["hp"] = cfg.to_c()
            st["hp_of"] = (cfg.lr, cfg.betas, cfg.adam_eps, cfg.huber_delta, cfg.max_grad_norm)
        prepared = st["last"] == (key, (counter - 1) & (2**64 - 1))
        _lib.check(lib.ts_rnnq_learn_step(
            self._ws.handle, st["aux"].handle, _lib.ptr(self.params), _lib.ptr(self.params_old), C.c_int(int(sync)), _lib.ptr(self.adam_m),
            _lib.ptr(self.adam_v), _lib.i64(self.adam_step), *self._dims(), C.byref(st["view"]), st["B"], st["T"],
            _lib.i64(cfg.n_step), _lib.f64(cfg.gamma), C.c_int(int(cfg.is_double)), C.byref(hp), C.c_uint64(key),
            C.c_uint64(counter), C.c_int(int(prepared)), st["ptr"], st["bytes"], _lib.ptr(td), _lib.ptr(loss), None,
            _lib.current_stream(self.device)))
        st["last"] = (key, counter)
        return loss, td

    def learn_reset(self) -> None:
        """Drops the batch `learn_step` prepared ahead of time (call it after transitions were written to the buffer)."""
        if self._learn is not None:
            self._learn["last"] = None

    # -- DQN._preprocess_batch ---------------------------------------------------------------------------------------------
    def preprocess(self, buffer: DeviceReplayBuffer, obs_rows: torch.Tensor, indices, stack_num: int,
                   obs_next_rows: torch.Tensor | None = None) -> torch.Tensor:
        """n-step returns float32[I] with target_q_fn = _target_q (dqn.py:257-275); without stored obs_next, s_{t+n} is read at
        next(indices_after_n) (buffer_base.py:624-626)."""

        def tq_fn(buf, after):
            if obs_next_rows is None:
                return self.target_q(gather_stacked_obs(obs_rows, buf, buf.next(after), stack_num))
            return self.target_q(gather_stacked_obs(obs_next_rows, buf, after, stack_num))

        class _B:
            pass

        return compute_nstep_return(_B(), buffer, indices, tq_fn, self.cfg.gamma, self.cfg.n_step).returns.reshape(-1)

    # -- the two halves of an update, for the data-parallel path (tianshou_amd.distributed.DataParallelDQN) --------------------
    def gradient(self, obs, act, returns, weight, grad_out: torch.Tensor):
        """Periodic target sync + forward + loss + backward through time; grad_out[:P] = d loss / d params (mean over THIS
        batch), no optimizer step.  -> (loss, td_error)."""
        return self.update_with_batch(obs, act, returns, weight, grad_out=grad_out, apply=False)

    def apply_gradient(self, grad: torch.Tensor) -> None:
        """clip_grad_norm_ + Adam on a flat gradient (algorithm_base.py:496-500)."""
        cfg = self.cfg
        self.adam_step += 1
        _lib.check(_lib.load().ts_adam_step(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v), _lib.ptr(grad),
            _lib.i64(self.P), _lib.i64(self.adam_step), _lib.f64(cfg.lr), _lib.f64(cfg.betas[0]),
            _lib.f64(cfg.betas[1]), _lib.f64(cfg.adam_eps), _lib.f64(cfg.max_grad_norm or 0.0),
            _lib.current_stream(self.device)))

    # -- DQN._update_with_batch ----------------------------------------------------------------------------------------------
    def update_with_batch(self, obs, act, returns, weight=None, grad_out: torch.Tensor | None = None, apply: bool = True):
        """-> (loss float32[1] device tensor, td_error float32[B]); td_error is the new batch.weight."""
        cfg = self.cfg
        if self.params_old is not None and self.iter % cfg.target_update_freq == 0:    # dqn.py:283-285
            full_parameter_update(self.params_old, self.params)
        self.iter += 1
        params_state = (self.params._version, self.adam_step)          # what a prefetched forward pass was computed with
        pre, self._pre = self._pre, None
        cached = pre is not None and pre[0] is obs
        obs = self._obs(obs)
        b, t = obs.shape[:2]
        cached = cached and pre[3] == params_state and pre[4] == b and pre[5] == t
        act = _i64_dev(act, self.device).reshape(-1)
        returns = torch.as_tensor(returns, device=self.device).to(torch.float32).reshape(-1).contiguous()
        if act.numel() != b or returns.numel() != b:
            raise ValueError("act / returns length mismatch")
        if weight is not None:
            weight = torch.as_tensor(weight, device=self.device).to(torch.float32).reshape(-1).contiguous()
        td = torch.empty(b, dtype=torch.float32, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        if apply:
            self.adam_step += 1
        hp = cfg.to_c(grad_only=not apply)
        if cached:
            torch.cuda.current_stream(self.device).wait_event(pre[2])          # the prefetched activations are complete
            _lib.check(_lib.load().ts_rnnq_update_cached(
                self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v),
                _lib.i64(max(self.adam_step, 1)), *self._dims(), _lib.ptr(obs), _lib.ptr(act), _lib.ptr(returns), _lib.ptr(weight),
                _lib.i64(b), _lib.i64(t), C.byref(hp), pre[1], _lib.ptr(td), _lib.ptr(loss), _lib.ptr(grad_out),
                _lib.current_stream(self.device)))
            return loss, td
        _lib.check(_lib.load().ts_rnnq_update(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v), _lib.i64(max(self.adam_step, 1)),
            *self._dims(), _lib.ptr(obs), _lib.ptr(act), _lib.ptr(returns), _lib.ptr(weight), _lib.i64(b), _lib.i64(t),
            C.byref(hp), _lib.ptr(td), _lib.ptr(loss), _lib.ptr(grad_out), _lib.current_stream(self.device)))
        return loss, td
