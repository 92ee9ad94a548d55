"""Rainbow learn() path on the MI355X engine.

Mirrors, on device tensors:
    RainbowNet.forward                       tianshou/env/atari/atari_network.py:189-208 (NoisyLinear, dueling, softmax)
    C51Policy.compute_q_value / argmax       tianshou/algorithm/modelfree/c51.py:66-67, dqn.py:141
    C51._target_q / _target_dist             c51.py:120-141 (n-step via tianshou_amd.returns)
    RainbowDQN._update_with_batch            rainbow.py:93-101 -> c51.py:143-160 (+ periodic hard sync dqn.py:277-285,
                                             which copies the online network's noise along with its weights)
The NoisyLinear noise is supplied by the caller (`noise_from_torch`), drawn as the reference draws it
(rainbow.py:77-91, discrete.py:357-364).  No CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .buffer import DeviceReplayBuffer, _i64_dev
from .distq import DistQConfig, DistQHParams
from .dqn import _u8_flag
from .lagged import full_parameter_update
from .returns import compute_nstep_return

NOISY = ["Q.0", "Q.2", "V.0", "V.2"]
TIANSHOU_KEYS = ["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias"] + \
    [f"{m}.{t}" for m in NOISY for t in ("mu_W", "sigma_W", "mu_bias", "sigma_bias")]
NOISE_KEYS = [f"{m}.{t}" for m in NOISY for t in ("eps_p", "eps_q")]


def layout(c: int, h: int, w: int, n_act: int, n_atoms: int) -> dict:
    out = (C.c_int64 * 20)()
    _lib.check(_lib.load().ts_rainbow_layout(_lib.i64(c), _lib.i64(h), _lib.i64(w), _lib.i64(n_act), _lib.i64(n_atoms), out))
    v = [int(x) for x in out]
    return {"F": v[0], "ldq": v[1], "ldv": v[2], "count": v[3], "noise_count": v[4], "conv": v[5:8], "lin": v[8:12],
            "noise": v[12:20]}


def _hwc(c: int, h: int, w: int):
    """(oh3, ow3) of the conv trunk."""
    for k, s in ((8, 4), (4, 2), (3, 1)):
        h, w = (h - k) // s + 1, (w - k) // s + 1
    return h, w


def _lin_block(wt: torch.Tensor, b: torch.Tensor, n_pad: int, feat_hw=None) -> torch.Tensor:
    """NoisyLinear tensor pair ([out, in], [out]) -> matrix [in + 1, n_pad]; feat_hw: the input is the flattened conv output
    in torch's (c, h, w) order and has to become (h, w, c)."""
    wt, b = wt.detach().float().cpu(), b.detach().float().cpu()
    out_f, in_f = wt.shape
    if feat_hw is not None:
        oh, ow = feat_hw
        wt = wt.reshape(out_f, 64, oh, ow).permute(0, 2, 3, 1).reshape(out_f, in_f)
    m = torch.zeros((in_f + 1, n_pad), dtype=torch.float32)
    m[:in_f, :out_f], m[in_f, :out_f] = wt.t(), b
    return m.reshape(-1)


def flat_from_torch(t: list[torch.Tensor], c: int, h: int, w: int, n_act: int, n_atoms: int, device="cuda") -> torch.Tensor:
    """Tensors in TIANSHOU_KEYS order (RainbowNet state_dict without the eps buffers; also valid for Adam moments)."""
    lay = layout(c, h, w, n_act, n_atoms)
    hw = _hwc(c, h, w)
    f = lambda x: x.detach().float().cpu()  # noqa: E731
    parts = []
    for i in range(3):
        parts += [f(t[2 * i]).permute(2, 3, 1, 0).reshape(-1), f(t[2 * i + 1]).reshape(-1)]
    pads = [512, lay["ldq"], 512, lay["ldv"]]
    for i in range(4):
        mw, sw, mb, sb = t[6 + 4 * i: 10 + 4 * i]
        fh = hw if i in (0, 2) else None
        parts += [_lin_block(mw, mb, pads[i], fh), _lin_block(sw, sb, pads[i], fh)]
    return torch.cat(parts).to(device).contiguous()


def flat_to_torch(flat: torch.Tensor, c: int, h: int, w: int, n_act: int, n_atoms: int) -> list[torch.Tensor]:
    lay = layout(c, h, w, n_act, n_atoms)
    oh, ow = _hwc(c, h, w)
    out = []
    geo = [(c, 8, 32), (32, 4, 64), (64, 3, 64)]
    ends = lay["conv"][1:] + [lay["lin"][0]]
    for i, (ic, k, oc) in enumerate(geo):
        wb = flat[lay["conv"][i]:ends[i]].reshape(k * k * ic + 1, oc)
        out += [wb[:-1].reshape(k, k, ic, oc).permute(3, 2, 0, 1).contiguous(), wb[-1].clone()]
    dims = [(lay["F"], 512, 512), (512, n_act * n_atoms, lay["ldq"]), (lay["F"], 512, 512), (512, n_atoms, lay["ldv"])]
    for i, (fin, fout, pad) in enumerate(dims):
        n = (fin + 1) * pad
        pair = []
        for blk in (flat[lay["lin"][i]: lay["lin"][i] + n], flat[lay["lin"][i] + n: lay["lin"][i] + 2 * n]):
            m = blk.reshape(fin + 1, pad)
            wt = m[:fin, :fout].t().contiguous()
            if i in (0, 2):
                wt = wt.reshape(fout, oh, ow, 64).permute(0, 3, 1, 2).reshape(fout, fin).contiguous()
            pair.append((wt, m[fin, :fout].clone()))
        out += [pair[0][0], pair[1][0], pair[0][1], pair[1][1]]           # mu_W, sigma_W, mu_bias, sigma_bias
    return out


def noise_from_torch(t: list[torch.Tensor], c: int, h: int, w: int, n_act: int, n_atoms: int, device="cuda") -> torch.Tensor:
    """[Q.0.eps_p, Q.0.eps_q, Q.2.eps_p, ..., V.2.eps_q] (NOISE_KEYS order) -> the engine's noise vector."""
    lay = layout(c, h, w, n_act, n_atoms)
    oh, ow = _hwc(c, h, w)
    out = torch.zeros(lay["noise_count"], dtype=torch.float32)
    for i in range(8):
        v = torch.as_tensor(np.asarray(t[i]) if not isinstance(t[i], torch.Tensor) else t[i]).detach().float().cpu().reshape(-1)
        if i in (0, 4):                                     # eps_p of the layers fed by the conv features
            v = v.reshape(64, oh, ow).permute(1, 2, 0).reshape(-1)
        out[lay["noise"][i]: lay["noise"][i] + v.numel()] = v
    return out.to(device).contiguous()


class RainbowEngine:
    """State of one Rainbow learner on one GPU (hyper-parameters: DistQConfig with kind 'c51')."""

    def __init__(self, c: int, h: int, w: int, n_act: int, flat_params: torch.Tensor, noise: torch.Tensor, cfg: DistQConfig):
        if not flat_params.is_cuda:
            raise RuntimeError("RainbowEngine needs parameters on an MI355X (no CPU fallback)")
        self.c, self.h, self.w, self.n_act, self.cfg = c, h, w, n_act, cfg
        self.lay = layout(c, h, w, n_act, cfg.n_atoms)
        if flat_params.numel() != self.lay["count"] or noise.numel() != self.lay["noise_count"]:
            raise ValueError("flat parameter / noise vectors do not match ts_rainbow_layout")
        self.P = self.lay["count"]
        self.device = flat_params.device
        self.params = flat_params.detach().float().contiguous().clone()
        self.noise = noise.detach().float().to(self.device).contiguous().clone()
        lagged = cfg.target_update_freq > 0
        self.params_old = self.params.clone() if lagged else None                 # deepcopy: weights and noise
        self.noise_old = self.noise.clone() if lagged else None
        self.adam_m, self.adam_v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        self.adam_step = 0
        self.iter = 0
        self.support = torch.linspace(cfg.v_min, cfg.v_max, cfg.n_atoms).to(self.device).contiguous()   # c51.py:61-64
        self._ws = _lib.default_workspace(self.device.index or 0)

    def _dims(self):
        return (_lib.i64(self.c), _lib.i64(self.h), _lib.i64(self.w), _lib.i64(self.n_act), _lib.i64(self.cfg.n_atoms),
                _lib.ptr(self.support))

    def _check_obs(self, obs: torch.Tensor) -> torch.Tensor:
        if tuple(obs.shape[1:]) != (self.h, self.w, self.c) or obs.dtype not in (torch.float32, torch.uint8):
            raise ValueError(f"obs must be float32 or uint8 [B, {self.h}, {self.w}, {self.c}] (NHWC)")
        return obs.contiguous()

    def set_noise(self, noise: torch.Tensor, noise_old: torch.Tensor | None = None) -> None:
        """The result of RainbowDQN._sample_noise on the online (and the lagged) network (rainbow.py:97-100)."""
        self.noise = self._own(noise, self.noise)
        if noise_old is not None and self.noise_old is not None:
            self.noise_old = self._own(noise_old, self.noise_old)

    def _own(self, new: torch.Tensor, cur: torch.Tensor) -> torch.Tensor:
        """A float32 device vector of the right length is KEPT (the engine only reads it; the caller draws a fresh one per
        update) -- a 6 MB device-to-device copy through the runtime's blit kernel took 45 us, twice per update; anything
        else is converted into the engine's own buffer."""
        if (isinstance(new, torch.Tensor) and new.is_cuda and new.device == cur.device and new.dtype == torch.float32
                and new.is_contiguous() and new.numel() == cur.numel()):
            return new.reshape(cur.shape)
        cur.copy_(torch.as_tensor(new).to(self.device).reshape(cur.shape))
        return cur

    def forward(self, obs_nhwc: torch.Tensor, training: bool = True, want_dist: bool = True):
        """-> (probabilities float32[B, A, N] or None, q float32[B, A], act int64[B]); training=False: eval-mode layers."""
        obs_nhwc = self._check_obs(obs_nhwc)
        b = obs_nhwc.shape[0]
        dist = torch.empty((b, self.n_act, self.cfg.n_atoms), dtype=torch.float32, device=self.device) if want_dist else None
        q = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device)
        act = torch.empty(b, dtype=torch.int64, device=self.device)
        _lib.check(_lib.load().ts_rainbow_forward(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.noise if training else None), *self._dims(),
            _lib.ptr(obs_nhwc), _u8_flag(obs_nhwc), _lib.i64(b), _lib.ptr(dist), _lib.ptr(q), _lib.ptr(act),
            _lib.current_stream(self.device)))
        return dist, q, act

    def next_dist(self, obs_next_nhwc: torch.Tensor) -> torch.Tensor:
        obs_next_nhwc = self._check_obs(obs_next_nhwc)
        b = obs_next_nhwc.shape[0]
        out = torch.empty((b, self.cfg.n_atoms), dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().ts_rainbow_next_dist(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.noise), _lib.ptr(self.params_old), _lib.ptr(self.noise_old),
            *self._dims(), _lib.ptr(obs_next_nhwc), _u8_flag(obs_next_nhwc), _lib.i64(b), _lib.ptr(out),
            _lib.current_stream(self.device)))
        return out

    def preprocess(self, buffer: DeviceReplayBuffer, indices) -> torch.Tensor:
        """n-step returns of the support, float32[I, N] (C51._target_q, c51.py:120-121)."""

        class _B:
            pass

        fn = lambda buf, after: self.support.repeat(after.numel(), 1)  # noqa: E731
        return compute_nstep_return(_B(), buffer, indices, fn, self.cfg.gamma, self.cfg.n_step).returns

    support_returns = preprocess          # the name distq.replay_prepare calls (same as DistQEngine.support_returns)

    def wait_td(self, stream: torch.cuda.Stream) -> None:
        """`stream` waits for the new priorities and the loss of the last `update_with_batch`, not for its backward pass and
        Adam step (ts_dqn_wait_td; see dqn.ReplayStream)."""
        _lib.check(_lib.load().ts_dqn_wait_td(self._ws.handle, C.c_void_p(stream.cuda_stream)))

    def update_with_batch(self, obs_nhwc, act, returns, obs_next_nhwc, weight=None, grad_out: torch.Tensor | None = None,
                          apply: bool = True, want_target: bool = False):
        """Call set_noise first (rainbow.py:97-100).  -> (loss float32[1], new batch.weight float32[B][, target_dist])."""
        cfg = self.cfg
        if apply:
            if self.params_old is not None and self.iter % cfg.target_update_freq == 0:    # dqn.py:283-285
                full_parameter_update(self.params_old, self.params)
                self.noise_old.copy_(self.noise)              # load_state_dict copies the eps buffers too
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
        nd = self.next_dist(obs_next_nhwc)
        prio = torch.empty(b, dtype=torch.float32, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        tgt = torch.empty((b, n), dtype=torch.float32, device=self.device) if want_target else None
        hp = DistQHParams(-1.0 if not apply else cfg.lr, cfg.betas[0], cfg.betas[1], cfg.adam_eps, cfg.max_grad_norm or 0.0,
                          cfg.v_min, cfg.v_max)
        _lib.check(_lib.load().ts_rainbow_update(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v),
            _lib.i64(max(self.adam_step, 1)), _lib.ptr(self.noise), *self._dims(), _lib.ptr(obs_nhwc), _u8_flag(obs_nhwc),
            _lib.ptr(act), _lib.ptr(returns), _lib.ptr(nd), _lib.ptr(weight), _lib.i64(b), C.byref(hp), _lib.ptr(prio),
            _lib.ptr(loss), _lib.ptr(tgt), _lib.ptr(grad_out), _lib.current_stream(self.device)))
        return (loss, prio, tgt) if want_target else (loss, prio)
