"""The collector's inference step on the engine.

`Collector._compute_action_policy_hidden` (tianshou/data/collector.py:707-772) calls, once per vector step,
    self.policy(obs_batch, hidden)       -> Policy.forward          (reinforce.py:167-192, dqn.py:101-143, sac.py:108-131)
    self.policy.map_action(act)          -> Algorithm.map_action    (algorithm_base.py:254-287)
`attach(policy, family, owner)` turns a policy object of the reference into an instance of a subclass of ITS OWN class
whose `forward` (and, for the Gaussian family, `map_action`) run on libtsengine's inference kernels
(`ts_ppo_policy_forward_bounded`, `ts_ppo_net_infer` + `ts_gauss_sample_map`, `ts_dqn_forward`,
`ts_sac_policy_forward_logits`) -- the inference halves of the training kernels, reading the engine's device-resident
parameter vector while an engine exists (no per-update write-back is needed for acting) and a flat copy of the torch
modules' parameters otherwise (before the first update, after `load_state_dict`, or for a policy that was pickled on its own).
Everything else of the class -- `compute_action`, `add_exploration_noise`, `map_action_inverse`, the `state_dict` keys --
is inherited unchanged; the returned batch has the reference's keys (`logits`, `act`, `state`, `dist` / `log_prob`).

There is no CPU path: the forward raises without a GPU.  A subclass is created per (reference class, family) and is
picklable (`__reduce__` rebuilds it from the reference class), so `torch.save(policy)` / `copy.deepcopy(policy)`
(highlevel/persistence.py:106) keep working.

Sampling noise: `sampling="device"` (default) draws N(0, 1) from the engine's counter-based generator
(`ts_normal_fill`, keyed by a private seed and a call counter: torch's global generator, which the reference consumes in
`dist.sample()`, is left untouched); `sampling="torch"` draws `torch.empty(n, A).normal_()` from torch's CPU generator --
the stream `Normal.sample()` consumes on a CPU-resident reference model (`torch.normal(mean, std)` = N(0, 1) draws scaled
and shifted), i.e. the seed-exact mode the fixture test replays.
"""
from __future__ import annotations

import weakref

import numpy as np
import torch

from . import _lib

_CLASSES: dict = {}
_BOUND = {None: 0, "": 0, "clip": 1, "tanh": 2}


def _rebuild(base_cls, family):
    cls = hip_policy_class(base_cls, family)
    return cls.__new__(cls)


class _HipForward:
    """Mixed in FRONT of the reference's policy class by `hip_policy_class`."""
    _hip_family = ""

    def __reduce__(self):
        self._hip_sync_owner()             # pickle / deepcopy / torch.save read the torch parameters
        state = {k: v for k, v in self.__dict__.items() if not k.startswith("_hip_rt_")}       # (run-time handles: weakrefs, caches)
        return _rebuild, (type(self).__mro__[2], self._hip_family), state

    def _hip_sync_owner(self) -> None:
        """An owner with `write_back="lazy"` keeps its updates in the engine until somebody reads the torch parameters: this
        policy's `state_dict()` / pickling are such readers (`_HipGlue.hip_sync`)."""
        owner = self._hip_owner()
        if owner is not None and hasattr(owner, "hip_sync"):
            owner.hip_sync()

    def state_dict(self, *args, **kwargs):
        self._hip_sync_owner()
        return super().state_dict(*args, **kwargs)

    # -- where the parameters come from -----------------------------------------------------------------------
    def _hip_owner(self):
        ref = self.__dict__.get("_hip_rt_owner")
        return ref() if ref is not None else None

    def _hip_device(self) -> torch.device:
        dev = self.__dict__.get("_hip_fwd_device")
        if dev is None:
            dev = torch.device("cuda")
        dev = torch.device(dev)
        if dev.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError(f"{type(self).__name__}.forward runs on libtsengine's HIP kernels and needs an MI355X "
                               "(device='cuda'); there is no CPU fallback")
        return dev

    def _hip_engine(self):
        """The owner's live engine, or None (never updated yet / dropped because somebody else wrote the torch parameters:
        `_HipGlue._hip_engine` compares the parameters' version counters on every access)."""
        owner = self._hip_owner()
        return None if owner is None else getattr(owner, "_hip_engine", None)

    def _hip_cached(self, modules, build):
        """`build()` (a flat device vector made from `modules`' parameters), cached until one of them is written
        (tensor version counters + storage pointers, the rule `_HipGlue` uses for its engine)."""
        key = tuple((p.data_ptr(), p._version) for m in modules for p in m.parameters())
        c = self.__dict__.get("_hip_rt_flat")
        if c is None or c[0] != key:
            c = self.__dict__["_hip_rt_flat"] = (key, build())
        return c[1]

    def _hip_noise(self, n: int, a: int, dev) -> torch.Tensor:
        if self.__dict__.get("_hip_sampling", "device") == "torch":
            return torch.empty(n, a, dtype=torch.float32).normal_().to(dev)
        from .buffer import normal_noise

        k = self.__dict__["_hip_rt_calls"] = self.__dict__.get("_hip_rt_calls", 0) + 1
        return normal_noise((n, a), int(self.__dict__.get("_hip_noise_seed", 0)), k, dev)

    def forward(self, batch, state=None, **kwargs):
        return _FAMILIES[self._hip_family](self, batch, state, **kwargs)


def hip_policy_class(base_cls, family: str):
    """The subclass of `base_cls` (a policy class of the reference, or its stand-in) for `family`."""
    key = (base_cls, family)
    cls = _CLASSES.get(key)
    if cls is None:
        ns = {"_hip_family": family, "__module__": __name__, "__doc__": f"{base_cls.__name__} whose forward runs on the HIP engine ({family})."}
        if family.startswith("gauss"):
            ns["map_action"] = _gauss_map_action
        cls = _CLASSES[key] = type("Hip" + base_cls.__name__, (_HipForward, base_cls), ns)
    return cls


def attach(policy, family: str, owner=None, *, device="cuda", sampling: str = "device", noise_seed: int | None = None, **spec):
    """In place: `policy` becomes an instance of `hip_policy_class(type(policy), family)`.  `owner`: the Hip* algorithm
    whose engine holds the live parameters (kept as a weak reference).  `spec`: the family's static description (shapes)."""
    if family not in _FAMILIES:
        raise ValueError(f"unknown policy family {family!r}")
    if sampling not in ("device", "torch"):
        raise ValueError("sampling must be 'device' or 'torch'")
    base = type(policy)
    if isinstance(policy, _HipForward):
        base = type(policy).__mro__[2]
    policy.__class__ = hip_policy_class(base, family)
    d = policy.__dict__
    d["_hip_fwd_device"] = str(device)
    d["_hip_sampling"] = sampling
    # (None: torch's seed -- READ, not consumed -- so that `torch.manual_seed` / `seed_everything` select the noise sequence
    # as they do for the reference's dist.sample(); NumPy's generator, which the collector shares, is not touched)
    d["_hip_noise_seed"] = int(torch.initial_seed() % (2**31 - 1)) if noise_seed is None else int(noise_seed)
    d["_hip_spec"] = dict(spec)
    d["_hip_rt_owner"] = weakref.ref(owner) if owner is not None else None
    return policy


def detach(policy):
    """Back to the reference's own class (the torch forward)."""
    if isinstance(policy, _HipForward):
        policy.__class__ = type(policy).__mro__[2]
        for k in [k for k in policy.__dict__ if k.startswith("_hip_")]:
            del policy.__dict__[k]
    return policy


def _obs_array(batch):
    obs = batch.obs
    return obs.obs if hasattr(obs, "obs") else obs


def _dev_f32(x, dev, cols=None) -> torch.Tensor:
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    t = t.to(device=dev, dtype=torch.float32)
    if cols is not None:
        t = t.reshape(t.shape[0], -1)
        if t.shape[1] != cols:
            raise ValueError(f"observation rows have {t.shape[1]} entries, the network reads {cols}")
    return t.contiguous()


def _box(policy, a: int, dev):
    """(bound method code, low, high) of Algorithm.map_action for a Box action space (algorithm_base.py:274-287); high is None
    when action_scaling is off or the space is not a Box."""
    space = getattr(policy, "action_space", None)
    bound = _BOUND[getattr(policy, "action_bound_method", None)]
    low = high = None
    if space is not None and hasattr(space, "low") and hasattr(space, "high"):
        if getattr(policy, "action_scaling", False):
            low = torch.as_tensor(np.broadcast_to(np.asarray(space.low, np.float32).reshape(-1), (a,)).copy(), device=dev)
            high = torch.as_tensor(np.broadcast_to(np.asarray(space.high, np.float32).reshape(-1), (a,)).copy(), device=dev)
    else:
        bound = 0                                            # not a Box: map_action leaves the action alone
    return bound, low, high


# ---------------------------------------------------------------------------------------------------------------------
# ProbabilisticActorPolicy over ContinuousActorProbabilistic (reinforce.py:167-192, continuous.py:220-238)
# ---------------------------------------------------------------------------------------------------------------------
def _gauss_params(policy):
    """The flat parameter vector the kernels read: the engine's while one is alive, else built from the actor's tensors."""
    eng = policy._hip_engine()
    if eng is not None:
        return eng.params
    spec, dev, actor, fam = policy._hip_spec, policy._hip_device(), policy.actor, policy._hip_family

    def build():
        sa = actor.state_dict()
        t = [sa[k].detach() for k in spec["actor_keys"]]
        if fam == "gauss":
            from .ppo import param_count

            flat_a = torch.cat([x.reshape(-1).float() for x in t]).to(dev)
            # the kernels stage both networks of the flat [actor | critic] vector; only the actor's half is read here
            return torch.cat([flat_a, torch.zeros(param_count(spec["obs_dim"], spec["act_dim"]) - flat_a.numel(), device=dev)]).contiguous()
        if fam == "gauss_wide":
            from . import npg as NG

            return NG.actor_flat_from_torch(t, spec["obs_dim"], spec["hidden"], spec["act_dim"], dev)
        from .ppo_wide import net_flat_from_tensors

        return net_flat_from_tensors(t, spec["obs_dim"], list(spec["hidden"]), spec["act_dim"], dev,
                                     layer_norm=spec.get("ln_eps") is not None)
    return policy._hip_cached([actor], build)


def _gauss_forward(policy, batch, state=None, **kwargs):
    """ProbabilisticActorPolicy.forward (reinforce.py:167-192): mu (bounded by max_action * tanh for the reference's default
    actor), sigma = exp(sigma_param), act = dist.sample() = mu + sigma * noise (dist.mode with deterministic_eval outside a
    training step) -- plus the bounded / scaled action of Algorithm.map_action, from the same launch."""
    import ctypes as C

    from torch.distributions import Independent, Normal

    from . import ppo as P

    spec, dev, fam = policy._hip_spec, policy._hip_device(), policy._hip_family
    obs_dim, a = spec["obs_dim"], spec["act_dim"]
    obs = _dev_f32(_obs_array(batch), dev, obs_dim)
    n = obs.shape[0]
    params = _gauss_params(policy)
    deterministic = bool(getattr(policy, "deterministic_eval", False)) and not policy.is_within_training_step
    noise = None if deterministic else policy._hip_noise(n, a, dev)
    bound, low, high = _box(policy, a, dev)
    if fam == "gauss":
        act, mapped, mu = P.policy_forward(params, obs_dim, a, obs, noise, bound_method={0: None, 1: "clip", 2: "tanh"}[bound],
                                           low=low, high=high, max_action=spec.get("max_action"), want_mu=True)
        off = P.HIDDEN * obs_dim + P.HIDDEN + P.HIDDEN * P.HIDDEN + P.HIDDEN + a * P.HIDDEN + a      # a_sigma in the flat layout
        log_sigma = params[off:off + a]
    else:
        lib = _lib.load()
        ws = _lib.default_workspace(dev.index or 0)
        n_actor = spec["n_actor"]
        actor_flat = params[:n_actor]
        mu = torch.empty((n, a), dtype=torch.float32, device=dev)
        if fam == "gauss_wide":
            _lib.check(lib.ts_npg_infer(ws.handle, _lib.ptr(actor_flat), None, _lib.i64(obs_dim), _lib.i64(spec["hidden"]), _lib.i64(a),
                                        _lib.ptr(obs), None, _lib.i64(n), None, None, _lib.ptr(mu), _lib.current_stream(dev)))
        else:
            ln_eps = spec.get("ln_eps")                  # MLP(norm_layer=nn.LayerNorm) trunks: TS_NET_LAYERNORM
            na = _lib.NetDesc.make(obs_dim, list(spec["hidden"]), spec["activation"], _lib.NetDesc.LAYERNORM if ln_eps is not None else 0,
                                   max_action=spec.get("max_action") or 0.0, ln_eps=ln_eps or 0.0)
            _lib.check(lib.ts_ppo_net_infer(ws.handle, _lib.ptr(actor_flat), None, C.byref(na), None, _lib.i64(a), _lib.ptr(obs), None,
                                            _lib.i64(n), None, None, _lib.ptr(mu), _lib.current_stream(dev)))
        log_sigma = actor_flat[n_actor - 32:n_actor - 32 + a].contiguous()       # the vector's last block: log_sigma padded to 32
        act, mapped = torch.empty_like(mu), torch.empty_like(mu)
        _lib.check(lib.ts_gauss_sample_map(_lib.ptr(mu), _lib.ptr(noise), _lib.ptr(log_sigma), _lib.i64(n), _lib.i64(a), C.c_int(bound),
                                           _lib.ptr(low), _lib.ptr(high), _lib.ptr(act), _lib.ptr(mapped), _lib.current_stream(dev)))
    host = torch.stack([act, mapped]).cpu()                                   # one D2H per collector step
    act_h = host[0]
    policy.__dict__["_hip_rt_last"] = (act_h.data_ptr(), tuple(act_h.shape), host[1].numpy())
    sigma = log_sigma.reshape(1, a).exp().expand(n, a)                        # continuous.py:236-238
    dist = policy.dist_fn((mu, sigma)) if getattr(policy, "dist_fn", None) is not None else Independent(Normal(mu, sigma), 1)
    return type(batch)(logits=(mu, sigma), act=act_h, state=None, dist=dist)


def _gauss_map_action(self, act):
    """Algorithm.map_action (algorithm_base.py:254-287).  The kernel that sampled the action has already bounded and scaled it
    (ts_ppo_policy_forward_bounded / ts_gauss_sample_map): when `act` is the array the last forward returned, that result is
    handed out; any other input (a different array, a copy that was edited) takes the reference's own NumPy code."""
    last = self.__dict__.get("_hip_rt_last")
    if last is not None:
        a = act.detach() if isinstance(act, torch.Tensor) else act
        ptr = a.data_ptr() if isinstance(a, torch.Tensor) else (a.__array_interface__["data"][0] if isinstance(a, np.ndarray) else None)
        if ptr == last[0] and tuple(a.shape) == last[1]:
            return last[2].copy()
    return super(type(self), self).map_action(act)


# ---------------------------------------------------------------------------------------------------------------------
# DiscreteQLearningPolicy over DQNet (dqn.py:101-143, atari_network.py:60-122)
# ---------------------------------------------------------------------------------------------------------------------
def _q_forward(policy, batch, state=None, model=None, **kwargs):
    from . import dqn as D

    base_forward = type(policy).__mro__[2].forward
    if model is not None and model is not policy.model:      # the lagged network passed by the reference's own _target_q
        return base_forward(policy, batch, state, model=model, **kwargs)
    dev = policy._hip_device()
    n_act = policy._hip_spec["n_act"]
    obs = _obs_array(batch)
    t = obs if isinstance(obs, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(obs))
    if t.dim() != 4:
        raise ValueError(f"observations must be [n, c, h, w], got {tuple(t.shape)}")
    n, c, h, w = (int(x) for x in t.shape)                   # (the frame geometry is the environment's: DQNet(c, h, w, ...))
    eng = policy._hip_engine()
    if eng is not None:
        if (eng.c, eng.h, eng.w) != (c, h, w):
            raise ValueError(f"observations are [{c}, {h}, {w}], the engine was built for [{eng.c}, {eng.h}, {eng.w}]")
        params = eng.params
    else:
        params = policy._hip_cached([policy.model], lambda: D.flat_from_torch(
            [policy.model.state_dict()[k] for k in D.TIANSHOU_KEYS], c, h, w, n_act, dev))
    t = t.to(dev)
    if t.dtype == torch.uint8:                               # [n, c, h, w] planes -> NHWC bytes, converted inside conv1
        t = t.contiguous()
        planes = torch.arange(n * c, device=dev, dtype=torch.int64).reshape(n, c)
        x = torch.empty((n, h, w, c), dtype=torch.uint8, device=dev)
        _lib.check(_lib.load().ts_gather_planes_nhwc_u8(_lib.ptr(t), _lib.i64(n * c), _lib.i64(h * w), _lib.ptr(planes), _lib.i64(n),
                                                        _lib.i64(c), _lib.ptr(x), _lib.current_stream(dev)))
    else:
        x = t.to(torch.float32).permute(0, 2, 3, 1).contiguous()
    import ctypes as C

    q = torch.empty((n, n_act), dtype=torch.float32, device=dev)
    act = torch.empty(n, dtype=torch.int64, device=dev)
    ws = _lib.default_workspace(dev.index or 0)
    _lib.check(_lib.load().ts_dqn_forward(ws.handle, _lib.ptr(params), _lib.i64(c), _lib.i64(h), _lib.i64(w), _lib.i64(n_act),
                                          _lib.ptr(x), C.c_int(1 if x.dtype == torch.uint8 else 0), _lib.i64(n), _lib.ptr(q),
                                          _lib.ptr(act), _lib.current_stream(dev)))
    mask = getattr(batch.obs, "mask", None)
    if mask is not None:                                     # compute_q_value (dqn.py:145-151) on the [n, A] logits
        qm = policy.compute_q_value(q, mask)
        act_np = qm.argmax(dim=1).cpu().numpy()
    else:
        act_np = act.cpu().numpy()
    return type(batch)(logits=q, act=act_np, state=None)


# ---------------------------------------------------------------------------------------------------------------------
# SACPolicy (sac.py:108-131) over the mujoco_sac.py actor
# ---------------------------------------------------------------------------------------------------------------------
def _sac_forward(policy, batch, state=None, **kwargs):
    from torch.distributions import Independent, Normal

    from . import sac as S

    dev, spec = policy._hip_device(), policy._hip_spec
    obs_dim, a, hid, depth = spec["obs_dim"], spec["act_dim"], spec["hidden"], int(spec.get("depth", 2))
    eng = policy._hip_engine()
    if eng is not None:
        actor = eng.actor
    else:
        actor = policy._hip_cached([policy.actor], lambda: S.actor_flat_from_torch(
            [policy.actor.state_dict()[k] for k in S.actor_keys(depth)], obs_dim, a, dev, hidden=hid))
    obs = _dev_f32(_obs_array(batch), dev, obs_dim)
    n = obs.shape[0]
    deterministic = bool(getattr(policy, "deterministic_eval", False)) and not policy.is_within_training_step
    noise = None if deterministic else policy._hip_noise(n, a, dev)
    act = torch.empty((n, a), dtype=torch.float32, device=dev)
    logp = torch.empty(n, dtype=torch.float32, device=dev)
    mu, sigma = torch.empty_like(act), torch.empty_like(act)
    ws = _lib.default_workspace(dev.index or 0)
    S.use_hidden(ws, hid, depth, float(spec.get("max_action") or 0.0), spec.get("activation", "relu"))
    _lib.check(_lib.load().ts_sac_policy_forward_logits(
        ws.handle, _lib.ptr(actor), _lib.ptr(obs), _lib.ptr(noise), _lib.i64(n), _lib.i64(obs_dim), _lib.i64(a), _lib.ptr(act),
        _lib.ptr(logp), _lib.ptr(mu), _lib.ptr(sigma), _lib.current_stream(dev)))
    return type(batch)(logits=(mu, sigma), act=act, state=None, dist=Independent(Normal(loc=mu, scale=sigma), 1),
                       log_prob=logp.unsqueeze(-1))


_FAMILIES = {"gauss": _gauss_forward, "gauss_net": _gauss_forward, "gauss_wide": _gauss_forward, "q": _q_forward,
             "sac": _sac_forward}
