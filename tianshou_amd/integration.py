"""Binding a Tianshou maintainer adds: `HipPPO`, a `PPO` subclass whose learn() hooks run on the
MI355X engine.  Needs `tianshou` importable (it is not on the GPU test box; there the same engine
is driven through `tianshou_amd.ppo.PPOEngine` directly).

Overrides exactly the two hooks `Algorithm._update` calls (algorithm_base.py:622-627):
    _preprocess_batch(batch, buffer, indices) -> batch          (ppo.py:146-162)
    _update_with_batch(batch, batch_size, repeat) -> A2CTrainingStats   (ppo.py:164-224)
keeping the Policy / Algorithm API, the Batch fields (`v_s`, `returns`, `adv`, `logp_old`, `act`),
the stats dataclasses and `state_dict()` (parameters and Adam moments are copied back into the
torch modules / optimizer after every update()).  Supported net: the MuJoCo actor-critic of
examples/mujoco/mujoco_ppo.py (Net[64,64] tanh, ContinuousActorProbabilistic(unbounded=True) with a
state-independent sigma, ContinuousCritic); anything else raises at construction.
"""
from __future__ import annotations

import os
import weakref

import numpy as np
import torch

from .checkpoint import adam_state, params_by_keys, store_adam_state
from .ppo import PPOConfig, PPOEngine, flat_from_modules, flat_to_modules, TIANSHOU_ACTOR_KEYS, TIANSHOU_CRITIC_KEYS


def _algorithm_state_loaded(module, incompatible_keys) -> None:
    """load_state_dict post-hook of the Hip* algorithms (module level, so that the algorithm stays picklable)."""
    module._hip_invalidate()


# Sub-modules of an algorithm with lazy write-back -> that algorithm.  `actor.state_dict()` / `torch.save(critic.state_dict())`
# on a SUB-module reads torch parameters the engine may be ahead of; a state_dict pre-hook on every sub-module syncs first.
# The hook is a module-level function (picklable by reference) and finds its owner through this process-local weak table, so
# the modules carry no closure; in another process (an unpickled copy) the lookup misses and the hook is a no-op.
_LAZY_OWNERS: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()


def _sync_owner_before_state_dict(module, prefix, keep_vars) -> None:
    owner = _LAZY_OWNERS.get(id(module))
    # (the table is keyed by id(): an entry may outlive its module -- a replaced sub-module whose id a foreign module took over --
    # so the owner is asked whether the module is still one of its own)
    if owner is not None and not owner.__dict__.get("_hip_in_update", False) and any(m is module for m in owner.modules()):
        owner.hip_sync()


class _HipGlue:
    """Mixed in (first base) by every Hip* subclass: the two places where the torch-side state of the
    reference moves underneath an engine that snapshotted it.

    * Learning rates.  `Algorithm._update` steps every scheduler after each update()
      (algorithm_base.py:516-518, 628-629; `LambdaLR` only rewrites `param_groups[i]["lr"]`, optim.py:22-53) and
      examples/mujoco/mujoco_ppo.py:124-131 turns linear decay on by default.  The engines rebuild their
      hyper-parameter struct from `eng.cfg` on every call, so `_hip_refresh_lr()` at the top of each
      `_update_with_batch` re-reads the optimizers' current values.
    * `load_state_dict` on the algorithm replaces parameters / Adam moments / lagged networks / counters: the engine
      is dropped and rebuilt from the loaded torch state on the next update.  A load into a sub-module
      (`algorithm.policy.load_state_dict(best)`) replaces parameters only: it is noticed through the parameters'
      version counters, the engine's Adam moments are first written into torch.optim and survive the rebuild.
    Only writes that bump a tensor's version counter are seen (`load_state_dict`, `param.copy_()`, in-place ops on the
    Parameter); writes through `param.data` (`param.data.copy_()`, optimizer-style `param.data.add_()`, `module.to()`) do
    NOT bump it -- after such an edit call `algorithm.hip_invalidate()`.
    `_HIP_LR`: (engine cfg field, attribute path of the Algorithm.Optimizer wrapper that owns it)."""
    _HIP_LR: tuple = (("lr", "optim"),)
    _hip_dp_on = False            # classes whose constructor takes data_parallel= call _hip_dp_setup

    def _hip_glue_init(self) -> None:
        # Only the algorithm itself carries a hook, and it is a module-level function: sub-modules stay free of
        # closures so that `torch.save(policy)` / `pickle.dumps(algorithm.policy)` (highlevel/persistence.py:106)
        # and `copy.deepcopy(policy)` keep working.  Loads into sub-modules are caught by the version check below.
        self.register_load_state_dict_post_hook(_algorithm_state_loaded)

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        f = cls.__dict__.get("_update_with_batch")
        if f is not None and not getattr(f, "_hip_wrapped", False):
            import functools

            @functools.wraps(f)
            def _update_with_batch(self, *a, **k):
                self.__dict__["_hip_in_update"] = True      # our own write-back must not look like a foreign write
                try:
                    out = f(self, *a, **k)
                finally:
                    self.__dict__["_hip_in_update"] = False
                self._hip_mark()          # the engine has just written the parameters back: that is "our" version
                return out

            _update_with_batch._hip_wrapped = True
            cls._update_with_batch = _update_with_batch

    # The engine snapshots the torch parameters.  `_hip_engine` is a property so that every access first checks
    # whether somebody else has written them since (`algorithm.policy.load_state_dict(best)`,
    # `actor.load_state_dict(...)`, an in-place edit): tensor `_version` counters of all parameters.  If so, the
    # engine-side Adam moments are flushed into torch.optim (they are still valid: only parameters were replaced) and
    # the engine is rebuilt from the torch state on the next use.
    def _hip_current_versions(self) -> tuple:
        # the modules' own `_parameters` dicts, collected once per engine (walking `self.parameters()` costs 0.16 ms on a SAC
        # algorithm and runs on every hook); a Parameter object replaced inside a module is still seen (the dict is live), a
        # sub-module added afterwards is not -- `hip_invalidate()` covers that
        dicts = self.__dict__.get("_hip_pdicts")
        if dicts is None:
            dicts = self.__dict__["_hip_pdicts"] = [m._parameters for m in self.modules() if m._parameters]
        return tuple((id(p), p._version) for d in dicts for p in d.values() if p is not None)

    def _hip_mark(self) -> None:
        if self.__dict__.get("_hip_engine_obj") is not None and not self.__dict__.get("_hip_clean_mark", False):
            self.__dict__["_hip_versions"] = self._hip_current_versions()
        self.__dict__["_hip_clean_mark"] = False

    @property
    def _hip_engine(self):
        eng = self.__dict__.get("_hip_engine_obj")
        if eng is not None and not self.__dict__.get("_hip_in_update", False):
            seen = self.__dict__.get("_hip_versions")
            if seen is not None:
                now = self._hip_current_versions()
                if seen != now:
                    self.__dict__["_hip_versions"] = None          # (the flush reads `_hip_engine` itself: no re-entry)
                    if self.__dict__.get("_hip_stale", False):
                        # lazy write-back: what the engine learnt since the last sync goes to every parameter that was NOT
                        # written by somebody else (theirs wins), then the engine is rebuilt from the torch state
                        old = dict(seen)
                        self.__dict__["_hip_skip_ids"] = {i for i, v in now if old.get(i, v) != v} | (set(dict(now)) - set(old))
                        try:
                            self._hip_write_back()
                        finally:
                            self.__dict__["_hip_skip_ids"] = None
                            self.__dict__["_hip_stale"] = False
                    self._hip_flush()
                    self.__dict__["_hip_engine_obj"] = eng = None
                    self._hip_adam_dirty = False
        return eng

    @_hip_engine.setter
    def _hip_engine(self, value) -> None:
        self.__dict__["_hip_engine_obj"] = value
        self.__dict__["_hip_pdicts"] = self.__dict__["_hip_modules"] = None
        self.__dict__["_hip_versions"] = None if value is None else self._hip_current_versions()

    # -- write-back of what the engine learnt (off-policy subclasses) ---------------------------------------------------
    # "eager": after every update (the torch modules are always current: what round 5 did).  "lazy": when somebody reads
    # them -- `state_dict()` of the algorithm or of the attached policy, pickling, `hip_sync()`, a foreign write, a
    # rebuild.  The collector does not: with `policy_forward="hip"` it acts with the engine's parameters
    # (tianshou_amd/policy.py).  An eager write-back is ~600 small torch ops (2 ms per SAC update against 0.36 ms of GPU work).
    def _hip_put(self, p, t) -> None:
        skip = self.__dict__.get("_hip_skip_ids")
        if skip and id(p) in skip:
            return
        p.copy_(t.reshape(p.shape) if t.shape != p.shape else t)

    def _hip_write_back(self) -> None:
        """Engine parameters / lagged parameters / optimizer state -> the torch modules (subclasses with lazy write-back)."""

    def _hip_after_update(self) -> None:
        if self.__dict__.get("_hip_lazy", False):
            self.__dict__["_hip_stale"] = True
            self.__dict__["_hip_clean_mark"] = True          # nothing was written to the torch parameters: the mark stands
        else:
            self._hip_write_back()

    def hip_sync(self) -> None:
        """Public: make the torch modules and optimizers current (no-op unless updates are pending under write_back="lazy")."""
        if self.__dict__.get("_hip_stale", False) and self.__dict__.get("_hip_engine_obj") is not None:
            self.__dict__["_hip_stale"] = False
            self.__dict__["_hip_in_update"] = True
            try:
                self._hip_write_back()
            finally:
                self.__dict__["_hip_in_update"] = False
            self.__dict__["_hip_versions"] = self._hip_current_versions()

    def _hip_set_write_back(self, write_back: str, attached: bool) -> None:
        if write_back not in ("auto", "lazy", "eager"):
            raise ValueError("write_back must be 'auto', 'lazy' or 'eager'")
        self.__dict__["_hip_lazy"] = write_back == "lazy" or (write_back == "auto" and attached)
        if self.__dict__["_hip_lazy"]:
            for m in self.modules():                      # (the algorithm's own state_dict() / hip_sync() are overridden)
                if m is self:
                    continue
                _LAZY_OWNERS[id(m)] = self                # (latest owner wins: a stale entry under a recycled id is replaced)
                if _sync_owner_before_state_dict not in m._state_dict_pre_hooks.values():
                    m.register_state_dict_pre_hook(_sync_owner_before_state_dict)

    def _hip_offpolicy_update(self, buffer, sample_size, Batch, TrainingStats=None):
        """`OffPolicyAlgorithm.update` -> `Algorithm._update` (algorithm_base.py:586-631, 893-903): the same steps in the same
        order, except that `buffer.sample(sample_size)` -- `sample_indices` + a host fancy-index copy of every key of the batch
        (1.5 ms for 4,096 Humanoid transitions) -- keeps only its first half: the hooks read the rows from the device mirror
        by index.  A prioritized buffer's importance weights are attached as its `__getitem__` does (prio.py:103-106)."""
        import time

        if not self.policy.is_within_training_step:
            raise RuntimeError(
                f"update() was called outside of a training step as signalled by {self.policy.is_within_training_step=} "
                "(see tianshou.utils.torch_utils.policy_within_training_step)")
        if buffer is None:
            return TrainingStats() if TrainingStats is not None else None
        start = time.time()
        indices = buffer.sample_indices(sample_size)
        batch = Batch()
        if hasattr(buffer, "get_weight"):
            w = buffer.get_weight(indices)
            batch.weight = w / np.max(w) if getattr(buffer, "_weight_norm", True) else w
        self.__dict__["_hip_own_sequence"] = True        # (HipSAC: preprocess + update of an n_step = 1 batch become one library call)
        try:
            batch = self._preprocess_batch(batch, buffer, indices)
        finally:
            self.__dict__["_hip_own_sequence"] = False
        # torch_train_mode (torch_utils.py:14-22) is `train(True)` ... `train(was_training)` around the update; the engine reads
        # no module flag, so only the second call is observable: every sub-module ends in the algorithm's mode.  `train()` walks
        # ~60 modules through `Module.__setattr__` (0.24 ms); reading their flags costs 5 us and is almost always enough
        was_training = self.training
        try:
            stat = self._update_with_batch(batch)
        finally:
            mods = self.__dict__.get("_hip_modules")
            if mods is None:
                mods = self.__dict__["_hip_modules"] = list(self.modules())
            if any(m.training != was_training for m in mods):
                self.train(was_training)
        self._postprocess_batch(batch, buffer, indices)
        for lr_scheduler in self.lr_schedulers:
            lr_scheduler.step()
        stat.train_time = time.time() - start
        return stat

    # -- data parallelism: one process per GPU, the replay buffer sharded by sub-buffer (env id), one
    # all-reduce of the flat gradient per minibatch step; replaces the reference's single-process nn.DataParallel
    # (utils/net/common.py:473-515).  `_hip_dp_setup` is called by the constructors that take `data_parallel=`.
    def _hip_dp_setup(self, data_parallel: bool, group, allreduce, shard_buffer: bool) -> None:
        self._hip_dp_on = bool(data_parallel)
        self._hip_group = group
        self._hip_allreduce = allreduce          # None: torch.distributed; "native": NativeAllReduce; or a callable
        self._hip_shard = bool(shard_buffer) and self._hip_dp_on
        self._hip_dp_obj = None

    def _hip_world(self) -> tuple[int, int]:
        import torch.distributed as dist

        if not getattr(self, "_hip_dp_on", False) or not dist.is_initialized():
            return 0, 1
        return dist.get_rank(self._hip_group), dist.get_world_size(self._hip_group)

    def _hip_shard_range(self, buffer):
        """This rank's sub-buffers of `buffer` (None: mirror everything)."""
        rank, world = self._hip_world()
        if not getattr(self, "_hip_shard", False) or world == 1:
            return None
        from .distributed import shard_envs

        n_env = len(buffer.buffers) if hasattr(buffer, "buffers") else 1
        if n_env < world:
            raise ValueError(f"cannot shard {n_env} sub-buffers over {world} ranks")
        return shard_envs(n_env, rank, world)

    def _hip_dp(self, wrapper_cls, eng):
        """The DataParallel* wrapper around the current engine (rebuilt when the engine is)."""
        obj = self._hip_dp_obj
        if obj is None or obj.eng is not eng:
            ar = self._hip_allreduce
            if ar == "native":
                from .collective import NativeAllReduce

                ar = self.__dict__.get("_hip_native_ar")
                if ar is None:
                    ar = self.__dict__["_hip_native_ar"] = NativeAllReduce(self._hip_device, group=self._hip_group)
            obj = self._hip_dp_obj = wrapper_cls(eng, group=self._hip_group, allreduce=ar)
        return obj

    def _hip_invalidate(self) -> None:
        """`Algorithm.load_state_dict` replaced parameters AND optimizer state: drop the engine without flushing."""
        self.__dict__["_hip_engine_obj"] = None
        self.__dict__["_hip_versions"] = None
        self.__dict__["_hip_pdicts"] = self.__dict__["_hip_modules"] = None
        self.__dict__["_hip_stale"] = False
        self._hip_adam_dirty = False

    def hip_invalidate(self, keep_optimizer: bool = True) -> None:
        """Public: the torch parameters were edited in a way the version check cannot see (`param.data` writes, `.to()`):
        drop the engine so that the next update rebuilds it from the torch state.  keep_optimizer=True first writes the
        engine's Adam moments / step counters into torch.optim (they survive the rebuild); False discards them (use after
        replacing the optimizer state yourself)."""
        if keep_optimizer and self.__dict__.get("_hip_engine_obj") is not None:
            self.__dict__["_hip_versions"] = None
            self._hip_flush()
        self._hip_invalidate()          # (pending lazy updates are dropped with the engine: the caller's edit of the torch state wins)

    def _hip_flush(self) -> None:
        """Engine-side optimizer state -> torch.optim state; the default wrappers store it after every update."""

    def state_dict(self, *args, **kwargs):
        self.hip_sync()
        self._hip_flush()
        return super().state_dict(*args, **kwargs)

    def _hip_refresh_lr(self) -> None:
        eng = self._hip_engine
        if eng is None:
            return
        seen: dict[str, float] = {}
        for field, path in self._HIP_LR:
            obj = self
            for part in path.split("."):
                obj = getattr(obj, part)
            opt = getattr(obj, "_optim", obj)
            if not hasattr(opt, "param_groups"):          # FixedAlpha (sac.py:161-172): nothing to learn
                continue
            lr = float(opt.param_groups[0]["lr"])
            if any(float(g["lr"]) != lr for g in opt.param_groups):
                raise NotImplementedError("the HIP engines take one learning rate per optimizer")
            if seen.setdefault(field, lr) != lr:
                raise NotImplementedError(f"the HIP engine has one `{field}`; the optimizers sharing it disagree")
            setattr(eng.cfg, field, lr)
            opt._opt_called = True       # the engine performs this optimizer's step (LRScheduler.step's order check)


def optimizer_fields(opt) -> dict:
    """The PPOConfig fields of a torch optimizer built by tianshou/algorithm/optim.py: AdamOptimizerFactory (:89-110, incl.
    weight_decay) or RMSpropOptimizerFactory (:113-140: alpha, eps, weight_decay, momentum, centered -- the optimizer of
    examples/mujoco/mujoco_a2c.py:117).  Everything else (amsgrad, maximize, other classes) raises."""
    g = opt.param_groups[0]
    name = type(opt).__name__
    if g.get("maximize", False) or g.get("amsgrad", False):
        raise NotImplementedError("the HIP engines do not implement amsgrad / maximize")
    if any(any(gi.get(k) != g.get(k) for k in g if k not in ("params", "lr", "initial_lr")) for gi in opt.param_groups):
        raise NotImplementedError("the HIP engines take one hyper-parameter set per optimizer")
    if name == "Adam":
        return dict(optimizer="adam", lr=g["lr"], betas=tuple(g["betas"]), adam_eps=g["eps"],
                    weight_decay=float(g.get("weight_decay", 0) or 0.0))
    if name == "RMSprop":
        if g.get("centered", False) and g.get("momentum", 0) > 0:
            raise NotImplementedError("RMSprop(centered=True, momentum > 0): the engines keep one auxiliary state vector")
        return dict(optimizer="rmsprop", lr=g["lr"], adam_eps=g["eps"], weight_decay=float(g.get("weight_decay", 0) or 0.0),
                    rms_alpha=float(g["alpha"]), rms_momentum=float(g.get("momentum", 0) or 0.0),
                    rms_centered=bool(g.get("centered", False)))
    raise NotImplementedError(f"the HIP engines implement torch.optim.Adam and torch.optim.RMSprop, not {name}")


def ppo_config_from(algorithm) -> PPOConfig:
    """Reads the reference PPO's hyper-parameters (ppo.py:126-144, a2c.py:95-113, optim.py:89-140) and the actor's bound
    (ContinuousActorProbabilistic.max_action / _unbounded, utils/net/continuous.py:194-231)."""
    opt = algorithm.optim._optim
    of = optimizer_fields(opt)
    is_ppo = hasattr(algorithm, "eps_clip")                  # A2C (a2c.py:187-247) has none of the clipping options
    actor = getattr(algorithm.policy, "actor", None)
    bounded = actor is not None and hasattr(actor, "_unbounded") and not actor._unbounded
    return PPOConfig(
        max_action=float(actor.max_action) if bounded else None, **of,
        algo="ppo" if is_ppo else "a2c",
        gamma=algorithm.gamma, gae_lambda=algorithm.gae_lambda, eps_clip=getattr(algorithm, "eps_clip", 0.2),
        dual_clip=getattr(algorithm, "dual_clip", None), value_clip=getattr(algorithm, "value_clip", False),
        advantage_normalization=getattr(algorithm, "advantage_normalization", False),
        recompute_advantage=getattr(algorithm, "recompute_adv", False), vf_coef=algorithm.vf_coef,
        ent_coef=algorithm.ent_coef, max_grad_norm=algorithm.optim._max_grad_norm,
        return_scaling=algorithm.return_scaling)


def _ref(ref, module: str, name: str):
    """`module.name` of the reference package - or `ref.name` when a namespace is injected: the GPU parity tests drive
    the very same hook bodies through minimal stand-ins of the reference classes (tests/standin.py), because
    /root/reference does not exist on the GPU box."""
    if ref is not None:
        return getattr(ref, name)
    import importlib

    return getattr(importlib.import_module(module), name)


def _on_policy_base(algo: str, ref=None):
    """PPO (ppo.py) or A2C (a2c.py:187-290): the hooks and the statistics class are the same."""
    if algo == "ppo":
        return _ref(ref, "tianshou.algorithm.modelfree.ppo", "PPO")
    if algo == "a2c":
        return _ref(ref, "tianshou.algorithm.modelfree.a2c", "A2C")
    raise ValueError("algo must be 'ppo' or 'a2c'")


def _trunk_spec(net, who: str, norm: bool = False):
    """Net -> MLP -> Sequential(Linear, act, Linear, act, ...) (utils/net/common.py:90-178): -> (linear-layer key stems,
    hidden sizes, activation name).  One activation class for all layers (nn.Tanh / nn.ReLU) or none; other activations are
    outside the engine's envelope.  `norm=True` (the PPO / A2C hooks): MLP(norm_layer=nn.LayerNorm) -- Linear -> LayerNorm ->
    activation in every hidden layer (common.py:25-39) -- is accepted as well, see `_trunk_norm`; elsewhere a norm layer raises."""
    try:
        seq = list(net.preprocess.model.model)
    except AttributeError as e:
        raise NotImplementedError(f"HipPPO: unsupported {who} (no preprocess.model.model Sequential)") from e
    stems, hidden, acts, followed = [], [], set(), []
    for i, m in enumerate(seq):
        if isinstance(m, torch.nn.Linear):
            stems.append(f"preprocess.model.model.{i}")
            hidden.append(int(m.out_features))
            followed.append(False)
        elif isinstance(m, (torch.nn.Tanh, torch.nn.ReLU)):
            acts.add("tanh" if isinstance(m, torch.nn.Tanh) else "relu")
            if not followed or followed[-1]:
                raise NotImplementedError(f"HipPPO: {who} trunk has an activation that does not follow a Linear layer")
            followed[-1] = True
        elif isinstance(m, torch.nn.Identity):
            pass
        elif norm and isinstance(m, torch.nn.LayerNorm):
            pass                                              # (position, shape and affine-ness are checked by _trunk_norm)
        else:
            raise NotImplementedError(f"HipPPO: {who} trunk contains {type(m).__name__}; Linear layers with nn.Tanh, nn.ReLU "
                                      "or no activation are supported (nn.LayerNorm under PPO / A2C only, no other norm layers)")
    if not stems or len(acts) > 1:
        raise NotImplementedError(f"HipPPO: {who} trunk needs at least one Linear layer and a single activation class")
    # The engines apply the activation after EVERY trunk layer.  A Net built with action_shape > 0 (MLP output_dim > 0,
    # utils/net/common.py:169-170) ends in a bare Linear layer and Net(softmax=True) appends a softmax (common.py:366-367):
    # neither is a (Linear, activation) pair, so they are outside the envelope rather than silently a different network.
    if acts and not all(followed):
        raise NotImplementedError(f"HipPPO: every Linear layer of the {who} trunk must be followed by its activation "
                                  "(a Net with action_shape / MLP output_dim > 0 ends in a bare Linear layer)")
    if getattr(net.preprocess, "softmax", False):
        raise NotImplementedError(f"HipPPO: the {who} trunk applies a softmax (Net(softmax=True)); not supported")
    return stems, hidden, (acts.pop() if acts else "none")


def _trunk_norm(net, who: str):
    """-> None for a trunk without norm layers, else (key stems of the LayerNorm modules, one per Linear layer, eps): every
    Linear layer is DIRECTLY followed by an nn.LayerNorm over its own width with an elementwise affine map (weight and bias) and
    one eps for all of them -- what MLP(norm_layer=nn.LayerNorm[, norm_args]) builds (utils/net/common.py:25-39, 123-137)."""
    seq = list(net.preprocess.model.model)
    norms = [(i, m) for i, m in enumerate(seq) if isinstance(m, torch.nn.LayerNorm)]
    if not norms:
        return None
    lin = [i for i, m in enumerate(seq) if isinstance(m, torch.nn.Linear)]
    if [i for i, _ in norms] != [i + 1 for i in lin]:
        raise NotImplementedError(f"HipPPO: {who} trunk: a LayerNorm must follow every Linear layer directly (Linear -> LayerNorm "
                                  "-> activation), or none")
    eps = {float(m.eps) for _, m in norms}
    for (i, m), j in zip(norms, lin):
        if tuple(m.normalized_shape) != (seq[j].out_features,) or m.weight is None or m.bias is None:
            raise NotImplementedError(f"HipPPO: {who} trunk: LayerNorm must normalise the layer's width with weight and bias")
    if len(eps) != 1:
        raise NotImplementedError(f"HipPPO: {who} trunk: one eps for all LayerNorm modules")
    return [f"preprocess.model.model.{i}" for i, _ in norms], eps.pop()


def _ln_eps_of(hidden):
    """The ("layer_norm", eps) tag of `_check_supported`'s "net" description -> eps, or None."""
    for tag in hidden[3:] if isinstance(hidden, tuple) else ():
        if isinstance(tag, tuple) and tag[0] == "layer_norm":
            return float(tag[1])
    return None


def _net_keys(actor, critic):
    """state_dict keys of an actor / critic over Net trunks of any depth, in the engine's flat order:
    actor trunk (w, b[, gamma, beta])*, mu (w, b), sigma_param | critic trunk (w, b[, gamma, beta])*, last (w, b)."""
    def trunk(net, who):
        st, _, _ = _trunk_spec(net, who, norm=True)
        nm = _trunk_norm(net, who)
        if nm is None:
            return [f"{x}.{y}" for x in st for y in ("weight", "bias")]
        return [f"{x}.{y}" for a, b in zip(st, nm[0]) for x in (a, b) for y in ("weight", "bias")]
    ka = trunk(actor, "actor") + ["mu.model.0.weight", "mu.model.0.bias"]
    ka += ["sigma.model.0.weight", "sigma.model.0.bias"] if getattr(actor, "_c_sigma", False) else ["sigma_param"]
    kc = trunk(critic, "critic") + ["last.model.0.weight", "last.model.0.bias"]
    return ka, kc


def _check_supported(actor, critic):
    """-> (obs_dim, act_dim, hidden, kind).  kind "fused": the MuJoCo example nets (hidden 64 x 64 tanh, obs <= 31, act <= 8) on
    the fused MFMA kernels of ts_ppo.hip; "wide": any other Net[h, h] tanh (h a multiple of 32 up to 1024, act <= 32, e.g.
    Humanoid's 376 / 17 / 256 x 256) on the implicit-GEMM layer kernels (tianshou_amd/ppo_wide.py); "net": every other trunk
    the reference's Net builds from `hidden_sizes` and one activation (1 .. 7 hidden layers of any widths up to 1024, nn.Tanh /
    nn.ReLU / none, actor and critic trunks may differ) on the same layer kernels (ppo_wide.NetPPOEngine) -- `hidden` is then
    (actor hidden sizes, critic hidden sizes, activation[, "conditioned_sigma"][, ("layer_norm", eps)]): the actor's sigma is
    a second linear head (continuous.py:212-234); every hidden layer is Linear -> LayerNorm -> activation (common.py:25-39)."""
    ka, kc = _net_keys(actor, critic)
    sa, sc = actor.state_dict(), critic.state_dict()
    if set(sa.keys()) != set(ka):
        raise NotImplementedError(f"HipPPO: unsupported actor (keys {sorted(set(sa) ^ set(ka))} differ from a Net trunk + linear mu "
                                  "head + sigma_param); see tianshou_amd/integration.py")
    if set(sc.keys()) != set(kc):
        raise NotImplementedError(f"HipPPO: unsupported critic (keys {sorted(set(sc) ^ set(kc))} differ from a Net trunk + linear head)")
    # ContinuousActorProbabilistic(unbounded=False) is the constructor default (continuous.py:194): mu = max_action * tanh(.)
    # -- built into the fused step / inference kernels (ts_ppo_hparams.max_action) and into the per-layer engine
    # (ts_net_desc.max_action); the Net[h, h] GEMM engine ("wide") stays unbounded, such actors take the per-layer engine
    bounded = not getattr(actor, "_unbounded", False)
    if bounded and not float(getattr(actor, "max_action", 1.0)) > 0.0:
        raise NotImplementedError("HipPPO: a bounded actor needs max_action > 0")
    c_sigma = bool(getattr(actor, "_c_sigma", False))
    _, ha, act_a = _trunk_spec(actor, "actor", norm=True)
    _, hc, act_c = _trunk_spec(critic, "critic", norm=True)
    nm_a, nm_c = _trunk_norm(actor, "actor"), _trunk_norm(critic, "critic")
    obs_dim, act_dim = int(sa[ka[0]].shape[1]), int(sa["mu.model.0.weight"].shape[0])
    if int(sc[kc[0]].shape[1]) != obs_dim or act_a != act_c:
        raise NotImplementedError("HipPPO: actor and critic must read the same observation and use the same activation")
    if (nm_a is None) != (nm_c is None) or (nm_a is not None and nm_a[1] != nm_c[1]):
        raise NotImplementedError("HipPPO: actor and critic trunks must both have LayerNorm (one eps) or neither")
    if sa["mu.model.0.weight"].shape[1] != ha[-1] or tuple(sc["last.model.0.weight"].shape) != (1, hc[-1]):
        raise NotImplementedError("HipPPO: the heads must be single Linear layers on the trunks' outputs")
    if act_dim > (16 if c_sigma else 32):
        raise NotImplementedError("HipPPO: at most 32 actions (16 with conditioned_sigma)")
    if c_sigma and tuple(sa["sigma.model.0.weight"].shape) != (act_dim, ha[-1]):
        raise NotImplementedError("HipPPO: the sigma head must be a single Linear layer on the trunk's output")
    if nm_a is None and not c_sigma and act_a == "tanh" and ha == hc and len(ha) == 2 and ha[0] == ha[1]:
        hidden = ha[0]
        if hidden == 64 and obs_dim <= 31 and act_dim <= 8:
            return obs_dim, act_dim, hidden, "fused"
        if hidden % 32 == 0 and 32 <= hidden <= 1024 and not bounded:
            return obs_dim, act_dim, hidden, "wide"
    if max(len(ha), len(hc)) > 7 or max(ha + hc) > 1024:
        raise NotImplementedError("HipPPO: trunks of up to 7 hidden layers of at most 1024 units")
    tags = (("conditioned_sigma",) if c_sigma else ()) + ((("layer_norm", nm_a[1]),) if nm_a is not None else ())
    return obs_dim, act_dim, (tuple(ha), tuple(hc), act_a) + tags, "net"


def _attach_gauss_policy(algorithm, policy_forward: str, sampling: str, noise_seed) -> None:
    """the collector's `policy(batch)` / `policy.map_action(act)` (data/collector.py:735-744) on the engine's
    inference kernels -- `tianshou_amd.policy.attach` gives `algorithm.policy` a subclass of its own class whose forward reads
    the engine's device-resident parameters.  policy_forward="torch" keeps the reference's torch forward."""
    if policy_forward not in ("hip", "torch"):
        raise ValueError("policy_forward must be 'hip' or 'torch'")
    if policy_forward == "torch":
        return
    from . import policy as HP

    obs_dim, act_dim, hidden, kind = algorithm._hip_dims
    actor = algorithm.policy.actor
    ka, _ = _net_keys(actor, algorithm.critic)
    max_action = None if getattr(actor, "_unbounded", False) else float(actor.max_action)
    kw = dict(device=str(algorithm._hip_device), sampling=sampling, noise_seed=noise_seed, obs_dim=obs_dim, act_dim=act_dim,
              max_action=max_action, actor_keys=tuple(ka))
    if kind == "fused":
        HP.attach(algorithm.policy, "gauss", algorithm, **kw)
    elif kind == "wide":
        from . import npg as NG

        HP.attach(algorithm.policy, "gauss_wide", algorithm, hidden=int(hidden), n_actor=int(NG.layout(obs_dim, hidden, act_dim)["actor_count"]), **kw)
    elif "conditioned_sigma" not in hidden[3:]:              # "net" without conditioned sigma (that head keeps the torch forward)
        import ctypes as C

        from . import _lib

        ln_eps = _ln_eps_of(hidden)
        out = (C.c_int64 * 3)()
        desc = _lib.NetDesc.make(obs_dim, list(hidden[0]), hidden[2], _lib.NetDesc.LAYERNORM if ln_eps is not None else 0, ln_eps=ln_eps or 0.0)
        _lib.check(_lib.load().ts_net_layout(C.byref(desc), _lib.i64(act_dim), out))
        HP.attach(algorithm.policy, "gauss_net", algorithm, hidden=tuple(hidden[0]), activation=hidden[2], n_actor=int(out[1]),
                  ln_eps=ln_eps, **kw)


def make_hip_ppo(algo: str = "ppo", ref=None):
    """Returns the HipPPO class (imports tianshou lazily); algo="a2c": HipA2C(A2C), same networks.
    `ref`: optional namespace replacing the tianshou imports (see `_ref`)."""
    A2CTrainingStats = _ref(ref, "tianshou.algorithm.modelfree.a2c", "A2CTrainingStats")
    SequenceSummaryStats = _ref(ref, "tianshou.data", "SequenceSummaryStats")
    Batch = _ref(ref, "tianshou.data", "Batch")
    PPO = _on_policy_base(algo, ref)

    class HipPPO(_HipGlue, PPO):
        def __init__(self, *args, device="cuda", permutations="device", perm_seed=None, data_parallel=False, group=None,
                     allreduce=None, shard_buffer=True, policy_forward="hip", sampling="device", noise_seed=None, **kwargs):
            """`data_parallel=True` (one process per GPU, torch.distributed initialised): `update()` mirrors only this
            rank's sub-buffers of `buffer` (`shard_buffer`, by env id; pass False when every rank collects into its own
            buffer), computes values / GAE / log pi_old shard-locally with GLOBAL return statistics, and every minibatch
            step all-reduces the flat gradient (`tianshou_amd.distributed.DataParallelPPO`); `allreduce`: None =
            torch.distributed (RCCL), "native" = the C-ABI exchange incl. the one-shot path for the 44 KB payload, or any
            in-place sum callable.  Replicas stay bit-identical.  `group`: the process group of the replicas.

            `permutations`: "device" (default) expands a private key (`perm_seed`, the update counter, the repeat and the
            rank) with ts_random_permutation on the GPU: a keyed bijection per repeat, statistically equivalent to
            Batch.split's shuffles, and NumPy's global generator -- which the collector shares -- is left untouched.
            `perm_seed=None` takes ONE draw from NumPy's global generator at construction, so that `np.random.seed` /
            `seed_everything` select the shuffle sequence as they do in the reference (pass an int to fix it); the seed and
            the update counter are saved with `hip_extra_state()` / restored with `load_hip_extra_state()` (a resumed run
            continues the sequence); `state_dict()` itself keeps the reference's format.
            "host" draws np.random.permutation(N) per repeat exactly like Batch.split (batch.py:1209): the reference's
            sequence for a given seed (the mode the parity tests use), at ~10 ms of host time per 2^20 entries -- 100 ms
            of a 12 ms update(), i.e. ~1.4 k instead of ~14 k update-steps/s at the C2 size.

            `policy_forward="hip"` (default): `self.policy` -- what the Collector calls once per vector step
            (collector.py:735-744) -- gets a subclass of its own class whose `forward` and `map_action` run on the engine's
            inference kernels and read its device-resident parameters (`tianshou_amd.policy`); "torch" keeps the reference's
            forward.  `sampling`: "device" draws dist.sample()'s noise with the engine's counter-based generator
            (`noise_seed`; torch's generator untouched), "torch" from torch's CPU generator in the reference's order."""
            super().__init__(*args, **kwargs)
            if permutations not in ("host", "device"):
                raise ValueError("permutations must be 'host' or 'device'")
            self._hip_perms = permutations
            if perm_seed is None:
                perm_seed = int(np.random.randint(0, 2**31 - 1)) if permutations == "device" else 0
            self._hip_perm_seed, self._hip_updates = int(perm_seed), 0
            self._hip_device = torch.device(device)
            self._hip_dims = _check_supported(self.policy.actor, self.critic)
            self._hip_dp_setup(data_parallel, group, allreduce, shard_buffer)
            self._hip_engine = None
            self._hip_glue_init()
            self._hip_batch = None
            self._hip_synced = False
            _attach_gauss_policy(self, policy_forward, sampling, noise_seed)

        # -- the shuffle key travels BESIDE the checkpoint ---------------------------------------------
        # state_dict() stays in the reference's format (a HipPPO checkpoint loads into the reference PPO / A2C class with
        # strict=True, nested in a parent module or not); the (seed, update counter) pair of the device permutations is
        # saved / restored explicitly, e.g. torch.save({"model": algo.state_dict(), "hip": algo.hip_extra_state()}, f).
        def hip_extra_state(self) -> dict:
            return {"perm_seed": int(self._hip_perm_seed), "updates": int(self._hip_updates)}

        def load_hip_extra_state(self, state: dict) -> None:
            self._hip_perm_seed, self._hip_updates = int(state["perm_seed"]), int(state["updates"])
            self._hip_key_loaded = True

        def load_state_dict(self, state_dict, *args, **kwargs):
            # checkpoints of round-4 builds carried the pair inline -- under "_hip_perm_state", or "<prefix>_hip_perm_state"
            # when the algorithm was saved as a sub-module of a parent: taken out so that strict loading sees the reference's keys
            legacy = [k for k in state_dict if k == "_hip_perm_state" or k.endswith("._hip_perm_state")]
            if legacy:
                state_dict = dict(state_dict)
                for k in legacy:
                    st = state_dict.pop(k)
                    if k == "_hip_perm_state":
                        self._hip_perm_seed, self._hip_updates = int(st[0]), int(st[1])
            out = super().load_state_dict(state_dict, *args, **kwargs)
            self._hip_key_loaded = "_hip_perm_state" in legacy        # (checked at the next update, see _update_with_batch)
            return out

        def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
            # the same legacy key when THIS module is loaded as part of a parent's state_dict (nn.Module.load_state_dict recurses
            # through _load_from_state_dict, not through load_state_dict)
            state_dict.pop(prefix + "_hip_perm_state", None)
            return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

        # -- engine life cycle ------------------------------------------------------------------
        def _hip_flat(self, tensors) -> torch.Tensor:
            """13 tensors in `_hip_params()` order (parameters or Adam moments) -> the engine's flat vector."""
            obs_dim, act_dim, hidden, kind = self._hip_dims
            if kind == "fused":
                return torch.cat([t.detach().reshape(-1).float() for t in tensors]).to(self._hip_device).contiguous()
            if kind == "net":
                from .ppo_wide import net_flat_from_tensors

                cs, ln = "conditioned_sigma" in hidden[3:], _ln_eps_of(hidden) is not None
                na = (4 if ln else 2) * len(hidden[0]) + 2 + (2 if cs else 1)
                return torch.cat([net_flat_from_tensors(list(tensors[:na]), obs_dim, list(hidden[0]), act_dim, self._hip_device,
                                                        conditioned_sigma=cs, layer_norm=ln),
                                  net_flat_from_tensors(list(tensors[na:]), obs_dim, list(hidden[1]), None, self._hip_device,
                                                        layer_norm=ln)]).contiguous()
            from .ppo_wide import flat_from_tensors

            return flat_from_tensors(list(tensors[:7]), list(tensors[7:]), obs_dim, hidden, act_dim, self._hip_device)

        def _hip_unflat(self, flat: torch.Tensor) -> list[torch.Tensor]:
            """The engine's flat vector -> 13 tensors in `_hip_params()` order (shapes as flattened / nn.Linear layout)."""
            obs_dim, act_dim, hidden, kind = self._hip_dims
            if kind == "fused":
                return list(torch.split(flat, [p.numel() for p in self._hip_params()]))
            if kind == "net":
                a, c = self._engine().flat_to_tensors(flat)
                return a + c
            from .ppo_wide import flat_to_tensors

            a, c = flat_to_tensors(flat, obs_dim, hidden, act_dim)
            return a + c

        def _engine(self):
            if self._hip_engine is None:
                obs_dim, act_dim, hidden, kind = self._hip_dims
                flat = self._hip_flat([p.detach() for p in self._hip_params()])
                if kind == "fused":
                    eng = PPOEngine(obs_dim, act_dim, flat, ppo_config_from(self))
                elif kind == "net":
                    from .ppo_wide import NetPPOEngine

                    ln_eps = _ln_eps_of(hidden)
                    eng = NetPPOEngine(obs_dim, act_dim, hidden[0], hidden[1], hidden[2], flat, ppo_config_from(self),
                                       conditioned_sigma="conditioned_sigma" in hidden[3:], layer_norm=ln_eps is not None,
                                       ln_eps=ln_eps or 1e-5)
                else:
                    from .ppo_wide import WidePPOEngine

                    eng = WidePPOEngine(obs_dim, act_dim, hidden, flat, ppo_config_from(self))
                self._hip_engine = eng
                eng.ret_rms = [float(self.ret_rms.mean), float(self.ret_rms.var), float(self.ret_rms.count)]
                # resume: Adam moments / step of a loaded checkpoint (algorithm_base.py:523-543)
                ms, vs, step = adam_state(self.optim._optim, self._hip_params())
                eng.adam_m, eng.adam_v, eng.adam_step = self._hip_flat(ms), self._hip_flat(vs), step
            return self._hip_engine

        def _hip_params(self):
            ka, kc = _net_keys(self.policy.actor, self.critic)      # == TIANSHOU_ACTOR_KEYS / _CRITIC_KEYS for Net[h, h]
            return params_by_keys(self.policy.actor, ka) + params_by_keys(self.critic, kc)

        def _hip_runner(self, eng):
            if not self._hip_dp_on:
                return eng
            from .distributed import DataParallelPPO, DataParallelWidePPO

            return self._hip_dp(DataParallelPPO if self._hip_dims[3] == "fused" else DataParallelWidePPO, eng)

        def _sync_back(self) -> None:
            """After every update(): engine parameters -> nn.Parameters (the collector acts with the torch modules;
            device-to-device when they live on the GPU, no host synchronisation) and the three ret_rms scalars.
            The Adam moments are only needed by `state_dict()` (algorithm_base.py:523-543) and move there
            (`_hip_flush`)."""
            eng = self._hip_engine
            with torch.no_grad():
                for p, t in zip(self._hip_params(), self._hip_unflat(eng.params)):
                    p.copy_(t.reshape(p.shape).to(p.device))
            self.ret_rms.mean, self.ret_rms.var, self.ret_rms.count = eng.ret_rms
            self._hip_adam_dirty = True

        def _hip_flush(self) -> None:
            eng = self._hip_engine
            if eng is None or not getattr(self, "_hip_adam_dirty", False):
                return
            store_adam_state(self.optim._optim, self._hip_params(), self._hip_unflat(eng.adam_m), self._hip_unflat(eng.adam_v),
                             eng.adam_step)
            self._hip_adam_dirty = False

        # -- Algorithm.update ---------------------------------------------------------------------------
        def update(self, buffer, batch_size, repeat):
            """`OnPolicyAlgorithm.update` -> `Algorithm._update` (algorithm_base.py:586-631, 854-865), same steps in the
            same order, with `buffer.sample(0)` - a host fancy-index copy of every key, 1.4 s at 2^20 transitions -
            replaced by the device mirror of the buffer: only the slots written since the previous update cross PCIe,
            `sample_indices(0)`, the gathers and the unfinished-slot cuts run as kernels."""
            _require_gpu(self._hip_device, type(self).__name__)
            if not self.policy.is_within_training_step:
                raise RuntimeError(
                    f"update() was called outside of a training step as signalled by {self.policy.is_within_training_step=} "
                    "(see tianshou.utils.torch_utils.policy_within_training_step)")
            if buffer is None:
                return super().update(buffer, batch_size, repeat)
            import time

            start = time.time()
            m = _mirror(self, buffer, self._hip_device)
            self._hip_synced = True
            indices = m.sample_indices(0)
            batch = self._preprocess_batch(Batch(), buffer, indices)
            was_training = self.training                          # torch_train_mode (torch_utils.py:14-22)
            try:
                self.train(True)
                stat = self._update_with_batch(batch, batch_size, repeat)
            finally:
                self.train(was_training)
            if hasattr(buffer, "update_weight"):
                self._postprocess_batch(batch, buffer, m.to_global(indices).cpu().numpy())
            for lr_scheduler in self.lr_schedulers:
                lr_scheduler.step()
            stat.train_time = time.time() - start
            return stat

        # -- hooks ------------------------------------------------------------------------------------
        def _preprocess_batch(self, batch, buffer, indices):
            """a2c.py:239-247 / ppo.py:146-162.  Works on the device mirror at `indices` (the host copies in `batch`,
            when the caller is the reference's own `Algorithm._update`, are not read)."""
            from .returns import cut_positions

            _require_gpu(self._hip_device, type(self).__name__)
            eng = self._engine()
            if self._hip_synced:
                m, self._hip_synced = self._hip_mirror, False
            else:
                m = _mirror(self, buffer, self._hip_device)
            idx = indices if isinstance(indices, torch.Tensor) else \
                torch.as_tensor(np.asarray(indices, np.int64), device=self._hip_device)
            whole = idx.numel() == m.maxsize and m.indices_are_identity()     # sample(0) of full, unwrapped sub-buffers
            take = (lambda x: x) if whole else (lambda x: m.gather_tensor(x, idx))   # noqa: E731
            obs_next = take(m.obs_next) if m.obs_next is not None else m.gather_tensor(m.obs, m.next(idx))
            cut, d_n = cut_positions(m, idx)                                   # algorithm_base.py:715
            runner = self._hip_runner(eng)                                     # the engine, or its data-parallel wrapper
            b = runner.preprocess(take(m.obs), obs_next, take(m.act), take(m.rew), take(m.terminated), take(m.truncated),
                                  cut, d_n)
            self._hip_batch = b
            batch.v_s, batch.returns, batch.adv = b["v_s"], b["returns"], b["adv"]
            batch.act = b["act"]
            if algo == "ppo":                                                  # A2C has none (a2c.py:239-247)
                batch.logp_old = b["logp_old"]
            return batch

        def _update_with_batch(self, batch, batch_size, repeat):
            self._hip_refresh_lr()
            eng = self._engine()
            n = int(self._hip_batch["obs"].shape[0])
            if self._hip_perms == "host":
                perms = [np.random.permutation(n) for _ in range(repeat)]    # Batch.split, batch.py:1209
            else:
                from .buffer import random_permutation

                if self.__dict__.get("_hip_key_loaded") is False and eng.adam_step > 0:
                    # a trained checkpoint came in through load_state_dict alone: the optimizer continues, the shuffle key does not
                    import warnings

                    warnings.warn("HipPPO: resuming from a checkpoint without its device-permutation key: the shuffle sequence "
                                  "restarts from this object's own (perm_seed, 0).  Save `hip_extra_state()` beside `state_dict()` "
                                  "and restore it with `load_hip_extra_state()` to continue the sequence.", stacklevel=2)
                self.__dict__["_hip_key_loaded"] = None
                self._hip_updates += 1
                rank = self._hip_world()[0]
                key = ((self._hip_perm_seed * 0x9E3779B97F4A7C15) ^ (self._hip_updates << 20) ^ (rank << 52)) & (2**64 - 1)
                perms = [random_permutation(n, key + r, self._hip_device) for r in range(repeat)]
            losses, steps = self._hip_runner(eng).update(self._hip_batch, batch_size, repeat, perms)
            arr = losses.cpu().numpy().astype(np.float64)              # one D2H per update()
            eng.check()                                                # surfaces a stuck GAE hand-off (never observed)
            self._sync_back()
            return A2CTrainingStats(
                loss=SequenceSummaryStats.from_sequence(arr[:, 0]),
                actor_loss=SequenceSummaryStats.from_sequence(arr[:, 1]),
                vf_loss=SequenceSummaryStats.from_sequence(arr[:, 2]),
                ent_loss=SequenceSummaryStats.from_sequence(arr[:, 3]),
                gradient_steps=steps,
            )

    if algo == "a2c":
        HipPPO.__name__ = HipPPO.__qualname__ = "HipA2C"
    return HipPPO


# ---------------------------------------------------------------------------------------------------
# NPG (npg.py) / TRPO (trpo.py) on the MuJoCo actor-critic
# ---------------------------------------------------------------------------------------------------
def _make_hip_natural(algo: str, ref=None):
    SequenceSummaryStats = _ref(ref, "tianshou.data", "SequenceSummaryStats")

    from . import npg as NG

    if algo == "npg":
        Base = _ref(ref, "tianshou.algorithm.modelfree.npg", "NPG")
        Stats = _ref(ref, "tianshou.algorithm.modelfree.npg", "NPGTrainingStats")
    else:
        Base = _ref(ref, "tianshou.algorithm.modelfree.trpo", "TRPO")
        Stats = _ref(ref, "tianshou.algorithm.modelfree.trpo", "TRPOTrainingStats")
    who = "HipNPG" if algo == "npg" else "HipTRPO"

    class HipNatural(_HipGlue, Base):
        def __init__(self, *args, device="cuda", **kwargs):
            """Net[h1, h2] tanh actor and critic (the nets of examples/mujoco/mujoco_npg.py, any two widths): the fused / GEMM
            passes of NPGEngine.  Every other `Net(hidden_sizes=[...], activation=Tanh | ReLU | None)` trunk -- other depths,
            ReLU, actor and critic trunks that differ -- (round 6): NetNPGEngine, layer by layer on the GEMM kernels."""
            super().__init__(*args, **kwargs)
            self._hip_device = torch.device(device)
            actor, critic = self.policy.actor, self.critic
            sa, sc = actor.state_dict(), critic.state_dict()
            if getattr(actor, "_c_sigma", True) or not getattr(actor, "_unbounded", False):
                raise NotImplementedError(f"{who}: actor must be unbounded with a state-independent sigma_param")
            _, ha, act_a = _trunk_spec(actor, "actor")               # (raises for anything but Linear + Tanh / ReLU / nothing)
            _, hc, act_c = _trunk_spec(critic, "critic")
            ka, kc = _net_keys(actor, critic)                        # == TIANSHOU_ACTOR_KEYS / _CRITIC_KEYS for two hidden layers
            if set(sa.keys()) != set(ka) or list(sc.keys()) != list(kc) or act_a != act_c:
                raise NotImplementedError(f"{who}: actor and critic must be Net trunks with one activation + a linear mu / value head "
                                          "(examples/mujoco/mujoco_npg.py)")
            self._hip_akeys, self._hip_ckeys = ka, kc
            self._hip_two = len(ha) == 2 and len(hc) == 2 and act_a == "tanh"
            if not 1 <= int(sa["mu.model.0.weight"].shape[0]) <= 32:
                raise NotImplementedError(f"{who}: at most 32 actions")
            if self._hip_two:
                from . import widths as WD

                try:        # any two hidden widths per network (round 6): embedded by zero padding (tianshou_amd.widths)
                    la, lc = [sa[k] for k in ka[:6]], [sc[k] for k in kc]
                    self._hip_sizes = {"actor": WD.two_layer_widths(la), "critic": WD.two_layer_widths(lc)}
                    self._hip_hidden = WD.common_hidden(la, lc)
                except NotImplementedError as e:
                    raise NotImplementedError(f"{who}: hidden layers of widths up to 1024 per network ({e})") from None
            else:
                if max(ha + hc) > 1024 or not 1 <= len(ha) <= 7 or not 1 <= len(hc) <= 7:
                    raise NotImplementedError(f"{who}: 1 .. 7 hidden layers of widths up to 1024 per network")
                self._hip_trunks = (list(ha), list(hc), act_a)
            _adam_of(self.optim)
            self._hip_engine = None
            self._hip_glue_init()

        def _engine(self):
            if self._hip_engine is None:
                sa, sc = self.policy.actor.state_dict(), self.critic.state_dict()
                ka, kc = self._hip_akeys, self._hip_ckeys
                obs_dim, act_dim = sa[ka[0]].shape[1], sa["mu.model.0.weight"].shape[0]
                opt, g = _adam_of(self.optim)
                cfg = NG.NPGConfig(algo=algo, gamma=self.gamma, gae_lambda=self.gae_lambda,
                                   optim_critic_iters=self.optim_critic_iters,
                                   trust_region_size=float(getattr(self, "trust_region_size", 0.5)),
                                   advantage_normalization=self.advantage_normalization, return_scaling=self.return_scaling,
                                   damping=float(self._damping), max_kl=float(getattr(self, "max_kl", 0.01)),
                                   backtrack_coeff=float(getattr(self, "backtrack_coeff", 0.8)),
                                   max_backtracks=int(getattr(self, "max_backtracks", 10)), lr=g["lr"], betas=tuple(g["betas"]),
                                   adam_eps=g["eps"], max_grad_norm=self.optim._max_grad_norm)
                dev = self._hip_device
                ms, vs, step = adam_state(opt, params_by_keys(self.critic, kc))     # resume
                if self._hip_two:
                    hidden = self._hip_hidden
                    eng = self._hip_engine = NG.NPGEngine(
                        obs_dim, act_dim, hidden, NG.actor_flat_from_torch([sa[k] for k in ka], obs_dim, hidden, act_dim, dev),
                        NG.critic_flat_from_torch([sc[k] for k in kc], obs_dim, hidden, dev), cfg)
                    eng.critic_m = NG.critic_flat_from_torch(ms, obs_dim, hidden, dev)
                    eng.critic_v = NG.critic_flat_from_torch(vs, obs_dim, hidden, dev)
                else:
                    from .ppo_wide import net_flat_from_tensors as nf

                    ha, hc, act_name = self._hip_trunks
                    eng = self._hip_engine = NG.NetNPGEngine(
                        obs_dim, act_dim, ha, hc, act_name, nf([sa[k] for k in ka], obs_dim, ha, act_dim, dev),
                        nf([sc[k] for k in kc], obs_dim, hc, None, dev), cfg)
                    eng.critic_m, eng.critic_v = eng.critic_from_tensors(ms), eng.critic_from_tensors(vs)
                eng.ret_rms = [float(self.ret_rms.mean), float(self.ret_rms.var), float(self.ret_rms.count)]
                eng.adam_step = step
            return self._hip_engine

        def _preprocess_batch(self, batch, buffer, indices):
            _require_gpu(self._hip_device, who)
            eng = self._engine()
            dev = self._hip_device
            t = lambda x, dt=None: torch.as_tensor(np.ascontiguousarray(x), device=dev) if dt is None \
                else torch.as_tensor(np.ascontiguousarray(x), device=dev).to(dt)  # noqa: E731
            cut = np.nonzero(np.isin(indices, buffer.unfinished_index()))[0]      # algorithm_base.py:715
            b = self._hip_pre = eng.preprocess(t(batch.obs, torch.float32), t(batch.obs_next, torch.float32),
                                               t(batch.act, torch.float32), t(batch.rew, torch.float64), t(batch.terminated),
                                               t(batch.truncated), t(cut))
            batch.v_s, batch.returns, batch.adv, batch.logp_old, batch.act = b["v_s"], b["returns"], b["adv"], b["logp_old"], b["act"]
            return batch

        def _update_with_batch(self, batch, batch_size, repeat):
            self._hip_refresh_lr()
            eng = self._hip_engine
            perms = [np.random.permutation(len(batch)) for _ in range(repeat)]     # Batch.split, batch.py:1209
            stats, _ = eng.update(self._hip_pre, batch_size, repeat, perms)
            arr = stats.cpu().numpy().astype(np.float64)                          # one D2H per update()
            if self._hip_two:
                dims, szc = (eng.obs_dim, eng.hidden), self._hip_sizes["critic"]
                actor_t = NG.actor_flat_to_torch(eng.actor, eng.obs_dim, eng.hidden, eng.act_dim, sizes=self._hip_sizes["actor"])
                critic_t, m_t, v_t = (NG.critic_flat_to_torch(x, *dims, sizes=szc) for x in (eng.critic, eng.critic_m, eng.critic_v))
            else:
                actor_t = eng.actor_to_tensors(eng.actor)
                critic_t, m_t, v_t = (eng.critic_to_tensors(x) for x in (eng.critic, eng.critic_m, eng.critic_v))
            cparams = params_by_keys(self.critic, self._hip_ckeys)
            with torch.no_grad():
                for p, t in zip(params_by_keys(self.policy.actor, self._hip_akeys), actor_t):
                    p.copy_(t.reshape(p.shape))
                for p, t in zip(cparams, critic_t):
                    p.copy_(t.reshape(p.shape))
            store_adam_state(self.optim._optim, cparams, [t.reshape(p.shape) for p, t in zip(cparams, m_t)],
                             [t.reshape(p.shape) for p, t in zip(cparams, v_t)], eng.adam_step)
            self.ret_rms.mean, self.ret_rms.var, self.ret_rms.count = eng.ret_rms
            seq = SequenceSummaryStats.from_sequence
            kw = dict(actor_loss=seq(arr[:, 0]), vf_loss=seq(arr[:, 1]), kl=seq(arr[:, 2]))
            if algo == "trpo":
                kw["step_size"] = seq(arr[:, 3])
            return Stats(**kw)

    HipNatural.__name__ = HipNatural.__qualname__ = who
    return HipNatural


def make_hip_npg(ref=None):
    """Returns HipNPG(NPG): `_preprocess_batch` / `_update_with_batch` (npg.py:123-193) on the engine.  Supported nets:
    examples/mujoco/mujoco_npg.py:103-128 (Net[h, h] tanh actor and critic, unbounded Gaussian actor with sigma_param).
    `ref`: optional namespace replacing the tianshou imports (see `_ref`)."""
    return _make_hip_natural("npg", ref)


def make_hip_trpo(ref=None):
    """Returns HipTRPO(TRPO): the same hooks with TRPO's step size and line search (trpo.py:123-214)."""
    return _make_hip_natural("trpo", ref)


def make_hip_reinforce(ref=None):
    """Returns HipReinforce(Reinforce): `_preprocess_batch` / `_update_with_batch` (reinforce.py:346-382) on the engine.
    Supported net: the actor of examples/mujoco/mujoco_reinforce.py:84-103 (Net[h, h] tanh, unbounded Gaussian, sigma_param).
    `ref`: optional namespace replacing the tianshou imports (see `_ref`)."""
    LossSequenceTrainingStats = _ref(ref, "tianshou.algorithm.modelfree.reinforce", "LossSequenceTrainingStats")
    Reinforce = _ref(ref, "tianshou.algorithm.modelfree.reinforce", "Reinforce")
    SequenceSummaryStats = _ref(ref, "tianshou.data", "SequenceSummaryStats")

    from . import npg as NG
    from . import reinforce as RF

    who = "HipReinforce"

    class HipReinforce(_HipGlue, Reinforce):
        def __init__(self, *args, device="cuda", **kwargs):
            """Net[h, h] tanh (h a multiple of 32) under an unbounded actor with plain Adam -- the nets of
            examples/mujoco/mujoco_reinforce.py -- runs on the fused step kernel / the Net[h, h] GEMM path as before; every
            other `Net(hidden_sizes=[...], activation=Tanh | ReLU | None)` trunk, the reference's default bounded actor
            (max_action * tanh), Adam with weight decay and RMSprop (optim.py:89-140) take the per-layer engine
            (`reinforce.NetReinforceEngine`, round 6), and so do `Net(norm_layer=nn.LayerNorm)` trunks (common.py:25-39).
            conditioned_sigma and other norm layers raise."""
            super().__init__(*args, **kwargs)
            self._hip_device = torch.device(device)
            actor = self.policy.actor
            sa = actor.state_dict()
            if getattr(actor, "_c_sigma", True):
                raise NotImplementedError(f"{who}: the actor needs a state-independent sigma_param (no conditioned sigma)")
            stems, hidden_sizes, act_name = _trunk_spec(actor, "actor", norm=True)   # (raises for anything but Linear + Tanh / ReLU)
            nm = _trunk_norm(actor, "actor")                                         # MLP(norm_layer=nn.LayerNorm): per-layer engine
            trunk_keys = ([f"{st}.{x}" for st in stems for x in ("weight", "bias")] if nm is None else
                          [f"{x}.{y}" for a, b in zip(stems, nm[0]) for x in (a, b) for y in ("weight", "bias")])
            self._hip_keys = trunk_keys + ["mu.model.0.weight", "mu.model.0.bias", "sigma_param"]
            if set(sa.keys()) != set(self._hip_keys) or sa["mu.model.0.weight"].shape[1] != hidden_sizes[-1]:
                raise NotImplementedError(f"{who}: the actor must be ContinuousActorProbabilistic over a Net trunk with a single-Linear mu head")
            of = optimizer_fields(self.optim._optim)
            bounded = not getattr(actor, "_unbounded", False)
            plain = of["optimizer"] == "adam" and not of["weight_decay"]
            legacy = (nm is None and act_name == "tanh" and len(hidden_sizes) == 2 and hidden_sizes[0] == hidden_sizes[1]
                      and hidden_sizes[0] % 32 == 0 and not bounded and plain and self._hip_keys == list(TIANSHOU_ACTOR_KEYS))
            if not legacy and (len(hidden_sizes) > 7 or max(hidden_sizes) > 1024 or sa["mu.model.0.weight"].shape[0] > 32):
                raise NotImplementedError(f"{who}: trunks of up to 7 hidden layers of at most 1024 units, at most 32 actions")
            self._hip_kind = "legacy" if legacy else "net"
            self._hip_net = (hidden_sizes, act_name, float(actor.max_action) if bounded else None, of)
            self._hip_ln = None if nm is None else float(nm[1])
            self._hip_engine = None
            self._hip_glue_init()

        def _dims(self):
            sa = self.policy.actor.state_dict()
            hidden, obs_dim = sa[self._hip_keys[0]].shape
            return obs_dim, hidden, sa["mu.model.0.weight"].shape[0]

        def _to_flat(self, tensors):
            obs_dim, hidden, act_dim = self._dims()
            if self._hip_kind == "legacy":
                return NG.actor_flat_from_torch(tensors, obs_dim, hidden, act_dim, self._hip_device)
            from .ppo_wide import net_flat_from_tensors

            return net_flat_from_tensors(list(tensors), obs_dim, list(self._hip_net[0]), act_dim, self._hip_device,
                                         layer_norm=self._hip_ln is not None)

        def _from_flat(self, flat):
            obs_dim, hidden, act_dim = self._dims()
            if self._hip_kind == "legacy":
                return NG.actor_flat_to_torch(flat, obs_dim, hidden, act_dim)
            from .ppo_wide import net_flat_to_tensors

            return net_flat_to_tensors(flat, obs_dim, list(self._hip_net[0]), act_dim, True, layer_norm=self._hip_ln is not None)

        def _engine(self):
            if self._hip_engine is None:
                sa = self.policy.actor.state_dict()
                dims = self._dims()
                opt = self.optim._optim
                of = self._hip_net[3]
                drc = self.discounted_return_computation
                cfg = RF.ReinforceConfig(gamma=drc.gamma, return_standardization=drc.return_standardization, lr=of["lr"],
                                         betas=tuple(of.get("betas", (0.9, 0.999))), adam_eps=of["adam_eps"],
                                         max_grad_norm=self.optim._max_grad_norm)
                flat = self._to_flat([sa[k] for k in self._hip_keys])
                if self._hip_kind == "legacy":
                    eng = self._hip_engine = RF.ReinforceEngine(dims[0], dims[2], dims[1], flat, cfg)
                else:
                    hidden_sizes, act_name, max_action, _ = self._hip_net
                    eng = self._hip_engine = RF.NetReinforceEngine(dims[0], dims[2], hidden_sizes, act_name, flat, cfg,
                                                                   max_action=max_action, optimizer=of,
                                                                   layer_norm=self._hip_ln is not None, ln_eps=self._hip_ln or 1e-5)
                eng.ret_rms = [float(drc.ret_rms.mean), float(drc.ret_rms.var), float(drc.ret_rms.count)]
                ms, vs, step = adam_state(opt, params_by_keys(self.policy.actor, self._hip_keys))     # resume
                eng.adam_m, eng.adam_v = self._to_flat(ms), self._to_flat(vs)
                eng.adam_step = step
            return self._hip_engine

        def _preprocess_batch(self, batch, buffer, indices):
            _require_gpu(self._hip_device, who)
            eng = self._engine()
            dev = self._hip_device
            t = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=dev)  # noqa: E731
            cut = np.nonzero(np.isin(indices, buffer.unfinished_index()))[0]      # algorithm_base.py:715
            batch.returns = eng.preprocess(t(batch.rew).to(torch.float64), t(batch.terminated), t(batch.truncated), t(cut))
            self._hip_obs, self._hip_act = t(batch.obs).to(torch.float32), t(batch.act).to(torch.float32)
            drc = self.discounted_return_computation
            drc.ret_rms.mean, drc.ret_rms.var, drc.ret_rms.count = eng.ret_rms
            return batch

        def _update_with_batch(self, batch, batch_size, repeat):
            self._hip_refresh_lr()
            eng = self._hip_engine
            perms = [np.random.permutation(len(batch)) for _ in range(repeat)]     # Batch.split, batch.py:1209
            losses, _ = eng.update(self._hip_obs, self._hip_act, batch.returns, batch_size, repeat, perms)
            arr = losses.cpu().numpy().astype(np.float64).reshape(-1)              # one D2H per update()
            aparams = params_by_keys(self.policy.actor, self._hip_keys)
            with torch.no_grad():
                for p, t in zip(aparams, self._from_flat(eng.actor)):
                    p.copy_(t.reshape(p.shape).to(p.device))
            store_adam_state(self.optim._optim, aparams, self._from_flat(eng.adam_m), self._from_flat(eng.adam_v), eng.adam_step)
            return LossSequenceTrainingStats(loss=SequenceSummaryStats.from_sequence(arr))

    return HipReinforce


def make_hip_a2c():
    """HipA2C(A2C) on the MuJoCo actor-critic of HipPPO (a2c.py:249-290 = the fused step kernel's algo 1)."""
    return make_hip_ppo("a2c")


def make_hip_a2c_discrete():
    """HipA2CDiscrete(A2C) on the shared-trunk MLP of HipPPODiscrete."""
    return make_hip_ppo_discrete("a2c")


def make_hip_a2c_cnn():
    """HipA2CCnn(A2C) on the Atari actor-critic of HipPPOCnn."""
    return make_hip_ppo_cnn("a2c")


# ---------------------------------------------------------------------------------------------------
# DQN (dqn.py:288-404) on DQNet
# ---------------------------------------------------------------------------------------------------
def _require_gpu(device: torch.device, who: str) -> None:
    if device.type != "cuda":
        raise RuntimeError(f"{who} needs an MI355X (device='cuda'); there is no CPU fallback")


def _adam_of(optim):
    opt = optim._optim
    g = opt.param_groups[0]
    if type(opt).__name__ != "Adam" or g.get("weight_decay", 0) != 0 or g.get("amsgrad", False):
        raise NotImplementedError("the HIP engines support torch.optim.Adam without weight decay / amsgrad")
    return opt, g


def _mirror(algorithm, buffer, device):
    """Device mirror of the host replay buffer, refreshed incrementally on every update()."""
    from .buffer import DeviceReplayBuffer

    m = getattr(algorithm, "_hip_mirror", None)
    if m is None or algorithm._hip_mirror_src is not buffer:
        shard = algorithm._hip_shard_range(buffer) if hasattr(algorithm, "_hip_shard_range") else None
        m = DeviceReplayBuffer.from_tianshou(buffer, device=device, env_range=shard)
        algorithm._hip_mirror, algorithm._hip_mirror_src = m, buffer
    else:
        m.sync_from_tianshou(buffer)
    return m


def make_hip_dqn(ref=None):
    """Returns HipDQN(DQN): `_preprocess_batch` / `_update_with_batch` (dqn.py:257-275, 381-404) on the engine.
    Supported model: DQNet(c, h, w, n_act) (atari_network.py:60-122), Adam; buffer either stores whole [c, h, w]
    observations or single frames with stack_num = c (save_only_last_obs); obs_next optional.
    `ref`: optional namespace replacing the tianshou imports (see `_ref`)."""
    DQN = _ref(ref, "tianshou.algorithm.modelfree.dqn", "DQN")
    SimpleLossTrainingStats = _ref(ref, "tianshou.algorithm.modelfree.reinforce", "SimpleLossTrainingStats")
    Batch = _ref(ref, "tianshou.data", "Batch")

    from . import dqn as D

    class HipDQN(_HipGlue, DQN):
        def __init__(self, *args, device="cuda", data_parallel=False, group=None, allreduce=None, policy_forward="hip",
                     write_back="auto", host_batch=False, **kwargs):
            """`data_parallel=True`: one process per GPU, every rank samples its own minibatch from its own buffer (its
            envs) and `_update_with_batch` all-reduces the flat gradient + loss (`DataParallelDQN`); PER priorities stay
            rank-local.  `allreduce`: None = torch.distributed, "native" = the C-ABI RCCL exchange, or a callable.
            `policy_forward="hip"`: `DiscreteQLearningPolicy.forward` (dqn.py:101-143), which the Collector
            calls per vector step, runs on `ts_dqn_forward` with the engine's parameters (`tianshou_amd.policy`); the
            epsilon-greedy `add_exploration_noise` stays the reference's.
            Hook-level throughput (as HipSAC's): `host_batch=False` lets `update()` sample indices only -- the hooks read frames,
            actions and rewards from the device mirror by index, the host copy `buffer.sample()` makes of the batch (two stacked
            observations per transition: 29 MB for 512 Atari transitions) is never looked at (True: the reference's own
            `Algorithm._update`); `write_back="auto"` keeps the updates in the engine until somebody reads the torch modules (lazy
            whenever the policy forward is the engine's; "eager" = the two networks and the optimizer state after every update,
            27 MB of device copies in ~40 torch ops; `hip_sync()` forces it)."""
            super().__init__(*args, **kwargs)
            self._hip_device = torch.device(device)
            self._hip_host_batch = bool(host_batch)
            sd = self.policy.model.state_dict()
            if list(sd.keys()) != D.TIANSHOU_KEYS:
                raise NotImplementedError("HipDQN: the model must be DQNet(c, h, w, action_shape) without extra layers")
            _adam_of(self.optim)
            self._hip_engine = None
            self._hip_glue_init()
            self._hip_dp_setup(data_parallel, group, allreduce, shard_buffer=False)
            if policy_forward not in ("hip", "torch"):
                raise ValueError("policy_forward must be 'hip' or 'torch'")
            if policy_forward == "hip":
                from . import policy as HP

                HP.attach(self.policy, "q", self, device=str(self._hip_device), n_act=int(sd[D.TIANSHOU_KEYS[-1]].numel()))
            self._hip_set_write_back(write_back, attached=policy_forward == "hip")

        def update(self, buffer, sample_size):
            if self._hip_host_batch or buffer is None:
                return super().update(buffer, sample_size)
            return self._hip_offpolicy_update(buffer, sample_size, Batch)

        def _engine(self, c, h, w):
            if self._hip_engine is None:
                sd = self.policy.model.state_dict()
                n_act = sd[D.TIANSHOU_KEYS[-1]].numel()
                opt, g = _adam_of(self.optim)
                cfg = D.DQNConfig(gamma=self.gamma, n_step=self.n_step, target_update_freq=self.target_update_freq,
                                  is_double=self.is_double, huber_delta=self.huber_loss_delta, lr=g["lr"],
                                  betas=tuple(g["betas"]), adam_eps=g["eps"], max_grad_norm=self.optim._max_grad_norm)
                flat = D.flat_from_torch([sd[k] for k in D.TIANSHOU_KEYS], c, h, w, n_act, self._hip_device)
                eng = self._hip_engine = D.DQNEngine(c, h, w, n_act, flat, cfg)
                eng.iter = self._iter
                ms, vs, step = adam_state(opt, list(self.policy.model.parameters()))       # resume from a checkpoint
                eng.adam_m = D.flat_from_torch(ms, c, h, w, n_act, self._hip_device)
                eng.adam_v = D.flat_from_torch(vs, c, h, w, n_act, self._hip_device)
                eng.adam_step = step
                if eng.params_old is not None:
                    old = [p.detach() for p in self.model_old.parameters()]
                    eng.params_old = D.flat_from_torch(old, c, h, w, n_act, self._hip_device)
            return self._hip_engine

        def _layout(self, buffer):
            obs = np.asarray(buffer.obs)
            stack = int(getattr(buffer, "stack_num", 1))
            if stack > 1:
                if obs.ndim != 3:
                    raise NotImplementedError("HipDQN: frame stacking needs single [h, w] frames per slot")
                return stack, obs.shape[1], obs.shape[2], stack
            if obs.ndim != 4:
                raise NotImplementedError("HipDQN: observations must be [c, h, w]")
            return obs.shape[1], obs.shape[2], obs.shape[3], 1

        def _preprocess_batch(self, batch, buffer, indices):
            _require_gpu(self._hip_device, "HipDQN")
            c, h, w, stack = self._layout(buffer)
            eng = self._engine(c, h, w)
            m = _mirror(self, buffer, self._hip_device)
            idx = torch.as_tensor(np.asarray(indices, np.int64), device=self._hip_device)
            self._hip_idx, self._hip_stack = idx, stack
            if hasattr(batch, "weight"):
                batch.weight = torch.as_tensor(np.asarray(batch.weight), dtype=torch.float32, device=self._hip_device)
            # Inside HipDQN.update()'s own sequence (`_hip_offpolicy_update`: nobody reads the batch between the two hooks) on the
            # Atari layout (single uint8 frames, stack 4, no stored obs_next) the two hooks are ONE library call, made by
            # `_update_with_batch` (ts_dqn_learn_rows; `batch.returns` is attached there).  Called on its own (the reference's
            # `_update`, `host_batch=True`, data parallel, other layouts) the hook computes the returns here.
            self.__dict__["_hip_deferred"] = (self.__dict__.get("_hip_own_sequence", False) and not self._hip_dp_on and stack == c
                                              and hasattr(eng, "rows_ok") and eng.rows_ok(m, m.obs, m.act)
                                              and not os.environ.get("TS_DQN_TWO_CALLS"))
            if self.__dict__["_hip_deferred"]:
                return batch
            nxt = m.obs_next if m.obs_next is not None else None
            # the batch's own observations, gathered here so that Q_online(batch.obs) of _update_with_batch can run beside the
            # two obs_next passes of _target_q (DQNEngine.prefetch_forward); data-parallel runs keep the plain order
            # (on a frame buffer without obs_next, one launch gathers both stacked observations: DQNEngine.preprocess_with_obs)
            self._hip_obs, ret = eng.preprocess_with_obs(m, m.obs, idx, stack, obs_next_frames=nxt,
                                                         prefetch=not self._hip_dp_on)
            batch.returns = ret.reshape(-1, 1)
            return batch

        def _update_with_batch(self, batch):
            self._hip_refresh_lr()
            eng, m = self._hip_engine, self._hip_mirror
            weight = batch.pop("weight", None)
            if self.__dict__.pop("_hip_deferred", False):
                loss, td, ret = eng.learn_rows(m, m.obs, m.act, self._hip_idx, weight)
                batch.returns = ret.reshape(-1, 1)
                self._iter = eng.iter
                batch.weight = td                                                 # prio-buffer, dqn.py:401
                self._hip_after_update()
                return SimpleLossTrainingStats(loss=float(loss.item()))
            obs = self._hip_obs
            # (index-only sampling: the batch carries no host copy of the actions; the mirror's rows are the same values)
            act = torch.as_tensor(np.asarray(batch.act), device=self._hip_device) if hasattr(batch, "act") else m.act[self._hip_idx]
            runner = eng
            if self._hip_dp_on:
                from .distributed import DataParallelDQN

                runner = self._hip_dp(DataParallelDQN, eng)
            loss, td = runner.update_with_batch(obs, act, batch.returns.reshape(-1), weight)
            self._iter = eng.iter
            batch.weight = td                                                     # prio-buffer, dqn.py:401
            self._hip_after_update()                                              # write-back now ("eager") or when read ("lazy")
            return SimpleLossTrainingStats(loss=float(loss.item()))

        def _hip_write_back(self) -> None:
            eng = self.__dict__.get("_hip_engine_obj")
            if eng is None:
                return
            dims = (eng.c, eng.h, eng.w, eng.n_act)
            with torch.no_grad():
                for p, t in zip(self.policy.model.parameters(), D.flat_to_torch(eng.params, *dims)):
                    self._hip_put(p, t)
                if eng.params_old is not None:
                    for p, t in zip(self.model_old.parameters(), D.flat_to_torch(eng.params_old, *dims)):
                        self._hip_put(p, t)
            store_adam_state(self.optim._optim, list(self.policy.model.parameters()), D.flat_to_torch(eng.adam_m, *dims),
                             D.flat_to_torch(eng.adam_v, *dims), eng.adam_step)

    return HipDQN


# ---------------------------------------------------------------------------------------------------
# DQN on the Recurrent Q network (DRQN, test/discrete/test_drqn.py)
# ---------------------------------------------------------------------------------------------------
def make_hip_drqn(ref=None):
    """Returns HipDRQN(DQN): the DQN hooks (dqn.py:257-275, 381-404) on the engine for a Recurrent model
    (utils/net/common.py:372-452) over a buffer with stack_num (the LSTM's sequence length) and vector observations.
    `ref`: optional namespace replacing the tianshou imports (see `_ref`)."""
    DQN = _ref(ref, "tianshou.algorithm.modelfree.dqn", "DQN")
    SimpleLossTrainingStats = _ref(ref, "tianshou.algorithm.modelfree.reinforce", "SimpleLossTrainingStats")

    from . import dqn as D
    from . import drqn as R

    class HipDRQN(_HipGlue, DQN):
        def __init__(self, *args, device="cuda", **kwargs):
            super().__init__(*args, **kwargs)
            self._hip_device = torch.device(device)
            sd = self.policy.model.state_dict()
            layers = sum(1 for k in sd if k.startswith("nn.weight_ih_l"))
            if layers < 1 or list(sd.keys()) != R.state_dict_keys(layers):
                raise NotImplementedError("HipDRQN: the model must be Recurrent(layer_num, state_shape, action_shape, hidden)")
            hidden, obs_dim = sd["fc1.weight"].shape
            n_act = sd["fc2.weight"].shape[0]
            if hidden % 32 or not 32 <= hidden <= 1024 or layers > 8 or n_act > 32:
                raise NotImplementedError("HipDRQN: hidden a multiple of 32 in [32, 1024], at most 8 layers and 32 actions")
            self._hip_dims = (obs_dim, hidden, layers, n_act)
            _adam_of(self.optim)
            self._hip_engine = None
            self._hip_glue_init()

        def _engine(self):
            if self._hip_engine is None:
                sd = self.policy.model.state_dict()
                dims, dev = self._hip_dims, self._hip_device
                keys = R.state_dict_keys(dims[2])
                opt, g = _adam_of(self.optim)
                cfg = D.DQNConfig(gamma=self.gamma, n_step=self.n_step, target_update_freq=self.target_update_freq,
                                  is_double=self.is_double, huber_delta=self.huber_loss_delta, lr=g["lr"],
                                  betas=tuple(g["betas"]), adam_eps=g["eps"], max_grad_norm=self.optim._max_grad_norm)
                eng = self._hip_engine = R.RecurrentDQNEngine(*dims, R.flat_from_torch([sd[k] for k in keys], *dims, dev), cfg)
                eng.iter = self._iter
                ms, vs, step = adam_state(opt, params_by_keys(self.policy.model, keys))       # resume from a checkpoint
                eng.adam_m, eng.adam_v = R.flat_from_torch(ms, *dims, dev), R.flat_from_torch(vs, *dims, dev)
                eng.adam_step = step
                if eng.params_old is not None:
                    old = getattr(self.model_old, "module", self.model_old).state_dict()       # EvalModeModuleWrapper
                    eng.params_old = R.flat_from_torch([old[k] for k in keys], *dims, dev)
            return self._hip_engine

        def _preprocess_batch(self, batch, buffer, indices):
            _require_gpu(self._hip_device, "HipDRQN")
            obs = np.asarray(buffer.obs)
            if obs.ndim != 2 or obs.shape[1] != self._hip_dims[0]:
                raise NotImplementedError("HipDRQN: the buffer must hold vector observations of the model's state_shape")
            eng = self._engine()
            m = _mirror(self, buffer, self._hip_device)
            idx = torch.as_tensor(np.asarray(indices, np.int64), device=self._hip_device)
            stack = int(getattr(buffer, "stack_num", 1))
            nxt = m.obs_next if m.obs_next is not None else None
            # the batch's own stacked observations are gathered here, so that their forward pass (the one _update_with_batch needs)
            # can run on a side stream beside the two obs_next passes of _target_q; data-parallel runs keep the plain order
            self._hip_obs, ret = eng.preprocess_with_obs(m, m.obs, idx, stack, obs_next_rows=nxt, prefetch=not self._hip_dp_on)
            batch.returns = ret.reshape(-1, 1)
            self._hip_idx, self._hip_stack = idx, stack
            if hasattr(batch, "weight"):
                batch.weight = torch.as_tensor(np.asarray(batch.weight), dtype=torch.float32, device=self._hip_device)
            return batch

        def _update_with_batch(self, batch):
            self._hip_refresh_lr()
            eng, m = self._hip_engine, self._hip_mirror
            weight = batch.pop("weight", None)
            obs = self._hip_obs
            act = torch.as_tensor(np.asarray(batch.act), device=self._hip_device)
            runner = eng
            if self._hip_dp_on:
                from .distributed import DataParallelDQN

                runner = self._hip_dp(DataParallelDQN, eng)
            loss, td = runner.update_with_batch(obs, act, batch.returns.reshape(-1), weight)
            self._iter = eng.iter
            batch.weight = td                                                     # prio-buffer, dqn.py:401
            dims = self._hip_dims
            keys = R.state_dict_keys(dims[2])
            params = params_by_keys(self.policy.model, keys)
            with torch.no_grad():
                for p, t in zip(params, R.flat_to_torch(eng.params, *dims)):
                    p.copy_(t)
                if eng.params_old is not None:
                    old_mod = getattr(self.model_old, "module", self.model_old)
                    for p, t in zip(params_by_keys(old_mod, keys), R.flat_to_torch(eng.params_old, *dims)):
                        p.copy_(t)
            store_adam_state(self.optim._optim, params, R.flat_to_torch(eng.adam_m, *dims), R.flat_to_torch(eng.adam_v, *dims),
                             eng.adam_step)
            return SimpleLossTrainingStats(loss=float(loss.item()))

    return HipDRQN


# ---------------------------------------------------------------------------------------------------
# QRDQN (qrdqn.py) / C51 (c51.py) on QRDQNet / C51Net
# ---------------------------------------------------------------------------------------------------
def _make_hip_distq(kind: str, ref=None):
    from . import distq as Q
    from . import dqn as D

    SimpleLossTrainingStats = _ref(ref, "tianshou.algorithm.modelfree.reinforce", "SimpleLossTrainingStats")
    LossSequenceTrainingStats = _ref(ref, "tianshou.algorithm.modelfree.reinforce", "LossSequenceTrainingStats")
    if kind == Q.QR:
        Base = _ref(ref, "tianshou.algorithm.modelfree.qrdqn", "QRDQN")
    else:
        Base = _ref(ref, "tianshou.algorithm.modelfree.c51", "C51")
    who = "HipQRDQN" if kind == Q.QR else "HipC51"

    class HipDistQ(_HipGlue, Base):
        def __init__(self, *args, device="cuda", **kwargs):
            super().__init__(*args, **kwargs)
            self._hip_device = torch.device(device)
            if list(self.policy.model.state_dict().keys()) != D.TIANSHOU_KEYS:
                raise NotImplementedError(f"{who}: the model must be QRDQNet / C51Net (DQNet without extra layers)")
            _adam_of(self.optim)
            self._hip_engine = None
            self._hip_glue_init()

        def _n_atoms(self) -> int:
            return int(self.num_quantiles if kind == Q.QR else self.policy.num_atoms)

        def _engine(self, c, h, w):
            if self._hip_engine is None:
                sd = self.policy.model.state_dict()
                n_atoms = self._n_atoms()
                n_out = sd[D.TIANSHOU_KEYS[-1]].numel()
                if n_out % n_atoms:
                    raise NotImplementedError(f"{who}: head width {n_out} is not a multiple of {n_atoms} atoms")
                n_act = n_out // n_atoms
                opt, g = _adam_of(self.optim)
                cfg = Q.DistQConfig(kind=kind, n_atoms=n_atoms, gamma=self.gamma, n_step=self.n_step,
                                    target_update_freq=self.target_update_freq, lr=g["lr"], betas=tuple(g["betas"]),
                                    adam_eps=g["eps"], max_grad_norm=self.optim._max_grad_norm,
                                    v_min=float(getattr(self.policy, "v_min", -10.0)),
                                    v_max=float(getattr(self.policy, "v_max", 10.0)))
                dev = self._hip_device
                flat = Q.flat_from_torch([sd[k] for k in D.TIANSHOU_KEYS], c, h, w, n_act, n_atoms, dev)
                eng = self._hip_engine = Q.DistQEngine(c, h, w, n_act, flat, cfg)
                eng.iter = self._iter
                ms, vs, step = adam_state(opt, list(self.policy.model.parameters()))       # resume from a checkpoint
                eng.adam_m = Q.flat_from_torch(ms, c, h, w, n_act, n_atoms, dev)
                eng.adam_v = Q.flat_from_torch(vs, c, h, w, n_act, n_atoms, dev)
                eng.adam_step = step
                if eng.params_old is not None:
                    old = [p.detach() for p in self.model_old.parameters()]
                    eng.params_old = Q.flat_from_torch(old, c, h, w, n_act, n_atoms, dev)
            return self._hip_engine

        @staticmethod
        def _layout(buffer):
            obs = np.asarray(buffer.obs)
            stack = int(getattr(buffer, "stack_num", 1))
            if stack > 1:
                if obs.ndim != 3:
                    raise NotImplementedError(f"{who}: frame stacking needs single [h, w] frames per slot")
                return stack, obs.shape[1], obs.shape[2], stack
            if obs.ndim != 4:
                raise NotImplementedError(f"{who}: observations must be [c, h, w]")
            return obs.shape[1], obs.shape[2], obs.shape[3], 1

        def _preprocess_batch(self, batch, buffer, indices):
            _require_gpu(self._hip_device, who)
            c, h, w, stack = self._layout(buffer)
            eng = self._engine(c, h, w)
            m = _mirror(self, buffer, self._hip_device)
            idx = torch.as_tensor(np.asarray(indices, np.int64), device=self._hip_device)
            batch.returns = eng.preprocess(m, m.obs, idx, stack, obs_next_frames=m.obs_next)
            self._hip_idx, self._hip_stack = idx, stack
            if hasattr(batch, "weight"):
                batch.weight = torch.as_tensor(np.asarray(batch.weight), dtype=torch.float32, device=self._hip_device)
            return batch

        def _update_with_batch(self, batch):
            self._hip_refresh_lr()
            eng, m = self._hip_engine, self._hip_mirror
            idx, stack = self._hip_idx, self._hip_stack
            weight = batch.pop("weight", None)
            obs = D.gather_obs_nhwc(m.obs, m, idx, stack, as_u8=True)
            obs_next = None
            if kind == Q.C51:                     # batch.obs_next = buffer[indices].obs_next (buffer_base.py:624-626)
                if m.obs_next is not None:
                    obs_next = D.gather_obs_nhwc(m.obs_next, m, idx, stack, as_u8=True)
                else:
                    obs_next = D.gather_obs_nhwc(m.obs, m, m.next(idx), stack, as_u8=True)
            act = torch.as_tensor(np.asarray(batch.act), device=self._hip_device)
            loss, prio = eng.update_with_batch(obs, act, batch.returns, weight, obs_next_nhwc=obs_next)
            self._iter = eng.iter
            batch.weight = prio                                                   # prio-buffer, qrdqn.py:128 / c51.py:157
            dims = (eng.c, eng.h, eng.w, eng.n_act, eng.cfg.n_atoms)
            with torch.no_grad():
                for p, t in zip(self.policy.model.parameters(), Q.flat_to_torch(eng.params, *dims)):
                    p.copy_(t)
                if eng.params_old is not None:
                    for p, t in zip(self.model_old.parameters(), Q.flat_to_torch(eng.params_old, *dims)):
                        p.copy_(t)
            store_adam_state(self.optim._optim, list(self.policy.model.parameters()), Q.flat_to_torch(eng.adam_m, *dims),
                             Q.flat_to_torch(eng.adam_v, *dims), eng.adam_step)
            if kind == Q.QR:
                return SimpleLossTrainingStats(loss=float(loss.item()))
            return LossSequenceTrainingStats(loss=float(loss.item()))            # as c51.py:160

    HipDistQ.__name__ = HipDistQ.__qualname__ = who
    return HipDistQ


def make_hip_qrdqn(ref=None):
    """Returns HipQRDQN(QRDQN): `_preprocess_batch` / `_update_with_batch` (dqn.py:257-275, qrdqn.py:93-131) on the
    engine.  Supported model: QRDQNet (atari_network.py:211-235), Adam; buffer layouts as HipDQN.
    `ref`: optional namespace replacing the tianshou imports (see `_ref`)."""
    return _make_hip_distq("qr", ref)


def make_hip_c51(ref=None):
    """Returns HipC51(C51): `_preprocess_batch` / `_update_with_batch` (dqn.py:257-275, c51.py:120-160) on the engine.
    Supported model: C51Net (atari_network.py:125-151) with C51Policy's support, Adam; buffer layouts as HipDQN.
    `ref`: optional namespace replacing the tianshou imports (see `_ref`)."""
    return _make_hip_distq("c51", ref)

def make_hip_rainbow(ref=None):
    """Returns HipRainbow(RainbowDQN): `_preprocess_batch` / `_update_with_batch` (dqn.py:257-275, rainbow.py:93-101 ->
    c51.py:120-160) on the engine.  Supported model: RainbowNet(is_dueling=True, is_noisy=True)
    (atari_network.py:154-208) with C51Policy's support, Adam; buffer layouts as HipDQN.  The NoisyLinear noise is drawn by
    the torch modules themselves (`RainbowDQN._sample_noise`, so torch's generator advances as in the reference) and
    handed to the engine.  `ref`: optional namespace replacing the tianshou imports (see `_ref`)."""
    RainbowDQN = _ref(ref, "tianshou.algorithm.modelfree.rainbow", "RainbowDQN")
    LossSequenceTrainingStats = _ref(ref, "tianshou.algorithm.modelfree.reinforce", "LossSequenceTrainingStats")

    from . import distq as Q
    from . import dqn as D
    from . import rainbow as RB

    class HipRainbow(_HipGlue, RainbowDQN):
        def __init__(self, *args, device="cuda", **kwargs):
            super().__init__(*args, **kwargs)
            self._hip_device = torch.device(device)
            keys = list(self.policy.model.state_dict().keys())
            if sorted(keys) != sorted(RB.TIANSHOU_KEYS + RB.NOISE_KEYS):
                raise NotImplementedError("HipRainbow: the model must be RainbowNet(is_dueling=True, is_noisy=True)")
            _adam_of(self.optim)
            self._hip_engine = None
            self._hip_glue_init()

        def _noise_of(self, model, dims):
            sd = model.state_dict()
            return RB.noise_from_torch([sd[k] for k in RB.NOISE_KEYS], *dims, self._hip_device)

        def _engine(self, c, h, w):
            if self._hip_engine is None:
                model = self.policy.model
                sd = model.state_dict()
                n_atoms = int(self.policy.num_atoms)
                n_act = sd["Q.2.mu_bias"].numel() // n_atoms
                dims = (c, h, w, n_act, n_atoms)
                opt, g = _adam_of(self.optim)
                cfg = Q.DistQConfig(kind=Q.C51, n_atoms=n_atoms, gamma=self.gamma, n_step=self.n_step,
                                    target_update_freq=self.target_update_freq, lr=g["lr"], betas=tuple(g["betas"]),
                                    adam_eps=g["eps"], max_grad_norm=self.optim._max_grad_norm,
                                    v_min=float(self.policy.v_min), v_max=float(self.policy.v_max))
                dev = self._hip_device
                eng = self._hip_engine = RB.RainbowEngine(c, h, w, n_act, RB.flat_from_torch([sd[k] for k in RB.TIANSHOU_KEYS], *dims, dev),
                                                          self._noise_of(model, dims), cfg)
                eng.iter = self._iter
                ms, vs, step = adam_state(opt, params_by_keys(model, RB.TIANSHOU_KEYS))          # resume from a checkpoint
                eng.adam_m, eng.adam_v = RB.flat_from_torch(ms, *dims, dev), RB.flat_from_torch(vs, *dims, dev)
                eng.adam_step = step
                if eng.params_old is not None:
                    so = self.model_old.state_dict()
                    eng.params_old = RB.flat_from_torch([so[k] for k in RB.TIANSHOU_KEYS], *dims, dev)
                    eng.noise_old = self._noise_of(self.model_old, dims)
            return self._hip_engine

        @staticmethod
        def _layout(buffer):
            obs = np.asarray(buffer.obs)
            stack = int(getattr(buffer, "stack_num", 1))
            if stack > 1:
                if obs.ndim != 3:
                    raise NotImplementedError("HipRainbow: frame stacking needs single [h, w] frames per slot")
                return stack, obs.shape[1], obs.shape[2], stack
            if obs.ndim != 4:
                raise NotImplementedError("HipRainbow: observations must be [c, h, w]")
            return obs.shape[1], obs.shape[2], obs.shape[3], 1

        def _preprocess_batch(self, batch, buffer, indices):
            _require_gpu(self._hip_device, "HipRainbow")
            c, h, w, stack = self._layout(buffer)
            eng = self._engine(c, h, w)
            m = _mirror(self, buffer, self._hip_device)
            idx = torch.as_tensor(np.asarray(indices, np.int64), device=self._hip_device)
            batch.returns = eng.preprocess(m, idx)
            self._hip_idx, self._hip_stack = idx, stack
            if hasattr(batch, "weight"):
                batch.weight = torch.as_tensor(np.asarray(batch.weight), dtype=torch.float32, device=self._hip_device)
            return batch

        def _update_with_batch(self, batch):
            self._hip_refresh_lr()
            eng, m = self._hip_engine, self._hip_mirror
            idx, stack = self._hip_idx, self._hip_stack
            dims = (eng.c, eng.h, eng.w, eng.n_act, eng.cfg.n_atoms)
            self._sample_noise(self.policy.model)                                 # rainbow.py:97-100
            noise_old = None
            if self.use_target_network:
                self._sample_noise(self.model_old)
                noise_old = self._noise_of(self.model_old, dims)
            eng.set_noise(self._noise_of(self.policy.model, dims), noise_old)
            weight = batch.pop("weight", None)
            obs = D.gather_obs_nhwc(m.obs, m, idx, stack, as_u8=True)
            if m.obs_next is not None:                     # batch.obs_next = buffer[indices].obs_next (buffer_base.py:624-626)
                obs_next = D.gather_obs_nhwc(m.obs_next, m, idx, stack, as_u8=True)
            else:
                obs_next = D.gather_obs_nhwc(m.obs, m, m.next(idx), stack, as_u8=True)
            act = torch.as_tensor(np.asarray(batch.act), device=self._hip_device)
            loss, prio = eng.update_with_batch(obs, act, batch.returns, obs_next, weight)
            self._iter = eng.iter
            batch.weight = prio                                                   # prio-buffer, c51.py:157
            model = self.policy.model
            with torch.no_grad():
                for p, t in zip(params_by_keys(model, RB.TIANSHOU_KEYS), RB.flat_to_torch(eng.params, *dims)):
                    p.copy_(t)
                if eng.params_old is not None:
                    for p, t in zip(params_by_keys(self.model_old, RB.TIANSHOU_KEYS), RB.flat_to_torch(eng.params_old, *dims)):
                        p.copy_(t)
                    if (eng.iter - 1) % eng.cfg.target_update_freq == 0:          # the sync carried the noise along
                        so, sn = self.model_old.state_dict(), model.state_dict()
                        for k in RB.NOISE_KEYS:
                            so[k].copy_(sn[k])
            store_adam_state(self.optim._optim, params_by_keys(model, RB.TIANSHOU_KEYS), RB.flat_to_torch(eng.adam_m, *dims),
                             RB.flat_to_torch(eng.adam_v, *dims), eng.adam_step)
            return LossSequenceTrainingStats(loss=float(loss.item()))            # as c51.py:160

    return HipRainbow



# ---------------------------------------------------------------------------------------------------
# SAC (sac.py:213-336) on the mujoco_sac.py networks
# ---------------------------------------------------------------------------------------------------
def _trunk_activation(mods, who: str) -> str:
    """The off-policy engines compute a `Net` trunk as Sequential(Linear, act, Linear, act, ...) with act = nn.ReLU -- `Net`'s
    default activation (utils/net/common.py:246-369) -- or nn.Tanh, the same for every network of the algorithm.  The state_dict
    keys only pin the Linear layers' positions; whatever sits between them (another activation, dropout, a norm layer without
    affine parameters) would be silently replaced, so the modules are checked here.  -> "relu" | "tanh"."""
    names = set()
    for mod in mods:
        seq = getattr(getattr(getattr(mod, "preprocess", None), "model", None), "model", None)
        ms = list(seq) if seq is not None else []
        ok = len(ms) >= 2 and len(ms) % 2 == 0 and all(
            type(ms[i]).__name__ in ("Linear", "EnsembleLinear") and type(ms[i + 1]) in (torch.nn.ReLU, torch.nn.Tanh) for i in range(0, len(ms), 2))
        if not ok:
            raise NotImplementedError(f"{who}: the trunk must be Net(hidden_sizes=[...]) with nn.ReLU or nn.Tanh after every Linear "
                                      f"layer (got {[type(m).__name__ for m in ms]})")
        names |= {"relu" if isinstance(ms[i], torch.nn.ReLU) else "tanh" for i in range(1, len(ms), 2)}
    if len(names) != 1:
        raise NotImplementedError(f"{who}: one activation class for all networks (got {sorted(names)})")
    return names.pop()


def _actor_bound(actor) -> float:
    """max_action of SAC's / REDQ's Gaussian actor for the engine: 0 for `unbounded=True` (examples/mujoco/mujoco_sac.py:88-94),
    `actor.max_action` for the class default `unbounded=False` (continuous.py:194, 230-231: mu = max_action * tanh(mu))."""
    return 0.0 if getattr(actor, "_unbounded", False) else float(getattr(actor, "max_action", 1.0))


def make_hip_sac(ref=None):
    """Returns HipSAC(SAC): `_preprocess_batch` / `_update_with_batch` (ddpg.py:287-301, sac.py:298-336) on the
    engine.  Supported nets: examples/mujoco/mujoco_sac.py:82-104 (Net[256, 256] ReLU, conditioned sigma,
    unbounded actor; concat critics);
    obs_next is the buffer's stored column or, with save_obs_next=False, obs[next(index)] (buffer_base.py:622-626).  rsample() noise: the engine's
    Philox stream by default, or (`update_noise="torch"`) torch's default generator on the host in the reference's order
    (target-policy call, then actor-loss call) -- the seed-exact mode the fixture replays use.
    `ref`: optional namespace replacing the tianshou imports (see `_ref`)."""
    SAC = _ref(ref, "tianshou.algorithm.modelfree.sac", "SAC")
    AutoAlpha = _ref(ref, "tianshou.algorithm.modelfree.sac", "AutoAlpha")
    SACTrainingStats = _ref(ref, "tianshou.algorithm.modelfree.sac", "SACTrainingStats")
    Batch = _ref(ref, "tianshou.data", "Batch")

    from . import sac as S

    class HipSAC(_HipGlue, SAC):
        _HIP_LR = (("actor_lr", "policy_optim"), ("critic_lr", "critic_optim"), ("critic_lr", "critic2_optim"),
                   ("alpha_lr", "alpha"))
        def __init__(self, *args, device="cuda", data_parallel=False, group=None, allreduce=None, policy_forward="hip",
                     sampling="device", noise_seed=None, write_back="auto", update_noise="device", host_batch=False, **kwargs):
            """`data_parallel=True`: one process per GPU, every rank samples its own minibatch from its own buffer and
            `_update_with_batch` runs the four phases of `DataParallelSAC` around two all-reduces (critic gradients,
            actor gradient + mean log-probability); replicas stay identical, PER weights rank-local.
            `policy_forward="hip"`: `SACPolicy.forward` (sac.py:108-131) as the Collector calls it runs on
            `ts_sac_policy_forward_logits` with the engine's actor (`tianshou_amd.policy`; `sampling` / `noise_seed` as in
            HipPPO).
            Hook-level throughput (round 6; `bench.py --workload sac` `hook_level`): `write_back="auto"` keeps the updates in the
            engine until somebody reads the torch modules (lazy whenever the policy forward is the engine's; "eager" = after every
            update; `hip_sync()` forces it); `update_noise="device"` draws the two rsample() noises of an update from the
            engine's Philox stream instead of torch's host generator ("torch" = the reference's stream, seed-exact);
            `host_batch=False` lets `update()` sample indices only -- the hooks read the rows from the device mirror, the host
            copy `buffer.sample()` makes of them is never looked at (True: the reference's own `Algorithm._update`)."""
            super().__init__(*args, **kwargs)
            if update_noise not in ("device", "torch"):
                raise ValueError("update_noise must be 'device' or 'torch'")
            self._hip_update_noise, self._hip_host_batch, self._hip_noise_calls = update_noise, bool(host_batch), 0
            self._hip_noise_key = int(torch.initial_seed() % (2**31 - 1)) if noise_seed is None else int(noise_seed)
            self._hip_device = torch.device(device)
            sa, sc = self.policy.actor.state_dict(), self.critic.state_dict()
            depth = S.keys_depth(sa.keys(), ("mu", "sigma"))
            if depth is None or any(S.keys_depth(c.state_dict().keys(), ("last",)) != depth for c in (self.critic, self.critic2)):
                raise NotImplementedError("HipSAC: networks must be those of examples/mujoco/mujoco_sac.py (Net trunks of one depth, "
                                          "1 .. 6 hidden layers, single-Linear mu / sigma / Q heads)")
            self._hip_depth, self._hip_akeys, self._hip_ckeys = depth, S.actor_keys(depth), S.critic_keys(depth)
            self._hip_actfn = _trunk_activation((self.policy.actor, self.critic, self.critic2), "HipSAC")
            self._hip_bound = _actor_bound(self.policy.actor)
            # any hidden widths per network (round 6): embedded by zero padding into the engine's Net[h] * depth, h = the largest
            # width of the three networks rounded up to 32 (tianshou_amd.widths)
            from . import widths as WD

            lists = {"actor": [sa[k] for k in self._hip_akeys], "critic1": [sc[k] for k in self._hip_ckeys],
                     "critic2": [self.critic2.state_dict()[k] for k in self._hip_ckeys]}
            try:
                self._hip_sizes = {n: WD.layer_widths(t, 2 if n == "actor" else 1) for n, t in lists.items()}
                hid = WD.engine_hidden(self._hip_sizes.values())
            except NotImplementedError as e:
                raise NotImplementedError(f"HipSAC: hidden layers of widths up to 1024 per network ({e})") from None
            self._hip_hidden = hid
            for o in (self.policy_optim, self.critic_optim, self.critic2_optim):
                _adam_of(o)
            self._hip_engine = None
            self._hip_glue_init()
            self._hip_dp_setup(data_parallel, group, allreduce, shard_buffer=False)
            if policy_forward not in ("hip", "torch"):
                raise ValueError("policy_forward must be 'hip' or 'torch'")
            if policy_forward == "hip":
                from . import policy as HP

                HP.attach(self.policy, "sac", self, device=str(self._hip_device), sampling=sampling, noise_seed=noise_seed,
                          obs_dim=int(sa[self._hip_akeys[0]].shape[1]), act_dim=int(sa[self._hip_akeys[2 * self._hip_depth]].shape[0]),
                          hidden=hid, depth=depth, max_action=self._hip_bound, activation=self._hip_actfn)
            self._hip_set_write_back(write_back, attached=policy_forward == "hip")

        def update(self, buffer, sample_size):
            if self._hip_host_batch or buffer is None:
                return super().update(buffer, sample_size)
            return self._hip_offpolicy_update(buffer, sample_size, Batch)

        def _hip_rsample_noise(self, n, a):
            """eps of one Normal.rsample() call of the reference (sac.py:124-131): torch's host generator ("torch": the
            reference's own stream) or the engine's Philox stream keyed by (noise key, call number)."""
            if self._hip_update_noise == "torch":
                return torch.randn(n, a)
            from .buffer import normal_noise

            self._hip_noise_calls += 1
            return normal_noise((n, a), self._hip_noise_key ^ 0x5AC, self._hip_noise_calls, self._hip_device)

        def _engine(self):
            if self._hip_engine is None:
                sa = self.policy.actor.state_dict()
                obs_dim = sa[self._hip_akeys[0]].shape[1]
                act_dim = sa[self._hip_akeys[2 * self._hip_depth]].shape[0]
                auto = isinstance(self.alpha, AutoAlpha)
                ga, gc = _adam_of(self.policy_optim)[1], _adam_of(self.critic_optim)[1]
                cfg = S.SACConfig(gamma=self.gamma, tau=self.tau, n_step=self.n_step_return_horizon,
                                  alpha=0.0 if auto else float(self.alpha.value), auto_alpha=auto,
                                  target_entropy=float(self.alpha._target_entropy) if auto else 0.0,
                                  log_alpha0=float(self.alpha._log_alpha.item()) if auto else 0.0,
                                  actor_lr=ga["lr"], critic_lr=gc["lr"],
                                  alpha_lr=self.alpha._optim.param_groups[0]["lr"] if auto else 0.0,
                                  betas=tuple(ga["betas"]), adam_eps=ga["eps"])
                dev = self._hip_device
                hid = self._hip_hidden
                flat_c = lambda mod: S.critic_flat_from_torch(  # noqa: E731
                    [mod.state_dict()[k] for k in self._hip_ckeys], obs_dim, act_dim, dev, hidden=hid)
                eng = self._hip_engine = S.SACEngine(
                    obs_dim, act_dim,
                    S.actor_flat_from_torch([sa[k] for k in self._hip_akeys], obs_dim, act_dim, dev, hidden=hid),
                    flat_c(self.critic), flat_c(self.critic2), cfg, hidden=self._hip_hidden, depth=self._hip_depth,
                    max_action=self._hip_bound, activation=self._hip_actfn)
                # resume: lagged critics, Adam moments / steps of a loaded checkpoint
                eng.critic1_old, eng.critic2_old = flat_c(self.critic_old.module), flat_c(self.critic2_old.module)
                for name, mod, optim, keys, conv in self._hip_parts(S):
                    ms, vs, step = adam_state(optim._optim, params_by_keys(mod, keys))
                    setattr(eng, name + "_m", conv(ms, obs_dim, act_dim, dev))
                    setattr(eng, name + "_v", conv(vs, obs_dim, act_dim, dev))
                    eng.adam_step = max(eng.adam_step, step)
                if auto:
                    st = self.alpha._optim.state.get(self.alpha._log_alpha, {})
                    if "exp_avg" in st:
                        eng.log_alpha_m[0], eng.log_alpha_v[0] = float(st["exp_avg"]), float(st["exp_avg_sq"])
            return self._hip_engine

        def _hip_parts(self, S):
            import functools

            fa = functools.partial(S.actor_flat_from_torch, hidden=self._hip_hidden)
            fc = functools.partial(S.critic_flat_from_torch, hidden=self._hip_hidden)
            return (("actor", self.policy.actor, self.policy_optim, self._hip_akeys, fa),
                    ("critic1", self.critic, self.critic_optim, self._hip_ckeys, fc),
                    ("critic2", self.critic2, self.critic2_optim, self._hip_ckeys, fc))

        def _preprocess_batch(self, batch, buffer, indices):
            _require_gpu(self._hip_device, "HipSAC")
            eng = self._engine()
            m = _mirror(self, buffer, self._hip_device)
            idx = torch.as_tensor(np.asarray(indices, np.int64), device=self._hip_device)
            self._hip_idx = idx
            # Inside HipSAC.update()'s own sequence (`_hip_offpolicy_update`: nobody reads the batch between the two hooks) an
            # n_step = 1 update with the engine's noise is ONE library call, made by `_update_with_batch` (ts_sac_learn_rows: the
            # target pass in front of the update, 16 launches instead of 22); `batch.returns` is attached there.  Called on its own
            # (the reference's `_update`, `host_batch=True`, "torch" noise, data parallel) the hook computes the returns here.
            self.__dict__["_hip_deferred"] = (self.__dict__.get("_hip_own_sequence", False) and self._hip_update_noise == "device"
                                              and not self._hip_dp_on and eng.cfg.n_step == 1 and eng._rows_ok(m)
                                              and not os.environ.get("TS_SAC_TWO_CALLS"))
            if self.__dict__["_hip_deferred"]:
                return batch
            noise = self._hip_rsample_noise(len(indices), eng.act_dim)      # Normal.rsample of the target policy call
            batch.returns = eng.preprocess(m, idx, noise).reshape(-1, 1)
            return batch

        def _update_with_batch(self, batch):
            self._hip_refresh_lr()
            from .buffer import gather_rows_multi

            eng, m = self._hip_engine, self._hip_mirror
            weight = getattr(batch, "weight", None)
            if self.__dict__.pop("_hip_deferred", False):
                # target pass + update as one call; the two rsample() draws keep their call numbers (target: c + 1, update: c + 2)
                key, c = self._hip_noise_key ^ 0x5AC, self._hip_noise_calls
                self._hip_noise_calls += 2
                stats, w, ret, _ = eng.learn_rows(m, self._hip_idx, noise_key=(key, c + 1), weight=weight, noise_streams=2)
                batch.returns = ret.reshape(-1, 1)
            else:
                noise = self._hip_rsample_noise(int(self._hip_idx.numel()), eng.act_dim)
                runner = eng
                if self._hip_dp_on:
                    from .distributed import DataParallelSAC

                    runner = self._hip_dp(DataParallelSAC, eng)
                if runner is eng and hasattr(eng, "update_with_rows"):           # the input packing reads the mirror's rows
                    stats, w = eng.update_with_rows(m, self._hip_idx, batch.returns.reshape(-1), noise, weight)
                else:
                    stats, w = runner.update_with_batch(*gather_rows_multi([m.obs, m.act], self._hip_idx),
                                                        batch.returns.reshape(-1), noise, weight)
            batch.weight = w                                                      # prio-buffer, sac.py:306
            s = stats.cpu().numpy()                                               # one D2H per update()
            self._hip_after_update()                                              # write-back now ("eager") or when read ("lazy")
            auto = eng.cfg.auto_alpha
            return SACTrainingStats(actor_loss=float(s[0]), critic1_loss=float(s[1]), critic2_loss=float(s[2]),
                                    alpha=float(s[3]), alpha_loss=float(s[4]) if auto else None)

        def _hip_write_back(self) -> None:
            eng = self.__dict__.get("_hip_engine_obj")
            if eng is None:
                return
            with torch.no_grad():
                sz = self._hip_sizes
                for mod, flat, conv, name in ((self.policy.actor, eng.actor, S.actor_flat_to_torch, "actor"),
                                              (self.critic, eng.critic1, S.critic_flat_to_torch, "critic1"),
                                              (self.critic2, eng.critic2, S.critic_flat_to_torch, "critic2"),
                                              (self.critic_old.module, eng.critic1_old, S.critic_flat_to_torch, "critic1"),
                                              (self.critic2_old.module, eng.critic2_old, S.critic_flat_to_torch, "critic2")):
                    for p, t in zip(mod.parameters(), conv(flat, eng.obs_dim, eng.act_dim, eng.hidden, sizes=sz[name])):
                        self._hip_put(p, t)
                if eng.cfg.auto_alpha:
                    self._hip_put(self.alpha._log_alpha, eng.log_alpha[0])
            back = {"actor": S.actor_flat_to_torch, "critic1": S.critic_flat_to_torch, "critic2": S.critic_flat_to_torch}
            for name, mod, optim, keys, _ in self._hip_parts(S):
                store_adam_state(optim._optim, params_by_keys(mod, keys),
                                 back[name](getattr(eng, name + "_m"), eng.obs_dim, eng.act_dim, eng.hidden, sizes=self._hip_sizes[name]),
                                 back[name](getattr(eng, name + "_v"), eng.obs_dim, eng.act_dim, eng.hidden, sizes=self._hip_sizes[name]),
                                 eng.adam_step)
            if eng.cfg.auto_alpha:
                store_adam_state(self.alpha._optim, [self.alpha._log_alpha], [eng.log_alpha_m[0]],
                                 [eng.log_alpha_v[0]], eng.adam_step)

    return HipSAC


# ---------------------------------------------------------------------------------------------------
# REDQ (redq.py) on the nets of test/continuous/test_redq.py
# ---------------------------------------------------------------------------------------------------
def make_hip_redq(ref=None):
    """Returns HipREDQ(REDQ): `_preprocess_batch` / `_update_with_batch` (ddpg.py:287-301, redq.py:248-304) on the engine.
    Supported nets: SAC's actor (Net[256, 256] ReLU, conditioned sigma, unbounded) and one critic module made of
    EnsembleLinear layers with hidden [256, 256] (test/continuous/test_redq.py:86-107);
    obs_next is the buffer's stored column or, with save_obs_next=False, obs[next(index)] (buffer_base.py:622-626).
    The rsample() noise comes from torch's default generator and the critic subset from NumPy's global generator, in
    the reference's order (target call: noise, then np.random.choice; actor step: noise).
    `ref`: optional namespace replacing the tianshou imports (see `_ref`)."""
    REDQ = _ref(ref, "tianshou.algorithm.modelfree.redq", "REDQ")
    REDQTrainingStats = _ref(ref, "tianshou.algorithm.modelfree.redq", "REDQTrainingStats")
    AutoAlpha = _ref(ref, "tianshou.algorithm.modelfree.sac", "AutoAlpha")

    from . import redq as RQ
    from . import sac as S

    class HipREDQ(_HipGlue, REDQ):
        _HIP_LR = (("actor_lr", "policy_optim"), ("critic_lr", "critic_optim"), ("alpha_lr", "alpha"))
        def __init__(self, *args, device="cuda", **kwargs):
            super().__init__(*args, **kwargs)
            self._hip_device = torch.device(device)
            sa, sc = self.policy.actor.state_dict(), self.critic.state_dict()
            depth = S.keys_depth(sa.keys(), ("mu", "sigma"))
            if depth is None or RQ.keys_depth(sc.keys()) != depth:
                raise NotImplementedError("HipREDQ: networks must be those of test/continuous/test_redq.py (SAC's actor; a critic of "
                                          "EnsembleLinear layers; trunks of one depth, 1 .. 6 hidden layers)")
            self._hip_depth, self._hip_akeys, self._hip_ckeys = depth, S.actor_keys(depth), RQ.critic_keys(depth)
            self._hip_actfn = _trunk_activation((self.policy.actor, self.critic), "HipREDQ")
            self._hip_bound = _actor_bound(self.policy.actor)
            from . import widths as WD

            try:        # any hidden widths (round 6): embedded by zero padding (tianshou_amd.widths)
                cw = [sc[k] for k in self._hip_ckeys[:2 * depth:2]]          # EnsembleLinear weights [E, in, out]
                if any(w.dim() != 3 or w.shape[0] != self.ensemble_size or (i > 0 and w.shape[1] != cw[i - 1].shape[2]) for i, w in enumerate(cw)):
                    raise NotImplementedError("EnsembleLinear weights [ensemble_size, in, out] are required")
                self._hip_sizes = {"actor": WD.layer_widths([sa[k] for k in self._hip_akeys], 2),
                                   "critic": tuple(int(w.shape[2]) for w in cw)}
                hid = WD.engine_hidden(self._hip_sizes.values())
            except NotImplementedError as e:
                raise NotImplementedError(f"HipREDQ: hidden layers of widths up to 1024 per network ({e})") from None
            self._hip_hidden = hid
            for o in (self.policy_optim, self.critic_optim):
                _adam_of(o)
            self._hip_engine = None
            self._hip_glue_init()

        def _critic_tensors(self, mod):
            return [mod.state_dict()[k] for k in self._hip_ckeys]

        def _engine(self):
            if self._hip_engine is None:
                sa = self.policy.actor.state_dict()
                obs_dim, act_dim = sa[self._hip_akeys[0]].shape[1], sa[self._hip_akeys[2 * self._hip_depth]].shape[0]
                auto = isinstance(self.alpha, AutoAlpha)
                ga, gc = _adam_of(self.policy_optim)[1], _adam_of(self.critic_optim)[1]
                cfg = RQ.REDQConfig(gamma=self.gamma, tau=self.tau, n_step=self.n_step_return_horizon,
                                    alpha=0.0 if auto else float(self.alpha.value), auto_alpha=auto,
                                    target_entropy=float(self.alpha._target_entropy) if auto else 0.0,
                                    log_alpha0=float(self.alpha._log_alpha.item()) if auto else 0.0,
                                    actor_lr=ga["lr"], critic_lr=gc["lr"],
                                    alpha_lr=self.alpha._optim.param_groups[0]["lr"] if auto else 0.0,
                                    betas=tuple(ga["betas"]), adam_eps=ga["eps"], ensemble_size=self.ensemble_size,
                                    subset_size=self.subset_size, actor_delay=self.actor_delay, target_mode=self.target_mode)
                dev = self._hip_device
                hid = self._hip_hidden
                eng = self._hip_engine = RQ.REDQEngine(
                    obs_dim, act_dim, S.actor_flat_from_torch([sa[k] for k in self._hip_akeys], obs_dim, act_dim, dev, hidden=hid),
                    RQ.ensemble_flat_from_torch(self._critic_tensors(self.critic), obs_dim, act_dim, dev, hidden=hid), cfg,
                    hidden=self._hip_hidden, depth=self._hip_depth, max_action=self._hip_bound, activation=self._hip_actfn)
                # resume: lagged ensemble, counters, Adam moments / steps of a loaded checkpoint
                eng.critics_old = RQ.ensemble_flat_from_torch(self._critic_tensors(self.critic_old.module), obs_dim, act_dim, dev, hidden=hid)
                eng.critic_gradient_step = int(self.critic_gradient_step)
                ms, vs, step = adam_state(self.policy_optim._optim, params_by_keys(self.policy.actor, self._hip_akeys))
                eng.actor_m, eng.actor_v = (S.actor_flat_from_torch(x, obs_dim, act_dim, dev, hidden=hid) for x in (ms, vs))
                eng.actor_steps = step
                ms, vs, _ = adam_state(self.critic_optim._optim, params_by_keys(self.critic, self._hip_ckeys))
                eng.critics_m, eng.critics_v = (RQ.ensemble_flat_from_torch(x, obs_dim, act_dim, dev, hidden=hid) for x in (ms, vs))
                eng._stats[0] = float(self._last_actor_loss)
                if auto:
                    st = self.alpha._optim.state.get(self.alpha._log_alpha, {})
                    if "exp_avg" in st:
                        eng.log_alpha_m[0], eng.log_alpha_v[0] = float(st["exp_avg"]), float(st["exp_avg_sq"])
            return self._hip_engine

        def _preprocess_batch(self, batch, buffer, indices):
            _require_gpu(self._hip_device, "HipREDQ")
            eng = self._engine()
            m = _mirror(self, buffer, self._hip_device)
            idx = torch.as_tensor(np.asarray(indices, np.int64), device=self._hip_device)
            noise = torch.randn(len(indices), eng.act_dim)                  # Normal.rsample of the target policy call
            subset = np.random.choice(self.ensemble_size, self.subset_size, replace=False)     # redq.py:252
            batch.returns = eng.preprocess(m, idx, noise, subset).reshape(-1, 1)
            self._hip_idx = idx
            return batch

        def _update_with_batch(self, batch):
            self._hip_refresh_lr()
            from .buffer import gather_rows_multi

            eng, m = self._hip_engine, self._hip_mirror
            weight = getattr(batch, "weight", None)
            did_actor = eng.will_update_actor()
            noise = torch.randn(len(batch), eng.act_dim) if did_actor else None
            stats, w = eng.update_with_batch(*gather_rows_multi([m.obs, m.act], self._hip_idx),
                                             batch.returns.reshape(-1), noise, weight)
            batch.weight = w                                                      # prio-buffer, redq.py:272
            s = stats.cpu().numpy()                                               # one D2H per update()
            self.critic_gradient_step = eng.critic_gradient_step
            self._last_actor_loss = float(s[0])
            dims = (eng.obs_dim, eng.act_dim, eng.hidden)
            E = eng.cfg.ensemble_size
            with torch.no_grad():
                sa_, sc_ = self._hip_sizes["actor"], self._hip_sizes["critic"]
                for p, t in zip(params_by_keys(self.policy.actor, self._hip_akeys), S.actor_flat_to_torch(eng.actor, *dims, sizes=sa_)):
                    p.copy_(t)
                for mod, flat in ((self.critic, eng.critics), (self.critic_old.module, eng.critics_old)):
                    for p, t in zip(params_by_keys(mod, self._hip_ckeys), RQ.ensemble_flat_to_torch(flat, E, *dims, sizes=sc_)):
                        p.copy_(t)
                if eng.cfg.auto_alpha:
                    self.alpha._log_alpha.copy_(eng.log_alpha[0])
            store_adam_state(self.critic_optim._optim, params_by_keys(self.critic, self._hip_ckeys),
                             RQ.ensemble_flat_to_torch(eng.critics_m, E, *dims, sizes=self._hip_sizes["critic"]),
                             RQ.ensemble_flat_to_torch(eng.critics_v, E, *dims, sizes=self._hip_sizes["critic"]),
                             eng.critic_gradient_step)
            if eng.actor_steps:
                store_adam_state(self.policy_optim._optim, params_by_keys(self.policy.actor, self._hip_akeys),
                                 S.actor_flat_to_torch(eng.actor_m, *dims, sizes=self._hip_sizes["actor"]),
                                 S.actor_flat_to_torch(eng.actor_v, *dims, sizes=self._hip_sizes["actor"]),
                                 eng.actor_steps)
                if eng.cfg.auto_alpha:
                    store_adam_state(self.alpha._optim, [self.alpha._log_alpha], [eng.log_alpha_m[0]], [eng.log_alpha_v[0]],
                                     eng.actor_steps)
            return REDQTrainingStats(actor_loss=float(s[0]), critic_loss=float(s[1]), alpha=float(s[2]),
                                     alpha_loss=None if np.isnan(s[3]) else float(s[3]))

    return HipREDQ


# ---------------------------------------------------------------------------------------------------
# DiscreteSAC (discrete_sac.py) on the MLP nets of test/discrete/test_discrete_sac.py
# ---------------------------------------------------------------------------------------------------
def make_hip_discrete_sac(ref=None):
    """Returns HipDiscreteSAC(DiscreteSAC): `_preprocess_batch` / `_update_with_batch` (ddpg.py:287-301,
    discrete_sac.py:147-196) on the engine.  Supported nets: Net(obs, [h, h]) ReLU under DiscreteActor(softmax_output=
    False) and DiscreteCritic(last_size=n_act) (test/discrete/test_discrete_sac.py:88-97), h a multiple of 32; the
    obs_next is the buffer's stored column or obs[next(index)] (buffer_base.py:622-626).  `match_rng_stream`: the reference's two policy calls per update draw
    `Categorical.sample()` values that are never used; with the flag set (default) the same draws are made from the
    engine's logits so that torch's global generator advances exactly as in the reference.
    `ref`: optional namespace replacing the tianshou imports (see `_ref`)."""
    from torch.distributions import Categorical

    DiscreteSAC = _ref(ref, "tianshou.algorithm.modelfree.discrete_sac", "DiscreteSAC")
    DiscreteSACTrainingStats = _ref(ref, "tianshou.algorithm.modelfree.discrete_sac", "DiscreteSACTrainingStats")
    AutoAlpha = _ref(ref, "tianshou.algorithm.modelfree.sac", "AutoAlpha")

    from . import dsac as DS
    from .sac import SACConfig

    class HipDiscreteSAC(_HipGlue, DiscreteSAC):
        _HIP_LR = (("actor_lr", "policy_optim"), ("critic_lr", "critic_optim"), ("critic_lr", "critic2_optim"),
                   ("alpha_lr", "alpha"))
        def __init__(self, *args, device="cuda", match_rng_stream: bool = True, **kwargs):
            super().__init__(*args, **kwargs)
            self._hip_device = torch.device(device)
            self._hip_match_rng = match_rng_stream
            mods = (self.policy.actor, self.critic, self.critic2)
            depth = DS.keys_depth(mods[0].state_dict().keys(), ("last",))
            if depth is None or any(DS.keys_depth(m.state_dict().keys(), ("last",)) != depth for m in mods):
                raise NotImplementedError("HipDiscreteSAC: networks must be Net(obs, [h, ...]) of one depth (1 .. 6 hidden layers) + a "
                                          "single Linear head")
            self._hip_depth, self._hip_keys = depth, DS.net_keys(depth)
            self._hip_actfn = _trunk_activation(mods, "HipDiscreteSAC")
            sa = self.policy.actor.state_dict()
            from . import widths as WD

            lists = {n: [m.state_dict()[k] for k in self._hip_keys] for n, m in zip(("actor", "critic1", "critic2"), mods)}
            try:        # any hidden widths per network (round 6): embedded by zero padding (tianshou_amd.widths)
                self._hip_sizes = {n: WD.layer_widths(t, 1) for n, t in lists.items()}
                self._hip_hidden = WD.engine_hidden(self._hip_sizes.values())
            except NotImplementedError as e:
                raise NotImplementedError(f"HipDiscreteSAC: hidden layers of widths up to 1024 per network ({e})") from None
            if not 2 <= sa[self._hip_keys[2 * self._hip_depth]].shape[0] <= 64:
                raise NotImplementedError("HipDiscreteSAC: 2..64 actions")
            if getattr(self.policy.actor, "softmax_output", False):
                raise NotImplementedError("HipDiscreteSAC: the actor must output logits (softmax_output=False)")
            for o in (self.policy_optim, self.critic_optim, self.critic2_optim):
                _adam_of(o)
            self._hip_engine = None
            self._hip_glue_init()

        def _hip_parts(self):
            return (("actor", self.policy.actor, self.policy_optim), ("critic1", self.critic, self.critic_optim),
                    ("critic2", self.critic2, self.critic2_optim))

        def _engine(self):
            if self._hip_engine is None:
                sa = self.policy.actor.state_dict()
                obs_dim = sa[self._hip_keys[0]].shape[1]
                n_act = sa[self._hip_keys[2 * self._hip_depth]].shape[0]
                dims = (obs_dim, n_act, self._hip_hidden)
                auto = isinstance(self.alpha, AutoAlpha)
                ga, gc = _adam_of(self.policy_optim)[1], _adam_of(self.critic_optim)[1]
                cfg = SACConfig(gamma=self.gamma, tau=self.tau, n_step=self.n_step_return_horizon,
                                alpha=0.0 if auto else float(self.alpha.value), auto_alpha=auto,
                                target_entropy=float(self.alpha._target_entropy) if auto else 0.0,
                                log_alpha0=float(self.alpha._log_alpha.item()) if auto else 0.0,
                                actor_lr=ga["lr"], critic_lr=gc["lr"],
                                alpha_lr=self.alpha._optim.param_groups[0]["lr"] if auto else 0.0,
                                betas=tuple(ga["betas"]), adam_eps=ga["eps"])
                dev = self._hip_device
                flat = lambda mod: DS.net_flat_from_torch(  # noqa: E731
                    [mod.state_dict()[k] for k in self._hip_keys], *dims, dev)
                eng = self._hip_engine = DS.DiscreteSACEngine(*dims, flat(self.policy.actor), flat(self.critic),
                                                              flat(self.critic2), cfg, depth=self._hip_depth, activation=self._hip_actfn)
                eng.critic1_old, eng.critic2_old = flat(self.critic_old.module), flat(self.critic2_old.module)
                for name, mod, optim in self._hip_parts():             # resume from a loaded checkpoint
                    ms, vs, step = adam_state(optim._optim, params_by_keys(mod, self._hip_keys))
                    setattr(eng, name + "_m", DS.net_flat_from_torch(ms, *dims, dev))
                    setattr(eng, name + "_v", DS.net_flat_from_torch(vs, *dims, dev))
                    eng.adam_step = max(eng.adam_step, step)
                if auto:
                    st = self.alpha._optim.state.get(self.alpha._log_alpha, {})
                    if "exp_avg" in st:
                        eng.log_alpha_m[0], eng.log_alpha_v[0] = float(st["exp_avg"]), float(st["exp_avg_sq"])
            return self._hip_engine

        def _hip_draw(self, obs):
            if self._hip_match_rng:                  # discrete_sac.py:60-66: dist.sample() inside the training step
                Categorical(logits=self._hip_engine.policy_forward(obs).cpu()).sample()

        def _preprocess_batch(self, batch, buffer, indices):
            from .buffer import gather_rows

            _require_gpu(self._hip_device, "HipDiscreteSAC")
            eng = self._engine()
            m = _mirror(self, buffer, self._hip_device)
            idx = torch.as_tensor(np.asarray(indices, np.int64), device=self._hip_device)
            if self._hip_match_rng:
                from .returns import nstep_indices

                self._hip_draw(gather_rows(m.obs_next, nstep_indices(m, idx, eng.cfg.n_step)))
            batch.returns = eng.preprocess(m, idx).reshape(-1, 1)
            self._hip_idx = idx
            return batch

        def _update_with_batch(self, batch):
            self._hip_refresh_lr()
            from .buffer import gather_rows

            eng, m = self._hip_engine, self._hip_mirror
            weight = getattr(batch, "weight", None)
            obs = gather_rows(m.obs, self._hip_idx)
            self._hip_draw(obs)
            stats, w = eng.update_with_batch(obs, gather_rows(m.act, self._hip_idx), batch.returns.reshape(-1), weight)
            batch.weight = w                                                      # prio-buffer, discrete_sac.py:174
            s = stats.cpu().numpy()                                               # one D2H per update()
            dims = (eng.obs_dim, eng.n_act, eng.hidden)
            with torch.no_grad():
                sz = self._hip_sizes
                for mod, flat, nm in ((self.policy.actor, eng.actor, "actor"), (self.critic, eng.critic1, "critic1"),
                                      (self.critic2, eng.critic2, "critic2"), (self.critic_old.module, eng.critic1_old, "critic1"),
                                      (self.critic2_old.module, eng.critic2_old, "critic2")):
                    for p, t in zip(params_by_keys(mod, self._hip_keys), DS.net_flat_to_torch(flat, *dims, sizes=sz[nm])):
                        p.copy_(t)
                if eng.cfg.auto_alpha:
                    self.alpha._log_alpha.copy_(eng.log_alpha[0])
            for name, mod, optim in self._hip_parts():
                store_adam_state(optim._optim, params_by_keys(mod, self._hip_keys),
                                 DS.net_flat_to_torch(getattr(eng, name + "_m"), *dims, sizes=self._hip_sizes[name]),
                                 DS.net_flat_to_torch(getattr(eng, name + "_v"), *dims, sizes=self._hip_sizes[name]), eng.adam_step)
            if eng.cfg.auto_alpha:
                store_adam_state(self.alpha._optim, [self.alpha._log_alpha], [eng.log_alpha_m[0]],
                                 [eng.log_alpha_v[0]], eng.adam_step)
            auto = eng.cfg.auto_alpha
            return DiscreteSACTrainingStats(actor_loss=float(s[0]), critic1_loss=float(s[1]), critic2_loss=float(s[2]),
                                            alpha=float(s[3]), alpha_loss=float(s[4]) if auto else None)

    return HipDiscreteSAC


# ---------------------------------------------------------------------------------------------------
# PPO on the Atari actor-critic (examples/atari/atari_ppo.py:106-135)
# ---------------------------------------------------------------------------------------------------
def make_hip_ppo_cnn(algo: str = "ppo", ref=None):
    """Returns HipPPOCnn(PPO) for DQNet(features_only=True, output_dim_added_layer=512) shared by
    DiscreteActor(softmax_output=False) and DiscreteCritic; Categorical policy; Adam.  algo="a2c": HipA2CCnn(A2C).
    `ref`: optional namespace replacing the tianshou imports (see `_ref`)."""
    A2CTrainingStats = _ref(ref, "tianshou.algorithm.modelfree.a2c", "A2CTrainingStats")
    SequenceSummaryStats = _ref(ref, "tianshou.data", "SequenceSummaryStats")
    PPO = _on_policy_base(algo, ref)

    from . import ppo_cnn as PC

    class HipPPOCnn(_HipGlue, PPO):
        def __init__(self, *args, device="cuda", **kwargs):
            super().__init__(*args, **kwargs)
            self._hip_device = torch.device(device)
            actor, critic = self.policy.actor, self.critic
            sa, sc = actor.state_dict(), critic.state_dict()
            if list(sa.keys()) != PC.TRUNK_KEYS + PC.HEAD_KEYS or list(sc.keys()) != PC.TRUNK_KEYS + PC.HEAD_KEYS \
                    or actor.preprocess is not critic.preprocess or getattr(actor, "softmax_output", True):
                raise NotImplementedError("HipPPOCnn: actor / critic must share one DQNet(features_only=True, "
                                          "output_dim_added_layer=512) trunk with single-Linear heads (logits)")
            optimizer_fields(self.optim._optim)                # Adam / RMSprop incl. weight decay (ts::optim_step)
            self._hip_engine = None
            self._hip_glue_init()

        def _hip_params(self):
            return params_by_keys(self.policy.actor, PC.TRUNK_KEYS + PC.HEAD_KEYS) + params_by_keys(self.critic, PC.HEAD_KEYS)

        def _layout(self, buffer):
            obs = np.asarray(buffer.obs)
            stack = int(getattr(buffer, "stack_num", 1))
            if stack > 1:
                return stack, obs.shape[1], obs.shape[2], stack
            if obs.ndim != 4:
                raise NotImplementedError("HipPPOCnn: observations must be [c, h, w]")
            return obs.shape[1], obs.shape[2], obs.shape[3], 1

        def _engine(self, c, h, w):
            if self._hip_engine is None:
                params = self._hip_params()
                n_act = params[8].shape[0]
                dev = self._hip_device
                eng = self._hip_engine = PC.CnnPPOEngine(c, h, w, n_act, PC.flat_from_torch(params, c, h, w, n_act, dev),
                                                         ppo_config_from(self))
                eng.ret_rms = [float(self.ret_rms.mean), float(self.ret_rms.var), float(self.ret_rms.count)]
                ms, vs, step = adam_state(self.optim._optim, params)
                eng.adam_m, eng.adam_v = PC.flat_from_torch(ms, c, h, w, n_act, dev), PC.flat_from_torch(vs, c, h, w, n_act, dev)
                eng.adam_step = step
            return self._hip_engine

        def _preprocess_batch(self, batch, buffer, indices):
            _require_gpu(self._hip_device, "HipPPOCnn")
            c, h, w, stack = self._layout(buffer)
            eng = self._engine(c, h, w)
            m = _mirror(self, buffer, self._hip_device)
            pre = eng.preprocess(m, m.obs, m.act, stack, obs_next_frames=m.obs_next)
            self._hip_pre, self._hip_stack = pre, stack
            batch.v_s, batch.returns, batch.adv, batch.logp_old = pre["v_s"], pre["returns"], pre["adv"], pre["logp_old"]
            return batch

        def _update_with_batch(self, batch, batch_size, repeat):
            self._hip_refresh_lr()
            eng, m = self._hip_engine, self._hip_mirror
            perms = [np.random.permutation(len(batch)) for _ in range(repeat)]     # Batch.split, batch.py:1209
            losses, steps = eng.update(m, m.obs, self._hip_pre, self._hip_stack, batch_size, repeat, perms)
            arr = losses.cpu().numpy().astype(np.float64)
            params = self._hip_params()
            dims = (eng.c, eng.h, eng.w, eng.n_act)
            with torch.no_grad():
                for p, t in zip(params, PC.flat_to_torch(eng.params, *dims)):
                    p.copy_(t)
            store_adam_state(self.optim._optim, params, PC.flat_to_torch(eng.adam_m, *dims),
                             PC.flat_to_torch(eng.adam_v, *dims), eng.adam_step)
            self.ret_rms.mean, self.ret_rms.var, self.ret_rms.count = eng.ret_rms
            return A2CTrainingStats(
                loss=SequenceSummaryStats.from_sequence(arr[:, 0]), actor_loss=SequenceSummaryStats.from_sequence(arr[:, 1]),
                vf_loss=SequenceSummaryStats.from_sequence(arr[:, 2]), ent_loss=SequenceSummaryStats.from_sequence(arr[:, 3]),
                gradient_steps=steps)

    if algo == "a2c":
        HipPPOCnn.__name__ = HipPPOCnn.__qualname__ = "HipA2CCnn"
    return HipPPOCnn


# ---------------------------------------------------------------------------------------------------
# PPO on the CartPole-shape networks (BASELINE.json configs[0], test/discrete/test_ppo_discrete.py:88-127)
# ---------------------------------------------------------------------------------------------------
def make_hip_ppo_discrete(algo: str = "ppo", ref=None):
    """Returns HipPPODiscrete(PPO) for Net(obs, [h, h]) shared by DiscreteActor and DiscreteCritic, Categorical policy
    (`softmax_output=True` with `dist_fn=torch.distributions.Categorical`, or `softmax_output=False` with the default
    logits dist_fn), Adam; h a multiple of 32, at most 31 actions;
    obs_next is the buffer's stored column or, with save_obs_next=False, obs[next(index)] (buffer_base.py:622-626).
    algo="a2c": HipA2CDiscrete(A2C).  `ref`: optional namespace replacing the tianshou imports (see `_ref`)."""
    from torch.distributions import Categorical

    A2CTrainingStats = _ref(ref, "tianshou.algorithm.modelfree.a2c", "A2CTrainingStats")
    dist_fn_categorical_from_logits = _ref(ref, "tianshou.algorithm.modelfree.reinforce", "dist_fn_categorical_from_logits")
    SequenceSummaryStats = _ref(ref, "tianshou.data", "SequenceSummaryStats")
    PPO = _on_policy_base(algo, ref)

    from . import ppo_discrete as PD

    class HipPPODiscrete(_HipGlue, PPO):
        def __init__(self, *args, device="cuda", **kwargs):
            super().__init__(*args, **kwargs)
            self._hip_device = torch.device(device)
            actor, critic = self.policy.actor, self.critic
            sa, sc = actor.state_dict(), critic.state_dict()
            keys = PD.TRUNK_KEYS + PD.HEAD_KEYS
            if list(sa.keys()) != keys or list(sc.keys()) != keys or actor.preprocess is not critic.preprocess:
                raise NotImplementedError("HipPPODiscrete: actor / critic must share one Net(obs, [h, h]) trunk with "
                                          "single-Linear heads")
            hidden, n_act = sa[keys[0]].shape[0], sa[keys[4]].shape[0]
            if hidden % 32 or sa[keys[2]].shape != (hidden, hidden) or n_act > 31 or sc[keys[4]].shape[0] != 1:
                raise NotImplementedError("HipPPODiscrete: hidden sizes [h, h] with h a multiple of 32, <= 31 actions")
            # the actor's outputs must reach Categorical as what they are: probabilities or logits
            softmax = bool(getattr(actor, "softmax_output", True))
            dist_fn = self.policy.dist_fn
            if not ((softmax and dist_fn is Categorical) or (not softmax and dist_fn is dist_fn_categorical_from_logits)):
                raise NotImplementedError("HipPPODiscrete: softmax_output=True needs dist_fn=Categorical, "
                                          "softmax_output=False the logits dist_fn")
            optimizer_fields(self.optim._optim)                # Adam / RMSprop incl. weight decay (ts::optim_step)
            self._hip_engine = None
            self._hip_glue_init()

        def _hip_params(self):
            return params_by_keys(self.policy.actor, PD.TRUNK_KEYS + PD.HEAD_KEYS) + params_by_keys(self.critic, PD.HEAD_KEYS)

        def _engine(self):
            if self._hip_engine is None:
                params = self._hip_params()
                hidden, obs_dim = params[0].shape
                dims, dev = (obs_dim, hidden, params[4].shape[0]), self._hip_device
                eng = self._hip_engine = PD.DiscretePPOEngine(*dims, PD.flat_from_torch(params, *dims, dev),
                                                              ppo_config_from(self))
                eng.ret_rms = [float(self.ret_rms.mean), float(self.ret_rms.var), float(self.ret_rms.count)]
                ms, vs, step = adam_state(self.optim._optim, params)
                eng.adam_m, eng.adam_v = PD.flat_from_torch(ms, *dims, dev), PD.flat_from_torch(vs, *dims, dev)
                eng.adam_step = step
            return self._hip_engine

        def _preprocess_batch(self, batch, buffer, indices):
            _require_gpu(self._hip_device, "HipPPODiscrete")
            eng = self._engine()
            m = _mirror(self, buffer, self._hip_device)
            pre = self._hip_pre = eng.preprocess(m)
            batch.v_s, batch.returns, batch.adv, batch.logp_old = pre["v_s"], pre["returns"], pre["adv"], pre["logp_old"]
            return batch

        def _update_with_batch(self, batch, batch_size, repeat):
            self._hip_refresh_lr()
            eng, m = self._hip_engine, self._hip_mirror
            perms = [np.random.permutation(len(batch)) for _ in range(repeat)]     # Batch.split, batch.py:1209
            losses, steps = eng.update(m, self._hip_pre, batch_size, repeat, perms)
            arr = losses.cpu().numpy().astype(np.float64)
            params = self._hip_params()
            dims = (eng.obs_dim, eng.hidden, eng.n_act)
            with torch.no_grad():
                for p, t in zip(params, PD.flat_to_torch(eng.params, *dims)):
                    p.copy_(t)
            store_adam_state(self.optim._optim, params, PD.flat_to_torch(eng.adam_m, *dims),
                             PD.flat_to_torch(eng.adam_v, *dims), eng.adam_step)
            self.ret_rms.mean, self.ret_rms.var, self.ret_rms.count = eng.ret_rms
            return A2CTrainingStats(
                loss=SequenceSummaryStats.from_sequence(arr[:, 0]), actor_loss=SequenceSummaryStats.from_sequence(arr[:, 1]),
                vf_loss=SequenceSummaryStats.from_sequence(arr[:, 2]), ent_loss=SequenceSummaryStats.from_sequence(arr[:, 3]),
                gradient_steps=steps)

    if algo == "a2c":
        HipPPODiscrete.__name__ = HipPPODiscrete.__qualname__ = "HipA2CDiscrete"
    return HipPPODiscrete


# ---------------------------------------------------------------------------------------------------
# TD3 / DDPG (td3.py:104-226, ddpg.py:343-411) on the mujoco_td3.py / mujoco_ddpg.py networks
# ---------------------------------------------------------------------------------------------------
def _make_hip_det(twin: bool, ref=None):
    DDPG = _ref(ref, "tianshou.algorithm.modelfree.ddpg", "DDPG")
    DDPGTrainingStats = _ref(ref, "tianshou.algorithm.modelfree.ddpg", "DDPGTrainingStats")
    TD3 = _ref(ref, "tianshou.algorithm.modelfree.td3", "TD3")
    TD3TrainingStats = _ref(ref, "tianshou.algorithm.modelfree.td3", "TD3TrainingStats")

    from . import td3 as T

    base = TD3 if twin else DDPG

    class HipDet(_HipGlue, base):
        _HIP_LR = (("actor_lr", "policy_optim"), ("critic_lr", "critic_optim")) + \
            ((("critic_lr", "critic2_optim"),) if twin else ())
        def __init__(self, *args, device="cuda", **kwargs):
            super().__init__(*args, **kwargs)
            self._hip_device = torch.device(device)
            sa = self.policy.actor.state_dict()
            critics = [self.critic] + ([self.critic2] if twin else [])
            depth = T.keys_depth(sa.keys(), ("last",))
            if depth is None or any(T.keys_depth(c.state_dict().keys(), ("last",)) != depth for c in critics):
                raise NotImplementedError("HipTD3 / HipDDPG: networks must be those of examples/mujoco/mujoco_td3.py (Net trunks of "
                                          "one depth, 1 .. 6 hidden layers, single-Linear action / Q heads)")
            self._hip_depth, self._hip_akeys, self._hip_ckeys = depth, T.actor_keys(depth), T.critic_keys(depth)
            self._hip_actfn = _trunk_activation([self.policy.actor] + critics, "HipTD3 / HipDDPG")
            # any hidden widths per network, e.g. the [400, 300] of the TD3 / DDPG papers (round 6, tianshou_amd.widths)
            from . import widths as WD

            lists = {"actor": [sa[k] for k in self._hip_akeys]}
            for i, c in enumerate(critics):
                lists[f"critic{i + 1}"] = [c.state_dict()[k] for k in self._hip_ckeys]
            try:
                self._hip_sizes = {n: WD.layer_widths(t, 1) for n, t in lists.items()}
                hid = WD.engine_hidden(self._hip_sizes.values())
            except NotImplementedError as e:
                raise NotImplementedError(f"HipTD3 / HipDDPG: hidden layers of widths up to 1024 per network ({e})") from None
            self._hip_hidden = hid
            for o in [self.policy_optim, self.critic_optim] + ([self.critic2_optim] if twin else []):
                _adam_of(o)
            self._hip_engine = None
            self._hip_glue_init()

        def _hip_parts(self):
            import functools

            P, hid, sz = functools.partial, self._hip_hidden, self._hip_sizes
            parts = [("actor", self.policy.actor, self.policy_optim, self._hip_akeys, P(T.actor_flat_from_torch, hidden=hid),
                      P(T.actor_flat_to_torch, sizes=sz["actor"]), self.actor_old.module),
                     ("critic1", self.critic, self.critic_optim, self._hip_ckeys, P(T.critic_flat_from_torch, hidden=hid),
                      P(T.critic_flat_to_torch, sizes=sz["critic1"]), self.critic_old.module)]
            if twin:
                parts.append(("critic2", self.critic2, self.critic2_optim, self._hip_ckeys, P(T.critic_flat_from_torch, hidden=hid),
                              P(T.critic_flat_to_torch, sizes=sz["critic2"]), self.critic2_old.module))
            return parts

        def _engine(self):
            if self._hip_engine is None:
                sa = self.policy.actor.state_dict()
                obs_dim, act_dim = sa[self._hip_akeys[0]].shape[1], sa[self._hip_akeys[2 * self._hip_depth]].shape[0]
                ga, gc = _adam_of(self.policy_optim)[1], _adam_of(self.critic_optim)[1]
                cfg = T.TD3Config(gamma=self.gamma, tau=self.tau, n_step=self.n_step_return_horizon, twin=twin,
                                  policy_noise=getattr(self, "policy_noise", 0.0), noise_clip=getattr(self, "noise_clip", 0.0),
                                  update_actor_freq=getattr(self, "update_actor_freq", 1),
                                  max_action=float(self.policy.actor.max_action), actor_lr=ga["lr"], critic_lr=gc["lr"],
                                  betas=tuple(ga["betas"]), adam_eps=ga["eps"])
                dev = self._hip_device
                flats = {n: conv([mod.state_dict()[k] for k in keys], obs_dim, act_dim, dev)
                         for n, mod, _, keys, conv, _, _ in self._hip_parts()}
                eng = self._hip_engine = T.TD3Engine(obs_dim, act_dim, flats["actor"], flats["critic1"],
                                                     flats.get("critic2"), cfg, hidden=self._hip_hidden, depth=self._hip_depth,
                                                     activation=self._hip_actfn)
                eng.cnt = getattr(self, "_cnt", 0)
                for n, mod, optim, keys, conv, _, old in self._hip_parts():           # resume from a checkpoint
                    setattr(eng, n + "_old", conv([old.state_dict()[k] for k in keys], obs_dim, act_dim, dev))
                    ms, vs, step = adam_state(optim._optim, params_by_keys(mod, keys))
                    setattr(eng, n + "_m", conv(ms, obs_dim, act_dim, dev))
                    setattr(eng, n + "_v", conv(vs, obs_dim, act_dim, dev))
                    if n == "actor":
                        eng.actor_steps = step
            return self._hip_engine

        def _preprocess_batch(self, batch, buffer, indices):
            _require_gpu(self._hip_device, "HipTD3 / HipDDPG")
            eng = self._engine()
            m = _mirror(self, buffer, self._hip_device)
            idx = torch.as_tensor(np.asarray(indices, np.int64), device=self._hip_device)
            noise = torch.randn(size=(len(indices), eng.act_dim)) if twin else None        # td3.py:196
            batch.returns = eng.preprocess(m, idx, noise).reshape(-1, 1)
            self._hip_idx = idx
            return batch

        def _update_with_batch(self, batch):
            self._hip_refresh_lr()
            from .buffer import gather_rows_multi

            eng, m = self._hip_engine, self._hip_mirror
            stats, w = eng.update_with_batch(*gather_rows_multi([m.obs, m.act], self._hip_idx),
                                             batch.returns.reshape(-1), getattr(batch, "weight", None))
            batch.weight = w
            if twin:
                self._cnt = eng.cnt
            s = stats.cpu().numpy()
            with torch.no_grad():
                for n, mod, optim, keys, _, back, old in self._hip_parts():
                    params = params_by_keys(mod, keys)
                    for p, t in zip(params, back(getattr(eng, n), eng.obs_dim, eng.act_dim, eng.hidden)):
                        p.copy_(t)
                    for p, t in zip(params_by_keys(old, keys), back(getattr(eng, n + "_old"), eng.obs_dim, eng.act_dim, eng.hidden)):
                        p.copy_(t)
                    step = eng.actor_steps if n == "actor" else eng.cnt
                    store_adam_state(optim._optim, params, back(getattr(eng, n + "_m"), eng.obs_dim, eng.act_dim, eng.hidden),
                                     back(getattr(eng, n + "_v"), eng.obs_dim, eng.act_dim, eng.hidden), step)
            if twin:
                self._last = float(s[0])
                return TD3TrainingStats(actor_loss=float(s[0]), critic1_loss=float(s[1]), critic2_loss=float(s[2]))
            return DDPGTrainingStats(actor_loss=float(s[0]), critic_loss=float(s[1]))

    HipDet.__name__ = "HipTD3" if twin else "HipDDPG"
    return HipDet


def make_hip_td3(ref=None):
    """Returns HipTD3(TD3) (hooks td3.py:190-226 on the engine).  `ref`: namespace replacing the tianshou imports."""
    return _make_hip_det(True, ref)


def make_hip_ddpg(ref=None):
    """Returns HipDDPG(DDPG) (hooks ddpg.py:397-411 on the engine)."""
    return _make_hip_det(False, ref)
