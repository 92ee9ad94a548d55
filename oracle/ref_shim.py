"""TEST INFRASTRUCTURE ONLY - never imported by the product path (tianshou_amd/).

Import shim that lets the *unmodified* reference (thu-ml/tianshou 2.0.1, mounted
read-only at /root/reference) be imported under Python 3.10 in the authoring
container, so that `oracle/gen_golden.py` can run the reference itself and dump
golden input/output vectors into `tests/golden/`.

The reference needs python ^3.11 plus numba / gymnasium / sensai-utils / h5py /
overrides / deepdiff / tensorboard / pettingzoo, none of which are installed here
(SURVEY.md section 8c, Appendix B).  Every stub below is behaviour-neutral for the
hot path:

* ``numba.njit`` -> identity: the 8 njit bodies (algorithm_base.py:1085-1222,
  manager.py:311-363, segtree.py:95-134) are NumPy-legal and execute with the
  same arithmetic, only slower.
* everything else is import-time scaffolding (typing.Self, StrEnum, logging,
  pickling helpers, gymnasium space classes).

/root/reference does not exist on the GPU box; nothing under tests/ -m gpu,
bench.py or __graft_entry__.smoke() may call `install()`.
"""
from __future__ import annotations

import enum
import logging as _logging
import os
import sys
import types
import typing

REFERENCE_ROOT = os.environ.get("TIANSHOU_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "tianshou"))


def _module(name: str, **attrs) -> types.ModuleType:
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, mod)
    return mod


class _Dummy:
    """Permissive placeholder class used for names that are only type hints."""

    def __init__(self, *a, **k):
        pass

    def __init_subclass__(cls, **k):
        pass

    def __class_getitem__(cls, item):
        return cls


def _lazy_module(name: str, **attrs) -> types.ModuleType:
    """Module whose unknown attributes resolve to fresh dummy classes."""
    mod = _module(name, **attrs)

    def __getattr__(attr: str):
        if attr.startswith("__"):
            raise AttributeError(attr)
        cls = type(attr, (_Dummy,), {})
        setattr(mod, attr, cls)
        return cls

    mod.__getattr__ = __getattr__  # type: ignore[attr-defined]
    return mod


_installed = False


def install() -> None:
    """Install the stubs and put the reference on sys.path (idempotent)."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(
            f"reference not found at {REFERENCE_ROOT}; golden vectors can only be "
            "regenerated in the authoring container"
        )

    # --- stdlib back-ports ---------------------------------------------------
    if not hasattr(typing, "Self"):
        import typing_extensions

        typing.Self = typing_extensions.Self  # type: ignore[attr-defined]
    if not hasattr(enum, "StrEnum"):

        class StrEnum(str, enum.Enum):
            def __str__(self) -> str:
                return str(self.value)

        enum.StrEnum = StrEnum  # type: ignore[attr-defined]

    # --- numba: njit -> identity ----------------------------------------------
    def njit(f=None, **k):
        if callable(f):
            return f
        return lambda g: g

    _module("numba", njit=njit, jit=njit)

    # --- overrides --------------------------------------------------------------
    _module("overrides", override=lambda f: f, overrides=lambda f: f)

    # --- h5py ---------------------------------------------------------------------
    _module("h5py", File=_Dummy, Dataset=_Dummy, Group=_Dummy)

    # --- deepdiff -----------------------------------------------------------------
    class DeepDiff(dict):
        def __init__(self, *a, **k):
            super().__init__()

    _module("deepdiff", DeepDiff=DeepDiff)

    # --- sensai ---------------------------------------------------------------------
    _module("sensai")
    _module("sensai.util")

    def pickle_hash(o, *a, **k):
        import hashlib
        import pickle

        return hashlib.sha1(pickle.dumps(o)).hexdigest()

    _module("sensai.util.hash", pickle_hash=pickle_hash)
    _module(
        "sensai.util.helper",
        mark_used=lambda *a, **k: None,
        count_none=lambda *a: sum(x is None for x in a),
    )

    def setstate(cls, obj, state, **k):
        new_defaults = k.get("new_default_properties") or {}
        for key, val in new_defaults.items():
            state.setdefault(key, val)
        obj.__dict__.update(state)

    _module(
        "sensai.util.pickle",
        setstate=setstate,
        dump_pickle=lambda *a, **k: None,
        load_pickle=lambda *a, **k: None,
    )

    class ToStringMixin:
        def __repr__(self) -> str:
            return f"{type(self).__name__}()"

        def _tostring_excludes(self):
            return []

        def _tostring_includes(self):
            return []

    _module("sensai.util.string", ToStringMixin=ToStringMixin)
    sl = _module("sensai.util.logging")
    sl.__dict__.update({k: v for k, v in _logging.__dict__.items() if not k.startswith("__")})
    sl.set_configure_callback = lambda *a, **k: None
    sl.datetime_tag = lambda: "00000000-000000"
    sl.run_main = lambda f: f()
    sl.run_cli = lambda f: f()
    sl.configure = lambda *a, **k: None
    sl.add_file_logger = lambda *a, **k: None
    sl.remove_log_handler = lambda *a, **k: None
    sl.FileLoggerContext = _Dummy
    sys.modules["sensai.util"].logging = sl
    _module("sensai.util.git", GitStatus=_Dummy, git_status=lambda *a, **k: None)

    # --- gymnasium ----------------------------------------------------------------------
    import numpy as np

    class Space(_Dummy):
        shape = None
        dtype = None

        def sample(self):
            raise NotImplementedError

        def seed(self, *a, **k):
            return []

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32, **k):
            if shape is None:
                shape = np.shape(low)
            self.shape = tuple(shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()
            self.dtype = np.dtype(dtype)

    class Discrete(Space):
        def __init__(self, n, start=0, **k):
            self.n = int(n)
            self.start = start
            self.shape = ()
            self.dtype = np.dtype(np.int64)

    class MultiDiscrete(Space):
        def __init__(self, nvec, **k):
            self.nvec = np.asarray(nvec)
            self.shape = self.nvec.shape

    class MultiBinary(Space):
        def __init__(self, n, **k):
            self.n = n
            self.shape = (n,) if isinstance(n, int) else tuple(n)

    class Tuple(Space, tuple):
        pass

    class Dict(Space, dict):
        pass

    gym = _lazy_module("gymnasium", __version__="1.0.0", Space=Space)
    spaces = _lazy_module(
        "gymnasium.spaces",
        Space=Space,
        Box=Box,
        Discrete=Discrete,
        MultiDiscrete=MultiDiscrete,
        MultiBinary=MultiBinary,
        Tuple=Tuple,
        Dict=Dict,
    )
    _lazy_module("gymnasium.spaces.discrete", Discrete=Discrete)
    for sub in (
        "vector",
        "core",
        "wrappers",
        "envs",
        "envs.registration",
        "utils",
        "error",
    ):
        _lazy_module(f"gymnasium.{sub}")
    gym.spaces = spaces

    # --- pettingzoo --------------------------------------------------------------------------
    _lazy_module("pettingzoo", __version__="1.24.0")
    _lazy_module("pettingzoo.utils")
    _lazy_module("pettingzoo.utils.env")
    _lazy_module("pettingzoo.utils.wrappers")

    # --- tensorboard -----------------------------------------------------------------------------
    _lazy_module("tensorboard")
    _lazy_module("tensorboard.backend")
    _lazy_module("tensorboard.backend.event_processing")
    _lazy_module("tensorboard.backend.event_processing.event_accumulator")
    import torch.utils  # noqa: F401

    _lazy_module("torch.utils.tensorboard")

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True
