"""ctypes binding of libtsengine.so (the C ABI declared in include/tsengine.h).

There is NO fallback: if the HIP library is missing or fails to load, importing a kernel
entry point raises.  Tensors cross the boundary as raw device pointers (`tensor.data_ptr()`),
sizes as int64 and the current torch HIP stream as a `hipStream_t` handle; torch is only the
owner of device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TS_LIB_PATH") or os.path.join(_PKG, "lib", "libtsengine.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "tsengine.h")

TS_OK = 0
TS_ERR_INVALID_ARG = -1
TS_ERR_SHAPE = -2
TS_ERR_HIP = -3
TS_ERR_UNSUPPORTED = -4
TS_ERR_WORKSPACE = -5


class EngineError(RuntimeError):
    """A libtsengine entry point returned a negative status."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"libtsengine error {code}: {msg}")
        self.code = code


class PPOHParams(C.Structure):
    """struct ts_ppo_hparams (include/tsengine.h)."""

    _fields_ = [
        ("eps_clip", C.c_double),
        ("dual_clip", C.c_double),
        ("vf_coef", C.c_double),
        ("ent_coef", C.c_double),
        ("max_grad_norm", C.c_double),
        ("lr", C.c_double),
        ("beta1", C.c_double),
        ("beta2", C.c_double),
        ("adam_eps", C.c_double),
        ("value_clip", C.c_int32),
        ("adv_norm", C.c_int32),
        ("algo", C.c_int32),
        ("nets", C.c_int32),
        ("optimizer", C.c_int32),
        ("rms_centered", C.c_int32),
        ("weight_decay", C.c_double),
        ("rms_alpha", C.c_double),
        ("rms_momentum", C.c_double),
        ("max_action", C.c_double),
    ]


class NetDesc(C.Structure):
    """struct ts_net_desc (include/tsengine.h): a Net / MLP trunk of any depth -- `hidden_sizes` and one activation
    (utils/net/common.py:90-178, 246-369)."""

    MAX_HIDDEN = 7
    ACTIVATIONS = {"tanh": 0, "relu": 1, "none": 2}
    CONDITIONED_SIGMA = 1
    LAYERNORM = 2                # MLP(norm_layer=nn.LayerNorm): Linear -> LayerNorm -> activation per hidden layer
    _fields_ = [("obs_dim", C.c_int64), ("n_hidden", C.c_int32), ("activation", C.c_int32), ("hidden", C.c_int64 * 7),
                ("flags", C.c_int64), ("max_action", C.c_double), ("ln_eps", C.c_double)]

    @classmethod
    def make(cls, obs_dim: int, hidden, activation: str, flags: int = 0, max_action: float = 0.0, ln_eps: float = 0.0) -> "NetDesc":
        hidden = [int(h) for h in hidden]
        if not 1 <= len(hidden) <= cls.MAX_HIDDEN:
            raise NotImplementedError(f"trunks of 1 .. {cls.MAX_HIDDEN} hidden layers are supported, got {len(hidden)}")
        if activation not in cls.ACTIVATIONS:
            raise NotImplementedError(f"activation must be one of {sorted(cls.ACTIVATIONS)}, got {activation!r}")
        d = cls(int(obs_dim), len(hidden), cls.ACTIVATIONS[activation], (C.c_int64 * 7)(*(hidden + [0] * (7 - len(hidden)))), int(flags),
                float(max_action or 0.0), float(ln_eps or 0.0))
        return d


_lib = None
KERNEL_KINDS = ("ppo_step", "ppo_reduce", "ppo_adam", "ppo_infer", "gae_maps", "gae_apply",
                "conv_fwd", "conv_wgrad", "conv_dgrad")


def declared_symbols() -> list[str]:
    """Every entry point include/tsengine.h declares (used by the CPU symbol test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ts_[a-z0-9_]+)\s*\(", text)))


def load() -> C.CDLL:
    """Loads the library (building it is __graft_entry__.build()'s / tianshou_amd.build's job)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP engine is not built. Run `python -m tianshou_amd.build` "
            "(needs hipcc, cross-compiles for gfx950 without a GPU). There is no CPU fallback."
        )
    # torch ships its own libamdhip64; it must be in the process before libtsengine resolves its
    # HIP dependency, otherwise two HIP runtimes get loaded and ours sees no device.
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    lib.ts_version.restype = C.c_char_p
    lib.ts_last_error.restype = C.c_char_p
    lib.ts_gae_num_tiles.restype = C.c_int64
    lib.ts_gae_num_tiles.argtypes = [C.c_int64]
    if hasattr(lib, "ts_ppo_param_count"):
        lib.ts_ppo_param_count.restype = C.c_int64
        lib.ts_ppo_param_count.argtypes = [C.c_int64, C.c_int64]
    _lib = lib
    return lib


def check(code: int) -> None:
    if code == TS_OK:
        return
    msg = load().ts_last_error().decode("utf-8", "replace")
    if code == TS_ERR_SHAPE:
        raise ValueError(msg)
    raise EngineError(code, msg)


def ptr(t) -> C.c_void_p:
    """Device pointer of a torch tensor (None -> NULL).  A host tensor here would make the kernels fault on a host
    address, so it is refused up front."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError("libtsengine entry points take device tensors (got a CPU tensor; there is no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError("libtsengine entry points take contiguous tensors")
    return C.c_void_p(t.data_ptr())


def i64(v) -> C.c_int64:
    return C.c_int64(int(v))


def f64(v) -> C.c_double:
    return C.c_double(float(v))


def _raw_stream(device=None) -> int:
    """hipStream_t of torch's current stream on `device` as an integer.  torch.cuda.current_stream() builds a Stream object
    through several layers of Python device-index resolution (4-5 us per call, twenty calls in a launch-bound update); the
    binding underneath it takes the index and returns the handle."""
    import torch

    if isinstance(device, int):
        idx = device
    else:
        idx = getattr(device, "index", None)
        if idx is None:
            idx = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


def current_stream(device=None) -> C.c_void_p:
    return C.c_void_p(_raw_stream(device))


class Workspace:
    """ts_workspace handle bound to one device (one per stream in concurrent use)."""

    def __init__(self, device_index: int = 0, max_bytes: int = 0):
        self._h = C.c_void_p(0)
        check(load().ts_workspace_create(C.byref(self._h), C.c_int(device_index), C.c_size_t(max_bytes)))

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    profiling = False          # between profile_begin and profile_end: engines keep their work on this workspace's stream

    def profile_begin(self) -> None:
        check(load().ts_profile_begin(self._h))
        self.profiling = True

    def profile_end(self) -> dict[str, tuple[float, int]]:
        """-> {kind: (total_ms, launches)} measured with HIP events on the launch stream."""
        n = len(KERNEL_KINDS)
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        check(load().ts_profile_end(self._h, ms, cnt, C.c_int(n)))
        self.profiling = False
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(KERNEL_KINDS)}

    def gae_check(self) -> int:
        """0 if every single-pass GAE scan on this workspace completed its hand-offs (synchronises)."""
        err = C.c_int(0)
        check(load().ts_gae_check(self._h, C.byref(err), current_stream()))
        return int(err.value)

    def close(self) -> None:
        if self._h:
            load().ts_workspace_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass


_default_ws: dict[tuple[int, int], Workspace] = {}


def default_workspace(device_index: int) -> Workspace:
    """The per-(device, current stream) default workspace: scratch areas (the GAE scan's hand-off messages, the
    gradient slabs) are written by the kernels of one stream at a time, so work issued on two streams of one
    process must not share them."""
    import torch

    try:
        stream = int(_raw_stream(device_index))
    except Exception:            # no GPU: Workspace() below raises the real error
        stream = 0
    key = (device_index, stream)
    ws = _default_ws.get(key)
    if ws is None:
        ws = _default_ws[key] = Workspace(device_index)
    return ws


class FrameReplay(C.Structure):
    """struct ts_frame_replay (include/tsengine.h)."""

    _fields_ = [("offset", C.c_void_p), ("E", C.c_int64), ("lengths", C.c_void_p), ("last_index", C.c_void_p),
                ("done", C.c_void_p), ("terminated", C.c_void_p), ("rew", C.c_void_p), ("frames", C.c_void_p),
                ("plane_elems", C.c_int64), ("act_col", C.c_void_p), ("slots", C.c_int64), ("tree", C.c_void_p),
                ("bound", C.c_int64), ("prio_minmax", C.c_void_p), ("alpha", C.c_double), ("beta", C.c_double),
                ("weight_norm", C.c_int32), ("reserved", C.c_int32)]


def aux_workspace(device_index: int) -> Workspace:
    """A second process-lifetime workspace per device for entry points that run two passes at once inside one call
    (ts_rnnq_learn_step's ahead-of-time forward pass beside the target passes)."""
    key = (device_index, -1)
    ws = _default_ws.get(key)
    if ws is None:
        ws = _default_ws[key] = Workspace(device_index)
    return ws
