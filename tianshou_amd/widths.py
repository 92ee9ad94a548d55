"""MLPs of ANY hidden widths on the engines written for Net[h, ..., h] (two hidden layers; since round 6 any 1 .. 6).

The reference's `Net(hidden_sizes=[h1, h2])` (tianshou/utils/net/common.py:246-369) takes any two widths -- [400, 300] is the
classic DDPG / TD3 architecture; the fused three-layer kernels of the SAC family and the NPG / TRPO passes take ONE width h
that is a multiple of 32.  A Net[h1, h2] is embedded into Net[h, h], h = max(h1, h2) rounded up to 32, by zero padding:
a padding unit has zero weights and bias on its input side and zero weights on its output side, so (ReLU(0) = tanh(0) = 0) it
contributes nothing to any activation, its output-side weights see a zero activation and its input-side weights a zero
upstream gradient: every gradient of a padding entry is exactly zero, Adam (m = v = 0 -> 0 / eps), weight decay and Polyak
averaging keep it at zero, the global gradient norm and the Fisher-vector products do not see it.  The embedded network IS the
reference's network, update after update (fixtures tests/golden/{sac,td3}_widths.npz, written by the unmodified reference).
Tensors here are in torch's nn.Linear layout: [w1 (h1, in), b1, w2 (h2, h1), b2, then heads (out, h2), bias, ...]."""
from __future__ import annotations

import torch

MAX_HIDDEN = 1024


def round32(n: int) -> int:
    return (int(n) + 31) // 32 * 32


def two_layer_widths(t: list[torch.Tensor]) -> tuple[int, int]:
    """(h1, h2) of a tensor list [w1, b1, w2, b2, ...]; raises if the shapes are not those of a two-hidden-layer MLP."""
    h1, h2 = int(t[0].shape[0]), int(t[2].shape[0])
    if t[2].dim() != 2 or int(t[2].shape[1]) != h1 or any(t[i].dim() != 2 or int(t[i].shape[1]) != h2 for i in range(4, len(t), 2)):
        raise NotImplementedError("a two-hidden-layer MLP [w1, b1, w2, b2, heads...] is required")
    return h1, h2


def common_hidden(*tensor_lists: list[torch.Tensor]) -> int:
    """The engine width for several networks that share one engine: the largest width, rounded up to a multiple of 32."""
    h = round32(max(max(two_layer_widths(t)) for t in tensor_lists))
    if not 32 <= h <= MAX_HIDDEN:
        raise NotImplementedError(f"hidden widths up to {MAX_HIDDEN} (got {h})")
    return h


def pad_two_layer(t: list[torch.Tensor], hidden: int) -> list[torch.Tensor]:
    """Net[h1, h2] tensors -> the tensors of the Net[hidden, hidden] that computes the same function (zeros elsewhere)."""
    h1, h2 = two_layer_widths(t)
    H = int(hidden)
    if h1 == H and h2 == H:
        return list(t)
    if h1 > H or h2 > H:
        raise ValueError(f"cannot embed widths ({h1}, {h2}) into {H}")
    t = [x.detach() for x in t]
    w1 = t[0].new_zeros((H, t[0].shape[1])); w1[:h1] = t[0]
    b1 = t[1].new_zeros(H); b1[:h1] = t[1]
    w2 = t[2].new_zeros((H, H)); w2[:h2, :h1] = t[2]
    b2 = t[3].new_zeros(H); b2[:h2] = t[3]
    out = [w1, b1, w2, b2]
    for i in range(4, len(t), 2):
        w = t[i].new_zeros((t[i].shape[0], H)); w[:, :h2] = t[i]
        out += [w, t[i + 1]]
    return out


def unpad_two_layer(t: list[torch.Tensor], h1: int, h2: int) -> list[torch.Tensor]:
    """The inverse of `pad_two_layer`: the Net[h1, h2] entries of Net[hidden, hidden] tensors."""
    if int(t[0].shape[0]) == h1 and int(t[2].shape[0]) == h2:
        return list(t)
    out = [t[0][:h1].contiguous(), t[1][:h1].contiguous(), t[2][:h2, :h1].contiguous(), t[3][:h2].contiguous()]
    for i in range(4, len(t), 2):
        out += [t[i][:, :h2].contiguous(), t[i + 1]]
    return out


def padding_is_zero(t: list[torch.Tensor], h1: int, h2: int) -> bool:
    """Whether every padding entry of Net[hidden, hidden] tensors is exactly zero (what the embedding guarantees)."""
    ok = not bool(t[0][h1:].any()) and not bool(t[1][h1:].any()) and not bool(t[2][h2:].any()) and not bool(t[2][:, h1:].any()) \
        and not bool(t[3][h2:].any())
    return ok and all(not bool(t[i][:, h2:].any()) for i in range(4, len(t), 2))


# ---- any number of hidden layers (round 6): the same embedding, layer by layer ------------------------------------
# Tensor lists are [w1, b1, ..., wd, bd, head_w, head_b, ...] with `heads` (weight, bias) pairs at the end.
MAX_DEPTH = 6


def depth_of(t: list, heads: int) -> int:
    d = len(t) // 2 - heads
    if len(t) % 2 or not 1 <= d <= MAX_DEPTH:
        raise NotImplementedError(f"an MLP of 1 .. {MAX_DEPTH} hidden layers and {heads} head(s) is required (got {len(t)} tensors)")
    return d


def layer_widths(t: list[torch.Tensor], heads: int = 1) -> tuple[int, ...]:
    """(h1, ..., hd) of [w1, b1, ..., wd, bd, heads...]; raises if the shapes do not chain like an MLP's."""
    d = depth_of(t, heads)
    hs = tuple(int(t[2 * i].shape[0]) for i in range(d))
    ok = all(t[2 * i].dim() == 2 and (i == 0 or int(t[2 * i].shape[1]) == hs[i - 1]) and tuple(t[2 * i + 1].shape) == (hs[i],) for i in range(d))
    ok = ok and all(t[i].dim() == 2 and int(t[i].shape[1]) == hs[-1] for i in range(2 * d, len(t), 2))
    if not ok:
        raise NotImplementedError("the tensors are not those of an MLP [w1, b1, ..., wd, bd, heads...]")
    return hs


def engine_hidden(sizes) -> int:
    """The engine width for networks of the given hidden widths sharing one engine: the largest, rounded up to 32."""
    h = round32(max(max(s) for s in sizes))
    if not 32 <= h <= MAX_HIDDEN:
        raise NotImplementedError(f"hidden widths up to {MAX_HIDDEN} (got {h})")
    return h


def pad_layers(t: list[torch.Tensor], hidden: int, heads: int = 1) -> list[torch.Tensor]:
    """Net[h1, ..., hd] tensors -> the tensors of the Net[hidden] * d that computes the same function (zeros elsewhere)."""
    hs = layer_widths(t, heads)
    H, d = int(hidden), len(hs)
    if all(h == H for h in hs):
        return list(t)
    if max(hs) > H:
        raise ValueError(f"cannot embed widths {hs} into {H}")
    t = [x.detach() for x in t]
    out = []
    for i in range(d):
        w = t[2 * i].new_zeros((H, t[2 * i].shape[1] if i == 0 else H))
        w[:hs[i], :t[2 * i].shape[1]] = t[2 * i]
        b = t[2 * i + 1].new_zeros(H)
        b[:hs[i]] = t[2 * i + 1]
        out += [w, b]
    for i in range(2 * d, len(t), 2):
        w = t[i].new_zeros((t[i].shape[0], H))
        w[:, :hs[-1]] = t[i]
        out += [w, t[i + 1]]
    return out


def unpad_layers(t: list[torch.Tensor], sizes) -> list[torch.Tensor]:
    """The inverse of `pad_layers`: the Net[*sizes] entries of Net[hidden] * d tensors."""
    d = len(sizes)
    if all(int(t[2 * i].shape[0]) == sizes[i] for i in range(d)):
        return list(t)
    out = []
    for i in range(d):
        w = t[2 * i][:sizes[i]] if i == 0 else t[2 * i][:sizes[i], :sizes[i - 1]]
        out += [w.contiguous(), t[2 * i + 1][:sizes[i]].contiguous()]
    for i in range(2 * d, len(t), 2):
        out += [t[i][:, :sizes[-1]].contiguous(), t[i + 1]]
    return out


def padding_is_zero_layers(t: list[torch.Tensor], sizes) -> bool:
    """Whether every padding entry of Net[hidden] * d tensors is exactly zero (what the embedding guarantees)."""
    d = len(sizes)
    for i in range(d):
        if bool(t[2 * i][sizes[i]:].any()) or bool(t[2 * i + 1][sizes[i]:].any()) or (i > 0 and bool(t[2 * i][:, sizes[i - 1]:].any())):
            return False
    return all(not bool(t[i][:, sizes[-1]:].any()) for i in range(2 * d, len(t), 2))
