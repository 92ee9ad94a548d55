"""tianshou_amd/widths.py on the CPU: the zero-padding embedding of Net[h1, h2] into Net[h, h] is a function-preserving,
gradient-preserving bijection on the real entries, and padding entries have exactly zero gradient (ReLU and tanh)."""
import pytest
import torch
import torch.nn.functional as F

from tianshou_amd import widths as W


def _net(in_dim, h1, h2, outs, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64) * 0.3  # noqa: E731  (float64: the statement is exact)
    t = [r(h1, in_dim), r(h1), r(h2, h1), r(h2)]
    for o in outs:
        t += [r(o, h2), r(o)]
    return t


def _forward(t, x, act):
    h = act(F.linear(x, t[0], t[1]))
    h = act(F.linear(h, t[2], t[3]))
    return [F.linear(h, t[i], t[i + 1]) for i in range(4, len(t), 2)]


@pytest.mark.parametrize("h1,h2,outs", [(400, 300, [6]), (48, 80, [5, 5]), (24, 56, [1]), (64, 64, [3])])
@pytest.mark.parametrize("act", [torch.relu, torch.tanh])
def test_embedding_preserves_function_and_gradients(h1, h2, outs, act):
    t = _net(11, h1, h2, outs, 3)
    H = W.common_hidden(t)
    assert H % 32 == 0 and H >= max(h1, h2) and H - max(h1, h2) < 32
    p = [x.clone().requires_grad_(True) for x in W.pad_two_layer(t, H)]
    q = [x.clone().requires_grad_(True) for x in t]
    x = torch.randn(37, 11, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    yp, yq = _forward(p, x, act), _forward(q, x, act)
    for a, b in zip(yp, yq):
        torch.testing.assert_close(a, b, rtol=1e-12, atol=1e-12)
    sum(((y - 0.3) ** 2).sum() for y in yp).backward()
    sum(((y - 0.3) ** 2).sum() for y in yq).backward()
    gp = [x.grad for x in p]
    assert W.padding_is_zero(gp, h1, h2)                       # exactly zero: Adam / weight decay / Polyak keep the padding at zero
    for a, b in zip(W.unpad_two_layer(gp, h1, h2), [x.grad for x in q]):
        torch.testing.assert_close(a, b, rtol=1e-12, atol=1e-12)
    back = W.unpad_two_layer([x.detach() for x in p], h1, h2)
    assert all(torch.equal(a, b) for a, b in zip(back, t))
    assert W.padding_is_zero([x.detach() for x in p], h1, h2)


def test_shapes_that_are_not_two_layer_mlps_are_refused():
    t = _net(7, 32, 32, [2], 1)
    with pytest.raises(NotImplementedError):
        W.two_layer_widths([t[0], t[1], t[2][:, :16], t[3], t[4], t[5]])
    with pytest.raises(NotImplementedError):
        W.common_hidden(_net(7, 1100, 64, [2], 1))
    with pytest.raises(ValueError):
        W.pad_two_layer(t, 16)
    assert W.two_layer_widths(t) == (32, 32) and W.pad_two_layer(t, 32)[0] is t[0]
