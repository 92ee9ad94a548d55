"""Second-generation implicit-GEMM kernels (tianshou_amd/csrc/ts_conv2.hip: weight block resident in LDS, border classes
for the input gradient, LDS-ring weight gradient) against the first generation and a float64 torch reference.

They replace nn.Conv2d / nn.Linear forward + autograd (tianshou/env/atari/atari_network.py:79-98, utils/net/common.py:172-178)
for large row counts (minibatch 65,536 of the Atari-shape PPO update); here they are forced on at test sizes with
ts_conv_set_generation(1).  Bars: 5e-6 of the output scale against float64 (fp32 accumulation over K <= 3136), and the same
against generation 1 (different summation order: k = 16 h + j inside a 32-chunk instead of sequential)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture()
def gen2():
    from tianshou_amd import _lib

    lib = _lib.load()
    prev = lib.ts_conv_set_generation(1)
    yield lib
    lib.ts_conv_set_generation(prev)


def _ref64(x, wb, K, S, dy, mask):
    IC, OC = x.shape[-1], wb.shape[1]
    w = wb[:-1].double().reshape(K, K, IC, OC).permute(3, 2, 0, 1)
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, wb[-1].double(), stride=S)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    gx = xr.grad.permute(0, 2, 3, 1)
    if mask is not None:
        gx = gx * (mask > 0).double()
    gw = torch.cat([wr.grad.permute(2, 3, 1, 0).reshape(K * K * IC, OC), dy.double().sum((0, 1, 2))[None]], 0)
    return yr.detach().permute(0, 2, 3, 1), gx, gw


SHAPES = [  # name, B, IH, IW, IC, K, S, OC, uint8 input
    ("conv1_u8", 37, 84, 84, 4, 8, 4, 32, True),        # resident, 32 columns, uint8 frames converted on load
    ("conv1", 19, 84, 84, 4, 8, 4, 32, False),
    ("conv2", 53, 20, 20, 32, 4, 2, 64, False),          # 36 border classes in the input gradient
    ("conv3", 96, 9, 9, 64, 3, 1, 64, False),            # 25 border classes, 144 KB weight block
    ("fc1", 300, 1, 1, 3136, 1, 1, 512, False),          # streamed weight slices; dgrad = forward over W^T (ragged 3136)
    ("head", 1000, 1, 1, 512, 1, 1, 32, False),
    ("hidden", 700, 1, 1, 384, 1, 1, 256, False),        # two resident 128-column blocks
    ("ragged", 513, 1, 1, 64, 1, 1, 96, False),          # streamed, 96 of 128 columns; single reduction chunk pair
    ("uncovered", 19, 19, 19, 32, 4, 2, 32, False),      # input rows / columns no window reaches: dX = 0 there
    ("one_chunk", 5000, 1, 1, 32, 1, 1, 96, False),      # K = 32: one (odd) reduction chunk
]


@pytest.mark.parametrize("name,B,IH,IW,IC,K,S,OC,u8", SHAPES, ids=[s[0] for s in SHAPES])
def test_generation_2_layers(gen2, name, B, IH, IW, IC, K, S, OC, u8):
    from tianshou_amd import dqn as D

    torch.manual_seed(3)
    x = torch.randint(0, 256, (B, IH, IW, IC), device="cuda", dtype=torch.uint8) if u8 else \
        torch.randn(B, IH, IW, IC, device="cuda")
    wb = torch.randn(K * K * IC + 1, OC, device="cuda") * 0.05
    can_dx = IC % 32 == 0 and K % S == 0
    mask = (torch.rand(x.shape, device="cuda") > 0.5).float() if can_dx and name in ("conv2", "conv3", "fc1") else None
    out = {}
    for g in (-1, 1):
        gen2.ts_conv_set_generation(g)
        y = D.conv_forward(x, wb, K, K, S, True)
        dy = torch.randn(y.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
        d_wb, dx = D.conv_backward(x, wb, dy, K, K, S, mask=mask, need_dx=can_dx)
        out[g] = (y, d_wb, dx)
    yr, gx, gw = _ref64(x.float(), wb, K, S, dy, mask)
    y1, w1, dx1 = out[-1]
    y2, w2, dx2 = out[1]
    scale = float(yr.abs().max())
    assert float((y2.double() - yr.clamp(min=0)).abs().max()) <= 5e-6 * scale
    assert float((y2 - y1).abs().max()) <= 5e-6 * scale
    assert float((w2.double() - gw).abs().max()) <= 3e-6 * float(gw.abs().max())
    assert float((w2 - w1).abs().max()) <= 3e-6 * float(gw.abs().max())
    if can_dx:
        assert float((dx2.double() - gx).abs().max()) <= 5e-6 * float(gx.abs().max())
        assert float((dx2 - dx1).abs().max()) <= 5e-6 * float(gx.abs().max())
        if name == "uncovered":
            assert float(dx2[:, 18].abs().max()) == 0.0 and float(dx2[:, :, 18].abs().max()) == 0.0


def test_generation_knob_round_trips(gen2):
    assert gen2.ts_conv_set_generation(0) == 1
    assert gen2.ts_conv_set_generation(-5) == 0
    assert gen2.ts_conv_set_generation(1) == -1


@pytest.mark.parametrize("B,masked", [(53, True), (300, False)])
def test_pixel_shuffle_input_gradient_is_bit_identical(gen2, monkeypatch, B, masked):
    """conv2's input gradient with the four stride parities of a super-pixel as the column blocks of ONE GEMM (Rows2Args.ps: 128
    columns, every dY operand fetched once for four MFMAs) against one GEMM per parity (TS_DGRAD_PS=0): the same taps in the same
    order per output element, so the results are equal bit for bit -- with and without the ReLU mask."""
    from tianshou_amd import dqn as D

    torch.manual_seed(5)
    IH = IW = 20; IC = 32; K = 4; S = 2; OC = 64
    x = torch.randn(B, IH, IW, IC, device="cuda")
    wb = torch.randn(K * K * IC + 1, OC, device="cuda") * 0.05
    mask = (torch.rand(x.shape, device="cuda") > 0.5).float() if masked else None
    dy = torch.randn(B, 9, 9, OC, device="cuda")
    out = {}
    for ps in ("1", "0"):
        monkeypatch.setenv("TS_DGRAD_PS", ps)
        out[ps] = D.conv_backward(x, wb, dy, K, K, S, mask=mask, need_dx=True)[1].clone()
    assert torch.equal(out["1"], out["0"])
    _, gx, _ = _ref64(x, wb, K, S, dy, mask)
    assert float((out["1"].double() - gx).abs().max()) <= 5e-6 * float(gx.abs().max())
