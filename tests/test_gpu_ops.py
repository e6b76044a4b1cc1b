"""GPU parity of the single operators (through the C ABI) against the CPU oracle.

Tolerance: the contract is 1e-4 max relative error in fp32 (max|a-b| / max|b|
per tensor); these operator-level checks use 2e-5, the whole-block and
whole-network checks use the contract value."""
import pytest
import torch

from oracle import restormer_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 2e-5


def maxrel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _ops():
    from superresolution_amd import ops
    return ops


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


LN_SHAPES = [(2, 48, 16, 24), (1, 96, 8, 12), (2, 192, 5, 7), (1, 384, 4, 6), (2, 12, 3, 9), (1, 96, 64, 64), (3, 24, 7, 5)]


@pytest.mark.parametrize("shape", LN_SHAPES)
@pytest.mark.parametrize("with_bias", [True, False])
def test_layernorm(shape, with_bias):
    ops = _ops()
    x = _rand(*shape, seed=1) + 0.3
    w = _rand(shape[1], seed=2) * 0.3 + 1.0
    b = _rand(shape[1], seed=3) * 0.1 if with_bias else None
    r = _rand(*shape, seed=4)
    xo = x.clone().requires_grad_(True); wo = w.clone().requires_grad_(True)
    bo = b.clone().requires_grad_(True) if with_bias else None
    yo = orc.layernorm_nchw(xo, wo, bo)
    (yo * r).sum().backward()

    xg = x.float().cuda().requires_grad_(True); wg = w.float().cuda().requires_grad_(True)
    bg = b.float().cuda().requires_grad_(True) if with_bias else None
    yg = ops.layernorm(xg, wg, bg)
    (yg * r.float().cuda()).sum().backward()
    assert maxrel(yg, yo) < TOL
    assert maxrel(xg.grad, xo.grad) < TOL
    assert maxrel(wg.grad, wo.grad) < TOL
    if with_bias:
        assert maxrel(bg.grad, bo.grad) < TOL


CONV_SHAPES = [  # B, Cin, Cout, H, W
    (2, 48, 144, 16, 24), (1, 96, 240, 8, 12), (2, 60, 48, 5, 7), (1, 384, 192, 4, 6), (2, 15, 12, 3, 9),
    (1, 96, 288, 32, 32), (1, 384, 1152, 6, 6), (2, 120, 96, 16, 16), (1, 192, 96, 10, 13),
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
@pytest.mark.parametrize("bias", [False, True])
def test_conv1x1(shape, bias):
    ops = _ops()
    B, Cin, Cout, H, W = shape
    x = _rand(B, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, 1, 1, seed=2) / Cin ** 0.5
    b = _rand(Cout, seed=3) * 0.1 if bias else None
    r = _rand(B, Cout, H, W, seed=4)
    xo = x.clone().requires_grad_(True); wo = w.clone().requires_grad_(True)
    bo = b.clone().requires_grad_(True) if bias else None
    yo = orc.pointwise_conv(xo, wo, bo)
    (yo * r).sum().backward()
    xg = x.float().cuda().requires_grad_(True); wg = w.float().cuda().requires_grad_(True)
    bg = b.float().cuda().requires_grad_(True) if bias else None
    yg = ops.conv1x1(xg, wg, bg)
    (yg * r.float().cuda()).sum().backward()
    assert maxrel(yg, yo) < TOL
    assert maxrel(xg.grad, xo.grad) < TOL
    assert maxrel(wg.grad, wo.grad) < TOL
    if bias:
        assert maxrel(bg.grad, bo.grad) < TOL


DW_SHAPES = [(2, 144, 16, 24), (1, 240, 8, 12), (2, 30, 5, 7), (1, 36, 3, 9), (1, 8, 64, 64), (2, 5, 1, 1), (1, 3, 2, 130)]


@pytest.mark.parametrize("shape", DW_SHAPES)
@pytest.mark.parametrize("bias", [False, True])
def test_dwconv3x3(shape, bias):
    ops = _ops()
    B, C, H, W = shape
    x = _rand(B, C, H, W, seed=1)
    w = _rand(C, 1, 3, 3, seed=2) / 3.0
    b = _rand(C, seed=3) * 0.1 if bias else None
    r = _rand(B, C, H, W, seed=4)
    xo = x.clone().requires_grad_(True); wo = w.clone().requires_grad_(True)
    bo = b.clone().requires_grad_(True) if bias else None
    yo = orc.depthwise3x3_explicit(xo, wo, bo)
    (yo * r).sum().backward()
    xg = x.float().cuda().requires_grad_(True); wg = w.float().cuda().requires_grad_(True)
    bg = b.float().cuda().requires_grad_(True) if bias else None
    yg = ops.dwconv3x3(xg, wg, bg)
    (yg * r.float().cuda()).sum().backward()
    assert maxrel(yg, yo) < TOL
    assert maxrel(xg.grad, xo.grad) < TOL
    assert maxrel(wg.grad, wo.grad) < TOL
    if bias:
        assert maxrel(bg.grad, bo.grad) < TOL


@pytest.mark.parametrize("shape", [(2, 6, 8, 12), (1, 48, 64, 64), (3, 5, 6, 10), (1, 2, 2, 2)])
def test_pixel_shuffles_exact(shape):
    """Permutations: bit-exact, and each is the other's inverse and backward."""
    ops = _ops()
    x = _rand(*shape, seed=5).float()
    xg = x.cuda()
    un = ops.pixel_unshuffle2(xg)
    assert torch.equal(un.cpu(), orc.pixel_unshuffle2(x))
    assert torch.equal(ops.pixel_shuffle2(un).cpu(), x)
    b, c, h, w = shape
    if c % 4 == 0:
        sh = ops.pixel_shuffle2(xg)
        assert torch.equal(sh.cpu(), orc.pixel_shuffle2(x))
    xr = xg.clone().requires_grad_(True)
    y = ops.pixel_unshuffle2(xr)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    assert torch.equal(xr.grad.cpu(), orc.pixel_shuffle2(r.cpu()))


def test_errors_are_loud():
    ops = _ops()
    with pytest.raises(RuntimeError):
        ops.layernorm(torch.randn(1, 4, 2, 2), torch.ones(4), None)          # CPU tensor
    with pytest.raises(RuntimeError):
        ops.layernorm(torch.randn(1, 4, 2, 2, device="cuda").half(), torch.ones(4, device="cuda").half(), None)
    with pytest.raises(RuntimeError):
        ops.pixel_unshuffle2(torch.randn(1, 4, 3, 2, device="cuda"))


@pytest.mark.parametrize("shape", [(2, 48, 3, 32, 40), (1, 12, 4, 9, 7), (2, 5, 1, 6, 10), (1, 24, 2, 64, 64), (1, 3, 4, 5, 5)])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("residual", [False, True])
def test_conv3x3_fewout(shape, bias, residual):
    """Last layer of the network: dense 3x3, few output channels, fused residual; forward through the
    HIP kernel, gradients through torch's convolution gradients, both against the fp64 oracle."""
    ops = _ops()
    B, Cin, Cout, H, W = shape
    x = _rand(B, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, 3, 3, seed=2) * 0.2
    b = _rand(Cout, seed=3) if bias else None
    r = _rand(B, Cout, H, W, seed=4) if residual else None
    up = _rand(B, Cout, H, W, seed=5)
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = orc.conv3x3(xo, wo, b) + (r if residual else 0.0)
    (ref * up).sum().backward()
    xg = x.float().cuda().requires_grad_(True)
    wg = w.float().cuda().requires_grad_(True)
    got = ops.conv3x3_fewout(xg, wg, None if b is None else b.float().cuda(), None if r is None else r.float().cuda())
    (got * up.float().cuda()).sum().backward()
    assert maxrel(got, ref) < TOL
    assert maxrel(xg.grad, xo.grad) < 1e-4
    assert maxrel(wg.grad, wo.grad) < 1e-4


def test_scratch_above_the_keep_limit_is_handed_back():
    """ops.workspace grows on demand; trim_workspaces (called by Restormer.forward after an inference call) drops the
    buffers above the limit and keeps the small ones."""
    from superresolution_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    ops.release_workspaces()
    small = ops.workspace(2 << 20, dev)
    assert ops.workspace(1 << 20, dev) is small                     # grow-only: a smaller request reuses it
    ops.trim_workspaces(8 << 20)
    assert ops.workspace(1 << 20, dev) is small
    big = ops.workspace(64 << 20, dev)
    assert big.numel() >= 64 << 20
    ops.trim_workspaces(8 << 20)
    assert not ops._workspaces
    assert ops.workspace(1 << 20, dev).numel() < 64 << 20
