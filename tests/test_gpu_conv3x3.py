"""Dense 3x3 convolution with the pixel (un)shuffle epilogue (rst_conv3x3_fwd) against the oracle's conv3x3 +
pixel_unshuffle2 / pixel_shuffle2 in fp64, at every tile geometry the kernel has (W = 32, 64, 128, 256 and widths
whose last tile is partial: 40 ... 960, widths that are not multiples of 8), channel counts that pad (3, 24, 48), the
network's own shapes; and its
backward (rst_conv3x3_bwd: dx, dW, db) against the fp64 oracle's autograd at the same shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _ref(x, w, b, mode):
    from oracle import restormer_oracle as orc
    y = orc.conv3x3(x.double(), w.double(), None if b is None else b.double())
    return {0: lambda t: t, 1: orc.pixel_unshuffle2, 2: orc.pixel_shuffle2}[mode](y)


CASES = [  # B, Cin, Cout, H, W, mode, bias
    (2, 3, 48, 8, 128, 0, False), (1, 3, 48, 4, 256, 0, True), (2, 48, 24, 6, 128, 1, False), (1, 96, 48, 4, 64, 1, False),
    (1, 192, 96, 8, 32, 1, False), (1, 384, 768, 4, 32, 2, False), (1, 192, 384, 4, 64, 2, False), (2, 96, 192, 2, 128, 2, False),
    (1, 96, 192, 3, 256, 2, False), (1, 48, 40, 5, 384, 0, True), (1, 33, 20, 4, 32, 2, True), (1, 70, 130, 2, 64, 0, False),
    # few input channels: the direct fp32 kernel (one row, narrow rows, 1 / 2 / 4 channels, an output count of 4 k)
    (2, 3, 48, 1, 128, 0, True), (1, 3, 48, 8, 32, 0, False), (1, 1, 8, 4, 64, 0, True), (2, 2, 12, 4, 128, 0, False),
    (1, 4, 20, 2, 256, 0, True), (1, 3, 6, 4, 64, 0, True),   # 6 outputs: not a multiple of 4, stays on the matrix kernel
    # widths that are not whole tiles (the last 128-pixel tile of a row partly outside the image): the levels of a
    # 320-pixel tiled window (320 / 160 / 80 / 40) and of a 3840-pixel frame (960 / 480)
    (1, 384, 768, 5, 40, 2, False), (1, 192, 96, 6, 80, 1, False), (1, 96, 48, 4, 160, 1, True), (1, 48, 24, 6, 320, 1, False),
    (1, 96, 192, 3, 480, 2, False), (1, 48, 40, 3, 960, 0, True), (2, 96, 192, 7, 320, 2, False), (2, 3, 48, 5, 40, 0, True),
    (1, 48, 24, 30, 32, 1, False),          # 32 wide, height not in whole 4-row tiles: rows of one 128-pixel tile
    (1, 33, 20, 3, 72, 2, True), (1, 24, 48, 9, 136, 0, False),
    # widths that are not multiples of 8 (the deeper levels of images whose width is not a multiple of 64, e.g. 200 ->
    # 100 / 50 / 25): zero columns added in the workspace, the result cropped
    (1, 384, 768, 5, 25, 2, False), (2, 192, 96, 6, 50, 1, False), (1, 96, 48, 4, 100, 1, True), (1, 96, 192, 3, 5, 2, False),
    (2, 384, 768, 1, 1, 2, False), (1, 24, 16, 7, 13, 0, True), (1, 48, 24, 2, 2, 1, False), (1, 3, 48, 5, 30, 0, True),
    (1, 96, 192, 4, 140, 2, True),
]


@pytest.mark.parametrize("B,Cin,Cout,H,W,mode,bias", CASES)
def test_conv3x3_vs_oracle(B, Cin, Cout, H, W, mode, bias):
    from superresolution_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H + W + mode)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g) if bias else None
    assert ops.conv3x3_supported(x.cuda(), w.cuda(), mode)
    y = ops.conv3x3(x.cuda(), w.cuda(), None if b is None else b.cuda(), mode)
    ref = _ref(x, w, b, mode)
    assert tuple(y.shape) == tuple(ref.shape)
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < TOL, err


@pytest.mark.parametrize("B,Cin,Cout,H,W,mode,bias", [c for i, c in enumerate(CASES) if i % 3 == 0])
def test_conv3x3_on_the_exact_fp32_instructions(B, Cin, Cout, H, W, mode, bias):
    """rst_set_exact_fp32(1): the same entry point on the exact-fp32 matrix instruction (nine shifted copies + one
    product), with activations far beyond the f16 range of the default route."""
    from superresolution_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H + W + mode + 1)
    x = torch.randn(B, Cin, H, W, generator=g) * 3.0e5
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g) * 1.0e5 if bias else None
    lib.rst_set_exact_fp32(1)
    try:
        assert ops.conv3x3_supported(x.cuda(), w.cuda(), mode)
        y = ops.conv3x3(x.cuda(), w.cuda(), None if b is None else b.cuda(), mode)
    finally:
        lib.rst_set_exact_fp32(0)
    ref = _ref(x, w, b, mode)
    assert tuple(y.shape) == tuple(ref.shape) and torch.isfinite(y).all()
    assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < TOL


def test_conv3x3_image_scale_inputs_and_borders():
    """Image-like input (all positive, mean far from zero) and a weight of ones: every border pixel must see exactly its
    zero padding (a wrong halo shows as an O(1) error)."""
    from superresolution_amd import ops
    x = torch.rand(1, 48, 4, 256, generator=torch.Generator().manual_seed(1)) + 3.0
    w = torch.ones(24, 48, 3, 3) / 432.0
    y = ops.conv3x3(x.cuda(), w.cuda(), None, 0)
    ref = _ref(x, w, None, 0)
    assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < TOL


def test_conv3x3_gradients_match_torch():
    from superresolution_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 48, 4, 64, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(24, 48, 3, 3, generator=g) / 20).cuda().requires_grad_(True)
    r = torch.randn(1, 96, 2, 32, generator=g).cuda()
    (ops.conv3x3(x, w, None, 1) * r).sum().backward()
    gx, gw = x.grad.clone(), w.grad.clone()
    x.grad = None; w.grad = None
    (torch.nn.functional.pixel_unshuffle(torch.nn.functional.conv2d(x, w, padding=1), 2) * r).sum().backward()
    assert torch.allclose(gx, x.grad, rtol=1e-4, atol=1e-5) and torch.allclose(gw, w.grad, rtol=1e-4, atol=1e-4)


def test_unsupported_width_is_reported():
    from superresolution_amd import ops
    for W in (40, 80, 160, 320, 480, 960):        # the levels of 320-pixel tiled windows and of a 3840-pixel frame
        assert ops.conv3x3_supported(torch.zeros(1, 8, 8, W, device="cuda"), torch.zeros(8, 8, 3, 3, device="cuda"), 0)
    assert ops.conv3x3_supported(torch.zeros(1, 8, 8, 36, device="cuda"), torch.zeros(8, 8, 3, 3, device="cuda"), 0)
    # an odd width in front of a PixelUnshuffle(2) is not a shape of the network (the reference fails there too)
    assert not ops.conv3x3_supported(torch.zeros(1, 8, 8, 35, device="cuda"), torch.zeros(8, 8, 3, 3, device="cuda"), 1)
    with pytest.raises(RuntimeError):
        ops.conv3x3(torch.zeros(1, 8, 8, 35, device="cuda"), torch.zeros(8, 8, 3, 3, device="cuda"), None, 1)


@pytest.mark.parametrize("shape", [(2, 48, 3, 16, 24), (1, 12, 4, 6, 10), (1, 48, 3, 8, 128)])
def test_last_layer_with_bilinear_residual_vs_oracle(shape):
    """rst_conv3x3_fewout_up2_fwd = conv3x3 + bilinear x2 (align_corners False) of the low-resolution input
    (model/Restromer.py:420-431)."""
    from oracle import restormer_oracle as orc
    from superresolution_amd import ops
    B, Cin, Cout, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, 2 * h, 2 * w, generator=g)
    wt = torch.randn(Cout, Cin, 3, 3, generator=g) / 20
    low = torch.rand(B, Cout, h, w, generator=g)
    y = ops.conv3x3_fewout_up2(x.cuda(), wt.cuda(), None, low.cuda())
    ref = orc.conv3x3(x.double(), wt.double(), None) + orc.bilinear_up2(low.double())
    assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < TOL
    tref = torch.nn.functional.interpolate(low, scale_factor=2, mode="bilinear", align_corners=False)
    assert torch.allclose(orc.bilinear_up2(low), tref, atol=1e-6)


@pytest.mark.parametrize("Ca,Cb,Cout,N", [(192, 192, 192, 4096), (96, 96, 96, 640), (40, 24, 30, 100), (64, 20, 48, 256)])
def test_conv1x1_of_a_concatenation_vs_oracle(Ca, Cb, Cout, N):
    """rst_conv1x1_cat_fwd = conv1x1(cat([a, b], 1)) (model/Restromer.py:402-403,407-408): the one-product route (Ca a multiple
    of 32, split-f16) and the two-product route."""
    from oracle import restormer_oracle as orc
    from superresolution_amd import ops
    g = torch.Generator().manual_seed(Ca + Cb + Cout)
    a, b = torch.randn(2, Ca, N // 4, 4, generator=g), torch.randn(2, Cb, N // 4, 4, generator=g)
    w = torch.randn(Cout, Ca + Cb, 1, 1, generator=g) / (Ca + Cb) ** 0.5
    bias = torch.randn(Cout, generator=g)
    y = ops.conv1x1_cat(a.cuda(), b.cuda(), w.cuda(), bias.cuda())
    ref = orc.pointwise_conv(torch.cat([a, b], 1).double(), w.double(), bias.double())
    assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < TOL
    # gradients equal those of the materialised form
    ac, bc, wc = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    r = torch.randn(y.shape, generator=g).cuda()
    (ops.conv1x1_cat(ac, bc, wc, None) * r).sum().backward()
    a2, b2, w2 = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    (ops.conv1x1(torch.cat([a2, b2], 1), w2, None) * r).sum().backward()
    assert torch.allclose(ac.grad, a2.grad, rtol=1e-4, atol=1e-5) and torch.allclose(bc.grad, b2.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(wc.grad, w2.grad, rtol=1e-4, atol=1e-4)


BWD_CASES = CASES + [(8, 96, 192, 16, 16, 2, False), (2, 48, 3, 12, 40, 0, True), (1, 48, 3, 6, 128, 0, False)]


@pytest.mark.parametrize("B,Cin,Cout,H,W,mode,bias", BWD_CASES)
def test_conv3x3_backward_vs_oracle(B, Cin, Cout, H, W, mode, bias):
    """dx, dW, db of conv3x3 (+ shuffle) through rst_conv3x3_bwd against fp64 autograd of the oracle, with an output
    gradient of the size a mean loss produces (1e-6: the range where an f16-split product would lose its accuracy)."""
    from oracle import restormer_oracle as orc
    from superresolution_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H + W + mode + 5)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    b64 = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = {0: lambda t: t, 1: orc.pixel_unshuffle2, 2: orc.pixel_shuffle2}[mode](orc.conv3x3(x64, w64, b64))
    dy = torch.randn(y.shape, generator=g) * 1e-6
    dx_ref, dw_ref, db_ref = torch.autograd.grad(y, (x64, w64, b64), dy.double())
    dx, dw, db = ops.conv3x3_backward(x.cuda(), dy.cuda(), w.cuda(), mode, True, bias)
    assert float((dx.double().cpu() - dx_ref).abs().max() / dx_ref.abs().max()) < TOL
    assert float((dw.double().cpu() - dw_ref).abs().max() / dw_ref.abs().max()) < TOL
    if bias:
        assert float((db.double().cpu() - db_ref).abs().max() / db_ref.abs().max()) < TOL
    else:
        assert db is None
