"""Dense 3x3 convolution on packed bf16 tensors with the pixel (un)shuffle epilogue (rst_bf16_conv3x3_fwd) against the
oracle's conv3x3 + pixel_unshuffle2 / pixel_shuffle2 in fp64, evaluated on the bf16-rounded input and the bf16-rounded
weights the kernel multiplies (so what is left is the fp32 summation order and the rounding of the result to bf16:
2^-9 relative).  Every tile geometry (widths 16 ... 480, partial tiles at the right and bottom edges), channel counts
that pad, the network's own shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, w, b, mode):
    from oracle import restormer_oracle as orc
    y = orc.conv3x3(x.double(), w.double(), None if b is None else b.double())
    return {0: lambda t: t, 1: orc.pixel_unshuffle2, 2: orc.pixel_shuffle2}[mode](y)


CASES = [  # B, Cin, Cout, H, W, mode, bias
    (2, 3, 48, 8, 128, 0, False), (1, 3, 48, 5, 256, 0, True), (2, 48, 24, 6, 128, 1, False), (1, 96, 48, 4, 64, 1, False),
    (1, 192, 96, 8, 32, 1, False), (1, 384, 768, 4, 32, 2, False), (1, 192, 384, 4, 64, 2, False), (2, 96, 192, 2, 128, 2, False),
    (1, 96, 192, 3, 256, 2, False), (1, 48, 40, 5, 384, 0, True), (1, 33, 32, 4, 32, 2, True), (1, 70, 136, 2, 64, 0, False),
    (1, 96, 192, 7, 480, 2, False),          # 3.75 tiles per row, odd height: what a 2160 x 3840 frame has at its deep levels
    (1, 48, 24, 6, 480, 1, True), (1, 192, 96, 10, 16, 1, False), (1, 96, 48, 6, 24, 1, False), (1, 8, 16, 9, 40, 0, True),
    (1, 384, 768, 3, 20, 2, False), (2, 384, 768, 4, 4, 2, False), (1, 192, 96, 8, 8, 1, False), (1, 3, 48, 8, 8, 0, True),
]


@pytest.mark.parametrize("B,Cin,Cout,H,W,mode,bias", CASES)
def test_conv3x3_bf16_vs_oracle(B, Cin, Cout, H, W, mode, bias):
    from superresolution_amd import bf16
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H + W + mode)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16()
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g) if bias else None
    assert bf16.conv3x3_supported(Cin, Cout, H, W, mode)
    y = bf16.conv3x3(bf16.pack(x.cuda()), w.cuda(), None if b is None else b.cuda(), mode)
    got = bf16.unpack(y, torch.float32).double().cpu()
    ref = _ref(x.float(), w.bfloat16().float(), b, mode)
    assert tuple(got.shape) == tuple(ref.shape)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 6e-3, err                       # measured <= 3e-3: half a bf16 unit of the largest outputs


def test_conv3x3_bf16_borders():
    """Image-like input and a weight of ones: every border pixel must see exactly its zero padding (a wrong halo or a
    chunk read past the image shows as an O(1) error)."""
    from superresolution_amd import bf16
    x = (torch.rand(1, 48, 5, 200, generator=torch.Generator().manual_seed(1)) + 3.0).bfloat16()
    w = torch.ones(24, 48, 3, 3) / 432.0
    y = bf16.unpack(bf16.conv3x3(bf16.pack(x.cuda()), w.cuda(), None, 0), torch.float32).double().cpu()
    ref = _ref(x.float(), w.bfloat16().float(), None, 0)
    assert float((y - ref).abs().max() / ref.abs().max()) < 6e-3


def test_conv3x3_bf16_rejects_what_it_cannot_pack():
    from superresolution_amd import bf16
    assert not bf16.conv3x3_supported(48, 20, 8, 64, 2)         # PixelShuffle needs whole blocks of 8 output channels
    assert not bf16.conv3x3_supported(48, 24, 7, 64, 1)         # PixelUnshuffle of an odd height
    x = bf16.pack(torch.randn(1, 48, 8, 64, device="cuda"))
    with pytest.raises(RuntimeError):
        bf16.conv3x3(x, torch.randn(20, 48, 3, 3, device="cuda"), None, 2)


@pytest.mark.parametrize("B,Cin,Cout,H,W,form", [(2, 48, 3, 12, 40, "lowres"), (1, 48, 3, 9, 26, "residual"), (1, 20, 4, 6, 8, "none"),
                                                  (1, 48, 1, 5, 130, "residual"), (2, 96, 3, 16, 64, "lowres")])
def test_conv3x3_fewout_bf16_vs_oracle(B, Cin, Cout, H, W, form):
    """The last layer on a packed input (rst_bf16_conv3x3_fewout_fwd): fp32 weights and sums on the bf16-rounded input, fp32
    result, plus a plain residual or the bilinear x2 of a low-resolution tensor -- against the oracle's conv3x3 (+ torch's
    bilinear interpolation in fp64) on the same rounded input: only the fp32 summation order is left."""
    from oracle import restormer_oracle as orc
    from superresolution_amd import bf16
    g = torch.Generator().manual_seed(B * 100 + Cin + Cout + H + W)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16()
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    ref = orc.conv3x3(x.double(), w.double(), b.double())
    res = low = None
    if form == "residual":
        res = torch.randn(B, Cout, H, W, generator=g)
        ref = ref + res.double()
    elif form == "lowres":
        low = torch.rand(B, Cout, H // 2, W // 2, generator=g)
        ref = ref + torch.nn.functional.interpolate(low.double(), scale_factor=2, mode="bilinear", align_corners=False)
    y = bf16.conv3x3_fewout(bf16.pack(x.cuda()), w.cuda(), b.cuda(), None if res is None else res.cuda(), None if low is None else low.cuda())
    assert y.dtype == torch.float32 and tuple(y.shape) == (B, Cout, H, W)
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err
