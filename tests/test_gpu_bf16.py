"""GPU parity of the bf16 storage path (rst_bf16_*) through the C ABI.

Layout conversion is bit-exact against torch's round-to-nearest-even cast.  Arithmetic results are
compared with the fp64 golden fixtures written from the reference module (tests/golden/block_*.npz) and
with the fp64 oracle.  Tolerance: bf16 keeps 8 significant bits (half an ulp = 2^-9 ~ 2e-3 relative), a block
rounds its activations six times between kernels, so a tensor may differ from the fp64 result by
BF16_TOL = 2e-2 of its largest magnitude (max|a-b| / max|b|); measured values are a few 1e-3.  The
end-to-end criterion of the bf16 path is PSNR: within 0.01 dB of the fp32 result."""
import math

import numpy as np
import pytest
import torch

from golden_util import load
from oracle import recipe

pytestmark = pytest.mark.gpu
BF16_TOL = 2e-2

BF16_BLOCK_CASES = [c for c in recipe.BLOCK_CASES if c[1] % 8 == 0 and (c[1] // c[2]) % 16 == 0]


def _maxrel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("C", [3, 8, 12, 48, 60])
@pytest.mark.parametrize("src", [torch.float32, torch.bfloat16])
def test_pack_unpack_exact(C, src):
    from superresolution_amd import bf16
    g = torch.Generator().manual_seed(C)
    x = torch.randn(2, C, 6, 10, generator=g).to(src).cuda()
    p = bf16.pack(x)
    assert p.shape == (2, C, 6, 10) and p.data.shape == (2, (C + 7) // 8, 6, 10, 8)
    ref = x.to(torch.bfloat16)
    # layout: data[b, cb, y, x, j] = channel 8 cb + j (zero beyond C)
    lay = torch.zeros(2, (C + 7) // 8 * 8, 6, 10, dtype=torch.bfloat16, device="cuda")
    lay[:, :C] = ref
    lay = lay.view(2, -1, 8, 6, 10).permute(0, 1, 3, 4, 2)
    assert torch.equal(p.data.view(torch.int16), lay.contiguous().view(torch.int16))
    for dt in (torch.float32, torch.bfloat16):
        back = bf16.unpack(p, dt)
        assert back.dtype == dt and torch.equal(back.float(), ref.float())


@pytest.mark.parametrize("shape", [(48, 144), (96, 48), (384, 192), (192, 96), (60, 48), (120, 96), (240, 192), (480, 384), (16, 24)])
@pytest.mark.parametrize("bias", [False, True])
def test_conv1x1_vs_fp64(shape, bias):
    from superresolution_amd import bf16
    cin, cout = shape
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(2, cin, 10, 14, generator=g).cuda()
    w = (torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)).cuda()
    b = torch.randn(cout, generator=g).cuda() if bias else None
    r = torch.randn(2, cout, 10, 14, generator=g).cuda()
    px, pr = bf16.pack(x), bf16.pack(r)
    y = bf16.unpack(bf16.conv1x1(px, w, b, pr), torch.float32)
    # same inputs as the kernel sees (bf16-rounded activations and weights), exact arithmetic
    xr, wr, rr = bf16.unpack(px, torch.float32).double(), w.bfloat16().double(), bf16.unpack(pr, torch.float32).double()
    ref = torch.nn.functional.conv2d(xr, wr, None if b is None else b.double()) + rr
    assert _maxrel(y, ref) < 4e-3          # one rounding of the result


@pytest.mark.parametrize("case", BF16_BLOCK_CASES, ids=[c[0] for c in BF16_BLOCK_CASES])
def test_block_vs_golden(case):
    from superresolution_amd import bf16
    from superresolution_amd.model.Restromer import TransformerBlock
    tag, dim, heads, ln_type, bias, b, h, w = case
    gold = load("block_" + tag)
    sd, x, r = recipe.block_case_tensors(case, torch.float32)
    blk = TransformerBlock(dim, heads, recipe.FFN_FACTOR, bias, ln_type)
    blk.load_state_dict(sd, strict=True)
    blk = blk.cuda()
    with torch.no_grad():
        p = bf16.pack(x.cuda())
        bf16.block_(p, heads, blk.ffn.hidden_features, blk._params())
        out = bf16.unpack(p, torch.float32)
    a = out.double().cpu().numpy().reshape(-1)
    idx = recipe.subsample_index(a.size)
    ref = gold["out/v"].astype(np.float64)
    err = np.abs(a[idx] - ref).max() / np.abs(ref).max()
    assert np.isfinite(a).all()
    assert err < BF16_TOL, f"{tag}: max rel err {err:.3e}"


@pytest.mark.parametrize("case", [BF16_BLOCK_CASES[0]], ids=lambda c: c[0])
def test_block_vs_oracle_large_mean(case):
    """The folded LayerNorm takes the variance in one pass; inputs with a channel mean several sigma from zero."""
    from oracle import restormer_oracle as orc
    from superresolution_amd import bf16
    from superresolution_amd.model.Restromer import TransformerBlock
    tag, dim, heads, ln_type, bias, b, h, w = case
    sd, x, r = recipe.block_case_tensors(case, torch.float32)
    x = x + 3.0
    blk = TransformerBlock(dim, heads, recipe.FFN_FACTOR, bias, ln_type)
    blk.load_state_dict(sd, strict=True)
    blk = blk.cuda()
    with torch.no_grad():
        p = bf16.pack(x.cuda())
        xq = bf16.unpack(p, torch.float32).double().cpu()        # the block's actual input
        bf16.block_(p, heads, blk.ffn.hidden_features, blk._params())
        out = bf16.unpack(p, torch.float32)
    sd64 = {k: v.double() for k, v in sd.items()}
    ref = orc.transformer_block(xq, sd64, "", heads)
    assert _maxrel(out, ref) < BF16_TOL


def _psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 10.0 * math.log10(1.0 / mse)


def test_model_bf16_end_to_end_psnr():
    """Whole network, dim 48: the bf16 path against the fp32 path of this library (itself pinned to the reference
    at 1e-4) on an image-like input, PSNR against a synthetic target within 0.01 dB, and the outputs close."""
    import bench
    net = bench.build_model(0).cuda().eval()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 64, 96, generator=g).cuda()
    with torch.no_grad():
        ref = net(x)
        out = net(x.bfloat16())
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out_ac = net(x)
    assert out.dtype == torch.bfloat16 and out_ac.dtype == torch.float32
    assert torch.isfinite(out.float()).all()
    # a target at a realistic distance from the output (~30 dB)
    target = (ref + 0.03 * torch.randn(ref.shape, generator=g).cuda()).clamp(0, 1)
    # fp32 image out of the bf16 kernels: against the fp32 path.  bf16 image out: against the fp32 result held in the
    # same container -- rounding an image in [0, 1] to bf16 is an error of 1.4e-3 rms by itself, 0.0095 dB at this
    # distance from the target, whichever kernels produced the image
    assert abs(_psnr(out_ac, target) - _psnr(ref, target)) < 0.01
    assert abs(_psnr(out.float(), target) - _psnr(ref.bfloat16().float(), target)) < 0.01
    assert _maxrel(out_ac, ref) < BF16_TOL


def test_bf16_calls_outside_the_packed_path_run_the_fp32_kernels():
    """The packed bf16 kernels serve inference at widths that are multiples of 8 with 16-multiple heads.  Any other
    bf16 / autocast call -- the reference's default Restormer() (dim 12) under autocast, or a call that wants
    gradients -- must still work, as it does with the reference module (ordinary torch ops under autocast): it
    runs the fp32 kernels and returns the input's dtype."""
    from superresolution_amd import bf16
    from superresolution_amd.model.Restromer import Restormer
    torch.manual_seed(2)
    net = Restormer().cuda().eval()                                   # reference defaults: 4 channels, dim 12
    x = torch.rand(1, 4, 16, 24, device="cuda")
    with torch.no_grad():
        ref = net(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got = net(x)
        got_bf = net(x.bfloat16())
    assert got.dtype == torch.float32 and torch.equal(got, ref)
    assert got_bf.dtype == torch.bfloat16 and _maxrel(got_bf.float(), ref) < BF16_TOL
    # gradients wanted: fp32 kernels with autograd, also under autocast
    net.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(x)
    out.float().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    with pytest.raises(RuntimeError):
        bf16.pack(torch.rand(1, 8, 4, 4))                             # CPU tensor


def test_prepared_weights_cache_follows_parameter_updates():
    """Prepared weights kept between calls give the same bits as preparing on every call, and are rebuilt
    after an in-place parameter update."""
    from superresolution_amd import bf16
    from superresolution_amd.model.Restromer import TransformerBlock
    case = BF16_BLOCK_CASES[2]
    tag, dim, heads, ln_type, bias, b, h, w = case
    sd, x, r = recipe.block_case_tensors(case, torch.float32)
    blk = TransformerBlock(dim, heads, recipe.FFN_FACTOR, bias, ln_type)
    blk.load_state_dict(sd, strict=True)
    blk = blk.cuda()
    cache = bf16.PreparedCache()

    def run(c):
        with torch.no_grad():
            p = bf16.pack(x.cuda())
            bf16.block_(p, heads, blk.ffn.hidden_features, blk._params(), c)
            return p.data.clone()
    first = run(cache)
    assert torch.equal(first.view(torch.int16), run(None).view(torch.int16))
    assert torch.equal(first.view(torch.int16), run(cache).view(torch.int16))
    with torch.no_grad():
        blk.attn.qkv.weight.mul_(1.5)
        blk.norm2.body.weight.add_(0.25)
    changed = run(cache)
    assert not torch.equal(changed.view(torch.int16), first.view(torch.int16))
    assert torch.equal(changed.view(torch.int16), run(None).view(torch.int16))


@pytest.mark.parametrize("switch", [{"RST_BF16_FUSE": "0"}, {"RST_BF16_RING": "0"}, {"RST_DW_MFMA": "3", "RST_BF16_FUSE": "0"},
                                    {"RST_BF16_RING_ROWS": "8"}, {"RST_DW_MFMA": "0", "RST_BF16_FUSE": "0"}],
                         ids=["unfused", "fused_in_registers", "depthwise_on_matrix_pipe", "ring_with_short_row_ranges",
                              "depthwise_on_vector_unit"])
def test_block_with_kernel_variants_forced(switch):
    """By default the 1x1 + depthwise pairs of the 48- and 96-channel levels run fused, the intermediate in wave-private
    LDS row rings and both stages on the matrix pipe (pwdw_ring_kernel).  RST_BF16_FUSE=0 switches the fusion off (1x1
    kernel, then depthwise kernel); RST_BF16_RING=0 takes the fused form that keeps the intermediate in registers and runs
    the taps on the vector unit; RST_BF16_RING_ROWS=8 cuts the rows into ranges of 8 instead of 32 (more range borders);
    RST_DW_MFMA=3 runs both unfused depthwise convolutions on the matrix pipe (dwm_pk_kernel, by default only the gated one
    of small images), =0 none.  The switches are read once per process, so the golden block cases re-run in a child."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, **switch)
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_bf16.py"), "-q", "-x", "-m", "gpu",
                        "-k", "test_block_vs_golden or test_block_vs_oracle_large_mean", "-p", "no:cacheprovider"],
                       env=env, cwd=os.path.dirname(here), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]


def test_tiled_inference_bf16_close_to_fp32():
    """Tile-sharded inference takes the bf16 path tile by tile when the image is bf16: same tiling, outputs within
    the bf16 tolerance of the fp32 tiles and the same PSNR against a target to 0.01 dB as the fp32 result rounded to bf16."""
    import bench
    from superresolution_amd.parallel import tiled_inference
    net = bench.build_model(0).cuda().eval()
    g = torch.Generator().manual_seed(9)
    img = torch.rand(1, 3, 96, 128, generator=g).cuda()
    with torch.no_grad():
        ref = tiled_inference(net, img, tile=64, overlap=16, tile_batch=2)
        out = tiled_inference(net, img.bfloat16(), tile=64, overlap=16, tile_batch=2)
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape
    assert _maxrel(out.float(), ref) < BF16_TOL
    target = (ref + 0.03 * torch.randn(ref.shape, generator=g).cuda()).clamp(0, 1)
    # (the bf16 image against the fp32 result in the same container: see test_model_bf16_end_to_end_psnr)
    assert abs(_psnr(out.float(), target) - _psnr(ref.bfloat16().float(), target)) < 0.01


@pytest.mark.parametrize("shape", [(2, 48, 6, 10), (1, 96, 8, 4), (1, 8, 2, 2)])
def test_channels_last_round_trip_exact(shape):
    from superresolution_amd import bf16
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).bfloat16().cuda()
    p = bf16.pack(x)
    cl = bf16.to_channels_last(p)
    assert cl.shape == x.shape and cl.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(cl.float(), x.float())
    back = bf16.from_channels_last(cl, 0)
    assert torch.equal(back.data.view(torch.int16), p.data.view(torch.int16))


@pytest.mark.parametrize("shape", [(2, 24, 6, 12), (1, 96, 8, 4), (1, 2, 4, 4)])
def test_pixel_unshuffle_into_packed_exact(shape):
    """rst_bf16_from_channels_last mode 1 == nn.PixelUnshuffle(2) (reference model/Restromer.py:196) then pack."""
    from superresolution_amd import bf16
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(*shape, generator=g).bfloat16().cuda()
    got = bf16.from_channels_last(x.contiguous(memory_format=torch.channels_last), 1)
    ref = torch.nn.functional.pixel_unshuffle(x, 2)
    assert got.shape == tuple(ref.shape)
    assert torch.equal(bf16.unpack(got).float(), ref.float())


@pytest.mark.parametrize("shape", [(2, 32, 6, 10), (1, 192, 4, 8), (1, 768, 2, 2)])
def test_pixel_shuffle_into_packed_exact(shape):
    """rst_bf16_from_channels_last mode 2 == nn.PixelShuffle(2) (reference model/Restromer.py:211) then pack."""
    from superresolution_amd import bf16
    g = torch.Generator().manual_seed(sum(shape) + 2)
    x = torch.randn(*shape, generator=g).bfloat16().cuda()
    got = bf16.from_channels_last(x.contiguous(memory_format=torch.channels_last), 2)
    ref = torch.nn.functional.pixel_shuffle(x, 2)
    assert got.shape == tuple(ref.shape)
    assert torch.equal(bf16.unpack(got).float(), ref.float())
