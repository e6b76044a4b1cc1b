"""GPU parity of MDTA, GDFN and the whole TransformerBlock against the golden
fixtures written from the reference module (tests/golden/block_*.npz), forward
and backward, through the C ABI.  Contract tolerance: 1e-4 max relative error
(max|a-b| / max|b| per tensor) in fp32."""
import pytest
import torch

from golden_util import check_pin, load
from oracle import recipe

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _block(case):
    from superresolution_amd.model.Restromer import TransformerBlock
    tag, dim, heads, ln_type, bias, b, h, w = case
    sd, x, r = recipe.block_case_tensors(case, torch.float32)
    blk = TransformerBlock(dim, heads, recipe.FFN_FACTOR, bias, ln_type)
    blk.load_state_dict(sd, strict=True)
    return blk.cuda(), x.cuda(), r.cuda()


@pytest.mark.parametrize("case", recipe.BLOCK_CASES, ids=[c[0] for c in recipe.BLOCK_CASES])
def test_block_forward_backward(case):
    tag = case[0]
    g = load("block_" + tag)
    blk, x, r = _block(case)
    x.requires_grad_(True)
    out = blk(x)
    (out * r).sum().backward()
    check_pin(g, "out", out, TOL, tag)
    check_pin(g, "dx", x.grad, TOL, tag)
    for k, p in blk.named_parameters():
        check_pin(g, "d/" + k, p.grad, TOL, tag)


@pytest.mark.parametrize("case", recipe.BLOCK_CASES, ids=[c[0] for c in recipe.BLOCK_CASES])
def test_submodules_forward(case):
    tag = case[0]
    g = load("block_" + tag)
    blk, x, r = _block(case)
    with torch.no_grad():
        check_pin(g, "norm1", blk.norm1(x), TOL, tag)
        check_pin(g, "attn", blk.attn(x), TOL, tag)
        check_pin(g, "ffn", blk.ffn(x), TOL, tag)
        check_pin(g, "out", blk(x), TOL, tag)          # inference path (no saved buffer)


@pytest.mark.parametrize("case", recipe.BLOCK_CASES[:5], ids=[c[0] for c in recipe.BLOCK_CASES[:5]])
def test_submodules_backward_vs_oracle(case):
    """Attention and FeedForward used on their own (autograd through
    rst_attention_bwd / rst_ffn_bwd) against autograd of the fp64 oracle."""
    from oracle import restormer_oracle as orc
    tag, dim, heads, ln_type, bias, b, h, w = case
    blk, x, r = _block(case)
    sd64, x64, r64 = recipe.block_case_tensors(case, torch.float64)
    for name, fn in (("attn", lambda t, s: orc.mdta(t, s, "attn.", heads)), ("ffn", lambda t, s: orc.gdfn(t, s, "ffn."))):
        s = {k: v.clone().requires_grad_(True) for k, v in sd64.items()}
        xo = x64.clone().requires_grad_(True)
        (fn(xo, s) * r64).sum().backward()
        blk.zero_grad()
        xg = x.detach().clone().requires_grad_(True)
        (getattr(blk, name)(xg) * r).sum().backward()
        scale = xo.grad.abs().max()
        assert float((xg.grad.double().cpu() - xo.grad).abs().max() / scale) < TOL, (tag, name, "dx")
        for k, p in getattr(blk, name).named_parameters():
            ref = s[name + "." + k].grad
            err = float((p.grad.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
            assert err < TOL, (tag, name, k, err)


@pytest.mark.parametrize("case", [recipe.BLOCK_CASES[0], recipe.BLOCK_CASES[3], recipe.BLOCK_CASES[6]],
                         ids=lambda c: c[0])
@pytest.mark.parametrize("offset", [0.0, 6.0])
def test_inference_folded_layernorm_vs_oracle(case, offset):
    """Inference folds both LayerNorms into the following products and takes mean and variance
    in one pass; a channel mean far from zero (offset / sigma ~ 12) is where a one-pass variance
    loses digits, so the block is checked there against the fp64 oracle at the same tolerance."""
    from oracle import restormer_oracle as orc
    tag, dim, heads, ln_type, bias, b, h, w = case
    blk, x, _ = _block(case)
    sd64, x64, _ = recipe.block_case_tensors(case, torch.float64)
    x = x * 0.5 + offset
    x64 = x64 * 0.5 + offset
    with torch.no_grad():
        got = blk(x)
    ref = orc.transformer_block(x64, sd64, "", heads)
    # the residual keeps the offset in the output: measure the error against the size of the update
    upd = (ref - x64).abs().max()
    err = float((got.double().cpu() - ref).abs().max() / upd)
    assert err < TOL, (tag, offset, err)


def test_unfused_pair_still_matches():
    """Inference runs the fused 1x1 + depthwise kernel wherever it is supported (48 and 96 input
    channels, W >= 64).  RST_FUSE_PWDW=0 switches it off so that the 1x1 product (with folded
    LayerNorm) followed by the depthwise kernels runs on the same shapes.  The switch is read once per
    process, so the wide-row golden cases and the folded-LayerNorm oracle check are re-run in a child."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, RST_FUSE_PWDW="0")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_block.py"), "-q", "-x", "-m", "gpu",
                        "-k", "(w136 or w72 or w128 or w64 or folded) and not unfused", "-p", "no:cacheprovider"],
                       env=env, cwd=os.path.dirname(here), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("switch", [{"RST_F32_MFMA": "1"}, {"RST_PWDW_FORM": "f"}, {"RST_PWDW_FORM": "w"}, {"RST_F16S_MINK": "48"},
                                    {"RST_F16S_MINK": "192"}, {"RST_GF_ONE_BLOCK": "1"}],
                         ids=["exact_fp32_mfma", "single_role_split_f16", "wave_specialised", "split_f16_gemm_everywhere",
                              "split_f16_gemm_from_192", "split_f16_gemm_one_block_per_cu"])
def test_product_routes_forced(switch):
    """The MFMA-bound products of the fp32 path run on the f16 matrix instructions with split operands by default.
    RST_F32_MFMA=1 keeps them on the exact-fp32 instruction (the route for activations beyond the f16 range);
    RST_PWDW_FORM=f / w forces the single-role / wave-specialised fused kernel wherever it supports the shape;
    RST_F16S_MINK=48 sends every 1x1 product from 48 input channels through the split-f16 GEMM, 192 keeps the 96-channel
    ones on the exact-fp32 kernel (the default bound is 96); RST_GF_ONE_BLOCK=1 runs the split-f16 GEMM as one block of
    up to 256 rows per compute unit with a 3-deep ring instead of two smaller ones.  Each route must pass the
    golden block cases (switches are read once per process: child run)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, **switch)
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_block.py"), "-q", "-x", "-m", "gpu",
                        "-k", "not unfused and not routes_forced", "-p", "no:cacheprovider"],
                       env=env, cwd=os.path.dirname(here), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("case", [("one_sample_c48", 48, 1, "WithBias", False, 1, 96, 96),
                                  ("one_sample_c96h2", 96, 2, "BiasFree", True, 1, 80, 128)], ids=lambda c: c[0])
def test_single_large_sample_many_pixel_splits(case):
    """One sample with many pixels (a full frame is the extreme): the Gram reduction is then split into more
    pixel ranges than the softmax kernel adds by itself and goes through a slab reduction first.  Forward
    (inference and training path) and input gradient against the fp64 oracle."""
    from oracle import restormer_oracle as orc
    tag, dim, heads, ln_type, bias, b, h, w = case
    blk, x, r = _block(case)
    sd64, x64, r64 = recipe.block_case_tensors(case, torch.float64)
    with torch.no_grad():
        got = blk(x)
    x64 = x64.clone().requires_grad_(True)
    ref = orc.transformer_block(x64, sd64, "", heads)
    (ref * r64).sum().backward()
    scale = ref.detach().abs().max()
    assert float((got.double().cpu() - ref.detach()).abs().max() / scale) < TOL, tag
    xg = x.detach().clone().requires_grad_(True)
    out = blk(xg)
    (out * r).sum().backward()
    assert float((out.detach().double().cpu() - ref.detach()).abs().max() / scale) < TOL, tag
    assert float((xg.grad.double().cpu() - x64.grad).abs().max() / x64.grad.abs().max()) < TOL, tag


@pytest.mark.parametrize("dim,heads,hw", [(96, 2, (12, 40)), (384, 8, (4, 8))], ids=["c96_hidden255", "c384_hidden1021"])
def test_odd_hidden_width_vs_oracle(dim, heads, hw):
    """The network's own expansion factor 2.66 gives odd hidden widths (255 at 96 channels, 1021 at 384): project_out then
    contracts over an odd number of channels, which the split-f16 product covers by re-reading the last channel row
    against zero weights.  Inference, training forward and input gradient against the fp64 oracle."""
    from oracle import restormer_oracle as orc
    from superresolution_amd.model.Restromer import TransformerBlock
    shapes = recipe.block_param_shapes(dim, heads, 2.66, True, "WithBias")
    sd = recipe.make_state_dict(shapes, 4242 + dim, torch.float32)
    sd64 = recipe.make_state_dict(shapes, 4242 + dim, torch.float64)
    blk = TransformerBlock(dim, heads, 2.66, True, "WithBias")
    blk.load_state_dict(sd, strict=True)
    blk = blk.cuda()
    assert blk.ffn.hidden_features % 2 == 1
    x64 = recipe.make_input((2, dim, *hw), 77, "normal", torch.float64)
    r64 = recipe.make_input((2, dim, *hw), 78, "normal", torch.float64)
    x, r = x64.float().cuda(), r64.float().cuda()
    with torch.no_grad():
        got = blk(x)
    xo = x64.clone().requires_grad_(True)
    ref = orc.transformer_block(xo, sd64, "", heads)
    (ref * r64).sum().backward()
    scale = ref.detach().abs().max()
    assert float((got.double().cpu() - ref.detach()).abs().max() / scale) < TOL
    xg = x.clone().requires_grad_(True)
    out = blk(xg)
    (out * r).sum().backward()
    assert float((out.detach().double().cpu() - ref.detach()).abs().max() / scale) < TOL
    assert float((xg.grad.double().cpu() - xo.grad).abs().max() / xo.grad.abs().max()) < TOL


def test_prepared_inference_equals_unprepared_and_follows_parameter_writes():
    """rst_block_fwd_prepared gives bit for bit what rst_block_fwd gives (same kernels, the weight folds done ahead), at a
    width on the fused kernels and at one on the split-f16 products; an in-place parameter write rebuilds the prepared
    weights, `invalidate()` covers writes through `.data`."""
    from superresolution_amd import ops
    from superresolution_amd.model.Restromer import TransformerBlock
    for dim, heads, hw in ((96, 2, (16, 128)), (192, 4, (16, 32))):
        torch.manual_seed(dim)
        blk = TransformerBlock(dim, heads, 1.25, True, "WithBias").cuda()
        x = torch.randn(2, dim, *hw, device="cuda")
        with torch.no_grad():
            plain = ops.transformer_block(x, heads, blk.ffn.hidden_features, blk._params())          # no cache: rst_block_fwd
            cached = blk(x)
            assert torch.equal(plain, cached)
            assert blk._f32_cache.blob is not None
            blob = blk._f32_cache.blob
            assert torch.equal(blk(x), cached) and blk._f32_cache.blob is blob                         # reused
            blk.norm1.body.weight.mul_(1.5)                                                          # version moves
            after = blk(x)
            assert blk._f32_cache.blob is not blob
            assert torch.equal(after, ops.transformer_block(x, heads, blk.ffn.hidden_features, blk._params()))
            assert not torch.equal(after, cached)
            blk.ffn.project_in.weight.data.mul_(0.5)                                                 # bypasses the version counter
            blk._f32_cache.invalidate()
            assert torch.equal(blk(x), ops.transformer_block(x, heads, blk.ffn.hidden_features, blk._params()))
