"""The CPU oracle against the fixtures produced by the reference module.

The fixtures (tests/golden/*.npz) were written by oracle/gen_golden.py from the
reference evaluated in fp64; here the oracle is re-run without the reference.
fp64 oracle vs fp64 reference: 1e-6 (the pins are stored as fp32);
fp32 oracle vs fp64 reference: 2e-5."""
import numpy as np
import pytest
import torch

from oracle import recipe
from oracle import restormer_oracle as orc
from golden_util import check_pin, load


@pytest.mark.parametrize("case", recipe.BLOCK_CASES, ids=[c[0] for c in recipe.BLOCK_CASES])
def test_block_oracle_fp64(case):
    tag, dim, heads, ln_type, bias, b, h, w = case
    g = load("block_" + tag)
    sd, x, r = recipe.block_case_tensors(case, torch.float64)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = x.clone().requires_grad_(True)
    out = orc.transformer_block(x, sd, "", heads)
    (out * r).sum().backward()
    check_pin(g, "out", out, 1e-6, tag)
    check_pin(g, "dx", x.grad, 1e-6, tag)
    for k, v in sd.items():
        check_pin(g, "d/" + k, v.grad, 1e-6, tag)
    with torch.no_grad():
        xd = x.detach()
        sdd = {k: v.detach() for k, v in sd.items()}
        check_pin(g, "norm1", orc.layernorm_nchw(xd, sdd["norm1.body.weight"], sdd.get("norm1.body.bias")), 1e-6, tag)
        check_pin(g, "attn", orc.mdta(xd, sdd, "attn.", heads), 1e-6, tag)
        check_pin(g, "ffn", orc.gdfn(xd, sdd, "ffn."), 1e-6, tag)


@pytest.mark.parametrize("case", recipe.BLOCK_CASES[:4], ids=[c[0] for c in recipe.BLOCK_CASES[:4]])
def test_block_oracle_fp32(case):
    tag, dim, heads, *_ = case
    g = load("block_" + tag)
    sd, x, r = recipe.block_case_tensors(case, torch.float32)
    with torch.no_grad():
        out = orc.transformer_block(x, sd, "", heads)
    check_pin(g, "out", out, 2e-5, tag)
    check_pin(g, "out32", out, 2e-5, tag)


@pytest.mark.parametrize("case", recipe.MODEL_CASES, ids=[c[0] for c in recipe.MODEL_CASES])
def test_model_oracle(case):
    tag, kwargs, shape = case
    g = load("model_" + tag)
    sd, x, tgt = recipe.model_case_tensors(case, torch.float64)
    with torch.no_grad():
        out = orc.restormer_forward(x, sd, kwargs.get("heads", (1, 2, 4, 8)))
    check_pin(g, "out", out, 1e-6, tag)
    assert abs(orc.l1_loss(out, tgt).item() - float(g["loss"])) < 1e-9
    assert abs(orc.psnr(out, tgt) - float(g["psnr"])) < 1e-6


def test_state_dict_contract():
    """Names, shapes and order of the dim=48 network: 495 tensors, 18,433,300 parameters."""
    shapes = orc.model_param_shapes(inp_channels=3, out_channels=3, dim=48)
    assert len(shapes) == 495
    assert sum(int(np.prod(s)) for _, s in shapes) == 18433300
    assert shapes[0] == ("patch_embed.proj.weight", (48, 3, 3, 3))
    assert shapes[-1] == ("output.weight", (3, 48, 3, 3))


def test_train_steps_oracle():
    g = load("train_dim12_default")
    case = recipe.MODEL_CASES[0]
    sd, x, tgt = recipe.model_case_tensors(case, torch.float64)
    losses, new_sd = orc.train_steps(x, tgt, sd, (1, 2, 4, 8), 3)
    assert np.allclose(losses, g["losses"], rtol=1e-9)
    for k in list(sd)[:40]:
        d = torch.sqrt(((new_sd[k] - sd[k]) ** 2).sum()).item()
        assert abs(d - float(g["delta_l2/" + k])) <= 1e-6 * max(float(g["delta_l2/" + k]), 1e-12), k


def test_elementary_maps():
    torch.manual_seed(1)
    x = torch.randn(2, 8, 6, 10, dtype=torch.float64)
    assert torch.equal(orc.pixel_shuffle2(orc.pixel_unshuffle2(x)), x)
    assert torch.equal(orc.pixel_unshuffle2(x), torch.nn.functional.pixel_unshuffle(x, 2))
    assert torch.equal(orc.pixel_shuffle2(x), torch.nn.functional.pixel_shuffle(x, 2))
    w = torch.randn(8, 1, 3, 3, dtype=torch.float64)
    b = torch.randn(8, dtype=torch.float64)
    assert torch.allclose(orc.depthwise3x3(x, w, b), orc.depthwise3x3_explicit(x, w, b), atol=1e-13)
    up = torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    assert torch.allclose(orc.bilinear_up2(x), up, atol=1e-13)
    assert torch.allclose(orc.gelu_erf(x), torch.nn.functional.gelu(x), atol=1e-14)
    assert orc.psnr(x, x) == float("inf")


def test_oracle_gradients_match_every_tensor_of_the_allgrads_fixture():
    """The oracle's fp64 autograd against `model_dim48_rgb_allgrads` (every parameter tensor of the dim-48 network,
    written by the reference module): the fixture the bf16 training test leans on."""
    import numpy as np
    case = recipe.MODEL_CASES[1]
    g = load("model_" + case[0] + "_allgrads")
    sd, x, tgt = recipe.model_case_tensors(case, torch.float64)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss = (orc.restormer_forward(x, sd) - tgt).abs().mean()
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-12
    for k, v in sd.items():
        a = v.grad.numpy().reshape(-1)[recipe.subsample_index(v.numel(), count=1024)]
        b = g["g/" + k].astype(np.float64)
        assert np.abs(a - b).max() <= 2e-7 * max(np.abs(b).max(), 1e-30) + 1e-12, k         # the fixture is stored as fp32
        assert abs(float(np.sqrt((v.grad ** 2).sum())) - float(g["gl2/" + k])) <= 1e-9 * float(g["gl2/" + k]) + 1e-15, k
