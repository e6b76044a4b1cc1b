"""Whole-network parity on the GPU against the fixtures written from the reference
(tests/golden/model_*.npz, train_*.npz).  fp32, contract tolerance 1e-4 max
relative error per tensor; scalar loss 1e-5 relative."""
import numpy as np
import pytest
import torch

from golden_util import check_pin, load
from oracle import recipe

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _net(case):
    from superresolution_amd.model.Restromer import Restormer
    tag, kwargs, shape = case
    sd, x, tgt = recipe.model_case_tensors(case, torch.float32)
    net = Restormer(**kwargs)
    net.load_state_dict(sd, strict=True)
    return net.cuda(), x.cuda(), tgt.cuda()


@pytest.mark.parametrize("case", recipe.MODEL_CASES, ids=[c[0] for c in recipe.MODEL_CASES])
def test_model_forward_backward(case):
    tag = case[0]
    g = load("model_" + tag)
    net, x, tgt = _net(case)
    with torch.no_grad():
        check_pin(g, "out", net(x), TOL, tag + " (no_grad)")
    x.requires_grad_(True)
    out = net(x)
    loss = (out - tgt).abs().mean()
    loss.backward()
    check_pin(g, "out", out, TOL, tag)
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * float(g["loss"])
    check_pin(g, "dx", x.grad, 5e-4, tag)     # sign(out - tgt) makes the L1 gradient discontinuous; see note below
    worst = 0.0
    for k, p in net.named_parameters():
        ref = float(g["gl2/" + k])
        got = float(p.grad.double().norm())
        worst = max(worst, abs(got - ref) / max(ref, 1e-12))
        if ("d/" + k + "/v") in g:
            check_pin(g, "d/" + k, p.grad, 5e-4, tag)
    # L1 loss: an output element within rounding of its target can flip sign(out - tgt),
    # which perturbs every gradient by O(1/numel); 5e-4 bounds that, the forward keeps 1e-4.
    assert worst < 5e-4, worst
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters())).item()
    assert abs(gn - float(g["grad_norm"])) <= 5e-4 * float(g["grad_norm"])


def test_training_steps_follow_reference():
    """Three optimiser steps in the order of the reference's loop (zero_grad, forward,
    L1, backward, clip_grad_norm_(1.0), Adam(1e-4, wd 1e-6)) reproduce its loss curve."""
    case = recipe.MODEL_CASES[0]
    g = load("train_" + case[0])
    net, x, tgt = _net(case)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=1e-6)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = torch.nn.functional.l1_loss(net(x), tgt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=1.0)
        opt.step()
        losses.append(loss.item())
    assert np.allclose(losses, g["losses"], rtol=2e-5), (losses, g["losses"])
    with torch.no_grad():
        check_pin(g, "out_after", net(x), 2e-3, "after 3 steps")


def test_large_shape_properties():
    """Full-size properties that need no oracle: the block is a residual update that is
    deterministic run to run, and in-place inference equals out-of-place inference."""
    from superresolution_amd import ops
    from superresolution_amd.model.Restromer import TransformerBlock
    torch.manual_seed(0)
    blk = TransformerBlock(96, 1, 1.25, False, "WithBias").cuda()
    x = torch.rand(2, 96, 256, 256, device="cuda")
    with torch.no_grad():
        y1 = blk(x)
        y2 = blk(x)
        assert torch.equal(y1, y2)
        z = ops.block_forward_inplace(x.clone(), 1, blk.ffn.hidden_features, blk._params())
        assert torch.equal(z, y1)
        assert torch.isfinite(y1).all()


def test_tiled_inference_matches_oracle_on_same_tiling():
    """Tile-sharded inference (one rank here) against the oracle applied to the same tiles:
    tiled output is not the untiled output (per-tile attention statistics), so the oracle is
    run tile by tile over the identical grid."""
    from oracle import restormer_oracle as orc
    from superresolution_amd.parallel import tile_grid, tiled_inference
    case = recipe.MODEL_CASES[0]
    net, _, _ = _net(case)
    sd, _, _ = recipe.model_case_tensors(case, torch.float64)
    img = recipe.make_input((1, 4, 48, 64), 77, "uniform", torch.float32)
    out = tiled_inference(net.eval(), img.cuda(), tile=32, overlap=8, tile_batch=2)
    ref = torch.zeros(1, 4, 96, 128, dtype=torch.float64)
    with torch.no_grad():
        for (y0, y1, x0, x1, cy0, cy1, cx0, cx1) in tile_grid(48, 64, 32, 8):
            o = orc.restormer_forward(img[:, :, y0:y1, x0:x1].double(), sd)
            ref[:, :, 2 * cy0:2 * cy1, 2 * cx0:2 * cx1] = o[:, :, 2 * (cy0 - y0):2 * (cy1 - y0), 2 * (cx0 - x0):2 * (cx1 - x0)]
    err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < TOL, err


def test_data_parallel_trainer_single_gpu():
    """world 1 on the GPU: the trainer's bucketed-gradient step equals the plain loop."""
    from superresolution_amd.parallel import DataParallelTrainer
    case = recipe.MODEL_CASES[0]
    g = load("train_" + case[0])
    net, x, tgt = _net(case)
    tr = DataParallelTrainer(net, world_size=1)
    losses = [tr.step(x, tgt).item() for _ in range(3)]
    assert np.allclose(losses, g["losses"], rtol=2e-5), (losses, g["losses"])


def test_activations_beyond_the_f16_range_never_return_inf():
    """The fp32 products run on the f16 matrix instructions with split operands: |activation| must stay below 65504.  An
    image scaled by 1e5 leaves that range in the first block.  With the range check on, the module notices the
    non-finite result, switches itself to the exact-fp32 instructions and returns what the fp64 oracle computes; the C
    switch (rst_set_exact_fp32) does the same for direct callers of the block entry points."""
    import warnings
    from oracle import recipe
    from oracle import restormer_oracle as orc
    from superresolution_amd import _lib
    from superresolution_amd.model.Restromer import Restormer
    case = recipe.MODEL_CASES[1]
    sd, x, _ = recipe.model_case_tensors(case, torch.float32)
    x = x * 1e5
    net = Restormer(**case[1]); net.load_state_dict(sd, strict=True); net = net.cuda().eval()
    with torch.no_grad():
        ref = orc.restormer_forward(x.double(), {k: v.double() for k, v in sd.items()})
        net.range_check = False
        raw = net(x.cuda())
        assert not bool(torch.isfinite(raw).all())                 # what the guard is for
        net.range_check = True
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = net(x.cuda())
        assert any("f16 range" in str(m.message) for m in w) and net.exact_fp32
        assert bool(torch.isfinite(got).all())
        assert float((got.double().cpu() - ref).abs().max() / ref.abs().max()) < 1e-4
        assert _lib.load().rst_get_exact_fp32() == 0               # the switch is back for everyone else
        again = net(x.cuda())                                      # stays on the exact instructions, silently
        # (not bit for bit: in this mode the dense 3x3 convolutions are torch's, whose algorithm choice may differ per call)
        assert float((again - got).abs().max() / got.abs().max()) < 1e-5
        # the C switch for callers of the block entry points
        blk = net.encoder_level1[0]
        h = net.patch_embed(x.cuda())
        assert float(h.abs().max()) > 65504
        lib = _lib.load()
        lib.rst_set_exact_fp32(1)
        try:
            assert lib.rst_get_exact_fp32() == 1
            y = blk(h)
        finally:
            lib.rst_set_exact_fp32(0)
        assert bool(torch.isfinite(y).all())


def test_input_contract_of_the_module():
    """What the reference's callers may hand the network (model/Restromer.py:385-433): non-contiguous views are
    accepted and give the result of their contiguous copy, bit for bit; a shape the network cannot take (not 4-D, H or W not a
    multiple of 8 -- the reference fails inside torch.cat there) is refused with a message, before any kernel runs;
    an empty batch is refused too."""
    from superresolution_amd.model.Restromer import Restormer
    torch.manual_seed(3)
    net = Restormer(inp_channels=3, out_channels=3, dim=48).cuda().eval()
    big = torch.rand(2, 3, 40, 48, device="cuda")
    view = big[:, :, 4:36, 8:40]                                   # 32 x 32 window, strides of the larger tensor
    assert not view.is_contiguous()
    with torch.no_grad():
        a = net(view)
        b = net(view.contiguous())
    assert torch.equal(a, b)
    window = big[:, :, :32, :32]
    chan_last = window.contiguous(memory_format=torch.channels_last)
    assert not chan_last.is_contiguous()
    with torch.no_grad():
        assert torch.equal(net(chan_last), net(window.contiguous()))
    for bad in (torch.rand(3, 32, 32, device="cuda"), torch.rand(1, 3, 30, 32, device="cuda"), torch.rand(1, 3, 32, 36, device="cuda")):
        with pytest.raises(RuntimeError):
            net(bad)
    with pytest.raises(RuntimeError):
        net(torch.rand(0, 3, 32, 32, device="cuda"))


@pytest.mark.parametrize("hw", [(40, 40), (16, 104), (8, 8)])
def test_image_sizes_with_odd_deep_levels(hw, monkeypatch):
    """Any H, W divisible by 8 is an input of the reference (model/Restromer.py:385-433); the W / 8 level is then 5, 13
    or 1 pixels wide.  fp32 forward against the fp64 oracle, with torch's own convolution made unavailable (every
    convolution of the call is the library's); a bf16 input of such a size is served by the fp32 kernels (the packed
    ones take two pixels of a row per lane) and comes back in bf16; the backward runs and gives finite gradients equal
    to those of a second call."""
    from oracle import restormer_oracle as orc
    case = recipe.MODEL_CASES[1]
    net, _, _ = _net(case)
    sd64, _, _ = recipe.model_case_tensors(case, torch.float64)
    img = recipe.make_input((1, case[2][1]) + hw, 5, "uniform", torch.float32)
    with torch.no_grad():
        ref = orc.restormer_forward(img.double(), sd64)

    def no_torch_conv(*a, **k):
        raise AssertionError("torch's convolution was called")
    with monkeypatch.context() as m:
        m.setattr(torch.nn.functional, "conv2d", no_torch_conv)
        with torch.no_grad():
            out = net.eval()(img.cuda())
            out_bf = net(img.cuda().bfloat16())
        net.train()
        grads = []
        for _ in range(2):
            net.zero_grad()
            net(img.cuda()).abs().mean().backward()
            grads.append([p.grad.clone() for p in net.parameters()])
    assert float((out.double().cpu() - ref).abs().max() / ref.abs().max()) < TOL
    assert out_bf.dtype == torch.bfloat16
    assert float((out_bf.double().cpu() - ref).abs().max() / ref.abs().max()) < 2e-2
    for g0, g1 in zip(*grads):
        assert torch.isfinite(g0).all() and torch.equal(g0, g1)


@pytest.mark.parametrize("shape", [(4, 3, 128, 128), (2, 3, 200, 136)])
def test_fp32_inference_is_bit_identical_run_to_run(shape):
    """No kernel of the fp32 forward pass uses atomics or an order that depends on timing, and no call reaches a
    vendor convolution: twelve calls on the same input give the same bits (a race in one of the LDS rings, whose stages
    are reused behind a single barrier, would show here as a sporadic difference).  200 x 136 goes through the
    zero-padded route of the 3x3 convolutions (levels 136 / 68 / 34 / 17)."""
    from superresolution_amd.model.Restromer import Restormer
    torch.manual_seed(11)
    net = Restormer(inp_channels=3, out_channels=3, dim=48).cuda().eval()
    x = torch.rand(*shape, device="cuda")
    with torch.no_grad():
        ref = net(x).clone()
        assert torch.isfinite(ref).all()
        for _ in range(12):
            assert torch.equal(net(x), ref)
