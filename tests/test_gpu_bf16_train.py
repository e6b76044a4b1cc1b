"""Training with bf16 activations (the precision of the bf16 training configuration: bf16 storage, fp32 sums, fp32 master parameters):
gradients against the fp64 oracle / the reference-generated fixture, and a short optimiser trajectory against the
oracle's.  Tolerances are bf16's: a block rounds its activations to 8 significant bits six times, the network has
44 blocks; measured values are quoted next to each bound."""
import math

import numpy as np
import pytest
import torch

from golden_util import load
from oracle import recipe

pytestmark = pytest.mark.gpu


def _net(case):
    from superresolution_amd.model.Restromer import Restormer
    sd, x, tgt = recipe.model_case_tensors(case, torch.float32)
    net = Restormer(**case[1])
    net.load_state_dict(sd, strict=True)
    return net.cuda(), x.cuda(), tgt.cuda()


def _kind(name):
    return ("temperature" if name.endswith("temperature") else "ln" if ".norm" in name else
            "dw" if "dwconv" in name else "dense")


def test_bf16_training_gradients_vs_reference_fixture():
    """dim 48, 1 x 3 x 32 x 32: loss and EVERY parameter gradient of the bf16-activation path, tensor by tensor, against
    the fixture the reference module wrote in fp64 (`model_dim48_rgb_allgrads`: tensors up to 1024 elements in full --
    every temperature and LayerNorm parameter -- the others as a fixed 1024-element sub-sample, plus every norm).
    Measured values (MI355X) are quoted beside the bounds; a sign error or a factor of two in any one tensor, however
    small, fails the cosine / ratio of that tensor."""
    case = recipe.MODEL_CASES[1]
    g = load("model_" + case[0] + "_allgrads")
    net, x, tgt = _net(case)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(x)
    assert out.dtype == torch.float32 and out.requires_grad
    loss = (out - tgt).abs().mean()
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-3 * float(g["loss"])           # measured 2.8e-4
    gn_ref = math.sqrt(sum(float(g["gl2/" + k]) ** 2 for k, _ in net.named_parameters()))
    worst = {}
    for k, p in net.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), k
        ref_n, got_n = float(g["gl2/" + k]), float(p.grad.double().norm())
        share = ref_n / gn_ref
        idx = recipe.subsample_index(p.grad.numel(), count=1024)
        a = p.grad.detach().double().cpu().numpy().reshape(-1)[idx]
        b = g["g/" + k].astype(np.float64)
        kind = _kind(k)
        ratio = got_n / ref_n
        if kind == "temperature":
            # one to eight numbers per tensor, each a small difference of large sums: sign and size, element by element,
            # where the element carries weight (measured: ratios 0.62 .. 1.06, the outliers below 1e-4 of the norm)
            if share > 1e-4:
                # (elements far below the tensor's largest are noise: the bound is on the vector, which a flipped sign
                # or a factor of two in the element that matters breaks)
                assert np.abs(a - b).max() <= 0.2 * np.abs(b).max(), (k, a, b)
                assert 0.8 < ratio < 1.2, (k, ratio)                                  # measured 0.96 .. 1.03
            else:
                assert 0.4 < ratio < 1.6, (k, ratio)
            continue
        cos = float((a * b).sum() / math.sqrt((a * a).sum() * (b * b).sum()))
        w = worst.setdefault(kind, [1.0, 1.0, 1.0])
        w[0] = min(w[0], cos); w[1] = min(w[1], ratio); w[2] = max(w[2], ratio)
        # measured over all 451 tensors: cosine >= 0.9965 (the 192 / 384-channel levels, whose activations have been
        # rounded to bf16 most often on the way; >= 0.9997 at 48 channels), norm ratio 0.937 .. 1.038
        assert cos > (0.995 if share > 1e-3 else 0.985), (k, cos, share)
        assert 0.9 < ratio < 1.1, (k, ratio)
    gn = math.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in net.parameters()))
    assert abs(gn - gn_ref) <= 1e-2 * gn_ref, (gn, gn_ref)                            # measured 3e-3
    print("worst cosine / ratio range per kind:", worst)


def test_bf16_training_trajectory_vs_oracle():
    """Three steps of the reference's loop (zero_grad, forward, L1, backward, clip 1.0, Adam 1e-4 / wd 1e-6) under bf16
    autocast (packed bf16 forward and backward kernels) follow the fp64 oracle's loss curve."""
    from oracle import restormer_oracle as orc
    case = recipe.MODEL_CASES[1]
    net, x, tgt = _net(case)
    sd64, x64, t64 = recipe.model_case_tensors(case, torch.float64)
    ref_losses = orc.train_steps(x64, t64, sd64, (1, 2, 4, 8), 3)[0]
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=1e-6)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.l1_loss(net(x), tgt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=1.0)
        opt.step()
        losses.append(loss.item())
    assert np.allclose(losses, [float(v) for v in ref_losses], rtol=5e-3), (losses, ref_losses)
    assert losses[2] < losses[0]


def test_bf16_stage_backward_matches_the_fp32_block_gradient():
    """One stage of two blocks: output, input gradient and parameter gradients of the packed bf16 kernels against the
    fp32 kernels on the same bf16-rounded input, to bf16 accuracy (largest element error over largest element)."""
    from superresolution_amd import bf16
    from superresolution_amd.model.Restromer import TransformerBlock
    torch.manual_seed(3)
    blocks = torch.nn.Sequential(*[TransformerBlock(96, 2, 1.25, False, "WithBias") for _ in range(2)]).cuda()
    x = torch.randn(2, 96, 16, 24, device="cuda").bfloat16()
    r = torch.randn(2, 96, 16, 24, device="cuda")
    xa = x.clone().requires_grad_(True)
    ya = bf16.stage(xa, blocks)
    (ya.float() * r).sum().backward()
    ga = {k: p.grad.clone() for k, p in blocks.named_parameters()}
    for p in blocks.parameters():
        p.grad = None
    xb = x.float().requires_grad_(True)
    yb = blocks(xb)
    (yb * r).sum().backward()
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    assert rel(ya.float(), yb) < 2e-2
    assert rel(xa.grad.float(), xb.grad) < 3e-2
    for k, p in blocks.named_parameters():
        assert rel(ga[k], p.grad) < 5e-2, k


@pytest.mark.parametrize("autocast", [True, False])
def test_trainer_steps_leave_no_stale_prepared_weights(autocast):
    """DataParallelTrainer uses torch's fused Adam, which updates parameters without moving their version counters:
    after its steps the next forward (training and evaluation, bf16 and fp32) must see the new weights -- the result
    has to be the one obtained after dropping every prepared-weight cache by hand."""
    from superresolution_amd.parallel import Collectives, DataParallelTrainer
    net, x, tgt = _net(recipe.MODEL_CASES[1])
    trainer = DataParallelTrainer(net, world_size=1, comm=Collectives(eager=False), lr=1e-3)   # steps that matter
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        before = net(x).float()                                   # fills the inference caches with the initial weights
    for _ in range(2):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            trainer.step(x, tgt)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        got = net(x).float()
        net.invalidate_prepared()
        want = net(x).float()
    # (stale weights would be off by what the steps changed, asserted below to be far more than this)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert float((got - before).abs().max()) > 1e-2 * float(want.abs().max())     # the steps did change the output
    # and a training forward after a plain fused-optimiser step outside the trainer
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        (net(x) - tgt).abs().mean().backward()
        opt.step()
        l_got = float((net(x) - tgt).abs().mean().detach())
        net.invalidate_prepared()
        l_want = float((net(x) - tgt).abs().mean().detach())
    assert abs(l_got - l_want) <= 1e-5 * abs(l_want)


@pytest.mark.parametrize("autocast", [False, True])
def test_plain_fused_adam_loop_then_eval_sees_the_new_weights(autocast):
    """A training loop that is not the library's trainer (torch's fused Adam: parameters change without their version
    counters moving), then model.eval() and a no_grad forward, as validate() / test_psnr do: the evaluation has to run
    on the updated weights.  The mode switch drops the prepared weights."""
    net, x, tgt = _net(recipe.MODEL_CASES[1])
    net.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        before = net(x).float()                                   # fills the inference caches with the initial weights
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    for _ in range(2):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            (net(x).float() - tgt).abs().mean().backward()
        opt.step()
    net.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        got = net(x).float()
        net.invalidate_prepared()
        want = net(x).float()
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert float((got - before).abs().max()) > 1e-2 * float(want.abs().max())     # the steps did change the output


@pytest.mark.parametrize("autocast", [False, True])
def test_fused_adam_steps_without_a_mode_switch_are_seen(autocast):
    """The same loop with NO train()/eval() call between the steps and the evaluation (a script that evaluates under
    no_grad in whatever mode the module is in): every optimiser step of the process moves the generation that the
    prepared-weight caches are keyed on, so the evaluation still runs on the updated weights."""
    from superresolution_amd import ops
    net, x, tgt = _net(recipe.MODEL_CASES[1])
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        before = net(x).float()
    gen = ops.weights_generation()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    for _ in range(2):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            (net(x).float() - tgt).abs().mean().backward()
        opt.step()
    assert ops.weights_generation() == gen + 2
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        got = net(x).float()
        net.invalidate_prepared()
        want = net(x).float()
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert float((got - before).abs().max()) > 1e-2 * float(want.abs().max())
