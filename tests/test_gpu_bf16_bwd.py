"""rst_bf16_block_bwd: the gradients of one TransformerBlock on packed bf16 tensors against the fp32 kernels'
gradients at the same bf16-rounded input (which are themselves pinned to the fp64 oracle in test_gpu_block.py).
Tolerances are bf16's (8 significant bits, activations and activation gradients rounded at every kernel boundary);
measured values are quoted next to the bounds."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (C, heads, hidden, H, W, LayerNorm with bias, convolution biases)
CASES = [
    (48, 1, 60, 32, 32, False, False),       # level 1 of the network (hidden = int(1.25 C))
    (48, 1, 60, 16, 24, True, True),
    (96, 2, 120, 16, 32, False, False),      # level 2; hidden 120 fills 15 channel blocks
    (96, 1, 120, 12, 16, True, False),       # decoder level 1 / refinement: one head of 96 channels
    (192, 4, 240, 8, 16, False, False),      # level 3: the qkv data gradient contracts 576 channels in two slices
    (384, 8, 480, 8, 8, True, True),         # latent: 1152 / 960 channels in three / two slices
    (48, 1, 127, 8, 16, False, True),        # a hidden width that is not a multiple of 4
]


def _cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-300))


@pytest.mark.parametrize("C,heads,hidden,H,W,ln_bias,conv_bias", CASES)
def test_block_backward_vs_fp32_kernels(C, heads, hidden, H, W, ln_bias, conv_bias):
    from superresolution_amd import bf16, ops
    from superresolution_amd.model.Restromer import TransformerBlock
    torch.manual_seed(C + hidden + H)
    blk = TransformerBlock(C, heads, (hidden + 0.5) / C, conv_bias, "WithBias" if ln_bias else "BiasFree").cuda()
    assert blk.ffn.hidden_features == hidden
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if p.dim() == 1:                      # LayerNorm weights / biases, temperature, conv biases: away from 1 / 0
                p.add_(0.3 * torch.randn_like(p))
    B = 2
    x = torch.randn(B, C, H, W, device="cuda").bfloat16()
    dy = torch.randn(B, C, H, W, device="cuda").bfloat16()
    params = blk._params()
    dx_ref, g_ref = ops.block_recompute_backward(x.float(), dy.float(), heads, hidden, params)
    dxp, g = bf16.block_backward(bf16.pack(x), bf16.pack(dy), heads, hidden, params)
    dx = bf16.unpack(dxp, torch.float32)
    assert torch.isfinite(dx).all()
    assert _cos(dx, dx_ref) > 0.999, _cos(dx, dx_ref)                         # measured 0.999997
    rel = float((dx - dx_ref).norm() / dx_ref.norm())
    assert rel < 3e-2, rel                                                    # measured 2.4e-3 (the output's own rounding)
    for name, ref in g_ref.items():
        if ref is None:
            assert g[name] is None
            continue
        got = g[name]
        assert got.dtype == torch.float32 and got.shape == ref.shape and torch.isfinite(got).all(), name
        if float(ref.norm()) < 1e-6:
            continue
        c = _cos(got, ref)
        r = float((got - ref).norm() / ref.norm())
        assert c > 0.995 and r < 0.1, (name, c, r)                            # measured cos >= 0.99998; rel <= 6e-3, temperature 2.9e-2


def test_stage_backward_routes():
    """bf16.stage differentiates through rst_bf16_block_bwd by default and through the fp32 kernels with
    RST_BF16_BWD=fp32; both give the same gradients to bf16 accuracy."""
    import os
    from superresolution_amd import bf16
    from superresolution_amd.model.Restromer import TransformerBlock
    torch.manual_seed(5)
    blocks = torch.nn.Sequential(*[TransformerBlock(48, 1, 1.25, False, "BiasFree") for _ in range(2)]).cuda()
    x = torch.randn(2, 48, 16, 16, device="cuda").bfloat16()
    r = torch.randn(2, 48, 16, 16, device="cuda")
    res = {}
    for mode in ("bf16", "fp32"):
        os.environ["RST_BF16_BWD"] = mode
        try:
            for p in blocks.parameters():
                p.grad = None
            xa = x.clone().requires_grad_(True)
            (bf16.stage(xa, blocks).float() * r).sum().backward()
            res[mode] = (xa.grad.float().clone(), {k: p.grad.clone() for k, p in blocks.named_parameters()})
        finally:
            os.environ.pop("RST_BF16_BWD", None)
    assert _cos(res["bf16"][0], res["fp32"][0]) > 0.998
    for k in res["fp32"][1]:
        assert _cos(res["bf16"][1][k], res["fp32"][1][k]) > 0.99, k


_ONE_STREAM_CHILD = r"""
import sys, torch
sys.path.insert(0, {root!r})
from superresolution_amd import bf16
from superresolution_amd.model.Restromer import TransformerBlock
out = []
for (C, heads, hidden, H, W, ln_bias, conv_bias) in {cases!r}:
    torch.manual_seed(C + hidden + H)
    blk = TransformerBlock(C, heads, (hidden + 0.5) / C, conv_bias, "WithBias" if ln_bias else "BiasFree").cuda()
    x = torch.randn(2, C, H, W, device="cuda").bfloat16()
    dy = torch.randn(2, C, H, W, device="cuda").bfloat16()
    dxp, g = bf16.block_backward(bf16.pack(x), bf16.pack(dy), heads, hidden, blk._params())
    out.append((bf16.unpack(dxp, torch.float32).cpu(), {{k: (v.cpu() if v is not None else None) for k, v in g.items()}}))
torch.save(out, {out!r})
"""


def test_block_backward_side_stream_changes_nothing(tmp_path):
    """With RST_BWD_STREAMS=1 (a child process: the switch is read once) the parameter gradients of a block run on a
    second stream beside the chain of data gradients; by default everything runs on the caller's stream.  Same kernels,
    same sums in the same order: the results are equal bit for bit, twice in a row (the second call reuses the events)."""
    import os
    import subprocess
    import sys
    from superresolution_amd import bf16
    from superresolution_amd.model.Restromer import TransformerBlock
    cases = [CASES[1], CASES[2], CASES[5]]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "side_stream.pt")
    env = dict(os.environ, RST_BWD_STREAMS="1")
    r = subprocess.run([sys.executable, "-c", _ONE_STREAM_CHILD.format(root=root, cases=cases, out=out)], env=env, cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    ref = torch.load(out, weights_only=True)
    for case, (dx_ref, g_ref) in zip(cases, ref):
        C, heads, hidden, H, W, ln_bias, conv_bias = case
        torch.manual_seed(C + hidden + H)
        blk = TransformerBlock(C, heads, (hidden + 0.5) / C, conv_bias, "WithBias" if ln_bias else "BiasFree").cuda()
        x = torch.randn(2, C, H, W, device="cuda").bfloat16()
        dy = torch.randn(2, C, H, W, device="cuda").bfloat16()
        for _ in range(2):
            dxp, g = bf16.block_backward(bf16.pack(x), bf16.pack(dy), heads, hidden, blk._params())
            assert torch.equal(bf16.unpack(dxp, torch.float32).cpu(), dx_ref), case
            for k, v in g_ref.items():
                if v is None:
                    assert g[k] is None
                else:
                    assert torch.equal(g[k].cpu(), v), (case, k)


@pytest.mark.parametrize("C,heads,hidden,H,W,ln_bias,conv_bias", [CASES[0], CASES[1], CASES[3], CASES[4], CASES[5]])
def test_block_backward_with_kept_attention(C, heads, hidden, H, W, ln_bias, conv_bias):
    """rst_bf16_block_fwd_keep + rst_bf16_block_bwd_kept: the backward pass reads the attention the forward pass kept
    instead of forming it again.  The forward output is that of rst_bf16_block_fwd bit for bit; the gradients agree
    with the recomputing route to bf16 accuracy (the two see q and k through differently fused kernels) and with the
    fp32 kernels' gradients within the bounds of the test above."""
    from superresolution_amd import bf16, ops
    from superresolution_amd.model.Restromer import TransformerBlock
    torch.manual_seed(C + hidden + H)
    blk = TransformerBlock(C, heads, (hidden + 0.5) / C, conv_bias, "WithBias" if ln_bias else "BiasFree").cuda()
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if p.dim() == 1:
                p.add_(0.3 * torch.randn_like(p))
    x = torch.randn(2, C, H, W, device="cuda").bfloat16()
    dy = torch.randn(2, C, H, W, device="cuda").bfloat16()
    params = blk._params()
    y_keep, kept = bf16.block_forward_keep(bf16.pack(x), heads, hidden, params)
    y_plain = bf16.block_(bf16.pack(x), heads, hidden, params)
    assert torch.equal(y_keep.data, y_plain.data)
    dxp_k, g_k = bf16.block_backward(bf16.pack(x), bf16.pack(dy), heads, hidden, params, kept_attn=kept)
    dxp_r, g_r = bf16.block_backward(bf16.pack(x), bf16.pack(dy), heads, hidden, params)
    dx_k, dx_r = bf16.unpack(dxp_k, torch.float32), bf16.unpack(dxp_r, torch.float32)
    assert torch.isfinite(dx_k).all()
    assert float((dx_k - dx_r).norm() / dx_r.norm()) < 1e-2                   # measured <= 3e-3
    dx_ref, g_ref = ops.block_recompute_backward(x.float(), dy.float(), heads, hidden, params)
    assert _cos(dx_k, dx_ref) > 0.999 and float((dx_k - dx_ref).norm() / dx_ref.norm()) < 3e-2
    for name, ref in g_ref.items():
        if ref is None:
            assert g_k[name] is None
            continue
        if float(ref.norm()) < 1e-6:
            continue
        assert torch.isfinite(g_k[name]).all(), name
        assert float((g_k[name] - g_r[name]).norm() / g_r[name].norm()) < 5e-2, name      # measured <= 1e-2 (temperature)
        assert _cos(g_k[name], ref) > 0.995 and float((g_k[name] - ref).norm() / ref.norm()) < 0.1, name
    with pytest.raises(RuntimeError):
        bf16.block_backward(bf16.pack(x), bf16.pack(dy), heads, hidden, params, kept_attn=kept[:16])


@pytest.mark.parametrize("C,heads,hidden,H,W,ln_bias,conv_bias", CASES)
def test_block_backward_with_saved_activations(C, heads, hidden, H, W, ln_bias, conv_bias):
    """rst_bf16_block_fwd_save + rst_bf16_block_bwd_saved: the forward pass leaves every intermediate the backward pass
    reads, the backward pass evaluates nothing again.  The forward pass runs the same unfused kernels the recomputing
    backward runs, so the gradients equal those of rst_bf16_block_bwd bit for bit; the output agrees with the
    inference forward to bf16 accuracy."""
    from superresolution_amd import bf16
    from superresolution_amd.model.Restromer import TransformerBlock
    torch.manual_seed(C + hidden + H)
    blk = TransformerBlock(C, heads, (hidden + 0.5) / C, conv_bias, "WithBias" if ln_bias else "BiasFree").cuda()
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if p.dim() == 1:
                p.add_(0.3 * torch.randn_like(p))
    x = torch.randn(2, C, H, W, device="cuda").bfloat16()
    dy = torch.randn(2, C, H, W, device="cuda").bfloat16()
    params = blk._params()
    xp = bf16.pack(x)
    y_save, saved = bf16.block_forward_save(xp, heads, hidden, params)
    assert torch.equal(xp.data, bf16.pack(x).data)                      # the input is left alone
    y_plain = bf16.block_(bf16.pack(x), heads, hidden, params)
    a, b = bf16.unpack(y_save, torch.float32), bf16.unpack(y_plain, torch.float32)
    # (not bit for bit even where nothing is fused: the gate reads the depthwise result after its rounding to bf16 here)
    assert float((a - b).norm() / b.norm()) < 1e-2                            # measured <= 2.5e-3
    for _ in range(2):                                                        # the saved buffer is only read
        dxp_s, g_s = bf16.block_backward(bf16.pack(x), bf16.pack(dy), heads, hidden, params, saved=saved)
        dxp_r, g_r = bf16.block_backward(bf16.pack(x), bf16.pack(dy), heads, hidden, params)
        assert torch.equal(dxp_s.data, dxp_r.data)
        for name, ref in g_r.items():
            if ref is None:
                assert g_s[name] is None
            else:
                assert torch.equal(g_s[name], ref), name
    with pytest.raises(RuntimeError):
        bf16.block_backward(bf16.pack(x), bf16.pack(dy), heads, hidden, params, saved=saved[:64])
    with pytest.raises(RuntimeError):
        bf16.block_backward(bf16.pack(x), bf16.pack(dy), heads, hidden, params, saved=saved, kept_attn=saved)


def test_stage_saves_or_recomputes(monkeypatch):
    """bf16.stage: saved activations (default while they fit), kept attention (RST_BF16_SAVE=0) and input only
    (RST_BF16_KEEP_ATTN=0 as well) give the same gradients to bf16 accuracy."""
    from superresolution_amd import bf16
    from superresolution_amd.model.Restromer import TransformerBlock
    torch.manual_seed(9)
    blocks = torch.nn.Sequential(*[TransformerBlock(48, 1, 1.25, False, "BiasFree") for _ in range(2)]).cuda()
    x = torch.randn(2, 48, 16, 16, device="cuda").bfloat16()
    r = torch.randn(2, 48, 16, 16, device="cuda")
    res = {}
    for name, env in (("saved", {}), ("forced", {"RST_BF16_SAVE": "1"}), ("kept", {"RST_BF16_SAVE": "0"}),
                      ("input", {"RST_BF16_SAVE": "0", "RST_BF16_KEEP_ATTN": "0"})):
        for k in ("RST_BF16_SAVE", "RST_BF16_KEEP_ATTN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for p in blocks.parameters():
            p.grad = None
        xa = x.clone().requires_grad_(True)
        y = bf16.stage(xa, blocks)
        (y.float() * r).sum().backward()
        res[name] = (y.detach().float().clone(), xa.grad.float().clone(), {k: p.grad.clone() for k, p in blocks.named_parameters()})
    assert torch.equal(res["saved"][1], res["forced"][1])
    for name in ("kept", "input"):
        assert float((res[name][0] - res["saved"][0]).norm() / res["saved"][0].norm()) < 1e-2
        assert _cos(res[name][1], res["saved"][1]) > 0.999
        for k in res["saved"][2]:
            assert _cos(res[name][2][k], res["saved"][2][k]) > 0.995, (name, k)


def test_stage_mixes_saved_and_recomputed_blocks(monkeypatch):
    """When only some blocks of a stage fit the memory budget, those save their intermediates and the rest keep their
    attention and recompute: the gradients are those of the all-recomputing stage to bf16 accuracy, block by block in the
    right order."""
    from superresolution_amd import bf16
    from superresolution_amd.model.Restromer import TransformerBlock
    torch.manual_seed(11)
    blocks = torch.nn.Sequential(*[TransformerBlock(48, 1, 1.25, False, "WithBias") for _ in range(3)]).cuda()
    x = torch.randn(2, 48, 16, 16, device="cuda").bfloat16()
    r = torch.randn(2, 48, 16, 16, device="cuda")

    def run():
        for p in blocks.parameters():
            p.grad = None
        xa = x.clone().requires_grad_(True)
        (bf16.stage(xa, blocks).float() * r).sum().backward()
        return xa.grad.float().clone(), {k: p.grad.clone() for k, p in blocks.named_parameters()}

    monkeypatch.setenv("RST_BF16_SAVE", "0")
    ref = run()
    monkeypatch.delenv("RST_BF16_SAVE")
    # a budget (a quarter of "free" memory) that holds the buffers of one block and a half
    lib = bf16._lib.load()
    need = bf16._saved_bytes(lib, bf16.pack(x), 1, blocks[0].ffn.hidden_features, blocks[0]._params())
    modes = []
    real_into = bf16._block_into

    def spy(lib_, src, dst, heads, hidden, params, cache, keep_attn=False, save=False):
        modes.append("saved" if save else ("kept" if keep_attn else "input"))
        return real_into(lib_, src, dst, heads, hidden, params, cache, keep_attn=keep_attn, save=save)
    monkeypatch.setattr(bf16, "_block_into", spy)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a, **k: (int(need * 1.5) * 4, 1 << 40))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *a, **k: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda *a, **k: 0)
    got = run()
    assert modes == ["saved", "kept", "kept"], modes
    assert _cos(got[0], ref[0]) > 0.999
    for k in ref[1]:
        assert _cos(got[1][k], ref[1][k]) > 0.995, k
