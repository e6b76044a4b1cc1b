"""Generate tests/golden/*.npz by running the REFERENCE module next to the oracle.

TEST INFRASTRUCTURE ONLY.  Runs in the build container, where the reference
checkout is mounted read-only; the GPU box never sees the reference, only the
.npz files written here.  Usage (from the repo root):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden [/root/reference] [--allgrads-only]

For every case in ``oracle/recipe.py`` the reference module is constructed with
its own constructor, loaded (strict) with the recipe's state-dict -- which also
pins parameter names, shapes and registration order -- and evaluated in fp64
and fp32.  The oracle is evaluated on the same tensors and must agree with the
reference before anything is written.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import recipe
from . import restormer_oracle as orc

_paths = [a for a in sys.argv[1:] if not a.startswith("--")]
REF = _paths[0] if _paths else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    sys.path.insert(0, REF)
    try:
        import importlib
        mod = importlib.import_module("model.Restromer")
    finally:
        sys.path.remove(REF)
    return mod


def _pin(name, t: torch.Tensor, store: dict):
    """Store a tensor fully if small, else a fixed subsample plus two norms."""
    a = t.detach().to(torch.float64).cpu().numpy().reshape(-1)
    idx = recipe.subsample_index(a.size)
    store[name + "/v"] = a[idx].astype(np.float32)
    store[name + "/l2"] = np.float64(np.sqrt((a * a).sum()))
    store[name + "/sum"] = np.float64(a.sum())
    store[name + "/shape"] = np.array(t.shape, dtype=np.int64)


def _maxrel(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def gen_blocks(ref):
    for case in recipe.BLOCK_CASES:
        tag, dim, heads, ln_type, bias, b, h, w = case
        sd, x, r = recipe.block_case_tensors(case, torch.float64)
        blk = ref.TransformerBlock(dim, heads, recipe.FFN_FACTOR, bias, ln_type).double()
        assert [k for k, _ in blk.state_dict().items()] == list(sd.keys()), (tag, "name/order mismatch")
        blk.load_state_dict(sd, strict=True)
        xr = x.clone().requires_grad_(True)
        out = blk(xr)
        (out * r).sum().backward()

        # oracle, same tensors, autograd through the restated forward
        sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xo = x.clone().requires_grad_(True)
        oo = orc.transformer_block(xo, sdo, "", heads)
        (oo * r).sum().backward()
        assert _maxrel(oo, out) < 1e-12, (tag, "oracle fwd", _maxrel(oo, out))
        assert _maxrel(xo.grad, xr.grad) < 1e-10, (tag, "oracle dx")
        for (k, p) in blk.named_parameters():
            assert _maxrel(sdo[k].grad, p.grad) < 1e-9, (tag, k, _maxrel(sdo[k].grad, p.grad))

        blk32 = ref.TransformerBlock(dim, heads, recipe.FFN_FACTOR, bias, ln_type)
        blk32.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
        with torch.no_grad():
            out32 = blk32(x.float())

        store = {}
        _pin("out", out, store)
        _pin("out32", out32, store)
        _pin("dx", xr.grad, store)
        for k, p in blk.named_parameters():
            _pin("d/" + k, p.grad, store)
        # sub-module pins: LayerNorm, attention and feed-forward on their own
        with torch.no_grad():
            _pin("norm1", blk.norm1(x), store)
            _pin("attn", blk.attn(x), store)
            _pin("ffn", blk.ffn(x), store)
        np.savez_compressed(os.path.join(OUT, f"block_{tag}.npz"), **store)
        print(f"block_{tag}: fwd fp32-vs-fp64 maxrel {_maxrel(out32.double(), out):.2e}")


def gen_models(ref):
    for case in recipe.MODEL_CASES:
        tag, kwargs, shape = case
        sd, x, tgt = recipe.model_case_tensors(case, torch.float64)
        net = ref.Restormer(**kwargs).double()
        assert list(net.state_dict().keys()) == list(sd.keys()), (tag, "name/order mismatch")
        for k, v in net.state_dict().items():
            assert tuple(v.shape) == tuple(sd[k].shape), (tag, k)
        net.load_state_dict(sd, strict=True)
        xr = x.clone().requires_grad_(True)
        out = net(xr)
        loss = (out - tgt).abs().mean()
        loss.backward()

        with torch.no_grad():
            oo = orc.restormer_forward(x, sd)
        assert _maxrel(oo, out) < 1e-11, (tag, "oracle fwd", _maxrel(oo, out))

        net32 = ref.Restormer(**kwargs)
        net32.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
        with torch.no_grad():
            out32 = net32(x.float())

        store = {}
        _pin("out", out, store)
        _pin("out32", out32, store)
        _pin("dx", xr.grad, store)
        store["loss"] = np.float64(loss.item())
        store["psnr"] = np.float64(orc.psnr(out.detach(), tgt))
        gn = torch.sqrt(sum((p.grad ** 2).sum() for p in net.parameters()))
        store["grad_norm"] = np.float64(gn.item())
        names = [k for k, _ in net.named_parameters()]
        pick = names[:12] + names[len(names) // 2: len(names) // 2 + 12] + names[-12:]
        for k, p in net.named_parameters():
            if k in pick:
                _pin("d/" + k, p.grad, store)
            store["gl2/" + k] = np.float64(torch.sqrt((p.grad ** 2).sum()).item())
        np.savez_compressed(os.path.join(OUT, f"model_{tag}.npz"), **store)
        print(f"model_{tag}: loss {loss.item():.6f} fp32-vs-fp64 maxrel {_maxrel(out32.double(), out):.2e}")


ALLGRADS_COUNT = 1024      # elements pinned per parameter tensor in the every-tensor gradient fixture


def gen_model_grads_all(ref):
    """Every parameter gradient of the dim-48 network (loss = L1 against the recipe's target, fp64 reference module):
    tensors up to ALLGRADS_COUNT elements in full (all temperatures and LayerNorm parameters), the others as a fixed
    sub-sample of that many elements.  tests/test_gpu_bf16_train.py holds the bf16 training path against it tensor by
    tensor; model_<tag>.npz keeps only 36 tensors' sub-samples."""
    case = recipe.MODEL_CASES[1]
    tag, kwargs, shape = case
    sd, x, tgt = recipe.model_case_tensors(case, torch.float64)
    net = ref.Restormer(**kwargs).double()
    net.load_state_dict(sd, strict=True)
    loss = (net(x) - tgt).abs().mean()
    loss.backward()
    store = {"loss": np.float64(loss.item())}
    for k, p in net.named_parameters():
        a = p.grad.detach().cpu().numpy().reshape(-1)
        store["g/" + k] = a[recipe.subsample_index(a.size, count=ALLGRADS_COUNT)].astype(np.float32)
        store["gl2/" + k] = np.float64(np.sqrt((a * a).sum()))
    np.savez_compressed(os.path.join(OUT, f"model_{tag}_allgrads.npz"), **store)
    print(f"model_{tag}_allgrads: {len(store)} arrays, loss {loss.item():.6f}")


def gen_train(ref):
    """Three optimiser steps of the reference module with the stock optimiser,
    in the order of model/train.py:82-90, on the dim=12 default network."""
    case = recipe.MODEL_CASES[0]
    tag, kwargs, shape = case
    sd, x, tgt = recipe.model_case_tensors(case, torch.float64)
    net = ref.Restormer(**kwargs).double()
    net.load_state_dict(sd, strict=True)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=1e-6)
    crit = torch.nn.L1Loss()
    losses = []
    net.train()
    for _ in range(3):
        opt.zero_grad()
        out = net(x)
        loss = crit(out, tgt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=1.0)
        opt.step()
        losses.append(loss.item())
    o_losses, o_sd = orc.train_steps(x, tgt, sd, (1, 2, 4, 8), 3)
    assert np.allclose(losses, o_losses, rtol=1e-10), (losses, o_losses)
    store = {"losses": np.array(losses, dtype=np.float64)}
    for k, v in net.state_dict().items():
        assert _maxrel(o_sd[k], v) < 1e-8, k
        store["delta_l2/" + k] = np.float64(torch.sqrt(((v - sd[k]) ** 2).sum()).item())
    with torch.no_grad():
        out = net(x)
    _pin("out_after", out, store)
    store["psnr_after"] = np.float64(orc.psnr(out, tgt))
    np.savez_compressed(os.path.join(OUT, f"train_{tag}.npz"), **store)
    print("train:", losses)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    ref = _import_reference()
    if "--allgrads-only" in sys.argv:          # adds the every-tensor gradient fixture without touching the others
        gen_model_grads_all(ref)
        return
    gen_blocks(ref)
    gen_models(ref)
    gen_model_grads_all(ref)
    gen_train(ref)


if __name__ == "__main__":
    main()
