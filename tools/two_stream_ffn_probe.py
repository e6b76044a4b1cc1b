#!/usr/bin/env python3
"""Development probe: rst_ffn_fwd at 192 channels on two streams, two instances side by side, with and without the
residual operand (the residual selects the RES instantiation of the split-f16 1x1 kernel for project_out)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import recipe  # noqa: E402
from superresolution_amd import _lib, ops  # noqa: E402
from superresolution_amd._lib import BlockParams  # noqa: E402
from superresolution_amd.model.Restromer import Restormer  # noqa: E402

lib = _lib.load()
case = recipe.MODEL_CASES[1]
sd, _, _ = recipe.model_case_tensors(case, torch.float32)
net = Restormer(**case[1])
net.load_state_dict(sd, strict=True)
net = net.cuda().eval()
blk = net.decoder_level3[0]
f = blk.ffn
B, C, H, W = 8, 192, 64, 64
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(3)).cuda()
res = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(4)).cuda()
tensors = {n: None for n in ops.PARAM_FIELDS}
tensors.update(pin_weight=f.project_in.weight.detach(), ffn_dw_weight=f.dwconv.weight.detach(), pout_weight=f.project_out.weight.detach())
desc = ops.make_desc(B, C, H, W, 1, f.hidden_features, False, False)
par = ops._fill(BlockParams(), tensors)
nb = lib.rst_ffn_workspace_bytes(ctypes.byref(desc))


def ffn(residual):
    ws = ops.workspace(nb, x.device)
    y = torch.empty_like(x)
    st = lib.rst_ffn_fwd(ctypes.byref(desc), ctypes.byref(par), x.data_ptr(), None if residual is None else residual.data_ptr(),
                         y.data_ptr(), None, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "rst_ffn_fwd")
    return y


def pair(residual):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        ob = ffn(residual)
    with torch.cuda.stream(s1):
        oa = ffn(residual)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return oa, ob


with torch.no_grad():
    for name, r in (("no residual", None), ("residual", res)):
        ref = ffn(r); ref = ffn(r)
        torch.cuda.synchronize()
        da, db = [], []
        for _ in range(40):
            oa, ob = pair(r)
            da.append(float((oa - ref).abs().max())); db.append(float((ob - ref).abs().max()))
        torch.cuda.synchronize()
        print(f"{name:12s}: second-launched wrong {sum(1 for d in da if d > 0)}/40 (max {max(da):.1e}), first-launched wrong {sum(1 for d in db if d > 0)}/40 (max {max(db):.1e})")

# ---- is the split-f16 kernel the culprit rather than the victim?  rst_ffn_fwd (three times) on one stream, something
# else of the 192-channel block on the other
from superresolution_amd import ops as O  # noqa: E402

a = blk.attn
t3 = torch.randn(B, 3 * C, H, W, generator=torch.Generator().manual_seed(6)).cuda()
others = [
    ("whole block", lambda: blk(x)),
    ("attention module", lambda: a(x)),
    ("layernorm", lambda: blk.norm1(x)),
    ("conv1x1 96->... exact (qkv weights, K=192 on f16s)", lambda: O.conv1x1(x, a.qkv.weight, None)),
    ("depthwise 3x3", lambda: O.dwconv3x3(t3, a.qkv_dwconv.weight, None)),
]
with torch.no_grad():
    for name, fn in others:
        ref = fn(); ref = fn()
        torch.cuda.synchronize()
        bad = 0; worst = 0.0
        for _ in range(30):
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s2):
                for _ in range(3):
                    ffn(res)
            with torch.cuda.stream(s1):
                o = fn()
            cur.wait_stream(s1); cur.wait_stream(s2)
            d = float((o - ref).abs().max())
            bad += d > 0; worst = max(worst, d)
        torch.cuda.synchronize()
        print(f"beside rst_ffn_fwd x3: {name:52s} wrong {bad}/30 (max {worst:.1e})")
