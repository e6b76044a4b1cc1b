"""Accuracy of rst_bf16_block_bwd against the fp32 kernels over the cases of tests/test_gpu_bf16_bwd.py (development aid)."""
import sys, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_bf16_bwd as t
from superresolution_amd import bf16, ops
from superresolution_amd.model.Restromer import TransformerBlock
for (C, heads, hidden, H, W, ln_bias, conv_bias) in t.CASES:
    torch.manual_seed(C + hidden + H)
    blk = TransformerBlock(C, heads, (hidden + 0.5) / C, conv_bias, "WithBias" if ln_bias else "BiasFree").cuda()
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if p.dim() == 1:
                p.add_(0.3 * torch.randn_like(p))
    x = torch.randn(2, C, H, W, device="cuda").bfloat16(); dy = torch.randn(2, C, H, W, device="cuda").bfloat16()
    params = blk._params()
    dx_ref, g_ref = ops.block_recompute_backward(x.float(), dy.float(), heads, hidden, params)
    dxp, g = bf16.block_backward(bf16.pack(x), bf16.pack(dy), heads, hidden, params)
    dx = bf16.unpack(dxp, torch.float32)
    out = [("dx", t._cos(dx, dx_ref), float((dx - dx_ref).norm() / dx_ref.norm()))]
    for name, ref in g_ref.items():
        if ref is None or float(ref.norm()) < 1e-6: continue
        out.append((name, t._cos(g[name], ref), float((g[name] - ref).norm() / ref.norm())))
    w = min(out, key=lambda o: o[1]); r = max(out, key=lambda o: o[2])
    print(C, heads, hidden, "dx cos %.5f rel %.4f | worst cos %s %.5f | worst rel %s %.4f" % (out[0][1], out[0][2], w[0], w[1], r[0], r[2]))
