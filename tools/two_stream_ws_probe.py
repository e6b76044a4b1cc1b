#!/usr/bin/env python3
"""Development probe: after a two-stream run in which one instance of a 192-channel fp32 block is wrong, which of the
buffers its call left in its workspace differ from those of a clean run?  (rst_block_fwd carves: xn | sub, and the
feed-forward's carve of sub, the last to be written: u = project_in output, g = gated depthwise output.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import recipe  # noqa: E402
from superresolution_amd import ops  # noqa: E402
from superresolution_amd.model.Restromer import Restormer  # noqa: E402

case = recipe.MODEL_CASES[1]
sd, _, _ = recipe.model_case_tensors(case, torch.float32)
net = Restormer(**case[1])
net.load_state_dict(sd, strict=True)
net = net.cuda().eval()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
big = net.decoder_level3[0]
B, C, H, W = 8, 192, 64, 64
N, Hd = H * W, big.ffn.hidden_features
ta = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(3)).cuda()


def up256(nbytes):
    return (nbytes + 255) // 256 * 256


xn_f = B * C * N
sub0 = up256(xn_f * 4) // 4
u_f, g_f = B * 2 * Hd * N, B * Hd * N
regions = [("xn (unused scratch here)", 0, xn_f), ("u: project_in output", sub0, sub0 + u_f),
           ("g: gated depthwise output", sub0 + up256(u_f * 4) // 4, sub0 + up256(u_f * 4) // 4 + g_f)]
with torch.no_grad():
    ref = big(ta)
    with torch.cuda.stream(s2):
        big(ta); big(ta)
    with torch.cuda.stream(s1):
        big(ta)
    torch.cuda.synchronize()
    key = (ta.device.index, s2.cuda_stream)
    clean = ops._workspaces[key].clone()
    found = 0
    for it in range(60):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            ob = big(ta)
        with torch.cuda.stream(s1):
            big(ta)
        cur.wait_stream(s1); cur.wait_stream(s2)
        torch.cuda.synchronize()
        if bool((ob != ref).any()):
            ws = ops._workspaces[key]
            assert ws.numel() == clean.numel()
            a = ws[: ws.numel() // 4 * 4].view(torch.float32); b = clean[: ws.numel() // 4 * 4].view(torch.float32)
            diff = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
            print(f"run {it}: output wrong in {int((ob != ref).sum())} values; workspace floats differing: {int(diff.sum())}")
            for name, lo, hi in regions:
                d = diff[lo:hi]
                n = int(d.sum())
                where = ""
                if n:
                    idx = d.nonzero().flatten()
                    where = f" first at +{int(idx[0])}, last at +{int(idx[-1])}"
                print(f"    {name:28s}: {n} of {hi - lo}{where}")
            tail = diff[regions[-1][2]:]
            print(f"    beyond g (fold, fragments, what the attention left): {int(tail.sum())} of {tail.numel()}")
            found += 1
            if found >= 3:
                break
    print("wrong runs seen:", found)
