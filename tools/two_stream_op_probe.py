#!/usr/bin/env python3
"""Development probe: operators of one 384-channel block, two half-batches on two streams against one stream."""
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
net.two_streams = False
side = torch.cuda.Stream()
B, C, H, W = 5, 384, 4, 32
if len(sys.argv) > 1:
    C = int(sys.argv[1]); H = int(sys.argv[2]); W = int(sys.argv[3]); B = int(sys.argv[4]) if len(sys.argv) > 4 else B
stage = {384: net.latent, 192: net.decoder_level3, 96: net.decoder_level2, 48: net.encoder_level1}[C]
blk = stage[0]
t = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(3)).cuda()


def two(fn, t):
    a, b = t.chunk(2)
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        ob = fn(b)
    oa = fn(a)
    cur.wait_stream(side)
    ob.record_stream(cur)
    return torch.cat([oa, ob])


def one(fn, t):
    a, b = t.chunk(2)
    return torch.cat([fn(a), fn(b)])


a = blk.attn
f = blk.ffn
hidden = f.hidden_features
t3 = torch.randn(B, 3 * C, H, W, generator=torch.Generator().manual_seed(4)).cuda()
t2h = torch.randn(B, 2 * hidden, H, W, generator=torch.Generator().manual_seed(5)).cuda()
th = torch.randn(B, hidden, H, W, generator=torch.Generator().manual_seed(6)).cuda()
cases = [
    ("block", blk, t), ("attention module", a, t), ("feed-forward module", f, t), ("layernorm", blk.norm1, t),
    ("conv1x1 qkv", lambda v: ops.conv1x1(v, a.qkv.weight, a.qkv.bias), t),
    ("dwconv qkv", lambda v: ops.dwconv3x3(v, a.qkv_dwconv.weight, a.qkv_dwconv.bias), t3),
    ("conv1x1 project_out", lambda v: ops.conv1x1(v, a.project_out.weight, a.project_out.bias), t),
    ("conv1x1 project_in", lambda v: ops.conv1x1(v, f.project_in.weight, f.project_in.bias), t),
    ("dwconv ffn", lambda v: ops.dwconv3x3(v, f.dwconv.weight, f.dwconv.bias), t2h),
    ("conv1x1 ffn out", lambda v: ops.conv1x1(v, f.project_out.weight, f.project_out.bias), th),
]
with torch.no_grad():
    for name, fn, inp in cases:
        ref = one(fn, inp)
        ref = one(fn, inp)
        d1 = max(float((one(fn, inp) - ref).abs().max()) for _ in range(5))
        ds = [float((two(fn, inp) - ref).abs().max()) for _ in range(40)]
        torch.cuda.synchronize()
        bad = sum(1 for d in ds if d > 0)
        print(f"{name:22s} one stream run-to-run {d1:.2e}   two streams max {max(ds):.2e}   differing {bad}/40")
