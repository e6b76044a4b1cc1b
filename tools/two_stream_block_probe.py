#!/usr/bin/env python3
"""Development probe: the prepared block path on two streams -- shared prepared buffer against one per stream, and the
unprepared in-place block."""
import copy
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
side = torch.cuda.Stream()
main2 = torch.cuda.Stream()
B, C, H, W = 16, 192, 64, 64
blk = net.decoder_level3[0]
blk2 = copy.deepcopy(blk)
blk2._f32_cache = None
t = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(3)).cuda()
a, b = t.chunk(2)


def run(fa, fb, sa, sb):
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sb):
        ob = fb(b)
    with torch.cuda.stream(sa):
        oa = fa(a)
    cur.wait_stream(sa); cur.wait_stream(sb)
    return torch.cat([oa, ob])


def inplace(block):
    def f(v):
        v = v.clone()
        return ops.block_forward_inplace(v, block.attn.num_heads, block.ffn.hidden_features, block._params())
    return f


with torch.no_grad():
    ref = torch.cat([blk(a), blk(b)]); ref = torch.cat([blk(a), blk(b)]); blk2(a)
    cur = torch.cuda.current_stream()
    for name, fa, fb, sa, sb in [
        ("shared prepared buffer, default stream + side", blk, blk, cur, side),
        ("shared prepared buffer, two side streams", blk, blk, main2, side),
        ("own prepared buffer per stream, two side streams", blk, blk2, main2, side),
        ("unprepared in-place block, two side streams", inplace(blk), inplace(blk), main2, side),
    ]:
        for _ in range(2):
            run(fa, fb, sa, sb)
        torch.cuda.synchronize()
        ds = [float((run(fa, fb, sa, sb) - ref).abs().max()) for _ in range(30)]
        torch.cuda.synchronize()
        print(f"{name:52s} differing {sum(1 for d in ds if d > 1e-6)}/30  max {max(ds):.2e}")
