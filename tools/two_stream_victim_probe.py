#!/usr/bin/env python3
"""Development probe: a 192-channel fp32 block on one stream beside a 48-channel block (or a plain copy kernel) on
another: which of the two gets wrong results?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import recipe  # noqa: E402
from superresolution_amd.model.Restromer import Restormer  # noqa: E402

case = recipe.MODEL_CASES[1]
sd, _, _ = recipe.model_case_tensors(case, torch.float32)
net = Restormer(**case[1])
net.load_state_dict(sd, strict=True)
net = net.cuda().eval()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
big = net.decoder_level3[0]
small = net.encoder_level1[0]
ta = torch.randn(8, 192, 64, 64, generator=torch.Generator().manual_seed(3)).cuda()
tb = torch.randn(8, 48, 256, 256, generator=torch.Generator().manual_seed(4)).cuda()
tc = torch.randn(64 << 20, generator=torch.Generator().manual_seed(5)).cuda()


def pair(fa, xa, fb, xb):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        ob = fb(xb)
    with torch.cuda.stream(s1):
        oa = fa(xa)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return oa, ob


with torch.no_grad():
    ra, rb = big(ta), small(tb)
    ra, rb = big(ta), small(tb)
    rc = tc * 2.0
    for name, fb, xb, ref_b in (("48-channel block", small, tb, rb), ("elementwise torch kernel", lambda v: v * 2.0, tc, rc),
                                ("another 192-channel block", big, ta, ra)):
        da, db = [], []
        for _ in range(30):
            oa, ob = pair(big, ta, fb, xb)
            da.append(float((oa - ra).abs().max())); db.append(float((ob - ref_b).abs().max()))
        torch.cuda.synchronize()
        print(f"192-channel block beside {name:28s}: 192-block wrong {sum(1 for d in da if d > 0)}/30 (max {max(da):.1e}), neighbour wrong {sum(1 for d in db if d > 0)}/30 (max {max(db):.1e})")
