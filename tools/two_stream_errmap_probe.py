#!/usr/bin/env python3
"""Development probe: where in the output is a 192-channel fp32 block wrong when a second instance runs beside it?"""
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
ta = torch.randn(8, 192, 64, 64, generator=torch.Generator().manual_seed(3)).cuda()
with torch.no_grad():
    ref = big(ta); ref = big(ta)
    shown = 0
    for it in range(40):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            ob = big(ta)
        with torch.cuda.stream(s1):
            oa = big(ta)
        cur.wait_stream(s1); cur.wait_stream(s2)
        torch.cuda.synchronize()
        for tag, o in (("first-launched", ob), ("second-launched", oa)):
            bad = (o != ref)
            n = int(bad.sum())
            if n and shown < 8:
                shown += 1
                idx = bad.nonzero()
                bs = sorted(set(idx[:, 0].tolist())); cs = idx[:, 1]; px = idx[:, 2] * 64 + idx[:, 3]
                per_px = bad.view(8, 192, -1).any(1).sum().item()
                per_ch = bad.view(8, 192, -1).any(2).sum().item()
                print(f"run {it} {tag}: {n} wrong values, samples {bs}, channels {int(cs.min())}..{int(cs.max())} ({per_ch} (sample, channel) rows touched), "
                      f"pixels {int(px.min())}..{int(px.max())} ({per_px} (sample, pixel) columns touched), max err {float((o - ref).abs().max()):.2e}")
