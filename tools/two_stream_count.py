#!/usr/bin/env python3
"""Development probe: two instances of one 192-channel fp32 block on two streams; how many of N runs leave the
first-launched instance with a result that differs from the single-stream result?"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
with torch.no_grad():
    ref = big(ta)
    with torch.cuda.stream(s2):
        big(ta)
    with torch.cuda.stream(s1):
        big(ta)
    torch.cuda.synchronize()
    wrong_first = wrong_second = 0
    for it in range(n):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            ob = big(ta)
        with torch.cuda.stream(s1):
            oa = big(ta)
        cur.wait_stream(s1); cur.wait_stream(s2)
        torch.cuda.synchronize()
        wrong_first += int(bool((ob != ref).any()))
        wrong_second += int(bool((oa != ref).any()))
print(f"{os.environ.get('PROBE_TAG', '')}: first-launched wrong {wrong_first}/{n}, second-launched wrong {wrong_second}/{n}")
