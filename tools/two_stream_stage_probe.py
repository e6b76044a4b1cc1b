#!/usr/bin/env python3
"""Development probe: which stage of the fp32 network gives different results when two half-batches run on two streams
at the same time?  Every top-level module is run 40 times both ways on its own recorded input."""
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
shape = tuple(int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (2, 3, 64, 96)
x = torch.rand(*shape, generator=torch.Generator().manual_seed(2)).cuda()

inputs = {}
hooks = []
for name, mod in net.named_children():
    def keep(m, a, o, name=name):
        inputs.setdefault(name, a[0].detach().clone())          # (returns None: the output stays as it is)
    hooks.append(mod.register_forward_hook(keep))
with torch.no_grad():
    net(x)
    net(x)
for h in hooks:
    h.remove()


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


with torch.no_grad():
    for name, mod in list(net.named_children()) + [("whole network", net)]:
        t = x if name == "whole network" else inputs.get(name)
        if t is None:
            continue
        ref = one(mod, t)
        ref = one(mod, t)
        d1 = max(float((one(mod, t) - ref).abs().max()) for _ in range(10))
        ds = [float((two(mod, t) - ref).abs().max()) for _ in range(40)]
        torch.cuda.synchronize()
        bad = sum(1 for d in ds if d > 10 * max(d1, 1e-6))
        print(f"{name:22s} one stream run-to-run {d1:.2e}   two streams max {max(ds):.2e}   outliers {bad}/40")
