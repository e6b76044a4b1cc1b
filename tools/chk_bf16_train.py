#!/usr/bin/env python3
"""Development probe: the bf16 training path against the fp32 HIP path on the same parameters, before and after a
number of optimiser steps taken the way bench.py takes them (loss, global gradient norm, worst per-parameter cosine)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_amd.model.Restromer import Restormer  # noqa: E402
from superresolution_amd.parallel import Collectives, DataParallelTrainer  # noqa: E402

torch.manual_seed(0)
net = Restormer(3, 3, dim=48).cuda()
params = [p for p in net.parameters()]
names = [n for n, _ in net.named_parameters()]


def compare(tag, B=1):
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, 128, 128, generator=gen).cuda()
    tgt = torch.rand(B, 3, 256, 256, generator=gen).cuda()
    l32 = torch.nn.functional.l1_loss(net(x), tgt)
    g32 = torch.autograd.grad(l32, params, allow_unused=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l16 = torch.nn.functional.l1_loss(net(x).float(), tgt)
    g16 = torch.autograd.grad(l16, params, allow_unused=True)
    n32 = torch.sqrt(sum((g.double() ** 2).sum() for g in g32))
    n16 = torch.sqrt(sum((g.double() ** 2).sum() for g in g16))
    cos = []
    for n, a, b in zip(names, g32, g16):
        cos.append((float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm() + 1e-300)), n,
                    float(b.double().norm() / (a.double().norm() + 1e-300))))
    cos.sort()
    print(f"{tag}: loss fp32 {float(l32.detach()):.6f} bf16 {float(l16.detach()):.6f}  grad norm fp32 {float(n32):.5f} bf16 {float(n16):.5f}")
    for c in cos[:4]:
        print("    ", c)


compare("fresh")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
trainer = DataParallelTrainer(net, world_size=1, comm=Collectives(eager=False))
gen = torch.Generator(device="cuda").manual_seed(100)
x = torch.rand(8, 3, 128, 128, device="cuda", generator=gen)
tgt = torch.rand(8, 3, 256, 256, device="cuda", generator=gen)
v0 = params[5]._version
for i in range(steps):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = trainer.step(x, tgt)
    if i % 6 == 0:
        print("step", i, float(loss))
print("version counter of a parameter before / after the fused optimiser's steps:", v0, params[5]._version)
compare(f"after {steps} bf16 steps")
net.invalidate_prepared()
compare("after invalidate_prepared")
