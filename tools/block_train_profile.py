"""Per-kernel times of one TransformerBlock's training forward + backward at one shape (development aid):
block_train_profile.py C heads H [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from superresolution_amd.model.Restromer import TransformerBlock  # noqa: E402

C, heads, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
torch.manual_seed(0)
blk = TransformerBlock(C, heads, 1.25, False, "WithBias").cuda()
x = torch.rand(B, C, H, H, device="cuda", requires_grad=True)
r = torch.rand(B, C, H, H, device="cuda")
reps = 5


def run():
    for _ in range(reps):
        (blk(x) * r).sum().backward()


run()
prof = bench.kernel_profile(run)
tot = 0.0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    us = v["ms"] * 1e3 / v["launches"]
    tot += v["ms"] * 1e3 / reps
    print(f"{k:22s} {v['launches'] // reps:3d} x {us:8.1f} us = {v['ms'] * 1e3 / reps:8.1f}  {v['bytes'] / v['ms'] / 1e6:7.1f} GB/s  {v['flops'] / v['ms'] / 1e9:7.1f} TFLOP/s")
print(f"block fwd+bwd total {tot:.1f} us")
