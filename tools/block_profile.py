"""Per-kernel times of the inference block at one shape, from the library's event timers (development aid):
block_profile.py C heads H [reps] [B] [W]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from superresolution_amd import ops  # noqa: E402
from superresolution_amd.model.Restromer import TransformerBlock  # noqa: E402

C, heads, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
B = int(sys.argv[5]) if len(sys.argv) > 5 else 16
W = int(sys.argv[6]) if len(sys.argv) > 6 else H
torch.manual_seed(0)
blk = TransformerBlock(C, heads, 1.25, False, "WithBias").cuda()
x = torch.rand(B, C, H, W, device="cuda")
par = blk._params()


def run():
    for _ in range(reps):
        ops.block_forward_inplace(x, heads, blk.ffn.hidden_features, par)


run()
prof = bench.kernel_profile(run)
tot = 0.0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    us = v["ms"] * 1e3 / v["launches"]
    tot += v["ms"] * 1e3 / reps
    print(f"{k:18s} {v['launches'] // reps:3d} x {us:8.1f} us  {v['bytes'] / v['ms'] / 1e6:7.1f} GB/s  {v['flops'] / v['ms'] / 1e9:7.1f} TFLOP/s")
print(f"block total {tot:.1f} us  ({os.environ.get('RST_F32_MFMA', '0')})")
