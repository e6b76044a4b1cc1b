"""Timing of the dense 3x3 convolutions of the network (rst_conv3x3_fwd) against torch / MIOpen (development aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_amd import ops  # noqa: E402

SHAPES = [("patch_embed", 3, 48, 256, 0), ("down1_2", 48, 24, 256, 1), ("down2_3", 96, 48, 128, 1), ("down3_4", 192, 96, 64, 1),
          ("up4_3", 384, 768, 32, 2), ("up3_2", 192, 384, 64, 2), ("up2_1", 96, 192, 128, 2), ("output_upscale", 96, 192, 256, 2)]


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B = 16
tot = [0.0, 0.0]
for name, ci, co, s, mode in SHAPES:
    x = torch.randn(B, ci, s, s, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)
    with torch.no_grad():
        a = timeit(lambda: ops.conv3x3(x, w, None, mode))
        b = timeit(lambda: torch.nn.functional.conv2d(x, w, padding=1))
    fl = 18.0 * ci * co * B * s * s
    tot[0] += a; tot[1] += b
    print(f"{name:15s} {ci:4d}->{co:4d} @{s:3d}: hip {a:8.1f} us ({fl / a / 1e6:6.1f} TFLOP/s eq.)  miopen conv only {b:8.1f} us")
print(f"total hip {tot[0]:.0f} us, miopen {tot[1]:.0f} us (+ shuffles)")

# the packed bf16 kernels on the same shapes
from superresolution_amd import bf16 as pk  # noqa: E402
tot16 = 0.0
for name, ci, co, s, mode in SHAPES:
    x = torch.randn(B, ci, s, s, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") / (3.0 * ci ** 0.5)
    p = pk.pack(x)
    us = timeit(lambda: pk.conv3x3(p, w, None, mode))
    tot16 += us
    print(f"{name:16s} {ci:3d}->{co:4d} @{s:3d}: bf16 packed {us:8.1f} us")
print(f"total bf16 packed {tot16:.0f} us")
