#!/usr/bin/env python3
"""Per-level kernel times of the bf16 TransformerBlock: one block of every level of Restormer(dim=48) at the
bench shape (16 x 3 x 256 x 256 input), library event timers on.  usage: bench_bf16_blocks.py [batch] [size]"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superresolution_amd import _lib, bf16                                   # noqa: E402
from superresolution_amd.model.Restromer import TransformerBlock             # noqa: E402

LEVELS = [("enc1", 48, 1, 1), ("enc2", 96, 2, 2), ("enc3", 192, 4, 4), ("latent", 384, 8, 8), ("dec1", 96, 1, 1)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    lib = _lib.load()
    torch.manual_seed(0)
    for name, C, heads, d in LEVELS:
        blk = TransformerBlock(C, heads, 1.25, False, "WithBias").cuda()
        x = torch.randn(B, C, S // d, S // d, device="cuda")
        cache = bf16.PreparedCache()
        with torch.no_grad():
            p = bf16.pack(x)
            for _ in range(3):
                bf16.block_(p, heads, blk.ffn.hidden_features, blk._params(), cache)
            torch.cuda.synchronize()
            lib.rst_profile_begin()
            for _ in range(5):
                bf16.block_(p, heads, blk.ffn.hidden_features, blk._params(), cache)
            torch.cuda.synchronize()
            buf = ctypes.create_string_buffer(1 << 16)
            lib.rst_profile_end(buf, len(buf))
        prof = json.loads(buf.value.decode())
        tot = sum(v["ms"] for v in prof.values()) / 5
        print(f"{name}: C={C} heads={heads} {S // d}x{S // d}  block {tot * 1e3:.0f} us")
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
            n = v["launches"]
            print(f"   {k:14s} {v['ms'] / n * 1e3:8.1f} us/launch x{n // 5}  {v['bytes'] / v['ms'] / 1e6:7.0f} GB/s  {v['flops'] / v['ms'] / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
