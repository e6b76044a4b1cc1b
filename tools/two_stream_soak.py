#!/usr/bin/env python3
"""Development probe: two half-batches on two streams against the same half-batches on one stream, bit for bit, many
times, whole network, on the bf16 path and in fp32."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

net = bench.build_model().cuda().eval()
net.two_streams_min_pixels = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
with torch.no_grad():
    for dtype in (torch.bfloat16, torch.float32):
        for shape in ((5, 3, 32, 256), (6, 3, 64, 256), (16, 3, 256, 256)):
            x = torch.rand(*shape, generator=torch.Generator().manual_seed(7)).cuda().to(dtype)
            net.two_streams = False
            h = x.chunk(2)
            ref = torch.cat([net(h[0]), net(h[1])])
            ref = torch.cat([net(h[0]), net(h[1])])
            net.two_streams = True
            net(x)
            ds = [float((net(x).float() - ref.float()).abs().max()) for _ in range(reps)]
            torch.cuda.synchronize()
            print(f"{str(dtype):15s} {shape}: differing {sum(1 for d in ds if d > 0)}/{reps}, max {max(ds):.2e}")
