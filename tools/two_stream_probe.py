#!/usr/bin/env python3
"""Development probe: a batch as one launch sequence against two half-batches on two streams (kernels that wait for
issue slots overlap with kernels that wait for memory), over batch sizes and image sizes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

net = bench.build_model().cuda().eval()
side = torch.cuda.Stream()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def two(x):
    a, b = x.chunk(2)
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        ob = net(b)
    oa = net(a)
    cur.wait_stream(side)
    ob.record_stream(cur)
    return torch.cat([oa, ob])


with torch.no_grad():
    for dtype in (torch.float32, torch.bfloat16):
        for B, S in ((2, 256), (4, 256), (8, 256), (16, 256), (32, 256), (2, 512), (4, 512), (16, 128), (64, 128)):
            x = torch.rand(B, 3, S, S, device="cuda").to(dtype)
            t1 = timeit(lambda: net(x))
            t2 = timeit(lambda: two(x))
            print(f"{str(dtype):15s} {B:3d} x {S}^2: one stream {t1:7.2f} ms, two streams {t2:7.2f} ms ({t2 / t1:.3f})")
