"""How long the host needs to QUEUE one bf16 training step against how long the GPU needs to run it: if the two are close
the step is bound by the launches, not by the kernels.  python tools/train_host_time.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from superresolution_amd.model.Restromer import Restormer
from superresolution_amd.parallel import Collectives, DataParallelTrainer


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    torch.manual_seed(0)
    net = Restormer(inp_channels=3, out_channels=3, dim=48).cuda()
    trainer = DataParallelTrainer(net, world_size=1, comm=Collectives(eager=False))
    x = torch.rand(8, 3, 128, 128, device="cuda")
    tgt = torch.rand(8, 3, 256, 256, device="cuda")

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return trainer.step(x, tgt)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    queued, total = 0.0, 0.0
    for _ in range(steps):
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        queued += t1 - t0
        total += t2 - t0
    print(f"host queues a step in {queued / steps * 1e3:.2f} ms; step done after {total / steps * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
