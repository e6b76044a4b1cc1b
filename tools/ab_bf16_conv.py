"""PSNR distance of the bf16 network from the fp64 oracle on the 272 x 480 frame of tests/test_gpu_frames.py, with the
dense 3x3 convolutions on the packed kernel and through torch (RST_BF16_CONV=torch) (development aid)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import restormer_oracle as orc  # noqa: E402


def psnr(a, b):
    return 10 * math.log10(1.0 / float(((a - b) ** 2).mean()))


net = bench.build_model(0).cuda().eval()
x = torch.rand(1, 3, 272, 480, generator=torch.Generator().manual_seed(21))
sd64 = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}
with torch.no_grad():
    ref = orc.restormer_forward(x.double(), sd64)
target = (ref + 0.03 * torch.randn(ref.shape, generator=torch.Generator().manual_seed(22), dtype=torch.float64)).clamp(0, 1)
for route in ("packed", "torch"):
    if route == "torch":
        os.environ["RST_BF16_CONV"] = "torch"
    with torch.no_grad():
        y = net(x.cuda().bfloat16()).double().cpu()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y_ac = net(x.cuda()).double().cpu()
    for name, o in (("bf16 in/out", y), ("autocast, fp32 out", y_ac)):
        print(f"{route:7s} {name:20s} rms err {float(((o - ref) ** 2).mean()) ** 0.5:.3e}  max rel {float((o - ref).abs().max() / ref.abs().max()):.3e}  "
              f"dPSNR {psnr(o, target) - psnr(ref, target):+.5f} dB")
