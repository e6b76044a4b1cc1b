"""PSNR of the bf16 storage path against a synthetic target, next to the fp32 path's, for a few image shapes and the tiled
route: how far the 0.01 dB end-to-end criterion is from what the path delivers (run on the GPU box:
python tools/bf16_psnr_margin.py; RST_BF16_RING=0 / RST_BF16_FUSE=0 select the other kernel routes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from superresolution_amd.parallel import tiled_inference
def psnr(a, b): return float(10 * torch.log10(1.0 / ((a.double() - b.double()) ** 2).mean()))
net = bench.build_model(0).cuda().eval()
for seed, shape in ((9, (1, 3, 96, 128)), (1, (1, 3, 128, 128)), (2, (2, 3, 64, 192)), (3, (1, 3, 256, 384))):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(*shape, generator=g).cuda()
    with torch.no_grad():
        ref = net(img)
        out = net(img.bfloat16()).float()
    target = (ref + 0.03 * torch.randn(ref.shape, generator=g).cuda()).clamp(0, 1)
    print(shape, "whole: dPSNR %.5f" % (psnr(out, target) - psnr(ref, target)), "rms err %.3e" % float(((out - ref) ** 2).mean().sqrt()))
g = torch.Generator().manual_seed(9)
img = torch.rand(1, 3, 96, 128, generator=g).cuda()
with torch.no_grad():
    ref = tiled_inference(net, img, tile=64, overlap=16, tile_batch=2)
    out = tiled_inference(net, img.bfloat16(), tile=64, overlap=16, tile_batch=2).float()
target = (ref + 0.03 * torch.randn(ref.shape, generator=g).cuda()).clamp(0, 1)
print("tiled: dPSNR %.5f" % (psnr(out, target) - psnr(ref, target)), "rms err %.3e" % float(((out - ref) ** 2).mean().sqrt()))
