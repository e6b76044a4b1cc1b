"""Run the inference block a few times at one shape (a target for rocprofv3): run_block.py C heads H [reps] [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from superresolution_amd import ops  # noqa: E402
from superresolution_amd.model.Restromer import TransformerBlock  # noqa: E402

C, heads, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
B = int(sys.argv[5]) if len(sys.argv) > 5 else 16
torch.manual_seed(0)
blk = TransformerBlock(C, heads, 1.25, False, "WithBias").cuda()
x = torch.rand(B, C, H, H, device="cuda")
par = blk._params()
for _ in range(reps):
    ops.block_forward_inplace(x, heads, blk.ffn.hidden_features, par)
torch.cuda.synchronize()
print("done", float(x.abs().mean()))
