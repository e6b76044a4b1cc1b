"""Exact row-sharded inference of one frame: every rank runs its row slab (+2 halo rows), the attention's sums over
all pixels are all-reduced inside every block.  The result must equal the network on the whole frame."""
import datetime
import os
import socket
import sys

import pytest

from dist_util import spawn_bounded
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {   # name: (dim, H, W)  -- dim 16 runs the separate kernels, dim 48 the fused 1x1 + depthwise ones (W >= 64)
    "dim16": (16, 64, 96),
    "dim48": (48, 64, 128),
}


def _net(dim):
    from superresolution_amd.model.Restromer import Restormer
    torch.manual_seed(3)
    net = Restormer(inp_channels=3, out_channels=3, dim=dim, num_blocks=[1, 2, 1, 1], num_refinement_blocks=1,
                    heads=[1, 2, 4, 8], bias=(dim == 16), LayerNorm_type="WithBias" if dim == 48 else "BiasFree")
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("temperature") or ".body.weight" in name:
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
    return net.cuda().eval()


def _image(h, w):
    return torch.rand(1, 3, h, w, generator=torch.Generator().manual_seed(9)).cuda()


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize("name", list(CASES))
def test_one_rank_equals_whole_frame(name):
    from superresolution_amd.parallel import sharded_frame_inference
    dim, h, w = CASES[name]
    net, x = _net(dim), _image(h, w)
    with torch.no_grad():
        ref = net(x)
    got = sharded_frame_inference(net, x, 0, 1)
    assert got.shape == ref.shape
    assert _rel(got, ref) < 2e-6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("C,heads,d,world", [(96, 1, 1, 3), (96, 1, 1, 2), (48, 1, 1, 3), (96, 2, 2, 3), (192, 4, 4, 2)])
def test_slab_kernels_in_one_process(C, heads, d, world):
    """The two C-ABI phases of a block on every rank's slab (own rows + halo: 18/26/28 rows at H=64, world 3), run
    one after the other in ONE process with the sums added by hand, against the block on the whole frame: the
    kernels of the sharded path at exactly the slab shapes of the multi-process test, without any collective."""
    from superresolution_amd import ops
    from superresolution_amd.model.Restromer import TransformerBlock
    from superresolution_amd.parallel import HALO, row_partition
    H, W = 64 // d, 128 // d                       # level with down-sampling factor d of a 64 x 128 frame
    torch.manual_seed(5)
    blk = TransformerBlock(C, heads, 1.25, False, "WithBias").cuda().eval()
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if n.endswith("temperature") or ".body." in n:
                p.copy_(torch.rand_like(p) + 0.5)
    x = torch.randn(1, C, H, W, generator=torch.Generator().manual_seed(6)).cuda()
    with torch.no_grad():
        ref = blk(x)
    par, hidden = blk._params(), blk.ffn.hidden_features
    slabs, total = [], None
    for a, b in ((a0 // d, b0 // d) for a0, b0 in row_partition(64, world)):
        lo, hi = max(a - HALO, 0), min(b + HALO, H)
        t = x[:, :, lo:hi].contiguous()
        st = ops.block_shard_stats(t, a - lo, b - lo, heads, hidden, par).clone()
        total = st if total is None else total + st
        slabs.append((a, b, lo, t))
    for a, b, lo, t in slabs:
        # phase 1 again: phase 2 expects this slab's qkv in the shared workspace
        ops.block_shard_stats(t, a - lo, b - lo, heads, hidden, par)
        y = ops.block_shard_apply(t, total, heads, hidden, par)
        assert _rel(y[:, :, a - lo:b - lo], ref[:, :, a:b]) < 2e-6, (a, b)


def _worker(rank, world, port, name, out_path):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=False)      # a stuck rank shows where, before the parent's deadline
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))      # the ranks share the one card: no RCCL here
    from dist_util import HostStagedCollectives
    from superresolution_amd.parallel import sharded_frame_inference
    dim, h, w = CASES[name]
    net, x = _net(dim), _image(h, w)
    out = sharded_frame_inference(net, x, rank, world, comm=HostStagedCollectives())
    if rank == 0:
        with torch.no_grad():
            ref = net(x)
        torch.save({"err": _rel(out, ref), "shape": tuple(out.shape), "ref_shape": tuple(ref.shape)}, out_path)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", list(CASES))
def test_ranks_on_one_card_equal_whole_frame(name, world, tmp_path):
    out_path = str(tmp_path / "r.pt")
    spawn_bounded(_worker, (world, _free_port(), name, out_path), world, log_dir=tmp_path)
    r = torch.load(out_path)
    assert r["shape"] == r["ref_shape"]
    assert r["err"] < 2e-6, r
