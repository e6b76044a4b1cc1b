"""Multi-process paths on CPU: world_size 2 over gloo.

The distributed logic (gradient buckets + all-reduce, tile sharding + stitching) is
independent of the HIP kernels, so it is exercised here with small stand-in
modules made of stock torch layers."""
import datetime
import os
import socket

import pytest

from dist_util import spawn_bounded
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

from superresolution_amd.parallel import DataParallelTrainer, tile_grid, tiled_inference


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _TinySR(nn.Module):
    """x2 super-resolution stand-in with a small receptive field (<= 3 px)."""

    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(3, 8, 3, padding=1)
        self.b = nn.Conv2d(8, 12, 3, padding=1)

    def forward(self, x):
        y = F.pixel_shuffle(self.b(F.gelu(self.a(x))), 2)
        return y + F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    torch.set_num_threads(1)


def _dp_worker(rank, world, port, out_path, overlap):
    _init(rank, world, port)
    torch.manual_seed(0)
    net = _TinySR()
    if rank == 1:      # start from different weights: the trainer must broadcast rank 0's
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    trainer = DataParallelTrainer(net, bucket_mb=0.001, overlap=overlap)        # tiny buckets -> several all-reduces
    assert len(trainer.buckets) > 1
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 3, 8, 8, generator=g)
    t = torch.rand(4, 3, 16, 16, generator=g)
    xs, ts = x[rank * 2:(rank + 1) * 2], t[rank * 2:(rank + 1) * 2]
    losses = [trainer.step(xs, ts).item() for _ in range(3)]
    if rank == 0:
        torch.save({"sd": net.state_dict(), "losses": losses}, out_path)
    # parameters identical on both ranks
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    assert torch.equal(other[0], other[1])
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_data_parallel_matches_single_process(tmp_path, overlap):
    """2 ranks x batch 2 == 1 process x batch 4 (L1 mean => averaged gradients), incl. clip + Adam; with the bucket
    all-reduces after the backward pass (the default) and started from inside it."""
    out = str(tmp_path / "dp.pt")
    spawn_bounded(_dp_worker, (2, _free_port(), out, overlap), 2)
    got = torch.load(out, weights_only=True)
    torch.manual_seed(0)
    net = _TinySR()
    trainer = DataParallelTrainer(net, world_size=1)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 3, 8, 8, generator=g)
    t = torch.rand(4, 3, 16, 16, generator=g)
    for _ in range(3):
        trainer.step(x, t)
    for k, v in net.state_dict().items():
        assert torch.allclose(got["sd"][k], v, atol=2e-6, rtol=1e-5), k


def test_trainer_step_order_matches_reference_loop():
    """world 1: same numbers as the reference's hand-written loop (model/train.py:82-90)."""
    torch.manual_seed(1)
    a, b = _TinySR(), _TinySR()
    b.load_state_dict(a.state_dict())
    x, t = torch.rand(2, 3, 8, 8), torch.rand(2, 3, 16, 16)
    tr = DataParallelTrainer(a, world_size=1)
    opt = torch.optim.Adam(b.parameters(), lr=1e-4, weight_decay=1e-6)
    for _ in range(3):
        la = tr.step(x, t)
        opt.zero_grad()
        lb = F.l1_loss(b(x), t)
        lb.backward()
        torch.nn.utils.clip_grad_norm_(b.parameters(), max_norm=1.0)
        opt.step()
        assert abs(la.item() - lb.item()) < 1e-7
    for (k, p), q in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.allclose(p, q, atol=1e-7), k


def _tile_worker(rank, world, port, out_path):
    _init(rank, world, port)
    torch.manual_seed(0)
    net = _TinySR().eval()
    img = torch.rand(1, 3, 40, 56, generator=torch.Generator().manual_seed(2))
    res = tiled_inference(net, img, tile=16, overlap=8, rank=rank, world=world, tile_batch=3)
    if rank == 0:
        torch.save(res, out_path)
    else:
        assert res is None
    dist.destroy_process_group()


def test_tile_sharded_inference_two_ranks(tmp_path):
    out = str(tmp_path / "tiles.pt")
    spawn_bounded(_tile_worker, (2, _free_port(), out), 2)
    got = torch.load(out, weights_only=True)
    torch.manual_seed(0)
    net = _TinySR().eval()
    img = torch.rand(1, 3, 40, 56, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        full = net(img)                                   # receptive field (3 px) < overlap (8 px): tiling is exact
        single = tiled_inference(net, img, tile=16, overlap=8)
    assert torch.allclose(got, full, atol=1e-6)
    assert torch.allclose(got, single, atol=1e-6)      # batch composition differs per rank count


@pytest.mark.parametrize("h,w,tile,ov", [(2048, 2048, 256, 32), (40, 56, 16, 8), (8, 8, 256, 32), (264, 96, 64, 16)])
def test_tile_grid_covers_image_once(h, w, tile, ov):
    cover = torch.zeros(h, w, dtype=torch.int32)
    for (y0, y1, x0, x1, cy0, cy1, cx0, cx1) in tile_grid(h, w, tile, ov):
        assert y0 <= cy0 < cy1 <= y1 and x0 <= cx0 < cx1 <= x1
        assert (y1 - y0) % 8 == 0 and (x1 - x0) % 8 == 0
        assert cy0 - y0 in (0, ov) and cx0 - x0 in (0, ov)
        cover[cy0:cy1, cx0:cx1] += 1
    assert int(cover.min()) == 1 and int(cover.max()) == 1
    if (h, w, tile) == (2048, 2048, 256):
        assert len(tile_grid(h, w, tile, ov)) == 64


# ---- row-sharded frame: partition and halo exchange (the block arithmetic itself needs the GPU) ----
def test_row_partition():
    from superresolution_amd.parallel import row_partition
    assert row_partition(2160, 8)[0] == (0, 272) and row_partition(2160, 8)[-1][1] == 2160
    assert all(a % 8 == 0 and b % 8 == 0 for a, b in row_partition(2160, 8))
    assert row_partition(64, 1) == [(0, 64)]
    with pytest.raises(ValueError):
        row_partition(64, 8)           # 8 rows per rank at full resolution = 1 row at the coarsest level: too thin
    with pytest.raises(ValueError):
        row_partition(100, 2)


def _halo_worker(rank, world, port, out_path):
    from superresolution_amd import parallel
    _init(rank, world, port)
    full = torch.arange(2 * 5 * 48 * 7, dtype=torch.float32).reshape(2, 5, 48, 7)
    a, b = parallel.row_partition(48, world)[rank]
    s = parallel._Slab(full[:, :, a:b].clone(), a, b, 48)
    s.t[:, :, :s.r0] = -1.0                     # halos start as garbage
    s.t[:, :, s.r1:] = -1.0
    parallel._refresh_halo(s, rank, world, None)
    ok = torch.equal(s.t, full[:, :, s.lo:s.hi]) and (s.lo, s.hi) == (max(a - 2, 0), min(b + 2, 48))
    # second refresh after the owners changed their rows
    s.t[:, :, s.r0:s.r1] += 1000.0
    parallel._refresh_halo(s, rank, world, None)
    ok = ok and torch.equal(s.t, full[:, :, s.lo:s.hi] + 1000.0)
    flags = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(flags, torch.tensor([1.0 if ok else 0.0]))
    if rank == 0:
        torch.save([float(f) for f in flags], out_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_rows_come_from_the_neighbours(world, tmp_path):
    out = str(tmp_path / "halo.pt")
    spawn_bounded(_halo_worker, (world, _free_port(), out), world)
    assert torch.load(out) == [1.0] * world


def test_null_collectives_keep_the_bucket_path_and_send_nothing():
    """bench.py measures the exchange of a training step as (step with RCCL) - (step with NullCollectives): the
    trainer then runs its buckets, hooks, division and clipping exactly as with real peers.  With a world of one the
    numbers must be those of the plain single-process trainer."""
    from superresolution_amd.parallel import NullCollectives
    torch.manual_seed(2)
    a, b = _TinySR(), _TinySR()
    b.load_state_dict(a.state_dict())
    x, t = torch.rand(2, 3, 8, 8), torch.rand(2, 3, 16, 16)
    ta = DataParallelTrainer(a, world_size=1, comm=NullCollectives(), bucket_mb=0.001)
    assert ta.reduce and len(ta.buckets) > 1
    tb = DataParallelTrainer(b, world_size=1)
    for _ in range(3):
        la, lb = ta.step(x, t), tb.step(x, t)
        assert abs(la.item() - lb.item()) < 1e-7
    for (k, p), q in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.allclose(p, q, atol=1e-7), k
