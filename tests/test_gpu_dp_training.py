"""Data-parallel training of the HIP network: two ranks (sharing the one card, gloo) with batch 2 each must walk
the same trajectory as one process with batch 4 (L1 mean => averaged gradients; clip and Adam after the all-reduce)."""
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


def _net():
    from superresolution_amd.model.Restromer import Restormer
    torch.manual_seed(11)
    return Restormer(inp_channels=3, out_channels=3, dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1,
                     heads=[1, 2, 4, 8]).cuda()


def _data():
    g = torch.Generator().manual_seed(5)
    return torch.rand(4, 3, 32, 32, generator=g).cuda(), torch.rand(4, 3, 64, 64, generator=g).cuda()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=False)      # a stuck rank shows where, before the parent's deadline
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from dist_util import HostStagedCollectives
    from superresolution_amd.parallel import DataParallelTrainer
    net = _net()
    trainer = DataParallelTrainer(net, bucket_mb=0.05, comm=HostStagedCollectives())          # several buckets
    assert len(trainer.buckets) > 1
    x, t = _data()
    xs, ts = x[rank * 2:(rank + 1) * 2], t[rank * 2:(rank + 1) * 2]
    losses = [float(trainer.step(xs, ts)) for _ in range(3)]
    if rank == 0:
        torch.save({"sd": {k: v.cpu() for k, v in net.state_dict().items()}, "losses": losses}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_one_process(tmp_path):
    from superresolution_amd.parallel import DataParallelTrainer
    out = str(tmp_path / "dp.pt")
    spawn_bounded(_worker, (2, _free_port(), out), 2, log_dir=tmp_path)
    got = torch.load(out, weights_only=True)
    net = _net()
    trainer = DataParallelTrainer(net, world_size=1)
    x, t = _data()
    ref_losses = [float(trainer.step(x, t)) for _ in range(3)]
    # per-rank losses are means over half the batch: compare the trajectory through the parameters
    worst = 0.0
    for k, v in net.state_dict().items():
        a, b = got["sd"][k].double(), v.detach().cpu().double()
        worst = max(worst, float((a - b).abs().max() / b.abs().max().clamp_min(1e-6)))
    assert worst < 1e-4, worst
    assert abs(got["losses"][0] - ref_losses[0]) < 0.2      # same ballpark: different halves of the same batch
