"""The RCCL (backend "nccl") branches of the multi-GPU paths, executed on the one card of the test box: a group of
ONE rank whose collectives are issued anyway (`Collectives(eager=True)`), so the calls that eight GPUs will make --
asynchronous bucketed all-reduce from the backward hooks, the broadcast of the parameters, the all-reduce of the
attention sums inside every block of the row-sharded frame, device point-to-point -- have run on device buffers
before the first multi-GPU launch.  Also bench.py under torch.distributed.run with one rank."""
import json
import os
import subprocess
import sys

import pytest
import torch

from dist_util import free_port, spawn_bounded

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, port, out_path):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=False)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from superresolution_amd.model.Restromer import Restormer
    from superresolution_amd.parallel import Collectives, DataParallelTrainer, sharded_frame_inference, tiled_inference
    comm = Collectives(eager=True)
    res = {}
    # device all-reduce, broadcast and point-to-point (to itself) through RCCL
    t = torch.arange(8, dtype=torch.float32, device="cuda")
    comm.all_reduce(t)
    comm.broadcast(t)
    recv = torch.empty_like(t)
    comm.exchange([(t, recv, 0)])
    res["p2p"] = bool(torch.equal(recv, torch.arange(8, dtype=torch.float32, device="cuda")))
    # data-parallel step: hooks issue the bucket all-reduces during the backward pass
    torch.manual_seed(11)
    net = Restormer(inp_channels=3, out_channels=3, dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, heads=[1, 2, 4, 8]).cuda()
    twin = Restormer(inp_channels=3, out_channels=3, dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, heads=[1, 2, 4, 8]).cuda()
    twin.load_state_dict(net.state_dict())
    g = torch.Generator().manual_seed(5)
    x, tgt = torch.rand(2, 3, 32, 32, generator=g).cuda(), torch.rand(2, 3, 64, 64, generator=g).cuda()
    tr = DataParallelTrainer(net, bucket_mb=0.05, comm=comm)
    plain = DataParallelTrainer(twin, world_size=1)
    assert tr.reduce and len(tr.buckets) > 1 and not plain.reduce
    for _ in range(2):
        la, lb = tr.step(x, tgt), plain.step(x, tgt)
    res["dp_loss"] = (float(la), float(lb))
    res["dp_param_diff"] = max(float((p - q).abs().max()) for p, q in zip(net.parameters(), twin.parameters()))
    # row-sharded frame and tiled inference with the collectives live
    net.eval()
    img = torch.rand(1, 3, 32, 64, generator=g).cuda()
    with torch.no_grad():
        ref = net(img)
        fr = sharded_frame_inference(net, img, 0, 1, comm=comm)
        tl = tiled_inference(net, img, tile=32, overlap=8, comm=comm)
        tl_ref = tiled_inference(net, img, tile=32, overlap=8)
    res["frame_err"] = float((fr - ref).abs().max() / ref.abs().max())
    # (not bitwise: the dense 3x3 convolutions are MIOpen's, whose first call on a shape may pick another algorithm)
    res["tiled_err"] = float((tl - tl_ref).abs().max() / tl_ref.abs().max())
    torch.save(res, out_path)
    dist.destroy_process_group()


def test_rccl_paths_run_on_one_gpu(tmp_path):
    out = str(tmp_path / "r.pt")
    spawn_bounded(_worker, (free_port(), out), 1)
    r = torch.load(out)
    assert r["p2p"]
    assert abs(r["dp_loss"][0] - r["dp_loss"][1]) < 1e-6 and r["dp_param_diff"] < 1e-6, r
    assert r["frame_err"] < 2e-6 and r["tiled_err"] < 2e-6, r


@pytest.mark.parametrize("mode,extra", [("train", ["--batch", "2", "--size", "64"]),
                                        ("frame_sharded", ["--size", "128"]),
                                        ("tiled", ["--size", "512", "--batch", "4"])])
def test_bench_under_torchrun_one_rank(mode, extra):
    """bench.py launched the way the driver launches it for N > 1, with one rank: RCCL initialised, collectives issued."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--mode", mode, "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and "roofline" in d
