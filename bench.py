#!/usr/bin/env python3
"""Restormer throughput on MI355X: input megapixels per second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode infer|train|tiled|frame|frame_sharded]

Default workload (one GPU): Restormer(inp_channels=3, out_channels=3, dim=48), fp32
inference on a synthetic batch of 16 x 3 x 256 x 256 (the configuration the
metric is quoted on); a step is one forward pass over the batch, inputs resident
in HBM.  With N > 1 (launched by torch.distributed.run, one rank per GPU) every
rank runs the same-size batch of its own (independent samples: no data-path
collective), so the scaling is weak; time is the maximum over ranks between two
barriers.  ``--mode train`` times the reference's training step instead
(forward, L1, backward, gradient all-reduce over RCCL, clip, Adam) at 8 x 3 x 128 x 128
per GPU.  ``--mode tiled`` is one 1 x 3 x 2048 x 2048 image in 256-pixel tiles with 32 pixels of overlap,
the tiles dealt round-robin to the ranks and the cores gathered on rank 0 (total work fixed: strong scaling);
``--mode frame`` is one untiled 1 x 3 x 2160 x 3840 frame per GPU; ``--mode frame_sharded`` is ONE such frame whose
rows are split over the GPUs, exact (per block one all-reduce of the attention's sums, halo rows point-to-point).

Rank 0 prints one JSON line.  ``roofline`` is measured by a separate pass with the
library's per-kernel event timers (HIP events on the launch stream around every
kernel) and reports the kernel class that takes the most time; ``cpu_baseline``
times the CPU oracle (a port of the reference's arithmetic) on the host cores on
a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3    # MI355X v_mfma_f32_* dense peak
PEAK_F16_MATRIX_TFLOPS = 2500.0    # MI355X v_mfma_f32_16x16x32_{f16,bf16} dense peak
PEAK_HBM_GBS = 8000.0              # MI355X HBM3E peak


def matrix_peak_tflops(name: str) -> float:
    """The matrix-pipe roof a kernel class is held against: the exact-fp32 classes run v_mfma_f32_16x16x4_f32; a
    split-f16 class (*_f16s*) spends three f16 instructions per fp32 product, so its roof in fp32-equivalent flops is
    a third of the f16 peak; the packed bf16 classes (pk_*) run one bf16 instruction per product."""
    if "f16s" in name:
        return PEAK_F16_MATRIX_TFLOPS / 3.0
    if name.startswith("pk_") or name.endswith("_f16"):
        return PEAK_F16_MATRIX_TFLOPS
    return PEAK_FP32_MATRIX_TFLOPS



def kernel_sources_hash() -> str:
    """sha256 over the kernel sources the library is built from: what a committed counter measurement is valid for
    (the GPU box has no git history, so the stamp cannot be a commit id)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "superresolution_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.hpp")) + [os.path.join(base, "Makefile")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def measured_peaks():
    """What tools/peaks.py measured on an MI355X (profiles/r*_peaks.json, newest), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_peaks.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
    except (OSError, ValueError):
        return None
    return {k: round(v, 1) for k, v in d.items() if isinstance(v, float)} | {"file": os.path.relpath(files[-1], ROOT)}


def build_model(seed: int = 0):
    """Default-initialised network under a fixed seed, then non-trivial LayerNorm
    scale/shift and attention temperature so those paths carry signal."""
    from superresolution_amd.model.Restromer import Restormer
    torch.manual_seed(seed)
    net = Restormer(inp_channels=3, out_channels=3, dim=48)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("temperature"):
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            elif ".body.weight" in name:
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            elif ".body.bias" in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    return net


def dist_setup(n_gpus: int):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or "RANK" in os.environ:
        # launched by torch.distributed.run (also with one rank: the RCCL paths then run as they do with eight)
        import torch.distributed as dist
        torch.cuda.set_device(local)
        # RCCL prints a version banner on stdout when its first communicator comes up: create it here with stdout
        # pointed at stderr, so that the JSON line stays the only thing this program writes to stdout
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
            dist.all_reduce(torch.zeros(1, device=torch.device("cuda", local)))
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
    else:
        torch.cuda.set_device(0)
    if world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}: launch with "
                         f"python -m torch.distributed.run --nproc-per-node {n_gpus} bench.py --gpus {n_gpus} ...")
    return rank, world, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x: float, world: int) -> float:
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def kernel_profile(run_once):
    """One pass of the workload with the library's per-kernel event timers on."""
    from superresolution_amd import _lib
    lib = _lib.load()
    torch.cuda.synchronize()
    lib.rst_profile_begin()
    run_once()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.check(lib.rst_profile_end(buf, len(buf)), "rst_profile_end")
    return json.loads(buf.value.decode())


def measured_traffic(name, batch, size, mode, dtype="f32"):
    """HBM bytes per launch of kernel group `name` from a committed `rocprofv3 --pmc` run of this same workload
    (profiles/*hbm_traffic*.json written by tools/summarize_pmc.py: FETCH_SIZE and WRITE_SIZE collected in separate
    passes, FETCH_SIZE doubled as gfx950 requires).  A file counts only if it was taken with the kernel sources of
    this tree (its `kernel_sources_sha` stamp) on this workload; otherwise None -- a stale number is worse than none."""
    import glob
    sha = kernel_sources_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic*.json")), reverse=True):
        if ("bf16" in os.path.basename(path)) != (dtype == "bf16"):
            continue
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        c = d.get("config", {})
        if d.get("kernel_sources_sha") != sha or (c.get("batch"), c.get("size"), c.get("mode")) != (batch, size, mode):
            continue
        k = d.get("kernels", {}).get(name)
        if k is not None:
            return k.get("bytes_per_launch")
    return None


def roofline_from(prof, traffic_of=lambda name: None):
    name = max(prof, key=lambda k: prof[k]["ms"])
    k = prof[name]
    sec = k["ms"] * 1e-3
    # the roof the kernel is nearer to: its algorithmic bytes against the HBM peak, or its algorithmic flops against the
    # matrix peak of the instruction its class runs on (exact fp32, split f16, bf16)
    tf, gbs = k["flops"] / sec / 1e12, k["bytes"] / sec / 1e9
    mpeak = matrix_peak_tflops(name)
    if tf / mpeak > gbs / PEAK_HBM_GBS:
        achieved = tf
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": mpeak, "unit": "TFLOP/s", "frac": round(achieved / mpeak, 4)}
    else:
        achieved = gbs
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(achieved / PEAK_HBM_GBS, 4)}
    peaks = measured_peaks()
    if peaks:
        # hbm: the best read + write stream measured on the device (c = a + b by torch's elementwise kernel where the
        # peaks file has it, else the library's copy kernel)
        key = {"mfma": "mfma_f32_16x16x4_TFLOPs", "hbm": "stream_GBps" if "stream_GBps" in peaks else "copy_GBps"}[roof["bound"]]
        if key in peaks and (roof["bound"] == "hbm" or mpeak == PEAK_FP32_MATRIX_TFLOPS):
            roof["measured_peak"] = peaks[key]
            roof["frac_of_measured_peak"] = round(achieved / peaks[key], 4)
            roof["measured_peak_source"] = peaks["file"]
    roof.update({"kernel": name, "launches_per_step": k["launches"], "avg_launch_us": round(k["ms"] * 1e3 / k["launches"], 2),
                 "traffic": traffic_of(name), "algorithmic_bytes_per_launch": round(k["bytes"] / k["launches"])})
    total = sum(v["ms"] for v in prof.values())
    breakdown = {n: {"ms": round(v["ms"], 3), "share": round(v["ms"] / total, 3),
                     "GB/s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1), "TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                     "frac_hbm": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 3),
                     "frac_matrix": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / matrix_peak_tflops(n), 3)}
                 for n, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
    return roof, breakdown, total


def cpu_baseline(net, budget_s: float = 25.0):
    """The CPU oracle (port of the reference arithmetic) on the host cores, on the
    reference's own CPU-runnable case: 4 x 3 x 128 x 128 forward, fp32, no_grad."""
    from oracle import restormer_oracle as orc
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    x = torch.rand(4, 3, 128, 128, generator=torch.Generator().manual_seed(3))
    times = []
    t_start = time.perf_counter()
    with torch.no_grad():
        ref = orc.restormer_forward(x, sd)      # warm-up
        while len(times) < 10 and (time.perf_counter() - t_start) < budget_s:
            t0 = time.perf_counter()
            orc.restormer_forward(x, sd)
            times.append(time.perf_counter() - t0)
        got = net(x.cuda()).cpu()
    times.sort()
    med = times[len(times) // 2]
    err = float((got.double() - ref.double()).abs().max() / ref.double().abs().max())
    return {"value": round(4 * 128 * 128 / med / 1e6, 4), "unit": "MP/s", "cores": cores, "kind": "port",
            "sample": f"4x3x128x128 fp32 forward (reference CPU-runnable config), median of {len(times)} runs, {med:.3f} s each",
            "gpu_vs_cpu_oracle_max_rel_err": err}


def cpu_baseline_train(net, bf16: bool, budget_s: float = 25.0):
    """The CPU oracle's forward + L1 + backward (the differentiated part of the step of model/train.py:82-90) on the
    host cores, on a bounded sample of the training workload (1 x 3 x 128 x 128), and the HIP network's loss and global
    gradient norm on the same sample and parameters against it."""
    from oracle import restormer_oracle as orc
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(1, 3, 128, 128, generator=gen)
    tgt = torch.rand(1, 3, 256, 256, generator=gen)

    def cpu_step():
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        loss = orc.l1_loss(orc.restormer_forward(x, params), tgt)
        grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
        return float(loss.detach()), float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads if g is not None)))

    ref_loss, ref_norm = cpu_step()             # warm-up
    times = []
    t_start = time.perf_counter()
    while len(times) < 5 and (time.perf_counter() - t_start) < budget_s:
        t0 = time.perf_counter()
        cpu_step()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    params = [p for p in net.parameters() if p.requires_grad]
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        loss = torch.nn.functional.l1_loss(net(x.cuda()).float(), tgt.cuda())
    grads = torch.autograd.grad(loss, params, allow_unused=True)
    norm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads if g is not None)))
    return {"value": round(128 * 128 / med / 1e6, 4), "unit": "MP/s", "cores": cores, "kind": "port",
            "sample": f"1x3x128x128 fp32 forward + L1 + backward of the CPU oracle, median of {len(times)} runs, {med:.3f} s each",
            "gpu_vs_cpu_oracle_loss_rel_err": abs(float(loss.detach()) - ref_loss) / abs(ref_loss),
            "gpu_vs_cpu_oracle_grad_norm_rel_err": abs(norm - ref_norm) / ref_norm}


def _timed(step, steps, warmup, world):
    for _ in range(warmup):
        step()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier(world)
    return max_over_ranks(time.perf_counter() - t0, world) / steps


def side_train_bf16(dev, rank, world, steps=8, warmup=3, B=8, S=128):
    """The bf16 training configuration beside the default line: the bf16 training step (forward, L1, backward, gradient all-reduce
    over RCCL, division by the world size, clip, Adam) on 8 x 3 x 128 x 128 per GPU, and the same step with the
    collectives replaced by no-ops -- the difference is what the exchange costs a step."""
    from superresolution_amd.parallel import Collectives, DataParallelTrainer, NullCollectives
    import torch.distributed as dist
    gen = torch.Generator(device=dev).manual_seed(200 + rank)
    x = torch.rand(B, 3, S, S, device=dev, generator=gen)
    tgt = torch.rand(B, 3, 2 * S, 2 * S, device=dev, generator=gen)
    res = {}
    runs = (("rccl", Collectives(eager=True)), ("no_exchange", NullCollectives(max(world, 2)))) if dist.is_initialized() \
        else (("plain", None),)          # one process without torch.distributed: no buckets, no exchange
    for name, comm in runs:
        net = build_model(0).to(dev)
        trainer = DataParallelTrainer(net, world_size=world, comm=comm)

        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return trainer.step(x, tgt)
        res[name] = _timed(step, steps, warmup, world)
        del trainer, net
    if "plain" in res:
        return {"value": round(B * S * S / res["plain"] / 1e6, 3), "unit": "MP/s", "ms_per_step": round(res["plain"] * 1e3, 3),
                "scaling": "weak", "dtype": "bf16", "steps": steps, "warmup": warmup,
                "workload": f"bf16 training step (fwd + L1 + bwd + clip + Adam), {B}x3x{S}x{S}, one process without "
                            f"torch.distributed (under torch.distributed.run this object also times the RCCL all-reduce)"}
    return {"value": round(world * B * S * S / res["rccl"] / 1e6, 3), "unit": "MP/s", "ms_per_step": round(res["rccl"] * 1e3, 3),
            "ms_per_step_without_exchange": round(res["no_exchange"] * 1e3, 3),
            "exposed_allreduce_ms": round((res["rccl"] - res["no_exchange"]) * 1e3, 3), "scaling": "weak", "dtype": "bf16",
            "workload": f"bf16 training step, {B}x3x{S}x{S} per GPU, 73.7 MB fp32 gradient all-reduce in ~25 MB buckets "
                        f"after the backward pass (RST_DP_OVERLAP=1: from inside it)", "steps": steps, "warmup": warmup}


def side_tiled_2048(net, dev, rank, world, steps=3, warmup=1):
    """The tiled-image configuration beside the default line: one 2048 x 2048 image in 256-pixel tiles + 32 pixels of overlap,
    tile t on rank t mod world, the cores gathered on rank 0 (bf16 storage path)."""
    from superresolution_amd.parallel import Collectives, tiled_inference
    img = torch.rand(1, 3, 2048, 2048, device=dev, generator=torch.Generator(device=dev).manual_seed(300)).bfloat16()
    import torch.distributed as dist
    comm = Collectives(eager=dist.is_initialized())
    net.eval()

    def step():
        with torch.no_grad():
            return tiled_inference(net, img, tile=256, overlap=32, rank=rank, world=world, tile_batch=16, comm=comm)
    t = _timed(step, steps, warmup, world)
    return {"value": round(2048 * 2048 / t / 1e6, 3), "unit": "MP/s of image pixels", "ms_per_image": round(t * 1e3, 2),
            "scaling": "strong", "dtype": "bf16", "workload": "one 1x3x2048x2048 image, 64 windows of up to 320 px dealt to the ranks, "
            "output gathered on rank 0", "steps": steps, "warmup": warmup}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["infer", "train", "tiled", "frame", "frame_sharded"], default="infer")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16: activations stored in bf16 between kernels, fp32 sums (infer, frame and tiled modes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-runs", action="store_true",
                    help="only the named workload: no bf16_storage side measurement (profiling runs)")
    args = ap.parse_args()

    if args.dtype == "bf16" and args.mode not in ("infer", "frame", "tiled", "train"):
        raise SystemExit("--dtype bf16 is available for --mode infer, --mode frame, --mode tiled and --mode train")
    rank, world, local = dist_setup(args.gpus)
    dev = torch.device("cuda", local)
    net = build_model(0).to(dev)
    prec = "fp32" if args.dtype == "f32" else "bf16-storage"

    if args.mode == "infer":
        B, S = args.batch or 16, args.size or 256
        x = torch.rand(B, 3, S, S, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + rank))
        if args.dtype == "bf16":
            x = x.bfloat16()
        net.eval()

        def step():
            with torch.no_grad():
                return net(x)
        workload = f"Restormer(3,3,dim=48) {prec} inference, {B}x3x{S}x{S} per GPU"
        metric = "restormer_inference_megapixels_per_sec"
    elif args.mode in ("tiled", "frame", "frame_sharded"):
        from superresolution_amd.parallel import Collectives, sharded_frame_inference, tiled_inference
        import torch.distributed as dist
        comm = Collectives(eager=dist.is_initialized())
        net.eval()
        if args.mode == "tiled":
            Hh = Ww = args.size or 2048
            seed = 100                       # every rank holds the same image
        else:
            Hh, Ww = (args.size, args.size) if args.size else (2160, 3840)
            seed = 100 if args.mode == "frame_sharded" else 100 + rank
        x = torch.rand(1, 3, Hh, Ww, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
        if args.dtype == "bf16":
            x = x.bfloat16()

        def step():
            with torch.no_grad():
                if args.mode == "tiled":
                    return tiled_inference(net, x, tile=256, overlap=32, rank=rank, world=world, tile_batch=args.batch or 16, comm=comm)
                if args.mode == "frame_sharded":
                    return sharded_frame_inference(net, x, rank, world, gather=True, comm=comm)
                return net(x)
        scaling = "weak" if args.mode == "frame" else "strong"
        pixels = Hh * Ww * (world if args.mode == "frame" else 1)
        workload = (f"Restormer(3,3,dim=48) {prec} tiled inference of one 1x3x{Hh}x{Ww} image, 256-px tiles + 32-px overlap, tiles sharded over the GPUs"
                    if args.mode == "tiled" else
                    f"Restormer(3,3,dim=48) fp32 exact inference of one 1x3x{Hh}x{Ww} frame, rows sharded over the GPUs, output gathered on rank 0"
                    if args.mode == "frame_sharded" else f"Restormer(3,3,dim=48) {prec} untiled inference, one 1x3x{Hh}x{Ww} frame per GPU")
        metric = "restormer_inference_megapixels_per_sec"
        B, S = 1, Hh
    else:
        from superresolution_amd.parallel import Collectives, DataParallelTrainer
        import torch.distributed as dist
        B, S = args.batch or 8, args.size or 128
        gen = torch.Generator(device=dev).manual_seed(100 + rank)
        x = torch.rand(B, 3, S, S, device=dev, generator=gen)
        tgt = torch.rand(B, 3, 2 * S, 2 * S, device=dev, generator=gen)
        trainer = DataParallelTrainer(net, world_size=world, comm=Collectives(eager=dist.is_initialized()))

        def step():
            if args.dtype == "bf16":
                # bf16 activations (kept per block: its bf16 input only), fp32 sums, fp32 master parameters and gradients
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    return trainer.step(x, tgt)
            return trainer.step(x, tgt)
        workload = f"Restormer(3,3,dim=48) {prec} training step (fwd+L1+bwd+allreduce+clip+Adam), {B}x3x{S}x{S} per GPU"
        metric = "restormer_training_megapixels_per_sec"

    for _ in range(args.warmup):
        step()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world)

    if args.mode in ("tiled", "frame", "frame_sharded"):
        mp = pixels * args.steps / 1e6
    else:
        scaling = "weak"
        mp = world * B * S * S * args.steps / 1e6
    out = {
        "metric": metric, "value": round(mp / elapsed, 3), "unit": "MP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": workload, "global_batch": world * B, "input_hw": [Hh, Ww] if args.mode in ("tiled", "frame", "frame_sharded") else [S, S], "parallelism": {"tiled": f"tiles{world}", "frame_sharded": f"rows{world}"}.get(args.mode, f"dp{world}"),
                   "weights": "default init, seed 0"},
    }
    if args.mode == "infer" and args.dtype == "f32" and not args.no_side_runs:
        # the same workload on the bf16 storage path (bf16 activations between kernels, fp32 sums), timed the same
        # way; reported beside the fp32 line, never as `value`
        xb = x.bfloat16()

        def step_bf16():
            with torch.no_grad():
                return net(xb)
        for _ in range(args.warmup):
            step_bf16()
        barrier(world)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_bf16()
        barrier(world)
        el = max_over_ranks(time.perf_counter() - t0, world)
        out["bf16_storage"] = {"value": round(world * B * S * S * args.steps / 1e6 / el, 3), "unit": "MP/s",
                               "ms_per_step": round(el / args.steps * 1e3, 3), "dtype": "bf16",
                               "note": "same workload, activations stored in bf16 between kernels, fp32 sums and parameters; "
                                       "PSNR within 0.01 dB of the fp32 result (tests/test_gpu_bf16.py)"}
        # and once more as two half-batches on two HIP streams (Restormer._forward_two_streams, opt-in)
        was = net.two_streams
        net.two_streams = True
        try:
            for _ in range(args.warmup):
                step_bf16()
            barrier(world)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_bf16()
            barrier(world)
            el2 = max_over_ranks(time.perf_counter() - t0, world)
        finally:
            net.two_streams = was
        out["bf16_storage"]["two_streams"] = {"value": round(world * B * S * S * args.steps / 1e6 / el2, 3), "unit": "MP/s",
                                              "ms_per_step": round(el2 / args.steps * 1e3, 3),
                                              "note": "opt-in (model.two_streams = True): the batch as two half-batches on two HIP streams"}
        # The default line also carries the two workloads whose data path has an exchange: the training configuration (bf16
        # training step, gradient all-reduce) and the tiled image (one 2048 x 2048 image in tiles dealt to the ranks, cores
        # gathered on rank 0).  Under torch.distributed.run they go through RCCL as they are (also with one rank), so a
        # 1/2/4/8-GPU sweep of the default command records them; in a plain one-process run they run without.  Never `value`.
        # (a failure here -- the same on every rank, e.g. out of memory -- is recorded, not fatal to the line)
        for key, fn in (("train_bf16", lambda: side_train_bf16(dev, rank, world)),
                        ("tiled_2048", lambda: side_tiled_2048(net, dev, rank, world))):
            try:
                out[key] = fn()
            except Exception as e:      # noqa: BLE001
                out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if rank == 0:
            # kernels of the training path that the inference workload does not launch, at a headline shape:
            # channel LayerNorm forward and backward (model/Restromer.py:26-52) at 16 x 96 x 256 x 256
            from superresolution_amd import ops
            xl = torch.randn(16, 96, 256, 256, device=dev).requires_grad_(True)
            wl = torch.rand(96, device=dev).add_(0.5).requires_grad_(True)
            bl = torch.randn(96, device=dev).requires_grad_(True)
            gl = torch.randn(16, 96, 256, 256, device=dev)

            def ln_step():
                xl.grad = wl.grad = bl.grad = None
                ops.layernorm(xl, wl, bl).backward(gl)
            ln_step()
            pl = kernel_profile(ln_step)
            out["side_kernels"] = {"shape": "16x96x256x256 fp32, WithBias",
                                   **{n: {"us": round(v["ms"] * 1e3, 1), "GB/s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
                                      for n, v in pl.items()}}
            del xl, gl
    if rank != 0 and args.mode in ("train", "tiled", "frame_sharded"):
        step()          # the profiled extra step on rank 0 contains collectives: every rank takes part
    if rank == 0:
        # The timed steps of the inference modes may run a batch as two half-batches on two streams
        # (Restormer._forward_two_streams).  Overlapped kernels stretch each other, so the per-kernel pass runs the
        # batch as ONE launch sequence: every kernel alone on the chip, at the full batch -- the durations that
        # rocprofv3 --kernel-trace gives with RST_TWO_STREAMS=0 and that the roofline is about.
        split = False
        if args.mode == "infer":
            with torch.no_grad():
                split = bool(net._two_streams_wanted(x))
        was, was_env = getattr(net, "two_streams", False), os.environ.get("RST_TWO_STREAMS")
        net.two_streams = False
        os.environ["RST_TWO_STREAMS"] = "0"
        try:
            if split:
                step()
            prof = kernel_profile(step)
        finally:
            net.two_streams = was
            if was_env is None:
                del os.environ["RST_TWO_STREAMS"]
            else:
                os.environ["RST_TWO_STREAMS"] = was_env
        roof, breakdown, ktotal = roofline_from(prof, lambda name: measured_traffic(name, B, S, args.mode, args.dtype))
        if split:
            roof["note"] = ("kernel timings: the batch as one launch sequence; the timed steps overlap two half-batches "
                            "on two HIP streams")
            out["config"]["streams"] = 2
        out["roofline"] = roof
        out["kernels"] = breakdown
        out["hip_kernel_ms_per_step"] = round(ktotal, 3)
        if world == 1 and not args.no_cpu_baseline and args.mode == "infer" and args.dtype == "f32":
            out["cpu_baseline"] = cpu_baseline(net)
        if world == 1 and not args.no_cpu_baseline and args.mode == "train":
            out["cpu_baseline"] = cpu_baseline_train(net, args.dtype == "bf16")
        print(json.dumps(out), flush=True)
    barrier(world)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
