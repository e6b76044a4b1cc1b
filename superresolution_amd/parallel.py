"""Multi-GPU use of the network: one process per GPU over torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" on CPU for tests).

* :class:`DataParallelTrainer` -- the reference's training step
  (model/train.py:82-90: zero_grad, forward, L1, backward, clip_grad_norm_(1.0),
  Adam) with one exchange per step: the gradients are averaged over ranks by
  bucketed all-reduces issued while the backward pass is still running.
* :func:`tiled_inference` -- full-image inference cut into overlapping tiles
  that are dealt round-robin to the ranks (independent units, no reduction),
  then stitched on rank 0.  The tiling scheme follows the reference's
  overlapping-partition test loop (model/basicsr/models/mambairv2_model.py:13-106).

Nothing here depends on the HIP kernels: both classes drive whatever
``nn.Module`` they are given, so the distributed logic is testable on CPU.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# data-parallel training
# ----------------------------------------------------------------------------
class _Bucket:
    def __init__(self, params: List[nn.Parameter], device, dtype):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.zeros(self.numel, device=device, dtype=dtype)
        self.views = []
        off = 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.pending = len(params)
        self.handle = None


class DataParallelTrainer:
    """Synchronous data parallelism with gradient buckets.

    Parameters are grouped, in reverse registration order (the order in which the
    backward pass produces their gradients), into buckets of about
    ``bucket_mb`` megabytes.  Every parameter's ``.grad`` is a view into its
    bucket's flat buffer, so autograd writes gradients straight into the
    communication buffer; when the last gradient of a bucket has been produced an
    asynchronous all-reduce of that buffer starts, overlapping the rest of the
    backward pass.  On xGMI a ring all-reduce is bound by one link (about
    153 GB/s), so a few large buckets beat many small ones: the 73.7 MB of fp32
    gradients of the dim=48 network go out as three ~25 MB messages.

    After the last bucket has been reduced the gradients are divided by the world
    size (L1 is a mean over the local batch, so the result equals the gradient of
    the mean over the global batch), the global norm is clipped and Adam steps --
    identically on every rank, so parameters stay in sync without a broadcast.
    """

    def __init__(self, model: nn.Module, world_size: Optional[int] = None, lr: float = 1e-4, weight_decay: float = 1e-6,
                 max_grad_norm: float = 1.0, bucket_mb: float = 25.0,
                 loss_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor] = F.l1_loss,
                 process_group=None):
        self.model = model
        self.group = process_group
        self.world = world_size if world_size is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        self.max_grad_norm = max_grad_norm
        self.loss_fn = loss_fn
        params = [p for p in model.parameters() if p.requires_grad]
        self.params = params
        dev, dt = params[0].device, params[0].dtype
        use_fused = dev.type == "cuda"
        self.optimizer = torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, fused=use_fused)
        # buckets in reverse registration order
        self.buckets: List[_Bucket] = []
        cap = int(bucket_mb * 1e6 / params[0].element_size())
        cur: List[nn.Parameter] = []
        n = 0
        for p in reversed(params):
            cur.append(p)
            n += p.numel()
            if n >= cap:
                self.buckets.append(_Bucket(cur, dev, dt))
                cur, n = [], 0
        if cur:
            self.buckets.append(_Bucket(cur, dev, dt))
        self._owner = {}
        for b in self.buckets:
            for i, (p, v) in enumerate(zip(b.params, b.views)):
                p.grad = v
                self._owner[p] = (b, i)
        if self.world > 1:
            for p in params:
                p.register_post_accumulate_grad_hook(self._on_grad_ready)
        self.broadcast_parameters()

    def broadcast_parameters(self):
        """Make every rank start from rank 0's parameters."""
        if self.world > 1:
            for p in self.model.parameters():
                dist.broadcast(p.data, src=0, group=self.group)

    def _on_grad_ready(self, p: nn.Parameter):
        b, i = self._owner[p]
        v = b.views[i]
        if p.grad.data_ptr() != v.data_ptr():
            # autograd replaced the view (first accumulation into a None grad): copy in and restore the view
            v.copy_(p.grad)
            p.grad = v
        b.pending -= 1
        if b.pending == 0:
            b.handle = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def zero_grad(self):
        for b in self.buckets:
            b.flat.zero_()
            b.pending = len(b.params)
            b.handle = None

    def step(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """One optimiser step on this rank's batch; returns the local loss (detached)."""
        self.zero_grad()
        out = self.model(x)
        loss = self.loss_fn(out, target)
        loss.backward()
        if self.world > 1:
            for b in self.buckets:
                if b.handle is None:      # a parameter without gradient this step kept the bucket open
                    b.handle = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for b in self.buckets:
                b.handle.wait()
                b.flat.div_(self.world)
        torch.nn.utils.clip_grad_norm_(self.params, max_norm=self.max_grad_norm)
        self.optimizer.step()
        return loss.detach()


# ----------------------------------------------------------------------------
# tile-sharded inference
# ----------------------------------------------------------------------------
def tile_grid(h: int, w: int, tile: int, overlap: int, multiple: int = 8) -> List[Tuple[int, int, int, int, int, int, int, int]]:
    """Cut an h x w image into cores of `tile` pixels with `overlap` pixels of
    context on every side (clipped at the image border).

    Returns (y0, y1, x0, x1, cy0, cy1, cx0, cx1) per tile: the input window
    [y0:y1, x0:x1] and the core [cy0:cy1, cx0:cx1] (image coordinates) it is
    responsible for.  Window sizes are multiples of `multiple` whenever h, w
    are, because windows are grown towards the inside of the image if needed."""
    if tile % multiple or overlap % multiple:
        raise ValueError(f"tile={tile} and overlap={overlap} must be multiples of {multiple}")
    out = []
    for cy0 in range(0, h, tile):
        cy1 = min(cy0 + tile, h)
        y0, y1 = max(cy0 - overlap, 0), min(cy1 + overlap, h)
        for cx0 in range(0, w, tile):
            cx1 = min(cx0 + tile, w)
            x0, x1 = max(cx0 - overlap, 0), min(cx1 + overlap, w)
            out.append((y0, y1, x0, x1, cy0, cy1, cx0, cx1))
    return out


@torch.no_grad()
def tiled_inference(model: nn.Module, image: torch.Tensor, tile: int = 256, overlap: int = 32, scale: int = 2,
                    rank: int = 0, world: int = 1, tile_batch: int = 8, process_group=None,
                    gather: bool = True) -> Optional[torch.Tensor]:
    """x`scale` inference of a 1 x C x H x W image in overlapping tiles.

    Tile t goes to rank t % world.  Each rank batches its same-shaped tiles
    (up to `tile_batch` per forward), keeps the cores of the outputs, and the
    cores are stitched into the full output on rank 0 (returned there; other
    ranks return None).  With world == 1 no collective is used.

    The result is the network applied per tile, which is NOT the untiled result:
    the channel attention normalises over the pixels of a tile."""
    if image.dim() != 4 or image.shape[0] != 1:
        raise ValueError("tiled_inference expects a 1 x C x H x W image")
    _, c, h, w = image.shape
    tiles = tile_grid(h, w, tile, overlap)
    mine = [(i, t) for i, t in enumerate(tiles) if i % world == rank]
    # group by window shape so tiles can share a forward pass
    by_shape = {}
    for i, t in mine:
        by_shape.setdefault((t[1] - t[0], t[3] - t[2]), []).append((i, t))
    results = {}
    for _, group in by_shape.items():
        for s in range(0, len(group), tile_batch):
            chunk = group[s:s + tile_batch]
            batch = torch.cat([image[:, :, t[0]:t[1], t[2]:t[3]] for _, t in chunk], 0).contiguous()
            out = model(batch)
            for j, (i, t) in enumerate(chunk):
                y0, _, x0, _, cy0, cy1, cx0, cx1 = t
                results[i] = out[j:j + 1, :, (cy0 - y0) * scale:(cy1 - y0) * scale,
                                 (cx0 - x0) * scale:(cx1 - x0) * scale].contiguous()
    c_out = next(iter(results.values())).shape[1] if results else None

    def place(canvas, i, core):
        _, _, _, _, cy0, cy1, cx0, cx1 = tiles[i]
        canvas[:, :, cy0 * scale:cy1 * scale, cx0 * scale:cx1 * scale] = core

    if world == 1:
        canvas = torch.empty((1, c_out, h * scale, w * scale), dtype=image.dtype, device=image.device)
        for i, core in results.items():
            place(canvas, i, core)
        return canvas
    if not gather:
        return None
    # every rank sends its cores to rank 0 as one flat message (sizes are known from the grid)
    flat = torch.cat([results[i].reshape(-1) for i in sorted(results)]) if results else image.new_empty(0)
    co = torch.tensor([c_out if c_out is not None else 0], device=image.device)
    dist.all_reduce(co, op=dist.ReduceOp.MAX, group=process_group)
    c_out = int(co.item())
    sizes = []
    for r in range(world):
        n = 0
        for i, t in enumerate(tiles):
            if i % world == r:
                n += c_out * (t[5] - t[4]) * scale * (t[7] - t[6]) * scale
        sizes.append(n)
    if rank == 0:
        bufs = [torch.empty(n, dtype=image.dtype, device=image.device) for n in sizes]
        bufs[0] = flat
        reqs = [dist.irecv(bufs[r], src=r, group=process_group) for r in range(1, world) if sizes[r] > 0]
        for q in reqs:
            q.wait()
        canvas = torch.empty((1, c_out, h * scale, w * scale), dtype=image.dtype, device=image.device)
        for r in range(world):
            off = 0
            for i, t in enumerate(tiles):
                if i % world != r:
                    continue
                ch, cw = (t[5] - t[4]) * scale, (t[7] - t[6]) * scale
                n = c_out * ch * cw
                place(canvas, i, bufs[r][off:off + n].view(1, c_out, ch, cw))
                off += n
        return canvas
    if sizes[rank] > 0:
        dist.send(flat, dst=0, group=process_group)
    return None
