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
import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# the exchanges the multi-GPU paths need
# ----------------------------------------------------------------------------
class Collectives:
    """torch.distributed as it is: every buffer goes to the backend where it lives (RCCL moves device buffers
    over xGMI directly; gloo moves host buffers).  The functions below take an object with this interface, so a
    harness with other needs (several ranks on one card, where RCCL cannot run and gloo would have to stage
    device buffers itself) passes its own."""

    def __init__(self, group=None, eager: bool = False):
        self.group = group
        # eager: issue the collectives even in a group of one rank (a single GPU then runs the very calls --
        # bucketed asynchronous all-reduce, the attention sums' all-reduce -- that eight GPUs will run)
        self.eager = bool(eager)

    def active(self, world: int) -> bool:
        return world > 1 or self.eager

    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def all_reduce(self, t: torch.Tensor, op=dist.ReduceOp.SUM, async_op: bool = False):
        """In-place reduction over the ranks; with `async_op` returns a handle whose wait() completes it."""
        return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)

    def broadcast(self, t: torch.Tensor, src: int = 0) -> None:
        dist.broadcast(t, src=src, group=self.group)

    def exchange(self, pairs) -> None:
        """pairs: (send tensor or None, receive tensor or None, peer) -- all transfers posted together, then
        waited for.  Tensors must be contiguous."""
        ops_ = []
        for send, recv, peer in pairs:
            if send is not None:
                ops_.append(dist.P2POp(dist.isend, send, peer, self.group))
            if recv is not None:
                ops_.append(dist.P2POp(dist.irecv, recv, peer, self.group))
        if ops_:
            for q in dist.batch_isend_irecv(ops_):
                q.wait()


class NullCollectives(Collectives):
    """The same call pattern with nothing sent: a trainer built on it keeps its buckets, hooks and divisions and skips
    only the exchange.  For measuring what the collectives of a step cost (bench.py: step time with RCCL minus step
    time with this)."""

    class _Done:
        def wait(self):
            return True

    def __init__(self, world: int = 2):
        super().__init__(None, eager=True)
        self._world = world

    def world_size(self) -> int:
        return self._world

    def all_reduce(self, t, op=None, async_op: bool = False):
        return self._Done() if async_op else None

    def broadcast(self, t, src: int = 0) -> None:
        return None

    def exchange(self, pairs) -> None:
        return None


def _comm(comm: Optional[Collectives], process_group) -> Collectives:
    return comm if comm is not None else Collectives(process_group)


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
    ``bucket_mb`` megabytes.  On xGMI a ring all-reduce is bound by one link (about
    153 GB/s), so a few large buckets beat many small ones: the 73.7 MB of fp32
    gradients of the dim=48 network go out as three ~25 MB messages.
    Default (``overlap=False``): the backward pass leaves its gradients where autograd
    puts them (no accumulation into a zeroed buffer: 495 add kernels and a 74 MB
    memset per step less), one multi-tensor copy per bucket gathers them into the
    flat buffer, the all-reduces go out together, and ``.grad`` then points into the
    reduced buffers.  ``overlap=True``: every parameter's ``.grad`` is a view into its
    bucket from the start, autograd accumulates straight into the communication
    buffer, and a bucket's all-reduce starts from inside the backward pass as soon
    as its last gradient is there.

    After the last bucket has been reduced the gradients are divided by the world
    size (L1 is a mean over the local batch, so the result equals the gradient of
    the mean over the global batch), the global norm is clipped and Adam steps --
    identically on every rank, so parameters stay in sync without a broadcast.
    """

    def __init__(self, model: nn.Module, world_size: Optional[int] = None, lr: float = 1e-4, weight_decay: float = 1e-6,
                 max_grad_norm: float = 1.0, bucket_mb: float = 25.0,
                 loss_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor] = F.l1_loss,
                 process_group=None, comm: Optional[Collectives] = None, overlap: Optional[bool] = None):
        self.model = model
        self.comm = _comm(comm, process_group)
        # overlap=True starts a bucket's all-reduce from the backward pass as soon as its last gradient is there;
        # False (the default; RST_DP_OVERLAP=1 flips it) starts them after the backward pass.  The collective's kernels
        # run on RCCL's own stream: overlapped, they share compute units with the backward's matrix kernels, and on
        # MI355X dense v_mfma_f32_16x16x32 waves were measured to corrupt a dependent multiply-add chain of a kernel
        # from another stream (tools/two_stream_repro.hip).  Sequenced, the exposed time is the all-reduce itself
        # (73.7 MB of fp32 gradients: a few hundred microseconds over xGMI) and nothing shares the chip with it.
        self.overlap = (os.environ.get("RST_DP_OVERLAP", "0") == "1") if overlap is None else bool(overlap)
        self.world = world_size if world_size is not None else self.comm.world_size()
        self.max_grad_norm = max_grad_norm
        self.loss_fn = loss_fn
        params = [p for p in model.parameters() if p.requires_grad]
        self.params = params
        dev, dt = params[0].device, params[0].dtype
        use_fused = dev.type == "cuda"
        self.optimizer = torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, fused=use_fused)
        self.reduce = self.comm.active(self.world)
        self.buckets: List[_Bucket] = []
        self._owner = {}
        if not self.reduce:
            # one rank, nothing to reduce: no communication buffers; gradients are dropped (not zeroed) between steps,
            # so autograd keeps each produced gradient as it is instead of adding it to a zero
            return
        # buckets in reverse registration order
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
        if self.overlap:
            for b in self.buckets:
                for i, (p, v) in enumerate(zip(b.params, b.views)):
                    p.grad = v
                    self._owner[p] = (b, i)
            for p in params:
                p.register_post_accumulate_grad_hook(self._on_grad_ready)
        self.broadcast_parameters()

    def broadcast_parameters(self):
        """Make every rank start from rank 0's parameters."""
        if self.reduce:
            ps = list(self.model.parameters())
            with torch.no_grad():
                flat = torch.cat([p.reshape(-1) for p in ps])         # one message instead of one per tensor
                self.comm.broadcast(flat, src=0)
                off = 0
                for p in ps:
                    # an in-place copy_ (not a write through .data) so that the parameter's version counter moves and
                    # anything derived from the old values (prepared bf16 weights) is rebuilt
                    p.copy_(flat[off:off + p.numel()].view_as(p))
                    off += p.numel()

    def _on_grad_ready(self, p: nn.Parameter):
        b, i = self._owner[p]
        v = b.views[i]
        if p.grad.data_ptr() != v.data_ptr():
            # autograd replaced the view (first accumulation into a None grad): copy in and restore the view
            v.copy_(p.grad)
            p.grad = v
        b.pending -= 1
        if b.pending == 0 and self.overlap:
            b.handle = self.comm.all_reduce(b.flat, async_op=True)

    def zero_grad(self):
        if not self.reduce or not self.overlap:
            for p in self.params:
                p.grad = None
            return
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
        if self.reduce and not self.overlap:
            with torch.no_grad():
                for b in self.buckets:
                    have = [(v, p.grad) for v, p in zip(b.views, b.params) if p.grad is not None]
                    for v, p in zip(b.views, b.params):
                        if p.grad is None:          # a parameter without gradient this step: it still takes part
                            v.zero_()
                    if have:
                        torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
                    b.handle = self.comm.all_reduce(b.flat, async_op=True)
                for b in self.buckets:
                    b.handle.wait()
                    b.flat.div_(self.world)
                    for v, p in zip(b.views, b.params):
                        p.grad = v
        elif self.reduce:
            for b in self.buckets:
                if b.handle is None:      # a parameter without gradient this step kept the bucket open
                    b.handle = self.comm.all_reduce(b.flat, async_op=True)
            for b in self.buckets:
                b.handle.wait()
                b.flat.div_(self.world)
        torch.nn.utils.clip_grad_norm_(self.params, max_norm=self.max_grad_norm)
        self.optimizer.step()
        # the fused optimiser updates the parameters without moving their version counters: weights prepared from the
        # old values (folded LayerNorms, bf16 fragments) must not survive into the next forward or an evaluation
        invalidate = getattr(self.model, "invalidate_prepared", None)
        if invalidate is not None:
            invalidate()
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
                    gather: bool = True, comm: Optional[Collectives] = None) -> Optional[torch.Tensor]:
    """x`scale` inference of a 1 x C x H x W image in overlapping tiles.

    Tile t goes to rank t % world.  Each rank batches its same-shaped tiles
    (up to `tile_batch` per forward), keeps the cores of the outputs, and the
    cores are stitched into the full output on rank 0 (returned there; other
    ranks return None).  With world == 1 no collective is used (unless the collectives object is `eager`).

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

    comm = _comm(comm, process_group)
    if comm.active(world):
        # the number of output channels, agreed on over the ranks (a rank may hold no tile)
        co = torch.tensor([c_out if c_out is not None else 0], device=image.device)
        comm.all_reduce(co, op=dist.ReduceOp.MAX)
        c_out = int(co.item())
    if world == 1:
        canvas = torch.empty((1, c_out, h * scale, w * scale), dtype=image.dtype, device=image.device)
        for i, core in results.items():
            place(canvas, i, core)
        return canvas
    if not gather:
        return None
    # every rank sends its cores to rank 0 as one flat message (sizes are known from the grid)
    flat = torch.cat([results[i].reshape(-1) for i in sorted(results)]) if results else image.new_empty(0)
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
        comm.exchange([(None, bufs[r], r) for r in range(1, world) if sizes[r] > 0])
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
        comm.exchange([(flat, None, 0)])
    return None


# ----------------------------------------------------------------------------
# exact row-sharded inference of one frame
# ----------------------------------------------------------------------------
HALO = 2        # rows: a TransformerBlock holds two depthwise 3x3 convolutions, and halos are refreshed after every block


def row_partition(h: int, world: int, multiple: int = 8) -> List[Tuple[int, int]]:
    """Contiguous row ranges [a, b) per rank, cut at multiples of `multiple` (three 2x down-samplings), as even
    as possible; every rank gets at least HALO rows at the coarsest level."""
    units = h // multiple
    if h % multiple or units < HALO * world:
        raise ValueError(f"cannot cut {h} rows into {world} slabs of at least {HALO * multiple} rows in steps of {multiple}")
    base, extra = divmod(units, world)
    out, a = [], 0
    for r in range(world):
        b = a + base + (1 if r < extra else 0)
        out.append((a * multiple, b * multiple))
        a = b
    return out


class _Slab:
    """Rows [lo, hi) of a level's tensor held by this rank; [a, b) are its own rows, the rest is halo."""

    def __init__(self, interior: torch.Tensor, a: int, b: int, rows_total: int):
        self.a, self.b, self.total = a, b, rows_total
        self.lo, self.hi = max(a - HALO, 0), min(b + HALO, rows_total)
        n, c, _, w = interior.shape
        self.t = interior.new_empty((n, c, self.hi - self.lo, w))
        self.t[:, :, self.r0:self.r1] = interior

    @property
    def r0(self) -> int:
        return self.a - self.lo

    @property
    def r1(self) -> int:
        return self.b - self.lo

    def interior(self) -> torch.Tensor:
        return self.t[:, :, self.r0:self.r1]


def _refresh_halo(s: _Slab, rank: int, world: int, comm: Optional[Collectives]) -> None:
    """The halo rows of the slab <- the neighbours' outermost own rows (and the other way round)."""
    if world == 1:
        return
    pairs, land = [], []
    if s.a > 0:
        up = torch.empty_like(s.t[:, :, :s.r0])
        pairs.append((s.t[:, :, s.r0:s.r0 + HALO].contiguous(), up, rank - 1))
        land.append((slice(0, s.r0), up))
    if s.b < s.total:
        dn = torch.empty_like(s.t[:, :, s.r1:])
        pairs.append((s.t[:, :, s.r1 - HALO:s.r1].contiguous(), dn, rank + 1))
        land.append((slice(s.r1, s.t.shape[2]), dn))
    _comm(comm, None).exchange(pairs)
    for sl, buf in land:
        s.t[:, :, sl] = buf


def sharded_frame_inference(model: nn.Module, image: torch.Tensor, rank: int = 0, world: int = 1, process_group=None,
                            gather: bool = True, comm: Optional[Collectives] = None) -> Optional[torch.Tensor]:
    """EXACT x2 inference of one 1 x C x H x W frame whose rows are split over `world` GPUs (same result as the
    network on the whole frame, up to summation order): each rank runs the network on its row slab plus two halo
    rows; inside every TransformerBlock the only global quantities -- the Gram matrices q.k^T and the squared
    norms of q and k of the channel attention -- are summed over the ranks by one small all-reduce
    (`ops.block_shard_stats` / `ops.block_shard_apply`), and after every block and every resampling the halo rows
    are refreshed from the neighbours (point-to-point, 2 rows x C x W each way).  Every rank passes the same
    `image`; rank 0 returns the assembled output when `gather`, otherwise each rank returns its own rows."""
    from . import ops
    if image.dim() != 4 or image.shape[0] != 1:
        raise ValueError("sharded_frame_inference expects a 1 x C x H x W image")
    if getattr(model, "dual_pixel_task", False):
        raise NotImplementedError("dual_pixel_task")
    _, _, h, w = image.shape
    if w % 8:
        raise ValueError("W must be divisible by 8")
    cuts = row_partition(h, world)
    a, b = cuts[rank]
    comm = _comm(comm, process_group)

    def slab(interior, factor_num, factor_den=1):
        """interior rows at the level whose row count is h * num / den"""
        s = _Slab(interior, a * factor_num // factor_den, b * factor_num // factor_den, h * factor_num // factor_den)
        _refresh_halo(s, rank, world, comm)
        return s

    def stage(blocks, s: _Slab) -> _Slab:
        for blk in blocks:
            heads, hidden, par = blk.attn.num_heads, blk.ffn.hidden_features, blk._params()
            stats = ops.block_shard_stats(s.t, s.r0, s.r1, heads, hidden, par)
            if comm.active(world):
                comm.all_reduce(stats)
            ops.block_shard_apply(s.t, stats, heads, hidden, par)
            _refresh_halo(s, rank, world, comm)
        return s

    def conv(s: _Slab, c: nn.Conv2d, mode: int = 0) -> torch.Tensor:
        """3x3 convolution of the slab (+ PixelUnshuffle / PixelShuffle); rows next to a cut are wrong: cropped by the callers"""
        if ops.conv3x3_supported(s.t, c.weight, mode):
            return ops.conv3x3(s.t, c.weight, c.bias, mode)
        y = F.conv2d(s.t, c.weight, c.bias, padding=1)
        return y if mode == 0 else ops.pixel_unshuffle2(y) if mode == 1 else ops.pixel_shuffle2(y)

    def down(s: _Slab, mod, den) -> _Slab:                            # conv3x3 + PixelUnshuffle(2): rows / 2
        y = conv(s, mod.body[0], 1)
        lo = s.lo // 2
        return slab(y[:, :, s.a // 2 - lo:s.b // 2 - lo], 1, den)

    def up(s: _Slab, mod, den) -> _Slab:                              # conv3x3 + PixelShuffle(2): rows * 2
        y = conv(s, mod.body[0], 2)
        lo = s.lo * 2
        return slab(y[:, :, s.a * 2 - lo:s.b * 2 - lo], 1, den)

    with torch.no_grad():
        x0 = slab(image[:, :, a:b].contiguous(), 1)                  # input rows + halo (kept for the bilinear base)
        e1 = slab(conv(x0, model.patch_embed.proj)[:, :, x0.r0:x0.r1], 1)
        e1 = stage(model.encoder_level1, e1)
        e2 = stage(model.encoder_level2, down(e1, model.down1_2, 2))
        e3 = stage(model.encoder_level3, down(e2, model.down2_3, 4))
        lat = stage(model.latent, down(e3, model.down3_4, 8))

        d3 = up(lat, model.up4_3, 4)
        d3.t = ops.conv1x1_cat(d3.t, e3.t, model.reduce_chan_level3.weight, model.reduce_chan_level3.bias)
        d3 = stage(model.decoder_level3, d3)
        d2 = up(d3, model.up3_2, 2)
        d2.t = ops.conv1x1_cat(d2.t, e2.t, model.reduce_chan_level2.weight, model.reduce_chan_level2.bias)
        d2 = stage(model.decoder_level2, d2)
        d1 = up(d2, model.up2_1, 1)
        d1.t = torch.cat([d1.t, e1.t], 1)
        d1 = stage(model.refinement, stage(model.decoder_level1, d1))

        # x2 tail on the slab: the halo of d1 (2 rows = 4 output rows) covers the last 3x3 convolution and the
        # bilinear base without another exchange
        upt = conv(d1, model.output_upscale.body[0], 2)
        base = F.interpolate(x0.t, scale_factor=2, mode="bilinear", align_corners=False)
        if model.output.weight.shape[0] <= 4:
            out = ops.conv3x3_fewout(upt, model.output.weight, model.output.bias, base)
        else:
            out = F.conv2d(upt, model.output.weight, model.output.bias, padding=1) + base
        mine = out[:, :, 2 * d1.r0:2 * d1.r1].contiguous()

    if world == 1 or not gather:
        return mine
    # rank 0 assembles the rows
    if rank == 0:
        canvas = mine.new_empty((1, mine.shape[1], 2 * h, 2 * w))
        canvas[:, :, 2 * a:2 * b] = mine
        bufs = [(canvas.new_empty((1, mine.shape[1], 2 * (cb - ca), 2 * w)), r) for r, (ca, cb) in enumerate(cuts) if r > 0]
        comm.exchange([(None, buf, r) for buf, r in bufs])
        for buf, r in bufs:
            canvas[:, :, 2 * cuts[r][0]:2 * cuts[r][1]] = buf
        return canvas
    comm.exchange([(mine, None, 0)])
    return None
