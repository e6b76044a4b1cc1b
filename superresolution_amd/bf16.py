"""bf16 storage path (inference): host side of the rst_bf16_* entry points.

Between kernels the activations are bf16 in the channel-packed layout of the
library, a torch tensor of shape (B, ceil(C/8), H, W, 8): the 8 channels of a
block for one pixel are one 16-byte chunk, which is what the matrix
instructions consume directly.  Parameters stay fp32; every sum is fp32.
torch supplies memory and the stream only; a CPU tensor raises RuntimeError.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import BlockGrads, BlockParams, PARAM_FIELDS, check
from .ops import _fill, _ptr, _stream, make_desc, workspace


class Packed:
    """A (B, C, H, W) activation in the packed bf16 layout."""
    __slots__ = ("data", "C")

    def __init__(self, data: torch.Tensor, C: int):
        if data.dtype != torch.bfloat16 or data.dim() != 5 or data.shape[-1] != 8 or not data.is_contiguous():
            raise RuntimeError("Packed needs a contiguous bf16 tensor of shape (B, ceil(C/8), H, W, 8)")
        if not data.is_cuda:
            raise RuntimeError("Packed tensors live on a ROCm GPU (there is no CPU fallback)")
        self.data, self.C = data, int(C)

    @property
    def shape(self):
        b, cb, h, w, _ = self.data.shape
        return (b, self.C, h, w)


def _f32(p: Optional[torch.Tensor], name: str) -> Optional[torch.Tensor]:
    if p is None:
        return None
    if not p.is_cuda:
        raise RuntimeError(f"{name}: parameter is on {p.device}; the bf16 path only runs on a ROCm GPU")
    return p.detach().float().contiguous()


def pack(x: torch.Tensor) -> Packed:
    """NCHW fp32 or bf16 -> packed bf16 (round to nearest even)."""
    lib = _lib.load()
    if not x.is_cuda:
        raise RuntimeError(f"pack: tensor is on {x.device}; this operator only runs on a ROCm GPU")
    if x.dtype not in (torch.float32, torch.bfloat16) or x.dim() != 4:
        raise RuntimeError(f"pack: expected a 4-D fp32 or bf16 tensor, got {x.dtype} {tuple(x.shape)}")
    x = x.contiguous()
    B, C, H, W = x.shape
    out = torch.empty((B, (C + 7) // 8, H, W, 8), dtype=torch.bfloat16, device=x.device)
    check(lib.rst_bf16_pack(_ptr(x), int(x.dtype == torch.float32), _ptr(out), B, C, H * W, _stream()), "rst_bf16_pack")
    return Packed(out, C)


def unpack(p: Packed, dtype=torch.bfloat16) -> torch.Tensor:
    """packed bf16 -> NCHW fp32 or bf16."""
    lib = _lib.load()
    if dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"unpack: dtype {dtype} not supported")
    B, C, H, W = p.shape
    out = torch.empty((B, C, H, W), dtype=dtype, device=p.data.device)
    check(lib.rst_bf16_unpack(_ptr(p.data), _ptr(out), int(dtype == torch.float32), B, C, H * W, _stream()), "rst_bf16_unpack")
    return out


def to_channels_last(p: Packed) -> torch.Tensor:
    """packed -> a (B, C, H, W) bf16 tensor in torch's channels_last memory format (what the dense 3x3
    convolutions are run on)."""
    lib = _lib.load()
    B, C, H, W = p.shape
    out = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=p.data.device)
    check(lib.rst_bf16_to_channels_last(_ptr(p.data), _ptr(out), B, C, H * W, _stream()), "rst_bf16_to_channels_last")
    return out.permute(0, 3, 1, 2)


def from_channels_last(x: torch.Tensor, shuffle: int = 0) -> Packed:
    """(B, C, H, W) bf16 in channels_last memory format -> packed; shuffle 1: through nn.PixelUnshuffle(2),
    2: through nn.PixelShuffle(2) (reference: model/Restromer.py:196,211)."""
    lib = _lib.load()
    if not x.is_cuda or x.dtype != torch.bfloat16 or x.dim() != 4:
        raise RuntimeError("from_channels_last: expected a 4-D bf16 tensor on a ROCm GPU")
    x = x.contiguous(memory_format=torch.channels_last)
    B, C, H, W = x.shape
    if shuffle == 0:
        Co, Ho, Wo = C, H, W
    elif shuffle == 1:
        Co, Ho, Wo = C * 4, H // 2, W // 2
    elif shuffle == 2:
        Co, Ho, Wo = C // 4, H * 2, W * 2
    else:
        raise RuntimeError(f"from_channels_last: shuffle mode {shuffle}")
    if Co % 8:
        raise RuntimeError(f"from_channels_last: {Co} result channels (a multiple of 8 is needed)")
    out = torch.empty((B, Co // 8, Ho, Wo, 8), dtype=torch.bfloat16, device=x.device)
    check(lib.rst_bf16_from_channels_last(_ptr(x), _ptr(out), int(shuffle), B, C, H, W, _stream()), "rst_bf16_from_channels_last")
    return Packed(out, Co)


def conv1x1(p: Packed, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[Packed] = None) -> Packed:
    """nn.Conv2d(kernel_size=1) on a packed tensor (reference: model/Restromer.py:309-311,326-328)."""
    lib = _lib.load()
    B, Cin, H, W = p.shape
    w = _f32(weight, "conv1x1 weight")
    Cout = w.shape[0]
    if w.numel() != Cout * Cin:
        raise RuntimeError(f"conv1x1: weight {tuple(weight.shape)} does not match {Cin} input channels")
    b = _f32(bias, "conv1x1 bias")
    nb = lib.rst_bf16_conv1x1_workspace_bytes(Cin, Cout)
    if nb == 0:
        raise RuntimeError(f"conv1x1: {Cin} input channels not supported by the bf16 path")
    ws = workspace(nb, p.data.device)
    out = torch.empty((B, (Cout + 7) // 8, H, W, 8), dtype=torch.bfloat16, device=p.data.device)
    if residual is not None and residual.shape != (B, Cout, H, W):
        raise RuntimeError("conv1x1: residual shape mismatch")
    check(lib.rst_bf16_conv1x1_fwd(_ptr(p.data), _ptr(w), _ptr(b), None if residual is None else _ptr(residual.data), _ptr(out),
                                   B, Cin, Cout, H * W, _ptr(ws), ws.numel(), _stream()), "rst_bf16_conv1x1_fwd")
    return Packed(out, Cout)


def conv3x3_supported(Cin: int, Cout: int, H: int, W: int, mode: int) -> bool:
    return bool(_lib.load().rst_bf16_conv3x3_supported(Cin, Cout, H, W, mode))


def conv3x3(p: Packed, weight: torch.Tensor, bias: Optional[torch.Tensor], mode: int = 0) -> Packed:
    """nn.Conv2d(Cin, Cout, 3, padding=1) on a packed tensor, followed by nothing (mode 0), nn.PixelUnshuffle(2)
    (mode 1) or nn.PixelShuffle(2) (mode 2) -- OverlapPatchEmbed / Downsample / Upsample of the reference
    (model/Restromer.py:172-215) in one kernel (rst_bf16_conv3x3_fwd)."""
    lib = _lib.load()
    B, Cin, H, W = p.shape
    w, b = _f32(weight, "conv3x3 weight"), _f32(bias, "conv3x3 bias")
    Cout = w.shape[0]
    if tuple(w.shape) != (Cout, Cin, 3, 3):
        raise RuntimeError(f"conv3x3: weight {tuple(w.shape)} does not fit {Cin} input channels")
    if not lib.rst_bf16_conv3x3_supported(Cin, Cout, H, W, mode):
        raise RuntimeError(f"conv3x3: shape Cin={Cin} Cout={Cout} {H}x{W} mode={mode} not supported by the bf16 path")
    Co, Ho, Wo = (Cout, H, W) if mode == 0 else (4 * Cout, H // 2, W // 2) if mode == 1 else (Cout // 4, 2 * H, 2 * W)
    out = torch.empty((B, (Co + 7) // 8, Ho, Wo, 8), dtype=torch.bfloat16, device=p.data.device)
    ws = workspace(lib.rst_bf16_conv3x3_workspace_bytes(Cin, Cout), p.data.device)
    check(lib.rst_bf16_conv3x3_fwd(_ptr(p.data), _ptr(w), _ptr(b), _ptr(out), B, Cin, Cout, H, W, mode, _ptr(ws), ws.numel(),
                                   _stream()), "rst_bf16_conv3x3_fwd")
    return Packed(out, Co)


def conv3x3_fewout(p: Packed, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                   lowres: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The network's last layer on a packed tensor (model/Restromer.py:381-383, 420-431): nn.Conv2d(Cin, Cout <= 4, 3,
    padding=1) with fp32 weights and sums and an fp32 NCHW result, plus `residual` (fp32, the result's shape) or the
    bilinear x2 of `lowres` (fp32, half the size) -- rst_bf16_conv3x3_fewout_fwd."""
    lib = _lib.load()
    B, Cin, H, W = p.shape
    w, b = _f32(weight, "conv3x3 weight"), _f32(bias, "conv3x3 bias")
    Cout = w.shape[0]
    if tuple(w.shape) != (Cout, Cin, 3, 3) or Cout > 4:
        raise RuntimeError(f"conv3x3_fewout: weight {tuple(w.shape)} does not fit {Cin} input channels / at most 4 outputs")
    res = _f32(residual, "conv3x3 residual")
    low = _f32(lowres, "conv3x3 low-resolution residual")
    if res is not None and tuple(res.shape) != (B, Cout, H, W):
        raise RuntimeError(f"conv3x3_fewout: residual {tuple(res.shape)} != {(B, Cout, H, W)}")
    if low is not None and tuple(low.shape) != (B, Cout, H // 2, W // 2):
        raise RuntimeError(f"conv3x3_fewout: low-resolution tensor {tuple(low.shape)} != {(B, Cout, H // 2, W // 2)}")
    y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=p.data.device)
    check(lib.rst_bf16_conv3x3_fewout_fwd(_ptr(p.data), _ptr(w), _ptr(b), _ptr(res), _ptr(low), _ptr(y), B, Cin, Cout, H, W, _stream()),
          "rst_bf16_conv3x3_fewout_fwd")
    return y


class PreparedCache:
    """Prepared weights of one block (rst_bf16_block_prepare), rebuilt when a parameter tensor is replaced or
    modified in place (torch bumps ``_version`` on every in-place write, including ``copy_`` and ``load_state_dict``)
    and after every optimiser step of the process (``ops.weights_generation``: fused optimisers do not move
    ``_version``).  A write through ``param.data`` is invisible to both: call :meth:`invalidate` (or
    ``Restormer.invalidate_prepared()``) after such a write."""
    __slots__ = ("key", "blob")

    def __init__(self):
        self.key, self.blob = None, None

    def invalidate(self):
        self.key, self.blob = None, None

    @staticmethod
    def _key(params, device):
        from .ops import weights_generation
        return tuple((n, t.data_ptr(), t._version) for n, t in params.items() if t is not None) + (str(device), weights_generation())

    def is_current(self, params, device) -> bool:
        """True when the kept weights were prepared from exactly these parameter tensors (nothing would be launched)."""
        return self.key is not None and self.key == self._key(params, device)

    def get(self, lib, desc, par, params, device):
        key = self._key(params, device)
        if key != self.key:
            nb = lib.rst_bf16_block_prepared_bytes(ctypes.byref(desc))
            blob = torch.empty(nb, dtype=torch.uint8, device=device)
            check(lib.rst_bf16_block_prepare(ctypes.byref(desc), ctypes.byref(par), _ptr(blob), _stream()), "rst_bf16_block_prepare")
            self.key, self.blob = key, blob
        return self.blob


def block_(p: Packed, heads: int, hidden: int, params: Dict[str, Optional[torch.Tensor]],
           cache: Optional[PreparedCache] = None) -> Packed:
    """TransformerBlock (reference: model/Restromer.py:163-167) applied in place to a packed tensor.  With a
    `cache` the prepared weights are kept between calls."""
    lib = _lib.load()
    B, C, H, W = p.shape
    tensors = {n: _f32(params.get(n), n) for n in PARAM_FIELDS}
    desc = make_desc(B, C, H, W, heads, hidden, tensors["norm1_bias"] is not None, tensors["qkv_bias"] is not None)
    nb = lib.rst_bf16_block_workspace_bytes(ctypes.byref(desc))
    if nb == 0:
        raise RuntimeError("bf16 block: " + lib.rst_last_error().decode("utf-8", "replace"))
    par = _fill(BlockParams(), tensors)
    prepared = None
    if cache is not None:
        # keyed on the module's own parameter tensors (the fp32 views above are the same objects for fp32 models)
        prepared = cache.get(lib, desc, par, {n: params.get(n) for n in PARAM_FIELDS}, p.data.device)
    ws = workspace(nb, p.data.device)
    check(lib.rst_bf16_block_fwd(ctypes.byref(desc), ctypes.byref(par), _ptr(prepared), _ptr(p.data), _ptr(p.data), _ptr(ws),
                                 ws.numel(), _stream()), "rst_bf16_block_fwd")
    return p


def cat(a: Packed, b: Packed) -> Packed:
    """torch.cat([a, b], 1) (reference: model/Restromer.py:402,407,412); needs a.C % 8 == 0."""
    if a.C % 8:
        raise RuntimeError("cat: the first tensor's channel count must be a multiple of 8 in the packed layout")
    return Packed(torch.cat([a.data, b.data], 1), a.C + b.C)


# ----------------------------------------------------------------------------
# training with bf16 activations: a stage of TransformerBlocks as one autograd node
# ----------------------------------------------------------------------------
class _StageFn(torch.autograd.Function):
    """A run of TransformerBlocks (one `nn.Sequential` of the network) on bf16 activations, differentiable.

    Forward, two ways per block.  Saving (`rst_bf16_block_fwd_save`, the default while it fits): the block's input and
    every intermediate its backward pass reads stay in packed bf16 -- 2 (8 C + 5 hidden) bytes per pixel and block; a
    288 GB device holds that for every block of the network at training crop sizes many times over -- and the backward
    pass (`rst_bf16_block_bwd_saved`) evaluates nothing again.  Recomputing (`rst_bf16_block_fwd_keep`, the fused
    inference kernels): what is kept is the block's INPUT only -- 2 bytes per element and block -- and its
    attention's small matrices (softmax, normalised Gram matrix, norms, the product with project_out:
    B*C*(C + 2c + 2) floats, nothing that grows with the image; ``RST_BF16_KEEP_ATTN=0``: not even those);
    `rst_bf16_block_bwd_kept` evaluates the block again on the kept input.  ``RST_BF16_SAVE=1`` / ``0`` forces one or
    the other; by default a stage saves as long as its blocks' buffers stay within a quarter of the device memory
    that is free when the stage starts.  Either way: bf16 activations and activation gradients, fp32 sums, fp32
    parameters and parameter gradients.  ``RST_BF16_BWD=fp32`` switches the backward to the fp32 kernels on the
    unpacked input (`ops.block_recompute_backward`: the gradient of the fp32 block at the bf16-rounded activations).
    x: (B, C, H, W) bf16."""

    @staticmethod
    def forward(ctx, x, metas, caches, *flat):
        lib = _lib.load()
        if not x.is_cuda or x.dtype != torch.bfloat16:
            raise RuntimeError("bf16 stage: expected a bf16 tensor on a ROCm GPU")
        import os
        # a 5-D tensor is the packed layout itself (B, C/8, H, W, 8): the training network hands its stages packed
        # tensors and gets packed tensors back, gradients included
        ctx.packed_io = x.dim() == 5
        p = Packed(x.contiguous(), x.shape[1] * 8) if ctx.packed_io else pack(x)
        kept, attn, modes = [], [], []
        keep_attn = os.environ.get("RST_BF16_KEEP_ATTN", "1") != "0"
        fp32_bwd = os.environ.get("RST_BF16_BWD", "bf16") == "fp32"
        save = os.environ.get("RST_BF16_SAVE", "auto")
        budget = 0
        if fp32_bwd or save == "0":
            budget = 0
        elif save == "1":
            budget = 1 << 62
        else:
            free, _ = torch.cuda.mem_get_info(p.data.device)
            free += torch.cuda.memory_reserved(p.data.device) - torch.cuda.memory_allocated(p.data.device)
            budget = free // 4
        n = len(PARAM_FIELDS)
        for i, (heads, hidden) in enumerate(metas):
            params = dict(zip(PARAM_FIELDS, flat[i * n:(i + 1) * n]))
            kept.append(p.data)
            out = Packed(torch.empty_like(p.data), p.C)
            need = _saved_bytes(lib, p, heads, hidden, params) if budget > 0 else 0
            if 0 < need <= budget:
                budget -= need
                attn.append(_block_into(lib, p, out, heads, hidden, params, caches[i], save=True))
                modes.append("saved")
            elif keep_attn and not fp32_bwd:
                attn.append(_block_into(lib, p, out, heads, hidden, params, caches[i], keep_attn=True))
                modes.append("kept")
            else:
                _block_into(lib, p, out, heads, hidden, params, caches[i])
                attn.append(None)
                modes.append("input")
            p = out
        ctx.metas, ctx.shape, ctx.caches = metas, tuple(p.shape), caches
        ctx.present = [t is not None for t in flat]
        ctx.modes = modes
        ctx.save_for_backward(*kept, *[a for a in attn if a is not None], *[t for t in flat if t is not None])
        return p.data if ctx.packed_io else unpack(p, torch.bfloat16)

    @staticmethod
    def backward(ctx, dy):
        import os
        lib = _lib.load()
        nb = len(ctx.metas)
        na = sum(m != "input" for m in ctx.modes)
        kept, bufs, rest = ctx.saved_tensors[:nb], iter(ctx.saved_tensors[nb:nb + na]), iter(ctx.saved_tensors[nb + na:])
        attn = [None if m == "input" else next(bufs) for m in ctx.modes]
        flat = [next(rest) if pr else None for pr in ctx.present]
        B, C, H, W = ctx.shape
        n = len(PARAM_FIELDS)
        grads = [None] * len(flat)
        if os.environ.get("RST_BF16_BWD", "bf16") == "fp32":
            from . import ops
            g = (unpack(Packed(dy.contiguous(), C), torch.float32) if ctx.packed_io else dy.float().contiguous())
            for i in reversed(range(nb)):
                heads, hidden = ctx.metas[i]
                params = dict(zip(PARAM_FIELDS, flat[i * n:(i + 1) * n]))
                x32 = unpack(Packed(kept[i], C), torch.float32)
                g, pg = ops.block_recompute_backward(x32, g, heads, hidden, params)
                for j, name in enumerate(PARAM_FIELDS):
                    grads[i * n + j] = pg[name]
            return ((pack(g).data if ctx.packed_io else g.to(torch.bfloat16)), None, None) + tuple(grads)
        g = Packed(dy.contiguous(), C) if ctx.packed_io else pack(dy.contiguous())
        for i in reversed(range(nb)):
            heads, hidden = ctx.metas[i]
            params = dict(zip(PARAM_FIELDS, flat[i * n:(i + 1) * n]))
            g, pg = block_backward(Packed(kept[i], C), g, heads, hidden, params, ctx.caches[i],
                                   kept_attn=attn[i] if ctx.modes[i] == "kept" else None,
                                   saved=attn[i] if ctx.modes[i] == "saved" else None)
            for j, name in enumerate(PARAM_FIELDS):
                grads[i * n + j] = pg[name]
        return ((g.data if ctx.packed_io else unpack(g, torch.bfloat16)), None, None) + tuple(grads)


def block_backward(x: Packed, dy: Packed, heads: int, hidden: int, params: Dict[str, Optional[torch.Tensor]],
                   cache: Optional[PreparedCache] = None, kept_attn: Optional[torch.Tensor] = None,
                   saved: Optional[torch.Tensor] = None):
    """Gradients of one TransformerBlock from its packed bf16 input and the packed gradient of its output
    (`rst_bf16_block_bwd`).  Returns (dx packed, {parameter name: fp32 gradient or None}).  With the block's `cache`
    the weights prepared for the forward pass are used again (they are rebuilt if a parameter changed since); with
    `kept_attn` (what `block_forward_keep` returned for this input) the attention is not formed again
    (`rst_bf16_block_bwd_kept`); with `saved` (what `block_forward_save` returned) nothing is
    (`rst_bf16_block_bwd_saved`)."""
    lib = _lib.load()
    B, C, H, W = x.shape
    tensors = {n: _f32(None if params.get(n) is None else params[n].detach(), n) for n in PARAM_FIELDS}
    desc = make_desc(B, C, H, W, heads, hidden, tensors["norm1_bias"] is not None, tensors["qkv_bias"] is not None)
    if saved is not None and kept_attn is not None:
        raise RuntimeError("bf16 block backward: either the kept attention or the saved activations")
    nb = (lib.rst_bf16_block_bwd_saved_workspace_bytes if saved is not None else lib.rst_bf16_block_bwd_workspace_bytes)(ctypes.byref(desc))
    if nb == 0:
        raise RuntimeError("bf16 block backward: " + lib.rst_last_error().decode("utf-8", "replace"))
    par = _fill(BlockParams(), tensors)
    grads = {n: (None if t is None else torch.empty_like(t)) for n, t in tensors.items()}
    gr = _fill(BlockGrads(), grads)
    prepared = None
    if cache is not None:
        prepared = cache.get(lib, desc, par, {n: params.get(n) for n in PARAM_FIELDS}, x.data.device)
    ws = workspace(nb, x.data.device)
    dx = Packed(torch.empty_like(x.data), x.C)
    if saved is not None:
        if saved.dtype != torch.uint8 or saved.numel() < lib.rst_bf16_block_saved_bytes(ctypes.byref(desc)) or not saved.is_cuda:
            raise RuntimeError("bf16 block backward: saved activations buffer of the wrong size, type or device")
        check(lib.rst_bf16_block_bwd_saved(ctypes.byref(desc), ctypes.byref(par), _ptr(prepared), _ptr(x.data), _ptr(dy.data), _ptr(dx.data),
                                           ctypes.byref(gr), _ptr(saved), _ptr(ws), ws.numel(), _stream()), "rst_bf16_block_bwd_saved")
    elif kept_attn is not None:
        if kept_attn.dtype != torch.uint8 or kept_attn.numel() < lib.rst_bf16_block_kept_bytes(ctypes.byref(desc)) or not kept_attn.is_cuda:
            raise RuntimeError("bf16 block backward: kept attention buffer of the wrong size, type or device")
        check(lib.rst_bf16_block_bwd_kept(ctypes.byref(desc), ctypes.byref(par), _ptr(prepared), _ptr(x.data), _ptr(dy.data), _ptr(dx.data),
                                          ctypes.byref(gr), _ptr(kept_attn), _ptr(ws), ws.numel(), _stream()), "rst_bf16_block_bwd_kept")
    else:
        check(lib.rst_bf16_block_bwd(ctypes.byref(desc), ctypes.byref(par), _ptr(prepared), _ptr(x.data), _ptr(dy.data), _ptr(dx.data), ctypes.byref(gr),
                                     _ptr(ws), ws.numel(), _stream()), "rst_bf16_block_bwd")
    return dx, grads


def block_forward_keep(x: Packed, heads: int, hidden: int, params: Dict[str, Optional[torch.Tensor]],
                       cache: Optional[PreparedCache] = None):
    """TransformerBlock on a packed tensor for a training step: returns (y packed, kept attention) -- the second is
    what `block_backward(..., kept_attn=)` wants (`rst_bf16_block_fwd_keep`)."""
    out = Packed(torch.empty_like(x.data), x.C)
    kept = _block_into(_lib.load(), x, out, heads, hidden, params, cache, True)
    return out, kept


def block_forward_save(x: Packed, heads: int, hidden: int, params: Dict[str, Optional[torch.Tensor]],
                       cache: Optional[PreparedCache] = None):
    """TransformerBlock on a packed tensor for a training step that may spend memory: returns (y packed, saved
    activations) -- the second is what `block_backward(..., saved=)` wants (`rst_bf16_block_fwd_save`)."""
    out = Packed(torch.empty_like(x.data), x.C)
    saved = _block_into(_lib.load(), x, out, heads, hidden, params, cache, save=True)
    return out, saved


def _saved_bytes(lib, src: Packed, heads, hidden, params) -> int:
    B, C, H, W = src.shape
    desc = make_desc(B, C, H, W, heads, hidden, params.get("norm1_bias") is not None, params.get("qkv_bias") is not None)
    return int(lib.rst_bf16_block_saved_bytes(ctypes.byref(desc)))


def _block_into(lib, src: Packed, dst: Packed, heads, hidden, params, cache, keep_attn=False, save=False):
    B, C, H, W = src.shape
    tensors = {n: _f32(params.get(n), n) for n in PARAM_FIELDS}
    desc = make_desc(B, C, H, W, heads, hidden, tensors["norm1_bias"] is not None, tensors["qkv_bias"] is not None)
    nb = lib.rst_bf16_block_workspace_bytes(ctypes.byref(desc))
    if nb == 0:
        raise RuntimeError("bf16 block: " + lib.rst_last_error().decode("utf-8", "replace"))
    par = _fill(BlockParams(), tensors)
    prepared = cache.get(lib, desc, par, {n: params.get(n) for n in PARAM_FIELDS}, src.data.device) if cache is not None else None
    ws = workspace(nb, src.data.device)
    if save:
        saved = torch.empty(lib.rst_bf16_block_saved_bytes(ctypes.byref(desc)), dtype=torch.uint8, device=src.data.device)
        check(lib.rst_bf16_block_fwd_save(ctypes.byref(desc), ctypes.byref(par), _ptr(prepared), _ptr(src.data), _ptr(dst.data), _ptr(saved),
                                          _ptr(ws), ws.numel(), _stream()), "rst_bf16_block_fwd_save")
        return saved
    if keep_attn:
        kept = torch.empty(lib.rst_bf16_block_kept_bytes(ctypes.byref(desc)), dtype=torch.uint8, device=src.data.device)
        check(lib.rst_bf16_block_fwd_keep(ctypes.byref(desc), ctypes.byref(par), _ptr(prepared), _ptr(src.data), _ptr(dst.data), _ptr(kept),
                                          _ptr(ws), ws.numel(), _stream()), "rst_bf16_block_fwd_keep")
        return kept
    check(lib.rst_bf16_block_fwd(ctypes.byref(desc), ctypes.byref(par), _ptr(prepared), _ptr(src.data), _ptr(dst.data), _ptr(ws),
                                 ws.numel(), _stream()), "rst_bf16_block_fwd")
    return None


def stage(x: torch.Tensor, blocks) -> torch.Tensor:
    """`blocks` (TransformerBlock modules) applied to a bf16 NCHW tensor, differentiable (see _StageFn)."""
    metas = tuple((b.attn.num_heads, b.ffn.hidden_features) for b in blocks)
    caches = []
    flat = []
    for b in blocks:
        if getattr(b, "_pk_cache", None) is None:
            b._pk_cache = PreparedCache()
        # A training forward prepares the weights afresh: they change every step, and not every optimiser moves the
        # version counter the cache is keyed on (torch's fused Adam does not).  The backward of this forward reuses them.
        b._pk_cache.invalidate()
        caches.append(b._pk_cache)
        par = b._params()
        flat.extend(par[n] for n in PARAM_FIELDS)
    return _StageFn.apply(x, metas, caches, *flat)


# ----------------------------------------------------------------------------
# the glue between the stages, differentiable, on packed tensors (training)
# ----------------------------------------------------------------------------
def _packed_data(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.bfloat16 or t.dim() != 5 or t.shape[-1] != 8:
        raise RuntimeError(f"{what}: expected a packed bf16 tensor (B, C/8, H, W, 8) on a ROCm GPU")
    return t.contiguous()


def conv3x3_backward(x: Packed, dy: Packed, weight: torch.Tensor, mode: int = 0, want_dx: bool = True, want_bias: bool = False):
    """Gradients of `conv3x3(x, weight, bias, mode)` from the packed gradient `dy` of its (shuffled) output
    (rst_bf16_conv3x3_bwd): (dx packed or None, dweight fp32, dbias fp32 or None)."""
    lib = _lib.load()
    B, Cin, H, W = x.shape
    w = _f32(weight, "conv3x3 weight")
    Cout = w.shape[0]
    if tuple(w.shape) != (Cout, Cin, 3, 3):
        raise RuntimeError(f"conv3x3 backward: weight {tuple(w.shape)} does not fit {Cin} input channels")
    Co, Ho, Wo = (Cout, H, W) if mode == 0 else (4 * Cout, H // 2, W // 2) if mode == 1 else (Cout // 4, 2 * H, 2 * W)
    if dy.shape != (B, Co, Ho, Wo):
        raise RuntimeError(f"conv3x3 backward: gradient shape {dy.shape} != {(B, Co, Ho, Wo)}")
    nb = lib.rst_bf16_conv3x3_bwd_workspace_bytes(B, Cin, Cout, H, W, mode, int(want_dx))
    if nb == 0:
        raise RuntimeError(f"conv3x3 backward: shape Cin={Cin} Cout={Cout} {H}x{W} mode={mode} not supported")
    ws = workspace(nb, x.data.device)
    dx = torch.empty_like(x.data) if want_dx else None
    dw = torch.empty_like(w)
    db = torch.empty(Cout, dtype=torch.float32, device=w.device) if want_bias else None
    check(lib.rst_bf16_conv3x3_bwd(_ptr(x.data), _ptr(dy.data), _ptr(w), _ptr(dx), _ptr(dw), _ptr(db), B, Cin, Cout, H, W, mode,
                                   _ptr(ws), ws.numel(), _stream()), "rst_bf16_conv3x3_bwd")
    return (Packed(dx, Cin) if want_dx else None), dw, db


class _Conv3x3TrainFn(torch.autograd.Function):
    """conv3x3 (+ PixelUnshuffle / PixelShuffle) on a packed tensor, differentiable: reference model/Restromer.py:172-215."""

    @staticmethod
    def forward(ctx, xdata, cin, weight, bias, mode):
        x = Packed(_packed_data(xdata, "conv3x3"), cin)
        y = conv3x3(x, weight, bias, mode)
        ctx.save_for_backward(x.data, weight)
        ctx.cin, ctx.cout_shuffled, ctx.mode, ctx.has_bias = cin, y.C, mode, bias is not None
        return y.data

    @staticmethod
    def backward(ctx, dy):
        xdata, weight = ctx.saved_tensors
        dx, dw, db = conv3x3_backward(Packed(xdata, ctx.cin), Packed(dy.contiguous(), ctx.cout_shuffled), weight, ctx.mode,
                                      ctx.needs_input_grad[0], ctx.has_bias)
        return (dx.data if dx is not None else None), None, dw.view_as(weight), db, None


class _Conv1x1TrainFn(torch.autograd.Function):
    """nn.Conv2d(kernel_size=1) on a packed tensor, differentiable: the reductions after the skip concatenations
    (model/Restromer.py:309-311, 326-328, 403, 408)."""

    @staticmethod
    def forward(ctx, xdata, cin, weight, bias):
        x = Packed(_packed_data(xdata, "conv1x1"), cin)
        y = conv1x1(x, weight, bias)
        ctx.save_for_backward(x.data, weight)
        ctx.cin, ctx.cout, ctx.has_bias = cin, y.C, bias is not None
        return y.data

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        xdata, weight = ctx.saved_tensors
        B, _, H, W, _ = xdata.shape
        w = _f32(weight, "conv1x1 weight")
        dy = dy.contiguous()
        ws = workspace(lib.rst_bf16_conv1x1_bwd_workspace_bytes(B, ctx.cin, ctx.cout, H * W), xdata.device)
        dx = torch.empty_like(xdata)
        dw = torch.empty_like(w)
        db = torch.empty(ctx.cout, dtype=torch.float32, device=w.device) if ctx.has_bias else None
        check(lib.rst_bf16_conv1x1_bwd(_ptr(xdata), _ptr(dy), _ptr(w), _ptr(dx), _ptr(dw), _ptr(db), B, ctx.cin, ctx.cout, H * W,
                                       _ptr(ws), ws.numel(), _stream()), "rst_bf16_conv1x1_bwd")
        return dx, None, dw.view_as(weight), db


def _copy_blocks(src, src_c0, dst, dst_c0, nc):
    lib = _lib.load()
    B, scb, H, W, _ = src.shape
    check(lib.rst_bf16_copy_channel_blocks(_ptr(src), scb * 8, src_c0, _ptr(dst), dst.shape[1] * 8, dst_c0, nc, B, H * W, _stream()),
          "rst_bf16_copy_channel_blocks")


class _CatTrainFn(torch.autograd.Function):
    """torch.cat([a, b], 1) of two packed tensors (model/Restromer.py:402, 407, 412) and its gradient, whole channel
    blocks copied by the library."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _packed_data(a, "cat"), _packed_data(b, "cat")
        if a.shape[0] != b.shape[0] or a.shape[2:] != b.shape[2:]:
            raise RuntimeError(f"cat: shapes {tuple(a.shape)} and {tuple(b.shape)} do not go together")
        out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]) + tuple(a.shape[2:]), dtype=a.dtype, device=a.device)
        _copy_blocks(a, 0, out, 0, a.shape[1] * 8)
        _copy_blocks(b, 0, out, a.shape[1] * 8, b.shape[1] * 8)
        ctx.ca, ctx.cb = a.shape[1], b.shape[1]
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, _, H, W, _ = dy.shape
        da = torch.empty((B, ctx.ca, H, W, 8), dtype=dy.dtype, device=dy.device)
        db = torch.empty((B, ctx.cb, H, W, 8), dtype=dy.dtype, device=dy.device)
        _copy_blocks(dy, 0, da, 0, ctx.ca * 8)
        _copy_blocks(dy, ctx.ca * 8, db, 0, ctx.cb * 8)
        return da, db


class _FewOutTrainFn(torch.autograd.Function):
    """The last layer on a packed tensor plus the bilinear x2 of the network input (model/Restromer.py:381-383, 420-431),
    fp32 result; backward: the fp32 output gradient goes into the packed bf16 layout (3 channels in one block) and
    through the same 3x3 backward as every other convolution."""

    @staticmethod
    def forward(ctx, xdata, cin, weight, bias, lowres):
        x = Packed(_packed_data(xdata, "last layer"), cin)
        ctx.save_for_backward(x.data, weight)
        ctx.cin, ctx.has_bias = cin, bias is not None
        return conv3x3_fewout(x, weight, bias, lowres=lowres)

    @staticmethod
    def backward(ctx, dy):
        xdata, weight = ctx.saved_tensors
        dx, dw, db = conv3x3_backward(Packed(xdata, ctx.cin), pack(dy.float().contiguous()), weight, 0, ctx.needs_input_grad[0],
                                      ctx.has_bias)
        return (dx.data if dx is not None else None), None, dw.view_as(weight), db, None


def conv3x3_train(xdata, cin, conv, mode=0):
    return _Conv3x3TrainFn.apply(xdata, cin, conv.weight, conv.bias, mode)


def conv1x1_train(xdata, cin, conv):
    return _Conv1x1TrainFn.apply(xdata, cin, conv.weight, conv.bias)


def cat_train(a, b):
    return _CatTrainFn.apply(a, b)


def fewout_train(xdata, cin, conv, lowres):
    return _FewOutTrainFn.apply(xdata, cin, conv.weight, conv.bias, lowres)
