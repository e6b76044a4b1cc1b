"""Host-side operators: torch tensors in, HIP kernels through the C ABI, autograd glue.

torch is used for device memory (outputs, workspaces, saved buffers) and for the
current stream; all arithmetic of these operators happens in
librestormer_hip.so.  Every function requires fp32 tensors on a ROCm device and
raises RuntimeError otherwise.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Sequence

import torch

from . import _lib
from ._lib import BlockDesc, BlockGrads, BlockParams, PARAM_FIELDS, check


# ----------------------------------------------------------------------------
# plumbing
# ----------------------------------------------------------------------------
def _require(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; this operator only runs on a ROCm GPU "
                           "(there is no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: dtype {t.dtype} not supported (fp32 only)")
    return t.contiguous()


def _opt(t: Optional[torch.Tensor], name: str) -> Optional[torch.Tensor]:
    return None if t is None else _require(t, name)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_workspaces: Dict[tuple, torch.Tensor] = {}


def workspace(nbytes: int, device: torch.device) -> torch.Tensor:
    """Grow-only scratch buffer, one per (device, stream).  Kernels enqueued on
    one stream run in order, so consecutive operators may share it."""
    key = (device.index, _stream())
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _workspaces.pop(key, None)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def release_workspaces() -> None:
    _workspaces.clear()


def trim_workspaces(keep_bytes: Optional[int] = None) -> None:
    """Hands scratch buffers above `keep_bytes` (default 4 GiB, ``RST_WORKSPACE_KEEP_GB``) back to torch's allocator:
    a 4K frame needs a multi-GiB scratch that a process serving small crops afterwards should not keep.  Safe between
    calls: a buffer is only ever used on the stream it is keyed on, which is also the stream it was allocated on."""
    if keep_bytes is None:
        import os
        keep_bytes = int(float(os.environ.get("RST_WORKSPACE_KEEP_GB", "4")) * (1 << 30))
    for key in [k for k, b in _workspaces.items() if b.numel() > keep_bytes]:
        del _workspaces[key]


def make_desc(B, C, H, W, heads, hidden, ln_with_bias, conv_bias) -> BlockDesc:
    return BlockDesc(int(B), int(C), int(H), int(W), int(heads), int(hidden), int(bool(ln_with_bias)), int(bool(conv_bias)))


def _fill(struct, tensors: Dict[str, Optional[torch.Tensor]]):
    for name in PARAM_FIELDS:
        t = tensors.get(name)
        setattr(struct, name, None if t is None else t.data_ptr())
    return struct


# ----------------------------------------------------------------------------
# LayerNorm (reference: LayerNorm.forward, model/Restromer.py:72-74)
# ----------------------------------------------------------------------------
class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        x = _require(x, "layernorm x")
        weight, bias = _require(weight, "layernorm weight"), _opt(bias, "layernorm bias")
        B, C, H, W = x.shape
        if weight.numel() != C or (bias is not None and bias.numel() != C):
            raise RuntimeError(f"layernorm: weight/bias must have {C} elements")
        y = torch.empty_like(x)
        check(lib.rst_layernorm_fwd(_ptr(x), _ptr(weight), _ptr(bias), _ptr(y), B, C, H * W, _stream()), "rst_layernorm_fwd")
        ctx.save_for_backward(x, weight)
        ctx.with_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        dy = _require(dy, "layernorm dy")
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        dw = torch.empty_like(weight)
        db = torch.empty_like(weight) if ctx.with_bias else None
        nb = lib.rst_layernorm_bwd_workspace_bytes(B, C, H * W)
        ws = workspace(nb, x.device)
        check(lib.rst_layernorm_bwd(_ptr(x), _ptr(weight), int(ctx.with_bias), _ptr(dy), _ptr(dx), 0, _ptr(dw), _ptr(db),
                                    B, C, H * W, _ptr(ws), ws.numel(), _stream()), "rst_layernorm_bwd")
        return dx, dw, db


def layernorm(x, weight, bias=None):
    return _LayerNormFn.apply(x, weight, bias)


# ----------------------------------------------------------------------------
# 1x1 convolution (reference: nn.Conv2d(kernel_size=1), model/Restromer.py:309-311,326-328)
# ----------------------------------------------------------------------------
class _Conv1x1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        x = _require(x, "conv1x1 x")
        w = _require(weight, "conv1x1 weight")
        bias = _opt(bias, "conv1x1 bias")
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        if w.numel() != Cout * Cin or (bias is not None and bias.numel() != Cout):
            raise RuntimeError(f"conv1x1: weight {tuple(w.shape)} / bias do not match {Cin} input channels")
        y = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device)
        check(lib.rst_conv1x1_fwd(_ptr(x), _ptr(w), _ptr(bias), None, _ptr(y), B, Cin, Cout, H * W, _stream()), "rst_conv1x1_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        dy = _require(dy, "conv1x1 dy")
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty(Cout, dtype=x.dtype, device=x.device) if ctx.has_bias else None
        nb = lib.rst_conv1x1_bwd_workspace_bytes(B, Cin, Cout, H * W)
        ws = workspace(nb, x.device)
        check(lib.rst_conv1x1_bwd(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), 0, _ptr(dw), _ptr(db), B, Cin, Cout, H * W,
                                  _ptr(ws), ws.numel(), _stream()), "rst_conv1x1_bwd")
        return dx, dw, db


def conv1x1(x, weight, bias=None):
    return _Conv1x1Fn.apply(x, weight, bias)


class _Conv1x1CatFn(torch.autograd.Function):
    """conv1x1(torch.cat([a, b], 1), weight, bias) without the concatenated tensor (reference: model/Restromer.py:402-403,
    407-408).  The backward pass goes through rst_conv1x1_bwd on the concatenation (built there only)."""

    @staticmethod
    def forward(ctx, a, b, weight, bias):
        lib = _lib.load()
        a, b = _require(a, "conv1x1_cat a"), _require(b, "conv1x1_cat b")
        w = _require(weight, "conv1x1_cat weight")
        bias = _opt(bias, "conv1x1_cat bias")
        B, Ca, H, W = a.shape
        Cb, Cout = b.shape[1], w.shape[0]
        if tuple(b.shape) != (B, Cb, H, W) or w.numel() != Cout * (Ca + Cb) or (bias is not None and bias.numel() != Cout):
            raise RuntimeError(f"conv1x1_cat: shapes {tuple(a.shape)}, {tuple(b.shape)}, weight {tuple(w.shape)} do not fit")
        y = torch.empty((B, Cout, H, W), dtype=a.dtype, device=a.device)
        ws = workspace(lib.rst_conv1x1_cat_workspace_bytes(Ca, Cb, Cout), a.device)
        check(lib.rst_conv1x1_cat_fwd(_ptr(a), _ptr(b), _ptr(w), _ptr(bias), _ptr(y), B, Ca, Cb, Cout, H * W, _ptr(ws), ws.numel(),
                                      _stream()), "rst_conv1x1_cat_fwd")
        ctx.save_for_backward(a, b, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        a, b, w = ctx.saved_tensors
        dy = _require(dy, "conv1x1_cat dy")
        x = torch.cat([a, b], 1)
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty(Cout, dtype=x.dtype, device=x.device) if ctx.has_bias else None
        nb = lib.rst_conv1x1_bwd_workspace_bytes(B, Cin, Cout, H * W)
        ws = workspace(nb, x.device)
        check(lib.rst_conv1x1_bwd(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), 0, _ptr(dw), _ptr(db), B, Cin, Cout, H * W,
                                  _ptr(ws), ws.numel(), _stream()), "rst_conv1x1_bwd")
        return dx[:, :a.shape[1]].contiguous(), dx[:, a.shape[1]:].contiguous(), dw, db


def conv1x1_cat(a, b, weight, bias=None):
    """1x1 convolution of the channel concatenation [a, b] that is never materialised."""
    return _Conv1x1CatFn.apply(a, b, weight, bias)


# ----------------------------------------------------------------------------
# depthwise 3x3 (reference: qkv_dwconv / dwconv, model/Restromer.py:87-95,116-124)
# ----------------------------------------------------------------------------
class _DwConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        x = _require(x, "dwconv x")
        w = _require(weight, "dwconv weight")
        bias = _opt(bias, "dwconv bias")
        B, C, H, W = x.shape
        if w.numel() != C * 9 or (bias is not None and bias.numel() != C):
            raise RuntimeError(f"dwconv: weight {tuple(w.shape)} / bias do not match {C} channels")
        y = torch.empty_like(x)
        check(lib.rst_dwconv3x3_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, C, H, W, _stream()), "rst_dwconv3x3_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        dy = _require(dy, "dwconv dy")
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty(C, dtype=x.dtype, device=x.device) if ctx.has_bias else None
        nb = lib.rst_dwconv3x3_bwd_workspace_bytes(B, C, H, W)
        ws = workspace(nb, x.device)
        check(lib.rst_dwconv3x3_bwd(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), _ptr(dw), _ptr(db), B, C, H, W,
                                    _ptr(ws), ws.numel(), _stream()), "rst_dwconv3x3_bwd")
        return dx, dw, db


def dwconv3x3(x, weight, bias=None):
    return _DwConvFn.apply(x, weight, bias)


# ----------------------------------------------------------------------------
# PixelUnshuffle(2) / PixelShuffle(2) (reference: model/Restromer.py:196,211)
# ----------------------------------------------------------------------------
def _unshuffle_raw(x):
    lib = _lib.load()
    x = _require(x, "pixel_unshuffle x")
    B, C, H, W = x.shape
    if H % 2 or W % 2:
        raise RuntimeError(f"pixel_unshuffle: H={H}, W={W} must be even")
    y = torch.empty((B, C * 4, H // 2, W // 2), dtype=x.dtype, device=x.device)
    check(lib.rst_pixel_unshuffle2(_ptr(x), _ptr(y), B, C, H, W, _stream()), "rst_pixel_unshuffle2")
    return y


def _shuffle_raw(x):
    lib = _lib.load()
    x = _require(x, "pixel_shuffle x")
    B, C4, H, W = x.shape
    if C4 % 4:
        raise RuntimeError(f"pixel_shuffle: channels {C4} not divisible by 4")
    y = torch.empty((B, C4 // 4, H * 2, W * 2), dtype=x.dtype, device=x.device)
    check(lib.rst_pixel_shuffle2(_ptr(x), _ptr(y), B, C4 // 4, H, W, _stream()), "rst_pixel_shuffle2")
    return y


class _PixelUnshuffleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _unshuffle_raw(x)

    @staticmethod
    def backward(ctx, dy):
        return _shuffle_raw(dy)


class _PixelShuffleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _shuffle_raw(x)

    @staticmethod
    def backward(ctx, dy):
        return _unshuffle_raw(dy)


def pixel_unshuffle2(x):
    return _PixelUnshuffleFn.apply(x)


def pixel_shuffle2(x):
    return _PixelShuffleFn.apply(x)


# ----------------------------------------------------------------------------
# dense 3x3 convolution with few output channels (+ residual): the last layer
# ----------------------------------------------------------------------------
def conv3x3_backward(x, dy, weight, mode: int, want_dx: bool, want_bias: bool):
    """Gradients of conv2d(x, weight, bias, padding=1) (+ PixelUnshuffle(2) / PixelShuffle(2) for mode 1 / 2) from the
    gradient of its output: (dx or None, dweight, dbias or None) through rst_conv3x3_bwd (exact-fp32 products); a shape
    it does not take (an odd width in front of a PixelUnshuffle, which the reference cannot run either) goes through
    torch's convolution gradients."""
    lib = _lib.load()
    x, dy, w = _require(x, "conv3x3 x"), _require(dy, "conv3x3 dy"), _require(weight, "conv3x3 weight")
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    nb = lib.rst_conv3x3_bwd_workspace_bytes(B, Cin, Cout, H, W, int(mode), int(want_dx), int(want_bias))
    if nb == 0:
        if mode == 1:
            dy = _shuffle_raw(dy)           # gradient of PixelUnshuffle is PixelShuffle
        elif mode == 2:
            dy = _unshuffle_raw(dy)
        dx = torch.nn.grad.conv2d_input(x.shape, w, dy, padding=1) if want_dx else None
        return dx, torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=1), (dy.sum(dim=(0, 2, 3)) if want_bias else None)
    ws = workspace(nb, x.device)
    dx = torch.empty_like(x) if want_dx else None
    dw = torch.empty_like(w)
    db = torch.empty(Cout, dtype=x.dtype, device=x.device) if want_bias else None
    check(lib.rst_conv3x3_bwd(_ptr(x), _ptr(dy), _ptr(w), _ptr(dx), _ptr(dw), _ptr(db), B, Cin, Cout, H, W, int(mode), _ptr(ws),
                              ws.numel(), _stream()), "rst_conv3x3_bwd")
    return dx, dw, db


class _Conv3x3FewOutFn(torch.autograd.Function):
    """y = conv2d(x, weight, bias, padding=1) (+ residual) for 1..4 output channels
    (reference: `self.output` and the `+ base` after it, model/Restromer.py:381-383,431).
    Forward and backward (rst_conv3x3_bwd) are HIP kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        lib = _lib.load()
        x = _require(x, "conv3x3 x")
        w = _require(weight, "conv3x3 weight")
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        if tuple(w.shape) != (Cout, Cin, 3, 3):
            raise RuntimeError(f"conv3x3: weight shape {tuple(w.shape)} does not match input channels {Cin}")
        res = None if residual is None else _require(residual, "conv3x3 residual")
        bias = _opt(bias, "conv3x3 bias")
        if bias is not None and bias.numel() != Cout:
            raise RuntimeError(f"conv3x3: bias must have {Cout} elements")
        if res is not None and tuple(res.shape) != (B, Cout, H, W):
            raise RuntimeError(f"conv3x3: residual shape {tuple(res.shape)} != {(B, Cout, H, W)}")
        y = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device)
        check(lib.rst_conv3x3_fewout_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(res), _ptr(y), B, Cin, Cout, H, W, _stream()),
              "rst_conv3x3_fewout_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx, dw, db = conv3x3_backward(x, dy, w, 0, ctx.needs_input_grad[0], ctx.has_bias and ctx.needs_input_grad[2])
        dr = dy if ctx.has_res and ctx.needs_input_grad[3] else None
        return dx, dw, db, dr


def conv3x3_fewout(x, weight, bias=None, residual=None):
    """3x3 convolution, padding 1, at most 4 output channels, optional residual add."""
    return _Conv3x3FewOutFn.apply(x, weight, bias, residual)


class _Conv3x3FewOutUpFn(torch.autograd.Function):
    """y = conv2d(x, weight, bias, padding=1) + bilinear x2 of `lowres` (reference: `self.output(...) + inp_img` with
    inp_img = F.interpolate(inp_img, scale_factor=2, mode="bilinear", align_corners=False), model/Restromer.py:420-431):
    the residual is computed inside the kernel from the low-resolution input, never materialised.  `lowres` gets no
    gradient (it is the network's input)."""

    @staticmethod
    def forward(ctx, x, weight, bias, lowres):
        lib = _lib.load()
        x = _require(x, "conv3x3 x")
        w = _require(weight, "conv3x3 weight")
        bias = _opt(bias, "conv3x3 bias")
        low = _require(lowres, "conv3x3 low-resolution residual")
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        if tuple(w.shape) != (Cout, Cin, 3, 3) or tuple(low.shape) != (B, Cout, H // 2, W // 2) or H % 2 or W % 2:
            raise RuntimeError(f"conv3x3: weight {tuple(w.shape)} / low-resolution tensor {tuple(low.shape)} do not match x {tuple(x.shape)}")
        y = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device)
        check(lib.rst_conv3x3_fewout_up2_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(low), _ptr(y), B, Cin, Cout, H, W, _stream()),
              "rst_conv3x3_fewout_up2_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw, db = conv3x3_backward(x, dy.contiguous(), w, 0, ctx.needs_input_grad[0], ctx.has_bias and ctx.needs_input_grad[2])
        return dx, dw, db, None


def conv3x3_fewout_up2(x, weight, bias, lowres):
    """Last layer + bilinear x2 residual of the network input in one kernel (at most 4 output channels)."""
    return _Conv3x3FewOutUpFn.apply(x, weight, bias, lowres)


# ----------------------------------------------------------------------------
# dense 3x3 convolution with the pixel (un)shuffle as its epilogue
# ----------------------------------------------------------------------------
def conv3x3_supported(x, weight, mode: int) -> bool:
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    B, Cin, H, W = x.shape
    return bool(_lib.load().rst_conv3x3_supported(Cin, weight.shape[0], H, W, mode))


class _Conv3x3Fn(torch.autograd.Function):
    """conv2d(x, weight, bias, padding=1) followed by nothing (mode 0), nn.PixelUnshuffle(2) (mode 1) or
    nn.PixelShuffle(2) (mode 2): reference model/Restromer.py:181-183, 188-200, 203-215.  Forward is one HIP kernel,
    the backward rst_conv3x3_bwd."""

    @staticmethod
    def forward(ctx, x, weight, bias, mode):
        lib = _lib.load()
        x = _require(x, "conv3x3 x")
        w = _require(weight, "conv3x3 weight")
        bias = _opt(bias, "conv3x3 bias")
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        if tuple(w.shape) != (Cout, Cin, 3, 3) or (bias is not None and bias.numel() != Cout):
            raise RuntimeError(f"conv3x3: weight {tuple(w.shape)} / bias do not match {Cin} input channels")
        shape = {0: (B, Cout, H, W), 1: (B, 4 * Cout, H // 2, W // 2), 2: (B, Cout // 4, 2 * H, 2 * W)}[mode]
        y = torch.empty(shape, dtype=x.dtype, device=x.device)
        nb = lib.rst_conv3x3_fwd_workspace_bytes(B, Cin, Cout, H, W, int(mode))
        if nb == 0:
            raise RuntimeError(f"conv3x3: shape Cin={Cin} Cout={Cout} {H}x{W} mode={mode} is not supported")
        ws = workspace(nb, x.device)
        check(lib.rst_conv3x3_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, Cin, Cout, H, W, int(mode), _ptr(ws), ws.numel(),
                                  _stream()), "rst_conv3x3_fwd")
        ctx.save_for_backward(x, w)
        ctx.mode, ctx.has_bias = mode, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw, db = conv3x3_backward(x, dy.contiguous(), w, ctx.mode, ctx.needs_input_grad[0], ctx.has_bias and ctx.needs_input_grad[2])
        return dx, dw, db, None


def conv3x3(x, weight, bias=None, mode: int = 0):
    """3x3 convolution (padding 1) + nothing / PixelUnshuffle(2) / PixelShuffle(2) in one kernel."""
    return _Conv3x3Fn.apply(x, weight, bias, mode)


# ----------------------------------------------------------------------------
# composite operators: MDTA, GDFN, TransformerBlock
# ----------------------------------------------------------------------------
ATTN_FIELDS = ("temperature", "qkv_weight", "qkv_bias", "qkv_dw_weight", "qkv_dw_bias", "proj_weight", "proj_bias")
FFN_FIELDS = ("pin_weight", "pin_bias", "ffn_dw_weight", "ffn_dw_bias", "pout_weight", "pout_bias")
NORM_FIELDS = ("norm1_weight", "norm1_bias", "norm2_weight", "norm2_bias")


def _composite(kind: str, fields: Sequence[str]):
    """Build an autograd Function for one of the composite C entry points.

    forward(x, heads, hidden, ln_with_bias, conv_bias, keep, *params) where
    params follow `fields` (None for absent biases) and `keep` says whether a
    backward pass can follow (see _keep): only then is the saved buffer
    allocated and filled."""
    fwd_name, bwd_name = f"rst_{kind}_fwd", f"rst_{kind}_bwd"

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, heads, hidden, ln_with_bias, conv_bias, keep, *params):
            lib = _lib.load()
            x = _require(x, f"{kind} x")
            B, C, H, W = x.shape
            if C % heads:
                raise RuntimeError(f"{kind}: channels {C} not divisible by heads {heads}")
            tensors = {n: (None if p is None else _require(p, f"{kind} {n}")) for n, p in zip(fields, params)}
            desc = make_desc(B, C, H, W, heads, hidden, ln_with_bias, conv_bias)
            par = _fill(BlockParams(), tensors)
            need_grad = bool(keep)
            saved = None
            if need_grad:
                nbytes = getattr(lib, f"rst_{kind}_saved_bytes")(ctypes.byref(desc))
                saved = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            ws = workspace(getattr(lib, f"rst_{kind}_workspace_bytes")(ctypes.byref(desc)), x.device)
            y = torch.empty_like(x)
            fn = getattr(lib, fwd_name)
            if kind == "block":
                st = fn(ctypes.byref(desc), ctypes.byref(par), _ptr(x), _ptr(y), _ptr(saved), _ptr(ws), ws.numel(), _stream())
            else:
                st = fn(ctypes.byref(desc), ctypes.byref(par), _ptr(x), None, _ptr(y), _ptr(saved), _ptr(ws), ws.numel(), _stream())
            check(st, fwd_name)
            if need_grad:
                ctx.desc_args = (B, C, H, W, heads, hidden, ln_with_bias, conv_bias)
                ctx.present = [p is not None for p in params]
                ctx.save_for_backward(x, saved, *[tensors[n] for n in fields if tensors[n] is not None])
            return y

        @staticmethod
        def backward(ctx, dy):
            lib = _lib.load()
            x, saved, *plist = ctx.saved_tensors
            dy = _require(dy, f"{kind} dy")
            desc = make_desc(*ctx.desc_args)
            it = iter(plist)
            tensors = {n: (next(it) if present else None) for n, present in zip(fields, ctx.present)}
            par = _fill(BlockParams(), tensors)
            grads = {n: (None if t is None else torch.empty_like(t)) for n, t in tensors.items()}
            gr = _fill(BlockGrads(), grads)
            ws = workspace(getattr(lib, f"rst_{kind}_bwd_workspace_bytes")(ctypes.byref(desc)), x.device)
            dx = torch.empty_like(x)
            fn = getattr(lib, bwd_name)
            if kind == "block":
                st = fn(ctypes.byref(desc), ctypes.byref(par), _ptr(x), _ptr(dy), _ptr(saved), _ptr(dx), ctypes.byref(gr),
                        _ptr(ws), ws.numel(), _stream())
            else:
                st = fn(ctypes.byref(desc), ctypes.byref(par), _ptr(x), _ptr(dy), _ptr(saved), _ptr(dx), 0, ctypes.byref(gr),
                        _ptr(ws), ws.numel(), _stream())
            check(st, bwd_name)
            return (dx, None, None, None, None, None) + tuple(grads[n] for n in fields)

    Fn.__name__ = f"_{kind.capitalize()}Fn"
    return Fn


_AttentionFn = _composite("attention", ATTN_FIELDS)
_FfnFn = _composite("ffn", FFN_FIELDS)
_BlockFn = _composite("block", PARAM_FIELDS)


def _keep(x, params) -> bool:
    """True when autograd will record the call.  This has to be decided outside
    Function.forward: inside it grad mode is always off, and ctx.needs_input_grad
    stays True for parameters even under torch.no_grad()."""
    return torch.is_grad_enabled() and (x.requires_grad or any(p is not None and p.requires_grad for p in params))


def attention(x, heads, temperature, qkv_weight, qkv_bias, qkv_dw_weight, qkv_dw_bias, proj_weight, proj_bias):
    """MDTA (reference: Attention.forward, model/Restromer.py:127-150)."""
    ps = (temperature, qkv_weight, qkv_bias, qkv_dw_weight, qkv_dw_bias, proj_weight, proj_bias)
    return _AttentionFn.apply(x, heads, 1, True, qkv_bias is not None, _keep(x, ps),
                              temperature, qkv_weight, qkv_bias, qkv_dw_weight, qkv_dw_bias, proj_weight, proj_bias)


def feed_forward(x, hidden, pin_weight, pin_bias, dw_weight, dw_bias, pout_weight, pout_bias):
    """GDFN (reference: FeedForward.forward, model/Restromer.py:99-104)."""
    ps = (pin_weight, pin_bias, dw_weight, dw_bias, pout_weight, pout_bias)
    return _FfnFn.apply(x, 1, hidden, True, pin_bias is not None, _keep(x, ps),
                        pin_weight, pin_bias, dw_weight, dw_bias, pout_weight, pout_bias)


# Every optimiser step of the process moves this counter (a global torch hook), and the prepared-weight caches carry
# it in their keys: torch's fused optimisers update parameters without moving their version counters, so a version
# alone would let an evaluation after `optimizer.step()` run on the weights of the previous preparation.
_optimizer_steps = [0]


def _count_optimizer_step(optimizer, args, kwargs):
    _optimizer_steps[0] += 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook
    _register_step_hook(_count_optimizer_step)
except ImportError:          # a torch without global optimiser hooks: the version counters and the mode switches remain
    pass


def weights_generation() -> int:
    """Number of optimiser steps this process has taken (part of every prepared-weight cache key)."""
    return _optimizer_steps[0]


class PreparedBlock:
    """The input-independent part of one block's fp32 inference (rst_block_prepare: LayerNorm-folded weights, operand
    fragments), kept between calls and rebuilt when a parameter tensor is replaced or modified in place (torch bumps
    ``_version`` on every in-place write).  A write through ``param.data`` does not move ``_version``: call
    :meth:`invalidate` after such a write."""
    __slots__ = ("key", "blob")

    def __init__(self):
        self.key, self.blob = None, None

    def invalidate(self):
        self.key, self.blob = None, None

    @staticmethod
    def _key(tensors, device):
        # (the prepared form depends on which instructions the products run on: rst_set_exact_fp32)
        return (tuple((n, t.data_ptr(), t._version) for n, t in tensors.items() if t is not None) +
                (str(device), int(_lib.load().rst_get_exact_fp32()), _optimizer_steps[0]))

    def is_current(self, tensors, device) -> bool:
        """True when the kept buffer was prepared from exactly these parameter tensors (nothing would be launched)."""
        return self.key is not None and self.key == self._key(tensors, device)

    def get(self, lib, desc, par, tensors, device):
        key = self._key(tensors, device)
        if key != self.key:
            blob = torch.empty(lib.rst_block_prepared_bytes(ctypes.byref(desc)), dtype=torch.uint8, device=device)
            check(lib.rst_block_prepare(ctypes.byref(desc), ctypes.byref(par), _ptr(blob), _stream()), "rst_block_prepare")
            self.key, self.blob = key, blob
        return self.blob


def transformer_block(x, heads, hidden, params: Dict[str, Optional[torch.Tensor]], cache: Optional[PreparedBlock] = None):
    """TransformerBlock (reference: model/Restromer.py:163-167).  `params` maps
    the names in PARAM_FIELDS to tensors (None for absent biases).  With a `cache` an inference call (no autograd
    graph wanted) reuses the block's prepared weights (rst_block_fwd_prepared)."""
    ln_with_bias = params.get("norm1_bias") is not None
    conv_bias = params.get("qkv_bias") is not None
    ps = [params.get(n) for n in PARAM_FIELDS]
    keep = _keep(x, ps)
    if cache is not None and not keep:
        lib = _lib.load()
        x = _require(x, "block x")
        B, C, H, W = x.shape
        if C % heads:
            raise RuntimeError(f"block: channels {C} not divisible by heads {heads}")
        tensors = {n: (None if params.get(n) is None else _require(params[n], f"block {n}")) for n in PARAM_FIELDS}
        desc = make_desc(B, C, H, W, heads, hidden, ln_with_bias, conv_bias)
        par = _fill(BlockParams(), tensors)
        prepared = cache.get(lib, desc, par, {n: params.get(n) for n in PARAM_FIELDS}, x.device)
        ws = workspace(lib.rst_block_workspace_bytes(ctypes.byref(desc)), x.device)
        y = torch.empty_like(x)
        check(lib.rst_block_fwd_prepared(ctypes.byref(desc), ctypes.byref(par), _ptr(prepared), _ptr(x), _ptr(y), _ptr(ws),
                                         ws.numel(), _stream()), "rst_block_fwd_prepared")
        return y
    return _BlockFn.apply(x, heads, hidden, ln_with_bias, conv_bias, keep, *ps)


def block_forward_inplace(x, heads, hidden, params: Dict[str, Optional[torch.Tensor]]):
    """Inference-only in-place block update (no autograd, no saved buffer)."""
    lib = _lib.load()
    x = _require(x, "block x")
    B, C, H, W = x.shape
    tensors = {n: (None if params.get(n) is None else _require(params[n], n)) for n in PARAM_FIELDS}
    desc = make_desc(B, C, H, W, heads, hidden, tensors["norm1_bias"] is not None, tensors["qkv_bias"] is not None)
    par = _fill(BlockParams(), tensors)
    ws = workspace(lib.rst_block_workspace_bytes(ctypes.byref(desc)), x.device)
    check(lib.rst_block_fwd(ctypes.byref(desc), ctypes.byref(par), _ptr(x), _ptr(x), None, _ptr(ws), ws.numel(), _stream()),
          "rst_block_fwd")
    return x


def block_recompute_backward(x, dy, heads, hidden, params: Dict[str, Optional[torch.Tensor]]):
    """Gradients of one TransformerBlock from its INPUT alone: the training forward (rst_block_fwd with a saved
    buffer) is run again on `x`, then rst_block_bwd.  Returns (dx, {parameter name: gradient or None}); used by
    training paths that keep only block inputs (activation recomputation)."""
    lib = _lib.load()
    x = _require(x, "block x")
    dy = _require(dy, "block dy")
    B, C, H, W = x.shape
    tensors = {n: (None if params.get(n) is None else _require(params[n].detach(), n)) for n in PARAM_FIELDS}
    desc = make_desc(B, C, H, W, heads, hidden, tensors["norm1_bias"] is not None, tensors["qkv_bias"] is not None)
    par = _fill(BlockParams(), tensors)
    saved = torch.empty(lib.rst_block_saved_bytes(ctypes.byref(desc)), dtype=torch.uint8, device=x.device)
    nws = max(lib.rst_block_workspace_bytes(ctypes.byref(desc)), lib.rst_block_bwd_workspace_bytes(ctypes.byref(desc)))
    ws = workspace(nws, x.device)
    y = torch.empty_like(x)
    check(lib.rst_block_fwd(ctypes.byref(desc), ctypes.byref(par), _ptr(x), _ptr(y), _ptr(saved), _ptr(ws), ws.numel(), _stream()),
          "rst_block_fwd")
    grads = {n: (None if t is None else torch.empty_like(t)) for n, t in tensors.items()}
    gr = _fill(BlockGrads(), grads)
    dx = torch.empty_like(x)
    check(lib.rst_block_bwd(ctypes.byref(desc), ctypes.byref(par), _ptr(x), _ptr(dy), _ptr(saved), _ptr(dx), ctypes.byref(gr),
                            _ptr(ws), ws.numel(), _stream()), "rst_block_bwd")
    return dx, grads


def _shard_args(x, heads, hidden, params):
    lib = _lib.load()
    x = _require(x, "block x")
    B, C, H, W = x.shape
    tensors = {n: (None if params.get(n) is None else _require(params[n], n)) for n in PARAM_FIELDS}
    desc = make_desc(B, C, H, W, heads, hidden, tensors["norm1_bias"] is not None, tensors["qkv_bias"] is not None)
    par = _fill(BlockParams(), tensors)
    ws = workspace(lib.rst_block_shard_workspace_bytes(ctypes.byref(desc)), x.device)
    return lib, x, desc, par, ws, tensors


def block_shard_stats(x, row0: int, row1: int, heads, hidden, params: Dict[str, Optional[torch.Tensor]]) -> torch.Tensor:
    """Phase 1 of a TransformerBlock on a row slab of a larger image (inference): this device's share of the
    attention's sums over all pixels (Gram matrices and squared norms of q and k, taken over rows [row0, row1)).
    The caller adds the returned vector over the devices and hands it to `block_shard_apply`, with no other
    operator of this library on the same stream in between (the slab's qkv waits in the shared workspace)."""
    lib, x, desc, par, ws, keep = _shard_args(x, heads, hidden, params)
    stats = torch.empty(lib.rst_block_shard_stats_floats(ctypes.byref(desc)), dtype=torch.float32, device=x.device)
    check(lib.rst_block_shard_fwd_stats(ctypes.byref(desc), ctypes.byref(par), _ptr(x), int(row0), int(row1), _ptr(stats),
                                        _ptr(ws), ws.numel(), _stream()), "rst_block_shard_fwd_stats")
    return stats


def block_shard_apply(x, stats, heads, hidden, params: Dict[str, Optional[torch.Tensor]]) -> torch.Tensor:
    """Phase 2: the block applied to the slab in place, given the sums over all devices.  Rows closer than two
    rows to a cut are not valid afterwards."""
    lib, x, desc, par, ws, keep = _shard_args(x, heads, hidden, params)
    stats = _require(stats, "stats")
    if stats.numel() != lib.rst_block_shard_stats_floats(ctypes.byref(desc)):
        raise RuntimeError(f"stats has {stats.numel()} elements, expected {lib.rst_block_shard_stats_floats(ctypes.byref(desc))}")
    check(lib.rst_block_shard_fwd_apply(ctypes.byref(desc), ctypes.byref(par), _ptr(x), _ptr(stats), _ptr(x),
                                        _ptr(ws), ws.numel(), _stream()), "rst_block_shard_fwd_apply")
    return x
