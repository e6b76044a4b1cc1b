"""Drop-in module surface of the reference's ``model/Restromer.py`` on MI355X.

Same class names, constructor signatures, parameter names/shapes/registration
order and forward signatures as the reference (citations per class), so a
reference ``state_dict`` loads with ``strict=True`` and optimiser state keeps
its parameter order.  The parameters live in ordinary ``nn.Conv2d`` /
``nn.Parameter`` containers (identical default initialisation), but no forward
here runs a torch convolution for the TransformerBlock path: every block,
LayerNorm, attention, feed-forward and pixel (un)shuffle goes through the
hand-written HIP kernels in ``librestormer_hip.so``.  Inputs must be fp32
tensors on a ROCm device; anything else raises ``RuntimeError``.
"""
from __future__ import annotations

import numbers

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def _flat(conv: nn.Conv2d):
    return conv.weight, conv.bias


# ----------------------------------------------------------------------------
# (b c h w) <-> (b hw c) views, reference model/Restromer.py:20-25.  Kept for API
# compatibility; the kernels normalise NCHW directly and never need them.
# ----------------------------------------------------------------------------
def to_3d(x):
    b, c, h, w = x.shape
    return x.reshape(b, c, h * w).transpose(1, 2)


def to_4d(x, h, w):
    b, hw, c = x.shape
    return x.transpose(1, 2).reshape(b, c, h, w)


class _ChannelNormBase(nn.Module):
    """Holds ``weight`` (and ``bias``) of shape (C,).  ``forward`` follows the
    reference bodies, which take the (b, hw, c) view (model/Restromer.py:40-42,
    58-61); LayerNorm below calls the NCHW kernel directly instead."""
    with_bias = False

    def __init__(self, normalized_shape):
        super().__init__()
        if isinstance(normalized_shape, numbers.Integral):
            normalized_shape = (normalized_shape,)
        normalized_shape = torch.Size(normalized_shape)
        if len(normalized_shape) != 1:
            raise AssertionError("channel LayerNorm takes a single normalised dimension")
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        if self.with_bias:
            self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.normalized_shape = normalized_shape

    def forward(self, x):
        # x: (b, hw, c)  ->  NCHW with W = 1, run the kernel, and back
        b, n, c = x.shape
        y = ops.layernorm(x.transpose(1, 2).reshape(b, c, n, 1), self.weight, self.bias if self.with_bias else None)
        return y.reshape(b, c, n).transpose(1, 2)


class BiasFree_LayerNorm(_ChannelNormBase):      # reference model/Restromer.py:28-42
    with_bias = False


class WithBias_LayerNorm(_ChannelNormBase):      # reference model/Restromer.py:45-61
    with_bias = True


class LayerNorm(nn.Module):
    """reference model/Restromer.py:64-74"""

    def __init__(self, dim, LayerNorm_type):
        super().__init__()
        self.body = BiasFree_LayerNorm(dim) if LayerNorm_type == "BiasFree" else WithBias_LayerNorm(dim)

    def forward(self, x):
        return ops.layernorm(x, self.body.weight, self.body.bias if self.body.with_bias else None)


class FeedForward(nn.Module):
    """GDFN, reference model/Restromer.py:79-104"""

    def __init__(self, dim, ffn_expansion_factor, bias):
        super().__init__()
        hidden_features = int(dim * ffn_expansion_factor)
        self.hidden_features = hidden_features
        self.project_in = nn.Conv2d(dim, hidden_features * 2, kernel_size=1, bias=bias)
        self.dwconv = nn.Conv2d(hidden_features * 2, hidden_features * 2, kernel_size=3, stride=1, padding=1,
                                groups=hidden_features * 2, bias=bias)
        self.project_out = nn.Conv2d(hidden_features, dim, kernel_size=1, bias=bias)

    def forward(self, x):
        return ops.feed_forward(x, self.hidden_features, *_flat(self.project_in), *_flat(self.dwconv),
                                *_flat(self.project_out))


class Attention(nn.Module):
    """MDTA, reference model/Restromer.py:109-150"""

    def __init__(self, dim, num_heads, bias):
        super().__init__()
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.qkv = nn.Conv2d(dim, dim * 3, kernel_size=1, bias=bias)
        self.qkv_dwconv = nn.Conv2d(dim * 3, dim * 3, kernel_size=3, stride=1, padding=1, groups=dim * 3, bias=bias)
        self.project_out = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)

    def forward(self, x):
        return ops.attention(x, self.num_heads, self.temperature, *_flat(self.qkv), *_flat(self.qkv_dwconv),
                             *_flat(self.project_out))


class TransformerBlock(nn.Module):
    """reference model/Restromer.py:154-167: x += attn(norm1(x)); x += ffn(norm2(x)).
    One fused call per direction (rst_block_fwd / rst_block_bwd)."""

    def __init__(self, dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type):
        super().__init__()
        self.norm1 = LayerNorm(dim, LayerNorm_type)
        self.attn = Attention(dim, num_heads, bias)
        self.norm2 = LayerNorm(dim, LayerNorm_type)
        self.ffn = FeedForward(dim, ffn_expansion_factor, bias)

    def _params(self):
        n1, n2, a, f = self.norm1.body, self.norm2.body, self.attn, self.ffn
        return {
            "norm1_weight": n1.weight, "norm1_bias": n1.bias if n1.with_bias else None,
            "temperature": a.temperature,
            "qkv_weight": a.qkv.weight, "qkv_bias": a.qkv.bias,
            "qkv_dw_weight": a.qkv_dwconv.weight, "qkv_dw_bias": a.qkv_dwconv.bias,
            "proj_weight": a.project_out.weight, "proj_bias": a.project_out.bias,
            "norm2_weight": n2.weight, "norm2_bias": n2.bias if n2.with_bias else None,
            "pin_weight": f.project_in.weight, "pin_bias": f.project_in.bias,
            "ffn_dw_weight": f.dwconv.weight, "ffn_dw_bias": f.dwconv.bias,
            "pout_weight": f.project_out.weight, "pout_bias": f.project_out.bias,
        }

    def invalidate_prepared(self):
        for name in ("_pk_cache", "_f32_cache"):
            cache = getattr(self, name, None)
            if cache is not None:
                cache.invalidate()

    def train(self, mode: bool = True):
        # The prepared weights are keyed on the parameters' version counters, which fused optimisers and writes through
        # .data do not move.  Every switch between training and evaluation (model.eval() in front of a validation
        # pass, model.train() after it) drops them, so an evaluation never runs on the weights of an earlier step.
        self.invalidate_prepared()
        return super().train(mode)

    def forward(self, x):
        if getattr(self, "_f32_cache", None) is None:
            self._f32_cache = ops.PreparedBlock()
        return ops.transformer_block(x, self.attn.num_heads, self.ffn.hidden_features, self._params(), self._f32_cache)


class OverlapPatchEmbed(nn.Module):
    """3x3 embedding conv, reference model/Restromer.py:172-183"""

    def __init__(self, in_c=3, embed_dim=48, bias=False):
        super().__init__()
        self.proj = nn.Conv2d(in_c, embed_dim, kernel_size=3, stride=1, padding=1, bias=bias)

    def forward(self, x):
        if ops.conv3x3_supported(x, self.proj.weight, 0):
            return ops.conv3x3(x, self.proj.weight, self.proj.bias, 0)
        return self.proj(x)            # not an fp32 ROCm tensor (a CPU copy of the module): torch's convolution


class _Permute(nn.Module):
    """Parameter-free stand-in that keeps index 1 of ``body`` occupied, so the
    conv stays ``body.0`` as in the reference's nn.Sequential."""

    def __init__(self, fn):
        super().__init__()
        self._fn = fn

    def forward(self, x):
        return self._fn(x)


class Downsample(nn.Module):
    """conv3x3 C -> C/2, PixelUnshuffle(2); reference model/Restromer.py:188-200"""

    def __init__(self, n_feat):
        super().__init__()
        self.body = nn.Sequential(
            nn.Conv2d(n_feat, n_feat // 2, kernel_size=3, stride=1, padding=1, bias=False),
            _Permute(ops.pixel_unshuffle2))

    def forward(self, x):
        conv = self.body[0]
        if ops.conv3x3_supported(x, conv.weight, 1):
            return ops.conv3x3(x, conv.weight, conv.bias, 1)        # convolution and PixelUnshuffle(2) in one kernel
        return self.body(x)                # not an fp32 ROCm tensor (a CPU copy of the module): torch's convolution


class Upsample(nn.Module):
    """conv3x3 C -> 2C, PixelShuffle(2); reference model/Restromer.py:203-215"""

    def __init__(self, n_feat):
        super().__init__()
        self.body = nn.Sequential(
            nn.Conv2d(n_feat, n_feat * 2, kernel_size=3, stride=1, padding=1, bias=False),
            _Permute(ops.pixel_shuffle2))

    def forward(self, x):
        conv = self.body[0]
        if ops.conv3x3_supported(x, conv.weight, 2):
            return ops.conv3x3(x, conv.weight, conv.bias, 2)        # convolution and PixelShuffle(2) in one kernel
        return self.body(x)                # not an fp32 ROCm tensor (a CPU copy of the module): torch's convolution


_SIDE_STREAMS = {}     # device index -> the second stream of two-stream inference (Restormer._forward_two_streams)


class Restormer(nn.Module):
    """U-shaped x2 super-resolution network, reference model/Restromer.py:220-433."""

    def __init__(self, inp_channels=4, out_channels=4, dim=12, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4,
                 heads=[1, 2, 4, 8], ffn_expansion_factor=1.25, bias=False, LayerNorm_type="WithBias",
                 dual_pixel_task=False):
        super().__init__()

        def stage(width, n_heads, depth):
            return nn.Sequential(*[TransformerBlock(dim=width, num_heads=n_heads, ffn_expansion_factor=ffn_expansion_factor,
                                                    bias=bias, LayerNorm_type=LayerNorm_type) for _ in range(depth)])

        w1, w2, w3, w4 = dim, dim * 2, dim * 4, dim * 8
        # registration order = reference order (optimizer state is positional)
        self.patch_embed = OverlapPatchEmbed(inp_channels, dim)
        self.encoder_level1 = stage(w1, heads[0], num_blocks[0])
        self.down1_2 = Downsample(w1)
        self.encoder_level2 = stage(w2, heads[1], num_blocks[1])
        self.down2_3 = Downsample(w2)
        self.encoder_level3 = stage(w3, heads[2], num_blocks[2])
        self.down3_4 = Downsample(w3)
        self.latent = stage(w4, heads[3], num_blocks[3])
        self.up4_3 = Upsample(w4)
        self.reduce_chan_level3 = nn.Conv2d(w4, w3, kernel_size=1, bias=bias)
        self.decoder_level3 = stage(w3, heads[2], num_blocks[2])
        self.up3_2 = Upsample(w3)
        self.reduce_chan_level2 = nn.Conv2d(w3, w2, kernel_size=1, bias=bias)
        self.decoder_level2 = stage(w2, heads[1], num_blocks[1])
        self.up2_1 = Upsample(w2)
        self.decoder_level1 = stage(w2, heads[0], num_blocks[0])
        self.refinement = stage(w2, heads[0], num_refinement_blocks)
        self.output_upscale = Upsample(w2)
        self.dual_pixel_task = dual_pixel_task
        if self.dual_pixel_task:
            self.skip_conv = nn.Conv2d(dim, w2, kernel_size=1, bias=bias)
        self.output = nn.Conv2d(dim, out_channels, kernel_size=3, stride=1, padding=1, bias=bias)

    def invalidate_prepared(self):
        """Drop the prepared weights of every block.  Happens by itself on every .train() / .eval() switch, .to() and
        after a DataParallelTrainer step; call it by hand after parameter writes that bypass torch's version counter
        (a fused optimiser, ``param.data``) when an evaluation follows WITHOUT such a switch."""
        for m in self.modules():
            if isinstance(m, TransformerBlock):
                m.invalidate_prepared()

    def _apply(self, fn, *args, **kwargs):
        # .to() / .float() / .cuda() may replace parameter storage behind the same version number
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_prepared()
        self._bf16_widths_ok = None
        return out

    def _forward_bf16(self, inp_img):
        """Inference with bf16 activations (a bf16 input, or an fp32 input under bf16 autocast): every
        TransformerBlock, 1x1 convolution and dense 3x3 convolution (PixelUnshuffle / PixelShuffle as its epilogue)
        runs on the library's channel-packed bf16 tensors with fp32 sums and fp32 parameters; the last layer and the
        bilinear residual run in fp32.  The result has the input's dtype."""
        from .. import bf16 as pk
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("the bf16 path is inference only: call it under torch.no_grad()")

        import os
        via_torch = os.environ.get("RST_BF16_CONV") == "torch"

        def conv3(p, conv, mode):
            if via_torch:
                # A/B aid: the convolution through torch (MIOpen) in bf16, channels-last, shuffle in the layout conversion
                bf, cl = torch.bfloat16, torch.channels_last
                src = pk.to_channels_last(p) if p.C % 8 == 0 else pk.unpack(p).contiguous(memory_format=cl)
                y = F.conv2d(src, conv.weight.detach().to(bf).contiguous(memory_format=cl),
                             None if conv.bias is None else conv.bias.detach().to(bf), padding=1)
                return pk.from_channels_last(y, mode)
            return pk.conv3x3(p, conv.weight, conv.bias, mode)

        def stage(p, blocks):
            for blk in blocks:
                if getattr(blk, "_pk_cache", None) is None:
                    blk._pk_cache = pk.PreparedCache()
                pk.block_(p, blk.attn.num_heads, blk.ffn.hidden_features, blk._params(), blk._pk_cache)
            return p

        def down(p, mod):        # conv3x3 then PixelUnshuffle(2)
            return conv3(p, mod.body[0], 1)

        def up(p, mod):          # conv3x3 then PixelShuffle(2)
            return conv3(p, mod.body[0], 2)

        e1_in = conv3(pk.pack(inp_img), self.patch_embed.proj, 0)
        skip = None
        if self.dual_pixel_task:
            skip = pk.unpack(pk.conv1x1(e1_in, self.skip_conv.weight, self.skip_conv.bias), torch.float32)
        e1 = stage(e1_in, self.encoder_level1)                       # in place: e1_in is consumed
        e2 = stage(down(e1, self.down1_2), self.encoder_level2)
        e3 = stage(down(e2, self.down2_3), self.encoder_level3)
        lat = stage(down(e3, self.down3_4), self.latent)
        d3 = pk.conv1x1(pk.cat(up(lat, self.up4_3), e3), self.reduce_chan_level3.weight, self.reduce_chan_level3.bias)
        d3 = stage(d3, self.decoder_level3)
        d2 = pk.conv1x1(pk.cat(up(d3, self.up3_2), e2), self.reduce_chan_level2.weight, self.reduce_chan_level2.bias)
        d2 = stage(d2, self.decoder_level2)
        d1 = stage(stage(pk.cat(up(d2, self.up2_1), e1), self.decoder_level1), self.refinement)
        upk = up(d1, self.output_upscale)
        if self.dual_pixel_task:
            out = self.output(pk.unpack(upk, torch.float32) + skip)
        elif self.output.weight.shape[0] <= 4 and self.output.weight.shape[0] == inp_img.shape[1] and not via_torch:
            # last layer + bilinear x2 of the input in one kernel that reads the packed tensor: fp32 weights and sums, and
            # the image never passes through a bf16 container on its way out
            out = pk.conv3x3_fewout(upk, self.output.weight, self.output.bias, lowres=inp_img.float())
        else:
            base = F.interpolate(inp_img.float(), scale_factor=2, mode="bilinear", align_corners=False)
            out = self.output(pk.unpack(upk, torch.float32)) + base
        return out.to(inp_img.dtype)

    def forward(self, inp_img):
        if inp_img.dim() != 4:
            raise RuntimeError(f"Restormer expects a 4-D NCHW tensor, got shape {tuple(inp_img.shape)}")
        if inp_img.shape[-2] % 8 or inp_img.shape[-1] % 8:
            raise RuntimeError(f"Restormer needs H and W divisible by 8, got {tuple(inp_img.shape[-2:])}")
        out = self._forward_two_streams(inp_img) if self._two_streams_wanted(inp_img) else self._forward_one(inp_img)
        if not torch.is_grad_enabled():
            ops.trim_workspaces()          # a frame-sized scratch goes back to the allocator, crop-sized ones stay
        return out

    # Inference of a batch on the bf16 path as two half-batches on two HIP streams.  The kernels of a block alternate
    # between ones that wait for issue slots / the matrix pipe (the fused 1x1 + depthwise kernels) and ones that wait
    # for memory (the residual products, the Gram sums); two independent launch sequences, a little out of step, fill
    # each other's gaps: 16 x 3 x 256^2 takes 16.5 instead of 17.6 ms, and halves that do not fill the chip lose
    # (2 x 256^2: 6.2 against 5.3 ms), hence the size bound.  The samples of a batch never meet (attention is per
    # sample), and the result is bit for bit that of the two half-batch calls.
    # Opt-in (model.two_streams = True, or RST_TWO_STREAMS=1 in the environment; RST_TWO_STREAMS=0: never), and never
    # for fp32: on MI355X a wave that issues v_mfma_f32_16x16x32_{f16,bf16} densely can make a wave of a kernel from
    # the OTHER stream on the same compute unit lose lanes 48..63 of one instruction of a dependent multiply-add chain
    # (tools/two_stream_repro.hip shows it without this library).  In fp32 that hit the gated depthwise kernel beside
    # the other half's attention product in most runs (up to 2e-2); the bf16 path has been bit-exact in every run so
    # far, which is an observation, not a guarantee.
    two_streams = False
    two_streams_min_pixels = 4 * 256 * 256  # input pixels per half-batch from which the split pays

    def _two_streams_wanted(self, inp_img) -> bool:
        import os
        env = os.environ.get("RST_TWO_STREAMS")
        if env == "0" or not (self.two_streams or env == "1"):
            return False
        if not inp_img.is_cuda or inp_img.shape[0] < 2 or torch.is_grad_enabled():
            return False
        if (inp_img.shape[0] // 2) * inp_img.shape[-2] * inp_img.shape[-1] < self.two_streams_min_pixels:
            return False
        if torch.cuda.is_current_stream_capturing():
            return False
        # Both halves read the blocks' prepared weights; a cache that is not current would be rebuilt by the first
        # half on ITS stream while the other reads it.  Such a call runs as one sequence (and leaves the caches current).
        bf16_path = (inp_img.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and
                                                          torch.get_autocast_dtype("cuda") == torch.bfloat16))
        if not (bf16_path and self._bf16_inference_ok(inp_img)):
            return False
        for m in self.modules():
            if isinstance(m, TransformerBlock):
                cache = getattr(m, "_pk_cache", None)
                if cache is None or not cache.is_current(m._params(), inp_img.device):
                    return False
        return True

    def _forward_two_streams(self, inp_img):
        cur = torch.cuda.current_stream(inp_img.device)
        side = _SIDE_STREAMS.get(inp_img.device.index)          # (not on the module: streams do not pickle or deep-copy)
        if side is None:
            side = _SIDE_STREAMS[inp_img.device.index] = torch.cuda.Stream(device=inp_img.device)
        first, second = inp_img.chunk(2)
        side.wait_stream(cur)                                   # the input (and the prepared weights) are ready there
        with torch.cuda.stream(side):
            out2 = self._forward_one(second)
        out1 = self._forward_one(first)
        cur.wait_stream(side)
        out2.record_stream(cur)
        return torch.cat([out1, out2])

    def _forward_one(self, inp_img):
        if inp_img.dtype == torch.bfloat16 or (inp_img.is_cuda and torch.is_autocast_enabled()
                                               and torch.get_autocast_dtype("cuda") == torch.bfloat16):
            if self._bf16_inference_ok(inp_img):
                return self._forward_bf16(inp_img)
            if self._bf16_widths_supported(inp_img):
                return self._forward_bf16_train(inp_img)
            # A channel width the packed bf16 kernels do not serve (e.g. the default dim=12), or an image whose width
            # is not a multiple of 16: the fp32 kernels
            # on an fp32 copy of the input, result in the input's dtype -- the reference module runs under
            # autocast through ordinary torch operators, so such a call must not fail.
            with torch.autocast("cuda", enabled=False):
                return self._forward_fp32_guarded(inp_img.float()).to(inp_img.dtype)
        return self._forward_fp32_guarded(inp_img)

    # The fp32 kernels multiply on the f16 matrix instructions with both operands split in two (fp32 accuracy), which
    # needs |activation| < 65504; beyond that a product returns inf / NaN, and non-finite values reach the output.  An
    # inference call therefore looks at its result once and, if it is not finite although the input was, runs again on
    # the exact-fp32 instructions (and keeps to them: `model.exact_fp32`).  The look is a device-to-host read of one
    # flag, i.e. a synchronisation: free when the call keeps the GPU busy longer than the host needs to queue it, not
    # free for small inputs -- range_check None (default) checks calls of at least 2^19 input pixels, True always,
    # False never; RST_RANGE_CHECK=0 / 1 in the environment overrides the attribute.
    range_check = None
    exact_fp32 = False

    def _range_check_wanted(self, inp_img) -> bool:
        if torch.is_grad_enabled() or not inp_img.is_cuda or torch.cuda.is_current_stream_capturing():
            return False
        env = os.environ.get("RST_RANGE_CHECK")
        want = self.range_check if env is None else env == "1"
        if want is None:
            want = inp_img.shape[0] * inp_img.shape[-2] * inp_img.shape[-1] >= (1 << 19)
        return bool(want)

    def _forward_fp32_guarded(self, inp_img):
        from .. import _lib
        lib = _lib.load() if inp_img.is_cuda else None
        if lib is None or lib.rst_get_exact_fp32():
            return self._forward_fp32(inp_img)
        if self.exact_fp32:
            lib.rst_set_exact_fp32(1)
            try:
                return self._forward_fp32(inp_img)
            finally:
                lib.rst_set_exact_fp32(0)
        out = self._forward_fp32(inp_img)
        if self._range_check_wanted(inp_img) and not bool(torch.isfinite(out).all()) and bool(torch.isfinite(inp_img).all()):
            import warnings
            warnings.warn("Restormer: an activation left the f16 range of the split-f16 products (|x| >= 65504); this model "
                          "now runs its fp32 products on the exact-fp32 instructions (model.exact_fp32 = True)")
            self.exact_fp32 = True
            return self._forward_fp32_guarded(inp_img)
        return out

    def _forward_bf16_train(self, inp_img):
        """Differentiable forward with bf16 activations (the bf16 training configuration: bf16 storage, fp32 sums, fp32 master
        parameters and gradients), on the library's packed tensors from the first layer to the last: every stage of
        TransformerBlocks is one autograd node that runs the packed bf16 kernels forward and, per block, either saves
        the intermediates its backward reads (while they fit) or keeps only the bf16 input and evaluates the block
        again (`bf16.stage`); the dense 3x3 convolutions with their
        pixel (un)shuffles, the skip concatenations and the 1x1 reductions between the stages are the library's
        kernels in both directions (`bf16.conv3x3_train`, `cat_train`, `conv1x1_train`); the last layer adds the
        bilinear x2 of the input inside its kernel and returns fp32.  Returns the input's dtype."""
        from .. import bf16 as pk
        if os.environ.get("RST_BF16_TRAIN_GLUE") == "torch" or self.dual_pixel_task or self.output.weight.shape[0] > 4 \
                or self.output.weight.shape[0] != inp_img.shape[1]:
            return self._forward_bf16_train_torch_glue(inp_img)
        dim = self.patch_embed.proj.out_channels
        with torch.autocast("cuda", enabled=False):
            x = pk.pack(inp_img.detach().float()).data          # the image needs no gradient
            e1 = pk.stage(pk.conv3x3_train(x, inp_img.shape[1], self.patch_embed.proj, 0), self.encoder_level1)
            e2 = pk.stage(pk.conv3x3_train(e1, dim, self.down1_2.body[0], 1), self.encoder_level2)
            e3 = pk.stage(pk.conv3x3_train(e2, 2 * dim, self.down2_3.body[0], 1), self.encoder_level3)
            lat = pk.stage(pk.conv3x3_train(e3, 4 * dim, self.down3_4.body[0], 1), self.latent)
            d3 = pk.cat_train(pk.conv3x3_train(lat, 8 * dim, self.up4_3.body[0], 2), e3)
            d3 = pk.stage(pk.conv1x1_train(d3, 8 * dim, self.reduce_chan_level3), self.decoder_level3)
            d2 = pk.cat_train(pk.conv3x3_train(d3, 4 * dim, self.up3_2.body[0], 2), e2)
            d2 = pk.stage(pk.conv1x1_train(d2, 4 * dim, self.reduce_chan_level2), self.decoder_level2)
            d1 = pk.cat_train(pk.conv3x3_train(d2, 2 * dim, self.up2_1.body[0], 2), e1)
            d1 = pk.stage(pk.stage(d1, self.decoder_level1), self.refinement)
            up = pk.conv3x3_train(d1, 2 * dim, self.output_upscale.body[0], 2)
            out = pk.fewout_train(up, dim, self.output, inp_img.detach().float())
        return out.to(inp_img.dtype)

    def _forward_bf16_train_torch_glue(self, inp_img):
        """The same network with the glue between the stages on torch's bf16 operators and their autograd (A/B aid,
        RST_BF16_TRAIN_GLUE=torch; also what serves dual_pixel_task and output widths the last-layer kernel does not)."""
        from .. import bf16 as pk
        bf = torch.bfloat16
        with torch.autocast("cuda", enabled=False):
            def conv(x, c):
                return F.conv2d(x, c.weight.to(bf), None if c.bias is None else c.bias.to(bf), padding=c.padding)

            x = inp_img.to(bf)
            e1_in = conv(x, self.patch_embed.proj)
            e1 = pk.stage(e1_in, self.encoder_level1)
            e2 = pk.stage(F.pixel_unshuffle(conv(e1, self.down1_2.body[0]), 2), self.encoder_level2)
            e3 = pk.stage(F.pixel_unshuffle(conv(e2, self.down2_3.body[0]), 2), self.encoder_level3)
            lat = pk.stage(F.pixel_unshuffle(conv(e3, self.down3_4.body[0]), 2), self.latent)
            d3 = torch.cat([F.pixel_shuffle(conv(lat, self.up4_3.body[0]), 2), e3], 1)
            d3 = pk.stage(conv(d3, self.reduce_chan_level3), self.decoder_level3)
            d2 = torch.cat([F.pixel_shuffle(conv(d3, self.up3_2.body[0]), 2), e2], 1)
            d2 = pk.stage(conv(d2, self.reduce_chan_level2), self.decoder_level2)
            d1 = torch.cat([F.pixel_shuffle(conv(d2, self.up2_1.body[0]), 2), e1], 1)
            d1 = pk.stage(pk.stage(d1, self.decoder_level1), self.refinement)
            up = F.pixel_shuffle(conv(d1, self.output_upscale.body[0]), 2).float()
            if self.dual_pixel_task:
                out = self.output(up + conv(e1_in, self.skip_conv).float())
            else:
                base = F.interpolate(inp_img.float(), scale_factor=2, mode="bilinear", align_corners=False)
                if self.output.weight.shape[0] <= 4:
                    out = ops.conv3x3_fewout(up, self.output.weight, self.output.bias, base)
                else:
                    out = self.output(up) + base
        return out.to(inp_img.dtype)

    def _bf16_inference_ok(self, inp_img) -> bool:
        """True when this call can run on the packed bf16 inference kernels: no autograd graph wanted, a ROCm
        tensor, and every stage's (width, heads, hidden) supported by rst_bf16_block_fwd (asked once per module)."""
        if torch.is_grad_enabled() and (inp_img.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False
        return self._bf16_widths_supported(inp_img)

    def _bf16_widths_supported(self, inp_img) -> bool:
        if not inp_img.is_cuda:
            return False
        if inp_img.shape[-1] % 16:
            return False           # the packed kernels take two pixels of a row per lane: the W / 8 level must be even
        ok = getattr(self, "_bf16_widths_ok", None)
        if ok is None:
            import ctypes
            from .. import _lib
            lib = _lib.load()
            ok = True
            for stage in (self.encoder_level1, self.encoder_level2, self.encoder_level3, self.latent, self.decoder_level3,
                          self.decoder_level2, self.decoder_level1, self.refinement):
                for blk in stage[:1]:
                    c = blk.attn.qkv.in_channels
                    desc = ops.make_desc(1, c, 16, 16, blk.attn.num_heads, blk.ffn.hidden_features,
                                         blk.norm1.body.with_bias, blk.attn.qkv.bias is not None)
                    ok = ok and c % 8 == 0 and lib.rst_bf16_block_workspace_bytes(ctypes.byref(desc)) != 0
            ok = ok and lib.rst_bf16_conv1x1_workspace_bytes(self.reduce_chan_level3.in_channels, self.reduce_chan_level3.out_channels) != 0
            for conv, mode in ((self.patch_embed.proj, 0), (self.down1_2.body[0], 1), (self.down2_3.body[0], 1), (self.down3_4.body[0], 1),
                               (self.up4_3.body[0], 2), (self.up3_2.body[0], 2), (self.up2_1.body[0], 2), (self.output_upscale.body[0], 2)):
                ok = ok and bool(lib.rst_bf16_conv3x3_supported(conv.in_channels, conv.out_channels, 16, 16, mode))
            self._bf16_widths_ok = bool(ok)
        return ok

    def _forward_fp32(self, inp_img):
        e1_in = self.patch_embed(inp_img)
        e1 = self.encoder_level1(e1_in)
        e2 = self.encoder_level2(self.down1_2(e1))
        e3 = self.encoder_level3(self.down2_3(e2))
        lat = self.latent(self.down3_4(e3))

        # (the concatenations in front of the 1x1 reductions are never written: the product reads both tensors)
        d3 = self.decoder_level3(ops.conv1x1_cat(self.up4_3(lat), e3, self.reduce_chan_level3.weight, self.reduce_chan_level3.bias))
        d2 = self.decoder_level2(ops.conv1x1_cat(self.up3_2(d3), e2, self.reduce_chan_level2.weight, self.reduce_chan_level2.bias))
        d1 = torch.cat([self.up2_1(d2), e1], 1)
        d1 = self.refinement(self.decoder_level1(d1))
        up = self.output_upscale(d1)

        if self.dual_pixel_task:
            # as written in the reference (model/Restromer.py:426-428)
            return self.output(up + ops.conv1x1(e1_in, self.skip_conv.weight, self.skip_conv.bias))
        if self.output.weight.shape[0] <= 4 and self.output.weight.shape[0] == inp_img.shape[1] and not inp_img.requires_grad:
            # last layer, bilinear x2 of the input and the residual add in one HIP kernel
            return ops.conv3x3_fewout_up2(up, self.output.weight, self.output.bias, inp_img)
        base = F.interpolate(inp_img, scale_factor=2, mode="bilinear", align_corners=False)
        if self.output.weight.shape[0] <= 4:
            return ops.conv3x3_fewout(up, self.output.weight, self.output.bias, base)
        return self.output(up) + base
