/*
 * restormer_hip.h -- C ABI of the MI355X (gfx950) Restormer TransformerBlock library.
 *
 * The reference (JagadeeshAI/SuperResolution) is pure Python: its "operator
 * interface" for this path is the set of nn.Module forwards in
 * model/Restromer.py.  Each entry point below replaces the forward (or the
 * autograd backward) of one of those modules and cites it.  The Python host
 * layer (superresolution_amd/model/Restromer.py) binds these symbols with
 * ctypes and re-exposes the reference's constructors and forward signatures.
 *
 * Conventions
 *  - All tensors are device pointers to contiguous fp32 NCHW data unless a
 *    stride argument says otherwise; "N" is H*W, the pixels of one sample.
 *  - Inputs are borrowed, outputs are written in place, nothing is retained
 *    after the call returns.  No entry point allocates or frees device memory,
 *    synchronises the device or the stream; scratch memory comes in through
 *    the `workspace` argument, sized by the matching *_workspace_bytes query.
 *  - `stream` is a hipStream_t (NULL = the default stream); all work of a call
 *    is enqueued on it in order.
 *  - Every function returns RST_OK or an error code; rst_last_error() gives the
 *    message of the last failure on the calling thread.  Arguments are
 *    validated before anything is launched.
 *  - Calls on different devices/streams are independent; a workspace must not
 *    be shared by calls that may overlap in time.
 *  - Keep the work of one GPU on ONE stream unless you have checked the overlap
 *    yourself.  Ordering inside a stream is sound on MI355X (a kernel trace of
 *    two busy streams shows no kernel starting before its predecessor ended),
 *    but a wave that issues v_mfma_f32_16x16x32_f16/_bf16 back to back can make
 *    a wave of a kernel from ANOTHER stream on the same compute unit lose lanes
 *    48..63 of one multiply-add of a dependent chain; most products here are
 *    such waves.  tools/two_stream_repro.hip shows it without this library.
 *  - One exception to "everything on `stream`, nothing created": with
 *    RST_BWD_STREAMS=1 in the environment (off by default, for the reason
 *    above) rst_bf16_block_bwd / _bwd_kept enqueue their parameter gradients on
 *    a second stream owned by the library (one per device, created with four
 *    events on the first such call: not capture-safe) and make `stream` wait
 *    for it before they return; concurrent callers on one device take turns.
 */
#ifndef RESTORMER_HIP_H
#define RESTORMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum rst_status {
    RST_OK = 0,
    RST_ERR_INVALID = 1,   /* bad shape / null pointer / unsupported size */
    RST_ERR_HIP = 2,       /* HIP runtime reported an error               */
    RST_ERR_WORKSPACE = 3  /* workspace pointer null or too small         */
};

/* Library ABI version (bumped on any signature change). */
int rst_version(void);
/* Message of the last failure on this thread ("" if none). */
const char* rst_last_error(void);
/* 1 if this process runs the parameter gradients of rst_bf16_block_bwd on the  */
/* library's own second stream (RST_BWD_STREAMS=1 in the environment when the   */
/* library first looked), 0 (the default) if every call keeps to its `stream`.  */
/* No counterpart in the reference, which never names a stream.                  */
int rst_bwd_side_stream_enabled(void);
/* The fp32 entry points run their large products on the f16 matrix instructions  */
/* with both operands split in two (22 significant bits; see gemm_f16s.hip), which */
/* needs |activation| < 65504: beyond it the results are inf / NaN.                */
/* rst_set_exact_fp32(1) keeps every product of this process on the exact-fp32     */
/* instruction from the next call on (slower; no range limit; weights prepared by   */
/* rst_block_prepare under the other setting must be prepared again), 0 switches    */
/* back; unset, RST_F32_MFMA=1 in the environment has the same effect.  No           */
/* counterpart in the reference, whose fp32 has no such limit.                      */
void rst_set_exact_fp32(int32_t on);
int32_t rst_get_exact_fp32(void);

/* ------------------------------------------------------------------------- */
/* Shapes and parameters of one TransformerBlock                              */
/* (model/Restromer.py:154-161 constructor; :110-125 Attention; :80-97 FFN).  */
/* ------------------------------------------------------------------------- */
typedef struct rst_block_desc {
    int32_t B, C, H, W;    /* input is B x C x H x W                              */
    int32_t heads;         /* num_heads; C % heads == 0                           */
    int32_t hidden;        /* int(C * ffn_expansion_factor), model/Restromer.py:83 */
    int32_t ln_with_bias;  /* 1: WithBias_LayerNorm, 0: BiasFree_LayerNorm        */
    int32_t conv_bias;     /* 1: the six convolutions carry a bias                */
} rst_block_desc;

/* Device pointers named after the reference state-dict entries of a block.
 * Bias pointers are ignored (may be NULL) when the matching flag is 0. */
typedef struct rst_block_params {
    const float* norm1_weight;    /* norm1.body.weight            [C]          */
    const float* norm1_bias;      /* norm1.body.bias              [C]          */
    const float* temperature;     /* attn.temperature             [heads]      */
    const float* qkv_weight;      /* attn.qkv.weight              [3C, C]      */
    const float* qkv_bias;        /* attn.qkv.bias                [3C]         */
    const float* qkv_dw_weight;   /* attn.qkv_dwconv.weight       [3C, 9]      */
    const float* qkv_dw_bias;     /* attn.qkv_dwconv.bias         [3C]         */
    const float* proj_weight;     /* attn.project_out.weight      [C, C]       */
    const float* proj_bias;       /* attn.project_out.bias        [C]          */
    const float* norm2_weight;    /* norm2.body.weight            [C]          */
    const float* norm2_bias;      /* norm2.body.bias              [C]          */
    const float* pin_weight;      /* ffn.project_in.weight        [2*hidden, C]*/
    const float* pin_bias;        /* ffn.project_in.bias          [2*hidden]   */
    const float* ffn_dw_weight;   /* ffn.dwconv.weight            [2*hidden, 9]*/
    const float* ffn_dw_bias;     /* ffn.dwconv.bias              [2*hidden]   */
    const float* pout_weight;     /* ffn.project_out.weight       [C, hidden]  */
    const float* pout_bias;       /* ffn.project_out.bias         [C]          */
} rst_block_params;

/* Same layout, writable: gradients of the parameters (overwritten, not accumulated). */
typedef struct rst_block_grads {
    float* norm1_weight;
    float* norm1_bias;
    float* temperature;
    float* qkv_weight;
    float* qkv_bias;
    float* qkv_dw_weight;
    float* qkv_dw_bias;
    float* proj_weight;
    float* proj_bias;
    float* norm2_weight;
    float* norm2_bias;
    float* pin_weight;
    float* pin_bias;
    float* ffn_dw_weight;
    float* ffn_dw_bias;
    float* pout_weight;
    float* pout_bias;
} rst_block_grads;

/* ------------------------------------------------------------------------- */
/* Channel LayerNorm -- replaces LayerNorm.forward, model/Restromer.py:72-74  */
/* (to_3d/to_4d :20-25 are views; WithBias :58-61, BiasFree :40-42).          */
/* bias == NULL selects the BiasFree flavour.  x, y: [B, C, N].               */
/* ------------------------------------------------------------------------- */
int rst_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y,
                      int32_t B, int32_t C, int64_t N, void* stream);

/* Backward of the above (autograd of model/Restromer.py:40-42,58-61).
 * dx is overwritten, or accumulated into when accumulate_dx != 0.
 * dweight [C] and dbias [C] (dbias ignored when with_bias == 0) are overwritten. */
size_t rst_layernorm_bwd_workspace_bytes(int32_t B, int32_t C, int64_t N);
int rst_layernorm_bwd(const float* x, const float* weight, int32_t with_bias, const float* dy,
                      float* dx, int32_t accumulate_dx, float* dweight, float* dbias,
                      int32_t B, int32_t C, int64_t N,
                      void* workspace, size_t workspace_bytes, void* stream);

/* The same backward pass for callers that cannot allocate device memory or
 * carry a workspace (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  The
 * per-block partial sums of dweight / dbias live in a 3 MiB array inside the
 * library, one copy per GPU the library runs on, held until the process ends;
 * the call allocates and frees nothing.  C <= 384 (RST_ERR_INVALID beyond).
 * All calls on one GPU share that array: calls that may overlap in time on
 * one GPU (different streams) race -- serialise them.  The first call on a
 * device looks the array's address up (hipGetSymbolAddress): make that call
 * before starting a stream capture.  Results are identical to
 * rst_layernorm_bwd. */
int rst_layernorm_bwd_noalloc(const float* x, const float* weight, int32_t with_bias, const float* dy,
                              float* dx, int32_t accumulate_dx, float* dweight, float* dbias,
                              int32_t B, int32_t C, int64_t N, void* stream);

/* ------------------------------------------------------------------------- */
/* 1x1 convolution -- replaces the nn.Conv2d(kernel_size=1) calls at          */
/* model/Restromer.py:100,103,130(qkv),149 and reduce_chan_level{3,2} :403,408.*/
/* y[b] = W x[b] (+ bias) (+ residual[b]);  x: [B, Cin, N], y: [B, Cout, N].  */
/* bias, residual may be NULL.                                                */
/* ------------------------------------------------------------------------- */
int rst_conv1x1_fwd(const float* x, const float* weight, const float* bias, const float* residual,
                    float* y, int32_t B, int32_t Cin, int32_t Cout, int64_t N, void* stream);

/* The same convolution applied to the channel concatenation of two tensors that is never written:
 * y = W [xa; xb] (+ bias) -- replaces torch.cat([up, skip], 1) followed by reduce_chan_level{3,2},
 * model/Restromer.py:402-403,407-408.  xa: [B, Ca, N], xb: [B, Cb, N], weight: [Cout, Ca + Cb].
 * The workspace holds the split weights (one product over both tensors when Ca is a multiple of 32). */
size_t rst_conv1x1_cat_workspace_bytes(int32_t Ca, int32_t Cb, int32_t Cout);
int rst_conv1x1_cat_fwd(const float* xa, const float* xb, const float* weight, const float* bias, float* y,
                        int32_t B, int32_t Ca, int32_t Cb, int32_t Cout, int64_t N,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Backward: dx = W^T dy (overwritten, or accumulated when accumulate_dx != 0; may be NULL to skip),
 * dweight [Cout, Cin] = sum_b dy[b] x[b]^T, dbias [Cout] (may be NULL). */
size_t rst_conv1x1_bwd_workspace_bytes(int32_t B, int32_t Cin, int32_t Cout, int64_t N);
int rst_conv1x1_bwd(const float* x, const float* weight, const float* dy,
                    float* dx, int32_t accumulate_dx, float* dweight, float* dbias,
                    int32_t B, int32_t Cin, int32_t Cout, int64_t N,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Dense 3x3 convolution, zero padding 1, with 1..4 output channels and an     */
/* optional residual added to the result -- replaces the network's last layer  */
/* `self.output` (nn.Conv2d(dim, out_channels, 3, padding=1),                  */
/* model/Restromer.py:381-383) and the `+ base` that follows it (:431).        */
/* x: [B, Cin, H, W]; weight: [Cout, Cin, 3, 3]; bias: [Cout] or NULL;         */
/* residual, y: [B, Cout, H, W] (residual may be NULL; y may alias residual).  */
/* Backward: rst_conv3x3_bwd with shuffle_mode 0 (it takes any Cout).           */
/* ------------------------------------------------------------------------- */
int rst_conv3x3_fewout_fwd(const float* x, const float* weight, const float* bias, const float* residual, float* y,
                           int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W, void* stream);
/* The same last layer with the reference's global residual computed inside: y = conv3x3(x) + bilinear x2 of the
 * network input (F.interpolate(inp_img, scale_factor=2, mode="bilinear", align_corners=False),
 * model/Restromer.py:420-422, added at :431).  lowres: [B, Cout, H/2, W/2]; H, W even. */
int rst_conv3x3_fewout_up2_fwd(const float* x, const float* weight, const float* bias, const float* lowres, float* y,
                               int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W, void* stream);

/* ------------------------------------------------------------------------- */
/* Dense 3x3 convolution, zero padding 1, with the pixel (un)shuffle that      */
/* follows it in the reference as its epilogue -- replaces                      */
/* OverlapPatchEmbed.forward (shuffle_mode 0), model/Restromer.py:181-183;      */
/* Downsample.forward = Conv2d(C, C/2, 3) + PixelUnshuffle(2) (mode 1),         */
/* :188-200; Upsample.forward = Conv2d(C, 2C, 3) + PixelShuffle(2) (mode 2),    */
/* :203-215.  x: [B, Cin, H, W]; weight: [Cout, Cin, 3, 3]; bias [Cout] or NULL. */
/* y: mode 0 [B, Cout, H, W]; mode 1 [B, 4 Cout, H/2, W/2]; mode 2              */
/* [B, Cout/4, 2H, 2W].  The product runs on the f16 matrix instructions with   */
/* both operands split in two (fp32 accuracy; |x| < 65504).  Shapes: any W      */
/* (rows of 128-pixel tiles, the last one masked; 32 / 64-pixel images in tiles */
/* of 4 / 2 rows; a width that is not a multiple of 8 runs on a copy with zero   */
/* columns added in the workspace), mode 1: even H and W, mode 2: Cout % 4 == 0  */
/* (rst_conv3x3_supported).  The workspace holds the split weights             */
/* (rst_conv3x3_workspace_bytes: enough whenever W % 8 == 0 and the exact-fp32   */
/* switch is off) and the padded copies; rst_conv3x3_fwd_workspace_bytes is      */
/* enough for the shape given under the current state of rst_set_exact_fp32      */
/* (0 for an unsupported shape).  With the switch on the product runs on the     */
/* exact-fp32 instruction over the nine shifted copies of the input, written to  */
/* the workspace per chunk of samples (at most 1 GiB at a time).                 */
/* ------------------------------------------------------------------------- */
size_t rst_conv3x3_workspace_bytes(int32_t Cin, int32_t Cout);
size_t rst_conv3x3_fwd_workspace_bytes(int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode);
int32_t rst_conv3x3_supported(int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode);
int rst_conv3x3_fwd(const float* x, const float* weight, const float* bias, float* y,
                    int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode,
                    void* workspace, size_t workspace_bytes, void* stream);
/* Its backward (what autograd derives from model/Restromer.py:172-215, 381-383):   */
/* from the forward's input x [B][Cin][H][W] and the gradient dy of its (shuffled)   */
/* output: dx (NULL: not wanted), dweight [Cout][Cin][3][3], dbias [Cout] (NULL:     */
/* none), overwritten.  Exact-fp32 products (gradients are far below f16's range).   */
/* Any Cin, Cout (also the last layer's 48 -> 3), any W (not a multiple of 4: on     */
/* zero-padded copies in the workspace); shuffle_mode as in the forward.  The        */
/* workspace query returns 0 for unsupported shapes.                                 */
size_t rst_conv3x3_bwd_workspace_bytes(int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W,
                                       int32_t shuffle_mode, int32_t want_dx, int32_t want_bias);
int rst_conv3x3_bwd(const float* x, const float* dy, const float* weight, float* dx, float* dweight, float* dbias,
                    int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Depthwise 3x3, zero padding 1 -- replaces qkv_dwconv / dwconv,             */
/* model/Restromer.py:87-95,116-124.  x, y: [B, C, H, W]; weight: [C, 9].     */
/* ------------------------------------------------------------------------- */
int rst_dwconv3x3_fwd(const float* x, const float* weight, const float* bias, float* y,
                      int32_t B, int32_t C, int32_t H, int32_t W, void* stream);
size_t rst_dwconv3x3_bwd_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W);
int rst_dwconv3x3_bwd(const float* x, const float* weight, const float* dy,
                      float* dx, float* dweight, float* dbias,
                      int32_t B, int32_t C, int32_t H, int32_t W,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* PixelUnshuffle(2) / PixelShuffle(2) -- model/Restromer.py:196,211.         */
/* unshuffle: [B, C, H, W] -> [B, 4C, H/2, W/2]; shuffle: [B, 4C, H, W] ->    */
/* [B, C, 2H, 2W].  Each is the other's backward.                             */
/* ------------------------------------------------------------------------- */
int rst_pixel_unshuffle2(const float* x, float* y, int32_t B, int32_t C, int32_t H, int32_t W, void* stream);
int rst_pixel_shuffle2(const float* x, float* y, int32_t B, int32_t C, int32_t H, int32_t W, void* stream);

/* ------------------------------------------------------------------------- */
/* MDTA -- replaces Attention.forward, model/Restromer.py:127-150.            */
/* y = [residual +] project_out( softmax(q^ k^T * temperature) v )            */
/* with q,k,v = split(dwconv3x3(conv1x1(x))).  Uses desc.{B,C,H,W,heads,      */
/* conv_bias} and params.{temperature,qkv_*,proj_*}.                          */
/* saved: NULL for inference; else a buffer of rst_attention_saved_bytes that */
/* receives what the backward needs.                                          */
/* ------------------------------------------------------------------------- */
size_t rst_attention_workspace_bytes(const rst_block_desc* desc);
size_t rst_attention_saved_bytes(const rst_block_desc* desc);
int rst_attention_fwd(const rst_block_desc* desc, const rst_block_params* params,
                      const float* x, const float* residual, float* y,
                      void* saved, void* workspace, size_t workspace_bytes, void* stream);
/* Backward: given dy, writes dx (gradient w.r.t. x only; the residual branch
 * is the caller's) and the gradients of temperature, qkv_*, proj_*. */
size_t rst_attention_bwd_workspace_bytes(const rst_block_desc* desc);
int rst_attention_bwd(const rst_block_desc* desc, const rst_block_params* params,
                      const float* x, const float* dy, const void* saved,
                      float* dx, int32_t accumulate_dx, rst_block_grads* grads,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* GDFN -- replaces FeedForward.forward, model/Restromer.py:99-104.           */
/* y = [residual +] project_out( gelu(d1) * d2 ),  d = dwconv3x3(project_in(x)). */
/* ------------------------------------------------------------------------- */
size_t rst_ffn_workspace_bytes(const rst_block_desc* desc);
size_t rst_ffn_saved_bytes(const rst_block_desc* desc);
int rst_ffn_fwd(const rst_block_desc* desc, const rst_block_params* params,
                const float* x, const float* residual, float* y,
                void* saved, void* workspace, size_t workspace_bytes, void* stream);
size_t rst_ffn_bwd_workspace_bytes(const rst_block_desc* desc);
int rst_ffn_bwd(const rst_block_desc* desc, const rst_block_params* params,
                const float* x, const float* dy, const void* saved,
                float* dx, int32_t accumulate_dx, rst_block_grads* grads,
                void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* TransformerBlock -- replaces TransformerBlock.forward,                     */
/* model/Restromer.py:163-167:  x += attn(norm1(x)); x += ffn(norm2(x)).      */
/* y may alias x only when saved == NULL.                                     */
/* ------------------------------------------------------------------------- */
size_t rst_block_workspace_bytes(const rst_block_desc* desc);
size_t rst_block_saved_bytes(const rst_block_desc* desc);
int rst_block_fwd(const rst_block_desc* desc, const rst_block_params* params,
                  const float* x, float* y,
                  void* saved, void* workspace, size_t workspace_bytes, void* stream);
/* Inference with the input-independent part done ahead: rst_block_prepare writes  */
/* the LayerNorm-folded weights of qkv and project_in and the operand fragments of */
/* project_out (model/Restromer.py:26-52 folded into :130,:100; :103) into a       */
/* caller-owned device buffer of rst_block_prepared_bytes (depends on C, hidden and */
/* the flags only); it stays valid until a parameter changes.                      */
/* rst_block_fwd_prepared == rst_block_fwd with saved == NULL, minus those launches. */
size_t rst_block_prepared_bytes(const rst_block_desc* desc);
int rst_block_prepare(const rst_block_desc* desc, const rst_block_params* params, void* prepared, void* stream);
int rst_block_fwd_prepared(const rst_block_desc* desc, const rst_block_params* params, const void* prepared,
                           const float* x, float* y, void* workspace, size_t workspace_bytes, void* stream);
size_t rst_block_bwd_workspace_bytes(const rst_block_desc* desc);
int rst_block_bwd(const rst_block_desc* desc, const rst_block_params* params,
                  const float* x, const float* dy, const void* saved,
                  float* dx, rst_block_grads* grads,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* One TransformerBlock of an image whose ROWS are split over several devices */
/* (inference only) -- the same TransformerBlock.forward,                     */
/* model/Restromer.py:163-167, whose attention (model/Restromer.py:128-150)   */
/* sums q.k^T and the norms of q and k over ALL pixels: the sums are the only */
/* thing the devices have to exchange inside a block.                         */
/*   desc->H = rows of the local slab INCLUDING halo rows (two on every side  */
/*   that has a neighbour: the block contains two depthwise 3x3 convolutions); */
/*   [row0, row1) = the rows this device owns.                                 */
/*   1. rst_block_shard_fwd_stats: qkv of the slab (kept in the workspace) and */
/*      this device's share of the sums over its own rows ->                   */
/*      stats[B*heads*c*c | B*C | B*C]  (Gram, |q|^2, |k|^2; c = C / heads).   */
/*   2. the caller adds `stats` over the devices (one all-reduce);             */
/*   3. rst_block_shard_fwd_apply with the summed stats and the SAME, untouched */
/*      workspace: y = block(x) on the slab; rows closer than two rows to a    */
/*      cut are not valid and are to be refreshed from the neighbour.          */
/* y may alias x.  With one device (row0 = 0, row1 = H) the pair equals        */
/* rst_block_fwd.                                                              */
/* ------------------------------------------------------------------------- */
size_t rst_block_shard_workspace_bytes(const rst_block_desc* desc);
size_t rst_block_shard_stats_floats(const rst_block_desc* desc);
int rst_block_shard_fwd_stats(const rst_block_desc* desc, const rst_block_params* params,
                              const float* x, int32_t row0, int32_t row1, float* stats,
                              void* workspace, size_t workspace_bytes, void* stream);
int rst_block_shard_fwd_apply(const rst_block_desc* desc, const rst_block_params* params,
                              const float* x, const float* stats, float* y,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* bf16 storage path (inference).  The reference runs this path in whatever    */
/* dtype the module was cast to (model/train.py:72 casts to fp32; autocast or   */
/* .bfloat16() give bf16 tensors between its torch operators).  Here the        */
/* activations between kernels are bf16 and every sum is fp32; parameters stay  */
/* fp32 device arrays as in rst_block_params.                                   */
/* Activation layout: "channel-packed" bf16, [B][ceil(C/8)][N][8] -- the 8       */
/* channels of a block for one pixel form one 16-byte chunk (channels beyond C   */
/* are zero).  rst_bf16_pack / rst_bf16_unpack convert from / to planar NCHW     */
/* (fp32 or bf16).  N = H*W must be even; C a multiple of 8 for the block.       */
/* ------------------------------------------------------------------------- */
size_t rst_bf16_packed_bytes(int32_t B, int32_t C, int64_t N);
int rst_bf16_pack(const void* x, int32_t x_is_fp32, void* y_packed,
                  int32_t B, int32_t C, int64_t N, void* stream);
int rst_bf16_unpack(const void* x_packed, void* y, int32_t y_is_fp32,
                    int32_t B, int32_t C, int64_t N, void* stream);
/* Packed <-> channels-last bf16 [B][H][W][C] (the layout the dense 3x3           */
/* convolutions of Downsample / Upsample / OverlapPatchEmbed, model/Restromer.py  */
/* :172-215, are handed to MIOpen in).  rst_bf16_from_channels_last applies, on    */
/* the way, shuffle_mode 0: nothing; 1: nn.PixelUnshuffle(2), model/Restromer.py   */
/* :196 (x is B x H x W x C, y packed 4C channels at H/2 x W/2); 2:                */
/* nn.PixelShuffle(2), :211 (y packed C/4 channels at 2H x 2W).  C % 8 == 0        */
/* (mode 1: C even and the result's 4C a multiple of 8; mode 2: C % 32 == 0).      */
int rst_bf16_to_channels_last(const void* x_packed, void* y_nhwc,
                              int32_t B, int32_t C, int64_t N, void* stream);
int rst_bf16_from_channels_last(const void* x_nhwc, void* y_packed, int32_t shuffle_mode,
                                int32_t B, int32_t C, int32_t H, int32_t W, void* stream);
/* nn.Conv2d(Cin, Cout, kernel_size=1) on packed tensors (+ packed residual),     */
/* model/Restromer.py:309-311,326-328.  Cin <= 480.  y may alias residual.        */
size_t rst_bf16_conv1x1_workspace_bytes(int32_t Cin, int32_t Cout);
int rst_bf16_conv1x1_fwd(const void* x_packed, const float* weight, const float* bias,
                         const void* residual_packed, void* y_packed,
                         int32_t B, int32_t Cin, int32_t Cout, int64_t N,
                         void* workspace, size_t workspace_bytes, void* stream);
/* OverlapPatchEmbed.proj, Downsample, Upsample (model/Restromer.py:172-215) on   */
/* packed tensors: nn.Conv2d(Cin, Cout, 3, padding=1) followed by nothing          */
/* (shuffle_mode 0; Cout % 8 == 0), nn.PixelUnshuffle(2) (1: y packed 4 Cout       */
/* channels at H/2 x W/2; H, W, Cout even) or nn.PixelShuffle(2) (2: y packed       */
/* Cout / 4 channels at 2H x 2W; Cout % 32 == 0).  x: packed Cin channels (the      */
/* padding channels of its last block must be zero, as rst_bf16_pack writes them).  */
/* weight / bias fp32.                                                              */
size_t rst_bf16_conv3x3_workspace_bytes(int32_t Cin, int32_t Cout);
int32_t rst_bf16_conv3x3_supported(int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode);
int rst_bf16_conv3x3_fwd(const void* x_packed, const float* weight, const float* bias, void* y_packed,
                         int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode,
                         void* workspace, size_t workspace_bytes, void* stream);
/* The network's last layer on the bf16 storage path (model/Restromer.py:381-383,  */
/* 420-431): nn.Conv2d(Cin, Cout, 3, padding=1) with at most 4 output channels on   */
/* a packed bf16 input, fp32 weights and sums, fp32 NCHW result; plus `residual`     */
/* ([B][Cout][H][W] fp32) or the bilinear x2 (align_corners=False) of `lowres`       */
/* ([B][Cout][H/2][W/2] fp32), at most one of the two, either may be null.  W even.  */
int32_t rst_bf16_conv3x3_fewout_fwd(const void* x_packed, const float* weight, const float* bias, const float* residual,
                                    const float* lowres, float* y, int32_t B, int32_t Cin, int32_t Cout, int32_t H,
                                    int32_t W, void* stream);
/* TransformerBlock.forward, model/Restromer.py:163-167, on a packed tensor;      */
/* y may alias x.  C % 8 == 0, C / heads a multiple of 16 (<= 192), C and hidden  */
/* at most 480, W even.  Returns 0 bytes / RST_ERR_INVALID for unsupported shapes. */
/* The convolution weights enter the kernels in a prepared form (LayerNorm scale   */
/* folded in, bf16, matrix-operand order; model/Restromer.py:110-125,80-97 hold    */
/* the originals).  rst_bf16_block_prepare writes it into a caller-owned device     */
/* buffer of rst_bf16_block_prepared_bytes (depends on C, heads, hidden and the     */
/* flags only, not on B, H, W); it stays valid until a parameter changes.  With     */
/* prepared == NULL rst_bf16_block_fwd prepares into its workspace on every call.   */
size_t rst_bf16_block_prepared_bytes(const rst_block_desc* desc);
int rst_bf16_block_prepare(const rst_block_desc* desc, const rst_block_params* params,
                           void* prepared, void* stream);
size_t rst_bf16_block_workspace_bytes(const rst_block_desc* desc);
int rst_bf16_block_fwd(const rst_block_desc* desc, const rst_block_params* params,
                       const void* prepared, const void* x_packed, void* y_packed,
                       void* workspace, size_t workspace_bytes, void* stream);
/* Its backward (what autograd derives from model/Restromer.py:163-167 when the   */
/* block runs on bf16 tensors): from the block's packed INPUT and the packed       */
/* gradient of its output, the packed gradient of the input and the fp32 gradients  */
/* of all parameters (overwritten).  The block is evaluated again inside, so        */
/* nothing but x has to be kept from the forward pass.  dx must not alias x or dy.  */
/* prepared: as for rst_bf16_block_fwd (the forward pass's buffer if the parameters */
/* have not changed since), or NULL.  Same shape limits as rst_bf16_block_fwd.      */
size_t rst_bf16_block_bwd_workspace_bytes(const rst_block_desc* desc);
int rst_bf16_block_bwd(const rst_block_desc* desc, const rst_block_params* params, const void* prepared,
                       const void* x_packed, const void* dy_packed, void* dx_packed,
                       rst_block_grads* grads, void* workspace, size_t workspace_bytes, void* stream);
/* What autograd derives from the glue between the stages when the network trains  */
/* on bf16 tensors (model/Restromer.py:172-215, 381-383: OverlapPatchEmbed,        */
/* Downsample, Upsample, the output convolution; :309-311, 326-328: the 1x1         */
/* reductions; :402, 407, 412: torch.cat; nn.PixelShuffle / nn.PixelUnshuffle):     */
/* rst_bf16_conv3x3_bwd: from the packed input x of rst_bf16_conv3x3_fwd and the    */
/* packed gradient dy of its (shuffled) output: dx (packed; NULL: not wanted -- the  */
/* first layer; otherwise Cin % 8 == 0), dweight [Cout][Cin][3][3] and dbias [Cout]  */
/* (NULL: none) in fp32, overwritten.  shuffle_mode 0 also takes Cout < 8 (the last  */
/* layer, whose forward is rst_bf16_conv3x3_fewout_fwd; dy then holds Cout channels  */
/* in one block).  workspace_bytes query returns 0 for unsupported shapes.           */
size_t rst_bf16_conv3x3_bwd_workspace_bytes(int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W,
                                            int32_t shuffle_mode, int32_t want_dx);
int rst_bf16_conv3x3_bwd(const void* x_packed, const void* dy_packed, const float* weight, void* dx_packed,
                         float* dweight, float* dbias, int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W,
                         int32_t shuffle_mode, void* workspace, size_t workspace_bytes, void* stream);
/* rst_bf16_conv1x1_bwd: gradients of rst_bf16_conv1x1_fwd (without residual): dx    */
/* packed, dweight [Cout][Cin], dbias [Cout] or NULL.                                */
size_t rst_bf16_conv1x1_bwd_workspace_bytes(int32_t B, int32_t Cin, int32_t Cout, int64_t N);
int rst_bf16_conv1x1_bwd(const void* x_packed, const void* dy_packed, const float* weight, void* dx_packed,
                         float* dweight, float* dbias, int32_t B, int32_t Cin, int32_t Cout, int64_t N,
                         void* workspace, size_t workspace_bytes, void* stream);
/* nn.PixelShuffle(2): [B][4 C_out][H_in][W_in] -> [B][C_out][2 H_in][2 W_in], and    */
/* nn.PixelUnshuffle(2): [B][C_in][H_in][W_in] -> [B][4 C_in][H_in/2][W_in/2], packed   */
/* in and out (each is the other's gradient); C_out / C_in multiples of 8.           */
int rst_bf16_pixel_shuffle2(const void* x_packed, void* y_packed, int32_t B, int32_t C_out, int32_t H_in, int32_t W_in,
                            void* stream);
int rst_bf16_pixel_unshuffle2(const void* x_packed, void* y_packed, int32_t B, int32_t C_in, int32_t H_in, int32_t W_in,
                              void* stream);
/* torch.cat along the channels and its gradient, one half at a time: channels        */
/* [src_c0, src_c0 + nC) of src (src_C channels) into [dst_c0, dst_c0 + nC) of dst    */
/* (dst_C channels); everything in whole blocks of 8.                                 */
int rst_bf16_copy_channel_blocks(const void* src_packed, int32_t src_C, int32_t src_c0, void* dst_packed, int32_t dst_C,
                                 int32_t dst_c0, int32_t nC, int32_t B, int64_t N, void* stream);
/* The same pair for a training step that can spare a little memory per block     */
/* (what autograd keeps of model/Restromer.py:141-146 -- attn before and after      */
/* the softmax -- plus the norms of q and k and the product with project_out):      */
/* rst_bf16_block_fwd_keep also writes these small fp32 matrices into `kept`, a     */
/* caller-owned device buffer of rst_bf16_block_kept_bytes (16-byte aligned;         */
/* B*C*(C + 2*C/heads + 2) floats), and rst_bf16_block_bwd_kept reads them instead   */
/* of forming them again from q and k: the gradients are those of the attention the */
/* forward pass applied.  Everything else as for rst_bf16_block_fwd / _bwd.          */
size_t rst_bf16_block_kept_bytes(const rst_block_desc* desc);
int rst_bf16_block_fwd_keep(const rst_block_desc* desc, const rst_block_params* params,
                            const void* prepared, const void* x_packed, void* y_packed, void* kept,
                            void* workspace, size_t workspace_bytes, void* stream);
int rst_bf16_block_bwd_kept(const rst_block_desc* desc, const rst_block_params* params, const void* prepared,
                            const void* x_packed, const void* dy_packed, void* dx_packed,
                            rst_block_grads* grads, const void* kept, void* workspace, size_t workspace_bytes,
                            void* stream);

/* The same pair for a training step that has memory to spend (what autograd keeps   */
/* of model/Restromer.py:99-104,128-150,163-167: every intermediate of the block):   */
/* rst_bf16_block_fwd_save evaluates the block with the unfused kernels and leaves,  */
/* besides the kept attention above, the two 1x1 outputs, their depthwise results,   */
/* x + attn(...) and the gated product in `saved` -- a caller-owned device buffer of */
/* rst_bf16_block_saved_bytes (16-byte aligned; the kept attention plus              */
/* 2*(7*C + 5*pad8(hidden)) bytes per pixel).  y_packed must not alias x_packed.     */
/* rst_bf16_block_bwd_saved reads them and evaluates nothing again; its workspace    */
/* (rst_bf16_block_bwd_saved_workspace_bytes) is smaller by those tensors.           */
/* Gradients as for rst_bf16_block_bwd, of the forward pass that was applied.        */
size_t rst_bf16_block_saved_bytes(const rst_block_desc* desc);
size_t rst_bf16_block_bwd_saved_workspace_bytes(const rst_block_desc* desc);
int rst_bf16_block_fwd_save(const rst_block_desc* desc, const rst_block_params* params,
                            const void* prepared, const void* x_packed, void* y_packed, void* saved,
                            void* workspace, size_t workspace_bytes, void* stream);
int rst_bf16_block_bwd_saved(const rst_block_desc* desc, const rst_block_params* params, const void* prepared,
                             const void* x_packed, const void* dy_packed, void* dx_packed,
                             rst_block_grads* grads, const void* saved, void* workspace, size_t workspace_bytes,
                             void* stream);

/* ------------------------------------------------------------------------- */
/* Dataset item on the GPU -- replaces LazyRAWDataset.__getitem__,             */
/* data/loader.py:33-52 (raw / max_val, odd trailing row / column dropped) and  */
/* downsample_raw, utils/util.py:313-321 (2 x 2 average pool), for a batch of   */
/* sensor samples already in device memory.  raw: [B, H, W, C] interleaved,     */
/* C <= 4, uint16 (raw_is_u16 != 0) or float32; max_val: [B] white levels       */
/* (device).  hr: [B, C, H2, W2] with H2 = H & ~1, W2 = W & ~1; lr: [B, C,       */
/* H2/2, W2/2].  Division and pooling run in float64 and are rounded to float32  */
/* once, as the reference's numpy / torch calls do.                              */
/* ------------------------------------------------------------------------- */
int rst_raw_prepare(const void* raw, int32_t raw_is_u16, const float* max_val, float* hr, float* lr,
                    int32_t B, int32_t H, int32_t W, int32_t C, void* stream);

/* ------------------------------------------------------------------------- */
/* Per-kernel timing aid.  No counterpart in the reference (it has no timing   */
/* code on this path); used by bench.py for the roofline numbers.  Between     */
/* begin and end every kernel launch of this library is bracketed by HIP       */
/* events on its stream; end waits for them and writes a JSON object           */
/* {label: {launches, ms, flops, bytes}} (algorithmic FLOPs / bytes) into json. */
/* Process-wide, not for concurrent use; adds two event records per launch.    */
/* ------------------------------------------------------------------------- */
int rst_profile_begin(void);
int rst_profile_end(char* json, size_t capacity);

/* Peak micro-benchmarks (no counterpart in the reference; tools/peaks.py times them):
 * a streaming float4 copy (or, read_only != 0, a pure read) of `bytes` bytes, and bare matrix-instruction loops
 * (kind 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_f32_16x16x32_bf16, 2: v_mfma_f32_16x16x32_f16) of
 * blocks x 4 waves x iters x 8 instructions on register operands; out holds blocks * 256 floats. */
int rst_peak_copy(const void* src, void* dst, size_t bytes, int32_t read_only, void* stream);
int rst_peak_mfma(int32_t kind, int32_t blocks, int32_t iters, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RESTORMER_HIP_H */
