// Internal launch interface between the kernel files and the C ABI (api.hip).
#pragma once
#include "common.hpp"

namespace rst {

// ---- layernorm.hip ----
int layernorm_fwd_launch(const float* x, const float* weight, const float* bias, float* y,
                         int B, int C, int64_t N, hipStream_t stream);
size_t layernorm_bwd_ws_bytes(int B, int C, int64_t N);
int layernorm_bwd_launch(const float* x, const float* weight, int with_bias, const float* dy,
                         const float* addend, float* dx, float* dweight, float* dbias,
                         int B, int C, int64_t N, void* ws, size_t ws_bytes, hipStream_t stream);
// dx = addend + LN-backward(dy); addend may be null or alias dx.
// the same with the partial sums in the library's own per-device scratch array (no workspace; C <= 384)
int layernorm_bwd_noalloc_launch(const float* x, const float* weight, int with_bias, const float* dy,
                                 const float* addend, float* dx, float* dweight, float* dbias,
                                 int B, int C, int64_t N, hipStream_t stream);

// ---- gemm_nn.hip ----
struct GemmNN {
    const float* A; int64_t a_s1, a_s2; int lda; int transA;
    const float* X; int64_t x_s1, x_s2; int64_t ldx;
    float* Y;       int64_t y_s1, y_s2; int64_t ldy;
    const float* R; int64_t r_s1, r_s2; int64_t ldr;
    const float* rscale; int64_t rs_s1, rs_s2;
    const float* bias;
    int M, K; int64_t N; int nb1, nb2;
    int accumulate;
    // Channel LayerNorm of X folded into the product (ln_mode != 0): A holds W o gamma (columns
    // scaled), ln_s[m] = sum_k A[m][k], ln_c[m] = sum_k W[m][k] beta[k] (+ conv bias), and the
    // kernel takes mean / variance over the K rows of every X column while it multiplies:
    //   mode 1 (with mean):  Y = rstd * (A X - mean * ln_s) + ln_c
    //   mode 2 (bias-free):  Y = rstd * (A X) + ln_c          (variance still about the mean)
    int ln_mode;
    const float* ln_s;
    const float* ln_c;
    // A as split-f16 operand fragments (gemm_f16s_prep_launch): when set, the product runs on the f16 matrix
    // instructions (gemm_f16s.hip) and A / lda / transA are not read; strides between batches in bytes
    const void* afrag; int64_t afrag_s1, afrag_s2;
    // (fragment path only) the input channels come from TWO tensors: rows [0, ksplit) from X, rows [ksplit, K) from X2
    // (same ldx; ksplit a multiple of 32): a channel concatenation that is never materialised
    const float* X2; int64_t x2_s1, x2_s2; int ksplit;
};
int gemm_nn_launch(const GemmNN& g, hipStream_t stream);
// gemm_f16s.hip
size_t gemm_f16s_frag_bytes(int M, int K);
bool gemm_f16s_enabled();                       // false under RST_F32_MFMA=1 or after rst_set_exact_fp32(1)
void set_exact_fp32(int on);
bool gemm_f16s_supported(const GemmNN& g);      // strides / alignment of X, Y, R
// w: nb matrices [M][K] (row stride lda; transA: stored [K][M]) w_bs elements apart -> fragments frag_bs_bytes apart;
// gamma (optional) scales column k; with s, c: s[m] = sum_k of the split folded row, c[m] = sum_k w[m][k] beta[k] + bias[m]
int gemm_f16s_prep_launch(const float* w, int64_t w_bs, int nb, int lda, int transA, const float* gamma, const float* beta,
                          const float* bias, int M, int K, void* frag, int64_t frag_bs_bytes, float* s, float* c,
                          hipStream_t stream);
int gemm_f16s_launch(const GemmNN& g, hipStream_t stream);
// true when a product with ln_mode set can be served (the fused kernels need 16-byte aligned
// operands); otherwise the caller normalises first and multiplies plainly
bool gemm_nn_ln_supported(const GemmNN& g);
// folded[m][k] = w[m][k] * gamma[k];  s[m] = sum_k folded[m][k];  c[m] = sum_k w[m][k] * beta[k] + bias[m]
// (beta, bias may be null)
int ln_fold_weights_launch(const float* w, const float* gamma, const float* beta, const float* bias, int M, int K,
                           float* folded, float* s, float* c, hipStream_t stream);

// ---- pwdw.hip: 1x1 convolution (+ folded LayerNorm) fused with the 3x3 depthwise convolution ----
struct PwDw {
    const float* X; int64_t x_bs;        // [B][K][H][W]
    const float* A;                      // [M][K]; W o gamma when ln_mode != 0
    int ln_mode;                         // as GemmNN::ln_mode; 0: plain product, ln_c (may be null) is the conv bias
    const float* ln_s; const float* ln_c;
    const float* dw_w;                   // [M][9]
    const float* dw_b;                   // [M] or null
    float* Y; int64_t y_bs;              // [B][M][H][W], or [B][hidden][H][W] with gate
    int B, K, M, H, W;
    int gate, hidden;                    // gate: M == 2*hidden, Y = gelu(d[:hidden]) * d[hidden:]
};
bool pwdw_supported(const PwDw& a);
bool pwdw_profitable(const PwDw& a);      // supported and measured faster than the unfused pair
int pwdw_launch(const PwDw& a, hipStream_t stream);
// pwdw_ws.hip: the same operation with the MFMAs and the depthwise work in different waves (96 input channels, no gate)
bool pwdw_ws_supported(const PwDw& a);
int pwdw_ws_launch(const PwDw& a, hipStream_t stream);

// pwdw_f16.hip: the same operation with the product on the f16 matrix instructions (operands split in two, fp32 accuracy)
bool pwdw_f16_supported(const PwDw& a);
int pwdw_f16_launch(const PwDw& a, hipStream_t stream);

// ---- gemm_nt.hip ----
struct GemmNT {
    const float* A; int64_t a_s1, a_s2; int64_t lda;
    const float* Bm; int64_t b_s1, b_s2; int64_t ldb;
    int M1, M2; int64_t N;
    int nb1, nb2;
    int splits; int64_t chunk;
    float* partial;
    float* sqA_partial;
    float* sqB_partial;
    int walk = 0;                 // set by the launcher: tile blocks of one pixel split placed on one XCD
};
struct NtDim { int T, W, blocks; };
struct NtPlan { NtDim d1, d2; int splits; int64_t chunk; };
// max_splits_cap > 0 bounds the number of pixel splits (the consumer adds the slabs serially)
NtPlan gemm_nt_plan(int M1, int M2, int64_t N, int nbatch, int max_splits_cap = 0);
size_t gemm_nt_partial_floats(const NtPlan& p, int M1, int M2, int nbatch);
int gemm_nt_launch(GemmNT g, const NtPlan& p, hipStream_t stream);
// gram_f16s.hip: the Gram use of GemmNT (M1 == M2, both squared-norm outputs) on split-f16 operands; same slabs
bool gram_f16s_supported(const GemmNT& g);
int gram_f16s_launch(const GemmNT& g, hipStream_t stream);
int reduce_slabs_launch(const float* partial, float* out, int64_t elems, int groups, int slabs_per_group,
                        int accumulate, hipStream_t stream);
// three tensors with the same (groups, slabs per group) in one launch (out = sum, no accumulation)
int reduce_slabs3_launch(const float* p0, float* o0, int64_t e0, const float* p1, float* o1, int64_t e1, const float* p2, float* o2,
                         int64_t e2, int groups, int slabs_per_group, hipStream_t stream);

// ---- dwconv.hip ----
struct DwArgs {
    const float* x; int64_t x_bstride;
    const float* w;
    const float* bias;
    float* y; int64_t y_bstride;
    int B, C, H, W;
    int flip;
};
struct GateArgs {
    const float* u; int64_t u_bstride;
    const float* w; const float* bias;
    float* gout; int64_t g_bstride;
    float* d; int64_t d_bstride;
    int B, hidden, H, W;
};
int dwconv_fwd_launch(const DwArgs& a, hipStream_t stream);
int dwconv_gate_fwd_launch(const GateArgs& a, hipStream_t stream);
int gate_bwd_launch(const float* dg, const float* d, float* dd, int B, int hidden, int64_t hw, hipStream_t stream);
size_t dwconv_bwd_weight_ws_bytes(int B, int C, int H, int W);
int dwconv_bwd_weight_launch(const float* x, int64_t x_bstride, const float* dy, int64_t dy_bstride,
                             float* dweight, float* dbias, int B, int C, int H, int W,
                             void* ws, size_t ws_bytes, hipStream_t stream);

// dense 3x3 convolution (zero padding 1) with 1..4 output channels, optional residual added to the result
struct Conv3Args {
    const float* x;          // [B][Cin][H][W]
    const float* w;          // [Cout][Cin][3][3]
    const float* bias;       // [Cout] or null
    const float* res;        // [B][Cout][H][W] or null
    const float* up_src;     // [B][Cout][H/2][W/2] or null: its bilinear x2 (align_corners = False) is added to the result
    float* y;                // [B][Cout][H][W]
    int B, Cin, Cout, H, W;
};
int conv3x3_fewout_launch(const Conv3Args& a, hipStream_t stream);

// ---- shuffle.hip ----
int pixel_shuffle2_launch(bool shuffle, const float* src, float* dst, int B, int C, int H, int W, hipStream_t stream);
// conv3x3_f16s.hip: dst[r][c] = c < ws ? src[r][c] : 0 over rows of ws -> wd values (zero columns added, or a crop)
int copy_cols_launch(const float* src, float* dst, int64_t rows, int ws, int wd, hipStream_t stream);
// conv3x3_bwd.hip: the dense 3x3 forward on the exact-fp32 matrix instruction (W a multiple of 4)
size_t conv3x3_exact_ws_bytes(int B, int Cin, int Cout, int H, int W, int mode);
int conv3x3_exact_launch(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int mode,
                         void* ws, size_t ws_bytes, hipStream_t stream);

// ---- attn_small.hip ----
int attn_softmax_launch(const float* gram_partial, const float* sqq_partial, const float* sqk_partial,
                        const float* temperature, float* attn, float* ghat, float* nq, float* nk,
                        int B, int heads, int c, int splits, hipStream_t stream);
int attn_fold_launch(const float* wo, const float* attn, float* mfold, int B, int C, int heads, int c, hipStream_t stream);
int attn_fold_f16s_launch(const float* wo, const float* attn, float* mfold, void* frag, int64_t frag_bytes, int B, int C, int heads,
                          int c, hipStream_t stream);
int attn_fold_frag_launch(const float* wo, const float* attn, float* mfold, uint16_t* frag, int64_t frag_bs, int ntiles, int KS,
                          int B, int C, int heads, int c, hipStream_t stream);

int attn_bwd_small_launch(const float* P, const float* wo, const float* attn, const float* ghat, const float* nq,
                          const float* nk, const float* temperature, float* Eq, float* Dq, float* Dk, float* dGG,
                          float* dtemp_row, float* dtemp_b, int B, int C, int heads, int c,
                          hipStream_t stream);
// dWo_b[b] = P[b] blockdiag(softmax)^T per sample (independent of attn_bwd_small_launch's chain)
int attn_bwd_wo_launch(const float* P, const float* attn, float* dwo_b, int B, int C, int heads, int c, hipStream_t stream);

// ---- reduce.hip ----
// out[row] = sum over batch and pixels of x[b][row][n]
size_t rowsum_ws_bytes(int B, int rows, int64_t N);
int rowsum_launch(const float* x, int64_t x_bstride, float* out, int B, int rows, int64_t N,
                  void* ws, size_t ws_bytes, hipStream_t stream);

// ---- bf16_pk.hip: bf16 storage path (inference).  Activations are channel-packed bf16, [B][Cb][N][8] with
// Cb = ceil(C/8): one 16-byte chunk holds 8 consecutive channels of one pixel; sums are fp32 ----
typedef uint16_t bf16_t;                 // raw bits of a bfloat16
// 1x1 convolution Y[b] = A[b] X[b] (+c) (+R[b]) or, with ln_mode != 0 (as GemmNN::ln_mode), the channel LayerNorm of
// X folded in.  A comes as MFMA operand fragments written by pk_prep_a_launch for gemm_pk_pairs(Kb, Mb) pairs per wave.
struct GemmPk {
    const bf16_t* A; int64_t a_bs;       // a_bs: 16-byte units between the samples' fragments (0: shared weights)
    const bf16_t* X; int64_t x_bs;       // strides in elements
    bf16_t* Y;       int64_t y_bs;
    const bf16_t* R; int64_t r_bs;       // residual in the layout of Y, or null (may alias Y)
    const float* ln_s; const float* ln_c; // per output position (padded to whole wave groups); ln_c alone: bias
    int ln_mode;
    int K;                               // real input channels (divisor of the LayerNorm statistics)
    int Kb, Mb;                          // channel blocks in / out
    int B; int64_t N;
    int pb, ngroups, nchunks; int64_t chunk_px;   // filled by the launcher (pb: pairs of row tiles per block)
};
int gemm_pk_pairs(int Kb, int Mb);
int gemm_pk_ks(int Kb);                  // k-steps of 32 channels the kernel for Kb runs (padded), -1: unsupported
int gemm_pk_launch(GemmPk g, hipStream_t stream);
// source channels come as segments of `seg` channels, each padded to `seg_pad` positions in the packed buffer
// (seg_pad < 0: two segments interleaved in groups of four channels)
int pk_prep_a_launch(const float* w, int64_t w_bs, int ldw, const float* gamma, int M, int K, int seg, int seg_pad,
                     int ntiles, int KS, bf16_t* frag, int64_t frag_bs, int nb, hipStream_t stream);
int pk_prep_sc_launch(const float* w, int ldw, const float* gamma, const float* beta, const float* bias, int M, int K,
                      int seg, int seg_pad, int npos, float* s, float* c, hipStream_t stream);
// Preparations gathered into one launch: a weight (fragments + its s / c vectors = two jobs) or a depthwise tap table.
struct PkPrepJob {
    int kind, blocks;                          // 0: fragments, 1: s / c vectors, 2: depthwise taps
    const float *w, *gamma, *beta, *bias;
    int ldw, M, K, seg, seg_pad, ntiles, KS, npos, Cb;
    bf16_t* frag; float *s, *c, *out;
};
struct PkPrepJobs { static constexpr int MAX = 8; PkPrepJob j[MAX]; int n = 0; };
void pk_prep_jobs_add_weight(PkPrepJobs& jobs, const float* src, int K, const float* gamma, const float* beta, const float* bias,
                             int M, int seg, int seg_pad, int ntiles, int KS, int npos, bf16_t* frag, float* s, float* c);
void pk_prep_jobs_add_dw(PkPrepJobs& jobs, const float* w, const float* bias, int C, int seg, int seg_pad, int Cb, float* out);
int pk_prep_jobs_launch(const PkPrepJobs& jobs, hipStream_t stream);
// flip: the taps rotated by 180 degrees (the data gradient of the convolution)
int pk_prep_dw_launch(const float* w, const float* bias, int C, int seg, int seg_pad, int Cb, float* out, hipStream_t stream,
                      int flip = 0);
// depthwise 3x3 over `Cb` channel blocks.  gate: every block holds 4 channels of x1 and the same 4 of x2 (segments
// interleaved by the weight preparation, seg_pad < 0); y[4 cb + i] = gelu(d[i]) * d[4 + i], y has Cb / 2 blocks
struct DwPk {
    const bf16_t* x; int64_t x_bs; const float* wp;   // wp: [Cb][80] from pk_prep_dw_launch
    bf16_t* y; int64_t y_bs;
    int B, Cb, H, W; int gate;
    int rows;                            // filled by the launcher
};
int dw_pk_launch(DwPk a, hipStream_t stream);
// fused 1x1 (folded LayerNorm, ln_mode 1 or 2) -> depthwise 3x3 (-> gate, layout as DwPk): Y = dw(A X) with the
// intermediate kept in registers.  A, ln_s, ln_c as for GemmPk (shared weights), dww as DwPk::wp.
struct PwDwPk {
    const bf16_t* A; const bf16_t* X; int64_t x_bs; bf16_t* Y; int64_t y_bs;
    const float* ln_s; const float* ln_c; int ln_mode;
    const float* dww;
    int K, Kb, Mb;                       // Mb: channel blocks of the intermediate
    int B, H, W; int gate;
    int pb, ngroups, patches_per_block, rows;  // filled by the launcher
};
// dense 3x3 with at most 4 output channels: packed bf16 in, fp32 NCHW out (+ residual or + bilinear x2 of up_src)
int fewout_pk_launch(const bf16_t* x, const float* w, const float* bias, const float* res, const float* up_src, float* y,
                     int B, int Cin, int Cout, int H, int W, hipStream_t stream);
bool pwdw_pk_supported(int Kb, int W, int H);
int pwdw_pk_launch(PwDwPk g, hipStream_t stream);
// Gram matrices q k^T and squared row norms over the pixels: `splits` slabs per (sample, head)
struct GramPk {
    const bf16_t* Q; const bf16_t* Kx; int64_t bs;    // first channel block of q / k; elements between samples
    int cb;                              // channel blocks per head
    int heads, B; int64_t N;
    int splits; int64_t chunk_px;        // from gram_pk_plan
    float* partial; float* sqq; float* sqk;           // [BH][splits][c][c], [BH][splits][c] x 2
    int nsub;                            // filled by the launcher
};
bool gram_pk_supported(int c);
void gram_pk_plan(int c, int64_t N, int BH, int* splits, int64_t* chunk);
int gram_pk_launch(GramPk g, hipStream_t stream);
int pk_pack_launch(const void* src, int src_is_fp32, bf16_t* dst, int B, int C, int64_t N, hipStream_t stream);
int pk_unpack_launch(const bf16_t* src, void* dst, int dst_is_fp32, int B, int C, int64_t N, hipStream_t stream);
// packed <-> channels-last bf16 [B][H][W][C]; mode 1 / 2: through PixelUnshuffle(2) / PixelShuffle(2)

// ---- bf16_bwd.hip: gradients on the packed layout ----
// fragments of A[pos][kpos] = W^(T)[channel(pos)][channel(kpos0 + kpos)], kpos < Krange: rows and contraction index
// both through the position -> channel map of a segmented packed buffer (seg / seg_pad as for pk_prep_a_launch)
int pk_prep_at_launch(const float* w, int64_t w_bs, int ldw, int transA, int M, int mseg, int mseg_pad, int Ksrc, int kseg,
                      int kseg_pad, int kpos0, int Krange, int ntiles, int KS, bf16_t* frag, int64_t frag_bs, int nb,
                      hipStream_t stream);
struct PrepAtJob {
    const float* w; int ldw, transA, M, mseg, mseg_pad, Ksrc, kseg, kseg_pad, kpos0, Krange, ntiles, KS;
    bf16_t* frag;
};
struct PrepAtJobs { int n; PrepAtJob job[8]; };
int pk_prep_at_jobs_launch(const PrepAtJobs& jobs, hipStream_t stream);
// partial[b][split] = [Ab*8][Bb*8] sums over the split's pixels of A[b][m][px] Bm[b][k][px], followed (with_rowsum) by
// the Ab*8 sums of A's rows.  T, nsa, nsb, splits, chunk_px come from wgrad_pk_plan (needs Ab, Bb, B, N).
struct WgradPk {
    const bf16_t* A; int64_t a_bs; int Ab;
    const bf16_t* Bm; int64_t b_bs; int Bb;
    int B; int64_t N;
    float* partial; int with_rowsum;
    int T, nsa, nsb, splits; int64_t chunk_px;
    // taps of a dense 3x3 (tap_wp > 0): Bb counts 9 x tap_cb VIRTUAL channel blocks; virtual block v is channel block
    // v % tap_cb of Bm read (v / tap_cb / 3 - 1) rows of tap_wp pixels and (v / tap_cb % 3 - 1) pixels further on.
    // Both tensors are zero-padded images of tap_wp pixels per row (the shifted reads stay inside Bm's margins).
    int tap_wp, tap_cb;
};
void wgrad_pk_plan(WgradPk* g);
size_t wgrad_pk_partial_floats(const WgradPk& g);
int wgrad_pk_launch(const WgradPk& g, hipStream_t stream);
// tap sums of a depthwise 3x3: red[Cb][80] = per channel block [9 taps][8] sums of dy[p] x[p + tap] and the 8 sums of dy
struct DwWgradPk {
    const bf16_t* dy; int64_t dy_bs; const bf16_t* x; int64_t x_bs;
    int B, Cb, H, W;
    float* partial;                      // dw_wgrad_pk_partial_floats
    int rows;                            // filled by the launcher
};
size_t dw_wgrad_pk_partial_floats(int B, int Cb, int H, int W);
int dw_wgrad_pk_launch(DwWgradPk a, float* red, hipStream_t stream);
int dw_grad_unpack_launch(const float* red, float* dweight, float* dbias, int C, int seg, int seg_pad, int Cb, hipStream_t stream);
// gelu(x1) * x2 on the interleaved layout (DwPk, gate) and its gradient; Gb = channel blocks of the product
int gate_fwd_pk_launch(const bf16_t* d, bf16_t* g, int B, int Gb, int64_t N, hipStream_t stream);
int gate_bwd_pk_launch(const bf16_t* dg, const bf16_t* d, bf16_t* dd, int B, int Gb, int64_t N, hipStream_t stream);
// normalisation part of the channel LayerNorm, and dx = addend + its input gradient for dn = d/d(gamma xhat + beta)
int ln_fwd_pk_launch(const bf16_t* x, bf16_t* y, int B, int Cb, int64_t N, int with_bias, hipStream_t stream);
int ln_bwd_pk_launch(const bf16_t* dn, const bf16_t* x, const float* gamma, const bf16_t* addend, bf16_t* dx, int B, int Cb,
                     int64_t N, int with_bias, hipStream_t stream);
// y = W (gamma xhat + beta) + bias: parameter gradients from G[pos][c] = sum dy[pos] xhat[c] and S[pos] = sum dy[pos]
int lnlin_grads_launch(const float* G, const float* S, const float* W, const float* gamma, const float* beta, float* dW,
                       float* dbias, float* dgamma, float* dbeta, int M, int C, int seg, int seg_pad, hipStream_t stream);
int crop2d_launch(const float* in, int ld_in, float* out, int rows, int cols, hipStream_t stream);
int attn_bwd_mats_launch(const float* Eq, const float* Dq, const float* Dk, float* out, int B, int C, int c, hipStream_t stream);

// ---- conv3x3_pk.hip: dense 3x3 convolution on packed tensors; mode 1 / 2: PixelUnshuffle(2) / PixelShuffle(2) epilogue ----
bool conv3x3_pk_supported(int Cin, int Cout, int H, int W, int mode);
size_t conv3x3_pk_ws_bytes(int Cin, int Cout);
// transposed: w is read as the weights of the convolution that gives the data gradient (rows and channels swapped, taps rotated)
int conv3x3_pk_launch(const bf16_t* x, const float* w, const float* bias, bf16_t* y, int B, int Cin, int Cout, int H, int W,
                      int mode, void* ws, size_t ws_bytes, hipStream_t stream, int transposed = 0);
// ---- conv3x3_bwd_pk.hip ----
bool conv3x3_pk_bwd_supported(int Cin, int Cout, int H, int W, int mode, bool want_dx);
size_t conv3x3_pk_bwd_ws_bytes(int B, int Cin, int Cout, int H, int W, int mode, int want_dx);
int conv3x3_pk_bwd_launch(const bf16_t* x, const bf16_t* dy, const float* w, bf16_t* dx, float* dweight, float* dbias, int B, int Cin,
                          int Cout, int H, int W, int mode, void* ws, size_t ws_bytes, hipStream_t stream);
int pk_shuffle2_launch(const bf16_t* src, bf16_t* dst, int B, int Cb_big, int h, int w, int to_big, hipStream_t stream);
int pk_copy_blocks_launch(const bf16_t* src, int src_cb_total, int src_cb0, bf16_t* dst, int dst_cb_total, int dst_cb0, int ncb,
                          int B, int64_t N, hipStream_t stream);

int pk_to_nhwc_launch(const bf16_t* src, bf16_t* dst, int B, int C, int64_t N, hipStream_t stream);
int pk_from_nhwc_launch(const bf16_t* src, bf16_t* dst, int mode, int B, int C, int H, int W, hipStream_t stream);

}  // namespace rst
