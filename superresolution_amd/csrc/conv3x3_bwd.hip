// Backward of the dense 3x3 convolutions (+ PixelUnshuffle / PixelShuffle) in fp32: what autograd derives from
// OverlapPatchEmbed, Downsample, Upsample and the output convolution (model/Restromer.py:172-215, 381-383).
//
// Gradients in this network are tiny (an L1 mean over millions of pixels: 1e-6 ... 1e-8), far into f16's subnormal
// range, so nothing here uses the split-f16 products of the forward pass: both gradients run on the exact-fp32 matrix
// instruction through the 1x1 kernels that already serve the blocks' backward.
//   col(t)[(tap, c)][y][x] = t[c][y + tap_y - 1][x + tap_x - 1] (zero outside): the nine shifted copies of a tensor,
//       written once by im2col3x3_kernel (HBM-bound: one read, nine writes);
//   data gradient   dx[ci] = sum over (tap, co) of W[co][ci][2 - tap_y][2 - tap_x] col(dy')[(tap, co)]: ONE 1x1
//       product (gemm_nn) with the [Cin][9 Cout] matrix conv3_wt_kernel lays out;
//   weight gradient dW[co][(tap, ci)] = sum over samples and pixels of dy'[co] col(x)[(tap, ci)]: ONE split-K product over
//       the pixels (gemm_nt) + its slab sum, then a permutation into [Cout][Cin][3][3];
//   bias gradient   the row sums of dy'.
// dy' is dy with the forward's shuffle undone (shuffle.hip).
#include "kernels.hpp"

namespace rst {

namespace {

struct CarveF {
    char* base; size_t off = 0;
    explicit CarveF(void* b) : base(static_cast<char*>(b)) {}
    void* bytes(size_t n) {
        off = (off + 255) & ~size_t(255);
        void* p = base ? static_cast<void*>(base + off) : nullptr;
        off += n;
        return p;
    }
    float* floats(size_t n) { return static_cast<float*>(bytes(n * sizeof(float))); }
    size_t total() const { return (off + 255) & ~size_t(255); }
};

// one thread: 4 consecutive pixels of one row of one plane -> the nine shifted copies (float4 stores)
__global__ void __launch_bounds__(256)
im2col3x3_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W, int64_t quads) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= quads) return;
    const int W4 = W >> 2;
    const int xq = (int)(q % W4), y = (int)((q / W4) % H);
    const int64_t bc = q / ((int64_t)W4 * H);
    const int c = (int)(bc % C);
    const int64_t b = bc / C;
    const int64_t HW = (int64_t)H * W;
    const float* pl = src + bc * HW;
    float in[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = y + r - 1;
        const bool ok = yy >= 0 && yy < H;
        const float* row = pl + (int64_t)(ok ? yy : y) * W + 4 * xq;
        const f32x4 mid = *reinterpret_cast<const f32x4*>(row);
        const float lf = xq > 0 ? row[-1] : 0.f, rt = xq < W4 - 1 ? row[4] : 0.f;
        in[r][0] = ok ? lf : 0.f; in[r][1] = ok ? mid.x : 0.f; in[r][2] = ok ? mid.y : 0.f;
        in[r][3] = ok ? mid.z : 0.f; in[r][4] = ok ? mid.w : 0.f; in[r][5] = ok ? rt : 0.f;
    }
    float* out = dst + (b * 9 * C + c) * HW + (int64_t)y * W + 4 * xq;
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
            *reinterpret_cast<f32x4*>(out + (int64_t)(ty * 3 + tx) * C * HW) = f32x4{in[ty][tx], in[ty][tx + 1], in[ty][tx + 2], in[ty][tx + 3]};
}

// wt[ci][(tap, co)] = w[co][ci][2 - tap_y][2 - tap_x]
__global__ void __launch_bounds__(256)
conv3_wt_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)Cin * 9 * Cout) return;
    const int co = (int)(i % Cout), tap = (int)((i / Cout) % 9), ci = (int)(i / (9 * (int64_t)Cout));
    wt[i] = w[((int64_t)co * Cin + ci) * 9 + (8 - tap)];
}

// dw[co][ci][tap] = red[co][(tap, ci)]
__global__ void __launch_bounds__(256)
conv3_dw_permute_kernel(const float* __restrict__ red, float* __restrict__ dw, int Cout, int Cin) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)Cout * Cin * 9) return;
    const int tap = (int)(i % 9), ci = (int)((i / 9) % Cin), co = (int)(i / (9 * (int64_t)Cin));
    dw[i] = red[((int64_t)co * 9 + tap) * Cin + ci];
}

struct Conv3BwdF {
    float *dyc, *col, *wt, *part, *red, *rsws;
    NtPlan plan;
    size_t rsws_bytes;
};

void conv3f_bwd_carve(int B, int Cin, int Cout, int H, int W, int mode, bool want_dx, bool want_bias, CarveF& cv, Conv3BwdF& b) {
    const int64_t N = (int64_t)H * W;
    const int cmax = (want_dx && Cout > Cin) ? Cout : Cin;
    b.dyc = mode != 0 ? cv.floats((size_t)B * Cout * N) : nullptr;
    b.col = cv.floats((size_t)B * 9 * cmax * N);                 // col(x) for the weights, then col(dy') for the data
    b.wt = want_dx ? cv.floats((size_t)Cin * 9 * Cout) : nullptr;
    b.plan = gemm_nt_plan(Cout, 9 * Cin, N, B);
    b.part = cv.floats(gemm_nt_partial_floats(b.plan, Cout, 9 * Cin, B));
    b.red = cv.floats((size_t)Cout * 9 * Cin);
    b.rsws_bytes = want_bias ? rowsum_ws_bytes(B, Cout, N) : 0;
    b.rsws = want_bias ? static_cast<float*>(cv.bytes(b.rsws_bytes)) : nullptr;
}

int im2col_launch(const float* src, float* dst, int B, int C, int H, int W, hipStream_t stream) {
    const int64_t quads = (int64_t)B * C * H * (W / 4);
    RST_REQUIRE(cdiv(quads, 256) < (1ll << 31), "conv3x3 backward: grid too large");
    ProfScope prof(stream, "im2col3x3", 0.0, 4.0 * 10.0 * B * C * H * W);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)cdiv(quads, 256)), dim3(256), 0, stream, src, dst, C, H, W, quads);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace

bool conv3x3_bwd_supported(int Cin, int Cout, int H, int W, int mode) {
    if (Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 2) return false;
    if (mode == 1 && (H % 2 != 0 || W % 2 != 0)) return false;
    if (mode == 2 && Cout % 4 != 0) return false;
    return true;
}

// The kernels take widths that are multiples of 4.  Any other width runs on copies of x and dy with zero columns added
// on the right: zero columns of dy add nothing to the weight and bias gradients, zero columns of x are the padding the
// convolution has there anyway, and the columns of dx past the image are dropped.
struct Conv3BwdPad { float *xp, *dyp, *dxp; void* inner; size_t inner_bytes; };

static void conv3f_bwd_pad_carve(int B, int Cin, int Cout, int H, int W, int Wp, int mode, bool want_dx, bool want_bias, CarveF& cv,
                                 Conv3BwdPad& p) {
    p.xp = cv.floats((size_t)B * Cin * H * Wp);
    p.dyp = cv.floats((size_t)B * Cout * H * Wp);                 // every mode's dy has Cout * H * W values per sample
    p.dxp = want_dx ? cv.floats((size_t)B * Cin * H * Wp) : nullptr;
    CarveF in(nullptr); Conv3BwdF b{};
    conv3f_bwd_carve(B, Cin, Cout, H, Wp, mode, want_dx, want_bias, in, b);
    p.inner_bytes = in.total();
    p.inner = cv.bytes(p.inner_bytes);
}

size_t conv3x3_bwd_ws_bytes(int B, int Cin, int Cout, int H, int W, int mode, int want_dx, int want_bias) {
    CarveF cv(nullptr);
    if (W % 4 != 0) {
        Conv3BwdPad p{};
        conv3f_bwd_pad_carve(B, Cin, Cout, H, W, (int)round_up(W, 4), mode, want_dx != 0, want_bias != 0, cv, p);
        return cv.total();
    }
    Conv3BwdF b{};
    conv3f_bwd_carve(B, Cin, Cout, H, W, mode, want_dx != 0, want_bias != 0, cv, b);
    return cv.total();
}

int conv3x3_bwd_launch(const float* x, const float* dy, const float* w, float* dx, float* dweight, float* dbias, int B, int Cin,
                       int Cout, int H, int W, int mode, void* ws, size_t ws_bytes, hipStream_t stream) {
    RST_REQUIRE(x && dy && w && dweight && B > 0, "conv3x3 backward: bad operand");
    RST_REQUIRE(conv3x3_bwd_supported(Cin, Cout, H, W, mode),
                "conv3x3 backward: unsupported shape Cin=%d Cout=%d H=%d W=%d mode=%d (unshuffle: even H and W; shuffle: Cout %% 4 == 0)",
                Cin, Cout, H, W, mode);
    RST_REQUIRE(aligned16(x) && aligned16(dy) && (dx == nullptr || aligned16(dx)), "conv3x3 backward: tensors must be 16-byte aligned");
    const bool want_dx = dx != nullptr, want_bias = dbias != nullptr;
    const size_t need = conv3x3_bwd_ws_bytes(B, Cin, Cout, H, W, mode, want_dx, want_bias);
    if (ws == nullptr || !aligned16(ws) || ws_bytes < need)
        return fail(RST_ERR_WORKSPACE, "conv3x3 backward: workspace %zu < %zu bytes", ws_bytes, need);
    if (W % 4 != 0) {
        const int Wp = (int)round_up(W, 4);
        CarveF pc(ws); Conv3BwdPad p{};
        conv3f_bwd_pad_carve(B, Cin, Cout, H, W, Wp, mode, want_dx, want_bias, pc, p);
        RST_TRY(copy_cols_launch(x, p.xp, (int64_t)B * Cin * H, W, Wp, stream));
        // rows of dy: mode 0 [Cout][H] x W, mode 1 [4 Cout][H / 2] x W / 2, mode 2 [Cout / 4][2 H] x 2 W
        const int64_t rows = (int64_t)B * Cout * H * (mode == 1 ? 2 : 1) / (mode == 2 ? 2 : 1);
        const int f2 = mode == 1 ? 1 : mode == 2 ? 4 : 2;             // twice the width factor
        RST_TRY(copy_cols_launch(dy, p.dyp, rows, W * f2 / 2, Wp * f2 / 2, stream));
        RST_TRY(conv3x3_bwd_launch(p.xp, p.dyp, w, p.dxp, dweight, dbias, B, Cin, Cout, H, Wp, mode, p.inner, p.inner_bytes, stream));
        if (want_dx) RST_TRY(copy_cols_launch(p.dxp, dx, (int64_t)B * Cin * H, Wp, W, stream));
        return RST_OK;
    }
    CarveF cv(ws); Conv3BwdF b{};
    conv3f_bwd_carve(B, Cin, Cout, H, W, mode, want_dx, want_bias, cv, b);
    const int64_t N = (int64_t)H * W;
    const float* dyc = dy;
    if (mode == 1) {          // forward: conv -> PixelUnshuffle; dy is [4 Cout][H/2][W/2]: its gradient is a PixelShuffle
        RST_TRY(pixel_shuffle2_launch(true, dy, b.dyc, B, Cout, H / 2, W / 2, stream));
        dyc = b.dyc;
    } else if (mode == 2) {   // forward: conv -> PixelShuffle; dy is [Cout / 4][2H][2W]
        RST_TRY(pixel_shuffle2_launch(false, dy, b.dyc, B, Cout / 4, H, W, stream));
        dyc = b.dyc;
    }
    // weight gradient
    RST_TRY(im2col_launch(x, b.col, B, Cin, H, W, stream));
    {
        GemmNT g{};
        g.A = dyc; g.a_s1 = (int64_t)Cout * N; g.a_s2 = 0; g.lda = N;
        g.Bm = b.col; g.b_s1 = (int64_t)9 * Cin * N; g.b_s2 = 0; g.ldb = N;
        g.M1 = Cout; g.M2 = 9 * Cin; g.N = N; g.nb1 = B; g.nb2 = 1;
        g.partial = b.part; g.sqA_partial = nullptr; g.sqB_partial = nullptr;
        RST_TRY(gemm_nt_launch(g, b.plan, stream));
        RST_TRY(reduce_slabs_launch(b.part, b.red, (int64_t)Cout * 9 * Cin, 1, B * b.plan.splits, 0, stream));
        const int64_t n = (int64_t)Cout * Cin * 9;
        hipLaunchKernelGGL(conv3_dw_permute_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, b.red, dweight, Cout, Cin);
        RST_LAUNCH_CHECK();
    }
    if (want_bias) RST_TRY(rowsum_launch(dyc, (int64_t)Cout * N, dbias, B, Cout, N, b.rsws, b.rsws_bytes, stream));
    // data gradient
    if (want_dx) {
        const int64_t n = (int64_t)Cin * 9 * Cout;
        hipLaunchKernelGGL(conv3_wt_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, w, b.wt, Cout, Cin);
        RST_LAUNCH_CHECK();
        RST_TRY(im2col_launch(dyc, b.col, B, Cout, H, W, stream));
        GemmNN g{};
        g.A = b.wt; g.a_s1 = 0; g.a_s2 = 0; g.lda = 9 * Cout; g.transA = 0;
        g.M = Cin; g.K = 9 * Cout;
        g.X = b.col; g.x_s1 = (int64_t)9 * Cout * N; g.x_s2 = 0; g.ldx = N;
        g.Y = dx; g.y_s1 = (int64_t)Cin * N; g.y_s2 = 0; g.ldy = N;
        g.N = N; g.nb1 = B; g.nb2 = 1;
        RST_TRY(gemm_nn_launch(g, stream));
    }
    return RST_OK;
}

// ---- forward on the exact-fp32 matrix instruction (rst_set_exact_fp32(1): activations beyond the f16 range) ----------
// y' = Wm col(x) + bias through the kernels of the backward above: Wm[co][(tap, ci)] = w[co][ci][tap], col(x) written per
// chunk of samples (at most 1 GiB of it at a time), the shuffle of mode 1 / 2 as its own pass.  Nine times the input in
// extra traffic: the price of the rarely taken exact route, not a tuned kernel.
namespace {

__global__ void __launch_bounds__(256)
conv3_wm_kernel(const float* __restrict__ w, float* __restrict__ wm, int Cout, int Cin) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)Cout * 9 * Cin) return;
    const int ci = (int)(i % Cin), tap = (int)((i / Cin) % 9), co = (int)(i / (9 * (int64_t)Cin));
    wm[i] = w[((int64_t)co * Cin + ci) * 9 + tap];
}

int conv3_exact_chunk(int B, int Cin, int64_t N) {
    const int64_t per = 9 * (int64_t)Cin * N;
    const int64_t fit = ((int64_t)1 << 28) / (per > 0 ? per : 1);
    return (int)(fit < 1 ? 1 : fit > B ? B : fit);
}

}  // namespace

size_t conv3x3_exact_ws_bytes(int B, int Cin, int Cout, int H, int W, int mode) {
    const int64_t N = (int64_t)H * W;
    const int bs = conv3_exact_chunk(B, Cin, N);
    CarveF cv(nullptr);
    cv.floats((size_t)Cout * 9 * Cin);
    cv.floats((size_t)bs * 9 * Cin * N);
    if (mode != 0) cv.floats((size_t)bs * Cout * N);
    return cv.total();
}

int conv3x3_exact_launch(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int mode,
                         void* ws, size_t ws_bytes, hipStream_t stream) {
    RST_REQUIRE(x && w && y && B > 0 && W % 4 == 0, "conv3x3 (exact fp32): bad operand");
    RST_REQUIRE(aligned16(x) && aligned16(y), "conv3x3: tensors must be 16-byte aligned");
    const size_t need = conv3x3_exact_ws_bytes(B, Cin, Cout, H, W, mode);
    if (ws == nullptr || !aligned16(ws) || ws_bytes < need)
        return fail(RST_ERR_WORKSPACE, "conv3x3 (exact fp32): workspace %zu < %zu bytes (rst_conv3x3_fwd_workspace_bytes)", ws_bytes, need);
    const int64_t N = (int64_t)H * W;
    const int bs = conv3_exact_chunk(B, Cin, N);
    CarveF cv(ws);
    float* wm = cv.floats((size_t)Cout * 9 * Cin);
    float* col = cv.floats((size_t)bs * 9 * Cin * N);
    float* tmp = mode != 0 ? cv.floats((size_t)bs * Cout * N) : nullptr;
    const int64_t nw = (int64_t)Cout * 9 * Cin;
    hipLaunchKernelGGL(conv3_wm_kernel, dim3((unsigned)cdiv(nw, 256)), dim3(256), 0, stream, w, wm, Cout, Cin);
    RST_LAUNCH_CHECK();
    for (int b0 = 0; b0 < B; b0 += bs) {
        const int nb = B - b0 < bs ? B - b0 : bs;
        RST_TRY(im2col_launch(x + (int64_t)b0 * Cin * N, col, nb, Cin, H, W, stream));
        float* yb = y + (int64_t)b0 * Cout * N;           // every mode writes Cout * H * W values per sample
        GemmNN g{};
        g.A = wm; g.a_s1 = 0; g.a_s2 = 0; g.lda = 9 * Cin; g.transA = 0;
        g.M = Cout; g.K = 9 * Cin;
        g.X = col; g.x_s1 = (int64_t)9 * Cin * N; g.x_s2 = 0; g.ldx = N;
        g.Y = mode != 0 ? tmp : yb; g.y_s1 = (int64_t)Cout * N; g.y_s2 = 0; g.ldy = N;
        g.bias = bias;
        g.N = N; g.nb1 = nb; g.nb2 = 1;
        RST_TRY(gemm_nn_launch(g, stream));
        if (mode == 1) RST_TRY(pixel_shuffle2_launch(false, tmp, yb, nb, Cout, H / 2, W / 2, stream));
        if (mode == 2) RST_TRY(pixel_shuffle2_launch(true, tmp, yb, nb, Cout / 4, H, W, stream));
    }
    return RST_OK;
}

}  // namespace rst

using namespace rst;

extern "C" {

size_t rst_conv3x3_bwd_workspace_bytes(int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode,
                                       int32_t want_dx, int32_t want_bias) {
    if (B <= 0 || !conv3x3_bwd_supported(Cin, Cout, H, W, shuffle_mode)) return 0;
    return conv3x3_bwd_ws_bytes(B, Cin, Cout, H, W, shuffle_mode, want_dx, want_bias);
}
int rst_conv3x3_bwd(const float* x, const float* dy, const float* weight, float* dx, float* dweight, float* dbias, int32_t B,
                    int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode, void* workspace, size_t workspace_bytes,
                    void* stream) {
    return conv3x3_bwd_launch(x, dy, weight, dx, dweight, dbias, B, Cin, Cout, H, W, shuffle_mode, workspace, workspace_bytes,
                              static_cast<hipStream_t>(stream));
}

}  // extern "C"
