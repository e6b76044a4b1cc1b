// Backward of the dense 3x3 convolutions (+ PixelUnshuffle / PixelShuffle) on channel-packed bf16 tensors, and the small
// layout kernels the bf16 training path needs between its stages: what autograd derives from OverlapPatchEmbed,
// Downsample, Upsample, the output convolution and the torch.cat of the skip connections
// (model/Restromer.py:172-215, 381-383, 402, 407, 412) when the network runs on bf16 tensors.
//
//   data gradient   dx = conv3x3(dy', W') with W'[ci][co][a][b] = W[co][ci][2 - a][2 - b]: the forward kernel
//                   (conv3x3_pk.hip) on fragments prepared from the transposed, rotated weights; dy' is dy with the
//                   forward's shuffle undone (pk_shuffle2_kernel)
//   weight gradient dW[co][ci][a][b] = sum over samples and pixels of dy'[co][p] x[ci][p + (a - 1, b - 1)]: x and dy'
//                   are copied into zero-padded images of (H + 2) x (W + 2) pixels (pk_pad_kernel), where a tap is a
//                   plain shift of the pixel index -- a packed chunk is 16 bytes, so every shift keeps the alignment --
//                   and all nine taps are ONE product over the pixels on the matrix pipe (wgrad_pk_kernel with 9 x Cin / 8
//                   virtual channel blocks); the bias gradient is the product's row sums
//   fp32 sums, fp32 parameter gradients, bf16 activation gradients.
#include "kernels.hpp"
#include "bf16_common.hpp"

namespace rst {

namespace {

struct Carve3 {
    char* base; size_t off = 0;
    explicit Carve3(void* b) : base(static_cast<char*>(b)) {}
    void* bytes(size_t n) {
        off = (off + 255) & ~size_t(255);
        void* p = base ? static_cast<void*>(base + off) : nullptr;
        off += n;
        return p;
    }
    size_t total() const { return (off + 255) & ~size_t(255); }
};

// dst[b][cb][(y + 1) * (W + 2) + x + 1] = src[b][cb][y * W + x], zeros on the border; one thread per chunk of dst
__global__ void __launch_bounds__(256)
pk_pad_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int H, int W, int64_t planes) {
    const int Wp = W + 2;
    const int64_t np = (int64_t)(H + 2) * Wp;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * np) return;
    const int64_t pl = i / np, r = i % np;
    const int y = (int)(r / Wp) - 1, x = (int)(r % Wp) - 1;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (y >= 0 && y < H && x >= 0 && x < W) v = src[pl * (int64_t)H * W + (int64_t)y * W + x];
    dst[i] = v;
}

// PixelShuffle(2) / PixelUnshuffle(2) on packed tensors.  `big`: [B][Cb][2h][2w] chunks, `small`: [B][4 Cb][h][w]
// chunks with small[4 c + 2 i + j][y][x] = big[c][2 y + i][2 x + j].  A thread owns the 2 x 2 pixels of one channel
// block of `big` (4 chunks) = one pixel of four consecutive channel blocks of `small` (4 chunks).
// TO_BIG: PixelShuffle (reads small), else PixelUnshuffle (reads big).
template <bool TO_BIG>
__global__ void __launch_bounds__(256)
pk_shuffle2_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int Cb, int h, int w, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (b, cb, y, x) of `small`'s pixel grid
    if (i >= total) return;
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const int64_t bc = i / ((int64_t)w * h);
    const int cb = (int)(bc % Cb);
    const int64_t b = bc / Cb;
    const int64_t hw = (int64_t)h * w;
    // element e (0..31) of the thread's 32 small-channels = channel 32 cb + e = 4 c + 2 i + j with c = 8 cb + e / 4
    const int64_t small0 = ((b * 4 * Cb + 4 * cb) * hw + (int64_t)y * w + x) * 8;     // chunk of small block 4 cb + q: + q hw 8
    const int64_t big0 = ((b * Cb + cb) * 4 * hw + (int64_t)(2 * y) * (2 * w) + 2 * x) * 8;   // pixel (2y + i, 2x + j): + (i 2w + j) 8
    uint16_t v[32];
    if constexpr (TO_BIG) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<u32x4*>(&v[8 * q]) = *reinterpret_cast<const u32x4*>(src + small0 + (int64_t)q * hw * 8);
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) {
            uint16_t o[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) o[c] = v[4 * c + ij];
            *reinterpret_cast<u32x4*>(dst + big0 + ((int64_t)(ij >> 1) * 2 * w + (ij & 1)) * 8) = *reinterpret_cast<const u32x4*>(o);
        }
    } else {
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) {
            uint16_t o[8];
            *reinterpret_cast<u32x4*>(o) = *reinterpret_cast<const u32x4*>(src + big0 + ((int64_t)(ij >> 1) * 2 * w + (ij & 1)) * 8);
#pragma unroll
            for (int c = 0; c < 8; ++c) v[4 * c + ij] = o[c];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<u32x4*>(dst + small0 + (int64_t)q * hw * 8) = *reinterpret_cast<const u32x4*>(&v[8 * q]);
    }
}

// dst[b][dst_cb0 + cb][px] = src[b][src_cb0 + cb][px] for cb < ncb: the halves of a concatenation along the channels
__global__ void __launch_bounds__(256)
pk_copy_blocks_kernel(const u32x4* __restrict__ src, int64_t src_bs, int src_cb0, u32x4* __restrict__ dst, int64_t dst_bs,
                      int dst_cb0, int ncb, int64_t N, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t per = (int64_t)ncb * N;
    const int64_t b = i / per, r = i % per;
    dst[b * dst_bs + (int64_t)dst_cb0 * N + r] = src[b * src_bs + (int64_t)src_cb0 * N + r];
}

// red: [Mp][K9] sums (+ Mp row sums), K9 = 9 x Cinb x 8 with k = (tap, channel block, channel in block)
// -> dW[Cout][Cin][3][3], dbias[Cout]
__global__ void __launch_bounds__(256)
conv3_wgrad_finish_kernel(const float* __restrict__ red, int Mp, int Cinb, int Cout, int Cin, float* __restrict__ dw,
                          float* __restrict__ dbias) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)Cout * Cin * 9;
    if (i < n) {
        const int tap = (int)(i % 9), ci = (int)((i / 9) % Cin), co = (int)(i / (9 * (int64_t)Cin));
        dw[i] = red[(int64_t)co * (9 * Cinb * 8) + (int64_t)tap * Cinb * 8 + ci];
    }
    if (dbias != nullptr && i < Cout) dbias[i] = red[(int64_t)Mp * (9 * Cinb * 8) + i];
}

struct Conv3BwdBufs {
    bf16_t *dyc, *xpad_base, *dypad;
    float *part, *red;
    void* frag;
    WgradPk g;
    int64_t np, margin;
    size_t part_floats, frag_bytes;
};

void conv3_bwd_carve(int B, int Cin, int Cout, int H, int W, int mode, bool want_dx, Carve3& cv, Conv3BwdBufs& b) {
    const int Cinb = (int)cdiv(Cin, 8), Coutb = (int)cdiv(Cout, 8);
    b.np = (int64_t)(H + 2) * (W + 2);
    b.margin = W + 3 + 32;                                   // chunks in front of and behind the padded images of x
    b.dyc = mode != 0 ? static_cast<bf16_t*>(cv.bytes((size_t)B * Coutb * H * W * 16)) : nullptr;
    b.xpad_base = static_cast<bf16_t*>(cv.bytes(((size_t)B * Cinb * b.np + 2 * b.margin) * 16));
    b.dypad = static_cast<bf16_t*>(cv.bytes((size_t)B * Coutb * b.np * 16));
    WgradPk g{};
    g.Ab = Coutb; g.Bb = 9 * Cinb; g.B = B; g.N = b.np; g.with_rowsum = 1;
    g.tap_wp = W + 2; g.tap_cb = Cinb;
    wgrad_pk_plan(&g);
    b.g = g;
    b.part_floats = wgrad_pk_partial_floats(g);
    b.part = static_cast<float*>(cv.bytes(b.part_floats * sizeof(float)));
    b.red = static_cast<float*>(cv.bytes(((size_t)Coutb * 8 * 9 * Cinb * 8 + Coutb * 8) * sizeof(float)));
    b.frag_bytes = want_dx ? conv3x3_pk_ws_bytes(Cout, Cin) : 0;
    b.frag = want_dx ? cv.bytes(b.frag_bytes) : nullptr;
}

}  // namespace

int pk_shuffle2_launch(const bf16_t* src, bf16_t* dst, int B, int Cb_big, int h, int w, int to_big, hipStream_t stream) {
    RST_REQUIRE(src && dst && B > 0 && Cb_big > 0 && h > 0 && w > 0, "bf16 pixel shuffle: bad operand");
    RST_REQUIRE(aligned16(src) && aligned16(dst), "bf16 pixel shuffle: tensors must be 16-byte aligned");
    const int64_t total = (int64_t)B * Cb_big * h * w;
    RST_REQUIRE(cdiv(total, 256) < (1ll << 31), "bf16 pixel shuffle: grid too large");
    ProfScope prof(stream, "pk_shuffle2", 0.0, 2.0 * 64.0 * total);
    if (to_big) hipLaunchKernelGGL(pk_shuffle2_kernel<true>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream,
                                   reinterpret_cast<const uint16_t*>(src), reinterpret_cast<uint16_t*>(dst), Cb_big, h, w, total);
    else hipLaunchKernelGGL(pk_shuffle2_kernel<false>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream,
                            reinterpret_cast<const uint16_t*>(src), reinterpret_cast<uint16_t*>(dst), Cb_big, h, w, total);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

int pk_copy_blocks_launch(const bf16_t* src, int src_cb_total, int src_cb0, bf16_t* dst, int dst_cb_total, int dst_cb0, int ncb,
                          int B, int64_t N, hipStream_t stream) {
    RST_REQUIRE(src && dst && B > 0 && ncb > 0 && N > 0, "bf16 channel-block copy: bad operand");
    RST_REQUIRE(src_cb0 >= 0 && dst_cb0 >= 0 && src_cb0 + ncb <= src_cb_total && dst_cb0 + ncb <= dst_cb_total,
                "bf16 channel-block copy: block range outside a tensor");
    RST_REQUIRE(aligned16(src) && aligned16(dst), "bf16 channel-block copy: tensors must be 16-byte aligned");
    const int64_t total = (int64_t)B * ncb * N;
    RST_REQUIRE(cdiv(total, 256) < (1ll << 31), "bf16 channel-block copy: grid too large");
    ProfScope prof(stream, "pk_copy_blocks", 0.0, 2.0 * 16.0 * total);
    hipLaunchKernelGGL(pk_copy_blocks_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream,
                       reinterpret_cast<const u32x4*>(src), (int64_t)src_cb_total * N, src_cb0, reinterpret_cast<u32x4*>(dst),
                       (int64_t)dst_cb_total * N, dst_cb0, ncb, N, total);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// (mode 0 takes any Cout here: the last layer, 48 -> 3, whose forward is another kernel)
bool conv3x3_pk_bwd_supported(int Cin, int Cout, int H, int W, int mode, bool want_dx) {
    if (Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 2) return false;
    if (mode == 1 && (H % 2 != 0 || W % 2 != 0 || Cout % 8 != 0)) return false;
    if (mode == 2 && Cout % 32 != 0) return false;
    return !want_dx || Cin % 8 == 0;
}

size_t conv3x3_pk_bwd_ws_bytes(int B, int Cin, int Cout, int H, int W, int mode, int want_dx) {
    Carve3 cv(nullptr); Conv3BwdBufs b{};
    conv3_bwd_carve(B, Cin, Cout, H, W, mode, want_dx != 0, cv, b);
    return cv.total();
}

// x: the forward's packed input [B][Cin/8][H][W]; dy: the packed gradient of the forward's output (after its shuffle);
// dx (optional): packed gradient of x; dweight [Cout][Cin][3][3], dbias (optional) [Cout]: overwritten
int conv3x3_pk_bwd_launch(const bf16_t* x, const bf16_t* dy, const float* w, bf16_t* dx, float* dweight, float* dbias, int B, int Cin,
                          int Cout, int H, int W, int mode, void* ws, size_t ws_bytes, hipStream_t stream) {
    RST_REQUIRE(x && dy && w && dweight && B > 0 && B <= 65535, "bf16 conv3x3 backward: bad operand");
    RST_REQUIRE(conv3x3_pk_bwd_supported(Cin, Cout, H, W, mode, dx != nullptr),
                "bf16 conv3x3 backward: unsupported shape Cin=%d Cout=%d H=%d W=%d mode=%d (shuffles: Cout %% 8 == 0, shuffle: Cout %% 32 == 0, "
                "unshuffle: even H, W; data gradient: Cin %% 8 == 0)", Cin, Cout, H, W, mode);
    RST_REQUIRE(aligned16(x) && aligned16(dy) && (dx == nullptr || aligned16(dx)), "bf16 conv3x3 backward: tensors must be 16-byte aligned");
    const bool want_dx = dx != nullptr;
    if (ws == nullptr || !aligned16(ws) || ws_bytes < conv3x3_pk_bwd_ws_bytes(B, Cin, Cout, H, W, mode, want_dx))
        return fail(RST_ERR_WORKSPACE, "bf16 conv3x3 backward: workspace %zu < %zu bytes", ws_bytes,
                    conv3x3_pk_bwd_ws_bytes(B, Cin, Cout, H, W, mode, want_dx));
    Carve3 cv(ws); Conv3BwdBufs b{};
    conv3_bwd_carve(B, Cin, Cout, H, W, mode, want_dx, cv, b);
    const int Cinb = (int)cdiv(Cin, 8), Coutb = (int)cdiv(Cout, 8);
    // the gradient of the convolution's own output: undo the forward's shuffle
    const bf16_t* dyc = dy;
    if (mode == 1) {          // forward: conv -> PixelUnshuffle: dy is [4 Cout][H/2][W/2]; its gradient is a PixelShuffle
        RST_TRY(pk_shuffle2_launch(dy, b.dyc, B, Coutb, H / 2, W / 2, 1, stream));
        dyc = b.dyc;
    } else if (mode == 2) {   // forward: conv -> PixelShuffle: dy is [Cout / 4][2H][2W]
        RST_TRY(pk_shuffle2_launch(dy, b.dyc, B, Coutb / 4, H, W, 0, stream));
        dyc = b.dyc;
    }
    if (want_dx)
        RST_TRY(conv3x3_pk_launch(dyc, w, nullptr, dx, B, Cout, Cin, H, W, 0, b.frag, b.frag_bytes, stream, 1));
    // weight gradient: padded images, one product for the nine taps
    bf16_t* xpad = b.xpad_base + b.margin * 8;
    RST_HIP(hipMemsetAsync(b.xpad_base, 0, (size_t)b.margin * 16, stream));
    RST_HIP(hipMemsetAsync(xpad + (size_t)B * Cinb * b.np * 8, 0, (size_t)b.margin * 16, stream));
    {
        const int64_t nx = (int64_t)B * Cinb * b.np, ny = (int64_t)B * Coutb * b.np;
        RST_REQUIRE(cdiv(nx, 256) < (1ll << 31) && cdiv(ny, 256) < (1ll << 31), "bf16 conv3x3 backward: grid too large");
        ProfScope prof(stream, "pk_pad", 0.0, 2.0 * 16.0 * (nx + ny));
        hipLaunchKernelGGL(pk_pad_kernel, dim3((unsigned)cdiv(nx, 256)), dim3(256), 0, stream, reinterpret_cast<const u32x4*>(x),
                           reinterpret_cast<u32x4*>(xpad), H, W, (int64_t)B * Cinb);
        hipLaunchKernelGGL(pk_pad_kernel, dim3((unsigned)cdiv(ny, 256)), dim3(256), 0, stream, reinterpret_cast<const u32x4*>(dyc),
                           reinterpret_cast<u32x4*>(b.dypad), H, W, (int64_t)B * Coutb);
        RST_LAUNCH_CHECK();
    }
    WgradPk g = b.g;
    g.A = b.dypad; g.a_bs = (int64_t)Coutb * b.np * 8;
    g.Bm = xpad; g.b_bs = (int64_t)Cinb * b.np * 8;
    g.partial = b.part;
    RST_TRY(wgrad_pk_launch(g, stream));
    const int64_t rec = (int64_t)Coutb * 8 * 9 * Cinb * 8 + Coutb * 8;
    RST_TRY(reduce_slabs_launch(b.part, b.red, rec, 1, B * g.splits, 0, stream));
    const int64_t n = (int64_t)Cout * Cin * 9;
    hipLaunchKernelGGL(conv3_wgrad_finish_kernel, dim3((unsigned)cdiv(n > Cout ? n : Cout, 256)), dim3(256), 0, stream, b.red, Coutb * 8,
                       Cinb, Cout, Cin, dweight, dbias);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst

using namespace rst;

extern "C" {

size_t rst_bf16_conv3x3_bwd_workspace_bytes(int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode,
                                            int32_t want_dx) {
    if (B <= 0 || !rst::conv3x3_pk_bwd_supported(Cin, Cout, H, W, shuffle_mode, want_dx != 0)) return 0;
    return rst::conv3x3_pk_bwd_ws_bytes(B, Cin, Cout, H, W, shuffle_mode, want_dx);
}
int rst_bf16_conv3x3_bwd(const void* x_packed, const void* dy_packed, const float* weight, void* dx_packed, float* dweight,
                         float* dbias, int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode,
                         void* workspace, size_t workspace_bytes, void* stream) {
    return rst::conv3x3_pk_bwd_launch(static_cast<const rst::bf16_t*>(x_packed), static_cast<const rst::bf16_t*>(dy_packed), weight,
                                      static_cast<rst::bf16_t*>(dx_packed), dweight, dbias, B, Cin, Cout, H, W, shuffle_mode,
                                      workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
int rst_bf16_pixel_shuffle2(const void* x_packed, void* y_packed, int32_t B, int32_t C_out, int32_t H_in, int32_t W_in, void* stream) {
    // [B][4 C_out][H_in][W_in] -> [B][C_out][2 H_in][2 W_in]
    RST_REQUIRE(C_out > 0 && C_out % 8 == 0, "bf16 pixel shuffle: the result needs whole channel blocks (C_out %% 8 == 0)");
    return rst::pk_shuffle2_launch(static_cast<const rst::bf16_t*>(x_packed), static_cast<rst::bf16_t*>(y_packed), B, C_out / 8, H_in, W_in,
                                   1, static_cast<hipStream_t>(stream));
}
int rst_bf16_pixel_unshuffle2(const void* x_packed, void* y_packed, int32_t B, int32_t C_in, int32_t H_in, int32_t W_in, void* stream) {
    // [B][C_in][H_in][W_in] -> [B][4 C_in][H_in / 2][W_in / 2]
    RST_REQUIRE(C_in > 0 && C_in % 8 == 0 && H_in % 2 == 0 && W_in % 2 == 0, "bf16 pixel unshuffle: C_in %% 8 == 0 and even H, W");
    return rst::pk_shuffle2_launch(static_cast<const rst::bf16_t*>(x_packed), static_cast<rst::bf16_t*>(y_packed), B, C_in / 8, H_in / 2,
                                   W_in / 2, 0, static_cast<hipStream_t>(stream));
}
int rst_bf16_copy_channel_blocks(const void* src_packed, int32_t src_C, int32_t src_c0, void* dst_packed, int32_t dst_C, int32_t dst_c0,
                                 int32_t nC, int32_t B, int64_t N, void* stream) {
    RST_REQUIRE(src_C % 8 == 0 && dst_C % 8 == 0 && src_c0 % 8 == 0 && dst_c0 % 8 == 0 && nC % 8 == 0,
                "bf16 channel copy: channel counts and offsets in whole blocks of 8");
    return rst::pk_copy_blocks_launch(static_cast<const rst::bf16_t*>(src_packed), src_C / 8, src_c0 / 8, static_cast<rst::bf16_t*>(dst_packed),
                                      dst_C / 8, dst_c0 / 8, nC / 8, B, N, static_cast<hipStream_t>(stream));
}

}  // extern "C"
