// Dense 3x3 convolution, zero padding 1, as an implicit GEMM on the f16 matrix instructions at fp32 accuracy, with
// PixelUnshuffle(2) / PixelShuffle(2) as its epilogue: OverlapPatchEmbed, Downsample, Upsample of the reference
// (model/Restromer.py:172-215) in one kernel each.
//
// Arithmetic as gemm_f16s.hip: both operands split in two f16 halves (22 significant bits), three
// v_mfma_f32_16x16x32_f16 per 32 input channels and tap.  Y[co][px] = sum over (dy, ci, dx) W[co][ci][dy][dx] X[ci][px + (dy, dx)]:
//   * the weights are split once per call into operand fragments in global memory, ordered [16-row tile][dy][32-channel
//     chunk][dx][hi, lo] (conv3x3_f16s_prep);
//   * a block owns 32 RT output channels x 128 pixels (TH rows x TW columns of one sample, TW = min(W, 128)).  Per
//     (dy, channel chunk) ONE slab of the input -- 32 channels x 128 pixels of the rows shifted by dy, plus the four
//     pixels either side when the image is wider than the tile -- comes in by LDS-DMA, and the weight fragments of its
//     three horizontal taps likewise;
//   * a landed slab is converted ONCE, by all eight waves together, into an operand image in LDS ([pixel slot][lane
//     group] x 16 B, hi and lo; pixels outside the image become zeros there); the three horizontal taps are three reads
//     of that image at slots shifted by one -- a tap costs two ds_read_b128 per 16 pixels instead of eight scalar reads
//     and a conversion (8 convolutions of the network: 4.08 -> 3.28 ms);
//   * one barrier per slab: [slab s+1 and the weights of slab s landed, image s complete, everyone done with slab s-1]
//     -> issue slab s+2 and the weights of slab s+1 -> convert slab s+1 -> multiply slab s.  145 KB of LDS at RT = 3,
//     one block per CU.  Measured at 96 -> 192 channels, 16 x 256 x 256: 1.44 ms = 720 TFLOP/s of f16 instructions,
//     about a third of the matrix pipe's cycles; opposite conversion / multiplication order in the two waves of a SIMD
//     and a third accumulator set (no back-to-back dependent instructions) changed nothing.
// Epilogue through a wave-private LDS transpose: mode 0 plain NCHW rows, mode 1 PixelUnshuffle(2)
// (out[4 m + 2 (y & 1) + (x & 1)][y / 2][x / 2]), mode 2 PixelShuffle(2) (out[m / 4][2 y + (m % 4) / 2][2 x + m % 2]).
#include "kernels.hpp"

namespace rst {

namespace {

typedef __attribute__((address_space(1))) const void* c3_gptr;
typedef __attribute__((address_space(3))) void* c3_lptr;
typedef _Float16 c3_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 c3_h2 __attribute__((ext_vector_type(2)));
typedef float c3_f32x2 __attribute__((ext_vector_type(2)));

constexpr int C3_XPIECE = 272;                 // floats between the 2-row pieces of a slab (64 B apart: bank quarters)
constexpr int C3_XS = 16 * C3_XPIECE + 256;    // slab + halo piece
constexpr int C3_SCR_LD = 36;

struct Conv3F {
    const float* x; const void* afrag; const float* bias; float* y;
    int Cin, Cout, H, W, TW, TH, kch, mblocks, mtiles, tiles_x, tiles_y, mode, walk;
};

__global__ void __launch_bounds__(64)
conv3x3_f16s_prep_kernel(const float* __restrict__ w, int Cout, int Cin, int kch, c3_h8* __restrict__ frag) {
    const int lane = threadIdx.x, m = lane & 15, g = lane >> 4;
    const int mt = blockIdx.x, q = blockIdx.y;             // q = (dyi * kch + kc) * 3 + dxi
    const int dxi = q % 3, kc = (q / 3) % kch, dyi = q / (3 * kch);
    const int row = 16 * mt + m;
    c3_h8 hi, lo;
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
        const int ci = 32 * kc + 2 * g + 8 * (idx >> 1) + (idx & 1);
        const float v = (row < Cout && ci < Cin) ? w[(((int64_t)row * Cin + ci) * 3 + dyi) * 3 + dxi] : 0.f;
        const _Float16 h = (_Float16)v;
        hi[idx] = h;
        lo[idx] = (_Float16)((v - (float)h) * 2048.f);
    }
    c3_h8* out = frag + ((int64_t)mt * gridDim.y + q) * 2 * 64 + lane;
    out[0] = hi;
    out[64] = lo;
}

constexpr int C3_SLOTS = 136;                  // pixel slots of an operand image: TH rows x (TW + 2) columns, at most 4 x 34
constexpr int C3_IMGH = C3_SLOTS * 4 * 4;      // floats of one half (hi or lo) of an image: [slot][lane group] x 16 B

template <int RT>
__global__ void __launch_bounds__(512)
conv3x3_f16s_kernel(Conv3F a) {
    constexpr int AS = 3 * 2 * RT * 2 * 256;              // three taps x 2 RT tiles x (hi, lo) x 1 KiB
    constexpr int PW = (12 * RT + 7) / 8;                 // weight pieces per wave and slab
    extern __shared__ __attribute__((aligned(16))) float c3_ring[];
    float* xring = c3_ring;                               // [2][C3_XS] fp32 slabs as they land
    float* wring = xring + 2 * C3_XS;                     // [2][AS] weight fragments
    float* imgs = wring + 2 * AS;                         // [2][hi, lo][C3_IMGH] split-f16 operand images

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kk = lane >> 4;
    const int nq = wave & 3, mh = wave >> 2;
    // a tile reads the rows of the tiles above and below it: each XCD takes a contiguous band of tiles
    int64_t bx = blockIdx.x, by = blockIdx.y;
    if (a.walk) {
        const int64_t lin = xcd_walk(by * gridDim.x + bx, (int64_t)gridDim.x * gridDim.y);
        bx = lin % gridDim.x; by = lin / gridDim.x;
    }
    const int mb = (int)(bx % a.mblocks);
    const int tile = (int)(bx / a.mblocks);
    const int b = (int)by;
    const int m0 = mb * 32 * RT;
    const int x0 = (tile % a.tiles_x) * a.TW, y0 = (tile / a.tiles_x) * a.TH;
    const int H = a.H, W = a.W, TW = a.TW;
    const int64_t HW = (int64_t)H * W;
    const float* Xb = a.x + (int64_t)b * a.Cin * HW;
    const char* Af = static_cast<const char*>(a.afrag);
    const int NS = 3 * a.kch, NQ = 9 * a.kch;

    // ---- DMA pieces of this wave: slab pieces 0..15 (two channel rows each), 16 the halo piece; weight pieces
    // (tap, tile, hi/lo) of 1 KiB ----
    auto issue_x = [&](int s) __attribute__((always_inline)) {
        float* st = xring + (s & 1) * C3_XS;
        const int dy = s / a.kch - 1, kc = s % a.kch;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int id = wave + 8 * i;
            if (id > 16) break;
            int chrow, ys, xs;
            float* dst;
            if (id < 16) {
                chrow = 2 * id + (lane >> 5);
                const int n = (lane & 31) * 4;
                ys = y0 + n / TW + dy; xs = x0 + n % TW;
                if (xs > W - 4) xs = W - 4;                           // right of the image (last tile of a row): zeroed when converted
                dst = st + id * C3_XPIECE;
            } else {
                chrow = lane >> 1;
                int col = (lane & 1) ? x0 + TW : x0 - 4;
                col = col < 0 ? 0 : (col > W - 4 ? W - 4 : col);      // outside the image: any valid address, zeroed when converted
                ys = y0 + dy; xs = col;
                dst = st + 16 * C3_XPIECE;
            }
            int ch = 32 * kc + chrow;
            if (ch > a.Cin - 1) ch = a.Cin - 1;                        // channels past Cin meet zero weights
            ys = ys < 0 ? 0 : (ys > H - 1 ? H - 1 : ys);               // rows outside the image are zeroed when converted
            const float* src = Xb + (int64_t)ch * HW + (int64_t)ys * W + xs;
            __builtin_amdgcn_global_load_lds((c3_gptr)src, (c3_lptr)dst, 16, 0, 0);
        }
    };
    auto issue_w = [&](int s) __attribute__((always_inline)) {
        float* st = wring + (s & 1) * AS;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int j = wave + 8 * i;
            if (j >= 12 * RT) break;
            const int hl = j & 1, t = (j >> 1) % (2 * RT), dxi = j / (4 * RT);
            int mt = m0 / 16 + t;
            if (mt > a.mtiles - 1) mt = a.mtiles - 1;
            const char* src = Af + (((int64_t)mt * NQ + (int64_t)s * 3 + dxi) * 2 + hl) * 1024 + lane * 16;
            float* dst = st + ((dxi * 2 * RT + t) * 2 + hl) * 256;
            __builtin_amdgcn_global_load_lds((c3_gptr)src, (c3_lptr)dst, 16, 0, 0);
        }
    };

    // ---- conversion: every lane turns the 8 channels of one (pixel, lane group) of a landed slab into the hi and lo
    // chunks of the operand image, once per slab; the three horizontal taps then read the image at shifted slots.
    // Pixels outside the image become zeros here.  Slot of tile pixel (ty, tx): ty (TW + 2) + tx + 1. ----
    const int cn = wave * 16 + (lane >> 2), cg = lane & 3;             // this lane's pixel of the tile and lane group
    const int cty = cn / TW, ctx = cn % TW;
    const int cslot = cty * (TW + 2) + ctx + 1;
    auto split_store = [&](const float* v, bool ok, float* im, int slot, int g) __attribute__((always_inline)) {
        c3_h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const c3_f32x2 vv = {ok ? v[2 * j] : 0.f, ok ? v[2 * j + 1] : 0.f};
            const c3_h2 h = __builtin_convertvector(vv, c3_h2);
            const c3_f32x2 sc = vv * 2048.f;
            const c3_f32x2 r2 = {fmaf((float)h[0], -2048.f, sc[0]), fmaf((float)h[1], -2048.f, sc[1])};
            const c3_h2 l = __builtin_convertvector(r2, c3_h2);
            hi[2 * j] = h[0]; hi[2 * j + 1] = h[1];
            lo[2 * j] = l[0]; lo[2 * j + 1] = l[1];
        }
        reinterpret_cast<c3_h8*>(im)[slot * 4 + g] = hi;
        reinterpret_cast<c3_h8*>(im + C3_IMGH)[slot * 4 + g] = lo;
    };
    auto convert = [&](int s) __attribute__((always_inline)) {
        const float* st = xring + (s & 1) * C3_XS;
        float* im = imgs + (s & 1) * 2 * C3_IMGH;
        const int dy = s / a.kch - 1;
        {
            float v[8];
            const float* xs = st + cg * C3_XPIECE + cn;
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = xs[4 * j * C3_XPIECE]; v[2 * j + 1] = xs[4 * j * C3_XPIECE + 128]; }
            const int yy = y0 + cty + dy;
            split_store(v, yy >= 0 && yy < H && x0 + ctx < W, im, cslot, cg);
        }
        if (TW == 128 && wave == 0 && lane < 8) {
            // the two columns either side of a full-width tile (one tile row): halo piece [channel][side][4 px]
            const int side = lane >> 2, g = lane & 3;
            float v[8];
            const float* hs = st + 16 * C3_XPIECE + (side ? 4 : 3);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = hs[(2 * g + 8 * j) * 8]; v[2 * j + 1] = hs[(2 * g + 8 * j + 1) * 8]; }
            const int yy = y0 + dy, xx = side ? x0 + TW : x0 - 1;
            split_store(v, yy >= 0 && yy < H && xx >= 0 && xx < W, im, side ? TW + 1 : 0, g);
        }
    };
    if (TW < 128) {
        // narrower images: a tile spans the width, the columns either side are outside the image -- zero, written once
        for (int e = tid; e < 2 * a.TH * 4 * 4; e += 512) {
            const int buf = e & 1, hl = (e >> 1) & 1, g = (e >> 2) & 3, r = e >> 4;      // r: (tile row, side)
            const int slot = (r >> 1) * (TW + 2) + ((r & 1) ? TW + 1 : 0);
            reinterpret_cast<c3_h8*>(imgs + (buf * 2 + hl) * C3_IMGH)[slot * 4 + g] = c3_h8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    // slots of this lane's two pixels (one per 16-pixel tile) for the multiplication
    int mslot[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 32 * nq + 16 * t + lm;
        mslot[t] = (n / TW) * (TW + 2) + n % TW + 1;
    }

    f32x4 acc1[RT][2], acc2[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + mh * 16 * RT + 16 * i + 4 * kk + r;
                b4[r] = a.bias[m < a.Cout ? m : a.Cout - 1];
            }
        }
        acc1[i][0] = b4; acc1[i][1] = b4;
        acc2[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- pipeline: one barrier per slab.  Iteration s: [slab s + 1 and the weights of slab s have landed, image s is
    // complete, everyone is done multiplying slab s - 1] -> issue slab s + 2 (into the stage slab s was converted out
    // of) and the weights of slab s + 1 -> convert slab s + 1 -> multiply slab s ----
    issue_x(0); issue_w(0);
    if (NS > 1) issue_x(1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    convert(0);
    for (int s = 0; s < NS; ++s) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s + 2 < NS) issue_x(s + 2);
        if (s + 1 < NS) { issue_w(s + 1); convert(s + 1); }
        const float* im = imgs + (s & 1) * 2 * C3_IMGH;
        const float* ws = wring + (s & 1) * AS;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            c3_h8 xh[2], xl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                xh[t] = reinterpret_cast<const c3_h8*>(im)[(mslot[t] + d - 1) * 4 + kk];
                xl[t] = reinterpret_cast<const c3_h8*>(im + C3_IMGH)[(mslot[t] + d - 1) * 4 + kk];
            }
            const c3_h8* af = reinterpret_cast<const c3_h8*>(ws) + (d * 2 * RT + mh * RT) * 2 * 64 + lane;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const c3_h8 ah = af[(i * 2 + 0) * 64], al = af[(i * 2 + 1) * 64];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc1[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[t], acc1[i][t], 0, 0, 0);
                    acc2[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[t], acc2[i][t], 0, 0, 0);
                    acc2[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[t], acc2[i][t], 0, 0, 0);
                }
            }
        }
    }
    float* ring = c3_ring;

    // ---- epilogue: the wave's 16 RT rows x 32 pixels (32 consecutive columns of ONE image row) ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the ring is free: reuse it
    float* scr = ring + wave * 2 * 16 * C3_SCR_LD;
    const int nw = 32 * nq;
    const int yw = y0 + nw / TW, xw = x0 + nw % TW;
    float* Yb = a.y + (int64_t)b * a.Cout * HW;          // every mode writes Cout * H * W values per sample
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        float* sl = scr + (i & 1) * 16 * C3_SCR_LD;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sl[(4 * kk + r) * C3_SCR_LD + lm] = fmaf(acc2[i][0][r], 1.0f / 2048.f, acc1[i][0][r]);
            sl[(4 * kk + r) * C3_SCR_LD + 16 + lm] = fmaf(acc2[i][1][r], 1.0f / 2048.f, acc1[i][1][r]);
        }
        const int mbase = m0 + mh * 16 * RT + 16 * i;
        if (a.mode == 0) {
            const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = rr + 8 * h, m = mbase + row;
                const f32x4 v = *reinterpret_cast<const f32x4*>(&sl[row * C3_SCR_LD + c4]);
                if (m < a.Cout && xw + c4 < W) *reinterpret_cast<f32x4*>(Yb + (int64_t)m * HW + (int64_t)yw * W + xw + c4) = v;
            }
        } else if (a.mode == 2) {
            // out[m / 4][2 y + (m % 4) / 2][2 x + m % 2]: lane -> output row (c, i) = lane >> 3, 8 output columns
            const int orow = lane >> 3, seg = lane & 7;
            const int mrow = 4 * (orow >> 1) + 2 * (orow & 1);
            const f32x4 e = *reinterpret_cast<const f32x4*>(&sl[mrow * C3_SCR_LD + 4 * seg]);
            const f32x4 o = *reinterpret_cast<const f32x4*>(&sl[(mrow + 1) * C3_SCR_LD + 4 * seg]);
            if (mbase + mrow < a.Cout && xw + 4 * seg < W) {
                float* dst = Yb + (int64_t)((mbase + mrow) / 4) * 4 * HW + (int64_t)(2 * yw + (orow & 1)) * 2 * W + 2 * xw + 8 * seg;
                *reinterpret_cast<f32x4*>(dst) = f32x4{e.x, o.x, e.y, o.y};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{e.z, o.z, e.w, o.w};
            }
        } else {
            // out[4 m + 2 (y & 1) + (x & 1)][y / 2][x / 2]: lane -> row m = lane >> 2, 8 pixels -> 4 even + 4 odd columns
            const int row = lane >> 2, part = lane & 3, m = mbase + row;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(&sl[row * C3_SCR_LD + 8 * part]);
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(&sl[row * C3_SCR_LD + 8 * part + 4]);
            if (m < a.Cout && xw + 8 * part < W) {
                float* dst = Yb + (int64_t)(4 * m + 2 * (yw & 1)) * (HW / 4) + (int64_t)(yw >> 1) * (W / 2) + ((xw + 8 * part) >> 1);
                *reinterpret_cast<f32x4*>(dst) = f32x4{p0.x, p0.z, p1.x, p1.z};
                *reinterpret_cast<f32x4*>(dst + HW / 4) = f32x4{p0.y, p0.w, p1.y, p1.w};
            }
        }
    }
}

// ---- few input channels (the patch embedding, 3 -> 48: model/Restromer.py:172-183) ----------------------------------
// A 32-channel chunk of the implicit GEMM would be nine tenths zeros, so this shape stays on the vector pipe in exact
// fp32: one thread = 4 consecutive pixels of a row x all output channels, four at a time; its CIN x 3 x 6 input window
// sits in registers, the weights [ci][dy][dx][Cout] in LDS (one 16-byte broadcast read serves 16 multiply-adds); every
// store is a float4 of one output plane, consecutive lanes consecutive chunks.
template <int CIN>
__global__ void __launch_bounds__(256)
conv3x3_fewin_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                     float* __restrict__ y, int Cout, int H, int W, int64_t quads) {
    extern __shared__ __attribute__((aligned(16))) float c3_fewin_w[];
    for (int e = threadIdx.x; e < CIN * 9 * Cout; e += 256) {
        const int co = e % Cout, t = e / Cout;
        c3_fewin_w[e] = w[(int64_t)co * CIN * 9 + t];
    }
    __syncthreads();
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= quads) return;
    const int W4 = W >> 2;
    const int xq = (int)(q % W4), row = (int)((q / W4) % H);
    const int64_t b = q / ((int64_t)W4 * H);
    const int64_t HW = (int64_t)H * W;
    float in[CIN][3][6];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = row + dy - 1;
            const bool rok = yy >= 0 && yy < H;
            const float* src = x + (b * CIN + ci) * HW + (int64_t)(rok ? yy : row) * W + 4 * xq;
            const f32x4 mid = *reinterpret_cast<const f32x4*>(src);
            const float lf = xq > 0 ? src[-1] : 0.f, rt = xq < W4 - 1 ? src[4] : 0.f;
            in[ci][dy][0] = rok ? lf : 0.f;
            in[ci][dy][1] = rok ? mid.x : 0.f; in[ci][dy][2] = rok ? mid.y : 0.f;
            in[ci][dy][3] = rok ? mid.z : 0.f; in[ci][dy][4] = rok ? mid.w : 0.f;
            in[ci][dy][5] = rok ? rt : 0.f;
        }
    float* dst = y + b * Cout * HW + (int64_t)row * W + 4 * xq;
    for (int co0 = 0; co0 < Cout; co0 += 4) {
        f32x4 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float bv = bias != nullptr ? bias[co0 + k] : 0.f;
            acc[k] = f32x4{bv, bv, bv, bv};
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(&c3_fewin_w[((ci * 3 + dy) * 3 + dx) * Cout + co0]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc[k].x = fmaf(w4[k], in[ci][dy][dx + 0], acc[k].x);
                        acc[k].y = fmaf(w4[k], in[ci][dy][dx + 1], acc[k].y);
                        acc[k].z = fmaf(w4[k], in[ci][dy][dx + 2], acc[k].z);
                        acc[k].w = fmaf(w4[k], in[ci][dy][dx + 3], acc[k].w);
                    }
                }
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(dst + (int64_t)(co0 + k) * HW) = acc[k];
    }
}

// dst[r][c] = c < ws ? src[r][c] : 0 for rows of ws -> wd values: zero padding of the columns (wd > ws) or a crop (wd < ws)
__global__ void __launch_bounds__(256)
copy_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n, int ws, int wd) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t r = i / wd;
    const int c = (int)(i - r * wd);
    dst[i] = c < ws ? src[r * ws + c] : 0.f;
}

int c3_pick_rt(int M) {
    static const int forced = getenv("RST_CONV_RT") != nullptr ? atoi(getenv("RST_CONV_RT")) : 0;      // A/B runs
    if (forced >= 1 && forced <= 3) return forced;      // (measured: the highest block that pads least is the fastest here)
    int best = 1;
    int64_t best_pad = INT64_MAX;
    for (int rt = 3; rt >= 1; --rt) {          // 4 would need 171 KB of LDS
        const int64_t pad = cdiv(M, 32 * rt) * 32 * rt;
        if (pad < best_pad) { best_pad = pad; best = rt; }
    }
    return best;
}

}  // namespace

// tile: 128 pixels = TH rows x TW columns.  Images 32 or 64 pixels wide (with a height in whole tiles) get 4 or 2 rows per
// tile; every other width a multiple of 8 gets tiles of one row x 128 columns, the last tile of a row partly outside
// the image (its pixels converted to zeros, its stores masked): 40 ... 960-pixel levels of tiled windows and frames.
static int c3_tile_width(int H, int W) {
    if ((W == 32 && H % 4 == 0) || (W == 64 && H % 2 == 0)) return W;
    return 128;
}
int copy_cols_launch(const float* src, float* dst, int64_t rows, int ws, int wd, hipStream_t stream) {
    const int64_t n = rows * wd;
    RST_REQUIRE(src && dst && rows > 0 && ws > 0 && wd > 0 && cdiv(n, 256) < (1ll << 31), "copy_cols: bad shape");
    ProfScope prof(stream, "copy_cols", 0.0, 4.0 * rows * ((ws < wd ? ws : wd) + wd));
    hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, src, dst, n, ws, wd);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// the kernels take widths that are multiples of 8; any other width runs on a copy with zero columns added on the right
// (the zero padding of the convolution itself, so the result is the same) whose result is cropped
static bool c3_kernel_width(int W) { return W % 8 == 0; }

bool conv3x3_f16s_supported(int Cin, int Cout, int H, int W, int mode) {
    if (Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return false;
    if (mode == 1 && (H % 2 != 0 || W % 2 != 0)) return false;
    if (mode == 2 && Cout % 4 != 0) return false;
    return mode >= 0 && mode <= 2;
}

size_t conv3x3_f16s_ws_bytes(int Cin, int Cout) {
    return (size_t)round_up(Cout, 16) * 9 * (size_t)round_up(Cin, 32) * 4;
}

// workspace of a call: the split weights, and for a width that is not a multiple of 8 the padded input and result
// (the first part: what the kernels of the current state of the exact-fp32 switch need at a width they take)
static size_t c3_inner_ws_bytes(int B, int Cin, int Cout, int H, int Wk, int mode) {
    const size_t n = gemm_f16s_enabled() ? conv3x3_f16s_ws_bytes(Cin, Cout) : conv3x3_exact_ws_bytes(B, Cin, Cout, H, Wk, mode);
    return (size_t)round_up(n, 256);
}
size_t conv3x3_f16s_fwd_ws_bytes(int B, int Cin, int Cout, int H, int W, int mode) {
    const int Wp = (int)round_up(W, 8);
    size_t n = c3_inner_ws_bytes(B, Cin, Cout, H, Wp, mode);
    if (!c3_kernel_width(W))
        n += round_up((size_t)B * Cin * H * Wp * 4, (size_t)256) + round_up((size_t)B * Cout * H * Wp * 4, (size_t)256);
    return n;
}

int conv3x3_f16s_launch(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int H, int W,
                        int mode, void* ws, size_t ws_bytes, hipStream_t stream) {
    RST_REQUIRE(x && w && y && B > 0 && B <= 65535, "conv3x3: bad operand");
    RST_REQUIRE(conv3x3_f16s_supported(Cin, Cout, H, W, mode),
                "conv3x3: unsupported shape Cin=%d Cout=%d H=%d W=%d mode=%d (unshuffle: even H and W; shuffle: Cout %% 4 == 0)", Cin, Cout, H, W, mode);
    RST_REQUIRE(aligned16(x) && aligned16(y), "conv3x3: tensors must be 16-byte aligned");
    if (!c3_kernel_width(W)) {
        const size_t need = conv3x3_f16s_fwd_ws_bytes(B, Cin, Cout, H, W, mode);
        if (ws == nullptr || ws_bytes < need || !aligned16(ws))
            return fail(RST_ERR_WORKSPACE, "conv3x3: workspace %zu < %zu bytes (rst_conv3x3_fwd_workspace_bytes)", ws_bytes, need);
        const int Wp = (int)round_up(W, 8);
        const size_t wb = c3_inner_ws_bytes(B, Cin, Cout, H, Wp, mode);
        const size_t xb = round_up((size_t)B * Cin * H * Wp * 4, (size_t)256);
        float* xp = reinterpret_cast<float*>(static_cast<char*>(ws) + wb);
        float* yp = reinterpret_cast<float*>(static_cast<char*>(ws) + wb + xb);
        RST_TRY(copy_cols_launch(x, xp, (int64_t)B * Cin * H, W, Wp, stream));
        RST_TRY(conv3x3_f16s_launch(xp, w, bias, yp, B, Cin, Cout, H, Wp, mode, ws, wb, stream));
        // rows of the result: mode 0 [Cout][H] x W, mode 1 [4 Cout][H / 2] x W / 2, mode 2 [Cout / 4][2 H] x 2 W
        const int64_t rows = (int64_t)B * Cout * H * (mode == 1 ? 2 : 1) / (mode == 2 ? 2 : 1);
        const int f2 = mode == 1 ? 1 : mode == 2 ? 4 : 2;             // twice the width factor
        return copy_cols_launch(yp, y, rows, Wp * f2 / 2, W * f2 / 2, stream);
    }
    if (!gemm_f16s_enabled()) return conv3x3_exact_launch(x, w, bias, y, B, Cin, Cout, H, W, mode, ws, ws_bytes, stream);
    if (ws == nullptr || ws_bytes < conv3x3_f16s_ws_bytes(Cin, Cout) || !aligned16(ws))
        return fail(RST_ERR_WORKSPACE, "conv3x3: workspace %zu < %zu bytes", ws_bytes, conv3x3_f16s_ws_bytes(Cin, Cout));
    static const bool fewin_off = getenv("RST_CONV_FEWIN") != nullptr && getenv("RST_CONV_FEWIN")[0] == '0';   // A/B runs
    if (!fewin_off && mode == 0 && Cin <= 4 && Cout % 4 == 0 && 4 * 9 * Cout * sizeof(float) <= 64 * 1024) {
        const int64_t quads = (int64_t)B * H * (W / 4);
        const size_t lds = (size_t)Cin * 9 * Cout * sizeof(float);
        const double px = (double)B * H * W;
        ProfScope prof(stream, "conv3x3_fewin", 18.0 * Cin * Cout * px, 4.0 * px * (Cin + Cout));
        const dim3 grid((unsigned)cdiv(quads, 256));
        RST_REQUIRE(cdiv(quads, 256) < (1ll << 31), "conv3x3: grid too large");
        switch (Cin) {
        case 1: hipLaunchKernelGGL(conv3x3_fewin_kernel<1>, grid, dim3(256), lds, stream, x, w, bias, y, Cout, H, W, quads); break;
        case 2: hipLaunchKernelGGL(conv3x3_fewin_kernel<2>, grid, dim3(256), lds, stream, x, w, bias, y, Cout, H, W, quads); break;
        case 3: hipLaunchKernelGGL(conv3x3_fewin_kernel<3>, grid, dim3(256), lds, stream, x, w, bias, y, Cout, H, W, quads); break;
        default: hipLaunchKernelGGL(conv3x3_fewin_kernel<4>, grid, dim3(256), lds, stream, x, w, bias, y, Cout, H, W, quads); break;
        }
        RST_LAUNCH_CHECK();
        return RST_OK;
    }
    Conv3F a{};
    a.x = x; a.afrag = ws; a.bias = bias; a.y = y;
    a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.mode = mode;
    a.TW = c3_tile_width(H, W); a.TH = 128 / a.TW;
    a.kch = (int)cdiv(Cin, 32);
    a.mtiles = (int)cdiv(Cout, 16);
    const int rt = c3_pick_rt(Cout);
    a.mblocks = (int)cdiv(Cout, 32 * rt);
    a.tiles_x = (int)cdiv(W, a.TW); a.tiles_y = H / a.TH;
    hipLaunchKernelGGL(conv3x3_f16s_prep_kernel, dim3((unsigned)a.mtiles, (unsigned)(9 * a.kch)), dim3(64), 0, stream, w, Cout, Cin,
                       a.kch, static_cast<c3_h8*>(ws));
    RST_LAUNCH_CHECK();
    const int64_t gx = (int64_t)a.tiles_x * a.tiles_y * a.mblocks;
    a.walk = xcd_walk_enabled();
    RST_REQUIRE(gx < (1ll << 31), "conv3x3: grid too large");
    const size_t lds = sizeof(float) * (size_t)(2 * (C3_XS + 3 * 2 * rt * 2 * 256) + 4 * C3_IMGH);
    const void* kern = rt == 1 ? reinterpret_cast<const void*>(&conv3x3_f16s_kernel<1>)
                     : rt == 2 ? reinterpret_cast<const void*>(&conv3x3_f16s_kernel<2>)
                               : reinterpret_cast<const void*>(&conv3x3_f16s_kernel<3>);
    KernelSetup su;
    RST_TRY(kernel_setup(kern, 512, lds, &su));
    const double px = (double)B * H * W;
    ProfScope prof(stream, "conv3x3_f16s", 18.0 * Cin * Cout * px, 4.0 * px * (Cin + Cout));
    void* args[] = {(void*)&a};
    RST_HIP(hipLaunchKernel(kern, dim3((unsigned)gx, (unsigned)B), dim3(512), args, lds, stream));
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst

extern "C" {
size_t rst_conv3x3_workspace_bytes(int32_t Cin, int32_t Cout) {
    return (Cin > 0 && Cout > 0) ? rst::conv3x3_f16s_ws_bytes(Cin, Cout) : 0;
}
size_t rst_conv3x3_fwd_workspace_bytes(int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode) {
    if (B <= 0 || !rst::conv3x3_f16s_supported(Cin, Cout, H, W, shuffle_mode)) return 0;
    return rst::conv3x3_f16s_fwd_ws_bytes(B, Cin, Cout, H, W, shuffle_mode);
}
int32_t rst_conv3x3_supported(int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode) {
    return rst::conv3x3_f16s_supported(Cin, Cout, H, W, shuffle_mode) ? 1 : 0;
}
int rst_conv3x3_fwd(const float* x, const float* weight, const float* bias, float* y, int32_t B, int32_t Cin, int32_t Cout,
                    int32_t H, int32_t W, int32_t shuffle_mode, void* workspace, size_t workspace_bytes, void* stream) {
    return rst::conv3x3_f16s_launch(x, weight, bias, y, B, Cin, Cout, H, W, shuffle_mode, workspace, workspace_bytes,
                                    static_cast<hipStream_t>(stream));
}
}
