// Dense 3x3 convolution, zero padding 1, on channel-packed bf16 tensors ([B][Cb][H*W][8], bf16_pk.hip), with
// PixelUnshuffle(2) / PixelShuffle(2) as its epilogue: OverlapPatchEmbed, Downsample, Upsample of the reference
// (model/Restromer.py:172-215) for the bf16 storage path, one kernel each, packed in and packed out.
//
// Implicit GEMM on v_mfma_f32_16x16x32_bf16.  A 16-byte chunk of the packed layout (8 channels of one pixel) IS the
// B operand of a lane, so the input needs no conversion at all: per (dy, 32-channel chunk) the chunks of the tile's
// pixels and of the column either side go by LDS-DMA straight into an operand image [pixel slot][lane group] x 16 B,
// one chunk per DMA lane (source addresses clamped into the image; what lies outside is zeroed when read), and the
// three horizontal taps are three reads of the image at slots shifted by one.  The weights are turned once per call
// into bf16 operand fragments ordered [16-row tile][dy][chunk][dx]; the rows of a tile pair are permuted as in
// gemm_pk so that a lane's 4 + 4 accumulator rows are 8 consecutive channels: mode 0 stores its result as packed
// chunks straight from the registers.  The shuffles regroup through a wave-private LDS scratch:
//   mode 2, PixelShuffle (out[m / 4][2 y + (m % 4) / 2][2 x + m % 2]): the four lane groups of a pixel hold 32 conv
//     channels = 8 output channels x the four sub-positions; lane group q writes sub-position q;
//   mode 1, PixelUnshuffle (out[4 m + 2 (y & 1) + (x & 1)][y / 2][x / 2]): a wave's two pixel tiles are the two rows of
//     a 16-column strip, so a 2 x 2 quad sits in two neighbouring lanes x two tiles; a chunk = 2 conv channels x quad.
// A block: 8 waves = 2 row halves x 4 pixel quarters, 32 RT output channels (RT = 2, 4 or 6 tiles per half) x 128
// pixels (TH rows x TW columns; partial tiles at the right and bottom edges are masked), 2-stage ring, one barrier per
// slab.
#include "kernels.hpp"
#include "bf16_common.hpp"

namespace rst {

namespace {

typedef __attribute__((address_space(1))) const void* cp_gptr;
typedef __attribute__((address_space(3))) void* cp_lptr;

constexpr int CP_CHUNKS = 9 * 64;              // chunk slots of an operand image (544 used at most: 136 pixel slots x 4 groups)
constexpr int CP_IMG = CP_CHUNKS * 16;         // bytes

struct Conv3Pk {
    const bf16_t* x; const void* afrag; const float* bias; bf16_t* y;
    int64_t x_bs, y_bs;                        // elements per sample
    int Cinb, Cout, H, W, TW, TH, kch, mblocks, mtiles, tiles_x, tiles_y, mode, walk;
};

// frag[mt][q = (dyi * kch + kc) * 3 + dxi][lane] x 16 B: row = permuted position of (mt, lane & 15), k = 32 kc + 8 (lane >> 4) + j
__global__ void __launch_bounds__(64)
conv3x3_pk_prep_kernel(const float* __restrict__ w, int Cout, int Cin, int kch, u32x4* __restrict__ frag, int transposed) {
    const int lane = threadIdx.x, rr = lane & 15, g = lane >> 4;
    const int mt = blockIdx.x, q = blockIdx.y;
    const int dxi = q % 3, kc = (q / 3) % kch, dyi = q / (3 * kch);
    const int m = 32 * (mt >> 1) + 8 * (rr >> 2) + 4 * (mt & 1) + (rr & 3);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = 32 * kc + 8 * g + j;
        // transposed: the convolution that gives the DATA gradient of y = conv(x, w), w stored [Cin][Cout][3][3] as seen
        // from here: rows and channels swapped, taps rotated by 180 degrees
        v[j] = !(m < Cout && ci < Cin) ? 0.f
             : transposed ? w[(((int64_t)ci * Cout + m) * 3 + (2 - dyi)) * 3 + (2 - dxi)]
                          : w[(((int64_t)m * Cin + ci) * 3 + dyi) * 3 + dxi];
    }
    frag[((int64_t)mt * gridDim.y + q) * 64 + lane] = pack8(v);
}

template <int RT>
__global__ void __launch_bounds__(512)
conv3x3_pk_kernel(Conv3Pk a) {
    constexpr int AS = 3 * 2 * RT * 1024;                  // bytes of weight fragments per slab: 3 taps x 2 RT tiles x 1 KiB
    constexpr int STAGE = CP_IMG + AS;
    constexpr int PW = (6 * RT + 7) / 8;
    extern __shared__ __attribute__((aligned(16))) char cp_ring[];

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
    const int H = a.H, W = a.W, TW = a.TW, TH = a.TH;
    const int64_t HW = (int64_t)H * W;
    const u32x4* Xb = reinterpret_cast<const u32x4*>(a.x + (int64_t)b * a.x_bs);
    const char* Af = static_cast<const char*>(a.afrag);
    const int NS = 3 * a.kch, NQ = 9 * a.kch;
    const int SW = TW + 2;                                  // slots per tile row

    // ---- DMA: lane `lane` of instruction k brings chunk 64 k + lane = (slot, group) of the image; instructions are
    // dealt to the waves round robin (k = wave, wave + 8) ----
    int dty[2], dx_[2], dg[2];
    bool dlive[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = wave + 8 * i;
        const int q = 64 * k + lane, slot = q >> 2;
        dlive[i] = k < 9;
        dg[i] = q & 3;
        const int ty = slot / SW, txp = slot % SW;
        dty[i] = ty < TH ? ty : 0;                          // chunk slots past the tile: any valid address
        int xx = x0 + txp - 1;
        dx_[i] = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
    }
    auto issue = [&](int s) __attribute__((always_inline)) {
        char* st = cp_ring + (s & 1) * STAGE;
        const int dy = s / a.kch - 1, kc = s % a.kch;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!dlive[i]) continue;                        // wave-uniform (k depends on the wave only)
            int cb = 4 * kc + dg[i];
            if (cb > a.Cinb - 1) cb = a.Cinb - 1;           // channel blocks past the input meet zero weights
            int yy = y0 + dty[i] + dy;
            yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
            const u32x4* src = Xb + (int64_t)cb * HW + (int64_t)yy * W + dx_[i];
            __builtin_amdgcn_global_load_lds((cp_gptr)src, (cp_lptr)(st + (wave + 8 * i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int j = wave + 8 * i;
            if (j >= 6 * RT) break;
            const int t = j % (2 * RT), dxi = j / (2 * RT);
            int mt = m0 / 16 + t;
            if (mt > a.mtiles - 1) mt = a.mtiles - 1;
            const char* src = Af + (((int64_t)mt * NQ + (int64_t)s * 3 + dxi) * 64 + lane) * 16;
            __builtin_amdgcn_global_load_lds((cp_gptr)src, (cp_lptr)(st + CP_IMG + (dxi * 2 * RT + t) * 1024), 16, 0, 0);
        }
    };

    // ---- this lane's two pixels.  Modes 0 and 2: 32 consecutive pixels of the tile per wave; mode 1: a strip of 16
    // columns on two rows (tile t = row parity) ----
    int pty[2], ptx[2], mslot[2];
    bool vx[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (a.mode == 1) {
            const int wcols = TW / 16;
            pty[t] = 2 * (nq / wcols) + t; ptx[t] = 16 * (nq % wcols) + lm;
        } else {
            const int n = 32 * nq + 16 * t + lm;
            pty[t] = n / TW; ptx[t] = n % TW;
        }
        mslot[t] = pty[t] * SW + ptx[t] + 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int xx = x0 + ptx[t] + d - 1;
            vx[t][d] = xx >= 0 && xx < W;
        }
    }

    f32x4 acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int mt = m0 / 16 + mh * RT + i;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 32 * (mt >> 1) + 8 * kk + 4 * (mt & 1) + r;
                b4[r] = a.bias[m < a.Cout ? m : a.Cout - 1];
            }
        }
        acc[i][0] = b4; acc[i][1] = b4;
    }

    // ---- one barrier per slab: [slab s landed, everyone done with slab s - 1] -> issue slab s + 1 -> multiply slab s ----
    issue(0);
    for (int s = 0; s < NS; ++s) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s + 1 < NS) issue(s + 1);
        const char* st = cp_ring + (s & 1) * STAGE;
        const int dy = s / a.kch - 1;
        bool vy[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { const int yy = y0 + pty[t] + dy; vy[t] = yy >= 0 && yy < H; }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            bf16x8 xb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 v = reinterpret_cast<const u32x4*>(st)[(mslot[t] + d - 1) * 4 + kk];
                const bool ok = vx[t][d] && vy[t];
                v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
                xb[t] = __builtin_bit_cast(bf16x8, v);
            }
            const bf16x8* af = reinterpret_cast<const bf16x8*>(st + CP_IMG) + (d * 2 * RT + mh * RT) * 64 + lane;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const bf16x8 aw = af[i * 64];
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw, xb[0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw, xb[1], acc[i][1], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: tile pairs (2 j, 2 j + 1) of this wave give 8 consecutive conv channels per lane ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the ring is free: scratch
    float* scr = reinterpret_cast<float*>(cp_ring) + wave * 1024;
    u32x4* Yb = reinterpret_cast<u32x4*>(a.y + (int64_t)b * a.y_bs);
    bool pv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) pv[t] = x0 + ptx[t] < W && y0 + pty[t] < H && pty[t] < TH;
#pragma unroll
    for (int j = 0; j < RT / 2; ++j) {
        const int J = (m0 / 16 + mh * RT) / 2 + j;          // pair index: conv channels 32 J + 8 kk + (0..7) in this lane
        const int cbo = 4 * J + kk;
        if (a.mode == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[2 * j][t][r]; v[4 + r] = acc[2 * j + 1][t][r]; }
                if (pv[t] && cbo * 8 < a.Cout) Yb[(int64_t)cbo * HW + (int64_t)(y0 + pty[t]) * W + x0 + ptx[t]] = pack8(v);
            }
        } else if (a.mode == 2) {
            // scr[lm][kk][8]; lane group q collects sub-position q: out channel 8 J + 2 kk' + h <- conv channel 8 (4 J + kk') + 4 h + q
            const int64_t OHW = 4 * HW;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float* row = scr + (lm * 4 + kk) * 8;
                *reinterpret_cast<f32x4*>(row) = acc[2 * j][t];
                *reinterpret_cast<f32x4*>(row + 4) = acc[2 * j + 1][t];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                float o[8];
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    o[2 * k2] = scr[(lm * 4 + k2) * 8 + kk];
                    o[2 * k2 + 1] = scr[(lm * 4 + k2) * 8 + 4 + kk];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int oy = 2 * (y0 + pty[t]) + (kk >> 1), ox = 2 * (x0 + ptx[t]) + (kk & 1);
                if (pv[t] && J * 32 < a.Cout) Yb[(int64_t)J * OHW + (int64_t)oy * 2 * W + ox] = pack8(o);
            }
        } else {
            // scr[kk][lm][t][8]; lane (lm, kk) writes the chunks i = 2 (lm & 1), 2 (lm & 1) + 1 of out pixel u = lm >> 1:
            // out block 4 cbo + i = conv channels 8 cbo + 2 i + (0, 1) x quad position 2 t + (column parity)
            const int64_t QHW = HW / 4;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float* row = scr + ((kk * 16 + lm) * 2 + t) * 8;
                *reinterpret_cast<f32x4*>(row) = acc[2 * j][t];
                *reinterpret_cast<f32x4*>(row + 4) = acc[2 * j + 1][t];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int u = lm >> 1;
            u32x4 out2[2];
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * (lm & 1) + ii;
                float o[8];
#pragma unroll
                for (int ee = 0; ee < 2; ++ee)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int px = 0; px < 2; ++px)
                            o[4 * ee + 2 * t + px] = scr[((kk * 16 + 2 * u + px) * 2 + t) * 8 + 2 * i + ee];
                out2[ii] = pack8(o);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (pty[0], ptx) of the even lane of the pair: the quad's top-left pixel
            const int qy = y0 + pty[0], qx = x0 + (ptx[0] & ~1);
            const bool qok = qx < W && qy < H && pty[0] < TH;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * (lm & 1) + ii;
                if (qok && 8 * cbo + 2 * i < a.Cout)
                    Yb[(int64_t)(4 * cbo + i) * QHW + (int64_t)(qy >> 1) * (W / 2) + (qx >> 1)] = out2[ii];
            }
        }
    }
}

int cp_pick_rt(int M) {
    static const int forced = getenv("RST_CONV_RT") != nullptr ? atoi(getenv("RST_CONV_RT")) : 0;      // A/B runs
    if (forced == 2 || forced == 4 || forced == 6) return forced;
    // least padding; among equals the lowest block (64 rows): two blocks share a CU's LDS and registers and one
    // multiplies while the other waits for its slab (1.36 against 1.43 ms per inference step for the network's seven
    // convolutions; the tiles of x a second block of rows reads again come from the L2)
    int best = 2;
    int64_t best_pad = INT64_MAX;
    for (int rt = 2; rt <= 6; rt += 2) {
        const int64_t pad = cdiv(M, 32 * rt) * 32 * rt;
        if (pad < best_pad) { best_pad = pad; best = rt; }
    }
    return best;
}

}  // namespace

bool conv3x3_pk_supported(int Cin, int Cout, int H, int W, int mode) {
    if (Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 2) return false;
    if (mode == 1 && (H % 2 != 0 || W % 2 != 0 || Cout % 2 != 0)) return false;
    if (mode == 2 && Cout % 32 != 0) return false;
    if (mode == 0 && Cout % 8 != 0) return false;
    return true;
}

// operand fragments of the weights: whole blocks of 32 RT rows, 9 taps, 32-channel chunks, 1 KiB each
size_t conv3x3_pk_ws_bytes(int Cin, int Cout) {
    const int rt = cp_pick_rt(Cout);
    return (size_t)cdiv(Cout, 32 * rt) * 2 * rt * 9 * (size_t)cdiv(Cin, 32) * 1024;
}

int conv3x3_pk_launch(const bf16_t* x, const float* w, const float* bias, bf16_t* y, int B, int Cin, int Cout, int H, int W,
                      int mode, void* ws, size_t ws_bytes, hipStream_t stream, int transposed) {
    RST_REQUIRE(x && w && y && B > 0 && B <= 65535, "bf16 conv3x3: bad operand");
    RST_REQUIRE(conv3x3_pk_supported(Cin, Cout, H, W, mode),
                "bf16 conv3x3: unsupported shape Cin=%d Cout=%d H=%d W=%d mode=%d (unshuffle: even H, W, Cout; shuffle: Cout %% 32 == 0; plain: Cout %% 8 == 0)",
                Cin, Cout, H, W, mode);
    RST_REQUIRE(aligned16(x) && aligned16(y), "bf16 conv3x3: tensors must be 16-byte aligned");
    if (ws == nullptr || ws_bytes < conv3x3_pk_ws_bytes(Cin, Cout) || !aligned16(ws))
        return fail(RST_ERR_WORKSPACE, "bf16 conv3x3: workspace %zu < %zu bytes", ws_bytes, conv3x3_pk_ws_bytes(Cin, Cout));
    Conv3Pk a{};
    a.x = x; a.afrag = ws; a.bias = bias; a.y = y;
    a.Cinb = (int)cdiv(Cin, 8); a.Cout = Cout; a.H = H; a.W = W; a.mode = mode;
    const int64_t HW = (int64_t)H * W;
    a.x_bs = (int64_t)a.Cinb * HW * 8;
    // elements per sample of the result: Cout (or 4 Cout / Cout / 4) channels in whole blocks of 8
    const int64_t outb = mode == 1 ? cdiv(4 * (int64_t)Cout, 8) : mode == 2 ? Cout / 32 : cdiv(Cout, 8);
    a.y_bs = outb * (mode == 1 ? HW / 4 : mode == 2 ? 4 * HW : HW) * 8;
    const int tw_max = mode == 1 ? 64 : 128;
    int tw = 16;
    while (tw * 2 <= tw_max && tw * 2 <= W) tw *= 2;          // widest power-of-two tile row the image holds
    a.TW = tw; a.TH = 128 / tw;
    a.kch = (int)cdiv(Cin, 32);
    const int rt = cp_pick_rt(Cout);
    a.mblocks = (int)cdiv(Cout, 32 * rt);
    a.mtiles = a.mblocks * 2 * rt;                            // whole blocks: the fragment buffer covers the padding rows
    a.tiles_x = (int)cdiv(W, a.TW); a.tiles_y = (int)cdiv(H, a.TH);
    hipLaunchKernelGGL(conv3x3_pk_prep_kernel, dim3((unsigned)a.mtiles, (unsigned)(9 * a.kch)), dim3(64), 0, stream, w, Cout, Cin,
                       a.kch, static_cast<u32x4*>(ws), transposed);
    RST_LAUNCH_CHECK();
    const int64_t gx = (int64_t)a.tiles_x * a.tiles_y * a.mblocks;
    a.walk = xcd_walk_enabled();
    RST_REQUIRE(gx < (1ll << 31), "bf16 conv3x3: grid too large");
    const size_t lds = (size_t)2 * (CP_IMG + 3 * 2 * rt * 1024);
    const void* kern = rt == 2 ? reinterpret_cast<const void*>(&conv3x3_pk_kernel<2>)
                     : rt == 4 ? reinterpret_cast<const void*>(&conv3x3_pk_kernel<4>)
                               : reinterpret_cast<const void*>(&conv3x3_pk_kernel<6>);
    KernelSetup su;
    RST_TRY(kernel_setup(kern, 512, lds, &su));
    const double px = (double)B * H * W;
    ProfScope prof(stream, "pk_conv3x3", 18.0 * Cin * Cout * px, 2.0 * px * (a.Cinb * 8 + Cout));
    void* args[] = {(void*)&a};
    RST_HIP(hipLaunchKernel(kern, dim3((unsigned)gx, (unsigned)B), dim3(512), args, lds, stream));
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst

extern "C" {
size_t rst_bf16_conv3x3_workspace_bytes(int32_t Cin, int32_t Cout) {
    return (Cin > 0 && Cout > 0) ? rst::conv3x3_pk_ws_bytes(Cin, Cout) : 0;
}
int32_t rst_bf16_conv3x3_supported(int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode) {
    return rst::conv3x3_pk_supported(Cin, Cout, H, W, shuffle_mode) ? 1 : 0;
}
int rst_bf16_conv3x3_fwd(const void* x_packed, const float* weight, const float* bias, void* y_packed, int32_t B, int32_t Cin,
                         int32_t Cout, int32_t H, int32_t W, int32_t shuffle_mode, void* workspace, size_t workspace_bytes,
                         void* stream) {
    return rst::conv3x3_pk_launch(static_cast<const rst::bf16_t*>(x_packed), weight, bias, static_cast<rst::bf16_t*>(y_packed), B, Cin,
                                  Cout, H, W, shuffle_mode, workspace, workspace_bytes, static_cast<hipStream_t>(stream), 0);
}
}
