// Y[b] = A[b] * X[b] (+bias) (+rscale o R[b]) -- general tiled variant for any K:
// the 1x1 convolutions of the low-resolution levels (K = 192..768) and every
// shape the resident-weight kernel (gemm_nn_res.hip) does not take.
//
// 8 waves per block (2 row halves x 4 pixel quarters), block tile
// (32*MTW rows) x 128 pixels, K walked in chunks of 16 through a 2-deep LDS ring
// that is filled by LDS-DMA only (global_load_lds, 16 B per lane: no staging
// registers, no address arithmetic in the VALU beyond one pointer per piece):
//   A chunk  [rows][16 k]   k contiguous: one 1-KiB piece = 16 rows; an MFMA lane
//            (m, kk) takes its four k-steps with a single ds_read_b128 -- the
//            k order inside a chunk is k = 4*kk + s, identically for both operands
//   X chunk  [16 k][128 px] one piece = 2 k-rows
// Each wave issues exactly two pieces per chunk, so the wait for chunk c is a
// counted `s_waitcnt vmcnt(2 (GD_STAGES - 2))` (the chunks behind c stay in
// flight), then one barrier, then the refill of the slot freed by chunk c-1.
// With TRANSA the A chunk is stored [16 k][rows] like X (memory is K-major).
// Exact fp32 (v_mfma_f32_16x16x4_f32).
#include "kernels.hpp"

namespace rst {

constexpr int GD_BN = 128;
constexpr int GD_BK = 16;
// Ring depth.  Two stages = 36 KB of LDS (the epilogue's transpose buffers are the larger part) and 62 registers: FOUR blocks (32 waves) per compute unit, each with one chunk
// in flight.  Measured on the residual products of a 16 x 3 x 256 x 256 forward pass (48 launches, same box): 2 stages
// 6.83 ms, 3 stages (three blocks per unit) 7.03, 4 stages 8.09, 5 stages 7.87; the residual requested ahead of the
// pixel stream 7.31 -- residency, not the depth of a block's own pipeline, is what keeps the memory system busy.
#ifndef GD_STG
#define GD_STG 2
#endif
constexpr int GD_STAGES = GD_STG;
constexpr int GD_SCR_LD = 36;

typedef __attribute__((address_space(1))) const void* gd_gptr;
typedef __attribute__((address_space(3))) void* gd_lptr;
typedef float gd_f32x2 __attribute__((ext_vector_type(2)));

template <int MTW, bool TRANSA, bool RES, bool LN>
__global__ void __launch_bounds__(512, 4)
gemm_nn_dma_kernel(GemmNN g, int mblocks) {
    constexpr int BM = 32 * MTW;
    constexpr int A_FLOATS = 128 * GD_BK;               // always 8 pieces (rows clamped), see header
    constexpr int X_FLOATS = GD_BK * GD_BN;
    constexpr int STAGE = A_FLOATS + X_FLOATS;
    // (the epilogue reuses the ring as eight wave-private transpose buffers of 2 x 16 x GD_SCR_LD floats: 36 KB, four blocks per unit still fit)
    constexpr int SCR_FLOATS = 8 * 2 * 16 * GD_SCR_LD;
    __shared__ __attribute__((aligned(16))) float ring[GD_STAGES * STAGE > SCR_FLOATS ? GD_STAGES * STAGE : SCR_FLOATS];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, kk = lane >> 4;
    const int nq = wave & 3, mh = wave >> 2;
    const int mb = blockIdx.x % mblocks;
    const int64_t nt = blockIdx.x / mblocks;
    const int b1 = blockIdx.y / g.nb2, b2 = blockIdx.y % g.nb2;
    const int m0 = mb * BM;
    const int64_t n0 = nt * GD_BN;
    const float* Ab = g.A + (int64_t)b1 * g.a_s1 + (int64_t)b2 * g.a_s2;
    const float* Xb = g.X + (int64_t)b1 * g.x_s1 + (int64_t)b2 * g.x_s2 + n0;
    const int nchunks = (g.K + GD_BK - 1) / GD_BK;

    // ---- this wave's two DMA pieces per chunk ----
    // X piece `wave`: k-rows 2*wave, 2*wave+1; lane -> row 2*wave + (lane>>5), 4 px at (lane&31)*4
    const int xrow = 2 * wave + (lane >> 5), xcol = (lane & 31) * 4;
    const bool xok = n0 + xcol < g.N;
    // A piece `wave`: non-transposed: rows 16*wave .. +15, lane -> row 16*wave + (lane>>2), k quad (lane&3)*4
    //                 transposed    : k-rows 2*wave, 2*wave+1 of [16][128 rows], lane -> 4 rows at (lane&31)*4
    const int arow = TRANSA ? (lane & 31) * 4 : 16 * wave + (lane >> 2);
    const int ak = TRANSA ? 2 * wave + (lane >> 5) : (lane & 3) * 4;
    const int arow_c = min(m0 + arow, TRANSA ? ((g.M - 4) > 0 ? (g.M - 4) : 0) : g.M - 1);   // clamped (duplicates are never stored)

    auto issue = [&](int c) {
        float* st = ring + (c % GD_STAGES) * STAGE;
        const int k0 = c * GD_BK;
        {   // A
            int k = k0 + ak;
            if (k > g.K - (TRANSA ? 1 : 4)) k = g.K - (TRANSA ? 1 : 4);              // clamped; masked in the MFMA loop
            const float* src = TRANSA ? Ab + (int64_t)k * g.lda + arow_c : Ab + (int64_t)arow_c * g.lda + k;
            __builtin_amdgcn_global_load_lds((gd_gptr)src, (gd_lptr)(st + wave * 256), 16, 0, 0);
        }
        {   // X
            int k = k0 + xrow;
            if (k > g.K - 1) k = g.K - 1;
            const float* src = Xb + (int64_t)k * g.ldx + (xok ? xcol : 0);
            __builtin_amdgcn_global_load_lds((gd_gptr)src, (gd_lptr)(st + A_FLOATS + wave * 256), 16, 0, 0);
        }
    };

    f32x4 acc[MTW][2];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (g.bias != nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + mh * 16 * MTW + 16 * i + 4 * kk + r;
                b4[r] = g.bias[m < g.M ? m : g.M - 1];
            }
        }
        acc[i][0] = b4;
        acc[i][1] = b4;
    }

    // folded LayerNorm: sum and sum of squares over the channels of this wave's two pixel columns
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};     // plain (not packed) ALU ops: packed fp32 beside MFMAs costs extra issue time

    // per-row constants of the folded LayerNorm, fetched now and used in the epilogue
    constexpr int LNN = LN ? MTW : 1;
    f32x4 lns[LNN], lnc[LNN];
    if constexpr (LN) {
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + mh * 16 * MTW + 16 * i + 4 * kk + r;
                const int mc = m < g.M ? m : g.M - 1;
                lns[i][r] = g.ln_s[mc];
                lnc[i][r] = g.ln_c[mc];
            }
    }

    // prologue: GD_STAGES - 1 chunks in flight (re-issuing the last chunk keeps the per-wave count fixed)
#pragma unroll
    for (int c = 0; c < GD_STAGES - 1; ++c) issue(c < nchunks ? c : nchunks - 1);

    for (int c = 0; c < nchunks; ++c) {
        // chunk c landed for this wave (the chunks behind it may stay in flight), then for everyone;
        // the same barrier says every wave is done reading chunk c-1, whose slot is refilled next
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(2 * (GD_STAGES - 2)) : "memory");
        issue(c + GD_STAGES - 1 < nchunks ? c + GD_STAGES - 1 : nchunks - 1);          // always two pieces per step per wave
        const float* st = ring + (c % GD_STAGES) * STAGE;
        const bool kvalid = c * GD_BK + 4 * kk < g.K;          // K % 4 == 0: a lane's four k are all valid or all past K
        f32x4 a4[MTW];
        float bv[4][2];
        if constexpr (!TRANSA) {
#pragma unroll
            for (int i = 0; i < MTW; ++i)
                a4[i] = *reinterpret_cast<const f32x4*>(&st[(mh * 16 * MTW + 16 * i + lm) * GD_BK + 4 * kk]);
        } else {
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) a4[i][s] = st[(4 * kk + s) * 128 + mh * 16 * MTW + 16 * i + lm];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bv[s][0] = st[A_FLOATS + (4 * kk + s) * GD_BN + nq * 32 + lm];
            bv[s][1] = st[A_FLOATS + (4 * kk + s) * GD_BN + nq * 32 + 16 + lm];
        }
        if (!kvalid) {
#pragma unroll
            for (int i = 0; i < MTW; ++i) a4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) { bv[s][0] = 0.f; bv[s][1] = 0.f; }
        }
        if constexpr (LN) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {      // rows past K were zeroed above
                s1[0] += bv[s][0]; s2[0] = fmaf(bv[s][0], bv[s][0], s2[0]);
                s1[1] += bv[s][1]; s2[1] = fmaf(bv[s][1], bv[s][1], s2[1]);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i][s], bv[s][0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i][s], bv[s][1], acc[i][1], 0, 0, 0);
            }
    }

    // ---- epilogue: per 16 x 32 tile through a wave-private LDS transpose, float4 row stores ----
    float* Yb = g.Y + (int64_t)b1 * g.y_s1 + (int64_t)b2 * g.y_s2;
    const float* Rb = RES ? g.R + (int64_t)b1 * g.r_s1 + (int64_t)b2 * g.r_s2 : nullptr;
    const float* RSb = (RES && g.rscale) ? g.rscale + (int64_t)b1 * g.rs_s1 + (int64_t)b2 * g.rs_s2 : nullptr;
    // the ring is free once every DMA has landed and every wave has left the loop: reuse it
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    float* scr = ring + wave * 2 * 16 * GD_SCR_LD;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
    const int64_t n = n0 + nq * 32 + c4;
    const bool nok = n < g.N;
    float rstd[2] = {1.f, 1.f}, tco[2] = {0.f, 0.f};
    if constexpr (LN) {
        const float invk = 1.0f / (float)g.K;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float a = s1[j], q = s2[j];
            a += __shfl_xor(a, 16); q += __shfl_xor(q, 16);
            a += __shfl_xor(a, 32); q += __shfl_xor(q, 32);
            const float mu = a * invk;
            const float var = fmaxf(q * invk - mu * mu, 0.f);
            rstd[j] = 1.0f / sqrtf(var + 1e-5f);
            tco[j] = g.ln_mode == 1 ? -mu : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        float* sl = scr + (i & 1) * 16 * GD_SCR_LD;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v0 = acc[i][0][r], v1 = acc[i][1][r];
            if constexpr (LN) {
                v0 = fmaf(rstd[0], fmaf(tco[0], lns[i][r], v0), lnc[i][r]);
                v1 = fmaf(rstd[1], fmaf(tco[1], lns[i][r], v1), lnc[i][r]);
            }
            sl[(4 * kk + r) * GD_SCR_LD + lm] = v0;
            sl[(4 * kk + r) * GD_SCR_LD + 16 + lm] = v1;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = rr + 8 * h;
            f32x4 v = *reinterpret_cast<const f32x4*>(&sl[row * GD_SCR_LD + c4]);
            const int m = m0 + mh * 16 * MTW + 16 * i + row;
            if (m < g.M && nok) {
                if constexpr (RES) {
                    const float rs = RSb ? RSb[m] : 1.f;
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(Rb + (int64_t)m * g.ldr + n);
                    v.x = fmaf(rs, rv.x, v.x); v.y = fmaf(rs, rv.y, v.y); v.z = fmaf(rs, rv.z, v.z); v.w = fmaf(rs, rv.w, v.w);
                }
                float* dst = Yb + (int64_t)m * g.ldy + n;
                if (g.accumulate) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
    }
}

// Rows per block = 32*MTW.  Per 128-pixel tile the matrix cores need ~pad*K*2/25 byte-times
// (25 flop/byte is the fp32-MFMA : HBM ridge) and memory moves (blocks*K + M)*4 bytes (every
// M-block re-reads the pixel tile): take the candidate with the smaller maximum of the two.
static int dma_pick_mtw(int M, int K) {
    int best = 1;
    double best_cost = 1e300;
    for (int mt = 4; mt >= 1; --mt) {
        const double blocks = (double)cdiv(M, 32 * mt), pad = blocks * 32 * mt;
        const double mfma = pad * K * 2.0 / 25.0, mem = (blocks * K + M) * 4.0;
        const double cost = (mfma > mem ? mfma : mem) + 0.05 * (mfma < mem ? mfma : mem);
        if (cost < best_cost) { best_cost = cost; best = mt; }
    }
    return best;
}

static bool dma_shape_ok(const GemmNN& g) {
    const bool al = aligned16(g.X) && aligned16(g.Y) && aligned16(g.A) && (g.R == nullptr || aligned16(g.R));
    const bool strides4 = (g.N % 4 == 0) && (g.ldx % 4 == 0) && (g.x_s1 % 4 == 0) && (g.x_s2 % 4 == 0) &&
                          (g.ldy % 4 == 0) && (g.y_s1 % 4 == 0) && (g.y_s2 % 4 == 0) &&
                          (g.lda % 4 == 0) && (g.a_s1 % 4 == 0) && (g.a_s2 % 4 == 0) &&
                          (g.R == nullptr || ((g.ldr % 4 == 0) && (g.r_s1 % 4 == 0) && (g.r_s2 % 4 == 0)));
    if (!al || !strides4 || g.K % 4 != 0 || g.K < 4) return false;
    if (g.transA && (g.M % 4 != 0 || g.M < 4)) return false;
    if ((int64_t)g.nb1 * g.nb2 > 65535) return false;
    return true;
}

bool gemm_nn_ln_supported(const GemmNN& g) {
    return dma_shape_ok(g) && !g.transA && g.R == nullptr && !g.accumulate && g.bias == nullptr &&
           g.a_s1 == 0 && g.a_s2 == 0;
}

int gemm_nn_dma_try(const GemmNN& g, hipStream_t stream, bool* launched) {
    *launched = false;
    if (!dma_shape_ok(g)) return RST_OK;
    if (g.ln_mode != 0 && !gemm_nn_ln_supported(g)) return fail(RST_ERR_INVALID, "gemm_nn: folded LayerNorm on an unsupported shape");
    const int mtw = dma_pick_mtw(g.M, g.K);
    const int mblocks = (int)cdiv(g.M, 32 * mtw);
    const int64_t ntiles = cdiv(g.N, GD_BN);
    const int64_t gx = ntiles * mblocks;
    if (gx >= (1ll << 31)) return RST_OK;
    dim3 grid((unsigned)gx, (unsigned)(g.nb1 * g.nb2)), blk(512);
    const double nb = (double)g.nb1 * g.nb2;
    ProfScope prof(stream, g.ln_mode ? "gemm_nn_ln" : "gemm_nn", 2.0 * g.M * g.K * g.N * nb,
                   4.0 * nb * g.N * (g.K + g.M * (1.0 + (g.R ? 1.0 : 0.0) + (g.accumulate ? 1.0 : 0.0))));
#define GD_LAUNCH(MT, TA, RS) hipLaunchKernelGGL((gemm_nn_dma_kernel<MT, TA, RS, false>), grid, blk, 0, stream, g, mblocks)
#define GD_CASE(MT)                                                        \
    case MT:                                                               \
        if (g.ln_mode != 0) hipLaunchKernelGGL((gemm_nn_dma_kernel<MT, false, false, true>), grid, blk, 0, stream, g, mblocks); \
        else if (g.transA) { if (g.R) GD_LAUNCH(MT, true, true); else GD_LAUNCH(MT, true, false); }   \
        else          { if (g.R) GD_LAUNCH(MT, false, true); else GD_LAUNCH(MT, false, false); } \
        break;
    switch (mtw) {
        GD_CASE(1) GD_CASE(2) GD_CASE(3) GD_CASE(4)
        default: return RST_OK;
    }
#undef GD_CASE
#undef GD_LAUNCH
    RST_LAUNCH_CHECK();
    *launched = true;
    return RST_OK;
}

}  // namespace rst
