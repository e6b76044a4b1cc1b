// Y[b] = A[b] * X[b] (+bias) (+rscale o R[b]) for small K (<= 128): the 1x1
// convolutions of the high-resolution levels, where the pixel stream is huge
// and the weight matrix tiny.  "Resident" variant of gemm_nn.hip:
//
//  * persistent workgroups (one per CU, 4 waves = one wave per SIMD, so every
//    lane owns up to 512 registers);
//  * the weight tile lives in REGISTERS for the whole launch, already in MFMA
//    A-operand form (v_mfma_f32_16x16x4_f32: lane (m = l&15, kk = l>>4) holds
//    A[m][4s+kk] for every k-step s), so the inner loop touches LDS only for
//    the pixel operand and nothing is re-staged per tile;
//  * the pixel tile [K][128] is brought in by LDS-DMA (global_load_lds, 16 B per
//    lane, no register staging) into a two-deep ring: tile t+1 streams in while
//    tile t is multiplied;
//  * the output leaves through a small per-wave LDS transpose so that global
//    stores (and residual loads) are float4 on 128-byte row segments.
//
// Exact fp32 (each MFMA is a k-ordered fmaf chain), same numerics as gemm_nn.hip.
#include "kernels.hpp"

namespace rst {

constexpr int GR_BN = 64;                  // pixels per tile: 2 pixel halves x 32 per wave
constexpr int GR_THREADS = 256;            // 4 waves = 2 row halves x 2 pixel halves; two such blocks share a CU
constexpr int GR_SCR_LD = 36;                 // per-wave transpose scratch row stride (floats)
constexpr int GR_SCRATCH = 4 * 2 * 16 * GR_SCR_LD;   // 4 waves x 2 slots; also the weight staging area (16 x (K+1) floats)
// scratch = 4 waves x 16 x 36 floats = 2304 floats; as weight staging it holds 16 x (K+1) <= 16 x 121

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MT, int KS, bool RES, bool LN>
__global__ void __launch_bounds__(GR_THREADS, 2)
gemm_nn_res_kernel(GemmNN g, int mblocks, int ntiles, int64_t total_tiles, int groups) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KROWS = 4 * KS;            // == K, checked by the launcher
    constexpr int BUF = KROWS * GR_BN;
    float* scratch = smem + 2 * BUF;          // [4 waves][2 slots][16][GR_SCR_LD]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lm = lane & 15, kk = lane >> 4;
    const int wn0 = (wave & 1) * 32;          // pixel half of this wave
    const int mg = wave >> 1;                 // which half of the block's rows
    // blocks 8 apart share an XCD: put the M-blocks of one pixel range there
    const int bid = blockIdx.x;
    const int rem = bid % (8 * mblocks);
    const int mb = rem / 8;
    const int j = (bid / (8 * mblocks)) * 8 + (rem % 8);
    const int64_t t_begin = total_tiles * j / groups, t_end = total_tiles * (j + 1) / groups;
    const int m0 = mb * 32 * MT + mg * 16 * MT;   // first row of this wave

    auto issue_tile = [&](int64_t t, int buf) {
        const int64_t bb = t / ntiles;
        const int64_t n0 = (t % ntiles) * GR_BN;
        const int b1 = (int)(bb / g.nb2), b2 = (int)(bb % g.nb2);
        const float* Xb = g.X + (int64_t)b1 * g.x_s1 + (int64_t)b2 * g.x_s2 + n0;
        // LDS chunk c of sub-row j holds pixel chunk c ^ 4*(j&1): odd k-rows are stored with their two
        // 16-pixel halves swapped, so the four k-rows of an operand read fall on disjoint banks
        const int col = (((lane & 15) ^ (((lane >> 4) & 1) << 2))) * 4;
        const bool ok = n0 + col < g.N;       // N % 4 == 0
        for (int p = wave; 4 * p < g.K; p += GR_THREADS / 64) {        // one wave instruction = 4 channel rows of 64 pixels
            const int row = 4 * p + (lane >> 4);
            float* dst = smem + buf * BUF + 4 * p * GR_BN;          // wave-uniform base
            if (ok)
                __builtin_amdgcn_global_load_lds((gptr_t)(Xb + (int64_t)row * g.ldx + col), (lptr_t)dst, 16, 0, 0);
        }
    };

    float wreg[MT][KS];
    int64_t cur_b = -1;
    f32x4 acc[MT][2];
    // folded LayerNorm: per-row constants of the block's rows, kept in LDS behind the scratch
    float* lnS = scratch + GR_SCRATCH;      // [32*MT] row sums, then [32*MT] row constants
    float* lnC = lnS + 32 * MT;
    if constexpr (LN) {
        for (int e = tid; e < 32 * MT; e += GR_THREADS) {
            const int m = mb * 32 * MT + e;
            const int mc = m < g.M ? m : g.M - 1;
            lnS[e] = g.ln_s[mc];
            lnC[e] = g.ln_c[mc];
        }
        __syncthreads();
    }

    if (t_begin < t_end) issue_tile(t_begin, 0);
    int buf = 0;
    for (int64_t t = t_begin; t < t_end; ++t, buf ^= 1) {
        const int64_t bb = t / ntiles;
        const int64_t n0 = (t % ntiles) * GR_BN;
        const int b1 = (int)(bb / g.nb2), b2 = (int)(bb % g.nb2);
        if (bb != cur_b) {
            cur_b = bb;
            const float* Ab = g.A + (int64_t)b1 * g.a_s1 + (int64_t)b2 * g.a_s2;
            // Stage each 16-row weight tile through LDS with coalesced loads, then every lane
            // picks its A-operand fragments (lane (m, kk) <- A[m][4s+kk]).  The staging area is
            // the epilogue scratch; the change of sample is uniform over the block.
            constexpr int K = 4 * KS;
            constexpr int SLD = K + 1;
            float* st = scratch;
            static_assert(16 * SLD <= GR_SCRATCH, "staging area too small");
            const int mblk = mb * 32 * MT;
#pragma unroll 1
            for (int i2 = 0; i2 < 2 * MT; ++i2) {
                __syncthreads();
                for (int e = tid; e < 16 * K; e += GR_THREADS) {
                    int r, k;
                    if (g.transA) { k = e >> 4; r = e & 15; } else { r = e / K; k = e - r * K; }
                    const int m = mblk + 16 * i2 + r;
                    float v = 0.f;
                    if (m < g.M) v = g.transA ? Ab[(int64_t)k * g.lda + m] : Ab[(int64_t)m * g.lda + k];
                    st[r * SLD + k] = v;
                }
                __syncthreads();
                // static register index needs a compile-time i: dispatch over the unrolled copy
#pragma unroll
                for (int ii = 0; ii < MT; ++ii) {
                    if (mg * MT + ii == i2) {
#pragma unroll
                        for (int s = 0; s < KS; ++s) wreg[ii][s] = st[lm * SLD + 4 * s + kk];
                    }
                }
            }
            __syncthreads();
        }
        // tile t has landed (own DMA done, then everyone's), and everyone has finished reading the other buffer
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + 1 < t_end) issue_tile(t + 1, buf ^ 1);

        // residual rows, bias and row scales of this tile: loaded now, consumed after the MFMAs
        const float* Rb = g.R ? g.R + (int64_t)b1 * g.r_s1 + (int64_t)b2 * g.r_s2 : nullptr;
        const float* RSb = g.rscale ? g.rscale + (int64_t)b1 * g.rs_s1 + (int64_t)b2 * g.rs_s2 : nullptr;
        const int rr = lane >> 3, c4 = (lane & 7) * 4;
        const int64_t n = n0 + wn0 + c4;
        const bool nok = n < g.N;
        constexpr int RN = RES ? MT : 1;
        f32x4 rv[RN][2];
        float rsv[RN][2];
        if constexpr (RES) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int m = m0 + 16 * i + rr + 8 * h;
                    const int mc = m < g.M ? m : g.M - 1;
                    rv[i][h] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (nok) rv[i][h] = *reinterpret_cast<const f32x4*>(Rb + (int64_t)mc * g.ldr + n);
                    rsv[i][h] = RSb ? RSb[mc] : 1.f;
                }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // B operands of a k-step: pixel columns wn0+lm and wn0+16+lm of k-row 4s+kk (halves swapped in
        // odd k-rows, see issue_tile).  The two reads of step s+1 are issued by hand before the MFMAs of
        // step s (the compiler would otherwise sink them next to their use and the matrix pipe would
        // wait for each LDS round trip); LDS reads return in order, so lgkmcnt(2) = "step s has landed".
        const int sw = (kk & 1) << 4;
        const unsigned xa0 = (unsigned)(uintptr_t)(lptr_t)(smem + buf * BUF + kk * GR_BN + ((wn0 + lm) ^ sw));
        const unsigned xa1 = (unsigned)(uintptr_t)(lptr_t)(smem + buf * BUF + kk * GR_BN + ((wn0 + 16 + lm) ^ sw));
        float xq[2][2];
#define GR_FETCH(S)                                                                                             \
        do {                                                                                                    \
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(xq[(S) & 1][0]) : "v"(xa0), "n"((S) * 4 * GR_BN * 4)); \
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(xq[(S) & 1][1]) : "v"(xa1), "n"((S) * 4 * GR_BN * 4)); \
        } while (0)
        // LayerNorm statistics of this wave's two pixel columns (sum and sum of squares over the
        // channels, two columns per packed instruction) ride along with the product
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};     // plain (not packed) ALU ops: packed fp32 beside MFMAs costs extra issue time
        GR_FETCH(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) {
                GR_FETCH(s + 1);
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xq[s & 1][0]), "+v"(xq[s & 1][1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xq[s & 1][0]), "+v"(xq[s & 1][1]));
            }
            const float x0 = xq[s & 1][0], x1 = xq[s & 1][1];
            if constexpr (LN) {
                s1[0] += x0; s2[0] = fmaf(x0, x0, s2[0]);
                s1[1] += x1; s2[1] = fmaf(x1, x1, s2[1]);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[i][s], x0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[i][s], x1, acc[i][1], 0, 0, 0);
            }
        }

#undef GR_FETCH
        // ---- epilogue: 16 x 32 tile -> wave-private LDS -> float4 rows ----
        float* Yb = g.Y + (int64_t)b1 * g.y_s1 + (int64_t)b2 * g.y_s2;
        // Two scratch slots per wave, software-pipelined: tile i+1 is written while tile i's rows
        // are still on their way back from LDS, so the LDS round trip hides behind the next
        // tile's writes instead of stalling every step.  (LDS operations of one wave complete in
        // issue order, so slot reuse two steps later needs no wait.)
        float* scr = scratch + wave * 2 * 16 * GR_SCR_LD;   // 4 waves x 2 slots
        float rstd[2] = {1.f, 1.f}, tco[2] = {0.f, 0.f};
        if constexpr (LN) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float a = s1[j], q = s2[j];
                a += __shfl_xor(a, 16); q += __shfl_xor(q, 16);
                a += __shfl_xor(a, 32); q += __shfl_xor(q, 32);
                const float mu = a * (1.0f / KROWS);
                const float var = fmaxf(q * (1.0f / KROWS) - mu * mu, 0.f);
                rstd[j] = 1.0f / sqrtf(var + 1e-5f);
                tco[j] = g.ln_mode == 1 ? -mu : 0.f;
            }
        }
        auto put = [&](int i) {
            float* sl = scr + (i & 1) * 16 * GR_SCR_LD;
            f32x4 ls = {0.f, 0.f, 0.f, 0.f}, lc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (LN) {
                ls = *reinterpret_cast<const f32x4*>(&lnS[mg * 16 * MT + 16 * i + 4 * kk]);
                lc = *reinterpret_cast<const f32x4*>(&lnC[mg * 16 * MT + 16 * i + 4 * kk]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v0 = acc[i][0][r], v1 = acc[i][1][r];
                if constexpr (LN) {
                    v0 = fmaf(rstd[0], fmaf(tco[0], ls[r], v0), lc[r]);
                    v1 = fmaf(rstd[1], fmaf(tco[1], ls[r], v1), lc[r]);
                }
                sl[(4 * kk + r) * GR_SCR_LD + lm] = v0;
                sl[(4 * kk + r) * GR_SCR_LD + 16 + lm] = v1;
            }
        };
        f32x4 pend[2];
        auto get = [&](int i) {
            const float* sl = scr + (i & 1) * 16 * GR_SCR_LD;
            pend[0] = *reinterpret_cast<const f32x4*>(&sl[rr * GR_SCR_LD + c4]);
            pend[1] = *reinterpret_cast<const f32x4*>(&sl[(rr + 8) * GR_SCR_LD + c4]);
        };
        auto emit = [&](int i, const f32x4 (&val)[2]) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 v = val[h];
                const int m = m0 + 16 * i + rr + 8 * h;
                if (g.bias) { const float a = g.bias[m < g.M ? m : g.M - 1]; v.x += a; v.y += a; v.z += a; v.w += a; }
                if constexpr (RES) {
                    const float rs = rsv[i][h];
                    v.x = fmaf(rs, rv[i][h].x, v.x); v.y = fmaf(rs, rv[i][h].y, v.y);
                    v.z = fmaf(rs, rv[i][h].z, v.z); v.w = fmaf(rs, rv[i][h].w, v.w);
                }
                if (m < g.M && nok) *reinterpret_cast<f32x4*>(Yb + (int64_t)m * g.ldy + n) = v;
            }
        };
        put(0);
        get(0);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (i + 1 < MT) put(i + 1);
            f32x4 cur[2] = {pend[0], pend[1]};
            if (i + 1 < MT) get(i + 1);
            emit(i, cur);
        }
    }
}

// Eligibility and launch.  Returns RST_OK and sets *launched = true when the
// resident kernel took the problem; otherwise leaves it to the generic kernel.
// m-tiles (16 rows) per wave; a block covers 2*MT tiles.  Fewest blocks, then least padding.
static int res_pick_mt(int M, int max_mt) {
    static const int cand[] = {5, 4, 3, 2};
    int best = 0;
    int64_t best_pad = INT64_MAX, best_blocks = INT64_MAX;
    for (int mt : cand) {
        if (mt > max_mt) continue;
        const int64_t blocks = cdiv(M, 32 * mt), pad = blocks * 32 * mt;
        if (blocks < best_blocks || (blocks == best_blocks && pad < best_pad)) { best = mt; best_pad = pad; best_blocks = blocks; }
    }
    return best;
}

int gemm_nn_res_try(const GemmNN& g, hipStream_t stream, bool* launched) {
    *launched = false;
    const bool al = aligned16(g.X) && aligned16(g.Y) && (g.R == nullptr || aligned16(g.R));
    const bool strides4 = (g.N % 4 == 0) && (g.ldx % 4 == 0) && (g.x_s1 % 4 == 0) && (g.x_s2 % 4 == 0) &&
                          (g.ldy % 4 == 0) && (g.y_s1 % 4 == 0) && (g.y_s2 % 4 == 0) &&
                          (g.R == nullptr || ((g.ldr % 4 == 0) && (g.r_s1 % 4 == 0) && (g.r_s2 % 4 == 0)));
    // residual shapes are served at least as well by the general DMA kernel (more waves in flight)
    if (!al || !strides4 || g.accumulate || g.R != nullptr) return RST_OK;
    if (g.ln_mode != 0 && g.K != 48 && g.K != 96) return RST_OK;    // folded LayerNorm: compiled for the two widths it meets here
    // weight tiles are compiled for the channel counts of the dim=48 network's high-resolution levels
    if (g.K != 48 && g.K != 60 && g.K != 96 && g.K != 120) return RST_OK;
    const int64_t ntiles = cdiv(g.N, GR_BN);
    const int64_t total = ntiles * g.nb1 * g.nb2;
    if (total < 512 || ntiles > (1ll << 30)) return RST_OK;      // too little work to amortise resident weights
    // five row tiles of 96-deep weights plus the LayerNorm statistics do not fit the register file
    const int mt = res_pick_mt(g.M, (g.ln_mode != 0 && g.K > 48) ? 4 : 5);
    const int mblocks = (int)cdiv(g.M, 32 * mt);
    if (mblocks > 4) return RST_OK;
    const int ksm = g.K / 4;

    int cus = 256;
    const size_t lds = (size_t)(2 * 4 * ksm * GR_BN + GR_SCRATCH + 2 * 32 * mt) * sizeof(float);   // scratch doubles as weight staging (16 x (K+1))
    int per_cu = 1;
    int groups = 0;
    dim3 blk(GR_THREADS);

    const double nb = (double)g.nb1 * g.nb2;
    ProfScope prof(stream, g.ln_mode ? "gemm_nn_ln" : "gemm_nn", 2.0 * g.M * g.K * g.N * nb,
                   4.0 * nb * g.N * (g.K + g.M * (1.0 + (g.R ? 1.0 : 0.0) + (g.accumulate ? 1.0 : 0.0))));
#define GR_LAUNCH(MTV, KSV)                                                                                   \
    do {                                                                                                      \
        auto kern = gemm_nn_res_kernel<MTV, KSV, false, false>;                                               \
        if constexpr (KSV == 12 || KSV == 24) { if (g.ln_mode != 0) kern = gemm_nn_res_kernel<MTV, KSV, false, true>; } \
        KernelSetup su;                                                                                       \
        RST_TRY(kernel_setup(reinterpret_cast<const void*>(kern), GR_THREADS, lds, &su));                     \
        cus = su.cus; per_cu = su.per_cu;                                                                     \
        if (per_cu > 2) per_cu = 2;                                                                           \
        groups = cus * per_cu / mblocks;                                                                      \
        groups -= groups % 8;                                                                                 \
        if (groups > total) groups = (int)(total - total % 8);                                                \
        if (groups < 8) return RST_OK;                                                                        \
        hipLaunchKernelGGL(kern, dim3((unsigned)(groups * mblocks)), blk, lds, stream, g, mblocks, (int)ntiles, total, groups); \
    } while (0)
#define GR_KS(MTV)                                  \
    switch (ksm) {                                  \
        case 12: GR_LAUNCH(MTV, 12); break;         \
        case 15: GR_LAUNCH(MTV, 15); break;         \
        case 24: GR_LAUNCH(MTV, 24); break;         \
        default: GR_LAUNCH(MTV, 30); break;         \
    }
    switch (mt) {
        case 2: GR_KS(2); break;
        case 3: GR_KS(3); break;
        case 4: GR_KS(4); break;
        case 5: GR_KS(5); break;
        default: return RST_OK;
    }
#undef GR_KS
#undef GR_LAUNCH
    RST_LAUNCH_CHECK();
    *launched = true;
    return RST_OK;
}

}  // namespace rst
