// Y[b] = A[b] * X[b] (+ bias) (+ R[b]):  the 1x1-convolution / "small matrix
// times pixel stream" product, exact fp32 on the matrix cores
// (v_mfma_f32_16x16x4_f32: bitwise a k-ordered fmaf chain).
//
//   A : M x K, tiny (a weight matrix, or a per-sample folded attention matrix)
//   X : K x N per sample, N = H*W contiguous (NCHW needs no transposition:
//       the pixel axis is the GEMM's N axis and the MFMA B-operand reads 16
//       consecutive pixels per k)
//
// Block = 4 waves; the block owns BM = 16*MT output channels x 128 pixels, each
// wave all BM channels x 32 pixels (MT x 2 accumulator tiles of 16x16).  K is
// walked in chunks of 16 staged through LDS:
//   As[k][m]  (A transposed, row stride = 16*(MT|1) floats  -> the two k rows a
//              32-lane group reads fall on disjoint bank halves)
//   Xs[k][n]  (row stride 144 floats, same reason)
// The next chunk's global loads are issued before the current chunk's MFMAs and
// written to LDS after them, so HBM latency hides under the matrix work.
#include "kernels.hpp"

namespace rst {


constexpr int GN_BK = 16;
constexpr int GN_BN = 128;
constexpr int GN_LDX = GN_BN + 16;

template <int MT, bool VECX>
__global__ void __launch_bounds__(256)
gemm_nn_kernel(GemmNN g, int mblocks) {
    constexpr int BM = 16 * MT;
    constexpr int LDA = 16 * (MT | 1);
    __shared__ __attribute__((aligned(16))) float As[GN_BK * LDA];
    __shared__ __attribute__((aligned(16))) float Xs[GN_BK * GN_LDX];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int mb = blockIdx.x % mblocks;
    const int64_t nt = blockIdx.x / mblocks;
    const int b1 = blockIdx.y / g.nb2, b2 = blockIdx.y % g.nb2;
    const int m0 = mb * BM;
    const int64_t n0 = nt * GN_BN;

    const float* Ab = g.A + (int64_t)b1 * g.a_s1 + (int64_t)b2 * g.a_s2;
    const float* Xb = g.X + (int64_t)b1 * g.x_s1 + (int64_t)b2 * g.x_s2;

    // ---- staging assignments ----
    // X chunk: 16 rows x 128 px; thread -> rows (tid>>5) and (tid>>5)+8, 4 px at (tid&31)*4
    const int xr = tid >> 5, xc = (tid & 31) * 4;
    // A chunk: BM x 16 elements, linear index e = tid + i*256
    constexpr int A_ELEMS = BM * GN_BK;
    constexpr int A_PER_T = (A_ELEMS + 255) / 256;

    f32x4 xreg[2];
    float areg[A_PER_T];

    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + xr + 8 * h;
            const int64_t n = n0 + xc;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < g.K) {
                const float* src = Xb + (int64_t)k * g.ldx + n;
                if constexpr (VECX) {
                    if (n < g.N) v = *reinterpret_cast<const f32x4*>(src);   // N % 4 == 0
                } else {
                    if (n + 0 < g.N) v.x = src[0];
                    if (n + 1 < g.N) v.y = src[1];
                    if (n + 2 < g.N) v.z = src[2];
                    if (n + 3 < g.N) v.w = src[3];
                }
            }
            xreg[h] = v;
        }
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            const int e = tid + i * 256;
            float v = 0.f;
            if (e < A_ELEMS) {
                int m, k;
                if (g.transA) { m = e % BM; k = e / BM; }     // memory is k-major: m fastest
                else          { k = e % GN_BK; m = e / GN_BK; } // memory is m-major: k fastest
                const int gm = m0 + m, gk = k0 + k;
                if (gm < g.M && gk < g.K)
                    v = g.transA ? Ab[(int64_t)gk * g.lda + gm] : Ab[(int64_t)gm * g.lda + gk];
            }
            areg[i] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            *reinterpret_cast<f32x4*>(&Xs[(xr + 8 * h) * GN_LDX + xc]) = xreg[h];
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            const int e = tid + i * 256;
            if (e < A_ELEMS) {
                int m, k;
                if (g.transA) { m = e % BM; k = e / BM; }
                else          { k = e % GN_BK; m = e / GN_BK; }
                As[k * LDA + m] = areg[i];
            }
        }
    };

    f32x4 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int lm = lane & 15, kk = lane >> 4;
    const int wn0 = wave * 32;

    load_chunk(0);
    for (int k0 = 0; k0 < g.K; k0 += GN_BK) {
        __syncthreads();            // previous chunk's MFMA reads are done
        store_chunk();
        __syncthreads();
        if (k0 + GN_BK < g.K) load_chunk(k0 + GN_BK);
#pragma unroll
        for (int s = 0; s < GN_BK / 4; ++s) {
            const int kr = 4 * s + kk;
            const float xv0 = Xs[kr * GN_LDX + wn0 + lm];
            const float xv1 = Xs[kr * GN_LDX + wn0 + 16 + lm];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float a = As[kr * LDA + 16 * i + lm];
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv1, acc[i][1], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: D[row = 4*(lane>>4) + r][col = lane & 15] ----
    float* Yb = g.Y + (int64_t)b1 * g.y_s1 + (int64_t)b2 * g.y_s2;
    const float* Rb = g.R ? g.R + (int64_t)b1 * g.r_s1 + (int64_t)b2 * g.r_s2 : nullptr;
    const float* RSb = g.rscale ? g.rscale + (int64_t)b1 * g.rs_s1 + (int64_t)b2 * g.rs_s2 : nullptr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * i + 4 * kk + r;
            if (m < g.M) {
                const float bv = g.bias ? g.bias[m] : 0.f;
                const float rs = RSb ? RSb[m] : 1.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int64_t n = n0 + wn0 + 16 * j + lm;
                    if (n < g.N) {
                        float v = acc[i][j][r] + bv;
                        if (Rb) v = fmaf(rs, Rb[(int64_t)m * g.ldr + n], v);
                        float* dst = Yb + (int64_t)m * g.ldy + n;
                        if (g.accumulate) v += *dst;
                        *dst = v;
                    }
                }
            }
        }
    }
}

// Pick the number of 16-row tiles per block: fewest blocks along M first (every
// M-block re-reads the pixel stream), then the least padded rows.
static int pick_mt(int M) {
    static const int cand[] = {9, 8, 6, 4, 3, 2, 1};
    int best = 1;
    int64_t best_pad = INT64_MAX, best_blocks = INT64_MAX;
    for (int mt : cand) {
        const int bm = 16 * mt;
        const int64_t blocks = cdiv(M, bm);
        const int64_t pad = blocks * bm;
        if (blocks < best_blocks || (blocks == best_blocks && pad < best_pad)) {
            best = mt; best_pad = pad; best_blocks = blocks;
        }
    }
    return best;
}

int gemm_nn_res_try(const GemmNN& g, hipStream_t stream, bool* launched);
int gemm_nn_dma_try(const GemmNN& g, hipStream_t stream, bool* launched);

// one wave per weight row
__global__ void __launch_bounds__(64)
ln_fold_weights_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                       const float* __restrict__ bias, int K, float* __restrict__ folded, float* __restrict__ s,
                       float* __restrict__ c) {
    const int m = blockIdx.x, lane = threadIdx.x;
    float ss = 0.f, cc = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float wv = w[(int64_t)m * K + k];
        const float f = wv * gamma[k];
        folded[(int64_t)m * K + k] = f;
        ss += f;
        if (beta != nullptr) cc = fmaf(wv, beta[k], cc);
    }
    ss = wave_sum(ss);
    cc = wave_sum(cc);
    if (lane == 0) {
        s[m] = ss;
        c[m] = cc + (bias != nullptr ? bias[m] : 0.f);
    }
}

int ln_fold_weights_launch(const float* w, const float* gamma, const float* beta, const float* bias, int M, int K,
                           float* folded, float* s, float* c, hipStream_t stream) {
    RST_REQUIRE(w && gamma && folded && s && c && M > 0 && K > 0, "ln_fold_weights: bad operand");
    ProfScope prof(stream, "ln_fold", 2.0 * M * K, 8.0 * M * K);
    hipLaunchKernelGGL(ln_fold_weights_kernel, dim3((unsigned)M), dim3(64), 0, stream, w, gamma, beta, bias, K, folded, s, c);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

int gemm_nn_launch(const GemmNN& g, hipStream_t stream) {
    RST_REQUIRE(g.M > 0 && g.K > 0 && g.N > 0 && g.nb1 > 0 && g.nb2 > 0,
                "gemm_nn: empty problem (M=%d K=%d N=%lld batch=%dx%d)", g.M, g.K, (long long)g.N, g.nb1, g.nb2);
    if (g.afrag != nullptr) return gemm_f16s_launch(g, stream);
    RST_REQUIRE(g.A && g.X && g.Y, "gemm_nn: null operand");
    RST_REQUIRE((int64_t)g.nb1 * g.nb2 <= 65535, "gemm_nn: batch %dx%d exceeds grid.y", g.nb1, g.nb2);
    {
        bool done = false;
        RST_TRY(gemm_nn_res_try(g, stream, &done));
        if (done) return RST_OK;
        RST_TRY(gemm_nn_dma_try(g, stream, &done));
        if (done) return RST_OK;
    }
    RST_REQUIRE(g.ln_mode == 0, "gemm_nn: folded LayerNorm on an unsupported shape");
    const int mt = pick_mt(g.M);
    const int mblocks = (int)cdiv(g.M, 16 * mt);
    const int64_t ntiles = cdiv(g.N, GN_BN);
    const int64_t gx = ntiles * mblocks;
    RST_REQUIRE(gx < (1ll << 31), "gemm_nn: grid too large");
    const bool vecx = (g.N % 4 == 0) && (g.ldx % 4 == 0) && (g.x_s1 % 4 == 0) && (g.x_s2 % 4 == 0) && aligned16(g.X);
    dim3 grid((unsigned)gx, (unsigned)(g.nb1 * g.nb2)), blk(256);
    const double nb = (double)g.nb1 * g.nb2;
    ProfScope prof(stream, "gemm_nn", 2.0 * g.M * g.K * g.N * nb,
                   4.0 * nb * g.N * (g.K + g.M * (1.0 + (g.R ? 1.0 : 0.0) + (g.accumulate ? 1.0 : 0.0))));
#define GN_CASE(MTV)                                                                          \
    case MTV:                                                                                 \
        if (vecx) hipLaunchKernelGGL((gemm_nn_kernel<MTV, true>), grid, blk, 0, stream, g, mblocks);  \
        else      hipLaunchKernelGGL((gemm_nn_kernel<MTV, false>), grid, blk, 0, stream, g, mblocks); \
        break;
    switch (mt) {
        GN_CASE(1) GN_CASE(2) GN_CASE(3) GN_CASE(4) GN_CASE(6) GN_CASE(8) GN_CASE(9)
        default: return fail(RST_ERR_INVALID, "gemm_nn: bad tile");
    }
#undef GN_CASE
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst
