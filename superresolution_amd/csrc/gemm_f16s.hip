// Y[b] = A[b] * X[b] (+bias) (+rscale o R[b]) with the product on the f16 matrix instructions at fp32 accuracy:
// the 1x1 convolutions (model/Restromer.py:100,103,130,149,403,408) wherever the exact-fp32 instruction
// (v_mfma_f32_16x16x4_f32, 1/10 of the measured f16 rate) would be the bound.
//
// Both operands are split in two, v = hi + lo / 2048 with hi = f16(v), lo = f16((v - hi) * 2048) -- 22 significant
// bits -- and  A x = Ahi xhi + (Ahi xlo + Alo xhi) / 2048  takes three v_mfma_f32_16x16x32_f16 per 32 channels.
//   * A (weights, or the per-sample attention matrix) is split ONCE by gemm_f16s_prep into MFMA operand fragments in
//     global memory ([16-row tile][32-channel chunk][hi, lo][64 lanes] x 16 B; LayerNorm scale folded in, transposed
//     on the way if asked); the kernel brings the fragments of its rows into LDS by LDS-DMA, 1 KiB per piece.
//   * X stays fp32 NCHW in memory: a chunk [32 channels][128 pixels] comes in by LDS-DMA (2 channel rows per piece,
//     pieces 64 B apart so that the four lane groups of a fragment read four different bank quarters), and every wave
//     converts the fragments of its own 32 pixels while it reads them (the sums for a folded LayerNorm fall out).
// Lane group g of a fragment holds channels {2 g + 8 j + e : j < 4, e < 2} of the chunk -- for both operands alike.
// 8 waves = 2 row halves x 4 pixel quarters, block tile (32 RT rows) x 128 pixels, one barrier per chunk; two blocks of
// at most 128 rows per compute unit with a 2-deep ring each (or, RST_GF_ONE_BLOCK=1, one block of up to 256 rows with a
// 3-deep ring).
// |x| must stay below 65504 (f16 range); RST_F32_MFMA=1 keeps every product on the exact-fp32 kernels.
#include "kernels.hpp"
#include <atomic>

namespace rst {

namespace {

typedef __attribute__((address_space(1))) const void* gf_gptr;
typedef __attribute__((address_space(3))) void* gf_lptr;
typedef _Float16 gf_h8 __attribute__((ext_vector_type(8)));
typedef float gf_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 gf_h2 __attribute__((ext_vector_type(2)));

// Timing experiments only (results are wrong): -DGF_ABL=1 every weight-fragment piece re-reads one address, 2 every pixel
// piece does, 3 no stores, 4 no matrix instructions, 5 no conversion and no matrix instructions.  Measured on the 40
// LayerNorm-folded products of a 16 x 3 x 256 x 256 forward pass with one block of 192 rows per compute unit (3.08 ms):
// 1: 3.10 (the 9 GB of weight fragments that come out of L2 per pass cost nothing), 3: 2.50, 4: 2.32, 5: 1.99 -- two
// thirds of the time is what is left when nothing is computed: per block one exposed round trip for its first chunk,
// 6 to 12 barriers, and the stores.  Two smaller blocks per unit hide each other's share of that: 3.08 -> 2.98 ms, the
// products with a residual 1.80 -> 1.59 ms, although every pixel is then converted twice as often.
#ifndef GF_ABL
#define GF_ABL 0
#endif
constexpr int GF_BN = 128, GF_BK = 32, GF_XPIECE = 272, GF_SCR_LD = 36;

// One block of four waves per 16-row tile (and matrix); wave w splits chunks w, w + 4, ...
__global__ void __launch_bounds__(256)
gemm_f16s_prep_kernel(const float* __restrict__ w, int64_t w_bs, int lda, int transA, const float* __restrict__ gamma,
                      const float* __restrict__ beta, const float* __restrict__ bias, int M, int K, int kch,
                      gf_h8* __restrict__ frag, int64_t frag_bs16, float* __restrict__ s, float* __restrict__ c) {
    __shared__ float red[2][4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    const int mt = blockIdx.x, row = 16 * mt + m;
    const float* W = w + (int64_t)blockIdx.y * w_bs;
    gf_h8* out = frag + (int64_t)blockIdx.y * frag_bs16 + (int64_t)mt * kch * 2 * 64 + lane;
    float ssum = 0.f, csum = 0.f;
#pragma unroll 2
    for (int kc = wave; kc < kch; kc += 4) {
        gf_h8 hi, lo;
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) {
            const int k = 32 * kc + 2 * g + 8 * (idx >> 1) + (idx & 1);
            float v = 0.f;
            if (row < M && k < K) {
                const float w0 = transA ? W[(int64_t)k * lda + row] : W[(int64_t)row * lda + k];
                v = gamma ? w0 * gamma[k] : w0;
                if (beta) csum = fmaf(w0, beta[k], csum);
            }
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)((v - (float)h) * 2048.f);
            hi[idx] = h; lo[idx] = l;
            ssum += (float)h + (float)l * (1.0f / 2048.f);     // what the product will see: the mean term cancels exactly
        }
        out[(kc * 2 + 0) * 64] = hi;
        out[(kc * 2 + 1) * 64] = lo;
    }
    if (s != nullptr) {
        ssum += __shfl_xor(ssum, 16); csum += __shfl_xor(csum, 16);
        ssum += __shfl_xor(ssum, 32); csum += __shfl_xor(csum, 32);
        if (g == 0) { red[0][wave][m] = ssum; red[1][wave][m] = csum; }
        __syncthreads();
        if (wave == 0 && g == 0 && row < M) {
            s[row] = (red[0][0][m] + red[0][1][m]) + (red[0][2][m] + red[0][3][m]);
            c[row] = (red[1][0][m] + red[1][1][m]) + (red[1][2][m] + red[1][3][m]) + (bias ? bias[row] : 0.f);
        }
    }
}

// GF_STAGES = 3: one block per compute unit (up to 256 rows), two chunks in flight.  GF_STAGES = 2: at most 128 rows
// and 128 registers, so that TWO blocks share a compute unit (at most 67 KB of LDS each): a block lives for 6 to 15
// chunks only, and its prologue (the first chunk's round trip) and epilogue (the stores) hide behind the other
// block's products, which one block per unit cannot do.
template <int RT, bool RES, bool LN, int GF_STAGES>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(GF_STAGES == 2 ? 4 : 2, GF_STAGES == 2 ? 4 : 2)))
gemm_f16s_kernel(GemmNN g, int mblocks, int mtiles, int kch, int walk) {
    static_assert(GF_STAGES == 3 || RT <= 4, "two blocks per compute unit: at most 128 rows");
    constexpr int A_FLOATS = 2 * RT * 2 * 256;            // 2 RT tiles x (hi, lo) x 1 KiB
    constexpr int STAGE = A_FLOATS + 16 * GF_XPIECE;
    constexpr int NP = 4 * RT + 16, P = (NP + 7) / 8;     // DMA pieces per chunk, per wave
    extern __shared__ __attribute__((aligned(16))) float gf_ring[];
    float* ring = gf_ring;                                // GF_STAGES stages + 256 floats for surplus pieces

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kk = lane >> 4;
    const int nq = wave & 3, mh = wave >> 2;
    // the row blocks of one pixel tile read the same pixels: neighbours on one XCD, so the tile leaves HBM once
    int64_t bx = blockIdx.x, by = blockIdx.y;
    if (walk) {
        const int64_t lin = xcd_walk(by * gridDim.x + bx, (int64_t)gridDim.x * gridDim.y);
        bx = lin % gridDim.x; by = lin / gridDim.x;
    }
    const int mb = (int)(bx % mblocks);
    const int64_t nt = bx / mblocks;
    const int b1 = (int)(by / g.nb2), b2 = (int)(by % g.nb2);
    const int m0 = mb * 32 * RT;
    const int64_t n0 = nt * GF_BN;
    const char* Af = static_cast<const char*>(g.afrag) + (int64_t)b1 * g.afrag_s1 + (int64_t)b2 * g.afrag_s2;
    const float* Xb = g.X + (int64_t)b1 * g.x_s1 + (int64_t)b2 * g.x_s2 + n0;
    const float* X2b = g.X2 ? g.X2 + (int64_t)b1 * g.x2_s1 + (int64_t)b2 * g.x2_s2 + n0 : nullptr;
    const int csplit = g.X2 ? g.ksplit / GF_BK : (1 << 30);       // first chunk that reads the second tensor

    // ---- this wave's DMA pieces: ids wave + 8 i; [0, 4 RT): A fragments (tile, hi/lo), then 16 X pieces of two rows ----
    // per piece: a lane-dependent byte offset from a wave-uniform base that advances per chunk (A: 2 KiB, X: 32 rows),
    // so that an issue costs one address add per piece; rows past K (last chunk only) re-read row K - 1, whose
    // partner weights are zero
    int64_t poff[P], poff_last[P];
    int dsto[P];                  // LDS float offset inside a stage (< 0: surplus piece)
    bool isx[P];
    const int xcol = (lane & 31) * 4;
    const bool xok = n0 + xcol < g.N;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int id = wave + 8 * i;
        poff[i] = 0; poff_last[i] = 0; dsto[i] = -1; isx[i] = false;
        if (id < 4 * RT) {
            const int t = id >> 1, hl = id & 1;
            int mt = m0 / 16 + t;
            if (mt > mtiles - 1) mt = mtiles - 1;          // duplicate rows of the last tile are never stored
            poff[i] = ((int64_t)mt * kch * 2 + hl) * 1024 + lane * 16;
            poff_last[i] = poff[i];
            dsto[i] = id * 256;
        } else if (id < NP) {
            const int p = id - 4 * RT;
            isx[i] = true;
            const int xrow = 2 * p + (lane >> 5);
            int last = g.K - 1 - GF_BK * (kch - 1);        // last valid row of the last chunk
            if (last > xrow) last = xrow;
            poff[i] = ((int64_t)xrow * g.ldx + (xok ? xcol : 0)) * 4;
            poff_last[i] = ((int64_t)last * g.ldx + (xok ? xcol : 0)) * 4;
            dsto[i] = A_FLOATS + p * GF_XPIECE;
        }
    }
    const int64_t xstep = (int64_t)GF_BK * g.ldx * 4;
    auto issue = [&](int c) __attribute__((always_inline)) {
        float* st = ring + (c % GF_STAGES) * STAGE;
        const char* abase = Af + (int64_t)c * 2048;
        const char* xbase = c < csplit ? reinterpret_cast<const char*>(Xb) + (int64_t)c * xstep
                                       : reinterpret_cast<const char*>(X2b) + (int64_t)(c - csplit) * xstep;
        const bool lastc = c == kch - 1;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            float* dst = dsto[i] >= 0 ? st + dsto[i] : ring + GF_STAGES * STAGE;
            const char* src = (isx[i] ? xbase : abase) + (lastc ? poff_last[i] : poff[i]);
            if (GF_ABL == 1 && !isx[i]) src = static_cast<const char*>(g.afrag) + lane * 16;
            if (GF_ABL == 2 && isx[i]) src = reinterpret_cast<const char*>(g.X) + lane * 16;
            __builtin_amdgcn_global_load_lds((gf_gptr)src, (gf_lptr)dst, 16, 0, 0);
        }
    };

    f32x4 acc1[RT][2], acc2[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (g.bias != nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + mh * 16 * RT + 16 * i + 4 * kk + r;
                b4[r] = g.bias[m < g.M ? m : g.M - 1];
            }
        }
        acc1[i][0] = b4; acc1[i][1] = b4;
        acc2[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};

    issue(0);
    if (GF_STAGES == 3) issue(kch > 1 ? 1 : 0);
    for (int c = 0; c < kch; ++c) {
        // chunk c landed for this wave (chunk c + 1 may stay in flight), then for everyone; the barrier also frees
        // the stage of chunk c - 1 for the refill
        if (GF_STAGES == 3) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(P) : "memory");
            issue(c + 2 < kch ? c + 2 : kch - 1);
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (c + 1 < kch) issue(c + 1);
        }
        const float* st = ring + (c % GF_STAGES) * STAGE;
        if (GF_ABL == 5) continue;
        // fragments of this wave's 2 x 16 pixels: lane (pixel lm, group kk) takes rows 2 kk + 8 j + e
        gf_h8 xh[2], xl[2];
        const float* xs = st + A_FLOATS + kk * GF_XPIECE + nq * 32 + lm;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[2 * j] = xs[4 * j * GF_XPIECE + 16 * t];
                v[2 * j + 1] = xs[4 * j * GF_XPIECE + 128 + 16 * t];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // hi pair in one conversion; residual (v - hi) * 2048 as fma(hi, -2048, 2048 v): f16 operand read in place
                const gf_f32x2 vv = {v[2 * j], v[2 * j + 1]};
                const gf_h2 h = __builtin_convertvector(vv, gf_h2);
                const gf_f32x2 sc = vv * 2048.f;
                const gf_f32x2 rr2 = {fmaf((float)h[0], -2048.f, sc[0]), fmaf((float)h[1], -2048.f, sc[1])};
                const gf_h2 l = __builtin_convertvector(rr2, gf_h2);
                xh[t][2 * j] = h[0]; xh[t][2 * j + 1] = h[1];
                xl[t][2 * j] = l[0]; xl[t][2 * j + 1] = l[1];
            }
            if constexpr (LN) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool kv = GF_BK * c + 2 * kk + 8 * j < g.K;      // K is even: a pair is inside or outside together
                    const float a0 = kv ? v[2 * j] : 0.f, a1 = kv ? v[2 * j + 1] : 0.f;
                    s1[t] += a0 + a1;
                    s2[t] = fmaf(a0, a0, fmaf(a1, a1, s2[t]));
                }
            }
        }
        const gf_h8* af = reinterpret_cast<const gf_h8*>(st) + mh * RT * 2 * 64 + lane;
        if (GF_ABL == 4) {
            asm volatile("" :: "v"(xh[0]), "v"(xh[1]), "v"(xl[0]), "v"(xl[1]));
            continue;
        }
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const gf_h8 ah = af[(i * 2 + 0) * 64], al = af[(i * 2 + 1) * 64];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc1[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[t], acc1[i][t], 0, 0, 0);
                acc2[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[t], acc2[i][t], 0, 0, 0);
                acc2[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[t], acc2[i][t], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: per 16 x 32 tile through a wave-private LDS transpose, float4 row stores ----
    float* Yb = g.Y + (int64_t)b1 * g.y_s1 + (int64_t)b2 * g.y_s2;
    const float* Rb = RES ? g.R + (int64_t)b1 * g.r_s1 + (int64_t)b2 * g.r_s2 : nullptr;
    const float* RSb = (RES && g.rscale) ? g.rscale + (int64_t)b1 * g.rs_s1 + (int64_t)b2 * g.rs_s2 : nullptr;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // the ring is free: reuse it
    float* scr = ring + wave * 2 * 16 * GF_SCR_LD;
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
    for (int i = 0; i < RT; ++i) {
        float* sl = scr + (i & 1) * 16 * GF_SCR_LD;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v0 = fmaf(acc2[i][0][r], 1.0f / 2048.f, acc1[i][0][r]);
            float v1 = fmaf(acc2[i][1][r], 1.0f / 2048.f, acc1[i][1][r]);
            if constexpr (LN) {
                // (read here, not ahead of the main loop: eight registers per row tile that the products do not have)
                const int m = m0 + mh * 16 * RT + 16 * i + 4 * kk + r;
                const int mc = m < g.M ? m : g.M - 1;
                const float lns = g.ln_s[mc], lnc = g.ln_c[mc];
                v0 = fmaf(rstd[0], fmaf(tco[0], lns, v0), lnc);
                v1 = fmaf(rstd[1], fmaf(tco[1], lns, v1), lnc);
            }
            sl[(4 * kk + r) * GF_SCR_LD + lm] = v0;
            sl[(4 * kk + r) * GF_SCR_LD + 16 + lm] = v1;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = rr + 8 * h;
            f32x4 v = *reinterpret_cast<const f32x4*>(&sl[row * GF_SCR_LD + c4]);
            const int m = m0 + mh * 16 * RT + 16 * i + row;
            if (m < g.M && nok && GF_ABL != 3) {
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

// rows per block = 32 RT: least padding, ties to the larger tile (the pixel tile is re-read once per row block)
int gf_pick_rt(int M, int rt_max) {
    int best = 1;
    int64_t best_pad = INT64_MAX;
    for (int rt = rt_max; rt >= 1; --rt) {
        const int64_t pad = cdiv(M, 32 * rt) * 32 * rt;
        if (pad < best_pad) { best_pad = pad; best = rt; }
    }
    return best;
}

}  // namespace

size_t gemm_f16s_frag_bytes(int M, int K) { return (size_t)round_up(M, 16) * (size_t)round_up(K, 32) * 4; }

// -1: not set by the caller, RST_F32_MFMA in the environment decides; 0 / 1: rst_set_exact_fp32
static std::atomic<int> g_exact_fp32{-1};
bool gemm_f16s_enabled() {
    const int e = g_exact_fp32.load(std::memory_order_relaxed);
    if (e >= 0) return e == 0;
    static const bool off = getenv("RST_F32_MFMA") != nullptr && getenv("RST_F32_MFMA")[0] == '1';
    return !off;
}
void set_exact_fp32(int on) { g_exact_fp32.store(on ? 1 : 0, std::memory_order_relaxed); }

bool gemm_f16s_supported(const GemmNN& g) {
    const bool al = aligned16(g.X) && aligned16(g.Y) && (g.R == nullptr || aligned16(g.R));
    const bool strides4 = (g.N % 4 == 0) && (g.ldx % 4 == 0) && (g.x_s1 % 4 == 0) && (g.x_s2 % 4 == 0) &&
                          (g.ldy % 4 == 0) && (g.y_s1 % 4 == 0) && (g.y_s2 % 4 == 0) &&
                          (g.R == nullptr || ((g.ldr % 4 == 0) && (g.r_s1 % 4 == 0) && (g.r_s2 % 4 == 0)));
    // (an odd K is fine without a folded LayerNorm: the missing row re-reads row K - 1 against zero weights; the
    // statistics of a folded LayerNorm mask channel pairs, so there K must be even)
    return al && strides4 && (g.K % 2 == 0 || g.ln_mode == 0) && g.K >= 2 && (int64_t)g.nb1 * g.nb2 <= 65535;
}

int gemm_f16s_prep_launch(const float* w, int64_t w_bs, int nb, int lda, int transA, const float* gamma, const float* beta,
                          const float* bias, int M, int K, void* frag, int64_t frag_bs_bytes, float* s, float* c,
                          hipStream_t stream) {
    RST_REQUIRE(w && frag && M > 0 && K > 0 && nb > 0 && nb <= 65535, "gemm_f16s_prep: bad operand");
    RST_REQUIRE(aligned16(frag) && frag_bs_bytes % 16 == 0, "gemm_f16s_prep: fragment buffer misaligned");
    RST_REQUIRE((s == nullptr) == (c == nullptr), "gemm_f16s_prep: s and c come together");
    const int mtiles = (int)cdiv(M, 16), kch = (int)cdiv(K, 32);
    hipLaunchKernelGGL(gemm_f16s_prep_kernel, dim3((unsigned)mtiles, (unsigned)nb), dim3(256), 0, stream, w, w_bs, lda, transA,
                       gamma, beta, bias, M, K, kch, static_cast<gf_h8*>(frag), frag_bs_bytes / 16, s, c);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

int gemm_f16s_launch(const GemmNN& g, hipStream_t stream) {
    RST_REQUIRE(g.afrag && g.X && g.Y && g.M > 0 && g.K > 0 && g.N > 0 && g.nb1 > 0 && g.nb2 > 0, "gemm_f16s: bad operand");
    RST_REQUIRE(gemm_f16s_supported(g), "gemm_f16s: unsupported strides / alignment");
    RST_REQUIRE(g.X2 == nullptr || (g.ksplit > 0 && g.ksplit < g.K && g.ksplit % GF_BK == 0 && aligned16(g.X2) &&
                                    g.x2_s1 % 4 == 0 && g.x2_s2 % 4 == 0 && g.ln_mode == 0),
                "gemm_f16s: the second input tensor needs a split at a multiple of 32 channels");
    RST_REQUIRE(g.ln_mode == 0 || (g.ln_s && g.ln_c && g.R == nullptr && !g.accumulate && g.bias == nullptr && g.K % 2 == 0),
                "gemm_f16s: folded LayerNorm takes no residual / bias / accumulation and an even channel count");
    // two blocks per compute unit (see the kernel) unless RST_GF_ONE_BLOCK=1 (A/B runs)
    static const bool one_block = getenv("RST_GF_ONE_BLOCK") != nullptr && getenv("RST_GF_ONE_BLOCK")[0] == '1';
    const int stages = one_block ? 3 : 2;
    const int rt = gf_pick_rt(g.M, one_block ? 8 : g.ln_mode != 0 ? 3 : 4);      // (the LayerNorm form spills at 4 x 32 rows)
    const int mblocks = (int)cdiv(g.M, 32 * rt), mtiles = (int)cdiv(g.M, 16), kch = (int)cdiv(g.K, 32);
    const int64_t gx = cdiv(g.N, GF_BN) * mblocks;
    RST_REQUIRE(gx < (1ll << 31), "gemm_f16s: grid too large");
    dim3 grid((unsigned)gx, (unsigned)(g.nb1 * g.nb2)), blk(512);
    const size_t lds = sizeof(float) * (size_t)(stages * (2 * rt * 2 * 256 + 16 * GF_XPIECE) + 256);
    const double nb = (double)g.nb1 * g.nb2;
    ProfScope prof(stream, g.ln_mode ? "gemm_f16s_ln" : "gemm_f16s", 2.0 * g.M * g.K * g.N * nb,
                   4.0 * nb * g.N * (g.K + g.M * (1.0 + (g.R ? 1.0 : 0.0) + (g.accumulate ? 1.0 : 0.0))));
    const void* kern = nullptr;
#define GF_PICK(RTV, STG)                                                                                   \
    case RTV:                                                                                               \
        kern = g.ln_mode != 0 ? reinterpret_cast<const void*>(&gemm_f16s_kernel<RTV, false, true, STG>)     \
             : g.R            ? reinterpret_cast<const void*>(&gemm_f16s_kernel<RTV, true, false, STG>)     \
                              : reinterpret_cast<const void*>(&gemm_f16s_kernel<RTV, false, false, STG>);   \
        break;
    if (stages == 2) {
        switch (rt) { GF_PICK(1, 2) GF_PICK(2, 2) GF_PICK(3, 2) GF_PICK(4, 2) default: return fail(RST_ERR_INVALID, "gemm_f16s: bad tile"); }
    } else {
        switch (rt) { GF_PICK(1, 3) GF_PICK(2, 3) GF_PICK(3, 3) GF_PICK(4, 3) GF_PICK(5, 3) GF_PICK(6, 3) GF_PICK(7, 3) GF_PICK(8, 3) default: return fail(RST_ERR_INVALID, "gemm_f16s: bad tile"); }
    }
#undef GF_PICK
    KernelSetup su;
    RST_TRY(kernel_setup(kern, 512, lds, &su));
    GemmNN gg = g;
    int walk = xcd_walk_enabled();
    void* args[] = {(void*)&gg, (void*)&mblocks, (void*)&mtiles, (void*)&kch, (void*)&walk};
    RST_HIP(hipLaunchKernel(kern, grid, blk, args, lds, stream));
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst

extern "C" {
void rst_set_exact_fp32(int32_t on) { rst::set_exact_fp32(on); }
int32_t rst_get_exact_fp32(void) { return rst::gemm_f16s_enabled() ? 0 : 1; }
}
