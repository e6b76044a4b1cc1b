// bf16 storage path (inference): activations are bf16 in memory, every sum is fp32.
//
// Layout.  Activations are "channel-packed": [B][Cb][H*W][8] with Cb = ceil(C/8); the 8 channels of a
// block sit in one 16-byte chunk per pixel.  That chunk is exactly the B operand of
// v_mfma_f32_16x16x32_bf16 (lane (n, g) holds 8 consecutive k of pixel n), so a 1x1 convolution reads
// its operand straight from memory into the matrix pipe: no LDS staging and no transposition.  The
// rows of the weight matrix are permuted when the weights are prepared so that the 4 + 4 accumulator
// rows a lane holds for two 16-row tiles are 8 consecutive channels: the result leaves as one 16-byte
// chunk per lane in the same packed layout.  The depthwise 3x3 reads and writes 16-byte chunks per
// pixel, and the Gram product q k^T over the pixels transposes its operands on the way out of LDS with
// ds_read_b64_tr_b16.
#include "kernels.hpp"
#include "bf16_common.hpp"
#include <type_traits>

namespace rst {

// ---- weight preparation -----------------------------------------------------------------------------------
// frag[b][mt][ks][lane][j] = A operand of m-tile mt, k-step ks:  row = permuted position, k = 32 ks + 8 (lane>>4) + j
// position of (mt, rr = lane & 15): 32 (mt>>1) + 8 (rr>>2) + 4 (mt&1) + (rr&3)
__device__ __forceinline__ void
pk_prep_a_body(int bx, int b, const float* __restrict__ w, int64_t w_bs, int ldw, const float* __restrict__ gamma,
               int M, int K, int seg, int seg_pad, int ntiles, int KS, bf16_t* __restrict__ frag, int64_t frag_bs) {
    const int64_t e = (int64_t)bx * 256 + threadIdx.x;                  // (mt, ks, lane)
    if (e >= (int64_t)ntiles * KS * 64) return;
    const int lane = (int)(e & 63), ks = (int)((e >> 6) % KS), mt = (int)((e >> 6) / KS);
    const int rr = lane & 15;
    const int pos = 32 * (mt >> 1) + 8 * (rr >> 2) + 4 * (mt & 1) + (rr & 3);
    const int ch = src_channel(pos, seg, seg_pad, M);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 32 * ks + 8 * (lane >> 4) + j;
        float x = 0.f;
        if (ch >= 0 && k < K) {
            x = w[(int64_t)b * w_bs + (int64_t)ch * ldw + k];
            if (gamma != nullptr) x *= gamma[k];
        }
        v[j] = x;
    }
    *reinterpret_cast<u32x4*>(frag + (int64_t)b * frag_bs + e * 8) = pack8(v);
}
__global__ void __launch_bounds__(256)
pk_prep_a_kernel(const float* __restrict__ w, int64_t w_bs, int ldw, const float* __restrict__ gamma,
                 int M, int K, int seg, int seg_pad, int ntiles, int KS, bf16_t* __restrict__ frag, int64_t frag_bs) {
    pk_prep_a_body(blockIdx.x, blockIdx.y, w, w_bs, ldw, gamma, M, K, seg, seg_pad, ntiles, KS, frag, frag_bs);
}

// s[pos] = sum_k bf16(w[ch][k] gamma[k]) (the rounded values, so that the mean term cancels what the product adds);
// c[pos] = sum_k w[ch][k] beta[k] + bias[ch];  zero at padding positions.  One wave per position.
__device__ __forceinline__ void
pk_prep_sc_body(int bx, const float* __restrict__ w, int ldw, const float* __restrict__ gamma, const float* __restrict__ beta,
                const float* __restrict__ bias, int M, int K, int seg, int seg_pad, int npos,
                float* __restrict__ s, float* __restrict__ c) {
    const int pos = bx * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pos >= npos) return;
    const int ch = src_channel(pos, seg, seg_pad, M);
    float ss = 0.f, cc = 0.f;
    if (ch >= 0) {
        for (int k = lane; k < K; k += 64) {
            const float x = w[(int64_t)ch * ldw + k];
            if (gamma != nullptr) ss += bf_lo(to_bf16_bits(x * gamma[k]));
            if (beta != nullptr) cc = fmaf(x, beta[k], cc);
        }
        ss = wave_sum(ss);
        cc = wave_sum(cc);
        if (bias != nullptr) cc += bias[ch];
    }
    if (lane == 0) { s[pos] = ss; c[pos] = cc; }
}
__global__ void __launch_bounds__(256)
pk_prep_sc_kernel(const float* __restrict__ w, int ldw, const float* __restrict__ gamma, const float* __restrict__ beta,
                  const float* __restrict__ bias, int M, int K, int seg, int seg_pad, int npos,
                  float* __restrict__ s, float* __restrict__ c) {
    pk_prep_sc_body(blockIdx.x, w, ldw, gamma, beta, bias, M, K, seg, seg_pad, npos, s, c);
}

// depthwise weights per channel block: 72 taps [tap][8 channels] + 8 biases
__device__ __forceinline__ void
pk_prep_dw_body(int bx, const float* __restrict__ w, const float* __restrict__ bias, int C, int seg, int seg_pad, int Cb,
                float* __restrict__ out, int flip) {
    const int e = bx * 256 + threadIdx.x;
    if (e >= Cb * 80) return;
    const int cb = e / 80, r = e % 80, tap = r / 8, j = r % 8;
    const int ch = src_channel(cb * 8 + j, seg, seg_pad, C);
    float v = 0.f;
    if (ch >= 0) v = tap < 9 ? w[ch * 9 + (flip ? 8 - tap : tap)] : (bias != nullptr ? bias[ch] : 0.f);
    out[e] = v;
}
__global__ void __launch_bounds__(256)
pk_prep_dw_kernel(const float* __restrict__ w, const float* __restrict__ bias, int C, int seg, int seg_pad, int Cb,
                  float* __restrict__ out, int flip) {
    pk_prep_dw_body(blockIdx.x, w, bias, C, seg, seg_pad, Cb, out, flip);
}

// Several preparations in one launch (all the weights of a block): grid = (blocks of the largest job, jobs).
__global__ void __launch_bounds__(256)
pk_prep_jobs_kernel(const PkPrepJobs jobs) {
    const PkPrepJob& j = jobs.j[blockIdx.y];
    const int bx = blockIdx.x;
    if (bx >= j.blocks) return;
    if (j.kind == 0) pk_prep_a_body(bx, 0, j.w, 0, j.ldw, j.gamma, j.M, j.K, j.seg, j.seg_pad, j.ntiles, j.KS, j.frag, 0);
    else if (j.kind == 1) pk_prep_sc_body(bx, j.w, j.ldw, j.gamma, j.beta, j.bias, j.M, j.K, j.seg, j.seg_pad, j.npos, j.s, j.c);
    else pk_prep_dw_body(bx, j.w, j.bias, j.M, j.seg, j.seg_pad, j.Cb, j.out, 0);
}

int pk_prep_a_launch(const float* w, int64_t w_bs, int ldw, const float* gamma, int M, int K, int seg, int seg_pad,
                     int ntiles, int KS, bf16_t* frag, int64_t frag_bs, int nb, hipStream_t stream) {
    const int64_t n = (int64_t)ntiles * KS * 64;
    hipLaunchKernelGGL(pk_prep_a_kernel, dim3((unsigned)cdiv(n, 256), nb), dim3(256), 0, stream, w, w_bs, ldw, gamma, M, K,
                       seg, seg_pad, ntiles, KS, frag, frag_bs);
    RST_LAUNCH_CHECK();
    return RST_OK;
}
int pk_prep_sc_launch(const float* w, int ldw, const float* gamma, const float* beta, const float* bias, int M, int K,
                      int seg, int seg_pad, int npos, float* s, float* c, hipStream_t stream) {
    hipLaunchKernelGGL(pk_prep_sc_kernel, dim3((unsigned)cdiv(npos, 4)), dim3(256), 0, stream, w, ldw, gamma, beta, bias,
                       M, K, seg, seg_pad, npos, s, c);
    RST_LAUNCH_CHECK();
    return RST_OK;
}
void pk_prep_jobs_add_weight(PkPrepJobs& jobs, const float* src, int K, const float* gamma, const float* beta, const float* bias,
                             int M, int seg, int seg_pad, int ntiles, int KS, int npos, bf16_t* frag, float* s, float* c) {
    PkPrepJob a{};
    a.kind = 0; a.w = src; a.ldw = K; a.gamma = gamma; a.M = M; a.K = K; a.seg = seg; a.seg_pad = seg_pad;
    a.ntiles = ntiles; a.KS = KS; a.frag = frag; a.blocks = (int)cdiv((int64_t)ntiles * KS * 64, 256);
    jobs.j[jobs.n++] = a;
    PkPrepJob b = a;
    b.kind = 1; b.beta = beta; b.bias = bias; b.npos = npos; b.s = s; b.c = c; b.blocks = (int)cdiv(npos, 4);
    jobs.j[jobs.n++] = b;
}
void pk_prep_jobs_add_dw(PkPrepJobs& jobs, const float* w, const float* bias, int C, int seg, int seg_pad, int Cb, float* out) {
    PkPrepJob a{};
    a.kind = 2; a.w = w; a.bias = bias; a.M = C; a.seg = seg; a.seg_pad = seg_pad; a.Cb = Cb; a.out = out;
    a.blocks = (int)cdiv(Cb * 80, 256);
    jobs.j[jobs.n++] = a;
}
int pk_prep_jobs_launch(const PkPrepJobs& jobs, hipStream_t stream) {
    RST_REQUIRE(jobs.n > 0 && jobs.n <= PkPrepJobs::MAX, "weight preparation: %d jobs", jobs.n);
    int most = 1;
    for (int i = 0; i < jobs.n; ++i) most = jobs.j[i].blocks > most ? jobs.j[i].blocks : most;
    hipLaunchKernelGGL(pk_prep_jobs_kernel, dim3((unsigned)most, (unsigned)jobs.n), dim3(256), 0, stream, jobs);
    RST_LAUNCH_CHECK();
    return RST_OK;
}
int pk_prep_dw_launch(const float* w, const float* bias, int C, int seg, int seg_pad, int Cb, float* out, hipStream_t stream,
                      int flip) {
    hipLaunchKernelGGL(pk_prep_dw_kernel, dim3((unsigned)cdiv(Cb * 80, 256)), dim3(256), 0, stream, w, bias, C, seg, seg_pad, Cb, out, flip);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// ---- 1x1 convolution ------------------------------------------------------------------------------------
// A block keeps the weight fragments of `pb` pairs of 16-row tiles in LDS (fragment order: a wave reads 1 KB
// contiguous per ds_read_b128, conflict-free) and walks over a range of pixels; each wave takes every fourth
// tile of 16*NT pixels.  The pixel operand of a tile is fetched once, straight from memory into MFMA operand
// registers (the next tile's is in flight meanwhile), and multiplied with one pair of row tiles after the
// other: 8 output channels x NT pixels per lane and pair, stored as 16-byte chunks.
// grid = (pixel chunks, row splits, samples)
template <int KS, int NT, bool LN>
__global__ void __launch_bounds__(256)
gemm_pk_kernel(const GemmPk g) {
    constexpr int TILE = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int pb = g.pb;
    bf16x8* As = reinterpret_cast<bf16x8*>(smem);                        // [pb*2][KS][64]
    float* sc = reinterpret_cast<float*>(smem + (size_t)pb * 2 * KS * 1024);   // [2][pb*32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_ = lane & 15, gq = lane >> 4;
    const int msplit = blockIdx.y, b = blockIdx.z;
    const bool has_c = g.ln_c != nullptr;
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(g.A) + (int64_t)b * g.a_bs + (int64_t)msplit * pb * 2 * KS * 64;
        u32x4* dst = reinterpret_cast<u32x4*>(smem);
        for (int i = tid; i < pb * 2 * KS * 64; i += 256) dst[i] = src[i];
        if (has_c)
            for (int i = tid; i < pb * 32; i += 256) {
                sc[i] = g.ln_s != nullptr ? g.ln_s[msplit * pb * 32 + i] : 0.f;
                sc[pb * 32 + i] = g.ln_c[msplit * pb * 32 + i];
            }
    }
    __syncthreads();

    const int64_t N = g.N;
    const bf16x8* Xp = reinterpret_cast<const bf16x8*>(g.X + (int64_t)b * g.x_bs);
    u32x4* Yp = reinterpret_cast<u32x4*>(g.Y + (int64_t)b * g.y_bs);
    const u32x4* Rp = g.R != nullptr ? reinterpret_cast<const u32x4*>(g.R + (int64_t)b * g.r_bs) : nullptr;
    const int64_t p0 = (int64_t)blockIdx.x * g.chunk_px;
    const int64_t p1 = p0 + g.chunk_px < N ? p0 + g.chunk_px : N;
    const float invK = 1.0f / (float)g.K;
    const int Kb = g.Kb, Mb = g.Mb;

    // no branch around a load: out-of-range lanes read a valid address and drop the value
    auto load = [&](bf16x8 (&dst)[KS][NT], int64_t n0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int cb = 4 * ks + gq;
                const int64_t px = n0 + nt * 16 + n_;
                const bool ok = cb < Kb && px < p1;
                const bf16x8 v = Xp[(int64_t)(cb < Kb ? cb : 0) * N + (px < p1 ? px : p1 - 1)];
                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                dst[ks][nt] = ok ? v : z;
            }
    };
    auto body = [&](const bf16x8 (&x)[KS][NT], int64_t n0) {
        float mean[NT], rstd[NT];
        if (LN) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    float v[8];
                    unpack8(__builtin_bit_cast(u32x4, x[ks][nt]), v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s1 += v[j]; s2 = fmaf(v[j], v[j], s2); }
                }
                s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                const float m = s1 * invK;
                const float var = fmaxf(s2 * invK - m * m, 0.f);
                mean[nt] = g.ln_mode == 1 ? m : 0.f;
                rstd[nt] = 1.0f / sqrtf(var + 1e-5f);
            }
        }
        for (int p = 0; p < pb; ++p) {
            const int cb = (msplit * pb + p) * 4 + gq;
            const bool cb_ok = cb < Mb;
            u32x4 rres[NT];
            if (Rp != nullptr) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int64_t px = n0 + nt * 16 + n_;
                    rres[nt] = Rp[(int64_t)(cb_ok ? cb : 0) * N + (px < p1 ? px : p1 - 1)];
                }
            }
            f32x4 acc[2][NT];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[h][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const bf16x8* Ap = As + (size_t)p * 2 * KS * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a0 = Ap[ks * 64], a1 = Ap[(KS + ks) * 64];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, x[ks][nt], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x[ks][nt], acc[1][nt], 0, 0, 0);
                }
            }
            float s8[8], c8[8];
            if (has_c) {
                const f32x4* sp = reinterpret_cast<const f32x4*>(sc + p * 32 + gq * 8);
                const f32x4* cp = reinterpret_cast<const f32x4*>(sc + pb * 32 + p * 32 + gq * 8);
                const f32x4 sa = sp[0], sb = sp[1], ca = cp[0], cc = cp[1];
#pragma unroll
                for (int j = 0; j < 4; ++j) { s8[j] = sa[j]; s8[4 + j] = sb[j]; c8[j] = ca[j]; c8[4 + j] = cc[j]; }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int64_t px = n0 + nt * 16 + n_;
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[0][nt][r]; v[4 + r] = acc[1][nt][r]; }
                if (LN) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaf(fmaf(-mean[nt], s8[j], v[j]), rstd[nt], c8[j]);
                } else if (has_c) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += c8[j];
                }
                if (Rp != nullptr) {
                    float r8[8];
                    unpack8(rres[nt], r8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += r8[j];
                }
                if (cb_ok && px < p1) Yp[(int64_t)cb * N + px] = pack8(v);
            }
        }
    };

    const int64_t first = p0 + (int64_t)wave * TILE;
    if constexpr (KS * NT <= 12) {
        bf16x8 t0[KS][NT], t1[KS][NT];
        load(t0, first);
        for (int64_t n0 = first; n0 < p1; n0 += 8 * TILE) {
            load(t1, n0 + 4 * TILE);
            body(t0, n0);
            if (n0 + 4 * TILE < p1) {
                load(t0, n0 + 8 * TILE);
                body(t1, n0 + 4 * TILE);
            }
        }
    } else {
        // many input channels: one operand buffer, but a weight fragment read from LDS serves more pixel tiles
        bf16x8 t0[KS][NT];
        for (int64_t n0 = first; n0 < p1; n0 += 4 * TILE) {
            load(t0, n0);
            body(t0, n0);
        }
    }
}

int gemm_pk_ks(int Kb) {
    const int KS = (int)cdiv(Kb, 4);
    const int have[] = {1, 2, 3, 4, 6, 8, 12, 15};
    for (int k : have) if (KS <= k) return k;
    return -1;
}
// pairs of 16-row tiles whose fragments a block holds in LDS (at most 56 KB), balanced over the row splits
int gemm_pk_pairs(int Kb, int Mb) {
    const int KS = gemm_pk_ks(Kb), np = (int)cdiv(Mb, 4);
    if (KS <= 0) return 1;
    int cap = 56 / (2 * KS);
    if (cap < 1) cap = 1;
    const int splits = (int)cdiv(np, cap);
    return (int)cdiv(np, splits);
}
bool gemm_pk_supported(int K, int M) { return K > 0 && M > 0 && gemm_pk_ks((int)cdiv(K, 8)) > 0; }

template <int KS, int NT>
static int gemm_pk_go(const GemmPk& g, hipStream_t stream, dim3 grid, size_t lds) {
    if (g.ln_mode != 0)
        hipLaunchKernelGGL((gemm_pk_kernel<KS, NT, true>), grid, dim3(256), lds, stream, g);
    else
        hipLaunchKernelGGL((gemm_pk_kernel<KS, NT, false>), grid, dim3(256), lds, stream, g);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

int gemm_pk_launch(GemmPk g, hipStream_t stream) {
    const int KS = gemm_pk_ks(g.Kb);
    RST_REQUIRE(KS > 0, "bf16 1x1 convolution: %d input channels not supported (at most 480)", g.Kb * 8);
    RST_REQUIRE(aligned16(g.A) && aligned16(g.X) && aligned16(g.Y) && aligned16(g.R), "bf16 1x1 convolution: operands must be 16-byte aligned");
    RST_REQUIRE(g.B <= 65535, "bf16 1x1 convolution: batch %d exceeds 65535", g.B);
    g.pb = gemm_pk_pairs(g.Kb, g.Mb);
    g.ngroups = (int)cdiv(cdiv(g.Mb, 4), g.pb);
    const int NT = KS <= 3 || KS == 6 ? 4 : 2;
    const int tile4 = 4 * 16 * NT;                       // pixels the four waves of a block take per round
    // about two blocks per CU over the chip, each at least two rounds long
    int64_t chunk = round_up(cdiv(g.N * g.B * g.ngroups, 512), tile4);
    if (chunk < 2 * tile4) chunk = 2 * tile4;
    g.chunk_px = chunk;
    g.nchunks = (int)cdiv(g.N, chunk);
    const dim3 grid((unsigned)g.nchunks, (unsigned)g.ngroups, (unsigned)g.B);
    const size_t lds = (size_t)g.pb * 2 * KS * 1024 + (size_t)g.pb * 32 * 2 * sizeof(float);
    RST_REQUIRE(lds <= 64 * 1024, "bf16 1x1 convolution: LDS plan of %zu bytes", lds);
    const double px = (double)g.N * g.B;
    ProfScope prof(stream, g.ln_mode ? "pk_gemm_ln" : "pk_gemm", 2.0 * px * g.Mb * 8 * g.Kb * 8,
                   px * 2.0 * (g.Kb * 8 + g.Mb * 8 * (g.R ? 2 : 1)));
    switch (KS) {
        case 1: return gemm_pk_go<1, 4>(g, stream, grid, lds);
        case 2: return gemm_pk_go<2, 4>(g, stream, grid, lds);
        case 3: return gemm_pk_go<3, 4>(g, stream, grid, lds);
        case 4: return gemm_pk_go<4, 2>(g, stream, grid, lds);
        case 6: return gemm_pk_go<6, 4>(g, stream, grid, lds);
        case 8: return gemm_pk_go<8, 2>(g, stream, grid, lds);
        case 12: return gemm_pk_go<12, 2>(g, stream, grid, lds);
        case 15: return gemm_pk_go<15, 2>(g, stream, grid, lds);
    }
    return fail(RST_ERR_INVALID, "bf16 1x1 convolution: no kernel for KS=%d", KS);
}

// ---- depthwise 3x3 (+ GELU gate) ----------------------------------------------------------------------
// gelu(x) = x * Phi(x), erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7), branch-free
__device__ __forceinline__ float pk_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);
    const float q = p * t * e;
    const float phi = x >= 0.f ? fmaf(-0.5f, q, 1.0f) : 0.5f * q;
    return x * phi;
}

// One thread = two neighbouring pixel columns of a strip of rows x one block of 8 channels.  An input row (four
// 16-byte chunks: x-1 .. x+2, the next row's already in flight) adds to the three output rows it touches; the three
// accumulator sets and the three chunk buffers change roles from row to row (the loop is unrolled by three, nothing
// is copied).  The 72 taps + 8 biases of the channel block are the same for the whole block and live in scalar
// registers.  No branch around a load: out-of-image chunks read a valid address and are zeroed.
// GATE: the block holds 4 channels of x1 and the same 4 channels of x2 (see src_channel); the thread writes
// gelu(d[0..3]) * d[4..7], 8 bytes per pixel, into its half of the output chunk.
// grid = (ceil(W/2 * strips / 256), channel blocks, B)
template <bool GATE, int PX>
__global__ void __launch_bounds__(256)
dw_pk_kernel(const DwPk a, const float* __restrict__ wp) {
    const int cb = blockIdx.y, b = blockIdx.z;
    const float* wl = wp + cb * 80;
    // the taps of the third window row wait in vector registers (all 72 + 8 do not fit the scalar file beside the
    // rest): the zero the compiler cannot see makes these vector loads
    float w2[24];
    {
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
#pragma unroll
        for (int j = 0; j < 24; ++j) w2[j] = wl[48 + j + z];
    }
    const int W2 = a.W / PX;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int strip = t / W2, x0 = (t % W2) * PX;
    const int y0 = strip * a.rows;
    if (y0 >= a.H) return;
    const int y1 = y0 + a.rows < a.H ? y0 + a.rows : a.H;
    const int64_t N = (int64_t)a.H * a.W;
    const u32x4* X = reinterpret_cast<const u32x4*>(a.x + (int64_t)b * a.x_bs) + (int64_t)cb * N;
    bf16_t* Yb = a.y + (int64_t)b * a.y_bs;
    const bool hasl = x0 > 0, hasr = x0 + PX < a.W;
    const int xl = hasl ? x0 - 1 : x0, xr = hasr ? x0 + PX : x0 + PX - 1;
    const u32x4 zero = {0u, 0u, 0u, 0u};

    // raw chunks only: what is outside the image is zeroed where the row is consumed, so that nothing waits for
    // the loads here
    auto fetch = [&](u32x4 (&dst)[PX + 2], int i) {
        const u32x4* row = X + (int64_t)(i < 0 ? 0 : (i < a.H ? i : a.H - 1)) * a.W;
        dst[0] = row[xl];
#pragma unroll
        for (int q = 0; q < PX; ++q) dst[1 + q] = row[x0 + q];
        dst[PX + 1] = row[xr];
    };
    // input row i is window row 0 of output row i+1 (P, started here), row 1 of output row i (Q) and row 2 of
    // output row i-1 (R), which is complete after this step
    auto step = [&](int i, float (&P)[PX][8], float (&Q)[PX][8], float (&R)[PX][8], const u32x4 (&cur)[PX + 2], u32x4 (&nxt)[PX + 2]) {
        fetch(nxt, i + 1 <= y1 ? i + 1 : y1);
        float c[PX + 2][8];
        const bool rv = i >= 0 && i < a.H;
#pragma unroll
        for (int k = 0; k < PX + 2; ++k) {
            const bool ok = rv && (k > 0 || hasl) && (k < PX + 1 || hasr);
            unpack8(ok ? cur[k] : zero, c[k]);
        }
#pragma unroll
        for (int q = 0; q < PX; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                P[q][j] = fmaf(wl[0 + j], c[q][j], fmaf(wl[8 + j], c[q + 1][j], wl[16 + j] * c[q + 2][j]));
                Q[q][j] = fmaf(wl[24 + j], c[q][j], fmaf(wl[32 + j], c[q + 1][j], fmaf(wl[40 + j], c[q + 2][j], Q[q][j])));
                R[q][j] = fmaf(w2[j], c[q][j], fmaf(w2[8 + j], c[q + 1][j], fmaf(w2[16 + j], c[q + 2][j], R[q][j])));
            }
        const int yo = i - 1;
        if (yo >= y0) {
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = R[q][j] + wl[72 + j];
                const int64_t px = (int64_t)yo * a.W + x0 + q;
                if (GATE) {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    u32x2 o;
                    o[0] = pack_bf16(pk_gelu(v[0]) * v[4], pk_gelu(v[1]) * v[5]);
                    o[1] = pack_bf16(pk_gelu(v[2]) * v[6], pk_gelu(v[3]) * v[7]);
                    reinterpret_cast<u32x2*>(Yb)[((int64_t)(cb >> 1) * N + px) * 2 + (cb & 1)] = o;
                } else {
                    reinterpret_cast<u32x4*>(Yb)[(int64_t)cb * N + px] = pack8(v);
                }
            }
        }
    };
    float A[PX][8], Bq[PX][8], C[PX][8];
#pragma unroll
    for (int q = 0; q < PX; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) { A[q][j] = 0.f; Bq[q][j] = 0.f; C[q][j] = 0.f; }
    u32x4 X0[PX + 2], X1[PX + 2], X2[PX + 2];
    fetch(X0, y0 - 1);
    for (int i = y0 - 1; i <= y1; i += 3) {
        step(i, A, Bq, C, X0, X1);
        if (i + 1 <= y1) step(i + 1, C, A, Bq, X1, X2);
        if (i + 2 <= y1) step(i + 2, Bq, C, A, X2, X0);
    }
}

// The same convolution on the matrix pipe.  A 16-byte chunk of a neighbouring pixel is already an MFMA B operand
// (8 consecutive k of pixel n), so the k axis is made of tap slots: a wave takes 16 columns x 2 output rows of one
// channel block, its 4 input rows x 3 column offsets are 12 slots = 3 k-steps x 4 lane groups, and lane (n, g) of
// k-step s loads the chunk of slot 4s + g for its column -- three loads, no unpacking, no vector multiply.  The A
// operand is the tap table as a sparse 16 x 96 matrix built once per wave: rows 0-7 are the 8 channels of the upper
// output row, rows 8-15 of the lower one, and row m has one non-zero per slot, the tap of its channel (bf16).
// The result lands 4 rows per lane: (GATE) the rows are ordered x1_0 x2_0 x1_1 x2_1 | ..., so a lane holds both
// factors of two gated outputs and stores 4 bytes; (plain) 4 channels, 8 bytes.
// grid = (ceil(W/64) * row ranges, channel blocks, B), one wave per 16-column strip, `rows` rows per range.
template <bool GATE>
__global__ void __launch_bounds__(256)
dwm_pk_kernel(const DwPk a, const float* __restrict__ wp) {
    const int cb = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_ = lane & 15, gq = lane >> 4;
    const float* wl = wp + cb * 80;
    const int strips64 = (a.W + 63) / 64;
    const int range = blockIdx.x / strips64;
    const int xs = (blockIdx.x % strips64) * 64 + wave * 16;        // first column of this wave's strip
    const int y_begin = range * a.rows;
    if (xs >= a.W || y_begin >= a.H) return;
    const int y_end = y_begin + a.rows < a.H ? y_begin + a.rows : a.H;
    const int64_t N = (int64_t)a.H * a.W;
    const u32x4* X = reinterpret_cast<const u32x4*>(a.x + (int64_t)b * a.x_bs) + (int64_t)cb * N;
    char* Yb = reinterpret_cast<char*>(a.y + (int64_t)b * a.y_bs);
    // position in the channel block of accumulator row q (0..7) of either output row
    auto pos_of = [](int q) { return GATE ? ((q & 1) ? 4 + (q >> 1) : (q >> 1)) : q; };
    // A fragments: this lane is matrix row m = n_, k-group gq
    bf16x8 af[3];
    {
        const int m = n_, orow = m >> 3, p = pos_of(m & 7);
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            const int t = 4 * s3 + gq, i = t / 3, d = t % 3, dy = i - orow;
            float w = 0.f;
            if (dy >= 0 && dy <= 2) w = wl[(dy * 3 + d) * 8 + p];
            const uint32_t bits = to_bf16_bits(w);
            u32x4 f = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((p >> 1) == k) f[k] = (p & 1) ? (bits << 16) : bits;
            af[s3] = __builtin_bit_cast(bf16x8, f);
        }
    }
    // epilogue constants of this lane's 4 accumulator rows
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = wl[72 + pos_of((4 * gq + r) & 7)];
    const int x = xs + n_;
    const bool x_ok = x < a.W;
    // slot of this lane at k-step s3: input row offset and column
    int si[3], sx[3]; bool sxok[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int t = 4 * s3 + gq;
        si[s3] = t / 3 - 1;
        const int xx = x + t % 3 - 1;
        sxok[s3] = xx >= 0 && xx < a.W;
        sx[s3] = xx < 0 ? 0 : (xx < a.W ? xx : a.W - 1);
    }
    auto fetch = [&](u32x4 (&dst)[3], int y) {
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            const int yy = y + si[s3];
            dst[s3] = X[(int64_t)(yy < 0 ? 0 : (yy < a.H ? yy : a.H - 1)) * a.W + sx[s3]];
        }
    };
    const u32x4 zero = {0u, 0u, 0u, 0u};
    u32x4 cur[3], nxt[3];
    fetch(cur, y_begin);
    for (int y = y_begin; y < y_end; y += 2) {
        fetch(nxt, y + 2 < y_end ? y + 2 : y);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            const int yy = y + si[s3];
            const bool ok = sxok[s3] && yy >= 0 && yy < a.H;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s3], __builtin_bit_cast(bf16x8, ok ? cur[s3] : zero), acc, 0, 0, 0);
        }
        const int yo = y + (gq >> 1);
        if (x_ok && yo < y_end) {
            const int64_t px = (int64_t)yo * a.W + x;
            if (GATE) {
                const uint32_t o = pack_bf16(pk_gelu(acc[0] + bias[0]) * (acc[1] + bias[1]), pk_gelu(acc[2] + bias[2]) * (acc[3] + bias[3]));
                *reinterpret_cast<uint32_t*>(Yb + (((int64_t)(cb >> 1) * N + px) * 16 + (cb & 1) * 8 + (gq & 1) * 4)) = o;
            } else {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                u32x2 o;
                o[0] = pack_bf16(acc[0] + bias[0], acc[1] + bias[1]);
                o[1] = pack_bf16(acc[2] + bias[2], acc[3] + bias[3]);
                *reinterpret_cast<u32x2*>(Yb + (((int64_t)cb * N + px) * 16 + (gq & 1) * 8)) = o;
            }
        }
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) cur[s3] = nxt[s3];
    }
}

int dw_pk_launch(DwPk a, hipStream_t stream) {
    RST_REQUIRE(a.x && a.wp && a.y, "bf16 depthwise: null operand");
    RST_REQUIRE(aligned16(a.x) && aligned16(a.y), "bf16 depthwise: operands must be 16-byte aligned");
    RST_REQUIRE(a.W % 2 == 0 && a.W >= 2, "bf16 depthwise: width %d must be even", a.W);
    RST_REQUIRE(a.B <= 65535 && a.Cb <= 65535, "bf16 depthwise: batch or channel count exceeds 65535");
    // rows per strip: long strips re-read fewer halo rows; short ones give the chip enough threads
    // RST_DW_MFMA: bit 0 forces the matrix-pipe kernel for the plain, bit 1 for the gated convolution (tests); unset,
    // it takes the gated convolution of small images, where it was measured ahead (16 x 192 x 64^2: 32 vs 38 us,
    // 16 x 384 x 32^2: 22 vs 29 us; 16 x 96 x 256^2: 251 vs 220 us)
    static const int mfma_mode = getenv("RST_DW_MFMA") ? atoi(getenv("RST_DW_MFMA")) : -1;
    const bool small_gate = mfma_mode < 0 && a.gate && (int64_t)a.H * a.W <= 64 * 64;
    if (small_gate || (mfma_mode > 0 && ((a.gate && (mfma_mode & 2)) || (!a.gate && (mfma_mode & 1))))) {
        a.rows = 32;
        const dim3 grid((unsigned)(cdiv(a.W, 64) * cdiv(a.H, a.rows)), a.Cb, a.B);
        const double el = (double)a.B * a.Cb * 8 * a.H * a.W;
        ProfScope prof(stream, a.gate ? "pk_dw_gate" : "pk_dw", el * 18.0, el * 2.0 * (a.gate ? 1.5 : 2.0));
        if (a.gate) hipLaunchKernelGGL(dwm_pk_kernel<true>, grid, dim3(256), 0, stream, a, a.wp);
        else hipLaunchKernelGGL(dwm_pk_kernel<false>, grid, dim3(256), 0, stream, a, a.wp);
        RST_LAUNCH_CHECK();
        return RST_OK;
    }
    // measured on 16 x 96 x 256 x 256: plain 1 pixel per thread / 8-row strips 255 us, 2 pixels / 16 rows 274 us;
    // gated 237 us and 222 us
    const int px = a.gate ? 2 : 1;
    a.rows = a.gate ? 16 : 8;
    while (a.rows > 4 && (int64_t)a.B * a.Cb * cdiv(a.H, a.rows) * (a.W / px) < 256 * 256 * 4) a.rows >>= 1;
    if (a.rows > a.H) a.rows = a.H;
    const int strips = (int)cdiv(a.H, a.rows);
    const dim3 grid((unsigned)cdiv((int64_t)(a.W / px) * strips, 256), a.Cb, a.B);
    const double el = (double)a.B * a.Cb * 8 * a.H * a.W;
    if (a.gate) {
        ProfScope prof(stream, "pk_dw_gate", el * 25.0, el * 2.0 * 1.5);
        if (px == 1) hipLaunchKernelGGL((dw_pk_kernel<true, 1>), grid, dim3(256), 0, stream, a, a.wp);
        else hipLaunchKernelGGL((dw_pk_kernel<true, 2>), grid, dim3(256), 0, stream, a, a.wp);
    } else {
        ProfScope prof(stream, "pk_dw", el * 18.0, el * 2.0 * 2.0);
        if (px == 1) hipLaunchKernelGGL((dw_pk_kernel<false, 1>), grid, dim3(256), 0, stream, a, a.wp);
        else hipLaunchKernelGGL((dw_pk_kernel<false, 2>), grid, dim3(256), 0, stream, a, a.wp);
    }
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// ---- fused 1x1 convolution (folded LayerNorm) -> depthwise 3x3 (-> GELU gate) -----------------------------
// The 1x1 result never goes to memory, and not to LDS either.  A wave takes a patch of 16 columns x 8 rows of
// pixels: tile nt of the matrix product is row nt of the patch, so a lane (n, g) ends up holding, for one pair of
// row tiles, 8 channels of column n in ALL 8 rows -- the vertical taps of the depthwise convolution are other
// registers of the same lane, the horizontal ones the neighbouring lanes of its group of 16 (DPP row shifts).
// The inner 14 x (NT-2) pixels of the patch are complete and are stored; patches overlap by the halo (1.5-1.7 x the
// matrix work and pixel-operand reads, which are the small side of this operation; NT = 8 rows, 6 at K = 96 where
// the patch's operand would not leave room for two waves per SIMD).  Pixels outside the image are
// zeroed after the LayerNorm epilogue (the depthwise convolution pads its INPUT with zeros).  The pixel operand
// of the patch stays in registers for all row-tile pairs; the halves of a pair (4 channels each) are multiplied and
// convolved one after the other to bound the registers.  Weight fragments and epilogue constants sit in LDS as in
// gemm_pk_kernel; the depthwise taps come from memory per pair (a few KB, cache resident).
// grid = (patch groups, row splits, samples); only K <= 96 (KS <= 3).
template <int KS, int NT, bool GATE>
__global__ void __launch_bounds__(256, 2)
pwdw_pk_kernel(const PwDwPk g) {
    constexpr int RO = NT - 2;                            // complete (stored) rows of a patch
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int pb = g.pb;
    bf16x8* As = reinterpret_cast<bf16x8*>(smem);
    float* sc = reinterpret_cast<float*>(smem + (size_t)pb * 2 * KS * 1024);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_ = lane & 15, gq = lane >> 4;
    const int msplit = blockIdx.y, b = blockIdx.z;
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(g.A) + (int64_t)msplit * pb * 2 * KS * 64;
        u32x4* dst = reinterpret_cast<u32x4*>(smem);
        for (int i = tid; i < pb * 2 * KS * 64; i += 256) dst[i] = src[i];
        for (int i = tid; i < pb * 32; i += 256) {
            sc[i] = g.ln_s[msplit * pb * 32 + i];
            sc[pb * 32 + i] = g.ln_c[msplit * pb * 32 + i];
        }
    }
    __syncthreads();
    const int H = g.H, W = g.W;
    const int64_t N = (int64_t)H * W;
    const bf16x8* Xp = reinterpret_cast<const bf16x8*>(g.X + (int64_t)b * g.x_bs);
    bf16_t* Yb = g.Y + (int64_t)b * g.y_bs;
    const float invK = 1.0f / (float)g.K;
    const int Kb = g.Kb, Mb = g.Mb;
    const int pxn = (W + 13) / 14, pyn = (H + RO - 1) / RO;
    const int npatch = pxn * pyn;
    const int first = blockIdx.x * g.patches_per_block;
    const int last = first + g.patches_per_block < npatch ? first + g.patches_per_block : npatch;

    for (int patch = first + wave; patch < last; patch += 4) {
        const int ty = patch / pxn, tx = patch % pxn;
        const int x = tx * 14 - 1 + n_, y0 = ty * RO - 1;
        const int xc = x < 0 ? 0 : (x < W ? x : W - 1);
        const bool x_in = x >= 0 && x < W;
        bf16x8 xf[KS][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int y = y0 + nt;
            const int yc = y < 0 ? 0 : (y < H ? y : H - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cb = 4 * ks + gq;
                const bf16x8 v = Xp[(int64_t)(cb < Kb ? cb : 0) * N + (int64_t)yc * W + xc];
                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                xf[ks][nt] = cb < Kb ? v : z;
            }
        }
        float mean[NT], rstd[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float v[8];
                unpack8(__builtin_bit_cast(u32x4, xf[ks][nt]), v);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s1 += v[j]; s2 = fmaf(v[j], v[j], s2); }
            }
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            const float m = s1 * invK;
            const float var = fmaxf(s2 * invK - m * m, 0.f);
            mean[nt] = g.ln_mode == 1 ? m : 0.f;
            rstd[nt] = 1.0f / sqrtf(var + 1e-5f);
        }
        for (int p = 0; p < pb; ++p) {
            const int cb = (msplit * pb + p) * 4 + gq;
            const int cbc = cb < Mb ? cb : 0;
            const bf16x8* Ap = As + (size_t)p * 2 * KS * 64 + lane;
            float o0[RO][4];                                   // the first half's results wait for the second
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                f32x4 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 a = Ap[(h * KS + ks) * 64];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xf[ks][nt], acc[nt], 0, 0, 0);
                }
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc + p * 32 + gq * 8 + 4 * h);
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(sc + pb * 32 + p * 32 + gq * 8 + 4 * h);
                // taps of this lane's 4 channels: [tap][4], then the bias
                f32x4 wt[9];
                const float* wsrc = g.dww + cbc * 80 + 4 * h;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) wt[tp] = *reinterpret_cast<const f32x4*>(wsrc + tp * 8);
                const f32x4 bias = *reinterpret_cast<const f32x4*>(wsrc + 72);
                float o[RO][4];
#pragma unroll
                for (int r = 0; r < RO; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[r][i] = bias[i];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int y = y0 + nt;
                    const bool in = x_in && y >= 0 && y < H;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float m = fmaf(fmaf(-mean[nt], s4[i], acc[nt][i]), rstd[nt], c4[i]);
                        m = in ? m : 0.f;
                        const float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x111, 0xf, 0xf, true));
                        const float rr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x101, 0xf, 0xf, true));
                        // patch row nt is window row dy of output row nt + 1 - dy (stored rows: 1..6)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int orow = nt + 1 - dy;
                            if (orow >= 1 && orow <= RO)
                                o[orow - 1][i] = fmaf(wt[dy * 3][i], l, fmaf(wt[dy * 3 + 1][i], m, fmaf(wt[dy * 3 + 2][i], rr, o[orow - 1][i])));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);        // one patch row at a time: keeps the live values few
                }
                if (h == 0) {
#pragma unroll
                    for (int r = 0; r < RO; ++r)
#pragma unroll
                        for (int i = 0; i < 4; ++i) o0[r][i] = o[r][i];
                } else {
                    const bool col_ok = n_ >= 1 && n_ <= 14 && x < W && cb < Mb;
#pragma unroll
                    for (int r = 0; r < RO; ++r) {
                        const int y = y0 + 1 + r;
                        if (!(col_ok && y < H)) continue;
                        const int64_t px = (int64_t)y * W + x;
                        if (GATE) {
                            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                            u32x2 w2;
                            w2[0] = pack_bf16(pk_gelu(o0[r][0]) * o[r][0], pk_gelu(o0[r][1]) * o[r][1]);
                            w2[1] = pack_bf16(pk_gelu(o0[r][2]) * o[r][2], pk_gelu(o0[r][3]) * o[r][3]);
                            reinterpret_cast<u32x2*>(Yb)[((int64_t)(cb >> 1) * N + px) * 2 + (cb & 1)] = w2;
                        } else {
                            u32x4 w4;
                            w4[0] = pack_bf16(o0[r][0], o0[r][1]); w4[1] = pack_bf16(o0[r][2], o0[r][3]);
                            w4[2] = pack_bf16(o[r][0], o[r][1]);   w4[3] = pack_bf16(o[r][2], o[r][3]);
                            reinterpret_cast<u32x4*>(Yb)[(int64_t)cb * N + px] = w4;
                        }
                    }
                }
            }
        }
    }
}

// ---- fused 1x1 (folded LayerNorm) -> depthwise 3x3 (-> gate), both on the matrix pipe ----------------------
// The work is cut into items (strip of 32 columns: 30 stored + the halo column on either side) x (range of rows) x
// (ONE pair of row tiles of the 1x1 product = 4 channel blocks of the intermediate).  A wave takes one item and walks
// down its rows on its own:
//   G (row r): the 1x1 product of the row for the wave's pair (weights resident in registers, the pixel operand
//     straight from memory, the next rows' already in flight); the LayerNorm statistics of the row's pixels on the
//     matrix pipe as well (ones^T X: channel sums in every accumulator row; X^T X: sums of squares on the diagonal,
//     handed to the other lane groups with one ds_bpermute -- bf16 products and fp32 sums, exact); LayerNorm
//     epilogue, zero outside the image, packed to bf16 chunks into the wave's PRIVATE ring of 4 rows in LDS --
//     written exactly as the depthwise convolution's B operand wants them;
//   D (rows y, y+1): the depthwise convolution as in dwm_pk_kernel, its operand chunks read from the ring with
//     ds_read_b128; the tap matrices have one non-zero per lane and k-step.
// A wave is the only producer and the only consumer of its ring and LDS operations of a wave complete in order, so
// the kernel has NO barrier and the waves share nothing: they drift apart and fill each other's latencies.  The
// intermediate never leaves the CU, nothing is unpacked, and the vector unit only runs the two epilogues.
// The vector unit is what bounds the loop, so it carries no address arithmetic: loads and stores go through buffer
// descriptors (lane-constant offset + one scalar or vector row offset, out-of-range lanes dropped by the
// descriptor instead of a branch), the row range starts at a multiple of 4 and the loop is unrolled by two, so that
// every ring slot is a compile-time constant.
// grid = (blocks of 4 items, 1, samples); blocks 8 apart share an XCD (and its L2) and take neighbouring items, so
// that the pairs of one strip read its input from memory once.  g.rows % 4 == 0.
template <int KS, bool GATE>
__global__ void __launch_bounds__(256, 2)
pwdw_ring_kernel(const PwDwPk g) {
    constexpr int TW = 32, SLOT = 4 * TW * 16;                       // bytes of a ring row (4 channel blocks)
    __shared__ __attribute__((aligned(16))) char ring_all[4][4 * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // (uniform: the descriptors are built from it)
    const int n_ = lane & 15, gq = lane >> 4;
    const int b = blockIdx.z;
    const int Mb = g.Mb, npairs = (Mb + 3) / 4;
    const int H = g.H, W = g.W, Kb = g.Kb;
    const int64_t N = (int64_t)H * W;
    const int nstrips = (W + 29) / 30;
    const int ntiles = nstrips * ((H + g.rows - 1) / g.rows);
    // item of this wave: the blocks of an XCD (ids 8 apart) walk through a contiguous eighth of the items
    const int per_xcd = gridDim.x >> 3;
    const int vblock = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int item = vblock * 4 + wave;
    if (item >= ntiles * npairs) return;                             // (no barrier anywhere)
    const int tile = item / npairs, pair = item % npairs;
    const int strip = tile % nstrips, range = tile / nstrips;
    const int x0 = strip * 30 - 1;                                   // image column of strip column 0
    const int y_begin = range * g.rows;
    const int y_end = y_begin + g.rows < H ? y_begin + g.rows : H;
    const int r0 = y_begin - 1;
    char* Yb = reinterpret_cast<char*>(g.Y + (int64_t)b * g.y_bs);
    char* myring = ring_all[wave];
    {
        const u32x4 z = {0u, 0u, 0u, 0u};                           // rows a short range never writes must stay finite
#pragma unroll
        for (int i = 0; i < 8; ++i) reinterpret_cast<u32x4*>(myring)[i * 64 + lane] = z;
    }
    // the sample's input through a buffer descriptor: lane-constant offset of (channel block, clamped column), the
    // clamped row as the scalar offset.  Channel blocks past Kb read block 0: the weights of those positions are zero,
    // so the product ignores them and nothing has to wait for the load to mask it (the statistics mask them).
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(g.X + (int64_t)b * g.x_bs), 0, (int)__builtin_amdgcn_readfirstlane((int)((int64_t)Kb * N * 16)), 0x00020000);
    bool x_in[2];
    unsigned xoff[KS][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int x = x0 + 16 * nt + n_;
        const int xc = x < 0 ? 0 : (x < W ? x : W - 1);
        x_in[nt] = x >= 0 && x < W;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cb = 4 * ks + gq;
            xoff[ks][nt] = (unsigned)(((int64_t)(cb < Kb ? cb : 0) * N + xc) * 16);
        }
    }
    auto xload = [&](u32x4 (&dst)[KS][2], int ry) {
        const int yc = ry < 0 ? 0 : (ry < H ? ry : H - 1);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dst[ks][nt] = __builtin_amdgcn_raw_buffer_load_b128(xr, xoff[ks][nt], yc * W * 16, 0);
    };
    u32x4 xa[KS][2], xb[KS][2];
    xload(xa, r0); xload(xb, r0 + 1);                                // in flight during the set-up below

    auto pos_of = [](int q) { return GATE ? ((q & 1) ? 4 + (q >> 1) : (q >> 1)) : q; };
    // weights of the pair: [half][k-step]; LayerNorm constants of the lane's 8 channels
    bf16x8 wa[2][KS];
    {
        const bf16x8* Ap = reinterpret_cast<const bf16x8*>(g.A) + (size_t)pair * 2 * KS * 64 + lane;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wa[h][ks] = Ap[(h * KS + ks) * 64];
    }
    const f32x4 sa = *reinterpret_cast<const f32x4*>(g.ln_s + pair * 32 + gq * 8), sb = *reinterpret_cast<const f32x4*>(g.ln_s + pair * 32 + gq * 8 + 4);
    const f32x4 ca = *reinterpret_cast<const f32x4*>(g.ln_c + pair * 32 + gq * 8), cc4 = *reinterpret_cast<const f32x4*>(g.ln_c + pair * 32 + gq * 8 + 4);
    // taps of the wave's 4 channel blocks as in dwm_pk_kernel: this lane is matrix row m = n_, k-group gq; its one
    // non-zero of k-step s3 sits in 32-bit word kq of the fragment.  Biases of the lane's 4 accumulator rows.
    const int kq = pos_of(n_ & 7) >> 1;
    bf16x8 af[4][3];
    f32x4 dbias[4];
    __amdgpu_buffer_rsrc_t yr[4];
    bool st_ok[4];
    const int plane_bytes = (int)__builtin_amdgcn_readfirstlane((int)(N * 16));      // (N < 2^27: checked by the launcher)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // (a channel block past Mb -- the last pair may be partial -- is computed on its all-zero ring rows and
        // dropped by the stores: no branch, every path issues the same number of memory operations)
        const bool cb_live = 4 * pair + j < Mb;
        const int cb = cb_live ? 4 * pair + j : Mb - 1;
        const float* wl = g.dww + cb * 80;
        const int orow = n_ >> 3, p = pos_of(n_ & 7);
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            const int t = 4 * s3 + gq, i = t / 3, d = t % 3, dy = i - orow;
            const float wr = wl[((dy < 0 ? 0 : (dy > 2 ? 2 : dy)) * 3 + d) * 8 + p];      // (no branch around the load)
            const uint32_t bits = to_bf16_bits((dy >= 0 && dy <= 2) ? wr : 0.f);
            const uint32_t tv = (p & 1) ? (bits << 16) : bits;
            u32x4 f;
#pragma unroll
            for (int k = 0; k < 4; ++k) f[k] = kq == k ? tv : 0u;
            af[j][s3] = __builtin_bit_cast(bf16x8, f);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dbias[j][r] = wl[72 + pos_of((4 * gq + r) & 7)];
        // stores through a buffer descriptor over the channel block's plane: lanes outside the strip, the image or
        // the row range get an offset past its end and are dropped
        yr[j] = __builtin_amdgcn_make_buffer_rsrc(Yb + (int64_t)(GATE ? cb >> 1 : cb) * N * 16, 0, plane_bytes, 0x00020000);
        {
            const int col = 16 * (gq & 1) + n_;
            st_ok[j] = cb_live && col >= 1 && col <= 30 && x0 + col < W;
        }
    }
    // after the exchange below a lane stores the whole chunk of strip column 16 (gq & 1) + n_ in row y + (gq >> 1)
    const unsigned st_col = (unsigned)(x0 + 16 * (gq & 1) + n_) * 16u;
    // LDS addresses of this lane's operand chunks: k-step s3 -> slot 4 s3 + gq -> (row offset, column); the ring row
    // of image row y + dr is (y + dr) & 3 and y % 4 is 0 (PAR 0) or 2 (PAR 1)
    const char* dsrc[2][3][2];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int t = 4 * s3 + gq, dr = t / 3 - 1;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = 16 * nt + n_ + t % 3 - 1;
            const int cc = c < 0 ? 0 : (c < TW ? c : TW - 1);          // only the unstored halo columns clamp
#pragma unroll
            for (int par = 0; par < 2; ++par) dsrc[par][s3][nt] = myring + ((2 * par + dr) & 3) * SLOT + cc * 16;
        }
    }
    char* gdst = myring + gq * (TW * 16) + n_ * 16;                  // + slot * SLOT + nt * 256
    const bool cb_ok = 4 * pair + gq < Mb;
    const float invK = 1.0f / (float)g.K;
    const bool with_mean = g.ln_mode == 1;
    const u32x4 ones_w = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_w);
    const int diag_lane = ((n_ >> 2) * 16 + n_) * 4;                 // byte index of the lane that holds pixel n_'s diagonal entry
    const int rsel = n_ & 3;
    const bool kmask = Kb % 4 != 0;                                  // the last k-step has channel blocks past Kb

    // G: row ry of the intermediate for the wave's 4 channel blocks -> ring row `slot`
    auto gemm_row = [&](const u32x4 (&xf)[KS][2], int ry, int slot) {
        f32x4 acc[2][2], a1[2], a2[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            acc[0][nt] = acc[1][nt] = a1[nt] = a2[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const bf16x8 xv = __builtin_bit_cast(bf16x8, xf[ks][nt]);
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][ks], xv, acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1][ks], xv, acc[1][nt], 0, 0, 0);
                const u32x4 z = {0u, 0u, 0u, 0u};
                const bf16x8 xm = (ks == KS - 1 && kmask) ? __builtin_bit_cast(bf16x8, 4 * ks + gq < Kb ? xf[ks][nt] : z) : xv;
                a1[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xm, a1[nt], 0, 0, 0);
                a2[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, xm, a2[nt], 0, 0, 0);
            }
        const bool row_in = ry >= 0 && ry < H;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float dsel = rsel == 0 ? a2[nt][0] : rsel == 1 ? a2[nt][1] : rsel == 2 ? a2[nt][2] : a2[nt][3];
            const float s2 = __int_as_float(__builtin_amdgcn_ds_bpermute(diag_lane, __float_as_int(dsel)));
            const float m = a1[nt][0] * invK;
            const float var = fmaxf(s2 * invK - m * m, 0.f);
            const float rstd = __builtin_amdgcn_rsqf(var + 1e-5f);       // (1 ulp; the result is rounded to bf16)
            const float nm = with_mean ? -m : 0.f;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = fmaf(fmaf(nm, sa[r], acc[0][nt][r]), rstd, ca[r]);
                v[4 + r] = fmaf(fmaf(nm, sb[r], acc[1][nt][r]), rstd, cc4[r]);
            }
            const u32x4 z = {0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(gdst + slot * SLOT + nt * 256) = (cb_ok && row_in && x_in[nt]) ? pack8(v) : z;
        }
    };
    // D: output rows y, y + 1 (the ring holds rows y-1 .. y+2)
    auto dw_rows = [&](int y, auto Par) {
        constexpr int PAR = decltype(Par)::value;
        const int yo = y + (gq >> 1);
        const unsigned rowoff = (unsigned)(yo * W) * 16u;
        const bool row_ok = yo < y_end;
        // the operand chunks of channel block j + 1 are requested before the products of block j start (the scheduling
        // barrier keeps the compiler from sinking each read next to its use, where every product would wait out the
        // LDS latency)
        u32x4 bw[2][3][2];
        auto fetch = [&](u32x4 (&dst)[3][2], int j) {
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)      // (read with the type the ring is written with: the compiler must see that these alias)
                    dst[s3][nt] = *reinterpret_cast<const u32x4*>(dsrc[PAR][s3][nt] + j * (TW * 16));
        };
        uint32_t gate_lo[2], gate_hi[2];
        fetch(bw[0], 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 3) fetch(bw[(j + 1) & 1], j + 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j][s3], __builtin_bit_cast(bf16x8, bw[j & 1][s3][nt]), acc[nt], 0, 0, 0);
            // The lane holds 4 channels (8 bytes; gated: 2 products, 4 bytes) of pixel n_ in both halves of the strip;
            // lane groups 2i and 2i+1 hold the two halves of the same chunks.  v_permlane16_swap hands the odd group's
            // share of the left half to the even group and the even group's share of the right half to the odd one:
            // every lane then stores one whole 16-byte chunk (gated: the 8 bytes of its channel block, and the two
            // blocks of a chunk are the same lane's j and j + 1) -- half (a quarter) as many, wider stores.
            if (GATE) {
                const uint32_t o0 = pack_bf16(pk_gelu(acc[0][0] + dbias[j][0]) * (acc[0][1] + dbias[j][1]),
                                              pk_gelu(acc[0][2] + dbias[j][2]) * (acc[0][3] + dbias[j][3]));
                const uint32_t o1 = pack_bf16(pk_gelu(acc[1][0] + dbias[j][0]) * (acc[1][1] + dbias[j][1]),
                                              pk_gelu(acc[1][2] + dbias[j][2]) * (acc[1][3] + dbias[j][3]));
                const auto sw = __builtin_amdgcn_permlane16_swap(o0, o1, false, false);
                gate_lo[j & 1] = sw[0]; gate_hi[j & 1] = sw[1];
                if (j & 1) {
                    // (the gated product has an even number of channel blocks: both blocks of a chunk are live together)
                    const unsigned vo = (st_ok[j - 1] && row_ok) ? rowoff + st_col : 0x80000000u;
                    const u32x4 o = {gate_lo[0], gate_hi[0], gate_lo[1], gate_hi[1]};
                    __builtin_amdgcn_raw_buffer_store_b128(o, yr[j - 1], vo, 0, 0);
                }
            } else {
                const uint32_t a0 = pack_bf16(acc[0][0] + dbias[j][0], acc[0][1] + dbias[j][1]);
                const uint32_t a1 = pack_bf16(acc[0][2] + dbias[j][2], acc[0][3] + dbias[j][3]);
                const uint32_t b0 = pack_bf16(acc[1][0] + dbias[j][0], acc[1][1] + dbias[j][1]);
                const uint32_t b1 = pack_bf16(acc[1][2] + dbias[j][2], acc[1][3] + dbias[j][3]);
                const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                const unsigned vo = (st_ok[j] && row_ok) ? rowoff + st_col : 0x80000000u;
                const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                __builtin_amdgcn_raw_buffer_store_b128(o, yr[j], vo, 0, 0);
            }
        }
    };

    gemm_row(xa, r0, 3); gemm_row(xb, r0 + 1, 0);
    xload(xa, r0 + 2); xload(xb, r0 + 3);
    gemm_row(xa, r0 + 2, 1); gemm_row(xb, r0 + 3, 2);               // (a row past y_end is past the image: written as zeros)
    xload(xa, y_begin + 3); xload(xb, y_begin + 4);                  // (clamped rows past the range are never used)
    auto step = [&](int y, auto Par) {
        constexpr int PAR = decltype(Par)::value;
        dw_rows(y, Par);
        if (y + 3 <= y_end) gemm_row(xa, y + 3, (2 * PAR + 3) & 3);  // rows y-1, y of the ring are free now
        if (y + 4 <= y_end) gemm_row(xb, y + 4, (2 * PAR) & 3);
        xload(xa, y + 5); xload(xb, y + 6);                          // land during the next D
    };
    for (int y = y_begin; y < y_end; y += 4) {
        step(y, std::integral_constant<int, 0>{});
        if (y + 2 < y_end) step(y + 2, std::integral_constant<int, 1>{});
    }
}

bool pwdw_pk_supported(int Kb, int W, int H) { return gemm_pk_ks(Kb) > 0 && gemm_pk_ks(Kb) <= 3 && W >= 16 && H >= 6; }

int pwdw_pk_launch(PwDwPk g, hipStream_t stream) {
    const int KS = gemm_pk_ks(g.Kb);
    RST_REQUIRE(KS > 0 && KS <= 3, "bf16 fused 1x1 + depthwise: %d input channels not supported (at most 96)", g.Kb * 8);
    RST_REQUIRE(g.A && g.X && g.Y && g.ln_s && g.ln_c && g.dww, "bf16 fused 1x1 + depthwise: null operand");
    RST_REQUIRE(aligned16(g.A) && aligned16(g.X) && aligned16(g.Y) && aligned16(g.dww), "bf16 fused 1x1 + depthwise: operands must be 16-byte aligned");
    RST_REQUIRE(g.B <= 65535, "bf16 fused 1x1 + depthwise: batch %d exceeds 65535", g.B);
    {
        // the ring form.  RST_BF16_RING=0 keeps the register form below (tests, A/B runs)
        static const int ring_mode = getenv("RST_BF16_RING") ? atoi(getenv("RST_BF16_RING")) : 1;
        if (ring_mode && g.H >= 2 && (int64_t)g.H * g.W < (1ll << 27) && (int64_t)g.Kb * g.H * g.W * 16 < (1ll << 31)) {
            const int npairs = (int)cdiv(g.Mb, 4);
            // rows per item: long ranges recompute fewer halo rows; short ones give the chip enough waves
            static const int ring_rows = getenv("RST_BF16_RING_ROWS") ? atoi(getenv("RST_BF16_RING_ROWS")) : 32;
            g.rows = ring_rows >= 4 ? ring_rows / 4 * 4 : 32;
            while (g.rows > 8 && (int64_t)cdiv(g.W, 30) * cdiv(g.H, g.rows) * g.B * npairs < 256 * 8 * 2) g.rows >>= 1;
            if (g.rows > g.H) g.rows = (int)round_up(g.H, 4);
            const int64_t items = cdiv(g.W, 30) * cdiv(g.H, g.rows) * npairs;
            const dim3 grid((unsigned)round_up(cdiv(items, 4), 8), 1, (unsigned)g.B);
            const double px = (double)g.H * g.W * g.B;
            const double mout = g.gate ? g.Mb * 4 : g.Mb * 8;
            ProfScope prof(stream, g.gate ? "pk_pwdw_gate" : "pk_pwdw", px * g.Mb * 8 * (2.0 * g.Kb * 8 + 18.0), px * 2.0 * (g.Kb * 8 + mout));
#define RING_GO(ks)                                                                                    \
    do {                                                                                               \
        if (g.gate) hipLaunchKernelGGL((pwdw_ring_kernel<ks, true>), grid, dim3(256), 0, stream, g);    \
        else hipLaunchKernelGGL((pwdw_ring_kernel<ks, false>), grid, dim3(256), 0, stream, g);          \
    } while (0)
            if (KS == 1) RING_GO(1); else if (KS == 2) RING_GO(2); else RING_GO(3);
#undef RING_GO
            RST_LAUNCH_CHECK();
            return RST_OK;
        }
    }
    g.pb = gemm_pk_pairs(g.Kb, g.Mb);
    g.ngroups = (int)cdiv(cdiv(g.Mb, 4), g.pb);
    const int RO = KS <= 2 ? 6 : 4;
    const int npatch = (int)(cdiv(g.W, 14) * cdiv(g.H, RO));
    // about three blocks per CU over the chip, at least one patch per wave
    int ppb = (int)cdiv((int64_t)npatch * g.B * g.ngroups, 768);
    ppb = (int)round_up(ppb < 4 ? 4 : ppb, 4);
    g.patches_per_block = ppb;
    const dim3 grid((unsigned)cdiv(npatch, ppb), (unsigned)g.ngroups, (unsigned)g.B);
    const size_t lds = (size_t)g.pb * 2 * KS * 1024 + (size_t)g.pb * 32 * 2 * sizeof(float);
    RST_REQUIRE(lds <= 64 * 1024, "bf16 fused 1x1 + depthwise: LDS plan of %zu bytes", lds);
    const double px = (double)g.H * g.W * g.B;
    const double mout = g.gate ? g.Mb * 4 : g.Mb * 8;
    ProfScope prof(stream, g.gate ? "pk_pwdw_gate" : "pk_pwdw", px * g.Mb * 8 * (2.0 * g.Kb * 8 + 18.0), px * 2.0 * (g.Kb * 8 + mout));
#define PWDW_GO(ks)                                                                                              \
    do {                                                                                                         \
        if (g.gate) hipLaunchKernelGGL((pwdw_pk_kernel<ks, (ks <= 2 ? 8 : 6), true>), grid, dim3(256), lds, stream, g);   \
        else hipLaunchKernelGGL((pwdw_pk_kernel<ks, (ks <= 2 ? 8 : 6), false>), grid, dim3(256), lds, stream, g);        \
    } while (0)
    if (KS == 1) PWDW_GO(1); else if (KS == 2) PWDW_GO(2); else PWDW_GO(3);
#undef PWDW_GO
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// ---- Gram matrices q k^T and squared norms over the pixels --------------------------------------------
// A wave takes a T x T block of 16 x 16 tiles of one (sample, head) over one range of pixels, 32 at a time:
// the 16-byte chunks of q and k go to a wave-private LDS image as they are, and come back transposed
// (8 pixels of one channel per lane) through ds_read_b64_tr_b16.  The squared norms are the diagonals of
// q q^T and k k^T, taken by the matrix pipe as well.  Every wave writes its own slab.
template <int T>
__global__ void __launch_bounds__(256)
gram_pk_kernel(const GramPk g) {
    constexpr int CBS = 576;                           // bytes between channel blocks of the image (512 + 64: bank spread)
    constexpr int IMG = 2 * T * CBS;                   // one operand
    __shared__ __attribute__((aligned(16))) char lds[4][2 * IMG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t id = (int64_t)blockIdx.x * 4 + wave;
    const int nsub = g.nsub;
    const int sub = (int)(id % (nsub * nsub)); id /= nsub * nsub;
    const int split = (int)(id % g.splits);
    const int bh = (int)(id / g.splits);
    if (bh >= g.B * g.heads) return;                   // no barrier in this kernel
    const int sm = sub / nsub, sn = sub % nsub;
    const int b = bh / g.heads, h = bh % g.heads;
    const int64_t N = g.N;
    const u32x4* Q = reinterpret_cast<const u32x4*>(g.Q + (int64_t)b * g.bs) + (int64_t)(h * g.cb + sm * 2 * T) * N;
    const u32x4* Kp = reinterpret_cast<const u32x4*>(g.Kx + (int64_t)b * g.bs) + (int64_t)(h * g.cb + sn * 2 * T) * N;
    const int64_t p0 = (int64_t)split * g.chunk_px;
    const int64_t p1 = p0 + g.chunk_px < N ? p0 + g.chunk_px : N;
    char* img = lds[wave];
    const uint32_t img_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)img;   // LDS byte address

    f32x4 acc[T][T], accq[T], acck[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
        accq[i] = acck[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < T; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // transposed-read address of this lane inside one operand image, for tile 0, first block of 4 pixels
    const int l16 = lane & 15, gq = lane >> 4;
    const int tq = l16 >> 2, tp = l16 & 3;
    const uint32_t tr_off = (uint32_t)((tp >> 1) * CBS + (8 * gq + tq) * 16 + (tp & 1) * 8);

    u32x4 rq[T], rk[T];
    auto gload = [&](int64_t n0) {
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int idx = i * 64 + lane, cbl = idx >> 5;
            const int64_t px = n0 + (idx & 31);
            const u32x4 z = {0u, 0u, 0u, 0u};
            rq[i] = px < p1 ? Q[(int64_t)cbl * N + px] : z;
            rk[i] = px < p1 ? Kp[(int64_t)cbl * N + px] : z;
        }
    };
    gload(p0);
    for (int64_t n0 = p0; n0 < p1; n0 += 32) {
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int idx = i * 64 + lane;
            const int off = (idx >> 5) * CBS + (idx & 31) * 16;
            *reinterpret_cast<u32x4*>(img + off) = rq[i];
            *reinterpret_cast<u32x4*>(img + IMG + off) = rk[i];
        }
        gload(n0 + 32);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bf16x8 fq[T], fk[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            s16x4 q0, q1, k0, k1;
            const uint32_t aq = img_addr + tr_off + t * 2 * CBS, ak = aq + IMG;
            asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:64\n\t"
                         "ds_read_b64_tr_b16 %2, %5\n\tds_read_b64_tr_b16 %3, %5 offset:64\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1), "=&v"(k0), "=&v"(k1) : "v"(aq), "v"(ak) : "memory");
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 qq = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
            const s16x8 kk = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
            fq[t] = __builtin_bit_cast(bf16x8, qq);
            fk[t] = __builtin_bit_cast(bf16x8, kk);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < T; ++i) {
#pragma unroll
            for (int j = 0; j < T; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[i], fk[j], acc[i][j], 0, 0, 0);
            if (sn == 0) accq[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[i], fq[i], accq[i], 0, 0, 0);
            if (sm == 0) acck[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[i], fk[i], acck[i], 0, 0, 0);
        }
    }
    const int c = g.cb * 8;
    float* out = g.partial + ((int64_t)bh * g.splits + split) * c * c;
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(int64_t)((sm * T + i) * 16 + 4 * gq + r) * c + (sn * T + j) * 16 + l16] = acc[i][j][r];
    float* oq = g.sqq + ((int64_t)bh * g.splits + split) * c;
    float* ok = g.sqk + ((int64_t)bh * g.splits + split) * c;
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (l16 == 4 * gq + r) {
                if (sn == 0) oq[(sm * T + i) * 16 + l16] = accq[i][r];
                if (sm == 0) ok[(sn * T + i) * 16 + l16] = acck[i][r];
            }
}

static int gram_pk_tiles(int c) {
    const int t = c / 16;
    if (c % 16 != 0 || t < 1) return 0;
    for (int T = 3; T >= 1; --T) if (t % T == 0) return T;
    return 0;
}
bool gram_pk_supported(int c) { return gram_pk_tiles(c) > 0 && c <= 192; }

void gram_pk_plan(int c, int64_t N, int BH, int* splits, int64_t* chunk) {
    const int T = gram_pk_tiles(c), nsub = T > 0 ? c / 16 / T : 1;
    // about two waves per SIMD over the chip (measured against one and four: 16 x 96 x 256^2 takes 119 us, 134 us with
    // half as many, 153 us with twice as many and their slabs), at least 8 tiles of 32 pixels each
    int64_t want = cdiv(2048, (int64_t)BH * nsub * nsub);
    if (want < 1) want = 1;
    int64_t ch = round_up(cdiv(N, want), 32);
    if (ch < 256) ch = 256;
    *chunk = ch;
    *splits = (int)cdiv(N, ch);
}

int gram_pk_launch(GramPk g, hipStream_t stream) {
    const int c = g.cb * 8;
    const int T = gram_pk_tiles(c);
    RST_REQUIRE(T > 0 && c <= 192, "bf16 attention: %d channels per head not supported (multiples of 16 up to 192)", c);
    RST_REQUIRE(aligned16(g.Q) && aligned16(g.Kx), "bf16 attention: operands must be 16-byte aligned");
    g.nsub = c / 16 / T;
    const int64_t items = (int64_t)g.nsub * g.nsub * g.splits * g.B * g.heads;
    const double px = (double)g.N * g.B * g.heads;
    ProfScope prof(stream, "pk_gram", 2.0 * px * c * (c + 2), px * 2.0 * 2.0 * c);
    const dim3 grid((unsigned)cdiv(items, 4));
    if (T == 3) hipLaunchKernelGGL(gram_pk_kernel<3>, grid, dim3(256), 0, stream, g);
    else if (T == 2) hipLaunchKernelGGL(gram_pk_kernel<2>, grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL(gram_pk_kernel<1>, grid, dim3(256), 0, stream, g);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// ---- layout conversion: planar NCHW (fp32 or bf16) <-> channel-packed bf16 ----------------------------
// one thread = 2 neighbouring pixels x one channel block.  Channels beyond C read as zero / are not written.
template <typename SRC>
__global__ void __launch_bounds__(256)
pk_pack_kernel(const SRC* __restrict__ src, bf16_t* __restrict__ dst, int C, int Cb, int64_t N) {
    const int64_t pp = (int64_t)blockIdx.x * 256 + threadIdx.x;        // pixel pair
    const int cb = blockIdx.y, b = blockIdx.z;
    if (pp * 2 >= N) return;
    float v0[8], v1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = cb * 8 + j;
        v0[j] = v1[j] = 0.f;
        if (ch < C) {
            const SRC* p = src + ((int64_t)b * C + ch) * N + pp * 2;
            if constexpr (sizeof(SRC) == 4) { v0[j] = p[0]; v1[j] = p[1]; }
            else { const uint32_t w = *reinterpret_cast<const uint32_t*>(p); v0[j] = bf_lo(w); v1[j] = bf_hi(w); }
        }
    }
    u32x4* d = reinterpret_cast<u32x4*>(dst) + ((int64_t)b * Cb + cb) * N + pp * 2;
    d[0] = pack8(v0);
    d[1] = pack8(v1);
}
template <typename DST>
__global__ void __launch_bounds__(256)
pk_unpack_kernel(const bf16_t* __restrict__ src, DST* __restrict__ dst, int C, int Cb, int64_t N) {
    const int64_t pp = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cb = blockIdx.y, b = blockIdx.z;
    if (pp * 2 >= N) return;
    const u32x4* s = reinterpret_cast<const u32x4*>(src) + ((int64_t)b * Cb + cb) * N + pp * 2;
    const u32x4 w0 = s[0], w1 = s[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = cb * 8 + j;
        if (ch >= C) continue;
        const uint32_t a = w0[j >> 1], c = w1[j >> 1];
        DST* p = dst + ((int64_t)b * C + ch) * N + pp * 2;
        if constexpr (sizeof(DST) == 4) {
            p[0] = (j & 1) ? bf_hi(a) : bf_lo(a);
            p[1] = (j & 1) ? bf_hi(c) : bf_lo(c);
        } else {
            const uint32_t lo = (j & 1) ? (a >> 16) : (a & 0xffffu), hi = (j & 1) ? (c & 0xffff0000u) : (c << 16);
            *reinterpret_cast<uint32_t*>(p) = lo | hi;
        }
    }
}

// ---- channels-last neighbours: the dense 3x3 convolutions run in a library that prefers [B][H][W][C] --------------
// packed -> channels-last bf16: one thread = one 16-byte chunk (C a multiple of 8)
__global__ void __launch_bounds__(256)
pk_to_nhwc_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int Cb, int64_t N) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (pixel, channel block), channel block fastest
    const int b = blockIdx.y;
    if (e >= N * Cb) return;
    const int cb = (int)(e % Cb);
    const int64_t n = e / Cb;
    reinterpret_cast<u32x4*>(dst)[(int64_t)b * N * Cb + e] = reinterpret_cast<const u32x4*>(src)[((int64_t)b * Cb + cb) * N + n];
}
// channels-last bf16 [B][H][W][C] -> packed, optionally through PixelUnshuffle(2) / PixelShuffle(2)
// (model/Restromer.py:196,211):
//   mode 0: y[c] = x[c]
//   mode 1: y[4c + 2i + j][yo][xo] = x[c][2yo + i][2xo + j]          (C -> 4C, H/2 x W/2)
//   mode 2: y[c][2y + i][2x + j]   = x[4c + 2i + j][y][x]            (C -> C/4, 2H x 2W)
// one thread = one output chunk (modes 0, 1) or the four output chunks of one input pixel (mode 2)
template <int MODE>
__global__ void __launch_bounds__(256)
pk_from_nhwc_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int C, int H, int W) {
    const int b = blockIdx.z, ob = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bf16_t* S = src + (int64_t)b * H * W * C;
    if (MODE == 0) {
        const int64_t N = (int64_t)H * W;
        if (t >= N) return;
        reinterpret_cast<u32x4*>(dst)[((int64_t)b * (C / 8) + ob) * N + t] = *reinterpret_cast<const u32x4*>(S + t * C + ob * 8);
    } else if (MODE == 1) {
        const int Ho = H / 2, Wo = W / 2;
        const int64_t No = (int64_t)Ho * Wo;
        if (t >= No) return;
        const int yo = (int)(t / Wo), xo = (int)(t % Wo);
        uint32_t w[4];                                     // (c0, c1) at (i, j) = (0,0) (0,1) (1,0) (1,1)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            w[q] = *reinterpret_cast<const uint32_t*>(S + ((int64_t)(2 * yo + (q >> 1)) * W + 2 * xo + (q & 1)) * C + ob * 2);
        u32x4 o;
        o[0] = (w[0] & 0xffffu) | (w[1] << 16);
        o[1] = (w[2] & 0xffffu) | (w[3] << 16);
        o[2] = (w[0] >> 16) | (w[1] & 0xffff0000u);
        o[3] = (w[2] >> 16) | (w[3] & 0xffff0000u);
        reinterpret_cast<u32x4*>(dst)[((int64_t)b * (C / 2) + ob) * No + t] = o;
    } else {
        const int64_t N = (int64_t)H * W;
        if (t >= N) return;
        const int y = (int)(t / W), x = (int)(t % W);
        const u32x4* in = reinterpret_cast<const u32x4*>(S + t * C + ob * 32);
        uint32_t e[16];                                    // 32 channels: element 4q + 2i + j, two per word
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4 v = in[k];
#pragma unroll
            for (int m = 0; m < 4; ++m) e[4 * k + m] = v[m];
        }
        const int64_t No = 4 * N;
        u32x4* out = reinterpret_cast<u32x4*>(dst) + ((int64_t)b * (C / 32) + ob) * No;
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) {
            // channel q of the chunk is element 4q + ij: word 2q + (ij >> 1), half ij & 1
            u32x4 o;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint32_t a = e[2 * (2 * m) + (ij >> 1)], c = e[2 * (2 * m + 1) + (ij >> 1)];
                o[m] = (ij & 1) ? ((a >> 16) | (c & 0xffff0000u)) : ((a & 0xffffu) | (c << 16));
            }
            out[(int64_t)(2 * y + (ij >> 1)) * (2 * W) + 2 * x + (ij & 1)] = o;
        }
    }
}

int pk_to_nhwc_launch(const bf16_t* src, bf16_t* dst, int B, int C, int64_t N, hipStream_t stream) {
    RST_REQUIRE(src && dst && B > 0 && C > 0 && N > 0, "bf16 channels-last: empty shape");
    RST_REQUIRE(C % 8 == 0, "bf16 channels-last: %d channels (a multiple of 8 is needed)", C);
    RST_REQUIRE(aligned16(src) && aligned16(dst), "bf16 channels-last: operands must be 16-byte aligned");
    RST_REQUIRE(B <= 65535, "bf16 channels-last: batch %d exceeds 65535", B);
    ProfScope prof(stream, "pk_to_nhwc", 0.0, (double)B * C * N * 4.0);
    hipLaunchKernelGGL(pk_to_nhwc_kernel, dim3((unsigned)cdiv(N * (C / 8), 256), B), dim3(256), 0, stream, src, dst, C / 8, N);
    RST_LAUNCH_CHECK();
    return RST_OK;
}
int pk_from_nhwc_launch(const bf16_t* src, bf16_t* dst, int mode, int B, int C, int H, int W, hipStream_t stream) {
    RST_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "bf16 channels-last: empty shape");
    RST_REQUIRE(mode >= 0 && mode <= 2, "bf16 channels-last: mode %d", mode);
    RST_REQUIRE(aligned16(src) && aligned16(dst), "bf16 channels-last: operands must be 16-byte aligned");
    RST_REQUIRE(B <= 65535, "bf16 channels-last: batch %d exceeds 65535", B);
    ProfScope prof(stream, mode == 0 ? "pk_from_nhwc" : (mode == 1 ? "pk_unshuffle" : "pk_shuffle"), 0.0, (double)B * C * H * W * 4.0);
    if (mode == 0) {
        RST_REQUIRE(C % 8 == 0, "bf16 channels-last: %d channels (a multiple of 8 is needed)", C);
        hipLaunchKernelGGL(pk_from_nhwc_kernel<0>, dim3((unsigned)cdiv((int64_t)H * W, 256), C / 8, B), dim3(256), 0, stream, src, dst, C, H, W);
    } else if (mode == 1) {
        RST_REQUIRE(C % 2 == 0 && H % 2 == 0 && W % 2 == 0, "bf16 pixel-unshuffle: C=%d H=%d W=%d must be even", C, H, W);
        hipLaunchKernelGGL(pk_from_nhwc_kernel<1>, dim3((unsigned)cdiv((int64_t)(H / 2) * (W / 2), 256), C / 2, B), dim3(256), 0, stream, src, dst, C, H, W);
    } else {
        RST_REQUIRE(C % 32 == 0, "bf16 pixel-shuffle: %d channels (a multiple of 32 is needed)", C);
        hipLaunchKernelGGL(pk_from_nhwc_kernel<2>, dim3((unsigned)cdiv((int64_t)H * W, 256), C / 32, B), dim3(256), 0, stream, src, dst, C, H, W);
    }
    RST_LAUNCH_CHECK();
    return RST_OK;
}

int pk_pack_launch(const void* src, int src_is_fp32, bf16_t* dst, int B, int C, int64_t N, hipStream_t stream) {
    RST_REQUIRE(src && dst && B > 0 && C > 0 && N > 0 && N % 2 == 0, "bf16 pack: bad shape B=%d C=%d N=%lld (N must be even)", B, C, (long long)N);
    RST_REQUIRE(aligned16(dst) && (reinterpret_cast<uintptr_t>(src) & 3u) == 0, "bf16 pack: misaligned operand");
    const int Cb = (int)cdiv(C, 8);
    const dim3 grid((unsigned)cdiv(N / 2, 256), Cb, B);
    ProfScope prof(stream, "pk_pack", 0.0, (double)B * C * N * (src_is_fp32 ? 6.0 : 4.0));
    if (src_is_fp32) hipLaunchKernelGGL(pk_pack_kernel<float>, grid, dim3(256), 0, stream, static_cast<const float*>(src), dst, C, Cb, N);
    else hipLaunchKernelGGL(pk_pack_kernel<bf16_t>, grid, dim3(256), 0, stream, static_cast<const bf16_t*>(src), dst, C, Cb, N);
    RST_LAUNCH_CHECK();
    return RST_OK;
}
int pk_unpack_launch(const bf16_t* src, void* dst, int dst_is_fp32, int B, int C, int64_t N, hipStream_t stream) {
    RST_REQUIRE(src && dst && B > 0 && C > 0 && N > 0 && N % 2 == 0, "bf16 unpack: bad shape B=%d C=%d N=%lld (N must be even)", B, C, (long long)N);
    RST_REQUIRE(aligned16(src) && (reinterpret_cast<uintptr_t>(dst) & 3u) == 0, "bf16 unpack: misaligned operand");
    const int Cb = (int)cdiv(C, 8);
    const dim3 grid((unsigned)cdiv(N / 2, 256), Cb, B);
    ProfScope prof(stream, "pk_unpack", 0.0, (double)B * C * N * (dst_is_fp32 ? 6.0 : 4.0));
    if (dst_is_fp32) hipLaunchKernelGGL(pk_unpack_kernel<float>, grid, dim3(256), 0, stream, src, static_cast<float*>(dst), C, Cb, N);
    else hipLaunchKernelGGL(pk_unpack_kernel<bf16_t>, grid, dim3(256), 0, stream, src, static_cast<bf16_t*>(dst), C, Cb, N);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst
