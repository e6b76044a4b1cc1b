// The c x c side of MDTA (model/Restromer.py:137-149): everything that happens
// between the two sweeps over the image works on (sample, head) matrices of at
// most a few hundred rows.
//
// forward:
//   G = sum of Gram slabs, nq/nk = clamped row norms  (F.normalize folds into a
//   rank-1 rescale of the raw Gram matrix: <q_i/|q_i|, k_j/|k_j|> = G_ij/(|q_i||k_j|))
//   Ghat = G / (nq nk^T);  A = softmax_j(Ghat * temperature)
//   Mfold[b] = W_o * blockdiag_h(A[b,h])   (project_out folded into attn.v)
// backward (see attention_bwd in api.hip for the derivation):
//   dA = (W_o^T P)_hh, dS = A o (dA - rowsum(dA o A)), dtemp, dGhat, the
//   matrices that turn q,k into dq,dk, and dW_o.
#include "kernels.hpp"
#include "bf16_common.hpp"

namespace rst {

constexpr float NORM_EPS = 1e-12f;

// Sum of one element over its slabs, slab 0 first (the order of a plain loop, so the result is the same bits), with
// the loads of eight slabs in flight at a time: a loop of dependent load + add pays the memory latency once per slab.
__device__ __forceinline__ float slab_sum(const float* __restrict__ p, int64_t stride, int splits) {
    float s = 0.f;
    for (int sp0 = 0; sp0 < splits; sp0 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = sp0 + k < splits ? p[(int64_t)(sp0 + k) * stride] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    return s;
}

// One wave per row i of one (b,h).  grid = (ceil(c/4), B*heads), block = 256.
__global__ void __launch_bounds__(256)
attn_softmax_kernel(const float* __restrict__ gram_partial,   // [BH][splits][c][c]
                    const float* __restrict__ sqq_partial,    // [BH][splits][c]
                    const float* __restrict__ sqk_partial,    // [BH][splits][c]
                    const float* __restrict__ temperature,    // [heads]
                    float* __restrict__ attn,                 // [BH][c][c]
                    float* __restrict__ ghat,                 // [BH][c][c] or null
                    float* __restrict__ nq_out,               // [BH][c] or null (clamped norms)
                    float* __restrict__ nk_out,               // [BH][c] or null
                    int heads, int c, int splits) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // nk[c] + 4 rows x c
    float* nk = smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int bh = blockIdx.y;
    const int h = bh % heads;
    for (int j = tid; j < c; j += 256) {
        const float s = slab_sum(sqk_partial + (int64_t)bh * splits * c + j, c, splits);
        const float n = fmaxf(sqrtf(s), NORM_EPS);
        nk[j] = n;
        if (nk_out != nullptr && blockIdx.x == 0) nk_out[(int64_t)bh * c + j] = n;
    }
    __syncthreads();
    const int i = blockIdx.x * 4 + wave;
    if (i >= c) return;
    const float sq = slab_sum(sqq_partial + (int64_t)bh * splits * c + i, c, splits);
    const float nq = fmaxf(sqrtf(sq), NORM_EPS);
    if (nq_out != nullptr && lane == 0) nq_out[(int64_t)bh * c + i] = nq;
    const float temp = temperature[h];
    float* row = smem + c + wave * c;
    float mx = -INFINITY;
    for (int j = lane; j < c; j += 64) {
        const float g = slab_sum(gram_partial + ((int64_t)bh * splits * c + i) * c + j, (int64_t)c * c, splits);
        const float gh = g / (nq * nk[j]);
        if (ghat != nullptr) ghat[((int64_t)bh * c + i) * c + j] = gh;
        const float l = gh * temp;
        row[j] = l;
        mx = fmaxf(mx, l);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.f;
    for (int j = lane; j < c; j += 64) {
        const float e = expf(row[j] - mx);
        row[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < c; j += 64) attn[((int64_t)bh * c + i) * c + j] = row[j] * inv;
}

// Element index of (row position m, contraction position k) in the bf16 fragment buffer pk_prep_a_kernel writes:
// frag[mt][ks][lane][j] with m = 32 (mt>>1) + 8 (rr>>2) + 4 (mt&1) + (rr&3), rr = lane & 15, k = 32 ks + 8 (lane>>4) + j.
// Rows m, m+1, m+2, m+3 of an aligned group of four are 8 elements apart.
__device__ __forceinline__ int64_t fold_frag_index(int m, int k, int KS) {
    const int mt = 2 * (m >> 5) + ((m >> 2) & 1), rr = 4 * ((m >> 3) & 3) + (m & 3);
    const int ks = k >> 5, lane = 16 * ((k >> 3) & 3) + rr;
    return (((int64_t)mt * KS + ks) * 64 + lane) * 8 + (k & 7);
}

// Mfold[b][m][h*c + j] = sum_i Wo[m][h*c + i] * A[b,h][i][j]
// grid = (ceil(rows/16), heads, B), block = 256: 16 rows m x c columns j (rows = C, or the padded row count of the
// fragment buffer when fragments are written).  A thread owns four rows of one column: per i one
// read of A (consecutive lanes, consecutive columns) and one 16-byte broadcast read of the transposed weight tile serve
// four multiply-adds; the sum runs over i in order.
__global__ void __launch_bounds__(256)
attn_fold_kernel(const float* __restrict__ wo,      // [C][C]
                 const float* __restrict__ attn,    // [B*heads][c][c]
                 float* __restrict__ mfold,         // [B][C][C] or null
                 bf16_t* __restrict__ frag,         // or null: Mfold[b] as bf16 operand fragments (layout of pk_prep_a_kernel)
                 int64_t frag_bs, int KS,
                 int C, int heads, int c,
                 _Float16* __restrict__ frag16,     // or null: Mfold[b] as split-f16 operand fragments (layout of gemm_f16s_prep_kernel)
                 int64_t frag16_bs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // W^T: c x 16, then A: c*c
    float* Wt = smem;
    float* As = smem + 16 * c;
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    const float* Abh = attn + ((int64_t)b * heads + h) * c * c;
    if ((c & 3) == 0 && (reinterpret_cast<uintptr_t>(Abh) & 15) == 0) {
        for (int e = tid; e < c * c / 4; e += 256)
            reinterpret_cast<f32x4*>(As)[e] = reinterpret_cast<const f32x4*>(Abh)[e];
    } else {
        for (int e = tid; e < c * c; e += 256) As[e] = Abh[e];
    }
    for (int r = tid >> 4; r < 16; r += 16)
        for (int i = tid & 15; i < c; i += 16)
            Wt[i * 16 + r] = (m0 + r < C) ? wo[(int64_t)(m0 + r) * C + h * c + i] : 0.f;
    __syncthreads();
    for (int t = tid; t < 4 * c; t += 256) {
        const int rq = t / c, j = t - rq * c;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* ap = As + j;
        const f32x4* wp = reinterpret_cast<const f32x4*>(Wt) + rq;
#pragma unroll 4
        for (int i = 0; i < c; ++i) {
            const float a = ap[i * c];
            const f32x4 w = wp[i * 4];
            acc.x = fmaf(w.x, a, acc.x); acc.y = fmaf(w.y, a, acc.y);
            acc.z = fmaf(w.z, a, acc.z); acc.w = fmaf(w.w, a, acc.w);
        }
        if (mfold != nullptr) {
            float* out = mfold + ((int64_t)b * C + m0 + 4 * rq) * C + h * c + j;
            if (m0 + 4 * rq + 0 < C) out[0] = acc.x;
            if (m0 + 4 * rq + 1 < C) out[C] = acc.y;
            if (m0 + 4 * rq + 2 < C) out[2 * (int64_t)C] = acc.z;
            if (m0 + 4 * rq + 3 < C) out[3 * (int64_t)C] = acc.w;
        }
        if (frag != nullptr) {
            // rows past C are padding of the fragment buffer: their weights were staged as zeros, so acc is zero
            uint16_t* fo = reinterpret_cast<uint16_t*>(frag) + (int64_t)b * frag_bs + fold_frag_index(m0 + 4 * rq, h * c + j, KS);
            fo[0] = to_bf16_bits(acc.x); fo[8] = to_bf16_bits(acc.y); fo[16] = to_bf16_bits(acc.z); fo[24] = to_bf16_bits(acc.w);
        }
        if (frag16 != nullptr) {
            // [16-row tile][32-column chunk][hi, lo][lane = row + 16 g][8]: lane group g holds columns 2 g + 8 i + e of the chunk
            const int k = h * c + j, kc = k >> 5, kin = k & 31;
            const int64_t at = ((((int64_t)blockIdx.x * KS + kc) * 2) * 64 + 4 * rq + 16 * ((kin & 7) >> 1)) * 8 + 2 * (kin >> 3) + (kin & 1);
            _Float16* fo = frag16 + (int64_t)b * frag16_bs + at;
            const float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 hi = (_Float16)v[e];
                fo[8 * e] = hi;
                fo[8 * e + 64 * 8] = (_Float16)((v[e] - (float)hi) * 2048.f);
            }
        }
    }
    if (frag16 != nullptr && h == heads - 1) {
        const int kpad = 32 * KS - C;
        for (int t = tid; t < 16 * kpad; t += 256) {
            const int r = t / kpad, k = C + t - r * kpad, kc = k >> 5, kin = k & 31;
            const int64_t at = ((((int64_t)blockIdx.x * KS + kc) * 2) * 64 + r + 16 * ((kin & 7) >> 1)) * 8 + 2 * (kin >> 3) + (kin & 1);
            frag16[(int64_t)b * frag16_bs + at] = (_Float16)0.f;
            frag16[(int64_t)b * frag16_bs + at + 64 * 8] = (_Float16)0.f;
        }
    }
    // contraction positions past C (up to a multiple of 32) are padding as well: the last head's blocks clear them
    if (frag != nullptr && h == heads - 1) {
        const int kpad = 32 * KS - C;
        for (int t = tid; t < 16 * kpad; t += 256) {
            const int r = t / kpad, k = C + t - r * kpad;
            reinterpret_cast<uint16_t*>(frag)[(int64_t)b * frag_bs + fold_frag_index(m0 + r, k, KS)] = 0;
        }
    }
}


// ---- backward ---------------------------------------------------------------------
// One wave per row i of one (b,h); grid = (ceil(c/4), B*heads), block 256.
//   dA_ij   = sum_m Wo[m][hc+i] * P[b][m][hc+j]   -- a (c x C)(C x c) product per (b,h), formed by
//             gemm_nn beforehand and handed over in the dGG buffer (overwritten in place here)
//   dS_ij   = A_ij * (dA_ij - sum_j dA_ij A_ij)
//   dtemp  += sum_ij dS_ij * Ghat_ij ;  dGhat = dS * temperature
//   Eq_ij   = dGhat_ij / (nq_i nk_j)              (dq_i = sum_j Eq_ij k_j + Dq_i q_i)
//   Dq_i    = -(sum_j dGhat_ij Ghat_ij) / nq_i^2  (0 where the norm was clamped)
//   dGG_ij  = dGhat_ij * Ghat_ij                  (column sums give Dk)
__global__ void __launch_bounds__(256)
attn_bwd_rows_kernel(const float* __restrict__ attn,
                     const float* __restrict__ ghat, const float* __restrict__ nq, const float* __restrict__ nk,
                     const float* __restrict__ temperature,
                     float* __restrict__ Eq, float* __restrict__ Dq, float* dGG, float* __restrict__ dtemp_row,
                     int heads, int c) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // 4 rows x c
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
    const int i = blockIdx.x * 4 + wave;
    if (i >= c) return;
    float* row = smem + wave * c;
    const int64_t rbase = ((int64_t)bh * c + i) * c;
    float rs = 0.f;
    for (int j = lane; j < c; j += 64) {
        const float s = dGG[rbase + j];
        row[j] = s;
        rs = fmaf(s, attn[rbase + j], rs);
    }
    rs = wave_sum(rs);
    const float temp = temperature[h];
    const float nqi = nq[(int64_t)bh * c + i];
    float dt = 0.f;
    for (int j = lane; j < c; j += 64) {
        const float a = attn[rbase + j], gh = ghat[rbase + j];
        const float dS = a * (row[j] - rs);
        dt = fmaf(dS, gh, dt);
        const float dG = dS * temp;
        Eq[rbase + j] = dG / (nqi * nk[(int64_t)bh * c + j]);
        dGG[rbase + j] = dG * gh;
    }
    dt = wave_sum(dt);
    if (lane == 0) {
        dtemp_row[(int64_t)bh * c + i] = dt;
        Dq[(int64_t)bh * c + i] = (nqi > NORM_EPS) ? -(dt * temp) / (nqi * nqi) : 0.f;
    }
}

// grid = B*heads, block 256: Dk_j = -(sum_i dGG_ij)/nk_j^2, dtemp_b[bh] = sum_i dtemp_row[i]
__global__ void __launch_bounds__(256)
attn_bwd_cols_kernel(const float* __restrict__ dGG, const float* __restrict__ nk, const float* __restrict__ dtemp_row,
                     float* __restrict__ Dk, float* __restrict__ dtemp_b, int c) {
    __shared__ float red[4];
    __shared__ float part[4][256];
    const int bh = blockIdx.x, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // a wave takes every fourth row, a lane its columns: c/4 independent loads per lane instead of a chain of c
    for (int j0 = 0; j0 < c; j0 += 256) {
        for (int jj = lane; jj < 256 && j0 + jj < c; jj += 64) {
            float s = 0.f;
            for (int i = wave; i < c; i += 4) s += dGG[((int64_t)bh * c + i) * c + j0 + jj];
            part[wave][jj] = s;
        }
        __syncthreads();
        if (j0 + tid < c) {
            const int j = j0 + tid;
            const float s = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
            const float n = nk[(int64_t)bh * c + j];
            Dk[(int64_t)bh * c + j] = (n > NORM_EPS) ? -s / (n * n) : 0.f;
        }
        __syncthreads();
    }
    float t = 0.f;
    for (int i = tid; i < c; i += 256) t += dtemp_row[(int64_t)bh * c + i];
    t = wave_sum(t);
    if ((tid & 63) == 0) red[tid >> 6] = t;
    __syncthreads();
    if (tid == 0) dtemp_b[bh] = red[0] + red[1] + red[2] + red[3];
}

// dWo_b[b][m][hc+i] = sum_j P[b][m][hc+j] * A[b,h][i][j];  grid = (ceil(C/16), heads, B)
__global__ void __launch_bounds__(256)
attn_bwd_wo_kernel(const float* __restrict__ P, const float* __restrict__ attn, float* __restrict__ dwo_b,
                   int C, int heads, int c) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // A: c*(c+1), P slab: 16*c
    float* As = smem;
    float* Ps = smem + c * (c + 1);
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    const float* Abh = attn + ((int64_t)b * heads + h) * c * c;
    for (int e = tid; e < c * c; e += 256) As[(e / c) * (c + 1) + (e % c)] = Abh[e];
    for (int e = tid; e < 16 * c; e += 256) {
        const int r = e / c, j = e % c;
        Ps[e] = (m0 + r < C) ? P[((int64_t)b * C + m0 + r) * C + h * c + j] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < 16 * c; e += 256) {
        const int r = e / c, i = e % c;
        if (m0 + r >= C) continue;
        float s = 0.f;
        for (int j = 0; j < c; ++j) s = fmaf(Ps[r * c + j], As[i * (c + 1) + j], s);
        dwo_b[((int64_t)b * C + m0 + r) * C + h * c + i] = s;
    }
}

int attn_bwd_small_launch(const float* P, const float* wo, const float* attn, const float* ghat, const float* nq,
                          const float* nk, const float* temperature, float* Eq, float* Dq, float* Dk, float* dGG,
                          float* dtemp_row, float* dtemp_b, int B, int C, int heads, int c,
                          hipStream_t stream) {
    {
        // dA[b,h] = Wo[:, h-slice]^T P[b][:, h-slice] into dGG
        GemmNN g{};
        g.A = wo; g.a_s1 = 0; g.a_s2 = c; g.lda = C; g.transA = 1;
        g.X = P; g.x_s1 = (int64_t)C * C; g.x_s2 = c; g.ldx = C;
        g.Y = dGG; g.y_s1 = (int64_t)heads * c * c; g.y_s2 = (int64_t)c * c; g.ldy = c;
        g.M = c; g.K = C; g.N = c; g.nb1 = B; g.nb2 = heads;
        RST_TRY(gemm_nn_launch(g, stream));
    }
    ProfScope prof(stream, "attn_bwd_small", 2.0 * B * C * C * c, 4.0 * B * (3.0 * C * C + 6.0 * heads * c * c));
    {
        const size_t lds = (size_t)4 * c * sizeof(float);
        RST_REQUIRE(lds <= 64 * 1024, "attention: channels per head %d too large", c);
        dim3 grid((unsigned)cdiv(c, 4), (unsigned)(B * heads));
        hipLaunchKernelGGL(attn_bwd_rows_kernel, grid, dim3(256), lds, stream, attn, ghat, nq, nk, temperature,
                           Eq, Dq, dGG, dtemp_row, heads, c);
        RST_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(attn_bwd_cols_kernel, dim3((unsigned)(B * heads)), dim3(256), 0, stream, dGG, nk, dtemp_row, Dk,
                       dtemp_b, c);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// dWo_b: reads P and the softmax only, nothing of the chain above -- callers may queue it on another stream
int attn_bwd_wo_launch(const float* P, const float* attn, float* dwo_b, int B, int C, int heads, int c, hipStream_t stream) {
    ProfScope prof(stream, "attn_bwd_wo", 2.0 * B * C * C * c, 4.0 * B * (2.0 * C * C + (double)heads * c * c));
    {
        const size_t lds = (size_t)(c * (c + 1) + 16 * c) * sizeof(float);
        RST_REQUIRE(lds <= 160 * 1024, "attention: channels per head %d too large", c);
        if (lds > 64 * 1024)
            RST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_wo_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        dim3 grid((unsigned)cdiv(C, 16), (unsigned)heads, (unsigned)B);
        hipLaunchKernelGGL(attn_bwd_wo_kernel, grid, dim3(256), lds, stream, P, attn, dwo_b, C, heads, c);
        RST_LAUNCH_CHECK();
    }
    return RST_OK;
}

int attn_softmax_launch(const float* gram_partial, const float* sqq_partial, const float* sqk_partial,
                        const float* temperature, float* attn, float* ghat, float* nq, float* nk,
                        int B, int heads, int c, int splits, hipStream_t stream) {
    const size_t lds = (size_t)(c + 4 * c) * sizeof(float);
    RST_REQUIRE(lds <= 64 * 1024, "attention: channels per head %d too large", c);
    dim3 grid((unsigned)cdiv(c, 4), (unsigned)(B * heads));
    ProfScope prof(stream, "attn_softmax", (double)B * heads * c * c * (splits + 8.0), 4.0 * B * heads * c * c * (splits + 1.0));
    hipLaunchKernelGGL(attn_softmax_kernel, grid, dim3(256), lds, stream, gram_partial, sqq_partial, sqk_partial,
                       temperature, attn, ghat, nq, nk, heads, c, splits);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

int attn_fold_launch(const float* wo, const float* attn, float* mfold, int B, int C, int heads, int c, hipStream_t stream) {
    return attn_fold_frag_launch(wo, attn, mfold, nullptr, 0, 0, 0, B, C, heads, c, stream);
}

// As attn_fold_launch; with `frag` the folded matrices also (or only, mfold == null) leave as the bf16 operand fragments
// of gemm_pk (ntiles 16-row tiles, KS 32-column steps per sample, frag_bs elements apart), padding cleared.
int attn_fold_frag_launch(const float* wo, const float* attn, float* mfold, bf16_t* frag, int64_t frag_bs, int ntiles, int KS,
                          int B, int C, int heads, int c, hipStream_t stream) {
    const size_t lds = (size_t)(c * c + 16 * c) * sizeof(float);
    RST_REQUIRE(lds <= 160 * 1024, "attention: channels per head %d too large for the fold kernel", c);
    RST_REQUIRE(mfold != nullptr || frag != nullptr, "attention fold: no output");
    RST_REQUIRE(frag == nullptr || (c % 8 == 0 && ntiles % 2 == 0 && 16 * ntiles >= C && 32 * KS >= C && 32 * (KS - 1) < C),
                "attention fold: fragment output needs 8 | channels per head and a fragment buffer that covers C");
    if (lds > 64 * 1024)
        RST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fold_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)(frag != nullptr ? ntiles : cdiv(C, 16)), (unsigned)heads, (unsigned)B);
    ProfScope prof(stream, "attn_fold", 2.0 * B * C * C * c, 4.0 * B * (2.0 * C * C + (double)heads * c * c));
    hipLaunchKernelGGL(attn_fold_kernel, grid, dim3(256), lds, stream, wo, attn, mfold, frag, frag_bs, KS, C, heads, c,
                       static_cast<_Float16*>(nullptr), (int64_t)0);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// As attn_fold_launch; the folded matrices also leave as the split-f16 operand fragments of gemm_f16s (what
// gemm_f16s_prep_launch would make of mfold: rows padded to 16, columns to 32 with zeros), frag_bytes apart per sample.
int attn_fold_f16s_launch(const float* wo, const float* attn, float* mfold, void* frag, int64_t frag_bytes, int B, int C, int heads,
                          int c, hipStream_t stream) {
    const size_t lds = (size_t)(c * c + 16 * c) * sizeof(float);
    RST_REQUIRE(lds <= 160 * 1024, "attention: channels per head %d too large for the fold kernel", c);
    RST_REQUIRE(frag != nullptr && frag_bytes % 16 == 0 && (size_t)frag_bytes >= gemm_f16s_frag_bytes(C, C),
                "attention fold: fragment buffer too small");
    if (lds > 64 * 1024)
        RST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fold_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)cdiv(C, 16), (unsigned)heads, (unsigned)B);
    ProfScope prof(stream, "attn_fold", 2.0 * B * C * C * c, 4.0 * B * (2.0 * C * C + (double)heads * c * c));
    hipLaunchKernelGGL(attn_fold_kernel, grid, dim3(256), lds, stream, wo, attn, mfold, static_cast<bf16_t*>(nullptr), (int64_t)0,
                       (int)cdiv(C, 32), C, heads, c, static_cast<_Float16*>(frag), frag_bytes / 2);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst
