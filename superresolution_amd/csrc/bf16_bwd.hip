// Backward kernels of the bf16 storage path: gradients of one TransformerBlock (model/Restromer.py:163-167) on
// channel-packed bf16 tensors ([B][Cb][H*W][8], see bf16_pk.hip), fp32 sums, fp32 parameter gradients.
//
//   * wgrad_pk: the products over the pixels (weight gradients sum_px dY X^T, the per-sample dY V^T of the attention)
//     on the bf16 matrix instructions, operands transposed on the way out of LDS as in the Gram kernel; optionally
//     the row sums of the first operand (bias gradients) through a product with a matrix of ones.
//   * dw_wgrad_pk: the nine tap sums and the bias sum of a depthwise 3x3 per channel.
//   * gate_fwd_pk / gate_bwd_pk: gelu(x1) * x2 and its gradient on the interleaved layout of the gated feed-forward.
//   * ln_fwd_pk / ln_bwd_pk: channel LayerNorm (normalisation only) and its input gradient, one pixel per thread.
//   * small fp32 kernels that turn the sums into parameter gradients.
// The data gradients (W^T dY) are 1x1 convolutions with transposed weights and run on gemm_pk (bf16_pk.hip).
#include "kernels.hpp"
#include "bf16_common.hpp"

namespace rst {

// ---- weight preparation with a transposed and / or gathered source ----------------------------------------
// As pk_prep_a (bf16_pk.hip), for A[pos][kpos] = W^(T)[channel(pos)][channel(kpos0 + kpos)]: both the rows and the
// contraction index may run over the positions of a segmented packed buffer (src_channel), `transA` reads
// w[k * ldw + row].  Krange = positions of the contraction this fragment set covers (from kpos0).
__global__ void __launch_bounds__(256)
pk_prep_at_kernel(const float* __restrict__ w, int64_t w_bs, int ldw, int transA, int M, int mseg, int mseg_pad,
                  int Ksrc, int kseg, int kseg_pad, int kpos0, int Krange, int ntiles, int KS,
                  bf16_t* __restrict__ frag, int64_t frag_bs) {
    const int b = blockIdx.y;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (mt, ks, lane)
    if (e >= (int64_t)ntiles * KS * 64) return;
    const int lane = (int)(e & 63), ks = (int)((e >> 6) % KS), mt = (int)((e >> 6) / KS);
    const int rr = lane & 15;
    const int pos = 32 * (mt >> 1) + 8 * (rr >> 2) + 4 * (mt & 1) + (rr & 3);
    const int ch = src_channel(pos, mseg, mseg_pad, M);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kp = 32 * ks + 8 * (lane >> 4) + j;
        const int kc = kp < Krange ? src_channel(kpos0 + kp, kseg, kseg_pad, Ksrc) : -1;
        float x = 0.f;
        if (ch >= 0 && kc >= 0) x = transA ? w[(int64_t)b * w_bs + (int64_t)kc * ldw + ch] : w[(int64_t)b * w_bs + (int64_t)ch * ldw + kc];
        v[j] = x;
    }
    *reinterpret_cast<u32x4*>(frag + (int64_t)b * frag_bs + e * 8) = pack8(v);
}

int pk_prep_at_launch(const float* w, int64_t w_bs, int ldw, int transA, int M, int mseg, int mseg_pad, int Ksrc, int kseg,
                      int kseg_pad, int kpos0, int Krange, int ntiles, int KS, bf16_t* frag, int64_t frag_bs, int nb,
                      hipStream_t stream) {
    const int64_t n = (int64_t)ntiles * KS * 64;
    hipLaunchKernelGGL(pk_prep_at_kernel, dim3((unsigned)cdiv(n, 256), nb), dim3(256), 0, stream, w, w_bs, ldw, transA, M, mseg,
                       mseg_pad, Ksrc, kseg, kseg_pad, kpos0, Krange, ntiles, KS, frag, frag_bs);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// several shared-weight fragment sets in one launch (blockIdx.y = job)
__global__ void __launch_bounds__(256)
pk_prep_at_jobs_kernel(const PrepAtJobs jobs) {
    const PrepAtJob& q = jobs.job[blockIdx.y];
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)q.ntiles * q.KS * 64) return;
    const int lane = (int)(e & 63), ks = (int)((e >> 6) % q.KS), mt = (int)((e >> 6) / q.KS);
    const int rr = lane & 15;
    const int pos = 32 * (mt >> 1) + 8 * (rr >> 2) + 4 * (mt & 1) + (rr & 3);
    const int ch = src_channel(pos, q.mseg, q.mseg_pad, q.M);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kp = 32 * ks + 8 * (lane >> 4) + j;
        const int kc = kp < q.Krange ? src_channel(q.kpos0 + kp, q.kseg, q.kseg_pad, q.Ksrc) : -1;
        float x = 0.f;
        if (ch >= 0 && kc >= 0) x = q.transA ? q.w[(int64_t)kc * q.ldw + ch] : q.w[(int64_t)ch * q.ldw + kc];
        v[j] = x;
    }
    *reinterpret_cast<u32x4*>(q.frag + e * 8) = pack8(v);
}
int pk_prep_at_jobs_launch(const PrepAtJobs& jobs, hipStream_t stream) {
    RST_REQUIRE(jobs.n > 0 && jobs.n <= 8, "bf16 weight preparation: %d jobs", jobs.n);
    int64_t most = 0;
    for (int i = 0; i < jobs.n; ++i) {
        const int64_t n = (int64_t)jobs.job[i].ntiles * jobs.job[i].KS * 64;
        most = most > n ? most : n;
    }
    hipLaunchKernelGGL(pk_prep_at_jobs_kernel, dim3((unsigned)cdiv(most, 256), (unsigned)jobs.n), dim3(256), 0, stream, jobs);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// ---- products over the pixels -------------------------------------------------------------------------------
// out[b][split][m][k] = sum over the pixels of the split of A[b][m][px] * Bm[b][k][px].  A wave takes a T x T block
// of 16 x 16 tiles, 32 pixels at a time (gram_pk_kernel's scheme: 16-byte chunks into a wave-private LDS image,
// back transposed through ds_read_b64_tr_b16).  Channel blocks past the end of a tensor read as zero.
template <int T>
__global__ void __launch_bounds__(256)
wgrad_pk_kernel(const WgradPk g) {
    constexpr int CBS = 576;
    constexpr int IMG = 2 * T * CBS;
    __shared__ __attribute__((aligned(16))) char lds[4][2 * IMG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t id = (int64_t)blockIdx.x * 4 + wave;
    const int nsub = g.nsa * g.nsb;
    const int sub = (int)(id % nsub); id /= nsub;
    const int split = (int)(id % g.splits);
    const int b = (int)(id / g.splits);
    if (b >= g.B) return;                              // no barrier in this kernel
    const int sm = sub / g.nsb, sn = sub % g.nsb;
    const int64_t N = g.N;
    const int a0 = sm * 2 * T, b0 = sn * 2 * T;
    const int an = g.Ab - a0, bn = g.Bb - b0;          // channel blocks that exist from this wave's first one
    const u32x4* Ap = reinterpret_cast<const u32x4*>(g.A + (int64_t)b * g.a_bs) + (int64_t)a0 * N;
    const u32x4* Bp = reinterpret_cast<const u32x4*>(g.Bm + (int64_t)b * g.b_bs) + (int64_t)b0 * N;
    const int64_t p0 = (int64_t)split * g.chunk_px;
    const int64_t p1 = p0 + g.chunk_px < N ? p0 + g.chunk_px : N;
    char* img = lds[wave];
    const uint32_t img_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)img;

    f32x4 acc[T][T], accs[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
        accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < T; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int l16 = lane & 15, gq = lane >> 4;
    const int tq = l16 >> 2, tp = l16 & 3;
    const uint32_t tr_off = (uint32_t)((tp >> 1) * CBS + (8 * gq + tq) * 16 + (tp & 1) * 8);
    const bool sums = g.with_rowsum != 0 && sn == 0;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 one8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, one8);

    u32x4 ra[T], rb[T];
    // where this lane's chunks of B start: channel block b0 + cbl, or (dense 3x3 taps) a shifted view of a real block
    int64_t boff[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int cbl = (i * 64 + lane) >> 5;
        boff[i] = (int64_t)cbl * N;
        if (g.tap_wp > 0) {
            const int v = b0 + cbl, tap = v / g.tap_cb, cb = v % g.tap_cb;
            boff[i] = (int64_t)(cb - b0) * N + (int64_t)(tap / 3 - 1) * g.tap_wp + (tap % 3 - 1);
        }
    }
    auto gload = [&](int64_t n0) {
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int idx = i * 64 + lane, cbl = idx >> 5;
            const int64_t px = n0 + (idx & 31);
            const u32x4 z = {0u, 0u, 0u, 0u};
            ra[i] = (px < p1 && cbl < an) ? Ap[(int64_t)cbl * N + px] : z;
            rb[i] = (px < p1 && cbl < bn) ? Bp[boff[i] + px] : z;
        }
    };
    gload(p0);
    for (int64_t n0 = p0; n0 < p1; n0 += 32) {
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int idx = i * 64 + lane;
            const int off = (idx >> 5) * CBS + (idx & 31) * 16;
            *reinterpret_cast<u32x4*>(img + off) = ra[i];
            *reinterpret_cast<u32x4*>(img + IMG + off) = rb[i];
        }
        gload(n0 + 32);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bf16x8 fa[T], fb[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            s16x4 q0, q1, k0, k1;
            const uint32_t aq = img_addr + tr_off + t * 2 * CBS, ak = aq + IMG;
            asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:64\n\t"
                         "ds_read_b64_tr_b16 %2, %5\n\tds_read_b64_tr_b16 %3, %5 offset:64\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1), "=&v"(k0), "=&v"(k1) : "v"(aq), "v"(ak) : "memory");
            const s16x8 qq = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
            const s16x8 kk = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
            fa[t] = __builtin_bit_cast(bf16x8, qq);
            fb[t] = __builtin_bit_cast(bf16x8, kk);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < T; ++i) {
#pragma unroll
            for (int j = 0; j < T; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (sums) accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accs[i], 0, 0, 0);
        }
    }
    const int M = g.Ab * 8, K = g.Bb * 8;
    const int64_t slab = (int64_t)b * g.splits + split;
    float* out = g.partial + slab * ((int64_t)M * K + (g.with_rowsum ? M : 0));     // [M][K] then the M row sums
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (sm * T + i) * 16 + 4 * gq + r, col = (sn * T + j) * 16 + l16;
                if (row < M && col < K) out[(int64_t)row * K + col] = acc[i][j][r];
            }
    if (sums && l16 == 0) {
        float* os = out + (int64_t)M * K;
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (sm * T + i) * 16 + 4 * gq + r;
                if (row < M) os[row] = accs[i][r];
            }
    }
}

void wgrad_pk_plan(WgradPk* g) {
    const int ta = (int)cdiv(g->Ab, 2), tb = (int)cdiv(g->Bb, 2);
    int best = 1;
    int64_t best_cost = INT64_MAX;
    for (int T = 3; T >= 1; --T) {
        // padded tile count, plus what re-reading the operands costs with small blocks
        const int64_t pa = round_up(ta, T), pb = round_up(tb, T);
        const int64_t cost = pa * pb * (T + 1) / T;
        if (cost < best_cost) { best_cost = cost; best = T; }
    }
    g->T = best;
    g->nsa = (int)cdiv(ta, best); g->nsb = (int)cdiv(tb, best);
    int64_t want = cdiv(2048, (int64_t)g->B * g->nsa * g->nsb);
    if (want < 1) want = 1;
    int64_t ch = round_up(cdiv(g->N, want), 32);
    if (ch < 256) ch = 256;
    g->chunk_px = ch;
    g->splits = (int)cdiv(g->N, ch);
}
size_t wgrad_pk_partial_floats(const WgradPk& g) {
    return (size_t)g.B * g.splits * ((size_t)g.Ab * 8 * g.Bb * 8 + (g.with_rowsum ? (size_t)g.Ab * 8 : 0));
}

int wgrad_pk_launch(const WgradPk& g, hipStream_t stream) {
    RST_REQUIRE(g.A && g.Bm && g.partial && g.Ab > 0 && g.Bb > 0 && g.B > 0 && g.N > 0, "bf16 weight gradient: bad operand");
    RST_REQUIRE(aligned16(g.A) && aligned16(g.Bm), "bf16 weight gradient: operands must be 16-byte aligned");
    RST_REQUIRE(g.T >= 1 && g.T <= 3 && g.splits > 0, "bf16 weight gradient: no plan");
    const int64_t items = (int64_t)g.nsa * g.nsb * g.splits * g.B;
    RST_REQUIRE(cdiv(items, 4) < (1ll << 31), "bf16 weight gradient: grid too large");
    const double px = (double)g.N * g.B;
    ProfScope prof(stream, "pk_wgrad", 2.0 * px * g.Ab * 8 * g.Bb * 8, px * 2.0 * 8 * ((double)g.Ab * g.nsb + (double)g.Bb * g.nsa));
    const dim3 grid((unsigned)cdiv(items, 4));
    if (g.T == 3) hipLaunchKernelGGL(wgrad_pk_kernel<3>, grid, dim3(256), 0, stream, g);
    else if (g.T == 2) hipLaunchKernelGGL(wgrad_pk_kernel<2>, grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL(wgrad_pk_kernel<1>, grid, dim3(256), 0, stream, g);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// ---- depthwise 3x3: tap sums --------------------------------------------------------------------------------
// partial[(b, row chunk)][cb][80]: [tap][8 channels] sums of dy[p] x[p + tap] over the chunk's pixels, then the 8 sums of dy.
// A thread takes runs of 8 pixels of a row: the 10 chunks of x a row of taps needs are read once for the run
// (38 16-byte loads per 8 pixels instead of 80), out-of-image chunks as zeros through the buffer descriptor's range.
__global__ void __launch_bounds__(256)
dw_wgrad_pk_kernel(const DwWgradPk a) {
    __shared__ float red[4][80];
    __shared__ float tr[4][40][65];
    const int cb = blockIdx.y, b = blockIdx.z;
    const int H = a.H, W = a.W;
    const int r0 = blockIdx.x * a.rows;
    const int r1 = r0 + a.rows < H ? r0 + a.rows : H;
    const int64_t plane = (int64_t)H * W;
    const __amdgpu_buffer_rsrc_t X = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(a.x + (int64_t)b * a.x_bs + (int64_t)cb * plane * 8), 0, (int)(plane * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t D = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(a.dy + (int64_t)b * a.dy_bs + (int64_t)cb * plane * 8), 0, (int)(plane * 16), 0x00020000);
    constexpr int OOR = (int)0x80000000;             // an offset no buffer reaches: the load returns zeros
    float acc[80];
#pragma unroll
    for (int k = 0; k < 80; ++k) acc[k] = 0.f;
    const int runs_w = (W + 7) >> 3;
    const int nruns = (r1 - r0) * runs_w;
    for (int idx = threadIdx.x; idx < nruns; idx += 256) {
        const int y = r0 + idx / runs_w, x0 = (idx % runs_w) * 8;
        u32x4 dq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            dq[i] = __builtin_amdgcn_raw_buffer_load_b128(D, x0 + i < W ? (y * W + x0 + i) * 16 : OOR, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float d[8];
            unpack8(dq[i], d);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[72 + j] += d[j];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            const bool row_in = yy >= 0 && yy < H;
            u32x4 xq[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int xx = x0 + i - 1;
                xq[i] = __builtin_amdgcn_raw_buffer_load_b128(X, row_in && xx >= 0 && xx < W ? (yy * W + xx) * 16 : OOR, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float d[8];
                unpack8(dq[i], d);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float v[8];
                    unpack8(xq[i + kx], v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[(ky * 3 + kx) * 8 + j] = fmaf(d[j], v[j], acc[(ky * 3 + kx) * 8 + j]);
                }
            }
        }
    }
    // sums over the block: 40 values at a time through an LDS image [wave][value][lane] (each lane then adds one row)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int k = 0; k < 40; ++k) tr[wave][k][lane] = acc[40 * half + k];
        __syncthreads();
        if (threadIdx.x < 160) {
            const int w = threadIdx.x / 40, k = threadIdx.x % 40;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
            for (int l = 0; l < 64; l += 2) { s0 += tr[w][k][l]; s1 += tr[w][k][l + 1]; }
            red[w][40 * half + k] = s0 + s1;
        }
        __syncthreads();
    }
    if (threadIdx.x < 80) {
        const int k = threadIdx.x;
        float* out = a.partial + (((int64_t)b * gridDim.x + blockIdx.x) * a.Cb + cb) * 80;
        out[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
}

int dw_wgrad_pk_rows(int B, int Cb, int H, int W) {
    // about 8 pixels per thread: the block-wide sums at the end cost as much as a few pixels' taps
    int64_t rows = cdiv(2048, W);
    if (rows > H) rows = H;
    return (int)rows;
}
size_t dw_wgrad_pk_partial_floats(int B, int Cb, int H, int W) {
    return (size_t)B * cdiv(H, dw_wgrad_pk_rows(B, Cb, H, W)) * Cb * 80;
}
// sums over the slabs land in red[Cb][80]
int dw_wgrad_pk_launch(DwWgradPk a, float* red, hipStream_t stream) {
    RST_REQUIRE(a.dy && a.x && a.partial && red && a.B > 0 && a.Cb > 0 && a.H > 0 && a.W > 0, "bf16 depthwise gradient: bad operand");
    RST_REQUIRE(a.B <= 65535 && a.Cb <= 65535, "bf16 depthwise gradient: grid too large");
    RST_REQUIRE((int64_t)a.H * a.W < (1ll << 27), "bf16 depthwise gradient: plane beyond a buffer descriptor's 2 GiB");
    a.rows = dw_wgrad_pk_rows(a.B, a.Cb, a.H, a.W);
    const int nchunks = (int)cdiv(a.H, a.rows);
    {
        const double px = (double)a.B * a.H * a.W * a.Cb * 8;
        ProfScope prof(stream, "pk_dw_wgrad", 20.0 * px, 2.0 * 2.0 * px);
        hipLaunchKernelGGL(dw_wgrad_pk_kernel, dim3((unsigned)nchunks, (unsigned)a.Cb, (unsigned)a.B), dim3(256), 0, stream, a);
        RST_LAUNCH_CHECK();
    }
    return reduce_slabs_launch(a.partial, red, (int64_t)a.Cb * 80, 1, a.B * nchunks, 0, stream);
}

// red[Cb][80] -> dweight[C][9], dbias[C] (null: none) through the position -> channel map of the packed buffer
__global__ void __launch_bounds__(256)
dw_grad_unpack_kernel(const float* __restrict__ red, float* __restrict__ dweight, float* __restrict__ dbias, int C, int seg,
                      int seg_pad, int Cb) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Cb * 80) return;
    const int cb = e / 80, r = e % 80, tap = r / 8, j = r % 8;
    const int ch = src_channel(cb * 8 + j, seg, seg_pad, C);
    if (ch < 0) return;
    if (tap < 9) dweight[ch * 9 + tap] = red[e];
    else if (dbias != nullptr) dbias[ch] = red[e];
}
int dw_grad_unpack_launch(const float* red, float* dweight, float* dbias, int C, int seg, int seg_pad, int Cb, hipStream_t stream) {
    hipLaunchKernelGGL(dw_grad_unpack_kernel, dim3((unsigned)cdiv(Cb * 80, 256)), dim3(256), 0, stream, red, dweight, dbias, C, seg, seg_pad, Cb);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// ---- gate ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bw_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float bw_gelu_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// d: [B][2 Gb][N] chunks {x1[0..3], x2[0..3]}; g: [B][Gb][N]: g block j = gelu(x1) * x2 of d blocks 2j, 2j + 1
__global__ void __launch_bounds__(256)
gate_fwd_pk_kernel(const bf16_t* __restrict__ d, bf16_t* __restrict__ g, int Gb, int64_t N) {
    const int64_t px = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (px >= N) return;
    const int j = blockIdx.y, b = blockIdx.z;
    const u32x4* D = reinterpret_cast<const u32x4*>(d) + ((int64_t)b * 2 * Gb + 2 * j) * N + px;
    float lo[8], hi[8], o[8];
    unpack8(D[0], lo);
    unpack8(D[N], hi);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = bw_gelu(lo[i]) * lo[4 + i]; o[4 + i] = bw_gelu(hi[i]) * hi[4 + i]; }
    reinterpret_cast<u32x4*>(g)[((int64_t)b * Gb + j) * N + px] = pack8(o);
}
__global__ void __launch_bounds__(256)
gate_bwd_pk_kernel(const bf16_t* __restrict__ dg, const bf16_t* __restrict__ d, bf16_t* __restrict__ dd, int Gb, int64_t N) {
    const int64_t px = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (px >= N) return;
    const int j = blockIdx.y, b = blockIdx.z;
    const int64_t off = ((int64_t)b * 2 * Gb + 2 * j) * N + px;
    const u32x4* D = reinterpret_cast<const u32x4*>(d) + off;
    float lo[8], hi[8], gg[8], olo[8], ohi[8];
    unpack8(D[0], lo);
    unpack8(D[N], hi);
    unpack8(reinterpret_cast<const u32x4*>(dg)[((int64_t)b * Gb + j) * N + px], gg);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        olo[i] = gg[i] * lo[4 + i] * bw_gelu_grad(lo[i]);
        olo[4 + i] = gg[i] * bw_gelu(lo[i]);
        ohi[i] = gg[4 + i] * hi[4 + i] * bw_gelu_grad(hi[i]);
        ohi[4 + i] = gg[4 + i] * bw_gelu(hi[i]);
    }
    u32x4* O = reinterpret_cast<u32x4*>(dd) + off;
    O[0] = pack8(olo);
    O[N] = pack8(ohi);
}
static int gate_grid(int B, int Gb, int64_t N, dim3* grid) {
    RST_REQUIRE(B > 0 && Gb > 0 && N > 0 && B <= 65535 && Gb <= 65535 && cdiv(N, 256) < (1ll << 31), "bf16 gate: bad shape");
    *grid = dim3((unsigned)cdiv(N, 256), (unsigned)Gb, (unsigned)B);
    return RST_OK;
}
int gate_fwd_pk_launch(const bf16_t* d, bf16_t* g, int B, int Gb, int64_t N, hipStream_t stream) {
    dim3 grid;
    RST_TRY(gate_grid(B, Gb, N, &grid));
    ProfScope prof(stream, "pk_gate", 30.0 * B * Gb * 8 * N, 2.0 * 3.0 * B * Gb * 8 * N);
    hipLaunchKernelGGL(gate_fwd_pk_kernel, grid, dim3(256), 0, stream, d, g, Gb, N);
    RST_LAUNCH_CHECK();
    return RST_OK;
}
int gate_bwd_pk_launch(const bf16_t* dg, const bf16_t* d, bf16_t* dd, int B, int Gb, int64_t N, hipStream_t stream) {
    dim3 grid;
    RST_TRY(gate_grid(B, Gb, N, &grid));
    ProfScope prof(stream, "pk_gate_bwd", 60.0 * B * Gb * 8 * N, 2.0 * 5.0 * B * Gb * 8 * N);
    hipLaunchKernelGGL(gate_bwd_pk_kernel, grid, dim3(256), 0, stream, dg, d, dd, Gb, N);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// ---- channel LayerNorm, LPP lanes per pixel -------------------------------------------------------------------
// Lane `sub` of a pixel's LPP lanes takes the channel blocks sub, sub + LPP, ...; the sums over the channels meet
// through xor shuffles.  LPP = 1 when there are enough pixels to fill the chip with one thread each, 8 otherwise.
template <int LPP>
__device__ __forceinline__ float lpp_sum(float v) {
#pragma unroll
    for (int o = 1; o < LPP; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// y = (x - mu) rstd  (with_bias)  or  x rstd  (bias-free form, model/Restromer.py:40-52): the normalised tensor
// before the affine part
template <int LPP>
__global__ void __launch_bounds__(256)
ln_fwd_pk_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int Cb, int64_t N, int with_bias) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t px0 = gid / LPP;
    const int sub = (int)(gid % LPP);
    const bool live = px0 < N;
    const int64_t px = live ? px0 : N - 1;          // idle lanes shadow the last pixel: every lane reaches the shuffles
    const int b = blockIdx.y;
    const u32x4* X = reinterpret_cast<const u32x4*>(x) + (int64_t)b * Cb * N + px;
    u32x4* Y = reinterpret_cast<u32x4*>(y) + (int64_t)b * Cb * N + px;
    float s1 = 0.f, s2 = 0.f;
    for (int cb = sub; cb < Cb; cb += LPP) {
        float v[8];
        unpack8(X[(int64_t)cb * N], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1 += v[j]; s2 = fmaf(v[j], v[j], s2); }
    }
    s1 = lpp_sum<LPP>(s1); s2 = lpp_sum<LPP>(s2);
    const float invC = 1.0f / (float)(Cb * 8);
    const float mu = s1 * invC;
    const float rstd = 1.0f / sqrtf(fmaxf(s2 * invC - mu * mu, 0.f) + 1e-5f);
    const float sh = with_bias ? mu : 0.f;
    if (!live) return;
    for (int cb = sub; cb < Cb; cb += LPP) {
        float v[8];
        unpack8(X[(int64_t)cb * N], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (v[j] - sh) * rstd;
        Y[(int64_t)cb * N] = pack8(v);
    }
}
// dx = addend + d(LayerNorm)/dx applied to dn, dn = gradient with respect to gamma xhat + beta
template <int LPP>
__global__ void __launch_bounds__(256)
ln_bwd_pk_kernel(const bf16_t* __restrict__ dn, const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                 const bf16_t* __restrict__ addend, bf16_t* __restrict__ dx, int Cb, int64_t N, int with_bias) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t px0 = gid / LPP;
    const int sub = (int)(gid % LPP);
    const bool live = px0 < N;
    const int64_t px = live ? px0 : N - 1;
    const int b = blockIdx.y;
    const int64_t base = (int64_t)b * Cb * N + px;
    const u32x4* X = reinterpret_cast<const u32x4*>(x) + base;
    const u32x4* G = reinterpret_cast<const u32x4*>(dn) + base;
    const u32x4* A = addend != nullptr ? reinterpret_cast<const u32x4*>(addend) + base : nullptr;
    u32x4* O = reinterpret_cast<u32x4*>(dx) + base;
    float s1 = 0.f, s2 = 0.f;
    for (int cb = sub; cb < Cb; cb += LPP) {
        float v[8];
        unpack8(X[(int64_t)cb * N], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1 += v[j]; s2 = fmaf(v[j], v[j], s2); }
    }
    s1 = lpp_sum<LPP>(s1); s2 = lpp_sum<LPP>(s2);
    const float invC = 1.0f / (float)(Cb * 8);
    const float mu = s1 * invC;
    const float rstd = 1.0f / sqrtf(fmaxf(s2 * invC - mu * mu, 0.f) + 1e-5f);
    float g1 = 0.f, g2 = 0.f;                 // sum_c g, sum_c g xc  with g = dn gamma, xc = (x - mu) rstd
    for (int cb = sub; cb < Cb; cb += LPP) {
        float v[8], g[8];
        unpack8(X[(int64_t)cb * N], v);
        unpack8(G[(int64_t)cb * N], g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gw = g[j] * gamma[cb * 8 + j];
            g1 += gw;
            g2 = fmaf(gw, (v[j] - mu) * rstd, g2);
        }
    }
    g1 = lpp_sum<LPP>(g1); g2 = lpp_sum<LPP>(g2);
    // with bias:  dx = rstd (g - g1/C - xc g2/C);   bias-free (y = x rstd gamma):  dx = rstd (g - xc (g2 + mu rstd g1)/C)
    const float k1 = with_bias ? g1 * invC : 0.f;
    const float k2 = with_bias ? g2 * invC : (g2 + mu * rstd * g1) * invC;
    if (!live) return;
    for (int cb = sub; cb < Cb; cb += LPP) {
        float v[8], g[8], o[8];
        unpack8(X[(int64_t)cb * N], v);
        unpack8(G[(int64_t)cb * N], g);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rstd * (g[j] * gamma[cb * 8 + j] - k1 - (v[j] - mu) * rstd * k2);
        if (A != nullptr) {
            float r[8];
            unpack8(A[(int64_t)cb * N], r);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += r[j];
        }
        O[(int64_t)cb * N] = pack8(o);
    }
}
static int ln_pk_lanes(int B, int Cb, int64_t N) { return (Cb >= 8 && (int64_t)B * N <= 262144) ? 8 : 1; }
int ln_fwd_pk_launch(const bf16_t* x, bf16_t* y, int B, int Cb, int64_t N, int with_bias, hipStream_t stream) {
    RST_REQUIRE(x && y && B > 0 && B <= 65535 && Cb > 0 && N > 0, "bf16 LayerNorm: bad operand");
    ProfScope prof(stream, "pk_ln", 6.0 * B * Cb * 8 * N, 2.0 * 2.0 * B * Cb * 8 * N);
    const int lpp = ln_pk_lanes(B, Cb, N);
    const dim3 grid((unsigned)cdiv(N * lpp, 256), (unsigned)B);
    if (lpp == 8) hipLaunchKernelGGL(ln_fwd_pk_kernel<8>, grid, dim3(256), 0, stream, x, y, Cb, N, with_bias);
    else hipLaunchKernelGGL(ln_fwd_pk_kernel<1>, grid, dim3(256), 0, stream, x, y, Cb, N, with_bias);
    RST_LAUNCH_CHECK();
    return RST_OK;
}
int ln_bwd_pk_launch(const bf16_t* dn, const bf16_t* x, const float* gamma, const bf16_t* addend, bf16_t* dx, int B, int Cb,
                     int64_t N, int with_bias, hipStream_t stream) {
    RST_REQUIRE(dn && x && gamma && dx && B > 0 && B <= 65535 && Cb > 0 && N > 0, "bf16 LayerNorm backward: bad operand");
    ProfScope prof(stream, "pk_ln_bwd", 14.0 * B * Cb * 8 * N, 2.0 * (addend ? 4.0 : 3.0) * B * Cb * 8 * N);
    const int lpp = ln_pk_lanes(B, Cb, N);
    const dim3 grid((unsigned)cdiv(N * lpp, 256), (unsigned)B);
    if (lpp == 8) hipLaunchKernelGGL(ln_bwd_pk_kernel<8>, grid, dim3(256), 0, stream, dn, x, gamma, addend, dx, Cb, N, with_bias);
    else hipLaunchKernelGGL(ln_bwd_pk_kernel<1>, grid, dim3(256), 0, stream, dn, x, gamma, addend, dx, Cb, N, with_bias);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// ---- small fp32 kernels -------------------------------------------------------------------------------------
// position of source channel ch in a segmented packed buffer (the inverse of src_channel)
__device__ __forceinline__ int dst_position(int ch, int seg, int seg_pad) {
    if (seg_pad < 0) {
        const int half = ch / seg, i = ch % seg;
        return 8 * (i >> 2) + 4 * half + (i & 3);
    }
    if (seg >= (1 << 30)) return ch;
    return (ch / seg) * seg_pad + ch % seg;
}

// A 1x1 convolution behind a LayerNorm, y = W (gamma xhat + beta) + bias, from G[pos][c] = sum_px dy[pos] xhat[c] and
// S[pos] = sum_px dy[pos]:  dW[m][c] = gamma[c] G[pos(m)][c] + beta[c] S[pos(m)],  dbias[m] = S[pos(m)]
__global__ void __launch_bounds__(256)
lnlin_dw_kernel(const float* __restrict__ G, const float* __restrict__ S, const float* __restrict__ gamma,
                const float* __restrict__ beta, float* __restrict__ dW, float* __restrict__ dbias, int M, int C, int seg, int seg_pad) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)M * C) return;
    const int m = (int)(e / C), c = (int)(e % C);
    const int pos = dst_position(m, seg, seg_pad);
    float v = gamma[c] * G[(int64_t)pos * C + c];
    if (beta != nullptr) v = fmaf(beta[c], S[pos], v);
    dW[e] = v;
    if (dbias != nullptr && c == 0) dbias[m] = S[pos];
}
// dgamma[c] = sum_m W[m][c] G[pos(m)][c],  dbeta[c] = sum_m W[m][c] S[pos(m)]; 16 columns x 64 row phases per block
// (a block per 16 columns is all the parallelism there is across blocks -- 3 to 24 of them -- so the rows are spread
// over 1024 threads: the kernel is a chain of dependent cache misses per thread, 7.7 us on average with 16 phases)
constexpr int LNLIN_PH = 64;
__global__ void __launch_bounds__(16 * LNLIN_PH)
lnlin_dgamma_kernel(const float* __restrict__ G, const float* __restrict__ S, const float* __restrict__ W,
                    float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int C, int seg, int seg_pad) {
    __shared__ float red[2][LNLIN_PH][17];
    const int cl = threadIdx.x & 15, ph = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float a = 0.f, bsum = 0.f;
    if (c < C) {
        // four rows in flight per thread
        float a4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
        int m = ph;
        for (; m + 3 * LNLIN_PH < M; m += 4 * LNLIN_PH) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int mm = m + LNLIN_PH * u;
                const int pos = dst_position(mm, seg, seg_pad);
                const float w = W[(int64_t)mm * C + c];
                a4[u] = fmaf(w, G[(int64_t)pos * C + c], a4[u]);
                b4[u] = fmaf(w, S[pos], b4[u]);
            }
        }
        for (; m < M; m += LNLIN_PH) {
            const int pos = dst_position(m, seg, seg_pad);
            const float w = W[(int64_t)m * C + c];
            a4[0] = fmaf(w, G[(int64_t)pos * C + c], a4[0]);
            b4[0] = fmaf(w, S[pos], b4[0]);
        }
        a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        bsum = (b4[0] + b4[1]) + (b4[2] + b4[3]);
    }
    red[0][ph][cl] = a; red[1][ph][cl] = bsum;
    __syncthreads();
    if (ph == 0 && c < C) {
        float ga = 0.f, gb = 0.f;
#pragma unroll 8
        for (int i = 0; i < LNLIN_PH; ++i) { ga += red[0][i][cl]; gb += red[1][i][cl]; }
        dgamma[c] = ga;
        if (dbeta != nullptr) dbeta[c] = gb;
    }
}
int lnlin_grads_launch(const float* G, const float* S, const float* W, const float* gamma, const float* beta, float* dW,
                       float* dbias, float* dgamma, float* dbeta, int M, int C, int seg, int seg_pad, hipStream_t stream) {
    RST_REQUIRE(G && S && W && gamma && dW && dgamma && M > 0 && C > 0, "bf16 LayerNorm + 1x1 gradients: bad operand");
    hipLaunchKernelGGL(lnlin_dw_kernel, dim3((unsigned)cdiv((int64_t)M * C, 256)), dim3(256), 0, stream, G, S, gamma, beta, dW, dbias, M, C, seg, seg_pad);
    RST_LAUNCH_CHECK();
    hipLaunchKernelGGL(lnlin_dgamma_kernel, dim3((unsigned)cdiv(C, 16)), dim3(16 * LNLIN_PH), 0, stream, G, S, W, dgamma, dbeta, M, C, seg, seg_pad);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// out[r][c] = in[r * ld_in + c]
__global__ void __launch_bounds__(256)
crop2d_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out, int rows, int cols) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)rows * cols) return;
    out[e] = in[(e / cols) * ld_in + e % cols];
}
int crop2d_launch(const float* in, int ld_in, float* out, int rows, int cols, hipStream_t stream) {
    hipLaunchKernelGGL(crop2d_kernel, dim3((unsigned)cdiv((int64_t)rows * cols, 256)), dim3(256), 0, stream, in, ld_in, out, rows, cols);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// The 2C x 2C matrix per sample of the attention's q / k gradients,
//   [dq; dk] = [[diag(Dq), BD(Eq)], [BD(Eq)^T, diag(Dk)]] [q; k]      (BD: block diagonal over the heads)
__global__ void __launch_bounds__(256)
attn_bwd_mats_kernel(const float* __restrict__ Eq, const float* __restrict__ Dq, const float* __restrict__ Dk,
                     float* __restrict__ out, int B, int C, int c) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int C2 = 2 * C;
    const int64_t cc = (int64_t)C2 * C2;
    if (e >= (int64_t)B * cc) return;
    const int col = (int)(e % C2), row = (int)((e / C2) % C2), b = (int)(e / cc);
    const int heads = C / c;
    const int r = row % C, k = col % C, hr = r / c, hk = k / c;
    const bool lower = row >= C, right = col >= C;
    float v = 0.f;
    if (lower == right) { if (r == k) v = (lower ? Dk : Dq)[(int64_t)b * C + r]; }
    else if (hr == hk) {
        const float* E = Eq + ((int64_t)b * heads + hr) * c * c;
        v = lower ? E[(k % c) * c + r % c] : E[(r % c) * c + k % c];
    }
    out[e] = v;
}
int attn_bwd_mats_launch(const float* Eq, const float* Dq, const float* Dk, float* out, int B, int C, int c, hipStream_t stream) {
    hipLaunchKernelGGL(attn_bwd_mats_kernel, dim3((unsigned)cdiv((int64_t)B * 4 * C * C, 256)), dim3(256), 0, stream, Eq, Dq, Dk, out, B, C, c);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst
