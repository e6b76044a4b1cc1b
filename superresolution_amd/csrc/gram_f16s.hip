// Gram matrices q k^T and the squared row norms of q and k over the pixels (model/Restromer.py:136-139: the
// L2-normalised attention logits are these sums) on the f16 matrix instructions at fp32 accuracy.
//
// The contraction runs over the pixels, and NCHW fp32 is pixel-contiguous: lane (row, group g) of an MFMA operand wants
// 8 consecutive pixels of one channel -- 32 contiguous bytes of q or k.  So both operands go from global memory
// straight into registers (two 16-byte loads per lane and tile, the next 32 pixels in flight), are split in registers
// (v = hi + lo / 2048, as gemm_f16s.hip) and multiplied with three v_mfma_f32_16x16x32_f16 per pair of tiles; no LDS,
// no transposition, no barrier.  The squared norms are summed on the vector pipe from the same registers.
// A block owns a T x T block of 16 x 16 tiles of one (sample, head) over one split of the pixels; its four waves take
// a quarter of the split each and add up through LDS at the end.  Splits and slab layout are gemm_nt_plan's, so the
// reduction and softmax kernels do not change.
// |q|, |k| must stay below 65504 (f16 range).
#include "kernels.hpp"

namespace rst {

namespace {

typedef _Float16 gr_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 gr_h2 __attribute__((ext_vector_type(2)));
typedef float gr_f32x2 __attribute__((ext_vector_type(2)));

// two blocks per CU: gemm_nt_plan aims at 1024 blocks = two full rounds of 512 (three per CU would leave a third round a third full)
template <int T>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
gram_f16s_kernel(const GemmNT g, int nsub, int walk) {
    __shared__ float red[3][(T * T * 4 + 2 * T) * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the nsub x nsub tile blocks of one (sample, head, split) read the same pixels: neighbours on one XCD
    int64_t id = walk ? xcd_walk(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
    const int sub = (int)(id % (nsub * nsub)); id /= nsub * nsub;
    const int split = (int)(id % g.splits);
    const int64_t bh = id / g.splits;
    const int sm = sub / nsub, sn = sub % nsub;
    const int b1 = (int)(bh / g.nb2), b2 = (int)(bh % g.nb2);
    const int l16 = lane & 15, gq = lane >> 4;
    const int64_t s0 = (int64_t)split * g.chunk;
    const int64_t s1 = s0 + g.chunk < g.N ? s0 + g.chunk : g.N;
    const int64_t quarter = ((s1 - s0 + 3) / 4 + 31) / 32 * 32;
    const int64_t p0 = s0 + wave * quarter < s1 ? s0 + wave * quarter : s1;
    const int64_t p1 = p0 + quarter < s1 ? p0 + quarter : s1;
    // this lane's row of tile 0 of each operand; of a step of 32 pixels lane group gq takes pixels 8 gq .. 8 gq + 7
    const float* Aq = g.A + (int64_t)b1 * g.a_s1 + (int64_t)b2 * g.a_s2 + (int64_t)(sm * T * 16 + l16) * g.lda;
    const float* Bk = g.Bm + (int64_t)b1 * g.b_s1 + (int64_t)b2 * g.b_s2 + (int64_t)(sn * T * 16 + l16) * g.ldb;

    f32x4 acc1[T][T], acc2[T][T];
    float sq[2][T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
        sq[0][i] = 0.f; sq[1][i] = 0.f;
#pragma unroll
        for (int j = 0; j < T; ++j) { acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }

    // The loop is  convert(step k) | issue the loads of step k + 1 | multiply(step k), in THAT order: the loads fly while
    // the matrix pipe works, and the (counted) wait for them sits at the next conversion.  Left alone the compiler moved
    // the conversion up to the loads and waited for them at once -- load latency + conversion + products in series, at
    // two waves per SIMD.  The loads have no branch: a lane group past the end of the wave's range reads its row at a
    // clamped pixel (valid memory) and is zeroed when converted, which only the last, partial step does.  (A second
    // register buffer -- the loads of two steps in flight -- spills at three tiles: 256 registers + 56.)
    f32x4 raw[2][T][2];
    auto gload = [&](int64_t px) __attribute__((always_inline)) {
        int64_t pl = px + 8 * gq;
        if (pl + 8 > g.N) pl = g.N - 8;                       // (N is a multiple of 8, at least 8)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const f32x4* a = reinterpret_cast<const f32x4*>(Aq + (int64_t)t * 16 * g.lda + pl);
            const f32x4* b = reinterpret_cast<const f32x4*>(Bk + (int64_t)t * 16 * g.ldb + pl);
            raw[0][t][0] = a[0]; raw[0][t][1] = a[1];
            raw[1][t][0] = b[0]; raw[1][t][1] = b[1];
        }
    };
    gr_h8 hi[2][T], lo[2][T];
    auto convert = [&](bool ok) __attribute__((always_inline)) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float s = sq[o][t];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const gr_f32x2 vv = {ok ? raw[o][t][h][2 * e] : 0.f, ok ? raw[o][t][h][2 * e + 1] : 0.f};
                        s = fmaf(vv[0], vv[0], fmaf(vv[1], vv[1], s));
                        const gr_h2 hh = __builtin_convertvector(vv, gr_h2);
                        const gr_f32x2 sc = vv * 2048.f;
                        const gr_f32x2 r2 = {fmaf((float)hh[0], -2048.f, sc[0]), fmaf((float)hh[1], -2048.f, sc[1])};
                        const gr_h2 ll = __builtin_convertvector(r2, gr_h2);
                        hi[o][t][4 * h + 2 * e] = hh[0]; hi[o][t][4 * h + 2 * e + 1] = hh[1];
                        lo[o][t][4 * h + 2 * e] = ll[0]; lo[o][t][4 * h + 2 * e + 1] = ll[1];
                    }
                sq[o][t] = s;
            }
    };
    auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j) {
                acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi[0][i], hi[1][j], acc1[i][j], 0, 0, 0);
                acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi[0][i], lo[1][j], acc2[i][j], 0, 0, 0);
                acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo[0][i], hi[1][j], acc2[i][j], 0, 0, 0);
            }
    };
    gload(p0);
    int64_t px = p0;
    for (; px + 32 <= p1; px += 32) {
        convert(true);
        __builtin_amdgcn_sched_barrier(0);
        if (px + 32 < p1) gload(px + 32);                     // (wave-uniform; a partial step: clamped, masked when converted)
        __builtin_amdgcn_sched_barrier(0);
        multiply();
    }
    if (px < p1) {
        convert(px + 8 * gq < p1);                            // whole groups of 8 pixels: chunk and N are multiples of 8
        multiply();
    }
    // per-lane results of this wave, then the sum over the four waves
    float v[T * T * 4 + 2 * T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[(i * T + j) * 4 + r] = fmaf(acc2[i][j][r], 1.0f / 2048.f, acc1[i][j][r]);
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float q = sq[o][t];                               // this lane's 8-pixel groups of row l16: add the four groups
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            v[T * T * 4 + o * T + t] = q;
        }
    constexpr int NV = T * T * 4 + 2 * T;
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) red[wave - 1][k * 64 + lane] = v[k];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] += (red[0][k * 64 + lane] + red[1][k * 64 + lane]) + red[2][k * 64 + lane];
    const int c = g.M1;
    const int64_t slab = bh * g.splits + split;
    float* out = g.partial + slab * c * c;
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(int64_t)((sm * T + i) * 16 + 4 * gq + r) * c + (sn * T + j) * 16 + l16] = v[(i * T + j) * 4 + r];
    if (gq == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (sn == 0) g.sqA_partial[slab * c + (sm * T + t) * 16 + l16] = v[T * T * 4 + t];
            if (sm == 0) g.sqB_partial[slab * c + (sn * T + t) * 16 + l16] = v[T * T * 4 + T + t];
        }
    }
}

}  // namespace

bool gram_f16s_supported(const GemmNT& g) {
    if (!gemm_f16s_enabled()) return false;
    if (g.sqA_partial == nullptr || g.sqB_partial == nullptr || g.M1 != g.M2 || g.M1 % 16 != 0 || g.M1 > 384) return false;
    const bool al = aligned16(g.A) && aligned16(g.Bm);
    return al && g.N % 8 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 && g.a_s1 % 4 == 0 && g.a_s2 % 4 == 0 && g.b_s1 % 4 == 0 &&
           g.b_s2 % 4 == 0 && g.chunk % 8 == 0;
}

int gram_f16s_launch(const GemmNT& g, hipStream_t stream) {
    RST_REQUIRE(gram_f16s_supported(g), "gram_f16s: unsupported operands");
    const int tiles = g.M1 / 16;
    const int T = tiles % 3 == 0 ? 3 : tiles % 2 == 0 ? 2 : 1;
    const int nsub = tiles / T;
    const int64_t items = (int64_t)g.nb1 * g.nb2 * g.splits * nsub * nsub;
    RST_REQUIRE(items < (1ll << 31), "gram_f16s: grid too large");
    const double nb = (double)g.nb1 * g.nb2;
    ProfScope prof(stream, "gram_f16s", 2.0 * g.M1 * (g.M2 + 2.0) * g.N * nb,
                   4.0 * nb * ((double)(g.M1 + g.M2) * g.N + (double)g.splits * g.M1 * g.M2));
    const dim3 grid((unsigned)items);
    const int walk = xcd_walk_enabled();
    if (T == 3) hipLaunchKernelGGL(gram_f16s_kernel<3>, grid, dim3(256), 0, stream, g, nsub, walk);
    else if (T == 2) hipLaunchKernelGGL(gram_f16s_kernel<2>, grid, dim3(256), 0, stream, g, nsub, walk);
    else hipLaunchKernelGGL(gram_f16s_kernel<1>, grid, dim3(256), 0, stream, g, nsub, walk);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst
