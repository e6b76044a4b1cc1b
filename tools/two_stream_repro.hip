// Library-free demonstration of a cross-dispatch hazard on MI355X (gfx950): a wave that issues
// v_mfma_f32_16x16x32_f16 / _bf16 back to back can make a wave of ANOTHER dispatch on the same compute unit lose
// lanes 48..63 of one multiply-add of a dependent chain (the next multiply-add of the chain reads the accumulator as it
// was before).  Ordering inside a stream is not involved: a rocprofv3 kernel trace of the two streams shows no kernel
// starting before its predecessor on the same queue has ended (tools/trace_overlap.py).
//
// Two streams, each with its own buffers:
//   writer  : GEMM-shaped: 512-thread blocks, 112 KiB of dynamic LDS, a 160 x 128 tile brought in by LDS-DMA, float4
//             row stores U[b][m][n] = X[b][m % C][n] + iter, blocks walked XCD by XCD; then `spin` rounds of LDS-DMA
//             re-loads + f16 hi/lo conversion of LDS operands + 36 matrix instructions into 24 accumulators (unused)
//   reader  : depthwise-shaped: 256-thread blocks, a lane owns a 4 x 4 strip, reads 6 rows (float4 + halo) and 9 taps
//             per half and runs 2 x 16 x 9 dependent multiply-adds: G = conv3x3(U[c]) + 2 conv3x3(U[c + HID])
//   check   : recomputes G per pixel from X and the taps alone (small integers: exact in fp32), counts mismatches
//
// usage: two_stream_repro ITERS STREAMS SPIN SPLIT FLAGS MFMA DMA
//   SPLIT 0: both streams run scribble -> writer -> reader -> check; N > 0: stream 0 only runs reader -> check on a
//            tensor written once, stream 1 runs a writer every N iterations; -1: ONE kernel whose blocks hold four
//            matrix waves and four multiply-add waves (never failed)
//   FLAGS 1: no matrix instructions, 2: no LDS-DMA inside the rounds, 4: no wait + barrier inside the rounds
//   MFMA  1: 16x16x32_f16, 2: 16x16x4_f32, 3: 16x16x16_f16, 4: 16x16x32_bf16, 0: a plain multiply-add instead
//   DMA   16 / 0: LDS-DMA width in bytes inside the rounds (0: none)
//   REPRO_CUMASK=1 puts the two streams on disjoint halves of the compute units.
// Measured (profiles/r3_two_stream_repro*.txt), 3000 iterations of `... 2 6 1 4 MFMA DMA`: wrong iterations of the
// reader 948 (16x16x32_f16, DMA 16), 1182 (16x16x32_f16, no DMA), 962 (16x16x32_bf16), 0 (16x16x4_f32), 0
// (16x16x16_f16), 0 (no matrix instruction); 0 with the streams on disjoint compute units; 0 with a wait + barrier in
// every round in this program (the library's 1x1 product, which has one per 32-channel chunk, does trigger it).
// The wrong values are always those of lanes 48..63 of a wave, one output register at a time.
// Build: hipcc -O3 --offload-arch=gfx950 two_stream_repro.hip -o bin/two_stream_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

constexpr int B = 8, C = 192, HID = 240, M = 2 * HID, H = 64, W = 64, N = H * W;
constexpr int ROWS_PER_BLOCK = 160, MBLOCKS = M / ROWS_PER_BLOCK, NTILES = N / 128;

typedef __attribute__((address_space(1))) const void* gptr;
typedef __attribute__((address_space(3))) void* lptr;

__device__ __forceinline__ long xcd_walk(long id, long total) {
    const long per = total / 8;               // total is a multiple of 8 here
    return (id % 8) * per + id / 8;
}

__global__ void scribble(float* U, long n, float v) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 4 < n) reinterpret_cast<float4*>(U)[i] = make_float4(v, v, v, v);
}

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
// MF: 1 v_mfma_f32_16x16x32_f16, 2 v_mfma_f32_16x16x4_f32, 3 v_mfma_f32_16x16x16_f16, 4 v_mfma_f32_16x16x32_bf16, 0 plain adds
template <int MF>
__device__ __forceinline__ f4_t mf(h8_t a, h8_t b, f4_t c) {
    if constexpr (MF == 1) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    else if constexpr (MF == 2) return __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[0], (float)b[0], c, 0, 0, 0);
    else if constexpr (MF == 3) return __builtin_amdgcn_mfma_f32_16x16x16f16(h4_t{a[0], a[1], a[2], a[3]}, h4_t{b[0], b[1], b[2], b[3]}, c, 0, 0, 0);
    else if constexpr (MF == 4) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8_t, a), __builtin_bit_cast(b8_t, b), c, 0, 0, 0);
    else { c[0] += (float)a[0] * (float)b[0]; return c; }
}

template <int MF, int DMA>
__global__ void __launch_bounds__(512) writer(const float* __restrict__ X, float* __restrict__ U, float iter, int spin, int flags) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const long lin = xcd_walk((long)blockIdx.y * gridDim.x + blockIdx.x, (long)gridDim.x * gridDim.y);
    const int bx = (int)(lin % gridDim.x), b = (int)(lin / gridDim.x);
    const int mb = bx % MBLOCKS, nt = bx / MBLOCKS;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // every wave brings 20 rows x 128 pixels in by LDS-DMA, 2 rows per instruction
    float* mine = lds + wave * 20 * 128;
    for (int p = 0; p < 10; ++p) {
        const int m = mb * ROWS_PER_BLOCK + wave * 20 + 2 * p + (lane >> 5);
        const float* src = X + ((long)b * C + (m % C)) * N + nt * 128 + (lane & 31) * 4;
        __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(mine + p * 256), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // matrix-pipe work of the shape of a split-f16 1x1 product: operands from LDS, converted to f16 hi/lo pairs,
    // 36 v_mfma_f32_16x16x32_f16 per round into 24 accumulators; the result only feeds a condition that is never true
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc1[6][2], acc2[6][2];
    for (int i = 0; i < 6; ++i) for (int t = 0; t < 2; ++t) { acc1[i][t] = f4{0.f, 0.f, 0.f, 0.f}; acc2[i][t] = f4{0.f, 0.f, 0.f, 0.f}; }
    const int lm = lane & 15, kk = lane >> 4;
    for (int round = 0; round < spin; ++round) {
        // the tile again by LDS-DMA, three pieces per wave and round, one round in flight (same bytes: harmless),
        // a counted wait and a block barrier per round as in the library's ring
        if (!(flags & 2))
        for (int p = 0; p < 3; ++p) {
            const int pp = (round * 3 + p) % 10;
            const int m = mb * ROWS_PER_BLOCK + wave * 20 + 2 * pp + (lane >> 5);
            const float* src = X + ((long)b * C + (m % C)) * N + nt * 128 + (lane & 31) * 4;
            if constexpr (DMA == 16) __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(mine + pp * 256), 16, 0, 0);
            if constexpr (DMA == 4) __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(mine + pp * 256), 4, 0, 0);
        }
        if (!(flags & 4)) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
        const float* xs = lds + ((round * 7 + wave) % 8) * 20 * 128 + kk * 128 + lm;
        h8 xh[2], xl[2];
        for (int t = 0; t < 2; ++t) {
            for (int j = 0; j < 4; ++j) {
                const f2 vv = {xs[(4 * j) * 128 + 16 * t], xs[(4 * j) * 128 + 64 + 16 * t]};
                const h2 h = __builtin_convertvector(vv, h2);
                const f2 sc = vv * 2048.f;
                const f2 rr2 = {fmaf((float)h[0], -2048.f, sc[0]), fmaf((float)h[1], -2048.f, sc[1])};
                const h2 l = __builtin_convertvector(rr2, h2);
                xh[t][2 * j] = h[0]; xh[t][2 * j + 1] = h[1]; xl[t][2 * j] = l[0]; xl[t][2 * j + 1] = l[1];
            }
        }
        const h8* af = reinterpret_cast<const h8*>(lds) + ((round * 3) % 16) * 64 + lane;
        if (!(flags & 1))
        for (int i = 0; i < 6; ++i) {
            const h8 ah = af[(i * 2 + 0) * 64], al = af[(i * 2 + 1) * 64];
            for (int t = 0; t < 2; ++t) {
                acc1[i][t] = mf<MF>(ah, xh[t], acc1[i][t]);
                acc2[i][t] = mf<MF>(ah, xl[t], acc2[i][t]);
                acc2[i][t] = mf<MF>(al, xh[t], acc2[i][t]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    float burn = 0.f;
    for (int i = 0; i < 6; ++i) for (int t = 0; t < 2; ++t) for (int r = 0; r < 4; ++r) burn += acc1[i][t][r] + acc2[i][t][r];
    for (int p = 0; p < 10; ++p) {
        const int r = 2 * p + (lane >> 5);
        const int m = mb * ROWS_PER_BLOCK + wave * 20 + r;
        float4 v = *reinterpret_cast<const float4*>(mine + p * 256 + lane * 4);
        const float add = iter + (burn == 12345.678f ? 1.f : 0.f);
        v.x += add; v.y += add; v.z += add; v.w += add;
        *reinterpret_cast<float4*>(U + ((long)b * M + m) * N + nt * 128 + (lane & 31) * 4) = v;
    }
}

__device__ __forceinline__ float box_row(const float* row, int x0, bool ok) {
    if (!ok) return 0.f;
    // sums of the three taps for the 4 pixels of the strip, added up: every tap position counted once per pixel
    float s = 0.f;
    for (int j = -1; j < 5; ++j) {
        const int x = x0 + j;
        const float v = (x >= 0 && x < W) ? row[x] : 0.f;
        const int cnt = (j == -1 || j == 4) ? 1 : (j == 0 || j == 3) ? 2 : 3;
        s += cnt * v;
    }
    return s;
}

// strip sum of a 3x3 box filter over a 4 x 4 strip: row yy contributes once per output row it touches
__device__ __forceinline__ float strip_box(const float* plane, int y0, int x0) {
    float s = 0.f;
    for (int r = -1; r < 5; ++r) {
        const int yy = y0 + r;
        const int cnt = (r == -1 || r == 4) ? 1 : (r == 0 || r == 3) ? 2 : 3;
        s += cnt * box_row(plane + (long)yy * W, x0, yy >= 0 && yy < H);
    }
    return s;
}

__device__ __forceinline__ void conv_half(float (&o)[4][4], const float* plane, int y0, int x0, const float* __restrict__ taps, int c) {
    float in[6][6];
    for (int r = 0; r < 6; ++r) {
        const int yy = y0 - 1 + r;
        const bool ok = yy >= 0 && yy < H;
        const float* row = plane + (long)yy * W;
        float4 cc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) cc = *reinterpret_cast<const float4*>(row + x0);
        in[r][0] = (ok && x0 > 0) ? row[x0 - 1] : 0.f;
        in[r][1] = cc.x; in[r][2] = cc.y; in[r][3] = cc.z; in[r][4] = cc.w;
        in[r][5] = (ok && x0 + 4 < W) ? row[x0 + 4] : 0.f;
    }
    float k[9];
    for (int i = 0; i < 9; ++i) k[i] = taps[(long)c * 9 + i];
    for (int r = 0; r < 4; ++r)
        for (int j = 0; j < 4; ++j) {
            float sacc = 0.f;
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) sacc = fmaf(in[r + dy][j + dx], k[dy * 3 + dx], sacc);
            o[r][j] = sacc;
        }
}

__global__ void __launch_bounds__(256) reader(const float* __restrict__ U, float* __restrict__ G, const float* __restrict__ taps) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // one block = one plane of 16 x 16 strips
    const int x0 = (int)(idx % 16) * 4, y0 = (int)((idx / 16) % 16) * 4;
    const long pl = idx / 256;
    const int b = (int)(pl / HID), c = (int)(pl % HID);
    const float* u1 = U + ((long)b * M + c) * N;
    float o1[4][4], o2[4][4];
    conv_half(o1, u1, y0, x0, taps, c);
    conv_half(o2, u1 + (long)HID * N, y0, x0, taps, c + HID);
    float* g = G + ((long)b * HID + c) * N;
    for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(g + (long)(y0 + r) * W + x0) =
            make_float4(o1[r][0] + 2.f * o2[r][0], o1[r][1] + 2.f * o2[r][1], o1[r][2] + 2.f * o2[r][2], o1[r][3] + 2.f * o2[r][3]);
}

__device__ __forceinline__ float conv_px(const float* plane, int y, int x, const float* taps, int c, float add) {
    float sacc = 0.f;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) sacc += (plane[(long)yy * W + xx] + add) * taps[(long)c * 9 + (dy + 1) * 3 + dx + 1];
        }
    return sacc;
}

__global__ void __launch_bounds__(256) check(const float* __restrict__ X, const float* __restrict__ G, const float* __restrict__ taps,
                                             float iter, unsigned* __restrict__ bad, unsigned* __restrict__ first_bad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // one thread per pixel of G
    const int x = (int)(idx % W), y = (int)((idx / W) % H);
    const long pl = idx / N;
    const int b = (int)(pl / HID), c = (int)(pl % HID);
    const float want = conv_px(X + ((long)b * C + (c % C)) * N, y, x, taps, c, iter) +
                       2.f * conv_px(X + ((long)b * C + ((c + HID) % C)) * N, y, x, taps, c + HID, iter);
    if (G[idx] != want) {
        atomicAdd(bad, 1u);
        atomicMin(first_bad, (unsigned)idx);
    }
}

// One kernel, two kinds of waves in a block: waves 0..3 issue v_mfma_f32_16x16x32_f16 back to back, waves 4..7 run the
// 16 x 9 multiply-add chains of the reader on small integers, forwards and backwards, and compare (exact either way).
template <int MF>
__global__ void __launch_bounds__(512) mixed(const float* __restrict__ X, const float* __restrict__ taps, int rounds,
                                             unsigned* __restrict__ bad, float* __restrict__ sink) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave < 4) {
        f4_t acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = f4_t{0.f, 0.f, 0.f, 0.f};
        h8_t a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(float)((lane + e) & 3); b[e] = (_Float16)(float)((lane * 3 + e) & 3); }
        for (int r = 0; r < rounds * 4; ++r)
            for (int i = 0; i < 12; ++i) acc[i] = mf<MF>(a, b, acc[i]);
        float s_ = 0.f;
        for (int i = 0; i < 12; ++i) s_ += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s_ == 12345.678f) sink[threadIdx.x] = s_;
        return;
    }
    const long idx = (long)blockIdx.x * 256 + (threadIdx.x - 256);
    const int x0 = (int)(idx % 16) * 4, y0 = (int)((idx / 16) % 16) * 4;
    const long pl = (idx / 256) % ((long)B * C);
    const float* plane = X + pl * N;
    const int c = (int)(pl % M);
    float in[6][6];
    for (int r = 0; r < 6; ++r) {
        const int yy = y0 - 1 + r;
        const bool ok = yy >= 0 && yy < H;
        const float* row = plane + (long)yy * W;
        for (int j = 0; j < 6; ++j) { const int xx = x0 - 1 + j; in[r][j] = (ok && xx >= 0 && xx < W) ? row[xx] : 0.f; }
    }
    float k[9];
    for (int i = 0; i < 9; ++i) k[i] = taps[(long)c * 9 + i];
    unsigned nbad = 0;
    for (int rep = 0; rep < rounds; ++rep) {
        for (int r = 0; r < 4; ++r)
            for (int j = 0; j < 4; ++j) {
                float f = (float)rep, g = (float)rep;
                for (int t = 0; t < 9; ++t) f = fmaf(in[r + t / 3][j + t % 3], k[t], f);
                for (int t = 8; t >= 0; --t) g = fmaf(in[r + t / 3][j + t % 3], k[t], g);
                nbad += f != g;
                asm volatile("" : "+v"(in[r][j]));        // keep the two chains from being folded across repetitions
            }
    }
    if (nbad) atomicAdd(bad, nbad);
}

typedef void (*writer_fn)(const float*, float*, float, int, int);
static writer_fn pick_writer(int mfk, int dma) {
#define PW(M_, D_) if (mfk == M_ && dma == D_) return writer<M_, D_>;
    PW(1, 16) PW(1, 4) PW(1, 0) PW(2, 16) PW(2, 4) PW(3, 16) PW(4, 16) PW(0, 16) PW(0, 4) PW(2, 0)
#undef PW
    printf("no such writer variant\n"); exit(2);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300, ns = argc > 2 ? atoi(argv[2]) : 2, spin = argc > 3 ? atoi(argv[3]) : 200;
    const int split = argc > 4 ? atoi(argv[4]) : 0;
    const int flags = argc > 5 ? atoi(argv[5]) : 0;
    const int mfk = argc > 6 ? atoi(argv[6]) : 1, dma = argc > 7 ? atoi(argv[7]) : 16;
    writer_fn wr = pick_writer(mfk, dma);
    const size_t lds = 8 * 20 * 128 * 4 + 32768;       // 112 KiB: one block per CU, like the library's kernel
    CK(hipFuncSetAttribute((const void*)wr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   // 1: no MFMAs, 2: no LDS-DMA inside the loop, 4: no barrier inside the loop   // 1: stream 0 only reads and checks a tensor written once, stream 1 only writes
    if (split < 0) {          // one kernel, mixed waves
        float* Xm; float* sink; unsigned* nb;
        std::vector<float> hx0((size_t)B * C * N);
        for (size_t i = 0; i < hx0.size(); ++i) hx0[i] = (float)((i * 2654435761u >> 20) & 7);
        std::vector<float> ht0((size_t)M * 9);
        for (size_t i = 0; i < ht0.size(); ++i) ht0[i] = (float)(1 + (i * 40503u >> 7) % 3);
        float* tp;
        CK(hipMalloc(&Xm, hx0.size() * 4)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&nb, 4)); CK(hipMalloc(&tp, ht0.size() * 4));
        CK(hipMemcpy(Xm, hx0.data(), hx0.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(tp, ht0.data(), ht0.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(nb, 0, 4));
        for (int it = 0; it < iters; ++it) {
            if (mfk == 1) hipLaunchKernelGGL(mixed<1>, dim3(2048), dim3(512), 0, 0, Xm, tp, spin, nb, sink);
            else if (mfk == 2) hipLaunchKernelGGL(mixed<2>, dim3(2048), dim3(512), 0, 0, Xm, tp, spin, nb, sink);
            else hipLaunchKernelGGL(mixed<0>, dim3(2048), dim3(512), 0, 0, Xm, tp, spin, nb, sink);
        }
        CK(hipDeviceSynchronize());
        unsigned h = 0; CK(hipMemcpy(&h, nb, 4, hipMemcpyDeviceToHost));
        printf("one kernel, mixed waves (mfma kind %d, %d launches x 2048 blocks, %d rounds): %u wrong chain results\n", mfk, iters, spin, h);
        return h ? 1 : 0;
    }
    std::vector<hipStream_t> st(ns);
    std::vector<float*> X(ns), U(ns), G(ns);
    std::vector<unsigned*> bad(ns);
    float* taps = nullptr;
    std::vector<float> ht((size_t)M * 9);
    for (size_t i = 0; i < ht.size(); ++i) ht[i] = (float)(1 + (i * 40503u >> 7) % 3);
    CK(hipMalloc(&taps, ht.size() * 4));
    CK(hipMemcpy(taps, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hx((size_t)B * C * N);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u >> 20) & 7);
    for (int s = 0; s < ns; ++s) {
        if (getenv("REPRO_CUMASK")) {
            // the two streams on disjoint sets of CUs: even / odd bit pairs of every mask word
            uint32_t mask[8];
            const uint32_t pat = atoi(getenv("REPRO_CUMASK")) == 1 ? (s == 0 ? 0x0000ffffu : 0xffff0000u) : (s == 0 ? 0x55555555u : 0xaaaaaaaau);
            for (int w = 0; w < 8; ++w) mask[w] = pat;
            CK(hipExtStreamCreateWithCUMask(&st[s], 8, mask));
        } else
        CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
        CK(hipMalloc(&X[s], hx.size() * 4)); CK(hipMalloc(&U[s], (size_t)B * M * N * 4)); CK(hipMalloc(&G[s], (size_t)B * HID * N * 4));
        CK(hipMalloc(&bad[s], 8 * iters + 8));
        CK(hipMemcpy(X[s], hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(bad[s], 0, 8 * iters + 8));
    }
    CK(hipDeviceSynchronize());
    const long un = (long)B * M * N;
    if (split && ns == 2) {
        hipLaunchKernelGGL(wr, dim3(NTILES * MBLOCKS, B), dim3(512), lds, st[0], X[0], U[0], 3.f, 1, 0);
        CK(hipDeviceSynchronize());
        for (int it = 0; it < iters; ++it) {
            CK(hipMemsetAsync(bad[0] + 2 * it + 1, 0xff, 4, st[0]));
            hipLaunchKernelGGL(reader, dim3(B * HID), dim3(256), 0, st[0], U[0], G[0], taps);
            hipLaunchKernelGGL(check, dim3(B * HID * N / 256), dim3(256), 0, st[0], X[0], G[0], taps, 3.f, bad[0] + 2 * it, bad[0] + 2 * it + 1);
            if (it % split == 0) hipLaunchKernelGGL(wr, dim3(NTILES * MBLOCKS, B), dim3(512), lds, st[1], X[1], U[1], (float)(it & 15), spin, flags);
        }
    } else
    for (int it = 0; it < iters; ++it) {
        for (int s = 0; s < ns; ++s) {
            CK(hipMemsetAsync(bad[s] + 2 * it + 1, 0xff, 4, st[s]));
            hipLaunchKernelGGL(scribble, dim3((unsigned)((un / 4 + 255) / 256)), dim3(256), 0, st[s], U[s], un, -1000.f - it);
            hipLaunchKernelGGL(wr, dim3(NTILES * MBLOCKS, B), dim3(512), lds, st[s], X[s], U[s], (float)(it & 15), spin, flags);
            hipLaunchKernelGGL(reader, dim3(B * HID), dim3(256), 0, st[s], U[s], G[s], taps);
            hipLaunchKernelGGL(check, dim3(B * HID * N / 256), dim3(256), 0, st[s], X[s], G[s], taps, (float)(it & 15), bad[s] + 2 * it, bad[s] + 2 * it + 1);
        }
    }
    CK(hipDeviceSynchronize());
    int rc = 0;
    for (int s = 0; s < ns; ++s) {
        std::vector<unsigned> hb(2 * iters);
        CK(hipMemcpy(hb.data(), bad[s], 8 * iters, hipMemcpyDeviceToHost));
        int wrong_iters = 0; unsigned long total = 0;
        for (int it = 0; it < iters; ++it) if (hb[2 * it]) {
            if (wrong_iters < 5) printf("  stream %d iter %d: %u wrong pixels, first at %u (sample %u, lane of its strip %u)\n", s, it, hb[2 * it], hb[2 * it + 1], hb[2 * it + 1] / (HID * N), ((hb[2 * it + 1] % N) / W / 4 * 16 + (hb[2 * it + 1] % W) / 4) % 64);
            ++wrong_iters; total += hb[2 * it];
        }
        printf("stream %d of %d: %d of %d iterations wrong (%lu pixels)\n", s, ns, wrong_iters, iters, total);
        if (wrong_iters) rc = 1;
    }
    return rc;
}
