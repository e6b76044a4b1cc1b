// Channel LayerNorm over NCHW, forward and backward (reference:
// model/Restromer.py:20-74).  HBM-bound: 2*C*4 bytes per pixel each way.
//
// Layout of a block: a tile of 64*VEC consecutive pixels of one sample; lane l
// of every wave owns pixels [l*VEC, l*VEC+VEC) of the tile, and the block's
// waves split the channels round-robin (wave w owns channels w, w+WAVES, ...).
// Adjacent lanes read adjacent pixels, so every load/store instruction of a
// wave is one fully coalesced 64*VEC*4-byte segment of one channel plane.  The
// whole channel column of a pixel stays in registers between the statistics
// and the normalisation, so x is read from HBM exactly once.
#include "kernels.hpp"

namespace rst {

template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<4> { typedef f32x4 type; };

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int VEC>
__device__ __forceinline__ void load_vec(float (&dst)[VEC], const float* p) {
    if constexpr (VEC == 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p);
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    } else if constexpr (VEC == 2) {
        f32x2 v = *reinterpret_cast<const f32x2*>(p);
        dst[0] = v.x; dst[1] = v.y;
    } else {
        dst[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&src)[VEC]) {
    if constexpr (VEC == 4) {
        f32x4 v = {src[0], src[1], src[2], src[3]};
        *reinterpret_cast<f32x4*>(p) = v;
    } else if constexpr (VEC == 2) {
        f32x2 v = {src[0], src[1]};
        *reinterpret_cast<f32x2*>(p) = v;
    } else {
        *p = src[0];
    }
}

// Sum `v[0..VEC)` over the waves of the block (each wave holds a partial for
// the same pixels).  red: [WAVES][64*VEC] floats of LDS.
template <int VEC>
__device__ __forceinline__ void block_channel_sum(float (&v)[VEC], float* red, int waves, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[wave * 64 * VEC + lane * VEC + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        float s = 0.f;
        for (int w = 0; w < waves; ++w) s += red[w * 64 * VEC + lane * VEC + i];
        v[i] = s;
    }
}

template <int VEC, int CPT>
__global__ void __launch_bounds__(512)
ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ weight, const float* __restrict__ bias,
              float* __restrict__ y, int C, int64_t N, int64_t tiles_per_sample) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int waves = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = blockIdx.x / tiles_per_sample;
    const int64_t p0 = (blockIdx.x % tiles_per_sample) * (64 * VEC) + lane * VEC;
    const bool pix_ok = p0 < N;     // N % VEC == 0 is guaranteed by the launcher
    const float* xb = x + b * (int64_t)C * N;
    float* yb = y + b * (int64_t)C * N;

    float v[CPT][VEC];
    float s[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = wave + j * waves;
        if (c < C && pix_ok) {
            load_vec<VEC>(v[j], xb + (int64_t)c * N + p0);
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[j][i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] += v[j][i];
    }
    float* red0 = smem;
    float* red1 = smem + waves * 64 * VEC;
    block_channel_sum<VEC>(s, red0, waves, wave, lane);
    float mu[VEC], q[VEC];
    const float invC = 1.0f / (float)C;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { mu[i] = s[i] * invC; q[i] = 0.f; }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = wave + j * waves;
        if (c < C) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) { float d = v[j][i] - mu[i]; q[i] += d * d; }
        }
    }
    block_channel_sum<VEC>(q, red1, waves, wave, lane);
    float rstd[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) rstd[i] = 1.0f / sqrtf(q[i] * invC + 1e-5f);

#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = wave + j * waves;
        if (c < C && pix_ok) {
            const float w = weight[c];
            float o[VEC];
            if (bias != nullptr) {
                const float bb = bias[c];
#pragma unroll
                for (int i = 0; i < VEC; ++i) o[i] = (v[j][i] - mu[i]) * rstd[i] * w + bb;
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) o[i] = v[j][i] * rstd[i] * w;
            }
            store_vec<VEC>(yb + (int64_t)c * N + p0, o);
        }
    }
}

// Backward.  With g = dy*w, xh = (x-mu)*rstd:
//   WithBias: dx = rstd * (g - mean_c(g) - xh * mean_c(g*xh));  dw = sum dy*xh;  db = sum dy
//   BiasFree: y = x*rstd*w  =>  dx = rstd*g - rstd^3 * (x-mu) * mean_c(g*x);  dw = sum dy*x*rstd
// Blocks stride over tiles and keep per-channel partial sums of dw/db in
// registers; one partial row per block goes to `partial` ([grid][2][C]) and a
// second kernel adds the rows in a fixed order (deterministic).
// Buffer-addressed vectors for the backward kernel: one descriptor per tensor and sample, the per-lane part of
// the address is a single 32-bit offset shared by all channels, the channel plane comes in as a scalar offset
// (flat 64-bit addresses for 24 channels x 3 tensors do not fit the register file).  A lane whose offset lies
// outside the descriptor reads zeros and its stores are dropped.
typedef unsigned ln_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned ln_u32x4 __attribute__((ext_vector_type(4)));
template <int VEC>
__device__ __forceinline__ void buf_load(float (&v)[VEC], __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (VEC == 4) {
        const ln_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(t[i]);
    } else if constexpr (VEC == 2) {
        const ln_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        v[0] = __uint_as_float(t[0]); v[1] = __uint_as_float(t[1]);
    } else {
        v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    }
}
// (the plane offset goes into the vector offset: a wide store with a scalar-register offset may still be reading
// its data registers when the next vector instruction overwrites them)
template <int VEC>
__device__ __forceinline__ void buf_store(const float (&v)[VEC], __amdgpu_buffer_rsrc_t r, unsigned voff) {
    if constexpr (VEC == 4) {
        ln_u32x4 t;
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = __float_as_uint(v[i]);
        __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, 0, 0);
    } else if constexpr (VEC == 2) {
        ln_u32x2 t;
        t[0] = __float_as_uint(v[0]); t[1] = __float_as_uint(v[1]);
        __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, 0, 0);
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0]), r, voff, 0, 0);
    }
}

template <int VEC, int CPT>
__global__ void __launch_bounds__(1024)
ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ weight, int with_bias,
              const float* __restrict__ dy, const float* addend, float* dx,
              float* __restrict__ partial, int C, int64_t N, int64_t tiles_per_sample, int64_t total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int waves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float* red0 = smem;
    float* red1 = smem + waves * 64 * VEC;
    const float invC = 1.0f / (float)C;
    const unsigned plane = (unsigned)(N * 4);           // bytes per channel plane (C * N * 4 < 2^32, checked by the launcher)
    const unsigned sample_bytes = (unsigned)((int64_t)C * N * 4);
    constexpr int GRP = VEC == 4 ? 2 : 6;            // dy vectors in flight per lane and pass
    static_assert(CPT % GRP == 0, "channel share must be whole groups");

    // Registers hold the centred, normalised x of the tile and the parameter-gradient sums; dy is
    // streamed twice (sums over the channels first, dx second -- the second read comes from the
    // cache).  Holding dy as well does not fit 128 registers at 24 channels per wave and spills.
    float pw[CPT], pb[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) { pw[j] = 0.f; pb[j] = 0.f; }

    for (int64_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int64_t b = tile / tiles_per_sample;
        const int64_t p0 = (tile % tiles_per_sample) * (64 * VEC) + lane * VEC;
        const bool pix_ok = p0 < N;
        const unsigned voff = pix_ok ? (unsigned)(p0 * 4) : 0xFFFFFFF0u;
        const int64_t sbase = b * (int64_t)C * N;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + sbase), 0, sample_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy + sbase), 0, sample_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t radd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((addend ? addend : dx) + sbase), 0, sample_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(dx + sbase, 0, sample_bytes, 0x00020000);

        float v[CPT][VEC];
        float s[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] = 0.f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = wave + j * waves;
            if (c < C) {
                buf_load<VEC>(v[j], rx, voff, (unsigned)c * plane);
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) v[j][i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) s[i] += v[j][i];
        }
        block_channel_sum<VEC>(s, red0, waves, wave, lane);
        float mu[VEC], q[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { mu[i] = s[i] * invC; q[i] = 0.f; }
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = wave + j * waves;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float d = c < C ? v[j][i] - mu[i] : 0.f;
                v[j][i] = d;
                q[i] += d * d;
            }
        }
        block_channel_sum<VEC>(q, red1, waves, wave, lane);
        float rstd[VEC], shift[VEC];                  // shift: what the bias-free form adds to the centred x-hat
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            rstd[i] = 1.0f / sqrtf(q[i] * invC + 1e-5f);
            shift[i] = with_bias ? 0.f : mu[i] * rstd[i];
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j)
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[j][i] *= rstd[i];                 // centred x-hat

        // pass 1 over dy: parameter-gradient sums and, per pixel, s1 = sum_c dy w, s2 = sum_c dy w xhat_c
        float s1[VEC], s2[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
#pragma unroll
        for (int j0 = 0; j0 < CPT; j0 += GRP) {
            float g[GRP][VEC];
#pragma unroll
            for (int jj = 0; jj < GRP; ++jj) {
                const int c = wave + (j0 + jj) * waves;
                if (c < C) {
                    buf_load<VEC>(g[jj], rdy, voff, (unsigned)c * plane);
                } else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) g[jj][i] = 0.f;
                }
            }
#pragma unroll
            for (int jj = 0; jj < GRP; ++jj) {
                const int j = j0 + jj;
                const int c = wave + j * waves;
                const float w = c < C ? weight[c] : 0.f;          // wave-uniform: a scalar load
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float dyv = g[jj][i];
                    pw[j] = fmaf(dyv, v[j][i] + shift[i], pw[j]);
                    pb[j] += dyv;
                    const float gw = dyv * w;
                    s1[i] += gw;
                    s2[i] = fmaf(gw, v[j][i], s2[i]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);        // the next group's loads stay behind this group's use
        }
        __syncthreads();   // red0/red1 are reused below
        block_channel_sum<VEC>(s1, red0, waves, wave, lane);
        block_channel_sum<VEC>(s2, red1, waves, wave, lane);
        // with bias:  dx = rstd (dy w - s1/C - xhat_c s2/C)
        // bias-free:  y = x rstd w, so dx = rstd dy w - rstd^3 (x - mu) (sum_c dy w x)/C with
        //             sum_c dy w x = s2/rstd + mu s1  and  rstd^3 (x - mu) = rstd^2 xhat_c
        float k1[VEC], k2[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            if (with_bias) { k1[i] = s1[i] * invC; k2[i] = s2[i] * invC; }
            else { k1[i] = 0.f; k2[i] = rstd[i] * (s2[i] / rstd[i] + mu[i] * s1[i]) * invC; }
        }
        // pass 2 over dy
#pragma unroll
        for (int j0 = 0; j0 < CPT; j0 += GRP) {
            float g[GRP][VEC], old[GRP][VEC];
#pragma unroll
            for (int jj = 0; jj < GRP; ++jj) {
                const int c = wave + (j0 + jj) * waves;
                if (c < C) {
                    buf_load<VEC>(g[jj], rdy, voff, (unsigned)c * plane);
                    if (addend != nullptr) buf_load<VEC>(old[jj], radd, voff, (unsigned)c * plane);
                }
            }
#pragma unroll
            for (int jj = 0; jj < GRP; ++jj) {
                const int j = j0 + jj;
                const int c = wave + j * waves;
                if (c < C) {
                    const float w = weight[c];
                    float o[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        o[i] = rstd[i] * (g[jj][i] * w - k1[i] - v[j][i] * k2[i]);
                        if (addend != nullptr) o[i] += old[jj][i];
                    }
                    buf_store<VEC>(o, rdx, pix_ok ? voff + (unsigned)c * plane : 0xFFFFFFF0u);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    // one partial row per block
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = wave + j * waves;
        const float a = wave_sum(pw[j]);
        const float bsum = wave_sum(pb[j]);
        if (lane == 0 && c < C) {
            partial[((int64_t)blockIdx.x * 2 + 0) * C + c] = a;
            partial[((int64_t)blockIdx.x * 2 + 1) * C + c] = bsum;
        }
    }
}

// out[r][c] = sum_i partial[i][r][c] (i over the partial rows), in a fixed order:
// block = 64 channels x 16 row groups, each thread adds every 16th row, then the
// 16 groups are added in order through LDS.
__global__ void __launch_bounds__(1024)
ln_reduce_partials_kernel(const float* __restrict__ partial, int rows, int C,
                          float* __restrict__ dweight, float* __restrict__ dbias) {
    __shared__ float red[2][16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float a = 0.f, b = 0.f;
    if (c < C) {
        for (int i = rg; i < rows; i += 16) {
            a += partial[((int64_t)i * 2 + 0) * C + c];
            b += partial[((int64_t)i * 2 + 1) * C + c];
        }
    }
    red[0][rg][cl] = a;
    red[1][rg][cl] = b;
    __syncthreads();
    if (rg == 0 && c < C) {
        float sa = 0.f, sb = 0.f;
        for (int k = 0; k < 16; ++k) { sa += red[0][k][cl]; sb += red[1][k][cl]; }
        dweight[c] = sa;
        if (dbias != nullptr) dbias[c] = sb;
    }
}

// ---- launch plan --------------------------------------------------------------
struct LnPlan {
    int vec, cpt, waves;
};
static bool ln_plan(int C, int64_t N, bool ptr_aligned, bool backward, LnPlan* p) {
    if (!backward) {
        // forward: at most 8 waves (512 threads -> 256 registers per lane); the channel column of a
        // pixel stays in registers
        int waves = (int)cdiv(C, 12);
        if (waves > 8) waves = 8;
        const int cpt = (int)cdiv(C, waves);
        if (cpt <= 12) p->cpt = 12;
        else if (cpt <= 24) p->cpt = 24;
        else if (cpt <= 48) p->cpt = 48;
        else return false;
        p->vec = (N % 4 == 0 && ptr_aligned && p->cpt <= 24) ? 4 : 1;
        p->waves = waves;
        return true;
    }
    // backward holds x and dy: up to 16 waves (1024 threads -> 128 registers per lane), so two
    // blocks fit on a CU; the pixel vector shrinks as the channel share of a wave grows
    // 49..96 channels: 12 channels per wave -> blocks of at most 8 waves, two or more of them per CU, out of
    // phase (a block is a chain of load -> reduce -> load steps; a second one fills its waits)
    int waves = (int)cdiv(C, (C > 48 && C <= 96) ? 12 : 6);
    if (waves > 16) waves = 16;
    const int cpt = (int)cdiv(C, waves);
    int want_vec;
    if (cpt <= 6) { p->cpt = 6; want_vec = 4; }
    else if (cpt <= 12) { p->cpt = 12; want_vec = 2; }
    else if (cpt <= 24) { p->cpt = 24; want_vec = 1; }
    else return false;
    p->vec = (ptr_aligned && N % want_vec == 0) ? want_vec : 1;
    p->waves = waves;
    return true;
}

static const int LN_BWD_MAX_BLOCKS = 1024;

int layernorm_fwd_launch(const float* x, const float* weight, const float* bias, float* y,
                         int B, int C, int64_t N, hipStream_t stream) {
    LnPlan p;
    RST_REQUIRE(ln_plan(C, N, aligned16(x) && aligned16(y), false, &p), "layernorm: C=%d too large (max 384)", C);
    const int64_t tps = cdiv(N, 64 * p.vec);
    const int64_t grid = (int64_t)B * tps;
    RST_REQUIRE(grid > 0 && grid < (1ll << 31), "layernorm: grid out of range");
    const size_t lds = (size_t)2 * p.waves * 64 * p.vec * sizeof(float);
    dim3 blk(p.waves * 64);
    ProfScope prof(stream, "layernorm_fwd", 8.0 * B * C * N, 2.0 * 4 * B * C * N);
#define LN_FWD(V, CP) hipLaunchKernelGGL((ln_fwd_kernel<V, CP>), dim3((unsigned)grid), blk, lds, stream, x, weight, bias, y, C, N, tps)
    if (p.vec == 4 && p.cpt == 12) LN_FWD(4, 12);
    else if (p.vec == 4 && p.cpt == 24) LN_FWD(4, 24);
    else if (p.vec == 1 && p.cpt == 12) LN_FWD(1, 12);
    else if (p.vec == 1 && p.cpt == 24) LN_FWD(1, 24);
    else LN_FWD(1, 48);
#undef LN_FWD
    RST_LAUNCH_CHECK();
    return RST_OK;
}

size_t layernorm_bwd_ws_bytes(int B, int C, int64_t N) {
    (void)B; (void)N;
    return (size_t)LN_BWD_MAX_BLOCKS * 2 * (size_t)C * sizeof(float);
}

int layernorm_bwd_launch(const float* x, const float* weight, int with_bias, const float* dy,
                         const float* addend, float* dx, float* dweight, float* dbias,
                         int B, int C, int64_t N, void* ws, size_t ws_bytes, hipStream_t stream) {
    LnPlan p;
    RST_REQUIRE(ln_plan(C, N, aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(addend), true, &p),
                "layernorm: C=%d too large (max 384)", C);
    if (ws == nullptr || ws_bytes < layernorm_bwd_ws_bytes(B, C, N))
        return fail(RST_ERR_WORKSPACE, "layernorm_bwd: workspace too small (%zu < %zu)", ws_bytes,
                    layernorm_bwd_ws_bytes(B, C, N));
    RST_REQUIRE((int64_t)C * N * 4 < (1ll << 32) - 64, "layernorm_bwd: a sample of %d x %lld floats is beyond the 32-bit buffer offsets", C, (long long)N);
    const int64_t tps = cdiv(N, 64 * p.vec);
    const int64_t total = (int64_t)B * tps;
    const int grid = (int)(total < LN_BWD_MAX_BLOCKS ? total : LN_BWD_MAX_BLOCKS);
    const size_t lds = (size_t)2 * p.waves * 64 * p.vec * sizeof(float);
    dim3 blk(p.waves * 64);
    float* partial = static_cast<float*>(ws);
    ProfScope prof(stream, "layernorm_bwd", 20.0 * B * C * N, (addend ? 4.0 : 3.0) * 4 * B * C * N);
#define LN_BWD(V, CP) hipLaunchKernelGGL((ln_bwd_kernel<V, CP>), dim3(grid), blk, lds, stream, x, weight, with_bias, dy, addend, dx, partial, C, N, tps, total)
    if (p.vec == 4 && p.cpt == 6) LN_BWD(4, 6);
    else if (p.vec == 2 && p.cpt == 12) LN_BWD(2, 12);
    else if (p.vec == 1 && p.cpt == 6) LN_BWD(1, 6);
    else if (p.vec == 1 && p.cpt == 12) LN_BWD(1, 12);
    else LN_BWD(1, 24);
#undef LN_BWD
    RST_LAUNCH_CHECK();
    hipLaunchKernelGGL(ln_reduce_partials_kernel, dim3((unsigned)cdiv(C, 64)), dim3(1024), 0, stream,
                       partial, grid, C, dweight, with_bias ? dbias : nullptr);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// ---- layernorm backward for callers that cannot allocate ---------------------------------------
// The per-block (dweight, dbias) partial sums -- the only scratch of the backward pass -- live in a
// file-scope device array instead of a caller workspace.  Every GPU the library runs on holds its own copy
// (3 MiB) at its own address until the process ends; the address is asked for once per device, with that
// device current, and from then on only that pointer is used in host code (the array's name is a
// host-side handle, not a device address).
static __device__ __align__(256) float scratch[LN_BWD_MAX_BLOCKS * 2 * 384];

static int scratch_for_current_device(float** out) {
    static std::mutex mu;
    static std::map<int, float*> per_device;
    int dev = 0;
    RST_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = per_device.find(dev);
    if (it == per_device.end()) {
        float* ptr = nullptr;
        RST_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)));
        RST_REQUIRE(ptr != nullptr, "layernorm_bwd_noalloc: no device address for the scratch array on device %d", dev);
        it = per_device.emplace(dev, ptr).first;
    }
    *out = it->second;
    return RST_OK;
}

int layernorm_bwd_noalloc_launch(const float* x, const float* weight, int with_bias, const float* dy,
                                 const float* addend, float* dx, float* dweight, float* dbias,
                                 int B, int C, int64_t N, hipStream_t stream) {
    RST_REQUIRE(C <= 384, "layernorm_bwd_noalloc: C=%d beyond the built-in scratch (at most 384 channels)", C);
    const size_t need = layernorm_bwd_ws_bytes(B, C, N);
    RST_REQUIRE(need <= sizeof(float) * (size_t)LN_BWD_MAX_BLOCKS * 2 * 384,
                "layernorm_bwd_noalloc: %zu bytes of scratch needed, %zu built in", need,
                sizeof(float) * (size_t)LN_BWD_MAX_BLOCKS * 2 * 384);
    float* ptr = nullptr;
    RST_TRY(scratch_for_current_device(&ptr));
    // every partial row that the reduction reads is written by this call's first kernel: nothing to clear
    return layernorm_bwd_launch(x, weight, with_bias, dy, addend, dx, dweight, dbias, B, C, N, ptr, need, stream);
}

}  // namespace rst
