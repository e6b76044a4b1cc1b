// G[g] = sum over pixels (and optionally over samples) of A[:, n] * Bm[:, n]^T
//
// The "reduce over all H*W pixels" products of the block:
//   * MDTA Gram matrix q k^T per (sample, head) together with the squared L2
//     norms of every q and k row (model/Restromer.py:137-140), K = H*W
//   * weight gradients of the 1x1 convolutions, dW = sum_b dY[b] X[b]^T
//   * dY v^T in the attention backward
// M1, M2 are small (<= 1152) and the contraction runs over up to millions of
// pixels, so the pixel axis is split over many workgroups (split-K): each
// block reduces its pixel range on the matrix cores (exact fp32
// v_mfma_f32_16x16x4_f32) and writes one partial slab; a second pass adds the
// slabs in a fixed order, so results are bitwise reproducible.
//
// A and Bm are read straight from their NCHW planes by LDS-DMA (no staging registers).
// In LDS a row keeps its pixels in 16-byte chunks; MFMA lane (i, kk) takes pixels
// 4*kk..4*kk+3 of row i with one ds_read_b128 and feeds them to four consecutive k-steps
// (the k order inside a 16-pixel group is permuted identically for both operands, which
// leaves the sum unchanged).  The squared row norms are accumulated from the same operand
// registers.
#include "kernels.hpp"

namespace rst {


typedef __attribute__((address_space(1))) const void* nt_gptr;
typedef __attribute__((address_space(3))) void* nt_lptr;

constexpr int NT_STAGES = 3;

// T1,T2: 16-row tiles per wave along M1/M2.  WR x WC x WK = 4 waves: WR*WC waves tile
// the output, WK waves split the pixels of a stage.  A stage holds SP = 32 pixels (64 when
// WK = 4) of every A row and B row of the block tile ([rows][SP], A rows first), brought in
// by LDS-DMA in 1-KiB pieces (256/SP rows each), PPW pieces per wave per stage, three
// stages deep.  The 16-byte chunk g of a row is stored at chunk position
// g ^ (row & (SP/4-1)): the permutation is applied to the per-lane SOURCE address (the LDS
// image of a piece stays lane-linear), so the 16 rows an operand read touches spread over
// the banks.
template <int T1, int T2>
__global__ void __launch_bounds__(256, 2)
gemm_nt_kernel(GemmNT g, int WR, int WC, int rblocks, int cblocks, int SP, int PPW) {
    extern __shared__ __attribute__((aligned(16))) float ring[];
    const int WK = 4 / (WR * WC);
    const int R1 = 16 * T1 * WR, R2 = 16 * T2 * WC;      // block tile
    const int cmask = SP / 4 - 1;
    const int rows_per_piece = 256 / SP;                  // 1 KiB / (SP * 4 B)
    const int lane_shift = SP == 64 ? 4 : 3;              // lanes per row of a piece = SP/4
    const int stage_floats = PPW * 4 * 256;               // PPW pieces x 4 waves x 256 floats

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // the row / column blocks of one pixel split read the same pixels: neighbours on one XCD
    int bid = g.walk ? (int)xcd_walk(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int cb = bid % cblocks; bid /= cblocks;
    const int rb = bid % rblocks; bid /= rblocks;
    const int sp = bid % g.splits; bid /= g.splits;
    const int batch = bid;
    const int b1 = batch / g.nb2, b2 = batch % g.nb2;

    const float* Ab = g.A + (int64_t)b1 * g.a_s1 + (int64_t)b2 * g.a_s2;
    const float* Bb = g.Bm + (int64_t)b1 * g.b_s1 + (int64_t)b2 * g.b_s2;
    const int64_t p_begin = (int64_t)sp * g.chunk;
    const int64_t p_end = min(g.N, p_begin + g.chunk);    // multiple of 4
    const int nstages = (int)((p_end - p_begin + SP - 1) / SP);

    // this lane's share of the PPW pieces of a stage (fixed for the whole kernel):
    // piece (wave + 4 j) covers image rows piece*rows_per_piece .. ; lane -> one 16-byte chunk of one row
    const float* rowbase[8];
    int gchunk[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (wave + 4 * j) * rows_per_piece + (lane >> lane_shift);
        if (row < R1) {
            rowbase[j] = Ab + (int64_t)min(rb * R1 + row, g.M1 - 1) * g.lda;
        } else {
            rowbase[j] = Bb + (int64_t)min(cb * R2 + min(row - R1, R2 - 1), g.M2 - 1) * g.ldb;
        }
        gchunk[j] = 4 * ((lane & cmask) ^ (row & cmask));
    }
    auto issue = [&](int s) {
        float* st = ring + (s % NT_STAGES) * stage_floats;
        const int64_t p = p_begin + (int64_t)s * SP;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < PPW) {
                int64_t px = p + gchunk[j];
                if (px > p_end - 4) px = p_end - 4;      // clamped duplicate; masked in the MFMA loop
                __builtin_amdgcn_global_load_lds((nt_gptr)(rowbase[j] + px), (nt_lptr)(st + (wave + 4 * j) * 256), 16, 0, 0);
            }
        }
    };

    f32x4 acc[T1][T2];
    float sqa[T1], sqb[T2];
#pragma unroll
    for (int i = 0; i < T1; ++i) {
        sqa[i] = 0.f;
#pragma unroll
        for (int j = 0; j < T2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < T2; ++j) sqb[j] = 0.f;

    const int wk = wave % WK;
    const int wt = wave / WK;
    const int wr = wt / WC, wc = wt % WC;
    const int li = lane & 15, kk = lane >> 4;
    const int px_per_wave = SP / WK;                 // 32, 16 or 16

    // prologue: two stages in flight (re-issuing the last stage keeps the per-wave count fixed)
    issue(0);
    issue(nstages > 1 ? 1 : 0);
    for (int s = 0; s < nstages; ++s) {
        // stage s landed for this wave (stage s+1 = PPW pieces may stay in flight), then for all;
        // the barrier also frees the slot of stage s-1 for the refill
        switch (PPW) {
            case 1: asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory"); break;
        }
        issue(s + 2 < nstages ? s + 2 : nstages - 1);
        const float* st = ring + (s % NT_STAGES) * stage_floats;
        const int64_t p = p_begin + (int64_t)s * SP;
        for (int q = 0; q < px_per_wave; q += 16) {
            const int o = wk * px_per_wave + q + 4 * kk;       // this lane's 4 pixels inside the stage
            const bool valid = p + o < p_end;
            const int gch = o >> 2;
            f32x4 av[T1], bv[T2];
#pragma unroll
            for (int i = 0; i < T1; ++i) {
                const int row = wr * 16 * T1 + 16 * i + li;
                av[i] = *reinterpret_cast<const f32x4*>(&st[row * SP + 4 * (gch ^ (row & cmask))]);
                if (!valid) av[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                sqa[i] += av[i].x * av[i].x + av[i].y * av[i].y + av[i].z * av[i].z + av[i].w * av[i].w;
            }
#pragma unroll
            for (int j = 0; j < T2; ++j) {
                const int row = R1 + wc * 16 * T2 + 16 * j + li;
                bv[j] = *reinterpret_cast<const f32x4*>(&st[row * SP + 4 * (gch ^ (row & cmask))]);
                if (!valid) bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                sqb[j] += bv[j].x * bv[j].x + bv[j].y * bv[j].y + bv[j].z * bv[j].z + bv[j].w * bv[j].w;
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < T1; ++i)
#pragma unroll
                    for (int j = 0; j < T2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][s4], bv[j][s4], acc[i][j], 0, 0, 0);
        }
    }

    // every DMA has landed and every wave has left the loop: the ring becomes reduction scratch
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    float* red = ring;        // [WK][WR*WC][T1*T2][64 lanes][4]
    if (WK > 1) {
#pragma unroll
        for (int i = 0; i < T1; ++i)
#pragma unroll
            for (int j = 0; j < T2; ++j)
                *reinterpret_cast<f32x4*>(&red[(((wk * (WR * WC) + wt) * (T1 * T2) + i * T2 + j) * 64 + lane) * 4]) = acc[i][j];
        __syncthreads();
    }
    float* slab = g.partial + ((int64_t)batch * g.splits + sp) * (int64_t)g.M1 * g.M2;
    if (wk == 0) {
#pragma unroll
        for (int i = 0; i < T1; ++i)
#pragma unroll
            for (int j = 0; j < T2; ++j) {
                f32x4 v = acc[i][j];
                for (int w = 1; w < WK; ++w) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(&red[(((w * (WR * WC) + wt) * (T1 * T2) + i * T2 + j) * 64 + lane) * 4]);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                const int col = cb * R2 + wc * 16 * T2 + 16 * j + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rb * R1 + wr * 16 * T1 + 16 * i + 4 * kk + r;
                    if (row < g.M1 && col < g.M2) slab[(int64_t)row * g.M2 + col] = v[r];
                }
            }
    }
    // ---- squared row norms: lane (li, kk) holds the part of row li over its pixels ----
    if (g.sqA_partial != nullptr || g.sqB_partial != nullptr) {
        __syncthreads();                              // the accumulator exchange above is done with `red`
        float* sq = ring;                             // [4 waves][T1 + T2][16]
#pragma unroll
        for (int i = 0; i < T1; ++i) {
            float v = sqa[i];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (kk == 0) sq[(wave * (T1 + T2) + i) * 16 + li] = v;
        }
#pragma unroll
        for (int j = 0; j < T2; ++j) {
            float v = sqb[j];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (kk == 0) sq[(wave * (T1 + T2) + T1 + j) * 16 + li] = v;
        }
        __syncthreads();
        // rows of A: tile (wr, i); summed over the WK pixel-split waves with wc == 0
        if (g.sqA_partial != nullptr && cb == 0) {
            for (int e = tid; e < R1; e += 256) {
                const int twr = e / (16 * T1), ti = (e / 16) % T1, l = e % 16;
                float v = 0.f;
                for (int w = 0; w < WK; ++w) v += sq[(((twr * WC + 0) * WK + w) * (T1 + T2) + ti) * 16 + l];
                const int row = rb * R1 + e;
                if (row < g.M1) g.sqA_partial[((int64_t)batch * g.splits + sp) * g.M1 + row] = v;
            }
        }
        if (g.sqB_partial != nullptr && rb == 0) {
            for (int e = tid; e < R2; e += 256) {
                const int twc = e / (16 * T2), tj = (e / 16) % T2, l = e % 16;
                float v = 0.f;
                for (int w = 0; w < WK; ++w) v += sq[(((0 * WC + twc) * WK + w) * (T1 + T2) + T1 + tj) * 16 + l];
                const int row = cb * R2 + e;
                if (row < g.M2) g.sqB_partial[((int64_t)batch * g.splits + sp) * g.M2 + row] = v;
            }
        }
    }
}

// ---- fallback for operands that are not 16-byte aligned / pixel counts not multiples of 4:
// register-staged loads with per-element bounds checks, same tiling and numerics ----
constexpr int NT_KP = 64;          // pixels staged per iteration
constexpr int NT_LD = NT_KP + 4;   // LDS row stride (floats): 17 x 16 B

// T1,T2: 16-row tiles per wave along M1/M2.  WR x WC x WK = 4 waves: WR*WC
// waves tile the output, WK waves split the staged pixels.
template <int T1, int T2, bool VEC>
__global__ void __launch_bounds__(256)
gemm_nt_generic_kernel(GemmNT g, int WR, int WC, int rblocks, int cblocks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int WK = 4 / (WR * WC);
    const int R1 = 16 * T1 * WR, R2 = 16 * T2 * WC;     // block tile
    float* As = smem;
    float* Bs = smem + R1 * NT_LD;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // the row / column blocks of one pixel split read the same pixels: neighbours on one XCD
    int bid = g.walk ? (int)xcd_walk(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int cb = bid % cblocks; bid /= cblocks;
    const int rb = bid % rblocks; bid /= rblocks;
    const int sp = bid % g.splits; bid /= g.splits;
    const int batch = bid;
    const int b1 = batch / g.nb2, b2 = batch % g.nb2;

    const float* Ab = g.A + (int64_t)b1 * g.a_s1 + (int64_t)b2 * g.a_s2 + (int64_t)(rb * R1) * g.lda;
    const float* Bb = g.Bm + (int64_t)b1 * g.b_s1 + (int64_t)b2 * g.b_s2 + (int64_t)(cb * R2) * g.ldb;
    const int rowsA = min(R1, g.M1 - rb * R1);     // valid rows in this tile
    const int rowsB = min(R2, g.M2 - cb * R2);
    const int64_t p_begin = (int64_t)sp * g.chunk;
    const int64_t p_end = min(g.N, p_begin + g.chunk);

    // staging: thread -> 4 pixels at (tid&15)*4 of rows (tid>>4) + 16*i
    const int sc = (tid & 15) * 4, sr = tid >> 4;
    constexpr int MAXR = 8;                       // up to 128 rows per operand
    f32x4 ra[MAXR], rbv[MAXR];
    float sqa[MAXR], sqb[MAXR];
#pragma unroll
    for (int i = 0; i < MAXR; ++i) { sqa[i] = 0.f; sqb[i] = 0.f; }
    const int itA = R1 / 16, itB = R2 / 16;

    auto load_rows = [&](f32x4* dst, const float* base, int64_t ld, int rows, int iters, int64_t p) {
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            if (i < iters) {
                const int r = sr + 16 * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                const int64_t n = p + sc;
                if (r < rows) {
                    const float* src = base + (int64_t)r * ld + n;
                    if constexpr (VEC) {
                        if (n < p_end) v = *reinterpret_cast<const f32x4*>(src);   // p_end % 4 == 0
                    } else {
                        if (n + 0 < p_end) v.x = src[0];
                        if (n + 1 < p_end) v.y = src[1];
                        if (n + 2 < p_end) v.z = src[2];
                        if (n + 3 < p_end) v.w = src[3];
                    }
                }
                dst[i] = v;
            }
        }
    };

    f32x4 acc[T1][T2];
#pragma unroll
    for (int i = 0; i < T1; ++i)
#pragma unroll
        for (int j = 0; j < T2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wk = wave % WK;
    const int wt = wave / WK;
    const int wr = wt / WC, wc = wt % WC;
    const int li = lane & 15, kk = lane >> 4;
    const int px_per_wave = NT_KP / WK;            // 64, 32 or 16

    if (p_begin < p_end) {
        load_rows(ra, Ab, g.lda, rowsA, itA, p_begin);
        load_rows(rbv, Bb, g.ldb, rowsB, itB, p_begin);
    }
    for (int64_t p = p_begin; p < p_end; p += NT_KP) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            if (i < itA) {
                *reinterpret_cast<f32x4*>(&As[(sr + 16 * i) * NT_LD + sc]) = ra[i];
                sqa[i] += ra[i].x * ra[i].x + ra[i].y * ra[i].y + ra[i].z * ra[i].z + ra[i].w * ra[i].w;
            }
            if (i < itB) {
                *reinterpret_cast<f32x4*>(&Bs[(sr + 16 * i) * NT_LD + sc]) = rbv[i];
                sqb[i] += rbv[i].x * rbv[i].x + rbv[i].y * rbv[i].y + rbv[i].z * rbv[i].z + rbv[i].w * rbv[i].w;
            }
        }
        __syncthreads();
        if (p + NT_KP < p_end) {
            load_rows(ra, Ab, g.lda, rowsA, itA, p + NT_KP);
            load_rows(rbv, Bb, g.ldb, rowsB, itB, p + NT_KP);
        }
        for (int q = 0; q < px_per_wave; q += 16) {
            const int px = wk * px_per_wave + q + 4 * kk;
            f32x4 av[T1], bv[T2];
#pragma unroll
            for (int i = 0; i < T1; ++i)
                av[i] = *reinterpret_cast<const f32x4*>(&As[(wr * 16 * T1 + 16 * i + li) * NT_LD + px]);
#pragma unroll
            for (int j = 0; j < T2; ++j)
                bv[j] = *reinterpret_cast<const f32x4*>(&Bs[(wc * 16 * T2 + 16 * j + li) * NT_LD + px]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < T1; ++i)
#pragma unroll
                    for (int j = 0; j < T2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
        }
    }

    // ---- combine the WK pixel-split waves through LDS, then write the slab ----
    __syncthreads();
    float* red = smem;        // reuse: [WK][WR*WC][T1*T2][64 lanes][4]
    if (WK > 1) {
#pragma unroll
        for (int i = 0; i < T1; ++i)
#pragma unroll
            for (int j = 0; j < T2; ++j)
                *reinterpret_cast<f32x4*>(&red[(((wk * (WR * WC) + wt) * (T1 * T2) + i * T2 + j) * 64 + lane) * 4]) = acc[i][j];
        __syncthreads();
    }
    float* slab = g.partial + ((int64_t)batch * g.splits + sp) * (int64_t)g.M1 * g.M2;
    if (wk == 0) {
#pragma unroll
        for (int i = 0; i < T1; ++i)
#pragma unroll
            for (int j = 0; j < T2; ++j) {
                f32x4 v = acc[i][j];
                for (int w = 1; w < WK; ++w) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(&red[(((w * (WR * WC) + wt) * (T1 * T2) + i * T2 + j) * 64 + lane) * 4]);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                const int col = cb * R2 + wc * 16 * T2 + 16 * j + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rb * R1 + wr * 16 * T1 + 16 * i + 4 * kk + r;
                    if (row < g.M1 && col < g.M2) slab[(int64_t)row * g.M2 + col] = v[r];
                }
            }
    }
    // ---- squared norms: the 16 lanes (tid&15) that staged a row hold its partial sums ----
    if (g.sqA_partial != nullptr && cb == 0) {
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            if (i < itA) {
                float s = sqa[i];
                s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
                const int row = rb * R1 + sr + 16 * i;
                if ((tid & 15) == 0 && row < g.M1)
                    g.sqA_partial[((int64_t)batch * g.splits + sp) * g.M1 + row] = s;
            }
        }
    }
    if (g.sqB_partial != nullptr && rb == 0) {
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            if (i < itB) {
                float s = sqb[i];
                s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
                const int row = cb * R2 + sr + 16 * i;
                if ((tid & 15) == 0 && row < g.M2)
                    g.sqB_partial[((int64_t)batch * g.splits + sp) * g.M2 + row] = s;
            }
        }
    }
}

// out[g][e] (+)= sum_{i < slabs_per_group} partial[(g*slabs_per_group + i)][e], fixed order
__global__ void reduce_slabs_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                    int64_t elems, int slabs_per_group, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t grp = blockIdx.y;
    if (e >= elems) return;
    const float* p = partial + grp * slabs_per_group * elems + e;
    float s = 0.f;
    for (int i = 0; i < slabs_per_group; ++i) s += p[(int64_t)i * elems];
    float* dst = out + grp * elems + e;
    *dst = accumulate ? (*dst + s) : s;
}

// Same sum for elems % 4 == 0: a block covers 128 elements with 32 float4 columns x 8 slab lanes;
// lane l adds slabs l, l+8, ... in order, then the 8 lanes are added in order through LDS.
__global__ void __launch_bounds__(256)
reduce_slabs_vec_kernel(const float* __restrict__ partial, float* __restrict__ out,
                        int64_t elems, int slabs_per_group, int accumulate) {
    __shared__ f32x4 red[8][32];
    const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int64_t e = ((int64_t)blockIdx.x * 32 + col) * 4;
    const int64_t grp = blockIdx.y;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (e < elems) {
        const float* p = partial + grp * slabs_per_group * elems + e;
        int i = sl;
        for (; i + 24 < slabs_per_group; i += 32) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p + (int64_t)i * elems);
            const f32x4 b = *reinterpret_cast<const f32x4*>(p + (int64_t)(i + 8) * elems);
            const f32x4 c = *reinterpret_cast<const f32x4*>(p + (int64_t)(i + 16) * elems);
            const f32x4 d = *reinterpret_cast<const f32x4*>(p + (int64_t)(i + 24) * elems);
            s += a; s += b; s += c; s += d;
        }
        for (; i < slabs_per_group; i += 8) s += *reinterpret_cast<const f32x4*>(p + (int64_t)i * elems);
    }
    red[sl][col] = s;
    __syncthreads();
    if (sl == 0 && e < elems) {
        f32x4 t = red[0][col];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][col];
        f32x4* dst = reinterpret_cast<f32x4*>(out + grp * elems + e);
        if (accumulate) t += *dst;
        *dst = t;
    }
}

// Tile plan along one output dimension: 16*T rows per wave, W waves.
static NtDim nt_pick_dim(int M, int maxW) {
    NtDim best{1, 1, (int)cdiv(M, 16)};
    int64_t best_pad = INT64_MAX;
    for (int W = 1; W <= maxW; W *= 2)
        for (int T = 1; T <= 4; ++T) {
            const int rows = 16 * T * W;
            if (rows > 128) continue;
            const int blocks = (int)cdiv(M, rows);
            const int64_t pad = (int64_t)blocks * rows;
            // least padding; ties -> fewer blocks (bigger tiles), then fewer waves (keeps WK large)
            if (pad < best_pad || (pad == best_pad && (blocks < best.blocks || (blocks == best.blocks && W < best.W)))) {
                best = NtDim{T, W, blocks}; best_pad = pad;
            }
        }
    return best;
}


NtPlan gemm_nt_plan(int M1, int M2, int64_t N, int nbatch, int max_splits_cap) {
    NtPlan p;
    p.d1 = nt_pick_dim(M1, 2);
    p.d2 = nt_pick_dim(M2, 2);
    const int64_t tiles = (int64_t)p.d1.blocks * p.d2.blocks * nbatch;
    // aim for ~1024 blocks on the chip, at least 512 pixels per block
    int64_t want = cdiv(1024, tiles);
    int64_t max_splits = cdiv(N, 512);
    if (want > max_splits) want = max_splits;
    if (max_splits_cap > 0 && want > max_splits_cap) want = max_splits_cap;
    if (want < 1) want = 1;
    p.chunk = round_up(cdiv(N, want), 64);
    p.splits = (int)cdiv(N, p.chunk);
    return p;
}

size_t gemm_nt_partial_floats(const NtPlan& p, int M1, int M2, int nbatch) {
    return (size_t)nbatch * p.splits * (size_t)M1 * M2;
}

int gemm_nt_launch(GemmNT g, const NtPlan& p, hipStream_t stream) {
    RST_REQUIRE(g.M1 > 0 && g.M2 > 0 && g.N > 0 && g.nb1 > 0 && g.nb2 > 0, "gemm_nt: empty problem");
    RST_REQUIRE(g.A && g.Bm && g.partial, "gemm_nt: null operand");
    g.splits = p.splits; g.chunk = p.chunk;
    g.walk = xcd_walk_enabled();
    if (gram_f16s_supported(g)) return gram_f16s_launch(g, stream);        // (RST_F32_MFMA=1 keeps the exact-fp32 kernel)
    const int WR = p.d1.W, WC = p.d2.W;
    const int64_t grid = (int64_t)g.nb1 * g.nb2 * p.splits * p.d1.blocks * p.d2.blocks;
    RST_REQUIRE(grid < (1ll << 31), "gemm_nt: grid too large");
    const bool vec = (g.N % 4 == 0) && (g.lda % 4 == 0) && (g.ldb % 4 == 0) && (g.a_s1 % 4 == 0) && (g.a_s2 % 4 == 0) &&
                     (g.b_s1 % 4 == 0) && (g.b_s2 % 4 == 0) && aligned16(g.A) && aligned16(g.Bm);
    const int R1 = 16 * p.d1.T * WR, R2 = 16 * p.d2.T * WC;
    const int WK = 4 / (WR * WC);
    const int SP = WK == 4 ? 64 : 32;
    const int rows_per_wave_piece = 4 * (256 / SP);                 // rows covered by one piece of each of the 4 waves
    const int PPW = (int)cdiv(R1 + R2, rows_per_wave_piece);
    RST_REQUIRE(PPW <= 8, "gemm_nt: tile too large");
    size_t lds = (size_t)NT_STAGES * PPW * 4 * 256 * sizeof(float);
    const size_t red = (size_t)WK * WR * WC * p.d1.T * p.d2.T * 64 * 4 * sizeof(float);
    if (WK > 1 && red > lds) lds = red;
    dim3 gd((unsigned)grid), blk(256);
    if (!vec) {
        size_t glds = (size_t)(R1 + R2) * NT_LD * sizeof(float);
        if (WK > 1 && red > glds) glds = red;
        ProfScope prof(stream, "gemm_nt_splitk", 2.0 * g.M1 * g.M2 * g.N * g.nb1 * g.nb2,
                       4.0 * g.nb1 * g.nb2 * ((double)(g.M1 + g.M2) * g.N + (double)p.splits * g.M1 * g.M2));
#define NTG_CASE(a, b) \
        if (p.d1.T == a && p.d2.T == b) hipLaunchKernelGGL((gemm_nt_generic_kernel<a, b, false>), gd, blk, glds, stream, g, WR, WC, p.d1.blocks, p.d2.blocks); else
        NTG_CASE(1, 1) NTG_CASE(1, 2) NTG_CASE(1, 3) NTG_CASE(1, 4)
        NTG_CASE(2, 1) NTG_CASE(2, 2) NTG_CASE(2, 3) NTG_CASE(2, 4)
        NTG_CASE(3, 1) NTG_CASE(3, 2) NTG_CASE(3, 3) NTG_CASE(3, 4)
        NTG_CASE(4, 1) NTG_CASE(4, 2) NTG_CASE(4, 3) NTG_CASE(4, 4)
        { return fail(RST_ERR_INVALID, "gemm_nt: bad tile"); }
#undef NTG_CASE
        RST_LAUNCH_CHECK();
        return RST_OK;
    }
    const double nb = (double)g.nb1 * g.nb2;
    ProfScope prof(stream, "gemm_nt_splitk", 2.0 * g.M1 * g.M2 * g.N * nb,
                   4.0 * nb * ((double)(g.M1 + g.M2) * g.N + (double)p.splits * g.M1 * g.M2));
#define NT_CASE(a, b)                                                                                         \
    if (p.d1.T == a && p.d2.T == b) {                                                                         \
        if (lds > 64 * 1024)                                                                                  \
            RST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<a, b>),                  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
        hipLaunchKernelGGL((gemm_nt_kernel<a, b>), gd, blk, lds, stream, g, WR, WC, p.d1.blocks, p.d2.blocks, SP, PPW); \
    } else
    NT_CASE(1, 1) NT_CASE(1, 2) NT_CASE(1, 3) NT_CASE(1, 4)
    NT_CASE(2, 1) NT_CASE(2, 2) NT_CASE(2, 3) NT_CASE(2, 4)
    NT_CASE(3, 1) NT_CASE(3, 2) NT_CASE(3, 3) NT_CASE(3, 4)
    NT_CASE(4, 1) NT_CASE(4, 2) NT_CASE(4, 3) NT_CASE(4, 4)
    { return fail(RST_ERR_INVALID, "gemm_nt: bad tile"); }
#undef NT_CASE
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// The vector form for up to four tensors with the same (groups, slabs per group) in one launch: the three sums of an
// attention (Gram matrix, squared norms of q and of k) are reduced together.  Same per-element order as above.
struct ReduceJobs { const float* partial[4]; float* out[4]; int64_t elems[4]; int first_block[5]; };
__global__ void __launch_bounds__(256)
reduce_slabs_multi_kernel(const ReduceJobs jobs, int slabs_per_group) {
    __shared__ f32x4 red[8][32];
    int j = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) j += (int)blockIdx.x >= jobs.first_block[k] ? 1 : 0;
    const float* partial = jobs.partial[0]; float* out = jobs.out[0]; int64_t elems = jobs.elems[0]; int fb = jobs.first_block[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (j == k) { partial = jobs.partial[k]; out = jobs.out[k]; elems = jobs.elems[k]; fb = jobs.first_block[k]; }
    const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int64_t e = ((int64_t)(blockIdx.x - fb) * 32 + col) * 4;
    const int64_t grp = blockIdx.y;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (e < elems) {
        const float* p = partial + grp * slabs_per_group * elems + e;
        int i = sl;
        for (; i + 24 < slabs_per_group; i += 32) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p + (int64_t)i * elems);
            const f32x4 b = *reinterpret_cast<const f32x4*>(p + (int64_t)(i + 8) * elems);
            const f32x4 c = *reinterpret_cast<const f32x4*>(p + (int64_t)(i + 16) * elems);
            const f32x4 d = *reinterpret_cast<const f32x4*>(p + (int64_t)(i + 24) * elems);
            s += a; s += b; s += c; s += d;
        }
        for (; i < slabs_per_group; i += 8) s += *reinterpret_cast<const f32x4*>(p + (int64_t)i * elems);
    }
    red[sl][col] = s;
    __syncthreads();
    if (sl == 0 && e < elems) {
        f32x4 t = red[0][col];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][col];
        *reinterpret_cast<f32x4*>(out + grp * elems + e) = t;
    }
}

int reduce_slabs3_launch(const float* p0, float* o0, int64_t e0, const float* p1, float* o1, int64_t e1, const float* p2, float* o2,
                         int64_t e2, int groups, int slabs_per_group, hipStream_t stream) {
    const bool vec = e0 % 4 == 0 && e1 % 4 == 0 && e2 % 4 == 0 && aligned16(p0) && aligned16(p1) && aligned16(p2) && aligned16(o0) &&
                     aligned16(o1) && aligned16(o2);
    if (!vec) {
        RST_TRY(reduce_slabs_launch(p0, o0, e0, groups, slabs_per_group, 0, stream));
        RST_TRY(reduce_slabs_launch(p1, o1, e1, groups, slabs_per_group, 0, stream));
        return reduce_slabs_launch(p2, o2, e2, groups, slabs_per_group, 0, stream);
    }
    ReduceJobs jobs{};
    const float* ps[3] = {p0, p1, p2}; float* os[3] = {o0, o1, o2}; const int64_t es[3] = {e0, e1, e2};
    int nb = 0;
    for (int k = 0; k < 3; ++k) { jobs.partial[k] = ps[k]; jobs.out[k] = os[k]; jobs.elems[k] = es[k]; jobs.first_block[k] = nb; nb += (int)cdiv(es[k], 128); }
    jobs.partial[3] = p2; jobs.out[3] = o2; jobs.elems[3] = 0; jobs.first_block[3] = nb; jobs.first_block[4] = nb;
    ProfScope prof(stream, "reduce_slabs", (double)(e0 + e1 + e2) * groups * slabs_per_group, 4.0 * (e0 + e1 + e2) * groups * (slabs_per_group + 1.0));
    hipLaunchKernelGGL(reduce_slabs_multi_kernel, dim3((unsigned)nb, (unsigned)groups), dim3(256), 0, stream, jobs, slabs_per_group);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

int reduce_slabs_launch(const float* partial, float* out, int64_t elems, int groups, int slabs_per_group,
                        int accumulate, hipStream_t stream) {
    dim3 gd((unsigned)cdiv(elems, 256), (unsigned)groups), blk(256);
    ProfScope prof(stream, "reduce_slabs", (double)elems * groups * slabs_per_group,
                   4.0 * elems * groups * (slabs_per_group + 1.0));
    if (elems % 4 == 0 && aligned16(partial) && aligned16(out)) {
        dim3 gv((unsigned)cdiv(elems, 128), (unsigned)groups);
        hipLaunchKernelGGL(reduce_slabs_vec_kernel, gv, blk, 0, stream, partial, out, elems, slabs_per_group, accumulate);
    } else {
        hipLaunchKernelGGL(reduce_slabs_kernel, gd, blk, 0, stream, partial, out, elems, slabs_per_group, accumulate);
    }
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst
