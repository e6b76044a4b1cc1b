// Wave-specialised form of the fused  Y = dw3x3(W1 * LN(X))  kernel (pwdw.hip has the single-role form
// and the description of the decomposition, the LDS ring and the LayerNorm folding; reference
// model/Restromer.py:130-131,116-124).
//
// The single-role kernel is bound by the issue rate of each wave's own instruction stream: every wave
// carries the MFMAs AND the depthwise vector work of its slab, and an in-order wave cannot issue its
// vector work while it waits for the matrix pipe.  Here the two kinds of work live in different waves:
//   * NM "matrix" waves: operand fetch + MFMA only (13 instructions per 4 MFMAs).  At the end of a row
//     they drop the raw 1x1 result of their slab (4 tiles x float4 per lane, plus the two halo-column
//     partial sums) into an LDS hand-off buffer.
//   * NM "vector" waves, one per slab: they own the DMA stream of the block (a matrix wave never touches
//     vmcnt), take the LayerNorm statistics from the ring, and run the depthwise phase of the PREVIOUS
//     row out of the hand-off buffer -- LayerNorm constants, zero padding, 3x3 taps with the two running
//     partial sums per tile, stores -- while the matrix waves multiply the next row.
// Matrix and vector pipes are separate and instructions of different waves issue side by side, so the
// depthwise work costs the matrix waves nothing.  12 waves of a block sit 3/3/3/3 on the SIMDs (six matrix
// waves 2/2/1/1, six vector waves 1/1/2/2).  All waves meet at one barrier per ring chunk; the hand-off
// buffer is single: it is read in the first chunk phase of a row and rewritten after the last.
//
// GATE (Y = gelu(d[:Hd]) * d[Hd:], model/Restromer.py:100-103): eight matrix waves = four slabs of each half;
// eight vector waves = (slab pair) x (half of the strip): a vector wave runs the depthwise phase of BOTH halves
// for its two tiles and multiplies them, so nothing has to be exchanged between waves; the one pixel it needs
// from the other half of the strip is read raw from the hand-off buffer and finished on the spot.
#pragma once
#include "kernels.hpp"
#include <type_traits>

namespace rst {

typedef __attribute__((address_space(1))) const void* ws_gptr;
typedef __attribute__((address_space(3))) void* ws_lptr;
typedef float ws_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned ws_u32x4 __attribute__((ext_vector_type(4)));

// Timing experiments only (results are wrong): -DWS_ABL=1 vector waves without the depthwise arithmetic and stores,
// 2 also without picking the row up, 3 matrix waves without MFMAs, 4 also without the conversion, 5 no stores
#ifndef WS_ABL
#define WS_ABL 0
#endif

constexpr int WS_T = 4;                  // MFMA tiles per strip row
constexpr int WS_WT = 16 * WS_T;         // strip width
constexpr int WS_CPR = 20;               // 16-byte pieces per channel row in LDS (4 px halo + 64 + 4 px halo, padded)
constexpr int WS_ROWF = WS_CPR * 4;      // == 16 (mod 64): the four k-rows of an operand read hit disjoint banks
constexpr int WS_NSG = WS_ROWF / 16;     // groups of 16 ring columns for the statistics
constexpr int WS_STAGES_F32 = 4;
// ring depth: the split-f16 form is short of bytes in flight, not of LDS (its weights are in registers): 8 stages
constexpr int ws_stages(bool f16s) { return f16s ? 8 : WS_STAGES_F32; }
// channels per ring stage (two stages per row at least); the split-f16 form multiplies 32 channels per instruction
constexpr int ws_ck(int ks, bool f16s = false) { return f16s ? 32 : ks >= 24 ? 48 : 24; }
constexpr int ws_ninstr(int ks, bool f16s = false) { return (ws_ck(ks, f16s) * WS_CPR + 63) / 64; }     // wave-wide DMA instructions per chunk
constexpr int ws_xch(int ks, bool f16s = false) { return ws_ninstr(ks, f16s) * 256; }                   // floats per ring stage
constexpr int WS_IMG_FLOATS = 2 * 5 * 2 * 64 * 4;       // split-f16 operand image: [2 buffers][4 tiles + halo tile][hi, lo][64 lanes] x 16 B
typedef _Float16 ws_h8 __attribute__((ext_vector_type(8)));

// gelu(x) = x * Phi(x) without a branch: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 on erf, measured
// <= 4.3e-7 absolute on gelu over [-8, 8] in fp32), one reciprocal and one exponential.  The library erff has two
// branches, a wave usually takes both, and in the gated form the vector waves are what everyone waits for.
__device__ __forceinline__ float ws_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);
    const float q = p * t * e;                     // 1 - erf(z)
    const float phi = x >= 0.f ? fmaf(-0.5f, q, 1.0f) : 0.5f * q;
    return x * phi;
}

template <int I, int N, typename F>
__device__ __forceinline__ void ws_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ws_static_for<I + 1, N>(f);
    }
}

// F16S: the 1x1 product on the f16 matrix instructions at fp32 accuracy.  Both operands are split,
// v = hi + lo / 2048 with hi = f16(v), lo = f16((v - hi) * 2048) (22 significant bits; v - hi is exact in fp32), and
//   W x = Whi xhi + (Whi xlo + Wlo xhi) / 2048        (the lo-lo term is below fp32 rounding)
// takes three v_mfma_f32_16x16x32_f16 (16 cycles each for 32 channels) where the fp32 form takes eight
// v_mfma_f32_16x16x4_f32 (32 cycles each): the matrix pipe stops being the bound.  The matrix waves turn each ring
// chunk (32 channels) into operand fragments once per block (waves 0..3 one pixel tile each, wave 4 the two halo
// columns as a fifth tile) in a double-buffered LDS image, one phase ahead of the multiplication, so the hand-off of
// a finished row moves from the last phase of its row to the first phase of the next and the vector waves pick it
// up in the second.  Weights live in registers as split B operands.  |x| must stay below 65504 (f16 range).
// In this form the matrix waves also own the block's DMA stream and the LayerNorm statistics (the sums over the
// channels fall out of the conversion), so the vector waves carry the depthwise phase only.
// Ablations at 16 x 96 x 256 x 256 (plain / gated, us): full kernel 679 / 496; DMA stream and barriers alone 238 / 170
// (the input strip is read once per channel group, 3x / 2x: 1.2 GB at ~5 TB/s); matrix side without the vector
// waves' work 507 / 355; vector side without conversion and MFMAs 591 / 350; without stores 579 / 487: neither
// side nor memory is the single bound.
// With every LDS read of a phase issued behind its barrier (one round trip per phase instead of six on either side)
// the plain kernel went from 7.17 to 6.91 ms per step of 16 x 3 x 256^2 (20 launches), so the round trips were not
// the bound either.  The same step with parts compiled out (-DWS_ABL=n, ms for the 20 launches): no stores 5.27; vector
// waves without arithmetic and stores 5.40, without the pick-up as well 5.38; matrix waves without MFMAs and
// conversion 6.20.  What moves the time is memory: a block owns 1/256 of ~4.7 TB/s = 8.7 bytes per cycle, and a row
// of a block moves 24.6 KB out and 15 KB in from HBM (the strip is fetched 1.83 times for the three channel groups)
// = 4 550 cycles of the 7 000 a row takes.  Tried without gain: non-temporal stores (7.18; the Gram kernel behind it
// gains what this one loses), ten ring stages (6.96), half of the depthwise arithmetic in the pick-up phase (7.07),
// no packed fp32 arithmetic (7.04).
template <int KS, int NM, bool GATE, bool F16S = false>
__global__ void __launch_bounds__(NM * 128) __attribute__((amdgpu_waves_per_eu(GATE ? 4 : 3, GATE ? 4 : 3)))
pwdw_ws_kernel(PwDw a, int nstrips, int nsplit, int G, int ncols) {
    constexpr int STG = ws_stages(F16S);
    constexpr int WS_CK = ws_ck(KS, F16S), WS_NINSTR = ws_ninstr(KS, F16S), WS_XCH = ws_xch(KS, F16S);
    static_assert(!F16S || (KS % 8 == 0 && NM >= 5), "split-f16 form: whole 32-channel chunks, five converting waves");
    constexpr int HC = F16S ? 1 : 0;                     // chunk phase in which the vector waves pick a finished row up
    constexpr int T = WS_T, WT = WS_WT, ROWF = WS_ROWF, XCH = WS_XCH;
    constexpr int ND = NM;                               // vector waves
    constexpr int PAIRS = NM / 2;                        // GATE: slab pairs per block
    constexpr int NH = GATE ? 2 : 1;                     // halves a vector wave works on
    constexpr int TV = GATE ? T / 2 : T;                 // tiles a vector wave works on
    constexpr int TV0 = GATE ? 1 : 0;                    // of which in the first chunk phase (the rest in the last)
    constexpr int P = (WS_NINSTR + ND - 1) / ND;         // DMA instructions per vector wave per chunk
    constexpr int CKS = WS_CK / 4;                       // k-steps per chunk
    constexpr int NC = KS / CKS;                         // chunks per row
    static_assert(KS % CKS == 0 && NC >= 2, "the depthwise phase is split over the first and the last chunk of a row");
    constexpr int NTW = (WS_NSG + ND - 1) / ND;          // statistics groups per vector wave
    extern __shared__ __attribute__((aligned(16))) float ws_smem[];
    float* ring = ws_smem;                               // STG stages + 256 floats for surplus DMA pieces
    float* statm = ring + STG * XCH + 256;         // [2][80] minus mean (or 0) per ring column, by row parity
    float* statr = statm + 2 * ROWF;                     // [2][80] 1/sqrt(var + eps)
    float* hand = statr + 2 * ROWF;                      // [NM][T][64] float4: raw 1x1 result of the last finished row
    float* halo = hand + NM * T * 256;                   // [NM][2][64]: per-lane shares of the two halo columns
    float* wlds = halo + NM * 2 * 64;                    // [NM][KS][64] weights as MFMA B operand; F16S: the operand image instead

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kk = lane >> 4;
    const int H = a.H, W = a.W;
    const int64_t HW = (int64_t)H * W;

    // blocks 8 apart share an XCD (and its L2): the G channel groups of one strip are placed there.  Plain form: every
    // XCD also takes a contiguous range of columns with the strips of one row range next to each other, so the 16
    // halo pixels a strip shares with its neighbour come out of the same L2 (6.97 -> 6.79 ms per step; the gated form
    // loses with it, 5.53 -> 5.59, and keeps the interleaved order)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = slot % G;
    int col, sp, strip, b;
    if constexpr (!GATE) {
        const int cpx = (ncols + 7) >> 3;
        col = xcd * cpx + slot / G;
        if (slot / G >= cpx || col >= ncols) return;
        strip = col % nstrips; sp = (col / nstrips) % nsplit; b = col / (nsplit * nstrips);
    } else {
        col = (slot / G) * 8 + xcd;
        if (col >= ncols) return;
        sp = col % nsplit; strip = (col / nsplit) % nstrips; b = col / (nsplit * nstrips);
    }
    const int px0 = strip * WT;
    const int R0 = (int)((int64_t)H * sp / nsplit), R1 = (int)((int64_t)H * (sp + 1) / nsplit);
    const int ra = R0 > 0 ? R0 - 1 : 0, rb = R1 < H ? R1 : H - 1;
    const int nrows = rb - ra + 1;

    const bool matrix = wave < NM;
    const int slab = matrix ? wave : wave - NM;          // matrix: hand-off slot; vector: wave number
    // plain: slab = 16 output channels.  GATE: matrix wave m multiplies half m / PAIRS of pair m % PAIRS;
    // vector wave d works on pair d % PAIRS, tiles [tv0, tv0 + TV) with tv0 = (d / PAIRS) * TV
    const int pair = GATE ? slab % PAIRS : slab;
    const int part = GATE ? slab / PAIRS : 0;            // matrix: half; vector: half of the strip
    const int chlim = GATE ? a.hidden : a.M;
    const int slab0 = (g * (GATE ? PAIRS : NM) + pair) * 16;
    const int ch = slab0 + lm;
    const bool chok = ch < chlim;
    const int wrow = (GATE && matrix ? part * a.hidden : 0) + (chok ? ch : 0);

    if (matrix) {
        // ================================ matrix wave ================================
        if constexpr (F16S) {
            f32x4* myhand = reinterpret_cast<f32x4*>(hand) + slab * T * 64 + lane;       // + j * 64
            float* myhalo = halo + slab * 2 * 64 + lane;                                  // + 64: right column
            ws_h8* img = reinterpret_cast<ws_h8*>(wlds);                                  // [2][5][2][64]
            // lane group kk holds the channels {32 c + kk + 4 i}, i = 0..7, of chunk c -- the same set for both
            // operands, which is all the instruction needs; with it the four groups of a fragment read four
            // different bank quarters of the ring
            ws_h8 whi[NC], wlo[NC];
            {
                const float* Ar = a.A + (int64_t)wrow * a.K;
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float w = chok ? Ar[32 * c + kk + 4 * i] : 0.f;
                        const _Float16 h = (_Float16)w;
                        whi[c][i] = h;
                        wlo[c][i] = (_Float16)((w - (float)h) * 2048.f);
                    }
            }
            // this wave's share of the conversion: tile `slab` (waves 0..3), the halo columns (wave 4: row 0 = the
            // column left of the strip, row 1 = the one right of it, other rows zero)
            const int ccol = slab < T ? 4 + 16 * slab + lm : (lm == 0 ? 3 : 4 + WT);
            const bool cvalid = slab < T || lm < 2;
            float cs1 = 0.f, cs2 = 0.f;                  // LayerNorm sums of this lane's pixel column over its channel share
            // Every LDS read of a phase is issued right behind its barrier -- the ten operand fragments of the chunk
            // to multiply, then the eight ring values of the chunk to convert -- and waited for by count as it is
            // needed (LDS returns in order): one exposed LDS round trip per phase instead of six.  (Left to the
            // compiler the reads went out two at a time, each pair followed by a full wait: 2 800 cycles per phase,
            // of which the matrix pipe worked 240.)
            ws_h8 ah[T + 1], al[T + 1];
            ws_f32x2 cv[4];
            auto fetch_phase = [&](int stage, int buf_mul) __attribute__((always_inline)) {
                const unsigned ia = (unsigned)(uintptr_t)(ws_lptr)(img + buf_mul * 5 * 2 * 64 + lane);
                ws_static_for<0, T + 1>([&](auto Jc) {
                    constexpr int j = decltype(Jc)::value;
                    ws_h8 &rh = ah[j], &rl = al[j];                 // (an asm operand cannot name a captured variable itself)
                    const unsigned addr = ia;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rh) : "v"(addr), "n"((j * 2 + 0) * 1024));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rl) : "v"(addr), "n"((j * 2 + 1) * 1024));
                });
                const unsigned ca = (unsigned)(uintptr_t)(ws_lptr)(ring + stage * XCH + kk * ROWF + ccol);
                ws_static_for<0, 4>([&](auto Pc) {
                    constexpr int p = decltype(Pc)::value;         // channels kk + 8 p and kk + 8 p + 4 of the chunk
                    ws_f32x2& r = cv[p];
                    const unsigned addr = ca;
                    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(addr), "n"(10 * p), "n"(10 * p + 5));
                });
            };
            static_assert(ROWF * 4 * 4 == 5 * 256, "ds_read2st64 offsets above assume 320 floats between the channels of a lane");
            auto convert = [&](int buf) __attribute__((always_inline)) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cv[0]), "+v"(cv[1]), "+v"(cv[2]), "+v"(cv[3]));
                if (slab <= T) {
                    ws_h8 hi, lo;
                    typedef _Float16 ws_h2 __attribute__((ext_vector_type(2)));
                    ws_f32x2 p1 = {0.f, 0.f}, p2 = {0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        // two channels at a time: packed conversions, the residual (x - hi) * 2048 as fma(hi, -2048, 2048 x)
                        const ws_f32x2 x = {cvalid ? cv[i / 2][0] : 0.f, cvalid ? cv[i / 2][1] : 0.f};
                        const ws_h2 h = __builtin_convertvector(x, ws_h2);
                        const ws_f32x2 sc = x * 2048.f;
                        const ws_f32x2 r2 = {fmaf((float)h[0], -2048.f, sc[0]), fmaf((float)h[1], -2048.f, sc[1])};
                        const ws_h2 l = __builtin_convertvector(r2, ws_h2);
                        hi[i] = h[0]; hi[i + 1] = h[1];
                        lo[i] = l[0]; lo[i + 1] = l[1];
                        p1 += x;
                        p2 += x * x;
                    }
                    cs1 += p1[0] + p1[1];
                    cs2 += p2[0] + p2[1];
                    img[((buf * 5 + slab) * 2 + 0) * 64 + lane] = hi;
                    img[((buf * 5 + slab) * 2 + 1) * 64 + lane] = lo;
                }
            };
            // ---- the block's DMA stream (the matrix waves have the time: their MFMAs fill a tenth of a phase) ----
            constexpr int PM = (WS_NINSTR + NM - 1) / NM;
            const char* Xb = reinterpret_cast<const char*>(a.X + (int64_t)b * a.x_bs);
            unsigned srcoff[PM];
            int dsto[PM];
#pragma unroll
            for (int p = 0; p < PM; ++p) {
                const int i = slab + p * NM;
                if (i < WS_NINSTR) {
                    const int e = 64 * i + lane;
                    const int k = e / WS_CPR < WS_CK ? e / WS_CPR : WS_CK - 1, cc = e % WS_CPR;
                    const int px = px0 - 4 + 4 * cc;
                    srcoff[p] = 4u * ((unsigned)(k * HW) + ((px >= 0 && px < W) ? px : 0));
                    dsto[p] = 256 * i;
                } else {
                    srcoff[p] = 0;
                    dsto[p] = -1;
                }
            }
            const int nq = nrows * NC;
            const char* isrc = Xb + 4 * (int64_t)ra * W;
            int ic = 0, istage = 0, ileft = nq;
            const int64_t step_c = 4 * (int64_t)WS_CK * HW, step_r = 4 * (int64_t)W - (NC - 1) * step_c;
            auto issue = [&]() __attribute__((always_inline)) {
                const char* src = isrc;
                float* sbase = ring + istage * XCH;
#pragma unroll
                for (int p = 0; p < PM; ++p) {
                    float* dst = dsto[p] >= 0 ? sbase + dsto[p] : ring + STG * XCH;
                    __builtin_amdgcn_global_load_lds((ws_gptr)(src + srcoff[p]), (ws_lptr)dst, 16, 0, 0);
                }
                if (ileft > 1) {
                    const bool wrap = ic == NC - 1;
                    isrc += wrap ? step_r : step_c;
                    ic = wrap ? 0 : ic + 1;
                    istage = istage == STG - 1 ? 0 : istage + 1;
                    --ileft;
                }
            };
            f32x4 acc1[T + 1], acc2[T + 1];
#pragma unroll
            for (int j = 0; j <= T; ++j) { acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int i = 0; i < STG - 1; ++i) issue();
            int stage_c = 0, q = 0;
            constexpr int NEWER_M = (STG - 2) * PM;
            for (int rr = 0; rr <= nrows; ++rr) {
                ws_static_for<0, NC>([&](auto Cc) {
                    constexpr int c = decltype(Cc)::value;
                    constexpr int cw = (c + NC - 1) % NC;            // chunk multiplied in this phase (converted in the previous one)
                    // chunk q landed for this wave (two newer chunks may be in flight), then for everyone
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NEWER_M) : "memory");
                    issue();
                    fetch_phase(stage_c, (q + 1) & 1);
                    if (WS_ABL != 3 && WS_ABL != 4 && q > 0 && q <= nq) {
                        ws_static_for<0, T + 1>([&](auto Jc) {
                            constexpr int j = decltype(Jc)::value;
                            // fragments 0 .. j are there; the younger ones and the four conversion reads may be in flight
                            ws_h8 &rh = ah[j], &rl = al[j];
                            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(rh), "+v"(rl) : "n"(4 + 2 * (T - j)));
                            acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[j], whi[cw], acc1[j], 0, 0, 0);
                            acc2[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[j], wlo[cw], acc2[j], 0, 0, 0);
                        });
#pragma unroll
                        for (int j = 0; j <= T; ++j)
                            acc2[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[j], whi[cw], acc2[j], 0, 0, 0);
                    }
                    if (c == 0 && rr > 0) {
                        // row rr - 1 is complete: hand it over (the vector waves read it in the next phase)
#pragma unroll
                        for (int j = 0; j < T; ++j) {
                            f32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = fmaf(acc2[j][e], 1.0f / 2048.f, acc1[j][e]);
                            myhand[j * 64] = o;
                        }
                        myhalo[0] = kk == 0 ? fmaf(acc2[T][0], 1.0f / 2048.f, acc1[T][0]) : 0.f;
                        myhalo[64] = kk == 0 ? fmaf(acc2[T][1], 1.0f / 2048.f, acc1[T][1]) : 0.f;
#pragma unroll
                        for (int j = 0; j <= T; ++j) { acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                    }
                    if (WS_ABL != 4 && q < nq) {
                        convert(q & 1);
                        if (c == NC - 1) {
                            // per-pixel LayerNorm constants of row ra + rr, read by the vector waves in the next row's
                            // second phase (two barriers from here)
                            float t1 = cs1, t2 = cs2;
                            t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
                            t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
                            const float mu = t1 * (1.0f / (4 * KS));
                            const float var = fmaxf(t2 * (1.0f / (4 * KS)) - mu * mu, 0.f);
                            if (slab <= T && cvalid && kk == 0 && a.ln_mode != 0) {
                                const int r = ra + rr;
                                statm[(r & 1) * ROWF + ccol] = a.ln_mode == 1 ? -mu : 0.f;
                                statr[(r & 1) * ROWF + ccol] = 1.0f / sqrtf(var + 1e-5f);
                            }
                            cs1 = 0.f; cs2 = 0.f;
                        }
                    }
                    stage_c = stage_c == STG - 1 ? 0 : stage_c + 1;
                    ++q;
                });
            }
            // all DMA (including the re-issues past the end) must have landed before the block's LDS is released
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        float* mywl = wlds + slab * KS * 64 + lane;
        {
            const float* Ar = a.A + (int64_t)wrow * a.K;
#pragma unroll
            for (int s = 0; s < KS; ++s) mywl[s * 64] = chok ? Ar[4 * s + kk] : 0.f;
        }
        f32x4* myhand = reinterpret_cast<f32x4*>(hand) + slab * T * 64 + lane;       // + j * 64
        float* myhalo = halo + slab * 2 * 64 + lane;                                  // + 64: right column
        int stage_c = 0;
        for (int rr = 0; rr < nrows; ++rr) {
            f32x4 acc[T];
#pragma unroll
            for (int j = 0; j < T; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            float hl = 0.f, hr = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const float* Xc = ring + stage_c * XCH + kk * ROWF;
                const unsigned tb = (unsigned)(uintptr_t)(ws_lptr)(Xc + lm);
                const unsigned hb = (unsigned)(uintptr_t)(ws_lptr)Xc;
                const unsigned wb = (unsigned)(uintptr_t)(ws_lptr)(mywl + c * CKS * 64);
                ws_f32x2 x01[3], x23[3], xh[3];
                float wq[3];
                // operands of a k-step, issued two steps ahead and waited for by count (see pwdw.hip)
                auto fetch = [&](auto Sc) {
                    constexpr int S = decltype(Sc)::value, SET = S % 3;
                    const unsigned tbs = tb + S * 4 * ROWF * 4, hbs = hb + S * 4 * ROWF * 4, wbs = wb;
                    ws_f32x2 &r01 = x01[SET], &r23 = x23[SET], &rh = xh[SET];
                    float& rw = wq[SET];
                    asm volatile("ds_read2_b32 %0, %1 offset0:4 offset1:20" : "=v"(r01) : "v"(tbs));
                    asm volatile("ds_read2_b32 %0, %1 offset0:36 offset1:52" : "=v"(r23) : "v"(tbs));
                    asm volatile("ds_read2_b32 %0, %1 offset0:3 offset1:68" : "=v"(rh) : "v"(hbs));
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(rw) : "v"(wbs), "n"(S * 256));
                };
                fetch(std::integral_constant<int, 0>{});
                fetch(std::integral_constant<int, 1>{});
                ws_static_for<0, CKS>([&](auto Sc) {
                    constexpr int S = decltype(Sc)::value, SET = S % 3;
                    if constexpr (S + 2 < CKS) fetch(std::integral_constant<int, S + 2>{});
                    constexpr int LATER = (S + 2 < CKS ? 2 : S + 1 < CKS ? 1 : 0) * 4;
                    ws_f32x2 &r01 = x01[SET], &r23 = x23[SET], &rh = xh[SET];
                    float& rw = wq[SET];
                    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r01), "+v"(r23), "+v"(rh), "+v"(rw) : "n"(LATER));
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x01[SET][0], wq[SET], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x01[SET][1], wq[SET], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x23[SET][0], wq[SET], acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x23[SET][1], wq[SET], acc[3], 0, 0, 0);
                    hl = fmaf(wq[SET], xh[SET][0], hl);
                    hr = fmaf(wq[SET], xh[SET][1], hr);
                });
                stage_c = stage_c == STG - 1 ? 0 : stage_c + 1;
            }
            // hand the row over (the vector wave read the previous row in the first chunk phase of this one)
#pragma unroll
            for (int j = 0; j < T; ++j) myhand[j * 64] = acc[j];
            myhalo[0] = hl;
            myhalo[64] = hr;
        }
        // the vector waves work one row behind: keep them company at the barriers of their last row
#pragma unroll
        for (int c = 0; c < NC; ++c) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        return;
    }

    // ================================ vector wave ================================
    const int dwave = slab;                              // 0 .. ND-1
    // exact-fp32 form: the vector waves are the younger waves of their SIMD and would only get the issue slots the
    // matrix waves leave; they have the shorter stream and everyone waits for them at the chunk barrier: let them go
    // first.  In the split-f16 form the matrix waves carry the DMA stream, the conversion and the statistics as well,
    // and a raised vector priority starves their DMA issue (barrier time stamps of one block: an issue of two DMA
    // instructions took up to 5 000 cycles in the phase in which the vector waves run their taps): no priority there
    // (7.64 -> 7.23 ms and 5.81 -> 5.51 ms per step for the plain and the gated kernel).
    if constexpr (!F16S) __builtin_amdgcn_s_setprio(3);
    for (int i = tid - NM * 64; i < 2 * ROWF; i += ND * 64) { statm[i] = 0.f; statr[i] = 1.f; }     // (0, 1) = no LayerNorm
    const int tv0 = part * TV;                           // first tile of this wave
    float wv[NH][9], lns[NH], lnc[NH], bq[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int64_t wr = (int64_t)h * a.hidden + wrow;
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[h][t] = chok ? a.dw_w[wr * 9 + t] : 0.f;
        lns[h] = (chok && a.ln_s) ? a.ln_s[wr] : 0.f;
        lnc[h] = (chok && a.ln_c) ? a.ln_c[wr] : 0.f;
        bq[h] = (chok && a.dw_b) ? a.dw_b[wr] : 0.f;
    }

    // ---- DMA pieces of this wave: ring column c holds pixel px0 - 4 + c ----
    const char* Xb = reinterpret_cast<const char*>(a.X + (int64_t)b * a.x_bs);
    unsigned srcoff[P];
    int dsto[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int i = dwave + p * ND;
        if (i < WS_NINSTR) {
            const int e = 64 * i + lane;
            const int k = e / WS_CPR < WS_CK ? e / WS_CPR : WS_CK - 1, cc = e % WS_CPR;
            const int px = px0 - 4 + 4 * cc;
            srcoff[p] = 4u * ((unsigned)(k * HW) + ((px >= 0 && px < W) ? px : 0));     // out-of-image pieces read a valid address and are masked later
            dsto[p] = 256 * i;
        } else {
            srcoff[p] = 0;
            dsto[p] = -1;
        }
    }
    const int nq = nrows * NC;
    const char* isrc = Xb + 4 * (int64_t)ra * W;
    int ic = 0, istage = 0, ileft = nq;
    const int64_t step_c = 4 * (int64_t)WS_CK * HW, step_r = 4 * (int64_t)W - (NC - 1) * step_c;
    auto issue = [&]() __attribute__((always_inline)) {
        const char* src = isrc;
        float* sbase = ring + istage * XCH;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float* dst = dsto[p] >= 0 ? sbase + dsto[p] : ring + STG * XCH;
            __builtin_amdgcn_global_load_lds((ws_gptr)(src + srcoff[p]), (ws_lptr)dst, 16, 0, 0);
        }
        if (ileft > 1) {
            const bool wrap = ic == NC - 1;
            isrc += wrap ? step_r : step_c;
            ic = wrap ? 0 : ic + 1;
            istage = istage == STG - 1 ? 0 : istage + 1;
            --ileft;
        }
    };

    // stores through a buffer descriptor over the 16 channel planes of the slab (out-of-range lanes are dropped)
    const int slab_n = chlim - slab0 < 0 ? 0 : (chlim - slab0 > 16 ? 16 : chlim - slab0);
    float* slab_base = a.Y + (int64_t)b * a.y_bs + (int64_t)slab0 * HW;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        slab_base, 0, (int)__builtin_amdgcn_readfirstlane((int)((int64_t)slab_n * HW * 4)), 0x00020000);
    const unsigned yoff = (unsigned)(((int64_t)lm * HW + px0 + 4 * kk) * 4);
    auto store_tile = [&](const f32x4& v, int row, int j, bool rowok) __attribute__((always_inline)) {
        const int px = px0 + 16 * j + 4 * kk;
        const unsigned vo = (rowok && px < W) ? yoff + (unsigned)(row * W * 4) + 64u * j : 0x80000000u;
        if (WS_ABL != 5) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, v), yrsrc, vo, 0, 0);
    };

    // Plain split-f16 form: a lane holds 4 pixels of ONE channel per tile, so a tile's store writes 64-byte pieces of 16
    // channel planes.  The four tiles of a row go through LDS instead -- the wave's own hand-off slab, free between its
    // pick-up and the next row's hand-over, 16 channels x 64 pixels with the 16-byte chunks of channel c rotated by
    // c (conflict-free both ways) -- and leave as four stores of four whole 256-byte row segments each: 6.85 -> 6.49 ms
    // per step of 16 x 3 x 256 x 256 (20 launches; the same bytes with made-up lane order, i.e. without the trip through
    // LDS: 6.12).  The gated form writes 128-byte pieces of 60 channels and gains a tenth of that: left as it is.
    constexpr bool COAL = F16S && !GATE;
    static_assert(!COAL || TV0 == 0, "all tiles of a row are finished in one phase");
    float* scrT = hand + slab * T * 256;
    auto flush_row = [&](int row, bool rowok) __attribute__((always_inline)) {
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            const int cp = 4 * jp + (lane >> 4), m = lane & 15;
            const f32x4 v = *reinterpret_cast<const f32x4*>(scrT + cp * 64 + ((m ^ cp) & 15) * 4);
            const int px = px0 + 4 * m;
            const unsigned vo = (rowok && px < W) ? (unsigned)((((int64_t)cp * HW + px) * 4)) + (unsigned)(row * W * 4) : 0x80000000u;
            if (WS_ABL != 5) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_u32x4, v), yrsrc, vo, 0, 0);
        }
    };

    f32x4 Pst[NH][TV], Qst[NH][TV], tp[NH][TV];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < TV; ++j) {
            Pst[h][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            Qst[h][j] = f32x4{bq[h], bq[h], bq[h], bq[h]};
            tp[h][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const int nbr_next = ((lane + 16) & 63) * 4, nbr_prev = ((lane + 48) & 63) * 4;
    const bool in_l = px0 - 1 >= 0, in_r = px0 + WT < W;
    float el[NH][TV], er[NH][TV];                        // the pixel before / after the four of a tile
    // finished depthwise output of a tile -> memory (GATE: gelu(first half) * second half)
    auto emit = [&](const f32x4 (&d)[NH], int row, int j, bool rowok) __attribute__((always_inline)) {
        if constexpr (GATE) {
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = ws_gelu(d[0][q]) * d[NH - 1][q];
            store_tile(o, row, j, rowok);
        } else if constexpr (COAL) {
            *reinterpret_cast<f32x4*>(scrT + lm * 64 + (((4 * j + kk) ^ lm) & 15) * 4) = d[0];
        } else {
            store_tile(d[0], row, j, rowok);
        }
    };
    // depthwise 3x3 of tile jj (of this wave): row prow's share of output rows prow-1 (now complete), prow, prow+1
    auto dw_tile = [&](int jj, int prow) __attribute__((always_inline)) {
        f32x4 done[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float e[6];
            e[0] = el[h][jj]; e[5] = er[h][jj];
#pragma unroll
            for (int q = 0; q < 4; ++q) e[1 + q] = tp[h][jj][q];
            f32x4 pn, qn;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                done[h][q] = fmaf(wv[h][8], e[q + 2], fmaf(wv[h][7], e[q + 1], fmaf(wv[h][6], e[q], Pst[h][jj][q])));
                pn[q] = fmaf(wv[h][5], e[q + 2], fmaf(wv[h][4], e[q + 1], fmaf(wv[h][3], e[q], Qst[h][jj][q])));
                qn[q] = fmaf(wv[h][2], e[q + 2], fmaf(wv[h][1], e[q + 1], fmaf(wv[h][0], e[q], bq[h])));
            }
            Pst[h][jj] = pn; Qst[h][jj] = qn;
        }
        emit(done, prow >= 1 ? prow - 1 : 0, tv0 + jj, prow - 1 >= R0 && prow - 1 < R1);
    };

    if constexpr (!F16S) {
#pragma unroll
        for (int i = 0; i < STG - 1; ++i) issue();
    }
    int stage_c = 0;

    for (int rr = 0; rr <= nrows; ++rr) {
        const int r = ra + rr, prow = r - 1;             // row being multiplied, row in the depthwise phase
        const bool have_row = rr < nrows, have_prev = rr > 0;
        ws_f32x2 st[NTW];
#pragma unroll
        for (int jj = 0; jj < NTW; ++jj) st[jj] = ws_f32x2{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            // chunk landed for this wave (two newer chunks may be in flight), then for everyone
            constexpr int NEWER = (STG - 2) * P;
            if constexpr (F16S) {
                // (the split-f16 form: the matrix waves own the DMA stream and the LayerNorm statistics)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NEWER) : "memory");
                issue();
            }
            const float* Xs = ring + stage_c * XCH;
            // this wave's ring columns for the LayerNorm statistics: read now, summed at the end of the phase
            constexpr bool SAHEAD = !GATE && KS >= 24;      // statistics columns read ahead of their use (costs NTW * CKS registers)
            float sx[NTW][SAHEAD ? CKS : 1];
            const bool do_stats = !F16S && have_row && a.ln_mode != 0;
            if (SAHEAD && do_stats) {
#pragma unroll
                for (int jj = 0; jj < NTW; ++jj) {
                    const int sg = dwave + jj * ND < WS_NSG ? dwave + jj * ND : WS_NSG - 1;
#pragma unroll
                    for (int s = 0; s < CKS; ++s) sx[jj][s] = Xs[(4 * s + kk) * ROWF + 16 * sg + lm];
                }
            }

            if (WS_ABL != 2 && c == HC && have_prev) {
                if constexpr (F16S && !GATE) {
                    // ---- raw row out of the hand-off buffer; LayerNorm constants, bias, zero padding ----
                    // Every LDS read of the phase goes out first and nothing is computed until they are all under way
                    // (one LDS round trip; in source order the compiler waited after each group of reads: six).
                    const float* sm = statm + (prow & 1) * ROWF;
                    const float* sr = statr + (prow & 1) * ROWF;
                    // the pixel left of the first tile and right of the last one: a halo column of the strip, or
                    // (GATE) a pixel of the other half of the strip, raw in the hand-off buffer
                    const int colL = tv0 == 0 ? 3 : 4 + 16 * tv0 - 1;
                    const int colR = tv0 + TV == T ? 4 + WT : 4 + 16 * (tv0 + TV);
                    const float smL = sm[colL], srL = sr[colL], smR = sm[colR], srR = sr[colR];
                    float hlq[NH][4], hrq[NH][4], nbl[NH], nbr[NH];
                    f32x4 rawt[NH][TV], tco[TV], rstd[TV];
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const int hs = GATE ? h * PAIRS + pair : slab;              // hand-off slot of this half
                        const f32x4* hd = reinterpret_cast<const f32x4*>(hand) + hs * T * 64;
                        const float* hh = halo + hs * 2 * 64 + lm;                  // + 16 * k: the k share
#pragma unroll
                        for (int k = 0; k < 4; ++k) { hlq[h][k] = hh[16 * k]; hrq[h][k] = hh[64 + 16 * k]; }
                        if constexpr (GATE) {
                            nbl[h] = reinterpret_cast<const float*>(hd + (tv0 > 0 ? tv0 - 1 : 0) * 64 + 48 + lm)[3];
                            nbr[h] = reinterpret_cast<const float*>(hd + (tv0 + TV < T ? tv0 + TV : T - 1) * 64 + lm)[0];
                        } else {
                            nbl[h] = 0.f; nbr[h] = 0.f;
                        }
#pragma unroll
                        for (int j = 0; j < TV; ++j) rawt[h][j] = hd[(tv0 + j) * 64 + lane];
                    }
#pragma unroll
                    for (int j = 0; j < TV; ++j) {
                        tco[j] = *reinterpret_cast<const f32x4*>(&sm[4 + 16 * (tv0 + j) + 4 * kk]);
                        rstd[j] = *reinterpret_cast<const f32x4*>(&sr[4 + 16 * (tv0 + j) + 4 * kk]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        float hlp = 0.f, hrp = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { hlp += hlq[h][k]; hrp += hrq[h][k]; }
                        const float rl = tv0 == 0 ? hlp : nbl[h], rr_ = tv0 + TV == T ? hrp : nbr[h];
                        const bool okl = tv0 == 0 ? in_l : px0 + colL - 4 < W;
                        const bool okr = tv0 + TV == T ? in_r : px0 + colR - 4 < W;
                        const float outl = okl ? fmaf(srL, fmaf(smL, lns[h], rl), lnc[h]) : 0.f;
                        const float outr = okr ? fmaf(srR, fmaf(smR, lns[h], rr_), lnc[h]) : 0.f;
#pragma unroll
                        for (int j = 0; j < TV; ++j) {
                            const int jt = tv0 + j;
                            const bool live = px0 + 16 * jt + 4 * kk < W;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float v = fmaf(rstd[j][q], fmaf(tco[j][q], lns[h], rawt[h][j][q]), lnc[h]);
                                tp[h][j][q] = live ? v : 0.f;
                            }
                        }
                        // horizontal neighbours across lane groups and tiles
#pragma unroll
                        for (int j = 0; j < TV; ++j) {
                            const float same_l = __int_as_float(__builtin_amdgcn_ds_bpermute(nbr_prev, __float_as_int(tp[h][j][3])));
                            const float oth_l = j > 0 ? __int_as_float(__builtin_amdgcn_ds_bpermute(nbr_prev, __float_as_int(tp[h][j > 0 ? j - 1 : 0][3]))) : outl;
                            el[h][j] = kk == 0 ? oth_l : same_l;
                            const float same_r = __int_as_float(__builtin_amdgcn_ds_bpermute(nbr_next, __float_as_int(tp[h][j][0])));
                            const float oth_r = j + 1 < TV ? __int_as_float(__builtin_amdgcn_ds_bpermute(nbr_next, __float_as_int(tp[h][j + 1 < TV ? j + 1 : j][0]))) : outr;
                            er[h][j] = kk == 3 ? oth_r : same_r;
                        }
                    }
                } else {
                // (the forms at their register limit -- gated: 128, exact fp32: 168 -- keep the reads next to their use)
                    // ---- raw row out of the hand-off buffer; LayerNorm constants, bias, zero padding ----
                    const float* sm = statm + (prow & 1) * ROWF;
                    const float* sr = statr + (prow & 1) * ROWF;
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const int hs = GATE ? h * PAIRS + pair : slab;              // hand-off slot of this half
                        const f32x4* hd = reinterpret_cast<const f32x4*>(hand) + hs * T * 64;
                        const float* hh = halo + hs * 2 * 64 + lm;                  // + 16 * k: the k share
                        // the pixel left of the first tile and right of the last one: a halo column of the strip, or
                        // (GATE) a pixel of the other half of the strip, raw in the hand-off buffer
                        float outl, outr;
                        if (tv0 == 0) {
                            float hlp = 0.f;
#pragma unroll
                            for (int k = 0; k < 4; ++k) hlp += hh[16 * k];
                            outl = in_l ? fmaf(sr[3], fmaf(sm[3], lns[h], hlp), lnc[h]) : 0.f;
                        } else {
                            const int colx = 4 + 16 * tv0 - 1;
                            const float raw = reinterpret_cast<const float*>(hd + (tv0 - 1) * 64 + 48 + lm)[3];
                            outl = px0 + colx - 4 < W ? fmaf(sr[colx], fmaf(sm[colx], lns[h], raw), lnc[h]) : 0.f;
                        }
                        if (tv0 + TV == T) {
                            float hrp = 0.f;
#pragma unroll
                            for (int k = 0; k < 4; ++k) hrp += hh[64 + 16 * k];
                            outr = in_r ? fmaf(sr[4 + WT], fmaf(sm[4 + WT], lns[h], hrp), lnc[h]) : 0.f;
                        } else {
                            const int colx = 4 + 16 * (tv0 + TV);
                            const float raw = reinterpret_cast<const float*>(hd + (tv0 + TV) * 64 + lm)[0];
                            outr = px0 + colx - 4 < W ? fmaf(sr[colx], fmaf(sm[colx], lns[h], raw), lnc[h]) : 0.f;
                        }
#pragma unroll
                        for (int j = 0; j < TV; ++j) {
                            const int jt = tv0 + j;
                            const f32x4 raw = hd[jt * 64 + lane];
                            const f32x4 tco = *reinterpret_cast<const f32x4*>(&sm[4 + 16 * jt + 4 * kk]);
                            const f32x4 rstd = *reinterpret_cast<const f32x4*>(&sr[4 + 16 * jt + 4 * kk]);
                            const bool live = px0 + 16 * jt + 4 * kk < W;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float v = fmaf(rstd[q], fmaf(tco[q], lns[h], raw[q]), lnc[h]);
                                tp[h][j][q] = live ? v : 0.f;
                            }
                        }
                        // horizontal neighbours across lane groups and tiles
#pragma unroll
                        for (int j = 0; j < TV; ++j) {
                            const float same_l = __int_as_float(__builtin_amdgcn_ds_bpermute(nbr_prev, __float_as_int(tp[h][j][3])));
                            const float oth_l = j > 0 ? __int_as_float(__builtin_amdgcn_ds_bpermute(nbr_prev, __float_as_int(tp[h][j > 0 ? j - 1 : 0][3]))) : outl;
                            el[h][j] = kk == 0 ? oth_l : same_l;
                            const float same_r = __int_as_float(__builtin_amdgcn_ds_bpermute(nbr_next, __float_as_int(tp[h][j][0])));
                            const float oth_r = j + 1 < TV ? __int_as_float(__builtin_amdgcn_ds_bpermute(nbr_next, __float_as_int(tp[h][j + 1 < TV ? j + 1 : j][0]))) : outr;
                            er[h][j] = kk == 3 ? oth_r : same_r;
                        }
                    }
                }
#pragma unroll
                for (int jj = 0; jj < TV0; ++jj) dw_tile(jj, prow);
            }
            if (WS_ABL != 1 && WS_ABL != 2 && c == NC - 1 && have_prev) {
#pragma unroll
                for (int jj = TV0; jj < TV; ++jj) dw_tile(jj, prow);
                if constexpr (COAL) flush_row(prow >= 1 ? prow - 1 : 0, prow - 1 >= R0 && prow - 1 < R1);
            }
            // ---- this wave's share of the LayerNorm statistics of row r ----
            if (do_stats) {
#pragma unroll
                for (int jj = 0; jj < NTW; ++jj) {
                    const int sg = dwave + jj * ND < WS_NSG ? dwave + jj * ND : WS_NSG - 1;
#pragma unroll
                    for (int s = 0; s < CKS; ++s) {
                        // (the forms that are short of registers read here instead of ahead)
                        const float x = SAHEAD ? sx[jj][s] : Xs[(4 * s + kk) * ROWF + 16 * sg + lm];
                        st[jj][0] += x;
                        st[jj][1] = fmaf(x, x, st[jj][1]);
                    }
                }
            }
            stage_c = stage_c == STG - 1 ? 0 : stage_c + 1;
        }
        // per-pixel LayerNorm constants of row r, read in the next row's first phase (behind its barrier)
        if (!F16S && have_row && a.ln_mode != 0) {
#pragma unroll
            for (int jj = 0; jj < NTW; ++jj) {
                const int sg = dwave + jj * ND;
                if (sg < WS_NSG) {
                    float s1 = st[jj][0], s2 = st[jj][1];
                    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                    const float mu = s1 * (1.0f / (4 * KS));
                    const float var = fmaxf(s2 * (1.0f / (4 * KS)) - mu * mu, 0.f);
                    if (kk == 0) {
                        statm[(r & 1) * ROWF + 16 * sg + lm] = a.ln_mode == 1 ? -mu : 0.f;
                        statr[(r & 1) * ROWF + 16 * sg + lm] = 1.0f / sqrtf(var + 1e-5f);
                    }
                }
            }
        }
    }
    // all DMA (including the re-issues past the end) must have landed before the block's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (R1 == H) {
        // bottom edge of the image: the row below is zero padding, so the running sum of the last row is final
#pragma unroll
        for (int j = 0; j < TV; ++j) {
            f32x4 last[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) last[h] = Pst[h][j];
            emit(last, H - 1, tv0 + j, true);
        }
        if constexpr (COAL) flush_row(H - 1, true);
    }
}

}  // namespace rst
