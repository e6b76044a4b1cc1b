// Y = dw3x3(W1 * LN(X))  /  Y = gelu(d[:Hd]) * d[Hd:]  in one kernel (reference model/Restromer.py:130-131,116-124 and
// :100-103,87-95), with the 1x1 product on the f16 matrix instructions at fp32 accuracy.
//
// Both operands are split in two, v = hi + lo / 2048 with hi = f16(v), lo = f16((v - hi) * 2048) -- 22 significant
// bits, v - hi is exact in fp32 -- and
//     W x = Whi xhi + (Whi xlo + Wlo xhi) / 2048            (the lo-lo term is below fp32 rounding)
// costs three f16 MFMAs over 32 (or 16) channels where the exact-fp32 instruction needs eight (four) over 4 channels
// each at twice the cycles: the matrix pipe stops being the bound and every wave can carry its own depthwise work.
// Inputs must stay inside the f16 range (|x| < 65504); RST_F32_MFMA=1 routes to the exact-fp32 kernels instead.
//
// Decomposition (as pwdw.hip): a block owns a 64-pixel column strip of one sample, a row range and NW slabs of 16
// output channels, and walks down the rows; wave w owns slab w.  The input channels of a row stream through an LDS
// ring by LDS-DMA in chunks of KC channels.  Per chunk phase (one barrier each), every wave
//   1. multiplies the PREVIOUS chunk out of a double-buffered image of split operand fragments (5 tiles: 4 x 16
//      strip pixels + one tile that carries the two halo columns), weights in registers as split B operands;
//   2. when that chunk was the last of a row: takes the row out of the accumulators and runs the depthwise phase on
//      it in registers -- pixels are the MFMA rows, so a lane holds 4 consecutive pixels of ONE channel: horizontal
//      taps are neighbouring registers / lanes, vertical taps two running sums per tile (pwdw.hip);
//   3. converts its share of the CURRENT chunk (one tile per wave) from the fp32 ring into fragments, summing x and
//      x^2 per pixel on the way: the folded LayerNorm's statistics cost no pass of their own.
// The k order inside a fragment is permuted (lane group g holds channels {g + 4 i}) identically for both operands,
// which leaves the product unchanged and makes the four groups read four different bank quarters of the ring.
#include "kernels.hpp"

namespace rst {

namespace {

typedef __attribute__((address_space(1))) const void* pf_gptr;
typedef __attribute__((address_space(3))) void* pf_lptr;
typedef unsigned pf_u32x4 __attribute__((ext_vector_type(4)));

constexpr int PF_T = 4;                  // strip tiles of 16 pixels
constexpr int PF_WT = 16 * PF_T;
constexpr int PF_CPR = 20;               // 16-byte pieces per channel row in the ring (4 px halo + 64 + 4 px halo, padded)
constexpr int PF_ROWF = PF_CPR * 4;      // floats per ring row; == 16 (mod 64)
constexpr int PF_STAGES = 4;
constexpr int pf_ninstr(int kc) { return (kc * PF_CPR + 63) / 64; }
constexpr int pf_xch(int kc) { return pf_ninstr(kc) * 256; }

template <int KC> struct PfFrag;
template <> struct PfFrag<32> {
    typedef _Float16 type __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x4 mfma(type a, type b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct PfFrag<16> {
    typedef _Float16 type __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 mfma(type a, type b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
};

// gelu(x) = x * Phi(x) without a branch (Abramowitz-Stegun 7.1.26, <= 4.3e-7 absolute on gelu in fp32)
__device__ __forceinline__ float pf_gelu(float x) {
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

// SPREAD: the depthwise phase of a row is spread over the chunk phases of the next row (balanced phases, 16 more live
// registers: two waves per SIMD); otherwise it runs in the first phase (three waves per SIMD).
template <int K, int KC, int NW, bool GATE, bool SPREAD>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(SPREAD ? 2 : 3, SPREAD ? 4 : 3)))
pwdw_f16_kernel(PwDw a, int nstrips, int nsplit, int G, int ncols) {
    typedef typename PfFrag<KC>::type frag;
    constexpr int FE = KC / 4;                           // f16 values per fragment lane
    constexpr int NC = K / KC;                           // chunks per row
    static_assert(K % KC == 0 && NC >= 2, "whole chunks, at least two per row");
    constexpr int T = PF_T, WT = PF_WT, ROWF = PF_ROWF, XCH = pf_xch(KC), NINSTR = pf_ninstr(KC);
    constexpr int P = (NINSTR + NW - 1) / NW;            // DMA instructions per wave per chunk
    constexpr int NIT = (T + 1 + NW - 1) / NW;           // conversion items (tiles + halo tile) per wave
    constexpr int PAIRS = NW / 2, HT = T / 2;
    extern __shared__ __attribute__((aligned(16))) float pf_smem[];
    float* ring = pf_smem;                               // PF_STAGES stages + 256 floats for surplus DMA pieces
    float* statm = ring + PF_STAGES * XCH + 256;         // [2][80] minus mean (or 0) per ring column, by row parity
    float* statr = statm + 2 * ROWF;                     // [2][80] 1/sqrt(var + eps)
    frag* img = reinterpret_cast<frag*>(statr + 2 * ROWF);                   // [2][5][2][64] fragments
    float* xch = reinterpret_cast<float*>(img + 2 * 5 * 2 * 64);             // GATE: [PAIRS][2][HT][64] float4 hand-over area

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kk = lane >> 4;
    const int H = a.H, W = a.W;
    const int64_t HW = (int64_t)H * W;

    // blocks 8 apart share an XCD (and its L2): the G channel groups of one strip are placed there
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = slot % G;
    const int col = (slot / G) * 8 + xcd;
    if (col >= ncols) return;
    const int sp = col % nsplit, strip = (col / nsplit) % nstrips, b = col / (nsplit * nstrips);
    const int px0 = strip * WT;
    const int R0 = (int)((int64_t)H * sp / nsplit), R1 = (int)((int64_t)H * (sp + 1) / nsplit);
    const int ra = R0 > 0 ? R0 - 1 : 0, rb = R1 < H ? R1 : H - 1;
    const int nrows = rb - ra + 1, nq = nrows * NC;

    const int half = GATE ? wave / PAIRS : 0;
    const int ch = (GATE ? (g * PAIRS + wave % PAIRS) * 16 : (g * NW + wave) * 16) + lm;
    const int chlim = GATE ? a.hidden : a.M;
    const bool chok = ch < chlim;
    const int wrow = (GATE ? half * a.hidden : 0) + (chok ? ch : 0);

    // ---- weights of this lane's channel as split B operands ----
    frag whi[NC], wlo[NC];
    {
        const float* Ar = a.A + (int64_t)wrow * a.K;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < FE; ++i) {
                const float w = chok ? Ar[KC * c + kk + 4 * i] : 0.f;
                const _Float16 h = (_Float16)w;
                whi[c][i] = h;
                wlo[c][i] = (_Float16)((w - (float)h) * 2048.f);
            }
    }
    for (int i = tid; i < 2 * ROWF; i += NW * 64) { statm[i] = 0.f; statr[i] = 1.f; }      // (0, 1) = no LayerNorm
    float wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = chok ? a.dw_w[(int64_t)wrow * 9 + t] : 0.f;
    const float lns = (chok && a.ln_s) ? a.ln_s[wrow] : 0.f;
    const float lnc = (chok && a.ln_c) ? a.ln_c[wrow] : 0.f;
    const float bq = (chok && a.dw_b) ? a.dw_b[wrow] : 0.f;

    // ---- DMA pieces of this wave: ring column c holds pixel px0 - 4 + c ----
    const char* Xb = reinterpret_cast<const char*>(a.X + (int64_t)b * a.x_bs);
    unsigned srcoff[P];
    int dsto[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int i = wave + p * NW;
        if (i < NINSTR) {
            const int e = 64 * i + lane;
            const int k = e / PF_CPR < KC ? e / PF_CPR : KC - 1, cc = e % PF_CPR;
            const int px = px0 - 4 + 4 * cc;
            srcoff[p] = 4u * ((unsigned)(k * HW) + ((px >= 0 && px < W) ? px : 0));     // out-of-image pieces read a valid address and are masked later
            dsto[p] = 256 * i;
        } else {
            srcoff[p] = 0;
            dsto[p] = -1;
        }
    }
    const char* isrc = Xb + 4 * (int64_t)ra * W;
    int ic = 0, istage = 0, ileft = nq;
    const int64_t step_c = 4 * (int64_t)KC * HW, step_r = 4 * (int64_t)W - (NC - 1) * step_c;
    auto issue = [&]() __attribute__((always_inline)) {
        const char* src = isrc;
        float* sbase = ring + istage * XCH;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float* dst = dsto[p] >= 0 ? sbase + dsto[p] : ring + PF_STAGES * XCH;
            __builtin_amdgcn_global_load_lds((pf_gptr)(src + srcoff[p]), (pf_lptr)dst, 16, 0, 0);
        }
        if (ileft > 1) {                                 // past the end the last chunk is issued again (fixed count in flight)
            const bool wrap = ic == NC - 1;
            isrc += wrap ? step_r : step_c;
            ic = wrap ? 0 : ic + 1;
            istage = istage == PF_STAGES - 1 ? 0 : istage + 1;
            --ileft;
        }
    };

    // ---- stores through a buffer descriptor over the 16 channel planes of the slab (out-of-range lanes are dropped) ----
    const int slab0 = __builtin_amdgcn_readfirstlane(ch - lm);
    const int slab_n = chlim - slab0 < 0 ? 0 : (chlim - slab0 > 16 ? 16 : chlim - slab0);
    float* slab_base = a.Y + (int64_t)b * a.y_bs + (int64_t)slab0 * HW;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        slab_base, 0, (int)__builtin_amdgcn_readfirstlane((int)((int64_t)slab_n * HW * 4)), 0x00020000);
    const unsigned yoff = (unsigned)(((int64_t)lm * HW + px0 + 4 * kk) * 4);
    auto store_tile = [&](const f32x4& v, int row, int j, bool rowok) __attribute__((always_inline)) {
        const int px = px0 + 16 * j + 4 * kk;
        const unsigned vo = (rowok && px < W) ? yoff + (unsigned)(row * W * 4) + 64u * j : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, v), yrsrc, vo, 0, 0);
    };

    // ---- conversion items of this wave ----
    int ccol[NIT];
    bool cvalid[NIT], cdo[NIT];
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
        const int it = wave + n * NW;
        cdo[n] = it <= T;
        ccol[n] = it < T ? 4 + 16 * it + lm : (lm == 0 ? 3 : 4 + WT);       // halo tile: row 0 = column left of the strip, row 1 = right of it
        cvalid[n] = it < T || lm < 2;
    }
    float s1[NIT], s2[NIT];
#pragma unroll
    for (int n = 0; n < NIT; ++n) { s1[n] = 0.f; s2[n] = 0.f; }

    // ---- depthwise state: two running sums per tile (rows r and r + 1), see pwdw.hip ----
    f32x4 Pst[T], Qst[T], acc1[T + 1], acc2[T + 1];
#pragma unroll
    for (int j = 0; j < T; ++j) { Pst[j] = f32x4{0.f, 0.f, 0.f, 0.f}; Qst[j] = f32x4{bq, bq, bq, bq}; }
#pragma unroll
    for (int j = 0; j <= T; ++j) { acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int nbr_next = ((lane + 16) & 63) * 4, nbr_prev = ((lane + 48) & 63) * 4;
    const bool in_l = px0 - 1 >= 0, in_r = px0 + WT < W;
    // GATE: waves w and w + PAIRS hold the two halves of channel ch; each hands the other half of the strip over
    f32x4 hold0 = {0.f, 0.f, 0.f, 0.f}, hold1 = {0.f, 0.f, 0.f, 0.f};
    int pend_row = -1;
    f32x4* myx = reinterpret_cast<f32x4*>(xch) + (GATE ? ((wave % PAIRS) * 2 + half) * HT * 64 + lane : 0);
    const f32x4* ptx = reinterpret_cast<const f32x4*>(xch) + (GATE ? ((wave % PAIRS) * 2 + (1 - half)) * HT * 64 + lane : 0);
    auto stash = [&](const f32x4& v, int j) __attribute__((always_inline)) {
        const bool mine = (j < HT) == (half == 0);
        if (mine) { if (j % HT == 0) hold0 = v; else hold1 = v; }
        else myx[(j % HT) * 64] = v;
    };
    auto finalize = [&](int row) __attribute__((always_inline)) {
#pragma unroll
        for (int jh = 0; jh < HT; ++jh) {
            const f32x4 other = ptx[jh * 64];
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float hv = jh == 0 ? hold0[q] : hold1[q];
                const float x1 = half == 0 ? hv : other[q];
                const float x2 = half == 0 ? other[q] : hv;
                o[q] = pf_gelu(x1) * x2;
            }
            store_tile(o, row >= 0 ? row : 0, half * HT + jh, row >= 0);
        }
    };
    // depthwise phase of the finished 1x1 row `prow` held in tp[] (raw) and hl / hr (raw halo columns)
    f32x4 tp[T];
    float hlf = 0.f, hrf = 0.f;
    int prow = 0;
    // part 1: LayerNorm constants, bias, zero padding of the row in tp[]
    auto dw_prepare = [&](float hl, float hr) __attribute__((always_inline)) {
        const float* sm = statm + (prow & 1) * ROWF;
        const float* sr = statr + (prow & 1) * ROWF;
        hlf = in_l ? fmaf(sr[3], fmaf(sm[3], lns, hl), lnc) : 0.f;
        hrf = in_r ? fmaf(sr[4 + WT], fmaf(sm[4 + WT], lns, hr), lnc) : 0.f;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const f32x4 tco = *reinterpret_cast<const f32x4*>(&sm[4 + 16 * j + 4 * kk]);
            const f32x4 rstd = *reinterpret_cast<const f32x4*>(&sr[4 + 16 * j + 4 * kk]);
            const bool live = px0 + 16 * j + 4 * kk < W;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = fmaf(rstd[q], fmaf(tco[q], lns, tp[j][q]), lnc);
                tp[j][q] = live ? v : 0.f;               // the depthwise filter pads the 1x1 OUTPUT with zeros
            }
        }
    };
    // part 2: the 3x3 taps of tiles [j0, j1)
    auto dw_tiles = [&](int j0, int j1) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < T; ++j) {
            if (j < j0 || j >= j1) continue;
            float e[6];
            {
                const float same = __int_as_float(__builtin_amdgcn_ds_bpermute(nbr_prev, __float_as_int(tp[j][3])));
                const float other = j > 0 ? __int_as_float(__builtin_amdgcn_ds_bpermute(nbr_prev, __float_as_int(tp[j > 0 ? j - 1 : 0][3]))) : hlf;
                e[0] = kk == 0 ? other : same;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) e[1 + q] = tp[j][q];
            {
                const float same = __int_as_float(__builtin_amdgcn_ds_bpermute(nbr_next, __float_as_int(tp[j][0])));
                const float other = j + 1 < T ? __int_as_float(__builtin_amdgcn_ds_bpermute(nbr_next, __float_as_int(tp[j + 1 < T ? j + 1 : j][0]))) : hrf;
                e[5] = kk == 3 ? other : same;
            }
            f32x4 done, pn, qn;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                done[q] = fmaf(wv[8], e[q + 2], fmaf(wv[7], e[q + 1], fmaf(wv[6], e[q], Pst[j][q])));
                pn[q] = fmaf(wv[5], e[q + 2], fmaf(wv[4], e[q + 1], fmaf(wv[3], e[q], Qst[j][q])));
                qn[q] = fmaf(wv[2], e[q + 2], fmaf(wv[1], e[q + 1], fmaf(wv[0], e[q], bq)));
            }
            Pst[j] = pn; Qst[j] = qn;
            if constexpr (GATE) stash(done, j);
            else store_tile(done, prow >= 1 ? prow - 1 : 0, j, prow - 1 >= R0 && prow - 1 < R1);
        }
    };

#pragma unroll
    for (int i = 0; i < PF_STAGES - 1; ++i) issue();
    int stage_c = 0, q = 0;
    constexpr int NEWER = (PF_STAGES - 2) * P;
    static_assert(NEWER <= 63, "vmcnt is a 6-bit counter");

    for (int rr = 0; rr <= nrows; ++rr) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            // chunk q has landed for this wave (two newer chunks may be in flight), then for everyone; LDS writes of
            // the previous phase (fragments, statistics, hand-over) are visible after the barrier
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NEWER) : "memory");
            issue();
            if (q > 0 && q <= nq) {
                const int cw = (c + NC - 1) % NC;        // the chunk converted in the previous phase
                const frag* im = img + ((q - 1) & 1) * 5 * 2 * 64 + lane;
#pragma unroll
                for (int j = 0; j <= T; ++j) {
                    const frag ah = im[(j * 2 + 0) * 64], al = im[(j * 2 + 1) * 64];
                    acc1[j] = PfFrag<KC>::mfma(ah, whi[cw], acc1[j]);
                    acc2[j] = PfFrag<KC>::mfma(ah, wlo[cw], acc2[j]);
                    acc2[j] = PfFrag<KC>::mfma(al, whi[cw], acc2[j]);
                }
            }
            if constexpr (GATE) {
                // the hand-over written in the last phase of the previous row is final behind this phase's barrier
                if (c == 0 && pend_row >= 0) { finalize(pend_row); pend_row = -1; }
            }
            if (c == 0 && rr > 0) {
                // the 1x1 row ra + rr - 1 is complete; its depthwise phase is spread over this row's chunk phases
#pragma unroll
                for (int j = 0; j < T; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) tp[j][e] = fmaf(acc2[j][e], 1.0f / 2048.f, acc1[j][e]);
                // halo tile: lanes of group 0 hold row 0 (left column) and row 1 (right column) of their channel
                const float h0 = fmaf(acc2[T][0], 1.0f / 2048.f, acc1[T][0]), h1 = fmaf(acc2[T][1], 1.0f / 2048.f, acc1[T][1]);
                const float hl = __shfl(h0, lm, 64), hr = __shfl(h1, lm, 64);
#pragma unroll
                for (int j = 0; j <= T; ++j) { acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                prow = ra + rr - 1;
                dw_prepare(hl, hr);
                if constexpr (!SPREAD) dw_tiles(0, T);
                else if constexpr (NC == 2) dw_tiles(0, T / 2);
            }
            if (rr > 0) {
                if constexpr (SPREAD) {
                    if constexpr (NC == 2) { if (c == 1) dw_tiles(T / 2, T); }
                    else { if (c == 1) dw_tiles(0, T / 2); if (c == 2) dw_tiles(T / 2, T); }
                }
                // (the hand-over is read behind the next row's first barrier; its writes end in this row's last or first phase)
                if constexpr (GATE) { if (c == NC - 1) pend_row = (prow - 1 >= R0 && prow - 1 < R1) ? prow - 1 : -1; }
            }
            if (q < nq) {
                const float* Xs = ring + stage_c * XCH + kk * ROWF;
#pragma unroll
                for (int n = 0; n < NIT; ++n) {
                    if (cdo[n]) {
                        float v[FE];
#pragma unroll
                        for (int i = 0; i < FE; ++i) v[i] = Xs[4 * i * ROWF + ccol[n]];
                        frag hi, lo;
#pragma unroll
                        for (int i = 0; i < FE; ++i) {
                            const float x = cvalid[n] ? v[i] : 0.f;
                            const _Float16 h = (_Float16)x;
                            hi[i] = h;
                            lo[i] = (_Float16)((x - (float)h) * 2048.f);
                            s1[n] += x;
                            s2[n] = fmaf(x, x, s2[n]);
                        }
                        const int it = wave + n * NW;
                        img[(((q & 1) * 5 + it) * 2 + 0) * 64 + lane] = hi;
                        img[(((q & 1) * 5 + it) * 2 + 1) * 64 + lane] = lo;
                    }
                }
                if (c == NC - 1) {
                    // per-pixel LayerNorm constants of row ra + rr (read by its depthwise phase, two barriers from here)
                    const int r = ra + rr;
#pragma unroll
                    for (int n = 0; n < NIT; ++n) {
                        float t1 = s1[n], t2 = s2[n];
                        t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
                        t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
                        const float mu = t1 * (1.0f / K);
                        const float var = fmaxf(t2 * (1.0f / K) - mu * mu, 0.f);
                        if (cdo[n] && cvalid[n] && kk == 0 && a.ln_mode != 0) {
                            statm[(r & 1) * ROWF + ccol[n]] = a.ln_mode == 1 ? -mu : 0.f;
                            statr[(r & 1) * ROWF + ccol[n]] = 1.0f / sqrtf(var + 1e-5f);
                        }
                        s1[n] = 0.f; s2[n] = 0.f;
                    }
                }
            }
            stage_c = stage_c == PF_STAGES - 1 ? 0 : stage_c + 1;
            ++q;
        }
    }
    // all DMA (including the re-issues past the end) must have landed before the block's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (GATE) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (pend_row >= 0) finalize(pend_row);
    }
    if (R1 == H) {
        // bottom edge of the image: the row below is zero padding, so the running sum of the last row is final
        if constexpr (GATE) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // partners are done reading the previous hand-over
#pragma unroll
            for (int j = 0; j < T; ++j) stash(Pst[j], j);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            finalize(H - 1);
        } else {
#pragma unroll
            for (int j = 0; j < T; ++j) store_tile(Pst[j], H - 1, j, true);
        }
    }
}

struct PfPlan { int nw, G, nstrips; bool ok; };
PfPlan pf_plan(const PwDw& a) {
    PfPlan p{};
    if (a.K != 48 && a.K != 96) return p;
    if (a.W % 4 != 0 || a.W < 64 || a.H < 2) return p;
    if (!aligned16(a.X) || a.x_bs % 4 != 0 || !aligned16(a.Y) || a.y_bs % 4 != 0) return p;
    if ((int64_t)a.K * a.H * a.W >= (1ll << 30)) return p;
    if (a.gate && a.M != 2 * a.hidden) return p;
    p.nstrips = (int)cdiv(a.W, PF_WT);
    if (a.gate) {
        p.nw = 8;
        p.G = (int)cdiv(cdiv(a.hidden, 16), 4);
    } else {
        const int slabs = (int)cdiv(a.M, 16);
        const int pad6 = (int)cdiv(slabs, 6) * 6, pad3 = (int)cdiv(slabs, 3) * 3;
        p.nw = pad6 <= pad3 ? 6 : 3;
        p.G = (int)cdiv(slabs, p.nw);
    }
    p.ok = true;
    return p;
}

}  // namespace

bool pwdw_f16_supported(const PwDw& a) { return pf_plan(a).ok; }

int pwdw_f16_launch(const PwDw& a, hipStream_t stream) {
    const PfPlan p = pf_plan(a);
    RST_REQUIRE(p.ok, "pwdw_f16: unsupported shape (K=%d M=%d W=%d)", a.K, a.M, a.W);
    RST_REQUIRE(a.X && a.A && a.dw_w && a.Y && a.B > 0 && a.M > 0, "pwdw_f16: bad operand");
    RST_REQUIRE(a.ln_mode == 0 || (a.ln_s && a.ln_c), "pwdw_f16: folded LayerNorm without its constants");
    const int KC = a.K == 96 ? 32 : 16;
    const void* kern = nullptr;
    // measured at 16 x K x 256 x 256: gated 48 channels 251 us spread (288 in one phase; exact-fp32 kernels 284), plain
    // 48 channels 307 us in one phase at three waves per SIMD (393 spread; exact-fp32 380)
    if (a.K == 96) {
        if (a.gate) kern = reinterpret_cast<const void*>(&pwdw_f16_kernel<96, 32, 8, true, true>);
        else if (p.nw == 6) kern = reinterpret_cast<const void*>(&pwdw_f16_kernel<96, 32, 6, false, true>);
        else kern = reinterpret_cast<const void*>(&pwdw_f16_kernel<96, 32, 3, false, true>);
    } else {
        if (a.gate) kern = reinterpret_cast<const void*>(&pwdw_f16_kernel<48, 16, 8, true, true>);
        else if (p.nw == 6) kern = reinterpret_cast<const void*>(&pwdw_f16_kernel<48, 16, 6, false, false>);
        else kern = reinterpret_cast<const void*>(&pwdw_f16_kernel<48, 16, 3, false, false>);
    }
    const size_t lds = sizeof(float) * (size_t)(PF_STAGES * pf_xch(KC) + 256 + 4 * PF_ROWF) +
                       (size_t)2 * 5 * 2 * 64 * (KC / 4) * 2 + (a.gate ? (size_t)(p.nw / 2) * 2 * (PF_T / 2) * 64 * 16 : 0);
    KernelSetup su;
    RST_TRY(kernel_setup(kern, p.nw * 64, lds, &su));
    // rows of every (sample, strip, channel group) column split so that the grid comes in whole rounds of resident
    // blocks; every row range recomputes two halo rows
    const int64_t cols0 = (int64_t)a.B * p.nstrips, slots = (int64_t)su.cus * su.per_cu;
    int nsplit = 1;
    double best = -1.0;
    for (int ns = 1; ns <= (a.H >= 16 ? a.H / 8 : 1); ++ns) {
        const int64_t blocks = cols0 * p.G * ns;
        const int64_t rounds = cdiv(blocks, slots);
        const double rows = (double)a.H / ns;
        const double eff = (double)blocks / (double)(rounds * slots) * rows / (rows + 3.0);
        if (eff > best + 1e-9) { best = eff; nsplit = ns; }
    }
    const int64_t ncols64 = cols0 * nsplit;
    RST_REQUIRE(ncols64 * p.G < (1ll << 30), "pwdw_f16: grid too large");
    const int ncols = (int)ncols64, nstrips = p.nstrips, G = p.G;
    const unsigned grid = (unsigned)(cdiv(ncols64, 8) * 8 * G);
    const double N = (double)a.B * a.H * a.W;
    ProfScope prof(stream, a.gate ? "pwdw_gate_f16" : "pwdw_f16", 2.0 * a.M * a.K * N + 18.0 * a.M * N,
                   4.0 * N * (a.K + (a.gate ? a.hidden : a.M)));
    void* args[] = {(void*)&a, (void*)&nstrips, (void*)&nsplit, (void*)&G, (void*)&ncols};
    RST_HIP(hipLaunchKernel(kern, dim3(grid), dim3(p.nw * 64), args, lds, stream));
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst
