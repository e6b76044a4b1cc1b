// Y = dw3x3(W1 * LN(X))            (qkv path of the attention,  reference model/Restromer.py:130-131,116-124)
// Y = gelu(d[:Hd]) * d[Hd:],  d = dw3x3(W1 * LN(X))     (gated FFN input, model/Restromer.py:100-103,87-95)
// in ONE kernel for the high-resolution levels: the 1x1 product never goes to memory.
//
// Decomposition.  A block owns a column strip of 64 pixels of one sample, a range of rows, and NW
// slabs of 16 output channels; it walks DOWN the rows.  Wave w owns slab w for the whole strip:
//   * its 16 weight rows sit in LDS in MFMA operand form (one float per lane and k-step) for the
//     whole launch;
//   * per row, the K input channels of the strip (plus four pixels either side) stream through a
//     4- or 5-deep LDS ring in chunks of 48 or 24 channels, filled by LDS-DMA only; the wave multiplies its
//     slab into 4 accumulator tiles of 16 pixels (v_mfma_f32_16x16x4_f32, exact fp32);
//   * the two halo columns (one pixel left and right of the strip) are plain dot products on the
//     vector ALU, taken from the same ring -- a fifth MFMA tile for two columns would cost 25 %;
//   * the finished row of the 1x1 output stays in registers.  The product is formed with the
//     pixels as MFMA rows, so a lane holds ONE channel and four consecutive pixels per tile:
//     the depthwise weights are nine scalars per lane, the horizontal taps are neighbouring
//     registers (the pixel before the first and after the fourth comes from the neighbouring
//     lane group, tile, or halo column), and stores are float4.  The rows above and below
//     enter through two running partial sums per output tile (row r adds its share to output
//     rows r-1, r, r+1; row r-1 is then complete and leaves for memory).
// The first and last row of a row range are recomputed by the neighbouring block (no exchange).
//
// LayerNorm in front of the product is folded as in gemm_nn (A = W o gamma, ln_s, ln_c): the
// per-pixel mean / variance over the K channels are taken from the LDS ring while the chunks
// pass (the waves share the 80 ring columns, results go through LDS once per row).
//
// What bounds the kernel is the issue rate of the wave's own instruction stream, not the matrix
// pipe, LDS latency or memory (timing-only builds without the depthwise pieces, without the
// statistics, without the DMA and without any LDS wait each take off just their own instructions'
// worth).  Hence: no branch inside a chunk (stores are dropped by the buffer range check instead of
// being jumped over, so that the depthwise pieces share a basic block with the MFMAs and are
// interleaved with them), stream position kept in running scalars, LDS reads placed by hand.
// Pixels outside the image are forced to zero AFTER the bias / LayerNorm constants, because the
// depthwise filter pads its input -- the 1x1 output -- with zeros.
//
// GATE: waves w and w + NW/2 hold channel c of the two halves; each hands the other half of its
// finished tiles over through LDS, so both apply gelu(a) * b to half of the strip.
#include "kernels.hpp"
#include <type_traits>

namespace rst {

typedef __attribute__((address_space(1))) const void* pw_gptr;
typedef __attribute__((address_space(3))) void* pw_lptr;
typedef float pw_f32x2 __attribute__((ext_vector_type(2)));

constexpr int PW_T = 4;                  // MFMA tiles per strip row
constexpr int PW_WT = 16 * PW_T;         // strip width
constexpr int PW_CPR = 20;               // 16-byte pieces per channel row in LDS: 4 px halo + 64 + 4 px halo, padded
constexpr int PW_ROWF = PW_CPR * 4;      // == 16 (mod 64): the four k-rows of an operand read hit disjoint banks
// CK = channels per ring stage (one barrier per CK channels = CK MFMAs per wave): 48 for K = 96 (the
// chunk boundary -- barrier, DMA issue, refilling the operand pipeline -- costs ~300 cycles), 24 for
// K = 48, where smaller stages let more of the small blocks share a CU.
constexpr int pw_ninstr(int ck) { return (ck * PW_CPR + 63) / 64; }   // wave-wide DMA instructions per chunk (the last may be half used)
constexpr int pw_xch(int ck) { return pw_ninstr(ck) * 256; }          // floats per ring stage (>= ck * PW_ROWF)
constexpr int pw_stages(int ck) { return ck >= 48 ? 4 : 5; }          // ring depth
constexpr int pw_ck(int K) { return K >= 96 ? 48 : 24; }

template <int I, int N, typename F>
__device__ __forceinline__ void pw_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pw_static_for<I + 1, N>(f);
    }
}
constexpr int PW_NSG = PW_ROWF / 16;     // groups of 16 ring columns for the statistics (5)

// QLDS: the running sum of the row below (see header) lives in LDS instead of registers.
// PIPE: the depthwise phase of row r-1 is interleaved with the MFMAs of row r (see epi_piece).
template <int KS, int NW, bool GATE, bool QLDS, bool PIPE, int PW_CK>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu((GATE || KS > 12) ? 2 : 3, (GATE || KS > 12) ? 2 : 3)))
pwdw_kernel(PwDw a, int nstrips, int nsplit, int G, int ncols) {
    constexpr int PW_NINSTR = pw_ninstr(PW_CK), PW_XCH = pw_xch(PW_CK), PW_STAGES = pw_stages(PW_CK);
    constexpr int T = PW_T, WT = PW_WT, ROWF = PW_ROWF, XCH = PW_XCH;
    constexpr int P = (PW_NINSTR + NW - 1) / NW;     // DMA instructions per wave per chunk (surplus ones land in a dummy area)
    constexpr int CKS = PW_CK / 4;                   // k-steps per chunk
    constexpr int NC = KS / CKS;                     // chunks per row
    static_assert(KS % CKS == 0, "K must be a multiple of the chunk depth");
    static_assert(PW_XCH >= PW_CK * PW_ROWF, "ring stage too small");
    constexpr int NTW = (PW_NSG + NW - 1) / NW;      // statistics groups per wave
    // SFETCH: the LayerNorm statistics read their ring columns inside the hand-scheduled operand fetch (no LDS
    // round trip of their own, no row-end barrier); it costs ~12 registers, which the 48-channel plain kernels
    // (three waves per SIMD) do not have -- they take the statistics after each chunk instead.
    constexpr bool SFETCH = PIPE && (GATE || KS > 12);
    constexpr int NTF = SFETCH ? NTW : 0;            // statistics reads per fetch
    constexpr int PAIRS = NW / 2, HT = T / 2;
    extern __shared__ __attribute__((aligned(16))) float pw_smem[];
    float* ring = pw_smem;                               // PW_STAGES stages + 256 floats for surplus DMA pieces
    float* statm = ring + PW_STAGES * XCH + 256;                 // [2][80] minus mean (or 0) per ring column, by row parity
    float* statr = statm + 2 * ROWF;                     // [2][80] 1/sqrt(var + eps)
    float* xch = statr + 2 * ROWF;                       // GATE: [PAIRS][2][HT][64] float4 hand-over area
    float* qbuf = xch + (GATE ? PAIRS * 2 * HT * 256 : 0);   // QLDS: [NW][T][64] float4
    float* wlds = qbuf + (QLDS ? NW * T * 256 : 0);      // [NW][KS][64] weights as MFMA B operand

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
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

    // ---- this lane's channel: plain: ch of M;  gate: channel ch of half `half` (row half*hidden + ch) ----
    const int half = GATE ? wave / PAIRS : 0;
    const int ch = (GATE ? (g * PAIRS + wave % PAIRS) * 16 : (g * NW + wave) * 16) + lm;
    const int chlim = GATE ? a.hidden : a.M;
    const bool chok = ch < chlim;
    const int wrow = (GATE ? half * a.hidden : 0) + (chok ? ch : 0);

    // weights of the channel as MFMA B operand: lane (n = channel, k = kk), k-step s at mywl[s * 64]
    float* mywl = wlds + wave * KS * 64 + lane;
    {
        const float* Ar = a.A + (int64_t)wrow * a.K;
#pragma unroll
        for (int s = 0; s < KS; ++s) mywl[s * 64] = chok ? Ar[4 * s + kk] : 0.f;
    }
    // (0, 1) = "no LayerNorm" in both statistics buffers: final when ln_mode == 0, and a defined value for the
    // depthwise pieces that ride on the first row of the block (they work on a row that does not exist)
    for (int i = tid; i < 2 * ROWF; i += NW * 64) { statm[i] = 0.f; statr[i] = 1.f; }
    float wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = chok ? a.dw_w[(int64_t)wrow * 9 + t] : 0.f;
    const float lns = (chok && a.ln_s) ? a.ln_s[wrow] : 0.f;
    const float lnc = (chok && a.ln_c) ? a.ln_c[wrow] : 0.f;
    const float bq = (chok && a.dw_b) ? a.dw_b[wrow] : 0.f;

    // ---- DMA pieces of this wave: ring column c holds pixel px0 - 4 + c ----
    const char* Xb = reinterpret_cast<const char*>(a.X + (int64_t)b * a.x_bs);
    unsigned srcoff[P];                                  // bytes from the chunk's first channel row
    int dstoff[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int i = wave + p * NW;
        if (i < PW_NINSTR) {
            const int e = 64 * i + lane;
            const int k = e / PW_CPR < PW_CK ? e / PW_CPR : PW_CK - 1, cc = e % PW_CPR;      // tail of the last instruction: any valid address
            const int px = px0 - 4 + 4 * cc;
            srcoff[p] = 4u * ((unsigned)(k * HW) + ((px >= 0 && px < W) ? px : 0));     // out-of-image pieces read a valid address and are masked later
            dstoff[p] = 256 * i;
        } else {
            srcoff[p] = 0;
            dstoff[p] = -1;
        }
    }
    // chunk q of the block's stream: row ra + q / NC, channels (q % NC) * PW_CK ..; lands in ring stage q % PW_STAGES.
    // The stream position is kept as running scalars (source address, channel chunk, stage, chunks left):
    // an issue is then a handful of scalar instructions -- every instruction of this loop is paid in full,
    // the wave's own issue rate is what bounds the kernel.  Past the end the last chunk is issued again
    // into its own stage (identical data), which keeps the per-wave count of outstanding pieces fixed.
    const int nq = (rb - ra + 1) * NC;
    int dsto[P];                                         // wave-uniform: LDS float offset inside a stage, < 0: surplus piece
#pragma unroll
    for (int p = 0; p < P; ++p) dsto[p] = __builtin_amdgcn_readfirstlane(dstoff[p]);
    const char* isrc = Xb + 4 * (int64_t)ra * W;
    int ic = 0, istage = 0, ileft = nq;
    const int64_t step_c = 4 * (int64_t)PW_CK * HW, step_r = 4 * (int64_t)W - (NC - 1) * step_c;
    auto issue = [&]() __attribute__((always_inline)) {
        const char* src = isrc;
        float* sbase = ring + istage * XCH;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float* dst = dsto[p] >= 0 ? sbase + dsto[p] : ring + PW_STAGES * XCH;
            __builtin_amdgcn_global_load_lds((pw_gptr)(src + srcoff[p]), (pw_lptr)dst, 16, 0, 0);
        }
        if (ileft > 1) {
            const bool wrap = ic == NC - 1;
            isrc += wrap ? step_r : step_c;
            ic = wrap ? 0 : ic + 1;
            istage = istage == PW_STAGES - 1 ? 0 : istage + 1;
            --ileft;
        }
    };

    // accumulator layout (operands swapped: pixels are the MFMA rows): register q of tile j is pixel
    // px0 + 16*j + 4*kk + q of this lane's channel
    const f32x4 bq4 = {bq, bq, bq, bq};
    constexpr int QN = QLDS ? 1 : T;
    f32x4 Pst[T], Qst[QN], acc[T];
    f32x4* myq = reinterpret_cast<f32x4*>(qbuf) + (QLDS ? wave * T * 64 + lane : 0);     // + j * 64
#pragma unroll
    for (int j = 0; j < T; ++j) {
        Pst[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (QLDS) myq[j * 64] = bq4; else Qst[j] = bq4;
    }
    constexpr int HN = GATE ? HT : 1;
    f32x4 hold0 = {0.f, 0.f, 0.f, 0.f}, hold1 = {0.f, 0.f, 0.f, 0.f};     // my half of the strip (HN <= 2 tiles), named so they stay in registers
    static_assert(HN <= 2, "two hand-over tiles per wave");
    int pend_row = -1;
    f32x4* myx = reinterpret_cast<f32x4*>(xch) + (GATE ? ((wave % PAIRS) * 2 + half) * HT * 64 + lane : 0);        // handed over by this wave
    const f32x4* ptx = reinterpret_cast<const f32x4*>(xch) + (GATE ? ((wave % PAIRS) * 2 + (1 - half)) * HT * 64 + lane : 0);

    // finished output tile `j` of row `row`: 4 consecutive pixels of one channel per lane
    // Stores go through a buffer descriptor over the 16 channel planes of this wave's slab: a lane that
    // must not store (channel beyond the last, pixel beyond the row, row outside the block's range)
    // gets an offset outside the descriptor and the hardware drops the store -- no branch, so the
    // depthwise pieces stay in the same basic block as the MFMAs they hide behind.
    typedef unsigned pw_u32x4 __attribute__((ext_vector_type(4)));
    const int slab0 = __builtin_amdgcn_readfirstlane(ch - lm);
    const int slab_n = chlim - slab0 < 0 ? 0 : (chlim - slab0 > 16 ? 16 : chlim - slab0);
    float* slab_base = a.Y + (int64_t)b * a.y_bs + (int64_t)slab0 * HW;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        slab_base, 0, (int)__builtin_amdgcn_readfirstlane((int)((int64_t)slab_n * HW * 4)), 0x00020000);
    constexpr unsigned PW_NOSTORE = 0x80000000u;
    const unsigned yoff = (unsigned)(((int64_t)lm * HW + px0 + 4 * kk) * 4);
    auto store_tile = [&](const f32x4& v, int row, int j, bool rowok) __attribute__((always_inline)) {
        const int px = px0 + 16 * j + 4 * kk;
        // (the row goes into the vector offset: with a scalar-register offset the 16-byte store still reads
        // its data registers while the next vector instruction may already overwrite them, and the compiler
        // does not guard that case)
        const unsigned vo = (rowok && px < W) ? yoff + (unsigned)(row * W * 4) + 64u * j : PW_NOSTORE;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pw_u32x4, v), yrsrc, vo, 0, 0);
    };
    auto stash = [&](const f32x4& v, int j) __attribute__((always_inline)) {           // GATE: keep my half of the strip, hand the other half over
        const bool mine = (j < HT) == (half == 0);
        if (mine) { if (j % HN == 0) hold0 = v; else hold1 = v; }
        else myx[(j % HN) * 64] = v;
    };
    auto finalize_tile = [&](int row, int jh) __attribute__((always_inline)) {         // GATE: gelu(first half) * second half on one of my tiles
        const f32x4 other = ptx[jh * 64];
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float hv = jh == 0 ? hold0[q] : hold1[q];
            const float x1 = half == 0 ? hv : other[q];
            const float x2 = half == 0 ? other[q] : hv;
            o[q] = gelu_erf(x1) * x2;
        }
        store_tile(o, row >= 0 ? row : 0, half * HT + jh, row >= 0);
    };
    auto finalize = [&](int row) __attribute__((always_inline)) {
#pragma unroll
        for (int jh = 0; jh < HN; ++jh) finalize_tile(row, jh);
    };
    // byte addresses (for ds_bpermute) of the lanes that hold the previous / next 4 pixels of the channel
    const int nbr_next = ((lane + 16) & 63) * 4, nbr_prev = ((lane + 48) & 63) * 4;
    const bool in_l = px0 - 1 >= 0, in_r = px0 + WT < W;

#pragma unroll
    for (int i = 0; i < PW_STAGES - 1; ++i) issue();
    int stage_c = 0;                             // chunk being consumed and its stage

    // ---- the depthwise phase of a finished row, cut into pieces ----
    // tp[] = the raw 1x1 accumulators of row `prow`, hlp/hrp its raw halo shares; their LayerNorm
    // constants sit in stat buffer prow & 1.  With PIPE the pieces are placed between the MFMAs of the
    // NEXT row (one piece after a k-step, spread over the row), so that the vector ALU works in the
    // shadow of the matrix pipe; without PIPE they run back to back right after the row.
    f32x4 tp[T];
    float hlp = 0.f, hrp = 0.f, hlf = 0.f, hrf = 0.f;
    int prow = -1;
    auto piece_halo = [&]() __attribute__((always_inline)) {
        hlp += __shfl_xor(hlp, 16); hrp += __shfl_xor(hrp, 16);
        hlp += __shfl_xor(hlp, 32); hrp += __shfl_xor(hrp, 32);
        const float* sm = statm + (prow & 1) * ROWF;
        const float* sr = statr + (prow & 1) * ROWF;
        const float tl = sm[3], rl = sr[3], tr = sm[4 + WT], rr = sr[4 + WT];     // (0, 1) without LayerNorm
        hlf = (in_l && prow >= 0) ? fmaf(rl, fmaf(tl, lns, hlp), lnc) : 0.f;
        hrf = (in_r && prow >= 0) ? fmaf(rr, fmaf(tr, lns, hrp), lnc) : 0.f;
    };
    // 1x1 output of a tile: LayerNorm constants and bias, then zero outside the image
    auto piece_finish = [&](int j) __attribute__((always_inline)) {
        const int px = px0 + 16 * j + 4 * kk;
        // without LayerNorm the buffers hold (0, 1): the expression is then tp + lnc exactly
        const f32x4 tco = *reinterpret_cast<const f32x4*>(&statm[(prow & 1) * ROWF + 4 + 16 * j + 4 * kk]);
        const f32x4 rstd = *reinterpret_cast<const f32x4*>(&statr[(prow & 1) * ROWF + 4 + 16 * j + 4 * kk]);
        const bool live = px < W && prow >= 0;        // W % 4 == 0: the four pixels are inside or outside together
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v = fmaf(rstd[q], fmaf(tco[q], lns, tp[j][q]), lnc);
            tp[j][q] = live ? v : 0.f;
        }
        asm volatile("" : "+v"(tp[j]));               // computed here, not sunk to where it is used
    };
    // depthwise 3x3 of tile j: row prow's share of output rows prow-1 (now complete), prow and prow+1,
    // in four sub-pieces small enough to hide between two k-steps: (a) the six pixels e[0..5] (the one
    // before the tile's four, the four, the one after), (b) the finished row, (c), (d) the two running sums
    float e[6];
    auto dw_a = [&](int j) __attribute__((always_inline)) {
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
    };
    auto dw_b = [&](int j) __attribute__((always_inline)) {
        f32x4 done;
#pragma unroll
        for (int q = 0; q < 4; ++q) done[q] = fmaf(wv[8], e[q + 2], fmaf(wv[7], e[q + 1], fmaf(wv[6], e[q], Pst[j][q])));
        if constexpr (GATE) stash(done, j);           // rows outside the range are dropped when they are finalized
        else store_tile(done, prow >= 1 ? prow - 1 : 0, j, prow - 1 >= R0 && prow - 1 < R1);
    };
    auto dw_c = [&](int j) __attribute__((always_inline)) {
        f32x4 qold;
        if constexpr (QLDS) qold = myq[j * 64]; else qold = Qst[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) Pst[j][q] = fmaf(wv[5], e[q + 2], fmaf(wv[4], e[q + 1], fmaf(wv[3], e[q], qold[q])));
        asm volatile("" : "+v"(Pst[j]));
    };
    auto dw_d = [&](int j) __attribute__((always_inline)) {
        f32x4 qn;
#pragma unroll
        for (int q = 0; q < 4; ++q) qn[q] = fmaf(wv[2], e[q + 2], fmaf(wv[1], e[q + 1], fmaf(wv[0], e[q], bq)));
        if constexpr (QLDS) myq[j * 64] = qn; else { Qst[j] = qn; asm volatile("" : "+v"(Qst[j])); }
    };
    auto piece_dw = [&](int j) __attribute__((always_inline)) { dw_a(j); dw_b(j); dw_c(j); dw_d(j); };
    // order: a tile is finished before the depthwise of its left neighbour needs its first pixel
    auto run_piece = [&](int p) __attribute__((always_inline)) {
        switch (p) {
            case 0: piece_halo(); piece_finish(0); break;
            case 1: piece_finish(1); break;
            case 2: piece_dw(0); break;
            case 3: piece_finish(2); break;
            case 4: piece_dw(1); break;
            case 5: piece_finish(3); break;
            case 6: piece_dw(2); break;
            case 7: piece_dw(3); break;
            default: break;
        }
    };
    constexpr int NPIECE = 8;
    // Interleaving: k-step `idx` of a row (0 .. KS-1) carries the pieces whose slot it is.  A k-step
    // is 4 MFMAs = 128 matrix-pipe cycles, of which about 96 are free for other vector instructions,
    // so the depthwise phase is cut into 20 pieces of 10-20 instructions and spread over the row.
    // GATE: the hand-over area is read by the finalize pieces (previous output row) and written by
    // the depthwise pieces; a chunk barrier has to lie between the two in both directions, so the
    // finalize pieces and the tile finishing take the first half of the row's k-steps and the
    // depthwise pieces the second half (every row has an even number of chunks >= 2).
#define PW_AT(SLOT, CALL) if (idx == (SLOT)) { CALL; }
    auto epi_piece = [&](int idx) __attribute__((always_inline)) {
        if constexpr (GATE) {
            static_assert(!PIPE || (NC % 2 == 0 && HN == 2), "the split of the pieces assumes two halves of whole chunks");
            constexpr int HALF = KS / 2;
            PW_AT(0 * HALF / 6, finalize_tile(pend_row, 0))
            PW_AT(1 * HALF / 6, finalize_tile(pend_row, 1))
            PW_AT(2 * HALF / 6, piece_halo(); piece_finish(0))
            PW_AT(3 * HALF / 6, piece_finish(1))
            PW_AT(4 * HALF / 6, piece_finish(2))
            PW_AT(5 * HALF / 6, piece_finish(3))
            PW_AT(HALF + 0 * HALF / 16, dw_a(0))  PW_AT(HALF + 1 * HALF / 16, dw_b(0))
            PW_AT(HALF + 2 * HALF / 16, dw_c(0))  PW_AT(HALF + 3 * HALF / 16, dw_d(0))
            PW_AT(HALF + 4 * HALF / 16, dw_a(1))  PW_AT(HALF + 5 * HALF / 16, dw_b(1))
            PW_AT(HALF + 6 * HALF / 16, dw_c(1))  PW_AT(HALF + 7 * HALF / 16, dw_d(1))
            PW_AT(HALF + 8 * HALF / 16, dw_a(2))  PW_AT(HALF + 9 * HALF / 16, dw_b(2))
            PW_AT(HALF + 10 * HALF / 16, dw_c(2)) PW_AT(HALF + 11 * HALF / 16, dw_d(2))
            PW_AT(HALF + 12 * HALF / 16, dw_a(3)) PW_AT(HALF + 13 * HALF / 16, dw_b(3))
            PW_AT(HALF + 14 * HALF / 16, dw_c(3)) PW_AT(HALF + 15 * HALF / 16, dw_d(3))
        } else {
            constexpr int NP = 20;
            PW_AT(0 * KS / NP, piece_halo(); piece_finish(0))
            PW_AT(1 * KS / NP, piece_finish(1))
            PW_AT(2 * KS / NP, dw_a(0))   PW_AT(3 * KS / NP, dw_b(0))   PW_AT(4 * KS / NP, dw_c(0))   PW_AT(5 * KS / NP, dw_d(0))
            PW_AT(6 * KS / NP, piece_finish(2))
            PW_AT(7 * KS / NP, dw_a(1))   PW_AT(8 * KS / NP, dw_b(1))   PW_AT(9 * KS / NP, dw_c(1))   PW_AT(10 * KS / NP, dw_d(1))
            PW_AT(11 * KS / NP, piece_finish(3))
            PW_AT(12 * KS / NP, dw_a(2))  PW_AT(13 * KS / NP, dw_b(2))  PW_AT(14 * KS / NP, dw_c(2))  PW_AT(15 * KS / NP, dw_d(2))
            PW_AT(16 * KS / NP, dw_a(3))  PW_AT(17 * KS / NP, dw_b(3))  PW_AT(18 * KS / NP, dw_c(3))  PW_AT(19 * KS / NP, dw_d(3))
        }
    };
#undef PW_AT

    for (int r = ra; r <= rb; ++r) {
#pragma unroll
        for (int j = 0; j < T; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        float hl = 0.f, hr = 0.f;                        // halo columns: this lane's k share of the dot products
        pw_f32x2 st[NTW];
#pragma unroll
        for (int jj = 0; jj < NTW; ++jj) st[jj] = pw_f32x2{0.f, 0.f};

#pragma unroll
        for (int c = 0; c < NC; ++c) {
            // chunk qc has landed for this wave (the next chunk's P pieces may still be in flight), then
            // for everyone; the barrier also frees the stage that is refilled next
            // (lgkmcnt: LDS writes of the previous phase -- the GATE hand-over -- are visible after the barrier)
            // (PW_STAGES - 2) newer chunks of P pieces each may still be in flight
            constexpr int NEWER = (PW_STAGES - 2) * P;      // pieces of the (PW_STAGES - 2) newer chunks
            if constexpr (NEWER == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if constexpr (NEWER == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if constexpr (NEWER == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if constexpr (NEWER == 9) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if constexpr (NEWER == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else static_assert(NEWER == 3, "add the wait count for this configuration");
            issue();
            if constexpr (GATE && !PIPE) {
                if (c == 0 && pend_row >= 0) { finalize(pend_row); pend_row = -1; }
            }
            const float* Xc = ring + stage_c * XCH + kk * ROWF;
            // Operands of a k-step: the four tile columns, the two halo columns and the weight = four
            // LDS reads.  They are issued by hand two steps (8 MFMAs) ahead of their use: left to the
            // compiler, every read is sunk next to its MFMA and the matrix pipe waits for each LDS
            // round trip.  LDS reads return in order, so "at most 8 newer reads outstanding" means the
            // reads of the step being multiplied have landed; the wait names the operand registers
            // so that the MFMAs cannot be scheduled above it.
            const unsigned tb = (unsigned)(uintptr_t)(pw_lptr)(Xc + lm);       // tile columns of this lane
            const unsigned hb = (unsigned)(uintptr_t)(pw_lptr)Xc;              // halo columns (same for 16 lanes)
            const unsigned wb = (unsigned)(uintptr_t)(pw_lptr)(mywl + c * CKS * 64);
            pw_f32x2 x01[3], x23[3], xh[3];
            float wq[3], xs[3][NTW];
            unsigned sb[NTW];
#pragma unroll
            for (int jj = 0; jj < NTW; ++jj) {
                const int sg = wave + jj * NW;
                sb[jj] = tb + 64u * (sg < PW_NSG ? sg : PW_NSG - 1);      // waves beyond the last group read along, their sums are dropped
            }
            auto fetch = [&](auto Sc) {
                constexpr int S = decltype(Sc)::value, SET = S % 3;
                const unsigned tbs = tb + S * 4 * ROWF * 4, hbs = hb + S * 4 * ROWF * 4, wbs = wb;
                // (asm operands do not capture implicitly: bind the captured registers to locals first)
                pw_f32x2 &r01 = x01[SET], &r23 = x23[SET], &rh = xh[SET];
                float& rw = wq[SET];
                asm volatile("ds_read2_b32 %0, %1 offset0:4 offset1:20" : "=v"(r01) : "v"(tbs));
                asm volatile("ds_read2_b32 %0, %1 offset0:36 offset1:52" : "=v"(r23) : "v"(tbs));
                asm volatile("ds_read2_b32 %0, %1 offset0:3 offset1:68" : "=v"(rh) : "v"(hbs));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(rw) : "v"(wbs), "n"(S * 256));
                // this wave's share of the LayerNorm statistics: one ring column per lane and k-step
                pw_static_for<0, NTF>([&](auto Jc) {
                    constexpr int JJ = decltype(Jc)::value;
                    float& rs = xs[SET][JJ];
                    const unsigned sbj = sb[JJ];
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(rs) : "v"(sbj), "n"(S * 4 * ROWF * 4));
                });
            };
            static_assert(T == 4 && WT == 64 && CKS >= 3, "the hand-scheduled k loop is written for 4 tiles");
            fetch(std::integral_constant<int, 0>{});
            fetch(std::integral_constant<int, 1>{});
            pw_static_for<0, CKS>([&](auto Sc) {
                constexpr int S = decltype(Sc)::value, SET = S % 3;
                if constexpr (S + 2 < CKS) fetch(std::integral_constant<int, S + 2>{});
                // reads of the (up to two) later steps may still be outstanding: 4 instructions each
                pw_f32x2 &r01 = x01[SET], &r23 = x23[SET], &rh = xh[SET];
                float& rw = wq[SET];
                constexpr int LATER = (S + 2 < CKS ? 2 : S + 1 < CKS ? 1 : 0) * (4 + NTF);      // reads of the later steps
                static_assert(LATER <= 15, "lgkmcnt is a 4-bit counter");
                // (every register is named once: a register tied twice is copied BEFORE the wait, i.e. before its read has landed)
                float& rs0 = xs[SET][0];
                static_assert(NTW <= 2, "name every statistics register in the wait");
                if constexpr (NTF == 0) {
                    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r01), "+v"(r23), "+v"(rh), "+v"(rw) : "n"(LATER));
                } else if constexpr (NTF == 1) {
                    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(r01), "+v"(r23), "+v"(rh), "+v"(rw), "+v"(rs0) : "n"(LATER));
                } else {
                    float& rs1 = xs[SET][NTW - 1];
                    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(r01), "+v"(r23), "+v"(rh), "+v"(rw), "+v"(rs0), "+v"(rs1) : "n"(LATER));
                }
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x01[SET][0], wq[SET], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x01[SET][1], wq[SET], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x23[SET][0], wq[SET], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x23[SET][1], wq[SET], acc[3], 0, 0, 0);
                hl = fmaf(wq[SET], xh[SET][0], hl);
                hr = fmaf(wq[SET], xh[SET][1], hr);
#pragma unroll
                for (int jj = 0; jj < NTF; ++jj) {
                    st[jj][0] += xs[SET][jj];
                    st[jj][1] = fmaf(xs[SET][jj], xs[SET][jj], st[jj][1]);
                }
                if constexpr (PIPE) epi_piece(c * CKS + S);
            });
            if constexpr (!SFETCH) {
                if (a.ln_mode != 0) {
#pragma unroll
                    for (int jj = 0; jj < NTW; ++jj) {
                        const int sg = wave + jj * NW;
                        if (sg < PW_NSG) {
#pragma unroll
                            for (int s = 0; s < CKS; ++s) {
                                const float x = Xc[4 * s * ROWF + 16 * sg + lm];
                                st[jj][0] += x;
                                st[jj][1] = fmaf(x, x, st[jj][1]);
                            }
                        }
                    }
                }
            }
            stage_c = stage_c == PW_STAGES - 1 ? 0 : stage_c + 1;
        }

        // ---- per-pixel LayerNorm constants of the row, shared through LDS (buffer r & 1) ----
        if (a.ln_mode != 0) {
#pragma unroll
            for (int jj = 0; jj < NTW; ++jj) {
                const int sg = wave + jj * NW;
                if (sg < PW_NSG) {
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
            // PIPE: the constants are first read by the pieces that ride on the next row, behind that row's
            // first chunk barrier -- no barrier of their own
            if constexpr (!SFETCH) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if constexpr (GATE && PIPE) {
            // the pieces that rode on this row finished output row prow-1 (handed over, final after the next barrier)
            pend_row = (prow >= 0 && prow - 1 >= R0 && prow - 1 < R1) ? prow - 1 : -1;
        }
        // hand the row over to the depthwise phase
#pragma unroll
        for (int j = 0; j < T; ++j) tp[j] = acc[j];
        hlp = hl; hrp = hr;
        prow = r;
        if constexpr (!PIPE) {
#pragma unroll
            for (int p = 0; p < NPIECE; ++p) run_piece(p);
            if constexpr (GATE) { if (prow - 1 >= R0 && prow - 1 < R1) pend_row = prow - 1; }
        }
    }
    if constexpr (PIPE) {
        // the last row's depthwise phase has no next row to hide behind
        if constexpr (GATE) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (pend_row >= 0) finalize(pend_row);
            pend_row = -1;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // partners are done reading before the next hand-over
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the last row's LayerNorm constants
        }
        if (prow >= 0) {
#pragma unroll
            for (int p = 0; p < NPIECE; ++p) run_piece(p);
            if constexpr (GATE) { if (prow - 1 >= R0 && prow - 1 < R1) pend_row = prow - 1; }
        }
    }

    // all DMA (including the surplus re-issues) must have landed before the block's LDS is released
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

// ---- host side ----
struct PwPlan { int ks, nw, G, nstrips, nsplit, ncols; bool ok; };

static PwPlan pwdw_plan(const PwDw& a) {
    PwPlan p{};
    p.ok = false;
    if (a.K != 48 && a.K != 96) return p;           // compiled channel depths
    if (a.W % 4 != 0 || a.W < 64 || a.H < 1) return p;
    if (!aligned16(a.X) || a.x_bs % 4 != 0 || !aligned16(a.Y) || a.y_bs % 4 != 0) return p;
    if ((int64_t)a.K * a.H * a.W >= (1ll << 30)) return p;          // byte offsets inside a sample stay below 2^32
    if (a.gate && a.M != 2 * a.hidden) return p;
    p.ks = a.K / 4;
    p.nstrips = (int)cdiv(a.W, PW_WT);
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

bool pwdw_supported(const PwDw& a) { return pwdw_plan(a).ok; }

// Measured on MI355X (16 x C x 256 x 256, one box, fused vs the product + depthwise pair): 48 channels
// 405 vs 530 us (qkv) and 390 vs 420 us (gated); 96 channels 1130 vs 1210 us and 935 vs 980 us; at
// 128 x 128 and 96 channels 320 vs 340 us and 245 vs 270 us.  Fused wherever it is supported.
bool pwdw_profitable(const PwDw& a) { return pwdw_plan(a).ok; }

int pwdw_launch(const PwDw& a, hipStream_t stream) {
    // The product runs on the f16 matrix instructions with split operands (fp32 accuracy) unless RST_F32_MFMA=1.  Two
    // forms of that: wave-specialised (pwdw_ws, 96 input channels: 685 / 503 us plain / gated at 16 x 96 x 256 x 256)
    // and single-role (pwdw_f16: 705 / 537 us there, but ahead at 48 channels).  RST_PWDW_FORM=f / w forces one (tests).
    static const char* form_env = getenv("RST_PWDW_FORM");
    const bool f32_mfma = !gemm_f16s_enabled();
    const bool single = form_env != nullptr ? form_env[0] == 'f' : a.K == 48;
    if (!f32_mfma && single && pwdw_f16_supported(a)) return pwdw_f16_launch(a, stream);
    if (pwdw_ws_supported(a)) return pwdw_ws_launch(a, stream);
    PwPlan p = pwdw_plan(a);
    RST_REQUIRE(p.ok, "pwdw: unsupported shape (K=%d W=%d)", a.K, a.W);
    RST_REQUIRE(a.X && a.A && a.dw_w && a.Y && a.B > 0 && a.M > 0, "pwdw: bad operand");
    RST_REQUIRE(a.ln_mode == 0 || (a.ln_s && a.ln_c), "pwdw: folded LayerNorm without its constants");

    const void* kern = nullptr;
#define PW_KS(NWV, GT, PP) do { if (p.ks == 12) kern = reinterpret_cast<const void*>(&pwdw_kernel<12, NWV, GT, false, PP, pw_ck(48)>); \
                                else kern = reinterpret_cast<const void*>(&pwdw_kernel<24, NWV, GT, false, PP, pw_ck(96)>); } while (0)
    if (a.gate) PW_KS(8, true, true);
    else if (p.nw == 6) PW_KS(6, false, true);
    else PW_KS(3, false, true);
#undef PW_KS
    const size_t lds = sizeof(float) * (size_t)(pw_stages(pw_ck(a.K)) * pw_xch(pw_ck(a.K)) + 256 + 4 * PW_ROWF +
                                                (a.gate ? (p.nw / 2) * 2 * (PW_T / 2) * 256 : 0) + p.nw * p.ks * 64 + 256);
    KernelSetup su;
    RST_TRY(kernel_setup(kern, p.nw * 64, lds, &su));
    const int cus = su.cus, per_cu = su.per_cu;
    // Split the rows of every (sample, strip, channel group) column so that the grid comes in whole
    // rounds of resident blocks; every row range recomputes two halo rows, so weigh that against
    // the fill of the last round.
    const int64_t cols0 = (int64_t)a.B * p.nstrips;
    const int64_t slots = (int64_t)cus * per_cu;
    int nsplit = 1;
    double best = -1.0;
    for (int ns = 1; ns <= (a.H >= 16 ? a.H / 8 : 1); ++ns) {
        const int64_t blocks = cols0 * p.G * ns;
        const int64_t rounds = cdiv(blocks, slots);
        const double rows = (double)a.H / ns;
        const double eff = (double)blocks / (double)(rounds * slots) * rows / (rows + 2.0);
        if (eff > best + 1e-9) { best = eff; nsplit = ns; }
    }
    p.nsplit = nsplit;
    const int64_t ncols = cols0 * nsplit;
    RST_REQUIRE(ncols * p.G < (1ll << 30), "pwdw: grid too large");
    p.ncols = (int)ncols;
    const int64_t cols8 = cdiv(ncols, 8) * 8;
    const unsigned grid = (unsigned)(cols8 * p.G);

    const double N = (double)a.B * a.H * a.W;
    ProfScope prof(stream, a.gate ? "pwdw_gate" : "pwdw", 2.0 * a.M * a.K * N + 18.0 * a.M * N,
                   4.0 * N * (a.K + (a.gate ? a.hidden : a.M)));
    void* args[] = {(void*)&a, (void*)&p.nstrips, (void*)&p.nsplit, (void*)&p.G, (void*)&p.ncols};
    RST_HIP(hipLaunchKernel(kern, dim3(grid), dim3(p.nw * 64), args, lds, stream));
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst
