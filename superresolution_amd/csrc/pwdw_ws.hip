// Host side of the wave-specialised fused 1x1 + depthwise kernels (pwdw_ws_kernel.hpp) and the plain
// instantiation.  This file is compiled WITH packing of adjacent scalar float operations into v_pk_* (the
// vector waves' depthwise arithmetic halves its instruction count, and the kernel is bound by the total
// instruction issue of a SIMD); the gated instantiation lives in pwdw_ws_gate.hip, compiled without it
// (packed temporaries push its 128-register vector waves into spilling).
#include "pwdw_ws_kernel.hpp"


namespace rst {

const void* pwdw_ws_gate_kernel(int ks, bool f16s);       // pwdw_ws_gate.hip

// ---- host side ----
bool pwdw_ws_supported(const PwDw& a) {
    static const bool off = getenv("RST_PWDW_WS") != nullptr && getenv("RST_PWDW_WS")[0] == '0';
    if (off || (a.K != 96 && a.K != 48)) return false;
    // 48 channels without gate: per row the vector waves have as much to do as at 96 channels and the matrix waves
    // half -- 517 us against 390 us for the single-role kernel at 16 x 48 x 256^2 (the gated form still gains: 291 vs 313)
    if (a.K == 48 && !a.gate) return false;
    // plain: six (96 channels) or three (48) slabs per block, no empty slab; gated: four slab pairs per block
    if (a.gate ? a.M != 2 * a.hidden : a.M % (a.K == 96 ? 96 : 48) != 0) return false;
    if (a.W % 4 != 0 || a.W < 64 || a.H < 2) return false;
    if (!aligned16(a.X) || a.x_bs % 4 != 0 || !aligned16(a.Y) || a.y_bs % 4 != 0) return false;
    if ((int64_t)a.K * a.H * a.W >= (1ll << 30)) return false;
    return true;
}

int pwdw_ws_launch(const PwDw& a, hipStream_t stream) {
    RST_REQUIRE(pwdw_ws_supported(a), "pwdw_ws: unsupported shape (K=%d M=%d W=%d)", a.K, a.M, a.W);
    RST_REQUIRE(a.X && a.A && a.dw_w && a.Y && a.B > 0, "pwdw_ws: bad operand");
    RST_REQUIRE(a.ln_mode == 0 || (a.ln_s && a.ln_c), "pwdw_ws: folded LayerNorm without its constants");
    const int KS = a.K / 4;
    const int NM = a.gate ? 8 : (a.K == 96 ? 6 : 3);
    const int nstrips = (int)cdiv(a.W, WS_WT), G = a.gate ? (int)cdiv(cdiv(a.hidden, 16), 4) : a.M / (16 * NM);
    // 96 input channels: the product runs on the f16 matrix instructions with both operands split in two (fp32
    // accuracy, see pwdw_ws_kernel.hpp); RST_F32_MFMA=1 keeps the exact-fp32 instructions (A/B runs, inputs beyond the
    // f16 range)
    const bool f32_mfma = !gemm_f16s_enabled();
    const bool f16s = KS == 24 && !f32_mfma;
    const void* kern = a.gate ? pwdw_ws_gate_kernel(KS, f16s)
                     : KS == 24 ? (f16s ? reinterpret_cast<const void*>(&pwdw_ws_kernel<24, 6, false, true>)
                                        : reinterpret_cast<const void*>(&pwdw_ws_kernel<24, 6, false>))
                                : reinterpret_cast<const void*>(&pwdw_ws_kernel<12, 3, false>);
    const size_t lds = sizeof(float) * (size_t)(ws_stages(f16s) * ws_xch(KS, f16s) + 256 + 4 * WS_ROWF + NM * WS_T * 256 + NM * 2 * 64 +
                                                (f16s ? WS_IMG_FLOATS : NM * KS * 64));
    KernelSetup su;
    RST_TRY(kernel_setup(kern, NM * 128, lds, &su));
    static std::mutex mu;
    static std::map<std::tuple<int64_t, int, int64_t>, int> nsplits;      // (columns x groups, H, slots) -> row split
    // rows of every (sample, strip, channel group) column split so that the grid comes in whole rounds of
    // resident blocks (one per CU); every row range recomputes two halo rows
    const int64_t cols0 = (int64_t)a.B * nstrips, slots = (int64_t)su.cus * su.per_cu;
    int nsplit = 1;
    {
        std::lock_guard<std::mutex> lock(mu);
        int& memo = nsplits[{cols0 * G, a.H, slots}];
        if (memo == 0) {
            double best = -1.0;
            memo = 1;
            for (int ns = 1; ns <= (a.H >= 16 ? a.H / 8 : 1); ++ns) {
                const int64_t blocks = cols0 * G * ns;
                const int64_t rounds = cdiv(blocks, slots);
                const double rows = (double)a.H / ns;
                const double eff = (double)blocks / (double)(rounds * slots) * rows / (rows + 3.0);
                if (eff > best + 1e-9) { best = eff; memo = ns; }
            }
        }
        nsplit = memo;
    }
    const int64_t ncols64 = cols0 * nsplit;
    RST_REQUIRE(ncols64 * G < (1ll << 30), "pwdw_ws: grid too large");
    const int ncols = (int)ncols64;
    const unsigned grid = (unsigned)(cdiv(ncols64, 8) * 8 * G);
    const double N = (double)a.B * a.H * a.W;
    ProfScope prof(stream, a.gate ? (f16s ? "pwdw_gate_f16s" : "pwdw_gate") : (f16s ? "pwdw_f16s" : "pwdw"), 2.0 * a.M * a.K * N + 18.0 * a.M * N,
                   4.0 * N * (a.K + (a.gate ? a.hidden : a.M)));
    void* args[] = {(void*)&a, (void*)&nstrips, (void*)&nsplit, (void*)&G, (void*)&ncols};
    RST_HIP(hipLaunchKernel(kern, dim3(grid), dim3(NM * 128), args, lds, stream));
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst
