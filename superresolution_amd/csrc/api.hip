// C ABI of the library (declared in include/restormer_hip.h): argument checks,
// workspace carving and the launch sequences of the composite operators.
#include "kernels.hpp"

namespace rst {

std::string& last_error_slot() {
    static thread_local std::string s;
    return s;
}

// Bump allocator over a caller-provided buffer.  With base == nullptr it only
// measures, so the same carving code serves the *_bytes queries.
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base(static_cast<char*>(b)) {}
    float* floats(size_t n) {
        off = (off + 255) & ~size_t(255);
        float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += n * sizeof(float);
        return p;
    }
    void* bytes(size_t n) {
        off = (off + 255) & ~size_t(255);
        void* p = base ? static_cast<void*>(base + off) : nullptr;
        off += n;
        return p;
    }
    size_t total() const { return (off + 255) & ~size_t(255); }
};

static inline size_t max2(size_t a, size_t b) { return a > b ? a : b; }

static int check_desc(const rst_block_desc* d) {
    RST_REQUIRE(d != nullptr, "null block descriptor");
    RST_REQUIRE(d->B > 0 && d->C > 0 && d->H > 0 && d->W > 0, "block: empty shape B=%d C=%d H=%d W=%d", d->B, d->C, d->H, d->W);
    RST_REQUIRE(d->heads > 0 && d->C % d->heads == 0, "block: C=%d not divisible by heads=%d", d->C, d->heads);
    RST_REQUIRE(d->hidden > 0, "block: hidden=%d", d->hidden);
    RST_REQUIRE(d->B <= 65535 / d->heads, "block: batch*heads exceeds 65535");
    return RST_OK;
}

// ---- plain 1x1 convolution helpers ---------------------------------------------
static GemmNN conv1x1_args(const float* w, int Cout, int Cin, bool transposed, const float* x, float* y, int B, int64_t N) {
    GemmNN g{};
    g.A = w; g.a_s1 = 0; g.a_s2 = 0; g.transA = transposed ? 1 : 0;
    // stored weight is [Cout][Cin]; transposed product uses it as [K=Cout][M=Cin]
    g.lda = Cin;
    g.M = transposed ? Cin : Cout;
    g.K = transposed ? Cout : Cin;
    g.X = x; g.x_s1 = (int64_t)g.K * N; g.x_s2 = 0; g.ldx = N;
    g.Y = y; g.y_s1 = (int64_t)g.M * N; g.y_s2 = 0; g.ldy = N;
    g.R = nullptr; g.rscale = nullptr; g.bias = nullptr;
    g.N = N; g.nb1 = B; g.nb2 = 1; g.accumulate = 0;
    return g;
}

// A channel LayerNorm that feeds a 1x1 convolution.  When the product kernels can fold it
// (gemm_nn_ln_supported) the normalised tensor is never written: the weights are rescaled into
// `fold` (Cout*Cin + 2*Cout floats) and the statistics are taken inside the product.  Otherwise
// the input is normalised into `scratch` first.
struct LnIn {
    const float* gamma;
    const float* beta;      // null: bias-free variant
    float* scratch;         // B*C*N floats
    const float* ready;     // folded weights prepared ahead (rst_block_prepare) in the form fold_kind() names, or null
};
// (also holds the split-f16 operand fragments of the folded weights: rows padded to 16, columns to 32)
static size_t ln_fold_floats(int Cout, int Cin) { return (size_t)round_up(Cout, 16) * round_up(Cin, 32) + 2 * (size_t)round_up(Cout, 16); }

// Products with at least this many input channels run on the f16 matrix instructions with split operands
// (gemm_f16s.hip: 159 -> 98 us / 151 -> 77 us for the LayerNorm-folded products at 192 / 384 channels).  At 96 channels
// the exact-fp32 kernel, whose matrix pipe is half busy there, used to win (277 us against 334 at 256 x 256); since
// two blocks of the split-f16 kernel share a compute unit it is the other way round (the residual products of a
// 16 x 3 x 256 x 256 forward pass: 8.52 -> 8.09 ms per step with the bound at 96, 8.24 at 48).  RST_F16S_MINK
// overrides (tests, A/B runs).
static int f16s_min_k() {
    static const int v = getenv("RST_F16S_MINK") != nullptr ? atoi(getenv("RST_F16S_MINK")) : 96;
    return v;
}
static bool use_f16s(const GemmNN& g) { return gemm_f16s_enabled() && g.K >= f16s_min_k() && gemm_f16s_supported(g); }
// which form a product with Cin input channels wants its folded weights in: 1 = split-f16 operand fragments, 0 = fp32 matrix
static int fold_kind(int Cin) { return gemm_f16s_enabled() && Cin >= f16s_min_k() ? 1 : 0; }
static int fold_prepare(int kind, const float* w, const float* gamma, const float* beta, const float* bias, int Cout, int Cin,
                        float* fold, hipStream_t stream) {
    if (kind == 1) {
        float* s = fold + (size_t)round_up(Cout, 16) * round_up(Cin, 32);
        return gemm_f16s_prep_launch(w, 0, 1, Cin, 0, gamma, beta, bias, Cout, Cin, fold, 0, s, s + round_up(Cout, 16), stream);
    }
    float* s = fold + (size_t)Cout * Cin;
    return ln_fold_weights_launch(w, gamma, beta, bias, Cout, Cin, fold, s, s + Cout, stream);
}

// y = conv1x1(LN(x)) (+bias), through the folded path when possible
static int ln_conv1x1(const LnIn& ln, const float* w, const float* bias, int Cout, int Cin, const float* x, float* y,
                      int B, int64_t N, float* fold, hipStream_t stream) {
    GemmNN g = conv1x1_args(w, Cout, Cin, false, x, y, B, N);
    if (Cin % 2 == 0 && use_f16s(g)) {
        const bool have = ln.ready != nullptr && fold_kind(Cin) == 1;
        if (!have) RST_TRY(fold_prepare(1, w, ln.gamma, ln.beta, bias, Cout, Cin, fold, stream));
        const float* f = have ? ln.ready : fold;
        const float* s = f + (size_t)round_up(Cout, 16) * round_up(Cin, 32);
        g.afrag = f; g.ln_mode = ln.beta ? 1 : 2; g.ln_s = s; g.ln_c = s + round_up(Cout, 16);
        return gemm_nn_launch(g, stream);
    }
    GemmNN probe = g;
    probe.A = fold;
    if (gemm_nn_ln_supported(probe)) {
        const bool have = ln.ready != nullptr && fold_kind(Cin) == 0;
        if (!have) RST_TRY(fold_prepare(0, w, ln.gamma, ln.beta, bias, Cout, Cin, fold, stream));
        const float* f = have ? ln.ready : fold;
        const float* s = f + (size_t)Cout * Cin;
        g.A = f; g.ln_mode = ln.beta ? 1 : 2; g.ln_s = s; g.ln_c = s + Cout;
        return gemm_nn_launch(g, stream);
    }
    RST_TRY(layernorm_fwd_launch(x, ln.gamma, ln.beta, ln.scratch, B, Cin, N, stream));
    g.X = ln.scratch;
    g.bias = bias;
    return gemm_nn_launch(g, stream);
}

// y = dw3x3(conv1x1(LN(x)))  or, with `hidden` > 0, gelu(d[:hidden]) * d[hidden:] of that -- one fused
// kernel when the shape allows (inference only: nothing is kept for a backward pass).
// Returns true in *done when the fused kernel ran.
static int ln_pw_dw(const LnIn* ln, const float* w, const float* bias, const float* dw_w, const float* dw_b,
                    int Cout, int Cin, int hidden, const float* x, float* y, int B, int H, int W, float* fold,
                    hipStream_t stream, bool* done) {
    *done = false;
    const int64_t N = (int64_t)H * W;
    PwDw a{};
    a.X = x; a.x_bs = (int64_t)Cin * N;
    a.dw_w = dw_w; a.dw_b = dw_b;
    a.Y = y; a.y_bs = (int64_t)(hidden > 0 ? hidden : Cout) * N;
    a.B = B; a.K = Cin; a.M = Cout; a.H = H; a.W = W;
    a.gate = hidden > 0 ? 1 : 0; a.hidden = hidden;
    a.A = w; a.ln_mode = 0; a.ln_s = nullptr; a.ln_c = bias;
    // RST_FUSE_PWDW=0 turns the fused kernel off (the product + depthwise pair runs instead): A/B runs, tests
    static const bool off = getenv("RST_FUSE_PWDW") != nullptr && getenv("RST_FUSE_PWDW")[0] == '0';
    if (off || !pwdw_profitable(a)) return RST_OK;
    if (ln) {
        const bool have = ln->ready != nullptr && fold_kind(Cin) == 0;
        if (!have) RST_TRY(fold_prepare(0, w, ln->gamma, ln->beta, bias, Cout, Cin, fold, stream));
        const float* f = have ? ln->ready : fold;
        const float* s = f + (size_t)Cout * Cin;
        a.A = f; a.ln_mode = ln->beta ? 1 : 2; a.ln_s = s; a.ln_c = s + Cout;
    }
    RST_TRY(pwdw_launch(a, stream));
    *done = true;
    return RST_OK;
}

// dW[M1][M2] = sum_b A[b] Bm[b]^T through split-K partials in `ws`
static size_t wgrad_ws_floats(int M1, int M2, int64_t N, int B) {
    const NtPlan p = gemm_nt_plan(M1, M2, N, B);
    return gemm_nt_partial_floats(p, M1, M2, B);
}
static int wgrad(const float* a, int64_t a_bs, const float* bm, int64_t b_bs, int M1, int M2, int64_t N, int B,
                 float* out, float* partial, hipStream_t stream) {
    const NtPlan p = gemm_nt_plan(M1, M2, N, B);
    GemmNT g{};
    g.A = a; g.a_s1 = a_bs; g.a_s2 = 0; g.lda = N;
    g.Bm = bm; g.b_s1 = b_bs; g.b_s2 = 0; g.ldb = N;
    g.M1 = M1; g.M2 = M2; g.N = N; g.nb1 = B; g.nb2 = 1;
    g.partial = partial; g.sqA_partial = nullptr; g.sqB_partial = nullptr;
    RST_TRY(gemm_nt_launch(g, p, stream));
    return reduce_slabs_launch(partial, out, (int64_t)M1 * M2, 1, B * p.splits, 0, stream);
}

// ================================ attention =======================================
struct AttnBufs {
    // kept for the backward pass (in `saved`) or scratch (in the workspace)
    float *t, *qkv, *attn, *ghat, *nq, *nk, *mfold;
    // scratch
    float *gram_part, *sqq_part, *sqk_part, *fold;
    float *gram_red, *sqq_red, *sqk_red;     // pre-reduced slabs when the pixel range is split more than 16 ways
    void* mfrag;                             // split-f16 operand fragments of mfold (one matrix per sample)
    NtPlan plan;
};

static void attn_carve_saved(const rst_block_desc* d, Carver& cv, AttnBufs& b) {
    const int64_t N = (int64_t)d->H * d->W;
    const int c = d->C / d->heads;
    b.t = cv.floats((size_t)d->B * 3 * d->C * N);
    b.qkv = cv.floats((size_t)d->B * 3 * d->C * N);
    b.attn = cv.floats((size_t)d->B * d->heads * c * c);
    b.ghat = cv.floats((size_t)d->B * d->heads * c * c);
    b.nq = cv.floats((size_t)d->B * d->C);
    b.nk = cv.floats((size_t)d->B * d->C);
    b.mfold = cv.floats((size_t)d->B * d->C * d->C);
}
static void attn_carve_ws(const rst_block_desc* d, bool have_saved, Carver& cv, AttnBufs& b) {
    const int64_t N = (int64_t)d->H * d->W;
    const int c = d->C / d->heads;
    const int BH = d->B * d->heads;
    // Up to 16 pixel splits the softmax kernel adds the Gram slabs itself.  A few large samples (one
    // full frame: BH = 1, N = 8.3 M) need far more splits than that to fill the chip; their slabs are
    // summed by a reduction pass first.
    {
        const NtPlan probe = gemm_nt_plan(c, c, N, BH, 16);
        const int64_t blocks16 = (int64_t)probe.d1.blocks * probe.d2.blocks * BH * probe.splits;
        b.plan = blocks16 >= 256 ? probe : gemm_nt_plan(c, c, N, BH, 0);
    }
    b.gram_part = cv.floats(gemm_nt_partial_floats(b.plan, c, c, BH));
    b.sqq_part = cv.floats((size_t)BH * b.plan.splits * c);
    b.sqk_part = cv.floats((size_t)BH * b.plan.splits * c);
    if (b.plan.splits > 16) {
        b.gram_red = cv.floats((size_t)BH * c * c);
        b.sqq_red = cv.floats((size_t)BH * c);
        b.sqk_red = cv.floats((size_t)BH * c);
    } else {
        b.gram_red = b.sqq_red = b.sqk_red = nullptr;
    }
    b.fold = cv.floats(ln_fold_floats(3 * d->C, d->C));
    b.mfrag = cv.bytes((size_t)d->B * gemm_f16s_frag_bytes(d->C, d->C));
    if (!have_saved) {
        b.t = cv.floats((size_t)d->B * 3 * d->C * N);
        b.qkv = cv.floats((size_t)d->B * 3 * d->C * N);
        b.attn = cv.floats((size_t)BH * c * c);
        b.mfold = cv.floats((size_t)d->B * d->C * d->C);
        b.ghat = nullptr; b.nq = nullptr; b.nk = nullptr;
    }
}

static size_t attention_ws_bytes(const rst_block_desc* d) {
    Carver cv(nullptr); AttnBufs b{};
    attn_carve_ws(d, false, cv, b);
    return cv.total();
}
static size_t attention_saved_bytes(const rst_block_desc* d) {
    Carver cv(nullptr); AttnBufs b{};
    attn_carve_saved(d, cv, b);
    return cv.total();
}

static int attention_qkv(const rst_block_desc* d, const rst_block_params* p, const float* x, AttnBufs& b, bool keep_t,
                         const LnIn* ln, hipStream_t stream);
static int attention_tail(const rst_block_desc* d, const rst_block_params* p, AttnBufs& b, const float* residual, float* y,
                          int row0, int row1, const float* stats, hipStream_t stream);

// `ln` (optional): x is the un-normalised input and the LayerNorm in front of the qkv convolution is applied here
static int attention_fwd(const rst_block_desc* d, const rst_block_params* p, const float* x, const float* residual,
                         float* y, void* saved, void* ws, size_t ws_bytes, hipStream_t stream, const LnIn* ln = nullptr) {
    RST_REQUIRE(x && y && p, "attention_fwd: null operand");
    RST_REQUIRE(p->temperature && p->qkv_weight && p->qkv_dw_weight && p->proj_weight, "attention_fwd: null parameter");
    RST_REQUIRE(!d->conv_bias || (p->qkv_bias && p->qkv_dw_bias && p->proj_bias), "attention_fwd: conv_bias set but a bias is null");
    if (ws == nullptr || ws_bytes < attention_ws_bytes(d))
        return fail(RST_ERR_WORKSPACE, "attention_fwd: workspace %zu < %zu bytes", ws_bytes, attention_ws_bytes(d));
    const int B = d->B, C = d->C, heads = d->heads, c = C / heads;
    const int64_t N = (int64_t)d->H * d->W;
    AttnBufs b{};
    if (saved) { Carver cs(saved); attn_carve_saved(d, cs, b); }
    Carver cw(ws);
    attn_carve_ws(d, saved != nullptr, cw, b);

    RST_TRY(attention_qkv(d, p, x, b, saved != nullptr, ln, stream));
    return attention_tail(d, p, b, residual, y, 0, d->H, nullptr, stream);
}

// t = W_qkv x (+bias);  qkv = dw3x3(t)
static int attention_qkv(const rst_block_desc* d, const rst_block_params* p, const float* x, AttnBufs& b, bool keep_t,
                         const LnIn* ln, hipStream_t stream) {
    const int B = d->B, C = d->C;
    const int64_t N = (int64_t)d->H * d->W;
    bool fused = false;
    if (!keep_t)
        RST_TRY(ln_pw_dw(ln, p->qkv_weight, d->conv_bias ? p->qkv_bias : nullptr, p->qkv_dw_weight,
                         d->conv_bias ? p->qkv_dw_bias : nullptr, 3 * C, C, 0, x, b.qkv, B, d->H, d->W, b.fold, stream, &fused));
    if (!fused) {
        if (ln) {
            RST_TRY(ln_conv1x1(*ln, p->qkv_weight, d->conv_bias ? p->qkv_bias : nullptr, 3 * C, C, x, b.t, B, N, b.fold, stream));
        } else {
            GemmNN g1 = conv1x1_args(p->qkv_weight, 3 * C, C, false, x, b.t, B, N);
            g1.bias = d->conv_bias ? p->qkv_bias : nullptr;
            RST_TRY(gemm_nn_launch(g1, stream));
        }
        DwArgs dw{};
        dw.x = b.t; dw.x_bstride = (int64_t)3 * C * N; dw.w = p->qkv_dw_weight;
        dw.bias = d->conv_bias ? p->qkv_dw_bias : nullptr;
        dw.y = b.qkv; dw.y_bstride = (int64_t)3 * C * N; dw.B = B; dw.C = 3 * C; dw.H = d->H; dw.W = d->W; dw.flip = 0;
        RST_TRY(dwconv_fwd_launch(dw, stream));
    }
    return RST_OK;
}

// Gram matrices and squared row norms over the rows [row0, row1) of every sample, split over the pixels
static int attention_gram(const rst_block_desc* d, const AttnBufs& b, const NtPlan& plan, int row0, int row1,
                          float* gram_part, float* sqq_part, float* sqk_part, hipStream_t stream) {
    const int B = d->B, C = d->C, heads = d->heads, c = C / heads;
    const int64_t N = (int64_t)d->H * d->W, off = (int64_t)row0 * d->W;
    GemmNT gt{};
    gt.A = b.qkv + off;                      gt.a_s1 = (int64_t)3 * C * N; gt.a_s2 = (int64_t)c * N; gt.lda = N;
    gt.Bm = b.qkv + (int64_t)C * N + off;    gt.b_s1 = (int64_t)3 * C * N; gt.b_s2 = (int64_t)c * N; gt.ldb = N;
    gt.M1 = c; gt.M2 = c; gt.N = (int64_t)(row1 - row0) * d->W; gt.nb1 = B; gt.nb2 = heads;
    gt.partial = gram_part; gt.sqA_partial = sqq_part; gt.sqB_partial = sqk_part;
    return gemm_nt_launch(gt, plan, stream);
}

// everything after the qkv tensor: Gram (unless `stats` brings the sums: [B*heads*c*c] Gram, [B*C] |q|^2, [B*C] |k|^2),
// softmax, project_out folded into attn.V, residual
static int attention_tail(const rst_block_desc* d, const rst_block_params* p, AttnBufs& b, const float* residual, float* y,
                          int row0, int row1, const float* stats, hipStream_t stream) {
    const int B = d->B, C = d->C, heads = d->heads, c = C / heads;
    const int64_t N = (int64_t)d->H * d->W;
    if (stats) {
        const float* sq = stats + (size_t)B * heads * c * c;
        RST_TRY(attn_softmax_launch(stats, sq, sq + (size_t)B * C, p->temperature, b.attn, b.ghat, b.nq, b.nk,
                                    B, heads, c, 1, stream));
    } else {
    RST_TRY(attention_gram(d, b, b.plan, row0, row1, b.gram_part, b.sqq_part, b.sqk_part, stream));
    if (b.plan.splits > 16) {
        RST_TRY(reduce_slabs3_launch(b.gram_part, b.gram_red, (int64_t)c * c, b.sqq_part, b.sqq_red, c, b.sqk_part, b.sqk_red, c, B * heads, b.plan.splits, stream));
        RST_TRY(attn_softmax_launch(b.gram_red, b.sqq_red, b.sqk_red, p->temperature, b.attn, b.ghat, b.nq, b.nk,
                                    B, heads, c, 1, stream));
    } else {
        RST_TRY(attn_softmax_launch(b.gram_part, b.sqq_part, b.sqk_part, p->temperature, b.attn, b.ghat, b.nq, b.nk,
                                    B, heads, c, b.plan.splits, stream));
    }
    }
    // y = Mfold[b] v (+bias) (+residual)
    GemmNN g2{};
    g2.A = b.mfold; g2.a_s1 = (int64_t)C * C; g2.a_s2 = 0; g2.lda = C; g2.transA = 0;
    g2.X = b.qkv + (int64_t)2 * C * N; g2.x_s1 = (int64_t)3 * C * N; g2.x_s2 = 0; g2.ldx = N;
    g2.Y = y; g2.y_s1 = (int64_t)C * N; g2.y_s2 = 0; g2.ldy = N;
    g2.R = residual; g2.r_s1 = (int64_t)C * N; g2.r_s2 = 0; g2.ldr = N; g2.rscale = nullptr;
    g2.bias = d->conv_bias ? p->proj_bias : nullptr;
    g2.M = C; g2.K = C; g2.N = N; g2.nb1 = B; g2.nb2 = 1; g2.accumulate = 0;
    if (use_f16s(g2)) {
        // (the fold kernel writes the split-f16 operand fragments of Mfold itself: no preparation launch per block)
        const int64_t fb = (int64_t)gemm_f16s_frag_bytes(C, C);
        RST_TRY(attn_fold_f16s_launch(p->proj_weight, b.attn, b.mfold, b.mfrag, fb, B, C, heads, c, stream));
        g2.afrag = b.mfrag; g2.afrag_s1 = fb; g2.afrag_s2 = 0;
    } else {
        RST_TRY(attn_fold_launch(p->proj_weight, b.attn, b.mfold, B, C, heads, c, stream));
    }
    return gemm_nn_launch(g2, stream);
}

// ---- attention backward --------------------------------------------------------
// With o = Mfold v (+bias) and dy the gradient of o:
//   P[b]   = dy[b] v[b]^T                    (C x C, split-K over the pixels)
//   dWo    = sum_b P[b] blockdiag(A[b])^T ;  dA[b,h] = (Wo^T P[b])_hh
//   softmax / temperature / normalisation on c x c matrices (attn_small.hip)
//   dq_h = Eq k_h + Dq o q_h ;  dk_h = Eq^T q_h + Dk o k_h ;  dv = Mfold^T dy
//   then the depthwise and the 1x1 convolution backwards.
struct AttnBwdBufs {
    float *P, *P_part, *Eq, *Dq, *Dk, *dGG, *dtemp_row, *dtemp_b, *dwo_b, *dqkv, *dt, *wg_part, *dwws, *rsws;
    NtPlan p_plan;
    size_t dwws_bytes, rsws_bytes;
};
static void attn_bwd_carve(const rst_block_desc* d, Carver& cv, AttnBwdBufs& b) {
    const int B = d->B, C = d->C, heads = d->heads, c = C / heads;
    const int64_t N = (int64_t)d->H * d->W;
    b.p_plan = gemm_nt_plan(C, C, N, B);
    b.P = cv.floats((size_t)B * C * C);
    b.P_part = cv.floats(gemm_nt_partial_floats(b.p_plan, C, C, B));
    b.Eq = cv.floats((size_t)B * heads * c * c);
    b.Dq = cv.floats((size_t)B * C);
    b.Dk = cv.floats((size_t)B * C);
    b.dGG = cv.floats((size_t)B * heads * c * c);
    b.dtemp_row = cv.floats((size_t)B * C);
    b.dtemp_b = cv.floats((size_t)B * heads);
    b.dwo_b = cv.floats((size_t)B * C * C);
    b.dqkv = cv.floats((size_t)B * 3 * C * N);
    b.dt = cv.floats((size_t)B * 3 * C * N);
    b.wg_part = cv.floats(wgrad_ws_floats(3 * C, C, N, B));
    b.dwws_bytes = dwconv_bwd_weight_ws_bytes(B, 3 * C, d->H, d->W);
    b.dwws = static_cast<float*>(cv.bytes(b.dwws_bytes));
    b.rsws_bytes = rowsum_ws_bytes(B, 3 * C, N);
    b.rsws = static_cast<float*>(cv.bytes(b.rsws_bytes));
}
static size_t attention_bwd_ws_bytes(const rst_block_desc* d) {
    Carver cv(nullptr); AttnBwdBufs b{};
    attn_bwd_carve(d, cv, b);
    return cv.total();
}

static int attention_bwd(const rst_block_desc* d, const rst_block_params* p, const float* x, const float* dy,
                         const void* saved, float* dx, int accumulate_dx, rst_block_grads* gr,
                         void* ws, size_t ws_bytes, hipStream_t stream) {
    RST_REQUIRE(x && dy && saved && dx && gr && p, "attention_bwd: null operand");
    RST_REQUIRE(gr->temperature && gr->qkv_weight && gr->qkv_dw_weight && gr->proj_weight, "attention_bwd: null gradient buffer");
    RST_REQUIRE(!d->conv_bias || (gr->qkv_bias && gr->qkv_dw_bias && gr->proj_bias), "attention_bwd: null bias gradient buffer");
    if (ws == nullptr || ws_bytes < attention_bwd_ws_bytes(d))
        return fail(RST_ERR_WORKSPACE, "attention_bwd: workspace %zu < %zu bytes", ws_bytes, attention_bwd_ws_bytes(d));
    const int B = d->B, C = d->C, heads = d->heads, c = C / heads;
    const int64_t N = (int64_t)d->H * d->W;
    AttnBufs s{};
    { Carver cs(const_cast<void*>(saved)); attn_carve_saved(d, cs, s); }
    AttnBwdBufs b{};
    { Carver cw(ws); attn_bwd_carve(d, cw, b); }
    const float* q = s.qkv;
    const float* k = s.qkv + (int64_t)C * N;
    const float* v = s.qkv + (int64_t)2 * C * N;
    const int64_t qs = (int64_t)3 * C * N;

    // P[b] = dy[b] v[b]^T
    {
        GemmNT g{};
        g.A = dy; g.a_s1 = (int64_t)C * N; g.a_s2 = 0; g.lda = N;
        g.Bm = v; g.b_s1 = qs; g.b_s2 = 0; g.ldb = N;
        g.M1 = C; g.M2 = C; g.N = N; g.nb1 = B; g.nb2 = 1;
        g.partial = b.P_part;
        RST_TRY(gemm_nt_launch(g, b.p_plan, stream));
        RST_TRY(reduce_slabs_launch(b.P_part, b.P, (int64_t)C * C, B, b.p_plan.splits, 0, stream));
    }
    if (d->conv_bias) RST_TRY(rowsum_launch(dy, (int64_t)C * N, gr->proj_bias, B, C, N, b.rsws, b.rsws_bytes, stream));
    RST_TRY(attn_bwd_small_launch(b.P, p->proj_weight, s.attn, s.ghat, s.nq, s.nk, p->temperature, b.Eq, b.Dq, b.Dk,
                                  b.dGG, b.dtemp_row, b.dtemp_b, B, C, heads, c, stream));
    RST_TRY(attn_bwd_wo_launch(b.P, s.attn, b.dwo_b, B, C, heads, c, stream));
    RST_TRY(reduce_slabs_launch(b.dwo_b, gr->proj_weight, (int64_t)C * C, 1, B, 0, stream));
    RST_TRY(reduce_slabs_launch(b.dtemp_b, gr->temperature, heads, 1, B, 0, stream));

    // dq, dk per (sample, head); dv per sample
    {
        GemmNN g{};
        g.A = b.Eq; g.a_s1 = (int64_t)heads * c * c; g.a_s2 = (int64_t)c * c; g.lda = c; g.transA = 0;
        g.X = k; g.x_s1 = qs; g.x_s2 = (int64_t)c * N; g.ldx = N;
        g.Y = b.dqkv; g.y_s1 = qs; g.y_s2 = (int64_t)c * N; g.ldy = N;
        g.R = q; g.r_s1 = qs; g.r_s2 = (int64_t)c * N; g.ldr = N;
        g.rscale = b.Dq; g.rs_s1 = C; g.rs_s2 = c;
        g.bias = nullptr; g.M = c; g.K = c; g.N = N; g.nb1 = B; g.nb2 = heads; g.accumulate = 0;
        RST_TRY(gemm_nn_launch(g, stream));
        g.transA = 1;
        g.X = q; g.Y = b.dqkv + (int64_t)C * N; g.R = k; g.rscale = b.Dk;
        RST_TRY(gemm_nn_launch(g, stream));
    }
    {
        GemmNN g{};
        g.A = s.mfold; g.a_s1 = (int64_t)C * C; g.a_s2 = 0; g.lda = C; g.transA = 1;
        g.X = dy; g.x_s1 = (int64_t)C * N; g.x_s2 = 0; g.ldx = N;
        g.Y = b.dqkv + (int64_t)2 * C * N; g.y_s1 = qs; g.y_s2 = 0; g.ldy = N;
        g.R = nullptr; g.rscale = nullptr; g.bias = nullptr;
        g.M = C; g.K = C; g.N = N; g.nb1 = B; g.nb2 = 1; g.accumulate = 0;
        RST_TRY(gemm_nn_launch(g, stream));
    }
    // depthwise conv backward
    RST_TRY(dwconv_bwd_weight_launch(s.t, qs, b.dqkv, qs, gr->qkv_dw_weight, d->conv_bias ? gr->qkv_dw_bias : nullptr,
                                     B, 3 * C, d->H, d->W, b.dwws, b.dwws_bytes, stream));
    {
        DwArgs dw{};
        dw.x = b.dqkv; dw.x_bstride = qs; dw.w = p->qkv_dw_weight; dw.bias = nullptr;
        dw.y = b.dt; dw.y_bstride = qs; dw.B = B; dw.C = 3 * C; dw.H = d->H; dw.W = d->W; dw.flip = 1;
        RST_TRY(dwconv_fwd_launch(dw, stream));
    }
    // 1x1 qkv conv backward
    RST_TRY(wgrad(b.dt, qs, x, (int64_t)C * N, 3 * C, C, N, B, gr->qkv_weight, b.wg_part, stream));
    if (d->conv_bias) RST_TRY(rowsum_launch(b.dt, qs, gr->qkv_bias, B, 3 * C, N, b.rsws, b.rsws_bytes, stream));
    GemmNN gx = conv1x1_args(p->qkv_weight, 3 * C, C, true, b.dt, dx, B, N);
    gx.accumulate = accumulate_dx;
    return gemm_nn_launch(gx, stream);
}

// ================================== GDFN ===========================================
struct FfnBufs { float *u, *d, *g; };
static void ffn_carve(const rst_block_desc* d, bool with_d, Carver& cv, FfnBufs& b) {
    const int64_t N = (int64_t)d->H * d->W;
    b.u = cv.floats((size_t)d->B * 2 * d->hidden * N);
    b.d = with_d ? cv.floats((size_t)d->B * 2 * d->hidden * N) : nullptr;
    b.g = cv.floats((size_t)d->B * d->hidden * N);
}
static size_t ffn_ws_bytes(const rst_block_desc* d) {
    Carver cv(nullptr); FfnBufs b{};
    ffn_carve(d, false, cv, b);
    cv.floats(ln_fold_floats(2 * d->hidden, d->C));
    cv.bytes(gemm_f16s_frag_bytes(d->C, d->hidden));
    return cv.total();
}
static size_t ffn_saved_bytes(const rst_block_desc* d) {
    Carver cv(nullptr); FfnBufs b{};
    ffn_carve(d, true, cv, b);
    return cv.total();
}

// `ln` (optional, only without `saved`): as in attention_fwd, for the LayerNorm in front of project_in
static int ffn_fwd(const rst_block_desc* d, const rst_block_params* p, const float* x, const float* residual, float* y,
                   void* saved, void* ws, size_t ws_bytes, hipStream_t stream, const LnIn* ln = nullptr,
                   const void* pout_ready = nullptr) {
    RST_REQUIRE(x && y && p, "ffn_fwd: null operand");
    RST_REQUIRE(p->pin_weight && p->ffn_dw_weight && p->pout_weight, "ffn_fwd: null parameter");
    RST_REQUIRE(!d->conv_bias || (p->pin_bias && p->ffn_dw_bias && p->pout_bias), "ffn_fwd: conv_bias set but a bias is null");
    FfnBufs b{};
    float* fold = nullptr;
    void* pfrag = nullptr;
    if (saved) {
        RST_REQUIRE(ln == nullptr, "ffn_fwd: the folded LayerNorm keeps nothing for a backward pass");
        Carver cs(saved); ffn_carve(d, true, cs, b);
    } else {
        if (ws == nullptr || ws_bytes < ffn_ws_bytes(d))
            return fail(RST_ERR_WORKSPACE, "ffn_fwd: workspace %zu < %zu bytes", ws_bytes, ffn_ws_bytes(d));
        Carver cw(ws); ffn_carve(d, false, cw, b);
        fold = cw.floats(ln_fold_floats(2 * d->hidden, d->C));
        pfrag = cw.bytes(gemm_f16s_frag_bytes(d->C, d->hidden));
    }
    const int B = d->B, C = d->C, Hd = d->hidden;
    const int64_t N = (int64_t)d->H * d->W;
    bool fused = false;
    if (!saved)
        RST_TRY(ln_pw_dw(ln, p->pin_weight, d->conv_bias ? p->pin_bias : nullptr, p->ffn_dw_weight,
                         d->conv_bias ? p->ffn_dw_bias : nullptr, 2 * Hd, C, Hd, x, b.g, B, d->H, d->W, fold, stream, &fused));
    if (!fused) {
        if (ln) {
            RST_TRY(ln_conv1x1(*ln, p->pin_weight, d->conv_bias ? p->pin_bias : nullptr, 2 * Hd, C, x, b.u, B, N, fold, stream));
        } else {
            GemmNN g1 = conv1x1_args(p->pin_weight, 2 * Hd, C, false, x, b.u, B, N);
            g1.bias = d->conv_bias ? p->pin_bias : nullptr;
            RST_TRY(gemm_nn_launch(g1, stream));
        }
        GateArgs ga{};
        ga.u = b.u; ga.u_bstride = (int64_t)2 * Hd * N; ga.w = p->ffn_dw_weight; ga.bias = d->conv_bias ? p->ffn_dw_bias : nullptr;
        ga.gout = b.g; ga.g_bstride = (int64_t)Hd * N; ga.d = b.d; ga.d_bstride = (int64_t)2 * Hd * N;
        ga.B = B; ga.hidden = Hd; ga.H = d->H; ga.W = d->W;
        RST_TRY(dwconv_gate_fwd_launch(ga, stream));
    }
    GemmNN g2 = conv1x1_args(p->pout_weight, C, Hd, false, b.g, y, B, N);
    g2.bias = d->conv_bias ? p->pout_bias : nullptr;
    g2.R = residual; g2.r_s1 = (int64_t)C * N; g2.r_s2 = 0; g2.ldr = N;
    if (pfrag != nullptr && use_f16s(g2)) {
        if (pout_ready != nullptr && fold_kind(Hd) == 1) {
            g2.afrag = pout_ready;
        } else {
            RST_TRY(gemm_f16s_prep_launch(p->pout_weight, 0, 1, Hd, 0, nullptr, nullptr, nullptr, C, Hd, pfrag, 0, nullptr, nullptr, stream));
            g2.afrag = pfrag;
        }
    }
    return gemm_nn_launch(g2, stream);
}

struct FfnBwdBufs { float *dg, *dd, *du, *wg_part, *dwws, *rsws; size_t dwws_bytes, rsws_bytes; };
static void ffn_bwd_carve(const rst_block_desc* d, Carver& cv, FfnBwdBufs& b) {
    const int B = d->B, C = d->C, Hd = d->hidden;
    const int64_t N = (int64_t)d->H * d->W;
    b.dg = cv.floats((size_t)B * Hd * N);
    b.dd = cv.floats((size_t)B * 2 * Hd * N);
    b.du = cv.floats((size_t)B * 2 * Hd * N);
    b.wg_part = cv.floats(max2(wgrad_ws_floats(C, Hd, N, B), wgrad_ws_floats(2 * Hd, C, N, B)));
    b.dwws_bytes = dwconv_bwd_weight_ws_bytes(B, 2 * Hd, d->H, d->W);
    b.dwws = static_cast<float*>(cv.bytes(b.dwws_bytes));
    b.rsws_bytes = rowsum_ws_bytes(B, max2(2 * Hd, C), N);
    b.rsws = static_cast<float*>(cv.bytes(b.rsws_bytes));
}
static size_t ffn_bwd_ws_bytes(const rst_block_desc* d) {
    Carver cv(nullptr); FfnBwdBufs b{};
    ffn_bwd_carve(d, cv, b);
    return cv.total();
}

static int ffn_bwd(const rst_block_desc* d, const rst_block_params* p, const float* x, const float* dy, const void* saved,
                   float* dx, int accumulate_dx, rst_block_grads* gr, void* ws, size_t ws_bytes, hipStream_t stream) {
    RST_REQUIRE(x && dy && saved && dx && gr && p, "ffn_bwd: null operand");
    RST_REQUIRE(gr->pin_weight && gr->ffn_dw_weight && gr->pout_weight, "ffn_bwd: null gradient buffer");
    RST_REQUIRE(!d->conv_bias || (gr->pin_bias && gr->ffn_dw_bias && gr->pout_bias), "ffn_bwd: null bias gradient buffer");
    if (ws == nullptr || ws_bytes < ffn_bwd_ws_bytes(d))
        return fail(RST_ERR_WORKSPACE, "ffn_bwd: workspace %zu < %zu bytes", ws_bytes, ffn_bwd_ws_bytes(d));
    const int B = d->B, C = d->C, Hd = d->hidden;
    const int64_t N = (int64_t)d->H * d->W;
    FfnBufs s{};
    { Carver cs(const_cast<void*>(saved)); ffn_carve(d, true, cs, s); }
    FfnBwdBufs b{};
    { Carver cw(ws); ffn_bwd_carve(d, cw, b); }

    // project_out: dW = sum dy g^T, dg = W^T dy
    RST_TRY(wgrad(dy, (int64_t)C * N, s.g, (int64_t)Hd * N, C, Hd, N, B, gr->pout_weight, b.wg_part, stream));
    if (d->conv_bias) RST_TRY(rowsum_launch(dy, (int64_t)C * N, gr->pout_bias, B, C, N, b.rsws, b.rsws_bytes, stream));
    GemmNN g1 = conv1x1_args(p->pout_weight, C, Hd, true, dy, b.dg, B, N);
    RST_TRY(gemm_nn_launch(g1, stream));
    // gate and depthwise conv
    RST_TRY(gate_bwd_launch(b.dg, s.d, b.dd, B, Hd, N, stream));
    RST_TRY(dwconv_bwd_weight_launch(s.u, (int64_t)2 * Hd * N, b.dd, (int64_t)2 * Hd * N, gr->ffn_dw_weight,
                                     d->conv_bias ? gr->ffn_dw_bias : nullptr, B, 2 * Hd, d->H, d->W, b.dwws, b.dwws_bytes, stream));
    DwArgs dw{};
    dw.x = b.dd; dw.x_bstride = (int64_t)2 * Hd * N; dw.w = p->ffn_dw_weight; dw.bias = nullptr;
    dw.y = b.du; dw.y_bstride = (int64_t)2 * Hd * N; dw.B = B; dw.C = 2 * Hd; dw.H = d->H; dw.W = d->W; dw.flip = 1;
    RST_TRY(dwconv_fwd_launch(dw, stream));
    // project_in
    RST_TRY(wgrad(b.du, (int64_t)2 * Hd * N, x, (int64_t)C * N, 2 * Hd, C, N, B, gr->pin_weight, b.wg_part, stream));
    if (d->conv_bias) RST_TRY(rowsum_launch(b.du, (int64_t)2 * Hd * N, gr->pin_bias, B, 2 * Hd, N, b.rsws, b.rsws_bytes, stream));
    GemmNN gx = conv1x1_args(p->pin_weight, 2 * Hd, C, true, b.du, dx, B, N);
    gx.accumulate = accumulate_dx;
    return gemm_nn_launch(gx, stream);
}

// ================================== block ==========================================
struct BlockSaved { float *xn1, *x1, *xn2; void *attn, *ffn; };
static void block_carve_saved(const rst_block_desc* d, Carver& cv, BlockSaved& s) {
    const size_t n = (size_t)d->B * d->C * d->H * d->W;
    s.xn1 = cv.floats(n);
    s.x1 = cv.floats(n);
    s.xn2 = cv.floats(n);
    s.attn = cv.bytes(attention_saved_bytes(d));
    s.ffn = cv.bytes(ffn_saved_bytes(d));
}
static size_t block_saved_bytes(const rst_block_desc* d) {
    Carver cv(nullptr); BlockSaved s{};
    block_carve_saved(d, cv, s);
    return cv.total();
}
static size_t block_ws_bytes(const rst_block_desc* d) {
    // xn (shared by both norms) + the larger of the two sub-operator workspaces
    Carver cv(nullptr);
    cv.floats((size_t)d->B * d->C * d->H * d->W);
    cv.bytes(max2(attention_ws_bytes(d), ffn_ws_bytes(d)));
    return cv.total();
}

// what does not depend on the input: both LayerNorm-folded weight sets and the fragments of project_out
struct BlockPrepared { float *fold_qkv, *fold_pin; void* pout_frag; };
static void block_carve_prepared(const rst_block_desc* d, Carver& cv, BlockPrepared& b) {
    b.fold_qkv = cv.floats(ln_fold_floats(3 * d->C, d->C));
    b.fold_pin = cv.floats(ln_fold_floats(2 * d->hidden, d->C));
    b.pout_frag = cv.bytes(gemm_f16s_frag_bytes(d->C, d->hidden));
}
static size_t block_prepared_bytes(const rst_block_desc* d) {
    Carver cv(nullptr); BlockPrepared b{};
    block_carve_prepared(d, cv, b);
    return cv.total();
}
static int block_prepare(const rst_block_desc* d, const rst_block_params* p, void* prepared, hipStream_t stream) {
    RST_REQUIRE(p && prepared, "block_prepare: null operand");
    RST_REQUIRE(p->norm1_weight && p->norm2_weight && p->qkv_weight && p->pin_weight && p->pout_weight, "block_prepare: null parameter");
    RST_REQUIRE(!d->ln_with_bias || (p->norm1_bias && p->norm2_bias), "block_prepare: ln_with_bias set but a bias is null");
    RST_REQUIRE(!d->conv_bias || (p->qkv_bias && p->pin_bias), "block_prepare: conv_bias set but a bias is null");
    Carver cp(prepared); BlockPrepared b{};
    block_carve_prepared(d, cp, b);
    const int C = d->C, Hd = d->hidden;
    RST_TRY(fold_prepare(fold_kind(C), p->qkv_weight, p->norm1_weight, d->ln_with_bias ? p->norm1_bias : nullptr,
                         d->conv_bias ? p->qkv_bias : nullptr, 3 * C, C, b.fold_qkv, stream));
    RST_TRY(fold_prepare(fold_kind(C), p->pin_weight, p->norm2_weight, d->ln_with_bias ? p->norm2_bias : nullptr,
                         d->conv_bias ? p->pin_bias : nullptr, 2 * Hd, C, b.fold_pin, stream));
    if (fold_kind(Hd) == 1)
        RST_TRY(gemm_f16s_prep_launch(p->pout_weight, 0, 1, Hd, 0, nullptr, nullptr, nullptr, C, Hd, b.pout_frag, 0, nullptr, nullptr, stream));
    return RST_OK;
}

static int block_fwd(const rst_block_desc* d, const rst_block_params* p, const float* x, float* y, void* saved,
                     void* ws, size_t ws_bytes, hipStream_t stream, const void* prepared = nullptr) {
    RST_REQUIRE(x && y && p, "block_fwd: null operand");
    RST_REQUIRE(p->norm1_weight && p->norm2_weight, "block_fwd: null LayerNorm weight");
    RST_REQUIRE(!d->ln_with_bias || (p->norm1_bias && p->norm2_bias), "block_fwd: ln_with_bias set but a bias is null");
    RST_REQUIRE(!(saved && x == y), "block_fwd: in-place update is only allowed without a saved buffer");
    if (ws == nullptr || ws_bytes < block_ws_bytes(d))
        return fail(RST_ERR_WORKSPACE, "block_fwd: workspace %zu < %zu bytes", ws_bytes, block_ws_bytes(d));
    const int B = d->B, C = d->C;
    const int64_t N = (int64_t)d->H * d->W;
    Carver cw(ws);
    float* xn = cw.floats((size_t)B * C * N);
    const size_t sub_bytes = max2(attention_ws_bytes(d), ffn_ws_bytes(d));
    void* sub = cw.bytes(sub_bytes);
    BlockSaved s{};
    float* xn1 = xn; float* xn2 = xn; float* x1 = y;
    if (saved) {
        Carver cs(saved); block_carve_saved(d, cs, s);
        xn1 = s.xn1; xn2 = s.xn2; x1 = s.x1;
    }
    if (!saved) {
        // inference: the normalised tensors are consumed by one product each and never stored
        BlockPrepared pre{};
        if (prepared != nullptr) { Carver cp(const_cast<void*>(prepared)); block_carve_prepared(d, cp, pre); }
        const LnIn ln1{p->norm1_weight, d->ln_with_bias ? p->norm1_bias : nullptr, xn, pre.fold_qkv};
        const LnIn ln2{p->norm2_weight, d->ln_with_bias ? p->norm2_bias : nullptr, xn, pre.fold_pin};
        RST_TRY(attention_fwd(d, p, x, x, x1, nullptr, sub, sub_bytes, stream, &ln1));
        return ffn_fwd(d, p, x1, x1, y, nullptr, sub, sub_bytes, stream, &ln2, pre.pout_frag);
    }
    RST_TRY(layernorm_fwd_launch(x, p->norm1_weight, d->ln_with_bias ? p->norm1_bias : nullptr, xn1, B, C, N, stream));
    RST_TRY(attention_fwd(d, p, xn1, x, x1, s.attn, sub, sub_bytes, stream));
    RST_TRY(layernorm_fwd_launch(x1, p->norm2_weight, d->ln_with_bias ? p->norm2_bias : nullptr, xn2, B, C, N, stream));
    return ffn_fwd(d, p, xn2, x1, y, s.ffn, sub, sub_bytes, stream);
}

// ---- one block of a row-sharded image (inference): the global sums of the attention come in from outside ----
// desc->H counts the rows of the local slab including its halo rows; [row0, row1) are the rows this device owns.
struct ShardBufs { float* xn; void* sub; size_t sub_bytes; float *gram_part, *sqq_part, *sqk_part; NtPlan plan; };

static void shard_carve(const rst_block_desc* d, int rows, Carver& cv, ShardBufs& sb) {
    const int c = d->C / d->heads, BH = d->B * d->heads;
    sb.xn = cv.floats((size_t)d->B * d->C * d->H * d->W);
    sb.sub_bytes = max2(attention_ws_bytes(d), ffn_ws_bytes(d));
    sb.sub = cv.bytes(sb.sub_bytes);
    // sized for the whole slab (the number of pixel splits grows with the pixel count), planned for the owned rows
    const NtPlan most = gemm_nt_plan(c, c, (int64_t)d->H * d->W, BH, 0);
    sb.plan = gemm_nt_plan(c, c, (int64_t)rows * d->W, BH, 0);
    sb.gram_part = cv.floats(gemm_nt_partial_floats(most, c, c, BH));
    sb.sqq_part = cv.floats((size_t)BH * most.splits * c);
    sb.sqk_part = cv.floats((size_t)BH * most.splits * c);
}
static size_t block_shard_ws_bytes(const rst_block_desc* d) {
    Carver cv(nullptr); ShardBufs sb{};
    shard_carve(d, d->H, cv, sb);
    return cv.total();
}
static size_t block_shard_stats_floats(const rst_block_desc* d) {
    const int c = d->C / d->heads;
    return (size_t)d->B * d->heads * c * c + 2 * (size_t)d->B * d->C;
}

static int block_shard_check(const rst_block_desc* d, const rst_block_params* p, const void* ws, size_t ws_bytes, const char* who) {
    RST_REQUIRE(p && p->norm1_weight && p->norm2_weight, "%s: null LayerNorm weight", who);
    RST_REQUIRE(!d->ln_with_bias || (p->norm1_bias && p->norm2_bias), "%s: ln_with_bias set but a bias is null", who);
    RST_REQUIRE(p->temperature && p->qkv_weight && p->qkv_dw_weight && p->proj_weight, "%s: null attention parameter", who);
    RST_REQUIRE(!d->conv_bias || (p->qkv_bias && p->qkv_dw_bias && p->proj_bias), "%s: conv_bias set but a bias is null", who);
    if (ws == nullptr || ws_bytes < block_shard_ws_bytes(d))
        return fail(RST_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes, block_shard_ws_bytes(d));
    return RST_OK;
}

// phase 1: qkv of the slab (kept in the workspace for phase 2), then this device's share of the Gram matrices and of
// the squared norms of q and k over its own rows -> stats[B*heads*c*c | B*C | B*C]
static int block_shard_stats(const rst_block_desc* d, const rst_block_params* p, const float* x, int row0, int row1,
                             float* stats, void* ws, size_t ws_bytes, hipStream_t stream) {
    RST_REQUIRE(x && stats, "block_shard_fwd_stats: null operand");
    RST_REQUIRE(0 <= row0 && row0 < row1 && row1 <= d->H, "block_shard_fwd_stats: rows [%d, %d) outside the slab of %d rows", row0, row1, d->H);
    RST_TRY(block_shard_check(d, p, ws, ws_bytes, "block_shard_fwd_stats"));
    const int B = d->B, C = d->C, heads = d->heads, c = C / heads;
    Carver cw(ws); ShardBufs sb{};
    shard_carve(d, row1 - row0, cw, sb);
    AttnBufs b{}; Carver ca(sb.sub);
    attn_carve_ws(d, false, ca, b);
    const LnIn ln1{p->norm1_weight, d->ln_with_bias ? p->norm1_bias : nullptr, sb.xn, nullptr};
    RST_TRY(attention_qkv(d, p, x, b, false, &ln1, stream));
    RST_TRY(attention_gram(d, b, sb.plan, row0, row1, sb.gram_part, sb.sqq_part, sb.sqk_part, stream));
    float* sq = stats + (size_t)B * heads * c * c;
    RST_TRY(reduce_slabs_launch(sb.gram_part, stats, (int64_t)c * c, B * heads, sb.plan.splits, 0, stream));
    RST_TRY(reduce_slabs_launch(sb.sqq_part, sq, c, B * heads, sb.plan.splits, 0, stream));
    return reduce_slabs_launch(sb.sqk_part, sq + (size_t)B * C, c, B * heads, sb.plan.splits, 0, stream);
}

// phase 2: `stats` summed over all devices -> attention output + residual, then the feed-forward half
static int block_shard_apply(const rst_block_desc* d, const rst_block_params* p, const float* x, const float* stats, float* y,
                             void* ws, size_t ws_bytes, hipStream_t stream) {
    RST_REQUIRE(x && stats && y, "block_shard_fwd_apply: null operand");
    RST_TRY(block_shard_check(d, p, ws, ws_bytes, "block_shard_fwd_apply"));
    Carver cw(ws); ShardBufs sb{};
    shard_carve(d, d->H, cw, sb);
    AttnBufs b{}; Carver ca(sb.sub);
    attn_carve_ws(d, false, ca, b);                       // same carving as phase 1: b.qkv still holds the slab's qkv
    RST_TRY(attention_tail(d, p, b, x, y, 0, d->H, stats, stream));
    const LnIn ln2{p->norm2_weight, d->ln_with_bias ? p->norm2_bias : nullptr, sb.xn, nullptr};
    return ffn_fwd(d, p, y, y, y, nullptr, sb.sub, sb.sub_bytes, stream, &ln2);
}

static size_t block_bwd_ws_bytes(const rst_block_desc* d) {
    Carver cv(nullptr);
    const size_t n = (size_t)d->B * d->C * d->H * d->W;
    cv.floats(n);   // dxn (shared)
    cv.floats(n);   // dx1
    cv.bytes(layernorm_bwd_ws_bytes(d->B, d->C, (int64_t)d->H * d->W));
    cv.bytes(max2(attention_bwd_ws_bytes(d), ffn_bwd_ws_bytes(d)));
    return cv.total();
}

static int block_bwd(const rst_block_desc* d, const rst_block_params* p, const float* x, const float* dy, const void* saved,
                     float* dx, rst_block_grads* gr, void* ws, size_t ws_bytes, hipStream_t stream) {
    RST_REQUIRE(x && dy && saved && dx && gr && p, "block_bwd: null operand");
    RST_REQUIRE(gr->norm1_weight && gr->norm2_weight, "block_bwd: null LayerNorm gradient buffer");
    RST_REQUIRE(!d->ln_with_bias || (gr->norm1_bias && gr->norm2_bias), "block_bwd: null LayerNorm bias gradient buffer");
    if (ws == nullptr || ws_bytes < block_bwd_ws_bytes(d))
        return fail(RST_ERR_WORKSPACE, "block_bwd: workspace %zu < %zu bytes", ws_bytes, block_bwd_ws_bytes(d));
    const int B = d->B, C = d->C;
    const int64_t N = (int64_t)d->H * d->W;
    const size_t n = (size_t)B * C * N;
    Carver cw(ws);
    float* dxn = cw.floats(n);
    float* dx1 = cw.floats(n);
    const size_t ln_bytes = layernorm_bwd_ws_bytes(B, C, N);
    void* lnws = cw.bytes(ln_bytes);
    const size_t sub_bytes = max2(attention_bwd_ws_bytes(d), ffn_bwd_ws_bytes(d));
    void* sub = cw.bytes(sub_bytes);
    BlockSaved s{};
    { Carver cs(const_cast<void*>(saved)); block_carve_saved(d, cs, s); }

    // y = x1 + ffn(LN2(x1))
    RST_TRY(ffn_bwd(d, p, s.xn2, dy, s.ffn, dxn, 0, gr, sub, sub_bytes, stream));
    RST_TRY(layernorm_bwd_launch(s.x1, p->norm2_weight, d->ln_with_bias, dxn, dy, dx1, gr->norm2_weight, gr->norm2_bias,
                                 B, C, N, lnws, ln_bytes, stream));
    // x1 = x + attn(LN1(x))
    RST_TRY(attention_bwd(d, p, s.xn1, dx1, s.attn, dxn, 0, gr, sub, sub_bytes, stream));
    return layernorm_bwd_launch(x, p->norm1_weight, d->ln_with_bias, dxn, dx1, dx, gr->norm1_weight, gr->norm1_bias,
                                B, C, N, lnws, ln_bytes, stream);
}

}  // namespace rst

// ====================================== C ABI ========================================
using namespace rst;

extern "C" {

int rst_version(void) { return 1; }
const char* rst_last_error(void) { return last_error_slot().c_str(); }

int rst_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y,
                      int32_t B, int32_t C, int64_t N, void* stream) {
    RST_REQUIRE(x && weight && y, "layernorm_fwd: null operand");
    RST_REQUIRE(B > 0 && C > 0 && N > 0, "layernorm_fwd: empty shape B=%d C=%d N=%lld", B, C, (long long)N);
    return layernorm_fwd_launch(x, weight, bias, y, B, C, N, static_cast<hipStream_t>(stream));
}
size_t rst_layernorm_bwd_workspace_bytes(int32_t B, int32_t C, int64_t N) { return layernorm_bwd_ws_bytes(B, C, N); }
int rst_layernorm_bwd(const float* x, const float* weight, int32_t with_bias, const float* dy,
                      float* dx, int32_t accumulate_dx, float* dweight, float* dbias,
                      int32_t B, int32_t C, int64_t N, void* workspace, size_t workspace_bytes, void* stream) {
    RST_REQUIRE(x && weight && dy && dx && dweight, "layernorm_bwd: null operand");
    RST_REQUIRE(!with_bias || dbias, "layernorm_bwd: with_bias set but dbias is null");
    RST_REQUIRE(B > 0 && C > 0 && N > 0, "layernorm_bwd: empty shape");
    return layernorm_bwd_launch(x, weight, with_bias, dy, accumulate_dx ? dx : nullptr, dx, dweight, dbias, B, C, N,
                                workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int rst_layernorm_bwd_noalloc(const float* x, const float* weight, int32_t with_bias, const float* dy,
                              float* dx, int32_t accumulate_dx, float* dweight, float* dbias,
                              int32_t B, int32_t C, int64_t N, void* stream) {
    RST_REQUIRE(x && weight && dy && dx && dweight, "layernorm_bwd_noalloc: null operand");
    RST_REQUIRE(!with_bias || dbias, "layernorm_bwd_noalloc: with_bias set but dbias is null");
    RST_REQUIRE(B > 0 && C > 0 && N > 0, "layernorm_bwd_noalloc: empty shape");
    return layernorm_bwd_noalloc_launch(x, weight, with_bias, dy, accumulate_dx ? dx : nullptr, dx, dweight, dbias, B, C, N,
                                        static_cast<hipStream_t>(stream));
}

int rst_conv1x1_fwd(const float* x, const float* weight, const float* bias, const float* residual, float* y,
                    int32_t B, int32_t Cin, int32_t Cout, int64_t N, void* stream) {
    RST_REQUIRE(x && weight && y, "conv1x1_fwd: null operand");
    RST_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && N > 0, "conv1x1_fwd: empty shape");
    GemmNN g = conv1x1_args(weight, Cout, Cin, false, x, y, B, N);
    g.bias = bias;
    g.R = residual; g.r_s1 = (int64_t)Cout * N; g.r_s2 = 0; g.ldr = N;
    return gemm_nn_launch(g, static_cast<hipStream_t>(stream));
}
size_t rst_conv1x1_cat_workspace_bytes(int32_t Ca, int32_t Cb, int32_t Cout) {
    return (Ca > 0 && Cb > 0 && Cout > 0) ? gemm_f16s_frag_bytes(Cout, Ca + Cb) : 0;
}
int rst_conv1x1_cat_fwd(const float* xa, const float* xb, const float* weight, const float* bias, float* y,
                        int32_t B, int32_t Ca, int32_t Cb, int32_t Cout, int64_t N,
                        void* workspace, size_t workspace_bytes, void* stream) {
    RST_REQUIRE(xa && xb && weight && y, "conv1x1_cat: null operand");
    RST_REQUIRE(B > 0 && Ca > 0 && Cb > 0 && Cout > 0 && N > 0, "conv1x1_cat: empty shape");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int K = Ca + Cb;
    GemmNN g = conv1x1_args(weight, Cout, K, false, xa, y, B, N);
    g.x_s1 = (int64_t)Ca * N;
    g.bias = bias;
    GemmNN probe = g;
    probe.X2 = xb;
    if (gemm_f16s_enabled() && Ca % 32 == 0 && gemm_f16s_supported(probe) && aligned16(xb) && ((int64_t)Cb * N) % 4 == 0) {
        // one product over both tensors: the channel chunks below Ca stream from xa, the others from xb
        if (workspace == nullptr || workspace_bytes < rst_conv1x1_cat_workspace_bytes(Ca, Cb, Cout) || !aligned16(workspace))
            return fail(RST_ERR_WORKSPACE, "conv1x1_cat: workspace %zu < %zu bytes", workspace_bytes, rst_conv1x1_cat_workspace_bytes(Ca, Cb, Cout));
        RST_TRY(gemm_f16s_prep_launch(weight, 0, 1, K, 0, nullptr, nullptr, nullptr, Cout, K, workspace, 0, nullptr, nullptr, st));
        g.afrag = workspace;
        g.X2 = xb; g.x2_s1 = (int64_t)Cb * N; g.x2_s2 = 0; g.ksplit = Ca;
        return gemm_nn_launch(g, st);
    }
    // two products into the same output: y = W[:, :Ca] xa (+bias), then y += W[:, Ca:] xb
    g.K = Ca;                                    // lda stays Ca + Cb: the first Ca columns of every weight row
    RST_TRY(gemm_nn_launch(g, st));
    GemmNN h = conv1x1_args(weight + Ca, Cout, K, false, xb, y, B, N);
    h.K = Cb; h.x_s1 = (int64_t)Cb * N; h.accumulate = 1;
    return gemm_nn_launch(h, st);
}

size_t rst_conv1x1_bwd_workspace_bytes(int32_t B, int32_t Cin, int32_t Cout, int64_t N) {
    Carver cv(nullptr);
    cv.floats(wgrad_ws_floats(Cout, Cin, N, B));
    cv.bytes(rowsum_ws_bytes(B, Cout, N));
    return cv.total();
}
int rst_conv1x1_bwd(const float* x, const float* weight, const float* dy, float* dx, int32_t accumulate_dx,
                    float* dweight, float* dbias, int32_t B, int32_t Cin, int32_t Cout, int64_t N,
                    void* workspace, size_t workspace_bytes, void* stream) {
    RST_REQUIRE(x && weight && dy && dweight, "conv1x1_bwd: null operand");
    RST_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && N > 0, "conv1x1_bwd: empty shape");
    if (workspace == nullptr || workspace_bytes < rst_conv1x1_bwd_workspace_bytes(B, Cin, Cout, N))
        return fail(RST_ERR_WORKSPACE, "conv1x1_bwd: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver cw(workspace);
    float* part = cw.floats(wgrad_ws_floats(Cout, Cin, N, B));
    const size_t rs_bytes = rowsum_ws_bytes(B, Cout, N);
    void* rs = cw.bytes(rs_bytes);
    RST_TRY(wgrad(dy, (int64_t)Cout * N, x, (int64_t)Cin * N, Cout, Cin, N, B, dweight, part, st));
    if (dbias) RST_TRY(rowsum_launch(dy, (int64_t)Cout * N, dbias, B, Cout, N, rs, rs_bytes, st));
    if (dx) {
        GemmNN g = conv1x1_args(weight, Cout, Cin, true, dy, dx, B, N);
        g.accumulate = accumulate_dx;
        RST_TRY(gemm_nn_launch(g, st));
    }
    return RST_OK;
}

int rst_dwconv3x3_fwd(const float* x, const float* weight, const float* bias, float* y,
                      int32_t B, int32_t C, int32_t H, int32_t W, void* stream) {
    DwArgs a{};
    a.x = x; a.x_bstride = (int64_t)C * H * W; a.w = weight; a.bias = bias;
    a.y = y; a.y_bstride = (int64_t)C * H * W; a.B = B; a.C = C; a.H = H; a.W = W; a.flip = 0;
    return dwconv_fwd_launch(a, static_cast<hipStream_t>(stream));
}
int rst_conv3x3_fewout_fwd(const float* x, const float* weight, const float* bias, const float* residual, float* y,
                           int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W, void* stream) {
    Conv3Args a{};
    a.x = x; a.w = weight; a.bias = bias; a.res = residual; a.y = y;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    return conv3x3_fewout_launch(a, static_cast<hipStream_t>(stream));
}
int rst_conv3x3_fewout_up2_fwd(const float* x, const float* weight, const float* bias, const float* lowres, float* y,
                               int32_t B, int32_t Cin, int32_t Cout, int32_t H, int32_t W, void* stream) {
    RST_REQUIRE(lowres != nullptr, "conv3x3_fewout_up2: null low-resolution tensor");
    Conv3Args a{};
    a.x = x; a.w = weight; a.bias = bias; a.res = nullptr; a.up_src = lowres; a.y = y;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    return conv3x3_fewout_launch(a, static_cast<hipStream_t>(stream));
}
size_t rst_dwconv3x3_bwd_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W) {
    return dwconv_bwd_weight_ws_bytes(B, C, H, W);
}
int rst_dwconv3x3_bwd(const float* x, const float* weight, const float* dy, float* dx, float* dweight, float* dbias,
                      int32_t B, int32_t C, int32_t H, int32_t W, void* workspace, size_t workspace_bytes, void* stream) {
    RST_REQUIRE(x && weight && dy && dweight, "dwconv3x3_bwd: null operand");
    RST_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dwconv3x3_bwd: empty shape");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t bs = (int64_t)C * H * W;
    RST_TRY(dwconv_bwd_weight_launch(x, bs, dy, bs, dweight, dbias, B, C, H, W, workspace, workspace_bytes, st));
    if (dx) {
        DwArgs a{};
        a.x = dy; a.x_bstride = bs; a.w = weight; a.bias = nullptr; a.y = dx; a.y_bstride = bs;
        a.B = B; a.C = C; a.H = H; a.W = W; a.flip = 1;
        RST_TRY(dwconv_fwd_launch(a, st));
    }
    return RST_OK;
}

int rst_pixel_unshuffle2(const float* x, float* y, int32_t B, int32_t C, int32_t H, int32_t W, void* stream) {
    RST_REQUIRE(H % 2 == 0 && W % 2 == 0, "pixel_unshuffle: H=%d W=%d must be even", H, W);
    return pixel_shuffle2_launch(false, x, y, B, C, H / 2, W / 2, static_cast<hipStream_t>(stream));
}
int rst_pixel_shuffle2(const float* x, float* y, int32_t B, int32_t C, int32_t H, int32_t W, void* stream) {
    return pixel_shuffle2_launch(true, x, y, B, C, H, W, static_cast<hipStream_t>(stream));
}

size_t rst_attention_workspace_bytes(const rst_block_desc* d) { return check_desc(d) == RST_OK ? attention_ws_bytes(d) : 0; }
size_t rst_attention_saved_bytes(const rst_block_desc* d) { return check_desc(d) == RST_OK ? attention_saved_bytes(d) : 0; }
int rst_attention_fwd(const rst_block_desc* d, const rst_block_params* p, const float* x, const float* residual, float* y,
                      void* saved, void* workspace, size_t workspace_bytes, void* stream) {
    RST_TRY(check_desc(d));
    return attention_fwd(d, p, x, residual, y, saved, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
size_t rst_attention_bwd_workspace_bytes(const rst_block_desc* d) { return check_desc(d) == RST_OK ? attention_bwd_ws_bytes(d) : 0; }
int rst_attention_bwd(const rst_block_desc* d, const rst_block_params* p, const float* x, const float* dy, const void* saved,
                      float* dx, int32_t accumulate_dx, rst_block_grads* grads, void* workspace, size_t workspace_bytes, void* stream) {
    RST_TRY(check_desc(d));
    return attention_bwd(d, p, x, dy, saved, dx, accumulate_dx, grads, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

size_t rst_ffn_workspace_bytes(const rst_block_desc* d) { return check_desc(d) == RST_OK ? ffn_ws_bytes(d) : 0; }
size_t rst_ffn_saved_bytes(const rst_block_desc* d) { return check_desc(d) == RST_OK ? ffn_saved_bytes(d) : 0; }
int rst_ffn_fwd(const rst_block_desc* d, const rst_block_params* p, const float* x, const float* residual, float* y,
                void* saved, void* workspace, size_t workspace_bytes, void* stream) {
    RST_TRY(check_desc(d));
    return ffn_fwd(d, p, x, residual, y, saved, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
size_t rst_ffn_bwd_workspace_bytes(const rst_block_desc* d) { return check_desc(d) == RST_OK ? ffn_bwd_ws_bytes(d) : 0; }
int rst_ffn_bwd(const rst_block_desc* d, const rst_block_params* p, const float* x, const float* dy, const void* saved,
                float* dx, int32_t accumulate_dx, rst_block_grads* grads, void* workspace, size_t workspace_bytes, void* stream) {
    RST_TRY(check_desc(d));
    return ffn_bwd(d, p, x, dy, saved, dx, accumulate_dx, grads, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

size_t rst_block_workspace_bytes(const rst_block_desc* d) { return check_desc(d) == RST_OK ? block_ws_bytes(d) : 0; }
size_t rst_block_saved_bytes(const rst_block_desc* d) { return check_desc(d) == RST_OK ? block_saved_bytes(d) : 0; }
int rst_block_fwd(const rst_block_desc* d, const rst_block_params* p, const float* x, float* y, void* saved,
                  void* workspace, size_t workspace_bytes, void* stream) {
    RST_TRY(check_desc(d));
    return block_fwd(d, p, x, y, saved, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
size_t rst_block_prepared_bytes(const rst_block_desc* d) { return check_desc(d) == RST_OK ? block_prepared_bytes(d) : 0; }
int rst_block_prepare(const rst_block_desc* d, const rst_block_params* p, void* prepared, void* stream) {
    RST_TRY(check_desc(d));
    return block_prepare(d, p, prepared, static_cast<hipStream_t>(stream));
}
int rst_block_fwd_prepared(const rst_block_desc* d, const rst_block_params* p, const void* prepared, const float* x, float* y,
                           void* workspace, size_t workspace_bytes, void* stream) {
    RST_TRY(check_desc(d));
    RST_REQUIRE(prepared != nullptr, "block_fwd_prepared: null prepared buffer");
    return block_fwd(d, p, x, y, nullptr, workspace, workspace_bytes, static_cast<hipStream_t>(stream), prepared);
}
size_t rst_block_shard_workspace_bytes(const rst_block_desc* d) { return check_desc(d) == RST_OK ? block_shard_ws_bytes(d) : 0; }
size_t rst_block_shard_stats_floats(const rst_block_desc* d) { return check_desc(d) == RST_OK ? block_shard_stats_floats(d) : 0; }
int rst_block_shard_fwd_stats(const rst_block_desc* d, const rst_block_params* p, const float* x, int32_t row0, int32_t row1,
                              float* stats, void* workspace, size_t workspace_bytes, void* stream) {
    RST_TRY(check_desc(d));
    return block_shard_stats(d, p, x, row0, row1, stats, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
int rst_block_shard_fwd_apply(const rst_block_desc* d, const rst_block_params* p, const float* x, const float* stats, float* y,
                              void* workspace, size_t workspace_bytes, void* stream) {
    RST_TRY(check_desc(d));
    return block_shard_apply(d, p, x, stats, y, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
size_t rst_block_bwd_workspace_bytes(const rst_block_desc* d) { return check_desc(d) == RST_OK ? block_bwd_ws_bytes(d) : 0; }
int rst_block_bwd(const rst_block_desc* d, const rst_block_params* p, const float* x, const float* dy, const void* saved,
                  float* dx, rst_block_grads* grads, void* workspace, size_t workspace_bytes, void* stream) {
    RST_TRY(check_desc(d));
    return block_bwd(d, p, x, dy, saved, dx, grads, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

}  // extern "C"
