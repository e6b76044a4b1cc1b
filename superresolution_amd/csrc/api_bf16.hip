// C ABI of the bf16 storage path (inference): channel-packed bf16 activations, fp32 parameters and sums.
// Launch sequences only; the kernels are in bf16_pk.hip and attn_small.hip.
#include "kernels.hpp"
#include <map>
#include <mutex>

namespace rst {

namespace {

struct Carve {
    char* base; size_t off = 0;
    explicit Carve(void* b) : base(static_cast<char*>(b)) {}
    void* bytes(size_t n) {
        off = (off + 255) & ~size_t(255);
        void* p = base ? static_cast<void*>(base + off) : nullptr;
        off += n;
        return p;
    }
    float* floats(size_t n) { return static_cast<float*>(bytes(n * sizeof(float))); }
    bf16_t* halves(size_t n) { return static_cast<bf16_t*>(bytes(n * sizeof(bf16_t))); }
    size_t total() const { return (off + 255) & ~size_t(255); }
};

inline int pad8(int c) { return (int)round_up(c, 8); }
constexpr int NOSEG = 1 << 30;          // "one segment": the source channels map to the positions one to one

// prepared form of one 1x1 weight matrix: operand fragments + per-position constants
struct PkWeight {
    int K, M, seg, seg_pad;      // source [M][K]; output positions in segments
    int Kb, Mb, KS, P, ngroups, ntiles, npos;
    bf16_t* frag; float* s; float* c;
    size_t frag_elems() const { return (size_t)ntiles * KS * 64 * 8; }
};
PkWeight pk_weight_shape(int M, int K, int seg, int seg_pad) {
    PkWeight w{};
    w.K = K; w.M = M; w.seg = seg; w.seg_pad = seg_pad;
    const int nseg = (int)cdiv(M, seg);
    w.Kb = (int)cdiv(K, 8);
    w.Mb = seg_pad >= NOSEG ? (int)cdiv(M, 8) : seg_pad < 0 ? 2 * pad8(seg) / 8 : nseg * seg_pad / 8;
    w.KS = gemm_pk_ks(w.Kb);
    w.P = gemm_pk_pairs(w.Kb, w.Mb);
    w.ngroups = (int)cdiv(cdiv(w.Mb, 4), w.P);
    w.ntiles = w.ngroups * w.P * 2;
    w.npos = w.ngroups * w.P * 32;
    return w;
}
void pk_weight_carve(PkWeight& w, Carve& cv, int nb) {
    w.frag = cv.halves(w.frag_elems() * nb);
    w.s = cv.floats(w.npos);
    w.c = cv.floats(w.npos);
}
// weights shared by the samples; gamma/beta: LayerNorm folded in front; bias: of the convolution
int pk_weight_prepare(const PkWeight& w, const float* src, const float* gamma, const float* beta, const float* bias,
                      hipStream_t stream) {
    RST_TRY(pk_prep_a_launch(src, 0, w.K, gamma, w.M, w.K, w.seg, w.seg_pad, w.ntiles, w.KS, w.frag, 0, 1, stream));
    return pk_prep_sc_launch(src, w.K, gamma, beta, bias, w.M, w.K, w.seg, w.seg_pad, w.npos, w.s, w.c, stream);
}

int check_pk_desc(const rst_block_desc* d) {
    RST_REQUIRE(d != nullptr, "null block descriptor");
    RST_REQUIRE(d->B > 0 && d->C > 0 && d->H > 0 && d->W > 0, "bf16 block: empty shape B=%d C=%d H=%d W=%d", d->B, d->C, d->H, d->W);
    RST_REQUIRE(d->heads > 0 && d->C % d->heads == 0, "bf16 block: C=%d not divisible by heads=%d", d->C, d->heads);
    RST_REQUIRE(d->hidden > 0, "bf16 block: hidden=%d", d->hidden);
    RST_REQUIRE(d->C % 8 == 0, "bf16 block: C=%d must be a multiple of 8", d->C);
    RST_REQUIRE(gram_pk_supported(d->C / d->heads), "bf16 block: %d channels per head not supported (multiples of 16 up to 192)", d->C / d->heads);
    RST_REQUIRE(d->W % 2 == 0, "bf16 block: W=%d must be even", d->W);
    RST_REQUIRE(gemm_pk_ks(d->C / 8) > 0 && gemm_pk_ks(pad8(d->hidden) / 8) > 0, "bf16 block: C=%d / hidden=%d too wide (at most 480 input channels per product)", d->C, d->hidden);
    return RST_OK;
}

struct PkBlockBufs {
    PkWeight wqkv, wattn, wpin, wpout;
    float *dw_qkv, *dw_ffn;
    bf16_t *buf1, *buf2;              // t / u   and   qkv / g
    float *gram_part, *sqq_part, *sqk_part, *gram_red, *sqq_red, *sqk_red, *attn, *mfold;
    int splits; int64_t chunk;
};

GemmPk pk_gemm_args(const PkWeight& w, const bf16_t* x, int64_t x_bs, bf16_t* y, int64_t y_bs, int B, int64_t N) {
    GemmPk g{};
    g.A = w.frag; g.a_bs = 0;
    g.X = x; g.x_bs = x_bs; g.Y = y; g.y_bs = y_bs; g.R = nullptr; g.r_bs = 0;
    g.ln_s = nullptr; g.ln_c = nullptr; g.ln_mode = 0;
    g.K = w.K; g.Kb = w.Kb; g.Mb = w.Mb; g.B = B; g.N = N;
    return g;
}

}  // namespace

static PkWeight pk_plain_weight(int M, int K) { return pk_weight_shape(M, K, NOSEG, NOSEG); }

// the part that depends on the parameters only (channel counts, not B, H, W): may be prepared once and kept
static void pk_block_carve_prepared(const rst_block_desc* d, Carve& cv, PkBlockBufs& b) {
    const int C = d->C, Hd = d->hidden, Hp = pad8(Hd);
    b.wqkv = pk_plain_weight(3 * C, C);
    b.wpin = pk_weight_shape(2 * Hd, C, Hd, -1);       // x1 / x2 interleaved in fours
    b.wpout = pk_plain_weight(C, Hd);
    pk_weight_carve(b.wqkv, cv, 1);
    pk_weight_carve(b.wpin, cv, 1);
    pk_weight_carve(b.wpout, cv, 1);
    b.dw_qkv = cv.floats((size_t)(3 * C / 8) * 80);
    b.dw_ffn = cv.floats((size_t)(2 * Hp / 8) * 80);
}
static size_t pk_block_prepared_bytes(const rst_block_desc* d) {
    Carve cv(nullptr); PkBlockBufs b{};
    pk_block_carve_prepared(d, cv, b);
    return cv.total();
}
static void pk_block_carve(const rst_block_desc* d, Carve& cv, PkBlockBufs& b);

static size_t pk_block_ws_bytes(const rst_block_desc* d) {
    Carve cv(nullptr); PkBlockBufs b{};
    cv.bytes(pk_block_prepared_bytes(d));
    pk_block_carve(d, cv, b);
    return cv.total();
}

static int pk_block_prepare(const rst_block_desc* d, const rst_block_params* p, const PkBlockBufs& b, hipStream_t stream) {
    RST_REQUIRE(p, "bf16 block: null parameters");
    RST_REQUIRE(p->norm1_weight && p->norm2_weight, "bf16 block: null LayerNorm weight");
    RST_REQUIRE(!d->ln_with_bias || (p->norm1_bias && p->norm2_bias), "bf16 block: ln_with_bias set but a bias is null");
    RST_REQUIRE(p->temperature && p->qkv_weight && p->qkv_dw_weight && p->proj_weight, "bf16 block: null attention parameter");
    RST_REQUIRE(p->pin_weight && p->ffn_dw_weight && p->pout_weight, "bf16 block: null feed-forward parameter");
    RST_REQUIRE(!d->conv_bias || (p->qkv_bias && p->qkv_dw_bias && p->proj_bias && p->pin_bias && p->ffn_dw_bias && p->pout_bias),
                "bf16 block: conv_bias set but a bias is null");
    const int C = d->C, Hd = d->hidden, Hp = pad8(Hd);
    const bool cb = d->conv_bias != 0;
    // all eight preparations of a block in one launch
    PkPrepJobs jobs;
    auto weight = [&](const PkWeight& w, const float* src, const float* gamma, const float* beta, const float* bias) {
        pk_prep_jobs_add_weight(jobs, src, w.K, gamma, beta, bias, w.M, w.seg, w.seg_pad, w.ntiles, w.KS, w.npos, w.frag, w.s, w.c);
    };
    weight(b.wqkv, p->qkv_weight, p->norm1_weight, d->ln_with_bias ? p->norm1_bias : nullptr, cb ? p->qkv_bias : nullptr);
    pk_prep_jobs_add_dw(jobs, p->qkv_dw_weight, cb ? p->qkv_dw_bias : nullptr, 3 * C, NOSEG, NOSEG, 3 * C / 8, b.dw_qkv);
    weight(b.wpin, p->pin_weight, p->norm2_weight, d->ln_with_bias ? p->norm2_bias : nullptr, cb ? p->pin_bias : nullptr);
    pk_prep_jobs_add_dw(jobs, p->ffn_dw_weight, cb ? p->ffn_dw_bias : nullptr, 2 * Hd, Hd, -1, 2 * Hp / 8, b.dw_ffn);
    weight(b.wpout, p->pout_weight, nullptr, nullptr, cb ? p->pout_bias : nullptr);
    return pk_prep_jobs_launch(jobs, stream);
}

static void pk_block_carve(const rst_block_desc* d, Carve& cv, PkBlockBufs& b) {
    const int C = d->C, Hd = d->hidden, Hp = pad8(Hd), heads = d->heads, c = C / heads, B = d->B;
    const int64_t N = (int64_t)d->H * d->W;
    b.wattn = pk_plain_weight(C, C);
    pk_weight_carve(b.wattn, cv, B);
    const int w1 = 3 * C > 2 * Hp ? 3 * C : 2 * Hp, w2 = 3 * C > Hp ? 3 * C : Hp;
    b.buf1 = cv.halves((size_t)B * w1 * N);
    b.buf2 = cv.halves((size_t)B * w2 * N);
    gram_pk_plan(c, N, B * heads, &b.splits, &b.chunk);
    const size_t BH = (size_t)B * heads;
    b.gram_part = cv.floats(BH * b.splits * c * c);
    b.sqq_part = cv.floats(BH * b.splits * c);
    b.sqk_part = cv.floats(BH * b.splits * c);
    b.gram_red = cv.floats(BH * c * c);
    b.sqq_red = cv.floats(BH * c);
    b.sqk_red = cv.floats(BH * c);
    b.attn = cv.floats(BH * c * c);
    b.mfold = cv.floats((size_t)B * C * C);
}

// What a training forward pass may keep of a block's attention for its backward pass: the softmax matrices, the
// normalised Gram matrices and norms they came from, and M_b = Wo blockdiag(softmax) -- a few hundred KB to a few MB per
// block, against five launches over the whole of q and k to form them again.
struct KeptAttn { float *attn, *ghat, *nq, *nk, *mfold; };
static KeptAttn kept_carve(const rst_block_desc* d, Carve& cv) {
    const size_t BH = (size_t)d->B * d->heads, c = (size_t)(d->C / d->heads);
    KeptAttn k{};
    k.attn = cv.floats(BH * c * c); k.ghat = cv.floats(BH * c * c);
    k.nq = cv.floats((size_t)d->B * d->C); k.nk = cv.floats((size_t)d->B * d->C);
    k.mfold = cv.floats((size_t)d->B * d->C * d->C);
    return k;
}
static size_t pk_block_kept_bytes(const rst_block_desc* d) {
    Carve cv(nullptr);
    kept_carve(d, cv);
    return cv.total();
}

// What a training forward pass may keep INSTEAD, when memory is no concern: the kept attention plus every intermediate
// the backward pass reads (packed bf16) -- the two 1x1 outputs t and u, their depthwise results qkv and d, x1 and the
// gated product g: 2 (7 C + 5 Hp) bytes per pixel and block, against evaluating the block again in the backward pass.
struct SavedActs { KeptAttn k; bf16_t *t, *qkv, *x1, *u, *d, *g; };
static SavedActs saved_carve(const rst_block_desc* d, Carve& cv) {
    const size_t bn = (size_t)d->B * d->H * d->W;
    const int C = d->C, Hp = pad8(d->hidden);
    SavedActs s{};
    s.k = kept_carve(d, cv);
    s.t = cv.halves(bn * 3 * C); s.qkv = cv.halves(bn * 3 * C); s.x1 = cv.halves(bn * C);
    s.u = cv.halves(bn * 2 * Hp); s.d = cv.halves(bn * 2 * Hp); s.g = cv.halves(bn * Hp);
    return s;
}
static size_t pk_block_saved_bytes(const rst_block_desc* d) {
    Carve cv(nullptr);
    saved_carve(d, cv);
    return cv.total();
}

// x1 = x + attention(LN1(x));  y = x1 + ffn(LN2(x1))   (model/Restromer.py:163-167), y may alias x.
// `kept`: the attention's small matrices are also written there; `saved`: those and every intermediate (never both)
static int pk_block_fwd(const rst_block_desc* d, const rst_block_params* p, const void* prepared, const bf16_t* x, bf16_t* y,
                        void* kept, void* saved, void* ws, size_t ws_bytes, hipStream_t stream) {
    RST_REQUIRE(x && y && p, "bf16 block: null operand");
    RST_REQUIRE(p->temperature && p->proj_weight, "bf16 block: null attention parameter");
    RST_REQUIRE(!d->conv_bias || p->proj_bias, "bf16 block: conv_bias set but a bias is null");
    if (ws == nullptr || ws_bytes < pk_block_ws_bytes(d))
        return fail(RST_ERR_WORKSPACE, "bf16 block: workspace %zu < %zu bytes", ws_bytes, pk_block_ws_bytes(d));
    const int C = d->C, Hd = d->hidden, Hp = pad8(Hd), heads = d->heads, c = C / heads, B = d->B;
    const int64_t N = (int64_t)d->H * d->W;
    const int big = NOSEG;
    const bool cb = d->conv_bias != 0;
    const int ln_mode = d->ln_with_bias ? 1 : 2;
    Carve cv(ws); PkBlockBufs b{};
    {
        void* own = cv.bytes(pk_block_prepared_bytes(d));
        Carve cp(prepared ? const_cast<void*>(prepared) : own);
        pk_block_carve_prepared(d, cp, b);
        if (!prepared) RST_TRY(pk_block_prepare(d, p, b, stream));
    }
    pk_block_carve(d, cv, b);
    KeptAttn keep{};
    SavedActs sv{};
    RST_REQUIRE(kept == nullptr || saved == nullptr, "bf16 block: either the kept attention or the saved activations");
    if (kept != nullptr) {
        RST_REQUIRE(aligned16(kept), "bf16 block: the buffer for the kept attention must be 16-byte aligned");
        Carve ck(kept);
        keep = kept_carve(d, ck);
        b.attn = keep.attn; b.mfold = keep.mfold;
    }
    if (saved != nullptr) {
        RST_REQUIRE(aligned16(saved), "bf16 block: the buffer for the saved activations must be 16-byte aligned");
        RST_REQUIRE(y != x, "bf16 block: a forward pass that saves its activations cannot run in place");
        Carve cs(saved);
        sv = saved_carve(d, cs);
        keep = sv.k;
        b.attn = keep.attn; b.mfold = keep.mfold;
        kept = saved;
    }
    bf16_t* t = saved ? sv.t : b.buf1; bf16_t* qkv = saved ? sv.qkv : b.buf2;
    bf16_t* u = saved ? sv.u : b.buf1; bf16_t* gbuf = saved ? sv.g : b.buf2;
    bf16_t* x1 = saved ? sv.x1 : y;
    const int64_t cn = (int64_t)C * N;

    // ---- MDTA (model/Restromer.py:128-150) ----
    // The fused 1x1 + depthwise kernel (pwdw_ring_kernel: the intermediate stays in LDS) takes both pairs wherever it
    // supports the shape (at most 96 input channels).  RST_BF16_FUSE=0: never (tests, A/B runs)
    static const int fuse_mode = getenv("RST_BF16_FUSE") != nullptr ? atoi(getenv("RST_BF16_FUSE")) : 1;
    const bool fuse = fuse_mode > 0 && saved == nullptr && pwdw_pk_supported(C / 8, d->W, d->H);
    const bool fuse_gate = fuse;
    if (fuse) {
        PwDwPk f{};
        f.A = b.wqkv.frag; f.X = x; f.x_bs = cn; f.Y = qkv; f.y_bs = 3 * cn;
        f.ln_s = b.wqkv.s; f.ln_c = b.wqkv.c; f.ln_mode = ln_mode; f.dww = b.dw_qkv;
        f.K = C; f.Kb = C / 8; f.Mb = 3 * C / 8; f.B = B; f.H = d->H; f.W = d->W; f.gate = 0;
        RST_TRY(pwdw_pk_launch(f, stream));
    } else {
        GemmPk g = pk_gemm_args(b.wqkv, x, cn, t, 3 * cn, B, N);
        g.ln_mode = ln_mode; g.ln_s = b.wqkv.s; g.ln_c = b.wqkv.c;
        RST_TRY(gemm_pk_launch(g, stream));
        DwPk a{};
        a.x = t; a.x_bs = 3 * cn; a.wp = b.dw_qkv; a.y = qkv; a.y_bs = 3 * cn;
        a.B = B; a.Cb = 3 * C / 8; a.H = d->H; a.W = d->W; a.gate = 0;
        RST_TRY(dw_pk_launch(a, stream));
    }
    {
        GramPk g{};
        g.Q = qkv; g.Kx = qkv + cn; g.bs = 3 * cn; g.cb = c / 8; g.heads = heads; g.B = B; g.N = N;
        g.splits = b.splits; g.chunk_px = b.chunk; g.partial = b.gram_part; g.sqq = b.sqq_part; g.sqk = b.sqk_part;
        RST_TRY(gram_pk_launch(g, stream));
    }
    if (b.splits > 16) {
        RST_TRY(reduce_slabs3_launch(b.gram_part, b.gram_red, (int64_t)c * c, b.sqq_part, b.sqq_red, c, b.sqk_part, b.sqk_red, c, B * heads, b.splits, stream));
        RST_TRY(attn_softmax_launch(b.gram_red, b.sqq_red, b.sqk_red, p->temperature, b.attn, keep.ghat, keep.nq, keep.nk,
                                    B, heads, c, 1, stream));
    } else {
        RST_TRY(attn_softmax_launch(b.gram_part, b.sqq_part, b.sqk_part, p->temperature, b.attn, keep.ghat, keep.nq, keep.nk,
                                    B, heads, c, b.splits, stream));
    }
    // M_b = W_o blockdiag(A) leaves the fold kernel as gemm_pk's operand fragments (and in fp32 when the backward keeps it)
    RST_TRY(attn_fold_frag_launch(p->proj_weight, b.attn, kept != nullptr ? b.mfold : nullptr, b.wattn.frag,
                                  (int64_t)b.wattn.frag_elems(), b.wattn.ntiles, b.wattn.KS, B, C, heads, c, stream));
    if (cb) RST_TRY(pk_prep_sc_launch(p->proj_weight, C, nullptr, nullptr, p->proj_bias, C, C, big, big, b.wattn.npos,
                                      b.wattn.s, b.wattn.c, stream));
    {
        GemmPk g = pk_gemm_args(b.wattn, qkv + 2 * cn, 3 * cn, x1, cn, B, N);
        g.a_bs = (int64_t)b.wattn.frag_elems() / 8;
        g.R = x; g.r_bs = cn;
        g.ln_c = cb ? b.wattn.c : nullptr;
        RST_TRY(gemm_pk_launch(g, stream));
    }
    // ---- GDFN (model/Restromer.py:99-104) ----
    if (fuse_gate) {
        PwDwPk f{};
        f.A = b.wpin.frag; f.X = x1; f.x_bs = cn; f.Y = gbuf; f.y_bs = (int64_t)Hp * N;
        f.ln_s = b.wpin.s; f.ln_c = b.wpin.c; f.ln_mode = ln_mode; f.dww = b.dw_ffn;
        f.K = C; f.Kb = C / 8; f.Mb = 2 * Hp / 8; f.B = B; f.H = d->H; f.W = d->W; f.gate = 1;
        RST_TRY(pwdw_pk_launch(f, stream));
    } else {
        GemmPk g = pk_gemm_args(b.wpin, x1, cn, u, (int64_t)2 * Hp * N, B, N);
        g.ln_mode = ln_mode; g.ln_s = b.wpin.s; g.ln_c = b.wpin.c;
        RST_TRY(gemm_pk_launch(g, stream));
        DwPk a{};
        a.x = u; a.x_bs = (int64_t)2 * Hp * N; a.wp = b.dw_ffn;
        a.B = B; a.Cb = 2 * Hp / 8; a.H = d->H; a.W = d->W;
        if (saved != nullptr) {
            // the backward pass wants both halves before the gate as well as their product
            a.y = sv.d; a.y_bs = (int64_t)2 * Hp * N; a.gate = 0;
            RST_TRY(dw_pk_launch(a, stream));
            RST_TRY(gate_fwd_pk_launch(sv.d, gbuf, B, Hp / 8, N, stream));
        } else {
            a.y = gbuf; a.y_bs = (int64_t)Hp * N; a.gate = 1;
            RST_TRY(dw_pk_launch(a, stream));
        }
    }
    {
        GemmPk g = pk_gemm_args(b.wpout, gbuf, (int64_t)Hp * N, y, cn, B, N);
        g.R = x1; g.r_bs = cn;
        g.ln_c = cb ? b.wpout.c : nullptr;
        RST_TRY(gemm_pk_launch(g, stream));
    }
    return RST_OK;
}

// ================================ backward ==========================================
// Gradients of one block from its packed bf16 input and output gradient.  The block is evaluated again first (the
// unfused kernels, every intermediate kept in bf16 in the workspace), then walked backwards: the data gradients are
// 1x1 convolutions with transposed weights (gemm_pk; contractions over more than 480 channels in slices that add
// through the output) and depthwise convolutions with rotated taps; the weight gradients are products over the pixels
// (wgrad_pk); the LayerNorm in front of a 1x1 convolution is differentiated through the product with the normalised
// tensor (lnlin_grads).  Parameter gradients are fp32.
namespace {

constexpr int PK_KSLICE = 480;          // widest contraction of one gemm_pk launch

// fragments of A[M positions][K positions] in slices of the contraction
struct PkSliced {
    int nslices;
    PkWeight w[4];
    int kpos0[4], krange[4];
};
PkSliced pk_sliced_shape(int Mpos, int Kpos) {
    PkSliced s{};
    s.nslices = (int)cdiv(Kpos, PK_KSLICE);
    for (int i = 0; i < s.nslices; ++i) {
        s.kpos0[i] = i * PK_KSLICE;
        s.krange[i] = Kpos - s.kpos0[i] < PK_KSLICE ? Kpos - s.kpos0[i] : PK_KSLICE;
        s.w[i] = pk_weight_shape(Mpos, s.krange[i], NOSEG, NOSEG);
    }
    return s;
}
void pk_sliced_carve(PkSliced& s, Carve& cv, int nb) {
    for (int i = 0; i < s.nslices; ++i) s.w[i].frag = cv.halves(s.w[i].frag_elems() * nb);
}
// Y[b] = A X[b] slice by slice; X holds the contraction's positions as channel blocks
int pk_sliced_gemm(const PkSliced& s, const bf16_t* x, int64_t x_bs, bf16_t* y, int64_t y_bs, int B, int64_t N, int64_t a_bs,
                   hipStream_t stream) {
    for (int i = 0; i < s.nslices; ++i) {
        GemmPk g = pk_gemm_args(s.w[i], x + (int64_t)(s.kpos0[i] / 8) * N * 8, x_bs, y, y_bs, B, N);
        g.a_bs = a_bs;
        if (i > 0) { g.R = y; g.r_bs = y_bs; }
        RST_TRY(gemm_pk_launch(g, stream));
    }
    return RST_OK;
}

struct PkBwdBufs {
    PkBlockBufs f;                                  // forward: prepared weights and attention scratch
    bf16_t *t, *qkv, *x1, *u, *d, *g;               // recomputed intermediates
    bf16_t *dg, *dd, *du, *xh, *dn, *dx1, *dqkv, *dt;
    float *ghat, *nq, *nk;
    float *P, *Eq, *Dq, *Dk, *dGG, *dtemp_row, *dtemp_b, *dwo_b, *mats;
    PkSliced wpoutT, wpinT, wqkvT, wmat;
    PkWeight wmfT;
    float *dw_qkv_flip, *dw_ffn_flip;
    float *part, *red, *dw_part, *dw_red;          // red: [M][K] sums, then M row sums
    float* part_p;                                  // slabs of P = dx1 v^T (its own: that product runs beside the others)
    size_t part_floats, part_p_floats, dw_floats;
};

WgradPk pk_wgrad_shape(int Ab, int Bb, int B, int64_t N, bool with_rowsum) {
    WgradPk g{};
    g.Ab = Ab; g.Bb = Bb; g.B = B; g.N = N; g.with_rowsum = with_rowsum ? 1 : 0;
    wgrad_pk_plan(&g);
    return g;
}

void pk_bwd_carve(const rst_block_desc* d, Carve& cv, PkBwdBufs& b, bool own_acts = true) {
    const int C = d->C, Hd = d->hidden, Hp = pad8(Hd), heads = d->heads, c = C / heads, B = d->B;
    const int64_t N = (int64_t)d->H * d->W;
    const size_t bn = (size_t)B * N;
    pk_block_carve_prepared(d, cv, b.f);
    pk_block_carve(d, cv, b.f);                      // (its buf1 / buf2 are not used here)
    if (own_acts) {
        b.t = cv.halves(bn * 3 * C); b.qkv = cv.halves(bn * 3 * C); b.x1 = cv.halves(bn * C);
        b.u = cv.halves(bn * 2 * Hp); b.d = cv.halves(bn * 2 * Hp); b.g = cv.halves(bn * Hp);
    }
    b.dg = cv.halves(bn * Hp); b.dd = cv.halves(bn * 2 * Hp); b.du = cv.halves(bn * 2 * Hp);
    b.xh = cv.halves(bn * C); b.dn = cv.halves(bn * C); b.dx1 = cv.halves(bn * C);
    b.dqkv = cv.halves(bn * 3 * C); b.dt = cv.halves(bn * 3 * C);
    const size_t BH = (size_t)B * heads;
    b.ghat = cv.floats(BH * c * c); b.nq = cv.floats((size_t)B * C); b.nk = cv.floats((size_t)B * C);
    b.P = cv.floats((size_t)B * C * C);
    b.Eq = cv.floats(BH * c * c); b.Dq = cv.floats((size_t)B * C); b.Dk = cv.floats((size_t)B * C);
    b.dGG = cv.floats(BH * c * c); b.dtemp_row = cv.floats((size_t)B * C); b.dtemp_b = cv.floats(BH);
    b.dwo_b = cv.floats((size_t)B * C * C);
    b.mats = cv.floats((size_t)B * 4 * C * C);
    b.wpoutT = pk_sliced_shape(Hp, C);        pk_sliced_carve(b.wpoutT, cv, 1);
    b.wpinT = pk_sliced_shape(C, 2 * Hp);     pk_sliced_carve(b.wpinT, cv, 1);
    b.wqkvT = pk_sliced_shape(C, 3 * C);      pk_sliced_carve(b.wqkvT, cv, 1);
    b.wmfT = pk_plain_weight(C, C);           b.wmfT.frag = cv.halves(b.wmfT.frag_elems() * B);
    b.wmat = pk_sliced_shape(2 * C, 2 * C);   pk_sliced_carve(b.wmat, cv, B);
    b.dw_qkv_flip = cv.floats((size_t)(3 * C / 8) * 80);
    b.dw_ffn_flip = cv.floats((size_t)(2 * Hp / 8) * 80);
    // the widest of the products over the pixels
    const int shapes[4][2] = {{C / 8, Hp / 8}, {2 * Hp / 8, C / 8}, {C / 8, C / 8}, {3 * C / 8, C / 8}};
    b.part_floats = 0;
    for (const auto& sh : shapes) {
        const WgradPk g = pk_wgrad_shape(sh[0], sh[1], B, N, true);
        b.part_floats = b.part_floats > wgrad_pk_partial_floats(g) ? b.part_floats : wgrad_pk_partial_floats(g);
    }
    b.part = cv.floats(b.part_floats);
    b.part_p_floats = wgrad_pk_partial_floats(pk_wgrad_shape(C / 8, C / 8, B, N, false));
    b.part_p = cv.floats(b.part_p_floats);
    const int wmax = 3 * C > 2 * Hp ? 3 * C : 2 * Hp;
    b.red = cv.floats((size_t)wmax * (C > Hp ? C : Hp) + wmax);
    const size_t dwa = dw_wgrad_pk_partial_floats(B, 3 * C / 8, d->H, d->W), dwb = dw_wgrad_pk_partial_floats(B, 2 * Hp / 8, d->H, d->W);
    b.dw_floats = dwa > dwb ? dwa : dwb;
    b.dw_part = cv.floats(b.dw_floats);
    b.dw_red = cv.floats((size_t)(wmax / 8) * 80);
}
size_t pk_block_bwd_ws_bytes(const rst_block_desc* d, bool own_acts = true) {
    Carve cv(nullptr); PkBwdBufs b{};
    pk_bwd_carve(d, cv, b, own_acts);
    return cv.total();
}

// out = sum over (samples and) splits of the product over the pixels, [Ab*8][Bb*8] (+ the Ab*8 row sums of A behind
// it when asked); `per_sample`: one such record per sample
int pk_wgrad(const bf16_t* a, int64_t a_bs, int Ab, const bf16_t* bm, int64_t b_bs, int Bb, int B, int64_t N, bool per_sample,
             bool with_rowsum, float* out, float* part, size_t part_floats, hipStream_t stream) {
    WgradPk g = pk_wgrad_shape(Ab, Bb, B, N, with_rowsum);
    g.A = a; g.a_bs = a_bs; g.Bm = bm; g.b_bs = b_bs;
    g.partial = part;
    RST_REQUIRE(wgrad_pk_partial_floats(g) <= part_floats, "bf16 block backward: weight-gradient scratch too small");
    RST_TRY(wgrad_pk_launch(g, stream));
    const int64_t rec = (int64_t)Ab * 8 * Bb * 8 + (with_rowsum ? Ab * 8 : 0);
    return reduce_slabs_launch(part, out, rec, per_sample ? B : 1, per_sample ? g.splits : B * g.splits, 0, stream);
}

// ---- 1x1 convolution on packed tensors, backward (the reductions after a skip concatenation) ----
struct Conv1BwdBufs {
    PkSliced wT;
    float *part, *red;
    size_t part_floats;
};
void conv1_bwd_carve(int B, int Cin, int Cout, int64_t N, Carve& cv, Conv1BwdBufs& b) {
    const int Cip = pad8(Cin), Cop = pad8(Cout);
    b.wT = pk_sliced_shape(Cip, Cop);
    pk_sliced_carve(b.wT, cv, 1);
    b.part_floats = wgrad_pk_partial_floats(pk_wgrad_shape(Cop / 8, Cip / 8, B, N, true));
    b.part = cv.floats(b.part_floats);
    b.red = cv.floats((size_t)Cop * Cip + Cop);
}

// The backward pass of a block is two kinds of work: the chain of data gradients (each kernel waits for the one before
// it) and the parameter gradients (products over the pixels, slab sums, small rearrangements), which only READ what
// the chain leaves behind.  At training sizes most of these kernels fill a fraction of the chip, so with
// RST_BWD_STREAMS=1 the parameter gradients run on a second stream of the library's own, one per device: it waits for
// an event the caller's stream records when an operand is ready, and the caller's stream waits for it once, before the
// call returns.  OFF by default: on MI355X a wave that issues v_mfma_f32_16x16x32_{f16,bf16} back to back can make a
// wave of ANOTHER dispatch on the same CU lose lanes 48..63 of one multiply-add of a dependent chain (measured with
// tools/two_stream_repro.hip, no library code involved; never seen between waves of one dispatch), and the weight
// gradient products on the side stream are exactly such waves beside the depthwise and LayerNorm kernels of the chain.
// With the switch on, the stream and its events are shared by every caller on the device, so a call holds the
// device's mutex from its first hand-over to its join: concurrent callers on one device take turns.
bool bwd_side_stream_enabled() {
    static const bool on = getenv("RST_BWD_STREAMS") != nullptr && getenv("RST_BWD_STREAMS")[0] == '1';
    return on;
}
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t fork[4] = {};
    hipEvent_t join = nullptr;
    std::mutex in_use;            // held by the one call that is handing work to `s`
};
int side_stream_for_current_device(SideStream** out) {
    static std::mutex mu;
    static std::map<int, SideStream> per_device;         // node-based: the addresses handed out stay valid
    int dev = 0;
    RST_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = per_device.find(dev);
    if (it == per_device.end()) {
        SideStream& sd = per_device[dev];
        if (hipStreamCreateWithFlags(&sd.s, hipStreamNonBlocking) != hipSuccess) {
            per_device.erase(dev);
            return fail(RST_ERR_HIP, "bf16 block backward: cannot create the side stream");
        }
        bool ok = true;
        for (auto& e : sd.fork) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            per_device.erase(dev);
            return fail(RST_ERR_HIP, "bf16 block backward: cannot create the side stream's events");
        }
        it = per_device.find(dev);
    }
    *out = &it->second;
    return RST_OK;
}

}  // namespace

static int pk_block_bwd(const rst_block_desc* d, const rst_block_params* p, const void* prepared, const bf16_t* x, const bf16_t* dy,
                        bf16_t* dx, rst_block_grads* gr, const void* kept, const void* saved, void* ws, size_t ws_bytes,
                        hipStream_t stream) {
    RST_REQUIRE(x && dy && dx && p && gr, "bf16 block backward: null operand");
    RST_REQUIRE(gr->norm1_weight && gr->norm2_weight && gr->temperature && gr->qkv_weight && gr->qkv_dw_weight && gr->proj_weight &&
                gr->pin_weight && gr->ffn_dw_weight && gr->pout_weight, "bf16 block backward: null gradient buffer");
    RST_REQUIRE(!d->ln_with_bias || (gr->norm1_bias && gr->norm2_bias), "bf16 block backward: null LayerNorm bias gradient buffer");
    RST_REQUIRE(!d->conv_bias || (gr->qkv_bias && gr->qkv_dw_bias && gr->proj_bias && gr->pin_bias && gr->ffn_dw_bias && gr->pout_bias),
                "bf16 block backward: null bias gradient buffer");
    RST_REQUIRE(kept == nullptr || saved == nullptr, "bf16 block backward: either the kept attention or the saved activations");
    const bool own_acts = saved == nullptr;
    if (ws == nullptr || ws_bytes < pk_block_bwd_ws_bytes(d, own_acts))
        return fail(RST_ERR_WORKSPACE, "bf16 block backward: workspace %zu < %zu bytes", ws_bytes, pk_block_bwd_ws_bytes(d, own_acts));
    const int C = d->C, Hd = d->hidden, Hp = pad8(Hd), heads = d->heads, c = C / heads, B = d->B;
    const int64_t N = (int64_t)d->H * d->W;
    const int64_t cn = (int64_t)C * N, hn = (int64_t)Hp * N;
    const bool cb = d->conv_bias != 0;
    const int wb = d->ln_with_bias ? 1 : 0;
    const int ln_mode = wb ? 1 : 2;
    Carve cv(ws); PkBwdBufs b{};
    pk_bwd_carve(d, cv, b, own_acts);
    if (prepared != nullptr) {
        Carve cp(const_cast<void*>(prepared));
        pk_block_carve_prepared(d, cp, b.f);             // the forward pass's prepared weights: nothing changed since
    } else {
        RST_TRY(pk_block_prepare(d, p, b.f, stream));
    }
    if (kept != nullptr) {
        // the forward pass kept its attention: the same matrices the output came from, nothing to form again
        RST_REQUIRE(aligned16(kept), "bf16 block backward: the kept attention must be 16-byte aligned");
        Carve ck(const_cast<void*>(kept));
        const KeptAttn k = kept_carve(d, ck);
        b.f.attn = k.attn; b.ghat = k.ghat; b.nq = k.nq; b.nk = k.nk; b.f.mfold = k.mfold;
    }
    if (saved != nullptr) {
        // the forward pass saved everything this pass reads: nothing is evaluated again
        RST_REQUIRE(aligned16(saved), "bf16 block backward: the saved activations must be 16-byte aligned");
        Carve cs(const_cast<void*>(saved));
        const SavedActs sv = saved_carve(d, cs);
        b.f.attn = sv.k.attn; b.ghat = sv.k.ghat; b.nq = sv.k.nq; b.nk = sv.k.nk; b.f.mfold = sv.k.mfold;
        b.t = sv.t; b.qkv = sv.qkv; b.x1 = sv.x1; b.u = sv.u; b.d = sv.d; b.g = sv.g;
    }

    SideStream* sd = nullptr;
    if (bwd_side_stream_enabled()) RST_TRY(side_stream_for_current_device(&sd));
    std::unique_lock<std::mutex> turn;
    if (sd != nullptr) turn = std::unique_lock<std::mutex>(sd->in_use);
    const hipStream_t side = sd ? sd->s : stream;
    int nfork = 0;
    // what the caller's stream has queued so far is complete before anything queued on the side stream from here on
    auto fork = [&]() -> int {
        if (sd == nullptr) return RST_OK;
        RST_REQUIRE(nfork < 4, "bf16 block backward: out of fork events");
        RST_HIP(hipEventRecord(sd->fork[nfork], stream));
        RST_HIP(hipStreamWaitEvent(side, sd->fork[nfork], 0));
        ++nfork;
        return RST_OK;
    };
    auto body = [&]() -> int {
    // ---------------- forward again, everything kept ----------------
    if (saved == nullptr) {
    {
        GemmPk g = pk_gemm_args(b.f.wqkv, x, cn, b.t, 3 * cn, B, N);
        g.ln_mode = ln_mode; g.ln_s = b.f.wqkv.s; g.ln_c = b.f.wqkv.c;
        RST_TRY(gemm_pk_launch(g, stream));
        DwPk a{};
        a.x = b.t; a.x_bs = 3 * cn; a.wp = b.f.dw_qkv; a.y = b.qkv; a.y_bs = 3 * cn;
        a.B = B; a.Cb = 3 * C / 8; a.H = d->H; a.W = d->W; a.gate = 0;
        RST_TRY(dw_pk_launch(a, stream));
    }
    {
        if (kept == nullptr) {
        GramPk g{};
        g.Q = b.qkv; g.Kx = b.qkv + cn; g.bs = 3 * cn; g.cb = c / 8; g.heads = heads; g.B = B; g.N = N;
        g.splits = b.f.splits; g.chunk_px = b.f.chunk; g.partial = b.f.gram_part; g.sqq = b.f.sqq_part; g.sqk = b.f.sqk_part;
        RST_TRY(gram_pk_launch(g, stream));
        if (b.f.splits > 16) {
            RST_TRY(reduce_slabs3_launch(b.f.gram_part, b.f.gram_red, (int64_t)c * c, b.f.sqq_part, b.f.sqq_red, c, b.f.sqk_part, b.f.sqk_red, c, B * heads, b.f.splits, stream));
            RST_TRY(attn_softmax_launch(b.f.gram_red, b.f.sqq_red, b.f.sqk_red, p->temperature, b.f.attn, b.ghat, b.nq, b.nk,
                                        B, heads, c, 1, stream));
        } else {
            RST_TRY(attn_softmax_launch(b.f.gram_part, b.f.sqq_part, b.f.sqk_part, p->temperature, b.f.attn, b.ghat, b.nq, b.nk,
                                        B, heads, c, b.f.splits, stream));
        }
        RST_TRY(attn_fold_launch(p->proj_weight, b.f.attn, b.f.mfold, B, C, heads, c, stream));
        }
        RST_TRY(pk_prep_a_launch(b.f.mfold, (int64_t)C * C, C, nullptr, C, C, NOSEG, NOSEG, b.f.wattn.ntiles, b.f.wattn.KS, b.f.wattn.frag,
                                 (int64_t)b.f.wattn.frag_elems(), B, stream));
        if (cb) RST_TRY(pk_prep_sc_launch(p->proj_weight, C, nullptr, nullptr, p->proj_bias, C, C, NOSEG, NOSEG, b.f.wattn.npos,
                                          b.f.wattn.s, b.f.wattn.c, stream));
        GemmPk m = pk_gemm_args(b.f.wattn, b.qkv + 2 * cn, 3 * cn, b.x1, cn, B, N);
        m.a_bs = (int64_t)b.f.wattn.frag_elems() / 8;
        m.R = x; m.r_bs = cn;
        m.ln_c = cb ? b.f.wattn.c : nullptr;
        RST_TRY(gemm_pk_launch(m, stream));
    }
    {
        GemmPk g = pk_gemm_args(b.f.wpin, b.x1, cn, b.u, 2 * hn, B, N);
        g.ln_mode = ln_mode; g.ln_s = b.f.wpin.s; g.ln_c = b.f.wpin.c;
        RST_TRY(gemm_pk_launch(g, stream));
        DwPk a{};
        a.x = b.u; a.x_bs = 2 * hn; a.wp = b.f.dw_ffn; a.y = b.d; a.y_bs = 2 * hn;
        a.B = B; a.Cb = 2 * Hp / 8; a.H = d->H; a.W = d->W; a.gate = 0;
        RST_TRY(dw_pk_launch(a, stream));
    }
    }

    // ---------------- weights of the data gradients ----------------
    {
        PrepAtJobs jobs{};
        auto add = [&](const PkSliced& sl, const float* w, int ldw, int M, int Ksrc, int kseg, int kseg_pad) {
            for (int i = 0; i < sl.nslices; ++i) {
                PrepAtJob& q = jobs.job[jobs.n++];
                q.w = w; q.ldw = ldw; q.transA = 1; q.M = M; q.mseg = NOSEG; q.mseg_pad = NOSEG;
                q.Ksrc = Ksrc; q.kseg = kseg; q.kseg_pad = kseg_pad; q.kpos0 = sl.kpos0[i]; q.Krange = sl.krange[i];
                q.ntiles = sl.w[i].ntiles; q.KS = sl.w[i].KS; q.frag = sl.w[i].frag;
            }
        };
        RST_REQUIRE(b.wpoutT.nslices + b.wpinT.nslices + b.wqkvT.nslices <= 8, "bf16 block backward: too many weight slices");
        add(b.wpoutT, p->pout_weight, Hd, Hd, C, NOSEG, NOSEG);          // dg = Wout^T dy: rows = hidden positions
        add(b.wpinT, p->pin_weight, C, C, 2 * Hd, Hd, -1);              // dn2 = Win^T du over the interleaved 2 Hp positions
        add(b.wqkvT, p->qkv_weight, C, C, 3 * C, NOSEG, NOSEG);         // dn1 = Wqkv^T dt
        RST_TRY(pk_prep_at_jobs_launch(jobs, stream));
    }
    RST_TRY(pk_prep_dw_launch(p->qkv_dw_weight, nullptr, 3 * C, NOSEG, NOSEG, 3 * C / 8, b.dw_qkv_flip, stream, 1));
    RST_TRY(pk_prep_dw_launch(p->ffn_dw_weight, nullptr, 2 * Hd, Hd, -1, 2 * Hp / 8, b.dw_ffn_flip, stream, 1));

    // ---------------- GDFN: y = x1 + Wout (gelu(d1) d2), d = dw(Win LN2(x1)) ----------------
    // Four hand-overs to the side stream per block, not one per operand: recording an event and waiting for it cost
    // the host about as much as a kernel launch, and the step must not become bound by the host's queueing.
    RST_TRY(fork());                                 // d (and x1, for the statistics below) are there
    if (saved == nullptr) RST_TRY(gate_fwd_pk_launch(b.d, b.g, B, Hp / 8, N, side));     // g: only project_out's weight gradient reads it
    RST_TRY(pk_wgrad(dy, cn, C / 8, b.g, hn, Hp / 8, B, N, false, cb, b.red, b.part, b.part_floats, side));
    RST_TRY(crop2d_launch(b.red, Hp, gr->pout_weight, C, Hd, side));
    if (cb) RST_HIP(hipMemcpyAsync(gr->pout_bias, b.red + (size_t)C * Hp, sizeof(float) * C, hipMemcpyDeviceToDevice, side));
    RST_TRY(ln_fwd_pk_launch(b.x1, b.xh, B, C / 8, N, wb, side));
    RST_TRY(pk_sliced_gemm(b.wpoutT, dy, cn, b.dg, hn, B, N, 0, stream));
    RST_TRY(gate_bwd_pk_launch(b.dg, b.d, b.dd, B, Hp / 8, N, stream));
    {
        DwPk w{};
        w.x = b.dd; w.x_bs = 2 * hn; w.wp = b.dw_ffn_flip; w.y = b.du; w.y_bs = 2 * hn;
        w.B = B; w.Cb = 2 * Hp / 8; w.H = d->H; w.W = d->W; w.gate = 0;
        RST_TRY(dw_pk_launch(w, stream));
    }
    RST_TRY(fork());                                 // dd, du
    {
        DwWgradPk a{};
        a.dy = b.dd; a.dy_bs = 2 * hn; a.x = b.u; a.x_bs = 2 * hn; a.B = B; a.Cb = 2 * Hp / 8; a.H = d->H; a.W = d->W;
        a.partial = b.dw_part;
        RST_TRY(dw_wgrad_pk_launch(a, b.dw_red, side));
        RST_TRY(dw_grad_unpack_launch(b.dw_red, gr->ffn_dw_weight, cb ? gr->ffn_dw_bias : nullptr, 2 * Hd, Hd, -1, 2 * Hp / 8, side));
    }
    RST_TRY(pk_wgrad(b.du, 2 * hn, 2 * Hp / 8, b.xh, cn, C / 8, B, N, false, true, b.red, b.part, b.part_floats, side));
    RST_TRY(lnlin_grads_launch(b.red, b.red + (size_t)2 * Hp * C, p->pin_weight, p->norm2_weight, wb ? p->norm2_bias : nullptr, gr->pin_weight,
                               cb ? gr->pin_bias : nullptr, gr->norm2_weight, wb ? gr->norm2_bias : nullptr, 2 * Hd, C, Hd, -1, side));
    RST_TRY(pk_sliced_gemm(b.wpinT, b.du, 2 * hn, b.dn, cn, B, N, 0, stream));
    RST_TRY(ln_bwd_pk_launch(b.dn, b.x1, p->norm2_weight, dy, b.dx1, B, C / 8, N, wb, stream));

    // ---------------- MDTA: x1 = x + M_b v, M_b = Wo blockdiag(softmax) ----------------
    const bf16_t* q = b.qkv; const bf16_t* k = b.qkv + cn; const bf16_t* v = b.qkv + 2 * cn;
    RST_TRY(pk_wgrad(b.dx1, cn, C / 8, v, 3 * cn, C / 8, B, N, true, false, b.P, b.part_p, b.part_p_floats, stream));
    RST_TRY(attn_bwd_small_launch(b.P, p->proj_weight, b.f.attn, b.ghat, b.nq, b.nk, p->temperature, b.Eq, b.Dq, b.Dk,
                                  b.dGG, b.dtemp_row, b.dtemp_b, B, C, heads, c, stream));
    RST_TRY(fork());                                 // dx1, P, the per-sample sums of the temperature's gradient
    if (cb) {
        // the projection's bias gradient: the sums of dx1's rows over everything (the product with its own rows is a
        // by-product nobody reads)
        RST_TRY(pk_wgrad(b.dx1, cn, C / 8, b.dx1, cn, 1, B, N, false, true, b.red, b.part, b.part_floats, side));
        RST_HIP(hipMemcpyAsync(gr->proj_bias, b.red + (size_t)C * 8, sizeof(float) * C, hipMemcpyDeviceToDevice, side));
    }
    RST_TRY(attn_bwd_wo_launch(b.P, b.f.attn, b.dwo_b, B, C, heads, c, side));
    RST_TRY(reduce_slabs_launch(b.dwo_b, gr->proj_weight, (int64_t)C * C, 1, B, 0, side));
    RST_TRY(reduce_slabs_launch(b.dtemp_b, gr->temperature, heads, 1, B, 0, side));
    RST_TRY(ln_fwd_pk_launch(x, b.xh, B, C / 8, N, wb, side));      // (the product that read x1's statistics is queued before it)
    RST_TRY(attn_bwd_mats_launch(b.Eq, b.Dq, b.Dk, b.mats, B, C, c, stream));
    for (int i = 0; i < b.wmat.nslices; ++i)
        RST_TRY(pk_prep_at_launch(b.mats, (int64_t)4 * C * C, 2 * C, 0, 2 * C, NOSEG, NOSEG, 2 * C, NOSEG, NOSEG, b.wmat.kpos0[i],
                                  b.wmat.krange[i], b.wmat.w[i].ntiles, b.wmat.w[i].KS, b.wmat.w[i].frag,
                                  (int64_t)b.wmat.w[i].frag_elems(), B, stream));
    RST_TRY(pk_prep_at_launch(b.f.mfold, (int64_t)C * C, C, 1, C, NOSEG, NOSEG, C, NOSEG, NOSEG, 0, C, b.wmfT.ntiles, b.wmfT.KS,
                              b.wmfT.frag, (int64_t)b.wmfT.frag_elems(), B, stream));
    {
        // [dq; dk] = A_b [q; k] (q and k are the first 2C channels of qkv);  dv = M^T dx1
        for (int i = 0; i < b.wmat.nslices; ++i) {
            GemmPk g = pk_gemm_args(b.wmat.w[i], b.qkv + (int64_t)(b.wmat.kpos0[i] / 8) * N * 8, 3 * cn, b.dqkv, 3 * cn, B, N);
            g.a_bs = (int64_t)b.wmat.w[i].frag_elems() / 8;
            if (i > 0) { g.R = b.dqkv; g.r_bs = 3 * cn; }
            RST_TRY(gemm_pk_launch(g, stream));
        }
        GemmPk g = pk_gemm_args(b.wmfT, b.dx1, cn, b.dqkv + 2 * cn, 3 * cn, B, N);
        g.a_bs = (int64_t)b.wmfT.frag_elems() / 8;
        RST_TRY(gemm_pk_launch(g, stream));
    }
    {
        DwPk w{};
        w.x = b.dqkv; w.x_bs = 3 * cn; w.wp = b.dw_qkv_flip; w.y = b.dt; w.y_bs = 3 * cn;
        w.B = B; w.Cb = 3 * C / 8; w.H = d->H; w.W = d->W; w.gate = 0;
        RST_TRY(dw_pk_launch(w, stream));
    }
    RST_TRY(fork());                                 // dqkv, dt
    {
        DwWgradPk a{};
        a.dy = b.dqkv; a.dy_bs = 3 * cn; a.x = b.t; a.x_bs = 3 * cn; a.B = B; a.Cb = 3 * C / 8; a.H = d->H; a.W = d->W;
        a.partial = b.dw_part;
        RST_TRY(dw_wgrad_pk_launch(a, b.dw_red, side));
        RST_TRY(dw_grad_unpack_launch(b.dw_red, gr->qkv_dw_weight, cb ? gr->qkv_dw_bias : nullptr, 3 * C, NOSEG, NOSEG, 3 * C / 8, side));
    }
    RST_TRY(pk_wgrad(b.dt, 3 * cn, 3 * C / 8, b.xh, cn, C / 8, B, N, false, true, b.red, b.part, b.part_floats, side));
    RST_TRY(lnlin_grads_launch(b.red, b.red + (size_t)3 * C * C, p->qkv_weight, p->norm1_weight, wb ? p->norm1_bias : nullptr, gr->qkv_weight,
                               cb ? gr->qkv_bias : nullptr, gr->norm1_weight, wb ? gr->norm1_bias : nullptr, 3 * C, C, NOSEG, NOSEG, side));
    RST_TRY(pk_sliced_gemm(b.wqkvT, b.dt, 3 * cn, b.dn, cn, B, N, 0, stream));
    return ln_bwd_pk_launch(b.dn, x, p->norm1_weight, b.dx1, dx, B, C / 8, N, wb, stream);
    };
    const int rc = body();
    if (sd != nullptr && nfork > 0) {
        // the caller's stream goes on only after the side stream has drained -- also when the body gave up half-way
        RST_HIP(hipEventRecord(sd->join, side));
        RST_HIP(hipStreamWaitEvent(stream, sd->join, 0));
    }
    return rc;
}

}  // namespace rst

using namespace rst;

extern "C" {

int rst_bwd_side_stream_enabled(void) { return bwd_side_stream_enabled() ? 1 : 0; }

size_t rst_bf16_packed_bytes(int32_t B, int32_t C, int64_t N) {
    if (B <= 0 || C <= 0 || N <= 0) return 0;
    return (size_t)B * (size_t)round_up(C, 8) * (size_t)N * sizeof(bf16_t);
}
int rst_bf16_pack(const void* x, int32_t x_is_fp32, void* y_packed, int32_t B, int32_t C, int64_t N, void* stream) {
    return pk_pack_launch(x, x_is_fp32, static_cast<bf16_t*>(y_packed), B, C, N, static_cast<hipStream_t>(stream));
}
int rst_bf16_unpack(const void* x_packed, void* y, int32_t y_is_fp32, int32_t B, int32_t C, int64_t N, void* stream) {
    return pk_unpack_launch(static_cast<const bf16_t*>(x_packed), y, y_is_fp32, B, C, N, static_cast<hipStream_t>(stream));
}

size_t rst_bf16_block_bwd_workspace_bytes(const rst_block_desc* desc) {
    if (check_pk_desc(desc) != RST_OK) return 0;
    return pk_block_bwd_ws_bytes(desc);
}
int rst_bf16_block_bwd(const rst_block_desc* desc, const rst_block_params* params, const void* prepared, const void* x_packed,
                       const void* dy_packed, void* dx_packed, rst_block_grads* grads, void* workspace, size_t workspace_bytes,
                       void* stream) {
    RST_TRY(check_pk_desc(desc));
    return pk_block_bwd(desc, params, prepared, static_cast<const bf16_t*>(x_packed), static_cast<const bf16_t*>(dy_packed),
                        static_cast<bf16_t*>(dx_packed), grads, nullptr, nullptr, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
size_t rst_bf16_block_kept_bytes(const rst_block_desc* desc) { return check_pk_desc(desc) == RST_OK ? pk_block_kept_bytes(desc) : 0; }
int rst_bf16_block_fwd_keep(const rst_block_desc* d, const rst_block_params* p, const void* prepared, const void* x_packed, void* y_packed,
                            void* kept, void* workspace, size_t workspace_bytes, void* stream) {
    RST_TRY(check_pk_desc(d));
    RST_REQUIRE(kept != nullptr, "bf16 block: null buffer for the kept attention");
    return pk_block_fwd(d, p, prepared, static_cast<const bf16_t*>(x_packed), static_cast<bf16_t*>(y_packed), kept, nullptr, workspace,
                        workspace_bytes, static_cast<hipStream_t>(stream));
}
int rst_bf16_block_bwd_kept(const rst_block_desc* desc, const rst_block_params* params, const void* prepared, const void* x_packed,
                            const void* dy_packed, void* dx_packed, rst_block_grads* grads, const void* kept, void* workspace,
                            size_t workspace_bytes, void* stream) {
    RST_TRY(check_pk_desc(desc));
    RST_REQUIRE(kept != nullptr, "bf16 block backward: null kept attention");
    return pk_block_bwd(desc, params, prepared, static_cast<const bf16_t*>(x_packed), static_cast<const bf16_t*>(dy_packed),
                        static_cast<bf16_t*>(dx_packed), grads, kept, nullptr, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}
size_t rst_bf16_block_saved_bytes(const rst_block_desc* desc) { return check_pk_desc(desc) == RST_OK ? pk_block_saved_bytes(desc) : 0; }
size_t rst_bf16_block_bwd_saved_workspace_bytes(const rst_block_desc* desc) {
    return check_pk_desc(desc) == RST_OK ? pk_block_bwd_ws_bytes(desc, false) : 0;
}
int rst_bf16_block_fwd_save(const rst_block_desc* d, const rst_block_params* p, const void* prepared, const void* x_packed, void* y_packed,
                            void* saved, void* workspace, size_t workspace_bytes, void* stream) {
    RST_TRY(check_pk_desc(d));
    RST_REQUIRE(saved != nullptr, "bf16 block: null buffer for the saved activations");
    return pk_block_fwd(d, p, prepared, static_cast<const bf16_t*>(x_packed), static_cast<bf16_t*>(y_packed), nullptr, saved, workspace,
                        workspace_bytes, static_cast<hipStream_t>(stream));
}
int rst_bf16_block_bwd_saved(const rst_block_desc* desc, const rst_block_params* params, const void* prepared, const void* x_packed,
                             const void* dy_packed, void* dx_packed, rst_block_grads* grads, const void* saved, void* workspace,
                             size_t workspace_bytes, void* stream) {
    RST_TRY(check_pk_desc(desc));
    RST_REQUIRE(saved != nullptr, "bf16 block backward: null saved activations");
    return pk_block_bwd(desc, params, prepared, static_cast<const bf16_t*>(x_packed), static_cast<const bf16_t*>(dy_packed),
                        static_cast<bf16_t*>(dx_packed), grads, nullptr, saved, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int rst_bf16_to_channels_last(const void* x_packed, void* y_nhwc, int32_t B, int32_t C, int64_t N, void* stream) {
    return pk_to_nhwc_launch(static_cast<const bf16_t*>(x_packed), static_cast<bf16_t*>(y_nhwc), B, C, N, static_cast<hipStream_t>(stream));
}
int rst_bf16_from_channels_last(const void* x_nhwc, void* y_packed, int32_t shuffle_mode, int32_t B, int32_t C, int32_t H, int32_t W,
                                void* stream) {
    return pk_from_nhwc_launch(static_cast<const bf16_t*>(x_nhwc), static_cast<bf16_t*>(y_packed), shuffle_mode, B, C, H, W,
                               static_cast<hipStream_t>(stream));
}

size_t rst_bf16_conv1x1_workspace_bytes(int32_t Cin, int32_t Cout) {
    if (Cin <= 0 || Cout <= 0 || gemm_pk_ks((int)cdiv(Cin, 8)) < 0) return 0;
    PkWeight w = pk_plain_weight(Cout, Cin);
    Carve cv(nullptr);
    pk_weight_carve(w, cv, 1);
    return cv.total();
}
int rst_bf16_conv1x1_fwd(const void* x_packed, const float* weight, const float* bias, const void* residual_packed,
                         void* y_packed, int32_t B, int32_t Cin, int32_t Cout, int64_t N,
                         void* workspace, size_t workspace_bytes, void* stream) {
    RST_REQUIRE(x_packed && weight && y_packed, "bf16 conv1x1: null operand");
    RST_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && N > 0, "bf16 conv1x1: empty shape");
    RST_REQUIRE(gemm_pk_ks((int)cdiv(Cin, 8)) > 0, "bf16 conv1x1: %d input channels not supported (at most 480)", Cin);
    if (workspace == nullptr || workspace_bytes < rst_bf16_conv1x1_workspace_bytes(Cin, Cout))
        return fail(RST_ERR_WORKSPACE, "bf16 conv1x1: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    PkWeight w = pk_plain_weight(Cout, Cin);
    Carve cv(workspace);
    pk_weight_carve(w, cv, 1);
    RST_TRY(pk_weight_prepare(w, weight, nullptr, nullptr, bias, st));
    const int64_t xs = (int64_t)round_up(Cin, 8) * N, ys = (int64_t)round_up(Cout, 8) * N;
    GemmPk g = pk_gemm_args(w, static_cast<const bf16_t*>(x_packed), xs, static_cast<bf16_t*>(y_packed), ys, B, N);
    g.R = static_cast<const bf16_t*>(residual_packed); g.r_bs = ys;
    g.ln_c = bias ? w.c : nullptr;
    return gemm_pk_launch(g, st);
}

size_t rst_bf16_conv1x1_bwd_workspace_bytes(int32_t B, int32_t Cin, int32_t Cout, int64_t N) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || N <= 0) return 0;
    Carve cv(nullptr); Conv1BwdBufs b{};
    conv1_bwd_carve(B, Cin, Cout, N, cv, b);
    return cv.total();
}
int rst_bf16_conv1x1_bwd(const void* x_packed, const void* dy_packed, const float* weight, void* dx_packed, float* dweight,
                         float* dbias, int32_t B, int32_t Cin, int32_t Cout, int64_t N, void* workspace, size_t workspace_bytes,
                         void* stream) {
    RST_REQUIRE(x_packed && dy_packed && weight && dx_packed && dweight, "bf16 conv1x1 backward: null operand");
    RST_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && N > 0, "bf16 conv1x1 backward: empty shape");
    RST_REQUIRE(aligned16(x_packed) && aligned16(dy_packed) && aligned16(dx_packed), "bf16 conv1x1 backward: tensors must be 16-byte aligned");
    if (workspace == nullptr || !aligned16(workspace) || workspace_bytes < rst_bf16_conv1x1_bwd_workspace_bytes(B, Cin, Cout, N))
        return fail(RST_ERR_WORKSPACE, "bf16 conv1x1 backward: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bf16_t* x = static_cast<const bf16_t*>(x_packed);
    const bf16_t* dy = static_cast<const bf16_t*>(dy_packed);
    Carve cv(workspace); Conv1BwdBufs b{};
    conv1_bwd_carve(B, Cin, Cout, N, cv, b);
    const int Cip = pad8(Cin), Cop = pad8(Cout);
    // dx = W^T dy: fragments of the transposed weights, the contraction over the output channels in slices of 480
    {
        PrepAtJobs jobs{};
        RST_REQUIRE(b.wT.nslices <= 8, "bf16 conv1x1 backward: too many slices");
        for (int i = 0; i < b.wT.nslices; ++i) {
            PrepAtJob& q = jobs.job[jobs.n++];
            q.w = weight; q.ldw = Cin; q.transA = 1; q.M = Cin; q.mseg = NOSEG; q.mseg_pad = NOSEG;
            q.Ksrc = Cout; q.kseg = NOSEG; q.kseg_pad = NOSEG; q.kpos0 = b.wT.kpos0[i]; q.Krange = b.wT.krange[i];
            q.ntiles = b.wT.w[i].ntiles; q.KS = b.wT.w[i].KS; q.frag = b.wT.w[i].frag;
        }
        RST_TRY(pk_prep_at_jobs_launch(jobs, st));
    }
    RST_TRY(pk_sliced_gemm(b.wT, dy, (int64_t)Cop * N, static_cast<bf16_t*>(dx_packed), (int64_t)Cip * N, B, N, 0, st));
    // dW = sum over samples and pixels of dy x^T, db = the sums of dy's rows
    RST_TRY(pk_wgrad(dy, (int64_t)Cop * N, Cop / 8, x, (int64_t)Cip * N, Cip / 8, B, N, false, true, b.red, b.part, b.part_floats, st));
    RST_TRY(crop2d_launch(b.red, Cip, dweight, Cout, Cin, st));
    if (dbias != nullptr) RST_HIP(hipMemcpyAsync(dbias, b.red + (size_t)Cop * Cip, sizeof(float) * Cout, hipMemcpyDeviceToDevice, st));
    return RST_OK;
}

size_t rst_bf16_block_workspace_bytes(const rst_block_desc* d) { return check_pk_desc(d) == RST_OK ? pk_block_ws_bytes(d) : 0; }
size_t rst_bf16_block_prepared_bytes(const rst_block_desc* d) { return check_pk_desc(d) == RST_OK ? pk_block_prepared_bytes(d) : 0; }
int rst_bf16_block_prepare(const rst_block_desc* d, const rst_block_params* p, void* prepared, void* stream) {
    RST_TRY(check_pk_desc(d));
    RST_REQUIRE(prepared != nullptr && aligned16(prepared), "bf16 block: prepared buffer null or misaligned");
    Carve cp(prepared); PkBlockBufs b{};
    pk_block_carve_prepared(d, cp, b);
    return pk_block_prepare(d, p, b, static_cast<hipStream_t>(stream));
}
int rst_bf16_block_fwd(const rst_block_desc* d, const rst_block_params* p, const void* prepared, const void* x_packed, void* y_packed,
                       void* workspace, size_t workspace_bytes, void* stream) {
    RST_TRY(check_pk_desc(d));
    return pk_block_fwd(d, p, prepared, static_cast<const bf16_t*>(x_packed), static_cast<bf16_t*>(y_packed), nullptr, nullptr, workspace,
                        workspace_bytes, static_cast<hipStream_t>(stream));
}

}  // extern "C"
