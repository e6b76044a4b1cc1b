// Depthwise 3x3 convolution (zero padding 1) over NCHW planes and its fused
// GDFN gate; reference: model/Restromer.py:87-95,99-104,116-124.
// HBM-bound: every input plane is read once and every output plane written
// once.  A work item is a strip of 4 consecutive pixels x ROWS rows of one
// channel plane; consecutive lanes take consecutive strips along a row, so the
// float4 loads of a wave cover contiguous row segments.  The left/right halo
// values come from the neighbouring lanes' registers (one cross-lane move
// each); only the first/last lane of a wave loads them itself.
#include "kernels.hpp"

namespace rst {

constexpr int DW_ROWS = 4;


// Load row `yy` of the plane for this lane's strip [x0, x0+4) plus halo.
// Returns 6 values v[0..5] = pixels x0-1 .. x0+4 (zeros outside the image).
template <bool VEC>
__device__ __forceinline__ void load_row6(float (&v)[6], const float* plane, int yy, int x0, int H, int W,
                                          bool active, int lane) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const bool row_ok = active && yy >= 0 && yy < H;
    const float* row = plane + (int64_t)yy * W;
    if (row_ok) {
        if constexpr (VEC) {
            c = *reinterpret_cast<const f32x4*>(row + x0);      // W % 4 == 0
        } else {
            c.x = row[x0];
            if (x0 + 1 < W) c.y = row[x0 + 1];
            if (x0 + 2 < W) c.z = row[x0 + 2];
            if (x0 + 3 < W) c.w = row[x0 + 3];
        }
    }
    float left, right;
    if constexpr (VEC) {
        // neighbours along the row live in the adjacent lanes (same row iff x0 > 0 / x0+4 < W)
        left = __shfl_up(c.w, 1, 64);
        right = __shfl_down(c.x, 1, 64);
        if (lane == 0) left = (row_ok && x0 > 0) ? row[x0 - 1] : 0.f;
        if (lane == 63) right = (row_ok && x0 + 4 < W) ? row[x0 + 4] : 0.f;
        if (x0 == 0) left = 0.f;
        if (x0 + 4 >= W) right = 0.f;
        if (!row_ok) { left = 0.f; right = 0.f; }
    } else {
        left = (row_ok && x0 > 0) ? row[x0 - 1] : 0.f;
        right = (row_ok && x0 + 4 < W) ? row[x0 + 4] : 0.f;
    }
    v[0] = left; v[1] = c.x; v[2] = c.y; v[3] = c.z; v[4] = c.w; v[5] = right;
}

__device__ __forceinline__ void load_taps(float (&k)[9], const float* w, int c, int flip) {
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[(int64_t)c * 9 + (flip ? 8 - i : i)];
}

// out[r][j] for the strip, given rows y0-1 .. y0+ROWS of 6 values each
__device__ __forceinline__ void conv_strip(float (&o)[DW_ROWS][4], const float (&in)[DW_ROWS + 2][6], const float (&k)[9],
                                           float bias) {
#pragma unroll
    for (int r = 0; r < DW_ROWS; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = bias;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) s = fmaf(in[r + dy][j + dx], k[dy * 3 + dx], s);
            o[r][j] = s;
        }
}

template <bool VEC>
__device__ __forceinline__ void store_strip(float* plane, const float (&o)[DW_ROWS][4], int y0, int x0, int H, int W) {
#pragma unroll
    for (int r = 0; r < DW_ROWS; ++r) {
        const int yy = y0 + r;
        if (yy < H) {
            float* dst = plane + (int64_t)yy * W + x0;
            if constexpr (VEC) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{o[r][0], o[r][1], o[r][2], o[r][3]};
            } else {
                dst[0] = o[r][0];
                if (x0 + 1 < W) dst[1] = o[r][1];
                if (x0 + 2 < W) dst[2] = o[r][2];
                if (x0 + 3 < W) dst[3] = o[r][3];
            }
        }
    }
}

// Decode a flat strip index -> (plane, y0, x0).  strips_x = ceil(W/4), strips_y = ceil(H/ROWS).
struct Strip { int64_t plane; int y0, x0; bool active; };
__device__ __forceinline__ Strip decode_strip(int64_t idx, int64_t total, int strips_x, int strips_y) {
    Strip s;
    s.active = idx < total;
    const int64_t i = s.active ? idx : 0;
    s.x0 = (int)(i % strips_x) * 4;
    const int64_t t = i / strips_x;
    s.y0 = (int)(t % strips_y) * DW_ROWS;
    s.plane = t / strips_y;
    return s;
}

template <bool VEC>
__global__ void __launch_bounds__(256)
dwconv_fwd_kernel(DwArgs a, int strips_x, int strips_y, int64_t total) {
    const int lane = threadIdx.x & 63;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const Strip s = decode_strip(idx, total, strips_x, strips_y);
    const int b = (int)(s.plane / a.C), c = (int)(s.plane % a.C);
    const float* xp = a.x + (int64_t)b * a.x_bstride + (int64_t)c * a.H * a.W;
    float in[DW_ROWS + 2][6];
#pragma unroll
    for (int r = 0; r < DW_ROWS + 2; ++r) load_row6<VEC>(in[r], xp, s.y0 - 1 + r, s.x0, a.H, a.W, s.active, lane);
    if (!s.active) return;
    float k[9];
    load_taps(k, a.w, c, a.flip);
    float o[DW_ROWS][4];
    conv_strip(o, in, k, a.bias ? a.bias[c] : 0.f);
    store_strip<VEC>(a.y + (int64_t)b * a.y_bstride + (int64_t)c * a.H * a.W, o, s.y0, s.x0, a.H, a.W);
}

// Dense 3x3 convolution with very few output channels (the network's last layer, 48 -> 3 at twice the
// input resolution; reference model/Restromer.py:381-383) plus the residual that follows it (:431).
// HBM-bound like the depthwise kernels (every input plane is read once, the output is small), so it
// reuses their work item: a strip of 4 pixels x 4 rows, here of ALL output channels of one sample.
// The input planes are walked one after the other; the 9*COUT weights of a plane are wave-uniform
// (scalar loads), the COUT*16 running sums stay in registers.
template <int COUT, bool VEC>
__global__ void __launch_bounds__(256)
conv3x3_fewout_kernel(Conv3Args a, int strips_x, int strips_y, int64_t total) {
    const int lane = threadIdx.x & 63;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const Strip s = decode_strip(idx, total, strips_x, strips_y);      // plane index = sample
    const int b = (int)s.plane;
    const int64_t hw = (int64_t)a.H * a.W;
    float o[COUT][DW_ROWS][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < DW_ROWS; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[co][r][j] = bv;
    }
    // the next input plane is fetched while the current one is multiplied
    float in[DW_ROWS + 2][6], nxt[DW_ROWS + 2][6];
    {
        const float* xp = a.x + (int64_t)b * a.Cin * hw;
#pragma unroll
        for (int r = 0; r < DW_ROWS + 2; ++r) load_row6<VEC>(in[r], xp, s.y0 - 1 + r, s.x0, a.H, a.W, s.active, lane);
    }
    for (int ci = 0; ci < a.Cin; ++ci) {
        {
            const int cn = ci + 1 < a.Cin ? ci + 1 : ci;
            const float* xp = a.x + ((int64_t)b * a.Cin + cn) * hw;
#pragma unroll
            for (int r = 0; r < DW_ROWS + 2; ++r) load_row6<VEC>(nxt[r], xp, s.y0 - 1 + r, s.x0, a.H, a.W, s.active, lane);
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float* wk = a.w + ((int64_t)co * a.Cin + ci) * 9;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float k = wk[dy * 3 + dx];
#pragma unroll
                    for (int r = 0; r < DW_ROWS; ++r)
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[co][r][j] = fmaf(in[r + dy][j + dx], k, o[co][r][j]);
                }
        }
#pragma unroll
        for (int r = 0; r < DW_ROWS + 2; ++r)
#pragma unroll
            for (int j = 0; j < 6; ++j) in[r][j] = nxt[r][j];
    }
    if (!s.active) return;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const int64_t off = ((int64_t)b * COUT + co) * hw;
        if (a.res != nullptr) {
#pragma unroll
            for (int r = 0; r < DW_ROWS; ++r) {
                const int yy = s.y0 + r;
                if (yy >= a.H) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (s.x0 + j < a.W) o[co][r][j] += a.res[off + (int64_t)yy * a.W + s.x0 + j];
            }
        }
        if (a.up_src != nullptr) {
            // + bilinear x2 of the low-resolution tensor (align_corners = False): source coordinate o / 2 - 1/4 clamped
            // at 0, weights (1/4, 3/4) or (3/4, 1/4), the far index clamped at the border
            const int h2 = a.H / 2, w2 = a.W / 2;
            const float* lp = a.up_src + ((int64_t)b * COUT + co) * h2 * w2;
#pragma unroll
            for (int r = 0; r < DW_ROWS; ++r) {
                const int yy = s.y0 + r;
                if (yy >= a.H) continue;
                const float sy = fmaxf(0.5f * (float)yy - 0.25f, 0.f);
                const int iy0 = (int)sy, iy1 = iy0 + 1 < h2 ? iy0 + 1 : h2 - 1;
                const float ly1 = sy - (float)iy0, ly0 = 1.0f - ly1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int xx = s.x0 + j;
                    if (xx >= a.W) continue;
                    const float sx = fmaxf(0.5f * (float)xx - 0.25f, 0.f);
                    const int ix0 = (int)sx, ix1 = ix0 + 1 < w2 ? ix0 + 1 : w2 - 1;
                    const float lx1 = sx - (float)ix0, lx0 = 1.0f - lx1;
                    const float p00 = lp[(int64_t)iy0 * w2 + ix0], p01 = lp[(int64_t)iy0 * w2 + ix1];
                    const float p10 = lp[(int64_t)iy1 * w2 + ix0], p11 = lp[(int64_t)iy1 * w2 + ix1];
                    o[co][r][j] += ly0 * (lx0 * p00 + lx1 * p01) + ly1 * (lx0 * p10 + lx1 * p11);
                }
            }
        }
        store_strip<VEC>(a.y + off, o[co], s.y0, s.x0, a.H, a.W);
    }
}

// GDFN gate: g[c] = gelu(dw(u[c])) * dw(u[c + hidden]), c < hidden
// (model/Restromer.py:101-102).  Optionally keeps the two depthwise outputs
// (d, [B][2*hidden][H][W]) for the backward pass.

template <bool VEC>
__global__ void __launch_bounds__(256)
dwconv_gate_fwd_kernel(GateArgs a, int strips_x, int strips_y, int64_t total) {
    const int lane = threadIdx.x & 63;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const Strip s = decode_strip(idx, total, strips_x, strips_y);
    const int b = (int)(s.plane / a.hidden), c = (int)(s.plane % a.hidden);
    const int64_t hw = (int64_t)a.H * a.W;
    const float* u1 = a.u + (int64_t)b * a.u_bstride + (int64_t)c * hw;
    const float* u2 = u1 + (int64_t)a.hidden * hw;
    float o1[DW_ROWS][4], o2[DW_ROWS][4];
    {
        float in[DW_ROWS + 2][6];
#pragma unroll
        for (int r = 0; r < DW_ROWS + 2; ++r) load_row6<VEC>(in[r], u1, s.y0 - 1 + r, s.x0, a.H, a.W, s.active, lane);
        float k[9];
        load_taps(k, a.w, c, 0);
        conv_strip(o1, in, k, a.bias ? a.bias[c] : 0.f);
    }
    {
        float in[DW_ROWS + 2][6];
#pragma unroll
        for (int r = 0; r < DW_ROWS + 2; ++r) load_row6<VEC>(in[r], u2, s.y0 - 1 + r, s.x0, a.H, a.W, s.active, lane);
        float k[9];
        load_taps(k, a.w, c + a.hidden, 0);
        conv_strip(o2, in, k, a.bias ? a.bias[c + a.hidden] : 0.f);
    }
    if (!s.active) return;
    if (a.d != nullptr) {
        float* d1 = a.d + (int64_t)b * a.d_bstride + (int64_t)c * hw;
        store_strip<VEC>(d1, o1, s.y0, s.x0, a.H, a.W);
        store_strip<VEC>(d1 + (int64_t)a.hidden * hw, o2, s.y0, s.x0, a.H, a.W);
    }
#pragma unroll
    for (int r = 0; r < DW_ROWS; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) o1[r][j] = gelu_erf(o1[r][j]) * o2[r][j];
    store_strip<VEC>(a.gout + (int64_t)b * a.g_bstride + (int64_t)c * hw, o1, s.y0, s.x0, a.H, a.W);
}

// Gate backward (elementwise): with (a, b2) = d[c], d[c+hidden] and upstream dg,
//   dd[c]        = dg * b2 * gelu'(a)
//   dd[c+hidden] = dg * gelu(a)
__global__ void gate_bwd_kernel(const float* __restrict__ dg, const float* __restrict__ d, float* __restrict__ dd,
                                int64_t hidden_hw, int64_t total /* B*hidden*HW */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t b = i / hidden_hw, r = i % hidden_hw;
    const int64_t i1 = b * 2 * hidden_hw + r, i2 = i1 + hidden_hw;
    const float a = d[i1], b2 = d[i2], g = dg[i];
    dd[i1] = g * b2 * gelu_erf_grad(a);
    dd[i2] = g * gelu_erf(a);
}

// Weight/bias gradient of the depthwise conv:
//   dW[c][dy][dx] = sum_{b,y,x} dY[b][c][y][x] * X[b][c][y+dy-1][x+dx-1],  db[c] = sum dY
// Grid (C, splits): a block walks its share of the strips of channel c over all
// samples, 10 running sums per thread, then one block reduction; partial rows
// [split][C][10] are added by a second kernel in fixed order.
template <bool VEC>
__global__ void __launch_bounds__(256)
dwconv_bwd_weight_kernel(const float* __restrict__ x, int64_t x_bstride, const float* __restrict__ dy, int64_t dy_bstride,
                         float* __restrict__ partial, int B, int C, int H, int W, int strips_x, int strips_y) {
    __shared__ float red[4][10];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x;
    const int64_t per_plane = (int64_t)strips_x * strips_y;
    const int64_t total = per_plane * B;
    const int64_t hw = (int64_t)H * W;
    float acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.f;
    // keep whole waves in the loop so the cross-lane halo exchange stays valid
    const int64_t stride = (int64_t)gridDim.y * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.y * blockDim.x; base < total; base += stride) {
        const int64_t idx = base + threadIdx.x;
        Strip s = decode_strip(idx, total, strips_x, strips_y);      // plane index here = sample
        const int b = (int)s.plane;
        const float* xp = x + (int64_t)b * x_bstride + (int64_t)c * hw;
        const float* gp = dy + (int64_t)b * dy_bstride + (int64_t)c * hw;
        float in[DW_ROWS + 2][6];
#pragma unroll
        for (int r = 0; r < DW_ROWS + 2; ++r) load_row6<VEC>(in[r], xp, s.y0 - 1 + r, s.x0, H, W, s.active, lane);
        if (s.active) {
#pragma unroll
            for (int r = 0; r < DW_ROWS; ++r) {
                const int yy = s.y0 + r;
                if (yy < H) {
                    float gv[4] = {0.f, 0.f, 0.f, 0.f};
                    const float* grow = gp + (int64_t)yy * W + s.x0;
                    if constexpr (VEC) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(grow);
                        gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
                    } else {
                        gv[0] = grow[0];
                        if (s.x0 + 1 < W) gv[1] = grow[1];
                        if (s.x0 + 2 < W) gv[2] = grow[2];
                        if (s.x0 + 3 < W) gv[3] = grow[3];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = fmaf(gv[j], in[r + ky][j + kx], acc[ky * 3 + kx]);
                        acc[9] += gv[j];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const float s = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        partial[((int64_t)blockIdx.y * C + c) * 10 + threadIdx.x] = s;
    }
}

__global__ void dwconv_bwd_weight_reduce_kernel(const float* __restrict__ partial, int splits, int C,
                                                float* __restrict__ dweight, float* __restrict__ dbias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;     // over C*10
    if (i >= C * 10) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += partial[(int64_t)k * C * 10 + i];
    const int c = i / 10, t = i % 10;
    if (t < 9) dweight[c * 9 + t] = s;
    else if (dbias != nullptr) dbias[c] = s;
}

// ---- launchers ------------------------------------------------------------------
static bool dw_vec_ok(int W, int64_t bs1, int64_t bs2, const void* p1, const void* p2) {
    return (W % 4 == 0) && (bs1 % 4 == 0) && (bs2 % 4 == 0) && aligned16(p1) && aligned16(p2);
}

int dwconv_fwd_launch(const DwArgs& a, hipStream_t stream) {
    RST_REQUIRE(a.B > 0 && a.C > 0 && a.H > 0 && a.W > 0, "dwconv3x3: empty problem");
    RST_REQUIRE(a.x && a.w && a.y, "dwconv3x3: null operand");
    const int sx = (int)cdiv(a.W, 4), sy = (int)cdiv(a.H, DW_ROWS);
    const int64_t total = (int64_t)a.B * a.C * sx * sy;
    const int64_t grid = cdiv(total, 256);
    RST_REQUIRE(grid < (1ll << 31), "dwconv3x3: grid too large");
    ProfScope prof(stream, "dwconv3x3", 18.0 * a.B * a.C * a.H * a.W, 2.0 * 4 * a.B * a.C * a.H * a.W);
    if (dw_vec_ok(a.W, a.x_bstride, a.y_bstride, a.x, a.y))
        hipLaunchKernelGGL(dwconv_fwd_kernel<true>, dim3((unsigned)grid), dim3(256), 0, stream, a, sx, sy, total);
    else
        hipLaunchKernelGGL(dwconv_fwd_kernel<false>, dim3((unsigned)grid), dim3(256), 0, stream, a, sx, sy, total);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

int dwconv_gate_fwd_launch(const GateArgs& a, hipStream_t stream) {
    RST_REQUIRE(a.B > 0 && a.hidden > 0 && a.H > 0 && a.W > 0, "dwconv_gate: empty problem");
    const int sx = (int)cdiv(a.W, 4), sy = (int)cdiv(a.H, DW_ROWS);
    const int64_t total = (int64_t)a.B * a.hidden * sx * sy;
    const int64_t grid = cdiv(total, 256);
    RST_REQUIRE(grid < (1ll << 31), "dwconv_gate: grid too large");
    ProfScope prof(stream, "dwconv3x3_gate", 46.0 * a.B * a.hidden * a.H * a.W,
                   4.0 * a.B * a.hidden * a.H * a.W * (3.0 + (a.d ? 2.0 : 0.0)));
    const bool vec = dw_vec_ok(a.W, a.u_bstride, a.g_bstride, a.u, a.gout) &&
                     (a.d == nullptr || (aligned16(a.d) && a.d_bstride % 4 == 0));
    if (vec) hipLaunchKernelGGL(dwconv_gate_fwd_kernel<true>, dim3((unsigned)grid), dim3(256), 0, stream, a, sx, sy, total);
    else     hipLaunchKernelGGL(dwconv_gate_fwd_kernel<false>, dim3((unsigned)grid), dim3(256), 0, stream, a, sx, sy, total);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

int gate_bwd_launch(const float* dg, const float* d, float* dd, int B, int hidden, int64_t hw, hipStream_t stream) {
    const int64_t total = (int64_t)B * hidden * hw;
    ProfScope prof(stream, "gate_bwd", 20.0 * total, 4.0 * 5 * total);
    hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, dg, d, dd,
                       (int64_t)hidden * hw, total);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

static int dw_bwd_splits(int B, int C, int H, int W) {
    const int64_t strips = (int64_t)B * cdiv(W, 4) * cdiv(H, DW_ROWS);
    int64_t s = cdiv(2048, C);
    const int64_t max_s = cdiv(strips, 256);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return (int)s;
}
size_t dwconv_bwd_weight_ws_bytes(int B, int C, int H, int W) {
    return (size_t)dw_bwd_splits(B, C, H, W) * C * 10 * sizeof(float);
}
int dwconv_bwd_weight_launch(const float* x, int64_t x_bstride, const float* dy, int64_t dy_bstride,
                             float* dweight, float* dbias, int B, int C, int H, int W,
                             void* ws, size_t ws_bytes, hipStream_t stream) {
    RST_REQUIRE(C <= 65535 * 1 && C > 0, "dwconv_bwd_weight: bad channel count");
    if (ws == nullptr || ws_bytes < dwconv_bwd_weight_ws_bytes(B, C, H, W))
        return fail(RST_ERR_WORKSPACE, "dwconv_bwd_weight: workspace too small");
    const int splits = dw_bwd_splits(B, C, H, W);
    const int sx = (int)cdiv(W, 4), sy = (int)cdiv(H, DW_ROWS);
    float* partial = static_cast<float*>(ws);
    dim3 grid(C, splits);
    ProfScope prof(stream, "dwconv3x3_bwd_weight", 20.0 * B * C * H * W, 2.0 * 4 * B * C * H * W);
    if (dw_vec_ok(W, x_bstride, dy_bstride, x, dy))
        hipLaunchKernelGGL(dwconv_bwd_weight_kernel<true>, grid, dim3(256), 0, stream, x, x_bstride, dy, dy_bstride, partial, B, C, H, W, sx, sy);
    else
        hipLaunchKernelGGL(dwconv_bwd_weight_kernel<false>, grid, dim3(256), 0, stream, x, x_bstride, dy, dy_bstride, partial, B, C, H, W, sx, sy);
    RST_LAUNCH_CHECK();
    hipLaunchKernelGGL(dwconv_bwd_weight_reduce_kernel, dim3((unsigned)cdiv(C * 10, 256)), dim3(256), 0, stream,
                       partial, splits, C, dweight, dbias);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

int conv3x3_fewout_launch(const Conv3Args& a, hipStream_t stream) {
    RST_REQUIRE(a.x && a.w && a.y, "conv3x3: null operand");
    RST_REQUIRE(a.B > 0 && a.Cin > 0 && a.H > 0 && a.W > 0, "conv3x3: empty shape");
    RST_REQUIRE(a.Cout >= 1 && a.Cout <= 4, "conv3x3: this kernel serves 1..4 output channels, got %d", a.Cout);
    RST_REQUIRE(a.up_src == nullptr || (a.H % 2 == 0 && a.W % 2 == 0), "conv3x3: the x2 residual needs even H, W");
    const int strips_x = (int)cdiv(a.W, 4), strips_y = (int)cdiv(a.H, DW_ROWS);
    const int64_t total = (int64_t)a.B * strips_x * strips_y;
    const bool vec = (a.W % 4 == 0) && aligned16(a.x) && aligned16(a.y);
    dim3 grid((unsigned)cdiv(total, 256)), blk(256);
    const double px = (double)a.B * a.H * a.W;
    ProfScope prof(stream, "conv3x3_fewout", 18.0 * a.Cin * a.Cout * px, 4.0 * px * (a.Cin + a.Cout * (a.res ? 2.0 : 1.0)));
#define C3_CASE(N)                                                                                              \
    case N:                                                                                                     \
        if (vec) hipLaunchKernelGGL((conv3x3_fewout_kernel<N, true>), grid, blk, 0, stream, a, strips_x, strips_y, total);  \
        else     hipLaunchKernelGGL((conv3x3_fewout_kernel<N, false>), grid, blk, 0, stream, a, strips_x, strips_y, total); \
        break;
    switch (a.Cout) { C3_CASE(1) C3_CASE(2) C3_CASE(3) C3_CASE(4) }
#undef C3_CASE
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst
