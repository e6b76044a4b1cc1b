// Per-channel sums over samples and pixels: out[r] = sum_b sum_n x[b][r][n].
// Used for the bias gradients of the convolutions (autograd of nn.Conv2d bias,
// model/Restromer.py:85-97,115-125 with bias=True).  Two deterministic stages:
// one block per (row, sample), then a fixed-order sum over samples.
#include "kernels.hpp"

namespace rst {

__global__ void __launch_bounds__(256)
rowsum_partial_kernel(const float* __restrict__ x, int64_t x_bstride, float* __restrict__ partial, int rows, int64_t N) {
    __shared__ float red[4];
    const int r = blockIdx.x, b = blockIdx.y;
    const float* p = x + (int64_t)b * x_bstride + (int64_t)r * N;
    float s = 0.f;
    for (int64_t n = threadIdx.x; n < N; n += 256) s += p[n];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)b * rows + r] = red[0] + red[1] + red[2] + red[3];
}

__global__ void rowsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int rows, int B) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += partial[(int64_t)b * rows + r];
    out[r] = s;
}

size_t rowsum_ws_bytes(int B, int rows, int64_t N) {
    (void)N;
    return (size_t)B * rows * sizeof(float);
}

int rowsum_launch(const float* x, int64_t x_bstride, float* out, int B, int rows, int64_t N,
                  void* ws, size_t ws_bytes, hipStream_t stream) {
    RST_REQUIRE(B > 0 && rows > 0 && N > 0 && B <= 65535, "rowsum: bad shape");
    if (ws == nullptr || ws_bytes < rowsum_ws_bytes(B, rows, N)) return fail(RST_ERR_WORKSPACE, "rowsum: workspace too small");
    float* partial = static_cast<float*>(ws);
    ProfScope prof(stream, "rowsum", (double)B * rows * N, 4.0 * B * rows * N);
    hipLaunchKernelGGL(rowsum_partial_kernel, dim3(rows, B), dim3(256), 0, stream, x, x_bstride, partial, rows, N);
    RST_LAUNCH_CHECK();
    hipLaunchKernelGGL(rowsum_final_kernel, dim3((unsigned)cdiv(rows, 128)), dim3(128), 0, stream, partial, out, rows, B);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst
