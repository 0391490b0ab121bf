// Time embedding: sinusoid -> Linear -> SiLU -> Linear (ncsnpp_more.py:273-280, layers.py:504-518), followed by the
// SiLU every Dense_0 applies to its input (layerspp.py:521); then ALL Dense_0 projections of the network in one
// GEMM-shaped launch (58 of them for a 4-level net), against a [K][N]-transposed concatenated weight matrix.
#include "../common.h"

namespace mcvd {

__device__ __forceinline__ float silu_t(float v) { return v / (1.0f + expf(-v)); }

// one workgroup per sample; hidden width T = 4*nf <= 1024 (nf <= 256)
__global__ __launch_bounds__(256) void temb_mlp_kernel(const void* labels, int labels_f32, const float* freqs, const float* w0,
                                                        const float* b0, const float* w1, const float* b1,
                                                        float* silu_temb, int nf, int out_stride, const float* emb_table,
                                                        const int32_t* mask) {
    __shared__ float emb[256];
    __shared__ float hid[1024];
    const int b = blockIdx.x;
    const int T = 4 * nf;
    const int half = nf / 2;
    // timesteps.float(): int64 labels (samplers) or already-float, possibly fractional ones (F-PNDM's (t + t_next) / 2)
    const float t = labels_f32 ? static_cast<const float*>(labels)[b] : (float)static_cast<const int64_t*>(labels)[b];
    for (int i = threadIdx.x; i < nf; i += 256) {
        float v = 0.0f;                                     // odd nf: zero pad (layers.py:515-516)
        if (i < half) v = sinf(t * freqs[i]);
        else if (i < 2 * half) v = cosf(t * freqs[i - half]);
        emb[i] = v;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < T; o += 256) {
        const float* w = w0 + (long)o * nf;                 // nn.Linear weight [out][in]
        float acc = 0.0f;
        for (int k = 0; k < nf; ++k) acc = fmaf(emb[k], w[k], acc);
        hid[o] = silu_t(acc + b0[o]);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < T; o += 256) {
        const float* w = w1 + (long)o * T;
        float acc = 0.0f;
        for (int k = 0; k < T; ++k) acc = fmaf(hid[k], w[k], acc);
        silu_temb[(long)b * out_stride + o] = silu_t(acc + b1[o]);
    }
    if (emb_table) {        // cond_emb: temb = cat([temb, Embedding(cond_mask)]) (ncsnpp_more.py:282-285); Dense_0 sees SiLU of it
        const int mb = mask ? (mask[b] != 0 ? 1 : 0) : 1;
        for (int j = threadIdx.x; j < half; j += 256) silu_temb[(long)b * out_stride + T + j] = silu_t(emb_table[mb * half + j]);
    }
}

int launch_temb_mlp(const void* labels, int labels_f32, const float* freqs, const float* w0, const float* b0, const float* w1,
                    const float* b1, float* silu_temb, int B, int nf, int out_stride, const float* emb_table, const int32_t* mask,
                    hipStream_t s) {
    MCVD_REQUIRE(nf <= 256 && nf >= 4, "temb: ngf=%d out of range [4,256]", nf);
    hipLaunchKernelGGL(temb_mlp_kernel, dim3(B), dim3(256), 0, s, labels, labels_f32, freqs, w0, b0, w1, b1, silu_temb, nf,
                       out_stride, emb_table, mask);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

// out[b][n] = bias[n] + sum_k act[b][k] * wt[k][n].  Block: 256 outputs n x 8 samples; act rows staged in LDS.
__global__ __launch_bounds__(256) void dense_all_kernel(const float* act, const float* wt, const float* bias, float* out,
                                                         int B, int K, int N) {
    __shared__ float sa[8 * 1152];
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int b0 = blockIdx.y * 8;
    const int nb = (B - b0) < 8 ? (B - b0) : 8;
    for (int i = threadIdx.x; i < 8 * K; i += 256) {
        const int bb = i / K;
        sa[i] = (bb < nb) ? act[(long)(b0 + bb) * K + (i - bb * K)] : 0.0f;
    }
    __syncthreads();
    if (n >= N) return;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float w = wt[(long)k * N + n];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(sa[j * K + k], w, acc[j]);
    }
    const float bv = bias[n];
    for (int j = 0; j < nb; ++j) out[(long)(b0 + j) * N + n] = acc[j] + bv;
}

int launch_dense_all(const float* act, const float* wt, const float* bias, float* out, int B, int K, int N,
                     hipStream_t s) {
    MCVD_REQUIRE(K <= 1152, "dense_all: K=%d > 1152", K);
    hipLaunchKernelGGL(dense_all_kernel, dim3((N + 255) / 256, (B + 7) / 8), dim3(256), 0, s, act, wt, bias, out, B, K, N);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

// wt[c][col_off + r] = w[r][c]   (w: [rows][cols] row-major; wt leading dimension ld_out)
__global__ void transpose_into_kernel(const float* w, float* wt, int rows, int cols, int ld_out, int col_off) {
    const long n = (long)rows * cols;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols), r = (int)(i / cols);
        wt[(long)c * ld_out + col_off + r] = w[i];
    }
}

int launch_transpose_into(const float* w, float* wt, int rows, int cols, int ld_out, int col_off, hipStream_t s) {
    const long n = (long)rows * cols;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(transpose_into_kernel, dim3(blocks), dim3(256), 0, s, w, wt, rows, cols, ld_out, col_off);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mcvd
