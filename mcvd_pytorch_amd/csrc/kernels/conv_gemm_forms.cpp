// Two 3x3 convs of the network have almost no channels on one side -- the stem (5 + 5 frame channels -> ngf) and the last conv (ngf -> 5
// or 15 frame channels) -- and the 3x3 kernels serve them badly: the direct fp32 kernel pads the stem's 10 input channels to 16 and runs
// on the fp32 matrix pipe (110 us for a launch whose traffic is worth 20), the Winograd kernel stages all ngf input channels of the last
// conv for a padded 32-cout tile (106 us).  Both are run as 1x1 GEMMs on the three-piece bf16 kernel (conv1x1_h2.cpp) instead
// (ncsnpp_more.py: the first conv3x3 of all_modules and the last one; layerspp.py conv3x3 = layers.py:107-113):
//   shape id 23, "im2col":  col[b][c * 9 + t][p] = x[b][c][p + offset(t)] (zero outside the image; rows up to a multiple of 16 zero), then
//                           y = W' col + bias with W'[co][c * 9 + t] = W[co][c][t]  -- the GEMM's epilogue (residual, scale, GroupNorm
//                           partials) is the conv's;
//   shape id 22, "taps as outputs":  z[b][t * Cout + co][p] = sum_c W[co][c][t] x'[b][c][p]  (x' = the conv's prologue-transformed input:
//                           the GEMM's prologue is the conv's), then  y[b][co][p] = bias[co] + sum_t z[b][t * Cout + co][p + offset(t)]
//                           with z read as zero outside the image (the conv pads x' with zeros, and z is linear in x').
// The products are the same fp32-equivalent six piece products; the summation order differs from the 3x3 kernels' (tolerance, not bits).
#include "../common.h"

namespace mcvd {

// one thread per (b, k, 4 pixels): K rows of `col` per sample, rows >= 9 * (C0 + C1) are zero.  blockIdx.y = b * K + k (no 64-bit division per thread)
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ x0, int C0, const float* __restrict__ x1, int C1, float* __restrict__ col,
                                                        int H, int W, int K) {
    const int W4 = W >> 2, HW = H * W;
    const int k = blockIdx.y % K, b = blockIdx.y / K;
    const int c = k / 9, t = k - 9 * c;
    const bool real = c < C0 + C1;
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    const float* plane = !real ? nullptr : (c < C0 ? x0 + ((long)b * C0 + c) * HW : x1 + ((long)b * C1 + (c - C0)) * HW);
    float* dst = col + ((long)b * K + k) * HW;
    // four float4 per thread, all loads in front of the stores: with one per thread the launch ran at 2.9 TB/s (8 workgroups x 256 x 16
    // bytes in flight per CU for a round trip each)
    for (int i0 = blockIdx.x * 1024 + threadIdx.x; i0 < H * W4; i0 += gridDim.x * 1024) {
        float4 o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 256;
            const int y = i / W4, xq = (i - y * W4) * 4;
            o[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int yy = y + dy;
            if (real && i < H * W4 && yy >= 0 && yy < H) {
                const float* row = plane + yy * W;
                const float4 q = *reinterpret_cast<const float4*>(row + xq);
                if (dx == 0) o[u] = q;
                else if (dx < 0) o[u] = make_float4(xq > 0 ? row[xq - 1] : 0.0f, q.x, q.y, q.z);
                else o[u] = make_float4(q.y, q.z, q.w, xq + 4 < W ? row[xq + 4] : 0.0f);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 256;
            if (i < H * W4) *reinterpret_cast<float4*>(dst + (long)i * 4) = o[u];
        }
    }
}

int launch_im2col3x3(const float* x0, int C0, const float* x1, int C1, float* col, int B, int H, int W, int K, hipStream_t s) {
    MCVD_REQUIRE(W % 4 == 0 && K >= 9 * (C0 + C1) && K <= 65535, "im2col: W=%d K=%d", W, K);
    const int per_plane = H * (W / 4), bmax = 65535 / K;
    const long HW = (long)H * W;
    for (int b0 = 0; b0 < B; b0 += bmax) {              // (grid.y holds 65535 planes)
        const int nb = B - b0 < bmax ? B - b0 : bmax;
        hipLaunchKernelGGL(im2col3x3_kernel, dim3((per_plane + 1023) / 1024, nb * K), dim3(256), 0, s, x0 + (long)b0 * C0 * HW,
                           C0, x1 ? x1 + (long)b0 * C1 * HW : nullptr, C1, col + (long)b0 * K * HW, H, W, K);
        MCVD_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

// one thread per (b, co, 4 pixels): nine shifted reads of z (row t * Cout + co of the sample's 9 * Cout rows), summed in tap order behind the
// bias.  blockIdx.y = b * Cout + co
__global__ __launch_bounds__(256) void taps_shift_add_kernel(const float* __restrict__ z, const float* __restrict__ bias, const float* __restrict__ res,
                                                             float scale, float* __restrict__ y, int Cout, int H, int W) {
    const int W4 = W >> 2, HW = H * W;
    const int co = blockIdx.y % Cout, b = blockIdx.y / Cout;
    const float bs = bias[co];
    const float* zb = z + ((long)b * 9 * Cout + co) * HW;
    const long plane = ((long)b * Cout + co) * HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W4; i += gridDim.x * 256) {
        const int yv = i / W4, xq = (i - yv * W4) * 4;
        float acc[4] = {bs, bs, bs, bs};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = yv + t / 3 - 1, dx = t % 3 - 1;
            if (yy < 0 || yy >= H) continue;
            const float* row = zb + (long)t * Cout * HW + yy * W;
            const float4 q = *reinterpret_cast<const float4*>(row + xq);
            float v[4];
            if (dx == 0) { v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
            else if (dx < 0) { v[0] = xq > 0 ? row[xq - 1] : 0.0f; v[1] = q.x; v[2] = q.y; v[3] = q.z; }
            else { v[0] = q.y; v[1] = q.z; v[2] = q.w; v[3] = xq + 4 < W ? row[xq + 4] : 0.0f; }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += v[j];
        }
        const long o = plane + yv * W + xq;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (res) r = *reinterpret_cast<const float4*>(res + o);
        *reinterpret_cast<float4*>(y + o) = make_float4((acc[0] + r.x) * scale, (acc[1] + r.y) * scale, (acc[2] + r.z) * scale, (acc[3] + r.w) * scale);
    }
}

int launch_taps_shift_add(const float* z, const float* bias, const float* res, float scale, float* y, int B, int Cout, int H, int W, hipStream_t s) {
    MCVD_REQUIRE(W % 4 == 0 && Cout <= 65535, "taps_shift_add: W=%d Cout=%d", W, Cout);
    const int per_plane = H * (W / 4), bmax = 65535 / Cout;
    const long HW = (long)H * W;
    for (int b0 = 0; b0 < B; b0 += bmax) {
        const int nb = B - b0 < bmax ? B - b0 : bmax;
        hipLaunchKernelGGL(taps_shift_add_kernel, dim3((per_plane + 255) / 256, nb * Cout), dim3(256), 0, s, z + (long)b0 * 9 * Cout * HW, bias,
                           res ? res + (long)b0 * Cout * HW : nullptr, scale, y + (long)b0 * Cout * HW, Cout, H, W);
        MCVD_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

// [Cout][Cin][3][3] -> the packed fp32 matrix wp[k * CoutP + n] of the GEMM form (the layout launch_pack_conv1x1_h2 splits into pieces):
//   form 23: k = ci * 9 + t, n = co;          form 22: k = ci, n = t * Cout + co.       (wp zeroed by the caller: padding rows / columns)
__global__ void pack_conv_gemm_form_kernel(const float* w, float* wp, int Cout, int Cin, int form, int CoutP) {
    const long n = (long)Cout * Cin * 9;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % 9), ci = (int)((i / 9) % Cin), co = (int)(i / (9L * Cin));
        if (form == 23) wp[((long)ci * 9 + t) * CoutP + co] = w[i];
        else wp[(long)ci * CoutP + t * Cout + co] = w[i];
    }
}

int launch_pack_conv_gemm_form(const float* w, float* wp, int Cout, int Cin, int form, int CoutP, hipStream_t s) {
    MCVD_REQUIRE(form == 22 || form == 23, "pack_conv_gemm_form: form %d", form);
    const long n = (long)Cout * Cin * 9;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_conv_gemm_form_kernel, dim3(blocks), dim3(256), 0, s, w, wp, Cout, Cin, form, CoutP);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mcvd
