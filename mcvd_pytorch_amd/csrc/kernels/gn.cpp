// GroupNorm statistics -> per-(sample, channel) affine coefficients (A, B) with y = A*x + B, so the consumer
// conv applies normalisation + temb scale/shift (+ SiLU) while it stages its input tile: the normalised tensor is
// never written to HBM.  One workgroup per (sample, group); two passes over the group's data (the second pass
// hits L2): mean first, then the centred sum of squares, which keeps the fp32 variance accurate.
// Input may be a virtual channel concat of two tensors (UNet up path: cat([h, skip])).
#include "../common.h"

namespace mcvd {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void gn_coef_kernel(GnArgs a) {
    __shared__ float red[4];
    const int C = a.C0 + a.C1;
    const int gs = C / a.groups;
    const int b = blockIdx.x / a.groups;
    const int g = blockIdx.x - b * a.groups;
    const int HW = a.HW;
    const int HW4 = HW >> 2;
    const int n4 = gs * HW4;
    const int c0 = g * gs;

    auto chan_ptr = [&](int c) -> const float* {
        return (c < a.C0) ? a.x0 + ((long)b * a.C0 + c) * HW : a.x1 + ((long)b * a.C1 + (c - a.C0)) * HW;
    };

    float s = 0.0f;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const int cl = i / HW4;
        const int p4 = i - cl * HW4;
        const float4 v = *reinterpret_cast<const float4*>(chan_ptr(c0 + cl) + p4 * 4);
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float inv_n = 1.0f / (float)(gs * HW);
    const float mean = block_sum_256(s, red) * inv_n;

    float q = 0.0f;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const int cl = i / HW4;
        const int p4 = i - cl * HW4;
        const float4 v = *reinterpret_cast<const float4*>(chan_ptr(c0 + cl) + p4 * 4);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    const float var = block_sum_256(q, red) * inv_n;
    const float rstd = 1.0f / sqrtf(var + a.eps);

    for (int cl = threadIdx.x; cl < gs; cl += 256) {
        const int c = c0 + cl;
        float A, Bc;
        if (a.mode == 1) {            // (1 + scale) * norm + shift        layerspp.py:523,535
            const float* e = a.p0 + (long)b * a.emb_stride + a.emb_off;
            const float sc = 1.0f + e[c];
            A = rstd * sc;
            Bc = e[C + c] - mean * rstd * sc;
        } else if (a.mode == 2) {     // weight * norm + bias              torch GroupNorm affine
            A = rstd * a.p0[c];
            Bc = a.p1[c] - mean * rstd * a.p0[c];
        } else {
            A = rstd;
            Bc = -mean * rstd;
        }
        a.coef[((long)b * C + c) * 2] = A;
        a.coef[((long)b * C + c) * 2 + 1] = Bc;
    }
}

int launch_gn_coef(const GnArgs& a, hipStream_t s) {
    const int C = a.C0 + a.C1;
    MCVD_REQUIRE(a.groups > 0 && C % a.groups == 0, "gn: %d channels not divisible by %d groups", C, a.groups);
    MCVD_REQUIRE(a.HW % 4 == 0, "gn: HW=%d must be a multiple of 4", a.HW);
    hipLaunchKernelGGL(gn_coef_kernel, dim3(a.B * a.groups), dim3(256), 0, s, a);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mcvd
