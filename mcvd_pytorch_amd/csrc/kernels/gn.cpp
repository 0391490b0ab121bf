// GroupNorm statistics -> per-(sample, channel) affine coefficients (A, B) with y = A*x + B, so the consumer
// conv applies normalisation + temb scale/shift (+ SiLU) while it stages its input tile: the normalised tensor is
// never written to HBM.  One workgroup per (sample, group), ONE pass over the group's data (pilot-shifted moments).
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

    // Single pass over the group: sums of (x - p) and (x - p)^2 around a pilot value p (the group's first element), which
    // keeps the fp32 variance free of the E[x^2] - mean^2 cancellation while reading the tensor from HBM exactly once.
    const float pilot = chan_ptr(c0)[0];
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    int i = threadIdx.x;
    for (; i + 256 < n4; i += 512) {                      // two independent float4 streams per thread
        const int cla = i / HW4, clb = (i + 256) / HW4;
        const float4 va = *reinterpret_cast<const float4*>(chan_ptr(c0 + cla) + (i - cla * HW4) * 4);
        const float4 vb = *reinterpret_cast<const float4*>(chan_ptr(c0 + clb) + (i + 256 - clb * HW4) * 4);
        const float ax = va.x - pilot, ay = va.y - pilot, az = va.z - pilot, aw = va.w - pilot;
        const float bx = vb.x - pilot, by = vb.y - pilot, bz = vb.z - pilot, bw = vb.w - pilot;
        s0 += (ax + ay) + (az + aw);
        s1 += (bx + by) + (bz + bw);
        q0 += (ax * ax + ay * ay) + (az * az + aw * aw);
        q1 += (bx * bx + by * by) + (bz * bz + bw * bw);
    }
    if (i < n4) {
        const int cl = i / HW4;
        const float4 v = *reinterpret_cast<const float4*>(chan_ptr(c0 + cl) + (i - cl * HW4) * 4);
        const float ax = v.x - pilot, ay = v.y - pilot, az = v.z - pilot, aw = v.w - pilot;
        s0 += (ax + ay) + (az + aw);
        q0 += (ax * ax + ay * ay) + (az * az + aw * aw);
    }
    const float inv_n = 1.0f / (float)(gs * HW);
    const float ds = block_sum_256(s0 + s1, red) * inv_n;            // E[x - p]
    const float dq = block_sum_256(q0 + q1, red) * inv_n;            // E[(x - p)^2]
    const float mean = pilot + ds;
    const float var = fmaxf(dq - ds * ds, 0.0f);
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

// One wave per (sample, group).  Partials (sum_i, M2_i, n_i) -> N = sum n_i, mean = sum sum_i / N,
// M2 = sum (M2_i + n_i (sum_i / n_i - mean)^2): exact identities, every term non-negative (no cancellation).
// The kernel is pure latency (a few hundred bytes per wave, 65 launches per forward): every global load -- the partials (kept in
// registers for both passes, up to GF_KEEP per lane) and the affine parameters of the lane's channel -- is requested before the first
// reduction, one memory round trip per launch instead of three.
constexpr int GF_KEEP = 8;
__global__ __launch_bounds__(64) void gn_finalize_kernel(GnArgs a, const float* st0, int np0, const float* st1, int np1) {
    const int C = a.C0 + a.C1;
    const int gs = C / a.groups;
    const int b = blockIdx.x / a.groups;
    const int g = blockIdx.x - b * a.groups;
    const int c0 = g * gs;
    const int lane = threadIdx.x;
    // channels of the group that live in source 0 / source 1
    const int n_in0 = max(0, min(c0 + gs, a.C0) - c0), n_in1 = gs - n_in0;
    const int P0 = n_in0 * np0, P = P0 + n_in1 * np1;
    const float n0 = (float)(a.HW / max(np0, 1)), n1 = (float)(a.HW / max(np1, 1));
    auto part = [&](int i, float& sum, float& m2, float& n) {
        if (i < P0) {
            const int cl = i / np0, p = i - cl * np0;
            const float2 q = *reinterpret_cast<const float2*>(st0 + (((long)b * a.C0 + c0 + cl) * np0 + p) * 2);
            sum = q.x; m2 = q.y; n = n0;
        } else {
            const int j = i - P0;
            const int cl = j / np1, p = j - cl * np1;
            const int c1 = max(c0, a.C0) - a.C0 + cl;
            const float2 q = *reinterpret_cast<const float2*>(st1 + (((long)b * a.C1 + c1) * np1 + p) * 2);
            sum = q.x; m2 = q.y; n = n1;
        }
    };
    float ks[GF_KEEP], km[GF_KEEP], kn[GF_KEEP];
#pragma unroll
    for (int k = 0; k < GF_KEEP; ++k) {
        ks[k] = 0.0f; km[k] = 0.0f; kn[k] = 1.0f;
        if (lane + 64 * k < P) part(lane + 64 * k, ks[k], km[k], kn[k]);
    }
    // affine parameters of channel c0 + lane (clamped: the loads are unconditional, the stores are not)
    const int cpar = c0 + min(lane, gs - 1);
    float par0 = 1.0f, par1 = 0.0f;
    if (a.mode == 1) {                // (1 + scale) * norm + shift        layerspp.py:523,535
        const float* e = a.p0 + (long)b * a.emb_stride + a.emb_off;
        par0 = 1.0f + e[cpar];
        par1 = e[C + cpar];
    } else if (a.mode == 2) {         // weight * norm + bias              torch GroupNorm affine
        par0 = a.p0[cpar];
        par1 = a.p1[cpar];
    }
    float tot = 0.0f;
#pragma unroll
    for (int k = 0; k < GF_KEEP; ++k)
        if (lane + 64 * k < P) tot += ks[k];
    for (int i = lane + 64 * GF_KEEP; i < P; i += 64) {
        float sm, m2, n;
        part(i, sm, m2, n);
        tot += sm;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    const float N = (float)gs * (float)a.HW;
    const float mean = tot / N;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < GF_KEEP; ++k)
        if (lane + 64 * k < P) {
            const float d = ks[k] / kn[k] - mean;
            acc += km[k] + kn[k] * d * d;
        }
    for (int i = lane + 64 * GF_KEEP; i < P; i += 64) {
        float sm, m2, n;
        part(i, sm, m2, n);
        const float d = sm / n - mean;
        acc += m2 + n * d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const float var = acc / N;
    const float rstd = 1.0f / sqrtf(var + a.eps);
    if (lane < gs) {
        reinterpret_cast<float2*>(a.coef)[(long)b * C + cpar] = make_float2(rstd * par0, par1 - mean * rstd * par0);
    }
    for (int cl = lane + 64; cl < gs; cl += 64) {       // (groups of more than 64 channels: not a shape of the reference's configs)
        const int c = c0 + cl;
        float A, Bc;
        if (a.mode == 1) {
            const float* e = a.p0 + (long)b * a.emb_stride + a.emb_off;
            const float sc = 1.0f + e[c];
            A = rstd * sc;
            Bc = e[C + c] - mean * rstd * sc;
        } else if (a.mode == 2) {
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

int launch_gn_finalize(const GnArgs& a, const float* st0, int np0, const float* st1, int np1, hipStream_t s) {
    const int C = a.C0 + a.C1;
    MCVD_REQUIRE(a.groups > 0 && C % a.groups == 0, "gn: %d channels not divisible by %d groups", C, a.groups);
    MCVD_REQUIRE(st0 && np0 > 0 && a.HW % np0 == 0 && (a.C1 == 0 || (st1 && np1 > 0 && a.HW % np1 == 0)),
                 "gn_finalize: bad partial statistics (np0=%d np1=%d HW=%d)", np0, np1, a.HW);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(a.B * a.groups), dim3(64), 0, s, a, st0, np0, st1, a.C1 ? np1 : 1);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- second pass of a K-split Winograd layer + the GroupNorm finalize of the norm over its output, in ONE launch (round 5).  The reduce
// pass (conv_wino.cpp: wino_ksplit_reduce_stats_kernel) walks the tensor flat; here a workgroup owns one (sample, group) slab --
// gs channel planes of 8 x 8 or 16 x 16 pixels -- so it holds every partial the group's statistics need and no workgroup waits for
// another: y and the per-plane partials exactly as the reduce pass writes them (same expressions, same order: later concat norms read
// the partials), then gn_finalize_kernel's reduction over the gs planes in wave 0 (same expressions: the table is bit-identical to the
// two-launch path).  Saves the gn_finalize launch behind 8 x 8 / 16 x 16 layers (4.3 us each inside the graph, r05_tail_launch_bound.txt).
typedef float gnf32x4 __attribute__((ext_vector_type(4)));
template <int PL>
__global__ __launch_bounds__(256) void ksplit_reduce_gn_kernel(const float* part, const float* bias, const float* res, float scale, float* y,
                                                               long half_stride, int Cout, float* stats, int ksp, GnOut g, int HW) {
    __shared__ float2 sp[64];
    const int gs = Cout / g.groups;
    const int b = blockIdx.x / g.groups, gi = blockIdx.x - b * g.groups;
    const int c0 = gi * gs;
    const long base4 = ((long)b * Cout + c0) * PL;
    const int n4 = gs * PL;
    for (int j0 = 0; j0 < n4; j0 += 256) {
        const int j = j0 + threadIdx.x;
        const bool in = j < n4;
        const long i = base4 + j;
        gnf32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (in) {
            gnf32x4 acc = reinterpret_cast<const gnf32x4*>(part)[i];
            for (int k = 1; k < ksp; ++k) acc = acc + reinterpret_cast<const gnf32x4*>(part + k * half_stride)[i];      // p0 + p1 + ... in this fixed order
            const float bv = bias[c0 + j / PL];
            v = acc + bv;
            if (res) v = v + reinterpret_cast<const gnf32x4*>(res)[i];
            v = v * scale;
            reinterpret_cast<gnf32x4*>(y)[i] = v;
        }
        float sm = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
        for (int o = PL / 2; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
        const float mu = sm * (1.0f / (4 * PL));
        const float d0 = v[0] - mu, d1 = v[1] - mu, d2 = v[2] - mu, d3 = v[3] - mu;
        float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
        for (int o = PL / 2; o > 0; o >>= 1) m2 += __shfl_xor(m2, o);
        if (in && (threadIdx.x & (PL - 1)) == 0) {
            const int plane = j / PL;
            stats[((long)b * Cout + c0 + plane) * 2] = sm;
            stats[((long)b * Cout + c0 + plane) * 2 + 1] = m2;
            sp[plane] = make_float2(sm, m2);
        }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    // gn_finalize_kernel with one source, one partial per channel: P = gs <= 64 partials, partial `lane` in lane `lane`
    const int lane = threadIdx.x;
    const int P = gs;
    float ks0 = 0.0f, km0 = 0.0f, kn0 = 1.0f;
    if (lane < P) { const float2 q = sp[lane]; ks0 = q.x; km0 = q.y; kn0 = (float)HW; }
    const int cpar = c0 + min(lane, gs - 1);
    float par0 = 1.0f, par1 = 0.0f;
    if (g.mode == 1) {
        const float* e = g.p0 + (long)b * g.emb_stride + g.emb_off;
        par0 = 1.0f + e[cpar];
        par1 = e[Cout + cpar];
    } else if (g.mode == 2) {
        par0 = g.p0[cpar];
        par1 = g.p1[cpar];
    }
    float tot = 0.0f;
    if (lane < P) tot += ks0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    const float N = (float)gs * (float)HW;
    const float mean = tot / N;
    float acc = 0.0f;
    if (lane < P) {
        const float d = ks0 / kn0 - mean;
        acc += km0 + kn0 * d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const float var = acc / N;
    const float rstd = 1.0f / sqrtf(var + g.eps);
    if (lane < gs) reinterpret_cast<float2*>(g.coef)[(long)b * Cout + cpar] = make_float2(rstd * par0, par1 - mean * rstd * par0);
}

bool ksplit_reduce_gn_usable(const ConvArgs& a) {
    const int hw4 = a.H * a.W / 4;
    return a.gno.coef && a.stats && a.part && (hw4 == 16 || hw4 == 64) && a.gno.groups > 0 && a.Cout % a.gno.groups == 0 &&
           a.Cout / a.gno.groups <= 64;
}

int launch_ksplit_reduce_gn(const ConvArgs& a, hipStream_t s) {
    MCVD_REQUIRE(ksplit_reduce_gn_usable(a), "ksplit_reduce_gn: unsupported (Cout=%d groups=%d HW=%d)", a.Cout, a.gno.groups, a.H * a.W);
    const int ksp = a.ksplit >= 2 ? a.ksplit : 2;
    const long n = (long)a.B * a.Cout * a.H * a.W;
    const int hw4 = a.H * a.W / 4;
    if (hw4 == 16)
        hipLaunchKernelGGL(ksplit_reduce_gn_kernel<16>, dim3(a.B * a.gno.groups), dim3(256), 0, s, a.part, a.bias, a.res, a.out_scale, a.y, n, a.Cout,
                           a.stats, ksp, a.gno, a.H * a.W);
    else
        hipLaunchKernelGGL(ksplit_reduce_gn_kernel<64>, dim3(a.B * a.gno.groups), dim3(256), 0, s, a.part, a.bias, a.res, a.out_scale, a.y, n, a.Cout,
                           a.stats, ksp, a.gno, a.H * a.W);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
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
