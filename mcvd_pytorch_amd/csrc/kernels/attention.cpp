// Multi-head self-attention core (layerspp.py:237-244) as a flash-style kernel on v_mfma_f32_32x32x2_f32:
// QK^T, online softmax and PV in one pass, the [HW x HW] score matrix never leaves registers.
//
// Layout is channel-major ([B, 3C, HW]: q | k | v, heads = contiguous channel chunks of D), which makes every
// MFMA operand a unit-stride read:
//   S^T[key][query] = sum_c K[c][key] * Q[c][query]      A = K tile (LDS, lane = key), B = Q (registers, lane = query)
//   O[c][query]    += sum_key V[c][key] * P[key][query]  A = V tile (LDS, lane = c),   B = P = the S^T accumulator
// Computing S^T (keys on accumulator rows) leaves each lane holding 16 keys of ONE query, so the softmax
// reduction is in-lane plus one exchange between the two half-waves, and the probabilities are already in the
// B-operand layout of the PV product (the K index of PV is permuted identically on both operands).
// One workgroup = 4 waves = 4 x 32 queries of one (sample, head); all waves share the K/V tiles in LDS.
#include <math.h>

#include "../common.h"

namespace mcvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DT>   // head dim D = 32*DT
__global__ __launch_bounds__(256) void attn_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C,
                                                         int heads, int S, float scale) {
    constexpr int D = 32 * DT;
    constexpr int VP = 33;                       // V tile pitch (odd -> conflict-free channel-strided reads)
    extern __shared__ __attribute__((aligned(16))) float smem_attn[];      // dynamic: D = 256 needs 65 KiB
    float* sK = smem_attn;                       // [D][32]
    float* sV = smem_attn + D * 32;              // [D][VP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y;
    const int b = bh / heads, hd = bh - b * heads;
    const float* qb = qkv + ((long)b * 3 * C + hd * D) * S;
    const float* kb = qb + (long)C * S;
    const float* vb = kb + (long)C * S;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bool active = q0 < S;

    // Q fragment: lane (query=l31, half) holds Q[c = 2s+half][q0+l31], s = 0..D/2-1
    float qreg[D / 2];
#pragma unroll
    for (int s = 0; s < D / 2; ++s) qreg[s] = active ? qb[(long)(2 * s + half) * S + q0 + l31] : 0.0f;

    f32x16 o[DT];
#pragma unroll
    for (int ct = 0; ct < DT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.0f;
    float m_run = -1e30f, l_run = 0.0f;

    constexpr int NLD = D * 8 / 256;             // float4 loads per thread per tile (D*32 floats / 4 / 256)
    f32x4 rk[NLD], rv[NLD];
    const int ntiles = S / 32;
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = i * 256 + tid;          // float4 index: row c = e/8, col4 = e%8
            const long off = (long)(e >> 3) * S + t * 32 + (e & 7) * 4;
            rk[i] = *reinterpret_cast<const f32x4*>(kb + off);
            rv[i] = *reinterpret_cast<const f32x4*>(vb + off);
        }
    };
    constexpr bool PREF = DT <= 6;               // the next tile's K/V prefetch lives in 2*NLD float4 across the MFMA phase; beyond
                                                 // D = 192 Q (D/2) + O (D/2) leave no room for it: load just before the LDS write
    if (PREF) gload(0);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();                          // previous tile fully consumed
        if (!PREF) gload(t);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = i * 256 + tid;
            *reinterpret_cast<f32x4*>(sK + e * 4) = rk[i];
            float* dv = sV + (e >> 3) * VP + (e & 7) * 4;
            dv[0] = rv[i][0]; dv[1] = rv[i][1]; dv[2] = rv[i][2]; dv[3] = rv[i][3];
        }
        __syncthreads();
        if (PREF && t + 1 < ntiles) gload(t + 1);

        // ---- S^T tile: 32 keys x 32 queries
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < D / 2; ++s)
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[(2 * s + half) * 32 + l31], qreg[s], st, 0, 0, 0);

        // ---- online softmax over keys (this lane: 16 keys of query l31; partner lane^32 holds the other 16)
        float mt = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] *= scale; mt = fmaxf(mt, st[r]); }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __expf(m_run - m_new);
        float ps = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = __expf(st[r] - m_new); ps += st[r]; }
        l_run = l_run * alpha + ps;               // per-half partial sum; halves are added at the end
        m_run = m_new;
#pragma unroll
        for (int ct = 0; ct < DT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;

        // ---- O += V * P : k-step s pairs key (s&3)+8*(s>>2) [half 0] with the same +4 [half 1], exactly the keys
        //      whose probabilities sit in accumulator register s of the two half-waves.
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int key = (s & 3) + 8 * (s >> 2) + 4 * half;
#pragma unroll
            for (int ct = 0; ct < DT; ++ct)
                o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[(ct * 32 + l31) * VP + key], st[s], o[ct], 0, 0, 0);
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (active) {
        float* ob = out + ((long)b * C + hd * D) * S + q0 + l31;
#pragma unroll
        for (int ct = 0; ct < DT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                ob[(long)c * S] = o[ct][r] * inv;
            }
    }
}

bool attention_mfma_supported(int C, int heads, int HW) {
    if (heads <= 0 || C % heads != 0) return false;
    const int D = C / heads;
    return D % 32 == 0 && D >= 32 && D <= 256 && HW % 32 == 0;
}

int launch_attention_mfma(const float* qkv, float* out, int B, int C, int heads, int HW, hipStream_t s) {
    MCVD_REQUIRE(heads > 0 && C % heads == 0, "attention: C=%d heads=%d", C, heads);
    const int D = C / heads;
    // The flash kernel keeps Q and the O accumulators of a head in registers: head dims 32..256 in steps of 32.  Anything else
    // (n_head_channels = -1 on a wide level -> one head of C channels, layerspp.py:219-228; odd widths; HW not a multiple of 32)
    // runs the general one-thread-per-query kernel: correct for every shape, far slower.
    if (!attention_mfma_supported(C, heads, HW)) return launch_attention_naive(qkv, out, B, C, heads, HW, s);
    const float scale = (float)pow((double)D, -0.5);   // int(C)**-0.5 as a Python double, then fp32 (layerspp.py:239)
    dim3 grid((HW + 127) / 128, B * heads);
    const size_t lds = (size_t)D * (32 + 33) * sizeof(float);
#define ATTN_CASE(DT)                                                                                              \
    case DT: {                                                                                                     \
        if (lds > 64 * 1024) {                                                                                     \
            static PerDeviceOnce raised;                                                                           \
            if (raised.first_use()) {                                                                              \
                MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<DT>),           \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));       \
                raised.done();                                                                                     \
            }                                                                                                      \
        }                                                                                                          \
        hipLaunchKernelGGL(attn_mfma_kernel<DT>, grid, dim3(256), lds, s, qkv, out, C, heads, HW, scale);          \
        break;                                                                                                     \
    }
    switch (D / 32) {
        ATTN_CASE(1) ATTN_CASE(2) ATTN_CASE(3) ATTN_CASE(4) ATTN_CASE(5) ATTN_CASE(6) ATTN_CASE(7) ATTN_CASE(8)
    }
#undef ATTN_CASE
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Naive: one thread per (sample, head, query); three passes (max, sum, weighted V), scores recomputed.
__global__ void attn_naive_kernel(const float* qkv, float* out, int B, int C, int heads, int S, float scale) {
    const int D = C / heads;
    const long n = (long)B * heads * S;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % S);
        const int hd = (int)((i / S) % heads);
        const int b = (int)(i / ((long)S * heads));
        const float* qb = qkv + ((long)b * 3 * C + hd * D) * S;
        const float* kb = qb + (long)C * S;
        const float* vb = kb + (long)C * S;
        float mx = -1e30f;
        for (int k = 0; k < S; ++k) {
            float d = 0.0f;
            for (int c = 0; c < D; ++c) d = fmaf(qb[(long)c * S + q], kb[(long)c * S + k], d);
            mx = fmaxf(mx, d * scale);
        }
        float sum = 0.0f;
        for (int k = 0; k < S; ++k) {
            float d = 0.0f;
            for (int c = 0; c < D; ++c) d = fmaf(qb[(long)c * S + q], kb[(long)c * S + k], d);
            sum += expf(d * scale - mx);
        }
        float* ob = out + ((long)b * C + hd * D) * S + q;
        for (int c = 0; c < D; ++c) ob[(long)c * S] = 0.0f;
        for (int k = 0; k < S; ++k) {
            float d = 0.0f;
            for (int c = 0; c < D; ++c) d = fmaf(qb[(long)c * S + q], kb[(long)c * S + k], d);
            const float p = expf(d * scale - mx) / sum;
            for (int c = 0; c < D; ++c) ob[(long)c * S] += p * vb[(long)c * S + k];
        }
    }
}

int launch_attention_naive(const float* qkv, float* out, int B, int C, int heads, int HW, hipStream_t s) {
    MCVD_REQUIRE(heads > 0 && C % heads == 0, "attention: C=%d heads=%d", C, heads);
    const float scale = (float)pow((double)(C / heads), -0.5);
    const long n = (long)B * heads * HW;
    hipLaunchKernelGGL(attn_naive_kernel, dim3((int)((n + 127) / 128)), dim3(128), 0, s, qkv, out, B, C, heads, HW, scale);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mcvd
