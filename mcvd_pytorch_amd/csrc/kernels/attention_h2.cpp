// Multi-head self-attention core (layerspp.py:237-244), the flash-style kernel of attention.cpp on the FP16 matrix pipe with every
// MFMA operand split into TWO fp16 pieces (conv_wino2h.cpp has the arithmetic: v ~= v1 + v2, 22 significant bits, three piece
// products a1 b2 + a2 b1 + a1 b1 accumulated in fp32).  attention.cpp is bound by the fp32 matrix pipe (MFMA-busy 0.68, waves
// issue-stalled 76 % of the time: profiles/r02_pmc_*_f16x2.txt); three v_mfma_f32_32x32x16_f16 per 16-deep step take 3/16 of its
// pipe time.  Same decomposition: one workgroup = 4 waves = 4 x 32 queries of one (sample, head), key tiles of 32, S^T (keys on the
// accumulator rows) so that the softmax is in-lane and P is already the B operand of the PV product:
//   S^T[key][query] = sum_c K[c][key] * Q[c][query]      A = K tile pieces (LDS, lane = key), B = Q pieces (registers, lane = query)
//   O[c][query]    += sum_key V[c][key] * P[key][query]  A = V tile pieces (LDS, lane = c),   B = P pieces (registers)
// What changes:
//   * Q is scaled by 2^4 and split once per workgroup (the pieces take the registers the fp32 fragment took).
//   * the threads that stage a K / V tile split it (each element once per workgroup, not once per wave) and park the pieces in the
//     MFMA A-operand order [step][piece][64 lanes][4 dwords]: one conflict-free ds_read_b128 per operand.  K-slot conventions:
//       S product : lane half h, element e of step st  <->  channel 16 st + 2e + h              (dword j = channels 4j + h, 4j + 2 + h)
//       PV product: lane half h, element e of step s2  <->  key (r&3) + 8 (r>>2) + 4h, r = 8 s2 + e  -- the key whose probability
//                   sits in accumulator register r of the S^T tile; a dword = two consecutive keys.
//   * the probabilities are produced times 2^12 (folded into the exponent: exp(s - m + 12 ln 2); the row sums carry the same
//     factor and it cancels), so their second pieces stay clear of the fp16 denormal range; K and V are staged times 2^4.  The score
//     scale D^-0.5 absorbs the 2^-8 of the S product, the final 1 / l the 2^-4 of V.  All scales are powers of two: exact.
// Head dims 32..128 (Q pieces + O accumulators + the register prefetch of the next tile fit two waves per SIMD); anything else stays
// on attention.cpp.
#include <math.h>

#include "../common.h"

namespace mcvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr float AH_OP_SCALE = 16.0f;          // Q, K, V enter the fp16 pieces times 2^4
constexpr float AH_F16_MAX = 65504.0f;
constexpr float AH_P_LOG = 8.317766166719343f;        // 12 ln 2: probabilities times 2^12

__device__ __forceinline__ unsigned ah_cvt_pk(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// two-way fp16 split of a pair (conv_wino2h.cpp: h2_split2); CLAMP: the inputs may exceed the fp16 range
template <bool CLAMP>
__device__ __forceinline__ void ah_split2(float x, float y, unsigned& w1, unsigned& w2) {
    if (CLAMP) {
        x = __builtin_amdgcn_fmed3f(x, -AH_F16_MAX, AH_F16_MAX);
        y = __builtin_amdgcn_fmed3f(y, -AH_F16_MAX, AH_F16_MAX);
    }
    w1 = ah_cvt_pk(x, y);
    float rx, ry;
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(w1), "v"(x));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(w1), "v"(y));
    w2 = ah_cvt_pk(rx, ry);
}
__device__ __forceinline__ f32x16 ah_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int DT>   // head dim D = 32*DT
__global__ __launch_bounds__(256, 2) void attn_h2_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, int heads, int S,
                                                          float scale_s) {
    constexpr int D = 32 * DT, NST = D / 16;
    constexpr int NIK = (4 * D + 255) / 256;     // K staging items per thread: (row pair, 4 keys) -> 2 float4
    constexpr int NIV = DT;                      // V staging items per thread: (channel, 4 keys) -> 1 float4   (8 D / 256)
    extern __shared__ __attribute__((aligned(16))) float smem_attn_h2[];
    unsigned* sK = reinterpret_cast<unsigned*>(smem_attn_h2);      // [NST][2 pieces][64 lanes][4]
    unsigned* sV = sK + NST * 512;                                  // [DT][2 steps][2 pieces][64 lanes][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y;
    const int b = bh / heads, hd = bh - b * heads;
    const float* qb = qkv + ((long)b * 3 * C + hd * D) * S;
    const float* kb = qb + (long)C * S;
    const float* vb = kb + (long)C * S;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bool active = q0 < S;

    // Q pieces: lane (query l31, half) holds channels 16 st + 2e + half, e = 0..7 of every step; dword j = elements (2j, 2j+1)
    u32x4 q1[NST], q2[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c0 = 16 * st + 4 * j + half;
            const float a0 = active ? qb[(long)c0 * S + q0 + l31] : 0.0f;
            const float a1 = active ? qb[(long)(c0 + 2) * S + q0 + l31] : 0.0f;
            unsigned w1, w2;
            ah_split2<true>(a0 * AH_OP_SCALE, a1 * AH_OP_SCALE, w1, w2);
            q1[st][j] = w1; q2[st][j] = w2;
        }

    f32x16 o[DT];
#pragma unroll
    for (int ct = 0; ct < DT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.0f;
    float m_run = -1e30f, l_run = 0.0f;

    // ---- staging roles.  K item idx = i*256 + tid < 4D: row pair rp = idx >> 3 = (st, j, h) -> rows c, c + 2 with c = 16 st + 4j + h,
    //      keys 4 kq .. 4 kq + 3 (kq = idx & 7).  V item idx = i*256 + tid: channel c = idx >> 3, keys 4q .. 4q + 3 (q = idx & 7).
    f32x4 rk0[NIK], rk1[NIK], rv[NIV];
    const int ntiles = S / 32;
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < NIK; ++i) {
            const int idx = min(i * 256 + tid, 4 * D - 1);
            const int rp = idx >> 3, kq = idx & 7;
            const int c = 16 * (rp >> 3) + 4 * ((rp >> 1) & 3) + (rp & 1);
            const long off = (long)c * S + t * 32 + kq * 4;
            rk0[i] = *reinterpret_cast<const f32x4*>(kb + off);
            rk1[i] = *reinterpret_cast<const f32x4*>(kb + off + 2 * (long)S);
        }
#pragma unroll
        for (int i = 0; i < NIV; ++i) {
            const int idx = i * 256 + tid;
            rv[i] = *reinterpret_cast<const f32x4*>(vb + (long)(idx >> 3) * S + t * 32 + (idx & 7) * 4);
        }
    };
    gload(0);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();                          // previous tile fully consumed
        // ---- split the staged tile and park the pieces in operand order
#pragma unroll
        for (int i = 0; i < NIK; ++i) {
            const int idx = i * 256 + tid;
            if (NIK * 256 == 4 * D || idx < 4 * D) {
                const int rp = idx >> 3, kq = idx & 7;
                const int st = rp >> 3, j = (rp >> 1) & 3, h = rp & 1;
                unsigned* d1 = sK + ((st * 2 + 0) * 64 + h * 32 + kq * 4) * 4 + j;       // key kq*4 + i4: + 4 dwords each
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    unsigned w1, w2;
                    ah_split2<true>(rk0[i][i4] * AH_OP_SCALE, rk1[i][i4] * AH_OP_SCALE, w1, w2);
                    d1[i4 * 4] = w1;
                    d1[i4 * 4 + 256] = w2;            // the second piece: + 64 lanes x 4 dwords
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NIV; ++i) {
            const int idx = i * 256 + tid;
            const int c = idx >> 3, q = idx & 7;
            const int ct = c >> 5, m = c & 31, h = q & 1, s2 = q >> 2, j0 = 2 * ((q >> 1) & 1);
            unsigned a1, a2, b1, b2;
            ah_split2<true>(rv[i][0] * AH_OP_SCALE, rv[i][1] * AH_OP_SCALE, a1, a2);
            ah_split2<true>(rv[i][2] * AH_OP_SCALE, rv[i][3] * AH_OP_SCALE, b1, b2);
            unsigned* d1 = sV + (((ct * 2 + s2) * 2 + 0) * 64 + h * 32 + m) * 4 + j0;
            *reinterpret_cast<u32x2*>(d1) = u32x2{a1, b1};
            *reinterpret_cast<u32x2*>(d1 + 256) = u32x2{a2, b2};
        }
        __syncthreads();
        if (t + 1 < ntiles) gload(t + 1);

        // ---- S^T tile: 32 keys x 32 queries, times 2^8
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.0f;
        const u32x4* sKl = reinterpret_cast<const u32x4*>(sK) + lane;
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            const u32x4 k1 = sKl[(s * 2 + 0) * 64], k2 = sKl[(s * 2 + 1) * 64];
            st = ah_mfma(k1, q2[s], st);
            st = ah_mfma(k2, q1[s], st);
            st = ah_mfma(k1, q1[s], st);
        }

        // ---- online softmax over keys (this lane: 16 keys of query l31; partner lane^32 holds the other 16); p times 2^12
        float mt = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] *= scale_s; mt = fmaxf(mt, st[r]); }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __expf(m_run - m_new);
        const float shift = AH_P_LOG - m_new;
        float ps = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = __expf(st[r] + shift); ps += st[r]; }
        l_run = l_run * alpha + ps;               // per-half partial sum; halves are added at the end
        m_run = m_new;
#pragma unroll
        for (int ct = 0; ct < DT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
        u32x4 p1[2], p2[2];                       // B operand of the PV product: step s2, dword j = registers 8 s2 + 2j, + 1
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned w1, w2;
                ah_split2<false>(st[8 * s2 + 2 * j], st[8 * s2 + 2 * j + 1], w1, w2);      // 0 <= p <= 2^12
                p1[s2][j] = w1; p2[s2][j] = w2;
            }

        // ---- O += V * P
        const u32x4* sVl = reinterpret_cast<const u32x4*>(sV) + lane;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            u32x4 v1[DT], v2[DT];
#pragma unroll
            for (int ct = 0; ct < DT; ++ct) {
                v1[ct] = sVl[((ct * 2 + s2) * 2 + 0) * 64];
                v2[ct] = sVl[((ct * 2 + s2) * 2 + 1) * 64];
            }
#pragma unroll
            for (int ct = 0; ct < DT; ++ct) o[ct] = ah_mfma(v1[ct], p2[s2], o[ct]);
#pragma unroll
            for (int ct = 0; ct < DT; ++ct) o[ct] = ah_mfma(v2[ct], p1[s2], o[ct]);
#pragma unroll
            for (int ct = 0; ct < DT; ++ct) o[ct] = ah_mfma(v1[ct], p1[s2], o[ct]);
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = (1.0f / AH_OP_SCALE) / l_tot;        // l carries the 2^12 of the probabilities, O the 2^12 * 2^4 of P and V
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

bool attention_h2_supported(int C, int heads, int HW) {
    if (heads <= 0 || C % heads != 0) return false;
    const int D = C / heads;
    return D % 32 == 0 && D >= 32 && D <= 128 && HW % 32 == 0;
}

int launch_attention_h2(const float* qkv, float* out, int B, int C, int heads, int HW, hipStream_t s) {
    MCVD_REQUIRE(attention_h2_supported(C, heads, HW), "attention f16x2: unsupported (C=%d heads=%d HW=%d)", C, heads, HW);
    const int D = C / heads;
    const float scale = (float)pow((double)D, -0.5);   // int(C)**-0.5 as a Python double, then fp32 (layerspp.py:239)
    const float scale_s = scale * (1.0f / (AH_OP_SCALE * AH_OP_SCALE));      // exact: the S product carries 2^8
    dim3 grid((HW + 127) / 128, B * heads);
    const size_t lds = (size_t)(D / 16 * 512 + D / 32 * 1024) * sizeof(unsigned);
    switch (D / 32) {
        case 1: hipLaunchKernelGGL(attn_h2_kernel<1>, grid, dim3(256), lds, s, qkv, out, C, heads, HW, scale_s); break;
        case 2: hipLaunchKernelGGL(attn_h2_kernel<2>, grid, dim3(256), lds, s, qkv, out, C, heads, HW, scale_s); break;
        case 3: hipLaunchKernelGGL(attn_h2_kernel<3>, grid, dim3(256), lds, s, qkv, out, C, heads, HW, scale_s); break;
        default: hipLaunchKernelGGL(attn_h2_kernel<4>, grid, dim3(256), lds, s, qkv, out, C, heads, HW, scale_s); break;
    }
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_attention(int mode, int f16x2, const float* qkv, float* out, int B, int C, int heads, int HW, hipStream_t s) {
    if (mode == 1) return launch_attention_naive(qkv, out, B, C, heads, HW, s);
    if ((mode == 3 || (mode == 0 && f16x2)) && attention_h2_supported(C, heads, HW)) return launch_attention_h2(qkv, out, B, C, heads, HW, s);
    return launch_attention_mfma(qkv, out, B, C, heads, HW, s);
}

}  // namespace mcvd
