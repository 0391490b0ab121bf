// 3x3 convolution by Winograd F(2x2, 3x3) on v_mfma_f32_32x32x2_f32 -- 2.25x fewer matrix-pipe MACs than the direct
// implicit GEMM, still exact-fp32 arithmetic (measured forward error vs fp64: 2.3e-6 against 1.8e-6 for the direct form).
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, d = 4x4 input patch of silu?(A_c x + B_c)
//
// For each of the 16 transform positions xi the channel contraction is an independent GEMM
//   M_xi[co][tile] = sum_ci U_xi[co][ci] * V_xi[ci][tile]
// MFMA A-operand = transformed weights U (rows = cout), B-operand = transformed patches V (cols = tiles).
//
// Workgroup (512 threads = 8 waves = TWO waves per SIMD, one workgroup per CU): a region of 8 x 16 output pixels =
// 4 x 8 = 32 tiles of one sample, 32*COT output channels, all 16 positions: wave w owns positions 2w, 2w+1 (2*COT
// accumulator tiles, <= 96 registers, so two waves fit a SIMD and one wave's staging VALU work overlaps its partner's MFMAs).
//   * U chunks (8 input channels x 16 positions x 32*COT couts) stream HBM/L2 -> LDS by LDS-DMA, double-buffered;
//     the packed layout is the direct kernel's with 16 "taps": Up[(ci*16 + xi)*CoutP + co].
//   * staging of chunk i+1 rides inside the MFMA loop of chunk i in two steps: (a) the region's 10 x 18 input patch of the
//     8 channels is loaded row-wise (coalesced, one chunk ahead), GroupNorm/temb affine + SiLU applied ONCE per pixel,
//     zero padding AFTER the activation, written to a small LDS patch; barrier; (b) thread pair (t, t+256) takes (channel
//     (t&255)>>5, tile t&31), reads its window from the LDS patch, applies B^T d B (two of the four rows each) and writes
//     position values (conflict-free) into the double-buffered V.  Two barriers per chunk, both between MFMA groups.
//   * epilogue: per 32-cout sub-tile the 16 position planes go through LDS, each thread inverse-transforms 2 (cout, tile)
//     pairs (A^T M A), adds bias (+ residual), scales, and stores 2x2 pixels as two 8-byte stores.
#include <stdlib.h>

#include "../common.h"

namespace mcvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float silu_w(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

constexpr int WINO_CK = 8;       // input channels per chunk
constexpr int WINO_T = 32;       // tiles per workgroup (4 x 8)

template <int COT, int PRO, int VAR>     // PRO: 0 raw input, 1 affine, 2 affine + SiLU;  VAR: software-pipeline variant (see the K loop)
__global__ __launch_bounds__(512) void conv_wino_kernel(ConvArgs a) {
    constexpr int NT = 512;
    constexpr int CK = WINO_CK, T = WINO_T, BCO = 32 * COT;
    constexpr int USZ = CK * 16 * BCO;          // floats per U chunk
    constexpr int VSZ = CK * 16 * T;            // floats per V chunk
    constexpr int UCOUNT = USZ / 4;             // float4 per U chunk
    constexpr int MAXU = UCOUNT / 512;
    static_assert(UCOUNT % 512 == 0, "every thread issues the same number of U DMA pieces");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PP = 20;                      // LDS patch row pitch (18 columns used)
    constexpr int PSZ = CK * 10 * PP;           // activated input patch of one chunk: [CK][10 rows][PP]
    constexpr int PCOUNT = CK * 10 * 18;        // patch elements per chunk
    constexpr int MAXP = (PCOUNT + 511) / 512;
    float* sU = smem;                           // [2][USZ]
    float* sV = smem + 2 * USZ;                 // [2][VSZ]
    constexpr int PBUF = PSZ + 4;               // + 4 floats of dump space for unused patch slots
    float* sP = smem + 2 * USZ + 2 * VSZ;       // [2][PBUF]: patch(ch+2) is written while patch(ch+1) is transformed

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int H = a.H, W = a.W, HW = H * W, Cin = a.Cin;
    const int rx_n = W >> 4, ry_n = H >> 3;
    const int reg_id = blockIdx.x;
    const int b = reg_id / (rx_n * ry_n);
    const int rr = reg_id - b * (rx_n * ry_n);
    const int oy0 = (rr / rx_n) * 8, ox0 = (rr % rx_n) * 16;
    const int co0 = blockIdx.y * BCO;

    // ---- staging role: (channel-in-chunk, tile)
    const int s_ci = (tid & 255) >> 5, s_tile = tid & 31, s_h = tid >> 8;      // s_h: which two rows of B^T d this thread makes
    const int s_ty = s_tile >> 3, s_tx = s_tile & 7;
    const int p_rd = s_ci * 10 * PP + (2 * s_ty + s_h) * PP + 2 * s_tx;        // rows s_h..s_h+2 of the 4x4 window in the LDS patch

    // ---- patch-load slots (chunk invariant): element e -> (channel, patch row, patch col)
    // p_ci = channel-in-chunk, or CK + channel-in-chunk when the element is zero padding / an unused slot
    int p_lds[MAXP], p_goff[MAXP], p_ci[MAXP];
#pragma unroll
    for (int sl = 0; sl < MAXP; ++sl) {
        const int e = sl * NT + tid;
        if (e < PCOUNT) {
            const int ci = e / 180, rem = e - ci * 180;
            const int r = rem / 18, c = rem - r * 18;
            const int y = oy0 - 1 + r, x = ox0 - 1 + c;
            const bool inside = y >= 0 && y < H && x >= 0 && x < W;
            p_lds[sl] = ci * 10 * PP + r * PP + c;
            p_goff[sl] = min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1);
            p_ci[sl] = ci + (inside ? 0 : CK);
        } else {
            p_lds[sl] = PSZ; p_goff[sl] = 0; p_ci[sl] = CK;         // unused slot: harmless load, store into the dump word
        }
    }

    int u_goff[MAXU];
#pragma unroll
    for (int s = 0; s < MAXU; ++s) {
        const int e = s * NT + tid;
        const int row = e / (BCO / 4);
        const int c4 = e - row * (BCO / 4);
        u_goff[s] = row * a.CoutP + co0 + c4 * 4;
    }

    float pd[MAXP], pA[MAXP], pB[MAXP];

#define WINO_DMA_U(ch)                                                                                          \
    {                                                                                                           \
        const float* usrc = a.wpw + (long)(ch) * CK * 16 * a.CoutP;                                             \
        float* udst = sU + (((ch) & 1) ? USZ : 0);                                                              \
        _Pragma("unroll") for (int s = 0; s < MAXU; ++s)                                                        \
            __builtin_amdgcn_global_load_lds(                                                                   \
                (const __attribute__((address_space(1))) void*)(usrc + u_goff[s]),                              \
                (__attribute__((address_space(3))) void*)(udst + (s * NT + wave * 64) * 4), 16, 0, 0);          \
    }
    /* issue the (unconditional, clamped) loads of the raw input patch of chunk `ch`.  The chunk never straddles the     \
       concat seam (launch check), so source tensor and base are wave-uniform; channels past Cin re-read the last one. */ \
#define WINO_LOAD_P(ch)                                                                                         \
    {                                                                                                           \
        const int cb = min((ch) * CK, Cin - 1);                                                                 \
        const int cmax = Cin - 1 - cb;                        /* last valid channel-in-chunk (>= 0) */           \
        const bool second = cb >= a.C0;                                                                         \
        const float* srcb = second ? a.x1 + ((long)b * a.C1 + (cb - a.C0)) * HW : a.x0 + ((long)b * a.C0 + cb) * HW; \
        const bool hc = PRO && a.coef;               /* SiLU without an affine: identity coefficients */        \
        const float* cfb = hc ? a.coef + ((long)b * Cin + cb) * 2 : a.wpw;                                      \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                                   \
            const int cl = min(p_ci[sl] & (CK - 1), cmax);                                                      \
            pd[sl] = srcb[cl * HW + p_goff[sl]];                                                                \
            const f32x2 cf = *reinterpret_cast<const f32x2*>(cfb + (hc ? cl * 2 : 0));                          \
            pA[sl] = cf.x;                              /* raw: consuming it here would expose the latency */ \
            pB[sl] = cf.y;                                                                                      \
        }                                                                                                       \
    }
    /* activate once per pixel and park the patch in LDS (zero padding applies AFTER the activation) */
#define WINO_WRITE_P(ch)                                                                                        \
    {                                                                                                           \
        float* sPw = sP + (((ch) & 1) ? PBUF : 0);                                                              \
        const int nvalid = Cin - (ch) * CK;                   /* channels-in-chunk below this are real */         \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                                   \
            float v = pd[sl];                                                                                   \
            if (PRO >= 1) v = a.coef ? v * pA[sl] + pB[sl] : v;                                                 \
            if (PRO == 2) v = silu_w(v);                                                                        \
            sPw[p_lds[sl]] = (p_ci[sl] < min(nvalid, CK)) ? v : 0.0f;                                           \
        }                                                                                                       \
    }
    /* B^T d B, two of the four rows of B^T d per thread (s_h), -> 8 of the 16 position planes of V(ch) */
#define WINO_WRITE_V(ch)                                                                                        \
    {                                                                                                           \
        float* vdst = sV + (((ch) & 1) ? VSZ : 0) + s_ci * 16 * T + s_tile;                                     \
        const float* sPr = sP + (((ch) & 1) ? PBUF : 0);                                                        \
        float ra[4], rb[4], rcc[4];                               /* window rows s_h, s_h+1, s_h+2 */           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            ra[j] = sPr[p_rd + j];                                                                              \
            rb[j] = sPr[p_rd + PP + j];                                                                         \
            rcc[j] = sPr[p_rd + 2 * PP + j];                                                                    \
        }                                                                                                       \
        float mA[4], mB[4];              /* s_h=0: rows 0 (t0-t2), 1 (t1+t2)   s_h=1: rows 3 (t1-t3), 2 (t2-t1) */ \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            mA[j] = ra[j] - rcc[j];                                                                             \
            mB[j] = s_h ? (rb[j] - ra[j]) : (rb[j] + rcc[j]);                                                   \
        }                                                                                                       \
        const int iA = s_h ? 3 : 0, iB = s_h ? 2 : 1;                                                           \
        vdst[(iA * 4 + 0) * T] = mA[0] - mA[2];                                                                 \
        vdst[(iA * 4 + 1) * T] = mA[1] + mA[2];                                                                 \
        vdst[(iA * 4 + 2) * T] = mA[2] - mA[1];                                                                 \
        vdst[(iA * 4 + 3) * T] = mA[1] - mA[3];                                                                 \
        vdst[(iB * 4 + 0) * T] = mB[0] - mB[2];                                                                 \
        vdst[(iB * 4 + 1) * T] = mB[1] + mB[2];                                                                 \
        vdst[(iB * 4 + 2) * T] = mB[2] - mB[1];                                                                 \
        vdst[(iB * 4 + 3) * T] = mB[1] - mB[3];                                                                 \
    }
#define WINO_MFMA(kp)                                                                                           \
    {                                                                                                           \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                         \
            const int row = (2 * (kp) + half) * 16 + wave * 2 + q;                                              \
            const float bv = sVc[row * T + l31];                                                                \
            _Pragma("unroll") for (int ct = 0; ct < COT; ++ct)                                                  \
                acc[q][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(sUc[row * BCO + ct * 32 + l31], bv, acc[q][ct], 0, 0, 0); \
        }                                                                                                       \
    }

    f32x16 acc[2][COT];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][ct][r] = 0.0f;

    const int nchunks = a.CinP / CK;
    if (VAR == 0) {
        // two barriers per chunk: patch(ch+1) is written before the mid-chunk barrier and transformed after it
        WINO_DMA_U(0);
        WINO_LOAD_P(0);
        WINO_WRITE_P(0);
        __syncthreads();                       // patch(0) visible
        WINO_WRITE_V(0);
        WINO_LOAD_P(1);
        __syncthreads();                       // V(0) visible, U(0) landed (the fence drains the DMA)
        for (int ch = 0; ch < nchunks; ++ch) {
            const bool more = ch + 1 < nchunks;
            if (more) WINO_DMA_U(ch + 1);
            const float* sUc = sU + ((ch & 1) ? USZ : 0);
            const float* sVc = sV + ((ch & 1) ? VSZ : 0);
            WINO_MFMA(0)
            if (more) WINO_WRITE_P(ch + 1);    // registers were loaded during the previous chunk
            WINO_MFMA(1)
            __syncthreads();                   // patch(ch+1) visible to the tile transforms
            if (more) WINO_LOAD_P(ch + 2);
            WINO_MFMA(2)
            if (more) WINO_WRITE_V(ch + 1);
            WINO_MFMA(3)
            __syncthreads();                   // chunk ch consumed by every wave; V(ch+1) visible; U(ch+1) landed
        }
    } else {
        // ONE barrier per chunk.  While the MFMAs consume (U, V)(ch): patch(ch+2) is activated out of the registers loaded
        // one chunk earlier into the other patch buffer, the raw loads of patch(ch+3) and the DMA of U(ch+1) are issued (all
        // at the top, so the barrier's vmcnt(0) never waits on a young load), V(ch+1) is made from patch(ch+1).
        // Chunks past the end are staged as zeros (clamped loads), so the loop body carries no branches.
        WINO_DMA_U(0);
        WINO_LOAD_P(0);
        WINO_WRITE_P(0);
        WINO_LOAD_P(1);
        __syncthreads();                       // patch(0) visible
        WINO_WRITE_V(0);
        WINO_WRITE_P(1);
        WINO_LOAD_P(2);
        __syncthreads();                       // V(0), patch(1) visible, U(0) landed
        for (int ch = 0; ch + 1 < nchunks; ++ch) {
            const float* sUc = sU + ((ch & 1) ? USZ : 0);
            const float* sVc = sV + ((ch & 1) ? VSZ : 0);
            WINO_WRITE_P(ch + 2);
            WINO_LOAD_P(ch + 3);
            WINO_DMA_U(ch + 1);
            if (VAR == 2) {
                // the two waves of a SIMD (w, w+4) run their VALU-heavy tile transforms at different points of the chunk
                if (wave < 4) {
                    WINO_MFMA(0)
                    WINO_WRITE_V(ch + 1);
                    WINO_MFMA(1)
                    WINO_MFMA(2)
                    WINO_MFMA(3)
                } else {
                    WINO_MFMA(0)
                    WINO_MFMA(1)
                    WINO_MFMA(2)
                    WINO_WRITE_V(ch + 1);
                    WINO_MFMA(3)
                }
            } else {
                WINO_MFMA(0)
                WINO_WRITE_V(ch + 1);
                WINO_MFMA(1)
                WINO_MFMA(2)
                WINO_MFMA(3)
                if (VAR == 3) {
                    // pin an interleave: every MFMA is followed by a slice of the staging work of this iteration
                    _Pragma("unroll") for (int i = 0; i < 8 * COT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // 2 DS reads
                        __builtin_amdgcn_sched_group_barrier(0x002, 24 / COT, 0);   // VALU
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // 1 DS write
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
                    }
                }
            }
            __syncthreads();                   // chunk ch consumed by every wave; V(ch+1), patch(ch+2) visible; U(ch+1) landed
        }
        {
            const int ch = nchunks - 1;
            const float* sUc = sU + ((ch & 1) ? USZ : 0);
            const float* sVc = sV + ((ch & 1) ? VSZ : 0);
            WINO_MFMA(0)
            WINO_MFMA(1)
            WINO_MFMA(2)
            WINO_MFMA(3)
            __syncthreads();                   // the epilogue reuses the LDS
        }
    }

    // ---------------- inverse transform + epilogue, one 32-cout sub-tile at a time ----------------
    float* sM = smem;                      // [16][32 couts][32 tiles] = 64 KiB, the K loop is done with the LDS
    const int e_tile = tid & 31;
    const int e_ty = e_tile >> 3, e_tx = e_tile & 7;
    const long pix = (long)(oy0 + 2 * e_ty) * W + ox0 + 2 * e_tx;
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int xi = wave * 2 + q;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = (r & 3) + 8 * (r >> 2) + 4 * half;
                sM[(xi * 32 + col) * T + l31] = acc[q][ct][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int col = (tid >> 5) + 16 * k;                // cout within the sub-tile
            const int co = co0 + ct * 32 + col;
            float mm[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) mm[xi] = sM[(xi * 32 + col) * T + e_tile];
            float t0[4], t1[4];                                 // A^T M
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                t0[l] = mm[0 * 4 + l] + mm[1 * 4 + l] + mm[2 * 4 + l];
                t1[l] = mm[1 * 4 + l] - mm[2 * 4 + l] - mm[3 * 4 + l];
            }
            float y00 = t0[0] + t0[1] + t0[2], y01 = t0[1] - t0[2] - t0[3];
            float y10 = t1[0] + t1[1] + t1[2], y11 = t1[1] - t1[2] - t1[3];
            const float bvv = a.bias[co];                       // zero-padded to CoutP
            if (co < a.Cout) {
                const long o = ((long)b * a.Cout + co) * HW + pix;
                y00 += bvv; y01 += bvv; y10 += bvv; y11 += bvv;
                if (a.res) {
                    const float2 r0 = *reinterpret_cast<const float2*>(a.res + o);
                    const float2 r1 = *reinterpret_cast<const float2*>(a.res + o + W);
                    y00 += r0.x; y01 += r0.y; y10 += r1.x; y11 += r1.y;
                }
                *reinterpret_cast<float2*>(a.y + o) = make_float2(y00 * a.out_scale, y01 * a.out_scale);
                *reinterpret_cast<float2*>(a.y + o + W) = make_float2(y10 * a.out_scale, y11 * a.out_scale);
            }
        }
        __syncthreads();
    }
#undef WINO_DMA_U
#undef WINO_LOAD_P
#undef WINO_WRITE_P
#undef WINO_WRITE_V
#undef WINO_MFMA
}

bool conv_wino_supported(int ks, int H, int W) { return ks == 3 && H % 8 == 0 && W % 16 == 0 && H >= 8 && W >= 16; }

int conv_wino_cout_tile(int Cout) {
    if (Cout % 96 == 0) return 3;
    if (Cout % 64 == 0) return 2;
    return 1;
}

static int env_int_w(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

template <int COT, int PRO, int VAR>
static int wino_launch3(const ConvArgs& a, hipStream_t s) {
    constexpr int BCO = 32 * COT;
    const size_t lds = (size_t)(2 * WINO_CK * 16 * BCO + 2 * WINO_CK * 16 * WINO_T + 2 * (WINO_CK * 10 * 20 + 4)) * sizeof(float);
    static bool raised = false;
    if (!raised) {
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<COT, PRO, VAR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    dim3 grid(a.B * (a.H / 8) * (a.W / 16), a.CoutP / BCO);
    hipLaunchKernelGGL((conv_wino_kernel<COT, PRO, VAR>), grid, dim3(512), lds, s, a);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int COT, int PRO>
static int wino_launch2(const ConvArgs& a, hipStream_t s) {
    switch (conv_wino_variant()) {
        case 0: return wino_launch3<COT, PRO, 0>(a, s);
        case 3: return wino_launch3<COT, PRO, 3>(a, s);
        case 1: return wino_launch3<COT, PRO, 1>(a, s);
        default: return wino_launch3<COT, PRO, 2>(a, s);
    }
}

template <int COT>
static int wino_launch(const ConvArgs& a, hipStream_t s) {
    if (!a.coef && !a.act) return wino_launch2<COT, 0>(a, s);
    if (!a.act) return wino_launch2<COT, 1>(a, s);
    return wino_launch2<COT, 2>(a, s);
}

// Which Winograd kernel serves shape id 4 (env MCVD_WINO_VAR, read once): 5 = 1024 threads, weights register-fed from global
// memory (conv_wino16r.cpp, default); 4 = 1024 threads, weights LDS-staged by DMA (conv_wino16.cpp); 0..3 = the 512-thread
// kernel of this file with its pipeline variants.  The packed weight layout depends on it (launch_pack_wino_weight).
int conv_wino_variant() {
    static const int var = env_int_w("MCVD_WINO_VAR", 5);
    return var;
}

bool conv_wino_usable(const ConvArgs& a) {
    if (!conv_wino_supported(a.ks, a.H, a.W) || !a.wpw) return false;
    if (conv_wino_variant() == 5) return conv_wino16r_supported(a);
    return a.CinP % WINO_CK == 0 && (a.C1 == 0 || a.C0 % WINO_CK == 0);
}

int launch_conv_wino(const ConvArgs& a, hipStream_t s) {
    MCVD_REQUIRE(conv_wino_usable(a), "winograd conv: unsupported (ks=%d H=%d W=%d Cin=%d C0=%d, packed weights %s)", a.ks, a.H, a.W,
                 a.Cin, a.C0, a.wpw ? "present" : "missing");
    const int cot = conv_wino_cout_tile(a.Cout);
    MCVD_REQUIRE(a.CoutP % (32 * cot) == 0, "winograd conv: CoutP=%d vs tile %d", a.CoutP, 32 * cot);
    const int var = conv_wino_variant();
    if (var == 5) return launch_conv_wino16r(a, cot, s);
    if (var == 4 && a.Cin <= 1024) return launch_conv_wino16(a, cot, s);    // 1024-thread workgroups (conv_wino16.cpp)
    switch (cot) {
        case 1: return wino_launch<1>(a, s);
        case 2: return wino_launch<2>(a, s);
        default: return wino_launch<3>(a, s);
    }
}

// U = G g G^T per (cout, cin):  [Cout][Cin][3][3] -> Up[(ci*16 + xi)*CoutP + co]
__global__ void pack_wino_weight_kernel(const float* w, float* up, int Cout, int Cin, int CoutP) {
    const long n = (long)Cout * Cin;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin), co = (int)(i / Cin);
        const float* g = w + i * 9;
        float t[4][3];                                          // G g
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            t[0][j] = g[0 * 3 + j];
            t[1][j] = 0.5f * (g[0 * 3 + j] + g[1 * 3 + j] + g[2 * 3 + j]);
            t[2][j] = 0.5f * (g[0 * 3 + j] - g[1 * 3 + j] + g[2 * 3 + j]);
            t[3][j] = g[2 * 3 + j];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {                           // (.) G^T
            const float u0 = t[r][0];
            const float u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
            const float u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
            const float u3 = t[r][2];
            float* dst = up + ((long)ci * 16 + r * 4) * CoutP + co;
            dst[0] = u0;
            dst[(long)CoutP] = u1;
            dst[2L * CoutP] = u2;
            dst[3L * CoutP] = u3;
        }
    }
}

// `up` (CinP*16*CoutP floats) must be zero-filled by the caller: padded channels stay zero in either layout.
int launch_pack_wino_weight(const float* w, float* up, int Cout, int Cin, int CinP, int CoutP, hipStream_t s) {
    if (conv_wino_variant() == 5) return launch_pack_wino_weight_r(w, up, Cout, Cin, CinP, CoutP, conv_wino_cout_tile(Cout), s);
    const long n = (long)Cout * Cin;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_wino_weight_kernel, dim3(blocks), dim3(256), 0, s, w, up, Cout, Cin, CoutP);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mcvd
