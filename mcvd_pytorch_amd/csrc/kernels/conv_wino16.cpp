// Winograd F(2x2, 3x3) on v_mfma_f32_32x32x2_f32, 1024-thread workgroups: the same data flow as conv_wino.cpp (see its header
// for the algebra and the LDS layouts) re-sliced over 16 waves = FOUR waves per SIMD, so that the matrix pipe always finds a
// wave with an MFMA ready while the others run the transform / staging VALU work or wait on the LDS.
//
//   * wave w owns transform position xi = w: COT accumulator tiles (48 registers at COT = 3, the 128-VGPR budget of 4 waves/SIMD
//     holds), per k-pair 1 V read + COT U reads + COT MFMAs;
//   * staging per chunk: U by LDS-DMA (COT 16-byte pieces per thread); the 10 x 18 x 8 activated input patch (<= 2 elements per
//     thread, loaded one chunk ahead, written to a double-buffered LDS patch); the tile transform B^T d B split by ROW: threads
//     256*i .. 256*i+255 (waves 4i..4i+3) make row i of B^T d for (channel, tile) = (t & 255) and its four column combinations;
//   * waves 4i..4i+3 share their SIMDs with the waves of the other three row groups (wave w sits on SIMD w % 4), and row group i
//     runs its transform after MFMA group i of the chunk: at any time at most one of the four waves of a SIMD is in its VALU-heavy
//     phase.  One barrier per chunk;
//   * prologue: the raw loads of the first three chunks are issued back to back (one HBM latency instead of three);
//   * epilogue: per 32-cout sub-tile the 16 position planes go through LDS and every thread inverse-transforms ONE (cout, tile).
#include <stdlib.h>

#include "../common.h"

namespace mcvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float silu_w16(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

constexpr int W16_CK = 8;        // input channels per chunk
constexpr int W16_T = 32;        // tiles per workgroup (4 x 8 tiles = 8 x 16 output pixels)
constexpr int W16_NT = 1024;

// PRO: 0 raw input, 1 affine, 2 affine + SiLU.
// DIAG (diagnostics only, wrong results): skip  1 the U DMA, 2 the patch loads, 4 the tile transform, 8 the MFMAs,
// 16 the patch activation/write, 32 the MFMA operand reads -- measures each component's marginal cost (tests/gpu_diag.py).
template <int COT, int PRO, int DIAG = 0>
__global__ __launch_bounds__(1024) void conv_wino16_kernel(ConvArgs a) {
    constexpr int NT = W16_NT, CK = W16_CK, T = W16_T, BCO = 32 * COT;
    constexpr int USZ = CK * 16 * BCO;          // floats per U chunk
    constexpr int VSZ = CK * 16 * T;            // floats per V chunk
    constexpr int MAXU = USZ / 4 / NT;          // 16-byte DMA pieces per thread and chunk (= COT)
    static_assert(USZ / 4 % NT == 0, "every thread issues the same number of U DMA pieces");
    constexpr int PP = 20;                      // LDS patch row pitch (18 columns used)
    constexpr int PSZ = CK * 10 * PP;           // activated input patch of one chunk: [CK][10 rows][PP]
    constexpr int PBUF = PSZ + 4;               // + dump space for unused patch slots
    constexpr int PCOUNT = CK * 10 * 18;
    constexpr int MAXP = (PCOUNT + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sU = smem;                           // [2][USZ]
    float* sV = smem + 2 * USZ;                 // [2][VSZ]
    float* sP = smem + 2 * USZ + 2 * VSZ;       // [2][PBUF]
    float* sCo = sP + 2 * PBUF;                 // [Cin][2] prologue coefficients (A_c, B_c) of this sample (PRO only)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int H = a.H, W = a.W, HW = H * W, Cin = a.Cin;
    const int rx_n = W >> 4, ry_n = H >> 3;
    // block id -> (region, cout tile): the cout tiles of one region get ids congruent mod 8 and adjacent in dispatch order, i.e.
    // they run at the same time on the SAME XCD and share the region's input patch through that XCD's L2.
    const int nct = a.CoutP / BCO;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int reg_id = (slot / nct) * 8 + xcd;
    if (reg_id >= a.B * rx_n * ry_n) return;
    const int b = reg_id / (rx_n * ry_n);
    const int rr = reg_id - b * (rx_n * ry_n);
    const int oy0 = (rr / rx_n) * 8, ox0 = (rr % rx_n) * 16;
    const int co0 = (slot - (slot / nct) * nct) * BCO;
    const int grp = wave >> 2;                  // row of B^T d this wave's threads make == pipeline phase of the wave

    // ---- transform role: (channel-in-chunk, tile) = tid & 255, row = grp
    const int s_ci = (tid & 255) >> 5, s_tile = tid & 31;
    const int s_ty = s_tile >> 3, s_tx = s_tile & 7;
    // row grp of B^T d:  0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3
    const int p_rd = s_ci * 10 * PP + 2 * s_ty * PP + 2 * s_tx;     // top-left of the 4x4 window in the LDS patch
    const int p_rdA = p_rd + (grp == 0 ? 0 : 1) * PP, p_rdB = p_rd + (grp == 3 ? 3 : 2) * PP;
    const float v_fa = grp == 2 ? -1.0f : 1.0f, v_fb = (grp == 1 || grp == 2) ? 1.0f : -1.0f;
    const int v_wr = s_ci * 16 * T + grp * 4 * T + s_tile;

    // ---- patch-load slots (chunk invariant); p_ci = channel-in-chunk, or CK + channel when the element is padding / unused
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
            p_lds[sl] = PSZ; p_goff[sl] = 0; p_ci[sl] = CK;
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

#define W16_DMA_U(ch)                                                                                           \
    {                                                                                                           \
        const float* usrc = a.wpw + (long)(ch) * CK * 16 * a.CoutP;                                             \
        float* udst = sU + (((ch) & 1) ? USZ : 0);                                                              \
        _Pragma("unroll") for (int s = 0; s < MAXU; ++s)                                                        \
            __builtin_amdgcn_global_load_lds(                                                                   \
                (const __attribute__((address_space(1))) void*)(usrc + u_goff[s]),                              \
                (__attribute__((address_space(3))) void*)(udst + (s * NT + wave * 64) * 4), 16, 0, 0);          \
    }
    /* unconditional, clamped raw loads of the patch of chunk `ch` into the register set D; the chunk never straddles the  \
       concat seam (launch check), channels past Cin re-read the last one and are zeroed at the write.                    \
       The loads are issued through inline asm ON PURPOSE: the values are consumed two chunks later, and the compiler's   \
       s_waitcnt insertion is not exact across the loop back-edge (it emits vmcnt(0) at the consumer, which also waits for \
       the one-chunk-old loads of the other register set and exposes the HBM latency).  Completion is guaranteed by the    \
       chunk barrier instead: vmcnt is an in-order counter and every barrier waits for all but the MAXP youngest VMEM      \
       operations, so a set loaded in chunk i has landed after the barrier of chunk i+1, before its use in chunk i+2.     \
       (Verified in the ISA: the destination registers are not copied between the load and the use.) */                    \
#define W16_LOAD_P(ch, D)                                                                                       \
    {                                                                                                           \
        const int cb = min((ch) * CK, Cin - 1);                                                                 \
        const int cmax = Cin - 1 - cb;                                                                          \
        const bool second = cb >= a.C0;                                                                         \
        const float* srcb = second ? a.x1 + ((long)b * a.C1 + (cb - a.C0)) * HW : a.x0 + ((long)b * a.C0 + cb) * HW; \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                                   \
            const unsigned off = (unsigned)(min(p_ci[sl] & (CK - 1), cmax) * HW + p_goff[sl]) * 4u;             \
            asm volatile("global_load_dword %0, %1, %2" : "=v"(D[sl]) : "v"(off), "s"(srcb) : "memory");        \
        }                                                                                                       \
    }
    /* activate once per pixel (coefficients from the LDS table) and park the patch in LDS; zero padding applies AFTER   \
       the activation */                                                                                         \
#define W16_WRITE_P(ch, D)                                                                                      \
    {                                                                                                           \
        float* sPw = sP + (((ch) & 1) ? PBUF : 0);                                                              \
        const int nvalid = Cin - (ch) * CK;                                                                     \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                                   \
            asm volatile("" : "+v"(D[sl]));      /* ordering token: nothing derived from D moves above this point */ \
            float v = D[sl];                                                                                    \
            if (PRO >= 1) {                                                                                     \
                const int c = min((ch) * CK + (p_ci[sl] & (CK - 1)), Cin - 1);                                  \
                const f32x2 cf = *reinterpret_cast<const f32x2*>(sCo + c * 2);                                  \
                v = v * cf.x + cf.y;                                                                            \
            }                                                                                                   \
            if (PRO == 2) v = silu_w16(v);                                                                      \
            sPw[p_lds[sl]] = (p_ci[sl] < min(nvalid, CK)) ? v : 0.0f;                                           \
        }                                                                                                       \
    }
    /* row grp of B^T d (rows RA, RB of the window, combined with wave-uniform +-1 factors), then (.) B: four position  \
       values -> V(ch)[ci][grp*4 + j][tile] */                                                                   \
#define W16_WRITE_V(ch)                                                                                         \
    {                                                                                                           \
        const float* sPr = sP + (((ch) & 1) ? PBUF : 0);                                                        \
        float* vdst = sV + (((ch) & 1) ? VSZ : 0) + v_wr;                                                       \
        float m[4];                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const float va = sPr[p_rdA + j], vb = sPr[p_rdB + j];                                               \
            m[j] = __builtin_fmaf(v_fb, vb, v_fa * va);      /* +-va +- vb, exact: the factors are +-1 */        \
        }                                                                                                       \
        vdst[0 * T] = m[0] - m[2];                                                                              \
        vdst[1 * T] = m[1] + m[2];                                                                              \
        vdst[2 * T] = m[2] - m[1];                                                                              \
        vdst[3 * T] = m[1] - m[3];                                                                              \
    }
    /* MFMA operands of k-pair `kp` of the current chunk: LDS -> registers (BV, AV[COT]) */
#define W16_LOAD_OPS(kp, BV, AV)                                                                                \
    {                                                                                                           \
        const int row = (2 * (kp) + half) * 16 + wave;                                                          \
        BV = sVc[row * T + l31];                                                                                \
        _Pragma("unroll") for (int ct = 0; ct < COT; ++ct) AV[ct] = sUc[row * BCO + ct * 32 + l31];             \
    }
#define W16_DO_MFMA(BV, AV)                                                                                     \
    _Pragma("unroll") for (int ct = 0; ct < COT; ++ct)                                                          \
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[ct], BV, acc[ct], 0, 0, 0);

    f32x16 acc[COT];
#pragma unroll
    for (int ct = 0; ct < COT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;

    // diagnostics (mcvd_ctx_set_debug_buffer): shader-clock time the wave a.wdma spends per phase
    const bool rec = a.dbg != nullptr && wave == a.wdma;
    unsigned long long tk0 = 0, tprev = 0, dt[2] = {0, 0};
    if (rec) tk0 = tprev = __builtin_amdgcn_s_memtime();
#define W16_STAMP(i)                                                                                            \
    if (rec) {                                                                                                  \
        const unsigned long long now = __builtin_amdgcn_s_memtime();                                            \
        dt[i] += now - tprev;                                                                                   \
        tprev = now;                                                                                            \
    }

    // ---- prologue: every global load of the first FOUR chunks + the coefficient table is issued before anything waits
    const int nchunks = a.CinP / CK;
    float pe[MAXP], po[MAXP];              // raw patch registers of the next even / odd chunk (loaded two chunks ahead)
    {
        float qd[MAXP], rd[MAXP];
        f32x2 cfl = {1.0f, 0.0f};
        W16_DMA_U(0);
        W16_LOAD_P(0, qd);
        W16_LOAD_P(1, rd);
        W16_LOAD_P(2, pe);
        W16_LOAD_P(3, po);
        if (PRO && a.coef && tid < Cin) cfl = *reinterpret_cast<const f32x2*>(a.coef + ((long)b * Cin + tid) * 2);
        if (PRO && tid < Cin) *reinterpret_cast<f32x2*>(sCo + tid * 2) = cfl;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the asm patch loads above (untracked by the compiler)
        if (PRO) __syncthreads();          // coefficient table visible
        W16_WRITE_P(0, qd);
        W16_WRITE_P(1, rd);
    }
    __syncthreads();                       // patch(0), patch(1) visible
    W16_WRITE_V(0);
    __syncthreads();                       // V(0) visible, U(0) landed
    W16_STAMP(0)

    // Skewed pipeline.  Per chunk a wave runs 4 MFMA groups (k-pairs) g0..g3 of COT MFMAs; the operands of a group are read
    // from LDS right after the previous group has been issued (register sets X / Y alternate).  The LAST group of a chunk is
    // only loaded before the barrier and issued AFTER it, so the matrix pipe has work while the first operand reads of the
    // next chunk are in flight and while the staging VMEM work is being issued (without the skew every barrier drains and
    // refills the pipe: ~1000 of the ~4000 cycles a chunk takes).  Staging is spread over four slots of the chunk (T: after
    // the deferred group, A / B / C: after g0 / g1 / g2) so that the four waves of a SIMD (one of each wave group) are in
    // different phases:
    //     stage unit = patch activation+write, U DMA, next patch loads:  slot T, every wave (straight-line code, so that the
    //                  compiler's vmcnt bookkeeping stays exact)
    //     tile transform (row grp of B^T d):                            slot T: grp 3,  A: grp 2,  B: grp 0,  C: grp 1
    // The raw patch loads run TWO chunks ahead of their use (HBM latency under load is of the order of one chunk time): the
    // loop is unrolled by two over the register sets pe / po, and neither the patch write (program order: before the DMA)
    // nor the barrier (vmcnt(MAXP)) waits for the youngest patch loads.
    static_assert(MAXP == 2, "the barrier in W16_CHUNK hard-codes vmcnt(MAXP)");
    float xb = 0.0f, yb = 0.0f, xa[COT], ya[COT];
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) xa[ct] = ya[ct] = 0.0f;          // the first "deferred" group multiplies zeros
#define W16_STAGE_UNIT(ch, D)                                                                                   \
    {                                                                                                           \
        if (!(DIAG & 16)) W16_WRITE_P((ch) + 2, D);                                                             \
        if (!(DIAG & 1)) W16_DMA_U((ch) + 1);                                                                   \
        if (!(DIAG & 2)) W16_LOAD_P((ch) + 4, D);                                                               \
    }
#define W16_CHUNK(ch, D)                                                                                        \
    {                                                                                                           \
        const float* sUc = sU + (((ch) & 1) ? USZ : 0);                                                         \
        const float* sVc = sV + (((ch) & 1) ? VSZ : 0);                                                         \
        if (!(DIAG & 8)) W16_DO_MFMA(yb, ya)              /* g3 of the previous chunk (read before the barrier) */ \
        if (!(DIAG & 40)) W16_LOAD_OPS(0, xb, xa)                                                               \
        W16_STAGE_UNIT(ch, D)                             /* slot T */                                          \
        if (grp == 3 && !(DIAG & 4)) W16_WRITE_V((ch) + 1)                                                      \
        if (!(DIAG & 8)) W16_DO_MFMA(xb, xa)              /* g0 */                                              \
        if (!(DIAG & 40)) W16_LOAD_OPS(1, yb, ya)                                                               \
        if (grp == 2 && !(DIAG & 4)) W16_WRITE_V((ch) + 1)                                                      \
        if (!(DIAG & 8)) W16_DO_MFMA(yb, ya)              /* g1 */                                              \
        if (!(DIAG & 40)) W16_LOAD_OPS(2, xb, xa)                                                               \
        if (grp == 0 && !(DIAG & 4)) W16_WRITE_V((ch) + 1) /* slot B */                                         \
        if (!(DIAG & 8)) W16_DO_MFMA(xb, xa)              /* g2 */                                              \
        if (!(DIAG & 40)) W16_LOAD_OPS(3, yb, ya)         /* deferred past the barrier */                       \
        if (grp == 1 && !(DIAG & 4)) W16_WRITE_V((ch) + 1) /* slot C */                                         \
        /* chunk ch read by every wave; V(ch+1), patch(ch+2) visible; U(ch+1) landed.  The barrier waits for this wave's  \
           LDS traffic and for all but its MAXP youngest VMEM operations: the U DMA must have landed, the patch loads      \
           issued after it stay in flight (in-order counter) */                                                          \
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                       \
    }
    for (int ch = 0; ch + 1 < nchunks; ch += 2) {
        W16_CHUNK(ch, pe)
        if (ch + 2 < nchunks) W16_CHUNK(ch + 1, po)
    }
#undef W16_CHUNK
#undef W16_STAGE_UNIT
    {
        const int ch = nchunks - 1;
        const float* sUc = sU + ((ch & 1) ? USZ : 0);
        const float* sVc = sV + ((ch & 1) ? VSZ : 0);
        W16_DO_MFMA(yb, ya)
        W16_LOAD_OPS(0, xb, xa)
        W16_LOAD_OPS(1, yb, ya)
        W16_DO_MFMA(xb, xa)
        W16_LOAD_OPS(2, xb, xa)
        W16_DO_MFMA(yb, ya)
        W16_LOAD_OPS(3, yb, ya)
        W16_DO_MFMA(xb, xa)
        W16_DO_MFMA(yb, ya)
    }
    __syncthreads();                       // the epilogue reuses the LDS
    W16_STAMP(1)

    // ---------------- inverse transform + epilogue, one 32-cout sub-tile at a time ----------------
    float* sM = smem;                      // [16 positions][32 couts][32 tiles] = 64 KiB
    const int e_tile = tid & 31, e_col = tid >> 5;
    const int e_ty = e_tile >> 3, e_tx = e_tile & 7;
    const long pix = (long)(oy0 + 2 * e_ty) * W + ox0 + 2 * e_tx;
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = (r & 3) + 8 * (r >> 2) + 4 * half;
            sM[(wave * 32 + col) * T + l31] = acc[ct][r];
        }
        __syncthreads();
        {
            const int co = co0 + ct * 32 + e_col;
            float mm[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) mm[xi] = sM[(xi * 32 + e_col) * T + e_tile];
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
    if (rec) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long* d = a.dbg + (long)blockIdx.x * 8;
            d[0] = dt[0]; d[1] = dt[1]; d[2] = 0; d[3] = 0; d[4] = 0;      // prologue, K loop
            d[5] = now - tprev;            // epilogue
            d[6] = (unsigned long long)nchunks;
            d[7] = now - tk0;
        }
    }
#undef W16_STAMP
#undef W16_LOAD_OPS
#undef W16_DO_MFMA
#undef W16_DMA_U
#undef W16_LOAD_P
#undef W16_WRITE_P
#undef W16_WRITE_V
}

static size_t wino16_lds_bytes(int cot, int Cin) {
    return (size_t)(2 * W16_CK * 16 * 32 * cot + 2 * W16_CK * 16 * W16_T + 2 * (W16_CK * 10 * 20 + 4) + 2 * Cin) * sizeof(float);
}

template <int COT, int PRO>
static int wino16_launch2(const ConvArgs& a, hipStream_t s) {
    constexpr int BCO = 32 * COT;
    const size_t lds = wino16_lds_bytes(COT, a.Cin);
    static bool raised = false;
    if (!raised) {
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino16_kernel<COT, PRO>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    const int nreg = a.B * (a.H / 8) * (a.W / 16);
    dim3 grid(((nreg + 7) / 8) * 8 * (a.CoutP / BCO));
    ConvArgs k = a;
    if (k.dbg) {
        const char* w = getenv("MCVD_DBG_WAVE");       // which wave records its phase times (diagnostics)
        k.wdma = w ? atoi(w) : 0;
    }
    static const int diag = getenv("MCVD_WINO_DIAG") ? atoi(getenv("MCVD_WINO_DIAG")) : 0;
    if (COT == 3 && PRO == 2 && diag) {
        switch (diag) {
#define W16_DIAG_CASE(D)                                                                                                   \
    case D:                                                                                                                \
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino16_kernel<3, 2, D>),                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                     \
        hipLaunchKernelGGL((conv_wino16_kernel<3, 2, D>), grid, dim3(W16_NT), lds, s, k);                                  \
        break;
            W16_DIAG_CASE(1) W16_DIAG_CASE(2) W16_DIAG_CASE(4) W16_DIAG_CASE(8) W16_DIAG_CASE(16) W16_DIAG_CASE(32)
            W16_DIAG_CASE(23) W16_DIAG_CASE(55) W16_DIAG_CASE(31) W16_DIAG_CASE(19) W16_DIAG_CASE(3)
#undef W16_DIAG_CASE
            default: break;
        }
        MCVD_HIP_CHECK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL((conv_wino16_kernel<COT, PRO>), grid, dim3(W16_NT), lds, s, k);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int COT>
static int wino16_launch(const ConvArgs& a, hipStream_t s) {
    if (!a.coef && !a.act) return wino16_launch2<COT, 0>(a, s);
    if (!a.act) return wino16_launch2<COT, 1>(a, s);
    return wino16_launch2<COT, 2>(a, s);
}

// Same contract as launch_conv_wino (conv_wino.cpp), which validates the arguments and forwards here.
int launch_conv_wino16(const ConvArgs& a, int cot, hipStream_t s) {
    MCVD_REQUIRE(a.Cin <= 1024 && wino16_lds_bytes(cot, a.Cin) <= 160 * 1024, "winograd conv: Cin=%d too large for the coefficient table", a.Cin);
    switch (cot) {
        case 1: return wino16_launch<1>(a, s);
        case 2: return wino16_launch<2>(a, s);
        default: return wino16_launch<3>(a, s);
    }
}

}  // namespace mcvd
