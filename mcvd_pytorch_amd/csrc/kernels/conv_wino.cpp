// 3x3 convolution by Winograd F(2x2, 3x3) on v_mfma_f32_32x32x2_f32 -- 2.25x fewer matrix-pipe multiplies than the direct
// implicit GEMM (conv_mfma.h), still exact-fp32 arithmetic (measured forward error vs fp64: 2.3e-6 against 1.8e-6 direct).
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, d = 4x4 input patch of silu?(A_c x + B_c)
//
// For each of the 16 transform positions xi the channel contraction is an independent GEMM
//   M_xi[co][tile] = sum_ci U_xi[co][ci] * V_xi[ci][tile]
// MFMA A operand = transformed weights U (rows = cout), B operand = transformed patches V (cols = tiles).
//
// Workgroup = 1024 threads = 16 waves = FOUR waves per SIMD, one workgroup per CU: a region of 8 x 16 output pixels = 4 x 8 = 32
// tiles of one sample, 32*COT output channels, all 16 positions; the transformed weights are fed to the matrix pipe
// STRAIGHT FROM GLOBAL MEMORY INTO REGISTERS:
//
// Wave w owns transform position xi = w (COT accumulator tiles = 48 registers at COT = 3, inside the 128-VGPR budget of four
// waves per SIMD).  Its MFMA A operand for (k-pair kp, cout sub-tile ct) is
//     lane l  ->  U_xi[co = ct*32 + (l & 31)][ci = 2*kp + (l >> 5)]
// i.e. ONE value per lane, and no other wave ever needs it: staging U through LDS (an earlier version: 48 KiB of DMA writes
// plus 48 KiB of operand reads per 8 channels and workgroup) buys nothing.  The weights are therefore packed per (cout tile,
// 8-channel unit, position) as [COT][64 lanes][4] floats -- float4 number = cout sub-tile, component = k-pair (pack_wino_weight_kernel) -- and every wave fetches its 4*COT A operands
// of an 8-channel unit with COT fully coalesced global_load_dwordx4 (L2-resident data), one unit ahead of its use.
//   * LDS holds only the transformed patches V (B operand) and the activated input patch: 16 input channels per chunk
//     (HALF the barriers of the 8-channel predecessor) in 96 KiB.
//   * per chunk and wave: 8 MFMA groups (k-pairs) of COT MFMAs; the LAST group is issued after the chunk barrier (its
//     operands are in registers), so the matrix pipe has work while the first B reads of the next chunk are in flight.
//   * staging per chunk: 2880 patch elements (<= 3 per thread, raw loads one chunk ahead, affine + SiLU from an LDS
//     coefficient table, zero padding after the activation); the tile transform B^T d B is split by row of B^T d over the four
//     256-thread wave groups, two (channel, tile) tasks per thread, and wave group g runs it in its own slot of the chunk so
//     that the four waves of a SIMD (one per group) are never in the same VALU-heavy phase; all of it sits in the first half of
//     the chunk, the second half is MFMAs only, so the waves reach the barrier together.
//   * every VMEM operation of the K loop is issued through inline asm and waited for with explicit in-order vmcnt counts:
//     the compiler's s_waitcnt insertion falls back to vmcnt(0) for loop-carried loads, which would serialise the weight
//     prefetch behind the patch loads.  The wait points and the counts are spelled out at each use below.  The compiler does not
//     know these registers are pending, so the build runs tools/check_wino_isa.py on the generated code (no read/copy/spill of
//     a destination register before a vmcnt wait, no foreign VMEM in the loop) whenever this file is recompiled.
//   * epilogue: per 32-cout sub-tile the 16 position planes go through LDS and every thread inverse-transforms one (cout, tile).
#include <stdlib.h>

#include "../common.h"

namespace mcvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float silu_wr(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

constexpr int WR_CK = 16;        // input channels per chunk (two 8-channel weight units)
constexpr int WR_T = 32;         // tiles per workgroup (4 x 8 tiles = 8 x 16 output pixels)
constexpr int WR_NT = 1024;
// LDS patch row pitch (18 columns used).  24: the tile transform reads 8-byte pairs at 2*PP*ty + 2*tx floats, 16 lanes per LDS
// cycle (ty in {0,1}, tx in 0..7): 2*PP = 48 = 16 mod 32 banks puts the two tile rows on disjoint banks (pitch 20 was a
// 2-way conflict on half the lanes: SQ_LDS_BANK_CONFLICT 29 % of the LDS-active cycles, profiles/r01_pmc_sq_wave_states.txt).
constexpr int WR_PP = 24;

// PRO: 0 raw input, 1 affine, 2 affine + SiLU, 3 SPADE: silu(((A x + B)(1 + gamma) + beta) * s1 + b2) with the per-pixel gamma | beta maps
//     of the conditioning frames (a.gb: [B][2*Cin][H][W], layerspp.py:164-171) and the temb pair (s1, b2) = (1 + scale, shift) of
//     a.coef2 ([B][Cin][2], NULL for the final norm: layerspp.py:530-535).  The maps never pass through registers: each thread's
//     gamma and beta of the patch elements it activates are fetched by LDS-DMA (global_load_lds_dword) into a wave-private LDS
//     slab, one chunk ahead like the patch itself, and read back by the same thread.
// G8: 8x8 images -- the 32 tiles of a workgroup are TWO whole images (16 tiles each); every halo element is zero padding, so
//     only the 2 x 64 interior pixels per channel are loaded (2 per thread and chunk) and the halo is zeroed once.
// a.ksplit == 2 (grid.y = 2): the workgroup contracts one half of the input channels and stores its raw partial result to
//     a.part[half]; wino_ksplit_reduce_kernel then forms y = s * (p0 + p1 + bias + res) in that fixed order (deterministic).
//     Used where the (region, cout tile) count alone cannot fill the 256 CUs (8x8 and some 16x16 layers).  An earlier
//     version added the halves into a zero-filled output with fp32 atomics: the memset + 3 M scalar atomics per launch cost
//     what the split gained.
// EXP != 0: timing-only ablations of the K loop (wrong results; env MCVD_WINO_EXP, tests/gpu_diag.py wexp): bit 0 no tile
//     transform (WRITE_V), bit 1 no patch activation/park (WRITE_P), bit 2 no VMEM in the loop, bit 3 no B-operand LDS reads,
//     bit 4 no MFMA, bit 5 no chunk barrier, bits 8-9: s_setprio scheme (1: prio = 3 - grp, 2: prio = grp, 3: all waves prio 1).
template <int COT, int PRO, bool G8, int EXP = 0>
__global__ __launch_bounds__(1024) void conv_wino_kernel(ConvArgs a) {
    constexpr int NT = WR_NT, CK = WR_CK, T = WR_T, BCO = 32 * COT;
    constexpr int VSZ = CK * 16 * T;            // floats per V chunk
    constexpr int PP = WR_PP;
    constexpr int PSZ = CK * 10 * PP;           // activated input patch of one chunk: [CK][10 rows][PP]
    constexpr int PBUF = PSZ + 4;               // + dump space for unused patch slots
    constexpr int PCOUNT = G8 ? CK * 2 * 64 : CK * 10 * 18;     // patch elements loaded per chunk
    constexpr int MAXP = (PCOUNT + NT - 1) / NT;                // 3 (2 for G8) loads per thread and chunk
    constexpr int VM_P = COT;                                   // vmcnt counts of the K loop, derived where they are used
    constexpr int VM_A = COT + MAXP * (PRO == 3 ? 3 : 1);       // PRO 3: + 2*MAXP gamma | beta DMA operations behind every patch load group
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sV = smem;                           // [2][VSZ]
    float* sP = smem + 2 * VSZ;                 // [2][PBUF]
    float* sCo = sP + 2 * PBUF;                 // [Cin][2] prologue coefficients (A_c, B_c) of this sample (PRO only)
    float* sC2 = sCo + (G8 ? 4 : 2) * a.Cin;    // PRO 3: [Cin][2] (s1, b2) of this sample (G8: + the second sample's)
    float* sGB = sC2 + (G8 ? 4 : 2) * a.Cin;    // PRO 3: [2][MAXP][NT] gamma | beta of the patch elements of ONE chunk, element e = sl*NT + tid

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int H = a.H, W = a.W, HW = H * W, Cin = a.Cin;
    const int rx_n = G8 ? 1 : W >> 4, ry_n = G8 ? 1 : H >> 3;
    const int nreg = G8 ? (a.B + 1) >> 1 : a.B * rx_n * ry_n;
    // block id -> (region, cout tile): the cout tiles of one region get ids congruent mod 8 and adjacent in dispatch order, i.e.
    // they run at the same time on the SAME XCD and share the region's input patch through that XCD's L2.
    const int nct = a.CoutP / BCO;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int reg_id = (slot / nct) * 8 + xcd;
    const int cotile = slot - (slot / nct) * nct;
    if (reg_id >= nreg) return;
    const int b = G8 ? 2 * reg_id : reg_id / (rx_n * ry_n);      // (first) sample of the region
    const int rr = G8 ? 0 : reg_id - b * (rx_n * ry_n);
    const int oy0 = (rr / rx_n) * 8, ox0 = (rr % rx_n) * 16;
    const int co0 = cotile * BCO;
    const int grp = wave >> 2;                  // row of B^T d this wave's threads make == pipeline phase of the wave

    // ---- transform role: tasks (channel-in-chunk, tile) = (tid & 255) and the same + 8 channels, row = grp
    const int s_ci = (tid & 255) >> 5, s_tile = tid & 31;
    // tile -> (row, column) of its 4x4 window's top-left corner in the LDS patch [ci][10 rows][PP]; G8: image i sits at columns 10*i
    const int s_ty = G8 ? (s_tile >> 2) & 3 : s_tile >> 3, s_tx = G8 ? (s_tile & 3) + 5 * (s_tile >> 4) : s_tile & 7;
    // row grp of B^T d:  0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3
    const int p_rd = s_ci * 10 * PP + 2 * s_ty * PP + 2 * s_tx;     // top-left of the 4x4 window in the LDS patch
    const int p_rdA = p_rd + (grp == 0 ? 0 : 1) * PP, p_rdB = p_rd + (grp == 3 ? 3 : 2) * PP;
    const float v_fa = grp == 2 ? -1.0f : 1.0f, v_fb = (grp == 1 || grp == 2) ? 1.0f : -1.0f;
    const int v_wr = s_ci * 16 * T + grp * 4 * T + s_tile;

    // ---- patch-load slots (chunk invariant); p_ci = channel-in-chunk, or CK + channel when the element is padding / unused
    int p_lds[MAXP], p_goff[MAXP], p_ci[MAXP];
    int p_img[MAXP];                            // G8: second image of the region (0 / 1), clamped to the last sample
#pragma unroll
    for (int sl = 0; sl < MAXP; ++sl) {
        const int e = sl * NT + tid;
        p_img[sl] = 0;
        if (G8) {                               // e -> (channel, image, row, col) of an interior pixel; always < PCOUNT
            const int ci = e >> 7, img = (e >> 6) & 1, r = (e >> 3) & 7, c = e & 7;
            const bool valid = b + img < a.B;
            p_lds[sl] = ci * 10 * PP + (r + 1) * PP + img * 10 + c + 1;
            p_goff[sl] = r * 8 + c;
            p_img[sl] = valid ? img : 0;
            p_ci[sl] = ci + (valid ? 0 : CK);
        } else if (e < PCOUNT) {
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

    // ---- weight fetch: unit u = 8 input channels; this wave's COT float4 per unit at  wr_base + u * (16 * COT * 256 floats)
    const int nunits = a.CinP / 8;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);       // provably uniform: the asm loads take an SGPR base
    const float* wr_base = a.wpw + (((long)cotile * nunits * 16 + wave_u) * COT) * 256;
    const unsigned wr_voff = (unsigned)lane * 16u;

    /* A operands of weight unit `u` -> register set S (COT float4: S[ct][kp] = operand of cout sub-tile ct, k-pair kp); asm: see header */
#define WR_LOAD_A(u, S)                                                                                         \
    {                                                                                                           \
        const float* ub = wr_base + (long)(u) * (16 * COT * 256);                                               \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(S[0]) : "v"(wr_voff), "s"(ub) : "memory");         \
        if (COT > 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(S[COT > 1 ? 1 : 0]) : "v"(wr_voff), "s"(ub) : "memory"); \
        if (COT > 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(S[COT > 2 ? 2 : 0]) : "v"(wr_voff), "s"(ub) : "memory"); \
    }
    /* wait until all but the N youngest VMEM operations of this wave have completed; the register set S is threaded through \
       the asm so that nothing reading it can be scheduled above the wait */                                               \
#define WR_WAIT_A(N, S)                                                                                         \
    {                                                                                                           \
        if (COT == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(S[0]) : "n"(N) : "memory");                     \
        if (COT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(S[0]), "+v"(S[COT > 1 ? 1 : 0]) : "n"(N) : "memory"); \
        if (COT == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(S[0]), "+v"(S[COT > 1 ? 1 : 0]), "+v"(S[COT > 2 ? 2 : 0]) : "n"(N) : "memory"); \
    }
    /* unconditional, clamped raw loads of the patch of chunk `ch` into pd[]; the chunk never straddles the concat seam      \
       (launch check: C0 % 16 == 0 when C1 > 0), channels past Cin re-read the last one and are zeroed at the write */         \
#define WR_LOAD_P(ch, D)                                                                                        \
    {                                                                                                           \
        const int cb = min((ch) * CK, Cin - 1);                                                                 \
        const int cmax = Cin - 1 - cb;                                                                          \
        const bool second = cb >= a.C0;                                                                         \
        const float* srcb = second ? a.x1 + ((long)b * a.C1 + (cb - a.C0)) * HW : a.x0 + ((long)b * a.C0 + cb) * HW; \
        const int istride = (second ? a.C1 : a.C0) * HW;      /* G8: distance to the region's second sample */  \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                                   \
            const unsigned off = (unsigned)(min(p_ci[sl] & (CK - 1), cmax) * HW + p_goff[sl] + (G8 ? p_img[sl] * istride : 0)) * 4u; \
            asm volatile("global_load_dword %0, %1, %2" : "=v"(D[sl]) : "v"(off), "s"(srcb) : "memory");        \
        }                                                                                                       \
    }
    /* PRO 3: gamma | beta of the patch elements of chunk `ch`, same clamped offsets as the patch, by LDS-DMA into this wave's slab \
       (2 * MAXP VMEM operations, counted by the vmcnt waits like the register loads) */                                    \
#define WR_LOAD_GB(ch)                                                                                          \
    if (PRO == 3) {                                                                                             \
        const int cb = min((ch) * CK, Cin - 1);                                                                 \
        const int cmax = Cin - 1 - cb;                                                                          \
        const float* gsrc = a.gb + ((long)b * 2 * Cin + cb) * HW;                                               \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                                   \
            const long off = (long)min(p_ci[sl] & (CK - 1), cmax) * HW + p_goff[sl] + (G8 ? (long)p_img[sl] * 2 * Cin * HW : 0); \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + off),       \
                                             (__attribute__((address_space(3))) void*)(sGB + sl * NT + wave_u * 64), 4, 0, 0); \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (long)Cin * HW + off), \
                                             (__attribute__((address_space(3))) void*)(sGB + (MAXP + sl) * NT + wave_u * 64), 4, 0, 0); \
        }                                                                                                       \
    }
#define WR_WAIT_P(N, D)                                                                                         \
    {                                                                                                           \
        if (MAXP == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(D[0]), "+v"(D[1]) : "n"(N) : "memory");        \
        if (MAXP == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(D[0]), "+v"(D[1]), "+v"(D[MAXP > 2 ? 2 : 0]) : "n"(N) : "memory"); \
    }
    /* activate once per pixel (coefficients from the LDS table) and park the patch in LDS; zero padding applies AFTER     \
       the activation */                                                                                           \
#define WR_WRITE_P(ch, D)                                                                                        \
    {                                                                                                           \
        float* sPw = sP + (((ch) & 1) ? PBUF : 0);                                                              \
        const int nvalid = Cin - (ch) * CK;                                                                     \
        f32x2 cfv[MAXP];                     /* all coefficient reads first: ONE LDS round trip, not MAXP */     \
        if (PRO >= 1) {                                                                                         \
            _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                               \
                const int cch = min((ch) * CK + (p_ci[sl] & (CK - 1)), Cin - 1) + (G8 ? p_img[sl] * Cin : 0);   \
                cfv[sl] = *reinterpret_cast<const f32x2*>(sCo + cch * 2);                                       \
            }                                                                                                   \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cfv[0]), "+v"(cfv[1]), "+v"(cfv[MAXP > 2 ? 2 : 0]) :: "memory"); \
        }                                                                                                       \
        f32x2 c2v[MAXP];                                                                                        \
        float gmv[MAXP], btv[MAXP];                                                                             \
        if (PRO == 3) {                                                                                         \
            _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                               \
                const int cch = min((ch) * CK + (p_ci[sl] & (CK - 1)), Cin - 1) + (G8 ? p_img[sl] * Cin : 0);   \
                c2v[sl] = *reinterpret_cast<const f32x2*>(sC2 + cch * 2);                                       \
                gmv[sl] = sGB[sl * NT + tid];                                                                   \
                btv[sl] = sGB[(MAXP + sl) * NT + tid];                                                          \
            }                                                                                                   \
        }                                                                                                       \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                                   \
            float v = D[sl];                                                                                    \
            if (PRO >= 1) v = v * cfv[sl].x + cfv[sl].y;                                                        \
            if (PRO == 3) v = (v * (1.0f + gmv[sl]) + btv[sl]) * c2v[sl].x + c2v[sl].y;   /* spade_apply_kernel's order */ \
            if (PRO >= 2) v = silu_wr(v);                                                                       \
            sPw[p_lds[sl]] = (p_ci[sl] < min(nvalid, CK)) ? v : 0.0f;                                           \
        }                                                                                                       \
    }
    /* row grp of B^T d (rows RA, RB of the window, combined with wave-uniform +-1 factors), then (.) B: four position     \
       values -> V(ch)[ci][grp*4 + j][tile]; two (channel, tile) tasks per thread (channels s_ci and s_ci + 8) */            \
#define WR_WRITE_V(ch)                                                                                          \
    {                                                                                                           \
        const float* sPr = sP + (((ch) & 1) ? PBUF : 0);                                                        \
        float* vdst = sV + (((ch) & 1) ? VSZ : 0) + v_wr;                                                       \
        _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2) {                                                      \
            float m[4];                                                                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
                const float va = sPr[p_rdA + k2 * 8 * 10 * PP + j], vb = sPr[p_rdB + k2 * 8 * 10 * PP + j];     \
                m[j] = __builtin_fmaf(v_fb, vb, v_fa * va);  /* +-va +- vb, exact: the factors are +-1 */        \
            }                                                                                                   \
            vdst[k2 * 8 * 16 * T + 0 * T] = m[0] - m[2];                                                        \
            vdst[k2 * 8 * 16 * T + 1 * T] = m[1] + m[2];                                                        \
            vdst[k2 * 8 * 16 * T + 2 * T] = m[2] - m[1];                                                        \
            vdst[k2 * 8 * 16 * T + 3 * T] = m[1] - m[3];                                                        \
        }                                                                                                       \
    }
    /* B operand of MFMA group g (k-pair g of the 16-channel chunk): V[ci = 2g + half][xi = wave][tile l31] */
#define WR_LOAD_B(g, BV) BV = sVc[((2 * (g) + half) * 16 + wave) * T + l31];
    /* MFMA group: k-pair kp (0..3) of the weight unit held in register set S */
#define WR_DO_MFMA(kp, BV, S)                                                                                   \
    _Pragma("unroll") for (int ct = 0; ct < COT; ++ct)                                                          \
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(S[ct][kp], BV, acc[ct], 0, 0, 0);

    f32x16 acc[COT];
#pragma unroll
    for (int ct = 0; ct < COT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;

    // diagnostics (mcvd_ctx_set_debug_buffer): shader-clock time the wave a.wdma spends per phase
    const bool rec = a.dbg != nullptr && wave == a.wdma;
    unsigned long long tk0 = 0, tprev = 0, dt[2] = {0, 0}, pt[3] = {0, 0, 0};
    if (rec) tk0 = tprev = __builtin_amdgcn_s_memtime();
#define WR_STAMP(i)                                                                                             \
    if (rec) {                                                                                                  \
        const unsigned long long now = __builtin_amdgcn_s_memtime();                                            \
        dt[i] += now - tprev;                                                                                   \
        tprev = now;                                                                                            \
    }

    // ---- chunk range of this workgroup (a.ksplit == 2: blockIdx.y picks one half of the input channels)
    const int nch_all = a.CinP / CK;
    const int ksp = a.ksplit == 2 ? 2 : 1, kh = ksp == 2 ? (int)blockIdx.y : 0;
    const int c_begin = kh * (nch_all / ksp), c_end = c_begin + nch_all / ksp;

    // ---- prologue: every global load of the first chunks + the coefficient table is issued before anything waits
    f32x4 A0[COT], A1[COT];                // A operands of the even / odd weight unit (first / second half of a chunk)
    float pd[MAXP];                        // raw patch registers, loaded one chunk ahead of their activation
    {
        float q0[MAXP], q1[MAXP];          // patches of the first two chunks: prologue only
        f32x2 cfl = {1.0f, 0.0f};
        WR_LOAD_A(2 * c_begin, A0)
        WR_LOAD_A(2 * c_begin + 1, A1)
        WR_LOAD_P(c_begin, q0)
        WR_LOAD_P(c_begin + 1, q1)
        WR_LOAD_P(c_begin + 2, pd)
        WR_LOAD_GB(c_begin)
        if (PRO && a.coef && tid < Cin) cfl = *reinterpret_cast<const f32x2*>(a.coef + ((long)b * Cin + tid) * 2);
        if (PRO && tid < Cin) *reinterpret_cast<f32x2*>(sCo + tid * 2) = cfl;
        if (PRO == 3 && tid < Cin) {           // (s1, b2): the temb pair, (1, 0) where the norm has none (final SPADE norm)
            f32x2 c2 = {1.0f, 0.0f};
            if (a.coef2) c2 = *reinterpret_cast<const f32x2*>(a.coef2 + ((long)b * Cin + tid) * 2);
            *reinterpret_cast<f32x2*>(sC2 + tid * 2) = c2;
            if (G8) {
                if (a.coef2 && b + 1 < a.B) c2 = *reinterpret_cast<const f32x2*>(a.coef2 + ((long)(b + 1) * Cin + tid) * 2);
                *reinterpret_cast<f32x2*>(sC2 + (Cin + tid) * 2) = c2;
            }
        }
        if (G8) {                          // the halo of both patch buffers is zero padding for the whole kernel
            for (int i = tid; i < 2 * PBUF; i += NT) sP[i] = 0.0f;
            if (PRO && a.coef && tid < Cin && b + 1 < a.B)     // second sample's coefficients: table rows Cin .. 2*Cin-1
                cfl = *reinterpret_cast<const f32x2*>(a.coef + ((long)(b + 1) * Cin + tid) * 2);
            if (PRO && tid < Cin) *reinterpret_cast<f32x2*>(sCo + (Cin + tid) * 2) = cfl;
        }
        if (rec) pt[0] = __builtin_amdgcn_s_memtime() - tk0;           // index setup + load issue
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ONE memory latency for everything above (the asm loads are
        WR_WAIT_A(0, A0)                                     // not tracked by the compiler: the waits are threaded through
        WR_WAIT_A(0, A1)                                     // the destination registers)
        WR_WAIT_P(0, q0)
        WR_WAIT_P(0, q1)
        WR_WAIT_P(0, pd)
        if (rec) pt[1] = __builtin_amdgcn_s_memtime() - tk0;           // ... + memory latency
        if (PRO || G8) __syncthreads();    // coefficient table (and the zeroed halo) visible
        WR_WRITE_P(c_begin, q0)
        if (PRO == 3) {                    // the gamma | beta slab holds ONE chunk: fetch the second chunk's now (one exposed latency
            WR_LOAD_GB(c_begin + 1)        // per workgroup), the third's behind it for the first loop iteration
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        WR_WRITE_P(c_begin + 1, q1)
        WR_LOAD_GB(c_begin + 2)
    }
    __syncthreads();                       // the first two patches visible
    if (rec) pt[2] = __builtin_amdgcn_s_memtime() - tk0;               // ... + table barrier + two patch blocks + barrier
    WR_WRITE_V(c_begin)
    __syncthreads();                       // V of the first chunk visible
    if (PRO == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the loop's first wait leaves COT operations in flight
    WR_STAMP(0)

    // ---- K loop.  VMEM issue order of a wave in chunk c (in-order vmcnt counter; nothing else is outstanding):
    //   top:  A1 <- unit 2c+1 (COT loads), patch(c+3) (MAXP loads)   mid (after group 3):  A0 <- unit 2c+2 (COT loads)
    // wait points:  patch(c+2) write at the top, before the new loads: all but the COT loads of `mid` of chunk c-1  (VM_P = COT)
    //               first use of A1 (group 4): issued at the top, younger = MAXP patch + COT mid loads            (VM_A)
    //               first use of A0 (group 0 of the next chunk): issued at mid, younger = COT + MAXP loads of that chunk's top
    float xb = 0.0f, yb = 0.0f;            // B operands; yb = 0: the first "deferred" group multiplies zeros
    if (EXP != 0) {        // ablation build of the loop: same order, pieces compiled out (timing only)
        if (((EXP >> 8) & 3) == 1) { if (grp == 0) __builtin_amdgcn_s_setprio(3); if (grp == 1) __builtin_amdgcn_s_setprio(2); if (grp == 2) __builtin_amdgcn_s_setprio(1); }
        if (((EXP >> 8) & 3) == 2) { if (grp == 3) __builtin_amdgcn_s_setprio(3); if (grp == 2) __builtin_amdgcn_s_setprio(2); if (grp == 1) __builtin_amdgcn_s_setprio(1); }
        if (((EXP >> 8) & 3) == 3) __builtin_amdgcn_s_setprio(1);
#define X_V(ch) if (!(EXP & 1)) WR_WRITE_V(ch)
#define X_P(ch, D) if (!(EXP & 2)) WR_WRITE_P(ch, D)
#define X_LA(u, S) if (!(EXP & 4)) WR_LOAD_A(u, S)
#define X_LP(ch, D) if (!(EXP & 4)) WR_LOAD_P(ch, D)
#define X_WA(N, S) if (!(EXP & 4)) WR_WAIT_A(N, S)
#define X_WP(N, D) if (!(EXP & 4)) WR_WAIT_P(N, D)
#define X_B(g, BV) if (!(EXP & 8)) WR_LOAD_B(g, BV)
#define X_M(kp, BV, S) if (!(EXP & 16)) WR_DO_MFMA(kp, BV, S)
        for (int c = c_begin; c + 1 < c_end; ++c) {
            const float* sVc = sV + ((c & 1) ? VSZ : 0);
            X_M(3, yb, A1)
            X_B(0, xb)
            X_WP(VM_P, pd)
            X_P(c + 2, pd)
            X_LA(2 * c + 1, A1)
            X_LP(c + 3, pd)
            if (grp == 3) X_V(c + 1)
            X_WA(VM_A, A0)
            X_M(0, xb, A0)
            X_B(1, yb)
            if (grp == 2) X_V(c + 1)
            X_M(1, yb, A0)
            X_B(2, xb)
            if (grp == 0) X_V(c + 1)
            X_M(2, xb, A0)
            X_B(3, yb)
            if (grp == 1) X_V(c + 1)
            X_M(3, yb, A0)
            X_B(4, xb)
            X_LA(2 * c + 2, A0)
            X_WA(VM_A, A1)
            X_M(0, xb, A1)
            X_B(5, yb)
            X_M(1, yb, A1)
            X_B(6, xb)
            X_M(2, xb, A1)
            X_B(7, yb)
            if (!(EXP & 32)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(0);
#undef X_V
#undef X_P
#undef X_LA
#undef X_LP
#undef X_WA
#undef X_WP
#undef X_B
#undef X_M
    } else
    for (int c = c_begin; c + 1 < c_end; ++c) {
        const float* sVc = sV + ((c & 1) ? VSZ : 0);
        WR_DO_MFMA(3, yb, A1)              // group 7 of the previous chunk (operands were read before the barrier)
        WR_LOAD_B(0, xb)
        WR_WAIT_P(VM_P, pd)
        WR_WRITE_P(c + 2, pd)
        WR_LOAD_A(2 * c + 1, A1)
        WR_LOAD_P(c + 3, pd)
        WR_LOAD_GB(c + 3)
        if (grp == 3) WR_WRITE_V(c + 1)    // slot T
        WR_WAIT_A(VM_A, A0)
        WR_DO_MFMA(0, xb, A0)              // g0
        WR_LOAD_B(1, yb)
        if (grp == 2) WR_WRITE_V(c + 1)
        WR_DO_MFMA(1, yb, A0)              // g1
        WR_LOAD_B(2, xb)
        if (grp == 0) WR_WRITE_V(c + 1)
        WR_DO_MFMA(2, xb, A0)              // g2
        WR_LOAD_B(3, yb)
        if (grp == 1) WR_WRITE_V(c + 1)
        WR_DO_MFMA(3, yb, A0)              // g3
        WR_LOAD_B(4, xb)
        WR_LOAD_A(2 * c + 2, A0)           // mid: first unit of the next chunk
        WR_WAIT_A(VM_A, A1)
        WR_DO_MFMA(0, xb, A1)              // g4
        WR_LOAD_B(5, yb)
        WR_DO_MFMA(1, yb, A1)              // g5
        WR_LOAD_B(6, xb)
        WR_DO_MFMA(2, xb, A1)              // g6
        WR_LOAD_B(7, yb)                   // g7 is deferred past the barrier
        // chunk c read by every wave; V(c+1), patch(c+2) visible.  LDS traffic only: no VMEM wait at the barrier.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    {
        const int c = c_end - 1;
        const float* sVc = sV + ((c & 1) ? VSZ : 0);
        WR_DO_MFMA(3, yb, A1)
        WR_LOAD_A(2 * c + 1, A1)
        WR_LOAD_B(0, xb)
        WR_WAIT_A(VM_P, A0)
        WR_DO_MFMA(0, xb, A0)
        WR_LOAD_B(1, yb)
        WR_DO_MFMA(1, yb, A0)
        WR_LOAD_B(2, xb)
        WR_DO_MFMA(2, xb, A0)
        WR_LOAD_B(3, yb)
        WR_DO_MFMA(3, yb, A0)
        WR_LOAD_B(4, xb)
        WR_WAIT_A(0, A1)
        WR_DO_MFMA(0, xb, A1)
        WR_LOAD_B(5, yb)
        WR_DO_MFMA(1, yb, A1)
        WR_LOAD_B(6, xb)
        WR_DO_MFMA(2, xb, A1)
        WR_LOAD_B(7, yb)
        WR_DO_MFMA(3, yb, A1)
    }
    // stray patch prefetches past the last chunk: pd stays reserved (an operand of this wait) until they have landed -- to the
    // compiler the registers are dead after the loop, and anything it put there would be overwritten by the late loads
    WR_WAIT_P(0, pd)

    // ---------------- inverse transform + epilogue, one 32-cout sub-tile at a time ----------------
    // The barriers below guard LDS only (s_waitcnt lgkmcnt(0) + s_barrier): the residual loads of the NEXT sub-tile are issued
    // a round ahead and stay in flight across them.
    float* sM = smem;                      // [16 positions][32 couts][32 tiles] = 64 KiB
    const int e_tile = tid & 31, e_col = tid >> 5;
    const int e_ty = G8 ? (e_tile >> 2) & 3 : e_tile >> 3, e_tx = G8 ? e_tile & 3 : e_tile & 7;
    const int e_b = min(b + (G8 ? e_tile >> 4 : 0), a.B - 1);          // G8: the tile's sample (clamped for the loads)
    const bool e_valid = !G8 || b + (e_tile >> 4) < a.B;
    const long pix = (long)(oy0 + 2 * e_ty) * W + ox0 + 2 * e_tx;
    const bool fin = ksp == 1;                 // K split: bias, residual and scale are applied by the reduce kernel
    float* const ydst = fin ? a.y : a.part + (long)kh * a.B * a.Cout * HW;
    f32x2 rn0 = {0.0f, 0.0f}, rn1 = {0.0f, 0.0f};
#define WR_LOAD_RES(ct)                                                                                         \
    if (a.res && fin) {                                                                                        \
        const long o = ((long)e_b * a.Cout + min(co0 + (ct) * 32 + e_col, a.Cout - 1)) * HW + pix;              \
        rn0 = *reinterpret_cast<const f32x2*>(a.res + o);                                                       \
        rn1 = *reinterpret_cast<const f32x2*>(a.res + o + W);                                                   \
    }
    WR_LOAD_RES(0)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the K loop is done with the LDS
    WR_STAMP(1)
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = (r & 3) + 8 * (r >> 2) + 4 * half;
            sM[(wave * 32 + col) * T + l31] = acc[ct][r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            const f32x2 r0 = rn0, r1 = rn1;
            if (ct + 1 < COT) WR_LOAD_RES(ct + 1)
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
            const float bvv = fin ? a.bias[co] : 0.0f;          // zero-padded to CoutP
            const float osc = fin ? a.out_scale : 1.0f;
            const float v00 = (y00 + bvv + r0.x) * osc, v01 = (y01 + bvv + r0.y) * osc;
            const float v10 = (y10 + bvv + r1.x) * osc, v11 = (y11 + bvv + r1.y) * osc;
            if (co < a.Cout && e_valid) {
                const long o = ((long)e_b * a.Cout + co) * HW + pix;
                *reinterpret_cast<float2*>(ydst + o) = make_float2(v00, v01);
                *reinterpret_cast<float2*>(ydst + o + W) = make_float2(v10, v11);
            }
            if (a.stats && fin) {
                // GroupNorm partials of the FINAL values for the next norm (ConvArgs::stats).  The 32 tiles of this cout are the 32
                // lanes of a half-wave (G8: 16 lanes = one DPP row per image).  Each lane folds its own 2x2 pixels into (mean, M2)
                // exactly, then the lanes are merged pairwise with the equal-count update  M2 = M2a + M2b + (ma - mb)^2 * n/2,
                // mean = (ma + mb)/2  over DPP moves (no LDS traffic): xor 1, xor 2 inside quads, rotate 4, rotate 8 inside the
                // 16-lane row (every lane then holds the row total), row_bcast:15 into the upper row of the half-wave.
                float mu = 0.25f * ((v00 + v01) + (v10 + v11));
                const float d0 = v00 - mu, d1 = v01 - mu, d2 = v10 - mu, d3 = v11 - mu;
                float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                float hn = 2.0f;                      // n/2 of the two partials being merged (4 pixels each at the first step)
#define WR_MERGE(CTRL, ROWMASK)                                                                                     \
                {                                                                                                   \
                    const float mo = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, mu), __builtin_bit_cast(int, mu), CTRL, ROWMASK, 0xf, false)); \
                    const float qo = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m2), __builtin_bit_cast(int, m2), CTRL, ROWMASK, 0xf, false)); \
                    const float dd = mu - mo;                                                                       \
                    m2 = (m2 + qo) + dd * dd * hn;                                                                  \
                    mu = 0.5f * (mu + mo);                                                                          \
                    hn += hn;                                                                                       \
                }
                WR_MERGE(0xB1, 0xf)                   // quad_perm [1,0,3,2]
                WR_MERGE(0x4E, 0xf)                   // quad_perm [2,3,0,1]
                WR_MERGE(0x124, 0xf)                  // row_ror:4
                WR_MERGE(0x128, 0xf)                  // row_ror:8   -> 16 lanes = 64 pixels merged, in every lane of the row
                if (!G8) WR_MERGE(0x142, 0xa)         // row_bcast:15 (rows 1 and 3 take the total of rows 0 and 2): lanes 16-31 / 48-63
#undef WR_MERGE
                constexpr int NPIX = G8 ? 64 : 128;
                const bool writer = G8 ? (e_tile & 15) == 0 : e_tile == 31;
                if (writer && co < a.Cout && e_valid) {
                    const int np = G8 ? 1 : rx_n * ry_n;
                    float* q = a.stats + (((long)e_b * a.Cout + co) * np + rr) * 2;
                    q[0] = mu * (float)NPIX;          // the partial's sum
                    q[1] = m2;
                }
            }
        }
        if (ct + 1 < COT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#undef WR_LOAD_RES
    if (rec) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long* d = a.dbg + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 8;
            d[0] = dt[0]; d[1] = dt[1]; d[2] = pt[0]; d[3] = pt[1]; d[4] = pt[2];      // prologue, K loop, prologue sub-stamps
            d[5] = now - tprev;            // epilogue
            d[6] = (unsigned long long)(c_end - c_begin);
            d[7] = now - tk0;
        }
    }
#undef WR_STAMP
#undef WR_LOAD_A
#undef WR_WAIT_A
#undef WR_LOAD_P
#undef WR_LOAD_GB
#undef WR_WAIT_P
#undef WR_WRITE_P
#undef WR_WRITE_V
#undef WR_LOAD_B
#undef WR_DO_MFMA
}

// y = s * (p0 + p1 + bias[c] + res): the second pass of the 2-way K split, 4 pixels per thread
__global__ __launch_bounds__(256) void wino_ksplit_reduce_kernel(const float* part, const float* bias, const float* res, float scale,
                                                                 float* y, long n4, long half_stride, int Cout, int HW4, int ksp) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 acc = reinterpret_cast<const f32x4*>(part)[i];
        for (int k = 1; k < ksp; ++k) acc = acc + reinterpret_cast<const f32x4*>(part + k * half_stride)[i];      // p0 + p1 + ... in this fixed order
        const float bv = bias[(i / HW4) % Cout];
        f32x4 v = acc + bv;
        if (res) v = v + reinterpret_cast<const f32x4*>(res)[i];
        reinterpret_cast<f32x4*>(y)[i] = v * scale;
    }
}

// The same pass with GroupNorm partials of the final values (ConvArgs::stats): one (sample, channel) plane = HW4 consecutive
// float4 of the flat index = PL lanes of one wave (HW4 = 16 or 64), one partial per plane over its HW pixels.
template <int PL>
__global__ __launch_bounds__(256) void wino_ksplit_reduce_stats_kernel(const float* part, const float* bias, const float* res, float scale,
                                                                       float* y, long n4, long half_stride, int Cout, float* stats, int ksp) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    const bool in = i < n4;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (in) {
        f32x4 acc = reinterpret_cast<const f32x4*>(part)[i];
        for (int k = 1; k < ksp; ++k) acc = acc + reinterpret_cast<const f32x4*>(part + k * half_stride)[i];      // p0 + p1 + ... in this fixed order
        const float bv = bias[(i / PL) % Cout];
        v = acc + bv;
        if (res) v = v + reinterpret_cast<const f32x4*>(res)[i];
        v = v * scale;
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
    float sm = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int o = PL / 2; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
    const float mu = sm * (1.0f / (4 * PL));
    const float d0 = v[0] - mu, d1 = v[1] - mu, d2 = v[2] - mu, d3 = v[3] - mu;
    float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
    for (int o = PL / 2; o > 0; o >>= 1) m2 += __shfl_xor(m2, o);
    if (in && (threadIdx.x & (PL - 1)) == 0) {          // a plane is never split between valid and invalid lanes (n4 % PL == 0)
        stats[(i / PL) * 2] = sm;
        stats[(i / PL) * 2 + 1] = m2;
    }
}

static size_t wino_lds_bytes(int Cin, bool g8, bool spade = false) {
    const int maxp = g8 ? 2 : 3;               // MAXP of the kernel
    const size_t k = (size_t)(2 * WR_CK * 16 * WR_T + 2 * (WR_CK * 10 * WR_PP + 4) + (g8 ? 4 : 2) * Cin * (spade ? 2 : 1) +
                              (spade ? 2 * maxp * WR_NT : 0)) * sizeof(float);
    const size_t epi = (size_t)16 * 32 * WR_T * sizeof(float);        // sM of the epilogue
    return k > epi ? k : epi;
}

#ifdef MCVD_DIAG
template <int EXP>
static int wino_launch_exp1(const ConvArgs& k, dim3 grid, size_t lds, hipStream_t s) {
    static PerDeviceOnce raised;
    if (raised.first_use()) {
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<3, 2, false, EXP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised.done();
    }
    hipLaunchKernelGGL((conv_wino_kernel<3, 2, false, EXP>), grid, dim3(WR_NT), lds, s, k);
    return 0;
}
static int wino_launch_exp(int e, const ConvArgs& k, dim3 grid, size_t lds, hipStream_t s) {
    switch (e) {
        case 1: return wino_launch_exp1<1>(k, grid, lds, s);          // no transform
        case 2: return wino_launch_exp1<2>(k, grid, lds, s);          // no patch activation / park
        case 3: return wino_launch_exp1<3>(k, grid, lds, s);          // neither
        case 4: return wino_launch_exp1<4>(k, grid, lds, s);          // no VMEM in the loop
        case 7: return wino_launch_exp1<7>(k, grid, lds, s);          // MFMA + B reads + barrier only
        case 15: return wino_launch_exp1<15>(k, grid, lds, s);        // MFMA + barrier only
        case 47: return wino_launch_exp1<47>(k, grid, lds, s);        // MFMA only, no barrier
        case 16: return wino_launch_exp1<16>(k, grid, lds, s);        // everything but the MFMAs
        case 32: return wino_launch_exp1<32>(k, grid, lds, s);        // no chunk barrier (racy)
        case 17: return wino_launch_exp1<17>(k, grid, lds, s);        // no MFMA, no transform
        case 18: return wino_launch_exp1<18>(k, grid, lds, s);        // no MFMA, no WRITE_P
        case 19: return wino_launch_exp1<19>(k, grid, lds, s);        // no MFMA, neither
        case 20: return wino_launch_exp1<20>(k, grid, lds, s);        // no MFMA, no VMEM
        case 23: return wino_launch_exp1<23>(k, grid, lds, s);        // barrier only (+ loop overhead)
        case 22: return wino_launch_exp1<22>(k, grid, lds, s);        // no MFMA, no VMEM, no WRITE_P: transform only
        case 21: return wino_launch_exp1<21>(k, grid, lds, s);        // no MFMA, no VMEM, no transform: WRITE_P only
        case 256: return wino_launch_exp1<256>(k, grid, lds, s);      // s_setprio 3 - grp
        case 512: return wino_launch_exp1<512>(k, grid, lds, s);      // s_setprio grp
        case 768: return wino_launch_exp1<768>(k, grid, lds, s);      // s_setprio 1 everywhere
        default: mcvd::set_error("MCVD_WINO_EXP=%d is not a built ablation", e); return -1;
    }
}
#endif

// Second pass of the K split (a.ksplit = 2 parts in conv_wino.cpp / conv_wino2h.cpp; 2, 4 or 8 in conv_wino3.cpp / conv_wino3p.cpp):
// y = s * (p0 + p1 + ... + bias + res) in that fixed order, with the GroupNorm partials of the final values where the plane size
// allows (one partial per (sample, channel) plane).
int launch_wino_ksplit_reduce(const ConvArgs& a, hipStream_t s) {
    const int ksp = a.ksplit >= 2 ? a.ksplit : 2;
    const long n = (long)a.B * a.Cout * a.H * a.W, n4 = n / 4;
    const int hw4 = a.H * a.W / 4;
    if (a.gno.coef && ksplit_reduce_gn_usable(a)) {       // ... and the (A, B) table of the norm over y: one workgroup per (sample, group), gn.cpp
        if (int rc = launch_ksplit_reduce_gn(a, s)) return rc;
        set_last_conv_stats_np(1);
        set_last_conv_gn_fused(1);
        return 0;
    }
    if (a.stats && (hw4 == 16 || hw4 == 64)) {     // ... with the GroupNorm partials of the final values (one per plane)
        const int blocks = (int)((n4 + 255) / 256);
        if (hw4 == 16)
            hipLaunchKernelGGL(wino_ksplit_reduce_stats_kernel<16>, dim3(blocks), dim3(256), 0, s, a.part, a.bias, a.res, a.out_scale,
                               a.y, n4, n, a.Cout, a.stats, ksp);
        else
            hipLaunchKernelGGL(wino_ksplit_reduce_stats_kernel<64>, dim3(blocks), dim3(256), 0, s, a.part, a.bias, a.res, a.out_scale,
                               a.y, n4, n, a.Cout, a.stats, ksp);
        set_last_conv_stats_np(1);
    } else {
        const int blocks = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
        hipLaunchKernelGGL(wino_ksplit_reduce_kernel, dim3(blocks), dim3(256), 0, s, a.part, a.bias, a.res, a.out_scale, a.y, n4, n,
                           a.Cout, hw4, ksp);
        if (a.stats) set_last_conv_stats_np(0);     // no partials for planes above 16x16: the consumer reduces the tensor itself
    }
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int COT, int PRO, bool G8>
static int wino_launch3(const ConvArgs& a, hipStream_t s) {
    constexpr int BCO = 32 * COT;
    const size_t lds = wino_lds_bytes(a.Cin, G8, PRO == 3);
    static PerDeviceOnce raised;
    if (raised.first_use()) {
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<COT, PRO, G8>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised.done();
    }
    const int nreg = G8 ? (a.B + 1) / 2 : a.B * (a.H / 8) * (a.W / 16);
    const int ksp = a.ksplit == 2 ? 2 : 1;
    dim3 grid(((nreg + 7) / 8) * 8 * (a.CoutP / BCO), ksp);
    ConvArgs k = a;
    if (k.dbg) k.wdma = 0;                 // wave 0 records its phase times
#ifdef MCVD_DIAG
    // diagnostics build only (build.py --diag): which wave records, and the timing-only ablations of the K loop (WRONG RESULTS; the
    // production library has neither the env hooks nor the ablation kernels)
    if (k.dbg) {
        const char* w = getenv("MCVD_DBG_WAVE");
        k.wdma = w ? atoi(w) : 0;
    }
    const char* exp_s = getenv("MCVD_WINO_EXP");
    const int exp_env = exp_s ? atoi(exp_s) : 0;
    if (COT == 3 && PRO == 2 && !G8 && exp_env != 0) {      // tests/gpu_diag.py wexp
        if (int rc = wino_launch_exp(exp_env, k, grid, lds, s)) return rc;
    } else
#endif
    hipLaunchKernelGGL((conv_wino_kernel<COT, PRO, G8>), grid, dim3(WR_NT), lds, s, k);
    MCVD_HIP_CHECK(hipGetLastError());
    if (ksp == 2) {
        if (int rc = launch_wino_ksplit_reduce(a, s)) return rc;
    } else if (a.stats) {
        set_last_conv_stats_np(G8 ? 1 : (a.H / 8) * (a.W / 16));
    }
    return 0;
}

template <int COT, int PRO>
static int wino_launch2(const ConvArgs& a, hipStream_t s) {
    return (a.H == 8 && a.W == 8) ? wino_launch3<COT, PRO, true>(a, s) : wino_launch3<COT, PRO, false>(a, s);
}

template <int COT>
static int wino_launch(const ConvArgs& a, hipStream_t s) {
    if (a.gb) {
        MCVD_REQUIRE(a.coef && a.act, "winograd conv: the SPADE prologue needs the GroupNorm coefficients and SiLU");
        return wino_launch2<COT, 3>(a, s);
    }
    if (!a.coef && !a.act) return wino_launch2<COT, 0>(a, s);
    if (!a.act) return wino_launch2<COT, 1>(a, s);
    return wino_launch2<COT, 2>(a, s);
}

// Geometry the kernel serves: regions of 8 x 16 output pixels, or whole 8 x 8 images in pairs.
bool conv_wino_supported(int ks, int H, int W) { return ks == 3 && ((H % 8 == 0 && W % 16 == 0 && H >= 8 && W >= 16) || (H == 8 && W == 8)); }

int conv_wino_cout_tile(int Cout) {
    if (Cout % 96 == 0) return 3;
    if (Cout % 64 == 0) return 2;
    return 1;
}

// Shape ids 4 / 8 apply to this launch: geometry, channel layout, packed weights present (8: and an even chunk count).
bool conv_wino_usable(const ConvArgs& a) {
    return conv_wino_supported(a.ks, a.H, a.W) && a.wpw && a.Cin <= 1024 && a.CinP % WR_CK == 0 &&
           (a.C1 == 0 || a.C0 % WR_CK == 0) &&                                       // a chunk never straddles the concat seam
           (long)a.B * (a.C0 > a.C1 ? a.C0 : a.C1) * a.H * a.W < (1L << 29) &&      // 32-bit byte offsets of the patch loads
           wino_lds_bytes(a.Cin, a.H == 8 && a.W == 8, a.gb != nullptr) <= 160 * 1024 &&
           (a.ksplit != 2 || ((a.CinP / WR_CK) % 2 == 0 && a.CinP / WR_CK >= 4 && conv_part_fits(a)));
}

// a.wpw must hold the operand-major layout (launch_pack_wino_weight) packed for conv_wino_cout_tile(Cout).
int launch_conv_wino(const ConvArgs& a, hipStream_t s) {
    MCVD_REQUIRE(conv_wino_usable(a), "winograd conv: unsupported (ks=%d H=%d W=%d Cin=%d C0=%d ksplit=%d, packed weights %s)", a.ks,
                 a.H, a.W, a.Cin, a.C0, a.ksplit, a.wpw ? "present" : "missing");
    const int cot = conv_wino_cout_tile(a.Cout);
    MCVD_REQUIRE(a.CoutP % (32 * cot) == 0, "winograd conv: CoutP=%d vs tile %d", a.CoutP, 32 * cot);
    switch (cot) {
        case 1: return wino_launch<1>(a, s);
        case 2: return wino_launch<2>(a, s);
        default: return wino_launch<3>(a, s);
    }
}

// U = G g G^T per (cout, cin), stored operand-major:
//   up[((((cotile*nunits + ci/8)*16 + xi)*COT + ct)*64 + lane)*4 + kp],   ct = (co%BCO)/32, kp = (ci%8)/2,
//   lane = (ci%2)*32 + co%32.   The buffer (CinP*16*CoutP floats) must be zero-filled: padded channels stay zero.
__global__ void pack_wino_weight_kernel(const float* w, float* up, int Cout, int Cin, int CinP, int CoutP, int COT) {
    const long n = (long)Cout * Cin;
    const int BCO = 32 * COT, nunits = CinP / 8;
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
        const int cotile = co / BCO, ct = (co % BCO) / 32, lane = (ci & 1) * 32 + (co & 31);
        const int idx = ct * 4 + ((ci & 7) >> 1);             // float4 number = cout sub-tile, component = k-pair
        const long base = ((((long)cotile * nunits + (ci >> 3)) * 16) * COT + (idx >> 2)) * 256 + lane * 4 + (idx & 3);
#pragma unroll
        for (int r = 0; r < 4; ++r) {                           // (.) G^T
            const float u0 = t[r][0];
            const float u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
            const float u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
            const float u3 = t[r][2];
            const long stride = (long)COT * 256;                // one position further
            up[base + (r * 4 + 0) * stride] = u0;
            up[base + (r * 4 + 1) * stride] = u1;
            up[base + (r * 4 + 2) * stride] = u2;
            up[base + (r * 4 + 3) * stride] = u3;
        }
    }
}

// `up` (CinP*16*CoutP floats) must be zero-filled by the caller: padded channels stay zero.
int launch_pack_wino_weight(const float* w, float* up, int Cout, int Cin, int CinP, int CoutP, hipStream_t s) {
    const int cot = conv_wino_cout_tile(Cout);
    MCVD_REQUIRE(CinP % 8 == 0 && CoutP % (32 * cot) == 0, "pack_wino_weight: CinP=%d CoutP=%d cot=%d", CinP, CoutP, cot);
    const long n = (long)Cout * Cin;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_wino_weight_kernel, dim3(blocks), dim3(256), 0, s, w, up, Cout, Cin, CinP, CoutP, cot);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mcvd
