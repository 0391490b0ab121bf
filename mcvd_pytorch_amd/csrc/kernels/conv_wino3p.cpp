// PERSISTENT form of the three-piece bf16 Winograd convolution (conv_wino3.cpp: same decomposition, same arithmetic, same packed
// weights, bit-identical results): one workgroup per CU walks a contiguous range of (region, cout tile[, K half]) items and the
// staging pipeline of its K loop never drains between them.
//
// Why (profiles/r03_timeline_wino3_conv1x1_b3.txt, r03_wino3_prologue.txt): a workgroup of conv_wino3_kernel spends 9-10 k cycles in
// its prologue (index arithmetic, first memory latency, activation of two patches by all eight waves at once, first transform) and
// 6-7 k in its epilogue before it retires, and the next workgroup of the CU starts ~1.1 us (1.9 k cycles) later: 17-19 k cycles per
// item against 6-12 chunks of 4.7-4.9 k -- 25-35 % of the CU's time without a single MFMA, and with 256-VGPR waves no second
// workgroup can run beside it.  Here
//   * the items of a workgroup are consecutive: [w n / G, (w + 1) n / G) of n = B * regions * cout tiles * K halves, ordered sample,
//     region, cout tile, K half -- the cout tiles of a region follow each other on the same CU (the patch stays in that XCD's L2)
//     and all workgroups walk the cout tiles in step (the weight slice of the moment is shared by the whole chip);
//   * inside a RUN of items of one sample the K loop is one stream of chunks: while the MFMAs of the last chunks of item j run, the
//     staging phases already fetch, activate and transform the first chunks of item j + 1 (its patch offsets come from a per-thread
//     LDS table written during item j, its weights are requested behind the last MFMAs of item j), so an item costs its chunks plus
//     the epilogue and nothing else; the epilogue (all eight waves: accumulators -> LDS -> inverse transform -> global) uses the V
//     buffer the last chunk has just vacated, 16 output channels per round;
//   * a new sample (new GroupNorm coefficient table) ends the run: the pipeline drains and the prologue runs again.  At the batch
//     sizes of BASELINE.json the item ranges are aligned with the samples and every workgroup has exactly one run.
// Shape ids 16 / 17 (17: 2-way K split, the halves are items).  8x8 images stay with conv_wino3_kernel<.., G8> (one item per CU:
// nothing to stream).  VMEM of the K loop is hand-counted exactly as in conv_wino3.cpp; tools/check_wino_isa.py checks this file too.
#include <stdlib.h>

#include "../common.h"

namespace mcvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_w3p(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

constexpr int WP_CK = 16;        // input channels per chunk = K of one bf16 MFMA
constexpr int WP_T = 32;         // tiles per item (4 x 8 tiles = 8 x 16 output pixels)
constexpr int WP_NT = 512;
#ifndef MCVD_W3_PP
#define MCVD_W3_PP 24
#endif
// LDS patch row pitch in channel-pair columns (18 used; 20 for the two 8x8 images side by side).  The patch park stores a lane's four pixels as
// single dwords at ((pair * 10 + row) * PP + col) * 2 + ce; the 32 lanes of a store group are 8 rows x 4 four-pixel items, bank = (row * 2 PP +
// 8 item) mod 32: with 2 PP = 48 = 16 (mod 32) they fall on FOUR banks -- the "x4 patch park" conflict behind 27-29 percent of the LDS-active
// cycles (profiles/r04_pmc_sq_wave_states.txt).  Round 5 built the conflict-free pitch (25: 2 PP = 18 mod 32, 16 banks, 2-way = free for
// ds_write_b32) and measured it against 24 on one box (tools/build_variant.sh, -DMCVD_W3_PP=25; profiles/r05_patch_pitch_ab.txt): 5577 vs 5592
// cycles per chunk, 248.1 / 248.5 vs 248.8 / 249.4 frames/s -- nothing.  The stores are not on the chunk's critical path; 24 stays.
constexpr int WP_PP = MCVD_W3_PP;
constexpr int WP_PW = 16 * 2 * 4 * WP_T;          // 32-bit words of one piece plane of a V chunk: [position][half][pair][tile]
constexpr int WP_VW = 3 * WP_PW;                  // 32-bit words of one V chunk: [piece][position][half][pair][tile]
constexpr int WP_NPL = 3;                         // patch-load instructions per thread and chunk (two four-pixel slots + one halo slot)
constexpr int WP_MINCH = 4;                       // chunks per item the stream needs (its staging looks three chunks ahead)

__device__ __forceinline__ unsigned wp_cvt_pk(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// exact three-way split of an adjacent register pair into packed bf16 pairs (conv_wino3.cpp: w3_split3)
__device__ __forceinline__ void wp_split3(f32x2 v, unsigned& w1, unsigned& w2, unsigned& w3) {
    w1 = wp_cvt_pk(v.x, v.y);
    const f32x2 h = {__builtin_bit_cast(float, w1 << 16), __builtin_bit_cast(float, w1 & 0xffff0000u)};
    v = v - h;
    w2 = wp_cvt_pk(v.x, v.y);
    const f32x2 g = {__builtin_bit_cast(float, w2 << 16), __builtin_bit_cast(float, w2 & 0xffff0000u)};
    v = v - g;
    w3 = wp_cvt_pk(v.x, v.y);
}

// PRO: 0 raw input, 1 affine, 2 affine + SiLU
template <int COT, int PRO>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(86))) void conv_wino3p_kernel(ConvArgs a) {
    // amdgpu_num_vgpr(86): the compiler may allocate v0-v171 (LLVM doubles the number on gfx90a+; conv_wino3.cpp); v172-v255 hold the
    // in-flight loads and the MFMA A operands and are named in the asm text only.
    constexpr int NT = WP_NT, CK = WP_CK, T = WP_T, BCO = 32 * COT, PP = WP_PP, VW = WP_VW, PW = WP_PW;
    constexpr int PSZ = CK * 10 * PP;           // activated input patch of one chunk: [pair 8][10 rows][PP][2 channels]
    constexpr int PBUF = 4096;                  // + dump space for unused patch slots, rounded up to 16 KiB: the epilogue parks four position planes there
    static_assert(PBUF >= PSZ + 8 && PBUF >= 4 * 32 * WP_T, "patch buffer: patch + dump space, and 4 positions x 32 couts x 32 tiles of the epilogue");
    constexpr int NPL = WP_NPL, NPV = 9;
    constexpr int NQ = 3 * COT;                 // weight quads per position: COT cout sub-tiles x 3 pieces
    constexpr int NA = 2 * NQ;                  // weight loads per wave and chunk
    constexpr int VM_A = NQ + NPL;
    constexpr int PQ = 5;                       // weight quads whose registers hold the first two patches in the prologue
    constexpr long WSTR = 16L * NQ * 256;       // dwords of one chunk of a cout tile's packed weights
    static_assert(NA <= 18 && NA > PQ, "named-register map below: v172-v180 patch, v184-v255 eighteen weight quads");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sV = reinterpret_cast<unsigned*>(smem);           // [2][VW]
    float* sP = smem + 2 * VW;                  // [2][PBUF]
    float* sCo = sP + 2 * PBUF;                 // [Cin][2] prologue coefficients (A_c, B_c) of the run's sample (PRO only)
    unsigned* sOff = reinterpret_cast<unsigned*>(sCo + 2 * a.Cin);      // [2 item parities][NPL][NT]: byte offset of the slot's (clamped) first pixel from
                                                                        // the chunk's first channel plane | bit 0: the slot is zero padding in this region / unused
                                                                        // (written and read by the owning thread only)
    {   // the kernel descriptor must allocate all 256 registers: the asm statements below name v172-v255 in their text only
        float top;
        asm volatile("" : "={v255}"(top));
    }
    asm volatile("" :: "s"(a.x0), "s"(a.x1), "s"(a.coef), "s"(a.wpb), "s"(a.B), "s"(a.H), "s"(a.W), "s"(a.Cin), "s"(a.CinP), "s"(a.C0),
                 "s"(a.C1), "s"(a.CoutP), "s"(a.ksplit), "s"(a.dbg), "s"(a.wdma));
    const unsigned long long t_start = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int H = a.H, W = a.W, HW = H * W, Cin = a.Cin;
    const int rx_n = W >> 4, rpi = rx_n * (H >> 3);              // regions per image
    const int nct = a.CoutP / BCO;
    const int ksp = a.ksplit >= 2 ? a.ksplit : 1;                // K parts per (region, cout tile): 1, 2, 4 or 8
    const int nch = (a.CinP / CK) / ksp;                         // chunks per item
    const int ips = rpi * nct * ksp;                             // items per sample
    const int n_items = a.B * ips;
    const int it0 = (int)((long)blockIdx.x * n_items / gridDim.x), it1 = (int)((long)(blockIdx.x + 1) * n_items / gridDim.x);
    if (it0 >= it1) return;
    const int rg = __builtin_amdgcn_readfirstlane(wave >> 2);   // rows 2rg, 2rg+1 of B^T d; phase order of the wave
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#ifdef MCVD_DIAG
    // diagnostics build only (env MCVD_W3P_STAGGER = cycles / 256): the odd workgroups start late.  All workgroups of this kernel run in
    // lock step (same work, same start); the experiment measures what the simultaneous epilogues / weight requests of 256 CUs cost.
    if ((a.wdma >> 8) != 0 && (blockIdx.x & 1)) {
        const unsigned long long until = __builtin_amdgcn_s_memtime() + ((unsigned long long)(a.wdma >> 8) << 8);
        while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(8);
    }
#endif

    // ---- transform role (conv_wino3.cpp): (channel pair, tile) = tid & 255
    const int s_tile = tid & 31, s_cp = (tid & 255) >> 5;
    const int s_ty = s_tile >> 3, s_tx = s_tile & 7;
    const int p_rd = ((s_cp * 10 + 2 * s_ty + rg) * PP + 2 * s_tx) * 2;
    const int v_wr = ((8 * rg * 2 + (s_cp & 1)) * 4 + (s_cp >> 1)) * T + s_tile;

    // ---- patch-load slots, the item-invariant part: p_pk = LDS float index of the slot's first element | channel in chunk << 12
    // (CK for an unused slot: parked in the dump space, never valid)
    unsigned p_pk[NPL];
#pragma unroll
    for (int sl = 0; sl < NPL; ++sl) {
        p_pk[sl] = (unsigned)PSZ | ((unsigned)CK << 12);
        if (sl < 2) {
            const int e = sl * NT + tid;
            if (e < CK * 40) {
                const int ci = e / 40, rem = e - ci * 40, r = rem >> 2, c = (rem & 3) * 4;
                const int cp = (ci >> 2) * 2 + (ci & 1), ce = (ci >> 1) & 1;
                p_pk[sl] = (unsigned)(((cp * 10 + r) * PP + c + 1) * 2 + ce) | ((unsigned)ci << 12);
            }
        } else if (tid < CK * 20) {
            const int e = tid, ci = e / 20, rem = e - ci * 20, r = rem >> 1, c = (rem & 1) * 17;
            const int cp = (ci >> 2) * 2 + (ci & 1), ce = (ci >> 1) & 1;
            p_pk[sl] = (unsigned)(((cp * 10 + r) * PP + c) * 2 + ce) | ((unsigned)ci << 12);
        }
    }
    // the item-dependent part: byte offsets of the slots for a region at (oy0, ox0) -> sOff[tab], bit 0 = "zero padding here"
#define WP_SLOT_TABLE(tab, oy0, ox0, OUT)                                                                       \
    {                                                                                                           \
        int tid_s;       /* an opaque copy of tid: the slot geometry is recomputed per item, not kept live across the K loops */ \
        asm volatile("v_mov_b32 %0, %1" : "=v"(tid_s) : "v"(tid));                                              \
        _Pragma("unroll") for (int sl = 0; sl < NPL; ++sl) {                                                    \
            unsigned off = 1u;                                                                                  \
            if (sl < 2) {                                                                                       \
                const int e = sl * NT + tid_s;                                                                  \
                if (e < CK * 40) {                                                                              \
                    const int ci = e / 40, rem = e - ci * 40, r = rem >> 2, c = (rem & 3) * 4;                  \
                    const int y = (oy0) - 1 + r;                                                                \
                    const bool inside = y >= 0 && y < H;                                                        \
                    off = (unsigned)(ci * HW + min(max(y, 0), H - 1) * W + (ox0) + c) * 4u + (inside ? 0u : 1u); \
                }                                                                                               \
            } else if (tid_s < CK * 20) {                                                                       \
                const int e = tid_s, ci = e / 20, rem = e - ci * 20, r = rem >> 1, c = (rem & 1) * 17;          \
                const int y = (oy0) - 1 + r, x = (ox0) - 1 + c;                                                 \
                const bool inside = y >= 0 && y < H && x >= 0 && x < W;                                         \
                off = (unsigned)(ci * HW + min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)) * 4u + (inside ? 0u : 1u); \
            }                                                                                                   \
            sOff[((tab) * NPL + sl) * NT + tid] = off;                                                          \
            OUT[sl] = off;                                                                                      \
        }                                                                                                       \
    }

    const unsigned wr_voff = (unsigned)lane * 16u;

    /* IN-FLIGHT DATA LIVES IN REGISTERS THE COMPILER DOES NOT ALLOCATE (conv_wino3.cpp has the full account): weight quad q in
       v[184 + 4q : 187 + 4q], patch slots in v[172:175], v[176:179], v180; bare s_waitcnt; the MFMAs name their A operand in the text. */
#define WP_QUADS(X, q, A1, A2) X(0, "v[184:187]", q, A1, A2) X(1, "v[188:191]", q, A1, A2) X(2, "v[192:195]", q, A1, A2) X(3, "v[196:199]", q, A1, A2) X(4, "v[200:203]", q, A1, A2) X(5, "v[204:207]", q, A1, A2) X(6, "v[208:211]", q, A1, A2) X(7, "v[212:215]", q, A1, A2) X(8, "v[216:219]", q, A1, A2) X(9, "v[220:223]", q, A1, A2) X(10, "v[224:227]", q, A1, A2) X(11, "v[228:231]", q, A1, A2) X(12, "v[232:235]", q, A1, A2) X(13, "v[236:239]", q, A1, A2) X(14, "v[240:243]", q, A1, A2) X(15, "v[244:247]", q, A1, A2) X(16, "v[248:251]", q, A1, A2) X(17, "v[252:255]", q, A1, A2)
#define WP_LD1(K, R, q, P, UNUSED) if ((q) == K) asm volatile("global_load_dwordx4 " R ", %0, %1" :: "v"(wr_voff), "s"(P) : "memory");
#define WP_MF1(K, R, q, ACC, BV) if ((q) == K) asm volatile("v_mfma_f32_32x32x16_bf16 %0, " R ", %1, %0" : "+v"(ACC) : "v"(BV));
    /* weight piece `pc` of every cout sub-tile of position 2w + i of the chunk whose (wave's) weights start at WPTR */
#define WP_LOAD_A_PIECE(WPTR, i, pc)                                                                            \
    {                                                                                                           \
        const unsigned* ua = (WPTR) + (i) * (NQ * 256);                                                         \
        _Pragma("unroll") for (int ct = 0; ct < COT; ++ct) { WP_QUADS(WP_LD1, (i) * NQ + ct * 3 + (pc), ua + (ct * 3 + (pc)) * 256, 0) } \
    }
#define WP_LD1D(K, R, q, P, DEP) if ((q) == K) asm volatile("global_load_dwordx4 " R ", %0, %1" :: "v"(wr_voff), "s"(P), "v"(DEP) : "memory");
#define WP_LOAD_A_RANGE(WPTR, Q0, Q1, DEP)                                                                      \
    {                                                                                                           \
        const unsigned* ua = (WPTR);                                                                            \
        _Pragma("unroll") for (int qq = (Q0); qq < (Q1); ++qq) { WP_QUADS(WP_LD1D, qq, ua + qq * 256, DEP) }    \
    }
#define WP_WAIT(N) asm volatile("s_waitcnt vmcnt(%1)\n\ts_mov_b32 %0, 0" : "=s"(vtok) : "n"(N) : "memory");
    /* unconditional, clamped raw loads of the patch of absolute chunk `ch` of sample bb: OFS = the slots' table words */
#define WP_LOAD_P(bb, ch, DEP, OFS) WP_LOAD_PR(bb, ch, DEP, OFS, "v[172:175]", "v[176:179]", "v180")
#define WP_LOAD_PR(bb, ch, DEP, OFS, RA, RB, RH)                                                                \
    {                                                                                                           \
        const int cb = min((ch) * CK, Cin - 1);                                                                 \
        const unsigned lim4 = (unsigned)((Cin - cb) * HW - 4) * 4u, lim1 = (unsigned)((Cin - cb) * HW - 1) * 4u; \
        const bool second = cb >= a.C0;                                                                         \
        const float* srcb = second ? a.x1 + ((long)(bb) * a.C1 + (cb - a.C0)) * HW : a.x0 + ((long)(bb) * a.C0 + cb) * HW; \
        const unsigned o0 = min(OFS[0] & ~3u, lim4), o1 = min(OFS[1] & ~3u, lim4), o2 = min(OFS[2] & ~3u, lim1); \
        asm volatile("global_load_dwordx4 " RA ", %0, %3\n\tglobal_load_dwordx4 " RB ", %1, %3\n\tglobal_load_dword " RH ", %2, %3" \
                     :: "v"(o0), "v"(o1), "v"(o2), "s"(srcb), "v"(DEP) : "memory");                               \
    }
#define WP_READ_C(ch, cfv)                                                                                      \
    {                                                                                                           \
        _Pragma("unroll") for (int sl = 0; sl < NPL; ++sl) {                                                    \
            cfv[sl] = f32x2{1.0f, 0.0f};                                                                        \
            if (PRO >= 1) {                                                                                     \
                const int cch = min((ch) * CK + (int)((p_pk[sl] >> 12) & (CK - 1)), Cin - 1);                   \
                cfv[sl] = *reinterpret_cast<const f32x2*>(sCo + cch * 2);                                       \
            }                                                                                                   \
        }                                                                                                       \
    }
#define WP_NOHOOK(e, v)
#define WP_WRITE_P(par, ch, FL, PV, cfv) WP_WRITE_PR(par, ch, FL, PV, cfv, "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", WP_NOHOOK)
    /* activate once per pixel and park the patch of absolute chunk `ch` in patch buffer `par`; FL bit sl = slot sl is zero padding;
       zero padding applies AFTER the activation.  HOOK(e, v): statements placed behind value e */
#define WP_WRITE_PR(par, ch, FL, PV, cfv, A0, A1, A2, A3, B0, B1, B2, B3, H0, HOOK)                             \
    {                                                                                                           \
        float* sPw = sP + ((par) ? PBUF : 0);                                                                   \
        const int nvalid = min(Cin - (ch) * CK, CK);                                                            \
        asm("v_fma_f32 %0, " A0 ", %9, %10\n\tv_fma_f32 %1, " A1 ", %9, %10\n\tv_fma_f32 %2, " A2 ", %9, %10\n\tv_fma_f32 %3, " A3 ", %9, %10\n\t" \
            "v_fma_f32 %4, " B0 ", %11, %12\n\tv_fma_f32 %5, " B1 ", %11, %12\n\tv_fma_f32 %6, " B2 ", %11, %12\n\tv_fma_f32 %7, " B3 ", %11, %12\n\t" \
            "v_fma_f32 %8, " H0 ", %13, %14"                                                                      \
            : "=&v"(PV[0]), "=&v"(PV[1]), "=&v"(PV[2]), "=&v"(PV[3]), "=&v"(PV[4]), "=&v"(PV[5]), "=&v"(PV[6]), "=&v"(PV[7]), "=&v"(PV[8]) \
            : "v"(cfv[0].x), "v"(cfv[0].y), "v"(cfv[1].x), "v"(cfv[1].y), "v"(cfv[2].x), "v"(cfv[2].y), "s"(vtok)); \
        _Pragma("unroll") for (int e = 0; e < NPV; ++e) {                                                       \
            const int sl = e >> 2;                            /* values 0-3: slot 0, 4-7: slot 1, 8: slot 2 */   \
            float v = PV[e];                                                                                    \
            if (PRO >= 2) v = silu_w3p(v);                                                                      \
            const bool keep = (((FL) >> sl) & 1u) == 0u && (int)((p_pk[sl] >> 12) & 0xff) < nvalid;             \
            sPw[(p_pk[sl] & 0xfff) + 2 * (e & 3)] = keep ? v : 0.0f;                                            \
            HOOK(e, v)                                                                                          \
        }                                                                                                       \
    }
#define WP_READ_R(par, RW)                                                                                      \
    {                                                                                                           \
        const f32x2* sPr = reinterpret_cast<const f32x2*>(sP + ((par) ? PBUF : 0) + p_rd);                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { RW[0][j] = sPr[j]; RW[1][j] = sPr[PP + j]; RW[2][j] = sPr[2 * PP + j]; } \
    }
#define WP_WRITE_V(par, RG, RW)                                                                                 \
    {                                                                                                           \
        unsigned* vdst = sV + ((par) ? VW : 0) + v_wr;                                                          \
        f32x2 mx[4], my[4];                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const f32x2 r0 = RW[0][j], r1 = RW[1][j], r2 = RW[2][j];                                            \
            if ((RG) == 0) { mx[j] = r0 - r2; my[j] = r1 + r2; }                                                \
            else { mx[j] = r1 - r0; my[j] = r0 - r2; }                                                          \
        }                                                                                                       \
        _Pragma("unroll") for (int row = 0; row < 2; ++row) {                                                   \
            const f32x2 m0 = row ? my[0] : mx[0], m1 = row ? my[1] : mx[1], m2 = row ? my[2] : mx[2], m3 = row ? my[3] : mx[3]; \
            const f32x2 v0 = m0 - m2, v1 = m1 + m2, v2 = m2 - m1, v3 = m1 - m3;                                 \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                     \
                unsigned w1, w2, w3;                                                                            \
                wp_split3(q == 0 ? v0 : q == 1 ? v1 : q == 2 ? v2 : v3, w1, w2, w3);                            \
                vdst[(row * 4 + q) * 256] = w1;                                                                 \
                vdst[(row * 4 + q) * 256 + PW] = w2;                                                            \
                vdst[(row * 4 + q) * 256 + 2 * PW] = w3;                                                        \
            }                                                                                                   \
        }                                                                                                       \
    }
#define WP_LOAD_B(i, BQ)                                                                                        \
    {                                                                                                           \
        const unsigned* q = sVc + (((2 * wave + (i)) * 2 + half) * 4) * T + l31;                                \
        _Pragma("unroll") for (int jp = 0; jp < 4; ++jp) {                                                      \
            BQ[0][jp] = q[jp * T]; BQ[1][jp] = q[PW + jp * T]; BQ[2][jp] = q[2 * PW + jp * T];                  \
        }                                                                                                       \
    }
#define WP_PRODUCT(i, PA, PB)                                                                                   \
    { _Pragma("unroll") for (int ct = 0; ct < COT; ++ct) { WP_QUADS(WP_MF1, 3 * ((i) * COT + ct) + (PA), acc[i][ct], bq[i][PB]) } }
    /* all MFMAs of the chunk whose V sits in buffer `par` (weights in the named registers); behind each piece's last product the
       same registers are re-requested from WNX, the (wave's) weights of the NEXT chunk of the stream (conv_wino3.cpp: W3_MFMA_PHASE) */
#define WP_MFMA_PHASE(par, WNX)                                                                                 \
    {                                                                                                           \
        const unsigned* sVc = sV + ((par) ? VW : 0);                                                            \
        u32x4 bq[2][3];                                                                                         \
        WP_LOAD_B(0, bq[0]) WP_LOAD_B(1, bq[1])                                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                         \
            WP_WAIT(VM_A)                                                                                       \
            WP_PRODUCT(i, 2, 0)                                                                                 \
            WP_LOAD_A_PIECE(WNX, i, 2)                                                                          \
            WP_PRODUCT(i, 1, 1) WP_PRODUCT(i, 0, 2) WP_PRODUCT(i, 1, 0)                                         \
            WP_LOAD_A_PIECE(WNX, i, 1)                                                                          \
            WP_PRODUCT(i, 0, 1) WP_PRODUCT(i, 0, 0)                                                             \
            WP_LOAD_A_PIECE(WNX, i, 0)                                                                          \
        }                                                                                                       \
    }
    /* staging of the stream at chunk g (V(g) is being multiplied): park chunk g+2 (absolute chunk CH2, patch buffer par: g+2 and g
       have the same parity) -- its raw values were requested one chunk ago together with the flags pfl --, request chunk g+3
       (absolute chunk CH3, slot table TAB3), transform chunk g+1 (buffers par ^ 1) */
#define WP_VALU_PHASE(par, CH2, CH3, TAB3, RG)                                                                  \
    {                                                                                                           \
        {                                                                                                       \
            f32x2 cfv[NPL];                                                                                     \
            unsigned ofs[NPL];                                                                                  \
            WP_READ_C(CH2, cfv)                                                                                 \
            _Pragma("unroll") for (int sl = 0; sl < NPL; ++sl) ofs[sl] = sOff[((TAB3) * NPL + sl) * NT + tid];  \
            WP_WAIT(NA)                                                                                         \
            float pv[NPV];                                                                                      \
            _Pragma("unroll") for (int e = 0; e < NPV; ++e) pv[e] = 0.0f;                                       \
            WP_WRITE_P(par, CH2, pfl, pv, cfv)                                                                  \
            pfl = (ofs[0] & 1u) | ((ofs[1] & 1u) << 1) | ((ofs[2] & 1u) << 2);                                  \
            WP_LOAD_P(b, CH3, pv[0], ofs)                                                                       \
        }                                                                                                       \
        {                                                                                                       \
            f32x2 rw[3][4];                                                                                     \
            WP_READ_R((par) ^ 1, rw)                                                                            \
            WP_WRITE_V((par) ^ 1, RG, rw)                                                                       \
        }                                                                                                       \
    }

    f32x16 acc[2][COT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][ct][r] = 0.0f;

    // diagnostics (mcvd_ctx_set_debug_buffer): shader-clock time wave (a.wdma & 7) spends in {prologues, K loops, epilogues}
    const bool rec = a.dbg != nullptr && wave == (a.wdma & 7);
    unsigned long long tprev = t_start, dt[3] = {0, 0, 0}, rt0 = 0;
    if (rec) rt0 = __builtin_amdgcn_s_memrealtime();
#define WP_STAMP(i)                                                                                             \
    if (rec) {                                                                                                  \
        const unsigned long long now = __builtin_amdgcn_s_memtime();                                            \
        dt[i] += now - tprev;                                                                                   \
        tprev = now;                                                                                            \
    }

    const unsigned* const wpb_u = reinterpret_cast<const unsigned*>(a.wpb);
    const bool fin = ksp == 1;                 // K split: bias, residual and scale are applied by the reduce kernel
    int vtok = 0;                              // ordering token: written by every VMEM wait, an operand of the register reads
    unsigned pfl = 0;                          // zero-padding flags of the patch whose raw values are in flight (bit sl)
    int chunks_done = 0;

    int it = it0;
    while (it < it1) {
        // ================= a run: the items [it, run_end) of sample b =================
        const int b = it / ips;
        const int run_end = min(it1, (b + 1) * ips);
        // -------- prologue (conv_wino3.cpp).  Issue order = need order: the coefficients of the sample, the raw patches of the run's first
        // two chunks (into the registers of weight quads 0 .. PQ-1), the third patch, then the weight quads of the first chunk.
        {
            const int q0 = it - b * ips, kh = q0 % ksp, q1 = q0 / ksp, cotile = q1 % nct, rr = q1 / nct;
            const int oy0 = (rr / rx_n) * 8, ox0 = (rr % rx_n) * 16, cbeg = kh * nch;
            const unsigned* w0 = wpb_u + (((long)cotile * (nch * ksp) + cbeg) * 16 + 2 * wave_u) * (NQ * 256);
            const float* co_src[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) co_src[k] = a.coef + ((long)b * Cin + min(tid + k * NT, Cin - 1)) * 2;
            if (PRO)
                asm volatile("global_load_dwordx2 v[172:173], %0, off\n\tglobal_load_dwordx2 v[174:175], %1, off"
                             :: "v"(co_src[0]), "v"(co_src[1]) : "memory");
            const float nodep = 0.0f;
            unsigned ofs[NPL];
            WP_SLOT_TABLE(it & 1, oy0, ox0, ofs)
            const unsigned fl0 = (ofs[0] & 1u) | ((ofs[1] & 1u) << 1) | ((ofs[2] & 1u) << 2);
            WP_LOAD_PR(b, cbeg, nodep, ofs, "v[184:187]", "v[188:191]", "v192")
            WP_LOAD_PR(b, cbeg + 1, nodep, ofs, "v[194:197]", "v[198:201]", "v202")
            WP_WAIT(0)                         // the coefficients and the two patches have landed
            float cdep = 0.0f;
            if (PRO) {
                f32x2 cpre[2];
                asm volatile("v_mov_b32 %0, v172\n\tv_mov_b32 %1, v173\n\tv_mov_b32 %2, v174\n\tv_mov_b32 %3, v175"
                             : "=v"(cpre[0].x), "=v"(cpre[0].y), "=v"(cpre[1].x), "=v"(cpre[1].y) : "s"(vtok));
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (tid + k * NT < Cin) *reinterpret_cast<f32x2*>(sCo + (tid + k * NT) * 2) = cpre[k];
                cdep = cpre[0].x + cpre[1].x;
            }
            WP_LOAD_P(b, cbeg + 2, cdep, ofs)  // (behind the reads of v172-v175)
            pfl = fl0;
            if (PRO) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // coefficient table visible
            {
                // the weight quads PQ.. of the first chunk are requested one behind each activated value (conv_wino3.cpp)
                constexpr int QR = (NA - PQ + 2 * NPV - 1) / (2 * NPV);
#define WP_HOOK0(e, v) WP_LOAD_A_RANGE(w0, PQ + (e) * QR, (PQ + ((e) + 1) * QR < NA ? PQ + ((e) + 1) * QR : NA), v)
#define WP_HOOK1(e, v) WP_LOAD_A_RANGE(w0, (PQ + (NPV + (e)) * QR < NA ? PQ + (NPV + (e)) * QR : NA), (PQ + (NPV + (e) + 1) * QR < NA ? PQ + (NPV + (e) + 1) * QR : NA), v)
                float pv0[NPV], pv1[NPV];
                f32x2 cf0[NPL], cf1[NPL];
                WP_READ_C(cbeg, cf0)
                WP_READ_C(cbeg + 1, cf1)
                WP_WRITE_PR(0, cbeg, fl0, pv0, cf0, "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", WP_HOOK0)
                WP_WRITE_PR(1, cbeg + 1, fl0, pv1, cf1, "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", WP_HOOK1)
#undef WP_HOOK0
#undef WP_HOOK1
                const float dep = pv0[0] + pv1[0];
                WP_LOAD_A_RANGE(w0, 0, PQ, dep)
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");               // the first two patches visible
            {
                f32x2 rw[3][4];
                WP_READ_R(0, rw)
                WP_WRITE_V(0, rg, rw)
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");               // V of the first chunk visible
            WP_WAIT(0)                         // the loop's in-order counts start from an empty queue
        }
        WP_STAMP(0)
        int par = 0;                           // parity of the stream's current chunk: V(g) in sV[par], patch(g + 2) goes to sP[par]
        for (; it < run_end; ++it) {
            // ---- this item and the next one of the run (the stream stages up to three chunks of it; the last item of a run stages its
            // own first chunks once more: valid addresses, results never used)
            const int q0 = it - b * ips, kh = q0 % ksp, q1 = q0 / ksp, cotile = q1 % nct, rr = q1 / nct;
            const int nx = it + 1 < run_end ? it + 1 : it;
            const int n0 = nx - b * ips, nkh = n0 % ksp, n1 = n0 / ksp, ncotile = n1 % nct, nrr = n1 / nct;
            const int cbeg = kh * nch, ncbeg = nkh * nch;
            const unsigned* wb_cur = wpb_u + (((long)cotile * (nch * ksp) + cbeg) * 16 + 2 * wave_u) * (NQ * 256);
            const unsigned* wb_nx = wpb_u + (((long)ncotile * (nch * ksp) + ncbeg) * 16 + 2 * wave_u) * (NQ * 256);
            const int tab_cur = it & 1, tab_nx = nx & 1;
            {   // slot table of the next item (first read three chunks before this item ends; nch >= 4)
                unsigned dummy[NPL];
                WP_SLOT_TABLE(tab_nx, (nrr / rx_n) * 8, (nrr % rx_n) * 16, dummy)
                (void)dummy;
            }
            // VMEM issue order of a wave per chunk (in-order vmcnt counter; nothing else of the loop is outstanding):
            //   waves 0-3:  [patch(g+3): NPL loads] [weights(g+1): NQ loads behind the MFMAs of each position]      waves 4-7:  weights, then patch
            //   patch(g+2) before its write: vmcnt(NA); weights(g) of a position before its MFMAs: vmcnt(VM_A)       (conv_wino3.cpp)
#define WP_CHUNK_SCALARS                                                                                        \
            const int l2 = c + 2, l3 = c + 3;                                                                   \
            const int ch2 = l2 < nch ? cbeg + l2 : ncbeg + l2 - nch;                                            \
            const int ch3 = l3 < nch ? cbeg + l3 : ncbeg + l3 - nch;                                            \
            const int tab3 = l3 < nch ? tab_cur : tab_nx;                                                       \
            const unsigned* wnx = c + 1 < nch ? wb_cur + (long)(c + 1) * WSTR : wb_nx;
            if (rg == 0) {
                for (int c = 0; c < nch; ++c) {
                    WP_CHUNK_SCALARS
                    WP_VALU_PHASE(par, ch2, ch3, tab3, rg)
                    WP_MFMA_PHASE(par, wnx)
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    par ^= 1;
                }
                // the loads still in flight target registers the compiler does not know: one wait behind each loop (two different asm
                // texts: identical statements would be sunk into the join block, behind code the checker cannot see through)
                asm volatile("s_waitcnt vmcnt(0) ; staging-first loop left" ::: "memory");
            } else {
                for (int c = 0; c < nch; ++c) {
                    WP_CHUNK_SCALARS
                    WP_MFMA_PHASE(par, wnx)
                    WP_VALU_PHASE(par, ch2, ch3, tab3, rg)
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    par ^= 1;
                }
                asm volatile("s_waitcnt vmcnt(0) ; matrix-first loop left" ::: "memory");
            }
#undef WP_CHUNK_SCALARS
            chunks_done += nch;
            // The MFMAs are inline asm: the compiler inserts none of the wait states a read of an MFMA result needs (8-pass MFMA -> VALU /
            // LDS read: 11).  The nops are tied to the accumulators.
            if constexpr (COT == 3)
                asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]) :: "memory");
            else if constexpr (COT == 2)
                asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]) :: "memory");
            else
                asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0][0]), "+v"(acc[1][0]) :: "memory");
            WP_STAMP(1)

            // ---------------- inverse transform + epilogue of the item, one 32-cout sub-tile per round ----------------
            // LDS: 64 KiB of position planes [16][32 couts][32 tiles] in what the stream has just vacated -- positions 0-11 in the V buffer of
            // the item's last chunk (par was flipped behind it: sV[par ^ 1], 48 KiB), positions 12-15 in the patch buffer whose chunk was
            // transformed during that last iteration (sP[par], 16 KiB).  The other V buffer and the other patch buffer hold the next item's
            // first chunks.  (A first form had 32 KiB only and ran six rounds of 16 couts: twice the barriers, one task per thread and round.)
            float* sM0 = smem + ((par ^ 1) ? VW : 0);                 // positions 0 .. 11
            float* sM1 = sP + (par ? PBUF : 0);                       // positions 12 .. 15
            // the epilogue's thread indices are recomputed per item from an opaque copy of tid: hoisted out of the item loop they would be
            // live across the K loops, whose register budget is the accumulators' (the allocator spilled them to scratch)
            int tid_e;
            asm volatile("v_mov_b32 %0, %1" : "=v"(tid_e) : "v"(tid));
            const int e_tile = tid_e & 31, e_col0 = tid_e >> 5;      // two (cout, tile) tasks per thread and round: couts e_col0 and e_col0 + 16
            const int e_ty = e_tile >> 3, e_tx = e_tile & 7;
            const int oy0 = (rr / rx_n) * 8, ox0 = (rr % rx_n) * 16, co0 = cotile * BCO;
            const long pix = (long)(oy0 + 2 * e_ty) * W + ox0 + 2 * e_tx;
            float* const ydst = fin ? a.y : a.part + (long)kh * a.B * a.Cout * HW;
            // bias and residual of a round are requested one round ahead (fetched where they are used, every round waited for two
            // dependent memory latencies between its barriers; all 2 * COT tasks up front do not fit beside the accumulators here)
            float e_bias[COT][2];
            f32x2 e_r0[COT][2], e_r1[COT][2];
#define WP_EPI_FETCH(ct)                                                                                        \
            {                                                                                                   \
                _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) {                                              \
                    const int co = co0 + (ct) * 32 + e_col0 + 16 * t2;                                          \
                    e_bias[ct][t2] = fin ? a.bias[co] : 0.0f;           /* zero-padded to CoutP */               \
                    e_r0[ct][t2] = e_r1[ct][t2] = f32x2{0.0f, 0.0f};                                            \
                    if (a.res && fin) {                                                                         \
                        const long o = ((long)b * a.Cout + min(co, a.Cout - 1)) * HW + pix;                     \
                        e_r0[ct][t2] = *reinterpret_cast<const f32x2*>(a.res + o);                              \
                        e_r1[ct][t2] = *reinterpret_cast<const f32x2*>(a.res + o + W);                          \
                    }                                                                                           \
                }                                                                                               \
            }
            WP_EPI_FETCH(0)
            float* const sMw = wave_u < 6 ? sM0 + (2 * wave_u) * (32 * T) : sM1 + (2 * wave_u - 12) * (32 * T);      // this wave's two planes
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int col = (r & 3) + 8 * (r >> 2) + 4 * half;
                        sMw[(i * 32 + col) * T + l31] = acc[i][ct][r];
                    }
                if (ct + 1 < COT) WP_EPI_FETCH(ct + 1 < COT ? ct + 1 : ct)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const int e_col = e_col0 + 16 * t2;
                    const int co = co0 + ct * 32 + e_col;
                    const f32x2 r0 = e_r0[ct][t2], r1 = e_r1[ct][t2];
                    float mm[16];
#pragma unroll
                    for (int xi = 0; xi < 16; ++xi) mm[xi] = xi < 12 ? sM0[(xi * 32 + e_col) * T + e_tile] : sM1[((xi - 12) * 32 + e_col) * T + e_tile];
                    float t0[4], t1[4];                                 // A^T M
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        t0[l] = mm[0 * 4 + l] + mm[1 * 4 + l] + mm[2 * 4 + l];
                        t1[l] = mm[1 * 4 + l] - mm[2 * 4 + l] - mm[3 * 4 + l];
                    }
                    const float y00 = t0[0] + t0[1] + t0[2], y01 = t0[1] - t0[2] - t0[3];
                    const float y10 = t1[0] + t1[1] + t1[2], y11 = t1[1] - t1[2] - t1[3];
                    const float bvv = e_bias[ct][t2];
                    const float osc = fin ? a.out_scale : 1.0f;
                    const float v00 = (y00 + bvv + r0.x) * osc, v01 = (y01 + bvv + r0.y) * osc;
                    const float v10 = (y10 + bvv + r1.x) * osc, v11 = (y11 + bvv + r1.y) * osc;
                    if (co < a.Cout) {
                        const long o = ((long)b * a.Cout + co) * HW + pix;
                        *reinterpret_cast<float2*>(ydst + o) = make_float2(v00, v01);
                        *reinterpret_cast<float2*>(ydst + o + W) = make_float2(v10, v11);
                    }
                    if (a.stats && fin) {
                        // GroupNorm partials of the FINAL values, pilot-shifted moments over the 32 tiles of a half-wave (conv_wino3.cpp)
                        float pil;
                        {
                            const int pv = __builtin_bit_cast(int, v00);
                            const int s0 = __builtin_amdgcn_readlane(pv, 0), s2 = __builtin_amdgcn_readlane(pv, 32);
                            pil = __builtin_bit_cast(float, (lane & 32) ? s2 : s0);
                        }
                        const float d0 = v00 - pil, d1 = v01 - pil, d2 = v10 - pil, d3 = v11 - pil;
                        float sm = (d0 + d1) + (d2 + d3);
                        float qm = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#define WP_MERGE(CTRL, ROWMASK)                                                                                     \
                        {                                                                                           \
                            sm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sm), CTRL, ROWMASK, 0xf, false)); \
                            qm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, qm), CTRL, ROWMASK, 0xf, false)); \
                        }
                        WP_MERGE(0xB1, 0xf)                   // quad_perm [1,0,3,2]
                        WP_MERGE(0x4E, 0xf)                   // quad_perm [2,3,0,1]
                        WP_MERGE(0x124, 0xf)                  // row_ror:4
                        WP_MERGE(0x128, 0xf)                  // row_ror:8: every lane of a row of 16 holds the row's totals
                        WP_MERGE(0x142, 0xa)                  // row_bcast:15: lanes 16-31 / 48-63 add the totals of the row below
#undef WP_MERGE
                        if (e_tile == 31 && co < a.Cout) {
                            float* q = a.stats + (((long)b * a.Cout + co) * rpi + rr) * 2;
                            q[0] = sm + 128.0f * pil;           // the partial's sum over its pixels
                            q[1] = fmaxf(qm - sm * sm * (1.0f / 128.0f), 0.0f);      // M2 about the partial's own mean
                        }
                    }
                }
                // the round's LDS reads done before the next round's writes -- and, behind the last round, before the stream's next
                // park / transform write into these buffers
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            // the accumulators of the next item start from zero (VALU writes; the first MFMA that reads them is hundreds of cycles away)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ct = 0; ct < COT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][ct][r] = 0.0f;
            WP_STAMP(2)
        }
        // end of the run: every asm load has landed (vmcnt(0) behind the last K loop), the epilogue's last barrier has passed
    }
    if (rec) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long* d = a.dbg + (long)blockIdx.x * 8;
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            d[0] = dt[0]; d[1] = dt[1]; d[2] = rt0; d[3] = __builtin_amdgcn_s_memrealtime();
            d[4] = ((unsigned long long)xcc << 32) | hwid;
            d[5] = dt[2];                  // epilogues
            d[6] = (unsigned long long)chunks_done | ((unsigned long long)(it1 - it0) << 32);
            d[7] = now - t_start;
        }
    }
#undef WP_STAMP
#undef WP_EPI_FETCH
#undef WP_SLOT_TABLE
#undef WP_QUADS
#undef WP_LD1
#undef WP_MF1
#undef WP_LOAD_A_PIECE
#undef WP_LD1D
#undef WP_LOAD_A_RANGE
#undef WP_WAIT
#undef WP_LOAD_P
#undef WP_LOAD_PR
#undef WP_READ_C
#undef WP_NOHOOK
#undef WP_WRITE_P
#undef WP_WRITE_PR
#undef WP_READ_R
#undef WP_WRITE_V
#undef WP_LOAD_B
#undef WP_PRODUCT
#undef WP_MFMA_PHASE
#undef WP_VALU_PHASE
}

static size_t wino3p_lds_bytes(int Cin) {
    return (size_t)(2 * WP_VW + 2 * 4096 + 2 * Cin + 2 * WP_NPL * WP_NT) * sizeof(float);
}

// compute units of the current device (one persistent workgroup each), cached per device
static int wino3p_num_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cus[dev].load(std::memory_order_acquire);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev].store(n, std::memory_order_release);
    return n;
}

template <int COT, int PRO>
static int wino3p_launch2(const ConvArgs& a, hipStream_t s) {
    constexpr int BCO = 32 * COT;
    static PerDeviceOnce raised;
    if (raised.first_use()) {
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3p_kernel<COT, PRO>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised.done();
    }
    const int ksp = a.ksplit >= 2 ? a.ksplit : 1;
    const long n_items = (long)a.B * (a.H / 8) * (a.W / 16) * (a.CoutP / BCO) * ksp;
    const int cus = a.pgrid > 0 ? a.pgrid : wino3p_num_cus();
    dim3 grid((unsigned)(n_items < cus ? n_items : cus));
    ConvArgs k = a;
    if (k.dbg) k.wdma = 0;                 // wave 0 records its phase times
#ifdef MCVD_DIAG
    if (k.dbg) {
        const char* w = getenv("MCVD_DBG_WAVE");
        k.wdma = w ? atoi(w) : 0;
    }
    k.wdma &= 0xff;
    if (const char* st = getenv("MCVD_W3P_STAGGER")) k.wdma |= atoi(st) << 8;
#else
    k.wdma &= 0xff;
#endif
    hipLaunchKernelGGL((conv_wino3p_kernel<COT, PRO>), grid, dim3(WP_NT), wino3p_lds_bytes(a.Cin), s, k);
    MCVD_HIP_CHECK(hipGetLastError());
    if (ksp >= 2) return launch_wino_ksplit_reduce(a, s);
    if (a.stats) set_last_conv_stats_np((a.H / 8) * (a.W / 16));
    return 0;
}

template <int COT>
static int wino3p_launch1(const ConvArgs& a, hipStream_t s) {
    if (!a.coef && !a.act) return wino3p_launch2<COT, 0>(a, s);
    if (!a.act) return wino3p_launch2<COT, 1>(a, s);
    return wino3p_launch2<COT, 2>(a, s);
}

// Shape ids 16 / 17 apply to this launch: regions of 8 x 16 output pixels (not the 8x8 images), no SPADE prologue, no consumer-side
// GroupNorm reduction, pre-split packed weights present, at least WP_MINCH chunks per item (17: per K half, and an even chunk count).
bool conv_wino3p_usable(const ConvArgs& a) {
    const int nchunks = a.CinP / WP_CK;
    return a.ks == 3 && a.H % 8 == 0 && a.W % 16 == 0 && a.H >= 8 && a.W >= 16 && a.wpb && !a.gb && a.Cin <= 1024 &&
           a.CinP % WP_CK == 0 && (a.C1 == 0 || a.C0 % WP_CK == 0) && a.H * a.W <= 16384 &&
           (long)a.B * (a.C0 > a.C1 ? a.C0 : a.C1) * a.H * a.W < (1L << 29) && wino3p_lds_bytes(a.Cin) <= 160 * 1024 &&
           (long)a.B * (a.H / 8) * (a.W / 16) * (a.CoutP / 32) * 8 < (1L << 31) &&
           (a.ksplit >= 2 ? ((a.ksplit == 2 || a.ksplit == 4 || a.ksplit == 8) && nchunks % a.ksplit == 0 && nchunks / a.ksplit >= WP_MINCH &&
                             conv_part_fits(a))
                          : nchunks >= WP_MINCH);
}

int launch_conv_wino3p(const ConvArgs& a, hipStream_t s) {
    MCVD_REQUIRE(conv_wino3p_usable(a), "persistent winograd bf16x3 conv: unsupported (ks=%d H=%d W=%d Cin=%d C0=%d ksplit=%d, packed weight pieces %s)",
                 a.ks, a.H, a.W, a.Cin, a.C0, a.ksplit, a.wpb ? "present" : "missing");
    const int cot = conv_wino_cout_tile(a.Cout);
    MCVD_REQUIRE(a.CoutP % (32 * cot) == 0, "persistent winograd bf16x3 conv: CoutP=%d vs tile %d", a.CoutP, 32 * cot);
    switch (cot) {
        case 1: return wino3p_launch1<1>(a, s);
        case 2: return wino3p_launch1<2>(a, s);
        default: return wino3p_launch1<3>(a, s);
    }
}

}  // namespace mcvd
