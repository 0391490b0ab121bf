// Fused conv (3x3 / 1x1) as an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, gfx950).
//
//   y[b, co, p] = out_scale * ( bias[co] + res[b, co, p] + sum_{ci,tap} W[co, ci, tap] * f(x)[b, ci, p + tap] )
//   f(x)[b, ci, .] = silu?( A[b,ci] * x[b,ci,.] + B[b,ci] )      (GroupNorm + temb scale/shift folded to A,B)
//
// MFMA roles: A-operand = weights (rows = output channels), B-operand = activations (cols = pixels), so an
// accumulator register holds 32 consecutive pixels of one output channel across lanes -> NCHW stores and
// residual loads are 128-byte coalesced rows.
//
// Work decomposition (256 threads = 4 wave64):
//   pixel tile  = BPX pixels = whole image rows (full width, so left/right halo is always the zero padding),
//   cout tile   = 32*COT output channels,
//   K loop      = input-channel chunks of CK channels staged in LDS (im2col is implicit: the 9 taps are LDS
//                 address offsets into the staged halo patch),
//   SPLIT=false : the 4 waves split the pixel tile (PXT 32-pixel sub-tiles each),
//   SPLIT=true  : the 4 waves split K inside each chunk and reduce through LDS (small-resolution layers,
//                 where M = B*H*W is too small to fill 256 CUs otherwise).
// Pipeline: global->register prefetch of chunk i+1 is issued before the MFMAs of chunk i; the prologue
// transform is applied on the way from registers to LDS.
#pragma once
#include "../common.h"
#ifdef MCVD_DIAG_LDS_PAD
#include <stdlib.h>
#endif

namespace mcvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float silu_f(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}

struct ConvGeom {
    int RT;        // image rows per pixel tile
    int rpi;       // rows per image inside a tile (= min(RT, H))
    int nimg;      // images per tile (RT / rpi)
    int P;         // LDS row pitch (floats)
    int IS;        // LDS floats per (channel, image)
    int PS;        // LDS floats per channel
    int n_ptiles;  // pixel tiles in the grid
};

template <int KS, int CK, int COT, int PXT, bool SPLIT>
struct ConvCfg {   // tile geometry (independent of how the weights are staged)
    static constexpr int KK = KS * KS;
    static constexpr int HALO = (KS == 3) ? 1 : 0;
    static constexpr int BPX = SPLIT ? PXT * 32 : 4 * PXT * 32;
    static constexpr int BCO = COT * 32;
    static constexpr int MAXA = (KS == 3) ? 4 : (CK * BPX / 4 + 255) / 256;
    static constexpr int WCOUNT = CK * KK * BCO / 4;          // float4 per weight chunk
    static constexpr int MAXW = (WCOUNT + 255) / 256;
    static constexpr int WSZ = CK * KK * BCO;                 // floats per weight chunk
    // Double-buffer the weight chunk when it is small enough to keep two workgroups per CU: the DMA for chunk i+1 is
    // then issued BEFORE the MFMAs of chunk i and its latency disappears behind them.
    static constexpr bool WDB = WSZ * 4 <= 30 * 1024;
    static_assert(WCOUNT % 64 == 0, "weight chunk must be wave-granular for the LDS-DMA path");
};

// WDMA: weight chunks go global -> LDS by LDS-DMA (global_load_lds_dwordx4: the packed weight slab of a chunk is one
// linear LDS image, wave-uniform base + lane*16) instead of through 28-36 staging VGPRs; that is what lets the 256-pixel
// x 96-cout tile fit two workgroups per CU, so one block's barrier/staging/epilogue phases hide under the other's MFMAs.
template <int KS, int CK, int COT, int PXT, bool SPLIT, bool WDMA, bool WDBF = false>
__global__ __launch_bounds__(256, (COT * PXT >= 8 || (COT * PXT >= 6 && !SPLIT && !WDMA)) ? 1 : 2) void conv_mfma_kernel(ConvArgs a, ConvGeom g) {
    using Cfg = ConvCfg<KS, CK, COT, PXT, SPLIT>;
    constexpr int KK = Cfg::KK, HALO = Cfg::HALO, BCO = Cfg::BCO, MAXA = Cfg::MAXA, MAXW = Cfg::MAXW;
    constexpr bool WDB = WDMA && (Cfg::WDB || WDBF);      // WDBF: double-buffer even a large weight chunk (one block per CU)
    // (A variant that double-buffered the activation patch as well, one barrier per chunk, measured 0-5 % slower with two
    // workgroups per CU -- profiles/r01_conv_phase_breakdown_pipe2.txt -- and was removed.)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sW = smem + CK * g.PS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int H = a.H, W = a.W, HW = H * W;
    const int Cin = a.Cin;

    const int ptile = blockIdx.x;
    const int co0 = blockIdx.y * BCO;
    const long grow0 = (long)ptile * g.RT;        // first row of the tile in flattened (b, y) row space
    const int b0 = (int)(grow0 / H);
    const int y0 = (int)(grow0 - (long)b0 * H);

    // ---------------- per-thread staging slots (chunk invariant) ----------------
    const int W4 = W >> 2;
    const int rows_l = g.rpi + 2 * HALO;
    const int per_cin = g.nimg * rows_l * W4;
    const int countA = CK * per_cin;
    int a_lds[MAXA], a_goff[MAXA], a_cb[MAXA];     // a_cb = (cin_l << 16) | (img << 1) | inimage ; -1 = unused slot
#pragma unroll
    for (int s = 0; s < MAXA; ++s) {
        const int e = s * 256 + tid;
        if (e < countA) {
            const int cin_l = e / per_cin;
            const int rem = e - cin_l * per_cin;
            const int rowi = rem / W4;
            const int c4 = rem - rowi * W4;
            const int img = rowi / rows_l;
            const int rl = rowi - img * rows_l;
            const int yy = y0 + rl - HALO;
            const int inimg = (yy >= 0 && yy < H && (b0 + img) < a.B) ? 1 : 0;
            a_lds[s] = cin_l * g.PS + img * g.IS + rl * g.P + (HALO ? 4 : 0) + c4 * 4;
            a_goff[s] = yy * W + c4 * 4;
            a_cb[s] = (cin_l << 16) | (img << 1) | inimg;
        } else {
            a_lds[s] = 0; a_goff[s] = 0; a_cb[s] = -1;
        }
    }
    int w_goff[MAXW];
#pragma unroll
    for (int s = 0; s < MAXW; ++s) {
        const int e = s * 256 + tid;
        const int row = e / (BCO / 4);
        const int c4 = e - row * (BCO / 4);
        w_goff[s] = (e < Cfg::WCOUNT) ? row * a.CoutP + co0 + c4 * 4 : -1;
    }

    // zero the activation patch once: halo columns (and unused pad) stay zero for the whole kernel
    for (int i = tid; i < CK * g.PS; i += 256) sA[i] = 0.0f;

    f32x4 ra[MAXA];      // native vector types: plain load/store, no struct memcpy (keeps them in VGPRs)
    f32x2 rc[MAXA];
    f32x4 rw[WDMA ? 1 : MAXW];

    // Loads are unconditional (invalid slots read a safe in-bounds address and are discarded at write time) so the
    // staging registers are always defined; plain macros (not lambdas) keep them out of scratch.
#define MCVD_LOAD_CHUNK(ch)                                                                                          \
    {                                                                                                                \
        const int cbase = (ch) * CK;                                                                                 \
        _Pragma("unroll") for (int s = 0; s < MAXA; ++s) {                                                           \
            const int c = cbase + (a_cb[s] >> 16);                                                                   \
            const int b = b0 + ((a_cb[s] >> 1) & 0x7fff);                                                            \
            const bool ok = (a_cb[s] >= 0) && (a_cb[s] & 1) && (c < Cin);                                            \
            const float* src = a.x0;                                                                                 \
            const float* csrc = a.coef ? a.coef : a.bias;                                                            \
            if (ok) {                                                                                                \
                src = ((c < a.C0) ? a.x0 + ((long)b * a.C0 + c) * HW                                                 \
                                  : a.x1 + ((long)b * a.C1 + (c - a.C0)) * HW) + a_goff[s];                          \
                if (a.coef) csrc = a.coef + ((long)b * Cin + c) * 2;                                                 \
            }                                                                                                        \
            ra[s] = *reinterpret_cast<const f32x4*>(src);                                                            \
            rc[s] = *reinterpret_cast<const f32x2*>(csrc);                                                           \
        }                                                                                                            \
        if (!WDMA) {                                                                                                 \
            const float* wsrc = a.wp + (long)cbase * KK * a.CoutP;                                                   \
            _Pragma("unroll") for (int s = 0; s < MAXW; ++s)                                                         \
                rw[s] = *reinterpret_cast<const f32x4*>(wsrc + (w_goff[s] >= 0 ? w_goff[s] : 0));                    \
        }                                                                                                            \
        if (WDB) { /* double-buffered: this chunk's weights fly into the idle buffer during the previous chunk's MFMAs */ \
            const float* wsrc = a.wp + (long)cbase * KK * a.CoutP;                                                   \
            float* wdst = sW + (((ch) & 1) ? Cfg::WSZ : 0);                                                          \
            _Pragma("unroll") for (int s = 0; s < MAXW; ++s) {                                                       \
                if (w_goff[s] >= 0)                                                                                  \
                    __builtin_amdgcn_global_load_lds(                                                                \
                        (const __attribute__((address_space(1))) void*)(wsrc + w_goff[s]),                           \
                        (__attribute__((address_space(3))) void*)(wdst + (s * 256 + wave * 64) * 4), 16, 0, 0);      \
            }                                                                                                        \
        }                                                                                                            \
    }
#ifdef MCVD_DIAG_COEF_BRANCH      /* diagnostic builds: keep `if (a.coef)` a branch (hipcc turns it into v_cndmask under an SGPR mask) */
#define MCVD_DIAG_BRANCH_HERE asm volatile("" ::: "memory");
#else
#define MCVD_DIAG_BRANCH_HERE
#endif
#define MCVD_WRITE_CHUNK(ch)                                                                                         \
    {                                                                                                                \
        const int cbase = (ch) * CK;                                                                                 \
        if (WDMA && !WDB) { /* single buffer: issue the DMA first, it flies while the activation patch is written */ \
            const float* wsrc = a.wp + (long)cbase * KK * a.CoutP;                                                   \
            _Pragma("unroll") for (int s = 0; s < MAXW; ++s) {                                                       \
                if (w_goff[s] >= 0) /* wave-granular: WCOUNT is a multiple of 64 */                                  \
                    __builtin_amdgcn_global_load_lds(                                                                \
                        (const __attribute__((address_space(1))) void*)(wsrc + w_goff[s]),                           \
                        (__attribute__((address_space(3))) void*)(sW + (s * 256 + wave * 64) * 4), 16, 0, 0);        \
            }                                                                                                        \
        }                                                                                                            \
        _Pragma("unroll") for (int s = 0; s < MAXA; ++s) {                                                           \
            if (a_cb[s] >= 0) {                                                                                      \
                f32x4 v = ra[s];                                                                                     \
                const bool live = (a_cb[s] & 1) && (cbase + (a_cb[s] >> 16) < Cin);                                  \
                if (live) {                                                                                          \
                    if (a.coef) {                                                                                    \
                        MCVD_DIAG_BRANCH_HERE                                                                        \
                        v.x = fma_unpacked(v.x, rc[s].x, rc[s].y); v.y = fma_unpacked(v.y, rc[s].x, rc[s].y);        \
                        v.z = fma_unpacked(v.z, rc[s].x, rc[s].y); v.w = fma_unpacked(v.w, rc[s].x, rc[s].y);        \
                    }                                                                                                \
                    if (a.act) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }       \
                } else {                                                                                             \
                    v = f32x4{0.f, 0.f, 0.f, 0.f}; /* zero padding applies AFTER the activation */                   \
                }                                                                                                    \
                *reinterpret_cast<f32x4*>(sA + a_lds[s]) = v;                                                        \
            }                                                                                                        \
        }                                                                                                            \
        if (!WDMA) {                                                                                                 \
            _Pragma("unroll") for (int s = 0; s < MAXW; ++s) {                                                       \
                if (w_goff[s] >= 0) *reinterpret_cast<f32x4*>(sW + (s * 256 + tid) * 4) = rw[s];                     \
            }                                                                                                        \
        }                                                                                                            \
    }

    // ---------------- this lane's MFMA operand addresses ----------------
    const int wpx0 = SPLIT ? 0 : wave * PXT * 32;
    int pixoff[PXT];
#pragma unroll
    for (int pt = 0; pt < PXT; ++pt) {
        const int m = wpx0 + pt * 32 + l31;
        const int rowt = m / W;
        const int c = m - rowt * W;
        const int img = rowt / g.rpi;
        const int r = rowt - img * g.rpi;
        pixoff[pt] = img * g.IS + (r + HALO) * g.P + (HALO ? 4 : 0) + c + half * g.PS;
    }
    const int woff = half * KK * BCO + l31;

    // Output coordinates of this lane's accumulator columns: 32-bit per-lane element offsets (pixel + the half-wave's
    // 4-channel shift); the per-register channel offset is wave-uniform and lives in SGPRs (saddr-form global ops).
    unsigned voff[PXT];
    bool pvalid[PXT];
#pragma unroll
    for (int pt = 0; pt < PXT; ++pt) {
        const int m = wpx0 + pt * 32 + l31;
        const int rowt = m / W;
        const int c = m - rowt * W;
        const int img = rowt / g.rpi;
        const int r = rowt - img * g.rpi;
        const int b = b0 + img;
        pvalid[pt] = b < a.B;
        voff[pt] = (unsigned)((long)b * a.Cout * HW + (long)(y0 + r) * W + c + (long)4 * half * HW);
    }

    // Accumulators start at bias (+ residual): the residual/bias loads overlap the first staging loads instead of
    // sitting, latency-exposed, in the epilogue.  All loads are UNCONDITIONAL (safe addresses + select): a load under a
    // per-lane predicate makes hipcc branch around it and wait vmcnt(0) per element -- 96 serialised round trips.
    // a.bias is zero-padded to CoutP.  (SPLIT keeps bias/residual for the reducer's epilogue: small layers, and the
    // up-front loads would cost the split kernel its second wave per SIMD.)
    const bool full_tile = co0 + BCO <= a.Cout;      // wave-uniform
    unsigned roff[PXT];                              // residual offsets, clamped to a valid element for dead lanes
#pragma unroll
    for (int pt = 0; pt < PXT; ++pt) roff[pt] = pvalid[pt] ? voff[pt] : (unsigned)(4 * half * HW);
    f32x16 acc[COT][PXT];
#pragma unroll
    for (int ct = 0; ct < COT; ++ct)
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int cos = co0 + ct * 32 + (rg & 3) + 8 * (rg >> 2);          // wave-uniform part of the channel
            const float bv = SPLIT ? 0.0f : a.bias[cos + 4 * half];
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt) acc[ct][pt][rg] = bv;
        }
    if (!SPLIT && a.res) {
        if (full_tile) {
            // one 32-channel sub-tile at a time: 16*PXT independent loads in flight, then the adds.  The scheduling
            // barriers keep hipcc from serialising them into load-wait-add pairs under register pressure (48 round trips).
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) {
                float tmp[16][PXT];
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const float* rb = a.res + (long)(co0 + ct * 32 + (rg & 3) + 8 * (rg >> 2)) * HW;
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt) tmp[rg][pt] = rb[roff[pt]];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rg = 0; rg < 16; ++rg)
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt) acc[ct][pt][rg] += pvalid[pt] ? tmp[rg][pt] : 0.0f;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {      // ragged cout tile (never on the UNet's residual convs): predicated loads
#pragma unroll
            for (int ct = 0; ct < COT; ++ct)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int cos = co0 + ct * 32 + (rg & 3) + 8 * (rg >> 2);
                    const float* rb = a.res + (long)cos * HW;
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt)
                        if (cos + 4 * half < a.Cout && pvalid[pt]) acc[ct][pt][rg] += rb[voff[pt]];
                }
        }
    }

    // optional phase timing (wave 0 of every block; shader cycles): 0 prologue, 1 MFMA phases, 2 barrier after MFMA (incl.
    // vmcnt drain), 3 staging writes, 4 second barrier, 5 split-K reduction, 6 epilogue, 7 total
    unsigned long long tk0 = 0, tacc[5] = {0, 0, 0, 0, 0}, tprev = 0;
    if (a.dbg) tk0 = tprev = __builtin_amdgcn_s_memtime();
#define MCVD_STAMP(i)                                                   \
    if (a.dbg) {                                                        \
        const unsigned long long tn = __builtin_amdgcn_s_memtime();     \
        tacc[i] += tn - tprev;                                          \
        tprev = tn;                                                     \
    }

    // LDS-DMA and barriers.  A weight chunk that arrives by global_load_lds is written to LDS ASYNCHRONOUSLY, tracked by the issuing wave's
    // vmcnt only.  To the compiler the instruction is a load without a destination register: nothing makes it wait for it in front of an
    // s_barrier (a release fence does not wait for loads), and a wait it inserts later, in front of the wave's own LDS reads, covers that
    // wave's part of the chunk but not the parts the other waves fetched.  So the wave that issued a DMA waits for it EXPLICITLY before the
    // barrier that publishes the chunk.  Round 6 found the single-buffered path (WDMA && !WDB) without any such wait -- the generated code
    // had s_waitcnt lgkmcnt(0); s_barrier behind the DMA -- i.e. a timing-dependent read of a weight chunk that has not landed: correct when
    // the kernel runs alone (the activation patch takes longer to write than the DMA takes to land), wrong in half of the launches beside a
    // memory-hungry kernel on another stream (tools/repro_coresident.cpp, profiles/r06_coresident_repro.txt).
#define MCVD_DMA_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
    const int nchunks = a.CinP / CK;
    MCVD_LOAD_CHUNK(0);
    if (WDB) MCVD_DMA_LANDED();
    __syncthreads();              // zero fill done
    MCVD_WRITE_CHUNK(0);
    if (WDMA && !WDB) MCVD_DMA_LANDED();
    __syncthreads();
    MCVD_STAMP(0)

    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) MCVD_LOAD_CHUNK(ch + 1);
        const float* sWc = sW + ((WDB && (ch & 1)) ? Cfg::WSZ : 0);
        // ---- MFMA over this chunk
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int tapoff = HALO ? ((tap / 3) - 1) * g.P + ((tap % 3) - 1) : 0;
            constexpr int NKP = CK / 2;
#pragma unroll
            for (int kq = 0; kq < (SPLIT ? NKP / 4 : NKP); ++kq) {
                const int kp = SPLIT ? (wave + 4 * kq) : kq;
                float aw[COT], bx[PXT];
#pragma unroll
                for (int ct = 0; ct < COT; ++ct) aw[ct] = sWc[(2 * kp * KK + tap) * BCO + ct * 32 + woff];
#pragma unroll
                for (int pt = 0; pt < PXT; ++pt) bx[pt] = sA[2 * kp * g.PS + pixoff[pt] + tapoff];
#pragma unroll
                for (int ct = 0; ct < COT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PXT; ++pt)
                        acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[ct], bx[pt], acc[ct][pt], 0, 0, 0);
            }
        }
        MCVD_STAMP(1)
        if (WDB) MCVD_DMA_LANDED();          // chunk ch + 1's weights (requested above, into the idle buffer) are published by this barrier
        __syncthreads();
        MCVD_STAMP(2)
        if (ch + 1 < nchunks) {
            MCVD_WRITE_CHUNK(ch + 1);
            MCVD_STAMP(3)
            if (WDMA && !WDB) MCVD_DMA_LANDED();   // single buffer: the DMA was issued inside MCVD_WRITE_CHUNK
            __syncthreads();
            MCVD_STAMP(4)
        }
    }
    unsigned long long t_loop_end = a.dbg ? __builtin_amdgcn_s_memtime() : 0;

    // ---------------- split-K reduction across the 4 waves (through LDS) ----------------
    // Wave 0 is the reducer; it also adds bias + residual here, one 32x32 tile per round, so only 16 (unconditional)
    // residual loads are in flight at a time and they overlap the other waves' LDS writes.
    if (SPLIT) {
        float* red = smem;      // 3 * 1024 floats, LDS is free now (last barrier above passed)
#pragma unroll
        for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt) {
                float rr[16];
                if (wave == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cos = co0 + ct * 32 + (r & 3) + 8 * (r >> 2);
                        float v = a.bias[cos + 4 * half];                         // zero-padded to CoutP
                        if (a.res) {
                            if (full_tile) {
                                const float x = a.res[(long)cos * HW + roff[pt]];
                                v += pvalid[pt] ? x : 0.0f;
                            } else if (cos + 4 * half < a.Cout && pvalid[pt]) {
                                v += a.res[(long)cos * HW + voff[pt]];
                            }
                        }
                        rr[r] = v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(wave - 1) * 1024 + r * 64 + lane] = acc[ct][pt][r];
                }
                __syncthreads();
                if (wave == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[ct][pt][r] += (red[r * 64 + lane] + red[1024 + r * 64 + lane]) + (red[2048 + r * 64 + lane] + rr[r]);
                }
                __syncthreads();
            }
        if (wave != 0) return;
    }
    const unsigned long long t_red_end = a.dbg ? __builtin_amdgcn_s_memtime() : 0;

    // ---------------- epilogue: scale, coalesced NCHW stores (bias/residual are already in the accumulators)
    // GroupNorm partials of the FINAL values (ConvArgs::stats; round 5: the stem runs this kernel, and without them BOTH norms that read its
    // output -- the first ResBlock's and, through the skip stack, the last up block's concat norm -- took a pass over the tensor, the only two
    // gn_coef launches of a forward, 29 us each).  Non-split tiles of whole rows of ONE image: a wave's PXT * 32 pixels are one contiguous
    // pixel block of the image = one partial; the 32 lanes of a half-wave hold one cout.  Pilot-shifted moments as in conv_wino3.cpp.
    const bool emit_stats = !SPLIT && a.stats != nullptr && g.nimg == 1 && pvalid[0];
    const int st_np = HW / (PXT * 32), st_p = (y0 * W + wpx0) / (PXT * 32);
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int cos = co0 + ct * 32 + (rg & 3) + 8 * (rg >> 2);
            float* yb = a.y + (long)cos * HW;
            float vv[PXT];
#pragma unroll
            for (int pt = 0; pt < PXT; ++pt) vv[pt] = acc[ct][pt][rg] * a.out_scale;
            if (cos + 4 * half < a.Cout) {
#pragma unroll
                for (int pt = 0; pt < PXT; ++pt)
                    if (pvalid[pt]) yb[voff[pt]] = vv[pt];
            }
            if (!SPLIT && emit_stats) {
                const int pv = __builtin_bit_cast(int, vv[0]);
                const int s0 = __builtin_amdgcn_readlane(pv, 0), s2 = __builtin_amdgcn_readlane(pv, 32);
                const float pil = __builtin_bit_cast(float, (lane & 32) ? s2 : s0);
                float sm = 0.0f, qm = 0.0f;
#pragma unroll
                for (int pt = 0; pt < PXT; ++pt) {
                    const float d = vv[pt] - pil;
                    sm += d;
                    qm += d * d;
                }
#define MCVD_MERGE(CTRL, ROWMASK)                                                                                     \
                {                                                                                                   \
                    sm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sm), CTRL, ROWMASK, 0xf, false)); \
                    qm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, qm), CTRL, ROWMASK, 0xf, false)); \
                }
                MCVD_MERGE(0xB1, 0xf)                 // quad_perm [1,0,3,2]
                MCVD_MERGE(0x4E, 0xf)                 // quad_perm [2,3,0,1]
                MCVD_MERGE(0x124, 0xf)                // row_ror:4
                MCVD_MERGE(0x128, 0xf)                // row_ror:8: every lane of a row of 16 holds the row's totals
                MCVD_MERGE(0x142, 0xa)                // row_bcast:15: lanes 16-31 / 48-63 add the totals of the row below
#undef MCVD_MERGE
                const int co = cos + 4 * half;
                if (l31 == 31 && co < a.Cout) {
                    const float npix = (float)(PXT * 32);
                    float* q = a.stats + (((long)b0 * a.Cout + co) * st_np + st_p) * 2;
                    q[0] = sm + npix * pil;
                    q[1] = fmaxf(qm - sm * sm * (1.0f / npix), 0.0f);
                }
            }
        }
    }
    if (a.dbg && tid == 0) {
        const unsigned long long te = __builtin_amdgcn_s_memtime();
        unsigned long long* d = a.dbg + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        d[0] = tacc[0]; d[1] = tacc[1]; d[2] = tacc[2]; d[3] = tacc[3]; d[4] = tacc[4];
        d[5] = t_red_end - t_loop_end; d[6] = te - t_red_end; d[7] = te - tk0;
    }
#undef MCVD_STAMP
#undef MCVD_LOAD_CHUNK
#undef MCVD_WRITE_CHUNK
}

// Host-side geometry + launch for one instantiation.
template <int KS, int CK, int COT, int PXT, bool SPLIT, bool WDMA, bool WDBF = false>
int conv_mfma_launch(const ConvArgs& a, hipStream_t s) {
    using Cfg = ConvCfg<KS, CK, COT, PXT, SPLIT>;
    ConvGeom g;
    const int BPX = Cfg::BPX;
    MCVD_REQUIRE(BPX % a.W == 0, "conv: pixel tile %d not a multiple of W=%d", BPX, a.W);
    g.RT = BPX / a.W;
    g.rpi = g.RT < a.H ? g.RT : a.H;
    MCVD_REQUIRE(g.RT % g.rpi == 0 && a.H % g.rpi == 0, "conv: tile rows %d vs H=%d", g.RT, a.H);
    g.nimg = g.RT / g.rpi;
    if (Cfg::HALO) {
        // row = [3 pad | left halo | W pixels]; the right halo of a row is the (always zero) first pad slot of the next row
        g.P = (a.W == 8) ? 20 : a.W + 4;
        g.IS = (g.rpi + 2) * g.P;
    } else {
        g.P = a.W;
        g.IS = g.rpi * g.P;
    }
    g.PS = round_up(g.nimg * g.IS + (Cfg::HALO ? 4 : 0), 4);    // +4: the last row's right halo spills one slot past IS
    const long rows = (long)a.B * a.H;
    g.n_ptiles = (int)((rows + g.RT - 1) / g.RT);
    const int countA = CK * g.nimg * (g.rpi + 2 * Cfg::HALO) * (a.W / 4);
    MCVD_REQUIRE(countA <= Cfg::MAXA * 256, "conv: staging slots exceeded (%d > %d)", countA, Cfg::MAXA * 256);
    MCVD_REQUIRE((double)a.B * a.Cout * a.H * a.W < 4.0e9, "conv: output tensor exceeds 32-bit element offsets");
    MCVD_REQUIRE(a.CinP % CK == 0 && a.CoutP % Cfg::BCO == 0, "conv: packed dims (%d,%d) vs chunk %d tile %d",
                 a.CinP, a.CoutP, CK, Cfg::BCO);
    constexpr bool WDB = WDMA && (Cfg::WDB || WDBF);
    size_t lds = (size_t)(CK * g.PS + (WDB ? 2 : 1) * Cfg::WSZ) * sizeof(float);
    if (SPLIT && lds < 3 * 1024 * sizeof(float)) lds = 3 * 1024 * sizeof(float);
#ifdef MCVD_DIAG_LDS_PAD      // diagnostic builds only (tools/build_variant.sh): ask for more LDS than the kernel uses (co-residency experiments)
    if (const char* pad = getenv("MCVD_DIAG_LDS_PAD")) lds += (size_t)atoi(pad);
#endif
    MCVD_REQUIRE(lds <= 160 * 1024, "conv: LDS %zu > 160KiB", lds);
    if (lds > 64 * 1024) {      // above the default dynamic-LDS limit: opt in once per instantiation
        static PerDeviceOnce raised;
        if (raised.first_use()) {
            MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<KS, CK, COT, PXT, SPLIT, WDMA, WDBF>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            raised.done();
        }
    }
    dim3 grid(g.n_ptiles, a.CoutP / Cfg::BCO);
    hipLaunchKernelGGL((conv_mfma_kernel<KS, CK, COT, PXT, SPLIT, WDMA, WDBF>), grid, dim3(256), lds, s, a, g);
    MCVD_HIP_CHECK(hipGetLastError());
    if (!SPLIT && a.stats && g.nimg == 1) set_last_conv_stats_np(a.H * a.W / (PXT * 32));      // one partial per wave's pixel block (epilogue)
    return 0;
}

// one translation unit per (KS, COT) instantiates the three tile shapes
template <int KS, int CK, int COT>
int conv_mfma_dispatch_shape(const ConvArgs& a, int shape, hipStream_t s) {
    const bool dma = a.wdma != 0;
    constexpr int CK2 = (KS == 3 && COT <= 3) ? 2 * CK : CK;
    switch (shape) {
        case 0: return dma ? conv_mfma_launch<KS, CK, COT, 2, false, true>(a, s) : conv_mfma_launch<KS, CK, COT, 2, false, false>(a, s);   // 256-pixel tile
        case 1: return dma ? conv_mfma_launch<KS, CK, COT, 1, false, true>(a, s) : conv_mfma_launch<KS, CK, COT, 1, false, false>(a, s);   // 128-pixel tile
        // 64-pixel tile, the 4 waves split K; 3x3 uses a 16-channel chunk there (more MFMAs per barrier; CinP is packed to 16)
        case 2: return dma ? conv_mfma_launch<KS, CK2, COT, 2, true, true>(a, s) : conv_mfma_launch<KS, CK2, COT, 2, true, false>(a, s);
        // same split-K tile with the (large) weight chunk double-buffered: one block per CU, but the ~100-cycle-per-KiB DMA issue
        // and its latency ride under the MFMAs -- for layers that have <= 256 tiles anyway (8x8 at B=64)
        case 3: return conv_mfma_launch<KS, CK2, COT, 2, true, true, true>(a, s);
    }
    set_error("conv: bad shape id %d", shape);
    return -1;
}

int conv3_cot1(const ConvArgs&, int, hipStream_t);
int conv3_cot2(const ConvArgs&, int, hipStream_t);
int conv3_cot3(const ConvArgs&, int, hipStream_t);
int conv3_cot4(const ConvArgs&, int, hipStream_t);
int conv1_cot1(const ConvArgs&, int, hipStream_t);
int conv1_cot2(const ConvArgs&, int, hipStream_t);
int conv1_cot3(const ConvArgs&, int, hipStream_t);
int conv1_cot4(const ConvArgs&, int, hipStream_t);

}  // namespace mcvd
