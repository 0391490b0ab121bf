// 1x1 convolution / NIN as a GEMM on the 16-bit matrix pipes with split operands (pieces.h has the arithmetic):
//   y[b][co][p] = out_scale * ( sum_ci W[ci][co] * pro(x[b][ci][p]) + bias[co] (+ res[b][co][p]) )
// NP = 3: three bf16 pieces per operand, six piece products -- fp32-equivalent, full fp32 range; the default (shape id 15).
// NP = 2: two fp16 pieces per operand, three piece products (22-bit operands; context option "f16x2"; shape id 14).
// The fp32-MFMA form of this GEMM (conv1x1_dma.cpp) is bound by the matrix pipe (66 % busy with no memory traffic at all,
// profiles/r02_conv1x1_ablation.txt); NP * (NP + 1) / 2 MFMAs of the 16x faster pipe per 16 channels take 3/16 (6/16) of that pipe
// time, which leaves the kernel with what a 1x1 conv should be bound by: reading the activations once.
//   * weights: split once, when the weights are packed (launch_pack_conv1x1_h2; NP = 2: per layer scaled by a power of two, max |w|
//     at 2^13..2^14, the inverse folded into the epilogue; NP = 3: unscaled), stored operand-major
//         [16-channel chunk][32-cout sub-tile][piece][64 lanes][4 dwords],  dword j of lane (h, m) = K slots (2j, 2j+1) = channels
//         (4j + h, 4j + 2 + h) of the chunk for cout m of the sub-tile (the K-slot convention of conv_wino2h.cpp),
//     so the COT sub-tiles of a workgroup are one contiguous block per chunk: it reaches the LDS by LDS-DMA (16 B per lane) and
//     every wave reads its A operands from there with one conflict-free ds_read_b128 per (sub-tile, piece).
//   * pixels x[b][ci][HW]: rows are contiguous pixels (NCHW) -> LDS-DMA, as in conv1x1_dma.cpp.  A wave owns 32 pixels; lane
//     (pixel n, half h) reads its 8 channels of a 16-channel step (ds_read_b32, conflict-free), applies the GroupNorm affine
//     (+ SiLU) from the LDS coefficient table and splits (NP = 2: times 2^4 first; no clamp -- an input beyond the fp16 range
//     becomes Inf and the output NaN, which the library reports: model.cpp range guard).
//   * K loop over 16-channel chunks, ONE barrier per chunk, FOUR LDS buffers: the DMA runs three chunks ahead of the MFMAs (a
//     chunk holds 3-6 * COT MFMAs = 200-800 cycles of matrix work, less than a memory latency; with two buffers the kernel
//     waited for every chunk).  The DMA groups are the only VMEM operations in flight: their waits are counted by hand.
// Workgroup = 4 waves = 128 consecutive pixels of the flattened [B*HW] axis x 32*COT couts; block id -> (pixel tile, cout tile)
// keeps the cout tiles of one pixel tile on one XCD (conv1x1_dma.cpp).  MFMAs and LDS traffic are compiler-scheduled builtins.
#include <stdlib.h>

#include "../common.h"
#include "pieces.h"

namespace mcvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_q(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

constexpr int Q1_PT = 128;       // pixels per workgroup
constexpr int Q1_CK = 16;        // input channels per chunk (one barrier, one MFMA K step)
constexpr int Q1_NB = 4;         // LDS buffers: the DMA runs three chunks ahead of the MFMAs
constexpr int Q1_MAXIMG = 4;     // images a pixel tile may span (HW >= 32)

// PRO: 0 raw input, 1 affine, 2 affine + SiLU
// EXP != 0 (built with -DMCVD_DIAG only; env MCVD_Q1_EXP): timing-only ablations of the K loop, wrong results: bit 0 no DMA behind the
// first three chunks, bit 1 no MFMA, bit 2 no affine / SiLU / split, bit 3 no barrier; bit 4 (right results): the split is not pinned in front of the next barrier
// KV (NP = 3, PRO = 1: the fused q|k|v projection of an attention block, round 5): the K and V thirds of the output leave the kernel
//     ALREADY SPLIT into the three bf16 pieces, as the LDS images attn_h2p_kernel (attention_h2.cpp) stages by LDS-DMA -- per (sample,
//     head, key tile of 32 pixels) [K image: step][piece][dword j][64 lanes] [V image: sub-tile][step s2][piece][64 lanes][4 dwords] --
//     instead of as fp32 rows (which nothing reads any more); the Q third is stored as before.  The attention kernel split every K / V
//     element once per QUERY TILE (8 times at 32 x 32) and spent a third of its VALU instructions on it.
// IM (NP = 3, PRO = 0; ConvArgs::im2col; north_star: "LDS-staged im2col"): the B operand of a 3x3 conv run as a GEMM (the stem: 10 input channels,
//     90 K rows) is gathered from a patch of the RAW input the workgroup stages once -- [C0 + C1 + 1 planes][PT / W + 2 rows][W + 2], zero outside
//     the image (W + 8 columns in the kernel: 16-byte aligned rows), the extra plane all zeros for the padding rows of K -- through a table of patch offsets per K row; no pixel DMA, no pixel
//     ring buffers, no `col` tensor in HBM (round 5 wrote and re-read 100 MB of it per forward).  Same K order, same MFMA order: bit-identical
//     to the GEMM over the materialised im2col.
template <int NP, int COT, int PRO, int EXP = 0, bool KV = false, bool IM = false>
__global__ __launch_bounds__(256, 2) void conv1x1_h2_kernel(ConvArgs a, int ptiles, int nct) {
    typedef Pieces<NP> PX;
    constexpr int PT = Q1_PT, CK = Q1_CK, NB = Q1_NB, BCO = 32 * COT;
    constexpr int PPR = PT / 4;               // 16-byte pieces per channel row of the pixel tile
    constexpr int RPS = 256 / PPR;            // channel rows one 256-thread DMA step covers (8)
    constexpr int XSZ = CK * PT;              // floats of one pixel chunk
    constexpr int WDW = COT * NP * 256;       // dwords of one weight chunk: [COT][NP pieces][64 lanes][4]
    constexpr int WPC = WDW / 4;              // its 16-byte pieces (COT * NP * 64)
    constexpr int MAXX = XSZ / 4 / 256;       // x DMA rounds per chunk (2)
    constexpr int MAXW = (WPC + 255) / 256;   // weight DMA rounds per chunk; a partial round re-fetches pieces from the start (same data, same place)
    constexpr int G = (IM ? 0 : MAXX) + MAXW; // DMA instructions per wave and chunk: the unit of the vmcnt bookkeeping below
    static_assert(MAXX * 256 * 4 == XSZ && WPC % 64 == 0, "DMA rounds");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sW = reinterpret_cast<unsigned*>(smem);      // [NB][WDW]
    float* sX = smem + NB * WDW;              // [NB][CK][PT]   (IM: no pixel ring -- the patch and the offset table live here)
    float* sC = sX + (IM ? 0 : NB * XSZ);     // [nimg][Cin][2] prologue coefficients of the images this pixel tile spans (PRO only)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ptile = (slot / nct) * 8 + xcd, ctile = slot - (slot / nct) * nct;
    if (ptile >= ptiles) return;
    const int co0 = ctile * BCO;
    const int HW = a.H * a.W, Cin = a.Cin;
    const long NPX = (long)a.B * HW;
    const long gp0 = (long)ptile * PT;
    const int NS = a.CoutP / 32;              // 32-cout sub-tiles of the layer

    // ---- x DMA role: piece e = s*256 + tid -> (channel-in-chunk e / PPR, 4 pixels (e % PPR)*4); pixel part is slot invariant
    long xg = gp0 + (tid % PPR) * 4;
    if (xg > NPX - 4) xg = NPX - 4;           // ragged last tile: fetch valid data, the stores are predicated
    const int xb = (int)(xg / HW), xp = (int)(xg - (long)xb * HW);
    const int x_ci = tid / PPR;               // + RPS*s
    const int voff0 = (xb * a.C0 + x_ci) * HW + xp;               // offset inside x0 (without the chunk base)
    const int voff1 = (xb * a.C1 + x_ci) * HW + xp;               // offset inside x1

    // ---- weight DMA role: round s moves pieces (s*256 + wave*64 .. + 63) % WPC of the chunk's image [COT][piece][lane]; global dword
    //      of piece q of chunk ch: (ch * NS + ctile*COT) * NP*256 + q*4
    const unsigned* wbase = reinterpret_cast<const unsigned*>(NP == 2 ? a.wph : a.wpb) + PX::HDR + (long)ctile * COT * (NP * 256);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int w_goff[MAXW], w_lds[MAXW];
#pragma unroll
    for (int s = 0; s < MAXW; ++s) {
        const int q0 = (s * 256 + wave_u * 64) % WPC;
        w_lds[s] = q0 * 4;                    // dword offset of the wave's 64 pieces inside the chunk image
        w_goff[s] = (q0 + lane) * 4;
    }

    // ---- coefficient table (PRO): (A, B) of every input channel for the images of this tile, fetched once -- by LDS-DMA, 16 bytes (two
    // channels) per lane, IN FRONT of the first chunk: these DMAs are the oldest VMEM operations of the wave, so the table has landed
    // when chunk 0 has (in-order return) and the barrier of the first stage makes it visible.  (Round 3's first form loaded the table
    // through registers and waited for it before the first chunk was requested: one exposed memory latency per workgroup.)
    const int b_first = (int)(gp0 / HW);
    const int nimg = HW >= PT ? 1 : PT / HW;
    if (PRO != 0) {
        const int ppi = Cin >> 1;                                  // 16-byte pieces per image (Cin % 16 == 0)
        for (int q0 = 0; q0 < nimg * ppi; q0 += 256) {
            const int q = q0 + tid;
            if (q < nimg * ppi) {
                const int im = q / ppi, w = q - im * ppi;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(a.coef + ((long)min(b_first + im, a.B - 1) * Cin) * 2 + w * 4),
                    (__attribute__((address_space(3))) void*)(sC + (q0 + wave * 64) * 4), 16, 0, 0);
            }
        }
    }
    const int my_img = HW >= PT ? 0 : (wave * 32) / HW;            // image (within the tile) of this wave's 32 pixels
    const int x_lane = wave * 32 + l31 + half * PT;                // the lane's pixel, row `half` of the chunk

#define Q1_DMA(ch)                                                                                              \
    {                                                                                                           \
        const int cb = (ch) * CK;                                                                               \
        const unsigned* wsrc = wbase + (long)(ch) * NS * (NP * 256);                                                  \
        unsigned* wdst = sW + ((ch) & (NB - 1)) * WDW;                                                          \
        _Pragma("unroll") for (int s = 0; s < MAXW; ++s)                                                        \
            __builtin_amdgcn_global_load_lds(                                                                   \
                (const __attribute__((address_space(1))) void*)(wsrc + w_goff[s]),                              \
                (__attribute__((address_space(3))) void*)(wdst + w_lds[s]), 16, 0, 0);                          \
        if constexpr (!IM) {                                                                                    \
            const bool second = cb >= a.C0;                                                                     \
            const float* xsrc = second ? a.x1 + (long)(cb - a.C0) * HW : a.x0 + (long)cb * HW;                  \
            const int voff = second ? voff1 : voff0;                                                            \
            float* xdst = sX + ((ch) & (NB - 1)) * XSZ;                                                         \
            _Pragma("unroll") for (int s = 0; s < MAXX; ++s)                                                    \
                __builtin_amdgcn_global_load_lds(                                                               \
                    (const __attribute__((address_space(1))) void*)(xsrc + (long)s * RPS * HW + voff),          \
                    (__attribute__((address_space(3))) void*)(xdst + (s * 256 + wave * 64) * 4), 16, 0, 0);     \
        }                                                                                                       \
    }

    f32x16 acc[COT];
#pragma unroll
    for (int ct = 0; ct < COT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;

    // diagnostics (mcvd_ctx_set_debug_buffer; tests/gpu_diag.py w2htl): wave 0's cycle counts per phase, wall-clock start / end, CU
    const bool rec = a.dbg != nullptr && wave == 0;
    unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, rt0 = 0;
    if (rec) { rt0 = __builtin_amdgcn_s_memrealtime(); tk0 = __builtin_amdgcn_s_memtime(); }

    // K loop: NB = 4 LDS buffers, the DMA of chunk k + 3 is issued while chunk k is staged (prefetch distance 3 chunks).  Every wave
    // issues exactly G DMA instructions per chunk, in order, and nothing else: chunk k has landed when at most the groups issued after
    // it (chunks k+1, k+2) are outstanding.  The barrier behind the wait makes every wave's part visible and certifies that every wave
    // has READ chunk k - 1 (its LDS reads were waited for: lgkmcnt(0)), whose buffer the next DMA overwrites.
    // The loop is software-pipelined in registers: while the MFMAs of chunk k run (B pieces split one iteration earlier), chunk k + 1 is
    // staged -- its pixel / coefficient reads are issued first, its affine / SiLU / split (the other B register set) is independent of
    // the MFMAs and the compiler interleaves the two.  Round 2's loop did LDS reads -> VALU -> MFMAs of ONE chunk in sequence between
    // two barriers: 2.3 k cycles per chunk with 0.58 k of matrix work per wave (profiles/r03_timeline_wino3_conv1x1_b3.txt).
    // What a chunk costs (profiles/r04_conv1x1_h2_ablation.txt, cycles per chunk at two workgroups per CU, 1152 of them MFMA time of the
    // two waves of a SIMD): the times of the parts ADD rather than overlap -- no MFMAs -51 %, no split -30 %, no DMA -17 % -- and a wave
    // issues one VALU instruction per ~7.6 cycles (profiles/r03_ubench_mfma_valu.txt), so every VALU instruction of the staging counts:
    // the eight pixel reads differ by immediates (two per ds_read2st64_b32), the affine is eight v_fma_f32 in asm (the vectoriser's
    // v_pk_fma_f32 cost three v_mov per pair), the barrier is the builtin (behind an asm wait the compiler's lgkmcnt bookkeeping held
    // the first MFMA of every chunk for the pixel reads): 90 -> 52 VALU per chunk, 1.88 k -> 1.68 k cycles.  Tried and dropped: two A
    // register sets with the reads behind the barrier (+5 %: the scheduler chains the MFMAs per accumulator), s_setprio around the MFMAs
    // (+3 %), three instead of two workgroups per CU (round 3: no gain).
    const int nchunks = Cin / CK;          // Cin % CK == 0 (launch check): the zero rows that pad the weights are never staged
    Q1_DMA(0);
    if (nchunks > 1) Q1_DMA(1);
    if (nchunks > 2) Q1_DMA(2);
    // IM: the raw patch of this pixel tile and the K-row -> patch-offset table, through registers, BEHIND the first weight DMAs so that the two
    // latencies overlap (the compiler waits for the patch loads before it parks them; they are younger than the DMAs, so that wait covers
    // the DMAs too and the counted waits of the first stages are trivially met).  The barrier of the first stage makes both visible.
    // Patch layout [Cc + 1 planes][PR = PT / W + 2 rows][PITCH = W + 8]: the W interior pixels of a row at columns 4 .. W + 3 (16-byte aligned:
    // float4 in, ds_write_b128 out), zero blocks at 0 .. 3 and W + 4 .. W + 7 -- a pixel tile is whole image rows, so the columns left and
    // right of the interior are ALWAYS outside the image; only the first / last patch row can be.  Plane Cc is all zeros (padding K rows).
    float* sPatch = sX;
    const int im_W = a.W, im_PR = PT / (IM ? a.W : PT) + 2, im_PITCH = a.W + 8, im_Cc = a.C0 + a.C1;
    const int im_plane = im_PR * im_PITCH;
    int* sTab = reinterpret_cast<int*>(sPatch + (im_Cc + 1) * im_plane);       // [chunk][half][8]: K row 16 k + 2 e + h at [k][h][e]
    int im_base = 0;
    if constexpr (IM) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int y0 = (int)((gp0 - (long)b_first * HW) / im_W);   // first image row of the tile (HW % PT == 0, PT % W == 0: whole rows of one image)
        const int w4 = im_W >> 2, lw4 = 31 - __builtin_clz(w4);    // W / 4 is a power of two (launch check)
        const int rows = (im_Cc + 1) * im_PR;
        for (int u0 = 0; u0 < rows * w4; u0 += 2 * 256) {          // interior: one float4 per unit, two units per thread in flight
            f4 v[2];
            int dst[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int u = u0 + j * 256 + tid;
                const int row = u >> lw4, q = u & (w4 - 1);
                const int c = row / im_PR, ry = row - c * im_PR;
                const int yy = y0 - 1 + ry;
                v[j] = f4{0.0f, 0.0f, 0.0f, 0.0f};
                dst[j] = u < rows * w4 ? row * im_PITCH + 4 + 4 * q : -1;
                if (dst[j] >= 0 && c < im_Cc && yy >= 0 && yy < a.H)
                    v[j] = *reinterpret_cast<const f4*>(c < a.C0 ? a.x0 + (((long)b_first * a.C0 + c) * a.H + yy) * im_W + 4 * q
                                                                 : a.x1 + (((long)b_first * a.C1 + (c - a.C0)) * a.H + yy) * im_W + 4 * q);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (dst[j] >= 0) *reinterpret_cast<f4*>(sPatch + dst[j]) = v[j];
        }
        for (int u = tid; u < 2 * rows; u += 256)                   // the two zero blocks of every row
            *reinterpret_cast<f4*>(sPatch + (u >> 1) * im_PITCH + (u & 1) * (im_W + 4)) = f4{0.0f, 0.0f, 0.0f, 0.0f};
        for (int kk = tid; kk < Cin; kk += 256) {
            const int c = kk / 9, t = kk - 9 * c;
            const int off = c < im_Cc ? c * im_plane + (t / 3) * im_PITCH + (t % 3) + 3 : im_Cc * im_plane + 4;
            sTab[(kk >> 4) * 16 + (kk & 1) * 8 + ((kk & 15) >> 1)] = off;
        }
        const int p = wave * 32 + l31;                              // the lane's pixel of the tile
        im_base = (p / im_W) * im_PITCH + (p % im_W);
    }
    if (rec) tk1 = __builtin_amdgcn_s_memtime();
    /* chunk k becomes visible, chunk k + 3 is requested, the LDS reads of chunk k are issued: 8 pixel values, their coefficients, the
       NP * COT A operands */
#define Q1_STAGE(k, BV, CF)                                                                                     \
    {                                                                                                           \
        const int after = nchunks - 1 - (k);                                                                    \
        if (after >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * G) : "memory");                           \
        else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(G) : "memory");                          \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
        if (!(EXP & 8)) {      /* builtins, not asm: the compiler's own lgkmcnt bookkeeping must see that the LDS reads have drained */ \
            asm volatile("" ::: "memory");                                                                      \
            __builtin_amdgcn_s_waitcnt(0xC07F);                 /* lgkmcnt(0) */                                 \
            __builtin_amdgcn_s_barrier();                                                                       \
            asm volatile("" ::: "memory");                                                                      \
        }                                                                                                       \
        if ((k) + 3 < nchunks && !(EXP & 1)) Q1_DMA((k) + 3);                                                   \
        const float* sXc = sX + ((k) & (NB - 1)) * XSZ + x_lane;       /* (the lane's half is in the base: the eight reads differ by immediates) */ \
        const float* sCb = sC + ((long)my_img * Cin + (k) * CK) * 2 + half * 2;                                 \
        if constexpr (IM) {     /* K rows 16 k + 2 e + h of lane (n, h): their patch offsets from the table, then the gather */ \
            const u32x4* tp = reinterpret_cast<const u32x4*>(sTab + (k) * 16 + half * 8);                       \
            const u32x4 t0 = tp[0], t1 = tp[1];                                                                 \
            const float* pb = sPatch + im_base;                                                                 \
            BV[0] = pb[t0[0]]; BV[1] = pb[t0[1]]; BV[2] = pb[t0[2]]; BV[3] = pb[t0[3]];                         \
            BV[4] = pb[t1[0]]; BV[5] = pb[t1[1]]; BV[6] = pb[t1[2]]; BV[7] = pb[t1[3]];                         \
        } else {                                                                                                \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {         /* element e of lane (n, h) = channel 2e + h of the chunk */ \
            BV[e] = sXc[2 * e * PT];                                                                            \
            if (PRO != 0) CF[e] = *reinterpret_cast<const f32x2*>(sCb + 4 * e);                                 \
        }                                                                                                       \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    }
    /* the NP * COT A operands of chunk k (visible since its Q1_STAGE): ONE register set -- read behind the MFMAs of the previous chunk,
       the latency passes under the split of this one (two sets cost 36 registers more than the kernel has at COT = 3) */
#define Q1_READ_AW(k, AW)                                                                                       \
    {                                                                                                           \
        const u32x4* sWc = reinterpret_cast<const u32x4*>(sW + ((k) & (NB - 1)) * WDW) + lane;                  \
        _Pragma("unroll") for (int ct = 0; ct < COT; ++ct)                                                      \
            _Pragma("unroll") for (int p = 0; p < NP; ++p) AW[ct][p] = sWc[(ct * NP + p) * 64];                 \
    }
    /* GroupNorm affine (+ SiLU), split into pieces: the B operand of the chunk */
#define Q1_PREP(BV, CF, BP)                                                                                     \
    if (EXP & 4) { _Pragma("unroll") for (int p = 0; p < NP; ++p) _Pragma("unroll") for (int j = 0; j < 4; ++j) BP[p][j] = __builtin_bit_cast(unsigned, BV[2 * j]); } else \
    {                                                                                                           \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                         \
            if (PRO != 0) {     /* (asm: the vectoriser pairs these into v_pk_fma_f32 and pays three v_mov per pair to line the operands up) */ \
                asm("v_fma_f32 %0, %1, %2, %3" : "=v"(BV[e]) : "v"(BV[e]), "v"(CF[e].x), "v"(CF[e].y));         \
                if (PRO == 2) BV[e] = silu_q(BV[e]);                                                            \
            }                                                                                                   \
            if (NP == 2) BV[e] *= PX::ACT_SCALE;                                                                \
        }                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            unsigned w[NP];                                                                                     \
            PX::template split<false>(BV[2 * j], BV[2 * j + 1], w);                                             \
            _Pragma("unroll") for (int p = 0; p < NP; ++p) BP[p][j] = w[p];                                     \
        }                                                                                                       \
    }
    /* product-major MFMA order, smallest product first: consecutive MFMAs write different accumulators */
#define Q1_MMA(AW, BP)                                                                                          \
    if (EXP & 2) { _Pragma("unroll") for (int ct = 0; ct < COT; ++ct) acc[ct][0] += __builtin_bit_cast(float, AW[ct][0][0] ^ BP[0][1] ^ AW[ct][NP - 1][2] ^ BP[NP - 1][3]); } else \
    {                                                                                                           \
        _Pragma("unroll") for (int k = 0; k < PX::NPROD; ++k)                                                   \
            _Pragma("unroll") for (int ct = 0; ct < COT; ++ct) acc[ct] = PX::mfma(AW[ct][PX::PA(k)], BP[PX::PB(k)], acc[ct]); \
    }
    /* one pipeline step (k + 1 < nchunks): chunk k + 1 staged, then -- ONE basic block, no fences, so that the scheduler interleaves them
       -- the MFMAs of chunk k (pieces BPC) with the split of chunk k + 1 (-> BPN), then the A operands of chunk k + 1 */
#define Q1_STEP(k, BPC, BPN)                                                                                    \
    {                                                                                                           \
        Q1_STAGE((k) + 1, bv, cf)                                                                               \
        Q1_MMA(aw, BPC)                                                                                         \
        Q1_PREP(bv, cf, BPN)                                                                                    \
        /* the split ends HERE (left alone the scheduler sinks every other chunk's split below the next barrier, where nothing overlaps it) */ \
        if (!(EXP & 16)) {                                                                                      \
            /* three MFMAs lead (the wait for the pixel reads comes behind them), the others follow a share of the split each */ \
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                  \
            _Pragma("unroll") for (int i_ = 3; i_ < PX::NPROD * COT; ++i_) {                                    \
                __builtin_amdgcn_sched_group_barrier(0x002, NVS, 0);                                            \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
            }                                                                                                   \
            _Pragma("unroll") for (int p_ = 0; p_ < NP; ++p_) asm volatile("" : "+v"(BPN[p_]));                 \
        }                                                                                                       \
        Q1_READ_AW((k) + 1, aw)                                                                                 \
    }
    constexpr int NVS = (PRO == 2 ? 120 : PRO == 1 ? 76 : 64) / (PX::NPROD * COT > 3 ? PX::NPROD * COT - 3 : 1) + 1;      // VALU instructions per MFMA
    {
        float bv[8];
        f32x2 cf[8];
        u32x4 aw[COT][NP], bpA[NP], bpB[NP];
        Q1_STAGE(0, bv, cf)
        Q1_READ_AW(0, aw)
        Q1_PREP(bv, cf, bpA)
        int ch = 0;                        // invariant at the top: aw and bpA belong to chunk ch
        for (; ch + 2 < nchunks; ch += 2) {
            Q1_STEP(ch, bpA, bpB)
            Q1_STEP(ch + 1, bpB, bpA)
        }
        if (ch + 1 < nchunks) {
            Q1_STEP(ch, bpA, bpB)
            Q1_MMA(aw, bpB)
        } else {
            Q1_MMA(aw, bpA)
        }
    }
#undef Q1_STEP
#undef Q1_STAGE
#undef Q1_READ_AW
#undef Q1_PREP
#undef Q1_MMA
#undef Q1_DMA
    if (rec) tk2 = __builtin_amdgcn_s_memtime();

    // ---- epilogue.  Lane (l31, half) holds pixel 32*wave + l31 of the tile for couts ct*32 + (r&3) + 8*(r>>2) + 4*half: a store from
    // there is 64 scattered dwords per instruction, 16 * COT instructions per wave, and the kernel spent 40 % of its time issuing them
    // (profiles/r02_conv1x1_h2_timeline.txt).  The tile goes through the LDS instead ([cout][128 pixels], the K loop's buffers are
    // free) and leaves as global_store_dwordx4: four consecutive pixels per lane, 1 KiB contiguous per wave instruction.
    const float inv = NP == 2 ? a.wph[2] * (1.0f / PX::ACT_SCALE) : 1.0f;      // NP = 2: 1 / (weight scale * activation scale), a power of two
    float* sO = smem;                                         // [BCO][OPT]
    constexpr int OPT = PT + 4;         // row pitch of the transposed tile: 132 = 4 (mod 64) dwords, so that the 16-byte reads of 16 DIFFERENT rows at
                                        // one column (the V piece images below: a lane per channel) fall on 64 different banks; rows read along
                                        // the pixels (the fp32 stores, the K images) are conflict-free at any pitch
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int px4 = tid & 31, cr = tid >> 5;                  // store role: 4 pixels px4*4 .. +3 of cout rows cr + 8*k
    const long gp = gp0 + px4 * 4;
    const bool valid = gp < NPX;                              // NPX % 4 == 0: a group of four is valid or invalid as a whole
    const long gpc = valid ? gp : 0;
    const int ob = (int)(gpc / HW), op = (int)(gpc - (long)ob * HW);          // HW % 4 == 0: the four pixels are in one image
    const long obase = (long)ob * a.Cout * HW + op;
    // bias and residual of every row the thread stores, requested before the tile is transposed: fetched inside the store loop, each
    // of its BCO / 8 rounds waited for a memory latency AND (vmcnt counts stores too) for the previous round's store to retire --
    // 8.5 k cycles for a 48 KB tile (profiles/r03_timeline_wino3_conv1x1_b3.txt)
    float e_bias[BCO / 8];
    f32x4 e_res[BCO / 8];
#pragma unroll
    for (int k = 0; k < BCO / 8; ++k) {
        const int co = co0 + cr + 8 * k;
        e_bias[k] = a.bias[co];                               // zero-padded to CoutP
        e_res[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (a.res) e_res[k] = *reinterpret_cast<const f32x4*>(a.res + obase + (long)min(co, a.Cout - 1) * HW);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave is done reading the chunk buffers
#pragma unroll
    for (int ct = 0; ct < COT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            sO[(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * OPT + wave * 32 + l31] = acc[ct][r] * inv;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    bool kv_tile[COT];
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) kv_tile[ct] = false;
    if constexpr (KV) {
        // sub-tile ct = couts co0 + 32 ct .. + 31 of [q | k | v] x [head][D]: K and V sub-tiles are written as piece images
        const int Cq = a.kv_C, D = a.kv_D, DT = D >> 5, heads = Cq / D, ntl = HW >> 5;
        unsigned* img = reinterpret_cast<unsigned*>(a.kv_img);
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) {
            const int cs = co0 + ct * 32;
            const int which = cs / Cq;                        // 0 q, 1 k, 2 v (>= 3: padding)
            if (which != 1 && which != 2) continue;
            kv_tile[ct] = true;
            const int within = cs - which * Cq, head = within / D, ch0 = within - head * D;       // ch0: multiple of 32
#pragma unroll
            for (int rnd = 0; rnd < 2; ++rnd) {
                const int item = rnd * 256 + tid;
                if (which == 1) {
                    // K: item = (row pair rp = (st_l, j, h), 4 pixels): channels c, c + 2 with c = ch0 + 16 st_l + 4 j + h
                    const int rp = item >> 5, p4 = item & 31;
                    const int st_l = rp >> 3, j = (rp >> 1) & 3, h = rp & 1;
                    const int rl = ct * 32 + 16 * st_l + 4 * j + h;                                 // row of sO
                    const long g = gp0 + p4 * 4;
                    if (g < NPX) {
                        const int bb = (int)(g / HW), pix = (int)(g - (long)bb * HW);
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sO + rl * OPT + p4 * 4) + a.bias[co0 + rl];
                        const f32x4 v1 = *reinterpret_cast<const f32x4*>(sO + (rl + 2) * OPT + p4 * 4) + a.bias[co0 + rl + 2];
                        unsigned w[4][3];
#pragma unroll
                        for (int i4 = 0; i4 < 4; ++i4) Pieces<3>::template split<false>(v0[i4], v1[i4], w[i4]);
                        const int st = (ch0 >> 4) + st_l;
                        unsigned* d1 = img + ((long)(bb * heads + head) * ntl + (pix >> 5)) * (3072 * DT) + (st * 3 * 4 + j) * 64 + h * 32 + (pix & 31);
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4*>(d1 + pc * 256) = u32x4{w[0][pc], w[1][pc], w[2][pc], w[3][pc]};
                    }
                } else {
                    // V: item = (channel m of the sub-tile, key tile tt of the pixel tile, step s2, half h): keys 16 s2 + 4 h + {0..3} and + 8
                    const int m = item & 31, grp = item >> 5;
                    const int tt = grp >> 2, s2 = (grp >> 1) & 1, h = grp & 1;
                    const int kb = 16 * s2 + 4 * h;
                    const long g = gp0 + tt * 32;
                    if (g < NPX) {
                        const int bb = (int)(g / HW), pix = (int)(g - (long)bb * HW);
                        const int rl = ct * 32 + m;
                        const float bs = a.bias[co0 + rl];
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sO + rl * OPT + tt * 32 + kb) + bs;
                        const f32x4 v1 = *reinterpret_cast<const f32x4*>(sO + rl * OPT + tt * 32 + kb + 8) + bs;
                        unsigned wa[3], wb[3], wc[3], wd[3];
                        Pieces<3>::template split<false>(v0[0], v0[1], wa);
                        Pieces<3>::template split<false>(v0[2], v0[3], wb);
                        Pieces<3>::template split<false>(v1[0], v1[1], wc);
                        Pieces<3>::template split<false>(v1[2], v1[3], wd);
                        unsigned* d1 = img + ((long)(bb * heads + head) * ntl + (pix >> 5)) * (3072 * DT) + 1536 * DT +
                                       ((((ch0 >> 5) * 2 + s2) * 3) * 64 + h * 32 + m) * 4;
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4*>(d1 + pc * 256) = u32x4{wa[pc], wb[pc], wc[pc], wd[pc]};
                    }
                }
            }
        }
    }
    {
#pragma unroll
        for (int k = 0; k < BCO / 8; ++k) {
            if (KV && kv_tile[k >> 2]) continue;              // rows cr + 8 k of sub-tile k / 4: written as pieces above
            const int cl = cr + 8 * k, co = co0 + cl;
            f32x4 v = *reinterpret_cast<const f32x4*>(sO + cl * OPT + px4 * 4);
            v = (v + e_bias[k] + e_res[k]) * a.out_scale;
            if (valid && co < a.Cout) *reinterpret_cast<f32x4*>(a.y + obase + (long)co * HW) = v;
            if (a.stats && (HW >= PT || HW == 64)) {
                // GroupNorm partials of the FINAL values (ConvArgs::stats): the 32 lanes of a half-wave hold the 128 pixels of this cout
                // row -- one image's pixel block (HW >= 128: partial index = block of the image), or two 8x8 images of 16 lanes = one
                // DPP row each.  Pilot-shifted moments as in conv_wino2h.cpp: plain sums merge with two DPP adds per level.
                const bool g16 = HW == 64;
                float pil;
                {
                    const int pv = __builtin_bit_cast(int, v[0]);
                    const int s0 = __builtin_amdgcn_readlane(pv, 0), s1 = __builtin_amdgcn_readlane(pv, 16);
                    const int s2 = __builtin_amdgcn_readlane(pv, 32), s3 = __builtin_amdgcn_readlane(pv, 48);
                    pil = __builtin_bit_cast(float, (lane & 32) ? ((g16 && (lane & 16)) ? s3 : s2) : ((g16 && (lane & 16)) ? s1 : s0));
                }
                const float d0 = v[0] - pil, d1 = v[1] - pil, d2 = v[2] - pil, d3 = v[3] - pil;
                float sm = (d0 + d1) + (d2 + d3);
                float qm = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#define Q1_MERGE(CTRL, ROWMASK)                                                                                     \
                {                                                                                                   \
                    sm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sm), CTRL, ROWMASK, 0xf, false)); \
                    qm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, qm), CTRL, ROWMASK, 0xf, false)); \
                }
                Q1_MERGE(0xB1, 0xf)                   // quad_perm [1,0,3,2]
                Q1_MERGE(0x4E, 0xf)                   // quad_perm [2,3,0,1]
                Q1_MERGE(0x124, 0xf)                  // row_ror:4
                Q1_MERGE(0x128, 0xf)                  // row_ror:8: every lane of a row of 16 holds the row's totals
                float sm2 = sm, qm2 = qm;             // row_bcast:15: lanes 16-31 / 48-63 add the totals of the row below (32-lane groups)
                sm2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sm), 0x142, 0xa, 0xf, false));
                qm2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, qm), 0x142, 0xa, 0xf, false));
#undef Q1_MERGE
                if (!g16) { sm = sm2; qm = qm2; }
                const bool writer = g16 ? (px4 & 15) == 0 : px4 == 31;
                if (writer && valid && co < a.Cout) {
                    const float npix = g16 ? 64.0f : 128.0f;
                    const int np = g16 ? 1 : HW >> 7, pidx = g16 ? 0 : op >> 7;
                    float* q = a.stats + (((long)ob * a.Cout + co) * np + pidx) * 2;
                    q[0] = sm + npix * pil;
                    q[1] = fmaxf(qm - sm * sm / npix, 0.0f);
                }
            }
        }
    }
    if (rec) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* d = a.dbg + (long)blockIdx.x * 8;
            d[0] = tk1 - tk0; d[1] = tk2 - tk1; d[2] = rt0; d[3] = __builtin_amdgcn_s_memrealtime();
            d[4] = ((unsigned long long)xcc << 32) | hwid;
            d[5] = now - tk2; d[6] = (unsigned long long)nchunks; d[7] = now - tk0;
        }
    }
}

// im_cc > 0: the im2col form (ConvArgs::im2col) -- no pixel ring; the raw patch [im_cc + 1][PT / W + 2][W + 8] and the offset table [Cin] instead
static size_t q1_lds_bytes(int np, int cot, int Cin, int HW, int im_cc = 0, int W = 0) {
    const int nimg = HW >= Q1_PT ? 1 : Q1_PT / HW;
    const size_t pix = im_cc > 0 ? (size_t)((im_cc + 1) * (Q1_PT / W + 2) * (W + 8) + Cin) : (size_t)Q1_NB * Q1_CK * Q1_PT + (size_t)nimg * Cin * 2;
    const size_t loop = ((size_t)Q1_NB * (cot * np * 256) + pix) * sizeof(float);
    const size_t epi = (size_t)(32 * cot) * (Q1_PT + 4) * sizeof(float);      // the transposed output tile (row pitch PT + 4) overlays the chunk buffers
    return loop > epi ? loop : epi;
}

// Shape ids 15 (np = 3) / 14 (np = 2) apply to this launch (cot = cout tile in 32-channel units, 1..4).
bool conv1x1_h2_supported(const ConvArgs& a, int cot, int np) {
    const int HW = a.H * a.W;
    if (np != 2 && np != 3) return false;
    if (a.ks != 1 || !(np == 2 ? a.wph : a.wpb) || HW % 32 != 0 || cot < 1 || cot > 4 || (a.CoutP / 32) % cot != 0) return false;
    if (a.im2col) {          // the stem as a GEMM over an im2col the kernel stages itself: whole image rows per pixel tile, raw input, three pieces
        return np == 3 && !a.coef && !a.act && !a.kv_img && a.W >= 8 && a.W <= Q1_PT && (a.W & (a.W - 1)) == 0 && HW % Q1_PT == 0 && a.Cin % Q1_CK == 0 &&
               a.Cin == a.CinP && a.Cin >= 9 * (a.C0 + a.C1) && (long)a.B * (a.C0 > a.C1 ? a.C0 : a.C1) * HW < (1L << 31) &&
               q1_lds_bytes(np, cot, a.Cin, HW, a.C0 + a.C1, a.W) <= 80 * 1024;
    }
    if (!(HW % Q1_PT == 0 || (HW < Q1_PT && Q1_PT % HW == 0 && Q1_PT / HW <= Q1_MAXIMG))) return false;
    if (a.Cin % Q1_CK != 0 || a.CinP % Q1_CK != 0) return false;            // no partial chunk: every staged row is real data
    if (a.C1 > 0 && a.C0 % Q1_CK != 0) return false;                        // a chunk never straddles the concat seam
    if ((long)a.B * (a.C0 > a.C1 ? a.C0 : a.C1) * HW >= (1L << 31)) return false;   // 32-bit lane offsets
    if (a.act && !a.coef) return false;
    return q1_lds_bytes(np, cot, a.Cin, HW) <= 80 * 1024;                   // two workgroups per CU
}

template <int NP, int COT>
static int q1_launch(const ConvArgs& a, hipStream_t s) {
    const int HW = a.H * a.W;
    const long NPX = (long)a.B * HW;
    const int ptiles = (int)((NPX + Q1_PT - 1) / Q1_PT);
    const int nct = a.CoutP / (32 * COT);
    const size_t lds = q1_lds_bytes(NP, COT, a.Cin, HW, a.im2col ? a.C0 + a.C1 : 0, a.W);
    const dim3 grid(((ptiles + 7) / 8) * 8 * nct);
    if constexpr (NP == 3) {
        if (a.im2col) {
            static PerDeviceOnce raised_im;
            if (lds > 48 * 1024 && raised_im.first_use()) {
                MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<3, COT, 0, 0, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                raised_im.done();
            }
            hipLaunchKernelGGL((conv1x1_h2_kernel<3, COT, 0, 0, false, true>), grid, dim3(256), lds, s, a, ptiles, nct);
            MCVD_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    static PerDeviceOnce raised;
    if (lds > 48 * 1024 && raised.first_use()) {
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<NP, COT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<NP, COT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<NP, COT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if constexpr (NP == 3)
            MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<3, COT, 1, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised.done();
    }
#ifdef MCVD_DIAG
    {   // diagnostics build only: timing-only ablations (WRONG RESULTS), tests/gpu_diag.py w2htl with MCVD_Q1_EXP
        const char* es = getenv("MCVD_Q1_EXP");
        const int e = es ? atoi(es) : 0;
        if (e != 0 && NP == 3 && COT == 3) {
            static PerDeviceOnce raised2;
            if (raised2.first_use()) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<3, 3, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<3, 3, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<3, 3, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<3, 3, 1, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<3, 3, 1, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<3, 3, 1, 15>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<3, 3, 1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                raised2.done();
            }
            switch (e) {      // (all with the affine prologue instantiation: the caller passes coef)
                case 1: hipLaunchKernelGGL((conv1x1_h2_kernel<3, 3, 1, 1>), grid, dim3(256), lds, s, a, ptiles, nct); break;
                case 2: hipLaunchKernelGGL((conv1x1_h2_kernel<3, 3, 1, 2>), grid, dim3(256), lds, s, a, ptiles, nct); break;
                case 4: hipLaunchKernelGGL((conv1x1_h2_kernel<3, 3, 1, 4>), grid, dim3(256), lds, s, a, ptiles, nct); break;
                case 6: hipLaunchKernelGGL((conv1x1_h2_kernel<3, 3, 1, 6>), grid, dim3(256), lds, s, a, ptiles, nct); break;
                case 7: hipLaunchKernelGGL((conv1x1_h2_kernel<3, 3, 1, 7>), grid, dim3(256), lds, s, a, ptiles, nct); break;
                case 16: hipLaunchKernelGGL((conv1x1_h2_kernel<3, 3, 1, 16>), grid, dim3(256), lds, s, a, ptiles, nct); break;
                default: hipLaunchKernelGGL((conv1x1_h2_kernel<3, 3, 1, 15>), grid, dim3(256), lds, s, a, ptiles, nct); break;
            }
            MCVD_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    if constexpr (NP == 3) {     // diagnostics build only: MCVD_Q1_NOPRIO=1 runs the loop of EXP bit 4 (right results) for the A/B
        static const bool noprio = getenv("MCVD_Q1_NOPRIO") && atoi(getenv("MCVD_Q1_NOPRIO")) != 0;
        if (noprio) {
            static PerDeviceOnce raised3;
            if (raised3.first_use()) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<NP, COT, 0, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<NP, COT, 1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_h2_kernel<NP, COT, 2, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                raised3.done();
            }
            if (!a.coef) hipLaunchKernelGGL((conv1x1_h2_kernel<NP, COT, 0, 16>), grid, dim3(256), lds, s, a, ptiles, nct);
            else if (!a.act) hipLaunchKernelGGL((conv1x1_h2_kernel<NP, COT, 1, 16>), grid, dim3(256), lds, s, a, ptiles, nct);
            else hipLaunchKernelGGL((conv1x1_h2_kernel<NP, COT, 2, 16>), grid, dim3(256), lds, s, a, ptiles, nct);
            MCVD_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
#endif
    if constexpr (NP == 3) {
        if (a.kv_img && a.coef && !a.act) {            // the q|k|v projection with its K and V thirds as piece images (conv1x1_h2_kv_supported)
            hipLaunchKernelGGL((conv1x1_h2_kernel<3, COT, 1, 0, true>), grid, dim3(256), lds, s, a, ptiles, nct);
            MCVD_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    if (!a.coef)
        hipLaunchKernelGGL((conv1x1_h2_kernel<NP, COT, 0>), grid, dim3(256), lds, s, a, ptiles, nct);
    else if (!a.act)
        hipLaunchKernelGGL((conv1x1_h2_kernel<NP, COT, 1>), grid, dim3(256), lds, s, a, ptiles, nct);
    else
        hipLaunchKernelGGL((conv1x1_h2_kernel<NP, COT, 2>), grid, dim3(256), lds, s, a, ptiles, nct);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int NP>
static int q1_launch_cot(const ConvArgs& a, int cot, hipStream_t s) {
    switch (cot) {
        case 1: return q1_launch<NP, 1>(a, s);
        case 2: return q1_launch<NP, 2>(a, s);
        case 3: return q1_launch<NP, 3>(a, s);
        default: return q1_launch<NP, 4>(a, s);
    }
}

// The K / V piece-image epilogue applies: the fused q|k|v projection (Cout = 3 C, affine prologue, no activation / residual / scale) of an
// attention block whose head dim the pre-split attention kernel serves (attention_h2.cpp: attn_h2p_supported)
bool conv1x1_h2_kv_supported(const ConvArgs& a, int cot) {
    return a.kv_img && a.kv_C > 0 && a.Cout == 3 * a.kv_C && a.kv_D % 32 == 0 && a.kv_C % a.kv_D == 0 && a.coef && !a.act && !a.res && a.out_scale == 1.0f &&
           !a.stats && conv1x1_h2_supported(a, cot, 3);
}

int launch_conv1x1_h2(const ConvArgs& a, int cot, hipStream_t s, int np) {
    MCVD_REQUIRE(!a.kv_img || (np == 3 && conv1x1_h2_kv_supported(a, cot)), "conv1x1 split-operand GEMM: K / V piece images requested for a launch that cannot write them");
    MCVD_REQUIRE(conv1x1_h2_supported(a, cot, np), "conv1x1 split-operand GEMM: unsupported (np=%d ks=%d H=%d W=%d Cin=%d C0=%d cot=%d, weight pieces %s)",
                 np, a.ks, a.H, a.W, a.Cin, a.C0, cot, (np == 2 ? a.wph : a.wpb) ? "present" : "missing");
    const int rc = np == 2 ? q1_launch_cot<2>(a, cot, s) : q1_launch_cot<3>(a, cot, s);
    const int HW = a.H * a.W;
    if (rc == 0 && a.stats && (HW >= Q1_PT || HW == 64)) set_last_conv_stats_np(HW >= Q1_PT ? HW / Q1_PT : 1);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight pieces from the packed fp32 matrix wp[ci * CoutP + co] (launch_pack_conv_weight: every fused weight and the zero padding
// are already in place).  Halfword index of (ci, co, piece):
//   (((ci/16) * NS + co/32) * NP + piece) * 512 + (lane*4 + j) * 2 + (el & 1),   cc = ci % 16 = 2 el + h, j = el >> 1, lane = h*32 + co%32
// np = 2: wh = [4 header floats][CinP * CoutP dwords]; header: max |w|, scale = 2^e (max |w| * 2^e in [2^13, 2^14)), 1 / scale;
//         pieces h1 = fp16(w * scale), h2 = fp16(w * scale - h1).
// np = 3: wh = [CinP * CoutP * 3 halfwords], no header, no scale; pieces b1 = bf16(w), b2 = bf16(w - b1), b3 = w - b1 - b2 (exact).
__global__ void q1_absmax_kernel(const float* w, long n, unsigned* hdr) {
    float m = 0.0f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(hdr, __float_as_uint(m));
}

__global__ void pack_conv1x1_h2_kernel(const float* wp, float* wh, int CinP, int CoutP) {
    const float wmax = wh[0];
    int k = 0;
    (void)frexpf(wmax, &k);                                     // wmax = m * 2^k, 0.5 <= m < 1
    const int e = (wmax > 0.0f && wmax < 3.0e38f) ? min(max(14 - k, -60), 60) : 0;
    const float scale = ldexpf(1.0f, e);
    if (blockIdx.x == 0 && threadIdx.x == 0) { wh[1] = scale; wh[2] = ldexpf(1.0f, -e); }
    _Float16* dst = reinterpret_cast<_Float16*>(wh + Pieces<2>::HDR);
    const long n = (long)CinP * CoutP;
    const int NS = CoutP / 32;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % CoutP), ci = (int)(i / CoutP);
        const float us = wp[i] * scale;                         // exact (power of two)
        const _Float16 h1 = (_Float16)us;
        const _Float16 h2 = (_Float16)(us - (float)h1);
        const int cc = ci & 15, h = cc & 1, el = cc >> 1, lane = h * 32 + (co & 31);
        const long o = (((long)(ci >> 4) * NS + (co >> 5)) * 2) * 512 + (lane * 4 + (el >> 1)) * 2 + (el & 1);
        dst[o] = h1;
        dst[o + 512] = h2;
    }
}

__global__ void pack_conv1x1_b3_kernel(const float* wp, unsigned short* dst, int CinP, int CoutP) {
    const long n = (long)CinP * CoutP;
    const int NS = CoutP / 32;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % CoutP), ci = (int)(i / CoutP);
        const float u = wp[i];
        const unsigned short h1 = px_bf16_rne(u);
        const float r1 = u - px_bf16_f32(h1);                   // exact
        const unsigned short h2 = px_bf16_rne(r1);
        const float r2 = r1 - px_bf16_f32(h2);                  // exact, at most 8 significant bits
        const unsigned short h3 = px_bf16_rne(r2);
        const int cc = ci & 15, h = cc & 1, el = cc >> 1, lane = h * 32 + (co & 31);
        const long o = (((long)(ci >> 4) * NS + (co >> 5)) * 3) * 512 + (lane * 4 + (el >> 1)) * 2 + (el & 1);
        dst[o] = h1;
        dst[o + 512] = h2;
        dst[o + 1024] = h3;
    }
}

long conv1x1_h2_weight_floats(int CinP, int CoutP, int np) {
    return np == 2 ? Pieces<2>::HDR + (long)CinP * CoutP : (long)CinP * CoutP / 2 * 3;
}

// np = 2: `wh` must start with a zero header word (the running maximum).  Every piece is written.
int launch_pack_conv1x1_h2(const float* wp, float* wh, int CinP, int CoutP, hipStream_t s, int np) {
    MCVD_REQUIRE(CinP % 16 == 0 && CoutP % 32 == 0 && (np == 2 || np == 3), "pack_conv1x1_h2: CinP=%d CoutP=%d np=%d", CinP, CoutP, np);
    const long n = (long)CinP * CoutP;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (np == 2) {
        hipLaunchKernelGGL(q1_absmax_kernel, dim3(blocks), dim3(256), 0, s, wp, n, reinterpret_cast<unsigned*>(wh));
        hipLaunchKernelGGL(pack_conv1x1_h2_kernel, dim3(blocks), dim3(256), 0, s, wp, wh, CinP, CoutP);
    } else {
        hipLaunchKernelGGL(pack_conv1x1_b3_kernel, dim3(blocks), dim3(256), 0, s, wp, reinterpret_cast<unsigned short*>(wh), CinP, CoutP);
    }
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mcvd
