// 3x3 convolution by Winograd F(2x2, 3x3) with the channel contraction on the FP16 matrix pipe, both MFMA operands split into TWO
// fp16 pieces  v ~= v1 + v2  (round-to-nearest at both levels: |v - v1 - v2| <= 2^-22 |v|, i.e. operands of 22-23 significant
// bits where fp32 has 24) and the product accumulated in fp32 from the three piece products of weight >= 2^-11:
//     u * v  ~=  u1 v2 + u2 v1 + u1 v1                                       (dropped: u2 v2 <= 2^-22 |u v|)
// Each piece product is exact in the fp32 accumulator (11 x 11 significant bits).  The representation error of the operands is
// random in sign and averages out over the K*16 products of an output, where the rounding of the fp32 accumulation (common to
// every kernel here, ~sqrt(K) * 2^-24) does not: tests/test_gpu_parity.py measures this kernel against an fp64 convolution next to
// the fp32-MFMA kernel (conv_wino.cpp) and holds it to the same parity tolerances.  It is NOT bit-equivalent to an fp32
// computation (conv_wino3.cpp, three bf16 pieces and six products, is to within one rounding per product): the context option
// "f16x2" decides whether the autotuner may pick it, and bench.py names the arithmetic in `dtype`.
//
// Why: v_mfma_f32_32x32x16_f16 retires 16 channels x 32 x 32 in 32 cycles; three of them cost 3/16 of the fp32 pipe time and
// half of conv_wino3.cpp's six.  More important, the WEIGHT pieces are split once, when the weights are packed
// (pack_wino2h_weight_kernel: per layer scaled by a power of two so that max |U| sits at 2^13..2^14, far from both ends of the fp16
// range, the inverse folded into the epilogue), so the K loop carries no VALU work for the A operand at all -- conv_wino3.cpp
// spends 240 of its ~430 VALU instructions per wave and chunk on that split and is VALU-bound (profiles/r02_wino3_kloop.txt).
// Two fp16 pieces are 4 bytes per weight: the stream from L2 is the same size as the fp32 weights.
//
// Same decomposition as conv_wino3.cpp (region of 8 x 16 output pixels = 32 tiles, 32*COT output channels, all 16 transform
// positions, 16 input channels per chunk; 512 threads = 8 waves = two waves per SIMD, wave w owns positions 2w and 2w+1):
//   * weights: [cout tile][chunk][position][cout sub-tile][piece][64 lanes][4 dwords], a dword = two fp16 = K slots (2j, 2j+1)
//     of the lane's half; the 4*COT quads a wave needs per chunk are one contiguous block, fetched with global_load_dwordx4
//     STRAIGHT INTO THE MFMA A-OPERAND REGISTERS (named registers the compiler does not allocate, v208-v255: see H2_LOAD_A) and
//     reloaded for the next chunk as soon as the position's MFMAs have been issued: prefetch distance = one chunk.
//   * K-slot convention of the 32x32x16 MFMA (both operands): lane half h, element e  <->  channel 2e + h of the chunk.
//   * activations: patch in LDS, channel pairs interleaved; the transform runs on packed fp32, clamps to the fp16 range, splits
//     (v_cvt_pk_f16_f32, v_fma_mix_f32 for the exact remainder, v_cvt_pk_f16_f32: 4 VALU per channel pair and position) and
//     parks two fp16 planes [piece][position][k half][k pair][tile].  Activations are scaled by 2^4 on their way into the patch.
//   * the two waves of a SIMD run the chunk in opposite orders (patch + transform | MFMAs), one barrier per chunk.
//   * the MFMAs are inline asm (their A operand is a named register); the 3 * COT MFMAs of a position are ordered piece-major, so
//     consecutive MFMAs target different accumulators.
// VMEM of the K loop is hand-counted (inline asm loads + s_waitcnt vmcnt(N)) exactly as in conv_wino3.cpp; tools/check_wino_isa.py
// checks the generated code of this file too.
#include <stdlib.h>

#include "../common.h"

namespace mcvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_h2(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

constexpr int H2_CK = 16;        // input channels per chunk = K of one fp16 MFMA
constexpr int H2_T = 32;         // tiles per workgroup (4 x 8 tiles = 8 x 16 output pixels)
constexpr int H2_NT = 512;
constexpr int H2_PP = 24;        // LDS patch row pitch (conv_wino.cpp: WR_PP)
constexpr int H2_VW = 2 * 16 * 2 * 4 * H2_T;      // 32-bit words of one V chunk: [piece][position][half][pair][tile]
constexpr int H2_HDR = 4;        // header floats in front of the packed weight pieces: |w|max, scale, 1 / scale, -
constexpr float H2_ACT_SCALE = 16.0f;             // activations enter the patch times 2^4 (exact): transformed values of the
                                                  // order 10..1e3, second pieces clear of the fp16 denormal range.  |B^T d B| must
                                                  // stay below 65504 / 16 ~ 4094 (GroupNorm-ed inputs -- the only ones the library
                                                  // gives this kernel -- cannot get there); beyond it the result is NaN, not a
                                                  // saturated number

// (lo, hi) -> packed fp16 pair, round to nearest even (v_cvt_pk_f16_f32)
__device__ __forceinline__ unsigned h2_cvt_pk(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// two-way split of two fp32 values into packed fp16 pairs: w1 = fp16(v), w2 = fp16(v - w1); v - w1 is exact (v_fma_mix_f32
// reads the fp16 half it subtracts straight out of the packed pair)
__device__ __forceinline__ void h2_split2(f32x2 v, unsigned& w1, unsigned& w2) {
    // no clamp: a transformed value beyond the fp16 range becomes Inf here and NaN in the accumulator -- loud, where a clamp would
    // saturate silently (the library keeps raw, unbounded tensors away from this kernel and reports non-finite results)
    w1 = h2_cvt_pk(v.x, v.y);
    float rx, ry;
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(w1), "v"(v.x));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(w1), "v"(v.y));
    w2 = h2_cvt_pk(rx, ry);
}

// PRO: 0 raw input, 1 affine, 2 affine + SiLU (the GroupNorm / temb prologue of conv_wino.cpp)
// a.ksplit == 2 (grid.y = 2): half of the input channels per workgroup, raw partial result to a.part[half] (conv_wino.cpp).
// EXP != 0 (built with -DMCVD_DIAG only): timing-only ablations of the K loop (wrong results; env MCVD_WINO2H_EXP, tests/gpu_diag.py w3exp): bit 0 no tile
//     transform, bit 1 no patch activation/park, bit 2 no VMEM in the loop, bit 3 no B-operand reads, bit 4 no MFMA.
// G8: 8x8 images -- the 32 tiles of a workgroup are TWO whole images (16 tiles each, image i at patch columns 10 i .. 10 i + 9); every
//     halo element is zero padding, so only the 2 x 64 interior pixels per channel are loaded (slots 0-3 of the six; the other two
//     fetch a dummy) and the halo of both patch buffers is zeroed once.  The coefficient table holds both samples.
template <int COT, int PRO, bool G8, int EXP = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(202))) void conv_wino2h_kernel(ConvArgs a) {
    // amdgpu_num_vgpr(202): registers the compiler may allocate; v202-v255 hold the in-flight loads and the A operands (H2_LOAD_A)
    constexpr int NT = H2_NT, CK = H2_CK, T = H2_T, BCO = 32 * COT, PP = H2_PP, VW = H2_VW;
    constexpr int PSZ = CK * 10 * PP;           // activated input patch of one chunk: [CK][10 rows][PP]
    constexpr int PBUF = PSZ + 4;               // + dump space for unused patch slots
    constexpr int PCOUNT = G8 ? CK * 2 * 64 : CK * 10 * 18;     // patch elements loaded per chunk
    constexpr int MAXP = 6;                                     // load slots per thread and chunk (G8 uses four of them)
    constexpr int NQ = 2 * COT;                                 // weight quads per position: COT cout sub-tiles x 2 pieces
    constexpr int NA = 2 * NQ;                                  // weight loads per wave and chunk
    constexpr int VM_A = NQ + MAXP;                             // see H2_MFMA_PHASE
    static_assert(MAXP == 6 && NA <= 12, "named-register map below: v202-v207 patch, v208-v255 twelve weight quads");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sV = reinterpret_cast<unsigned*>(smem);           // [2][VW]
    float* sP = smem + 2 * VW;                  // [2][PBUF]
    float* sCo = sP + 2 * PBUF;                 // [Cin][2] prologue coefficients (A_c, B_c) of this sample (PRO only; G8: [2][Cin][2])
    unsigned* sOff = reinterpret_cast<unsigned*>(sCo + (G8 ? 4 : 2) * a.Cin);      // [MAXP][NT] byte offsets of the patch-load slots (read by their owner only)

    {   // the kernel descriptor must allocate all 256 registers: the asm statements below name v202-v255 in their text only
        float top;
        asm volatile("" : "={v255}"(top));
    }
    // every kernel argument the prologue needs, fetched NOW (one batch of scalar loads, one wait; conv_wino3.cpp)
    asm volatile("" :: "s"(a.x0), "s"(a.x1), "s"(a.coef), "s"(a.wph), "s"(a.B), "s"(a.H), "s"(a.W), "s"(a.Cin), "s"(a.CinP), "s"(a.C0),
                 "s"(a.C1), "s"(a.CoutP), "s"(a.ksplit), "s"(a.dbg), "s"(a.wdma));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int H = a.H, W = a.W, HW = H * W, Cin = a.Cin;
    const int rx_n = G8 ? 1 : W >> 4, ry_n = G8 ? 1 : H >> 3;
    const int nreg = G8 ? (a.B + 1) >> 1 : a.B * rx_n * ry_n;
    // block id -> (region, cout tile): the cout tiles of one region run at the same time on the same XCD (conv_wino.cpp)
    const int nct = a.CoutP / BCO;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int reg_id = (slot / nct) * 8 + xcd;
    const int cotile = slot - (slot / nct) * nct;
    if (reg_id >= nreg) return;
    const int b = G8 ? 2 * reg_id : reg_id / (rx_n * ry_n);      // (first) sample of the region
    const int rr = G8 ? 0 : reg_id - b * (rx_n * ry_n);
    const int oy0 = (rr / rx_n) * 8, ox0 = (rr % rx_n) * 16;
    const int co0 = cotile * BCO;
    const int rg = __builtin_amdgcn_readfirstlane(wave >> 2);   // rows 2rg, 2rg+1 of B^T d; phase order of the wave

    // prologue coefficients of this sample: (A_c, B_c) of channel tid + k * 512 (Cin <= 1024: at most two table entries per thread and
    // sample).  Regions of a plane: asm loads into v202-v205 (the registers of the third patch) IN FRONT of the first patches, parked in
    // the LDS table once they have landed (conv_wino3.cpp).  G8 (two samples, eight values: more than the six patch registers): ordinary
    // loads, waited for before the first patch request.
    f32x2 cpre[2] = {{1.0f, 0.0f}, {1.0f, 0.0f}};
    f32x2 cpre2[2] = {{1.0f, 0.0f}, {1.0f, 0.0f}};              // G8: the region's second sample (clamped to the last one)
    const float* co_src[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) co_src[k] = a.coef + ((long)b * Cin + min(tid + k * NT, Cin - 1)) * 2;
    if (PRO && G8) {                                            // unconditional (clamped) loads: the wait belongs at the use
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            cpre[k] = *reinterpret_cast<const f32x2*>(co_src[k]);
            cpre2[k] = *reinterpret_cast<const f32x2*>(a.coef + ((long)min(b + 1, a.B - 1) * Cin + min(tid + k * NT, Cin - 1)) * 2);
        }
    }

    // ---- transform role: (channel pair, tile) = tid & 255.  Pair s_cp = channels (s_ca, s_ca + 2), s_ca = 4*(s_cp >> 1) + (s_cp & 1):
    //      the low and high fp16 of word (k half s_cp & 1, k pair s_cp >> 1) of the B operand.
    const int s_tile = tid & 31, s_cp = (tid & 255) >> 5;
    const int s_ty = G8 ? (s_tile >> 2) & 3 : s_tile >> 3, s_tx = G8 ? (s_tile & 3) + 5 * (s_tile >> 4) : s_tile & 7;
    // LDS patch: [pair 8][10 rows][PP columns][2 channels] floats.  Rows rg, rg+1, rg+2 of the tile's 4x4 window:
    const int p_rd = ((s_cp * 10 + 2 * s_ty + rg) * PP + 2 * s_tx) * 2;
    // word of (piece 0, position 8*rg, half, pair, tile); one position further = 256 words, one piece = 4096
    const int v_wr = ((8 * rg * 2 + (s_cp & 1)) * 4 + (s_cp >> 1)) * T + s_tile;

    // ---- patch-load slots (chunk invariant): p_pk = LDS float index of the element (12 bits) | channel code << 12, code = channel
    // in chunk, + CK when the element is padding / unused (| G8: image of the region << 20); sOff[sl][tid] = byte offset of the
    // (clamped) pixel from the chunk's first channel plane (parked in LDS: six registers the MFMA phase needs more)
    unsigned p_pk[MAXP];
#pragma unroll
    for (int sl = 0; sl < MAXP; ++sl) {
        const int e = sl * NT + tid;
        if (G8) {
            if (e < PCOUNT) {                   // e -> (channel, image, row, col) of an interior pixel
                const int ci = e >> 7, img = (e >> 6) & 1, r = (e >> 3) & 7, c = e & 7;
                const bool valid = b + img < a.B;
                const int cp = (ci >> 2) * 2 + (ci & 1), ce = (ci >> 1) & 1;
                p_pk[sl] = (unsigned)(((cp * 10 + r + 1) * PP + img * 10 + c + 1) * 2 + ce) | ((unsigned)(ci + (valid ? 0 : CK)) << 12) |
                           ((unsigned)(valid ? img : 0) << 20);
                sOff[sl * NT + tid] = (unsigned)(ci * HW + r * 8 + c) * 4u;
            } else {
                p_pk[sl] = (unsigned)PSZ | ((unsigned)CK << 12);
                sOff[sl * NT + tid] = 0;
            }
        } else if (e < PCOUNT) {
            const int ci = e / 180, rem = e - ci * 180;
            const int r = rem / 18, c = rem - r * 18;
            const int y = oy0 - 1 + r, x = ox0 - 1 + c;
            const bool inside = y >= 0 && y < H && x >= 0 && x < W;
            const int cp = (ci >> 2) * 2 + (ci & 1), ce = (ci >> 1) & 1;
            p_pk[sl] = (unsigned)(((cp * 10 + r) * PP + c) * 2 + ce) | ((unsigned)(ci + (inside ? 0 : CK)) << 12);
            sOff[sl * NT + tid] = (unsigned)(ci * HW + min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)) * 4u;
        } else {
            p_pk[sl] = (unsigned)PSZ | ((unsigned)CK << 12);
            sOff[sl * NT + tid] = 0;
        }
    }

    // ---- weight fetch: the NA quads of a wave per chunk are contiguous: quad q = (i * COT + ct) * 2 + piece of positions 2w + i at
    //      wr_base + chunk * (16*COT*512) + q * 256 + lane * 4   dwords
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned* wr_base = reinterpret_cast<const unsigned*>(a.wph) + H2_HDR + ((long)cotile * (a.CinP / CK) * 16 + 2 * wave_u) * (COT * 512);
    const unsigned wr_voff = (unsigned)lane * 16u;

    /* IN-FLIGHT DATA LIVES IN REGISTERS THE COMPILER DOES NOT ALLOCATE (conv_wino3.cpp has the story).  The kernel is compiled with
       amdgpu_num_vgpr(202): v202-v255 are never touched by generated code.  The asm loads write them (weight quad q: v[208 + 4q :
       211 + 4q]; patch slots: v202-v207), the waits are bare s_waitcnt, the MFMAs name their A operand in the instruction text.
       Every statement that touches a named register is `asm volatile` (program order among them is kept) except the patch FMAs, which
       take the wait's token (an SGPR) as an operand. */
#define H2_QUADS(X, q, A1, A2) X(0, "v[208:211]", q, A1, A2) X(1, "v[212:215]", q, A1, A2) X(2, "v[216:219]", q, A1, A2) X(3, "v[220:223]", q, A1, A2) X(4, "v[224:227]", q, A1, A2) X(5, "v[228:231]", q, A1, A2) X(6, "v[232:235]", q, A1, A2) X(7, "v[236:239]", q, A1, A2) X(8, "v[240:243]", q, A1, A2) X(9, "v[244:247]", q, A1, A2) X(10, "v[248:251]", q, A1, A2) X(11, "v[252:255]", q, A1, A2)
#define H2_LD1(K, R, q, P, UNUSED) if ((q) == K) asm volatile("global_load_dwordx4 " R ", %0, %1" :: "v"(wr_voff), "s"(P) : "memory");
#define H2_MF1(K, R, q, ACC, BV) if ((q) == K) asm volatile("v_mfma_f32_32x32x16_f16 %0, " R ", %1, %0" : "+v"(ACC) : "v"(BV));
    /* the NQ weight quads of position 2w + i of chunk `ch` */
#define H2_LOAD_A(ch, i)                                                                                        \
    {                                                                                                           \
        const unsigned* ua = wr_base + (long)(ch) * (16 * COT * 512) + (i) * (NQ * 256);                        \
        _Pragma("unroll") for (int qq = 0; qq < NQ; ++qq) { H2_QUADS(H2_LD1, (i) * NQ + qq, ua + qq * 256, 0) } \
    }
    /* weight piece `pc` of every cout sub-tile of position 2w + i of chunk `ch` (quads 2 * (i * COT + ct) + pc) */
#define H2_LOAD_A_PIECE(ch, i, pc)                                                                              \
    {                                                                                                           \
        const unsigned* ua = wr_base + (long)(ch) * (16 * COT * 512) + (i) * (NQ * 256);                        \
        _Pragma("unroll") for (int ct = 0; ct < COT; ++ct) { H2_QUADS(H2_LD1, (i) * NQ + 2 * ct + (pc), ua + (2 * ct + (pc)) * 256, 0) } \
    }
    /* prologue: quads Q0 .. Q1-1 of chunk `ch`, issued behind the instructions that produced DEP (which read the registers) */
#define H2_LD1D(K, R, q, P, DEP) if ((q) == K) asm volatile("global_load_dwordx4 " R ", %0, %1" :: "v"(wr_voff), "s"(P), "v"(DEP[0]), "v"(DEP[1]), "v"(DEP[2]), "v"(DEP[3]), "v"(DEP[4]), "v"(DEP[5]) : "memory");
#define H2_LOAD_A_RANGE(ch, Q0, Q1, DEP)                                                                        \
    {                                                                                                           \
        const unsigned* ua = wr_base + (long)(ch) * (16 * COT * 512);                                           \
        _Pragma("unroll") for (int qq = (Q0); qq < (Q1); ++qq) { H2_QUADS(H2_LD1D, qq, ua + qq * 256, DEP) }    \
    }
#define H2_WAIT(N) asm volatile("s_waitcnt vmcnt(%1)\n\ts_mov_b32 %0, 0" : "=s"(vtok) : "n"(N) : "memory");
    /* unconditional, clamped raw loads of the patch of chunk `ch` (conv_wino.cpp: WR_LOAD_P) */
#define H2_READ_OFF(OFS) { _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) OFS[sl] = sOff[sl * NT + tid]; }
#define H2_LOAD_P(ch, DEP, OFS) H2_LOAD_PR(ch, DEP, OFS, "v202", "v203", "v204", "v205", "v206", "v207")
#define H2_LOAD_PR(ch, DEP, OFS, R0, R1, R2, R3, R4, R5)                                                        \
    {                                                                                                           \
        const int cb = min((ch) * CK, Cin - 1);                                                                 \
        const unsigned lim = (unsigned)((Cin - cb) * HW - 1) * 4u;                                              \
        const bool second = cb >= a.C0;                                                                         \
        const float* srcb = second ? a.x1 + ((long)b * a.C1 + (cb - a.C0)) * HW : a.x0 + ((long)b * a.C0 + cb) * HW; \
        const unsigned istride = (unsigned)((second ? a.C1 : a.C0) * HW) * 4u;      /* G8: distance to the region's second sample */ \
        unsigned off[MAXP];                                                                                     \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl)                                                     \
            off[sl] = min(OFS[sl], lim) + (G8 ? ((p_pk[sl] >> 20) & 1u) * istride : 0u);      /* channels past the last one are zeroed at the write: any address inside the source will do */ \
        asm volatile("global_load_dword " R0 ", %0, %6\n\tglobal_load_dword " R1 ", %1, %6\n\tglobal_load_dword " R2 ", %2, %6\n\t" \
                     "global_load_dword " R3 ", %3, %6\n\tglobal_load_dword " R4 ", %4, %6\n\tglobal_load_dword " R5 ", %5, %6"       \
                     :: "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "s"(srcb),           \
                        "v"(DEP[0]), "v"(DEP[1]), "v"(DEP[2]), "v"(DEP[3]), "v"(DEP[4]), "v"(DEP[5]) : "memory");           \
    }
    /* activate once per pixel (coefficients from the LDS table) and park the patch in LDS; zero padding applies AFTER     \
       the activation */                                                                                           \
#define H2_READ_C(ch, cfv)                                                                                      \
    {                                                                                                           \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                                   \
            cfv[sl] = f32x2{1.0f, 0.0f};                                                                        \
            if (PRO >= 1) {                                                                                     \
                const int cch = min((ch) * CK + (int)((p_pk[sl] >> 12) & (CK - 1)), Cin - 1) + (G8 ? (int)((p_pk[sl] >> 20) & 1u) * Cin : 0); \
                cfv[sl] = *reinterpret_cast<const f32x2*>(sCo + cch * 2);                                       \
            }                                                                                                   \
        }                                                                                                       \
    }
#define H2_NOHOOK(e, v)
#define H2_WRITE_P(ch, PV, cfv) H2_WRITE_PR(ch, PV, cfv, "v202", "v203", "v204", "v205", "v206", "v207", H2_NOHOOK)
    /* HOOK(sl, v): statements placed behind value sl (the prologue issues its weight loads there, one at a time: conv_wino3.cpp) */ \
#define H2_WRITE_PR(ch, PV, cfv, R0, R1, R2, R3, R4, R5, HOOK)                                                  \
    {                                                                                                           \
        float* sPw = sP + (((ch) & 1) ? PBUF : 0);                                                              \
        const int nvalid = Cin - (ch) * CK;                                                                     \
        /* v = A * raw + B straight out of the patch registers (PRO 0: A = 1, B = 0, exact) */                  \
        asm("v_fma_f32 %0, " R0 ", %6, %7\n\tv_fma_f32 %1, " R1 ", %8, %9\n\tv_fma_f32 %2, " R2 ", %10, %11\n\t"             \
                     "v_fma_f32 %3, " R3 ", %12, %13\n\tv_fma_f32 %4, " R4 ", %14, %15\n\tv_fma_f32 %5, " R5 ", %16, %17"     \
            : "=&v"(PV[0]), "=&v"(PV[1]), "=&v"(PV[2]), "=&v"(PV[3]), "=&v"(PV[4]), "=&v"(PV[5])                          \
            : "v"(cfv[0].x), "v"(cfv[0].y), "v"(cfv[1].x), "v"(cfv[1].y), "v"(cfv[2].x), "v"(cfv[2].y),                   \
              "v"(cfv[3].x), "v"(cfv[3].y), "v"(cfv[4].x), "v"(cfv[4].y), "v"(cfv[5].x), "v"(cfv[5].y), "s"(vtok));       \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                                   \
            float v = PV[sl];                                                                                   \
            if (PRO >= 2) v = silu_h2(v);                                                                       \
            sPw[p_pk[sl] & 0xfff] = ((int)((p_pk[sl] >> 12) & 0xff) < min(nvalid, CK)) ? v * H2_ACT_SCALE : 0.0f; \
            HOOK(sl, v)                                                                                         \
        }                                                                                                       \
    }
    /* rows 2rg and 2rg+1 of B^T d for the two channels of the pair (packed fp32: .x = channel s_ca, .y = s_ca + 2), (.) B,      \
       two-way fp16 split, 16 stores:                                                                                        \
       row 0: d0 - d2   row 1: d1 + d2   row 2: d2 - d1   row 3: d1 - d3;   (.) B: m0 - m2, m1 + m2, m2 - m1, m1 - m3 */      \
#define H2_READ_R(ch, RW)                                                                                       \
    {                                                                                                           \
        const f32x2* sPr = reinterpret_cast<const f32x2*>(sP + (((ch) & 1) ? PBUF : 0) + p_rd);                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { RW[0][j] = sPr[j]; RW[1][j] = sPr[PP + j]; RW[2][j] = sPr[2 * PP + j]; } \
    }
#define H2_WRITE_V(ch, RG, RW)                                                                                  \
    {                                                                                                           \
        unsigned* vdst = sV + (((ch) & 1) ? VW : 0) + v_wr;                                                     \
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
                unsigned w1, w2;                                                                                \
                h2_split2(q == 0 ? v0 : q == 1 ? v1 : q == 2 ? v2 : v3, w1, w2);                                \
                vdst[(row * 4 + q) * 256] = w1;                                                                 \
                vdst[(row * 4 + q) * 256 + 4096] = w2;                                                          \
            }                                                                                                   \
        }                                                                                                       \
    }
    /* B operand of position 2w+i -> BQ[piece][pair]  <-  word (((p*16 + pos)*2 + half)*4 + jp)*T + l31 */
#define H2_LOAD_B(i, BQ)                                                                                        \
    {                                                                                                           \
        const unsigned* q = sVc + (((2 * wave + (i)) * 2 + half) * 4) * T + l31;                                \
        _Pragma("unroll") for (int jp = 0; jp < 4; ++jp) {                                                      \
            BQ[0][jp] = q[jp * T]; BQ[1][jp] = q[4096 + jp * T];                                                \
        }                                                                                                       \
    }
    /* all MFMAs of chunk `ch` (V(ch) in LDS, weights(ch) in the named registers).  Per position 2w + i: wait for its NQ quads, 3*COT  \
       MFMAs ordered piece-major (u2 v1, u1 v2, u1 v1: smallest class first; consecutive MFMAs write different accumulators), and -- NEXT \
       -- the same NQ quads are reloaded for chunk ch+1 (the matrix pipe has read its A operands by the time the wave gets past the   \
       MFMA: it issues in order).  In-order VMEM bookkeeping: when the quads of a position are needed, the loads issued after them    \
       are the other position's NQ quads and one patch group: vmcnt(NQ + MAXP), in both phase orders. */                              \
#define H2_MFMA_PHASE(ch, NEXT)                                                                                 \
    {                                                                                                           \
        const unsigned* sVc = sV + (((ch) & 1) ? VW : 0);                                                       \
        u32x4 bq[2][2];                                                                                         \
        if (!(EXP & 8)) { H2_LOAD_B(0, bq[0]) H2_LOAD_B(1, bq[1]) }                                             \
        else { _Pragma("unroll") for (int p = 0; p < 2; ++p) { bq[0][p] = u32x4{1, 2, 3, 4}; bq[1][p] = u32x4{5, 6, 7, 8}; } } \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                         \
            if (NEXT && !(EXP & 4)) H2_WAIT(VM_A)                                                               \
            if (!(EXP & 16)) {                                                                                  \
                /* u2 v1 first: the second weight piece is re-requested for chunk ch+1 right behind its only product, the first     \
                   piece behind the last one (conv_wino3.cpp: all quads in one burst stall the wave in the issue) */                 \
                _Pragma("unroll") for (int ct = 0; ct < COT; ++ct) { H2_QUADS(H2_MF1, 2 * (i * COT + ct) + 1, acc[i][ct], bq[i][0]) }  \
                if (NEXT && !(EXP & 4)) H2_LOAD_A_PIECE((ch) + 1, i, 1)                                         \
                _Pragma("unroll") for (int ct = 0; ct < COT; ++ct) { H2_QUADS(H2_MF1, 2 * (i * COT + ct), acc[i][ct], bq[i][1]) }      \
                _Pragma("unroll") for (int ct = 0; ct < COT; ++ct) { H2_QUADS(H2_MF1, 2 * (i * COT + ct), acc[i][ct], bq[i][0]) }      \
                if (NEXT && !(EXP & 4)) H2_LOAD_A_PIECE((ch) + 1, i, 0)                                         \
            } else {                                                                                            \
                _Pragma("unroll") for (int ct = 0; ct < COT; ++ct)                                              \
                    acc[i][ct][0] += __builtin_bit_cast(float, bq[i][0][0] ^ bq[i][1][1] ^ bq[i][0][2] ^ bq[i][1][3]); \
                if (NEXT && !(EXP & 4)) H2_LOAD_A((ch) + 1, i)                                                  \
            }                                                                                                   \
        }                                                                                                       \
    }
    /* patch of chunk ch+2 -> LDS, raw patch of chunk ch+3 requested, V(ch+1) -> LDS */
#define H2_VALU_PHASE(ch, RG)                                                                                   \
    {                                                                                                           \
        /* every LDS read of the phase is issued up front: ONE round trip (coefficients, load offsets, the 12 patch pairs) */ \
        f32x2 cfv[MAXP], rw[3][4];                                                                              \
        unsigned ofs[MAXP];                                                                                     \
        if (!(EXP & 2)) H2_READ_C((ch) + 2, cfv)                                                                \
        if (!(EXP & 4)) H2_READ_OFF(ofs)                                                                        \
        if (!(EXP & 1)) H2_READ_R((ch) + 1, rw)                                                                 \
        if (!(EXP & 4)) H2_WAIT(NA)                                                                             \
        float pv[MAXP] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                                                        \
        if (!(EXP & 2)) H2_WRITE_P((ch) + 2, pv, cfv)                                                           \
        if (!(EXP & 4)) H2_LOAD_P((ch) + 3, pv, ofs)                                                            \
        if (!(EXP & 1)) H2_WRITE_V((ch) + 1, RG, rw)                                                            \
    }

    f32x16 acc[2][COT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][ct][r] = 0.0f;

    // diagnostics (mcvd_ctx_set_debug_buffer): shader-clock time the wave a.wdma spends per phase
    const bool rec = a.dbg != nullptr && wave == (a.wdma & 63);
    const bool sub = (a.wdma & 64) != 0;           // record prologue / epilogue sub-phase stamps instead of the wall clock
    unsigned long long sp[3] = {0, 0, 0};
    unsigned long long tk0 = 0, tprev = 0, dt[2] = {0, 0}, rt0 = 0;
    if (rec) {
        rt0 = __builtin_amdgcn_s_memrealtime();          // constant 100 MHz: start / end of the workgroup on the wall clock
        tk0 = tprev = __builtin_amdgcn_s_memtime();
    }
#define H2_STAMP(i)                                                                                             \
    if (rec) {                                                                                                  \
        const unsigned long long now = __builtin_amdgcn_s_memtime();                                            \
        dt[i] += now - tprev;                                                                                   \
        tprev = now;                                                                                            \
    }

    // ---- chunk range of this workgroup (a.ksplit == 2: blockIdx.y picks one half of the input channels)
    const int nch_all = a.CinP / CK;
    const int ksp = a.ksplit == 2 ? 2 : 1, kh = ksp == 2 ? (int)blockIdx.y : 0;
    const int c_begin = kh * (nch_all / ksp), c_end = c_begin + nch_all / ksp;

    // ---- prologue.  Issue order = need order: the coefficients of the sample (into the registers of the third patch), the raw patches of
    // the first two chunks (into the registers of weight quads 0-2, which are not needed before the first MFMA phase), then the weight
    // quads 3.. of the first chunk.  Coefficients and patches are consumed as soon as THEY have landed (the weights, 3/4 of the bytes, are
    // still in flight); the third patch and quads 0-2 follow once their registers have been read.  (With one wait for everything the prologue took 10-12 k cycles, of
    // which 5-8 k went into pulling ~130 KB through the CU's memory pipe before any work started: profiles/r02_wino2h_prologue.txt.)
    int vtok = 0;                                       // ordering token: written by every VMEM wait, an operand of the register reads
    {
        float nodep[MAXP] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (PRO && G8) {                   // (the compiler waits for the coefficient loads here: nothing else is in flight yet)
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (tid + k * NT < Cin) {
                    *reinterpret_cast<f32x2*>(sCo + (tid + k * NT) * 2) = cpre[k];
                    *reinterpret_cast<f32x2*>(sCo + (Cin + tid + k * NT) * 2) = cpre2[k];
                }
        }
        if (G8)                            // the halo of both patch buffers is zero padding for the whole kernel
            for (int i = tid; i < 2 * PBUF; i += NT) sP[i] = 0.0f;
        unsigned ofs[MAXP];
        H2_READ_OFF(ofs)
        if (PRO && !G8)
            asm volatile("global_load_dwordx2 v[202:203], %0, off\n\tglobal_load_dwordx2 v[204:205], %1, off"
                         :: "v"(co_src[0]), "v"(co_src[1]) : "memory");
        H2_LOAD_PR(c_begin, nodep, ofs, "v208", "v209", "v210", "v211", "v212", "v213")
        H2_LOAD_PR(c_begin + 1, nodep, ofs, "v214", "v215", "v216", "v217", "v218", "v219")
        if (rec) sp[0] = __builtin_amdgcn_s_memtime() - tk0;      // loads issued
        H2_WAIT(0)                         // the coefficients and the two patches have landed
        if (rec) sp[1] = __builtin_amdgcn_s_memtime() - tk0;      // first patches landed
        float cdep[MAXP] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (PRO && !G8) {
            asm volatile("v_mov_b32 %0, v202\n\tv_mov_b32 %1, v203\n\tv_mov_b32 %2, v204\n\tv_mov_b32 %3, v205"
                         : "=v"(cpre[0].x), "=v"(cpre[0].y), "=v"(cpre[1].x), "=v"(cpre[1].y) : "s"(vtok));
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (tid + k * NT < Cin) *reinterpret_cast<f32x2*>(sCo + (tid + k * NT) * 2) = cpre[k];
            cdep[0] = cdep[1] = cpre[0].x + cpre[0].y;
            cdep[2] = cdep[3] = cpre[1].x + cpre[1].y;
        }
        H2_LOAD_P(c_begin + 2, cdep, ofs)  // (behind the reads of v202-v205)
        if (PRO || G8) __syncthreads();    // coefficient table (and the zeroed halo) visible
        {
            float pv0[MAXP], pv1[MAXP];
            f32x2 cf0[MAXP], cf1[MAXP];
            H2_READ_C(c_begin, cf0)
            H2_READ_C(c_begin + 1, cf1)
            // the weight quads 3.. of the first chunk are requested one behind each activated value (conv_wino3.cpp: in one burst they
            // fill the CU's vector-memory queue and the waves sit in the issue instead of working on their patches)
            constexpr int QR = (NA - 3 + 2 * MAXP - 1) / (2 * MAXP) > 0 ? (NA - 3 + 2 * MAXP - 1) / (2 * MAXP) : 1;       // quads per value
#define H2_HOOKQ(j, v) { const float hd[MAXP] = {v, v, v, v, v, v}; H2_LOAD_A_RANGE(c_begin, (3 + (j) * QR < NA ? 3 + (j) * QR : NA), (3 + ((j) + 1) * QR < NA ? 3 + ((j) + 1) * QR : NA), hd) }
#define H2_HOOK0(sl, v) H2_HOOKQ(sl, v)
#define H2_HOOK1(sl, v) H2_HOOKQ(MAXP + (sl), v)
            H2_WRITE_PR(c_begin, pv0, cf0, "v208", "v209", "v210", "v211", "v212", "v213", H2_HOOK0)
            H2_WRITE_PR(c_begin + 1, pv1, cf1, "v214", "v215", "v216", "v217", "v218", "v219", H2_HOOK1)
#undef H2_HOOK0
#undef H2_HOOK1
#undef H2_HOOKQ
            float dep[MAXP];
            _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) dep[sl] = pv0[sl] + pv1[sl];
            H2_LOAD_A_RANGE(c_begin, 0, NA < 3 ? NA : 3, dep)
        }
    }
    __syncthreads();                       // the first two patches visible
    if (rec) sp[2] = __builtin_amdgcn_s_memtime() - tk0;          // first two patches activated and parked
    {
        f32x2 rw[3][4];
        H2_READ_R(c_begin, rw)
        H2_WRITE_V(c_begin, rg, rw)
    }
    __syncthreads();                       // V of the first chunk visible
    H2_STAMP(0)

    // ---- K loop.  VMEM issue order of a wave per chunk c (in-order vmcnt counter; nothing else is outstanding):
    //   waves 0-3:  [patch(c+3): MAXP loads] [weights(c+1): NQ loads behind the MFMAs of each position]      waves 4-7:  weights, then patch
    // wait points (the same counts in both orders):
    //   patch(c+2) before its write: one chunk's weight loads were issued after it                              vmcnt(NA)
    //   weights(c) of a position before its MFMAs: see H2_MFMA_PHASE                                            vmcnt(VM_A)
    // (the loads still in flight when a loop is left target registers the compiler does not know: one wait behind the loops)
    H2_WAIT(0)                             // weight quads 0-2 were issued last: the loop's in-order counts start from an empty queue
    const int ph = (EXP & 128) ? __builtin_amdgcn_readfirstlane(wave & 1) : rg;     // phase order of the wave
    if (ph == 0) {
        for (int c = c_begin; c + 1 < c_end; ++c) {
            H2_VALU_PHASE(c, rg)
            H2_MFMA_PHASE(c, true)
            // chunk c read by every wave; V(c+1), patch(c+2) visible.  LDS traffic only: no VMEM wait at the barrier.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
        for (int c = c_begin; c + 1 < c_end; ++c) {
            H2_MFMA_PHASE(c, true)
            H2_VALU_PHASE(c, rg)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    H2_WAIT(0)
    {
        const int c = c_end - 1;
        H2_MFMA_PHASE(c, false)
    }
    // The MFMAs are inline asm: the compiler does not know that the accumulators were written by the matrix pipe and inserts none of
    // the wait states a read of an MFMA result needs (8-pass MFMA -> VALU / LDS read: 11).  Nothing in the K loop reads them.
    // The nops are tied to the accumulators ("+v"): a free-standing asm could be scheduled away from the values it protects.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (COT == 3)
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]) :: "memory");
    else if constexpr (COT == 2)
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]) :: "memory");
    else
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0][0]), "+v"(acc[1][0]) :: "memory");

    // ---------------- inverse transform + epilogue, one 32-cout sub-tile at a time ----------------
    float* sM = smem;                      // [16 positions][32 couts][32 tiles] = 64 KiB
    const int e_tile = tid & 31, e_col0 = tid >> 5;            // two (cout, tile) tasks per thread: couts e_col0 and e_col0 + 16
    const int e_ty = G8 ? (e_tile >> 2) & 3 : e_tile >> 3, e_tx = G8 ? e_tile & 3 : e_tile & 7;
    const int e_b = min(b + (G8 ? e_tile >> 4 : 0), a.B - 1);          // G8: the tile's sample (clamped for the loads)
    const bool e_valid = !G8 || b + (e_tile >> 4) < a.B;
    const long pix = (long)(oy0 + 2 * e_ty) * W + ox0 + 2 * e_tx;
    const bool fin = ksp == 1;                 // K split: bias, residual and scale are applied by the reduce kernel
    float* const ydst = fin ? a.y : a.part + (long)kh * a.B * a.Cout * HW;
    const float inv = a.wph[2] * (1.0f / H2_ACT_SCALE);       // 1 / (weight scale of this layer * activation scale): both powers of two
    // bias and residual of all 2 * COT tasks of the thread are requested up front (conv_wino3.cpp: fetched where they are used, every
    // sub-tile waited for two dependent memory latencies between its barriers)
    float e_bias[COT][2];
    f32x2 e_r0[COT][2], e_r1[COT][2];
#pragma unroll
    for (int ct = 0; ct < COT; ++ct)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int co = co0 + ct * 32 + e_col0 + 16 * t2;
            e_bias[ct][t2] = fin ? a.bias[co] : 0.0f;           // zero-padded to CoutP
            e_r0[ct][t2] = e_r1[ct][t2] = f32x2{0.0f, 0.0f};
            if (a.res && fin) {
                const long o = ((long)e_b * a.Cout + min(co, a.Cout - 1)) * HW + pix;
                e_r0[ct][t2] = *reinterpret_cast<const f32x2*>(a.res + o);
                e_r1[ct][t2] = *reinterpret_cast<const f32x2*>(a.res + o + W);
            }
        }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the K loop is done with the LDS
    H2_STAMP(1)
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = (r & 3) + 8 * (r >> 2) + 4 * half;
                sM[((2 * wave + i) * 32 + col) * T + l31] = acc[i][ct][r];
            }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int e_col = e_col0 + 16 * t2;
            const int co = co0 + ct * 32 + e_col;
            const f32x2 r0 = e_r0[ct][t2], r1 = e_r1[ct][t2];
            float mm[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) mm[xi] = sM[(xi * 32 + e_col) * T + e_tile];
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
            const float v00 = (y00 * inv + bvv + r0.x) * osc, v01 = (y01 * inv + bvv + r0.y) * osc;
            const float v10 = (y10 * inv + bvv + r1.x) * osc, v11 = (y11 * inv + bvv + r1.y) * osc;
            if (co < a.Cout && e_valid) {
                const long o = ((long)e_b * a.Cout + co) * HW + pix;
                *reinterpret_cast<float2*>(ydst + o) = make_float2(v00, v01);
                *reinterpret_cast<float2*>(ydst + o + W) = make_float2(v10, v11);
            }
            if (a.stats && fin) {
                // GroupNorm partials of the FINAL values (ConvArgs::stats): the tiles of this cout are the 32 lanes of a half-wave (G8:
                // 16 lanes = one DPP row per image).  Pilot-shifted moments: with p = the group's first value (a sample of the
                // distribution, so |mean - p| is a few sigma at most and M2 = q - s^2 / n loses a digit, not the result),
                // s = sum (v - p) and q = sum (v - p)^2 merge by plain addition -- two v_add_f32 with a DPP source per level, where the
                // exact (mean, M2) pairs of conv_wino.cpp cost eight instructions per level and a quarter of this epilogue.
                float pil;
                {
                    const int pv = __builtin_bit_cast(int, v00);
                    const int s0 = __builtin_amdgcn_readlane(pv, 0), s2 = __builtin_amdgcn_readlane(pv, 32);
                    if (G8) {
                        const int s1 = __builtin_amdgcn_readlane(pv, 16), s3 = __builtin_amdgcn_readlane(pv, 48);
                        pil = __builtin_bit_cast(float, (lane & 32) ? ((lane & 16) ? s3 : s2) : ((lane & 16) ? s1 : s0));
                    } else {
                        pil = __builtin_bit_cast(float, (lane & 32) ? s2 : s0);
                    }
                }
                const float d0 = v00 - pil, d1 = v01 - pil, d2 = v10 - pil, d3 = v11 - pil;
                float sm = (d0 + d1) + (d2 + d3);
                float qm = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#define H2_MERGE(CTRL, ROWMASK)                                                                                     \
                {                                                                                                   \
                    sm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sm), CTRL, ROWMASK, 0xf, false)); \
                    qm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, qm), CTRL, ROWMASK, 0xf, false)); \
                }
                H2_MERGE(0xB1, 0xf)                   // quad_perm [1,0,3,2]
                H2_MERGE(0x4E, 0xf)                   // quad_perm [2,3,0,1]
                H2_MERGE(0x124, 0xf)                  // row_ror:4
                H2_MERGE(0x128, 0xf)                  // row_ror:8: every lane of a row of 16 holds the row's totals
                if (!G8) H2_MERGE(0x142, 0xa)         // row_bcast:15: lanes 16-31 / 48-63 add the totals of the row below
#undef H2_MERGE
                const bool writer = G8 ? (e_tile & 15) == 0 : e_tile == 31;
                if (writer && co < a.Cout && e_valid) {
                    constexpr float NPIX = G8 ? 64.0f : 128.0f;
                    float* q = a.stats + (((long)e_b * a.Cout + co) * (rx_n * ry_n) + rr) * 2;
                    q[0] = sm + NPIX * pil;           // the partial's sum over its pixels
                    q[1] = fmaxf(qm - sm * sm * (1.0f / NPIX), 0.0f);      // M2 about the partial's own mean
                }
            }
        }
        if (ct + 1 < COT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (rec) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long* d = a.dbg + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 8;
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            d[0] = dt[0]; d[1] = dt[1]; d[2] = rt0; d[3] = __builtin_amdgcn_s_memrealtime();
            d[4] = ((unsigned long long)xcc << 32) | hwid;          // which CU ran it (gpu_diag.py w2htl: per-CU timeline)
            if (sub) { d[2] = sp[0]; d[3] = sp[1]; d[4] = sp[2]; }   // MCVD_DBG_WAVE >= 64: prologue sub-phases (cycles from the start)
            d[5] = now - tprev;            // epilogue
            d[6] = (unsigned long long)(c_end - c_begin);
            d[7] = now - tk0;
        }
    }
#undef H2_STAMP
#undef H2_LOAD_A
#undef H2_LOAD_A_PIECE
#undef H2_QUADS
#undef H2_LD1
#undef H2_MF1
#undef H2_LOAD_P
#undef H2_LOAD_PR
#undef H2_WRITE_PR
#undef H2_LD1D
#undef H2_LOAD_A_RANGE
#undef H2_WAIT
#undef H2_WRITE_P
#undef H2_WRITE_V
#undef H2_READ_R
#undef H2_READ_C
#undef H2_READ_OFF
#undef H2_LOAD_B
#undef H2_MFMA_PHASE
#undef H2_VALU_PHASE
}

static size_t wino2h_lds_bytes(int Cin, bool g8) {
    return (size_t)(2 * H2_VW + 2 * (H2_CK * 10 * H2_PP + 4) + (g8 ? 4 : 2) * Cin + 6 * H2_NT) * sizeof(float);
}

// the K-split second pass lives in conv_wino.cpp
int launch_wino_ksplit_reduce(const ConvArgs& a, hipStream_t s);

template <int COT, int PRO, bool G8, int EXP>
static int wino2h_launch_k(const ConvArgs& k, dim3 grid, size_t lds, hipStream_t s) {
    static PerDeviceOnce raised;
    if (raised.first_use()) {
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino2h_kernel<COT, PRO, G8, EXP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised.done();
    }
    hipLaunchKernelGGL((conv_wino2h_kernel<COT, PRO, G8, EXP>), grid, dim3(H2_NT), lds, s, k);
    return 0;
}

template <int COT, int PRO, bool G8>
static int wino2h_launch2(const ConvArgs& a, hipStream_t s) {
    constexpr int BCO = 32 * COT;
    const size_t lds = wino2h_lds_bytes(a.Cin, G8);
    const int nreg = G8 ? (a.B + 1) / 2 : a.B * (a.H / 8) * (a.W / 16);
    const int ksp = a.ksplit == 2 ? 2 : 1;
    dim3 grid(((nreg + 7) / 8) * 8 * (a.CoutP / BCO), ksp);
    ConvArgs k = a;
    int rc = 0;
    if (k.dbg) k.wdma = 0;                 // wave 0 records its phase times
#ifdef MCVD_DIAG
    // diagnostics build only (build.py --diag): which wave records, and the timing-only ablations of the K loop (WRONG RESULTS; the
    // production library has neither the env hooks nor the ablation kernels)
    if (k.dbg) {
        const char* w = getenv("MCVD_DBG_WAVE");
        k.wdma = w ? atoi(w) : 0;
    }
    const char* exp_s = getenv("MCVD_WINO2H_EXP");
    const int e = exp_s ? atoi(exp_s) : 0;
    if (COT == 3 && PRO == 2 && !G8 && e != 0) {           // tests/gpu_diag.py w3exp
        switch (e) {
            case 1: rc = wino2h_launch_k<3, 2, false, 1>(k, grid, lds, s); break;        // no transform
            case 2: rc = wino2h_launch_k<3, 2, false, 2>(k, grid, lds, s); break;        // no patch activation / park
            case 3: rc = wino2h_launch_k<3, 2, false, 3>(k, grid, lds, s); break;        // neither
            case 4: rc = wino2h_launch_k<3, 2, false, 4>(k, grid, lds, s); break;        // no VMEM in the loop
            case 16: rc = wino2h_launch_k<3, 2, false, 16>(k, grid, lds, s); break;      // everything but the MFMAs
            case 15: rc = wino2h_launch_k<3, 2, false, 15>(k, grid, lds, s); break;      // MFMA only
            case 27: rc = wino2h_launch_k<3, 2, false, 27>(k, grid, lds, s); break;      // VMEM only
            case 11: rc = wino2h_launch_k<3, 2, false, 11>(k, grid, lds, s); break;      // VMEM + MFMA only
            default: mcvd::set_error("MCVD_WINO2H_EXP=%d is not a built ablation", e); return -1;
        }
    } else
#endif
    {
        rc = wino2h_launch_k<COT, PRO, G8, 0>(k, grid, lds, s);
    }
    if (rc) return rc;
    MCVD_HIP_CHECK(hipGetLastError());
    if (ksp == 2) return launch_wino_ksplit_reduce(a, s);
    if (a.stats) set_last_conv_stats_np(G8 ? 1 : (a.H / 8) * (a.W / 16));
    return 0;
}

template <int COT, bool G8>
static int wino2h_launch1(const ConvArgs& a, hipStream_t s) {
    if (!a.coef && !a.act) return wino2h_launch2<COT, 0, G8>(a, s);
    if (!a.act) return wino2h_launch2<COT, 1, G8>(a, s);
    return wino2h_launch2<COT, 2, G8>(a, s);
}

template <int COT>
static int wino2h_launch(const ConvArgs& a, hipStream_t s) {
    return (a.H == 8 && a.W == 8) ? wino2h_launch1<COT, true>(a, s) : wino2h_launch1<COT, false>(a, s);
}

// Shape ids 12 / 13 apply to this launch: regions of 8 x 16 output pixels or 8 x 8 images (two per workgroup), no SPADE prologue,
// pre-split packed weights present (13: and an even chunk count).
bool conv_wino2h_usable(const ConvArgs& a) {
    const bool g8 = a.H == 8 && a.W == 8;
    return a.ks == 3 && ((a.H % 8 == 0 && a.W % 16 == 0 && a.H >= 8 && a.W >= 16) || g8) && a.wph && !a.gb && a.Cin <= 1024 &&
           a.CinP % H2_CK == 0 && (a.C1 == 0 || a.C0 % H2_CK == 0) && a.H * a.W <= 16384 &&
           (long)a.B * (a.C0 > a.C1 ? a.C0 : a.C1) * a.H * a.W < (1L << 29) && wino2h_lds_bytes(a.Cin, g8) <= 160 * 1024 &&
           (a.ksplit != 2 || ((a.CinP / H2_CK) % 2 == 0 && a.CinP / H2_CK >= 4 && conv_part_fits(a)));
}

// a.wph: the layout of launch_pack_wino2h_weight, packed for conv_wino_cout_tile(Cout).
int launch_conv_wino2h(const ConvArgs& a, hipStream_t s) {
    MCVD_REQUIRE(conv_wino2h_usable(a), "winograd f16x2 conv: unsupported (ks=%d H=%d W=%d Cin=%d C0=%d ksplit=%d, packed weight pieces %s)",
                 a.ks, a.H, a.W, a.Cin, a.C0, a.ksplit, a.wph ? "present" : "missing");
    const int cot = conv_wino_cout_tile(a.Cout);
    MCVD_REQUIRE(a.CoutP % (32 * cot) == 0, "winograd f16x2 conv: CoutP=%d vs tile %d", a.CoutP, 32 * cot);
    switch (cot) {
        case 1: return wino2h_launch<1>(a, s);
        case 2: return wino2h_launch<2>(a, s);
        default: return wino2h_launch<3>(a, s);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight pieces.  wh = [H2_HDR header floats][CinP * 16 * CoutP dwords]; header: |w|max of the layer, scale = 2^e, 1 / scale.
//   U = G g G^T per (cout, cin), times 2^e with e chosen so that 2.25 * |w|max * 2^e <= 2^14 (|U| <= 2.25 |w|max: the rows of G
//   have absolute sums <= 1.5), split u1 = fp16(U), u2 = fp16(U - u1), stored as 16-bit halves at
//   ((((cotile*nchunks + ci/16)*16 + xi)*COT + ct)*2 + piece)*512 + (lane*4 + j)*2 + (e & 1),
//   cc = ci % 16 = 2e + h,  j = e >> 1,  lane = h*32 + co%32,  ct = (co % BCO) / 32.
// The destination must be zero-filled (padded channels stay zero; the header's maximum starts at 0).
__global__ void wino2h_absmax_kernel(const float* w, long n, unsigned* hdr) {
    float m = 0.0f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(hdr, __float_as_uint(m));          // non-negative floats order like their bit patterns
}

__global__ void pack_wino2h_weight_kernel(const float* w, float* wh, int Cout, int Cin, int CinP, int CoutP, int COT) {
    const float wmax = wh[0];
    int k = 0;
    (void)frexpf(2.25f * wmax, &k);                             // 2.25 * wmax = m * 2^k, 0.5 <= m < 1
    const int e = (wmax > 0.0f && wmax < 3.0e38f) ? min(max(14 - k, -60), 60) : 0;
    const float scale = ldexpf(1.0f, e);
    if (blockIdx.x == 0 && threadIdx.x == 0) { wh[1] = scale; wh[2] = ldexpf(1.0f, -e); }
    _Float16* dst = reinterpret_cast<_Float16*>(wh + H2_HDR);
    const long n = (long)Cout * Cin;
    const int BCO = 32 * COT, nch = CinP / 16;
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
        const int cotile = co / BCO, ct = (co % BCO) / 32, cc = ci & 15, h = cc & 1, el = cc >> 1;
        const int lane = h * 32 + (co & 31);
        // halfword index of (xi = 0, piece 0); one position further = COT * 1024 halfwords, the second piece = + 512
        const long base = ((((long)cotile * nch + (ci >> 4)) * 16) * COT + ct) * 1024 + (lane * 4 + (el >> 1)) * 2 + (el & 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {                           // (.) G^T   (the same arithmetic as pack_wino_weight_kernel)
            float u[4];
            u[0] = t[r][0];
            u[1] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
            u[2] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
            u[3] = t[r][2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float us = u[c] * scale;                  // exact (power of two)
                const _Float16 h1 = (_Float16)us;
                const _Float16 h2 = (_Float16)(us - (float)h1);
                const long o = base + (long)(r * 4 + c) * COT * 1024;
                dst[o] = h1;
                dst[o + 512] = h2;
            }
        }
    }
}

// `wh` (conv_wino2h_weight_floats(CinP, CoutP) floats) must be zero-filled by the caller.
int launch_pack_wino2h_weight(const float* w, float* wh, int Cout, int Cin, int CinP, int CoutP, hipStream_t s) {
    const int cot = conv_wino_cout_tile(Cout);
    MCVD_REQUIRE(CinP % 16 == 0 && CoutP % (32 * cot) == 0, "pack_wino2h_weight: CinP=%d CoutP=%d cot=%d", CinP, CoutP, cot);
    const long n = (long)Cout * Cin;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(wino2h_absmax_kernel, dim3(blocks), dim3(256), 0, s, w, n * 9, reinterpret_cast<unsigned*>(wh));
    hipLaunchKernelGGL(pack_wino2h_weight_kernel, dim3(blocks), dim3(256), 0, s, w, wh, Cout, Cin, CinP, CoutP, cot);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

long conv_wino2h_weight_floats(int CinP, int CoutP) { return H2_HDR + (long)CinP * 16 * CoutP; }

}  // namespace mcvd
