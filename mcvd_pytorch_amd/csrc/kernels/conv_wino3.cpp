// 3x3 convolution by Winograd F(2x2, 3x3) with the channel contraction on the BF16 matrix pipe AT FP32 ACCURACY:
// both MFMA operands are split, exactly, into three bf16 pieces  v = v1 + v2 + v3  (round-to-nearest at every level, each
// remainder is exact in fp32, |v - v1 - v2 - v3| <= 2^-27 |v|) and the product is accumulated in fp32 from the six piece
// products of weight >= 2^-16:
//     u * v  ~=  u1 v3 + u3 v1 + u2 v2 + u1 v2 + u2 v1 + u1 v1          (dropped: u2 v3 + u3 v2 + u3 v3 <= 2^-23.4 |u v|)
// Each piece product is exact in the fp32 accumulator (8 x 8 significant bits), so the result differs from the fp32-MFMA kernel
// (conv_wino.cpp) by less than one fp32 rounding per product; tests/test_gpu_parity.py holds both kernels to the same
// tolerances and compares them with an fp64 convolution.  v_mfma_f32_32x32x16_bf16 retires 16 channels x 32 x 32 in 32 cycles
// where v_mfma_f32_32x32x2_f32 needs 8 x 64: six of them cost 3/8 of the fp32 pipe time.
//
// Same decomposition as conv_wino.cpp (region of 8 x 16 output pixels = 32 tiles, 32*COT output channels, all 16 transform
// positions, 16 input channels per chunk, transformed weights streamed from global memory straight into registers in the SAME
// packed layout), different machine mapping:
//   * 512 threads = 8 waves = TWO waves per SIMD, 256 registers each: wave w owns positions 2w and 2w+1 (2*COT accumulator
//     tiles = 96 registers at COT = 3).  The operand pieces need registers the 128-register budget of four waves per SIMD
//     does not have.
//   * weights arrive as fp32 (4 bytes per element from L2, the scarcest stream of this kernel) and are split in registers
//     just before their MFMAs.  The packed layout keeps the four k-pairs of one (unit, cout sub-tile) in ONE float4, so the
//     pairs the split works on are adjacent registers (v_pk_add_f32, no moves).  Every weight register is reloaded for the
//     next chunk as soon as its split has been issued: the prefetch distance is a full chunk for all of them.
//   * the activated input patch sits in LDS with the two channels of a K pair interleaved ([pair][row][col][2]): the transform
//     reads (channel a, channel a+2) of two columns with one ds_read_b128 and runs on packed fp32 (v_pk_add_f32); its results
//     are split by the transform threads and parked as three bf16 planes [piece][position][k half][k pair][tile] in 32-bit
//     words (two channels per word): the transform's stores and the B-operand reads are both conflict-free.
//   * K-slot convention of the 32x32x16 MFMA (both operands): lane half h, element e  <->  channel 2e + h of the chunk.  That is
//     the order in which the packed weights already sit in a lane (conv_wino.cpp: pack_wino_weight_kernel).
//   * the two waves of a SIMD run the chunk in opposite orders -- waves 0-3: patch + transform, then MFMAs; waves 4-7: MFMAs,
//     then patch + transform -- so the matrix pipe and the VALU of a SIMD are both busy through the chunk.  Measured
//     (profiles/r02_wino3_kloop.txt): the VALU issues one wave instruction per 4 cycles per SIMD, and the operand splits make
//     this kernel VALU-bound, not matrix-bound.
// VMEM of the K loop is hand-counted (inline asm loads + s_waitcnt vmcnt(N)) exactly as in conv_wino.cpp; tools/check_wino_isa.py
// checks the generated code of this file too.
#include <stdlib.h>

#include "../common.h"

namespace mcvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_w3(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

constexpr int W3_CK = 16;        // input channels per chunk = K of one bf16 MFMA
constexpr int W3_T = 32;         // tiles per workgroup (4 x 8 tiles = 8 x 16 output pixels)
constexpr int W3_NT = 512;
constexpr int W3_PP = 24;        // LDS patch row pitch (conv_wino.cpp: WR_PP)
constexpr int W3_VW = 3 * 16 * 2 * 4 * W3_T;      // 32-bit words of one V chunk: [piece][position][half][pair][tile]

// (lo, hi) -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned w3_cvt_pk(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// exact three-way split of two fp32 values into packed bf16 pairs (w1 the leading pieces)
__device__ __forceinline__ void w3_split(float x, float y, unsigned& w1, unsigned& w2, unsigned& w3) {
    w1 = w3_cvt_pk(x, y);
    float rx = x - __builtin_bit_cast(float, w1 << 16), ry = y - __builtin_bit_cast(float, w1 & 0xffff0000u);
    w2 = w3_cvt_pk(rx, ry);
    rx -= __builtin_bit_cast(float, w2 << 16);
    ry -= __builtin_bit_cast(float, w2 & 0xffff0000u);
    w3 = w3_cvt_pk(rx, ry);
}
// the same on an adjacent register pair: the two subtractions of a level are one v_pk_add_f32
__device__ __forceinline__ void w3_split2(f32x2 v, unsigned& w1, unsigned& w2, unsigned& w3) {
    w1 = w3_cvt_pk(v.x, v.y);
    f32x2 h = {__builtin_bit_cast(float, w1 << 16), __builtin_bit_cast(float, w1 & 0xffff0000u)};
    v = v - h;
    w2 = w3_cvt_pk(v.x, v.y);
    f32x2 g = {__builtin_bit_cast(float, w2 << 16), __builtin_bit_cast(float, w2 & 0xffff0000u)};
    v = v - g;
    w3 = w3_cvt_pk(v.x, v.y);
}
__device__ __forceinline__ f32x16 w3_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// PRO: 0 raw input, 1 affine, 2 affine + SiLU (the GroupNorm / temb prologue of conv_wino.cpp)
// a.ksplit == 2 (grid.y = 2): half of the input channels per workgroup, raw partial result to a.part[half] (conv_wino.cpp).
// EXP != 0: timing-only ablations of the K loop (wrong results; env MCVD_WINO3_EXP, tests/gpu_diag.py w3exp): bit 0 no tile
//     transform, bit 1 no patch activation/park, bit 2 no VMEM in the loop, bit 3 no B-operand reads, bit 4 no MFMA, bit 6 no
//     weight split (the raw bits are fed to the matrix pipe).
constexpr int W3_NVGPR = 202;    // registers the compiler may allocate; v202-v255 hold the in-flight loads (see W3_LOAD_A)
template <int COT, int PRO, int EXP = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(202))) void conv_wino3_kernel(ConvArgs a) {
    constexpr int NT = W3_NT, CK = W3_CK, T = W3_T, BCO = 32 * COT, PP = W3_PP, VW = W3_VW;
    constexpr int PSZ = CK * 10 * PP;           // activated input patch of one chunk: [CK][10 rows][PP]
    constexpr int PBUF = PSZ + 4;               // + dump space for unused patch slots
    constexpr int PCOUNT = CK * 10 * 18;        // patch elements loaded per chunk
    constexpr int MAXP = (PCOUNT + NT - 1) / NT;                // 6 loads per thread and chunk
    constexpr int NA = 4 * COT;                                 // weight loads per wave and chunk: 2 units x 2 positions x COT
    constexpr int VM_A = 2 * (2 * COT - 1) + MAXP;              // see W3_MFMA_PHASE
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sV = reinterpret_cast<unsigned*>(smem);           // [2][VW]
    float* sP = smem + 2 * VW;                  // [2][PBUF]
    float* sCo = sP + 2 * PBUF;                 // [Cin][2] prologue coefficients (A_c, B_c) of this sample (PRO only)
    unsigned* sOff = reinterpret_cast<unsigned*>(sCo + 2 * a.Cin);      // [MAXP][NT] byte offsets of the patch-load slots (read by their owner only)

    {   // the kernel descriptor must allocate all 256 registers: the asm statements below name v202-v255 in their text only
        float top;
        asm volatile("" : "={v255}"(top));
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int H = a.H, W = a.W, HW = H * W, Cin = a.Cin;
    const int rx_n = W >> 4, ry_n = H >> 3;
    const int nreg = a.B * rx_n * ry_n;
    // block id -> (region, cout tile): the cout tiles of one region run at the same time on the same XCD (conv_wino.cpp)
    const int nct = a.CoutP / BCO;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int reg_id = (slot / nct) * 8 + xcd;
    const int cotile = slot - (slot / nct) * nct;
    if (reg_id >= nreg) return;
    const int b = reg_id / (rx_n * ry_n);
    const int rr = reg_id - b * (rx_n * ry_n);
    const int oy0 = (rr / rx_n) * 8, ox0 = (rr % rx_n) * 16;
    const int co0 = cotile * BCO;
    const int rg = __builtin_amdgcn_readfirstlane(wave >> 2);   // rows 2rg, 2rg+1 of B^T d; phase order of the wave

    // ---- transform role: (channel pair, tile) = tid & 255.  Pair s_cp = channels (s_ca, s_ca + 2), s_ca = 4*(s_cp >> 1) + (s_cp & 1):
    //      the low and high bf16 of word (k half s_cp & 1, k pair s_cp >> 1) of the B operand.
    const int s_tile = tid & 31, s_cp = (tid & 255) >> 5;
    const int s_ty = s_tile >> 3, s_tx = s_tile & 7;
    // LDS patch: [pair 8][10 rows][PP columns][2 channels] floats.  Rows rg, rg+1, rg+2 of the tile's 4x4 window:
    const int p_rd = ((s_cp * 10 + 2 * s_ty + rg) * PP + 2 * s_tx) * 2;
    // word of (piece 0, position 8*rg, half, pair, tile); one position further = 256 words, one piece = 4096
    const int v_wr = ((8 * rg * 2 + (s_cp & 1)) * 4 + (s_cp >> 1)) * T + s_tile;

    // ---- patch-load slots (chunk invariant): p_pk = LDS float index of the element (12 bits) | channel code << 12, code = channel
    // in chunk, + CK when the element is padding / unused; sOff[sl][tid] = byte offset of the (clamped) pixel from the chunk's first
    // channel plane (parked in LDS: six registers the MFMA phase needs more)
    unsigned p_pk[MAXP];
#pragma unroll
    for (int sl = 0; sl < MAXP; ++sl) {
        const int e = sl * NT + tid;
        if (e < PCOUNT) {
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

    // ---- weight fetch: unit u = 8 input channels; positions 2w and 2w+1 are adjacent in the packed layout:
    //      float4 f of position 2w+i of unit u at  wr_base + u * (16*COT*256) + (i*COT + f) * 256 + lane*4   floats
    const int nunits = a.CinP / 8;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const float* wr_base = a.wpw + (((long)cotile * nunits * 16 + 2 * wave_u) * COT) * 256;
    const unsigned wr_voff = (unsigned)lane * 16u;

    /* weights of chunk `ch`, step st = (position 2w + st / COT, cout sub-tile st % COT): the float4 of the first unit (channels  \
       0-7 of the chunk) -> AW[st][0], of the second -> AW[st][1]; float4 ct of position i of unit u at                        \
       wr_base + u * (16*COT*256) + (i*COT + ct) * 256 + lane*4 floats */                                                      \
#define W3_LOAD_A(ch, st, D0, D1, D2, D3)                                                                       \
    {                                                                                                           \
        const float* ua = wr_base + (long)(2 * (ch)) * (16 * COT * 256) + (st) * 256;                           \
        const float* ub = ua + 16 * COT * 256;                                                                  \
        W3_QUADS(W3_LD1, 2 * (st), ua, D0, D1, D2, D3)                                                          \
        W3_QUADS(W3_LD1, 2 * (st) + 1, ub, D0, D1, D2, D3)                                                      \
    }
    /* IN-FLIGHT DATA LIVES IN REGISTERS THE COMPILER DOES NOT ALLOCATE.  The kernel is compiled with amdgpu_num_vgpr(W3_NVGPR): v202-v255
       are never touched by generated code.  The asm loads write them (weight quad q = 2*step + unit: v[208 + 4q : 211 + 4q]; patch
       slots: v202-v207), the waits are bare s_waitcnt, and the first instructions that consume the data read them by name.  Loads
       whose results are compiler-visible values ("=v" outputs, even with the registers threaded through the wait as "+v" operands)
       are not safe here: the register allocator may assign the result and the operand of the later wait to different registers and
       copy between them while the load is still in flight (it did -- a wrong result once in ~10^4 launches).
       Ordering without "volatile" on the consumers (volatile asm fences the instruction scheduler): a wait hands out a token (an SGPR)
       that the reads of the registers take as an operand, and the reload of a register takes the values computed from its old
       contents as operands -- none of them appears in the instruction text. */
#define W3_QUADS(X, q, P, D0, D1, D2, D3) X(0, "v[208:211]", q, P, D0, D1, D2, D3) X(1, "v[212:215]", q, P, D0, D1, D2, D3) X(2, "v[216:219]", q, P, D0, D1, D2, D3) X(3, "v[220:223]", q, P, D0, D1, D2, D3) X(4, "v[224:227]", q, P, D0, D1, D2, D3) X(5, "v[228:231]", q, P, D0, D1, D2, D3) X(6, "v[232:235]", q, P, D0, D1, D2, D3) X(7, "v[236:239]", q, P, D0, D1, D2, D3) X(8, "v[240:243]", q, P, D0, D1, D2, D3) X(9, "v[244:247]", q, P, D0, D1, D2, D3) X(10, "v[248:251]", q, P, D0, D1, D2, D3) X(11, "v[252:255]", q, P, D0, D1, D2, D3)
#define W3_LD1(K, R, q, P, D0, D1, D2, D3) if ((q) == K) asm volatile("global_load_dwordx4 " R ", %0, %1" :: "v"(wr_voff), "s"(P), "v"(D0), "v"(D1), "v"(D2), "v"(D3) : "memory");
#define W3_WAIT(N) asm volatile("s_waitcnt vmcnt(%1)\n\ts_mov_b32 %0, 0" : "=s"(vtok) : "n"(N) : "memory");
    /* pair pi = 4*step + r of the weight registers (k slots 2r, 2r+1 of the step): W1 <- its leading bf16 pieces, REM <- the pair
       minus H (the caller's unpacked W1): the only two reads of the raw weights */
#define W3_PAIRS(X, pi, W1, REM, H) X(0, "v208", "v209", "v[208:209]", pi, W1, REM, H) X(1, "v210", "v211", "v[210:211]", pi, W1, REM, H) X(2, "v212", "v213", "v[212:213]", pi, W1, REM, H) X(3, "v214", "v215", "v[214:215]", pi, W1, REM, H) X(4, "v216", "v217", "v[216:217]", pi, W1, REM, H) X(5, "v218", "v219", "v[218:219]", pi, W1, REM, H) X(6, "v220", "v221", "v[220:221]", pi, W1, REM, H) X(7, "v222", "v223", "v[222:223]", pi, W1, REM, H) X(8, "v224", "v225", "v[224:225]", pi, W1, REM, H) X(9, "v226", "v227", "v[226:227]", pi, W1, REM, H) X(10, "v228", "v229", "v[228:229]", pi, W1, REM, H) X(11, "v230", "v231", "v[230:231]", pi, W1, REM, H) X(12, "v232", "v233", "v[232:233]", pi, W1, REM, H) X(13, "v234", "v235", "v[234:235]", pi, W1, REM, H) X(14, "v236", "v237", "v[236:237]", pi, W1, REM, H) X(15, "v238", "v239", "v[238:239]", pi, W1, REM, H) X(16, "v240", "v241", "v[240:241]", pi, W1, REM, H) X(17, "v242", "v243", "v[242:243]", pi, W1, REM, H) X(18, "v244", "v245", "v[244:245]", pi, W1, REM, H) X(19, "v246", "v247", "v[246:247]", pi, W1, REM, H) X(20, "v248", "v249", "v[248:249]", pi, W1, REM, H) X(21, "v250", "v251", "v[250:251]", pi, W1, REM, H) X(22, "v252", "v253", "v[252:253]", pi, W1, REM, H) X(23, "v254", "v255", "v[254:255]", pi, W1, REM, H)
#define W3_CVT1(K, RA, RB, RP, pi, W1, REM, H) if ((pi) == K) asm("v_cvt_pk_bf16_f32 %0, " RA ", " RB : "=v"(W1) : "s"(vtok));
#define W3_SUB1(K, RA, RB, RP, pi, W1, REM, H) if ((pi) == K) asm("v_pk_add_f32 %0, " RP ", %1 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(REM) : "v"(H), "s"(vtok));
#define W3_RAW1(K, RA, RB, RP, pi, W1, REM, H) if ((pi) == K) asm("v_mov_b32 %0, " RA "\n\tv_mov_b32 %1, " RB : "=&v"(W1), "=&v"(REM) : "s"(vtok));
    /* unconditional, clamped raw loads of the patch of chunk `ch` (conv_wino.cpp: WR_LOAD_P) */
#define W3_LOAD_P(ch, DEP)                                                                                      \
    {                                                                                                           \
        const int cb = min((ch) * CK, Cin - 1);                                                                 \
        const unsigned lim = (unsigned)((Cin - cb) * HW - 1) * 4u;                                              \
        const bool second = cb >= a.C0;                                                                         \
        const float* srcb = second ? a.x1 + ((long)b * a.C1 + (cb - a.C0)) * HW : a.x0 + ((long)b * a.C0 + cb) * HW; \
        unsigned off[MAXP];          /* all offsets first: ONE LDS round trip (the asm loads below are not reordered) */ \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl)                                                     \
            off[sl] = min(sOff[sl * NT + tid], lim);      /* channels past the last one are zeroed at the write: any address inside the source will do */ \
        asm volatile("global_load_dword v202, %0, %6\n\tglobal_load_dword v203, %1, %6\n\tglobal_load_dword v204, %2, %6\n\t" \
                     "global_load_dword v205, %3, %6\n\tglobal_load_dword v206, %4, %6\n\tglobal_load_dword v207, %5, %6"       \
                     :: "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "s"(srcb),           \
                        "v"(DEP[0]), "v"(DEP[1]), "v"(DEP[2]), "v"(DEP[3]), "v"(DEP[4]), "v"(DEP[5]) : "memory");           \
    }
    /* the same loads of the first two chunks as ordinary (compiler-tracked) loads: prologue only */
#define W3_LOAD_Q(ch, D)                                                                                        \
    {                                                                                                           \
        const int cb = min((ch) * CK, Cin - 1);                                                                 \
        const unsigned lim = (unsigned)((Cin - cb) * HW - 1) * 4u;                                              \
        const bool second = cb >= a.C0;                                                                         \
        const float* srcb = second ? a.x1 + ((long)b * a.C1 + (cb - a.C0)) * HW : a.x0 + ((long)b * a.C0 + cb) * HW; \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl)                                                     \
            D[sl] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(srcb) + min(sOff[sl * NT + tid], lim)); \
    }
    /* activate once per pixel (coefficients from the LDS table) and park the patch in LDS; zero padding applies AFTER     \
       the activation */                                                                                           \
#define W3_WRITE_P(ch, D, FROM_REGS, PV)                                                                         \
    {                                                                                                           \
        float* sPw = sP + (((ch) & 1) ? PBUF : 0);                                                              \
        const int nvalid = Cin - (ch) * CK;                                                                     \
        f32x2 cfv[MAXP];                     /* all coefficient reads first: ONE LDS round trip */               \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                                   \
            cfv[sl] = f32x2{1.0f, 0.0f};                                                                        \
            if (PRO >= 1) {                                                                                     \
                const int cch = min((ch) * CK + (int)((p_pk[sl] >> 12) & (CK - 1)), Cin - 1);                   \
                cfv[sl] = *reinterpret_cast<const f32x2*>(sCo + cch * 2);                                       \
            }                                                                                                   \
        }                                                                                                       \
        if (FROM_REGS) {      /* v = A * raw + B straight out of the patch registers (PRO 0: A = 1, B = 0, exact) */ \
            asm("v_fma_f32 %0, v202, %6, %7\n\tv_fma_f32 %1, v203, %8, %9\n\tv_fma_f32 %2, v204, %10, %11\n\t"            \
                         "v_fma_f32 %3, v205, %12, %13\n\tv_fma_f32 %4, v206, %14, %15\n\tv_fma_f32 %5, v207, %16, %17"    \
                : "=&v"(PV[0]), "=&v"(PV[1]), "=&v"(PV[2]), "=&v"(PV[3]), "=&v"(PV[4]), "=&v"(PV[5])                      \
                : "v"(cfv[0].x), "v"(cfv[0].y), "v"(cfv[1].x), "v"(cfv[1].y), "v"(cfv[2].x), "v"(cfv[2].y),               \
                  "v"(cfv[3].x), "v"(cfv[3].y), "v"(cfv[4].x), "v"(cfv[4].y), "v"(cfv[5].x), "v"(cfv[5].y), "s"(vtok));   \
        } else {                                                                                                \
            _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) PV[sl] = PRO >= 1 ? __builtin_fmaf(D[sl], cfv[sl].x, cfv[sl].y) : D[sl]; \
        }                                                                                                       \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                                   \
            float v = PV[sl];                                                                                   \
            if (PRO >= 2) v = silu_w3(v);                                                                       \
            sPw[p_pk[sl] & 0xfff] = ((int)(p_pk[sl] >> 12) < min(nvalid, CK)) ? v : 0.0f;                      \
        }                                                                                                       \
    }
    /* rows 2rg and 2rg+1 of B^T d for the two channels of the pair (packed fp32: .x = channel s_ca, .y = s_ca + 2), (.) B,      \
       three-way split, 24 stores:                                                                                          \
       row 0: d0 - d2   row 1: d1 + d2   row 2: d2 - d1   row 3: d1 - d3;   (.) B: m0 - m2, m1 + m2, m2 - m1, m1 - m3 */      \
#define W3_WRITE_V(ch, RG)                                                                                      \
    {                                                                                                           \
        const f32x2* sPr = reinterpret_cast<const f32x2*>(sP + (((ch) & 1) ? PBUF : 0) + p_rd);                 \
        unsigned* vdst = sV + (((ch) & 1) ? VW : 0) + v_wr;                                                     \
        f32x2 mx[4], my[4];                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            const f32x2 r0 = sPr[j], r1 = sPr[PP + j], r2 = sPr[2 * PP + j];                                    \
            if ((RG) == 0) { mx[j] = r0 - r2; my[j] = r1 + r2; }                                                \
            else { mx[j] = r1 - r0; my[j] = r0 - r2; }                                                          \
        }                                                                                                       \
        _Pragma("unroll") for (int row = 0; row < 2; ++row) {                                                   \
            const f32x2 m0 = row ? my[0] : mx[0], m1 = row ? my[1] : mx[1], m2 = row ? my[2] : mx[2], m3 = row ? my[3] : mx[3]; \
            const f32x2 v0 = m0 - m2, v1 = m1 + m2, v2 = m2 - m1, v3 = m1 - m3;                                 \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                     \
                unsigned w1, w2, w3;                                                                            \
                w3_split2(q == 0 ? v0 : q == 1 ? v1 : q == 2 ? v2 : v3, w1, w2, w3);                            \
                vdst[(row * 4 + q) * 256] = w1;                                                                 \
                vdst[(row * 4 + q) * 256 + 4096] = w2;                                                          \
                vdst[(row * 4 + q) * 256 + 8192] = w3;                                                          \
            }                                                                                                   \
        }                                                                                                       \
    }
    /* B operand of position 2w+i -> BQ[piece][pair]  <-  word (((p*16 + pos)*2 + half)*4 + jp)*T + l31 */
#define W3_LOAD_B(i, BQ)                                                                                        \
    {                                                                                                           \
        const unsigned* q = sVc + (((2 * wave + (i)) * 2 + half) * 4) * T + l31;                                \
        _Pragma("unroll") for (int jp = 0; jp < 4; ++jp) {                                                      \
            BQ[0][jp] = q[jp * T]; BQ[1][jp] = q[4096 + jp * T]; BQ[2][jp] = q[8192 + jp * T];                  \
        }                                                                                                       \
    }
    /* step st, pair r (k slots 2r, 2r+1): the first two levels of the three-way split of its two weights per lane -> D[0][r],   \
       D[1][r]; RV[r] keeps the remainder for W3_SPLIT_B (the third piece).  Registers r = 0, 1 of a piece come from the first   \
       unit's quad (k slots 0-3), r = 2, 3 from the second's (slots 4-7): adjacent register pairs */                           \
#define W3_SPLIT_A(st, r, D, RV)                                                                                \
    {                                                                                                           \
        unsigned w1, w2;                                                                                        \
        if (EXP & 64) {                                                                                         \
            float ax, ay;                                                                                       \
            W3_PAIRS(W3_RAW1, 4 * (st) + (r), ax, ay, ax)                                                       \
            w1 = __builtin_bit_cast(unsigned, ax); w2 = __builtin_bit_cast(unsigned, ay);                       \
            RV[r] = f32x2{ax, ay};                                                                              \
        } else {                                                                                                \
            f32x2 v;                                                                                            \
            W3_PAIRS(W3_CVT1, 4 * (st) + (r), w1, v, v)                                                         \
            const f32x2 h = {__builtin_bit_cast(float, w1 << 16), __builtin_bit_cast(float, w1 & 0xffff0000u)}; \
            W3_PAIRS(W3_SUB1, 4 * (st) + (r), w1, v, h)                                                         \
            w2 = w3_cvt_pk(v.x, v.y);                                                                           \
            const f32x2 g = {__builtin_bit_cast(float, w2 << 16), __builtin_bit_cast(float, w2 & 0xffff0000u)}; \
            RV[r] = v - g;                                                                                      \
        }                                                                                                       \
        D[0][r] = w1; D[1][r] = w2;                                                                             \
    }
#define W3_SPLIT_B(D, RV) { _Pragma("unroll") for (int r = 0; r < 4; ++r) D[2][r] = w3_cvt_pk(RV[r].x, RV[r].y); }
    /* all MFMAs of chunk `ch` (V(ch) in LDS).  One wave issues at most one VALU instruction per ~5 cycles and its six MFMAs of a  \
       step depend on each other (one accumulator): each MFMA is therefore followed, in program order, by a slice of the NEXT      \
       step's weight split (8 VALU ~ the 32 cycles the matrix pipe needs), with a scheduling fence behind every slice -- left to   \
       itself the compiler issues the six MFMAs back to back and the wave sits through 6 x 32 cycles without issuing anything.    \
       NEXT: the two weight quads of a step are reloaded for chunk ch+1 right after their split (the reload takes the split's     \
       remainders as operands: ordering).  In-order VMEM bookkeeping: when step st's weights are needed, the loads issued after    \
       them are the later steps' of the same chunk, one patch group and the earlier steps' of the next chunk: always               \
       2*(2*COT - 1) + MAXP */                                                                                                  \
#define W3_MFMA_PHASE(ch, NEXT)                                                                                 \
    {                                                                                                           \
        const unsigned* sVc = sV + (((ch) & 1) ? VW : 0);                                                       \
        u32x4 bq[3], pc[2][3];                                                                                  \
        f32x2 rv[4];                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if (!(EXP & 8)) W3_LOAD_B(0, bq)                                                                        \
        else { _Pragma("unroll") for (int p = 0; p < 3; ++p) bq[p] = u32x4{1, 2, 3, 4}; }                       \
        if (NEXT && !(EXP & (4 | 512))) W3_WAIT((EXP & 256) ? VM_A - MAXP : VM_A)                               \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) W3_SPLIT_A(0, r, pc[0], rv)                               \
        W3_SPLIT_B(pc[0], rv)                                                                                   \
        if (NEXT && !(EXP & (4 | 512))) W3_LOAD_A((ch) + 1, 0, rv[0].x, rv[1].x, rv[2].x, rv[3].x)              \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int st = 0; st < 2 * COT; ++st) {                                                \
            if (st == COT && !(EXP & 8)) W3_LOAD_B(1, bq)      /* (behind the last MFMA of position 2w) */      \
            const bool more = st + 1 < 2 * COT;                                                                 \
            if (more && NEXT && !(EXP & (4 | 512))) W3_WAIT((EXP & 256) ? VM_A - MAXP : VM_A)                   \
            f32x16 ca = acc[st / COT][st % COT];                                                                 \
            /* the six piece products, smallest first; D = pc[st & 1], next step's pieces -> pc[(st + 1) & 1] */ \
            if (!(EXP & 16)) ca = w3_mfma(pc[st & 1][0], bq[2], ca);                                            \
            if (more) W3_SPLIT_A(st + 1, 0, pc[(st + 1) & 1], rv)                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            if (!(EXP & 16)) ca = w3_mfma(pc[st & 1][2], bq[0], ca);                                            \
            if (more) W3_SPLIT_A(st + 1, 1, pc[(st + 1) & 1], rv)                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            if (!(EXP & 16)) ca = w3_mfma(pc[st & 1][1], bq[1], ca);                                            \
            if (more) W3_SPLIT_A(st + 1, 2, pc[(st + 1) & 1], rv)                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            if (!(EXP & 16)) ca = w3_mfma(pc[st & 1][0], bq[1], ca);                                            \
            if (more) W3_SPLIT_A(st + 1, 3, pc[(st + 1) & 1], rv)                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            if (!(EXP & 16)) ca = w3_mfma(pc[st & 1][1], bq[0], ca);                                            \
            if (more) W3_SPLIT_B(pc[(st + 1) & 1], rv)                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            if (!(EXP & 16)) ca = w3_mfma(pc[st & 1][0], bq[0], ca);                                            \
            else ca[0] += __builtin_bit_cast(float, pc[st & 1][0][0] ^ pc[st & 1][1][1] ^ pc[st & 1][2][2] ^ pc[st & 1][0][3] ^ bq[0][0] ^ bq[1][1] ^ bq[2][2]); \
            acc[st / COT][st % COT] = ca;                                                                       \
            if (more && NEXT && !(EXP & (4 | 512))) W3_LOAD_A((ch) + 1, st + 1, rv[0].x, rv[1].x, rv[2].x, rv[3].x) \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
        }                                                                                                       \
    }
    /* patch of chunk ch+2 -> LDS, raw patch of chunk ch+3 requested, V(ch+1) -> LDS */
#define W3_VALU_PHASE(ch, RG)                                                                                   \
    {                                                                                                           \
        if (!(EXP & (4 | 256))) W3_WAIT((EXP & 512) ? 0 : NA)                                                   \
        float pv[MAXP] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                                                        \
        if (!(EXP & 2)) W3_WRITE_P((ch) + 2, q0, true, pv)                                                      \
        if (!(EXP & (4 | 256))) W3_LOAD_P((ch) + 3, pv)                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if (!(EXP & 1)) W3_WRITE_V((ch) + 1, RG)                                                                 \
    }

    f32x16 acc[2][COT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][ct][r] = 0.0f;

    // diagnostics (mcvd_ctx_set_debug_buffer): shader-clock time the wave a.wdma spends per phase
    const bool rec = a.dbg != nullptr && wave == a.wdma;
    unsigned long long tk0 = 0, tprev = 0, dt[2] = {0, 0}, pt[3] = {0, 0, 0};
    if (rec) tk0 = tprev = __builtin_amdgcn_s_memtime();
#define W3_STAMP(i)                                                                                             \
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
    float q0[MAXP], q1[MAXP];                           // patches of the first two chunks: prologue only
    int vtok = 0;                                       // ordering token: written by every VMEM wait, an operand of the register reads
    {
        f32x2 cfl = {1.0f, 0.0f};
        _Pragma("unroll") for (int st = 0; st < 2 * COT; ++st) W3_LOAD_A(c_begin, st, 0.f, 0.f, 0.f, 0.f)
        W3_LOAD_Q(c_begin, q0)
        W3_LOAD_Q(c_begin + 1, q1)
        W3_LOAD_P(c_begin + 2, q0)
        if (PRO) {
            for (int c = tid; c < Cin; c += NT) {
                if (a.coef) cfl = *reinterpret_cast<const f32x2*>(a.coef + ((long)b * Cin + c) * 2);
                *reinterpret_cast<f32x2*>(sCo + c * 2) = cfl;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ONE memory latency for everything above
        if (PRO) __syncthreads();          // coefficient table visible
        {
            float pv[MAXP];
            W3_WRITE_P(c_begin, q0, false, pv)
            W3_WRITE_P(c_begin + 1, q1, false, pv)
        }
    }
    __syncthreads();                       // the first two patches visible
    W3_WRITE_V(c_begin, rg)
    __syncthreads();                       // V of the first chunk visible
    W3_STAMP(0)

    // ---- K loop.  VMEM issue order of a wave per chunk c (in-order vmcnt counter; nothing else is outstanding):
    //   waves 0-3:  [patch(c+3): MAXP loads] [weights(c+1): 2 loads after each of the 2*COT splits]      waves 4-7:  weights, then patch
    // wait points (the same counts in both orders):
    //   patch(c+2) before its write: one chunk's weight loads were issued after it                              vmcnt(NA)
    //   weights(c) of step st before their split: see W3_MFMA_PHASE                                             vmcnt(VM_A)
    // (the loads still in flight when a loop is left target registers the compiler does not know: one wait behind the loops)
    const int ph = (EXP & 128) ? __builtin_amdgcn_readfirstlane(wave & 1) : rg;     // phase order of the wave
    if (ph == 0) {
        for (int c = c_begin; c + 1 < c_end; ++c) {
            W3_VALU_PHASE(c, rg)
            W3_MFMA_PHASE(c, true)
            // chunk c read by every wave; V(c+1), patch(c+2) visible.  LDS traffic only: no VMEM wait at the barrier.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
        for (int c = c_begin; c + 1 < c_end; ++c) {
            W3_MFMA_PHASE(c, true)
            W3_VALU_PHASE(c, rg)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    W3_WAIT(0)
    {
        const int c = c_end - 1;
        W3_MFMA_PHASE(c, false)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---------------- inverse transform + epilogue, one 32-cout sub-tile at a time ----------------
    float* sM = smem;                      // [16 positions][32 couts][32 tiles] = 64 KiB
    const int e_tile = tid & 31, e_col0 = tid >> 5;            // two (cout, tile) tasks per thread: couts e_col0 and e_col0 + 16
    const int e_ty = e_tile >> 3, e_tx = e_tile & 7;
    const long pix = (long)(oy0 + 2 * e_ty) * W + ox0 + 2 * e_tx;
    const bool fin = ksp == 1;                 // K split: bias, residual and scale are applied by the reduce kernel
    float* const ydst = fin ? a.y : a.part + (long)kh * a.B * a.Cout * HW;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the K loop is done with the LDS
    W3_STAMP(1)
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
            f32x2 r0 = {0.0f, 0.0f}, r1 = {0.0f, 0.0f};
            if (a.res && fin) {
                const long o = ((long)b * a.Cout + min(co, a.Cout - 1)) * HW + pix;
                r0 = *reinterpret_cast<const f32x2*>(a.res + o);
                r1 = *reinterpret_cast<const f32x2*>(a.res + o + W);
            }
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
            const float bvv = fin ? a.bias[co] : 0.0f;          // zero-padded to CoutP
            const float osc = fin ? a.out_scale : 1.0f;
            const float v00 = (y00 + bvv + r0.x) * osc, v01 = (y01 + bvv + r0.y) * osc;
            const float v10 = (y10 + bvv + r1.x) * osc, v11 = (y11 + bvv + r1.y) * osc;
            if (co < a.Cout) {
                const long o = ((long)b * a.Cout + co) * HW + pix;
                *reinterpret_cast<float2*>(ydst + o) = make_float2(v00, v01);
                *reinterpret_cast<float2*>(ydst + o + W) = make_float2(v10, v11);
            }
            if (a.stats && fin) {
                // GroupNorm partials of the FINAL values (ConvArgs::stats): the 32 tiles of this cout are the 32 lanes of a
                // half-wave; exact per-lane (mean, M2) of its 2x2 pixels, then equal-count pairwise merges over DPP moves
                // (conv_wino.cpp has the derivation).
                float mu = 0.25f * ((v00 + v01) + (v10 + v11));
                const float d0 = v00 - mu, d1 = v01 - mu, d2 = v10 - mu, d3 = v11 - mu;
                float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                float hn = 2.0f;
#define W3_MERGE(CTRL, ROWMASK)                                                                                     \
                {                                                                                                   \
                    const float mo = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, mu), __builtin_bit_cast(int, mu), CTRL, ROWMASK, 0xf, false)); \
                    const float qo = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m2), __builtin_bit_cast(int, m2), CTRL, ROWMASK, 0xf, false)); \
                    const float dd = mu - mo;                                                                       \
                    m2 = (m2 + qo) + dd * dd * hn;                                                                  \
                    mu = 0.5f * (mu + mo);                                                                          \
                    hn += hn;                                                                                       \
                }
                W3_MERGE(0xB1, 0xf)                   // quad_perm [1,0,3,2]
                W3_MERGE(0x4E, 0xf)                   // quad_perm [2,3,0,1]
                W3_MERGE(0x124, 0xf)                  // row_ror:4
                W3_MERGE(0x128, 0xf)                  // row_ror:8
                W3_MERGE(0x142, 0xa)                  // row_bcast:15: lanes 16-31 / 48-63 take the total of the row below
#undef W3_MERGE
                if (e_tile == 31 && co < a.Cout) {
                    float* q = a.stats + (((long)b * a.Cout + co) * (rx_n * ry_n) + rr) * 2;
                    q[0] = mu * 128.0f;               // the partial's sum over its 128 pixels
                    q[1] = m2;
                }
            }
        }
        if (ct + 1 < COT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (rec) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long* d = a.dbg + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 8;
            d[0] = dt[0]; d[1] = dt[1]; d[2] = pt[0]; d[3] = pt[1]; d[4] = pt[2];
            d[5] = now - tprev;            // epilogue
            d[6] = (unsigned long long)(c_end - c_begin);
            d[7] = now - tk0;
        }
    }
#undef W3_STAMP
#undef W3_LOAD_A
#undef W3_QUADS
#undef W3_LD1
#undef W3_SPLIT_A
#undef W3_SPLIT_B
#undef W3_LOAD_P
#undef W3_LOAD_Q
#undef W3_WAIT
#undef W3_PAIRS
#undef W3_CVT1
#undef W3_SUB1
#undef W3_RAW1
#undef W3_WRITE_P
#undef W3_WRITE_V
#undef W3_LOAD_B
#undef W3_MFMA_PHASE
#undef W3_VALU_PHASE
}

static size_t wino3_lds_bytes(int Cin) {
    return (size_t)(2 * W3_VW + 2 * (W3_CK * 10 * W3_PP + 4) + 2 * Cin + 6 * W3_NT) * sizeof(float);
}

// the K-split second pass lives in conv_wino.cpp
int launch_wino_ksplit_reduce(const ConvArgs& a, hipStream_t s);

template <int COT, int PRO, int EXP>
static int wino3_launch_k(const ConvArgs& k, dim3 grid, size_t lds, hipStream_t s) {
    static PerDeviceOnce raised;
    if (raised.first_use()) {
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3_kernel<COT, PRO, EXP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised.done();
    }
    hipLaunchKernelGGL((conv_wino3_kernel<COT, PRO, EXP>), grid, dim3(W3_NT), lds, s, k);
    return 0;
}

template <int COT, int PRO>
static int wino3_launch2(const ConvArgs& a, hipStream_t s) {
    constexpr int BCO = 32 * COT;
    const size_t lds = wino3_lds_bytes(a.Cin);
    const int nreg = a.B * (a.H / 8) * (a.W / 16);
    const int ksp = a.ksplit == 2 ? 2 : 1;
    dim3 grid(((nreg + 7) / 8) * 8 * (a.CoutP / BCO), ksp);
    ConvArgs k = a;
    if (k.dbg) {
        const char* w = getenv("MCVD_DBG_WAVE");       // which wave records its phase times (diagnostics)
        k.wdma = w ? atoi(w) : 0;
    }
    const char* exp_s = getenv("MCVD_WINO3_EXP");          // read per launch: the diagnostics script flips it between runs
    const int e = exp_s ? atoi(exp_s) : 0;
    int rc = 0;
    if (COT == 3 && PRO == 2 && e != 0) {                  // timing-only ablations (tests/gpu_diag.py w3exp)
        switch (e) {
            case 1: rc = wino3_launch_k<3, 2, 1>(k, grid, lds, s); break;        // no transform
            case 2: rc = wino3_launch_k<3, 2, 2>(k, grid, lds, s); break;        // no patch activation / park
            case 4: rc = wino3_launch_k<3, 2, 4>(k, grid, lds, s); break;        // no VMEM in the loop
            case 64: rc = wino3_launch_k<3, 2, 64>(k, grid, lds, s); break;      // no weight split
            case 15: rc = wino3_launch_k<3, 2, 15>(k, grid, lds, s); break;      // weight split + MFMA only
            case 79: rc = wino3_launch_k<3, 2, 79>(k, grid, lds, s); break;      // MFMA only
            case 16: rc = wino3_launch_k<3, 2, 16>(k, grid, lds, s); break;      // everything but the MFMAs
            case 80: rc = wino3_launch_k<3, 2, 80>(k, grid, lds, s); break;      // no MFMA, no weight split
            case 27: rc = wino3_launch_k<3, 2, 27>(k, grid, lds, s); break;      // VMEM + weight split only
            case 91: rc = wino3_launch_k<3, 2, 91>(k, grid, lds, s); break;      // VMEM only
            case 128: rc = wino3_launch_k<3, 2, 128>(k, grid, lds, s); break;    // phase order by wave parity instead of wave / 4
            case 256: rc = wino3_launch_k<3, 2, 256>(k, grid, lds, s); break;    // no patch loads in the loop
            case 512: rc = wino3_launch_k<3, 2, 512>(k, grid, lds, s); break;    // no weight loads in the loop
            default: mcvd::set_error("MCVD_WINO3_EXP=%d is not a built ablation", e); return -1;
        }
    } else {
        rc = wino3_launch_k<COT, PRO, 0>(k, grid, lds, s);
    }
    if (rc) return rc;
    MCVD_HIP_CHECK(hipGetLastError());
    if (ksp == 2) return launch_wino_ksplit_reduce(a, s);
    if (a.stats) set_last_conv_stats_np((a.H / 8) * (a.W / 16));
    return 0;
}

template <int COT>
static int wino3_launch(const ConvArgs& a, hipStream_t s) {
    if (!a.coef && !a.act) return wino3_launch2<COT, 0>(a, s);
    if (!a.act) return wino3_launch2<COT, 1>(a, s);
    return wino3_launch2<COT, 2>(a, s);
}

// Shape ids 10 / 11 apply to this launch: regions of 8 x 16 output pixels (the 8x8 layers stay with conv_wino.cpp), no SPADE
// prologue, packed weights present (11: and an even chunk count).
bool conv_wino3_usable(const ConvArgs& a) {
    return a.ks == 3 && a.H % 8 == 0 && a.W % 16 == 0 && a.H >= 8 && a.W >= 16 && a.wpw && !a.gb && a.Cin <= 1024 &&
           a.CinP % W3_CK == 0 && (a.C1 == 0 || a.C0 % W3_CK == 0) && a.H * a.W <= 16384 &&
           (long)a.B * (a.C0 > a.C1 ? a.C0 : a.C1) * a.H * a.W < (1L << 29) && wino3_lds_bytes(a.Cin) <= 160 * 1024 &&
           (a.ksplit != 2 || ((a.CinP / W3_CK) % 2 == 0 && a.CinP / W3_CK >= 4 && a.part != nullptr));
}

// a.wpw: the layout of launch_pack_wino_weight, packed for conv_wino_cout_tile(Cout) (shared with conv_wino.cpp).
int launch_conv_wino3(const ConvArgs& a, hipStream_t s) {
    MCVD_REQUIRE(conv_wino3_usable(a), "winograd bf16x3 conv: unsupported (ks=%d H=%d W=%d Cin=%d C0=%d ksplit=%d, packed weights %s)",
                 a.ks, a.H, a.W, a.Cin, a.C0, a.ksplit, a.wpw ? "present" : "missing");
    const int cot = conv_wino_cout_tile(a.Cout);
    MCVD_REQUIRE(a.CoutP % (32 * cot) == 0, "winograd bf16x3 conv: CoutP=%d vs tile %d", a.CoutP, 32 * cot);
    switch (cot) {
        case 1: return wino3_launch<1>(a, s);
        case 2: return wino3_launch<2>(a, s);
        default: return wino3_launch<3>(a, s);
    }
}

}  // namespace mcvd
