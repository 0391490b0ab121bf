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
//     just before their MFMAs; the transformed patches V are split by the transform threads and parked in LDS as three bf16
//     planes, laid out [piece][position][k half][k pair][tile] in 32-bit words (two channels per word), so that the transform's
//     stores and the B-operand reads are both conflict-free.
//   * K-slot convention of the 32x32x16 MFMA (both operands): lane half h, element e  <->  channel 2e + h of the chunk.  That is
//     the order in which the packed weights already sit in a lane (conv_wino.cpp: pack_wino_weight_kernel).
//   * the two waves of a SIMD run the chunk in opposite orders -- waves 0-3: patch + transform, then MFMAs; waves 4-7: MFMAs,
//     then patch + transform -- so the matrix pipe and the VALU of a SIMD are both busy through the chunk.
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
__device__ __forceinline__ f32x16 w3_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// PRO: 0 raw input, 1 affine, 2 affine + SiLU (the GroupNorm / temb prologue of conv_wino.cpp)
// a.ksplit == 2 (grid.y = 2): half of the input channels per workgroup, raw partial result to a.part[half] (conv_wino.cpp).
// EXP != 0: timing-only ablations of the K loop (wrong results; env MCVD_WINO3_EXP, tests/gpu_diag.py w3exp): bit 0 no tile
//     transform, bit 1 no patch activation/park, bit 2 no VMEM in the loop, bit 3 no B-operand reads, bit 4 no MFMA, bit 6 no
//     weight split (the raw bits are fed to the matrix pipe).
template <int COT, int PRO, int EXP = 0>
__global__ __launch_bounds__(512) void conv_wino3_kernel(ConvArgs a) {
    constexpr int NT = W3_NT, CK = W3_CK, T = W3_T, BCO = 32 * COT, PP = W3_PP, VW = W3_VW;
    constexpr int PSZ = CK * 10 * PP;           // activated input patch of one chunk: [CK][10 rows][PP]
    constexpr int PBUF = PSZ + 4;               // + dump space for unused patch slots
    constexpr int PCOUNT = CK * 10 * 18;        // patch elements loaded per chunk
    constexpr int MAXP = (PCOUNT + NT - 1) / NT;                // 6 loads per thread and chunk
    constexpr int NA = 4 * COT;                                 // weight loads per wave and chunk: 2 units x 2 positions x COT
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sV = reinterpret_cast<unsigned*>(smem);           // [2][VW]
    float* sP = smem + 2 * VW;                  // [2][PBUF]
    float* sCo = sP + 2 * PBUF;                 // [Cin][2] prologue coefficients (A_c, B_c) of this sample (PRO only)

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

    // ---- transform role: (channel pair, tile) = tid & 255: the word holds channels s_ca (low bf16) and s_ca + 2 (high bf16)
    const int s_tile = tid & 31, s_ci = (tid & 255) >> 5;
    const int s_h = s_ci & 1, s_jp = s_ci >> 1;
    const int s_ca = 4 * s_jp + s_h;
    const int s_ty = s_tile >> 3, s_tx = s_tile & 7;
    // rows rg, rg+1, rg+2 of the tile's 4x4 window in the LDS patch [ci][10 rows][PP]
    const int p_rd = s_ca * 10 * PP + (2 * s_ty + rg) * PP + 2 * s_tx;
    // word of (piece 0, position 8*rg, half s_h, pair s_jp, tile); one position further = 256 words, one piece = 4096
    const int v_wr = ((8 * rg * 2 + s_h) * 4 + s_jp) * T + s_tile;

    // ---- patch-load slots (chunk invariant); p_ci = channel-in-chunk, or CK + channel when the element is padding / unused
    // one register per slot: LDS word of the element (12 bits) | channel code << 12 (6 bits) | clamped pixel offset << 18 (HW <= 16384)
    unsigned p_pk[MAXP];
#pragma unroll
    for (int sl = 0; sl < MAXP; ++sl) {
        const int e = sl * NT + tid;
        if (e < PCOUNT) {
            const int ci = e / 180, rem = e - ci * 180;
            const int r = rem / 18, c = rem - r * 18;
            const int y = oy0 - 1 + r, x = ox0 - 1 + c;
            const bool inside = y >= 0 && y < H && x >= 0 && x < W;
            p_pk[sl] = (unsigned)(ci * 10 * PP + r * PP + c) | ((unsigned)(ci + (inside ? 0 : CK)) << 12) |
                       ((unsigned)(min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)) << 18);
        } else {
            p_pk[sl] = (unsigned)PSZ | ((unsigned)CK << 12);
        }
    }

    // ---- weight fetch: unit u = 8 input channels; positions 2w and 2w+1 are adjacent in the packed layout:
    //      float4 f of position 2w+i of unit u at  wr_base + u * (16*COT*256) + (i*COT + f) * 256 + lane*4   floats
    const int nunits = a.CinP / 8;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const float* wr_base = a.wpw + (((long)cotile * nunits * 16 + 2 * wave_u) * COT) * 256;
    const unsigned wr_voff = (unsigned)lane * 16u;

    /* weights of chunk `ch`, position 2w+i -> AW[i][0] (first unit: channels 0-7) and AW[i][1] (second unit) */
#define W3_LOAD_A(ch, i) W3_LOAD_A2(ch, i, AW[i][0], AW[i][1])
#define W3_LOAD_A2(ch, i, SA, SB)                                                                               \
    {                                                                                                           \
        const float* ua = wr_base + (long)(2 * (ch)) * (16 * COT * 256) + (i) * (COT * 256);                    \
        const float* ub = ua + 16 * COT * 256;                                                                  \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(SA[0]) : "v"(wr_voff), "s"(ua) : "memory");        \
        if (COT > 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(SA[COT > 1 ? 1 : 0]) : "v"(wr_voff), "s"(ua) : "memory"); \
        if (COT > 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(SA[COT > 2 ? 2 : 0]) : "v"(wr_voff), "s"(ua) : "memory"); \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(SB[0]) : "v"(wr_voff), "s"(ub) : "memory");        \
        if (COT > 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(SB[COT > 1 ? 1 : 0]) : "v"(wr_voff), "s"(ub) : "memory"); \
        if (COT > 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(SB[COT > 2 ? 2 : 0]) : "v"(wr_voff), "s"(ub) : "memory"); \
    }
    /* wait until all but the N youngest VMEM operations of this wave have completed; the register sets are threaded through \
       the asm so that nothing reading them can be scheduled above the wait */                                             \
#define W3_WAIT_S(N, S)                                                                                         \
    {                                                                                                           \
        if (COT == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(S[0]) : "n"(N) : "memory");                     \
        if (COT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(S[0]), "+v"(S[COT > 1 ? 1 : 0]) : "n"(N) : "memory"); \
        if (COT == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(S[0]), "+v"(S[COT > 1 ? 1 : 0]), "+v"(S[COT > 2 ? 2 : 0]) : "n"(N) : "memory"); \
    }
    /* unconditional, clamped raw loads of the patch of chunk `ch` (conv_wino.cpp: WR_LOAD_P) */
#define W3_LOAD_P(ch, D)                                                                                        \
    {                                                                                                           \
        const int cb = min((ch) * CK, Cin - 1);                                                                 \
        const int cmax = Cin - 1 - cb;                                                                          \
        const bool second = cb >= a.C0;                                                                         \
        const float* srcb = second ? a.x1 + ((long)b * a.C1 + (cb - a.C0)) * HW : a.x0 + ((long)b * a.C0 + cb) * HW; \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                                   \
            const unsigned off = (unsigned)(min((int)((p_pk[sl] >> 12) & (CK - 1)), cmax) * HW + (int)(p_pk[sl] >> 18)) * 4u; \
            asm volatile("global_load_dword %0, %1, %2" : "=v"(D[sl]) : "v"(off), "s"(srcb) : "memory");        \
        }                                                                                                       \
    }
#define W3_WAIT_P(N, D)                                                                                         \
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(D[0]), "+v"(D[1]), "+v"(D[2]), "+v"(D[3]), "+v"(D[4]), "+v"(D[5]) : "n"(N) : "memory");
    /* activate once per pixel (coefficients from the LDS table) and park the patch in LDS; zero padding applies AFTER     \
       the activation */                                                                                           \
#define W3_WRITE_P(ch, D)                                                                                        \
    {                                                                                                           \
        float* sPw = sP + (((ch) & 1) ? PBUF : 0);                                                              \
        const int nvalid = Cin - (ch) * CK;                                                                     \
        f32x2 cfv[MAXP];                     /* all coefficient reads first: ONE LDS round trip */               \
        if (PRO >= 1) {                                                                                         \
            _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                               \
                const int cch = min((ch) * CK + (int)((p_pk[sl] >> 12) & (CK - 1)), Cin - 1);                   \
                cfv[sl] = *reinterpret_cast<const f32x2*>(sCo + cch * 2);                                       \
            }                                                                                                   \
        }                                                                                                       \
        _Pragma("unroll") for (int sl = 0; sl < MAXP; ++sl) {                                                   \
            float v = D[sl];                                                                                    \
            if (PRO >= 1) v = v * cfv[sl].x + cfv[sl].y;                                                        \
            if (PRO >= 2) v = silu_w3(v);                                                                       \
            sPw[p_pk[sl] & 0xfff] = ((int)((p_pk[sl] >> 12) & 63) < min(nvalid, CK)) ? v : 0.0f;               \
        }                                                                                                       \
    }
    /* rows 2rg and 2rg+1 of B^T d for the two channels of the pair, (.) B, three-way split, 24 stores:                     \
       row 0: d0 - d2   row 1: d1 + d2   row 2: d2 - d1   row 3: d1 - d3;   (.) B: m0 - m2, m1 + m2, m2 - m1, m1 - m3 */      \
#define W3_WRITE_V(ch, RG)                                                                                      \
    {                                                                                                           \
        const float* sPr = sP + (((ch) & 1) ? PBUF : 0) + p_rd;                                                 \
        unsigned* vdst = sV + (((ch) & 1) ? VW : 0) + v_wr;                                                     \
        float mx[2][4], my[2][4];                                                                               \
        _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2) {                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
                const float r0 = sPr[k2 * 2 * 10 * PP + j], r1 = sPr[k2 * 2 * 10 * PP + PP + j],                \
                            r2 = sPr[k2 * 2 * 10 * PP + 2 * PP + j];                                            \
                if ((RG) == 0) { mx[k2][j] = r0 - r2; my[k2][j] = r1 + r2; }                                      \
                else { mx[k2][j] = r1 - r0; my[k2][j] = r0 - r2; }                                              \
            }                                                                                                   \
        }                                                                                                       \
        _Pragma("unroll") for (int row = 0; row < 2; ++row) {                                                   \
            float v[2][4];                                                                                      \
            _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2) {                                                  \
                const float m0 = row ? my[k2][0] : mx[k2][0], m1 = row ? my[k2][1] : mx[k2][1];                 \
                const float m2 = row ? my[k2][2] : mx[k2][2], m3 = row ? my[k2][3] : mx[k2][3];                 \
                v[k2][0] = m0 - m2; v[k2][1] = m1 + m2; v[k2][2] = m2 - m1; v[k2][3] = m1 - m3;                 \
            }                                                                                                   \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                     \
                unsigned w1, w2, w3;                                                                            \
                w3_split(v[0][q], v[1][q], w1, w2, w3);                                                         \
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
    /* slot j (0..7) of the A operand of cout sub-tile ct: unit j>>2, k-pair j&3, packed operand index kp*COT + ct */
#define W3_AVAL(i, ct, j) AW[i][(j) >> 2][(((j) & 3) * COT + (ct)) >> 2][(((j) & 3) * COT + (ct)) & 3]
    /* step st = (position st / COT, sub-tile st % COT): three-way split of its eight weights per lane -> D[piece] */
#define W3_SPLIT(st, D)                                                                                         \
    {                                                                                                           \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                         \
            unsigned w1, w2, w3;                                                                                \
            const float ax = W3_AVAL((st) / COT, (st) % COT, 2 * r), ay = W3_AVAL((st) / COT, (st) % COT, 2 * r + 1); \
            if (EXP & 64) { w1 = __builtin_bit_cast(unsigned, ax); w2 = __builtin_bit_cast(unsigned, ay); w3 = w1 ^ w2; } \
            else w3_split(ax, ay, w1, w2, w3);                                                                  \
            D[0][r] = w1; D[1][r] = w2; D[2][r] = w3;                                                           \
        }                                                                                                       \
    }
    /* the six piece products of step st, smallest first */
#define W3_MMA(st, D, BQ)                                                                                       \
    {                                                                                                           \
        if (!(EXP & 16)) {                                                                                      \
            f32x16 c = acc[(st) / COT][(st) % COT];                                                             \
            c = w3_mfma(D[0], BQ[2], c);                                                                        \
            c = w3_mfma(D[2], BQ[0], c);                                                                        \
            c = w3_mfma(D[1], BQ[1], c);                                                                        \
            c = w3_mfma(D[0], BQ[1], c);                                                                        \
            c = w3_mfma(D[1], BQ[0], c);                                                                        \
            c = w3_mfma(D[0], BQ[0], c);                                                                        \
            acc[(st) / COT][(st) % COT] = c;                                                                    \
        } else {                                                                                                \
            acc[(st) / COT][(st) % COT][0] += __builtin_bit_cast(float, D[0][0] ^ D[1][1] ^ D[2][2] ^ D[0][3] ^ BQ[0][0] ^ BQ[1][1] ^ BQ[2][2]); \
        }                                                                                                       \
    }
    /* all MFMAs of chunk `ch` (V(ch) in LDS): step st runs its six MFMAs while the VALU splits the weights of step st+1      \
       (scheduling fences between the steps keep the compiler from splitting everything up front: registers).  Weight traffic:  \
       position 2w+1's weights of THIS chunk are requested at the top and land under position 2w's MFMAs; position 2w's weights  \
       of the NEXT chunk are requested as soon as its last split has been issued -- so only one position's weights (2*COT        \
       registers x 4) are pinned while the wave runs its patch + transform phase.  WN = VMEM operations that may stay in flight \
       when position 2w's weights are needed (the loads issued after them) */                                                \
#define W3_MFMA_PHASE(ch, NEXT, WN)                                                                             \
    {                                                                                                           \
        const unsigned* sVc = sV + (((ch) & 1) ? VW : 0);                                                       \
        u32x4 bq[3], pc[2][3];                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if (!(EXP & 4)) {                                                                                       \
            W3_LOAD_A(ch, 1)                                                                                    \
            W3_WAIT_S(WN, AW[0][0]) W3_WAIT_S(WN, AW[0][1])                                                     \
        }                                                                                                       \
        if (!(EXP & 8)) W3_LOAD_B(0, bq)                                                                        \
        else { _Pragma("unroll") for (int p = 0; p < 3; ++p) bq[p] = u32x4{1, 2, 3, 4}; }                       \
        W3_SPLIT(0, pc[0])                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int st = 0; st < 2 * COT; ++st) {                                                \
            if (NEXT && !(EXP & 4) && st == COT - 1) W3_LOAD_A((ch) + 1, 0)                                     \
            if (st == COT && !(EXP & 8)) W3_LOAD_B(1, bq)      /* (behind the last MFMA of position 2w) */      \
            W3_MMA(st, pc[st & 1], bq)                                                                          \
            if (st == COT - 1 && !(EXP & 4)) { W3_WAIT_S(NEXT ? 2 * COT : 0, AW[1][0]) W3_WAIT_S(NEXT ? 2 * COT : 0, AW[1][1]) } \
            if (st + 1 < 2 * COT) W3_SPLIT(st + 1, pc[(st + 1) & 1])                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
        }                                                                                                       \
    }
    /* patch of chunk ch+2 -> LDS, raw patch of chunk ch+3 requested, V(ch+1) -> LDS */
#define W3_VALU_PHASE(ch, RG)                                                                                   \
    {                                                                                                           \
        if (!(EXP & 4)) W3_WAIT_P(NA, pd)                                                                       \
        if (!(EXP & 2)) W3_WRITE_P((ch) + 2, pd)                                                                \
        if (!(EXP & 4)) W3_LOAD_P((ch) + 3, pd)                                                                 \
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
    f32x4 AW[2][2][COT];                                // [position i][unit][float4]: weights of the chunk in flight / in use
    float pd[MAXP];                                     // raw patch registers, loaded one chunk ahead of their activation
    {
        float q0[MAXP], q1[MAXP];                       // patches of the first two chunks: prologue only
        f32x2 cfl = {1.0f, 0.0f};
        W3_LOAD_A(c_begin, 0)
        W3_LOAD_P(c_begin, q0)
        W3_LOAD_P(c_begin + 1, q1)
        W3_LOAD_P(c_begin + 2, pd)
        if (PRO) {
            for (int c = tid; c < Cin; c += NT) {
                if (a.coef) cfl = *reinterpret_cast<const f32x2*>(a.coef + ((long)b * Cin + c) * 2);
                *reinterpret_cast<f32x2*>(sCo + c * 2) = cfl;
            }
        }
        if (rec) pt[0] = __builtin_amdgcn_s_memtime() - tk0;           // index setup + load issue
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ONE memory latency for everything above
        W3_WAIT_S(0, AW[0][0]) W3_WAIT_S(0, AW[0][1])
        W3_WAIT_P(0, q0)
        W3_WAIT_P(0, q1)
        W3_WAIT_P(0, pd)
        if (rec) pt[1] = __builtin_amdgcn_s_memtime() - tk0;           // ... + memory latency
        if (PRO) __syncthreads();          // coefficient table visible
        W3_WRITE_P(c_begin, q0)
        W3_WRITE_P(c_begin + 1, q1)
    }
    __syncthreads();                       // the first two patches visible
    if (rec) pt[2] = __builtin_amdgcn_s_memtime() - tk0;
    W3_WRITE_V(c_begin, rg)
    __syncthreads();                       // V of the first chunk visible
    W3_STAMP(0)

    // ---- K loop.  VMEM issue order of a wave per chunk c (in-order vmcnt counter; nothing else is outstanding); C2 = 2*COT loads:
    //   waves 0-3:  [patch(c+3): MAXP] [weights(c) of position 2w+1: C2] [weights(c+1) of position 2w: C2]
    //   waves 4-7:  [weights(c) of position 2w+1: C2] [weights(c+1) of position 2w: C2] [patch(c+3): MAXP]
    // wait points (the same counts in both orders):
    //   patch(c+2) before its write: two weight groups were issued after it                                     vmcnt(NA)
    //   weights(c) of position 2w: the patch of the previous phase and this chunk's 2w+1 group came after       vmcnt(MAXP + C2)
    //   weights(c) of position 2w+1: the next chunk's 2w group came after                                       vmcnt(C2)
    const int ph = (EXP & 128) ? __builtin_amdgcn_readfirstlane(wave & 1) : rg;     // phase order of the wave
    // On leaving a loop the weights of the last chunk (position 2w) and, for waves 4-7, a stray patch prefetch are still in
    // flight.  W3_DRAIN waits for them INSIDE each branch, with every destination register as an operand: where the two loops
    // join the compiler reconciles their register assignments with copies, and a copy of a register whose load has not landed
    // yet copies garbage (and a register it considers dead -- the stray prefetch -- is reused and then overwritten).
#define W3_DRAIN { W3_WAIT_P(0, pd) W3_WAIT_S(0, AW[0][0]) W3_WAIT_S(0, AW[0][1]) }
    if (ph == 0) {
        for (int c = c_begin; c + 1 < c_end; ++c) {
            W3_VALU_PHASE(c, rg)
            W3_MFMA_PHASE(c, true, MAXP + 2 * COT)
            // chunk c read by every wave; V(c+1), patch(c+2) visible.  LDS traffic only: no VMEM wait at the barrier.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        W3_DRAIN
    } else {
        for (int c = c_begin; c + 1 < c_end; ++c) {
            W3_MFMA_PHASE(c, true, MAXP + 2 * COT)
            W3_VALU_PHASE(c, rg)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        W3_DRAIN
    }
#undef W3_DRAIN
    {
        const int c = c_end - 1;
        W3_MFMA_PHASE(c, false, 2 * COT)
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
#undef W3_LOAD_A2
#undef W3_SPLIT
#undef W3_MMA
#undef W3_WAIT_S
#undef W3_WAIT_A
#undef W3_LOAD_P
#undef W3_WAIT_P
#undef W3_WRITE_P
#undef W3_WRITE_V
#undef W3_LOAD_B
#undef W3_AVAL
#undef W3_MFMA_PHASE
#undef W3_VALU_PHASE
}

static size_t wino3_lds_bytes(int Cin) {
    return (size_t)(2 * W3_VW + 2 * (W3_CK * 10 * W3_PP + 4) + 2 * Cin) * sizeof(float);
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
            case 132: rc = wino3_launch_k<3, 2, 132>(k, grid, lds, s); break;    // ... without VMEM
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
