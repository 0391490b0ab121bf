// 3x3 convolution by Winograd F(2x2, 3x3) with the channel contraction on the BF16 matrix pipe AT FP32 ACCURACY and with the
// full fp32 exponent range: both MFMA operands are split, exactly, into three bf16 pieces  v = v1 + v2 + v3  (round-to-nearest at
// every level; each remainder is exact in fp32 and the third one has at most 8 significant bits, so the three pieces carry all 24
// bits of the operand), and the product is accumulated in fp32 from the six piece products of weight >= 2^-16:
//     u * v  ~=  u1 v3 + u3 v1 + u2 v2 + u1 v2 + u2 v1 + u1 v1          (dropped: u2 v3 + u3 v2 + u3 v3 <= 2^-23.4 |u v|)
// Each piece product is exact in the fp32 accumulator (8 x 8 significant bits), so the result differs from the fp32-MFMA kernel
// (conv_wino.cpp) by less than one fp32 rounding per product.  bf16 has the exponent range of fp32: nothing is scaled, nothing is
// clamped, Inf / NaN propagate.  (The split is bit-exact for 2^-110 <= |v|; below, the third piece reaches the bf16 denormals, ulp
// 2^-133, and the operand loses bits gradually -- tests/test_bf16x3_arithmetic_cpu.py.)  tests/test_gpu_parity.py holds this kernel to the fixtures of the fp32 kernels and measures it
// against an fp64 convolution (random and structured inputs, 1e-6 ... 1e5 magnitudes).
//
// Round 2's form of this kernel split the WEIGHTS on the fly and was VALU-bound on exactly that (240 of ~430 VALU instructions per
// wave and chunk, profiles/r02_wino3_kloop.txt).  Here the weight pieces are split ONCE, when the weights are packed
// (pack_wino3_weight_kernel at mcvd_model_finalize: three bf16 planes in MFMA A-operand order, 6 bytes per transformed weight), as
// conv_wino2h.cpp does for its two fp16 pieces; the K loop carries no VALU work for the A operand.  v_mfma_f32_32x32x16_bf16
// retires 16 channels x 32 x 32 in 32 cycles: six of them cost 3/8 of the fp32 pipe time.
//
// Same decomposition and machine mapping as conv_wino2h.cpp (region of 8 x 16 output pixels = 32 tiles or two 8x8 images, 32*COT
// output channels, all 16 transform positions, 16 input channels per chunk; 512 threads = 8 waves = two waves per SIMD, wave w
// owns positions 2w and 2w+1):
//   * weights: [cout tile][chunk][position][cout sub-tile][piece][64 lanes][4 dwords], a dword = two bf16 = K slots (2j, 2j+1)
//     of the lane's half; the 6*COT quads a wave needs per chunk are one contiguous block, fetched with global_load_dwordx4
//     STRAIGHT INTO THE MFMA A-OPERAND REGISTERS (named registers the compiler does not allocate, v184-v255: see W3_LOAD_A) and
//     reloaded for the next chunk as soon as the position's MFMAs have been issued: prefetch distance = one chunk.
//   * K-slot convention of the 32x32x16 MFMA (both operands): lane half h, element e  <->  channel 2e + h of the chunk.
//   * activations: raw patch fetched four pixels per lane (global_load_dwordx4) + one halo pixel, activated and parked in LDS with the
//     channels of a pair interleaved; the transform runs on packed fp32, splits three ways
//     (v_cvt_pk_bf16_f32, expand, v_pk_add_f32 per level: 9 VALU per channel pair and position) and parks three bf16 planes
//     [piece][position][k half][k pair][tile].
//   * the two waves of a SIMD run the chunk in opposite orders (patch + transform | MFMAs), one barrier per chunk.
//   * the MFMAs are inline asm (their A operand is a named register); the 6 * COT MFMAs of a position are ordered product-major,
//     smallest product first, so consecutive MFMAs target different accumulators.
// VMEM of the K loop is hand-counted (inline asm loads + s_waitcnt vmcnt(N)) exactly as in conv_wino2h.cpp; tools/check_wino_isa.py
// checks the generated code of this file too.
#include <stdlib.h>

#include "../common.h"

namespace mcvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_w3(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

constexpr int W3_CK = 16;        // input channels per chunk = K of one bf16 MFMA
constexpr int W3_T = 32;         // tiles per workgroup (4 x 8 tiles = 8 x 16 output pixels)
constexpr int W3_NT = 512;
#ifndef MCVD_W3_PP
#define MCVD_W3_PP 24
#endif
// LDS patch row pitch in channel-pair columns (18 used; 20 for the two 8x8 images side by side).  The patch park stores a lane's four pixels as
// single dwords at ((pair * 10 + row) * PP + col) * 2 + ce; the 32 lanes of a store group are 8 rows x 4 four-pixel items, bank = (row * 2 PP +
// 8 item) mod 32: with 2 PP = 48 = 16 (mod 32) they fall on FOUR banks -- the "x4 patch park" conflict behind 27-29 percent of the LDS-active
// cycles (profiles/r04_pmc_sq_wave_states.txt).  Round 5 built the conflict-free pitch (25: 2 PP = 18 mod 32, 16 banks, 2-way = free for
// ds_write_b32) and measured it against 24 on one box (tools/build_variant.sh, -DMCVD_W3_PP=25; profiles/r05_patch_pitch_ab.txt): 5577 vs 5592
// cycles per chunk, 248.1 / 248.5 vs 248.8 / 249.4 frames/s -- nothing.  The stores are not on the chunk's critical path; 24 stays.
constexpr int W3_PP = MCVD_W3_PP;
constexpr int W3_PW = 16 * 2 * 4 * W3_T;          // 32-bit words of one piece plane of a V chunk: [position][half][pair][tile]
constexpr int W3_VW = 3 * W3_PW;                  // 32-bit words of one V chunk: [piece][position][half][pair][tile]

// (lo, hi) -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned w3_cvt_pk(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// exact three-way split of an adjacent register pair into packed bf16 pairs (w1 the leading pieces): the two subtractions of a
// level are one v_pk_add_f32
__device__ __forceinline__ void w3_split3(f32x2 v, unsigned& w1, unsigned& w2, unsigned& w3) {
    w1 = w3_cvt_pk(v.x, v.y);
    const f32x2 h = {__builtin_bit_cast(float, w1 << 16), __builtin_bit_cast(float, w1 & 0xffff0000u)};
    v = v - h;
    w2 = w3_cvt_pk(v.x, v.y);
    const f32x2 g = {__builtin_bit_cast(float, w2 << 16), __builtin_bit_cast(float, w2 & 0xffff0000u)};
    v = v - g;
    w3 = w3_cvt_pk(v.x, v.y);
}

// PRO: 0 raw input, 1 affine, 2 affine + SiLU (the GroupNorm / temb prologue of conv_wino.cpp)
// a.ksplit == 2 / 4 / 8 (grid.y = that many): one part of the input channels per workgroup, raw partial result to a.part[part]; the reduce
//     pass (conv_wino.cpp) sums the parts in index order.
// EXP != 0 (built with -DMCVD_DIAG only): timing-only ablations of the K loop (wrong results; env MCVD_WINO3_EXP, tests/gpu_diag.py
//     w3exp): bit 0 no tile transform, bit 1 no patch activation/park, bit 2 no VMEM in the loop, bit 3 no B-operand reads, bit 4 no MFMA.
// G8: 8x8 images -- the 32 tiles of a workgroup are TWO whole images (16 tiles each, image i at patch columns 10 i .. 10 i + 9); every
//     halo element is zero padding, so only the 2 x 64 interior pixels per channel are loaded (slots 0-3 of the six; the other two
//     fetch a dummy) and the halo of both patch buffers is zeroed once.  The coefficient table holds both samples.
template <int COT, int PRO, bool G8, int EXP = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(86))) void conv_wino3_kernel(ConvArgs a) {
    // amdgpu_num_vgpr(86): the compiler may allocate v0-v171; v172-v255 hold the in-flight loads and the A operands (W3_LOAD_A).
    // On gfx90a+ LLVM DOUBLES the requested number (unified VGPR + AGPR file) before it checks it against the occupancy bound and
    // drops it silently when the doubled value exceeds 256: amdgpu_num_vgpr(172) would be ignored, amdgpu_num_vgpr(86) caps the
    // allocator at 172 registers (this kernel uses no AGPRs).  tools/check_wino_isa.py verifies the outcome on the generated code.
    constexpr int NT = W3_NT, CK = W3_CK, T = W3_T, BCO = 32 * COT, PP = W3_PP, VW = W3_VW, PW = W3_PW;
    constexpr int PSZ = CK * 10 * PP;           // activated input patch of one chunk: [CK][10 rows][PP]
    constexpr int PBUF = PSZ + 8;               // + dump space for unused patch slots (four floats, two apart)
    // patch-load slots per thread and chunk.  A row of the patch is 16 interior pixels (64-byte aligned in the image: four 16-byte loads)
    // plus one halo pixel on each side: 16 channels x 10 rows x 4 = 640 four-pixel items (slot 0: item tid; slot 1: item 512 + tid for
    // tid < 128) and 16 x 10 x 2 = 320 halo items (slot 2: item tid for tid < 320).  Round 3's first form loaded 2880 single pixels in six
    // dword slots: 48 wave instructions per chunk where this takes 15, and the vector memory path -- 64 bytes per clock and CU, which the
    // weight stream alone keeps busy 2304 of a chunk's cycles -- had 768 more cycles of work per chunk (profiles/r03_wino3_kloop.txt).
    // G8: 16 channels x 2 images x 8 rows x 2 = 512 four-pixel items, one per thread, no halo loads (the halo is zero padding).
    constexpr int NPL = G8 ? 1 : 3;                             // load instructions per thread and chunk
    constexpr int NPV = G8 ? 4 : 9;                             // values they bring
    constexpr int NQ = 3 * COT;                                 // weight quads per position: COT cout sub-tiles x 3 pieces
    constexpr int NA = 2 * NQ;                                  // weight loads per wave and chunk
    constexpr int VM_A = NQ + NPL;                              // see W3_MFMA_PHASE
    constexpr int PQ = 5;                                       // weight quads whose registers hold the first two patches in the prologue
    static_assert(NA <= 18 && NA > PQ, "named-register map below: v172-v180 patch, v184-v255 eighteen weight quads");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sV = reinterpret_cast<unsigned*>(smem);           // [2][VW]
    float* sP = smem + 2 * VW;                  // [2][PBUF]
    float* sCo = sP + 2 * PBUF;                 // [Cin][2] prologue coefficients (A_c, B_c) of this sample (PRO only; G8: [2][Cin][2])
    unsigned* sOff = reinterpret_cast<unsigned*>(sCo + (G8 ? 4 : 2) * a.Cin);      // [NPL][NT] byte offsets of the patch-load slots (read by their owner only)

    {   // the kernel descriptor must allocate all 256 registers: the asm statements below name v172-v255 in their text only
        float top;
        asm volatile("" : "={v255}"(top));
    }
    // every kernel argument the prologue needs, fetched NOW (one batch of scalar loads, one wait): left to itself the compiler fetches
    // each where it is first used, eight dependent scalar-cache round trips in front of the first patch request
    asm volatile("" :: "s"(a.x0), "s"(a.x1), "s"(a.coef), "s"(a.wpb), "s"(a.B), "s"(a.H), "s"(a.W), "s"(a.Cin), "s"(a.CinP), "s"(a.C0),
                 "s"(a.C1), "s"(a.CoutP), "s"(a.ksplit), "s"(a.dbg), "s"(a.wdma));
    const unsigned long long t_start = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;     // diagnostics: the phase clock starts here
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
    // sample; G8: also the region's second sample, clamped to the last one), fetched by asm loads into v172-v179 IN FRONT of the first
    // patches (see the prologue below) and parked in the LDS table once they have landed
    const float* co_src[G8 ? 4 : 2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        co_src[k] = a.coef + ((long)b * Cin + min(tid + k * NT, Cin - 1)) * 2;
        if (G8) co_src[2 + k] = a.coef + ((long)min(b + 1, a.B - 1) * Cin + min(tid + k * NT, Cin - 1)) * 2;
    }
    if (PRO) {                             // unconditional (clamped) loads into the registers of the third patch (requested later); the
                                           // oldest VMEM operations of the wave: their latency passes under the index arithmetic below
        asm volatile("global_load_dwordx2 v[172:173], %0, off\n\tglobal_load_dwordx2 v[174:175], %1, off"
                     :: "v"(co_src[0]), "v"(co_src[1]) : "memory");
        if constexpr (G8)
            asm volatile("global_load_dwordx2 v[176:177], %0, off\n\tglobal_load_dwordx2 v[178:179], %1, off"
                         :: "v"(co_src[2]), "v"(co_src[3]) : "memory");
    }

    // ---- transform role: (channel pair, tile) = tid & 255.  Pair s_cp = channels (s_ca, s_ca + 2), s_ca = 4*(s_cp >> 1) + (s_cp & 1):
    //      the low and high bf16 of word (k half s_cp & 1, k pair s_cp >> 1) of the B operand.
    const int s_tile = tid & 31, s_cp = (tid & 255) >> 5;
    const int s_ty = G8 ? (s_tile >> 2) & 3 : s_tile >> 3, s_tx = G8 ? (s_tile & 3) + 5 * (s_tile >> 4) : s_tile & 7;
    // LDS patch: [pair 8][10 rows][PP columns][2 channels] floats.  Rows rg, rg+1, rg+2 of the tile's 4x4 window:
    const int p_rd = ((s_cp * 10 + 2 * s_ty + rg) * PP + 2 * s_tx) * 2;
    // word of (piece 0, position 8*rg, half, pair, tile); one position further = 256 words, one piece = PW
    const int v_wr = ((8 * rg * 2 + (s_cp & 1)) * 4 + (s_cp >> 1)) * T + s_tile;

    // ---- patch-load slots (chunk invariant): p_pk = LDS float index of the slot's FIRST element (12 bits; a four-pixel item continues at
    // + 2, + 4, + 6: the patch interleaves the two channels of a pair) | channel code << 12, code = channel in chunk, + CK when the slot is
    // padding / unused (| G8: image of the region << 20); sOff[sl][tid] = byte offset of the (clamped) first pixel from the chunk's first
    // channel plane (parked in LDS: registers the MFMA phase needs more)
    unsigned p_pk[NPL];
#pragma unroll
    for (int sl = 0; sl < NPL; ++sl) {
        p_pk[sl] = (unsigned)PSZ | ((unsigned)CK << 12);        // unused: a dummy load, parked in the dump space
        unsigned off = 0;
        if (G8) {                               // item tid -> (channel, image, row, half row) of interior pixels
            const int e = tid, ci = e >> 5, img = (e >> 4) & 1, r = (e >> 1) & 7, c = (e & 1) * 4;
            const bool valid = b + img < a.B;
            const int cp = (ci >> 2) * 2 + (ci & 1), ce = (ci >> 1) & 1;
            p_pk[sl] = (unsigned)(((cp * 10 + r + 1) * PP + img * 10 + c + 1) * 2 + ce) | ((unsigned)(ci + (valid ? 0 : CK)) << 12) |
                       ((unsigned)(valid ? img : 0) << 20);
            off = (unsigned)(ci * HW + r * 8 + c) * 4u;
        } else if (sl < 2) {                    // four interior pixels of a row
            const int e = sl * NT + tid;
            if (e < CK * 40) {
                const int ci = e / 40, rem = e - ci * 40, r = rem >> 2, c = (rem & 3) * 4;
                const int y = oy0 - 1 + r;
                const bool inside = y >= 0 && y < H;
                const int cp = (ci >> 2) * 2 + (ci & 1), ce = (ci >> 1) & 1;
                p_pk[sl] = (unsigned)(((cp * 10 + r) * PP + c + 1) * 2 + ce) | ((unsigned)(ci + (inside ? 0 : CK)) << 12);
                off = (unsigned)(ci * HW + min(max(y, 0), H - 1) * W + ox0 + c) * 4u;
            }
        } else if (tid < CK * 20) {             // one halo pixel: left (column 0) or right (column 17) of a row
            const int e = tid, ci = e / 20, rem = e - ci * 20, r = rem >> 1, c = (rem & 1) * 17;
            const int y = oy0 - 1 + r, x = ox0 - 1 + c;
            const bool inside = y >= 0 && y < H && x >= 0 && x < W;
            const int cp = (ci >> 2) * 2 + (ci & 1), ce = (ci >> 1) & 1;
            p_pk[sl] = (unsigned)(((cp * 10 + r) * PP + c) * 2 + ce) | ((unsigned)(ci + (inside ? 0 : CK)) << 12);
            off = (unsigned)(ci * HW + min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)) * 4u;
        }
        sOff[sl * NT + tid] = off;
    }

    // ---- weight fetch: the NA quads of a wave per chunk are contiguous: quad q = (i * COT + ct) * 3 + piece of positions 2w + i at
    //      wr_base + chunk * (16*NQ*256) + q * 256 + lane * 4   dwords
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned* wr_base = reinterpret_cast<const unsigned*>(a.wpb) + ((long)cotile * (a.CinP / CK) * 16 + 2 * wave_u) * (NQ * 256);
    const unsigned wr_voff = (unsigned)lane * 16u;

    /* IN-FLIGHT DATA LIVES IN REGISTERS THE COMPILER DOES NOT ALLOCATE.  The kernel is compiled with amdgpu_num_vgpr(86): v172-v255 are
       never touched by generated code.  The asm loads write them (weight quad q: v[184 + 4q : 187 + 4q]; patch slots: v[172:175], v[176:179], v180), the
       waits are bare s_waitcnt, the MFMAs name their A operand in the instruction text.  (Loads whose results are compiler-visible
       values are not safe here: the register allocator may assign the result and the operand of the later wait to different registers
       and copy between them while the load is still in flight -- it did, a wrong result once in ~10^4 launches.)
       Every statement that touches a named register is `asm volatile` (program order among them is kept) except the patch FMAs, which
       take the wait's token (an SGPR) as an operand. */
#define W3_QUADS(X, q, A1, A2) X(0, "v[184:187]", q, A1, A2) X(1, "v[188:191]", q, A1, A2) X(2, "v[192:195]", q, A1, A2) X(3, "v[196:199]", q, A1, A2) X(4, "v[200:203]", q, A1, A2) X(5, "v[204:207]", q, A1, A2) X(6, "v[208:211]", q, A1, A2) X(7, "v[212:215]", q, A1, A2) X(8, "v[216:219]", q, A1, A2) X(9, "v[220:223]", q, A1, A2) X(10, "v[224:227]", q, A1, A2) X(11, "v[228:231]", q, A1, A2) X(12, "v[232:235]", q, A1, A2) X(13, "v[236:239]", q, A1, A2) X(14, "v[240:243]", q, A1, A2) X(15, "v[244:247]", q, A1, A2) X(16, "v[248:251]", q, A1, A2) X(17, "v[252:255]", q, A1, A2)
#define W3_LD1(K, R, q, P, UNUSED) if ((q) == K) asm volatile("global_load_dwordx4 " R ", %0, %1" :: "v"(wr_voff), "s"(P) : "memory");
#define W3_MF1(K, R, q, ACC, BV) if ((q) == K) asm volatile("v_mfma_f32_32x32x16_bf16 %0, " R ", %1, %0" : "+v"(ACC) : "v"(BV));
    /* the NQ weight quads of position 2w + i of chunk `ch` */
#define W3_LOAD_A(ch, i)                                                                                        \
    {                                                                                                           \
        const unsigned* ua = wr_base + (long)(ch) * (16 * NQ * 256) + (i) * (NQ * 256);                         \
        _Pragma("unroll") for (int qq = 0; qq < NQ; ++qq) { W3_QUADS(W3_LD1, (i) * NQ + qq, ua + qq * 256, 0) } \
    }
    /* weight piece `pc` of every cout sub-tile of position 2w + i of chunk `ch` (quads (i * COT + ct) * 3 + pc) */            \
#define W3_LOAD_A_PIECE(ch, i, pc)                                                                              \
    {                                                                                                           \
        const unsigned* ua = wr_base + (long)(ch) * (16 * NQ * 256) + (i) * (NQ * 256);                         \
        _Pragma("unroll") for (int ct = 0; ct < COT; ++ct) { W3_QUADS(W3_LD1, (i) * NQ + ct * 3 + (pc), ua + (ct * 3 + (pc)) * 256, 0) } \
    }
    /* prologue: quads Q0 .. Q1-1 of chunk `ch`, issued behind the instructions that produced DEP (which read the registers) */
#define W3_LD1D(K, R, q, P, DEP) if ((q) == K) asm volatile("global_load_dwordx4 " R ", %0, %1" :: "v"(wr_voff), "s"(P), "v"(DEP) : "memory");
#define W3_LOAD_A_RANGE(ch, Q0, Q1, DEP)                                                                        \
    {                                                                                                           \
        const unsigned* ua = wr_base + (long)(ch) * (16 * NQ * 256);                                            \
        _Pragma("unroll") for (int qq = (Q0); qq < (Q1); ++qq) { W3_QUADS(W3_LD1D, qq, ua + qq * 256, DEP) }    \
    }
#define W3_WAIT(N) asm volatile("s_waitcnt vmcnt(%1)\n\ts_mov_b32 %0, 0" : "=s"(vtok) : "n"(N) : "memory");
    /* unconditional, clamped raw loads of the patch of chunk `ch`: RA / RB = the two four-pixel slots, RH = the halo slot; DEP = a value
       computed from the previous contents of those registers (ordering) */
#define W3_READ_OFF(OFS) { _Pragma("unroll") for (int sl = 0; sl < NPL; ++sl) OFS[sl] = sOff[sl * NT + tid]; }
#define W3_LOAD_P(ch, DEP, OFS) W3_LOAD_PR(ch, DEP, OFS, "v[172:175]", "v[176:179]", "v180")
#define W3_LOAD_PR(ch, DEP, OFS, RA, RB, RH)                                                                    \
    {                                                                                                           \
        const int cb = min((ch) * CK, Cin - 1);                                                                 \
        const unsigned lim4 = (unsigned)((Cin - cb) * HW - 4) * 4u, lim1 = (unsigned)((Cin - cb) * HW - 1) * 4u; \
        const bool second = cb >= a.C0;                                                                         \
        const float* srcb = second ? a.x1 + ((long)b * a.C1 + (cb - a.C0)) * HW : a.x0 + ((long)b * a.C0 + cb) * HW; \
        const unsigned istride = (unsigned)((second ? a.C1 : a.C0) * HW) * 4u;      /* G8: distance to the region's second sample */ \
        /* channels past the last one are zeroed at the write: any (aligned) address inside the source will do */ \
        if constexpr (G8) {                                                                                     \
            const unsigned o0 = min(OFS[0], lim4) + ((p_pk[0] >> 20) & 1u) * istride;                           \
            asm volatile("global_load_dwordx4 " RA ", %0, %1" :: "v"(o0), "s"(srcb), "v"(DEP) : "memory");     \
        } else {                                                                                                \
            const unsigned o0 = min(OFS[0], lim4), o1 = min(OFS[NPL > 1 ? 1 : 0], lim4), o2 = min(OFS[NPL > 2 ? 2 : 0], lim1); \
            asm volatile("global_load_dwordx4 " RA ", %0, %3\n\tglobal_load_dwordx4 " RB ", %1, %3\n\tglobal_load_dword " RH ", %2, %3" \
                         :: "v"(o0), "v"(o1), "v"(o2), "s"(srcb), "v"(DEP) : "memory");                           \
        }                                                                                                       \
    }
    /* activate once per pixel (coefficients from the LDS table: one pair per slot, a slot is one channel) and park the patch in LDS;  \
       zero padding applies AFTER the activation */                                                                \
#define W3_READ_C(ch, cfv)                                                                                      \
    {                                                                                                           \
        _Pragma("unroll") for (int sl = 0; sl < NPL; ++sl) {                                                    \
            cfv[sl] = f32x2{1.0f, 0.0f};                                                                        \
            if (PRO >= 1) {                                                                                     \
                const int cch = min((ch) * CK + (int)((p_pk[sl] >> 12) & (CK - 1)), Cin - 1) + (G8 ? (int)((p_pk[sl] >> 20) & 1u) * Cin : 0); \
                cfv[sl] = *reinterpret_cast<const f32x2*>(sCo + cch * 2);                                       \
            }                                                                                                   \
        }                                                                                                       \
    }
#define W3_NOHOOK(e, v)
#define W3_WRITE_P(ch, PV, cfv) W3_WRITE_PR(ch, PV, cfv, "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", W3_NOHOOK)
    /* HOOK(e, v): statements placed behind value e (the prologue issues its weight loads there, one at a time) */ \
#define W3_WRITE_PR(ch, PV, cfv, A0, A1, A2, A3, B0, B1, B2, B3, H0, HOOK)                                      \
    {                                                                                                           \
        float* sPw = sP + (((ch) & 1) ? PBUF : 0);                                                              \
        const int nvalid = Cin - (ch) * CK;                                                                     \
        /* v = A * raw + B straight out of the patch registers (PRO 0: A = 1, B = 0, exact): the only reads of those registers */ \
        if constexpr (G8) {                                                                                     \
            asm("v_fma_f32 %0, " A0 ", %4, %5\n\tv_fma_f32 %1, " A1 ", %4, %5\n\tv_fma_f32 %2, " A2 ", %4, %5\n\tv_fma_f32 %3, " A3 ", %4, %5" \
                : "=&v"(PV[0]), "=&v"(PV[1]), "=&v"(PV[2]), "=&v"(PV[3]) : "v"(cfv[0].x), "v"(cfv[0].y), "s"(vtok)); \
        } else {                                                                                                \
            asm("v_fma_f32 %0, " A0 ", %9, %10\n\tv_fma_f32 %1, " A1 ", %9, %10\n\tv_fma_f32 %2, " A2 ", %9, %10\n\tv_fma_f32 %3, " A3 ", %9, %10\n\t" \
                "v_fma_f32 %4, " B0 ", %11, %12\n\tv_fma_f32 %5, " B1 ", %11, %12\n\tv_fma_f32 %6, " B2 ", %11, %12\n\tv_fma_f32 %7, " B3 ", %11, %12\n\t" \
                "v_fma_f32 %8, " H0 ", %13, %14"                                                                  \
                : "=&v"(PV[0]), "=&v"(PV[1]), "=&v"(PV[2]), "=&v"(PV[3]), "=&v"(PV[4]), "=&v"(PV[5]), "=&v"(PV[6]), "=&v"(PV[7]), "=&v"(PV[NPV > 8 ? 8 : 0]) \
                : "v"(cfv[0].x), "v"(cfv[0].y), "v"(cfv[NPL > 1 ? 1 : 0].x), "v"(cfv[NPL > 1 ? 1 : 0].y), "v"(cfv[NPL > 2 ? 2 : 0].x), \
                  "v"(cfv[NPL > 2 ? 2 : 0].y), "s"(vtok));                                                        \
        }                                                                                                       \
        _Pragma("unroll") for (int e = 0; e < NPV; ++e) {                                                       \
            const int sl = e >> 2;                            /* values 0-3: slot 0, 4-7: slot 1, 8: slot 2 */   \
            float v = PV[e];                                                                                    \
            if (PRO >= 2) v = silu_w3(v);                                                                       \
            sPw[(p_pk[sl] & 0xfff) + 2 * (e & 3)] = ((int)((p_pk[sl] >> 12) & 0xff) < min(nvalid, CK)) ? v : 0.0f; \
            HOOK(e, v)                                                                                          \
        }                                                                                                       \
    }
    /* rows 2rg and 2rg+1 of B^T d for the two channels of the pair (packed fp32: .x = channel s_ca, .y = s_ca + 2), (.) B,      \
       three-way bf16 split, 24 stores:                                                                                      \
       row 0: d0 - d2   row 1: d1 + d2   row 2: d2 - d1   row 3: d1 - d3;   (.) B: m0 - m2, m1 + m2, m2 - m1, m1 - m3 */      \
#define W3_READ_R(ch, RW)                                                                                       \
    {                                                                                                           \
        const f32x2* sPr = reinterpret_cast<const f32x2*>(sP + (((ch) & 1) ? PBUF : 0) + p_rd);                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { RW[0][j] = sPr[j]; RW[1][j] = sPr[PP + j]; RW[2][j] = sPr[2 * PP + j]; } \
    }
#define W3_WRITE_V(ch, RG, RW)                                                                                  \
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
                unsigned w1, w2, w3;                                                                            \
                w3_split3(q == 0 ? v0 : q == 1 ? v1 : q == 2 ? v2 : v3, w1, w2, w3);                            \
                vdst[(row * 4 + q) * 256] = w1;                                                                 \
                vdst[(row * 4 + q) * 256 + PW] = w2;                                                            \
                vdst[(row * 4 + q) * 256 + 2 * PW] = w3;                                                        \
            }                                                                                                   \
        }                                                                                                       \
    }
    /* B operand of position 2w+i -> BQ[piece][pair]  <-  word (((p*16 + pos)*2 + half)*4 + jp)*T + l31 */
#define W3_LOAD_B(i, BQ)                                                                                        \
    {                                                                                                           \
        const unsigned* q = sVc + (((2 * wave + (i)) * 2 + half) * 4) * T + l31;                                \
        _Pragma("unroll") for (int jp = 0; jp < 4; ++jp) {                                                      \
            BQ[0][jp] = q[jp * T]; BQ[1][jp] = q[PW + jp * T]; BQ[2][jp] = q[2 * PW + jp * T];                  \
        }                                                                                                       \
    }
    /* the COT MFMAs of one piece product (weight piece PA x activation piece PB) of position 2w + i */
#define W3_PRODUCT(i, PA, PB)                                                                                   \
    { _Pragma("unroll") for (int ct = 0; ct < COT; ++ct) { W3_QUADS(W3_MF1, 3 * ((i) * COT + ct) + (PA), acc[i][ct], bq[i][PB]) } }
    /* all MFMAs of chunk `ch` (V(ch) in LDS, weights(ch) in the named registers).  Per position 2w + i: wait for its NQ quads, 6*COT  \
       MFMAs ordered product-major (u3 v1, u2 v2, u1 v3, u2 v1, u1 v2, u1 v1: smallest class first; consecutive MFMAs write different \
       accumulators), and -- NEXT -- the same NQ quads are reloaded for chunk ch+1 piece by piece (the matrix pipe has read its A operands by the     \
       time the wave gets past the MFMA: it issues in order).  In-order VMEM bookkeeping: when the quads of a position are needed, the \
       loads issued after them are the other position's NQ quads and one patch group: vmcnt(NQ + NPL), in both phase orders. */      \
#define W3_MFMA_PHASE(ch, NEXT)                                                                                 \
    {                                                                                                           \
        const unsigned* sVc = sV + (((ch) & 1) ? VW : 0);                                                       \
        u32x4 bq[2][3];                                                                                         \
        W3_TS(7)                                                                                                \
        if (!(EXP & 8)) { W3_LOAD_B(0, bq[0]) W3_LOAD_B(1, bq[1]) }                                             \
        else { _Pragma("unroll") for (int p = 0; p < 3; ++p) { bq[0][p] = u32x4{1, 2, 3, 4}; bq[1][p] = u32x4{5, 6, 7, 8}; } } \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                         \
            if (NEXT && !(EXP & 4)) W3_WAIT(VM_A)                                                               \
            if (!(EXP & 16)) {                                                                                  \
                /* a weight piece is re-requested for chunk ch+1 as soon as its LAST product has been issued: u3 behind the first    \
                   product, u2 behind the fourth, u1 behind the sixth -- three requests at a time, spread over the position's 576   \
                   cycles of matrix work (nine in one burst stall the wave in the issue: the CU's vector-memory path is 70 % busy) */ \
                W3_PRODUCT(i, 2, 0)                                                                             \
                if (NEXT && !(EXP & 4)) W3_LOAD_A_PIECE((ch) + 1, i, 2)                                         \
                W3_PRODUCT(i, 1, 1) W3_PRODUCT(i, 0, 2) W3_PRODUCT(i, 1, 0)                                     \
                if (NEXT && !(EXP & 4)) W3_LOAD_A_PIECE((ch) + 1, i, 1)                                         \
                W3_PRODUCT(i, 0, 1) W3_PRODUCT(i, 0, 0)                                                         \
                if (NEXT && !(EXP & 4)) W3_LOAD_A_PIECE((ch) + 1, i, 0)                                         \
            } else {                                                                                            \
                _Pragma("unroll") for (int ct = 0; ct < COT; ++ct)                                              \
                    acc[i][ct][0] += __builtin_bit_cast(float, bq[i][0][0] ^ bq[i][1][1] ^ bq[i][2][2] ^ bq[i][1][3]); \
                if (NEXT && !(EXP & 4)) W3_LOAD_A((ch) + 1, i)                                                  \
            }                                                                                                   \
            W3_TS(4 + i)                                                                                        \
        }                                                                                                       \
    }
    /* patch of chunk ch+2 -> LDS, raw patch of chunk ch+3 requested, V(ch+1) -> LDS.  Two LDS round trips (coefficients + load offsets,   \
       then the 12 patch pairs): reading everything up front as conv_wino2h.cpp does needs more registers than the 178 - 96 this   \
       kernel has beside its accumulators */                                                                                     \
#define W3_VALU_PHASE(ch, RG)                                                                                   \
    {                                                                                                           \
        {                                                                                                       \
            f32x2 cfv[NPL];                                                                                     \
            unsigned ofs[NPL];                                                                                  \
            W3_TS(7)                                                                                            \
            if (!(EXP & 2)) W3_READ_C((ch) + 2, cfv)                                                            \
            if (!(EXP & 4)) W3_READ_OFF(ofs)                                                                    \
            if (!(EXP & 4)) W3_WAIT(NA)                                                                         \
            W3_TS(0)                                                                                            \
            float pv[NPV];                                                                                      \
            _Pragma("unroll") for (int e = 0; e < NPV; ++e) pv[e] = 0.0f;                                       \
            if (!(EXP & 2)) W3_WRITE_P((ch) + 2, pv, cfv)                                                       \
            W3_TS(1)                                                                                            \
            if (!(EXP & 4)) W3_LOAD_P((ch) + 3, pv[0], ofs)                                                     \
            W3_TS(2)                                                                                            \
        }                                                                                                       \
        if (!(EXP & 1)) {                                                                                       \
            f32x2 rw[3][4];                                                                                     \
            W3_READ_R((ch) + 1, rw)                                                                             \
            W3_WRITE_V((ch) + 1, RG, rw)                                                                        \
        }                                                                                                       \
        W3_TS(3)                                                                                                \
    }

    f32x16 acc[2][COT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][ct][r] = 0.0f;

    // diagnostics (mcvd_ctx_set_debug_buffer): shader-clock time the wave a.wdma spends per phase
    const bool rec = a.dbg != nullptr && wave == (a.wdma & 7);
    const bool sub = (a.wdma & 64) != 0;           // record prologue / epilogue sub-phase stamps instead of the wall clock
    unsigned long long sp[5] = {0, 0, 0, 0, 0};
    unsigned long long tk0 = 0, tprev = 0, dt[2] = {0, 0}, rt0 = 0;
    if (rec) {
        rt0 = __builtin_amdgcn_s_memrealtime();          // constant 100 MHz: start / end of the workgroup on the wall clock
        tk0 = tprev = t_start;
    }
#define W3_STAMP(i)                                                                                             \
    if (rec) {                                                                                                  \
        const unsigned long long now = __builtin_amdgcn_s_memtime();                                            \
        dt[i] += now - tprev;                                                                                   \
        tprev = now;                                                                                            \
    }
#ifdef MCVD_DIAG
    /* diagnostics build, MCVD_DBG_WAVE >= 128: cycles the recording wave spends per sub-phase of the K loop (slot 7 = discarded: the
       time up to the phase start; 0 C/OFF reads + patch wait, 1 activation + park, 2 patch-load issue, 3 transform + split + stores,
       4 / 5 B reads + weight wait + MFMAs + reload of position 0 / 1, 6 barrier) */
    const bool tsub = rec && (a.wdma & 128) != 0;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tsp = 0;
#define W3_TS(i)                                                                                                \
    if (tsub) {                                                                                                 \
        const unsigned long long now = __builtin_amdgcn_s_memtime();                                            \
        ts[i] += now - tsp;                                                                                     \
        tsp = now;                                                                                              \
    }
#else
#define W3_TS(i)
#endif

    // ---- chunk range of this workgroup (a.ksplit == 2: blockIdx.y picks one half of the input channels)
    const int nch_all = a.CinP / CK;
    const int ksp = a.ksplit >= 2 ? a.ksplit : 1, kh = ksp >= 2 ? (int)blockIdx.y : 0;          // 2, 4 or 8 parts of the input channels
    const int c_begin = kh * (nch_all / ksp), c_end = c_begin + nch_all / ksp;

    // ---- prologue.  Issue order = need order: the coefficients of the sample (into the registers of the third patch), the raw patches of
    // the first two chunks (into the registers of weight quads 0 .. PQ-1, which are not needed before the first MFMA phase), then the
    // weight quads PQ.. of the first chunk.  Coefficients and patches are consumed as soon as THEY have landed (the weights, most of the
    // bytes, are still in flight); the third patch and quads 0 .. PQ-1 follow once their registers have been read.
    int vtok = 0;                                       // ordering token: written by every VMEM wait, an operand of the register reads
    {
        const float nodep = 0.0f;
        unsigned ofs[NPL];
        W3_READ_OFF(ofs)
        if (rec) sp[3] = __builtin_amdgcn_s_memtime() - tk0;      // index arithmetic done, only the coefficients requested yet
        // VMEM issue order = need order.  Round 3's first form fetched the coefficients with ordinary loads and parked them before
        // anything else was requested: a full memory latency (2-3 k cycles under load) in front of every workgroup's first patch.
        // the first two patches land in the registers of weight quads 0 .. PQ-1 (v184-v203)
        W3_LOAD_PR(c_begin, nodep, ofs, "v[184:187]", "v[188:191]", "v192")
        W3_LOAD_PR(c_begin + 1, nodep, ofs, "v[194:197]", "v[198:201]", "v202")
        if (G8)                            // the halo of both patch buffers is zero padding for the whole kernel
            for (int i = tid; i < 2 * PBUF; i += NT) sP[i] = 0.0f;
        if (rec) sp[0] = __builtin_amdgcn_s_memtime() - tk0;      // loads issued
        W3_WAIT(0)                         // the coefficients and the two patches have landed
        if (rec) sp[1] = __builtin_amdgcn_s_memtime() - tk0;      // first patches landed
        float cdep = 0.0f;
        if (PRO) {
            f32x2 cpre[G8 ? 4 : 2];
            asm volatile("v_mov_b32 %0, v172\n\tv_mov_b32 %1, v173\n\tv_mov_b32 %2, v174\n\tv_mov_b32 %3, v175"
                         : "=v"(cpre[0].x), "=v"(cpre[0].y), "=v"(cpre[1].x), "=v"(cpre[1].y) : "s"(vtok));
            if constexpr (G8)
                asm volatile("v_mov_b32 %0, v176\n\tv_mov_b32 %1, v177\n\tv_mov_b32 %2, v178\n\tv_mov_b32 %3, v179"
                             : "=v"(cpre[2].x), "=v"(cpre[2].y), "=v"(cpre[3].x), "=v"(cpre[3].y) : "s"(vtok));
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (tid + k * NT < Cin) {
                    *reinterpret_cast<f32x2*>(sCo + (tid + k * NT) * 2) = cpre[k];
                    if constexpr (G8) *reinterpret_cast<f32x2*>(sCo + (Cin + tid + k * NT) * 2) = cpre[2 + k];
                }
            cdep = cpre[0].x + cpre[1].x + (G8 ? cpre[2].x + cpre[3].x : 0.0f);
        }
        W3_LOAD_P(c_begin + 2, cdep, ofs)  // (behind the reads of v172-v179)
        if (PRO || G8) __syncthreads();    // coefficient table (and the zeroed halo) visible
        if (rec) sp[4] = __builtin_amdgcn_s_memtime() - tk0;      // coefficient table visible
        {
            // The weight quads PQ.. of the first chunk (most of the prologue's bytes: 13 KB per wave at COT = 3) are requested ONE BEHIND
            // EACH ACTIVATED VALUE: issued in one burst they fill the CU's vector-memory queue and every wave sat 2-3 k cycles in the
            // issue (8 waves x 21 KB at 64 bytes per clock; profiles/r03_wino3_prologue.txt) before it could touch its patches.
            constexpr int QR = (NA - PQ + 2 * NPV - 1) / (2 * NPV);       // quads per value
#define W3_HOOK0(e, v) W3_LOAD_A_RANGE(c_begin, PQ + (e) * QR, (PQ + ((e) + 1) * QR < NA ? PQ + ((e) + 1) * QR : NA), v)
#define W3_HOOK1(e, v) W3_LOAD_A_RANGE(c_begin, (PQ + (NPV + (e)) * QR < NA ? PQ + (NPV + (e)) * QR : NA), (PQ + (NPV + (e) + 1) * QR < NA ? PQ + (NPV + (e) + 1) * QR : NA), v)
            float pv0[NPV], pv1[NPV];
            f32x2 cf0[NPL], cf1[NPL];
            W3_READ_C(c_begin, cf0)
            W3_READ_C(c_begin + 1, cf1)
            W3_WRITE_PR(c_begin, pv0, cf0, "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", W3_HOOK0)
            W3_WRITE_PR(c_begin + 1, pv1, cf1, "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", W3_HOOK1)
#undef W3_HOOK0
#undef W3_HOOK1
            const float dep = pv0[0] + pv1[0];
            W3_LOAD_A_RANGE(c_begin, 0, PQ, dep)
        }
    }
    __syncthreads();                       // the first two patches visible
    if (rec) sp[2] = __builtin_amdgcn_s_memtime() - tk0;          // first two patches activated and parked
    {
        f32x2 rw[3][4];
        W3_READ_R(c_begin, rw)
        W3_WRITE_V(c_begin, rg, rw)
    }
    __syncthreads();                       // V of the first chunk visible
    W3_STAMP(0)

    // ---- K loop.  VMEM issue order of a wave per chunk c (in-order vmcnt counter; nothing else is outstanding):
    //   waves 0-3:  [patch(c+3): NPL loads] [weights(c+1): NQ loads behind the MFMAs of each position]      waves 4-7:  weights, then patch
    // wait points (the same counts in both orders):
    //   patch(c+2) before its write: one chunk's weight loads were issued after it                              vmcnt(NA)
    //   weights(c) of a position before its MFMAs: see W3_MFMA_PHASE                                            vmcnt(VM_A)
    // (the loads still in flight when a loop is left target registers the compiler does not know: one wait behind the loops)
    W3_WAIT(0)                             // weight quads 0 .. PQ-1 were issued last: the loop's in-order counts start from an empty queue
    const int ph = rg;                     // phase order of the wave
    if (ph == 0) {
        for (int c = c_begin; c + 1 < c_end; ++c) {
            W3_VALU_PHASE(c, rg)
            W3_MFMA_PHASE(c, true)
            // chunk c read by every wave; V(c+1), patch(c+2) visible.  LDS traffic only: no VMEM wait at the barrier.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            W3_TS(6)
        }
    } else {
        for (int c = c_begin; c + 1 < c_end; ++c) {
            W3_MFMA_PHASE(c, true)
            W3_VALU_PHASE(c, rg)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            W3_TS(6)
        }
    }
    W3_WAIT(0)
    {
        const int c = c_end - 1;
        W3_MFMA_PHASE(c, false)
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
    // bias and residual of all 2 * COT tasks of the thread are requested up front: fetched where they are used, each sub-tile waited
    // for two dependent memory latencies between its barriers and the epilogue took 7 k cycles, twice what its LDS traffic needs
    // (profiles/r03_wino3_kloop_merged_vs_two_phase.txt: "epi")
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
            const float v00 = (y00 + bvv + r0.x) * osc, v01 = (y01 + bvv + r0.y) * osc;
            const float v10 = (y10 + bvv + r1.x) * osc, v11 = (y11 + bvv + r1.y) * osc;
            if (co < a.Cout && e_valid) {
                const long o = ((long)e_b * a.Cout + co) * HW + pix;
                *reinterpret_cast<float2*>(ydst + o) = make_float2(v00, v01);
                *reinterpret_cast<float2*>(ydst + o + W) = make_float2(v10, v11);
            }
            if (a.stats && fin) {
                // GroupNorm partials of the FINAL values (ConvArgs::stats): the tiles of this cout are the 32 lanes of a half-wave (G8:
                // 16 lanes = one DPP row per image).  Pilot-shifted moments (conv_wino2h.cpp has the derivation): with p = the group's
                // first value, s = sum (v - p) and q = sum (v - p)^2 merge by plain addition, two DPP adds per level.
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
#define W3_MERGE(CTRL, ROWMASK)                                                                                     \
                {                                                                                                   \
                    sm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sm), CTRL, ROWMASK, 0xf, false)); \
                    qm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, qm), CTRL, ROWMASK, 0xf, false)); \
                }
                W3_MERGE(0xB1, 0xf)                   // quad_perm [1,0,3,2]
                W3_MERGE(0x4E, 0xf)                   // quad_perm [2,3,0,1]
                W3_MERGE(0x124, 0xf)                  // row_ror:4
                W3_MERGE(0x128, 0xf)                  // row_ror:8: every lane of a row of 16 holds the row's totals
                if (!G8) W3_MERGE(0x142, 0xa)         // row_bcast:15: lanes 16-31 / 48-63 add the totals of the row below
#undef W3_MERGE
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
            d[5] = now - tprev;            // epilogue
            d[6] = (unsigned long long)(c_end - c_begin);
            d[7] = now - tk0;
            // MCVD_DBG_WAVE >= 64: prologue sub-phases instead (cycles from the start of the kernel): index arithmetic done, loads issued,
            // coefficients + first patches landed, coefficient table visible, first two patches parked
            if (sub) { d[1] = sp[3]; d[2] = sp[0]; d[3] = sp[1]; d[4] = sp[4]; d[5] = sp[2]; }
#ifdef MCVD_DIAG
            if (tsub) { d[0] = ts[0]; d[1] = ts[1]; d[2] = ts[2]; d[3] = ts[3]; d[4] = ts[4]; d[5] = ts[5]; d[7] = ts[6]; }
#endif
        }
    }
#undef W3_STAMP
#undef W3_TS
#undef W3_LOAD_A
#undef W3_LOAD_A_PIECE
#undef W3_QUADS
#undef W3_LD1
#undef W3_MF1
#undef W3_PRODUCT
#undef W3_LOAD_P
#undef W3_LOAD_PR
#undef W3_WRITE_PR
#undef W3_LD1D
#undef W3_LOAD_A_RANGE
#undef W3_WAIT
#undef W3_WRITE_P
#undef W3_WRITE_V
#undef W3_READ_R
#undef W3_READ_C
#undef W3_READ_OFF
#undef W3_LOAD_B
#undef W3_MFMA_PHASE
#undef W3_VALU_PHASE
}

static size_t wino3_lds_bytes(int Cin, bool g8) {
    return (size_t)(2 * W3_VW + 2 * (W3_CK * 10 * W3_PP + 8) + (g8 ? 4 : 2) * Cin + (g8 ? 1 : 3) * W3_NT) * sizeof(float);
}

// the K-split second pass lives in conv_wino.cpp
int launch_wino_ksplit_reduce(const ConvArgs& a, hipStream_t s);

template <int COT, int PRO, bool G8, int EXP>
static int wino3_launch_k(const ConvArgs& k, dim3 grid, size_t lds, hipStream_t s) {
    static PerDeviceOnce raised;
    if (raised.first_use()) {
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino3_kernel<COT, PRO, G8, EXP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised.done();
    }
    hipLaunchKernelGGL((conv_wino3_kernel<COT, PRO, G8, EXP>), grid, dim3(W3_NT), lds, s, k);
    return 0;
}

template <int COT, int PRO, bool G8>
static int wino3_launch2(const ConvArgs& a, hipStream_t s) {
    constexpr int BCO = 32 * COT;
    const size_t lds = wino3_lds_bytes(a.Cin, G8);
    const int nreg = G8 ? (a.B + 1) / 2 : a.B * (a.H / 8) * (a.W / 16);
    const int ksp = a.ksplit >= 2 ? a.ksplit : 1;
    dim3 grid(((nreg + 7) / 8) * 8 * (a.CoutP / BCO), ksp);
    ConvArgs k = a;
    int rc = 0;
    if (k.dbg) k.wdma = 0;                 // wave 0 records its phase times
#ifdef MCVD_DIAG
    // diagnostics build only: which wave records, and the timing-only ablations of the K loop (WRONG RESULTS; the production library
    // has neither the env hooks nor the ablation kernels)
    if (k.dbg) {
        const char* w = getenv("MCVD_DBG_WAVE");
        k.wdma = w ? atoi(w) : 0;
    }
    const char* exp_s = getenv("MCVD_WINO3_EXP");
    const int e = exp_s ? atoi(exp_s) : 0;
    if (COT == 3 && PRO == 2 && !G8 && e != 0) {
        switch (e) {
            case 1: rc = wino3_launch_k<3, 2, false, 1>(k, grid, lds, s); break;        // no transform
            case 2: rc = wino3_launch_k<3, 2, false, 2>(k, grid, lds, s); break;        // no patch activation / park
            case 3: rc = wino3_launch_k<3, 2, false, 3>(k, grid, lds, s); break;        // neither
            case 4: rc = wino3_launch_k<3, 2, false, 4>(k, grid, lds, s); break;        // no VMEM in the loop
            case 16: rc = wino3_launch_k<3, 2, false, 16>(k, grid, lds, s); break;      // everything but the MFMAs
            case 15: rc = wino3_launch_k<3, 2, false, 15>(k, grid, lds, s); break;      // MFMA only
            case 27: rc = wino3_launch_k<3, 2, false, 27>(k, grid, lds, s); break;      // VMEM only
            case 11: rc = wino3_launch_k<3, 2, false, 11>(k, grid, lds, s); break;      // VMEM + MFMA only
            default: mcvd::set_error("MCVD_WINO3_EXP=%d is not a built ablation", e); return -1;
        }
    } else
#endif
    {
        rc = wino3_launch_k<COT, PRO, G8, 0>(k, grid, lds, s);
    }
    if (rc) return rc;
    MCVD_HIP_CHECK(hipGetLastError());
    if (ksp >= 2) return launch_wino_ksplit_reduce(a, s);
    if (a.stats) set_last_conv_stats_np(G8 ? 1 : (a.H / 8) * (a.W / 16));
    return 0;
}

template <int COT, bool G8>
static int wino3_launch1(const ConvArgs& a, hipStream_t s) {
    if (!a.coef && !a.act) return wino3_launch2<COT, 0, G8>(a, s);
    if (!a.act) return wino3_launch2<COT, 1, G8>(a, s);
    return wino3_launch2<COT, 2, G8>(a, s);
}

template <int COT>
static int wino3_launch(const ConvArgs& a, hipStream_t s) {
    return (a.H == 8 && a.W == 8) ? wino3_launch1<COT, true>(a, s) : wino3_launch1<COT, false>(a, s);
}

// Shape ids 10 / 11 apply to this launch: regions of 8 x 16 output pixels or 8 x 8 images (two per workgroup), no SPADE prologue,
// pre-split packed weights present (11: and an even chunk count).
bool conv_wino3_usable(const ConvArgs& a) {
    const bool g8 = a.H == 8 && a.W == 8;
    return a.ks == 3 && ((a.H % 8 == 0 && a.W % 16 == 0 && a.H >= 8 && a.W >= 16) || g8) && a.wpb && !a.gb && a.Cin <= 1024 &&
           a.CinP % W3_CK == 0 && (a.C1 == 0 || a.C0 % W3_CK == 0) && a.H * a.W <= 16384 &&
           (long)a.B * (a.C0 > a.C1 ? a.C0 : a.C1) * a.H * a.W < (1L << 29) && wino3_lds_bytes(a.Cin, g8) <= 160 * 1024 &&
           (a.ksplit < 2 || ((a.ksplit == 2 || a.ksplit == 4 || a.ksplit == 8) && (a.CinP / W3_CK) % a.ksplit == 0 &&
                             a.CinP / W3_CK >= 2 * a.ksplit && conv_part_fits(a)));        // every part at least two chunks
}

// a.wpb: the layout of launch_pack_wino3_weight, packed for conv_wino_cout_tile(Cout).
int launch_conv_wino3(const ConvArgs& a, hipStream_t s) {
    MCVD_REQUIRE(conv_wino3_usable(a), "winograd bf16x3 conv: unsupported (ks=%d H=%d W=%d Cin=%d C0=%d ksplit=%d, packed weight pieces %s)",
                 a.ks, a.H, a.W, a.Cin, a.C0, a.ksplit, a.wpb ? "present" : "missing");
    const int cot = conv_wino_cout_tile(a.Cout);
    MCVD_REQUIRE(a.CoutP % (32 * cot) == 0, "winograd bf16x3 conv: CoutP=%d vs tile %d", a.CoutP, 32 * cot);
    switch (cot) {
        case 1: return wino3_launch<1>(a, s);
        case 2: return wino3_launch<2>(a, s);
        default: return wino3_launch<3>(a, s);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight pieces.  wb = CinP * 16 * CoutP * 3 bf16 halfwords, no header (nothing is scaled).
//   U = G g G^T per (cout, cin) in fp32 (the arithmetic of pack_wino_weight_kernel), split exactly u1 = bf16(U), u2 = bf16(U - u1),
//   u3 = bf16(U - u1 - u2), stored as 16-bit halves at
//   ((((cotile*nchunks + ci/16)*16 + xi)*COT + ct)*3 + piece)*512 + (lane*4 + j)*2 + (e & 1),
//   cc = ci % 16 = 2e + h,  j = e >> 1,  lane = h*32 + co%32,  ct = (co % BCO) / 32.
// The destination must be zero-filled (padded channels stay zero).
__device__ __forceinline__ unsigned short w3_bf16_rne(float v) {      // finite inputs
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float w3_bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__global__ void pack_wino3_weight_kernel(const float* w, unsigned short* dst, int Cout, int Cin, int CinP, int CoutP, int COT) {
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
        // halfword index of (xi = 0, piece 0); one position further = COT * 1536 halfwords, one piece = + 512
        const long base = ((((long)cotile * nch + (ci >> 4)) * 16) * COT + ct) * 1536 + (lane * 4 + (el >> 1)) * 2 + (el & 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {                           // (.) G^T   (the same arithmetic as pack_wino_weight_kernel)
            float u[4];
            u[0] = t[r][0];
            u[1] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
            u[2] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
            u[3] = t[r][2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned short h1 = w3_bf16_rne(u[c]);
                const float r1 = u[c] - w3_bf16_f32(h1);        // exact
                const unsigned short h2 = w3_bf16_rne(r1);
                const float r2 = r1 - w3_bf16_f32(h2);          // exact, at most 8 significant bits
                const unsigned short h3 = w3_bf16_rne(r2);
                const long o = base + (long)(r * 4 + c) * COT * 1536;
                dst[o] = h1;
                dst[o + 512] = h2;
                dst[o + 1024] = h3;
            }
        }
    }
}

// `wb` (conv_wino3_weight_floats(CinP, CoutP) floats) must be zero-filled by the caller.
int launch_pack_wino3_weight(const float* w, float* wb, int Cout, int Cin, int CinP, int CoutP, hipStream_t s) {
    const int cot = conv_wino_cout_tile(Cout);
    MCVD_REQUIRE(CinP % 16 == 0 && CoutP % (32 * cot) == 0, "pack_wino3_weight: CinP=%d CoutP=%d cot=%d", CinP, CoutP, cot);
    const long n = (long)Cout * Cin;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_wino3_weight_kernel, dim3(blocks), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(wb), Cout, Cin, CinP,
                       CoutP, cot);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

long conv_wino3_weight_floats(int CinP, int CoutP) { return (long)CinP * 16 * CoutP / 2 * 3; }

}  // namespace mcvd
