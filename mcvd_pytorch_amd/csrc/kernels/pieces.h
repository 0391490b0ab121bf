// Split-operand arithmetic on the 16-bit matrix pipes (v_mfma_f32_32x32x16_{f16,bf16}: 16x the rate of the fp32 MFMA), shared by the
// 1x1 GEMM (conv1x1_h2.cpp) and the attention kernel (attention_h2.cpp); the Winograd kernels (conv_wino2h.cpp, conv_wino3.cpp)
// carry the same arithmetic in their hand-scheduled loops.  An fp32 operand v is represented by NP 16-bit pieces and a product by
// the piece products that matter, accumulated in fp32, smallest first:
//
//   Pieces<3> -- THREE BF16 PIECES, FP32-EQUIVALENT (the default arithmetic of the library).  v = v1 + v2 + v3 exactly (round to
//       nearest at every level; the remainders are exact in fp32 and the third has at most 8 significant bits), six products
//       u1 v3 + u3 v1 + u2 v2 + u1 v2 + u2 v1 + u1 v1; dropped: <= 2^-23.4 |u v|, less than one fp32 rounding per product.  bf16
//       has the fp32 exponent range: no scale, no clamp, Inf / NaN propagate.  (Bit-exact split for |v| >= 2^-110; below that the
//       third piece is a bf16 denormal, ulp 2^-133, and v loses bits gradually: tests/test_bf16x3_arithmetic_cpu.py.)
//   Pieces<2> -- TWO FP16 PIECES (context option "f16x2", off by default).  v ~= v1 + v2, |v - v1 - v2| <= 2^-22 |v| (22-23
//       significant bits where fp32 has 24), three products u1 v2 + u2 v1 + u1 v1.  fp16 has 5 exponent bits: operands are scaled
//       by powers of two into its range (weights per layer at pack time, activations by 2^4).  Nothing is clamped: a value beyond
//       the range becomes Inf and the accumulator NaN.  The library keeps raw (unbounded) tensors away from these kernels and scans
//       every forward's output for non-finite values (model.cpp: f16x2 range guard; MCVD_ERANGE).
#pragma once
#include <hip/hip_runtime.h>

namespace mcvd {

typedef float px_f32x16 __attribute__((ext_vector_type(16)));
typedef float px_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 px_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 px_f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 px_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 px_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned px_u32x4 __attribute__((ext_vector_type(4)));

constexpr float F16X2_ACT_SCALE = 16.0f;                    // activations enter the fp16 pieces times 2^4 (exact): finite up to
                                                            // 65504 / 16 = 4094 (a Winograd transform value is a sum of four
                                                            // activations: ~1e3 per activation in the worst case)

template <int NP>
struct Pieces;

template <>
struct Pieces<2> {
    static constexpr int N = 2;
    static constexpr int NPROD = 3;
    static constexpr float ACT_SCALE = F16X2_ACT_SCALE;
    static constexpr int HDR = 4;            // header floats in front of packed weight pieces: |w|max, scale 2^e, 2^-e, -
    // product k = weight piece PA(k) x activation piece PB(k), smallest first
    __device__ static constexpr int PA(int k) { return k == 1 ? 1 : 0; }
    __device__ static constexpr int PB(int k) { return k == 0 ? 1 : 0; }
    __device__ static __forceinline__ unsigned cvt_pk(float lo, float hi) {
        const px_f32x2 v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, px_f16x2));
    }
    // (x, y) -> w[piece] = packed pair (x piece in the low half); CLAMP (unused by the library: overflow is meant to be loud):
    // saturate to the fp16 range first
    template <bool CLAMP>
    __device__ static __forceinline__ void split(float x, float y, unsigned (&w)[2]) {
        if (CLAMP) {
            x = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
            y = __builtin_amdgcn_fmed3f(y, -65504.0f, 65504.0f);
        }
        w[0] = cvt_pk(x, y);
        float rx, ry;                        // exact remainders straight from the packed halves
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(w[0]), "v"(x));
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(w[0]), "v"(y));
        w[1] = cvt_pk(rx, ry);
    }
    __device__ static __forceinline__ px_f32x16 mfma(const px_u32x4& a, const px_u32x4& b, const px_f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(px_f16x8, a), __builtin_bit_cast(px_f16x8, b), c, 0, 0, 0);
    }
};

template <>
struct Pieces<3> {
    static constexpr int N = 3;
    static constexpr int NPROD = 6;
    static constexpr float ACT_SCALE = 1.0f;
    static constexpr int HDR = 0;
    __device__ static constexpr int PA(int k) { return k == 0 ? 0 : k == 1 ? 2 : k == 2 ? 1 : k == 3 ? 0 : k == 4 ? 1 : 0; }
    __device__ static constexpr int PB(int k) { return k == 0 ? 2 : k == 1 ? 0 : k == 2 ? 1 : k == 3 ? 1 : k == 4 ? 0 : 0; }
    __device__ static __forceinline__ unsigned cvt_pk(float lo, float hi) {
        const px_f32x2 v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, px_bf16x2));
    }
    template <bool CLAMP>                    // (no range to clamp to)
    __device__ static __forceinline__ void split(float x, float y, unsigned (&w)[3]) {
        px_f32x2 v = {x, y};
        w[0] = cvt_pk(v.x, v.y);
        const px_f32x2 h = {__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xffff0000u)};
        v = v - h;
        w[1] = cvt_pk(v.x, v.y);
        const px_f32x2 g = {__builtin_bit_cast(float, w[1] << 16), __builtin_bit_cast(float, w[1] & 0xffff0000u)};
        v = v - g;
        w[2] = cvt_pk(v.x, v.y);
    }
    __device__ static __forceinline__ px_f32x16 mfma(const px_u32x4& a, const px_u32x4& b, const px_f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(px_bf16x8, a), __builtin_bit_cast(px_bf16x8, b), c, 0, 0, 0);
    }
};

// host / pack-kernel side: bf16 round-to-nearest-even of a finite fp32 value and back
__host__ __device__ inline unsigned short px_bf16_rne(float v) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__host__ __device__ inline float px_bf16_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

}  // namespace mcvd
