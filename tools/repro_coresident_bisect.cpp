// Co-residency corruption, the bisect (round 6): from "a bad launch of conv_mfma_kernel" down to two instructions.
//
// profiles/r06_coresident_cause.txt has the chain.  In a bad launch of the library's direct conv beside attn_h2_kernel<3,3> the staged activation of
// one (channel, row) -- 16 consecutive lanes, ONE VGPR -- holds f(A x) instead of f(A x + B); everything else is right.  hipcc's code for
//     if (a.coef) { v.x = v.x * rc.x + rc.y; ... }
// is   v_pk_fma_f32 v[4:5], v[100:101], v[138:139], v[138:139] op_sel:[0,0,1] op_sel_hi:[1,0,1]   (x2), four v_cndmask_b32_e64, an s_andn2_b64.
// This program runs that sequence (inline asm, same physical registers) as a stand-alone VICTIM, and variants of it (VARS), beside the
// stand-alone attention kernel or beside SYNTHETIC aggressors that loop over one instruction class each (AGGRS), and counts launches whose
// result differs from the same launch made alone.  Outcome: the victim is the packed FMA that reads one VGPR pair as src1 AND src2 (variants
// 0 1 2 5 6 8-12; not 3 4 7 13), the aggressor is a matrix instruction with 128-bit A / B operands in ANOTHER wave of the SIMD (aggressors
// 0 2 9 11; not 1 3-8 10 12 13; not the same instruction inside the victim's own wave: variants 8-12 alone).  The 100-line version that
// shows only the end of the chain is tools/repro_pk_fma_beside_mfma.cpp.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/repro_coresident_bisect.cpp -o tools/bin/repro_coresident_bisect -lpthread
//   tools/bin/repro_coresident_bisect [seconds per phase]     env VARS ("0 1 ... 17"), AGGRS ("0"; 1..13 synthetic), AITER, AGRID, GRID (48), LDSKB (68), ITER (64)
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "../mcvd_pytorch_amd/csrc/kernels/attention_h2.cpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

namespace mcvd {      // the aggressor's translation unit reports errors through these
static thread_local char g_err[512];
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
const char* get_error() { return g_err; }
int launch_attention_naive(const float*, float*, int, int, int, int, hipStream_t) { return -1; }
int launch_attention_mfma(const float*, float*, int, int, int, int, hipStream_t) { return -1; }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// VAR 0: the sequence as generated.  1: without the SALU instruction between the selects.  2: selects under VCC instead of an SGPR pair.
// 3: four v_fma_f32 instead of the two packed ones.  4: the packed FMAs without op_sel (coefficients broadcast beforehand).
// 5: 0 + the SiLU tail (v_exp_f32 / v_rcp_f32 / v_pk_mul_f32) as generated.  6: the two v_pk_fma_f32 alone.  7: 6 with the addend read from a copy.
template <int VAR>
__global__ __launch_bounds__(256) void seq_victim(const f32x4* __restrict__ x, const f32x2* __restrict__ coef, f32x4* __restrict__ y, long n4, int iters,
                                                  unsigned long long mask, unsigned long long other) {
    extern __shared__ float pad_lds[];
    if (mask == 12345ull) pad_lds[threadIdx.x] = 1.0f;      // keep the allocation
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = x[i];
        const f32x2 c = coef[(i >> 4) & 1023];                // 16 consecutive lanes share (A, B), like one (channel, row) of the conv's patch
        f32x4 r = v, acc = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            asm volatile(
                "v_mov_b32 v98, %4\n v_mov_b32 v99, %5\n v_mov_b32 v100, %6\n v_mov_b32 v101, %7\n"
                "v_mov_b32 v138, %8\n v_mov_b32 v139, %9\n"
                "v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n"
                "s_nop 4\n"
                : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w)
                : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(c.x), "v"(c.y)
                : "v2", "v3", "v4", "v5", "v98", "v99", "v100", "v101", "v138", "v139");
#define MCVD_INWAVE(NOP) asm volatile("v_mfma_f32_16x16x32_bf16 v[160:163], v[150:153], v[154:157], v[160:163]\n" NOP \
                                      "v_pk_fma_f32 v[4:5], v[100:101], v[138:139], v[138:139] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n" \
                                      "v_pk_fma_f32 v[2:3], v[98:99], v[138:139], v[138:139] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n" \
                                      ::: "v2", "v3", "v4", "v5", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v160", "v161", "v162", "v163")
            if (VAR >= 8 && VAR <= 12) {      // the matrix instruction in the SAME wave, N wait states in front of the packed FMAs (no second kernel needed?)
                if (it == 0) asm volatile("v_mov_b32 v150, 0\n v_mov_b32 v151, 0\n v_mov_b32 v152, 0\n v_mov_b32 v153, 0\n v_mov_b32 v154, 0\n v_mov_b32 v155, 0\n v_mov_b32 v156, 0\n v_mov_b32 v157, 0\n"
                                          "v_mov_b32 v160, 0\n v_mov_b32 v161, 0\n v_mov_b32 v162, 0\n v_mov_b32 v163, 0\n"
                                          ::: "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v160", "v161", "v162", "v163");
                if (VAR == 8) MCVD_INWAVE("");
                else if (VAR == 9) MCVD_INWAVE("s_nop 0\n");
                else if (VAR == 10) MCVD_INWAVE("s_nop 3\n");
                else if (VAR == 11) MCVD_INWAVE("s_nop 7\n");
                else MCVD_INWAVE("s_nop 15\n");
            } else if (VAR == 14)      // unpacked: ONE 32-bit register as src1 and src2 (x * c + c)
                asm volatile("v_fma_f32 v4, v100, v138, v138\n v_fma_f32 v5, v101, v139, v139\n v_fma_f32 v2, v98, v138, v138\n v_fma_f32 v3, v99, v139, v139\n"
                             ::: "v2", "v3", "v4", "v5");
            else if (VAR == 15)        // packed, one pair as src1 and src2, NO op_sel (x * c + c per half)
                asm volatile("v_pk_fma_f32 v[4:5], v[100:101], v[138:139], v[138:139]\n v_pk_fma_f32 v[2:3], v[98:99], v[138:139], v[138:139]\n"
                             ::: "v2", "v3", "v4", "v5");
            else if (VAR == 16)        // packed, one pair as src0 and src2 (the form hipcc gave the clean conv instantiations): A from src0, x src1, B src2
                asm volatile("v_pk_fma_f32 v[4:5], v[138:139], v[100:101], v[138:139] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n"
                             "v_pk_fma_f32 v[2:3], v[138:139], v[98:99], v[138:139] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n"
                             ::: "v2", "v3", "v4", "v5");
            else if (VAR == 17)        // packed multiply, one pair as src0 and src1, different halves broadcast (c.x * c.y in both halves) + add x
                asm volatile("v_pk_mul_f32 v[4:5], v[138:139], v[138:139] op_sel:[0,1] op_sel_hi:[0,1]\n v_pk_mul_f32 v[2:3], v[138:139], v[138:139] op_sel:[0,1] op_sel_hi:[0,1]\n"
                             "s_nop 1\n v_pk_add_f32 v[4:5], v[4:5], v[100:101]\n v_pk_add_f32 v[2:3], v[2:3], v[98:99]\n"
                             ::: "v2", "v3", "v4", "v5");
            else if (VAR == 13)      // the other dual read hipcc generates (temb_mlp_kernel, gamma_noise_kernel): src0 == src1, halves swapped
                asm volatile("v_pk_add_f32 v[4:5], v[100:101], v[100:101] op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_mul_f32 v[2:3], v[98:99], v[98:99] op_sel:[0,1] op_sel_hi:[1,0]\n"
                             ::: "v2", "v3", "v4", "v5");
            else if (VAR == 3)
                asm volatile("v_fma_f32 v4, v100, v138, v139\n v_fma_f32 v5, v101, v138, v139\n v_fma_f32 v2, v98, v138, v139\n v_fma_f32 v3, v99, v138, v139\n"
                             ::: "v2", "v3", "v4", "v5");
            else if (VAR == 4)
                asm volatile("v_mov_b32 v140, v138\n v_mov_b32 v141, v138\n v_mov_b32 v142, v139\n v_mov_b32 v143, v139\n s_nop 1\n"
                             "v_pk_fma_f32 v[4:5], v[100:101], v[140:141], v[142:143]\n v_pk_fma_f32 v[2:3], v[98:99], v[140:141], v[142:143]\n"
                             ::: "v2", "v3", "v4", "v5", "v140", "v141", "v142", "v143");
            else if (VAR == 7)      // the addend from its own register pair (a copy): is reading ONE pair for two operands part of it?
                asm volatile("v_mov_b32 v142, v138\n v_mov_b32 v143, v139\n s_nop 1\n"
                             "v_pk_fma_f32 v[4:5], v[100:101], v[138:139], v[142:143] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 v[2:3], v[98:99], v[138:139], v[142:143] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                             ::: "v2", "v3", "v4", "v5", "v142", "v143");
            else
                asm volatile("v_pk_fma_f32 v[4:5], v[100:101], v[138:139], v[138:139] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 v[2:3], v[98:99], v[138:139], v[138:139] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
                             ::: "v2", "v3", "v4", "v5");
            if (VAR >= 6) {}      // no selects at all
            else if (VAR == 2)
                asm volatile("s_mov_b64 vcc, %0\n s_nop 4\n"
                             "v_cndmask_b32_e32 v5, v5, v101, vcc\n v_cndmask_b32_e32 v3, v3, v99, vcc\n v_cndmask_b32_e32 v2, v2, v98, vcc\n"
                             "v_cndmask_b32_e32 v4, v4, v100, vcc\n"
                             :: "s"(mask) : "v2", "v3", "v4", "v5", "vcc", "scc");
            else if (VAR == 1)
                asm volatile("v_cndmask_b32_e64 v5, v5, v101, %0\n v_cndmask_b32_e64 v3, v3, v99, %0\n v_cndmask_b32_e64 v2, v2, v98, %0\n"
                             "v_cndmask_b32_e64 v4, v4, v100, %0\n"
                             :: "s"(mask) : "v2", "v3", "v4", "v5");
            else
                asm volatile("v_cndmask_b32_e64 v5, v5, v101, %0\n v_cndmask_b32_e64 v3, v3, v99, %0\n v_cndmask_b32_e64 v2, v2, v98, %0\n"
                             "s_andn2_b64 vcc, exec, %1\n"
                             "v_cndmask_b32_e64 v4, v4, v100, %0\n"
                             :: "s"(mask), "s"(other) : "v2", "v3", "v4", "v5", "vcc", "scc");
            if (VAR == 5)
                asm volatile("v_mul_f32_e32 v8, 0xbfb8aa3b, v2\n v_exp_f32_e32 v8, v8\n v_mul_f32_e32 v22, 0xbfb8aa3b, v3\n v_mul_f32_e32 v23, 0xbfb8aa3b, v4\n"
                             "v_exp_f32_e32 v24, v22\n v_add_f32_e32 v8, 1.0, v8\n v_rcp_f32_e32 v22, v8\n v_exp_f32_e32 v8, v23\n v_mul_f32_e32 v23, 0xbfb8aa3b, v5\n"
                             "v_exp_f32_e32 v23, v23\n v_add_f32_e32 v26, 1.0, v24\n v_add_f32_e32 v8, 1.0, v8\n v_rcp_f32_e32 v24, v8\n v_add_f32_e32 v8, 1.0, v23\n"
                             "v_rcp_f32_e32 v25, v8\n v_rcp_f32_e32 v23, v26\n s_nop 1\n v_pk_mul_f32 v[4:5], v[4:5], v[24:25]\n v_pk_mul_f32 v[2:3], v[2:3], v[22:23]\n"
                             ::: "v2", "v3", "v4", "v5", "v8", "v22", "v23", "v24", "v25", "v26");
            asm volatile("s_nop 4\n v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n v_mov_b32 %2, v4\n v_mov_b32 %3, v5\n"
                         : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) :: "v2", "v3", "v4", "v5");
            acc += r;
        }
        y[i] = acc;
    }
}

// Synthetic aggressors (AGGRS="1 2 ..."): what in attn_h2_kernel<3,3> does it?  One instruction class each, in a long loop, 256 threads per workgroup.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int K>
__global__ __launch_bounds__(256, 2) void synth_aggressor(float* sink, int iters) {
    __shared__ f32x4 lds[1024];
    const int t = threadIdx.x;
    f32x2 a = {1.0f + t * 1e-3f, 0.5f - t * 1e-3f}, b = {0.25f, -0.75f}, c = {t * 1e-4f, 1.0f};
    f32x4 q = {a.x, a.y, b.x, b.y};
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    unsigned u = t;
    lds[t] = q; lds[t + 256] = q; lds[t + 512] = q; lds[t + 768] = q;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (K == 1) {          // v_cvt_pk_bf16_f32 (new on gfx950)
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(a.x), "v"(__builtin_bit_cast(float, u)));
        } else if (K == 2) {   // v_mfma_f32_32x32x16_bf16 (new on gfx950)
            const mcvd::px_bf16x8 x8 = __builtin_bit_cast(mcvd::px_bf16x8, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x8, x8, acc, 0, 0, 0);
        } else if (K == 3) {   // the packed fp32 forms the attention kernel uses (neg / op_sel_hi modifiers)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %0, %0, %2 op_sel_hi:[0,1]\n"
                             "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n" : "+v"(c) : "v"(a), "v"(b));
        } else if (K == 4) {   // transcendental
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(a.x));
        } else if (K == 5) {   // LDS traffic
#pragma unroll
            for (int k = 0; k < 4; ++k) { q += lds[(t + 64 * k + it) & 1023]; }
            lds[(t + it) & 1023] = q;
        } else if (K == 6) {   // plain packed FMA
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
        } else if (K == 7) {   // plain fp32 MFMA
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        } else if (K == 8) {   // bf16 MFMA fed from freshly converted registers, as the kernel does: cvt -> mfma
            unsigned w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[k]) : "v"(q[k]), "v"(a.x));
            const mcvd::px_bf16x8 x8 = __builtin_bit_cast(mcvd::px_bf16x8, (f32x4){__builtin_bit_cast(float, w[0]), __builtin_bit_cast(float, w[1]), __builtin_bit_cast(float, w[2]), __builtin_bit_cast(float, w[3])});
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x8, x8, acc, 0, 0, 0);
        } else if (K == 9) {   // v_mfma_f32_32x32x16_f16 (new on gfx950, 128-bit A / B operands)
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            const h8 x8 = __builtin_bit_cast(h8, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x8, x8, acc, 0, 0, 0);
        } else if (K == 10) {  // v_mfma_f32_32x32x8_f16 (gfx908..., 64-bit A / B operands)
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 x4 = __builtin_bit_cast(h4, a);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x8f16(x4, x4, acc, 0, 0, 0);
        } else if (K == 11) {  // v_mfma_f32_16x16x32_bf16 (new on gfx950, 128-bit operands, 4-register accumulator)
            const mcvd::px_bf16x8 x8 = __builtin_bit_cast(mcvd::px_bf16x8, q);
            f32x4 a4 = {acc[0], acc[1], acc[2], acc[3]};
#pragma unroll
            for (int k = 0; k < 8; ++k) a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x8, x8, a4, 0, 0, 0);
            acc[0] = a4[0]; acc[1] = a4[1]; acc[2] = a4[2]; acc[3] = a4[3];
        } else if (K == 12) {  // v_mfma_f32_32x32x8_bf16_1k (gfx90a..., 64-bit operands)
            typedef short s4 __attribute__((ext_vector_type(4)));
            const s4 x4 = __builtin_bit_cast(s4, a);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(x4, x4, acc, 0, 0, 0);
        } else if (K == 13) {  // v_mfma_f32_32x32x16_fp8_fp8 (gfx940..., 64-bit operands)
            const long x1 = __builtin_bit_cast(long, a);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(x1, x1, acc, 0, 0, 0);
        }
    }
    float r = a.x + c.x + c.y + q.x + q.y + __builtin_bit_cast(float, u);
    for (int i = 0; i < 16; ++i) r += acc[i];
    if (r == 12345.678f) sink[t] = r;
}

__global__ void compare_kernel(const unsigned* y, const unsigned* ref, long n, unsigned* bad, unsigned* first) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        if (y[i] != ref[i]) { const unsigned k = atomicAdd(bad, 1u); if (k < 32) first[k] = (unsigned)i; }
}

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    const int grid = getenv("GRID") ? atoi(getenv("GRID")) : 48, ldskb = getenv("LDSKB") ? atoi(getenv("LDSKB")) : 68, iters = getenv("ITER") ? atoi(getenv("ITER")) : 64;
    const char* vars = getenv("VARS") ? getenv("VARS") : "0 1 2 3 4 5 6 7";
    const int B = 3, heads = 2, C = 192, HW = 1024;
    const long nq = (long)B * 3 * C * HW, no = (long)B * C * HW, n4 = 48L * 256 * 4;      // victim: 48 x 256 threads x 4 float4 each
    std::vector<float> hq(nq), hx(n4 * 4), hc(1024 * 2);
    unsigned seed = 9;
    for (auto& v : hq) v = frand(seed);
    for (auto& v : hx) v = 2.0f * frand(seed);
    for (size_t i = 0; i < hc.size(); i += 2) { hc[i] = 1.0f + 0.5f * frand(seed); hc[i + 1] = 0.25f + 0.5f * frand(seed); }
    float *qkv, *out, *x, *coef, *y, *ref;
    unsigned *bad, *first;
    CK(hipMalloc(&qkv, nq * 4)); CK(hipMalloc(&out, no * 4)); CK(hipMalloc(&x, n4 * 16)); CK(hipMalloc(&coef, hc.size() * 4)); CK(hipMalloc(&y, n4 * 16)); CK(hipMalloc(&ref, n4 * 16));
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&first, 32 * 4));
    CK(hipMemcpy(qkv, hq.data(), nq * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), n4 * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(coef, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sv, sa;
    CK(hipStreamCreate(&sv)); CK(hipStreamCreate(&sa));
    int aggr = 0;      // 0 the attention kernel, 1.. a synthetic one
    const int aiters = getenv("AITER") ? atoi(getenv("AITER")) : 4000, agrid = getenv("AGRID") ? atoi(getenv("AGRID")) : 1024;
    auto aggressor_once = [&]() {
        switch (aggr) {
            case 0: return mcvd::launch_attention_h2(qkv, out, B, C, heads, HW, sa, 3);
#define SA(K) case K: hipLaunchKernelGGL(synth_aggressor<K>, dim3(agrid), dim3(256), 0, sa, out, aiters); return 0;
            SA(1) SA(2) SA(3) SA(4) SA(5) SA(6) SA(7) SA(8) SA(9) SA(10) SA(11) SA(12) SA(13)
        }
        return -1;
    };
    static const char* aname[14] = {"attn_h2_kernel<3,3>", "v_cvt_pk_bf16_f32 loop", "v_mfma_f32_32x32x16_bf16 loop", "v_pk_* with neg/op_sel_hi", "v_exp_f32 loop", "LDS read/write loop",
                                   "plain v_pk_fma_f32 loop", "v_mfma_f32_32x32x2_f32 loop", "cvt_pk_bf16 -> bf16 MFMA",
                                   "v_mfma_f32_32x32x16_f16 loop", "v_mfma_f32_32x32x8_f16 loop", "v_mfma_f32_16x16x32_bf16 loop", "v_mfma_f32_32x32x8_bf16_1k", "v_mfma_f32_32x32x16_fp8_fp8"};
    const char* aggrs = getenv("AGGRS") ? getenv("AGGRS") : "0";
    if (aggressor_once() != 0) { fprintf(stderr, "aggressor launch failed: %s\n", mcvd::get_error()); return 2; }
    CK(hipStreamSynchronize(sa));
    printf("# seq_victim: grid %d x 256 threads, %d KB dynamic LDS, %d repetitions of the sequence per element; %.1f s per phase\n", grid, ldskb, iters, secs);
    static const char* vname[18] = {"as generated", "no SALU between the selects", "selects under VCC", "four v_fma_f32", "v_pk_fma_f32 without op_sel", "as generated + SiLU tail",
                                   "the two v_pk_fma_f32 only", "same, addend from a copy",
                                   "MFMA in the wave, 0 wait", "MFMA in the wave, s_nop 0", "MFMA in the wave, s_nop 3", "MFMA in the wave, s_nop 7", "MFMA in the wave, s_nop 15",
                                   "v_pk_add/mul x, x swapped",
                                   "v_fma_f32 d, x, c, c (32-bit)", "v_pk_fma d,x,c,c no op_sel", "v_pk_fma d,c,x,c (src0==src2)", "v_pk_mul d,c,c lo*hi"};
#define FOR_VAR(V, ...) case V: { auto kern = seq_victim<V>; __VA_ARGS__; } break;
    for (const char* ap = aggrs; *ap;) {
    if (*ap < '0' || *ap > '9') { ++ap; continue; }
    char* endp;
    aggr = (int)strtol(ap, &endp, 10);
    ap = endp;
    if (aggr > 13) continue;
    for (const char* p = vars; *p;) {
        if (*p < '0' || *p > '9') { ++p; continue; }
        char* vend;
        const int var = (int)strtol(p, &vend, 10);
        p = vend;
        if (var > 17) continue;
        auto victim = [&](float* dst) {
            switch (var) {
#define LV(V) FOR_VAR(V, CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldskb * 1024, sv, (const f32x4*)x, (const f32x2*)coef, (f32x4*)dst, n4, iters, 0ull, ~0ull))
                LV(0) LV(1) LV(2) LV(3) LV(4) LV(5) LV(6) LV(7) LV(8) LV(9) LV(10) LV(11) LV(12) LV(13) LV(14) LV(15) LV(16) LV(17)
            }
            CK(hipGetLastError());
        };
        victim(ref);
        CK(hipStreamSynchronize(sv));
        for (int with = (aggr ? 1 : 0); with <= 1; ++with) {
            std::atomic<bool> stop{false};
            std::atomic<long> n_aggr{0};
            std::thread th;
            if (with) {
                th = std::thread([&]() {
                    CK(hipSetDevice(0));
                    while (!stop.load()) { for (int i = 0; i < 64; ++i) aggressor_once(); n_aggr += 64; (void)hipStreamSynchronize(sa); }
                });
                std::this_thread::sleep_for(std::chrono::milliseconds(200));
            }
            long n = 0, n_bad = 0, el_bad = 0;
            unsigned sig[32]; int nsig = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
                CK(hipMemsetAsync(bad, 0, 4, sv));
                victim(y);
                hipLaunchKernelGGL(compare_kernel, dim3(256), dim3(256), 0, sv, (const unsigned*)y, (const unsigned*)ref, n4 * 4, bad, first);
                unsigned hb = 0;
                CK(hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, sv));
                CK(hipStreamSynchronize(sv));
                ++n;
                if (hb) { if (!n_bad) { nsig = hb < 32 ? (int)hb : 32; CK(hipMemcpy(sig, first, nsig * 4, hipMemcpyDeviceToHost)); } ++n_bad; el_bad += hb; }
            }
            stop = true;
            if (with) th.join();
            printf("victim %d (%-28s) beside %-30s: %6ld of %7ld launches differ, %8ld elements", var, vname[var], with ? aname[aggr] : "alone", n_bad, n, el_bad);
            if (with) printf("  (aggressor launches %ld)", n_aggr.load());
            if (n_bad) {
                printf("\n    first bad launch: element (float4 index : component)");
                for (int i = 0; i < nsig && i < 20; ++i) printf(" %u:%u", sig[i] / 4, sig[i] % 4);
            }
            printf("\n");
            fflush(stdout);
        }
    }
    }
    return 0;
}
