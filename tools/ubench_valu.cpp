// VALU issue-cost microbenchmark for gfx950: cycles per wave64 instruction on one SIMD, one wave per SIMD and two.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.cpp -o tools/bin/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define REP 64
#define BODY(ASM)                                                                          \
    for (int it = 0; it < iters; ++it) {                                                   \
        _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) {                              \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)           \
                         : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), \
                           "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])  \
                         : "v"(y));                                                        \
        }                                                                                  \
    }
typedef float f2 __attribute__((ext_vector_type(2)));
#define I_ADD(i) "v_add_f32 %" #i ", %" #i ", %16\n"
#define I_AND(i) "v_and_b32 %" #i ", 0xffff0000, %" #i "\n"
#define I_LSHL(i) "v_lshlrev_b32 %" #i ", 16, %" #i "\n"
#define I_CVT(i) "v_cvt_pk_bf16_f32 %" #i ", %" #i ", %16\n"
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %16, %16\n"
#define I_PKADD(i) "v_pk_add_f32 %1" #i ", %1" #i ", %1" #i "\n"
#define I_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define I_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %16, %16\n"
#define I_MOV(i) "v_mov_b32 %" #i ", %16\n"
#define I_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5\n"
template <int K>
__global__ void k(unsigned long long* out, float y, int iters) {
    float x[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; p[i] = f2{x[i], x[i] + 1}; }
    // operands %10..%17 are p[0..7]  ("%1" #i)
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (K == 0) BODY(I_ADD) else if (K == 1) BODY(I_AND) else if (K == 2) BODY(I_LSHL) else if (K == 3) BODY(I_CVT)
    else if (K == 4) BODY(I_PERM) else if (K == 5) {
        for (int it = 0; it < iters; ++it) {
            _Pragma("unroll") for (int r = 0; r < REP / 8; ++r)
                asm volatile("v_pk_add_f32 %0, %0, %0\nv_pk_add_f32 %1, %1, %1\nv_pk_add_f32 %2, %2, %2\nv_pk_add_f32 %3, %3, %3\n"
                             "v_pk_add_f32 %4, %4, %4\nv_pk_add_f32 %5, %5, %5\nv_pk_add_f32 %6, %6, %6\nv_pk_add_f32 %7, %7, %7\n"
                             : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
        }
    } else if (K == 6) BODY(I_EXP) else if (K == 7) BODY(I_RCP) else if (K == 8) BODY(I_FMA) else if (K == 9) BODY(I_MOV) else BODY(I_BFE)
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y;
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) + (s == 12345.f);
}
template <int K> void run(const char* name, unsigned long long* d) {
    const int iters = 200;
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(k<K>, dim3(256), dim3(threads), 0, 0, d, 1.5f, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * threads / 64);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto v : h) m += v; m /= h.size();
        printf("%-22s %d waves/SIMD: %.2f cycles per instruction per wave, %.2f per SIMD\n", name, threads / 256, m / (iters * REP), m / (iters * REP) / (threads / 256));
    }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 1 << 20);
    run<0>("v_add_f32", d); run<1>("v_and_b32 (literal)", d); run<2>("v_lshlrev_b32", d); run<3>("v_cvt_pk_bf16_f32", d);
    run<4>("v_perm_b32", d); run<5>("v_pk_add_f32", d); run<6>("v_exp_f32", d); run<7>("v_rcp_f32", d); run<8>("v_fma_f32", d);
    run<9>("v_mov_b32", d); run<10>("v_bfe_u32", d);
    return 0;
}
