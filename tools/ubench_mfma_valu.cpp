// Does matrix work of one wave overlap with VALU work of the OTHER wave of the same SIMD on gfx950, and with VALU work of the SAME wave?
// One workgroup of 512 threads per CU (100 KB of LDS requested: nothing else fits) = two waves per SIMD.  Waves 0-3 ("M") issue NM
// v_mfma_f32_32x32x16_bf16 over three accumulators; waves 4-7 ("V") issue NV VALU instructions of one kind over eight registers.
// Reported: shader cycles (s_memtime) of the M waves and of the V waves when only one half works and when both do, plus a same-wave
// run (F fillers behind every MFMA of the M waves, the V waves idle).
// Build + run (on the GPU box): hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_valu.cpp -o /tmp/ubench_mfma_valu && /tmp/ubench_mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

// KIND: 0 v_fma_f32, 1 v_pk_add_f32, 2 v_exp_f32, 3 v_cvt_pk_bf16_f32
template <int KIND>
__device__ __forceinline__ void valu8(float (&x)[8], f2 (&p)[8], float y) {
    if (KIND == 0)
        asm volatile("v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\n"
                     "v_fma_f32 %4, %4, %8, %8\nv_fma_f32 %5, %5, %8, %8\nv_fma_f32 %6, %6, %8, %8\nv_fma_f32 %7, %7, %8, %8\n"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
    else if (KIND == 1)
        asm volatile("v_pk_add_f32 %0, %0, %0\nv_pk_add_f32 %1, %1, %1\nv_pk_add_f32 %2, %2, %2\nv_pk_add_f32 %3, %3, %3\n"
                     "v_pk_add_f32 %4, %4, %4\nv_pk_add_f32 %5, %5, %5\nv_pk_add_f32 %6, %6, %6\nv_pk_add_f32 %7, %7, %7\n"
                     : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
    else if (KIND == 2)
        asm volatile("v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\n"
                     "v_exp_f32 %4, %4\nv_exp_f32 %5, %5\nv_exp_f32 %6, %6\nv_exp_f32 %7, %7\n"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    else
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %8\nv_cvt_pk_bf16_f32 %1, %1, %8\nv_cvt_pk_bf16_f32 %2, %2, %8\nv_cvt_pk_bf16_f32 %3, %3, %8\n"
                     "v_cvt_pk_bf16_f32 %4, %4, %8\nv_cvt_pk_bf16_f32 %5, %5, %8\nv_cvt_pk_bf16_f32 %6, %6, %8\nv_cvt_pk_bf16_f32 %7, %7, %8\n"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
}

// mode bit 0: the M waves work; bit 1: the V waves work.  fill = VALU instructions (v_fma_f32) behind every MFMA of the M waves.
template <int KIND, int FILL>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int mode, int nm3, int nv8, float y) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6;
    float x[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; p[i] = f2{x[i], x[i] + 1}; }
    f32x16 acc[3];
    for (int a = 0; a < 3; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(float)(threadIdx.x & 7); bv[i] = (__bf16)1.0f; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode & 1)
            for (int it = 0; it < nm3; ++it) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[a]) : "v"(av), "v"(bv));
                    if (FILL >= 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[0]) : "v"(y));
                    if (FILL >= 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[1]) : "v"(y));
                    if (FILL >= 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[2]) : "v"(y));
                    if (FILL >= 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[3]) : "v"(y));
                    if (FILL >= 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[4]) : "v"(y));
                    if (FILL >= 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[5]) : "v"(y));
                    if (FILL >= 7) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[6]) : "v"(y));
                    if (FILL >= 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[7]) : "v"(y));
                }
            }
    } else {
        if (mode & 2)
            for (int it = 0; it < nv8; ++it) valu8<KIND>(x, p, y);
    }
    asm volatile("s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y;
    for (int a = 0; a < 3; ++a) s += acc[a][0];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (t1 - t0) + (s == 12345.f);
    if (threadIdx.x == 9999) lds[0] = s;
}

template <int KIND, int FILL>
static void run(const char* name, unsigned long long* d, int mode, int nm3, int nv8) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND, FILL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, FILL>), dim3(256), dim3(512), 100 * 1024, 0, d, mode, nm3, nv8, 1.0001f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
    m /= 1024; v /= 1024;
    printf("%-44s M waves %8.0f cycles (%5.1f per MFMA)   V waves %8.0f cycles (%5.2f per VALU)\n", name, m, (mode & 1) ? m / (3.0 * nm3) : 0.0, v,
           (mode & 2) ? v / (8.0 * nv8) : 0.0);
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 1 << 20);
    const int nm3 = 400;                       // 1200 MFMAs: 38.4 k cycles at 32 per MFMA
    printf("# two waves per SIMD; M = 1200 x v_mfma_f32_32x32x16_bf16 (three accumulators), V = VALU of one kind\n");
    run<0, 0>("MFMA waves alone", d, 1, nm3, 0);
#define PAIR(KIND, NAME, NV8)                                                       \
    run<KIND, 0>(NAME " waves alone", d, 2, nm3, NV8);                              \
    run<KIND, 0>("MFMA waves + " NAME " waves", d, 3, nm3, NV8);
    PAIR(0, "v_fma_f32 x9600", 1200)
    PAIR(0, "v_fma_f32 x4800", 600)
    PAIR(1, "v_pk_add_f32 x9600", 1200)
    PAIR(2, "v_exp_f32 x2400", 300)
    PAIR(3, "v_cvt_pk_bf16_f32 x9600", 1200)
    printf("# same wave: F x v_fma_f32 behind every MFMA (V waves idle)\n");
    run<0, 1>("MFMA + 1 filler", d, 1, nm3, 0);
    run<0, 2>("MFMA + 2 fillers", d, 1, nm3, 0);
    run<0, 4>("MFMA + 4 fillers", d, 1, nm3, 0);
    run<0, 6>("MFMA + 6 fillers", d, 1, nm3, 0);
    run<0, 8>("MFMA + 8 fillers", d, 1, nm3, 0);
    printf("# same wave fillers WITH the partner wave issuing v_fma_f32 x9600\n");
    run<0, 4>("MFMA + 4 fillers, partner VALU", d, 3, nm3, 1200);
    return 0;
}
