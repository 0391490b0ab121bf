// Rate of DEPENDENT v_mfma_f32_32x32x16_bf16 chains on gfx950: NA accumulators used round robin by one wave (NA = 1: every MFMA takes the
// previous one's result as SrcC), one or two waves per SIMD.  The S product of the attention kernels is ONE chain of 36 per key tile.
// Reported: shader cycles (s_memtime) per MFMA of a wave.   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_chain.cpp -o /tmp/ubench_mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NA>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int n, float y) {
    extern __shared__ float lds[];
    f32x16 acc[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = y * (a + r);
    bf16x8 A, B;
#pragma unroll
    for (int j = 0; j < 8; ++j) { A[j] = (__bf16)(y + j); B[j] = (__bf16)(y - j); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 12 / NA; ++u)
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[a][r];
    if (sum == 12345.678f) lds[threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NA>
static void run(int threads, const char* what) {
    unsigned long long* d; hipMalloc(&d, 256 * 8 * 8);
    const int n = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<NA>, dim3(256), dim3(threads), 100 * 1024, 0, d, n, 0.001f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double s = 0; int c = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { s += (double)h[b * 8 + w]; ++c; }
    // s_memtime counts at 100 MHz on gfx950: report per-MFMA time in ns too
    printf("%-34s accumulators %d: %.2f s_memtime ticks per MFMA of a wave (x 10 ns = %.1f ns)\n", what, NA, s / c / (n * 12.0), s / c / (n * 12.0) * 10.0);
    hipFree(d);
}

int main() {
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    run<1>(256, "one wave per SIMD,"); run<2>(256, "one wave per SIMD,"); run<3>(256, "one wave per SIMD,"); run<4>(256, "one wave per SIMD,");
    run<1>(512, "two waves per SIMD,"); run<2>(512, "two waves per SIMD,"); run<3>(512, "two waves per SIMD,"); run<4>(512, "two waves per SIMD,");
    return 0;
}
