// What makes a v_mfma_f32_32x32x16_bf16 cost 47 cycles instead of 32 inside attn_h2p_kernel (profiles/r05_attention_experiments.txt, MFMA-count
// ablation)?  One or two waves per SIMD, each issuing N MFMAs on ONE accumulator chain with
//   mode 0: the same A and B operand every time (tools/ubench_mfma_chain.cpp);
//   mode 1: the six-product pattern of the three-piece arithmetic -- A in {a0,a1,a2}, B in {b0,b1,b2} from registers;
//   mode 2: mode 1 with the three A operands RE-READ from the LDS (ds_read_b128 x 3) in front of every six products, as the kernels do;
//   mode 3: mode 2 + 24 VALU instructions (v_fma_f32) behind every six products (the partner wave's softmax stands for them).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_pattern.cpp -o tools/bin/ubench_mfma_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mm(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int n, unsigned y) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0x3f803f80u + y;
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    u32x4 a[3], b[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) { a[p] = u32x4{y + p, y, y, y} + 0x3c003c00u; b[p] = u32x4{y, y + p, y, y} + 0x3c003c00u; }
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)(y + j);
    const u32x4* src = reinterpret_cast<const u32x4*>(lds) + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
        if (MODE >= 2) {
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = src[((i & 3) * 3 + p) * 64];
        }
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) acc = mm(a[0], b[0], acc);
        } else {
            acc = mm(a[0], b[2], acc); acc = mm(a[2], b[0], acc); acc = mm(a[1], b[1], acc);
            acc = mm(a[0], b[1], acc); acc = mm(a[1], b[0], acc); acc = mm(a[0], b[0], acc);
        }
        if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = __builtin_fmaf(f[j], 1.0001f, 0.5f);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc[r];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += f[j];
    if (sum == 12345.678f) lds[threadIdx.x] = (unsigned)sum;
    if (lane == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static void run(int threads) {
    unsigned long long* d; (void)hipMalloc(&d, 256 * 8 * 8);
    const int n = 4000;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 100 * 1024, 0, d, n, 1u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 100 * 1024, 0, d, n, 1u);
    hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // MFMAs per SIMD = waves per SIMD * 6 n; report wall ns per MFMA and SIMD (32 cycles at 2.2 GHz = 14.5 ns)
    const double per = ms * 1e6 / ((threads / 256) * 6.0 * n);
    printf("mode %d, %d wave(s) per SIMD: %.1f us, %.2f ns per MFMA and SIMD\n", MODE, threads / 256, ms * 1e3, per);
    hipFree(d);
}

int main() {
    run<0>(256); run<1>(256); run<2>(256); run<3>(256);
    run<0>(512); run<1>(512); run<2>(512); run<3>(512);
    return 0;
}
