// gfx950 (MI355X): v_pk_fma_f32 that reads ONE VGPR pair as src1 and src2 loses its low addend while another kernel's wave on the same SIMD
// executes a matrix instruction with 128-bit A / B operands.  Stand-alone reproducer: two kernels, two streams, no library.
//
//   victim     v_pk_fma_f32 vD, vX, vC, vC op_sel:[0,0,1] op_sel_hi:[1,0,1]        D.lo = X.lo * C.lo + C.hi,  D.hi = X.hi * C.lo + C.hi
//              -- what hipcc generates for  x * c.x + c.y  over consecutive floats with (c.x, c.y) loaded as a float2;
//   control    the same instruction with the addend read from a COPY of the pair (vD, vX, vC, vC');
//   aggressor  a loop of v_mfma_f32_16x16x32_bf16 (or, control, v_mfma_f32_32x32x2_f32) on other data, other stream.
//
// Every victim launch is compared bit for bit with the same launch made while the aggressor was idle.  Measured on MI355X (ROCm 7.2,
// profiles/r06_coresident_cause.txt): victim beside the bf16 MFMA loop: EVERY launch differs -- in the wrong elements D.lo = X.lo * C.lo,
// 16 consecutive lanes at a time; control victim, or fp32 MFMA loop: none; victim alone, or the MFMA inside the victim's own wave: none.
// Other aggressor instructions that do it: v_mfma_f32_32x32x16_bf16, v_mfma_f32_32x32x16_f16 (the three with 128-bit A / B operands);
// that do not: v_mfma_f32_32x32x8_f16, _32x32x8_bf16_1k, _32x32x16_fp8_fp8, _32x32x2_f32, VALU / transcendental / LDS loops.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/repro_pk_fma_beside_mfma.cpp -o tools/bin/repro_pk_fma_beside_mfma -lpthread
//   tools/bin/repro_pk_fma_beside_mfma [seconds per phase]          exit code 1 when the victim differed beside the bf16 MFMA loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool COPY>
__global__ __launch_bounds__(256) void victim(const f32x2* __restrict__ x, const f32x2* __restrict__ coef, f32x2* __restrict__ y, long n, int reps) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const f32x2 v = x[i], c = coef[(i >> 4) & 1023];
        f32x2 c2 = c, acc = {0.f, 0.f};
        asm volatile("" : "+v"(c2));                       // a second register pair with the same contents
        for (int k = 0; k < reps; ++k) {
            f32x2 r;
            if (COPY) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(v), "v"(c), "v"(c2));
            else asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(v), "v"(c));
            acc += r;
        }
        y[i] = acc;
    }
}

template <bool BF16>
__global__ __launch_bounds__(256, 2) void aggressor(float* sink, int iters) {
    const f32x4 q = {1.0f + threadIdx.x * 1e-3f, 0.5f, 0.25f, -0.75f};
    f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
    f32x16 a16;
    for (int i = 0; i < 16; ++i) a16[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (BF16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, q), __builtin_bit_cast(bf16x8, q), a4, 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) a16 = __builtin_amdgcn_mfma_f32_32x32x2f32(q.x, q.y, a16, 0, 0, 0);
        }
    }
    const float r = a4[0] + a4[1] + a4[2] + a4[3] + a16[0] + a16[5];
    if (r == 12345.678f) sink[threadIdx.x] = r;
}

__global__ void compare(const unsigned* y, const unsigned* ref, long n, unsigned* bad) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L)
        if (y[i] != ref[i]) atomicAdd(bad, 1u);
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    const long n = 48L * 256 * 8;
    std::vector<float> hx(2 * n), hc(2048);
    unsigned s = 9;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hx) v = 2.0f * rnd();
    for (size_t i = 0; i < hc.size(); i += 2) { hc[i] = 1.0f + 0.5f * rnd(); hc[i + 1] = 0.25f + 0.5f * rnd(); }
    float *x, *coef, *y, *ref, *sink;
    unsigned* bad;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&coef, hc.size() * 4)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&ref, n * 8)); CK(hipMalloc(&sink, 1024)); CK(hipMalloc(&bad, 4));
    CK(hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(coef, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sv, sa;
    CK(hipStreamCreate(&sv)); CK(hipStreamCreate(&sa));
    long differed_beside_bf16 = 0;
    for (int copy = 0; copy <= 1; ++copy)
        for (int aggr = 0; aggr <= 2; ++aggr) {              // 0 alone, 1 beside the bf16 MFMA loop, 2 beside the fp32 MFMA loop
            auto launch = [&](float* dst) {
                if (copy) hipLaunchKernelGGL(victim<true>, dim3(48), dim3(256), 0, sv, (const f32x2*)x, (const f32x2*)coef, (f32x2*)dst, n, 64);
                else hipLaunchKernelGGL(victim<false>, dim3(48), dim3(256), 0, sv, (const f32x2*)x, (const f32x2*)coef, (f32x2*)dst, n, 64);
            };
            launch(ref);
            CK(hipStreamSynchronize(sv));
            std::atomic<bool> stop{false};
            std::thread th;
            if (aggr) {
                th = std::thread([&]() {
                    CK(hipSetDevice(0));
                    while (!stop.load()) {
                        for (int i = 0; i < 64; ++i) {
                            if (aggr == 1) hipLaunchKernelGGL(aggressor<true>, dim3(512), dim3(256), 0, sa, sink, 400);
                            else hipLaunchKernelGGL(aggressor<false>, dim3(512), dim3(256), 0, sa, sink, 100);
                        }
                        (void)hipStreamSynchronize(sa);
                    }
                });
                std::this_thread::sleep_for(std::chrono::milliseconds(200));
            }
            long launches = 0, differ = 0, elements = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
                unsigned hb = 0;
                CK(hipMemsetAsync(bad, 0, 4, sv));
                launch(y);
                hipLaunchKernelGGL(compare, dim3(256), dim3(256), 0, sv, (const unsigned*)y, (const unsigned*)ref, 2 * n, bad);
                CK(hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, sv));
                CK(hipStreamSynchronize(sv));
                ++launches; differ += hb != 0; elements += hb;
            }
            stop = true;
            if (aggr) th.join();
            printf("v_pk_fma_f32 d, x, %-6s %-34s: %6ld of %6ld launches differ (%ld elements)\n", copy ? "c, c'" : "c, c",
                   aggr == 0 ? "alone" : aggr == 1 ? "beside v_mfma_f32_16x16x32_bf16" : "beside v_mfma_f32_32x32x2_f32", differ, launches, elements);
            if (!copy && aggr == 1) differed_beside_bf16 = differ;
        }
    return differed_beside_bf16 ? 1 : 0;
}
