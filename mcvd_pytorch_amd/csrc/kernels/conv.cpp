// conv dispatch (tile-shape heuristic), the simple one-thread-per-output HIP kernel used to triangulate the
// MFMA kernel in tests, and the weight repacking kernel.
#include <stdlib.h>

#include "conv_mfma.h"

namespace mcvd {

int conv_cout_tile(int Cout) {
    if (Cout % 96 == 0) return 3;
    if (Cout % 128 == 0) return 4;
    if (Cout % 64 == 0) return 2;
    return 1;
}

int conv_chunk(int ks) { return ks == 3 ? 16 : 32; }   // packing granule of Cin (the largest chunk any tile shape uses)

#ifdef MCVD_DIAG
static int env_int(const char* name, int dflt) {      // diagnostics build only: the product library reads no environment on a launch path
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
#endif

// which kernel family the last launch_conv_mfma of this thread dispatched to (tests assert a forced shape did not fall back
// silently): 0..3 direct implicit GEMM tile shapes, 4 Winograd, 8 Winograd + K split, 5 / 6 all-DMA 1x1 GEMM (16 / 32 channels),
// 9 the 1x1 GEMM with 64 pixels per wave, 10 / 11 Winograd on the bf16 pipe with three-piece operands (11: + K split), 12 / 13 Winograd on
// the fp16 pipe with two-piece operands (13: + K split), 14 the 1x1 GEMM on the fp16 pipe with two-piece operands, 15 the 1x1 GEMM on
// the bf16 pipe with three-piece operands, 16 / 17 the three-piece bf16 Winograd kernel as persistent workgroups (17: + K split)
static thread_local int g_last_conv_kernel = -1;
int last_conv_kernel() { return g_last_conv_kernel; }

// partials per (sample, channel) the last conv launch of this thread wrote into a.stats (0: the kernel that ran does not emit)
static thread_local int g_last_stats_np = 0;
int last_conv_stats_np() { return g_last_stats_np; }
void set_last_conv_stats_np(int np) { g_last_stats_np = np; }

// 1: the last conv launch of this thread also wrote the (A, B) table of the norm over its output (ConvArgs::gno)
static thread_local int g_last_gn_fused = 0;
int last_conv_gn_fused() { return g_last_gn_fused; }
void set_last_conv_gn_fused(int v) { g_last_gn_fused = v; }

int launch_conv_mfma(const ConvArgs& a, hipStream_t s) {
    g_last_stats_np = 0;
    g_last_gn_fused = 0;
    MCVD_REQUIRE(a.ks == 1 || a.ks == 3, "conv: kernel size %d unsupported", a.ks);
    MCVD_REQUIRE(a.W >= 8 && (a.W & (a.W - 1)) == 0 && a.W <= 256, "conv: W=%d must be a power of two in [8,256]", a.W);
#ifdef MCVD_DIAG
    static const int forced = env_int("MCVD_CONV_SHAPE", -1);             // tile-heuristic experiments (tests/gpu_diag.py)
    static const int min_blocks = env_int("MCVD_CONV_MIN_BLOCKS", 400);
#else
    constexpr int forced = -1;              // (the "conv_shape" context option forces a kernel family; this is the direct kernel's tile)
    constexpr int min_blocks = 400;
#endif
    const long px = (long)a.B * a.H * a.W;
    const int ntc = a.CoutP / (32 * (a.cot > 0 ? a.cot : 1));
    auto blocks = [&](int bpx) { return ((px + bpx - 1) / bpx) * ntc; };
    auto fits = [&](int bpx) { return bpx % a.W == 0 && (bpx / a.W <= a.H ? a.H % (bpx / a.W) == 0 : (bpx / a.W) % a.H == 0); };
    int shape;
    // hints 4 / 5 select the specialised kernels where they apply and fall back to the tile heuristic elsewhere
    if (a.shape_hint == 13) {                                       // f16x2 Winograd with a 2-way K split
        ConvArgs b = a;
        b.ksplit = 2;
        if (conv_wino2h_usable(b)) { g_last_conv_kernel = 13; return launch_conv_wino2h(b, s); }
    }
    if (a.shape_hint == 12 || a.shape_hint == 13) {                 // Winograd on the fp16 matrix pipe, two-piece operands
        ConvArgs b = a;
        b.ksplit = 0;
        if (conv_wino2h_usable(b)) { g_last_conv_kernel = 12; return launch_conv_wino2h(b, s); }
    }
    // bf16x3 Winograd ids: 10 plain, 11 / 18 / 19 = K split in 2 / 4 / 8 parts (small-batch 8x8 / 16x16 layers: more workgroups than
    // (region, cout tile) pairs), 16 / 17 / 20 = persistent workgroups with 1 / 2 / 4 K parts.  A deeper split that a layer cannot take
    // (too few channel chunks) degrades to the next shallower one.
    int h = a.shape_hint;
    if (h == 20) {
        ConvArgs b = a;
        b.ksplit = 4;
        if (conv_wino3p_usable(b)) { g_last_conv_kernel = 20; return launch_conv_wino3p(b, s); }
        h = 17;
    }
    if (h == 19) {
        ConvArgs b = a;
        b.ksplit = 8;
        if (conv_wino3_usable(b)) { g_last_conv_kernel = 19; return launch_conv_wino3(b, s); }
        h = 18;
    }
    if (h == 18) {
        ConvArgs b = a;
        b.ksplit = 4;
        if (conv_wino3_usable(b)) { g_last_conv_kernel = 18; return launch_conv_wino3(b, s); }
        h = 11;
    }
    if (h == 17) {                                                  // persistent bf16x3 Winograd, the two K halves are items
        ConvArgs b = a;
        b.ksplit = 2;
        if (conv_wino3p_usable(b)) { g_last_conv_kernel = 17; return launch_conv_wino3p(b, s); }
    }
    if (h == 16 || h == 17) {                                       // persistent bf16x3 Winograd (one workgroup per CU walks an item range)
        ConvArgs b = a;
        b.ksplit = 0;
        if (conv_wino3p_usable(b)) { g_last_conv_kernel = 16; return launch_conv_wino3p(b, s); }
    }
    if (h == 11 || h == 17) {                                       // bf16x3 Winograd with a 2-way K split
        ConvArgs b = a;
        b.ksplit = 2;
        if (conv_wino3_usable(b)) { g_last_conv_kernel = 11; return launch_conv_wino3(b, s); }
    }
    if (h == 10 || h == 11 || h == 16 || h == 17) {                 // Winograd on the bf16 matrix pipe, operands split three ways
        ConvArgs b = a;
        b.ksplit = 0;
        if (conv_wino3_usable(b)) { g_last_conv_kernel = 10; return launch_conv_wino3(b, s); }
    }
    if (a.shape_hint == 8) {                                        // Winograd with a 2-way K split (fills the CUs on 8x8 layers)
        ConvArgs b = a;
        b.ksplit = 2;
        if (conv_wino_usable(b)) { g_last_conv_kernel = 8; return launch_conv_wino(b, s); }
    }
    if ((a.shape_hint == 4 || a.shape_hint == 8 || (a.shape_hint >= 10 && a.shape_hint <= 13) || (a.shape_hint >= 16 && a.shape_hint <= 20)) && conv_wino_usable(a)) { g_last_conv_kernel = 4; return launch_conv_wino(a, s); }   // Winograd F(2x2,3x3)
    if (a.shape_hint == 15 && conv1x1_h2_supported(a, a.cot, 3)) { g_last_conv_kernel = 15; return launch_conv1x1_h2(a, a.cot, s, 3); }   // 1x1 GEMM, bf16 pipe, three exact pieces
    if (a.shape_hint == 14 && conv1x1_h2_supported(a, a.cot, 2)) { g_last_conv_kernel = 14; return launch_conv1x1_h2(a, a.cot, s, 2); }   // 1x1 GEMM, fp16 pipe, two-piece operands
    if (a.shape_hint == 5 && conv1x1_dma_supported(a, 16)) { g_last_conv_kernel = 5; return launch_conv1x1_dma(a, a.cot, 16, s); }   // all-DMA 1x1 GEMM
    if (a.shape_hint == 6 && conv1x1_dma_supported(a, 32) && a.cot != 9) { g_last_conv_kernel = 6; return launch_conv1x1_dma(a, a.cot, 32, s); }
    if (a.shape_hint == 9 && conv1x1_dma_supported(a, 16, 2)) { g_last_conv_kernel = 9; return launch_conv1x1_dma(a, a.cot, 16, s, 2); }   // 64 pixels per wave
    if (a.cot < 1 || a.cot > 4 || a.CoutP % (32 * a.cot) != 0) {   // a cout tile meant for another kernel: use this one's
        ConvArgs b = a;
        b.cot = conv_cout_tile(a.Cout);
        if (a.CoutP % (32 * b.cot) != 0) b.cot = 1;
        return launch_conv_mfma(b, s);
    }
    const int want = (a.shape_hint >= 0 && a.shape_hint <= 3) ? a.shape_hint : forced;
    if (want >= 0 && want <= 3 && fits(want == 0 ? 256 : want == 1 ? 128 : 64)) shape = want;
    else if (fits(256) && blocks(256) >= min_blocks) shape = 0;
    else if (fits(128) && blocks(128) >= min_blocks) shape = 1;
    else if (fits(64)) shape = 2;
    else if (fits(128)) shape = 1;
    else shape = 0;
    typedef int (*fn_t)(const ConvArgs&, int, hipStream_t);
    static const fn_t tab3[4] = {conv3_cot1, conv3_cot2, conv3_cot3, conv3_cot4};
    static const fn_t tab1[4] = {conv1_cot1, conv1_cot2, conv1_cot3, conv1_cot4};
    g_last_conv_kernel = shape;
    return (a.ks == 3 ? tab3 : tab1)[a.cot - 1](a, shape, s);
}

// ---------------------------------------------------------------------------------------------------------
// Naive direct convolution: one thread per output element, fp32 FMA chain in (ci, tap) order.
__global__ void conv_naive_kernel(ConvArgs a) {
    const long n = (long)a.B * a.Cout * a.H * a.W;
    const int HW = a.H * a.W;
    const int KK = a.ks * a.ks;
    const int pad = a.ks / 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.W);
        const int y = (int)((i / a.W) % a.H);
        const int co = (int)((i / HW) % a.Cout);
        const int b = (int)(i / ((long)HW * a.Cout));
        float acc = 0.0f;
        for (int ci = 0; ci < a.Cin; ++ci) {
            const float* src = (ci < a.C0) ? a.x0 + ((long)b * a.C0 + ci) * HW : a.x1 + ((long)b * a.C1 + (ci - a.C0)) * HW;
            float ca = 1.0f, cb = 0.0f;
            if (a.coef) { ca = a.coef[((long)b * a.Cin + ci) * 2]; cb = a.coef[((long)b * a.Cin + ci) * 2 + 1]; }
            for (int t = 0; t < KK; ++t) {
                const int yy = y + t / a.ks - pad, xx = x + t % a.ks - pad;
                if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
                float v = src[yy * a.W + xx];
                if (a.coef) v = v * ca + cb;
                if (a.act) v = silu_f(v);
                acc = fmaf(a.wp[((long)ci * KK + t) * a.CoutP + co], v, acc);
            }
        }
        float v = acc + a.bias[co];
        if (a.res) v += a.res[i];
        a.y[i] = v * a.out_scale;
    }
}

int launch_conv_naive(const ConvArgs& a, hipStream_t s) {
    const long n = (long)a.B * a.Cout * a.H * a.W;
    const int blocks = (int)((n + 255) / 256 > 65535 * 8 ? 65535 * 8 : (n + 255) / 256);
    hipLaunchKernelGGL(conv_naive_kernel, dim3(blocks), dim3(256), 0, s, a);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// [Cout][Cin][ks][ks] (nn.Conv2d) or [Cin][Cout] (NIN.W, layers.py:538)  ->  wp[(ci*KK + tap)*CoutP + cout_off + co]
__global__ void pack_conv_weight_kernel(const float* w, float* wp, int Cout, int Cin, int KK, int CoutP, int nin,
                                        int cout_off) {
    const long n = (long)Cout * Cin * KK;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int co, ci, t;
        if (nin) {                    // i = ci*Cout + co
            co = (int)(i % Cout); ci = (int)(i / Cout); t = 0;
        } else {                      // i = (co*Cin + ci)*KK + t
            t = (int)(i % KK); ci = (int)((i / KK) % Cin); co = (int)(i / ((long)KK * Cin));
        }
        wp[((long)ci * KK + t) * CoutP + cout_off + co] = w[i];
    }
}

int launch_pack_conv_weight(const float* w, float* wp, int Cout, int Cin, int ks, int CinP, int CoutP, int nin,
                            int cout_off, hipStream_t s) {
    (void)CinP;
    const long n = (long)Cout * Cin * ks * ks;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, s, w, wp, Cout, Cin, ks * ks, CoutP, nin,
                       cout_off);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mcvd
