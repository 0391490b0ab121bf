// Stand-alone reproducer of the co-residency corruption (profiles/r04_two_process_corruption.txt,
// profiles/r05_two_stream_corruption.txt).  NO library: this file compiles the aggressor's source -- attn_h2_kernel<3,3>, the
// three-piece bf16 attention at head dim 96, compiler-scheduled, no inline asm -- together with a 20-line victim and runs them on two streams
// of one process:
//     aggressor thread: launches attn_h2_kernel<3,3> (B = 3, 2 heads of 96 channels, 32 x 32 tokens: the shape of the reports) back to back;
//     victim (main thread), one of
//         FIR+SiLU  fir_up2_kernel as the library launches it behind a GroupNorm (x2 FIR upsampling of silu(A x + B): v_exp_f32 / v_rcp_f32 feeding
//                   FMAs) -- the victim of the round-4 / round-5 reports;
//         FIR raw   the same kernel without the prologue (no transcendental instruction);
//         PK / SC   y[i] = fma(fma(... x[i] ...)), eight dependent v_pk_fma_f32 / v_fma_f32 per element, nothing else;
//     and compares EVERY launch's output bit for bit with the output of the same launch made while the aggressor was idle.
// Prints, per phase: launches made, launches that differ, elements that differ, lane signature of the first bad launch.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/repro_coresident.cpp -o tools/bin/repro_coresident -lpthread
//   discriminators: build the AGGRESSOR with other launch bounds / function attributes and see whether the victim still breaks
//     -DMCVD_AH_LB='__launch_bounds__(256,1)'                           one workgroup per CU budget: up to 512 registers (256 VGPR + AGPRs)
//     -DMCVD_AH_ATTR='__attribute__((amdgpu_waves_per_eu(1,1)))'        one wave per SIMD asked for
//     -DMCVD_AH_ATTR='__attribute__((amdgpu_num_vgpr(128)))'            128 architected VGPRs: the rest of the live range in AGPRs / scratch
//   run:  tools/bin/repro_coresident [seconds per phase, default 4] [path to libmcvd_hip.so]
//   With the library given, two more parties join through its C ABI (include/mcvd_hip.h): the LIBRARY'S build of the aggressor
//   (mcvd_op_attention, option naive_attn = 4) and the victim that still breaks today, the direct 3x3 conv (mcvd_op_conv2d, 96 -> 96 at 64 x 64 behind
//   a GroupNorm + SiLU prologue, option conv_shape = 0: conv_mfma_kernel) -- every (aggressor, victim) pair of {stand-alone, library} is run.
#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "../mcvd_pytorch_amd/csrc/kernels/attention_h2.cpp"
#include "../mcvd_pytorch_amd/csrc/kernels/fir.cpp"          // the report's victim: fir_up2_kernel (register form), with / without its SiLU prologue

namespace mcvd {                 // what the library's other translation units would provide
static char g_err[512];
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); }
const char* get_error() { return g_err; }
int launch_attention_naive(const float*, float*, int, int, int, int, hipStream_t) { return -1; }
int launch_attention_mfma(const float*, float*, int, int, int, int, hipStream_t) { return -1; }
}  // namespace mcvd

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(2); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool PK>
__global__ __launch_bounds__(256) void victim_kernel(const f32x2* __restrict__ x, f32x2* __restrict__ y, long n2, float a0, float a1, float b0, float b1) {
    const f32x2 a = {a0, a1}, b = {b0, b1};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        f32x2 v = x[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(a), "v"(b));
            else { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v.x) : "v"(v.x), "v"(a0), "v"(b0)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v.y) : "v"(v.y), "v"(a1), "v"(b1)); }
        }
        y[i] = v;
    }
}

// AGGR=lds: not the attention kernel but a kernel that only FILLS its LDS (and a few hundred registers) with a recognisable value and exits --
// does a victim pick up what a previous tenant of the CU left behind (LDS or registers it reads before it has written them)?
__global__ __launch_bounds__(256) void polluter_kernel(float* sink, unsigned pattern, int spin) {
    extern __shared__ unsigned pl[];
    for (int i = threadIdx.x; i < 15 * 1024; i += 256) pl[i] = pattern;
    __syncthreads();
    unsigned acc = 0;
    for (int k = 0; k < spin; ++k) acc += pl[(threadIdx.x * 17 + k * 256) % (15 * 1024)];
    if (acc == 12345u) sink[0] = 1.0f;
}

__global__ void compare_kernel(const unsigned* y, const unsigned* ref, long n, unsigned* bad, unsigned* first) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        if (y[i] != ref[i]) {
            const unsigned k = atomicAdd(bad, 1u);
            if (k < 32) first[k] = (unsigned)i;
        }
}

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }

typedef int (*ctx_create_t)(int, void*, void**);
typedef int (*set_option_t)(void*, const char*, int);
typedef int (*op_attention_t)(void*, const float*, float*, int, int, int, int);
typedef int (*op_conv2d_t)(void*, const float*, int, const float*, int, const float*, const float*, int, int, const float*, int, const float*, float, float*, int, int, int);

// DUMP=n: where the first n bad launches of the library victim differ from the clean result -- which images, rows, columns and output channels,
// and what the wrong values look like (zero? the clean value of another place? off by how much?)
static std::vector<float> g_hx, g_hw;      // the library victim's input [3][96][64][64] and weights [96][96][ks][ks] (host copies)
static int g_act = 1, g_coef = 1, g_ks = 3;
static float g_A = 1.0f, g_B = 1.0f;      // VICTIM_AB="A B": the affine of the victim's prologue (the same for every channel)
static float silu_of(float v) { return g_act ? v / (1.0f + expf(-v)) : v; }
static float act_of(float v) { if (g_coef) v = v * g_A + g_B; return silu_of(v); }

// WHAT was wrong in the staged input?  The differences of a bad launch are whole image rows x all output channels x three of every four
// columns: the signature of ONE input value per float4 of one staged row being wrong (its column c feeds the outputs c-1, c, c+1).  At an
// output column c of the residue that all three taps see only through the centre tap, diff[co] = w[co][ci][ty][1] * delta for the culprit
// (ci, ty): project on every candidate, report the best, and look the wrong staged value up among the values the kernel handles.
static void analyze(const std::vector<float>& y, const std::vector<float>& r, int b, int row, int C, int H, int W) {
    if (g_ks != 3 || g_hx.empty()) return;
    int cnt[4] = {0, 0, 0, 0};
    for (int x = 0; x < W; ++x) { bool d = false; for (int c = 0; c < C && !d; ++c) { const long i = (((long)b * C + c) * H + row) * W + x; d = memcmp(&y[i], &r[i], 4) != 0; } cnt[x & 3] += d; }
    int k = -1;      // the residue of the wrong INPUT columns: its neighbours k-1, k+1 differ too, k+2 does not
    for (int q = 0; q < 4; ++q) if (cnt[q] && cnt[(q + 1) & 3] && cnt[(q + 3) & 3] && !cnt[(q + 2) & 3]) k = q;
    printf("      analysis of row %d:%d: differing columns by residue mod 4 = %d %d %d %d -> wrong input columns = %d mod 4\n", b, row, cnt[0], cnt[1], cnt[2], cnt[3], k);
    if (k < 0) return;
    for (int x = k + 4; x < W; x += 20) {
        std::vector<double> d(C);
        double dd = 0;
        for (int c = 0; c < C; ++c) { const long i = (((long)b * C + c) * H + row) * W + x; d[c] = (double)y[i] - (double)r[i]; dd += d[c] * d[c]; }
        double best = 1e30, balpha = 0; int bci = -1, bty = -1;
        for (int ci = 0; ci < 96; ++ci) for (int ty = 0; ty < 3; ++ty) {
            double ww = 0, dw = 0;
            for (int c = 0; c < C; ++c) { const double w = g_hw[(((long)c * 96 + ci) * 3 + ty) * 3 + 1]; ww += w * w; dw += w * d[c]; }
            const double alpha = dw / ww, res = dd - alpha * dw;
            if (res < best) { best = res; balpha = alpha; bci = ci; bty = ty; }
        }
        const int rin = row + bty - 1;
        printf("        column %2d: input channel %2d, tap row %d (input row %d), delta %+.6f, unexplained %.2e of %.2e", x, bci, bty, rin, balpha, best, dd);
        if (rin < 0 || rin >= H) { printf("  (padding row)\n"); continue; }
        const float raw = g_hx[(((long)b * 96 + bci) * H + rin) * W + x], clean = act_of(raw), got = clean + (float)balpha;
        printf("; staged value clean %.6f (raw %.6f) -> got %.6f;", clean, raw, got);
        // candidates: the same position in another channel (a stale staging register holds the previous / next chunk's value), the raw value, zero
        int shown = 0;
        if (fabsf(silu_of(raw) - got) < 2e-5f) { printf(" = f(x): the affine had NO effect;"); ++shown; }
        if (fabsf(silu_of(raw * g_A) - got) < 2e-5f && g_B != 0.0f) { printf(" = f(A x): B was lost;"); ++shown; }
        if (fabsf(silu_of(raw + g_B) - got) < 2e-5f && g_A != 1.0f) { printf(" = f(x + B): A was lost;"); ++shown; }
        if (fabsf(silu_of(raw * g_A + g_A) - got) < 2e-5f && g_A != g_B) { printf(" = f(A x + A): op_sel of the addend lost;"); ++shown; }
        if (fabsf(silu_of(raw * g_B + g_B) - got) < 2e-5f && g_A != g_B) { printf(" = f(B x + B): op_sel of the factor wrong;"); ++shown; }
        for (int cj = 0; cj < 96 && shown < 4; ++cj) {
            const float rj = g_hx[(((long)b * 96 + cj) * H + rin) * W + x];
            if (fabsf(act_of(rj) - got) < 2e-4f) { printf(" = staged value of channel %d at the same place;", cj); ++shown; }
            if (fabsf(rj - got) < 2e-4f) { printf(" = RAW value of channel %d at the same place;", cj); ++shown; }
        }
        for (int bj = 0; bj < 3 && shown < 4; ++bj) for (int rj = 0; rj < H && shown < 4; ++rj) {
            if (bj == b && rj == rin) continue;
            const float v = g_hx[(((long)bj * 96 + bci) * H + rj) * W + x];
            if (fabsf(act_of(v) - got) < 2e-5f) { printf(" = staged value of the same channel at image %d row %d;", bj, rj); ++shown; }
        }
        if (fabsf(got) < 2e-4f) printf(" = 0;");
        if (!shown) printf(" no match among the candidates;");
        printf("\n");
    }
}

static void dump_diff(const float* dy, const float* dref, int B, int C, int H, int W) {
    const long n = (long)B * C * H * W;
    std::vector<float> y(n), r(n);
    CK(hipMemcpy(y.data(), dy, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r.data(), dref, n * 4, hipMemcpyDeviceToHost));
    long bad = 0, zero = 0, nan = 0;
    double maxd = 0, sumd = 0;
    std::vector<long> per_b(B, 0), per_c(C, 0), per_row(B * H, 0), per_x(W, 0);
    for (long i = 0; i < n; ++i) {
        if (memcmp(&y[i], &r[i], 4) == 0) continue;
        ++bad;
        const int x = (int)(i % W), yy = (int)(i / W % H), c = (int)(i / ((long)W * H) % C), b = (int)(i / ((long)W * H * C));
        ++per_b[b]; ++per_c[c]; ++per_row[b * H + yy]; ++per_x[x];
        if (y[i] == 0.0f) ++zero;
        if (y[i] != y[i]) ++nan;
        const double d = fabs((double)y[i] - (double)r[i]);
        if (d == d) { sumd += d; if (d > maxd) maxd = d; }
    }
    printf("    DUMP: %ld elements differ (zero %ld, nan %ld), |diff| max %.4g mean %.4g\n      per image:", bad, zero, nan, maxd, bad ? sumd / bad : 0.0);
    for (int b = 0; b < B; ++b) printf(" %ld", per_b[b]);
    printf("\n      rows (image:row=count):");
    for (int i = 0; i < B * H; ++i) if (per_row[i]) printf(" %d:%d=%ld", i / H, i % H, per_row[i]);
    printf("\n      columns with differences:");
    for (int x = 0; x < W; ++x) if (per_x[x]) printf(" %d=%ld", x, per_x[x]);
    printf("\n      output channels with differences:");
    for (int c = 0; c < C; ++c) if (per_c[c]) printf(" %d=%ld", c, per_c[c]);
    // a few samples
    printf("\n      samples (b,c,y,x: got / clean):");
    int shown = 0;
    for (long i = 0; i < n && shown < 12; ++i) if (memcmp(&y[i], &r[i], 4)) {
        printf(" (%d,%d,%d,%d: %.6g / %.6g)", (int)(i / ((long)W * H * C)), (int)(i / ((long)W * H) % C), (int)(i / W % H), (int)(i % W), y[i], r[i]);
        ++shown; i += 997;
    }
    printf("\n");
    for (int i = 0, done = 0; i < B * H && done < 3; ++i) if (per_row[i]) { analyze(y, r, i / H, i % H, C, H, W); ++done; }
}

// (The cause this program helped to find is in profiles/r06_coresident_cause.txt; tools/repro_pk_fma_beside_mfma.cpp is the 100-line version.)
int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    const int dump_max = getenv("DUMP") ? atoi(getenv("DUMP")) : 0;
    void* lib = argc > 2 ? dlopen(argv[2], RTLD_NOW | RTLD_LOCAL) : nullptr;
    if (argc > 2 && !lib) { fprintf(stderr, "dlopen %s: %s\n", argv[2], dlerror()); return 2; }
    const int B = 3, heads = 2, C = 192, HW = 1024;
    const long nq = (long)B * 3 * C * HW, no = (long)B * C * HW;
    const long nv = 3L * 192 * 64 * 64;                                  // victim elements (the fir victim's output size)
    std::vector<float> hq(nq), hx(nv);
    unsigned seed = 9;
    for (auto& v : hq) v = frand(seed);
    for (auto& v : hx) v = frand(seed);
    float *qkv, *out, *x, *y, *ref;
    unsigned *bad, *first;
    CK(hipMalloc(&qkv, nq * 4)); CK(hipMalloc(&out, no * 4)); CK(hipMalloc(&x, nv * 4)); CK(hipMalloc(&y, nv * 4)); CK(hipMalloc(&ref, nv * 4));
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&first, 32 * 4));
    CK(hipMemcpy(qkv, hq.data(), nq * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(x, hx.data(), nv * 4, hipMemcpyHostToDevice));
    hipStream_t sv, sa;
    CK(hipStreamCreate(&sv)); CK(hipStreamCreate(&sa));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&mcvd::attn_h2_kernel<3, 3>)));
    printf("# aggressor attn_h2_kernel<3,3>: %d registers per thread (numRegs), %zu B static LDS, max %d threads per block; build: LB=%s ATTR=%s\n", fa.numRegs,
           fa.sharedSizeBytes, fa.maxThreadsPerBlock,
#define STR2(x) #x
#define STR(x) STR2(x)
           STR(MCVD_AH_LB), STR(MCVD_AH_ATTR));
    void *ctx_v = nullptr, *ctx_a = nullptr;
    op_attention_t lib_attn = nullptr; op_conv2d_t lib_conv = nullptr;
    float *cx = nullptr, *cw = nullptr, *cb = nullptr, *cco = nullptr;
    int vks = 3;
    if (lib) {
        setenv("MCVD_ALLOW_SHARED_DEVICE", "1", 1);
        auto create = (ctx_create_t)dlsym(lib, "mcvd_ctx_create");
        auto setopt = (set_option_t)dlsym(lib, "mcvd_ctx_set_option");
        lib_attn = (op_attention_t)dlsym(lib, "mcvd_op_attention");
        lib_conv = (op_conv2d_t)dlsym(lib, "mcvd_op_conv2d");
        if (!create || !setopt || !lib_attn || !lib_conv) { fprintf(stderr, "library symbols missing\n"); return 2; }
        if (create(0, sv, &ctx_v) || create(0, sa, &ctx_a)) { fprintf(stderr, "mcvd_ctx_create failed\n"); return 2; }
        setopt(ctx_v, "conv_shape", getenv("VICTIM_SHAPE") ? atoi(getenv("VICTIM_SHAPE")) : 0);      // default 0: the direct implicit-GEMM 3x3 kernel (256-pixel tile)
        if (getenv("VICTIM_WDMA")) setopt(ctx_v, "conv_wdma", atoi(getenv("VICTIM_WDMA")));      // 0: weight chunks through registers instead of LDS-DMA
        if (getenv("VICTIM_SHAPE")) printf("# library victim: conv_shape %s, conv_wdma %s, kernel size %s\n", getenv("VICTIM_SHAPE"),
                                           getenv("VICTIM_WDMA") ? getenv("VICTIM_WDMA") : "1", getenv("VICTIM_KS") ? getenv("VICTIM_KS") : "3");
        setopt(ctx_a, "naive_attn", 4);          // the three-piece bf16 attention kernel, whatever the device fence would choose
        vks = getenv("VICTIM_KS") ? atoi(getenv("VICTIM_KS")) : 3;
        g_ks = vks;
        g_act = getenv("VICTIM_ACT") ? atoi(getenv("VICTIM_ACT")) : 1;          // 0: no SiLU in the victim's prologue
        g_coef = getenv("VICTIM_COEF") ? atoi(getenv("VICTIM_COEF")) : 1;       // 0: raw input (no affine, no SiLU: the staging is load -> LDS)
        if (!g_coef) g_act = 0;
        if (getenv("VICTIM_AB")) { sscanf(getenv("VICTIM_AB"), "%f %f", &g_A, &g_B); printf("# library victim affine: A %g, B %g\n", g_A, g_B); }
        if (getenv("VICTIM_ACT") || getenv("VICTIM_COEF")) printf("# library victim prologue: affine %d, SiLU %d\n", g_coef, g_act);
        std::vector<float> h(3L * 96 * 64 * 64), hw(96L * 96 * vks * vks), hc(3L * 96 * 2, 1.0f);
        for (size_t i = 0; i < hc.size(); i += 2) { hc[i] = g_A; hc[i + 1] = g_B; }
        for (auto& v : h) v = 2.0f * frand(seed);
        for (auto& v : hw) v = frand(seed) / 29.0f;
        CK(hipMalloc(&cx, h.size() * 4)); CK(hipMalloc(&cw, hw.size() * 4)); CK(hipMalloc(&cb, 96 * 4)); CK(hipMalloc(&cco, hc.size() * 4));
        CK(hipMemcpy(cx, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(cw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(cb, 0, 96 * 4)); CK(hipMemcpy(cco, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
        g_hx = h; g_hw = hw;
    }
    int aggr_kind = 0;                           // 0 the stand-alone build of the aggressor, 1 the library's
    const bool pollute = getenv("AGGR") && !strcmp(getenv("AGGR"), "lds");
    const unsigned pattern = getenv("PATTERN") ? (unsigned)strtoul(getenv("PATTERN"), nullptr, 16) : 0x7f800000u;      // +Inf
    if (pollute) printf("# aggressor: polluter_kernel (60 KB of LDS filled with 0x%08x), NOT the attention kernel\n", pattern);
    auto aggressor_once = [&]() {
        if (pollute) { hipLaunchKernelGGL(polluter_kernel, dim3(2048), dim3(256), 60 * 1024, sa, out, pattern, 64); return 0; }
        if (aggr_kind == 1) return lib_attn(ctx_a, qkv, out, B, C, heads, HW);
        return mcvd::launch_attention_h2(qkv, out, B, C, heads, HW, sa, 3);
    };
    if (aggressor_once() != 0) { fprintf(stderr, "aggressor launch failed: %s\n", mcvd::get_error()); return 2; }
    CK(hipStreamSynchronize(sa));
    // FIR victim data: x [3, 192, 32, 32] -> y [3, 192, 64, 64] (= nv elements), coefficients (A, B) per (sample, channel)
    float *fx, *fcoef;
    {
        std::vector<float> hfx(3L * 192 * 32 * 32), hco(3L * 192 * 2);
        for (auto& v : hfx) v = 2.0f * frand(seed);
        for (size_t i = 0; i < hco.size(); i += 2) { hco[i] = 1.0f + 0.3f * frand(seed); hco[i + 1] = 0.3f * frand(seed); }
        CK(hipMalloc(&fx, hfx.size() * 4)); CK(hipMalloc(&fcoef, hco.size() * 4));
        CK(hipMemcpy(fx, hfx.data(), hfx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(fcoef, hco.data(), hco.size() * 4, hipMemcpyHostToDevice));
    }
    static const char* vname[5] = {"v_fma_f32", "v_pk_fma_f32", "fir_up2 raw", "fir_up2+SiLU", "LIB conv3x3"};
    static const char* aname[2] = {"attn_h2_kernel<3,3> (this build)", "attn_h2_kernel<3,3> (LIBRARY)"};
    for (aggr_kind = 0; aggr_kind <= (lib ? 1 : 0); ++aggr_kind)
    for (int pk = lib ? 4 : 3; pk >= (getenv("ONLY_LIB") ? 4 : 0); --pk) {
        auto victim = [&](float* dst) {
            if (pk == 4) { if (lib_conv(ctx_v, cx, 96, nullptr, 0, cw, cb, 96, vks, g_coef ? cco : nullptr, g_act, nullptr, 1.0f, dst, 3, 64, 64)) { fprintf(stderr, "lib conv failed\n"); exit(2); } }
            else if (pk == 3) { if (mcvd::launch_fir2(fx, fcoef, 1, 1, dst, 3, 192, 32, 32, nullptr, nullptr, nullptr, nullptr, sv, 1)) { fprintf(stderr, "fir: %s\n", mcvd::get_error()); exit(2); } }
            else if (pk == 2) { if (mcvd::launch_fir2(fx, nullptr, 0, 1, dst, 3, 192, 32, 32, nullptr, nullptr, nullptr, nullptr, sv, 1)) { fprintf(stderr, "fir: %s\n", mcvd::get_error()); exit(2); } }
            else if (pk) hipLaunchKernelGGL(victim_kernel<true>, dim3(2048), dim3(256), 0, sv, (const f32x2*)x, (f32x2*)dst, nv / 2, 1.0009765625f, 0.99951171875f, 0.03125f, -0.0625f);
            else hipLaunchKernelGGL(victim_kernel<false>, dim3(2048), dim3(256), 0, sv, (const f32x2*)x, (f32x2*)dst, nv / 2, 1.0009765625f, 0.99951171875f, 0.03125f, -0.0625f);
        };
        victim(ref);
        CK(hipStreamSynchronize(sv));
        for (int with = (aggr_kind == 1 ? 1 : 0); with <= 1; ++with) {
            std::atomic<bool> stop{false};
            std::atomic<long> n_aggr{0};
            std::thread th;
            if (with) {
                th = std::thread([&]() {
                    CK(hipSetDevice(0));
                    while (!stop.load()) {
                        for (int i = 0; i < 64; ++i) aggressor_once();
                        n_aggr += 64;
                        hipStreamSynchronize(sa);
                    }
                });
                std::this_thread::sleep_for(std::chrono::milliseconds(200));
            }
            long n = 0, n_bad = 0, el_bad = 0;
            unsigned sig[32]; int nsig = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
                CK(hipMemsetAsync(bad, 0, 4, sv));
                victim(y);
                hipLaunchKernelGGL(compare_kernel, dim3(1024), dim3(256), 0, sv, (const unsigned*)y, (const unsigned*)ref, nv, bad, first);
                unsigned hb = 0;
                CK(hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, sv));
                CK(hipStreamSynchronize(sv));
                ++n;
                if (hb) {
                    if (!n_bad) { nsig = hb < 32 ? (int)hb : 32; CK(hipMemcpy(sig, first, nsig * 4, hipMemcpyDeviceToHost)); }
                    if (pk == 4 && n_bad < dump_max) dump_diff(y, ref, 3, 96, 64, 64);
                    ++n_bad; el_bad += hb;
                }
            }
            stop = true;
            if (with) th.join();
            printf("victim %-12s beside %-34s: %6ld of %7ld launches differ, %8ld elements", vname[pk], with ? aname[aggr_kind] : "alone", n_bad, n, el_bad);
            if (with) printf("  (aggressor launches %ld)", n_aggr.load());
            if (n_bad) {
                printf("\n    first bad launch, element indices (lane = index / 2 %% 64 of its wave):");
                for (int i = 0; i < nsig && i < 16; ++i) printf(" %u(l%u)", sig[i], (sig[i] / 2) % 64);
            }
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
