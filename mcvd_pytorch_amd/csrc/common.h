// Internal declarations shared by the kernel translation units and the host executor.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>


namespace mcvd {

// x * a + b as ONE scalar-lane fused multiply-add, never packed.  Over a float4 with (a, b) in a register pair hipcc writes
//     v_pk_fma_f32 vD, vX, v[n:n+1], v[n:n+1] op_sel:[0,0,1] op_sel_hi:[1,0,1]
// and on gfx950 that form -- ONE VGPR pair read as src1 and src2 -- returns x * a + 0 in its low half for one 16-lane pass while a wave of
// ANOTHER kernel on the same SIMD executes a matrix instruction with 128-bit operands (v_mfma_f32_32x32x16_bf16 / _f16,
// v_mfma_f32_16x16x32_bf16): the "co-residency corruption" of rounds 4-6 (profiles/r06_coresident_cause.txt; 60-line reproducer
// tools/repro_pk_fma_beside_mfma.cpp).  Same rounding as the packed form (both fused).  tools/check_vop3p_dual_read.py fails the build if
// any kernel of the library contains the form.
__device__ __forceinline__ float fma_unpacked(float x, float a, float b) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(a), "v"(b));
    return r;
}

void set_error(const char* fmt, ...);
const char* get_error();

#define MCVD_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            mcvd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -2;                                                                        \
        }                                                                                     \
    } while (0)

#define MCVD_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            mcvd::set_error(__VA_ARGS__);  \
            return -1;                     \
        }                                  \
    } while (0)

// hipFuncSetAttribute (dynamic-LDS opt-in) is a per-DEVICE property of a kernel: each launcher keeps one of these per
// instantiation and raises the limit the first time it launches on a device.  Lock-free: ctxs of different devices may be
// driven from different threads (include/mcvd_hip.h threading contract).
struct PerDeviceOnce {
    std::atomic<unsigned long long> mask{0};
    // true exactly once per device (the caller then sets the attribute; a racing second caller on the same device may
    // also see true -- setting the attribute twice is harmless)
    bool first_use() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
        const unsigned long long bit = 1ull << dev;
        return (mask.load(std::memory_order_acquire) & bit) == 0;
    }
    void done() {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) mask.fetch_or(1ull << dev, std::memory_order_release);
    }
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// ------------------------------------------------------------------ conv (implicit GEMM on fp32 MFMA)
// Packed weight layout: wp[(cin * KK + tap) * CoutP + cout], cin padded to CinP (multiple of the kernel's
// channel chunk) and cout padded to CoutP (multiple of the cout tile) with zeros.
// GroupNorm (A, B) table a conv writes for the norm that consumes its output, where its LAST pass sees whole (sample, group) slabs: the
// second pass of a K-split Winograd layer (gn.cpp: ksplit_reduce_gn_kernel).  coef == NULL: off.  Fields as GnArgs.
struct GnOut {
    float* coef;            // [B][Cout][2]
    int groups;
    float eps;
    int mode;
    const float* p0;
    const float* p1;
    int emb_stride, emb_off;
};

struct ConvArgs {
    const float* x0;
    const float* x1;      // second source of a virtual channel concat (may be null when C1 == 0)
    int C0, C1;
    const float* coef;    // [B][Cin][2] per-(sample,channel) affine (A,B), or null
    int act;              // SiLU after the affine
    const float* wp;      // packed weights
    const float* wpw;     // Winograd-transformed weights, operand-major (conv_wino.cpp: pack_wino_weight_kernel; 3x3 only), or null
    const float* wph;     // the weights pre-split into two fp16 pieces + header: 3x3 Winograd-transformed (conv_wino2h.cpp:
                          // launch_pack_wino2h_weight) or 1x1 (conv1x1_h2.cpp: launch_pack_conv1x1_h2), or null
    const float* wpb;     // the weights pre-split into three bf16 pieces (exact; no header): 3x3 Winograd-transformed (conv_wino3.cpp:
                          // launch_pack_wino3_weight) or 1x1 (launch_pack_conv1x1_h2 with np = 3), or null
    const float* bias;    // [Cout]
    const float* res;     // residual [B][Cout][H][W] or null
    float out_scale;
    float* y;             // [B][Cout][H][W]
    int B, Cin, CinP, Cout, CoutP, H, W;
    int ks;               // 1 or 3
    int cot;              // cout tile in units of 32 channels (1..4); CoutP % (32*cot) == 0
    int shape_hint;       // -1 auto; 0/1/2 force the 256/128/64-pixel tile, 3 split-K+WDB, 4 Winograd (8: with a 2-way K split), 5 / 6 all-DMA 1x1 GEMM (16 / 32 channels per chunk),
                          // 9 the 1x1 GEMM with 64 pixels per wave, 10 Winograd on the bf16 pipe with three-piece operands (11: with a 2-way K split),
                          // 12 Winograd on the fp16 pipe with two-piece operands (13: with a 2-way K split), 14 the 1x1 GEMM on the fp16 pipe
                          // with two-piece operands, 15 the 1x1 GEMM on the bf16 pipe with three-piece operands, 16 / 17 the three-piece bf16 Winograd
                          // kernel as persistent workgroups (17: with a 2-way K split)
    int ksplit;           // Winograd kernel only: 2 = two workgroups per tile contract half the input channels each into
                          // `part`, a second pass sums the halves; 0/1 = off
    float* part;          // ksplit >= 2: scratch for the partial results, ksplit * B*Cout*H*W floats
    size_t part_floats;   // capacity of `part` in floats: a K split that needs more is not usable (the *_usable tests refuse it)
    int wdma;             // 1: stage weight chunks by LDS-DMA (default), 0: through registers
    int pgrid;            // persistent Winograd kernel (conv_wino3p.cpp) only: > 0 = number of workgroups (context option "persist_grid": tests
                          // drive long item ranges and sample changes with a few workgroups); 0 = one per CU
    unsigned long long* dbg;   // optional: per-block phase cycle counters [n_blocks][8] (diagnostics), else null
    // GroupNorm statistics from the producer's epilogue (layerspp.py:518-549 consumes them): when non-null, the kernel writes for
    // every (sample, cout) `np` partial pairs (sum, M2 about the partial's own mean) over disjoint pixel sets of HW / np pixels
    // each, stats[((b * Cout + co) * np + p) * 2 + {0, 1}], of the FINAL output values.  np depends on the kernel that runs
    // (last_conv_stats_np() reports it; 0 = this kernel does not emit, the consumer then reads the tensor itself).
    float* stats;
    // SPADE prologue (Winograd kernel only): gamma | beta maps [B][2*Cin][H][W] of the conditioning frames and the temb pair
    // (1 + scale, shift) [B][Cin][2] (may be null); requires coef (plain GroupNorm coefficients) and act
    const float* gb;
    const float* coef2;
    // q|k|v projection of an attention block (conv1x1_h2.cpp, KV): when non-null the K and V thirds of the output are written HERE as
    // three-piece bf16 LDS images per (sample, head, key tile) instead of as fp32 rows of y: 3 * kv_C * HW dwords per sample
    float* kv_img;
    int kv_C, kv_D;       // channels of one of q / k / v; head dim
    // conv1x1_h2.cpp only (shape id 23, the stem as a GEMM): 1 = x0 / x1 are the RAW sources of a 3x3 conv (C0 + C1 channels, H x W) and
    // the GEMM's B operand is their im2col, staged in LDS by the kernel itself: K row k = c * 9 + t <-> x[c][y + t / 3 - 1][x + t % 3 - 1]
    // (zero outside the image), rows 9 (C0 + C1) .. Cin - 1 zero; Cin = CinP = the padded row count.  No HBM `col` tensor.
    int im2col;
    // the norm over THIS conv's output (single source), finalized by the K-split reduce pass when the launch has one (8 x 8 / 16 x 16
    // planes): last_conv_gn_fused() tells whether it happened
    GnOut gno;
};
int last_conv_gn_fused();
void set_last_conv_gn_fused(int v);
// the K-split partials of this launch fit the scratch the caller handed over
inline bool conv_part_fits(const ConvArgs& a) {
    return a.part != nullptr && (size_t)(a.ksplit < 2 ? 2 : a.ksplit) * a.B * a.Cout * a.H * a.W <= a.part_floats;
}
bool ksplit_reduce_gn_usable(const ConvArgs& a);
int launch_ksplit_reduce_gn(const ConvArgs& a, hipStream_t s);
bool conv1x1_h2_kv_supported(const ConvArgs& a, int cot);
void set_last_conv_stats_np(int np);          // (launchers)
int last_conv_stats_np();                     // partials per (sample, channel) the thread's last conv launch wrote to a.stats; 0 = none
int conv_cout_tile(int Cout);                 // 32-channel units per block along Cout
int conv_chunk(int ks);                       // input-channel chunk the MFMA kernel consumes per stage
int launch_conv_mfma(const ConvArgs& a, hipStream_t s);
int launch_conv_naive(const ConvArgs& a, hipStream_t s);
int last_conv_kernel();                       // kernel family of this thread's last launch_conv_mfma (see conv.cpp)
// 3x3 convs with a handful of channels on one side as 1x1 GEMMs (conv_gemm_forms.cpp): shape ids 22 (taps as outputs + shift-and-add) / 23 (im2col)
int launch_im2col3x3(const float* x0, int C0, const float* x1, int C1, float* col, int B, int H, int W, int K, hipStream_t s);
int launch_taps_shift_add(const float* z, const float* bias, const float* res, float scale, float* y, int B, int Cout, int H, int W, hipStream_t s);
int launch_pack_conv_gemm_form(const float* w, float* wp, int Cout, int Cin, int form, int CoutP, hipStream_t s);
// Winograd F(2x2,3x3) kernel (conv_wino.cpp): tile shape id 4 of the dispatcher
bool conv_wino_supported(int ks, int H, int W);      // geometry only (decides whether transformed weights are packed at all)
int conv_wino_cout_tile(int Cout);
bool conv_wino_usable(const ConvArgs& a);            // shape id 4 applies to this launch
int launch_conv_wino(const ConvArgs& a, hipStream_t s);
int launch_wino_ksplit_reduce(const ConvArgs& a, hipStream_t s);      // second pass of the 2-way K split (both Winograd kernels)
// the same convolution with the channel contraction on the bf16 matrix pipe at fp32 accuracy (three exact bf16 pieces per operand,
// six piece products, weights pre-split at pack time; conv_wino3.cpp): tile shape ids 10 / 11 (11: 2-way K split); needs ConvArgs::wpb
bool conv_wino3_usable(const ConvArgs& a);
int launch_conv_wino3(const ConvArgs& a, hipStream_t s);
long conv_wino3_weight_floats(int CinP, int CoutP);           // size of the wpb buffer, in floats
// the same kernel as PERSISTENT workgroups (one per CU, each walks a contiguous range of (region, cout tile[, K half]) items, the K loop's
// staging pipeline runs on across item boundaries; conv_wino3p.cpp): tile shape ids 16 / 17 (17: 2-way K split); bit-identical results
bool conv_wino3p_usable(const ConvArgs& a);
int launch_conv_wino3p(const ConvArgs& a, hipStream_t s);
int launch_pack_wino3_weight(const float* w, float* wb, int Cout, int Cin, int CinP, int CoutP, hipStream_t s);    // wb zero-filled
// the same convolution on the fp16 matrix pipe, operands split into two fp16 pieces (22-bit operands, three piece products; weights
// pre-split at pack time; conv_wino2h.cpp): tile shape ids 12 / 13 (13: 2-way K split); needs ConvArgs::wph
bool conv_wino2h_usable(const ConvArgs& a);
int launch_conv_wino2h(const ConvArgs& a, hipStream_t s);
long conv_wino2h_weight_floats(int CinP, int CoutP);          // size of the wph buffer (header + pieces), in floats
int launch_pack_wino2h_weight(const float* w, float* wh, int Cout, int Cin, int CinP, int CoutP, hipStream_t s);   // wh zero-filled
// 1x1 GEMM on the 16-bit matrix pipes with split operands (conv1x1_h2.cpp), cout tile a.cot = 1..4: np = 3 three bf16 pieces
// (fp32-equivalent; tile shape id 15; needs ConvArgs::wpb), np = 2 two fp16 pieces (tile shape id 14; needs ConvArgs::wph); the
// pieces come from the packed fp32 matrix wp (launch_pack_conv1x1_h2)
bool conv1x1_h2_supported(const ConvArgs& a, int cot, int np);
int launch_conv1x1_h2(const ConvArgs& a, int cot, hipStream_t s, int np);
long conv1x1_h2_weight_floats(int CinP, int CoutP, int np);
int launch_pack_conv1x1_h2(const float* wp, float* wh, int CinP, int CoutP, hipStream_t s, int np);      // np = 2: wh[0] zero on entry
// all-DMA 1x1 GEMM (conv1x1_dma.cpp): tile shape ids 5 / 6; cot_req <= 0 picks the default cout tile
bool conv1x1_dma_supported(const ConvArgs& a, int ck, int pxw = 1);     // ck: channels per chunk, 16 (shape id 5) or 32 (shape id 6); pxw = 2: 256-pixel tiles (shape id 9, ck 16)
int conv1x1_dma_cout_tile(int CoutP);
int launch_conv1x1_dma(const ConvArgs& a, int cot_req, int ck, hipStream_t s, int pxw = 1);
// [Cout][Cin][3][3] -> operand-major transformed weights (zero-filled destination of CinP*16*CoutP floats)
int launch_pack_wino_weight(const float* w, float* up, int Cout, int Cin, int CinP, int CoutP, hipStream_t s);
// repack reference-layout weights [Cout][Cin][ks][ks] (or NIN [Cin][Cout] when nin=1) -> packed layout above
int launch_pack_conv_weight(const float* w, float* wp, int Cout, int Cin, int ks, int CinP, int CoutP, int nin,
                            int cout_off, hipStream_t s);

// ------------------------------------------------------------------ GroupNorm -> affine coefficients
struct GnArgs {
    const float* x0;
    const float* x1;
    int C0, C1;
    int groups;
    float eps;
    int mode;           // 0 plain, 1 temb scale/shift, 2 affine weight/bias
    const float* p0;    // mode1: emb [B][emb_stride]; mode2: weight [C]
    const float* p1;    // mode2: bias [C]
    int emb_stride, emb_off;
    float* coef;        // [B][C][2]
    int B, HW;
};
int launch_gn_coef(const GnArgs& a, hipStream_t s);
// The same coefficients from producer-side partial statistics (ConvArgs::stats) instead of a pass over the tensor: partial p of
// channel c of source i holds (sum, M2) over HW / np_i pixels; combined per group with the pairwise (Chan et al.) update, in a
// fixed order.  x0 / x1 of `a` are not read.
int launch_gn_finalize(const GnArgs& a, const float* st0, int np0, const float* st1, int np1, hipStream_t s);

// ------------------------------------------------------------------ attention
bool attention_mfma_supported(int C, int heads, int HW);      // head dim 32..256 in steps of 32, HW % 32 == 0; else the general kernel
int launch_attention_mfma(const float* qkv, float* out, int B, int C, int heads, int HW, hipStream_t s);
int launch_attention_naive(const float* qkv, float* out, int B, int C, int heads, int HW, hipStream_t s);
// the flash kernel on the 16-bit matrix pipes with split operands (attention_h2.cpp; np = 3 three bf16 pieces, fp32-equivalent; np = 2
// two fp16 pieces): head dim 32..128 in steps of 32, HW % 32 == 0
bool attention_h2_supported(int C, int heads, int HW);
int launch_attention_h2(const float* qkv, float* out, int B, int C, int heads, int HW, hipStream_t s, int np);
// the same kernel (np = 3) fed with K and V ALREADY SPLIT by the q|k|v projection's epilogue (ConvArgs::kv_img): the tiles reach the LDS by
// LDS-DMA, no VALU instruction touches them; head dims 32 / 64 / 96.  Bit-identical to launch_attention_h2(.., 3).
bool attn_h2p_supported(int C, int heads, int HW);
int launch_attention_h2p(const float* qkv, const float* kv_img, float* out, int B, int C, int heads, int HW, hipStream_t s);
// which attention kernel a launch takes: mode = the "naive_attn" option (0 auto: the fp16-pipe kernel when f16x2 is on and it applies,
// else the fp32 flash kernel; 1 the one-thread-per-query kernel; 2 the fp32 flash kernel; 3 the fp16-pipe kernel where it applies)
int launch_attention(int mode, int f16x2, int bf16x3, const float* qkv, float* out, int B, int C, int heads, int HW, hipStream_t s);

// ------------------------------------------------------------------ FIR resampling
// form 0: the LDS-strip kernels where their geometry applies (power-of-two widths 8..256), else the register forms; 1: register forms only
int launch_fir2(const float* x, const float* coef, int act, int up, float* y, int B, int C, int H, int W,
                const float* gamma, const float* beta, const float* coef2, float* y_raw, hipStream_t s, int form = 0);
int launch_upfirdn2d(const float* in, const float* kernel_dev, int kh, int kw, int up, int down, int pad0, int pad1,
                     float* out, int NC, int H, int W, int oh, int ow, hipStream_t s);
int launch_spade_apply(const float* x0, int C0, const float* x1, int C1, const float* coef, const float* gb,
                       const float* coef2, float* y, int B, int HW, hipStream_t s);
int launch_coef2(const float* emb, int emb_stride, int emb_off, float* coef2, int B, int C, hipStream_t s);
// all tables of a forward at once: desc_dev[3 t] = {arena offset per sample, emb_off, C} (device, int64); coef2 table t = arena + off * B
int launch_coef2_all(const float* emb, int emb_stride, const long long* desc_dev, int ntab, int cmax, float* arena, int B, hipStream_t s);
int launch_nearest_resize(const float* in, float* out, int BC, int H, int W, int oh, int ow, hipStream_t s);

// ------------------------------------------------------------------ time embedding
// silu_temb[b][:] = SiLU( W1 * SiLU(W0 * emb(t_b) + b0) + b1 )      (ncsnpp_more.py:273-280 + layerspp.py:521)
// labels: int64 [B], or float [B] when labels_f32 (fractional timesteps of the F-PNDM sampler)
// cond_emb: the row continues with SiLU(emb_table[mask_b][0 .. nf/2)) at column 4*nf (mask NULL = 1); out_stride = row length
int launch_temb_mlp(const void* labels, int labels_f32, const float* freqs, const float* w0, const float* b0, const float* w1,
                    const float* b1, float* silu_temb, int B, int nf, int out_stride, const float* emb_table, const int32_t* mask,
                    hipStream_t s);
// out[b][n] = sum_k act[b][k] * wt[k][n] + bias[n]   (all Dense_0 projections of a forward in one launch)
int launch_dense_all(const float* act, const float* wt, const float* bias, float* out, int B, int K, int N,
                     hipStream_t s);
int launch_transpose_into(const float* w, float* wt, int rows, int cols, int ld_out, int col_off, hipStream_t s);

// ------------------------------------------------------------------ sampler elementwise
int launch_fill_labels(int64_t* labels, int64_t value, int B, hipStream_t s);
int launch_fill_labels_f(float* labels, float value, int B, hipStream_t s);
// kind 0 ddpm / 1 ddim.  noise may be null (then c_noise must be 0 or use_philox != 0).
int launch_sampler_update(int kind, float* x, const float* eps, const float* noise, float c_x0a, float c_x0b,
                          float c_mean0, float c_mean1, float c_noise, int clip, int64_t n, int use_philox,
                          uint64_t seed, uint64_t sample_offset, uint64_t draw, int64_t per_sample, hipStream_t s);
int launch_renoise(float* x, const float* noise, float ca, float cb, int64_t n, int use_philox, uint64_t seed,
                   uint64_t sample_offset, uint64_t draw, int64_t per_sample, hipStream_t s);
int launch_axpy_out(float* x, const float* eps, float c, int64_t n, hipStream_t s);   // x -= c * eps
int launch_nonfinite_flag(const float* v, int64_t n, int* flag, hipStream_t s);          // *flag = 1 if any v[i] is Inf / NaN (f16x2 range guard)
int launch_pack_frames_u8(const float* in, uint8_t* out, int B, int T, int C, int HW, hipStream_t s);
int launch_randn(float* out, uint64_t seed, uint64_t sample_offset, uint64_t draw, int B, int64_t per_sample,
                 hipStream_t s);
// standardised gamma variates (models/__init__.py:273-276, :319-322): out = (g - kt) / sd with g = raw[i] when raw != NULL, else
// g = theta * Gamma(k) drawn from the Philox stream (Marsaglia-Tsang); kt = k * theta and sd = sqrt(1 - alpha) as fp32 scalars
int launch_gamma_noise(float* out, const float* raw, float k, float theta, float kt, float sd, uint64_t seed,
                       uint64_t sample_offset, uint64_t draw, int B, int64_t per_sample, hipStream_t s);
// noise_in_cond (ncsnpp_more.py:755-768): out[b] = sqrt(alphas[t_b]) * cond[b] + sqrt(1 - alphas[t_b]) * z[b], t_b = labels[b]
int launch_cond_noise(const float* cond, const float* z, const float* alphas_dev, const int64_t* labels, int T, float* out, int B,
                      int64_t per_sample, hipStream_t s);
// out = scale * sum_k w[k] * in[k], k < nin <= 4 (left-to-right, separately rounded, as the reference's tensor expression);
// out may alias an input
int launch_lincomb(float* out, const float* const* in, const float* w, float scale, int nin, int64_t n, hipStream_t s);
// F-PNDM transfer (models/pndm.py:19-33): out = clip?( x + d * (c1 * x - c2 * e) )
int launch_pndm_transfer(float* out, const float* x, const float* e, float d, float c1, float c2, int clip, int64_t n,
                         hipStream_t s);

}  // namespace mcvd
