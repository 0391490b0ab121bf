// FIR x2 resampling with the [1,3,3,1] kernel (ResBlock up/down paths, up_or_down_sampling.py:196-258 -> upfirdn2d),
// fused with the optional GroupNorm/temb/SiLU prologue; the reference's generic native op upfirdn2d; nearest resize
// (SPADE segmap, layerspp.py:165).  All HBM-bound elementwise kernels: one thread per 4 consecutive outputs.
#include "../common.h"

namespace mcvd {

__device__ __forceinline__ float silu1(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

struct FirArgs {
    const float* x; const float* coef; int act; int up; float* y; int B, C, H, W;
    float* y_raw;      // optional second output: the same resampling of the RAW input (ResBlock shortcut path), one read
    const float* gamma; const float* beta; const float* coef2;   // SPADE maps live in a [B][2C][H][W] tensor (gamma | beta)
};

struct V2 { float h, r; };    // (prologue-transformed, raw)

// (transformed, raw) value of the input plane at (yy, xx); zero outside (padding applies after the activation)
__device__ __forceinline__ V2 fir_src(const FirArgs& a, const float* plane, long pidx, int yy, int xx, float cA,
                                      float cB, float sA, float sB) {
    V2 o{0.0f, 0.0f};
    if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) return o;
    float v = plane[yy * a.W + xx];
    o.r = v;
    if (a.coef) v = v * cA + cB;
    if (a.gamma) {
        const long bc = pidx / ((long)a.H * a.W);
        const long b = bc / a.C, c = bc - b * a.C;
        const long gi = ((b * 2 * a.C + c) * a.H + yy) * a.W + xx;
        v = v * (1.0f + a.gamma[gi]) + a.beta[gi];
        v = v * sA + sB;
    }
    if (a.act) v = silu1(v);
    o.h = v;
    return o;
}

__global__ __launch_bounds__(256) void fir2_kernel(FirArgs a) {
    const int OH = a.up ? a.H * 2 : a.H / 2, OW = a.up ? a.W * 2 : a.W / 2;
    const int OW4 = OW >> 2;
    const long n4 = (long)a.B * a.C * OH * OW4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int ox0 = (int)(i % OW4) * 4;
        const int oy = (int)((i / OW4) % OH);
        const long bc = i / ((long)OW4 * OH);
        const long pidx = bc * a.H * a.W;
        const float* plane = a.x + pidx;
        float cA = 1.f, cB = 0.f, sA = 1.f, sB = 0.f;
        if (a.coef) { cA = a.coef[bc * 2]; cB = a.coef[bc * 2 + 1]; }
        if (a.coef2) { sA = a.coef2[bc * 2]; sB = a.coef2[bc * 2 + 1]; }
        float o[4], r[4];
        if (a.up) {
            // per axis: y[2n] = x[n-1]/4 + 3x[n]/4 ; y[2n+1] = 3x[n]/4 + x[n+1]/4     
            const int ny = oy >> 1;
            const int ya = (oy & 1) ? ny : ny - 1, yb = (oy & 1) ? ny + 1 : ny;      // rows with weights (wa, wb)
            const float wya = (oy & 1) ? 0.75f : 0.25f, wyb = (oy & 1) ? 0.25f : 0.75f;
            const int nx0 = ox0 >> 1;
            float col[4], colr[4];                                                  // vertical blend of cols nx0-1..nx0+2
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = nx0 - 1 + j;
                const V2 p = fir_src(a, plane, pidx, ya, xx, cA, cB, sA, sB), q = fir_src(a, plane, pidx, yb, xx, cA, cB, sA, sB);
                col[j] = wya * p.h + wyb * q.h;
                colr[j] = wya * p.r + wyb * q.r;
            }
            o[0] = 0.25f * col[0] + 0.75f * col[1];
            o[1] = 0.75f * col[1] + 0.25f * col[2];
            o[2] = 0.25f * col[1] + 0.75f * col[2];
            o[3] = 0.75f * col[2] + 0.25f * col[3];
            r[0] = 0.25f * colr[0] + 0.75f * colr[1];
            r[1] = 0.75f * colr[1] + 0.25f * colr[2];
            r[2] = 0.25f * colr[1] + 0.75f * colr[2];
            r[3] = 0.75f * colr[2] + 0.25f * colr[3];
        } else {
            // per axis: y[m] = (x[2m-1] + 3x[2m] + 3x[2m+1] + x[2m+2]) / 8
            float col[10], colr[10];
            {
                // the 4 x 10 input window as two aligned float4 + the two edge columns per row, unconditional with clamped coordinates
                // (16 loads instead of 40 predicated ones), zeroed AFTER the activation where outside.  SPADE: the gamma and beta maps
                // of the same window the same way -- until round 4 that case took the element-wise path of fir_src: 120 predicated
                // scalar loads per thread, 95 us for a launch that takes 24 this way (profiles/r04_rocprofv3_cfg4_bair_big_spade.txt)
                const int x0 = 2 * ox0;                                   // multiple of 8
                const bool in_l = x0 > 0, in_r = x0 + 8 < a.W;
                const float* gplane = nullptr;
                const float* bplane = nullptr;
                if (a.gamma) {
                    const long b = bc / a.C, c = bc - b * a.C;
                    gplane = a.gamma + (b * 2 * a.C + c) * (long)a.H * a.W;
                    bplane = a.beta + (b * 2 * a.C + c) * (long)a.H * a.W;
                }
                float hv[4][10], rv[4][10];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int yy = 2 * oy - 1 + r;
                    const bool in_y = yy >= 0 && yy < a.H;
                    const int roff = min(max(yy, 0), a.H - 1) * a.W;
                    const int xl = max(x0 - 1, 0), xr = min(x0 + 8, a.W - 1);
                    const float* rowp = plane + roff;
                    const float4 q0 = *reinterpret_cast<const float4*>(rowp + x0);
                    const float4 q1 = *reinterpret_cast<const float4*>(rowp + x0 + 4);
                    const float vals[10] = {rowp[xl], q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, rowp[xr]};
                    float gv[10], bv[10];
                    if (a.gamma) {
                        const float* grow = gplane + roff;
                        const float* brow = bplane + roff;
                        const float4 g0 = *reinterpret_cast<const float4*>(grow + x0), g1 = *reinterpret_cast<const float4*>(grow + x0 + 4);
                        const float4 b0 = *reinterpret_cast<const float4*>(brow + x0), b1 = *reinterpret_cast<const float4*>(brow + x0 + 4);
                        const float gt[10] = {grow[xl], g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, grow[xr]};
                        const float bt[10] = {brow[xl], b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, brow[xr]};
#pragma unroll
                        for (int j = 0; j < 10; ++j) { gv[j] = gt[j]; bv[j] = bt[j]; }
                    }
#pragma unroll
                    for (int j = 0; j < 10; ++j) {
                        const bool in = in_y && (j == 0 ? in_l : j == 9 ? in_r : true);
                        float v = vals[j];
                        if (a.coef) v = v * cA + cB;
                        if (a.gamma) {                                   // fir_src's order
                            v = v * (1.0f + gv[j]) + bv[j];
                            v = v * sA + sB;
                        }
                        if (a.act) v = silu1(v);
                        hv[r][j] = in ? v : 0.0f;
                        rv[r][j] = in ? vals[j] : 0.0f;
                    }
                }
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    col[j] = (hv[0][j] + 3.0f * hv[1][j] + 3.0f * hv[2][j] + hv[3][j]) * 0.125f;
                    colr[j] = (rv[0][j] + 3.0f * rv[1][j] + 3.0f * rv[2][j] + rv[3][j]) * 0.125f;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = (col[2 * k] + 3.0f * col[2 * k + 1] + 3.0f * col[2 * k + 2] + col[2 * k + 3]) * 0.125f;
                r[k] = (colr[2 * k] + 3.0f * colr[2 * k + 1] + 3.0f * colr[2 * k + 2] + colr[2 * k + 3]) * 0.125f;
            }
        }
        *reinterpret_cast<float4*>(a.y + (bc * OH + oy) * OW + ox0) = make_float4(o[0], o[1], o[2], o[3]);
        if (a.y_raw) *reinterpret_cast<float4*>(a.y_raw + (bc * OH + oy) * OW + ox0) = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// FIR x2 UP, one thread per (input row ny, 4 input columns): the 3 x 6 input window (rows ny-1..ny+1, columns nx0-1..nx0+4) as one
// aligned float4 + the two edge columns per row, unconditional with clamped coordinates, activated ONCE per element and zeroed
// after the activation where outside -> 2 x 8 outputs (rows 2ny, 2ny+1; columns 2nx0..2nx0+7) as four float4 stores.
// (The generic path above loads 8 predicated scalars and activates them for every 4 outputs: 152 us -> see DESIGN.md for the
// 32x32 -> 64x64, 192-channel, B=64 launch.)
__global__ __launch_bounds__(256) void fir_up2_kernel(FirArgs a) {
    const int W4 = a.W >> 2;
    const int OW = 2 * a.W;
    const long n = (long)a.B * a.C * a.H * W4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int nx0 = (int)(i % W4) * 4;
        const int ny = (int)((i / W4) % a.H);
        const long bc = i / ((long)W4 * a.H);
        const float* plane = a.x + bc * a.H * a.W;
        float cA = 1.f, cB = 0.f, sA = 1.f, sB = 0.f;
        if (a.coef) { cA = a.coef[bc * 2]; cB = a.coef[bc * 2 + 1]; }
        if (a.coef2) { sA = a.coef2[bc * 2]; sB = a.coef2[bc * 2 + 1]; }
        const float* gpl = nullptr;
        const float* bpl = nullptr;
        if (a.gamma) {
            const long b = bc / a.C, c = bc - b * a.C;
            gpl = a.gamma + (b * 2 * a.C + c) * a.H * a.W;
            bpl = a.beta + (b * 2 * a.C + c) * a.H * a.W;
        }
        const bool in_l = nx0 > 0, in_r = nx0 + 4 < a.W;
        float hv[3][6], rv[3][6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = ny - 1 + r;
            const bool in_y = yy >= 0 && yy < a.H;
            const int yc = min(max(yy, 0), a.H - 1);
            const float* rowp = plane + yc * a.W;
            const float4 q = *reinterpret_cast<const float4*>(rowp + nx0);
            const float vals[6] = {rowp[max(nx0 - 1, 0)], q.x, q.y, q.z, q.w, rowp[min(nx0 + 4, a.W - 1)]};
            float gv[6], bv[6];
            if (a.gamma) {
                const float4 g4 = *reinterpret_cast<const float4*>(gpl + yc * a.W + nx0);
                const float4 b4 = *reinterpret_cast<const float4*>(bpl + yc * a.W + nx0);
                gv[0] = gpl[yc * a.W + max(nx0 - 1, 0)]; gv[1] = g4.x; gv[2] = g4.y; gv[3] = g4.z; gv[4] = g4.w; gv[5] = gpl[yc * a.W + min(nx0 + 4, a.W - 1)];
                bv[0] = bpl[yc * a.W + max(nx0 - 1, 0)]; bv[1] = b4.x; bv[2] = b4.y; bv[3] = b4.z; bv[4] = b4.w; bv[5] = bpl[yc * a.W + min(nx0 + 4, a.W - 1)];
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const bool in = in_y && (j == 0 ? in_l : j == 5 ? in_r : true);
                float v = vals[j];
                if (a.coef) v = v * cA + cB;
                if (a.gamma) {
                    v = v * (1.0f + gv[j]) + bv[j];
                    v = v * sA + sB;
                }
                if (a.act) v = silu1(v);
                hv[r][j] = in ? v : 0.0f;
                rv[r][j] = in ? vals[j] : 0.0f;
            }
        }
        // per axis: y[2n] = x[n-1]/4 + 3x[n]/4 ; y[2n+1] = 3x[n]/4 + x[n+1]/4     (SURVEY 9.4), rows first then columns,
        // with the operation order of the generic path (wya * p + wyb * q; 0.25 * c0 + 0.75 * c1 ...)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            float col[6], colr[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                col[j] = rr == 0 ? 0.25f * hv[0][j] + 0.75f * hv[1][j] : 0.75f * hv[1][j] + 0.25f * hv[2][j];
                colr[j] = rr == 0 ? 0.25f * rv[0][j] + 0.75f * rv[1][j] : 0.75f * rv[1][j] + 0.25f * rv[2][j];
            }
            float o[8], r8[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[2 * k] = 0.25f * col[k] + 0.75f * col[k + 1];
                o[2 * k + 1] = 0.75f * col[k + 1] + 0.25f * col[k + 2];
                r8[2 * k] = 0.25f * colr[k] + 0.75f * colr[k + 1];
                r8[2 * k + 1] = 0.75f * colr[k + 1] + 0.25f * colr[k + 2];
            }
            const long orow = (bc * 2 * a.H + 2 * ny + rr) * OW + 2 * nx0;
            *reinterpret_cast<float4*>(a.y + orow) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(a.y + orow + 4) = make_float4(o[4], o[5], o[6], o[7]);
            if (a.y_raw) {
                *reinterpret_cast<float4*>(a.y_raw + orow) = make_float4(r8[0], r8[1], r8[2], r8[3]);
                *reinterpret_cast<float4*>(a.y_raw + orow + 4) = make_float4(r8[4], r8[5], r8[6], r8[7]);
            }
        }
    }
}

int launch_fir2(const float* x, const float* coef, int act, int up, float* y, int B, int C, int H, int W,
                const float* gamma, const float* beta, const float* coef2, float* y_raw, hipStream_t s) {
    MCVD_REQUIRE((up ? W * 2 : W / 2) % 4 == 0 && H % 2 == 0, "fir2: H=%d W=%d unsupported", H, W);
    FirArgs a{x, coef, act, up, y, B, C, H, W, y_raw, gamma, beta, coef2};
    if (up && W % 4 == 0) {
        const long n = (long)B * C * H * (W / 4);
        const int blocks = (int)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
        hipLaunchKernelGGL(fir_up2_kernel, dim3(blocks), dim3(256), 0, s, a);
        MCVD_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const long n4 = (long)B * C * (up ? H * 2 : H / 2) * ((up ? W * 2 : W / 2) / 4);
    const int blocks = (int)((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256);
    hipLaunchKernelGGL(fir2_kernel, dim3(blocks), dim3(256), 0, s, a);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

// Generic upfirdn2d (op/upfirdn2d.py:163-204): zero-insert upsample, pad/crop, correlate with the flipped kernel, decimate.
__global__ void upfirdn2d_kernel(const float* in, const float* k, int kh, int kw, int up, int down, int pad0, float* out,
                                 int NC, int H, int W, int oh, int ow) {
    const long n = (long)NC * oh * ow;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
        const long pc = i / ((long)ow * oh);
        const float* plane = in + pc * H * W;
        float acc = 0.0f;
        for (int a = 0; a < kh; ++a) {
            const int qy = oy * down + a - pad0;                  // coordinate in the zero-inserted image
            if (qy < 0 || qy >= H * up || qy % up) continue;
            for (int b = 0; b < kw; ++b) {
                const int qx = ox * down + b - pad0;
                if (qx < 0 || qx >= W * up || qx % up) continue;
                acc += k[(kh - 1 - a) * kw + (kw - 1 - b)] * plane[(qy / up) * W + qx / up];
            }
        }
        out[i] = acc;
    }
}

int launch_upfirdn2d(const float* in, const float* kernel_dev, int kh, int kw, int up, int down, int pad0, int pad1,
                     float* out, int NC, int H, int W, int oh, int ow, hipStream_t s) {
    (void)pad1;
    const long n = (long)NC * oh * ow;
    const int blocks = (int)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
    hipLaunchKernelGGL(upfirdn2d_kernel, dim3(blocks), dim3(256), 0, s, in, kernel_dev, kh, kw, up, down, pad0, out, NC, H,
                       W, oh, ow);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

// SPADE modulation + temb scale/shift + SiLU (layerspp.py:171, :535, :548) on a (virtual concat) tensor:
//   y = silu( ((A x + B) (1 + gamma) + beta) * sA + sB ),  gamma|beta in gb:[B][2C][HW], (A,B) = plain GroupNorm coefficients
struct SpadeArgs {
    const float* x0; const float* x1; int C0, C1; const float* coef; const float* gb; const float* coef2; float* y; int B, HW;
};
__global__ __launch_bounds__(256) void spade_apply_kernel(SpadeArgs a) {
    const int C = a.C0 + a.C1;
    const int HW4 = a.HW >> 2;
    const long n4 = (long)a.B * C * HW4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int p4 = (int)(i % HW4);
        const long bc = i / HW4;
        const int c = (int)(bc % C);
        const long b = bc / C;
        const float* src = (c < a.C0) ? a.x0 + (b * a.C0 + c) * a.HW : a.x1 + (b * a.C1 + (c - a.C0)) * a.HW;
        const float4 v = reinterpret_cast<const float4*>(src)[p4];
        const float4 g = reinterpret_cast<const float4*>(a.gb + (b * 2 * C + c) * a.HW)[p4];
        const float4 be = reinterpret_cast<const float4*>(a.gb + (b * 2 * C + C + c) * a.HW)[p4];
        const float cA = a.coef[bc * 2], cB = a.coef[bc * 2 + 1];
        float sA = 1.f, sB = 0.f;
        if (a.coef2) { sA = a.coef2[bc * 2]; sB = a.coef2[bc * 2 + 1]; }
        float4 o;
        o.x = silu1(((v.x * cA + cB) * (1.0f + g.x) + be.x) * sA + sB);
        o.y = silu1(((v.y * cA + cB) * (1.0f + g.y) + be.y) * sA + sB);
        o.z = silu1(((v.z * cA + cB) * (1.0f + g.z) + be.z) * sA + sB);
        o.w = silu1(((v.w * cA + cB) * (1.0f + g.w) + be.w) * sA + sB);
        reinterpret_cast<float4*>(a.y + bc * a.HW)[p4] = o;
    }
}

int launch_spade_apply(const float* x0, int C0, const float* x1, int C1, const float* coef, const float* gb,
                       const float* coef2, float* y, int B, int HW, hipStream_t s) {
    MCVD_REQUIRE(HW % 4 == 0, "spade_apply: HW=%d", HW);
    SpadeArgs a{x0, x1, C0, x1 ? C1 : 0, coef, gb, coef2, y, B, HW};
    const long n4 = (long)B * (a.C0 + a.C1) * (HW / 4);
    const int blocks = (int)((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256);
    hipLaunchKernelGGL(spade_apply_kernel, dim3(blocks), dim3(256), 0, s, a);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- SPADE modulation fused into the normalisation itself (north_star: "SPADE gamma/beta fused into the normalization epilogue";
// layerspp.py:152-173 MySPADE.forward + :530-535 + :543-547).  Until round 5 a SPADE norm was THREE launches: gn_finalize (GroupNorm
// statistics of the producers' epilogue partials -> (A, B) per channel), spade_apply (modulate, activate, materialise) and the conv.
// Here the first two are one: a workgroup owns one (sample, group, pixel part); its first wave reduces the group's partial statistics
// EXACTLY as gn_finalize_kernel does (same partial order, same two passes, same shuffles: the coefficients are bit-identical to that
// kernel's), parks the group's (A, B) in LDS, and all waves stream the group's channels:
//     y = silu( ((A x + B) (1 + gamma) + beta) s1 + b2 )          (s1, b2) = the temb pair (1 + scale, shift), absent for the final norm
// with the same expression as spade_apply_kernel (bit-identical output).  The reduction is repeated by the `parts` workgroups of a
// group: a few hundred bytes of L2-resident partials each.  Groups of more than 64 channels keep the two-launch path (not a shape of
// the reference's configs).
struct SpadeNormArgs {
    const float* x0; const float* x1; int C0, C1; int groups; float eps;
    const float* st0; int np0; const float* st1; int np1;
    const float* gb; const float* coef2; float* y; float* coef_out; int B, HW, parts;
};
constexpr int SN_KEEP = 8;
constexpr int SN_PRE = 4;        // items per thread whose loads are issued in front of the reduction
__global__ __launch_bounds__(256) void spade_norm_apply_kernel(SpadeNormArgs a) {
    __shared__ float sAB[2 * 64];
    const int C = a.C0 + a.C1;
    const int gs = C / a.groups;
    const int part_id = blockIdx.x % a.parts;
    const int bg = blockIdx.x / a.parts;
    const int b = bg / a.groups, g = bg - b * a.groups;
    const int c0 = g * gs;
    const int tid = threadIdx.x, lane = tid & 63;
    // the thread's first SN_PRE items (x, gamma, beta: 16 bytes each) are requested BEFORE the statistics are reduced: their latency
    // passes under the reduction's two dependent round trips instead of behind them (most workgroups have 2-4 items per thread)
    const int HW4 = a.HW >> 2;
    const int per = (HW4 + a.parts - 1) / a.parts;                 // float4 columns of this part, for every channel of the group
    const int p_lo = part_id * per, p_hi = min(HW4, p_lo + per);
    const int span = p_hi - p_lo;
    const int nitems = gs * span;
    float4 pv[SN_PRE], pg[SN_PRE], pb[SN_PRE];
#pragma unroll
    for (int k = 0; k < SN_PRE; ++k) {
        const int i = min(tid + 256 * k, nitems - 1);
        const int cl = i / span, p4 = p_lo + (i - cl * span);
        const int c = c0 + cl;
        const float* src = (c < a.C0) ? a.x0 + ((long)b * a.C0 + c) * a.HW : a.x1 + ((long)b * a.C1 + (c - a.C0)) * a.HW;
        pv[k] = reinterpret_cast<const float4*>(src)[p4];
        pg[k] = reinterpret_cast<const float4*>(a.gb + ((long)b * 2 * C + c) * a.HW)[p4];
        pb[k] = reinterpret_cast<const float4*>(a.gb + ((long)b * 2 * C + C + c) * a.HW)[p4];
    }
    if (tid < 64) {                       // ---- gn_finalize_kernel's reduction, instruction for instruction (mode 0: no affine)
        const int n_in0 = max(0, min(c0 + gs, a.C0) - c0), n_in1 = gs - n_in0;
        const int P0 = n_in0 * a.np0, P = P0 + n_in1 * a.np1;
        const float n0 = (float)(a.HW / max(a.np0, 1)), n1 = (float)(a.HW / max(a.np1, 1));
        auto part = [&](int i, float& sum, float& m2, float& n) {
            if (i < P0) {
                const int cl = i / a.np0, p = i - cl * a.np0;
                const float2 q = *reinterpret_cast<const float2*>(a.st0 + (((long)b * a.C0 + c0 + cl) * a.np0 + p) * 2);
                sum = q.x; m2 = q.y; n = n0;
            } else {
                const int j = i - P0;
                const int cl = j / a.np1, p = j - cl * a.np1;
                const int c1 = max(c0, a.C0) - a.C0 + cl;
                const float2 q = *reinterpret_cast<const float2*>(a.st1 + (((long)b * a.C1 + c1) * a.np1 + p) * 2);
                sum = q.x; m2 = q.y; n = n1;
            }
        };
        float ks[SN_KEEP], km[SN_KEEP], kn[SN_KEEP];
#pragma unroll
        for (int k = 0; k < SN_KEEP; ++k) {
            ks[k] = 0.0f; km[k] = 0.0f; kn[k] = 1.0f;
            if (lane + 64 * k < P) part(lane + 64 * k, ks[k], km[k], kn[k]);
        }
        float tot = 0.0f;
#pragma unroll
        for (int k = 0; k < SN_KEEP; ++k)
            if (lane + 64 * k < P) tot += ks[k];
        for (int i = lane + 64 * SN_KEEP; i < P; i += 64) {
            float sm, m2, n;
            part(i, sm, m2, n);
            tot += sm;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        const float N = (float)gs * (float)a.HW;
        const float mean = tot / N;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < SN_KEEP; ++k)
            if (lane + 64 * k < P) {
                const float d = ks[k] / kn[k] - mean;
                acc += km[k] + kn[k] * d * d;
            }
        for (int i = lane + 64 * SN_KEEP; i < P; i += 64) {
            float sm, m2, n;
            part(i, sm, m2, n);
            const float d = sm / n - mean;
            acc += m2 + n * d * d;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        const float var = acc / N;
        const float rstd = 1.0f / sqrtf(var + a.eps);
        if (lane < gs) {
            const float A = rstd * 1.0f, Bc = 0.0f - mean * rstd * 1.0f;           // gn_finalize: rstd * par0, par1 - mean * rstd * par0 with (1, 0)
            sAB[2 * lane] = A;
            sAB[2 * lane + 1] = Bc;
            if (a.coef_out && part_id == 0) reinterpret_cast<float2*>(a.coef_out)[(long)b * C + c0 + lane] = make_float2(A, Bc);
        }
    }
    __syncthreads();
    for (int i = tid, k = 0; i < nitems; i += 256, ++k) {
        const int cl = i / span, p4 = p_lo + (i - cl * span);
        const int c = c0 + cl;
        const long bc = (long)b * C + c;
        float4 v, gm, be;
        if (k < SN_PRE) {
#pragma unroll
            for (int q = 0; q < SN_PRE; ++q)
                if (q == k) { v = pv[q]; gm = pg[q]; be = pb[q]; }
        } else {
            const float* src = (c < a.C0) ? a.x0 + ((long)b * a.C0 + c) * a.HW : a.x1 + ((long)b * a.C1 + (c - a.C0)) * a.HW;
            v = reinterpret_cast<const float4*>(src)[p4];
            gm = reinterpret_cast<const float4*>(a.gb + ((long)b * 2 * C + c) * a.HW)[p4];
            be = reinterpret_cast<const float4*>(a.gb + ((long)b * 2 * C + C + c) * a.HW)[p4];
        }
        const float cA = sAB[2 * cl], cB = sAB[2 * cl + 1];
        float sA = 1.f, sB = 0.f;
        if (a.coef2) { sA = a.coef2[bc * 2]; sB = a.coef2[bc * 2 + 1]; }
        float4 o;
        o.x = silu1(((v.x * cA + cB) * (1.0f + gm.x) + be.x) * sA + sB);
        o.y = silu1(((v.y * cA + cB) * (1.0f + gm.y) + be.y) * sA + sB);
        o.z = silu1(((v.z * cA + cB) * (1.0f + gm.z) + be.z) * sA + sB);
        o.w = silu1(((v.w * cA + cB) * (1.0f + gm.w) + be.w) * sA + sB);
        reinterpret_cast<float4*>(a.y + bc * a.HW)[p4] = o;
    }
}

bool spade_norm_apply_supported(int C, int groups, int HW) { return groups > 0 && C % groups == 0 && C / groups <= 64 && HW % 4 == 0; }

int launch_spade_norm_apply(const float* x0, int C0, const float* x1, int C1, int groups, float eps, const float* st0, int np0,
                            const float* st1, int np1, const float* gb, const float* coef2, float* y, float* coef_out, int B, int HW,
                            hipStream_t s) {
    const int C = C0 + (x1 ? C1 : 0);
    MCVD_REQUIRE(spade_norm_apply_supported(C, groups, HW), "spade_norm_apply: C=%d groups=%d HW=%d", C, groups, HW);
    MCVD_REQUIRE(st0 && np0 > 0 && HW % np0 == 0 && (!x1 || C1 == 0 || (st1 && np1 > 0 && HW % np1 == 0)),
                 "spade_norm_apply: bad partial statistics (np0=%d np1=%d HW=%d)", np0, np1, HW);
    // enough workgroups to fill the chip: (sample, group) pairs x pixel parts, a part no smaller than 256 float4 columns per channel
    const int gs = C / groups;
    int parts = 1;
    while ((long)B * groups * parts < 1024 && (HW / 4) / (parts * 2) * gs >= 256 && (HW / 4) % (parts * 2) == 0) parts *= 2;
    SpadeNormArgs a{x0, x1, C0, x1 ? C1 : 0, groups, eps, st0, np0, st1, (x1 && C1) ? np1 : 1, gb, coef2, y, coef_out, B, HW, parts};
    hipLaunchKernelGGL(spade_norm_apply_kernel, dim3((unsigned)(B * groups * parts)), dim3(256), 0, s, a);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

// coef2[b][c] = (1 + scale, shift) from the fused Dense_0 output (layerspp.py:523,535)
__global__ void coef2_kernel(const float* emb, int emb_stride, int emb_off, float* coef2, int B, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    const float* e = emb + (long)b * emb_stride + emb_off;
    coef2[2 * i] = 1.0f + e[c];
    coef2[2 * i + 1] = e[C + c];
}

int launch_coef2(const float* emb, int emb_stride, int emb_off, float* coef2, int B, int C, hipStream_t s) {
    hipLaunchKernelGGL(coef2_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, emb, emb_stride, emb_off, coef2, B, C);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

// every (1 + scale, shift) table of a forward in ONE launch (a SPADE net has one per normalisation: 56 launches of coef2_kernel per
// forward at BASELINE config 4, each a few hundred bytes of work behind a full launch): table t = blockIdx.y, desc[t] = {arena offset per
// sample of the table, column of its Dense_0 block in the fused embedding row, channels}
__global__ void coef2_all_kernel(const float* emb, int emb_stride, const long long* desc, float* arena, int B) {
    const long long* d = desc + 3 * blockIdx.y;
    const int C = (int)d[2], emb_off = (int)d[1];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    const float* e = emb + (long)b * emb_stride + emb_off;
    float* coef2 = arena + d[0] * (long long)B;
    coef2[2 * i] = 1.0f + e[c];
    coef2[2 * i + 1] = e[C + c];
}

int launch_coef2_all(const float* emb, int emb_stride, const long long* desc_dev, int ntab, int cmax, float* arena, int B, hipStream_t s) {
    hipLaunchKernelGGL(coef2_all_kernel, dim3((B * cmax + 255) / 256, ntab), dim3(256), 0, s, emb, emb_stride, desc_dev, arena, B);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

// F.interpolate(mode='nearest'): src = floor(dst * in/out)
__global__ void nearest_kernel(const float* in, float* out, int BC, int H, int W, int oh, int ow) {
    const long n = (long)BC * oh * ow;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
        const long pc = i / ((long)ow * oh);
        const int sy = (int)(((long)oy * H) / oh), sx = (int)(((long)ox * W) / ow);
        out[i] = in[pc * H * W + sy * W + sx];
    }
}

int launch_nearest_resize(const float* in, float* out, int BC, int H, int W, int oh, int ow, hipStream_t s) {
    const long n = (long)BC * oh * ow;
    const int blocks = (int)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
    hipLaunchKernelGGL(nearest_kernel, dim3(blocks), dim3(256), 0, s, in, out, BC, H, W, oh, ow);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mcvd
