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

// The arithmetic of the resamplers, spelled out so that every form below (element-wise, register window, LDS strip) rounds alike
// whatever the compiler would contract: the kernel a shape takes may differ between batch sizes (strip geometry), the bits must not.
__device__ __forceinline__ float fir_pro(const FirArgs& a, float v, float cA, float cB, float g, float bt, float sA, float sB);
__device__ __forceinline__ float fir_mix(float wa, float pa, float wb, float pb) {      // wa * pa + wb * pb
#pragma clang fp contract(off)
    const float t = wb * pb;
    return __builtin_fmaf(wa, pa, t);
}
__device__ __forceinline__ float fir_1331(float p0, float p1, float p2, float p3) {      // (p0 + 3 p1 + 3 p2 + p3) / 8
#pragma clang fp contract(off)
    float t = __builtin_fmaf(3.0f, p1, p0);
    t = __builtin_fmaf(3.0f, p2, t);
    t = t + p3;
    return t * 0.125f;
}

// (transformed, raw) value of the input plane at (yy, xx); zero outside (padding applies after the activation)
__device__ __forceinline__ V2 fir_src(const FirArgs& a, const float* plane, long pidx, int yy, int xx, float cA,
                                      float cB, float sA, float sB) {
    V2 o{0.0f, 0.0f};
    if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) return o;
    float v = plane[yy * a.W + xx];
    o.r = v;
    float g = 0.0f, bt = 0.0f;
    if (a.gamma) {
        const long bc = pidx / ((long)a.H * a.W);
        const long b = bc / a.C, c = bc - b * a.C;
        const long gi = ((b * 2 * a.C + c) * a.H + yy) * a.W + xx;
        g = a.gamma[gi]; bt = a.beta[gi];
    }
    o.h = fir_pro(a, v, cA, cB, g, bt, sA, sB);
    return o;
}

__device__ __forceinline__ float fir_pro(const FirArgs& a, float v, float cA, float cB, float g, float bt, float sA, float sB) {
#pragma clang fp contract(off)
    if (a.coef) v = __builtin_fmaf(v, cA, cB);
    if (a.gamma) {                                       // (1 + gamma) * normalised + beta, then the temb pair (layerspp.py:164-171)
        const float g1 = 1.0f + g;
        v = __builtin_fmaf(v, g1, bt);
        v = __builtin_fmaf(v, sA, sB);
    }
    if (a.act) v = silu1(v);
    return v;
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
                col[j] = fir_mix(wya, p.h, wyb, q.h);
                colr[j] = fir_mix(wya, p.r, wyb, q.r);
            }
            o[0] = fir_mix(0.25f, col[0], 0.75f, col[1]);
            o[1] = fir_mix(0.75f, col[1], 0.25f, col[2]);
            o[2] = fir_mix(0.25f, col[1], 0.75f, col[2]);
            o[3] = fir_mix(0.75f, col[2], 0.25f, col[3]);
            r[0] = fir_mix(0.25f, colr[0], 0.75f, colr[1]);
            r[1] = fir_mix(0.75f, colr[1], 0.25f, colr[2]);
            r[2] = fir_mix(0.25f, colr[1], 0.75f, colr[2]);
            r[3] = fir_mix(0.75f, colr[2], 0.25f, colr[3]);
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
                        const float v = fir_pro(a, vals[j], cA, cB, a.gamma ? gv[j] : 0.0f, a.gamma ? bv[j] : 0.0f, sA, sB);
                        hv[r][j] = in ? v : 0.0f;
                        rv[r][j] = in ? vals[j] : 0.0f;
                    }
                }
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    col[j] = fir_1331(hv[0][j], hv[1][j], hv[2][j], hv[3][j]);
                    colr[j] = fir_1331(rv[0][j], rv[1][j], rv[2][j], rv[3][j]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = fir_1331(col[2 * k], col[2 * k + 1], col[2 * k + 2], col[2 * k + 3]);
                r[k] = fir_1331(colr[2 * k], colr[2 * k + 1], colr[2 * k + 2], colr[2 * k + 3]);
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
                const float v = fir_pro(a, vals[j], cA, cB, a.gamma ? gv[j] : 0.0f, a.gamma ? bv[j] : 0.0f, sA, sB);
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
                col[j] = rr == 0 ? fir_mix(0.25f, hv[0][j], 0.75f, hv[1][j]) : fir_mix(0.75f, hv[1][j], 0.25f, hv[2][j]);
                colr[j] = rr == 0 ? fir_mix(0.25f, rv[0][j], 0.75f, rv[1][j]) : fir_mix(0.75f, rv[1][j], 0.25f, rv[2][j]);
            }
            float o[8], r8[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[2 * k] = fir_mix(0.25f, col[k], 0.75f, col[k + 1]);
                o[2 * k + 1] = fir_mix(0.75f, col[k + 1], 0.25f, col[k + 2]);
                r8[2 * k] = fir_mix(0.25f, colr[k], 0.75f, colr[k + 1]);
                r8[2 * k + 1] = fir_mix(0.75f, colr[k + 1], 0.25f, colr[k + 2]);
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

// ---- the two resamplers through the LDS (round 5).  The register forms above activate every input element once per thread whose window
// holds it -- 4.5 x (up: 3 x 6 window per 4 inputs) and 2.5 x (down: 4 x 10 window per 4 outputs) the SiLU's exp + rcp, two quarter-rate
// instructions, on kernels that should run at the HBM rate -- and read the window as 3-4 row requests per thread.  Here a workgroup takes a
// strip of 1024 input elements (whole planes where a plane is smaller; R = 1024 / W rows of one plane otherwise), every thread loads ONE
// aligned float4 (+ the strip's two halo rows by the first 2 W / 4 threads), applies the prologue ONCE and parks the result in the LDS
// behind a zero frame; the FIR then reads its windows from the LDS with the operation order of the register forms (bit-identical) and
// writes rows of consecutive float4 / float2.
// rows of a plane per strip, planes per strip, LDS row pitch, LDS rows per plane slot (R + 2), log2(W / 4), log2(R), strips per plane
struct FirStrip { int R, P, pitch, rows, lw4, lr, spp; };

__device__ __forceinline__ float4 fir_act4(const FirArgs& a, float4 v, long goff, float cA, float cB, float sA, float sB, const float* gpl,
                                           const float* bpl) {
    float e[4] = {v.x, v.y, v.z, v.w};
    float g[4] = {0.f, 0.f, 0.f, 0.f}, bt[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.gamma) {
        const float4 g4 = *reinterpret_cast<const float4*>(gpl + goff), b4 = *reinterpret_cast<const float4*>(bpl + goff);
        g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w; bt[0] = b4.x; bt[1] = b4.y; bt[2] = b4.z; bt[3] = b4.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = fir_pro(a, e[j], cA, cB, g[j], bt[j], sA, sB);
    return make_float4(e[0], e[1], e[2], e[3]);
}

// one aligned float4 of plane `plane` at element offset off: (prologue-transformed, raw)
__device__ __forceinline__ void fir_load4(const FirArgs& a, unsigned plane, long off, float4* h, float4* r) {
    const long HW = (long)a.H * a.W;
    *r = *reinterpret_cast<const float4*>(a.x + plane * HW + off);
    float cA = 1.f, cB = 0.f, sA = 1.f, sB = 0.f;
    if (a.coef) { const float2 c = *reinterpret_cast<const float2*>(a.coef + 2L * plane); cA = c.x; cB = c.y; }
    if (a.coef2) { const float2 c = *reinterpret_cast<const float2*>(a.coef2 + 2L * plane); sA = c.x; sB = c.y; }
    const float* gpl = nullptr; const float* bpl = nullptr;
    if (a.gamma) {
        const unsigned b = plane / (unsigned)a.C, c = plane - b * (unsigned)a.C;
        gpl = a.gamma + ((long)b * 2 * a.C + c) * HW; bpl = a.beta + ((long)b * 2 * a.C + c) * HW;
    }
    *h = fir_act4(a, *r, off, cA, cB, sA, sB, gpl, bpl);
}

// fills the strip's LDS image(s): sh = prologue-transformed, shr = raw (RAW only).  Row l of plane slot p sits at (p * rows + l) * pitch,
// l = 0 / R + 1 the halo rows; column x at + 4 + x, the zero frame at + 3 and + 4 + W.  plane0 / r0: first plane and first row of the strip.
template <bool RAW, int NL>
__device__ __forceinline__ void fir_strip_fill(const FirArgs& a, const FirStrip& st, unsigned plane0, int r0, float* sh, float* shr) {
    const int tid = threadIdx.x;
    const int W4 = 1 << st.lw4;
    float4 h[NL], r[NL];
    int l[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {   // interior: thread -> NL x (plane slot, row, float4 column); all the loads of a thread in flight together
        const int i = k * 256 + tid;
        const int c4 = i & (W4 - 1), rr = (i >> st.lw4) & (st.R - 1), p = i >> (st.lw4 + st.lr);
        fir_load4(a, plane0 + p, (long)(r0 + rr) * a.W + c4 * 4, &h[k], &r[k]);
        l[k] = (p * st.rows + rr + 1) * st.pitch + 4 + c4 * 4;
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        *reinterpret_cast<float4*>(sh + l[k]) = h[k];
        if (RAW) *reinterpret_cast<float4*>(shr + l[k]) = r[k];
    }
    if (tid < 2 * st.P * W4) {   // halo rows: 2 per plane slot; outside the image -> zeros
        const int c4 = tid & (W4 - 1), which = (tid >> st.lw4) & 1, p = tid >> (st.lw4 + 1);
        const int yy = which ? r0 + st.R : r0 - 1;
        float4 hh = make_float4(0.f, 0.f, 0.f, 0.f), rh = hh;
        if (yy >= 0 && yy < a.H) fir_load4(a, plane0 + p, (long)yy * a.W + c4 * 4, &hh, &rh);
        const int lh = (p * st.rows + (which ? st.R + 1 : 0)) * st.pitch + 4 + c4 * 4;
        *reinterpret_cast<float4*>(sh + lh) = hh;
        if (RAW) *reinterpret_cast<float4*>(shr + lh) = rh;
    }
}

// value of the lane below / above in the wave (the strip's rows are power-of-two runs of lanes: the caller masks the row ends)
__device__ __forceinline__ float lane_prev(float v) { return __shfl_up(v, 1); }
__device__ __forceinline__ float lane_next(float v) { return __shfl_down(v, 1); }

template <int NL>
__global__ __launch_bounds__(256) void fir_up2_lds_kernel(FirArgs a, FirStrip st, unsigned nstrips) {
    extern __shared__ __attribute__((aligned(16))) float fir_sh[];
    const int OW = 2 * a.W, OC4 = a.W >> 1;               // output float4 columns per row
    for (unsigned strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
        const unsigned plane0 = st.P > 1 ? strip * st.P : strip / (unsigned)st.spp;
        const int r0 = st.P > 1 ? 0 : (int)(strip - plane0 * st.spp) * st.R;
        fir_strip_fill<false, NL>(a, st, plane0, r0, fir_sh, nullptr);
        __syncthreads();
        // item -> (plane slot, input row ny, output float4 column): output rows 2 ny, 2 ny + 1, columns 4 oc4 .. 4 oc4 + 3 from input
        // rows ny - 1 .. ny + 1, columns 2 oc4 - 1 .. 2 oc4 + 2: one aligned float2 per row, the outer two columns from the neighbour lanes
#pragma unroll
        for (int k = 0; k < 2 * NL; ++k) {
            const int it = k * 256 + threadIdx.x;
            const int oc4 = it & (OC4 - 1), ny = (it >> (st.lw4 + 1)) & (st.R - 1), p = it >> (st.lw4 + 1 + st.lr);
            const float* base = fir_sh + (p * st.rows + ny) * st.pitch + 4 + 2 * oc4;       // row ny - 1, column 2 oc4
            float hv[3][4];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float2 q = *reinterpret_cast<const float2*>(base + r * st.pitch);
                const float lo = lane_prev(q.y), hi = lane_next(q.x);
                hv[r][0] = oc4 == 0 ? 0.0f : lo; hv[r][1] = q.x; hv[r][2] = q.y; hv[r][3] = oc4 == OC4 - 1 ? 0.0f : hi;
            }
            float* orow = a.y + ((long)(plane0 + p) * 2 * a.H + 2 * (r0 + ny)) * OW + 4 * oc4;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                float col[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    col[j] = rr == 0 ? fir_mix(0.25f, hv[0][j], 0.75f, hv[1][j]) : fir_mix(0.75f, hv[1][j], 0.25f, hv[2][j]);
                const float4 o = make_float4(fir_mix(0.25f, col[0], 0.75f, col[1]), fir_mix(0.75f, col[1], 0.25f, col[2]),
                                             fir_mix(0.25f, col[1], 0.75f, col[2]), fir_mix(0.75f, col[2], 0.25f, col[3]));
                *reinterpret_cast<float4*>(orow + rr * OW) = o;
            }
        }
        __syncthreads();
    }
}

template <int NL>
__global__ __launch_bounds__(256) void fir_down2_lds_kernel(FirArgs a, FirStrip st, unsigned nstrips) {
    extern __shared__ __attribute__((aligned(16))) float fir_sh[];
    const int OH = a.H >> 1, OW = a.W >> 1, OC2 = a.W >> 2;       // output float2 columns per row
    float* shr = fir_sh + st.P * st.rows * st.pitch;
    for (unsigned strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
        const unsigned plane0 = st.P > 1 ? strip * st.P : strip / (unsigned)st.spp;
        const int r0 = st.P > 1 ? 0 : (int)(strip - plane0 * st.spp) * st.R;
        if (a.y_raw) fir_strip_fill<true, NL>(a, st, plane0, r0, fir_sh, shr);
        else fir_strip_fill<false, NL>(a, st, plane0, r0, fir_sh, nullptr);
        __syncthreads();
        // waves 0, 1: the transformed tensor, waves 2, 3: the raw one; item -> (plane slot, output row, output float2 column):
        // outputs (oy, 2 oc2), (oy, 2 oc2 + 1) from input rows 2 oy - 1 .. 2 oy + 2, columns 4 oc2 - 1 .. 4 oc2 + 4: one aligned
        // float4 per row, the outer two columns from the neighbour lanes
        const int half = threadIdx.x >> 7;
        if (half == 0 || a.y_raw) {
#pragma unroll 1
            for (int k = 0; k < NL; ++k) {
                const int it = k * 128 + (threadIdx.x & 127);
                const int oc2 = it & (OC2 - 1), oyl = (it >> st.lw4) & ((st.R >> 1) - 1), p = it >> (st.lw4 + st.lr - 1);
                const float* base = (half ? shr : fir_sh) + (p * st.rows + 2 * oyl) * st.pitch + 4 + 4 * oc2;    // row 2 oy - 1, column 4 oc2
                float w[4][6];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 q = *reinterpret_cast<const float4*>(base + r * st.pitch);
                    const float lo = lane_prev(q.w), hi = lane_next(q.x);
                    w[r][0] = oc2 == 0 ? 0.0f : lo; w[r][1] = q.x; w[r][2] = q.y; w[r][3] = q.z; w[r][4] = q.w; w[r][5] = oc2 == OC2 - 1 ? 0.0f : hi;
                }
                float col[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) col[j] = fir_1331(w[0][j], w[1][j], w[2][j], w[3][j]);
                const float2 o = make_float2(fir_1331(col[0], col[1], col[2], col[3]), fir_1331(col[2], col[3], col[4], col[5]));
                *reinterpret_cast<float2*>((half ? a.y_raw : a.y) + ((long)(plane0 + p) * OH + (r0 >> 1) + oyl) * OW + 2 * oc2) = o;
            }
        }
        __syncthreads();
    }
}

// strip geometry of the LDS forms, or false where they do not apply (the register forms above take those)
static bool fir_strip_geometry(int up, int nl, long planes, int H, int W, bool raw, FirStrip* st, unsigned* nstrips) {
    if (W < 8 || W > (up ? 128 : 256) || (W & (W - 1)) || H < 2 || (up && raw)) return false;     // (a row of items = at most one wave)
    const long HW = (long)H * W;
    const int E = 1024 * nl;                          // input elements of a strip: nl aligned float4 per thread
    int R, P;
    if (HW <= E) {
        if (E % HW) return false;
        P = (int)(E / HW); R = H;
        if (planes % P) return false;
    } else {
        P = 1; R = E / W;
        if (H % R) return false;
    }
    if (R & (R - 1)) return false;                    // (whole planes of a non-power-of-two height: the register forms)
    if (!up && R < 2) return false;
    // the halo rows of a strip (2 per plane, W / 4 float4 each) are filled by ONE pass of the 256 threads (fir_strip_fill: tid < 2 P W/4):
    // geometries with more halo float4s than threads (e.g. H = 2 planes of a power-of-two width under the 2048-element strip) would leave
    // part of the halo unwritten -- the register forms take those
    if (2L * P * (W / 4) > 256) return false;
    const long n = P > 1 ? planes / P : planes * (H / R);
    if (n >= (1L << 31) || planes >= (1L << 31)) return false;
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    st->R = R; st->P = P; st->rows = R + 2; st->pitch = W + 8; st->lw4 = lg(W / 4); st->lr = lg(R); st->spp = P > 1 ? 1 : H / R;
    *nstrips = (unsigned)n;
    return true;
}

int launch_fir2(const float* x, const float* coef, int act, int up, float* y, int B, int C, int H, int W,
                const float* gamma, const float* beta, const float* coef2, float* y_raw, hipStream_t s, int form) {
    MCVD_REQUIRE((up ? W * 2 : W / 2) % 4 == 0 && H % 2 == 0, "fir2: H=%d W=%d unsupported", H, W);
    FirArgs a{x, coef, act, up, y, B, C, H, W, y_raw, gamma, beta, coef2};
    FirStrip st; unsigned nstrips = 0;
    // strips of 2048 elements for the down kernel (it reads 4 bytes for every 2 it writes: two loads per thread in flight -- 1024-element
    // strips measured 3.1 TB/s on the 64 x 64 layer, latency-bound), 1024 for the up kernel (write-bound: 5.4 TB/s)
    int nl = up ? 1 : 2;
    bool ok = form == 0 && fir_strip_geometry(up, nl, (long)B * C, H, W, y_raw != nullptr, &st, &nstrips);
    if (!ok && form == 0 && nl == 2) { nl = 1; ok = fir_strip_geometry(up, nl, (long)B * C, H, W, y_raw != nullptr, &st, &nstrips); }
    if (ok) {
        const size_t lds = (size_t)st.P * st.rows * st.pitch * sizeof(float) * (y_raw ? 2 : 1);
        const int blocks = (int)(nstrips > 65536u ? 65536u : nstrips);
        if (up) hipLaunchKernelGGL(fir_up2_lds_kernel<1>, dim3(blocks), dim3(256), lds, s, a, st, nstrips);
        else if (nl == 2) hipLaunchKernelGGL(fir_down2_lds_kernel<2>, dim3(blocks), dim3(256), lds, s, a, st, nstrips);
        else hipLaunchKernelGGL(fir_down2_lds_kernel<1>, dim3(blocks), dim3(256), lds, s, a, st, nstrips);
        MCVD_HIP_CHECK(hipGetLastError());
        return 0;
    }
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
        o.x = silu1(fma_unpacked(fma_unpacked(v.x, cA, cB) * (1.0f + g.x) + be.x, sA, sB));
        o.y = silu1(fma_unpacked(fma_unpacked(v.y, cA, cB) * (1.0f + g.y) + be.y, sA, sB));
        o.z = silu1(fma_unpacked(fma_unpacked(v.z, cA, cB) * (1.0f + g.z) + be.z, sA, sB));
        o.w = silu1(fma_unpacked(fma_unpacked(v.w, cA, cB) * (1.0f + g.w) + be.w, sA, sB));
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
