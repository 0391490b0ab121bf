// 3x3 convolution with a HANDFUL of output channels (the network's last layer: ngf -> C * num_frames = 5 or 15 channels,
// ncsnpp_more.py:247 / :586; reference op: layers.py:107-113 nn.Conv2d(3, stride 1, pad 1)).  Shape id 21.
// Every matrix-pipe kernel of this library works on 32-cout tiles: for Cout = 5 the Winograd kernel multiplies 27 of 32 output columns
// of zeros and its launch takes 109 us for 111 MB of algorithmic traffic (17 us at the achievable HBM rate): VERDICT r4 item 6.
// With so few couts the layer is an fp32 VALU job: 9 * Cin * Cout FMAs per pixel (4320 for 96 -> 5) against one activated input read.
//   * workgroup = 256 threads = a 16 x 64 pixel tile of one sample, EVERY cout; thread = four consecutive pixels of a row, 4 x Cout
//     accumulators in registers (one pixel per thread, the first form, read 1.7 FMAs' worth of LDS per FMA: LDS-bound);
//   * input channels in chunks of 4: the 18 x 66 x 4 patch is fetched (aligned float4 + two halo pixels per row), passed through the
//     GroupNorm affine (+ SiLU) ONCE per element and parked in LDS, double-buffered: chunk k + 1 is fetched and activated while chunk k
//     is multiplied; a thread reads six patch values per (channel, row) for 12 * Cout FMAs;
//   * the weights of a chunk (8 x 9 x COUT floats) are parked in LDS beside the patch and read back with BROADCAST ds_read_b128 (every
//     lane the same address: one LDS cycle, four weights) -- the first form fed the FMAs from scalar loads and waited for the scalar
//     cache five times per input channel (lgkmcnt is shared with the patch reads): 146 us where the Winograd kernel takes 119;
//   * exact fp32: one FMA chain per (pixel, cout) in (ci, tap) order -- the summation order of the one-thread-per-output test kernel.
#include "../common.h"

namespace mcvd {

__device__ __forceinline__ float silu_sc(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

constexpr int SC_CK = 4;                       // input channels per staged chunk
constexpr int SC_TW = 64, SC_TH = 16;          // pixel tile: 16 rows x 64 columns, four consecutive pixels of a row per thread
constexpr int SC_PW = 68;                      // LDS row pitch in floats (66 used; 272 bytes: every row 16-byte aligned)
constexpr int SC_PSZ = SC_CK * (SC_TH + 2) * SC_PW;     // floats of one staged chunk (19.1 KB)

template <int COUT, int PRO>                   // PRO: 0 raw, 1 affine, 2 affine + SiLU
__global__ __launch_bounds__(256, 3) void conv_small_cout_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) float sP[2][SC_PSZ];
    constexpr int CW = (COUT + 3) & ~3;                            // couts padded to the 16-byte LDS reads
    __shared__ __attribute__((aligned(16))) float sW[2][SC_CK * 9 * CW];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;                         // pixels (ty, 4 tx .. 4 tx + 3) of the tile
    const int tiles_x = a.W / SC_TW, tiles_y = a.H / SC_TH;
    const int b = blockIdx.x / (tiles_x * tiles_y);
    const int tr = blockIdx.x - b * (tiles_x * tiles_y);
    const int oy0 = (tr / tiles_x) * SC_TH, ox0 = (tr % tiles_x) * SC_TW;
    const int HW = a.H * a.W, Cin = a.Cin;
    const int nchunks = (Cin + SC_CK - 1) / SC_CK;

    float acc[4][COUT];
#pragma unroll
    for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[px][co] = 0.0f;

    // staging role: the 4 x 18 x 66 patch as 16 four-pixel items + 2 halo pixels per (channel, row): item e = k * 256 + tid of
    // 4 * 18 * 16 = 1152 aligned float4 (5 rounds, the last ragged) and h = tid of 4 * 18 * 2 = 144 halo pixels
    constexpr int NI = SC_CK * (SC_TH + 2) * 16, NR = (NI + 255) / 256;
    float4 pre[NR];
    float preh;
    auto prologue = [&](float v, int cg) -> float {
        if (PRO >= 1) {
            const float2 cf = *reinterpret_cast<const float2*>(a.coef + ((long)b * Cin + cg) * 2);
            v = v * cf.x + cf.y;
        }
        if (PRO >= 2) v = silu_sc(v);
        return v;
    };
    auto src_of = [&](int cg) -> const float* {
        return cg < a.C0 ? a.x0 + ((long)b * a.C0 + cg) * HW : a.x1 + ((long)b * a.C1 + (cg - a.C0)) * HW;
    };
    auto fetch = [&](int ch) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int e = k * 256 + tid;
            const int ci = e / ((SC_TH + 2) * 16), rem = e - ci * ((SC_TH + 2) * 16), r = rem >> 4, c4 = rem & 15;
            const int cg = ch * SC_CK + ci, y = oy0 - 1 + r;
            const bool in = e < NI && cg < Cin && y >= 0 && y < a.H;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in) {
                v = *reinterpret_cast<const float4*>(src_of(cg) + y * a.W + ox0 + c4 * 4);
                v.x = prologue(v.x, cg); v.y = prologue(v.y, cg); v.z = prologue(v.z, cg); v.w = prologue(v.w, cg);
            }
            pre[k] = v;                         // zero padding applies AFTER the activation
        }
        {
            const int e = tid;
            const int ci = e / ((SC_TH + 2) * 2), rem = e - ci * ((SC_TH + 2) * 2), r = rem >> 1, side = rem & 1;
            const int cg = ch * SC_CK + ci, y = oy0 - 1 + r, x = side ? ox0 + SC_TW : ox0 - 1;
            const bool in = e < SC_CK * (SC_TH + 2) * 2 && cg < Cin && y >= 0 && y < a.H && x >= 0 && x < a.W;
            preh = in ? prologue(src_of(cg)[y * a.W + x], cg) : 0.0f;
        }
    };
    // weights of a chunk: element e = (ci * 9 + tap) * CW + co  <-  wp[((ch * 4 + ci) * 9 + tap) * CoutP + co] (zero beyond Cin / COUT)
    constexpr int NWE = SC_CK * 9 * CW, NWR = (NWE + 255) / 256;
    float prew[NWR];
    auto fetch_w = [&](int ch) {
#pragma unroll
        for (int k = 0; k < NWR; ++k) {
            const int e = k * 256 + tid;
            const int row = e / CW, co = e - row * CW;             // row = ci * 9 + tap
            const int cg = ch * SC_CK + row / 9;
            prew[k] = (e < NWE && co < COUT && cg < Cin) ? a.wp[(long)(ch * SC_CK * 9 + row) * a.CoutP + co] : 0.0f;
        }
    };
    auto park = [&](int buf) {                  // LDS column of image column x: x - ox0 + 4 (the left halo at 3: interior float4s stay 16-byte aligned)
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int e = k * 256 + tid;
            if (e < NI) {
                const int ci = e / ((SC_TH + 2) * 16), rem = e - ci * ((SC_TH + 2) * 16), r = rem >> 4, c4 = rem & 15;
                *reinterpret_cast<float4*>(&sP[buf][(ci * (SC_TH + 2) + r) * SC_PW + 4 + c4 * 4]) = pre[k];
            }
        }
        if (tid < SC_CK * (SC_TH + 2) * 2) {
            const int ci = tid / ((SC_TH + 2) * 2), rem = tid - ci * ((SC_TH + 2) * 2), r = rem >> 1, side = rem & 1;
            sP[buf][(ci * (SC_TH + 2) + r) * SC_PW + (side ? 4 + SC_TW : 3)] = preh;
        }
#pragma unroll
        for (int k = 0; k < NWR; ++k) {
            const int e = k * 256 + tid;
            if (e < NWE) sW[buf][e] = prew[k];
        }
    };
    fetch(0);
    fetch_w(0);
    park(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) { fetch(ch + 1); fetch_w(ch + 1); }  // in flight under the FMAs below
        const float* sp = sP[ch & 1] + ty * SC_PW + 4 * tx;
        const float* sw = sW[ch & 1];
#pragma unroll 1
        for (int cr = 0; cr < SC_CK * 3; ++cr) {                   // (channel, patch row) pairs; channels past Cin: zero patch rows, zero weights
            const int ci = cr / 3, r = cr - 3 * ci;
            const float* pr = sp + (ci * (SC_TH + 2) + r) * SC_PW;
            // the six patch values the four pixels' taps of this row need: columns 4 tx + 3 .. 4 tx + 8 of the LDS row
            const float pl = pr[3];
            const float4 pm = *reinterpret_cast<const float4*>(pr + 4);
            const float pe = pr[8];
            const float p[6] = {pl, pm.x, pm.y, pm.z, pm.w, pe};
            const float* wr = sw + (ci * 9 + r * 3) * CW;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float w[CW];
#pragma unroll
                for (int q = 0; q < CW / 4; ++q)                    // broadcast reads: the address does not depend on the lane
                    *reinterpret_cast<float4*>(w + 4 * q) = *reinterpret_cast<const float4*>(wr + t * CW + 4 * q);
#pragma unroll
                for (int px = 0; px < 4; ++px)
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[px][co] = fmaf(w[co], p[px + t], acc[px][co]);
            }
        }
        if (ch + 1 < nchunks) park((ch + 1) & 1);                  // (the other buffer: last read one iteration ago, before the barrier below)
        __syncthreads();
    }
    const int oy = oy0 + ty, ox = ox0 + 4 * tx;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        if (co < a.Cout) {
            const long o = ((long)b * a.Cout + co) * HW + oy * a.W + ox;
            const float bs = a.bias[co];
            float4 v = make_float4(acc[0][co] + bs, acc[1][co] + bs, acc[2][co] + bs, acc[3][co] + bs);
            if (a.res) {
                const float4 rv = *reinterpret_cast<const float4*>(a.res + o);
                v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            }
            v.x *= a.out_scale; v.y *= a.out_scale; v.z *= a.out_scale; v.w *= a.out_scale;
            *reinterpret_cast<float4*>(a.y + o) = v;
        }
    }
}

bool conv_small_cout_usable(const ConvArgs& a) {
    return a.ks == 3 && a.Cout >= 1 && a.Cout <= 16 && a.CoutP >= a.Cout && a.H % SC_TH == 0 && a.W % SC_TW == 0 && !a.gb && !a.stats && !a.gni.st0 && (!a.act || a.coef);
}

template <int COUT>
static int sc_launch(const ConvArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)(a.B * (a.H / SC_TH) * (a.W / SC_TW)));
    if (!a.coef) hipLaunchKernelGGL((conv_small_cout_kernel<COUT, 0>), grid, dim3(256), 0, s, a);
    else if (!a.act) hipLaunchKernelGGL((conv_small_cout_kernel<COUT, 1>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_small_cout_kernel<COUT, 2>), grid, dim3(256), 0, s, a);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_conv_small_cout(const ConvArgs& a, hipStream_t s) {
    MCVD_REQUIRE(conv_small_cout_usable(a), "small-cout direct conv: unsupported (ks=%d Cout=%d H=%d W=%d)", a.ks, a.Cout, a.H, a.W);
    if (a.Cout <= 5) return sc_launch<5>(a, s);
    if (a.Cout <= 8) return sc_launch<8>(a, s);
    return sc_launch<16>(a, s);
}

}  // namespace mcvd
