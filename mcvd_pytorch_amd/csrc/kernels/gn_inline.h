// GroupNorm coefficients computed by the CONSUMER (context option "gn_inline", off by default): the conv that applies (A, B) reduces
// the producers' partial statistics itself, in its prologue, instead of reading a table a separate gn_finalize launch wrote.
// gn_finalize is pure launch latency by its durations (65 launches per forward of config 2 at ~5 us each for a few hundred bytes per
// wave: 2.4 % of the GPU time, profiles/r03_rocprofv3_summary.txt).  MEASURED, the fusion does not pay: every workgroup of the consumer
// repeats the reduction (~1.5 us on its critical path: one memory round trip it cannot start early, three barriers), while the
// launches it spares largely overlap with their neighbours' ramp-up and tail inside the graph -- 0.3-1 % slower end to end on every
// config, whether applied wherever a channel has at most GN_INLINE_MAX_NP partials or only on launches of at most two workgroup
// rounds (profiles/r03_gn_inline_ab.txt).  Kept as a tested option.
//
// Same identities as gn_finalize_kernel (gn.cpp), merged in two levels -- the partials of a channel first, then the channels of a
// group: N = sum n_i, mean = sum sum_i / N, M2 = sum (M2_i + n_i (mean_i - mean)^2); every term non-negative, fixed order.
#pragma once
#include <hip/hip_runtime.h>

namespace mcvd {

constexpr int GN_INLINE_MAX_NP = 8;

struct GnInline {
    const float* st0;       // partial (sum, M2) pairs of source 0: [B][C0][np0][2]; NULL = the conv reads ConvArgs::coef as before
    const float* st1;       // source 1 of a virtual concat: [B][C1][np1][2]
    int np0, np1;           // partials per (sample, channel): 1 .. GN_INLINE_MAX_NP, each over HW / np pixels
    int groups;
    float eps;
    int mode;               // GnArgs::mode: 0 plain, 1 temb scale / shift, 2 affine weight / bias
    const float* p0;        // mode 1: emb [B][emb_stride]; mode 2: weight [C]
    const float* p1;        // mode 2: bias [C]
    int emb_stride, emb_off;
};

// (A, B) of channels 0 .. C-1 of samples min(b_first + s, B - 1), s < nsamp, into tab[(s * C + c) * 2 + {0, 1}] (LDS; also the
// scratch of the reduction).  Every one of the NT threads of the workgroup calls it; it holds two barriers and the caller places one
// more before the table is read.  nsamp * C <= KMAX * NT.
template <int NT, int KMAX>
__device__ __forceinline__ void gn_inline_coef(const GnInline& g, int b_first, int nsamp, int B, int C0, int C1, int HW, float* tab, int tid) {
    const int C = C0 + C1, gs = C / g.groups, n_items = nsamp * C;
    const float fHW = (float)HW;
    float2* tab2 = reinterpret_cast<float2*>(tab);
    float par0[KMAX], par1[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int it = tid + k * NT;
        par0[k] = 1.0f;
        par1[k] = 0.0f;
        if (it < n_items) {
            const int s = it / C, c = it - s * C, b = min(b_first + s, B - 1);
            const bool second = c >= C0;
            const int cl = second ? c - C0 : c, Cs = second ? C1 : C0, np = second ? g.np1 : g.np0;
            const float2* q = reinterpret_cast<const float2*>(second ? g.st1 : g.st0) + ((long)b * Cs + cl) * np;
            float2 v[GN_INLINE_MAX_NP];
#pragma unroll
            for (int p = 0; p < GN_INLINE_MAX_NP; ++p) v[p] = p < np ? q[p] : make_float2(0.0f, 0.0f);
            if (g.mode == 1) {                 // (1 + scale) * norm + shift        layerspp.py:523,535
                const float* e = g.p0 + (long)b * g.emb_stride + g.emb_off;
                par0[k] = 1.0f + e[c];
                par1[k] = e[C + c];
            } else if (g.mode == 2) {          // weight * norm + bias              torch GroupNorm affine
                par0[k] = g.p0[c];
                par1[k] = g.p1[c];
            }
            const float n = fHW / (float)np;
            float sum = 0.0f;
#pragma unroll
            for (int p = 0; p < GN_INLINE_MAX_NP; ++p) sum += v[p].x;          // (absent partials are zero)
            const float mean_c = sum / fHW;
            float m2 = 0.0f;
#pragma unroll
            for (int p = 0; p < GN_INLINE_MAX_NP; ++p)
                if (p < np) {
                    const float d = v[p].x / n - mean_c;
                    m2 += v[p].y + n * d * d;
                }
            tab2[it] = make_float2(sum, m2);
        }
    }
    __syncthreads();
    float ca[KMAX], cb[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int it = tid + k * NT;
        ca[k] = cb[k] = 0.0f;
        if (it < n_items) {
            const int s = it / C, c = it - s * C;
            const float2* grp = tab2 + s * C + (c / gs) * gs;
            float tot = 0.0f;
            for (int j = 0; j < gs; ++j) tot += grp[j].x;
            const float N = (float)gs * fHW;
            const float mean = tot / N;
            float acc = 0.0f;
            for (int j = 0; j < gs; ++j) {
                const float2 t = grp[j];
                const float d = t.x / fHW - mean;
                acc += t.y + fHW * d * d;
            }
            const float rstd = 1.0f / sqrtf(acc / N + g.eps);
            ca[k] = rstd * par0[k];
            cb[k] = par1[k] - mean * rstd * par0[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int it = tid + k * NT;
        if (it < n_items) tab2[it] = make_float2(ca[k], cb[k]);
    }
}

}  // namespace mcvd
