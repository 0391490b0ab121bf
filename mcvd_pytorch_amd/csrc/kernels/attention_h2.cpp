// Multi-head self-attention core (layerspp.py:237-244), the flash-style kernel of attention.cpp on the 16-bit matrix pipes with every
// MFMA operand split into pieces (pieces.h has the arithmetic):
//   NP = 3: three bf16 pieces, six piece products -- fp32-equivalent, full fp32 range: the default attention kernel.
//   NP = 2: two fp16 pieces, three piece products (22-bit operands; context option "f16x2").
// attention.cpp is bound by the fp32 matrix pipe (MFMA-busy 0.68, waves issue-stalled 76 % of the time:
// profiles/r02_pmc_*.txt); 3 (6) MFMAs of the 16x faster pipe per 16-deep step take 3/16 (6/16) of its pipe time.  Same
// decomposition: one workgroup = 4 waves = 4 x 32 queries of one (sample, head), key tiles of 32, S^T (keys on the accumulator
// rows) so that the softmax is in-lane and P is already the B operand of the PV product:
//   S^T[key][query] = sum_c K[c][key] * Q[c][query]      A = K tile pieces (LDS, lane = key), B = Q pieces (registers, lane = query)
//   O[c][query]    += sum_key V[c][key] * P[key][query]  A = V tile pieces (LDS, lane = c),   B = P pieces (registers)
// What changes against attention.cpp:
//   * Q is split once per workgroup (the pieces take the registers the fp32 fragment took).
//   * the threads that stage a K / V tile split it (each element once per workgroup, not once per wave) and park the pieces in the
//     MFMA A-operand order [step][piece][64 lanes][4 dwords]: one ds_read_b128 per operand.  K-slot conventions:
//       S product : lane half h, element e of step st  <->  channel 16 st + 2e + h              (dword j = channels 4j + h, 4j + 2 + h)
//       PV product: lane half h, element e of step s2  <->  key (r&3) + 8 (r>>2) + 4h, r = 8 s2 + e  -- the key whose probability
//                   sits in accumulator register r of the S^T tile; a dword = two consecutive keys.
//   * NP = 2 only: Q, K, V enter the pieces times 2^4 and the probabilities times 2^12 (folded into the exponent:
//     exp(s - m + 12 ln 2); the row sums carry the same factor and it cancels), so second pieces stay clear of the fp16 denormal
//     range.  The score scale D^-0.5 absorbs the 2^-8 of the S product, the final 1 / l the 2^-4 of V.  All scales are powers of
//     two: exact.  Nothing is clamped: a q / k / v beyond 65504 / 16 becomes Inf and the output NaN (reported: model.cpp range guard).
//     NP = 3 scales nothing.
//   * LDS layouts chosen for the STAGING stores (round 2 scattered single dwords 16 bytes apart: 8-way bank conflicts on the K stores,
//     4-way on the V stores, 62 % of the LDS-active cycles, profiles/r02_pmc_sq_wave_states_f16x2.txt):
//       K pieces as dword planes [step][piece][dword j][64 lanes]: a staging thread owns 4 consecutive keys of a channel pair = 4
//         consecutive dwords of one plane -> ONE ds_write_b128 per piece, 8 lanes = 128 contiguous bytes; the S product reads its
//         operand with four conflict-free ds_read_b32.
//       V pieces stay in operand order [sub-tile][step][piece][64 lanes][4 dwords] (one ds_read_b128 per operand); the staging lanes
//         are permuted (lane = g | row << 1 | h << 4 | s2 << 5) so that 16 consecutive lanes write 128 contiguous bytes.
// Head dims 32..128 (Q pieces + O accumulators + the register prefetch of the next tile fit two waves per SIMD); anything else stays
// on attention.cpp.
#include <math.h>

#include "../common.h"
#include "pieces.h"

namespace mcvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr float AH_P_LOG = 8.317766166719343f;        // 12 ln 2: NP = 2 probabilities times 2^12
// NP = 3 softmax (round 5): Q enters the S product times D^-0.5 log2(e), so the scores arrive in the base-2 exponent domain -- no score
// scaling, no multiply inside the exponential (32 of the ~240 VALU instructions a wave issued per key tile) -- and the running
// reference of the exponent is LAZY: O and l are rescaled only when a tile's maximum exceeds the reference by more than 2^AH_LAZY
// (wave-uniform branch; in practice the first tile only), not by exp(m_old - m_new) on every tile (48 + multiplies per tile at D = 96).
// Probabilities then range up to 2^AH_LAZY instead of 1 -- the three bf16 pieces carry the fp32 exponent, the row sum carries the same
// factor and it cancels in O / l; the global maximum's own term is >= 1 (the reference never exceeds a tile maximum), so l cannot
// underflow.  Same accuracy, different rounding than exp(s - m_running).  Measured: attention -1.3 ... -3.2 % (the kernel is not bound by
// its VALU count, nor -- each ablated or A/B-ed on the same box -- by the DMA prefetch depth, the V operand reads' latency, the LDS
// operand traffic or the wave priorities: profiles/r05_attention_experiments.txt).
constexpr float AH_LOG2E = 1.4426950408889634f;
constexpr float AH_LAZY = 40.0f;

// the NP = 3 online softmax of one key tile: st = scores (base-2 domain) -> probabilities relative to the lazy reference m_run;
// o / l_run rescaled only when some lane's tile maximum outgrows its reference (alpha == 1 exactly for the lanes that do not need it)
#ifdef MCVD_AH_EAGER      /* A/B build only (tools/build_variant.sh): round 4's softmax, exp(s - m_running) with the rescale on every tile */
#define AH_SOFTMAX_LAZY                                                                                            \
    {                                                                                                              \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { st[r] *= scale_s; mt = fmaxf(mt, st[r]); }               \
        mt = fmaxf(mt, __shfl_xor(mt, 32));                                                                        \
        const float m_new = fmaxf(m_run, mt);                                                                      \
        const float alpha = __expf(m_run - m_new);                                                                 \
        const float shift = 0.0f - m_new;                                                                          \
        float ps = 0.0f;                                                                                           \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { st[r] = __expf(st[r] + shift); ps += st[r]; }            \
        l_run = l_run * alpha + ps;                                                                                \
        m_run = m_new;                                                                                             \
        _Pragma("unroll") for (int ct = 0; ct < DT; ++ct)                                                          \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;                                      \
    }
#define AH_QS(scale_s) 1.0f
#else
#define AH_SOFTMAX_LAZY                                                                                            \
    {                                                                                                              \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[r]);                                      \
        mt = fmaxf(mt, __shfl_xor(mt, 32));                                                                        \
        const bool need = mt > m_run + AH_LAZY;                                                                    \
        if (__builtin_amdgcn_ballot_w64(need)) {                                                                   \
            const float m_new = need ? mt : m_run;                                                                 \
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);                                             \
            l_run *= alpha;                                                                                        \
            _Pragma("unroll") for (int ct = 0; ct < DT; ++ct)                                                      \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;                                  \
            m_run = m_new;                                                                                         \
        }                                                                                                          \
        float ps = 0.0f;                                                                                           \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r] - m_run); ps += st[r]; } \
        l_run += ps;                                                                                               \
    }
#define AH_QS(scale_s) ((scale_s) * AH_LOG2E)
#endif

// (tools/repro_coresident.cpp compiles this kernel stand-alone with other launch bounds / function attributes: the co-residency report of
// DESIGN.md section 7.  The product build leaves both macros at their defaults.)
#ifndef MCVD_AH_LB
#define MCVD_AH_LB __launch_bounds__(256, 2)
#endif
#ifndef MCVD_AH_ATTR
#define MCVD_AH_ATTR
#endif
template <int NP, int DT>   // head dim D = 32*DT
__global__ MCVD_AH_LB MCVD_AH_ATTR void attn_h2_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, int heads, int S,
                                                          float scale_s, int nbh, int nqt) {
    typedef Pieces<NP> PX;
    constexpr int D = 32 * DT, NST = D / 16;
    constexpr int NIK = (4 * D + 255) / 256;     // K staging items per thread: (row pair, 4 keys) -> 2 float4
    constexpr int NIV = DT;                      // V staging items per thread: (channel, 4 keys) -> 1 float4   (8 D / 256)
    constexpr float OPS = NP == 2 ? PX::ACT_SCALE : 1.0f;          // operand scale of Q, K, V
    extern __shared__ __attribute__((aligned(16))) float smem_attn_h2[];
    unsigned* sK = reinterpret_cast<unsigned*>(smem_attn_h2);      // [NST][NP pieces][4 dwords j][64 lanes]
    unsigned* sV = sK + NST * NP * 256;                             // [DT][2 steps][NP pieces][64 lanes][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    // block id -> (sample-head, query tile): workgroups are dealt to the 8 XCDs round robin, and every query tile of a (sample, head)
    // streams the same K and V.  With (query tile, sample-head) as a 2-D grid the 8 query tiles of a 32 x 32 head landed on 8 different
    // XCDs -- 8 private L2s each fetched every K / V once: L2 hit rate 0.078, 4x the algorithmic read traffic
    // (profiles/r03_pmc_l2_tcc_tcp.txt).  Here the query tiles of a (sample, head) run on ONE XCD, adjacent in time.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bh = (slot / nqt) * 8 + xcd, qt = slot - (slot / nqt) * nqt;
    if (bh >= nbh) return;
    const int b = bh / heads, hd = bh - b * heads;
    const float* qb = qkv + ((long)b * 3 * C + hd * D) * S;
    const float* kb = qb + (long)C * S;
    const float* vb = kb + (long)C * S;
    const int q0 = qt * 128 + wave * 32;
    const bool active = q0 < S;

    // Q pieces: lane (query l31, half) holds channels 16 st + 2e + half, e = 0..7 of every step; dword j = elements (2j, 2j+1)
    const float qs = NP == 3 ? AH_QS(scale_s) : OPS;              // NP = 3: the score scale and log2(e) ride on Q (see AH_LAZY)
    u32x4 qp[NST][NP];
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c0 = 16 * st + 4 * j + half;
            const float a0 = active ? qb[(long)c0 * S + q0 + l31] : 0.0f;
            const float a1 = active ? qb[(long)(c0 + 2) * S + q0 + l31] : 0.0f;
            unsigned w[NP];
            PX::template split<false>(a0 * qs, a1 * qs, w);
#pragma unroll
            for (int p = 0; p < NP; ++p) qp[st][p][j] = w[p];
        }

    f32x16 o[DT];
#pragma unroll
    for (int ct = 0; ct < DT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.0f;
    float m_run = -1e30f, l_run = 0.0f;

    // ---- staging roles.  K item idx = i*256 + tid < 4D: row pair rp = idx >> 3 = (st, j, h) -> rows c, c + 2 with c = 16 st + 4j + h,
    //      keys 4 kq .. 4 kq + 3 (kq = idx & 7).  V item (i, wave, lane): channel c = 32 i + 8 wave + ((lane >> 1) & 7), keys 4q .. 4q + 3 with
    //      q = (s2, g, h) = (lane >> 5, lane & 1, (lane >> 4) & 1): the lane order that makes the b64 stores below contiguous.
    const int v_c = wave * 8 + ((lane >> 1) & 7), v_h = (lane >> 4) & 1, v_s2 = lane >> 5, v_g = lane & 1;
    const int v_q = v_s2 * 4 + v_g * 2 + v_h;
    f32x4 rk0[NIK], rk1[NIK], rv[NIV];
    const int ntiles = S / 32;
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < NIK; ++i) {
            const int idx = min(i * 256 + tid, 4 * D - 1);
            const int rp = idx >> 3, kq = idx & 7;
            const int c = 16 * (rp >> 3) + 4 * ((rp >> 1) & 3) + (rp & 1);
            const long off = (long)c * S + t * 32 + kq * 4;
            rk0[i] = *reinterpret_cast<const f32x4*>(kb + off);
            rk1[i] = *reinterpret_cast<const f32x4*>(kb + off + 2 * (long)S);
        }
#pragma unroll
        for (int i = 0; i < NIV; ++i) {
            rv[i] = *reinterpret_cast<const f32x4*>(vb + (long)(i * 32 + v_c) * S + t * 32 + v_q * 4);
        }
    };
    gload(0);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();                          // previous tile fully consumed
        // ---- split the staged tile and park the pieces in operand order
#pragma unroll
        for (int i = 0; i < NIK; ++i) {
            const int idx = i * 256 + tid;
            if (NIK * 256 == 4 * D || idx < 4 * D) {
                const int rp = idx >> 3, kq = idx & 7;
                const int st = rp >> 3, j = (rp >> 1) & 3, h = rp & 1;
                // plane (st, piece, j): dwords h*32 + key; the thread's four keys are four consecutive dwords
                unsigned* d1 = sK + (st * NP * 4 + j) * 64 + h * 32 + kq * 4;
                unsigned w[4][NP];
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) PX::template split<false>(rk0[i][i4] * OPS, rk1[i][i4] * OPS, w[i4]);
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(d1 + p * 256) = u32x4{w[0][p], w[1][p], w[2][p], w[3][p]};
            }
        }
#pragma unroll
        for (int i = 0; i < NIV; ++i) {
            const int c = i * 32 + v_c;
            const int ct = c >> 5, m = c & 31, j0 = 2 * v_g;
            unsigned wa[NP], wb[NP];
            PX::template split<false>(rv[i][0] * OPS, rv[i][1] * OPS, wa);
            PX::template split<false>(rv[i][2] * OPS, rv[i][3] * OPS, wb);
            unsigned* d1 = sV + ((ct * 2 + v_s2) * NP * 64 + v_h * 32 + m) * 4 + j0;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(d1 + p * 256) = u32x2{wa[p], wb[p]};
        }
        __syncthreads();
        if (t + 1 < ntiles) gload(t + 1);

        // ---- S^T tile: 32 keys x 32 queries (NP = 2: times 2^8)
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.0f;
        const unsigned* sKl = sK + lane;
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            u32x4 kp[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) kp[p][j] = sKl[((s * NP + p) * 4 + j) * 64];
#pragma unroll
            for (int k = 0; k < PX::NPROD; ++k) st = PX::mfma(kp[PX::PA(k)], qp[s][PX::PB(k)], st);
        }

        // ---- online softmax over keys (this lane: 16 keys of query l31; partner lane^32 holds the other 16); NP = 2: p times 2^12
        float mt = -1e30f;
        if constexpr (NP == 3) {
            AH_SOFTMAX_LAZY
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] *= scale_s; mt = fmaxf(mt, st[r]); }
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = fmaxf(m_run, mt);
            const float alpha = __expf(m_run - m_new);
            const float shift = AH_P_LOG - m_new;
            float ps = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = __expf(st[r] + shift); ps += st[r]; }
            l_run = l_run * alpha + ps;               // per-half partial sum; halves are added at the end
            m_run = m_new;
#pragma unroll
            for (int ct = 0; ct < DT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
        }
        u32x4 pp[2][NP];                          // B operand of the PV product: step s2, dword j = registers 8 s2 + 2j, + 1
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned w[NP];
                PX::template split<false>(st[8 * s2 + 2 * j], st[8 * s2 + 2 * j + 1], w);      // 0 <= p <= 1 (NP = 2: 2^12)
#pragma unroll
                for (int p = 0; p < NP; ++p) pp[s2][p][j] = w[p];
            }

        // ---- O += V * P
        const u32x4* sVl = reinterpret_cast<const u32x4*>(sV) + lane;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            // two cout sub-tiles at a time: their MFMAs alternate between two accumulators, and only 2 * NP operand quads are live
#pragma unroll
            for (int c0 = 0; c0 < DT; c0 += 2) {
                constexpr int G = 2;
                u32x4 vp[G][NP];
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (c0 + g < DT) {
#pragma unroll
                        for (int p = 0; p < NP; ++p) vp[g][p] = sVl[(((c0 + g) * 2 + s2) * NP + p) * 64];
                    }
#pragma unroll
                for (int k = 0; k < PX::NPROD; ++k)
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        if (c0 + g < DT) o[c0 + g] = PX::mfma(vp[g][PX::PA(k)], pp[s2][PX::PB(k)], o[c0 + g]);
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = (1.0f / OPS) / l_tot;      // NP = 2: l carries the 2^12 of the probabilities, O the 2^12 * 2^4 of P and V
    if (active) {
        float* ob = out + ((long)b * C + hd * D) * S + q0 + l31;
#pragma unroll
        for (int ct = 0; ct < DT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                ob[(long)c * S] = o[ct][r] * inv;
            }
    }
}

// ---- the NP = 3 kernel fed with PRE-SPLIT K and V (round 5).  attn_h2_kernel<3, *> fetches a K / V tile through
// registers, splits every element three ways and parks the pieces -- once per QUERY TILE, eight times per element at 32 x 32, 126 of the
// 367 VALU instructions a wave issues per key tile -- and prefetches the next tile in 28 registers.  Here the q|k|v projection has
// already written the pieces (conv1x1_h2.cpp, KV): per (sample, head, key tile) ONE contiguous image
//     [K: step][piece][dword j][64 lanes]  [V: sub-tile][step s2][piece][64 lanes][4 dwords]            3072 DT dwords = 12 KB x DT
// in exactly the order the two products read their operands from the LDS, so a tile reaches the LDS by LDS-DMA (16 bytes per lane, no
// VALU instruction, no register) into one of TWO buffers: the DMA of tile t + 1 is issued behind the barrier that publishes tile t and
// lands under that tile's products; one barrier per tile instead of two.  Same pieces, same products in the same order: the output is
// bit-identical to attn_h2_kernel<3, DT>.  Head dims 32 ... 128.
// NW = waves per workgroup = 32-query tiles that share a staged K / V tile: 4 for head dims up to 96 (two workgroups per CU, 24 KB x DT of
// LDS each), 8 for head dim 128 (ONE workgroup of 256 queries per CU: its two 48 KB buffers are 96 KB; two waves per SIMD as before, and
// none of attn_h2_kernel<3, 4>'s 17 spilled registers -- the staging registers are gone).
template <int DT, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_h2p_kernel(const float* __restrict__ qkv, const unsigned* __restrict__ kv_img, float* __restrict__ out,
                                                              int C, int heads, int S, float scale_s, int nbh, int nqt) {
    typedef Pieces<3> PX;
    constexpr int NP = 3, D = 32 * DT, NST = D / 16, NT = 64 * NW;
    constexpr int IMG = 3072 * DT;                  // dwords of one (K | V) tile image
    constexpr int NPC = IMG / 4;                    // its 16-byte pieces: 768 DT
    constexpr int NR = (NPC + NT - 1) / NT;         // DMA rounds per thread and tile
    static_assert(NPC % 64 == 0, "a wave's 64 pieces are all inside the image or all outside");
    extern __shared__ __attribute__((aligned(16))) float smem_attn_h2p[];
    unsigned* sT = reinterpret_cast<unsigned*>(smem_attn_h2p);      // [2][IMG]: K image at 0, V image at 1536 DT

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bh = (slot / nqt) * 8 + xcd, qt = slot - (slot / nqt) * nqt;
    if (bh >= nbh) return;
    const int b = bh / heads, hd = bh - b * heads;
    const float* qb = qkv + ((long)b * 3 * C + hd * D) * S;
    const int q0 = qt * (32 * NW) + wave * 32;
    const bool active = q0 < S;
    const int ntiles = S / 32;
    const unsigned* ib = kv_img + (long)bh * ntiles * IMG;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    auto dma = [&](int t) {
        const unsigned* src = ib + (long)t * IMG;
        unsigned* dst = sT + (t & 1) * IMG;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int q0p = r * NT + wave_u * 64;                   // the wave's first piece of this round (wave-uniform)
            if ((r + 1) * NT <= NPC || q0p < NPC)                   // (only a ragged last round is predicated)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (q0p + lane) * 4),
                                                 (__attribute__((address_space(3))) void*)(dst + q0p * 4), 16, 0, 0);
        }
    };
    dma(0);

    const float qs = AH_QS(scale_s);                // as attn_h2_kernel<3, DT>: bit-identical to it
    u32x4 qp[NST][NP];
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c0 = 16 * st + 4 * j + half;
            const float a0 = active ? qb[(long)c0 * S + q0 + l31] : 0.0f;
            const float a1 = active ? qb[(long)(c0 + 2) * S + q0 + l31] : 0.0f;
            unsigned w[NP];
            PX::template split<false>(a0 * qs, a1 * qs, w);
#pragma unroll
            for (int p = 0; p < NP; ++p) qp[st][p][j] = w[p];
        }

    f32x16 o[DT];
#pragma unroll
    for (int ct = 0; ct < DT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.0f;
    float m_run = -1e30f, l_run = 0.0f;

    for (int t = 0; t < ntiles; ++t) {
        // tile t has landed (the only VMEM operations in flight are its DMA pieces -- the Q loads were consumed above), every wave's part
        // is visible behind the barrier, and every wave has finished tile t - 1: its buffer is free for tile t + 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 1 < ntiles) dma(t + 1);
        const unsigned* sK = sT + (t & 1) * IMG;
        const unsigned* sV = sK + NST * NP * 256;

        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.0f;
        const unsigned* sKl = sK + lane;
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            u32x4 kp[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) kp[p][j] = sKl[((s * NP + p) * 4 + j) * 64];
#pragma unroll
            for (int k = 0; k < PX::NPROD; ++k) st = PX::mfma(kp[PX::PA(k)], qp[s][PX::PB(k)], st);
        }

        float mt = -1e30f;
        AH_SOFTMAX_LAZY
        u32x4 pp[2][NP];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned w[NP];
                PX::template split<false>(st[8 * s2 + 2 * j], st[8 * s2 + 2 * j + 1], w);
#pragma unroll
                for (int p = 0; p < NP; ++p) pp[s2][p][j] = w[p];
            }

        const u32x4* sVl = reinterpret_cast<const u32x4*>(sV) + lane;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int c0 = 0; c0 < DT; c0 += 2) {
                constexpr int G = 2;
                u32x4 vp[G][NP];
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (c0 + g < DT) {
#pragma unroll
                        for (int p = 0; p < NP; ++p) vp[g][p] = sVl[(((c0 + g) * 2 + s2) * NP + p) * 64];
                    }
#pragma unroll
                for (int k = 0; k < PX::NPROD; ++k)
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        if (c0 + g < DT) o[c0 + g] = PX::mfma(vp[g][PX::PA(k)], pp[s2][PX::PB(k)], o[c0 + g]);
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (active) {
        float* ob = out + ((long)b * C + hd * D) * S + q0 + l31;
#pragma unroll
        for (int ct = 0; ct < DT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                ob[(long)c * S] = o[ct][r] * inv;
            }
    }
}

bool attn_h2p_supported(int C, int heads, int HW) {
    if (heads <= 0 || C % heads != 0) return false;
    const int D = C / heads;
    if (D % 32 != 0 || D < 32 || D > 128 || HW % 32 != 0) return false;
    return D <= 96 || HW >= 256;                    // head dim 128: 256-query workgroups (8 x 8 images stay on attn_h2_kernel)
}

int launch_attention_h2p(const float* qkv, const float* kv_img, float* out, int B, int C, int heads, int HW, hipStream_t s) {
    MCVD_REQUIRE(attn_h2p_supported(C, heads, HW) && kv_img, "pre-split attention: unsupported (C=%d heads=%d HW=%d)", C, heads, HW);
    const int D = C / heads;
    const float scale_s = (float)pow((double)D, -0.5);
    const int NWq = D <= 96 ? 4 : 8;
    const int nqt = (HW + 32 * NWq - 1) / (32 * NWq), nbh = B * heads;
    dim3 grid((unsigned)(((nbh + 7) / 8) * 8 * nqt));
    const size_t lds = (size_t)2 * 3072 * (D / 32) * sizeof(unsigned);
    const unsigned* img = reinterpret_cast<const unsigned*>(kv_img);
#define AHP_CASE(DT, NW)                                                                                                  \
    case DT: {                                                                                                            \
        static PerDeviceOnce raised;                                                                                      \
        if (lds > 48 * 1024 && raised.first_use()) {                                                                      \
            MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_h2p_kernel<DT, NW>),                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                  \
            raised.done();                                                                                                \
        }                                                                                                                 \
        hipLaunchKernelGGL((attn_h2p_kernel<DT, NW>), grid, dim3(64 * NW), lds, s, qkv, img, out, C, heads, HW, scale_s, nbh, nqt); \
        break;                                                                                                            \
    }
    switch (D / 32) {
        AHP_CASE(1, 4) AHP_CASE(2, 4) AHP_CASE(3, 4) AHP_CASE(4, 8)
    }
#undef AHP_CASE
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

bool attention_h2_supported(int C, int heads, int HW) {
    if (heads <= 0 || C % heads != 0) return false;
    const int D = C / heads;
    return D % 32 == 0 && D >= 32 && D <= 128 && HW % 32 == 0;
}

template <int NP>
static int attn_h2_launch(const float* qkv, float* out, int B, int C, int heads, int HW, hipStream_t s) {
    const int D = C / heads;
    const float scale = (float)pow((double)D, -0.5);   // int(C)**-0.5 as a Python double, then fp32 (layerspp.py:239)
    const float ops = NP == 2 ? Pieces<2>::ACT_SCALE : 1.0f;
    const float scale_s = scale * (1.0f / (ops * ops));      // exact: the S product carries the square of the operand scale
    const int nqt = (HW + 127) / 128, nbh = B * heads;
    dim3 grid((unsigned)(((nbh + 7) / 8) * 8 * nqt));
    const size_t lds = (size_t)(D / 16 + 2 * (D / 32)) * NP * 256 * sizeof(unsigned);
    switch (D / 32) {
        case 1: hipLaunchKernelGGL((attn_h2_kernel<NP, 1>), grid, dim3(256), lds, s, qkv, out, C, heads, HW, scale_s, nbh, nqt); break;
        case 2: hipLaunchKernelGGL((attn_h2_kernel<NP, 2>), grid, dim3(256), lds, s, qkv, out, C, heads, HW, scale_s, nbh, nqt); break;
        case 3: hipLaunchKernelGGL((attn_h2_kernel<NP, 3>), grid, dim3(256), lds, s, qkv, out, C, heads, HW, scale_s, nbh, nqt); break;
        default: hipLaunchKernelGGL((attn_h2_kernel<NP, 4>), grid, dim3(256), lds, s, qkv, out, C, heads, HW, scale_s, nbh, nqt); break;
    }
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_attention_h2(const float* qkv, float* out, int B, int C, int heads, int HW, hipStream_t s, int np) {
    MCVD_REQUIRE(attention_h2_supported(C, heads, HW) && (np == 2 || np == 3), "split-operand attention: unsupported (C=%d heads=%d HW=%d np=%d)", C,
                 heads, HW, np);
    return np == 2 ? attn_h2_launch<2>(qkv, out, B, C, heads, HW, s) : attn_h2_launch<3>(qkv, out, B, C, heads, HW, s);
}

// mode = the "naive_attn" option: 0 auto (the split-operand kernel where it applies: two fp16 pieces when f16x2 is on, else three bf16
// pieces when bf16x3 is on; else the fp32 flash kernel), 1 the one-thread-per-query kernel, 2 the fp32 flash kernel, 3 two fp16 pieces
// where the kernel applies, 4 three bf16 pieces where it applies
int launch_attention(int mode, int f16x2, int bf16x3, const float* qkv, float* out, int B, int C, int heads, int HW, hipStream_t s) {
    if (mode == 1) return launch_attention_naive(qkv, out, B, C, heads, HW, s);
    const bool sup = attention_h2_supported(C, heads, HW);
    if (sup && (mode == 3 || (mode == 0 && f16x2))) return launch_attention_h2(qkv, out, B, C, heads, HW, s, 2);
    if (sup && (mode == 4 || (mode == 0 && bf16x3))) return launch_attention_h2(qkv, out, B, C, heads, HW, s, 3);
    return launch_attention_mfma(qkv, out, B, C, heads, HW, s);
}

}  // namespace mcvd
