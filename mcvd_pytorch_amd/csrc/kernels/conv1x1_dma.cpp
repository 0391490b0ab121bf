// 1x1 convolution / NIN as a plain GEMM on v_mfma_f32_32x32x2_f32 whose operands reach the LDS ONLY by LDS-DMA:
//   y[b][co][p] = out_scale * ( sum_ci W[ci][co] * pro(x[b][ci][p]) + bias[co] (+ res[b][co][p]) )
// A 1x1 conv has 9x less arithmetic per staged element than the 3x3 one, so the register-staged implicit-GEMM kernel
// (conv_mfma.h) spends its time staging.  Here no VALU instruction touches an operand before the matrix pipe:
//   * weights  wp[ci][CoutP]  : rows are contiguous couts        -> global_load_lds, 16 B per lane
//   * pixels   x[b][ci][HW]   : rows are contiguous pixels (NCHW) -> global_load_lds, 16 B per lane
//   * the GroupNorm affine (A_c, B_c) of the prologue is applied to the B operand on its way from the LDS to the MFMA
//     (one FMA per MFMA group, coefficients broadcast from a small LDS table), optional SiLU likewise.
// Workgroup = 4 waves = 128 consecutive pixels of the flattened [B*HW] axis x 32*COT couts; wave w owns pixels 32w..32w+31
// and all COT cout sub-tiles.  K loop over 16-channel chunks, double-buffered, ONE barrier per chunk.
// Block id -> (pixel tile, cout tile) keeps the cout tiles of one pixel tile on one XCD (ids congruent mod 8), so the
// pixel rows are fetched from HBM once and re-read from that XCD's L2.
#include "../common.h"

namespace mcvd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float silu_g(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

constexpr int G1_PT = 128;       // pixels per workgroup
constexpr int G1_MAXIMG = 4;     // images a pixel tile may span (HW >= 32)

// PRO: 0 raw input, 1 affine, 2 affine + SiLU;  CK: input channels per chunk (16 or 32)
// PXW: 32-pixel blocks per wave (1 or 2).  PXW = 2: a workgroup covers 256 pixels, every weight operand read from the LDS feeds
// two MFMAs and a chunk holds 2*8*COT MFMAs per wave between barriers instead of 8*COT (tile shape id 9).
template <int COT, int PRO, int CK, int PXW = 1>
// second launch bound: 4 waves/SIMD (128 registers) up to cout tile 3, where accumulators + staging fit without spills
__global__ __launch_bounds__(256, COT * PXW <= 3 ? 4 : (COT * PXW <= 4 ? 3 : 1)) void conv1x1_dma_kernel(ConvArgs a, int ptiles, int nct) {
    constexpr int PT = G1_PT * PXW, BCO = 32 * COT;
    constexpr int PPR = PT / 4;               // 16-byte pieces per channel row of the pixel tile
    constexpr int RPS = 256 / PPR;            // channel rows one 256-thread DMA step covers (8 at 128 pixels, 4 at 256)
    constexpr int WSZ = CK * BCO, XSZ = CK * PT;
    constexpr int WPIECES = WSZ / 4, XPIECES = XSZ / 4;          // 16-byte pieces per chunk
    constexpr int MAXW = (WPIECES + 255) / 256, MAXX = XPIECES / 256;
    static_assert(XPIECES % 256 == 0, "x pieces");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW = smem;                         // [2][CK][BCO]
    float* sX = smem + 2 * WSZ;               // [2][CK][PT]
    float* sC = smem + 2 * WSZ + 2 * XSZ;     // [2][MAXIMG][CK][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ptile = (slot / nct) * 8 + xcd, ctile = slot - (slot / nct) * nct;
    if (ptile >= ptiles) return;
    const int co0 = ctile * BCO;
    const int HW = a.H * a.W, Cin = a.Cin;
    const long NPX = (long)a.B * HW;
    const long gp0 = (long)ptile * PT;

    // ---- x DMA role: piece e = s*256 + tid -> (channel-in-chunk e / PPR, 4 pixels (e % PPR)*4); pixel part is slot invariant
    long xg = gp0 + (tid % PPR) * 4;
    if (xg > NPX - 4) xg = NPX - 4;           // ragged last tile: fetch valid data, the stores are predicated
    const int xb = (int)(xg / HW), xp = (int)(xg - (long)xb * HW);
    const int x_ci = tid / PPR;               // + RPS*s
    const int voff0 = (xb * a.C0 + x_ci) * HW + xp;               // offset inside x0 (without the chunk base)
    const int voff1 = (xb * a.C1 + x_ci) * HW + xp;               // offset inside x1

    int w_goff[MAXW];
#pragma unroll
    for (int s = 0; s < MAXW; ++s) {
        int e = s * 256 + tid;
        if (e >= WPIECES) e = WPIECES - 1;    // only when WPIECES is not a multiple of 256: duplicate DMA of the last piece
        const int row = e / (BCO / 4), c4 = e - row * (BCO / 4);
        w_goff[s] = row * a.CoutP + co0 + c4 * 4;
    }

    // ---- coefficient role (PRO): thread t < nimg*CK loads (A,B) of (image t / CK, channel t % CK)
    const int b_first = (int)(gp0 / HW);
    const int nimg = HW >= PT ? 1 : PT / HW;
    const int c_img = min(b_first + tid / CK, a.B - 1), c_ci = tid % CK;
    const bool c_role = PRO != 0 && tid < nimg * CK;
    // image (within the tile) of this wave's pixel block j: pixels wave*32*PXW + 32*j .. +31
    int my_img[PXW];
#pragma unroll
    for (int j = 0; j < PXW; ++j) my_img[j] = HW >= PT ? 0 : (wave * 32 * PXW + 32 * j) / HW;
    f32x2 cf_next = {1.0f, 0.0f};

#define G1_DMA(ch)                                                                                              \
    {                                                                                                           \
        const int cb = (ch) * CK;                                                                               \
        const float* wsrc = a.wp + (long)cb * a.CoutP;                                                          \
        float* wdst = sW + (((ch) & 1) ? WSZ : 0);                                                              \
        _Pragma("unroll") for (int s = 0; s < MAXW; ++s)                                                        \
            if (MAXW * 256 == WPIECES || s * 256 + wave * 64 < WPIECES)                                         \
                __builtin_amdgcn_global_load_lds(                                                               \
                    (const __attribute__((address_space(1))) void*)(wsrc + w_goff[s]),                          \
                    (__attribute__((address_space(3))) void*)(wdst + (s * 256 + wave * 64) * 4), 16, 0, 0);   \
        const bool second = cb >= a.C0;                                                                         \
        const float* xsrc = second ? a.x1 + (long)(cb - a.C0) * HW : a.x0 + (long)cb * HW;                      \
        const int voff = second ? voff1 : voff0;                                                                \
        float* xdst = sX + (((ch) & 1) ? XSZ : 0);                                                              \
        _Pragma("unroll") for (int s = 0; s < MAXX; ++s)                                                        \
            __builtin_amdgcn_global_load_lds(                                                                   \
                (const __attribute__((address_space(1))) void*)(xsrc + (long)s * RPS * HW + voff),              \
                (__attribute__((address_space(3))) void*)(xdst + (s * 256 + wave * 64) * 4), 16, 0, 0);         \
        if (c_role) cf_next = *reinterpret_cast<const f32x2*>(a.coef + ((long)c_img * Cin + cb + c_ci) * 2);    \
    }
#define G1_WRITE_C(ch)                                                                                          \
    if (c_role) *reinterpret_cast<f32x2*>(sC + (((ch) & 1) ? G1_MAXIMG * CK * 2 : 0) + tid * 2) = cf_next;

    f32x16 acc[PXW][COT];
#pragma unroll
    for (int j = 0; j < PXW; ++j)
#pragma unroll
        for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][ct][r] = 0.0f;

    const int nchunks = Cin / CK;          // Cin % CK == 0 (launch check): the zero rows that pad wp to CinP are never staged
    // (An LDS-DMA is a load without a destination register: the compiler makes nobody wait for it in front of a barrier.  The wave that
    // issued it waits for it explicitly before the barrier that publishes the chunk -- conv_mfma.h has the story; with a prologue the wait
    // for the coefficient load, which is younger than the DMAs, used to cover them by accident, with a raw input nothing did.)
    G1_DMA(0);
    G1_WRITE_C(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        if (more) G1_DMA(ch + 1);
        const float* sWc = sW + ((ch & 1) ? WSZ : 0);
        const float* sXc = sX + ((ch & 1) ? XSZ : 0) + wave * 32 * PXW + l31;
        const float* sCb = sC + ((ch & 1) ? G1_MAXIMG * CK * 2 : 0);
#pragma unroll
        for (int kp = 0; kp < CK / 2; ++kp) {
            const int row = 2 * kp + half;
            float bv[PXW];
#pragma unroll
            for (int j = 0; j < PXW; ++j) {
                bv[j] = sXc[row * PT + 32 * j];
                if (PRO != 0) {
                    const f32x2 cf = *reinterpret_cast<const f32x2*>(sCb + my_img[j] * CK * 2 + row * 2);
                    bv[j] = bv[j] * cf.x + cf.y;
                    if (PRO == 2) bv[j] = silu_g(bv[j]);
                }
            }
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) {
                const float av = sWc[row * BCO + ct * 32 + l31];
#pragma unroll
                for (int j = 0; j < PXW; ++j) acc[j][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[j][ct], 0, 0, 0);
            }
        }
        if (more) G1_WRITE_C(ch + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's part of chunk ch + 1 has landed ...
        __syncthreads();                       // ... chunk ch consumed by every wave, chunk ch + 1 published
    }
#undef G1_DMA
#undef G1_WRITE_C

    // ---- epilogue: lane (l31, half) holds pixels gp0 + 32*PXW*wave + 32*j + l31, couts ct*32 + (r&3) + 8*(r>>2) + 4*half
#pragma unroll
    for (int j = 0; j < PXW; ++j) {
        const long gp = gp0 + wave * 32 * PXW + 32 * j + l31;
        if (gp >= NPX) continue;                  // NPX % 32 == 0: a 32-pixel block is valid or invalid as a whole
        const int ob = (int)(gp / HW), op = (int)(gp - (long)ob * HW);
        const long obase = (long)ob * a.Cout * HW + op;
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) {
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int coc = min(co, a.Cout - 1);
                rv[r] = a.res ? a.res[obase + (long)coc * HW] : 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float v = (acc[j][ct][r] + a.bias[co] + rv[r]) * a.out_scale;     // bias is zero-padded to CoutP
                if (co < a.Cout) a.y[obase + (long)co * HW] = v;
                if (a.stats) {
                    // GroupNorm partials for the next norm (ConvArgs::stats): the 32 lanes of a half-wave hold 32 consecutive pixels
                    // of ONE image (HW % 32 == 0) for this cout -> (sum, M2 about their mean), partial index = pixel block op / 32
                    float sm = v;
#pragma unroll
                    for (int o2 = 16; o2 > 0; o2 >>= 1) sm += __shfl_xor(sm, o2);
                    const float d = v - sm * (1.0f / 32.0f);
                    float m2 = d * d;
#pragma unroll
                    for (int o2 = 16; o2 > 0; o2 >>= 1) m2 += __shfl_xor(m2, o2);
                    if (l31 == 0 && co < a.Cout) {
                        float* q = a.stats + (((long)ob * a.Cout + co) * (HW >> 5) + (op >> 5)) * 2;
                        q[0] = sm;
                        q[1] = m2;
                    }
                }
            }
        }
    }
}

static int g1_cot_ok(int n32, int cot) { return cot >= 1 && n32 % cot == 0 && (cot <= 4 || cot == 6 || cot == 9); }

// Largest supported cout tile (in 32-channel units) for a padded channel count.
int conv1x1_dma_cout_tile(int CoutP) {
    const int n32 = CoutP / 32;
    static const int pref[] = {6, 9, 4, 3, 2, 1};
    for (int c : pref)
        if (n32 % c == 0) return c;
    return 1;
}

bool conv1x1_dma_supported(const ConvArgs& a, int ck, int pxw) {
    const int HW = a.H * a.W;
    const int G1_CK = ck, PT = G1_PT * pxw;
    if (a.ks != 1 || HW % 32 != 0 || (pxw != 1 && pxw != 2) || (pxw == 2 && ck != 16)) return false;
    if (!(HW % PT == 0 || (HW < PT && PT % HW == 0 && PT / HW <= G1_MAXIMG))) return false;
    if (a.Cin % G1_CK != 0 || a.CinP % G1_CK != 0) return false;            // no partial chunk: every staged row is real data
    if (a.C1 > 0 && a.C0 % G1_CK != 0) return false;                        // a chunk never straddles the concat seam
    if ((long)a.B * (a.C0 > a.C1 ? a.C0 : a.C1) * HW >= (1L << 31)) return false;   // 32-bit lane offsets
    if (a.act && !a.coef) return false;
    return true;
}

template <int COT, int G1_CK, int PXW = 1>
static int g1_launch(const ConvArgs& a, hipStream_t s) {
    const int HW = a.H * a.W;
    const long NPX = (long)a.B * HW;
    const int ptiles = (int)((NPX + G1_PT * PXW - 1) / (G1_PT * PXW));
    const int nct = a.CoutP / (32 * COT);
    const size_t lds = (size_t)(2 * G1_CK * 32 * COT + 2 * G1_CK * G1_PT * PXW + 2 * G1_MAXIMG * G1_CK * 2) * sizeof(float);
    const dim3 grid(((ptiles + 7) / 8) * 8 * nct);
    static PerDeviceOnce raised;
    if (lds > 48 * 1024 && raised.first_use()) {
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_dma_kernel<COT, 0, G1_CK, PXW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_dma_kernel<COT, 1, G1_CK, PXW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MCVD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_dma_kernel<COT, 2, G1_CK, PXW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised.done();
    }
    if (!a.coef)
        hipLaunchKernelGGL((conv1x1_dma_kernel<COT, 0, G1_CK, PXW>), grid, dim3(256), lds, s, a, ptiles, nct);
    else if (!a.act)
        hipLaunchKernelGGL((conv1x1_dma_kernel<COT, 1, G1_CK, PXW>), grid, dim3(256), lds, s, a, ptiles, nct);
    else
        hipLaunchKernelGGL((conv1x1_dma_kernel<COT, 2, G1_CK, PXW>), grid, dim3(256), lds, s, a, ptiles, nct);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int G1_CK>
static int g1_dispatch(const ConvArgs& a, int cot, hipStream_t s) {
    switch (cot) {
        case 1: return g1_launch<1, G1_CK>(a, s);
        case 2: return g1_launch<2, G1_CK>(a, s);
        case 3: return g1_launch<3, G1_CK>(a, s);
        case 4: return g1_launch<4, G1_CK>(a, s);
        case 6: return g1_launch<6, G1_CK>(a, s);
        default: return g1_launch<9, G1_CK>(a, s);
    }
}

// cot_req: requested cout tile (32-channel units); <= 0 or unsupported -> the default rule.  ck: channels per chunk, 16 or 32.
int launch_conv1x1_dma(const ConvArgs& a, int cot_req, int ck, hipStream_t s, int pxw) {
    if (pxw == 2) {                            // 64 pixels per wave: cout tiles 1 and 2 (64 / 128 accumulator registers... 32 / 64)
        MCVD_REQUIRE(ck == 16 && conv1x1_dma_supported(a, 16, 2), "conv1x1 dma (256-pixel tile): unsupported shape (H=%d W=%d Cin=%d C0=%d)", a.H, a.W, a.Cin, a.C0);
        const int n32 = a.CoutP / 32;
        const int cot = (cot_req == 2 && n32 % 2 == 0) ? 2 : 1;
        const int rc = cot == 2 ? g1_launch<2, 16, 2>(a, s) : g1_launch<1, 16, 2>(a, s);
        if (rc == 0 && a.stats) set_last_conv_stats_np(a.H * a.W / 32);
        return rc;
    }
    MCVD_REQUIRE((ck == 16 || ck == 32) && conv1x1_dma_supported(a, ck), "conv1x1 dma: unsupported shape (ks=%d H=%d W=%d Cin=%d C0=%d ck=%d)",
                 a.ks, a.H, a.W, a.Cin, a.C0, ck);
    const int n32 = a.CoutP / 32;
    const int cot = g1_cot_ok(n32, cot_req) ? cot_req : conv1x1_dma_cout_tile(a.CoutP);
    MCVD_REQUIRE(!(ck == 32 && cot == 9), "conv1x1 dma: cout tile 9 with 32-channel chunks exceeds the LDS budget");
    const int rc = ck == 32 ? g1_dispatch<32>(a, cot, s) : g1_dispatch<16>(a, cot, s);
    if (rc == 0 && a.stats) set_last_conv_stats_np(a.H * a.W / 32);
    return rc;
}

}  // namespace mcvd
