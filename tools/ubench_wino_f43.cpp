// VERDICT r5 item 3, measured: the K loop of a Winograd F(4,3) x F(2,3) HYBRID 3x3 conv (4 x 2 output pixels per tile from a 6 x 4 patch: 24
// transform positions per 8 pixels = 3 per pixel, against the 16 per 4 = 4 per pixel of the F(2x2,3x3) kernels that ship) next to the
// K loop of the shipped form, in ONE harness, timing only (no results), instruction mix of conv_wino3_kernel's chunk:
//   workgroup = 512 threads = 8 waves = two per SIMD, 32 tiles (the N of v_mfma_f32_32x32x16_bf16) x 32 COT couts, 16-channel chunks,
//   wave w owns positions PPW w .. PPW w + PPW - 1 (PPW = 2 shipped, 3 hybrid); per position and chunk: 12 B-operand dwords from the LDS
//   (three bf16 pieces), 3 COT weight quads (global_load_dwordx4 from an L2-resident layer image) and 6 COT MFMAs (six piece products);
//   staging per thread and chunk: its share of the patch (load, affine, SiLU, park) and half a (tile, channel pair) transform unit --
//   shipped: 12 patch reads of 8 bytes, 16 packed adds, 8 three-way bf16 splits, 24 dword stores;
//   hybrid : 18 patch reads, 12 + 2 x 13 = 38 packed adds / FMAs (the F(4,3) row transform has 4, -5, 2 coefficients), 12 splits, 36 stores.
// What the hybrid runs into BEFORE any cycle is counted (sizes, not opinions):
//   registers: 24 positions x 96 couts x 32 tiles of fp32 accumulators = 144 per lane at 8 waves -- with 9 weight quads per position in
//              flight and the staging temporaries that does not fit the 256 registers two waves per SIMD have; the hybrid exists at COT = 2
//              (64 couts per workgroup) only, so a layer pays one activation staging per 64 couts instead of one per 96;
//   LDS      : one V chunk is 24 x 3 KB = 72 KB; two of them (the shipped loop's double buffer: one barrier per chunk, the two waves of a
//              SIMD in opposite phase) + two 27 KB patches (18 x 18) = 198 KB > 160 KB.  The hybrid has ONE V buffer: a position may be
//              rewritten only after its owner has read it -> a barrier per position group (PPW per chunk).
// Variants (cycles per chunk by s_memtime, mean over 256 workgroups, one per CU; nch = 48 chunks):
//   S2/S3  shipped form, COT = 2 / 3: two V buffers, one barrier per chunk, waves 0-3 multiply then stage, waves 4-7 stage then multiply
//   H2     hybrid, COT = 2, ONE V buffer, three barriers per chunk (what fits)
//   H2x    hybrid, COT = 2, the shipped loop's structure on an LDS that does not exist (second V buffer ALIASED onto the first: same
//          traffic, no hazards to respect in a timing-only run): the upper bound of what the hybrid could be
// Reported: cycles per chunk and cycles per (output pixel x 32 couts) = the figure of merit (a chunk covers 128 pixels shipped, 256 hybrid).
// GO / NO-GO line, stated before the run: hybrid cycles per output pixel <= 0.92 x the shipped loop's -- H2 against S3 for the
// layers the shipped kernel serves at COT = 3 (Cout = 96, 192, 288, 384: all of config 2), H2 against S2 otherwise.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench_wino_f43.cpp -o /tmp/ubench_wino_f43 && /tmp/ubench_wino_f43
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int T = 32, PP = 24;

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3(f32x2 v, unsigned& w1, unsigned& w2, unsigned& w3) {
    w1 = cvt_pk(v.x, v.y);
    const f32x2 h = {__builtin_bit_cast(float, w1 << 16), __builtin_bit_cast(float, w1 & 0xffff0000u)};
    v = v - h;
    w2 = cvt_pk(v.x, v.y);
    const f32x2 g = {__builtin_bit_cast(float, w2 << 16), __builtin_bit_cast(float, w2 & 0xffff0000u)};
    v = v - g;
    w3 = cvt_pk(v.x, v.y);
}
__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// F(4,3) input transform of one row of six packed values (13 packed operations with the common subexpressions shared)
__device__ __forceinline__ void bt43(const f32x2 d[6], f32x2 t[6]) {
    const f32x2 a = d[4] - 4.0f * d[2], b = d[3] - 4.0f * d[1];
    t[1] = a + b; t[2] = a - b;
    const f32x2 c = d[4] - d[2], e = 2.0f * (d[3] - d[1]);
    t[3] = c + e; t[4] = c - e;
    t[0] = 4.0f * d[0] + (d[4] - 5.0f * d[2]);
    t[5] = 4.0f * d[1] + (d[5] - 5.0f * d[3]);
}

// NPOS 16 (shipped) / 24 (hybrid); VPHYS physical V buffers (2 = double buffered; 1 = single); NSYNC barriers per chunk (1 = the shipped loop's
// structure, PPW = one per position group); MODE bit 0 matrix side, bit 1 staging side
template <int NPOS, int COT, int VPHYS, int NSYNC, int MODE>
__global__ __launch_bounds__(512) void kw(const unsigned* __restrict__ wts, const float* __restrict__ x, unsigned long long* out, int nch,
                                          long wstride_chunk, int hw) {
    constexpr int PPW = NPOS / 8, NQ = 3 * COT, POSW = 2 * 4 * T, PWp = NPOS * POSW, VW = 3 * PWp;
    constexpr bool HY = NPOS == 24;
    constexpr int PR = HY ? 18 : 10, PSZ = 8 * PR * PP * 2;
    constexpr int NRD = HY ? 18 : 12, NSP = HY ? 12 : 8, NPV = HY ? 11 : 9;      // per thread and chunk: patch reads (8 B), splits, patch values
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sV = reinterpret_cast<unsigned*>(smem);          // [VPHYS][VW]
    float* sP = smem + VPHYS * VW;                              // [2][PSZ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    for (int i = tid; i < VPHYS * VW; i += 512) sV[i] = 0x3f803f80u;
    for (int i = tid; i < 2 * PSZ; i += 512) sP[i] = 0.5f;
    __syncthreads();
    f32x16 acc[PPW][COT];
#pragma unroll
    for (int p = 0; p < PPW; ++p)
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][c][r] = 0.0f;
    const unsigned* wbase = wts + (long)(wave * PPW) * (NQ * 256) + lane * 4;
    const float* xb = x + (long)(blockIdx.x % 64) * 16 * hw + tid * 4;
    // transform role: half a (tile, channel pair) unit
    const int s_tile = tid & 31, s_cp = (tid >> 5) & 7, s_hf = tid >> 8;
    const int s_ty = HY ? (s_tile >> 2) * 2 : (s_tile >> 3) * 2, s_tx = HY ? (s_tile & 3) * 4 : (s_tile & 7) * 2;
    float4 raw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) raw[i] = *reinterpret_cast<const float4*>(xb + (long)i * 2048);
    u32x4 aq[PPW][NQ];
    if (MODE & 1) {
#pragma unroll
        for (int s = 0; s < PPW; ++s)
#pragma unroll
            for (int q = 0; q < NQ; ++q) aq[s][q] = *reinterpret_cast<const u32x4*>(wbase + (s * NQ + q) * 256);
    }
    __syncthreads();
    const bool mfirst = wave < 4;            // the two waves of a SIMD (w, w + 4) run a phase in opposite orders
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();

    // ---- matrix side of local position s, chunk ch
    auto matrix = [&](int ch, int s) {
        if (!(MODE & 1)) return;
        const unsigned* q = sV + (VPHYS == 2 ? (ch & 1) * VW : 0) + (((PPW * wave + s) * 2 + half) * 4) * T + l31;
        u32x4 bq[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) bq[p][jp] = q[p * PWp + jp * T];
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int ct = 0; ct < COT; ++ct)
                acc[s][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aq[s][ct * 3 + PA[k]]), __builtin_bit_cast(bf16x8, bq[PB[k]]),
                                                                     acc[s][ct], 0, 0, 0);
        // the position's weights of the NEXT chunk into the registers the MFMAs have just read (prefetch distance: one chunk)
        const unsigned* wn = wbase + (long)((ch + 1) % 24) * wstride_chunk + s * (NQ * 256);
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) aq[s][qq] = *reinterpret_cast<const u32x4*>(wn + qq * 256);
    };

    // ---- staging side: part `part` of `nparts` of the thread's work for chunk ch + 1
    auto staging = [&](int ch, int part, int nparts) {
        if (!(MODE & 2)) return;
        const f32x2* sPr = reinterpret_cast<const f32x2*>(sP + (ch & 1) * PSZ) + (s_cp * PR + s_ty + s_hf) * PP + s_tx;
        unsigned* vdst = sV + (VPHYS == 2 ? ((ch + 1) & 1) * VW : 0) + ((s_hf * (NPOS / 2) * 2 + (s_cp & 1)) * 4 + (s_cp >> 1)) * T + s_tile;
        float* sPw = sP + ((ch + 1) & 1) * PSZ;
        if (!HY) {
            // F(2,3) x F(2,3): 3 rows x 4 columns of packed pairs -> 2 x 4 positions (this thread's output-row pair)
            const int r0 = part * (2 / (nparts > 2 ? 2 : nparts)), r1 = nparts == 1 ? 2 : r0 + 1;       // output rows of this part
            f32x2 rw[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { rw[0][j] = sPr[j]; rw[1][j] = sPr[PP + j]; rw[2][j] = sPr[2 * PP + j]; }
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                if (row < r0 || row >= r1) continue;
                f32x2 m[4], vv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = row == 0 ? rw[0][j] - rw[2][j] : rw[1][j] + rw[2][j];
                vv[0] = m[0] - m[2]; vv[1] = m[1] + m[2]; vv[2] = m[2] - m[1]; vv[3] = m[1] - m[3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned w1, w2, w3;
                    split3(vv[q], w1, w2, w3);
                    vdst[(row * 4 + q) * POSW] = w1;
                    vdst[(row * 4 + q) * POSW + PWp] = w2;
                    vdst[(row * 4 + q) * POSW + 2 * PWp] = w3;
                }
            }
        } else {
            // F(2,3) over the rows x F(4,3) over the columns: 3 rows x 6 columns of packed pairs -> 2 x 6 positions
            const int per = 2 * 6 / nparts;                       // positions of this part (12, or 4 per part at three parts)
            f32x2 rw[3][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) { rw[0][j] = sPr[j]; rw[1][j] = sPr[PP + j]; rw[2][j] = sPr[2 * PP + j]; }
            f32x2 m[2][6], vv[2][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) { m[0][j] = rw[0][j] - rw[2][j]; m[1][j] = rw[1][j] + rw[2][j]; }
            if (nparts == 1 || part < 2) bt43(m[nparts == 1 ? 0 : part], vv[nparts == 1 ? 0 : part]);
            if (nparts == 1) bt43(m[1], vv[1]);
            if (nparts == 3 && part == 2) {                       // the third part splits what parts 0 / 1 left: recompute is cheaper than keeping 12 pairs live
                bt43(m[0], vv[0]);
                bt43(m[1], vv[1]);
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const bool mine = nparts == 1 || (part < 2 ? (i / 6 == part && i % 6 < 4) : (i % 6 >= 4));
                if (!mine) continue;
                unsigned w1, w2, w3;
                split3(vv[i / 6][i % 6], w1, w2, w3);
                vdst[i * POSW] = w1;
                vdst[i * POSW + PWp] = w2;
                vdst[i * POSW + 2 * PWp] = w3;
            }
            (void)per;
        }
        // this part's share of the next chunk's patch: NPV values activated and parked
        {
            const float pv[12] = {raw[0].x, raw[0].y, raw[0].z, raw[0].w, raw[1].x, raw[1].y, raw[1].z, raw[1].w, raw[2].x, raw[2].y, raw[2].z, raw[2].w};
            const int e0 = part * NPV / nparts, e1 = (part + 1) * NPV / nparts;
#pragma unroll
            for (int e = 0; e < 12; ++e)
                if (e >= e0 && e < e1) sPw[(tid * 12 + e) % PSZ] = silu(pv[e] * 1.01f + 0.02f);
        }
        if (part == nparts - 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i) raw[i] = *reinterpret_cast<const float4*>(xb + (long)((ch + 2) % 32) * hw + (long)i * 2048);
        }
        (void)NRD; (void)NSP;
    };

    for (int ch = 0; ch < nch; ++ch) {
        if (NSYNC == 1) {
            if (mfirst) {
#pragma unroll
                for (int s = 0; s < PPW; ++s) matrix(ch, s);
                __builtin_amdgcn_sched_barrier(0);
                staging(ch, 0, 1);
            } else {
                staging(ch, 0, 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < PPW; ++s) matrix(ch, s);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
        } else {
#pragma unroll
            for (int s = 0; s < PPW; ++s) {
                if (mfirst) { matrix(ch, s); __builtin_amdgcn_sched_barrier(0); staging(ch, s, PPW); }
                else { staging(ch, s, PPW); __builtin_amdgcn_sched_barrier(0); matrix(ch, s); }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = raw[0].x + raw[2].w;
#pragma unroll
    for (int p = 0; p < PPW; ++p)
#pragma unroll
        for (int c = 0; c < COT; ++c) sum += acc[p][c][3];
    if (MODE & 1) sum += __builtin_bit_cast(float, aq[0][0][0]);
    if (lane == 0) out[blockIdx.x * 8 + wave] = (t1 - t0) + (sum == 123.456f);
}

template <int NPOS, int COT, int VPHYS, int NSYNC, int MODE>
static double run(const char* name, const unsigned* w, const float* x, unsigned long long* d, int nch) {
    constexpr int PPW = NPOS / 8, NQ = 3 * COT;
    constexpr int VW = 3 * NPOS * 256, PSZ = 8 * (NPOS == 24 ? 18 : 10) * PP * 2;
    const size_t lds = (size_t)(VPHYS * VW + 2 * PSZ) * 4;
    const long wstride = (long)NPOS * NQ * 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&kw<NPOS, COT, VPHYS, NSYNC, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((kw<NPOS, COT, VPHYS, NSYNC, MODE>), dim3(256), dim3(512), lds, 0, w, x, d, nch, wstride, 4096);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipError_t err = hipGetLastError();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += (double)v;
    m /= h.size();
    const double per_chunk = m / nch, px = NPOS == 24 ? 256.0 : 128.0;
    const int mfmas = 8 * PPW * 6 * COT;
    printf("%-74s %7.0f cycles / chunk  %6.2f cycles / (pixel x 32 couts)  %3d MFMAs / workgroup: %5.1f cycles per MFMA and SIMD  LDS %3zu KB  kernel %6.1f us  %s\n",
           name, per_chunk, per_chunk / (px * COT), mfmas, per_chunk / (mfmas / 4.0), lds / 1024, ms * 1e3, err == hipSuccess ? "" : hipGetErrorString(err));
    return per_chunk / (px * COT);
}

int main() {
    const int nch = 48;
    unsigned* w; float* x; unsigned long long* d;
    const size_t wbytes = (size_t)26 * 24 * 9 * 256 * 4 + (1 << 20);
    hipMalloc(&w, wbytes);
    hipMalloc(&x, (size_t)64 * 16 * 4096 * 4 * 2 + (1 << 24));
    hipMalloc(&d, 1 << 16);
    hipMemset(w, 0x3f, wbytes);
    hipMemset(x, 0x3c, (size_t)64 * 16 * 4096 * 4 * 2 + (1 << 24));
    printf("# two waves per SIMD (512 threads), 32 tiles x 32 COT couts per workgroup, 16-channel chunks, 256 workgroups, %d chunks\n", nch);
    printf("# ---- shipped form: F(2x2,3x3), 16 positions, 128 output pixels per chunk, two V buffers, one barrier per chunk\n");
    const double s3 = run<16, 3, 2, 1, 3>("S3  shipped form, COT = 3: everything", w, x, d, nch);
    run<16, 3, 2, 1, 1>("S3  matrix side only (MFMAs + weight stream + B reads)", w, x, d, nch);
    run<16, 3, 2, 1, 2>("S3  staging side only (patch + transform + split)", w, x, d, nch);
    const double s2 = run<16, 2, 2, 1, 3>("S2  shipped form, COT = 2: everything", w, x, d, nch);
    run<16, 3, 2, 1, 3>("S3  everything, again", w, x, d, nch);
    printf("# ---- hybrid: F(4,3) x F(2,3), 24 positions, 256 output pixels per chunk, COT = 2 (COT = 3 does not fit the registers)\n");
    const double h2 = run<24, 2, 1, 3, 3>("H2  ONE V buffer (what fits the LDS), three barriers per chunk: everything", w, x, d, nch);
    run<24, 2, 1, 3, 1>("H2  matrix side only", w, x, d, nch);
    run<24, 2, 1, 3, 2>("H2  staging side only", w, x, d, nch);
    const double h2x = run<24, 2, 1, 1, 3>("H2x the shipped structure on an LDS that does not exist (V buffers aliased): everything", w, x, d, nch);
    run<24, 2, 1, 3, 3>("H2  everything, again", w, x, d, nch);
    printf("# figure of merit: cycles per (output pixel x 32 couts).  GO line (stated before the run): hybrid <= 0.92 x shipped\n");
    printf("# H2  / S3 = %.3f   H2  / S2 = %.3f   (the hybrid that can be built)\n", h2 / s3, h2 / s2);
    printf("# H2x / S3 = %.3f   H2x / S2 = %.3f   (upper bound: double-buffered V that does not fit)\n", h2x / s3, h2x / s2);
    printf("# (+ for Cout = 96 layers the hybrid pads 96 -> 128 couts at COT = 2: x 1.333 on its per-pixel figure there)\n");
    return 0;
}
