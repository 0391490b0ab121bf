// VERDICT r4 item 1, measured: "separate the wave roles in the dominant kernel and MEASURE N = 64".
//
// What the hardware allows.  The verdict's design -- consumer waves holding 256 accumulator registers per SIMD beside two LOW-register
// producer waves -- cannot be launched: a dispatch has ONE register allocation (the kernel descriptor's granulated VGPR count), so every
// wave of the workgroup gets the consumer's 400+ registers and a SIMD (512 per lane) holds one of them.  The 64-tile x 96-cout tile
// (16 positions x 96 couts x 64 tiles of fp32 accumulators = 384 registers per SIMD lane) therefore has exactly one residency: ONE wave
// per SIMD that multiplies AND stages, the staging instructions placed in the shadows of its own MFMAs.  This program measures the
// K loop of that form, timing only (no results), next to its two halves, with the instruction mix of conv_wino3_kernel's chunk:
//   per 16-channel chunk and wave: 144 v_mfma_f32_32x32x16_bf16 (4 positions x 3 cout sub-tiles x 2 tile groups x 6 piece products),
//   36 weight quads (global_load_dwordx4 from an L2-resident layer image, every workgroup of a cout tile the same stream), 96 B-operand
//   dwords from LDS, and a quarter of the staging of 64 tiles x 16 channels: per thread 4 (tile, channel pair, row pair) transform units
//   (12 x 8-byte patch reads, row + column transform, eight 3-way bf16 splits, 24 dword stores) + 20 patch elements (load, affine,
//   SiLU, park).  Four barriers per chunk (a position's V is rewritten only after its owner has multiplied it: one V buffer of 96 KB).
// Modes: 0 everything, 1 MFMAs + weight stream + B reads only (the verdict's ablation (a)), 2 staging only (ablation (b), run by the
// same four waves), 3 MFMAs only.  Reference point (profiles/r04_wino3p_vs_wino3_layers.txt): today's kernel spends 4.7-5.6 k cycles
// per chunk on HALF this work (32 tiles: 288 MFMAs per workgroup), i.e. 9.4-11.2 k cycles for what one chunk is here.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench_wino64.cpp -o /tmp/ubench_wino64 && /tmp/ubench_wino64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int T = 64, CK = 16;
constexpr int PW = 16 * 2 * 4 * T;            // dwords of one piece plane of V: [position][half][pair][tile]
constexpr int VW = 3 * PW;                    // 24576 dwords = 96 KB
constexpr int PP = 24;                        // patch pitch (channel-pair columns), 18 x 18 patch
constexpr int PSZ = 8 * 18 * PP * 2;          // floats of one activated patch: [pair][18 rows][PP][2]

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

// MODE bits: 1 = matrix side (weights, B reads, MFMAs), 2 = staging side
template <int MODE, int COT>
__global__ __launch_bounds__(256, 1) void k64(const unsigned* __restrict__ wts, const float* __restrict__ x, unsigned long long* out, int nch,
                                             long wstride_chunk, int hw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sV = reinterpret_cast<unsigned*>(smem);          // [VW]
    float* sP = smem + VW;                                      // [2][PSZ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    for (int i = tid; i < VW; i += 256) sV[i] = 0x3f803f80u;
    for (int i = tid; i < 2 * PSZ; i += 256) sP[i] = 0.5f;
    __syncthreads();
    f32x16 acc[4][COT][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][c][t][r] = 0.0f;
    // weight stream of this wave: chunk ch, local position s: NQ = 3 COT quads (COT cout sub-tiles x 3 pieces) of 1 KB
    constexpr int NQ = 3 * COT;
    const unsigned* wbase = wts + ((blockIdx.x & 1) * 16 + wave * 4) * (NQ * 256) + lane * 4;
    // patch role: 20 elements per thread and chunk as five 16-byte loads
    const float* xb = x + (long)(blockIdx.x % 64) * 16 * hw + tid * 4;
    // transform role: unit u of the chunk's four: (tile, pair, row pair)
    const int s_tile = tid & 63, s_cp = (tid >> 6) * 2;         // + (u >> 1): pair; u & 1: row pair
    const int s_ty = s_tile >> 3, s_tx = s_tile & 7;
    float4 raw[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) raw[i] = *reinterpret_cast<const float4*>(xb + (long)i * 1024);
    u32x4 aq[NQ];
    if (MODE & 1) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) aq[q] = *reinterpret_cast<const u32x4*>(wbase + q * 256);
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // ---- matrix side: position 4 * wave + s of chunk ch
            u32x4 bq[2][3];
            if (MODE & 1) {
                const unsigned* q = sV + (((4 * wave + s) * 2 + half) * 4) * T + l31;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int jp = 0; jp < 4; ++jp) bq[tt][p][jp] = q[p * PW + jp * T + tt * 32];
            }
            // ---- staging side: transform unit s of chunk ch + 1 (reads the patch parked one chunk ago), a quarter of the next patch
            f32x2 rw[3][4];
            if (MODE & 2) {
                const int cp = s_cp + (s >> 1), rg = s & 1;
                const f32x2* sPr = reinterpret_cast<const f32x2*>(sP + (ch & 1) * PSZ) + (cp * 18 + 2 * s_ty + 2 * rg) * PP + 2 * s_tx;
#pragma unroll
                for (int j = 0; j < 4; ++j) { rw[0][j] = sPr[j]; rw[1][j] = sPr[PP + j]; rw[2][j] = sPr[2 * PP + j]; }
            }
            // ---- the sub-phase's MFMAs product by product (2 COT MFMAs each), a slice of the staging work behind each product, pinned by
            // scheduling barriers: the hand placement a real kernel would use (the compiler alone runs the two parts one after the other)
            static const int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
            const int cp = s_cp + (s >> 1), rg = s & 1;
            unsigned* vdst = sV + ((8 * rg * 2 + (cp & 1)) * 4 + (cp >> 1)) * T + s_tile;
            f32x2 mx[4], my[4], vv[4];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (MODE & 1) {
#pragma unroll
                    for (int ct = 0; ct < COT; ++ct)
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            // the 256 COT / 2 accumulator registers do not fit one register class (256 VGPRs + 256 AGPRs): left to itself the
                            // compiler shuffles them between the two and spills.  Tiles 0-15 are pinned to AGPRs, the rest to VGPRs.
                            if ((s * COT + ct) * 2 + tt < 16)
                                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[s][ct][tt]) : "v"(aq[ct * 3 + PA[k]]), "v"(bq[tt][PB[k]]));
                            else
                                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[s][ct][tt]) : "v"(aq[ct * 3 + PA[k]]), "v"(bq[tt][PB[k]]));
                        }
                    if (k == 5) {
                        // the next position's weights (chunk ch, s + 1; or chunk ch + 1, 0) into the registers the MFMAs have just read: the
                        // latency passes under the next sub-phase's B reads and first staging slices
                        const int nxt = ch * 4 + s + 1;
                        const unsigned* wn = wbase + (long)((nxt >> 2) % 24) * wstride_chunk + (nxt & 3) * (NQ * 256);
#pragma unroll
                        for (int qq = 0; qq < NQ; ++qq) aq[qq] = *reinterpret_cast<const u32x4*>(wn + qq * 256);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (MODE & 2) {
                    if (k == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x2 r0 = rw[0][j], r1 = rw[1][j], r2 = rw[2][j];
                            if (rg == 0) { mx[j] = r0 - r2; my[j] = r1 + r2; } else { mx[j] = r1 - r0; my[j] = r0 - r2; }
                        }
                    }
                    if (k == 0 || k == 3) {
                        const int row = k == 0 ? 0 : 1;
                        const f32x2 m0 = row ? my[0] : mx[0], m1 = row ? my[1] : mx[1], m2 = row ? my[2] : mx[2], m3 = row ? my[3] : mx[3];
                        vv[0] = m0 - m2; vv[1] = m1 + m2; vv[2] = m2 - m1; vv[3] = m1 - m3;
                    }
                    if (k == 1 || k == 2 || k == 4 || k == 5) {
                        const int row = k >= 4 ? 1 : 0, q0 = (k == 1 || k == 4) ? 0 : 2;
#pragma unroll
                        for (int q = q0; q < q0 + 2; ++q) {
                            unsigned w1, w2, w3;
                            split3(vv[q], w1, w2, w3);
                            vdst[(row * 4 + q) * 2 * 4 * T] = w1;
                            vdst[(row * 4 + q) * 2 * 4 * T + PW] = w2;
                            vdst[(row * 4 + q) * 2 * 4 * T + 2 * PW] = w3;
                        }
                    }
                    if (k == 3 || k == 5) {
                        // a quarter of the next chunk's patch: five elements activated and parked, two or three per slice
                        float* sPw = sP + ((ch + 1) & 1) * PSZ;
                        const float pv[5] = {raw[s][0], raw[s][1], raw[s][2], raw[s][3], raw[4][s]};
#pragma unroll
                        for (int e = (k == 3 ? 0 : 3); e < (k == 3 ? 3 : 5); ++e) sPw[(tid * 5 + e + s * 1280) % PSZ] = silu(pv[e] * 1.01f + 0.02f);
                    }
                    if (k == 5 && s == 3) {
#pragma unroll
                        for (int i = 0; i < 5; ++i) raw[i] = *reinterpret_cast<const float4*>(xb + (long)((ch + 2) % 32) * hw + (long)i * 1024);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // a position's V of chunk ch + 1 may be written once its owner has multiplied chunk ch's: one barrier per sub-phase
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = raw[0][0] + raw[4][3];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t) sum += acc[p][c][t][3];
    if (MODE & 1) sum += __builtin_bit_cast(float, aq[0][0]);
    if (lane == 0) out[blockIdx.x * 4 + wave] = (t1 - t0) + (sum == 123.456f);
}

template <int MODE, int COT>
static void run(const char* name, const unsigned* w, const float* x, unsigned long long* d, int nch, long wstride, int hw) {
    const size_t lds = (size_t)(VW + 2 * PSZ) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k64<MODE, COT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k64<MODE, COT>), dim3(256), dim3(256), lds, 0, w, x, d, nch, wstride, hw);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipError_t err = hipGetLastError();
    std::vector<unsigned long long> h(256 * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += (double)v;
    m /= h.size();
    printf("%-62s %8.0f cycles per chunk of 64 tiles (%d MFMAs per workgroup: %5.1f cycles per MFMA and SIMD)   kernel %7.1f us  clock %.2f GHz  %s\n", name,
           m / nch, 192 * COT, m / nch / (48.0 * COT), ms * 1e3, m / (ms * 1e-3) / 1e9, err == hipSuccess ? "" : hipGetErrorString(err));
}

int main() {
    const int nch = 48;                                  // eight 6-chunk items (96 -> 96 @64 x 64, B = 64: what a CU does per launch today: 16 items of 32 tiles)
    const long wstride = 16 * 9 * 256;                   // dwords per chunk of a cout tile: 16 positions x 9 quads x 1 KB = 147 KB
    unsigned* w; float* x; unsigned long long* d;
    hipMalloc(&w, (size_t)(24 + 1) * wstride * 4 + (1 << 20));
    hipMalloc(&x, (size_t)64 * 16 * 4096 * 4 * 2 + (1 << 22));
    hipMalloc(&d, 1 << 16);
    hipMemset(w, 0x3f, (size_t)(24 + 1) * wstride * 4 + (1 << 20));
    hipMemset(x, 0x3c, (size_t)64 * 16 * 4096 * 4 * 2 + (1 << 22));
    printf("# one wave per SIMD (256 threads, 512 registers per lane), 64 tiles x 32 COT couts x 16 positions per workgroup, 16-channel chunks\n");
    printf("# COT = 2 (64 couts: 256 accumulator registers per lane, all in AGPRs; 98 KB of weights per chunk)\n");
    run<3, 2>("everything (MFMAs + weights + B reads + staging)", w, x, d, nch, 16 * 6 * 256, 4096);
    run<1, 2>("matrix side only: MFMAs + weight stream + B reads   (a)", w, x, d, nch, 16 * 6 * 256, 4096);
    run<2, 2>("staging side only: patch + transform + split        (b)", w, x, d, nch, 16 * 6 * 256, 4096);
    run<3, 2>("everything, again", w, x, d, nch, 16 * 6 * 256, 4096);
    printf("# COT = 3 (96 couts: 384 accumulator registers per lane, 256 in AGPRs + 128 in VGPRs; 147 KB of weights per chunk)\n");
    run<3, 3>("everything (MFMAs + weights + B reads + staging)", w, x, d, nch, wstride, 4096);
    run<1, 3>("matrix side only: MFMAs + weight stream + B reads   (a)", w, x, d, nch, wstride, 4096);
    run<3, 3>("everything, again", w, x, d, nch, wstride, 4096);
    printf("# today's kernel (conv_wino3p_kernel<3,2>, two waves per SIMD, 32 tiles x 96 couts): 4.7-5.6 k cycles per 288-MFMA chunk = 65-78 cycles per MFMA and SIMD\n");
    return 0;
}
