// Time embedding: sinusoid -> Linear -> SiLU -> Linear (ncsnpp_more.py:273-280, layers.py:504-518), followed by the
// SiLU every Dense_0 applies to its input (layerspp.py:521); then ALL Dense_0 projections of the network in one
// GEMM-shaped launch (58 of them for a 4-level net), against a [K][N]-transposed concatenated weight matrix.
#include "../common.h"

namespace mcvd {

__device__ __forceinline__ float silu_t(float v) { return v / (1.0f + expf(-v)); }

constexpr int TEMB_NW = 8;       // waves per workgroup (256 registers per lane: sixteen weight vectors in flight)

// One workgroup of TEMB_NW waves per row of labels (the device loops label every sample alike: ONE row then serves the batch, model.cpp).
// nn.Linear weights are [out][in]: a wave takes eight outputs per pass and its lanes the input index (four consecutive inputs each when
// K % 4 == 0), so a weight row is read as contiguous 16-byte pieces.  The kernel is a chain of memory round trips at the head of every
// forward -- 0.2 M multiply-adds -- and what it costs is their number: the weight vectors of a pass (16 per lane) and its eight biases
// are requested first, all of them, then used; the partial sums cross the wave through the LDS.  rocprofv3, config 2: 57 us (one
// thread per output walking a weight row: nf + 4 nf dependent loads) -> 44 us (batched loads) -> 34 us (without the six dependent
// ds_bpermute per output); the rest is 12 passes of one L2 round trip each, the large-argument sinf / cosf and the cold first touch.
// The sum of one output is formed in the same order whatever the number of rows.
//
// out[o] = silu( sum_k in[k] * w[o * K + k] + bias[o] ), o < N; `in` in LDS (16-byte aligned), K <= 1152
__device__ __forceinline__ void temb_linear(const float* in, const float* w, const float* bias, float* out, float* red, int K, int N, int wave, int lane) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int NO = 8;
    for (int o0 = wave * NO; o0 < N; o0 += TEMB_NW * NO) {
        float acc[NO], bv[NO];
#pragma unroll
        for (int q = 0; q < NO; ++q) {
            acc[q] = 0.0f;
            bv[q] = bias[min(o0 + q, N - 1)];
        }
        if ((K & 3) == 0) {
            const int K4 = K >> 2;
            for (int k4 = lane; k4 < K4; k4 += 128) {          // two input quads per lane and round: 16 weight vectors in flight
                const int k4b = k4 + 64;
                const bool second = k4b < K4;
                f32x4 wa[NO], wb[NO];
#pragma unroll
                for (int q = 0; q < NO; ++q) {
                    const float* row = w + (long)min(o0 + q, N - 1) * K;
                    wa[q] = *reinterpret_cast<const f32x4*>(row + 4 * k4);
                    wb[q] = *reinterpret_cast<const f32x4*>(row + 4 * (second ? k4b : k4));
                }
                const f32x4 a = reinterpret_cast<const f32x4*>(in)[k4];
                f32x4 b = reinterpret_cast<const f32x4*>(in)[second ? k4b : k4];
                if (!second) b = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int q = 0; q < NO; ++q) {
                    acc[q] += (a[0] * wa[q][0] + a[1] * wa[q][1]) + (a[2] * wa[q][2] + a[3] * wa[q][3]);
                    acc[q] += (b[0] * wb[q][0] + b[1] * wb[q][1]) + (b[2] * wb[q][2] + b[3] * wb[q][3]);
                }
            }
        } else {
            for (int k = lane; k < K; k += 64) {
                const float a = in[k];
                float wv[NO];
#pragma unroll
                for (int q = 0; q < NO; ++q) wv[q] = w[(long)min(o0 + q, N - 1) * K + k];
#pragma unroll
                for (int q = 0; q < NO; ++q) acc[q] = fmaf(a, wv[q], acc[q]);
            }
        }
        // the eight lane-wise partial sums cross the wave through the LDS (its own 2 KB; LDS operations of a wave complete in order):
        // lane q adds the 64 partials of output o0 + q in lane order.  (Six dependent ds_bpermute per output -- __shfl_xor -- cost
        // 3 us a pass, 35 of the kernel's 44 us.)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          // the previous pass's reads are done
#pragma unroll
        for (int q = 0; q < NO; ++q) red[q * 64 + lane] = acc[q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < NO) {
            const f32x4* rq = reinterpret_cast<const f32x4*>(red + lane * 64);
            float t = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const f32x4 v = rq[j];
                t = (((t + v[0]) + v[1]) + v[2]) + v[3];
            }
            float bq = bv[0];
#pragma unroll
            for (int q = 1; q < NO; ++q) bq = lane == q ? bv[q] : bq;
            if (o0 + lane < N) out[o0 + lane] = silu_t(t + bq);
        }
    }
}

__global__ __launch_bounds__(64 * TEMB_NW) void temb_mlp_kernel(const void* labels, int labels_f32, const float* freqs, const float* w0,
                                                         const float* b0, const float* w1, const float* b1,
                                                         float* silu_temb, int nf, int out_stride, const float* emb_table,
                                                         const int32_t* mask) {
    __shared__ __attribute__((aligned(16))) float emb[256];
    __shared__ __attribute__((aligned(16))) float hid[1024];
    __shared__ __attribute__((aligned(16))) float red[TEMB_NW * 512];      // [wave][8 outputs][64 lanes]
    const int b = blockIdx.x;
    const int T = 4 * nf;
    const int half = nf / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Both weight matrices are cold at the head of a forward (0.7-1.3 MB that nothing else reads; HBM and TLB misses, ~5 us a round
    // trip) and the layers below fetch them in several dependent rounds: one dword of every 128-byte line (the first 1 MB of w1) is requested
    // HERE, all at once, so that those rounds hit the L2.  (The values are not used; the empty asm below keeps the loads.)
    float pfv[24];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const long i = ((long)threadIdx.x + q * 64 * TEMB_NW) * 32;
        pfv[q] = i < (long)T * T ? w1[i] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const long i = ((long)threadIdx.x + q * 64 * TEMB_NW) * 32;
        pfv[16 + q] = i < (long)T * nf ? w0[i] : 0.0f;
    }
    // timesteps.float(): int64 labels (samplers) or already-float, possibly fractional ones (F-PNDM's (t + t_next) / 2)
    const float t = labels_f32 ? static_cast<const float*>(labels)[b] : (float)static_cast<const int64_t*>(labels)[b];
    for (int i = threadIdx.x; i < nf; i += 64 * TEMB_NW) {
        float v = 0.0f;                                     // odd nf: zero pad (layers.py:515-516)
        if (i < half) v = sinf(t * freqs[i]);
        else if (i < 2 * half) v = cosf(t * freqs[i - half]);
        emb[i] = v;
    }
    __syncthreads();
    temb_linear(emb, w0, b0, hid, red + wave * 512, nf, T, wave, lane);                              // Linear + SiLU
    __syncthreads();
    temb_linear(hid, w1, b1, silu_temb + (long)b * out_stride, red + wave * 512, T, T, wave, lane); // Linear, then the SiLU of every Dense_0
    _Pragma("unroll") for (int q = 0; q < 24; ++q) asm volatile("" :: "v"(pfv[q]));
    if (emb_table) {        // cond_emb: temb = cat([temb, Embedding(cond_mask)]) (ncsnpp_more.py:282-285); Dense_0 sees SiLU of it
        const int mb = mask ? (mask[b] != 0 ? 1 : 0) : 1;
        for (int j = threadIdx.x; j < half; j += 64 * TEMB_NW) silu_temb[(long)b * out_stride + T + j] = silu_t(emb_table[mb * half + j]);
    }
}

int launch_temb_mlp(const void* labels, int labels_f32, const float* freqs, const float* w0, const float* b0, const float* w1,
                    const float* b1, float* silu_temb, int B, int nf, int out_stride, const float* emb_table, const int32_t* mask,
                    hipStream_t s) {
    MCVD_REQUIRE(nf <= 256 && nf >= 4, "temb: ngf=%d out of range [4,256]", nf);
    hipLaunchKernelGGL(temb_mlp_kernel, dim3(B), dim3(64 * TEMB_NW), 0, s, labels, labels_f32, freqs, w0, b0, w1, b1, silu_temb, nf,
                       out_stride, emb_table, mask);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

// out[b][n] = bias[n] + sum_k act[b][k] * wt[k][n]: a GEMV per row (ONE row in the device loops), 11 k outputs x K = 384..576 for a
// 4-level net: 17-25 MB of weights per forward.  Block = 64 outputs (16 lanes x float4) x 16 slices of K; every slice keeps its whole
// share of weight rows in flight (K / 16 <= 72 float4 loads per thread, unrolled by 8), the slices are summed in index order through
// the LDS -- the order of a row's sum does not depend on the number of rows.  R rows share one pass over the weights.  (Until round 4:
// 256 outputs per block, one thread per output walking all of K: 44 blocks, 53-71 us.)
template <int R>
__global__ __launch_bounds__(256) void dense_all_kernel(const float* __restrict__ act, const float* __restrict__ wt,
                                                         const float* __restrict__ bias, float* __restrict__ out, int B, int K, int N) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ f32x4 red[16][R][16];
    const int l = threadIdx.x & 15, ks = threadIdx.x >> 4;
    const int n0 = (blockIdx.x * 16 + l) * 4;                   // N % 4 == 0 (every entry is 2 * channels)
    const int b0 = blockIdx.y * R;
    const int kper = (K + 15) / 16, k0 = ks * kper, k1 = min(K, k0 + kper);
    const int nc = min(n0, N - 4);                              // ragged last block: valid addresses, the stores are predicated
    f32x4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
    for (int k = k0; k < k1; ++k) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(wt + (long)k * N + nc);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float a = act[(long)min(b0 + r, B - 1) * K + k];
            acc[r] = acc[r] + a * w;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) red[ks][r][l] = acc[r];
    __syncthreads();
    // thread (l, r = ks) of the first R slices finishes row r
    if (ks < R && n0 < N && b0 + ks < B) {
        f32x4 t = red[0][ks][l];
#pragma unroll
        for (int q = 1; q < 16; ++q) t = t + red[q][ks][l];
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n0);
        *reinterpret_cast<f32x4*>(out + (long)(b0 + ks) * N + n0) = t + bv;
    }
}

int launch_dense_all(const float* act, const float* wt, const float* bias, float* out, int B, int K, int N,
                     hipStream_t s) {
    // N = NE = sum over the Dense_0 entries of 2 * channels, channels = ngf * ch_mult (+ a skip's): build_plan admits ngf % 4 == 0 only
    // (model.cpp), so every entry starts at a multiple of 8 floats and N % 4 == 0 holds for every model the plan accepts (ADVICE r4 asked
    // whether the float4 rows narrowed the supported widths: they did not -- the same plan check predates them).
    MCVD_REQUIRE(K <= 1152 && N % 4 == 0 && N >= 4, "dense_all: K=%d N=%d (N is a sum of 2 * channels with ngf %% 4 == 0)", K, N);
    const int nb = (N / 4 + 15) / 16;
    if (B == 1)
        hipLaunchKernelGGL(dense_all_kernel<1>, dim3(nb, 1), dim3(256), 0, s, act, wt, bias, out, B, K, N);
    else
        hipLaunchKernelGGL(dense_all_kernel<8>, dim3(nb, (B + 7) / 8), dim3(256), 0, s, act, wt, bias, out, B, K, N);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

// wt[c][col_off + r] = w[r][c]   (w: [rows][cols] row-major; wt leading dimension ld_out)
__global__ void transpose_into_kernel(const float* w, float* wt, int rows, int cols, int ld_out, int col_off) {
    const long n = (long)rows * cols;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols), r = (int)(i / cols);
        wt[(long)c * ld_out + col_off + r] = w[i];
    }
}

int launch_transpose_into(const float* w, float* wt, int rows, int cols, int ld_out, int col_off, hipStream_t s) {
    const long n = (long)rows * cols;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(transpose_into_kernel, dim3(blocks), dim3(256), 0, s, w, wt, rows, cols, ld_out, col_off);
    MCVD_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mcvd
